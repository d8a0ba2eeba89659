// ORACLE tooling -- test infrastructure only.
//
// hlsl_resources.h: HLSL resource objects (Texture2D / RWTexture2D / Texture3D / StructuredBuffer / SamplerState /
// ResourceDescriptorHeap) over plain host memory, for the reference shader code compiled as C++ (see hlsl_shim.h).
// Typed loads / stores convert between the plane's DXGI storage format and the shader's element type with the ABI's
// conversions (include/zr_detmath.h: RTNE half / R11G11B10F, UNORM = (uint)fma(saturate(x), 2^n - 1, 0.5), x / 255 decode);
// filtered sampling is the ABI's software filtering (include/zr_texture.h; the rho LUT: fp32 trilinear, clamp, texel centres at
// (i + 0.5) / N, as oracle/zro_bsdf.h states it).  These are the arithmetic the reference leaves to the hardware.
#pragma once
#include "hlsl_shim.h"
#include "../../include/zr_texture.h"

namespace hlsl {

enum TexFormat
{
    FMT_UNKNOWN = 0,
    FMT_R8_UINT, FMT_RG8_UINT, FMT_RGBA8_UINT, FMT_R8_UNORM, FMT_RG8_UNORM, FMT_RGBA8_UNORM,
    FMT_R16_UINT, FMT_RG16_UINT, FMT_RGBA16_UINT, FMT_R16_UNORM, FMT_RG16_UNORM, FMT_RG16_SNORM,
    FMT_R16_FLOAT, FMT_RG16_FLOAT, FMT_RGBA16_FLOAT,
    FMT_R32_UINT, FMT_RG32_UINT, FMT_RGBA32_UINT, FMT_R32_FLOAT, FMT_RG32_FLOAT, FMT_RGBA32_FLOAT,
    FMT_R11G11B10_FLOAT, FMT_R9G9B9E5,
    FMT_MATERIAL_TEXTURE          // an entry of the scene's zr_tex_heap (decoded RGBA8 / RG8 texels with mips; sampled through zr_texture.h)
};

struct TexStorage
{
    void* data = nullptr;
    const void* readData = nullptr;     // when set, loads come from this snapshot and stores go to `data` (race-free in-place passes: FireflyFilter.hlsl)
    uint32_t w = 0, h = 0, d = 1;
    int fmt = FMT_UNKNOWN;
    const zr_tex_heap* heap = nullptr; uint32_t heapIdx = 0;     // FMT_MATERIAL_TEXTURE
};

static inline uint32_t FormatBytes(int f)
{
    switch (f)
    {
    case FMT_R8_UINT: case FMT_R8_UNORM: return 1;
    case FMT_RG8_UINT: case FMT_RG8_UNORM: case FMT_R16_UINT: case FMT_R16_UNORM: case FMT_R16_FLOAT: return 2;
    case FMT_RGBA8_UINT: case FMT_RGBA8_UNORM: case FMT_RG16_UINT: case FMT_RG16_UNORM: case FMT_RG16_SNORM: case FMT_RG16_FLOAT:
    case FMT_R32_UINT: case FMT_R32_FLOAT: case FMT_R11G11B10_FLOAT: case FMT_R9G9B9E5: return 4;
    case FMT_RGBA16_UINT: case FMT_RGBA16_FLOAT: case FMT_RG32_UINT: case FMT_RG32_FLOAT: return 8;
    case FMT_RGBA32_UINT: case FMT_RGBA32_FLOAT: return 16;
    default: return 0;
    }
}
static inline bool FormatIsUint(int f)
{ return f == FMT_R8_UINT || f == FMT_RG8_UINT || f == FMT_RGBA8_UINT || f == FMT_R16_UINT || f == FMT_RG16_UINT || f == FMT_RGBA16_UINT || f == FMT_R32_UINT || f == FMT_RG32_UINT || f == FMT_RGBA32_UINT; }

// raw element -> 4 lanes (floats or uints, by format class); missing channels read (0, 0, 0, 1) like D3D
static inline void LoadRaw(const TexStorage& s, size_t idx, float f[4], uint32_t u[4])
{
    const uint8_t* p = (const uint8_t*)(s.readData ? s.readData : s.data) + idx * FormatBytes(s.fmt);
    f[0] = f[1] = f[2] = 0.0f; f[3] = 1.0f; u[0] = u[1] = u[2] = 0u; u[3] = 1u;
    uint16_t h[4]; uint32_t w[4];
    switch (s.fmt)
    {
    case FMT_R8_UINT: u[0] = p[0]; break;
    case FMT_RG8_UINT: u[0] = p[0]; u[1] = p[1]; break;
    case FMT_RGBA8_UINT: for (int i = 0; i < 4; i++) u[i] = p[i]; break;
    case FMT_R8_UNORM: f[0] = zr_div255((float)p[0]); break;
    case FMT_RG8_UNORM: f[0] = zr_div255((float)p[0]); f[1] = zr_div255((float)p[1]); break;
    case FMT_RGBA8_UNORM: for (int i = 0; i < 4; i++) f[i] = zr_div255((float)p[i]); break;
    case FMT_R16_UINT: memcpy(h, p, 2); u[0] = h[0]; break;
    case FMT_RG16_UINT: memcpy(h, p, 4); u[0] = h[0]; u[1] = h[1]; break;
    case FMT_RGBA16_UINT: memcpy(h, p, 8); for (int i = 0; i < 4; i++) u[i] = h[i]; break;
    case FMT_R16_UNORM: memcpy(h, p, 2); f[0] = zr_div65535((float)h[0]); break;
    case FMT_RG16_UNORM: memcpy(h, p, 4); f[0] = zr_div65535((float)h[0]); f[1] = zr_div65535((float)h[1]); break;
    case FMT_RG16_SNORM: { int16_t sv[2]; memcpy(sv, p, 4); for (int i = 0; i < 2; i++) { float v = (float)sv[i] / 32767.0f; f[i] = v < -1.0f ? -1.0f : v; } break; }
    case FMT_R16_FLOAT: memcpy(h, p, 2); f[0] = zr_f16_to_f32(h[0]); break;
    case FMT_RG16_FLOAT: memcpy(h, p, 4); f[0] = zr_f16_to_f32(h[0]); f[1] = zr_f16_to_f32(h[1]); break;
    case FMT_RGBA16_FLOAT: memcpy(h, p, 8); for (int i = 0; i < 4; i++) f[i] = zr_f16_to_f32(h[i]); break;
    case FMT_R32_UINT: memcpy(w, p, 4); u[0] = w[0]; break;
    case FMT_RG32_UINT: memcpy(w, p, 8); u[0] = w[0]; u[1] = w[1]; break;
    case FMT_RGBA32_UINT: memcpy(w, p, 16); for (int i = 0; i < 4; i++) u[i] = w[i]; break;
    case FMT_R32_FLOAT: memcpy(f, p, 4); break;
    case FMT_RG32_FLOAT: memcpy(f, p, 8); break;
    case FMT_RGBA32_FLOAT: memcpy(f, p, 16); break;
    case FMT_R11G11B10_FLOAT: memcpy(w, p, 4); f[0] = zr_unpack_ufloat(w[0] & 0x7ffu, 6); f[1] = zr_unpack_ufloat((w[0] >> 11) & 0x7ffu, 6); f[2] = zr_unpack_ufloat(w[0] >> 22, 5); break;
    case FMT_R9G9B9E5:       // shared exponent: mantissa * 2^(e - 15 - 9), exact in fp32
    {
        memcpy(w, p, 4);
        const float sc = zr_asfloat((uint32_t)((int)(w[0] >> 27) - 15 - 9 + 127) << 23);
        f[0] = (float)(w[0] & 0x1ffu) * sc; f[1] = (float)((w[0] >> 9) & 0x1ffu) * sc; f[2] = (float)((w[0] >> 18) & 0x1ffu) * sc;
        break;
    }
    default: break;
    }
}

template<class T> struct Lanes;       // element type <-> 4 lanes
template<> struct Lanes<float> { static float get(const float* f, const uint32_t*) { return f[0]; } static void put(float v, float* f, uint32_t*) { f[0] = v; } };
template<> struct Lanes<uint32_t> { static uint32_t get(const float*, const uint32_t* u) { return u[0]; } static void put(uint32_t v, float*, uint32_t* u) { u[0] = v; } };
template<> struct Lanes<uint16_t> { static uint16_t get(const float*, const uint32_t* u) { return (uint16_t)u[0]; } static void put(uint16_t v, float*, uint32_t* u) { u[0] = v; } };
template<> struct Lanes<half> { static half get(const float* f, const uint32_t*) { return half(f[0]); } static void put(half v, float* f, uint32_t*) { f[0] = (float)v; } };
template<int N> struct Lanes<vec<float, N>> { static vec<float, N> get(const float* f, const uint32_t*) { vec<float, N> r; for (int i = 0; i < N; i++) r.d[i] = f[i]; return r; }
    static void put(const vec<float, N>& v, float* f, uint32_t*) { for (int i = 0; i < N; i++) f[i] = v.d[i]; } };
template<int N> struct Lanes<vec<half, N>> { static vec<half, N> get(const float* f, const uint32_t*) { vec<half, N> r; for (int i = 0; i < N; i++) r.d[i] = half(f[i]); return r; }
    static void put(const vec<half, N>& v, float* f, uint32_t*) { for (int i = 0; i < N; i++) f[i] = (float)v.d[i]; } };
template<int N> struct Lanes<vec<uint32_t, N>> { static vec<uint32_t, N> get(const float*, const uint32_t* u) { vec<uint32_t, N> r; for (int i = 0; i < N; i++) r.d[i] = u[i]; return r; }
    static void put(const vec<uint32_t, N>& v, float*, uint32_t* u) { for (int i = 0; i < N; i++) u[i] = v.d[i]; } };
template<int N> struct Lanes<vec<uint16_t, N>> { static vec<uint16_t, N> get(const float*, const uint32_t* u) { vec<uint16_t, N> r; for (int i = 0; i < N; i++) r.d[i] = (uint16_t)u[i]; return r; }
    static void put(const vec<uint16_t, N>& v, float*, uint32_t* u) { for (int i = 0; i < N; i++) u[i] = v.d[i]; } };

// float -> unsigned small float with `mbits` mantissa bits, 5 exponent bits (the channels of R11G11B10_FLOAT), round to nearest even:
// the ABI's store rule (same statement as oracle/zro_math.h PackUFloat)
static inline uint32_t PackUFloat(float f, int mbits)
{
    uint32_t x = zr_asuint(f);
    if (x & 0x80000000u) return 0;
    if (x >= 0x7f800000u) return x > 0x7f800000u ? ((0x1fu << mbits) | 1u) : (0x1fu << mbits);
    const int shift = 23 - mbits;
    if (x >= 0x47800000u) return (0x1eu << mbits) | ((1u << mbits) - 1u);
    if (x < 0x38800000u)
    {
        if (x < 0x33000000u) return 0;
        uint32_t e = x >> 23;
        uint32_t m = (x & 0x007fffffu) | 0x00800000u;
        uint32_t sh = (uint32_t)shift + (113u - e);
        if (sh > 24) return 0;
        uint32_t r = m >> sh, rem = m & ((1u << sh) - 1u), hf = 1u << (sh - 1u);
        if (rem > hf || (rem == hf && (r & 1u))) r++;
        return r;
    }
    uint32_t r = (x - 0x38000000u) >> shift;
    uint32_t rem = x & ((1u << shift) - 1u), hf = 1u << (shift - 1);
    if (rem > hf || (rem == hf && (r & 1u))) r++;
    uint32_t maxv = (0x1eu << mbits) | ((1u << mbits) - 1u);
    return r > maxv ? maxv : r;
}

static inline void StoreRaw(const TexStorage& s, size_t idx, const float f[4], const uint32_t u[4])
{
    uint8_t* p = (uint8_t*)s.data + idx * FormatBytes(s.fmt);
    uint16_t h[4]; uint32_t w[4];
    auto un8 = [](float x) { return (uint8_t)(uint32_t)zr_fma(zr_saturate(x), 255.0f, 0.5f); };
    auto un16 = [](float x) { return (uint16_t)(uint32_t)zr_fma(zr_saturate(x), 65535.0f, 0.5f); };
    switch (s.fmt)
    {
    case FMT_R8_UINT: p[0] = (uint8_t)u[0]; break;
    case FMT_RG8_UINT: p[0] = (uint8_t)u[0]; p[1] = (uint8_t)u[1]; break;
    case FMT_RGBA8_UINT: for (int i = 0; i < 4; i++) p[i] = (uint8_t)u[i]; break;
    case FMT_R8_UNORM: p[0] = un8(f[0]); break;
    case FMT_RG8_UNORM: p[0] = un8(f[0]); p[1] = un8(f[1]); break;
    case FMT_RGBA8_UNORM: for (int i = 0; i < 4; i++) p[i] = un8(f[i]); break;
    case FMT_R16_UINT: h[0] = (uint16_t)u[0]; memcpy(p, h, 2); break;
    case FMT_RG16_UINT: h[0] = (uint16_t)u[0]; h[1] = (uint16_t)u[1]; memcpy(p, h, 4); break;
    case FMT_RGBA16_UINT: for (int i = 0; i < 4; i++) h[i] = (uint16_t)u[i]; memcpy(p, h, 8); break;
    case FMT_R16_UNORM: h[0] = un16(f[0]); memcpy(p, h, 2); break;
    case FMT_RG16_UNORM: h[0] = un16(f[0]); h[1] = un16(f[1]); memcpy(p, h, 4); break;
    case FMT_RG16_SNORM:
        for (int i = 0; i < 2; i++)
        {   // round half away from zero (ABI)
            float c = f[i]; c = zr_isnan(c) ? 0.0f : (c < -1.0f ? -1.0f : (c > 1.0f ? 1.0f : c));
            c = c * 32767.0f; c = c >= 0.0f ? c + 0.5f : c - 0.5f;
            int16_t q = (int16_t)(int32_t)c; memcpy(p + 2 * i, &q, 2);
        }
        break;
    case FMT_R16_FLOAT: h[0] = zr_f32_to_f16(f[0]); memcpy(p, h, 2); break;
    case FMT_RG16_FLOAT: h[0] = zr_f32_to_f16(f[0]); h[1] = zr_f32_to_f16(f[1]); memcpy(p, h, 4); break;
    case FMT_RGBA16_FLOAT: for (int i = 0; i < 4; i++) h[i] = zr_f32_to_f16(f[i]); memcpy(p, h, 8); break;
    case FMT_R32_UINT: memcpy(p, u, 4); break;
    case FMT_RG32_UINT: memcpy(p, u, 8); break;
    case FMT_RGBA32_UINT: memcpy(p, u, 16); break;
    case FMT_R32_FLOAT: memcpy(p, f, 4); break;
    case FMT_RG32_FLOAT: memcpy(p, f, 8); break;
    case FMT_RGBA32_FLOAT: memcpy(p, f, 16); break;
    case FMT_R11G11B10_FLOAT: w[0] = PackUFloat(f[0], 6) | (PackUFloat(f[1], 6) << 11) | (PackUFloat(f[2], 5) << 22); memcpy(p, w, 4); break;
    default: break;
    }
}

struct SamplerState
{
    enum Kind { MIP0, POINT_WRAP, POINT_CLAMP, LINEAR_WRAP, LINEAR_CLAMP, ANISO_WRAP, ANISO_WRAP_2X, ANISO_WRAP_4X, IMGUI };
    int kind = LINEAR_WRAP;
    static SamplerState Named(const char* n)
    {
        SamplerState s;
        if (!strcmp(n, "g_samMip0")) s.kind = MIP0; else if (!strcmp(n, "g_samPointWrap")) s.kind = POINT_WRAP;
        else if (!strcmp(n, "g_samPointClamp")) s.kind = POINT_CLAMP; else if (!strcmp(n, "g_samLinearWrap")) s.kind = LINEAR_WRAP;
        else if (!strcmp(n, "g_samLinearClamp")) s.kind = LINEAR_CLAMP; else if (!strcmp(n, "g_samAnisotropicWrap")) s.kind = ANISO_WRAP;
        else if (!strcmp(n, "g_samAnisotropicWrap_2x")) s.kind = ANISO_WRAP_2X; else if (!strcmp(n, "g_samAnisotropicWrap_4x")) s.kind = ANISO_WRAP_4X;
        else s.kind = IMGUI;
        return s;
    }
};

// SamplerDescriptorHeap[i]: the drivers number the samplers like SamplerState::Kind
struct SamplerHeapRef { SamplerState operator[](uint32_t i) const { SamplerState s; s.kind = (int)i; return s; } };
static const SamplerHeapRef SamplerDescriptorHeap;

struct HeapHandle { TexStorage* s; };
struct DescriptorHeap
{
    static constexpr uint32_t kSize = 8192;
    TexStorage table[kSize];
    HeapHandle operator[](uint32_t i) { return HeapHandle{&table[i < kSize ? i : kSize - 1]}; }
};
static thread_local DescriptorHeap* g_heapPtr = nullptr;
struct DescriptorHeapRef { HeapHandle operator[](uint32_t i) const { return (*g_heapPtr)[i]; } };
static const DescriptorHeapRef ResourceDescriptorHeap;

// see RWTexture2D below: the one pending read-modify-write proxy of this thread
struct PendingRW { void (*flush)(void*) = nullptr; void* obj = nullptr; };
static thread_local PendingRW g_pendingRW;
static inline void FlushPendingRW() { if (g_pendingRW.flush) { void (*f)(void*) = g_pendingRW.flush; g_pendingRW.flush = nullptr; f(g_pendingRW.obj); } }

template<class T> struct Texture2D
{
    TexStorage* s = nullptr;
    Texture2D() {}
    Texture2D(HeapHandle h) : s(h.s) {}
    T LoadPx(uint32_t x, uint32_t y) const
    {
        FlushPendingRW();
        float f[4]; uint32_t u[4];
        if (x >= s->w || y >= s->h) { f[0] = f[1] = f[2] = f[3] = 0.0f; u[0] = u[1] = u[2] = u[3] = 0u; return Lanes<T>::get(f, u); }   // out-of-bounds loads return 0
        LoadRaw(*s, (size_t)y * s->w + x, f, u);
        return Lanes<T>::get(f, u);
    }
    T operator[](const uint2& p) const { return LoadPx(p.x, p.y); }
    T operator[](const int2& p) const { return LoadPx((uint32_t)p.x, (uint32_t)p.y); }
    T operator[](const uint16_t2& p) const { return LoadPx(p.x, p.y); }
    T operator[](const int16_t2& p) const { return LoadPx((uint32_t)(int32_t)p.x, (uint32_t)(int32_t)p.y); }
    template<int M, int A, int B> T operator[](const Swz<uint32_t, M, A, B>& p) const { uint2 q = p; return LoadPx(q.x, q.y); }
    template<int M, int A, int B> T operator[](const Swz<int32_t, M, A, B>& p) const { int2 q = p; return LoadPx((uint32_t)q.x, (uint32_t)q.y); }
    T Load(const int3& p) const { return LoadPx((uint32_t)p.x, (uint32_t)p.y); }
    void GetDimensions(uint32_t& w, uint32_t& h) const { w = s->w; h = s->h; }
    void GetDimensions(float& w, float& h) const { w = (float)s->w; h = (float)s->h; }
    // ---- filtered sampling
    T Bilinear(float u, float v, bool wrap) const
    {
        FlushPendingRW();
        // texel centres at (i + 0.5) / N, fp32 weights, lerp as a + t (b - a); wrap or clamp addressing
        const float W = (float)s->w, H = (float)s->h;
        if (wrap) { u = zr_tex_wrap(u); v = zr_tex_wrap(v); }
        const float x = u * W - 0.5f, y = v * H - 0.5f;
        const float fx = zr_floor(x), fy = zr_floor(y);
        const float tx = x - fx, ty = y - fy;
        int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
        auto fix = [&](int c, int n) { if (wrap) { c = c % n; return c < 0 ? c + n : c; } return c < 0 ? 0 : (c >= n ? n - 1 : c); };
        x0 = fix(x0, (int)s->w); x1 = fix(x1, (int)s->w); y0 = fix(y0, (int)s->h); y1 = fix(y1, (int)s->h);
        float c[4][4]; uint32_t uu[4];
        LoadRaw(*s, (size_t)y0 * s->w + x0, c[0], uu); LoadRaw(*s, (size_t)y0 * s->w + x1, c[1], uu);
        LoadRaw(*s, (size_t)y1 * s->w + x0, c[2], uu); LoadRaw(*s, (size_t)y1 * s->w + x1, c[3], uu);
        float o[4];
        for (int k = 0; k < 4; k++)
        {
            const float top = c[0][k] + tx * (c[1][k] - c[0][k]);
            const float bot = c[2][k] + tx * (c[3][k] - c[2][k]);
            o[k] = top + ty * (bot - top);
        }
        uu[0] = uu[1] = uu[2] = uu[3] = 0u;
        return Lanes<T>::get(o, uu);
    }
    T SampleLevel(const SamplerState& sam, const float2& uv, float lod) const
    {
        float o[4]; uint32_t uu[4] = {0, 0, 0, 0};
        if (s->fmt == FMT_MATERIAL_TEXTURE)
        {
            if (sam.kind == SamplerState::POINT_WRAP || sam.kind == SamplerState::POINT_CLAMP) zr_tex_point(s->heap, s->heapIdx, uv.x, uv.y, o);
            else zr_tex_sample_level(s->heap, s->heapIdx, uv.x, uv.y, sam.kind == SamplerState::MIP0 ? 0.0f : lod, o);
            return Lanes<T>::get(o, uu);
        }
        if (sam.kind == SamplerState::POINT_WRAP || sam.kind == SamplerState::POINT_CLAMP)
        {
            const bool wrap = sam.kind == SamplerState::POINT_WRAP;
            float u = wrap ? zr_tex_wrap(uv.x) : zr_saturate(uv.x), v = wrap ? zr_tex_wrap(uv.y) : zr_saturate(uv.y);
            uint32_t x = (uint32_t)(u * (float)s->w), y = (uint32_t)(v * (float)s->h);
            x = x < s->w ? x : s->w - 1; y = y < s->h ? y : s->h - 1;
            return LoadPx(x, y);
        }
        return Bilinear(uv.x, uv.y, sam.kind != SamplerState::LINEAR_CLAMP);
    }
    T SampleGrad(const SamplerState& sam, const float2& uv, const float2& ddx, const float2& ddy) const
    {
        float o[4]; uint32_t uu[4] = {0, 0, 0, 0};
        // the ABI's software filtering (zr_texture.h): MaxAnisotropy per sampler (RendererCore.cpp:450-553), g_samMip0 clamps to mip 0
        const uint32_t filter = sam.kind == SamplerState::MIP0 ? ZR_TEX_FILTER_MIP0 : (sam.kind == SamplerState::ANISO_WRAP ? ZR_TEX_FILTER_ANISOTROPIC_16X :
            (sam.kind == SamplerState::ANISO_WRAP_2X ? ZR_TEX_FILTER_ANISOTROPIC_2X : (sam.kind == SamplerState::ANISO_WRAP_4X ? ZR_TEX_FILTER_ANISOTROPIC_4X : ZR_TEX_FILTER_TRI_LINEAR)));
        zr_tex_sample_grad_filter(s->heap, s->heapIdx, filter, uv.x, uv.y, ddx.x, ddx.y, ddy.x, ddy.y, o);
        return Lanes<T>::get(o, uu);
    }
};

// RWTexture2D<T>::operator[] -> an lvalue proxy: reads as T (so swizzles work); whole-element assignment and partial writes through a
// swizzle (`g_out[px].rgb = c`, GBufferRT.hlsli:120; `g_outA[px].x = v`, Reservoir.hlsli:354) both land in the plane.  The proxy lives in
// per-thread storage and encodes its value back when it changed; at most one proxy is pending, and every resource access, fiber switch
// and dispatch end flushes it first (FlushPendingRW), so no reader ever sees a stale element.

template<class T, bool IsClass = std::is_class<T>::value && !std::is_same<T, half>::value> struct RWRef;
template<class T> struct RWRef<T, true> : T
{
    TexStorage* s = nullptr; size_t idx = 0; bool ok = false; T orig;
    void Bind(TexStorage* st, size_t i, bool inBounds, const T& v) { T::operator=(v); s = st; idx = i; ok = inBounds; orig = v; }
    void Store(const T& v) { float f[4] = {0, 0, 0, 0}; uint32_t u[4] = {0, 0, 0, 0}; Lanes<T>::put(v, f, u); StoreRaw(*s, idx, f, u); }
    static void FlushFn(void* p)
    {
        RWRef* r = (RWRef*)p; const T& cur = *r;
        if (r->ok && memcmp(&cur, &r->orig, sizeof(T)) != 0) { r->Store(cur); r->orig = cur; }
    }
    RWRef& operator=(const T& v) { T::operator=(v); if (ok) Store(v); orig = v; return *this; }     // a whole-element store always writes
    RWRef& operator=(const RWRef& o) { const T v = (const T&)o; return *this = v; }
};
template<class T> struct RWRef<T, false>
{
    TexStorage* s = nullptr; size_t idx = 0; bool ok = false; T val;
    void Bind(TexStorage* st, size_t i, bool inBounds, const T& v) { s = st; idx = i; ok = inBounds; val = v; }
    static void FlushFn(void*) {}
    operator T() const { return val; }
    RWRef& operator=(const T& v) { val = v; if (ok) { float f[4] = {0, 0, 0, 0}; uint32_t u[4] = {0, 0, 0, 0}; Lanes<T>::put(v, f, u); StoreRaw(*s, idx, f, u); } return *this; }
    RWRef& operator=(const RWRef& o) { const T v = o.val; return *this = v; }
    RWRef& operator+=(const T& v) { return *this = (T)(val + v); }
};
template<class T> struct RWTexture2D
{
    TexStorage* s = nullptr;
    RWTexture2D() {}
    RWTexture2D(HeapHandle h) : s(h.s) {}
    RWRef<T>& At(uint32_t x, uint32_t y) const
    {
        static thread_local RWRef<T> slot;
        FlushPendingRW();
        float f[4] = {0, 0, 0, 0}; uint32_t u[4] = {0, 0, 0, 0};
        const bool ok = x < s->w && y < s->h;
        const size_t idx = ok ? (size_t)y * s->w + x : 0;
        if (ok) LoadRaw(*s, idx, f, u);
        slot.Bind(s, idx, ok, Lanes<T>::get(f, u));
        g_pendingRW.flush = &RWRef<T>::FlushFn; g_pendingRW.obj = &slot;
        return slot;
    }
    RWRef<T>& operator[](const uint2& p) const { return At(p.x, p.y); }
    RWRef<T>& operator[](const int2& p) const { return At((uint32_t)p.x, (uint32_t)p.y); }
    RWRef<T>& operator[](const uint16_t2& p) const { return At(p.x, p.y); }
    template<int M, int A, int B> RWRef<T>& operator[](const Swz<uint32_t, M, A, B>& p) const { uint2 q = p; return At(q.x, q.y); }
    void GetDimensions(uint32_t& w, uint32_t& h) const { w = s->w; h = s->h; }
};

template<class T> struct Texture3D
{
    TexStorage* s = nullptr;
    Texture3D() {}
    Texture3D(HeapHandle h) : s(h.s) {}
    // fp32 trilinear, clamp addressing, texel centres at (i + 0.5) / N (the ABI's definition for the rho LUT)
    T SampleLevel(const SamplerState&, const float3& uvw, float) const
    {
        const float c[3] = {uvw.x, uvw.y, uvw.z};
        const uint32_t dim[3] = {s->w, s->h, s->d};
        int i0[3], i1[3]; float fr[3];
        for (int a = 0; a < 3; a++)
        {
            float x = c[a] * (float)dim[a] - 0.5f;
            float fl = zr_floor(x);
            fr[a] = x - fl;
            int i = (int)fl, hi = (int)dim[a] - 1;
            i0[a] = i < 0 ? 0 : (i > hi ? hi : i);
            i1[a] = (i + 1) < 0 ? 0 : ((i + 1) > hi ? hi : (i + 1));
        }
        float t[8][4]; uint32_t u[4];
        for (int k = 0; k < 8; k++)
        {
            t[k][0] = t[k][1] = t[k][2] = 0.0f; t[k][3] = 1.0f;
            LoadRaw(*s, ((size_t)((k & 4) ? i1[2] : i0[2]) * dim[1] + ((k & 2) ? i1[1] : i0[1])) * dim[0] + ((k & 1) ? i1[0] : i0[0]), t[k], u);
        }
        float o[4] = {0, 0, 0, 1}; uint32_t uu[4] = {0, 0, 0, 0};
        for (int ch = 0; ch < 3; ch++)
        {
            const float c00 = zr_lerp(t[0][ch], t[1][ch], fr[0]), c10 = zr_lerp(t[2][ch], t[3][ch], fr[0]);
            const float c01 = zr_lerp(t[4][ch], t[5][ch], fr[0]), c11 = zr_lerp(t[6][ch], t[7][ch], fr[0]);
            o[ch] = zr_lerp(zr_lerp(c00, c10, fr[1]), zr_lerp(c01, c11, fr[1]), fr[2]);
        }
        return Lanes<T>::get(o, uu);
    }
};

// RWByteAddressBuffer over host memory (atomics are plain read-modify-writes: the emulated group runs one lane at a time)
struct RWByteAddressBuffer
{
    uint8_t* p = nullptr;
    RWByteAddressBuffer() {}
    explicit RWByteAddressBuffer(void* ptr) : p((uint8_t*)ptr) {}
    uint Load(uint off) const { uint v; memcpy(&v, p + off, 4); return v; }
    void Store(uint off, uint v) const { memcpy(p + off, &v, 4); }
    void InterlockedAdd(uint off, uint v) const { Store(off, Load(off) + v); }
};

template<class T> struct StructuredBuffer
{
    const T* p = nullptr; uint32_t n = 0;
    StructuredBuffer() {}
    StructuredBuffer(const T* ptr, uint32_t count) : p(ptr), n(count) {}
    const T& operator[](uint32_t i) const { static const T zero{}; return i < n ? p[i] : zero; }
};
template<class T> struct RWStructuredBuffer
{
    T* p = nullptr; uint32_t n = 0;
    RWStructuredBuffer() {}
    RWStructuredBuffer(T* ptr, uint32_t count) : p(ptr), n(count) {}
    T& operator[](uint32_t i) const { static thread_local T sink; return i < n ? p[i] : sink; }
};

} // namespace hlsl
