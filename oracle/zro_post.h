// zro_post.h -- ORACLE (test infrastructure only): CPU restatement of the auto-exposure and display passes.
//
// Follows, statement by statement:
//   Source/ZetaRenderPass/AutoExposure/AutoExposure_Histogram.hlsl:27-73   (CalculateeBin, per-group histogram)
//   Source/ZetaRenderPass/AutoExposure/AutoExposure_WeightedAvg.hlsl:20-109 (ComputeAutoExposure, main)
//   Source/ZetaRenderPass/Display/Display.hlsl:41-77                        (mainPS, DisplayOption::DEFAULT)
//   Source/ZetaRenderPass/Display/Tonemap.hlsli:10-138                      (tony_mc_mapface, AgX)
// Pinned by the ABI (include/zetaray_amd.h): WaveGetLaneCount() = 64 and WaveActiveSum = the canonical xor butterfly (RPT::WaveSum64);
// the LUT fetch is fp32 trilinear / clamp / texel centres at (i + 0.5) / N; transcendental functions are zr_detmath.h's.
// Parity: pinned against the reference's own shaders compiled as C++ (oracle/_ref, tests/test_ref_passes.py post_* cases).
#pragma once
#include "zro_math.h"
#include "zro_rpt.h"      // RPT::WaveSum64

namespace zro {
namespace Post {

struct Image      // what Texture2D<half4> / Texture2D<float4> on an R16G16B16A16_FLOAT texture returns
{
    const void* data; bool f16; uint32_t w, h;
    float3 Load(uint32_t x, uint32_t y) const
    {
        const size_t px = (size_t)y * w + x;
        if (f16) { const uint16_t* p = (const uint16_t*)data + 4 * px; return f3(zr_f16_to_f32(p[0]), zr_f16_to_f32(p[1]), zr_f16_to_f32(p[2])); }
        const float* p = (const float*)data + 4 * px;
        return f3(zr_round_f16(p[0]), zr_round_f16(p[1]), zr_round_f16(p[2]));
    }
};

struct cbAutoExposureHist { float MinLum, LumRange, LumMapExp, AdaptationRate; };
static const uint32_t HIST_BIN_COUNT = 256;

// AutoExposure_Histogram.hlsl:27-45
static inline uint32_t CalculateeBin(const Image& g_input, uint32_t x, uint32_t y, const cbAutoExposureHist& g_local)
{
    const float3 color = g_input.Load(x, y);
    const float lum = Math::Luminance(color);
    if (lum <= 1e-4f) return 0;
    float t = zr_saturate((lum - g_local.MinLum) / g_local.LumRange);
    t = zr_pow(t, g_local.LumMapExp);
    uint32_t bin = (uint32_t)(t * (float)(HIST_BIN_COUNT - 2)) + 1;
    return bin;
}
static inline void Histogram(const Image& img, const cbAutoExposureHist& cb, uint32_t* hist)
{
    for (uint32_t i = 0; i < HIST_BIN_COUNT; i++) hist[i] = 0;
    for (uint32_t y = 0; y < img.h; y++) for (uint32_t x = 0; x < img.w; x++) hist[CalculateeBin(img, x, y, cb)]++;
}

// AutoExposure_WeightedAvg.hlsl:20-28
static inline float ComputeAutoExposure(float avgLum)
{
    const float S = 100.0f;
    const float K = 12.5f;
    const float EV100 = zr_log2((avgLum * S) / K);
    const float q = 0.65f;
    const float luminanceMax = (78.0f / (q * S)) * zr_pow(2.0f, EV100);
    return 1 / luminanceMax;
}
// AutoExposure_WeightedAvg.hlsl:44-109, one group of 256 threads = 4 waves of 64 lanes; g_out = (exposure, adapted luminance)
static inline void WeightedAvg(const uint32_t* g_hist, uint32_t renderW, uint32_t renderH, float dt, const cbAutoExposureHist& g_local, float* g_out)
{
    float val[HIST_BIN_COUNT];
    for (uint32_t Gidx = 0; Gidx < HIST_BIN_COUNT; Gidx++)
    {
        const bool isFirstBin = (Gidx == 0);
        const uint32_t binSize = isFirstBin ? 0 : g_hist[Gidx];
        const bool skip = isFirstBin;
        val[Gidx] = skip ? 0 : (float)binSize * ((float)(Gidx - 1) + 0.5f) / (float)HIST_BIN_COUNT;
    }
    const uint32_t numExcludedSamples = g_hist[0];
    const uint32_t numSamples = renderW * renderH - numExcludedSamples;
    float g_waveSum[64];
    for (int i = 0; i < 64; i++) g_waveSum[i] = 0.0f;
    for (int wave = 0; wave < 4; wave++) g_waveSum[wave] = RPT::WaveSum64(val + 64 * wave);
    float mean = RPT::WaveSum64(g_waveSum);      // lanes 4..63 of the first wave contribute 0
    mean /= (float)(numSamples > 1 ? numSamples : 1);
    float result = zr_pow(mean, 1.0f / g_local.LumMapExp);
    result = result * g_local.LumRange + g_local.MinLum;
    const float prev = g_out[1];
    if (prev < 1e8f) result = prev + (result - prev) * (1 - zr_exp(-dt * 1000.0f * g_local.AdaptationRate));
    const float exposure = ComputeAutoExposure(result);
    g_out[0] = exposure; g_out[1] = result;
}

// ---- Tonemap.hlsli
struct Lut { const uint32_t* data; uint32_t dim; };
static inline float3 LutTexel(const Lut& l, int x, int y, int z)      // R9G9B9E5_SHAREDEXP
{
    const uint32_t v = l.data[((size_t)z * l.dim + (size_t)y) * l.dim + (size_t)x];
    const int e = (int)(v >> 27) - 15 - 9;
    const float scale = zr_asfloat((uint32_t)(e + 127) << 23);
    return f3((float)(v & 0x1ff) * scale, (float)((v >> 9) & 0x1ff) * scale, (float)((v >> 18) & 0x1ff) * scale);
}
static inline float Tri(float a000, float a100, float a010, float a110, float a001, float a101, float a011, float a111, float tx, float ty, float tz)
{
    const float c00 = zr_lerp(a000, a100, tx), c10 = zr_lerp(a010, a110, tx), c01 = zr_lerp(a001, a101, tx), c11 = zr_lerp(a011, a111, tx);
    return zr_lerp(zr_lerp(c00, c10, ty), zr_lerp(c01, c11, ty), tz);
}
static inline float3 SampleLinearClamp(const Lut& l, float3 uvw)
{
    const float c[3] = {uvw.x, uvw.y, uvw.z};
    int lo[3], hi[3]; float t[3];
    for (int a = 0; a < 3; a++)
    {
        const float x = c[a] * (float)l.dim - 0.5f;
        const float fl = zr_floor(x);
        t[a] = x - fl;
        const int i = (int)fl, m = (int)l.dim - 1;
        lo[a] = i < 0 ? 0 : (i > m ? m : i);
        hi[a] = i + 1 < 0 ? 0 : (i + 1 > m ? m : i + 1);
    }
    const float3 a000 = LutTexel(l, lo[0], lo[1], lo[2]), a100 = LutTexel(l, hi[0], lo[1], lo[2]), a010 = LutTexel(l, lo[0], hi[1], lo[2]), a110 = LutTexel(l, hi[0], hi[1], lo[2]);
    const float3 a001 = LutTexel(l, lo[0], lo[1], hi[2]), a101 = LutTexel(l, hi[0], lo[1], hi[2]), a011 = LutTexel(l, lo[0], hi[1], hi[2]), a111 = LutTexel(l, hi[0], hi[1], hi[2]);
    return f3(Tri(a000.x, a100.x, a010.x, a110.x, a001.x, a101.x, a011.x, a111.x, t[0], t[1], t[2]),
              Tri(a000.y, a100.y, a010.y, a110.y, a001.y, a101.y, a011.y, a111.y, t[0], t[1], t[2]),
              Tri(a000.z, a100.z, a010.z, a110.z, a001.z, a101.z, a011.z, a111.z, t[0], t[1], t[2]));
}
// Tonemap.hlsli:10-23
static inline float3 tony_mc_mapface(float3 stimulus, const Lut& lut)
{
    const float3 encoded = stimulus / (stimulus + 1.0f);
    const float LUT_DIMS = 48.0f;
    const float3 uv = encoded * ((LUT_DIMS - 1.0f) / LUT_DIMS) + 0.5f / LUT_DIMS;
    return SampleLinearClamp(lut, uv);
}
static inline float3 pow3(float3 v, float e) { return f3(zr_pow(v.x, e), zr_pow(v.y, e), zr_pow(v.z, e)); }
static inline float3 mul_v_M(float3 v, const float M[3][3])      // mul(v, M): sum over rows r of v[r] * M[r][c], left to right
{
    float o[3];
    for (int c = 0; c < 3; c++) o[c] = v.x * M[0][c] + v.y * M[1][c] + v.z * M[2][c];
    return f3(o[0], o[1], o[2]);
}
// Tonemap.hlsli:30-44
static inline float3 agxDefaultContrastApprox(float3 x)
{
    float3 x2 = x * x;
    float3 x4 = x2 * x2;
    float3 x6 = x4 * x2;
    return -17.86f * x6 * x + 78.01f * x6 - 126.7f * x4 * x + 92.06f * x4 - 28.72f * x2 * x + 4.361f * x2 - 0.1718f * x + 0.002857f;
}
// Tonemap.hlsli:46-67
static inline float3 agxInset(float3 val)
{
    static const float agx_mat[3][3] = {{0.842479062253094f, 0.0423282422610123f, 0.0423756549057051f},
                                        {0.0784335999999992f, 0.878468636469772f, 0.0784336f},
                                        {0.0792237451477643f, 0.0791661274605434f, 0.879142973793104f}};
    const float min_ev = -12.47393f;
    const float max_ev = 4.026069f;
    val = mul_v_M(val, agx_mat);
    val = f3(zr_clamp(zr_log2(val.x), min_ev, max_ev), zr_clamp(zr_log2(val.y), min_ev, max_ev), zr_clamp(zr_log2(val.z), min_ev, max_ev));
    val = (val - min_ev) / (max_ev - min_ev);
    val = agxDefaultContrastApprox(val);
    return val;
}
// Tonemap.hlsli:69-85
static inline float3 agxEotf(float3 val)
{
    static const float agx_mat_inv[3][3] = {{1.19687900512017f, -0.0528968517574562f, -0.0529716355144438f},
                                            {-0.0980208811401368f, 1.15190312990417f, -0.0980434501171241f},
                                            {-0.0990297440797205f, -0.0989611768448433f, 1.15107367264116f}};
    val = mul_v_M(val, agx_mat_inv);
    val = pow3(val, 2.2f);
    return val;
}
// Tonemap.hlsli:87-95
static inline float3 agxLook(float3 val, float offset, float3 slope, float exp, float saturation)
{
    const float3 lw = f3(0.2126f, 0.7152f, 0.0722f);
    float luma = dot(val, lw);
    val = pow3(val * slope + offset, exp);
    return f3(luma) + saturation * (val - luma);
}

struct cbDisplayPass { uint32_t Tonemapper, AutoExposure; float Saturation, AgXExp; };

// Display.hlsl:41-77 (DisplayOption::DEFAULT): PosSS.xy = pixel centre; g_samPointClamp fetch of the composited texture
static inline float4 mainPS(uint32_t px, uint32_t py, uint32_t displayW, uint32_t displayH, const Image& g_composited, const float* g_exposure,
    const cbDisplayPass& g_local, const Lut& lut)
{
    const float2 uv = f2(((float)px + 0.5f) / (float)displayW, ((float)py + 0.5f) / (float)displayH);
    int sx = (int)zr_floor(uv.x * (float)g_composited.w), sy = (int)zr_floor(uv.y * (float)g_composited.h);
    sx = sx < 0 ? 0 : (sx >= (int)g_composited.w ? (int)g_composited.w - 1 : sx);
    sy = sy < 0 ? 0 : (sy >= (int)g_composited.h ? (int)g_composited.h - 1 : sy);
    float3 composited = g_composited.Load((uint32_t)sx, (uint32_t)sy);
    float3 display = composited;
    if (g_local.AutoExposure)
    {
        const float exposure = g_exposure[0];
        const float3 exposedColor = composited * exposure;
        display = exposedColor;
    }
    if (g_local.Tonemapper == 1)
    {
        display = tony_mc_mapface(display, lut);
        float3 desaturation = f3(Math::Luminance(display));
        display = Math::Lerp(desaturation, display, g_local.Saturation);
    }
    else if (g_local.Tonemapper == 2) display = agxEotf(agxInset(display));
    else if (g_local.Tonemapper == 3) display = agxEotf(agxLook(agxInset(display), 0.0f, f3(1.0f, 0.9f, 0.5f), 0.8f, 0.8f));
    else if (g_local.Tonemapper == 4) display = agxEotf(agxLook(agxInset(display), 0.0f, f3(1.0f), 1.35f, 1.4f));
    else if (g_local.Tonemapper == 5) display = agxEotf(agxLook(agxInset(display), 0.0f, f3(1.0f), g_local.AgXExp, g_local.Saturation));
    return f4(display.x, display.y, display.z, 1.0f);
}

// R8G8B8A8_UNORM_SRGB store as the ABI defines it (include/zetaray_amd.h ZR_OUT_DISPLAY_SRGB8)
static inline uint32_t LinearToSrgb8(float c)
{
    c = zr_isnan(c) ? 0.0f : zr_saturate(c);
    const float e = c <= 0.0031308f ? 12.92f * c : 1.055f * zr_pow(c, 1.0f / 2.4f) - 0.055f;
    return (uint32_t)zr_fma(zr_saturate(e), 255.0f, 0.5f);
}

} // namespace Post
} // namespace zro
