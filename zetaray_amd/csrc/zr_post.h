// zr_post.h -- stage functions of the auto-exposure and display (tone mapping) passes (post stack).
//
// Reference: Source/ZetaRenderPass/AutoExposure/AutoExposure_Histogram.hlsl:27-73 (log-luminance histogram, 256 bins, bin 0 = "too dark"),
// AutoExposure_WeightedAvg.hlsl:20-109 (bin-weighted mean, inverse mapping, exponential adaptation, ISO-100 saturation-based exposure),
// AutoExposure.cpp:100-166 (histogram cleared every frame; exposure texture R32G32_FLOAT 1x1, zero-initialised),
// Display/Display.hlsl:41-77 (exposure, tone mapper switch), Display/Tonemap.hlsli:10-138 (Tony McMapface LUT, AgX default / golden /
// punchy / custom), Display.cpp:69-74 (defaults: NEUTRAL, auto-exposure on, saturation 1, AgX exponent 1).
// Pinned by this ABI: the input image is what an R16G16B16A16_FLOAT texture holds (an RGBA32F input is rounded to half on read);
// the WaveActiveSum of the 256-thread average is the canonical 64-lane xor butterfly (4 waves, then one more butterfly over the 4
// partial sums), as in zr_rpt.h; pow / exp / log2 are the ABI's (zr_detmath.h); mul(v, M) sums v[r] * M[r][c] left to right;
// the LUT is sampled with fp32 trilinear filtering, clamp addressing, texel centres at (i + 0.5) / N (like the rho LUT);
// the display output is the pixel shader's float4 return value (the reference's R8G8B8A8_UNORM_SRGB back buffer then applies the
// sRGB OETF and rounds to 8 bits in the ROP -- ZR_OUT_DISPLAY_SRGB8 does that with the IEC 61966-2-1 formula through zr_pow).
#pragma once
#include "zr_dev_math.h"

namespace zr {
namespace post {

static constexpr uint32_t kHistBins = 256;      // AutoExposure_Common.h:9

struct AeParams { float minLum, lumRange, lumMapExp, adaptationRate; };

ZR_HD V3 HalfRounded(F4 c) { return v3(zr_round_f16(c.x), zr_round_f16(c.y), zr_round_f16(c.z)); }
ZR_HD V3 LoadHalf3(const uint16_t* rgba16f, size_t px)
{ const uint16_t* p = rgba16f + 4 * px; return v3(zr_f16_to_f32(p[0]), zr_f16_to_f32(p[1]), zr_f16_to_f32(p[2])); }

// CalculateeBin, AutoExposure_Histogram.hlsl:27-45 (in-screen pixels)
ZR_HD uint32_t AeBin(V3 color, const AeParams& p)
{
    const float lum = Luminance(color);
    if (lum <= 1e-4f) return 0;
    float t = zr_saturate((lum - p.minLum) / p.lumRange);
    t = zr_pow(t, p.lumMapExp);
    return (uint32_t)(t * (float)(kHistBins - 2)) + 1u;
}

// a bin's contribution to the weighted mean, AutoExposure_WeightedAvg.hlsl:77-78
ZR_HD float AeBinValue(uint32_t gidx, uint32_t binSize)
{ return gidx == 0 ? 0.0f : (float)binSize * ((float)(gidx - 1u) + 0.5f) / (float)kHistBins; }

// ComputeAutoExposure, AutoExposure_WeightedAvg.hlsl:20-28
ZR_HD float ComputeAutoExposure(float avgLum)
{
    const float S = 100.0f, K = 12.5f;
    const float EV100 = zr_log2((avgLum * S) / K);
    const float q = 0.65f;
    const float luminanceMax = (78.0f / (q * S)) * zr_pow(2.0f, EV100);
    return 1 / luminanceMax;
}

// thread 0 of AutoExposure_WeightedAvg.hlsl:88-108: `sum` = the group's summed bin values; exposure[0] = exposure, [1] = adapted luminance
ZR_HD void AeResolve(float sum, uint32_t numSamples, float dt, const AeParams& p, float* exposure)
{
    float mean = sum / (float)(numSamples > 1u ? numSamples : 1u);
    float result = zr_pow(mean, 1.0f / p.lumMapExp);
    result = result * p.lumRange + p.minLum;
    const float prev = exposure[1];
    if (prev < 1e8f) result = prev + (result - prev) * (1 - zr_exp(-dt * 1000.0f * p.adaptationRate));
    exposure[0] = ComputeAutoExposure(result);
    exposure[1] = result;
}

// ---- Tonemap.hlsli
enum Tonemapper { TM_NONE = 0, TM_NEUTRAL, TM_AGX_DEFAULT, TM_AGX_GOLDEN, TM_AGX_PUNCHY, TM_AGX_CUSTOM, TM_COUNT };      // Display_Common.h:21-30

// R9G9B9E5_SHAREDEXP texel -> RGB (exact in fp32)
ZR_HD V3 DecodeRGB9E5(uint32_t v)
{
    const float scale = zr_asfloat((uint32_t)((int)(v >> 27) - 15 - 9 + 127) << 23);
    return v3((float)(v & 0x1ffu) * scale, (float)((v >> 9) & 0x1ffu) * scale, (float)((v >> 18) & 0x1ffu) * scale);
}
struct Lut3D { const uint32_t* data; uint32_t dim; };      // dim^3 RGB9E5 texels (tony_mc_mapface.dds: 48^3)

ZR_HD V3 SampleLut(const Lut3D& lut, V3 uvw)
{
    const float c[3] = {uvw.x, uvw.y, uvw.z};
    int i0[3], i1[3]; float fr[3];
    const int hi = (int)lut.dim - 1;
    for (int a = 0; a < 3; a++)
    {
        const float x = c[a] * (float)lut.dim - 0.5f, fl = zr_floor(x);
        fr[a] = x - fl;
        const int i = (int)fl;
        i0[a] = i < 0 ? 0 : (i > hi ? hi : i);
        i1[a] = (i + 1) < 0 ? 0 : ((i + 1) > hi ? hi : (i + 1));
    }
#define ZR_LUT(X, Y, Z) DecodeRGB9E5(lut.data[((size_t)(Z) * lut.dim + (size_t)(Y)) * lut.dim + (size_t)(X)])
    const V3 c000 = ZR_LUT(i0[0], i0[1], i0[2]), c100 = ZR_LUT(i1[0], i0[1], i0[2]), c010 = ZR_LUT(i0[0], i1[1], i0[2]), c110 = ZR_LUT(i1[0], i1[1], i0[2]);
    const V3 c001 = ZR_LUT(i0[0], i0[1], i1[2]), c101 = ZR_LUT(i1[0], i0[1], i1[2]), c011 = ZR_LUT(i0[0], i1[1], i1[2]), c111 = ZR_LUT(i1[0], i1[1], i1[2]);
#undef ZR_LUT
    V3 o;
    {
        const float c00 = zr_lerp(c000.x, c100.x, fr[0]), c10 = zr_lerp(c010.x, c110.x, fr[0]), c01 = zr_lerp(c001.x, c101.x, fr[0]), c11 = zr_lerp(c011.x, c111.x, fr[0]);
        o.x = zr_lerp(zr_lerp(c00, c10, fr[1]), zr_lerp(c01, c11, fr[1]), fr[2]);
    }
    {
        const float c00 = zr_lerp(c000.y, c100.y, fr[0]), c10 = zr_lerp(c010.y, c110.y, fr[0]), c01 = zr_lerp(c001.y, c101.y, fr[0]), c11 = zr_lerp(c011.y, c111.y, fr[0]);
        o.y = zr_lerp(zr_lerp(c00, c10, fr[1]), zr_lerp(c01, c11, fr[1]), fr[2]);
    }
    {
        const float c00 = zr_lerp(c000.z, c100.z, fr[0]), c10 = zr_lerp(c010.z, c110.z, fr[0]), c01 = zr_lerp(c001.z, c101.z, fr[0]), c11 = zr_lerp(c011.z, c111.z, fr[0]);
        o.z = zr_lerp(zr_lerp(c00, c10, fr[1]), zr_lerp(c01, c11, fr[1]), fr[2]);
    }
    return o;
}

// tony_mc_mapface, Tonemap.hlsli:10-23
ZR_HD V3 TonyMcMapface(V3 stimulus, const Lut3D& lut)
{
    const V3 encoded = v3(stimulus.x / (stimulus.x + 1.0f), stimulus.y / (stimulus.y + 1.0f), stimulus.z / (stimulus.z + 1.0f));
    const float LUT_DIMS = 48.0f;
    const V3 uv = encoded * ((LUT_DIMS - 1.0f) / LUT_DIMS) + v3(0.5f / LUT_DIMS);
    return SampleLut(lut, uv);
}

ZR_HD V3 Pow3(V3 v, float e) { return v3(zr_pow(v.x, e), zr_pow(v.y, e), zr_pow(v.z, e)); }
ZR_HD V3 MulRowVec(V3 v, const float* M)      // mul(v, float3x3 M), M row-major
{
    return v3(v.x * M[0] + v.y * M[3] + v.z * M[6], v.x * M[1] + v.y * M[4] + v.z * M[7], v.x * M[2] + v.y * M[5] + v.z * M[8]);
}
// agxDefaultContrastApprox, Tonemap.hlsli:30-44
ZR_HD V3 AgxContrast(V3 x)
{
    const V3 x2 = x * x, x4 = x2 * x2, x6 = x4 * x2;
    return -17.86f * x6 * x + 78.01f * x6 - 126.7f * x4 * x + 92.06f * x4 - 28.72f * x2 * x + 4.361f * x2 - 0.1718f * x + v3(0.002857f);
}
// agxInset, Tonemap.hlsli:46-67
ZR_HD V3 AgxInset(V3 val)
{
    const float agx_mat[9] = {0.842479062253094f, 0.0423282422610123f, 0.0423756549057051f,
                              0.0784335999999992f, 0.878468636469772f, 0.0784336f,
                              0.0792237451477643f, 0.0791661274605434f, 0.879142973793104f};
    const float min_ev = -12.47393f, max_ev = 4.026069f;
    val = MulRowVec(val, agx_mat);
    val = v3(zr_clamp(zr_log2(val.x), min_ev, max_ev), zr_clamp(zr_log2(val.y), min_ev, max_ev), zr_clamp(zr_log2(val.z), min_ev, max_ev));
    val = (val - v3(min_ev)) / (max_ev - min_ev);
    return AgxContrast(val);
}
// agxEotf, Tonemap.hlsli:69-85
ZR_HD V3 AgxEotf(V3 val)
{
    const float agx_mat_inv[9] = {1.19687900512017f, -0.0528968517574562f, -0.0529716355144438f,
                                  -0.0980208811401368f, 1.15190312990417f, -0.0980434501171241f,
                                  -0.0990297440797205f, -0.0989611768448433f, 1.15107367264116f};
    val = MulRowVec(val, agx_mat_inv);
    return Pow3(val, 2.2f);
}
// agxLook, Tonemap.hlsli:87-95
ZR_HD V3 AgxLook(V3 val, float offset, V3 slope, float exp, float saturation)
{
    const float luma = dot(val, v3(0.2126f, 0.7152f, 0.0722f));
    val = Pow3(val * slope + v3(offset), exp);
    return v3(luma) + saturation * (val - v3(luma));
}

struct DisplayParams { uint32_t tonemapper, autoExposure; float saturation, agxExp; };

// mainPS of Display.hlsl:41-77 with DisplayOption::DEFAULT: `composited` = the sampled input, returns .rgb (alpha is 1)
ZR_HD V3 DisplayPixel(V3 composited, float exposure, const DisplayParams& p, const Lut3D& lut)
{
    V3 display = composited;
    if (p.autoExposure) display = composited * exposure;
    if (p.tonemapper == TM_NEUTRAL)
    {
        display = TonyMcMapface(display, lut);
        const V3 desaturation = v3(Luminance(display));
        display = mad(p.saturation, display, mad(-p.saturation, desaturation, desaturation));      // Math::Lerp, Math.hlsli:66-70
    }
    else if (p.tonemapper == TM_AGX_DEFAULT) display = AgxEotf(AgxInset(display));
    else if (p.tonemapper == TM_AGX_GOLDEN) display = AgxEotf(AgxLook(AgxInset(display), 0.0f, v3(1.0f, 0.9f, 0.5f), 0.8f, 0.8f));
    else if (p.tonemapper == TM_AGX_PUNCHY) display = AgxEotf(AgxLook(AgxInset(display), 0.0f, v3(1.0f), 1.35f, 1.4f));
    else if (p.tonemapper == TM_AGX_CUSTOM) display = AgxEotf(AgxLook(AgxInset(display), 0.0f, v3(1.0f), p.agxExp, p.saturation));
    return display;
}

// what the R8G8B8A8_UNORM_SRGB render target stores for a linear value: IEC 61966-2-1 OETF, then UNORM8 (the ABI's (uint)fma(x, 255, 0.5))
ZR_HD uint32_t LinearToSrgb8(float c)
{
    c = zr_isnan(c) ? 0.0f : zr_saturate(c);
    const float e = c <= 0.0031308f ? 12.92f * c : 1.055f * zr_pow(c, 1.0f / 2.4f) - 0.055f;
    return (uint32_t)zr_fma(zr_saturate(e), 255.0f, 0.5f);
}

} // namespace post
} // namespace zr
