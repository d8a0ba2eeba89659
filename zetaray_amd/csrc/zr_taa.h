// zr_taa.h -- per-pixel stage function of the TAA pass (post stack).
//
// Reference: Source/ZetaRenderPass/TAA/TAA.hlsl:28-188 (Mitchell-weighted 3x3 reconstruction, closest-depth motion vector,
// Catmull-Rom history fetch, variance clipping, inverse-luminance blend), Common/Common.hlsli:63-105 (SampleTextureCatmullRom),
// TAA.cpp:120-146 (two R16G16B16A16_FLOAT outputs, ping-pong), TAA.h:72 (BlendWeight 0.1).
// Pinned by this ABI: the input signal is the RGBA32F composited image (the reference's is R16G16B16A16_FLOAT: every channel is
// rounded to half on read, which is what that format holds); g_samLinearClamp on the history = software bilinear with texel
// centres at (i + 0.5) / N, clamp addressing, fp32 weights (include/zr_texture.h conventions); rcp(x) = 1 / x.
#pragma once
#include "zr_rpt.h"      // DecodeMotion

namespace zr {
namespace taa {

struct TaaFrame
{
    const F4* signal;             // composited image of this frame, RGBA32F
    const float* depth;           // G-buffer planes of this frame
    const uint32_t* motion;
    const uint16_t* prevOut;      // RGBA16F history (previous frame's output)
    uint16_t* currOut;            // RGBA16F output; only .rgb is written
    uint32_t w, h;
    float blendWeight;
    uint32_t temporalIsValid;
};

ZR_HD float Mitchell1D(float x, float B, float C)      // TAA.hlsl:28-46
{
    x = zr_abs(2.0f * x);
    const float oneDivSix = 1.0f / 6.0f;
    if (x > 1)
        return ((-B - 6.0f * C) * x * x * x + (6.0f * B + 30.0f * C) * x * x + (-12.0f * B - 48.0f * C) * x + (8.0f * B + 24.0f * C)) * oneDivSix;
    return ((12.0f - 9.0f * B - 6.0f * C) * x * x * x + (-18.0f + 12.0f * B + 6.0f * C) * x * x + (6.0f - 2.0f * B)) * oneDivSix;
}

ZR_HD V3 ClipAABB(V3 aabbMin, V3 aabbMax, V3 histSample)      // TAA.hlsl:49-64
{
    const V3 center = 0.5f * (aabbMax + aabbMin);
    const V3 extents = 0.5f * (aabbMax - aabbMin);
    const V3 rayToCenter = histSample - center;
    V3 u = v3(rayToCenter.x / extents.x, rayToCenter.y / extents.y, rayToCenter.z / extents.z);
    u = v3(zr_abs(u.x), zr_abs(u.y), zr_abs(u.z));
    const float m = zr_max(u.x, zr_max(u.y, u.z));
    if (m > 1.0f) return center + rayToCenter / m;
    return histSample;
}

ZR_HD V3 LoadSignal(const TaaFrame& F, int x, int y)
{
    // the signal is Compositing's output, R32G32B32A32_FLOAT (Compositing.h:96; bound as TAA's input at PostProcessor.cpp:158): read as stored
    const F4 c = F.signal[(size_t)y * F.w + x];
    return v3(c.x, c.y, c.z);
}
// One 8-byte load per RGBA16F texel (the Catmull-Rom fetch reads 36 of them per pixel: three 2-byte loads each made TAA load-issue bound).
// zr_f16_to_f32 special-cases Inf / NaN per value -- ~8 extra instructions each, 108 conversions per pixel: most of this kernel's VALU work
// (profiles/r04b_post_sqA.csv: 1449 VALU instructions per pixel, the kernel VALU-bound at 0.08 of the HBM roof).  EXACT = false converts with plain
// v_cvt_f32_f16 and ORs a "some value of this texel is Inf / NaN" bit into `special`: adding 0x0400 to an exponent field of all ones carries into the
// field's top bit.  The caller re-runs the fetch with EXACT = true in the (practically never taken) case that the bit is set, so the result is
// zr_f16_to_f32's for every input.
template<bool EXACT>
ZR_HD V3 LoadHistoryTexel(const TaaFrame& F, int x, int y, uint64_t& special)
{
    uint64_t t;
    __builtin_memcpy(&t, F.prevOut + 4 * ((size_t)y * F.w + x), 8);
#if defined(__HIP_DEVICE_COMPILE__)
    if (!EXACT)
    {
        special |= (t & 0x00007c007c007c00ull) + 0x0000040004000400ull;
        union { uint16_t u; _Float16 h; } a, b, c;
        a.u = (uint16_t)(t & 0xffffu); b.u = (uint16_t)((t >> 16) & 0xffffu); c.u = (uint16_t)((t >> 32) & 0xffffu);
        return v3((float)a.h, (float)b.h, (float)c.h);
    }
#endif
    return v3(zr_f16_to_f32((uint16_t)(t & 0xffffu)), zr_f16_to_f32((uint16_t)((t >> 16) & 0xffffu)), zr_f16_to_f32((uint16_t)((t >> 32) & 0xffffu)));
}
// SampleLevel(g_samLinearClamp, uv, 0) on the history
template<bool EXACT>
ZR_HD V3 SampleHistory(const TaaFrame& F, float u, float v, uint64_t& special)
{
    const float x = u * (float)F.w - 0.5f, y = v * (float)F.h - 0.5f;
    const float fx = zr_floor(x), fy = zr_floor(y);
    const float tx = x - fx, ty = y - fy;
    int x0 = zr_f2i_sat(fx), y0 = zr_f2i_sat(fy);
    int x1 = x0 < 2147483647 ? x0 + 1 : x0, y1 = y0 < 2147483647 ? y0 + 1 : y0;
    const int W1 = (int)F.w - 1, H1 = (int)F.h - 1;
    x0 = x0 < 0 ? 0 : (x0 > W1 ? W1 : x0); x1 = x1 < 0 ? 0 : (x1 > W1 ? W1 : x1);
    y0 = y0 < 0 ? 0 : (y0 > H1 ? H1 : y0); y1 = y1 < 0 ? 0 : (y1 > H1 ? H1 : y1);
    const V3 c00 = LoadHistoryTexel<EXACT>(F, x0, y0, special), c10 = LoadHistoryTexel<EXACT>(F, x1, y0, special);
    const V3 c01 = LoadHistoryTexel<EXACT>(F, x0, y1, special), c11 = LoadHistoryTexel<EXACT>(F, x1, y1, special);
    const V3 top = c00 + tx * (c10 - c00);
    const V3 bot = c01 + tx * (c11 - c01);
    return top + ty * (bot - top);
}
// Common::SampleTextureCatmullRom, Common.hlsli:63-105 (9 bilinear fetches)
template<bool EXACT>
ZR_HD V3 SampleHistoryCatmullRomT(const TaaFrame& F, V2 uv, V2 texSize, uint64_t& special)
{
    const V2 samplePos = v2(uv.x * texSize.x, uv.y * texSize.y);
    const V2 texPos1 = v2(zr_floor(samplePos.x - 0.5f) + 0.5f, zr_floor(samplePos.y - 0.5f) + 0.5f);
    const V2 f = samplePos - texPos1;
    const V2 w0 = v2(f.x * (-0.5f + f.x * (1.0f - 0.5f * f.x)), f.y * (-0.5f + f.y * (1.0f - 0.5f * f.y)));
    const V2 w1 = v2(1.0f + f.x * f.x * (-2.5f + 1.5f * f.x), 1.0f + f.y * f.y * (-2.5f + 1.5f * f.y));
    const V2 w2 = v2(f.x * (0.5f + f.x * (2.0f - 1.5f * f.x)), f.y * (0.5f + f.y * (2.0f - 1.5f * f.y)));
    const V2 w3 = v2(f.x * f.x * (-0.5f + 0.5f * f.x), f.y * f.y * (-0.5f + 0.5f * f.y));
    const V2 w12 = w1 + w2;
    const V2 offset12 = v2(w2.x / (w1.x + w2.x), w2.y / (w1.y + w2.y));
    V2 texPos0 = texPos1 - v2(1.0f, 1.0f);
    V2 texPos3 = texPos1 + v2(2.0f, 2.0f);
    V2 texPos12 = texPos1 + offset12;
    texPos0 = v2(texPos0.x / texSize.x, texPos0.y / texSize.y);
    texPos3 = v2(texPos3.x / texSize.x, texPos3.y / texSize.y);
    texPos12 = v2(texPos12.x / texSize.x, texPos12.y / texSize.y);
    V3 result = v3(0.0f);
    result = result + SampleHistory<EXACT>(F, texPos0.x, texPos0.y, special) * w0.x * w0.y;
    result = result + SampleHistory<EXACT>(F, texPos12.x, texPos0.y, special) * w12.x * w0.y;
    result = result + SampleHistory<EXACT>(F, texPos3.x, texPos0.y, special) * w3.x * w0.y;
    result = result + SampleHistory<EXACT>(F, texPos0.x, texPos12.y, special) * w0.x * w12.y;
    result = result + SampleHistory<EXACT>(F, texPos12.x, texPos12.y, special) * w12.x * w12.y;
    result = result + SampleHistory<EXACT>(F, texPos3.x, texPos12.y, special) * w3.x * w12.y;
    result = result + SampleHistory<EXACT>(F, texPos0.x, texPos3.y, special) * w0.x * w3.y;
    result = result + SampleHistory<EXACT>(F, texPos12.x, texPos3.y, special) * w12.x * w3.y;
    result = result + SampleHistory<EXACT>(F, texPos3.x, texPos3.y, special) * w3.x * w3.y;
    return result;
}
ZR_HD V3 SampleHistoryCatmullRom(const TaaFrame& F, V2 uv, V2 texSize)
{
    uint64_t special = 0;
    V3 r = SampleHistoryCatmullRomT<false>(F, uv, texSize, special);
    if (special & 0x0000800080008000ull) r = SampleHistoryCatmullRomT<true>(F, uv, texSize, special);      // an Inf / NaN among the 36 texels
    return r;
}

ZR_HD void StoreRGB16F(const TaaFrame& F, uint32_t x, uint32_t y, V3 c)
{
    uint16_t* p = F.currOut + 4 * ((size_t)y * F.w + x);
    p[0] = zr_f32_to_f16(c.x); p[1] = zr_f32_to_f16(c.y); p[2] = zr_f32_to_f16(c.z);
}

// TAA.hlsl main (:70-188) for pixel (x, y)
ZR_HD void TaaPixel(const TaaFrame& F, uint32_t x, uint32_t y)
{
    const float depth = F.depth[(size_t)y * F.w + x];
    const V3 currColor = LoadSignal(F, (int)x, (int)y);
    if (!F.temporalIsValid || depth == ZR_FLT_MAX) { StoreRGB16F(F, x, y, currColor); return; }

    float weightSum = Mitchell1D(0, 0.33f, 0.33f) * Mitchell1D(0, 0.33f, 0.33f);
    V3 reconstructed = currColor * weightSum;
    V3 firstMoment = currColor;
    V3 secondMoment = currColor * currColor;
    float closestDepth = depth;
    int cdx = 0, cdy = 0;
    int numNeighbors = 1;
    for (int i = -1; i < 2; i++)
        for (int j = -1; j < 2; j++)
        {
            if (i == 0 && j == 0) continue;
            const int nx = (int)x + i, ny = (int)y + j;
            if (nx < 0 || ny < 0 || nx >= (int)F.w || ny >= (int)F.h) continue;
            const V3 neighborColor = vmax(LoadSignal(F, nx, ny), 0.0f);
            float weight = Mitchell1D((float)i, 0.33f, 0.33f) * Mitchell1D((float)j, 0.33f, 0.33f);
            weight *= 1.0f / (1.0f + Luminance(neighborColor));
            reconstructed = reconstructed + neighborColor * weight;
            weightSum += weight;
            firstMoment = firstMoment + neighborColor;
            secondMoment = secondMoment + neighborColor * neighborColor;
            const float neighborDepth = F.depth[(size_t)ny * F.w + nx];
            if (neighborDepth < closestDepth) { closestDepth = neighborDepth; cdx = i; cdy = j; }
            numNeighbors += 1;
        }
    reconstructed = reconstructed / zr_max(weightSum, 1e-5f);

    const V2 motionVec = rpt::DecodeMotion(F.motion[(size_t)((int)y + cdy) * F.w + ((int)x + cdx)]);
    const V2 renderDim = v2((float)F.w, (float)F.h);
    const V2 currUV = v2(((float)x + 0.5f) / renderDim.x, ((float)y + 0.5f) / renderDim.y);
    const V2 prevUV = currUV - motionVec;
    if (prevUV.x < 0.0f || prevUV.y < 0.0f || prevUV.x > 1.0f || prevUV.y > 1.0f) { StoreRGB16F(F, x, y, reconstructed); return; }

    const V3 history = SampleHistoryCatmullRom(F, prevUV, renderDim);
    const float n = (float)numNeighbors;
    const V3 mean = firstMoment / n;
    V3 sd = secondMoment - (firstMoment * firstMoment) / n;
    sd = v3(zr_abs(sd.x), zr_abs(sd.y), zr_abs(sd.z));
    sd = sd / (n - 1.0f);
    sd = v3(zr_sqrt(sd.x), zr_sqrt(sd.y), zr_sqrt(sd.z));
    const V3 clippedHistory = ClipAABB(mean - sd, mean + sd, history);
    const float currWeight = zr_saturate(F.blendWeight * (1.0f / (1.0f + Luminance(reconstructed))));
    const float histWeight = zr_saturate((1.0f - F.blendWeight) * (1.0f / (1.0f + Luminance(clippedHistory))));
    V3 result = (currWeight * reconstructed + histWeight * clippedHistory) / (currWeight + histWeight);
    result = any_nan(result) ? reconstructed : result;
    StoreRGB16F(F, x, y, result);
}

} // namespace taa
} // namespace zr
