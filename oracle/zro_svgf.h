// zro_svgf.h -- ORACLE (test infrastructure only): CPU restatement of the denoise pass (ZR_PASS_DENOISE), whole-image loops.
//
// PARITY UNPINNED: the reference has no denoiser, so there is no reference code to follow or to pin against.  The pass is defined by the
// written specification at the top of zetaray_amd/csrc/zr_svgf.h (after Schied et al., "Spatiotemporal Variance-Guided Filtering", HPG 2017;
// DEFINITION VERSION 3: compact-support falloff E(x) = max(0, 1 - x / 16)^16, written fused multiply-adds, predicated taps, one reciprocal per
// pixel, sanitised + clamped signal, fp16 colour / normal in the planes between the stages); this file restates that specification independently of the HIP stage functions -- image-level passes over
// plain arrays, its own helper structure, full frames only (the product's tile windows are compared against THIS full frame) -- with the same fp32
// operations in the same order, so the two can be compared bit for bit (tests/test_denoise.py).
#pragma once
#include "zro_math.h"
#include "zro_rpt.h"      // RPT::DecodeMotion
#include <vector>
#include <cstring>

namespace zro {
namespace SVGF {

struct Params { float alpha, alphaMoments, sigmaL, sigmaZ; uint32_t normalPowerLog2, iterations; };

struct Guide { float z, fw; float3 n; };

static inline float3 NormalOf(uint32_t bits) { const uint16_t e[2] = {(uint16_t)(bits & 0xffffu), (uint16_t)(bits >> 16)}; return Math::DecodeOct32(e); }
static inline bool Miss(float z) { return z == ZR_FLT_MAX; }
static inline bool IsFinite(float v) { return !zr_isnan(v) && !zr_isinf(v); }
// Lum(c) = fma(0.2126, c.x, fma(0.7152, c.y, 0.0722 * c.z))
static inline float Lum(float3 c) { const float b = 0.0722f * c.z; const float gb = zr_fma(0.7152f, c.y, b); return zr_fma(0.2126f, c.x, gb); }
// E(x) = max(0, 1 - x / 16)^16
static inline float E16(float x)
{
    const float lin = zr_fma(x, -0.0625f, 1.0f);
    const float t1 = zr_max(0.0f, lin);
    const float t2 = t1 * t1, t4 = t2 * t2, t8 = t4 * t4;
    return t8 * t8;
}
static inline int ClampI(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
// h(v): through fp16 (round to nearest even) and back -- the storage format of colour and normal between two filter stages
static inline float AsHalf(float v) { return zr_f16_to_f32_portable(zr_f32_to_f16_portable(v)); }
static inline void ColourThroughHalf(float* rgba, size_t n) { for (size_t i = 0; i < n; i++) for (int k = 0; k < 3; k++) rgba[4 * i + k] = AsHalf(rgba[4 * i + k]); }

static inline std::vector<Guide> BuildGuide(const float* depth, const uint32_t* normal, int W, int H)
{
    std::vector<Guide> g((size_t)W * H);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++)
        {
            Guide& o = g[(size_t)y * W + x];
            o.z = depth[(size_t)y * W + x];
            o.n = NormalOf(normal[(size_t)y * W + x]);
            o.n = f3(AsHalf(o.n.x), AsHalf(o.n.y), AsHalf(o.n.z));      // the guide normal is what an fp16 plane holds
            o.fw = 0.0f;
            if (Miss(o.z)) continue;
            float ddx = 0.0f, ddy = 0.0f;
            const int xn = (x + 1 < W) ? x + 1 : x - 1;
            const int yn = (y + 1 < H) ? y + 1 : y - 1;
            if (xn >= 0 && !Miss(depth[(size_t)y * W + xn])) ddx = zr_abs(depth[(size_t)y * W + xn] - o.z);
            if (yn >= 0 && !Miss(depth[(size_t)yn * W + x])) ddy = zr_abs(depth[(size_t)yn * W + x] - o.z);
            o.fw = zr_max(ddx, ddy);
        }
    return g;
}

// Nw(n, nq) = max(0, fma(n.x, nq.x, fma(n.y, nq.y, n.z * nq.z)))^(2^log2)
static inline float PowNormal(float3 a, float3 b, uint32_t log2)
{
    const float zz = a.z * b.z;
    const float yz = zr_fma(a.y, b.y, zz);
    float d = zr_max(0.0f, zr_fma(a.x, b.x, yz));
    for (uint32_t k = 0; k < log2; k++) d = d * d;
    return d;
}

struct History { const float* color; const float* moments; const float* prevDepth; const uint32_t* prevNormal; };

static inline bool Usable(const History& h, int W, int H, int qx, int qy, float z, float3 n)
{
    if (qx < 0 || qy < 0 || qx >= W || qy >= H) return false;
    const size_t j = (size_t)qy * W + qx;
    if (Miss(h.prevDepth[j])) return false;
    if (!(zr_abs(h.prevDepth[j] - z) <= 0.1f * z)) return false;
    return dot(NormalOf(h.prevNormal[j]), n) >= 0.9f;
}

// signal RGBA32F; accum RGBA32F (rgb + history length); moments 2 floats per pixel
static inline void Temporal(const float* signal, const float* depth, const uint32_t* normal, const uint32_t* motion, const History& h, bool temporalValid,
    const Params& prm, int W, int H, float* accum, float* moments)
{
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++)
        {
            const size_t i = (size_t)y * W + x;
            float3 c = f3(signal[4 * i], signal[4 * i + 1], signal[4 * i + 2]);
            if (!IsFinite(c.x) || !IsFinite(c.y) || !IsFinite(c.z)) c = f3(0.0f, 0.0f, 0.0f);
            else c = f3(zr_min(zr_max(c.x, -60000.0f), 60000.0f), zr_min(zr_max(c.y, -60000.0f), 60000.0f), zr_min(zr_max(c.z, -60000.0f), 60000.0f));
            const float lum = Lum(c);
            float3 out = c; float m1 = lum, m2 = lum * lum, length = 1.0f;
            const float z = depth[i];
            if (!Miss(z) && temporalValid)
            {
                const float3 n = NormalOf(normal[i]);
                const float2 mv = RPT::DecodeMotion(motion[i]);
                const float pu = ((float)x + 0.5f) / (float)W - mv.x, pv = ((float)y + 0.5f) / (float)H - mv.y;
                const float qx = pu * (float)W - 0.5f, qy = pv * (float)H - 0.5f;
                const float bx = zr_floor(qx), by = zr_floor(qy);
                const float tx = qx - bx, ty = qy - by;
                const bool outside = !(bx > -4.0f && by > -4.0f && bx < (float)W + 4.0f && by < (float)H + 4.0f);
                float3 hc = f3(0.0f, 0.0f, 0.0f); float h1 = 0, h2 = 0, hl = 0, ws = 0;
                if (!outside)
                {
                    const int ix = (int)bx, iy = (int)by;
                    const float wgt[4] = {(1.0f - tx) * (1.0f - ty), tx * (1.0f - ty), (1.0f - tx) * ty, tx * ty};
                    for (int k = 0; k < 4; k++)
                    {
                        const int sx = ix + (k & 1), sy = iy + (k >> 1);
                        if (!Usable(h, W, H, sx, sy, z, n)) continue;
                        const size_t j = (size_t)sy * W + sx;
                        hc = hc + wgt[k] * f3(h.color[4 * j], h.color[4 * j + 1], h.color[4 * j + 2]);
                        hl += wgt[k] * h.color[4 * j + 3];
                        h1 += wgt[k] * h.moments[2 * j]; h2 += wgt[k] * h.moments[2 * j + 1];
                        ws += wgt[k];
                    }
                    if (!(ws > 0.01f))
                    {
                        hc = f3(0.0f, 0.0f, 0.0f); h1 = 0; h2 = 0; hl = 0; ws = 0;
                        const int rx = (int)zr_floor(qx + 0.5f), ry = (int)zr_floor(qy + 0.5f);
                        for (int oy = -1; oy <= 1; oy++)
                            for (int ox = -1; ox <= 1; ox++)
                            {
                                if (!Usable(h, W, H, rx + ox, ry + oy, z, n)) continue;
                                const size_t j = (size_t)(ry + oy) * W + (rx + ox);
                                hc = hc + f3(h.color[4 * j], h.color[4 * j + 1], h.color[4 * j + 2]);
                                hl += h.color[4 * j + 3];
                                h1 += h.moments[2 * j]; h2 += h.moments[2 * j + 1];
                                ws += 1.0f;
                            }
                    }
                }
                const bool sumsFinite = IsFinite(hc.x) && IsFinite(hc.y) && IsFinite(hc.z) && IsFinite(hl) && IsFinite(h1) && IsFinite(h2);
                if (ws > 0.01f && sumsFinite)
                {
                    hc = hc / ws; h1 = h1 / ws; h2 = h2 / ws; hl = hl / ws;
                    length = zr_min(hl + 1.0f, 255.0f);
                    const float ac = zr_max(prm.alpha, 1.0f / length), am = zr_max(prm.alphaMoments, 1.0f / length);
                    out = hc + ac * (c - hc);
                    m1 = h1 + am * (lum - h1);
                    m2 = h2 + am * (lum * lum - h2);
                }
            }
            accum[4 * i] = out.x; accum[4 * i + 1] = out.y; accum[4 * i + 2] = out.z; accum[4 * i + 3] = length;
            moments[2 * i] = m1; moments[2 * i + 1] = m2;
        }
}

// accum (rgb + length), moments -> dst (rgb + variance)
static inline void Variance(const float* accum, const float* moments, const std::vector<Guide>& g, const Params& prm, int W, int H, float* dst)
{
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++)
        {
            const size_t i = (size_t)y * W + x;
            const Guide& c0 = g[i];
            const float length = accum[4 * i + 3];
            float col[3] = {accum[4 * i], accum[4 * i + 1], accum[4 * i + 2]};
            float m1 = moments[2 * i], m2 = moments[2 * i + 1];
            float var;
            if (Miss(c0.z)) var = 0.0f;
            else if (length >= 4.0f) var = zr_max(0.0f, m2 - m1 * m1);
            else
            {
                const float rcpPhiZ = 1.0f / (prm.sigmaZ * zr_max(c0.fw, 1e-8f));
                float ws = 1.0f;
                for (int dy = -3; dy <= 3; dy++)
                    for (int dx = -3; dx <= 3; dx++)
                    {
                        if (dx == 0 && dy == 0) continue;
                        const int qx = x + dx, qy = y + dy;
                        const bool inside = qx >= 0 && qy >= 0 && qx < W && qy < H;
                        const size_t j = (size_t)ClampI(qy, 0, H - 1) * W + ClampI(qx, 0, W - 1);      // every tap is read (position clamped) ...
                        const float rcpDist = 1.0f / zr_sqrt((float)(dx * dx + dy * dy));
                        const float wz = zr_abs(c0.z - g[j].z) * (rcpPhiZ * rcpDist);
                        float w = E16(wz) * PowNormal(c0.n, g[j].n, prm.normalPowerLog2);
                        if (!inside || Miss(g[j].z)) w = 0.0f;                                           // ... and weighted 0 when it does not count
                        for (int k = 0; k < 3; k++) col[k] = zr_fma(w, accum[4 * j + k], col[k]);
                        m1 = zr_fma(w, moments[2 * j], m1); m2 = zr_fma(w, moments[2 * j + 1], m2);
                        ws = ws + w;
                    }
                const float r = 1.0f / ws;
                for (int k = 0; k < 3; k++) col[k] = col[k] * r;
                m1 = m1 * r; m2 = m2 * r;
                var = zr_max(0.0f, m2 - m1 * m1) * (4.0f / length);
            }
            dst[4 * i] = col[0]; dst[4 * i + 1] = col[1]; dst[4 * i + 2] = col[2]; dst[4 * i + 3] = var;
        }
}

static inline void Atrous(const float* src, const std::vector<Guide>& g, const Params& prm, int W, int H, int step, float* dst)
{
    static const float kB3[3] = {1.0f, 2.0f / 3.0f, 1.0f / 6.0f};
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++)
        {
            const size_t i = (size_t)y * W + x;
            const Guide& c0 = g[i];
            float col[3] = {src[4 * i], src[4 * i + 1], src[4 * i + 2]};
            float var = src[4 * i + 3];
            if (!Miss(c0.z))
            {
                float blurred = 0.0f;
                for (int dy = -1; dy <= 1; dy++)
                    for (int dx = -1; dx <= 1; dx++)
                    {
                        const int qx = ClampI(x + dx, 0, W - 1), qy = ClampI(y + dy, 0, H - 1);
                        const float k = (dx == 0 ? 0.5f : 0.25f) * (dy == 0 ? 0.5f : 0.25f);
                        blurred = zr_fma(k, src[4 * ((size_t)qy * W + qx) + 3], blurred);
                    }
                const float rcpPhiL = 1.0f / zr_fma(prm.sigmaL, zr_sqrt(zr_max(0.0f, blurred)), 1e-4f);
                const float rcpPhiZ = 1.0f / ((prm.sigmaZ * zr_max(c0.fw, 1e-8f)) * (float)step);
                const float lum = Lum(f3(col[0], col[1], col[2]));
                float ws = 1.0f;
                for (int dy = -2; dy <= 2; dy++)
                    for (int dx = -2; dx <= 2; dx++)
                    {
                        if (dx == 0 && dy == 0) continue;
                        const int qx = x + dx * step, qy = y + dy * step;
                        const bool inside = qx >= 0 && qy >= 0 && qx < W && qy < H;
                        const size_t j = (size_t)ClampI(qy, 0, H - 1) * W + ClampI(qx, 0, W - 1);
                        const float3 cq = f3(src[4 * j], src[4 * j + 1], src[4 * j + 2]);
                        const float h = kB3[dx < 0 ? -dx : dx] * kB3[dy < 0 ? -dy : dy];
                        const float wl = zr_abs(lum - Lum(cq)) * rcpPhiL;
                        const float rcpDist = 1.0f / zr_sqrt((float)(dx * dx + dy * dy));
                        const float wz = zr_abs(c0.z - g[j].z) * (rcpPhiZ * rcpDist);
                        float w = (h * E16(wl + wz)) * PowNormal(c0.n, g[j].n, prm.normalPowerLog2);
                        if (!inside || Miss(g[j].z)) w = 0.0f;
                        col[0] = zr_fma(w, cq.x, col[0]); col[1] = zr_fma(w, cq.y, col[1]); col[2] = zr_fma(w, cq.z, col[2]);
                        var = zr_fma(w * w, src[4 * j + 3], var);
                        ws = ws + w;
                    }
                const float r = 1.0f / ws;
                col[0] = col[0] * r; col[1] = col[1] * r; col[2] = col[2] * r;
                var = var * (r * r);
            }
            dst[4 * i] = col[0]; dst[4 * i + 1] = col[1]; dst[4 * i + 2] = col[2]; dst[4 * i + 3] = var;
        }
}

// One frame.  hist_color (RGBA32F: rgb + length) and hist_moments are read as the previous frame's history and overwritten with this frame's.
static inline void Frame(const float* signal, const float* depth, const uint32_t* normal, const uint32_t* motion, const float* prevDepth,
    const uint32_t* prevNormal, float* histColor, float* histMoments, bool temporalValid, const Params& prm, int W, int H, float* out)
{
    const size_t n = (size_t)W * H;
    std::vector<float> accum(4 * n), moments(2 * n), a(4 * n), b(4 * n);
    const History h = {histColor, histMoments, prevDepth, prevNormal};
    Temporal(signal, depth, normal, motion, h, temporalValid, prm, W, H, accum.data(), moments.data());
    const std::vector<Guide> g = BuildGuide(depth, normal, W, H);
    Variance(accum.data(), moments.data(), g, prm, W, H, a.data());
    std::memcpy(histMoments, moments.data(), 2 * n * sizeof(float));
    auto feedback = [&](const float* filtered) { for (size_t i = 0; i < n; i++) { histColor[4 * i] = filtered[4 * i]; histColor[4 * i + 1] = filtered[4 * i + 1]; histColor[4 * i + 2] = filtered[4 * i + 2]; histColor[4 * i + 3] = accum[4 * i + 3]; } };
    if (prm.iterations == 0) feedback(a.data());
    else ColourThroughHalf(a.data(), n);            // a stage's output that another stage reads is stored with fp16 colour; history and the last stage's stay fp32
    float* src = a.data(); float* dst = b.data();
    for (uint32_t it = 0; it < prm.iterations; it++)
    {
        Atrous(src, g, prm, W, H, 1 << it, dst);
        if (it == 0) feedback(dst);
        if (it + 1 != prm.iterations) ColourThroughHalf(dst, n);
        float* t = src; src = dst; dst = t;
    }
    std::memcpy(out, src, 4 * n * sizeof(float));
}

} // namespace SVGF
} // namespace zro
