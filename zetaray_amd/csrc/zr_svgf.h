// zr_svgf.h -- per-pixel stage functions of the denoise pass (ZR_PASS_DENOISE): a spatiotemporal variance-guided filter in the manner of
// Schied et al., "Spatiotemporal Variance-Guided Filtering" (HPG 2017): temporal accumulation of colour and luminance moments, a variance
// estimate (temporal once a pixel has 4 frames of history, a 7 x 7 bilateral one before), and N a-trous wavelet iterations whose edge-stopping
// weights come from depth, normal and the variance-normalised luminance difference.
//
// NO REFERENCE COUNTERPART: the reference has no denoiser (it presents ReSTIR PT through TAA / FSR2).  The pass exists because BASELINE.json's
// config 5 names "ReSTIR PT + SVGF denoise tile pass" at 3840 x 2160.  Its arithmetic is therefore DEFINED HERE (and restated independently
// in oracle/zro_svgf.h, which the tests compare bit for bit): parity "unpinned" by construction, DESIGN.md section 5.13.
//
// DEFINITION, VERSION 3 (round 4; version 2 + item (6)).  What changed against version 1, and why: the a-trous kernel measured VALU-bound, not bandwidth-bound
// (profiles/r04b_post_sq*.csv: 1622 VALU instructions per pixel and iteration, SIMD VALU ~0.9 busy, 0.03 of the HBM roof), two thirds of it the
// Cephes exp, unfused multiply-adds and per-tap branches.  Version 2 keeps the filter (5 x 5 B3 taps, depth / normal / luminance edge stops,
// variance propagation) and prices its arithmetic for the machine: (1) the edge-stopping falloff is the compact-support E(x) = max(0, 1 - x / 16)^16
// (four squarings; e^-x to 1.5 % up to x = 1, zero from x = 16) instead of exp; (2) fused multiply-adds are WRITTEN (zr_fma: correctly rounded on both
// sides, so HIP == oracle stays bit for bit) in the luminance, the dot products and every accumulation; (3) taps are PREDICATED, not skipped: every
// tap position is clamped into the image, loaded, weighted, and its weight replaced by 0 when the tap is not usable -- no per-tap branch; (4) one
// reciprocal per pixel instead of four divisions; (5) non-finite signal values are treated as 0 and the radiance is clamped to +-1e15, so every
// value downstream is finite (0 x finite = 0 is what makes (3) exact) and an overflowed sample can no longer poison the history;
// (6) with (1) - (4) in place the iterations that gather their taps from the planes (steps 8 and 16; 4 without its LDS tile) are bound by the L1's 64 B per
// clock and CU -- 25 taps x 32 B per pixel -- so the planes BETWEEN the stages hold what a tap needs in 20 B instead: colour and normal as fp16
// (round to nearest even), variance and depth as fp32.  The history planes and the pass's output stay fp32: rounding happens between filter stages only.
//
// Definition (fp32, operations in the order written, -ffp-contract=off, fma(a, b, c) = zr_fma; pixel p = (x, y) in FRAME coordinates, W x H frame):
//   Lum(c)       = fma(0.2126, c.x, fma(0.7152, c.y, 0.0722 * c.z))
//   E(x)         = t^16, t = max(0, fma(x, -0.0625, 1)), by squaring four times
//   Nw(n, nq)    = max(0, fma(n.x, nq.x, fma(n.y, nq.y, n.z * nq.z))) raised to 2^normal_power_log2 by repeated squaring
//   h(v)         = v rounded to fp16 (round to nearest even) and back
//   guide(p)     = (n, z, fw): z = linear depth of the G-buffer (FLT_MAX = miss), n = h(the decoded oct32 normal), fw = max(|z(x+1, y) - z|, |z(x, y+1) - z|)
//                  with the neighbour replaced by the one on the other side at the last column / row and differences to a miss counted as 0
//   temporal(p)  : c = signal.rgb, (0, 0, 0) if a component is not finite, then each component clamped to [-60000, 60000] (fp16's range); l = Lum(c).
//                  History position q = (uv - motion) * (W, H) - 0.5 with uv = (p + 0.5) / (W, H); the four texels around q with bilinear weights,
//                  a texel usable when inside the image, not a miss in the previous G-buffer, |z_prev - z| <= 0.1 * z and dot(n_prev, n) >= 0.9;
//                  if their weight sum is <= 0.01 the nine texels around round(q) are tried with weight 1 each.
//                  Usable history (weight sum > 0.01 and the weighted sums of colour, length and moments all finite): colour / moments / length = weighted means, length' = min(length + 1, 255),
//                  a_c = max(alpha, 1 / length'), a_m = max(alpha_moments, 1 / length'), accumulated = hist + a * (new - hist).
//                  No usable history (or a miss, or temporal_valid == 0): accumulated = new, length' = 1.   (unchanged from version 1 but for the sanitising)
//   variance(p)  : length' >= 4: max(0, m2 - m1 * m1), colour unchanged.  Else the 7 x 7 neighbourhood, taps q = p + (dx, dy) != p in row order, position
//                  clamped into the image, w = usable ? E(wz) * Nw(n, nq) : 0 (usable: inside the image and not a miss),
//                  wz = |z - zq| * (rz * (1 / sqrt(dx^2 + dy^2))), rz = 1 / (sigma_z * max(fw, 1e-8)); colour = fma(w, cq, colour), moments likewise,
//                  ws = ws + w from 1; r = 1 / ws, colour * r, m1 * r, m2 * r, variance = max(0, m2 - m1 * m1) * (4 / length').
//   atrous_i(p)  : step s = 2^i; v3 = 3 x 3 binomial of the variance (weights 1/4, 1/8, 1/16, clamped addressing, fma(k, v, v3) from 0 in row order),
//                  rl = 1 / fma(sigma_l, sqrt(max(0, v3)), 1e-4), rz = 1 / ((sigma_z * max(fw, 1e-8)) * s), l = Lum(c);
//                  taps q = p + s * (dx, dy), dx, dy in -2..2 except (0, 0), row order, position clamped into the image:
//                  w = usable ? (h(dx) h(dy) * E(wl + wz)) * Nw(n, nq) : 0, h = {1, 2/3, 1/6}[|d|], wl = |l - Lum(cq)| * rl,
//                  wz = |z - zq| * (rz * (1 / sqrt(dx^2 + dy^2)));  c = fma(w, cq, c), var = fma(w * w, vq, var), ws = ws + w from 1;
//                  r = 1 / ws: colour' = c * r, variance' = var * (r * r).  Miss pixels pass through.
//                  The colour after iteration 0 (or after the variance stage when there are no iterations) is the next frame's colour history.
//   between stages: the colour a later a-trous iteration reads (from the variance stage or the iteration before it) is h(colour); the colour history and
//                  the output of the last stage are the fp32 values the stage computed.
//
// TILES: the planes of a pass may cover a window (ox, oy, pw, ph) of the frame -- a device's tile + its 32-px apron.  Every
// formula above uses frame coordinates and frame dimensions; a position that leaves the window is clamped into it and the tap counted unusable.  A
// window pixel closer to the window's edge than a stage's reach therefore differs from the full frame's -- which is why the host exchanges the
// apron between stages (zetaray_amd/tiling.py: denoise_schedule) before the error can reach an owned pixel.
#pragma once
#include "zr_rpt.h"      // DecodeMotion, GBuf

namespace zr {
namespace svgf {

struct SvgfParams { float alpha, alphaMoments, sigmaL, sigmaZ; uint32_t normalPowerLog2, iterations; };

// the window of the frame that the planes cover: pixel (x, y) of the frame is plane texel (y - oy) * pw + (x - ox)
struct Window
{
    int ox, oy, pw, ph, W, H;
    ZR_HDM bool InImage(int x, int y) const { return x >= 0 && y >= 0 && x < W && y < H; }
    ZR_HDM bool InPlanes(int x, int y) const { return x >= ox && y >= oy && x < ox + pw && y < oy + ph; }
    ZR_HDM int ClampX(int x) const { return x < ox ? ox : (x > ox + pw - 1 ? ox + pw - 1 : x); }
    ZR_HDM int ClampY(int y) const { return y < oy ? oy : (y > oy + ph - 1 ? oy + ph - 1 : y); }
    ZR_HDM size_t Idx(int x, int y) const { return (size_t)(y - oy) * (size_t)pw + (size_t)(x - ox); }
};

// the guide normal of a pixel as the planes hold it: h(n.x) | h(n.y) << 16, h(n.z)
struct GuideN { uint32_t xy, z; };

struct SvgfFrame
{
    const F4* signal;                                        // RGBA32F noisy radiance of this frame
    const float* depth; const uint32_t* normal; const uint32_t* motion;      // this frame's G-buffer planes
    const float* prevDepth; const uint32_t* prevNormal;      // the previous frame's
    const F4* histColor; const float* histMoments;           // previous frame: rgb + history length; (m1, m2) per pixel
    F4* accum; float* moments;                               // this frame's accumulated colour + length, moments (become the history)
    GuideN* guide; float* guideFw;                           // h(n) of every pixel (8 B: what the variance stage reads next to the depth plane); fw, which only a stage's centre needs
    Window win; uint32_t temporalValid;
    SvgfParams prm;
};

ZR_HD bool Finite(float x) { return (zr_asuint(x) & 0x7f800000u) != 0x7f800000u; }
ZR_HD float Lum(V3 c) { return zr_fma(0.2126f, c.x, zr_fma(0.7152f, c.y, 0.0722f * c.z)); }
ZR_HD float Falloff(float x)
{
    float t = zr_max(0.0f, zr_fma(x, -0.0625f, 1.0f));
    t = t * t; t = t * t; t = t * t; t = t * t;
    return t;
}
ZR_HD float Clamp15(float x) { return zr_min(zr_max(x, -60000.0f), 60000.0f); }      // (fp16's finite range: the planes between the stages hold fp16 colour)
// fp16 <-> fp32 for values that are finite by construction (no Inf / NaN special cases): the conversion instructions on the device, the portable code on the host
ZR_HD uint32_t HalfBits(float f)
{
#if defined(__HIP_DEVICE_COMPILE__)
    union { _Float16 h; uint16_t u; } c; c.h = (_Float16)f; return c.u;
#else
    return zr_f32_to_f16_portable(f);
#endif
}
ZR_HD float HalfValue(uint32_t bits)
{
#if defined(__HIP_DEVICE_COMPILE__)
    union { _Float16 h; uint16_t u; } c; c.u = (uint16_t)bits; return (float)c.h;
#else
    return zr_f16_to_f32_portable((uint16_t)bits);
#endif
}
ZR_HD float RoundHalf(float f) { return HalfValue(HalfBits(f)); }
// the 16-byte texel of a plane between two filter stages: {h(r) | h(g) << 16, h(b) | h(n.x) << 16, h(n.y) | h(n.z) << 16, variance}
ZR_HD U4 PackStage(V3 c, float var, V3 n)
{
    U4 p;
    p.x = HalfBits(c.x) | (HalfBits(c.y) << 16); p.y = HalfBits(c.z) | (HalfBits(n.x) << 16); p.z = HalfBits(n.y) | (HalfBits(n.z) << 16); p.w = zr_asuint(var);
    return p;
}
ZR_HD void UnpackStage(const U4& p, float z, F4& gq, F4& q)
{
    q = f4(v3(HalfValue(p.x & 0xffffu), HalfValue(p.x >> 16), HalfValue(p.y & 0xffffu)), zr_asfloat(p.w));
    gq = f4(v3(HalfValue(p.y >> 16), HalfValue(p.z & 0xffffu), HalfValue(p.z >> 16)), z);
}
ZR_HD V3 SanitizeSignal(V3 c)
{
    if (!(Finite(c.x) && Finite(c.y) && Finite(c.z))) return v3(0.0f);
    return v3(Clamp15(c.x), Clamp15(c.y), Clamp15(c.z));
}

// guide planes of pixel (x, y)
ZR_HD void MakeGuide(const float* depth, const uint32_t* normal, int x, int y, const Window& w, GuideN* guide, float* guideFw)
{
    const size_t i = w.Idx(x, y);
    const float z = depth[i];
    float fw = 0.0f;
    if (z != ZR_FLT_MAX)
    {
        const int xn = x + 1 < w.W ? x + 1 : x - 1, yn = y + 1 < w.H ? y + 1 : y - 1;
        float dx = 0.0f, dy = 0.0f;
        if (xn >= 0 && w.InPlanes(xn, y)) { const float zn = depth[w.Idx(xn, y)]; if (zn != ZR_FLT_MAX) dx = zr_abs(zn - z); }
        if (yn >= 0 && w.InPlanes(x, yn)) { const float zn = depth[w.Idx(x, yn)]; if (zn != ZR_FLT_MAX) dy = zr_abs(zn - z); }
        fw = zr_max(dx, dy);
    }
    const V3 n = DecodeOct32u(normal[i]);
    GuideN gn; gn.xy = HalfBits(n.x) | (HalfBits(n.y) << 16); gn.z = HalfBits(n.z);
    guide[i] = gn;
    guideFw[i] = fw;
}

template<int POW>      // POW >= 0: the exponent's log2 as a compile-time constant (the default 7 unrolls); -1: prm.normalPowerLog2 at run time
ZR_HD float NormalWeightT(V3 n, V3 nq, uint32_t powerLog2)
{
    float d = zr_max(0.0f, zr_fma(n.x, nq.x, zr_fma(n.y, nq.y, n.z * nq.z)));
    if (POW >= 0) { for (int k = 0; k < POW; k++) d = d * d; }
    else for (uint32_t k = 0; k < powerLog2; k++) d = d * d;
    return d;
}

// usable history texel? (previous G-buffer against this pixel's depth / normal; finite history)
ZR_HD bool HistoryUsable(const SvgfFrame& F, int qx, int qy, float z, V3 n)
{
    if (!F.win.InImage(qx, qy) || !F.win.InPlanes(qx, qy)) return false;
    const size_t j = F.win.Idx(qx, qy);
    const float zp = F.prevDepth[j];
    if (zp == ZR_FLT_MAX) return false;
    if (!(zr_abs(zp - z) <= 0.1f * z)) return false;
    return dot(DecodeOct32u(F.prevNormal[j]), n) >= 0.9f;
}

ZR_HD void TemporalPixel(const SvgfFrame& F, int x, int y)
{
    const int W = F.win.W, H = F.win.H;
    const size_t i = F.win.Idx(x, y);
    MakeGuide(F.depth, F.normal, x, y, F.win, F.guide, F.guideFw);
    const F4 s = F.signal[i];
    const V3 c = SanitizeSignal(v3(s.x, s.y, s.z));
    const float l = Lum(c);
    V3 acc = c; float m1 = l, m2 = l * l, len = 1.0f;
    const float z = F.depth[i];
    if (z != ZR_FLT_MAX && F.temporalValid)
    {
        const V3 n = DecodeOct32u(F.normal[i]);
        const V2 mv = rpt::DecodeMotion(F.motion[i]);
        const float u = ((float)x + 0.5f) / (float)W - mv.x, v = ((float)y + 0.5f) / (float)H - mv.y;
        const float qx = u * (float)W - 0.5f, qy = v * (float)H - 0.5f;
        const float fx0 = zr_floor(qx), fy0 = zr_floor(qy);
        const float tx = qx - fx0, ty = qy - fy0;
        // (guard the float -> int conversion: a motion vector can point far outside)
        const bool farOut = !(fx0 > -4.0f && fy0 > -4.0f && fx0 < (float)W + 4.0f && fy0 < (float)H + 4.0f);
        const int ix = farOut ? -8 : (int)fx0, iy = farOut ? -8 : (int)fy0;
        V3 hc = v3(0.0f); float hm1 = 0, hm2 = 0, hlen = 0, wsum = 0;
        for (int k = 0; k < 4; k++)
        {
            const int ox = k & 1, oy = k >> 1;
            const float wgt = (ox ? tx : 1.0f - tx) * (oy ? ty : 1.0f - ty);
            if (!HistoryUsable(F, ix + ox, iy + oy, z, n)) continue;
            const size_t j = F.win.Idx(ix + ox, iy + oy);
            const F4 h4 = F.histColor[j];
            hc = hc + wgt * v3(h4.x, h4.y, h4.z); hlen += wgt * h4.w;
            hm1 += wgt * F.histMoments[2 * j]; hm2 += wgt * F.histMoments[2 * j + 1];
            wsum += wgt;
        }
        if (!(wsum > 0.01f))
        {
            hc = v3(0.0f); hm1 = 0; hm2 = 0; hlen = 0; wsum = 0;
            const int rx = farOut ? -8 : (int)zr_floor(qx + 0.5f), ry = farOut ? -8 : (int)zr_floor(qy + 0.5f);
            for (int oy = -1; oy <= 1; oy++)
                for (int ox = -1; ox <= 1; ox++)
                {
                    if (!HistoryUsable(F, rx + ox, ry + oy, z, n)) continue;
                    const size_t j = F.win.Idx(rx + ox, ry + oy);
                    const F4 h4 = F.histColor[j];
                    hc = hc + v3(h4.x, h4.y, h4.z); hlen += h4.w;
                    hm1 += F.histMoments[2 * j]; hm2 += F.histMoments[2 * j + 1];
                    wsum += 1.0f;
                }
        }
        // (a non-finite history value -- there is none unless the planes were corrupted from outside: the signal is sanitised -- makes the sums
        // non-finite; such history counts as none instead of spreading)
        if (wsum > 0.01f && Finite(hc.x) && Finite(hc.y) && Finite(hc.z) && Finite(hlen) && Finite(hm1) && Finite(hm2))
        {
            hc = hc / wsum; hm1 = hm1 / wsum; hm2 = hm2 / wsum; hlen = hlen / wsum;
            len = zr_min(hlen + 1.0f, 255.0f);
            const float ac = zr_max(F.prm.alpha, 1.0f / len), am = zr_max(F.prm.alphaMoments, 1.0f / len);
            acc = hc + ac * (c - hc);
            m1 = hm1 + am * (l - hm1); m2 = hm2 + am * (l * l - hm2);
        }
    }
    F.accum[i] = f4(acc, len);
    F.moments[2 * i] = m1; F.moments[2 * i + 1] = m2;
}

struct FilterFrame
{
    const F4* src;            // variance stage: rgb + history length (fp32); a-trous: the stage texels of PackStage, 16 B each, read through `srcP`
    const float* moments;     // variance stage only
    const GuideN* guide; const float* guideFw;
    const float* guideZ;      // the depth a tap reads next to its normal / stage texel: the G-buffer's linear-depth plane itself
    F4* dst;                  // rgb + variance: fp32 when this is the pass's last stage, else stage texels (PackStage) written through `dstP`
    bool dstPacked;
    F4* history;              // != null: the filtered rgb also goes here with the history length of `lenSrc` (colour history of the next frame)
    const F4* lenSrc;
    Window win; uint32_t step;
    SvgfParams prm;
};

ZR_HD F4 LoadGuide(const GuideN* guide, const float* depth, size_t i)
{ const GuideN gn = guide[i]; return f4(v3(HalfValue(gn.xy & 0xffffu), HalfValue(gn.xy >> 16), HalfValue(gn.z & 0xffffu)), depth[i]); }

// 1 / sqrt(dx^2 + dy^2) for the 7 x 7 stencil of the variance stage: the correctly rounded 1.0f / sqrtf((float)n), n = 0 .. 18, as fp32 literals (the loops
// are real loops there -- the stage's neighbourhood path only runs for pixels with a short history -- and a run-time sqrt + divide per tap would
// double its cost); tests/test_denoise.py checks the table against the expression
ZR_HD float RcpLen(int dx, int dy)
{
    const float kTab[19] = {0.0f, 0x1.000000p+0f, 0x1.6a09e6p-1f, 0x1.279a74p-1f, 0x1.000000p-1f, 0x1.c9f25cp-2f, 0x1.a20bd6p-2f, 0x1.830920p-2f, 0x1.6a09e6p-2f, 0x1.555556p-2f, 0x1.43d136p-2f, 0x1.34bf64p-2f, 0x1.279a74p-2f, 0x1.1c01aap-2f, 0x1.11aceep-2f, 0x1.08654ap-2f, 0x1.000000p-2f, 0x1.f0b686p-3f, 0x1.e2b7e0p-3f};
    return kTab[dx * dx + dy * dy];
}

// variance stage.  (The planes lie inside the image, so "inside the image and inside the planes" is InPlanes, and clamping into the image and then
// into the planes is clamping into the planes.  Column / row offsets and validity are computed once per pixel, not once per tap.)
template<int POW>
ZR_HD void VariancePixelT(const FilterFrame& F, int x, int y)
{
    const Window& w = F.win;
    const size_t i = w.Idx(x, y);
    const F4 a = F.src[i];
    const F4 g = LoadGuide(F.guide, F.guideZ, i);
    const float z = g.w, len = a.w;
    V3 c = v3(a.x, a.y, a.z);
    float m1 = F.moments[2 * i], m2 = F.moments[2 * i + 1];
    float var;
    if (z == ZR_FLT_MAX) var = 0.0f;
    else if (len >= 4.0f) var = zr_max(0.0f, m2 - m1 * m1);
    else
    {
        const V3 n = v3(g.x, g.y, g.z);
        const float rz = 1.0f / (F.prm.sigmaZ * zr_max(F.guideFw[i], 1e-8f));
        float wsum = 1.0f;
        for (int dy = -3; dy <= 3; dy++)
        {
            const int qy = y + dy;
            const bool iny = qy >= w.oy && qy < w.oy + w.ph;
            const size_t row = (size_t)(w.ClampY(qy) - w.oy) * (size_t)w.pw;
            for (int dx = -3; dx <= 3; dx++)
            {
                if (dx == 0 && dy == 0) continue;
                const int qx = x + dx;
                const bool in = iny && qx >= w.ox && qx < w.ox + w.pw;
                const size_t j = row + (size_t)(w.ClampX(qx) - w.ox);
                const F4 gq = LoadGuide(F.guide, F.guideZ, j);
                const float zq = gq.w;
                const float wz = zr_abs(z - zq) * (rz * RcpLen(dx, dy));
                const float wv = Falloff(wz) * NormalWeightT<POW>(n, v3(gq.x, gq.y, gq.z), F.prm.normalPowerLog2);
                const float wgt = (in && zq != ZR_FLT_MAX) ? wv : 0.0f;
                const F4 q = F.src[j];
                c = v3(zr_fma(wgt, q.x, c.x), zr_fma(wgt, q.y, c.y), zr_fma(wgt, q.z, c.z));
                m1 = zr_fma(wgt, F.moments[2 * j], m1); m2 = zr_fma(wgt, F.moments[2 * j + 1], m2);
                wsum = wsum + wgt;
            }
        }
        const float r = 1.0f / wsum;
        c = v3(c.x * r, c.y * r, c.z * r); m1 = m1 * r; m2 = m2 * r;
        var = zr_max(0.0f, m2 - m1 * m1) * (4.0f / len);
    }
    if (F.dstPacked) ((U4*)F.dst)[i] = PackStage(c, var, v3(g.x, g.y, g.z)); else F.dst[i] = f4(c, var);
    if (F.history) F.history[i] = f4(c, len);
}
ZR_HD void VariancePixel(const FilterFrame& F, int x, int y)
{ if (F.prm.normalPowerLog2 == 7u) VariancePixelT<7>(F, x, y); else VariancePixelT<-1>(F, x, y); }

// where an a-trous iteration reads its taps from: the planes, or a tile of them staged in LDS (zr_api.hip: k_svgf_atrous_lds).  Taps are addressed by
// (column, row) already clamped into the planes; an implementation turns a row into a base offset once (RowBase) and adds the column (At)
struct PlaneTaps
{
    const U4* srcP; const float* guideZ; Window win;      // the stage texels (PackStage) + the depth plane: 20 B per tap
    ZR_HDM int RowBase(int y) const { return (y - win.oy) * win.pw - win.ox; }
    ZR_HDM void Load(int rowBase, int x, F4& gq, F4& q) const { const U4 p = srcP[rowBase + x]; UnpackStage(p, guideZ[rowBase + x], gq, q); }
    ZR_HDM float Var(int rowBase, int x) const { return zr_asfloat(srcP[rowBase + x].w); }
};

// B3 tap weights h(d) and 1 / sqrt(dx^2 + dy^2) of the 5 x 5 stencil as fp32 literals (= the correctly rounded 1.0f / sqrtf(n) the definition names;
// tests/test_denoise.py checks the table against that expression): the row-loop form of the kernel indexes them with the row, a run-time value
ZR_HD float TapH(int a) { return a == 0 ? 1.0f : (a == 1 ? 2.0f / 3.0f : 1.0f / 6.0f); }
ZR_HD float TapRcpLen(int ax, int ay)
{
    const int d2 = ax * ax + ay * ay;
    return d2 == 1 ? 1.0f : (d2 == 2 ? 0x1.6a09e6p-1f : (d2 == 4 ? 0.5f : (d2 == 5 ? 0x1.c9f25cp-2f : 0x1.6a09e6p-2f)));
}

// one tap of an a-trous iteration: texels (gq, q) at the clamped position, `ok` = the unclamped position lies in the planes
struct AtrousAcc { V3 c; float var, wsum; };
template<int POW>
ZR_HD void AtrousTap(AtrousAcc& A, const F4& gq, const F4& q, bool ok, float hxy, float rcpLen, float l, float z, V3 n, float rl, float rz, uint32_t powerLog2)
{
    const float zq = gq.w;
    const V3 cq = v3(q.x, q.y, q.z);
    const float wl = zr_abs(l - Lum(cq)) * rl;
    const float wz = zr_abs(z - zq) * (rz * rcpLen);
    const float wv = (hxy * Falloff(wl + wz)) * NormalWeightT<POW>(n, v3(gq.x, gq.y, gq.z), powerLog2);
    const float wgt = (ok && zq != ZR_FLT_MAX) ? wv : 0.0f;
    A.c = v3(zr_fma(wgt, cq.x, A.c.x), zr_fma(wgt, cq.y, A.c.y), zr_fma(wgt, cq.z, A.c.z));
    A.var = zr_fma(wgt * wgt, q.w, A.var);
    A.wsum = A.wsum + wgt;
}

// one a-trous iteration.  ROWLOOP = false: the 24 taps fully unrolled (every constant folded; the scheduler issues all 48 loads first: 128+ VGPRs);
// true: a loop over the five tap rows with the row's constants read from tables (half the registers, a fifth of the code)
template<int POW, bool ROWLOOP, class Taps>
ZR_HD void AtrousPixelT(const FilterFrame& F, int x, int y, const Taps& taps)
{
    const Window& w = F.win;
    const int s = (int)F.step;
    const size_t i = w.Idx(x, y);
    const int row0 = taps.RowBase(y);
    F4 a, g;
    taps.Load(row0, x, g, a);
    const float z = g.w;
    AtrousAcc A; A.c = v3(a.x, a.y, a.z); A.var = a.w; A.wsum = 1.0f;
    if (z != ZR_FLT_MAX)
    {
        // 3 x 3 binomial of the variance (clamped addressing)
        float v3x3 = 0.0f;
        {
            const int xs[3] = {w.ClampX(x - 1), x, w.ClampX(x + 1)};
#pragma unroll
            for (int dy = -1; dy <= 1; dy++)
            {
                const int rb = dy == 0 ? row0 : taps.RowBase(w.ClampY(y + dy));
#pragma unroll
                for (int dx = -1; dx <= 1; dx++)
                {
                    const float k = (dx == 0 ? 0.5f : 0.25f) * (dy == 0 ? 0.5f : 0.25f);
                    v3x3 = zr_fma(k, (dx == 0 && dy == 0) ? a.w : taps.Var(rb, xs[dx + 1]), v3x3);
                }
            }
        }
        const float rl = 1.0f / zr_fma(F.prm.sigmaL, zr_sqrt(zr_max(0.0f, v3x3)), 1e-4f);
        const float rz = 1.0f / ((F.prm.sigmaZ * zr_max(F.guideFw[i], 1e-8f)) * (float)s);
        const V3 n = v3(g.x, g.y, g.z);
        const float l = Lum(A.c);
        // the five tap columns: clamped position and whether the unclamped one lies in the planes
        int cx[5]; bool inx[5];
#pragma unroll
        for (int k = 0; k < 5; k++) { const int qx = x + (k - 2) * s; cx[k] = w.ClampX(qx); inx[k] = qx >= w.ox && qx < w.ox + w.pw; }
        if (ROWLOOP)
        {
#pragma unroll 1
            for (int dy = -2; dy <= 2; dy++)
            {
                const int qy = y + dy * s, ay = dy < 0 ? -dy : dy;
                const bool iny = qy >= w.oy && qy < w.oy + w.ph;
                const int rb = taps.RowBase(w.ClampY(qy));
                const float hy = TapH(ay);
#pragma unroll
                for (int dx = -2; dx <= 2; dx++)
                {
                    if (dx == 0 && dy == 0) continue;
                    const int ax = dx < 0 ? -dx : dx;
                    F4 gq, q;
                    taps.Load(rb, cx[dx + 2], gq, q);
                    AtrousTap<POW>(A, gq, q, iny && inx[dx + 2], TapH(ax) * hy, TapRcpLen(ax, ay), l, z, n, rl, rz, F.prm.normalPowerLog2);
                }
            }
        }
        else
        {
#pragma unroll
            for (int dy = -2; dy <= 2; dy++)
            {
                const int qy = y + dy * s, ay = dy < 0 ? -dy : dy;
                const bool iny = qy >= w.oy && qy < w.oy + w.ph;
                const int rb = dy == 0 ? row0 : taps.RowBase(w.ClampY(qy));
#pragma unroll
                for (int dx = -2; dx <= 2; dx++)
                {
                    if (dx == 0 && dy == 0) continue;
                    const int ax = dx < 0 ? -dx : dx;
                    F4 gq, q;
                    taps.Load(rb, cx[dx + 2], gq, q);
                    AtrousTap<POW>(A, gq, q, iny && inx[dx + 2], TapH(ax) * TapH(ay), TapRcpLen(ax, ay), l, z, n, rl, rz, F.prm.normalPowerLog2);
                }
            }
        }
        const float r = 1.0f / A.wsum;
        A.c = v3(A.c.x * r, A.c.y * r, A.c.z * r);
        A.var = A.var * (r * r);
    }
    if (F.dstPacked) ((U4*)F.dst)[i] = PackStage(A.c, A.var, v3(g.x, g.y, g.z)); else F.dst[i] = f4(A.c, A.var);
    if (F.history) F.history[i] = f4(A.c, F.lenSrc[i].w);
}
template<class Taps>
ZR_HD void AtrousPixelTaps(const FilterFrame& F, int x, int y, const Taps& taps)
{ if (F.prm.normalPowerLog2 == 7u) AtrousPixelT<7, true>(F, x, y, taps); else AtrousPixelT<-1, true>(F, x, y, taps); }
ZR_HD void AtrousPixel(const FilterFrame& F, int x, int y)
{ PlaneTaps t; t.srcP = (const U4*)F.src; t.guideZ = F.guideZ; t.win = F.win; AtrousPixelTaps(F, x, y, t); }

} // namespace svgf
} // namespace zr
