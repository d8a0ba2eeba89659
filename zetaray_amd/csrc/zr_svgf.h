// zr_svgf.h -- per-pixel stage functions of the denoise pass (ZR_PASS_DENOISE): a spatiotemporal variance-guided filter in the manner of
// Schied et al., "Spatiotemporal Variance-Guided Filtering" (HPG 2017): temporal accumulation of colour and luminance moments, a variance
// estimate (temporal once a pixel has 4 frames of history, a 7 x 7 bilateral one before), and N a-trous wavelet iterations whose edge-stopping
// weights come from depth, normal and the variance-normalised luminance difference.
//
// NO REFERENCE COUNTERPART: the reference has no denoiser (it presents ReSTIR PT through TAA / FSR2).  The pass exists because BASELINE.json's
// config 5 names "ReSTIR PT + SVGF denoise tile pass" at 3840 x 2160.  Its arithmetic is therefore DEFINED HERE (and restated independently
// in oracle/zro_svgf.h, which the tests compare bit for bit): parity "unpinned" by construction, DESIGN.md section 5.13.
//
// Definition (fp32, operations in the order written, -ffp-contract=off, zr_exp of zr_detmath.h; pixel p = (x, y), W x H image):
//   guide(p)     = (z, fw, n): z = linear depth of the G-buffer (FLT_MAX = miss), n = the decoded oct32 normal, fw = max(|z(x+1, y) - z|, |z(x, y+1) - z|)
//                  with the neighbour replaced by the one on the other side at the last column / row and differences to a miss counted as 0
//   temporal(p)  : c = signal.rgb (NaN -> 0), l = Luminance(c).  History position q = (uv - motion) * (W, H) - 0.5 with uv = (p + 0.5) / (W, H);
//                  the four texels around q with bilinear weights, a texel usable when inside the image, not a miss in the previous G-buffer,
//                  |z_prev - z| <= 0.1 * z and dot(n_prev, n) >= 0.9; if their weight sum is <= 0.01 the nine texels around round(q) are tried
//                  with weight 1 each.  Usable history: colour / moments / length = weighted means, length' = min(length + 1, 255),
//                  a_c = max(alpha, 1 / length'), a_m = max(alpha_moments, 1 / length'), accumulated = hist + a * (new - hist).
//                  No usable history (or a miss, or temporal_valid == 0): accumulated = new, length' = 1.
//   variance(p)  : length' >= 4: max(0, m2 - m1 * m1), colour unchanged.  Else the 7 x 7 neighbourhood with weights
//                  w = exp(0 - wz) * wn (centre 1), wz = |z - zq| * ((1 / (sigma_z * max(fw, 1e-8))) * (1 / sqrt(dx^2 + dy^2))), wn as below:
//                  colour = sum(w c) / sum(w), moments likewise, variance = max(0, m2 - m1 * m1) * (4 / length').
//   atrous_i(p)  : step s = 2^i; v3 = 3 x 3 binomial of the variance (1/4, 1/8, 1/16), phi_l = sigma_l * sqrt(max(0, v3)) + 1e-4,
//                  phi_z = sigma_z * max(fw, 1e-8) * s; taps q = p + s * (dx, dy), dx, dy in -2..2 except (0, 0), inside the image, not a miss:
//                  w = h(dx) h(dy) * exp(0 - wl - wz) * wn, h = {1, 2/3, 1/6}[|d|], wl = |l - lq| * (1 / phi_l), wz = |z - zq| * ((1 / phi_z) * (1 / sqrt(dx^2 + dy^2))),
//                  wn = max(0, dot(n, nq)) raised to 2^normal_power_log2 by repeated squaring.
//                  colour' = (c + sum w cq) / (1 + sum w), variance' = (v + sum w^2 vq) / (1 + sum w)^2.  Miss pixels pass through.
//                  The colour after iteration 0 (or after the variance stage when there are no iterations) is the next frame's colour history.
#pragma once
#include "zr_rpt.h"      // DecodeMotion, GBuf

namespace zr {
namespace svgf {

struct SvgfParams { float alpha, alphaMoments, sigmaL, sigmaZ; uint32_t normalPowerLog2, iterations; };

struct SvgfFrame
{
    const F4* signal;                                        // RGBA32F noisy radiance of this frame
    const float* depth; const uint32_t* normal; const uint32_t* motion;      // this frame's G-buffer planes
    const float* prevDepth; const uint32_t* prevNormal;      // the previous frame's
    const F4* histColor; const float* histMoments;           // previous frame: rgb + history length; (m1, m2) per pixel
    F4* accum; float* moments;                               // this frame's accumulated colour + length, moments (become the history)
    F4* guide; float* guideFw;                               // (n.x, n.y, n.z, z) that every tap reads; fw, which only the centre needs
    uint32_t w, h, temporalValid;
    SvgfParams prm;
};

ZR_HD V3 Sanitize3Z(V3 c) { return any_nan(c) ? v3(0.0f) : c; }

// guide planes of pixel (x, y)
ZR_HD void MakeGuide(const float* depth, const uint32_t* normal, int x, int y, int W, int H, F4* guide, float* guideFw)
{
    const size_t i = (size_t)y * W + x;
    const float z = depth[i];
    float fw = 0.0f;
    if (z != ZR_FLT_MAX)
    {
        const int xn = x + 1 < W ? x + 1 : x - 1, yn = y + 1 < H ? y + 1 : y - 1;
        float dx = 0.0f, dy = 0.0f;
        if (xn >= 0) { const float zn = depth[(size_t)y * W + xn]; if (zn != ZR_FLT_MAX) dx = zr_abs(zn - z); }
        if (yn >= 0) { const float zn = depth[(size_t)yn * W + x]; if (zn != ZR_FLT_MAX) dy = zr_abs(zn - z); }
        fw = zr_max(dx, dy);
    }
    guide[i] = f4(DecodeOct32u(normal[i]), z);
    guideFw[i] = fw;
}

ZR_HD float NormalWeight(V3 n, V3 nq, uint32_t powerLog2)
{
    float d = zr_max(0.0f, dot(n, nq));
    for (uint32_t k = 0; k < powerLog2; k++) d = d * d;
    return d;
}

// usable history texel? (previous G-buffer against this pixel's depth / normal)
ZR_HD bool HistoryUsable(const SvgfFrame& F, int qx, int qy, float z, V3 n)
{
    if (qx < 0 || qy < 0 || qx >= (int)F.w || qy >= (int)F.h) return false;
    const size_t j = (size_t)qy * F.w + qx;
    const float zp = F.prevDepth[j];
    if (zp == ZR_FLT_MAX) return false;
    if (!(zr_abs(zp - z) <= 0.1f * z)) return false;
    return dot(DecodeOct32u(F.prevNormal[j]), n) >= 0.9f;
}

ZR_HD void TemporalPixel(const SvgfFrame& F, int x, int y)
{
    const int W = (int)F.w, H = (int)F.h;
    const size_t i = (size_t)y * W + x;
    MakeGuide(F.depth, F.normal, x, y, W, H, F.guide, F.guideFw);
    const F4 s = F.signal[i];
    const V3 c = Sanitize3Z(v3(s.x, s.y, s.z));
    const float l = Luminance(c);
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
            const size_t j = (size_t)(iy + oy) * W + (ix + ox);
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
                    const size_t j = (size_t)(ry + oy) * W + (rx + ox);
                    const F4 h4 = F.histColor[j];
                    hc = hc + v3(h4.x, h4.y, h4.z); hlen += h4.w;
                    hm1 += F.histMoments[2 * j]; hm2 += F.histMoments[2 * j + 1];
                    wsum += 1.0f;
                }
        }
        if (wsum > 0.01f)
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
    const F4* src;            // rgb + variance (a-trous) / rgb + history length (variance stage)
    const float* moments;     // variance stage only
    const F4* guide; const float* guideFw;
    F4* dst;                  // rgb + variance
    F4* history;              // != null: the filtered rgb also goes here with the history length of `lenSrc` (colour history of the next frame)
    const F4* lenSrc;
    uint32_t w, h, step;
    SvgfParams prm;
};

// variance stage
ZR_HD void VariancePixel(const FilterFrame& F, int x, int y)
{
    const int W = (int)F.w, H = (int)F.h;
    const size_t i = (size_t)y * W + x;
    const F4 a = F.src[i];
    const F4 g = F.guide[i];
    const float z = g.w, len = a.w;
    V3 c = v3(a.x, a.y, a.z);
    float m1 = F.moments[2 * i], m2 = F.moments[2 * i + 1];
    float var;
    if (z == ZR_FLT_MAX) var = 0.0f;
    else if (len >= 4.0f) var = zr_max(0.0f, m2 - m1 * m1);
    else
    {
        const V3 n = v3(g.x, g.y, g.z);
        const float invPhiZ = 1.0f / (F.prm.sigmaZ * zr_max(F.guideFw[i], 1e-8f));
        float wsum = 1.0f;
        for (int dy = -3; dy <= 3; dy++)
            for (int dx = -3; dx <= 3; dx++)
            {
                if (dx == 0 && dy == 0) continue;
                const int qx = x + dx, qy = y + dy;
                if (qx < 0 || qy < 0 || qx >= W || qy >= H) continue;
                const size_t j = (size_t)qy * W + qx;
                const F4 gq = F.guide[j];
                const float zq = gq.w;
                if (zq == ZR_FLT_MAX) continue;
                const float wz = zr_abs(z - zq) * (invPhiZ * (1.0f / zr_sqrt((float)(dx * dx + dy * dy))));
                const float wgt = zr_exp(0.0f - wz) * NormalWeight(n, v3(gq.x, gq.y, gq.z), F.prm.normalPowerLog2);
                const F4 q = F.src[j];
                c = c + wgt * v3(q.x, q.y, q.z);
                m1 += wgt * F.moments[2 * j]; m2 += wgt * F.moments[2 * j + 1];
                wsum += wgt;
            }
        c = c / wsum; m1 = m1 / wsum; m2 = m2 / wsum;
        var = zr_max(0.0f, m2 - m1 * m1) * (4.0f / len);
    }
    F.dst[i] = f4(c, var);
    if (F.history) F.history[i] = f4(c, len);
}

// where an a-trous iteration reads its taps from: the planes, or a tile of them staged in LDS (zr_api.hip: k_svgf_atrous_lds)
struct PlaneTaps
{
    const F4* src; const F4* guide; int W;
    ZR_HDM F4 Src(int x, int y) const { return src[(size_t)y * W + x]; }
    ZR_HDM F4 Guide(int x, int y) const { return guide[(size_t)y * W + x]; }
};

// one a-trous iteration
template<class Taps>
ZR_HD void AtrousPixelT(const FilterFrame& F, int x, int y, const Taps& taps)
{
    const int W = (int)F.w, H = (int)F.h, s = (int)F.step;
    const size_t i = (size_t)y * W + x;
    const F4 a = taps.Src(x, y);
    const F4 g = taps.Guide(x, y);
    const float z = g.w;
    V3 c = v3(a.x, a.y, a.z); float var = a.w;
    if (z != ZR_FLT_MAX)
    {
        // 3 x 3 binomial of the variance (clamped addressing)
        float v3x3 = 0.0f;
        for (int dy = -1; dy <= 1; dy++)
            for (int dx = -1; dx <= 1; dx++)
            {
                const int qx = x + dx < 0 ? 0 : (x + dx >= W ? W - 1 : x + dx), qy = y + dy < 0 ? 0 : (y + dy >= H ? H - 1 : y + dy);
                const float k = (dx == 0 ? 0.5f : 0.25f) * (dy == 0 ? 0.5f : 0.25f);
                v3x3 += k * taps.Src(qx, qy).w;
            }
        const float invPhiL = 1.0f / (F.prm.sigmaL * zr_sqrt(zr_max(0.0f, v3x3)) + 1e-4f);
        const float invPhiZ = 1.0f / (F.prm.sigmaZ * zr_max(F.guideFw[i], 1e-8f) * (float)s);
        const V3 n = v3(g.x, g.y, g.z);
        const float l = Luminance(c);
        float wsum = 1.0f;
        for (int dy = -2; dy <= 2; dy++)
            for (int dx = -2; dx <= 2; dx++)
            {
                if (dx == 0 && dy == 0) continue;
                const int qx = x + dx * s, qy = y + dy * s;
                if (qx < 0 || qy < 0 || qx >= W || qy >= H) continue;
                const F4 gq = taps.Guide(qx, qy);
                const float zq = gq.w;
                if (zq == ZR_FLT_MAX) continue;
                const F4 q = taps.Src(qx, qy);
                const V3 cq = v3(q.x, q.y, q.z);
                const int ax = dx < 0 ? -dx : dx, ay = dy < 0 ? -dy : dy;
                const float hx = ax == 0 ? 1.0f : (ax == 1 ? 2.0f / 3.0f : 1.0f / 6.0f), hy = ay == 0 ? 1.0f : (ay == 1 ? 2.0f / 3.0f : 1.0f / 6.0f);
                const float wl = zr_abs(l - Luminance(cq)) * invPhiL;
                const float wz = zr_abs(z - zq) * (invPhiZ * (1.0f / zr_sqrt((float)(dx * dx + dy * dy))));
                const float wgt = ((hx * hy) * zr_exp((0.0f - wl) - wz)) * NormalWeight(n, v3(gq.x, gq.y, gq.z), F.prm.normalPowerLog2);
                c = c + wgt * cq;
                var += (wgt * wgt) * q.w;
                wsum += wgt;
            }
        c = c / wsum;
        var = var / (wsum * wsum);
    }
    F.dst[i] = f4(c, var);
    if (F.history) F.history[i] = f4(c, F.lenSrc[i].w);
}
ZR_HD void AtrousPixel(const FilterFrame& F, int x, int y)
{ PlaneTaps t; t.src = F.src; t.guide = F.guide; t.W = (int)F.w; AtrousPixelT(F, x, y, t); }

} // namespace svgf
} // namespace zr
