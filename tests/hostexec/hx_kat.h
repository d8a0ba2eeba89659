// tests/hostexec -- TEST-ONLY.  The function-level known-answer probes of oracle/zro_kat_layout.h over the product's HIP stage
// functions (zetaray_amd/csrc/zr_dev_math.h, zr_dev_bsdf.h, zr_rpt.h, zr_sdi.h), compiled for the host.  tests/test_ref_pins.py
// requires them to agree bit for bit with the reference's own shader code compiled as C++ (oracle/_ref/libzref_hlsl.so, or the
// committed tests/golden/ref_hlsl_kat.npz where /root/reference is absent).  Columns of functions the product never needs are 0.
#pragma once
#include "../../oracle/zro_kat_layout.h"

namespace hxkat {
using namespace zr;
static inline float F(uint32_t u) { return zr_asfloat(u); }
static inline uint32_t U(float f) { return zr_asuint(f); }

static void Sampling_(const float* in, float* out, uint32_t n)
{
    for (uint32_t i = 0; i < n; i++)
    {
        const float* p = in + ZR_KAT_SAMPLING_IN * i; float* o = out + ZR_KAT_SAMPLING_OUT * i;
        for (int k = 0; k < ZR_KAT_SAMPLING_OUT; k++) o[k] = 0.0f;
        V2 u = v2(p[0], p[1]);
        float pdf;
        V3 a = SampleCosineWeightedHemisphere(u, &pdf); o[4] = a.x; o[5] = a.y; o[6] = a.z; o[7] = pdf;
        a = UniformSampleCone(u, p[2], &pdf); o[8] = a.x; o[9] = a.y; o[10] = a.z; o[11] = pdf;
        V2 d = UniformSampleDiskConcentric(u); o[14] = d.x; o[15] = d.y;
        d = UniformSampleTriangle(u); o[19] = d.x; o[20] = d.y;
        const uint32_t seed = U(p[3]);
        Rng r = Rng::Init(seed & 0xfffu, (seed >> 12) & 0xfffu, seed >> 24);
        o[21] = F(r.s);
        o[22] = r.Uniform();
        o[23] = F(r.UniformUintBounded(1u + (seed % 1000u)));
        o[24] = F(r.UniformUintBounded_Faster(1u + (seed % 977u)));
        V2 u2 = r.Uniform2D(); o[25] = u2.x; o[26] = u2.y;
        { uint32_t x = seed & 0xfffu, y = (seed >> 12) & 0xfffu, z = seed >> 24, w = seed & 7u; zr_pcg4d(&x, &y, &z, &w); o[27] = F(x); }
        o[28] = F(zr_pcg(seed + zr_pcg(seed >> 24)));
        uint32_t hx = seed, hy = seed * 3u, hz = seed ^ 0x9e3779b9u; zr_pcg3d(&hx, &hy, &hz); o[29] = F(hx); o[30] = F(hy); o[31] = F(hz);
    }
}

static void Math_(const float* in, float* out, uint32_t n)
{
    for (uint32_t i = 0; i < n; i++)
    {
        const float* p = in + ZR_KAT_MATH_IN * i; float* o = out + ZR_KAT_MATH_OUT * i;
        for (int k = 0; k < ZR_KAT_MATH_OUT; k++) o[k] = 0.0f;
        V3 v = v3(p[0], p[1], p[2]);
        V4 q = v4(p[3], p[4], p[5], p[6]);
        V3 s = v3(p[7], p[8], p[9]), t = v3(p[10], p[11], p[12]);
        float x = p[13];
        V2 e = EncodeUnitVector(v); o[0] = e.x; o[1] = e.y;
        V3 dv = DecodeUnitVector(e); o[2] = dv.x; o[3] = dv.y; o[4] = dv.z;
        const uint16_t o32[2] = {(uint16_t)FloatToUNorm16(e.x), (uint16_t)FloatToUNorm16(e.y)}; o[5] = F(o32[0]); o[6] = F(o32[1]);
        V3 d32 = DecodeOct32(o32); o[7] = d32.x; o[8] = d32.y; o[9] = d32.z;
        const uint16_t un[2] = {(uint16_t)FloatToUNorm16(p[14]), (uint16_t)FloatToUNorm16(p[15])}; o[10] = F(un[0]); o[11] = F(un[1]);
        o[12] = zr_div65535((float)un[0]); o[13] = zr_div65535((float)un[1]);
        o[14] = F(Float3ToRGB8(saturate(v3(zr_abs(v.x), zr_abs(v.y), zr_abs(v.z)))));
        V3 rgb = UnpackRGB8(U(p[16]) & 0xffffffu); o[15] = rgb.x; o[16] = rgb.y; o[17] = rgb.z;
        V3 rv = RotateVector(v, q); o[18] = rv.x; o[19] = rv.y; o[20] = rv.z;
        V3 tr = TransformTRS(v, t, q, s); o[21] = tr.x; o[22] = tr.y; o[23] = tr.z;
        V3 it = InverseTransformTRS(tr, t, q, s); o[24] = it.x; o[25] = it.y; o[26] = it.z;
        ONB onb = BuildONB(v);
        o[27] = onb.b1.x; o[28] = onb.b1.y; o[29] = onb.b1.z; o[30] = onb.b2.x; o[31] = onb.b2.y; o[32] = onb.b2.z;
        o[33] = ArcCos(x);
        V2 sph = SphericalFromCartesian(v); o[34] = sph.x; o[35] = sph.y;
        o[36] = NextFloat32(x); o[37] = PrevFloat32(x);
        const uint16_t u4[4] = {(uint16_t)(U(p[16]) & 0xffffu), (uint16_t)(U(p[16]) >> 16), (uint16_t)(U(p[17]) & 0xffffu), (uint16_t)(U(p[17]) >> 16)};
        V4 dq = DecodeNormalized4(u4); o[38] = dq.x; o[39] = dq.y; o[40] = dq.z; o[41] = dq.w;
        V3 w2l = sdi::WorldToTangentFrame(v, v3(p[18], p[19], p[20])); o[42] = w2l.x; o[43] = w2l.y; o[44] = w2l.z;
        V3 l2w = sdi::FromTangentFrameToWorld(v, v3(p[18], p[19], p[20])); o[45] = l2w.x; o[46] = l2w.y; o[47] = l2w.z;
    }
}

static void RT_(const float* in, float* out, uint32_t n)
{
    for (uint32_t i = 0; i < n; i++)
    {
        const float* p = in + ZR_KAT_RT_IN * i; float* o = out + ZR_KAT_RT_OUT * i;
        for (int k = 0; k < ZR_KAT_RT_OUT; k++) o[k] = 0.0f;
        V3 pos = v3(p[0], p[1], p[2]), nrm = v3(p[3], p[4], p[5]);
        V3 a = OffsetRayRTG(pos, nrm); o[0] = a.x; o[1] = a.y; o[2] = a.z;
        a = OffsetRayRTG(pos, -nrm); o[3] = a.x; o[4] = a.y; o[5] = a.z;
        o[7] = BalanceHeuristic3(p[9], p[10], p[11], p[9]);
        o[8] = PowerHeuristic(p[9], p[10], v3(p[11]), 1.0f, 1.0f).x;
        V3 d = GeneratePinholeCameraRay_CS((int)(uint32_t)(p[12] * 1920.0f), (int)(uint32_t)(p[13] * 1080.0f), v2(1920.0f, 1080.0f), 1920.0f / 1080.0f, p[14], v2(p[15], p[16]));
        o[9] = d.x; o[10] = d.y; o[11] = d.z;
    }
}

static void BSDF_(const uint16_t* rhoData, const uint32_t* rhoDim, const float* in, float* out, uint32_t n)
{
    RhoView rho; rho.data = rhoData; rho.dx = rhoDim[0]; rho.dy = rhoDim[1]; rho.dz = rhoDim[2];
    for (uint32_t i = 0; i < n; i++)
    {
        const float* p = in + ZR_KAT_BSDF_IN * i; float* o = out + ZR_KAT_BSDF_OUT * i;
        for (int k = 0; k < ZR_KAT_BSDF_OUT; k++) o[k] = 0.0f;
        V3 nrm = v3(p[0], p[1], p[2]), wo = v3(p[3], p[4], p[5]), wi = v3(p[6], p[7], p[8]);
        const bool metallic = p[9] > 0.5f; const float roughness = p[10]; V3 base = v3(p[11], p[12], p[13]);
        const bool specTr = p[14] > 0.5f; const float coat_w = p[15]; V3 coat_c = v3(p[16], p[17], p[18]);
        const float coat_r = p[19], eta_coat = p[20], ior = p[21];
        const bool exiting = p[24] > 0.5f;
        const uint32_t seed = U(p[25]);
        const float eta_curr = exiting ? ior : 1.0f, eta_next = exiting ? 1.0f : ior;
        Surface s = InitSurface(nrm, wo, metallic, roughness, base, eta_curr, eta_next, specTr, p[22], p[23], coat_w, coat_c, coat_r, eta_coat);
        o[0] = s.alpha; o[1] = s.eta; o[2] = s.g_wo; o[3] = s.coat_alpha; o[4] = s.coat_eta; o[5] = s.ndotwo;
        V3 wh = s.SetWi(wi, nrm);
        o[6] = wh.x; o[7] = wh.y; o[8] = wh.z;
        o[9] = s.ndotwi; o[10] = s.ndotwh; o[11] = s.whdotwo; o[12] = s.whdotwi; o[13] = s.wodotwi;
        o[14] = F((s.invalid ? 1u : 0u) | (s.reflection ? 2u : 0u) | (s.backfacing_wo ? 4u : 0u));
        Eval e = Unified(rho, s);
        o[15] = e.f.x; o[16] = e.f.y; o[17] = e.f.z; o[18] = e.Fr_g.x; o[19] = e.Fr_g.y; o[20] = e.Fr_g.z; o[21] = F(e.tir ? 1u : 0u);
        Rng rng = Rng::Seed(seed);
        BsdfSample bs = SampleBSDF(rho, nrm, s, rng);
        o[22] = bs.wi.x; o[23] = bs.wi.y; o[24] = bs.wi.z; o[25] = bs.pdf;
        o[26] = bs.bsdfOverPdf.x; o[27] = bs.bsdfOverPdf.y; o[28] = bs.bsdfOverPdf.z;
        o[29] = bs.f.x; o[30] = bs.f.y; o[31] = bs.f.z; o[32] = F(bs.lobe); o[33] = F(rng.s);
        Rng rng2 = Rng::Seed(seed ^ 0x5bd1e995u);
        o[34] = BSDFSamplerPdf(rho, nrm, s, wi, rng2); o[35] = F(rng2.s);
        Rng rng3 = Rng::Seed(seed);
        rpt::SamplerEval se = rpt::EvalBSDFSampler(rho, nrm, s, bs.wi, bs.lobe, rng3);
        o[36] = se.pdf; o[37] = se.bsdfOverPdf.x; o[38] = se.bsdfOverPdf.y; o[39] = se.bsdfOverPdf.z; o[40] = se.f.x; o[41] = se.f.y; o[42] = se.f.z;
        Rng rng4 = Rng::Seed(seed + 17u);
        V2 u_c = rng4.Uniform2D(); V2 u_g = rng4.Uniform2D(); float w0 = rng4.Uniform(); float w1 = rng4.Uniform();
        BsdfSample nd = SampleBSDF_NoDiffuse(rho, nrm, s, u_c, u_g, w0, w1);
        o[43] = nd.wi.x; o[44] = nd.wi.y; o[45] = nd.wi.z; o[46] = nd.pdf; o[47] = nd.bsdfOverPdf.x; o[48] = nd.bsdfOverPdf.y; o[49] = nd.bsdfOverPdf.z;
        o[50] = F(nd.lobe);
        o[51] = BSDFSamplerPdf_NoDiffuse(rho, nrm, s, wi);
        const float a2 = zr_max(s.alpha, 1e-4f) * zr_max(s.alpha, 1e-4f);
        o[52] = GGX(s.ndotwh, a2);
        o[53] = SmithG2OverG1(a2, s.ndotwi, s.ndotwo);
        o[54] = GGXReflectance_Dielectric(rho, zr_max(s.alpha, 0.002025f), s.ndotwo, 1.0f / 1.5f);
        V2 uu = v2(zr_asfloat((seed >> 9) | 0x3f800000u) - 1.0f, zr_asfloat(((seed * 747796405u) >> 9) | 0x3f800000u) - 1.0f);
        V3 whs = SampleGGXMicrofacet(wo, s.alpha, nrm, uu); o[58] = whs.x; o[59] = whs.y; o[60] = whs.z;
    }
}
} // namespace hxkat
