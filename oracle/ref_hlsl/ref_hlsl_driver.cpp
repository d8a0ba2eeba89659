// ORACLE tooling -- test infrastructure only.
//
// C entry points over the REFERENCE's own shader code (Source/ZetaRenderPass/Common/*.hlsli, compiled as C++ from the
// lexically rewritten copies under oracle/_ref/gen/, see hlsl2cpp.py / hlsl_shim.h), built by oracle/_ref.mk into
// oracle/_ref/libzref_hlsl.so.  Used only to pin the oracle and the HIP stage functions (tests/test_ref_pins.py) and to
// generate tests/golden/ref_hlsl_*.npz (tools/make_ref_goldens.py).  Every probe ("KAT family") has the same row layout in the
// three implementations that are compared: this file (reference code), oracle/zro_kat.h (oracle), tests/hostexec (HIP stage code).
#include "hlsl_resources.h"
#include "../zro_kat_layout.h"

namespace hlsl {
// cbFrameConstants, samplers, math, BSDF, RT ...
#include "ZetaRenderPass/Common/FrameConstants.h"
#include "ZetaRenderPass/Common/BSDFSampling.hlsli"
#include "ZetaRenderPass/Common/RT.hlsli"
#include "ZetaRenderPass/Common/GBuffers.hlsli"
}

using namespace hlsl;

static DescriptorHeap g_heap;
static inline float F(uint32_t u) { return zr_asfloat(u); }
static inline uint32_t U(float f) { return zr_asuint(f); }

extern "C" {

// rho.dds payload (R16_UNORM w x h x d) bound at descriptor-heap slot 0 (BSDF.hlsli:281)
void zrefh_bind_rho(const uint16_t* data, uint32_t w, uint32_t h, uint32_t d)
{
    g_heapPtr = &g_heap;
    TexStorage& s = g_heap.table[0];
    s.data = (void*)data; s.w = w; s.h = h; s.d = d; s.fmt = FMT_R16_UNORM;
}

// ---- ZR_KAT_SAMPLING: Sampling.hlsli:165-288 warps + RNG stream.  in: u0, u1, cosThetaMax, seed bits
void zrefh_kat_sampling(const float* in, float* out, uint32_t n)
{
    for (uint32_t i = 0; i < n; i++)
    {
        const float* p = in + ZR_KAT_SAMPLING_IN * i; float* o = out + ZR_KAT_SAMPLING_OUT * i;
        float2 u(p[0], p[1]);
        float pdf;
        float3 a = Sampling::UniformSampleHemisphere(u, pdf); o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = pdf;
        a = Sampling::SampleCosineWeightedHemisphere(u, pdf); o[4] = a.x; o[5] = a.y; o[6] = a.z; o[7] = pdf;
        a = Sampling::UniformSampleCone(u, p[2], pdf); o[8] = a.x; o[9] = a.y; o[10] = a.z; o[11] = pdf;
        float2 d = Sampling::UniformSampleDisk(u); o[12] = d.x; o[13] = d.y;
        d = Sampling::UniformSampleDiskConcentric(u); o[14] = d.x; o[15] = d.y;
        a = Sampling::UniformSampleSphere(u); o[16] = a.x; o[17] = a.y; o[18] = a.z;
        d = Sampling::UniformSampleTriangle(u); o[19] = d.x; o[20] = d.y;
        // RNG (Sampling.hlsli:12-159)
        const uint32_t seed = U(p[3]);
        RNG r = RNG::Init(uint2(seed & 0xfffu, (seed >> 12) & 0xfffu), seed >> 24);
        o[21] = F(r.State);
        o[22] = r.Uniform();
        o[23] = F(r.UniformUintBounded(1u + (seed % 1000u)));
        o[24] = F(r.UniformUintBounded_Faster(1u + (seed % 977u)));
        float2 u2 = r.Uniform2D(); o[25] = u2.x; o[26] = u2.y;
        RNG r2 = RNG::Init(uint2(seed & 0xfffu, (seed >> 12) & 0xfffu), seed >> 24, seed & 7u);
        o[27] = F(r2.State);
        RNG r3 = RNG::Init(seed, seed >> 24);
        o[28] = F(r3.State);
        uint3 h3 = RNG::PCG3d(uint3(seed, seed * 3u, seed ^ 0x9e3779b9u)); o[29] = F(h3.x); o[30] = F(h3.y); o[31] = F(h3.z);
    }
}

// ---- ZR_KAT_MATH: Math.hlsli packing + geometry helpers.  in: v(3) unit, q(4) unit quaternion, s(3) scale, t(3) translation, x, uv(2)
void zrefh_kat_math(const float* in, float* out, uint32_t n)
{
    for (uint32_t i = 0; i < n; i++)
    {
        const float* p = in + ZR_KAT_MATH_IN * i; float* o = out + ZR_KAT_MATH_OUT * i;
        float3 v(p[0], p[1], p[2]);
        float4 q(p[3], p[4], p[5], p[6]);
        float3 s(p[7], p[8], p[9]), t(p[10], p[11], p[12]);
        float x = p[13];
        float2 uv(p[14], p[15]);
        float2 e = Math::EncodeUnitVector(v); o[0] = e.x; o[1] = e.y;
        float3 dv = Math::DecodeUnitVector(e); o[2] = dv.x; o[3] = dv.y; o[4] = dv.z;
        uint16_t2 o32 = Math::EncodeOct32(v); o[5] = F(o32.x); o[6] = F(o32.y);
        float3 d32 = Math::DecodeOct32(o32); o[7] = d32.x; o[8] = d32.y; o[9] = d32.z;
        uint16_t2 un = Math::EncodeAsUNorm2(uv); o[10] = F(un.x); o[11] = F(un.y);
        float2 dun = Math::DecodeUNorm2(un); o[12] = dun.x; o[13] = dun.y;
        o[14] = F(Math::Float3ToRGB8(saturate(abs(v))));
        float3 rgb = Math::UnpackRGB8(U(p[16]) & 0xffffffu); o[15] = rgb.x; o[16] = rgb.y; o[17] = rgb.z;
        float3 rv = Math::RotateVector(v, q); o[18] = rv.x; o[19] = rv.y; o[20] = rv.z;
        float3 tr = Math::TransformTRS(v, t, q, s); o[21] = tr.x; o[22] = tr.y; o[23] = tr.z;
        float3 it = Math::InverseTransformTRS(tr, t, q, s); o[24] = it.x; o[25] = it.y; o[26] = it.z;
        Math::CoordinateSystem onb = Math::CoordinateSystem::Build(v);
        o[27] = onb.b1.x; o[28] = onb.b1.y; o[29] = onb.b1.z; o[30] = onb.b2.x; o[31] = onb.b2.y; o[32] = onb.b2.z;
        o[33] = Math::ArcCos(x);
        float2 sph = Math::SphericalFromCartesian(v); o[34] = sph.x; o[35] = sph.y;
        o[36] = Math::NextFloat32(x); o[37] = Math::PrevFloat32(x);
        float4 dq = Math::DecodeNormalized4(uint16_t4((uint16_t)(U(p[16]) & 0xffffu), (uint16_t)(U(p[16]) >> 16), (uint16_t)(U(p[17]) & 0xffffu), (uint16_t)(U(p[17]) >> 16)));
        o[38] = dq.x; o[39] = dq.y; o[40] = dq.z; o[41] = dq.w;
        float3 w2l = Math::WorldToTangentFrame(v, float3(p[18], p[19], p[20])); o[42] = w2l.x; o[43] = w2l.y; o[44] = w2l.z;
        float3 l2w = Math::FromTangentFrameToWorld(v, float3(p[18], p[19], p[20])); o[45] = l2w.x; o[46] = l2w.y; o[47] = l2w.z;
    }
}

// ---- ZR_KAT_RT: RT.hlsli helpers.  in: pos(3), normal(3) unit, wi(3) unit, pdfs(3)
void zrefh_kat_rt(const float* in, float* out, uint32_t n)
{
    for (uint32_t i = 0; i < n; i++)
    {
        const float* p = in + ZR_KAT_RT_IN * i; float* o = out + ZR_KAT_RT_OUT * i;
        float3 pos(p[0], p[1], p[2]), nrm(p[3], p[4], p[5]);
        float3 a = RT::OffsetRayRTG(pos, nrm); o[0] = a.x; o[1] = a.y; o[2] = a.z;
        a = RT::OffsetRayRTG(pos, -nrm); o[3] = a.x; o[4] = a.y; o[5] = a.z;
        o[6] = RT::BalanceHeuristic(p[9], p[10], p[11]);
        o[7] = RT::BalanceHeuristic3(p[9], p[10], p[11], p[9]);
        o[8] = RT::PowerHeuristic(p[9], p[10], p[11]);
        float3 d = RT::GeneratePinholeCameraRay_CS(uint2((uint32_t)(p[12] * 1920.0f), (uint32_t)(p[13] * 1080.0f)), float2(1920.0f, 1080.0f), 1920.0f / 1080.0f, p[14], float2(p[15], p[16]));
        o[9] = d.x; o[10] = d.y; o[11] = d.z;
    }
}

// ---- ZR_KAT_BSDF: ShadingData::Init + SetWi + Unified + SampleBSDF + BSDFSamplerPdf + EvalBSDFSampler (BSDF.hlsli, BSDFSampling.hlsli)
void zrefh_kat_bsdf(const float* in, float* out, uint32_t n)
{
    for (uint32_t i = 0; i < n; i++)
    {
        const float* p = in + ZR_KAT_BSDF_IN * i; float* o = out + ZR_KAT_BSDF_OUT * i;
        for (int k = 0; k < ZR_KAT_BSDF_OUT; k++) o[k] = 0.0f;
        float3 nrm(p[0], p[1], p[2]), wo(p[3], p[4], p[5]), wi(p[6], p[7], p[8]);
        const bool metallic = p[9] > 0.5f; const float roughness = p[10]; float3 base(p[11], p[12], p[13]);
        const bool specTr = p[14] > 0.5f; const float coat_w = p[15]; float3 coat_c(p[16], p[17], p[18]);
        const float coat_r = p[19], eta_coat = p[20], ior = p[21];
        const half trDepth = half(p[22]), subsurf = half(p[23]);
        const bool exiting = p[24] > 0.5f;
        const uint32_t seed = U(p[25]);
        const float eta_curr = exiting ? ior : ETA_AIR, eta_next = exiting ? ETA_AIR : ior;
        BSDF::ShadingData s = BSDF::ShadingData::Init(nrm, wo, metallic, roughness, base, eta_curr, eta_next, specTr, trDepth, subsurf, coat_w, coat_c, coat_r, eta_coat);
        o[0] = s.alpha; o[1] = s.eta; o[2] = s.g_wo; o[3] = s.coat_alpha; o[4] = s.coat_eta; o[5] = s.ndotwo;
        float3 wh = s.SetWi(wi, nrm);
        o[6] = wh.x; o[7] = wh.y; o[8] = wh.z;
        o[9] = s.ndotwi; o[10] = s.ndotwh; o[11] = s.whdotwo; o[12] = s.whdotwi; o[13] = s.wodotwi;
        o[14] = F((s.invalid ? 1u : 0u) | (s.reflection ? 2u : 0u) | (s.backfacing_wo ? 4u : 0u));
        BSDF::BSDFEval e = BSDF::Unified(s);
        o[15] = e.f.x; o[16] = e.f.y; o[17] = e.f.z; o[18] = e.Fr_g.x; o[19] = e.Fr_g.y; o[20] = e.Fr_g.z; o[21] = F(e.tir ? 1u : 0u);
        RNG rng = RNG::Init(seed);
        BSDF::BSDFSample bs = BSDF::SampleBSDF(nrm, s, rng);
        o[22] = bs.wi.x; o[23] = bs.wi.y; o[24] = bs.wi.z; o[25] = bs.pdf;
        o[26] = bs.bsdfOverPdf.x; o[27] = bs.bsdfOverPdf.y; o[28] = bs.bsdfOverPdf.z;
        o[29] = bs.f.x; o[30] = bs.f.y; o[31] = bs.f.z; o[32] = F((uint32_t)BSDF::LobeToValue(bs.lobe)); o[33] = F(rng.State);
        RNG rng2 = RNG::Init(seed ^ 0x5bd1e995u);
        o[34] = BSDF::BSDFSamplerPdf(nrm, s, wi, rng2); o[35] = F(rng2.State);
        // replay the sampler with the random numbers that produced bs (what the ReSTIR PT shifts do)
        RNG rng3 = RNG::Init(seed);
        BSDF::BSDFSamplerEval se = BSDF::EvalBSDFSampler(nrm, s, bs.wi, bs.lobe, rng3);
        o[36] = se.pdf; o[37] = se.bsdfOverPdf.x; o[38] = se.bsdfOverPdf.y; o[39] = se.bsdfOverPdf.z; o[40] = se.f.x; o[41] = se.f.y; o[42] = se.f.z;
        // the two restricted samplers
        RNG rng4 = RNG::Init(seed + 17u);
        BSDF::BSDFSample nd = BSDF::SampleBSDF_NoDiffuse(nrm, s, rng4);
        o[43] = nd.wi.x; o[44] = nd.wi.y; o[45] = nd.wi.z; o[46] = nd.pdf; o[47] = nd.bsdfOverPdf.x; o[48] = nd.bsdfOverPdf.y; o[49] = nd.bsdfOverPdf.z;
        o[50] = F((uint32_t)BSDF::LobeToValue(nd.lobe));
        o[51] = BSDF::BSDFSamplerPdf_NoDiffuse(nrm, s, wi);
        // microfacet building blocks on this configuration
        o[52] = BSDF::GGX(s.ndotwh, max(s.alpha, 1e-4f) * max(s.alpha, 1e-4f));
        o[53] = BSDF::SmithHeightCorrelatedG2OverG1(max(s.alpha, 1e-4f) * max(s.alpha, 1e-4f), s.ndotwi, s.ndotwo);
        o[54] = BSDF::GGXReflectance_Dielectric(max(s.alpha, 0.002025f), s.ndotwo, 1.0f / 1.5f);
        float3 rm = BSDF::GGXReflectance_Metal(base, max(s.alpha, 0.002025f), s.ndotwo); o[55] = rm.x; o[56] = rm.y; o[57] = rm.z;
        float2 uu(zr_asfloat((seed >> 9) | 0x3f800000u) - 1.0f, zr_asfloat(((seed * 747796405u) >> 9) | 0x3f800000u) - 1.0f);
        float3 whs = BSDF::SampleGGXMicrofacet(wo, s.alpha, nrm, uu); o[58] = whs.x; o[59] = whs.y; o[60] = whs.z;
    }
}

// ---- ZR_KAT_GBUFFER: GBuffers.hlsli encode / decode helpers
void zrefh_kat_gbuffer(const float* in, float* out, uint32_t n)
{
    for (uint32_t i = 0; i < n; i++)
    {
        const float* p = in + ZR_KAT_GBUFFER_IN * i; float* o = out + ZR_KAT_GBUFFER_OUT * i;
        const uint32_t bits = U(p[0]);
        const bool isMetal = bits & 1u, isTr = bits & 2u, isEm = bits & 4u, trDepthGt0 = bits & 8u, subs = bits & 16u, coated = bits & 32u;
        float enc = GBuffer::EncodeMetallic(isMetal, isTr, isEm, trDepthGt0, subs, coated);
        o[0] = enc;
        GBuffer::Flags fl = GBuffer::DecodeMetallic(enc);
        o[1] = F((fl.metallic ? 1u : 0u) | (fl.transmissive ? 2u : 0u) | (fl.emissive ? 4u : 0u) | (fl.invalid ? 8u : 0u) | (fl.trDepthGt0 ? 16u : 0u) |
                 (fl.subsurface ? 32u : 0u) | (fl.coated ? 64u : 0u));
        o[2] = GBuffer::EncodeIOR(p[1]);
        o[3] = GBuffer::DecodeIOR(o[2]);
    }
}

} // extern "C"
