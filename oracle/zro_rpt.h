// ORACLE -- test infrastructure only (see zro_math.h header).  PARITY UNPINNED against the reference (no executable
// reference exists for this path); follows the shaders line by line.
//
// zro_rpt.h: CPU restatement of ReSTIR PT (emissive-NEE variants, NEE_EMISSIVE == 1, no presampled sets):
//   K11 ReSTIR_PT/ReSTIR_PT_PathTrace.hlsl, ReSTIR_PT_NEE.hlsli, Reservoir.hlsli, Shift.hlsli
//   K13 ReSTIR_PT_Replay.hlsl (CtT, TtC, CtS, StC)      K14 ReSTIR_PT_Reconnect_CtT.hlsl / _TtC.hlsl
//   K15 ReSTIR_PT_SpatialSearch.hlsl + SampleSet.hlsli    K16 ReSTIR_PT_Reconnect_CtS.hlsl / _StC.hlsl
//   host order: IndirectLighting.cpp:877-1004 (RenderReSTIR_PT), :370-596 (Temporal), :598-875 (Spatial)
// K12 (sort) only permutes which thread handles which pixel; it does not change per-pixel results except through the
// wave-dependent boiling suppression, for which the ABI pins "wave" = the 8x8 pixel group.
#pragma once
#include "zro_scene.h"
#include "../include/zetaray_amd.h"

namespace zro {
namespace RPT {

using BSDF::LOBE;
using Light::TYPE;

static const float MAX_PLANE_DIST_REUSE = 1.0f;
static const float MAX_ROUGHNESS_DIFF_TEMPORAL_REUSE = 0.3f;
static const float MAX_ROUGHNESS_DIFF_SPATIAL_REUSE = 0.05f;
static const float MIN_NORMAL_SIMILARITY_SPATIAL_REUSE = 0.9f;
static const uint32_t M_MAX_X_K_TRANSMISSIVE = 4, M_MAX_X_K_IN_MOTION = 4;
static const int SPATIAL_NEIGHBOR_OFFSET = 32;
static const int SPATIAL_SEARCH_RADIUS = 15;

static inline float3 RoundHalf3(float3 v) { return f3(zr_round_f16(v.x), zr_round_f16(v.y), zr_round_f16(v.z)); }
static inline float Sanitize(float x) { return (zr_isnan(x) || zr_isinf(x)) ? 0.0f : x; }
static inline float3 Sanitize3(float3 v)
{ bool bad = any_nan(v) || zr_isinf(v.x) || zr_isinf(v.y) || zr_isinf(v.z); return bad ? f3(0.0f) : v; }

// BSDF.hlsli:864-895
static inline bool IsLobeValid(const BSDF::ShadingData& s, LOBE lt)
{
    if (lt == LOBE::ALL) return true;
    if (s.metallic && (lt != LOBE::GLOSSY_R) && (lt != LOBE::COAT)) return false;
    if (!s.specTr && (lt == LOBE::GLOSSY_T)) return false;
    if (s.specTr && (lt == LOBE::DIFFUSE_R)) return false;
    if (!s.ThinWalled() && (lt == LOBE::DIFFUSE_T)) return false;
    if (!s.Coated() && (lt == LOBE::COAT)) return false;
    return true;
}
static inline float LobeAlpha(const BSDF::ShadingData& s, LOBE lt)
{
    if (lt == LOBE::GLOSSY_R || lt == LOBE::GLOSSY_T) return s.alpha;
    if (lt == LOBE::COAT) return s.coat_alpha;
    return 1.0f;
}

// BSDFSampling.hlsli:340-428 (NoOp target)
static BSDF::BSDFSamplerEval EvalBSDFSampler_NoSpecTr(float3 normal, BSDF::ShadingData surface, float3 wi, LOBE lobe,
    float2 u_c, float2 u_g, float2 u_d)
{
    using namespace BSDF;
    BSDFSamplerEval ret;
    const float3 targetScale_z = f3(1.0f);
    float w_sum = 0;
    float3 target = f3(0.0f);
    if (surface.Coated())
    {
        const bool isZ_c = lobe == LOBE::COAT;
        const float3 wi_c = isZ_c ? wi : SampleCoat(surface, normal, u_c);
        surface.SetWi_Refl(wi_c, normal);
        target = Unified(surface).f * targetScale_z;
        const float targetLum_c = Math::Luminance(target);
        const float pdf_c = CoatPdf(surface);
        const float pdf_g = GlossPdf(surface);
        const float pdf_d = !surface.metallic ? DiffusePdf(surface) : 0;
        w_sum = RT::BalanceHeuristic3(pdf_c, pdf_g, pdf_d, targetLum_c);
    }
    {
        const bool isZ_g = lobe == LOBE::GLOSSY_R;
        const float3 wi_g = isZ_g ? wi : SampleGloss(surface, normal, u_g);
        surface.SetWi_Refl(wi_g, normal);
        const float3 target_g = Unified(surface).f * targetScale_z;
        const float targetLum_g = Math::Luminance(target_g);
        const float pdf_g = GlossPdf(surface);
        const float pdf_d = !surface.metallic ? DiffusePdf(surface) : 0;
        const float pdf_c = surface.Coated() ? CoatPdf(surface) : 0;
        w_sum += RT::BalanceHeuristic3(pdf_g, pdf_d, pdf_c, targetLum_g);
        target = isZ_g ? target_g : target;
    }
    if (!surface.metallic)
    {
        float unused;
        float3 w_d = SampleDiffuse(normal, u_d, unused);
        float Fr_g;
        {
            const bool isZ_dr = lobe == LOBE::DIFFUSE_R;
            const float3 wi_d = isZ_dr ? wi : w_d;
            surface.SetWi_Refl(wi_d, normal);
            BSDFEval eval = Unified(surface);
            const float3 target_dr = eval.f * targetScale_z;
            Fr_g = eval.Fr_g.x;
            const float targetLum_dr = Math::Luminance(target_dr);
            const float pdf_d = DiffusePdf(surface);
            const float pdf_g = GlossPdf(surface);
            const float pdf_c = surface.Coated() ? CoatPdf(surface) : 0;
            w_sum += RT::BalanceHeuristic3(pdf_d, pdf_g, pdf_c, targetLum_dr);
            target = isZ_dr ? target_dr : target;
        }
        if (surface.ThinWalled())
        {
            const bool isZ_dt = lobe == LOBE::DIFFUSE_T;
            const float3 target_dt = DielectricBaseDiffuseTr(surface, Fr_g) * targetScale_z;
            const float targetLum_dt = Math::Luminance(target_dt);
            const float pdf_d = DiffusePdf(surface);
            w_sum += targetLum_dt / pdf_d;
            target = isZ_dt ? target_dt : target;
        }
    }
    float targetLum = Math::Luminance(target);
    ret.bsdfOverPdf = targetLum > 0 ? target * w_sum / targetLum : f3(0.0f);
    ret.pdf = w_sum > 0 ? targetLum / w_sum : 0;
    ret.f = target;
    return ret;
}

// BSDFSampling.hlsli:430-502
template<typename Func>
static BSDF::BSDFSamplerEval EvalBSDFSampler_NoDiffuse(float3 normal, BSDF::ShadingData surface, float3 wi, LOBE lobe, Func func)
{
    using namespace BSDF;
    float3 wh = surface.SetWi(wi, normal);
    BSDFEval eval = Unified(surface);
    const float3 targetScale = func(wi);
    float pdf_base = 1;
    BSDFSamplerEval ret;
    ret.f = eval.f * targetScale;
    if (surface.Coated())
    {
        float reflectance_c = GGXReflectance_Dielectric(surface.coat_alpha, surface.ndotwo, surface.coat_eta);
        float pdf_coat = reflectance_c * surface.coat_weight;
        pdf_base = 1 - pdf_coat;
        if (lobe == LOBE::COAT)
        {
            ret.pdf = CoatPdf(surface) * pdf_coat;
            ret.bsdfOverPdf = ret.f / ret.pdf;
            return ret;
        }
    }
    const float wh_pdf = GGXMicrofacetPdf(surface.alpha, surface.ndotwh, surface.ndotwo);
    ret.pdf = !surface.GlossSpecular() ? wh_pdf / 4.0f : (surface.ndotwh >= MIN_N_DOT_H_SPECULAR ? 1.0f : 0.0f);
    ret.pdf *= pdf_base;
    ret.bsdfOverPdf = ret.f / ret.pdf;
    if (surface.metallic || !surface.specTr || eval.tir) return ret;
    const float3 wi_other = lobe == LOBE::GLOSSY_T ? reflect(-surface.wo, wh) : refract(-surface.wo, wh, 1 / surface.eta);
    float targetScaleLum = Math::Luminance(targetScale);
    float targetScaleOtherLum = Math::Luminance(func(wi_other));
    float p_r = eval.Fr_g.x * (lobe == LOBE::GLOSSY_R ? targetScaleLum : targetScaleOtherLum);
    p_r = p_r / (p_r + (1 - eval.Fr_g.x) * (lobe == LOBE::GLOSSY_R ? targetScaleOtherLum : targetScaleLum));
    if (lobe == LOBE::GLOSSY_R)
    {
        ret.bsdfOverPdf /= p_r;
        ret.pdf *= p_r;
        return ret;
    }
    ret.bsdfOverPdf = ((!surface.invalid ? 1.0f : 0.0f) * (!surface.reflection ? 1.0f : 0.0f)) * TranslucentTrOverPdf(surface, eval.Fr_g.x);
    ret.bsdfOverPdf *= TransmittanceToDielectricBaseTr(surface);
    ret.bsdfOverPdf *= targetScale;
    ret.bsdfOverPdf /= pdf_base;
    ret.bsdfOverPdf /= (1 - p_r);
    ret.pdf = 1 - p_r;
    ret.pdf *= surface.GlossSpecular() ? (surface.ndotwh >= MIN_N_DOT_H_SPECULAR ? 1.0f : 0.0f) : wh_pdf * surface.whdotwo;
    ret.pdf *= pdf_base;
    if (!surface.GlossSpecular())
    {
        float dwh_dwi = JacobianHalfVecToIncident_Tr(surface.eta, surface.whdotwo, surface.whdotwi);
        ret.pdf *= dwh_dwi;
    }
    return ret;
}

// BSDFSampling.hlsli:548-563
static BSDF::BSDFSamplerEval EvalBSDFSampler(float3 normal, const BSDF::ShadingData& surface, float3 wi, LOBE lobe, RNG& rng)
{
    float2 u_c = rng.Uniform2D();
    float2 u_g = rng.Uniform2D();
    float2 u_d = rng.Uniform2D();
    rng.Uniform(); rng.Uniform(); rng.Uniform();
    if (!surface.specTr) return EvalBSDFSampler_NoSpecTr(normal, surface, wi, lobe, u_c, u_g, u_d);
    return EvalBSDFSampler_NoDiffuse(normal, surface, wi, lobe, BSDF::NoOp());
}

// NEE.hlsli:28-73
struct DirectLightingEstimate
{
    float3 ld, le, wi, pos, normal; float pdf_solidAngle, dwdA; TYPE lt; LOBE lobe; uint32_t ID; float pdf_light; bool twoSided;
    static DirectLightingEstimate Init()
    {
        DirectLightingEstimate r;
        r.ld = f3(0.0f); r.le = f3(0.0f); r.wi = f3(0.0f); r.pdf_solidAngle = 0; r.dwdA = 1; r.lt = TYPE::NONE; r.ID = 0xffffffffu;
        r.pos = f3(0.0f); r.pdf_light = 0; r.twoSided = true; r.normal = f3(0.0f); r.lobe = LOBE::ALL;
        return r;
    }
};

// Shift.hlsli:16-172
struct Reconnection
{
    static const uint16_t EMPTY = 0xf;
    float3 x_k; uint32_t ID, meshIdx; float partialJacobian; float3 w_k_lightNormal_w_sky; float lightPdf;
    uint32_t seed_replay, seed_nee; float dwdA; float3 L; uint16_t k; LOBE lobe_k_min_1, lobe_k; TYPE lt_k, lt_k_plus_1; bool x_k_in_motion;

    static Reconnection Init()
    {
        Reconnection r;
        r.k = EMPTY; r.lt_k = TYPE::NONE; r.lt_k_plus_1 = TYPE::NONE; r.partialJacobian = 0; r.x_k = f3(ZR_FLT_MAX); r.seed_replay = 0;
        r.w_k_lightNormal_w_sky = f3(0.0f); r.L = f3(0.0f); r.lightPdf = 0; r.seed_nee = 0; r.dwdA = 0;
        // not initialised by the reference; pinned to 0 / ALL here
        r.ID = 0; r.meshIdx = 0; r.lobe_k_min_1 = LOBE::DIFFUSE_R; r.lobe_k = LOBE::DIFFUSE_R; r.x_k_in_motion = false;
        return r;
    }
    bool Empty() const { return k == EMPTY; }
    bool IsCase2() const { return lt_k_plus_1 != TYPE::NONE; }
    bool IsCase3() const { return lt_k != TYPE::NONE; }
    bool IsCase1() const { return !IsCase2() && !IsCase3(); }
    void Clear() { k = EMPTY; lt_k = TYPE::NONE; lt_k_plus_1 = TYPE::NONE; }

    void SetCase1(int k_, float3 x_k_, float t, float3 normal_k, uint32_t hitID, uint32_t meshIdx_, float3 w_k_min_1, LOBE l_k_min_1,
        float pdf_w_k_min_1, float3 w_k, LOBE l_k, float pdf_w_k)
    {
        lobe_k_min_1 = l_k_min_1; k = (uint16_t)k_; x_k = x_k_; ID = hitID; meshIdx = meshIdx_; lt_k = TYPE::NONE; lobe_k = l_k;
        w_k_lightNormal_w_sky = w_k; lt_k_plus_1 = TYPE::NONE;
        partialJacobian = pdf_w_k_min_1;
        float cos_theta_k = zr_abs(dot(-w_k_min_1, normal_k));
        partialJacobian *= cos_theta_k / (t * t);
        partialJacobian *= pdf_w_k;
    }
    void SetCase2(int k_, float3 x_k_, float t, float3 normal_k, uint32_t hitID, uint32_t meshIdx_, float3 w_k_min_1, LOBE l_k_min_1,
        float pdf_w_k_min_1, float3 w_k, LOBE l_k, float pdf_w_k, TYPE t_k_plus_1, float pdf_light, float3 le, uint32_t seed, float dwdA_)
    {
        lobe_k_min_1 = l_k_min_1; k = (uint16_t)k_; x_k = x_k_; ID = hitID; meshIdx = meshIdx_; lt_k = TYPE::NONE; lobe_k = l_k;
        w_k_lightNormal_w_sky = w_k; lt_k_plus_1 = t_k_plus_1; lightPdf = pdf_light; dwdA = dwdA_; seed_nee = seed; L = RoundHalf3(le);
        partialJacobian = pdf_w_k_min_1;
        float cos_theta_k = zr_abs(dot(-w_k_min_1, normal_k));
        partialJacobian *= cos_theta_k / (t * t);
        if (lobe_k != LOBE::ALL) partialJacobian *= pdf_w_k;
    }
    void SetCase3(int k_, float3 x_k_, TYPE t, LOBE l_k_min_1, uint32_t lightID, float3 le, float3 lightNormal, float pdf_solidAngle,
        float pdf_light, float dwdA_, float3 w_sky, bool twoSided, uint32_t seed)
    {
        lobe_k_min_1 = l_k_min_1; k = (uint16_t)k_; x_k = x_k_; ID = lightID; lt_k = t; seed_nee = seed;
        partialJacobian = l_k_min_1 == LOBE::ALL ? 1.0f : pdf_solidAngle * dwdA_;
        lightPdf = twoSided ? pdf_light : -pdf_light;
        L = RoundHalf3(le);
        lt_k_plus_1 = TYPE::NONE;
        if (t == TYPE::EMISSIVE) w_k_lightNormal_w_sky = lightNormal;
        else if (t == TYPE::SKY) w_k_lightNormal_w_sky = w_sky;
    }
};

// Shift.hlsli:360-375
static inline bool CanReconnect(float alpha_lobe_k_min_1, float alpha_lobe_k, LOBE lobe_k_min_1, LOBE lobe_k, float alpha_min)
{
    if ((alpha_lobe_k_min_1 < alpha_min) || (alpha_lobe_k < alpha_min)) return false;
    if ((lobe_k_min_1 == LOBE::GLOSSY_T) && (lobe_k == LOBE::GLOSSY_T)) return false;
    return true;
}

// ---- reservoir storage: the reference's 7 planes (IndirectLighting.h:128-144, Reservoir.hlsli:267-456)
struct ReservoirPlanes
{
    std::vector<uint32_t> A;        // RGBA8_UINT: x | y << 8 | z << 16
    std::vector<float> B;           // RG32F (w_sum, W)
    std::vector<uint32_t> C, D;     // RGBA32_UINT
    std::vector<uint16_t> E;        // R16F
    std::vector<float> F;           // RG32F
    std::vector<uint32_t> G;        // RG32_UINT
    void Resize(size_t n) { A.assign(n, 0); B.assign(2 * n, 0); C.assign(4 * n, 0); D.assign(4 * n, 0); E.assign(n, 0); F.assign(2 * n, 0); G.assign(2 * n, 0); }
};

struct Reservoir
{
    float w_sum, W; float3 target; Reconnection rc; uint16_t M;
    static Reservoir Init() { Reservoir r; r.rc = Reconnection::Init(); r.w_sum = 0; r.W = 0; r.M = 0; r.target = f3(0.0f); return r; }

    // Reservoir.hlsli:23-45
    bool Update(float weight, float3 target_, const Reconnection& rc_, RNG& rng)
    {
        if (zr_isnan(weight) || zr_isinf(weight)) return false;
        M += 1;
        if (weight == 0) return false;
        w_sum += weight;
        if (rng.Uniform() < (weight / w_sum)) { rc = rc_; target = target_; return true; }
        return false;
    }
    void UnpackMetadata(uint32_t a)
    {
        uint32_t mx = a & 0xff, my = (a >> 8) & 0xff, mz = (a >> 16) & 0xff;
        uint16_t kk = (uint16_t)(mx & 0xf);
        rc.k = kk == Reconnection::EMPTY ? kk : (uint16_t)(kk + 2);
        rc.lobe_k_min_1 = (LOBE)std::min<uint32_t>(my & 0x7, 5);
        rc.lobe_k = (LOBE)std::min<uint32_t>((my >> 3) & 0x7, 5);
        rc.lt_k = (TYPE)((my >> 6) & 0x3);
        rc.lt_k_plus_1 = (TYPE)(mz & 0x3);
        rc.x_k_in_motion = (mz >> 2) != 0;
        M = (uint16_t)(mx >> 4);
    }
    static Reservoir Load_Metadata(const ReservoirPlanes& p, size_t i)
    { Reservoir r = Init(); r.UnpackMetadata(p.A[i]); return r; }
    static Reservoir Load_NonReconnection(const ReservoirPlanes& p, size_t i)
    { Reservoir r = Init(); r.UnpackMetadata(p.A[i]); r.w_sum = p.B[2 * i]; r.W = p.B[2 * i + 1]; return r; }
    // LoadCase1/2/3<Emissive> (Reservoir.hlsli:47-139)
    void Load_Reconnection(const ReservoirPlanes& p, size_t i, bool emissive = true)
    {
        const uint32_t* inC = &p.C[4 * i]; const uint32_t* inD = &p.D[4 * i];
        auto oct = [](uint32_t e) { uint16_t v[2] = {(uint16_t)(e & 0xffff), (uint16_t)(e >> 16)}; return Math::DecodeOct32(v); };
        if (rc.IsCase1())
        {
            rc.partialJacobian = zr_asfloat(inC[0]); rc.seed_replay = inC[1]; rc.ID = inC[2];
            rc.w_k_lightNormal_w_sky = oct(inD[2]);
            rc.x_k = f3(zr_asfloat(inC[3]), zr_asfloat(inD[0]), zr_asfloat(inD[1]));
            rc.meshIdx = p.G[2 * i + 1];
            rc.L = f3(zr_f16_to_f32(inD[3] & 0xffff), zr_f16_to_f32(inD[3] >> 16), zr_f16_to_f32(p.E[i]));
        }
        else if (rc.IsCase2() && !emissive)
        {
            rc.partialJacobian = zr_asfloat(inC[0]); rc.seed_replay = inC[1]; rc.ID = inC[2];
            rc.x_k = f3(zr_asfloat(inC[3]), zr_asfloat(inD[0]), zr_asfloat(inD[1]));
            if (rc.lt_k_plus_1 == TYPE::SKY) { rc.w_k_lightNormal_w_sky = oct(inD[2]); rc.seed_nee = inD[3]; }
            rc.meshIdx = p.G[2 * i + 1];
        }
        else if (!rc.IsCase2() && !emissive)
        {
            rc.seed_replay = inC[1]; rc.ID = inC[2];
            rc.partialJacobian = zr_asfloat(inC[0]);
            if (rc.lt_k == TYPE::SKY) { rc.w_k_lightNormal_w_sky = oct(inD[2]); rc.seed_nee = inD[3]; }
        }
        else if (rc.IsCase2())
        {
            rc.partialJacobian = zr_asfloat(inC[0]); rc.seed_replay = inC[1]; rc.ID = inC[2];
            rc.x_k = f3(zr_asfloat(inC[3]), zr_asfloat(inD[0]), zr_asfloat(inD[1]));
            rc.L = f3(zr_f16_to_f32(inD[3] & 0xffff), zr_f16_to_f32(inD[3] >> 16), zr_f16_to_f32(p.E[i]));
            rc.w_k_lightNormal_w_sky = oct(inD[2]);
            rc.lightPdf = p.F[2 * i]; rc.dwdA = p.F[2 * i + 1];
            rc.seed_nee = p.G[2 * i]; rc.meshIdx = p.G[2 * i + 1];
        }
        else
        {
            rc.seed_replay = inC[1]; rc.ID = inC[2];
            rc.partialJacobian = rc.lobe_k_min_1 == LOBE::ALL ? 1.0f : zr_asfloat(inC[0]);
            rc.x_k = f3(zr_asfloat(inC[3]), zr_asfloat(inD[0]), zr_asfloat(inD[1]));
            rc.L = f3(zr_f16_to_f32(inD[3] & 0xffff), zr_f16_to_f32(inD[3] >> 16), zr_f16_to_f32(p.E[i]));
            rc.lightPdf = p.F[2 * i];
            rc.seed_nee = inC[0];
            rc.w_k_lightNormal_w_sky = oct(inD[2]);
        }
    }
    static Reservoir Load(const ReservoirPlanes& p, size_t i, bool emissive = true)
    {
        Reservoir r = Load_NonReconnection(p, i);
        if (r.rc.Empty()) return r;
        r.Load_Reconnection(p, i, emissive);
        return r;
    }
    static uint32_t PackA_x(const Reconnection& rc, uint32_t m)
    { uint32_t k = rc.Empty() ? rc.k : (uint32_t)std::max<int>(rc.k, 2) - 2; return k | (m << 4); }
    void WriteReservoirData(ReservoirPlanes& p, size_t i, uint32_t M_max) const
    {
        uint32_t m = std::min<uint32_t>(M, M_max);
        p.A[i] = (p.A[i] & 0xffffff00u) | (PackA_x(rc, m) & 0xff);
        p.B[2 * i] = w_sum; p.B[2 * i + 1] = W;
    }
    void WriteReservoirData2(ReservoirPlanes& p, size_t i, uint32_t M_max) const
    {
        uint32_t m = std::min<uint32_t>(M, M_max);
        p.A[i] = (p.A[i] & 0xffffff00u) | (PackA_x(rc, m) & 0xff);
        p.B[2 * i + 1] = W;
    }
    // Write<Emissive>, Reservoir.hlsli:283-330, 367-456 (the non-emissive variant writes only some components of C / D / G)
    void Write(ReservoirPlanes& p, size_t i, uint32_t M_max = 0, bool emissive = true)
    {
        uint32_t m = M_max == 0 ? M : std::min<uint32_t>(M, M_max);
        uint32_t mx = PackA_x(rc, m) & 0xff;
        uint32_t my = ((uint32_t)rc.lobe_k_min_1 | ((uint32_t)rc.lobe_k << 3) | ((uint32_t)rc.lt_k << 6)) & 0xff;
        uint32_t mz = ((uint32_t)rc.lt_k_plus_1 | ((uint32_t)rc.x_k_in_motion << 2)) & 0xff;
        p.A[i] = (p.A[i] & 0xff000000u) | mx | (my << 8) | (mz << 16);
        w_sum = Sanitize(w_sum); W = Sanitize(W);
        p.B[2 * i] = w_sum; p.B[2 * i + 1] = W;
        if (rc.Empty()) return;
        uint16_t e[2]; Math::EncodeOct32(rc.w_k_lightNormal_w_sky, e);
        uint32_t w_k_encoded = e[0] | ((uint32_t)e[1] << 16);
        uint32_t lh = (uint32_t)zr_f32_to_f16(rc.L.x) | ((uint32_t)zr_f32_to_f16(rc.L.y) << 16);
        uint32_t* C = &p.C[4 * i]; uint32_t* D = &p.D[4 * i];
        if (rc.IsCase1())
        {
            C[0] = zr_asuint(rc.partialJacobian); C[1] = rc.seed_replay; C[2] = rc.ID; C[3] = zr_asuint(rc.x_k.x);
            D[0] = zr_asuint(rc.x_k.y); D[1] = zr_asuint(rc.x_k.z); D[2] = w_k_encoded; D[3] = lh;
            p.E[i] = zr_f32_to_f16(rc.L.z);
            p.G[2 * i + 1] = rc.meshIdx;
        }
        else if (rc.IsCase2() && !emissive)
        {
            C[0] = zr_asuint(rc.partialJacobian); C[1] = rc.seed_replay; C[2] = rc.ID; C[3] = zr_asuint(rc.x_k.x);
            if (rc.lt_k_plus_1 == TYPE::SKY) { D[0] = zr_asuint(rc.x_k.y); D[1] = zr_asuint(rc.x_k.z); D[2] = w_k_encoded; D[3] = rc.seed_nee; }
            else { D[0] = zr_asuint(rc.x_k.y); D[1] = zr_asuint(rc.x_k.z); }
            p.G[2 * i + 1] = rc.meshIdx;
        }
        else if (!rc.IsCase2() && !emissive)
        {
            C[0] = zr_asuint(rc.partialJacobian); C[1] = rc.seed_replay; C[2] = rc.ID;
            if (rc.lt_k == TYPE::SKY) { D[2] = w_k_encoded; D[3] = rc.seed_nee; }
        }
        else if (rc.IsCase2())
        {
            C[0] = zr_asuint(rc.partialJacobian); C[1] = rc.seed_replay; C[2] = rc.ID; C[3] = zr_asuint(rc.x_k.x);
            D[0] = zr_asuint(rc.x_k.y); D[1] = zr_asuint(rc.x_k.z); D[2] = w_k_encoded; D[3] = lh;
            p.E[i] = zr_f32_to_f16(rc.L.z);
            p.F[2 * i] = rc.lightPdf; p.F[2 * i + 1] = rc.dwdA;
            p.G[2 * i] = rc.seed_nee; p.G[2 * i + 1] = rc.meshIdx;
        }
        else
        {
            uint32_t v = rc.lobe_k_min_1 == LOBE::ALL ? rc.seed_nee : zr_asuint(rc.partialJacobian);
            C[0] = v; C[1] = rc.seed_replay; C[2] = rc.ID; C[3] = zr_asuint(rc.x_k.x);
            D[0] = zr_asuint(rc.x_k.y); D[1] = zr_asuint(rc.x_k.z); D[2] = w_k_encoded; D[3] = lh;
            p.E[i] = zr_f32_to_f16(rc.L.z);
            p.F[2 * i] = rc.lightPdf;
        }
    }
};

// canonical 64-lane wave sum of the ABI: xor butterfly with strides 1, 2, ..., 32; lanes that
// do not take part contribute +0.0f
static inline float WaveSum64(const float* in)
{
    float v[64], t[64];
    for (int i = 0; i < 64; i++) v[i] = in[i];
    for (int s = 1; s < 64; s <<= 1)
    {
        for (int i = 0; i < 64; i++) t[i] = v[i] + v[i ^ s];
        for (int i = 0; i < 64; i++) v[i] = t[i];
    }
    return v[0];
}

struct Globals { const Scene* sc; uint32_t numEmissives; int maxNumBounces; float alpha_min; bool presampled = false; uint32_t sampleSetIdx = 0;
    const zr_frame_constants* frame = nullptr; };      // numEmissives == 0 selects the NEE_EMISSIVE == 0 shader variants (sun + sky)

struct SkyFunc { const SkyLUT* lut; float3 operator()(float3 w) const { return Light::Le_Sky(w, *lut); } };

// ---- ReSTIR_PT_NEE.hlsli:134-207
static DirectLightingEstimate NEE_Bsdf(const Globals& g, float3 pos, float3 normal, const BSDF::ShadingData& surface, int nextBounce,
    BSDF::BSDFSample& bsdfSample, RtRayQuery::Hit_Emissive& hitInfo, RNG& rng)
{
    DirectLightingEstimate ret = DirectLightingEstimate::Init();
    const bool specular = surface.GlossSpecular() && (surface.metallic || surface.specTr) && (!surface.Coated() || surface.CoatSpecular());
    const int numLightSamples = specular ? 0 : 1;
    // `out` parameter the reference leaves unset when nextBounce > maxNumBounces (cannot happen: nextBounce <= maxNumBounces - 1 + 1)
    bsdfSample = BSDF::BSDFSample::Init();
    if (nextBounce <= g.maxNumBounces) bsdfSample = BSDF::SampleBSDF(normal, surface, rng);
    const float wiPdf = bsdfSample.pdf;
    const float3 wi = bsdfSample.wi;
    const float3 f = bsdfSample.f;
    hitInfo = RtRayQuery::Hit_Emissive::FindClosest(*g.sc, pos, normal, wi, surface.Transmissive());
    if (hitInfo.HitWasEmissive())
    {
        EmTri emissive; emissive.t = g.sc->emissives[hitInfo.emissiveTriIdx];
        const float3 le = Light::Le_EmissiveTriangle(*g.sc, emissive, hitInfo.bary);
        const float3 vtx0 = emissive.Vtx0(), vtx1 = emissive.V1(), vtx2 = emissive.V2();
        float3 lightNormal = cross(vtx1 - vtx0, vtx2 - vtx0);
        float twoArea = length(lightNormal);
        lightNormal = dot(lightNormal, lightNormal) == 0 ? f3(0.0f) : lightNormal / twoArea;
        lightNormal = emissive.IsDoubleSided() && (dot(-wi, lightNormal) < 0) ? -lightNormal : lightNormal;
        float lightPdf = 0;
        if (!specular)
        {
            const float lightSourcePdf = numLightSamples > 0 ? g.sc->alias[hitInfo.emissiveTriIdx].cached_p_orig : 0;
            lightPdf = twoArea > 0 ? lightSourcePdf * (2.0f / twoArea) : 0;
        }
        float dwdA = zr_saturate(dot(lightNormal, -wi)) / (hitInfo.t * hitInfo.t);
        float wiPdf_area = wiPdf * dwdA;
        float3 ld = le * f * dwdA;
        ret.ld = specular ? (wiPdf_area > 0 ? ld / wiPdf_area : f3(0.0f)) : RT::PowerHeuristic(wiPdf_area, lightPdf, ld);
        ret.le = le; ret.wi = wi; ret.pdf_solidAngle = wiPdf; ret.dwdA = dwdA; ret.ID = emissive.t.id;
        ret.pos = mad3(hitInfo.t, wi, pos); ret.normal = lightNormal; ret.pdf_light = lightPdf; ret.lobe = bsdfSample.lobe;
        ret.lt = TYPE::EMISSIVE; ret.twoSided = emissive.IsDoubleSided();
    }
    if (nextBounce >= g.maxNumBounces) bsdfSample.bsdfOverPdf = f3(0.0f);
    return ret;
}

// ---- ReSTIR_PT_NEE.hlsli:209-284 (alias-table branch); APPROXIMATE_EMISSIVE_SHADOW_RAY == 1 for ReSTIR PT
static DirectLightingEstimate NEE_Emissive(const Globals& g, float3 pos, float3 normal, BSDF::ShadingData surface, RNG& rng)
{
    DirectLightingEstimate ret = DirectLightingEstimate::Init();
    ret.lt = TYPE::EMISSIVE; ret.lobe = LOBE::ALL;
    Light::EmissiveTriSample lightSample; float3 le; float lightPdf; uint32_t lightID; bool twoSided;
    if (g.presampled)       // USE_PRESAMPLED_SETS, ReSTIR_PT_NEE.hlsli:217-236
    {
        Light::PresampledLight pl = Light::SamplePresampledSet(*g.sc, g.sampleSetIdx, pos, rng);
        lightSample.pos = pl.pos; lightSample.normal = pl.normal; le = pl.le; lightPdf = pl.pdf; lightID = pl.ID; twoSided = pl.twoSided;
        rng.Uniform3D();    // "deterministic RNG state regardless of USE_PRESAMPLED_SETS"
    }
    else
    {
        Light::AliasTableSample entry = Light::AliasTableSample::get(*g.sc, g.numEmissives, rng);
        EmTri tri; tri.t = g.sc->emissives[entry.idx];
        lightSample = Light::EmissiveTriSample::get(pos, tri, rng);
        le = Light::Le_EmissiveTriangle(*g.sc, tri, lightSample.bary);
        lightPdf = entry.pdf * lightSample.pdf;
        lightID = tri.t.id;
        twoSided = tri.IsDoubleSided();
    }
    const float t = length(lightSample.pos - pos);
    const float3 wi = (lightSample.pos - pos) / t;
    if ((dot(lightSample.normal, -wi) > 0) && (t > 0))
    {
        const float dwdA = zr_saturate(dot(lightSample.normal, -wi)) / (t * t);
        surface.SetWi(wi, normal);
        float3 ld = le * BSDF::Unified(surface).f * dwdA;
        if (dot(ld, ld) > 0)
            ld *= RtRayQuery::Visibility_Segment(*g.sc, true, pos, wi, t, normal, lightID, surface.Transmissive()) ? 1.0f : 0.0f;
        float bsdfPdf = 0;
        if (dot(ld, ld) > 0)
        {
            bsdfPdf = BSDF::BSDFSamplerPdf(normal, surface, wi, BSDF::NoOp(), rng);
            bsdfPdf *= dwdA;
        }
        ret.ld = RT::PowerHeuristic(lightPdf, bsdfPdf, ld);
        ret.le = le; ret.wi = wi; ret.pdf_solidAngle = lightPdf / dwdA; ret.dwdA = dwdA; ret.ID = lightID;
        ret.pos = lightSample.pos; ret.normal = lightSample.normal; ret.pdf_light = lightPdf; ret.twoSided = twoSided;
    }
    return ret;
}

// ---- ReSTIR_PT_NEE.hlsli:306-391
static DirectLightingEstimate EvalDirect_Emissive_Case2(float3 normal, BSDF::ShadingData surface, float3 wi, float3 le, float dwdA,
    float lightPdf, LOBE lobe, RNG& rngReplay, RNG& rngNEE)
{
    surface.SetWi(wi, normal);
    float3 ld = le * BSDF::Unified(surface).f * dwdA;
    DirectLightingEstimate ret = DirectLightingEstimate::Init();
    if (dot(ld, ld) == 0) return ret;
    if (lobe == LOBE::ALL)
    {
        rngNEE.Uniform(); rngNEE.Uniform(); rngNEE.Uniform(); rngNEE.Uniform();
        float bsdfPdf = BSDF::BSDFSamplerPdf(normal, surface, wi, BSDF::NoOp(), rngNEE);
        float bsdfPdf_area = bsdfPdf * dwdA;
        ret.ld = RT::PowerHeuristic(lightPdf, bsdfPdf_area, ld);
        ret.pdf_solidAngle = 1.0f;
    }
    else
    {
        BSDF::BSDFSamplerEval eval = EvalBSDFSampler(normal, surface, wi, lobe, rngReplay);
        const bool specular = surface.GlossSpecular() && (surface.metallic || surface.specTr) && (!surface.Coated() || surface.CoatSpecular());
        float bsdfPdf_area = eval.pdf * dwdA;
        ret.ld = specular ? (bsdfPdf_area > 0 ? ld / bsdfPdf_area : f3(0.0f)) : RT::PowerHeuristic(bsdfPdf_area, lightPdf, ld);
        ret.pdf_solidAngle = eval.pdf;
    }
    return ret;
}
static DirectLightingEstimate EvalDirect_Emissive_Case3(const Globals& g, float3 pos, float3 normal, BSDF::ShadingData surface, float3 wi, float t,
    float3 le, float3 lightNormal, float lightPdf, uint32_t lightID, bool twoSided, LOBE lobe, RNG& rngReplay, RNG& rngNEE)
{
    float wiDotLightNormal = dot(lightNormal, -wi);
    float dwdA = zr_abs(wiDotLightNormal) / (t * t);
    surface.SetWi(wi, normal);
    float3 ld = (wiDotLightNormal > 0) || twoSided ? le * BSDF::Unified(surface).f * dwdA : f3(0.0f);
    if (dot(ld, ld) > 0)
        ld *= RtRayQuery::Visibility_Segment(*g.sc, true, pos, wi, t, normal, lightID, surface.Transmissive()) ? 1.0f : 0.0f;
    DirectLightingEstimate ret = DirectLightingEstimate::Init();
    if (dot(ld, ld) == 0) return ret;
    if (lobe == LOBE::ALL)
    {
        rngNEE.Uniform(); rngNEE.Uniform(); rngNEE.Uniform(); rngNEE.Uniform();
        float bsdfPdf = BSDF::BSDFSamplerPdf(normal, surface, wi, BSDF::NoOp(), rngNEE);
        float bsdfPdf_area = bsdfPdf * dwdA;
        ret.ld = RT::PowerHeuristic(lightPdf, bsdfPdf_area, ld);
        ret.pdf_solidAngle = 1.0f;
    }
    else
    {
        BSDF::BSDFSamplerEval eval = EvalBSDFSampler(normal, surface, wi, lobe, rngReplay);
        const bool specular = surface.GlossSpecular() && (surface.metallic || surface.specTr) && (!surface.Coated() || surface.CoatSpecular());
        float bsdfPdf_area = eval.pdf * dwdA;
        ret.ld = specular ? (bsdfPdf_area > 0 ? ld / bsdfPdf_area : f3(0.0f)) : RT::PowerHeuristic(bsdfPdf_area, lightPdf, ld);
        ret.pdf_solidAngle = bsdfPdf_area;
    }
    return ret;
}

// ---- ReSTIR_PT_PathTrace.hlsl:36-192
// RPT_Util::NEE_NonEmissive, ReSTIR_PT_NEE.hlsli:10-132 (SKY_SAMPLING_PREFER_PERFORMANCE == 1)
static DirectLightingEstimate NEE_NonEmissive(const Globals& g, float3 pos, float3 normal, BSDF::ShadingData surface, RNG& rng)
{
    const Scene& sc = *g.sc; const zr_frame_constants& fr = *g.frame;
    DirectLightingEstimate ret = DirectLightingEstimate::Init();
    ret.dwdA = 1;
    SkyFunc leFunc; leFunc.lut = &sc.sky;
    const bool specular = surface.GlossSpecular() && (surface.metallic || surface.specTr) && (!surface.Coated() || surface.CoatSpecular());
    float w_sum = 0;
    float3 target_z = f3(0.0f);
    const float2 u_wrs = rng.Uniform2D();
    const float2 u_d = rng.Uniform2D();
    const float2 u_c = rng.Uniform2D();
    const float2 u_g = rng.Uniform2D();
    const float u_wrs_b0 = rng.Uniform();
    const float u_wrs_b1 = rng.Uniform();
    {
        const float3 wi_s = -f3(fr.sun_dir);
        const bool visible = (wi_s.y > 0) && ((dot(wi_s, normal) > 0) || surface.Transmissive());
        float pdf_b = 0, pdf_d = 0;
        if (visible)
        {
            surface.SetWi(wi_s, normal);
            target_z = Light::Le_Sun(pos, fr) * BSDF::Unified(surface).f;
            float ndotWi = dot(wi_s, normal);
            pdf_b = (ndotWi < 0) && surface.ThinWalled() ? 0 : BSDF::BSDFSamplerPdf_NoDiffuse(normal, surface, wi_s, leFunc);
            pdf_d = (!specular ? 1.0f : 0.0f) * zr_abs(ndotWi) * ZR_ONE_OVER_PI;
            pdf_d *= surface.ThinWalled() ? 0.5f : (ndotWi > 0 ? 1.0f : 0.0f);
        }
        w_sum = RT::BalanceHeuristic3(1, pdf_b, pdf_d, Math::Luminance(target_z));
        ret.lt = TYPE::SUN; ret.lobe = LOBE::ALL; ret.wi = wi_s;
    }
    if (!specular)
    {
        float pdf_e;
        float3 wi_e = BSDF::SampleDiffuse(normal, u_d, pdf_e);
        if (surface.ThinWalled()) { wi_e = u_wrs_b1 > 0.5f ? -wi_e : wi_e; pdf_e *= 0.5f; }
        surface.SetWi(wi_e, normal);
        const float3 target = leFunc(wi_e) * BSDF::Unified(surface).f;
        const float pdf_b = !surface.reflection && surface.ThinWalled() ? 0 : BSDF::BSDFSamplerPdf_NoDiffuse(normal, surface, wi_e, leFunc);
        const float denom = pdf_e + pdf_b;
        const float w_e = denom == 0 ? 0.0f : Math::Luminance(target) / denom;
        w_sum += w_e;
        if ((w_sum > 0) && (u_wrs.y < (w_e / w_sum))) { ret.lt = TYPE::SKY; ret.lobe = LOBE::ALL; ret.wi = wi_e; target_z = target; }
    }
    {
        BSDF::BSDFSample bsdfSample = BSDF::SampleBSDF_NoDiffuse(normal, surface, u_c, u_g, u_wrs_b0, u_wrs_b1, leFunc);
        float ndotwi = dot(bsdfSample.wi, normal);
        float pdf_e = (!specular ? 1.0f : 0.0f) * zr_abs(ndotwi) * ZR_ONE_OVER_PI;
        pdf_e *= surface.ThinWalled() ? 0.5f : (ndotwi > 0 ? 1.0f : 0.0f);
        const float denom = bsdfSample.pdf + pdf_e;
        const float w_b = denom == 0 ? 0.0f : Math::Luminance(bsdfSample.f) / denom;
        w_sum += w_b;
        if ((w_sum > 0) && (u_wrs.x < (w_b / w_sum))) { ret.lt = TYPE::SKY; ret.lobe = bsdfSample.lobe; ret.wi = bsdfSample.wi; target_z = bsdfSample.f; }
    }
    const float targetLum = Math::Luminance(target_z);
    ret.ld = targetLum > 0 ? target_z * w_sum / targetLum : f3(0.0f);
    ret.pdf_solidAngle = w_sum > 0 ? targetLum / w_sum : 0;
    if (dot(ret.ld, ret.ld) > 0) ret.ld *= RtRayQuery::Visibility_Ray(sc, pos, ret.wi, normal, surface.Transmissive()) ? 1.0f : 0.0f;
    return ret;
}

struct PrevHit { float alpha_lobe; float3 wi; float pdf; LOBE lobe; };

static void MaybeSetCase2OrCase3(const Globals& g, int pathVertex, float3 pos, float3 normal, float t, uint32_t ID, uint32_t meshIdx,
    const BSDF::ShadingData& surface, const PrevHit& prevHit, const DirectLightingEstimate& ls, uint32_t seed_nee, Reconnection& rc)
{
    const float alpha_lobe_direct = LobeAlpha(surface, ls.lobe);
    if (rc.Empty() && CanReconnect(prevHit.alpha_lobe, alpha_lobe_direct, prevHit.lobe, ls.lobe, g.alpha_min))
        rc.SetCase2(pathVertex, pos, t, normal, ID, meshIdx, prevHit.wi, prevHit.lobe, prevHit.pdf, ls.wi, ls.lobe, ls.pdf_solidAngle, ls.lt,
            ls.pdf_light, ls.le, seed_nee, ls.dwdA);
    if (rc.Empty() && (alpha_lobe_direct >= g.alpha_min))
        rc.SetCase3(pathVertex + 1, ls.pos, ls.lt, ls.lobe, ls.ID, ls.le, ls.normal, ls.pdf_solidAngle, ls.pdf_light, ls.dwdA, ls.wi, ls.twoSided, seed_nee);
}

static void EstimateDirectAndUpdateRC(const Globals& g, int pathVertex, float3 pos, const RtRayQuery::Hit& hitInfo, const BSDF::ShadingData& surface,
    const PrevHit& prevHit, float3 throughput, float3 throughput_k, float3& li, BSDF::BSDFSample& bsdfSample, RtRayQuery::Hit_Emissive& nextHit,
    Reconnection& rc, Reservoir& r, RNG& rngNEE, RNG& rngReplay)
{
    if (g.numEmissives == 0)      // EstimateDirectAndUpdateRC<false>, ReSTIR_PT_PathTrace.hlsl:172-191
    {
        const uint32_t seed_nee = rngNEE.State;
        DirectLightingEstimate ls = NEE_NonEmissive(g, pos, hitInfo.normal, surface, rngNEE);
        const float3 fOverPdf = throughput * ls.ld;
        li += fOverPdf;
        rc.L = RoundHalf3(ls.ld * throughput_k);
        MaybeSetCase2OrCase3(g, pathVertex, pos, hitInfo.normal, hitInfo.t, hitInfo.ID, hitInfo.meshIdx, surface, prevHit, ls, seed_nee, rc);
        float risWeight = Math::Luminance(fOverPdf);
        r.Update(risWeight, fOverPdf, rc, rngNEE);
        return;
    }
    BSDF::BSDFSample nextBsdfSample;
    int nextBounce = pathVertex - 1;
    DirectLightingEstimate ls_b = NEE_Bsdf(g, pos, hitInfo.normal, surface, nextBounce, nextBsdfSample, nextHit, rngReplay);
    if (nextHit.HitWasEmissive())
    {
        const float3 fOverPdf = throughput * ls_b.ld;
        li += fOverPdf;
        rc.L = RoundHalf3(ls_b.ld * throughput_k);
        MaybeSetCase2OrCase3(g, pathVertex, pos, hitInfo.normal, hitInfo.t, hitInfo.ID, hitInfo.meshIdx, surface, prevHit, ls_b, 0, rc);
        float risWeight = Math::Luminance(fOverPdf);
        r.Update(risWeight, fOverPdf, rc, rngNEE);
    }
    const bool specular = surface.GlossSpecular() && (surface.metallic || surface.specTr) && (!surface.Coated() || surface.CoatSpecular());
    if (!specular)
    {
        const uint32_t seed_nee = rngNEE.State;
        DirectLightingEstimate ls = NEE_Emissive(g, pos, hitInfo.normal, surface, rngNEE);
        const float3 fOverPdf = throughput * ls.ld;
        li += fOverPdf;
        if (rc.IsCase2() || rc.IsCase3()) rc.Clear();
        rc.L = RoundHalf3(ls.ld * throughput_k);
        MaybeSetCase2OrCase3(g, pathVertex, pos, hitInfo.normal, hitInfo.t, hitInfo.ID, hitInfo.meshIdx, surface, prevHit, ls, seed_nee, rc);
        float risWeight = Math::Luminance(fOverPdf);
        r.Update(risWeight, fOverPdf, rc, rngNEE);
    }
    bsdfSample = nextBsdfSample;
}

// One lane of PathTrace (ReSTIR_PT_PathTrace.hlsl:194-358), cut at the Russian-roulette point so the 64 lanes of a wave
// can be stepped in lockstep (wave = 16x4 pixel block of the 16x8 thread group).
struct PTLane
{
    bool active = false, atRR = false, inFrame = false, valid = false;
    uint32_t x = 0, y = 0;
    float3 pos, normal; BSDF::ShadingData surface; BSDF::BSDFSample bsdfSample; RT::RayDifferentials rd;
    RNG rngReplay, rngThread, rngGroup;
    Reconnection reconnection; Reservoir r; float3 li, throughput, throughput_k; int bounce; PrevHit prevHit; float eta_curr, eta_next;
    bool inTranslucentMedium; RtRayQuery::Hit_Emissive nextHit; uint32_t seed_replay;
    // carried from phase A to phase B
    RtRayQuery::Hit hitInfo; float3 tr, dpdx, dpdy; float prevBsdfSamplePdf; LOBE prevBsdfSampleLobe; int pathVertex;
};

static void PT_PhaseA(const Globals& gl, PTLane& P, bool russianRoulette)
{
    P.atRR = false;
    if (!P.active) return;
    const Scene& sc = *gl.sc;
    P.pathVertex = P.bounce + 2;
    // NEE_EMISSIVE == 0: Hit::FindClosest<true, true>; == 1: the BSDF ray of the previous vertex's NEE (ReSTIR_PT_PathTrace.hlsl:232-239)
    P.hitInfo = gl.numEmissives == 0 ? RtRayQuery::FindClosest(sc, true, true, P.pos, P.normal, P.bsdfSample.wi, P.surface.Transmissive())
                                     : P.nextHit.ToHitInfo(sc, true);
    if (!P.hitInfo.hit) { P.active = false; return; }
    float3 newPos = mad3(P.hitInfo.t, P.bsdfSample.wi, P.pos);
    P.rd.dpdx_dpdy(newPos, P.hitInfo.normal, P.dpdx, P.dpdy);
    P.rd.ComputeUVDifferentials(P.dpdx, P.dpdy, P.hitInfo.triDiffs.dpdu, P.hitInfo.triDiffs.dpdv);
    if (!RtRayQuery::GetMaterialData(sc, -P.bsdfSample.wi, P.eta_curr, P.rd.uv_grads, P.hitInfo, P.surface, P.eta_next)) { P.active = false; return; }
    P.pos = newPos;
    P.normal = P.hitInfo.normal;
    P.prevBsdfSamplePdf = P.bsdfSample.pdf;
    P.prevBsdfSampleLobe = P.bsdfSample.lobe;
    P.tr = f3(1.0f);
    if (P.inTranslucentMedium && (P.surface.trDepth > 0))
    {
        float3 extCoeff = -log3(P.surface.baseColor_Fr0_TrCol) / P.surface.trDepth;
        P.tr = exp3(-P.hitInfo.t * extCoeff);
        P.throughput *= P.tr;
    }
    EstimateDirectAndUpdateRC(gl, P.pathVertex, P.pos, P.hitInfo, P.surface, P.prevHit, P.throughput, P.throughput_k, P.li, P.bsdfSample,
        P.nextHit, P.reconnection, P.r, P.rngThread, P.rngReplay);
    if (P.bounce >= (gl.maxNumBounces - 1)) { P.active = false; return; }
    if (P.reconnection.IsCase2() || P.reconnection.IsCase3()) P.reconnection.Clear();
    P.bounce++;
    P.atRR = russianRoulette && (P.bounce >= 3);
}

static void PT_PhaseB(const Globals& gl, PTLane& P, float waveThroughput)
{
    if (!P.active) return;
    if (P.atRR && waveThroughput < 1)
    {
        float p_terminate = zr_max(0.05f, 1 - waveThroughput);
        if (P.rngGroup.Uniform() < p_terminate) { P.active = false; return; }
        P.throughput /= (1 - p_terminate);
        P.throughput_k /= (P.reconnection.k <= P.bounce) ? (1 - p_terminate) : 1.0f;
    }
    if (gl.numEmissives == 0)     // ReSTIR_PT_PathTrace.hlsl:310-316
    {
        P.bsdfSample = BSDF::BSDFSample::Init();
        if (P.bounce < gl.maxNumBounces) P.bsdfSample = BSDF::SampleBSDF(P.normal, P.surface, P.rngReplay);
    }
    if (dot(P.bsdfSample.bsdfOverPdf, P.bsdfSample.bsdfOverPdf) == 0) { P.active = false; return; }
    const float alpha_lobe = LobeAlpha(P.surface, P.bsdfSample.lobe);
    if (P.reconnection.Empty() && CanReconnect(P.prevHit.alpha_lobe, alpha_lobe, P.prevHit.lobe, P.bsdfSample.lobe, gl.alpha_min))
    {
        P.reconnection.SetCase1(P.pathVertex, P.pos, P.hitInfo.t, P.hitInfo.normal, P.hitInfo.ID, P.hitInfo.meshIdx, -P.surface.wo,
            P.prevBsdfSampleLobe, P.prevBsdfSamplePdf, P.bsdfSample.wi, P.bsdfSample.lobe, P.bsdfSample.pdf);
        P.throughput_k = f3(1.0f);
    }
    if (P.reconnection.k <= P.bounce) P.throughput_k *= P.bsdfSample.bsdfOverPdf * P.tr;
    bool transmitted = dot(P.normal, P.bsdfSample.wi) < 0;
    P.throughput *= P.bsdfSample.bsdfOverPdf;
    P.eta_curr = transmitted ? (P.eta_curr == ETA_AIR ? P.eta_next : ETA_AIR) : P.eta_curr;
    P.inTranslucentMedium = P.eta_curr != ETA_AIR;
    P.prevHit.alpha_lobe = alpha_lobe; P.prevHit.lobe = P.bsdfSample.lobe; P.prevHit.wi = P.bsdfSample.wi; P.prevHit.pdf = P.bsdfSample.pdf;
    P.rd.UpdateRays(P.pos, P.normal, P.bsdfSample.wi, P.surface.wo, P.hitInfo.triDiffs, P.dpdx, P.dpdy, transmitted, P.surface.eta);
}

// ---- G-buffer access helpers shared by all passes
struct GBufRead
{
    uint32_t w, h; const uint32_t* baseColor; const uint32_t* normal; const uint16_t* mr; const uint32_t* motion; const uint8_t* ior;
    const uint16_t* coat; const float* depth; const uint32_t* triA; const uint32_t* triB;
    explicit GBufRead(const zr_gbuffer_planes* p)
    {
        w = p->width; h = p->height;
        baseColor = (const uint32_t*)p->plane[ZR_GB_BASE_COLOR]; normal = (const uint32_t*)p->plane[ZR_GB_NORMAL];
        mr = (const uint16_t*)p->plane[ZR_GB_METALLIC_ROUGHNESS]; motion = (const uint32_t*)p->plane[ZR_GB_MOTION_VECTOR];
        ior = (const uint8_t*)p->plane[ZR_GB_IOR]; coat = (const uint16_t*)p->plane[ZR_GB_COAT]; depth = (const float*)p->plane[ZR_GB_DEPTH];
        triA = (const uint32_t*)p->plane[ZR_GB_TRI_DIFF_GEO_A]; triB = (const uint32_t*)p->plane[ZR_GB_TRI_DIFF_GEO_B];
    }
};
struct GFlags { bool metallic, transmissive, emissive, invalid, trDepthGt0, subsurface, coated; };
static inline GFlags DecodeFlags(uint16_t mrp)
{
    const uint32_t v = (uint32_t)zr_fma((float)(mrp & 0xff) / 255.0f, 255.0f, 0.5f);
    GFlags f; f.transmissive = v & 1; f.emissive = v & 2; f.invalid = v & 4; f.trDepthGt0 = v & 8; f.subsurface = v & 16; f.coated = v & 32; f.metallic = v & 128;
    return f;
}
static inline float Roughness(uint16_t mrp) { return (float)(mrp >> 8) / 255.0f; }
static inline float2 DecodeMotion(uint32_t m)
{
    auto sn = [](uint32_t u) { int16_t s = (int16_t)(uint16_t)u; float f = (float)s / 32767.0f; return f < -1.0f ? -1.0f : f; };
    return {sn(m & 0xffff), sn(m >> 16)};
}

struct Camera
{
    float2 renderDim, jitter; float3 vbx, vby, vbz, origin; float tanHalfFOV, aspect; bool dof; float focusDepth, lensRadius;
};
static inline Camera CurrCamera(const zr_frame_constants& g)
{
    Camera c; c.renderDim = {(float)g.render_width, (float)g.render_height}; c.jitter = {g.curr_camera_jitter[0], g.curr_camera_jitter[1]};
    c.vbx = f3(g.curr_view[0], g.curr_view[1], g.curr_view[2]); c.vby = f3(g.curr_view[4], g.curr_view[5], g.curr_view[6]);
    c.vbz = f3(g.curr_view[8], g.curr_view[9], g.curr_view[10]); c.origin = f3(g.camera_pos);
    c.tanHalfFOV = g.tan_half_fov; c.aspect = g.aspect_ratio; c.dof = g.dof; c.focusDepth = g.focus_depth; c.lensRadius = g.lens_radius;
    return c;
}
static inline Camera PrevCamera(const zr_frame_constants& g)
{
    Camera c = CurrCamera(g); c.jitter = {g.prev_camera_jitter[0], g.prev_camera_jitter[1]};
    c.vbx = f3(g.prev_view[0], g.prev_view[1], g.prev_view[2]); c.vby = f3(g.prev_view[4], g.prev_view[5], g.prev_view[6]);
    c.vbz = f3(g.prev_view[8], g.prev_view[9], g.prev_view[10]); c.origin = f3(g.prev_view_inv[3], g.prev_view_inv[7], g.prev_view_inv[11]);
    return c;
}
static inline float2 LensSample(const Camera& c, uint32_t x, uint32_t y, uint32_t frame)
{
    if (!c.dof) return {0, 0};
    uint32_t hx = x, hy = y, hz = x; zr_pcg3d(&hx, &hy, &hz);
    RNG r = RNG::Init(hz, hy, frame);
    float2 l = Sampling::UniformSampleDiskConcentric(r.Uniform2D());
    return l * c.lensRadius;
}

// everything a pass reconstructs about a primary hit from the G-buffer
struct PixelSurface { float3 pos, normal, origin; float2 lensSample; float eta_next; BSDF::ShadingData surface; GFlags flags; float roughness; float z; };

// coatFromPixel: the reference reads the coat plane at a different pixel than the others in two places
// (ReSTIR_PT_Reconnect_CtT.hlsl:80 and _CtS.hlsl:99 use DTid instead of the shifted pixel); restated as is.
static PixelSurface LoadPixelSurfaceEx(const GBufRead& gb, const Camera& cam, uint32_t x, uint32_t y, uint32_t frameForLens, size_t coatPixel, bool useTrDepth)
{
    PixelSurface ps;
    const size_t px = (size_t)y * gb.w + x;
    ps.flags = DecodeFlags(gb.mr[px]); ps.roughness = Roughness(gb.mr[px]); ps.z = gb.depth[px];
    ps.lensSample = LensSample(cam, x, y, frameForLens);
    ps.origin = cam.origin;
    ps.pos = Math::WorldPosFromScreenSpace2(f2((float)x, (float)y), cam.renderDim, ps.z, cam.tanHalfFOV, cam.aspect, cam.jitter, cam.vbx, cam.vby,
        cam.vbz, cam.dof, ps.lensSample, cam.focusDepth, ps.origin);
    const uint32_t np = gb.normal[px];
    ps.normal = Math::DecodeUnitVector(f2((float)(np & 0xffff) / 65535.0f, (float)(np >> 16) / 65535.0f));
    const uint32_t bc = gb.baseColor[px];
    const float3 baseColor = Math::UnpackRGB8(bc);
    const float subsurface = ps.flags.subsurface ? (float)(bc >> 24) / 255.0f : 0.0f;
    ps.eta_next = DEFAULT_ETA_MAT;
    if (ps.flags.transmissive) ps.eta_next = zr_fma((float)gb.ior[px] / 255.0f, MAX_IOR - MIN_IOR, MIN_IOR);
    float coat_weight = 0, coat_roughness = 0, coat_ior = DEFAULT_ETA_COAT; float3 coat_color = f3(0.0f);
    if (ps.flags.coated)
    {
        const uint16_t* p = &gb.coat[4 * coatPixel];     // GBuffer::UnpackCoat, GBuffers.hlsli:107-121
        coat_weight = Math::UNorm8ToFloat((p[1] >> 8) & 0xff);
        coat_roughness = Math::UNorm8ToFloat(p[2] & 0xff);
        uint32_t c = (uint32_t)p[0] | (((uint32_t)p[1] & 0xff) << 16);
        coat_color = Math::UnpackRGB8(c);
        float normalized = Math::UNorm8ToFloat(p[2] >> 8);
        coat_ior = zr_fma(normalized, MAX_IOR - MIN_IOR, MIN_IOR);
    }
    const float3 wo = normalize(ps.origin - ps.pos);
    ps.surface = BSDF::ShadingData::Init(ps.normal, wo, ps.flags.metallic, ps.roughness, baseColor, ETA_AIR, ps.eta_next, ps.flags.transmissive,
        (useTrDepth && ps.flags.trDepthGt0) ? 1.0f : 0.0f, subsurface, coat_weight, coat_color, coat_roughness, coat_ior);
    return ps;
}
static inline PixelSurface LoadPixelSurface(const GBufRead& gb, const Camera& cam, uint32_t x, uint32_t y, uint32_t frameForLens, size_t coatPixel)
{ return LoadPixelSurfaceEx(gb, cam, x, y, frameForLens, coatPixel, true); }

// ---- r-buffers (Shift.hlsli:191-358): OffsetPathContext
struct RBuffer
{
    std::vector<uint16_t> A;   // RGBA16F (throughput, max uv grad)
    std::vector<uint32_t> B, C;  // RGBA32_UINT
    std::vector<uint16_t> D;   // R16_UINT
    void Resize(size_t n) { A.assign(4 * n, 0); B.assign(4 * n, 0); C.assign(4 * n, 0); D.assign(n, 0); }
};
struct OffsetPathContext
{
    float3 throughput, pos, normal; RT::RayDifferentials rd; BSDF::ShadingData surface; float eta_curr, eta_next; RNG rngReplay;
    static OffsetPathContext Init()
    {
        OffsetPathContext c; c.throughput = f3(0.0f); c.pos = f3(0.0f); c.normal = f3(0.0f);
        c.rd.origin_x = c.rd.dir_x = c.rd.origin_y = c.rd.dir_y = f3(0.0f); c.rd.uv_grads = {0, 0, 0, 0};
        c.surface = BSDF::ShadingData::Init(f3(0, 0, 1), f3(0, 0, 1), false, 0, f3(0.0f));   // ShadingData::Init(): placeholder, overwritten before use
        c.eta_curr = ETA_AIR; c.eta_next = DEFAULT_ETA_MAT; c.rngReplay.State = 0;
        return c;
    }
    static OffsetPathContext Load(const RBuffer& rb, size_t i, bool isCase3)
    {
        OffsetPathContext ctx = Init();
        float inAw = 0;
        if (!isCase3) { inAw = zr_f16_to_f32(rb.A[4 * i + 3]); ctx.rd.uv_grads = {inAw, inAw, inAw, inAw}; }
        ctx.throughput = f3(zr_f16_to_f32(rb.A[4 * i]), zr_f16_to_f32(rb.A[4 * i + 1]), zr_f16_to_f32(rb.A[4 * i + 2]));
        if (dot(ctx.throughput, ctx.throughput) == 0) return ctx;
        const uint32_t* inB = &rb.B[4 * i]; const uint32_t* inC = &rb.C[4 * i];
        ctx.pos = f3(zr_asfloat(inB[0]), zr_asfloat(inB[1]), zr_asfloat(inB[2]));
        auto oct = [](uint32_t e) { uint16_t v[2] = {(uint16_t)(e & 0xffff), (uint16_t)(e >> 16)}; return Math::DecodeOct32(v); };
        ctx.normal = oct(inB[3]);
        ctx.eta_curr = zr_fma(Math::UNorm8ToFloat((inC[2] >> 8) & 0xff), 1.5f, 1.0f);
        ctx.eta_next = zr_fma(Math::UNorm8ToFloat((inC[2] >> 16) & 0xff), 1.5f, 1.0f);
        float3 wo = oct(inC[0]);
        float roughness = Math::UNorm8ToFloat(inC[2] & 0xff);
        float3 baseColor = Math::UnpackRGB8(inC[1] & 0xffffff);
        uint32_t flags = inC[1] >> 24;
        bool metallic = flags & 0x1, specTr = (flags & 0x4) == 0x4;
        float trDepth = (flags & 0x8) == 0x8 ? 1.0f : 0.0f;
        bool coated = (flags & 0x10) == 0x10;
        float subsurface = Math::UNorm8ToFloat((inC[2] >> 24) & 0xff);
        float eta_next = ctx.eta_curr == ETA_AIR ? ctx.eta_next : ETA_AIR;
        float coat_weight = 0, coat_roughness = 0, coat_ior = DEFAULT_ETA_COAT; float3 coat_color = f3(0.0f);
        if (coated)
        {
            uint32_t c_w = inC[3]; uint32_t d_w = rb.D[i];
            coat_weight = Math::UNorm8ToFloat((c_w >> 24) & 0xff);
            coat_color = Math::UnpackRGB8(c_w & 0xffffff);
            coat_roughness = Math::UNorm8ToFloat(d_w & 0xff);
            coat_ior = zr_fma(Math::UNorm8ToFloat((d_w >> 8) & 0xff), 1.5f, 1.0f);
        }
        ctx.surface = BSDF::ShadingData::Init(ctx.normal, wo, metallic, roughness, baseColor, ctx.eta_curr, eta_next, specTr, trDepth,
            zr_round_f16(subsurface), coat_weight, coat_color, coat_roughness, coat_ior);
        return ctx;
    }
    void Write(RBuffer& rb, size_t i, bool isCase3) const
    {
        if (!isCase3)
        {
            float ddx_uv = zr_sqrt(uv(0) * uv(0) + uv(1) * uv(1));
            float ddy_uv = zr_sqrt(uv(2) * uv(2) + uv(3) * uv(3));
            float grad_max = zr_max(ddx_uv, ddy_uv);
            rb.A[4 * i + 3] = zr_f32_to_f16(grad_max);
        }
        rb.A[4 * i] = zr_f32_to_f16(throughput.x); rb.A[4 * i + 1] = zr_f32_to_f16(throughput.y); rb.A[4 * i + 2] = zr_f32_to_f16(throughput.z);
        if (dot(throughput, throughput) == 0) return;
        uint16_t e1[2]; Math::EncodeOct32(normal, e1);
        uint16_t e2[2]; Math::EncodeOct32(surface.wo, e2);
        uint32_t wo = e2[0] | ((uint32_t)e2[1] << 16);
        bool hasVolumetricInterior = surface.trDepth > 0;
        uint32_t flags = (uint32_t)surface.metallic | ((uint32_t)surface.specTr << 2) | ((uint32_t)hasVolumetricInterior << 3) | ((uint32_t)surface.Coated() << 4);
        uint32_t baseColor_Flags = Math::Float3ToRGB8(surface.baseColor_Fr0_TrCol) | (flags << 24);
        uint32_t roughness = Math::FloatToUNorm8(!surface.GlossSpecular() ? zr_sqrt(surface.alpha) : 0);
        uint32_t ec = Math::FloatToUNorm8((eta_curr - 1.0f) / 1.5f), en = Math::FloatToUNorm8((eta_next - 1.0f) / 1.5f);
        uint32_t ss = Math::FloatToUNorm8(surface.subsurface);
        uint32_t packed = roughness | (ec << 8) | (en << 16) | (ss << 24);
        rb.B[4 * i] = zr_asuint(pos.x); rb.B[4 * i + 1] = zr_asuint(pos.y); rb.B[4 * i + 2] = zr_asuint(pos.z); rb.B[4 * i + 3] = e1[0] | ((uint32_t)e1[1] << 16);
        rb.C[4 * i] = wo; rb.C[4 * i + 1] = baseColor_Flags; rb.C[4 * i + 2] = packed;
        if (surface.Coated())
        {
            uint32_t cw = Math::FloatToUNorm8(surface.coat_weight), cc = Math::Float3ToRGB8(surface.coat_color);
            uint32_t cr = Math::FloatToUNorm8(!surface.CoatSpecular() ? zr_sqrt(surface.coat_alpha) : 0);
            float coat_eta = surface.coat_eta >= 1.0f ? surface.coat_eta : 1.0f / surface.coat_eta;
            uint32_t ce = Math::FloatToUNorm8((coat_eta - 1.0f) / 1.5f);
            rb.C[4 * i + 3] = cc | (cw << 24);
            rb.D[i] = (uint16_t)(cr | (ce << 8));
        }
    }
    float uv(int k) const { return k == 0 ? rd.uv_grads.x : k == 1 ? rd.uv_grads.y : k == 2 ? rd.uv_grads.z : rd.uv_grads.w; }
};

// Shift.hlsli:377-474
static void Replay(const Globals& g, bool InCurrFrame, int numBounces, BSDF::BSDFSample bsdfSample, OffsetPathContext& ctx)
{
    const Scene& sc = *g.sc;
    ctx.throughput = bsdfSample.bsdfOverPdf;
    int bounce = 0;
    ctx.eta_curr = dot(ctx.normal, bsdfSample.wi) < 0 ? ctx.eta_next : ETA_AIR;
    bool inTranslucentMedium = ctx.eta_curr != ETA_AIR;
    float alpha_lobe_prev = LobeAlpha(ctx.surface, bsdfSample.lobe);
    LOBE lobe_prev = bsdfSample.lobe;
    while (true)
    {
        RtRayQuery::Hit hitInfo = RtRayQuery::FindClosest(sc, false, InCurrFrame, ctx.pos, ctx.normal, bsdfSample.wi, ctx.surface.Transmissive());
        if (!hitInfo.hit) { ctx.throughput = f3(0.0f); return; }
        float3 newPos = mad3(hitInfo.t, bsdfSample.wi, ctx.pos);
        float3 dpdx, dpdy;
        ctx.rd.dpdx_dpdy(newPos, hitInfo.normal, dpdx, dpdy);
        ctx.rd.ComputeUVDifferentials(dpdx, dpdy, hitInfo.triDiffs.dpdu, hitInfo.triDiffs.dpdv);
        if (!RtRayQuery::GetMaterialData(sc, -bsdfSample.wi, ctx.eta_curr, ctx.rd.uv_grads, hitInfo, ctx.surface, ctx.eta_next)) { ctx.throughput = f3(0.0f); return; }
        ctx.pos = mad3(hitInfo.t, bsdfSample.wi, ctx.pos);
        ctx.normal = hitInfo.normal;
        bounce++;
        if (inTranslucentMedium && (ctx.surface.trDepth > 0))
        {
            float3 extCoeff = -log3(ctx.surface.baseColor_Fr0_TrCol) / ctx.surface.trDepth;
            ctx.throughput *= exp3(-hitInfo.t * extCoeff);
        }
        if (bounce >= numBounces) break;
        bsdfSample = BSDF::SampleBSDF(ctx.normal, ctx.surface, ctx.rngReplay);
        if (dot(bsdfSample.bsdfOverPdf, bsdfSample.bsdfOverPdf) == 0) { ctx.throughput = f3(0.0f); return; }
        const float alpha_lobe = LobeAlpha(ctx.surface, bsdfSample.lobe);
        if (CanReconnect(alpha_lobe_prev, alpha_lobe, lobe_prev, bsdfSample.lobe, g.alpha_min)) { ctx.throughput = f3(0.0f); return; }
        const bool transmitted = dot(ctx.normal, bsdfSample.wi) < 0;
        ctx.eta_curr = transmitted ? (ctx.eta_curr == ETA_AIR ? ctx.eta_next : ETA_AIR) : ctx.eta_curr;
        ctx.throughput *= bsdfSample.bsdfOverPdf;
        inTranslucentMedium = ctx.eta_curr != ETA_AIR;
        alpha_lobe_prev = alpha_lobe; lobe_prev = bsdfSample.lobe;
        ctx.rd.UpdateRays(ctx.pos, ctx.normal, bsdfSample.wi, ctx.surface.wo, hitInfo.triDiffs, dpdx, dpdy, transmitted, ctx.surface.eta);
    }
}

// Shift.hlsli:818-859
static OffsetPathContext Replay_kGt2(const Globals& g, bool InCurrFrame, float3 pos, float3 normal, float ior, const BSDF::ShadingData& surface,
    RT::RayDifferentials rd, const Math::TriDifferentials& triDiffs, const Reconnection& rc)
{
    OffsetPathContext ctx = OffsetPathContext::Init();
    ctx.pos = pos; ctx.normal = normal; ctx.surface = surface; ctx.rngReplay = RNG::InitSeed(rc.seed_replay); ctx.rd = rd;
    ctx.eta_curr = ETA_AIR; ctx.eta_next = ior; ctx.throughput = f3(1.0f);
    const int numBounces = (int)rc.k - 2;
    BSDF::BSDFSample bsdfSample = BSDF::SampleBSDF(ctx.normal, ctx.surface, ctx.rngReplay);
    if (dot(bsdfSample.bsdfOverPdf, bsdfSample.bsdfOverPdf) == 0) { ctx.throughput = f3(0.0f); return ctx; }
    float3 dpdx, dpdy;
    ctx.rd.dpdx_dpdy(ctx.pos, ctx.normal, dpdx, dpdy);
    ctx.rd.ComputeUVDifferentials(dpdx, dpdy, triDiffs.dpdu, triDiffs.dpdv);
    ctx.rd.UpdateRays(ctx.pos, ctx.normal, bsdfSample.wi, ctx.surface.wo, triDiffs, dpdx, dpdy, dot(bsdfSample.wi, ctx.normal) < 0, ctx.surface.eta);
    Replay(g, InCurrFrame, numBounces, bsdfSample, ctx);
    return ctx;
}

// Shift.hlsli:476-546
static float StepPath(const Globals& g, bool InCurrFrame, OffsetPathContext& ctx, const Reconnection& rc)
{
    const Scene& sc = *g.sc;
    if (!IsLobeValid(ctx.surface, rc.lobe_k_min_1)) return 0;
    float alpha_lobe_k_min_1 = LobeAlpha(ctx.surface, rc.lobe_k_min_1);
    if (!CanReconnect(alpha_lobe_k_min_1, 1, rc.lobe_k_min_1, rc.lobe_k, g.alpha_min)) return 0;
    float3 w_k_min_1 = normalize(rc.x_k - ctx.pos);
    BSDF::BSDFSamplerEval eval = EvalBSDFSampler(ctx.normal, ctx.surface, w_k_min_1, rc.lobe_k_min_1, ctx.rngReplay);
    if (dot(eval.bsdfOverPdf, eval.bsdfOverPdf) == 0) return 0;
    RtRayQuery::Hit hitInfo = RtRayQuery::FindClosest(sc, true, InCurrFrame, ctx.pos, ctx.normal, w_k_min_1, ctx.surface.Transmissive());
    if (!hitInfo.hit || (hitInfo.ID != rc.ID)) return 0;
    const float3 y_k = mad3(hitInfo.t, w_k_min_1, ctx.pos);
    const bool transmitted = dot(ctx.normal, w_k_min_1) < 0;
    ctx.eta_curr = transmitted ? (ctx.eta_curr == ETA_AIR ? ctx.eta_next : ETA_AIR) : ctx.eta_curr;
    const bool inTranslucentMedium = ctx.eta_curr != ETA_AIR;
    if (!RtRayQuery::GetMaterialData(sc, -w_k_min_1, ctx.eta_curr, ctx.rd.uv_grads, hitInfo, ctx.surface, ctx.eta_next, RtRayQuery::TexSampler::Isotropic)) return 0;
    if (inTranslucentMedium && (ctx.surface.trDepth > 0))
    {
        float3 extCoeff = -log3(ctx.surface.baseColor_Fr0_TrCol) / ctx.surface.trDepth;
        ctx.throughput *= exp3(-hitInfo.t * extCoeff);
    }
    float partialJacobian = eval.pdf;
    partialJacobian *= zr_abs(dot(-w_k_min_1, hitInfo.normal));
    partialJacobian /= (hitInfo.t * hitInfo.t);
    ctx.pos = y_k; ctx.normal = hitInfo.normal; ctx.throughput *= eval.bsdfOverPdf;
    return partialJacobian;
}

// RPT_Util::EstimateDirect_y_k_min_1, Shift.hlsli:548-660 (SKY_SAMPLING_PREFER_PERFORMANCE == 1): re-runs the sun / sky RIS of
// NEE_NonEmissive at the offset path's y_{k-1} with the base path's pick (lt, wd, lobe) forced; returns ld, writes its pdf
static float3 EstimateDirect_y_k_min_1(const Globals& g, OffsetPathContext ctx, TYPE lt, float3 wd, LOBE lobe, RNG rngNEE, float& pdfOut)
{
    const Scene& sc = *g.sc; const zr_frame_constants& fr = *g.frame;
    const float2 u_wrs = rngNEE.Uniform2D(); (void)u_wrs;
    const float2 u_d = rngNEE.Uniform2D();
    const float2 u_c = rngNEE.Uniform2D();
    const float2 u_g = rngNEE.Uniform2D();
    const float u_wrs_b0 = rngNEE.Uniform();
    const float u_wrs_b1 = rngNEE.Uniform();
    SkyFunc leFunc; leFunc.lut = &sc.sky;
    const bool specular = ctx.surface.GlossSpecular() && (ctx.surface.metallic || ctx.surface.specTr) && (!ctx.surface.Coated() || ctx.surface.CoatSpecular());
    float3 target_z = f3(0.0f);
    float w_sum;
    {
        float3 wi_sun = -f3(fr.sun_dir);
        float pdf_b = 0, pdf_e = 0;
        const bool visible = (wi_sun.y > 0) && ((dot(wi_sun, ctx.normal) > 0) || ctx.surface.Transmissive());
        if (visible)
        {
            ctx.surface.SetWi(wi_sun, ctx.normal);
            target_z = Light::Le_Sun(ctx.pos, fr) * BSDF::Unified(ctx.surface).f;
            float ndotWi = dot(wi_sun, ctx.normal);
            pdf_b = (ndotWi < 0) && ctx.surface.ThinWalled() ? 0 : BSDF::BSDFSamplerPdf_NoDiffuse(ctx.normal, ctx.surface, wi_sun, leFunc);
            pdf_e = (!specular ? 1.0f : 0.0f) * zr_abs(ndotWi) * ZR_ONE_OVER_PI;
            pdf_e *= ctx.surface.ThinWalled() ? 0.5f : (ndotWi > 0 ? 1.0f : 0.0f);
        }
        w_sum = RT::BalanceHeuristic3(1, pdf_e, pdf_b, Math::Luminance(target_z));
    }
    if (!specular)
    {
        const bool isZ_e = lt == TYPE::SKY && lobe == LOBE::ALL;
        float pdf_unused;
        float3 wi_e = isZ_e ? wd : BSDF::SampleDiffuse(ctx.normal, u_d, pdf_unused);
        float pdf_e = zr_saturate(dot(ctx.normal, wi_e)) * ZR_ONE_OVER_PI;
        if (ctx.surface.ThinWalled()) { wi_e = u_wrs_b1 > 0.5f ? -wi_e : wi_e; pdf_e *= 0.5f; }
        ctx.surface.SetWi(wi_e, ctx.normal);
        float3 target = leFunc(wi_e) * BSDF::Unified(ctx.surface).f;
        target_z = isZ_e ? target : target_z;
        const float pdf_b = !ctx.surface.reflection && ctx.surface.ThinWalled() ? 0 : BSDF::BSDFSamplerPdf_NoDiffuse(ctx.normal, ctx.surface, wi_e, leFunc);
        const float denom = pdf_e + pdf_b;
        w_sum += denom == 0 ? 0.0f : Math::Luminance(target) / denom;
    }
    const bool isZ_b = lt == TYPE::SKY && lobe != LOBE::ALL;
    if (isZ_b)
    {
        BSDF::BSDFSamplerEval eval = EvalBSDFSampler_NoDiffuse(ctx.normal, ctx.surface, wd, lobe, leFunc);
        target_z = eval.f;
        float ndotwi = dot(wd, ctx.normal);
        float pdf_e = (!specular ? 1.0f : 0.0f) * zr_abs(ndotwi) * ZR_ONE_OVER_PI;
        pdf_e *= ctx.surface.ThinWalled() ? 0.5f : (ndotwi > 0 ? 1.0f : 0.0f);
        const float denom = eval.pdf + pdf_e;
        w_sum += denom == 0 ? 0.0f : Math::Luminance(eval.f) / denom;
    }
    else
    {
        BSDF::BSDFSample bsdfSample = BSDF::SampleBSDF_NoDiffuse(ctx.normal, ctx.surface, u_c, u_g, u_wrs_b0, u_wrs_b1, leFunc);
        float ndotwi = dot(bsdfSample.wi, ctx.normal);
        float pdf_e = (!specular ? 1.0f : 0.0f) * zr_abs(ndotwi) * ZR_ONE_OVER_PI;
        pdf_e *= ctx.surface.ThinWalled() ? 0.5f : (ndotwi > 0 ? 1.0f : 0.0f);
        const float denom = bsdfSample.pdf + pdf_e;
        w_sum += denom == 0 ? 0.0f : Math::Luminance(bsdfSample.f) / denom;
    }
    const float targetLum = Math::Luminance(target_z);
    pdfOut = w_sum > 0 ? targetLum / w_sum : 0;
    return targetLum > 0 ? target_z * w_sum / targetLum : f3(0.0f);
}

struct OffsetPath { float3 target; float partialJacobian; bool surfKMin1Tramsmissive; };

// Shift2<Emissive = true>, Shift.hlsli:662-816
static OffsetPath Shift2(const Globals& g, bool InCurrFrame, size_t DTidIdx, float3 pos, float3 normal, float ior, const BSDF::ShadingData& surface,
    RT::RayDifferentials rd, const Math::TriDifferentials& triDiffs, const Reconnection& rc, const RBuffer& rbuffer)
{
    OffsetPathContext ctx = OffsetPathContext::Init();
    ctx.pos = pos; ctx.normal = normal; ctx.surface = surface; ctx.rngReplay = RNG::InitSeed(rc.seed_replay); ctx.rd = rd;
    ctx.eta_curr = ETA_AIR; ctx.eta_next = ior; ctx.throughput = f3(1.0f);
    OffsetPath ret; ret.target = f3(0.0f); ret.partialJacobian = 0; ret.surfKMin1Tramsmissive = false;
    const int numBounces = (int)rc.k - 2;
    if (numBounces == 0)
    {
        float3 dpdx, dpdy;
        ctx.rd.dpdx_dpdy(ctx.pos, ctx.normal, dpdx, dpdy);
        ctx.rd.ComputeUVDifferentials(dpdx, dpdy, triDiffs.dpdu, triDiffs.dpdv);
        float eta = ior;
        float3 wi = normalize(rc.x_k - ctx.pos);
        ctx.rd.UpdateRays(ctx.pos, ctx.normal, wi, ctx.surface.wo, triDiffs, dpdx, dpdy, dot(wi, ctx.normal) < 0, eta);
    }
    else
    {
        ctx = OffsetPathContext::Load(rbuffer, DTidIdx, rc.IsCase3());
        if (dot(ctx.throughput, ctx.throughput) == 0) return ret;
        // Load() resets rngReplay to 0; the reference then advances that state (Shift.hlsli:707-713) -- restated as is
        for (int bounce = 0; bounce < numBounces; bounce++) for (int k = 0; k < 9; k++) ctx.rngReplay.Uniform();
    }
    ret.surfKMin1Tramsmissive = ctx.surface.specTr;
    if (!rc.IsCase3())
    {
        ret.partialJacobian = StepPath(g, InCurrFrame, ctx, rc);
        if (ret.partialJacobian == 0) return ret;
        if (rc.IsCase1())
        {
            float3 w_k = rc.w_k_lightNormal_w_sky;
            BSDF::BSDFSamplerEval eval = EvalBSDFSampler(ctx.normal, ctx.surface, w_k, rc.lobe_k, ctx.rngReplay);
            ctx.throughput *= eval.bsdfOverPdf;
            ret.target = ctx.throughput * rc.L;
            ret.partialJacobian *= eval.pdf;
            return ret;
        }
    }
    else
    {
        if (!IsLobeValid(ctx.surface, rc.lobe_k_min_1)) return ret;
        float alpha_lobe_k_min_1 = LobeAlpha(ctx.surface, rc.lobe_k_min_1);
        if (alpha_lobe_k_min_1 < g.alpha_min) return ret;
    }
    RNG rngNEE = RNG::InitSeed(rc.seed_nee);
    if (g.numEmissives == 0)      // Shift2<Emissive = false>, Shift.hlsli:788-813
    {
        const TYPE lt = rc.IsCase2() ? rc.lt_k_plus_1 : rc.lt_k;
        const LOBE lobe = rc.IsCase2() ? rc.lobe_k : rc.lobe_k_min_1;
        float pdf;
        const float3 target = EstimateDirect_y_k_min_1(g, ctx, lt, rc.w_k_lightNormal_w_sky, lobe, rngNEE, pdf);
        ret.target = ctx.throughput * target;
        if (rc.IsCase2()) ret.partialJacobian *= pdf;
        else
        {
            ret.partialJacobian = pdf;
            if (dot(ret.target, ret.target) > 0)
            {
                float3 wi = rc.lt_k == TYPE::SUN ? -f3(g.frame->sun_dir) : rc.w_k_lightNormal_w_sky;
                ret.target *= RtRayQuery::Visibility_Ray(*g.sc, ctx.pos, wi, ctx.normal, ctx.surface.Transmissive()) ? 1.0f : 0.0f;
            }
        }
        return ret;
    }
    if (rc.IsCase2())
    {
        float3 w_k = rc.w_k_lightNormal_w_sky;
        DirectLightingEstimate ls = EvalDirect_Emissive_Case2(ctx.normal, ctx.surface, w_k, rc.L, rc.dwdA, rc.lightPdf, rc.lobe_k, ctx.rngReplay, rngNEE);
        ret.target = ctx.throughput * ls.ld;
        ret.partialJacobian *= ls.pdf_solidAngle;
    }
    else
    {
        float3 wi_k_min_1 = rc.x_k - ctx.pos;
        float t = length(wi_k_min_1);
        wi_k_min_1 /= t;
        float3 lightNormal = rc.w_k_lightNormal_w_sky;
        bool twoSided = rc.lightPdf > 0;
        // note: the reference passes ctx.pos for both `pos` and `normal` (Shift.hlsli:780-782); restated as is
        DirectLightingEstimate ls = EvalDirect_Emissive_Case3(g, ctx.pos, ctx.pos, ctx.surface, wi_k_min_1, t, rc.L, lightNormal, zr_abs(rc.lightPdf),
            rc.ID, twoSided, rc.lobe_k_min_1, ctx.rngReplay, rngNEE);
        ret.target = ctx.throughput * ls.ld;
        ret.partialJacobian = ls.pdf_solidAngle;
    }
    return ret;
}

// ------------------------------------------------------------------------------------------------ renderer state
struct State
{
    uint32_t w = 0, h = 0;
    ReservoirPlanes reservoirs[2];
    RBuffer rbuffer[2];                    // [0] = CtN, [1] = NtC
    std::vector<float> target;             // RGBA32F (xyz)
    std::vector<uint8_t> neighbor;         // RG8_UINT
    std::vector<uint16_t> threadMap[2];    // R16_UINT: [0] = CtN, [1] = NtC (K12, ReSTIR_PT_Sort.hlsl)
    std::vector<uint16_t> sampleSet;       // 512 x half2 (SampleSet.hlsli)
    bool temporalValid = false;
    int currIdx = 0;
    void Resize(uint32_t w_, uint32_t h_)
    {
        w = w_; h = h_; size_t n = (size_t)w * h;
        for (auto& r : reservoirs) r.Resize(n);
        for (auto& r : rbuffer) r.Resize(n);
        target.assign(4 * n, 0); neighbor.assign(2 * n, 0); threadMap[0].assign(n, 0); threadMap[1].assign(n, 0);
        temporalValid = false; currIdx = 0;
    }
};

// Active rectangle of the passes below: the per-pixel / per-group loops visit only pixels inside it (default: everything).  It exists for the
// at-size parity tests (tests/window_parity.py): a full-resolution frame is compared on scattered windows, so the oracle renders a window + apron of
// a full-size frame -- global pixel coordinates, full-size planes, nothing else changes -- with the apron's reservoirs written in from the frame under
// test, the way the host executor is driven.  The rectangle is aligned to 32 pixels (or ends at the frame's edge), so the 16 x 4 path-tracing waves,
// the 8 x 8 groups of Reconnect_StC and the 32 x 32 sort tiles are inside or outside as a whole.
struct ActiveRect { uint32_t x0 = 0, y0 = 0, x1 = 0xffffffffu, y1 = 0xffffffffu; bool Has(uint32_t x, uint32_t y) const { return x >= x0 && y >= y0 && x < x1 && y < y1; } };
static ActiveRect g_active;

// Util.hlsli:141-159
static void WriteOutputColor(const zr_frame_constants& g, float* finalRGBA, size_t px, float3 li)
{
    li = any_nan(li) ? f3(0.0f) : li;
    float* o = finalRGBA + 4 * px;
    if (g.accumulate && g.camera_static && g.num_frames_camera_static > 1) { o[0] += li.x; o[1] += li.y; o[2] += li.z; }
    else { o[0] = li.x; o[1] = li.y; o[2] = li.z; }
}

static inline Math::TriDifferentials LoadTriDiffs(const GBufRead& gb, size_t px) { return Math::TriDifferentials::Unpack(&gb.triA[4 * px], &gb.triB[2 * px]); }
static inline RT::RayDifferentials InitRD(const Camera& c, int x, int y, float2 lens, float3 origin)
{ return RT::RayDifferentials::Init(x, y, c.renderDim, c.tanHalfFOV, c.aspect, c.jitter, c.vbx, c.vby, c.vbz, c.dof, c.focusDepth, lens, origin); }
static inline RT::RayDifferentials ZeroRD()
{ RT::RayDifferentials r; r.origin_x = r.dir_x = r.origin_y = r.dir_y = f3(0.0f); r.uv_grads = {0, 0, 0, 0}; return r; }

// ---- K11 (ReSTIR_PT_PathTrace.hlsl:360-559)
static void PathTracePass(const Scene& sc, const zr_frame_constants& g, const GBufRead& gb, const zr_params& prm, State& st, bool doTemporal,
    bool writeReservoirs, float* finalRGBA)
{
    BSDF::g_rho = &sc.rhoLUT;
    const uint32_t W = g.render_width, H = g.render_height;
    const Camera cam = CurrCamera(g);
    const bool rr = prm.flags & ZR_IND_RUSSIAN_ROULETTE;
    const bool accumulate = g.accumulate && g.camera_static;
    ReservoirPlanes& out = st.reservoirs[st.currIdx];
    std::vector<PTLane> lanes(64);
    // wave = 16 x 4 pixel block (half of a 16x8 thread group)
    for (uint32_t by = 0; by < (H + 3) / 4; by++)
    for (uint32_t bx = 0; bx < (W + 15) / 16; bx++)
    {
        if (!g_active.Has(bx * 16, by * 4)) continue;
        Globals gl[64];
        for (uint32_t l = 0; l < 64; l++)
        {
            PTLane& P = lanes[l]; P = PTLane();
            const uint32_t x = bx * 16 + (l & 15), y = by * 4 + (l >> 4);
            P.x = x; P.y = y;
            gl[l].sc = &sc; gl[l].frame = &g; gl[l].numEmissives = g.num_emissive_triangles; gl[l].alpha_min = prm.alpha_min; gl[l].maxNumBounces = (int)prm.max_non_tr_bounces;
            if (x >= W || y >= H) continue;
            P.inFrame = true;
            const size_t px = (size_t)y * W + x;
            GFlags flags = DecodeFlags(gb.mr[px]);
            if (flags.invalid || flags.emissive)
            {
                if (!accumulate) { float* o = finalRGBA + 4 * px; o[0] = o[1] = o[2] = 0; }
                continue;
            }
            P.valid = true;
            PixelSurface ps = LoadPixelSurface(gb, cam, x, y, g.frame_num, px);
            gl[l].maxNumBounces = ps.surface.specTr ? (int)prm.max_glossy_tr_bounces : (int)prm.max_non_tr_bounces;
            // RIS_InitialCandidates
            P.rngGroup = RNG::Init4(x / 16, y / 8, g.frame_num, 1);
            uint32_t sx = x, sy = y, sz = g.frame_num; zr_pcg3d(&sx, &sy, &sz);
            P.rngReplay = RNG::InitSeed(sx); P.rngThread = RNG::InitSeed(sy); P.seed_replay = sx;
            P.r = Reservoir::Init(); P.li = f3(0.0f);
            BSDF::BSDFSample bsdfSample = BSDF::SampleBSDF(ps.normal, ps.surface, P.rngReplay);
            if (dot(bsdfSample.bsdfOverPdf, bsdfSample.bsdfOverPdf) == 0) continue;
            Math::TriDifferentials triDiffs = LoadTriDiffs(gb, px);
            RT::RayDifferentials rd = InitRD(cam, (int)x, (int)y, ps.lensSample, ps.origin);
            float3 dpdx, dpdy;
            rd.dpdx_dpdy(ps.pos, ps.normal, dpdx, dpdy);
            rd.ComputeUVDifferentials(dpdx, dpdy, triDiffs.dpdu, triDiffs.dpdv);
            rd.UpdateRays(ps.pos, ps.normal, bsdfSample.wi, ps.surface.wo, triDiffs, dpdx, dpdy, dot(bsdfSample.wi, ps.normal) < 0, ps.surface.eta);
            const uint32_t numSets = prm.presampling ? prm.num_sample_sets : 0;
            gl[l].presampled = prm.presampling != 0;
            gl[l].sampleSetIdx = g.num_emissive_triangles ? P.rngGroup.UniformUintBounded_Faster(numSets) : 0u;   // ReSTIR_PT_PathTrace.hlsl:406-408
            // PathTrace prologue
            P.reconnection = Reconnection::Init();
            P.bounce = 0; P.throughput = bsdfSample.bsdfOverPdf;
            P.prevHit.alpha_lobe = LobeAlpha(ps.surface, bsdfSample.lobe); P.prevHit.lobe = bsdfSample.lobe; P.prevHit.wi = bsdfSample.wi; P.prevHit.pdf = bsdfSample.pdf;
            P.eta_curr = dot(ps.normal, bsdfSample.wi) < 0 ? ps.eta_next : ETA_AIR;
            P.throughput_k = f3(1.0f);
            P.inTranslucentMedium = P.eta_curr != ETA_AIR;
            P.pos = ps.pos; P.normal = ps.normal; P.surface = ps.surface; P.bsdfSample = bsdfSample; P.rd = rd; P.eta_next = ps.eta_next;
            if (g.num_emissive_triangles) P.nextHit = RtRayQuery::Hit_Emissive::FindClosest(sc, ps.pos, ps.normal, bsdfSample.wi, ps.surface.Transmissive());
            P.active = true;
        }
        for (;;)
        {
            bool any = false;
            for (uint32_t l = 0; l < 64; l++) { if (lanes[l].active) any = true; PT_PhaseA(gl[l], lanes[l], rr); }
            if (!any) break;
            // WaveActiveMax pinned to an integer max over the luminance bit patterns (NaN / negative -> 0)
            uint32_t bits = 0;
            for (uint32_t l = 0; l < 64; l++)
                if (lanes[l].active && lanes[l].atRR)
                {
                    float lum = Math::Luminance(lanes[l].throughput);
                    uint32_t b = (zr_isnan(lum) || lum < 0) ? 0u : zr_asuint(lum);
                    bits = b > bits ? b : bits;
                }
            const float waveThroughput = zr_asfloat(bits);
            for (uint32_t l = 0; l < 64; l++) PT_PhaseB(gl[l], lanes[l], waveThroughput);
        }
        for (uint32_t l = 0; l < 64; l++)
        {
            PTLane& P = lanes[l];
            if (!P.valid) continue;
            const size_t px = (size_t)P.y * W + P.x;
            Reservoir& r = P.r;
            r.rc.seed_replay = P.seed_replay;
            float targetLum = Math::Luminance(r.target);
            r.W = targetLum > 0 ? zr_max(r.w_sum / targetLum, 1.0f) : 0;
            if (writeReservoirs) r.Write(out, px, 0, g.num_emissive_triangles != 0);
            if (doTemporal)
            {
                float3 t = Sanitize3(r.target);
                st.target[4 * px] = t.x; st.target[4 * px + 1] = t.y; st.target[4 * px + 2] = t.z;
            }
            else
            {
                float3 li = any_nan(P.li) ? f3(0.0f) : P.li;
                float* o = finalRGBA + 4 * px;
                if (accumulate) { o[0] += li.x; o[1] += li.y; o[2] += li.z; }
                else { o[0] = li.x; o[1] = li.y; o[2] = li.z; }
            }
        }
    }
}

// x_k of a (case 1/2) reconnection moved between the current and the previous frame's instance transform
// (ReSTIR_PT_Reconnect_CtT.hlsl:258-272, _TtC.hlsl:309-329)
static void MoveXk(const Scene& sc, Reconnection& rc, bool currToPrev, bool setMotionFlag)
{
    const zr_mesh_instance& md = sc.instances[rc.meshIdx];
    float4 q_curr = normalize(Math::DecodeNormalized4(md.rotation));
    float4 q_prev = normalize(Math::DecodeNormalized4(md.prev_rotation));
    float3 s_curr = f3(zr_f16_to_f32(md.scale[0]), zr_f16_to_f32(md.scale[1]), zr_f16_to_f32(md.scale[2]));
    float3 s_prev = f3(zr_f16_to_f32(md.prev_scale[0]), zr_f16_to_f32(md.prev_scale[1]), zr_f16_to_f32(md.prev_scale[2]));
    float3 dT = f3(zr_f16_to_f32(md.d_translation[0]), zr_f16_to_f32(md.d_translation[1]), zr_f16_to_f32(md.d_translation[2]));
    float3 t_curr = f3(md.translation), t_prev = t_curr - dT;
    if (currToPrev)
    {
        float3 x_local = Math::InverseTransformTRS(rc.x_k, t_curr, q_curr, s_curr);
        rc.x_k = Math::TransformTRS(x_local, t_prev, q_prev, s_prev);
    }
    else
    {
        float3 x_local = Math::InverseTransformTRS(rc.x_k, t_prev, q_prev, s_prev);
        rc.x_k = Math::TransformTRS(x_local, t_curr, q_curr, s_curr);
    }
    if (setMotionFlag)
    {
        float4 dRot = {q_prev.x - q_curr.x, q_prev.y - q_curr.y, q_prev.z - q_curr.z, q_prev.w - q_curr.w};
        float3 dScale = s_prev - s_curr;
        rc.x_k_in_motion = dot(dT, dT) > 0;
        rc.x_k_in_motion = rc.x_k_in_motion || dot(dRot, dRot) > 0;
        rc.x_k_in_motion = rc.x_k_in_motion || dot(dScale, dScale) > 0;
    }
}

// temporal validity shared by Replay (CtT/TtC) and Reconnect CtT/TtC
struct TemporalPixel { bool ok; int px, py; PixelSurface prev; };
static TemporalPixel FindTemporal(const zr_frame_constants& g, const GBufRead& gb, const GBufRead& gbPrev, uint32_t x, uint32_t y, const PixelSurface& cur,
    float planeTh, size_t coatPixelForPrev)
{
    TemporalPixel t; t.ok = false; t.px = t.py = 0;
    const float2 renderDim = {(float)g.render_width, (float)g.render_height};
    const size_t px = (size_t)y * gb.w + x;
    const float2 motionVec = DecodeMotion(gb.motion[px]);
    const float2 currUV = {((float)x + 0.5f) / renderDim.x, ((float)y + 0.5f) / renderDim.y};
    const float2 prevUV = currUV - motionVec;
    int ppx = (int)(prevUV.x * renderDim.x), ppy = (int)(prevUV.y * renderDim.y);
    if (prevUV.x < 0 || prevUV.y < 0 || prevUV.x > 1 || prevUV.y > 1) return t;
    // prevUV == 1 maps to pixel W (out of bounds); D3D returns 0 for an out-of-bounds load, restated as "no history"
    if (ppx >= (int)gb.w || ppy >= (int)gb.h) return t;
    const size_t pp = (size_t)ppy * gb.w + ppx;
    if (gbPrev.depth[pp] == ZR_FLT_MAX) return t;
    const Camera pcam = PrevCamera(g);
    t.prev = LoadPixelSurface(gbPrev, pcam, (uint32_t)ppx, (uint32_t)ppy, g.frame_num - 1, coatPixelForPrev == (size_t)-1 ? pp : coatPixelForPrev);
    float planeDist = zr_abs(dot(cur.normal, t.prev.pos - cur.pos));
    if (!(planeDist <= planeTh * cur.z)) return t;
    if (t.prev.flags.emissive || (zr_abs(t.prev.roughness - cur.roughness) > MAX_ROUGHNESS_DIFF_TEMPORAL_REUSE) || (t.prev.flags.transmissive != cur.flags.transmissive)) return t;
    t.ok = true; t.px = ppx; t.py = ppy;
    return t;
}

// ---- K12 ReSTIR_PT_Sort.hlsl:99-368 (+ FindNeighbor :23-64, WriteOutput :72-93, Util.hlsli:20-42 EncodeSorted / DecodeSorted).
// One 16 x 16 thread group per 32 x 32 pixel tile, every thread owns a 2 x 2 quad; pixels are bucketed by the reconnection depth of the reservoir
// their shift reads (k = 2, 3, 4, >= 5, skip) and the map stores, at the position of the thread that will process a pixel, the pixel's offset
// (6 + 6 bits, biased by 31) and an error bit.  The shader orders the waves of a group by the arrival of an LDS InterlockedAdd, which a GPU
// leaves unspecified; the ABI fixes wave order = wave index, so inside a bucket pixels are in (thread, quad slot) order.
enum SortVariant { SORT_CtT = 0, SORT_TtC = 1, SORT_CtS = 2, SORT_StC = 3 };
static const uint32_t SE_SUCCESS = 0, SE_INVALID_PIXEL = 1, SE_NOT_FOUND = 2, SE_EMPTY = 4;      // Shift.hlsli:8-14
static inline void EncodeSorted(uint32_t x, uint32_t y, uint32_t mx, uint32_t my, uint32_t W, std::vector<uint16_t>& map, uint32_t error)
{
    const uint32_t dx = (uint32_t)((int)x - (int)mx + 31), dy = (uint32_t)((int)y - (int)my + 31);
    map[(size_t)my * W + mx] = (uint16_t)(dx | (dy << 7) | ((error > 0 ? 1u : 0u) << 15));
}
static inline void DecodeSorted(uint32_t x, uint32_t y, uint32_t W, const std::vector<uint16_t>& map, int& ox, int& oy, bool& error)
{
    const uint32_t e = map[(size_t)y * W + x];
    error = (e & (1u << 15)) != 0;
    ox = (int)x + (int)(e & 0x3f) - 31; oy = (int)y + (int)((e >> 7) & 0x3f) - 31;
}
// metaA: the A plane the variant reads reservoir metadata from; neighbor: K15's output (StC only)
static void SortPass(SortVariant variant, const zr_frame_constants& g, const GBufRead& gb, bool spatialResample,
    const std::vector<uint32_t>& metaA, const std::vector<uint8_t>& neighbor, std::vector<uint16_t>& map)
{
    const uint32_t W = g.render_width, H = g.render_height;
    const uint32_t dimX = (W + 31) / 32, dimY = (H + 31) / 32;
    struct Px { uint32_t x, y, gtx, gty, result; };
    std::vector<Px> bucket[5], all;
    for (uint32_t gy = 0; gy < dimY; gy++) for (uint32_t gx = 0; gx < dimX; gx++)
    {
        if (!g_active.Has(gx * 32, gy * 32)) continue;
        for (auto& b : bucket) b.clear();
        all.clear();
        const bool againstEdge = (gx == dimX - 1) || (gy == dimY - 1);
        const bool lastGroup = (gx == dimX - 1) && (gy == dimY - 1);
        for (uint32_t gidx = 0; gidx < 256; gidx++) for (uint32_t i = 0; i < 4; i++)
        {
            Px p; p.gtx = (gidx & 15) * 2 + (i & 1); p.gty = (gidx >> 4) * 2 + (i >> 1);
            p.x = gx * 32 + p.gtx; p.y = gy * 32 + p.gty;
            // FindNeighbor
            uint32_t err = SE_SUCCESS; int nx = 0, ny = 0;
            if (p.x >= W || p.y >= H) err = SE_INVALID_PIXEL;
            else
            {
                const size_t px = (size_t)p.y * W + p.x;
                const GFlags flags = DecodeFlags(gb.mr[px]);
                if (flags.invalid || flags.emissive) err = SE_INVALID_PIXEL;
                else if (variant == SORT_TtC)
                {
                    const float2 renderDim = {(float)W, (float)H};
                    const float2 motionVec = DecodeMotion(gb.motion[px]);
                    const float2 currUV = {((float)p.x + 0.5f) / renderDim.x, ((float)p.y + 0.5f) / renderDim.y};
                    const float2 prevUV = currUV - motionVec;
                    nx = (int)(prevUV.x * renderDim.x); ny = (int)(prevUV.y * renderDim.y);
                    if (prevUV.x < 0.0f || prevUV.y < 0.0f || prevUV.x > 1.0f || prevUV.y > 1.0f) err = SE_NOT_FOUND;
                }
                else if (variant == SORT_StC)
                {
                    const uint32_t ox = neighbor[2 * px], oy = neighbor[2 * px + 1];
                    if (ox == 255) err = SE_NOT_FOUND;
                    nx = (int)ox - 32 + (int)p.x; ny = (int)oy - 32 + (int)p.y;
                }
            }
            bool skip = err != SE_SUCCESS;
            uint32_t result = err, k = Reconnection::EMPTY;
            if (err == SE_SUCCESS)
            {
                const bool fromNeighbor = variant == SORT_TtC || variant == SORT_StC;
                const int rx = fromNeighbor ? nx : (int)p.x, ry = fromNeighbor ? ny : (int)p.y;
                // out-of-bounds texture loads return 0 (k field 0 = reconnection at k = 2)
                const uint32_t a = (rx >= 0 && ry >= 0 && rx < (int)W && ry < (int)H) ? metaA[(size_t)ry * W + rx] : 0u;
                const uint32_t kk = a & 0xf;
                k = kk == Reconnection::EMPTY ? kk : kk + 2;
            }
            if (k == Reconnection::EMPTY) { result |= SE_EMPTY; skip = true; }
            bool edgeCase = false;
            if (skip && againstEdge && p.x < W && p.y < H) { result = SE_SUCCESS; skip = false; edgeCase = true; }
            p.result = result;
            all.push_back(p);
            if (!skip && k == 2) bucket[0].push_back(p);
            else if (!skip && k == 3) bucket[1].push_back(p);
            else if (!skip && k == 4) bucket[2].push_back(p);
            else if (!skip && (k >= 5 || edgeCase)) bucket[3].push_back(p);
            else if (skip) bucket[4].push_back(p);
        }
        auto write = [&](const Px& p, uint32_t mgx, uint32_t mgy)
        {
            if (gx == dimX - 1 && gy != dimY - 1) std::swap(mgx, mgy);
            const uint32_t mx = gx * 32 + mgx, my = gy * 32 + mgy;
            uint32_t error;
            if (variant == SORT_TtC) error = p.result & (spatialResample ? (SE_INVALID_PIXEL | SE_NOT_FOUND) : SE_INVALID_PIXEL);
            else if (variant == SORT_StC) error = p.result & SE_INVALID_PIXEL;
            else error = p.result & (SE_INVALID_PIXEL | SE_EMPTY);
            if (mx < W && my < H) EncodeSorted(p.x, p.y, mx, my, W, map, error);
        };
        if (lastGroup) { for (const Px& p : all) write(p, p.gtx, p.gty); continue; }      // one-to-one mapping for the very last thread group
        uint32_t idx = 0;
        for (auto& b : bucket) for (const Px& p : b) { write(p, idx & 31u, idx >> 5); idx++; }
    }
}

static void TemporalPass(const Scene& sc, const zr_frame_constants& g, const GBufRead& gb, const GBufRead& gbPrev, const zr_params& prm, State& st,
    bool doSpatial, float* finalRGBA)
{
    BSDF::g_rho = &sc.rhoLUT;
    const uint32_t W = g.render_width, H = g.render_height;
    const Camera cam = CurrCamera(g);
    ReservoirPlanes& cur = st.reservoirs[st.currIdx];
    const ReservoirPlanes& prev = st.reservoirs[1 - st.currIdx];
    Globals gl; gl.sc = &sc; gl.frame = &g; gl.numEmissives = g.num_emissive_triangles; gl.alpha_min = prm.alpha_min; gl.maxNumBounces = 0;
    // the CtT passes bind the PREVIOUS acceleration structure and mesh-instance buffer (IndirectLighting.cpp:465-471, 542-548)
    Globals glPrev = gl; glPrev.sc = &sc.Prev();
    const uint32_t M_max = prm.m_max_temporal & 0xf;

    // ---- K12 Sort_TtC, Sort_CtT (IndirectLighting.cpp:383-441: dispatched whether or not SORT_TEMPORAL is set).  The temporal reconnect passes
    // have no wave operations, so which thread shifts which pixel (what the maps say) cannot change their results; the maps are outputs here.
    SortPass(SORT_TtC, g, gb, doSpatial, prev.A, st.neighbor, st.threadMap[1]);
    SortPass(SORT_CtT, g, gb, doSpatial, cur.A, st.neighbor, st.threadMap[0]);

    // ---- K13 Replay_CtT and Replay_TtC (ReSTIR_PT_Replay.hlsl:289-534); plane threshold 0.01 here
    for (int variant = 0; variant < 2; variant++)
    for (uint32_t y = 0; y < H; y++) for (uint32_t x = 0; x < W; x++)
    {
        if (!g_active.Has(x, y)) continue;
        const size_t px = (size_t)y * W + x;
        GFlags flags = DecodeFlags(gb.mr[px]);
        if (flags.invalid || flags.emissive) continue;
        PixelSurface ps = LoadPixelSurface(gb, cam, x, y, g.frame_num, px);
        TemporalPixel tp = FindTemporal(g, gb, gbPrev, x, y, ps, 0.01f, (size_t)-1);
        if (!tp.ok) continue;
        gl.maxNumBounces = flags.transmissive ? (int)prm.max_glossy_tr_bounces : (int)prm.max_non_tr_bounces;
        glPrev.maxNumBounces = gl.maxNumBounces;
        const size_t pp = (size_t)tp.py * W + tp.px;
        if (variant == 0)
        {
            Reservoir r_curr = Reservoir::Load_Metadata(cur, px);
            if (!r_curr.rc.Empty() && (r_curr.rc.k > 2))
            {
                r_curr.Load_Reconnection(cur, px, g.num_emissive_triangles != 0);
                const Camera pcam = PrevCamera(g);
                Math::TriDifferentials triDiffs = LoadTriDiffs(gbPrev, pp);
                RT::RayDifferentials rd = InitRD(pcam, tp.px, tp.py, tp.prev.lensSample, tp.prev.origin);
                float3 dpdx, dpdy;     // computed and discarded by the reference (ReSTIR_PT_Replay.hlsl:114-117)
                rd.dpdx_dpdy(tp.prev.pos, tp.prev.normal, dpdx, dpdy);
                rd.ComputeUVDifferentials(dpdx, dpdy, triDiffs.dpdu, triDiffs.dpdv);
                OffsetPathContext ctx = Replay_kGt2(glPrev, false, tp.prev.pos, tp.prev.normal, tp.prev.eta_next, tp.prev.surface, rd, triDiffs, r_curr.rc);
                ctx.Write(st.rbuffer[0], px, r_curr.rc.IsCase3());
            }
        }
        else
        {
            Reservoir r_prev = Reservoir::Load_Metadata(prev, pp);
            if (!r_prev.rc.Empty() && (r_prev.rc.k > 2))
            {
                r_prev.Load_Reconnection(prev, pp, g.num_emissive_triangles != 0);
                Math::TriDifferentials triDiffs = LoadTriDiffs(gb, px);
                RT::RayDifferentials rd = InitRD(cam, (int)x, (int)y, ps.lensSample, ps.origin);
                OffsetPathContext ctx = Replay_kGt2(gl, true, ps.pos, ps.normal, ps.eta_next, ps.surface, rd, triDiffs, r_prev.rc);
                ctx.Write(st.rbuffer[1], px, r_prev.rc.IsCase3());
            }
        }
    }

    // ---- K14 Reconnect_CtT (ReSTIR_PT_Reconnect_CtT.hlsl:130-292)
    for (uint32_t y = 0; y < H; y++) for (uint32_t x = 0; x < W; x++)
    {
        if (!g_active.Has(x, y)) continue;
        const size_t px = (size_t)y * W + x;
        GFlags flags = DecodeFlags(gb.mr[px]);
        if (flags.invalid || flags.emissive) continue;
        PixelSurface ps = LoadPixelSurface(gb, cam, x, y, g.frame_num, px);
        // coat plane of the previous G-buffer read at DTid (ReSTIR_PT_Reconnect_CtT.hlsl:80)
        TemporalPixel tp = FindTemporal(g, gb, gbPrev, x, y, ps, MAX_PLANE_DIST_REUSE, px);
        if (!tp.ok) continue;
        const size_t pp = (size_t)tp.py * W + tp.px;
        Reservoir r_curr = Reservoir::Load_NonReconnection(cur, px);
        Reservoir r_prev = Reservoir::Load_Metadata(prev, pp);
        if (r_curr.w_sum != 0 && r_prev.M > 0 && !r_curr.rc.Empty())
        {
            r_curr.Load_Reconnection(cur, px, g.num_emissive_triangles != 0);
            if (r_curr.rc.IsCase1() || r_curr.rc.IsCase2()) MoveXk(*glPrev.sc, r_curr.rc, true, false);
            glPrev.maxNumBounces = flags.transmissive ? (int)prm.max_glossy_tr_bounces : (int)prm.max_non_tr_bounces;
            Math::TriDifferentials triDiffs; RT::RayDifferentials rd = ZeroRD();
            triDiffs.dpdu = triDiffs.dpdv = triDiffs.dndu = triDiffs.dndv = f3(0.0f);
            if (r_curr.rc.k == 2)
            {
                const Camera pcam = PrevCamera(g);
                triDiffs = LoadTriDiffs(gbPrev, pp);
                rd = InitRD(pcam, tp.px, tp.py, tp.prev.lensSample, tp.prev.origin);
            }
            OffsetPath shift = Shift2(glPrev, false, px, tp.prev.pos, tp.prev.normal, tp.prev.eta_next, tp.prev.surface, rd, triDiffs, r_curr.rc, st.rbuffer[0]);
            float target_prev = Math::Luminance(shift.target);
            if (target_prev > 0)
            {
                float targetLum_curr = r_curr.W > 0 ? r_curr.w_sum / r_curr.W : 0;
                float jacobian = r_curr.rc.partialJacobian > 0 ? shift.partialJacobian / r_curr.rc.partialJacobian : 0;
                float m_curr = targetLum_curr / (targetLum_curr + (float)r_prev.M * target_prev * jacobian);
                r_curr.w_sum *= m_curr;
                cur.B[2 * px] = r_curr.w_sum;
            }
        }
    }

    // ---- K14 Reconnect_TtC (ReSTIR_PT_Reconnect_TtC.hlsl:124-390)
    for (uint32_t y = 0; y < H; y++) for (uint32_t x = 0; x < W; x++)
    {
        if (!g_active.Has(x, y)) continue;
        const size_t px = (size_t)y * W + x;
        GFlags flags = DecodeFlags(gb.mr[px]);
        if (flags.invalid || flags.emissive) continue;
        Reservoir r_curr = Reservoir::Load_NonReconnection(cur, px);
        r_curr.target = f3(st.target[4 * px], st.target[4 * px + 1], st.target[4 * px + 2]);
        PixelSurface ps = LoadPixelSurface(gb, cam, x, y, g.frame_num, px);
        TemporalPixel tp = FindTemporal(g, gb, gbPrev, x, y, ps, MAX_PLANE_DIST_REUSE, (size_t)-1);
        if (!tp.ok)
        {
            if (!doSpatial) WriteOutputColor(g, finalRGBA, px, r_curr.target * r_curr.W);
            continue;
        }
        const size_t pp = (size_t)tp.py * W + tp.px;
        Reservoir r_prev = Reservoir::Load_NonReconnection(prev, pp);
        gl.maxNumBounces = flags.transmissive ? (int)prm.max_glossy_tr_bounces : (int)prm.max_non_tr_bounces;
        const uint16_t M_new = (uint16_t)(r_curr.M + r_prev.M);
        if (r_prev.rc.Empty())
        {
            float targetLum = Math::Luminance(r_curr.target);
            r_curr.W = targetLum > 0 ? r_curr.w_sum / targetLum : 0;
            r_curr.M = M_new;
            r_curr.WriteReservoirData2(cur, px, M_max);
            if (!doSpatial) WriteOutputColor(g, finalRGBA, px, r_curr.target * r_curr.W);
            continue;
        }
        r_prev.Load_Reconnection(prev, pp, g.num_emissive_triangles != 0);
        if (r_prev.rc.IsCase1() || r_prev.rc.IsCase2()) MoveXk(sc, r_prev.rc, false, true);
        Math::TriDifferentials triDiffs; RT::RayDifferentials rd = ZeroRD();
        triDiffs.dpdu = triDiffs.dpdv = triDiffs.dndu = triDiffs.dndv = f3(0.0f);
        if (r_prev.rc.k == 2) { triDiffs = LoadTriDiffs(gb, px); rd = InitRD(cam, (int)x, (int)y, ps.lensSample, ps.origin); }
        OffsetPath shift = Shift2(gl, true, px, ps.pos, ps.normal, ps.eta_next, ps.surface, rd, triDiffs, r_prev.rc, st.rbuffer[1]);
        float targetLum_curr = Math::Luminance(shift.target);
        float jacobian = r_prev.rc.partialJacobian > 0 ? shift.partialJacobian / r_prev.rc.partialJacobian : 0;
        bool changed = false;
        if (targetLum_curr > 1e-6f && jacobian > 1e-5f)
        {
            RNG rng = RNG::Init(y, x, g.frame_num + 31);
            float targetLum_prev = r_prev.W > 0 ? r_prev.w_sum / r_prev.W : 0;
            float numerator = (float)r_prev.M * targetLum_prev;
            float denom = numerator / jacobian + targetLum_curr;
            float m_prev = denom > 0 ? numerator / denom : 0;
            float w_prev = m_prev * r_prev.W * targetLum_curr;
            if (r_curr.Update(w_prev, shift.target, r_prev.rc, rng)) { r_curr.rc.partialJacobian = shift.partialJacobian; changed = true; }
        }
        float targetLum = Math::Luminance(r_curr.target);
        r_curr.W = targetLum > 0 ? r_curr.w_sum / targetLum : 0;
        r_curr.M = M_new;
        if (changed)
        {
            r_curr.Write(cur, px, M_max, g.num_emissive_triangles != 0);
            if (doSpatial)
            {
                float3 t = Sanitize3(r_curr.target);
                st.target[4 * px] = t.x; st.target[4 * px + 1] = t.y; st.target[4 * px + 2] = t.z;
            }
        }
        else r_curr.WriteReservoirData(cur, px, M_max);
        if (!doSpatial) WriteOutputColor(g, finalRGBA, px, r_curr.target * r_curr.W);
    }
}

// Math::WorldPosFromScreenSpace, Math.hlsli:205-216
static inline float3 WorldPosFromScreenSpace(float2 pos_ss, float2 renderDim, float z_view, float tanHalfFOV, float aspect, const float* viewInv, float2 jitter)
{
    float2 uv = {(pos_ss.x + 0.5f + jitter.x) / renderDim.x, (pos_ss.y + 0.5f + jitter.y) / renderDim.y};
    float2 ndc = Math::NDCFromUV(uv);
    float3 dir_v = f3(ndc.x * aspect * tanHalfFOV * z_view, ndc.y * tanHalfFOV * z_view, z_view);
    return f3(viewInv[0] * dir_v.x + viewInv[1] * dir_v.y + viewInv[2] * dir_v.z + viewInv[3],
              viewInv[4] * dir_v.x + viewInv[5] * dir_v.y + viewInv[6] * dir_v.z + viewInv[7],
              viewInv[8] * dir_v.x + viewInv[9] * dir_v.y + viewInv[10] * dir_v.z + viewInv[11]);
}

static void SpatialPass(const Scene& sc, const zr_frame_constants& g, const GBufRead& gb, const zr_params& prm, State& st, float* finalRGBA)
{
    BSDF::g_rho = &sc.rhoLUT;
    const uint32_t W = g.render_width, H = g.render_height;
    const Camera cam = CurrCamera(g);
    const float2 renderDim = cam.renderDim;
    Globals gl; gl.sc = &sc; gl.frame = &g; gl.numEmissives = g.num_emissive_triangles; gl.alpha_min = prm.alpha_min; gl.maxNumBounces = 0;

    // ---- K15 SpatialSearch (ReSTIR_PT_SpatialSearch.hlsl:21-146)
    for (uint32_t y = 0; y < H; y++) for (uint32_t x = 0; x < W; x++)
    {
        if (!g_active.Has(x, y)) continue;
        const size_t px = (size_t)y * W + x;
        GFlags flags = DecodeFlags(gb.mr[px]);
        if (flags.invalid || flags.emissive) continue;
        const float roughness = Roughness(gb.mr[px]);
        const float viewDepth = gb.depth[px];
        const float3 pos = WorldPosFromScreenSpace(f2((float)x, (float)y), renderDim, viewDepth, g.tan_half_fov, g.aspect_ratio, g.curr_view_inv, cam.jitter);
        const uint32_t np = gb.normal[px];
        const float3 normal = Math::DecodeUnitVector(f2((float)(np & 0xffff) / 65535.0f, (float)(np >> 16) / 65535.0f));
        uint32_t sx = x, sy = y, sz = g.frame_num; zr_pcg3d(&sx, &sy, &sz);
        RNG rng = RNG::Init(sx, sy, g.frame_num);
        const float u0 = rng.Uniform();
        const uint32_t offset = rng.UniformUint();
        const float theta = u0 * ZR_TWO_PI;
        float sinTheta, cosTheta; zr_sincos(theta, &sinTheta, &cosTheta);
        int nx = 0xffff, ny = 0xffff;
        for (uint32_t i = 0; i < 3; i++)
        {
            const uint32_t si = (offset + i) & 511u;
            const float2 sampleUV = {zr_f16_to_f32(st.sampleSet[2 * si]), zr_f16_to_f32(st.sampleSet[2 * si + 1])};
            float2 rotated = {sampleUV.x * cosTheta + sampleUV.y * -sinTheta, sampleUV.x * sinTheta + sampleUV.y * cosTheta};
            rotated = rotated * (float)SPATIAL_SEARCH_RADIUS;
            // HLSL round(): round-half-to-even
            const int sxp = (int)__builtin_rintf((float)x + rotated.x), syp = (int)__builtin_rintf((float)y + rotated.y);
            if (sxp < 0 || syp < 0 || sxp >= (int)W || syp >= (int)H) continue;
            if (sxp == (int)x && syp == (int)y) continue;
            const size_t sp = (size_t)syp * W + sxp;
            GFlags sf = DecodeFlags(gb.mr[sp]);
            if (sf.invalid || sf.emissive) continue;
            if (flags.metallic != sf.metallic) continue;
            if (flags.transmissive != sf.transmissive) continue;
            if (zr_abs(Roughness(gb.mr[sp]) - roughness) > MAX_ROUGHNESS_DIFF_SPATIAL_REUSE) continue;
            const float sampleDepth = gb.depth[sp];
            const float3 samplePos = WorldPosFromScreenSpace(f2((float)sxp, (float)syp), renderDim, sampleDepth, g.tan_half_fov, g.aspect_ratio, g.curr_view_inv, cam.jitter);
            const uint32_t snp = gb.normal[sp];
            const float3 sampleNormal = Math::DecodeUnitVector(f2((float)(snp & 0xffff) / 65535.0f, (float)(snp >> 16) / 65535.0f));
            float planeDist = zr_abs(dot(normal, samplePos - pos));
            if (!(planeDist <= 0.01f * viewDepth)) continue;
            if (dot(sampleNormal, normal) < MIN_NORMAL_SIMILARITY_SPATIAL_REUSE) continue;
            nx = sxp; ny = syp;
            break;
        }
        if (nx == 0xffff) { st.neighbor[2 * px] = 255; st.neighbor[2 * px + 1] = 255; }
        else { st.neighbor[2 * px] = (uint8_t)(nx - (int)x + SPATIAL_NEIGHBOR_OFFSET); st.neighbor[2 * px + 1] = (uint8_t)(ny - (int)y + SPATIAL_NEIGHBOR_OFFSET); }
    }

    // Spatial: inputs = current temporal reservoirs, outputs = the previous frame's buffers (IndirectLighting.cpp:609-612, 682-685)
    const ReservoirPlanes& in = st.reservoirs[st.currIdx];
    ReservoirPlanes& out = st.reservoirs[1 - st.currIdx];
    st.currIdx = 1 - st.currIdx;
    // ---- K12 Sort_CtS, Sort_StC (IndirectLighting.cpp:690-742)
    const bool sortSpatial = (prm.flags & ZR_IND_SORT_SPATIAL) != 0;
    if (sortSpatial)
    {
        SortPass(SORT_CtS, g, gb, true, in.A, st.neighbor, st.threadMap[0]);
        SortPass(SORT_StC, g, gb, true, in.A, st.neighbor, st.threadMap[1]);
    }

    auto neighborOf = [&](uint32_t x, uint32_t y, int& sx, int& sy) {
        const size_t px = (size_t)y * W + x;
        if (st.neighbor[2 * px] == 255) return false;
        sx = (int)st.neighbor[2 * px] - SPATIAL_NEIGHBOR_OFFSET + (int)x; sy = (int)st.neighbor[2 * px + 1] - SPATIAL_NEIGHBOR_OFFSET + (int)y;
        return true; };

    // ---- K13 Replay_CtS / Replay_StC
    for (int variant = 0; variant < 2; variant++)
    for (uint32_t y = 0; y < H; y++) for (uint32_t x = 0; x < W; x++)
    {
        if (!g_active.Has(x, y)) continue;
        const size_t px = (size_t)y * W + x;
        GFlags flags = DecodeFlags(gb.mr[px]);
        if (flags.invalid || flags.emissive) continue;
        gl.maxNumBounces = flags.transmissive ? (int)prm.max_glossy_tr_bounces : (int)prm.max_non_tr_bounces;
        int sx, sy;
        if (variant == 0)
        {
            Reservoir r_curr = Reservoir::Load_Metadata(in, px);
            if (!r_curr.rc.Empty() && (r_curr.rc.k > 2))
            {
                r_curr.Load_Reconnection(in, px, g.num_emissive_triangles != 0);
                if (!neighborOf(x, y, sx, sy)) continue;
                const size_t sp = (size_t)sy * W + sx;
                PixelSurface pn = LoadPixelSurface(gb, cam, (uint32_t)sx, (uint32_t)sy, g.frame_num, sp);
                Math::TriDifferentials triDiffs = LoadTriDiffs(gb, sp);
                RT::RayDifferentials rd = InitRD(cam, sx, sy, pn.lensSample, pn.origin);
                OffsetPathContext ctx = Replay_kGt2(gl, true, pn.pos, pn.normal, pn.eta_next, pn.surface, rd, triDiffs, r_curr.rc);
                ctx.Write(st.rbuffer[0], px, r_curr.rc.IsCase3());
            }
        }
        else
        {
            if (!neighborOf(x, y, sx, sy)) continue;
            const size_t sp = (size_t)sy * W + sx;
            Reservoir r_spatial = Reservoir::Load_Metadata(in, sp);
            if (!r_spatial.rc.Empty() && (r_spatial.rc.k > 2))
            {
                r_spatial.Load_Reconnection(in, sp, g.num_emissive_triangles != 0);
                PixelSurface ps = LoadPixelSurface(gb, cam, x, y, g.frame_num, px);
                Math::TriDifferentials triDiffs = LoadTriDiffs(gb, px);
                RT::RayDifferentials rd = InitRD(cam, (int)x, (int)y, ps.lensSample, ps.origin);
                OffsetPathContext ctx = Replay_kGt2(gl, true, ps.pos, ps.normal, ps.eta_next, ps.surface, rd, triDiffs, r_spatial.rc);
                ctx.Write(st.rbuffer[1], px, r_spatial.rc.IsCase3());
            }
        }
    }

    // ---- K16 Reconnect_CtS (ReSTIR_PT_Reconnect_CtS.hlsl:149-230)
    for (uint32_t y = 0; y < H; y++) for (uint32_t x = 0; x < W; x++)
    {
        if (!g_active.Has(x, y)) continue;
        const size_t px = (size_t)y * W + x;
        int sx, sy;
        if (!neighborOf(x, y, sx, sy)) continue;
        GFlags flags = DecodeFlags(gb.mr[px]);
        if (flags.invalid || flags.emissive) continue;
        const size_t sp = (size_t)sy * W + sx;
        Reservoir r_curr = Reservoir::Load_NonReconnection(in, px);
        Reservoir r_spatial = Reservoir::Load_Metadata(in, sp);
        if ((r_curr.w_sum != 0) && !r_curr.rc.Empty())
        {
            r_curr.Load_Reconnection(in, px, g.num_emissive_triangles != 0);
            gl.maxNumBounces = flags.transmissive ? (int)prm.max_glossy_tr_bounces : (int)prm.max_non_tr_bounces;
            // coat plane read at DTid (ReSTIR_PT_Reconnect_CtS.hlsl:99)
            PixelSurface pn = LoadPixelSurface(gb, cam, (uint32_t)sx, (uint32_t)sy, g.frame_num, px);
            Math::TriDifferentials triDiffs; RT::RayDifferentials rd = ZeroRD();
            triDiffs.dpdu = triDiffs.dpdv = triDiffs.dndu = triDiffs.dndv = f3(0.0f);
            if (r_curr.rc.k == 2) { triDiffs = LoadTriDiffs(gb, sp); rd = InitRD(cam, sx, sy, pn.lensSample, pn.origin); }
            OffsetPath shift = Shift2(gl, true, px, pn.pos, pn.normal, pn.eta_next, pn.surface, rd, triDiffs, r_curr.rc, st.rbuffer[0]);
            float target_spatial = Math::Luminance(shift.target);
            if (target_spatial > 0)
            {
                float targetLum_curr = r_curr.W > 0 ? r_curr.w_sum / r_curr.W : 0;
                float jacobian = r_curr.rc.partialJacobian > 0 ? shift.partialJacobian / r_curr.rc.partialJacobian : 0;
                float numerator = (float)r_curr.M * targetLum_curr;
                float denom = numerator + (float)r_spatial.M * target_spatial * jacobian;
                float m_curr = denom > 0 ? numerator / denom : 0;
                r_curr.w_sum *= m_curr;
            }
            out.B[2 * px] = r_curr.w_sum;
        }
    }

    // ---- K16 Reconnect_StC (ReSTIR_PT_Reconnect_StC.hlsl:112-352); wave = 8x8 pixel group
    const bool boiling = prm.flags & ZR_IND_BOILING_SUPPRESSION;
    auto copyToNextFrame = [&](size_t px, Reservoir& r, uint32_t M_max) {
        if (!r.rc.Empty()) { r.Load_Reconnection(in, px, g.num_emissive_triangles != 0); r.Write(out, px, M_max, g.num_emissive_triangles != 0); }
        else r.WriteReservoirData(out, px, M_max); };
    auto suppress = [&](float waveAvgExclusive, Reservoir& r) {
        if (r.w_sum > 50 * waveAvgExclusive) { r.M = 0; r.w_sum = 0; r.W = 0; r.rc.Clear(); } };
    struct Lane { bool valid, hasN, spatialEmpty, resample; size_t px, sp; Reservoir r_curr, r_spatial; PixelSurface ps; uint32_t M_max; uint16_t M_new; GFlags flags; uint32_t x, y; };
    std::vector<Lane> L(64);
    for (uint32_t gy = 0; gy < (H + 7) / 8; gy++) for (uint32_t gx = 0; gx < (W + 7) / 8; gx++)
    {
        if (!g_active.Has(gx * 8, gy * 8)) continue;
        // WaveActiveSum #1 / #2: lanes that passed the invalid/emissive early-out
        float v1[64], v2[64], v3[64], v4[64];
        for (uint32_t l = 0; l < 64; l++) v1[l] = v2[l] = v3[l] = v4[l] = 0.0f;
        for (uint32_t l = 0; l < 64; l++)
        {
            Lane& a = L[l]; a.valid = a.hasN = a.spatialEmpty = a.resample = false;
            uint32_t x = gx * 8 + (l & 7), y = gy * 8 + (l >> 3);
            a.x = x; a.y = y;
            if (x >= W || y >= H) continue;
            if (sortSpatial)      // ReSTIR_PT_Reconnect_StC.hlsl:133-140: the thread shifts the pixel the NtC map assigns to its position
            {
                int mx, my; bool error;
                DecodeSorted(x, y, W, st.threadMap[1], mx, my, error);
                if (error) continue;
                x = (uint32_t)mx; y = (uint32_t)my; a.x = x; a.y = y;
            }
            a.px = (size_t)y * W + x;
            a.flags = DecodeFlags(gb.mr[a.px]);
            if (a.flags.invalid || a.flags.emissive) continue;
            a.valid = true;
            a.ps = LoadPixelSurface(gb, cam, x, y, g.frame_num, a.px);
            a.r_curr = Reservoir::Load_NonReconnection(in, a.px);
            a.r_curr.target = f3(st.target[4 * a.px], st.target[4 * a.px + 1], st.target[4 * a.px + 2]);
            int sx, sy;
            a.hasN = neighborOf(x, y, sx, sy);
            if (a.hasN) a.sp = (size_t)sy * W + sx;
            v1[l] = a.r_curr.w_sum;
            v2[l] = a.r_curr.w_sum * (a.hasN ? 0.0f : 1.0f);
        }
        const float sum1 = WaveSum64(v1), sum2 = WaveSum64(v2);
        // lanes without a neighbour finish; the rest load the spatial reservoir and join WaveActiveSum #3
        for (uint32_t l = 0; l < 64; l++)
        {
            Lane& a = L[l];
            if (!a.valid) continue;
            const float waveAvgExclusive = (sum1 - a.r_curr.w_sum) / 64.0f;
            a.M_max = prm.m_max_spatial & 0xf;
            a.M_max = !a.r_curr.rc.Empty() && a.r_curr.rc.lobe_k_min_1 == LOBE::GLOSSY_T ? std::min<uint32_t>(a.M_max, M_MAX_X_K_TRANSMISSIVE) : a.M_max;
            if (!a.hasN)
            {
                if (boiling) suppress(waveAvgExclusive, a.r_curr);
                WriteOutputColor(g, finalRGBA, a.px, a.r_curr.target * a.r_curr.W);
                copyToNextFrame(a.px, a.r_curr, a.M_max);
                continue;
            }
            a.r_spatial = Reservoir::Load_NonReconnection(in, a.sp);
            if ((a.r_curr.w_sum != 0) && (a.r_spatial.M > 0) && !a.r_curr.rc.Empty()) a.r_curr.w_sum = out.B[2 * a.px];
            a.M_new = (uint16_t)(a.r_curr.M + a.r_spatial.M);
            a.spatialEmpty = a.r_spatial.rc.Empty();
            v3[l] = a.r_curr.w_sum * (a.spatialEmpty ? 1.0f : 0.0f);
        }
        const float sum3 = WaveSum64(v3);
        for (uint32_t l = 0; l < 64; l++)
        {
            Lane& a = L[l];
            if (!a.valid || !a.hasN) continue;
            const float waveAvgExclusive = (sum1 - Reservoir::Load_NonReconnection(in, a.px).w_sum) / 64.0f;
            if (a.spatialEmpty)
            {
                if (boiling) suppress(waveAvgExclusive, a.r_curr);
                float targetLum = Math::Luminance(a.r_curr.target);
                a.r_curr.W = targetLum > 0 ? a.r_curr.w_sum / targetLum : 0;
                a.r_curr.M = a.M_new;
                copyToNextFrame(a.px, a.r_curr, a.M_max);
                WriteOutputColor(g, finalRGBA, a.px, a.r_curr.target * a.r_curr.W);
                continue;
            }
            a.resample = true;
            a.M_max = a.r_spatial.rc.x_k_in_motion ? std::min<uint32_t>(a.M_max, M_MAX_X_K_IN_MOTION) : a.M_max;
            a.r_spatial.rc.x_k_in_motion = false;
            a.r_spatial.Load_Reconnection(in, a.sp, g.num_emissive_triangles != 0);
            gl.maxNumBounces = a.flags.transmissive ? (int)prm.max_glossy_tr_bounces : (int)prm.max_non_tr_bounces;
            Math::TriDifferentials triDiffs; RT::RayDifferentials rd = ZeroRD();
            triDiffs.dpdu = triDiffs.dpdv = triDiffs.dndu = triDiffs.dndv = f3(0.0f);
            if (a.r_spatial.rc.k == 2) { triDiffs = LoadTriDiffs(gb, a.px); rd = InitRD(cam, (int)a.x, (int)a.y, a.ps.lensSample, a.ps.origin); }
            OffsetPath shift = Shift2(gl, true, a.px, a.ps.pos, a.ps.normal, a.ps.eta_next, a.ps.surface, rd, triDiffs, a.r_spatial.rc, st.rbuffer[1]);
            float targetLum_curr = Math::Luminance(shift.target);
            float targetLum_spatial = a.r_spatial.W > 0 ? a.r_spatial.w_sum / a.r_spatial.W : 0;
            float jacobian = a.r_spatial.rc.partialJacobian > 0 ? shift.partialJacobian / a.r_spatial.rc.partialJacobian : 0;
            bool changed = false;
            if (targetLum_curr > 1e-6f && jacobian > 1e-5f && jacobian < 100)
            {
                uint32_t hx = a.x, hy = a.y, hz = a.y; zr_pcg3d(&hx, &hy, &hz);
                RNG rng = RNG::Init(hx, hz, g.frame_num + 511);
                float numerator = (float)a.r_spatial.M * targetLum_spatial;
                float denom = numerator / jacobian + (float)a.r_curr.M * targetLum_curr;
                float m_spatial = denom > 0 ? numerator / denom : 0;
                float w_spatial = m_spatial * a.r_spatial.W * targetLum_curr;
                if (a.r_curr.Update(w_spatial, shift.target, a.r_spatial.rc, rng)) { a.r_curr.rc.partialJacobian = shift.partialJacobian; changed = true; }
            }
            float targetLum = Math::Luminance(a.r_curr.target);
            a.r_curr.W = targetLum > 0 ? a.r_curr.w_sum / targetLum : 0;
            a.r_curr.M = a.M_new;
            a.spatialEmpty = changed;                    // reuse the flag to carry `changed`
            a.M_max = (changed && shift.surfKMin1Tramsmissive) ? std::min<uint32_t>(a.M_max, M_MAX_X_K_TRANSMISSIVE) : a.M_max;
            v4[l] = a.r_curr.w_sum;
        }
        const float sum4 = WaveSum64(v4);
        for (uint32_t l = 0; l < 64; l++)
        {
            Lane& a = L[l];
            if (!a.resample) continue;
            if (boiling)
            {
                float waveSum = sum2 + sum3 + sum4;
                float waveAvgExclusive = (waveSum - a.r_curr.w_sum) / 64.0f;
                suppress(waveAvgExclusive, a.r_curr);
            }
            if (a.spatialEmpty) a.r_curr.Write(out, a.px, a.M_max, g.num_emissive_triangles != 0);
            else copyToNextFrame(a.px, a.r_curr, a.M_max);
            WriteOutputColor(g, finalRGBA, a.px, a.r_curr.target * a.r_curr.W);
        }
    }
}

// IndirectLighting::RenderReSTIR_PT + Render tail (IndirectLighting.cpp:877-1025)
static void Render(const Scene& sc, const zr_frame_constants& g, const zr_gbuffer_planes* gbCurr, const zr_gbuffer_planes* gbPrev, const zr_params& prm,
    State& st, float* finalRGBA)
{
    GBufRead gb(gbCurr);
    const bool doTemporal = (prm.flags & ZR_IND_TEMPORAL_RESAMPLE) && st.temporalValid && gbPrev != nullptr;
    const bool doSpatial = (prm.flags & ZR_IND_SPATIAL_RESAMPLE) && doTemporal && prm.num_spatial_passes > 0;      // IndirectLighting.cpp:906
    // reservoirs are written when temporal resampling is on for this frame or when the temporal textures were just reset
    const bool writeReservoirs = doTemporal || !st.temporalValid;
    PathTracePass(sc, g, gb, prm, st, doTemporal, writeReservoirs, finalRGBA);
    if (doTemporal) { GBufRead gp(gbPrev); TemporalPass(sc, g, gb, gp, prm, st, doSpatial, finalRGBA); }
    // for (pass < m_numSpatialPasses), IndirectLighting.cpp:616-875: every round searches again, swaps inputs and outputs (SpatialPass flips
    // st.currIdx, :682-688) and leaves the target plane alone (ReSTIR_PT_Reconnect_StC.hlsl:328-346: numPasses = 1 is hard-coded in the shader)
    if (doSpatial) for (uint32_t pass = 0; pass < prm.num_spatial_passes && pass < 2u; pass++) SpatialPass(sc, g, gb, prm, st, finalRGBA);
    st.temporalValid = true;
    st.currIdx = 1 - st.currIdx;
}

// The same frame in two steps over `rect` (x0, y0, x1, y1) of a full-size frame: stage 1 = K11 + the temporal passes, stage 2 = the spatial rounds +
// the end-of-frame bookkeeping.  Between the two the caller may overwrite reservoirs outside the rectangle's owned part (zro_rpt_write_plane_rect).
static void RenderStage(const Scene& sc, const zr_frame_constants& g, const zr_gbuffer_planes* gbCurr, const zr_gbuffer_planes* gbPrev, const zr_params& prm,
    State& st, float* finalRGBA, int stage, const uint32_t rect[4])
{
    GBufRead gb(gbCurr);
    const bool doTemporal = (prm.flags & ZR_IND_TEMPORAL_RESAMPLE) && st.temporalValid && gbPrev != nullptr;
    const bool doSpatial = (prm.flags & ZR_IND_SPATIAL_RESAMPLE) && doTemporal && prm.num_spatial_passes > 0;
    const bool writeReservoirs = doTemporal || !st.temporalValid;
    g_active.x0 = rect[0]; g_active.y0 = rect[1]; g_active.x1 = rect[2]; g_active.y1 = rect[3];
    if (stage == 1)
    {
        PathTracePass(sc, g, gb, prm, st, doTemporal, writeReservoirs, finalRGBA);
        if (doTemporal) { GBufRead gp(gbPrev); TemporalPass(sc, g, gb, gp, prm, st, doSpatial, finalRGBA); }
    }
    else
    {
        if (doSpatial) for (uint32_t pass = 0; pass < prm.num_spatial_passes && pass < 2u; pass++) SpatialPass(sc, g, gb, prm, st, finalRGBA);
        st.temporalValid = true;
        st.currIdx = 1 - st.currIdx;
    }
    g_active = ActiveRect();
}

// debug / property test: shift every pixel's current reservoir sample onto its own pixel.  For a correct shift the
// offset path equals the base path: target ratio ~= 1 (L is stored in fp16) and Jacobian == 1.
// out: 6 floats per pixel (lum(shift.target), w_sum / W, shift.partialJacobian, rc.partialJacobian, k, case)
static void SelfShift(const Scene& sc, const zr_frame_constants& g, const zr_gbuffer_planes* gbCurr, const zr_params& prm, State& st, int which, float* out)
{
    BSDF::g_rho = &sc.rhoLUT;
    GBufRead gb(gbCurr);
    const uint32_t W = g.render_width, H = g.render_height;
    const Camera cam = CurrCamera(g);
    const ReservoirPlanes& in = st.reservoirs[which == 0 ? 1 - st.currIdx : st.currIdx];
    RBuffer rb; rb.Resize((size_t)W * H);
    Globals gl; gl.sc = &sc; gl.frame = &g; gl.numEmissives = g.num_emissive_triangles; gl.alpha_min = prm.alpha_min;
    for (uint32_t y = 0; y < H; y++) for (uint32_t x = 0; x < W; x++)
    {
        if (!g_active.Has(x, y)) continue;
        const size_t px = (size_t)y * W + x;
        float* o = out + 6 * px; for (int i = 0; i < 6; i++) o[i] = 0;
        GFlags flags = DecodeFlags(gb.mr[px]);
        if (flags.invalid || flags.emissive) continue;
        Reservoir r = Reservoir::Load(in, px, g.num_emissive_triangles != 0);
        if (r.rc.Empty()) continue;
        gl.maxNumBounces = flags.transmissive ? (int)prm.max_glossy_tr_bounces : (int)prm.max_non_tr_bounces;
        PixelSurface ps = LoadPixelSurface(gb, cam, x, y, g.frame_num, px);
        Math::TriDifferentials triDiffs = LoadTriDiffs(gb, px);
        RT::RayDifferentials rd = InitRD(cam, (int)x, (int)y, ps.lensSample, ps.origin);
        if (r.rc.k > 2)
        {
            OffsetPathContext ctx = Replay_kGt2(gl, true, ps.pos, ps.normal, ps.eta_next, ps.surface, rd, triDiffs, r.rc);
            ctx.Write(rb, px, r.rc.IsCase3());
        }
        OffsetPath shift = Shift2(gl, true, px, ps.pos, ps.normal, ps.eta_next, ps.surface, rd, triDiffs, r.rc, rb);
        o[0] = Math::Luminance(shift.target); o[1] = r.W > 0 ? r.w_sum / r.W : 0; o[2] = shift.partialJacobian; o[3] = r.rc.partialJacobian;
        o[4] = (float)r.rc.k; o[5] = r.rc.IsCase1() ? 1.0f : r.rc.IsCase2() ? 2.0f : 3.0f;
    }
}

} // namespace RPT
} // namespace zro
