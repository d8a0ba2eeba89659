// zr_rpt.h -- per-pixel stage functions of ReSTIR PT (emissive-NEE variant), SURVEY.md section 8 rows K11-K16.
//
// Reference (Source/ZetaRenderPass/IndirectLighting/ReSTIR_PT/):
//   K11 ReSTIR_PT_PathTrace.hlsl:36-559 (+ ReSTIR_PT_NEE.hlsli, Reservoir.hlsli, Shift.hlsli, Util.hlsli)
//   K13 ReSTIR_PT_Replay.hlsl:70-534             K14 ReSTIR_PT_Reconnect_CtT.hlsl:130-292, _TtC.hlsl:124-390
//   K15 ReSTIR_PT_SpatialSearch.hlsl:21-146      K16 ReSTIR_PT_Reconnect_CtS.hlsl:149-230, _StC.hlsl:112-352
//   host order: IndirectLighting.cpp:877-1004
// K12 (thread sort) is a scheduling aid of the reference (it permutes which thread shades which pixel); on MI355X the
// passes run in pixel order and the wave-dependent reductions are pinned to fixed pixel groups:
//   * K11 Russian roulette: "wave" = 16x4 pixel block, max over the lanes' luminance bit patterns
//   * K16 StC boiling suppression: "wave" = 8x8 pixel group, sum = 64-lane xor-butterfly (strides 1..32), absent lanes 0
//
// Round-1 structure: one thread per pixel per pass with inline BVH traversal (same cut as the reference's passes).
// Ray differentials (RT.hlsli:309-479) only feed texture LODs: the kernels are instantiated per TEXTURED permutation
// (RptParams.textured / Globals.textured hold the template constant) and carry them only when the scene has a texture heap;
// otherwise the r-buffer's uv-gradient channel is written as 0 and never read.
#pragma once
#include "zr_stages.h"

namespace zr {
namespace rpt {

static constexpr float kMaxPlaneDistReuse = 1.0f;
static constexpr float kMaxRoughDiffTemporal = 0.3f;
static constexpr float kMaxRoughDiffSpatial = 0.05f;
static constexpr float kMinNormalSimSpatial = 0.9f;
static constexpr uint32_t kMmaxXkTransmissive = 4, kMmaxXkInMotion = 4;
static constexpr int kNeighborOffset = 32;
// which wo-only term groups (zr_dev_bsdf.h WO_*) the reconnection shifts and replays prepare on the surfaces they evaluate two to four times.
// K11 prepares all of them (six evaluations per bounce); the shift kernels run at 128 VGPRs, where every prepared term is a register taken
// from the evaluation that follows -- measured on MI355X (profiles/r05_ab_*.json)
// ZR_KEEP_IN_MEMORY(obj): the empty asm takes the object's address, which keeps it a whole object in scratch memory instead of a set of registers.
// For state that is live across a heavy call but not used by it, in a kernel that spills anyway, the memory object is the cheaper spill (K16, zr_kernels.h).
#if defined(__HIP_DEVICE_COMPILE__)
#define ZR_KEEP_IN_MEMORY(obj) asm volatile("" :: "v"(&(obj)) : "memory")
#else
#define ZR_KEEP_IN_MEMORY(obj) ((void)0)
#endif
#ifndef ZR_PIN_TEMPORAL      // (A/B: 1 = the pixel's surface, 2 = the temporal neighbour's, in the general K14)
#define ZR_PIN_TEMPORAL 0
#endif
#ifndef ZR_PREP_SHIFT
#define ZR_PREP_SHIFT 1
#endif
static constexpr uint32_t kPrepShift = ZR_PREP_SHIFT;
// ... and in the PLAIN permutations, whose lanes have the registers for every group (Cornell: temporal 0.405 -> 0.397 ms, spatial 0.593 -> 0.579, profiles/r05_ab_summary.md visit 13)
#ifndef ZR_PREP_SHIFT_PLAIN
#define ZR_PREP_SHIFT_PLAIN 15
#endif
static constexpr uint32_t kPrepShiftPlain = ZR_PREP_SHIFT_PLAIN;
ZR_HD uint32_t PrepShiftGroups(uint32_t plain) { return plain ? kPrepShiftPlain : kPrepShift; }
#ifndef ZR_PREP_K11
#define ZR_PREP_K11 15
#endif
static constexpr uint32_t kPrepK11 = ZR_PREP_K11;
static constexpr int kSearchRadius = 15;
enum LT : uint32_t { LT_NONE = 0, LT_SUN = 1, LT_SKY = 2, LT_EMISSIVE = 3 };

ZR_HD V3 RoundHalf3(V3 v) { return v3(zr_round_f16(v.x), zr_round_f16(v.y), zr_round_f16(v.z)); }
ZR_HD float Sanitize(float x) { return (zr_isnan(x) || zr_isinf(x)) ? 0.0f : x; }
ZR_HD V3 Sanitize3(V3 v) { bool bad = any_nan(v) || zr_isinf(v.x) || zr_isinf(v.y) || zr_isinf(v.z); return bad ? v3(0.0f) : v; }
ZR_HD uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }

// BSDF.hlsli:864-895
ZR_HD bool IsLobeValid(const Surface& s, uint32_t lt)
{
    if (lt == LOBE_ALL) return true;
    if (s.metallic && (lt != LOBE_GLOSSY_R) && (lt != LOBE_COAT)) return false;
    if (!s.specTr && (lt == LOBE_GLOSSY_T)) return false;
    if (s.specTr && (lt == LOBE_DIFFUSE_R)) return false;
    if (!s.ThinWalled() && (lt == LOBE_DIFFUSE_T)) return false;
    if (!s.Coated() && (lt == LOBE_COAT)) return false;
    return true;
}
ZR_HD float LobeAlpha(const Surface& s, uint32_t lt)
{
    if (lt == LOBE_GLOSSY_R || lt == LOBE_GLOSSY_T) return s.alpha;
    if (lt == LOBE_COAT) return s.coat_alpha;
    return 1.0f;
}

// BSDF.hlsli:1078-1092
ZR_HD V3 TranslucentTrOverPdf(const Surface& s, float fr)
{
    if (s.GlossSpecular()) return (1 - fr) * s.TransmissionTint();
    float alphaSq = s.alpha * s.alpha;
    return SmithG2OverG1(alphaSq, s.ndotwi, s.ndotwo) * (1 - fr) * s.TransmissionTint();
}

struct SamplerEval { float pdf; V3 bsdfOverPdf; V3 f; };

// BSDFSampling.hlsli:340-428 (NoOp target)
ZR_HD SamplerEval EvalBSDFSampler_NoSpecTr(const RhoView& rho, V3 n, Surface s, V3 wi, uint32_t lobe, V2 u_c, V2 u_g, V2 u_d)
{
    SamplerEval ret;
    const V3 scale_z = v3(1.0f);
    float w_sum = 0;
    V3 target = v3(0.0f);
    if (s.Coated())
    {
        const bool isZ = lobe == LOBE_COAT;
        const V3 wi_c = isZ ? wi : SampleCoat(s, n, u_c);
        s.SetWi_Refl(wi_c, n);
        target = Unified(rho, s).f * scale_z;
        const float lum = Luminance(target);
        const float pdf_c = CoatPdf(s), pdf_g = GlossPdf(s);
        const float pdf_d = !s.metallic ? DiffusePdf(s) : 0;
        w_sum = BalanceHeuristic3(pdf_c, pdf_g, pdf_d, lum);
    }
    {
        const bool isZ = lobe == LOBE_GLOSSY_R;
        const V3 wi_g = isZ ? wi : SampleGloss(s, n, u_g);
        s.SetWi_Refl(wi_g, n);
        const V3 target_g = Unified(rho, s).f * scale_z;
        const float lum = Luminance(target_g);
        const float pdf_g = GlossPdf(s);
        const float pdf_d = !s.metallic ? DiffusePdf(s) : 0;
        const float pdf_c = s.Coated() ? CoatPdf(s) : 0;
        w_sum += BalanceHeuristic3(pdf_g, pdf_d, pdf_c, lum);
        target = isZ ? target_g : target;
    }
    if (!s.metallic)
    {
        float unused;
        V3 w_d = SampleDiffuse(n, u_d, &unused);
        float Fr_g;
        {
            const bool isZ = lobe == LOBE_DIFFUSE_R;
            const V3 wi_d = isZ ? wi : w_d;
            s.SetWi_Refl(wi_d, n);
            Eval e = Unified(rho, s);
            const V3 target_dr = e.f * scale_z;
            Fr_g = e.Fr_g.x;
            const float lum = Luminance(target_dr);
            const float pdf_d = DiffusePdf(s), pdf_g = GlossPdf(s);
            const float pdf_c = s.Coated() ? CoatPdf(s) : 0;
            w_sum += BalanceHeuristic3(pdf_d, pdf_g, pdf_c, lum);
            target = isZ ? target_dr : target;
        }
        if (s.ThinWalled())
        {
            const bool isZ = lobe == LOBE_DIFFUSE_T;
            const V3 target_dt = DielectricBaseDiffuseTr(rho, s, Fr_g) * scale_z;
            const float lum = Luminance(target_dt);
            const float pdf_d = DiffusePdf(s);
            w_sum += lum / pdf_d;
            target = isZ ? target_dt : target;
        }
    }
    float targetLum = Luminance(target);
    ret.bsdfOverPdf = targetLum > 0 ? target * w_sum / targetLum : v3(0.0f);
    ret.pdf = w_sum > 0 ? targetLum / w_sum : 0;
    ret.f = target;
    return ret;
}

// BSDFSampling.hlsli:430-502
template<typename Func>
ZR_HD SamplerEval EvalBSDFSampler_NoDiffuse(const RhoView& rho, V3 n, Surface s, V3 wi, uint32_t lobe, Func func)
{
    const V3 wh = s.SetWi(wi, n);
    Eval e = Unified(rho, s);
    const V3 targetScale = func(wi);
    float pdf_base = 1;
    SamplerEval ret;
    ret.f = e.f * targetScale;
    if (s.Coated())
    {
        float refl_c = ReflC(rho, s);
        float pdf_coat = refl_c * s.coat_weight;
        pdf_base = 1 - pdf_coat;
        if (lobe == LOBE_COAT)
        {
            ret.pdf = CoatPdf(s) * pdf_coat;
            ret.bsdfOverPdf = ret.f / ret.pdf;
            return ret;
        }
    }
    const float wh_pdf = GlossWhPdf(s);
    ret.pdf = !s.GlossSpecular() ? wh_pdf / 4.0f : (s.ndotwh >= kMinNdotHSpecular ? 1.0f : 0.0f);
    ret.pdf *= pdf_base;
    ret.bsdfOverPdf = ret.f / ret.pdf;
    if (s.metallic || !s.specTr || e.tir) return ret;
    const V3 wi_other = lobe == LOBE_GLOSSY_T ? reflect(-s.wo, wh) : refract(-s.wo, wh, 1 / s.eta);
    float lumA = Luminance(targetScale), lumB = Luminance(func(wi_other));
    float p_r = e.Fr_g.x * (lobe == LOBE_GLOSSY_R ? lumA : lumB);
    p_r = p_r / (p_r + (1 - e.Fr_g.x) * (lobe == LOBE_GLOSSY_R ? lumB : lumA));
    if (lobe == LOBE_GLOSSY_R)
    {
        ret.bsdfOverPdf = ret.bsdfOverPdf / p_r;
        ret.pdf *= p_r;
        return ret;
    }
    ret.bsdfOverPdf = ((!s.invalid ? 1.0f : 0.0f) * (!s.reflection ? 1.0f : 0.0f)) * TranslucentTrOverPdf(s, e.Fr_g.x);
    ret.bsdfOverPdf = ret.bsdfOverPdf * TransmittanceToDielectricBaseTr(rho, s);
    ret.bsdfOverPdf = ret.bsdfOverPdf * targetScale;
    ret.bsdfOverPdf = ret.bsdfOverPdf / pdf_base;
    ret.bsdfOverPdf = ret.bsdfOverPdf / (1 - p_r);
    ret.pdf = 1 - p_r;
    ret.pdf *= s.GlossSpecular() ? (s.ndotwh >= kMinNdotHSpecular ? 1.0f : 0.0f) : wh_pdf * s.whdotwo;
    ret.pdf *= pdf_base;
    if (!s.GlossSpecular()) ret.pdf *= JacobianHalfVecToIncident_Tr(s.eta, s.whdotwo, s.whdotwi);
    return ret;
}

// BSDFSampling.hlsli:548-563
ZR_HD SamplerEval EvalBSDFSampler(const RhoView& rho, V3 n, const Surface& s, V3 wi, uint32_t lobe, Rng& rng)
{
    V2 u_c = rng.Uniform2D();
    V2 u_g = rng.Uniform2D();
    V2 u_d = rng.Uniform2D();
    rng.Uniform(); rng.Uniform(); rng.Uniform();
    if (!s.specTr) return EvalBSDFSampler_NoSpecTr(rho, n, s, wi, lobe, u_c, u_g, u_d);
    return EvalBSDFSampler_NoDiffuse(rho, n, s, wi, lobe, NoOpTarget());
}

// NEE.hlsli:28-73
struct Direct
{
    V3 ld, le, wi, pos, normal; float pdf_solidAngle, dwdA; uint32_t lt, lobe, ID; float pdf_light; bool twoSided;
};
ZR_HD Direct InitDirect()
{
    Direct r; r.ld = v3(0.0f); r.le = v3(0.0f); r.wi = v3(0.0f); r.pdf_solidAngle = 0; r.dwdA = 1; r.lt = LT_NONE; r.ID = 0xffffffffu;
    r.pos = v3(0.0f); r.pdf_light = 0; r.twoSided = true; r.normal = v3(0.0f); r.lobe = LOBE_ALL;
    return r;
}

// Shift.hlsli:16-172
struct Reconnection
{
    V3 x_k; uint32_t ID, meshIdx; float partialJacobian; V3 w; float lightPdf; uint32_t seed_replay, seed_nee; float dwdA; V3 L;
    uint32_t k, lobe_k_min_1, lobe_k, lt_k, lt_k_plus_1; bool x_k_in_motion;
    static constexpr uint32_t EMPTY = 0xf;
    ZR_HDM bool Empty() const { return k == EMPTY; }
    ZR_HDM bool IsCase2() const { return lt_k_plus_1 != LT_NONE; }
    ZR_HDM bool IsCase3() const { return lt_k != LT_NONE; }
    ZR_HDM bool IsCase1() const { return !IsCase2() && !IsCase3(); }
    ZR_HDM void Clear() { k = EMPTY; lt_k = LT_NONE; lt_k_plus_1 = LT_NONE; }
    ZR_HDM void SetCase1(int k_, V3 x_k_, float t, V3 normal_k, uint32_t hitID, uint32_t meshIdx_, V3 w_k_min_1, uint32_t l_k_min_1,
        float pdf_w_k_min_1, V3 w_k, uint32_t l_k, float pdf_w_k)
    {
        lobe_k_min_1 = l_k_min_1; k = (uint32_t)k_; x_k = x_k_; ID = hitID; meshIdx = meshIdx_; lt_k = LT_NONE; lobe_k = l_k;
        w = w_k; lt_k_plus_1 = LT_NONE;
        partialJacobian = pdf_w_k_min_1;
        float cos_theta_k = zr_abs(dot(-w_k_min_1, normal_k));
        partialJacobian *= cos_theta_k / (t * t);
        partialJacobian *= pdf_w_k;
    }
    ZR_HDM void SetCase2(int k_, V3 x_k_, float t, V3 normal_k, uint32_t hitID, uint32_t meshIdx_, V3 w_k_min_1, uint32_t l_k_min_1,
        float pdf_w_k_min_1, V3 w_k, uint32_t l_k, float pdf_w_k, uint32_t t_k_plus_1, float pdf_light, V3 le, uint32_t seed, float dwdA_)
    {
        lobe_k_min_1 = l_k_min_1; k = (uint32_t)k_; x_k = x_k_; ID = hitID; meshIdx = meshIdx_; lt_k = LT_NONE; lobe_k = l_k;
        w = w_k; lt_k_plus_1 = t_k_plus_1; lightPdf = pdf_light; dwdA = dwdA_; seed_nee = seed; L = RoundHalf3(le);
        partialJacobian = pdf_w_k_min_1;
        float cos_theta_k = zr_abs(dot(-w_k_min_1, normal_k));
        partialJacobian *= cos_theta_k / (t * t);
        if (lobe_k != LOBE_ALL) partialJacobian *= pdf_w_k;
    }
    ZR_HDM void SetCase3(int k_, V3 x_k_, uint32_t t, uint32_t l_k_min_1, uint32_t lightID, V3 le, V3 lightNormal, float pdf_solidAngle,
        float pdf_light, float dwdA_, V3 w_sky, bool twoSided, uint32_t seed)
    {
        lobe_k_min_1 = l_k_min_1; k = (uint32_t)k_; x_k = x_k_; ID = lightID; lt_k = t; seed_nee = seed;
        partialJacobian = l_k_min_1 == LOBE_ALL ? 1.0f : pdf_solidAngle * dwdA_;
        lightPdf = twoSided ? pdf_light : -pdf_light;
        L = RoundHalf3(le);
        lt_k_plus_1 = LT_NONE;
        if (t == LT_EMISSIVE) w = lightNormal;
        else if (t == LT_SKY) w = w_sky;
    }
};
ZR_HD Reconnection InitReconnection()
{
    Reconnection r;
    r.k = Reconnection::EMPTY; r.lt_k = LT_NONE; r.lt_k_plus_1 = LT_NONE; r.partialJacobian = 0; r.x_k = v3(ZR_FLT_MAX); r.seed_replay = 0;
    r.w = v3(0.0f); r.L = v3(0.0f); r.lightPdf = 0; r.seed_nee = 0; r.dwdA = 0;
    r.ID = 0; r.meshIdx = 0; r.lobe_k_min_1 = LOBE_DIFFUSE_R; r.lobe_k = LOBE_DIFFUSE_R; r.x_k_in_motion = false;
    return r;
}

// Shift.hlsli:360-375
ZR_HD bool CanReconnect(float alpha_k_min_1, float alpha_k, uint32_t lobe_k_min_1, uint32_t lobe_k, float alpha_min)
{
    if ((alpha_k_min_1 < alpha_min) || (alpha_k < alpha_min)) return false;
    if ((lobe_k_min_1 == LOBE_GLOSSY_T) && (lobe_k == LOBE_GLOSSY_T)) return false;
    return true;
}

// ---- reservoir planes in HBM (reference formats, IndirectLighting.h:128-144): 62 B per pixel and set
struct ResPlanes
{
    uint32_t* A;     // RGBA8_UINT  (x | y << 8 | z << 16)
    float* B;        // RG32F  (w_sum, W)
    U4* C; U4* D;    // RGBA32_UINT
    uint16_t* E;     // R16F
    float* F;        // RG32F
    uint32_t* G;     // RG32_UINT
};

// K11: while a path is traced the reservoir's SELECTED reconnection is cold state -- written when a candidate wins the
// resampling, read once in the epilogue -- 23 words that the register allocator otherwise spills to scratch.  With a park bound, Update stores a
// winning candidate into it (17 words: the small integers share one) and the epilogue reads it back; the registers of Reservoir::rc are dead in
// between.  p: this lane's first word, word k at p[k * stride] (LDS, [word][lane]: conflict-free); p == nullptr (every other user of Reservoir):
// plain member copies, and the branch folds away.
struct RcPark { ZR_LDS_AS uint32_t* p; uint32_t stride; };
static constexpr uint32_t kRcParkWords = 17;
ZR_HD void RcParkStore(const RcPark& k, const Reconnection& rc)
{
    const uint32_t s = k.stride;
    k.p[0] = zr_asuint(rc.x_k.x); k.p[s] = zr_asuint(rc.x_k.y); k.p[2 * s] = zr_asuint(rc.x_k.z); k.p[3 * s] = rc.ID; k.p[4 * s] = rc.meshIdx;
    k.p[5 * s] = zr_asuint(rc.partialJacobian); k.p[6 * s] = zr_asuint(rc.w.x); k.p[7 * s] = zr_asuint(rc.w.y); k.p[8 * s] = zr_asuint(rc.w.z);
    k.p[9 * s] = zr_asuint(rc.lightPdf); k.p[10 * s] = rc.seed_replay; k.p[11 * s] = rc.seed_nee; k.p[12 * s] = zr_asuint(rc.dwdA);
    k.p[13 * s] = zr_asuint(rc.L.x); k.p[14 * s] = zr_asuint(rc.L.y); k.p[15 * s] = zr_asuint(rc.L.z);
    k.p[16 * s] = (rc.k & 0xffu) | (rc.lobe_k_min_1 << 8) | (rc.lobe_k << 12) | (rc.lt_k << 16) | (rc.lt_k_plus_1 << 20) | ((rc.x_k_in_motion ? 1u : 0u) << 24);
}
ZR_HD void RcParkLoad(const RcPark& k, Reconnection& rc)
{
    const uint32_t s = k.stride;
    rc.x_k = v3(zr_asfloat(k.p[0]), zr_asfloat(k.p[s]), zr_asfloat(k.p[2 * s])); rc.ID = k.p[3 * s]; rc.meshIdx = k.p[4 * s];
    rc.partialJacobian = zr_asfloat(k.p[5 * s]); rc.w = v3(zr_asfloat(k.p[6 * s]), zr_asfloat(k.p[7 * s]), zr_asfloat(k.p[8 * s]));
    rc.lightPdf = zr_asfloat(k.p[9 * s]); rc.seed_replay = k.p[10 * s]; rc.seed_nee = k.p[11 * s]; rc.dwdA = zr_asfloat(k.p[12 * s]);
    rc.L = v3(zr_asfloat(k.p[13 * s]), zr_asfloat(k.p[14 * s]), zr_asfloat(k.p[15 * s]));
    const uint32_t m = k.p[16 * s];
    rc.k = m & 0xffu; rc.lobe_k_min_1 = (m >> 8) & 0xfu; rc.lobe_k = (m >> 12) & 0xfu; rc.lt_k = (m >> 16) & 0xfu; rc.lt_k_plus_1 = (m >> 20) & 0xfu; rc.x_k_in_motion = ((m >> 24) & 1u) != 0;
}

struct Reservoir
{
    float w_sum, W; V3 target; Reconnection rc; uint32_t M;
    RcPark park = {nullptr, 0}; bool parked = false;      // K11 only (see RcPark): the selected reconnection lives at `park` once `parked`
    // Reservoir.hlsli:23-45
    ZR_HDM bool Update(float weight, V3 target_, const Reconnection& rc_, Rng& rng)
    {
        if (zr_isnan(weight) || zr_isinf(weight)) return false;
        M += 1;
        if (weight == 0) return false;
        w_sum += weight;
        if (rng.Uniform() < (weight / w_sum))
        {
            if (park.p) { RcParkStore(park, rc_); parked = true; } else rc = rc_;
            target = target_; return true;
        }
        return false;
    }
    ZR_HDM void UnpackMetadata(uint32_t a)
    {
        uint32_t mx = a & 0xff, my = (a >> 8) & 0xff, mz = (a >> 16) & 0xff;
        uint32_t kk = mx & 0xf;
        rc.k = kk == Reconnection::EMPTY ? kk : kk + 2;
        rc.lobe_k_min_1 = umin(my & 0x7, 5);
        rc.lobe_k = umin((my >> 3) & 0x7, 5);
        rc.lt_k = (my >> 6) & 0x3;
        rc.lt_k_plus_1 = mz & 0x3;
        rc.x_k_in_motion = (mz >> 2) != 0;
        M = mx >> 4;
    }
    // LoadCase1/2/3<Emissive> (Reservoir.hlsli:47-139)
    ZR_HDM void Load_Reconnection(const ResPlanes& p, size_t i, bool emissive)
    {
        const U4 c = p.C[i], d = p.D[i];
        // meshIdx and seed_nee are assigned once, on every path, from values selected per case: stores to DIFFERENT fields from the branches get merged by
        // the compiler into one store through a selected address, and a reservoir touched that way stays in scratch memory as a whole (SROA gives up on it)
        uint32_t meshIdx_ = rc.meshIdx, seed_nee_ = rc.seed_nee;
        if (!emissive && !rc.IsCase1())
        {
            rc.seed_replay = c.y; rc.ID = c.z; rc.partialJacobian = zr_asfloat(c.x);
            if (rc.IsCase2())
            {
                rc.x_k = v3(zr_asfloat(c.w), zr_asfloat(d.x), zr_asfloat(d.y));
                if (rc.lt_k_plus_1 == LT_SKY) { rc.w = DecodeOct32u(d.z); seed_nee_ = d.w; }
                meshIdx_ = p.G[2 * i + 1];
            }
            else if (rc.lt_k == LT_SKY) { rc.w = DecodeOct32u(d.z); seed_nee_ = d.w; }
            rc.meshIdx = meshIdx_; rc.seed_nee = seed_nee_;
            return;
        }
        rc.seed_replay = c.y; rc.ID = c.z;
        rc.x_k = v3(zr_asfloat(c.w), zr_asfloat(d.x), zr_asfloat(d.y));
        rc.w = DecodeOct32u(d.z);
        rc.L = v3(zr_f16_to_f32((uint16_t)(d.w & 0xffff)), zr_f16_to_f32((uint16_t)(d.w >> 16)), zr_f16_to_f32(p.E[i]));
        if (rc.IsCase1()) { rc.partialJacobian = zr_asfloat(c.x); meshIdx_ = p.G[2 * i + 1]; }
        else if (rc.IsCase2())
        {
            rc.partialJacobian = zr_asfloat(c.x);
            rc.lightPdf = p.F[2 * i]; rc.dwdA = p.F[2 * i + 1];
            seed_nee_ = p.G[2 * i]; meshIdx_ = p.G[2 * i + 1];
        }
        else
        {
            rc.partialJacobian = rc.lobe_k_min_1 == LOBE_ALL ? 1.0f : zr_asfloat(c.x);
            rc.lightPdf = p.F[2 * i];
            seed_nee_ = c.x;
        }
        rc.meshIdx = meshIdx_; rc.seed_nee = seed_nee_;
    }
    static ZR_HDM uint32_t PackA_x(const Reconnection& rc, uint32_t m)
    { uint32_t k = rc.Empty() ? rc.k : (rc.k > 2 ? rc.k : 2) - 2; return k | (m << 4); }
    ZR_HDM void WriteReservoirData(const ResPlanes& p, size_t i, uint32_t M_max) const
    {
        uint32_t m = umin(M, M_max);
        p.A[i] = (p.A[i] & 0xffffff00u) | (PackA_x(rc, m) & 0xff);
        p.B[2 * i] = w_sum; p.B[2 * i + 1] = W;
    }
    ZR_HDM void WriteReservoirData2(const ResPlanes& p, size_t i, uint32_t M_max) const
    {
        uint32_t m = umin(M, M_max);
        p.A[i] = (p.A[i] & 0xffffff00u) | (PackA_x(rc, m) & 0xff);
        p.B[2 * i + 1] = W;
    }
    // Write<Emissive>, Reservoir.hlsli:283-330, 367-456 (the non-emissive variant writes only some components of C / D / G)
    ZR_HDM void Write(const ResPlanes& p, size_t i, uint32_t M_max, bool emissive)
    {
        uint32_t m = M_max == 0 ? M : umin(M, M_max);
        uint32_t mx = PackA_x(rc, m) & 0xff;
        uint32_t my = (rc.lobe_k_min_1 | (rc.lobe_k << 3) | (rc.lt_k << 6)) & 0xff;
        uint32_t mz = (rc.lt_k_plus_1 | ((uint32_t)rc.x_k_in_motion << 2)) & 0xff;
        p.A[i] = (p.A[i] & 0xff000000u) | mx | (my << 8) | (mz << 16);
        w_sum = Sanitize(w_sum); W = Sanitize(W);
        p.B[2 * i] = w_sum; p.B[2 * i + 1] = W;
        if (rc.Empty()) return;
        V2 e = EncodeUnitVector(rc.w);
        uint32_t w_enc = FloatToUNorm16(e.x) | (FloatToUNorm16(e.y) << 16);
        uint32_t lh = (uint32_t)zr_f32_to_f16(rc.L.x) | ((uint32_t)zr_f32_to_f16(rc.L.y) << 16);
        if (!emissive && !rc.IsCase1())
        {
            U4 c = p.C[i], d = p.D[i];
            c.x = zr_asuint(rc.partialJacobian); c.y = rc.seed_replay; c.z = rc.ID;
            if (rc.IsCase2())
            {
                c.w = zr_asuint(rc.x_k.x);
                d.x = zr_asuint(rc.x_k.y); d.y = zr_asuint(rc.x_k.z);
                if (rc.lt_k_plus_1 == LT_SKY) { d.z = w_enc; d.w = rc.seed_nee; }
                p.G[2 * i + 1] = rc.meshIdx;
            }
            else if (rc.lt_k == LT_SKY) { d.z = w_enc; d.w = rc.seed_nee; }
            p.C[i] = c; p.D[i] = d;
            return;
        }
        U4 c, d;
        c.y = rc.seed_replay; c.z = rc.ID; c.w = zr_asuint(rc.x_k.x);
        d.x = zr_asuint(rc.x_k.y); d.y = zr_asuint(rc.x_k.z); d.z = w_enc; d.w = lh;
        p.E[i] = zr_f32_to_f16(rc.L.z);
        if (rc.IsCase1()) { c.x = zr_asuint(rc.partialJacobian); p.G[2 * i + 1] = rc.meshIdx; }
        else if (rc.IsCase2())
        {
            c.x = zr_asuint(rc.partialJacobian);
            p.F[2 * i] = rc.lightPdf; p.F[2 * i + 1] = rc.dwdA;
            p.G[2 * i] = rc.seed_nee; p.G[2 * i + 1] = rc.meshIdx;
        }
        else
        {
            c.x = rc.lobe_k_min_1 == LOBE_ALL ? rc.seed_nee : zr_asuint(rc.partialJacobian);
            p.F[2 * i] = rc.lightPdf;
        }
        p.C[i] = c; p.D[i] = d;
    }
};
ZR_HD Reservoir InitReservoir() { Reservoir r; r.rc = InitReconnection(); r.w_sum = 0; r.W = 0; r.M = 0; r.target = v3(0.0f); return r; }
ZR_HD Reservoir Load_Metadata(const ResPlanes& p, size_t i) { Reservoir r = InitReservoir(); r.UnpackMetadata(p.A[i]); return r; }
ZR_HD Reservoir Load_NonReconnection(const ResPlanes& p, size_t i)
{ Reservoir r = InitReservoir(); r.UnpackMetadata(p.A[i]); r.w_sum = p.B[2 * i]; r.W = p.B[2 * i + 1]; return r; }

// cnt: this lane's ray counters {closest-hit queries, shadow / visibility queries} (never null)
// emissive == false selects the NEE_EMISSIVE == 0 shader variants (sun + sky lighting): the kernels are instantiated per variant and set
// it from a template constant, so the other variant's code folds away; frame = cbFrameConstants (sun, atmosphere)
// scPrev: the previous frame's acceleration structure + mesh instances (RT_SCENE_BVH_PREV / RT_FRAME_MESH_INSTANCES_PREV), bound instead of
// `sc` by the CtT passes of ReSTIR PT (IndirectLighting.cpp:465-471, 542-548) and by the temporal shifts of the DI passes (g_bvh_prev)
struct Globals { bool textured = false; const SceneView* sc; const SceneView* scPrev = nullptr; uint32_t numEmissives; int maxNumBounces; float alpha_min; TravStack stack; uint32_t* cnt; bool presampled; uint32_t sampleSetIdx;
    const zr_frame_constants* frame; bool emissive; };

// ---- ray queries.  Every query is "build the ray" -> traverse -> "read the hit"; the halves are separate functions so that a kernel can
// run the traversal somewhere else than the calling lane (the block-cooperative ray pool of zr_kernels.h: BlockTrace).  want == false: the
// query ends without a ray (the reference's early-outs).
struct TraceReq { bool want; V3 o, d; float tmin, tmax; uint32_t mask; bool anyHit, filterID; uint32_t ignoreID; };
ZR_HD TraceReq NoTraceReq() { TraceReq q; q.want = false; q.o = v3(0.0f); q.d = v3(0.0f); q.tmin = 0; q.tmax = 0; q.mask = 0; q.anyHit = false; q.filterID = false; q.ignoreID = 0; return q; }
ZR_HD RawHit NoRawHit() { RawHit h; h.t = 0; h.u = 0; h.v = 0; h.tri = kInvalidTri; return h; }
// the query kind is a compile-time constant where the query is traced in place (the flags of `q` are for pools that mix kinds):
// AnyHit / FilterID = false, false for the closest-hit queries, true, true for the approximate light segment
template<bool AnyHit, bool FilterID>
ZR_HD RawHit TraceInline(const SceneView& sc, const TraceReq& q, const TravStack& stack)
{ return q.want ? TraverseDyn(sc, q.o, q.d, q.tmin, q.tmax, AnyHit ? (uint32_t)ZR_SUBGROUP_NON_EMISSIVE : (uint32_t)ZR_SUBGROUP_ALL, stack, AnyHit, FilterID, q.ignoreID) : NoRawHit(); }

struct HitEm { bool hit; float t; uint32_t mesh, prim, emissiveTriIdx; float bu, bv; };
// Hit_Emissive::FindClosest, RayQuery.hlsli:146-207
ZR_HD TraceReq ClosestEmReq(uint32_t* cnt, V3 pos, V3 normal, V3 wi, bool transmissive)
{
    TraceReq q = NoTraceReq();
    F4 ro, rd;
    if (!MakeClosestRay(pos, normal, wi, transmissive, true, &ro, &rd)) return q;
    cnt[0]++;
    q.want = true; q.o = xyz(ro); q.d = xyz(rd); q.tmin = ro.w; q.tmax = rd.w; q.mask = ZR_SUBGROUP_ALL;
    return q;
}
ZR_HD HitEm ClosestEmResult(const SceneView& sc, const TraceReq& q, const RawHit& h)
{
    HitEm r; r.hit = false; r.emissiveTriIdx = 0xffffffffu; r.t = 0; r.mesh = 0; r.prim = 0; r.bu = 0; r.bv = 0;
    if (!q.want || h.tri == kInvalidTri) return r;
    const TriMeta tm = sc.triMeta[h.tri];
    r.hit = true; r.t = h.t; r.bu = h.u; r.bv = h.v; r.mesh = tm.mesh; r.prim = tm.prim;
    const uint32_t base = sc.instances[tm.mesh].base_emissive_tri_offset;
    if (base != 0xffffffffu) r.emissiveTriIdx = base + tm.prim;
    return r;
}
ZR_HD HitEm FindClosestEm(const Globals& g, V3 pos, V3 normal, V3 wi, bool transmissive)
{
    const TraceReq q = ClosestEmReq(g.cnt, pos, normal, wi, transmissive);
    return ClosestEmResult(*g.sc, q, TraceInline<false, false>(*g.sc, q, g.stack));
}
// Hit::FindClosest<ID = true>, RayQuery.hlsli:15-144
ZR_HD bool FindClosestID(const Globals& g, bool currFrame, V3 pos, V3 normal, V3 wi, bool transmissive, HitInfo& hit, bool wantDiffs = false)
{
    F4 ro, rd;
    if (!MakeClosestRay(pos, normal, wi, transmissive, false, &ro, &rd)) return false;
    g.cnt[0]++;
    RawHit h = Traverse<false>(*g.sc, xyz(ro), xyz(rd), ro.w, rd.w, ZR_SUBGROUP_ALL, g.stack);
    if (h.tri == kInvalidTri) return false;
    const TriMeta tm = g.sc->triMeta[h.tri];
    hit.t = h.t;
    if (wantDiffs) FillHit<true>(*g.sc, tm.mesh, tm.prim, h.u, h.v, true, hit, currFrame);
    else FillHit<false>(*g.sc, tm.mesh, tm.prim, h.u, h.v, true, hit, currFrame);
    return true;
}
// Visibility_Segment with APPROXIMATE_EMISSIVE_SHADOW_RAY == 1 (RayQuery.hlsli:337-406); visible iff the ray was emitted and hit nothing
ZR_HD TraceReq SegmentApproxReq(uint32_t* cnt, V3 origin, V3 wi, float rayT, V3 normal, uint32_t triID, bool transmissive)
{
    TraceReq q = NoTraceReq();
    if (triID == 0xffffffffu) return q;
    if (rayT < 1e-6f) return q;
    float ndotwi = dot(normal, wi);
    if (ndotwi == 0) return q;
    if (ndotwi < 0)
    {
        if (transmissive) normal = normal * -1.0f;
        else return q;
    }
    const V3 o = OffsetRayRTG(origin, normal);
    const float tminv = 3e-6f;
    const float tmax = PrevFloat32(rayT * 0.999f - NextFloat32(tminv));
    cnt[1]++;
    // "first accepted hit, visible iff its ID is the target's" (RayQuery.hlsli:372-405) depends on the traversal order; pinned
    // order-independently: triangles carrying the target's ID are not occluders, any other hit in the shortened segment is
    q.want = true; q.o = o; q.d = wi; q.tmin = tminv; q.tmax = tmax; q.mask = ZR_SUBGROUP_NON_EMISSIVE; q.anyHit = true; q.filterID = true; q.ignoreID = triID;
    return q;
}
ZR_HD bool SegmentVisible(const TraceReq& q, const RawHit& h) { return q.want && h.tri == kInvalidTri; }
ZR_HD bool VisibilitySegmentApprox(const Globals& g, V3 origin, V3 wi, float rayT, V3 normal, uint32_t triID, bool transmissive)
{
    const TraceReq q = SegmentApproxReq(g.cnt, origin, wi, rayT, normal, triID, transmissive);
    return SegmentVisible(q, TraceInline<true, true>(*g.sc, q, g.stack));
}

ZR_HD bool IsSpecular(const Surface& s) { return s.GlossSpecular() && (s.metallic || s.specTr) && (!s.Coated() || s.CoatSpecular()); }

// ReSTIR_PT_NEE.hlsli:134-207, cut at its FindClosest: the BSDF sample ...
ZR_HD void NEE_Bsdf_Pre(const Globals& g, V3 normal, const Surface& surface, int nextBounce, BsdfSample& bs, Rng& rng)
{
    bs = InitBsdfSample();
    if (nextBounce <= g.maxNumBounces) { ZR_PROF_SCOPE(ZRP_BSDF); bs = SampleBSDF(g.sc->rho, normal, surface, rng); }
}
// ... and what its ray found
ZR_HD Direct NEE_Bsdf_Post(const Globals& g, V3 pos, const Surface& surface, int nextBounce, BsdfSample& bs, const HitEm& hitInfo)
{
    const SceneView& sc = *g.sc;
    Direct ret = InitDirect();
    const bool specular = IsSpecular(surface);
    const int numLightSamples = specular ? 0 : 1;
    const float wiPdf = bs.pdf;
    const V3 wi = bs.wi;
    const V3 f = bs.f;
    if (hitInfo.emissiveTriIdx != 0xffffffffu)
    {
        const zr_emissive_triangle em = sc.emissives[hitInfo.emissiveTriIdx];
        const V3 le = EmLe(sc, em, v2(hitInfo.bu, hitInfo.bv));
        const V3 vtx0 = v3p(em.vtx0), vtx1 = EmV1(em), vtx2 = EmV2(em);
        V3 ln = cross(vtx1 - vtx0, vtx2 - vtx0);
        float twoArea = length(ln);
        ln = dot(ln, ln) == 0 ? v3(0.0f) : ln / twoArea;
        ln = EmDoubleSided(em) && (dot(-wi, ln) < 0) ? -ln : ln;
        float lightPdf = 0;
        if (!specular)
        {
            const float lightSourcePdf = numLightSamples > 0 ? sc.alias[hitInfo.emissiveTriIdx].cached_p_orig : 0;
            lightPdf = twoArea > 0 ? lightSourcePdf * (2.0f / twoArea) : 0;
        }
        float dwdA = zr_saturate(dot(ln, -wi)) / (hitInfo.t * hitInfo.t);
        float wiPdf_area = wiPdf * dwdA;
        V3 ld = le * f * dwdA;
        ret.ld = specular ? (wiPdf_area > 0 ? ld / wiPdf_area : v3(0.0f)) : PowerHeuristic(wiPdf_area, lightPdf, ld, 1.0f, 1.0f);
        ret.le = le; ret.wi = wi; ret.pdf_solidAngle = wiPdf; ret.dwdA = dwdA; ret.ID = em.id;
        ret.pos = mad(hitInfo.t, wi, pos); ret.normal = ln; ret.pdf_light = lightPdf; ret.lobe = bs.lobe;
        ret.lt = LT_EMISSIVE; ret.twoSided = EmDoubleSided(em);
    }
    if (nextBounce >= g.maxNumBounces) bs.bsdfOverPdf = v3(0.0f);
    return ret;
}
ZR_HD Direct NEE_Bsdf(const Globals& g, V3 pos, V3 normal, const Surface& surface, int nextBounce, BsdfSample& bs, HitEm& hitInfo, Rng& rng)
{
    NEE_Bsdf_Pre(g, normal, surface, nextBounce, bs, rng);
    hitInfo = FindClosestEm(g, pos, normal, bs.wi, surface.Transmissive());
    return NEE_Bsdf_Post(g, pos, surface, nextBounce, bs, hitInfo);
}

// ReSTIR_PT_NEE.hlsli:209-284 (alias-table branch), cut at its Visibility_Segment: the light sample and its unshadowed contribution ...
struct NeeEmState { Direct ret; Surface surface; V3 ld, le, ln, lpos, wi, fz; float lightPdf, t, dwdA; uint32_t lightID; bool twoSided, facing; };      // fz: Unified(surface).f towards the light
ZR_HD TraceReq NEE_Emissive_Pre(const Globals& g, V3 pos, V3 normal, const Surface& surfaceIn, Rng& rng, NeeEmState& S)
{
    const SceneView& sc = *g.sc;
    S.surface = surfaceIn;
    S.ret = InitDirect();
    S.ret.lt = LT_EMISSIVE; S.ret.lobe = LOBE_ALL;
    V3 lpos, ln, le; float lightPdf; uint32_t lightID; bool twoSided;
    if (g.presampled)       // USE_PRESAMPLED_SETS, ReSTIR_PT_NEE.hlsli:217-236
    {
        PresampledLight pl = SamplePresampledSet(sc, g.sampleSetIdx, pos, rng);
        lpos = pl.pos; ln = pl.normal; le = pl.le; lightPdf = pl.pdf; lightID = pl.ID; twoSided = pl.twoSided;
        rng.Uniform(); rng.Uniform(); rng.Uniform();       // "deterministic RNG state regardless of USE_PRESAMPLED_SETS"
    }
    else
    {
        // Light::AliasTableSample::get, LightSource.hlsli:72-98
        uint32_t u0 = rng.UniformUintBounded(g.numEmissives);
        const zr_alias_entry ae = sc.alias[u0];
        uint32_t lidx; float lpdfSrc;
        if (rng.Uniform() < ae.p_curr) { lpdfSrc = ae.cached_p_orig; lidx = u0; }
        else { lpdfSrc = ae.cached_p_alias; lidx = ae.alias; }
        const zr_emissive_triangle em = sc.emissives[lidx];
        // Light::EmissiveTriSample::get, LightSource.hlsli:109-137
        V2 u = rng.Uniform2D();
        V2 bary = UniformSampleTriangle(u);
        const V3 vtx0 = v3p(em.vtx0), vtx1 = EmV1(em), vtx2 = EmV2(em);
        lpos = (1.0f - bary.x - bary.y) * vtx0 + bary.x * vtx1 + bary.y * vtx2;
        ln = cross(vtx1 - vtx0, vtx2 - vtx0);
        bool normalIs0 = dot(ln, ln) == 0;
        float twoArea = length(ln);
        float lpdfPos = normalIs0 ? 0.0f : 2.0f / twoArea;
        ln = normalIs0 ? ln : ln / twoArea;
        ln = EmDoubleSided(em) && dot(pos - lpos, ln) < 0 ? -ln : ln;
        le = EmLe(sc, em, bary);
        lightPdf = lpdfSrc * lpdfPos;
        lightID = em.id; twoSided = EmDoubleSided(em);
    }
    const float t = length(lpos - pos);
    const V3 wi = (lpos - pos) / t;
    S.lpos = lpos; S.ln = ln; S.le = le; S.lightPdf = lightPdf; S.lightID = lightID; S.twoSided = twoSided; S.t = t; S.wi = wi;
    S.facing = (dot(ln, -wi) > 0) && (t > 0);
    S.ld = v3(0.0f); S.dwdA = 0; S.fz = v3(0.0f);
    if (!S.facing) return NoTraceReq();
    S.dwdA = zr_saturate(dot(ln, -wi)) / (t * t);
    S.surface.SetWi(wi, normal);
    S.fz = Unified(sc.rho, S.surface).f;
    S.ld = le * S.fz * S.dwdA;
    if (!(dot(S.ld, S.ld) > 0)) return NoTraceReq();
    return SegmentApproxReq(g.cnt, pos, wi, t, normal, lightID, S.surface.Transmissive());
}
// ... and the rest once the segment's visibility is known (`visible` is only read when the unshadowed contribution was non-zero)
ZR_HD Direct NEE_Emissive_Post(const Globals& g, V3 normal, NeeEmState& S, bool visible, Rng& rng)
{
    if (!S.facing) return S.ret;
    V3 ld = S.ld;
    if (dot(ld, ld) > 0) ld = ld * (visible ? 1.0f : 0.0f);
    float bsdfPdf = 0;
    if (dot(ld, ld) > 0)
    {
        { ZR_PROF_SCOPE(ZRP_NEE); bsdfPdf = BSDFSamplerPdf_AtZ(g.sc->rho, normal, S.surface, S.wi, S.fz, rng); }
        bsdfPdf *= S.dwdA;
    }
    Direct& ret = S.ret;
    ret.ld = PowerHeuristic(S.lightPdf, bsdfPdf, ld, 1.0f, 1.0f);
    ret.le = S.le; ret.wi = S.wi; ret.pdf_solidAngle = S.lightPdf / S.dwdA; ret.dwdA = S.dwdA; ret.ID = S.lightID;
    ret.pos = S.lpos; ret.normal = S.ln; ret.pdf_light = S.lightPdf; ret.twoSided = S.twoSided;
    return ret;
}
ZR_HD Direct NEE_Emissive(const Globals& g, V3 pos, V3 normal, const Surface& surface, Rng& rng)
{
    NeeEmState S;
    const TraceReq q = NEE_Emissive_Pre(g, pos, normal, surface, rng, S);
    return NEE_Emissive_Post(g, normal, S, SegmentVisible(q, TraceInline<true, true>(*g.sc, q, g.stack)), rng);
}

// ---- the same queries and NEE functions in their FUSED form (one function, traversal in the middle): what the inline megakernel
// k_rpt_pathtrace compiles.  Statement for statement the cut forms above (the GPU parity tests run both: ZR_K11=inline / pool, and the host
// executor runs the cut forms); kept because the cut costs the megakernel registers -- +38 % spill traffic, K11 +3 % on the Cornell box and
// +9 % on the atrium (profiles/r03_*).
// Hit_Emissive::FindClosest, RayQuery.hlsli:146-207
ZR_HD HitEm FindClosestEm_Fused(const Globals& g, V3 pos, V3 normal, V3 wi, bool transmissive)
{
    HitEm r; r.hit = false; r.emissiveTriIdx = 0xffffffffu; r.t = 0; r.mesh = 0; r.prim = 0; r.bu = 0; r.bv = 0;
    F4 ro, rd;
    if (!MakeClosestRay(pos, normal, wi, transmissive, true, &ro, &rd)) return r;
    g.cnt[0]++;
    RawHit h = Traverse<false>(*g.sc, xyz(ro), xyz(rd), ro.w, rd.w, ZR_SUBGROUP_ALL, g.stack);
    if (h.tri == kInvalidTri) return r;
    const TriMeta tm = g.sc->triMeta[h.tri];
    r.hit = true; r.t = h.t; r.bu = h.u; r.bv = h.v; r.mesh = tm.mesh; r.prim = tm.prim;
    const uint32_t base = g.sc->instances[tm.mesh].base_emissive_tri_offset;
    if (base != 0xffffffffu) r.emissiveTriIdx = base + tm.prim;
    return r;
}

// Visibility_Segment with APPROXIMATE_EMISSIVE_SHADOW_RAY == 1 (RayQuery.hlsli:337-406)
ZR_HD bool VisibilitySegmentApprox_Fused(const Globals& g, V3 origin, V3 wi, float rayT, V3 normal, uint32_t triID, bool transmissive)
{
    if (triID == 0xffffffffu) return false;
    if (rayT < 1e-6f) return false;
    float ndotwi = dot(normal, wi);
    if (ndotwi == 0) return false;
    if (ndotwi < 0)
    {
        if (transmissive) normal = normal * -1.0f;
        else return false;
    }
    const V3 o = OffsetRayRTG(origin, normal);
    const float tminv = 3e-6f;
    const float tmax = PrevFloat32(rayT * 0.999f - NextFloat32(tminv));
    g.cnt[1]++;
    // "first accepted hit, visible iff its ID is the target's" (RayQuery.hlsli:372-405) depends on the traversal order; pinned
    // order-independently: triangles carrying the target's ID are not occluders, any other hit in the shortened segment is
    RawHit h = Traverse<true>(*g.sc, o, wi, tminv, tmax, ZR_SUBGROUP_NON_EMISSIVE, g.stack, true, triID);
    return h.tri == kInvalidTri;
}


// ReSTIR_PT_NEE.hlsli:134-207
ZR_HD Direct NEE_Bsdf_Fused(const Globals& g, V3 pos, V3 normal, const Surface& surface, int nextBounce, BsdfSample& bs, HitEm& hitInfo, Rng& rng)
{
    const SceneView& sc = *g.sc;
    Direct ret = InitDirect();
    const bool specular = IsSpecular(surface);
    const int numLightSamples = specular ? 0 : 1;
    bs = InitBsdfSample();
    if (nextBounce <= g.maxNumBounces) { ZR_PROF_SCOPE(ZRP_BSDF); bs = SampleBSDF(sc.rho, normal, surface, rng); }
    const float wiPdf = bs.pdf;
    const V3 wi = bs.wi;
    const V3 f = bs.f;
    hitInfo = FindClosestEm_Fused(g, pos, normal, wi, surface.Transmissive());
    if (hitInfo.emissiveTriIdx != 0xffffffffu)
    {
        const zr_emissive_triangle em = sc.emissives[hitInfo.emissiveTriIdx];
        const V3 le = EmLe(sc, em, v2(hitInfo.bu, hitInfo.bv));
        const V3 vtx0 = v3p(em.vtx0), vtx1 = EmV1(em), vtx2 = EmV2(em);
        V3 ln = cross(vtx1 - vtx0, vtx2 - vtx0);
        float twoArea = length(ln);
        ln = dot(ln, ln) == 0 ? v3(0.0f) : ln / twoArea;
        ln = EmDoubleSided(em) && (dot(-wi, ln) < 0) ? -ln : ln;
        float lightPdf = 0;
        if (!specular)
        {
            const float lightSourcePdf = numLightSamples > 0 ? sc.alias[hitInfo.emissiveTriIdx].cached_p_orig : 0;
            lightPdf = twoArea > 0 ? lightSourcePdf * (2.0f / twoArea) : 0;
        }
        float dwdA = zr_saturate(dot(ln, -wi)) / (hitInfo.t * hitInfo.t);
        float wiPdf_area = wiPdf * dwdA;
        V3 ld = le * f * dwdA;
        ret.ld = specular ? (wiPdf_area > 0 ? ld / wiPdf_area : v3(0.0f)) : PowerHeuristic(wiPdf_area, lightPdf, ld, 1.0f, 1.0f);
        ret.le = le; ret.wi = wi; ret.pdf_solidAngle = wiPdf; ret.dwdA = dwdA; ret.ID = em.id;
        ret.pos = mad(hitInfo.t, wi, pos); ret.normal = ln; ret.pdf_light = lightPdf; ret.lobe = bs.lobe;
        ret.lt = LT_EMISSIVE; ret.twoSided = EmDoubleSided(em);
    }
    if (nextBounce >= g.maxNumBounces) bs.bsdfOverPdf = v3(0.0f);
    return ret;
}


// ReSTIR_PT_NEE.hlsli:209-284 (alias-table branch)
ZR_HD Direct NEE_Emissive_Fused(const Globals& g, V3 pos, V3 normal, Surface surface, Rng& rng)
{
    const SceneView& sc = *g.sc;
    Direct ret = InitDirect();
    ret.lt = LT_EMISSIVE; ret.lobe = LOBE_ALL;
    V3 lpos, ln, le; float lightPdf; uint32_t lightID; bool twoSided;
    if (g.presampled)       // USE_PRESAMPLED_SETS, ReSTIR_PT_NEE.hlsli:217-236
    {
        PresampledLight pl = SamplePresampledSet(sc, g.sampleSetIdx, pos, rng);
        lpos = pl.pos; ln = pl.normal; le = pl.le; lightPdf = pl.pdf; lightID = pl.ID; twoSided = pl.twoSided;
        rng.Uniform(); rng.Uniform(); rng.Uniform();       // "deterministic RNG state regardless of USE_PRESAMPLED_SETS"
    }
    else
    {
        // Light::AliasTableSample::get, LightSource.hlsli:72-98
        uint32_t u0 = rng.UniformUintBounded(g.numEmissives);
        const zr_alias_entry ae = sc.alias[u0];
        uint32_t lidx; float lpdfSrc;
        if (rng.Uniform() < ae.p_curr) { lpdfSrc = ae.cached_p_orig; lidx = u0; }
        else { lpdfSrc = ae.cached_p_alias; lidx = ae.alias; }
        const zr_emissive_triangle em = sc.emissives[lidx];
        // Light::EmissiveTriSample::get, LightSource.hlsli:109-137
        V2 u = rng.Uniform2D();
        V2 bary = UniformSampleTriangle(u);
        const V3 vtx0 = v3p(em.vtx0), vtx1 = EmV1(em), vtx2 = EmV2(em);
        lpos = (1.0f - bary.x - bary.y) * vtx0 + bary.x * vtx1 + bary.y * vtx2;
        ln = cross(vtx1 - vtx0, vtx2 - vtx0);
        bool normalIs0 = dot(ln, ln) == 0;
        float twoArea = length(ln);
        float lpdfPos = normalIs0 ? 0.0f : 2.0f / twoArea;
        ln = normalIs0 ? ln : ln / twoArea;
        ln = EmDoubleSided(em) && dot(pos - lpos, ln) < 0 ? -ln : ln;
        le = EmLe(sc, em, bary);
        lightPdf = lpdfSrc * lpdfPos;
        lightID = em.id; twoSided = EmDoubleSided(em);
    }
    const float t = length(lpos - pos);
    const V3 wi = (lpos - pos) / t;
    if ((dot(ln, -wi) > 0) && (t > 0))
    {
        const float dwdA = zr_saturate(dot(ln, -wi)) / (t * t);
        surface.SetWi(wi, normal);
        const V3 fz = Unified(sc.rho, surface).f;
        V3 ld = le * fz * dwdA;
        if (dot(ld, ld) > 0)
            ld = ld * (VisibilitySegmentApprox_Fused(g, pos, wi, t, normal, lightID, surface.Transmissive()) ? 1.0f : 0.0f);
        float bsdfPdf = 0;
        if (dot(ld, ld) > 0)
        {
            { ZR_PROF_SCOPE(ZRP_NEE); bsdfPdf = BSDFSamplerPdf_AtZ(sc.rho, normal, surface, wi, fz, rng); }
            bsdfPdf *= dwdA;
        }
        ret.ld = PowerHeuristic(lightPdf, bsdfPdf, ld, 1.0f, 1.0f);
        ret.le = le; ret.wi = wi; ret.pdf_solidAngle = lightPdf / dwdA; ret.dwdA = dwdA; ret.ID = lightID;
        ret.pos = lpos; ret.normal = ln; ret.pdf_light = lightPdf; ret.twoSided = twoSided;
    }
    return ret;
}


// ReSTIR_PT_NEE.hlsli:306-391
ZR_HD Direct EvalDirect_Case2(const Globals& g, V3 normal, Surface surface, V3 wi, V3 le, float dwdA, float lightPdf, uint32_t lobe,
    Rng& rngReplay, Rng& rngNEE)
{
    const RhoView& rho = g.sc->rho;
    surface.SetWi(wi, normal);
    const V3 fz = Unified(rho, surface).f;
    V3 ld = le * fz * dwdA;
    Direct ret = InitDirect();
    if (dot(ld, ld) == 0) return ret;
    if (lobe == LOBE_ALL)
    {
        rngNEE.Uniform(); rngNEE.Uniform(); rngNEE.Uniform(); rngNEE.Uniform();
        float bsdfPdf = BSDFSamplerPdf_AtZ(rho, normal, surface, wi, fz, rngNEE);
        ret.ld = PowerHeuristic(lightPdf, bsdfPdf * dwdA, ld, 1.0f, 1.0f);
        ret.pdf_solidAngle = 1.0f;
    }
    else
    {
        SamplerEval e = EvalBSDFSampler(rho, normal, surface, wi, lobe, rngReplay);
        const bool specular = IsSpecular(surface);
        float bsdfPdf_area = e.pdf * dwdA;
        ret.ld = specular ? (bsdfPdf_area > 0 ? ld / bsdfPdf_area : v3(0.0f)) : PowerHeuristic(bsdfPdf_area, lightPdf, ld, 1.0f, 1.0f);
        ret.pdf_solidAngle = e.pdf;
    }
    return ret;
}
ZR_HD Direct EvalDirect_Case3(const Globals& g, V3 pos, V3 normal, Surface surface, V3 wi, float t, V3 le, V3 lightNormal, float lightPdf,
    uint32_t lightID, bool twoSided, uint32_t lobe, Rng& rngReplay, Rng& rngNEE)
{
    const RhoView& rho = g.sc->rho;
    float wiDotLN = dot(lightNormal, -wi);
    float dwdA = zr_abs(wiDotLN) / (t * t);
    surface.SetWi(wi, normal);
    V3 fz = v3(0.0f), ld = v3(0.0f);
    if ((wiDotLN > 0) || twoSided) { fz = Unified(rho, surface).f; ld = le * fz * dwdA; }
    if (dot(ld, ld) > 0)
        ld = ld * (VisibilitySegmentApprox(g, pos, wi, t, normal, lightID, surface.Transmissive()) ? 1.0f : 0.0f);
    Direct ret = InitDirect();
    if (dot(ld, ld) == 0) return ret;
    if (lobe == LOBE_ALL)
    {
        rngNEE.Uniform(); rngNEE.Uniform(); rngNEE.Uniform(); rngNEE.Uniform();
        float bsdfPdf = BSDFSamplerPdf_AtZ(rho, normal, surface, wi, fz, rngNEE);
        ret.ld = PowerHeuristic(lightPdf, bsdfPdf * dwdA, ld, 1.0f, 1.0f);
        ret.pdf_solidAngle = 1.0f;
    }
    else
    {
        SamplerEval e = EvalBSDFSampler(rho, normal, surface, wi, lobe, rngReplay);
        const bool specular = IsSpecular(surface);
        float bsdfPdf_area = e.pdf * dwdA;
        ret.ld = specular ? (bsdfPdf_area > 0 ? ld / bsdfPdf_area : v3(0.0f)) : PowerHeuristic(bsdfPdf_area, lightPdf, ld, 1.0f, 1.0f);
        ret.pdf_solidAngle = bsdfPdf_area;
    }
    return ret;
}

// ---- ReSTIR_PT_PathTrace.hlsl:36-192
// RtRayQuery::Visibility_Ray (RayQuery.hlsli:302-334), traced in place
ZR_HD bool VisibilityRay(const Globals& g, V3 origin, V3 wi, V3 normal, bool transmissive)
{
    F4 ro, rd;
    if (!MakeVisibilityRay(origin, wi, normal, transmissive, &ro, &rd)) return false;
    g.cnt[1]++;
    RawHit h = Traverse<true>(*g.sc, xyz(ro), xyz(rd), ro.w, rd.w, ZR_SUBGROUP_ALL, g.stack);
    return h.tri == kInvalidTri;
}

// RPT_Util::NEE_NonEmissive, ReSTIR_PT_NEE.hlsli:10-132 (SKY_SAMPLING_PREFER_PERFORMANCE == 1): one RIS over {sun, cosine sky sample,
// BSDF (no-diffuse sampler) sky sample} with Le_Sky as the lobe-RIS target, then a single visibility ray
ZR_HD Direct NEE_NonEmissive(const Globals& g, V3 pos, V3 normal, Surface surface, Rng& rng)
{
    const SceneView& sc = *g.sc; const zr_frame_constants& fr = *g.frame; const RhoView& rho = sc.rho;
    Direct ret = InitDirect();
    ret.dwdA = 1;
    SkyIncidentRadiance leFunc; leFunc.lut = sc.sky;
    const bool specular = IsSpecular(surface);
    float w_sum = 0;
    V3 target_z = v3(0.0f);
    const V2 u_wrs = rng.Uniform2D();
    const V2 u_d = rng.Uniform2D();
    const V2 u_c = rng.Uniform2D();
    const V2 u_g = rng.Uniform2D();
    const float u_wrs_b0 = rng.Uniform();
    const float u_wrs_b1 = rng.Uniform();
    {
        const V3 wi_s = -v3p(fr.sun_dir);
        const bool visible = (wi_s.y > 0) && ((dot(wi_s, normal) > 0) || surface.Transmissive());
        float pdf_b = 0, pdf_d = 0;
        if (visible)
        {
            surface.SetWi(wi_s, normal);
            target_z = Le_Sun(pos, fr) * Unified(rho, surface).f;
            const float ndotWi = dot(wi_s, normal);
            pdf_b = (ndotWi < 0) && surface.ThinWalled() ? 0 : BSDFSamplerPdf_NoDiffuse(rho, normal, surface, wi_s, leFunc);
            pdf_d = (!specular ? 1.0f : 0.0f) * zr_abs(ndotWi) * ZR_ONE_OVER_PI;
            pdf_d *= surface.ThinWalled() ? 0.5f : (ndotWi > 0 ? 1.0f : 0.0f);
        }
        w_sum = BalanceHeuristic3(1, pdf_b, pdf_d, Luminance(target_z));
        ret.lt = LT_SUN; ret.lobe = LOBE_ALL; ret.wi = wi_s;
    }
    if (!specular)
    {
        float pdf_e;
        V3 wi_e = SampleDiffuse(normal, u_d, &pdf_e);
        if (surface.ThinWalled()) { wi_e = u_wrs_b1 > 0.5f ? -wi_e : wi_e; pdf_e *= 0.5f; }
        surface.SetWi(wi_e, normal);
        const V3 target = leFunc(wi_e) * Unified(rho, surface).f;
        const float pdf_b = !surface.reflection && surface.ThinWalled() ? 0 : BSDFSamplerPdf_NoDiffuse(rho, normal, surface, wi_e, leFunc);
        const float denom = pdf_e + pdf_b;
        const float w_e = denom == 0 ? 0.0f : Luminance(target) / denom;
        w_sum += w_e;
        if ((w_sum > 0) && (u_wrs.y < (w_e / w_sum))) { ret.lt = LT_SKY; ret.lobe = LOBE_ALL; ret.wi = wi_e; target_z = target; }
    }
    {
        const BsdfSample bs = SampleBSDF_NoDiffuse(rho, normal, surface, u_c, u_g, u_wrs_b0, u_wrs_b1, leFunc);
        const float ndotwi = dot(bs.wi, normal);
        float pdf_e = (!specular ? 1.0f : 0.0f) * zr_abs(ndotwi) * ZR_ONE_OVER_PI;
        pdf_e *= surface.ThinWalled() ? 0.5f : (ndotwi > 0 ? 1.0f : 0.0f);
        const float denom = bs.pdf + pdf_e;
        const float w_b = denom == 0 ? 0.0f : Luminance(bs.f) / denom;
        w_sum += w_b;
        if ((w_sum > 0) && (u_wrs.x < (w_b / w_sum))) { ret.lt = LT_SKY; ret.lobe = bs.lobe; ret.wi = bs.wi; target_z = bs.f; }
    }
    const float targetLum = Luminance(target_z);
    ret.ld = targetLum > 0 ? target_z * w_sum / targetLum : v3(0.0f);
    ret.pdf_solidAngle = w_sum > 0 ? targetLum / w_sum : 0;
    if (dot(ret.ld, ret.ld) > 0) ret.ld = ret.ld * (VisibilityRay(g, pos, ret.wi, normal, surface.Transmissive()) ? 1.0f : 0.0f);
    return ret;
}

struct PrevHit { float alpha_lobe; V3 wi; float pdf; uint32_t lobe; };

ZR_HD void MaybeSetCase2OrCase3(const Globals& g, int pathVertex, V3 pos, V3 normal, float t, uint32_t ID, uint32_t meshIdx,
    const Surface& surface, const PrevHit& prevHit, const Direct& ls, uint32_t seed_nee, Reconnection& rc)
{
    const float alpha_direct = LobeAlpha(surface, ls.lobe);
    if (rc.Empty() && CanReconnect(prevHit.alpha_lobe, alpha_direct, prevHit.lobe, ls.lobe, g.alpha_min))
        rc.SetCase2(pathVertex, pos, t, normal, ID, meshIdx, prevHit.wi, prevHit.lobe, prevHit.pdf, ls.wi, ls.lobe, ls.pdf_solidAngle, ls.lt,
            ls.pdf_light, ls.le, seed_nee, ls.dwdA);
    if (rc.Empty() && (alpha_direct >= g.alpha_min))
        rc.SetCase3(pathVertex + 1, ls.pos, ls.lt, ls.lobe, ls.ID, ls.le, ls.normal, ls.pdf_solidAngle, ls.pdf_light, ls.dwdA, ls.wi, ls.twoSided, seed_nee);
}

ZR_HD void EstimateDirectAndUpdateRC(const Globals& g, int pathVertex, V3 pos, const HitInfo& hit, const Surface& surface, const PrevHit& prevHit,
    V3 throughput, V3 throughput_k, V3& li, BsdfSample& bs, HitEm& nextHit, Reconnection& rc, Reservoir& r, Rng& rngNEE, Rng& rngReplay)
{
    if (!g.emissive)      // EstimateDirectAndUpdateRC<false>, ReSTIR_PT_PathTrace.hlsl:172-191
    {
        const uint32_t seed_nee = rngNEE.s;
        Direct ls = NEE_NonEmissive(g, pos, hit.normal, surface, rngNEE);
        const V3 fOverPdf = throughput * ls.ld;
        li = li + fOverPdf;
        rc.L = RoundHalf3(ls.ld * throughput_k);
        MaybeSetCase2OrCase3(g, pathVertex, pos, hit.normal, hit.t, hit.ID, hit.meshIdx, surface, prevHit, ls, seed_nee, rc);
        r.Update(Luminance(fOverPdf), fOverPdf, rc, rngNEE);
        return;
    }
    BsdfSample nbs;
    int nextBounce = pathVertex - 1;
    Direct ls_b;
    { ZR_PROF_SCOPE(ZRP_MISC4); ls_b = NEE_Bsdf(g, pos, hit.normal, surface, nextBounce, nbs, nextHit, rngReplay); }
    if (nextHit.emissiveTriIdx != 0xffffffffu)
    {
        const V3 fOverPdf = throughput * ls_b.ld;
        li = li + fOverPdf;
        rc.L = RoundHalf3(ls_b.ld * throughput_k);
        MaybeSetCase2OrCase3(g, pathVertex, pos, hit.normal, hit.t, hit.ID, hit.meshIdx, surface, prevHit, ls_b, 0, rc);
        r.Update(Luminance(fOverPdf), fOverPdf, rc, rngNEE);
    }
    if (!IsSpecular(surface))
    {
        const uint32_t seed_nee = rngNEE.s;
        Direct ls;
        { ZR_PROF_SCOPE(ZRP_MISC3); ls = NEE_Emissive(g, pos, hit.normal, surface, rngNEE); }
        const V3 fOverPdf = throughput * ls.ld;
        li = li + fOverPdf;
        if (rc.IsCase2() || rc.IsCase3()) rc.Clear();
        rc.L = RoundHalf3(ls.ld * throughput_k);
        MaybeSetCase2OrCase3(g, pathVertex, pos, hit.normal, hit.t, hit.ID, hit.meshIdx, surface, prevHit, ls, seed_nee, rc);
        r.Update(Luminance(fOverPdf), fOverPdf, rc, rngNEE);
    }
    bs = nbs;
}

// fused form (see NEE_Bsdf_Fused)
ZR_HD void EstimateDirectAndUpdateRC_Fused(const Globals& g, int pathVertex, V3 pos, const HitInfo& hit, const Surface& surface, const PrevHit& prevHit,
    V3 throughput, V3 throughput_k, V3& li, BsdfSample& bs, HitEm& nextHit, Reconnection& rc, Reservoir& r, Rng& rngNEE, Rng& rngReplay)
{
    if (!g.emissive)      // EstimateDirectAndUpdateRC<false>, ReSTIR_PT_PathTrace.hlsl:172-191
    {
        const uint32_t seed_nee = rngNEE.s;
        Direct ls = NEE_NonEmissive(g, pos, hit.normal, surface, rngNEE);
        const V3 fOverPdf = throughput * ls.ld;
        li = li + fOverPdf;
        rc.L = RoundHalf3(ls.ld * throughput_k);
        MaybeSetCase2OrCase3(g, pathVertex, pos, hit.normal, hit.t, hit.ID, hit.meshIdx, surface, prevHit, ls, seed_nee, rc);
        r.Update(Luminance(fOverPdf), fOverPdf, rc, rngNEE);
        return;
    }
    BsdfSample nbs;
    int nextBounce = pathVertex - 1;
    Direct ls_b;
    { ZR_PROF_SCOPE(ZRP_MISC4); ls_b = NEE_Bsdf_Fused(g, pos, hit.normal, surface, nextBounce, nbs, nextHit, rngReplay); }
    if (nextHit.emissiveTriIdx != 0xffffffffu)
    {
        const V3 fOverPdf = throughput * ls_b.ld;
        li = li + fOverPdf;
        rc.L = RoundHalf3(ls_b.ld * throughput_k);
        MaybeSetCase2OrCase3(g, pathVertex, pos, hit.normal, hit.t, hit.ID, hit.meshIdx, surface, prevHit, ls_b, 0, rc);
        r.Update(Luminance(fOverPdf), fOverPdf, rc, rngNEE);
    }
    if (!IsSpecular(surface))
    {
        const uint32_t seed_nee = rngNEE.s;
        Direct ls;
        { ZR_PROF_SCOPE(ZRP_MISC3); ls = NEE_Emissive_Fused(g, pos, hit.normal, surface, rngNEE); }
        const V3 fOverPdf = throughput * ls.ld;
        li = li + fOverPdf;
        if (rc.IsCase2() || rc.IsCase3()) rc.Clear();
        rc.L = RoundHalf3(ls.ld * throughput_k);
        MaybeSetCase2OrCase3(g, pathVertex, pos, hit.normal, hit.t, hit.ID, hit.meshIdx, surface, prevHit, ls, seed_nee, rc);
        r.Update(Luminance(fOverPdf), fOverPdf, rc, rngNEE);
    }
    bs = nbs;
}


// EstimateDirectAndUpdateRC<true> (ReSTIR_PT_PathTrace.hlsl:77-170) cut at its two BVH queries -- the BSDF ray of NEE_Bsdf and the light
// segment of NEE_Emissive -- for kernels that trace elsewhere (zr_kernels.h: RptPathtraceBodyCoop).  Same statements in the same order as
// the function above; `M` carries what lives across the cuts.
struct PtMid { BsdfSample nbs; NeeEmState nee; bool doNee; uint32_t seed_nee; TraceReq q1, q2; };
ZR_HD void EstimateDirectEm_Pre(const Globals& g, int pathVertex, V3 pos, const HitInfo& hit, const Surface& surface, Rng& rngReplay, PtMid& M)
{
    const int nextBounce = pathVertex - 1;
    NEE_Bsdf_Pre(g, hit.normal, surface, nextBounce, M.nbs, rngReplay);
    M.q1 = ClosestEmReq(g.cnt, pos, hit.normal, M.nbs.wi, surface.Transmissive());
}
ZR_HD void EstimateDirectEm_Mid(const Globals& g, int pathVertex, V3 pos, const HitInfo& hit, const Surface& surface, const PrevHit& prevHit,
    V3 throughput, V3 throughput_k, V3& li, HitEm& nextHit, Reconnection& rc, Reservoir& r, Rng& rngNEE, PtMid& M, const RawHit& h1)
{
    const int nextBounce = pathVertex - 1;
    nextHit = ClosestEmResult(*g.sc, M.q1, h1);
    const Direct ls_b = NEE_Bsdf_Post(g, pos, surface, nextBounce, M.nbs, nextHit);
    if (nextHit.emissiveTriIdx != 0xffffffffu)
    {
        const V3 fOverPdf = throughput * ls_b.ld;
        li = li + fOverPdf;
        rc.L = RoundHalf3(ls_b.ld * throughput_k);
        MaybeSetCase2OrCase3(g, pathVertex, pos, hit.normal, hit.t, hit.ID, hit.meshIdx, surface, prevHit, ls_b, 0, rc);
        r.Update(Luminance(fOverPdf), fOverPdf, rc, rngNEE);
    }
    M.doNee = !IsSpecular(surface);
    M.q2 = NoTraceReq();
    if (M.doNee)
    {
        M.seed_nee = rngNEE.s;
        M.q2 = NEE_Emissive_Pre(g, pos, hit.normal, surface, rngNEE, M.nee);
    }
}
ZR_HD void EstimateDirectEm_Post(const Globals& g, int pathVertex, V3 pos, const HitInfo& hit, const Surface& surface, const PrevHit& prevHit,
    V3 throughput, V3 throughput_k, V3& li, BsdfSample& bs, Reconnection& rc, Reservoir& r, Rng& rngNEE, PtMid& M, const RawHit& h2)
{
    if (M.doNee)
    {
        const Direct ls = NEE_Emissive_Post(g, hit.normal, M.nee, SegmentVisible(M.q2, h2), rngNEE);
        const V3 fOverPdf = throughput * ls.ld;
        li = li + fOverPdf;
        if (rc.IsCase2() || rc.IsCase3()) rc.Clear();
        rc.L = RoundHalf3(ls.ld * throughput_k);
        MaybeSetCase2OrCase3(g, pathVertex, pos, hit.normal, hit.t, hit.ID, hit.meshIdx, surface, prevHit, ls, M.seed_nee, rc);
        r.Update(Luminance(fOverPdf), fOverPdf, rc, rngNEE);
    }
    bs = M.nbs;
}

// ---- G-buffer reads
struct GFlags { bool metallic, transmissive, emissive, invalid, trDepthGt0, subsurface, coated; };
ZR_HD GFlags DecodeFlags(uint16_t mrp)
{
    const uint32_t v = (uint32_t)zr_fma(zr_div255((float)(mrp & 0xff)), 255.0f, 0.5f);
    GFlags f; f.transmissive = v & 1; f.emissive = v & 2; f.invalid = v & 4; f.trDepthGt0 = v & 8; f.subsurface = v & 16; f.coated = v & 32; f.metallic = v & 128;
    return f;
}
ZR_HD float RoughnessOf(uint16_t mrp) { return zr_div255((float)(mrp >> 8)); }
ZR_HD V2 DecodeMotion(uint32_t m)
{
    float fx = (float)(int16_t)(uint16_t)(m & 0xffff) / 32767.0f, fy = (float)(int16_t)(uint16_t)(m >> 16) / 32767.0f;
    return v2(fx < -1.0f ? -1.0f : fx, fy < -1.0f ? -1.0f : fy);
}

// plane index of the global pixel (x, y): planes cover the extended tile [gb.x0, gb.x0 + gb.w) x [gb.y0, gb.y0 + gb.h)
ZR_HD size_t Pix(const GBuf& gb, uint32_t x, uint32_t y) { return (size_t)(y - gb.y0) * gb.w + (x - gb.x0); }
ZR_HD bool InPlanes(const GBuf& gb, int x, int y) { return x >= (int)gb.x0 && y >= (int)gb.y0 && x < (int)(gb.x0 + gb.w) && y < (int)(gb.y0 + gb.h); }

struct Camera { V2 renderDim, jitter; V3 vbx, vby, vbz, origin; float tanHalfFOV, aspect; bool dof; float focusDepth, lensRadius; };
ZR_HD Camera CurrCamera(const zr_frame_constants& g)
{
    Camera c; c.renderDim = v2((float)g.render_width, (float)g.render_height); c.jitter = v2(g.curr_camera_jitter[0], g.curr_camera_jitter[1]);
    c.vbx = Row3(g.curr_view, 0); c.vby = Row3(g.curr_view, 1); c.vbz = Row3(g.curr_view, 2); c.origin = v3p(g.camera_pos);
    c.tanHalfFOV = g.tan_half_fov; c.aspect = g.aspect_ratio; c.dof = g.dof; c.focusDepth = g.focus_depth; c.lensRadius = g.lens_radius;
    return c;
}
ZR_HD Camera PrevCamera(const zr_frame_constants& g)
{
    Camera c = CurrCamera(g); c.jitter = v2(g.prev_camera_jitter[0], g.prev_camera_jitter[1]);
    c.vbx = Row3(g.prev_view, 0); c.vby = Row3(g.prev_view, 1); c.vbz = Row3(g.prev_view, 2);
    c.origin = v3(g.prev_view_inv[3], g.prev_view_inv[7], g.prev_view_inv[11]);
    return c;
}
// Math::WorldPosFromScreenSpace2, Math.hlsli:218-248
ZR_HD V3 WorldPosSS2(const Camera& c, float px, float py, float z_view, V2 lens, V3& origin)
{
    V2 uv = v2((px + 0.5f + c.jitter.x) / c.renderDim.x, (py + 0.5f + c.jitter.y) / c.renderDim.y);
    V2 ndc = NDCFromUV(uv);
    V3 dir_w;
    if (!c.dof)
    {
        V3 dv = v3(ndc.x * c.aspect * c.tanHalfFOV * z_view, ndc.y * c.tanHalfFOV * z_view, z_view);
        dir_w = mad(dv.x, c.vbx, mad(dv.y, c.vby, dv.z * c.vbz));
    }
    else
    {
        V3 dv = v3(ndc.x * c.aspect * c.tanHalfFOV, ndc.y * c.tanHalfFOV, 1);
        dv = c.focusDepth * dv - v3(lens.x, lens.y, 0);
        dir_w = normalize(mad(dv.x, c.vbx, mad(dv.y, c.vby, dv.z * c.vbz)));
        dir_w = dir_w * z_view;
        origin = origin + mad(lens.x, c.vbx, lens.y * c.vby);
    }
    return origin + dir_w;
}

struct PixelSurface { V3 pos, normal; float eta_next; Surface surface; GFlags flags; float roughness, z; V2 lens; V3 origin; };   // lens / origin: what RayDifferentials::Init needs

// coatPixel: the reference reads the coat plane at DTid instead of the shifted pixel in two passes
// (ReSTIR_PT_Reconnect_CtT.hlsl:80, _CtS.hlsl:99); restated as is.
ZR_HD PixelSurface LoadPixelSurfaceEx(const GBuf& gb, const Camera& cam, uint32_t x, uint32_t y, uint32_t frameForLens, size_t coatPixel, bool useTrDepth)
{
    PixelSurface ps;
    const size_t px = Pix(gb, x, y);
    const uint16_t mrp = gb.mr[px];
    ps.flags = DecodeFlags(mrp); ps.roughness = RoughnessOf(mrp); ps.z = gb.depth[px];
    if (gb.plain) { ps.flags.metallic = false; ps.flags.transmissive = false; ps.flags.trDepthGt0 = false; ps.flags.subsurface = false; ps.flags.coated = false; }      // what the planes of a plain scene hold (zr_dev_bsdf.h InitSurface)
    V2 lens = v2(0, 0);
    if (cam.dof)
    {
        uint32_t hx = x, hy = y, hz = x; zr_pcg3d(&hx, &hy, &hz);
        Rng r = Rng::Init(hz, hy, frameForLens);
        lens = UniformSampleDiskConcentric(r.Uniform2D());
        lens = lens * cam.lensRadius;
    }
    V3 origin = cam.origin;
    ps.pos = WorldPosSS2(cam, (float)x, (float)y, ps.z, lens, origin);
    ps.normal = DecodeOct32u(gb.normal[px]);
    const uint32_t bc = gb.baseColor[px];
    const V3 baseColor = UnpackRGB8(bc);
    const float subsurface = ps.flags.subsurface ? zr_div255((float)(bc >> 24)) : 0.0f;
    ps.eta_next = kDefaultEtaMat;
    if (ps.flags.transmissive) ps.eta_next = DecodeIOR(zr_div255((float)gb.ior[px]));
    float coat_weight = 0, coat_roughness = 0, coat_ior = kDefaultEtaCoat; V3 coat_color = v3(0.0f);
    if (ps.flags.coated)
    {
        const uint16_t* p = &gb.coat[4 * coatPixel];     // GBuffer::UnpackCoat, GBuffers.hlsli:107-121
        coat_weight = zr_div255((float)((p[1] >> 8) & 0xff));
        coat_roughness = zr_div255((float)(p[2] & 0xff));
        uint32_t c = (uint32_t)p[0] | (((uint32_t)p[1] & 0xff) << 16);
        coat_color = UnpackRGB8(c);
        coat_ior = DecodeIOR(zr_div255((float)(p[2] >> 8)));
    }
    const V3 wo = normalize(origin - ps.pos);
    ps.surface = InitSurface(ps.normal, wo, ps.flags.metallic, ps.roughness, baseColor, kEtaAir, ps.eta_next, ps.flags.transmissive,
        (useTrDepth && ps.flags.trDepthGt0) ? 1.0f : 0.0f, subsurface, coat_weight, coat_color, coat_roughness, coat_ior, gb.plain != 0);
    ps.lens = lens; ps.origin = origin;
    return ps;
}
ZR_HD PixelSurface LoadPixelSurface(const GBuf& gb, const Camera& cam, uint32_t x, uint32_t y, uint32_t frameForLens, size_t coatPixel)
{ return LoadPixelSurfaceEx(gb, cam, x, y, frameForLens, coatPixel, true); }

// ---- K11: one lane of PathTrace (ReSTIR_PT_PathTrace.hlsl:194-358), cut at the Russian-roulette point so the 64
// lanes of a wave step in lockstep around the WaveActiveMax
struct PTLane
{
    bool active, atRR, valid;
    uint32_t x, y;
    V3 pos, normal; Surface surface; BsdfSample bs;
    Rng rngReplay, rngThread, rngGroup;
    Reconnection rc; Reservoir r; V3 li, throughput, throughput_k; int bounce; PrevHit prevHit; float eta_curr, eta_next;
    bool inMedium; HitEm nextHit; uint32_t seed_replay, sampleSetIdx; int maxNumBounces;
    HitInfo hit; V3 tr; float prevPdf; uint32_t prevLobe; int pathVertex;
    RayDiffs rd; V3 dpdx, dpdy;      // textured scenes only
};

ZR_HD RayDiffs InitRD(const Camera& c, int x, int y, V2 lens, V3 origin)
{ return RayDiffs::Init(x, y, c.renderDim, c.tanHalfFOV, c.aspect, c.jitter, c.vbx, c.vby, c.vbz, c.dof, c.focusDepth, lens, origin); }
ZR_HD TriDiffs LoadTriDiffs(const GBuf& gb, size_t px) { return UnpackTriDiffs(&gb.triA[4 * px], &gb.triB[2 * px]); }
// the three calls every shift / path prologue makes at the primary hit (e.g. ReSTIR_PT_PathTrace.hlsl:380-392)
ZR_HD RayDiffs PrimaryRayDiffs(const Camera& c, int x, int y, const PixelSurface& ps, const TriDiffs& td, V3 wi)
{
    RayDiffs rd = InitRD(c, x, y, ps.lens, ps.origin);
    V3 dpdx, dpdy;
    rd.dpdx_dpdy(ps.pos, ps.normal, dpdx, dpdy);
    rd.ComputeUVDifferentials(dpdx, dpdy, td.dpdu, td.dpdv);
    rd.UpdateRays(ps.pos, ps.normal, wi, ps.surface.wo, td.dndu, td.dndv, dpdx, dpdy, dot(wi, ps.normal) < 0, ps.surface.eta);
    return rd;
}

struct RptParams
{
    uint32_t maxNonTrBounces, maxGlossyTrBounces, russianRoulette, numSampleSets, accumulate, boiling, M_max_temporal, M_max_spatial;
    float alpha_min;
    uint32_t doTemporal, doSpatial, writeReservoirs;
    uint32_t emissive;      // NEE_EMISSIVE: the scene has emissive triangles (else sun + sky)
    uint32_t textured;      // the scene has a texture heap: carry ray differentials, sample base-colour / metallic-roughness maps
    uint32_t sortTemporal, sortSpatial;      // CB_IND_FLAGS::SORT_TEMPORAL / SORT_SPATIAL: K12 thread maps are built and consumed
    uint32_t temporalMap;                    // which map schedules the fused CtT + TtC kernel: 0 none, 1 CtN, 2 NtC
};

ZR_HD Globals PtGlobals(const SceneView& sc, const zr_frame_constants& g, const RptParams& prm, const TravStack& stack, uint32_t* cnt, int maxNumBounces, uint32_t sampleSetIdx)
{
    Globals gl; gl.sc = &sc; gl.frame = &g; gl.emissive = prm.emissive != 0; gl.numEmissives = g.num_emissive_triangles; gl.maxNumBounces = maxNumBounces; gl.alpha_min = prm.alpha_min; gl.stack = stack; gl.cnt = cnt;
    gl.presampled = prm.numSampleSets != 0; gl.sampleSetIdx = sampleSetIdx;
    return gl;
}
// main() prologue + RIS_InitialCandidates up to the first FindClosest (ReSTIR_PT_PathTrace.hlsl:360-530, 194-236); q0 = that query
// (emissive variant only: the sun + sky variant traces at the top of PtPhaseA)
ZR_HD void PtInitLane_Pre(const SceneView& sc, const zr_frame_constants& g, const GBuf& gb, const RptParams& prm, bool owned, uint32_t x, uint32_t y,
    float* finalRGBA, uint32_t* cnt, PTLane& P, TraceReq& q0)
{
    q0 = NoTraceReq();
    P.active = false; P.atRR = false; P.valid = false; P.x = x; P.y = y;
    if (!owned) return;
    const size_t px = Pix(gb, x, y);
    GFlags flags = DecodeFlags(gb.mr[px]);
    if (flags.invalid || flags.emissive)
    {
        if (!prm.accumulate) { float* o = finalRGBA + 4 * px; o[0] = 0; o[1] = 0; o[2] = 0; }
        return;
    }
    P.valid = true;
    const Camera cam = CurrCamera(g);
    PixelSurface ps = LoadPixelSurface(gb, cam, x, y, g.frame_num, px);
    PrepareWo(sc.rho, ps.surface, kPrepK11);
    P.maxNumBounces = ps.surface.specTr ? (int)prm.maxGlossyTrBounces : (int)prm.maxNonTrBounces;
    { uint32_t a = x / 16, b = y / 8, c = g.frame_num, d = 1; zr_pcg4d(&a, &b, &c, &d); P.rngGroup = Rng::Seed(a); }
    uint32_t sx = x, sy = y, sz = g.frame_num; zr_pcg3d(&sx, &sy, &sz);
    P.rngReplay = Rng::Seed(sx); P.rngThread = Rng::Seed(sy); P.seed_replay = sx;
    P.r = InitReservoir(); P.li = v3(0.0f);
    BsdfSample bs;
    { ZR_PROF_SCOPE(ZRP_BSDF); bs = SampleBSDF(sc.rho, ps.normal, ps.surface, P.rngReplay); }
    if (dot(bs.bsdfOverPdf, bs.bsdfOverPdf) == 0) return;
    if (prm.textured) P.rd = PrimaryRayDiffs(cam, (int)x, (int)y, ps, LoadTriDiffs(gb, px), bs.wi);
    P.sampleSetIdx = prm.emissive ? P.rngGroup.UniformUintBounded_Faster(prm.numSampleSets) : 0u;      // ReSTIR_PT_PathTrace.hlsl:406-408
    P.rc = InitReconnection();
    P.bounce = 0; P.throughput = bs.bsdfOverPdf;
    P.prevHit.alpha_lobe = LobeAlpha(ps.surface, bs.lobe); P.prevHit.lobe = bs.lobe; P.prevHit.wi = bs.wi; P.prevHit.pdf = bs.pdf;
    P.eta_curr = dot(ps.normal, bs.wi) < 0 ? ps.eta_next : kEtaAir;
    P.throughput_k = v3(1.0f);
    P.inMedium = P.eta_curr != kEtaAir;
    P.pos = ps.pos; P.normal = ps.normal; P.surface = ps.surface; P.bs = bs; P.eta_next = ps.eta_next;
    if (prm.emissive) q0 = ClosestEmReq(cnt, ps.pos, ps.normal, bs.wi, ps.surface.Transmissive());
    P.active = true;
}
ZR_HD void PtInitLane_Post(const SceneView& sc, const RptParams& prm, PTLane& P, const TraceReq& q0, const RawHit& h0)
{ if (P.active && prm.emissive) P.nextHit = ClosestEmResult(sc, q0, h0); }
ZR_HD void PtInitLane(const SceneView& sc, const zr_frame_constants& g, const GBuf& gb, const RptParams& prm, bool owned, uint32_t x, uint32_t y,
    float* finalRGBA, TravStack stack, uint32_t* cnt, PTLane& P)
{
    TraceReq q0;
    PtInitLane_Pre(sc, g, gb, prm, owned, x, y, finalRGBA, cnt, P, q0);
    PtInitLane_Post(sc, prm, P, q0, TraceInline<false, false>(sc, q0, stack));
}

// PtPhaseA of the emissive variant, cut at its two BVH queries (M.q1 after _Pre, M.q2 after _Mid)
ZR_HD void PtPhaseA_Pre(const SceneView& sc, const zr_frame_constants& g, const RptParams& prm, uint32_t* cnt, PTLane& P, PtMid& M)
{
    P.atRR = false;
    M.q1 = NoTraceReq(); M.q2 = NoTraceReq(); M.doNee = false;
    if (!P.active) return;
    TravStack noStack; noStack.lds = nullptr; noStack.stride = 0; noStack.mem = nullptr;
    const Globals gl = PtGlobals(sc, g, prm, noStack, cnt, P.maxNumBounces, P.sampleSetIdx);
    P.pathVertex = P.bounce + 2;
    V3 newPos;
    {
    ZR_PROF_SCOPE(ZRP_MATERIAL);
    // the BSDF ray of the previous vertex's NEE (ReSTIR_PT_PathTrace.hlsl:235-239)
    if (!P.nextHit.hit) { P.active = false; return; }
    P.hit.t = P.nextHit.t;
    if (prm.textured) FillHit<true>(sc, P.nextHit.mesh, P.nextHit.prim, P.nextHit.bu, P.nextHit.bv, true, P.hit, true);
    else FillHit<false>(sc, P.nextHit.mesh, P.nextHit.prim, P.nextHit.bu, P.nextHit.bv, true, P.hit, true);
    newPos = mad(P.hit.t, P.bs.wi, P.pos);
    float eta_mat;
    V4 uvGrads = v4(0, 0, 0, 0);
    if (prm.textured)
    {
        P.rd.dpdx_dpdy(newPos, P.hit.normal, P.dpdx, P.dpdy);
        P.rd.ComputeUVDifferentials(P.dpdx, P.dpdy, P.hit.dpdu, P.hit.dpdv);
        uvGrads = P.rd.uv_grads;
    }
    if (!GetMaterialData(sc, -P.bs.wi, P.eta_curr, P.hit, P.surface, eta_mat, uvGrads, prm.textured)) { P.active = false; return; }
    PrepareWo(sc.rho, P.surface, kPrepK11);      // six evaluations per bounce share the surface's wo-only terms (zr_dev_bsdf.h)
    P.eta_next = eta_mat;
    }
    P.pos = newPos;
    P.normal = P.hit.normal;
    P.prevPdf = P.bs.pdf; P.prevLobe = P.bs.lobe;
    P.tr = v3(1.0f);
    if (P.inMedium && (P.surface.trDepth > 0))
    {
        V3 ext = -vlog(P.surface.base) / P.surface.trDepth;
        P.tr = vexp(-P.hit.t * ext);
        P.throughput = P.throughput * P.tr;
    }
    EstimateDirectEm_Pre(gl, P.pathVertex, P.pos, P.hit, P.surface, P.rngReplay, M);
}
// `live`: the lane went through _Pre without leaving the path (P.active may only be cleared by _Post's bookkeeping)
ZR_HD void PtPhaseA_Mid(const SceneView& sc, const zr_frame_constants& g, const RptParams& prm, uint32_t* cnt, PTLane& P, PtMid& M, const RawHit& h1)
{
    if (!P.active) return;
    TravStack noStack; noStack.lds = nullptr; noStack.stride = 0; noStack.mem = nullptr;
    const Globals gl = PtGlobals(sc, g, prm, noStack, cnt, P.maxNumBounces, P.sampleSetIdx);
    ZR_PROF_SCOPE(ZRP_MISC4);
    EstimateDirectEm_Mid(gl, P.pathVertex, P.pos, P.hit, P.surface, P.prevHit, P.throughput, P.throughput_k, P.li, P.nextHit, P.rc, P.r, P.rngThread, M, h1);
}
ZR_HD void PtPhaseA_Post(const SceneView& sc, const zr_frame_constants& g, const RptParams& prm, uint32_t* cnt, PTLane& P, PtMid& M, const RawHit& h2)
{
    if (!P.active) return;
    TravStack noStack; noStack.lds = nullptr; noStack.stride = 0; noStack.mem = nullptr;
    const Globals gl = PtGlobals(sc, g, prm, noStack, cnt, P.maxNumBounces, P.sampleSetIdx);
    { ZR_PROF_SCOPE(ZRP_MISC3);
    EstimateDirectEm_Post(gl, P.pathVertex, P.pos, P.hit, P.surface, P.prevHit, P.throughput, P.throughput_k, P.li, P.bs, P.rc, P.r, P.rngThread, M, h2); }
    if (P.bounce >= (P.maxNumBounces - 1)) { P.active = false; return; }
    if (P.rc.IsCase2() || P.rc.IsCase3()) P.rc.Clear();
    P.bounce++;
    P.atRR = prm.russianRoulette && (P.bounce >= 3);
}

ZR_HD void PtPhaseA(const SceneView& sc, const zr_frame_constants& g, const RptParams& prm, TravStack stack, uint32_t* cnt, PTLane& P)
{
    if (prm.emissive)
    {   // the cut form with its queries traced in place: one statement sequence for the megakernel, the host executor and the cooperative kernel
        PtMid M;
        PtPhaseA_Pre(sc, g, prm, cnt, P, M);
        const RawHit h1 = TraceInline<false, false>(sc, M.q1, stack);
        PtPhaseA_Mid(sc, g, prm, cnt, P, M, h1);
        const RawHit h2 = TraceInline<true, true>(sc, M.q2, stack);
        PtPhaseA_Post(sc, g, prm, cnt, P, M, h2);
        return;
    }
    P.atRR = false;
    if (!P.active) return;
    Globals gl = PtGlobals(sc, g, prm, stack, cnt, P.maxNumBounces, P.sampleSetIdx);
    P.pathVertex = P.bounce + 2;
    V3 newPos;
    {
    ZR_PROF_SCOPE(ZRP_MATERIAL);
    if (!FindClosestID(gl, true, P.pos, P.normal, P.bs.wi, P.surface.Transmissive(), P.hit, prm.textured)) { P.active = false; return; }   // Hit::FindClosest<true, true>
    newPos = mad(P.hit.t, P.bs.wi, P.pos);
    float eta_mat;
    V4 uvGrads = v4(0, 0, 0, 0);
    if (prm.textured)
    {
        P.rd.dpdx_dpdy(newPos, P.hit.normal, P.dpdx, P.dpdy);
        P.rd.ComputeUVDifferentials(P.dpdx, P.dpdy, P.hit.dpdu, P.hit.dpdv);
        uvGrads = P.rd.uv_grads;
    }
    if (!GetMaterialData(sc, -P.bs.wi, P.eta_curr, P.hit, P.surface, eta_mat, uvGrads, prm.textured)) { P.active = false; return; }
    PrepareWo(sc.rho, P.surface, kPrepK11);      // six evaluations per bounce share the surface's wo-only terms (zr_dev_bsdf.h)
    P.eta_next = eta_mat;
    }
    P.pos = newPos;
    P.normal = P.hit.normal;
    P.prevPdf = P.bs.pdf; P.prevLobe = P.bs.lobe;
    P.tr = v3(1.0f);
    if (P.inMedium && (P.surface.trDepth > 0))
    {
        V3 ext = -vlog(P.surface.base) / P.surface.trDepth;
        P.tr = vexp(-P.hit.t * ext);
        P.throughput = P.throughput * P.tr;
    }
    EstimateDirectAndUpdateRC(gl, P.pathVertex, P.pos, P.hit, P.surface, P.prevHit, P.throughput, P.throughput_k, P.li, P.bs, P.nextHit, P.rc, P.r,
        P.rngThread, P.rngReplay);
    if (P.bounce >= (P.maxNumBounces - 1)) { P.active = false; return; }
    if (P.rc.IsCase2() || P.rc.IsCase3()) P.rc.Clear();
    P.bounce++;
    P.atRR = prm.russianRoulette && (P.bounce >= 3);
}

// fused forms for the inline megakernel (see NEE_Bsdf_Fused)
// main() prologue + RIS_InitialCandidates up to the first FindClosest (ReSTIR_PT_PathTrace.hlsl:360-530, 194-236)
ZR_HD void PtInitLane_Fused(const SceneView& sc, const zr_frame_constants& g, const GBuf& gb, const RptParams& prm, bool owned, uint32_t x, uint32_t y,
    float* finalRGBA, TravStack stack, uint32_t* cnt, PTLane& P)
{
    P.active = false; P.atRR = false; P.valid = false; P.x = x; P.y = y;
    if (!owned) return;
    const size_t px = Pix(gb, x, y);
    GFlags flags = DecodeFlags(gb.mr[px]);
    if (flags.invalid || flags.emissive)
    {
        if (!prm.accumulate) { float* o = finalRGBA + 4 * px; o[0] = 0; o[1] = 0; o[2] = 0; }
        return;
    }
    P.valid = true;
    const Camera cam = CurrCamera(g);
    PixelSurface ps = LoadPixelSurface(gb, cam, x, y, g.frame_num, px);
    PrepareWo(sc.rho, ps.surface, kPrepK11);
    P.maxNumBounces = ps.surface.specTr ? (int)prm.maxGlossyTrBounces : (int)prm.maxNonTrBounces;
    { uint32_t a = x / 16, b = y / 8, c = g.frame_num, d = 1; zr_pcg4d(&a, &b, &c, &d); P.rngGroup = Rng::Seed(a); }
    uint32_t sx = x, sy = y, sz = g.frame_num; zr_pcg3d(&sx, &sy, &sz);
    P.rngReplay = Rng::Seed(sx); P.rngThread = Rng::Seed(sy); P.seed_replay = sx;
    P.r = InitReservoir(); P.li = v3(0.0f);
    BsdfSample bs;
    { ZR_PROF_SCOPE(ZRP_BSDF); bs = SampleBSDF(sc.rho, ps.normal, ps.surface, P.rngReplay); }
    if (dot(bs.bsdfOverPdf, bs.bsdfOverPdf) == 0) return;
    if (prm.textured) P.rd = PrimaryRayDiffs(cam, (int)x, (int)y, ps, LoadTriDiffs(gb, px), bs.wi);
    P.sampleSetIdx = prm.emissive ? P.rngGroup.UniformUintBounded_Faster(prm.numSampleSets) : 0u;      // ReSTIR_PT_PathTrace.hlsl:406-408
    P.rc = InitReconnection();
    P.bounce = 0; P.throughput = bs.bsdfOverPdf;
    P.prevHit.alpha_lobe = LobeAlpha(ps.surface, bs.lobe); P.prevHit.lobe = bs.lobe; P.prevHit.wi = bs.wi; P.prevHit.pdf = bs.pdf;
    P.eta_curr = dot(ps.normal, bs.wi) < 0 ? ps.eta_next : kEtaAir;
    P.throughput_k = v3(1.0f);
    P.inMedium = P.eta_curr != kEtaAir;
    P.pos = ps.pos; P.normal = ps.normal; P.surface = ps.surface; P.bs = bs; P.eta_next = ps.eta_next;
    Globals gl; gl.sc = &sc; gl.frame = &g; gl.emissive = prm.emissive != 0; gl.numEmissives = g.num_emissive_triangles; gl.maxNumBounces = P.maxNumBounces; gl.alpha_min = prm.alpha_min; gl.stack = stack; gl.cnt = cnt;
    gl.presampled = prm.numSampleSets != 0; gl.sampleSetIdx = P.sampleSetIdx;
    if (prm.emissive) P.nextHit = FindClosestEm_Fused(gl, ps.pos, ps.normal, bs.wi, ps.surface.Transmissive());
    P.active = true;
}

ZR_HD void PtPhaseA_Fused(const SceneView& sc, const zr_frame_constants& g, const RptParams& prm, TravStack stack, uint32_t* cnt, PTLane& P)
{
    P.atRR = false;
    if (!P.active) return;
    Globals gl; gl.sc = &sc; gl.frame = &g; gl.emissive = prm.emissive != 0; gl.numEmissives = g.num_emissive_triangles; gl.maxNumBounces = P.maxNumBounces; gl.alpha_min = prm.alpha_min; gl.stack = stack; gl.cnt = cnt;
    gl.presampled = prm.numSampleSets != 0; gl.sampleSetIdx = P.sampleSetIdx;
    P.pathVertex = P.bounce + 2;
    V3 newPos;
    {
    ZR_PROF_SCOPE(ZRP_MATERIAL);
    if (prm.emissive)
    {
        // the BSDF ray of the previous vertex's NEE (ReSTIR_PT_PathTrace.hlsl:235-239)
        if (!P.nextHit.hit) { P.active = false; return; }
        P.hit.t = P.nextHit.t;
        if (prm.textured) FillHit<true>(sc, P.nextHit.mesh, P.nextHit.prim, P.nextHit.bu, P.nextHit.bv, true, P.hit, true);
        else FillHit<false>(sc, P.nextHit.mesh, P.nextHit.prim, P.nextHit.bu, P.nextHit.bv, true, P.hit, true);
    }
    else if (!FindClosestID(gl, true, P.pos, P.normal, P.bs.wi, P.surface.Transmissive(), P.hit, prm.textured)) { P.active = false; return; }   // Hit::FindClosest<true, true>
    newPos = mad(P.hit.t, P.bs.wi, P.pos);
    float eta_mat;
    V4 uvGrads = v4(0, 0, 0, 0);
    if (prm.textured)
    {
        P.rd.dpdx_dpdy(newPos, P.hit.normal, P.dpdx, P.dpdy);
        P.rd.ComputeUVDifferentials(P.dpdx, P.dpdy, P.hit.dpdu, P.hit.dpdv);
        uvGrads = P.rd.uv_grads;
    }
    if (!GetMaterialData(sc, -P.bs.wi, P.eta_curr, P.hit, P.surface, eta_mat, uvGrads, prm.textured)) { P.active = false; return; }
    PrepareWo(sc.rho, P.surface, kPrepK11);      // six evaluations per bounce share the surface's wo-only terms (zr_dev_bsdf.h)
    P.eta_next = eta_mat;
    }
    P.pos = newPos;
    P.normal = P.hit.normal;
    P.prevPdf = P.bs.pdf; P.prevLobe = P.bs.lobe;
    P.tr = v3(1.0f);
    if (P.inMedium && (P.surface.trDepth > 0))
    {
        V3 ext = -vlog(P.surface.base) / P.surface.trDepth;
        P.tr = vexp(-P.hit.t * ext);
        P.throughput = P.throughput * P.tr;
    }
    EstimateDirectAndUpdateRC_Fused(gl, P.pathVertex, P.pos, P.hit, P.surface, P.prevHit, P.throughput, P.throughput_k, P.li, P.bs, P.nextHit, P.rc, P.r,
        P.rngThread, P.rngReplay);
    if (P.bounce >= (P.maxNumBounces - 1)) { P.active = false; return; }
    if (P.rc.IsCase2() || P.rc.IsCase3()) P.rc.Clear();
    P.bounce++;
    P.atRR = prm.russianRoulette && (P.bounce >= 3);
}


// bit pattern a lane contributes to the wave max (luminance of a non-negative throughput; NaN / negative -> 0)
ZR_HD uint32_t PtRRKey(const PTLane& P)
{
    if (!(P.active && P.atRR)) return 0;
    float lum = Luminance(P.throughput);
    return (zr_isnan(lum) || lum < 0) ? 0u : zr_asuint(lum);
}

ZR_HD void PtPhaseB(const SceneView& sc, const RptParams& prm, PTLane& P, uint32_t waveMaxBits)
{
    if (!P.active) return;
    const float waveThroughput = zr_asfloat(waveMaxBits);
    if (P.atRR && waveThroughput < 1)
    {
        float p_terminate = zr_max(0.05f, 1 - waveThroughput);
        if (P.rngGroup.Uniform() < p_terminate) { P.active = false; return; }
        P.throughput = P.throughput / (1 - p_terminate);
        P.throughput_k = P.throughput_k / (((int)P.rc.k <= P.bounce) ? (1 - p_terminate) : 1.0f);
    }
    if (!prm.emissive)      // ReSTIR_PT_PathTrace.hlsl:310-316
    {
        P.bs = InitBsdfSample();
        if (P.bounce < P.maxNumBounces) P.bs = SampleBSDF(sc.rho, P.normal, P.surface, P.rngReplay);
    }
    if (dot(P.bs.bsdfOverPdf, P.bs.bsdfOverPdf) == 0) { P.active = false; return; }
    const float alpha_lobe = LobeAlpha(P.surface, P.bs.lobe);
    if (P.rc.Empty() && CanReconnect(P.prevHit.alpha_lobe, alpha_lobe, P.prevHit.lobe, P.bs.lobe, prm.alpha_min))
    {
        P.rc.SetCase1(P.pathVertex, P.pos, P.hit.t, P.hit.normal, P.hit.ID, P.hit.meshIdx, -P.surface.wo, P.prevLobe, P.prevPdf, P.bs.wi, P.bs.lobe, P.bs.pdf);
        P.throughput_k = v3(1.0f);
    }
    if ((int)P.rc.k <= P.bounce) P.throughput_k = P.throughput_k * (P.bs.bsdfOverPdf * P.tr);
    bool transmitted = dot(P.normal, P.bs.wi) < 0;
    P.throughput = P.throughput * P.bs.bsdfOverPdf;
    P.eta_curr = transmitted ? (P.eta_curr == kEtaAir ? P.eta_next : kEtaAir) : P.eta_curr;
    P.inMedium = P.eta_curr != kEtaAir;
    P.prevHit.alpha_lobe = alpha_lobe; P.prevHit.lobe = P.bs.lobe; P.prevHit.wi = P.bs.wi; P.prevHit.pdf = P.bs.pdf;
    if (prm.textured) P.rd.UpdateRays(P.pos, P.normal, P.bs.wi, P.surface.wo, P.hit.dndu, P.hit.dndv, P.dpdx, P.dpdy, transmitted, P.surface.eta);
}

// ---- the path state that lives across a bounce boundary (after PtPhaseB, before the next PtPhaseA) of the untextured kernels, as 32-bit
// words: what K11 with per-bounce path compaction (zr_kernels.h: k_rpt_pt_first / k_rpt_pt_next) moves through its SoA planes.  Everything
// else in PTLane is either recomputed by PtPhaseA before it is read (surface, hit, tr, prevPdf / prevLobe, pathVertex, eta_next, atRR) or dead.
// One function enumerates the fields for both directions: V::f / u / i move a float / uint32_t / int, V::Store says which way.
static constexpr uint32_t kPtCarryWords = 78;
template<class V> ZR_HD void PtCarryRc(V& v, Reconnection& rc)
{
    v.f(rc.x_k.x); v.f(rc.x_k.y); v.f(rc.x_k.z); v.u(rc.ID); v.u(rc.meshIdx); v.f(rc.partialJacobian); v.f(rc.w.x); v.f(rc.w.y); v.f(rc.w.z);
    v.f(rc.lightPdf); v.u(rc.seed_replay); v.u(rc.seed_nee); v.f(rc.dwdA); v.f(rc.L.x); v.f(rc.L.y); v.f(rc.L.z);
    uint32_t m = 0;
    if (V::Store) m = (rc.k & 0xffu) | ((rc.lobe_k_min_1 & 0xfu) << 8) | ((rc.lobe_k & 0xfu) << 12) | ((rc.lt_k & 0xfu) << 16) | ((rc.lt_k_plus_1 & 0xfu) << 20) | ((rc.x_k_in_motion ? 1u : 0u) << 24);
    v.u(m);
    if (!V::Store) { rc.k = m & 0xffu; rc.lobe_k_min_1 = (m >> 8) & 0xfu; rc.lobe_k = (m >> 12) & 0xfu; rc.lt_k = (m >> 16) & 0xfu; rc.lt_k_plus_1 = (m >> 20) & 0xfu; rc.x_k_in_motion = ((m >> 24) & 1u) != 0; }
}
template<class V> ZR_HD void PtCarry(V& v, PTLane& P)
{
    uint32_t pix = 0, flags = 0;
    if (V::Store)
    {
        pix = P.x | (P.y << 16);
        flags = (P.inMedium ? 1u : 0u) | (P.nextHit.hit ? 2u : 0u) | (P.surface.Transmissive() ? 4u : 0u) | ((uint32_t)P.bounce << 8) | ((uint32_t)P.maxNumBounces << 16);
    }
    v.u(pix); v.u(flags);
    if (!V::Store)
    {
        P.x = pix & 0xffffu; P.y = pix >> 16; P.valid = true; P.active = true; P.atRR = false;
        P.inMedium = (flags & 1u) != 0; P.nextHit.hit = (flags & 2u) != 0; P.bounce = (int)((flags >> 8) & 0xffu); P.maxNumBounces = (int)(flags >> 16);
        // (the sun + sky variant traces its continuation ray at the top of PtPhaseA and only asks the old surface whether it transmits)
        P.surface.specTr = (flags & 4u) != 0; P.surface.subsurface = 0.0f;
    }
    v.f(P.pos.x); v.f(P.pos.y); v.f(P.pos.z); v.f(P.normal.x); v.f(P.normal.y); v.f(P.normal.z);
    v.f(P.bs.wi.x); v.f(P.bs.wi.y); v.f(P.bs.wi.z); v.f(P.bs.pdf); v.u(P.bs.lobe);
    v.u(P.rngReplay.s); v.u(P.rngThread.s); v.u(P.rngGroup.s);
    PtCarryRc(v, P.rc);
    v.f(P.r.w_sum); v.f(P.r.target.x); v.f(P.r.target.y); v.f(P.r.target.z); v.u(P.r.M);
    PtCarryRc(v, P.r.rc);
    v.f(P.li.x); v.f(P.li.y); v.f(P.li.z); v.f(P.throughput.x); v.f(P.throughput.y); v.f(P.throughput.z);
    v.f(P.throughput_k.x); v.f(P.throughput_k.y); v.f(P.throughput_k.z);
    v.f(P.prevHit.alpha_lobe); v.f(P.prevHit.wi.x); v.f(P.prevHit.wi.y); v.f(P.prevHit.wi.z); v.f(P.prevHit.pdf); v.u(P.prevHit.lobe);
    v.f(P.eta_curr);
    v.f(P.nextHit.t); v.u(P.nextHit.mesh); v.u(P.nextHit.prim); v.f(P.nextHit.bu); v.f(P.nextHit.bv);
    v.u(P.seed_replay); v.u(P.sampleSetIdx);
}
struct PtCarryStore
{
    static constexpr bool Store = true;
    uint32_t* p; size_t stride; uint32_t n = 0;
    ZR_HDM void f(float& x) { *p = zr_asuint(x); p += stride; n++; }
    ZR_HDM void u(uint32_t& x) { *p = x; p += stride; n++; }
};
struct PtCarryLoad
{
    static constexpr bool Store = false;
    const uint32_t* p; size_t stride; uint32_t n = 0;
    ZR_HDM void f(float& x) { x = zr_asfloat(*p); p += stride; n++; }
    ZR_HDM void u(uint32_t& x) { x = *p; p += stride; n++; }
};

struct RptTex   // per-pass auxiliary planes
{
    F4* target;          // RGBA32F (xyz)
    uint8_t* neighbor;   // RG8_UINT
};

// main() epilogue (ReSTIR_PT_PathTrace.hlsl:526-559)
ZR_HD void PtFinishLane(const GBuf& gb, const RptParams& prm, const ResPlanes& out, const RptTex& tex, float* finalRGBA, PTLane& P)
{
    if (!P.valid) return;
    const size_t px = Pix(gb, P.x, P.y);
    Reservoir& r = P.r;
    if (r.park.p && r.parked) RcParkLoad(r.park, r.rc);      // K11 with the selected reconnection parked in LDS (RcPark)
    r.rc.seed_replay = P.seed_replay;
    float targetLum = Luminance(r.target);
    r.W = targetLum > 0 ? zr_max(r.w_sum / targetLum, 1.0f) : 0;
    if (prm.writeReservoirs) r.Write(out, px, 0, prm.emissive != 0);
    if (prm.doTemporal) tex.target[px] = f4(Sanitize3(r.target), 0.0f);
    else
    {
        V3 li = any_nan(P.li) ? v3(0.0f) : P.li;
        float* o = finalRGBA + 4 * px;
        if (prm.accumulate) { o[0] += li.x; o[1] += li.y; o[2] += li.z; }
        else { o[0] = li.x; o[1] = li.y; o[2] = li.z; }
    }
}

// Util.hlsli:141-159
ZR_HD void WriteOutputColor(const zr_frame_constants& g, float* finalRGBA, size_t px, V3 li)
{
    li = any_nan(li) ? v3(0.0f) : li;
    float* o = finalRGBA + 4 * px;
    if (g.accumulate && g.camera_static && g.num_frames_camera_static > 1) { o[0] += li.x; o[1] += li.y; o[2] += li.z; }
    else { o[0] = li.x; o[1] = li.y; o[2] = li.z; }
}

// ---- r-buffers (Shift.hlsli:191-358)
struct RBuf
{
    uint16_t* A;   // RGBA16F (throughput, max uv grad)
    U4* B; U4* C;  // RGBA32_UINT
    uint16_t* D;   // R16_UINT
    uint32_t plain = 0;      // as GBuf::plain, for the surfaces rebuilt from the r-buffer
};
struct OffsetCtx { V3 throughput, pos, normal; Surface surface; float eta_curr, eta_next; Rng rngReplay; RayDiffs rd; };   // rd: textured scenes only
ZR_HD OffsetCtx InitOffsetCtx()
{
    OffsetCtx c; c.throughput = v3(0.0f); c.pos = v3(0.0f); c.normal = v3(0.0f);
    c.surface = InitSurface(v3(0, 0, 1), v3(0, 0, 1), false, 0, v3(0.0f), kEtaAir, kDefaultEtaMat, false, 0, 0, 0, v3(0.0f), 0, kDefaultEtaCoat);
    c.eta_curr = kEtaAir; c.eta_next = kDefaultEtaMat; c.rngReplay.s = 0;
    c.rd = RayDiffs::Zero();
    return c;
}
ZR_HD OffsetCtx LoadOffsetCtx(const RBuf& rb, size_t i, bool isCase3 = false, bool tex = false)
{
    OffsetCtx ctx = InitOffsetCtx();
    if (tex && !isCase3) { const float inAw = zr_f16_to_f32(rb.A[4 * i + 3]); ctx.rd.uv_grads = v4(inAw, inAw, inAw, inAw); }
    ctx.throughput = v3(zr_f16_to_f32(rb.A[4 * i]), zr_f16_to_f32(rb.A[4 * i + 1]), zr_f16_to_f32(rb.A[4 * i + 2]));
    if (dot(ctx.throughput, ctx.throughput) == 0) return ctx;
    const U4 b = rb.B[i], c = rb.C[i];
    ctx.pos = v3(zr_asfloat(b.x), zr_asfloat(b.y), zr_asfloat(b.z));
    ctx.normal = DecodeOct32u(b.w);
    ctx.eta_curr = zr_fma(zr_div255((float)((c.z >> 8) & 0xff)), 1.5f, 1.0f);
    ctx.eta_next = zr_fma(zr_div255((float)((c.z >> 16) & 0xff)), 1.5f, 1.0f);
    V3 wo = DecodeOct32u(c.x);
    float roughness = zr_div255((float)(c.z & 0xff));
    V3 baseColor = UnpackRGB8(c.y & 0xffffff);
    uint32_t flags = c.y >> 24;
    if (rb.plain) flags = 0;
    bool metallic = flags & 0x1, specTr = (flags & 0x4) == 0x4;
    float trDepth = (flags & 0x8) == 0x8 ? 1.0f : 0.0f;
    bool coated = (flags & 0x10) == 0x10;
    float subsurface = zr_div255((float)((c.z >> 24) & 0xff));
    float eta_next = ctx.eta_curr == kEtaAir ? ctx.eta_next : kEtaAir;
    float coat_weight = 0, coat_roughness = 0, coat_ior = kDefaultEtaCoat; V3 coat_color = v3(0.0f);
    if (coated)
    {
        uint32_t c_w = c.w; uint32_t d_w = rb.D[i];
        coat_weight = zr_div255((float)((c_w >> 24) & 0xff));
        coat_color = UnpackRGB8(c_w & 0xffffff);
        coat_roughness = zr_div255((float)(d_w & 0xff));
        coat_ior = zr_fma(zr_div255((float)((d_w >> 8) & 0xff)), 1.5f, 1.0f);
    }
    ctx.surface = InitSurface(ctx.normal, wo, metallic, roughness, baseColor, ctx.eta_curr, eta_next, specTr, trDepth, zr_round_f16(subsurface),
        coat_weight, coat_color, coat_roughness, coat_ior, rb.plain != 0);
    return ctx;
}
ZR_HD void WriteOffsetCtx(const OffsetCtx& ctx, const RBuf& rb, size_t i, bool isCase3, bool tex = false)
{
    if (!isCase3)
    {
        // max uv gradient (untextured scenes: never read, written as 0)
        uint16_t gradMax = 0;
        if (tex)
        {
            const V4 uv = ctx.rd.uv_grads;
            const float ddx_uv = zr_sqrt(uv.x * uv.x + uv.y * uv.y);
            const float ddy_uv = zr_sqrt(uv.z * uv.z + uv.w * uv.w);
            gradMax = zr_f32_to_f16(zr_max(ddx_uv, ddy_uv));
        }
        rb.A[4 * i + 3] = gradMax;
    }
    rb.A[4 * i] = zr_f32_to_f16(ctx.throughput.x); rb.A[4 * i + 1] = zr_f32_to_f16(ctx.throughput.y); rb.A[4 * i + 2] = zr_f32_to_f16(ctx.throughput.z);
    if (dot(ctx.throughput, ctx.throughput) == 0) return;
    const Surface& s = ctx.surface;
    V2 e1 = EncodeUnitVector(ctx.normal), e2 = EncodeUnitVector(s.wo);
    uint32_t flags = (uint32_t)s.metallic | ((uint32_t)s.specTr << 2) | ((uint32_t)(s.trDepth > 0) << 3) | ((uint32_t)s.Coated() << 4);
    uint32_t roughness = FloatToUNorm8(!s.GlossSpecular() ? zr_sqrt(s.alpha) : 0);
    uint32_t ec = FloatToUNorm8((ctx.eta_curr - 1.0f) / 1.5f), en = FloatToUNorm8((ctx.eta_next - 1.0f) / 1.5f);
    uint32_t ss = FloatToUNorm8(s.subsurface);
    U4 b, c;
    b.x = zr_asuint(ctx.pos.x); b.y = zr_asuint(ctx.pos.y); b.z = zr_asuint(ctx.pos.z); b.w = FloatToUNorm16(e1.x) | (FloatToUNorm16(e1.y) << 16);
    c.x = FloatToUNorm16(e2.x) | (FloatToUNorm16(e2.y) << 16);
    c.y = Float3ToRGB8(s.base) | (flags << 24);
    c.z = roughness | (ec << 8) | (en << 16) | (ss << 24);
    c.w = rb.C[i].w;
    if (s.Coated())
    {
        uint32_t cw = FloatToUNorm8(s.coat_weight), cc = Float3ToRGB8(s.coat_color);
        uint32_t cr = FloatToUNorm8(!s.CoatSpecular() ? zr_sqrt(s.coat_alpha) : 0);
        float coat_eta = s.coat_eta >= 1.0f ? s.coat_eta : 1.0f / s.coat_eta;
        uint32_t ce = FloatToUNorm8((coat_eta - 1.0f) / 1.5f);
        c.w = cc | (cw << 24);
        rb.D[i] = (uint16_t)(cr | (ce << 8));
    }
    rb.B[i] = b; rb.C[i] = c;
}

// Shift.hlsli:377-474
ZR_HD_FLAT void Replay(const Globals& g, bool currFrame, int numBounces, BsdfSample bs, OffsetCtx& ctx)
{
    const SceneView& sc = *g.sc;
    ctx.throughput = bs.bsdfOverPdf;
    int bounce = 0;
    ctx.eta_curr = dot(ctx.normal, bs.wi) < 0 ? ctx.eta_next : kEtaAir;
    bool inMedium = ctx.eta_curr != kEtaAir;
    float alpha_prev = LobeAlpha(ctx.surface, bs.lobe);
    uint32_t lobe_prev = bs.lobe;
    while (true)
    {
        HitInfo hit;
        // Hit::FindClosest<false, ...>: no ID
        {
            F4 ro, rd;
            if (!MakeClosestRay(ctx.pos, ctx.normal, bs.wi, ctx.surface.Transmissive(), false, &ro, &rd)) { ctx.throughput = v3(0.0f); return; }
            g.cnt[0]++;
            RawHit h = Traverse<false>(sc, xyz(ro), xyz(rd), ro.w, rd.w, ZR_SUBGROUP_ALL, g.stack);
            if (h.tri == kInvalidTri) { ctx.throughput = v3(0.0f); return; }
            const TriMeta tm = sc.triMeta[h.tri];
            hit.t = h.t;
            if (g.textured) FillHit<true>(sc, tm.mesh, tm.prim, h.u, h.v, false, hit, currFrame);
            else FillHit<false>(sc, tm.mesh, tm.prim, h.u, h.v, false, hit, currFrame);
        }
        float eta_mat;
        V3 dpdx = v3(0.0f), dpdy = v3(0.0f);
        if (g.textured)
        {
            const V3 newPos = mad(hit.t, bs.wi, ctx.pos);
            ctx.rd.dpdx_dpdy(newPos, hit.normal, dpdx, dpdy);
            ctx.rd.ComputeUVDifferentials(dpdx, dpdy, hit.dpdu, hit.dpdv);
        }
        if (!GetMaterialData(sc, -bs.wi, ctx.eta_curr, hit, ctx.surface, eta_mat, ctx.rd.uv_grads, g.textured)) { ctx.throughput = v3(0.0f); return; }
        ctx.eta_next = eta_mat;
        ctx.pos = mad(hit.t, bs.wi, ctx.pos);
        ctx.normal = hit.normal;
        bounce++;
        if (inMedium && (ctx.surface.trDepth > 0))
        {
            V3 ext = -vlog(ctx.surface.base) / ctx.surface.trDepth;
            ctx.throughput = ctx.throughput * vexp(-hit.t * ext);
        }
        if (bounce >= numBounces) break;
        if (PrepShiftGroups(sc.plain)) PrepareWo(sc.rho, ctx.surface, PrepShiftGroups(sc.plain));
        bs = SampleBSDF(sc.rho, ctx.normal, ctx.surface, ctx.rngReplay);
        if (dot(bs.bsdfOverPdf, bs.bsdfOverPdf) == 0) { ctx.throughput = v3(0.0f); return; }
        const float alpha_lobe = LobeAlpha(ctx.surface, bs.lobe);
        if (CanReconnect(alpha_prev, alpha_lobe, lobe_prev, bs.lobe, g.alpha_min)) { ctx.throughput = v3(0.0f); return; }
        const bool transmitted = dot(ctx.normal, bs.wi) < 0;
        ctx.eta_curr = transmitted ? (ctx.eta_curr == kEtaAir ? ctx.eta_next : kEtaAir) : ctx.eta_curr;
        ctx.throughput = ctx.throughput * bs.bsdfOverPdf;
        inMedium = ctx.eta_curr != kEtaAir;
        alpha_prev = alpha_lobe; lobe_prev = bs.lobe;
        if (g.textured) ctx.rd.UpdateRays(ctx.pos, ctx.normal, bs.wi, ctx.surface.wo, hit.dndu, hit.dndv, dpdx, dpdy, transmitted, ctx.surface.eta);
    }
}

// Shift.hlsli:818-859
// What RayDifferentials::Init and the triangle differentials of a shift's primary hit are built from (textured scenes): the
// pixel the offset path starts at, its camera, lens sample / ray origin and the G-buffer holding its triangle differentials.
struct PrimaryDiffs { Camera cam; int x, y; V2 lens; V3 origin; const GBuf* gb; size_t px; };
ZR_HD PrimaryDiffs MakePrimaryDiffs(const Camera& cam, int x, int y, const PixelSurface& ps, const GBuf& gb, size_t px)
{ PrimaryDiffs d; d.cam = cam; d.x = x; d.y = y; d.lens = ps.lens; d.origin = ps.origin; d.gb = &gb; d.px = px; return d; }

// preComputed: Replay_CtT evaluates dpdx_dpdy / ComputeUVDifferentials once more before the call (ReSTIR_PT_Replay.hlsl:114-117)
ZR_HD_FLAT OffsetCtx Replay_kGt2(const Globals& g, bool currFrame, V3 pos, V3 normal, float ior, const Surface& surface, const Reconnection& rc,
    const PrimaryDiffs& pd, bool preComputed = false)
{
    OffsetCtx ctx = InitOffsetCtx();
    ctx.pos = pos; ctx.normal = normal; ctx.surface = surface; ctx.rngReplay = Rng::Seed(rc.seed_replay);
    ctx.eta_curr = kEtaAir; ctx.eta_next = ior; ctx.throughput = v3(1.0f);
    const int numBounces = (int)rc.k - 2;
    if (PrepShiftGroups(g.sc->plain)) PrepareWo(g.sc->rho, ctx.surface, PrepShiftGroups(g.sc->plain));
    BsdfSample bs = SampleBSDF(g.sc->rho, ctx.normal, ctx.surface, ctx.rngReplay);
    if (dot(bs.bsdfOverPdf, bs.bsdfOverPdf) == 0) { ctx.throughput = v3(0.0f); return ctx; }
    if (g.textured)
    {
        const TriDiffs td = LoadTriDiffs(*pd.gb, pd.px);
        ctx.rd = InitRD(pd.cam, pd.x, pd.y, pd.lens, pd.origin);
        V3 dpdx, dpdy;
        if (preComputed) { ctx.rd.dpdx_dpdy(pos, normal, dpdx, dpdy); ctx.rd.ComputeUVDifferentials(dpdx, dpdy, td.dpdu, td.dpdv); }
        ctx.rd.dpdx_dpdy(ctx.pos, ctx.normal, dpdx, dpdy);
        ctx.rd.ComputeUVDifferentials(dpdx, dpdy, td.dpdu, td.dpdv);
        ctx.rd.UpdateRays(ctx.pos, ctx.normal, bs.wi, ctx.surface.wo, td.dndu, td.dndv, dpdx, dpdy, dot(bs.wi, ctx.normal) < 0, ctx.surface.eta);
    }
    Replay(g, currFrame, numBounces, bs, ctx);
    return ctx;
}

// Shift.hlsli:476-546
ZR_HD_FLAT float StepPath(const Globals& g, bool currFrame, OffsetCtx& ctx, const Reconnection& rc)
{
    const SceneView& sc = *g.sc;
    if (!IsLobeValid(ctx.surface, rc.lobe_k_min_1)) return 0;
    float alpha_k_min_1 = LobeAlpha(ctx.surface, rc.lobe_k_min_1);
    if (!CanReconnect(alpha_k_min_1, 1, rc.lobe_k_min_1, rc.lobe_k, g.alpha_min)) return 0;
    V3 w_k_min_1 = normalize(rc.x_k - ctx.pos);
    SamplerEval e;
    {
    ZR_PROF_SCOPE(ZRP_BSDF);      // (-DZR_PROF builds: the evaluation at y_{k-1})
    if (PrepShiftGroups(sc.plain)) PrepareWo(sc.rho, ctx.surface, PrepShiftGroups(sc.plain));      // y_{k-1}: the lobe candidates of EvalBSDFSampler share its wo-only terms
    e = EvalBSDFSampler(sc.rho, ctx.normal, ctx.surface, w_k_min_1, rc.lobe_k_min_1, ctx.rngReplay);
    }
    if (dot(e.bsdfOverPdf, e.bsdfOverPdf) == 0) return 0;
    HitInfo hit;
    if (!FindClosestID(g, currFrame, ctx.pos, ctx.normal, w_k_min_1, ctx.surface.Transmissive(), hit)) return 0;
    if (hit.ID != rc.ID) return 0;
    const V3 y_k = mad(hit.t, w_k_min_1, ctx.pos);
    const bool transmitted = dot(ctx.normal, w_k_min_1) < 0;
    ctx.eta_curr = transmitted ? (ctx.eta_curr == kEtaAir ? ctx.eta_next : kEtaAir) : ctx.eta_curr;
    const bool inMedium = ctx.eta_curr != kEtaAir;
    float eta_mat;
    if (g.textured) { hit.dndu = v3(0.0f); hit.dndv = v3(0.0f); }     // (not fetched here; GetMaterialData may flip them)
    // RtRayQuery::IsotropicSampler with g_samLinearWrap (Shift.hlsli:519-521)
    { ZR_PROF_SCOPE(ZRP_MATERIAL);
    if (!GetMaterialData(sc, -w_k_min_1, ctx.eta_curr, hit, ctx.surface, eta_mat, ctx.rd.uv_grads, g.textured, true)) return 0; }
    if (PrepShiftGroups(sc.plain)) PrepareWo(sc.rho, ctx.surface, PrepShiftGroups(sc.plain));      // y_k: evaluated two to four times by the case-1 / case-2 branches of Shift2
    ctx.eta_next = eta_mat;
    if (inMedium && (ctx.surface.trDepth > 0))
    {
        V3 ext = -vlog(ctx.surface.base) / ctx.surface.trDepth;
        ctx.throughput = ctx.throughput * vexp(-hit.t * ext);
    }
    float pj = e.pdf;
    pj *= zr_abs(dot(-w_k_min_1, hit.normal));
    pj /= (hit.t * hit.t);
    ctx.pos = y_k; ctx.normal = hit.normal; ctx.throughput = ctx.throughput * e.bsdfOverPdf;
    return pj;
}

// RPT_Util::EstimateDirect_y_k_min_1, Shift.hlsli:548-660 (SKY_SAMPLING_PREFER_PERFORMANCE == 1): the sun / sky RIS of NEE_NonEmissive
// re-run at the offset path's y_{k-1} with the base path's pick (lt, wd, lobe) forced; returns ld, writes its pdf
ZR_HD V3 EstimateDirect_y_k_min_1(const Globals& g, OffsetCtx ctx, uint32_t lt, V3 wd, uint32_t lobe, Rng rngNEE, float& pdfOut)
{
    const SceneView& sc = *g.sc; const zr_frame_constants& fr = *g.frame; const RhoView& rho = sc.rho;
    rngNEE.Uniform2D();
    const V2 u_d = rngNEE.Uniform2D();
    const V2 u_c = rngNEE.Uniform2D();
    const V2 u_g = rngNEE.Uniform2D();
    const float u_wrs_b0 = rngNEE.Uniform();
    const float u_wrs_b1 = rngNEE.Uniform();
    SkyIncidentRadiance leFunc; leFunc.lut = sc.sky;
    const bool specular = IsSpecular(ctx.surface);
    V3 target_z = v3(0.0f);
    float w_sum;
    {
        const V3 wi_sun = -v3p(fr.sun_dir);
        float pdf_b = 0, pdf_e = 0;
        const bool visible = (wi_sun.y > 0) && ((dot(wi_sun, ctx.normal) > 0) || ctx.surface.Transmissive());
        if (visible)
        {
            ctx.surface.SetWi(wi_sun, ctx.normal);
            target_z = Le_Sun(ctx.pos, fr) * Unified(rho, ctx.surface).f;
            const float ndotWi = dot(wi_sun, ctx.normal);
            pdf_b = (ndotWi < 0) && ctx.surface.ThinWalled() ? 0 : BSDFSamplerPdf_NoDiffuse(rho, ctx.normal, ctx.surface, wi_sun, leFunc);
            pdf_e = (!specular ? 1.0f : 0.0f) * zr_abs(ndotWi) * ZR_ONE_OVER_PI;
            pdf_e *= ctx.surface.ThinWalled() ? 0.5f : (ndotWi > 0 ? 1.0f : 0.0f);
        }
        w_sum = BalanceHeuristic3(1, pdf_e, pdf_b, Luminance(target_z));
    }
    if (!specular)
    {
        const bool isZ_e = lt == LT_SKY && lobe == LOBE_ALL;
        float pdf_unused;
        V3 wi_e = isZ_e ? wd : SampleDiffuse(ctx.normal, u_d, &pdf_unused);
        float pdf_e = zr_saturate(dot(ctx.normal, wi_e)) * ZR_ONE_OVER_PI;
        if (ctx.surface.ThinWalled()) { wi_e = u_wrs_b1 > 0.5f ? -wi_e : wi_e; pdf_e *= 0.5f; }
        ctx.surface.SetWi(wi_e, ctx.normal);
        const V3 target = leFunc(wi_e) * Unified(rho, ctx.surface).f;
        target_z = isZ_e ? target : target_z;
        const float pdf_b = !ctx.surface.reflection && ctx.surface.ThinWalled() ? 0 : BSDFSamplerPdf_NoDiffuse(rho, ctx.normal, ctx.surface, wi_e, leFunc);
        const float denom = pdf_e + pdf_b;
        w_sum += denom == 0 ? 0.0f : Luminance(target) / denom;
    }
    const bool isZ_b = lt == LT_SKY && lobe != LOBE_ALL;
    if (isZ_b)
    {
        const SamplerEval e = EvalBSDFSampler_NoDiffuse(rho, ctx.normal, ctx.surface, wd, lobe, leFunc);
        target_z = e.f;
        const float ndotwi = dot(wd, ctx.normal);
        float pdf_e = (!specular ? 1.0f : 0.0f) * zr_abs(ndotwi) * ZR_ONE_OVER_PI;
        pdf_e *= ctx.surface.ThinWalled() ? 0.5f : (ndotwi > 0 ? 1.0f : 0.0f);
        const float denom = e.pdf + pdf_e;
        w_sum += denom == 0 ? 0.0f : Luminance(e.f) / denom;
    }
    else
    {
        const BsdfSample bs = SampleBSDF_NoDiffuse(rho, ctx.normal, ctx.surface, u_c, u_g, u_wrs_b0, u_wrs_b1, leFunc);
        const float ndotwi = dot(bs.wi, ctx.normal);
        float pdf_e = (!specular ? 1.0f : 0.0f) * zr_abs(ndotwi) * ZR_ONE_OVER_PI;
        pdf_e *= ctx.surface.ThinWalled() ? 0.5f : (ndotwi > 0 ? 1.0f : 0.0f);
        const float denom = bs.pdf + pdf_e;
        w_sum += denom == 0 ? 0.0f : Luminance(bs.f) / denom;
    }
    const float targetLum = Luminance(target_z);
    pdfOut = w_sum > 0 ? targetLum / w_sum : 0;
    return targetLum > 0 ? target_z * w_sum / targetLum : v3(0.0f);
}

struct OffsetPath { V3 target; float partialJacobian; bool surfKMin1Transmissive; };

// Shift2<Emissive = true>, Shift.hlsli:662-816
ZR_HD_FLAT OffsetPath Shift2(const Globals& g, bool currFrame, size_t DTidIdx, V3 pos, V3 normal, float ior, const Surface& surface,
    const Reconnection& rc, const RBuf& rbuffer, const PrimaryDiffs& pd)
{
    OffsetCtx ctx = InitOffsetCtx();
    ctx.pos = pos; ctx.normal = normal; ctx.surface = surface; ctx.rngReplay = Rng::Seed(rc.seed_replay);
    ctx.eta_curr = kEtaAir; ctx.eta_next = ior; ctx.throughput = v3(1.0f);
    OffsetPath ret; ret.target = v3(0.0f); ret.partialJacobian = 0; ret.surfKMin1Transmissive = false;
    const int numBounces = (int)rc.k - 2;
    if (numBounces == 0)
    {
        if (g.textured)
        {
            // Shift.hlsli:688-700: the uv gradients of the primary hit feed the isotropic LOD of the reconnection vertex
            const TriDiffs td = LoadTriDiffs(*pd.gb, pd.px);
            ctx.rd = InitRD(pd.cam, pd.x, pd.y, pd.lens, pd.origin);
            V3 dpdx, dpdy;
            ctx.rd.dpdx_dpdy(ctx.pos, ctx.normal, dpdx, dpdy);
            ctx.rd.ComputeUVDifferentials(dpdx, dpdy, td.dpdu, td.dpdv);
            const V3 wi = normalize(rc.x_k - ctx.pos);
            ctx.rd.UpdateRays(ctx.pos, ctx.normal, wi, ctx.surface.wo, td.dndu, td.dndv, dpdx, dpdy, dot(wi, ctx.normal) < 0, ior);
        }
    }
    else
    {
        ctx = LoadOffsetCtx(rbuffer, DTidIdx, rc.IsCase3(), g.textured);
        if (dot(ctx.throughput, ctx.throughput) == 0) return ret;
        // Load() leaves rngReplay at state 0; the reference then advances that state (Shift.hlsli:707-713) -- restated as is
        for (int b = 0; b < numBounces; b++) for (int k = 0; k < 9; k++) ctx.rngReplay.Uniform();
    }
    ret.surfKMin1Transmissive = ctx.surface.specTr;
    if (!rc.IsCase3())
    {
        ret.partialJacobian = StepPath(g, currFrame, ctx, rc);
        if (ret.partialJacobian == 0) return ret;
        if (rc.IsCase1())
        {
            ZR_PROF_SCOPE(ZRP_NEE);      // (-DZR_PROF builds: ZRP_NEE = the evaluation at y_k, all three cases)
            SamplerEval e = EvalBSDFSampler(g.sc->rho, ctx.normal, ctx.surface, rc.w, rc.lobe_k, ctx.rngReplay);
            ctx.throughput = ctx.throughput * e.bsdfOverPdf;
            ret.target = ctx.throughput * rc.L;
            ret.partialJacobian *= e.pdf;
            return ret;
        }
    }
    else
    {
        if (!IsLobeValid(ctx.surface, rc.lobe_k_min_1)) return ret;
        if (LobeAlpha(ctx.surface, rc.lobe_k_min_1) < g.alpha_min) return ret;
        if (PrepShiftGroups(g.sc->plain)) PrepareWo(g.sc->rho, ctx.surface, PrepShiftGroups(g.sc->plain));
    }
    Rng rngNEE = Rng::Seed(rc.seed_nee);
    if (!g.emissive)      // Shift2<Emissive = false>, Shift.hlsli:788-813
    {
        const uint32_t lt = rc.IsCase2() ? rc.lt_k_plus_1 : rc.lt_k;
        // (both fields read, then selected: a load through a selected ADDRESS keeps a whole reservoir in scratch memory -- SROA gives up on the object)
        const uint32_t lobe_a = rc.lobe_k, lobe_b = rc.lobe_k_min_1;
        const uint32_t lobe = rc.IsCase2() ? lobe_a : lobe_b;
        float pdf;
        const V3 target = EstimateDirect_y_k_min_1(g, ctx, lt, rc.w, lobe, rngNEE, pdf);
        ret.target = ctx.throughput * target;
        if (rc.IsCase2()) ret.partialJacobian *= pdf;
        else
        {
            ret.partialJacobian = pdf;
            if (dot(ret.target, ret.target) > 0)
            {
                const V3 wi = rc.lt_k == LT_SUN ? -v3p(g.frame->sun_dir) : rc.w;
                ret.target = ret.target * (VisibilityRay(g, ctx.pos, wi, ctx.normal, ctx.surface.Transmissive()) ? 1.0f : 0.0f);
            }
        }
        return ret;
    }
    ZR_PROF_SCOPE(ZRP_NEE);
    if (rc.IsCase2())
    {
        Direct ls = EvalDirect_Case2(g, ctx.normal, ctx.surface, rc.w, rc.L, rc.dwdA, rc.lightPdf, rc.lobe_k, ctx.rngReplay, rngNEE);
        ret.target = ctx.throughput * ls.ld;
        ret.partialJacobian *= ls.pdf_solidAngle;
    }
    else
    {
        V3 wi = rc.x_k - ctx.pos;
        float t = length(wi);
        wi = wi / t;
        bool twoSided = rc.lightPdf > 0;
        // the reference passes ctx.pos for both `pos` and `normal` (Shift.hlsli:780-782); restated as is
        Direct ls = EvalDirect_Case3(g, ctx.pos, ctx.pos, ctx.surface, wi, t, rc.L, rc.w, zr_abs(rc.lightPdf), rc.ID, twoSided, rc.lobe_k_min_1,
            ctx.rngReplay, rngNEE);
        ret.target = ctx.throughput * ls.ld;
        ret.partialJacobian = ls.pdf_solidAngle;
    }
    return ret;
}

// ---- everything one frame of the temporal / spatial passes needs
struct RptFrame
{
    SceneView sc; GBuf gb, gbPrev; ResPlanes cur, prev;    // cur = this frame's reservoirs, prev = the other set
    SceneView scPrev;                                      // the scene as it was last frame (== sc while nothing moves)
    RBuf rbCtN, rbNtC; RptTex tex; float* finalRGBA; const uint16_t* sampleSet; RptParams prm;
    uint16_t* mapCtN; uint16_t* mapNtC;                    // K12 thread maps (R16_UINT, plane layout)
    // pixels this device is responsible for (global coordinates); the planes also hold an apron of neighbouring tiles'
    // pixels: G-buffer rendered locally, reservoirs received through the halo exchange
    uint32_t ox0, oy0, ow, oh;
    ZR_HDM bool Owns(uint32_t x, uint32_t y) const { return x >= ox0 && y >= oy0 && x < ox0 + ow && y < oy0 + oh; }
    // GPU time per 32 x 32-pixel cell of the planes (cell (0, 0) at the plane origin gb.x0, gb.y0): wave lifetimes of K11 / K14 / K16, one
    // atomic per wave: what the cost-balanced tile split of the multi-GPU path is computed from (tiling.balanced_layout); null = off
    // costMode 1 (ZR_COST_MAP_RAYS): the cells accumulate the BVH queries (closest + any-hit) issued for their pixels by EVERY kernel of the pass
    // instead -- the per-window ray counts the at-size parity tests compare with the host executor's
    uint32_t* costMap; uint32_t costW; uint32_t costMode;
    // diagnostic (ZR_K11=trip, zr_kernels.h: k_rpt_pathtrace_trip): {alive lanes, lane slots, words per path} of the waves that pass a bounce
    // boundary, and the planes the carried state is sent through there
    uint32_t* trip; unsigned long long* tripStats; size_t tripStride;
    // K11 with per-bounce path compaction: path state planes [word][slot] written by one bounce's kernel and read by the next (ping-pong),
    // carryCount[b] = paths alive after bounce b (their slots are 0 .. count - 1), carryCap = slots per plane
    uint32_t* carryOut; const uint32_t* carryIn; uint32_t* carryCount; size_t carryCap; uint32_t carryBounce;
};

// the PLAIN permutation of a kernel writes its template constant into every structure a surface is rebuilt from (scene, G-buffers, r-buffers): the
// material class then is a compile-time fact in InitSurface, LoadPixelSurfaceEx and LoadOffsetCtx, and the metal / transmission / thin-wall / coat code folds away
ZR_HD void SetMaterialClass(RptFrame& F, bool plain)
{ const uint32_t p = plain ? 1u : 0u; F.sc.plain = p; F.scPrev.plain = p; F.gb.plain = p; F.gbPrev.plain = p; F.rbCtN.plain = p; F.rbNtC.plain = p; }
ZR_HD Globals MakeGlobals(const RptFrame& F, const zr_frame_constants& g, bool transmissive, TravStack stack, uint32_t* cnt)
{
    Globals gl; gl.textured = F.prm.textured != 0; gl.sc = &F.sc; gl.scPrev = &F.scPrev; gl.frame = &g; gl.emissive = F.prm.emissive != 0; gl.numEmissives = g.num_emissive_triangles; gl.alpha_min = F.prm.alpha_min; gl.stack = stack; gl.cnt = cnt; gl.presampled = false; gl.sampleSetIdx = 0;
    gl.maxNumBounces = transmissive ? (int)F.prm.maxGlossyTrBounces : (int)F.prm.maxNonTrBounces;
    return gl;
}

struct TemporalPixel { bool ok; int px, py; PixelSurface prev; };
ZR_HD TemporalPixel FindTemporal(const RptFrame& F, const zr_frame_constants& g, uint32_t x, uint32_t y, const PixelSurface& cur, float planeTh,
    bool coatAtDTid)
{
    TemporalPixel t; t.ok = false; t.px = 0; t.py = 0;
    const V2 renderDim = v2((float)g.render_width, (float)g.render_height);
    const size_t px = Pix(F.gb, x, y);
    const V2 motionVec = DecodeMotion(F.gb.motion[px]);
    const V2 currUV = v2(((float)x + 0.5f) / renderDim.x, ((float)y + 0.5f) / renderDim.y);
    const V2 prevUV = currUV - motionVec;
    int ppx = (int)(prevUV.x * renderDim.x), ppy = (int)(prevUV.y * renderDim.y);
    if (prevUV.x < 0 || prevUV.y < 0 || prevUV.x > 1 || prevUV.y > 1) return t;
    if (ppx >= (int)g.render_width || ppy >= (int)g.render_height) return t;      // prevUV == 1: out of bounds, pinned to "no history"
    if (!InPlanes(F.gb, ppx, ppy)) return t;      // screen-tile split: reprojection beyond the apron = no history (SURVEY 8(e) caveat)
    const size_t pp = Pix(F.gb, (uint32_t)ppx, (uint32_t)ppy);
    if (F.gbPrev.depth[pp] == ZR_FLT_MAX) return t;
    const Camera pcam = PrevCamera(g);
    t.prev = LoadPixelSurface(F.gbPrev, pcam, (uint32_t)ppx, (uint32_t)ppy, g.frame_num - 1, coatAtDTid ? px : pp);
    float planeDist = zr_abs(dot(cur.normal, t.prev.pos - cur.pos));
    if (!(planeDist <= planeTh * cur.z)) return t;
    if (t.prev.flags.emissive || (zr_abs(t.prev.roughness - cur.roughness) > kMaxRoughDiffTemporal) || (t.prev.flags.transmissive != cur.flags.transmissive)) return t;
    t.ok = true; t.px = ppx; t.py = ppy;
    return t;
}

// K13 Replay_CtT / Replay_TtC (ReSTIR_PT_Replay.hlsl:289-534)
ZR_HD_FLAT void ReplayTemporalPixel(const RptFrame& F, const zr_frame_constants& g, int variant, uint32_t x, uint32_t y, TravStack stack, uint32_t* cnt)
{
    const size_t px = Pix(F.gb, x, y);
    GFlags flags = DecodeFlags(F.gb.mr[px]);
    if (flags.invalid || flags.emissive) return;
    const Camera cam = CurrCamera(g);
    PixelSurface ps = LoadPixelSurface(F.gb, cam, x, y, g.frame_num, px);
    TemporalPixel tp = FindTemporal(F, g, x, y, ps, 0.01f, false);
    if (!tp.ok) return;
    Globals gl = MakeGlobals(F, g, flags.transmissive, stack, cnt);
    const size_t pp = Pix(F.gb, (uint32_t)tp.px, (uint32_t)tp.py);
    if (variant == 0)
    {
        Reservoir r = Load_Metadata(F.cur, px);
        if (!r.rc.Empty() && (r.rc.k > 2))
        {
            r.Load_Reconnection(F.cur, px, F.prm.emissive != 0);
            Globals glP = gl; glP.sc = &F.scPrev;       // Replay_CtT binds the previous acceleration structure + mesh instances (IndirectLighting.cpp:465-471)
            OffsetCtx ctx = Replay_kGt2(glP, false, tp.prev.pos, tp.prev.normal, tp.prev.eta_next, tp.prev.surface, r.rc,
                MakePrimaryDiffs(PrevCamera(g), tp.px, tp.py, tp.prev, F.gbPrev, pp), true);
            WriteOffsetCtx(ctx, F.rbCtN, px, r.rc.IsCase3(), gl.textured);
        }
    }
    else
    {
        Reservoir r = Load_Metadata(F.prev, pp);
        if (!r.rc.Empty() && (r.rc.k > 2))
        {
            r.Load_Reconnection(F.prev, pp, F.prm.emissive != 0);
            OffsetCtx ctx = Replay_kGt2(gl, true, ps.pos, ps.normal, ps.eta_next, ps.surface, r.rc, MakePrimaryDiffs(cam, (int)x, (int)y, ps, F.gb, px));
            WriteOffsetCtx(ctx, F.rbNtC, px, r.rc.IsCase3(), gl.textured);
        }
    }
}

// x_k moved between the current and the previous frame's instance transform (CtT.hlsl:258-272, TtC.hlsl:309-329)
ZR_HD void MoveXk(const SceneView& sc, Reconnection& rc, bool currToPrev, bool setMotionFlag)
{
    const zr_mesh_instance& md = sc.instances[rc.meshIdx];
    V4 q_curr = normalize(DecodeNormalized4(md.rotation));
    V4 q_prev = normalize(DecodeNormalized4(md.prev_rotation));
    V3 s_curr = v3(zr_f16_to_f32(md.scale[0]), zr_f16_to_f32(md.scale[1]), zr_f16_to_f32(md.scale[2]));
    V3 s_prev = v3(zr_f16_to_f32(md.prev_scale[0]), zr_f16_to_f32(md.prev_scale[1]), zr_f16_to_f32(md.prev_scale[2]));
    V3 dT = v3(zr_f16_to_f32(md.d_translation[0]), zr_f16_to_f32(md.d_translation[1]), zr_f16_to_f32(md.d_translation[2]));
    V3 t_curr = v3p(md.translation), t_prev = t_curr - dT;
    if (currToPrev) rc.x_k = TransformTRS(InverseTransformTRS(rc.x_k, t_curr, q_curr, s_curr), t_prev, q_prev, s_prev);
    else rc.x_k = TransformTRS(InverseTransformTRS(rc.x_k, t_prev, q_prev, s_prev), t_curr, q_curr, s_curr);
    if (setMotionFlag)
    {
        V4 dRot = v4(q_prev.x - q_curr.x, q_prev.y - q_curr.y, q_prev.z - q_curr.z, q_prev.w - q_curr.w);
        V3 dScale = s_prev - s_curr;
        rc.x_k_in_motion = dot(dT, dT) > 0;
        rc.x_k_in_motion = rc.x_k_in_motion || dot(dRot, dRot) > 0;
        rc.x_k_in_motion = rc.x_k_in_motion || dot(dScale, dScale) > 0;
    }
}

// K14 Reconnect_CtT (ReSTIR_PT_Reconnect_CtT.hlsl:130-292) followed by Reconnect_TtC (ReSTIR_PT_Reconnect_TtC.hlsl:124-390)
// for one pixel.  The reference runs them as two dispatches; both only read/write this pixel's current reservoir (CtT
// writes w_sum, TtC reads it back) and read the previous frame's set, so running them back to back per pixel gives the
// same result and shares the G-buffer reconstruction and the temporal-pixel search.
ZR_HD void ReconnectTemporalPixel(const RptFrame& F, const zr_frame_constants& g, uint32_t x, uint32_t y, TravStack stack, uint32_t* cnt)
{
    const size_t px = Pix(F.gb, x, y);
    GFlags flags = DecodeFlags(F.gb.mr[px]);
    if (flags.invalid || flags.emissive) return;
    const bool doSpatial = F.prm.doSpatial;
    const Camera cam = CurrCamera(g);
    PixelSurface ps; TemporalPixel tp;
    if ((ZR_PIN_TEMPORAL & 1) && !F.sc.plain) ZR_KEEP_IN_MEMORY(ps);
    if ((ZR_PIN_TEMPORAL & 2) && !F.sc.plain) ZR_KEEP_IN_MEMORY(tp);
    {
    ZR_PROF_SCOPE(ZRP_MISC4);      // (-DZR_PROF builds: the two pixel surfaces)
    ps = LoadPixelSurface(F.gb, cam, x, y, g.frame_num, px);
    // CtT reads the previous G-buffer's coat plane at DTid (ReSTIR_PT_Reconnect_CtT.hlsl:80), TtC at prevPixel
    tp = FindTemporal(F, g, x, y, ps, kMaxPlaneDistReuse, true);
    }
    Globals gl = MakeGlobals(F, g, flags.transmissive, stack, cnt);
    const size_t pp = Pix(F.gb, (uint32_t)tp.px, (uint32_t)tp.py);

    // ---- current -> temporal: MIS weight of the current sample
    if (tp.ok)
    {
        Reservoir r_curr = Load_NonReconnection(F.cur, px);
        Reservoir r_prev = Load_Metadata(F.prev, pp);
        if (r_curr.w_sum != 0 && r_prev.M > 0 && !r_curr.rc.Empty())
        {
            r_curr.Load_Reconnection(F.cur, px, F.prm.emissive != 0);
            // the CtT pass binds the previous acceleration structure and mesh instances
            Globals glP = gl; glP.sc = &F.scPrev;
            if (r_curr.rc.IsCase1() || r_curr.rc.IsCase2()) MoveXk(F.scPrev, r_curr.rc, true, false);
            OffsetPath shift = Shift2(glP, false, px, tp.prev.pos, tp.prev.normal, tp.prev.eta_next, tp.prev.surface, r_curr.rc, F.rbCtN,
                MakePrimaryDiffs(PrevCamera(g), tp.px, tp.py, tp.prev, F.gbPrev, pp));
            float target_prev = Luminance(shift.target);
            if (target_prev > 0)
            {
                float targetLum_curr = r_curr.W > 0 ? r_curr.w_sum / r_curr.W : 0;
                float jacobian = r_curr.rc.partialJacobian > 0 ? shift.partialJacobian / r_curr.rc.partialJacobian : 0;
                float m_curr = targetLum_curr / (targetLum_curr + (float)r_prev.M * target_prev * jacobian);
                r_curr.w_sum *= m_curr;
                F.cur.B[2 * px] = r_curr.w_sum;
            }
        }
    }

    // ---- temporal -> current: resample
    Reservoir r_curr = Load_NonReconnection(F.cur, px);
    r_curr.target = xyz(F.tex.target[px]);
    if (!tp.ok)
    {
        if (!doSpatial) WriteOutputColor(g, F.finalRGBA, px, r_curr.target * r_curr.W);
        return;
    }
    Reservoir r_prev = Load_NonReconnection(F.prev, pp);
    const uint32_t M_max = F.prm.M_max_temporal;
    const uint32_t M_new = (r_curr.M + r_prev.M) & 0xffffu;
    if (r_prev.rc.Empty())
    {
        float targetLum = Luminance(r_curr.target);
        r_curr.W = targetLum > 0 ? r_curr.w_sum / targetLum : 0;
        r_curr.M = M_new;
        r_curr.WriteReservoirData2(F.cur, px, M_max);
        if (!doSpatial) WriteOutputColor(g, F.finalRGBA, px, r_curr.target * r_curr.W);
        return;
    }
    r_prev.Load_Reconnection(F.prev, pp, F.prm.emissive != 0);
    if (r_prev.rc.IsCase1() || r_prev.rc.IsCase2()) MoveXk(F.sc, r_prev.rc, false, true);
    OffsetPath shift = Shift2(gl, true, px, ps.pos, ps.normal, ps.eta_next, ps.surface, r_prev.rc, F.rbNtC, MakePrimaryDiffs(cam, (int)x, (int)y, ps, F.gb, px));
    float targetLum_curr = Luminance(shift.target);
    float jacobian = r_prev.rc.partialJacobian > 0 ? shift.partialJacobian / r_prev.rc.partialJacobian : 0;
    bool changed = false;
    if (targetLum_curr > 1e-6f && jacobian > 1e-5f)
    {
        Rng rng = Rng::Init(y, x, g.frame_num + 31);
        float targetLum_prev = r_prev.W > 0 ? r_prev.w_sum / r_prev.W : 0;
        float numerator = (float)r_prev.M * targetLum_prev;
        float denom = numerator / jacobian + targetLum_curr;
        float m_prev = denom > 0 ? numerator / denom : 0;
        float w_prev = m_prev * r_prev.W * targetLum_curr;
        if (r_curr.Update(w_prev, shift.target, r_prev.rc, rng)) { r_curr.rc.partialJacobian = shift.partialJacobian; changed = true; }
    }
    float targetLum = Luminance(r_curr.target);
    r_curr.W = targetLum > 0 ? r_curr.w_sum / targetLum : 0;
    r_curr.M = M_new;
    if (changed)
    {
        r_curr.Write(F.cur, px, M_max, F.prm.emissive != 0);
        if (doSpatial) F.tex.target[px] = f4(Sanitize3(r_curr.target), 0.0f);
    }
    else r_curr.WriteReservoirData(F.cur, px, M_max);
    if (!doSpatial) WriteOutputColor(g, F.finalRGBA, px, r_curr.target * r_curr.W);
}

// cheap predicates for the replay work lists (supersets of the pixels the replay passes act on; the passes re-check).  They return the replay
// class of the reservoir's stored k - 2: 0 = no replay (empty or k = 2), 1 / 2 / 3 = k == 3 / k == 4 / k >= 5 -- the buckets of K12, by which
// k_rpt_light orders the entries of a block so that the lanes of a replay wave walk paths of equal length
ZR_HD uint32_t ReplayClass(uint32_t a) { const uint32_t k = a & 0xfu; return (k == Reconnection::EMPTY || k == 0u) ? 0u : (k < 3u ? k : 3u); }
ZR_HD uint32_t NeedsReplayCtT(const RptFrame& F, uint32_t x, uint32_t y)
{
    const size_t px = Pix(F.gb, x, y);
    GFlags flags = DecodeFlags(F.gb.mr[px]);
    if (flags.invalid || flags.emissive) return 0u;
    return ReplayClass(F.cur.A[px]);
}
ZR_HD uint32_t NeedsReplayTtC(const RptFrame& F, const zr_frame_constants& g, uint32_t x, uint32_t y)
{
    const size_t px = Pix(F.gb, x, y);
    GFlags flags = DecodeFlags(F.gb.mr[px]);
    if (flags.invalid || flags.emissive) return false;
    const V2 renderDim = v2((float)g.render_width, (float)g.render_height);
    const V2 motionVec = DecodeMotion(F.gb.motion[px]);
    const V2 prevUV = v2(((float)x + 0.5f) / renderDim.x, ((float)y + 0.5f) / renderDim.y) - motionVec;
    if (prevUV.x < 0 || prevUV.y < 0 || prevUV.x > 1 || prevUV.y > 1) return false;
    const int ppx = (int)(prevUV.x * renderDim.x), ppy = (int)(prevUV.y * renderDim.y);
    if (ppx >= (int)g.render_width || ppy >= (int)g.render_height || !InPlanes(F.gb, ppx, ppy)) return false;
    return ReplayClass(F.prev.A[Pix(F.gb, (uint32_t)ppx, (uint32_t)ppy)]);
}

// Math::WorldPosFromScreenSpace, Math.hlsli:205-216
ZR_HD V3 WorldPosSS(float px, float py, V2 renderDim, float z_view, float tanHalfFOV, float aspect, const float* viewInv, V2 jitter)
{
    V2 uv = v2((px + 0.5f + jitter.x) / renderDim.x, (py + 0.5f + jitter.y) / renderDim.y);
    V2 ndc = NDCFromUV(uv);
    V3 d = v3(ndc.x * aspect * tanHalfFOV * z_view, ndc.y * tanHalfFOV * z_view, z_view);
    return v3(viewInv[0] * d.x + viewInv[1] * d.y + viewInv[2] * d.z + viewInv[3],
              viewInv[4] * d.x + viewInv[5] * d.y + viewInv[6] * d.z + viewInv[7],
              viewInv[8] * d.x + viewInv[9] * d.y + viewInv[10] * d.z + viewInv[11]);
}

// How K15 reads a candidate's G-buffer texels: straight from the planes ...
struct PlaneFetch
{
    const GBuf* gb;
    ZR_HDM void operator()(int sx, int sy, uint16_t& mr, float& depth, uint32_t& normal) const
    { const size_t sp = Pix(*gb, (uint32_t)sx, (uint32_t)sy); mr = gb->mr[sp]; depth = gb->depth[sp]; normal = gb->normal[sp]; }
};
// K15 SpatialSearch (ReSTIR_PT_SpatialSearch.hlsl:21-146); Fetch: see PlaneFetch (zr_kernels.h has the LDS-tile variant measured for DESIGN's N3 row)
template<typename Fetch>
ZR_HD void SpatialSearchPixelT(const RptFrame& F, const zr_frame_constants& g, uint32_t x, uint32_t y, const Fetch& fetch)
{
    const GBuf& gb = F.gb;
    const uint32_t W = g.render_width, H = g.render_height;
    const size_t px = Pix(gb, x, y);
    const uint16_t mrp = gb.mr[px];
    GFlags flags = DecodeFlags(mrp);
    if (flags.invalid || flags.emissive) return;
    const V2 renderDim = v2((float)g.render_width, (float)g.render_height);
    const V2 jitter = v2(g.curr_camera_jitter[0], g.curr_camera_jitter[1]);
    const float roughness = RoughnessOf(mrp);
    const float viewDepth = gb.depth[px];
    const V3 pos = WorldPosSS((float)x, (float)y, renderDim, viewDepth, g.tan_half_fov, g.aspect_ratio, g.curr_view_inv, jitter);
    const V3 normal = DecodeOct32u(gb.normal[px]);
    uint32_t sx = x, sy = y, sz = g.frame_num; zr_pcg3d(&sx, &sy, &sz);
    Rng rng = Rng::Init(sx, sy, g.frame_num);
    const float u0 = rng.Uniform();
    const uint32_t offset = rng.UniformUint();
    const float theta = u0 * ZR_TWO_PI;
    float sinTheta, cosTheta; zr_sincos(theta, &sinTheta, &cosTheta);
    int nx = 0xffff, ny = 0xffff;
    for (uint32_t i = 0; i < 3; i++)
    {
        const uint32_t si = (offset + i) & 511u;
        const float ux = zr_f16_to_f32(F.sampleSet[2 * si]), uy = zr_f16_to_f32(F.sampleSet[2 * si + 1]);
        float rx = ux * cosTheta + uy * -sinTheta, ry = ux * sinTheta + uy * cosTheta;
        rx = rx * (float)kSearchRadius; ry = ry * (float)kSearchRadius;
        const int sxp = (int)__builtin_rintf((float)x + rx), syp = (int)__builtin_rintf((float)y + ry);
        if (sxp < 0 || syp < 0 || sxp >= (int)W || syp >= (int)H) continue;
        if (sxp == (int)x && syp == (int)y) continue;
        if (!InPlanes(gb, sxp, syp)) continue;      // cannot happen with an apron >= the search radius
        uint16_t smr; float sdepth; uint32_t snormal;
        fetch(sxp, syp, smr, sdepth, snormal);
        GFlags sf = DecodeFlags(smr);
        if (sf.invalid || sf.emissive) continue;
        if (flags.metallic != sf.metallic) continue;
        if (flags.transmissive != sf.transmissive) continue;
        if (zr_abs(RoughnessOf(smr) - roughness) > kMaxRoughDiffSpatial) continue;
        const V3 samplePos = WorldPosSS((float)sxp, (float)syp, renderDim, sdepth, g.tan_half_fov, g.aspect_ratio, g.curr_view_inv, jitter);
        const V3 sampleNormal = DecodeOct32u(snormal);
        float planeDist = zr_abs(dot(normal, samplePos - pos));
        if (!(planeDist <= 0.01f * viewDepth)) continue;
        if (dot(sampleNormal, normal) < kMinNormalSimSpatial) continue;
        nx = sxp; ny = syp;
        break;
    }
    if (nx == 0xffff) { F.tex.neighbor[2 * px] = 255; F.tex.neighbor[2 * px + 1] = 255; }
    else { F.tex.neighbor[2 * px] = (uint8_t)(nx - (int)x + kNeighborOffset); F.tex.neighbor[2 * px + 1] = (uint8_t)(ny - (int)y + kNeighborOffset); }
}

ZR_HD void SpatialSearchPixel(const RptFrame& F, const zr_frame_constants& g, uint32_t x, uint32_t y)
{ PlaneFetch f; f.gb = &F.gb; SpatialSearchPixelT(F, g, x, y, f); }

ZR_HD bool NeighborOf(const RptFrame& F, uint32_t x, uint32_t y, int& sx, int& sy)
{
    const size_t px = Pix(F.gb, x, y);
    if (F.tex.neighbor[2 * px] == 255) return false;
    sx = (int)F.tex.neighbor[2 * px] - kNeighborOffset + (int)x; sy = (int)F.tex.neighbor[2 * px + 1] - kNeighborOffset + (int)y;
    return true;
}

// In the spatial passes F.cur = the temporal pass's output ("in"), F.prev = the set written for the next frame ("out")
// K13 Replay_CtS / Replay_StC
ZR_HD_FLAT void ReplaySpatialPixel(const RptFrame& F, const zr_frame_constants& g, int variant, uint32_t x, uint32_t y, TravStack stack, uint32_t* cnt)
{
    const size_t px = Pix(F.gb, x, y);
    GFlags flags = DecodeFlags(F.gb.mr[px]);
    if (flags.invalid || flags.emissive) return;
    Globals gl = MakeGlobals(F, g, flags.transmissive, stack, cnt);
    const Camera cam = CurrCamera(g);
    int sx, sy;
    if (variant == 0)
    {
        Reservoir r = Load_Metadata(F.cur, px);
        if (!r.rc.Empty() && (r.rc.k > 2))
        {
            r.Load_Reconnection(F.cur, px, F.prm.emissive != 0);
            if (!NeighborOf(F, x, y, sx, sy)) return;
            const size_t sp = Pix(F.gb, (uint32_t)sx, (uint32_t)sy);
            PixelSurface pn = LoadPixelSurface(F.gb, cam, (uint32_t)sx, (uint32_t)sy, g.frame_num, sp);
            OffsetCtx ctx = Replay_kGt2(gl, true, pn.pos, pn.normal, pn.eta_next, pn.surface, r.rc, MakePrimaryDiffs(cam, sx, sy, pn, F.gb, sp));
            WriteOffsetCtx(ctx, F.rbCtN, px, r.rc.IsCase3(), gl.textured);
        }
    }
    else
    {
        if (!NeighborOf(F, x, y, sx, sy)) return;
        const size_t sp = Pix(F.gb, (uint32_t)sx, (uint32_t)sy);
        Reservoir r = Load_Metadata(F.cur, sp);
        if (!r.rc.Empty() && (r.rc.k > 2))
        {
            r.Load_Reconnection(F.cur, sp, F.prm.emissive != 0);
            PixelSurface ps = LoadPixelSurface(F.gb, cam, x, y, g.frame_num, px);
            OffsetCtx ctx = Replay_kGt2(gl, true, ps.pos, ps.normal, ps.eta_next, ps.surface, r.rc, MakePrimaryDiffs(cam, (int)x, (int)y, ps, F.gb, px));
            WriteOffsetCtx(ctx, F.rbNtC, px, r.rc.IsCase3(), gl.textured);
        }
    }
}

ZR_HD uint32_t NeedsReplayCtS(const RptFrame& F, uint32_t x, uint32_t y)
{
    const size_t px = Pix(F.gb, x, y);
    GFlags flags = DecodeFlags(F.gb.mr[px]);
    if (flags.invalid || flags.emissive) return false;
    if (F.tex.neighbor[2 * px] == 255) return false;
    return ReplayClass(F.cur.A[px]);
}
ZR_HD uint32_t NeedsReplayStC(const RptFrame& F, uint32_t x, uint32_t y)
{
    const size_t px = Pix(F.gb, x, y);
    GFlags flags = DecodeFlags(F.gb.mr[px]);
    if (flags.invalid || flags.emissive) return false;
    int sx, sy;
    if (!NeighborOf(F, x, y, sx, sy)) return false;
    return ReplayClass(F.cur.A[Pix(F.gb, (uint32_t)sx, (uint32_t)sy)]);
}

// K16 Reconnect_CtS (ReSTIR_PT_Reconnect_CtS.hlsl:149-230).  Like CtT/TtC it only touches this pixel's entries (reads the
// "in" set, writes w_sum of the "out" set that StC reads back), so the StC kernel runs it per lane before its phase 1.
ZR_HD void ReconnectCtSPixel(const RptFrame& F, const zr_frame_constants& g, uint32_t x, uint32_t y, TravStack stack, uint32_t* cnt)
{
    const size_t px = Pix(F.gb, x, y);
    int sx, sy;
    if (!NeighborOf(F, x, y, sx, sy)) return;
    GFlags flags = DecodeFlags(F.gb.mr[px]);
    if (flags.invalid || flags.emissive) return;
    const size_t sp = Pix(F.gb, (uint32_t)sx, (uint32_t)sy);
    Reservoir r_curr = Load_NonReconnection(F.cur, px);
    Reservoir r_spatial = Load_Metadata(F.cur, sp);
    if ((r_curr.w_sum != 0) && !r_curr.rc.Empty())
    {
        r_curr.Load_Reconnection(F.cur, px, F.prm.emissive != 0);
        Globals gl = MakeGlobals(F, g, flags.transmissive, stack, cnt);
        const Camera cam = CurrCamera(g);
        PixelSurface pn = LoadPixelSurface(F.gb, cam, (uint32_t)sx, (uint32_t)sy, g.frame_num, px);
        OffsetPath shift = Shift2(gl, true, px, pn.pos, pn.normal, pn.eta_next, pn.surface, r_curr.rc, F.rbCtN, MakePrimaryDiffs(cam, sx, sy, pn, F.gb, sp));
        float target_spatial = Luminance(shift.target);
        if (target_spatial > 0)
        {
            float targetLum_curr = r_curr.W > 0 ? r_curr.w_sum / r_curr.W : 0;
            float jacobian = r_curr.rc.partialJacobian > 0 ? shift.partialJacobian / r_curr.rc.partialJacobian : 0;
            float numerator = (float)r_curr.M * targetLum_curr;
            float denom = numerator + (float)r_spatial.M * target_spatial * jacobian;
            float m_curr = denom > 0 ? numerator / denom : 0;
            r_curr.w_sum *= m_curr;
        }
        F.prev.B[2 * px] = r_curr.w_sum;
    }
}

// K16 Reconnect_StC (ReSTIR_PT_Reconnect_StC.hlsl:112-352), cut at its four WaveActiveSum points.
// ---- K12 thread maps (ReSTIR_PT_Sort.hlsl; Util.hlsli:20-42 EncodeSorted / DecodeSorted): the entry at a thread's position holds the offset of
// the pixel that thread shifts (6 + 6 bits biased by 31, inside the 32 x 32 tile) and an error bit (nothing to do for this thread)
enum { RPT_SORT_CTT = 0, RPT_SORT_TTC = 1, RPT_SORT_CTS = 2, RPT_SORT_STC = 3 };
static constexpr uint32_t kSeInvalidPixel = 1, kSeNotFound = 2, kSeEmpty = 4;      // RPT_Util::SHIFT_ERROR, Shift.hlsli:8-14
ZR_HD uint16_t EncodeSorted(uint32_t x, uint32_t y, uint32_t mx, uint32_t my, uint32_t error)
{ return (uint16_t)(((uint32_t)((int)x - (int)mx + 31)) | (((uint32_t)((int)y - (int)my + 31)) << 7) | ((error ? 1u : 0u) << 15)); }
// false: the map says this thread has no pixel
ZR_HD bool DecodeSorted(uint16_t e, uint32_t& x, uint32_t& y)
{
    if (e & 0x8000u) return false;
    x = (uint32_t)((int)x + (int)(e & 0x3fu) - 31); y = (uint32_t)((int)y + (int)((e >> 7) & 0x3fu) - 31);
    return true;
}
// One pixel of the sort: FindNeighbor + the reservoir metadata the variant buckets by (ReSTIR_PT_Sort.hlsl:23-64, 139-226).
// Returns the bucket (0..3 = k == 2, 3, 4, >= 5 / edge case; 4 = skip) and the SHIFT_ERROR bits in `result`.
ZR_HD uint32_t SortClassify(const RptFrame& F, const zr_frame_constants& g, int variant, uint32_t x, uint32_t y, bool againstEdge, uint32_t& result)
{
    const uint32_t W = g.render_width, H = g.render_height;
    uint32_t err = 0; int nx = 0, ny = 0;
    if (x >= W || y >= H || !InPlanes(F.gb, (int)x, (int)y)) err = kSeInvalidPixel;
    else
    {
        const size_t px = Pix(F.gb, x, y);
        const GFlags flags = DecodeFlags(F.gb.mr[px]);
        if (flags.invalid || flags.emissive) err = kSeInvalidPixel;
        else if (variant == RPT_SORT_TTC)
        {
            const V2 motionVec = DecodeMotion(F.gb.motion[px]);
            const float rw = (float)W, rh = (float)H;
            const float pu = ((float)x + 0.5f) / rw - motionVec.x, pv = ((float)y + 0.5f) / rh - motionVec.y;
            nx = (int)(pu * rw); ny = (int)(pv * rh);
            if (pu < 0.0f || pv < 0.0f || pu > 1.0f || pv > 1.0f) err = kSeNotFound;
        }
        else if (variant == RPT_SORT_STC)
        {
            const uint32_t ox = F.tex.neighbor[2 * px], oy = F.tex.neighbor[2 * px + 1];
            if (ox == 255u) err = kSeNotFound;
            nx = (int)ox - kNeighborOffset + (int)x; ny = (int)oy - kNeighborOffset + (int)y;
        }
    }
    bool skip = err != 0;
    result = err;
    uint32_t k = Reconnection::EMPTY;
    if (err == 0)
    {
        const bool fromNeighbor = variant == RPT_SORT_TTC || variant == RPT_SORT_STC;
        const int rx = fromNeighbor ? nx : (int)x, ry = fromNeighbor ? ny : (int)y;
        // a load outside the render target returns 0 (k field 0 = reconnection at k = 2); outside this device's planes (a temporal
        // neighbour beyond the apron of a screen tile) the same -- the temporal maps only schedule, they cannot change results
        uint32_t a = 0;
        if (rx >= 0 && ry >= 0 && rx < (int)W && ry < (int)H && InPlanes(F.gb, rx, ry))
            a = (variant == RPT_SORT_TTC ? F.prev.A : F.cur.A)[Pix(F.gb, (uint32_t)rx, (uint32_t)ry)];
        const uint32_t kk = a & 0xfu;
        k = kk == Reconnection::EMPTY ? kk : kk + 2u;
    }
    if (k == Reconnection::EMPTY) { result |= kSeEmpty; skip = true; }
    // valid threads of groups at the right / bottom image boundary must not end up outside the screen (ReSTIR_PT_Sort.hlsl:203-219):
    // they join the k >= 5 bucket (their k is EMPTY = 15 here) with the error cleared
    if (skip && againstEdge && x < W && y < H) { result = 0; skip = false; }
    if (skip) return 4u;
    if (k == 2u) return 0u;
    if (k == 3u) return 1u;
    if (k == 4u) return 2u;
    return 3u;      // k >= 5 or the edge case
}
ZR_HD uint32_t SortErrorBits(int variant, uint32_t result, bool spatialResample)
{
    if (variant == RPT_SORT_TTC) return result & (spatialResample ? (kSeInvalidPixel | kSeNotFound) : kSeInvalidPixel);
    if (variant == RPT_SORT_STC) return result & kSeInvalidPixel;
    return result & (kSeInvalidPixel | kSeEmpty);
}

struct StcLane
{
    // (kept small across the heavy calls: the pixel's surface is rebuilt inside phase 2 and r_curr is loaded after the CtS shift, so
    // neither is live -- i.e. spilled -- across ReconnectCtSPixel; CtS writes only the *other* reservoir set's w_sum)
    bool valid, hasN, spatialEmpty, resample, changed; size_t px, sp; Reservoir r_curr, r_spatial; uint32_t M_max, M_new; GFlags flags;
    uint32_t x, y; float w_sum_loaded;
};
ZR_HD void StcCopyToNextFrame(const RptFrame& F, size_t px, Reservoir& r, uint32_t M_max)
{
    if (!r.rc.Empty()) { r.Load_Reconnection(F.cur, px, F.prm.emissive != 0); r.Write(F.prev, px, M_max, F.prm.emissive != 0); }
    else r.WriteReservoirData(F.prev, px, M_max);
}
ZR_HD void StcSuppress(float waveAvgExclusive, Reservoir& r)
{ if (r.w_sum > 50 * waveAvgExclusive) { r.M = 0; r.w_sum = 0; r.W = 0; r.rc.Clear(); } }

// phase 0: classify; contributes (sum1, sum2)
ZR_HD void StcPhase0(const RptFrame& F, const zr_frame_constants& g, uint32_t x, uint32_t y, StcLane& a, float& s1, float& s2)
{
    a.valid = a.hasN = a.spatialEmpty = a.resample = a.changed = false; a.x = x; a.y = y; s1 = 0; s2 = 0;
    if (!F.Owns(x, y)) return;
    a.px = Pix(F.gb, x, y);
    a.flags = DecodeFlags(F.gb.mr[a.px]);
    if (a.flags.invalid || a.flags.emissive) return;
    a.valid = true;
    a.w_sum_loaded = F.cur.B[2 * a.px];
    int sx, sy;
    a.hasN = NeighborOf(F, x, y, sx, sy);
    if (a.hasN) a.sp = Pix(F.gb, (uint32_t)sx, (uint32_t)sy);
    s1 = a.w_sum_loaded;
    s2 = a.w_sum_loaded * (a.hasN ? 0.0f : 1.0f);
}
// phase 1: lanes without a neighbour finish; the others load the spatial reservoir; contributes sum3
ZR_HD void StcPhase1(const RptFrame& F, const zr_frame_constants& g, StcLane& a, float sum1, float& s3)
{
    s3 = 0;
    if (!a.valid) return;
    a.r_curr = Load_NonReconnection(F.cur, a.px);
    a.r_curr.target = xyz(F.tex.target[a.px]);
    const float waveAvgExclusive = (sum1 - a.w_sum_loaded) / 64.0f;
    a.M_max = F.prm.M_max_spatial;
    a.M_max = !a.r_curr.rc.Empty() && a.r_curr.rc.lobe_k_min_1 == LOBE_GLOSSY_T ? umin(a.M_max, kMmaxXkTransmissive) : a.M_max;
    if (!a.hasN)
    {
        if (F.prm.boiling) StcSuppress(waveAvgExclusive, a.r_curr);
        WriteOutputColor(g, F.finalRGBA, a.px, a.r_curr.target * a.r_curr.W);
        StcCopyToNextFrame(F, a.px, a.r_curr, a.M_max);
        return;
    }
    a.r_spatial = Load_NonReconnection(F.cur, a.sp);
    if ((a.r_curr.w_sum != 0) && (a.r_spatial.M > 0) && !a.r_curr.rc.Empty()) a.r_curr.w_sum = F.prev.B[2 * a.px];
    a.M_new = (a.r_curr.M + a.r_spatial.M) & 0xffffu;
    a.spatialEmpty = a.r_spatial.rc.Empty();
    s3 = a.r_curr.w_sum * (a.spatialEmpty ? 1.0f : 0.0f);
}
// phase 2: lanes whose neighbour is empty finish; the others shift + resample; contributes sum4
ZR_HD void StcPhase2(const RptFrame& F, const zr_frame_constants& g, StcLane& a, float sum1, TravStack stack, uint32_t* cnt, float& s4)
{
    s4 = 0;
    if (!a.valid || !a.hasN) return;
    const float waveAvgExclusive = (sum1 - a.w_sum_loaded) / 64.0f;
    if (a.spatialEmpty)
    {
        if (F.prm.boiling) StcSuppress(waveAvgExclusive, a.r_curr);
        float targetLum = Luminance(a.r_curr.target);
        a.r_curr.W = targetLum > 0 ? a.r_curr.w_sum / targetLum : 0;
        a.r_curr.M = a.M_new;
        StcCopyToNextFrame(F, a.px, a.r_curr, a.M_max);
        WriteOutputColor(g, F.finalRGBA, a.px, a.r_curr.target * a.r_curr.W);
        return;
    }
    a.resample = true;
    a.M_max = a.r_spatial.rc.x_k_in_motion ? umin(a.M_max, kMmaxXkInMotion) : a.M_max;
    a.r_spatial.rc.x_k_in_motion = false;
    a.r_spatial.Load_Reconnection(F.cur, a.sp, F.prm.emissive != 0);
    Globals gl = MakeGlobals(F, g, a.flags.transmissive, stack, cnt);
    const Camera cam = CurrCamera(g);
    const PixelSurface ps = LoadPixelSurface(F.gb, cam, a.x, a.y, g.frame_num, a.px);
    OffsetPath shift = Shift2(gl, true, a.px, ps.pos, ps.normal, ps.eta_next, ps.surface, a.r_spatial.rc, F.rbNtC,
        MakePrimaryDiffs(cam, (int)a.x, (int)a.y, ps, F.gb, a.px));
    float targetLum_curr = Luminance(shift.target);
    float targetLum_spatial = a.r_spatial.W > 0 ? a.r_spatial.w_sum / a.r_spatial.W : 0;
    float jacobian = a.r_spatial.rc.partialJacobian > 0 ? shift.partialJacobian / a.r_spatial.rc.partialJacobian : 0;
    if (targetLum_curr > 1e-6f && jacobian > 1e-5f && jacobian < 100)
    {
        uint32_t hx = a.x, hy = a.y, hz = a.y; zr_pcg3d(&hx, &hy, &hz);
        Rng rng = Rng::Init(hx, hz, g.frame_num + 511);
        float numerator = (float)a.r_spatial.M * targetLum_spatial;
        float denom = numerator / jacobian + (float)a.r_curr.M * targetLum_curr;
        float m_spatial = denom > 0 ? numerator / denom : 0;
        float w_spatial = m_spatial * a.r_spatial.W * targetLum_curr;
        if (a.r_curr.Update(w_spatial, shift.target, a.r_spatial.rc, rng)) { a.r_curr.rc.partialJacobian = shift.partialJacobian; a.changed = true; }
    }
    float targetLum = Luminance(a.r_curr.target);
    a.r_curr.W = targetLum > 0 ? a.r_curr.w_sum / targetLum : 0;
    a.r_curr.M = a.M_new;
    a.M_max = (a.changed && shift.surfKMin1Transmissive) ? umin(a.M_max, kMmaxXkTransmissive) : a.M_max;
    s4 = a.r_curr.w_sum;
}
// phase 3: boiling suppression with the full wave sum, write
ZR_HD void StcPhase3(const RptFrame& F, const zr_frame_constants& g, StcLane& a, float waveSum)
{
    if (!a.resample) return;
    if (F.prm.boiling)
    {
        float waveAvgExclusive = (waveSum - a.r_curr.w_sum) / 64.0f;
        StcSuppress(waveAvgExclusive, a.r_curr);
    }
    if (a.changed) a.r_curr.Write(F.prev, a.px, a.M_max, F.prm.emissive != 0);
    else StcCopyToNextFrame(F, a.px, a.r_curr, a.M_max);
    WriteOutputColor(g, F.finalRGBA, a.px, a.r_curr.target * a.r_curr.W);
}

// canonical 64-lane sum (xor butterfly, strides 1, 2, 4, 8, 16, 32): what the device computes with DPP / shuffles
ZR_HD float ButterflySum64(float* v)
{
    for (int s = 1; s < 64; s <<= 1)
    {
        float t[64];
        for (int i = 0; i < 64; i++) t[i] = v[i] + v[i ^ s];
        for (int i = 0; i < 64; i++) v[i] = t[i];
    }
    return v[0];
}

} // namespace rpt
} // namespace zr
