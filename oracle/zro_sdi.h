// ORACLE -- test infrastructure only (see zro_math.h header).  PARITY UNPINNED against the reference (no executable
// reference exists for this path); follows the shaders line by line.
//
// zro_sdi.h: CPU restatement of ReSTIR DI for sun + sky (K7 / K8):
//   DirectLighting/Sky/SkyDI_Temporal.hlsl:27-303, SkyDI_Spatial.hlsl:20-135, Resampling.hlsli:10-410, PairwiseMIS.hlsli:11-157,
//   Reservoir.hlsli:9-184, Params.hlsli, Util.hlsli; host order SkyDI.cpp:135-259 (defaults :81-82, SkyDI.h:86-92);
//   Light::SunSample::get LightSource.hlsli:218-262; BSDF::IsLobeValid / LobeAlpha BSDF.hlsli:864-895;
//   Math::WorldToTangentFrame / FromTangentFrameToWorld Math.hlsli:308-322.
// Restated as is: Reservoir::Load leaves partialJacobian at Init()'s 1 (the spatial pass then divides by it);
// mul(float3x3(b1, b2, n), w) = three dots, mul(w, float3x3(b1, b2, n)) = w.x * b1 + w.y * b2 + w.z * n (left to right).
#pragma once
#include "zro_rdi.h"

namespace zro {
namespace SDI {

using RPT::GBufRead; using RPT::GFlags; using RPT::DecodeFlags; using RPT::Roughness; using RPT::DecodeMotion;
using RPT::Camera; using RPT::CurrCamera; using RPT::PrevCamera; using RPT::PixelSurface; using RPT::LoadPixelSurface; using RPT::LoadPixelSurfaceEx;
using Light::TYPE; using BSDF::LOBE;

static const float MAX_PLANE_DIST_REUSE = 5e-1f, MAX_ROUGHNESS_DIFF_REUSE = 0.1f;
static const int NUM_SPATIAL_SAMPLES = 2;
static const float SPATIAL_SEARCH_RADIUS = 16.0f;

static inline float3 WorldToTangentFrame(float3 normal, float3 w)
{ Math::CoordinateSystem onb = Math::CoordinateSystem::Build(normal); return f3(dot(onb.b1, w), dot(onb.b2, w), dot(normal, w)); }
static inline float3 FromTangentFrameToWorld(float3 normal, float3 w_local)
{ Math::CoordinateSystem onb = Math::CoordinateSystem::Build(normal); return w_local.x * onb.b1 + w_local.y * onb.b2 + w_local.z * normal; }

static inline bool IsLobeValid(const BSDF::ShadingData& surface, LOBE lt)
{
    if (lt == LOBE::ALL) return true;
    if (surface.metallic && (lt != LOBE::GLOSSY_R) && (lt != LOBE::COAT)) return false;
    if (!surface.specTr && (lt == LOBE::GLOSSY_T)) return false;
    if (surface.specTr && (lt == LOBE::DIFFUSE_R)) return false;
    if (!surface.ThinWalled() && (lt == LOBE::DIFFUSE_T)) return false;
    if (!surface.Coated() && (lt == LOBE::COAT)) return false;
    return true;
}
static inline float LobeAlpha(const BSDF::ShadingData& surface, LOBE lt)
{
    if (lt == LOBE::GLOSSY_R || lt == LOBE::GLOSSY_T) return surface.alpha;
    if (lt == LOBE::COAT) return surface.coat_alpha;
    return 1.0f;
}

struct SkyFunc { const SkyLUT* lut; float3 operator()(float3 w) const { return Light::Le_Sky(w, *lut); } };

// Reservoir.hlsli:9-184.  Planes: A R8_UINT metadata, B RG16_UINT oct32(wx), C RG32F (w_sum, W)
struct Reservoir
{
    float w_sum, W; float3 wx, target; float partialJacobian; bool halfVectorCopyShift; uint16_t M; LOBE lobe; TYPE lightType;
    static Reservoir Init()
    {
        Reservoir r; r.M = 0; r.w_sum = 0; r.W = 0; r.wx = f3(0.0f); r.target = f3(0.0f); r.lightType = TYPE::NONE; r.partialJacobian = 1;
        r.halfVectorCopyShift = false; r.lobe = LOBE::ALL; return r;
    }
    static Reservoir Load(const uint8_t* A, const uint16_t* B, const float* C, size_t i)
    {
        Reservoir ret = Init();
        const uint32_t metadata = A[i];
        ret.M = (uint16_t)(metadata & 0xf);
        if (!((metadata >> 7) > 0)) return ret;
        ret.lightType = ((metadata >> 4) & 0x1) ? TYPE::SKY : TYPE::SUN;
        ret.halfVectorCopyShift = (metadata >> 5) & 0x1;
        const bool lobeIsCoat = (metadata >> 6) & 0x1;
        ret.lobe = ret.halfVectorCopyShift ? (lobeIsCoat ? LOBE::COAT : LOBE::GLOSSY_R) : LOBE::ALL;
        const uint16_t e[2] = {B[2 * i], B[2 * i + 1]};
        ret.wx = Math::DecodeOct32(e);
        ret.w_sum = C[2 * i]; ret.W = C[2 * i + 1];
        return ret;
    }
    bool IsValid() const { return w_sum > 0; }
    // initial BSDF candidate: wi -> (optionally) local half vector
    bool Update(float weight, float3 wi, float3 wo, float3 normal, TYPE lt, LOBE lb, bool halfVecShift, float3 target_, RNG& rng)
    {
        if (zr_isnan(weight)) return false;
        M += 1;
        if (weight == 0) return false;
        w_sum += weight;
        if (rng.Uniform() < (weight / w_sum))
        {
            target = target_; lightType = lt; lobe = lb; halfVectorCopyShift = halfVecShift;
            if (halfVecShift)
            {
                float3 wh = normalize(wo + wi);
                wx = WorldToTangentFrame(normal, wh);
                partialJacobian = zr_abs(dot(wh, wo));
            }
            else wx = wi;
            return true;
        }
        return false;
    }
    bool Update(float weight, float3 wi_or_wh, TYPE lt, LOBE lb, bool halfVecShift, float whdotwo, float3 target_, RNG& rng)
    {
        if (zr_isnan(weight)) return false;
        M += 1;
        if (weight == 0) return false;
        w_sum += weight;
        if (rng.Uniform() < (weight / w_sum))
        { wx = wi_or_wh; target = target_; lightType = lt; lobe = lb; halfVectorCopyShift = halfVecShift; partialJacobian = whdotwo; return true; }
        return false;
    }
    void Write(uint8_t* A, uint16_t* B, float* C, size_t i, uint32_t M_max) const
    {
        const uint32_t M_capped = std::min<uint32_t>(M, M_max) & 0xf;
        const bool wSumGt0 = w_sum > 0;
        A[i] = (uint8_t)(M_capped | ((uint32_t)(lightType == TYPE::SKY) << 4) | ((uint32_t)halfVectorCopyShift << 5) |
            ((uint32_t)(lobe == LOBE::COAT) << 6) | ((uint32_t)wSumGt0 << 7));
        if (!wSumGt0) return;
        uint16_t e[2]; Math::EncodeOct32(wx, e);
        B[2 * i] = e[0]; B[2 * i + 1] = e[1];
        C[2 * i] = w_sum; C[2 * i + 1] = W;
    }
};

static inline bool IsShiftInvertible(const Reservoir& r_base, const BSDF::ShadingData& surface_offset, float alpha_min)
{ return !r_base.halfVectorCopyShift || (IsLobeValid(surface_offset, r_base.lobe) && (LobeAlpha(surface_offset, r_base.lobe) <= alpha_min)); }

static inline float3 LightLe(const Scene& sc, const zr_frame_constants& g, TYPE lt, float3 wi, float3 pos)
{ return lt == TYPE::SKY ? Light::Le_Sky(wi, sc.sky) : Light::Le_Sun(pos, g); }

// SkyDI_Temporal.hlsl:27-128
static Reservoir RIS_InitialCandidates(const Scene& sc, const zr_frame_constants& g, float alpha_min, float3 pos, float3 normal,
    BSDF::ShadingData surface, RNG& rng)
{
    Reservoir r = Reservoir::Init();
    SkyFunc leFunc; leFunc.lut = &sc.sky;
    const float3 sunDir = f3(g.sun_dir);
    {
        // Light::SunSample::get(-SunDir, cosAngularRadius, normal, surface, rng), LightSource.hlsli:230-256
        float3 sun_f = f3(0.0f), wi_s = f3(0.0f);
        {
            const float3 toSun = -sunDir;
            const float ndotSunDir = dot(toSun, normal);
            if (!(ndotSunDir < 0 && !surface.Transmissive()))
            {
                float pdf_light;
                float3 sampleLocal = Sampling::UniformSampleCone(rng.Uniform2D(), g.sun_cos_angular_radius, pdf_light);
                Math::CoordinateSystem onb = Math::CoordinateSystem::Build(toSun);
                float3 wi_light = mad3(sampleLocal.x, onb.b1, mad3(sampleLocal.y, onb.b2, sampleLocal.z * toSun));
                surface.SetWi(wi_light, normal);
                sun_f = BSDF::Unified(surface).f;
                wi_s = wi_light;
            }
        }
        float3 target = f3(0.0f);
        const bool trace = (wi_s.y > 0) && ((dot(wi_s, normal) > 0) || surface.Transmissive()) && (dot(wi_s, -sunDir) >= g.sun_cos_angular_radius);
        if (trace && (dot(sun_f, sun_f) > 0))
        {
            if (RtRayQuery::Visibility_Ray(sc, pos, wi_s, normal, surface.Transmissive())) target = Light::Le_Sun(pos, g) * sun_f;
        }
        const float targetLum = Math::Luminance(target);
        float ndotwi = zr_saturate(dot(wi_s, normal));
        const float pdf_e = ndotwi * ZR_ONE_OVER_PI;
        const float pdf_s = BSDF::BSDFSamplerPdf(normal, surface, wi_s, leFunc, rng);
        const float w_s = RT::BalanceHeuristic3(1, pdf_e, pdf_s, targetLum);
        r.Update(w_s, wi_s, TYPE::SUN, LOBE::ALL, false, 1, target, rng);
    }
    const bool specular = surface.GlossSpecular() && (surface.metallic || surface.specTr) && (!surface.Coated() || surface.CoatSpecular());
    if (!specular)
    {
        const float2 u = rng.Uniform2D();
        float pdf_e;
        float3 wi_e = BSDF::SampleDiffuse(normal, u, pdf_e);
        const float3 le = Light::Le_Sky(wi_e, sc.sky);
        surface.SetWi(wi_e, normal);
        float3 target = le * BSDF::Unified(surface).f;
        if (dot(target, target) > 0) target = target * (RtRayQuery::Visibility_Ray(sc, pos, wi_e, normal, surface.Transmissive()) ? 1.0f : 0.0f);
        const float targetLum = Math::Luminance(target);
        const float pdf_b = BSDF::BSDFSamplerPdf(normal, surface, wi_e, leFunc, rng);
        const float denom = pdf_e + pdf_b;                                   // RT::BalanceHeuristic(pdf_e, pdf_b, targetLum)
        const float w_e = denom == 0 ? 0.0f : targetLum / denom;
        r.Update(w_e, wi_e, TYPE::SKY, LOBE::ALL, false, 1, target, rng);
    }
    {
        BSDF::BSDFSample bsdfSample = BSDF::SampleBSDF(normal, surface, leFunc, rng);
        float3 wi_b = bsdfSample.wi;
        float pdf_b = bsdfSample.pdf;
        float3 target = bsdfSample.f;
        if (dot(target, target) > 0) target = target * (RtRayQuery::Visibility_Ray(sc, pos, wi_b, normal, surface.Transmissive()) ? 1.0f : 0.0f);
        const float targetLum = Math::Luminance(target);
        float ndotwi = zr_saturate(dot(wi_b, normal));
        const float pdf_e = ndotwi * ZR_ONE_OVER_PI;
        const float denom = pdf_b + pdf_e;
        const float w_b = denom == 0 ? 0.0f : targetLum / denom;
        const bool useHalfVecShift = LobeAlpha(surface, bsdfSample.lobe) <= alpha_min;
        r.Update(w_b, wi_b, surface.wo, normal, TYPE::SKY, bsdfSample.lobe, useHalfVecShift, target, rng);
    }
    float targetLum = Math::Luminance(r.target);
    r.W = targetLum > 0.0f ? r.w_sum / targetLum : 0.0f;
    return r;
}

// Resampling.hlsli:11-145
struct TemporalCandidate { BSDF::ShadingData surface; float3 pos, normal; int px, py; bool valid; };
static TemporalCandidate FindTemporalCandidate(const zr_frame_constants& g, const GBufRead& gbPrev, float3 pos, float3 normal, float z_view,
    float roughness, const BSDF::ShadingData& surface, float2 prevUV)
{
    TemporalCandidate c; c.valid = false; c.px = c.py = 0;
    const float2 renderDim = {(float)g.render_width, (float)g.render_height};
    if (prevUV.x < 0 || prevUV.y < 0 || prevUV.x > 1 || prevUV.y > 1) return c;
    const int ppx = (int)(prevUV.x * renderDim.x), ppy = (int)(prevUV.y * renderDim.y);
    if (ppx >= (int)gbPrev.w || ppy >= (int)gbPrev.h) return c;        // prevUV == 1: out-of-bounds texel reads 0 = invalid flags
    const size_t pp = (size_t)ppy * gbPrev.w + ppx;
    GFlags pf = DecodeFlags(gbPrev.mr[pp]);
    const float prevRoughness = Roughness(gbPrev.mr[pp]);
    if (pf.invalid || pf.emissive || (zr_abs(prevRoughness - roughness) > 0.3f) || (pf.metallic != surface.metallic) ||
        (pf.transmissive != surface.specTr)) return c;
    const Camera pcam = PrevCamera(g);
    PixelSurface ps = LoadPixelSurface(gbPrev, pcam, (uint32_t)ppx, (uint32_t)ppy, g.frame_num - 1, pp);
    float planeDist = dot(normal, ps.pos - pos);
    if (!(zr_abs(planeDist) <= MAX_PLANE_DIST_REUSE * z_view)) return c;
    c.surface = ps.surface; c.pos = ps.pos; c.normal = ps.normal; c.px = ppx; c.py = ppy; c.valid = true;
    return c;
}

// Resampling.hlsli:147-249
static void TemporalResample(const Scene& sc, const zr_frame_constants& g, TemporalCandidate candidate, float3 pos, float3 normal,
    BSDF::ShadingData surface, const Reservoir& r_prevLoaded, float alpha_min, Reservoir& r, RNG& rng)
{
    Reservoir r_prev = r_prevLoaded;
    r_prev.M = (r.lightType == TYPE::SUN) && g.sun_moved ? (uint16_t)0 : r_prev.M;
    const uint16_t newM = (uint16_t)(r.M + r_prev.M);
    if (r.w_sum != 0)
    {
        float targetLum_prev = 0;
        float3 wi_offset = r.wx;
        float jacobian = 1;
        if (IsShiftInvertible(r, candidate.surface, alpha_min))
        {
            if (r.halfVectorCopyShift)
            {
                float3 wh_t = FromTangentFrameToWorld(candidate.normal, r.wx);
                wi_offset = reflect(-candidate.surface.wo, wh_t);
                jacobian = r.partialJacobian == 0 ? 0 : zr_abs(dot(candidate.surface.wo, wh_t)) / r.partialJacobian;
            }
            candidate.surface.SetWi(wi_offset, candidate.normal);
            const float3 le = LightLe(sc, g, r.lightType, wi_offset, candidate.pos);
            const float3 target_prev = le * BSDF::Unified(candidate.surface).f;
            targetLum_prev = Math::Luminance(target_prev);
            if (targetLum_prev > 0)      // g_bvh_prev: static scenes, same BVH
                targetLum_prev *= RtRayQuery::Visibility_Ray(sc.Prev(), candidate.pos, wi_offset, candidate.normal, candidate.surface.Transmissive()) ? 1.0f : 0.0f;
        }
        const float numerator = (float)r.M * Math::Luminance(r.target);
        const float denom = numerator + (float)r_prev.M * targetLum_prev * jacobian;
        const float m_curr = denom > 0 ? numerator / denom : 0;
        r.w_sum *= m_curr;
    }
    if (r_prev.IsValid() && r_prev.M > 0)
    {
        float3 wi_offset = r_prev.wx;
        float jacobian = 1;
        float3 target_curr = f3(0.0f);
        if (IsShiftInvertible(r_prev, surface, alpha_min))
        {
            if (r_prev.halfVectorCopyShift)
            {
                float3 wh_c = FromTangentFrameToWorld(normal, r_prev.wx);
                float3 wh_t = FromTangentFrameToWorld(candidate.normal, r_prev.wx);
                wi_offset = reflect(-surface.wo, wh_c);
                float whdotwo_t = zr_abs(dot(candidate.surface.wo, wh_t));
                jacobian = whdotwo_t > 0 ? zr_abs(dot(surface.wo, wh_c)) / whdotwo_t : 1;
            }
            surface.SetWi(wi_offset, normal);
            const float3 le = LightLe(sc, g, r_prev.lightType, wi_offset, pos);
            target_curr = le * BSDF::Unified(surface).f;
        }
        if (dot(target_curr, target_curr) > 0)
        {
            if (RtRayQuery::Visibility_Ray(sc, pos, wi_offset, normal, surface.Transmissive()))
            {
                const float targetLum_curr = Math::Luminance(target_curr);
                const float targetLum_prev = r_prev.W > 0 ? r_prev.w_sum / r_prev.W : 0;
                const float numerator = (float)r_prev.M * targetLum_prev;
                const float denom = numerator / jacobian + (float)r.M * targetLum_curr;
                const float m_prev = denom > 0 ? numerator / denom : 0;
                const float w_prev = m_prev * targetLum_curr * r_prev.W;
                r.Update(w_prev, r_prev.wx, r_prev.lightType, r_prev.lobe, r_prev.halfVectorCopyShift, surface.whdotwo, target_curr, rng);
            }
        }
    }
    float targetLum = Math::Luminance(r.target);
    r.W = targetLum > 0.0f ? r.w_sum / targetLum : 0.0f;
    r.M = newM;
}

// PairwiseMIS.hlsli:11-157
struct PairwiseMIS
{
    Reservoir r_s; float m_c; uint16_t M_s, k;
    static PairwiseMIS Init(uint16_t numStrategies, const Reservoir& r_c)
    { PairwiseMIS p; p.r_s = Reservoir::Init(); p.m_c = 1.0f; p.M_s = r_c.M; p.k = numStrategies; return p; }
    float Compute_m_i(const Reservoir& r_c, float targetLum, const Reservoir& r_i, float jacobian) const
    {
        const float p_i_y_i = r_i.W > 0 ? r_i.w_sum / r_i.W : 0;
        float numerator = (float)r_i.M * p_i_y_i;
        float denom = (numerator / jacobian) + ((float)r_c.M / (float)k) * targetLum;
        return denom > 0 ? numerator / denom : 0;
    }
    void Update_m_c(const Reservoir& r_c, const Reservoir& r_i, float targetLum, float jacobian)
    {
        const float p_c_y_c = Math::Luminance(r_c.target);
        const float numerator = (float)r_i.M * targetLum * jacobian;
        const float denom = numerator + ((float)r_c.M / (float)k) * p_c_y_c;
        m_c += 1 - (numerator / denom);
    }
    void Stream(const Scene& sc, const zr_frame_constants& g, const Reservoir& r_c, float3 pos_c, float3 normal_c, BSDF::ShadingData surface_c,
        const Reservoir& r_i, float3 pos_i, float3 normal_i, BSDF::ShadingData surface_i, float alpha_min, RNG& rng)
    {
        float m_i = 0;
        float3 target_c_y_i = f3(0.0f);
        if (r_i.IsValid())
        {
            float3 wi_offset = r_i.wx;
            float jacobian = 1;
            if (IsShiftInvertible(r_i, surface_c, alpha_min))
            {
                if (r_i.halfVectorCopyShift)
                {
                    float3 wh_c = FromTangentFrameToWorld(normal_c, r_i.wx);
                    float3 wh_i = FromTangentFrameToWorld(normal_i, r_i.wx);
                    wi_offset = reflect(-surface_c.wo, wh_c);
                    float whdotwo_i = zr_abs(dot(surface_i.wo, wh_i));
                    jacobian = whdotwo_i > 0 ? zr_abs(dot(surface_c.wo, wh_c)) / whdotwo_i : 1;
                }
                surface_c.SetWi(wi_offset, normal_c);
                const float3 le = LightLe(sc, g, r_i.lightType, wi_offset, pos_c);
                target_c_y_i = le * BSDF::Unified(surface_c).f;
                if (dot(target_c_y_i, target_c_y_i) > 0)
                    target_c_y_i = target_c_y_i * (RtRayQuery::Visibility_Ray(sc, pos_c, wi_offset, normal_c, surface_c.Transmissive()) ? 1.0f : 0.0f);
            }
            const float targetLum = Math::Luminance(target_c_y_i);
            m_i = Compute_m_i(r_c, targetLum, r_i, jacobian);
        }
        float3 target_i_y_c = f3(0.0f);
        float jacobian = 1;
        if (r_c.IsValid())
        {
            float3 wi_offset = r_c.wx;
            if (IsShiftInvertible(r_c, surface_i, alpha_min))
            {
                if (r_c.halfVectorCopyShift)
                {
                    float3 wh_i = FromTangentFrameToWorld(normal_i, r_c.wx);
                    wi_offset = reflect(-surface_i.wo, wh_i);
                    float whdotwo_i = zr_abs(dot(surface_i.wo, wh_i));
                    jacobian = whdotwo_i > 0 ? zr_abs(dot(surface_i.wo, wh_i)) / r_c.partialJacobian : 1;
                }
                surface_i.SetWi(wi_offset, normal_i);
                const float3 le = LightLe(sc, g, r_c.lightType, wi_offset, pos_i);
                target_i_y_c = le * BSDF::Unified(surface_i).f;
                if (dot(target_i_y_c, target_i_y_c) > 0)
                    target_i_y_c = target_i_y_c * (RtRayQuery::Visibility_Ray(sc, pos_i, wi_offset, normal_i, surface_i.Transmissive()) ? 1.0f : 0.0f);
            }
        }
        const float targetLum = Math::Luminance(target_i_y_c);
        Update_m_c(r_c, r_i, targetLum, jacobian);
        if (r_i.IsValid())
        {
            const float w_i = m_i * Math::Luminance(target_c_y_i) * r_i.W;
            r_s.Update(w_i, r_i.wx, r_i.lightType, r_i.lobe, r_i.halfVectorCopyShift, surface_c.whdotwo, target_c_y_i, rng);
        }
        M_s = (uint16_t)(M_s + r_i.M);
    }
    void End(const Reservoir& r_c, RNG& rng)
    {
        const float w_c = m_c * r_c.w_sum;
        r_s.Update(w_c, r_c.wx, r_c.lightType, r_c.lobe, r_c.halfVectorCopyShift, r_c.partialJacobian, r_c.target, rng);
        r_s.M = M_s;
        const float targetLum = Math::Luminance(r_s.target);
        r_s.W = targetLum > 0 ? r_s.w_sum / (targetLum * (1 + (float)k)) : 0;
    }
};

static const float k_samples[16][2] = {
    {-0.899423f, 0.365076f}, {-0.744442f, -0.124006f}, {-0.229714f, 0.245876f}, {-0.545186f, 0.741148f}, {-0.156274f, -0.336366f},
    {0.468400f, 0.348798f}, {0.035776f, 0.606928f}, {-0.208966f, 0.904852f}, {-0.491070f, -0.484810f}, {0.162490f, -0.081156f},
    {0.232062f, -0.851382f}, {0.641310f, -0.162124f}, {0.320798f, 0.922460f}, {0.959086f, 0.263642f}, {0.531136f, -0.519002f},
    {-0.223014f, -0.774740f}};       // Resampling.hlsli:259-277, `static const half2`: rounded to fp16 on load

struct State
{
    uint32_t w = 0, h = 0;
    std::vector<uint8_t> A[2]; std::vector<uint16_t> B[2]; std::vector<float> C[2]; std::vector<float> target;
    bool temporalValid = false; int currIdx = 0;
    void Resize(uint32_t w_, uint32_t h_)
    {
        w = w_; h = h_; size_t n = (size_t)w * h;
        for (int i = 0; i < 2; i++) { A[i].assign(n, 0); B[i].assign(2 * n, 0); C[i].assign(2 * n, 0); }
        target.assign(4 * n, 0); temporalValid = false; currIdx = 0;
    }
};

static inline float3 Le_SkyWithSunDisk(const Scene& sc, const zr_frame_constants& g, uint32_t x, uint32_t y) { return Light::Le_SkyWithSunDisk(x, y, g, sc.sky); }      // zro_sky.h

// SkyDI::Render (SkyDI.cpp:135-259): K7 over all pixels, then K8
static void Render(const Scene& sc, const zr_frame_constants& g, const zr_gbuffer_planes* gbCurr, const zr_gbuffer_planes* gbPrevPlanes,
    const zr_params& zp, State& st, float* finalRGBA)
{
    BSDF::g_rho = &sc.rhoLUT;
    GBufRead gb(gbCurr);
    const uint32_t W = g.render_width, H = g.render_height;
    const bool doTemporal = st.temporalValid && (zp.flags & ZR_IND_TEMPORAL_RESAMPLE) && gbPrevPlanes;
    const bool doSpatial = doTemporal && (zp.flags & ZR_IND_SPATIAL_RESAMPLE);
    const bool writeReservoirs = doTemporal || !st.temporalValid;       // TEMPORAL_RESAMPLE || RESET_TEMPORAL_TEXTURES
    const float alpha_min = zp.alpha_min;
    const uint32_t M_max_sky = zp.m_max_temporal, M_max_sun = zp.m_max_spatial;   // zr_params: m_max_temporal = M_max (Sky), m_max_spatial = M_max (Sun)
    const Camera cam = CurrCamera(g);
    const uint32_t* emissivePlane = (const uint32_t*)gbCurr->plane[ZR_GB_EMISSIVE_COLOR];
    uint8_t* curA = st.A[st.currIdx].data(); uint16_t* curB = st.B[st.currIdx].data(); float* curC = st.C[st.currIdx].data();
    const uint8_t* prevA = st.A[1 - st.currIdx].data(); const uint16_t* prevB = st.B[1 - st.currIdx].data(); const float* prevC = st.C[1 - st.currIdx].data();
    const bool accumulate = g.accumulate && g.camera_static;

    for (uint32_t y = 0; y < H; y++) for (uint32_t x = 0; x < W; x++)
    {
        const size_t px = (size_t)y * W + x;
        GFlags flags = DecodeFlags(gb.mr[px]);
        float* o = finalRGBA + 4 * px;
        if (flags.invalid)
        {
            if (accumulate)
            {
                const float3 le = Le_SkyWithSunDisk(sc, g, x, y);
                const float k = g.num_frames_camera_static > 1 ? 1.0f : 0.0f;
                o[0] = o[0] * k + le.x; o[1] = o[1] * k + le.y; o[2] = o[2] * k + le.z;
            }
            else { o[0] = o[1] = o[2] = 0; }
            continue;
        }
        if (flags.emissive)
        {
            float3 le = RDI::EmissiveColor(emissivePlane, px);
            if (accumulate) { o[0] += le.x; o[1] += le.y; o[2] += le.z; }
            else { o[0] = le.x; o[1] = le.y; o[2] = le.z; }
            continue;
        }
        PixelSurface ps = LoadPixelSurface(gb, cam, x, y, g.frame_num, px);
        uint32_t hx = y, hy = x, hz = x; zr_pcg3d(&hx, &hy, &hz);                     // RNG::PCG3d(DTid.yxx).yz
        RNG rng = RNG::Init(hy, hz, g.frame_num);
        Reservoir r = RIS_InitialCandidates(sc, g, alpha_min, ps.pos, ps.normal, ps.surface, rng);
        if (doTemporal)
        {
            GBufRead gbPrev(gbPrevPlanes);
            float2 motionVec = DecodeMotion(gb.motion[px]);
            const float2 currUV = {((float)x + 0.5f) / (float)W, ((float)y + 0.5f) / (float)H};
            float2 prevUV = currUV - motionVec;
            TemporalCandidate tc = FindTemporalCandidate(g, gbPrev, ps.pos, ps.normal, ps.z, ps.roughness, ps.surface, prevUV);
            if (tc.valid)
            {
                const size_t pp = (size_t)tc.py * W + tc.px;
                Reservoir r_prev = Reservoir::Load(prevA, prevB, prevC, pp);
                TemporalResample(sc, g, tc, ps.pos, ps.normal, ps.surface, r_prev, alpha_min, r, rng);
            }
            if (doSpatial)
            {
                float3 t = r.target;                    // WriteTarget: Math::Sanitize (any NaN / inf -> the whole vector 0)
                if (any_nan(t) || zr_isinf(t.x) || zr_isinf(t.y) || zr_isinf(t.z)) t = f3(0.0f);
                r.target = t;
                // the TARGET texture is R16G16B16A16_FLOAT (SkyDI.h:61): the spatial pass reads fp16-rounded values
                st.target[4 * px] = zr_round_f16(t.x); st.target[4 * px + 1] = zr_round_f16(t.y); st.target[4 * px + 2] = zr_round_f16(t.z);
            }
        }
        if (writeReservoirs) r.Write(curA, curB, curC, px, r.lightType == TYPE::SKY ? M_max_sky : M_max_sun);
        if (!doSpatial)
        {
            float3 ld = r.target * r.W;
            RDI::WriteFinal(g, finalRGBA, px, ld);
        }
    }

    if (doSpatial)
    {
        for (uint32_t y = 0; y < H; y++) for (uint32_t x = 0; x < W; x++)
        {
            const size_t px = (size_t)y * W + x;
            GFlags flags = DecodeFlags(gb.mr[px]);
            if (flags.invalid || flags.emissive) continue;
            PixelSurface ps = LoadPixelSurface(gb, cam, x, y, g.frame_num, px);
            uint32_t hx = y, hy = x, hz = x; zr_pcg3d(&hx, &hy, &hz);
            RNG rng = RNG::Init(hy, hz, g.frame_num);
            Reservoir r_c = Reservoir::Load(curA, curB, curC, px);
            r_c.target = f3(st.target[4 * px], st.target[4 * px + 1], st.target[4 * px + 2]);
            // SpatialResample, Resampling.hlsli:251-409
            const float u0 = rng.Uniform();
            const int offset = (int)rng.UniformUintBounded_Faster(16);
            const float theta = u0 * ZR_TWO_PI;
            const float sinTheta = zr_sin(theta), cosTheta = zr_cos(theta);
            PairwiseMIS pw = PairwiseMIS::Init((uint16_t)NUM_SPATIAL_SAMPLES, r_c);
            struct Cand { uint32_t x, y; } cand[NUM_SPATIAL_SAMPLES];
            uint16_t k = 0;
            for (int i = 0; i < NUM_SPATIAL_SAMPLES; i++)
            {
                const float ux = zr_round_f16(k_samples[(offset + i) & 15][0]), uy = zr_round_f16(k_samples[(offset + i) & 15][1]);
                float rx = ux * cosTheta + uy * -sinTheta, ry = ux * sinTheta + uy * cosTheta;
                rx *= SPATIAL_SEARCH_RADIUS; ry *= SPATIAL_SEARCH_RADIUS;
                const int sx = zr_f2i_sat(__builtin_rintf((float)x + rx)), sy = zr_f2i_sat(__builtin_rintf((float)y + ry));
                if (!(sx >= 0 && sy >= 0 && sx < (int)W && sy < (int)H)) continue;       // Math::IsWithinBounds
                const size_t sp = (size_t)sy * W + sx;
                GFlags fi = DecodeFlags(gb.mr[sp]);
                if (fi.invalid || fi.emissive) continue;
                PixelSurface pi = LoadPixelSurfaceEx(gb, cam, (uint32_t)sx, (uint32_t)sy, g.frame_num, sp, false);
                bool valid = zr_abs(dot(ps.normal, pi.pos - ps.pos)) <= MAX_PLANE_DIST_REUSE * ps.z;
                valid = valid && (zr_abs(pi.roughness - ps.roughness) < MAX_ROUGHNESS_DIFF_REUSE);
                if (!valid) continue;
                cand[k].x = (uint32_t)sx; cand[k].y = (uint32_t)sy; k++;
            }
            pw.k = k;
            for (int i = 0; i < k; i++)
            {
                const size_t sp = (size_t)cand[i].y * W + cand[i].x;
                PixelSurface pi = LoadPixelSurfaceEx(gb, cam, cand[i].x, cand[i].y, g.frame_num, sp, false);
                Reservoir r_spatial = Reservoir::Load(curA, curB, curC, sp);
                pw.Stream(sc, g, r_c, ps.pos, ps.normal, ps.surface, r_spatial, pi.pos, pi.normal, pi.surface, alpha_min, rng);
            }
            pw.End(r_c, rng);
            r_c = pw.r_s;
            float3 ld = r_c.target * r_c.W;
            RDI::WriteFinal(g, finalRGBA, px, ld);
        }
    }
    st.temporalValid = true;
    st.currIdx = 1 - st.currIdx;
}

} // namespace SDI
} // namespace zro
