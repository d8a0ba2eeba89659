// zr_sdi.h -- per-pixel stage functions of ReSTIR DI for sun + sky (K7 temporal, K8 spatial).
//
// Reference (Source/ZetaRenderPass/DirectLighting/Sky/): SkyDI_Temporal.hlsl:27-303, SkyDI_Spatial.hlsl:20-135,
// Resampling.hlsli:10-410, PairwiseMIS.hlsli:11-157, Reservoir.hlsli:9-184, Params.hlsli; host order SkyDI.cpp:135-259;
// Light::SunSample::get LightSource.hlsli:218-262; BSDF::IsLobeValid / LobeAlpha BSDF.hlsli:864-895;
// Math::WorldToTangentFrame / FromTangentFrameToWorld Math.hlsli:308-322; Light::Le_SkyWithSunDisk LightSource.hlsli:176-199.
// Persistent state in the reference's formats: two reservoir sets x (A R8_UINT metadata, B RG16_UINT oct32(wi or local wh),
// C RG32F w_sum / W) = 13 B/px, target RGBA32F.  Every candidate is checked with one any-hit visibility ray; sun samples read
// the 6-step transmittance (Le_Sun), sky samples the sky-view LUT (Le_Sky): ALU + traversal bound, ~3 + 2 + 4 rays per pixel.
// Restated as is: Reservoir::Load leaves partialJacobian at 1 (the spatial pass divides by it).
#pragma once
#include "zr_rdi.h"

namespace zr {
namespace sdi {

using rpt::Pix; using rpt::GFlags; using rpt::DecodeFlags; using rpt::RoughnessOf; using rpt::DecodeMotion; using rpt::Camera;
using rpt::CurrCamera; using rpt::PrevCamera; using rpt::PixelSurface; using rpt::LoadPixelSurface; using rpt::LoadPixelSurfaceEx;

static constexpr float kMaxPlaneDist = 5e-1f, kMaxRoughDiff = 0.1f, kSearchRadius = 16.0f;
static constexpr int kNumSpatial = 2;
enum : uint32_t { LT_NONE = 0, LT_SUN = 1, LT_SKY = 2 };

ZR_HD V3 WorldToTangentFrame(V3 normal, V3 w) { ONB o = BuildONB(normal); return v3(dot(o.b1, w), dot(o.b2, w), dot(normal, w)); }
ZR_HD V3 FromTangentFrameToWorld(V3 normal, V3 wl) { ONB o = BuildONB(normal); return wl.x * o.b1 + wl.y * o.b2 + wl.z * normal; }

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

struct SkyPlanes { uint8_t* A; uint16_t* B; float* C; };

struct Reservoir { float w_sum, W; V3 wx, target; float partialJacobian; bool halfVectorCopyShift; uint32_t M, lobe, lightType; };
ZR_HD Reservoir InitReservoir()
{
    Reservoir r; r.M = 0; r.w_sum = 0; r.W = 0; r.wx = v3(0.0f); r.target = v3(0.0f); r.lightType = LT_NONE; r.partialJacobian = 1;
    r.halfVectorCopyShift = false; r.lobe = LOBE_ALL; return r;
}
ZR_HD Reservoir LoadReservoir(const SkyPlanes& p, size_t i)
{
    Reservoir ret = InitReservoir();
    const uint32_t metadata = p.A[i];
    ret.M = metadata & 0xf;
    if (!((metadata >> 7) > 0)) return ret;
    ret.lightType = ((metadata >> 4) & 0x1) ? LT_SKY : LT_SUN;
    ret.halfVectorCopyShift = (metadata >> 5) & 0x1;
    const bool lobeIsCoat = (metadata >> 6) & 0x1;
    ret.lobe = ret.halfVectorCopyShift ? (lobeIsCoat ? LOBE_COAT : LOBE_GLOSSY_R) : LOBE_ALL;
    ret.wx = DecodeOct32u((uint32_t)p.B[2 * i] | ((uint32_t)p.B[2 * i + 1] << 16));
    ret.w_sum = p.C[2 * i]; ret.W = p.C[2 * i + 1];
    return ret;
}
ZR_HD bool IsValid(const Reservoir& r) { return r.w_sum > 0; }
// initial BSDF candidate: wi -> (optionally) the local half vector
ZR_HD bool UpdateWi(Reservoir& r, float weight, V3 wi, V3 wo, V3 normal, uint32_t lt, uint32_t lb, bool halfVecShift, V3 target, Rng& rng)
{
    if (zr_isnan(weight)) return false;
    r.M += 1;
    if (weight == 0) return false;
    r.w_sum += weight;
    if (rng.Uniform() < (weight / r.w_sum))
    {
        r.target = target; r.lightType = lt; r.lobe = lb; r.halfVectorCopyShift = halfVecShift;
        if (halfVecShift)
        {
            V3 wh = normalize(wo + wi);
            r.wx = WorldToTangentFrame(normal, wh);
            r.partialJacobian = zr_abs(dot(wh, wo));
        }
        else r.wx = wi;
        return true;
    }
    return false;
}
ZR_HD bool Update(Reservoir& r, float weight, V3 wi_or_wh, uint32_t lt, uint32_t lb, bool halfVecShift, float whdotwo, V3 target, Rng& rng)
{
    if (zr_isnan(weight)) return false;
    r.M += 1;
    if (weight == 0) return false;
    r.w_sum += weight;
    if (rng.Uniform() < (weight / r.w_sum))
    { r.wx = wi_or_wh; r.target = target; r.lightType = lt; r.lobe = lb; r.halfVectorCopyShift = halfVecShift; r.partialJacobian = whdotwo; return true; }
    return false;
}
ZR_HD void WriteReservoir(const Reservoir& r, const SkyPlanes& p, size_t i, uint32_t M_max)
{
    const uint32_t M16 = r.M & 0xffffu;
    const uint32_t M_capped = (M16 < M_max ? M16 : M_max) & 0xf;
    const bool wSumGt0 = r.w_sum > 0;
    p.A[i] = (uint8_t)(M_capped | ((uint32_t)(r.lightType == LT_SKY) << 4) | ((uint32_t)r.halfVectorCopyShift << 5) |
        ((uint32_t)(r.lobe == LOBE_COAT) << 6) | ((uint32_t)wSumGt0 << 7));
    if (!wSumGt0) return;
    const V2 e = EncodeUnitVector(r.wx);                      // Math::EncodeOct32
    p.B[2 * i] = (uint16_t)FloatToUNorm16(e.x); p.B[2 * i + 1] = (uint16_t)FloatToUNorm16(e.y);
    p.C[2 * i] = r.w_sum; p.C[2 * i + 1] = r.W;
}
ZR_HD bool IsShiftInvertible(const Reservoir& r_base, const Surface& surface_offset, float alpha_min)
{ return !r_base.halfVectorCopyShift || (IsLobeValid(surface_offset, r_base.lobe) && (LobeAlpha(surface_offset, r_base.lobe) <= alpha_min)); }

struct SkyParams { uint32_t M_max_sky, M_max_sun, accumulate, doTemporal, doSpatial, writeReservoirs; float alpha_min; };
struct SkyFrame
{
    SceneView sc; GBuf gb, gbPrev; SkyPlanes cur, prev; F4* target; float* finalRGBA; SkyParams prm;
    SceneView scPrev;        // previous frame's acceleration structure (g_bvh_prev, SkyDI_Temporal.hlsl:14)
    uint32_t ox0, oy0, ow, oh;       // owned rect (global pixels): the part of the planes this device shades (multi-GPU tile split)
    ZR_HDM bool Owns(uint32_t x, uint32_t y) const { return x >= ox0 && y >= oy0 && x < ox0 + ow && y < oy0 + oh; }
};
struct Ctx { const SceneView* sc; const zr_frame_constants* g; TravStack stack; uint32_t* cnt; const SceneView* scPrev = nullptr; };

// RtRayQuery::Visibility_Ray, RayQuery.hlsli:302-334
ZR_HD bool VisibilityRay(const Ctx& c, V3 origin, V3 wi, V3 normal, bool transmissive)
{
    F4 ro, rd;
    if (!MakeVisibilityRay(origin, wi, normal, transmissive, &ro, &rd)) return false;
    c.cnt[1]++;
    RawHit h = Traverse<true>(*c.sc, xyz(ro), xyz(rd), ro.w, rd.w, ZR_SUBGROUP_ALL, c.stack);
    return h.tri == kInvalidTri;
}
ZR_HD V3 LightLe(const Ctx& c, uint32_t lt, V3 wi, V3 pos) { return lt == LT_SKY ? Le_Sky(wi, c.sc->sky) : Le_Sun(pos, *c.g); }

// SkyDI_Temporal.hlsl:27-128
ZR_HD Reservoir RIS_InitialCandidates(const Ctx& c, float alpha_min, V3 pos, V3 normal, Surface surface, Rng& rng)
{
    const zr_frame_constants& g = *c.g;
    const RhoView& rho = c.sc->rho;
    Reservoir r = InitReservoir();
    SkyIncidentRadiance leFunc; leFunc.lut = c.sc->sky;
    const V3 sunDir = v3p(g.sun_dir);
    {
        V3 sun_f = v3(0.0f), wi_s = v3(0.0f);
        {
            const V3 toSun = -sunDir;
            const float ndotSunDir = dot(toSun, normal);
            if (!(ndotSunDir < 0 && !surface.Transmissive()))
            {
                float pdf_light;
                V3 sl = UniformSampleCone(rng.Uniform2D(), g.sun_cos_angular_radius, &pdf_light);
                ONB onb = BuildONB(toSun);
                V3 wi_light = mad(sl.x, onb.b1, mad(sl.y, onb.b2, sl.z * toSun));
                surface.SetWi(wi_light, normal);
                sun_f = Unified(rho, surface).f;
                wi_s = wi_light;
            }
        }
        V3 target = v3(0.0f);
        const bool trace = (wi_s.y > 0) && ((dot(wi_s, normal) > 0) || surface.Transmissive()) && (dot(wi_s, -sunDir) >= g.sun_cos_angular_radius);
        if (trace && (dot(sun_f, sun_f) > 0))
        {
            if (VisibilityRay(c, pos, wi_s, normal, surface.Transmissive())) target = Le_Sun(pos, g) * sun_f;
        }
        const float targetLum = Luminance(target);
        float ndotwi = zr_saturate(dot(wi_s, normal));
        const float pdf_e = ndotwi * ZR_ONE_OVER_PI;
        const float pdf_s = BSDFSamplerPdf(rho, normal, surface, wi_s, leFunc, rng);
        const float w_s = BalanceHeuristic3(1, pdf_e, pdf_s, targetLum);
        Update(r, w_s, wi_s, LT_SUN, LOBE_ALL, false, 1, target, rng);
    }
    if (!rpt::IsSpecular(surface))
    {
        const V2 u = rng.Uniform2D();
        float pdf_e;
        V3 wi_e = SampleDiffuse(normal, u, &pdf_e);
        const V3 le = Le_Sky(wi_e, c.sc->sky);
        surface.SetWi(wi_e, normal);
        V3 target = le * Unified(rho, surface).f;
        if (dot(target, target) > 0) target = target * (VisibilityRay(c, pos, wi_e, normal, surface.Transmissive()) ? 1.0f : 0.0f);
        const float targetLum = Luminance(target);
        const float pdf_b = BSDFSamplerPdf(rho, normal, surface, wi_e, leFunc, rng);
        const float denom = pdf_e + pdf_b;
        const float w_e = denom == 0 ? 0.0f : targetLum / denom;
        Update(r, w_e, wi_e, LT_SKY, LOBE_ALL, false, 1, target, rng);
    }
    {
        BsdfSample bs = SampleBSDF(rho, normal, surface, leFunc, rng);
        V3 target = bs.f;
        if (dot(target, target) > 0) target = target * (VisibilityRay(c, pos, bs.wi, normal, surface.Transmissive()) ? 1.0f : 0.0f);
        const float targetLum = Luminance(target);
        float ndotwi = zr_saturate(dot(bs.wi, normal));
        const float pdf_e = ndotwi * ZR_ONE_OVER_PI;
        const float denom = bs.pdf + pdf_e;
        const float w_b = denom == 0 ? 0.0f : targetLum / denom;
        const bool useHalfVecShift = LobeAlpha(surface, bs.lobe) <= alpha_min;
        UpdateWi(r, w_b, bs.wi, surface.wo, normal, LT_SKY, bs.lobe, useHalfVecShift, target, rng);
    }
    float targetLum = Luminance(r.target);
    r.W = targetLum > 0.0f ? r.w_sum / targetLum : 0.0f;
    return r;
}

// Resampling.hlsli:11-145
struct TemporalCandidate { Surface surface; V3 pos, normal; int px, py; bool valid; };
ZR_HD TemporalCandidate FindTemporalCandidate(const SkyFrame& F, const zr_frame_constants& g, V3 pos, V3 normal, float z_view, float roughness,
    const Surface& surface, V2 prevUV)
{
    TemporalCandidate c; c.valid = false; c.px = 0; c.py = 0;
    if (prevUV.x < 0 || prevUV.y < 0 || prevUV.x > 1 || prevUV.y > 1) return c;
    const V2 renderDim = v2((float)g.render_width, (float)g.render_height);
    const int ppx = (int)(prevUV.x * renderDim.x), ppy = (int)(prevUV.y * renderDim.y);
    if (ppx >= (int)g.render_width || ppy >= (int)g.render_height || !rpt::InPlanes(F.gbPrev, ppx, ppy)) return c;
    const size_t pp = Pix(F.gbPrev, (uint32_t)ppx, (uint32_t)ppy);
    const uint16_t pmr = F.gbPrev.mr[pp];
    GFlags pf = DecodeFlags(pmr);
    if (pf.invalid || pf.emissive || (zr_abs(RoughnessOf(pmr) - roughness) > 0.3f) || (pf.metallic != surface.metallic) ||
        (pf.transmissive != surface.specTr)) return c;
    const Camera pcam = PrevCamera(g);
    PixelSurface ps = LoadPixelSurface(F.gbPrev, pcam, (uint32_t)ppx, (uint32_t)ppy, g.frame_num - 1, pp);
    float planeDist = dot(normal, ps.pos - pos);
    if (!(zr_abs(planeDist) <= kMaxPlaneDist * z_view)) return c;
    c.surface = ps.surface; c.pos = ps.pos; c.normal = ps.normal; c.px = ppx; c.py = ppy; c.valid = true;
    return c;
}

// Resampling.hlsli:147-249
ZR_HD void TemporalResample(const Ctx& c, TemporalCandidate candidate, V3 pos, V3 normal, Surface surface, Reservoir r_prev, float alpha_min,
    Reservoir& r, Rng& rng)
{
    const zr_frame_constants& g = *c.g;
    const RhoView& rho = c.sc->rho;
    r_prev.M = (r.lightType == LT_SUN) && g.sun_moved ? 0u : r_prev.M;
    const uint32_t newM = (r.M + r_prev.M) & 0xffffu;
    if (r.w_sum != 0)
    {
        float targetLum_prev = 0;
        V3 wi_offset = r.wx;
        float jacobian = 1;
        if (IsShiftInvertible(r, candidate.surface, alpha_min))
        {
            if (r.halfVectorCopyShift)
            {
                V3 wh_t = FromTangentFrameToWorld(candidate.normal, r.wx);
                wi_offset = reflect(-candidate.surface.wo, wh_t);
                jacobian = r.partialJacobian == 0 ? 0 : zr_abs(dot(candidate.surface.wo, wh_t)) / r.partialJacobian;
            }
            candidate.surface.SetWi(wi_offset, candidate.normal);
            const V3 le = LightLe(c, r.lightType, wi_offset, candidate.pos);
            const V3 target_prev = le * Unified(rho, candidate.surface).f;
            targetLum_prev = Luminance(target_prev);
            if (targetLum_prev > 0)
                { Ctx cp = c; if (c.scPrev) cp.sc = c.scPrev;      // g_bvh_prev (Sky/Resampling.hlsli:142-183)
                  targetLum_prev *= VisibilityRay(cp, candidate.pos, wi_offset, candidate.normal, candidate.surface.Transmissive()) ? 1.0f : 0.0f; }
        }
        const float numerator = (float)r.M * Luminance(r.target);
        const float denom = numerator + (float)r_prev.M * targetLum_prev * jacobian;
        const float m_curr = denom > 0 ? numerator / denom : 0;
        r.w_sum *= m_curr;
    }
    if (IsValid(r_prev) && r_prev.M > 0)
    {
        V3 wi_offset = r_prev.wx;
        float jacobian = 1;
        V3 target_curr = v3(0.0f);
        if (IsShiftInvertible(r_prev, surface, alpha_min))
        {
            if (r_prev.halfVectorCopyShift)
            {
                V3 wh_c = FromTangentFrameToWorld(normal, r_prev.wx);
                V3 wh_t = FromTangentFrameToWorld(candidate.normal, r_prev.wx);
                wi_offset = reflect(-surface.wo, wh_c);
                float whdotwo_t = zr_abs(dot(candidate.surface.wo, wh_t));
                jacobian = whdotwo_t > 0 ? zr_abs(dot(surface.wo, wh_c)) / whdotwo_t : 1;
            }
            surface.SetWi(wi_offset, normal);
            const V3 le = LightLe(c, r_prev.lightType, wi_offset, pos);
            target_curr = le * Unified(rho, surface).f;
        }
        if (dot(target_curr, target_curr) > 0)
        {
            if (VisibilityRay(c, pos, wi_offset, normal, surface.Transmissive()))
            {
                const float targetLum_curr = Luminance(target_curr);
                const float targetLum_prev = r_prev.W > 0 ? r_prev.w_sum / r_prev.W : 0;
                const float numerator = (float)r_prev.M * targetLum_prev;
                const float denom = numerator / jacobian + (float)r.M * targetLum_curr;
                const float m_prev = denom > 0 ? numerator / denom : 0;
                const float w_prev = m_prev * targetLum_curr * r_prev.W;
                Update(r, w_prev, r_prev.wx, r_prev.lightType, r_prev.lobe, r_prev.halfVectorCopyShift, surface.whdotwo, target_curr, rng);
            }
        }
    }
    float targetLum = Luminance(r.target);
    r.W = targetLum > 0.0f ? r.w_sum / targetLum : 0.0f;
    r.M = newM;
}

ZR_HD V3 Le_SkyWithSunDisk(const SceneView& sc, const zr_frame_constants& g, uint32_t x, uint32_t y) { return zr::Le_SkyWithSunDisk(sc.sky, g, x, y); }      // zr_sky.h

// K7: SkyDI_Temporal.hlsl main (:169-303) + InitialCandidatesAndTemporalReuse (:130-167) for one pixel
ZR_HD void TemporalPixel(const SkyFrame& F, const zr_frame_constants& g, uint32_t x, uint32_t y, TravStack stack, uint32_t* cnt)
{
    const SkyParams& prm = F.prm;
    const size_t px = Pix(F.gb, x, y);
    GFlags flags = DecodeFlags(F.gb.mr[px]);
    float* o = F.finalRGBA + 4 * px;
    if (flags.invalid)
    {
        if (prm.accumulate)
        {
            const V3 le = Le_SkyWithSunDisk(F.sc, g, x, y);
            const float k = g.num_frames_camera_static > 1 ? 1.0f : 0.0f;
            o[0] = o[0] * k + le.x; o[1] = o[1] * k + le.y; o[2] = o[2] * k + le.z;
        }
        else { o[0] = 0; o[1] = 0; o[2] = 0; }
        return;
    }
    if (flags.emissive)
    {
        V3 le = rdi::EmissiveColor(F.gb, px);
        if (prm.accumulate) { o[0] += le.x; o[1] += le.y; o[2] += le.z; }
        else { o[0] = le.x; o[1] = le.y; o[2] = le.z; }
        return;
    }
    const Camera cam = CurrCamera(g);
    PixelSurface ps = LoadPixelSurface(F.gb, cam, x, y, g.frame_num, px);
    if (rdi::kPrepDi) PrepareWo(F.sc.rho, ps.surface, rdi::kPrepDi);
    uint32_t hx = y, hy = x, hz = x; zr_pcg3d(&hx, &hy, &hz);                 // RNG::PCG3d(DTid.yxx).yz
    Rng rng = Rng::Init(hy, hz, g.frame_num);
    Ctx c; c.sc = &F.sc; c.scPrev = &F.scPrev; c.g = &g; c.stack = stack; c.cnt = cnt;
    Reservoir r = RIS_InitialCandidates(c, prm.alpha_min, ps.pos, ps.normal, ps.surface, rng);
    if (prm.doTemporal)
    {
        V2 motionVec = DecodeMotion(F.gb.motion[px]);
        const V2 currUV = v2(((float)x + 0.5f) / (float)g.render_width, ((float)y + 0.5f) / (float)g.render_height);
        V2 prevUV = currUV - motionVec;
        TemporalCandidate tc = FindTemporalCandidate(F, g, ps.pos, ps.normal, ps.z, ps.roughness, ps.surface, prevUV);
        if (tc.valid)
        {
            Reservoir r_prev = LoadReservoir(F.prev, Pix(F.gbPrev, (uint32_t)tc.px, (uint32_t)tc.py));
            TemporalResample(c, tc, ps.pos, ps.normal, ps.surface, r_prev, prm.alpha_min, r, rng);
        }
        if (prm.doSpatial)
        {
            r.target = rpt::Sanitize3(r.target);
            // the reference's TARGET texture is R16G16B16A16_FLOAT (SkyDI.h:61): the spatial pass reads fp16-rounded values
            F.target[px] = f4(zr_round_f16(r.target.x), zr_round_f16(r.target.y), zr_round_f16(r.target.z), 0.0f);
        }
    }
    if (prm.writeReservoirs) WriteReservoir(r, F.cur, px, r.lightType == LT_SKY ? prm.M_max_sky : prm.M_max_sun);
    if (!prm.doSpatial) rdi::WriteFinal(g, F.finalRGBA, px, r.target * r.W);
}

// PairwiseMIS.hlsli:11-157
struct PairwiseMIS { Reservoir r_s; float m_c; uint32_t M_s, k; };
ZR_HD float Compute_m_i(const PairwiseMIS& p, const Reservoir& r_c, float targetLum, const Reservoir& r_i, float jacobian)
{
    const float p_i_y_i = r_i.W > 0 ? r_i.w_sum / r_i.W : 0;
    float numerator = (float)r_i.M * p_i_y_i;
    float denom = (numerator / jacobian) + ((float)r_c.M / (float)p.k) * targetLum;
    return denom > 0 ? numerator / denom : 0;
}
ZR_HD void Update_m_c(PairwiseMIS& p, const Reservoir& r_c, const Reservoir& r_i, float targetLum, float jacobian)
{
    const float p_c_y_c = Luminance(r_c.target);
    const float numerator = (float)r_i.M * targetLum * jacobian;
    const float denom = numerator + ((float)r_c.M / (float)p.k) * p_c_y_c;
    p.m_c += 1 - (numerator / denom);
}
ZR_HD void Stream(PairwiseMIS& p, const Ctx& c, const Reservoir& r_c, V3 pos_c, V3 normal_c, Surface surface_c, const Reservoir& r_i, V3 pos_i,
    V3 normal_i, Surface surface_i, float alpha_min, Rng& rng)
{
    const RhoView& rho = c.sc->rho;
    float m_i = 0;
    V3 target_c_y_i = v3(0.0f);
    if (IsValid(r_i))
    {
        V3 wi_offset = r_i.wx;
        float jacobian = 1;
        if (IsShiftInvertible(r_i, surface_c, alpha_min))
        {
            if (r_i.halfVectorCopyShift)
            {
                V3 wh_c = FromTangentFrameToWorld(normal_c, r_i.wx);
                V3 wh_i = FromTangentFrameToWorld(normal_i, r_i.wx);
                wi_offset = reflect(-surface_c.wo, wh_c);
                float whdotwo_i = zr_abs(dot(surface_i.wo, wh_i));
                jacobian = whdotwo_i > 0 ? zr_abs(dot(surface_c.wo, wh_c)) / whdotwo_i : 1;
            }
            surface_c.SetWi(wi_offset, normal_c);
            const V3 le = LightLe(c, r_i.lightType, wi_offset, pos_c);
            target_c_y_i = le * Unified(rho, surface_c).f;
            if (dot(target_c_y_i, target_c_y_i) > 0)
                target_c_y_i = target_c_y_i * (VisibilityRay(c, pos_c, wi_offset, normal_c, surface_c.Transmissive()) ? 1.0f : 0.0f);
        }
        m_i = Compute_m_i(p, r_c, Luminance(target_c_y_i), r_i, jacobian);
    }
    V3 target_i_y_c = v3(0.0f);
    float jacobian = 1;
    if (IsValid(r_c))
    {
        V3 wi_offset = r_c.wx;
        if (IsShiftInvertible(r_c, surface_i, alpha_min))
        {
            if (r_c.halfVectorCopyShift)
            {
                V3 wh_i = FromTangentFrameToWorld(normal_i, r_c.wx);
                wi_offset = reflect(-surface_i.wo, wh_i);
                float whdotwo_i = zr_abs(dot(surface_i.wo, wh_i));
                jacobian = whdotwo_i > 0 ? zr_abs(dot(surface_i.wo, wh_i)) / r_c.partialJacobian : 1;
            }
            surface_i.SetWi(wi_offset, normal_i);
            const V3 le = LightLe(c, r_c.lightType, wi_offset, pos_i);
            target_i_y_c = le * Unified(rho, surface_i).f;
            if (dot(target_i_y_c, target_i_y_c) > 0)
                target_i_y_c = target_i_y_c * (VisibilityRay(c, pos_i, wi_offset, normal_i, surface_i.Transmissive()) ? 1.0f : 0.0f);
        }
    }
    Update_m_c(p, r_c, r_i, Luminance(target_i_y_c), jacobian);
    if (IsValid(r_i))
    {
        const float w_i = m_i * Luminance(target_c_y_i) * r_i.W;
        Update(p.r_s, w_i, r_i.wx, r_i.lightType, r_i.lobe, r_i.halfVectorCopyShift, surface_c.whdotwo, target_c_y_i, rng);
    }
    p.M_s = (p.M_s + r_i.M) & 0xffffu;
}
ZR_HD void End(PairwiseMIS& p, const Reservoir& r_c, Rng& rng)
{
    const float w_c = p.m_c * r_c.w_sum;
    Update(p.r_s, w_c, r_c.wx, r_c.lightType, r_c.lobe, r_c.halfVectorCopyShift, r_c.partialJacobian, r_c.target, rng);
    p.r_s.M = p.M_s;
    const float targetLum = Luminance(p.r_s.target);
    p.r_s.W = targetLum > 0 ? p.r_s.w_sum / (targetLum * (1 + (float)p.k)) : 0;
}

// Resampling.hlsli:259-277 (`static const half2`: values round to fp16 on load)
ZR_HD V2 SpatialSample(uint32_t i)
{
    const float k[16][2] = {
        {-0.899423f, 0.365076f}, {-0.744442f, -0.124006f}, {-0.229714f, 0.245876f}, {-0.545186f, 0.741148f}, {-0.156274f, -0.336366f},
        {0.468400f, 0.348798f}, {0.035776f, 0.606928f}, {-0.208966f, 0.904852f}, {-0.491070f, -0.484810f}, {0.162490f, -0.081156f},
        {0.232062f, -0.851382f}, {0.641310f, -0.162124f}, {0.320798f, 0.922460f}, {0.959086f, 0.263642f}, {0.531136f, -0.519002f},
        {-0.223014f, -0.774740f}};
    return v2(zr_round_f16(k[i & 15][0]), zr_round_f16(k[i & 15][1]));
}

// K8: SkyDI_Spatial.hlsl main + SpatialResample (Resampling.hlsli:251-409) for one pixel
ZR_HD void SpatialPixel(const SkyFrame& F, const zr_frame_constants& g, uint32_t x, uint32_t y, TravStack stack, uint32_t* cnt)
{
    const SkyParams& prm = F.prm;
    const size_t px = Pix(F.gb, x, y);
    GFlags flags = DecodeFlags(F.gb.mr[px]);
    if (flags.invalid || flags.emissive) return;
    const uint32_t W = g.render_width, H = g.render_height;
    const Camera cam = CurrCamera(g);
    PixelSurface ps = LoadPixelSurface(F.gb, cam, x, y, g.frame_num, px);
    if (rdi::kPrepDi) PrepareWo(F.sc.rho, ps.surface, rdi::kPrepDi);
    uint32_t hx = y, hy = x, hz = x; zr_pcg3d(&hx, &hy, &hz);
    Rng rng = Rng::Init(hy, hz, g.frame_num);
    Ctx c; c.sc = &F.sc; c.scPrev = &F.scPrev; c.g = &g; c.stack = stack; c.cnt = cnt;
    Reservoir r_c = LoadReservoir(F.cur, px);
    r_c.target = xyz(F.target[px]);
    const float u0 = rng.Uniform();
    const int offset = (int)rng.UniformUintBounded_Faster(16);
    const float theta = u0 * ZR_TWO_PI;
    const float sinTheta = zr_sin(theta), cosTheta = zr_cos(theta);
    PairwiseMIS pw; pw.r_s = InitReservoir(); pw.m_c = 1.0f; pw.M_s = r_c.M; pw.k = kNumSpatial;
    uint32_t candX[kNumSpatial], candY[kNumSpatial];
    uint32_t k = 0;
    for (int i = 0; i < kNumSpatial; i++)
    {
        const V2 uv = SpatialSample((uint32_t)(offset + i));
        float rx = uv.x * cosTheta + uv.y * -sinTheta, ry = uv.x * sinTheta + uv.y * cosTheta;
        rx *= kSearchRadius; ry *= kSearchRadius;
        const int sx = zr_f2i_sat(__builtin_rintf((float)x + rx)), sy = zr_f2i_sat(__builtin_rintf((float)y + ry));
        if (!(sx >= 0 && sy >= 0 && sx < (int)W && sy < (int)H)) continue;
        if (!rpt::InPlanes(F.gb, sx, sy)) continue;
        const size_t sp = Pix(F.gb, (uint32_t)sx, (uint32_t)sy);
        GFlags fi = DecodeFlags(F.gb.mr[sp]);
        if (fi.invalid || fi.emissive) continue;
        PixelSurface pi = LoadPixelSurfaceEx(F.gb, cam, (uint32_t)sx, (uint32_t)sy, g.frame_num, sp, false);
        bool valid = zr_abs(dot(ps.normal, pi.pos - ps.pos)) <= kMaxPlaneDist * ps.z;
        valid = valid && (zr_abs(pi.roughness - ps.roughness) < kMaxRoughDiff);
        if (!valid) continue;
        candX[k] = (uint32_t)sx; candY[k] = (uint32_t)sy; k++;
    }
    pw.k = k;
    for (uint32_t i = 0; i < k; i++)
    {
        const size_t sp = Pix(F.gb, candX[i], candY[i]);
        PixelSurface pi = LoadPixelSurfaceEx(F.gb, cam, candX[i], candY[i], g.frame_num, sp, false);
        Reservoir r_spatial = LoadReservoir(F.cur, sp);
        Stream(pw, c, r_c, ps.pos, ps.normal, ps.surface, r_spatial, pi.pos, pi.normal, pi.surface, prm.alpha_min, rng);
    }
    End(pw, r_c, rng);
    rdi::WriteFinal(g, F.finalRGBA, px, pw.r_s.target * pw.r_s.W);
}

} // namespace sdi
} // namespace zr
