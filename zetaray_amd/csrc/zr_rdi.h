// zr_rdi.h -- per-pixel stage functions of ReSTIR DI for emissive lights (K5 temporal, K6 spatial).  The reference's compile-time switch
// USE_HALF_VECTOR_COPY_SHIFT (Params.hlsli:12, 0 in its tree) is DiParams::halfVec here: a template constant of the kernels (zr_kernels_di.h HVS), selected by
// ZR_DI_HALF_VECTOR_COPY_SHIFT; the permutation without it folds every line of the shift away.
//
// Reference (Source/ZetaRenderPass/DirectLighting/Emissive/): ReSTIR_DI_Temporal.hlsl:29-390, ReSTIR_DI_Spatial.hlsl:24-192,
// Resampling.hlsli:10-521, PairwiseMIS.hlsli:11-231, Reservoir.hlsli:11-226, Util.hlsli:11-119, Params.hlsli;
// host order DirectLighting.cpp:166-296.  Persistent state in the reference's formats: two reservoir sets x (A RGBA32_UINT:
// bary unorm2 | le half2 | le.z half + M << 16 | lightIdx, B RG32F: w_sum, W) = 24 B/px, target RGBA32F.
// Pinned: Le_SkyWithSunDisk for miss pixels = 0 while the scene has no sky-view LUT (no ZR_PASS_SKY rendered yet); ftou of a negative neighbour position
// = 0 (D3D rule); the spatial pass's WaveActiveSum(disoccluded) runs over the 8x8 pixel group = one wave64.
#pragma once
#include "zr_rpt.h"

namespace zr {
namespace rdi {

using rpt::Pix; using rpt::GFlags; using rpt::DecodeFlags; using rpt::RoughnessOf; using rpt::DecodeMotion; using rpt::Camera;
using rpt::CurrCamera; using rpt::PrevCamera; using rpt::PixelSurface; using rpt::LoadPixelSurface; using rpt::LoadPixelSurfaceEx;
// wo-only term groups (zr_dev_bsdf.h WO_*) the direct-lighting kernels prepare on the pixel's surface: the rho-LUT read (group 1), as in the shifts
#ifndef ZR_PREP_DI
#define ZR_PREP_DI 1
#endif
static constexpr uint32_t kPrepDi = ZR_PREP_DI;
using rpt::Globals; using rpt::VisibilitySegmentApprox; using rpt::IsSpecular; using rpt::IsLobeValid; using rpt::LobeAlpha;
// Math.hlsli:308-322
ZR_HD V3 WorldToTangentFrame(V3 normal, V3 w) { ONB o = BuildONB(normal); return v3(dot(o.b1, w), dot(o.b2, w), dot(normal, w)); }
ZR_HD V3 FromTangentFrameToWorld(V3 normal, V3 wl) { ONB o = BuildONB(normal); return wl.x * o.b1 + wl.y * o.b2 + wl.z * normal; }

static constexpr int kNumLightCandidates = 3;
static constexpr int kMinSpatial = 1, kExtraSpatial = 1, kMaxSpatial = 4;
static constexpr float kProbExtraSpatial = 0.6f, kSearchRadius = 16.0f, kMaxPlaneDist = 1e-1f, kMaxRoughDiff = 0.15f;

struct DiPlanes { U4* A; float* B; };

// Reservoir.hlsli:11-213
struct Reservoir
{
    float w_sum, W; V3 le; uint32_t lightIdx; V2 bary; uint32_t M;
    V3 target; uint32_t lightID; V3 lightPos, lightNormal; bool doubleSided;
    // half-vector copy shift (Reservoir.hlsli:56-119, 203-212): the selected sample came from a lobe narrower than alpha_min; it is reused by copying its half vector
    // in the shading frame (wh_local) and re-tracing the reflected ray; partialJacobian = |wh . wo| where it was drawn; lobe = LOBE_* (LOBE_ALL for light samples)
    bool halfVectorCopyShift; uint32_t lobe; V3 wh_local; float partialJacobian;
    ZR_HDM bool Update(float weight, V3 le_, uint32_t lightIdx_, V2 bary_, Rng& rng)
    {
        if (zr_isnan(weight)) return false;
        M += 1;
        if (weight == 0) return false;
        w_sum += weight;
        if (rng.Uniform() < (weight / w_sum)) { le = le_; lightIdx = lightIdx_; bary = bary_; halfVectorCopyShift = false; lobe = LOBE_ALL; return true; }
        return false;
    }
    // a BSDF-sampled candidate (Reservoir.hlsli:56-91)
    ZR_HDM bool Update(float weight, bool halfVecShift, V3 wi, V3 wo, V3 normal, uint32_t lb, V3 le_, uint32_t lightIdx_, V2 bary_, Rng& rng)
    {
        if (zr_isnan(weight)) return false;
        M += 1;
        if (weight == 0) return false;
        w_sum += weight;
        if (rng.Uniform() < (weight / w_sum))
        {
            le = le_; lightIdx = lightIdx_; bary = bary_; halfVectorCopyShift = halfVecShift; lobe = lb;
            if (halfVecShift)
            {
                V3 wh = normalize(wo + wi);
                wh_local = WorldToTangentFrame(normal, wh);
                partialJacobian = zr_abs(dot(wh, wo));
            }
            return true;
        }
        return false;
    }
    // a reused sample (Reservoir.hlsli:93-119)
    ZR_HDM bool Update(float weight, bool halfVecShift, V3 wh, float whdotwo, uint32_t lb, V3 le_, uint32_t lightIdx_, V2 bary_, Rng& rng)
    {
        if (zr_isnan(weight)) return false;
        M += 1;
        if (weight == 0) return false;
        w_sum += weight;
        if (rng.Uniform() < (weight / w_sum))
        { le = le_; lightIdx = lightIdx_; bary = bary_; halfVectorCopyShift = halfVecShift; lobe = lb; wh_local = wh; partialJacobian = whdotwo; return true; }
        return false;
    }
    ZR_HDM void Write(const DiPlanes& p, size_t i, uint32_t M_max, bool halfVec = false) const
    {
        uint32_t lx = zr_f32_to_f16(le.x), ly = zr_f32_to_f16(le.y), lz = zr_f32_to_f16(le.z);
        uint32_t M_capped = (M & 0xffffu) < M_max ? (M & 0xffffu) : M_max;
        uint32_t bx = FloatToUNorm16(bary.x), by = FloatToUNorm16(bary.y);
        if (halfVec)
        {   // Reservoir.hlsli:177-184: metadata bit 5 = the flag, bits 6..8 = the lobe; the oct-encoded half vector takes the barycentrics' place
            M_capped |= ((halfVectorCopyShift ? 1u : 0u) << 5) | ((lobe > LOBE_ALL ? (uint32_t)LOBE_ALL : lobe) << 6);
            if (halfVectorCopyShift) { const V2 e = EncodeUnitVector(wh_local); bx = FloatToUNorm16(e.x); by = FloatToUNorm16(e.y); }
        }
        U4 a; a.x = (by << 16) | bx; a.y = (ly << 16) | lx; a.z = (M_capped << 16) | lz; a.w = lightIdx;
        p.A[i] = a;
        p.B[2 * i] = w_sum; p.B[2 * i + 1] = W;
    }
};
ZR_HD Reservoir InitReservoir()
{
    Reservoir r; r.le = v3(0.0f); r.M = 0; r.w_sum = 0; r.W = 0; r.lightIdx = 0xffffffffu; r.bary = v2(0, 0);
    r.target = v3(0.0f); r.lightID = 0xffffffffu; r.lightPos = v3(0.0f); r.lightNormal = v3(0.0f); r.doubleSided = false;
    r.halfVectorCopyShift = false; r.lobe = LOBE_ALL; r.wh_local = v3(0.0f); r.partialJacobian = 1;      // (wh_local: not initialised by the reference's Init(); only read behind the flag)
    return r;
}
ZR_HD Reservoir LoadReservoir(const DiPlanes& p, size_t i, bool halfVec = false)
{
    const U4 a = p.A[i];
    Reservoir r = InitReservoir();
    r.M = (a.z >> 16) & 0x1f;
    r.w_sum = p.B[2 * i]; r.W = p.B[2 * i + 1];
    r.le = v3(zr_f16_to_f32((uint16_t)(a.y & 0xffff)), zr_f16_to_f32((uint16_t)(a.y >> 16)), zr_f16_to_f32((uint16_t)(a.z & 0xffff)));
    r.lightIdx = a.w;
    r.bary = v2(zr_div65535((float)(a.x & 0xffff)), zr_div65535((float)(a.x >> 16)));
    if (halfVec)
    {   // Reservoir.hlsli:156-160: A.x is read both as the barycentrics and as the oct-encoded half vector
        const uint32_t metadata = a.z >> 16;
        r.halfVectorCopyShift = ((metadata >> 5) & 0x1u) != 0;
        const uint32_t lv = (metadata >> 6) & 0x7u;
        r.lobe = lv > (uint32_t)LOBE_ALL ? (uint32_t)LOBE_ALL : lv;
        r.wh_local = DecodeOct32u(a.x);
    }
    return r;
}

// Util.hlsli:11-57
struct EmissiveData { V3 wi; float t; uint32_t ID; V3 lightPos, lightNormal; bool doubleSided; };
ZR_HD EmissiveData InitEmissiveData(const SceneView& sc, uint32_t lightIdx, V2 bary)
{
    EmissiveData ret;
    const zr_emissive_triangle tri = sc.emissives[lightIdx];
    ret.ID = tri.id;
    const V3 vtx0 = v3p(tri.vtx0), vtx1 = EmV1(tri), vtx2 = EmV2(tri);
    ret.lightPos = (1.0f - bary.x - bary.y) * vtx0 + bary.x * vtx1 + bary.y * vtx2;
    ret.lightNormal = cross(vtx1 - vtx0, vtx2 - vtx0);
    ret.lightNormal = dot(ret.lightNormal, ret.lightNormal) == 0 ? ret.lightNormal : normalize(ret.lightNormal);
    ret.doubleSided = EmDoubleSided(tri);
    ret.wi = v3(0.0f); ret.t = 0;
    return ret;
}
ZR_HD void SetSurfacePos(EmissiveData& e, V3 pos)
{
    e.wi = e.lightPos - pos;
    e.t = dot(e.wi, e.wi) == 0 ? 0 : length(e.wi);
    e.wi = e.t == 0 ? v3(0.0f) : e.wi / e.t;
    e.lightNormal = e.doubleSided && dot(-e.wi, e.lightNormal) < 0 ? -e.lightNormal : e.lightNormal;
}
ZR_HD float dWdA(const EmissiveData& e)
{
    float cosThetaPrime = zr_saturate(dot(e.lightNormal, -e.wi));
    return e.t == 0 ? 0 : cosThetaPrime / (e.t * e.t);
}

// Util.hlsli:59-119
struct BSDFHitInfo { uint32_t emissiveTriIdx; V2 bary; V3 lightPos; float t; bool hit; };
ZR_HD BSDFHitInfo FindClosestHit(const Globals& g, V3 pos, V3 normal, V3 wi, bool transmissive)
{
    BSDFHitInfo ret; ret.hit = false; ret.emissiveTriIdx = 0xffffffffu; ret.bary = v2(0, 0); ret.lightPos = v3(0.0f); ret.t = 0;
    float ndotwi = dot(normal, wi);
    if (ndotwi == 0) return ret;
    bool wiBackface = ndotwi < 0;
    if (wiBackface)
    {
        if (transmissive) normal = normal * -1.0f;
        else return ret;
    }
    const V3 o = OffsetRayRTG(pos, normal);
    g.cnt[0]++;
    RawHit h = Traverse<false>(*g.sc, o, wi, wiBackface ? 3e-4f : 0.0f, ZR_FLT_MAX, ZR_SUBGROUP_ALL, g.stack);
    if (h.tri == kInvalidTri) return ret;
    const TriMeta tm = g.sc->triMeta[h.tri];
    const uint32_t base = g.sc->instances[tm.mesh].base_emissive_tri_offset;
    if (base == 0xffffffffu) return ret;
    ret.emissiveTriIdx = base + tm.prim;
    ret.bary = v2(h.u, h.v);
    ret.lightPos = mad(h.t, wi, o);
    ret.t = h.t;
    ret.hit = true;
    return ret;
}

// SampleBSDF_NoDiffuse(normal, surface, rng), BSDFSampling.hlsli:154-165
ZR_HD BsdfSample SampleBSDF_NoDiffuseRng(const RhoView& rho, V3 n, const Surface& s, Rng& rng)
{
    V2 u_c = rng.Uniform2D();
    V2 u_g = rng.Uniform2D();
    float u0 = rng.Uniform(), u1 = rng.Uniform();
    return SampleBSDF_NoDiffuse(rho, n, s, u_c, u_g, u0, u1);
}

struct DiParams { uint32_t flags, M_max, numSampleSets, accumulate, doTemporal, doSpatial, writeReservoirs, halfVec; float alpha_min; };

struct DiFrame
{
    SceneView sc; GBuf gb, gbPrev; DiPlanes cur, prev; F4* target; float* finalRGBA; const uint16_t* sampleSet; DiParams prm;
    SceneView scPrev;        // previous frame's acceleration structure (g_bvh_prev, ReSTIR_DI_Temporal.hlsl:17)
    uint32_t ox0, oy0, ow, oh;
    ZR_HDM bool Owns(uint32_t x, uint32_t y) const { return x >= ox0 && y >= oy0 && x < ox0 + ow && y < oy0 + oh; }
};

// Reservoir.hlsli:216-225
ZR_HD bool IsShiftInvertible(const DiParams& prm, const Reservoir& r_base, const Surface& surface_offset)
{
    if (!prm.halfVec) return true;
    return !r_base.halfVectorCopyShift || (IsLobeValid(surface_offset, r_base.lobe) && (LobeAlpha(surface_offset, r_base.lobe) <= prm.alpha_min));
}
// the half-vector copy shift's offset path (Resampling.hlsli:146-179, 220-252; PairwiseMIS.hlsli:72-102, 140-168): the copied half vector's reflection is traced from the
// offset surface; target = Le * dwdA of the light it lands on (0 when it lands on the light's back).  false: nothing emissive was hit.  `surface` gets wi = the traced
// direction whenever a light was hit.
ZR_HD bool HalfVectorOffsetTarget(const Globals& gl, V3 pos, V3 normal, Surface& surface, V3 wh, V3& target)
{
    const SceneView& sc = *gl.sc;
    const V3 wi_offset = reflect(-surface.wo, wh);
    BSDFHitInfo hitInfo = FindClosestHit(gl, pos, normal, wi_offset, surface.Transmissive());
    if (!hitInfo.hit) return false;
    const zr_emissive_triangle em = sc.emissives[hitInfo.emissiveTriIdx];
    const V3 le = EmLe(sc, em, hitInfo.bary);
    const V3 vtx0 = v3p(em.vtx0), vtx1 = EmV1(em), vtx2 = EmV2(em);
    V3 lightNormal = cross(vtx1 - vtx0, vtx2 - vtx0);
    float twoArea = length(lightNormal);
    lightNormal = dot(lightNormal, lightNormal) == 0 ? v3(0.0f) : lightNormal / twoArea;
    lightNormal = EmDoubleSided(em) && dot(-wi_offset, lightNormal) < 0 ? -lightNormal : lightNormal;
    if (dot(-wi_offset, lightNormal) > 0)
    {
        float dwdA = zr_saturate(dot(lightNormal, -wi_offset)) / (hitInfo.t * hitInfo.t);
        target = le * dwdA;
    }
    surface.SetWi(wi_offset, normal);
    return true;
}

// ReSTIR_DI_Temporal.hlsl:29-203
ZR_HD Reservoir RIS_InitialCandidates(const Globals& gl, const zr_frame_constants& g, const DiParams& prm, V3 pos, V3 normal, Surface surface,
    uint32_t sampleSetIdx, int numBsdfSamples, Rng& rng)
{
    const SceneView& sc = *gl.sc;
    Reservoir r = InitReservoir();
    const int numLightSamples = !IsSpecular(surface) ? kNumLightCandidates : 0;
    for (int s_b = 0; s_b < numBsdfSamples; s_b++)
    {
        BsdfSample bs = SampleBSDF_NoDiffuseRng(sc.rho, normal, surface, rng);
        V3 wi = bs.wi;
        float pdf_w = bs.pdf;
        const bool useHalfVecShift = prm.halfVec ? (LobeAlpha(surface, bs.lobe) <= prm.alpha_min) : false;      // ReSTIR_DI_Temporal.hlsl:45-50
        BSDFHitInfo hitInfo = FindClosestHit(gl, pos, normal, wi, surface.Transmissive());
        float w_b = 0; V3 le = v3(0.0f), lightNormal = v3(0.0f), target = v3(0.0f); uint32_t emissiveID = 0xffffffffu; bool doubleSided = false;
        if (hitInfo.hit)
        {
            const zr_emissive_triangle em = sc.emissives[hitInfo.emissiveTriIdx];
            le = EmLe(sc, em, hitInfo.bary);
            const V3 vtx0 = v3p(em.vtx0), vtx1 = EmV1(em), vtx2 = EmV2(em);
            lightNormal = cross(vtx1 - vtx0, vtx2 - vtx0);
            float twoArea = length(lightNormal);
            lightNormal = dot(lightNormal, lightNormal) == 0 ? v3(0.0f) : lightNormal / twoArea;
            lightNormal = EmDoubleSided(em) && dot(-wi, lightNormal) < 0 ? -lightNormal : lightNormal;
            doubleSided = EmDoubleSided(em);
            emissiveID = em.id;
            if (dot(-wi, lightNormal) > 0)
            {
                const float lightSourcePdf = sc.alias[hitInfo.emissiveTriIdx].cached_p_orig;
                const float pdf_light = lightSourcePdf * (1.0f / (0.5f * twoArea));
                const float dwdA = zr_saturate(dot(lightNormal, -wi)) / (hitInfo.t * hitInfo.t);
                pdf_w *= dwdA;
                const bool sampleIsSpecular = (surface.GlossSpecular() && bs.lobe == LOBE_GLOSSY_R) || (surface.CoatSpecular() && bs.lobe == LOBE_COAT);
                float denom = (float)numBsdfSamples * pdf_w + (float)(!sampleIsSpecular ? 1 : 0) * (float)numLightSamples * pdf_light;
                const float m_i = 1.0f / denom;
                target = le * bs.f * dwdA;
                w_b = m_i * Luminance(target);
            }
        }
        if (prm.halfVec ? r.Update(w_b, useHalfVecShift, wi, surface.wo, normal, bs.lobe, le, hitInfo.emissiveTriIdx, hitInfo.bary, rng)
                        : r.Update(w_b, le, hitInfo.emissiveTriIdx, hitInfo.bary, rng))
        { r.target = target; r.lightID = emissiveID; r.lightPos = hitInfo.lightPos; r.lightNormal = lightNormal; r.doubleSided = doubleSided; }
    }
    for (int s_l = 0; s_l < numLightSamples; s_l++)
    {
        V3 lpos, ln, le; V2 lbary; float pdf_light; uint32_t emissiveIdx, lightID; bool doubleSided;
        if (prm.numSampleSets)
        {
            uint32_t u = rng.UniformUintBounded_Faster(sc.sampleSetSize);
            const zr_presampled_tri t = sc.sampleSets[(size_t)sampleSetIdx * sc.sampleSetSize + u];
            lpos = v3p(t.pos); ln = DecodeOct32(t.normal);
            lbary = v2(zr_div65535((float)t.bary[0]), zr_div65535((float)t.bary[1]));
            le = v3(zr_f16_to_f32(t.le[0]), zr_f16_to_f32(t.le[1]), zr_f16_to_f32(t.le[2]));
            pdf_light = t.pdf; emissiveIdx = t.idx; lightID = t.id; doubleSided = t.two_sided != 0;
            if (doubleSided && dot(pos - lpos, ln) < 0) ln = -ln;
        }
        else
        {
            uint32_t u0 = rng.UniformUintBounded(g.num_emissive_triangles);
            const zr_alias_entry ae = sc.alias[u0];
            float lpdfSrc;
            if (rng.Uniform() < ae.p_curr) { lpdfSrc = ae.cached_p_orig; emissiveIdx = u0; }
            else { lpdfSrc = ae.cached_p_alias; emissiveIdx = ae.alias; }
            const zr_emissive_triangle em = sc.emissives[emissiveIdx];
            lbary = UniformSampleTriangle(rng.Uniform2D());
            const V3 vtx0 = v3p(em.vtx0), vtx1 = EmV1(em), vtx2 = EmV2(em);
            lpos = (1.0f - lbary.x - lbary.y) * vtx0 + lbary.x * vtx1 + lbary.y * vtx2;
            ln = cross(vtx1 - vtx0, vtx2 - vtx0);
            bool normalIs0 = dot(ln, ln) == 0;
            float twoArea = length(ln);
            float lpdfPos = normalIs0 ? 0.0f : 2.0f / twoArea;
            ln = normalIs0 ? ln : ln / twoArea;
            ln = EmDoubleSided(em) && dot(pos - lpos, ln) < 0 ? -ln : ln;
            le = EmLe(sc, em, lbary);
            pdf_light = lpdfSrc * lpdfPos;
            lightID = em.id; doubleSided = EmDoubleSided(em);
        }
        V3 target = v3(0.0f);
        V3 wi = lpos - pos;
        const bool isZero = dot(wi, wi) == 0;
        const float t = isZero ? 0 : length(wi);
        wi = isZero ? wi : wi / t;
        const float dwdA = isZero ? 0 : zr_saturate(dot(ln, -wi)) / (t * t);
        surface.SetWi(wi, normal);
        if (dot(ln, -wi) > 0)
        {
            target = le * Unified(sc.rho, surface).f * dwdA;
            if (dot(target, target) > 0)
                target = target * (VisibilitySegmentApprox(gl, pos, wi, t, normal, lightID, surface.Transmissive()) ? 1.0f : 0.0f);
        }
        const float denom = (float)numLightSamples * pdf_light + (float)numBsdfSamples * BSDFSamplerPdf_NoDiffuse(sc.rho, normal, surface, wi) * dwdA;
        const float m_l = denom > 0 ? 1.0f / denom : 0;
        const float w_l = m_l * Luminance(target);
        if (r.Update(w_l, le, emissiveIdx, lbary, rng))
        { r.target = target; r.lightID = lightID; r.lightNormal = ln; r.lightPos = lpos; r.doubleSided = doubleSided; }
    }
    float targetLum = Luminance(r.target);
    r.W = targetLum > 0.0f ? r.w_sum / targetLum : 0.0f;
    return r;
}

// Resampling.hlsli:10-136
struct TemporalCandidate { Surface surface; V3 pos, normal; int px, py; bool valid; };
ZR_HD TemporalCandidate FindTemporalCandidate(const DiFrame& F, const zr_frame_constants& g, V3 pos, V3 normal, float roughness, const Surface& surface, V2 prevUV)
{
    TemporalCandidate c; c.valid = false; c.px = 0; c.py = 0;
    if (prevUV.x < 0 || prevUV.y < 0 || prevUV.x > 1 || prevUV.y > 1) return c;
    const V2 renderDim = v2((float)g.render_width, (float)g.render_height);
    const int ppx = (int)(prevUV.x * renderDim.x), ppy = (int)(prevUV.y * renderDim.y);
    if (ppx >= (int)g.render_width || ppy >= (int)g.render_height || !rpt::InPlanes(F.gbPrev, ppx, ppy)) return c;
    const size_t pp = Pix(F.gbPrev, (uint32_t)ppx, (uint32_t)ppy);
    const uint16_t pmr = F.gbPrev.mr[pp];
    GFlags pf = DecodeFlags(pmr);
    if (pf.invalid || pf.emissive || (zr_abs(RoughnessOf(pmr) - roughness) > kMaxRoughDiff) || (pf.metallic != surface.metallic) ||
        (pf.transmissive != surface.specTr)) return c;
    const Camera pcam = PrevCamera(g);
    PixelSurface ps = LoadPixelSurface(F.gbPrev, pcam, (uint32_t)ppx, (uint32_t)ppy, g.frame_num - 1, pp);
    float planeDist = dot(normal, ps.pos - pos);
    if (!(zr_abs(planeDist) <= kMaxPlaneDist * ps.z)) return c;
    c.surface = ps.surface; c.pos = ps.pos; c.normal = ps.normal; c.px = ppx; c.py = ppy; c.valid = true;
    return c;
}

// Resampling.hlsli:138-285 (no half-vector shift)
ZR_HD float OffsetPathTarget_CtT(const Globals& gl, const DiParams& prm, const Reservoir& r_curr, const TemporalCandidate& candidate, V3 wh)
{
    Surface surface = candidate.surface;
    if (!IsShiftInvertible(prm, r_curr, surface)) return 0;
    V3 target = v3(0.0f);
    V3 wi = v3(0.0f);
    float t = 0;
    if (prm.halfVec && r_curr.halfVectorCopyShift)
    {
        Globals gp = gl; if (gl.scPrev) gp.sc = gl.scPrev;       // g_bvh_prev
        if (!HalfVectorOffsetTarget(gp, candidate.pos, candidate.normal, surface, wh, target)) return 0;
    }
    else
    {
        wi = r_curr.lightPos - candidate.pos;
        const bool isZero = dot(wi, wi) == 0;
        t = isZero ? 0 : length(wi);
        wi = isZero ? wi : wi / t;
        surface.SetWi(wi, candidate.normal);
        V3 ln = r_curr.lightNormal;
        if (r_curr.doubleSided && dot(-wi, ln) < 0) ln = -ln;
        float cosThetaPrime = zr_saturate(dot(ln, -wi));
        const float dwdA = isZero ? 0 : cosThetaPrime / (t * t);
        target = r_curr.le * dwdA;
    }
    target = target * Unified(gl.sc->rho, surface).f;
    float lum = Luminance(target);
    if (!(prm.halfVec && r_curr.halfVectorCopyShift) && lum > 0)
    {
        Globals gp = gl; if (gl.scPrev) gp.sc = gl.scPrev;       // g_bvh_prev (Resampling.hlsli:134-200)
        lum *= VisibilitySegmentApprox(gp, candidate.pos, wi, t, candidate.normal, r_curr.lightID, surface.Transmissive()) ? 1.0f : 0.0f;
    }
    return lum;
}
ZR_HD V3 OffsetPathTarget_TtC(const Globals& gl, const DiParams& prm, const Reservoir& r_prev, V3 pos, V3 normal, Surface surface, V3 wh)
{
    if (!IsShiftInvertible(prm, r_prev, surface)) return v3(0.0f);
    V3 target = v3(0.0f);
    V3 wi_offset = v3(0.0f);
    float t_offset = 0;
    uint32_t lightID = 0xffffffffu;
    if (prm.halfVec && r_prev.halfVectorCopyShift)
    {
        if (!HalfVectorOffsetTarget(gl, pos, normal, surface, wh, target)) return v3(0.0f);
    }
    else
    {
        EmissiveData e = InitEmissiveData(*gl.sc, r_prev.lightIdx, r_prev.bary);
        SetSurfacePos(e, pos);
        wi_offset = e.wi; t_offset = e.t; lightID = e.ID;
        float dwdA_ = dWdA(e);
        surface.SetWi(e.wi, normal);
        target = r_prev.le * dwdA_;
    }
    target = target * Unified(gl.sc->rho, surface).f;
    if (!(prm.halfVec && r_prev.halfVectorCopyShift) && dot(target, target) > 0)
        target = target * (VisibilitySegmentApprox(gl, pos, wi_offset, t_offset, normal, lightID, surface.Transmissive()) ? 1.0f : 0.0f);
    return target;
}

// Resampling.hlsli:287-339
ZR_HD void TemporalResample1(const Globals& gl, const DiFrame& F, V3 pos, V3 normal, const Surface& surface, const TemporalCandidate& candidate,
    Reservoir& r_curr, Rng& rng)
{
    const DiParams& prm = F.prm;
    Reservoir r_prev = LoadReservoir(F.prev, Pix(F.gb, (uint32_t)candidate.px, (uint32_t)candidate.py), prm.halfVec != 0);
    const uint32_t newM = (r_curr.M + r_prev.M) & 0xffffu;
    if (r_curr.w_sum != 0)
    {
        float jacobian = 1; V3 wh_prev = v3(0.0f);
        if (prm.halfVec)
        {
            wh_prev = FromTangentFrameToWorld(candidate.normal, r_curr.wh_local);
            float whdotwo = zr_abs(dot(candidate.surface.wo, wh_prev));
            jacobian = r_curr.partialJacobian == 0 ? 0 : whdotwo / r_curr.partialJacobian;
            jacobian = r_curr.halfVectorCopyShift ? jacobian : 1;
        }
        float targetLum_prev = OffsetPathTarget_CtT(gl, prm, r_curr, candidate, wh_prev);
        const float numerator = (float)r_curr.M * Luminance(r_curr.target);
        const float denom = numerator + (float)r_prev.M * targetLum_prev * jacobian;
        const float m_curr = denom > 0 ? numerator / denom : 0;
        r_curr.w_sum *= m_curr;
    }
    if (r_prev.lightIdx != 0xffffffffu)
    {
        float jacobian = 1, whdotwo_curr = 0; V3 wh_curr = v3(0.0f);
        if (prm.halfVec)
        {
            wh_curr = FromTangentFrameToWorld(normal, r_prev.wh_local);
            V3 wh_prev = FromTangentFrameToWorld(candidate.normal, r_prev.wh_local);
            float whdotwo_prev = zr_abs(dot(candidate.surface.wo, wh_prev));
            whdotwo_curr = zr_abs(dot(surface.wo, wh_curr));
            jacobian = whdotwo_prev > 0 ? whdotwo_curr / whdotwo_prev : 0;
            jacobian = r_prev.halfVectorCopyShift ? jacobian : 1;
        }
        const V3 target_curr = OffsetPathTarget_TtC(gl, prm, r_prev, pos, normal, surface, wh_curr);
        const float targetLum_curr = Luminance(target_curr);
        if (targetLum_curr > 0)
        {
            const float targetLum_prev = r_prev.W > 0 ? r_prev.w_sum / r_prev.W : 0;
            const float numerator = (float)r_prev.M * targetLum_prev;
            const float denom = numerator / jacobian + (float)r_curr.M * targetLum_curr;
            const float m_prev = denom > 0 ? numerator / denom : 0;
            const float w_prev = m_prev * targetLum_curr * r_prev.W;
            if (prm.halfVec ? r_curr.Update(w_prev, r_prev.halfVectorCopyShift, r_prev.wh_local, whdotwo_curr, r_prev.lobe, r_prev.le, r_prev.lightIdx, r_prev.bary, rng)
                            : r_curr.Update(w_prev, r_prev.le, r_prev.lightIdx, r_prev.bary, rng)) r_curr.target = target_curr;
        }
    }
    float targetLum = Luminance(r_curr.target);
    r_curr.W = targetLum > 0.0f ? r_curr.w_sum / targetLum : 0.0f;
    r_curr.M = newM;
}

// un-swizzled SV_GroupID of the 8x8 group that shades pixel group (sx, sy): Common.hlsli:127-157 inverted (tile = 16 groups wide)
ZR_HD void UnswizzleGid(uint32_t sx, uint32_t sy, uint32_t dispatchDimX, uint32_t dispatchDimY, uint32_t& gx, uint32_t& gy)
{
    const uint32_t tileWidth = 16, numGroupsInTile = tileWidth * dispatchDimY;
    const uint32_t numFullTiles = dispatchDimX / tileWidth;
    const uint32_t tileID = sx / tileWidth, inX = sx % tileWidth;
    uint32_t inFlat;
    if (tileID >= numFullTiles) { const uint32_t lastTileDimX = dispatchDimX - tileWidth * numFullTiles; inFlat = sy * lastTileDimX + inX; }
    else inFlat = sy * tileWidth + inX;
    const uint32_t flat = tileID * numGroupsInTile + inFlat;
    gx = flat % dispatchDimX; gy = flat / dispatchDimX;
}

ZR_HD void WriteFinal(const zr_frame_constants& g, float* finalRGBA, size_t px, V3 li)
{
    li = any_nan(li) ? v3(0.0f) : li;
    float* o = finalRGBA + 4 * px;
    if (g.accumulate && g.camera_static && g.num_frames_camera_static > 1) { o[0] += li.x; o[1] += li.y; o[2] += li.z; }
    else { o[0] = li.x; o[1] = li.y; o[2] = li.z; }
}
ZR_HD V3 EmissiveColor(const GBuf& gb, size_t px)
{ uint32_t v = gb.emissive[px]; return v3(zr_unpack_ufloat(v & 0x7ff, 6), zr_unpack_ufloat((v >> 11) & 0x7ff, 6), zr_unpack_ufloat(v >> 22, 5)); }

ZR_HD Globals MakeGlobals(const DiFrame& F, const zr_frame_constants& g, TravStack stack, uint32_t* cnt)
{
    Globals gl; gl.sc = &F.sc; gl.scPrev = &F.scPrev; gl.frame = &g; gl.emissive = g.num_emissive_triangles != 0; gl.numEmissives = g.num_emissive_triangles; gl.alpha_min = 0; gl.stack = stack; gl.cnt = cnt; gl.maxNumBounces = 1;
    gl.presampled = false; gl.sampleSetIdx = 0;
    return gl;
}

// K5: ReSTIR_DI_Temporal.hlsl main (:263-390) + EstimateDirectLighting (:205-257) for one pixel
ZR_HD void TemporalPixel(const DiFrame& F, const zr_frame_constants& g, uint32_t x, uint32_t y, TravStack stack, uint32_t* cnt)
{
    const DiParams& prm = F.prm;
    const size_t px = Pix(F.gb, x, y);
    GFlags flags = DecodeFlags(F.gb.mr[px]);
    float* o = F.finalRGBA + 4 * px;
    if (flags.invalid)
    {
        // ReSTIR_DI_Temporal.hlsl:274-286: prev * (N > 1) + Le_SkyWithSunDisk (0 while the scene has no sky-view LUT: no ZR_PASS_SKY rendered)
        if (prm.accumulate)
        {
            const float k = g.num_frames_camera_static > 1 ? 1.0f : 0.0f;
            const V3 sky = F.sc.sky.data ? Le_SkyWithSunDisk(F.sc.sky, g, x, y) : v3(0.0f);
            o[0] = o[0] * k + sky.x; o[1] = o[1] * k + sky.y; o[2] = o[2] * k + sky.z;
        }
        else { o[0] = 0; o[1] = 0; o[2] = 0; }
        return;
    }
    if (flags.emissive && !prm.doSpatial)
    {
        V3 le = EmissiveColor(F.gb, px);
        if (prm.accumulate) { o[0] += le.x; o[1] += le.y; o[2] += le.z; }
        else { o[0] = le.x; o[1] = le.y; o[2] = le.z; }
        return;
    }
    const Camera cam = CurrCamera(g);
    PixelSurface ps = LoadPixelSurface(F.gb, cam, x, y, g.frame_num, px);
    if (kPrepDi) PrepareWo(F.sc.rho, ps.surface, kPrepDi);      // the pixel's surface is evaluated once per light candidate and per temporal target
    const uint32_t dispX = (g.render_width + 7) / 8, dispY = (g.render_height + 7) / 8;
    uint32_t ugx, ugy; UnswizzleGid(x / 8, y / 8, dispX, dispY, ugx, ugy);
    Rng rng_group = Rng::Init(ugx, ugy, g.frame_num);
    const uint32_t sampleSetIdx = rng_group.UniformUintBounded_Faster(prm.numSampleSets);
    Rng rng = Rng::Init(x, y, g.frame_num);
    Globals gl = MakeGlobals(F, g, stack, cnt);
    const int numBsdfSamples = !ps.surface.GlossSpecular() && ps.roughness < 0.3f ? 2 : 1;
    Reservoir r = RIS_InitialCandidates(gl, g, prm, ps.pos, ps.normal, ps.surface, sampleSetIdx, numBsdfSamples, rng);
    if (prm.doTemporal)
    {
        V2 motionVec = DecodeMotion(F.gb.motion[px]);
        const V2 currUV = v2(((float)x + 0.5f) / (float)g.render_width, ((float)y + 0.5f) / (float)g.render_height);
        V2 prevUV = currUV - motionVec;
        TemporalCandidate tc = FindTemporalCandidate(F, g, ps.pos, ps.normal, ps.roughness, ps.surface, prevUV);
        if (tc.valid) TemporalResample1(gl, F, ps.pos, ps.normal, ps.surface, tc, r, rng);
        if (prm.doSpatial)
        {
            bool disoccluded = !tc.valid && ((motionVec.x * motionVec.x + motionVec.y * motionVec.y) > 0);
            r.target = disoccluded ? -r.target : r.target;
            r.target = rpt::Sanitize3(r.target);
            // the reference's TARGET texture is R16G16B16A16_FLOAT (DirectLighting.h:71): the spatial pass reads fp16-rounded values
            F.target[px] = f4(zr_round_f16(r.target.x), zr_round_f16(r.target.y), zr_round_f16(r.target.z), 0.0f);
        }
    }
    if (prm.writeReservoirs) r.Write(F.cur, px, prm.M_max, prm.halfVec != 0);
    if (!prm.doSpatial || !prm.doTemporal) WriteFinal(g, F.finalRGBA, px, r.target * r.W);
}

// PairwiseMIS.hlsli:11-231 (no half-vector shift: Jacobians are 1)
struct PairwiseMIS { Reservoir r_s; float m_c; uint32_t M_s; uint32_t k; };
ZR_HD float Compute_m_i(const PairwiseMIS& p, const Reservoir& r_c, const Reservoir& r_i, float targetLum, float jacobian)
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
ZR_HD void Stream(PairwiseMIS& p, const Globals& gl, const DiParams& prm, const Reservoir& r_c, V3 pos_c, V3 normal_c, Surface surface_c, const Reservoir& r_i, V3 pos_i,
    V3 normal_i, Surface surface_i, Rng& rng)
{
    const RhoView& rho = gl.sc->rho;
    V3 target_c_y_i = v3(0.0f), target_i_y_c = v3(0.0f);
    float m_i = 0;
    if (r_i.lightIdx != 0xffffffffu)
    {
        float jacobian_i_to_c = 0;
        if (IsShiftInvertible(prm, r_i, surface_c))
        {
            jacobian_i_to_c = 1; V3 wh_c = v3(0.0f);
            if (prm.halfVec)
            {
                wh_c = FromTangentFrameToWorld(normal_c, r_i.wh_local);
                V3 wh_i = FromTangentFrameToWorld(normal_i, r_i.wh_local);
                float whdotwo_i = zr_abs(dot(surface_i.wo, wh_i));
                float whdotwo_c = zr_abs(dot(surface_c.wo, wh_c));
                jacobian_i_to_c = whdotwo_i > 0 ? whdotwo_c / whdotwo_i : 0;
                jacobian_i_to_c = r_c.halfVectorCopyShift ? jacobian_i_to_c : 1;      // (sic: r_c, PairwiseMIS.hlsli:70)
            }
            if (prm.halfVec && r_i.halfVectorCopyShift) (void)HalfVectorOffsetTarget(gl, pos_c, normal_c, surface_c, wh_c, target_c_y_i);
            else
            {
                EmissiveData e = InitEmissiveData(*gl.sc, r_i.lightIdx, r_i.bary);
                SetSurfacePos(e, pos_c);
                float dwdA_ = dWdA(e);
                surface_c.SetWi(e.wi, normal_c);
                target_c_y_i = r_i.le * dwdA_;
                if (dot(target_c_y_i, target_c_y_i) > 0)
                    target_c_y_i = target_c_y_i * (VisibilitySegmentApprox(gl, pos_c, e.wi, e.t, normal_c, e.ID, surface_c.Transmissive()) ? 1.0f : 0.0f);
            }
            target_c_y_i = target_c_y_i * Unified(rho, surface_c).f;
        }
        m_i = Compute_m_i(p, r_c, r_i, Luminance(target_c_y_i), jacobian_i_to_c);
    }
    float jacobian_c_to_i = 0;
    if (r_c.lightIdx != 0xffffffffu)
    {
        bool invertible = IsShiftInvertible(prm, r_c, surface_i);
        V3 wh_i = v3(0.0f);
        if (invertible)
        {
            jacobian_c_to_i = 1;
            if (prm.halfVec)
            {
                wh_i = FromTangentFrameToWorld(normal_i, r_c.wh_local);
                V3 wh_c = FromTangentFrameToWorld(normal_c, r_c.wh_local);
                float whdotwo_i = zr_abs(dot(surface_i.wo, wh_i));
                float whdotwo_c = zr_abs(dot(surface_c.wo, wh_c));
                jacobian_c_to_i = whdotwo_c == 0 ? 0 : whdotwo_i / whdotwo_c;
                jacobian_c_to_i = r_c.halfVectorCopyShift ? jacobian_c_to_i : 1;
            }
        }
        if (invertible && prm.halfVec && r_i.halfVectorCopyShift) (void)HalfVectorOffsetTarget(gl, pos_i, normal_i, surface_i, wh_i, target_i_y_c);      // (sic: r_i, PairwiseMIS.hlsli:141)
        else if (invertible)
        {
            V3 wi_i = r_c.lightPos - pos_i;
            const bool isZero = dot(wi_i, wi_i) == 0;
            float t_i = isZero ? 0 : length(wi_i);
            wi_i = isZero ? v3(0.0f) : wi_i / t_i;
            surface_i.SetWi(wi_i, normal_i);
            const V3 ln = dot(r_c.lightNormal, -wi_i) < 0 && r_c.doubleSided ? -r_c.lightNormal : r_c.lightNormal;
            const float cosThetaPrime = zr_saturate(dot(ln, -wi_i));
            const float dwdA_ = isZero ? 0 : cosThetaPrime / (t_i * t_i);
            target_i_y_c = r_c.le * dwdA_;
            if (dot(target_i_y_c, target_i_y_c) > 0)
                target_i_y_c = target_i_y_c * (VisibilitySegmentApprox(gl, pos_i, wi_i, t_i, normal_i, r_c.lightID, surface_i.Transmissive()) ? 1.0f : 0.0f);
        }
        target_i_y_c = target_i_y_c * Unified(rho, surface_i).f;
    }
    Update_m_c(p, r_c, r_i, Luminance(target_i_y_c), jacobian_c_to_i);
    if (r_i.lightIdx != 0xffffffffu)
    {
        const float w_i = m_i * Luminance(target_c_y_i) * r_i.W;
        if (prm.halfVec ? p.r_s.Update(w_i, r_i.halfVectorCopyShift, r_i.wh_local, 0.0f /*unused*/, r_i.lobe, r_i.le, r_i.lightIdx, r_i.bary, rng)
                        : p.r_s.Update(w_i, r_i.le, r_i.lightIdx, r_i.bary, rng)) p.r_s.target = target_c_y_i;
    }
    p.M_s += r_i.M;
}
ZR_HD void End(PairwiseMIS& p, const DiParams& prm, const Reservoir& r_c, Rng& rng)
{
    const float w_c = p.m_c * r_c.w_sum;
    if (prm.halfVec ? p.r_s.Update(w_c, r_c.halfVectorCopyShift, r_c.wh_local, 0.0f /*unused*/, r_c.lobe, r_c.le, r_c.lightIdx, r_c.bary, rng)
                    : p.r_s.Update(w_c, r_c.le, r_c.lightIdx, r_c.bary, rng)) p.r_s.target = r_c.target;
    p.r_s.M = p.M_s & 0xffffu;
    const float targetLum = Luminance(p.r_s.target);
    p.r_s.W = targetLum > 0 ? p.r_s.w_sum / (targetLum * (float)(1 + p.k)) : 0;
}

// K6: ReSTIR_DI_Spatial.hlsl main, cut at its WaveActiveSum
struct SpatialLane { bool active, disoccluded; uint32_t x, y; size_t px; PixelSurface ps; Reservoir r; };
ZR_HD void SpatialPhase0(const DiFrame& F, const zr_frame_constants& g, uint32_t x, uint32_t y, SpatialLane& a)
{
    a.active = false; a.disoccluded = false; a.x = x; a.y = y;
    if (!F.Owns(x, y)) return;
    a.px = Pix(F.gb, x, y);
    GFlags flags = DecodeFlags(F.gb.mr[a.px]);
    if (flags.invalid) return;
    if (flags.emissive)
    {
        V3 le = EmissiveColor(F.gb, a.px);
        float* o = F.finalRGBA + 4 * a.px;
        if (F.prm.accumulate) { o[0] += le.x; o[1] += le.y; o[2] += le.z; }
        else { o[0] = le.x; o[1] = le.y; o[2] = le.z; }
        return;
    }
    a.active = true;
    const Camera cam = CurrCamera(g);
    a.ps = LoadPixelSurface(F.gb, cam, x, y, g.frame_num, a.px);
    if (kPrepDi) PrepareWo(F.sc.rho, a.ps.surface, kPrepDi);      // ... once per spatial neighbour
    Reservoir r = LoadReservoir(F.cur, a.px, F.prm.halfVec != 0);
    if (r.lightIdx != 0xffffffffu)
    {
        EmissiveData e = InitEmissiveData(F.sc, r.lightIdx, r.bary);
        r.lightID = e.ID; r.lightPos = e.lightPos; r.lightNormal = e.lightNormal; r.doubleSided = e.doubleSided;
        r.target = xyz(F.target[a.px]);
        a.disoccluded = r.target.x < 0 || r.target.y < 0 || r.target.z < 0;
        r.target = v3(zr_abs(r.target.x), zr_abs(r.target.y), zr_abs(r.target.z));
    }
    a.r = r;
}
ZR_HD void SpatialPhase1(const DiFrame& F, const zr_frame_constants& g, SpatialLane& a, uint32_t waveDisoccluded, TravStack stack, uint32_t* cnt)
{
    if (!a.active) return;
    const DiParams& prm = F.prm;
    const uint32_t W = g.render_width, H = g.render_height;
    bool disoccluded = a.disoccluded;
    if (prm.flags & ZR_DI_EXTRA_DISOCCLUSION_SAMPLING) disoccluded = disoccluded && (waveDisoccluded > 3);
    const uint32_t dispX = (W + 7) / 8, dispY = (H + 7) / 8;
    uint32_t ugx, ugy; UnswizzleGid(a.x / 8, a.y / 8, dispX, dispY, ugx, ugy);
    Rng rng_group = Rng::Init(ugx, ugy, g.frame_num);
    (void)rng_group.UniformUintBounded_Faster(prm.numSampleSets);
    Rng rng = Rng::Init(a.x, a.y, g.frame_num);
    int numSamples = !(prm.flags & ZR_DI_STOCHASTIC_SPATIAL) || (rng_group.Uniform() < kProbExtraSpatial) ? kMinSpatial + kExtraSpatial : kMinSpatial;
    numSamples = !disoccluded ? numSamples : kMaxSpatial;
    Globals gl = MakeGlobals(F, g, stack, cnt);
    const Camera cam = CurrCamera(g);
    // SpatialResample (Resampling.hlsli:341-519)
    const float u0 = rng.Uniform();
    const int offset = (int)rng.UniformUintBounded_Faster(8);
    const float theta = u0 * ZR_TWO_PI;
    float sinTheta, cosTheta; zr_sincos(theta, &sinTheta, &cosTheta);
    PairwiseMIS pw; pw.r_s = InitReservoir(); pw.m_c = 1.0f; pw.M_s = a.r.M; pw.k = (uint32_t)numSamples;
    uint32_t candX[kMaxSpatial], candY[kMaxSpatial];
    uint32_t k = 0;
    for (int i = 0; i < numSamples; i++)
    {
        const uint32_t si = (uint32_t)(offset + i) & 31u;
        const float ux = zr_f16_to_f32(F.sampleSet[2 * si]), uy = zr_f16_to_f32(F.sampleSet[2 * si + 1]);
        float rx = ux * cosTheta + uy * -sinTheta, ry = ux * sinTheta + uy * cosTheta;
        rx *= kSearchRadius; ry *= kSearchRadius;
        const uint32_t sx = zr_f2u_sat(__builtin_rintf((float)a.x + rx)), sy = zr_f2u_sat(__builtin_rintf((float)a.y + ry));
        if (sx >= W || sy >= H) continue;
        if (!rpt::InPlanes(F.gb, (int)sx, (int)sy)) continue;
        const size_t sp = Pix(F.gb, sx, sy);
        GFlags fi = DecodeFlags(F.gb.mr[sp]);
        if (fi.invalid || fi.emissive) continue;
        PixelSurface pi = LoadPixelSurface(F.gb, cam, sx, sy, g.frame_num, sp);
        bool valid = zr_abs(dot(a.ps.normal, pi.pos - a.ps.pos)) <= kMaxPlaneDist * a.ps.z;
        valid = valid && (zr_abs(pi.roughness - a.ps.roughness) < kMaxRoughDiff);
        if (!valid) continue;
        candX[k] = sx; candY[k] = sy; k++;
    }
    pw.k = k;
    for (uint32_t i = 0; i < k; i++)
    {
        const size_t sp = Pix(F.gb, candX[i], candY[i]);
        // the neighbour's surface is built with transmission depth = false (Resampling.hlsli:505-508)
        PixelSurface pi = LoadPixelSurfaceEx(F.gb, cam, candX[i], candY[i], g.frame_num, sp, false);
        Reservoir r_spatial = LoadReservoir(F.cur, sp, prm.halfVec != 0);
        Stream(pw, gl, prm, a.r, a.ps.pos, a.ps.normal, a.ps.surface, r_spatial, pi.pos, pi.normal, pi.surface, rng);
    }
    End(pw, prm, a.r, rng);
    WriteFinal(g, F.finalRGBA, a.px, pw.r_s.target * pw.r_s.W);
}

} // namespace rdi
} // namespace zr
