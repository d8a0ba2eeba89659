// ORACLE -- test infrastructure only (see zro_math.h header).  PARITY UNPINNED against the reference (no executable
// reference exists for this path); follows the shaders line by line.
//
// zro_rdi.h: CPU restatement of ReSTIR DI for emissive lights (K5 / K6); USE_HALF_VECTOR_COPY_SHIFT (Params.hlsli:12, 0 in the reference's tree) is the run-time
// switch Params::halfVec here (ZR_DI_HALF_VECTOR_COPY_SHIFT), pinned for both values against the reference's shaders compiled with the macro at 0 and at 1
// (oracle/_ref.mk: libzref_di_e1.so / libzref_di_e1h.so):
//   DirectLighting/Emissive/ReSTIR_DI_Temporal.hlsl:29-390, ReSTIR_DI_Spatial.hlsl:24-192, Resampling.hlsli:10-521,
//   PairwiseMIS.hlsli:11-231, Reservoir.hlsli:11-226, Util.hlsli:11-119, Params.hlsli; host order DirectLighting.cpp:166-296.
// Pinned where the reference leaves it open: Le_SkyWithSunDisk for miss pixels = 0 while the scene has no sky-view LUT;
// ftou of a negative sample position = 0 (D3D rule); the spatial pass's WaveActiveSum runs over the 8x8 pixel group.
#pragma once
#include "zro_rpt.h"

namespace zro {
namespace RDI {

using RPT::GBufRead; using RPT::GFlags; using RPT::DecodeFlags; using RPT::Roughness; using RPT::DecodeMotion;
using BSDF::LOBE; using RPT::IsLobeValid; using RPT::LobeAlpha;
using RPT::Camera; using RPT::CurrCamera; using RPT::PrevCamera; using RPT::PixelSurface; using RPT::LoadPixelSurface;

static const int NUM_LIGHT_CANDIDATES = 3;
static const int MIN_NUM_SPATIAL_SAMPLES = 1, NUM_EXTRA_SPATIAL_SAMPLES = 1, MAX_NUM_SPATIAL_SAMPLES = 4;
static const float PROB_EXTRA_SPATIAL_SAMPLES = 0.6f;
static const float SPATIAL_SEARCH_RADIUS = 16.0f;
static const float MAX_PLANE_DIST_REUSE = 1e-1f, MAX_ROUGHNESS_DIFF_REUSE = 0.15f;

// Math.hlsli:308-322
static inline float3 WorldToTangentFrame(float3 normal, float3 w)
{ Math::CoordinateSystem onb = Math::CoordinateSystem::Build(normal); return f3(dot(onb.b1, w), dot(onb.b2, w), dot(normal, w)); }
static inline float3 FromTangentFrameToWorld(float3 normal, float3 w_local)
{ Math::CoordinateSystem onb = Math::CoordinateSystem::Build(normal); return w_local.x * onb.b1 + w_local.y * onb.b2 + w_local.z * normal; }
// BSDF.hlsli:62-98
static inline uint32_t LobeToValue(LOBE t)
{ return t == LOBE::DIFFUSE_R ? 0u : t == LOBE::DIFFUSE_T ? 1u : t == LOBE::GLOSSY_R ? 2u : t == LOBE::GLOSSY_T ? 3u : t == LOBE::COAT ? 4u : 5u; }
static inline LOBE LobeFromValue(uint32_t x)
{ return x == 0 ? LOBE::DIFFUSE_R : x == 1 ? LOBE::DIFFUSE_T : x == 2 ? LOBE::GLOSSY_R : x == 3 ? LOBE::GLOSSY_T : x == 4 ? LOBE::COAT : LOBE::ALL; }

// Reservoir.hlsli:11-213
struct Reservoir
{
    float w_sum, W; float3 le; uint32_t lightIdx; float2 bary; uint16_t M;
    float3 target; uint32_t lightID; float3 lightPos, lightNormal; bool doubleSided;
    // half-vector copy shift (USE_HALF_VECTOR_COPY_SHIFT == 1): the sample was drawn from a lobe narrower than alpha_min and is reused by copying its half vector
    // in the shading frame (wh_local) and re-tracing the reflected ray; partialJacobian = |wh . wo| where it was drawn.  (wh_local is not initialised by the
    // reference's Init(); it is only ever used behind halfVectorCopyShift, which Init() clears: 0 here.)
    bool halfVectorCopyShift; LOBE lobe; float3 wh_local; float partialJacobian;
    static Reservoir Init()
    {
        Reservoir r; r.le = f3(0.0f); r.M = 0; r.w_sum = 0; r.W = 0; r.lightIdx = 0xffffffffu; r.bary = {0, 0};
        r.target = f3(0.0f); r.lightID = 0xffffffffu; r.lightPos = f3(0.0f); r.lightNormal = f3(0.0f); r.doubleSided = false;
        r.halfVectorCopyShift = false; r.lobe = LOBE::ALL; r.wh_local = f3(0.0f); r.partialJacobian = 1;
        return r;
    }
    bool Update(float weight, float3 le_, uint32_t lightIdx_, float2 bary_, RNG& rng)
    {
        if (zr_isnan(weight)) return false;
        M += 1;
        if (weight == 0) return false;
        w_sum += weight;
        if (rng.Uniform() < (weight / w_sum)) { le = le_; lightIdx = lightIdx_; bary = bary_; halfVectorCopyShift = false; lobe = LOBE::ALL; return true; }
        return false;
    }
    // Reservoir.hlsli:56-91: a BSDF-sampled candidate
    bool Update(float weight, bool halfVecShift, float3 wi, float3 wo, float3 normal, LOBE lb, float3 le_, uint32_t lightIdx_, float2 bary_, RNG& rng)
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
                float3 wh = normalize(wo + wi);
                wh_local = WorldToTangentFrame(normal, wh);
                partialJacobian = zr_abs(dot(wh, wo));
            }
            return true;
        }
        return false;
    }
    // Reservoir.hlsli:93-119: a reused sample (its half vector travels in the shading frame)
    bool Update(float weight, bool halfVecShift, float3 wh, float whdotwo, LOBE lb, float3 le_, uint32_t lightIdx_, float2 bary_, RNG& rng)
    {
        if (zr_isnan(weight)) return false;
        M += 1;
        if (weight == 0) return false;
        w_sum += weight;
        if (rng.Uniform() < (weight / w_sum))
        { le = le_; lightIdx = lightIdx_; bary = bary_; halfVectorCopyShift = halfVecShift; lobe = lb; wh_local = wh; partialJacobian = whdotwo; return true; }
        return false;
    }
    static Reservoir Load(const uint32_t* A, const float* B, size_t i, bool halfVec = false)
    {
        const uint32_t* a = A + 4 * i;
        Reservoir r = Init();
        r.M = (uint16_t)((a[2] >> 16) & 0x1f);
        r.w_sum = B[2 * i]; r.W = B[2 * i + 1];
        r.le = f3(zr_f16_to_f32((uint16_t)(a[1] & 0xffff)), zr_f16_to_f32((uint16_t)(a[1] >> 16)), zr_f16_to_f32((uint16_t)(a[2] & 0xffff)));
        r.lightIdx = a[3];
        r.bary = {(float)(a[0] & 0xffff) / 65535.0f, (float)(a[0] >> 16) / 65535.0f};
        if (halfVec)
        {   // Reservoir.hlsli:156-160: metadata bit 5 = the flag, bits 6..8 = the lobe; A.x is read BOTH as the barycentrics and as the oct-encoded half vector
            const uint32_t metadata = a[2] >> 16;
            r.halfVectorCopyShift = ((metadata >> 5) & 0x1u) != 0;
            r.lobe = LobeFromValue((metadata >> 6) & 0x7u);
            const uint16_t e[2] = {(uint16_t)(a[0] & 0xffff), (uint16_t)(a[0] >> 16)};
            r.wh_local = Math::DecodeOct32(e);
        }
        return r;
    }
    void Write(uint32_t* A, float* B, size_t i, uint16_t M_max, bool halfVec = false) const
    {
        uint32_t lx = zr_f32_to_f16(le.x), ly = zr_f32_to_f16(le.y), lz = zr_f32_to_f16(le.z);
        uint32_t M_capped = std::min<uint32_t>(M, M_max);
        uint32_t bx = Math::FloatToUNorm16(bary.x), by = Math::FloatToUNorm16(bary.y);
        uint32_t metadata = M_capped;
        if (halfVec)
        {   // Reservoir.hlsli:177-184
            metadata = M_capped | ((halfVectorCopyShift ? 1u : 0u) << 5) | (LobeToValue(lobe) << 6);
            if (halfVectorCopyShift) { uint16_t e[2]; Math::EncodeOct32(wh_local, e); bx = e[0]; by = e[1]; }
        }
        uint32_t* a = A + 4 * i;
        a[0] = (by << 16) | bx; a[1] = (ly << 16) | lx; a[2] = (metadata << 16) | lz; a[3] = lightIdx;
        B[2 * i] = w_sum; B[2 * i + 1] = W;
    }
};

// Util.hlsli:11-57
struct EmissiveData
{
    float3 wi; float t; uint32_t ID; float3 lightPos, lightNormal; bool doubleSided;
    static EmissiveData Init(const Scene& sc, uint32_t lightIdx, float2 bary)
    {
        EmissiveData ret;
        EmTri tri; tri.t = sc.emissives[lightIdx];
        ret.ID = tri.t.id;
        const float3 vtx0 = tri.Vtx0(), vtx1 = tri.V1(), vtx2 = tri.V2();
        ret.lightPos = (1.0f - bary.x - bary.y) * vtx0 + bary.x * vtx1 + bary.y * vtx2;
        ret.lightNormal = cross(vtx1 - vtx0, vtx2 - vtx0);
        ret.lightNormal = dot(ret.lightNormal, ret.lightNormal) == 0 ? ret.lightNormal : normalize(ret.lightNormal);
        ret.doubleSided = tri.IsDoubleSided();
        ret.wi = f3(0.0f); ret.t = 0;
        return ret;
    }
    void SetSurfacePos(float3 pos)
    {
        wi = lightPos - pos;
        t = dot(wi, wi) == 0 ? 0 : length(wi);
        wi = t == 0 ? f3(0.0f) : wi / t;
        lightNormal = doubleSided && dot(-wi, lightNormal) < 0 ? -lightNormal : lightNormal;
    }
    float dWdA() const
    {
        float cosThetaPrime = zr_saturate(dot(lightNormal, -wi));
        return t == 0 ? 0 : cosThetaPrime / (t * t);
    }
};

// Util.hlsli:59-119
struct BSDFHitInfo { uint32_t emissiveTriIdx; float2 bary; float3 lightPos; float t; bool hit; };
static BSDFHitInfo FindClosestHit(const Scene& sc, float3 pos, float3 normal, float3 wi, bool transmissive)
{
    BSDFHitInfo ret; ret.hit = false; ret.emissiveTriIdx = 0xffffffffu; ret.bary = {0, 0}; ret.lightPos = f3(0.0f); ret.t = 0;
    float ndotwi = dot(normal, wi);
    if (ndotwi == 0) return ret;
    bool wiBackface = ndotwi < 0;
    if (wiBackface)
    {
        if (transmissive) normal = normal * -1.0f;
        else return ret;
    }
    const float3 adjustedOrigin = RT::OffsetRayRTG(pos, normal);
    sc.counters.n_closest++;
    Scene::RawHit h = sc.Trace(adjustedOrigin, wi, wiBackface ? 3e-4f : 0.0f, ZR_FLT_MAX, ZR_SUBGROUP_ALL, false);
    if (h.hit)
    {
        const WorldTri& T = sc.tris[h.tri];
        const zr_mesh_instance& md = sc.instances[T.mesh_idx];
        if (md.base_emissive_tri_offset == 0xffffffffu) return ret;
        ret.emissiveTriIdx = md.base_emissive_tri_offset + T.prim_idx;
        ret.bary = f2(h.u, h.v);
        ret.lightPos = mad3(h.t, wi, adjustedOrigin);
        ret.t = h.t;
        ret.hit = true;
    }
    return ret;
}

struct Params { uint32_t flags; uint16_t M_max; bool presampled; uint32_t numSampleSets; bool halfVec = false; float alpha_min = 0; };

// Reservoir.hlsli:216-225
static inline bool IsShiftInvertible(const Params& prm, const Reservoir& r_base, const BSDF::ShadingData& surface_offset)
{
    if (!prm.halfVec) return true;
    return !r_base.halfVectorCopyShift || (IsLobeValid(surface_offset, r_base.lobe) && (LobeAlpha(surface_offset, r_base.lobe) <= prm.alpha_min));
}
// the half-vector copy shift's offset path (Resampling.hlsli:146-179, 220-252, PairwiseMIS.hlsli:72-102, 140-168): the copied half vector's reflection is traced from
// the offset surface; target = Le * dwdA of the light it lands on (0 when it misses the lights or hits one from behind).  Returns false when nothing was hit;
// `surface` gets wi = the traced direction whenever a light was hit.
static inline bool HalfVectorOffsetTarget(const Scene& sc, float3 pos, float3 normal, BSDF::ShadingData& surface, float3 wh, float3& target)
{
    const float3 wi_offset = reflect(-surface.wo, wh);
    BSDFHitInfo hitInfo = FindClosestHit(sc, pos, normal, wi_offset, surface.Transmissive());
    if (!hitInfo.hit) return false;
    EmTri emissive; emissive.t = sc.emissives[hitInfo.emissiveTriIdx];
    const float3 le = Light::Le_EmissiveTriangle(sc, emissive, hitInfo.bary);
    const float3 vtx0 = emissive.Vtx0(), vtx1 = emissive.V1(), vtx2 = emissive.V2();
    float3 lightNormal = cross(vtx1 - vtx0, vtx2 - vtx0);
    float twoArea = length(lightNormal);
    lightNormal = dot(lightNormal, lightNormal) == 0 ? f3(0.0f) : lightNormal / twoArea;
    lightNormal = emissive.IsDoubleSided() && dot(-wi_offset, lightNormal) < 0 ? -lightNormal : lightNormal;
    if (dot(-wi_offset, lightNormal) > 0)
    {
        float dwdA = zr_saturate(dot(lightNormal, -wi_offset)) / (hitInfo.t * hitInfo.t);
        target = le * dwdA;
    }
    surface.SetWi(wi_offset, normal);
    return true;
}

// ReSTIR_DI_Temporal.hlsl:29-203
static Reservoir RIS_InitialCandidates(const Scene& sc, const zr_frame_constants& g, const Params& prm, float3 pos, float3 normal,
    BSDF::ShadingData surface, uint32_t sampleSetIdx, int numBsdfSamples, RNG& rng)
{
    Reservoir r = Reservoir::Init();
    const bool specular = surface.GlossSpecular() && (surface.metallic || surface.specTr) && (!surface.Coated() || surface.CoatSpecular());
    const int numLightSamples = !specular ? NUM_LIGHT_CANDIDATES : 0;
    for (int s_b = 0; s_b < numBsdfSamples; s_b++)
    {
        BSDF::BSDFSample bsdfSample = BSDF::SampleBSDF_NoDiffuse(normal, surface, rng);
        float3 wi = bsdfSample.wi;
        float pdf_w = bsdfSample.pdf;
        // ReSTIR_DI_Temporal.hlsl:45-50: glossy reflection or coat with lobe roughness below the threshold
        const bool useHalfVecShift = prm.halfVec ? (LobeAlpha(surface, bsdfSample.lobe) <= prm.alpha_min) : false;
        BSDFHitInfo hitInfo = FindClosestHit(sc, pos, normal, wi, surface.Transmissive());
        float w_b = 0; float3 le = f3(0.0f), lightNormal = f3(0.0f), target = f3(0.0f); uint32_t emissiveID = 0xffffffffu; bool doubleSided = false;
        if (hitInfo.hit)
        {
            EmTri emissive; emissive.t = sc.emissives[hitInfo.emissiveTriIdx];
            le = Light::Le_EmissiveTriangle(sc, emissive, hitInfo.bary);
            const float3 vtx0 = emissive.Vtx0(), vtx1 = emissive.V1(), vtx2 = emissive.V2();
            lightNormal = cross(vtx1 - vtx0, vtx2 - vtx0);
            float twoArea = length(lightNormal);
            lightNormal = dot(lightNormal, lightNormal) == 0 ? f3(0.0f) : lightNormal / twoArea;
            lightNormal = emissive.IsDoubleSided() && dot(-wi, lightNormal) < 0 ? -lightNormal : lightNormal;
            doubleSided = emissive.IsDoubleSided();
            emissiveID = emissive.t.id;
            if (dot(-wi, lightNormal) > 0)
            {
                const float lightSourcePdf = sc.alias[hitInfo.emissiveTriIdx].cached_p_orig;
                const float pdf_light = lightSourcePdf * (1.0f / (0.5f * twoArea));
                const float dwdA = zr_saturate(dot(lightNormal, -wi)) / (hitInfo.t * hitInfo.t);
                pdf_w *= dwdA;
                const bool sampleIsSpecular = (surface.GlossSpecular() && bsdfSample.lobe == BSDF::LOBE::GLOSSY_R) ||
                    (surface.CoatSpecular() && bsdfSample.lobe == BSDF::LOBE::COAT);
                float denom = (float)numBsdfSamples * pdf_w + (float)(!sampleIsSpecular ? 1 : 0) * (float)numLightSamples * pdf_light;
                const float m_i = 1.0f / denom;
                target = le * bsdfSample.f * dwdA;
                w_b = m_i * Math::Luminance(target);
            }
        }
        if (r.Update(w_b, useHalfVecShift, wi, surface.wo, normal, bsdfSample.lobe, le, hitInfo.emissiveTriIdx, hitInfo.bary, rng))
        { r.target = target; r.lightID = emissiveID; r.lightPos = hitInfo.lightPos; r.lightNormal = lightNormal; r.doubleSided = doubleSided; }
    }
    for (int s_l = 0; s_l < numLightSamples; s_l++)
    {
        Light::EmissiveTriSample lightSample; float3 le; float pdf_light; uint32_t emissiveIdx, lightID; bool doubleSided;
        if (prm.presampled)
        {
            uint32_t u = rng.UniformUintBounded_Faster(sc.sampleSetSize);
            const zr_presampled_tri& t = sc.sampleSets[(size_t)sampleSetIdx * sc.sampleSetSize + u];
            lightSample.pos = f3(t.pos); lightSample.normal = Math::DecodeOct32(t.normal);
            lightSample.bary = {(float)t.bary[0] / 65535.0f, (float)t.bary[1] / 65535.0f};
            le = f3(zr_f16_to_f32(t.le[0]), zr_f16_to_f32(t.le[1]), zr_f16_to_f32(t.le[2]));
            pdf_light = t.pdf; emissiveIdx = t.idx; lightID = t.id; doubleSided = t.two_sided != 0;
            if (doubleSided && dot(pos - lightSample.pos, lightSample.normal) < 0) lightSample.normal = -lightSample.normal;
        }
        else
        {
            Light::AliasTableSample entry = Light::AliasTableSample::get(sc, g.num_emissive_triangles, rng);
            EmTri tri; tri.t = sc.emissives[entry.idx];
            lightSample = Light::EmissiveTriSample::get(pos, tri, rng);
            le = Light::Le_EmissiveTriangle(sc, tri, lightSample.bary);
            pdf_light = entry.pdf * lightSample.pdf;
            emissiveIdx = entry.idx; lightID = tri.t.id; doubleSided = tri.IsDoubleSided();
        }
        float3 target = f3(0.0f);
        float3 wi = lightSample.pos - pos;
        const bool isZero = dot(wi, wi) == 0;
        const float t = isZero ? 0 : length(wi);
        wi = isZero ? wi : wi / t;
        const float dwdA = isZero ? 0 : zr_saturate(dot(lightSample.normal, -wi)) / (t * t);
        surface.SetWi(wi, normal);
        if (dot(lightSample.normal, -wi) > 0)
        {
            target = le * BSDF::Unified(surface).f * dwdA;
            if (dot(target, target) > 0)
                target *= RtRayQuery::Visibility_Segment(sc, true, pos, wi, t, normal, lightID, surface.Transmissive()) ? 1.0f : 0.0f;
        }
        const float denom = (float)numLightSamples * pdf_light + (float)numBsdfSamples * BSDF::BSDFSamplerPdf_NoDiffuse(normal, surface, wi, BSDF::NoOp()) * dwdA;
        const float m_l = denom > 0 ? 1.0f / denom : 0;
        const float w_l = m_l * Math::Luminance(target);
        if (r.Update(w_l, le, emissiveIdx, lightSample.bary, rng))
        { r.target = target; r.lightID = lightID; r.lightNormal = lightSample.normal; r.lightPos = lightSample.pos; r.doubleSided = doubleSided; }
    }
    float targetLum = Math::Luminance(r.target);
    r.W = targetLum > 0.0f ? r.w_sum / targetLum : 0.0f;
    return r;
}

// Resampling.hlsli:10-136
struct TemporalCandidate { BSDF::ShadingData surface; float3 pos, normal; int px, py; bool valid; };
static TemporalCandidate FindTemporalCandidate(const zr_frame_constants& g, const GBufRead& gbPrev, float3 pos, float3 normal, float roughness,
    const BSDF::ShadingData& surface, float2 prevUV)
{
    TemporalCandidate c; c.valid = false; c.px = c.py = 0;
    if (prevUV.x < 0 || prevUV.y < 0 || prevUV.x > 1 || prevUV.y > 1) return c;
    const float2 renderDim = {(float)g.render_width, (float)g.render_height};
    const int ppx = (int)(prevUV.x * renderDim.x), ppy = (int)(prevUV.y * renderDim.y);
    if (ppx >= (int)gbPrev.w || ppy >= (int)gbPrev.h) return c;        // prevUV == 1: out of bounds, pinned to "no candidate"
    const size_t pp = (size_t)ppy * gbPrev.w + ppx;
    GFlags pf = DecodeFlags(gbPrev.mr[pp]);
    const float prevRoughness = Roughness(gbPrev.mr[pp]);
    if (pf.invalid || pf.emissive || (zr_abs(prevRoughness - roughness) > MAX_ROUGHNESS_DIFF_REUSE) || (pf.metallic != surface.metallic) ||
        (pf.transmissive != surface.specTr)) return c;
    const Camera pcam = PrevCamera(g);
    PixelSurface ps = LoadPixelSurface(gbPrev, pcam, (uint32_t)ppx, (uint32_t)ppy, g.frame_num - 1, pp);
    float planeDist = dot(normal, ps.pos - pos);
    if (!(zr_abs(planeDist) <= MAX_PLANE_DIST_REUSE * ps.z)) return c;
    c.surface = ps.surface; c.pos = ps.pos; c.normal = ps.normal; c.px = ppx; c.py = ppy; c.valid = true;
    return c;
}

// Resampling.hlsli:130-203
static float OffsetPathTarget_CtT(const Scene& sc, const Params& prm, const Reservoir& r_curr, TemporalCandidate candidate, float3 wh)
{
    if (!IsShiftInvertible(prm, r_curr, candidate.surface)) return 0;
    float3 target_offset = f3(0.0f);
    float3 wi_offset = f3(0.0f);
    float t_offset = 0;
    if (prm.halfVec && r_curr.halfVectorCopyShift)
    {
        if (!HalfVectorOffsetTarget(sc.Prev(), candidate.pos, candidate.normal, candidate.surface, wh, target_offset)) return 0;
    }
    else
    {
    wi_offset = r_curr.lightPos - candidate.pos;
    const bool isZero = dot(wi_offset, wi_offset) == 0;
    t_offset = isZero ? 0 : length(wi_offset);
    wi_offset = isZero ? wi_offset : wi_offset / t_offset;
    candidate.surface.SetWi(wi_offset, candidate.normal);
    float3 lightNormal = r_curr.lightNormal;
    if (r_curr.doubleSided && dot(-wi_offset, lightNormal) < 0) lightNormal = -lightNormal;
    float cosThetaPrime = zr_saturate(dot(lightNormal, -wi_offset));
    const float dwdA = isZero ? 0 : cosThetaPrime / (t_offset * t_offset);
    target_offset = r_curr.le * dwdA;
    }
    target_offset *= BSDF::Unified(candidate.surface).f;
    float targetLum_offset = Math::Luminance(target_offset);
    if (!r_curr.halfVectorCopyShift && targetLum_offset > 0)
        targetLum_offset *= RtRayQuery::Visibility_Segment(sc.Prev(), true, candidate.pos, wi_offset, t_offset, candidate.normal, r_curr.lightID,
            candidate.surface.Transmissive()) ? 1.0f : 0.0f;
    return targetLum_offset;
}
// Resampling.hlsli:205-274
static float3 OffsetPathTarget_TtC(const Scene& sc, const Params& prm, const Reservoir& r_prev, float3 pos, float3 normal, BSDF::ShadingData surface, float3 wh)
{
    if (!IsShiftInvertible(prm, r_prev, surface)) return f3(0.0f);
    float3 target_offset = f3(0.0f);
    float3 wi_offset = f3(0.0f);
    float t_offset = 0;
    uint32_t lightID = 0xffffffffu;
    if (prm.halfVec && r_prev.halfVectorCopyShift)
    {
        if (!HalfVectorOffsetTarget(sc, pos, normal, surface, wh, target_offset)) return f3(0.0f);
    }
    else
    {
        EmissiveData prevEmissive = EmissiveData::Init(sc, r_prev.lightIdx, r_prev.bary);
        prevEmissive.SetSurfacePos(pos);
        wi_offset = prevEmissive.wi; t_offset = prevEmissive.t; lightID = prevEmissive.ID;
        float dwdA = prevEmissive.dWdA();
        surface.SetWi(prevEmissive.wi, normal);
        target_offset = r_prev.le * dwdA;
    }
    target_offset *= BSDF::Unified(surface).f;
    if (!r_prev.halfVectorCopyShift && dot(target_offset, target_offset) > 0)
        target_offset *= RtRayQuery::Visibility_Segment(sc, true, pos, wi_offset, t_offset, normal, lightID, surface.Transmissive()) ? 1.0f : 0.0f;
    return target_offset;
}

// Resampling.hlsli:287-339
static void TemporalResample1(const Scene& sc, const Params& prm, float3 pos, float3 normal, const BSDF::ShadingData& surface, const TemporalCandidate& candidate,
    const uint32_t* prevA, const float* prevB, uint32_t planeW, Reservoir& r_curr, RNG& rng)
{
    Reservoir r_prev = Reservoir::Load(prevA, prevB, (size_t)candidate.py * planeW + candidate.px, prm.halfVec);
    const uint16_t newM = (uint16_t)(r_curr.M + r_prev.M);
    if (r_curr.w_sum != 0)
    {
        float3 wh_prev = FromTangentFrameToWorld(candidate.normal, r_curr.wh_local);
        float whdotwo = zr_abs(dot(candidate.surface.wo, wh_prev));
        float jacobian = r_curr.partialJacobian == 0 ? 0 : whdotwo / r_curr.partialJacobian;
        jacobian = r_curr.halfVectorCopyShift ? jacobian : 1;
        float targetLum_prev = OffsetPathTarget_CtT(sc, prm, r_curr, candidate, wh_prev);
        const float numerator = (float)r_curr.M * Math::Luminance(r_curr.target);
        const float denom = numerator + (float)r_prev.M * targetLum_prev * jacobian;
        const float m_curr = denom > 0 ? numerator / denom : 0;
        r_curr.w_sum *= m_curr;
    }
    if (r_prev.lightIdx != 0xffffffffu)
    {
        float3 wh_curr = FromTangentFrameToWorld(normal, r_prev.wh_local);
        float3 wh_prev = FromTangentFrameToWorld(candidate.normal, r_prev.wh_local);
        float whdotwo_prev = zr_abs(dot(candidate.surface.wo, wh_prev));
        float whdotwo_curr = zr_abs(dot(surface.wo, wh_curr));
        float jacobian = whdotwo_prev > 0 ? whdotwo_curr / whdotwo_prev : 0;
        jacobian = r_prev.halfVectorCopyShift ? jacobian : 1;
        const float3 target_curr = OffsetPathTarget_TtC(sc, prm, r_prev, pos, normal, surface, wh_curr);
        const float targetLum_curr = Math::Luminance(target_curr);
        if (targetLum_curr > 0)
        {
            const float targetLum_prev = r_prev.W > 0 ? r_prev.w_sum / r_prev.W : 0;
            const float numerator = (float)r_prev.M * targetLum_prev;
            const float denom = numerator / jacobian + (float)r_curr.M * targetLum_curr;
            const float m_prev = denom > 0 ? numerator / denom : 0;
            const float w_prev = m_prev * targetLum_curr * r_prev.W;
            if (r_curr.Update(w_prev, r_prev.halfVectorCopyShift, r_prev.wh_local, whdotwo_curr, r_prev.lobe, r_prev.le, r_prev.lightIdx, r_prev.bary, rng))
                r_curr.target = target_curr;
        }
    }
    float targetLum = Math::Luminance(r_curr.target);
    r_curr.W = targetLum > 0.0f ? r_curr.w_sum / targetLum : 0.0f;
    r_curr.M = newM;
}

// PairwiseMIS.hlsli:11-231 (no half-vector shift: Jacobians are 1)
struct PairwiseMIS
{
    Reservoir r_s; float m_c; uint32_t M_s; uint16_t k;
    static PairwiseMIS Init(uint16_t numStrategies, const Reservoir& r_c)
    { PairwiseMIS p; p.r_s = Reservoir::Init(); p.m_c = 1.0f; p.M_s = r_c.M; p.k = numStrategies; return p; }
    float Compute_m_i(const Reservoir& r_c, const Reservoir& r_i, float targetLum, float jacobian) const
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
    void Stream(const Scene& sc, const Params& prm, const Reservoir& r_c, float3 pos_c, float3 normal_c, BSDF::ShadingData surface_c, const Reservoir& r_i, float3 pos_i,
        float3 normal_i, BSDF::ShadingData surface_i, RNG& rng)
    {
        float3 target_c_y_i = f3(0.0f), target_i_y_c = f3(0.0f);
        float m_i = 0;
        if (r_i.lightIdx != 0xffffffffu)
        {
            float jacobian_i_to_c = 0;
            if (IsShiftInvertible(prm, r_i, surface_c))
            {
                float3 wh_c = FromTangentFrameToWorld(normal_c, r_i.wh_local);
                float3 wh_i = FromTangentFrameToWorld(normal_i, r_i.wh_local);
                float whdotwo_i = zr_abs(dot(surface_i.wo, wh_i));
                float whdotwo_c = zr_abs(dot(surface_c.wo, wh_c));
                jacobian_i_to_c = whdotwo_i > 0 ? whdotwo_c / whdotwo_i : 0;
                jacobian_i_to_c = r_c.halfVectorCopyShift ? jacobian_i_to_c : 1;      // (sic: r_c, PairwiseMIS.hlsli:70)
                if (prm.halfVec && r_i.halfVectorCopyShift)
                    (void)HalfVectorOffsetTarget(sc, pos_c, normal_c, surface_c, wh_c, target_c_y_i);
                else
                {
                    EmissiveData emissive_i = EmissiveData::Init(sc, r_i.lightIdx, r_i.bary);
                    emissive_i.SetSurfacePos(pos_c);
                    float dwdA = emissive_i.dWdA();
                    surface_c.SetWi(emissive_i.wi, normal_c);
                    target_c_y_i = r_i.le * dwdA;
                    if (dot(target_c_y_i, target_c_y_i) > 0)
                        target_c_y_i *= RtRayQuery::Visibility_Segment(sc, true, pos_c, emissive_i.wi, emissive_i.t, normal_c, emissive_i.ID, surface_c.Transmissive()) ? 1.0f : 0.0f;
                }
                target_c_y_i *= BSDF::Unified(surface_c).f;
            }
            m_i = Compute_m_i(r_c, r_i, Math::Luminance(target_c_y_i), jacobian_i_to_c);
        }
        float jacobian_c_to_i = 0;
        if (r_c.lightIdx != 0xffffffffu)
        {
            if (IsShiftInvertible(prm, r_c, surface_i))
            {
                float3 wh_i = FromTangentFrameToWorld(normal_i, r_c.wh_local);
                float3 wh_c = FromTangentFrameToWorld(normal_c, r_c.wh_local);
                float whdotwo_i = zr_abs(dot(surface_i.wo, wh_i));
                float whdotwo_c = zr_abs(dot(surface_c.wo, wh_c));
                jacobian_c_to_i = whdotwo_c == 0 ? 0 : whdotwo_i / whdotwo_c;
                jacobian_c_to_i = r_c.halfVectorCopyShift ? jacobian_c_to_i : 1;
                if (prm.halfVec && r_i.halfVectorCopyShift)      // (sic: r_i, PairwiseMIS.hlsli:141)
                    (void)HalfVectorOffsetTarget(sc, pos_i, normal_i, surface_i, wh_i, target_i_y_c);
                else
                {
                    float3 wi_i = r_c.lightPos - pos_i;
                    const bool isZero = dot(wi_i, wi_i) == 0;
                    float t_i = isZero ? 0 : length(wi_i);
                    wi_i = isZero ? f3(0.0f) : wi_i / t_i;
                    surface_i.SetWi(wi_i, normal_i);
                    const float3 lightNormal = dot(r_c.lightNormal, -wi_i) < 0 && r_c.doubleSided ? -r_c.lightNormal : r_c.lightNormal;
                    const float cosThetaPrime = zr_saturate(dot(lightNormal, -wi_i));
                    const float dwdA = isZero ? 0 : cosThetaPrime / (t_i * t_i);
                    target_i_y_c = r_c.le * dwdA;
                    if (dot(target_i_y_c, target_i_y_c) > 0)
                        target_i_y_c *= RtRayQuery::Visibility_Segment(sc, true, pos_i, wi_i, t_i, normal_i, r_c.lightID, surface_i.Transmissive()) ? 1.0f : 0.0f;
                }
            }
            target_i_y_c *= BSDF::Unified(surface_i).f;
        }
        Update_m_c(r_c, r_i, Math::Luminance(target_i_y_c), jacobian_c_to_i);
        if (r_i.lightIdx != 0xffffffffu)
        {
            const float w_i = m_i * Math::Luminance(target_c_y_i) * r_i.W;
            if (r_s.Update(w_i, r_i.halfVectorCopyShift, r_i.wh_local, 0.0f /*unused*/, r_i.lobe, r_i.le, r_i.lightIdx, r_i.bary, rng)) r_s.target = target_c_y_i;
        }
        M_s += r_i.M;
    }
    void End(const Reservoir& r_c, RNG& rng)
    {
        const float w_c = m_c * r_c.w_sum;
        if (r_s.Update(w_c, r_c.halfVectorCopyShift, r_c.wh_local, 0.0f /*unused*/, r_c.lobe, r_c.le, r_c.lightIdx, r_c.bary, rng)) r_s.target = r_c.target;
        r_s.M = (uint16_t)M_s;
        const float targetLum = Math::Luminance(r_s.target);
        r_s.W = targetLum > 0 ? r_s.w_sum / (targetLum * (float)(1 + k)) : 0;
    }
};

struct State
{
    uint32_t w = 0, h = 0;
    std::vector<uint32_t> A[2]; std::vector<float> B[2]; std::vector<float> target;
    std::vector<uint16_t> sampleSet;      // 32 x half2
    bool temporalValid = false; int currIdx = 0;
    void Resize(uint32_t w_, uint32_t h_)
    {
        w = w_; h = h_; size_t n = (size_t)w * h;
        for (int i = 0; i < 2; i++) { A[i].assign(4 * n, 0); B[i].assign(2 * n, 0); }
        target.assign(4 * n, 0); temporalValid = false; currIdx = 0;
    }
};

// un-swizzled SV_GroupID of the 8x8 group that shades pixel group (sx, sy) (Common.hlsli:127-157 inverted; tile width 16 groups)
static inline void UnswizzleGid(uint32_t sx, uint32_t sy, uint32_t dispatchDimX, uint32_t dispatchDimY, uint32_t& gx, uint32_t& gy)
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

static void WriteFinal(const zr_frame_constants& g, float* finalRGBA, size_t px, float3 li)
{
    li = any_nan(li) ? f3(0.0f) : li;
    float* o = finalRGBA + 4 * px;
    if (g.accumulate && g.camera_static && g.num_frames_camera_static > 1) { o[0] += li.x; o[1] += li.y; o[2] += li.z; }
    else { o[0] = li.x; o[1] = li.y; o[2] = li.z; }
}
static inline float3 EmissiveColor(const uint32_t* plane, size_t px)
{ uint32_t v = plane[px]; return f3(zr_unpack_ufloat(v & 0x7ff, 6), zr_unpack_ufloat((v >> 11) & 0x7ff, 6), zr_unpack_ufloat(v >> 22, 5)); }

// DirectLighting::Render (DirectLighting.cpp:166-296)
static void Render(const Scene& sc, const zr_frame_constants& g, const zr_gbuffer_planes* gbCurr, const zr_gbuffer_planes* gbPrevPlanes, const zr_params& zp,
    State& st, float* finalRGBA)
{
    BSDF::g_rho = &sc.rhoLUT;
    GBufRead gb(gbCurr);
    const uint32_t W = g.render_width, H = g.render_height;
    const bool doTemporal = st.temporalValid && (zp.flags & ZR_IND_TEMPORAL_RESAMPLE) && gbPrevPlanes;
    const bool doSpatial = doTemporal && (zp.flags & ZR_IND_SPATIAL_RESAMPLE);
    const bool writeReservoirs = doTemporal || !st.temporalValid;       // TEMPORAL_RESAMPLE || RESET_TEMPORAL_TEXTURES
    Params prm; prm.flags = zp.flags; prm.M_max = (uint16_t)zp.m_max_temporal; prm.presampled = zp.presampling != 0;
    prm.numSampleSets = zp.presampling ? zp.num_sample_sets : 0;
    prm.halfVec = (zp.flags & ZR_DI_HALF_VECTOR_COPY_SHIFT) != 0; prm.alpha_min = zp.alpha_min;
    const Camera cam = CurrCamera(g);
    const uint32_t* emissivePlane = (const uint32_t*)gbCurr->plane[ZR_GB_EMISSIVE_COLOR];
    uint32_t* curA = st.A[st.currIdx].data(); float* curB = st.B[st.currIdx].data();
    const uint32_t* prevA = st.A[1 - st.currIdx].data(); const float* prevB = st.B[1 - st.currIdx].data();
    const uint32_t dispX = (W + 7) / 8, dispY = (H + 7) / 8;
    const bool accumulate = g.accumulate && g.camera_static;

    // ---- K5: ReSTIR_DI_Temporal.hlsl main (:263-390)
    for (uint32_t y = 0; y < H; y++) for (uint32_t x = 0; x < W; x++)
    {
        const size_t px = (size_t)y * W + x;
        GFlags flags = DecodeFlags(gb.mr[px]);
        float* o = finalRGBA + 4 * px;
        if (flags.invalid)
        {
            // ReSTIR_DI_Temporal.hlsl:274-286 (0 while no sky-view LUT is bound to the scene)
            if (accumulate)
            {
                const float k = g.num_frames_camera_static > 1 ? 1.0f : 0.0f;
                const float3 sky = sc.sky.data ? Light::Le_SkyWithSunDisk(x, y, g, sc.sky) : f3(0.0f);
                o[0] = o[0] * k + sky.x; o[1] = o[1] * k + sky.y; o[2] = o[2] * k + sky.z;
            }
            else { o[0] = o[1] = o[2] = 0; }
            continue;
        }
        if (flags.emissive && !doSpatial)
        {
            float3 le = EmissiveColor(emissivePlane, px);
            if (accumulate) { o[0] += le.x; o[1] += le.y; o[2] += le.z; }
            else { o[0] = le.x; o[1] = le.y; o[2] = le.z; }
            continue;
        }
        PixelSurface ps = LoadPixelSurface(gb, cam, x, y, g.frame_num, px);
        uint32_t ugx, ugy; UnswizzleGid(x / 8, y / 8, dispX, dispY, ugx, ugy);
        RNG rng_group = RNG::Init(ugx, ugy, g.frame_num);
        const uint32_t sampleSetIdx = rng_group.UniformUintBounded_Faster(prm.numSampleSets);
        RNG rng_thread = RNG::Init(x, y, g.frame_num);
        // EstimateDirectLighting (:205-257)
        const int numBsdfSamples = !ps.surface.GlossSpecular() && ps.roughness < 0.3f ? 2 : 1;
        Reservoir r = RIS_InitialCandidates(sc, g, prm, ps.pos, ps.normal, ps.surface, sampleSetIdx, numBsdfSamples, rng_thread);
        if (doTemporal)
        {
            GBufRead gbPrev(gbPrevPlanes);
            float2 motionVec = DecodeMotion(gb.motion[px]);
            const float2 currUV = {((float)x + 0.5f) / (float)W, ((float)y + 0.5f) / (float)H};
            float2 prevUV = currUV - motionVec;
            TemporalCandidate tc = FindTemporalCandidate(g, gbPrev, ps.pos, ps.normal, ps.roughness, ps.surface, prevUV);
            if (tc.valid) TemporalResample1(sc, prm, ps.pos, ps.normal, ps.surface, tc, prevA, prevB, W, r, rng_thread);
            if (doSpatial)
            {
                bool disoccluded = !tc.valid && (dot(motionVec, motionVec) > 0);
                r.target = disoccluded ? -r.target : r.target;
                float3 t = RPT::Sanitize3(r.target);
                r.target = t;
                // the TARGET texture is R16G16B16A16_FLOAT (DirectLighting.h:71): the spatial pass reads fp16-rounded values
                st.target[4 * px] = zr_round_f16(t.x); st.target[4 * px + 1] = zr_round_f16(t.y); st.target[4 * px + 2] = zr_round_f16(t.z);
            }
        }
        if (writeReservoirs) r.Write(curA, curB, px, prm.M_max, prm.halfVec);
        if (!doSpatial || !doTemporal) WriteFinal(g, finalRGBA, px, r.target * r.W);
    }

    // ---- K6: ReSTIR_DI_Spatial.hlsl main (:24-192); wave = 8x8 pixel group
    if (doSpatial)
    {
        struct Lane { bool active, disoccluded; uint32_t x, y; size_t px; PixelSurface ps; Reservoir r; };
        std::vector<Lane> L(64);
        for (uint32_t gy = 0; gy < dispY; gy++) for (uint32_t gx = 0; gx < dispX; gx++)
        {
            uint32_t waveSum = 0;
            for (uint32_t l = 0; l < 64; l++)
            {
                Lane& a = L[l]; a.active = false; a.disoccluded = false;
                const uint32_t x = gx * 8 + (l & 7), y = gy * 8 + (l >> 3);
                a.x = x; a.y = y;
                if (x >= W || y >= H) continue;
                a.px = (size_t)y * W + x;
                GFlags flags = DecodeFlags(gb.mr[a.px]);
                if (flags.invalid) continue;
                if (flags.emissive)
                {
                    float3 le = EmissiveColor(emissivePlane, a.px);
                    float* o = finalRGBA + 4 * a.px;
                    if (accumulate) { o[0] += le.x; o[1] += le.y; o[2] += le.z; }
                    else { o[0] = le.x; o[1] = le.y; o[2] = le.z; }
                    continue;
                }
                a.active = true;
                a.ps = LoadPixelSurface(gb, cam, x, y, g.frame_num, a.px);
                Reservoir r = Reservoir::Load(curA, curB, a.px, prm.halfVec);
                if (r.lightIdx != 0xffffffffu)
                {
                    EmissiveData e = EmissiveData::Init(sc, r.lightIdx, r.bary);
                    r.lightID = e.ID; r.lightPos = e.lightPos; r.lightNormal = e.lightNormal; r.doubleSided = e.doubleSided;
                    r.target = f3(st.target[4 * a.px], st.target[4 * a.px + 1], st.target[4 * a.px + 2]);
                    a.disoccluded = r.target.x < 0 || r.target.y < 0 || r.target.z < 0;
                    r.target = abs3(r.target);
                }
                a.r = r;
                waveSum += a.disoccluded ? 1u : 0u;
            }
            for (uint32_t l = 0; l < 64; l++)
            {
                Lane& a = L[l];
                if (!a.active) continue;
                bool disoccluded = a.disoccluded;
                if (zp.flags & ZR_DI_EXTRA_DISOCCLUSION_SAMPLING) disoccluded = disoccluded && (waveSum > 3);
                uint32_t ugx, ugy; UnswizzleGid(gx, gy, dispX, dispY, ugx, ugy);
                RNG rng_group = RNG::Init(ugx, ugy, g.frame_num);
                (void)rng_group.UniformUintBounded_Faster(prm.numSampleSets);
                RNG rng = RNG::Init(a.x, a.y, g.frame_num);
                int numSamples = !(zp.flags & ZR_DI_STOCHASTIC_SPATIAL) || (rng_group.Uniform() < PROB_EXTRA_SPATIAL_SAMPLES) ?
                    MIN_NUM_SPATIAL_SAMPLES + NUM_EXTRA_SPATIAL_SAMPLES : MIN_NUM_SPATIAL_SAMPLES;
                numSamples = !disoccluded ? numSamples : MAX_NUM_SPATIAL_SAMPLES;
                // SpatialResample (Resampling.hlsli:341-519)
                Reservoir& r = a.r;
                const float u0 = rng.Uniform();
                const int offset = (int)rng.UniformUintBounded_Faster(8);
                const float theta = u0 * ZR_TWO_PI;
                float sinTheta, cosTheta; zr_sincos(theta, &sinTheta, &cosTheta);
                PairwiseMIS pw = PairwiseMIS::Init((uint16_t)numSamples, r);
                struct Cand { uint32_t x, y; } cand[MAX_NUM_SPATIAL_SAMPLES];
                uint16_t k = 0;
                for (int i = 0; i < numSamples; i++)
                {
                    const uint32_t si = (uint32_t)(offset + i) & 31u;
                    const float ux = zr_f16_to_f32(st.sampleSet[2 * si]), uy = zr_f16_to_f32(st.sampleSet[2 * si + 1]);
                    float rx = ux * cosTheta + uy * -sinTheta, ry = ux * sinTheta + uy * cosTheta;
                    rx *= SPATIAL_SEARCH_RADIUS; ry *= SPATIAL_SEARCH_RADIUS;
                    const uint32_t sx = zr_f2u_sat(__builtin_rintf((float)a.x + rx)), sy = zr_f2u_sat(__builtin_rintf((float)a.y + ry));
                    if (sx >= W || sy >= H) continue;
                    const size_t sp = (size_t)sy * W + sx;
                    GFlags fi = DecodeFlags(gb.mr[sp]);
                    if (fi.invalid || fi.emissive) continue;
                    PixelSurface pi = LoadPixelSurface(gb, cam, sx, sy, g.frame_num, sp);
                    bool valid = zr_abs(dot(a.ps.normal, pi.pos - a.ps.pos)) <= MAX_PLANE_DIST_REUSE * a.ps.z;
                    valid = valid && (zr_abs(pi.roughness - a.ps.roughness) < MAX_ROUGHNESS_DIFF_REUSE);
                    if (!valid) continue;
                    cand[k].x = sx; cand[k].y = sy; k++;
                }
                pw.k = k;
                for (int i = 0; i < k; i++)
                {
                    const size_t sp = (size_t)cand[i].y * W + cand[i].x;
                    // the neighbour's surface is built with transmission depth = false (Resampling.hlsli:505-508)
                    PixelSurface pi = RPT::LoadPixelSurfaceEx(gb, cam, cand[i].x, cand[i].y, g.frame_num, sp, false);
                    Reservoir r_spatial = Reservoir::Load(curA, curB, sp, prm.halfVec);
                    pw.Stream(sc, prm, r, a.ps.pos, a.ps.normal, a.ps.surface, r_spatial, pi.pos, pi.normal, pi.surface, rng);
                }
                pw.End(r, rng);
                r = pw.r_s;
                WriteFinal(g, finalRGBA, a.px, r.target * r.W);
            }
        }
    }
    st.temporalValid = true;
    st.currIdx = 1 - st.currIdx;
}

} // namespace RDI
} // namespace zro
