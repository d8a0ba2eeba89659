// ORACLE -- test infrastructure only (see zro_math.h header).  PARITY UNPINNED against the reference (no executable
// reference exists for this path); follows the shaders line by line.
//
// zro_rgi.h: CPU restatement of ReSTIR GI (K10), emissive-NEE variant:
//   IndirectLighting/ReSTIR_GI/ReSTIR_GI.hlsl:62-165, Resampling.hlsli:37-612, Reservoir.hlsli:9-131, PathTracing.hlsli:10-99,
//   ReSTIR_GI_NEE.hlsli:8-118,189-272 with ReSTIR_GI/Params.hlsli (MIS on the first hit only, MIS_NON_DIFFUSE_BSDF_SAMPLING 1,
//   NEE_NUM_LIGHT_SAMPLES 1, APPROXIMATE_EMISSIVE_SHADOW_RAY 1, ACCOUNT_FOR_TRANSMITTANCE 0), NEE.hlsli:150-222;
//   host: IndirectLighting.cpp:277-368, 1006-1025.
// Wave intrinsics pinned like K9 / K16: wave = 8x8 pixel group; WaveActiveMax = max over luminance bit patterns,
// WaveActiveSum = 64-lane xor butterfly with absent lanes contributing 0, WaveGetLaneCount() = 64.
// Out-of-range texel reads (negative coordinates of the temporal search) return 0, as D3D does.
#pragma once
#include "zro_rpt.h"
#include "zro_lvg.h"

namespace zro {
namespace RGI {

using RPT::GBufRead; using RPT::GFlags; using RPT::DecodeFlags; using RPT::Roughness; using RPT::DecodeMotion;
using RPT::Camera; using RPT::CurrCamera; using RPT::PrevCamera; using RPT::LensSample;

static const float MAX_PLANE_DIST_REUSE = 0.005f;
static const int NUM_TEMPORAL_SEARCH_ITER = 3;
static const float TEMPORAL_SEARCH_RADIUS = 16.0f;

// Reservoir.hlsli:9-70
struct Reservoir
{
    float3 pos, Lo, normal; float W, w_sum; uint32_t ID; float3 target_z; uint16_t M;
    static Reservoir Init()
    { Reservoir r; r.pos = f3(ZR_FLT_MAX); r.normal = f3(0.0f); r.Lo = f3(0.0f); r.M = 0; r.w_sum = 0; r.W = 0; r.ID = 0xffffffffu; r.target_z = f3(0.0f); return r; }
    bool Update(float weight, float3 vtxPos, float3 vtxNormal, uint32_t vtxID, float3 vtxLo, float3 target, RNG& rng)
    {
        if (zr_isnan(weight)) return false;
        w_sum += weight;
        M += 1;
        if (rng.Uniform() < (weight / zr_max(1e-6f, w_sum))) { pos = vtxPos; normal = vtxNormal; ID = vtxID; Lo = vtxLo; target_z = target; return true; }
        return false;
    }
};

struct State
{
    uint32_t w = 0, h = 0;
    std::vector<float> A[2]; std::vector<uint16_t> B[2]; std::vector<float> C[2];
    bool temporalValid = false; int currIdx = 0;
    void Resize(uint32_t w_, uint32_t h_)
    {
        w = w_; h = h_; size_t n = (size_t)w * h;
        for (int i = 0; i < 2; i++) { A[i].assign(4 * n, 0); B[i].assign(4 * n, 0); C[i].assign(4 * n, 0); }
        temporalValid = false; currIdx = 0;
    }
};

// texel fetch with D3D out-of-bounds semantics (returns the index or SIZE_MAX)
static inline size_t Texel(const GBufRead& gb, int x, int y)
{ return (x < 0 || y < 0 || x >= (int)gb.w || y >= (int)gb.h) ? (size_t)-1 : (size_t)y * gb.w + x; }

static Reservoir PartialRead_Reuse(const State& st, int set, size_t i)
{
    Reservoir r = Reservoir::Init();
    if (i == (size_t)-1) { r.pos = f3(0.0f); r.Lo = f3(0.0f); r.ID = 0; r.M = 0; r.normal = f3(0.0f); return r; }
    const float* a = &st.A[set][4 * i]; const uint16_t* b = &st.B[set][4 * i];
    r.pos = f3(a[0], a[1], a[2]); r.ID = zr_asuint(a[3]);
    r.Lo = f3(zr_f16_to_f32(b[0]), zr_f16_to_f32(b[1]), zr_f16_to_f32(b[2]));
    r.M = (uint16_t)zr_f2u_sat(zr_f16_to_f32(b[3]));
    r.normal = f3(0.0f); r.w_sum = 0; r.W = 0;
    return r;
}
static void PartialRead_Rest(const State& st, int set, size_t i, Reservoir& r)
{
    if (i == (size_t)-1) { r.w_sum = 0; r.W = 0; uint16_t z[2] = {0, 0}; r.normal = Math::DecodeOct32(z); return; }
    const float* c = &st.C[set][4 * i];
    r.w_sum = c[0]; r.W = c[1];
    uint32_t n = zr_asuint(c[2]);
    uint16_t ns[2] = {(uint16_t)(n & 0xffff), (uint16_t)(n >> 16)};
    r.normal = Math::DecodeOct32(ns);
}
static void WriteReservoir(State& st, int set, size_t i, const Reservoir& r, float M_max)
{
    uint16_t n[2]; Math::EncodeOct32(r.normal, n);
    uint32_t nu = n[0] | ((uint32_t)n[1] << 16);
    float M_clamped = zr_min((float)r.M, zr_round_f16(M_max));
    float* a = &st.A[set][4 * i]; uint16_t* b = &st.B[set][4 * i]; float* c = &st.C[set][4 * i];
    a[0] = r.pos.x; a[1] = r.pos.y; a[2] = r.pos.z; a[3] = zr_asfloat(r.ID);
    b[0] = zr_f32_to_f16(r.Lo.x); b[1] = zr_f32_to_f16(r.Lo.y); b[2] = zr_f32_to_f16(r.Lo.z); b[3] = zr_f32_to_f16(M_clamped);
    c[0] = r.w_sum; c[1] = r.W; c[2] = zr_asfloat(nu);
}

// RGI_Util::NEE_Emissive_LVG, ReSTIR_GI_NEE.hlsli:121-187 (numSamples = 1; globals.extents / offset_y are fp16, ReSTIR_GI.hlsl:51-53)
static float3 NEE_Emissive_LVG(const Scene& sc, const zr_frame_constants& g, float3 pos, float3 normal, BSDF::ShadingData surface,
    uint32_t sampleSetIdx, RNG& rng)
{
    float3 ret = f3(0.0f);
    const float3 extents = f3(zr_round_f16(sc.lvgExtents[0]), zr_round_f16(sc.lvgExtents[1]), zr_round_f16(sc.lvgExtents[2]));
    const float offset_y = zr_round_f16(sc.lvgOffsetY);
    zr_voxel_sample s;
    float3 lpos, lnormal, le; float lightPdf; uint32_t lightID;
    if (LVG::Sample(pos, sc, 64, g.curr_view, s, rng, extents, offset_y))
    {
        lpos = f3(s.pos); lnormal = Math::DecodeOct32(s.normal);
        le = f3(zr_f16_to_f32(s.le[0]), zr_f16_to_f32(s.le[1]), zr_f16_to_f32(s.le[2]));
        lightPdf = s.pdf; lightID = s.id;
        if (s.two_sided && dot(lnormal, pos - lpos) < 0) lnormal = lnormal * -1.0f;
    }
    else
    {
        Light::PresampledLight pl = Light::SamplePresampledSet(sc, sampleSetIdx, pos, rng);
        lpos = pl.pos; lnormal = pl.normal; le = pl.le; lightPdf = pl.pdf; lightID = pl.ID;
    }
    const float t = length(lpos - pos);
    const float3 wi = (lpos - pos) / t;
    if (lightID != 0xffffffffu && dot(lnormal, -wi) > 0)
    {
        const float dwdA = zr_saturate(dot(lnormal, -wi)) / (t * t);
        surface.SetWi(wi, normal);
        le = le * (BSDF::Unified(surface).f * dwdA);
        if (Math::Luminance(le) > 1e-6f)
            le *= RtRayQuery::Visibility_Segment(sc, true, pos, wi, t, normal, lightID, surface.Transmissive()) ? 1.0f : 0.0f;
        ret += le / zr_max(lightPdf, 1e-6f);
    }
    return ret;       // ld /= numSamples (1)
}

// NEE.hlsli:150-222 (NumSamples = 1)
static float3 NEE_Emissive_Power(const Scene& sc, float3 pos, float3 normal, BSDF::ShadingData surface, uint32_t numEmissives, bool presampled,
    uint32_t sampleSetIdx, RNG& rng)
{
    float3 ret = f3(0.0f);
    Light::EmissiveTriSample lightSample; float3 le; float lightPdf; uint32_t lightID;
    if (presampled)
    {
        Light::PresampledLight pl = Light::SamplePresampledSet(sc, sampleSetIdx, pos, rng);
        lightSample.pos = pl.pos; lightSample.normal = pl.normal; le = pl.le; lightPdf = pl.pdf; lightID = pl.ID;
    }
    else
    {
        Light::AliasTableSample entry = Light::AliasTableSample::get(sc, numEmissives, rng);
        EmTri tri; tri.t = sc.emissives[entry.idx];
        lightSample = Light::EmissiveTriSample::get(pos, tri, rng);
        le = Light::Le_EmissiveTriangle(sc, tri, lightSample.bary);
        lightPdf = entry.pdf * lightSample.pdf;
        lightID = tri.t.id;
    }
    const float t = length(lightSample.pos - pos);
    const float3 wi = (lightSample.pos - pos) / t;
    if (dot(lightSample.normal, -wi) > 0)
    {
        const float dwdA = zr_saturate(dot(lightSample.normal, -wi)) / (t * t);
        surface.SetWi(wi, normal);
        float3 ld = le * BSDF::Unified(surface).f * dwdA;
        if (Math::Luminance(ld) > 1e-6f)
            ld *= RtRayQuery::Visibility_Segment(sc, true, pos, wi, t, normal, lightID, surface.Transmissive()) ? 1.0f : 0.0f;
        ret += ld / lightPdf;
    }
    ret = ret / 1.0f;
    return ret;
}

struct Lane
{
    bool inFrame = false, valid = false, active = false, atRR = false, hasSample = false;
    uint32_t x = 0, y = 0; size_t px = 0;
    // primary
    float3 origin, pos, normal; float roughness, ior, z_view; BSDF::ShadingData surface; float2 lensSample; bool transmissive;
    RNG rngThread, rngGroup; int maxNumBounces; uint32_t sampleSetIdx;
    BSDF::BSDFSample firstSample; float3 hitPos, hitNormal; uint32_t hitID;
    // path state
    float3 li, throughput, ppos, pnormal; float eta_curr, eta_next; int bounce; bool inMedium;
    BSDF::BSDFSample bsdfSample; RtRayQuery::Hit hitInfo; RT::RayDifferentials rd; BSDF::ShadingData psurface; float3 dpdx, dpdy;
    Reservoir r;
};

// Jacobian of the reconnection shift, Resampling.hlsli:283-303
static float JacobianReconnectionShift(float3 x2_normal, float3 x1_r, float3 x1_q, float3 x2_q)
{
    float3 v_r = x1_r - x2_q;
    const float t_r2 = dot(v_r, v_r);
    v_r = dot(v_r, v_r) == 0 ? v_r : v_r / zr_max(zr_sqrt(t_r2), 1e-6f);
    float3 v_q = x1_q - x2_q;
    const float t_q2 = dot(v_q, v_q);
    v_q = dot(v_q, v_q) == 0 ? v_q : v_q / zr_max(zr_sqrt(t_q2), 1e-6f);
    float cosPhi_r = dot(v_r, x2_normal);
    float cosPhi_q = dot(v_q, x2_normal);
    return (zr_abs(cosPhi_r) * t_q2) / zr_max(zr_abs(cosPhi_q) * t_r2, 1e-6f);
}

struct TemporalSampleData { float3 posW, normal; float roughness; int px, py; bool metallic, transmissive; float eta_next; };

// Resampling.hlsli:129-231
static void FindTemporalCandidate(const zr_frame_constants& g, const GBufRead& gbPrev, uint32_t DTx, uint32_t DTy, float3 posW, float3 normal, float viewZ,
    float roughness, bool transmissive, float2 prevUV, RNG& rng, TemporalSampleData data[2], bool valid[2])
{
    valid[0] = valid[1] = false;
    if (prevUV.x < 0 || prevUV.y < 0 || prevUV.x > 1 || prevUV.y > 1) return;
    const float2 renderDim = {(float)g.render_width, (float)g.render_height};
    const int ppx = (int)(prevUV.x * renderDim.x), ppy = (int)(prevUV.y * renderDim.y);
    int curr = 0;
    const Camera pcam = PrevCamera(g);
    for (int i = 0; i < NUM_TEMPORAL_SEARCH_ITER; i++)
    {
        const float theta = rng.Uniform() * ZR_TWO_PI;
        float sinTheta, cosTheta; zr_sincos(theta, &sinTheta, &cosTheta);
        const float2 offset = {TEMPORAL_SEARCH_RADIUS * sinTheta, TEMPORAL_SEARCH_RADIUS * cosTheta};
        const float k = i > 0 ? 1.0f : 0.0f;
        const int sx = zr_f2i_sat((float)ppx + k * offset.x), sy = zr_f2i_sat((float)ppy + k * offset.y);
        if ((float)sx >= renderDim.x || (float)sy >= renderDim.y) continue;
        if (i > 0 && sx == (int)DTx && sy == (int)DTy) continue;
        const size_t sp = Texel(gbPrev, sx, sy);
        const uint16_t mrp = sp == (size_t)-1 ? (uint16_t)0 : gbPrev.mr[sp];
        GFlags pf = DecodeFlags(mrp);
        if (pf.emissive) continue;
        float viewZ_prev = sp == (size_t)-1 ? 0.0f : gbPrev.depth[sp];
        float2 lens = {0, 0};
        if (pcam.dof)
        {
            uint32_t hx = (uint32_t)sx, hy = (uint32_t)sy, hz = (uint32_t)sx; zr_pcg3d(&hx, &hy, &hz);
            RNG rr = RNG::Init(hz, hy, g.frame_num - 1);
            lens = Sampling::UniformSampleDiskConcentric(rr.Uniform2D());
            lens = lens * pcam.lensRadius;
        }
        float3 origin = pcam.origin;
        float3 prevPos = Math::WorldPosFromScreenSpace2(f2((float)sx, (float)sy), renderDim, viewZ_prev, pcam.tanHalfFOV, pcam.aspect, pcam.jitter,
            pcam.vbx, pcam.vby, pcam.vbz, pcam.dof, lens, pcam.focusDepth, origin);
        float tolerance = MAX_PLANE_DIST_REUSE * (g.dof ? 10.0f : 1.0f);
        if (!(zr_abs(dot(normal, prevPos - posW)) <= tolerance * viewZ)) continue;
        const uint32_t np = sp == (size_t)-1 ? 0u : gbPrev.normal[sp];
        const float3 prevNormal = Math::DecodeUnitVector(f2((float)(np & 0xffff) / 65535.0f, (float)(np >> 16) / 65535.0f));
        const float prevRough = Roughness(mrp);
        valid[curr] = dot(prevNormal, normal) > 0.1f;
        if (roughness < 0.5f) valid[curr] = valid[curr] && (zr_abs(prevRough - roughness) < 0.15f);
        float prevEta_mat = DEFAULT_ETA_MAT;
        if (pf.transmissive) prevEta_mat = zr_fma((float)(sp == (size_t)-1 ? 0 : gbPrev.ior[sp]) / 255.0f, MAX_IOR - MIN_IOR, MIN_IOR);
        valid[curr] = valid[curr] && (pf.transmissive == transmissive);
        valid[curr] = g.dof ? true : valid[curr];
        if (valid[curr])
        {
            TemporalSampleData& d = data[curr];
            d.px = sx; d.py = sy; d.posW = prevPos; d.normal = prevNormal; d.metallic = pf.metallic; d.roughness = prevRough;
            d.transmissive = pf.transmissive; d.eta_next = prevEta_mat;
            curr++;
            if (curr == 2) break;
        }
    }
}

// Resampling.hlsli:233-281
static float TargetLumAtTemporalPixel(const Scene& sc, const zr_frame_constants& g, const GBufRead& gbPrev, const Reservoir& r_curr, const TemporalSampleData& c,
    bool testVisibility = true)
{
    float3 wi = r_curr.pos - c.posW;
    if (dot(wi, wi) == 0) return 0;
    float t = length(wi);
    wi = wi / zr_max(t, 1e-6f);
    const size_t sp = Texel(gbPrev, c.px, c.py);
    const float3 baseColor_prev = Math::UnpackRGB8(sp == (size_t)-1 ? 0u : gbPrev.baseColor[sp]);
    const Camera pcam = PrevCamera(g);
    float3 camPos_prev = pcam.origin;
    if (pcam.dof)
    {
        uint32_t hx = (uint32_t)c.px, hy = (uint32_t)c.py, hz = (uint32_t)c.px; zr_pcg3d(&hx, &hy, &hz);
        RNG rr = RNG::Init(hz, hy, g.frame_num - 1);
        float2 lens = Sampling::UniformSampleDiskConcentric(rr.Uniform2D());
        lens = lens * pcam.lensRadius;
        camPos_prev += mad3(lens.x, pcam.vbx, lens.y * pcam.vby);
    }
    const float3 wo_prev = normalize(camPos_prev - c.posW);
    BSDF::ShadingData surface_prev = BSDF::ShadingData::Init(c.normal, wo_prev, c.metallic, c.roughness, baseColor_prev, ETA_AIR, c.eta_next, c.transmissive);
    surface_prev.SetWi(wi, c.normal);
    const float3 target_prev = r_curr.Lo * BSDF::Unified(surface_prev).f;
    const float targetLum_prev = Math::Luminance(target_prev);
    if (testVisibility && targetLum_prev > 1e-5f)
        if (!RtRayQuery::Visibility_Segment(sc, true, c.posW, wi, t, c.normal, r_curr.ID, surface_prev.Transmissive())) return 0;
    return targetLum_prev;
}

// Resampling.hlsli:305-369
static void TemporalResample1(const Scene& sc, const zr_frame_constants& g, const GBufRead& gbPrev, const State& st, int prevSet, float3 posW, float3 normal,
    BSDF::ShadingData surface, const TemporalSampleData& c, Reservoir& r, RNG& rng)
{
    const size_t sp = Texel(gbPrev, c.px, c.py);
    Reservoir r_prev = PartialRead_Reuse(st, prevSet, sp);
    const uint16_t M_new = (uint16_t)(r.M + r_prev.M);
    if (r.w_sum != 0)
    {
        float targetLum_prev = 0.0f;
        if (r_prev.M > 0 && Math::Luminance(r.Lo) > 1e-6f) targetLum_prev = TargetLumAtTemporalPixel(sc, g, gbPrev, r, c);
        const float p_curr = Math::Luminance(r.target_z);
        const float J = JacobianReconnectionShift(r.normal, c.posW, posW, r.pos);
        const float m_curr = p_curr / zr_max(p_curr + (float)r_prev.M * targetLum_prev * J, 1e-6f);
        r.w_sum *= m_curr;
    }
    if (r_prev.ID == 0xffffffffu || dot(r_prev.Lo, f3(1.0f)) == 0)
    {
        float targetLum = Math::Luminance(r.target_z);
        r.W = targetLum > 0.0f ? r.w_sum / targetLum : 0.0f;
        r.M = M_new;
        return;
    }
    float3 wi = r_prev.pos - posW;
    float t = length(wi);
    wi = wi / t;
    surface.SetWi(wi, normal);
    const float3 target_curr = r_prev.Lo * BSDF::Unified(surface).f;
    const float targetLum_curr = Math::Luminance(target_curr);
    if (targetLum_curr > 1e-6f)
    {
        if (RtRayQuery::Visibility_Segment(sc, true, posW, wi, t, normal, r_prev.ID, surface.Transmissive()))
        {
            PartialRead_Rest(st, prevSet, sp, r_prev);
            const float targetLum_prev = r_prev.W > 0 ? r_prev.w_sum / r_prev.W : 0;
            const float J = JacobianReconnectionShift(r_prev.normal, posW, c.posW, r_prev.pos);
            const float numerator = (float)r_prev.M * targetLum_prev;
            const float denom = numerator / zr_max(J, 1e-6f) + targetLum_curr;
            const float m_prev = numerator / zr_max(denom, 1e-6f);
            const float w_prev = m_prev * targetLum_curr * r_prev.W;
            r.Update(w_prev, r_prev.pos, r_prev.normal, r_prev.ID, r_prev.Lo, target_curr, rng);
        }
    }
    float targetLum = Math::Luminance(r.target_z);
    r.W = targetLum > 0.0f ? r.w_sum / targetLum : 0.0f;
    r.M = M_new;
}

// Resampling.hlsli:371-454
static void TemporalResample2(const Scene& sc, const zr_frame_constants& g, const GBufRead& gbPrev, const State& st, int prevSet, float3 posW, float3 normal,
    BSDF::ShadingData surface, const TemporalSampleData c[2], Reservoir& r, RNG& rng)
{
    uint16_t M_new = r.M;
    Reservoir r_prev[2]; size_t sp[2];
    for (int k = 0; k < 2; k++)
    {
        sp[k] = Texel(gbPrev, c[k].px, c[k].py);
        r_prev[k] = PartialRead_Reuse(st, prevSet, sp[k]);
        M_new = (uint16_t)(M_new + r_prev[k].M);
    }
    {
        const float p_curr = Math::Luminance(r.target_z);
        float denom = p_curr;
        if (Math::Luminance(r.Lo) > 1e-5f)
        {
            for (int p = 0; p < 2; p++)
            {
                if (r_prev[p].M == 0) continue;
                float targetLum_prev = TargetLumAtTemporalPixel(sc, g, gbPrev, r, c[p], p != 0);
                float J = JacobianReconnectionShift(r.normal, c[p].posW, posW, r.pos);
                denom += (float)r_prev[p].M * J * targetLum_prev;
            }
        }
        const float m_curr = denom == 0 ? 0 : p_curr / denom;
        r.w_sum *= m_curr;
    }
    for (int i = 0; i < 2; i++)
    {
        float3 wi = r_prev[i].pos - posW;
        float t = (wi.x == 0 && wi.y == 0 && wi.z == 0) ? 0 : length(wi);
        wi = wi / zr_max(t, 1e-6f);
        surface.SetWi(wi, normal);
        const float3 target_curr = r_prev[i].Lo * BSDF::Unified(surface).f;
        const float targetLum_curr = Math::Luminance(target_curr);
        if (targetLum_curr < 1e-5f) continue;
        if (RtRayQuery::Visibility_Segment(sc, true, posW, wi, t, normal, r_prev[i].ID, surface.Transmissive()))
        {
            PartialRead_Rest(st, prevSet, sp[i], r_prev[i]);
            const float targetLum_prev = r_prev[i].W > 0 ? r_prev[i].w_sum / r_prev[i].W : 0;
            const float J = JacobianReconnectionShift(r_prev[i].normal, posW, c[i].posW, r_prev[i].pos);
            const float numerator = (float)r_prev[i].M * targetLum_prev;
            float denom = (numerator / J) + targetLum_curr;
            if (r_prev[1 - i].M > 0 && targetLum_prev > 0)
            {
                const float J_tt = JacobianReconnectionShift(r_prev[i].normal, c[1 - i].posW, c[i].posW, r_prev[i].pos);
                const float targetLum_other = TargetLumAtTemporalPixel(sc, g, gbPrev, r_prev[i], c[1 - i]);
                denom += (float)r_prev[1 - i].M * targetLum_other / zr_max(J_tt, 1e-6f);
            }
            denom = J == 0 ? 0 : denom;
            const float m_prev = denom == 0 ? 0 : numerator / denom;
            const float w_prev = m_prev * targetLum_curr * r_prev[i].W;
            r.Update(w_prev, r_prev[i].pos, r_prev[i].normal, r_prev[i].ID, r_prev[i].Lo, target_curr, rng);
        }
    }
    float targetLum = Math::Luminance(r.target_z);
    r.W = targetLum > 0.0f ? r.w_sum / targetLum : 0.0f;
    r.M = M_new;
}

// IndirectLighting::RenderReSTIR_GI + Render tail
static void Render(const Scene& sc, const zr_frame_constants& g, const zr_gbuffer_planes* gbCurr, const zr_gbuffer_planes* gbPrevPlanes, const zr_params& prm,
    State& st, float* finalRGBA)
{
    BSDF::g_rho = &sc.rhoLUT;
    GBufRead gb(gbCurr);
    const uint32_t W = g.render_width, H = g.render_height;
    const bool doTemporal = (prm.flags & ZR_IND_TEMPORAL_RESAMPLE) && st.temporalValid && gbPrevPlanes;
    const bool writeReservoirs = doTemporal || !st.temporalValid;
    const bool rr = prm.flags & ZR_IND_RUSSIAN_ROULETTE;
    const bool accumulate = g.accumulate && g.camera_static;
    const bool presampled = prm.presampling != 0;
    const uint32_t numSampleSets = presampled ? prm.num_sample_sets : 0;
    const Camera cam = CurrCamera(g);
    const int curSet = st.currIdx, prevSet = 1 - st.currIdx;
    std::vector<Lane> L(64);
    for (uint32_t gy = 0; gy < (H + 7) / 8; gy++) for (uint32_t gx = 0; gx < (W + 7) / 8; gx++)
    {
        for (uint32_t l = 0; l < 64; l++)
        {
            Lane& P = L[l]; P = Lane();
            const uint32_t x = gx * 8 + (l & 7), y = gy * 8 + (l >> 3);
            P.x = x; P.y = y;
            if (x >= W || y >= H) continue;
            P.inFrame = true; P.px = (size_t)y * W + x;
            GFlags flags = DecodeFlags(gb.mr[P.px]);
            if (flags.invalid || flags.emissive)
            {
                if (!accumulate) { float* o = finalRGBA + 4 * P.px; o[0] = o[1] = o[2] = 0; }
                continue;
            }
            P.valid = true;
            P.z_view = gb.depth[P.px];
            P.lensSample = LensSample(cam, x, y, g.frame_num);
            P.origin = cam.origin;
            P.pos = Math::WorldPosFromScreenSpace2(f2((float)x, (float)y), cam.renderDim, P.z_view, cam.tanHalfFOV, cam.aspect, cam.jitter, cam.vbx, cam.vby, cam.vbz,
                cam.dof, P.lensSample, cam.focusDepth, P.origin);
            const uint32_t np = gb.normal[P.px];
            P.normal = Math::DecodeUnitVector(f2((float)(np & 0xffff) / 65535.0f, (float)(np >> 16) / 65535.0f));
            const float3 baseColor = Math::UnpackRGB8(gb.baseColor[P.px]);
            P.roughness = Roughness(gb.mr[P.px]);
            P.ior = DEFAULT_ETA_MAT;
            if (flags.transmissive) P.ior = zr_fma((float)gb.ior[P.px] / 255.0f, MAX_IOR - MIN_IOR, MIN_IOR);
            const float3 wo = normalize(P.origin - P.pos);
            P.surface = BSDF::ShadingData::Init(P.normal, wo, flags.metallic, P.roughness, baseColor, ETA_AIR, P.ior, flags.transmissive);
            P.transmissive = flags.transmissive;
            P.rngGroup = RNG::Init(gx ^ 61u, gy ^ 61u, g.frame_num);
            P.rngThread = RNG::Init(x ^ 511u, y ^ 31u, g.frame_num);
            P.maxNumBounces = flags.transmissive ? (int)prm.max_glossy_tr_bounces : (int)prm.max_non_tr_bounces;
            // EstimateIndirectLighting (Resampling.hlsli:529-612)
            if ((prm.flags & ZR_IND_STOCHASTIC_MULTI_BOUNCE) && (P.roughness >= 0.1f || g.camera_static))
                P.maxNumBounces = P.rngGroup.Uniform() < 0.5f ? 1 : P.maxNumBounces;
            P.sampleSetIdx = P.rngGroup.UniformUintBounded_Faster(numSampleSets);
            // RIS_InitialCandidates (:37-117) up to the PathTrace call
            P.r = Reservoir::Init();
            P.firstSample = BSDF::SampleBSDF(P.normal, P.surface, P.rngThread);
            if (P.firstSample.pdf == 0) continue;
            Math::TriDifferentials triDiffs = RPT::LoadTriDiffs(gb, P.px);
            RT::RayDifferentials rd = RPT::InitRD(cam, (int)x, (int)y, P.lensSample, P.origin);
            float3 dpdx, dpdy;
            rd.dpdx_dpdy(P.pos, P.normal, dpdx, dpdy);
            rd.ComputeUVDifferentials(dpdx, dpdy, triDiffs.dpdu, triDiffs.dpdv);
            rd.UpdateRays(P.pos, P.normal, P.firstSample.wi, P.surface.wo, triDiffs, dpdx, dpdy, dot(P.firstSample.wi, P.normal) < 0, P.surface.eta);
            RtRayQuery::Hit hitInfo = RtRayQuery::FindClosest(sc, true, true, P.pos, P.normal, P.firstSample.wi, P.surface.Transmissive());
            if (!hitInfo.hit) continue;
            P.hasSample = true;
            P.hitPos = P.pos + hitInfo.t * P.firstSample.wi;
            P.hitNormal = hitInfo.normal; P.hitID = hitInfo.ID;
            // ReSTIR_RT::PathTrace prologue (PathTracing.hlsli:16-21)
            P.active = true; P.li = f3(0.0f); P.throughput = f3(1.0f);
            P.ppos = P.pos; P.pnormal = P.normal;
            P.eta_curr = dot(P.normal, P.firstSample.wi) < 0 ? P.ior : ETA_AIR;
            P.bounce = 0; P.inMedium = dot(P.normal, P.firstSample.wi) < 0;
            P.bsdfSample = P.firstSample; P.hitInfo = hitInfo; P.rd = rd;
        }
        // ---- PathTrace loop in lockstep (PathTracing.hlsli:23-96), GI parameters
        for (;;)
        {
            bool any = false;
            for (uint32_t l = 0; l < 64; l++)
            {
                Lane& P = L[l];
                P.atRR = false;
                if (!P.active) continue;
                any = true;
                float3 hitPos = mad3(P.hitInfo.t, P.bsdfSample.wi, P.ppos);
                P.rd.dpdx_dpdy(hitPos, P.hitInfo.normal, P.dpdx, P.dpdy);
                P.rd.ComputeUVDifferentials(P.dpdx, P.dpdy, P.hitInfo.triDiffs.dpdu, P.hitInfo.triDiffs.dpdv);
                if (!RtRayQuery::GetMaterialData(sc, -P.bsdfSample.wi, P.eta_curr, P.rd.uv_grads, P.hitInfo, P.psurface, P.eta_next)) { P.active = false; continue; }
                // RGI_Util::NEE (NEE_EMISSIVE == 1, USE_MIS == 1, MIS_ALL_BOUNCES == 0)
                float3 ld;
                if (g.num_emissive_triangles == 0)      // NEE_EMISSIVE == 0: sun + sky (ReSTIR_GI_NEE.hlsli:194-226)
                    ld = NEE_SunSky(sc, g, hitPos, P.hitInfo.normal, P.psurface, P.rngThread);
                else if (P.bounce == 0)
                    ld = NEE_Emissive_MIS(sc, 1, true, hitPos, P.hitInfo.normal, P.psurface, g.num_emissive_triangles, P.rngThread, presampled, P.sampleSetIdx, true);
                else if (prm.use_lvg && presampled)      // USE_LVG && USE_PRESAMPLED_SETS (ReSTIR_GI_NEE.hlsli:240-243)
                    ld = NEE_Emissive_LVG(sc, g, hitPos, P.hitInfo.normal, P.psurface, P.sampleSetIdx, P.rngThread);
                else
                    ld = NEE_Emissive_Power(sc, hitPos, P.hitInfo.normal, P.psurface, g.num_emissive_triangles, presampled, P.sampleSetIdx, P.rngThread);
                P.li += P.throughput * ld;
                if (P.bounce >= (P.maxNumBounces - 1)) { P.active = false; continue; }
                P.ppos = hitPos; P.pnormal = P.hitInfo.normal;
                P.bounce++;
                P.atRR = rr && (P.bounce >= 3);
            }
            if (!any) break;
            uint32_t bits = 0;
            for (uint32_t l = 0; l < 64; l++)
                if (L[l].active && L[l].atRR)
                {
                    float lum = Math::Luminance(L[l].throughput);
                    uint32_t b = (zr_isnan(lum) || lum < 0) ? 0u : zr_asuint(lum);
                    bits = b > bits ? b : bits;
                }
            const float waveThroughput = zr_asfloat(bits);
            for (uint32_t l = 0; l < 64; l++)
            {
                Lane& P = L[l];
                if (!P.active) continue;
                if (P.atRR)
                {
                    float p_terminate = zr_max(0.05f, 1 - waveThroughput);
                    if (P.rngGroup.Uniform() < p_terminate) { P.active = false; continue; }
                    P.throughput /= (1 - p_terminate);
                }
                P.bsdfSample = BSDF::BSDFSample::Init();
                if (P.bounce < P.maxNumBounces) P.bsdfSample = BSDF::SampleBSDF(P.pnormal, P.psurface, P.rngThread);
                if (Math::Luminance(P.bsdfSample.bsdfOverPdf) == 0) { P.active = false; continue; }
                P.hitInfo = RtRayQuery::FindClosest(sc, false, true, P.ppos, P.pnormal, P.bsdfSample.wi, P.psurface.Transmissive());
                if (!P.hitInfo.hit) { P.active = false; continue; }
                P.throughput *= P.bsdfSample.bsdfOverPdf;
                bool transmitted = dot(P.pnormal, P.bsdfSample.wi) < 0;
                P.eta_curr = transmitted ? (P.eta_curr == ETA_AIR ? P.eta_next : ETA_AIR) : P.eta_curr;
                P.inMedium = transmitted ? !P.inMedium : P.inMedium;
                P.rd.UpdateRays(P.ppos, P.pnormal, P.bsdfSample.wi, P.psurface.wo, P.hitInfo.triDiffs, P.dpdx, P.dpdy, transmitted, P.psurface.eta);
            }
        }
        // ---- finish RIS_InitialCandidates, temporal resampling, boiling suppression, outputs
        float wsum[64];
        for (uint32_t l = 0; l < 64; l++)
        {
            Lane& P = L[l]; wsum[l] = 0.0f;
            if (!P.valid) continue;
            if (P.hasSample)
            {
                float3 lo = P.li;
                float3 target = lo;
                if (dot(lo, lo) > 0) { P.surface.SetWi(P.firstSample.wi, P.normal); target *= BSDF::Unified(P.surface).f; }
                float targetLum = Math::Luminance(target);
                float w = targetLum / zr_max(P.firstSample.pdf, 1e-6f);
                P.r.Update(w, P.hitPos, P.hitNormal, P.hitID, lo, target, P.rngThread);
                P.r.W = targetLum > 0 ? 1.0f / P.firstSample.pdf : 0.0f;
            }
            if (doTemporal)
            {
                GBufRead gbPrev(gbPrevPlanes);
                const float2 motionVec = DecodeMotion(gb.motion[P.px]);
                const float2 currUV = {((float)P.x + 0.5f) / cam.renderDim.x, ((float)P.y + 0.5f) / cam.renderDim.y};
                const float2 prevUV = currUV - motionVec;
                TemporalSampleData data[2]; bool valid[2];
                FindTemporalCandidate(g, gbPrev, P.x, P.y, P.pos, P.normal, P.z_view, P.roughness, P.surface.specTr, prevUV, P.rngThread, data, valid);
                if (valid[1] && P.roughness > 0.05f) TemporalResample2(sc, g, gbPrev, st, prevSet, P.pos, P.normal, P.surface, data, P.r, P.rngThread);
                else if (valid[0]) TemporalResample1(sc, g, gbPrev, st, prevSet, P.pos, P.normal, P.surface, data[0], P.r, P.rngThread);
                wsum[l] = P.r.w_sum;
            }
        }
        if (doTemporal && (prm.flags & ZR_IND_BOILING_SUPPRESSION))
        {
            const float waveSum = RPT::WaveSum64(wsum);
            for (uint32_t l = 0; l < 64; l++)
            {
                Lane& P = L[l];
                if (!P.valid) continue;
                float waveAvg = (waveSum - P.r.w_sum) / 63.0f;
                if (P.r.w_sum > 25 * waveAvg) P.r.M = 1;
            }
        }
        for (uint32_t l = 0; l < 64; l++)
        {
            Lane& P = L[l];
            if (!P.valid) continue;
            if (writeReservoirs) WriteReservoir(st, curSet, P.px, P.r, (float)prm.m_max_temporal);
            float3 li = P.r.target_z * P.r.W;
            li = any_nan(li) ? f3(0.0f) : li;
            float* o = finalRGBA + 4 * P.px;
            if (accumulate) { o[0] += li.x; o[1] += li.y; o[2] += li.z; }
            else { o[0] = li.x; o[1] = li.y; o[2] = li.z; }
        }
    }
    st.temporalValid = true;
    st.currIdx = 1 - st.currIdx;
}

} // namespace RGI
} // namespace zro
