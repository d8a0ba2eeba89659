// zr_rgi.h -- per-lane stage functions of ReSTIR GI (K10), emissive-NEE variant.
//
// Reference (Source/ZetaRenderPass/IndirectLighting/ReSTIR_GI/): ReSTIR_GI.hlsl:62-165, Resampling.hlsli:37-612,
// Reservoir.hlsli:9-131, PathTracing.hlsli:10-99, ReSTIR_GI_NEE.hlsli:8-118,189-272 with ReSTIR_GI/Params.hlsli (MIS on the
// first hit only, MIS_NON_DIFFUSE_BSDF_SAMPLING 1, NEE_NUM_LIGHT_SAMPLES 1, APPROXIMATE_EMISSIVE_SHADOW_RAY 1,
// ACCOUNT_FOR_TRANSMITTANCE 0), ../NEE.hlsli:150-222; host IndirectLighting.cpp:277-368.
// Persistent state in the reference's formats: two reservoir sets x (A RGBA32F pos + ID bits, B RGBA16F Lo + M,
// C RGBA32F w_sum, W, oct32 normal bits) = 40 B/px.
// One kernel, one 8x8 pixel group per wave64 walking the bounce loop in lockstep (like K11): WaveActiveMax = integer max of
// luminance bit patterns, WaveActiveSum = 64-lane xor butterfly (absent lanes 0), WaveGetLaneCount() = 64.
// Out-of-range texel reads of the temporal search return 0 (D3D rule).  Ray differentials are not carried (no textures).
#pragma once
#include "zr_rpt.h"
#include "zr_lvg.h"

namespace zr {
namespace rgi {

using rpt::Pix; using rpt::GFlags; using rpt::DecodeFlags; using rpt::RoughnessOf; using rpt::DecodeMotion; using rpt::Camera;
using rpt::CurrCamera; using rpt::PrevCamera; using rpt::Globals; using rpt::VisibilitySegmentApprox; using rpt::IsSpecular; using rpt::HitEm;

static constexpr float kMaxPlaneDist = 0.005f, kTemporalRadius = 16.0f;

struct GiPlanes { F4* A; uint16_t* B; F4* C; };

// Reservoir.hlsli:9-70
struct Reservoir
{
    V3 pos, Lo, normal; float W, w_sum; uint32_t ID; V3 target_z; uint32_t M;
    ZR_HDM bool Update(float weight, V3 vtxPos, V3 vtxNormal, uint32_t vtxID, V3 vtxLo, V3 target, Rng& rng)
    {
        if (zr_isnan(weight)) return false;
        w_sum += weight;
        M += 1;
        if (rng.Uniform() < (weight / zr_max(1e-6f, w_sum))) { pos = vtxPos; normal = vtxNormal; ID = vtxID; Lo = vtxLo; target_z = target; return true; }
        return false;
    }
};
ZR_HD Reservoir InitReservoir()
{ Reservoir r; r.pos = v3(ZR_FLT_MAX); r.normal = v3(0.0f); r.Lo = v3(0.0f); r.M = 0; r.w_sum = 0; r.W = 0; r.ID = 0xffffffffu; r.target_z = v3(0.0f); return r; }

// textured: the scene has a texture heap -> the lanes carry ray differentials (RT.hlsli:309-479) for the texture LODs.  The kernel
// is instantiated per value and overwrites the field with its template constant, so untextured frames compile all of it away.
struct GiParams { uint32_t flags, maxNonTrBounces, maxGlossyTrBounces, numSampleSets, accumulate, doTemporal, writeReservoirs, useLVG; float M_max; uint32_t textured; };
struct GiFrame
{
    SceneView sc; GBuf gb, gbPrev; GiPlanes cur, prev; float* finalRGBA; GiParams prm;
    uint32_t ox0, oy0, ow, oh;
    ZR_HDM bool Owns(uint32_t x, uint32_t y) const { return x >= ox0 && y >= oy0 && x < ox0 + ow && y < oy0 + oh; }
};

// texel index with D3D out-of-bounds semantics: (size_t)-1 = reads return 0
ZR_HD size_t Texel(const GBuf& gb, int x, int y, uint32_t W, uint32_t H)
{
    if (x < 0 || y < 0 || x >= (int)W || y >= (int)H) return (size_t)-1;
    if (!rpt::InPlanes(gb, x, y)) return (size_t)-1;     // screen-tile split: beyond the apron behaves like the frame border
    return Pix(gb, (uint32_t)x, (uint32_t)y);
}
ZR_HD Reservoir PartialRead_Reuse(const GiPlanes& p, size_t i)
{
    Reservoir r = InitReservoir();
    if (i == (size_t)-1) { r.pos = v3(0.0f); r.ID = 0; return r; }
    const F4 a = p.A[i]; const uint16_t* b = &p.B[4 * i];
    r.pos = xyz(a); r.ID = zr_asuint(a.w);
    r.Lo = v3(zr_f16_to_f32(b[0]), zr_f16_to_f32(b[1]), zr_f16_to_f32(b[2]));
    r.M = zr_f2u_sat(zr_f16_to_f32(b[3])) & 0xffffu;
    return r;
}
ZR_HD void PartialRead_Rest(const GiPlanes& p, size_t i, Reservoir& r)
{
    if (i == (size_t)-1) { r.w_sum = 0; r.W = 0; r.normal = DecodeOct32u(0u); return; }
    const F4 c = p.C[i];
    r.w_sum = c.x; r.W = c.y;
    r.normal = DecodeOct32u(zr_asuint(c.z));
}
ZR_HD void WriteReservoir(const GiPlanes& p, size_t i, const Reservoir& r, float M_max)
{
    V2 e = EncodeUnitVector(r.normal);
    uint32_t nu = FloatToUNorm16(e.x) | (FloatToUNorm16(e.y) << 16);
    float M_clamped = zr_min((float)r.M, zr_round_f16(M_max));
    p.A[i] = f4(r.pos, zr_asfloat(r.ID));
    uint16_t* b = &p.B[4 * i];
    b[0] = zr_f32_to_f16(r.Lo.x); b[1] = zr_f32_to_f16(r.Lo.y); b[2] = zr_f32_to_f16(r.Lo.z); b[3] = zr_f32_to_f16(M_clamped);
    F4 c = p.C[i]; c.x = r.w_sum; c.y = r.W; c.z = zr_asfloat(nu);
    p.C[i] = c;
}

// light sample of the NEE routines (alias table or presampled set)
struct LightDraw { V3 pos, normal, le; float pdf; uint32_t ID; };
ZR_HD LightDraw DrawLight(const Globals& g, V3 shadingPos, Rng& rng)
{
    const SceneView& sc = *g.sc;
    LightDraw d;
    if (g.presampled)
    {
        PresampledLight pl = SamplePresampledSet(sc, g.sampleSetIdx, shadingPos, rng);
        d.pos = pl.pos; d.normal = pl.normal; d.le = pl.le; d.pdf = pl.pdf; d.ID = pl.ID;
        return d;
    }
    uint32_t u0 = rng.UniformUintBounded(g.numEmissives);
    const zr_alias_entry ae = sc.alias[u0];
    uint32_t lidx; float lpdfSrc;
    if (rng.Uniform() < ae.p_curr) { lpdfSrc = ae.cached_p_orig; lidx = u0; }
    else { lpdfSrc = ae.cached_p_alias; lidx = ae.alias; }
    const zr_emissive_triangle em = sc.emissives[lidx];
    V2 bary = UniformSampleTriangle(rng.Uniform2D());
    const V3 vtx0 = v3p(em.vtx0), vtx1 = EmV1(em), vtx2 = EmV2(em);
    d.pos = (1.0f - bary.x - bary.y) * vtx0 + bary.x * vtx1 + bary.y * vtx2;
    V3 ln = cross(vtx1 - vtx0, vtx2 - vtx0);
    bool normalIs0 = dot(ln, ln) == 0;
    float twoArea = length(ln);
    float lpdfPos = normalIs0 ? 0.0f : 2.0f / twoArea;
    ln = normalIs0 ? ln : ln / twoArea;
    d.normal = EmDoubleSided(em) && dot(shadingPos - d.pos, ln) < 0 ? -ln : ln;
    d.le = EmLe(sc, em, bary);
    d.pdf = lpdfSrc * lpdfPos;
    d.ID = em.id;
    return d;
}

// RGI_Util::NEE_Emissive_LVG, ReSTIR_GI_NEE.hlsli:121-187 (numSamples = 1; globals.extents / offset_y are fp16, ReSTIR_GI.hlsl:51-53)
ZR_HD V3 NEE_Emissive_LVG(const Globals& gl, const zr_frame_constants& g, V3 pos, V3 normal, Surface surface, Rng& rng)
{
    const SceneView& sc = *gl.sc;
    const V3 ext = v3(zr_round_f16(sc.lvgExtents[0]), zr_round_f16(sc.lvgExtents[1]), zr_round_f16(sc.lvgExtents[2]));
    const float offset_y = zr_round_f16(sc.lvgOffsetY);
    zr_voxel_sample s;
    V3 lpos, lnormal, le; float lightPdf; uint32_t lightID;
    if (LvgSample(sc, pos, g.curr_view, ext, offset_y, s, rng))
    {
        lpos = v3p(s.pos); lnormal = DecodeOct32(s.normal);
        le = v3(zr_f16_to_f32(s.le[0]), zr_f16_to_f32(s.le[1]), zr_f16_to_f32(s.le[2]));
        lightPdf = s.pdf; lightID = s.id;
        if (s.two_sided && dot(lnormal, pos - lpos) < 0) lnormal = lnormal * -1.0f;
    }
    else
    {
        PresampledLight pl = SamplePresampledSet(sc, gl.sampleSetIdx, pos, rng);
        lpos = pl.pos; lnormal = pl.normal; le = pl.le; lightPdf = pl.pdf; lightID = pl.ID;
    }
    V3 ret = v3(0.0f);
    const float t = length(lpos - pos);
    const V3 wi = (lpos - pos) / t;
    if (lightID != 0xffffffffu && dot(lnormal, -wi) > 0)
    {
        const float dwdA = zr_saturate(dot(lnormal, -wi)) / (t * t);
        surface.SetWi(wi, normal);
        le = le * (Unified(sc.rho, surface).f * dwdA);
        if (Luminance(le) > 1e-6f) le = le * (VisibilitySegmentApprox(gl, pos, wi, t, normal, lightID, surface.Transmissive()) ? 1.0f : 0.0f);
        ret = ret + le / zr_max(lightPdf, 1e-6f);
    }
    return ret;
}

using rpt::VisibilityRay;       // RtRayQuery::Visibility_Ray, traced in place (zr_rpt.h)
// RGI_Util::NEE with NEE_EMISSIVE == 0 (ReSTIR_GI_NEE.hlsli:194-226): ReSTIR_Util::NEE_Sun<true> with probability q,
// else NEE_Sky<true> (NEE.hlsli:86-152); P_SUN_VS_SKY 0.65, SUN_DISK_SAMPLING 0 (ReSTIR_GI/Params.hlsli:8,28)
ZR_HD V3 NEE_SunSky(const Globals& gl, const zr_frame_constants& g, V3 pos, V3 normal, const Surface& surface, Rng& rng)
{
    const SceneView& sc = *gl.sc;
    const float p_sun = rng.Uniform();
    const V3 sunDir = v3p(g.sun_dir);
    if (!(-sunDir.y > 0)) return v3(0.0f);
    const float q = (surface.Transmissive() ? 1.0f : (dot(-sunDir, normal) > 0 ? 1.0f : 0.0f)) * 0.65f;
    if (p_sun < q)
    {
        const V3 wi = -sunDir;
        Surface ss = surface;
        ss.SetWi(wi, normal);
        const V3 f = Unified(sc.rho, ss).f;
        V3 ld = v3(0.0f);
        if (!(dot(f, f) == 0) && VisibilityRay(gl, pos, wi, normal, surface.Transmissive())) ld = f * Le_Sun(pos, g);
        return ld / q;
    }
    SkyIncidentRadiance leFunc; leFunc.lut = sc.sky;
    const BsdfSample bs = SampleBSDF(sc.rho, normal, surface, leFunc, rng);
    V3 ld = bs.bsdfOverPdf;
    if (dot(ld, ld) > 0) ld = ld * (VisibilityRay(gl, pos, bs.wi, normal, surface.Transmissive()) ? 1.0f : 0.0f);
    return ld / (1 - q);
}

// RGI_Util::NEE_Emissive_MIS<1, skipDiffuse = true> (ReSTIR_GI_NEE.hlsli:8-118), APPROXIMATE_EMISSIVE_SHADOW_RAY 1
ZR_HD V3 NEE_Emissive_MIS(const Globals& g, V3 pos, V3 normal, Surface surface, Rng& rng)
{
    const SceneView& sc = *g.sc;
    V3 ld = v3(0.0f);
    const int numLightSamples = IsSpecular(surface) ? 0 : 1;
    {
        BsdfSample bs;
        { V2 u_c = rng.Uniform2D(); V2 u_g = rng.Uniform2D(); float u0 = rng.Uniform(), u1 = rng.Uniform(); bs = SampleBSDF_NoDiffuse(sc.rho, normal, surface, u_c, u_g, u0, u1); }
        V3 wi = bs.wi, f = bs.f; float wiPdf = bs.pdf;
        HitEm hitInfo = rpt::FindClosestEm(g, pos, normal, wi, surface.Transmissive());
        if (hitInfo.emissiveTriIdx != 0xffffffffu)
        {
            const zr_emissive_triangle em = sc.emissives[hitInfo.emissiveTriIdx];
            V3 le = EmLe(sc, em, v2(hitInfo.bu, hitInfo.bv));
            const V3 vtx0 = v3p(em.vtx0), vtx1 = EmV1(em), vtx2 = EmV2(em);
            V3 ln = cross(vtx1 - vtx0, vtx2 - vtx0);
            float twoArea = length(ln);
            twoArea = zr_max(twoArea, 1e-6f);
            ln = dot(ln, ln) == 0 ? v3(1.0f) : ln / twoArea;
            ln = EmDoubleSided(em) && dot(-wi, ln) < 0 ? -ln : ln;
            const float lightSourcePdf = numLightSamples > 0 ? sc.alias[hitInfo.emissiveTriIdx].cached_p_orig : 0;
            const float lightPdf = lightSourcePdf * (2.0f / twoArea);
            float dwdA = hitInfo.t > 0 ? zr_saturate(dot(ln, -wi)) / (hitInfo.t * hitInfo.t) : 0;
            wiPdf *= dwdA;
            le = le * (f * dwdA);
            ld = PowerHeuristic(wiPdf, lightPdf, le, 1.0f, (float)numLightSamples);
        }
    }
    for (int s_l = 0; s_l < numLightSamples; s_l++)
    {
        LightDraw d = DrawLight(g, pos, rng);
        V3 le = d.le;
        const float t = length(d.pos - pos);
        const V3 wi = (d.pos - pos) / t;
        if (dot(d.normal, -wi) > 0)
        {
            const float dwdA = zr_saturate(dot(d.normal, -wi)) / (t * t);
            surface.SetWi(wi, normal);
            le = le * (Unified(sc.rho, surface).f * dwdA);
            if (dot(le, le) > 0) le = le * (VisibilitySegmentApprox(g, pos, wi, t, normal, d.ID, surface.Transmissive()) ? 1.0f : 0.0f);
            float bsdfPdf = BSDFSamplerPdf_NoDiffuse(sc.rho, normal, surface, wi);
            bsdfPdf *= dwdA;
            ld = ld + PowerHeuristic(d.pdf, bsdfPdf, le, (float)numLightSamples, 1.0f);
        }
    }
    return ld;
}

// ReSTIR_Util::NEE_Emissive<1> (NEE.hlsli:150-222)
ZR_HD V3 NEE_Emissive_Power(const Globals& g, V3 pos, V3 normal, Surface surface, Rng& rng)
{
    V3 ret = v3(0.0f);
    LightDraw d = DrawLight(g, pos, rng);
    const float t = length(d.pos - pos);
    const V3 wi = (d.pos - pos) / t;
    if (dot(d.normal, -wi) > 0)
    {
        const float dwdA = zr_saturate(dot(d.normal, -wi)) / (t * t);
        surface.SetWi(wi, normal);
        V3 ld = d.le * Unified(g.sc->rho, surface).f * dwdA;
        if (Luminance(ld) > 1e-6f) ld = ld * (VisibilitySegmentApprox(g, pos, wi, t, normal, d.ID, surface.Transmissive()) ? 1.0f : 0.0f);
        ret = ret + ld / d.pdf;
    }
    return ret / 1.0f;
}

struct Lane
{
    bool valid, active, atRR, hasSample;
    uint32_t x, y; size_t px;
    V3 pos, normal; float roughness, ior, z_view; Surface surface;
    Rng rngThread, rngGroup; int maxNumBounces; uint32_t sampleSetIdx;
    BsdfSample firstSample; V3 hitPos, hitNormal; uint32_t hitID;
    V3 li, throughput, ppos, pnormal; float eta_curr, eta_next; int bounce; bool inMedium;
    BsdfSample bs; HitInfo hit; Surface psurface;
    Reservoir r;
    RayDiffs rd; V3 dpdx, dpdy;      // textured scenes only
};

ZR_HD Globals MakeGlobals(const GiFrame& F, const zr_frame_constants& g, const Lane& P, TravStack stack, uint32_t* cnt)
{
    Globals gl; gl.sc = &F.sc; gl.frame = &g; gl.emissive = g.num_emissive_triangles != 0; gl.numEmissives = g.num_emissive_triangles; gl.alpha_min = 0; gl.stack = stack; gl.cnt = cnt; gl.maxNumBounces = P.maxNumBounces;
    gl.presampled = F.prm.numSampleSets != 0; gl.sampleSetIdx = P.sampleSetIdx;
    return gl;
}

// Hit::FindClosest<ID, true> over this lane's continuation ray
ZR_HD bool TraceContinuation(const Globals& gl, V3 pos, V3 normal, V3 wi, bool transmissive, bool wantID, HitInfo& hit, bool wantDiffs = false)
{
    F4 ro, rd;
    if (!MakeClosestRay(pos, normal, wi, transmissive, false, &ro, &rd)) return false;
    gl.cnt[0]++;
    RawHit h = Traverse<false>(*gl.sc, xyz(ro), xyz(rd), ro.w, rd.w, ZR_SUBGROUP_ALL, gl.stack);
    if (h.tri == kInvalidTri) return false;
    const TriMeta tm = gl.sc->triMeta[h.tri];
    hit.t = h.t;
    if (wantDiffs) FillHit<true>(*gl.sc, tm.mesh, tm.prim, h.u, h.v, wantID, hit, true);
    else FillHit<false>(*gl.sc, tm.mesh, tm.prim, h.u, h.v, wantID, hit, true);
    return true;
}

// ReSTIR_GI.hlsl main prologue + EstimateIndirectLighting / RIS_InitialCandidates up to the PathTrace call
// the primary hit of pixel (x, y) as the G-buffer holds it: position, normal, material -> P.pos / normal / roughness / ior / z_view / surface.
// -DZR_RGI_REMAT=1 calls it AGAIN after the path loop instead of keeping ~45 registers of primary-hit state live across it (none of it is
// touched while the path is traced; the 128-VGPR kernel spills it).  Measured (scripts/gpu_r03_trip.sh [rounds 1-4: git history up to 31e92fa; today scripts/gpu.sh ab], Cornell 1080p): k_rgi 1.282 ms without,
// 1.295 ms with; HBM-side traffic 2.21 -> 2.32 GB per launch -- the register allocator spills something else instead, nothing gained.  Off.
ZR_HD void LoadPrimary(const GiFrame& F, const zr_frame_constants& g, uint32_t x, uint32_t y, size_t px, Lane& P, V2& lens, V3& origin)
{
    const uint16_t mrp = F.gb.mr[px];
    const GFlags flags = DecodeFlags(mrp);
    const Camera cam = CurrCamera(g);
    P.z_view = F.gb.depth[px];
    lens = v2(0, 0);
    if (cam.dof)
    {
        uint32_t hx = x, hy = y, hz = x; zr_pcg3d(&hx, &hy, &hz);
        Rng rr = Rng::Init(hz, hy, g.frame_num);
        lens = UniformSampleDiskConcentric(rr.Uniform2D());
        lens = lens * cam.lensRadius;
    }
    origin = cam.origin;
    P.pos = rpt::WorldPosSS2(cam, (float)x, (float)y, P.z_view, lens, origin);
    P.normal = DecodeOct32u(F.gb.normal[px]);
    const V3 baseColor = UnpackRGB8(F.gb.baseColor[px]);
    P.roughness = RoughnessOf(mrp);
    P.ior = kDefaultEtaMat;
    if (flags.transmissive && !F.gb.plain) P.ior = DecodeIOR(zr_div255((float)F.gb.ior[px]));
    const V3 wo = normalize(origin - P.pos);
    P.surface = InitSurface(P.normal, wo, flags.metallic, P.roughness, baseColor, kEtaAir, P.ior, flags.transmissive, 0.0f, 0.0f, 0.0f, v3(0.0f), 0.0f, kDefaultEtaCoat, F.gb.plain != 0);
}
// wo-only term groups (zr_dev_bsdf.h WO_*) K10 prepares on the surfaces of its path
#ifndef ZR_PREP_RGI
#define ZR_PREP_RGI 1
#endif
static constexpr uint32_t kPrepRgi = ZR_PREP_RGI;
#ifndef ZR_RGI_REMAT
#define ZR_RGI_REMAT 0
#endif
// after the path loop: the primary hit again, through addresses the compiler cannot connect to the first load
ZR_HD void ReloadPrimary(const GiFrame& F, const zr_frame_constants& g, Lane& P)
{
#if ZR_RGI_REMAT && defined(__HIP_DEVICE_COMPILE__)
    if (!P.valid) return;
    uint32_t x = P.x, y = P.y; uint32_t pxl = (uint32_t)P.px;
    __asm__ volatile("" : "+v"(x), "+v"(y), "+v"(pxl));
    V2 lens; V3 origin;
    LoadPrimary(F, g, x, y, (size_t)pxl, P, lens, origin);
#endif
}

ZR_HD void InitLane(const GiFrame& F, const zr_frame_constants& g, uint32_t x, uint32_t y, TravStack stack, uint32_t* cnt, Lane& P)
{
    const GiParams& prm = F.prm;
    P.valid = false; P.active = false; P.atRR = false; P.hasSample = false; P.x = x; P.y = y;
    if (!F.Owns(x, y)) return;
    P.px = Pix(F.gb, x, y);
    const uint16_t mrp = F.gb.mr[P.px];
    GFlags flags = DecodeFlags(mrp);
    if (flags.invalid || flags.emissive)
    {
        if (!prm.accumulate) { float* o = F.finalRGBA + 4 * P.px; o[0] = 0; o[1] = 0; o[2] = 0; }
        return;
    }
    P.valid = true;
    const Camera cam = CurrCamera(g);
    V2 lens; V3 origin;
    LoadPrimary(F, g, x, y, P.px, P, lens, origin);
    P.rngGroup = Rng::Init((x >> 3) ^ 61u, (y >> 3) ^ 61u, g.frame_num);
    P.rngThread = Rng::Init(x ^ 511u, y ^ 31u, g.frame_num);
    P.maxNumBounces = flags.transmissive ? (int)prm.maxGlossyTrBounces : (int)prm.maxNonTrBounces;
    if ((prm.flags & ZR_IND_STOCHASTIC_MULTI_BOUNCE) && (P.roughness >= 0.1f || g.camera_static))
        P.maxNumBounces = P.rngGroup.Uniform() < 0.5f ? 1 : P.maxNumBounces;
    P.sampleSetIdx = P.rngGroup.UniformUintBounded_Faster(prm.numSampleSets);
    P.r = InitReservoir();
    {   // the lobe candidates of the first sample share the primary surface's wo-only terms; a copy, so that they are not live across the path loop
        Surface sp = P.surface;
        if (kPrepRgi) PrepareWo(F.sc.rho, sp, kPrepRgi);
        P.firstSample = SampleBSDF(F.sc.rho, P.normal, sp, P.rngThread);
    }
    if (P.firstSample.pdf == 0) return;
    if (prm.textured)
    {
        // Resampling.hlsli:60-75: camera ray differentials -> uv gradients at the primary hit -> differentials of the first bounce
        const TriDiffs td = UnpackTriDiffs(&F.gb.triA[4 * P.px], &F.gb.triB[2 * P.px]);
        P.rd = RayDiffs::Init((int)x, (int)y, cam.renderDim, cam.tanHalfFOV, cam.aspect, cam.jitter, cam.vbx, cam.vby, cam.vbz, cam.dof,
            cam.focusDepth, lens, origin);
        V3 dpdx, dpdy;
        P.rd.dpdx_dpdy(P.pos, P.normal, dpdx, dpdy);
        P.rd.ComputeUVDifferentials(dpdx, dpdy, td.dpdu, td.dpdv);
        P.rd.UpdateRays(P.pos, P.normal, P.firstSample.wi, P.surface.wo, td.dndu, td.dndv, dpdx, dpdy, dot(P.firstSample.wi, P.normal) < 0, P.surface.eta);
    }
    Globals gl = MakeGlobals(F, g, P, stack, cnt);
    if (!TraceContinuation(gl, P.pos, P.normal, P.firstSample.wi, P.surface.Transmissive(), true, P.hit, prm.textured)) return;
    P.hasSample = true;
    P.hitPos = P.pos + P.hit.t * P.firstSample.wi;
    P.hitNormal = P.hit.normal; P.hitID = P.hit.ID;
    P.active = true; P.li = v3(0.0f); P.throughput = v3(1.0f);
    P.ppos = P.pos; P.pnormal = P.normal;
    P.eta_curr = dot(P.normal, P.firstSample.wi) < 0 ? P.ior : kEtaAir;
    P.bounce = 0; P.inMedium = dot(P.normal, P.firstSample.wi) < 0;
    P.bs = P.firstSample;
}

// PathTracing.hlsli:23-62 (GI parameters)
ZR_HD void PhaseA(const GiFrame& F, const zr_frame_constants& g, TravStack stack, uint32_t* cnt, Lane& P)
{
    P.atRR = false;
    if (!P.active) return;
    Globals gl = MakeGlobals(F, g, P, stack, cnt);
    V3 hitPos = mad(P.hit.t, P.bs.wi, P.ppos);
    float eta_mat;
    V4 uvGrads = v4(0, 0, 0, 0);
    if (F.prm.textured)
    {
        P.rd.dpdx_dpdy(hitPos, P.hit.normal, P.dpdx, P.dpdy);
        P.rd.ComputeUVDifferentials(P.dpdx, P.dpdy, P.hit.dpdu, P.hit.dpdv);
        uvGrads = P.rd.uv_grads;
    }
    if (!GetMaterialData(F.sc, -P.bs.wi, P.eta_curr, P.hit, P.psurface, eta_mat, uvGrads, F.prm.textured)) { P.active = false; return; }
    if (kPrepRgi) PrepareWo(F.sc.rho, P.psurface, kPrepRgi);      // the vertex is evaluated three to five times (NEE, its MIS sampler, the continuation's lobes)
    P.eta_next = eta_mat;
    // RGI_Util::NEE (NEE_EMISSIVE == 1, USE_MIS == 1, MIS_ALL_BOUNCES == 0)
    V3 ld;
    if (g.num_emissive_triangles == 0) ld = NEE_SunSky(gl, g, hitPos, P.hit.normal, P.psurface, P.rngThread);     // NEE_EMISSIVE == 0
    else if (P.bounce == 0) ld = NEE_Emissive_MIS(gl, hitPos, P.hit.normal, P.psurface, P.rngThread);
    else if (F.prm.useLVG && gl.presampled) ld = NEE_Emissive_LVG(gl, g, hitPos, P.hit.normal, P.psurface, P.rngThread);     // USE_LVG && USE_PRESAMPLED_SETS
    else ld = NEE_Emissive_Power(gl, hitPos, P.hit.normal, P.psurface, P.rngThread);
    P.li = P.li + P.throughput * ld;
    if (P.bounce >= (P.maxNumBounces - 1)) { P.active = false; return; }
    P.ppos = hitPos; P.pnormal = P.hit.normal;
    P.bounce++;
    P.atRR = (F.prm.flags & ZR_IND_RUSSIAN_ROULETTE) && (P.bounce >= 3);
}
ZR_HD uint32_t RRKey(const Lane& P)
{
    if (!(P.active && P.atRR)) return 0;
    float lum = Luminance(P.throughput);
    return (zr_isnan(lum) || lum < 0) ? 0u : zr_asuint(lum);
}
// PathTracing.hlsli:62-95
ZR_HD void PhaseB(const GiFrame& F, const zr_frame_constants& g, TravStack stack, uint32_t* cnt, Lane& P, uint32_t waveMaxBits)
{
    if (!P.active) return;
    if (P.atRR)
    {
        float p_terminate = zr_max(0.05f, 1 - zr_asfloat(waveMaxBits));
        if (P.rngGroup.Uniform() < p_terminate) { P.active = false; return; }
        P.throughput = P.throughput / (1 - p_terminate);
    }
    P.bs = InitBsdfSample();
    if (P.bounce < P.maxNumBounces) P.bs = SampleBSDF(F.sc.rho, P.pnormal, P.psurface, P.rngThread);
    if (Luminance(P.bs.bsdfOverPdf) == 0) { P.active = false; return; }
    Globals gl = MakeGlobals(F, g, P, stack, cnt);
    if (!TraceContinuation(gl, P.ppos, P.pnormal, P.bs.wi, P.psurface.Transmissive(), false, P.hit, F.prm.textured)) { P.active = false; return; }
    P.throughput = P.throughput * P.bs.bsdfOverPdf;
    bool transmitted = dot(P.pnormal, P.bs.wi) < 0;
    P.eta_curr = transmitted ? (P.eta_curr == kEtaAir ? P.eta_next : kEtaAir) : P.eta_curr;
    P.inMedium = transmitted ? !P.inMedium : P.inMedium;
    // with the new hit's triangle differentials, as the reference does (PathTracing.hlsli:93-94)
    if (F.prm.textured) P.rd.UpdateRays(P.ppos, P.pnormal, P.bs.wi, P.psurface.wo, P.hit.dndu, P.hit.dndv, P.dpdx, P.dpdy, transmitted, P.psurface.eta);
}

// Resampling.hlsli:283-303
ZR_HD float JacobianReconnectionShift(V3 x2_normal, V3 x1_r, V3 x1_q, V3 x2_q)
{
    V3 v_r = x1_r - x2_q;
    const float t_r2 = dot(v_r, v_r);
    v_r = dot(v_r, v_r) == 0 ? v_r : v_r / zr_max(zr_sqrt(t_r2), 1e-6f);
    V3 v_q = x1_q - x2_q;
    const float t_q2 = dot(v_q, v_q);
    v_q = dot(v_q, v_q) == 0 ? v_q : v_q / zr_max(zr_sqrt(t_q2), 1e-6f);
    float cosPhi_r = dot(v_r, x2_normal);
    float cosPhi_q = dot(v_q, x2_normal);
    return (zr_abs(cosPhi_r) * t_q2) / zr_max(zr_abs(cosPhi_q) * t_r2, 1e-6f);
}

struct TemporalSampleData { V3 posW, normal; float roughness; int px, py; bool metallic, transmissive; float eta_next; };

// Resampling.hlsli:129-231
ZR_HD void FindTemporalCandidate(const GiFrame& F, const zr_frame_constants& g, uint32_t DTx, uint32_t DTy, V3 posW, V3 normal, float viewZ, float roughness,
    bool transmissive, V2 prevUV, Rng& rng, TemporalSampleData* data, bool* valid)
{
    valid[0] = false; valid[1] = false;
    if (prevUV.x < 0 || prevUV.y < 0 || prevUV.x > 1 || prevUV.y > 1) return;
    const uint32_t W = g.render_width, H = g.render_height;
    const V2 renderDim = v2((float)W, (float)H);
    const int ppx = (int)(prevUV.x * renderDim.x), ppy = (int)(prevUV.y * renderDim.y);
    int curr = 0;
    const Camera pcam = PrevCamera(g);
    for (int i = 0; i < 3; i++)
    {
        const float theta = rng.Uniform() * ZR_TWO_PI;
        float sinTheta, cosTheta; zr_sincos(theta, &sinTheta, &cosTheta);
        const float ox = kTemporalRadius * sinTheta, oy = kTemporalRadius * cosTheta;
        const float k = i > 0 ? 1.0f : 0.0f;
        const int sx = zr_f2i_sat((float)ppx + k * ox), sy = zr_f2i_sat((float)ppy + k * oy);
        if ((float)sx >= renderDim.x || (float)sy >= renderDim.y) continue;
        if (i > 0 && sx == (int)DTx && sy == (int)DTy) continue;
        const size_t sp = Texel(F.gbPrev, sx, sy, W, H);
        const uint16_t mrp = sp == (size_t)-1 ? (uint16_t)0 : F.gbPrev.mr[sp];
        GFlags pf = DecodeFlags(mrp);
        if (pf.emissive) continue;
        float viewZ_prev = sp == (size_t)-1 ? 0.0f : F.gbPrev.depth[sp];
        V2 lens = v2(0, 0);
        if (pcam.dof)
        {
            uint32_t hx = (uint32_t)sx, hy = (uint32_t)sy, hz = (uint32_t)sx; zr_pcg3d(&hx, &hy, &hz);
            Rng rr = Rng::Init(hz, hy, g.frame_num - 1);
            lens = UniformSampleDiskConcentric(rr.Uniform2D());
            lens = lens * pcam.lensRadius;
        }
        V3 origin = pcam.origin;
        V3 prevPos = rpt::WorldPosSS2(pcam, (float)sx, (float)sy, viewZ_prev, lens, origin);
        float tolerance = kMaxPlaneDist * (g.dof ? 10.0f : 1.0f);
        if (!(zr_abs(dot(normal, prevPos - posW)) <= tolerance * viewZ)) continue;
        const V3 prevNormal = DecodeOct32u(sp == (size_t)-1 ? 0u : F.gbPrev.normal[sp]);
        const float prevRough = RoughnessOf(mrp);
        // (the candidate in locals, then stored to a slot chosen by a branch: indexing the two-slot arrays with `curr` keeps them in scratch memory)
        bool ok = dot(prevNormal, normal) > 0.1f;
        if (roughness < 0.5f) ok = ok && (zr_abs(prevRough - roughness) < 0.15f);
        float prevEta_mat = kDefaultEtaMat;
        if (pf.transmissive) prevEta_mat = DecodeIOR(zr_div255((float)(sp == (size_t)-1 ? 0 : F.gbPrev.ior[sp])));
        ok = ok && (pf.transmissive == transmissive);
        ok = g.dof ? true : ok;
        if (curr == 0) valid[0] = ok; else valid[1] = ok;
        if (ok)
        {
            TemporalSampleData d;
            d.px = sx; d.py = sy; d.posW = prevPos; d.normal = prevNormal; d.metallic = pf.metallic; d.roughness = prevRough;
            d.transmissive = pf.transmissive; d.eta_next = prevEta_mat;
            if (curr == 0) data[0] = d; else data[1] = d;
            curr++;
            if (curr == 2) break;
        }
    }
}

// Resampling.hlsli:233-281
ZR_HD float TargetLumAtTemporalPixel(const Globals& gl, const GiFrame& F, const zr_frame_constants& g, const Reservoir& r_curr, const TemporalSampleData& c,
    bool testVisibility)
{
    V3 wi = r_curr.pos - c.posW;
    if (dot(wi, wi) == 0) return 0;
    float t = length(wi);
    wi = wi / zr_max(t, 1e-6f);
    const size_t sp = Texel(F.gbPrev, c.px, c.py, g.render_width, g.render_height);
    const V3 baseColor_prev = UnpackRGB8(sp == (size_t)-1 ? 0u : F.gbPrev.baseColor[sp]);
    const Camera pcam = PrevCamera(g);
    V3 camPos_prev = pcam.origin;
    if (pcam.dof)
    {
        uint32_t hx = (uint32_t)c.px, hy = (uint32_t)c.py, hz = (uint32_t)c.px; zr_pcg3d(&hx, &hy, &hz);
        Rng rr = Rng::Init(hz, hy, g.frame_num - 1);
        V2 lens = UniformSampleDiskConcentric(rr.Uniform2D());
        lens = lens * pcam.lensRadius;
        camPos_prev = camPos_prev + mad(lens.x, pcam.vbx, lens.y * pcam.vby);
    }
    const V3 wo_prev = normalize(camPos_prev - c.posW);
    Surface surface_prev = InitSurface(c.normal, wo_prev, c.metallic, c.roughness, baseColor_prev, kEtaAir, c.eta_next, c.transmissive, 0.0f, 0.0f, 0.0f,
        v3(0.0f), 0.0f, kDefaultEtaCoat, F.gbPrev.plain != 0);
    surface_prev.SetWi(wi, c.normal);
    const V3 target_prev = r_curr.Lo * Unified(gl.sc->rho, surface_prev).f;
    const float targetLum_prev = Luminance(target_prev);
    if (testVisibility && targetLum_prev > 1e-5f)
        if (!VisibilitySegmentApprox(gl, c.posW, wi, t, c.normal, r_curr.ID, surface_prev.Transmissive())) return 0;
    return targetLum_prev;
}

// Resampling.hlsli:305-369
ZR_HD void TemporalResample1(const Globals& gl, const GiFrame& F, const zr_frame_constants& g, V3 posW, V3 normal, Surface surface, const TemporalSampleData& c,
    Reservoir& r, Rng& rng)
{
    const size_t sp = Texel(F.gbPrev, c.px, c.py, g.render_width, g.render_height);
    Reservoir r_prev = PartialRead_Reuse(F.prev, sp);
    const uint32_t M_new = (r.M + r_prev.M) & 0xffffu;
    if (r.w_sum != 0)
    {
        float targetLum_prev = 0.0f;
        if (r_prev.M > 0 && Luminance(r.Lo) > 1e-6f) targetLum_prev = TargetLumAtTemporalPixel(gl, F, g, r, c, true);
        const float p_curr = Luminance(r.target_z);
        const float J = JacobianReconnectionShift(r.normal, c.posW, posW, r.pos);
        const float m_curr = p_curr / zr_max(p_curr + (float)r_prev.M * targetLum_prev * J, 1e-6f);
        r.w_sum *= m_curr;
    }
    if (r_prev.ID == 0xffffffffu || dot(r_prev.Lo, v3(1.0f)) == 0)
    {
        float targetLum = Luminance(r.target_z);
        r.W = targetLum > 0.0f ? r.w_sum / targetLum : 0.0f;
        r.M = M_new;
        return;
    }
    V3 wi = r_prev.pos - posW;
    float t = length(wi);
    wi = wi / t;
    surface.SetWi(wi, normal);
    const V3 target_curr = r_prev.Lo * Unified(gl.sc->rho, surface).f;
    const float targetLum_curr = Luminance(target_curr);
    if (targetLum_curr > 1e-6f)
    {
        if (VisibilitySegmentApprox(gl, posW, wi, t, normal, r_prev.ID, surface.Transmissive()))
        {
            PartialRead_Rest(F.prev, sp, r_prev);
            const float targetLum_prev = r_prev.W > 0 ? r_prev.w_sum / r_prev.W : 0;
            const float J = JacobianReconnectionShift(r_prev.normal, posW, c.posW, r_prev.pos);
            const float numerator = (float)r_prev.M * targetLum_prev;
            const float denom = numerator / zr_max(J, 1e-6f) + targetLum_curr;
            const float m_prev = numerator / zr_max(denom, 1e-6f);
            const float w_prev = m_prev * targetLum_curr * r_prev.W;
            r.Update(w_prev, r_prev.pos, r_prev.normal, r_prev.ID, r_prev.Lo, target_curr, rng);
        }
    }
    float targetLum = Luminance(r.target_z);
    r.W = targetLum > 0.0f ? r.w_sum / targetLum : 0.0f;
    r.M = M_new;
}

// Resampling.hlsli:371-454
ZR_HD void TemporalResample2(const Globals& gl, const GiFrame& F, const zr_frame_constants& g, V3 posW, V3 normal, Surface surface, const TemporalSampleData* c,
    Reservoir& r, Rng& rng)
{
    uint32_t M_new = r.M;
    Reservoir r_prev[2]; size_t sp[2];
    ZR_UNROLL for (int k = 0; k < 2; k++)
    {
        sp[k] = Texel(F.gbPrev, c[k].px, c[k].py, g.render_width, g.render_height);
        r_prev[k] = PartialRead_Reuse(F.prev, sp[k]);
        M_new = (M_new + r_prev[k].M) & 0xffffu;
    }
    {
        const float p_curr = Luminance(r.target_z);
        float denom = p_curr;
        if (Luminance(r.Lo) > 1e-5f)
        {
            ZR_UNROLL for (int p = 0; p < 2; p++)
            {
                if (r_prev[p].M == 0) continue;
                float targetLum_prev = TargetLumAtTemporalPixel(gl, F, g, r, c[p], p != 0);
                float J = JacobianReconnectionShift(r.normal, c[p].posW, posW, r.pos);
                denom += (float)r_prev[p].M * J * targetLum_prev;
            }
        }
        const float m_curr = denom == 0 ? 0 : p_curr / denom;
        r.w_sum *= m_curr;
    }
    ZR_UNROLL for (int i = 0; i < 2; i++)
    {
        V3 wi = r_prev[i].pos - posW;
        float t = (wi.x == 0 && wi.y == 0 && wi.z == 0) ? 0 : length(wi);
        wi = wi / zr_max(t, 1e-6f);
        surface.SetWi(wi, normal);
        const V3 target_curr = r_prev[i].Lo * Unified(gl.sc->rho, surface).f;
        const float targetLum_curr = Luminance(target_curr);
        if (targetLum_curr < 1e-5f) continue;
        if (VisibilitySegmentApprox(gl, posW, wi, t, normal, r_prev[i].ID, surface.Transmissive()))
        {
            PartialRead_Rest(F.prev, sp[i], r_prev[i]);
            const float targetLum_prev = r_prev[i].W > 0 ? r_prev[i].w_sum / r_prev[i].W : 0;
            const float J = JacobianReconnectionShift(r_prev[i].normal, posW, c[i].posW, r_prev[i].pos);
            const float numerator = (float)r_prev[i].M * targetLum_prev;
            float denom = (numerator / J) + targetLum_curr;
            if (r_prev[1 - i].M > 0 && targetLum_prev > 0)
            {
                const float J_tt = JacobianReconnectionShift(r_prev[i].normal, c[1 - i].posW, c[i].posW, r_prev[i].pos);
                const float targetLum_other = TargetLumAtTemporalPixel(gl, F, g, r_prev[i], c[1 - i], true);
                denom += (float)r_prev[1 - i].M * targetLum_other / zr_max(J_tt, 1e-6f);
            }
            denom = J == 0 ? 0 : denom;
            const float m_prev = denom == 0 ? 0 : numerator / denom;
            const float w_prev = m_prev * targetLum_curr * r_prev[i].W;
            r.Update(w_prev, r_prev[i].pos, r_prev[i].normal, r_prev[i].ID, r_prev[i].Lo, target_curr, rng);
        }
    }
    float targetLum = Luminance(r.target_z);
    r.W = targetLum > 0.0f ? r.w_sum / targetLum : 0.0f;
    r.M = M_new;
}

// tail of RIS_InitialCandidates + the temporal branch of EstimateIndirectLighting; returns the lane's w_sum for the wave sum
ZR_HD float FinishAndResample(const GiFrame& F, const zr_frame_constants& g, TravStack stack, uint32_t* cnt, Lane& P)
{
    if (!P.valid) return 0.0f;
    if (P.hasSample)
    {
        V3 lo = P.li;
        V3 target = lo;
        if (dot(lo, lo) > 0) { P.surface.SetWi(P.firstSample.wi, P.normal); target = target * Unified(F.sc.rho, P.surface).f; }
        float targetLum = Luminance(target);
        float w = targetLum / zr_max(P.firstSample.pdf, 1e-6f);
        P.r.Update(w, P.hitPos, P.hitNormal, P.hitID, lo, target, P.rngThread);
        P.r.W = targetLum > 0 ? 1.0f / P.firstSample.pdf : 0.0f;
    }
    if (!F.prm.doTemporal) return 0.0f;
    Globals gl = MakeGlobals(F, g, P, stack, cnt);
    const V2 motionVec = DecodeMotion(F.gb.motion[P.px]);
    const V2 currUV = v2(((float)P.x + 0.5f) / (float)g.render_width, ((float)P.y + 0.5f) / (float)g.render_height);
    const V2 prevUV = currUV - motionVec;
    TemporalSampleData data[2]; bool valid[2];
    FindTemporalCandidate(F, g, P.x, P.y, P.pos, P.normal, P.z_view, P.roughness, P.surface.specTr, prevUV, P.rngThread, data, valid);
    if (valid[1] && P.roughness > 0.05f) TemporalResample2(gl, F, g, P.pos, P.normal, P.surface, data, P.r, P.rngThread);
    else if (valid[0]) TemporalResample1(gl, F, g, P.pos, P.normal, P.surface, data[0], P.r, P.rngThread);
    return P.r.w_sum;
}

// SuppressOutlierReservoirs (Resampling.hlsli:519-527) + WriteReservoir + the shader's output
ZR_HD void SuppressAndWrite(const GiFrame& F, Lane& P, float waveSum)
{
    if (!P.valid) return;
    if (F.prm.doTemporal && (F.prm.flags & ZR_IND_BOILING_SUPPRESSION))
    {
        float waveAvg = (waveSum - P.r.w_sum) / 63.0f;
        if (P.r.w_sum > 25 * waveAvg) P.r.M = 1;
    }
    if (F.prm.writeReservoirs) WriteReservoir(F.cur, P.px, P.r, F.prm.M_max);
    V3 li = P.r.target_z * P.r.W;
    li = any_nan(li) ? v3(0.0f) : li;
    float* o = F.finalRGBA + 4 * P.px;
    if (F.prm.accumulate) { o[0] += li.x; o[1] += li.y; o[2] += li.z; }
    else { o[0] = li.x; o[1] = li.y; o[2] = li.z; }
}

} // namespace rgi
} // namespace zr
