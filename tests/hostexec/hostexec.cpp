// tests/hostexec -- TEST-ONLY serial executor for the stage functions of zetaray_amd/csrc/zr_stages.h.
//
// The product library (libzetaray_amd.so) only runs these functions inside HIP kernels and has no CPU path.  This
// harness compiles the very same ZR_HD functions with g++ and drives them with plain loops (queues, slot allocation
// and the trace stage in program order), so the `-m "not gpu"` suite can check the wavefront decomposition against
// the oracle in a container without a GPU.  It is never linked into, or loaded by, the product.
#include <vector>
#include "../../include/zr_srgb_table.h"
#include <cstring>
#include "../../zetaray_amd/csrc/zr_stages.h"
#include "../../zetaray_amd/csrc/zr_bvh.h"
#include "../../zetaray_amd/csrc/zr_sdi.h"
#include "../../zetaray_amd/csrc/zr_rpt.h"
#include "../../zetaray_amd/csrc/zr_rdi.h"
#include "../../zetaray_amd/csrc/zr_rgi.h"
#include "../../zetaray_amd/csrc/zr_taa.h"
#include "../../zetaray_amd/csrc/zr_svgf.h"

static const uint16_t kRptSampleSet[1024] = {
#include "../../zetaray_amd/csrc/zr_rpt_sample_set.inc"
};
static const uint16_t kRdiSampleSet[64] = {
#include "../../zetaray_amd/csrc/zr_rdi_sample_set.inc"
};

#include "hx_kat.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
using namespace zr;
static bool g_k11_park = false;       // zhx_set_k11_park: K11 keeps the reservoir's selected reconnection in a park outside the lane (zr_rpt.h RcPark)
static bool g_k11_fused = false;      // zhx_set_k11_fused: K11 emulation runs the FUSED stage functions (PtInitLane_Fused / PtPhaseA_Fused: what k_rpt_pathtrace compiles) instead of the cut ones
static bool g_plain = false;          // zhx_set_material_class: run the ReSTIR PT stage functions the way the PLAIN kernel permutations do (rpt::SetMaterialClass(F, true))
static bool g_k11_carry = false;      // zhx_set_k11_carry: K11 emulation sends live paths through rpt::PtCarry at every bounce boundary

struct HxScene
{
    std::vector<zr_vertex> vertices; std::vector<uint32_t> indices; std::vector<zr_mesh_instance> instances;
    std::vector<zr_material> materials; std::vector<zr_emissive_triangle> emissives; std::vector<zr_alias_entry> alias;
    std::vector<uint16_t> rho; BuiltBvh bvh; mutable SceneView view;
    // zhx_scene_update_instances: last frame's instance buffer + BVH (what the CtT / temporal-shift stages bind)
    std::vector<zr_mesh_instance> instancesPrev; BuiltBvh bvhPrev; bool hasPrev = false;
    std::vector<uint8_t> mask; std::vector<uint32_t> numTris;
    std::vector<uint8_t> ownSubtree;      // zhx_scene_set_own_subtree: instances the next rebuild keeps in subtrees of their own (zr_bvh.h Build)
    SceneView PrevView() const
    {
        SceneView v = view;
        if (hasPrev) { v.instances = instancesPrev.data(); v.nodes = bvhPrev.nodes4.data(); v.tris = bvhPrev.tris.data(); v.triMeta = bvhPrev.meta.data();
                       v.numNodes = (uint32_t)bvhPrev.nodes4.size(); v.numTris = (uint32_t)bvhPrev.tris.size(); }
        return v;
    } std::vector<zr_texture_desc> texDescs; std::vector<uint8_t> texels; std::vector<zr_presampled_tri> sampleSets; std::vector<uint32_t> skyLut; std::vector<zr_voxel_sample> lvg;
};

// the descriptor-table offsets of the frame constants, latched into the scene view like zr_pass_render does
static void Latch(const HxScene* s, const zr_frame_constants* cb, const zr_params* indirectParams = nullptr)
{
    // the INDIRECT pass binds its TEXTURE_FILTER sampler; every other pass samples with ANISOTROPIC_4X (zr_api.hip FrameView)
    s->view.texFilter = indirectParams ? indirectParams->tex_filter : (uint32_t)ZR_TEX_FILTER_ANISOTROPIC_4X;
    s->view.baseColorMapsOffset = cb->base_color_maps_desc_heap_offset; s->view.normalMapsOffset = cb->normal_maps_desc_heap_offset;
    s->view.mrMapsOffset = cb->metallic_roughness_maps_desc_heap_offset; s->view.emissiveMapsOffset = cb->emissive_maps_desc_heap_offset;
}

struct HxQueue
{
    std::vector<U4> s0, hitC, hitM; std::vector<F4> f[8], rays[6], t[8]; std::vector<uint32_t> lightID, visS;
    void Resize(size_t n)
    {
        for (auto& v : t) v.resize(n);
        s0.resize(n); hitC.resize(n); hitM.resize(n); lightID.resize(n); visS.resize(n);
        for (auto& v : f) v.resize(n);
        for (auto& v : rays) v.resize(n);
    }
    PathQueue View()
    {
        PathQueue q;
        q.s0 = s0.data(); q.s1 = f[0].data(); q.s2 = f[1].data(); q.s3 = f[2].data(); q.s4 = f[3].data(); q.s5 = f[4].data();
        q.s6 = f[5].data(); q.s7 = f[6].data(); q.s8 = f[7].data();
        q.rayC_o = rays[0].data(); q.rayC_d = rays[1].data(); q.rayM_o = rays[2].data(); q.rayM_d = rays[3].data();
        q.rayS_o = rays[4].data(); q.rayS_d = rays[5].data();
        q.sLightID = lightID.data(); q.hitC = hitC.data(); q.hitM = hitM.data(); q.visS = visS.data();
        for (int k = 0; k < 8; k++) q.t[k] = t[k].data();
        return q;
    }
};

static GBuf ViewOf(const zr_gbuffer_planes* p)
{
    GBuf g; g.w = p->width; g.h = p->height; g.x0 = 0; g.y0 = 0;
    g.baseColor = (uint32_t*)p->plane[ZR_GB_BASE_COLOR]; g.normal = (uint32_t*)p->plane[ZR_GB_NORMAL];
    g.mr = (uint16_t*)p->plane[ZR_GB_METALLIC_ROUGHNESS]; g.motion = (uint32_t*)p->plane[ZR_GB_MOTION_VECTOR];
    g.emissive = (uint32_t*)p->plane[ZR_GB_EMISSIVE_COLOR]; g.ior = (uint8_t*)p->plane[ZR_GB_IOR];
    g.coat = (uint16_t*)p->plane[ZR_GB_COAT]; g.depth = (float*)p->plane[ZR_GB_DEPTH];
    g.triA = (uint32_t*)p->plane[ZR_GB_TRI_DIFF_GEO_A]; g.triB = (uint32_t*)p->plane[ZR_GB_TRI_DIFF_GEO_B];
    return g;
}

extern "C" {

HxScene* zhx_scene_create(const zr_scene_desc* d)
{
    HxScene* s = new HxScene();
    s->vertices.assign(d->vertices, d->vertices + d->num_vertices);
    s->indices.assign(d->indices, d->indices + d->num_indices);
    s->instances.assign(d->instances, d->instances + d->num_instances);
    s->materials.assign(d->materials, d->materials + d->num_materials);
    if (d->num_emissives) s->emissives.assign(d->emissives, d->emissives + d->num_emissives);
    s->rho.assign(d->rho_lut, d->rho_lut + (size_t)d->rho_dim[0] * d->rho_dim[1] * d->rho_dim[2]);
    BvhBuilder b;
    s->bvh = b.Build(*d);
    s->mask.assign(d->instance_mask, d->instance_mask + d->num_instances); s->numTris.assign(d->instance_num_tris, d->instance_num_tris + d->num_instances);
    SceneView& v = s->view;
    v.vertices = s->vertices.data(); v.indices = s->indices.data(); v.instances = s->instances.data(); v.materials = s->materials.data();
    v.emissives = s->emissives.data(); v.alias = nullptr; v.sampleSets = nullptr; v.sampleSetSize = 0; v.nodes = s->bvh.nodes4.data(); v.tris = s->bvh.tris.data(); v.triMeta = s->bvh.meta.data();
    v.rho.data = s->rho.data(); v.rho.dx = d->rho_dim[0]; v.rho.dy = d->rho_dim[1]; v.rho.dz = d->rho_dim[2];
    v.numEmissives = d->num_emissives; v.numNodes = (uint32_t)s->bvh.nodes4.size(); v.numTris = (uint32_t)s->bvh.tris.size();
    if (d->num_textures) { s->texDescs.assign(d->textures, d->textures + d->num_textures); s->texels.assign(d->texels, d->texels + d->texel_bytes); }
    v.tex.descs = s->texDescs.data(); v.tex.texels = s->texels.data(); v.tex.srgb = zr_srgb_to_linear_table; v.tex.count = d->num_textures;
    return s;
}
// zr_scene_update_emissives (the view points into the vector, whose size does not change)
void zhx_scene_update_emissives(HxScene* s, const zr_emissive_triangle* tris, uint32_t first, uint32_t count)
{ std::copy(tris, tris + count, s->emissives.begin() + first); }
void zhx_scene_update_instances(HxScene* s, const zr_mesh_instance* instances, const float* instance_to_world, uint32_t n)
{
    s->instancesPrev = s->instances; s->bvhPrev = s->bvh; s->hasPrev = true;
    s->instances.assign(instances, instances + n);
    zr_scene_desc d; memset(&d, 0, sizeof(d));
    d.vertices = s->vertices.data(); d.num_vertices = (uint32_t)s->vertices.size(); d.indices = s->indices.data(); d.num_indices = (uint32_t)s->indices.size();
    d.instances = s->instances.data(); d.num_instances = n; d.instance_to_world = instance_to_world; d.instance_mask = s->mask.data(); d.instance_num_tris = s->numTris.data();
    BvhBuilder b;
    s->bvh = b.Build(d, s->ownSubtree.size() == n ? s->ownSubtree.data() : nullptr);
    SceneView& v = s->view;
    v.instances = s->instances.data(); v.nodes = s->bvh.nodes4.data(); v.tris = s->bvh.tris.data(); v.triMeta = s->bvh.meta.data();
    v.numNodes = (uint32_t)s->bvh.nodes4.size(); v.numTris = (uint32_t)s->bvh.tris.size();
}
void zhx_scene_set_own_subtree(HxScene* s, const uint8_t* flags, uint32_t n) { s->ownSubtree.assign(flags, flags + n); }
void zhx_scene_destroy(HxScene* s) { delete s; }
void zhx_scene_set_alias(HxScene* s, const zr_alias_entry* e, uint32_t n) { s->alias.assign(e, e + n); s->view.alias = s->alias.data(); }
void zhx_bvh_info(const HxScene* s, uint32_t* nodes, uint32_t* tris, uint32_t* depth)
{ *nodes = s->view.numNodes; *tris = s->view.numTris; *depth = s->bvh.maxDepth; }
// K3 for frame `frame_num`
void zhx_presample(HxScene* s, uint32_t frame_num, uint32_t num_sets, uint32_t set_size, zr_presampled_tri* out)
{
    const uint32_t total = num_sets * set_size;
    s->sampleSets.resize(total);
    for (uint32_t i = 0; i < total; i++) s->sampleSets[i] = PresampleEmissive(s->view, i, frame_num, s->view.numEmissives);
    s->view.sampleSets = s->sampleSets.data(); s->view.sampleSetSize = set_size;
    if (out) std::memcpy(out, s->sampleSets.data(), (size_t)total * sizeof(zr_presampled_tri));
}
// K4 through the device stage functions: the 64 threads of a voxel serially, group sums = the canonical butterfly
void zhx_build_lvg(HxScene* s, const zr_frame_constants* cb, const uint32_t* dim, const float* extents, float offset_y, zr_voxel_sample* out)
{ Latch(s, cb);
    const size_t nv = (size_t)dim[0] * dim[1] * dim[2];
    s->lvg.resize(nv * 64);
    const V3 ext = v3(extents[0], extents[1], extents[2]);
    for (uint32_t z = 0; z < dim[2]; z++) for (uint32_t y = 0; y < dim[1]; y++) for (uint32_t x = 0; x < dim[0]; x++)
    {
        const int v[3] = {(int)x, (int)y, (int)z};
        zr_voxel_sample r[64]; float w[64], tz[64]; uint32_t n = 0;
        for (uint32_t t = 0; t < 64; t++) { uint32_t nl; LvgThread(s->view, *cb, dim, ext, offset_y, v, t, r[t], w[t], tz[t], nl); n += nl; }
        const float sum = rpt::ButterflySum64(w);
        for (uint32_t t = 0; t < 64; t++) { LvgFinish(r[t], tz[t], sum, n); s->lvg[(size_t)LvgFlatten(v, dim) * 64 + t] = r[t]; }
    }
    s->view.lvg = s->lvg.data();
    for (int a = 0; a < 3; a++) { s->view.lvgDim[a] = dim[a]; s->view.lvgExtents[a] = extents[a]; }
    s->view.lvgOffsetY = offset_y;
    if (out) std::memcpy(out, s->lvg.data(), s->lvg.size() * sizeof(zr_voxel_sample));
}
// K17 through the device stage function; binds the LUT to the scene like ZR_PASS_SKY does
void zhx_sky_lut(HxScene* s, const zr_frame_constants* cb, uint32_t w, uint32_t h, uint32_t* out)
{
    s->skyLut.resize((size_t)w * h);
    for (uint32_t y = 0; y < h; y++) for (uint32_t x = 0; x < w; x++) s->skyLut[(size_t)y * w + x] = SkyViewLutTexel(*cb, x, y, w, h);
    s->view.sky.data = s->skyLut.data(); s->view.sky.w = w; s->view.sky.h = h;
    if (out) std::memcpy(out, s->skyLut.data(), s->skyLut.size() * 4);
}
void zhx_le_sky(const HxScene* s, const float* dirs, uint32_t n, float* out)
{ for (uint32_t i = 0; i < n; i++) { V3 r = Le_Sky(v3p(dirs + 3 * i), s->view.sky); out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z; } }
void zhx_le_sun(const zr_frame_constants* cb, const float* pos, uint32_t n, float* out)
{ for (uint32_t i = 0; i < n; i++) { V3 r = Le_Sun(v3p(pos + 3 * i), *cb); out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z; } }
void zhx_tex_sample(const HxScene* s, uint32_t tex, int mode, const float* uv, const float* g, uint32_t n, float* out)
{
    for (uint32_t i = 0; i < n; i++)
    {
        float* o = out + 4 * i;
        if (mode == 0) zr_tex_point(&s->view.tex, tex, uv[2 * i], uv[2 * i + 1], o);
        else if (mode == 1) zr_tex_sample_level(&s->view.tex, tex, uv[2 * i], uv[2 * i + 1], g[4 * i], o);
        else zr_tex_sample_grad(&s->view.tex, tex, uv[2 * i], uv[2 * i + 1], g[4 * i], g[4 * i + 1], g[4 * i + 2], g[4 * i + 3], o);
    }
}
float zhx_halton(int i, int b) { return Halton(i, b); }
void zhx_taa(const float* signal, const float* depth, const uint32_t* motion, const uint16_t* prevOut, uint16_t* currOut, uint32_t w, uint32_t h,
    float blendWeight, int temporalValid)
{
    taa::TaaFrame F; F.signal = (const F4*)signal; F.depth = depth; F.motion = motion; F.prevOut = prevOut; F.currOut = currOut; F.w = w; F.h = h;
    F.blendWeight = blendWeight; F.temporalIsValid = temporalValid ? 1u : 0u;
    for (uint32_t y = 0; y < h; y++) for (uint32_t x = 0; x < w; x++) taa::TaaPixel(F, x, y);
}
// the denoise pass's stage functions (zr_svgf.h) with the pass's state, over a WINDOW of the frame (the whole frame, or a tile + apron of the
// multi-device split) and step by step (the ZR_STAGE_DENOISE_* bits of include/zetaray_amd.h), in the order RenderDenoise launches them
struct HxSvgf
{
    svgf::Window win;
    std::vector<F4> hist, accum, ping, pong; std::vector<svgf::GuideN> guide; std::vector<float> moments[2], fw, gz;
    int momIdx = 0; F4* cur = nullptr; const F4* out = nullptr;
};
HxSvgf* zhx_svgf_create(int ox, int oy, int pw, int ph, int W, int H)
{
    HxSvgf* S = new HxSvgf();
    S->win.ox = ox; S->win.oy = oy; S->win.pw = pw; S->win.ph = ph; S->win.W = W; S->win.H = H;
    const size_t n = (size_t)pw * ph;
    S->hist.assign(n, F4{0, 0, 0, 0}); S->accum.assign(n, F4{0, 0, 0, 0}); S->guide.assign(n, svgf::GuideN{0, 0}); S->ping.assign(n, F4{0, 0, 0, 0}); S->pong.assign(n, F4{0, 0, 0, 0});
    S->moments[0].assign(2 * n, 0.0f); S->moments[1].assign(2 * n, 0.0f); S->fw.assign(n, 0.0f); S->gz.assign(n, 0.0f);
    S->cur = S->ping.data(); S->out = S->ping.data();
    return S;
}
void zhx_svgf_destroy(HxSvgf* S) { delete S; }
// planes are the window's (pw x ph); steps: ZR_STAGE_DENOISE_* bits
void zhx_svgf_render(HxSvgf* S, const float* signal, const float* depth, const uint32_t* normal, const uint32_t* motion, const float* prevDepth, const uint32_t* prevNormal,
    int temporalValid, const float* params4, uint32_t normalPowerLog2, uint32_t iterations, uint32_t steps)
{
    const svgf::Window& w = S->win;
    const int mi = S->momIdx;
    if (steps & ZR_STAGE_DENOISE_TEMPORAL) memcpy(S->gz.data(), depth, S->gz.size() * sizeof(float));      // the a-trous taps read the G-buffer's depth plane (kept for the later steps of the frame)
    svgf::SvgfParams sp; sp.alpha = params4[0]; sp.alphaMoments = params4[1]; sp.sigmaL = params4[2]; sp.sigmaZ = params4[3]; sp.normalPowerLog2 = normalPowerLog2; sp.iterations = iterations;
    if (steps & ZR_STAGE_DENOISE_TEMPORAL)
    {
        svgf::SvgfFrame T; T.signal = (const F4*)signal; T.depth = depth; T.normal = normal; T.motion = motion; T.prevDepth = prevDepth; T.prevNormal = prevNormal;
        T.histColor = S->hist.data(); T.histMoments = S->moments[mi].data(); T.accum = S->accum.data(); T.moments = S->moments[mi ^ 1].data(); T.guide = S->guide.data(); T.guideFw = S->fw.data();
        T.win = w; T.temporalValid = temporalValid ? 1u : 0u; T.prm = sp;
        for (int y = w.oy; y < w.oy + w.ph; y++) for (int x = w.ox; x < w.ox + w.pw; x++) svgf::TemporalPixel(T, x, y);
    }
    svgf::FilterFrame V; V.src = S->accum.data(); V.moments = S->moments[mi ^ 1].data(); V.guide = S->guide.data(); V.guideFw = S->fw.data(); V.guideZ = S->gz.data(); V.dst = S->ping.data(); V.lenSrc = S->accum.data();
    V.history = iterations == 0 ? S->hist.data() : nullptr; V.win = w; V.step = 1; V.prm = sp; V.dstPacked = iterations != 0;
    if (steps & ZR_STAGE_DENOISE_VARIANCE)
    {
        for (int y = w.oy; y < w.oy + w.ph; y++) for (int x = w.ox; x < w.ox + w.pw; x++) svgf::VariancePixel(V, x, y);
        S->cur = S->ping.data();
    }
    for (uint32_t it = 0; it < iterations; it++)
    {
        if (!(steps & ZR_STAGE_DENOISE_ATROUS(it))) continue;
        F4* src = S->cur; F4* dst = src == S->ping.data() ? S->pong.data() : S->ping.data();
        svgf::FilterFrame A = V; A.src = src; A.dst = dst; A.moments = nullptr; A.step = 1u << it; A.history = it == 0 ? S->hist.data() : nullptr; A.dstPacked = it + 1u != iterations;
        // odd iterations through the unrolled form of the stencil, even ones through the row loop: both forms of zr_svgf.h run on the host
        for (int y = w.oy; y < w.oy + w.ph; y++) for (int x = w.ox; x < w.ox + w.pw; x++)
        {
            svgf::PlaneTaps t; t.srcP = (const U4*)A.src; t.guideZ = A.guideZ; t.win = w;
            if (normalPowerLog2 == 7u) { if (it & 1u) svgf::AtrousPixelT<7, false>(A, x, y, t); else svgf::AtrousPixelT<7, true>(A, x, y, t); }
            else { if (it & 1u) svgf::AtrousPixelT<-1, false>(A, x, y, t); else svgf::AtrousPixelT<-1, true>(A, x, y, t); }
        }
        S->cur = dst;
    }
    const uint32_t last = iterations ? ZR_STAGE_DENOISE_ATROUS(iterations - 1u) : (uint32_t)ZR_STAGE_DENOISE_VARIANCE;
    if (steps & last) { S->out = S->cur; S->momIdx = mi ^ 1; }
}
// which: 0 = colour history (4 floats), 1 = moment history (2 floats; the set the next temporal step reads), 2 = the plane the next a-trous iteration reads
// (4 floats), 3 = the frame's output (4 floats)
static float* SvgfPlane(HxSvgf* S, int which, int* ch)
{
    switch (which)
    {
    case 0: *ch = 4; return (float*)S->hist.data();
    case 1: *ch = 2; return S->moments[S->momIdx].data();
    case 2: *ch = 4; return (float*)S->cur;
    default: *ch = 4; return (float*)S->out;
    }
}
void zhx_svgf_read_plane(HxSvgf* S, int which, float* out) { int ch; const float* p = SvgfPlane(S, which, &ch); memcpy(out, p, (size_t)S->win.pw * S->win.ph * ch * sizeof(float)); }
// copy the rect (x, y, w, h) (window-local texels) of the window-sized array `full` into the plane
void zhx_svgf_write_plane_rect(HxSvgf* S, int which, const float* full, uint32_t x, uint32_t y, uint32_t w, uint32_t h)
{
    int ch; float* p = SvgfPlane(S, which, &ch);
    for (uint32_t r = 0; r < h; r++) { const size_t o = ((size_t)(y + r) * S->win.pw + x) * ch; memcpy(p + o, full + o, (size_t)w * ch * sizeof(float)); }
}
// one full frame in one call (hist_color / hist_moments in and out): the interface of oracle zro_svgf
void zhx_svgf(const float* signal, const float* depth, const uint32_t* normal, const uint32_t* motion, const float* prevDepth, const uint32_t* prevNormal,
    float* histColor, float* histMoments, int temporalValid, const float* params4, uint32_t normalPowerLog2, uint32_t iterations, uint32_t w, uint32_t h, float* out)
{
    HxSvgf* S = zhx_svgf_create(0, 0, (int)w, (int)h, (int)w, (int)h);
    const size_t n = (size_t)w * h;
    memcpy(S->hist.data(), histColor, n * sizeof(F4)); memcpy(S->moments[0].data(), histMoments, 2 * n * sizeof(float));
    zhx_svgf_render(S, signal, depth, normal, motion, prevDepth, prevNormal, temporalValid, params4, normalPowerLog2, iterations, ZR_STAGE_DENOISE_MASK);
    memcpy(histColor, S->hist.data(), n * sizeof(F4)); memcpy(histMoments, S->moments[S->momIdx].data(), 2 * n * sizeof(float));
    memcpy(out, S->out, n * sizeof(F4));
    zhx_svgf_destroy(S);
}
// FNV-1a over the built tree: the 4-wide nodes, the triangles in leaf order, the stack bound (what the device would be given)
uint64_t zhx_bvh_digest(const HxScene* s, uint32_t* numNodes, uint32_t* numTris, uint32_t* stackNeed)
{
    uint64_t h = 1469598103934665603ull;
    auto eat = [&](const void* p, size_t n) { const uint8_t* b = (const uint8_t*)p; for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; } };
    eat(s->bvh.nodes4.data(), s->bvh.nodes4.size() * sizeof(Bvh4Node));
    eat(s->bvh.tris.data(), s->bvh.tris.size() * sizeof(BvhTri));
    eat(&s->bvh.stackNeed, 4);
    if (numNodes) *numNodes = (uint32_t)s->bvh.nodes4.size();
    if (numTris) *numTris = (uint32_t)s->bvh.tris.size();
    if (stackNeed) *stackNeed = s->bvh.stackNeed;
    return h;
}
void zhx_set_k11_carry(int on) { g_k11_carry = on != 0; }
void zhx_set_k11_fused(int on) { g_k11_fused = on != 0; }
void zhx_set_k11_park(int on) { g_k11_park = on != 0; }
// the material class as the PLAIN kernel permutations see it (zr_kernels.h: SceneView::plain, GBuf::plain, RBuf::plain all 1): only legal for scenes whose
// material table IS plain (zr_api.hip MaterialsArePlain) -- the caller's responsibility here as it is the host's in the product
void zhx_set_material_class(int plain) { g_plain = plain != 0; }
void zhx_latch_heap_offsets(const HxScene* s, const zr_frame_constants* cb) { Latch(s, cb); }
void zhx_estimate_power(const HxScene* s, float* out) { for (size_t i = 0; i < s->emissives.size(); i++) out[i] = EstimateTriPower(s->view, s->emissives[i]); }

static uint32_t g_tile_x0 = 0, g_tile_y0 = 0;
// screen-tile origin for the next zhx_gbuffer / zhx_pathtrace calls (planes then have the tile's size)
void zhx_set_tile_origin(uint32_t x0, uint32_t y0) { g_tile_x0 = x0; g_tile_y0 = y0; }

static uint32_t g_pick_xy = 0xffffffffu, g_picked = 0xfffffffeu;
// GBufferRT::PickPixel for the next zhx_gbuffer calls (x = 0xffff: none); zhx_picked = what the last one wrote
void zhx_pick_pixel(uint32_t x, uint32_t y) { g_pick_xy = x | (y << 16); }
uint32_t zhx_picked() { return g_picked; }
void zhx_gbuffer(const HxScene* s, const zr_frame_constants* cb, zr_gbuffer_planes* planes)
{ Latch(s, cb);
    GBuf gb = ViewOf(planes);
    gb.x0 = g_tile_x0; gb.y0 = g_tile_y0;
    zr::StackEntry stackMem[zr::kTravStack]; zr::TravStack stack; stack.lds = nullptr; stack.stride = 0; stack.mem = stackMem;
    for (uint32_t y = gb.y0; y < gb.y0 + gb.h; y++)
        for (uint32_t x = gb.x0; x < gb.x0 + gb.w; x++)
            GBufferPixel(s->view, *cb, gb, x, y, stack, nullptr, (x | (y << 16)) == g_pick_xy ? &g_picked : nullptr);
}

void zhx_pathtrace(const HxScene* s, const zr_frame_constants* cb, const zr_gbuffer_planes* planes, const zr_params* params,
    float* finalRGBA, zr_counters* counters)
{ Latch(s, cb, params);
    GBuf gb = ViewOf(planes);
    gb.x0 = g_tile_x0; gb.y0 = g_tile_y0;
    const uint32_t W = gb.w, H = gb.h;
    const size_t cap = (size_t)W * H;
    PtParams prm;
    prm.maxNonTrBounces = params->max_non_tr_bounces; prm.maxGlossyTrBounces = params->max_glossy_tr_bounces;
    prm.russianRoulette = (params->flags & ZR_IND_RUSSIAN_ROULETTE) ? 1u : 0u;
    prm.numSampleSets = params->presampling ? params->num_sample_sets : 0u;
    prm.accumulate = (cb->accumulate && cb->camera_static) ? 1u : 0u;
    prm.tileW = W; prm.groupsX = (W + 7) / 8;
    std::vector<uint32_t> groupMax((size_t)prm.groupsX * ((H + 7) / 8));
    HxQueue q[2]; q[0].Resize(cap); q[1].Resize(cap);
    std::vector<F4> firstBOP(cap);
    uint64_t nClosest = 0, nShadow = 0;
    zr::StackEntry stackMem[zr::kTravStack]; zr::TravStack stack; stack.lds = nullptr; stack.stride = 0; stack.mem = stackMem;

    const bool tex = s->view.tex.count != 0;
    uint32_t count = 0;
    PathQueue q0 = q[0].View();
    // same pixel order as the kernel: 16x16 tiles, 8x8 quadrants
    for (uint32_t ty = 0; ty < (H + 15) / 16; ty++) for (uint32_t tx = 0; tx < (W + 15) / 16; tx++)
    for (uint32_t t = 0; t < 256; t++)
    {
        uint32_t wave = t >> 6, lane = t & 63;
        uint32_t x = gb.x0 + tx * 16 + (wave & 1) * 8 + (lane & 7), y = gb.y0 + ty * 16 + (wave >> 1) * 8 + (lane >> 3);
        if (x >= gb.x0 + W || y >= gb.y0 + H) continue;
        PathOut po;
        PtInitPixel(s->view, *cb, gb, prm, x, y, finalRGBA, firstBOP.data(), po, tex);
        if (po.alive) WritePath(q0, count++, po, tex);
    }
    const uint32_t maxB = prm.maxNonTrBounces > prm.maxGlossyTrBounces ? prm.maxNonTrBounces : prm.maxGlossyTrBounces;
    for (uint32_t r = 0; r < maxB + 1; r++)
    {
        PathQueue in = q[r & 1].View(), out = q[(r + 1) & 1].View();
        for (uint32_t i = 0; i < count; i++)
        {
            if (in.rayC_d[i].w >= 0) { in.hitC[i] = TraceClosestRay(s->view, in.rayC_o[i], in.rayC_d[i], ZR_SUBGROUP_ALL, stack); nClosest++; }
            else { U4 m; m.x = m.y = m.z = 0; m.w = kInvalidTri; in.hitC[i] = m; }
            if (in.rayM_d[i].w >= 0) { in.hitM[i] = TraceClosestRay(s->view, in.rayM_o[i], in.rayM_d[i], ZR_SUBGROUP_ALL, stack); nClosest++; }
            if (in.rayS_d[i].w >= 0) { in.visS[i] = TraceSegmentRay(s->view, in.rayS_o[i], in.rayS_d[i], in.sLightID[i], stack); nShadow++; }
        }
        uint32_t outCount = 0;
        std::fill(groupMax.begin(), groupMax.end(), 0u);
        for (uint32_t i = 0; i < count; i++)
        {
            PathOut po;
            PtShadePath(s->view, *cb, prm, in, i, finalRGBA, firstBOP.data(), groupMax.data(), po, tex);
            if (po.alive) WritePath(out, outCount++, po, tex);
        }
        count = outCount;
        for (uint32_t i = 0; i < count; i++) PtRussianRoulette(s->view, prm, out, i, groupMax.data(), tex);
    }
    if (counters) { counters->n_closest = nClosest; counters->n_shadow = nShadow; }
}

void zhx_trace_closest(const HxScene* s, const float* rays, uint32_t n, uint32_t mask, uint32_t* hits)
{
    zr::StackEntry stackMem[zr::kTravStack]; zr::TravStack stack; stack.lds = nullptr; stack.stride = 0; stack.mem = stackMem;
    for (uint32_t i = 0; i < n; i++)
    {
        F4 ro = f4(rays[8 * i], rays[8 * i + 1], rays[8 * i + 2], rays[8 * i + 3]);
        F4 rd = f4(rays[8 * i + 4], rays[8 * i + 5], rays[8 * i + 6], rays[8 * i + 7]);
        U4 h = TraceClosestRay(s->view, ro, rd, mask, stack);
        hits[4 * i] = h.x; hits[4 * i + 1] = h.y; hits[4 * i + 2] = h.z; hits[4 * i + 3] = h.w;
    }
}
// the built tree's topology for inspection (tools/bvh_quality.py --dump): 4 child words per wide node
void zhx_bvh_children(const HxScene* s, uint32_t* out) { for (size_t i = 0; i < s->bvh.nodes4.size(); i++) for (int c = 0; c < 4; c++) out[4 * i + c] = s->bvh.nodes4[i].child[c]; }
// tree quality (tools/bvh_quality.py): the ordered stack traversal of zr_dev_scene.h over `rays` with its steps counted -- out = {inner nodes visited (4 box tests each),
// leaves visited, triangles tested, hits}; any_hit: stop at the first hit (shadow rays)
void zhx_trace_stats(const HxScene* s, const float* rays, uint32_t n, uint32_t mask, int any_hit, uint64_t* out)
{
    zr::StackEntry stackMem[zr::kTravStack]; zr::TravStack stack; stack.lds = nullptr; stack.stride = 0; stack.mem = stackMem;
    uint64_t nodes = 0, leaves = 0, tris = 0, hits = 0;
    for (uint32_t i = 0; i < n; i++)
    {
        TravState st;
        TravInit(s->view, st, v3(rays[8 * i], rays[8 * i + 1], rays[8 * i + 2]), v3(rays[8 * i + 4], rays[8 * i + 5], rays[8 * i + 6]), rays[8 * i + 3], rays[8 * i + 7], mask, false, 0);
        for (;;)
        {
            if (st.cur & kLeafBit) { leaves++; tris += st.cur == kWholeSceneLeaf ? s->view.numTris : (st.cur & 7u) + 1u; } else nodes++;
            if (TravStep(s->view, st, stack, any_hit != 0)) break;
        }
        hits += st.best.tri != kInvalidTri;
    }
    out[0] = nodes; out[1] = leaves; out[2] = tris; out[3] = hits;
}
// The device's VOTED scheduling of the same state machine (zr_dev_scene.h TraverseDyn, device branch) replayed on the host: consecutive groups of 64 rays are
// "waves"; every iteration the lanes still traversing vote for the inner-node phase or the leaf phase and the wave runs the one with more takers.  What a
// traversal call costs a wave is its ITERATIONS, not its rays' own steps (the section profiler: 10.5 iterations per call for rays that need 3.8 + 1.4), and
// that depends on the tree's shape -- depth, how alike the rays' step counts are -- in ways the per-ray counts do not show.  out = {node iterations, leaf
// iterations, lanes in node iterations, lanes in leaf iterations, calls}.  active (optional, one byte per ray): lanes that issue a ray (a wave's idle lanes).
void zhx_trace_vote_stats(const HxScene* s, const float* rays, const uint8_t* active, uint32_t n, uint32_t mask, int any_hit, uint64_t* out)
{
    uint64_t nodeIt = 0, triIt = 0, nodeLanes = 0, triLanes = 0, calls = 0;
    std::vector<zr::StackEntry> stackMem((size_t)64 * zr::kTravStack);
    for (uint32_t base = 0; base < n; base += 64)
    {
        const uint32_t cnt = std::min(64u, n - base);
        TravState st[64]; TravLane L[64]; zr::TravStack stack[64]; bool live[64];
        bool any = false;
        for (uint32_t l = 0; l < cnt; l++)
        {
            const uint32_t i = base + l;
            live[l] = !active || active[i];
            stack[l].lds = nullptr; stack[l].stride = 0; stack[l].mem = stackMem.data() + (size_t)l * zr::kTravStack;
            L[l].triCur = 0; L[l].triEnd = 0; L[l].done = !live[l];
            if (!live[l]) continue;
            any = true;
            TravInit(s->view, st[l], v3(rays[8 * i], rays[8 * i + 1], rays[8 * i + 2]), v3(rays[8 * i + 4], rays[8 * i + 5], rays[8 * i + 6]), rays[8 * i + 3], rays[8 * i + 7], mask, false, 0);
            TravEnter(s->view, st[l], L[l], st[l].cur);
        }
        if (!any) continue;
        calls++;
        for (;;)
        {
            uint32_t nNode = 0, nTri = 0;
            for (uint32_t l = 0; l < cnt; l++) { const bool atTri = L[l].triCur < L[l].triEnd; nTri += atTri; nNode += !L[l].done && !atTri; }
            if (nNode + nTri == 0) break;
            static const int wn = std::getenv("ZHX_VOTE_WN") ? std::atoi(std::getenv("ZHX_VOTE_WN")) : ZR_VOTE_WN, wt = std::getenv("ZHX_VOTE_WT") ? std::atoi(std::getenv("ZHX_VOTE_WT")) : ZR_VOTE_WT;
            if ((uint32_t)wn * nNode >= (uint32_t)wt * nTri)
            {
                nodeIt++; nodeLanes += nNode;
                for (uint32_t l = 0; l < cnt; l++) if (!L[l].done && !(L[l].triCur < L[l].triEnd)) TravNodePhase(s->view, st[l], L[l], stack[l]);
            }
            else
            {
                triIt++; triLanes += nTri;
                for (uint32_t l = 0; l < cnt; l++) if (L[l].triCur < L[l].triEnd) TravTriPhase(s->view, st[l], L[l], stack[l], any_hit != 0);
            }
        }
    }
    out[0] = nodeIt; out[1] = triIt; out[2] = nodeLanes; out[3] = triLanes; out[4] = calls;
}
void zhx_trace_any(const HxScene* s, const float* rays, uint32_t n, uint32_t mask, uint32_t* occ)
{
    zr::StackEntry stackMem[zr::kTravStack]; zr::TravStack stack; stack.lds = nullptr; stack.stride = 0; stack.mem = stackMem;
    for (uint32_t i = 0; i < n; i++)
    {
        RawHit h = Traverse<true>(s->view, v3(rays[8 * i], rays[8 * i + 1], rays[8 * i + 2]), v3(rays[8 * i + 4], rays[8 * i + 5], rays[8 * i + 6]),
            rays[8 * i + 3], rays[8 * i + 7], mask, stack);
        occ[i] = h.tri != kInvalidTri ? 1u : 0u;
    }
}


// ---------------------------------------------------------------- ReSTIR PT (zr_rpt.h) in program order
struct HxRpt
{
    uint32_t w = 0, h = 0; bool temporalValid = false, doTemporal = false, doSpatial = false, frameOpen = false; int currIdx = 0;
    std::vector<uint16_t> map[2];      // K12 thread maps (CtN, NtC)
    struct Planes { std::vector<uint32_t> A, G; std::vector<float> B, F; std::vector<U4> C, D; std::vector<uint16_t> E;
        void Resize(size_t n) { A.assign(n, 0); B.assign(2 * n, 0); C.assign(n, U4{0, 0, 0, 0}); D.assign(n, U4{0, 0, 0, 0}); E.assign(n, 0); F.assign(2 * n, 0); G.assign(2 * n, 0); }
        rpt::ResPlanes View() { rpt::ResPlanes p; p.A = A.data(); p.B = B.data(); p.C = C.data(); p.D = D.data(); p.E = E.data(); p.F = F.data(); p.G = G.data(); return p; } } res[2];
    struct RB { std::vector<uint16_t> A, D; std::vector<U4> B, C;
        void Resize(size_t n) { A.assign(4 * n, 0); D.assign(n, 0); B.assign(n, U4{0, 0, 0, 0}); C.assign(n, U4{0, 0, 0, 0}); }
        rpt::RBuf View() { rpt::RBuf r; r.A = A.data(); r.B = B.data(); r.C = C.data(); r.D = D.data(); return r; } } rb[2];
    std::vector<F4> target; std::vector<uint8_t> neighbor;
};

HxRpt* zhx_rpt_create(uint32_t w, uint32_t h)
{
    HxRpt* r = new HxRpt(); r->w = w; r->h = h; size_t n = (size_t)w * h;
    for (auto& p : r->res) p.Resize(n);
    for (auto& p : r->rb) p.Resize(n);
    r->target.assign(n, F4{0, 0, 0, 0}); r->neighbor.assign(2 * n, 0);
    return r;
}
void zhx_rpt_destroy(HxRpt* r) { delete r; }
void zhx_rpt_reset_temporal(HxRpt* r) { r->temporalValid = false; }

// owned rect (global pixel coordinates) for the screen-tile split; w == 0 -> the whole plane rect
static uint32_t g_own[4] = {0, 0, 0, 0};
void zhx_rpt_set_owned_rect(uint32_t x0, uint32_t y0, uint32_t w, uint32_t h) { g_own[0] = x0; g_own[1] = y0; g_own[2] = w; g_own[3] = h; }

// stages: 1 = K11 + temporal passes, 2 = spatial passes + end-of-frame bookkeeping, 3 = whole frame.
// The planes (`curr`, `prev`, reservoirs, finalRGBA) cover the extended tile whose origin zhx_set_tile_origin gave.
void zhx_rpt_render_stage(const HxScene* s, HxRpt* R, const zr_frame_constants* cb, const zr_gbuffer_planes* curr, const zr_gbuffer_planes* prev,
    const zr_params* params, float* finalRGBA, zr_counters* counters, int stages)
{ Latch(s, cb, params);
    using namespace rpt;
    uint32_t cnt[2] = {0, 0}; uint64_t total[2] = {0, 0};
    auto flush = [&]() { total[0] += cnt[0]; total[1] += cnt[1]; cnt[0] = cnt[1] = 0; };
    const zr_frame_constants& g = *cb;
    RptFrame F;
    F.sc = s->view; F.scPrev = s->PrevView(); F.gb = ViewOf(curr); F.gb.x0 = g_tile_x0; F.gb.y0 = g_tile_y0;
    F.gbPrev = prev ? ViewOf(prev) : F.gb; F.gbPrev.x0 = g_tile_x0; F.gbPrev.y0 = g_tile_y0;
    F.ox0 = g_own[2] ? g_own[0] : F.gb.x0; F.oy0 = g_own[2] ? g_own[1] : F.gb.y0;
    F.ow = g_own[2] ? g_own[2] : F.gb.w; F.oh = g_own[2] ? g_own[3] : F.gb.h;
    const uint32_t X0 = F.ox0, Y0 = F.oy0, X1 = F.ox0 + F.ow, Y1 = F.oy0 + F.oh;
    F.rbCtN = R->rb[0].View(); F.rbNtC = R->rb[1].View(); F.tex.target = R->target.data(); F.tex.neighbor = R->neighbor.data();
    F.finalRGBA = finalRGBA; F.sampleSet = kRptSampleSet;
    SetMaterialClass(F, g_plain);
    RptParams& prm = F.prm;
    prm.maxNonTrBounces = params->max_non_tr_bounces; prm.maxGlossyTrBounces = params->max_glossy_tr_bounces;
    prm.russianRoulette = (params->flags & ZR_IND_RUSSIAN_ROULETTE) ? 1u : 0u;
    prm.numSampleSets = params->presampling ? params->num_sample_sets : 0u;
    prm.accumulate = (g.accumulate && g.camera_static) ? 1u : 0u;
    prm.boiling = (params->flags & ZR_IND_BOILING_SUPPRESSION) ? 1u : 0u;
    prm.M_max_temporal = params->m_max_temporal & 0xf; prm.M_max_spatial = params->m_max_spatial & 0xf; prm.alpha_min = params->alpha_min;
    prm.emissive = g.num_emissive_triangles ? 1u : 0u;
    prm.textured = s->view.tex.count ? 1u : 0u;
    prm.temporalMap = 0; prm.sortTemporal = (params->flags & ZR_IND_SORT_TEMPORAL) ? 1u : 0u; prm.sortSpatial = (params->flags & ZR_IND_SORT_SPATIAL) ? 1u : 0u;
    R->map[0].resize((size_t)F.gb.w * F.gb.h); R->map[1].resize((size_t)F.gb.w * F.gb.h);
    F.mapCtN = R->map[0].data(); F.mapNtC = R->map[1].data();
    // K12 on the host: the kernel's bucket order (wave, lane, quad slot) is plain thread order when the group runs serially
    auto sortPass = [&](int variant, uint16_t* map)
    {
        const uint32_t W = g.render_width, H = g.render_height, dimX = (W + 31) / 32, dimY = (H + 31) / 32;
        struct Px { uint32_t x, y, gtx, gty, res; };
        std::vector<Px> bucket[5];
        for (uint32_t gy = F.oy0 / 32; gy < (F.oy0 + F.oh + 31) / 32; gy++) for (uint32_t gx = F.ox0 / 32; gx < (F.ox0 + F.ow + 31) / 32; gx++)
        {
            for (auto& b : bucket) b.clear();
            const bool againstEdge = gx == dimX - 1 || gy == dimY - 1, lastGroup = gx == dimX - 1 && gy == dimY - 1;
            auto write = [&](const Px& q, uint32_t mgx, uint32_t mgy)
            {
                if (gx == dimX - 1 && gy != dimY - 1) std::swap(mgx, mgy);
                const uint32_t mx = gx * 32 + mgx, my = gy * 32 + mgy;
                if (mx < W && my < H && InPlanes(F.gb, (int)mx, (int)my))
                    map[Pix(F.gb, mx, my)] = EncodeSorted(q.x, q.y, mx, my, SortErrorBits(variant, q.res, prm.doSpatial != 0));
            };
            for (uint32_t gidx = 0; gidx < 256; gidx++) for (uint32_t i = 0; i < 4; i++)
            {
                Px q; q.gtx = (gidx & 15) * 2 + (i & 1); q.gty = (gidx >> 4) * 2 + (i >> 1); q.x = gx * 32 + q.gtx; q.y = gy * 32 + q.gty;
                const uint32_t c = SortClassify(F, g, variant, q.x, q.y, againstEdge, q.res);
                if (lastGroup) write(q, q.gtx, q.gty); else bucket[c].push_back(q);
            }
            uint32_t idx = 0;
            for (auto& b : bucket) for (const Px& q : b) { write(q, idx & 31u, idx >> 5); idx++; }
        }
    };
    if (stages & 1)
    {
        R->doTemporal = (params->flags & ZR_IND_TEMPORAL_RESAMPLE) && R->temporalValid && prev;
        R->doSpatial = (params->flags & ZR_IND_SPATIAL_RESAMPLE) && R->doTemporal && params->num_spatial_passes > 0;
    }
    const uint32_t numSpatialPasses = params->num_spatial_passes > 2u ? 2u : params->num_spatial_passes;
    prm.doTemporal = R->doTemporal ? 1u : 0u;
    prm.doSpatial = R->doSpatial ? 1u : 0u;
    prm.writeReservoirs = (prm.doTemporal || !R->temporalValid) ? 1u : 0u;
    F.cur = R->res[R->currIdx].View(); F.prev = R->res[1 - R->currIdx].View();
    zr::StackEntry stackMem[zr::kTravStack]; zr::TravStack stack; stack.lds = nullptr; stack.stride = 0; stack.mem = stackMem;

    if (stages & 1)
    {
        // K11: waves = 16x4 pixel blocks of the global grid
        std::vector<PTLane> lanes(64);
        std::vector<uint32_t> parkWords(kRcParkWords * 64, 0xCDCDCDCDu);
        for (uint32_t by = Y0 / 4; by < (Y1 + 3) / 4; by++) for (uint32_t bx = X0 / 16; bx < (X1 + 15) / 16; bx++)
        {
            for (uint32_t l = 0; l < 64; l++)
            {
                const uint32_t x = bx * 16 + (l & 15), y = by * 4 + (l >> 4);
                if (g_k11_fused) PtInitLane_Fused(F.sc, g, F.gb, prm, F.Owns(x, y), x, y, finalRGBA, stack, cnt, lanes[l]);
                else PtInitLane(F.sc, g, F.gb, prm, F.Owns(x, y), x, y, finalRGBA, stack, cnt, lanes[l]);
                // k_rpt_pathtrace_park (zr_rpt.h RcPark): the reservoir's selected reconnection parked outside the lane, [word][lane] like the LDS block
                if (g_k11_park) { lanes[l].r.park.p = parkWords.data() + l; lanes[l].r.park.stride = 64; lanes[l].r.parked = false; }
            }
            for (;;)
            {
                bool any = false;
                for (uint32_t l = 0; l < 64; l++)
                {
                    if (lanes[l].active) any = true;
                    if (g_k11_fused) PtPhaseA_Fused(F.sc, g, prm, stack, cnt, lanes[l]); else PtPhaseA(F.sc, g, prm, stack, cnt, lanes[l]);
                }
                if (!any) break;
                uint32_t bits = 0;
                for (uint32_t l = 0; l < 64; l++) { uint32_t b = PtRRKey(lanes[l]); bits = b > bits ? b : bits; }
                for (uint32_t l = 0; l < 64; l++) PtPhaseB(F.sc, prm, lanes[l], bits);
                if (g_k11_carry)
                {   // the bounce boundary of the compacting kernels (zr_kernels.h k_rpt_pt_first / _next): a live path continues from NOTHING but the
                    // words rpt::PtCarry moves -- stored from this lane, loaded into a lane whose every other byte is poison
                    for (uint32_t l = 0; l < 64; l++)
                    {
                        if (!lanes[l].active) continue;
                        uint32_t words[kPtCarryWords + 8];
                        PtCarryStore st; st.p = words; st.stride = 1; PtCarry(st, lanes[l]);
                        if (st.n != kPtCarryWords) { std::fprintf(stderr, "PtCarry moves %u words, kPtCarryWords says %u\n", st.n, kPtCarryWords); std::abort(); }
                        PTLane fresh; std::memset((void*)&fresh, 0xCD, sizeof(fresh));
                        fresh.r.park.p = nullptr; fresh.r.park.stride = 0; fresh.r.parked = false;      // (a default-constructed lane has no park: k_rpt_pt_next)
                        PtCarryLoad ld; ld.p = words; ld.stride = 1; PtCarry(ld, fresh);
                        lanes[l] = fresh;
                    }
                }
            }
            for (uint32_t l = 0; l < 64; l++) PtFinishLane(F.gb, prm, F.cur, F.tex, finalRGBA, lanes[l]);
            flush();
        }
        if (prm.doTemporal)
        {
            sortPass(RPT_SORT_TTC, F.mapNtC); sortPass(RPT_SORT_CTT, F.mapCtN);      // unconditional, like the reference
            for (int v = 0; v < 2; v++) for (uint32_t y = Y0; y < Y1; y++) for (uint32_t x = X0; x < X1; x++) { ReplayTemporalPixel(F, g, v, x, y, stack, cnt); flush(); }
            for (uint32_t y = Y0; y < Y1; y++) for (uint32_t x = X0; x < X1; x++) { ReconnectTemporalPixel(F, g, x, y, stack, cnt); flush(); }
        }
    }
    for (uint32_t spass = 0; spass < numSpatialPasses && prm.doSpatial; spass++)
    {
        if (!(stages & (spass == 0 ? 2 : 4))) continue;      // ZR_STAGE_SPATIAL / ZR_STAGE_SPATIAL2
        F.cur = R->res[R->currIdx].View(); F.prev = R->res[1 - R->currIdx].View();      // a round reads the current set and writes the other, which becomes current
        for (uint32_t y = Y0; y < Y1; y++) for (uint32_t x = X0; x < X1; x++) SpatialSearchPixel(F, g, x, y);
        if (prm.sortSpatial) { sortPass(RPT_SORT_CTS, F.mapCtN); sortPass(RPT_SORT_STC, F.mapNtC); }
        for (int v = 0; v < 2; v++) for (uint32_t y = Y0; y < Y1; y++) for (uint32_t x = X0; x < X1; x++) { ReplaySpatialPixel(F, g, v, x, y, stack, cnt); flush(); }
        std::vector<StcLane> L(64);
        float v1[64], v2[64], v3[64], v4[64];
        for (uint32_t gy = Y0 / 8; gy < (Y1 + 7) / 8; gy++) for (uint32_t gx = X0 / 8; gx < (X1 + 7) / 8; gx++)
        {
            for (uint32_t l = 0; l < 64; l++)
            {
                uint32_t x = gx * 8 + (l & 7), y = gy * 8 + (l >> 3);
                if (prm.sortSpatial && F.Owns(x, y) && !DecodeSorted(F.mapNtC[Pix(F.gb, x, y)], x, y)) x = 0xffffffffu;      // k_rpt_stc
                StcPhase0(F, g, x, y, L[l], v1[l], v2[l]);
                if (L[l].valid && L[l].hasN) ReconnectCtSPixel(F, g, L[l].x, L[l].y, stack, cnt);
            }
            const float sum1 = ButterflySum64(v1), sum2 = ButterflySum64(v2);
            for (uint32_t l = 0; l < 64; l++) StcPhase1(F, g, L[l], sum1, v3[l]);
            const float sum3 = ButterflySum64(v3);
            for (uint32_t l = 0; l < 64; l++) StcPhase2(F, g, L[l], sum1, stack, cnt, v4[l]);
            const float sum4 = ButterflySum64(v4);
            for (uint32_t l = 0; l < 64; l++) StcPhase3(F, g, L[l], sum2 + sum3 + sum4);
        }
        R->currIdx = 1 - R->currIdx;      // one flip per round: the round's outputs are the next round's inputs
    }
    flush();
    if (counters) { counters->n_closest = total[0]; counters->n_shadow = total[1]; }
    if (stages & 1) R->frameOpen = true;
    // the frame ends with its last stage (the second round when there is one this frame, else stage 2); Render() flips again
    const bool lastStage = (prm.doSpatial && numSpatialPasses == 2u) ? (stages & 4) != 0 : (stages & 2) != 0;
    if (lastStage && R->frameOpen)
    {
        R->temporalValid = true;
        R->currIdx = 1 - R->currIdx;
        R->frameOpen = false;
    }
}
void zhx_rpt_render(const HxScene* s, HxRpt* R, const zr_frame_constants* cb, const zr_gbuffer_planes* curr, const zr_gbuffer_planes* prev,
    const zr_params* params, float* finalRGBA, zr_counters* counters)
{ Latch(s, cb, params); zhx_rpt_render_stage(s, R, cb, curr, prev, params, finalRGBA, counters, 3); }

// which: 0 = the set the next frame reads as "previous", 1 = the other.  plane: 0..6 = A..G, 7 = target, 8 = neighbor
int zhx_rpt_read_plane(const HxRpt* R, int which, int plane, void* out)
{
    const HxRpt::Planes& p = R->res[which == 0 ? 1 - R->currIdx : R->currIdx];
    auto cp = [&](const void* src, size_t bytes) { std::memcpy(out, src, bytes); return 0; };
    switch (plane)
    {
    case 0: return cp(p.A.data(), p.A.size() * 4);
    case 1: return cp(p.B.data(), p.B.size() * 4);
    case 2: return cp(p.C.data(), p.C.size() * 16);
    case 3: return cp(p.D.data(), p.D.size() * 16);
    case 4: return cp(p.E.data(), p.E.size() * 2);
    case 5: return cp(p.F.data(), p.F.size() * 4);
    case 6: return cp(p.G.data(), p.G.size() * 4);
    case 7: return cp(R->target.data(), R->target.size() * 16);
    case 8: return cp(R->neighbor.data(), R->neighbor.size());
    case 18: case 19: return cp(R->map[plane - 18].data(), R->map[plane - 18].size() * 2);
    }
    if (plane >= 10 && plane <= 17)
    {
        const HxRpt::RB& b = R->rb[(plane - 10) / 4];
        switch ((plane - 10) % 4)
        {
        case 0: return cp(b.A.data(), b.A.size() * 2);
        case 1: return cp(b.B.data(), b.B.size() * 16);
        case 2: return cp(b.C.data(), b.C.size() * 16);
        default: return cp(b.D.data(), b.D.size() * 2);
        }
    }
    return 1;
}

// overwrite the rows [y0, y0 + h) x columns [x0, x0 + w) (plane-local coordinates) of a reservoir plane from a full-size host array
int zhx_rpt_write_plane_rect(HxRpt* R, int which, int plane, const void* src, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h)
{
    HxRpt::Planes& p = R->res[which == 0 ? 1 - R->currIdx : R->currIdx];
    void* base; size_t bpp;
    switch (plane)
    {
    case 0: base = p.A.data(); bpp = 4; break;
    case 1: base = p.B.data(); bpp = 8; break;
    case 2: base = p.C.data(); bpp = 16; break;
    case 3: base = p.D.data(); bpp = 16; break;
    case 4: base = p.E.data(); bpp = 2; break;
    case 5: base = p.F.data(); bpp = 8; break;
    case 6: base = p.G.data(); bpp = 8; break;
    default: return 1;
    }
    for (uint32_t y = y0; y < y0 + h; y++)
        std::memcpy((char*)base + ((size_t)y * R->w + x0) * bpp, (const char*)src + ((size_t)y * R->w + x0) * bpp, (size_t)w * bpp);
    return 0;
}

// ---------------------------------------------------------------- ReSTIR DI (zr_rdi.h) in program order
struct HxRdi
{
    uint32_t w = 0, h = 0; bool temporalValid = false; int currIdx = 0;
    std::vector<U4> A[2]; std::vector<float> B[2]; std::vector<F4> target;
};
HxRdi* zhx_rdi_create(uint32_t w, uint32_t h)
{
    HxRdi* r = new HxRdi(); r->w = w; r->h = h; size_t n = (size_t)w * h;
    for (int i = 0; i < 2; i++) { r->A[i].assign(n, U4{0, 0, 0, 0}); r->B[i].assign(2 * n, 0.0f); }
    r->target.assign(n, F4{0, 0, 0, 0});
    return r;
}
void zhx_rdi_destroy(HxRdi* r) { delete r; }
void zhx_rdi_reset_temporal(HxRdi* r) { r->temporalValid = false; r->currIdx = 0; }
void zhx_rdi_render(const HxScene* s, HxRdi* R, const zr_frame_constants* cb, const zr_gbuffer_planes* curr, const zr_gbuffer_planes* prev,
    const zr_params* params, float* finalRGBA, zr_counters* counters)
{ Latch(s, cb);
    using namespace rdi;
    uint32_t cnt[2] = {0, 0};
    const zr_frame_constants& g = *cb;
    const uint32_t W = g.render_width, H = g.render_height;
    DiFrame F;
    F.sc = s->view; F.gb = ViewOf(curr); F.gbPrev = prev ? ViewOf(prev) : F.gb;
    F.scPrev = s->PrevView();
    F.ox0 = 0; F.oy0 = 0; F.ow = W; F.oh = H;
    F.cur.A = R->A[R->currIdx].data(); F.cur.B = R->B[R->currIdx].data();
    F.prev.A = R->A[1 - R->currIdx].data(); F.prev.B = R->B[1 - R->currIdx].data();
    F.target = R->target.data(); F.finalRGBA = finalRGBA; F.sampleSet = kRdiSampleSet;
    DiParams& prm = F.prm;
    prm.flags = params->flags; prm.M_max = params->m_max_temporal; prm.numSampleSets = params->presampling ? params->num_sample_sets : 0u;
    prm.accumulate = (g.accumulate && g.camera_static) ? 1u : 0u;
    prm.doTemporal = (R->temporalValid && (params->flags & ZR_IND_TEMPORAL_RESAMPLE) && prev) ? 1u : 0u;
    prm.doSpatial = (prm.doTemporal && (params->flags & ZR_IND_SPATIAL_RESAMPLE)) ? 1u : 0u;
    prm.writeReservoirs = (prm.doTemporal || !R->temporalValid) ? 1u : 0u;
    prm.halfVec = (params->flags & ZR_DI_HALF_VECTOR_COPY_SHIFT) ? 1u : 0u; prm.alpha_min = params->alpha_min;
    zr::StackEntry stackMem[zr::kTravStack]; zr::TravStack stack; stack.lds = nullptr; stack.stride = 0; stack.mem = stackMem;
    for (uint32_t y = 0; y < H; y++) for (uint32_t x = 0; x < W; x++) TemporalPixel(F, g, x, y, stack, cnt);
    if (prm.doSpatial)
    {
        std::vector<SpatialLane> L(64);
        for (uint32_t gy = 0; gy < (H + 7) / 8; gy++) for (uint32_t gx = 0; gx < (W + 7) / 8; gx++)
        {
            uint32_t sum = 0;
            for (uint32_t l = 0; l < 64; l++) { SpatialPhase0(F, g, gx * 8 + (l & 7), gy * 8 + (l >> 3), L[l]); sum += L[l].disoccluded ? 1u : 0u; }
            for (uint32_t l = 0; l < 64; l++) SpatialPhase1(F, g, L[l], sum, stack, cnt);
        }
    }
    if (counters) { counters->n_closest = cnt[0]; counters->n_shadow = cnt[1]; }
    R->temporalValid = true;
    R->currIdx = 1 - R->currIdx;
}
// plane 0 = reservoir A, 1 = B of the set written by the last frame, 2 = target
int zhx_rdi_read_plane(const HxRdi* R, int plane, void* out)
{
    const int last = 1 - R->currIdx;
    if (plane == 0) std::memcpy(out, R->A[last].data(), R->A[last].size() * 16);
    else if (plane == 1) std::memcpy(out, R->B[last].data(), R->B[last].size() * 4);
    else std::memcpy(out, R->target.data(), R->target.size() * 16);
    return 0;
}

// ---------------------------------------------------------------- ReSTIR GI (zr_rgi.h) in program order
struct HxRgi
{
    uint32_t w = 0, h = 0; bool temporalValid = false; int currIdx = 0;
    std::vector<F4> A[2], C[2]; std::vector<uint16_t> B[2];
};
HxRgi* zhx_rgi_create(uint32_t w, uint32_t h)
{
    HxRgi* r = new HxRgi(); r->w = w; r->h = h; size_t n = (size_t)w * h;
    for (int i = 0; i < 2; i++) { r->A[i].assign(n, F4{0, 0, 0, 0}); r->C[i].assign(n, F4{0, 0, 0, 0}); r->B[i].assign(4 * n, 0); }
    return r;
}
void zhx_rgi_destroy(HxRgi* r) { delete r; }
void zhx_rgi_reset_temporal(HxRgi* r) { r->temporalValid = false; }
void zhx_rgi_render(const HxScene* s, HxRgi* R, const zr_frame_constants* cb, const zr_gbuffer_planes* curr, const zr_gbuffer_planes* prev,
    const zr_params* params, float* finalRGBA, zr_counters* counters)
{ Latch(s, cb, params);
    using namespace rgi;
    uint32_t cnt[2] = {0, 0};
    const zr_frame_constants& g = *cb;
    const uint32_t W = g.render_width, H = g.render_height;
    GiFrame F;
    F.sc = s->view; F.gb = ViewOf(curr); F.gbPrev = prev ? ViewOf(prev) : F.gb;
    F.ox0 = 0; F.oy0 = 0; F.ow = W; F.oh = H;
    F.cur.A = R->A[R->currIdx].data(); F.cur.B = R->B[R->currIdx].data(); F.cur.C = R->C[R->currIdx].data();
    F.prev.A = R->A[1 - R->currIdx].data(); F.prev.B = R->B[1 - R->currIdx].data(); F.prev.C = R->C[1 - R->currIdx].data();
    F.finalRGBA = finalRGBA;
    GiParams& prm = F.prm;
    prm.flags = params->flags; prm.maxNonTrBounces = params->max_non_tr_bounces; prm.maxGlossyTrBounces = params->max_glossy_tr_bounces;
    prm.numSampleSets = params->presampling ? params->num_sample_sets : 0u;
    prm.accumulate = (g.accumulate && g.camera_static) ? 1u : 0u;
    prm.doTemporal = ((params->flags & ZR_IND_TEMPORAL_RESAMPLE) && R->temporalValid && prev) ? 1u : 0u;
    prm.writeReservoirs = (prm.doTemporal || !R->temporalValid) ? 1u : 0u;
    prm.M_max = (float)params->m_max_temporal;
    prm.useLVG = (params->use_lvg && params->presampling) ? 1u : 0u;
    prm.textured = s->view.tex.count ? 1u : 0u;
    zr::StackEntry stackMem[zr::kTravStack]; zr::TravStack stack; stack.lds = nullptr; stack.stride = 0; stack.mem = stackMem;
    std::vector<Lane> L(64);
    float wsum[64];
    for (uint32_t gy = 0; gy < (H + 7) / 8; gy++) for (uint32_t gx = 0; gx < (W + 7) / 8; gx++)
    {
        for (uint32_t l = 0; l < 64; l++) InitLane(F, g, gx * 8 + (l & 7), gy * 8 + (l >> 3), stack, cnt, L[l]);
        for (;;)
        {
            bool any = false;
            for (uint32_t l = 0; l < 64; l++) { if (L[l].active) any = true; PhaseA(F, g, stack, cnt, L[l]); }
            if (!any) break;
            uint32_t bits = 0;
            for (uint32_t l = 0; l < 64; l++) { uint32_t b = RRKey(L[l]); bits = b > bits ? b : bits; }
            for (uint32_t l = 0; l < 64; l++) PhaseB(F, g, stack, cnt, L[l], bits);
        }
        for (uint32_t l = 0; l < 64; l++) wsum[l] = FinishAndResample(F, g, stack, cnt, L[l]);
        const float waveSum = rpt::ButterflySum64(wsum);
        for (uint32_t l = 0; l < 64; l++) SuppressAndWrite(F, L[l], waveSum);
    }
    if (counters) { counters->n_closest = cnt[0]; counters->n_shadow = cnt[1]; }
    R->temporalValid = true;
    R->currIdx = 1 - R->currIdx;
}
int zhx_rgi_read_plane(const HxRgi* R, int plane, void* out)
{
    const int last = 1 - R->currIdx;
    if (plane == 0) std::memcpy(out, R->A[last].data(), R->A[last].size() * 16);
    else if (plane == 1) std::memcpy(out, R->B[last].data(), R->B[last].size() * 2);
    else std::memcpy(out, R->C[last].data(), R->C[last].size() * 16);
    return 0;
}


// ---- ReSTIR DI for sun + sky (zr_sdi.h) ----
struct HxSdi
{
    uint32_t w = 0, h = 0; bool temporalValid = false; int currIdx = 0;
    std::vector<uint8_t> A[2]; std::vector<uint16_t> B[2]; std::vector<float> C[2]; std::vector<F4> target;
};
HxSdi* zhx_sdi_create(uint32_t w, uint32_t h)
{
    HxSdi* r = new HxSdi(); r->w = w; r->h = h; size_t n = (size_t)w * h;
    for (int i = 0; i < 2; i++) { r->A[i].assign(n, 0); r->B[i].assign(2 * n, 0); r->C[i].assign(2 * n, 0.0f); }
    r->target.assign(n, F4{0, 0, 0, 0});
    return r;
}
void zhx_sdi_destroy(HxSdi* r) { delete r; }
void zhx_sdi_reset_temporal(HxSdi* r) { r->temporalValid = false; r->currIdx = 0; }
void zhx_sdi_render(const HxScene* s, HxSdi* R, const zr_frame_constants* cb, const zr_gbuffer_planes* curr, const zr_gbuffer_planes* prev,
    const zr_params* params, float* finalRGBA, zr_counters* counters)
{ Latch(s, cb);
    using namespace sdi;
    uint32_t cnt[2] = {0, 0};
    const zr_frame_constants& g = *cb;
    const uint32_t W = g.render_width, H = g.render_height;
    SkyFrame F;
    F.sc = s->view; F.gb = ViewOf(curr); F.gbPrev = prev ? ViewOf(prev) : F.gb;
    F.scPrev = s->PrevView();
    F.cur.A = R->A[R->currIdx].data(); F.cur.B = R->B[R->currIdx].data(); F.cur.C = R->C[R->currIdx].data();
    F.prev.A = R->A[1 - R->currIdx].data(); F.prev.B = R->B[1 - R->currIdx].data(); F.prev.C = R->C[1 - R->currIdx].data();
    F.target = R->target.data(); F.finalRGBA = finalRGBA;
    F.ox0 = 0; F.oy0 = 0; F.ow = W; F.oh = H;
    SkyParams& prm = F.prm;
    prm.M_max_sky = params->m_max_temporal; prm.M_max_sun = params->m_max_spatial; prm.alpha_min = params->alpha_min;
    prm.accumulate = (g.accumulate && g.camera_static) ? 1u : 0u;
    prm.doTemporal = (R->temporalValid && (params->flags & ZR_IND_TEMPORAL_RESAMPLE) && prev) ? 1u : 0u;
    prm.doSpatial = (prm.doTemporal && (params->flags & ZR_IND_SPATIAL_RESAMPLE)) ? 1u : 0u;
    prm.writeReservoirs = (prm.doTemporal || !R->temporalValid) ? 1u : 0u;
    zr::StackEntry stackMem[zr::kTravStack]; zr::TravStack stack; stack.lds = nullptr; stack.stride = 0; stack.mem = stackMem;
    for (uint32_t y = 0; y < H; y++) for (uint32_t x = 0; x < W; x++) TemporalPixel(F, g, x, y, stack, cnt);
    if (prm.doSpatial) for (uint32_t y = 0; y < H; y++) for (uint32_t x = 0; x < W; x++) SpatialPixel(F, g, x, y, stack, cnt);
    if (counters) { counters->n_closest = cnt[0]; counters->n_shadow = cnt[1]; }
    R->temporalValid = true;
    R->currIdx = 1 - R->currIdx;
}
// plane 0 = A, 1 = B, 2 = C of the set written by the last frame, 3 = target
int zhx_sdi_read_plane(const HxSdi* R, int plane, void* out)
{
    const int last = 1 - R->currIdx;
    if (plane == 0) std::memcpy(out, R->A[last].data(), R->A[last].size());
    else if (plane == 1) std::memcpy(out, R->B[last].data(), R->B[last].size() * 2);
    else if (plane == 2) std::memcpy(out, R->C[last].data(), R->C[last].size() * 4);
    else std::memcpy(out, R->target.data(), R->target.size() * sizeof(F4));
    return 0;
}
// function-level probes shared with the reference build and the oracle (hx_kat.h)
void zhx_kat_sampling(const float* in, float* out, uint32_t n) { hxkat::Sampling_(in, out, n); }
void zhx_kat_math(const float* in, float* out, uint32_t n) { hxkat::Math_(in, out, n); }
void zhx_kat_rt(const float* in, float* out, uint32_t n) { hxkat::RT_(in, out, n); }
void zhx_kat_bsdf(const uint16_t* rho, const uint32_t* rho_dim, const float* in, float* out, uint32_t n) { hxkat::BSDF_(rho, rho_dim, in, out, n); }

} // extern "C"
