// zr_dev_scene.h -- device view of the scene, BVH traversal, hit reconstruction, material fetch, light sampling.
//
// Replaces, for the HIP path, what the reference gets from the D3D12 driver (TLAS/BLAS + inline RayQuery,
// Source/ZetaRenderPass/Common/RayQuery.hlsli:42-53,168-179,317-331,372-396) with an explicit BVH4 over world-space
// triangles in HBM, and restates the shader-side code around it:
//   RayQuery.hlsli:15-144  (Hit::FindClosest vertex fetch / TRS / tri differentials / PCG3d ID)
//   RayQuery.hlsli:452-524 (GetMaterialData)      Material.h:268-417 (accessors)
//   LightSource.hlsli:48-137, 202-223 (alias table draw, emissive triangle decode/sample, Le)
//
// HBM layout:
//   nodes : 128 B each = BVH4 node {lo.x[4], lo.y[4], lo.z[4], hi.x[4], hi.y[4], hi.z[4], child[4], pad[4]}; child >= 0x80000000
//           is a leaf: bits 0..2 = count-1, bits 3..30 = first triangle slot; 0xffffffff = empty slot
//   tris  : 48 B each in leaf order = {v0 xyz, globalTriIdx | e1 xyz, subgroup mask | e2 xyz, meshIdx}
//   triMeta : 8 B per *global* triangle = {meshIdx, primIdx} (hit reconstruction)
#pragma once
#include "zr_dev_bsdf.h"
#include "zr_sky.h"
#include "../../include/zr_texture.h"
#include "../../include/zr_intersect.h"

namespace zr {

struct BvhNode { float lmin[3], lmax[3], rmin[3], rmax[3]; uint32_t left, right, pad0, pad1; };    // builder's BVH2 (host only)
// device node, 64 B = four 16-byte loads per lane: the child boxes are 8-bit offsets from the node's own min corner in
// units of a per-axis power of two (plane = fma(q, 2^(e-127), origin), one rounding; the builder rounds q outwards
// against exactly that expression, so the decoded box always contains the child).  Byte k of a q word = child k.
struct Bvh4Node
{
    float ox, oy, oz; uint32_t exps;            // exps: bits 0..7 / 8..15 / 16..23 = biased exponent of the x / y / z scale
    uint32_t child[4];
    uint32_t qlox, qloy, qloz, qhix;
    uint32_t qhiy, qhiz, pad0, pad1;
};
// 48 B, three 16-byte loads.  mask = the instance's ZR_SUBGROUP_* / ZR_INSTANCE_NON_OPAQUE bits; id = the reference's hashed triangle ID
// (TriID(mesh, prim)), stored so that rays which must ignore one triangle (approximate visibility segments) compare one word per candidate
// instead of fetching TriMeta and hashing
struct BvhTri { float v0[3]; uint32_t gidx; float e1[3]; uint32_t mask; float e2[3]; uint32_t id; };
struct TriMeta { uint32_t mesh, prim; };

static constexpr uint32_t kLeafBit = 0x80000000u;
static constexpr uint32_t kInvalidTri = 0xffffffffu;
static constexpr uint32_t kEmptyChild = 0xffffffffu;      // unused child slot of a Bvh4Node
static constexpr uint32_t kWholeSceneLeaf = 0xfffffffeu;  // traversal root of a scene without nodes

struct SceneView
{
    const zr_vertex* vertices;
    const uint32_t* indices;
    const zr_mesh_instance* instances;
    const zr_material* materials;
    const zr_emissive_triangle* emissives;
    const zr_alias_entry* alias;
    const zr_presampled_tri* sampleSets;   // K3 output: numSampleSets x sampleSetSize (null until a PRELIGHTING pass presampled)
    uint32_t sampleSetSize;
    SkyLutView sky;                        // K17 output (null until a SKY pass rendered)
    const zr_voxel_sample* lvg;            // K4 output: lvgDim.x * y * z voxels x 64 samples (null unless PRELIGHTING built it)
    uint32_t lvgDim[3]; float lvgExtents[3]; float lvgOffsetY;
    const Bvh4Node* nodes;
    const BvhTri* tris;
    const TriMeta* triMeta;
    RhoView rho;
    // material texture heap (zr_wire.h zr_texture_desc, zr_texture.h) + the four descriptor-table offsets of the frame
    // constants (FrameConstants.h:31-34), latched from the cb of the zr_pass_render call that launches the kernel
    zr_tex_heap tex;
    uint32_t baseColorMapsOffset, normalMapsOffset, mrMapsOffset, emissiveMapsOffset;
    uint32_t texFilter;      // zr_params.tex_filter of the pass that launches the kernel (cb_ReSTIR_*::TexFilterDescHeapIdx); ZR_TEX_FILTER_*
    uint32_t numEmissives;
    // material class of the scene: every material is an opaque, uncoated, non-metallic dielectric and there is no texture heap (zr_api.hip MaterialsArePlain).
    // The host leaves it 0; kernels of the PLAIN permutation -- which the host launches only for such scenes, zr_api.hip PlainClass -- overwrite it with their
    // template constant, so that InitSurface(..., plain) folds (zr_dev_bsdf.h).  The host executor of the tests sets it through zhx_set_material_class.
    uint32_t plain;
    uint32_t numNodes;       // 0 => single leaf covering tris[0 .. numTris)
    uint32_t numTris;
#ifdef ZR_PROF
    unsigned long long* prof;     // -DZR_PROF builds only (scripts/gpu.sh prof): wave-cycle counters per kernel / section
#endif
};

// Section timers of the -DZR_PROF measurement build (scripts/gpu.sh prof): wave cycles (s_memtime) between construction and
// destruction, summed per wave in LDS (one lane, no atomics on the hot path) and flushed to sc.prof[32 * kernel + i] when
// the kernel ends.  Empty in the product build.
#if defined(ZR_PROF) && defined(__HIP_DEVICE_COMPILE__)
__shared__ unsigned long long zrProfAcc[4 * 32];
__device__ __forceinline__ void ProfAdd(int i, unsigned long long v)
{
    const unsigned long long m = __ballot(1);
    if (__lane_id() == (uint32_t)(__ffsll((long long)m) - 1)) zrProfAcc[(threadIdx.x >> 6) * 32 + i] += v;
}
struct ProfScope
{
    int i; unsigned long long t0;
    __device__ __forceinline__ ProfScope(int i_) : i(i_), t0(__builtin_readcyclecounter()) {}
    __device__ __forceinline__ ~ProfScope() { ProfAdd(i, __builtin_readcyclecounter() - t0); }
};
struct ProfKernel
{
    unsigned long long* dst; unsigned long long t0;
    __device__ __forceinline__ ProfKernel(unsigned long long* d) : dst(d), t0(__builtin_readcyclecounter())
    { if ((threadIdx.x & 63u) < 32u) zrProfAcc[(threadIdx.x >> 6) * 32 + (threadIdx.x & 63u)] = 0; }
    __device__ __forceinline__ ~ProfKernel()
    {
        ProfAdd(0, __builtin_readcyclecounter() - t0);
        if ((threadIdx.x & 63u) < 32u) atomicAdd(dst + (threadIdx.x & 63u), zrProfAcc[(threadIdx.x >> 6) * 32 + (threadIdx.x & 63u)]);
    }
};
#define ZR_PROF_SCOPE(i) ProfScope zrProfScope##i(i)
#define ZR_PROF_KERNEL(sc, k) ProfKernel zrProfKernel((sc).prof + 32 * (k))
#else
#define ZR_PROF_SCOPE(i)
#define ZR_PROF_KERNEL(sc, k)
#endif
enum { ZRP_KERNEL = 0, ZRP_TRAV, ZRP_TRAV_CALLS, ZRP_RAYS, ZRP_NODE_ITERS, ZRP_NODE_LANES, ZRP_TRI_ITERS, ZRP_TRI_LANES, ZRP_MATERIAL, ZRP_NEE, ZRP_BSDF, ZRP_MISC0, ZRP_MISC1, ZRP_MISC2, ZRP_MISC3, ZRP_MISC4, ZRP_STEALS, ZRP_STEAL_PAIRS };

struct RawHit { float t, u, v; uint32_t tri; };     // tri = global triangle index, kInvalidTri on miss

// hashed triangle ID of the reference: PCG3d(GeometryIndex = meshIdx, InstanceID = 0, PrimitiveIndex).x
ZR_HD uint32_t TriID(uint32_t meshIdx, uint32_t primIdx)
{ uint32_t kx = meshIdx, ky = 0, kz = primIdx; zr_pcg3d(&kx, &ky, &kz); return kx; }

// TestOpacity, GBufferRT_Inline.hlsl:37-70: alpha test of a candidate hit on non-opaque geometry (primary rays only;
// g_samLinearWrap at mip 0).  true = commit the candidate.
ZR_HD bool TestOpacity(const SceneView& sc, uint32_t meshIdx, uint32_t primIdx, float bu, float bv)
{
    const zr_mesh_instance& md = sc.instances[meshIdx];
    const float alphaFactor = zr_div255((float)(md.alpha_factor_cutoff & 0xffu));      // Math::UnpackRG
    const float cutoff = zr_div255((float)(md.alpha_factor_cutoff >> 8));
    if (cutoff == 1.0f) return false;
    float alpha = alphaFactor;
    if (md.base_color_tex != 0xffffu)
    {
        const uint32_t tri = primIdx * 3 + md.base_idx_offset;
        const zr_vertex& V0 = sc.vertices[sc.indices[tri] + md.base_vtx_offset];
        const zr_vertex& V1 = sc.vertices[sc.indices[tri + 1] + md.base_vtx_offset];
        const zr_vertex& V2_ = sc.vertices[sc.indices[tri + 2] + md.base_vtx_offset];
        const float u = V0.uv[0] + bu * (V1.uv[0] - V0.uv[0]) + bv * (V2_.uv[0] - V0.uv[0]);
        const float v = V0.uv[1] + bu * (V1.uv[1] - V0.uv[1]) + bv * (V2_.uv[1] - V0.uv[1]);
        float c[4];
        zr_tex_sample_level(&sc.tex, sc.baseColorMapsOffset + md.base_color_tex, u, v, 0.0f, c);
        alpha *= c[3];
    }
    if (alpha < cutoff) return false;
    return true;
}

struct TravStack;
ZR_HD BvhTri FetchTri(const SceneView& sc, uint32_t i, const TravStack* st);
ZR_HD void IntersectTri(const SceneView& sc, uint32_t i, V3 o, V3 d, float tmin, float tmax,
    uint32_t mask, RawHit& best, bool filterID = false, uint32_t ignoreID = 0, bool alphaTest = false, const TravStack* st = nullptr)
{
    const BvhTri T = FetchTri(sc, i, st);
    if (!(T.mask & mask)) return;
    // (before the intersection test on purpose: behind it -- hits only -- every traversal kernel got 8-12 % slower on the atrium, measured A/B in one run)
    if (filterID && T.id == ignoreID) return;
    float t, u, v;
    if (zr_ray_tri(o.x, o.y, o.z, d.x, d.y, d.z, T.v0[0], T.v0[1], T.v0[2], T.e1[0], T.e1[1], T.e1[2],
            T.e2[0], T.e2[1], T.e2[2], tmin, tmax, &t, &u, &v))
    {
        if (alphaTest && (T.mask & ZR_INSTANCE_NON_OPAQUE))
        { const TriMeta tm = sc.triMeta[T.gidx]; if (!TestOpacity(sc, tm.mesh, tm.prim, u, v)) return; }
        // closest hit; equal t goes to the smaller global triangle index (ABI tie-break)
        if (best.tri == kInvalidTri || t < best.t || (t == best.t && T.gidx < best.tri))
        { best.t = t; best.u = u; best.v = v; best.tri = T.gidx; }
    }
}
ZR_HD void IntersectLeaf(const SceneView& sc, uint32_t first, uint32_t count, V3 o, V3 d, float tmin, float tmax,
    uint32_t mask, RawHit& best, bool filterID = false, uint32_t ignoreID = 0, bool alphaTest = false)
{
    for (uint32_t i = first; i < first + count; i++) IntersectTri(sc, i, o, d, tmin, tmax, mask, best, filterID, ignoreID, alphaTest);
}

// Stack-based BVH4 traversal, written as an explicit state machine so that a kernel can either run it to completion
// (Traverse, used inline by the ReSTIR kernels) or advance many rays one step at a time and refill finished lanes
// (k_trace).  `stack` is this lane's private stack of (child, entry distance) pairs.  Children
// are visited near-to-far and a popped child whose entry distance lies beyond the current closest hit is skipped; none
// of that changes the result (closest hit + index tie-break / any hit are order independent, zr_intersect.h).
static constexpr int kTravStack = 64;                     // entries; the host checks the built tree against it
static constexpr int kTravStackWords = 2 * kTravStack;
#ifndef ZR_TRI_PHASE_WHOLE_LEAF
#define ZR_TRI_PHASE_WHOLE_LEAF 1
#endif
#ifndef ZR_VOTE_WN
#define ZR_VOTE_WN 1
#define ZR_VOTE_WT 1
#endif
#ifndef ZR_TRAV_LDS_ENTRIES
#define ZR_TRAV_LDS_ENTRIES 8
#endif
// intra-wave work stealing inside Traverse (see TraverseDyn): idle lanes take the top stack entry of busy lanes.  Bit-exact (the GPU parity
// suite passes with it) and 23 % fewer vote iterations per call, but SLOWER on every workload measured (ReSTIR PT Cornell
// 2.28 -> 2.65 ms, atrium 17.8 -> 20.7 ms, K9 trace 5.19 -> 5.44 ms), so it is compiled out; -DZR_STEAL=1 builds it (scripts/gpu_steal.sh [rounds 1-4: git history up to 31e92fa; today scripts/gpu.sh ab]).
#ifndef ZR_STEAL
#define ZR_STEAL 0
#endif
#ifndef ZR_STEAL_MIN_IDLE
#define ZR_STEAL_MIN_IDLE 16
#endif
#ifndef ZR_STEAL_MIN_PAIRS
#define ZR_STEAL_MIN_PAIRS 2
#endif
#ifndef ZR_STEAL_COOLDOWN
#define ZR_STEAL_COOLDOWN 1
#endif
static constexpr int kTravLdsEntries = ZR_TRAV_LDS_ENTRIES;                 // device: the first entries live in LDS, deeper ones in scratch

// One lane's traversal stack.  Device kernels keep the bottom kTravLdsEntries entries in LDS (entry e of lane l at
// lds[e * stride + l]: conflict-free, and off the vector-memory path that the node / triangle fetches saturate) and the
// rest in scratch; the host executor keeps everything in `mem`.
struct StackEntry { uint32_t child; float t; };
#ifdef __HIP_DEVICE_COMPILE__
#define ZR_LDS_AS __attribute__((address_space(3)))
#define ZR_PRIVATE_AS __attribute__((address_space(5)))
#else
#define ZR_LDS_AS
#define ZR_PRIVATE_AS
#endif
// `aux` (device, ZR_STEAL): this wave's work-stealing region in LDS -- 64 x u64 merge keys, 64 x (t, u, v) payloads, 64 x u32 donor lanes
// `cache` (device, -DZR_NODE_CACHE=N): the first N nodes of the tree -- its top levels, nodes are numbered breadth-first -- copied into LDS by the
// block (north_star's "LDS-staged node cache"); measured, DESIGN 5.7
// Measured (MI355X, 1080p, N = 64 = 4 KB per block, later 32; scripts/gpu_r03_k11.sh [rounds 1-4: git history up to 31e92fa; today scripts/gpu.sh ab] with a -DZR_NODE_CACHE=64 build of every kernel): K11 on the 380k-triangle
// atrium 8.37 -> 7.63 ms; the reconnect kernels get slower with it (K14 3.30 -> 3.43 ms, Cornell 0.508 -> 0.538 ms: their traversals are a quarter
// of the kernel and the fill + the extra branch cost more than the top-level hits save); Cornell's tree has fewer than N nodes.  So only the
// large-scene K11 (k_rpt_pathtrace_w4) fills it; kernels that do not fill keep cache == nullptr, a compile-time constant that folds the branch away.
// (size: 32 nodes = the top three levels measure like 64 -- 7.64 / 7.62 ms; 128 nodes cost K11 occupancy through LDS: 10.2 ms; one 256-node cache
// per 256-thread block: 7.94 ms)
#ifndef ZR_NODE_CACHE
#define ZR_NODE_CACHE 32
#endif
struct alignas(16) NodeQuad { uint32_t x, y, z, w; };
struct TravStack { ZR_LDS_AS StackEntry* lds; uint32_t stride; ZR_PRIVATE_AS StackEntry* mem; uint32_t* aux = nullptr; const ZR_LDS_AS NodeQuad* cache = nullptr;
    uint32_t cacheNodes = 0;      // nodes 0 .. cacheNodes - 1 are in `cache` (<= ZR_NODE_CACHE; a scene with fewer nodes is cached whole)
    const ZR_LDS_AS NodeQuad* triCache = nullptr; uint32_t cacheTris = 0;       // leaf-order triangles 0 .. cacheTris - 1 (3 quads each) in LDS: tiny scenes
    // weight of a lane waiting at a leaf in the phase vote of TraverseDyn (a lane waiting at an inner node weighs ZR_VOTE_WN = 1): 1 = plain majority.  The spatial
    // reconnect kernel sets 2 -- its dense calls (60 of 64 lanes issue a ray) finish sooner when the leaf phase runs as soon as a third of the lanes wait for it:
    // k_rpt_stc 0.509 -> 0.490 ms on the Cornell frame, 2.68 -> 2.62 ms on the atrium; K11's sparse calls get SLOWER that way (atrium 7.12 -> 8.8 ms) and keep 1
    // (profiles/r06i_ab_vote_weights.txt; tools/bvh_quality.py replays the schedule on the host)
    uint32_t voteTri = ZR_VOTE_WT; };
static constexpr uint32_t kStealAuxWords = 64 * 2 + 64 * 3 + 64;

// triangle i of the leaf order: from the block's LDS copy when the scene is cached whole (tiny scenes, ZR_SCENE_CACHE_FILL), else from HBM / L2
ZR_HD BvhTri FetchTri(const SceneView& sc, uint32_t i, const TravStack* st)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if (st != nullptr && st->triCache != nullptr && i < st->cacheTris)
    {
        const ZR_LDS_AS NodeQuad* c = st->triCache + 3u * i;
        BvhTri T;
        T.v0[0] = zr_asfloat(c[0].x); T.v0[1] = zr_asfloat(c[0].y); T.v0[2] = zr_asfloat(c[0].z); T.gidx = c[0].w;
        T.e1[0] = zr_asfloat(c[1].x); T.e1[1] = zr_asfloat(c[1].y); T.e1[2] = zr_asfloat(c[1].z); T.mask = c[1].w;
        T.e2[0] = zr_asfloat(c[2].x); T.e2[1] = zr_asfloat(c[2].y); T.e2[2] = zr_asfloat(c[2].z); T.id = c[2].w;
        return T;
    }
#endif
    return sc.tris[i];
}
// (member-wise accesses: copying a whole StackEntry through an address-space-qualified pointer would go through a generic
// pointer, and ROCm 7.2's gfx950 backend rejects the aperture check it emits for that cast)
ZR_HD void StackWrite(const TravStack& st, int e, uint32_t c, float t)
{
#ifdef __HIP_DEVICE_COMPILE__
    if (e < kTravLdsEntries) { ZR_LDS_AS StackEntry* p = st.lds + (uint32_t)e * st.stride; p->child = c; p->t = t; return; }
    e -= kTravLdsEntries;
#endif
    st.mem[e].child = c; st.mem[e].t = t;
}
ZR_HD void StackRead(const TravStack& st, int e, uint32_t& c, float& t)
{
#ifdef __HIP_DEVICE_COMPILE__
    if (e < kTravLdsEntries) { const ZR_LDS_AS StackEntry* p = st.lds + (uint32_t)e * st.stride; c = p->child; t = p->t; return; }
    e -= kTravLdsEntries;
#endif
    c = st.mem[e].child; t = st.mem[e].t;
}

struct TravState
{
    V3 o, d; float idx, idy, idz, tmin, tmax;
    RawHit best;
    uint32_t cur, mask, ignoreID; int sp; bool filterID;
};

ZR_HD void TravInit(const SceneView& sc, TravState& s, V3 o, V3 d, float tmin, float tmax, uint32_t mask, bool filterID, uint32_t ignoreID)
{
    s.o = o; s.d = d; s.tmin = tmin; s.tmax = tmax; s.mask = mask; s.filterID = filterID; s.ignoreID = ignoreID;
    s.best.t = tmax; s.best.u = 0; s.best.v = 0; s.best.tri = kInvalidTri;
    s.idx = zr_safe_rcp_dir(d.x); s.idy = zr_safe_rcp_dir(d.y); s.idz = zr_safe_rcp_dir(d.z);
    s.sp = 0;
    // no nodes: a single leaf covering all triangles (tiny scenes)
    s.cur = sc.numNodes ? 0u : kWholeSceneLeaf;
}

// pops the next child still worth visiting; false when the stack ran empty
ZR_HD bool TravPop(TravState& s, const TravStack& stack)
{
    while (s.sp > 0)
    {
        --s.sp;
        uint32_t c; float t;
        StackRead(stack, s.sp, c, t);
        // same condition as re-running zr_ray_box with the current best t
        if (t <= s.best.t * 1.0000003576278687f) { s.cur = c; return true; }
    }
    return false;
}

#define ZR_TRAV_CSWAP(a, b) { const bool sw = t##b < t##a; const float tt = sw ? t##a : t##b; t##a = sw ? t##b : t##a; t##b = tt; \
    const uint32_t cc = sw ? c##a : c##b; c##a = sw ? c##b : c##a; c##b = cc; }

// inner node s.cur: tests the 4 child boxes, pushes the far hits and returns the nearest one (kEmptyChild: none hit)
// unordered (any-hit rays, -DZR_ANYHIT_UNSORTED=1): the children need no near-to-far order -- "some hit" is order-free and the entry-distance cull of TravPop never
// triggers (best.t stays tmax until the ray ends) -- so the sorting network is skipped: the first child hit is visited, the others pushed
#ifndef ZR_ANYHIT_UNSORTED
#define ZR_ANYHIT_UNSORTED 0
#endif
ZR_HD uint32_t TravNode(const SceneView& sc, TravState& s, const TravStack& stack, bool unordered = false)
{
#if ZR_NODE_CACHE && defined(__HIP_DEVICE_COMPILE__)
    Bvh4Node n;
    if (stack.cache != nullptr && s.cur < stack.cacheNodes)
    {
        const ZR_LDS_AS NodeQuad* c = stack.cache + 4u * s.cur;
        NodeQuad a, b, cc, d;
        a.x = c[0].x; a.y = c[0].y; a.z = c[0].z; a.w = c[0].w; b.x = c[1].x; b.y = c[1].y; b.z = c[1].z; b.w = c[1].w;
        cc.x = c[2].x; cc.y = c[2].y; cc.z = c[2].z; cc.w = c[2].w; d.x = c[3].x; d.y = c[3].y; d.z = c[3].z; d.w = c[3].w;
        n.ox = zr_asfloat(a.x); n.oy = zr_asfloat(a.y); n.oz = zr_asfloat(a.z); n.exps = a.w;
        n.child[0] = b.x; n.child[1] = b.y; n.child[2] = b.z; n.child[3] = b.w;
        n.qlox = cc.x; n.qloy = cc.y; n.qloz = cc.z; n.qhix = cc.w; n.qhiy = d.x; n.qhiz = d.y; n.pad0 = 0; n.pad1 = 0;
    }
    else n = sc.nodes[s.cur];
#else
    const Bvh4Node n = sc.nodes[s.cur];
#endif
    const float inf = zr_asfloat(0x7f800000u);
    const float sx = zr_asfloat((n.exps & 0xffu) << 23), sy = zr_asfloat(((n.exps >> 8) & 0xffu) << 23), sz = zr_asfloat(((n.exps >> 16) & 0xffu) << 23);
    uint32_t c0 = n.child[0], c1 = n.child[1], c2 = n.child[2], c3 = n.child[3];
    float t0, t1, t2, t3;
    // cull against the current best t (inclusive + widened, so equal-t candidates for the tie-break are visited)
#define ZR_Q(w, k) ((float)(((w) >> (8 * (k))) & 0xffu))
    // (all four slots are tested unconditionally -- an empty slot decodes to a harmless box -- to keep the phase branch-free.)
    // Per axis the ray's direction sign says which quantised plane is the entry and which the exit, so the words are swapped once per node
    // instead of taking min / max of the two plane distances per child: the same tn / tf as zr_ray_box_native, 18 fewer VALU ops per node.
    const bool ngx = s.idx < 0.0f, ngy = s.idy < 0.0f, ngz = s.idz < 0.0f;
    const uint32_t qnx = ngx ? n.qhix : n.qlox, qfx = ngx ? n.qlox : n.qhix;
    const uint32_t qny = ngy ? n.qhiy : n.qloy, qfy = ngy ? n.qloy : n.qhiy;
    const uint32_t qnz = ngz ? n.qhiz : n.qloz, qfz = ngz ? n.qloz : n.qhiz;
    const float tfmax = s.best.t;
#define ZR_TRAV_BOX(k) { \
        const float nx = (zr_fma(ZR_Q(qnx, k), sx, n.ox) - s.o.x) * s.idx, ny = (zr_fma(ZR_Q(qny, k), sy, n.oy) - s.o.y) * s.idy, nz = (zr_fma(ZR_Q(qnz, k), sz, n.oz) - s.o.z) * s.idz; \
        const float fx = (zr_fma(ZR_Q(qfx, k), sx, n.ox) - s.o.x) * s.idx, fy = (zr_fma(ZR_Q(qfy, k), sy, n.oy) - s.o.y) * s.idy, fz = (zr_fma(ZR_Q(qfz, k), sz, n.oz) - s.o.z) * s.idz; \
        const float tn = __builtin_fmaxf(__builtin_fmaxf(nx, ny), __builtin_fmaxf(nz, s.tmin)); \
        const float tf = __builtin_fminf(__builtin_fminf(fx, fy), __builtin_fminf(fz, tfmax)) * 1.0000003576278687f; \
        const bool ok = (tn <= tf) & (c##k != kEmptyChild); t##k = ok ? tn : inf; c##k = ok ? c##k : kEmptyChild; }
    ZR_TRAV_BOX(0) ZR_TRAV_BOX(1) ZR_TRAV_BOX(2) ZR_TRAV_BOX(3)
#undef ZR_TRAV_BOX
#undef ZR_Q
#if ZR_ANYHIT_UNSORTED
    if (unordered)
    {
        uint32_t next = c0;
        if (c1 != kEmptyChild) { if (next != kEmptyChild) StackWrite(stack, s.sp++, c1, t1); else next = c1; }
        if (c2 != kEmptyChild) { if (next != kEmptyChild) StackWrite(stack, s.sp++, c2, t2); else next = c2; }
        if (c3 != kEmptyChild) { if (next != kEmptyChild) StackWrite(stack, s.sp++, c3, t3); else next = c3; }
        return next;
    }
#endif
    // sorting network: near to far, misses (t = inf) last
    ZR_TRAV_CSWAP(0, 1) ZR_TRAV_CSWAP(2, 3) ZR_TRAV_CSWAP(0, 2) ZR_TRAV_CSWAP(1, 3) ZR_TRAV_CSWAP(1, 2)
    if (c3 != kEmptyChild) StackWrite(stack, s.sp++, c3, t3);
    if (c2 != kEmptyChild) StackWrite(stack, s.sp++, c2, t2);
    if (c1 != kEmptyChild) StackWrite(stack, s.sp++, c1, t1);
    return c0;
}
#undef ZR_TRAV_CSWAP

// one step: a leaf (all its triangles) or an inner node (4 box tests).  Returns true when the ray is finished.
ZR_HD bool TravStep(const SceneView& sc, TravState& s, const TravStack& stack, bool anyHit, bool alphaTest = false)
{
    if (s.cur & kLeafBit)
    {
        uint32_t first = (s.cur & 0x7fffffffu) >> 3, count = (s.cur & 7u) + 1u;
        if (s.cur == kWholeSceneLeaf) { first = 0; count = sc.numTris; }
        IntersectLeaf(sc, first, count, s.o, s.d, s.tmin, s.tmax, s.mask, s.best, s.filterID, s.ignoreID, alphaTest);
        if (anyHit && s.best.tri != kInvalidTri) return true;
        return !TravPop(s, stack);
    }
    const uint32_t next = TravNode(sc, s, stack);
    if (next == kEmptyChild) return !TravPop(s, stack);
    s.cur = next;
    return false;
}

// Device scheduling of the same state machine.  The lanes of a wave that are inside Traverse together vote each
// iteration for one of two phases -- "inner node" (4 box tests) or "one triangle of the current leaf" -- and the wave
// executes only the phase with more takers; the others keep their state and wait.  A phase costs the wave the same
// whether 1 or 64 lanes take part, so this is what raises the fraction of useful lanes (PMC: 22 % with every lane
// running its own node / leaf sequence).  Results cannot depend on the schedule (see above).
struct TravLane { uint32_t triCur, triEnd; bool done; };
ZR_HD void TravEnter(const SceneView& sc, TravState& s, TravLane& L, uint32_t c)
{
    if (c & kLeafBit)
    {
        uint32_t first = (c & 0x7fffffffu) >> 3, count = (c & 7u) + 1u;
        if (c == kWholeSceneLeaf) { first = 0; count = sc.numTris; }
        L.triCur = first; L.triEnd = first + count;
    }
    else s.cur = c;
}
ZR_HD void TravPopEnter(const SceneView& sc, TravState& s, TravLane& L, const TravStack& stack)
{
    if (TravPop(s, stack)) TravEnter(sc, s, L, s.cur);
    else L.done = true;
}
ZR_HD void TravNodePhase(const SceneView& sc, TravState& s, TravLane& L, const TravStack& stack, bool unordered = false)
{
    const uint32_t next = TravNode(sc, s, stack, unordered);
    if (next == kEmptyChild) TravPopEnter(sc, s, L, stack);
    else TravEnter(sc, s, L, next);
}
ZR_HD void TravTriPhase(const SceneView& sc, TravState& s, TravLane& L, const TravStack& stack, bool anyHit, bool alphaTest = false)
{
#if ZR_TRI_PHASE_WHOLE_LEAF
    // leaves hold at most two triangles (zr_bvh.h): both in one phase
    if (L.triEnd - L.triCur <= 2u)
    {
        IntersectTri(sc, L.triCur, s.o, s.d, s.tmin, s.tmax, s.mask, s.best, s.filterID, s.ignoreID, alphaTest, &stack);
        if (L.triCur + 1u < L.triEnd) IntersectTri(sc, L.triCur + 1u, s.o, s.d, s.tmin, s.tmax, s.mask, s.best, s.filterID, s.ignoreID, alphaTest, &stack);
        L.triCur = L.triEnd;
        if (anyHit && s.best.tri != kInvalidTri) L.done = true;
        else TravPopEnter(sc, s, L, stack);
        return;
    }
#endif
    IntersectTri(sc, L.triCur, s.o, s.d, s.tmin, s.tmax, s.mask, s.best, s.filterID, s.ignoreID, alphaTest, &stack);
    L.triCur++;
    if (anyHit && s.best.tri != kInvalidTri) { L.done = true; L.triCur = L.triEnd; }
    else if (L.triCur == L.triEnd) TravPopEnter(sc, s, L, stack);
}

#if ZR_STEAL
#define ZR_STEAL_ANYHIT anyH
#else
#define ZR_STEAL_ANYHIT anyHit
#endif
#if defined(__HIP_DEVICE_COMPILE__) && ZR_STEAL
// ---- intra-wave work stealing.  A Traverse call ends when the slowest ray of the wave ends; measured a call runs 2 x the vote
// iterations its average ray needs and a third of the lanes work per iteration.  So lanes whose ray is finished take over pending subtrees of
// the lanes still busy: the donor hands over the TOP entry of its stack (in LDS, where any lane can read it) together with a copy of the ray, both
// go on independently, and the pieces of one ray are merged through a 64-bit LDS atomic min on (t, global triangle index) -- the closest hit
// with the ABI's index tie-break is exactly that minimum, and any-hit is "some piece hit", so the result cannot depend on who traced what.
// (t, u, v) of the winning piece travel beside the key: a piece writes them iff its key is the slot's key after its atomic (the LDS executes a
// wave's instructions in order; equal keys mean the same triangle and the same ray, hence the same u, v).
__device__ __forceinline__ unsigned long long StealKey(const RawHit& b)
{ return ((unsigned long long)zr_asuint(b.t + 0.0f) << 32) | b.tri; }      // + 0: -0 and +0 compare equal in IntersectTri, so they must tie here too
__device__ __forceinline__ void StealPublish(const TravStack& st, uint32_t owner, const RawHit& b)
{
    if (b.tri == kInvalidTri) return;
    unsigned long long* keys = (unsigned long long*)st.aux;
    const unsigned long long key = StealKey(b);
    atomicMin(&keys[owner], key);
    if (((volatile unsigned long long*)keys)[owner] == key)
    { float* pl = (float*)(st.aux + 128) + 3 * owner; pl[0] = b.t; pl[1] = b.u; pl[2] = b.v; }
}
#endif

// alphaTest (primary rays): candidates on ZR_INSTANCE_NON_OPAQUE geometry must pass TestOpacity
ZR_HD RawHit TraverseDyn(const SceneView& sc, V3 o, V3 d, float tmin, float tmax, uint32_t mask, const TravStack& stack, bool anyHit,
    bool filterID = false, uint32_t ignoreID = 0, bool alphaTest = false)
{
    TravState s;
    TravInit(sc, s, o, d, tmin, tmax, mask, filterID, ignoreID);
#ifdef __HIP_DEVICE_COMPILE__
#ifdef ZR_PROF
    ZR_PROF_SCOPE(ZRP_TRAV);
    unsigned long long pNI = 0, pNL = 0, pTI = 0, pTL = 0, pST = 0, pSP = 0;
    ProfAdd(ZRP_TRAV_CALLS, 1); ProfAdd(ZRP_RAYS, __popcll(__ballot(1)));
#endif
    TravLane L; L.triCur = 0; L.triEnd = 0; L.done = false;
    TravEnter(sc, s, L, s.cur);
#if ZR_STEAL
    const uint32_t lane = __lane_id();
    uint32_t owner = lane;          // whose ray this lane is working on
    bool pub = true;                // false: this lane holds an unpublished piece of `owner`'s ray
    bool stolen = false;            // wave-uniform: some piece changed lanes in this call
    int cool = 0;                   // wave-uniform
    bool anyH = anyHit;             // of the ray this lane works on (a wave of k_trace_simple mixes closest-hit and any-hit rays)
#endif
    for (;;)
    {
        const bool atTri = L.triCur < L.triEnd;
        const bool atNode = !L.done && !atTri;
        const uint64_t mNode = __ballot(atNode), mTri = __ballot(atTri);
        if ((mNode | mTri) == 0) break;
#if ZR_STEAL
        if (cool > 0) cool--;
        else
        {
            const uint64_t mIdle = __ballot(L.done);
            if (__popcll(mIdle) >= ZR_STEAL_MIN_IDLE)
            {
                const bool canGive = !L.done && s.sp >= 1 && s.sp <= kTravLdsEntries;
                const uint64_t mDon = __ballot(canGive);
                const uint32_t nPairs = (uint32_t)min(__popcll(mIdle), __popcll(mDon));
                if (nPairs >= ZR_STEAL_MIN_PAIRS)
                {
                    unsigned long long* keys = (unsigned long long*)stack.aux;
                    uint32_t* pairLane = stack.aux + 128 + 192;
                    if (!stolen) { keys[lane] = ~0ull; stolen = true; pub = false; }
                    const uint64_t lt = (1ull << lane) - 1ull;
                    const uint32_t rI = (uint32_t)__popcll(mIdle & lt), rD = (uint32_t)__popcll(mDon & lt);
                    const bool give = canGive && rD < nPairs, take = L.done && rI < nPairs;
                    if (give) pairLane[rD] = lane;
                    // the piece this lane finished goes to its owner's slot before the lane's state is overwritten
                    if (take && !pub) StealPublish(stack, owner, s.best);
                    const int src = take ? (int)((volatile uint32_t*)pairLane)[rI] : (int)lane;
                    const float ox = __shfl(s.o.x, src), oy = __shfl(s.o.y, src), oz = __shfl(s.o.z, src);
                    const float dx = __shfl(s.d.x, src), dy = __shfl(s.d.y, src), dz = __shfl(s.d.z, src);
                    const float tmn = __shfl(s.tmin, src), bt = __shfl(s.best.t, src);
                    const uint32_t msk = __shfl(s.mask, src), ign = __shfl(s.ignoreID, src), own = __shfl(owner, src);
                    const int dsp = __shfl(s.sp, src);
                    const int flg = __shfl((int)s.filterID | ((int)anyH << 1) | ((int)(s.best.tri != kInvalidTri) << 2), src);
                    if (take)
                    {
                        // the piece starts without a hit of its own; the donor's current best t bounds it (hits beyond it cannot win the merge;
                        // zr_ray_tri accepts t < tmax, and an equal-t hit with a smaller index must still get through)
                        TravInit(sc, s, v3(ox, oy, oz), v3(dx, dy, dz), tmn, (flg & 4) ? NextFloat32(bt) : bt, msk, (flg & 1) != 0, ign);
                        anyH = (flg & 2) != 0; owner = own; pub = false;
                        const ZR_LDS_AS StackEntry* e = stack.lds + (uint32_t)(dsp - 1) * stack.stride + (src - (int)lane);
                        const uint32_t c = e->child; const float et = e->t;
                        L.triCur = 0; L.triEnd = 0;
                        if (et <= bt * 1.0000003576278687f) { L.done = false; TravEnter(sc, s, L, c); }
                    }
                    if (give) s.sp--;
                    // any-hit rays: a piece that hit ends the whole ray
                    if (anyH && !L.done && !take && ((volatile unsigned long long*)keys)[owner] != ~0ull) { L.done = true; L.triCur = L.triEnd; }
                    cool = ZR_STEAL_COOLDOWN;
#ifdef ZR_PROF
                    pST++; pSP += nPairs;
#endif
                    continue;
                }
            }
        }
#endif
#ifdef ZR_PROF
        if (ZR_VOTE_WN * __popcll(mNode) >= (int)stack.voteTri * __popcll(mTri)) { pNI++; pNL += __popcll(mNode); } else { pTI++; pTL += __popcll(mTri); }
#endif
        if (ZR_VOTE_WN * __popcll(mNode) >= (int)stack.voteTri * __popcll(mTri)) { if (atNode) TravNodePhase(sc, s, L, stack, ZR_STEAL_ANYHIT); }
        else { if (atTri) TravTriPhase(sc, s, L, stack, ZR_STEAL_ANYHIT, alphaTest); }
    }
#if ZR_STEAL
    if (stolen)
    {
        if (!pub) StealPublish(stack, owner, s.best);
        const unsigned long long key = ((volatile unsigned long long*)stack.aux)[lane];
        if (key == ~0ull) { s.best.t = tmax; s.best.u = 0; s.best.v = 0; s.best.tri = kInvalidTri; }
        else
        {
            const volatile float* pl = (volatile float*)(stack.aux + 128) + 3 * lane;
            s.best.t = pl[0]; s.best.u = pl[1]; s.best.v = pl[2]; s.best.tri = (uint32_t)(key & 0xffffffffull);
        }
    }
#endif
#ifdef ZR_PROF
    ProfAdd(ZRP_NODE_ITERS, pNI); ProfAdd(ZRP_NODE_LANES, pNL); ProfAdd(ZRP_TRI_ITERS, pTI); ProfAdd(ZRP_TRI_LANES, pTL);
    ProfAdd(ZRP_STEALS, pST); ProfAdd(ZRP_STEAL_PAIRS, pSP);
#endif
#else
    while (!TravStep(sc, s, stack, anyHit, alphaTest)) {}
#endif
    return s.best;
}
template<bool AnyHit>
ZR_HD RawHit Traverse(const SceneView& sc, V3 o, V3 d, float tmin, float tmax, uint32_t mask, const TravStack& stack, bool filterID = false, uint32_t ignoreID = 0)
{ return TraverseDyn(sc, o, d, tmin, tmax, mask, stack, AnyHit, filterID, ignoreID); }

// ---- Material.h accessors ----
ZR_HD bool MatDoubleSided(const zr_material& m) { return m.coat_color_flags & (1u << ZR_MAT_DOUBLE_SIDED_BIT); }
ZR_HD bool MatMetallic(const zr_material& m) { return m.coat_color_flags & (1u << ZR_MAT_METALLIC_BIT); }
ZR_HD bool MatTransmissive(const zr_material& m) { return m.coat_color_flags & (1u << ZR_MAT_TRANSMISSIVE_BIT); }
ZR_HD bool MatThinWalled(const zr_material& m) { return m.coat_color_flags & (1u << ZR_MAT_THIN_WALLED_BIT); }
ZR_HD float MatRoughness(const zr_material& m) { return zr_div255((float)((m.mr_tex_spec_roughness_coat_roughness >> 16) & 0xff)); }
ZR_HD float MatCoatRoughness(const zr_material& m) { return zr_div255((float)((m.mr_tex_spec_roughness_coat_roughness >> 24) & 0xff)); }
ZR_HD float MatIOR(const zr_material& m) { return zr_fma(1.5f / 65535.0f, (float)(m.emissive_strength_ior >> 16), kMinIOR); }
ZR_HD float MatCoatIOR(const zr_material& m) { return zr_fma(1.5f / 255.0f, (float)((m.emissive_tex_alpha_cutoff_coat_ior >> 24) & 0xff), kMinIOR); }
ZR_HD float MatTrDepth(const zr_material& m) { return zr_f16_to_f32((uint16_t)(m.normal_tex_tr_depth >> 16)); }
ZR_HD float MatSubsurface(const zr_material& m) { return zr_div255((float)((m.base_color_tex_subsurf_coat_weight >> 16) & 0xff)); }
ZR_HD float MatCoatWeight(const zr_material& m) { return zr_div255((float)((m.base_color_tex_subsurf_coat_weight >> 24) & 0xff)); }
ZR_HD float MatEmissiveStrength(const zr_material& m) { return zr_f16_to_f32((uint16_t)(m.emissive_strength_ior & 0xffff)); }

// ---- hit reconstruction (RayQuery.hlsli:55-131 / 209-289) ----
struct HitInfo { float t; V3 normal; uint32_t ID; uint32_t meshIdx; uint32_t matIdx; V3 dndu, dndv, dpdu, dpdv; V2 uv; };

template<bool WantDiffs>
ZR_HD void FillHit(const SceneView& sc, uint32_t meshIdx, uint32_t primIdx, float bu, float bv, bool wantID, HitInfo& ret, bool currFrame = true)
{
    const zr_mesh_instance& md = sc.instances[meshIdx];
    ret.matIdx = md.mat_idx;
    ret.meshIdx = meshIdx;
    uint32_t tri = primIdx * 3 + md.base_idx_offset;
    const zr_vertex& V0 = sc.vertices[sc.indices[tri] + md.base_vtx_offset];
    const zr_vertex& V1 = sc.vertices[sc.indices[tri + 1] + md.base_vtx_offset];
    const zr_vertex& V2_ = sc.vertices[sc.indices[tri + 2] + md.base_vtx_offset];

    // InCurrFrame == false: previous frame's instance transform (RayQuery.hlsli:75-90)
    V4 q = normalize(DecodeNormalized4(currFrame ? md.rotation : md.prev_rotation));
    const uint16_t* sh = currFrame ? md.scale : md.prev_scale;
    V3 s = v3(zr_f16_to_f32(sh[0]), zr_f16_to_f32(sh[1]), zr_f16_to_f32(sh[2]));

    float tmp = 1 - bu - bv;
    V2 uv = v2(zr_fma(bv, V2_.uv[0], tmp * V0.uv[0]), zr_fma(bv, V2_.uv[1], tmp * V0.uv[1]));
    ret.uv = v2(zr_fma(bu, V1.uv[0], uv.x), zr_fma(bu, V1.uv[1], uv.y));

    V3 v0_n = DecodeOct32(V0.normal), v1_n = DecodeOct32(V1.normal), v2_n = DecodeOct32(V2_.normal);
    V3 hn = mad(bv, v2_n, tmp * v0_n);
    hn = mad(bu, v1_n, hn);
    const V3 scaleInv = v3(1.0f / s.x, 1.0f / s.y, 1.0f / s.z);
    hn = hn * scaleInv;
    hn = RotateVector(hn, q);
    ret.normal = normalize(hn);

    if (WantDiffs)
    {
        V3 trn = v3p(md.translation);
        if (!currFrame) trn = trn - v3(zr_f16_to_f32(md.d_translation[0]), zr_f16_to_f32(md.d_translation[1]), zr_f16_to_f32(md.d_translation[2]));
        V3 v0W = TransformTRS(v3p(V0.pos), trn, q, s);
        V3 v1W = TransformTRS(v3p(V1.pos), trn, q, s);
        V3 v2W = TransformTRS(v3p(V2_.pos), trn, q, s);
        V3 n0W = normalize(RotateVector(v0_n * scaleInv, q));
        V3 n1W = normalize(RotateVector(v1_n * scaleInv, q));
        V3 n2W = normalize(RotateVector(v2_n * scaleInv, q));
        TriDiffs td = ComputeTriDiffs(v0W, v1W, v2W, n0W, n1W, n2W, v2(V0.uv[0], V0.uv[1]), v2(V1.uv[0], V1.uv[1]), v2(V2_.uv[0], V2_.uv[1]));
        ret.dpdu = td.dpdu; ret.dpdv = td.dpdv; ret.dndu = td.dndu; ret.dndv = td.dndv;
    }
    ret.ID = 0xffffffffu;
    if (wantID)
    {
        uint32_t kx = meshIdx, ky = 0, kz = primIdx;   // static BLAS: GeometryIndex = meshIdx, InstanceID = 0
        zr_pcg3d(&kx, &ky, &kz);
        ret.ID = kx;
    }
}

// GetMaterialData, RayQuery.hlsli:452-524 (texture maps not bound: factors only)
// The two TexSampler policies of RayQuery.hlsli:408-450.  Anisotropic = SampleGrad(samp, uv, ddx, ddy) with the sampler the pass's
// TEXTURE_FILTER selects (default ANISOTROPIC_4X, IndirectLighting.h:243; zr_texture.h); isotropic = SampleLevel(g_samLinearWrap, uv,
// log2(max(dd.x * w, dd.y * h))) with dd = uv_grads.xy (the reconnection shift, Shift.hlsli:519).
ZR_HD void SampleMaterialTex(const SceneView& sc, uint32_t tex, bool isotropic, V2 uv, V4 g, float out[4])
{
    if (!isotropic) { zr_tex_sample_grad_filter(&sc.tex, tex, sc.texFilter, uv.x, uv.y, g.x, g.y, g.z, g.w, out); return; }
    const zr_texture_desc& d = sc.tex.descs[tex];
    const float mip = zr_log2(zr_max(g.x * (float)d.width, g.y * (float)d.height));
    zr_tex_sample_level(&sc.tex, tex, uv.x, uv.y, mip, out);
}

// `tex` is a compile-time constant at every device call site (kernels are instantiated per TEXTURED permutation): without
// a texture heap nothing below reads uv_grads, and kernels then do not carry ray differentials at all.
ZR_HD bool GetMaterialData(const SceneView& sc, V3 wo, float eta_curr, HitInfo& hit, Surface& surface, float& eta,
    V4 uv_grads = v4(0, 0, 0, 0), bool tex = false, bool isotropic = false)
{
    const zr_material mat = sc.materials[hit.matIdx];
    const bool hitBackface = dot(wo, hit.normal) < 0;
    eta = kDefaultEtaMat;
    const bool ds = MatDoubleSided(mat);
    if (!ds && hitBackface) return false;
    if (ds && hitBackface)
    {
        hit.normal = hit.normal * -1.0f;
        if (tex) { hit.dndu = hit.dndu * -1.0f; hit.dndv = hit.dndv * -1.0f; }
    }
    V3 baseColor = UnpackRGB8(mat.base_color_factor);
    float metallic = MatMetallic(mat) ? 1.0f : 0.0f;
    float roughness = MatRoughness(mat);
    bool tr = MatTransmissive(mat);
    eta = MatIOR(mat);
    float trDepth = tr ? MatTrDepth(mat) : 0;
    if (tex)
    {
        const uint32_t baseColorTex = mat.base_color_tex_subsurf_coat_weight & 0xffffu;
        const uint32_t mrTex = mat.mr_tex_spec_roughness_coat_roughness & 0xffffu;
        if ((trDepth == 0) && (baseColorTex != ZR_INVALID_TEX))
        {
            float c[4];
            SampleMaterialTex(sc, sc.baseColorMapsOffset + baseColorTex, isotropic, hit.uv, uv_grads, c);
            baseColor = baseColor * v3(c[0], c[1], c[2]);
        }
        if (mrTex != ZR_INVALID_TEX)
        {
            float c[4];
            SampleMaterialTex(sc, sc.mrMapsOffset + mrTex, isotropic, hit.uv, uv_grads, c);
            metallic *= c[0];
            roughness *= c[1];
        }
    }
    float eta_next = eta_curr == kEtaAir ? eta : kEtaAir;
    float subsurface = MatThinWalled(mat) ? zr_round_f16(MatSubsurface(mat)) : 0;
    surface = InitSurface(hit.normal, wo, metallic >= kMinMetalnessMetal, roughness, baseColor, eta_curr, eta_next, tr, trDepth,
        subsurface, MatCoatWeight(mat), UnpackRGB8(mat.coat_color_flags), MatCoatRoughness(mat), MatCoatIOR(mat), sc.plain != 0);
    return true;
}

// ---- emissive triangles (RtCommon.h:66-131, LightSource.hlsli:48-70) ----
ZR_HD bool EmDoubleSided(const zr_emissive_triangle& t) { return t.packed_a & (1u << 25); }
ZR_HD V3 EmV1(const zr_emissive_triangle& t)
{
    V3 d = DecodeUnitVector(v2(zr_div65535((float)t.v0v1[0]), zr_div65535((float)t.v0v1[1])));
    return mad(zr_f16_to_f32(t.edge_lengths[0]), d, v3p(t.vtx0));
}
ZR_HD V3 EmV2(const zr_emissive_triangle& t)
{
    V3 d = DecodeUnitVector(v2(zr_div65535((float)t.v0v2[0]), zr_div65535((float)t.v0v2[1])));
    return mad(zr_f16_to_f32(t.edge_lengths[1]), d, v3p(t.vtx0));
}
// Light::SamplePresampledSet (LightSource.hlsli:99-106) + the decode of the USE_PRESAMPLED_SETS branches
// (ReSTIR_GI_NEE.hlsli:68-85, ReSTIR_PT_NEE.hlsli:217-236)
struct PresampledLight { V3 pos, normal, le; float pdf; uint32_t idx, ID; bool twoSided; };
ZR_HD PresampledLight SamplePresampledSet(const SceneView& sc, uint32_t sampleSetIdx, V3 shadingPos, Rng& rng)
{
    uint32_t u = rng.UniformUintBounded_Faster(sc.sampleSetSize);
    const zr_presampled_tri t = sc.sampleSets[(size_t)sampleSetIdx * sc.sampleSetSize + u];
    PresampledLight r;
    r.pos = v3p(t.pos); r.normal = DecodeOct32(t.normal);
    r.le = v3(zr_f16_to_f32(t.le[0]), zr_f16_to_f32(t.le[1]), zr_f16_to_f32(t.le[2]));
    r.pdf = t.pdf; r.idx = t.idx; r.ID = t.id; r.twoSided = t.two_sided != 0;
    if (r.twoSided && dot(shadingPos - r.pos, r.normal) < 0) r.normal = r.normal * -1.0f;
    return r;
}

// Le_EmissiveTriangle, LightSource.hlsli:202-223 (default sampler g_samPointWrap: the mip-0 texel under texUV)
ZR_HD V2 EmUV(const uint16_t* h) { return v2(zr_f16_to_f32(h[0]), zr_f16_to_f32(h[1])); }
ZR_HD V3 EmLe(const SceneView& sc, const zr_emissive_triangle& t, V2 bary)
{
    V3 le = UnpackRGB8(t.packed_a) * zr_f16_to_f32((uint16_t)(t.packed_b >> 16));
    if (Luminance(le) == 0) return v3(0.0f);
    const uint32_t emissiveTex = t.packed_b & 0xffffu;
    if (emissiveTex != ZR_INVALID_TEX)
    {
        const V2 texUV = (1.0f - bary.x - bary.y) * EmUV(t.uv0) + bary.x * EmUV(t.uv1) + bary.y * EmUV(t.uv2);
        float c[4];
        zr_tex_point(&sc.tex, sc.emissiveMapsOffset + emissiveTex, texUV.x, texUV.y, c);
        le = le * v3(c[0], c[1], c[2]);
    }
    return le;
}

} // namespace zr
