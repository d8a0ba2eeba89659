// zr_kernels.h -- device-side helpers (lane <-> pixel mapping, wave-ballot slot allocation, traversal stack, ray counters) and the
// ReSTIR PT kernel templates, shared by the translation units of libzetaray_amd.so.
//
// The library is built from three translation units so that `make -j` compiles them side by side (one TU took 6.5 minutes):
//   zr_api.hip       host side (C-ABI) + the K1 / K9 / K10 / DI / sky / pre-lighting kernels; the ReSTIR PT kernels are only
//                    *declared* there (extern template below) and launched through their host stubs
//   zr_tu_rpt_a.hip  explicit instantiations of K11 (k_rpt_pathtrace, k_rpt_pathtrace_tex) and K14 (k_rpt_temporal)
//   zr_tu_rpt_b.hip  explicit instantiations of K13 (k_rpt_replay)
//   zr_tu_rpt_d.hip  explicit instantiations of K16 (k_rpt_stc)
//   zr_tu_di.hip     K5 - K8 and K10 (zr_kernels_di.h)
#pragma once
#include <hip/hip_runtime.h>
#include "zr_stages.h"
#include "zr_rpt.h"

using namespace zr;

// ------------------------------------------------------------------------------------------------ device helpers
static constexpr int kBlock = 256;
// threads per block of K11 (k_rpt_pathtrace*): 64 = one block per wave of the 16 x 16 tile.  A block's registers and LDS are released when its
// slowest wave ends, and the waves of K11 run for very different times (path lengths): one-wave blocks measured 1.006 -> 0.947 ms (Cornell) and
// 8.34 -> 8.01 ms (atrium) at 1080p.  K14 / K16 do the same work in every wave and are 2 - 5 % slower that way (0.492 -> 0.518 ms), so they keep
// 256 (scripts/gpu_block.sh [rounds 1-4: git history up to 31e92fa; today scripts/gpu.sh ab]; -DZR_RPT_BLOCK=256 restores one block per tile).
#ifndef ZR_RPT_BLOCK
#define ZR_RPT_BLOCK 64
#endif
static constexpr int kRptBlock = ZR_RPT_BLOCK;
// Occupancy targets of the register-heavy shading kernels, waves per SIMD (the compiler spills a little to reach them).
// Measured on MI355X, Cornell 1080p / 380k-triangle atrium: k_rpt_pathtrace 2 -> 3 waves: 1.45 -> 1.21 ms /
// 14.9 -> 11.8 ms (4 waves: 1.19 / 10.7 ms, but its ~300 B/lane of spills stream 3.4 GB through L2 per launch, PMC); k_rgi 2 -> 4: 2.19 -> 1.68 ms; k_rdi_* 2 -> 3: 0.72 -> 0.67, 0.39 -> 0.33 ms; k_sdi_spatial -> 4: 0.58 ->
// 0.49 ms.  k_rpt_stc, k_pt_shade and k_sdi_temporal got slower with more waves and keep the default.
// Re-measured in round 2 after the traversal changes (2-triangle leaves, whole-leaf triangle phase): k_rpt_pathtrace 3 -> 4 waves 1.013 -> 1.000 ms
// (5: 1.17), k_rpt_temporal default -> 4 waves 0.546 -> 0.536 ms / atrium 3.53 -> 3.27 ms (3: 0.57, 5: 0.70, 6: 0.82), k_rpt_stc 3 / 5 waves 0.77 / 0.72
// against 0.64 at its natural 4, k_rgi 3 / 5 waves 1.37 / 1.45 against 1.30 at 4.  The TEXTURED permutations of the path-tracing kernels want more (K11 6,
// k_rgi_tex 6: texel-gather latency); the reconnect kernels do not even there (textured atrium, 5 / 6 waves: temporal 3.92 / 4.06 against 3.62 ms, K16
// 2.93 / 3.48 against 2.82), nor does the textured K9 shade kernel (4 / 6 waves: 6.0 / 8.7 against 5.2 ms at >= 2).
#define ZR_WAVES(n) __attribute__((amdgpu_waves_per_eu(n, n)))
#define ZR_WAVES_MIN(n) __attribute__((amdgpu_waves_per_eu(n)))
// (with one-wave blocks 3 and 4 waves tie on small scenes -- 0.945 / 0.950 ms Cornell, 5: 1.12 -- and 3 waves spill a third of the bytes (PMC: 0.56 GB per
// launch against 1.44 GB), so scenes whose BVH fits the caches run at 3; large scenes take k_rpt_pathtrace_w4)
// (round 6: on the collapsed trees k_rpt_pathtrace_w4 -- 4 waves + the LDS node cache -- wins at every scene size and is what the host launches, zr_api.hip kLargeSceneNodes)
// Occupancy targets as numbers (`make variant EXTRA=-DZR_WAVES_STC_N=3`: a parenthesised macro value does not survive make + sh quoting).
// The PLAIN permutations (material-class kernels, zr_tu_rpt_e.hip) have their own: a third of the code, other register needs.
#ifndef ZR_WAVES_PATHTRACE_N
#define ZR_WAVES_PATHTRACE_N 3
#endif
#ifndef ZR_WAVES_PATHTRACE_LARGE_N
#define ZR_WAVES_PATHTRACE_LARGE_N 4      // k_rpt_pathtrace_w4: scenes whose BVH does not fit the caches
#endif
#ifndef ZR_WAVES_TEMPORAL_N
#define ZR_WAVES_TEMPORAL_N 4
#endif
// (round 5: with the prepared wo-only terms in Surface the allocator's own choice for k_rpt_stc became 145 VGPRs = 3 waves: 0.686 ms against 0.657 at 4,
// atrium 3.13 against 2.74 -- pinned to the 4 it chose before, profiles/r05_ab_*.json)
#ifndef ZR_WAVES_STC_N
#define ZR_WAVES_STC_N 4
#endif
#ifndef ZR_WAVES_PATHTRACE_PLAIN_N
#define ZR_WAVES_PATHTRACE_PLAIN_N 4      // (Cornell: 0.753 ms at 3, 0.726 at 4, 0.911 at 2)
#endif
#ifndef ZR_WAVES_PATHTRACE_LARGE_PLAIN_N
#define ZR_WAVES_PATHTRACE_LARGE_PLAIN_N 4
#endif
#ifndef ZR_WAVES_TEMPORAL_PLAIN_N
#define ZR_WAVES_TEMPORAL_PLAIN_N 4
#endif
#ifndef ZR_WAVES_STC_PLAIN_N
#define ZR_WAVES_STC_PLAIN_N 4
#endif
// (the attribute's arguments may depend on a template parameter)
#define ZR_WAVES_SEL(plain, nPlain, n) __attribute__((amdgpu_waves_per_eu((plain) ? (nPlain) : (n), (plain) ? (nPlain) : (n))))
#define ZR_WAVES_PATHTRACE ZR_WAVES_SEL(PLAIN, ZR_WAVES_PATHTRACE_PLAIN_N, ZR_WAVES_PATHTRACE_N)
#define ZR_WAVES_PATHTRACE_LARGE ZR_WAVES_SEL(PLAIN, ZR_WAVES_PATHTRACE_LARGE_PLAIN_N, ZR_WAVES_PATHTRACE_LARGE_N)
#define ZR_WAVES_TEMPORAL ZR_WAVES_SEL(PLAIN, ZR_WAVES_TEMPORAL_PLAIN_N, ZR_WAVES_TEMPORAL_N)
#define ZR_WAVES_STC ZR_WAVES_SEL(PLAIN, ZR_WAVES_STC_PLAIN_N, ZR_WAVES_STC_N)
#ifndef ZR_WAVES_RGI
#define ZR_WAVES_RGI ZR_WAVES(4)
#endif
#define ZR_WAVES_RDI_T ZR_WAVES(3)
#define ZR_WAVES_RDI_S ZR_WAVES(3)
#define ZR_WAVES_SDI_S ZR_WAVES(4)
#define ZR_WAVES_SDI_T
static constexpr uint32_t kCounterStride = 64;     // queue counters live 256 B apart: their atomics spread over L2 channels
// this lane's traversal stack: kTravLdsEntries entries in LDS (16 KB per 256-lane block at 8 entries; + 6 KB for the work-stealing slots, zr_dev_scene.h), the rest in scratch
#define ZR_TRAV_STACK_B(name, B) \
    __shared__ StackEntry name##Lds[kTravLdsEntries * (B)]; StackEntry name##Mem[kTravStack - kTravLdsEntries]; \
    __shared__ __attribute__((aligned(8))) uint32_t name##Aux[ZR_STEAL ? kStealAuxWords * ((B) / 64) : 2]; \
    TravStack name; name.lds = (ZR_LDS_AS StackEntry*)name##Lds + threadIdx.x; name.stride = (B); name.mem = (ZR_PRIVATE_AS StackEntry*)name##Mem; \
    name.aux = name##Aux + (ZR_STEAL ? kStealAuxWords * (threadIdx.x / 64) : 0)
#define ZR_TRAV_STACK(name) ZR_TRAV_STACK_B(name, kBlock)
// -DZR_NODE_CACHE=N: the block copies the first N nodes (the top levels: breadth-first numbering) into LDS once; TravNode reads those from there
#if ZR_NODE_CACHE
#define ZR_NODE_CACHE_FILL(name, scene, B) \
    __shared__ uint4 name##NodeCache[4 * ZR_NODE_CACHE]; \
    { const uint32_t nc_ = (scene).numNodes < (uint32_t)ZR_NODE_CACHE ? (scene).numNodes : (uint32_t)ZR_NODE_CACHE; \
      for (uint32_t i_ = threadIdx.x; i_ < 4u * nc_; i_ += (B)) name##NodeCache[i_] = ((const uint4*)(scene).nodes)[i_]; \
      __syncthreads(); name.cache = nc_ ? (const ZR_LDS_AS NodeQuad*)name##NodeCache : nullptr; name.cacheNodes = nc_; }
#else
#define ZR_NODE_CACHE_FILL(name, scene, B)
#endif
// Tiny scenes (Cornell class: <= kSceneCacheNodes nodes and <= kSceneCacheTris triangles, 2 + 4.5 KB): the block copies the WHOLE acceleration
// structure into LDS once -- north_star's "LDS-staged node caches" taken to the scene that fits.  Every traversal iteration is a dependent
// fetch; from LDS it returns in ~60 cycles instead of the ~250 of an L2 hit (the scene is L2- but not L1-resident: scratch and the planes
// stream through the 32 KB L1), and the megakernels are latency-bound at 3 - 4 waves per SIMD.  The host picks the kernel
// permutation per scene (zr_api.hip); larger scenes run the kernels without this code.
#ifndef ZR_SCENE_LDS
#define ZR_SCENE_LDS 0
#endif
static constexpr uint32_t kSceneCacheNodes = 32, kSceneCacheTris = 96;
#define ZR_SCENE_CACHE_FILL(name, scene, B) \
    __shared__ uint4 name##SceneCache[4 * kSceneCacheNodes + 3 * kSceneCacheTris]; \
    { const uint32_t nn_ = (scene).numNodes, nt_ = (scene).numTris; \
      if (nn_ <= kSceneCacheNodes && nt_ <= kSceneCacheTris) { \
        for (uint32_t i_ = threadIdx.x; i_ < 4u * nn_; i_ += (B)) name##SceneCache[i_] = ((const uint4*)(scene).nodes)[i_]; \
        for (uint32_t i_ = threadIdx.x; i_ < 3u * nt_; i_ += (B)) name##SceneCache[4 * kSceneCacheNodes + i_] = ((const uint4*)(scene).tris)[i_]; \
        __syncthreads(); \
        name.cache = nn_ ? (const ZR_LDS_AS NodeQuad*)name##SceneCache : nullptr; name.cacheNodes = nn_; \
        name.triCache = (const ZR_LDS_AS NodeQuad*)name##SceneCache + 4 * kSceneCacheNodes; name.cacheTris = nt_; } }

// one atomic per wave: lanes that `want` a slot get consecutive indices
__device__ __forceinline__ uint32_t AllocSlotWave(uint32_t* counter, bool want)
{
    const uint64_t m = __ballot(want);
    if (m == 0) return 0;
    const uint32_t lane = __lane_id();
    const uint32_t prefix = __popcll(m & ((1ull << lane) - 1ull));
    const int leader = __ffsll((long long)m) - 1;
    uint32_t base = 0;
    if ((int)lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(m));
    base = __shfl(base, leader);
    return base + prefix;
}
// appends this wave's new rays to the queue's compacted ray lists, one list per ray type so that the trace stage's waves
// stay type-uniform: one atomic per wave and type (all 64 lanes must call this)
__device__ __forceinline__ void AppendRays(uint32_t* rayList, uint32_t cap, uint32_t* rayCount, uint32_t slot, bool c, bool m, bool sh)
{
    const uint32_t s0 = AllocSlotWave(rayCount, c), s1 = AllocSlotWave(rayCount + kCounterStride, m), s2 = AllocSlotWave(rayCount + 2 * kCounterStride, sh);
    if (c) rayList[s0] = slot;
    if (m) rayList[cap + s1] = slot;
    if (sh) rayList[2 * cap + s2] = slot;
}
// The same allocations for a whole 256-thread block: ONE returning atomic per counter and block instead of one per wave.  A word sustains ~88
// returning atomics per microsecond (MI355X_MICROARCH.md); a 1080p stage of K9 is 32 640 waves, i.e. 0.37 ms of atomics per counter when every
// wave allocates -- k_pt_init took 0.25 ms for work that needs a third of that.  want[k]: this lane wants a slot of counter k (k = 0: the path
// queue, 1 - 3: the C / M / S ray lists; counters[k] may be null = unused).  All threads of the block must call this (two barriers inside).
__device__ __forceinline__ void AllocSlotsBlock(uint32_t* const counters[4], const bool want[4], uint32_t slot[4])
{
    __shared__ uint32_t sCnt[4][kBlock / 64], sBase[4];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    uint64_t m[4];
    for (int k = 0; k < 4; k++) { m[k] = __ballot(want[k]); if (lane == 0) sCnt[k][wave] = (uint32_t)__popcll(m[k]); }
    __syncthreads();
    if (threadIdx.x < 4)
    {
        uint32_t total = 0;
        for (int w = 0; w < kBlock / 64; w++) total += sCnt[threadIdx.x][w];
        sBase[threadIdx.x] = (total && counters[threadIdx.x]) ? atomicAdd(counters[threadIdx.x], total) : 0u;
    }
    __syncthreads();
    const uint64_t below = (1ull << lane) - 1ull;
    for (int k = 0; k < 4; k++)
    {
        uint32_t before = 0;
        for (uint32_t w = 0; w < wave; w++) before += sCnt[k][w];
        slot[k] = sBase[k] + before + (uint32_t)__popcll(m[k] & below);
    }
    __syncthreads();      // (the next call of a grid-stride loop rewrites sCnt)
}
// path slot + ray-list entries of one wavefront stage through AllocSlotsBlock; returns the path's slot in the queue
__device__ __forceinline__ uint32_t AllocPathAndRaysBlock(uint32_t* outCount, uint32_t* rayList, uint32_t cap, uint32_t* rayCount, bool alive, bool c, bool m, bool sh,
    bool haveSlot = false, uint32_t givenSlot = 0)
{
    uint32_t* const ctr[4] = {haveSlot ? nullptr : outCount, rayCount, rayCount + kCounterStride, rayCount + 2 * kCounterStride};
    const bool want[4] = {!haveSlot && alive, c, m, sh};
    uint32_t s[4];
    AllocSlotsBlock(ctr, want, s);
    const uint32_t slot = haveSlot ? givenSlot : s[0];
    if (c) rayList[s[1]] = slot;
    if (m) rayList[cap + s[2]] = slot;
    if (sh) rayList[2 * cap + s[3]] = slot;
    return slot;
}
// entry j of the concatenation (C rays, M rays, S rays) of a queue's ray lists -> type, slot
__device__ __forceinline__ void RayOfIndex(const uint32_t* rayList, uint32_t cap, uint32_t nC, uint32_t nM, uint32_t j, uint32_t& type, uint32_t& slot)
{
    type = j < nC ? 0u : (j < nC + nM ? 1u : 2u);
    slot = rayList[type * cap + (j - (type == 0 ? 0u : (type == 1 ? nC : nC + nM)))];
}

__device__ __forceinline__ void CountWave(unsigned long long* counter, bool pred)
{
    const uint64_t m = __ballot(pred);
    if (m && __lane_id() == (uint32_t)(__ffsll((long long)m) - 1)) atomicAdd(counter, (unsigned long long)__popcll(m));
}

// pixel mapping: a 256-thread block covers a 16x16 tile; each wave64 covers one 8x8 quadrant in row-major order, so
// a wave is exactly one 8x8 thread group of the reference (GBufferRT_Common.h:6-7, IndirectLighting_Common.h:6-7).
__device__ __forceinline__ void PixelOfThread(uint32_t tilesX, uint32_t x0, uint32_t y0, uint32_t* x, uint32_t* y)
{
    const uint32_t tile = blockIdx.x;
    const uint32_t tx = tile % tilesX, ty = tile / tilesX;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    *x = x0 + tx * 16u + (wave & 1u) * 8u + (lane & 7u);
    *y = y0 + ty * 16u + (wave >> 1) * 8u + (lane >> 3);
}

// K11 / K10: (tile, wave of the tile, lane) of this thread for either block size (256: block = tile, 64: block = one wave of the tile)
template<int B>
__device__ __forceinline__ void TileWaveLaneB(uint32_t* tile, uint32_t* wave, uint32_t* lane)
{
    if (B == 256) { *tile = blockIdx.x; *wave = threadIdx.x >> 6; *lane = threadIdx.x & 63u; }
    else if (B == 128) { *tile = blockIdx.x >> 1; *wave = ((blockIdx.x & 1u) << 1) | (threadIdx.x >> 6); *lane = threadIdx.x & 63u; }      // two waves: half a tile
    else { *tile = blockIdx.x >> 2; *wave = blockIdx.x & 3u; *lane = threadIdx.x; }
}
__device__ __forceinline__ void RptTileWaveLane(uint32_t* tile, uint32_t* wave, uint32_t* lane) { TileWaveLaneB<kRptBlock>(tile, wave, lane); }
// PixelOfThread for either block size
template<int B>
__device__ __forceinline__ void PixelOfThreadB(uint32_t tilesX, uint32_t x0, uint32_t y0, uint32_t* x, uint32_t* y)
{
    uint32_t tile, wave, lane; TileWaveLaneB<B>(&tile, &wave, &lane);
    const uint32_t tx = tile % tilesX, ty = tile / tilesX;
    *x = x0 + tx * 16u + (wave & 1u) * 8u + (lane & 7u);
    *y = y0 + ty * 16u + (wave >> 1) * 8u + (lane >> 3);
}
#ifndef ZR_RGI_BLOCK
#define ZR_RGI_BLOCK 64
#endif
static constexpr int kRgiBlock = ZR_RGI_BLOCK;      // K10 (k_rgi), same trade as kRptBlock: 1.298 -> 1.262 ms Cornell, 10.01 -> 9.57 ms atrium

// ------------------------------------------------------------------------------------------------ ReSTIR PT kernels
// per-lane ray counters -> one atomic pair per wave (all 64 lanes must call this)
__device__ __forceinline__ void FlushRayCounters(unsigned long long* counters, const uint32_t* cnt)
{
#ifdef ZR_NO_RAY_COUNTERS      // (measurement build: what the per-wave counter atomics cost)
    return;
#endif
    uint32_t a = cnt[0], b = cnt[1];
    for (int s = 1; s < 64; s <<= 1) { a += __shfl_xor(a, s); b += __shfl_xor(b, s); }
    if (__lane_id() == 0)
    {
        if (a) atomicAdd(counters + 0, (unsigned long long)a);
        if (b) atomicAdd(counters + 1, (unsigned long long)b);
    }
}

// the wave's LIFETIME (shader cycles / 16, from t0 = the cycle counter at kernel entry) also goes to the cost-map cell of its pixels: ray counts
// turned out to be a poor load signal (rays of a dense region cost several times those of an open one: the ray-balanced 8-way split of the
// atrium had a 4.1 ms tile against 3.6 ms for the equal-area grid), time is the thing to balance.  All 64 lanes call this; (x, y): any pixel
// of the wave -- a wave's pixels share a 32 x 32 cell (16 x 4 / 8 x 8 groups of 32-aligned tiles, or one K12 sort tile).
__device__ __forceinline__ void FlushRayCountersCost(const rpt::RptFrame& F, unsigned long long* counters, const uint32_t* cnt, uint32_t x, uint32_t y, bool inFrame,
    unsigned long long t0)
{
    if (F.costMap != nullptr)
    {
        const uint64_t m = __ballot(inFrame);
        if (m != 0)
        {
            const int leader = __ffsll((long long)m) - 1;
            unsigned long long dt = (__builtin_readcyclecounter() - t0) >> 4;
            if (F.costMode)
            {   // ray mode: the wave's queries (lanes outside the frame issued none)
                uint32_t r = cnt[0] + cnt[1];
                for (int s = 1; s < 64; s <<= 1) r += __shfl_xor(r, s);
                dt = r;
            }
            if ((int)__lane_id() == leader && dt) atomicAdd(&F.costMap[((y - F.gb.y0) >> 5) * F.costW + ((x - F.gb.x0) >> 5)], (uint32_t)(dt > 0xffffffull ? 0xffffffull : dt));
        }
    }
    FlushRayCounters(counters, cnt);
}

// K11: block = 16x16 pixels, wave w = rows 4w..4w+3 (a 16x4 block: the RR "wave" of the ABI, zr_rpt.h header)
// EMISSIVE: the NEE_EMISSIVE shader permutation (emissive triangles vs sun + sky); a template constant so the other variant folds away
// TEX: the scene has a texture heap (ray differentials carried, material maps sampled); likewise a template constant
// PARK: the reservoir's selected reconnection in LDS instead of registers / scratch (zr_rpt.h RcPark: 17 words x 64 lanes = 4.25 KB per one-wave block)
// PLAIN: the scene's material class (SceneView::plain: opaque uncoated non-metallic dielectrics only); likewise a template constant -- the kernel then has
// no code for the other lobes (24 283 -> 9 529 VALU instructions; Cornell 1080p 0.853 -> 0.77 ms).  The host launches it only for such scenes.
template<bool EMISSIVE, bool TEX, bool NODE_CACHE = false, bool PARK = false, bool PLAIN = false>
__device__ __forceinline__ void RptPathtraceBody(rpt::RptFrame& F, const zr_frame_constants& g, uint32_t tilesX, unsigned long long* counters)
{
    const unsigned long long t0 = __builtin_readcyclecounter();
    F.prm.emissive = EMISSIVE ? 1u : 0u; F.prm.textured = TEX ? 1u : 0u;
    rpt::SetMaterialClass(F, PLAIN);
    uint32_t tile, wave, lane; RptTileWaveLane(&tile, &wave, &lane);
    const uint32_t tx = tile % tilesX, ty = tile / tilesX;
    const uint32_t x = F.ox0 + tx * 16u + (lane & 15u), y = F.oy0 + ty * 16u + wave * 4u + (lane >> 4);
    ZR_TRAV_STACK_B(stack, kRptBlock);
    if (NODE_CACHE) { ZR_NODE_CACHE_FILL(stack, F.sc, kRptBlock); }      // the tree's top ZR_NODE_CACHE nodes in LDS (large scenes: 8.37 -> 7.63 ms on the atrium)
#if ZR_SCENE_LDS
    else { ZR_SCENE_CACHE_FILL(stack, F.sc, kRptBlock); }                 // small scenes: all of it
#endif
    ZR_PROF_KERNEL(F.sc, 1);
    uint32_t cnt[2] = {0u, 0u};
    rpt::PTLane P;
    { ZR_PROF_SCOPE(ZRP_MISC0); rpt::PtInitLane_Fused(F.sc, g, F.gb, F.prm, F.Owns(x, y), x, y, F.finalRGBA, stack, cnt, P); }
    __shared__ uint32_t rcParkLds[PARK ? rpt::kRcParkWords * kRptBlock : 1];
    if (PARK) { P.r.park.p = (ZR_LDS_AS uint32_t*)rcParkLds + threadIdx.x; P.r.park.stride = kRptBlock; P.r.parked = false; }
    for (;;)
    {
        const bool any = __ballot(P.active) != 0;
        rpt::PtPhaseA_Fused(F.sc, g, F.prm, stack, cnt, P);
        if (!any) break;
        uint32_t key = rpt::PtRRKey(P);
        if (__ballot(key != 0) != 0)
        {
            for (int s = 1; s < 64; s <<= 1) { uint32_t o = __shfl_xor(key, s); key = o > key ? o : key; }
        }
        { ZR_PROF_SCOPE(ZRP_MISC1); rpt::PtPhaseB(F.sc, F.prm, P, key); }
    }
    { ZR_PROF_SCOPE(ZRP_MISC2); rpt::PtFinishLane(F.gb, F.prm, F.cur, F.tex, F.finalRGBA, P); }
    FlushRayCountersCost(F, counters, cnt, x, y, F.Owns(x, y), t0);
}
template<bool EMISSIVE, bool PLAIN = false>
__global__ void __launch_bounds__(kRptBlock) ZR_WAVES_PATHTRACE k_rpt_pathtrace(rpt::RptFrame F, zr_frame_constants g, uint32_t tilesX, unsigned long long* counters)
{ RptPathtraceBody<EMISSIVE, false, false, false, PLAIN>(F, g, tilesX, counters); }
// The same kernel at 4 waves per SIMD (128 VGPRs, more spills): used for scenes whose BVH does not fit the caches, where the inline
// traversal is latency-bound and the extra wave hides more than the spills cost (380 k-triangle atrium: 11.7 -> 10.6 ms; on the
// 58-triangle Cornell box both took 1.16 ms in round 3; round 6: 0.817 -> 0.803 general, 0.699 -> 0.679 PLAIN, whose tree is then all in LDS -- every scene takes it).
template<bool EMISSIVE, bool PLAIN = false>
__global__ void __launch_bounds__(kRptBlock) ZR_WAVES_PATHTRACE_LARGE k_rpt_pathtrace_w4(rpt::RptFrame F, zr_frame_constants g, uint32_t tilesX, unsigned long long* counters)
{ RptPathtraceBody<EMISSIVE, false, true, false, PLAIN>(F, g, tilesX, counters); }
// The TEXTURED permutation at 6 waves per SIMD: its dependent texel gathers are latency that more waves hide -- textured atrium 11.99 ms at the
// compiler's 2 waves (255 VGPRs), 10.74 at >= 3, 10.28 at 4, 10.11 at 5, **9.36 at 6**, 9.69 at 7, 9.96 at 8 (scripts/gpu_tex.sh [rounds 1-4: git history up to 31e92fa; today scripts/gpu.sh ab], gpu_waves.sh); the
// untextured large-scene build stays at 4 (5: 8.49, 6: 8.35 against 8.02 ms).  Round 1 found that forcing it to exactly 3 waves
// (amdgpu_waves_per_eu(3, 3): ~150 spilled VGPRs) makes ROCm 7.2's clang miscompile the <sun + sky, textured> instance -- the y / z components
// of the reconnection radiance rc.L of case-1 samples were written as 0 in ~70 % of the pixels; the builds above all pass the 15 textured parity
// tests that caught it (tests/test_gpu_parity.py::test_textured_integrators_on_gpu, the *_textured reference-pass cases).
#ifndef ZR_WAVES_PATHTRACE_TEX
#define ZR_WAVES_PATHTRACE_TEX ZR_WAVES(6)
#endif
template<bool EMISSIVE>
__global__ void __launch_bounds__(kRptBlock) ZR_WAVES_PATHTRACE_TEX k_rpt_pathtrace_tex(rpt::RptFrame F, zr_frame_constants g, uint32_t tilesX, unsigned long long* counters)
{ RptPathtraceBody<EMISSIVE, true>(F, g, tilesX, counters); }

#ifdef ZR_EXPERIMENTS
#include "zr_kernels_exp.h"      // K11's park / compact / trip / pool forms (negative results; experiments build only)
#endif

enum RptPixelPass { RPT_REPLAY_CTT = 0, RPT_REPLAY_TTC, RPT_RECONNECT_TEMPORAL, RPT_SPATIAL_SEARCH, RPT_REPLAY_CTS, RPT_REPLAY_STC };

// light per-pixel kernels (no traversal, no scratch)
// tiles per block of the light per-pixel kernels.  They end in ONE returning atomic per block and work list, and both list counters live in one cache
// line: with a tile per block the 2 x 8160 atomics of a 1080p frame were the whole kernel (0.093 ms on the atrium for 29 MB of plane reads -- the
// ~88 dequeues per microsecond one word sustains, MI355X_MICROARCH.md).  Four tiles per block: a quarter of the atomics.
// (the search kernel does real work per pixel -- three dependent candidate gathers -- and is latency-bound rather than atomic-bound: two tiles)
static constexpr uint32_t kLightTilesPerBlock = 4, kSearchTilesPerBlock = 2;
template<int PASS>
__global__ void __launch_bounds__(kBlock) k_rpt_light(rpt::RptFrame F, zr_frame_constants g, uint32_t tilesX, uint32_t* listA, uint32_t* listB, uint32_t* counts)
{
    constexpr uint32_t TPB = PASS == 0 ? kLightTilesPerBlock : (PASS == 1 ? kSearchTilesPerBlock : 1u);      // (the LDS-tile experiment stages one tile's neighbourhood)
    constexpr uint32_t VW = TPB * (uint32_t)(kBlock / 64);             // "virtual waves" of the block: (tile, wave) pairs, in that order
    const uint32_t numTiles = tilesX * ((F.oh + 15u) / 16u);
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    uint32_t a[TPB], b[TPB], pid[TPB];      // replay class of this thread's pixel of tile t for list A / B (0 = not on the list; zr_rpt.h ReplayClass)
    __shared__ uint32_t sCnt[2][3][VW], sBase[2];
    for (uint32_t t = 0; t < TPB; t++)
    {
        const uint32_t tile = blockIdx.x * TPB + t;
        const uint32_t tx = tile % tilesX, ty = tile / tilesX;
        const uint32_t x = F.ox0 + tx * 16u + (wave & 1u) * 8u + (lane & 7u), y = F.oy0 + ty * 16u + (wave >> 1) * 8u + (lane >> 3);
        const bool in = tile < numTiles && F.Owns(x, y);
        a[t] = 0; b[t] = 0;
        if (PASS == 0)          // temporal work lists: pixels whose current / temporal reservoir needs a replay (k > 2)
        {
            if (in) { a[t] = rpt::NeedsReplayCtT(F, x, y); b[t] = rpt::NeedsReplayTtC(F, g, x, y); }
        }
        else                    // K15 spatial search, then the spatial work lists
        {
            if (PASS == 2)
            {
                // N3 experiment (north_star: "LDS-staged reservoir tiles for the spatial-reuse stencil"): the 46 x 46 texels (mr, depth, normal: 10 B)
                // the 16 x 16 block's searches can touch (radius 15) staged in LDS first; same arithmetic, candidates read from the tile
                constexpr int R = rpt::kSearchRadius, T = 16 + 2 * R;
                __shared__ uint16_t tMr[T * T]; __shared__ float tDepth[T * T]; __shared__ uint32_t tNormal[T * T];
                const uint32_t tx0 = F.ox0 + tx * 16u, ty0 = F.oy0 + ty * 16u;
                for (uint32_t i = threadIdx.x; i < (uint32_t)(T * T); i += kBlock)
                {
                    const int gx = (int)tx0 - R + (int)(i % T), gy = (int)ty0 - R + (int)(i / T);
                    const bool ok = gx >= 0 && gy >= 0 && gx < (int)g.render_width && gy < (int)g.render_height && rpt::InPlanes(F.gb, gx, gy);
                    const size_t sp = ok ? rpt::Pix(F.gb, (uint32_t)gx, (uint32_t)gy) : 0;
                    tMr[i] = ok ? F.gb.mr[sp] : (uint16_t)0; tDepth[i] = ok ? F.gb.depth[sp] : 0.0f; tNormal[i] = ok ? F.gb.normal[sp] : 0u;
                }
                __syncthreads();
                struct TileFetch
                {
                    const uint16_t* mr; const float* depth; const uint32_t* normal; int x0, y0;
                    __device__ void operator()(int sx, int sy, uint16_t& m, float& d, uint32_t& n) const
                    { const int i = (sy - y0) * T + (sx - x0); m = mr[i]; d = depth[i]; n = normal[i]; }
                } tf{tMr, tDepth, tNormal, (int)tx0 - R, (int)ty0 - R};
                if (in) { rpt::SpatialSearchPixelT(F, g, x, y, tf); a[t] = rpt::NeedsReplayCtS(F, x, y); b[t] = rpt::NeedsReplayStC(F, x, y); }
            }
            else if (in) { rpt::SpatialSearchPixel(F, g, x, y); a[t] = rpt::NeedsReplayCtS(F, x, y); b[t] = rpt::NeedsReplayStC(F, x, y); }
        }
        pid[t] = in ? (uint32_t)rpt::Pix(F.gb, x, y) : 0u;
        for (uint32_t c = 0; c < 3u; c++)
        {
            const uint64_t ma = __ballot(a[t] == c + 1u), mb = __ballot(b[t] == c + 1u);
            if (lane == 0) { sCnt[0][c][t * (kBlock / 64) + wave] = (uint32_t)__popcll(ma); sCnt[1][c][t * (kBlock / 64) + wave] = (uint32_t)__popcll(mb); }
        }
    }
    // one atomic per block and list (see kLightTilesPerBlock).  Inside the block's chunk of a list the entries are ordered by replay class
    // (k = 3, 4, >= 5 -- K12's buckets), so that the 64 lanes of a replay wave mostly walk paths of the same length; which thread replays which
    // pixel has no effect on the result.
    __syncthreads();
    if (threadIdx.x < 2)
    {
        uint32_t total = 0;
        for (uint32_t c = 0; c < 3u; c++) for (uint32_t w = 0; w < VW; w++) total += sCnt[threadIdx.x][c][w];
        sBase[threadIdx.x] = total ? atomicAdd(counts + threadIdx.x, total) : 0u;
    }
    __syncthreads();
    const uint64_t below = (1ull << lane) - 1ull;
    uint32_t oa = sBase[0], ob = sBase[1];
    for (uint32_t c = 0; c < 3u; c++)
        for (uint32_t t = 0; t < TPB; t++)
        {
            const uint64_t ma = __ballot(a[t] == c + 1u), mb = __ballot(b[t] == c + 1u);
            uint32_t wa = 0, wb = 0, ta = 0, tb = 0;      // entries of this (class, tile) before this wave, and in all of its waves
            for (uint32_t w = 0; w < (uint32_t)(kBlock / 64); w++) { const uint32_t na = sCnt[0][c][t * (kBlock / 64) + w], nb = sCnt[1][c][t * (kBlock / 64) + w]; ta += na; tb += nb; if (w < wave) { wa += na; wb += nb; } }
            if (a[t] == c + 1u) listA[oa + wa + (uint32_t)__popcll(ma & below)] = pid[t];
            if (b[t] == c + 1u) listB[ob + wb + (uint32_t)__popcll(mb & below)] = pid[t];
            oa += ta; ob += tb;
        }
}

// K13 replays over work lists (device-side counts, fixed grid, grid-stride): only pixels with k > 2 pay for the heavy kernel.  One launch runs the
// two replays of a stage (CtT + TtC, CtS + StC: independent -- each writes its own r-buffer): the first half of the grid walks list A with
// PASS_A, the second half list B with PASS_A + 1, each specialisation compiled for its pass.
// K13: the replay passes run over work lists (pixels whose reservoir holds a path with k > 2).  A path's replay costs anything between a few
// hundred instructions and two full bounces, so a static split of a list over the blocks leaves the kernel waiting for its unluckiest block
// (atrium: 1.99 of 3 resident waves per SIMD on average, PMC).  DYNAMIC: a persistent grid (as many waves as are resident at once) in which
// every wave pulls the next 128 entries from a cursor next to the list's count -- one returning atomic per wave and chunk -- first from list A
// until it is empty, then from list B.  Which wave replays a pixel has no influence on the result.  Measured (scripts/gpu_r03_replay.sh [rounds 1-4: git history up to 31e92fa; today scripts/gpu.sh ab],
// atrium): the temporal replays 1.27 -> 1.19 ms at 1080p, 4.14 -> 3.46 ms at 3840 x 2160; the spatial replays get SLOWER that way (0.88 -> 1.03 ms at
// 1080p, unchanged at 4K) and keep the static split; with 64-entry chunks and a 2048-block grid the cursor itself was the bottleneck (16 k
// returning atomics on one word: + 0.5 ms on the Cornell frame, whose lists are nearly empty).
template<int PASS, bool EMISSIVE, bool TEX>
__device__ __forceinline__ void RptReplayPixel(rpt::RptFrame& F, const zr_frame_constants& g, uint32_t pid, const TravStack& stack, uint32_t* cnt)
{
    const uint32_t x = F.gb.x0 + pid % F.gb.w, y = F.gb.y0 + pid / F.gb.w;
    const uint32_t before = cnt[0] + cnt[1];
    if (PASS == RPT_REPLAY_CTT) rpt::ReplayTemporalPixel(F, g, 0, x, y, stack, cnt);
    else if (PASS == RPT_REPLAY_TTC) rpt::ReplayTemporalPixel(F, g, 1, x, y, stack, cnt);
    else if (PASS == RPT_REPLAY_CTS) rpt::ReplaySpatialPixel(F, g, 0, x, y, stack, cnt);
    else rpt::ReplaySpatialPixel(F, g, 1, x, y, stack, cnt);
    // ray-mode cost map (diagnostic): the lanes of a replay wave come from anywhere in the frame, so every lane adds to its own pixel's cell
    if (F.costMap != nullptr && F.costMode && cnt[0] + cnt[1] != before) atomicAdd(&F.costMap[((y - F.gb.y0) >> 5) * F.costW + ((x - F.gb.x0) >> 5)], cnt[0] + cnt[1] - before);
}
template<int PASS, bool EMISSIVE, bool TEX, bool DYNAMIC>
__device__ __forceinline__ void RptReplayList(rpt::RptFrame& F, const zr_frame_constants& g, const uint32_t* list, uint32_t n, uint32_t* cursor,
    unsigned long long* counters, uint32_t block, uint32_t numBlocks, const TravStack& stack)
{
    uint32_t cnt[2] = {0u, 0u};
    if (DYNAMIC)
    {
        const uint32_t lane = threadIdx.x & 63u;
        // 128 entries per pull (two replays per lane) keep the cursor's atomics rare on a full frame's list; a list too short to give every resident
        // wave such a chunk -- a device's tile of the screen split: 75 k entries against 3072 waves on the 8-way 1080p atrium -- is pulled 64 at a
        // time instead, so that it spreads over twice the waves and nobody replays two paths back to back while most of the chip idles (r04: the
        // tile's temporal replays took 0.63 ms for an eighth of the frame's 1.10 ms list)
        const uint32_t chunk = n < 128u * gridDim.x * (blockDim.x / 64u) ? 64u : 128u;
        for (; n != 0;)
        {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(cursor, chunk);
            base = __shfl(base, 0);
            if (base >= n) break;
            if (base + lane < n) RptReplayPixel<PASS, EMISSIVE, TEX>(F, g, list[base + lane], stack, cnt);
            if (chunk == 128u && base + 64u + lane < n) RptReplayPixel<PASS, EMISSIVE, TEX>(F, g, list[base + 64u + lane], stack, cnt);
        }
    }
    else
        for (uint32_t i = block * blockDim.x + threadIdx.x; i < n; i += numBlocks * blockDim.x) RptReplayPixel<PASS, EMISSIVE, TEX>(F, g, list[i], stack, cnt);
    if (n) FlushRayCounters(counters, cnt);
}
// the replay kernels wait for memory with their VALUs idle 30 - 40 % of the time (profiles/r06_pmc_rpt_atrium.json), so they take the spills of 128 VGPRs for a fourth
// wave per SIMD: atrium temporal replays 1.155 -> 1.02 ms (3 waves, 144 VGPRs, 768 blocks before), frame 14.7 -> 14.35 ms; 5 waves: 1.15 (profiles/r06v_ab_replay_waves.txt)
#ifndef ZR_WAVES_REPLAY_N
#define ZR_WAVES_REPLAY_N 4
#endif
#define ZR_WAVES_REPLAY __attribute__((amdgpu_waves_per_eu(ZR_WAVES_REPLAY_N, ZR_WAVES_REPLAY_N)))
static constexpr uint32_t kReplayPersistentBlocks = 256u * ZR_WAVES_REPLAY_N;      // what is resident at once: N waves per SIMD x 1024 SIMDs / 4 waves per block
// counts: {entries of list A, entries of list B}; counts[4], counts[5]: the two cursors (zeroed with the counts)
template<int PASS_A, bool EMISSIVE, bool TEX>
__global__ void __launch_bounds__(kBlock) ZR_WAVES_REPLAY k_rpt_replay(rpt::RptFrame F, zr_frame_constants g, const uint32_t* listA, const uint32_t* listB, uint32_t* counts,
    unsigned long long* counters)
{
    F.prm.emissive = EMISSIVE ? 1u : 0u; F.prm.textured = TEX ? 1u : 0u;
    rpt::SetMaterialClass(F, false);      // (no PLAIN permutation of the replays: the class must still be a constant, or both forms of everything it selects are compiled in)
    ZR_TRAV_STACK(stack);
#ifdef ZR_NODE_CACHE_MORE
    ZR_NODE_CACHE_FILL(stack, F.sc, kBlock);
#endif
    // (counters: the per-kernel slots of the two passes are neighbours, zr_api.hip kCounterNames)
    if (PASS_A == RPT_REPLAY_CTT)
    {
        RptReplayList<PASS_A, EMISSIVE, TEX, true>(F, g, listA, counts[0], counts + 4, counters, 0, 0, stack);
        RptReplayList<PASS_A + 1, EMISSIVE, TEX, true>(F, g, listB, counts[1], counts + 5, counters + 2, 0, 0, stack);
    }
    else
    {
        const uint32_t half = gridDim.x / 2u;
        if (blockIdx.x < half) RptReplayList<PASS_A, EMISSIVE, TEX, false>(F, g, listA, counts[0], nullptr, counters, blockIdx.x, half, stack);
        else RptReplayList<PASS_A + 1, EMISSIVE, TEX, false>(F, g, listB, counts[1], nullptr, counters + 2, blockIdx.x - half, half, stack);
    }
}

// K12 (ReSTIR_PT_Sort.hlsl:99-368): one 256-thread block per 32 x 32 pixel tile, thread = 2 x 2 quad, wave w = threads 64 w .. 64 w + 63 of the
// group (SV_GroupIndex order).  Buckets by reconnection depth; inside a bucket: wave, then lane, then quad slot.  The shader takes the wave
// order from the arrival of an LDS InterlockedAdd (unspecified); the ABI fixes it to the wave index, which is what a prefix over per-wave
// counts gives without atomics.  tile0 / tilesX: the 32 x 32 tiles of the owned rect, in render-target tile coordinates.
template<int VARIANT>
__device__ __forceinline__ void RptSortTile(const rpt::RptFrame& F, const zr_frame_constants& g, uint32_t tile, uint32_t tilesX, uint32_t tile0x, uint32_t tile0y, uint16_t* map)
{
    const uint32_t W = g.render_width, H = g.render_height, dimX = (W + 31u) / 32u, dimY = (H + 31u) / 32u;
    const uint32_t gx = tile0x + tile % tilesX, gy = tile0y + tile / tilesX;
    const uint32_t gidx = threadIdx.x, wave = gidx >> 6, lane = gidx & 63u;
    const bool againstEdge = gx == dimX - 1u || gy == dimY - 1u, lastGroup = gx == dimX - 1u && gy == dimY - 1u;
    uint32_t cls[4], res[4], px[4], py[4], gtx[4], gty[4];
    for (int i = 0; i < 4; i++)
    {
        gtx[i] = (gidx & 15u) * 2u + (uint32_t)(i & 1); gty[i] = (gidx >> 4) * 2u + (uint32_t)(i >> 1);
        px[i] = gx * 32u + gtx[i]; py[i] = gy * 32u + gty[i];
        cls[i] = rpt::SortClassify(F, g, VARIANT, px[i], py[i], againstEdge, res[i]);
    }
    // per bucket: this wave's count and this lane's exclusive prefix (WaveActiveSum / WavePrefixSum of dot(1, flags))
    __shared__ uint32_t sCnt[4][5];
    const uint64_t lt = (1ull << lane) - 1ull;
    uint32_t lanePrefix[5];
    for (uint32_t c = 0; c < 5u; c++)
    {
        uint32_t n = 0, pre = 0;
        for (int i = 0; i < 4; i++) { const uint64_t m = __ballot(cls[i] == c); n += (uint32_t)__popcll(m); pre += (uint32_t)__popcll(m & lt); }
        lanePrefix[c] = pre;
        if (lane == 0) sCnt[wave][c] = n;
    }
    __syncthreads();
    uint32_t base[5], run = 0;
    for (uint32_t c = 0; c < 5u; c++)
    {
        uint32_t before = 0, total = 0;
        for (uint32_t w = 0; w < 4u; w++) { const uint32_t n = sCnt[w][c]; total += n; if (w < wave) before += n; }
        base[c] = run + before; run += total;
    }
    const bool spatialResample = F.prm.doSpatial != 0;
    for (int i = 0; i < 4; i++)
    {
        uint32_t mgx = gtx[i], mgy = gty[i];
        if (!lastGroup)      // (the very last group maps one to one)
        {
            uint32_t in2x2 = 0;
            for (int j = 0; j < i; j++) in2x2 += cls[j] == cls[i] ? 1u : 0u;
            uint32_t b = base[4], lp = lanePrefix[4];      // (selects, not indexed register arrays)
            for (uint32_t c = 0; c < 4u; c++) if (cls[i] == c) { b = base[c]; lp = lanePrefix[c]; }
            const uint32_t idx = b + lp + in2x2;
            mgx = idx & 31u; mgy = idx >> 5;
        }
        if (gx == dimX - 1u && gy != dimY - 1u) { const uint32_t t = mgx; mgx = mgy; mgy = t; }      // transposed at the right image boundary
        const uint32_t mx = gx * 32u + mgx, my = gy * 32u + mgy;
        if (mx < W && my < H && rpt::InPlanes(F.gb, (int)mx, (int)my))
            map[rpt::Pix(F.gb, mx, my)] = rpt::EncodeSorted(px[i], py[i], mx, my, rpt::SortErrorBits(VARIANT, res[i], spatialResample));
    }
}

// one launch sorts a stage's two maps (Sort_TtC + Sort_CtT, Sort_CtS + Sort_StC): first half of the grid = variant VA into mapA, second half = VB into mapB
template<int VA, int VB>
__global__ void __launch_bounds__(256) k_rpt_sort(rpt::RptFrame F, zr_frame_constants g, uint32_t tilesX, uint32_t tile0x, uint32_t tile0y, uint16_t* mapA, uint16_t* mapB)
{
    const uint32_t half = gridDim.x / 2u;
    if (blockIdx.x < half) RptSortTile<VA>(F, g, blockIdx.x, tilesX, tile0x, tile0y, mapA);
    else RptSortTile<VB>(F, g, blockIdx.x - half, tilesX, tile0x, tile0y, mapB);
}

// Block size of the two reconnect kernels (K14, K16): nothing in them is shared between the waves of a block, so a 16 x 16 tile can be one
// 256-thread block or four one-wave blocks (the trade of kRptBlock: a block's registers and LDS are only released when its slowest wave ends).
#ifndef ZR_RECON_BLOCK
#define ZR_RECON_BLOCK 256
#endif
static constexpr int kReconBlock = ZR_RECON_BLOCK;
// K16's own block size (ZR_STC_BLOCK, default = kReconBlock) and where its lane record lives.  rpt::StcLane -- two reservoirs with their
// reconnections, ~110 words -- is indexed through merged stores (Makefile note), so it cannot be promoted to registers: as a local it is
// scratch, i.e. it streams through L1 / L2 and every dirty line is written to HBM when the wave's scratch is recycled (k_rpt_stc moved
// 2.1 GB per 1080p launch against 0.87 GB of planes, profiles/r05_pmc_rpt_cornell.json).  -DZR_STC_LDS=1 -DZR_STC_BLOCK=64 keeps it in LDS
// instead (one-wave blocks: 64 x 440 B = 28 KB + the 4 KB stack, five blocks per CU).
#ifndef ZR_STC_BLOCK
#define ZR_STC_BLOCK ZR_RECON_BLOCK
#endif
#ifndef ZR_STC_LDS
#define ZR_STC_LDS 0
#endif
static constexpr int kStcBlock = ZR_STC_BLOCK;
#ifndef ZR_VOTE_WT_STC
#define ZR_VOTE_WT_STC 2
#endif

// K14: CtT + TtC fused per pixel (zr_rpt.h ReconnectTemporalPixel)
template<bool EMISSIVE, bool TEX, bool PLAIN = false>
__global__ void __launch_bounds__(kReconBlock) ZR_WAVES_TEMPORAL k_rpt_temporal(rpt::RptFrame F, zr_frame_constants g, uint32_t tilesX, unsigned long long* counters)
{
    const unsigned long long t0 = __builtin_readcyclecounter();
    F.prm.emissive = EMISSIVE ? 1u : 0u; F.prm.textured = TEX ? 1u : 0u;
    rpt::SetMaterialClass(F, PLAIN);
    uint32_t x, y; PixelOfThreadB<kReconBlock>(tilesX, F.ox0, F.oy0, &x, &y);
    // SORT_TEMPORAL: threads take their pixel from a K12 map, which puts reservoirs of equal reconnection depth into the same wave.  Scheduling
    // only (no wave operation in CtT / TtC); the error bit is ignored because the fused kernel runs both shifts of a pixel
    if (F.prm.temporalMap && F.Owns(x, y)) { const uint16_t e = (F.prm.temporalMap == 1u ? F.mapCtN : F.mapNtC)[rpt::Pix(F.gb, x, y)]; rpt::DecodeSorted(e & 0x7fffu, x, y); }
    ZR_TRAV_STACK_B(stack, kReconBlock);
#if ZR_SCENE_LDS
    ZR_SCENE_CACHE_FILL(stack, F.sc, kReconBlock);
#endif
    ZR_PROF_KERNEL(F.sc, 2);
    uint32_t cnt[2] = {0u, 0u};
    if (F.Owns(x, y)) rpt::ReconnectTemporalPixel(F, g, x, y, stack, cnt);
    FlushRayCountersCost(F, counters, cnt, x, y, F.Owns(x, y), t0);
}

// canonical wave sum of the ABI: xor butterfly, strides 1..32 (zr_rpt.h ButterflySum64 is the host statement of it)
__device__ __forceinline__ float WaveSumButterfly(float v)
{
    for (int s = 1; s < 64; s <<= 1) v = v + __shfl_xor(v, s);
    return v;
}

// K16 CtS + StC: wave = 8x8 pixel group; every lane of the wave walks all four phases (absent lanes contribute 0)
template<bool EMISSIVE, bool TEX, bool PLAIN = false>
__global__ void __launch_bounds__(kStcBlock) ZR_WAVES_STC k_rpt_stc(rpt::RptFrame F, zr_frame_constants g, uint32_t tilesX, unsigned long long* counters)
{
    const unsigned long long t0 = __builtin_readcyclecounter();
    F.prm.emissive = EMISSIVE ? 1u : 0u; F.prm.textured = TEX ? 1u : 0u;
    rpt::SetMaterialClass(F, PLAIN);
    uint32_t x, y; PixelOfThreadB<kStcBlock>(tilesX, F.ox0, F.oy0, &x, &y);
    // SORT_SPATIAL (ReSTIR_PT_Reconnect_StC.hlsl:133-140): the thread at (x, y) shifts the pixel the NtC map assigns to its position, so the
    // four wave sums below run over the 64 pixels K12 put together (error bit: nothing to do -- the lane stays in the wave, contributing 0)
    if (F.prm.sortSpatial && F.Owns(x, y) && !rpt::DecodeSorted(F.mapNtC[rpt::Pix(F.gb, x, y)], x, y)) x = 0xffffffffu;
    ZR_TRAV_STACK_B(stack, kStcBlock);
    stack.voteTri = ZR_VOTE_WT_STC;      // (zr_dev_scene.h TravStack::voteTri)
#if ZR_SCENE_LDS
    ZR_SCENE_CACHE_FILL(stack, F.sc, kStcBlock);
#endif
    uint32_t cnt[2] = {0u, 0u};
    ZR_PROF_KERNEL(F.sc, 3);
#if ZR_STC_LDS
    // (raw words: a __shared__ object may not have initialisers, and StcLane's reservoirs carry default member initialisers)
    __shared__ __attribute__((aligned(16))) uint32_t stcLaneLds[kStcBlock * ((sizeof(rpt::StcLane) + 15) / 16 * 4)];
    rpt::StcLane& a = *reinterpret_cast<rpt::StcLane*>(stcLaneLds + threadIdx.x * ((sizeof(rpt::StcLane) + 15) / 16 * 4));
    a.r_curr.park.p = nullptr; a.r_curr.park.stride = 0; a.r_curr.parked = false; a.r_spatial.park.p = nullptr; a.r_spatial.park.stride = 0; a.r_spatial.parked = false;
#else
    rpt::StcLane a;
    // The general permutation keeps the lane's two reservoirs (304 B) as an object in scratch memory: split into registers they cost it 160 - 220 spilled
    // VGPRs at its 128 (atrium: 2.72 -> 2.85 ms), whereas the PLAIN permutation, a third of the code, gains from the split (Cornell: 0.578 -> 0.539 ms).
    if (!PLAIN) ZR_KEEP_IN_MEMORY(a);
#endif
    float v1, v2, v3, v4;
    { ZR_PROF_SCOPE(ZRP_MISC0); rpt::StcPhase0(F, g, x, y, a, v1, v2); }
    { ZR_PROF_SCOPE(ZRP_MISC1);
    if (a.valid && a.hasN) rpt::ReconnectCtSPixel(F, g, x, y, stack, cnt); }      // K16 CtS of this pixel (see zr_rpt.h)
    const float sum1 = WaveSumButterfly(v1), sum2 = WaveSumButterfly(v2);
    rpt::StcPhase1(F, g, a, sum1, v3);
    const float sum3 = WaveSumButterfly(v3);
    ZR_PROF_SCOPE(ZRP_MISC2);
    rpt::StcPhase2(F, g, a, sum1, stack, cnt, v4);
    const float sum4 = WaveSumButterfly(v4);
    rpt::StcPhase3(F, g, a, sum2 + sum3 + sum4);
    FlushRayCountersCost(F, counters, cnt, x, y, x != 0xffffffffu && F.Owns(x, y), t0);
}


// ------------------------------------------------------------------------------------------------ translation-unit split
// ZR_RPT_GROUP_*(X): X = `template` in the TU that owns the group (zr_tu_rpt_<letter>.hip), `extern template` everywhere else.  The groups are cut for
// the build's wall clock (8 jobs): roughly equal compile times, no group above a minute.
#define ZR_RPT_ARGS_TILE (rpt::RptFrame, zr_frame_constants, uint32_t, unsigned long long*)
#define ZR_RPT_ARGS_LIST (rpt::RptFrame, zr_frame_constants, const uint32_t*, const uint32_t*, uint32_t*, unsigned long long*)
// K11, untextured
#define ZR_RPT_GROUP_A(X) \
    X __global__ void k_rpt_pathtrace<true, false> ZR_RPT_ARGS_TILE; X __global__ void k_rpt_pathtrace<false, false> ZR_RPT_ARGS_TILE; \
    X __global__ void k_rpt_pathtrace_w4<true, false> ZR_RPT_ARGS_TILE; X __global__ void k_rpt_pathtrace_w4<false, false> ZR_RPT_ARGS_TILE;
// K11 + K14, textured
#define ZR_RPT_GROUP_G(X) \
    X __global__ void k_rpt_pathtrace_tex<true> ZR_RPT_ARGS_TILE; X __global__ void k_rpt_pathtrace_tex<false> ZR_RPT_ARGS_TILE; \
    X __global__ void k_rpt_temporal<true, true, false> ZR_RPT_ARGS_TILE; X __global__ void k_rpt_temporal<false, true, false> ZR_RPT_ARGS_TILE;
// K14 + K16, untextured
#define ZR_RPT_GROUP_H(X) \
    X __global__ void k_rpt_temporal<true, false, false> ZR_RPT_ARGS_TILE; X __global__ void k_rpt_temporal<false, false, false> ZR_RPT_ARGS_TILE; \
    X __global__ void k_rpt_stc<true, false, false> ZR_RPT_ARGS_TILE; X __global__ void k_rpt_stc<false, false, false> ZR_RPT_ARGS_TILE;
// K16, textured
#define ZR_RPT_GROUP_D(X) \
    X __global__ void k_rpt_stc<true, true, false> ZR_RPT_ARGS_TILE; X __global__ void k_rpt_stc<false, true, false> ZR_RPT_ARGS_TILE;
// the material-class permutation (PLAIN = true: a scene whose material table has no metal, no transmission, no thin wall, no coat and no texture)
#define ZR_RPT_GROUP_E(X) \
    X __global__ void k_rpt_pathtrace<true, true> ZR_RPT_ARGS_TILE; X __global__ void k_rpt_pathtrace<false, true> ZR_RPT_ARGS_TILE; \
    X __global__ void k_rpt_pathtrace_w4<true, true> ZR_RPT_ARGS_TILE; X __global__ void k_rpt_pathtrace_w4<false, true> ZR_RPT_ARGS_TILE; \
    X __global__ void k_rpt_temporal<true, false, true> ZR_RPT_ARGS_TILE; X __global__ void k_rpt_temporal<false, false, true> ZR_RPT_ARGS_TILE; \
    X __global__ void k_rpt_stc<true, false, true> ZR_RPT_ARGS_TILE; X __global__ void k_rpt_stc<false, false, true> ZR_RPT_ARGS_TILE;
// (experiments build only: zr_tu_rpt_c.hip)
#define ZR_RPT_GROUP_C(X) \
    X __global__ void k_rpt_pathtrace_park<true> ZR_RPT_ARGS_TILE; X __global__ void k_rpt_pathtrace_park<false> ZR_RPT_ARGS_TILE; \
    X __global__ void k_rpt_pathtrace_coop<false> ZR_RPT_ARGS_TILE; X __global__ void k_rpt_pathtrace_coop_w4<false> ZR_RPT_ARGS_TILE; \
    X __global__ void k_rpt_pathtrace_trip<false> ZR_RPT_ARGS_TILE; X __global__ void k_rpt_pathtrace_trip_w4<false> ZR_RPT_ARGS_TILE; \
    X __global__ void k_rpt_pt_first<true> ZR_RPT_ARGS_TILE; X __global__ void k_rpt_pt_first<false> ZR_RPT_ARGS_TILE; \
    X __global__ void k_rpt_pt_next<true> ZR_RPT_ARGS_TILE; X __global__ void k_rpt_pt_next<false> ZR_RPT_ARGS_TILE;
// K13: the temporal pass's replays, emissive (B) / sun + sky (I) lighting; the spatial pass's (F)
#define ZR_RPT_GROUP_B(X) \
    X __global__ void k_rpt_replay<RPT_REPLAY_CTT, true, true> ZR_RPT_ARGS_LIST; X __global__ void k_rpt_replay<RPT_REPLAY_CTT, true, false> ZR_RPT_ARGS_LIST;
#define ZR_RPT_GROUP_I(X) \
    X __global__ void k_rpt_replay<RPT_REPLAY_CTT, false, true> ZR_RPT_ARGS_LIST; X __global__ void k_rpt_replay<RPT_REPLAY_CTT, false, false> ZR_RPT_ARGS_LIST;
#define ZR_RPT_GROUP_F(X) \
    X __global__ void k_rpt_replay<RPT_REPLAY_CTS, true, true> ZR_RPT_ARGS_LIST; X __global__ void k_rpt_replay<RPT_REPLAY_CTS, true, false> ZR_RPT_ARGS_LIST; \
    X __global__ void k_rpt_replay<RPT_REPLAY_CTS, false, true> ZR_RPT_ARGS_LIST; X __global__ void k_rpt_replay<RPT_REPLAY_CTS, false, false> ZR_RPT_ARGS_LIST;
#define ZR_RPT_GROUPS_PRODUCT(X) ZR_RPT_GROUP_A(X) ZR_RPT_GROUP_B(X) ZR_RPT_GROUP_D(X) ZR_RPT_GROUP_E(X) ZR_RPT_GROUP_F(X) ZR_RPT_GROUP_G(X) ZR_RPT_GROUP_H(X) ZR_RPT_GROUP_I(X)
