// zr_kernels_exp.h -- the K11 forms that were built, are bit-exact, and did not pay (docs/NEGATIVE_RESULTS.md): the reservoir's selected
// reconnection parked in LDS (k_rpt_pathtrace_park), a kernel per bounce with path compaction (k_rpt_pt_first / _next), the alive-lane
// diagnostic (k_rpt_pathtrace_trip) and the block-cooperative ray pool (k_rpt_pathtrace_coop).  Compiled only with -DZR_EXPERIMENTS
// (`make experiments` -> libzetaray_amd_exp.so, where ZR_K11 / ZR_K11_PARK / ZR_SEARCH select them at run time); the product library
// has neither the kernels nor the switches.  Included from the middle of zr_kernels.h.
#pragma once
// the same with the reservoir's selected reconnection parked in LDS (ZR_K11_PARK=1; zr_rpt.h RcPark)
template<bool EMISSIVE>
__global__ void __launch_bounds__(kRptBlock) ZR_WAVES(ZR_WAVES_PATHTRACE_N) k_rpt_pathtrace_park(rpt::RptFrame F, zr_frame_constants g, uint32_t tilesX, unsigned long long* counters)
{ RptPathtraceBody<EMISSIVE, false, false, true>(F, g, tilesX, counters); }

// ------------------------------------------------------------------------------------------------ K11 with per-bounce path compaction (round 3)
// On scenes where paths end early (Cornell box: open front, paths that reach the light) the megakernel's waves run every bounce with the lanes
// of the paths that are still alive -- 35 % of them on average at the bounce boundaries of the Cornell frame (zr_pass_debug_trip_stats), 95 % on
// the atrium.  Here a bounce is a kernel: k_rpt_pt_first runs the prologue and the first bounce for every pixel in 16 x 4 tiles like the
// megakernel, then the paths still alive move into consecutive slots of SoA planes (78 words = 312 B per path, rpt::PtCarry; slots come from a
// wave-aggregated atomic, so a wave's paths sit side by side and the stores coalesce); k_rpt_pt_next runs one more bounce over slots
// 0 .. count - 1 with full waves and compacts again; a path that ends writes its reservoir (PtFinishLane) from wherever it is.  Per-pixel
// arithmetic is the megakernel's, statement for statement, so the results are bit-identical; tiles whose paths can reach Russian roulette (a
// maximum over the 16 x 4 tile) stay whole, see k_rpt_pt_first.
template<bool EMISSIVE>
__device__ __forceinline__ void PtBounceAndCompact(rpt::RptFrame& F, const zr_frame_constants& g, const TravStack& stack, uint32_t* cnt, rpt::PTLane& P)
{
    rpt::PtPhaseA_Fused(F.sc, g, F.prm, stack, cnt, P);
    rpt::PtPhaseB(F.sc, F.prm, P, 0u);
    const bool alive = P.active;
    const uint32_t slot = AllocSlotWave(F.carryCount + F.carryBounce, alive);
    if (alive)
    {
        rpt::PtCarryStore st; st.p = F.carryOut + slot; st.stride = F.carryCap;
        rpt::PtCarry(st, P);
    }
    else rpt::PtFinishLane(F.gb, F.prm, F.cur, F.tex, F.finalRGBA, P);
}
template<bool EMISSIVE>
__global__ void __launch_bounds__(kRptBlock) ZR_WAVES(ZR_WAVES_PATHTRACE_N) k_rpt_pt_first(rpt::RptFrame F, zr_frame_constants g, uint32_t tilesX, unsigned long long* counters)
{
    const unsigned long long t0 = __builtin_readcyclecounter();
    F.prm.emissive = EMISSIVE ? 1u : 0u; F.prm.textured = 0u; rpt::SetMaterialClass(F, false);
    uint32_t tile, wave, lane; RptTileWaveLane(&tile, &wave, &lane);
    const uint32_t tx = tile % tilesX, ty = tile / tilesX;
    const uint32_t x = F.ox0 + tx * 16u + (lane & 15u), y = F.oy0 + ty * 16u + wave * 4u + (lane >> 4);
    ZR_TRAV_STACK_B(stack, kRptBlock);
    uint32_t cnt[2] = {0u, 0u};
    rpt::PTLane P;
    rpt::PtInitLane_Fused(F.sc, g, F.gb, F.prm, F.Owns(x, y), x, y, F.finalRGBA, stack, cnt, P);
    // Russian roulette starts at the fourth bounce, which only paths with maxNumBounces >= 4 reach (glossy-transmissive primary hits), and its
    // survival probability is the maximum over the tile's lanes that are at it: a tile with such a pixel runs the megakernel's loop as a whole
    if (F.prm.russianRoulette && __ballot(P.valid && P.maxNumBounces >= 4) != 0)
    {
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
            rpt::PtPhaseB(F.sc, F.prm, P, key);
        }
        rpt::PtFinishLane(F.gb, F.prm, F.cur, F.tex, F.finalRGBA, P);
    }
    else if (__ballot(P.valid) != 0) PtBounceAndCompact<EMISSIVE>(F, g, stack, cnt, P);
    FlushRayCountersCost(F, counters, cnt, x, y, F.Owns(x, y), t0);
}
template<bool EMISSIVE>
__global__ void __launch_bounds__(kRptBlock) ZR_WAVES(ZR_WAVES_PATHTRACE_N) k_rpt_pt_next(rpt::RptFrame F, zr_frame_constants g, uint32_t tilesX, unsigned long long* counters)
{
    const unsigned long long t0 = __builtin_readcyclecounter();
    F.prm.emissive = EMISSIVE ? 1u : 0u; F.prm.textured = 0u; rpt::SetMaterialClass(F, false);
    const uint32_t n = F.carryCount[F.carryBounce - 1u];
    const uint32_t slot = blockIdx.x * kRptBlock + threadIdx.x;
    if (blockIdx.x * kRptBlock >= n) return;
    ZR_TRAV_STACK_B(stack, kRptBlock);
    uint32_t cnt[2] = {0u, 0u};
    rpt::PTLane P;
    P.valid = false; P.active = false; P.atRR = false; P.x = 0; P.y = 0;
    if (slot < n)
    {
        rpt::PtCarryLoad ld; ld.p = F.carryIn + slot; ld.stride = F.carryCap;
        rpt::PtCarry(ld, P);
    }
    PtBounceAndCompact<EMISSIVE>(F, g, stack, cnt, P);
    FlushRayCountersCost(F, counters, cnt, P.x, P.y, P.valid, t0);
}

// ------------------------------------------------------------------------------------------------ alive-lane diagnostic (round 3)
// The megakernel with counters at its bounce boundaries: lanes alive / lane slots of the waves that pass a boundary (zr_pass_debug_trip_stats) --
// what compaction between bounces could win back: 0.35 on the Cornell frame, 0.95 on the atrium.  It also sends the carried state (rpt::PtCarry)
// through memory and back at each boundary; results are unchanged, but the kernel's TIME means nothing: in the middle of the loop the 78 carried
// words all become live at one point and go through scratch (624 -> 1232 B per lane, 0.95 -> 2.6 ms).  What carrying the state really costs is
// measured by the kernels above.  ZR_K11=trip, emissive untextured permutation.
template<bool NODE_CACHE>
__device__ __forceinline__ void RptPathtraceBodyTrip(rpt::RptFrame& F, const zr_frame_constants& g, uint32_t tilesX, unsigned long long* counters)
{
    const unsigned long long t0 = __builtin_readcyclecounter();
    F.prm.emissive = 1u; F.prm.textured = 0u; rpt::SetMaterialClass(F, false);
    uint32_t tile, wave, lane; RptTileWaveLane(&tile, &wave, &lane);
    const uint32_t tx = tile % tilesX, ty = tile / tilesX;
    const uint32_t x = F.ox0 + tx * 16u + (lane & 15u), y = F.oy0 + ty * 16u + wave * 4u + (lane >> 4);
    ZR_TRAV_STACK_B(stack, kRptBlock);
    if (NODE_CACHE) { ZR_NODE_CACHE_FILL(stack, F.sc, kRptBlock); }
    uint32_t cnt[2] = {0u, 0u};
    rpt::PTLane P;
    rpt::PtInitLane_Fused(F.sc, g, F.gb, F.prm, F.Owns(x, y), x, y, F.finalRGBA, stack, cnt, P);
    uint32_t alive = 0, slots = 0, words = 0;
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
        rpt::PtPhaseB(F.sc, F.prm, P, key);
        // ---- the bounce boundary
        const uint64_t live = __ballot(P.active);
        if (live != 0) { alive += (uint32_t)__popcll((unsigned long long)live); slots += 64u; }
        if (P.active)
        {
            rpt::PtCarryStore st; st.p = F.trip + rpt::Pix(F.gb, x, y); st.stride = F.tripStride;
            rpt::PtCarry(st, P);
            words = st.n;
        }
        __asm__ volatile("" ::: "memory");
        if (P.active)
        {
            rpt::PtCarryLoad ld; ld.p = F.trip + rpt::Pix(F.gb, x, y); ld.stride = F.tripStride;
            rpt::PtCarry(ld, P);
        }
    }
    rpt::PtFinishLane(F.gb, F.prm, F.cur, F.tex, F.finalRGBA, P);
    if (lane == 0 && slots != 0)
    {
        atomicAdd(F.tripStats + 0, (unsigned long long)alive); atomicAdd(F.tripStats + 1, (unsigned long long)slots);
    }
    if (words != 0) F.tripStats[2] = words;
    FlushRayCountersCost(F, counters, cnt, x, y, F.Owns(x, y), t0);
}
template<bool UNUSED>
__global__ void __launch_bounds__(kRptBlock) ZR_WAVES(ZR_WAVES_PATHTRACE_N) k_rpt_pathtrace_trip(rpt::RptFrame F, zr_frame_constants g, uint32_t tilesX, unsigned long long* counters)
{ RptPathtraceBodyTrip<false>(F, g, tilesX, counters); }
template<bool UNUSED>
__global__ void __launch_bounds__(kRptBlock) ZR_WAVES(ZR_WAVES_PATHTRACE_LARGE_N) k_rpt_pathtrace_trip_w4(rpt::RptFrame F, zr_frame_constants g, uint32_t tilesX, unsigned long long* counters)
{ RptPathtraceBodyTrip<true>(F, g, tilesX, counters); }

// ------------------------------------------------------------------------------------------------ block-cooperative ray pool (round 3)
// The inline traversals of K11 ran at 23 % lane utilisation (section profile): a query is entered by the 29 of 64 lanes whose
// shading branch needs it, and the call lasts until its slowest ray is done (10.5 vote iterations for rays that need 5.2 steps).  Here the
// lanes of a 256-thread block put their rays into an LDS pool instead; after a barrier a FEW waves (one per 64 x ZR_POOL_RAYS_PER_LANE rays)
// traverse the pooled rays with every lane busy: a lane whose ray is finished stores the hit into the ray's slot and takes the next ray from
// the pool (wave-aggregated LDS atomic), so node / triangle phases run with nearly full waves until the pool is empty.  The other waves
// wait at the barrier and issue nothing -- the kernel is VALU-issue bound, idle waves cost registers but no issue slots.  The traversing
// waves rotate from call to call so that the work spreads over the four SIMDs of the CU.  Results are those of Traverse: closest hit with
// the index tie-break / any hit do not depend on who traverses a ray or in which order (zr_intersect.h).
#ifndef ZR_POOL_RAYS_PER_LANE
#define ZR_POOL_RAYS_PER_LANE 2
#endif
#ifndef ZR_POOL_REFILL_MIN
#define ZR_POOL_REFILL_MIN 8          // idle lanes before a wave goes back to the pool (always when no lane is busy)
#endif
static constexpr int kPoolWords = 10;   // o.xyz, d.xyz, tmin, tmax, mask | anyHit << 8 | filterID << 9, ignoreID; the hit (t, u, v, tri) overwrites words 0-3
template<int B> struct RayPool
{
    ZR_LDS_AS uint32_t* words;      // [2][kPoolWords][B], ping-pong between consecutive calls
    ZR_LDS_AS uint32_t* ctr;        // [2][2]: rays deposited, rays taken
    uint32_t phase;                 // calls made so far (block-uniform)
};
#define ZR_RAY_POOL(name, B) \
    __shared__ uint32_t name##Words[2 * kPoolWords * (B)]; __shared__ uint32_t name##Ctr[4]; \
    RayPool<B> name; name.words = (ZR_LDS_AS uint32_t*)name##Words; name.ctr = (ZR_LDS_AS uint32_t*)name##Ctr; name.phase = 0; \
    if (threadIdx.x < 4) name##Ctr[threadIdx.x] = 0; \
    __syncthreads()

// one wave's share of a pooled trace: traverse rays [*, n) of the pool until it is empty, all lanes refilling from the shared cursor
template<int B>
__device__ __forceinline__ void PoolTraverse(const SceneView& sc, ZR_LDS_AS uint32_t* w, ZR_LDS_AS uint32_t* next, uint32_t n, const TravStack& stack)
{
    const uint32_t lane = __lane_id();
    const uint64_t lt = (1ull << lane) - 1ull;
    TravState s; TravLane L; L.triCur = 0; L.triEnd = 0; L.done = true;
    s.sp = 0; s.cur = 0; s.filterID = false; s.ignoreID = 0; s.mask = 0; s.tmin = 0; s.tmax = 0; s.o = v3(0.0f); s.d = v3(0.0f); s.idx = 0; s.idy = 0; s.idz = 0;
    s.best.t = 0; s.best.u = 0; s.best.v = 0; s.best.tri = kInvalidTri;
    bool has = false, anyH = false; uint32_t slot = 0;
    bool more = true;       // wave-uniform: the pool may still hold rays
    for (;;)
    {
        // a finished ray: its hit goes into its slot
        if (has && L.done)
        {
            w[0 * B + slot] = zr_asuint(s.best.t); w[1 * B + slot] = zr_asuint(s.best.u); w[2 * B + slot] = zr_asuint(s.best.v); w[3 * B + slot] = s.best.tri;
            has = false;
        }
        const uint64_t mIdle = __ballot(!has);
        if (more && mIdle != 0 && ((uint32_t)__popcll(mIdle) >= (uint32_t)ZR_POOL_REFILL_MIN || mIdle == ~0ull))
        {
            const uint32_t cnt = (uint32_t)__popcll(mIdle);
            const int leader = __ffsll((long long)mIdle) - 1;
            uint32_t base = 0;
            if ((int)lane == leader) base = atomicAdd((uint32_t*)next, cnt);
            base = __shfl(base, leader);
            const uint32_t idx = base + (uint32_t)__popcll(mIdle & lt);
            if (!has && idx < n)
            {
                const V3 o = v3(zr_asfloat(w[0 * B + idx]), zr_asfloat(w[1 * B + idx]), zr_asfloat(w[2 * B + idx]));
                const V3 d = v3(zr_asfloat(w[3 * B + idx]), zr_asfloat(w[4 * B + idx]), zr_asfloat(w[5 * B + idx]));
                const float tmin = zr_asfloat(w[6 * B + idx]), tmax = zr_asfloat(w[7 * B + idx]);
                const uint32_t fl = w[8 * B + idx], ign = w[9 * B + idx];
                TravInit(sc, s, o, d, tmin, tmax, fl & 0xffu, (fl & 0x200u) != 0, ign);
                anyH = (fl & 0x100u) != 0;
                L.triCur = 0; L.triEnd = 0; L.done = false;
                TravEnter(sc, s, L, s.cur);
                has = true; slot = idx;
            }
            if (base + cnt >= n) more = false;
        }
        const bool atTri = has && L.triCur < L.triEnd;
        const bool atNode = has && !L.done && !atTri;
        const uint64_t mNode = __ballot(atNode), mTri = __ballot(atTri);
        if ((mNode | mTri) == 0)
        {
            if (__ballot(has) == 0 && !more) break;
            continue;       // hits to store and / or rays to fetch
        }
        if (ZR_VOTE_WN * __popcll(mNode) >= ZR_VOTE_WT * __popcll(mTri)) { if (atNode) TravNodePhase(sc, s, L, stack); }
        else { if (atTri) TravTriPhase(sc, s, L, stack, anyH, false); }
    }
}

// All B threads of the block call this together (uniform control flow).  q.want == false: no ray, the result is "no hit".
template<int B>
__device__ __forceinline__ RawHit BlockTrace(const SceneView& sc, const rpt::TraceReq& q, const TravStack& stack, RayPool<B>& pool)
{
    constexpr uint32_t W = B / 64;
    const uint32_t lane = __lane_id(), wave = threadIdx.x >> 6;
    const uint32_t pp = pool.phase & 1u;
    ZR_LDS_AS uint32_t* w = pool.words + pp * (kPoolWords * B);
    ZR_LDS_AS uint32_t* ctr = pool.ctr + pp * 2u;
    // 1. deposit: one LDS atomic per wave
    const uint64_t m = __ballot(q.want);
    uint32_t slot = 0;
    if (m != 0)
    {
        const int leader = __ffsll((long long)m) - 1;
        uint32_t base = 0;
        if ((int)lane == leader) base = atomicAdd((uint32_t*)ctr, (uint32_t)__popcll(m));
        base = __shfl(base, leader);
        slot = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if (q.want)
        {
            w[0 * B + slot] = zr_asuint(q.o.x); w[1 * B + slot] = zr_asuint(q.o.y); w[2 * B + slot] = zr_asuint(q.o.z);
            w[3 * B + slot] = zr_asuint(q.d.x); w[4 * B + slot] = zr_asuint(q.d.y); w[5 * B + slot] = zr_asuint(q.d.z);
            w[6 * B + slot] = zr_asuint(q.tmin); w[7 * B + slot] = zr_asuint(q.tmax);
            w[8 * B + slot] = (q.mask & 0xffu) | (q.anyHit ? 0x100u : 0u) | (q.filterID ? 0x200u : 0u); w[9 * B + slot] = q.ignoreID;
        }
    }
    __syncthreads();
    const uint32_t n = ((volatile ZR_LDS_AS uint32_t*)ctr)[0];
    // the other buffer's counters: its last readers passed the barrier above one call ago, its next writers come after the barrier below
    if (threadIdx.x == 0) { pool.ctr[(pp ^ 1u) * 2u] = 0; pool.ctr[(pp ^ 1u) * 2u + 1u] = 0; }
    // 2. a few waves traverse everything
    if (n != 0)
    {
        uint32_t k = (n + 64u * ZR_POOL_RAYS_PER_LANE - 1u) / (64u * ZR_POOL_RAYS_PER_LANE);
        k = k > W ? W : k;
        const uint32_t rot = (wave + W - (pool.phase % W)) % W;
        if (rot < k) PoolTraverse<B>(sc, w, ctr + 1, n, stack);
    }
    __syncthreads();
    // 3. pick the hit up
    RawHit h; h.t = 0; h.u = 0; h.v = 0; h.tri = kInvalidTri;
    if (q.want) { h.t = zr_asfloat(w[0 * B + slot]); h.u = zr_asfloat(w[1 * B + slot]); h.v = zr_asfloat(w[2 * B + slot]); h.tri = w[3 * B + slot]; }
    pool.phase++;
    return h;
}

// K11 with pooled traces (emissive-NEE variant; zr_rpt.h: the stage functions cut at their BVH queries).  Block = the 16 x 16 tile, wave w =
// rows 4w .. 4w+3 like k_rpt_pathtrace; the bounce loop runs until every wave of the block is done (finished waves keep tracing for the others).
static constexpr int kCoopBlock = 256;
#ifndef ZR_K11_PARK_DEFAULT
#define ZR_K11_PARK_DEFAULT 0     // 1 = k_rpt_pathtrace_park (RcPark) for scenes that take the 3-wave build; ZR_K11_PARK=0|1 overrides at run time
#endif
#ifndef ZR_K11_DEFAULT
#define ZR_K11_DEFAULT 0          // 0 = the inline megakernel, 1 = pooled traces; ZR_K11=inline|pool overrides at run time
#endif
template<bool TEX>
__device__ __forceinline__ void RptPathtraceBodyCoop(rpt::RptFrame& F, const zr_frame_constants& g, uint32_t tilesX, unsigned long long* counters)
{
    const unsigned long long t0 = __builtin_readcyclecounter();
    F.prm.emissive = 1u; F.prm.textured = TEX ? 1u : 0u; rpt::SetMaterialClass(F, false);
    const uint32_t tile = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t tx = tile % tilesX, ty = tile / tilesX;
    const uint32_t x = F.ox0 + tx * 16u + (lane & 15u), y = F.oy0 + ty * 16u + wave * 4u + (lane >> 4);
    ZR_TRAV_STACK_B(stack, kCoopBlock);
    ZR_RAY_POOL(pool, kCoopBlock);
    ZR_PROF_KERNEL(F.sc, 1);
    uint32_t cnt[2] = {0u, 0u};
    rpt::PTLane P;
    rpt::TraceReq q0;
    rpt::PtInitLane_Pre(F.sc, g, F.gb, F.prm, F.Owns(x, y), x, y, F.finalRGBA, cnt, P, q0);
    {
        const RawHit h0 = BlockTrace<kCoopBlock>(F.sc, q0, stack, pool);
        rpt::PtInitLane_Post(F.sc, F.prm, P, q0, h0);
    }
    for (;;)
    {
        if (!__syncthreads_or(P.active ? 1 : 0)) break;
        rpt::PtMid M;
        rpt::PtPhaseA_Pre(F.sc, g, F.prm, cnt, P, M);
        const RawHit h1 = BlockTrace<kCoopBlock>(F.sc, M.q1, stack, pool);
        rpt::PtPhaseA_Mid(F.sc, g, F.prm, cnt, P, M, h1);
        const RawHit h2 = BlockTrace<kCoopBlock>(F.sc, M.q2, stack, pool);
        rpt::PtPhaseA_Post(F.sc, g, F.prm, cnt, P, M, h2);
        uint32_t key = rpt::PtRRKey(P);
        if (__ballot(key != 0) != 0)
        {
            for (int s = 1; s < 64; s <<= 1) { uint32_t o = __shfl_xor(key, s); key = o > key ? o : key; }
        }
        rpt::PtPhaseB(F.sc, F.prm, P, key);
    }
    rpt::PtFinishLane(F.gb, F.prm, F.cur, F.tex, F.finalRGBA, P);
    FlushRayCountersCost(F, counters, cnt, x, y, F.Owns(x, y), t0);
}
#ifndef ZR_WAVES_PATHTRACE_COOP
#define ZR_WAVES_PATHTRACE_COOP ZR_WAVES(3)
#endif
template<bool TEX>
__global__ void __launch_bounds__(kCoopBlock) ZR_WAVES_PATHTRACE_COOP k_rpt_pathtrace_coop(rpt::RptFrame F, zr_frame_constants g, uint32_t tilesX, unsigned long long* counters)
{ RptPathtraceBodyCoop<TEX>(F, g, tilesX, counters); }
template<bool TEX>
__global__ void __launch_bounds__(kCoopBlock) ZR_WAVES(4) k_rpt_pathtrace_coop_w4(rpt::RptFrame F, zr_frame_constants g, uint32_t tilesX, unsigned long long* counters)
{ RptPathtraceBodyCoop<TEX>(F, g, tilesX, counters); }
