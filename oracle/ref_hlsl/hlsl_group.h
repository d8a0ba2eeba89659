// ORACLE tooling -- test infrastructure only.
//
// hlsl_group.h: thread-group execution of a reference compute shader compiled as C++ (see hlsl_shim.h / hlsl_rt.h).
// Every thread of a group runs as a fiber (ucontext); wave intrinsics and group barriers are rendezvous points: a lane that
// reaches one parks, and once every live lane of its wave (or group) has parked the operation is evaluated over the parked lanes'
// operands and the lanes continue.  A wave = 64 consecutive threads by SV_GroupIndex (wave64 hardware; DESIGN.md 5.5: the 8 x 8
// group, or a 16 x 4 block of the 16 x 8 ReSTIR PT groups).  Floating-point WaveActiveSum is the ABI's canonical 64-lane xor
// butterfly (strides 1 ... 32, absent lanes contribute +0), max / min / integer sums are order-free.
#pragma once
#include <ucontext.h>
#include <vector>
#include <functional>
#include "hlsl_rt.h"

namespace hlsl {

struct GroupRunner
{
    int kWave = 64;                          // lanes per wave: 64 (wave64 hardware) unless the shader pins [WaveSize(32)] (EstimateTriEmissivePower.hlsl)
    bool sumAscending = false;               // float WaveActiveSum in ascending lane order instead of the xor butterfly (the ABI's definition for K2)
    static constexpr size_t kStack = 1 << 20;
    enum Scope { NONE, WAVE, GROUP };
    struct Lane
    {
        ucontext_t ctx; std::vector<char> stack; bool done = true; Scope waiting = NONE; bool runnable = false;
        uint32_t u[4] = {0, 0, 0, 0};           // this lane's operand of the pending cross-lane operation
        int site = 0;                           // reconvergence rank of the operation the lane is parked at (see Run)
    };
    std::vector<Lane> lanes; ucontext_t sched; int cur = -1; int n = 0;
    // operands of the lanes of one wave at the last rendezvous (stable while the lanes consume them)
    std::vector<uint32_t> snapU; std::vector<uint8_t> snapActive;
    std::function<void(int)> body;

    static GroupRunner*& Current() { static thread_local GroupRunner* g = nullptr; return g; }
    static void Trampoline(int lane) { GroupRunner* g = Current(); g->body(lane); FlushPendingRW(); g->lanes[lane].done = true; g->lanes[lane].waiting = NONE; swapcontext(&g->lanes[lane].ctx, &g->sched); }

    void Run(int numThreads, std::function<void(int)> f)
    {
        Current() = this; body = std::move(f); n = numThreads;
        if ((int)lanes.size() < n) lanes.resize(n);
        snapU.assign((size_t)n * 4, 0u); snapActive.assign(n, 0);
        for (int i = 0; i < n; i++)
        {
            Lane& L = lanes[i];
            if (L.stack.size() != kStack) L.stack.resize(kStack);
            getcontext(&L.ctx); L.ctx.uc_stack.ss_sp = L.stack.data(); L.ctx.uc_stack.ss_size = kStack; L.ctx.uc_link = &sched;
            makecontext(&L.ctx, (void (*)())Trampoline, 1, i);
            L.done = false; L.waiting = NONE; L.runnable = true;
        }
        for (;;)
        {
            bool ran = false;
            for (int i = 0; i < n; i++)
                if (!lanes[i].done && lanes[i].runnable) { lanes[i].runnable = false; cur = i; swapcontext(&sched, &lanes[i].ctx); ran = true; }
            // release waves whose live lanes all parked at a wave rendezvous
            bool released = false;
            for (int w0 = 0; w0 < n; w0 += kWave)
            {
                const int w1 = w0 + kWave < n ? w0 + kWave : n;
                bool all = true, any = false;
                for (int i = w0; i < w1; i++) if (!lanes[i].done) { any = true; if (lanes[i].waiting != WAVE) all = false; }
                if (!any || !all) continue;
                // SIMT reconvergence: lanes parked at an operation inside a loop (the Russian-roulette WaveActiveMax of the bounce loops,
                // rank 0) run on -- with exactly those lanes as the active mask -- while lanes that already left the loop wait at their
                // operation after it (rank 1) until everyone has arrived, like a masked-off lane waits at the loop's exit on hardware
                int rank = 1 << 30;
                for (int i = w0; i < w1; i++) if (!lanes[i].done && lanes[i].site < rank) rank = lanes[i].site;
                for (int i = w0; i < w1; i++)
                {
                    const bool go = !lanes[i].done && lanes[i].site == rank;
                    snapActive[i] = go;
                    for (int k = 0; k < 4; k++) snapU[4 * i + k] = go ? lanes[i].u[k] : 0u;
                    if (go) { lanes[i].waiting = NONE; lanes[i].runnable = true; }
                }
                released = true;
            }
            // group barrier: every live lane of the group parked at it
            bool allG = true, anyG = false;
            for (int i = 0; i < n; i++) if (!lanes[i].done) { anyG = true; if (lanes[i].waiting != GROUP) allG = false; }
            if (anyG && allG) { for (int i = 0; i < n; i++) if (!lanes[i].done) { lanes[i].waiting = NONE; lanes[i].runnable = true; } released = true; }
            if (!anyG) break;
            if (!ran && !released) { std::fprintf(stderr, "GroupRunner: deadlock (lanes parked at different rendezvous)\n"); std::abort(); }
        }
        Current() = nullptr;
    }
    // called from inside a fiber
    void Park(Scope s) { FlushPendingRW(); Lane& L = lanes[cur]; const int me = cur; L.waiting = s; swapcontext(&L.ctx, &sched); cur = me; }
    int WaveBase() const { return (cur / kWave) * kWave; }
};

static inline GroupRunner* GR() { return GroupRunner::Current(); }
static inline uint32_t WaveGetLaneCount() { GroupRunner* g = GR(); return g ? (uint32_t)g->kWave : 64u; }
static inline uint32_t WaveGetLaneIndex() { GroupRunner* g = GR(); return g ? (uint32_t)(g->cur % g->kWave) : 0u; }
static inline void GroupMemoryBarrierWithGroupSync() { if (GroupRunner* g = GR()) g->Park(GroupRunner::GROUP); }
static inline void GroupMemoryBarrier() {}
static inline void DeviceMemoryBarrier() {}
static inline void AllMemoryBarrierWithGroupSync() { GroupMemoryBarrierWithGroupSync(); }

// park with a 4-dword operand; afterwards snapU / snapActive of this lane's wave hold every participant's operand
static inline void WaveRendezvous(const uint32_t* u, int nwords, int site = 1)
{
    GroupRunner* g = GR();
    GroupRunner::Lane& L = g->lanes[g->cur];
    L.site = site;
    for (int k = 0; k < 4; k++) L.u[k] = k < nwords ? u[k] : 0u;
    g->Park(GroupRunner::WAVE);
}
static inline float ButterflySum64(const float* v)    // v[64], absent lanes hold +0
{
    float a[64], b[64];
    for (int i = 0; i < 64; i++) a[i] = v[i];
    for (int s = 1; s < 64; s <<= 1) { for (int i = 0; i < 64; i++) b[i] = a[i] + a[i ^ s]; for (int i = 0; i < 64; i++) a[i] = b[i]; }
    return a[0];
}
static inline float WaveActiveSum(float x)
{
    GroupRunner* g = GR(); if (!g) return x;
    uint32_t u = zr_asuint(x); WaveRendezvous(&u, 1);
    float v[64]; const int b = g->WaveBase();
    for (int i = 0; i < 64; i++) v[i] = (i < g->kWave && b + i < g->n && g->snapActive[b + i]) ? zr_asfloat(g->snapU[4 * (b + i)]) : 0.0f;
    if (g->sumAscending) { float acc = 0.0f; for (int i = 0; i < g->kWave; i++) acc += v[i]; return acc; }
    return ButterflySum64(v);
}
static inline uint32_t WaveActiveSum(uint32_t x)
{
    GroupRunner* g = GR(); if (!g) return x;
    WaveRendezvous(&x, 1);
    uint32_t s = 0; const int b = g->WaveBase();
    for (int i = 0; i < g->kWave && b + i < g->n; i++) if (g->snapActive[b + i]) s += g->snapU[4 * (b + i)];
    return s;
}
static inline int WaveActiveSum(int x) { return (int)WaveActiveSum((uint32_t)x); }
// WaveMatch: per lane, the mask of the wave's active lanes holding the same value (64 lanes: .x = lanes 0-31, .y = 32-63)
static inline uint4 WaveMatch(uint32_t x)
{
    GroupRunner* g = GR(); if (!g) return uint4(1u, 0u, 0u, 0u);
    WaveRendezvous(&x, 1);
    uint32_t m[4] = {0, 0, 0, 0}; const int b = g->WaveBase();
    for (int i = 0; i < g->kWave && b + i < g->n; i++) if (g->snapActive[b + i] && g->snapU[4 * (b + i)] == x) m[i >> 5] |= 1u << (i & 31);
    return uint4(m[0], m[1], m[2], m[3]);
}
static inline uint32_t WaveActiveSum(bool x) { return WaveActiveSum((uint32_t)(x ? 1u : 0u)); }
static inline uint16_t WaveActiveSum(uint16_t x) { return (uint16_t)WaveActiveSum((uint32_t)x); }
static inline float3 WaveActiveSum(const float3& x) { return float3(WaveActiveSum(x.x), WaveActiveSum(x.y), WaveActiveSum(x.z)); }
static inline float WaveActiveMax(float x)
{
    GroupRunner* g = GR(); if (!g) return x;
    uint32_t u = zr_asuint(x); WaveRendezvous(&u, 1, 0);
    const int b = g->WaveBase(); bool first = true; float m = 0.0f;
    for (int i = 0; i < g->kWave && b + i < g->n; i++) if (g->snapActive[b + i]) { float v = zr_asfloat(g->snapU[4 * (b + i)]); m = first ? v : zr_max(m, v); first = false; }
    return m;
}
static inline uint32_t WaveActiveMax(uint32_t x)
{
    GroupRunner* g = GR(); if (!g) return x;
    WaveRendezvous(&x, 1);
    const int b = g->WaveBase(); uint32_t m = 0;
    for (int i = 0; i < g->kWave && b + i < g->n; i++) if (g->snapActive[b + i]) m = g->snapU[4 * (b + i)] > m ? g->snapU[4 * (b + i)] : m;
    return m;
}
static inline bool WaveActiveAnyTrue(bool x) { return WaveActiveSum((uint32_t)(x ? 1u : 0u)) != 0; }
static inline bool WaveActiveAllTrue(bool x) { return WaveActiveSum((uint32_t)(x ? 0u : 1u)) == 0; }
static inline uint32_t WaveActiveCountBits(bool x) { return WaveActiveSum((uint32_t)(x ? 1u : 0u)); }
static inline uint32_t WavePrefixSum(uint32_t x)
{
    GroupRunner* g = GR(); if (!g) return 0;
    WaveRendezvous(&x, 1);
    const int b = g->WaveBase(), me = g->cur - b; uint32_t s = 0;
    for (int i = 0; i < me; i++) if (g->snapActive[b + i]) s += g->snapU[4 * (b + i)];
    return s;
}
static inline uint16_t WavePrefixSum(uint16_t x) { return (uint16_t)WavePrefixSum((uint32_t)x); }
static inline int WavePrefixSum(int x) { return (int)WavePrefixSum((uint32_t)x); }
static inline uint32_t WavePrefixCountBits(bool x) { return WavePrefixSum((uint32_t)(x ? 1u : 0u)); }
static inline bool WaveIsFirstLane()
{
    GroupRunner* g = GR(); if (!g) return true;
    uint32_t z = 0; WaveRendezvous(&z, 1);
    const int b = g->WaveBase();
    for (int i = 0; i < g->kWave && b + i < g->n; i++) if (g->snapActive[b + i]) return b + i == g->cur;
    return true;
}
template<class T> static inline T WaveReadLaneAt(T x, uint32_t lane)
{
    static_assert(sizeof(T) <= 16, "operand too large");
    GroupRunner* g = GR(); if (!g) return x;
    uint32_t u[4] = {0, 0, 0, 0}; memcpy(u, &x, sizeof(T)); WaveRendezvous(u, 4);
    T r; memcpy(&r, &g->snapU[4 * (g->WaveBase() + (int)lane)], sizeof(T)); return r;
}
template<class T> static inline T WaveReadLaneFirst(T x)
{
    GroupRunner* g = GR(); if (!g) return x;
    uint32_t u[4] = {0, 0, 0, 0}; memcpy(u, &x, sizeof(T)); WaveRendezvous(u, 4);
    const int b = g->WaveBase();
    for (int i = 0; i < g->kWave && b + i < g->n; i++) if (g->snapActive[b + i]) { T r; memcpy(&r, &g->snapU[4 * (b + i)], sizeof(T)); return r; }
    return x;
}
// atomics on groupshared / UAV memory: lanes of a group are fibers of one OS thread, so plain read-modify-write is atomic here
template<class T, class V> static inline void InterlockedAdd(T& dst, V v) { dst = (T)(dst + (T)v); }
template<class T, class V, class O> static inline void InterlockedAdd(T& dst, V v, O& original) { original = (O)dst; dst = (T)(dst + (T)v); }
template<class T, class V> static inline void InterlockedMax(T& dst, V v) { dst = dst > (T)v ? dst : (T)v; }
template<class T, class V> static inline void InterlockedMin(T& dst, V v) { dst = dst < (T)v ? dst : (T)v; }
template<class T, class V> static inline void InterlockedOr(T& dst, V v) { dst = (T)(dst | (T)v); }

} // namespace hlsl
