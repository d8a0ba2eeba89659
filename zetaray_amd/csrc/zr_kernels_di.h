// zr_kernels_di.h -- the direct-lighting (K5 / K6 emissive ReSTIR DI, K7 / K8 sun + sky ReSTIR DI) and ReSTIR GI (K10) kernels, shared by zr_api.hip
// (launches) and the translation units that instantiate them: zr_tu_di.hip (PLAIN = false, + the TEXTURED K10) and zr_tu_di_e.hip (PLAIN = true)
#pragma once
#include "zr_kernels.h"
#include "zr_rdi.h"
#include "zr_sdi.h"
#include "zr_rgi.h"
static constexpr int kSdiBlock = 256, kDiBlock = 64;
// PLAIN: the scene's material class (zr_rpt.h SetMaterialClass) as a compile-time fact in the frame's scene + G-buffer views
template<class Frame> __device__ __forceinline__ void SetMaterialClassDi(Frame& F, bool plain)
{ F.sc.plain = plain; F.scPrev.plain = plain; F.gb.plain = plain; F.gbPrev.plain = plain; }
__device__ __forceinline__ void SetMaterialClassGi(rgi::GiFrame& F, bool plain) { F.sc.plain = plain; F.gb.plain = plain; F.gbPrev.plain = plain; }

// ------------------------------------------------------------------------------------------------ sun + sky ReSTIR DI kernels
// threads per block (see kRptBlock in zr_kernels.h; scripts/gpu_block3.sh [rounds 1-4: git history up to 31e92fa; today scripts/gpu.sh ab]): one-wave blocks pay for the emissive DI kernels (K5 0.580 -> 0.568 ms
// Cornell, 5.42 -> 5.12 ms atrium; K6 0.258 -> 0.241 / 2.99 -> 2.61 ms), not for the sun + sky ones (K7 / K8 within +-0.6 %)
// K7: initial candidates (sun, cosine-sky, BSDF-sky) + temporal reuse; K8: pairwise-MIS spatial reuse.  One thread per pixel.
template<bool PLAIN>
__global__ void __launch_bounds__(kSdiBlock) ZR_WAVES_SDI_T k_sdi_temporal(sdi::SkyFrame F, zr_frame_constants g, uint32_t tilesX, unsigned long long* counters)
{
    SetMaterialClassDi(F, PLAIN);
    uint32_t x, y; PixelOfThreadB<kSdiBlock>(tilesX, F.ox0, F.oy0, &x, &y);
    ZR_TRAV_STACK_B(stack, kSdiBlock);
    uint32_t cnt[2] = {0u, 0u};
    if (F.Owns(x, y)) sdi::TemporalPixel(F, g, x, y, stack, cnt);
    FlushRayCounters(counters, cnt);
}
template<bool PLAIN>
__global__ void __launch_bounds__(kSdiBlock) ZR_WAVES_SDI_S k_sdi_spatial(sdi::SkyFrame F, zr_frame_constants g, uint32_t tilesX, unsigned long long* counters)
{
    SetMaterialClassDi(F, PLAIN);
    uint32_t x, y; PixelOfThreadB<kSdiBlock>(tilesX, F.ox0, F.oy0, &x, &y);
    ZR_TRAV_STACK_B(stack, kSdiBlock);
    uint32_t cnt[2] = {0u, 0u};
    if (F.Owns(x, y)) sdi::SpatialPixel(F, g, x, y, stack, cnt);
    FlushRayCounters(counters, cnt);
}

// ------------------------------------------------------------------------------------------------ ReSTIR DI kernels
// K5: initial candidates + temporal reuse, one thread per pixel (8x8 quadrant per wave, like the reference's thread group)
// HVS: USE_HALF_VECTOR_COPY_SHIFT (ReSTIR_DI/Params.hlsli:12) as a compile-time fact, so the default build keeps the code it had before the shift existed
template<bool PLAIN, bool HVS>
__global__ void __launch_bounds__(kDiBlock) ZR_WAVES_RDI_T k_rdi_temporal(rdi::DiFrame F, zr_frame_constants g, uint32_t tilesX, unsigned long long* counters)
{
    SetMaterialClassDi(F, PLAIN);
    F.prm.halfVec = HVS ? 1u : 0u;
    uint32_t x, y; PixelOfThreadB<kDiBlock>(tilesX, F.ox0, F.oy0, &x, &y);
    ZR_TRAV_STACK_B(stack, kDiBlock);
    uint32_t cnt[2] = {0u, 0u};
    if (F.Owns(x, y)) rdi::TemporalPixel(F, g, x, y, stack, cnt);
    FlushRayCounters(counters, cnt);
}
// K6: spatial reuse with pairwise MIS; WaveActiveSum(disoccluded) = popcount of a ballot over the 8x8 group
template<bool PLAIN, bool HVS>
__global__ void __launch_bounds__(kDiBlock) ZR_WAVES_RDI_S k_rdi_spatial(rdi::DiFrame F, zr_frame_constants g, uint32_t tilesX, unsigned long long* counters)
{
    SetMaterialClassDi(F, PLAIN);
    F.prm.halfVec = HVS ? 1u : 0u;
    uint32_t x, y; PixelOfThreadB<kDiBlock>(tilesX, F.ox0, F.oy0, &x, &y);
    ZR_TRAV_STACK_B(stack, kDiBlock);
    uint32_t cnt[2] = {0u, 0u};
    rdi::SpatialLane a;
    rdi::SpatialPhase0(F, g, x, y, a);
    const uint32_t waveDisoccluded = (uint32_t)__popcll(__ballot(a.disoccluded));
    rdi::SpatialPhase1(F, g, a, waveDisoccluded, stack, cnt);
    FlushRayCounters(counters, cnt);
}

// ------------------------------------------------------------------------------------------------ ReSTIR GI kernel
// K10: one 8x8 pixel group per wave, bounce loop in lockstep around the Russian-roulette wave max, then temporal resampling
// and the boiling-suppression wave sum (zr_rgi.h)
// (the kernel body as a macro: routing both kernels through one inline function taking the frame by reference cost the untextured one 4 %)
#ifdef ZR_NODE_CACHE_MORE
#define ZR_RGI_NODE_CACHE ZR_NODE_CACHE_FILL(stack, F.sc, kRgiBlock);
#else
#define ZR_RGI_NODE_CACHE
#endif
#define ZR_RGI_KERNEL_BODY(TEX, PLAIN) \
    F.prm.textured = TEX; \
    SetMaterialClassGi(F, PLAIN); \
    uint32_t x, y; PixelOfThreadB<kRgiBlock>(tilesX, F.ox0, F.oy0, &x, &y); \
    ZR_TRAV_STACK_B(stack, kRgiBlock); \
    ZR_RGI_NODE_CACHE \
    uint32_t cnt[2] = {0u, 0u}; \
    rgi::Lane P; \
    rgi::InitLane(F, g, x, y, stack, cnt, P); \
    for (;;) \
    { \
        const bool any = __ballot(P.active) != 0; \
        rgi::PhaseA(F, g, stack, cnt, P); \
        if (!any) break; \
        uint32_t key = rgi::RRKey(P); \
        if (__ballot(key != 0) != 0) \
        { \
            for (int s = 1; s < 64; s <<= 1) { uint32_t o = __shfl_xor(key, s); key = o > key ? o : key; } \
        } \
        rgi::PhaseB(F, g, stack, cnt, P, key); \
    } \
    rgi::ReloadPrimary(F, g, P); \
    const float w = rgi::FinishAndResample(F, g, stack, cnt, P); \
    const float waveSum = WaveSumButterfly(w); \
    rgi::SuppressAndWrite(F, P, waveSum); \
    FlushRayCounters(counters, cnt);
template<bool PLAIN>
__global__ void __launch_bounds__(kRgiBlock) ZR_WAVES_RGI k_rgi(rgi::GiFrame F, zr_frame_constants g, uint32_t tilesX, unsigned long long* counters)
{ ZR_RGI_KERNEL_BODY(false, PLAIN) }
// the TEXTURED permutation hides its texel-gather latency with more waves, like K11's (textured atrium: 12.12 ms at 4 waves, 11.42 at 5, 10.96 at 6;
// the untextured kernel is best at 4; scripts/gpu_waves2.sh [rounds 1-4: git history up to 31e92fa; today scripts/gpu.sh ab])
__global__ void k_rgi_tex(rgi::GiFrame F, zr_frame_constants g, uint32_t tilesX, unsigned long long* counters);

// X = `template` in the TU that owns the group, `extern template` everywhere else (like ZR_RPT_GROUP_*)
#define ZR_DI_ARGS(Frame) (Frame, zr_frame_constants, uint32_t, unsigned long long*)
#define ZR_DI_GROUP(X, PLAIN) \
    X __global__ void k_sdi_temporal<PLAIN> ZR_DI_ARGS(sdi::SkyFrame); X __global__ void k_sdi_spatial<PLAIN> ZR_DI_ARGS(sdi::SkyFrame); \
    X __global__ void k_rdi_temporal<PLAIN, false> ZR_DI_ARGS(rdi::DiFrame); X __global__ void k_rdi_spatial<PLAIN, false> ZR_DI_ARGS(rdi::DiFrame); \
    X __global__ void k_rdi_temporal<PLAIN, true> ZR_DI_ARGS(rdi::DiFrame); X __global__ void k_rdi_spatial<PLAIN, true> ZR_DI_ARGS(rdi::DiFrame); \
    X __global__ void k_rgi<PLAIN> ZR_DI_ARGS(rgi::GiFrame);
