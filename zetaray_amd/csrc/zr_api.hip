// zr_api.hip -- HIP kernels (gfx950) and the C-ABI of libzetaray_amd.so (include/zetaray_amd.h).
//
// Kernel wrappers around the stage functions of zr_stages.h: pixel <-> lane mapping, wave-ballot stream compaction
// into the SoA path queues, ray counters, per-kernel hipEvent timing.  Host side: scene upload + BVH build, G-buffer
// planes, the pass objects mirroring the reference's RenderPass surface (citations in include/zetaray_amd.h).
// There is no CPU fallback in this library: without a usable HIP device every compute entry point fails loudly.
#include <hip/hip_runtime.h>
#include <immintrin.h>
#include <string>
#include <vector>
#include <cstdio>
#include <cstdarg>
#include <cstring>
#include <new>
#include <mutex>
#include "zr_stages.h"
#include "zr_rpt.h"
#include "zr_rdi.h"
#include "zr_sdi.h"
#include "zr_rgi.h"
#include "zr_bvh.h"
#include "zr_taa.h"
#include "zr_svgf.h"
#include "zr_post.h"
#include "../../include/zr_srgb_table.h"

// 512 x half2 spatial-search points (generated from zetaray_amd/assets/rpt_sample_set_f16.bin by the Makefile)
static const uint16_t kRptSampleSet[1024] = {
#include "zr_rpt_sample_set.inc"
};


// ------------------------------------------------------------------------------------------------ errors
static thread_local std::string g_err;
static int Fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
    g_err = buf;
    return code;
}
#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return Fail(ZR_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)

static int RequireDevice(int device)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return Fail(ZR_ERR_NO_DEVICE, "no usable HIP device (hipGetDeviceCount: %s); libzetaray_amd has no CPU path", hipGetErrorString(e));
    if (device < 0 || device >= n) return Fail(ZR_ERR_INVALID_ARG, "device %d out of range (0..%d)", device, n - 1);
    HIP_TRY(hipSetDevice(device));
    return ZR_OK;
}

#include "zr_kernels.h"

// A/B switches of the measurement builds.  The product library has none: each reads as its default.  `make experiments` (-DZR_EXPERIMENTS ->
// libzetaray_amd_exp.so) turns them into environment variables read once per process.
#ifdef ZR_EXPERIMENTS
#define ZR_EXP_ENV(name) std::getenv(name)
#else
#define ZR_EXP_ENV(name) ((const char*)nullptr)
#endif
#include "zr_bvh_device.h"
namespace zr { int DeviceProbeRun(int device, float min_ms, zr_device_probe* out, std::string& err); }      // zr_tu_probe.hip
// the ReSTIR PT kernels are compiled in zr_tu_rpt_[a-i].hip (see zr_kernels.h)
ZR_RPT_GROUPS_PRODUCT(extern template)
#ifdef ZR_EXPERIMENTS
ZR_RPT_GROUP_C(extern template)
#endif

// pickXY: x | y << 16 of the pixel to pick (render-target coordinates), 0xffffffff = none
__global__ void __launch_bounds__(kBlock) k_gbuffer(SceneView sc, zr_frame_constants g, GBuf gb, uint32_t tilesX, uint32_t pickXY, uint32_t* pick)
{
    uint32_t x, y; PixelOfThread(tilesX, gb.x0, gb.y0, &x, &y);
    if (x >= gb.x0 + gb.w || y >= gb.y0 + gb.h) return;
    ZR_TRAV_STACK(stack);
    GBufferPixel(sc, g, gb, x, y, stack, nullptr, (x | (y << 16)) == pickXY ? pick : nullptr);
}

// (PLAIN: the material-class permutation of the K9 stages, like K11's -- zr_rpt.h SetMaterialClass)
template<bool TEX, bool PLAIN>
__global__ void __launch_bounds__(kBlock) k_pt_init(SceneView sc, zr_frame_constants g, GBuf gb, PtParams prm, float* finalRGBA,
    F4* firstBOP, PathQueue out, uint32_t* outCount, uint32_t* outRays, uint32_t cap, uint32_t tilesX)
{
    sc.plain = PLAIN; gb.plain = PLAIN;
    uint32_t x, y; PixelOfThread(tilesX, gb.x0, gb.y0, &x, &y);
    PathOut po; po.alive = false;
    if (x < gb.x0 + gb.w && y < gb.y0 + gb.h) PtInitPixel(sc, g, gb, prm, x, y, finalRGBA, firstBOP, po, TEX);
    const uint32_t slot = AllocPathAndRaysBlock(outCount, out.rayList, cap, outRays, po.alive, po.alive && po.rayC_d.w >= 0, po.alive && po.rayM_d.w >= 0, po.alive && po.rayS_d.w >= 0);
    if (po.alive) WritePath(out, slot, po, TEX);
}

// trace stage, run-to-completion variant (default): grid-stride over the queue's compacted ray list
__global__ void __launch_bounds__(kBlock) k_trace_simple(SceneView sc, PathQueue q, const uint32_t* rayCount, uint32_t cap, unsigned long long* counters)
{
    const uint32_t nC = rayCount[0], nM = rayCount[kCounterStride], total = nC + nM + rayCount[2 * kCounterStride];
    ZR_TRAV_STACK(stack);
#ifdef ZR_NODE_CACHE_MORE
    ZR_NODE_CACHE_FILL(stack, sc, kBlock);
#endif
    for (uint32_t base = blockIdx.x * kBlock; base < total; base += gridDim.x * kBlock)
    {
        const uint32_t j = base + threadIdx.x;
        bool closest = false, shadow = false;
        if (j < total)
        {
            // one Traverse call site for the three ray types: a wave holds a mix of them
            uint32_t type, i;
            RayOfIndex(q.rayList, cap, nC, nM, j, type, i);
            F4 ro, rd;
            if (type == 0) { rd = q.rayC_d[i]; ro = q.rayC_o[i]; }
            else if (type == 1) { rd = q.rayM_d[i]; ro = q.rayM_o[i]; }
            else { rd = q.rayS_d[i]; ro = q.rayS_o[i]; }
            // S rays: segment to an emissive triangle (closest over NON_EMISSIVE) or sun / sky visibility (any hit over ALL)
            const uint32_t lightID = type == 2 ? q.sLightID[i] : 0u;
            const bool vis = type == 2 && lightID == kVisibilityRayID;
            const RawHit h = TraverseDyn(sc, xyz(ro), xyz(rd), ro.w, rd.w, (type == 2 && !vis) ? ZR_SUBGROUP_NON_EMISSIVE : ZR_SUBGROUP_ALL, stack, vis);
            if (type == 0) q.hitC[i] = PackRawHit(h);
            else if (type == 1) q.hitM[i] = PackRawHit(h);
            else q.visS[i] = SegmentVisible(sc, h, lightID);
            closest = type != 2; shadow = type == 2;
        }
        CountWave(&counters[0], closest);
        CountWave(&counters[1], shadow);
    }
}

// trace stage, persistent variant (ZR_TRACE_MODE=1).  Every lane owns one ray at a time; each iteration the wave votes
// for one phase -- refill idle lanes from the wave's chunk of ray slots, inner-node step, or one triangle test -- and runs
// the one most lanes are waiting for (the phases of Traverse, zr_dev_scene.h, plus the refill).  Waves take chunks of
// kTraceChunk entries of the compacted ray list from a global cursor.  Which lane traces which ray, and in which order,
// has no effect on the results.
static constexpr uint32_t kTraceChunk = 256;
static constexpr uint32_t kTraceRefillAt = 16;
__global__ void __launch_bounds__(kBlock) k_trace(SceneView sc, PathQueue q, const uint32_t* rayCount, uint32_t cap, uint32_t* cursor, unsigned long long* counters)
{
    const uint32_t nC = rayCount[0], nM = rayCount[kCounterStride], total = nC + nM + rayCount[2 * kCounterStride];
    ZR_TRAV_STACK(stack);
    const uint32_t lane = __lane_id();
    TravState st; TravLane L; L.triCur = 0; L.triEnd = 0; L.done = true;
    uint32_t slot = 0;                              // this lane's ray: list entry (slot | type << 30)
    bool laneAnyHit = false;
    uint32_t nClosest = 0, nShadow = 0;
    uint32_t chunkPos = 0, chunkEnd = 0;            // wave-uniform
    bool exhausted = false;                         // wave-uniform: the global cursor ran past `total`
    for (;;)
    {
        const bool atTri = L.triCur < L.triEnd;
        const bool atNode = !L.done && !atTri;
        const uint64_t mIdle = __ballot(L.done);
        const uint32_t nNode = (uint32_t)__popcll(__ballot(atNode)), nTri = (uint32_t)__popcll(__ballot(atTri)), nIdle = (uint32_t)__popcll(mIdle);
        const bool canRefill = !(exhausted && chunkPos == chunkEnd);
        if (nNode + nTri == 0 && !canRefill) break;
        if (canRefill && (nIdle >= kTraceRefillAt || nNode + nTri == 0))
        {
            if (chunkPos == chunkEnd)
            {
                uint32_t b = 0;
                if (lane == 0) b = atomicAdd(cursor, kTraceChunk);
                b = __builtin_amdgcn_readfirstlane(b);
                exhausted = b >= total;
                chunkPos = b < total ? b : total;
                chunkEnd = b + kTraceChunk < total ? b + kTraceChunk : total;
            }
            const uint32_t rank = (uint32_t)__popcll(mIdle & ((1ull << lane) - 1ull));
            const uint32_t j = chunkPos + rank;
            if (L.done && j < chunkEnd)
            {
                uint32_t type, i;
                RayOfIndex(q.rayList, cap, nC, nM, j, type, i);
                slot = i | (type << 30);
                F4 ro, rd;
                if (type == 0) { rd = q.rayC_d[i]; ro = q.rayC_o[i]; }
                else if (type == 1) { rd = q.rayM_d[i]; ro = q.rayM_o[i]; }
                else { rd = q.rayS_d[i]; ro = q.rayS_o[i]; }
                L.done = false;
                if (type == 2)
                {
                    laneAnyHit = q.sLightID[i] == kVisibilityRayID;
                    TravInit(sc, st, xyz(ro), xyz(rd), ro.w, rd.w, laneAnyHit ? ZR_SUBGROUP_ALL : ZR_SUBGROUP_NON_EMISSIVE, false, 0); nShadow++;
                }
                else { laneAnyHit = false; TravInit(sc, st, xyz(ro), xyz(rd), ro.w, rd.w, ZR_SUBGROUP_ALL, false, 0); nClosest++; }
                TravEnter(sc, st, L, st.cur);
            }
            chunkPos = chunkPos + nIdle < chunkEnd ? chunkPos + nIdle : chunkEnd;
            continue;
        }
        bool stepped = false;
        if (nNode >= nTri) { if (atNode) { TravNodePhase(sc, st, L, stack); stepped = true; } }
        else if (atTri) { TravTriPhase(sc, st, L, stack, laneAnyHit); stepped = true; }
        if (stepped && L.done)
        {
            // shadow segments run to the closest hit: visible iff it is the light's own triangle (TraceSegmentRay)
            const uint32_t type = slot >> 30, i = slot & 0x3fffffffu;
            if (type == 0) q.hitC[i] = PackRawHit(st.best);
            else if (type == 1) q.hitM[i] = PackRawHit(st.best);
            else q.visS[i] = SegmentVisible(sc, st.best, q.sLightID[i]);
        }
    }
    uint32_t a = nClosest, b = nShadow;
    for (int s = 1; s < 64; s <<= 1) { a += __shfl_xor(a, s); b += __shfl_xor(b, s); }
    if (lane == 0) { if (a) atomicAdd(&counters[0], (unsigned long long)a); if (b) atomicAdd(&counters[1], (unsigned long long)b); }
}

template<bool TEX>
__device__ __forceinline__ void PtShadeBody(const SceneView& sc, const zr_frame_constants& g, const PtParams& prm, const PathQueue& in, const uint32_t* inCount,
    const PathQueue& out, uint32_t* outCount, uint32_t* outRays, uint32_t cap, float* finalRGBA, const F4* firstBOP, uint32_t* groupMax)
{
    const uint32_t n = *inCount;
    for (uint32_t base = blockIdx.x * kBlock; base < n; base += gridDim.x * kBlock)
    {
        const uint32_t i = base + threadIdx.x;
        PathOut po; po.alive = false;
        if (i < n) PtShadePath(sc, g, prm, in, i, finalRGBA, firstBOP, groupMax, po, TEX);
        const uint32_t slot = AllocPathAndRaysBlock(outCount, out.rayList, cap, outRays, po.alive, po.alive && po.rayC_d.w >= 0, po.alive && po.rayM_d.w >= 0, po.alive && po.rayS_d.w >= 0);
        if (po.alive) WritePath(out, slot, po, TEX);
    }
}
template<bool PLAIN>
__global__ void __launch_bounds__(kBlock) k_pt_shade(SceneView sc, zr_frame_constants g, PtParams prm, PathQueue in, const uint32_t* inCount,
    PathQueue out, uint32_t* outCount, uint32_t* outRays, uint32_t cap, float* finalRGBA, const F4* firstBOP, uint32_t* groupMax)
{ sc.plain = PLAIN; PtShadeBody<false>(sc, g, prm, in, inCount, out, outCount, outRays, cap, finalRGBA, firstBOP, groupMax); }
// textured permutation: 258 VGPRs by default = 1 wave per SIMD; asking for 2 costs 2 registers and takes the atrium's shade
// stage from 8.6 to 5.6 ms per frame (the same request on the untextured kernel, 246 VGPRs, made it 10 % slower -- not applied)
#ifndef ZR_WAVES_PT_SHADE_TEX
#define ZR_WAVES_PT_SHADE_TEX ZR_WAVES_MIN(2)
#endif
__global__ void __launch_bounds__(kBlock) ZR_WAVES_PT_SHADE_TEX k_pt_shade_tex(SceneView sc, zr_frame_constants g, PtParams prm, PathQueue in, const uint32_t* inCount,
    PathQueue out, uint32_t* outCount, uint32_t* outRays, uint32_t cap, float* finalRGBA, const F4* firstBOP, uint32_t* groupMax)
{ sc.plain = 0; PtShadeBody<true>(sc, g, prm, in, inCount, out, outCount, outRays, cap, finalRGBA, firstBOP, groupMax); }

// Russian-roulette stage: finishes the vertices PtShadePath parked (only launched for rounds in which RR can trigger)
template<bool TEX>
__global__ void __launch_bounds__(kBlock) k_pt_rr(SceneView sc, PtParams prm, PathQueue q, const uint32_t* count, uint32_t* rays, uint32_t cap, const uint32_t* groupMax)
{
    sc.plain = 0;
    const uint32_t n = *count;
    for (uint32_t base = blockIdx.x * kBlock; base < n; base += gridDim.x * kBlock)
    {
        const uint32_t i = base + threadIdx.x;
        const bool cont = i < n && PtRussianRoulette(sc, prm, q, i, groupMax, TEX);
        AllocPathAndRaysBlock(nullptr, q.rayList, cap, rays, false, cont, false, false, true, i);
    }
}

// TAA: one thread per pixel; 9 signal + 9 depth reads from a 3 x 3 neighbourhood that the L2 serves after the first touch, 9 bilinear
// history fetches (36 half4 texels) around the reprojected position, one half4 store: HBM-bound (16 + 4 + 4 + 8 B read, 8 B written per
// pixel algorithmically)
#ifndef ZR_WAVES_TAA
#define ZR_WAVES_TAA
#endif
__global__ void __launch_bounds__(256) ZR_WAVES_TAA k_taa(taa::TaaFrame F)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < F.w * F.h) taa::TaaPixel(F, i % F.w, i / F.w);
}

// Denoise pass (zr_svgf.h): blocks of 32 x 8 pixels of the pass's planes (a window of the frame: the whole of it on one device, a tile + apron in
// the multi-GPU split), one thread per pixel.  The a-trous kernel is VALU-bound (r04: ~1.6 k VALU instructions per pixel in definition 1, SIMD VALU
// ~0.9 busy at 0.03 of the HBM roof): definition 2 of zr_svgf.h is what addresses that; the LDS tiles of the dense steps save the remaining tap latency.
__device__ __forceinline__ bool SvgfPixel(const svgf::Window& w, int* x, int* y)
{
    const int lx = (int)(blockIdx.x * 32u + (threadIdx.x & 31u)), ly = (int)(blockIdx.y * 8u + (threadIdx.x >> 5));
    *x = w.ox + lx; *y = w.oy + ly;
    return lx < w.pw && ly < w.ph;
}
__global__ void __launch_bounds__(256) k_svgf_temporal(svgf::SvgfFrame F) { int x, y; if (SvgfPixel(F.win, &x, &y)) svgf::TemporalPixel(F, x, y); }
// POW: svgf_normal_power_log2 as a compile-time constant (7, the default: the squarings unroll) or -1 (read from the parameters)
template<int POW> __global__ void __launch_bounds__(256) k_svgf_variance(svgf::FilterFrame F) { int x, y; if (SvgfPixel(F.win, &x, &y)) svgf::VariancePixelT<POW>(F, x, y); }
// ROWLOOP: zr_svgf.h AtrousPixelT -- false = the 24 taps unrolled (held to 128 VGPRs: 4 waves per SIMD), true = a loop over the tap rows (8 waves)
template<int POW, bool ROWLOOP> __global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) k_svgf_atrous(svgf::FilterFrame F)
{ int x, y; if (SvgfPixel(F.win, &x, &y)) { svgf::PlaneTaps t; t.srcP = (const U4*)F.src; t.guideZ = F.guideZ; t.win = F.win; svgf::AtrousPixelT<POW, ROWLOOP>(F, x, y, t); } }
// The same iteration with the block's (32 + 4 S) x (8 + 4 S) neighbourhood of both planes staged in LDS first (steps 1 and 2: 13.8 / 20.5 KB per block),
// so that the 25 taps + the 3 x 3 variance blur are ds_read_b128 instead of cache hits.  Same stage function, same results.  The default for
// steps 1 and 2 (RenderDenoise).  Tap positions arrive clamped into the planes, which keeps them inside the tile (a clamped position is nearer to the block
// than the unclamped one).
struct LdsTaps
{
    const ZR_LDS_AS F4* c; const ZR_LDS_AS F4* g; int x0, y0, tw;      // the tile holds the stage texels unpacked: (rgb, variance) and (n, z) in fp32
    __device__ __forceinline__ int RowBase(int y) const { return (y - y0) * tw - x0; }
    __device__ __forceinline__ void Load(int rb, int x, F4& gq, F4& q) const
    {
        const ZR_LDS_AS F4* a = c + rb + x; const ZR_LDS_AS F4* b = g + rb + x;
        q.x = a->x; q.y = a->y; q.z = a->z; q.w = a->w; gq.x = b->x; gq.y = b->y; gq.z = b->z; gq.w = b->w;
    }
    __device__ __forceinline__ float Var(int rb, int x) const { return (c + rb + x)->w; }
};
template<int S, int POW, bool ROWLOOP>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) k_svgf_atrous_lds(svgf::FilterFrame F)
{
    constexpr int TW = 32 + 4 * S, TH = 8 + 4 * S;
    __shared__ F4 sC[TW * TH];
    __shared__ F4 sG[TW * TH];
    const svgf::Window& w = F.win;
    const int bx0 = w.ox + (int)(blockIdx.x * 32u) - 2 * S, by0 = w.oy + (int)(blockIdx.y * 8u) - 2 * S;
    for (int t = (int)threadIdx.x; t < TW * TH; t += 256)
    {
        const int gx = bx0 + t % TW, gy = by0 + t / TW;
        if (w.InPlanes(gx, gy)) { const size_t j = w.Idx(gx, gy); F4 gq, q; svgf::UnpackStage(((const U4*)F.src)[j], F.guideZ[j], gq, q); sC[t] = q; sG[t] = gq; }
    }
    __syncthreads();
    int x, y;
    if (!SvgfPixel(w, &x, &y)) return;
    LdsTaps taps; taps.c = (const ZR_LDS_AS F4*)sC; taps.g = (const ZR_LDS_AS F4*)sG; taps.x0 = bx0; taps.y0 = by0; taps.tw = TW;
    svgf::AtrousPixelT<POW, ROWLOOP>(F, x, y, taps);
}

// AutoExposure_Histogram.hlsl: per-block LDS histogram (256 bins = 256 threads), one global atomic per non-empty bin and block.
// in16 / in32: exactly one is non-null (RGBA16F plane, or RGBA32F read rounded to half).  HBM-bound: 8 (16) B read per pixel.
__global__ void __launch_bounds__(256) k_ae_histogram(const uint16_t* in16, const F4* in32, uint32_t n, post::AeParams prm, uint32_t* hist)
{
    __shared__ uint32_t bins[post::kHistBins];
    bins[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    {
        const V3 c = in16 ? post::LoadHalf3(in16, i) : post::HalfRounded(in32[i]);
        atomicAdd(&bins[post::AeBin(c, prm)], 1u);
    }
    __syncthreads();
    if (bins[threadIdx.x]) atomicAdd(hist + threadIdx.x, bins[threadIdx.x]);
}
// AutoExposure_WeightedAvg.hlsl: one 256-thread group = 4 waves of 64; WaveActiveSum = the ABI's xor butterfly
__global__ void __launch_bounds__(256) k_ae_resolve(const uint32_t* hist, uint32_t numPixels, float dt, post::AeParams prm, float* exposure)
{
    const uint32_t gidx = threadIdx.x;
    float val = post::AeBinValue(gidx, gidx == 0 ? 0u : hist[gidx]);
    val = WaveSumButterfly(val);
    __shared__ float waveSum[4];
    if ((gidx & 63u) == 0) waveSum[gidx >> 6] = val;
    __syncthreads();
    float mean = gidx < 4 ? waveSum[gidx] : 0.0f;
    mean = WaveSumButterfly(mean);
    if (gidx == 0) post::AeResolve(mean, numPixels - hist[0], dt, prm, exposure);
}
// Display.hlsl mainPS: one thread per display pixel; point-clamp fetch of the render-resolution image
__global__ void __launch_bounds__(256) k_display(const uint16_t* in16, const F4* in32, uint32_t rw, uint32_t rh, uint32_t dw, uint32_t dh,
    const float* exposure, post::DisplayParams prm, post::Lut3D lut, F4* out, uint32_t* outSrgb)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= dw * dh) return;
    const uint32_t x = i % dw, y = i / dw;
    const float u = ((float)x + 0.5f) / (float)dw, v = ((float)y + 0.5f) / (float)dh;
    int sx = (int)zr_floor(u * (float)rw), sy = (int)zr_floor(v * (float)rh);
    sx = sx < 0 ? 0 : (sx > (int)rw - 1 ? (int)rw - 1 : sx); sy = sy < 0 ? 0 : (sy > (int)rh - 1 ? (int)rh - 1 : sy);
    const size_t sp = (size_t)sy * rw + sx;
    // Texture2D<float4> on the composited texture: the stored (half) values
    const V3 c = in16 ? post::LoadHalf3(in16, sp) : post::HalfRounded(in32[sp]);
    const V3 d = post::DisplayPixel(c, (prm.autoExposure && exposure) ? exposure[0] : 1.0f, prm, lut);
    out[i] = f4(d, 1.0f);
    outSrgb[i] = post::LinearToSrgb8(d.x) | (post::LinearToSrgb8(d.y) << 8) | (post::LinearToSrgb8(d.z) << 16) | 0xff000000u;
}

// self-test of zr_detmath.h's half conversions (instruction path vs portable path), see zr_selftest_half_conversions
__global__ void __launch_bounds__(256) k_selftest_half(unsigned long long* bad)
{
    unsigned long long b0 = 0, b1 = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride)
    {
        const float f = zr_asfloat((uint32_t)i);
        if (zr_f32_to_f16(f) != zr_f32_to_f16_portable(f)) b0++;
        if (i < 65536u)
        {
            if (zr_asuint(zr_f16_to_f32((uint16_t)i)) != zr_asuint(zr_f16_to_f32_portable((uint16_t)i))) b1++;
            // the UNORM divisions by constant (zr_div255 / zr_div65535) against the IEEE division
            const float x = (float)(uint32_t)i;
            volatile float d255 = 255.0f, d65535 = 65535.0f;      // (volatile: keep real divisions)
            if (zr_asuint(zr_div255(x)) != zr_asuint(x / d255)) b1++;
            if (zr_asuint(zr_div65535(x)) != zr_asuint(x / d65535)) b1++;
        }
    }
    if (b0) atomicAdd(bad, b0);
    if (b1) atomicAdd(bad + 1, b1);
}

// Compositing: pure streaming kernel (2 + 16 + 16 B read, 16 B written per pixel)
__global__ void __launch_bounds__(256) k_composite(zr_frame_constants g, const uint16_t* mr, const F4* skyDI, const F4* emissiveDI, const F4* indirect, F4* out, uint32_t n,
    SkyLutView sky, uint32_t w)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = CompositePixel(g, mr[i], skyDI, emissiveDI, indirect, i, out[i], sky, i % w, i / w);
}

// Firefly filter (FireflyFilter.hlsl): a 3 x 3 stencil over RGBA32F + depth.  A block filters a 64 x 16 pixel tile, 4 pixels per thread;
// colour and depth of the tile and its 1-pixel border are staged through LDS once (66 x 18 texels x 20 B = 23.8 KB, rows of 1 KB so the
// global loads coalesce and ~5 of them are in flight per thread), so every texel leaves HBM / L2 once per block instead of 9 times per
// pixel: 20 B read + 16 B written per pixel, HBM-bound.
static constexpr int kFfW = 64, kFfH = 16, kFfLW = kFfW + 2, kFfLH = kFfH + 2;
__global__ void __launch_bounds__(256) k_firefly(const F4* in, const float* depth, F4* out, uint32_t W, uint32_t H)
{
    __shared__ F4 sCol[kFfLW * kFfLH];
    __shared__ float sDep[kFfLW * kFfLH];
    const int bx = (int)blockIdx.x * kFfW, by = (int)blockIdx.y * kFfH;
    for (int t = threadIdx.x; t < kFfLW * kFfLH; t += 256)
    {
        const int lx = t % kFfLW, ly = t / kFfLW;
        const int gx = bx + lx - 1, gy = by + ly - 1;
        const bool inside = gx >= 0 && gy >= 0 && gx < (int)W && gy < (int)H;
        sCol[t] = inside ? in[(size_t)gy * W + gx] : f4(v3(0.0f), 0.0f);
        sDep[t] = inside ? depth[(size_t)gy * W + gx] : ZR_FLT_MAX;
    }
    __syncthreads();
    auto col = [&](int tx, int ty) { return xyz(sCol[ty * kFfLW + tx]); };
    auto dep = [&](int tx, int ty) { return sDep[ty * kFfLW + tx]; };
    for (int k = 0; k < 4; k++)
    {
        const int lx = threadIdx.x & 63, ly = (threadIdx.x >> 6) + 4 * k;
        const int x = bx + lx, y = by + ly;
        if (x >= (int)W || y >= (int)H) continue;
        const F4 c = sCol[(ly + 1) * kFfLW + lx + 1];
        V3 color = xyz(c);
        if (sDep[(ly + 1) * kFfLW + lx + 1] != ZR_FLT_MAX) color = FireflyClamp(col, dep, lx + 1, ly + 1, x, y, (int)W, (int)H, color);
        out[(size_t)y * W + x] = f4(color, c.w);
    }
}

__global__ void k_presample(SceneView sc, uint32_t total, uint32_t frameNum, uint32_t numEmissives, zr_presampled_tri* out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) out[i] = PresampleEmissive(sc, i, frameNum, numEmissives);
}

__global__ void k_estimate_power(SceneView sc, uint32_t n, float* power)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) power[i] = EstimateTriPower(sc, sc.emissives[i]);
}

__device__ __forceinline__ float WaveSumButterfly(float v);
// K4 light voxel grid: one wave64 per voxel = the reference's 64-thread group; the group sums are wave reductions
__global__ void __launch_bounds__(64) k_build_lvg(SceneView sc, zr_frame_constants g, uint32_t dx, uint32_t dy, uint32_t dz, float ex, float ey, float ez,
    float offset_y, zr_voxel_sample* out)
{
    const uint32_t dim[3] = {dx, dy, dz};
    const int v[3] = {(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z};
    zr_voxel_sample r; float w_sum, target_z; uint32_t numLights;
    LvgThread(sc, g, dim, v3(ex, ey, ez), offset_y, v, threadIdx.x, r, w_sum, target_z, numLights);
    const float w_sum_group = WaveSumButterfly(w_sum);
    uint32_t n = numLights;
    for (int s = 1; s < 64; s <<= 1) n += __shfl_xor(n, s);
    LvgFinish(r, target_z, w_sum_group, n);
    out[(size_t)LvgFlatten(v, dim) * ZR_LVG_SAMPLES_PER_VOXEL + threadIdx.x] = r;
}

// K17 sky-view LUT: one thread per texel, 8 x 8 threads per group like the reference (ALU-bound: ~300 exp per texel)
__global__ void __launch_bounds__(64) k_sky_lut(zr_frame_constants g, uint32_t w, uint32_t h, uint32_t* out)
{
    const uint32_t x = blockIdx.x * 8 + (threadIdx.x & 7), y = blockIdx.y * 8 + (threadIdx.x >> 3);
    if (x < w && y < h) out[(size_t)y * w + x] = SkyViewLutTexel(g, x, y, w, h);
}

// generic ray-query kernels behind zr_trace_closest / zr_trace_any
__global__ void __launch_bounds__(kBlock) k_trace_rays(SceneView sc, const F4* rays, uint32_t n, uint32_t mask, U4* hits)
{
    ZR_TRAV_STACK(stack);
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock)
        hits[i] = TraceClosestRay(sc, rays[2 * i], rays[2 * i + 1], mask, stack);
}
__global__ void __launch_bounds__(kBlock) k_trace_rays_any(SceneView sc, const F4* rays, uint32_t n, uint32_t mask, uint32_t* occ)
{
    ZR_TRAV_STACK(stack);
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock)
    {
        const F4 ro = rays[2 * i], rd = rays[2 * i + 1];
        RawHit h = Traverse<true>(sc, xyz(ro), xyz(rd), ro.w, rd.w, mask, stack);
        occ[i] = h.tri != kInvalidTri ? 1u : 0u;
    }
}

// ------------------------------------------------------------------------------------------------ direct-lighting and ReSTIR GI kernels
// K5 - K8 and K10 are defined in zr_tu_di.hip (a translation unit of their own: zr_api.hip was the build's long pole); launched from here through their host stubs.
// threads per block (see kRptBlock in zr_kernels.h): one-wave blocks pay for the emissive DI kernels (K5 0.580 -> 0.568 ms
// Cornell, 5.42 -> 5.12 ms atrium; K6 0.258 -> 0.241 / 2.99 -> 2.61 ms), not for the sun + sky ones (K7 / K8 within +-0.6 %)
#include "zr_kernels_di.h"
ZR_DI_GROUP(extern template, false)
ZR_DI_GROUP(extern template, true)
static const uint16_t kRdiSampleSet[64] = {
#include "zr_rdi_sample_set.inc"
};

// ------------------------------------------------------------------------------------------------ host objects
template<typename T> struct DevBuf
{
    T* p = nullptr; size_t n = 0;
    int Alloc(size_t count) { Free(); n = count; if (!count) return ZR_OK; hipError_t e = hipMalloc((void**)&p, count * sizeof(T)); if (e != hipSuccess) { p = nullptr; return Fail(ZR_ERR_OOM, "hipMalloc(%zu B) failed: %s", count * sizeof(T), hipGetErrorString(e)); } return ZR_OK; }
    int Upload(const T* src, size_t count) { int r = Alloc(count); if (r) return r; if (count) HIP_TRY(hipMemcpy(p, src, count * sizeof(T), hipMemcpyHostToDevice)); return ZR_OK; }
    void Free() { if (p) { (void)hipFree(p); p = nullptr; } n = 0; }
    ~DevBuf() { Free(); }
};

struct zr_scene
{
    int device = 0;
    DevBuf<zr_vertex> vertices; DevBuf<uint32_t> indices; DevBuf<zr_mesh_instance> instances; DevBuf<zr_material> materials;
    DevBuf<zr_emissive_triangle> emissives; DevBuf<zr_alias_entry> alias; DevBuf<Bvh4Node> nodes; DevBuf<BvhTri> tris;
    // material class (SceneView::plain): host copy of the material table + "every material is an opaque uncoated non-metallic dielectric"; kept current by
    // zr_scene_create / zr_scene_update_materials on the calling thread, read by zr_pass_render when it picks a kernel permutation
    std::vector<zr_material> hMaterials; std::atomic<bool> plainMaterials{false};
    // identity of this scene for the G-buffers rendered from it (zr_gbuffer::sceneAt): never reused, unlike an address
    uint64_t uid = [] { static std::atomic<uint64_t> next{1}; return next.fetch_add(1); }();
    DevBuf<TriMeta> meta; DevBuf<uint16_t> rho;
    DevBuf<zr_texture_desc> texDescs; DevBuf<uint8_t> texels; DevBuf<float> srgb;   // material texture heap (zr_texture.h)
    DevBuf<zr_presampled_tri> sampleSets; uint32_t numSampleSets = 0;      // K3 output, refreshed by every PRELIGHTING render
    DevBuf<zr_voxel_sample> lvg;                                           // K4 output, rebuilt by every PRELIGHTING render when use_lvg
    std::vector<zr_alias_entry> aliasHost;
    // dynamic instances (zr_scene_update_instances): host copies of what a BVH rebuild needs, and the previous frame's instance buffer + BVH
    std::vector<zr_vertex> hVertices; std::vector<uint32_t> hIndices; std::vector<uint8_t> hMask; std::vector<uint32_t> hNumTris;
    DevBuf<zr_mesh_instance> instancesPrev; DevBuf<Bvh4Node> nodesPrev; DevBuf<BvhTri> trisPrev; DevBuf<TriMeta> metaPrev;
    uint32_t numNodesPrev = 0, numTrisPrev = 0; bool hasPrev = false;
    // device refit: node indices grouped by tree level (deepest level first) + the offsets of the groups, per-node float bounds, object-to-world matrices
    DevBuf<uint32_t> levelNodes; std::vector<uint32_t> levelOffsets, hLevelOrder; DevBuf<float> nodeBounds, toWorld; bool refitReady = false;
    // device-side BVH build (zr_tu_bvh.hip): instance masks on the device, scratch buffers kept between builds
    DevBuf<uint8_t> dMask; DeviceBvhScratch bvhScratch; bool deviceBuilt = false;
    bool aliasStale = false;      // zr_scene_invalidate_alias_table_deferred: the old table stays bound until the rebuilt one has been uploaded
    // `view` is what kernels receive by value.  Passes of one dependency level may record concurrently from several host threads
    // (RenderGraph.cpp:442-541) while Sky / PreLighting publish scene-owned state (sky LUT, alias table, presampled sets, LVG):
    // every reader takes a private copy through FrameView() and every writer updates `view` under `mtx`.
    SceneView view{};
    mutable std::mutex mtx;
    int64_t maxTex[4] = {-1, -1, -1, -1};     // largest tex16 used per table (base colour, normal, metallic-roughness, emissive)
    uint32_t maxDepth = 0;
    // ---- stream-ordered updates (zr_scene_update_*_async; RtAccelerationStructure.cpp:708-789 records them on the frame's command list).
    // Host records travel through a ring of pinned staging buffers (a slot is reused only after its copy has run); `updated` is recorded
    // behind every update and waited for by renders on OTHER streams; `users` holds, per stream that rendered with this scene, an event
    // recorded behind its last render, which an update on another stream waits for before it overwrites what those renders read.
    struct StageSlot { void* host = nullptr; size_t cap = 0; hipEvent_t ev = nullptr; bool pending = false; };
    StageSlot stage[4]; int stageNext = 0;
    hipEvent_t updated = nullptr; hipStream_t updatedOn = nullptr; bool hasUpdate = false;
    std::vector<std::pair<hipStream_t, hipEvent_t>> users;
    // ---- background SAH rebuild (zr_scene_set_background_rebuild; VERDICT r3 item 9, RtAccelerationStructure.cpp:708-789 rebuilds its TLAS every frame).
    // The device refit keeps the topology of the last build, and a tree built for where the instances WERE gets worse the further they move.  With this
    // on, an update that finds no build in flight snapshots the new transforms and starts the host's binned-SAH builder on a thread; the first update
    // after it has finished uploads the new topology into the buffer set that is about to become current and refits THAT to the transforms of the
    // update at hand.  No render waits, no result depends on the tree (include/zr_intersect.h's tie-break), the previous structure stays what it was.
    // What the build thread hands over is the TOPOLOGY only, packed in pinned memory by the thread itself: [4 child references per node | the scene
    // triangle each leaf slot holds | node ids level by level] -- 3.6 MB for the 380 k-triangle atrium, where whole node + triangle arrays are 24.8 MB.
    // Everything else in a node (origin, scales, quantised planes) and in a triangle slot (vertices, mask, ID) is recomputed on the device for the
    // transforms of the installing update (k_install_topology, then the refit kernels), so the installing update costs the host three async copies.
    struct Background
    {
        bool enabled = false; std::thread th; std::atomic<int> state{0};      // 0 idle, 1 building, 2 built
        std::vector<uint32_t> levelOffsets;
        uint32_t numNodes = 0, numTris = 0, stackNeed = 0, maxDepth = 0; bool packed = false;
        uint32_t* pkg = nullptr; size_t pkgCap = 0; hipEvent_t pkgCopied = nullptr; bool pkgInFlight = false;      // pkgCap in words
        std::vector<zr_mesh_instance> inst; std::vector<float> xf; std::vector<uint8_t> own;
        uint32_t refitsSince = 0; std::atomic<uint64_t> started{0}, installed{0};      // (atomic: zr_scene_background_rebuild_stats may read them from another thread)
        bool movedSinceBuild = false;      // a transform changed since the last build was started: updates that repeat the same matrices start no build
        double buildMs = 0, packMs = 0;
    } bg;
    DevBuf<uint32_t> bgDev;                      // device side of the package (children + slot triangles)
    // instances whose transform has ever changed: the background build gives each a subtree of its own (zr_bvh.h Build, ownSubtree)
    std::vector<float> hToWorld; std::vector<uint8_t> movedEver;
    ~zr_scene()
    {
        if (bg.th.joinable()) bg.th.join();
        if (bg.pkg) (void)hipHostFree(bg.pkg);
        if (bg.pkgCopied) (void)hipEventDestroy(bg.pkgCopied);
        for (StageSlot& t : stage) { if (t.ev) (void)hipEventDestroy(t.ev); if (t.host) (void)hipHostFree(t.host); }
        if (updated) (void)hipEventDestroy(updated);
        for (auto& u : users) (void)hipEventDestroy(u.second);
    }
};

// a pinned staging buffer of >= bytes whose previous transfer (if any) has completed; the caller fills it, enqueues its copy on `st`
// and calls StageCommit
static int StageAcquire(zr_scene* s, size_t bytes, zr_scene::StageSlot** out)
{
    zr_scene::StageSlot& t = s->stage[s->stageNext];
    s->stageNext = (s->stageNext + 1) % 4;
    if (t.pending) { HIP_TRY(hipEventSynchronize(t.ev)); t.pending = false; }      // only blocks when the host runs > 4 updates ahead of the device
    if (!t.ev) HIP_TRY(hipEventCreateWithFlags(&t.ev, hipEventDisableTiming));
    if (t.cap < bytes)
    {
        if (t.host) { (void)hipHostFree(t.host); t.host = nullptr; t.cap = 0; }
        hipError_t e = hipHostMalloc(&t.host, bytes, hipHostMallocDefault);
        if (e != hipSuccess) { t.host = nullptr; return Fail(ZR_ERR_OOM, "hipHostMalloc(%zu B) failed: %s", bytes, hipGetErrorString(e)); }
        t.cap = bytes;
    }
    *out = &t;
    return ZR_OK;
}
static int StageCommit(zr_scene::StageSlot* t, hipStream_t st) { HIP_TRY(hipEventRecord(t->ev, st)); t->pending = true; return ZR_OK; }
// an update on `st` must not overwrite what renders enqueued on other streams still read
static int SceneWaitUsers(zr_scene* s, hipStream_t st)
{
    std::lock_guard<std::mutex> lock(s->mtx);
    // the FIRST update of a scene: renders have not recorded user events yet (SceneReleaseAfterRender starts doing so once hasUpdate is set, so
    // that static scenes pay nothing), and kernels of earlier frames on non-blocking streams may still read the buffers this update overwrites
    // in place -- wait for the device once.  (Not legal inside a stream capture: capture a scene's update path only after its first update.)
    if (!s->hasUpdate) HIP_TRY(hipDeviceSynchronize());
    for (auto& u : s->users) if (u.first != st) HIP_TRY(hipStreamWaitEvent(st, u.second, 0));
    // ... and updates on different streams are ordered against each other: this one runs behind the previous one (a render on `st` then only has
    // to wait for the latest update, which is what SceneAcquireForRender does)
    if (s->hasUpdate && s->updated && s->updatedOn != st) HIP_TRY(hipStreamWaitEvent(st, s->updated, 0));
    return ZR_OK;
}
static int SceneMarkUpdated(zr_scene* s, hipStream_t st)
{
    std::lock_guard<std::mutex> lock(s->mtx);
    if (!s->updated) HIP_TRY(hipEventCreateWithFlags(&s->updated, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(s->updated, st));
    s->updatedOn = st; s->hasUpdate = true;
    return ZR_OK;
}
// a render on `st`: behind the last update if that ran on another stream ...
static int SceneAcquireForRender(const zr_scene* sc, hipStream_t st)
{
    std::lock_guard<std::mutex> lock(sc->mtx);
    if (sc->hasUpdate && sc->updatedOn != st) HIP_TRY(hipStreamWaitEvent(st, sc->updated, 0));
    return ZR_OK;
}
// ... and remembered as a user of the scene's buffers (only once the scene has ever been updated: static scenes pay nothing)
static int SceneReleaseAfterRender(const zr_scene* scc, hipStream_t st)
{
    zr_scene* sc = const_cast<zr_scene*>(scc);
    std::lock_guard<std::mutex> lock(sc->mtx);
    if (!sc->hasUpdate) return ZR_OK;
    for (auto& u : sc->users) if (u.first == st) { HIP_TRY(hipEventRecord(u.second, st)); return ZR_OK; }
    hipEvent_t ev; HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    sc->users.emplace_back(st, ev);
    HIP_TRY(hipEventRecord(ev, st));
    return ZR_OK;
}

// The scene as one launch sees it: a private copy of the scene's view with the texture descriptor-table offsets of THIS frame's
// constants (per-frame data in the reference, FrameConstants.h:31-34) -- nothing per-frame is latched on the shared scene.
#ifdef ZR_PROF
// measurement build only (scripts/gpu.sh prof): 8 kernels x 32 wave-cycle / event counters, read and cleared by zr_debug_prof_read
static unsigned long long* ProfBuffer()
{
    static unsigned long long* p = nullptr;
    if (!p) { (void)hipMalloc(&p, 256 * sizeof(unsigned long long)); (void)hipMemset(p, 0, 256 * sizeof(unsigned long long)); }
    return p;
}
extern "C" int zr_debug_prof_read(unsigned long long* out256)
{
    (void)hipDeviceSynchronize();
    if (hipMemcpy(out256, ProfBuffer(), 256 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    (void)hipMemset(ProfBuffer(), 0, 256 * sizeof(unsigned long long));
    return 0;
}
#endif
static SceneView FrameView(const zr_scene* sc, const zr_frame_constants* cb)
{
    std::lock_guard<std::mutex> lock(sc->mtx);
    SceneView v = sc->view;
#ifdef ZR_PROF
    v.prof = ProfBuffer();
#endif
    if (cb)
    {
        v.baseColorMapsOffset = cb->base_color_maps_desc_heap_offset; v.normalMapsOffset = cb->normal_maps_desc_heap_offset;
        v.mrMapsOffset = cb->metallic_roughness_maps_desc_heap_offset; v.emissiveMapsOffset = cb->emissive_maps_desc_heap_offset;
    }
    v.texFilter = ZR_TEX_FILTER_ANISOTROPIC_4X;      // the INDIRECT pass overrides it with its zr_params.tex_filter
    return v;
}
// ... and as the passes that bind the PREVIOUS acceleration structure and mesh-instance buffer see it (RT_SCENE_BVH_PREV /
// RT_FRAME_MESH_INSTANCES_PREV: CtT replay / reconnect of ReSTIR PT, the temporal shifts of the DI passes); == FrameView while nothing moved
static SceneView FrameViewPrev(const zr_scene* sc, const zr_frame_constants* cb)
{
    SceneView v = FrameView(sc, cb);
    std::lock_guard<std::mutex> lock(sc->mtx);
    if (sc->hasPrev)
    {
        v.instances = sc->instancesPrev.p; v.nodes = sc->nodesPrev.p; v.tris = sc->trisPrev.p; v.triMeta = sc->metaPrev.p;
        v.numNodes = sc->numNodesPrev; v.numTris = sc->numTrisPrev;
    }
    return v;
}

struct zr_gbuffer
{
    int device = 0; uint32_t w = 0, h = 0, x0 = 0, y0 = 0;
    // double-buffered like the reference's GBufferData (DefaultRendererImpl.h:80-130): every GBUFFER pass render flips
    // `cur`; the other set is the previous frame's G-buffer that the temporal passes of ReSTIR read
    // (a third set once the G-buffer is stream-tracked, zr_pass_set_frame_overlap: the GBUFFER render of frame N + 2 then runs while the temporal passes of
    // frame N + 1 still read the sets of frames N + 1 and N)
    DevBuf<uint8_t> planeSets[3][ZR_GB_COUNT];
    int cur = 0, numSets = 2; uint64_t numRendered = 0;
    int Next() const { return (cur + 1) % numSets; }
    int Prev() const { return (cur + numSets - 1) % numSets; }
    // the scene's material class (zr_scene::plainMaterials) at the time each plane set was rendered: the PLAIN kernel permutations take a pixel's flags as known,
    // so both the current and the previous frame's planes must come from a plain material table (a scene that BECAME plain renders one more frame with the general kernels)
    bool plainAt[3] = {true, true, true};
    // ... and WHICH scene rendered it (zr_scene::uid; 0 = never rendered by a GBUFFER pass): planes an engine filled itself through zr_gbuffer_device_plane, or
    // rendered from another scene than the lighting pass is given, carry flags the lighting pass cannot vouch for -- they run the general kernels
    uint64_t sceneAt[3] = {0, 0, 0};
    // stream tracking (zr_pass_set_frame_overlap): with the passes of a frame spread over two streams, a GBUFFER render must not overwrite a plane set
    // that a pass on another stream still reads, and a pass on another stream must not read a set before its GBUFFER render has finished
    bool tracked = false;
    hipEvent_t evWritten = nullptr; hipStream_t writer = nullptr; bool hasWrite = false;
    struct Reader { hipStream_t st; hipEvent_t ev; };
    std::vector<Reader> readers[3];      // per plane set: the last read of every stream that has read it
    ~zr_gbuffer() { if (evWritten) (void)hipEventDestroy(evWritten); for (auto& v : readers) for (auto& r : v) (void)hipEventDestroy(r.ev); }
    DevBuf<uint8_t>* Planes() { return planeSets[cur]; }
    const DevBuf<uint8_t>* Planes() const { return planeSets[cur]; }
    GBuf View() const { return ViewOf(cur); }
    GBuf PrevView() const { return ViewOf(Prev()); }
    GBuf ViewOf(int which) const
    {
        const DevBuf<uint8_t>* planes = planeSets[which];
        GBuf g; g.w = w; g.h = h; g.x0 = x0; g.y0 = y0;
        g.baseColor = (uint32_t*)planes[ZR_GB_BASE_COLOR].p; g.normal = (uint32_t*)planes[ZR_GB_NORMAL].p;
        g.mr = (uint16_t*)planes[ZR_GB_METALLIC_ROUGHNESS].p; g.motion = (uint32_t*)planes[ZR_GB_MOTION_VECTOR].p;
        g.emissive = (uint32_t*)planes[ZR_GB_EMISSIVE_COLOR].p; g.ior = (uint8_t*)planes[ZR_GB_IOR].p;
        g.coat = (uint16_t*)planes[ZR_GB_COAT].p; g.depth = (float*)planes[ZR_GB_DEPTH].p;
        g.triA = (uint32_t*)planes[ZR_GB_TRI_DIFF_GEO_A].p; g.triB = (uint32_t*)planes[ZR_GB_TRI_DIFF_GEO_B].p;
        return g;
    }
};

// tracked G-buffers: `s` has finished reading plane set `set` at this point of its queue
static int GBufferMarkRead(zr_gbuffer* gb, int set, hipStream_t s)
{
    if (!gb || !gb->tracked) return ZR_OK;
    for (auto& r : gb->readers[set]) if (r.st == s) { HIP_TRY(hipEventRecord(r.ev, s)); return ZR_OK; }
    hipEvent_t ev; HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    gb->readers[set].push_back({s, ev});
    HIP_TRY(hipEventRecord(ev, s));
    return ZR_OK;
}
// ... a pass on `s` is about to read the current planes: behind their GBUFFER render if that ran on another stream
static int GBufferAcquireRead(zr_gbuffer* gb, hipStream_t s)
{
    if (!gb || !gb->tracked || !gb->hasWrite || gb->writer == s) return ZR_OK;
    HIP_TRY(hipStreamWaitEvent(s, gb->evWritten, 0));
    return ZR_OK;
}
// ... the GBUFFER pass on `s` is about to overwrite plane set `set`: behind every other stream's last read of it
static int GBufferAcquireWrite(zr_gbuffer* gb, int set, hipStream_t s)
{
    if (!gb->tracked) return ZR_OK;
    for (auto& r : gb->readers[set]) if (r.st != s) HIP_TRY(hipStreamWaitEvent(s, r.ev, 0));
    return ZR_OK;
}
static int GBufferMarkWritten(zr_gbuffer* gb, hipStream_t s)
{
    if (!gb->tracked) return ZR_OK;
    if (!gb->evWritten) HIP_TRY(hipEventCreateWithFlags(&gb->evWritten, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(gb->evWritten, s));
    gb->writer = s; gb->hasWrite = true;
    return ZR_OK;
}

struct QueueStorage
{
    DevBuf<U4> s0; DevBuf<F4> f[8]; DevBuf<F4> rays[6]; DevBuf<uint32_t> lightID; DevBuf<U4> hitC, hitM; DevBuf<uint32_t> visS, rayList;
    DevBuf<F4> t[8];      // ray-differential state, allocated by the first render of a textured scene
    int AllocTex(size_t cap)
    {
        int r;
        for (auto& b : t) if (b.n != cap && (r = b.Alloc(cap))) return r;
        return ZR_OK;
    }
    int Alloc(size_t cap)
    {
        int r;
        if ((r = s0.Alloc(cap))) return r;
        for (auto& b : f) if ((r = b.Alloc(cap))) return r;
        for (auto& b : rays) if ((r = b.Alloc(cap))) return r;
        if ((r = lightID.Alloc(cap))) return r;
        if ((r = hitC.Alloc(cap))) return r;
        if ((r = hitM.Alloc(cap))) return r;
        if ((r = visS.Alloc(cap))) return r;
        if ((r = rayList.Alloc(3 * cap))) return r;
        return ZR_OK;
    }
    PathQueue View() const
    {
        PathQueue q;
        q.s0 = s0.p; q.s1 = f[0].p; q.s2 = f[1].p; q.s3 = f[2].p; q.s4 = f[3].p; q.s5 = f[4].p; q.s6 = f[5].p; q.s7 = f[6].p; q.s8 = f[7].p;
        q.rayC_o = rays[0].p; q.rayC_d = rays[1].p; q.rayM_o = rays[2].p; q.rayM_d = rays[3].p; q.rayS_o = rays[4].p; q.rayS_d = rays[5].p;
        q.sLightID = lightID.p; q.hitC = hitC.p; q.hitM = hitM.p; q.visS = visS.p; q.rayList = rayList.p;
        for (int k = 0; k < 8; k++) q.t[k] = t[k].p;
        return q;
    }
};

static constexpr uint32_t kRptListWords = 12;      // replay work-list counts + cursors of the ReSTIR PT pass (layout: where the buffer is allocated)
// the node count from which K11 runs as k_rpt_pathtrace_w4 (4 waves per SIMD + the top 32 nodes of the tree in LDS).  Rounds 3 - 5: 16384 (1 MB of nodes = "does not fit
// the caches"; smaller scenes tied).  Round 6, re-measured on the collapsed trees (profiles/r06w_ab_k11_node_cache_small_scenes.txt): the cached build wins at every size --
// Cornell PLAIN K11 0.699 -> 0.679 ms (its 13 nodes are all in LDS), general Cornell 0.817 -> 0.803, sun + sky 0.899 -> 0.874, a 20 k-triangle scene 5.69 -> 5.26 -- so every
// scene that has a tree takes it; the other instantiation stays selectable (zr_debug_set_large_scene_nodes) and covered by the parity tests.
static constexpr uint32_t kLargeSceneNodes = 1;
static std::atomic<uint32_t> g_largeSceneNodes{kLargeSceneNodes};
static std::atomic<bool> g_materialClassKernels{true};      // zr_debug_set_material_class_kernels: plain scenes run the PLAIN kernel permutations
// the PLAIN kernel permutations apply: the scene's material table is of the plain class (and has no texture heap)
static bool PlainClass(const zr_scene* sc, const zr_gbuffer* gb)
{
    if (!sc->plainMaterials.load(std::memory_order_relaxed) || !g_materialClassKernels.load(std::memory_order_relaxed)) return false;
    const int c = gb->cur;
    if (!(gb->plainAt[c] && gb->sceneAt[c] == sc->uid)) return false;
    // the other set is the previous frame's G-buffer; nothing reads it before the second GBUFFER render (havePrevGBuffer)
    if (gb->numRendered >= 2 && !(gb->plainAt[gb->Prev()] && gb->sceneAt[gb->Prev()] == sc->uid)) return false;
    return true;
}
static constexpr int kMaxRounds = 16;
static constexpr int kMaxTimers = 64;
// ray-counter slots (pairs of u64 on the device): 0 = wavefront path tracer, 1.. = ReSTIR PT kernels in launch order
static constexpr int kCounterSlots = 16;
static const char* const kCounterNames[kCounterSlots] = {"trace", "rpt_pathtrace", "rpt_replay_ctt", "rpt_replay_ttc", "rpt_reconnect_temporal",
    "rpt_replay_cts", "rpt_replay_stc", "rpt_reconnect_spatial", "rdi_temporal", "rdi_spatial", "rgi", "sdi_temporal", "sdi_spatial", "", "", ""};

struct zr_pass
{
    int kind = 0, device = 0, integrator = 0;
    bool initialized = false;
    uint32_t w = 0, h = 0;
    zr_params params{};
    // INDIRECT
    QueueStorage q[2];
    DevBuf<uint32_t> skyLut;                       // ZR_PASS_SKY: R11G11B10F texels
    DevBuf<float> finalRGBA; DevBuf<F4> firstBOP; DevBuf<uint32_t> counts; DevBuf<unsigned long long> counters;
    DevBuf<uint32_t> groupMax;      // kMaxRounds x (8x8 groups of the tile): RR reduction keys
    zr_counters hostCounters{0, 0};
    // GBUFFER: GBufferRT::PickPixel (GBufferRT.h:36-46).  pickXY = x | y << 16, 0xffffffff = no pick pending; pickBuf[0] = the last picked mesh index
    uint32_t pickXY = 0xffffffffu; DevBuf<uint32_t> pickBuf; bool pickWritten = false;
    // INDIRECT / ReSTIR PT: two reservoir sets (7 planes each), two r-buffers, target, spatial neighbour, sample set
    struct ResStorage
    {
        DevBuf<uint32_t> A, G; DevBuf<float> B, F; DevBuf<U4> C, D; DevBuf<uint16_t> E;
        int Alloc(size_t n)
        {
            int r;
            if ((r = A.Alloc(n)) || (r = B.Alloc(2 * n)) || (r = C.Alloc(n)) || (r = D.Alloc(n)) || (r = E.Alloc(n)) || (r = F.Alloc(2 * n)) || (r = G.Alloc(2 * n))) return r;
            return ZR_OK;
        }
        rpt::ResPlanes View() const { rpt::ResPlanes v; v.A = A.p; v.B = B.p; v.C = C.p; v.D = D.p; v.E = E.p; v.F = F.p; v.G = G.p; return v; }
    } res[3];      // [2]: allocated by zr_pass_set_frame_overlap (the set K11 of the next frame writes while this frame's reuse passes read the other two)
    struct RBufStorage
    {
        DevBuf<uint16_t> A, D; DevBuf<U4> B, C;
        int Alloc(size_t n) { int r; if ((r = A.Alloc(4 * n)) || (r = B.Alloc(n)) || (r = C.Alloc(n)) || (r = D.Alloc(n))) return r; return ZR_OK; }
        rpt::RBuf View() const { rpt::RBuf v; v.A = A.p; v.B = B.p; v.C = C.p; v.D = D.p; return v; }
    } rb[2];
    DevBuf<F4> rptTarget, rptTargetAlt; DevBuf<uint8_t> rptNeighbor; DevBuf<uint16_t> rptSampleSet;
    // Frame overlap (zr_pass_set_frame_overlap).  rptSet: which storage plays the two roles currIdx flips between ([0], [1]) and which one is free ([2]);
    // the CANDIDATES stage of an overlapped frame writes the free set and hands the set it replaces back.  tgtIdx / finIdx: the target / FINAL plane of
    // the frame whose CANDIDATES stage ran last; finOut: the FINAL plane of the last frame whose final stage has been enqueued (zr_pass_get_output)
    bool overlap = false, overlapCarry = false; int rptSet[3] = {0, 1, 2}; int tgtIdx = 0, finIdx = 0, finOut = 0;
    DevBuf<float> finalAlt, finalAlt2;      // FINAL rotates over three planes: K11 of frame N + 2 (which clears the pixels without a surface) may run while frame N's is still being consumed
    // the last stage of the frame before the previous one done (depth-2 pipelining, product mode): evDone[k & 1] is recorded when frame k closes
    hipEvent_t evDone[2] = {nullptr, nullptr}; bool haveDone[2] = {false, false}; hipStream_t doneStream[2] = {nullptr, nullptr}; uint64_t ovFrame = 0;
    hipStream_t overlapStream = nullptr;                     // "stream A" for callers without streams of their own
    hipEvent_t evCand = nullptr, evTemporal = nullptr;       // K11 of the open frame done (on candStream) / K14 of the last frame done (on reuseStream)
    hipStream_t candStream = nullptr, reuseStream = nullptr; bool haveCand = false, haveTemporal = false;
    zr_pass::ResStorage& RptCur() { return res[rptSet[currIdx]]; }
    zr_pass::ResStorage& RptOth() { return res[rptSet[1 - currIdx]]; }
    const zr_pass::ResStorage& RptOth() const { return res[rptSet[1 - currIdx]]; }
    F4* Target() const { return tgtIdx ? rptTargetAlt.p : rptTarget.p; }
    float* Final(int i) const { return i == 0 ? finalRGBA.p : (i == 1 ? finalAlt.p : finalAlt2.p); }
    DevBuf<uint16_t> rptMap[2];      // K12 thread maps: [0] CtN, [1] NtC
    DevBuf<uint32_t> trip; DevBuf<unsigned long long> tripStats;      // ZR_K11=trip diagnostic
    DevBuf<uint32_t> carry[2], carryCount;                             // K11 with per-bounce compaction: path-state planes (ping-pong), alive counts
    bool frameOpen = false;      // ReSTIR PT staged rendering: a frame's TEMPORAL stage has run, its last stage has not
    DevBuf<uint32_t> costMap; bool costOn = false, costRays = false;      // rays per 32 x 32-px cell (zr_pass_enable_cost_map)
    DevBuf<uint32_t> rptLists, rptListCounts;      // 4 replay work lists (pixel ids) + their device-side counts
    // DI_EMISSIVE: two reservoir sets (A RGBA32_UINT, B RG32F), target, sample set
    DevBuf<U4> diA[2]; DevBuf<float> diB[2]; DevBuf<F4> diTarget; DevBuf<uint16_t> diSampleSet;
    DevBuf<uint8_t> skyA[2]; DevBuf<uint16_t> skyB[2]; DevBuf<float> skyC[2];      // sun + sky DI reservoirs (R8_UINT, RG16_UINT, RG32F)
    // INDIRECT / ReSTIR GI: two reservoir sets (A RGBA32F, B RGBA16F, C RGBA32F)
    DevBuf<F4> giA[2], giC[2]; DevBuf<uint16_t> giB[2];
    bool temporalValid = false, doTemporal = false, doSpatial = false; int currIdx = 0;
    const F4* compIn[4] = {nullptr, nullptr, nullptr, nullptr};     // COMPOSITING inputs (emissive DI, indirect, sky DI); [3] = TAA signal
    DevBuf<F4> svgfHist, svgfAccum, svgfPing, svgfPong; DevBuf<float> svgfMoments[2], svgfGuideFw; DevBuf<svgf::GuideN> svgfGuide; int svgfMomIdx = 0; const F4* svgfOut = nullptr; F4* svgfCur = nullptr; uint32_t svgfStepsDone = 0;      // DENOISE
    DevBuf<uint16_t> taaOut[2]; int taaIdx = 0;            // TAA: ping-pong RGBA16F outputs; taaIdx = the one written last
    // AUTO_EXPOSURE / DISPLAY
    const uint16_t* postIn16 = nullptr; const F4* postIn32 = nullptr; const float* exposureIn = nullptr;
    DevBuf<uint32_t> aeHist; DevBuf<float> aeExposure;
    DevBuf<uint32_t> tonemapLut; uint32_t tonemapLutDim = 0; DevBuf<F4> displayOut; DevBuf<uint32_t> displaySrgb;
    uint32_t own[4] = {0, 0, 0, 0};                // owned rect (global pixels); w == 0 -> the whole G-buffer rect
    // PRELIGHTING
    DevBuf<float> power;
    // deferred alias-table rebuild (zr_scene_invalidate_alias_table_deferred): K2's powers are read back into pinned memory behind an event
    float* powerHost = nullptr; size_t powerHostCap = 0; hipEvent_t powerEv = nullptr; bool powerPending = false;
    // timing
    bool timing = false;
    struct Timer { const char* name; hipEvent_t a, b; bool used; };
    std::vector<Timer> timers;
    int numTimers = 0;
    std::vector<std::string> timerNames; std::vector<float> timerMs; std::vector<uint32_t> timerLaunches;
};

// ------------------------------------------------------------------------------------------------ timing helpers
static void TimerBegin(zr_pass* p, hipStream_t s, const char* name)
{
    if (!p->timing) return;
    if (p->numTimers >= (int)p->timers.size())
    {
        zr_pass::Timer t; t.name = name; t.used = false;
        if (hipEventCreate(&t.a) != hipSuccess || hipEventCreate(&t.b) != hipSuccess) return;
        p->timers.push_back(t);
    }
    zr_pass::Timer& t = p->timers[p->numTimers];
    t.name = name; t.used = true;
    (void)hipEventRecord(t.a, s);
}
static void TimerEnd(zr_pass* p, hipStream_t s)
{
    if (!p->timing) return;
    if (p->numTimers >= (int)p->timers.size()) return;
    (void)hipEventRecord(p->timers[p->numTimers].b, s);
    p->numTimers++;
}

// ------------------------------------------------------------------------------------------------ alias table (host)
// Math::KahanSum (Source/ZetaCore/Math/Common.cpp:72-139) with the reference's AVX2 lane structure.  `phase` floats of
// scalar head emulate a pointer that is `phase` floats short of 32-byte alignment (0 for the reference's allocations).
static float KahanSumAVX2(const float* data, int64_t N, int64_t phase)
{
    float sum = 0.0f, compensation = 0.0f;
    const int64_t start = phase < N ? phase : N;
    for (int64_t i = 0; i < start; i++)
    {
        volatile float corrected = data[i] - compensation;
        volatile float newSum = sum + corrected;
        compensation = (newSum - sum) - corrected;
        sum = newSum;
    }
    int64_t numSimd = (N - start);
    numSimd -= numSimd & 15;
    __m256 vSum = _mm256_setzero_ps(), vComp = _mm256_setzero_ps();
    for (int64_t c = start; c < start + numSimd; c += 16)
    {
        __m256 V1 = _mm256_loadu_ps(data + c), V2 = _mm256_loadu_ps(data + c + 8);
        __m256 vCurr = _mm256_add_ps(V1, V2);
        __m256 vCorrected = _mm256_sub_ps(vCurr, vComp);
        __m256 vNewSum = _mm256_add_ps(vSum, vCorrected);
        vComp = _mm256_sub_ps(vNewSum, vSum);
        vComp = _mm256_sub_ps(vComp, vCorrected);
        vSum = vNewSum;
    }
    alignas(32) float simdSum[8], simdComp[8];
    _mm256_store_ps(simdSum, vSum); _mm256_store_ps(simdComp, vComp);
    for (int i = 0; i < 8; i++)
    {
        volatile float corrected = simdSum[i] - compensation - simdComp[i];
        volatile float newSum = sum + corrected;
        compensation = (newSum - sum) - corrected;
        sum = newSum;
    }
    for (int64_t i = start + numSimd; i < N; i++)
    {
        volatile float corrected = data[i] - compensation;
        volatile float newSum = sum + corrected;
        compensation = (newSum - sum) - corrected;
        sum = newSum;
    }
    return sum;
}

// BuildAliasTable (Source/ZetaRenderPass/PreLighting/PreLighting.cpp:27-158): Vose's method with LIFO index stacks.
static void BuildAliasTableHost(std::vector<float>& probs, zr_alias_entry* table, uint32_t phase)
{
    const int64_t N = (int64_t)probs.size();
    const float oneDivN = 1.0f / (float)N;
    const float sum = KahanSumAVX2(probs.data(), N, phase);
    const float sumRcp = (float)N / sum;
    for (int64_t i = 0; i < N; i++) probs[i] *= sumRcp;
    for (int64_t i = 0; i < N; i++) { table[i].cached_p_orig = probs[i] * oneDivN; table[i].alias = 0xffffffffu; table[i].p_curr = 0.0f; }
    std::vector<uint32_t> larger, smaller;
    larger.reserve(N); smaller.reserve(N);
    for (int64_t i = 0; i < N; i++) (probs[i] < 1.0f ? smaller : larger).push_back((uint32_t)i);
    while (!smaller.empty() && !larger.empty())
    {
        const uint32_t si = smaller.back(); smaller.pop_back();
        const float sp = probs[si];
        const uint32_t li = larger.back();
        float lp = probs[li];
        table[si].alias = li; table[si].p_curr = sp;
        lp = (sp + lp) - 1.0f;
        probs[li] = lp;
        if (lp < 1.0f) { larger.pop_back(); smaller.push_back(li); }
    }
    for (; !larger.empty(); larger.pop_back()) { table[larger.back()].alias = larger.back(); table[larger.back()].p_curr = 1.0f; }
    for (; !smaller.empty(); smaller.pop_back()) { table[smaller.back()].alias = smaller.back(); table[smaller.back()].p_curr = 1.0f; }
    for (int64_t i = 0; i < N; i++) table[i].cached_p_alias = table[table[i].alias].cached_p_orig;
}

// ------------------------------------------------------------------------------------------------ C-ABI
// ---- device-side BVH refit (zr_scene_update_instances): the topology of the BVH4 stays, triangles are re-transformed and node boxes are
// recomputed bottom-up, one launch per tree level (deepest first; levels are a few dozen at most and the nodes of a level are independent).
// World-space triangles: the builder's expression (zr_bvh.h Build), so a refit scene traces the same triangles a rebuilt one would.
__global__ void __launch_bounds__(256) k_refit_tris(BvhTri* tris, uint32_t n, const TriMeta* meta, const zr_mesh_instance* instances, const float* toWorld,
    const zr_vertex* vertices, const uint32_t* indices)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    BvhTri t = tris[i];
    const TriMeta tm = meta[t.gidx];
    const zr_mesh_instance mi = instances[tm.mesh];
    const float* M = toWorld + 12 * (size_t)tm.mesh;
    float w[3][3];
    for (int k = 0; k < 3; k++)
    {
        const uint32_t vi = indices[mi.base_idx_offset + 3 * tm.prim + k] + mi.base_vtx_offset;
        const float* P = vertices[vi].pos;
        for (int r = 0; r < 3; r++) w[k][r] = M[4 * r + 0] * P[0] + M[4 * r + 1] * P[1] + M[4 * r + 2] * P[2] + M[4 * r + 3];
    }
    for (int r = 0; r < 3; r++) { t.v0[r] = w[0][r]; t.e1[r] = w[1][r] - w[0][r]; t.e2[r] = w[2][r] - w[0][r]; }
    tris[i] = t;
}
// A tree built in the background arrives as topology only (zr_scene::Background): child references per node, scene triangle per leaf slot.  The
// slot's constant words are the builder's (zr_bvh.h Build: mask of the instance, hashed triangle ID); vertices follow in k_refit_tris, boxes in k_refit_level.
__global__ void __launch_bounds__(256) k_install_topology(Bvh4Node* nodes, const uint32_t* children, uint32_t numNodes, BvhTri* tris, const uint32_t* slotTri, uint32_t numTris,
    const TriMeta* meta, const uint8_t* instanceMask)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < numNodes)
    {
        const uint4 c = ((const uint4*)children)[i];
        Bvh4Node N; N.ox = N.oy = N.oz = 0.0f; N.exps = 0; N.child[0] = c.x; N.child[1] = c.y; N.child[2] = c.z; N.child[3] = c.w;
        N.qlox = N.qloy = N.qloz = N.qhix = N.qhiy = N.qhiz = 0; N.pad0 = 0; N.pad1 = 0;
        nodes[i] = N;
    }
    if (i < numTris)
    {
        const uint32_t g = slotTri[i];
        const TriMeta tm = meta[g];
        BvhTri t; for (int r = 0; r < 3; r++) { t.v0[r] = 0.0f; t.e1[r] = 0.0f; t.e2[r] = 0.0f; }
        t.gidx = g; t.mask = instanceMask[tm.mesh]; t.id = TriID(tm.mesh, tm.prim);
        tris[i] = t;
    }
}
// bounds of leaf `c` (1-ulp padded like the builder's, since v0 + e1 is a rounded v1)
__device__ __forceinline__ void LeafBounds(const BvhTri* tris, uint32_t c, float lo[3], float hi[3])
{
    const uint32_t first = (c & 0x7fffffffu) >> 3, count = (c & 7u) + 1u;
    for (int r = 0; r < 3; r++) { lo[r] = 3.402823466e+38f; hi[r] = -3.402823466e+38f; }
    for (uint32_t i = first; i < first + count; i++)
    {
        const BvhTri t = tris[i];
        for (int r = 0; r < 3; r++)
        {
            const float a = t.v0[r], b = t.v0[r] + t.e1[r], cc = t.v0[r] + t.e2[r];
            lo[r] = fminf(lo[r], zr::PrevFloat32(fminf(a, fminf(b, cc)))); hi[r] = fmaxf(hi[r], zr::NextFloat32(fmaxf(a, fmaxf(b, cc))));
        }
    }
}
// one level: node = levelNodes[i]; its inner children were finished by the previous (deeper) launch.  Quantisation = BvhBuilder::Collapse:
// origin = the node's min corner, per-axis power-of-two scale whose 255 steps reach the max corner, child planes rounded outwards and verified
// against the traversal's decode expression fma(q, scale, origin).
__global__ void __launch_bounds__(64) k_refit_level(Bvh4Node* nodes, const uint32_t* levelNodes, uint32_t n, const BvhTri* tris, float* nodeBounds)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t ni = levelNodes[i];
    Bvh4Node N = nodes[ni];
    float clo[4][3], chi[4][3];
    for (int c = 0; c < 4; c++)
    {
        const uint32_t ref = N.child[c];
        if (ref == kEmptyChild) continue;
        if (ref & kLeafBit) LeafBounds(tris, ref, clo[c], chi[c]);
        else for (int r = 0; r < 3; r++) { clo[c][r] = nodeBounds[6 * (size_t)ref + r]; chi[c][r] = nodeBounds[6 * (size_t)ref + 3 + r]; }
    }
    float lo[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f}, hi[3] = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
    for (int c = 0; c < 4; c++) if (N.child[c] != kEmptyChild) for (int r = 0; r < 3; r++) { lo[r] = fminf(lo[r], clo[c][r]); hi[r] = fmaxf(hi[r], chi[c][r]); }
    uint32_t ex[3], qlo[3] = {0, 0, 0}, qhi[3] = {0, 0, 0};
    for (int r = 0; r < 3; r++)
    {
        int e = 1;
        const float ext = hi[r] - lo[r];
        if (ext > 0) { int ee; (void)frexpf(ext / 255.0f, &ee); e = max(1, min(254, ee + 126 - 1)); }
        while (e < 254 && !(fmaf(255.0f, zr_asfloat((uint32_t)e << 23), lo[r]) >= hi[r])) e++;
        ex[r] = (uint32_t)e;
        const float sc = zr_asfloat(ex[r] << 23);
        for (int c = 0; c < 4; c++)
        {
            if (N.child[c] == kEmptyChild) continue;
            int a = (int)floorf((clo[c][r] - lo[r]) / sc); a = max(0, min(255, a));
            while (a > 0 && fmaf((float)a, sc, lo[r]) > clo[c][r]) a--;
            int b = (int)ceilf((chi[c][r] - lo[r]) / sc); b = max(0, min(255, b));
            while (b < 255 && fmaf((float)b, sc, lo[r]) < chi[c][r]) b++;
            qlo[r] |= (uint32_t)a << (8 * c); qhi[r] |= (uint32_t)b << (8 * c);
        }
    }
    N.ox = lo[0]; N.oy = lo[1]; N.oz = lo[2]; N.exps = ex[0] | (ex[1] << 8) | (ex[2] << 16);
    N.qlox = qlo[0]; N.qloy = qlo[1]; N.qloz = qlo[2]; N.qhix = qhi[0]; N.qhiy = qhi[1]; N.qhiz = qhi[2];
    nodes[ni] = N;
    // what the parent sees: the box the quantised planes of the LARGEST child extent decode to would be looser; the exact union is enough,
    // because the parent quantises it outwards again
    for (int r = 0; r < 3; r++) { nodeBounds[6 * (size_t)ni + r] = lo[r]; nodeBounds[6 * (size_t)ni + 3 + r] = hi[r]; }
}

// ---- fused halo transfer: every plane x every rect in ONE launch (the per-plane hipMemcpy2DAsync path above costs planes x peers copies:
// 49 per exchange for ReSTIR PT on a 4 x 2 tile grid, each a few microseconds of launch latency for strips of a few hundred KB)
struct HaloJob
{
    struct Plane { char* base; uint32_t bpp; } planes[8];
    struct Rect { uint32_t first, w, h; uint32_t pad; uint64_t offset; } rects[ZR_HALO_MAX_RECTS];      // first = index of the rect's top-left pixel in the planes
    uint32_t numPlanes, numRects, pitch;      // pitch = plane width in pixels
    char* buf;
};
template<bool PACK> __global__ void __launch_bounds__(256) k_halo(HaloJob J)
{
    const HaloJob::Rect r = J.rects[blockIdx.y];
    const uint32_t n = r.w * r.h;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    {
        const size_t px = (size_t)r.first + (size_t)(i / r.w) * J.pitch + (i % r.w);
        char* cursor = J.buf + r.offset;
        for (uint32_t k = 0; k < J.numPlanes; k++)
        {
            const uint32_t bpp = J.planes[k].bpp;
            char* a = J.planes[k].base + px * bpp;      // in the plane
            char* b = cursor + (size_t)i * bpp;         // in the packed block (planes back to back, rows of the rect contiguous)
            char* dst = PACK ? b : a; const char* src = PACK ? a : b;
            switch (bpp)
            {
            case 16: *(uint4*)dst = *(const uint4*)src; break;
            case 8: *(uint2*)dst = *(const uint2*)src; break;
            case 4: *(uint32_t*)dst = *(const uint32_t*)src; break;
            case 2: *(uint16_t*)dst = *(const uint16_t*)src; break;
            default: *dst = *src; break;
            }
            cursor += (size_t)n * bpp;
        }
    }
}

extern "C" {

int zr_abi_version(void) { return ZR_ABI_VERSION; }
const char* zr_last_error(void) { return g_err.c_str(); }

int zr_device_count(int* count)
{
    if (!count) return Fail(ZR_ERR_INVALID_ARG, "count is null");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    *count = (e == hipSuccess) ? n : 0;
    return ZR_OK;
}

int zr_device_probe_run(int device, float min_ms, zr_device_probe* out)
{
    if (!out) return Fail(ZR_ERR_INVALID_ARG, "out is null");
    if (int r = RequireDevice(device)) return r;
    std::string err;
    if (zr::DeviceProbeRun(device, min_ms, out, err)) return Fail(ZR_ERR_HIP, "zr_device_probe_run: %s", err.c_str());
    return ZR_OK;
}

int zr_selftest_half_conversions(int device, uint64_t* m0, uint64_t* m1)
{
    if (!m0 || !m1) return Fail(ZR_ERR_INVALID_ARG, "null argument");
    int r = RequireDevice(device);
    if (r) return r;
    DevBuf<unsigned long long> bad;
    if ((r = bad.Alloc(2))) return r;
    HIP_TRY(hipMemset(bad.p, 0, 2 * sizeof(unsigned long long)));
    hipLaunchKernelGGL(k_selftest_half, dim3(256 * 64), dim3(256), 0, 0, bad.p);
    HIP_TRY(hipGetLastError());
    unsigned long long h[2];
    HIP_TRY(hipMemcpy(h, bad.p, sizeof(h), hipMemcpyDeviceToHost));
    *m0 = h[0]; *m1 = h[1];
    return ZR_OK;
}

int zr_params_default(zr_params* p)
{
    if (!p) return Fail(ZR_ERR_INVALID_ARG, "params is null");
    memset(p, 0, sizeof(*p));
    p->flags = ZR_IND_TEMPORAL_RESAMPLE | ZR_IND_SPATIAL_RESAMPLE | ZR_IND_RUSSIAN_ROULETTE | ZR_IND_BOILING_SUPPRESSION |
               ZR_IND_SORT_TEMPORAL | ZR_IND_SORT_SPATIAL;
    p->max_non_tr_bounces = 3; p->max_glossy_tr_bounces = 4; p->m_max_temporal = 10; p->m_max_spatial = 8;
    p->alpha_min = 0.175f * 0.175f; p->presampling = 0; p->num_sample_sets = 128; p->sample_set_size = 512;
    // light voxel grid: off; VOXEL_GRID_DIM (32, 8, 40), VOXEL_EXTENTS (0.6, 0.45, 0.6), y offset 0.1 (DefaultRendererImpl.h:42-43, 73-77)
    p->taa_blend_weight = 0.1f;      // TAA.h:72
    p->ae_min_lum = 5e-3f; p->ae_max_lum = 4.0f; p->ae_lum_map_exp = 0.5f; p->ae_adaptation_rate = 1.0f;      // AutoExposure.h:73-81
    p->tex_filter = ZR_TEX_FILTER_ANISOTROPIC_4X;      // IndirectLighting.h:243
    p->svgf_alpha = 0.2f; p->svgf_alpha_moments = 0.2f; p->svgf_sigma_l = 4.0f; p->svgf_sigma_z = 1.0f; p->svgf_normal_power_log2 = 7; p->svgf_iterations = 5;   // ZR_PASS_DENOISE (zr_svgf.h)
    p->num_spatial_passes = 1;        // IndirectLighting.h:392
    p->display_tonemapper = ZR_TONEMAP_NEUTRAL; p->display_auto_exposure = 1; p->display_saturation = 1.0f; p->display_agx_exp = 1.0f;   // Display.cpp:69-74
    p->use_lvg = 0; p->lvg_grid_dim = 32u | (8u << 10) | (40u << 20);
    p->lvg_extents[0] = 0.6f; p->lvg_extents[1] = 0.45f; p->lvg_extents[2] = 0.6f; p->lvg_offset_y = 0.1f;
    return ZR_OK;
}

int zr_alias_table_build(const float* power, uint32_t n, uint32_t align_phase, zr_alias_entry* out)
{
    if (!power || !out || n == 0) return Fail(ZR_ERR_INVALID_ARG, "zr_alias_table_build: null/empty input");
    std::vector<float> probs(power, power + n);
    BuildAliasTableHost(probs, out, align_phase);
    return ZR_OK;
}

// Tree levels of a BVH4 (root = node 0), deepest first: what the refit walks bottom-up
static void BvhLevels(const std::vector<Bvh4Node>& nodes, std::vector<uint32_t>& order, std::vector<uint32_t>& offsets)
{
    order.clear(); offsets.clear();
    if (nodes.empty()) return;
    std::vector<std::vector<uint32_t>> levels(1, std::vector<uint32_t>(1, 0u));
    for (size_t l = 0; l < levels.size(); l++)
    {
        std::vector<uint32_t> next;
        for (uint32_t ni : levels[l]) for (int c = 0; c < 4; c++) { const uint32_t r = nodes[ni].child[c]; if (r != kEmptyChild && !(r & kLeafBit)) next.push_back(r); }
        if (!next.empty()) levels.push_back(std::move(next));
    }
    for (size_t l = levels.size(); l-- > 0;) { offsets.push_back((uint32_t)order.size()); order.insert(order.end(), levels[l].begin(), levels[l].end()); }
    offsets.push_back((uint32_t)order.size());
}

// Builds the CURRENT acceleration structure on the device from the scene's current instance buffer + object-to-world matrices (both already
// on the device, or enqueued on `st`): LBVH topology (zr_tu_bvh.hip), then the level-by-level box computation + quantisation of the refit.
// Host-synchronous (the level structure comes back).  The node / triangle buffers are grown to the build's capacity on first use.
static int DeviceRebuild(zr_scene* s, hipStream_t st)
{
    const uint32_t nt = (uint32_t)s->meta.n;
    int r;
    if (s->nodes.n < nt || s->tris.n < nt || s->nodeBounds.n < 6 * (size_t)nt)
    {
        HIP_TRY(hipDeviceSynchronize());
        if (s->nodes.n < nt && (r = s->nodes.Alloc(nt))) return r;
        if (s->tris.n < nt && (r = s->tris.Alloc(nt))) return r;
        if (s->nodeBounds.n < 6 * (size_t)nt && (r = s->nodeBounds.Alloc(6 * (size_t)nt))) return r;
    }
    if (!s->dMask.p && (r = s->dMask.Upload(s->hMask.data(), s->hMask.size()))) return r;
    DeviceBvhInputs in; in.numTris = nt; in.meta = s->meta.p; in.instances = s->instances.p; in.toWorld = s->toWorld.p; in.vertices = s->vertices.p;
    in.indices = s->indices.p; in.instanceMask = s->dMask.p;
    DeviceBvhOutputs out; out.tris = s->tris.p; out.nodes = s->nodes.p; out.nodeCap = (uint32_t)s->nodes.n;
    std::string err;
    if (DeviceBuildBvh4(st, s->bvhScratch, in, out, err)) return Fail(ZR_ERR_HIP, "device BVH build: %s", err.c_str());
    if (out.stackNeed + 1 > (uint32_t)kTravStack) return Fail(ZR_ERR_UNSUPPORTED, "device-built BVH has %u levels: needs %u traversal stack entries (limit %d); use the host builder", out.numLevels, out.stackNeed, kTravStack - 1);
    s->hLevelOrder = out.levelOrder; s->levelOffsets = out.levelOffsets;
    if (s->levelNodes.n < s->hLevelOrder.size()) { HIP_TRY(hipStreamSynchronize(st)); if ((r = s->levelNodes.Alloc(nt))) return r; }
    HIP_TRY(hipMemcpyAsync(s->levelNodes.p, s->hLevelOrder.data(), s->hLevelOrder.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    for (size_t l = 0; l + 1 < s->levelOffsets.size(); l++)
    {
        const uint32_t first = s->levelOffsets[l], cnt = s->levelOffsets[l + 1] - first;
        hipLaunchKernelGGL(k_refit_level, dim3((cnt + 63) / 64), dim3(64), 0, st, s->nodes.p, s->levelNodes.p + first, cnt, s->tris.p, s->nodeBounds.p);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));          // hLevelOrder is pageable: its copy must have been staged and the kernels' errors seen
    SceneView& v = s->view;
    v.nodes = s->nodes.p; v.tris = s->tris.p; v.numNodes = out.numNodes; v.numTris = nt;
    if (out.numLevels > s->maxDepth) s->maxDepth = out.numLevels;
    s->deviceBuilt = true;
    return ZR_OK;
}

// SceneView::plain: no material of the table is metallic, transmissive, thin-walled or coated (the fields GetMaterialData turns into those lobes, Material.h:268-427)
static bool MaterialsArePlain(const std::vector<zr_material>& m)
{
    const uint32_t lobes = (1u << ZR_MAT_METALLIC_BIT) | (1u << ZR_MAT_TRANSMISSIVE_BIT) | (1u << ZR_MAT_THIN_WALLED_BIT);
    for (const zr_material& x : m) if ((x.coat_color_flags & lobes) || ((x.base_color_tex_subsurf_coat_weight >> 24) & 0xffu)) return false;
    return !m.empty();
}
int zr_scene_create(int device, const zr_scene_desc* d, zr_scene** out)
{
    if (!d || !out) return Fail(ZR_ERR_INVALID_ARG, "zr_scene_create: null argument");
    if (!d->vertices || !d->indices || !d->instances || !d->materials || !d->rho_lut || !d->instance_to_world || !d->instance_mask || !d->instance_num_tris)
        return Fail(ZR_ERR_INVALID_ARG, "zr_scene_create: incomplete scene description");
    int r = RequireDevice(device);
    if (r) return r;
    zr_scene* s = new (std::nothrow) zr_scene();
    if (!s) return Fail(ZR_ERR_OOM, "out of host memory");
    s->device = device;
    // ZR_BVH_BUILD=device: the acceleration structure is built on the GPU (LBVH, zr_tu_bvh.hip) instead of by the host's binned-SAH builder
    uint64_t totalTris = 0;
    for (uint32_t i = 0; i < d->num_instances; i++) totalTris += d->instance_num_tris[i];
    const char* buildEnv = std::getenv("ZR_BVH_BUILD");
    const bool deviceBuild = buildEnv && !std::strcmp(buildEnv, "device") && totalTris > BvhBuilder::kTinyScene;
    BuiltBvh bvh;
    if (deviceBuild)
    {   // only the global-order (mesh, primitive) table is made on the host; geometry never passes through it
        bvh.meta.reserve((size_t)totalTris);
        for (uint32_t i = 0; i < d->num_instances; i++) for (uint32_t p = 0; p < d->instance_num_tris[i]; p++) { TriMeta m; m.mesh = i; m.prim = p; bvh.meta.push_back(m); }
    }
    else
    {
        BvhBuilder builder;
        bvh = builder.Build(*d);
    }
    if (bvh.stackNeed + 1 > (uint32_t)kTravStack) { delete s; return Fail(ZR_ERR_UNSUPPORTED, "BVH needs %u traversal stack entries (limit %d)", bvh.stackNeed, kTravStack - 1); }
    s->maxDepth = bvh.maxDepth;
#define UP(buf, ptr, cnt) if ((r = s->buf.Upload(ptr, cnt))) { delete s; return r; }
    UP(vertices, d->vertices, d->num_vertices);
    UP(indices, d->indices, d->num_indices);
    UP(instances, d->instances, d->num_instances);
    UP(materials, d->materials, d->num_materials);
    UP(emissives, d->emissives, d->num_emissives);
    UP(nodes, bvh.nodes4.data(), bvh.nodes4.size());
    UP(tris, bvh.tris.data(), bvh.tris.size());
    UP(meta, bvh.meta.data(), bvh.meta.size());
    UP(rho, d->rho_lut, (size_t)d->rho_dim[0] * d->rho_dim[1] * d->rho_dim[2]);
    if (d->num_textures)
    {
        if (!d->textures || !d->texels) { delete s; return Fail(ZR_ERR_INVALID_ARG, "zr_scene_create: num_textures > 0 without textures / texels"); }
        for (uint32_t i = 0; i < d->num_textures; i++)
        {
            const zr_texture_desc& t = d->textures[i];
            const zr_tex_mip last = zr_tex_mip_of(&t, t.num_mips ? t.num_mips - 1u : 0u);
            const uint64_t end = last.offset + (uint64_t)last.w * last.h * (t.format == ZR_TEX_RG8 ? 2u : 4u);
            if (!t.num_mips || !t.width || !t.height || t.format > ZR_TEX_RG8 || (t.offset & 3u) || end > d->texel_bytes)
            { delete s; return Fail(ZR_ERR_INVALID_ARG, "zr_scene_create: texture %u is malformed or lies outside the texel blob", i); }
        }
        UP(texDescs, d->textures, d->num_textures);
        UP(texels, d->texels, d->texel_bytes);
    }
    UP(srgb, zr_srgb_to_linear_table, 256);
#undef UP
    SceneView& v = s->view;
    v.vertices = s->vertices.p; v.indices = s->indices.p; v.instances = s->instances.p; v.materials = s->materials.p;
    v.emissives = s->emissives.p; v.alias = nullptr; v.sampleSets = nullptr; v.sampleSetSize = 0; v.nodes = s->nodes.p; v.tris = s->tris.p; v.triMeta = s->meta.p;
    v.rho.data = s->rho.p; v.rho.dx = d->rho_dim[0]; v.rho.dy = d->rho_dim[1]; v.rho.dz = d->rho_dim[2];
    v.numEmissives = d->num_emissives; v.numNodes = (uint32_t)bvh.nodes4.size(); v.numTris = (uint32_t)bvh.tris.size();
    s->hMaterials.assign(d->materials, d->materials + d->num_materials);
    s->plainMaterials = MaterialsArePlain(s->hMaterials) && d->num_textures == 0;
    v.tex.descs = s->texDescs.p; v.tex.texels = s->texels.p; v.tex.srgb = s->srgb.p; v.tex.count = d->num_textures;
    // largest tex16 per descriptor table, so that zr_pass_render can reject frame constants whose table offsets would
    // send a texture fetch outside the heap
    for (uint32_t i = 0; i < d->num_materials; i++)
    {
        const zr_material& m = d->materials[i];
        const uint32_t t[4] = { m.base_color_tex_subsurf_coat_weight & 0xffffu, m.normal_tex_tr_depth & 0xffffu,
                                m.mr_tex_spec_roughness_coat_roughness & 0xffffu, m.emissive_tex_alpha_cutoff_coat_ior & 0xffffu };
        for (int k = 0; k < 4; k++) if (t[k] != ZR_INVALID_TEX && (int64_t)t[k] > s->maxTex[k]) s->maxTex[k] = t[k];
    }
    for (uint32_t i = 0; i < d->num_instances; i++)
        if (d->instances[i].base_color_tex != ZR_INVALID_TEX && (int64_t)d->instances[i].base_color_tex > s->maxTex[0]) s->maxTex[0] = d->instances[i].base_color_tex;
    for (uint32_t i = 0; i < d->num_emissives; i++)
    { const uint32_t t = d->emissives[i].packed_b & 0xffffu; if (t != ZR_INVALID_TEX && (int64_t)t > s->maxTex[3]) s->maxTex[3] = t; }
    { const char* e = getenv("ZR_SCENE_UPDATE"); s->bg.enabled = e && !strcmp(e, "refit_sah"); }      // (zr_scene_set_background_rebuild for every scene of the process)
    s->hVertices.assign(d->vertices, d->vertices + d->num_vertices); s->hIndices.assign(d->indices, d->indices + d->num_indices);
    s->hMask.assign(d->instance_mask, d->instance_mask + d->num_instances); s->hNumTris.assign(d->instance_num_tris, d->instance_num_tris + d->num_instances);
    s->hToWorld.assign(d->instance_to_world, d->instance_to_world + 12 * (size_t)d->num_instances); s->movedEver.assign(d->num_instances, 0);
    BvhLevels(bvh.nodes4, s->hLevelOrder, s->levelOffsets);
    if (!s->hLevelOrder.empty() && (r = s->levelNodes.Upload(s->hLevelOrder.data(), s->hLevelOrder.size()))) { delete s; return r; }
    if (deviceBuild)
    {
        if ((r = s->toWorld.Upload(d->instance_to_world, 12 * (size_t)d->num_instances)) || (r = DeviceRebuild(s, nullptr))) { delete s; return r; }
    }
    // uploads from pageable memory return once staged; renders on non-blocking streams must not start before the DMA has landed
    { hipError_t e = hipDeviceSynchronize(); if (e != hipSuccess) { delete s; return Fail(ZR_ERR_HIP, "hipDeviceSynchronize failed: %s", hipGetErrorString(e)); } }
    *out = s;
    return ZR_OK;
}

// TLAS / instance-buffer update of a frame (RtAccelerationStructure.cpp:382-506, 708-787): the current instance buffer and acceleration
// structure become the previous ones, the new ones follow the new object-to-world matrices.  Default: a **refit on the device** -- the BVH4 built
// at zr_scene_create keeps its topology, k_refit_tris re-transforms the triangles and k_refit_level recomputes + re-quantises the node boxes
// level by level (what a D3D12 TLAS / BLAS update with ALLOW_UPDATE does); ZR_SCENE_UPDATE=rebuild selects a full binned-SAH rebuild on the
// host instead (better trees after large motion, three orders of magnitude slower).  Results do not depend on the tree (zr_intersect.h).
// Texture indices of updated records raise the per-table maxima the descriptor-table bounds check of zr_pass_render relies on
static void RaiseMaxTex(zr_scene* s, int table, uint32_t tex) { if (tex != ZR_INVALID_TEX && (int64_t)tex > s->maxTex[table]) s->maxTex[table] = tex; }

int zr_scene_update_emissives_async(zr_scene* s, void* stream, const zr_emissive_triangle* triangles, uint32_t first, uint32_t count)
{
    if (!s || (!triangles && count)) return Fail(ZR_ERR_INVALID_ARG, "zr_scene_update_emissives: null argument");
    if ((uint64_t)first + count > s->emissives.n) return Fail(ZR_ERR_INVALID_ARG, "zr_scene_update_emissives: [%u, %u) exceeds the scene's %zu emissive triangles", first, first + count, s->emissives.n);
    if (!count) return ZR_OK;
    HIP_TRY(hipSetDevice(s->device));
    hipStream_t st = (hipStream_t)stream;
    int r; zr_scene::StageSlot* t;
    if ((r = StageAcquire(s, (size_t)count * sizeof(zr_emissive_triangle), &t))) return r;
    memcpy(t->host, triangles, (size_t)count * sizeof(zr_emissive_triangle));
    for (uint32_t i = 0; i < count; i++) RaiseMaxTex(s, 3, triangles[i].packed_b & 0xffffu);
    if ((r = SceneWaitUsers(s, st))) return r;          // kernels of earlier frames on other streams may still sample the old records
    HIP_TRY(hipMemcpyAsync(s->emissives.p + first, t->host, (size_t)count * sizeof(zr_emissive_triangle), hipMemcpyHostToDevice, st));
    if ((r = StageCommit(t, st))) return r;
    return SceneMarkUpdated(s, st);
}
int zr_scene_update_materials_async(zr_scene* s, void* stream, const zr_material* materials, uint32_t first, uint32_t count)
{
    if (!s || (!materials && count)) return Fail(ZR_ERR_INVALID_ARG, "zr_scene_update_materials: null argument");
    if ((uint64_t)first + count > s->materials.n) return Fail(ZR_ERR_INVALID_ARG, "zr_scene_update_materials: [%u, %u) exceeds the scene's %zu materials", first, first + count, s->materials.n);
    if (!count) return ZR_OK;
    HIP_TRY(hipSetDevice(s->device));
    hipStream_t st = (hipStream_t)stream;
    int r; zr_scene::StageSlot* t;
    if ((r = StageAcquire(s, (size_t)count * sizeof(zr_material), &t))) return r;
    memcpy(t->host, materials, (size_t)count * sizeof(zr_material));
    for (uint32_t i = 0; i < count; i++)
    {   // Material.h:268-427: base colour / normal / metallic-roughness / emissive texture ids (16 bits each)
        RaiseMaxTex(s, 0, materials[i].base_color_tex_subsurf_coat_weight & 0xffffu); RaiseMaxTex(s, 1, materials[i].normal_tex_tr_depth & 0xffffu);
        RaiseMaxTex(s, 2, materials[i].mr_tex_spec_roughness_coat_roughness & 0xffffu); RaiseMaxTex(s, 3, materials[i].emissive_tex_alpha_cutoff_coat_ior & 0xffffu);
    }
    if ((r = SceneWaitUsers(s, st))) return r;
    std::copy(materials, materials + count, s->hMaterials.begin() + first);
    s->plainMaterials = MaterialsArePlain(s->hMaterials) && s->view.tex.count == 0;
    HIP_TRY(hipMemcpyAsync(s->materials.p + first, t->host, (size_t)count * sizeof(zr_material), hipMemcpyHostToDevice, st));
    if ((r = StageCommit(t, st))) return r;
    return SceneMarkUpdated(s, st);
}
// the original host-synchronous entry points: the update on the null stream, then wait for it (errors of the copies / refit kernels surface here)
int zr_scene_update_emissives(zr_scene* s, const zr_emissive_triangle* triangles, uint32_t first, uint32_t count)
{
    int r = zr_scene_update_emissives_async(s, nullptr, triangles, first, count);
    if (r || !count) return r;
    HIP_TRY(hipDeviceSynchronize());
    return ZR_OK;
}
int zr_scene_update_materials(zr_scene* s, const zr_material* materials, uint32_t first, uint32_t count)
{
    int r = zr_scene_update_materials_async(s, nullptr, materials, first, count);
    if (r || !count) return r;
    HIP_TRY(hipDeviceSynchronize());
    return ZR_OK;
}
int zr_scene_invalidate_alias_table(zr_scene* s)
{
    if (!s) return Fail(ZR_ERR_INVALID_ARG, "zr_scene_invalidate_alias_table: null argument");
    // kernels of earlier frames keep sampling the old table through the pointer they were launched with; the device buffer is only
    // rewritten by zr_scene_set_alias_table, which orders itself behind them
    std::lock_guard<std::mutex> lock(s->mtx);
    s->view.alias = nullptr; s->aliasHost.clear();
    return ZR_OK;
}
// The reference's steady-state behaviour (EmissiveTriangleAliasTable::Render, PreLighting.cpp:527-540: "fence hasn't passed, returning ..."): the
// old table keeps being sampled until the re-estimated powers have come back and the new table has been uploaded -- a frame or two later --
// and no render call ever waits for the device.  zr_scene_invalidate_alias_table is the deterministic form (new table in the very next frame).
int zr_scene_invalidate_alias_table_deferred(zr_scene* s)
{
    if (!s) return Fail(ZR_ERR_INVALID_ARG, "zr_scene_invalidate_alias_table_deferred: null argument");
    std::lock_guard<std::mutex> lock(s->mtx);
    if (s->view.alias) s->aliasStale = true;      // without a table there is nothing to keep: the next PRELIGHTING render builds one (first-frame path)
    return ZR_OK;
}
int zr_scene_update_instances_async(zr_scene* s, void* stream, const zr_mesh_instance* instances, const float* instance_to_world, uint32_t n)
{
    if (!s || !instances || !instance_to_world) return Fail(ZR_ERR_INVALID_ARG, "zr_scene_update_instances: null argument");
    if (n != s->instances.n) return Fail(ZR_ERR_INVALID_ARG, "zr_scene_update_instances: %u instances, the scene has %zu", n, s->instances.n);
    HIP_TRY(hipSetDevice(s->device));
    hipStream_t st = (hipStream_t)stream;
    const char* modeEnv = std::getenv("ZR_SCENE_UPDATE");
    bool rebuild = modeEnv && !std::strcmp(modeEnv, "rebuild");
    int r;
    for (uint32_t i = 0; i < n; i++) RaiseMaxTex(s, 0, instances[i].base_color_tex);
    for (uint32_t i = 0; i < n; i++)      // the reference's static -> dynamic conversion of an instance that starts to move (SceneCore.cpp:1038)
        if (std::memcmp(s->hToWorld.data() + 12 * (size_t)i, instance_to_world + 12 * (size_t)i, 12 * sizeof(float))) { s->movedEver[i] = 1; s->bg.movedSinceBuild = true; }
    std::memcpy(s->hToWorld.data(), instance_to_world, 12 * (size_t)n * sizeof(float));
    const bool rebuildHost = modeEnv && !std::strcmp(modeEnv, "rebuild_host");
    if (rebuild && s->meta.n > BvhBuilder::kTinyScene)
    {
        // ---- a NEW tree on the device (ZR_SCENE_UPDATE=rebuild): LBVH over the moved triangles, ~1 ms for the 380k-triangle atrium against 192 ms
        // for the host's binned-SAH rebuild (ZR_SCENE_UPDATE=rebuild_host); what the reference's per-frame TLAS rebuild is (RtAccelerationStructure.cpp:708-789)
        const size_t nt = s->meta.n;
        HIP_TRY(hipDeviceSynchronize());           // buffers change roles and may be reallocated below: a host-synchronous path
        std::lock_guard<std::mutex> lock(s->mtx);
        // everything the failure path below must put back: a build that fails (allocation, a tree deeper than the traversal stack) must not leave
        // view.nodes / view.tris pointing at buffers that were just demoted to "previous" next to the NEW instance records
        const SceneView viewBefore = s->view;
        const uint32_t numNodesPrevBefore = s->numNodesPrev, numTrisPrevBefore = s->numTrisPrev;
        if ((r = (s->instancesPrev.n == n ? ZR_OK : s->instancesPrev.Alloc(n))) || (s->nodesPrev.n < nt && (r = s->nodesPrev.Alloc(nt))) ||
            (s->trisPrev.n < nt && (r = s->trisPrev.Alloc(nt))) || (s->metaPrev.n != s->meta.n && (r = s->metaPrev.Alloc(s->meta.n))) ||
            (s->toWorld.n != 12 * (size_t)n && (r = s->toWorld.Alloc(12 * (size_t)n)))) return r;
        if (!s->hasPrev && !s->refitReady) HIP_TRY(hipMemcpy(s->metaPrev.p, s->meta.p, s->meta.n * sizeof(TriMeta), hipMemcpyDeviceToDevice));
        std::swap(s->instances.p, s->instancesPrev.p); std::swap(s->instances.n, s->instancesPrev.n);
        std::swap(s->nodes.p, s->nodesPrev.p); std::swap(s->nodes.n, s->nodesPrev.n);
        std::swap(s->tris.p, s->trisPrev.p); std::swap(s->tris.n, s->trisPrev.n);
        std::swap(s->meta.p, s->metaPrev.p); std::swap(s->meta.n, s->metaPrev.n);
        s->numNodesPrev = s->view.numNodes; s->numTrisPrev = s->view.numTris; s->hasPrev = true;
        HIP_TRY(hipMemcpy(s->instances.p, instances, (size_t)n * sizeof(zr_mesh_instance), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(s->toWorld.p, instance_to_world, 12 * (size_t)n * sizeof(float), hipMemcpyHostToDevice));
        s->view.instances = s->instances.p; s->view.triMeta = s->meta.p;
        if ((r = DeviceRebuild(s, st)))
        {
            // roll back: the buffers take their old roles again and the scene renders as before the call.  The "previous" set was used as the build
            // target, so there is no previous structure any more (hasPrev = false: the CtT / temporal passes bind the current one, as in frame 1).
            (void)hipDeviceSynchronize();
            std::swap(s->instances.p, s->instancesPrev.p); std::swap(s->instances.n, s->instancesPrev.n);
            std::swap(s->nodes.p, s->nodesPrev.p); std::swap(s->nodes.n, s->nodesPrev.n);
            std::swap(s->tris.p, s->trisPrev.p); std::swap(s->tris.n, s->trisPrev.n);
            std::swap(s->meta.p, s->metaPrev.p); std::swap(s->meta.n, s->metaPrev.n);
            s->view = viewBefore; s->numNodesPrev = numNodesPrevBefore; s->numTrisPrev = numTrisPrevBefore; s->hasPrev = false;
            return r;
        }
        s->refitReady = false;      // the two buffer sets no longer share a topology
        return ZR_OK;
    }
    if (rebuild || rebuildHost)
    {
        zr_scene_desc d; memset(&d, 0, sizeof(d));
        d.vertices = s->hVertices.data(); d.num_vertices = (uint32_t)s->hVertices.size(); d.indices = s->hIndices.data(); d.num_indices = (uint32_t)s->hIndices.size();
        d.instances = instances; d.num_instances = n; d.instance_to_world = instance_to_world; d.instance_mask = s->hMask.data(); d.instance_num_tris = s->hNumTris.data();
        BvhBuilder builder;
        BuiltBvh bvh = builder.Build(d);
        if (bvh.stackNeed + 1 > (uint32_t)kTravStack) return Fail(ZR_ERR_UNSUPPORTED, "BVH needs %u traversal stack entries (limit %d)", bvh.stackNeed, kTravStack - 1);
        HIP_TRY(hipDeviceSynchronize());           // the host rebuild reallocates: a host-synchronous path by construction (ZR_SCENE_UPDATE=rebuild)
        std::lock_guard<std::mutex> lock(s->mtx);
        std::swap(s->instances.p, s->instancesPrev.p); std::swap(s->instances.n, s->instancesPrev.n);
        std::swap(s->nodes.p, s->nodesPrev.p); std::swap(s->nodes.n, s->nodesPrev.n);
        std::swap(s->tris.p, s->trisPrev.p); std::swap(s->tris.n, s->trisPrev.n);
        std::swap(s->meta.p, s->metaPrev.p); std::swap(s->meta.n, s->metaPrev.n);
        s->numNodesPrev = s->view.numNodes; s->numTrisPrev = s->view.numTris; s->hasPrev = true;
        if ((r = s->instances.Upload(instances, n)) || (r = s->nodes.Upload(bvh.nodes4.data(), bvh.nodes4.size())) ||
            (r = s->tris.Upload(bvh.tris.data(), bvh.tris.size())) || (r = s->meta.Upload(bvh.meta.data(), bvh.meta.size()))) return r;
        SceneView& v = s->view;
        v.instances = s->instances.p; v.nodes = s->nodes.p; v.tris = s->tris.p; v.triMeta = s->meta.p;
        v.numNodes = (uint32_t)bvh.nodes4.size(); v.numTris = (uint32_t)bvh.tris.size();
        if (bvh.maxDepth > s->maxDepth) s->maxDepth = bvh.maxDepth;
        BvhLevels(bvh.nodes4, s->hLevelOrder, s->levelOffsets);
        if ((r = s->levelNodes.Upload(s->hLevelOrder.data(), s->hLevelOrder.size()))) return r;
        s->refitReady = false;      // the two buffer sets no longer share a topology
        return ZR_OK;
    }
    // ---- refit on the device, stream-ordered: nothing below waits on the host (the staging ring aside, when the host runs far ahead)
    // background SAH rebuild: is a finished tree waiting to be installed by this update?
    const bool install = s->bg.enabled && s->bg.state.load(std::memory_order_acquire) == 2 && s->bg.packed && s->bg.inst.size() == n &&
                         s->bg.numTris == s->view.numTris && s->bg.stackNeed + 1 <= (uint32_t)kTravStack;
    if (s->bg.enabled && s->bg.state.load(std::memory_order_acquire) == 2 && !install)
    {   // a tree that cannot be used (the scene changed shape under it, or it is too deep for the traversal stack): drop it
        if (s->bg.th.joinable()) s->bg.th.join();
        s->bg.state.store(0);
    }
    if (install && s->bg.th.joinable()) s->bg.th.join();
    if (s->bg.enabled)
    {   // both buffer sets must be able to hold any topology over these triangles (a BVH4 over nt triangles has < nt nodes): grown once, with their contents
        const size_t cap = s->view.numTris;
        if (s->nodes.n < cap || (s->refitReady && s->nodesPrev.n < cap) || s->levelNodes.n < cap || s->nodeBounds.n < 6 * cap || s->bgDev.n < 5 * cap || !s->dMask.p)
        {
            HIP_TRY(hipDeviceSynchronize());
            if (!s->dMask.p && (r = s->dMask.Upload(s->hMask.data(), s->hMask.size()))) return r;
            if (s->bgDev.n < 5 * cap && (r = s->bgDev.Alloc(5 * cap))) return r;      // 4 child words per node (< cap nodes) + one word per triangle slot
            auto grow = [&](auto& buf, size_t count, size_t keep) -> int {
                if (buf.n >= count) return ZR_OK;
                std::remove_reference_t<decltype(buf)> nb; int rr = nb.Alloc(count); if (rr) return rr;
                if (keep && buf.p) HIP_TRY(hipMemcpy(nb.p, buf.p, keep * sizeof(*buf.p), hipMemcpyDeviceToDevice));
                std::swap(buf.p, nb.p); std::swap(buf.n, nb.n);
                return ZR_OK; };
            std::lock_guard<std::mutex> lockGrow(s->mtx);
            if ((r = grow(s->nodes, cap, s->view.numNodes)) || (s->nodesPrev.p && (r = grow(s->nodesPrev, cap, s->refitReady || s->hasPrev ? s->numNodesPrev : 0))) ||
                (r = grow(s->levelNodes, cap, s->hLevelOrder.size())) || (r = grow(s->nodeBounds, 6 * cap, 0))) return r;
            s->view.nodes = s->nodes.p;
        }
    }
    const size_t nn = s->view.numNodes, nt = s->view.numTris;
    const size_t instBytes = (size_t)n * sizeof(zr_mesh_instance), xfBytes = 12 * (size_t)n * sizeof(float);
    zr_scene::StageSlot* t;
    if ((r = StageAcquire(s, instBytes + xfBytes, &t))) return r;
    memcpy(t->host, instances, instBytes); memcpy((char*)t->host + instBytes, instance_to_world, xfBytes);
    if ((r = SceneWaitUsers(s, st))) return r;         // renders of earlier frames on other streams still read the buffers that change roles below
    std::lock_guard<std::mutex> lock(s->mtx);
    if (!s->refitReady)
    {
        // both buffer sets must hold the same tree: duplicate the current one (once, or after a rebuild)
        const size_t nodeCap = s->bg.enabled ? std::max(nn, nt) : nn;
        if ((s->instancesPrev.n != n && (r = s->instancesPrev.Alloc(n))) || (nodeCap && s->nodesPrev.n < nodeCap && (r = s->nodesPrev.Alloc(nodeCap))) ||
            (s->trisPrev.n != nt && (r = s->trisPrev.Alloc(nt))) || (s->metaPrev.n != s->meta.n && (r = s->metaPrev.Alloc(s->meta.n))) ||
            (nodeCap && s->nodeBounds.n < 6 * nodeCap && (r = s->nodeBounds.Alloc(6 * nodeCap))) || (s->toWorld.n != 12 * (size_t)n && (r = s->toWorld.Alloc(12 * (size_t)n)))) return r;
        HIP_TRY(hipMemcpyAsync(s->instancesPrev.p, s->instances.p, instBytes, hipMemcpyDeviceToDevice, st));
        if (nn) HIP_TRY(hipMemcpyAsync(s->nodesPrev.p, s->nodes.p, nn * sizeof(Bvh4Node), hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemcpyAsync(s->trisPrev.p, s->tris.p, nt * sizeof(BvhTri), hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemcpyAsync(s->metaPrev.p, s->meta.p, s->meta.n * sizeof(TriMeta), hipMemcpyDeviceToDevice, st));
        s->refitReady = true;
    }
    std::swap(s->instances.p, s->instancesPrev.p); std::swap(s->nodes.p, s->nodesPrev.p); std::swap(s->tris.p, s->trisPrev.p); std::swap(s->meta.p, s->metaPrev.p);
    s->numNodesPrev = (uint32_t)nn; s->numTrisPrev = (uint32_t)nt; s->hasPrev = true;
    HIP_TRY(hipMemcpyAsync(s->instances.p, t->host, instBytes, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(s->toWorld.p, (char*)t->host + instBytes, xfBytes, hipMemcpyHostToDevice, st));
    if ((r = StageCommit(t, st))) return r;
    uint32_t numNodesNow = (uint32_t)nn;
    if (install)
    {
        // the background build's topology goes into the set that has just become current (last frame's "previous": nobody needs it any more);
        // its triangles and boxes are then computed for THIS update's transforms by the refit kernels below, like any other frame's
        zr_scene::Background& B = s->bg;
        const size_t devWords = 4 * (size_t)B.numNodes + B.numTris;
        if (!B.pkgCopied) HIP_TRY(hipEventCreateWithFlags(&B.pkgCopied, hipEventDisableTiming));
        HIP_TRY(hipMemcpyAsync(s->bgDev.p, B.pkg, devWords * sizeof(uint32_t), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(s->levelNodes.p, B.pkg + devWords, (size_t)B.numNodes * sizeof(uint32_t), hipMemcpyHostToDevice, st));
        HIP_TRY(hipEventRecord(B.pkgCopied, st)); B.pkgInFlight = true;      // the next build's thread waits for it before it overwrites the package
        hipLaunchKernelGGL(k_install_topology, dim3((uint32_t)((std::max<size_t>(B.numNodes, B.numTris) + 255) / 256)), dim3(256), 0, st,
            s->nodes.p, s->bgDev.p, B.numNodes, s->tris.p, s->bgDev.p + 4 * (size_t)B.numNodes, B.numTris, s->meta.p, s->dMask.p);
        s->hLevelOrder.assign(B.pkg + devWords, B.pkg + devWords + B.numNodes); s->levelOffsets = B.levelOffsets;
        numNodesNow = B.numNodes;
        if (B.maxDepth > s->maxDepth) s->maxDepth = B.maxDepth;
        s->refitReady = false;      // the two sets hold different topologies now: the next update duplicates this one first
        s->deviceBuilt = false;
        B.installed++; B.refitsSince = 0;
        B.state.store(0, std::memory_order_release);
    }
    else if (s->bg.enabled) s->bg.refitsSince++;
    hipLaunchKernelGGL(k_refit_tris, dim3((uint32_t)((nt + 255) / 256)), dim3(256), 0, st, s->tris.p, (uint32_t)nt, s->meta.p, s->instances.p, s->toWorld.p, s->vertices.p, s->indices.p);
    for (size_t l = 0; l + 1 < s->levelOffsets.size(); l++)
    {
        const uint32_t first = s->levelOffsets[l], cnt = s->levelOffsets[l + 1] - first;
        hipLaunchKernelGGL(k_refit_level, dim3((cnt + 63) / 64), dim3(64), 0, st, s->nodes.p, s->levelNodes.p + first, cnt, s->tris.p, s->nodeBounds.p);
    }
    HIP_TRY(hipGetLastError());
    SceneView& v = s->view;
    v.instances = s->instances.p; v.nodes = s->nodes.p; v.tris = s->tris.p; v.triMeta = s->meta.p; v.numNodes = numNodesNow;
    if (!s->updated) HIP_TRY(hipEventCreateWithFlags(&s->updated, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(s->updated, st));
    s->updatedOn = st; s->hasUpdate = true;
    if (s->bg.enabled && s->bg.state.load(std::memory_order_acquire) == 0 && s->bg.refitsSince >= 1 && s->bg.movedSinceBuild && s->meta.n > BvhBuilder::kTinyScene)
    {
        // start the next build on these transforms (host copies: the caller's arrays are only valid during the call)
        zr_scene::Background& B = s->bg;
        B.movedSinceBuild = false;
        if (B.th.joinable()) B.th.join();
        B.inst.assign(instances, instances + n); B.xf.assign(instance_to_world, instance_to_world + 12 * (size_t)n);
        { const char* e = ZR_EXP_ENV("ZR_BVH_GROUP"); if (e && !std::strcmp(e, "0")) B.own.clear(); else B.own = s->movedEver; }
        B.state.store(1, std::memory_order_release); B.started++;
        zr_scene* sp = s;
        B.th = std::thread([sp] {
            zr_scene::Background& Q = sp->bg;
            const auto t0 = std::chrono::steady_clock::now();
            zr_scene_desc d; memset(&d, 0, sizeof(d));
            d.vertices = sp->hVertices.data(); d.num_vertices = (uint32_t)sp->hVertices.size(); d.indices = sp->hIndices.data(); d.num_indices = (uint32_t)sp->hIndices.size();
            d.instances = Q.inst.data(); d.num_instances = (uint32_t)Q.inst.size(); d.instance_to_world = Q.xf.data(); d.instance_mask = sp->hMask.data(); d.instance_num_tris = sp->hNumTris.data();
            BvhBuilder builder;
            BuiltBvh bvh = builder.Build(d, Q.own.empty() ? nullptr : Q.own.data());
            std::vector<uint32_t> levelOrder;
            BvhLevels(bvh.nodes4, levelOrder, Q.levelOffsets);
            const auto t1 = std::chrono::steady_clock::now();
            // pack the topology into pinned memory (this thread's time, not the installing update's)
            const size_t nn = bvh.nodes4.size(), nt = bvh.tris.size(), words = 5 * nn + nt;
            Q.packed = false;
            if (hipSetDevice(sp->device) == hipSuccess)
            {
                if (Q.pkgInFlight) { (void)hipEventSynchronize(Q.pkgCopied); Q.pkgInFlight = false; }      // the previous package's copies have left the buffer
                if (Q.pkgCap < words)
                {
                    if (Q.pkg) { (void)hipHostFree(Q.pkg); Q.pkg = nullptr; Q.pkgCap = 0; }
                    void* h = nullptr;
                    if (hipHostMalloc(&h, (words + words / 8) * sizeof(uint32_t), hipHostMallocDefault) == hipSuccess) { Q.pkg = (uint32_t*)h; Q.pkgCap = words + words / 8; }
                    else (void)hipGetLastError();
                }
                if (Q.pkg)
                {
                    uint32_t* w = Q.pkg;
                    for (size_t i = 0; i < nn; i++) for (int c = 0; c < 4; c++) *w++ = bvh.nodes4[i].child[c];
                    for (size_t i = 0; i < nt; i++) *w++ = bvh.tris[i].gidx;
                    std::memcpy(w, levelOrder.data(), nn * sizeof(uint32_t));
                    Q.packed = true;
                }
            }
            Q.numNodes = (uint32_t)nn; Q.numTris = (uint32_t)nt; Q.stackNeed = bvh.stackNeed; Q.maxDepth = bvh.maxDepth;
            Q.buildMs = std::chrono::duration<double, std::milli>(t1 - t0).count();
            Q.packMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
            if (ZR_EXP_ENV("ZR_BVH_TIMING"))
                std::fprintf(stderr, "[zr_scene] background tree: %zu nodes over %zu triangles, build %.1f ms, package (%.1f MB pinned) %.1f ms\n", nn, nt, Q.buildMs, words * 4e-6, Q.packMs);
            Q.state.store(2, std::memory_order_release);
        });
    }
    return ZR_OK;
}
// A tree built for where the instances are NOW, in the background, swapped in by a later zr_scene_update_instances(_async) (see zr_scene::Background).
int zr_scene_set_background_rebuild(zr_scene* s, int enable)
{
    if (!s) return Fail(ZR_ERR_INVALID_ARG, "zr_scene_set_background_rebuild: null scene");
    std::lock_guard<std::mutex> lock(s->mtx);
    s->bg.enabled = enable != 0;
    return ZR_OK;
}
int zr_scene_background_rebuild_stats(zr_scene* s, uint64_t* started, uint64_t* installed, int* building)
{
    if (!s) return Fail(ZR_ERR_INVALID_ARG, "zr_scene_background_rebuild_stats: null scene");
    if (started) *started = s->bg.started.load(std::memory_order_relaxed);
    if (installed) *installed = s->bg.installed.load(std::memory_order_relaxed);
    if (building) *building = s->bg.state.load(std::memory_order_acquire);
    return ZR_OK;
}
// host-synchronous form: the update on the null stream, then wait for the copies and the refit kernels (their errors surface here)
int zr_scene_update_instances(zr_scene* s, const zr_mesh_instance* instances, const float* instance_to_world, uint32_t n)
{
    int r = zr_scene_update_instances_async(s, nullptr, instances, instance_to_world, n);
    if (r) return r;
    HIP_TRY(hipDeviceSynchronize());
    return ZR_OK;
}

int zr_wire_layout(char* buf, size_t cap)
{
    std::string s;
    char line[160];
#define ZR_L_SIZE(name, T) do { snprintf(line, sizeof(line), name " %zu\n", sizeof(T)); s += line; } while (0)
#define ZR_L_FIELD(name, T, f) do { snprintf(line, sizeof(line), name " %zu %zu\n", offsetof(T, f), sizeof(((T*)0)->f)); s += line; } while (0)
    ZR_L_SIZE("Vertex", zr_vertex); ZR_L_FIELD("Vertex.Position", zr_vertex, pos); ZR_L_FIELD("Vertex.TexUV", zr_vertex, uv);
    ZR_L_FIELD("Vertex.Normal", zr_vertex, normal); ZR_L_FIELD("Vertex.Tangent", zr_vertex, tangent);
    ZR_L_SIZE("MeshInstance", zr_mesh_instance);
    ZR_L_FIELD("MeshInstance.BaseVtxOffset", zr_mesh_instance, base_vtx_offset); ZR_L_FIELD("MeshInstance.BaseIdxOffset", zr_mesh_instance, base_idx_offset);
    ZR_L_FIELD("MeshInstance.Rotation", zr_mesh_instance, rotation); ZR_L_FIELD("MeshInstance.Scale", zr_mesh_instance, scale);
    ZR_L_FIELD("MeshInstance.MatIdx", zr_mesh_instance, mat_idx); ZR_L_FIELD("MeshInstance.BaseEmissiveTriOffset", zr_mesh_instance, base_emissive_tri_offset);
    ZR_L_FIELD("MeshInstance.Translation", zr_mesh_instance, translation); ZR_L_FIELD("MeshInstance.PrevRotation", zr_mesh_instance, prev_rotation);
    ZR_L_FIELD("MeshInstance.PrevScale", zr_mesh_instance, prev_scale); ZR_L_FIELD("MeshInstance.dTranslation", zr_mesh_instance, d_translation);
    ZR_L_FIELD("MeshInstance.BaseColorTex", zr_mesh_instance, base_color_tex); ZR_L_FIELD("MeshInstance.AlphaFactor_Cutoff", zr_mesh_instance, alpha_factor_cutoff);
    ZR_L_SIZE("EmissiveTriangle", zr_emissive_triangle);
    ZR_L_FIELD("EmissiveTriangle.Vtx0", zr_emissive_triangle, vtx0); ZR_L_FIELD("EmissiveTriangle.V0V1", zr_emissive_triangle, v0v1);
    ZR_L_FIELD("EmissiveTriangle.V0V2", zr_emissive_triangle, v0v2); ZR_L_FIELD("EmissiveTriangle.EdgeLengths", zr_emissive_triangle, edge_lengths);
    ZR_L_FIELD("EmissiveTriangle.ID", zr_emissive_triangle, id); ZR_L_FIELD("EmissiveTriangle.PackedA", zr_emissive_triangle, packed_a);
    ZR_L_FIELD("EmissiveTriangle.PackedB", zr_emissive_triangle, packed_b); ZR_L_FIELD("EmissiveTriangle.UV0", zr_emissive_triangle, uv0);
    ZR_L_FIELD("EmissiveTriangle.UV1", zr_emissive_triangle, uv1); ZR_L_FIELD("EmissiveTriangle.UV2", zr_emissive_triangle, uv2);
    ZR_L_SIZE("EmissiveLumenAliasTableEntry", zr_alias_entry);
    ZR_L_FIELD("EmissiveLumenAliasTableEntry.CachedP_Orig", zr_alias_entry, cached_p_orig); ZR_L_FIELD("EmissiveLumenAliasTableEntry.CachedP_Alias", zr_alias_entry, cached_p_alias);
    ZR_L_FIELD("EmissiveLumenAliasTableEntry.P_Curr", zr_alias_entry, p_curr); ZR_L_FIELD("EmissiveLumenAliasTableEntry.Alias", zr_alias_entry, alias);
    ZR_L_SIZE("PresampledEmissiveTriangle", zr_presampled_tri);
    ZR_L_FIELD("PresampledEmissiveTriangle.pos", zr_presampled_tri, pos); ZR_L_FIELD("PresampledEmissiveTriangle.normal", zr_presampled_tri, normal);
    ZR_L_FIELD("PresampledEmissiveTriangle.pdf", zr_presampled_tri, pdf); ZR_L_FIELD("PresampledEmissiveTriangle.ID", zr_presampled_tri, id);
    ZR_L_FIELD("PresampledEmissiveTriangle.idx", zr_presampled_tri, idx); ZR_L_FIELD("PresampledEmissiveTriangle.bary", zr_presampled_tri, bary);
    ZR_L_FIELD("PresampledEmissiveTriangle.le", zr_presampled_tri, le); ZR_L_FIELD("PresampledEmissiveTriangle.twoSided", zr_presampled_tri, two_sided);
    ZR_L_SIZE("VoxelSample", zr_voxel_sample);
    ZR_L_FIELD("VoxelSample.pos", zr_voxel_sample, pos); ZR_L_FIELD("VoxelSample.normal", zr_voxel_sample, normal); ZR_L_FIELD("VoxelSample.pdf", zr_voxel_sample, pdf);
    ZR_L_FIELD("VoxelSample.ID", zr_voxel_sample, id); ZR_L_FIELD("VoxelSample.le", zr_voxel_sample, le); ZR_L_FIELD("VoxelSample.twoSided", zr_voxel_sample, two_sided);
    ZR_L_SIZE("Material", zr_material);
    ZR_L_FIELD("Material.BaseColorFactor", zr_material, base_color_factor); ZR_L_FIELD("Material.BaseColorTex_Subsurf_CoatWeight", zr_material, base_color_tex_subsurf_coat_weight);
    ZR_L_FIELD("Material.NormalTex_TrDepth", zr_material, normal_tex_tr_depth); ZR_L_FIELD("Material.MRTex_SpecRoughness_CoatRoughness", zr_material, mr_tex_spec_roughness_coat_roughness);
    ZR_L_FIELD("Material.EmissiveFactor_NormalScale", zr_material, emissive_factor_normal_scale); ZR_L_FIELD("Material.EmissiveStrength_IOR", zr_material, emissive_strength_ior);
    ZR_L_FIELD("Material.EmissiveTex_AlphaCutoff_CoatIOR", zr_material, emissive_tex_alpha_cutoff_coat_ior); ZR_L_FIELD("Material.CoatColor_Flags", zr_material, coat_color_flags);
    ZR_L_SIZE("cbFrameConstants", zr_frame_constants);
#define ZR_L_CB(ref, f) ZR_L_FIELD("cbFrameConstants." ref, zr_frame_constants, f)
    ZR_L_CB("CurrView", curr_view); ZR_L_CB("PrevView", prev_view); ZR_L_CB("CurrViewInv", curr_view_inv); ZR_L_CB("PrevViewInv", prev_view_inv);
    ZR_L_CB("CurrViewProj", curr_view_proj); ZR_L_CB("PrevViewProj", prev_view_proj); ZR_L_CB("CameraPos", camera_pos); ZR_L_CB("CameraNear", camera_near);
    ZR_L_CB("AspectRatio", aspect_ratio); ZR_L_CB("PixelSpreadAngle", pixel_spread_angle); ZR_L_CB("TanHalfFOV", tan_half_fov); ZR_L_CB("dt", dt);
    ZR_L_CB("FrameNum", frame_num); ZR_L_CB("CurrGBufferDescHeapOffset", curr_gbuffer_desc_heap_offset); ZR_L_CB("PrevGBufferDescHeapOffset", prev_gbuffer_desc_heap_offset);
    ZR_L_CB("BaseColorMapsDescHeapOffset", base_color_maps_desc_heap_offset); ZR_L_CB("NormalMapsDescHeapOffset", normal_maps_desc_heap_offset);
    ZR_L_CB("MetallicRoughnessMapsDescHeapOffset", metallic_roughness_maps_desc_heap_offset); ZR_L_CB("EmissiveMapsDescHeapOffset", emissive_maps_desc_heap_offset);
    ZR_L_CB("EnvMapDescHeapOffset", env_map_desc_heap_offset); ZR_L_CB("RenderWidth", render_width); ZR_L_CB("RenderHeight", render_height);
    ZR_L_CB("DisplayWidth", display_width); ZR_L_CB("DisplayHeight", display_height); ZR_L_CB("CurrCameraJitter", curr_camera_jitter); ZR_L_CB("PrevCameraJitter", prev_camera_jitter);
    ZR_L_CB("PlanetRadius", planet_radius); ZR_L_CB("SunCosAngularRadius", sun_cos_angular_radius); ZR_L_CB("SunSinAngularRadius", sun_sin_angular_radius); ZR_L_CB("pad", pad);
    ZR_L_CB("SunDir", sun_dir); ZR_L_CB("SunIlluminance", sun_illuminance); ZR_L_CB("RayleighSigmaSColor", rayleigh_sigma_s_color); ZR_L_CB("RayleighSigmaSScale", rayleigh_sigma_s_scale);
    ZR_L_CB("OzoneSigmaAColor", ozone_sigma_a_color); ZR_L_CB("OzoneSigmaAScale", ozone_sigma_a_scale); ZR_L_CB("MieSigmaS", mie_sigma_s); ZR_L_CB("MieSigmaA", mie_sigma_a);
    ZR_L_CB("AtmosphereAltitude", atmosphere_altitude); ZR_L_CB("g", g); ZR_L_CB("NumFramesCameraStatic", num_frames_camera_static); ZR_L_CB("CameraStatic", camera_static);
    ZR_L_CB("Accumulate", accumulate); ZR_L_CB("SunMoved", sun_moved); ZR_L_CB("CameraRayUVGradsScale", camera_ray_uv_grads_scale); ZR_L_CB("MipBias", mip_bias);
    ZR_L_CB("OneDivNumEmissiveTriangles", one_div_num_emissive_triangles); ZR_L_CB("NumEmissiveTriangles", num_emissive_triangles); ZR_L_CB("FocusDepth", focus_depth);
    ZR_L_CB("LensRadius", lens_radius); ZR_L_CB("DoF", dof); ZR_L_CB("pad2", pad2);
#undef ZR_L_CB
#undef ZR_L_FIELD
#undef ZR_L_SIZE
    if (!buf || s.size() + 1 > cap) return -(int)(s.size() + 1);
    memcpy(buf, s.c_str(), s.size() + 1);
    return (int)s.size();
}

int zr_scene_destroy(zr_scene* s) { delete s; return ZR_OK; }

int zr_scene_set_alias_table_async(zr_scene* s, void* stream, const zr_alias_entry* e, uint32_t n)
{
    if (!s || !e) return Fail(ZR_ERR_INVALID_ARG, "zr_scene_set_alias_table: null argument");
    if (n != s->view.numEmissives) return Fail(ZR_ERR_INVALID_ARG, "alias table size %u != number of emissive triangles %u", n, s->view.numEmissives);
    HIP_TRY(hipSetDevice(s->device));
    hipStream_t st = (hipStream_t)stream;
    int r; zr_scene::StageSlot* t;
    if (s->alias.n != n && (r = s->alias.Alloc(n))) return r;      // (re)allocation only when the light count changed: hipFree waits for the device
    if ((r = StageAcquire(s, (size_t)n * sizeof(zr_alias_entry), &t))) return r;
    memcpy(t->host, e, (size_t)n * sizeof(zr_alias_entry));
    if ((r = SceneWaitUsers(s, st))) return r;          // renders of earlier frames on other streams may still draw from the old table
    HIP_TRY(hipMemcpyAsync(s->alias.p, t->host, (size_t)n * sizeof(zr_alias_entry), hipMemcpyHostToDevice, st));
    if ((r = StageCommit(t, st))) return r;
    {
        std::lock_guard<std::mutex> lock(s->mtx);
        s->aliasHost.assign(e, e + n);
        s->view.alias = s->alias.p;
    }
    return SceneMarkUpdated(s, st);
}
int zr_scene_set_alias_table(zr_scene* s, const zr_alias_entry* e, uint32_t n)
{
    int r = zr_scene_set_alias_table_async(s, nullptr, e, n);
    if (r) return r;
    HIP_TRY(hipDeviceSynchronize());
    return ZR_OK;
}

int zr_scene_get_alias_table(const zr_scene* s, zr_alias_entry* out, uint32_t n)
{
    if (!s || !out) return Fail(ZR_ERR_INVALID_ARG, "null argument");
    if (n != s->aliasHost.size()) return Fail(ZR_ERR_INVALID_ARG, "alias table has %zu entries", s->aliasHost.size());
    memcpy(out, s->aliasHost.data(), n * sizeof(zr_alias_entry));
    return ZR_OK;
}

int zr_scene_get_light_voxel_grid(const zr_scene* s, void* stream, zr_voxel_sample* out, uint32_t n)
{
    if (!s || !out) return Fail(ZR_ERR_INVALID_ARG, "null argument");
    if (!s->view.lvg) return Fail(ZR_ERR_NOT_INITIALIZED, "no light voxel grid: render the PRELIGHTING pass with use_lvg first");
    if (n != s->lvg.n) return Fail(ZR_ERR_INVALID_ARG, "the light voxel grid has %zu samples", s->lvg.n);
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    HIP_TRY(hipMemcpy(out, s->lvg.p, (size_t)n * sizeof(zr_voxel_sample), hipMemcpyDeviceToHost));
    return ZR_OK;
}

int zr_scene_get_presampled_sets(const zr_scene* s, void* stream, zr_presampled_tri* out, uint32_t n)
{
    if (!s || !out) return Fail(ZR_ERR_INVALID_ARG, "null argument");
    if (!s->view.sampleSets) return Fail(ZR_ERR_NOT_INITIALIZED, "no presampled light sets: render the PRELIGHTING pass with presampling first");
    if (n != s->sampleSets.n) return Fail(ZR_ERR_INVALID_ARG, "the presampled sets hold %zu samples", s->sampleSets.n);
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    HIP_TRY(hipMemcpy(out, s->sampleSets.p, (size_t)n * sizeof(zr_presampled_tri), hipMemcpyDeviceToHost));
    return ZR_OK;
}

int zr_scene_bvh_info(const zr_scene* s, uint32_t* num_nodes, uint32_t* num_tris, uint32_t* max_depth)
{
    if (!s) return Fail(ZR_ERR_INVALID_ARG, "null scene");
    if (num_nodes) *num_nodes = s->view.numNodes;
    if (num_tris) *num_tris = s->view.numTris;
    if (max_depth) *max_depth = s->maxDepth;
    return ZR_OK;
}

int zr_gbuffer_create(int device, uint32_t w, uint32_t h, zr_gbuffer** out)
{
    if (!out || !w || !h) return Fail(ZR_ERR_INVALID_ARG, "zr_gbuffer_create: bad argument");
    int r = RequireDevice(device);
    if (r) return r;
    zr_gbuffer* g = new (std::nothrow) zr_gbuffer();
    if (!g) return Fail(ZR_ERR_OOM, "out of host memory");
    g->device = device; g->w = w; g->h = h;
    for (int k = 0; k < 2; k++)
    for (int i = 0; i < ZR_GB_COUNT; i++)
    {
        if ((r = g->planeSets[k][i].Alloc((size_t)w * h * ZR_GB_PLANE_BYTES[i]))) { delete g; return r; }
        hipError_t e = hipMemset(g->planeSets[k][i].p, 0, g->planeSets[k][i].n);
        if (e != hipSuccess) { delete g; return Fail(ZR_ERR_HIP, "hipMemset failed: %s", hipGetErrorString(e)); }
    }
    { hipError_t e = hipDeviceSynchronize(); if (e != hipSuccess) { delete g; return Fail(ZR_ERR_HIP, "hipDeviceSynchronize failed: %s", hipGetErrorString(e)); } }   // see zr_pass_init
    *out = g;
    return ZR_OK;
}
int zr_gbuffer_destroy(zr_gbuffer* g) { delete g; return ZR_OK; }
int zr_gbuffer_set_tile_origin(zr_gbuffer* g, uint32_t x0, uint32_t y0)
{
    if (!g) return Fail(ZR_ERR_INVALID_ARG, "null gbuffer");
    if ((x0 & 31u) || (y0 & 31u)) return Fail(ZR_ERR_INVALID_ARG, "tile origin must be 32-pixel aligned");
    g->x0 = x0; g->y0 = y0;
    return ZR_OK;
}

int zr_gbuffer_download(const zr_gbuffer* g, void* stream, zr_gbuffer_planes* hp)
{
    if (!g || !hp) return Fail(ZR_ERR_INVALID_ARG, "null argument");
    if (hp->width != g->w || hp->height != g->h) return Fail(ZR_ERR_INVALID_ARG, "plane size mismatch");
    HIP_TRY(hipSetDevice(g->device));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    for (int i = 0; i < ZR_GB_COUNT; i++)
        if (hp->plane[i]) HIP_TRY(hipMemcpy(hp->plane[i], g->Planes()[i].p, g->Planes()[i].n, hipMemcpyDeviceToHost));
    return ZR_OK;
}
int zr_gbuffer_device_plane(const zr_gbuffer* g, int plane, void** dev)
{
    if (!g || !dev || plane < 0 || plane >= ZR_GB_COUNT) return Fail(ZR_ERR_INVALID_ARG, "bad argument");
    *dev = g->Planes()[plane].p;
    return ZR_OK;
}

int zr_pass_create(int kind, int device, zr_pass** out)
{
    if (!out) return Fail(ZR_ERR_INVALID_ARG, "null out");
    if (kind < ZR_PASS_GBUFFER || kind > ZR_PASS_DENOISE) return Fail(ZR_ERR_INVALID_ARG, "unknown pass kind %d", kind);
    int r = RequireDevice(device);
    if (r) return r;
    zr_pass* p = new (std::nothrow) zr_pass();
    if (!p) return Fail(ZR_ERR_OOM, "out of host memory");
    p->kind = kind; p->device = device;
    zr_params_default(&p->params);
    if (kind == ZR_PASS_DI_SKY)
    {
        // SkyDI.cpp:81-82, SkyDI.h:86-92: M_max (Sky) 15 -> m_max_temporal, M_max (Sun) 3 -> m_max_spatial, Alpha_min = 0.35^2
        p->params.flags = ZR_IND_TEMPORAL_RESAMPLE | ZR_IND_SPATIAL_RESAMPLE;
        p->params.m_max_temporal = 15; p->params.m_max_spatial = 3; p->params.alpha_min = 0.35f * 0.35f;
    }
    if (kind == ZR_PASS_DI_EMISSIVE)
    {
        // DirectLighting.cpp:100-107, DirectLighting.h:93-98
        p->params.flags = ZR_IND_TEMPORAL_RESAMPLE | ZR_IND_SPATIAL_RESAMPLE | ZR_DI_STOCHASTIC_SPATIAL | ZR_DI_EXTRA_DISOCCLUSION_SAMPLING;
        p->params.m_max_temporal = 20; p->params.m_max_spatial = 20; p->params.alpha_min = 0.05f * 0.05f;
    }
    *out = p;
    return ZR_OK;
}

// the planes frame overlap adds to a ReSTIR PT pass (zr_pass_set_frame_overlap): a third reservoir set, a second target plane, a second FINAL plane
static int AllocOverlapPlanes(zr_pass* p)
{
    const size_t cap = (size_t)p->w * p->h;
    int r;
    zr_pass::ResStorage& R = p->res[2];
    if ((r = R.Alloc(cap)) || (r = p->rptTargetAlt.Alloc(cap)) || (r = p->finalAlt.Alloc(cap * 4)) || (r = p->finalAlt2.Alloc(cap * 4))) return r;
    HIP_TRY(hipMemset(R.A.p, 0, cap * 4)); HIP_TRY(hipMemset(R.B.p, 0, cap * 8)); HIP_TRY(hipMemset(R.C.p, 0, cap * 16));
    HIP_TRY(hipMemset(R.D.p, 0, cap * 16)); HIP_TRY(hipMemset(R.E.p, 0, cap * 2)); HIP_TRY(hipMemset(R.F.p, 0, cap * 8)); HIP_TRY(hipMemset(R.G.p, 0, cap * 8));
    HIP_TRY(hipMemset(p->rptTargetAlt.p, 0, cap * 16)); HIP_TRY(hipMemset(p->finalAlt.p, 0, cap * 4 * sizeof(float))); HIP_TRY(hipMemset(p->finalAlt2.p, 0, cap * 4 * sizeof(float)));
    return ZR_OK;
}
static int AllocPass(zr_pass* p)
{
    int r;
    if (p->kind == ZR_PASS_SKY) { if ((r = p->skyLut.Alloc((size_t)p->w * p->h))) return r; }
    if (p->kind == ZR_PASS_TAA)
    {
        const size_t n = (size_t)p->w * p->h * 4;
        for (int k = 0; k < 2; k++) { if ((r = p->taaOut[k].Alloc(n))) return r; HIP_TRY(hipMemset(p->taaOut[k].p, 0, n * sizeof(uint16_t))); }
        p->taaIdx = 0; p->temporalValid = false;
    }
    if (p->kind == ZR_PASS_DENOISE)
    {
        const size_t n = (size_t)p->w * p->h;
        if ((r = p->svgfHist.Alloc(n)) || (r = p->svgfAccum.Alloc(n)) || (r = p->svgfPing.Alloc(n)) || (r = p->svgfPong.Alloc(n)) || (r = p->svgfGuide.Alloc(n)) || (r = p->svgfGuideFw.Alloc(n))) return r;
        for (int k = 0; k < 2; k++) { if ((r = p->svgfMoments[k].Alloc(2 * n))) return r; HIP_TRY(hipMemset(p->svgfMoments[k].p, 0, 2 * n * sizeof(float))); }
        HIP_TRY(hipMemset(p->svgfHist.p, 0, n * sizeof(F4))); HIP_TRY(hipMemset(p->svgfPing.p, 0, n * sizeof(F4))); HIP_TRY(hipMemset(p->svgfPong.p, 0, n * sizeof(F4)));
        p->svgfMomIdx = 0; p->svgfOut = p->svgfPing.p; p->svgfCur = p->svgfPing.p; p->temporalValid = false;
    }
    if (p->kind == ZR_PASS_AUTO_EXPOSURE)
    {
        if ((r = p->aeHist.Alloc(post::kHistBins)) || (r = p->aeExposure.Alloc(2))) return r;
        HIP_TRY(hipMemset(p->aeHist.p, 0, post::kHistBins * sizeof(uint32_t)));
        HIP_TRY(hipMemset(p->aeExposure.p, 0, 2 * sizeof(float)));      // TEXTURE_FLAGS::INIT_TO_ZERO, AutoExposure.cpp:150-155
    }
    if (p->kind == ZR_PASS_DISPLAY)
    {
        const size_t cap = (size_t)p->w * p->h;
        if ((r = p->displayOut.Alloc(cap)) || (r = p->displaySrgb.Alloc(cap))) return r;
        HIP_TRY(hipMemset(p->displayOut.p, 0, cap * sizeof(F4))); HIP_TRY(hipMemset(p->displaySrgb.p, 0, cap * 4));
    }
    if (p->kind == ZR_PASS_COMPOSITING)
    {
        const size_t cap = (size_t)p->w * p->h;
        if ((r = p->firstBOP.Alloc(cap))) return r;            // scratch plane of the firefly filter (composited, unfiltered)
        HIP_TRY(hipMemset(p->firstBOP.p, 0, cap * sizeof(F4)));
        if ((r = p->finalRGBA.Alloc(cap * 4))) return r;
        HIP_TRY(hipMemset(p->finalRGBA.p, 0, cap * 4 * sizeof(float)));
    }
    if (p->kind == ZR_PASS_DI_SKY)
    {
        const size_t cap = (size_t)p->w * p->h;
        if ((r = p->finalRGBA.Alloc(cap * 4))) return r;
        if ((r = p->counters.Alloc(2 * kCounterSlots))) return r;
        HIP_TRY(hipMemset(p->finalRGBA.p, 0, cap * 4 * sizeof(float)));
        HIP_TRY(hipMemset(p->counters.p, 0, 2 * kCounterSlots * sizeof(unsigned long long)));
        for (int k = 0; k < 2; k++)
        {
            if ((r = p->skyA[k].Alloc(cap)) || (r = p->skyB[k].Alloc(2 * cap)) || (r = p->skyC[k].Alloc(2 * cap))) return r;
            HIP_TRY(hipMemset(p->skyA[k].p, 0, cap)); HIP_TRY(hipMemset(p->skyB[k].p, 0, cap * 4)); HIP_TRY(hipMemset(p->skyC[k].p, 0, cap * 8));
        }
        if ((r = p->diTarget.Alloc(cap))) return r;
        HIP_TRY(hipMemset(p->diTarget.p, 0, cap * 16));
        p->temporalValid = false; p->currIdx = 0;
    }
    if (p->kind == ZR_PASS_DI_EMISSIVE)
    {
        const size_t cap = (size_t)p->w * p->h;
        if ((r = p->finalRGBA.Alloc(cap * 4))) return r;
        if ((r = p->counters.Alloc(2 * kCounterSlots))) return r;
        HIP_TRY(hipMemset(p->finalRGBA.p, 0, cap * 4 * sizeof(float)));
        HIP_TRY(hipMemset(p->counters.p, 0, 2 * kCounterSlots * sizeof(unsigned long long)));
        for (int k = 0; k < 2; k++)
        {
            if ((r = p->diA[k].Alloc(cap)) || (r = p->diB[k].Alloc(2 * cap))) return r;
            HIP_TRY(hipMemset(p->diA[k].p, 0, cap * 16)); HIP_TRY(hipMemset(p->diB[k].p, 0, cap * 8));
        }
        if ((r = p->diTarget.Alloc(cap))) return r;
        HIP_TRY(hipMemset(p->diTarget.p, 0, cap * 16));
        if ((r = p->diSampleSet.Upload(kRdiSampleSet, 64))) return r;
        p->temporalValid = false; p->currIdx = 0;
    }
    if (p->kind == ZR_PASS_INDIRECT)
    {
        const size_t cap = (size_t)p->w * p->h;
        if ((r = p->q[0].Alloc(cap))) return r;
        if ((r = p->q[1].Alloc(cap))) return r;
        if ((r = p->finalRGBA.Alloc(cap * 4))) return r;
        if ((r = p->firstBOP.Alloc(cap))) return r;
        if ((r = p->counts.Alloc(5 * (kMaxRounds + 2) * kCounterStride))) return r;   // per round: live paths, k_trace cursor, 3 ray counts
        if ((r = p->counters.Alloc(2 * kCounterSlots))) return r;
        if ((r = p->groupMax.Alloc((size_t)kMaxRounds * ((p->w + 7) / 8) * ((p->h + 7) / 8)))) return r;
        HIP_TRY(hipMemset(p->finalRGBA.p, 0, cap * 4 * sizeof(float)));
        HIP_TRY(hipMemset(p->counters.p, 0, 2 * kCounterSlots * sizeof(unsigned long long)));
        if (p->integrator == ZR_INTEGRATOR_RESTIR_GI)
        {
            for (int k = 0; k < 2; k++)
            {
                if ((r = p->giA[k].Alloc(cap)) || (r = p->giB[k].Alloc(4 * cap)) || (r = p->giC[k].Alloc(cap))) return r;
                HIP_TRY(hipMemset(p->giA[k].p, 0, cap * 16)); HIP_TRY(hipMemset(p->giB[k].p, 0, cap * 8)); HIP_TRY(hipMemset(p->giC[k].p, 0, cap * 16));
            }
            p->temporalValid = false; p->currIdx = 0;
        }
        if (p->integrator == ZR_INTEGRATOR_RESTIR_PT)
        {
            for (int k = 0; k < 2; k++)
            {
                if ((r = p->res[k].Alloc(cap))) return r;
                if ((r = p->rb[k].Alloc(cap))) return r;
                zr_pass::ResStorage& R = p->res[k];
                HIP_TRY(hipMemset(R.A.p, 0, cap * 4)); HIP_TRY(hipMemset(R.B.p, 0, cap * 8)); HIP_TRY(hipMemset(R.C.p, 0, cap * 16));
                HIP_TRY(hipMemset(R.D.p, 0, cap * 16)); HIP_TRY(hipMemset(R.E.p, 0, cap * 2)); HIP_TRY(hipMemset(R.F.p, 0, cap * 8));
                HIP_TRY(hipMemset(R.G.p, 0, cap * 8));
                zr_pass::RBufStorage& B = p->rb[k];
                HIP_TRY(hipMemset(B.A.p, 0, cap * 8)); HIP_TRY(hipMemset(B.B.p, 0, cap * 16)); HIP_TRY(hipMemset(B.C.p, 0, cap * 16)); HIP_TRY(hipMemset(B.D.p, 0, cap * 2));
            }
            if ((r = p->rptTarget.Alloc(cap))) return r;
            if ((r = p->rptNeighbor.Alloc(2 * cap))) return r;
            HIP_TRY(hipMemset(p->rptTarget.p, 0, cap * 16)); HIP_TRY(hipMemset(p->rptNeighbor.p, 0, cap * 2));
            for (auto& m : p->rptMap) { if ((r = m.Alloc(cap))) return r; HIP_TRY(hipMemset(m.p, 0, cap * 2)); }
            { const size_t cells = (size_t)((p->w + 31u) / 32u + 1u) * ((p->h + 31u) / 32u + 1u); if ((r = p->costMap.Alloc(cells))) return r; HIP_TRY(hipMemset(p->costMap.p, 0, cells * 4)); }
            if ((r = p->rptSampleSet.Upload(kRptSampleSet, 1024))) return r;
            if ((r = p->rptLists.Alloc(4 * cap))) return r;
            p->rptSet[0] = 0; p->rptSet[1] = 1; p->rptSet[2] = 2; p->tgtIdx = 0; p->finIdx = 0; p->finOut = 0; p->frameOpen = false; p->haveCand = false; p->haveTemporal = false; p->haveDone[0] = p->haveDone[1] = false;
            if (p->overlap) { if ((r = AllocOverlapPlanes(p))) return r; }
            // word layout (kRptListWords): [0, 1] temporal counts, [2, 3] first spatial round, [4, 5] the temporal replays' cursors (counts + 4 / + 5 of
            // base 0: the only DYNAMIC replay), [6, 7] second spatial round, [8 .. 11] = base 6's cursor slots -- unused (the spatial replays split
            // their lists statically) but allocated and zeroed, so that no base ever reaches past the buffer
            if ((r = p->rptListCounts.Alloc(kRptListWords))) return r;
            p->temporalValid = false; p->currIdx = 0;
        }
    }
    return ZR_OK;
}

int zr_pass_init(zr_pass* p, uint32_t w, uint32_t h, int integrator)
{
    if (!p || !w || !h) return Fail(ZR_ERR_INVALID_ARG, "zr_pass_init: bad argument");
    if (p->kind == ZR_PASS_INDIRECT && (integrator < ZR_INTEGRATOR_PATH_TRACING || integrator > ZR_INTEGRATOR_RESTIR_PT))
        return Fail(ZR_ERR_INVALID_ARG, "unknown integrator %d", integrator);
    HIP_TRY(hipSetDevice(p->device));
    p->w = w; p->h = h; p->integrator = integrator;
    int r = AllocPass(p);
    if (r) return r;
    // the clears above are null-stream work and asynchronous to the host; a first render on a non-blocking stream must not overtake them
    HIP_TRY(hipDeviceSynchronize());
    p->initialized = true;
    return ZR_OK;
}
int zr_pass_resize(zr_pass* p, uint32_t w, uint32_t h)
{
    if (!p) return Fail(ZR_ERR_INVALID_ARG, "null pass");
    if (!p->initialized) return Fail(ZR_ERR_NOT_INITIALIZED, "pass not initialised");
    return zr_pass_init(p, w, h, p->integrator);
}
int zr_pass_reset_temporal(zr_pass* p)
{
    if (!p) return Fail(ZR_ERR_INVALID_ARG, "null pass");
    if (!p->initialized) return Fail(ZR_ERR_NOT_INITIALIZED, "pass not initialised");
    HIP_TRY(hipSetDevice(p->device));
    if (p->kind == ZR_PASS_INDIRECT) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(hipMemset(p->finalRGBA.p, 0, p->finalRGBA.n * sizeof(float))); if (p->finalAlt.p) { HIP_TRY(hipMemset(p->finalAlt.p, 0, p->finalAlt.n * sizeof(float))); HIP_TRY(hipMemset(p->finalAlt2.p, 0, p->finalAlt2.n * sizeof(float))); } }
    p->temporalValid = false;       // IndirectLighting::ResetTemporal -> RESET_TEMPORAL_TEXTURES next frame
    if (p->kind == ZR_PASS_DI_EMISSIVE || p->kind == ZR_PASS_DI_SKY) { HIP_TRY(hipMemset(p->finalRGBA.p, 0, p->finalRGBA.n * sizeof(float))); p->currIdx = 0; }   // DirectLighting.cpp:159-164, SkyDI.cpp:128-133
    HIP_TRY(hipDeviceSynchronize());      // a host call between frames: renders on non-blocking streams must see the cleared plane
    return ZR_OK;
}
int zr_pass_set_params(zr_pass* p, const zr_params* prm)
{
    if (!p || !prm) return Fail(ZR_ERR_INVALID_ARG, "null argument");
    if (p->kind == ZR_PASS_DI_EMISSIVE && (prm->m_max_temporal < 1 || prm->m_max_temporal > 30)) return Fail(ZR_ERR_INVALID_ARG, "DI M_max must be in 1..30");
    if (p->kind == ZR_PASS_DI_SKY && (prm->m_max_temporal < 1 || prm->m_max_temporal > 15 || prm->m_max_spatial < 1 || prm->m_max_spatial > 15))
        return Fail(ZR_ERR_INVALID_ARG, "sky DI M_max (sky = m_max_temporal, sun = m_max_spatial) must be in 1..15");
    if (prm->use_lvg)
    {
        const uint32_t dx = prm->lvg_grid_dim & 1023u, dy = (prm->lvg_grid_dim >> 10) & 1023u, dz = (prm->lvg_grid_dim >> 20) & 1023u;
        if (!prm->presampling) return Fail(ZR_ERR_INVALID_ARG, "the light voxel grid needs light presampling (IndirectLighting.h:93)");
        if (!dx || !dy || !dz || !(prm->lvg_extents[0] > 0) || !(prm->lvg_extents[1] > 0) || !(prm->lvg_extents[2] > 0))
            return Fail(ZR_ERR_INVALID_ARG, "light voxel grid: dimension / extents must be positive");
    }
    if (prm->max_non_tr_bounces < 1 || prm->max_non_tr_bounces > 15 || prm->max_glossy_tr_bounces < 1 || prm->max_glossy_tr_bounces > 15)
        return Fail(ZR_ERR_INVALID_ARG, "bounce counts must be in 1..15");
    // the reservoir caps travel in 4-bit fields of cb_ReSTIR_PT_*::Packed (IndirectLighting.cpp:1260-1270: "M_max (Temporal)" 1..15, "M_max (Spatial)" 1..12)
    if (p->kind == ZR_PASS_INDIRECT && (prm->m_max_temporal < 1 || prm->m_max_temporal > 15 || prm->m_max_spatial < 1 || prm->m_max_spatial > 15))
        return Fail(ZR_ERR_INVALID_ARG, "indirect lighting: M_max (temporal %u, spatial %u) must be in 1..15", prm->m_max_temporal, prm->m_max_spatial);
    if (p->kind == ZR_PASS_DENOISE && (prm->svgf_iterations > 8u || prm->svgf_normal_power_log2 > 16u))
        return Fail(ZR_ERR_INVALID_ARG, "DENOISE: svgf_iterations must be <= 8, svgf_normal_power_log2 <= 16");
    if (p->kind == ZR_PASS_INDIRECT && prm->num_spatial_passes > 2u) return Fail(ZR_ERR_INVALID_ARG, "ReSTIR PT: num_spatial_passes must be 0, 1 or 2 (IndirectLighting.cpp:1238-1240)");
    if (prm->presampling && (prm->num_sample_sets == 0 || prm->sample_set_size == 0 || prm->num_sample_sets > 65535 || prm->sample_set_size > 65535))
        return Fail(ZR_ERR_INVALID_ARG, "presampling needs 1..65535 sample sets of 1..65535 samples");
    if (prm->tex_filter >= ZR_TEX_FILTER_COUNT) return Fail(ZR_ERR_INVALID_ARG, "tex_filter must be a ZR_TEX_FILTER_* value");
    p->params = *prm;
    return ZR_OK;
}

static int RenderGBuffer(zr_pass* p, hipStream_t s, const zr_frame_constants* cb, const zr_scene* sc, zr_gbuffer* gb)
{
    if (!gb) return Fail(ZR_ERR_INVALID_ARG, "GBUFFER pass needs a gbuffer");
    if (gb->x0 + gb->w > cb->render_width || gb->y0 + gb->h > cb->render_height) return Fail(ZR_ERR_INVALID_ARG, "gbuffer tile lies outside the render target of the frame constants");
    const uint32_t tilesX = (gb->w + 15) / 16, tilesY = (gb->h + 15) / 16;
    if (int wr = GBufferAcquireWrite(gb, gb->Next(), s)) return wr;      // (tracked G-buffers: the set about to be overwritten may still be read on another stream)
    gb->cur = gb->Next(); gb->numRendered++;
    gb->plainAt[gb->cur] = sc->plainMaterials.load(std::memory_order_relaxed); gb->sceneAt[gb->cur] = sc->uid;
    TimerBegin(p, s, "gbuffer");
    uint32_t pickXY = 0xffffffffu;
    if (p->pickXY != 0xffffffffu)
    {
        // a pending pick: only a G-buffer (tile) that holds the pixel writes
        if (!p->pickBuf.p) { int r = p->pickBuf.Alloc(1); if (r) return r; }
        const uint32_t px = p->pickXY & 0xffffu, py = p->pickXY >> 16;
        if (px >= gb->x0 && px < gb->x0 + gb->w && py >= gb->y0 && py < gb->y0 + gb->h) { pickXY = p->pickXY; p->pickWritten = true; }
    }
    hipLaunchKernelGGL(k_gbuffer, dim3(tilesX * tilesY), dim3(kBlock), 0, s, FrameView(sc, cb), *cb, gb->View(), tilesX, pickXY, p->pickBuf.p);
    TimerEnd(p, s);
    HIP_TRY(hipGetLastError());
    p->hostCounters.n_closest += (uint64_t)gb->w * gb->h;
    return GBufferMarkWritten(gb, s);
}

// GBufferRT::PickPixel / ClearPick / GetPickReadbackBuffer (GBufferRT.h:36-46): every GBUFFER render while a pick is pending writes the mesh index under
// the pixel (GBufferRT_Inline.hlsl:241-242: hitMeshIdx, UINT32_MAX on a miss)
int zr_pass_pick_pixel(zr_pass* p, uint32_t x, uint32_t y)
{
    if (!p || p->kind != ZR_PASS_GBUFFER) return Fail(ZR_ERR_INVALID_ARG, "zr_pass_pick_pixel: needs a GBUFFER pass");
    if (x >= 0xffffu || y >= 0xffffu) return Fail(ZR_ERR_INVALID_ARG, "zr_pass_pick_pixel: pixel (%u, %u) out of range", x, y);      // (UINT16_MAX is the reference's "no pick")
    p->pickXY = x | (y << 16); p->pickWritten = false;
    return ZR_OK;
}
int zr_pass_clear_pick(zr_pass* p)
{
    if (!p || p->kind != ZR_PASS_GBUFFER) return Fail(ZR_ERR_INVALID_ARG, "zr_pass_clear_pick: needs a GBUFFER pass");
    p->pickXY = 0xffffffffu;
    return ZR_OK;
}
int zr_pass_read_pick(zr_pass* p, void* stream, uint32_t* mesh_idx)
{
    if (!p || p->kind != ZR_PASS_GBUFFER || !mesh_idx) return Fail(ZR_ERR_INVALID_ARG, "zr_pass_read_pick: needs a GBUFFER pass and an output");
    if (!p->pickWritten) return Fail(ZR_ERR_NOT_INITIALIZED, "zr_pass_read_pick: no G-buffer has been rendered over the picked pixel since zr_pass_pick_pixel");
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipMemcpyAsync(mesh_idx, p->pickBuf.p, sizeof(uint32_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return ZR_OK;
}

// Sky::Render (Sky.cpp:120-164): K17, then bind the LUT to the scene
static int RenderSky(zr_pass* p, hipStream_t s, const zr_frame_constants* cb, zr_scene* sc)
{
    TimerBegin(p, s, "sky_view_lut");
    hipLaunchKernelGGL(k_sky_lut, dim3((p->w + 7) / 8, (p->h + 7) / 8), dim3(64), 0, s, *cb, p->w, p->h, p->skyLut.p);
    TimerEnd(p, s);
    HIP_TRY(hipGetLastError());
    { std::lock_guard<std::mutex> lock(sc->mtx); sc->view.sky.data = p->skyLut.p; sc->view.sky.w = p->w; sc->view.sky.h = p->h; }
    return ZR_OK;
}

// K3: PreLighting::Render presampling branch (PreLighting.cpp:369-411): every frame, seeded by FrameNum
static int RenderPresample(zr_pass* p, hipStream_t s, const zr_frame_constants* cb, zr_scene* sc)
{
    const uint32_t total = p->params.num_sample_sets * p->params.sample_set_size;
    if (sc->sampleSets.n != total) { int r = sc->sampleSets.Alloc(total); if (r) return r; }
    {
        std::lock_guard<std::mutex> lock(sc->mtx);
        sc->numSampleSets = p->params.num_sample_sets;
        sc->view.sampleSets = sc->sampleSets.p; sc->view.sampleSetSize = p->params.sample_set_size;
    }
    TimerBegin(p, s, "presample_emissives");
    hipLaunchKernelGGL(k_presample, dim3((total + 63) / 64), dim3(64), 0, s, FrameView(sc, cb), total, cb->frame_num, sc->view.numEmissives, sc->sampleSets.p);
    TimerEnd(p, s);
    HIP_TRY(hipGetLastError());
    return ZR_OK;
}

// K4: PreLighting::Render light-voxel-grid branch (PreLighting.cpp:405-428): every frame (the grid follows the camera)
static int RenderLVG(zr_pass* p, hipStream_t s, const zr_frame_constants* cb, zr_scene* sc)
{
    const zr_params& ip = p->params;
    const uint32_t dx = ip.lvg_grid_dim & 1023u, dy = (ip.lvg_grid_dim >> 10) & 1023u, dz = (ip.lvg_grid_dim >> 20) & 1023u;
    const size_t total = (size_t)dx * dy * dz * ZR_LVG_SAMPLES_PER_VOXEL;
    if (sc->lvg.n != total) { int r = sc->lvg.Alloc(total); if (r) return r; }
    TimerBegin(p, s, "build_lvg");
    hipLaunchKernelGGL(k_build_lvg, dim3(dx, dy, dz), dim3(64), 0, s, FrameView(sc, cb), *cb, dx, dy, dz, ip.lvg_extents[0], ip.lvg_extents[1], ip.lvg_extents[2],
        ip.lvg_offset_y, sc->lvg.p);
    TimerEnd(p, s);
    HIP_TRY(hipGetLastError());
    std::lock_guard<std::mutex> lock(sc->mtx);
    sc->view.lvg = sc->lvg.p; sc->view.lvgDim[0] = dx; sc->view.lvgDim[1] = dy; sc->view.lvgDim[2] = dz;
    for (int a = 0; a < 3; a++) sc->view.lvgExtents[a] = ip.lvg_extents[a];
    sc->view.lvgOffsetY = ip.lvg_offset_y;
    return ZR_OK;
}

static int RenderPreLightingInner(zr_pass* p, hipStream_t s, const zr_frame_constants* cb, zr_scene* sc);
static int RenderPreLighting(zr_pass* p, hipStream_t s, const zr_frame_constants* cb, zr_scene* sc)
{
    int r = RenderPreLightingInner(p, s, cb, sc);
    if (r || !p->params.use_lvg || sc->view.numEmissives == 0) return r;
    return RenderLVG(p, s, cb, sc);
}
static int RenderPreLightingInner(zr_pass* p, hipStream_t s, const zr_frame_constants* cb, zr_scene* sc)
{
    const uint32_t n = sc->view.numEmissives;
    if (n == 0) return ZR_OK;
    // the alias table is built once per emissive set (EmissiveTriangleAliasTable is only re-run when emissive materials change):
    // zr_scene_invalidate_alias_table or a new scene forces a rebuild
    if (sc->view.alias && sc->aliasStale)
    {
        // deferred rebuild: (1) nothing in flight -> estimate the powers and start their read-back; (2) read-back finished -> build on the host and
        // enqueue the upload; either way this call only enqueues, and the bound table stays valid for this frame's sampling
        int r;
        if (!p->powerPending)
        {
            if (p->power.n != n && (r = p->power.Alloc(n))) return r;
            if (p->powerHostCap < n) { if (p->powerHost) (void)hipHostFree(p->powerHost); p->powerHost = nullptr; HIP_TRY(hipHostMalloc((void**)&p->powerHost, n * sizeof(float), hipHostMallocDefault)); p->powerHostCap = n; }
            if (!p->powerEv) HIP_TRY(hipEventCreateWithFlags(&p->powerEv, hipEventDisableTiming));
            hipLaunchKernelGGL(k_estimate_power, dim3((n + 255) / 256), dim3(256), 0, s, FrameView(sc, cb), n, p->power.p);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(p->powerHost, p->power.p, n * sizeof(float), hipMemcpyDeviceToHost, s));
            HIP_TRY(hipEventRecord(p->powerEv, s));
            p->powerPending = true;
        }
        else if (hipEventQuery(p->powerEv) == hipSuccess)
        {
            std::vector<float> power(p->powerHost, p->powerHost + n);
            std::vector<zr_alias_entry> table(n);
            BuildAliasTableHost(power, table.data(), 0);
            if ((r = zr_scene_set_alias_table_async(sc, s, table.data(), n))) return r;
            p->powerPending = false;
            std::lock_guard<std::mutex> lock(sc->mtx);
            sc->aliasStale = false;
        }
        return p->params.presampling ? RenderPresample(p, s, cb, sc) : ZR_OK;
    }
    if (sc->view.alias) return p->params.presampling ? RenderPresample(p, s, cb, sc) : ZR_OK;
    int r = p->power.Alloc(n);
    if (r) return r;
    TimerBegin(p, s, "estimate_power");
    hipLaunchKernelGGL(k_estimate_power, dim3((n + 255) / 256), dim3(256), 0, s, FrameView(sc, cb), n, p->power.p);
    TimerEnd(p, s);
    HIP_TRY(hipGetLastError());
    // EmissiveTriangleAliasTable::Render (PreLighting.cpp:512-585): read back, build on the host, upload
    std::vector<float> power(n);
    HIP_TRY(hipMemcpyAsync(power.data(), p->power.p, n * sizeof(float), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    std::vector<zr_alias_entry> table(n);
    BuildAliasTableHost(power, table.data(), 0);
    if ((r = zr_scene_set_alias_table_async(sc, s, table.data(), n))) return r;      // the upload is ordered on the pass stream, ahead of the sampling kernels
    return p->params.presampling ? RenderPresample(p, s, cb, sc) : ZR_OK;
}

// The rect of the G-buffer tile this pass shades (zr_pass_set_owned_rect; width 0 = the whole tile), validated once for every
// pass with cross-pixel reuse: inside the tile (planes are indexed relative to the tile origin), 32-px aligned origin (thread
// groups / RNG group ids coincide with the single-device run) and -- on a proper sub-rect of the frame -- an apron of >= 16 px
// wherever the frame continues (spatial / temporal-candidate search radius: 15 px ReSTIR PT, 16 px DI / sky DI / GI).
static int ResolveOwnedRect(const zr_pass* p, const zr_gbuffer* gb, const zr_frame_constants* cb, uint32_t* ox0, uint32_t* oy0, uint32_t* ow, uint32_t* oh)
{
    const uint32_t x0 = p->own[2] ? p->own[0] : gb->x0, y0 = p->own[2] ? p->own[1] : gb->y0;
    const uint32_t w = p->own[2] ? p->own[2] : gb->w, h = p->own[2] ? p->own[3] : gb->h;
    if (w == 0 || h == 0) return Fail(ZR_ERR_INVALID_ARG, "owned rect is empty");
    if (x0 < gb->x0 || y0 < gb->y0 || (uint64_t)x0 + w > (uint64_t)gb->x0 + gb->w || (uint64_t)y0 + h > (uint64_t)gb->y0 + gb->h)
        return Fail(ZR_ERR_INVALID_ARG, "owned rect lies outside the G-buffer tile");
    if ((x0 & 31u) || (y0 & 31u)) return Fail(ZR_ERR_INVALID_ARG, "owned rect origin must be 32-pixel aligned");
    const bool wholeFrame = x0 == 0 && y0 == 0 && w == cb->render_width && h == cb->render_height;
    if (!wholeFrame)
    {
        auto apronOk = [&](uint32_t lo, uint32_t olo, uint32_t hi, uint32_t ohi, uint32_t frame) {
            return (olo == 0 || olo - lo >= 16) && (ohi == frame || hi - ohi >= 16); };
        if (!apronOk(gb->x0, x0, gb->x0 + gb->w, x0 + w, cb->render_width) || !apronOk(gb->y0, y0, gb->y0 + gb->h, y0 + h, cb->render_height))
            return Fail(ZR_ERR_INVALID_ARG, "a pass with cross-pixel reuse on a screen tile needs a G-buffer apron of >= 16 px around the owned rect (zr_pass_set_owned_rect)");
    }
    *ox0 = x0; *oy0 = y0; *ow = w; *oh = h;
    return ZR_OK;
}

// DirectLighting::Render (DirectLighting.cpp:166-296)
static int RenderDirectEmissive(zr_pass* p, hipStream_t s, const zr_frame_constants* cb, const zr_scene* sc, zr_gbuffer* gb, int stages)
{
    using namespace rdi;
    if (!gb) return Fail(ZR_ERR_INVALID_ARG, "DI_EMISSIVE pass needs a gbuffer");
    if (gb->w != p->w || gb->h != p->h) return Fail(ZR_ERR_INVALID_ARG, "gbuffer / pass size mismatch");
    if (gb->x0 + gb->w > cb->render_width || gb->y0 + gb->h > cb->render_height) return Fail(ZR_ERR_INVALID_ARG, "gbuffer tile outside the render target");
    if (sc->view.numEmissives == 0) return Fail(ZR_ERR_INVALID_ARG, "DI_EMISSIVE needs emissive triangles");
    if (!sc->view.alias) return Fail(ZR_ERR_NOT_INITIALIZED, "emissive alias table missing: render the PRELIGHTING pass first");
    if (cb->num_emissive_triangles != sc->view.numEmissives) return Fail(ZR_ERR_INVALID_ARG, "cbFrameConstants.NumEmissiveTriangles != scene");
    const zr_params& ip = p->params;
    if (ip.presampling && (!sc->view.sampleSets || sc->numSampleSets != ip.num_sample_sets || sc->view.sampleSetSize != ip.sample_set_size))
        return Fail(ZR_ERR_NOT_INITIALIZED, "presampled light sets missing or of another size: render the PRELIGHTING pass with the same presampling params first");
    DiFrame F;
    F.sc = FrameView(sc, cb); F.gb = gb->View(); F.gbPrev = gb->PrevView();
    F.scPrev = FrameViewPrev(sc, cb);
    if (int orc = ResolveOwnedRect(p, gb, cb, &F.ox0, &F.oy0, &F.ow, &F.oh)) return orc;
    F.cur.A = p->diA[p->currIdx].p; F.cur.B = p->diB[p->currIdx].p; F.prev.A = p->diA[1 - p->currIdx].p; F.prev.B = p->diB[1 - p->currIdx].p;
    F.target = p->diTarget.p; F.finalRGBA = p->finalRGBA.p; F.sampleSet = p->diSampleSet.p;
    DiParams& prm = F.prm;
    prm.flags = ip.flags; prm.M_max = ip.m_max_temporal; prm.numSampleSets = ip.presampling ? ip.num_sample_sets : 0u;
    prm.accumulate = (cb->accumulate && cb->camera_static) ? 1u : 0u;
    prm.doTemporal = (p->temporalValid && (ip.flags & ZR_IND_TEMPORAL_RESAMPLE) && gb->numRendered >= 2) ? 1u : 0u;
    prm.doSpatial = (prm.doTemporal && (ip.flags & ZR_IND_SPATIAL_RESAMPLE)) ? 1u : 0u;
    prm.writeReservoirs = (prm.doTemporal || !p->temporalValid) ? 1u : 0u;
    const bool hvs = (ip.flags & ZR_DI_HALF_VECTOR_COPY_SHIFT) != 0;      // USE_HALF_VECTOR_COPY_SHIFT (ReSTIR_DI/Params.hlsli:12): its own kernel instantiations
    prm.halfVec = hvs ? 1u : 0u; prm.alpha_min = ip.alpha_min;
    const uint32_t tilesX = (F.ow + 15) / 16, tilesY = (F.oh + 15) / 16;
    const dim3 grid(tilesX * tilesY), block(kBlock);
    const bool plainDi = PlainClass(sc, gb);
    if (stages & ZR_STAGE_TEMPORAL)
    {
        TimerBegin(p, s, "rdi_temporal");
        hipLaunchKernelGGL(hvs ? (plainDi ? k_rdi_temporal<true, true> : k_rdi_temporal<false, true>) : (plainDi ? k_rdi_temporal<true, false> : k_rdi_temporal<false, false>), dim3(grid.x * (256 / kDiBlock)), dim3(kDiBlock), 0, s, F, *cb, tilesX, p->counters.p + 2 * 8);
        TimerEnd(p, s);
    }
    if (!(stages & ZR_STAGE_SPATIAL)) { HIP_TRY(hipGetLastError()); return ZR_OK; }
    if (prm.doSpatial)
    {
        TimerBegin(p, s, "rdi_spatial");
        hipLaunchKernelGGL(hvs ? (plainDi ? k_rdi_spatial<true, true> : k_rdi_spatial<false, true>) : (plainDi ? k_rdi_spatial<true, false> : k_rdi_spatial<false, false>), dim3(grid.x * (256 / kDiBlock)), dim3(kDiBlock), 0, s, F, *cb, tilesX, p->counters.p + 2 * 9);
        TimerEnd(p, s);
    }
    HIP_TRY(hipGetLastError());
    p->temporalValid = true;
    p->currIdx = 1 - p->currIdx;
    return ZR_OK;
}

// SkyDI::Render (SkyDI.cpp:135-259)
static int RenderDirectSky(zr_pass* p, hipStream_t s, const zr_frame_constants* cb, const zr_scene* sc, zr_gbuffer* gb, int stages)
{
    using namespace sdi;
    if (!gb) return Fail(ZR_ERR_INVALID_ARG, "DI_SKY pass needs a gbuffer");
    if (gb->w != p->w || gb->h != p->h) return Fail(ZR_ERR_INVALID_ARG, "gbuffer / pass size mismatch");
    if (gb->x0 + gb->w > cb->render_width || gb->y0 + gb->h > cb->render_height) return Fail(ZR_ERR_INVALID_ARG, "gbuffer tile outside the render target");
    if (!sc->view.sky.data) return Fail(ZR_ERR_NOT_INITIALIZED, "sky-view LUT missing: render a ZR_PASS_SKY pass first");
    const zr_params& ip = p->params;
    SkyFrame F;
    F.sc = FrameView(sc, cb); F.gb = gb->View(); F.gbPrev = gb->PrevView();
    F.scPrev = FrameViewPrev(sc, cb);
    if (int orc = ResolveOwnedRect(p, gb, cb, &F.ox0, &F.oy0, &F.ow, &F.oh)) return orc;
    F.cur.A = p->skyA[p->currIdx].p; F.cur.B = p->skyB[p->currIdx].p; F.cur.C = p->skyC[p->currIdx].p;
    F.prev.A = p->skyA[1 - p->currIdx].p; F.prev.B = p->skyB[1 - p->currIdx].p; F.prev.C = p->skyC[1 - p->currIdx].p;
    F.target = p->diTarget.p; F.finalRGBA = p->finalRGBA.p;
    SkyParams& prm = F.prm;
    prm.M_max_sky = ip.m_max_temporal; prm.M_max_sun = ip.m_max_spatial; prm.alpha_min = ip.alpha_min;
    prm.accumulate = (cb->accumulate && cb->camera_static) ? 1u : 0u;
    prm.doTemporal = (p->temporalValid && (ip.flags & ZR_IND_TEMPORAL_RESAMPLE) && gb->numRendered >= 2) ? 1u : 0u;
    prm.doSpatial = (prm.doTemporal && (ip.flags & ZR_IND_SPATIAL_RESAMPLE)) ? 1u : 0u;
    prm.writeReservoirs = (prm.doTemporal || !p->temporalValid) ? 1u : 0u;
    const uint32_t tilesX = (F.ow + 15) / 16, tilesY = (F.oh + 15) / 16;
    const dim3 grid(tilesX * tilesY), block(kBlock);
    const bool plainDi = PlainClass(sc, gb);
    if (stages & ZR_STAGE_TEMPORAL)
    {
        TimerBegin(p, s, "sdi_temporal");
        hipLaunchKernelGGL(plainDi ? k_sdi_temporal<true> : k_sdi_temporal<false>, dim3(grid.x * (256 / kSdiBlock)), dim3(kSdiBlock), 0, s, F, *cb, tilesX, p->counters.p + 2 * 11);
        TimerEnd(p, s);
    }
    if (!(stages & ZR_STAGE_SPATIAL)) { HIP_TRY(hipGetLastError()); return ZR_OK; }
    if (prm.doSpatial)
    {
        TimerBegin(p, s, "sdi_spatial");
        hipLaunchKernelGGL(plainDi ? k_sdi_spatial<true> : k_sdi_spatial<false>, dim3(grid.x * (256 / kSdiBlock)), dim3(kSdiBlock), 0, s, F, *cb, tilesX, p->counters.p + 2 * 12);
        TimerEnd(p, s);
    }
    HIP_TRY(hipGetLastError());
    p->temporalValid = true;
    p->currIdx = 1 - p->currIdx;
    return ZR_OK;
}

// IndirectLighting::RenderReSTIR_GI (IndirectLighting.cpp:277-368) + the Render() tail (:1006-1025)
static int RenderReSTIR_GI(zr_pass* p, hipStream_t s, const zr_frame_constants* cb, const zr_scene* sc, zr_gbuffer* gb)
{
    using namespace rgi;
    const zr_params& ip = p->params;
    GiFrame F;
    F.sc = FrameView(sc, cb); F.gb = gb->View(); F.gbPrev = gb->PrevView();
    F.sc.texFilter = ip.tex_filter;
    if (int orc = ResolveOwnedRect(p, gb, cb, &F.ox0, &F.oy0, &F.ow, &F.oh)) return orc;
    F.cur.A = p->giA[p->currIdx].p; F.cur.B = p->giB[p->currIdx].p; F.cur.C = p->giC[p->currIdx].p;
    F.prev.A = p->giA[1 - p->currIdx].p; F.prev.B = p->giB[1 - p->currIdx].p; F.prev.C = p->giC[1 - p->currIdx].p;
    F.finalRGBA = p->finalRGBA.p;
    GiParams& prm = F.prm;
    prm.flags = ip.flags; prm.maxNonTrBounces = ip.max_non_tr_bounces; prm.maxGlossyTrBounces = ip.max_glossy_tr_bounces;
    prm.numSampleSets = ip.presampling ? ip.num_sample_sets : 0u;
    prm.accumulate = (cb->accumulate && cb->camera_static) ? 1u : 0u;
    prm.doTemporal = ((ip.flags & ZR_IND_TEMPORAL_RESAMPLE) && p->temporalValid && gb->numRendered >= 2) ? 1u : 0u;
    prm.writeReservoirs = (prm.doTemporal || !p->temporalValid) ? 1u : 0u;
    prm.M_max = (float)ip.m_max_temporal;
    prm.useLVG = (ip.use_lvg && ip.presampling) ? 1u : 0u;
    prm.textured = sc->view.tex.count ? 1u : 0u;
    if (prm.useLVG && !sc->view.lvg) return Fail(ZR_ERR_NOT_INITIALIZED, "light voxel grid missing: render the PRELIGHTING pass with use_lvg first");
    const uint32_t tilesX = (F.ow + 15) / 16, tilesY = (F.oh + 15) / 16;
    TimerBegin(p, s, "rgi");
    hipLaunchKernelGGL(sc->view.tex.count ? k_rgi_tex : PlainClass(sc, gb) ? k_rgi<true> : k_rgi<false>, dim3(tilesX * tilesY * (256 / kRgiBlock)), dim3(kRgiBlock), 0, s, F, *cb, tilesX, p->counters.p + 2 * 10);
    TimerEnd(p, s);
    HIP_TRY(hipGetLastError());
    p->temporalValid = true;
    p->currIdx = 1 - p->currIdx;
    return ZR_OK;
}

// K11 is launched as one-wave blocks, and a wave lives for the whole path: a grid of a few thousand waves runs in ROUNDS of as many waves as are
// resident at once -- 3072 for the 3-wave build, 4096 for the 4-wave build.  A 480 x 544 tile of the 8-way screen split is 4080 waves: one round and
// a third of a second one at 3 waves per SIMD, exactly one at 4.  Measured per tile of the Cornell frame (scripts/gpu_r03_rounds.sh [rounds 1-4: git history up to 31e92fa; today scripts/gpu.sh ab]): the busiest
// 8-way tiles 0.365 -> 0.31 ms with the 4-wave build (slowest tile of the split 0.82 -> 0.76 ms); with two rounds or more the per-wave cost of the
// 4-wave build (the kernel is VALU-bound: a wave shares its SIMD with one more) eats the saving -- the 4-way split's tiles (8160 waves) get 4 - 17 %
// slower, the 2-way split's do not care -- so only the one-round case switches.
// frame overlap, carry mode: every plane of reservoir set `src` (+ its target plane) into `dst`, pixel by pixel (62 + 16 B read and written per pixel)
__global__ void __launch_bounds__(256) k_rpt_carry(rpt::ResPlanes dst, rpt::ResPlanes src, F4* tdst, const F4* tsrc, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    {
        dst.A[i] = src.A[i]; dst.B[2 * i] = src.B[2 * i]; dst.B[2 * i + 1] = src.B[2 * i + 1]; dst.C[i] = src.C[i]; dst.D[i] = src.D[i]; dst.E[i] = src.E[i];
        dst.F[2 * i] = src.F[2 * i]; dst.F[2 * i + 1] = src.F[2 * i + 1]; dst.G[2 * i] = src.G[2 * i]; dst.G[2 * i + 1] = src.G[2 * i + 1];
        tdst[i] = tsrc[i];
    }
}
static bool FewerRoundsAtFourWaves(uint32_t waves)
{
    static const bool off = [] { const char* e = ZR_EXP_ENV("ZR_K11_ROUNDS"); return e && !strcmp(e, "0"); }();      // (A/B switch)
    return !off && waves > 3072u && waves <= 4096u;
}

static int RenderReSTIR_PT(zr_pass* p, hipStream_t s, const zr_frame_constants* cb, const zr_scene* sc, zr_gbuffer* gb, int stages)
{
    using namespace rpt;
    // the TEMPORAL stage in its two halves (zetaray_amd.h): K11 alone / K12 - K14
    const zr_pass::ResStorage* carryFrom = nullptr;
    const bool stageCand = (stages & (ZR_STAGE_TEMPORAL | ZR_STAGE_CANDIDATES)) != 0, stageReuseT = (stages & (ZR_STAGE_TEMPORAL | ZR_STAGE_TEMPORAL_REUSE)) != 0;
    if (stageCand && p->overlap)
    {
        // two frames in flight need the G-buffer's third plane set and its stream tracking: both belong to the G-buffer handed to zr_pass_set_frame_overlap (a renderer
        // that replaces its G-buffer -- a resize -- hands the new one over again)
        if (!gb->tracked || gb->numSets < 3) return Fail(ZR_ERR_INVALID_ARG, "ReSTIR PT with frame overlap: this G-buffer is not the one given to zr_pass_set_frame_overlap (call it again with the new G-buffer)");
        if (p->frameOpen) return Fail(ZR_ERR_INVALID_ARG, "ReSTIR PT with frame overlap: the previous frame's last stage has not been enqueued (stage order: CANDIDATES, TEMPORAL_REUSE, SPATIAL[, SPATIAL2])");
        // Frame overlap: this frame's K11 runs beside the previous frame's reuse passes, which still read the two sets in play and that frame's target
        // plane -- so it writes the free set (which takes the "current" role; the set it replaces is free from here on) and the other target / FINAL
        // plane, behind the previous frame's temporal reuse: the last kernel that read what is recycled here (K16 of the frame before it precedes that
        // K14 on its stream) and the last reader of the G-buffer set this frame's GBUFFER render overwrote.
        std::swap(p->rptSet[p->currIdx], p->rptSet[2]);
        p->tgtIdx ^= 1;
        carryFrom = p->overlapCarry ? &p->res[p->rptSet[2]] : nullptr;
        if (!(cb->accumulate && cb->camera_static)) p->finIdx = (p->finIdx + 1) % 3;      // (an accumulating frame adds to the plane the frames before it wrote)
        if (p->overlapCarry)
        {   // carry mode copies the replaced set, whose last writer is the previous frame's K14: one frame of overlap (beside that frame's spatial stage)
            if (p->haveTemporal && p->reuseStream != s) HIP_TRY(hipStreamWaitEvent(s, p->evTemporal, 0));
        }
        else
        {   // product mode: what this stage recycles -- the free reservoir set and this parity's target plane -- was last touched by the frame BEFORE the
            // previous one (its spatial stage read them), and the G-buffer has a third plane set: two frames of overlap, the first half of frame N + 2 also
            // runs beside the temporal reuse of frame N + 1
            const int k = (int)(p->ovFrame & 1);
            if (p->haveDone[k] && p->doneStream[k] != s) HIP_TRY(hipStreamWaitEvent(s, p->evDone[k], 0));
        }
    }
    RptFrame F;
    F.sc = FrameView(sc, cb); F.gb = gb->View(); F.gbPrev = gb->PrevView();
    F.scPrev = FrameViewPrev(sc, cb);
    F.sc.texFilter = F.scPrev.texFilter = p->params.tex_filter;
    if (int orc = ResolveOwnedRect(p, gb, cb, &F.ox0, &F.oy0, &F.ow, &F.oh)) return orc;
    F.rbCtN = p->rb[0].View(); F.rbNtC = p->rb[1].View(); F.tex.target = p->Target(); F.tex.neighbor = p->rptNeighbor.p;
    F.finalRGBA = p->Final(p->finIdx); F.sampleSet = p->rptSampleSet.p;
    F.mapCtN = p->rptMap[0].p; F.mapNtC = p->rptMap[1].p;
    F.costMap = p->costOn ? p->costMap.p : nullptr; F.costW = (p->w + 31u) / 32u + 1u; F.costMode = p->costRays ? 1u : 0u;
    F.trip = nullptr; F.tripStats = nullptr; F.tripStride = 0;
    F.carryOut = nullptr; F.carryIn = nullptr; F.carryCount = nullptr; F.carryCap = 0; F.carryBounce = 0;
    // K12 sorts whole 32 x 32 tiles: an owned rect may end inside one only where the render target ends
    if (((F.ox0 + F.ow) & 31u) && F.ox0 + F.ow != cb->render_width) return Fail(ZR_ERR_INVALID_ARG, "ReSTIR PT: the owned rect must end on a 32-pixel boundary or at the right edge of the render target");
    if (((F.oy0 + F.oh) & 31u) && F.oy0 + F.oh != cb->render_height) return Fail(ZR_ERR_INVALID_ARG, "ReSTIR PT: the owned rect must end on a 32-pixel boundary or at the bottom edge of the render target");
    RptParams& prm = F.prm;
    const zr_params& ip = p->params;
    const bool havePrevGBuffer = gb->numRendered >= 2;
    prm.maxNonTrBounces = ip.max_non_tr_bounces; prm.maxGlossyTrBounces = ip.max_glossy_tr_bounces;
    prm.russianRoulette = (ip.flags & ZR_IND_RUSSIAN_ROULETTE) ? 1u : 0u;
    prm.numSampleSets = ip.presampling ? ip.num_sample_sets : 0u;
    prm.accumulate = (cb->accumulate && cb->camera_static) ? 1u : 0u;
    prm.boiling = (ip.flags & ZR_IND_BOILING_SUPPRESSION) ? 1u : 0u;
    prm.M_max_temporal = ip.m_max_temporal & 0xf; prm.M_max_spatial = ip.m_max_spatial & 0xf; prm.alpha_min = ip.alpha_min;
    prm.emissive = cb->num_emissive_triangles ? 1u : 0u;
    prm.textured = sc->view.tex.count ? 1u : 0u;
    prm.sortTemporal = (ip.flags & ZR_IND_SORT_TEMPORAL) ? 1u : 0u; prm.sortSpatial = (ip.flags & ZR_IND_SORT_SPATIAL) ? 1u : 0u;
    // the CtN map (current reservoirs bucketed by k) schedules the fused CtT + TtC kernel: 0.540 -> 0.495 ms Cornell, 3.27 -> 3.16 ms atrium at
    // 1080p; the NtC map does not pay (0.546 / 3.34).  ZR_TEMPORAL_MAP = 0 / 1 / 2 overrides (scripts/gpu_sortmap.sh [rounds 1-4: git history up to 31e92fa; today scripts/gpu.sh ab])
    static const uint32_t temporalMapEnv = [] { const char* e = ZR_EXP_ENV("ZR_TEMPORAL_MAP"); return e ? (uint32_t)atoi(e) : 1u; }();
    prm.temporalMap = prm.sortTemporal ? temporalMapEnv : 0u;
    if (stageCand)
    {
        p->doTemporal = (ip.flags & ZR_IND_TEMPORAL_RESAMPLE) && p->temporalValid && havePrevGBuffer;
        p->doSpatial = (ip.flags & ZR_IND_SPATIAL_RESAMPLE) && p->doTemporal && ip.num_spatial_passes > 0;      // IndirectLighting.cpp:906
    }
    // m_numSpatialPasses (IndirectLighting.cpp:616-621, 1240): 0..2
    if (ip.num_spatial_passes > 2u) return Fail(ZR_ERR_INVALID_ARG, "ReSTIR PT: num_spatial_passes must be 0, 1 or 2");
    const uint32_t numSpatialPasses = ip.num_spatial_passes;
    // staged (tile-split) rendering with two rounds: ZR_STAGE_SPATIAL runs the first, ZR_STAGE_SPATIAL2 the second -- it reads the first round's outputs at
    // neighbouring pixels, so the host exchanges ZR_HALO_POST_TEMPORAL (= the set the next stage reads: res[currIdx]) once more in between
    prm.doTemporal = p->doTemporal ? 1u : 0u;
    prm.doSpatial = p->doSpatial ? 1u : 0u;
    prm.writeReservoirs = (prm.doTemporal || !p->temporalValid) ? 1u : 0u;
    F.cur = p->RptCur().View(); F.prev = p->RptOth().View();
    const uint32_t tilesX = (F.ow + 15) / 16, tilesY = (F.oh + 15) / 16;
    const dim3 grid(tilesX * tilesY), block(kBlock);
    const dim3 gridRpt(tilesX * tilesY * (256 / kRptBlock)), blockRpt(kRptBlock);      // K11 (zr_kernels.h kRptBlock)
    const dim3 gridRecon(tilesX * tilesY * (256 / kReconBlock)), blockRecon(kReconBlock);      // K14 (zr_kernels.h kReconBlock)
    const dim3 gridStc(tilesX * tilesY * (256 / kStcBlock)), blockStc(kStcBlock);                // K16 (zr_kernels.h kStcBlock)
    const uint32_t sortTilesX = (F.ow + 31) / 32;
    const dim3 gridSort(sortTilesX * ((F.oh + 31) / 32));
    // work lists for the replay passes: [0] CtT, [1] TtC, [2] CtS, [3] StC (plane-local pixel ids, device-side counts)
    uint32_t* listCnt = p->rptListCounts.p;
    const size_t cap = (size_t)p->w * p->h;
    uint32_t* lists[4] = {p->rptLists.p, p->rptLists.p + cap, p->rptLists.p + 2 * cap, p->rptLists.p + 3 * cap};
    const dim3 gridList((uint32_t)std::min<size_t>((cap + kBlock - 1) / kBlock, 1024));
    const dim3 gridLight((tilesX * tilesY + kLightTilesPerBlock - 1u) / kLightTilesPerBlock), gridSearch((tilesX * tilesY + kSearchTilesPerBlock - 1u) / kSearchTilesPerBlock);      // k_rpt_light<0> / <1>: several tiles per block
    const dim3 gridReplay((uint32_t)std::min<size_t>((cap + kBlock - 1) / kBlock, kReplayPersistentBlocks));      // the temporal replays pull their work (zr_kernels.h)
    unsigned long long* ctr = p->counters.p;
#define RPT_TIMED(name, ...) do { TimerBegin(p, s, name); __VA_ARGS__; TimerEnd(p, s); } while (0)
    // the NEE_EMISSIVE permutation of a kernel (the reference compiles separate shaders, IndirectLighting.h:251-300)
    const bool emissiveVariant = prm.emissive != 0;
    // (zr_debug_set_large_scene_nodes: test hook, lets the parity tests run the large-scene kernel build on their small scenes)
    const uint32_t largeSceneNodes = g_largeSceneNodes.load(std::memory_order_relaxed);
    // ... and the TEXTURED permutation (this ABI's: untextured scenes carry no ray differentials)
    const bool texVariant = prm.textured != 0;
    // ... and the material-class permutation (zr_kernels.h PLAIN): scenes of opaque uncoated non-metallic dielectrics without a texture heap run kernels that have no code for the other lobes
    const bool plainVariant = PlainClass(sc, gb) && !texVariant;
#define RPT_LAUNCH_E(kern, ...) do { \
        if (plainVariant) { if (emissiveVariant) hipLaunchKernelGGL((kern<true, false, true>), __VA_ARGS__); else hipLaunchKernelGGL((kern<false, false, true>), __VA_ARGS__); } \
        else if (emissiveVariant) { if (texVariant) hipLaunchKernelGGL((kern<true, true, false>), __VA_ARGS__); else hipLaunchKernelGGL((kern<true, false, false>), __VA_ARGS__); } \
        else { if (texVariant) hipLaunchKernelGGL((kern<false, true, false>), __VA_ARGS__); else hipLaunchKernelGGL((kern<false, false, false>), __VA_ARGS__); } } while (0)
#define RPT_LAUNCH_PE(kern, PASS, ...) do { \
        if (emissiveVariant) { if (texVariant) hipLaunchKernelGGL((kern<PASS, true, true>), __VA_ARGS__); else hipLaunchKernelGGL((kern<PASS, true, false>), __VA_ARGS__); } \
        else { if (texVariant) hipLaunchKernelGGL((kern<PASS, false, true>), __VA_ARGS__); else hipLaunchKernelGGL((kern<PASS, false, false>), __VA_ARGS__); } } while (0)
    if (stageCand)
    {
        if (int ar = GBufferAcquireRead(gb, s)) return ar;
        if (carryFrom)
        {
            // ZR_FRAME_OVERLAP_CARRY: K11 and the passes behind it update reservoir records in place (Reservoir.hlsli:283-456 writes the components a record's case
            // uses), so the bytes a record does NOT use are whatever the set held before -- in the plain order the set K11 wrote last frame, here the free set.
            // Copying the replaced set (and target plane) into the one that takes its place makes every byte of every plane equal to the plain order's.
            const size_t n = (size_t)p->w * p->h;
            RPT_TIMED("rpt_overlap_carry", hipLaunchKernelGGL(k_rpt_carry, dim3((uint32_t)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, s, F.cur, carryFrom->View(), p->Target(),
                (const F4*)(p->tgtIdx ? p->rptTarget.p : p->rptTargetAlt.p), n));
        }
        TimerBegin(p, s, "rpt_pathtrace");
#ifdef ZR_EXPERIMENTS
        // (experiments build, zr_kernels_exp.h) ZR_K11=pool: K11 with block-pooled traces (k_rpt_pathtrace_coop; emissive untextured permutation);
        // compact: a kernel per bounce; trip: the alive-lane diagnostic; ZR_K11=inline: the megakernel
        static const int k11Mode = [] { const char* e = getenv("ZR_K11"); return e && !strcmp(e, "inline") ? 0 : (e && !strcmp(e, "pool") ? 1 : (e && !strcmp(e, "trip") ? 2 : (e && !strcmp(e, "compact") ? 3 : ZR_K11_DEFAULT))); }();
        static const bool park = [] { const char* e = getenv("ZR_K11_PARK"); return e ? atoi(e) != 0 : (ZR_K11_PARK_DEFAULT != 0); }();      // the 3-wave build with the reservoir's selected reconnection in LDS (zr_rpt.h RcPark)
        if (k11Mode == 1 && emissiveVariant && !texVariant)
        {
            const dim3 gridCoop(tilesX * tilesY), blockCoop(kCoopBlock);
            if (sc->view.numNodes >= largeSceneNodes) hipLaunchKernelGGL(k_rpt_pathtrace_coop_w4<false>, gridCoop, blockCoop, 0, s, F, *cb, tilesX, ctr + 2 * 1);
            else hipLaunchKernelGGL(k_rpt_pathtrace_coop<false>, gridCoop, blockCoop, 0, s, F, *cb, tilesX, ctr + 2 * 1);
        }
        else if (k11Mode == 3 && !texVariant)
        {   // a kernel per bounce, live paths compacted in between (zr_kernels.h: k_rpt_pt_first / k_rpt_pt_next)
            const size_t cap = (size_t)F.gb.w * F.gb.h;
            if (p->carry[0].n != cap * rpt::kPtCarryWords) { int rr; for (int k = 0; k < 2; k++) if ((rr = p->carry[k].Alloc(cap * rpt::kPtCarryWords))) return rr; if ((rr = p->carryCount.Alloc(16))) return rr; }
            HIP_TRY(hipMemsetAsync(p->carryCount.p, 0, 16 * sizeof(uint32_t), s));
            uint32_t maxB = prm.maxNonTrBounces > prm.maxGlossyTrBounces ? prm.maxNonTrBounces : prm.maxGlossyTrBounces;
            if (prm.russianRoulette && maxB > 3u) maxB = 3u;      // paths that go further are traced by whole tiles inside k_rpt_pt_first
            F.carryCap = cap; F.carryCount = p->carryCount.p; F.carryBounce = 0; F.carryOut = p->carry[0].p; F.carryIn = nullptr;
            if (emissiveVariant) hipLaunchKernelGGL(k_rpt_pt_first<true>, gridRpt, blockRpt, 0, s, F, *cb, tilesX, ctr + 2 * 1);
            else hipLaunchKernelGGL(k_rpt_pt_first<false>, gridRpt, blockRpt, 0, s, F, *cb, tilesX, ctr + 2 * 1);
            const dim3 gridNext((uint32_t)((cap + kRptBlock - 1) / kRptBlock));
            for (uint32_t b = 1; b + 1 <= maxB && b < 15u; b++)
            {
                F.carryBounce = b; F.carryIn = p->carry[(b - 1) & 1].p; F.carryOut = p->carry[b & 1].p;
                if (emissiveVariant) hipLaunchKernelGGL(k_rpt_pt_next<true>, gridNext, blockRpt, 0, s, F, *cb, tilesX, ctr + 2 * 1);
                else hipLaunchKernelGGL(k_rpt_pt_next<false>, gridNext, blockRpt, 0, s, F, *cb, tilesX, ctr + 2 * 1);
            }
        }
        else if (k11Mode == 2 && emissiveVariant && !texVariant)
        {   // diagnostic: the megakernel with a path-state round trip through SoA planes at every bounce boundary (zr_kernels.h)
            const size_t stride = (size_t)F.gb.w * F.gb.h;
            if (!p->trip.p) { int rr; if ((rr = p->trip.Alloc(stride * rpt::kPtCarryWords))) return rr; if ((rr = p->tripStats.Alloc(4))) return rr; HIP_TRY(hipMemsetAsync(p->tripStats.p, 0, 32, s)); }
            F.trip = p->trip.p; F.tripStats = p->tripStats.p; F.tripStride = stride;
            if (sc->view.numNodes >= largeSceneNodes) hipLaunchKernelGGL(k_rpt_pathtrace_trip_w4<false>, gridRpt, blockRpt, 0, s, F, *cb, tilesX, ctr + 2 * 1);
            else hipLaunchKernelGGL(k_rpt_pathtrace_trip<false>, gridRpt, blockRpt, 0, s, F, *cb, tilesX, ctr + 2 * 1);
        }
        else if (k11Mode == 0 && park && !texVariant && !(sc->view.numNodes >= largeSceneNodes || FewerRoundsAtFourWaves(gridRpt.x)))
        { if (emissiveVariant) hipLaunchKernelGGL(k_rpt_pathtrace_park<true>, gridRpt, blockRpt, 0, s, F, *cb, tilesX, ctr + 2 * 1); else hipLaunchKernelGGL(k_rpt_pathtrace_park<false>, gridRpt, blockRpt, 0, s, F, *cb, tilesX, ctr + 2 * 1); }
        else
#endif
        if (texVariant) { if (emissiveVariant) hipLaunchKernelGGL(k_rpt_pathtrace_tex<true>, gridRpt, blockRpt, 0, s, F, *cb, tilesX, ctr + 2 * 1); else hipLaunchKernelGGL(k_rpt_pathtrace_tex<false>, gridRpt, blockRpt, 0, s, F, *cb, tilesX, ctr + 2 * 1); }
        else if (sc->view.numNodes >= largeSceneNodes || FewerRoundsAtFourWaves(gridRpt.x))     // BVH beyond the caches, or a small grid: the 4-wave build of K11 (zr_kernels.h)
        {
            if (plainVariant) { if (emissiveVariant) hipLaunchKernelGGL((k_rpt_pathtrace_w4<true, true>), gridRpt, blockRpt, 0, s, F, *cb, tilesX, ctr + 2 * 1); else hipLaunchKernelGGL((k_rpt_pathtrace_w4<false, true>), gridRpt, blockRpt, 0, s, F, *cb, tilesX, ctr + 2 * 1); }
            else if (emissiveVariant) hipLaunchKernelGGL((k_rpt_pathtrace_w4<true, false>), gridRpt, blockRpt, 0, s, F, *cb, tilesX, ctr + 2 * 1); else hipLaunchKernelGGL((k_rpt_pathtrace_w4<false, false>), gridRpt, blockRpt, 0, s, F, *cb, tilesX, ctr + 2 * 1);
        }
        else
        {
            if (plainVariant) { if (emissiveVariant) hipLaunchKernelGGL((k_rpt_pathtrace<true, true>), gridRpt, blockRpt, 0, s, F, *cb, tilesX, ctr + 2 * 1); else hipLaunchKernelGGL((k_rpt_pathtrace<false, true>), gridRpt, blockRpt, 0, s, F, *cb, tilesX, ctr + 2 * 1); }
            else if (emissiveVariant) hipLaunchKernelGGL((k_rpt_pathtrace<true, false>), gridRpt, blockRpt, 0, s, F, *cb, tilesX, ctr + 2 * 1); else hipLaunchKernelGGL((k_rpt_pathtrace<false, false>), gridRpt, blockRpt, 0, s, F, *cb, tilesX, ctr + 2 * 1);
        }
        TimerEnd(p, s);
        if (p->overlap) { HIP_TRY(hipEventRecord(p->evCand, s)); p->candStream = s; p->haveCand = true; }
    }
    if (stageReuseT)
    {
        if (p->overlap && p->haveCand && p->candStream != s) HIP_TRY(hipStreamWaitEvent(s, p->evCand, 0));      // (K11, and with it the GBUFFER render before it on that stream)
        else if (int ar = GBufferAcquireRead(gb, s)) return ar;
        HIP_TRY(hipMemsetAsync(listCnt, 0, kRptListWords * sizeof(uint32_t), s));      // (the work lists are the reuse passes' alone: K11 of the next frame may already run)
        if (prm.doTemporal)
        {
            // K12 Sort_TtC / Sort_CtT (IndirectLighting.cpp:383-441: dispatched whether or not SORT_TEMPORAL is set).  The temporal reconnect
            // passes have no wave operations, so these two maps cannot change a result; they are outputs (ZR_OUT_RPT_THREAD_MAP_*)
            RPT_TIMED("rpt_sort_temporal", hipLaunchKernelGGL((k_rpt_sort<rpt::RPT_SORT_TTC, rpt::RPT_SORT_CTT>), dim3(gridSort.x * 2), dim3(256), 0, s, F, *cb, sortTilesX, F.ox0 / 32u, F.oy0 / 32u, F.mapNtC, F.mapCtN));
            RPT_TIMED("rpt_classify_temporal", hipLaunchKernelGGL(k_rpt_light<0>, gridLight, block, 0, s, F, *cb, tilesX, lists[0], lists[1], listCnt + 0));
            RPT_TIMED("rpt_replay_temporal", RPT_LAUNCH_PE(k_rpt_replay, RPT_REPLAY_CTT, gridReplay, block, 0, s, F, *cb, lists[0], lists[1], listCnt + 0, ctr + 2 * 2));
            RPT_TIMED("rpt_reconnect_temporal", RPT_LAUNCH_E(k_rpt_temporal, gridRecon, blockRecon, 0, s, F, *cb, tilesX, ctr + 2 * 4));
        }
        // nothing after this point reads the PREVIOUS frame's G-buffer, its final reservoirs or scene (the spatial passes read this frame's only)
        if (p->overlap) { HIP_TRY(hipEventRecord(p->evTemporal, s)); p->reuseStream = s; p->haveTemporal = true; }
        if (int mr = GBufferMarkRead(gb, gb->Prev(), s)) return mr;
    }
    for (uint32_t spass = 0; spass < numSpatialPasses && prm.doSpatial; spass++)
    {
        if (!(stages & (spass == 0 ? ZR_STAGE_SPATIAL : ZR_STAGE_SPATIAL2))) continue;
        // a round reads res[currIdx] and writes the other set, which then becomes "current" (IndirectLighting.cpp:609-612, 682-688: one flip per round)
        F.cur = p->RptCur().View(); F.prev = p->RptOth().View();
        // replay work lists + their device-side counts of this round: {2, 3} for the first, {6, 7} for the second (zeroed at the start of the frame)
        uint32_t* const sCnt = listCnt + (spass == 0 ? 2 : 6);
#ifdef ZR_EXPERIMENTS
        // ZR_SEARCH=tile: the LDS-tiled K15 (k_rpt_light<2>), kept for the A/B of DESIGN's N3 row -- measured slower than the plain gathers
        static const bool searchTile = [] { const char* e = getenv("ZR_SEARCH"); return e && !strcmp(e, "tile"); }();
        if (searchTile) RPT_TIMED("rpt_spatial_search", hipLaunchKernelGGL(k_rpt_light<2>, grid, block, 0, s, F, *cb, tilesX, lists[2], lists[3], sCnt));
        else
#endif
        RPT_TIMED("rpt_spatial_search", hipLaunchKernelGGL(k_rpt_light<1>, gridSearch, block, 0, s, F, *cb, tilesX, lists[2], lists[3], sCnt));
        // K12 Sort_CtS / Sort_StC (IndirectLighting.cpp:690-742): the NtC map decides which pixels share a wave in Reconnect_StC, i.e. the
        // population of its boiling-suppression averages
        if (prm.sortSpatial)
        {
            RPT_TIMED("rpt_sort_spatial", hipLaunchKernelGGL((k_rpt_sort<rpt::RPT_SORT_CTS, rpt::RPT_SORT_STC>), dim3(gridSort.x * 2), dim3(256), 0, s, F, *cb, sortTilesX, F.ox0 / 32u, F.oy0 / 32u, F.mapCtN, F.mapNtC));
        }
        RPT_TIMED("rpt_replay_spatial", RPT_LAUNCH_PE(k_rpt_replay, RPT_REPLAY_CTS, dim3(gridList.x * 2), block, 0, s, F, *cb, lists[2], lists[3], sCnt, ctr + 2 * 5));
        RPT_TIMED("rpt_reconnect_spatial", RPT_LAUNCH_E(k_rpt_stc, gridStc, blockStc, 0, s, F, *cb, tilesX, ctr + 2 * 7));
        // "Prepare for next iteration" (IndirectLighting.cpp:860-870: std::swap(inputs, outputs)) is the flip itself here: the next round starts from res[currIdx]
        p->currIdx = 1 - p->currIdx;
    }
#undef RPT_TIMED
#undef RPT_LAUNCH_E
#undef RPT_LAUNCH_PE
    HIP_TRY(hipGetLastError());
    if (stageCand) p->frameOpen = true;
    // the frame ends with its last stage: the second round when there is one this frame, else ZR_STAGE_SPATIAL; Render() flips once more (:1018-1024)
    const bool lastStage = (prm.doSpatial && numSpatialPasses == 2u) ? (stages & ZR_STAGE_SPATIAL2) != 0 : (stages & ZR_STAGE_SPATIAL) != 0;
    if (lastStage && p->frameOpen)
    {
        p->temporalValid = true;
        p->currIdx = 1 - p->currIdx;
        p->frameOpen = false;
        p->finOut = p->finIdx;
        if (p->overlap) { const int k = (int)(p->ovFrame & 1); HIP_TRY(hipEventRecord(p->evDone[k], s)); p->haveDone[k] = true; p->doneStream[k] = s; p->ovFrame++; }
        if (int mr = GBufferMarkRead(gb, gb->cur, s)) return mr;
    }
    return ZR_OK;
}

static int RenderIndirect(zr_pass* p, hipStream_t s, const zr_frame_constants* cb, const zr_scene* sc, zr_gbuffer* gb, int stages)
{
    if (!gb) return Fail(ZR_ERR_INVALID_ARG, "INDIRECT pass needs a gbuffer");
    if (gb->w != p->w || gb->h != p->h || gb->x0 + gb->w > cb->render_width || gb->y0 + gb->h > cb->render_height)
        return Fail(ZR_ERR_INVALID_ARG, "frame constants / gbuffer tile / pass size mismatch");
    if (sc->view.numEmissives == 0)
    {
        // NEE_EMISSIVE == 0 shader variants: sun + sky next-event estimation
        if (!sc->view.sky.data) return Fail(ZR_ERR_NOT_INITIALIZED, "sky-view LUT missing: render a ZR_PASS_SKY pass first");
    }
    else if (!sc->view.alias) return Fail(ZR_ERR_NOT_INITIALIZED, "emissive alias table missing: render the PRELIGHTING pass (or zr_scene_set_alias_table) first");
    if (cb->num_emissive_triangles != sc->view.numEmissives) return Fail(ZR_ERR_INVALID_ARG, "cbFrameConstants.NumEmissiveTriangles != scene");
    if (p->params.presampling && (!sc->view.sampleSets || sc->numSampleSets != p->params.num_sample_sets || sc->view.sampleSetSize != p->params.sample_set_size))
        return Fail(ZR_ERR_NOT_INITIALIZED, "presampled light sets missing or of another size: render the PRELIGHTING pass with the same presampling params first");
    if (p->integrator == ZR_INTEGRATOR_RESTIR_PT) return RenderReSTIR_PT(p, s, cb, sc, gb, stages);
    if (!(stages & ZR_STAGE_TEMPORAL)) return ZR_OK;
    if (p->integrator == ZR_INTEGRATOR_RESTIR_GI) return RenderReSTIR_GI(p, s, cb, sc, gb);      // single-stage integrators render in the first stage
    PtParams prm;
    prm.maxNonTrBounces = p->params.max_non_tr_bounces; prm.maxGlossyTrBounces = p->params.max_glossy_tr_bounces;
    prm.russianRoulette = (p->params.flags & ZR_IND_RUSSIAN_ROULETTE) ? 1u : 0u;
    prm.numSampleSets = p->params.presampling ? p->params.num_sample_sets : 0u;
    prm.accumulate = (cb->accumulate && cb->camera_static) ? 1u : 0u;
    prm.tileW = p->w; prm.groupsX = (p->w + 7) / 8;
    const uint32_t numGroups = prm.groupsX * ((p->h + 7) / 8);
    const uint32_t maxB = prm.maxNonTrBounces > prm.maxGlossyTrBounces ? prm.maxNonTrBounces : prm.maxGlossyTrBounces;
    const int rounds = (int)maxB + 1;
    if (rounds > kMaxRounds) return Fail(ZR_ERR_INVALID_ARG, "too many bounces");

    HIP_TRY(hipMemsetAsync(p->counts.p, 0, 5 * (kMaxRounds + 2) * kCounterStride * sizeof(uint32_t), s));
    // counter (kind, round): kind 0 = live paths, 1 = k_trace cursor, 2..4 = C / M / S rays
    auto Ctr = [&](int kind, int round) { return p->counts.p + ((size_t)round * 5 + kind) * kCounterStride; };
    // Russian roulette can only trigger once bounce >= 3, i.e. from round 2 on and only if some path may take >= 4 bounces
    const bool rrPossible = prm.russianRoulette && maxB >= 4;
    if (rrPossible) HIP_TRY(hipMemsetAsync(p->groupMax.p, 0, (size_t)rounds * numGroups * sizeof(uint32_t), s));
    const uint32_t tilesX = (p->w + 15) / 16, tilesY = (p->h + 15) / 16;
    const GBuf gbv = gb->View();
    SceneView scv = FrameView(sc, cb);
    scv.texFilter = p->params.tex_filter;
    const bool tex = scv.tex.count != 0;      // kernels carry ray differentials only when there is a texture heap
    const bool plainPt = PlainClass(sc, gb);
    if (tex) { int r; if ((r = p->q[0].AllocTex((size_t)p->w * p->h)) || (r = p->q[1].AllocTex((size_t)p->w * p->h))) return r; }
    TimerBegin(p, s, "pt_init");
    hipLaunchKernelGGL((tex ? k_pt_init<true, false> : plainPt ? k_pt_init<false, true> : k_pt_init<false, false>), dim3(tilesX * tilesY), dim3(kBlock), 0, s, scv, *cb, gbv, prm, p->finalRGBA.p, p->firstBOP.p,
        p->q[0].View(), Ctr(0, 0), Ctr(2, 0), (uint32_t)((size_t)p->w * p->h), tilesX);
    TimerEnd(p, s);
    const size_t cap = (size_t)p->w * p->h;
    const uint32_t gridShade = (uint32_t)std::min<size_t>((cap + kBlock - 1) / kBlock, 4096);
    const uint32_t gridTrace = (uint32_t)std::min<size_t>((3 * cap + kBlock - 1) / kBlock, 2048);    // persistent: 256 CUs x 8 blocks
    const uint32_t gridTraceSimple = (uint32_t)std::min<size_t>((3 * cap + kBlock - 1) / kBlock, 8192);
    static const int traceMode = [] { const char* e = ZR_EXP_ENV("ZR_TRACE_MODE"); return e ? atoi(e) : 0; }();
    for (int r = 0; r < rounds; r++)
    {
        const PathQueue qin = p->q[r & 1].View(), qout = p->q[(r + 1) & 1].View();
        TimerBegin(p, s, "trace");
        if (traceMode == 0) hipLaunchKernelGGL(k_trace_simple, dim3(gridTraceSimple), dim3(kBlock), 0, s, scv, qin, Ctr(2, r), (uint32_t)cap, p->counters.p);
        else hipLaunchKernelGGL(k_trace, dim3(gridTrace), dim3(kBlock), 0, s, scv, qin, Ctr(2, r), (uint32_t)cap, Ctr(1, r), p->counters.p);
        TimerEnd(p, s);
        TimerBegin(p, s, "pt_shade");
        hipLaunchKernelGGL(tex ? k_pt_shade_tex : plainPt ? k_pt_shade<true> : k_pt_shade<false>, dim3(gridShade), dim3(kBlock), 0, s, scv, *cb, prm, qin, Ctr(0, r), qout, Ctr(0, r + 1), Ctr(2, r + 1), (uint32_t)cap,
            p->finalRGBA.p, p->firstBOP.p, p->groupMax.p + (size_t)r * numGroups);
        TimerEnd(p, s);
        if (rrPossible && r >= 2)
        {
            TimerBegin(p, s, "pt_rr");
            hipLaunchKernelGGL(tex ? k_pt_rr<true> : k_pt_rr<false>, dim3(gridShade), dim3(kBlock), 0, s, scv, prm, qout, Ctr(0, r + 1), Ctr(2, r + 1), (uint32_t)cap, p->groupMax.p + (size_t)r * numGroups);
            TimerEnd(p, s);
        }
    }
    HIP_TRY(hipGetLastError());
    return ZR_OK;
}

int zr_pass_render(zr_pass* p, void* stream, const zr_frame_constants* cb, const zr_scene* sc, zr_gbuffer* gb)
{ return zr_pass_render_stage(p, stream, cb, sc, gb, ZR_STAGE_ALL | ZR_STAGE_SPATIAL2); }

int zr_pass_set_input(zr_pass* p, int which, const void* dev)
{
    if (p && p->kind == ZR_PASS_TAA && which == ZR_IN_TAA_SIGNAL) { p->compIn[3] = (const F4*)dev; return ZR_OK; }
    if (p && p->kind == ZR_PASS_DENOISE && which == ZR_IN_DENOISE_SIGNAL) { p->compIn[3] = (const F4*)dev; return ZR_OK; }
    if (p && (p->kind == ZR_PASS_AUTO_EXPOSURE || p->kind == ZR_PASS_DISPLAY))
    {
        if (which == ZR_IN_POST_SIGNAL_F16) { p->postIn16 = (const uint16_t*)dev; p->postIn32 = nullptr; return ZR_OK; }
        if (which == ZR_IN_POST_SIGNAL_F32) { p->postIn32 = (const F4*)dev; p->postIn16 = nullptr; return ZR_OK; }
        if (which == ZR_IN_DISPLAY_EXPOSURE && p->kind == ZR_PASS_DISPLAY) { p->exposureIn = (const float*)dev; return ZR_OK; }
        return Fail(ZR_ERR_INVALID_ARG, "zr_pass_set_input: bad input id for an AUTO_EXPOSURE / DISPLAY pass");
    }
    if (!p || p->kind != ZR_PASS_COMPOSITING || which < 0 || which > 2) return Fail(ZR_ERR_INVALID_ARG, "zr_pass_set_input: not a COMPOSITING / TAA pass or bad input id");
    p->compIn[which] = (const F4*)dev;
    return ZR_OK;
}
extern "C" int zr_pass_set_tonemap_lut(zr_pass* p, const uint32_t* rgb9e5, uint32_t dim)
{
    if (!p || p->kind != ZR_PASS_DISPLAY || !rgb9e5 || dim == 0 || dim > 256) return Fail(ZR_ERR_INVALID_ARG, "zr_pass_set_tonemap_lut: needs a DISPLAY pass and a dim^3 LUT");
    HIP_TRY(hipSetDevice(p->device));
    const size_t n = (size_t)dim * dim * dim;
    int r; if ((r = p->tonemapLut.Alloc(n))) return r;
    HIP_TRY(hipMemcpy(p->tonemapLut.p, rgb9e5, n * sizeof(uint32_t), hipMemcpyHostToDevice));
    p->tonemapLutDim = dim;
    return ZR_OK;
}

static post::AeParams AeParamsOf(const zr_params& ip)
{ post::AeParams a; a.minLum = ip.ae_min_lum; a.lumRange = ip.ae_max_lum - ip.ae_min_lum; a.lumMapExp = ip.ae_lum_map_exp; a.adaptationRate = ip.ae_adaptation_rate; return a; }

// AutoExposure::Render (AutoExposure.cpp:100-143): clear the histogram, HISTOGRAM over the image, WEIGHTED_AVG in one group
static int RenderAutoExposure(zr_pass* p, hipStream_t s, const zr_frame_constants* cb)
{
    if (cb->render_width != p->w || cb->render_height != p->h) return Fail(ZR_ERR_INVALID_ARG, "AUTO_EXPOSURE: frame constants / pass size mismatch");
    if (!p->postIn16 && !p->postIn32) return Fail(ZR_ERR_NOT_INITIALIZED, "AUTO_EXPOSURE: no input bound (zr_pass_set_input(ZR_IN_POST_SIGNAL_*))");
    const uint32_t n = p->w * p->h;
    const post::AeParams prm = AeParamsOf(p->params);
    HIP_TRY(hipMemsetAsync(p->aeHist.p, 0, post::kHistBins * sizeof(uint32_t), s));
    TimerBegin(p, s, "ae_histogram");
    const uint32_t grid = std::min<uint32_t>((n + 255) / 256, 2048u);
    hipLaunchKernelGGL(k_ae_histogram, dim3(grid), dim3(256), 0, s, p->postIn16, p->postIn32, n, prm, p->aeHist.p);
    TimerEnd(p, s);
    TimerBegin(p, s, "ae_resolve");
    hipLaunchKernelGGL(k_ae_resolve, dim3(1), dim3(256), 0, s, p->aeHist.p, n, cb->dt, prm, p->aeExposure.p);
    TimerEnd(p, s);
    HIP_TRY(hipGetLastError());
    return ZR_OK;
}
// DisplayPass::Render (Display.cpp:188-260), the full-screen triangle of mainPS as one thread per display pixel
static int RenderDisplay(zr_pass* p, hipStream_t s, const zr_frame_constants* cb)
{
    if (cb->display_width != p->w || cb->display_height != p->h) return Fail(ZR_ERR_INVALID_ARG, "DISPLAY: the pass size must be the display size of the frame constants");
    if (!p->postIn16 && !p->postIn32) return Fail(ZR_ERR_NOT_INITIALIZED, "DISPLAY: no input bound (zr_pass_set_input(ZR_IN_POST_SIGNAL_*))");
    const zr_params& ip = p->params;
    if (ip.display_tonemapper >= post::TM_COUNT) return Fail(ZR_ERR_INVALID_ARG, "DISPLAY: unknown tone mapper %u", ip.display_tonemapper);
    if (ip.display_tonemapper == ZR_TONEMAP_NEUTRAL && !p->tonemapLutDim) return Fail(ZR_ERR_NOT_INITIALIZED, "DISPLAY: the NEUTRAL tone mapper needs zr_pass_set_tonemap_lut");
    if (ip.display_auto_exposure && !p->exposureIn) return Fail(ZR_ERR_NOT_INITIALIZED, "DISPLAY: auto exposure is on but no exposure bound (ZR_IN_DISPLAY_EXPOSURE)");
    post::DisplayParams prm; prm.tonemapper = ip.display_tonemapper; prm.autoExposure = ip.display_auto_exposure; prm.saturation = ip.display_saturation; prm.agxExp = ip.display_agx_exp;
    post::Lut3D lut; lut.data = p->tonemapLut.p; lut.dim = p->tonemapLutDim;
    const uint32_t n = p->w * p->h;
    TimerBegin(p, s, "display");
    hipLaunchKernelGGL(k_display, dim3((n + 255) / 256), dim3(256), 0, s, p->postIn16, p->postIn32, cb->render_width, cb->render_height, p->w, p->h,
        p->exposureIn, prm, lut, p->displayOut.p, p->displaySrgb.p);
    TimerEnd(p, s);
    HIP_TRY(hipGetLastError());
    return ZR_OK;
}

// TAA::Render (TAA.cpp:75-118): reads the other output as history, then the roles swap
static int RenderTAA(zr_pass* p, hipStream_t s, const zr_frame_constants* cb, zr_gbuffer* gb)
{
    if (!gb || gb->w != p->w || gb->h != p->h) return Fail(ZR_ERR_INVALID_ARG, "TAA needs a gbuffer of the pass size");
    if (cb->render_width != p->w || cb->render_height != p->h) return Fail(ZR_ERR_INVALID_ARG, "TAA: frame constants / pass size mismatch");
    if (!p->compIn[3]) return Fail(ZR_ERR_NOT_INITIALIZED, "TAA: no input bound (zr_pass_set_input(ZR_IN_TAA_SIGNAL))");
    const int curr = 1 - p->taaIdx;
    taa::TaaFrame F;
    F.signal = p->compIn[3]; F.depth = (const float*)gb->Planes()[ZR_GB_DEPTH].p; F.motion = (const uint32_t*)gb->Planes()[ZR_GB_MOTION_VECTOR].p;
    F.prevOut = p->taaOut[p->taaIdx].p; F.currOut = p->taaOut[curr].p; F.w = p->w; F.h = p->h;
    F.blendWeight = p->params.taa_blend_weight; F.temporalIsValid = p->temporalValid ? 1u : 0u;
    const uint32_t n = p->w * p->h;
    TimerBegin(p, s, "taa");
    hipLaunchKernelGGL(k_taa, dim3((n + 255) / 256), dim3(256), 0, s, F);
    TimerEnd(p, s);
    HIP_TRY(hipGetLastError());
    p->taaIdx = curr; p->temporalValid = true;
    return ZR_OK;
}

// Denoise pass: temporal accumulation -> variance estimate -> a-trous iterations (zr_svgf.h; no reference counterpart).
// `steps`: which of them this call runs (ZR_STAGE_DENOISE_*; zr_pass_render = all).  A device of the tile split runs them in groups with halo
// exchanges in between (zetaray_amd/tiling.py denoise_schedule): the planes then cover the tile + its apron, a window of the frame.
static int RenderDenoise(zr_pass* p, hipStream_t s, const zr_frame_constants* cb, zr_gbuffer* gb, uint32_t steps)
{
    if (!gb || gb->w != p->w || gb->h != p->h) return Fail(ZR_ERR_INVALID_ARG, "DENOISE needs a gbuffer of the pass size");
    if (gb->x0 + gb->w > cb->render_width || gb->y0 + gb->h > cb->render_height) return Fail(ZR_ERR_INVALID_ARG, "DENOISE: the planes' window leaves the frame of the frame constants");
    if (!p->compIn[3]) return Fail(ZR_ERR_NOT_INITIALIZED, "DENOISE: no input signal (zr_pass_set_input(pass, ZR_IN_DENOISE_SIGNAL, rgba32f))");
    const zr_params& prm = p->params;
    if (prm.svgf_iterations > 8u || prm.svgf_normal_power_log2 > 16u) return Fail(ZR_ERR_INVALID_ARG, "DENOISE: svgf_iterations must be <= 8, svgf_normal_power_log2 <= 16");
    if (!(prm.svgf_sigma_z > 0.0f) || !(prm.svgf_sigma_l >= 0.0f) || !(prm.svgf_sigma_z < 1e30f) || !(prm.svgf_sigma_l < 1e30f)) return Fail(ZR_ERR_INVALID_ARG, "DENOISE: svgf_sigma_z must be positive, svgf_sigma_l non-negative, both finite");
    const GBuf cur = gb->View(), prev = gb->PrevView();
    svgf::SvgfParams sp; sp.alpha = prm.svgf_alpha; sp.alphaMoments = prm.svgf_alpha_moments; sp.sigmaL = prm.svgf_sigma_l; sp.sigmaZ = prm.svgf_sigma_z;
    sp.normalPowerLog2 = prm.svgf_normal_power_log2; sp.iterations = prm.svgf_iterations;
    svgf::Window win; win.ox = (int)gb->x0; win.oy = (int)gb->y0; win.pw = (int)p->w; win.ph = (int)p->h; win.W = (int)cb->render_width; win.H = (int)cb->render_height;
    const dim3 grid((p->w + 31u) / 32u, (p->h + 7u) / 8u), block(256);
    const bool pow7 = sp.normalPowerLog2 == 7u;
    const int mi = p->svgfMomIdx;      // histMoments = [mi], this frame's = [mi ^ 1]; flipped when the frame's last step has run
    if (steps & ZR_STAGE_DENOISE_TEMPORAL)
    {
        svgf::SvgfFrame T;
        T.signal = p->compIn[3]; T.depth = cur.depth; T.normal = cur.normal; T.motion = cur.motion; T.prevDepth = prev.depth; T.prevNormal = prev.normal;
        T.histColor = p->svgfHist.p; T.histMoments = p->svgfMoments[mi].p; T.accum = p->svgfAccum.p; T.moments = p->svgfMoments[mi ^ 1].p; T.guide = p->svgfGuide.p; T.guideFw = p->svgfGuideFw.p;
        T.win = win; T.temporalValid = (p->temporalValid && gb->numRendered >= 2) ? 1u : 0u; T.prm = sp;
        TimerBegin(p, s, "denoise_temporal");
        hipLaunchKernelGGL(k_svgf_temporal, grid, block, 0, s, T);
        TimerEnd(p, s);
        p->svgfStepsDone = 0;
    }
    svgf::FilterFrame V;
    V.src = p->svgfAccum.p; V.moments = p->svgfMoments[mi ^ 1].p; V.guide = p->svgfGuide.p; V.guideFw = p->svgfGuideFw.p; V.guideZ = cur.depth; V.dst = p->svgfPing.p; V.lenSrc = p->svgfAccum.p;
    V.history = sp.iterations == 0 ? p->svgfHist.p : nullptr; V.win = win; V.step = 1; V.prm = sp;
    V.dstPacked = sp.iterations != 0;      // the planes between two stages hold fp16 colour + normal (zr_svgf.h PackStage); the pass's last stage writes fp32
    if (steps & ZR_STAGE_DENOISE_VARIANCE)
    {
        TimerBegin(p, s, "denoise_variance");
        if (pow7) hipLaunchKernelGGL(k_svgf_variance<7>, grid, block, 0, s, V); else hipLaunchKernelGGL(k_svgf_variance<-1>, grid, block, 0, s, V);
        TimerEnd(p, s);
        p->svgfCur = p->svgfPing.p;
    }
    bool timing = false;
    for (uint32_t it = 0; it < sp.iterations; it++)
    {
        if (!(steps & ZR_STAGE_DENOISE_ATROUS(it))) continue;
        if (!timing) { TimerBegin(p, s, "denoise_atrous"); timing = true; }
        F4* src = p->svgfCur; F4* dst = src == p->svgfPing.p ? p->svgfPong.p : p->svgfPing.p;
        svgf::FilterFrame A = V;
        A.src = src; A.dst = dst; A.moments = nullptr; A.step = 1u << it; A.history = it == 0 ? p->svgfHist.p : nullptr; A.dstPacked = it + 1u != sp.iterations;
        // LDS-staged tiles for the dense iterations: steps 1, 2 and 4 (48 x 24 tile, 36.8 KB per block, for the last).  With definition 2 the iterations that
        // read their taps from the planes are L1-bound (800 B of taps per pixel through 64 B / clk / CU: 0.24 ms at 3840 x 2160 against 0.14 - 0.15 ms from
        // LDS, profiles/r04c_post_sqA.csv); steps 8 and 16 do not fit a tile.  ZR_DENOISE=plain: every iteration from the planes; ZR_DENOISE=lds2: steps 1 and 2 only
        static const int ldsSteps = [] { const char* e = ZR_EXP_ENV("ZR_DENOISE"); return e && !strcmp(e, "plain") ? 0 : (e && !strcmp(e, "lds2") ? 2 : 3); }();
        // the row-loop form of the tap stencil (zr_svgf.h AtrousPixelT) is the default: 5 iterations 0.978 ms against 1.006 ms for the fully unrolled form at
        // 3840 x 2160 (profiles/r04c_post_chain*.jsonl), a fifth of the code; ZR_DENOISE_TAPS=unroll selects the other
        static const bool rowLoop = [] { const char* e = ZR_EXP_ENV("ZR_DENOISE_TAPS"); return !(e && !strcmp(e, "unroll")); }();
#define ZR_SVGF_LAUNCH(K, ...) do { if (!pow7) hipLaunchKernelGGL((K<__VA_ARGS__ -1, true>), grid, block, 0, s, A); else if (rowLoop) hipLaunchKernelGGL((K<__VA_ARGS__ 7, true>), grid, block, 0, s, A); \
            else hipLaunchKernelGGL((K<__VA_ARGS__ 7, false>), grid, block, 0, s, A); } while (0)
        if (it == 0 && ldsSteps >= 1) ZR_SVGF_LAUNCH(k_svgf_atrous_lds, 1,);
        else if (it == 1 && ldsSteps >= 2) ZR_SVGF_LAUNCH(k_svgf_atrous_lds, 2,);
        else if (it == 2 && ldsSteps >= 3) ZR_SVGF_LAUNCH(k_svgf_atrous_lds, 4,);
        else ZR_SVGF_LAUNCH(k_svgf_atrous,);
#undef ZR_SVGF_LAUNCH
        p->svgfCur = dst;
    }
    if (timing) TimerEnd(p, s);
    HIP_TRY(hipGetLastError());
    // the frame is complete when its last step has run: the last a-trous iteration, or the variance stage when there are none
    const uint32_t last = sp.iterations ? ZR_STAGE_DENOISE_ATROUS(sp.iterations - 1u) : (uint32_t)ZR_STAGE_DENOISE_VARIANCE;
    if (steps & last) { p->svgfOut = p->svgfCur; p->svgfMomIdx = mi ^ 1; p->temporalValid = true; }
    return ZR_OK;
}

static int RenderCompositing(zr_pass* p, hipStream_t s, const zr_frame_constants* cb, const zr_scene* sc, zr_gbuffer* gb)
{
    if (!gb || gb->w != p->w || gb->h != p->h) return Fail(ZR_ERR_INVALID_ARG, "COMPOSITING needs a gbuffer of the pass size");
    const uint32_t n = p->w * p->h;
    TimerBegin(p, s, "compositing");
    // with the firefly filter on (Compositing.cpp: m_filterFirefly) the composited image goes to a scratch plane and the filter writes FINAL
    const bool firefly = (p->params.flags & ZR_COMPOSIT_FIREFLY_FILTER) != 0;
    F4* composited = firefly ? p->firstBOP.p : (F4*)p->finalRGBA.p;
    hipLaunchKernelGGL(k_composite, dim3((n + 255) / 256), dim3(256), 0, s, *cb, (const uint16_t*)gb->Planes()[ZR_GB_METALLIC_ROUGHNESS].p, p->compIn[ZR_IN_SKY_DI],
        p->compIn[ZR_IN_EMISSIVE_DI], p->compIn[ZR_IN_INDIRECT], composited, n, FrameView(sc, cb).sky, p->w);
    TimerEnd(p, s);
    if (firefly)
    {
        TimerBegin(p, s, "firefly_filter");
        hipLaunchKernelGGL(k_firefly, dim3((p->w + kFfW - 1) / kFfW, (p->h + kFfH - 1) / kFfH), dim3(256), 0, s, composited, (const float*)gb->Planes()[ZR_GB_DEPTH].p,
            (F4*)p->finalRGBA.p, p->w, p->h);
        TimerEnd(p, s);
    }
    HIP_TRY(hipGetLastError());
    return ZR_OK;
}

int zr_device_synchronize(int device)
{
    if (int r = RequireDevice(device)) return r;
    HIP_TRY(hipDeviceSynchronize());
    return ZR_OK;
}
int zr_pass_set_frame_overlap(zr_pass* p, zr_gbuffer* gb, int enable)
{
    if (!p || !gb) return Fail(ZR_ERR_INVALID_ARG, "zr_pass_set_frame_overlap: null argument");
    if (!p->initialized) return Fail(ZR_ERR_NOT_INITIALIZED, "pass not initialised");
    if (p->kind != ZR_PASS_INDIRECT || p->integrator != ZR_INTEGRATOR_RESTIR_PT) return Fail(ZR_ERR_UNSUPPORTED, "frame overlap is a mode of the ReSTIR PT pass");
    if (p->frameOpen) return Fail(ZR_ERR_INVALID_ARG, "zr_pass_set_frame_overlap: call between frames (a frame's last stage has not been enqueued)");
    if (gb->device != p->device) return Fail(ZR_ERR_INVALID_ARG, "gbuffer / pass live on different devices");
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipDeviceSynchronize());      // (a host call between frames; the planes allocated below are cleared on the null stream)
    if (enable && !p->res[2].A.p) { if (int r = AllocOverlapPlanes(p)) return r; HIP_TRY(hipDeviceSynchronize()); }
    if (enable && !p->evCand)
    {
        HIP_TRY(hipEventCreateWithFlags(&p->evCand, hipEventDisableTiming)); HIP_TRY(hipEventCreateWithFlags(&p->evTemporal, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&p->evDone[0], hipEventDisableTiming)); HIP_TRY(hipEventCreateWithFlags(&p->evDone[1], hipEventDisableTiming));
        HIP_TRY(hipStreamCreateWithFlags(&p->overlapStream, hipStreamNonBlocking));
    }
    // switching off keeps the plane roles as they stand (the set last written as FINAL stays the one the next frame reads); FINAL goes on in the plane it is in
    p->overlap = enable != 0; p->overlapCarry = enable == ZR_FRAME_OVERLAP_CARRY; p->haveCand = false; p->haveTemporal = false; p->haveDone[0] = p->haveDone[1] = false;
    if (enable && gb->numSets == 2)
    {   // the third plane set (never given back: the sets' roles rotate)
        for (int i = 0; i < ZR_GB_COUNT; i++)
        {
            if (int r = gb->planeSets[2][i].Alloc((size_t)gb->w * gb->h * ZR_GB_PLANE_BYTES[i])) return r;
            HIP_TRY(hipMemset(gb->planeSets[2][i].p, 0, gb->planeSets[2][i].n));
        }
        HIP_TRY(hipDeviceSynchronize());
        // sets 0 / 1 hold the last two frames; the rotation continues from the current one: cur = 0 -> next 1 (two frames old), cur = 1 -> next 2 (fresh)
        gb->numSets = 3;
    }
    gb->tracked = enable != 0;
    if (!enable) { gb->hasWrite = false; for (auto& v : gb->readers) { for (auto& r : v) (void)hipEventDestroy(r.ev); v.clear(); } }
    return ZR_OK;
}
int zr_pass_frame_overlap_stream(zr_pass* p, void** stream)
{
    if (!p || !stream) return Fail(ZR_ERR_INVALID_ARG, "null argument");
    if (!p->overlapStream) return Fail(ZR_ERR_NOT_INITIALIZED, "frame overlap has never been enabled on this pass (zr_pass_set_frame_overlap)");
    *stream = (void*)p->overlapStream;
    return ZR_OK;
}
int zr_pass_set_owned_rect(zr_pass* p, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h)
{
    if (!p) return Fail(ZR_ERR_INVALID_ARG, "null pass");
    if ((w == 0) != (h == 0)) return Fail(ZR_ERR_INVALID_ARG, "owned rect: width and height must both be 0 (whole tile) or both be positive");
    if (w && ((x0 & 31u) || (y0 & 31u))) return Fail(ZR_ERR_INVALID_ARG, "owned rect origin must be 32-pixel aligned");
    if (w && p->initialized && (w > p->w || h > p->h)) return Fail(ZR_ERR_INVALID_ARG, "owned rect %ux%u larger than the pass %ux%u", w, h, p->w, p->h);
    p->own[0] = x0; p->own[1] = y0; p->own[2] = w; p->own[3] = h;
    return ZR_OK;
}

// which reservoir set a halo transfer addresses (zr_halo_set)
static zr_pass::ResStorage* HaloSet(zr_pass* p, int which)
{
    // between the stages of a frame the post-temporal reservoirs are res[currIdx]; after the frame the set the next
    // frame reads as "previous" is res[1 - currIdx]
    return &p->res[p->rptSet[which == ZR_HALO_POST_TEMPORAL ? p->currIdx : 1 - p->currIdx]];
}
struct HaloPlane { void* base; size_t bpp; };
// the planes a halo transfer of this pass moves, and their bytes per pixel
static int HaloPlanes(zr_pass* p, int which, HaloPlane* planes, size_t* bytesPerPixel)
{
    // between the stages of a frame the post-temporal set is [currIdx]; after the frame the set the next frame reads as
    // "previous" is [1 - currIdx]
    const int set = which == ZR_HALO_POST_TEMPORAL ? p->currIdx : 1 - p->currIdx;
    int n = 0;
    if (p->kind == ZR_PASS_INDIRECT && p->integrator == ZR_INTEGRATOR_RESTIR_PT)
    {
        zr_pass::ResStorage* R = &p->res[p->rptSet[set]];
        const HaloPlane pl[7] = {{R->A.p, 4}, {R->B.p, 8}, {R->C.p, 16}, {R->D.p, 16}, {R->E.p, 2}, {R->F.p, 8}, {R->G.p, 8}};
        for (auto& q : pl) planes[n++] = q;
    }
    else if (p->kind == ZR_PASS_INDIRECT && p->integrator == ZR_INTEGRATOR_RESTIR_GI)
    { planes[n++] = {p->giA[set].p, 16}; planes[n++] = {p->giB[set].p, 8}; planes[n++] = {p->giC[set].p, 16}; }
    else if (p->kind == ZR_PASS_DI_EMISSIVE) { planes[n++] = {p->diA[set].p, 16}; planes[n++] = {p->diB[set].p, 8}; }
    else if (p->kind == ZR_PASS_DI_SKY) { planes[n++] = {p->skyA[set].p, 1}; planes[n++] = {p->skyB[set].p, 4}; planes[n++] = {p->skyC[set].p, 8}; }
    else if (p->kind == ZR_PASS_DENOISE)
    {
        // INPUT: what the temporal step of a tile reads in its apron -- this frame's signal (the bound input plane: its owner shaded it) and the
        // history the previous frame left (colour + length, moments); ITER: the plane the next a-trous iteration reads
        if (which == ZR_HALO_DENOISE_INPUT && p->compIn[3])
        { planes[n++] = {(void*)p->compIn[3], 16}; planes[n++] = {p->svgfHist.p, 16}; planes[n++] = {p->svgfMoments[p->svgfMomIdx].p, 8}; }
        else if (which == ZR_HALO_DENOISE_ITER) planes[n++] = {p->svgfCur, 16};
    }
    size_t b = 0;
    for (int i = 0; i < n; i++) b += planes[i].bpp;
    *bytesPerPixel = b;
    return n;
}
int zr_pass_halo_bytes_per_pixel(zr_pass* p, uint32_t* bytes)
{
    if (!p || !bytes) return Fail(ZR_ERR_INVALID_ARG, "null argument");
    HaloPlane pl[8]; size_t b = 0;
    if (p->initialized && p->kind == ZR_PASS_DENOISE) { *bytes = 40; return ZR_OK; }      // the larger of its two exchanges (ZR_HALO_DENOISE_INPUT; _ITER moves 16)
    if (!p->initialized || !HaloPlanes(p, ZR_HALO_FINAL, pl, &b)) return Fail(ZR_ERR_NOT_INITIALIZED, "pass has no reservoir planes to exchange (or is not initialised)");
    *bytes = (uint32_t)b;
    return ZR_OK;
}
static int HaloCopy(zr_pass* p, hipStream_t s, const zr_gbuffer* gb, int which, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, void* packed, size_t bytes, bool pack)
{
    if (!p || !gb || !packed) return Fail(ZR_ERR_INVALID_ARG, "null argument");
    HaloPlane planes[8]; size_t bpp = 0;
    const int np = p->initialized ? HaloPlanes(p, which, planes, &bpp) : 0;
    if (!np) return Fail(ZR_ERR_NOT_INITIALIZED, "pass has no reservoir planes to exchange (or is not initialised)");
    if (x0 < gb->x0 || y0 < gb->y0 || x0 + w > gb->x0 + gb->w || y0 + h > gb->y0 + gb->h) return Fail(ZR_ERR_INVALID_ARG, "halo rect lies outside this device's planes");
    const size_t n = (size_t)w * h;
    if (bytes != n * bpp) return Fail(ZR_ERR_INVALID_ARG, "halo buffer must hold %zu bytes", n * bpp);
    if (!n) return ZR_OK;
    HIP_TRY(hipSetDevice(p->device));
    char* cursor = (char*)packed;
    const size_t first = (size_t)(y0 - gb->y0) * gb->w + (x0 - gb->x0);
    for (int i = 0; i < np; i++)
    {
        const HaloPlane& pl = planes[i];
        char* tile = (char*)pl.base + first * pl.bpp;
        if (pack) HIP_TRY(hipMemcpy2DAsync(cursor, w * pl.bpp, tile, gb->w * pl.bpp, w * pl.bpp, h, hipMemcpyDeviceToDevice, s));
        else HIP_TRY(hipMemcpy2DAsync(tile, gb->w * pl.bpp, cursor, w * pl.bpp, w * pl.bpp, h, hipMemcpyDeviceToDevice, s));
        cursor += n * pl.bpp;
    }
    return ZR_OK;
}
static int HaloAll(zr_pass* p, hipStream_t s, const zr_gbuffer* gb, int which, const zr_halo_rect* rects, uint32_t n, void* buf, size_t bytes, bool pack)
{
    if (!p || !gb || (n && (!rects || !buf))) return Fail(ZR_ERR_INVALID_ARG, "null argument");
    if (n > ZR_HALO_MAX_RECTS) return Fail(ZR_ERR_INVALID_ARG, "at most %d rects per fused halo transfer", ZR_HALO_MAX_RECTS);
    HaloPlane planes[8]; size_t bpp = 0;
    const int np = p->initialized ? HaloPlanes(p, which, planes, &bpp) : 0;
    if (!np) return Fail(ZR_ERR_NOT_INITIALIZED, "pass has no reservoir planes to exchange (or is not initialised)");
    if (!n) return ZR_OK;
    HaloJob J; std::memset(&J, 0, sizeof(J));
    J.numPlanes = (uint32_t)np; J.numRects = n; J.pitch = gb->w; J.buf = (char*)buf;
    for (int i = 0; i < np; i++) { J.planes[i].base = (char*)planes[i].base; J.planes[i].bpp = (uint32_t)planes[i].bpp; }
    uint32_t maxPx = 0;
    for (uint32_t i = 0; i < n; i++)
    {
        const zr_halo_rect& r = rects[i];
        if (r.x0 < gb->x0 || r.y0 < gb->y0 || r.x0 + r.w > gb->x0 + gb->w || r.y0 + r.h > gb->y0 + gb->h) return Fail(ZR_ERR_INVALID_ARG, "halo rect %u lies outside this device's planes", i);
        const size_t blk = (size_t)r.w * r.h * bpp;
        if ((r.offset & 15u) || r.offset + blk > bytes) return Fail(ZR_ERR_INVALID_ARG, "halo rect %u: block [%llu, +%zu) must be 16-byte aligned and inside the %zu-byte buffer", i, (unsigned long long)r.offset, blk, bytes);
        // planes are copied with accesses of their own width: a block's plane sections start at sums of w * h * bpp_k (bpp 1 .. 16), which are
        // aligned for every plane when w * h is a multiple of 8 -- strips of 32-px aligned tiles with a 32-px apron always are
        if (((size_t)r.w * r.h) & 7u) return Fail(ZR_ERR_INVALID_ARG, "halo rect %u: w * h must be a multiple of 8 pixels", i);
        J.rects[i].first = (r.y0 - gb->y0) * gb->w + (r.x0 - gb->x0); J.rects[i].w = r.w; J.rects[i].h = r.h; J.rects[i].offset = r.offset;
        maxPx = std::max(maxPx, r.w * r.h);
    }
    HIP_TRY(hipSetDevice(p->device));
    const dim3 grid(std::min<uint32_t>((maxPx + 255) / 256, 1024u), n);
    if (pack) hipLaunchKernelGGL(k_halo<true>, grid, dim3(256), 0, s, J);
    else hipLaunchKernelGGL(k_halo<false>, grid, dim3(256), 0, s, J);
    HIP_TRY(hipGetLastError());
    return ZR_OK;
}
int zr_pass_halo_pack_all(zr_pass* p, void* stream, const zr_gbuffer* gb, int which, const zr_halo_rect* rects, uint32_t n, void* dev_buf, size_t bytes)
{ return HaloAll(p, (hipStream_t)stream, gb, which, rects, n, dev_buf, bytes, true); }
int zr_pass_halo_unpack_all(zr_pass* p, void* stream, const zr_gbuffer* gb, int which, const zr_halo_rect* rects, uint32_t n, const void* dev_buf, size_t bytes)
{ return HaloAll(p, (hipStream_t)stream, gb, which, rects, n, const_cast<void*>(dev_buf), bytes, false); }

int zr_pass_halo_pack(zr_pass* p, void* stream, const zr_gbuffer* gb, int which, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, void* dev_dst, size_t bytes)
{ return HaloCopy(p, (hipStream_t)stream, gb, which, x0, y0, w, h, dev_dst, bytes, true); }
int zr_pass_halo_unpack(zr_pass* p, void* stream, const zr_gbuffer* gb, int which, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, const void* dev_src, size_t bytes)
{ return HaloCopy(p, (hipStream_t)stream, gb, which, x0, y0, w, h, const_cast<void*>(dev_src), bytes, false); }

static int RenderStageInner(zr_pass* p, void* stream, const zr_frame_constants* cb, const zr_scene* sc, zr_gbuffer* gb, int stages);
int zr_pass_render_stage(zr_pass* p, void* stream, const zr_frame_constants* cb, const zr_scene* sc, zr_gbuffer* gb, int stages)
{
    if (!p || !cb || !sc) return Fail(ZR_ERR_INVALID_ARG, "zr_pass_render: null argument");
    // stream-ordered scene updates: wait (on the device) for an update enqueued on another stream, and leave an event behind this
    // render for the next update to wait for.  Both are no-ops until the scene has been updated once.
    HIP_TRY(hipSetDevice(p->device));
    int r = SceneAcquireForRender(sc, (hipStream_t)stream);
    if (r) return r;
    // tracked G-buffers (zr_pass_set_frame_overlap): the ReSTIR PT pass orders its stages itself, finer than a whole call; every other reader waits for
    // the planes' GBUFFER render when that ran on another stream and leaves an event behind its reads of both plane sets
    const bool rptPass = p->kind == ZR_PASS_INDIRECT && p->integrator == ZR_INTEGRATOR_RESTIR_PT;
    const bool trackedReader = gb && gb->tracked && p->kind != ZR_PASS_GBUFFER && !rptPass;
    if (trackedReader) { if ((r = GBufferAcquireRead(gb, (hipStream_t)stream))) return r; }
    r = RenderStageInner(p, stream, cb, sc, gb, stages);
    if (r) return r;
    if (trackedReader) { if ((r = GBufferMarkRead(gb, gb->cur, (hipStream_t)stream)) || (r = GBufferMarkRead(gb, gb->Prev(), (hipStream_t)stream))) return r; }
    return SceneReleaseAfterRender(sc, (hipStream_t)stream);
}
static int RenderStageInner(zr_pass* p, void* stream, const zr_frame_constants* cb, const zr_scene* sc, zr_gbuffer* gb, int stages)
{
    if (!p || !cb || !sc) return Fail(ZR_ERR_INVALID_ARG, "zr_pass_render: null argument");
    if (!(stages & (ZR_STAGE_ALL | ZR_STAGE_SPATIAL2 | ZR_STAGE_CANDIDATES | ZR_STAGE_TEMPORAL_REUSE | (p && p->kind == ZR_PASS_DENOISE ? ZR_STAGE_DENOISE_MASK : 0)))) return Fail(ZR_ERR_INVALID_ARG, "no stage selected");
    if (!p->initialized) return Fail(ZR_ERR_NOT_INITIALIZED, "pass not initialised (zr_pass_init)");
    if (sc->device != p->device || (gb && gb->device != p->device)) return Fail(ZR_ERR_INVALID_ARG, "scene / gbuffer / pass live on different devices");
    HIP_TRY(hipSetDevice(p->device));
    hipStream_t s = (hipStream_t)stream;
    if ((stages & (ZR_STAGE_TEMPORAL | ZR_STAGE_CANDIDATES)) || (p->kind == ZR_PASS_DENOISE && (stages & (ZR_STAGE_SPATIAL | ZR_STAGE_DENOISE_TEMPORAL)))) p->numTimers = 0;      // timings accumulate over the stages of one frame
    {
        const uint32_t off[4] = { cb->base_color_maps_desc_heap_offset, cb->normal_maps_desc_heap_offset,
                                  cb->metallic_roughness_maps_desc_heap_offset, cb->emissive_maps_desc_heap_offset };
        for (int k = 0; k < 4; k++)
            if (sc->maxTex[k] >= 0 && (uint64_t)off[k] + (uint64_t)sc->maxTex[k] >= sc->view.tex.count)
                return Fail(ZR_ERR_INVALID_ARG, "texture table %d: offset %u + index %lld lies outside the scene's %u textures", k, off[k], (long long)sc->maxTex[k], sc->view.tex.count);
    }
    switch (p->kind)
    {
    case ZR_PASS_GBUFFER: return (stages & ZR_STAGE_TEMPORAL) ? RenderGBuffer(p, s, cb, sc, gb) : ZR_OK;
    case ZR_PASS_PRELIGHTING: return (stages & ZR_STAGE_TEMPORAL) ? RenderPreLighting(p, s, cb, const_cast<zr_scene*>(sc)) : ZR_OK;
    case ZR_PASS_INDIRECT: return RenderIndirect(p, s, cb, sc, gb, stages);
    case ZR_PASS_DI_EMISSIVE: return RenderDirectEmissive(p, s, cb, sc, gb, stages);
    case ZR_PASS_DI_SKY: return RenderDirectSky(p, s, cb, sc, gb, stages);
    case ZR_PASS_COMPOSITING: return (stages & ZR_STAGE_SPATIAL) ? RenderCompositing(p, s, cb, sc, gb) : ZR_OK;
    case ZR_PASS_SKY: return (stages & ZR_STAGE_TEMPORAL) ? RenderSky(p, s, cb, const_cast<zr_scene*>(sc)) : ZR_OK;
    case ZR_PASS_TAA: return (stages & ZR_STAGE_SPATIAL) ? RenderTAA(p, s, cb, gb) : ZR_OK;
    case ZR_PASS_AUTO_EXPOSURE: return (stages & ZR_STAGE_SPATIAL) ? RenderAutoExposure(p, s, cb) : ZR_OK;
    case ZR_PASS_DISPLAY: return (stages & ZR_STAGE_SPATIAL) ? RenderDisplay(p, s, cb) : ZR_OK;
    case ZR_PASS_DENOISE:
    {   // ZR_STAGE_SPATIAL (what zr_pass_render passes): the whole pass; ZR_STAGE_DENOISE_* bits: the steps of a tile's schedule
        const uint32_t steps = ((uint32_t)stages & ZR_STAGE_DENOISE_MASK) | ((stages & ZR_STAGE_SPATIAL) ? (uint32_t)ZR_STAGE_DENOISE_MASK : 0u);
        return steps ? RenderDenoise(p, s, cb, gb, steps) : ZR_OK;
    }
    default: return Fail(ZR_ERR_UNSUPPORTED, "pass kind %d not implemented", p->kind);
    }
}

int zr_pass_get_output(const zr_pass* p, int which, void** dev, uint32_t* w, uint32_t* h, uint32_t* bpp)
{
    if (!p || !dev) return Fail(ZR_ERR_INVALID_ARG, "null argument");
    if (!p->initialized) return Fail(ZR_ERR_NOT_INITIALIZED, "pass not initialised");
    if (p->kind == ZR_PASS_SKY)
    {
        if (which != ZR_OUT_SKY_LUT) return Fail(ZR_ERR_INVALID_ARG, "pass has no such output");
        *dev = p->skyLut.p;
        if (w) *w = p->w;
        if (h) *h = p->h;
        if (bpp) *bpp = 4;
        return ZR_OK;
    }
    if (p->kind == ZR_PASS_AUTO_EXPOSURE)
    {
        if (which == ZR_OUT_EXPOSURE) { *dev = p->aeExposure.p; if (w) *w = 1; if (h) *h = 1; if (bpp) *bpp = 8; return ZR_OK; }
        if (which == ZR_OUT_AE_HISTOGRAM) { *dev = p->aeHist.p; if (w) *w = post::kHistBins; if (h) *h = 1; if (bpp) *bpp = 4; return ZR_OK; }
        return Fail(ZR_ERR_INVALID_ARG, "pass has no such output");
    }
    if (p->kind == ZR_PASS_DISPLAY)
    {
        if (which == ZR_OUT_DISPLAY) { *dev = p->displayOut.p; if (w) *w = p->w; if (h) *h = p->h; if (bpp) *bpp = 16; return ZR_OK; }
        if (which == ZR_OUT_DISPLAY_SRGB8) { *dev = p->displaySrgb.p; if (w) *w = p->w; if (h) *h = p->h; if (bpp) *bpp = 4; return ZR_OK; }
        return Fail(ZR_ERR_INVALID_ARG, "pass has no such output");
    }
    if (p->kind == ZR_PASS_DENOISE)
    {
        if (w) *w = p->w;
        if (h) *h = p->h;
        if (which == ZR_OUT_DENOISED) { *dev = (void*)p->svgfOut; if (bpp) *bpp = 16; return ZR_OK; }
        if (which == ZR_OUT_DENOISE_HISTORY) { *dev = p->svgfHist.p; if (bpp) *bpp = 16; return ZR_OK; }
        if (which == ZR_OUT_DENOISE_MOMENTS) { *dev = p->svgfMoments[p->svgfMomIdx].p; if (bpp) *bpp = 8; return ZR_OK; }
        return Fail(ZR_ERR_INVALID_ARG, "pass has no such output");
    }
    if (p->kind == ZR_PASS_TAA)
    {
        if (which != ZR_OUT_TAA) return Fail(ZR_ERR_INVALID_ARG, "pass has no such output");
        *dev = p->taaOut[p->taaIdx].p;
        if (w) *w = p->w;
        if (h) *h = p->h;
        if (bpp) *bpp = 8;
        return ZR_OK;
    }
    if (p->kind == ZR_PASS_COMPOSITING)
    {
        if (which != ZR_OUT_FINAL) return Fail(ZR_ERR_INVALID_ARG, "pass has no such output");
        *dev = p->finalRGBA.p;
        if (w) *w = p->w;
        if (h) *h = p->h;
        if (bpp) *bpp = 16;
        return ZR_OK;
    }
    if (p->kind == ZR_PASS_DI_SKY)
    {
        uint32_t b = 16;
        const int last = 1 - p->currIdx;
        if (which == ZR_OUT_FINAL) *dev = p->finalRGBA.p;
        else if (which == ZR_OUT_SDI_RESERVOIR_A) { *dev = p->skyA[last].p; b = 1; }
        else if (which == ZR_OUT_SDI_RESERVOIR_B) { *dev = p->skyB[last].p; b = 4; }
        else if (which == ZR_OUT_SDI_RESERVOIR_C) { *dev = p->skyC[last].p; b = 8; }
        else if (which == ZR_OUT_SDI_TARGET) *dev = p->diTarget.p;
        else return Fail(ZR_ERR_INVALID_ARG, "pass has no such output");
        if (w) *w = p->w;
        if (h) *h = p->h;
        if (bpp) *bpp = b;
        return ZR_OK;
    }
    if (p->kind == ZR_PASS_DI_EMISSIVE)
    {
        uint32_t b = 16;
        const int last = 1 - p->currIdx;        // the reservoir set written by the last frame
        if (which == ZR_OUT_FINAL) *dev = p->finalRGBA.p;
        else if (which == ZR_OUT_RDI_RESERVOIR_A) *dev = p->diA[last].p;
        else if (which == ZR_OUT_RDI_RESERVOIR_B) { *dev = p->diB[last].p; b = 8; }
        else if (which == ZR_OUT_RDI_TARGET) *dev = p->diTarget.p;
        else return Fail(ZR_ERR_INVALID_ARG, "pass has no such output");
        if (w) *w = p->w;
        if (h) *h = p->h;
        if (bpp) *bpp = b;
        return ZR_OK;
    }
    if (p->kind != ZR_PASS_INDIRECT) return Fail(ZR_ERR_INVALID_ARG, "pass has no such output");
    uint32_t bytes = 16;
    if (which == ZR_OUT_FINAL) *dev = p->Final(p->finOut);
    else if (which >= ZR_OUT_RGI_RESERVOIR_A && which <= ZR_OUT_RGI_RESERVOIR_C && p->integrator == ZR_INTEGRATOR_RESTIR_GI)
    {
        const int last = 1 - p->currIdx;
        if (which == ZR_OUT_RGI_RESERVOIR_A) *dev = p->giA[last].p;
        else if (which == ZR_OUT_RGI_RESERVOIR_B) { *dev = p->giB[last].p; bytes = 8; }
        else *dev = p->giC[last].p;
    }
    else if (which >= ZR_OUT_RPT_RBUF_CTN_A && which <= ZR_OUT_RPT_RBUF_NTC_D && p->integrator == ZR_INTEGRATOR_RESTIR_PT)
    {
        const zr_pass::RBufStorage& B = p->rb[(which - ZR_OUT_RPT_RBUF_CTN_A) / 4];
        switch ((which - ZR_OUT_RPT_RBUF_CTN_A) % 4)
        {
        case 0: *dev = B.A.p; bytes = 8; break;
        case 1: *dev = B.B.p; bytes = 16; break;
        case 2: *dev = B.C.p; bytes = 16; break;
        default: *dev = B.D.p; bytes = 2; break;
        }
    }
    else if ((which == ZR_OUT_RPT_THREAD_MAP_CTN || which == ZR_OUT_RPT_THREAD_MAP_NTC) && p->integrator == ZR_INTEGRATOR_RESTIR_PT)
    { *dev = p->rptMap[which - ZR_OUT_RPT_THREAD_MAP_CTN].p; bytes = 2; }
    else if (which >= ZR_OUT_RPT_RESERVOIR_A && which <= ZR_OUT_RPT_NEIGHBOR && p->integrator == ZR_INTEGRATOR_RESTIR_PT)
    {
        const zr_pass::ResStorage& R = p->RptOth();      // the set the next frame reads as "previous"
        switch (which)
        {
        case ZR_OUT_RPT_RESERVOIR_A: *dev = R.A.p; bytes = 4; break;
        case ZR_OUT_RPT_RESERVOIR_B: *dev = R.B.p; bytes = 8; break;
        case ZR_OUT_RPT_RESERVOIR_C: *dev = R.C.p; bytes = 16; break;
        case ZR_OUT_RPT_RESERVOIR_D: *dev = R.D.p; bytes = 16; break;
        case ZR_OUT_RPT_RESERVOIR_E: *dev = R.E.p; bytes = 2; break;
        case ZR_OUT_RPT_RESERVOIR_F: *dev = R.F.p; bytes = 8; break;
        case ZR_OUT_RPT_RESERVOIR_G: *dev = R.G.p; bytes = 8; break;
        case ZR_OUT_RPT_TARGET: *dev = p->Target(); bytes = 16; break;
        default: *dev = p->rptNeighbor.p; bytes = 2; break;
        }
    }
    else return Fail(ZR_ERR_INVALID_ARG, "pass has no such output");
    if (w) *w = p->w;
    if (h) *h = p->h;
    if (bpp) *bpp = bytes;
    return ZR_OK;
}
int zr_pass_download_output(const zr_pass* p, int which, void* stream, void* dst, size_t bytes)
{
    void* dev = nullptr; uint32_t w, h, bpp;
    int r = zr_pass_get_output(p, which, &dev, &w, &h, &bpp);
    if (r) return r;
    if (!dst || bytes != (size_t)w * h * bpp) return Fail(ZR_ERR_INVALID_ARG, "destination must hold %zu bytes", (size_t)w * h * bpp);
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    HIP_TRY(hipMemcpy(dst, dev, bytes, hipMemcpyDeviceToHost));
    return ZR_OK;
}
// Cost map of a ReSTIR PT pass: wave lifetimes (shader cycles / 16) of K11 / K14 / K16 per 32 x 32-px cell of the pass's planes (cell (0, 0) at the
// plane origin) since the last reset -- the load signal of the cost-balanced screen split (zetaray_amd/tiling.py balanced_layout).  One atomic per wave while enabled.
int zr_pass_enable_cost_map(zr_pass* p, int enable)
{
    if (!p || !p->initialized) return Fail(ZR_ERR_NOT_INITIALIZED, "pass not initialised");
    if (p->kind != ZR_PASS_INDIRECT || p->integrator != ZR_INTEGRATOR_RESTIR_PT) return Fail(ZR_ERR_UNSUPPORTED, "the cost map is an output of the ReSTIR PT pass");
    p->costOn = enable != 0;
    p->costRays = enable == ZR_COST_MAP_RAYS;
    return ZR_OK;
}
int zr_scene_material_class(const zr_scene* s, uint32_t* out)
{
    if (!s || !out) return Fail(ZR_ERR_INVALID_ARG, "zr_scene_material_class: null argument");
    *out = s->plainMaterials.load() ? 1u : 0u;
    return ZR_OK;
}
int zr_debug_set_material_class_kernels(int enable) { g_materialClassKernels.store(enable != 0, std::memory_order_relaxed); return ZR_OK; }
int zr_debug_set_bvh_depth_cap(uint32_t levels) { DeviceBvhSetDepthCap(levels); return ZR_OK; }
int zr_debug_set_large_scene_nodes(uint32_t n) { g_largeSceneNodes.store(n ? n : kLargeSceneNodes, std::memory_order_relaxed); return ZR_OK; }
int zr_pass_debug_trip_stats(zr_pass* p, uint64_t out[3])
{
    if (!p || !out) return Fail(ZR_ERR_INVALID_ARG, "null argument");
    out[0] = out[1] = out[2] = 0;
    if (!p->tripStats.p) return ZR_OK;
    HIP_TRY(hipDeviceSynchronize());
    unsigned long long h[4];
    HIP_TRY(hipMemcpy(h, p->tripStats.p, 32, hipMemcpyDeviceToHost));
    out[0] = h[0]; out[1] = h[1]; out[2] = h[2];
    return ZR_OK;
}
int zr_pass_read_cost_map(zr_pass* p, void* stream, uint32_t* out, uint32_t cells_w, uint32_t cells_h, int reset)
{
    if (!p || !out || !p->initialized) return Fail(ZR_ERR_INVALID_ARG, "zr_pass_read_cost_map: bad argument");
    const uint32_t cw = (p->w + 31u) / 32u + 1u, ch = (p->h + 31u) / 32u + 1u;
    if (!p->costMap.p || cells_w != cw || cells_h != ch) return Fail(ZR_ERR_INVALID_ARG, "the cost map is %u x %u cells", cw, ch);
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipMemcpyAsync(out, p->costMap.p, (size_t)cw * ch * 4, hipMemcpyDeviceToHost, (hipStream_t)stream));
    if (reset) HIP_TRY(hipMemsetAsync(p->costMap.p, 0, (size_t)cw * ch * 4, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return ZR_OK;
}
int zr_pass_read_counters(zr_pass* p, void* stream, zr_counters* out, int reset)
{
    if (!p || !out) return Fail(ZR_ERR_INVALID_ARG, "null argument");
    HIP_TRY(hipSetDevice(p->device));
    out->n_closest = p->hostCounters.n_closest; out->n_shadow = p->hostCounters.n_shadow;
    if ((p->kind == ZR_PASS_INDIRECT || p->kind == ZR_PASS_DI_EMISSIVE || p->kind == ZR_PASS_DI_SKY) && p->initialized)
    {
        unsigned long long c[2 * kCounterSlots];
        HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
        if (p->haveCand) HIP_TRY(hipStreamSynchronize(p->candStream));      // (frame overlap: K11 may have run on another stream)
        HIP_TRY(hipMemcpy(c, p->counters.p, sizeof(c), hipMemcpyDeviceToHost));
        for (int i = 0; i < kCounterSlots; i++) { out->n_closest += c[2 * i]; out->n_shadow += c[2 * i + 1]; }
        if (reset) { HIP_TRY(hipMemsetAsync(p->counters.p, 0, sizeof(c), (hipStream_t)stream)); HIP_TRY(hipStreamSynchronize((hipStream_t)stream)); }
    }
    if (reset) { p->hostCounters.n_closest = 0; p->hostCounters.n_shadow = 0; }
    return ZR_OK;
}
int zr_pass_read_kernel_counters(zr_pass* p, void* stream, uint32_t max_entries, const char** names, uint64_t* closest, uint64_t* shadow,
    uint32_t* count)
{
    if (!p || !names || !closest || !shadow || !count) return Fail(ZR_ERR_INVALID_ARG, "null argument");
    if ((p->kind != ZR_PASS_INDIRECT && p->kind != ZR_PASS_DI_EMISSIVE && p->kind != ZR_PASS_DI_SKY) || !p->initialized) return Fail(ZR_ERR_NOT_INITIALIZED, "pass has no ray counters or is not initialised");
    HIP_TRY(hipSetDevice(p->device));
    unsigned long long c[2 * kCounterSlots];
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    if (p->haveCand) HIP_TRY(hipStreamSynchronize(p->candStream));
    HIP_TRY(hipMemcpy(c, p->counters.p, sizeof(c), hipMemcpyDeviceToHost));
    uint32_t n = 0;
    for (int i = 0; i < kCounterSlots && n < max_entries; i++)
    {
        if (!kCounterNames[i][0] || (c[2 * i] == 0 && c[2 * i + 1] == 0)) continue;
        names[n] = kCounterNames[i]; closest[n] = c[2 * i]; shadow[n] = c[2 * i + 1]; n++;
    }
    *count = n;
    return ZR_OK;
}
int zr_pass_enable_timing(zr_pass* p, int enable)
{
    if (!p) return Fail(ZR_ERR_INVALID_ARG, "null pass");
    p->timing = enable != 0;
    return ZR_OK;
}
int zr_pass_get_timings(zr_pass* p, uint32_t max_entries, const char** names, float* ms, uint32_t* launches, uint32_t* count)
{
    if (!p || !count) return Fail(ZR_ERR_INVALID_ARG, "null argument");
    HIP_TRY(hipSetDevice(p->device));
    p->timerNames.clear(); p->timerMs.clear(); p->timerLaunches.clear();
    for (int i = 0; i < p->numTimers; i++)
    {
        zr_pass::Timer& t = p->timers[i];
        HIP_TRY(hipEventSynchronize(t.b));
        float dt = 0;
        HIP_TRY(hipEventElapsedTime(&dt, t.a, t.b));
        size_t k = 0;
        for (; k < p->timerNames.size(); k++) if (p->timerNames[k] == t.name) break;
        if (k == p->timerNames.size()) { p->timerNames.push_back(t.name); p->timerMs.push_back(0); p->timerLaunches.push_back(0); }
        p->timerMs[k] += dt; p->timerLaunches[k]++;
    }
    *count = (uint32_t)p->timerNames.size();
    for (uint32_t k = 0; k < *count && k < max_entries; k++)
    {
        if (names) names[k] = p->timerNames[k].c_str();
        if (ms) ms[k] = p->timerMs[k];
        if (launches) launches[k] = p->timerLaunches[k];
    }
    return ZR_OK;
}
int zr_pass_destroy(zr_pass* p)
{
    if (!p) return ZR_OK;
    (void)hipSetDevice(p->device);
    for (auto& t : p->timers) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
    if (p->powerEv) (void)hipEventDestroy(p->powerEv);
    if (p->powerHost) (void)hipHostFree(p->powerHost);
    if (p->overlapStream) { (void)hipStreamSynchronize(p->overlapStream); (void)hipStreamDestroy(p->overlapStream); }
    if (p->evCand) (void)hipEventDestroy(p->evCand);
    if (p->evTemporal) (void)hipEventDestroy(p->evTemporal);
    for (auto& e : p->evDone) if (e) (void)hipEventDestroy(e);
    delete p;
    return ZR_OK;
}

int zr_trace_closest(const zr_scene* sc, void* stream, const float* d_rays, uint32_t n, uint32_t mask, uint32_t* d_hits)
{
    if (!sc || !d_rays || !d_hits) return Fail(ZR_ERR_INVALID_ARG, "null argument");
    if (n == 0) return ZR_OK;
    HIP_TRY(hipSetDevice(sc->device));
    const uint32_t grid = std::min<uint32_t>((n + kBlock - 1) / kBlock, 8192u);
    hipLaunchKernelGGL(k_trace_rays, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, FrameView(sc, nullptr), (const F4*)d_rays, n, mask, (U4*)d_hits);
    HIP_TRY(hipGetLastError());
    return ZR_OK;
}
int zr_trace_any(const zr_scene* sc, void* stream, const float* d_rays, uint32_t n, uint32_t mask, uint32_t* d_occ)
{
    if (!sc || !d_rays || !d_occ) return Fail(ZR_ERR_INVALID_ARG, "null argument");
    if (n == 0) return ZR_OK;
    HIP_TRY(hipSetDevice(sc->device));
    const uint32_t grid = std::min<uint32_t>((n + kBlock - 1) / kBlock, 8192u);
    hipLaunchKernelGGL(k_trace_rays_any, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, FrameView(sc, nullptr), (const F4*)d_rays, n, mask, d_occ);
    HIP_TRY(hipGetLastError());
    return ZR_OK;
}

} // extern "C"
