// zr_tu_bvh.hip -- the acceleration-structure BUILD on the device (round 3).
//
// The reference builds its BLAS / TLAS with D3D12 driver calls on the GPU (RtAccelerationStructure.cpp:121-200 StaticBLAS::Rebuild, :708-789
// TLAS::Render).  Here: a linear BVH over the world-space triangles of all instances -- Morton code of the triangle centroid (15 - 21 bits per axis, k_bvh_keys) with the
// global triangle index as the low ceil(log2 n) key bits (unique keys: no duplicate handling), one 64-bit radix sort (hipCUB / rocPRIM), a breadth-first
// construction of the 4-wide tree straight from the sorted keys (a node = a key range; it is cut up to three times at the highest differing
// key bit, largest piece first, into 2 - 4 children; ranges of <= 2 triangles become leaves), and the per-level box computation + 8-bit
// quantisation that the refit already has (k_refit_level, zr_api.hip), bottom-up.  Breadth-first node allocation makes the nodes of a level
// contiguous, which is exactly what the level-by-level refit wants.  Results of every query are independent of the tree (zr_intersect.h:
// conservative boxes, closest hit with the index tie-break), so a device-built scene renders bit-identically to a host-built one; only the
// traversal cost differs (LBVH against binned SAH).  The host builder (zr_bvh.h) stays the default at zr_scene_create: its trees are better,
// and 0.2 s once per scene is affordable; the device build is what ZR_SCENE_UPDATE=rebuild and ZR_BVH_BUILD=device use.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <vector>
#include "zr_dev_scene.h"
#include "zr_bvh_device.h"

using namespace zr;

namespace {

__device__ __forceinline__ uint32_t FloatOrdered(float f) { const uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float OrderedFloat(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }

// world-space triangle of global index g (the builder's expression, zr_bvh.h Build / k_refit_tris) + its centroid into the scene bounds
__global__ void __launch_bounds__(256) k_bvh_tris(BvhTri* out, uint32_t n, const TriMeta* meta, const zr_mesh_instance* instances, const float* toWorld,
    const zr_vertex* vertices, const uint32_t* indices, const uint8_t* instanceMask, uint32_t* sceneBounds)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    float c[3] = {0, 0, 0};
    const bool in = g < n;
    if (in)
    {
        const TriMeta tm = meta[g];
        const zr_mesh_instance mi = instances[tm.mesh];
        const float* M = toWorld + 12 * (size_t)tm.mesh;
        float w[3][3];
        for (int k = 0; k < 3; k++)
        {
            const uint32_t vi = indices[mi.base_idx_offset + 3 * tm.prim + k] + mi.base_vtx_offset;
            const float* P = vertices[vi].pos;
            for (int r = 0; r < 3; r++) w[k][r] = M[4 * r + 0] * P[0] + M[4 * r + 1] * P[1] + M[4 * r + 2] * P[2] + M[4 * r + 3];
        }
        BvhTri t;
        for (int r = 0; r < 3; r++)
        {
            t.v0[r] = w[0][r]; t.e1[r] = w[1][r] - w[0][r]; t.e2[r] = w[2][r] - w[0][r];
            const float a = t.v0[r], b = t.v0[r] + t.e1[r], cc = t.v0[r] + t.e2[r];
            c[r] = 0.5f * (fminf(a, fminf(b, cc)) + fmaxf(a, fmaxf(b, cc)));
        }
        t.gidx = g; t.mask = instanceMask[tm.mesh]; t.id = TriID(tm.mesh, tm.prim);
        out[g] = t;
    }
    // centroid bounds: wave reduction, then one atomic pair per axis and wave
    for (int r = 0; r < 3; r++)
    {
        float lo = in ? c[r] : 3.402823466e+38f, hi = in ? c[r] : -3.402823466e+38f;
        for (int s = 1; s < 64; s <<= 1) { lo = fminf(lo, __shfl_xor(lo, s)); hi = fmaxf(hi, __shfl_xor(hi, s)); }
        if ((threadIdx.x & 63u) == 0 && lo <= hi) { atomicMin(&sceneBounds[r], FloatOrdered(lo)); atomicMax(&sceneBounds[3 + r], FloatOrdered(hi)); }
    }
}

// key = Morton code of the centroid (B bits per axis, x the most significant of each triple) << idxBits | triangle index.  The index only needs
// ceil(log2 n) bits, the code gets the rest: B = min(21, (64 - idxBits) / 3) -- 15 bits per axis for the 380 k-triangle atrium instead of the 10 of a
// 32-bit code, so clustered geometry is separated by position rather than by index order.
__global__ void __launch_bounds__(256) k_bvh_keys(const BvhTri* tris, uint32_t n, const uint32_t* sceneBounds, unsigned long long* keys, uint32_t idxBits)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    const BvhTri t = tris[g];
    const uint32_t B = (64u - idxBits) / 3u < 21u ? (64u - idxBits) / 3u : 21u;
    unsigned long long q[3];
    for (int r = 0; r < 3; r++)
    {
        const float lo = OrderedFloat(sceneBounds[r]), hi = OrderedFloat(sceneBounds[3 + r]);
        const float a = t.v0[r], b = t.v0[r] + t.e1[r], cc = t.v0[r] + t.e2[r];
        const float c = 0.5f * (fminf(a, fminf(b, cc)) + fmaxf(a, fmaxf(b, cc)));
        const float ext = hi - lo;
        float u = ext > 0 ? (c - lo) / ext : 0.0f;
        u = fminf(fmaxf(u, 0.0f), 1.0f);
        const unsigned long long cells = 1ull << B;
        const unsigned long long v = (unsigned long long)((double)u * (double)cells);
        q[r] = v < cells ? v : cells - 1ull;
    }
    unsigned long long m = 0;
    for (uint32_t bit = 0; bit < B; bit++)
        m |= (((q[0] >> bit) & 1ull) << (3u * bit + 2u)) | (((q[1] >> bit) & 1ull) << (3u * bit + 1u)) | (((q[2] >> bit) & 1ull) << (3u * bit));
    keys[g] = (m << idxBits) | g;
}
__global__ void __launch_bounds__(256) k_bvh_emit(const BvhTri* byGlobal, const unsigned long long* keys, uint32_t n, BvhTri* sorted, uint32_t idxBits)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) sorted[i] = byGlobal[(uint32_t)(keys[i] & ((1ull << idxBits) - 1ull))];
}

// Karras' split of a sorted key range [a, b), b - a >= 2: the position where the highest differing bit flips.  Left = [a, s), right = [s, b).
__device__ __forceinline__ uint32_t FindSplit(const unsigned long long* keys, uint32_t a, uint32_t b)
{
    const unsigned long long first = keys[a], last = keys[b - 1];
    if (first == last) return (a + b) >> 1;
    const int common = __clzll((long long)(first ^ last));
    uint32_t split = a, step = b - 1 - a;
    do
    {
        step = (step + 1) >> 1;
        const uint32_t ns = split + step;
        if (ns < b - 1 && __clzll((long long)(first ^ keys[ns])) > common) split = ns;
    } while (step > 1);
    return split + 1;
}

// one tree level: nodes [ctl[level], ctl[level + 1]) each cut their key range into 2 - 4 children; inner children get the next free node ids
// (so the next level is contiguous too).  ctl[0 .. kMaxLevels + 1] = level starts, ctl[kCtlCount] = node counter.
constexpr uint32_t kMaxLevels = 62, kCtlCount = 64, kMaxLeafTris = 2;
// Depth cap.  An ordered traversal pushes at most three entries per level, and a lane's stack holds kTravStack = 64: a tree may have 21 levels
// (node levels 0 .. kLastLevel).  Morton cuts alone do not promise that -- a geometric cascade of clusters spends one 4-wide level per three key
// bits (21 levels for 63 bits) before the clusters themselves are divided -- and a deeper tree used to be rejected (missing 4).  A
// node whose range could no longer fit below the cap if it were cut unevenly is cut into four equal parts of its (Morton-ordered) range instead:
// need(c) = levels a balanced subtree over c triangles occupies; a Morton child is never larger than its parent, so `level + need(count) >
// kLastLevel` is the last moment to switch, and from there every level is balanced (need drops by one per level).  Any n <= 2 * 4^20 fits.
constexpr uint32_t kLastLevel = 20;
static uint32_t g_lastLevel = kLastLevel;      // (zr_debug_set_bvh_depth_cap: the parity tests lower it to run the balanced cuts on an ordinary scene)
__device__ __forceinline__ uint32_t BalancedLevels(uint32_t c) { uint32_t n = 0; while (c > kMaxLeafTris) { c = (c + 3u) >> 2; n++; } return n; }
__global__ void __launch_bounds__(64) k_bvh_level(const unsigned long long* keys, uint2* ranges, Bvh4Node* nodes, uint32_t* ctl, uint32_t level, uint32_t cap, uint32_t lastLevel)
{
    const uint32_t start = ctl[level], end = ctl[level + 1];
    for (uint32_t node = start + blockIdx.x * blockDim.x + threadIdx.x; ; node += gridDim.x * blockDim.x)
    {
        const bool live = node < end;
        uint32_t sa[4], sb[4]; int k = 0;
        if (live)
        {
            const uint2 rg = ranges[node];
            sa[0] = rg.x; sb[0] = rg.y; k = 1;
            const uint32_t cnt = rg.y - rg.x;
            if (level + BalancedLevels(cnt) > lastLevel)
            {   // equal parts (the depth cap above); parts are never empty: cnt > kMaxLeafTris >= 2, and 4 parts of >= 3 triangles ...
                k = 0;
                for (uint32_t i = 0; i < 4u; i++)
                {
                    const uint32_t a = rg.x + (uint32_t)(((unsigned long long)cnt * i) >> 2), b = rg.x + (uint32_t)(((unsigned long long)cnt * (i + 1u)) >> 2);
                    if (b > a) { sa[k] = a; sb[k] = b; k++; }      // ... (3 triangles: one part stays empty and is skipped)
                }
            }
            else while (k < 4)
            {
                int pick = -1; uint32_t best = kMaxLeafTris;
                for (int i = 0; i < k; i++) { const uint32_t c = sb[i] - sa[i]; if (c > best) { best = c; pick = i; } }
                if (pick < 0) break;
                const uint32_t s = FindSplit(keys, sa[pick], sb[pick]);
                sa[k] = s; sb[k] = sb[pick]; sb[pick] = s; k++;
            }
        }
        // inner children: consecutive ids from one atomic per wave
        uint32_t nInner = 0;
        for (int i = 0; i < k; i++) nInner += (sb[i] - sa[i]) > kMaxLeafTris ? 1u : 0u;
        uint32_t pre = nInner;      // inclusive wave scan
        for (int s = 1; s < 64; s <<= 1) { const uint32_t o = __shfl_up(pre, s); if ((threadIdx.x & 63u) >= (uint32_t)s) pre += o; }
        const uint32_t total = __shfl(pre, 63);
        uint32_t base = 0;
        if ((threadIdx.x & 63u) == 63u && total) base = atomicAdd(&ctl[kCtlCount], total);
        base = __shfl(base, 63) + pre - nInner;
        if (live)
        {
            Bvh4Node N;
            N.ox = 0; N.oy = 0; N.oz = 0; N.exps = 0; N.qlox = N.qloy = N.qloz = N.qhix = N.qhiy = N.qhiz = 0; N.pad0 = 0; N.pad1 = 0;
            for (int i = 0; i < 4; i++)
            {
                if (i >= k) { N.child[i] = kEmptyChild; continue; }
                const uint32_t c = sb[i] - sa[i];
                if (c <= kMaxLeafTris) N.child[i] = kLeafBit | (sa[i] << 3) | (c - 1u);
                else { const uint32_t id = base++; if (id < cap) ranges[id] = make_uint2(sa[i], sb[i]); N.child[i] = id; }
            }
            nodes[node] = N;
        }
        if (__ballot(node + gridDim.x * blockDim.x < end) == 0) break;      // the whole wave is past the level
    }
}
__global__ void k_bvh_close(uint32_t* ctl, uint32_t level) { ctl[level + 2] = ctl[kCtlCount]; }

} // namespace

namespace zr {

struct DeviceBvhScratch::Impl
{
    void* keys[2] = {nullptr, nullptr}; void* byGlobal = nullptr; void* ranges = nullptr; void* ctl = nullptr; void* sortTemp = nullptr;
    size_t cap = 0, sortBytes = 0;
    ~Impl() { for (void* p : {keys[0], keys[1], byGlobal, ranges, ctl, sortTemp}) if (p) (void)hipFree(p); }
};
DeviceBvhScratch::DeviceBvhScratch() : impl(new Impl()) {}
DeviceBvhScratch::~DeviceBvhScratch() { delete impl; }

#define BVH_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { err = hipGetErrorString(e_); return -1; } } while (0)

void DeviceBvhSetDepthCap(uint32_t levels) { g_lastLevel = (levels >= 2u && levels <= kLastLevel + 1u) ? levels - 1u : kLastLevel; }
int DeviceBuildBvh4(hipStream_t st, DeviceBvhScratch& S, const DeviceBvhInputs& in, DeviceBvhOutputs& out, std::string& err)
{
    const uint32_t n = in.numTris;
    DeviceBvhScratch::Impl& I = *S.impl;
    if (I.cap < n)
    {
        for (void** p : {&I.keys[0], &I.keys[1], &I.byGlobal, &I.ranges, &I.ctl, &I.sortTemp}) if (*p) { (void)hipFree(*p); *p = nullptr; }
        BVH_TRY(hipMalloc(&I.keys[0], (size_t)n * 8)); BVH_TRY(hipMalloc(&I.keys[1], (size_t)n * 8));
        BVH_TRY(hipMalloc(&I.byGlobal, (size_t)n * sizeof(BvhTri))); BVH_TRY(hipMalloc(&I.ranges, (size_t)n * sizeof(uint2)));
        BVH_TRY(hipMalloc(&I.ctl, (kCtlCount + 8) * sizeof(uint32_t)));
        I.sortBytes = 0;
        BVH_TRY(hipcub::DeviceRadixSort::SortKeys(nullptr, I.sortBytes, (const unsigned long long*)I.keys[0], (unsigned long long*)I.keys[1], (int)n, 0, 64, st));
        BVH_TRY(hipMalloc(&I.sortTemp, I.sortBytes ? I.sortBytes : 16));
        I.cap = n;
    }
    uint32_t* ctl = (uint32_t*)I.ctl;
    // ctl[0] = 0, ctl[1] = 1 (the root level), node counter = 1; scene bounds behind the counters: 3 x +inf-ordered, 3 x -inf-ordered
    uint32_t init[kCtlCount + 8];
    for (uint32_t i = 0; i < kCtlCount + 8; i++) init[i] = 0;
    init[1] = 1; init[kCtlCount] = 1;
    for (int r = 0; r < 3; r++) { init[kCtlCount + 1 + r] = 0xffffffffu; init[kCtlCount + 4 + r] = 0u; }
    BVH_TRY(hipMemcpyAsync(ctl, init, sizeof(init), hipMemcpyHostToDevice, st));      // (pageable source: staged before the call returns)
    uint32_t* bounds = ctl + kCtlCount + 1;
    const dim3 grid((n + 255) / 256), block(256);
    hipLaunchKernelGGL(k_bvh_tris, grid, block, 0, st, (BvhTri*)I.byGlobal, n, in.meta, in.instances, in.toWorld, in.vertices, in.indices, in.instanceMask, bounds);
    uint32_t idxBits = 1; while ((1ull << idxBits) < n) idxBits++;
    hipLaunchKernelGGL(k_bvh_keys, grid, block, 0, st, (const BvhTri*)I.byGlobal, n, bounds, (unsigned long long*)I.keys[0], idxBits);
    BVH_TRY(hipcub::DeviceRadixSort::SortKeys(I.sortTemp, I.sortBytes, (const unsigned long long*)I.keys[0], (unsigned long long*)I.keys[1], (int)n, 0, 64, st));
    hipLaunchKernelGGL(k_bvh_emit, grid, block, 0, st, (const BvhTri*)I.byGlobal, (const unsigned long long*)I.keys[1], n, out.tris, idxBits);
    // the root covers every key; levels until one comes out empty (checked on the host afterwards: the launches are unconditional)
    const uint2 rootRange = make_uint2(0u, n);
    BVH_TRY(hipMemcpyAsync(I.ranges, &rootRange, sizeof(rootRange), hipMemcpyHostToDevice, st));
    for (uint32_t l = 0; l < kMaxLevels; l++)
    {
        // a level of a tree over n keys has at most n / 3 + 1 inner nodes; 256 blocks of one wave grid-stride over whatever there is
        hipLaunchKernelGGL(k_bvh_level, dim3(512), dim3(64), 0, st, (const unsigned long long*)I.keys[1], (uint2*)I.ranges, out.nodes, ctl, l, out.nodeCap, g_lastLevel);
        hipLaunchKernelGGL(k_bvh_close, dim3(1), dim3(1), 0, st, ctl, l);
        if (l >= 7 && (l & 3u) == 3u)
        {   // peek every fourth level from level 7 on: stop launching once the tree has ended
            uint32_t host[kCtlCount + 1];
            BVH_TRY(hipMemcpyAsync(host, ctl, sizeof(host), hipMemcpyDeviceToHost, st));
            BVH_TRY(hipStreamSynchronize(st));
            if (host[l + 2] == host[l + 1]) break;
        }
    }
    uint32_t host[kCtlCount + 1];
    BVH_TRY(hipMemcpyAsync(host, ctl, sizeof(host), hipMemcpyDeviceToHost, st));
    BVH_TRY(hipStreamSynchronize(st));
    BVH_TRY(hipGetLastError());
    const uint32_t numNodes = host[kCtlCount];
    if (numNodes > out.nodeCap) { err = "device BVH build: node capacity exceeded"; return -1; }
    uint32_t levels = 0;
    while (levels < kMaxLevels && host[levels + 1] > host[levels]) levels++;
    if (host[levels] != numNodes) { err = "device BVH build: tree deeper than " + std::to_string(kMaxLevels) + " levels"; return -1; }
    out.numNodes = numNodes; out.numLevels = levels;
    out.stackNeed = 3u * levels;          // <= 3 pushes per level (ordered traversal)
    // refit order: deepest level first; the nodes of a level are contiguous
    out.levelOrder.clear(); out.levelOffsets.clear();
    for (uint32_t l = levels; l-- > 0;)
    {
        out.levelOffsets.push_back((uint32_t)out.levelOrder.size());
        for (uint32_t i = host[l]; i < host[l + 1]; i++) out.levelOrder.push_back(i);
    }
    out.levelOffsets.push_back((uint32_t)out.levelOrder.size());
    return 0;
}

} // namespace zr
