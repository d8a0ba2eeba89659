// zr_bvh.h -- host-side BVH2 builder (binned SAH) over world-space triangles.
//
// Stands in for the reference's acceleration-structure build, which is a D3D12 driver call
// (Source/ZetaCore/RayTracing/RtAccelerationStructure.cpp:121-200 StaticBLAS::Rebuild, :789 TLAS::Render): all static
// mesh instances are flattened into one triangle soup with the instance's float 3x4 transform baked in, exactly what
// the reference's static BLAS holds.  Indexing contract kept from the reference: a hit reports
// meshIdx = GeometryIndex() + InstanceID() and the primitive index within that mesh (RtAccelerationStructure.cpp:393-405).
// Runs once per scene on the host; the device only ever sees the flat node / triangle arrays (zr_dev_scene.h).
#pragma once
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cstring>
#include <cmath>
#include <atomic>
#include <thread>
#include <chrono>
#include <cstdio>
#include "zr_dev_scene.h"

namespace zr {

struct BuiltBvh
{
    std::vector<BvhNode> nodes;      // binned-SAH BVH2 (builder intermediate)
    std::vector<Bvh4Node> nodes4;    // what the device traverses: the BVH2 collapsed to 4-wide nodes
    uint32_t stackNeed = 0;          // traversal stack entries the BVH4 can require (ordered traversal, <= 3 pushes per level)
    std::vector<BvhTri> tris;        // leaf order
    std::vector<TriMeta> meta;       // global order
    uint32_t maxDepth = 0;
};

struct BuildTri { float bmin[3], bmax[3], cent[3]; uint32_t gidx; };

static inline float NextF(float f) { return zr::NextFloat32(f); }
static inline float PrevF(float f) { return zr::PrevFloat32(f); }

class BvhBuilder
{
public:
    static constexpr int kBins = 16;
    // Triangles per leaf.  Measured on MI355X (scripts/gpu_bvh.sh [rounds 1-4: git history up to 31e92fa; today scripts/gpu.sh ab], 1080p): 4 -> 2 triangles cuts the triangle phase of the voted traversal (its lane
    // utilisation is the lowest of the loop) for a few more inner nodes: atrium ReSTIR PT 22.05 -> 19.57 ms, K9 10.69 -> 9.89 ms, Cornell
    // ReSTIR PT 2.51 -> 2.37 ms, ReSTIR GI 1.70 -> 1.51 ms; 1 and 3 are worse than 2, 8 much worse (25.5 ms), SAH leaf termination equals 2.
    static constexpr uint32_t kMaxLeaf = 2;
    static constexpr uint32_t kTinyScene = 8;      // up to this many triangles: one leaf, no nodes
    // experiment knobs (scripts/gpu_bvh.sh [rounds 1-4: git history up to 31e92fa; today scripts/gpu.sh ab]): ZR_BVH_MAX_LEAF = 1..8 triangles per leaf; ZR_BVH_SAH_LEAF = node cost in triangle tests (> 0: a range of
    // <= max-leaf triangles becomes a leaf when splitting it would not pay for the extra node)
    uint32_t maxLeaf_ = kMaxLeaf; float nodeCost_ = 0.0f;
    BvhBuilder()
    {
        if (const char* e = std::getenv("ZR_BVH_MAX_LEAF")) { const int v = std::atoi(e); if (v >= 1 && v <= 8) maxLeaf_ = (uint32_t)v; }
        if (const char* e = std::getenv("ZR_BVH_SAH_LEAF")) nodeCost_ = (float)std::atof(e);
        if (const char* e = std::getenv("ZR_BVH_SWEEP")) sweepBelow_ = (uint32_t)std::atoi(e);
        if (const char* e = std::getenv("ZR_BVH_COLLAPSE")) optimalCollapse_ = std::strcmp(e, "greedy") != 0;
        if (const char* e = std::getenv("ZR_BVH_SPLIT")) medianBelow_ = !std::strcmp(e, "median") ? 0xffffffffu : (uint32_t)std::atoi(e);
        if (const char* e = std::getenv("ZR_BVH_CNODE")) { const float v = (float)std::atof(e); if (v > 0) collapseNodeCost_ = v; }
        // the top of the recursion forks: a range keeps one half and hands the other to a new thread while threads are left (ZR_BVH_THREADS, default
        // = hardware threads, at most 16).  Every range's split depends only on its own triangles, leaves land at their range's position and the
        // BVH2 node numbers only matter as references, so the collapsed tree is the same bit for bit whatever the schedule (380 k-triangle atrium:
        // 192 ms on one thread).
        int nt = (int)std::thread::hardware_concurrency(); if (nt < 1) nt = 1; if (nt > 16) nt = 16;
        if (const char* e = std::getenv("ZR_BVH_THREADS")) { const int v = std::atoi(e); if (v >= 1 && v <= 64) nt = v; }
        threads_ = nt;
    }
    int threads_ = 1;
    uint32_t sweepBelow_ = 0;      // ranges of fewer triangles than this are split by an exact SAH sweep instead of 16 bins
    // BVH2 -> BVH4: which descendants of a binary node become the (up to four) children of its wide node.  true: the SAH-optimal choice for the given binary
    // tree by dynamic programming (Ylitie, Karras, Laine 2017, "Efficient Incoherent Ray Traversal on GPUs Through Compressed Wide BVHs", section 4.1);
    // false (ZR_BVH_COLLAPSE=greedy): rounds 1 - 5's rule -- open the inner child of largest area until there are four.  The greedy rule leaves the bottom of the
    // tree half empty (a 4-triangle range is a node with two 2-triangle leaves and two empty slots: 99.7 k nodes, 2.9 children per node on the 380 k-triangle
    // atrium); measured per ray by tools/bvh_quality.py and in the kernels by the section profiler.  collapseNodeCost_: one node visit (4 box
    // tests, sort, pushes: ~190 VALU instructions) in units of one triangle test (~65).  Results never depend on the tree (zr_intersect.h).
    bool optimalCollapse_ = true; float collapseNodeCost_ = 3.0f;
    uint32_t medianBelow_ = 0;      // (experiment, ZR_BVH_SPLIT=median | N: ranges of fewer than N triangles are split at the object median of the widest centroid axis)

    // ownSubtree (optional, one byte per instance): instances flagged here are kept out of the common SAH tree and get a subtree of their own, joined
    // to the rest near the root -- the flat-tree form of the reference's TLAS over one static BLAS and one BLAS per dynamic instance
    // (RtAccelerationStructure.cpp:121 StaticBLAS::Rebuild, :807 TLAS::BuildDynamicBLASes, :1484 TLAS::RebuildTLAS; an instance that starts to move is
    // converted by SceneCore::ConvertInstanceDynamic, SceneCore.cpp:1038, and TLAS::UpdateFrameMeshInstances_StaticToDynamic, :508).  A rigid motion of such an instance then moves a subtree of triangles that
    // belong together: the device refit (zr_api.hip k_refit_level) keeps its boxes tight, where the same motion inside a common tree inflates every
    // node that mixes the mover's triangles with static ones.  Closest hits and any-hit answers do not depend on the tree, so results are unchanged.
    BuiltBvh Build(const zr_scene_desc& d, const uint8_t* ownSubtree = nullptr)
    {
        BuiltBvh out;
        // ---- flatten instances -> world-space triangles (global order = instance order, then primitive order)
        std::vector<BvhTri> soup;
        for (uint32_t i = 0; i < d.num_instances; i++)
        {
            const zr_mesh_instance& mi = d.instances[i];
            const float* M = d.instance_to_world + 12 * i;
            for (uint32_t p = 0; p < d.instance_num_tris[i]; p++)
            {
                float w[3][3];
                for (int k = 0; k < 3; k++)
                {
                    uint32_t vi = d.indices[mi.base_idx_offset + 3 * p + k] + mi.base_vtx_offset;
                    const float* P = d.vertices[vi].pos;
                    for (int r = 0; r < 3; r++)
                        w[k][r] = M[4 * r + 0] * P[0] + M[4 * r + 1] * P[1] + M[4 * r + 2] * P[2] + M[4 * r + 3];
                }
                BvhTri t;
                for (int r = 0; r < 3; r++) { t.v0[r] = w[0][r]; t.e1[r] = w[1][r] - w[0][r]; t.e2[r] = w[2][r] - w[0][r]; }
                t.gidx = (uint32_t)soup.size(); t.mask = d.instance_mask[i]; t.id = zr::TriID(i, p);
                soup.push_back(t);
                TriMeta m; m.mesh = i; m.prim = p;
                out.meta.push_back(m);
            }
        }
        const uint32_t N = (uint32_t)soup.size();
        bt_.resize(N);
        for (uint32_t i = 0; i < N; i++)
        {
            const BvhTri& t = soup[i];
            for (int r = 0; r < 3; r++)
            {
                float a = t.v0[r], b = t.v0[r] + t.e1[r], c = t.v0[r] + t.e2[r];
                float lo = std::min(a, std::min(b, c)), hi = std::max(a, std::max(b, c));
                bt_[i].bmin[r] = PrevF(lo); bt_[i].bmax[r] = NextF(hi);   // one-ulp pad: v0 + e1 is a rounded v1
                bt_[i].cent[r] = 0.5f * (lo + hi);
            }
            bt_[i].gidx = i;
        }
        if (N <= kTinyScene)
        {
            // tiny scene: a single leaf, no nodes
            out.tris = soup;
            out.maxDepth = 0;
            return out;
        }
        out.nodes.resize(N);          // a BVH2 over N triangles has fewer than N inner nodes; sized once: threads append through nodeCount_
        out.tris.resize(N);           // a leaf's triangles land at its range's position (leaves in depth-first order == ranges left to right)
        soup_ = &soup; out_ = &out;
        nodeCount_.store(1); maxDepth_.store(0); spare_.store(threads_ - 1);
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<Group> groups;
        if (ownSubtree) MakeGroups(d, ownSubtree, groups);
        if (groups.size() >= 2) BuildGroups(groups.data(), (uint32_t)groups.size(), 0, 1);
        else BuildInternal(0, 0, N, 1);
        const auto t1 = std::chrono::steady_clock::now();
        out.nodes.resize(nodeCount_.load());
        out.maxDepth = maxDepth_.load();
        out.nodes4.reserve(out.nodes.size() / 2 + 1);
        if (optimalCollapse_) PlanCollapse();
        Collapse(0, out.stackNeed);
        plan_.clear(); plan_.shrink_to_fit();
        ReorderBreadthFirst(out.nodes4);
        if (std::getenv("ZR_BVH_TIMING"))
            std::fprintf(stderr, "[zr_bvh] %u triangles, %d threads: SAH build %.1f ms, collapse + reorder %.1f ms\n", N, threads_,
                std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
        return out;
    }

    // Numbers the TOP of the tree breadth-first: the first kTopNodes nodes (five levels of a full 4-wide tree) are the root, then level by level
    // -- what every ray visits becomes one contiguous prefix (the optional LDS node cache of the kernels copies exactly that prefix); all other
    // nodes keep their depth-first order behind it, so subtrees stay contiguous.  Results and the stack bound do not depend on node numbers.
    static constexpr size_t kTopNodes = 341;
    static void ReorderBreadthFirst(std::vector<Bvh4Node>& nodes)
    {
        if (nodes.empty()) return;
        std::vector<uint32_t> order; order.reserve(nodes.size());
        std::vector<uint8_t> placed(nodes.size(), 0);
        order.push_back(0u); placed[0] = 1;
        for (size_t i = 0; i < order.size() && order.size() < kTopNodes; i++)
            for (int c = 0; c < 4 && order.size() < kTopNodes; c++)
            { const uint32_t r = nodes[order[i]].child[c]; if (r != kEmptyChild && !(r & kLeafBit)) { order.push_back(r); placed[r] = 1; } }
        for (uint32_t i = 0; i < (uint32_t)nodes.size(); i++) if (!placed[i]) order.push_back(i);
        std::vector<uint32_t> newIdx(nodes.size(), 0u);
        for (size_t i = 0; i < order.size(); i++) newIdx[order[i]] = (uint32_t)i;
        std::vector<Bvh4Node> out(order.size());
        for (size_t i = 0; i < order.size(); i++)
        {
            Bvh4Node n = nodes[order[i]];
            for (int c = 0; c < 4; c++) if (n.child[c] != kEmptyChild && !(n.child[c] & kLeafBit)) n.child[c] = newIdx[n.child[c]];
            out[i] = n;
        }
        nodes.swap(out);
    }

private:
    // BVH2 -> BVH4: start from a node's two children and keep replacing the inner child of largest surface area by its
    // own two children until there are four (or only leaves are left).  Returns the new node's index; `need` receives the
    // stack entries a traversal below this node can hold at once.
    struct C { uint32_t ref; float lo[3], hi[3]; };
    // ---- the optimal collapse.  For a binary node x, cost[i - 1] = the least SAH cost of x's subtree when it may occupy up to i child slots of the wide node above it:
    //   i = 1: x is a wide node itself: area(x) * cNode + min over k of cost(left, k) + cost(right, 4 - k);
    //   i > 1: either fewer slots (cost i - 1), or x is dissolved and its children share the slots: min over k of cost(left, k) + cost(right, i - k).
    // A leaf costs area * triangles whatever it is given.  split[i - 1] remembers the choice (0: "as with i - 1 slots", else k = the left child's share).
    struct Plan { float cost[4]; uint8_t split[4]; };
    std::vector<Plan> plan_;
    float LeafCost(uint32_t ref, const float lo[3], const float hi[3]) const { return Area(lo, hi) * (float)((ref & 7u) + 1u); }
    float PlanCost(uint32_t ref, const float lo[3], const float hi[3], int slots) const
    { return (ref & kLeafBit) ? LeafCost(ref, lo, hi) : plan_[ref].cost[slots - 1]; }
    void PlanNode(uint32_t x, const float lo[3], const float hi[3])
    {
        const BvhNode& n = out_->nodes[x];
        if (!(n.left & kLeafBit)) PlanNode(n.left, n.lmin, n.lmax);
        if (!(n.right & kLeafBit)) PlanNode(n.right, n.rmin, n.rmax);
        Plan& P = plan_[x];
        float best = 3.402823466e+38f; int bk = 1;
        for (int k = 1; k <= 3; k++)
        {
            const float c = PlanCost(n.left, n.lmin, n.lmax, k) + PlanCost(n.right, n.rmin, n.rmax, 4 - k);
            if (c < best) { best = c; bk = k; }
        }
        P.cost[0] = Area(lo, hi) * collapseNodeCost_ + best; P.split[0] = (uint8_t)bk;
        for (int i = 2; i <= 4; i++)
        {
            float bi = P.cost[i - 2]; int bs = 0;
            for (int k = 1; k < i; k++)
            {
                const float c = PlanCost(n.left, n.lmin, n.lmax, k) + PlanCost(n.right, n.rmin, n.rmax, i - k);
                if (c < bi) { bi = c; bs = k; }
            }
            P.cost[i - 1] = bi; P.split[i - 1] = (uint8_t)bs;
        }
    }
    void PlanCollapse()
    {
        plan_.assign(out_->nodes.size(), Plan());
        const BvhNode& n = out_->nodes[0];
        float lo[3], hi[3];
        for (int r = 0; r < 3; r++) { lo[r] = std::min(n.lmin[r], n.rmin[r]); hi[r] = std::max(n.lmax[r], n.rmax[r]); }
        PlanNode(0, lo, hi);
    }
    // the children of wide node x according to the plan: x's two children share four slots
    void Resolve(uint32_t ref, const float lo[3], const float hi[3], int slots, C* c, int& k)
    {
        if (!(ref & kLeafBit))
        {
            while (slots > 1 && plan_[ref].split[slots - 1] == 0) slots--;
            if (slots > 1)
            {
                const BvhNode& n = out_->nodes[ref];
                const int kl = plan_[ref].split[slots - 1];
                Resolve(n.left, n.lmin, n.lmax, kl, c, k);
                Resolve(n.right, n.rmin, n.rmax, slots - kl, c, k);
                return;
            }
        }
        c[k].ref = ref;
        for (int r = 0; r < 3; r++) { c[k].lo[r] = lo[r]; c[k].hi[r] = hi[r]; }
        k++;
    }
    void GatherPlanned(uint32_t x, C* c, int& k)
    {
        const BvhNode& n = out_->nodes[x];
        const int kl = plan_[x].split[0];
        k = 0;
        Resolve(n.left, n.lmin, n.lmax, kl, c, k);
        Resolve(n.right, n.rmin, n.rmax, 4 - kl, c, k);
    }
    uint32_t Collapse(uint32_t node2, uint32_t& need)
    {
        C c[4];
        int k = 0;
        if (optimalCollapse_) GatherPlanned(node2, c, k);
        else
        {
            const BvhNode& n = out_->nodes[node2];
            k = 2;
            c[0].ref = n.left; c[1].ref = n.right;
            for (int r = 0; r < 3; r++) { c[0].lo[r] = n.lmin[r]; c[0].hi[r] = n.lmax[r]; c[1].lo[r] = n.rmin[r]; c[1].hi[r] = n.rmax[r]; }
            while (k < 4)
            {
                int pick = -1; float bestA = -1.0f;
                for (int i = 0; i < k; i++)
                    if (!(c[i].ref & kLeafBit)) { float a = Area(c[i].lo, c[i].hi); if (a > bestA) { bestA = a; pick = i; } }
                if (pick < 0) break;
                const BvhNode& m = out_->nodes[c[pick].ref];
                c[pick].ref = m.left; c[k].ref = m.right;
                for (int r = 0; r < 3; r++) { c[pick].lo[r] = m.lmin[r]; c[pick].hi[r] = m.lmax[r]; c[k].lo[r] = m.rmin[r]; c[k].hi[r] = m.rmax[r]; }
                k++;
            }
        }
        const uint32_t idx = (uint32_t)out_->nodes4.size();
        out_->nodes4.push_back(Bvh4Node());
        uint32_t refs[4]; uint32_t below = 0;
        for (int i = 0; i < k; i++)
        {
            refs[i] = c[i].ref;
            if (!(c[i].ref & kLeafBit)) { uint32_t nd = 0; refs[i] = Collapse(c[i].ref, nd); below = std::max(below, nd); }
        }
        // quantise: origin = the node's min corner, per-axis scale = the smallest power of two whose 255 steps reach the
        // max corner; child planes rounded outwards against the device's decode expression fma(q, scale, origin)
        float lo[3], hi[3];
        for (int r = 0; r < 3; r++) { lo[r] = c[0].lo[r]; hi[r] = c[0].hi[r]; for (int i = 1; i < k; i++) { lo[r] = std::min(lo[r], c[i].lo[r]); hi[r] = std::max(hi[r], c[i].hi[r]); } }
        uint32_t ex[3], qlo[3] = {0, 0, 0}, qhi[3] = {0, 0, 0};
        for (int r = 0; r < 3; r++)
        {
            int e = 1;
            const double ext = (double)hi[r] - (double)lo[r];
            if (ext > 0) { int ee; std::frexp(ext / 255.0, &ee); e = std::max(1, std::min(254, ee + 126 - 1)); }
            while (e < 254 && !(std::fmaf(255.0f, ScaleOf((uint32_t)e), lo[r]) >= hi[r])) e++;
            ex[r] = (uint32_t)e;
            const float sc = ScaleOf(ex[r]);
            for (int i = 0; i < k; i++)
            {
                int a = (int)std::floor(((double)c[i].lo[r] - (double)lo[r]) / (double)sc); a = std::max(0, std::min(255, a));
                while (a > 0 && std::fmaf((float)a, sc, lo[r]) > c[i].lo[r]) a--;
                int b = (int)std::ceil(((double)c[i].hi[r] - (double)lo[r]) / (double)sc); b = std::max(0, std::min(255, b));
                while (b < 255 && std::fmaf((float)b, sc, lo[r]) < c[i].hi[r]) b++;
                qlo[r] |= (uint32_t)a << (8 * i); qhi[r] |= (uint32_t)b << (8 * i);
            }
        }
        Bvh4Node& o = out_->nodes4[idx];
        o.ox = lo[0]; o.oy = lo[1]; o.oz = lo[2]; o.exps = ex[0] | (ex[1] << 8) | (ex[2] << 16);
        for (int i = 0; i < 4; i++) o.child[i] = i < k ? refs[i] : kEmptyChild;
        o.qlox = qlo[0]; o.qloy = qlo[1]; o.qloz = qlo[2]; o.qhix = qhi[0]; o.qhiy = qhi[1]; o.qhiz = qhi[2]; o.pad0 = 0; o.pad1 = 0;
        need = (uint32_t)(k - 1) + below;
        return idx;
    }

    static float ScaleOf(uint32_t biasedExp) { uint32_t u = biasedExp << 23; float f; std::memcpy(&f, &u, 4); return f; }

    // ---- instances with a subtree of their own (Build's ownSubtree)
    struct Group { uint32_t first, count, order; float bmin[3], bmax[3]; };
    // bt_ arrives in scene order (instance by instance); rearranged to [all triangles of unflagged instances | flagged instance | flagged instance ...]
    void MakeGroups(const zr_scene_desc& d, const uint8_t* own, std::vector<Group>& groups)
    {
        std::vector<BuildTri> re; re.reserve(bt_.size());
        std::vector<uint32_t> base(d.num_instances + 1, 0u);
        for (uint32_t i = 0; i < d.num_instances; i++) base[i + 1] = base[i] + d.instance_num_tris[i];
        auto add = [&](uint32_t first) {
            const uint32_t count = (uint32_t)re.size() - first;
            if (!count) return;
            Group g; g.first = first; g.count = count; g.order = (uint32_t)groups.size();
            for (int r = 0; r < 3; r++) { g.bmin[r] = 3.402823466e+38f; g.bmax[r] = -3.402823466e+38f; }
            for (uint32_t k = first; k < first + count; k++) Grow(g.bmin, g.bmax, re[k].bmin, re[k].bmax);
            groups.push_back(g); };
        for (uint32_t i = 0; i < d.num_instances; i++) if (!own[i]) re.insert(re.end(), bt_.begin() + base[i], bt_.begin() + base[i + 1]);
        add(0);
        for (uint32_t i = 0; i < d.num_instances; i++)
            if (own[i]) { const uint32_t first = (uint32_t)re.size(); re.insert(re.end(), bt_.begin() + base[i], bt_.begin() + base[i + 1]); add(first); }
        bt_.swap(re);
    }
    uint32_t GroupChild(Group* g, uint32_t n, uint32_t depth)
    {
        if (n == 1) return BuildChild(g[0].first, g[0].count, depth);
        const uint32_t idx = nodeCount_.fetch_add(1);
        BuildGroups(g, n, idx, depth + 1);
        return idx;
    }
    // the tree ABOVE the groups: exact SAH sweep over the groups' boxes along each axis, a group weighing its triangle count (there are few groups)
    void BuildGroups(Group* g, uint32_t n, uint32_t nodeIdx, uint32_t depth)
    {
        float bestC = 3.402823466e+38f; int bAxis = 0; uint32_t bK = n / 2;
        std::vector<float> rightArea(n); std::vector<uint32_t> rightN(n);
        auto byAxis = [](int axis) { return [axis](const Group& a, const Group& b) {
            const float ca = a.bmin[axis] + a.bmax[axis], cb = b.bmin[axis] + b.bmax[axis]; return ca < cb || (ca == cb && a.order < b.order); }; };
        for (int axis = 0; axis < 3; axis++)
        {
            std::sort(g, g + n, byAxis(axis));
            float lo[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f}, hi[3] = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
            uint32_t cnt = 0;
            for (uint32_t i = n; i-- > 1;) { Grow(lo, hi, g[i].bmin, g[i].bmax); cnt += g[i].count; rightArea[i] = Area(lo, hi); rightN[i] = cnt; }
            for (int r = 0; r < 3; r++) { lo[r] = 3.402823466e+38f; hi[r] = -3.402823466e+38f; }
            cnt = 0;
            for (uint32_t k = 1; k < n; k++)
            {
                Grow(lo, hi, g[k - 1].bmin, g[k - 1].bmax); cnt += g[k - 1].count;
                const float c = Area(lo, hi) * (float)cnt + rightArea[k] * (float)rightN[k];
                if (c < bestC) { bestC = c; bAxis = axis; bK = k; }
            }
        }
        std::sort(g, g + n, byAxis(bAxis));
        float lmin[3], lmax[3], rmin[3], rmax[3];
        for (int r = 0; r < 3; r++) { lmin[r] = rmin[r] = 3.402823466e+38f; lmax[r] = rmax[r] = -3.402823466e+38f; }
        for (uint32_t i = 0; i < bK; i++) Grow(lmin, lmax, g[i].bmin, g[i].bmax);
        for (uint32_t i = bK; i < n; i++) Grow(rmin, rmax, g[i].bmin, g[i].bmax);
        const uint32_t l = GroupChild(g, bK, depth), r = GroupChild(g + bK, n - bK, depth);
        BvhNode& nd = out_->nodes[nodeIdx];
        for (int k = 0; k < 3; k++) { nd.lmin[k] = lmin[k]; nd.lmax[k] = lmax[k]; nd.rmin[k] = rmin[k]; nd.rmax[k] = rmax[k]; }
        nd.left = l; nd.right = r; nd.pad0 = 0; nd.pad1 = 0;
    }

    std::vector<BuildTri> bt_;
    const std::vector<BvhTri>* soup_ = nullptr;
    BuiltBvh* out_ = nullptr;
    std::atomic<uint32_t> nodeCount_{0}, maxDepth_{0};
    std::atomic<int> spare_{0};      // threads that may still be started
    void NoteDepth(uint32_t d) { uint32_t cur = maxDepth_.load(); while (d > cur && !maxDepth_.compare_exchange_weak(cur, d)) {} }

    static void Grow(float bmin[3], float bmax[3], const float lo[3], const float hi[3])
    { for (int r = 0; r < 3; r++) { bmin[r] = std::min(bmin[r], lo[r]); bmax[r] = std::max(bmax[r], hi[r]); } }
    static float Area(const float bmin[3], const float bmax[3])
    {
        float dx = bmax[0] - bmin[0], dy = bmax[1] - bmin[1], dz = bmax[2] - bmin[2];
        return 2.0f * (dx * dy + dy * dz + dz * dx);
    }
    void Bounds(uint32_t first, uint32_t count, float bmin[3], float bmax[3]) const
    {
        for (int r = 0; r < 3; r++) { bmin[r] = 3.402823466e+38f; bmax[r] = -3.402823466e+38f; }
        for (uint32_t i = first; i < first + count; i++) Grow(bmin, bmax, bt_[i].bmin, bt_[i].bmax);
    }
    uint32_t MakeLeaf(uint32_t first, uint32_t count)
    {
        const uint32_t slot = first;
        // deterministic leaf order: ascending global index
        std::sort(bt_.begin() + first, bt_.begin() + first + count, [](const BuildTri& a, const BuildTri& b) { return a.gidx < b.gidx; });
        for (uint32_t i = first; i < first + count; i++) out_->tris[i] = (*soup_)[bt_[i].gidx];
        return kLeafBit | (slot << 3) | (count - 1);
    }
    // splits [first, first+count) and returns the child reference (leaf or node index)
    uint32_t BuildChild(uint32_t first, uint32_t count, uint32_t depth)
    {
        if (count <= (nodeCost_ > 0 ? 1u : maxLeaf_) || (count <= maxLeaf_ && LeafIsCheaper(first, count))) { NoteDepth(depth); return MakeLeaf(first, count); }
        const uint32_t idx = nodeCount_.fetch_add(1);
        BuildInternal(idx, first, count, depth + 1);
        return idx;
    }
    // SAH leaf test for small ranges: cost(leaf) = count; cost(split) = nodeCost + sum over the best object split of area fraction x count
    bool LeafIsCheaper(uint32_t first, uint32_t count)
    {
        if (!(nodeCost_ > 0) || count < 2) return count < 2;
        float pmin[3], pmax[3]; Bounds(first, count, pmin, pmax);
        const float parentArea = Area(pmin, pmax);
        if (!(parentArea > 0)) return true;
        float best = 3.402823466e+38f;
        std::vector<BuildTri> tmp(bt_.begin() + first, bt_.begin() + first + count);
        for (int axis = 0; axis < 3; axis++)
        {
            std::sort(tmp.begin(), tmp.end(), [axis](const BuildTri& a, const BuildTri& b) { return a.cent[axis] < b.cent[axis]; });
            for (uint32_t k = 1; k < count; k++)
            {
                float lmin[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f}, lmax[3] = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
                float rmin[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f}, rmax[3] = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
                for (uint32_t i = 0; i < k; i++) Grow(lmin, lmax, tmp[i].bmin, tmp[i].bmax);
                for (uint32_t i = k; i < count; i++) Grow(rmin, rmax, tmp[i].bmin, tmp[i].bmax);
                best = std::min(best, (Area(lmin, lmax) * (float)k + Area(rmin, rmax) * (float)(count - k)) / parentArea);
            }
        }
        return (float)count <= nodeCost_ + best;
    }
    void BuildInternal(uint32_t nodeIdx, uint32_t first, uint32_t count, uint32_t depth)
    {
        // centroid bounds
        float cmin[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f}, cmax[3] = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
        for (uint32_t i = first; i < first + count; i++)
            for (int r = 0; r < 3; r++) { cmin[r] = std::min(cmin[r], bt_[i].cent[r]); cmax[r] = std::max(cmax[r], bt_[i].cent[r]); }
        int bestAxis = -1, bestSplit = -1; float bestCost = 3.402823466e+38f;
        for (int axis = 0; axis < 3; axis++)
        {
            float ext = cmax[axis] - cmin[axis];
            if (!(ext > 0)) continue;
            struct Bin { float bmin[3], bmax[3]; uint32_t n; } bins[kBins];
            for (auto& b : bins) { for (int r = 0; r < 3; r++) { b.bmin[r] = 3.402823466e+38f; b.bmax[r] = -3.402823466e+38f; } b.n = 0; }
            float scale = (float)kBins / ext;
            for (uint32_t i = first; i < first + count; i++)
            {
                int b = std::min(kBins - 1, (int)((bt_[i].cent[axis] - cmin[axis]) * scale));
                Grow(bins[b].bmin, bins[b].bmax, bt_[i].bmin, bt_[i].bmax); bins[b].n++;
            }
            float la[kBins], ra[kBins]; uint32_t ln[kBins], rn[kBins];
            float lb[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f}, ub[3] = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
            uint32_t n = 0;
            for (int b = 0; b < kBins; b++) { if (bins[b].n) Grow(lb, ub, bins[b].bmin, bins[b].bmax); n += bins[b].n; la[b] = n ? Area(lb, ub) : 0; ln[b] = n; }
            for (int r = 0; r < 3; r++) { lb[r] = 3.402823466e+38f; ub[r] = -3.402823466e+38f; }
            n = 0;
            for (int b = kBins - 1; b >= 0; b--) { if (bins[b].n) Grow(lb, ub, bins[b].bmin, bins[b].bmax); n += bins[b].n; ra[b] = n ? Area(lb, ub) : 0; rn[b] = n; }
            for (int s = 0; s < kBins - 1; s++)
            {
                if (ln[s] == 0 || rn[s + 1] == 0) continue;
                float cost = la[s] * (float)ln[s] + ra[s + 1] * (float)rn[s + 1];
                if (cost < bestCost) { bestCost = cost; bestAxis = axis; bestSplit = s; }
            }
        }
        uint32_t mid;
        if (count < medianBelow_ && bestAxis >= 0)
        {
            int ax = 0; float we = -1.0f;
            for (int r = 0; r < 3; r++) if (cmax[r] - cmin[r] > we) { we = cmax[r] - cmin[r]; ax = r; }
            std::sort(bt_.begin() + first, bt_.begin() + first + count, [ax](const BuildTri& a, const BuildTri& b) { return a.cent[ax] < b.cent[ax] || (a.cent[ax] == b.cent[ax] && a.gidx < b.gidx); });
            uint32_t half = count / 2; if ((half & 1u) && half + 1 < count) half++;      // even halves: leaves of two
            mid = first + half;
        }
        else if (count < sweepBelow_ && bestAxis >= 0)
        {
            // exact sweep: every object split along every axis
            float bestC = 3.402823466e+38f; int bAxis = -1; uint32_t bK = 0;
            std::vector<float> rightArea(count);
            for (int axis = 0; axis < 3; axis++)
            {
                std::sort(bt_.begin() + first, bt_.begin() + first + count, [axis](const BuildTri& a, const BuildTri& b) { return a.cent[axis] < b.cent[axis] || (a.cent[axis] == b.cent[axis] && a.gidx < b.gidx); });
                float lo[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f}, hi[3] = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
                for (uint32_t i = count; i-- > 1;) { Grow(lo, hi, bt_[first + i].bmin, bt_[first + i].bmax); rightArea[i] = Area(lo, hi); }
                for (int r = 0; r < 3; r++) { lo[r] = 3.402823466e+38f; hi[r] = -3.402823466e+38f; }
                for (uint32_t k = 1; k < count; k++)
                {
                    Grow(lo, hi, bt_[first + k - 1].bmin, bt_[first + k - 1].bmax);
                    const float c = Area(lo, hi) * (float)k + rightArea[k] * (float)(count - k);
                    if (c < bestC) { bestC = c; bAxis = axis; bK = k; }
                }
            }
            std::sort(bt_.begin() + first, bt_.begin() + first + count, [bAxis](const BuildTri& a, const BuildTri& b) { return a.cent[bAxis] < b.cent[bAxis] || (a.cent[bAxis] == b.cent[bAxis] && a.gidx < b.gidx); });
            mid = first + bK;
        }
        else if (bestAxis < 0)
        {
            // all centroids coincide: split by index
            mid = first + count / 2;
            std::sort(bt_.begin() + first, bt_.begin() + first + count, [](const BuildTri& a, const BuildTri& b) { return a.gidx < b.gidx; });
        }
        else
        {
            float ext = cmax[bestAxis] - cmin[bestAxis];
            float scale = (float)kBins / ext;
            float c0 = cmin[bestAxis];
            auto it = std::partition(bt_.begin() + first, bt_.begin() + first + count, [&](const BuildTri& t) {
                int b = std::min(kBins - 1, (int)((t.cent[bestAxis] - c0) * scale));
                return b <= bestSplit; });
            mid = (uint32_t)(it - bt_.begin());
            if (mid == first || mid == first + count) mid = first + count / 2;
        }
        float lmin[3], lmax[3], rmin[3], rmax[3];
        Bounds(first, mid - first, lmin, lmax);
        Bounds(mid, first + count - mid, rmin, rmax);
        uint32_t l, r;
        bool forked = false;
        if (count >= 8192u)      // (small ranges are not worth a thread)
        {
            int have = spare_.load();
            while (have > 0 && !spare_.compare_exchange_weak(have, have - 1)) {}
            forked = have > 0;
        }
        if (forked)
        {
            std::thread other([&] { l = BuildChild(first, mid - first, depth); });
            r = BuildChild(mid, first + count - mid, depth);
            other.join();
            spare_.fetch_add(1);
        }
        else
        {
            l = BuildChild(first, mid - first, depth);
            r = BuildChild(mid, first + count - mid, depth);
        }
        BvhNode& n = out_->nodes[nodeIdx];
        for (int k = 0; k < 3; k++) { n.lmin[k] = lmin[k]; n.lmax[k] = lmax[k]; n.rmin[k] = rmin[k]; n.rmax[k] = rmax[k]; }
        n.left = l; n.right = r; n.pad0 = 0; n.pad1 = 0;
    }
};

} // namespace zr
