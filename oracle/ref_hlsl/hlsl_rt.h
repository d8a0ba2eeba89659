// ORACLE tooling -- test infrastructure only.
//
// hlsl_rt.h: DXR inline ray tracing (RayQuery / RayDesc / RaytracingAccelerationStructure) and thread-group / wave intrinsics for the
// reference's shader PASSES compiled as C++ (see hlsl_shim.h).  The reference leaves traversal and ray / triangle intersection to the
// D3D12 driver (unpinnable); here a query runs the ABI's definition of them (include/zr_intersect.h through the oracle's
// scene, oracle/zro_scene.h): closest hit = smallest t, ties to the smaller global triangle index; any-hit queries are order-free.
#pragma once
#include "hlsl_resources.h"

namespace hlsl {

enum RAY_FLAG : uint32_t
{
    RAY_FLAG_NONE = 0, RAY_FLAG_FORCE_OPAQUE = 0x1, RAY_FLAG_FORCE_NON_OPAQUE = 0x2, RAY_FLAG_ACCEPT_FIRST_HIT_AND_END_SEARCH = 0x4,
    RAY_FLAG_SKIP_CLOSEST_HIT_SHADER = 0x8, RAY_FLAG_CULL_BACK_FACING_TRIANGLES = 0x10, RAY_FLAG_CULL_FRONT_FACING_TRIANGLES = 0x20,
    RAY_FLAG_CULL_OPAQUE = 0x40, RAY_FLAG_CULL_NON_OPAQUE = 0x80, RAY_FLAG_SKIP_TRIANGLES = 0x100, RAY_FLAG_SKIP_PROCEDURAL_PRIMITIVES = 0x200
};
enum COMMITTED_STATUS : uint32_t { COMMITTED_NOTHING = 0, COMMITTED_TRIANGLE_HIT = 1, COMMITTED_PROCEDURAL_PRIMITIVE_HIT = 2 };
enum CANDIDATE_TYPE : uint32_t { CANDIDATE_NON_OPAQUE_TRIANGLE = 0, CANDIDATE_PROCEDURAL_PRIMITIVE = 1 };

struct RayDesc { float3 Origin; float TMin; float3 Direction; float TMax; };
struct RaytracingAccelerationStructure { const zro::Scene* scene = nullptr; };

// Visibility_Segment with APPROXIMATE_EMISSIVE_SHADOW_RAY (RayQuery.hlsli:372-403) asks for "any hit" and then compares the hit's ID with the
// light's: which hit the driver reports first depends on its traversal order.  The ABI pins it order-independently: triangles
// carrying the target's ID are not occluders, any other hit inside the shortened segment is.  hlsl2cpp.py passes the target ID to the query.
static thread_local uint32_t g_rqIgnoreID = 0; static thread_local bool g_rqHasIgnoreID = false;

template<uint32_t StaticFlags> struct RayQuery
{
    const zro::Scene* sc = nullptr;
    RayDesc ray; uint32_t mask = 0xff; uint32_t flags = StaticFlags;
    zro::Scene::RawHit committed; bool done = false;
    // non-opaque candidate enumeration (primary rays, GBufferRT_Inline.hlsl:80-95)
    struct Cand { float t, u, v; uint32_t tri; bool opaque; };
    std::vector<Cand> cands; size_t next = 0; bool enumerating = false; Cand cur;

    void TraceRayInline(const RaytracingAccelerationStructure& as, uint32_t runtimeFlags, uint32_t instanceMask, const RayDesc& r)
    {
        sc = as.scene; ray = r; mask = instanceMask; flags = StaticFlags | runtimeFlags; done = false; enumerating = false; next = 0; cands.clear();
        committed.hit = false; committed.t = r.TMax; committed.u = committed.v = 0; committed.tri = 0xffffffffu;
    }
    static zro::float3 Z(const float3& v) { return zro::f3(v.x, v.y, v.z); }
    bool Proceed()
    {
        if (done) return false;
        const bool anyHit = (flags & RAY_FLAG_ACCEPT_FIRST_HIT_AND_END_SEARCH) != 0;
        if (flags & RAY_FLAG_FORCE_OPAQUE)
        {
            const bool filt = anyHit && g_rqHasIgnoreID;
            committed = sc->Trace(Z(ray.Origin), Z(ray.Direction), ray.TMin, ray.TMax, mask, anyHit, filt, g_rqIgnoreID, false);
            g_rqHasIgnoreID = false;
            done = true;
            return false;
        }
        if (!enumerating)
        {
            // every intersection inside (TMin, TMax), ordered by (t, global triangle index): the first ACCEPTED one is the closest hit
            enumerating = true;
            for (uint32_t ti = 0; ti < sc->tris.size(); ti++)
            {
                const zro::WorldTri& T = sc->tris[ti];
                if (!(T.mask & mask)) continue;
                float t, u, v;
                if (zr_ray_tri(ray.Origin.x, ray.Origin.y, ray.Origin.z, ray.Direction.x, ray.Direction.y, ray.Direction.z, T.v0[0], T.v0[1], T.v0[2],
                        T.e1[0], T.e1[1], T.e1[2], T.e2[0], T.e2[1], T.e2[2], ray.TMin, ray.TMax, &t, &u, &v))
                    cands.push_back(Cand{t, u, v, ti, !(T.mask & ZR_INSTANCE_NON_OPAQUE)});
            }
            std::sort(cands.begin(), cands.end(), [](const Cand& a, const Cand& b) { return a.t < b.t || (a.t == b.t && a.tri < b.tri); });
        }
        while (next < cands.size())
        {
            cur = cands[next++];
            if (cur.opaque) { Commit(cur); done = true; return false; }
            return true;            // non-opaque candidate: the shader decides (CommitNonOpaqueTriangleHit)
        }
        done = true;
        return false;
    }
    void Commit(const Cand& c) { committed.hit = true; committed.t = c.t; committed.u = c.u; committed.v = c.v; committed.tri = c.tri; }
    void CommitNonOpaqueTriangleHit() { Commit(cur); next = cands.size(); }     // ordered candidates: the first accepted one is final
    void Abort() { done = true; }
    uint32_t CandidateType() const { return CANDIDATE_NON_OPAQUE_TRIANGLE; }
    uint32_t CandidateGeometryIndex() const { return sc->tris[cur.tri].mesh_idx; }
    uint32_t CandidateInstanceID() const { return 0; }
    uint32_t CandidatePrimitiveIndex() const { return sc->tris[cur.tri].prim_idx; }
    float2 CandidateTriangleBarycentrics() const { return float2(cur.u, cur.v); }
    float CandidateTriangleRayT() const { return cur.t; }
    uint32_t CommittedStatus() const { return committed.hit ? COMMITTED_TRIANGLE_HIT : COMMITTED_NOTHING; }
    // static geometry lives in one BLAS: GeometryIndex = mesh index, InstanceID = 0 (RtAccelerationStructure.cpp:393-405)
    uint32_t CommittedGeometryIndex() const { return sc->tris[committed.tri].mesh_idx; }
    uint32_t CommittedInstanceID() const { return 0; }
    uint32_t CommittedInstanceIndex() const { return 0; }
    uint32_t CommittedPrimitiveIndex() const { return sc->tris[committed.tri].prim_idx; }
    float2 CommittedTriangleBarycentrics() const { return float2(committed.u, committed.v); }
    float CommittedRayT() const { return committed.t; }
    float3 WorldRayDirection() const { return ray.Direction; }
    float3 WorldRayOrigin() const { return ray.Origin; }
    bool CommittedTriangleFrontFace() const { return true; }
};

template<class T> inline T NonUniformResourceIndex(T i) { return i; }

// ---- thread group / wave execution model --------------------------------------------------------------------------------------
// A pass driver runs one thread group at a time.  Shaders without cross-lane operations run their threads one after the other;
// shaders with wave intrinsics / group barriers run every thread of the group as a fiber and meet at each cross-lane operation
// (GroupRunner in hlsl_group.h).  These hooks are what the intrinsics call.
struct LaneHooks
{
    uint32_t laneIndex = 0, laneCount = 1;
    // cross-lane reduce: every live lane contributes `v` (as 4 x uint32 payload); returns after all live lanes have arrived
    void (*sync)(void* ctx) = nullptr; void* ctx = nullptr;
};
static thread_local LaneHooks* g_lane = nullptr;

} // namespace hlsl
