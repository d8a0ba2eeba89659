// zr_dev_scene.h -- device view of the scene, BVH traversal, hit reconstruction, material fetch, light sampling.
//
// Replaces, for the HIP path, what the reference gets from the D3D12 driver (TLAS/BLAS + inline RayQuery,
// Source/ZetaRenderPass/Common/RayQuery.hlsli:42-53,168-179,317-331,372-396) with an explicit BVH2 over world-space
// triangles in HBM, and restates the shader-side code around it:
//   RayQuery.hlsli:15-144  (Hit::FindClosest vertex fetch / TRS / tri differentials / PCG3d ID)
//   RayQuery.hlsli:452-524 (GetMaterialData)      Material.h:268-417 (accessors)
//   LightSource.hlsli:48-137, 202-223 (alias table draw, emissive triangle decode/sample, Le)
//
// HBM layout:
//   nodes : 64 B each  = {L.min xyz, L.max xyz, R.min xyz, R.max xyz, left, right, pad, pad}; child >= 0x80000000 is a
//           leaf: bits 0..2 = count-1, bits 3..30 = first triangle slot
//   tris  : 48 B each in leaf order = {v0 xyz, globalTriIdx | e1 xyz, subgroup mask | e2 xyz, meshIdx}
//   triMeta : 8 B per *global* triangle = {meshIdx, primIdx} (hit reconstruction)
#pragma once
#include "zr_dev_bsdf.h"
#include "../../include/zr_intersect.h"

namespace zr {

struct BvhNode { float lmin[3], lmax[3], rmin[3], rmax[3]; uint32_t left, right, pad0, pad1; };
struct BvhTri { float v0[3]; uint32_t gidx; float e1[3]; uint32_t mask; float e2[3]; uint32_t mesh; };
struct TriMeta { uint32_t mesh, prim; };

static constexpr uint32_t kLeafBit = 0x80000000u;
static constexpr uint32_t kInvalidTri = 0xffffffffu;

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
    const BvhNode* nodes;
    const BvhTri* tris;
    const TriMeta* triMeta;
    RhoView rho;
    uint32_t numEmissives;
    uint32_t numNodes;       // 0 => single leaf covering tris[0 .. numTris)
    uint32_t numTris;
};

struct RawHit { float t, u, v; uint32_t tri; };     // tri = global triangle index, kInvalidTri on miss

// hashed triangle ID of the reference: PCG3d(GeometryIndex = meshIdx, InstanceID = 0, PrimitiveIndex).x
ZR_HD uint32_t TriID(uint32_t meshIdx, uint32_t primIdx)
{ uint32_t kx = meshIdx, ky = 0, kz = primIdx; zr_pcg3d(&kx, &ky, &kz); return kx; }

ZR_HD void IntersectLeaf(const SceneView& sc, uint32_t first, uint32_t count, V3 o, V3 d, float tmin, float tmax,
    uint32_t mask, RawHit& best, bool filterID = false, uint32_t ignoreID = 0)
{
    for (uint32_t i = first; i < first + count; i++)
    {
        const BvhTri& T = sc.tris[i];
        if (!(T.mask & mask)) continue;
        if (filterID) { const TriMeta tm = sc.triMeta[T.gidx]; if (TriID(tm.mesh, tm.prim) == ignoreID) continue; }
        float t, u, v;
        if (zr_ray_tri(o.x, o.y, o.z, d.x, d.y, d.z, T.v0[0], T.v0[1], T.v0[2], T.e1[0], T.e1[1], T.e1[2],
                T.e2[0], T.e2[1], T.e2[2], tmin, tmax, &t, &u, &v))
        {
            // closest hit; equal t goes to the smaller global triangle index (ABI tie-break)
            if (best.tri == kInvalidTri || t < best.t || (t == best.t && T.gidx < best.tri))
            { best.t = t; best.u = u; best.v = v; best.tri = T.gidx; }
        }
    }
}

// Stack-based BVH2 traversal.  `stack` points at this lane's private stack (scratch on the host executor, registers /
// scratch / LDS slice on the device -- the caller decides).  anyHit: return on the first accepted hit.
template<bool AnyHit>
ZR_HD RawHit Traverse(const SceneView& sc, V3 o, V3 d, float tmin, float tmax, uint32_t mask, uint32_t* stack, bool filterID = false, uint32_t ignoreID = 0)
{
    RawHit best; best.t = tmax; best.u = 0; best.v = 0; best.tri = kInvalidTri;
    if (sc.numNodes == 0)
    {
        IntersectLeaf(sc, 0, sc.numTris, o, d, tmin, tmax, mask, best, filterID, ignoreID);
        return best;
    }
    const float idx = zr_safe_rcp_dir(d.x), idy = zr_safe_rcp_dir(d.y), idz = zr_safe_rcp_dir(d.z);
    int sp = 0;
    uint32_t cur = 0;
    for (;;)
    {
        if (cur & kLeafBit)
        {
            uint32_t first = (cur & 0x7fffffffu) >> 3, count = (cur & 7u) + 1u;
            IntersectLeaf(sc, first, count, o, d, tmin, tmax, mask, best, filterID, ignoreID);
            if (AnyHit && best.tri != kInvalidTri) return best;
            if (sp == 0) break;
            cur = stack[--sp];
            continue;
        }
        const BvhNode& n = sc.nodes[cur];
        float tl, tr;
        // cull against the current best t (inclusive + widened, so equal-t candidates for the tie-break are visited)
        bool hl = zr_ray_box(o.x, o.y, o.z, idx, idy, idz, n.lmin[0], n.lmin[1], n.lmin[2], n.lmax[0], n.lmax[1], n.lmax[2], tmin, best.t, &tl);
        bool hr = zr_ray_box(o.x, o.y, o.z, idx, idy, idz, n.rmin[0], n.rmin[1], n.rmin[2], n.rmax[0], n.rmax[1], n.rmax[2], tmin, best.t, &tr);
        if (hl && hr)
        {
            uint32_t nearC = n.left, farC = n.right;
            if (tr < tl) { nearC = n.right; farC = n.left; }
            stack[sp++] = farC;
            cur = nearC;
        }
        else if (hl) cur = n.left;
        else if (hr) cur = n.right;
        else
        {
            if (sp == 0) break;
            cur = stack[--sp];
        }
    }
    return best;
}

// ---- Material.h accessors ----
ZR_HD bool MatDoubleSided(const zr_material& m) { return m.coat_color_flags & (1u << ZR_MAT_DOUBLE_SIDED_BIT); }
ZR_HD bool MatMetallic(const zr_material& m) { return m.coat_color_flags & (1u << ZR_MAT_METALLIC_BIT); }
ZR_HD bool MatTransmissive(const zr_material& m) { return m.coat_color_flags & (1u << ZR_MAT_TRANSMISSIVE_BIT); }
ZR_HD bool MatThinWalled(const zr_material& m) { return m.coat_color_flags & (1u << ZR_MAT_THIN_WALLED_BIT); }
ZR_HD float MatRoughness(const zr_material& m) { return (float)((m.mr_tex_spec_roughness_coat_roughness >> 16) & 0xff) / 255.0f; }
ZR_HD float MatCoatRoughness(const zr_material& m) { return (float)((m.mr_tex_spec_roughness_coat_roughness >> 24) & 0xff) / 255.0f; }
ZR_HD float MatIOR(const zr_material& m) { return zr_fma(1.5f / 65535.0f, (float)(m.emissive_strength_ior >> 16), kMinIOR); }
ZR_HD float MatCoatIOR(const zr_material& m) { return zr_fma(1.5f / 255.0f, (float)((m.emissive_tex_alpha_cutoff_coat_ior >> 24) & 0xff), kMinIOR); }
ZR_HD float MatTrDepth(const zr_material& m) { return zr_f16_to_f32((uint16_t)(m.normal_tex_tr_depth >> 16)); }
ZR_HD float MatSubsurface(const zr_material& m) { return (float)((m.base_color_tex_subsurf_coat_weight >> 16) & 0xff) / 255.0f; }
ZR_HD float MatCoatWeight(const zr_material& m) { return (float)((m.base_color_tex_subsurf_coat_weight >> 24) & 0xff) / 255.0f; }
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
ZR_HD bool GetMaterialData(const SceneView& sc, V3 wo, float eta_curr, HitInfo& hit, Surface& surface, float& eta)
{
    const zr_material mat = sc.materials[hit.matIdx];
    const bool hitBackface = dot(wo, hit.normal) < 0;
    eta = kDefaultEtaMat;
    const bool ds = MatDoubleSided(mat);
    if (!ds && hitBackface) return false;
    if (ds && hitBackface) hit.normal = hit.normal * -1.0f;
    V3 baseColor = UnpackRGB8(mat.base_color_factor);
    float metallic = MatMetallic(mat) ? 1.0f : 0.0f;
    float roughness = MatRoughness(mat);
    bool tr = MatTransmissive(mat);
    eta = MatIOR(mat);
    float trDepth = tr ? MatTrDepth(mat) : 0;
    float eta_next = eta_curr == kEtaAir ? eta : kEtaAir;
    float subsurface = MatThinWalled(mat) ? zr_round_f16(MatSubsurface(mat)) : 0;
    surface = InitSurface(hit.normal, wo, metallic >= kMinMetalnessMetal, roughness, baseColor, eta_curr, eta_next, tr, trDepth,
        subsurface, MatCoatWeight(mat), UnpackRGB8(mat.coat_color_flags), MatCoatRoughness(mat), MatCoatIOR(mat));
    return true;
}

// ---- emissive triangles (RtCommon.h:66-131, LightSource.hlsli:48-70) ----
ZR_HD bool EmDoubleSided(const zr_emissive_triangle& t) { return t.packed_a & (1u << 25); }
ZR_HD V3 EmV1(const zr_emissive_triangle& t)
{
    V3 d = DecodeUnitVector(v2((float)t.v0v1[0] / 65535.0f, (float)t.v0v1[1] / 65535.0f));
    return mad(zr_f16_to_f32(t.edge_lengths[0]), d, v3p(t.vtx0));
}
ZR_HD V3 EmV2(const zr_emissive_triangle& t)
{
    V3 d = DecodeUnitVector(v2((float)t.v0v2[0] / 65535.0f, (float)t.v0v2[1] / 65535.0f));
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

// Le_EmissiveTriangle, LightSource.hlsli:202-223 (emissive textures not bound)
ZR_HD V3 EmLe(const zr_emissive_triangle& t)
{
    V3 le = UnpackRGB8(t.packed_a) * zr_f16_to_f32((uint16_t)(t.packed_b >> 16));
    if (Luminance(le) == 0) return v3(0.0f);
    return le;
}

} // namespace zr
