// ORACLE -- test infrastructure only (see zro_math.h header).
//
// zro_scene.h: scene container + CPU restatement of the reference's ray queries, material fetch and light sampling:
//   Source/ZetaRenderPass/Common/RayQuery.hlsli, LightSource.hlsli, Source/ZetaCore/Core/Material.h,
//   Source/ZetaCore/RayTracing/RtCommon.h.  Traversal = the oracle's own median-split BVH2 over world-space triangles
//   (or brute force), using the ABI's intersection arithmetic (include/zr_intersect.h).
#pragma once
#include <vector>
#include <algorithm>
#include "zro_bsdf.h"
#include "zro_sky.h"
#include "../include/zr_texture.h"
#include "../include/zr_srgb_table.h"
#include "../include/zr_intersect.h"

namespace zro {

// Material.h:268-417 accessors
struct Mat
{
    zr_material m;
    bool DoubleSided() const { return m.coat_color_flags & (1u << ZR_MAT_DOUBLE_SIDED_BIT); }
    bool Metallic() const { return m.coat_color_flags & (1u << ZR_MAT_METALLIC_BIT); }
    bool Transmissive() const { return m.coat_color_flags & (1u << ZR_MAT_TRANSMISSIVE_BIT); }
    bool ThinWalled() const { return m.coat_color_flags & (1u << ZR_MAT_THIN_WALLED_BIT); }
    float3 GetBaseColorFactor() const { return Math::UnpackRGB8(m.base_color_factor); }
    float3 GetCoatColor() const { return Math::UnpackRGB8(m.coat_color_flags); }
    float3 GetEmissiveFactor() const { return Math::UnpackRGB8(m.emissive_factor_normal_scale); }
    float GetNormalScale() const { return Math::UNorm8ToFloat(m.emissive_factor_normal_scale >> 24); }
    uint32_t GetBaseColorTex() const { return m.base_color_tex_subsurf_coat_weight & 0xffffu; }
    uint32_t GetNormalTex() const { return m.normal_tex_tr_depth & 0xffffu; }
    uint32_t GetMetallicRoughnessTex() const { return m.mr_tex_spec_roughness_coat_roughness & 0xffffu; }
    uint32_t GetEmissiveTex() const { return m.emissive_tex_alpha_cutoff_coat_ior & 0xffffu; }
    float GetAlphaCutoff() const { return Math::UNorm8ToFloat((m.emissive_tex_alpha_cutoff_coat_ior >> 16) & 0xff); }
    float GetCoatIOR() const
    { return zr_fma(1.5f / 255.0f, (float)((m.emissive_tex_alpha_cutoff_coat_ior >> 24) & 0xff), MIN_IOR); }
    float GetSpecularRoughness() const { return Math::UNorm8ToFloat((m.mr_tex_spec_roughness_coat_roughness >> 16) & 0xff); }
    float GetCoatRoughness() const { return Math::UNorm8ToFloat((m.mr_tex_spec_roughness_coat_roughness >> 24) & 0xff); }
    float GetEmissiveStrength() const { return zr_f16_to_f32((uint16_t)(m.emissive_strength_ior & 0xffff)); }
    float GetSpecularIOR() const { return zr_fma(1.5f / 65535.0f, (float)(m.emissive_strength_ior >> 16), MIN_IOR); }
    float GetTransmissionDepth() const { return zr_f16_to_f32((uint16_t)(m.normal_tex_tr_depth >> 16)); }
    float GetSubsurface() const { return Math::UNorm8ToFloat((m.base_color_tex_subsurf_coat_weight >> 16) & 0xff); }
    float GetCoatWeight() const { return Math::UNorm8ToFloat((m.base_color_tex_subsurf_coat_weight >> 24) & 0xff); }
};

// RtCommon.h:66-131 accessors, LightSource.hlsli:48-70 decode
struct EmTri
{
    zr_emissive_triangle t;
    bool IsDoubleSided() const { return t.packed_a & (1u << 25); }
    float GetStrength() const { return zr_f16_to_f32((uint16_t)(t.packed_b >> 16)); }
    float3 GetFactor() const { return Math::UnpackRGB8(t.packed_a); }
    uint32_t GetTex() const { return t.packed_b & 0xffffu; }
    float2 UV0() const { return {zr_f16_to_f32(t.uv0[0]), zr_f16_to_f32(t.uv0[1])}; }     // EMISSIVE_UV_HALF == 1
    float2 UV1() const { return {zr_f16_to_f32(t.uv1[0]), zr_f16_to_f32(t.uv1[1])}; }
    float2 UV2() const { return {zr_f16_to_f32(t.uv2[0]), zr_f16_to_f32(t.uv2[1])}; }
    float3 Vtx0() const { return f3(t.vtx0); }
    float3 V1() const
    {
        float2 e = {(float)t.v0v1[0] / 65535.0f, (float)t.v0v1[1] / 65535.0f};
        float3 d = Math::DecodeUnitVector(e);
        return mad3(zr_f16_to_f32(t.edge_lengths[0]), d, Vtx0());      // mad(decoded, len, Vtx0)
    }
    float3 V2() const
    {
        float2 e = {(float)t.v0v2[0] / 65535.0f, (float)t.v0v2[1] / 65535.0f};
        float3 d = Math::DecodeUnitVector(e);
        return mad3(zr_f16_to_f32(t.edge_lengths[1]), d, Vtx0());
    }
};

struct WorldTri { float v0[3], e1[3], e2[3]; uint32_t mesh_idx, prim_idx, mask; };
struct BVHNode { float bmin[3], bmax[3]; uint32_t left, right, first, count; };   // count > 0 => leaf

struct Counters { uint64_t n_closest = 0, n_shadow = 0; };

struct Scene
{
    std::vector<zr_vertex> vertices;
    std::vector<uint32_t> indices;
    std::vector<zr_mesh_instance> instances;
    std::vector<zr_material> materials;
    std::vector<zr_emissive_triangle> emissives;
    std::vector<zr_alias_entry> alias;
    std::vector<zr_presampled_tri> sampleSets;   // K3 output (PresampleEmissives.hlsl), numSets x setSize
    uint32_t sampleSetSize = 0;
    std::vector<zr_voxel_sample> lvgData;        // K4 output (BuildLightVoxelGrid.hlsl): 64 samples per voxel
    uint32_t lvgDim[3] = {0, 0, 0}; float lvgExtents[3] = {0, 0, 0}; float lvgOffsetY = 0;
    std::vector<uint32_t> skyData;               // K17 output (SkyViewLUT.hlsl), R11G11B10_FLOAT texels
    SkyLUT sky;
    std::vector<uint16_t> rho;
    RhoLUT rhoLUT;
    // material texture heap (zr_wire.h zr_texture_desc / zr_texture.h) and the four descriptor-table offsets of the frame
    // constants (FrameConstants.h:31-34), latched by every render entry point before it shades
    std::vector<zr_texture_desc> texDescs;
    std::vector<uint8_t> texels;
    zr_tex_heap tex = {nullptr, nullptr, nullptr, 0};
    mutable uint32_t baseColorMapsOffset = 0, normalMapsOffset = 0, mrMapsOffset = 0, emissiveMapsOffset = 0;
    // TEXTURE_FILTER of the indirect-lighting pass being rendered (cb_ReSTIR_*::TexFilterDescHeapIdx); the other passes use ANISOTROPIC_4X
    mutable uint32_t texFilter = ZR_TEX_FILTER_ANISOTROPIC_4X;
    void LatchHeapOffsets(const zr_frame_constants& g) const
    {
        texFilter = ZR_TEX_FILTER_ANISOTROPIC_4X;
        baseColorMapsOffset = g.base_color_maps_desc_heap_offset; normalMapsOffset = g.normal_maps_desc_heap_offset;
        mrMapsOffset = g.metallic_roughness_maps_desc_heap_offset; emissiveMapsOffset = g.emissive_maps_desc_heap_offset;
    }
    std::vector<WorldTri> tris;          // global triangle order = instance order, then primitive order
    std::vector<BVHNode> nodes;
    std::vector<uint32_t> triOrder;      // BVH leaf order -> global triangle index
    bool bruteForce = true;
    mutable Counters counters;
    // previous frame's scene (RT_SCENE_BVH_PREV + RT_FRAME_MESH_INSTANCES_PREV, RtAccelerationStructure.cpp:382-506): what the CtT passes of
    // ReSTIR PT and the temporal shifts of the DI passes bind; null = nothing moved since the last frame
    const Scene* prev = nullptr;
    const Scene& Prev() const { return prev ? *prev : *this; }

    void Build(const zr_scene_desc& d, bool forceBVH_)
    {
        vertices.assign(d.vertices, d.vertices + d.num_vertices);
        indices.assign(d.indices, d.indices + d.num_indices);
        instances.assign(d.instances, d.instances + d.num_instances);
        materials.assign(d.materials, d.materials + d.num_materials);
        if (d.num_emissives) emissives.assign(d.emissives, d.emissives + d.num_emissives);
        size_t nrho = (size_t)d.rho_dim[0] * d.rho_dim[1] * d.rho_dim[2];
        rho.assign(d.rho_lut, d.rho_lut + nrho);
        rhoLUT.data = rho.data(); rhoLUT.dim[0] = d.rho_dim[0]; rhoLUT.dim[1] = d.rho_dim[1]; rhoLUT.dim[2] = d.rho_dim[2];
        if (d.num_textures)
        {
            texDescs.assign(d.textures, d.textures + d.num_textures);
            texels.assign(d.texels, d.texels + d.texel_bytes);
        }
        tex.descs = texDescs.data(); tex.texels = texels.data(); tex.srgb = zr_srgb_to_linear_table; tex.count = (uint32_t)texDescs.size();

        instMask.assign(d.instance_mask, d.instance_mask + d.num_instances);
        instNumTris.assign(d.instance_num_tris, d.instance_num_tris + d.num_instances);
        forceBVH = forceBVH_;
        BuildTris(d.instance_to_world);
    }
    std::vector<uint8_t> instMask; std::vector<uint32_t> instNumTris; bool forceBVH = false;

    // zr_scene_update_instances on the oracle side: new per-frame MeshInstance records + object-to-world matrices (the caller fills the
    // Prev* fields like TLAS::FillMeshInstanceData does); world-space triangles and the BVH are rebuilt.  The caller keeps a copy of the scene
    // as it was (`prev`) for the passes that bind the previous acceleration structure.
    void UpdateInstances(const zr_mesh_instance* inst, const float* instance_to_world, uint32_t n)
    {
        instances.assign(inst, inst + n);
        BuildTris(instance_to_world);
    }

    void BuildTris(const float* instance_to_world)
    {
        const struct { const zr_mesh_instance* instances; const float* instance_to_world; const uint32_t* indices; const zr_vertex* vertices;
                       const uint32_t* instance_num_tris; const uint8_t* instance_mask; uint32_t num_instances; }
            d = {instances.data(), instance_to_world, indices.data(), vertices.data(), instNumTris.data(), instMask.data(), (uint32_t)instances.size()};
        const bool forceBVH_ = forceBVH;
        tris.clear();
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
                WorldTri t;
                for (int r = 0; r < 3; r++) { t.v0[r] = w[0][r]; t.e1[r] = w[1][r] - w[0][r]; t.e2[r] = w[2][r] - w[0][r]; }
                t.mesh_idx = i; t.prim_idx = p; t.mask = d.instance_mask[i];
                tris.push_back(t);
            }
        }
        bruteForce = !forceBVH_ && tris.size() <= 256;
        if (!bruteForce) BuildBVH();
    }

    // ---- oracle BVH: median split on the largest centroid axis, leaves of <= 4 triangles ----
    void TriBounds(const WorldTri& t, float bmin[3], float bmax[3]) const
    {
        for (int r = 0; r < 3; r++)
        {
            float a = t.v0[r], b = t.v0[r] + t.e1[r], c = t.v0[r] + t.e2[r];
            // e1 = v1 - v0 was rounded, so v0 + e1 may differ from v1 by an ulp: pad the box by one ulp each side
            float lo = zr_min(a, zr_min(b, c)), hi = zr_max(a, zr_max(b, c));
            bmin[r] = Math::PrevFloat32(lo); bmax[r] = Math::NextFloat32(hi);
        }
    }
    uint32_t BuildNode(uint32_t first, uint32_t count, std::vector<float>& cent)
    {
        BVHNode n;
        for (int r = 0; r < 3; r++) { n.bmin[r] = ZR_FLT_MAX; n.bmax[r] = -ZR_FLT_MAX; }
        float cmin[3] = {ZR_FLT_MAX, ZR_FLT_MAX, ZR_FLT_MAX}, cmax[3] = {-ZR_FLT_MAX, -ZR_FLT_MAX, -ZR_FLT_MAX};
        for (uint32_t i = first; i < first + count; i++)
        {
            float lo[3], hi[3]; TriBounds(tris[triOrder[i]], lo, hi);
            for (int r = 0; r < 3; r++)
            {
                n.bmin[r] = zr_min(n.bmin[r], lo[r]); n.bmax[r] = zr_max(n.bmax[r], hi[r]);
                float c = cent[3 * triOrder[i] + r];
                cmin[r] = zr_min(cmin[r], c); cmax[r] = zr_max(cmax[r], c);
            }
        }
        n.left = n.right = 0; n.first = first; n.count = count;
        uint32_t idx = (uint32_t)nodes.size();
        nodes.push_back(n);
        if (count <= 4) return idx;
        int axis = 0; float ext = cmax[0] - cmin[0];
        for (int r = 1; r < 3; r++) if (cmax[r] - cmin[r] > ext) { ext = cmax[r] - cmin[r]; axis = r; }
        uint32_t mid = first + count / 2;
        std::nth_element(triOrder.begin() + first, triOrder.begin() + mid, triOrder.begin() + first + count,
            [&](uint32_t a, uint32_t b) {
                float ca = cent[3 * a + axis], cb = cent[3 * b + axis];
                return ca < cb || (ca == cb && a < b); });
        uint32_t l = BuildNode(first, mid - first, cent);
        uint32_t r = BuildNode(mid, first + count - mid, cent);
        nodes[idx].left = l; nodes[idx].right = r; nodes[idx].count = 0;
        return idx;
    }
    void BuildBVH()
    {
        triOrder.resize(tris.size());
        std::vector<float> cent(3 * tris.size());
        for (size_t i = 0; i < tris.size(); i++)
        {
            triOrder[i] = (uint32_t)i;
            for (int r = 0; r < 3; r++) cent[3 * i + r] = tris[i].v0[r] + (tris[i].e1[r] + tris[i].e2[r]) * (1.0f / 3.0f);
        }
        nodes.clear();
        nodes.reserve(tris.size());
        BuildNode(0, (uint32_t)tris.size(), cent);
    }

    struct RawHit { bool hit; float t, u, v; uint32_t tri; };

    // One candidate test.  zr_ray_tri applies the ray's own (tmin, tmax); the closest-hit rule on top of it is:
    // smaller t wins, equal t goes to the smaller global triangle index (ABI tie-break, include/zr_intersect.h).
    // GBufferRT_Inline.hlsl:37-70: the alpha test primary rays run on candidate hits of non-opaque geometry
    // (g_samLinearWrap, mip 0).  true = the candidate is committed.
    bool TestOpacity(uint32_t meshIdx, uint32_t primIdx, float bu, float bv) const
    {
        const zr_mesh_instance& md = instances[meshIdx];
        const float alphaFactor = (float)(md.alpha_factor_cutoff & 0xffu) / 255.0f;      // Math::UnpackRG
        const float cutoff = (float)(md.alpha_factor_cutoff >> 8) / 255.0f;
        if (cutoff == 1.0f) return false;
        float alpha = alphaFactor;
        if (md.base_color_tex != 0xffffu)
        {
            uint32_t tri = primIdx * 3 + md.base_idx_offset;
            const zr_vertex& V0 = vertices[indices[tri] + md.base_vtx_offset];
            const zr_vertex& V1 = vertices[indices[tri + 1] + md.base_vtx_offset];
            const zr_vertex& V2 = vertices[indices[tri + 2] + md.base_vtx_offset];
            float u = V0.uv[0] + bu * (V1.uv[0] - V0.uv[0]) + bv * (V2.uv[0] - V0.uv[0]);
            float v = V0.uv[1] + bu * (V1.uv[1] - V0.uv[1]) + bv * (V2.uv[1] - V0.uv[1]);
            float c[4];
            zr_tex_sample_level(&tex, baseColorMapsOffset + md.base_color_tex, u, v, 0.0f, c);
            alpha *= c[3];
        }
        if (alpha < cutoff) return false;
        return true;
    }

    inline void TestTri(uint32_t ti, float3 o, float3 d, float tmin, float rayTmax, uint32_t mask, RawHit& best, bool filterID = false, uint32_t ignoreID = 0,
        bool alphaTest = false) const
    {
        const WorldTri& T = tris[ti];
        if (!(T.mask & mask)) return;
        if (filterID) { uint32_t kx = T.mesh_idx, ky = 0, kz = T.prim_idx; zr_pcg3d(&kx, &ky, &kz); if (kx == ignoreID) return; }
        float t, u, v;
        if (zr_ray_tri(o.x, o.y, o.z, d.x, d.y, d.z, T.v0[0], T.v0[1], T.v0[2], T.e1[0], T.e1[1], T.e1[2],
                T.e2[0], T.e2[1], T.e2[2], tmin, rayTmax, &t, &u, &v))
        {
            if (alphaTest && (T.mask & ZR_INSTANCE_NON_OPAQUE) && !TestOpacity(T.mesh_idx, T.prim_idx, u, v)) return;
            if (!best.hit || t < best.t || (t == best.t && ti < best.tri))
            { best.hit = true; best.t = t; best.u = u; best.v = v; best.tri = ti; }
        }
    }

    // closest hit (anyHit=false) or first accepted hit (anyHit=true) with tmin < t < tmax over triangles whose
    // instance mask intersects `mask`
    // filterID: triangles whose hashed ID equals ignoreID are not candidates (approximate shadow segments, see Visibility_Segment)
    // alphaTest: candidates on ZR_INSTANCE_NON_OPAQUE geometry must pass TestOpacity (primary rays only)
    RawHit Trace(float3 o, float3 d, float tmin, float tmax, uint32_t mask, bool anyHit, bool filterID = false, uint32_t ignoreID = 0,
        bool alphaTest = false) const
    {
        RawHit best; best.hit = false; best.t = tmax; best.u = best.v = 0; best.tri = 0xffffffffu;
        if (bruteForce)
        {
            for (uint32_t i = 0; i < tris.size(); i++)
            {
                TestTri(i, o, d, tmin, tmax, mask, best, filterID, ignoreID, alphaTest);
                if (anyHit && best.hit) return best;
            }
            return best;
        }
        float idx = zr_safe_rcp_dir(d.x), idy = zr_safe_rcp_dir(d.y), idz = zr_safe_rcp_dir(d.z);
        uint32_t stack[64]; int sp = 0; stack[sp++] = 0;
        while (sp)
        {
            const BVHNode& n = nodes[stack[--sp]];
            float te;
            // cull against the current best t (inclusive, so equal-t candidates for the tie-break are still visited)
            if (!zr_ray_box(o.x, o.y, o.z, idx, idy, idz, n.bmin[0], n.bmin[1], n.bmin[2], n.bmax[0], n.bmax[1], n.bmax[2],
                    tmin, best.t, &te)) continue;
            if (n.count)
            {
                for (uint32_t i = n.first; i < n.first + n.count; i++)
                {
                    TestTri(triOrder[i], o, d, tmin, tmax, mask, best, filterID, ignoreID, alphaTest);
                    if (anyHit && best.hit) return best;
                }
            }
            else { stack[sp++] = n.left; stack[sp++] = n.right; }
        }
        return best;
    }
};

namespace RtRayQuery {

static const float T_MIN_REFL_RAY = 1e-6f;
static const float T_MIN_TR_RAY = 5e-5f;

// RayQuery.hlsli:15-144
struct Hit
{
    float t; float2 uv; float3 normal; uint32_t ID; uint32_t meshIdx; bool hit;
    Math::TriDifferentials triDiffs; uint16_t matIdx;
    uint32_t primIdx;   // oracle extra (debug)
};

// vertex fetch + TRS + tri differentials shared by Hit::FindClosest and Hit_Emissive::ToHitInfo
static inline void FillHitFromTriangle(const Scene& sc, uint32_t meshIdx, uint32_t primIdx, float2 bary, bool Curr, bool wantID, Hit& ret)
{
    const zr_mesh_instance& md = sc.instances[meshIdx];
    ret.matIdx = md.mat_idx;
    ret.meshIdx = meshIdx;
    ret.primIdx = primIdx;
    uint32_t tri = primIdx * 3 + md.base_idx_offset;
    uint32_t i0 = sc.indices[tri] + md.base_vtx_offset;
    uint32_t i1 = sc.indices[tri + 1] + md.base_vtx_offset;
    uint32_t i2 = sc.indices[tri + 2] + md.base_vtx_offset;
    const zr_vertex& V0 = sc.vertices[i0];
    const zr_vertex& V1 = sc.vertices[i1];
    const zr_vertex& V2 = sc.vertices[i2];

    float3 trn = f3(md.translation);
    if (!Curr)
        trn = trn - f3(zr_f16_to_f32(md.d_translation[0]), zr_f16_to_f32(md.d_translation[1]), zr_f16_to_f32(md.d_translation[2]));
    float4 q = Math::DecodeNormalized4(Curr ? md.rotation : md.prev_rotation);
    const uint16_t* sh = Curr ? md.scale : md.prev_scale;
    float3 s = f3(zr_f16_to_f32(sh[0]), zr_f16_to_f32(sh[1]), zr_f16_to_f32(sh[2]));
    q = normalize(q);

    float tmp = 1 - bary.x - bary.y;
    float2 uv = {zr_fma(bary.y, V2.uv[0], tmp * V0.uv[0]), zr_fma(bary.y, V2.uv[1], tmp * V0.uv[1])};
    uv = {zr_fma(bary.x, V1.uv[0], uv.x), zr_fma(bary.x, V1.uv[1], uv.y)};
    ret.uv = uv;

    float3 v0_n = Math::DecodeOct32(V0.normal);
    float3 v1_n = Math::DecodeOct32(V1.normal);
    float3 v2_n = Math::DecodeOct32(V2.normal);
    float3 hitNormal = mad3(bary.y, v2_n, tmp * v0_n);
    hitNormal = mad3(bary.x, v1_n, hitNormal);
    const float3 scaleInv = f3(1.0f / s.x, 1.0f / s.y, 1.0f / s.z);
    hitNormal *= scaleInv;
    hitNormal = Math::RotateVector(hitNormal, q);
    hitNormal = normalize(hitNormal);
    ret.normal = hitNormal;

    float3 v0W = Math::TransformTRS(f3(V0.pos), trn, q, s);
    float3 v1W = Math::TransformTRS(f3(V1.pos), trn, q, s);
    float3 v2W = Math::TransformTRS(f3(V2.pos), trn, q, s);
    float3 n0W = normalize(Math::RotateVector(v0_n * scaleInv, q));
    float3 n1W = normalize(Math::RotateVector(v1_n * scaleInv, q));
    float3 n2W = normalize(Math::RotateVector(v2_n * scaleInv, q));
    ret.triDiffs = Math::TriDifferentials::Compute(v0W, v1W, v2W, n0W, n1W, n2W,
        f2(V0.uv[0], V0.uv[1]), f2(V1.uv[0], V1.uv[1]), f2(V2.uv[0], V2.uv[1]));

    ret.ID = 0xffffffffu;
    if (wantID)
    {
        // static BLAS: GeometryIndex = meshIdx, InstanceID = 0 (RtAccelerationStructure.cpp:393-405)
        uint32_t kx = meshIdx, ky = 0, kz = primIdx;
        zr_pcg3d(&kx, &ky, &kz);
        ret.ID = kx;
    }
}

static inline Hit FindClosest(const Scene& sc, bool ID, bool Curr, float3 pos, float3 normal, float3 wi, bool transmissive)
{
    Hit ret; ret.hit = false; ret.ID = 0xffffffffu;
    float ndotwi = dot(normal, wi);
    if (ndotwi == 0) return ret;
    bool wiBackface = ndotwi < 0;
    if (wiBackface)
    {
        if (!transmissive) return ret;
        normal = -normal;
    }
    const float3 adjustedOrigin = RT::OffsetRayRTG(pos, normal);
    sc.counters.n_closest++;
    Scene::RawHit h = sc.Trace(adjustedOrigin, wi, wiBackface ? T_MIN_TR_RAY : T_MIN_REFL_RAY, ZR_FLT_MAX, ZR_SUBGROUP_ALL, false);
    if (h.hit)
    {
        const WorldTri& T = sc.tris[h.tri];
        ret.t = h.t;
        FillHitFromTriangle(sc, T.mesh_idx, T.prim_idx, f2(h.u, h.v), Curr, ID, ret);
        ret.hit = true;
    }
    return ret;
}

// RayQuery.hlsli:146-299
struct Hit_Emissive
{
    bool hit; float t; uint32_t geoIdx, insID, primIdx, emissiveTriIdx; float2 bary; float3 lightPos;
    bool HitWasEmissive() const { return emissiveTriIdx != 0xffffffffu; }

    static Hit_Emissive FindClosest(const Scene& sc, float3 pos, float3 normal, float3 wi, bool transmissive)
    {
        Hit_Emissive ret; ret.hit = false; ret.emissiveTriIdx = 0xffffffffu;
        bool wiBackface = dot(normal, wi) <= 0;
        if (wiBackface)
        {
            if (transmissive) normal = normal * -1.0f;
            else return ret;
        }
        const float3 adjustedOrigin = RT::OffsetRayRTG(pos, normal);
        sc.counters.n_closest++;
        Scene::RawHit h = sc.Trace(adjustedOrigin, wi, wiBackface ? T_MIN_TR_RAY : T_MIN_REFL_RAY, ZR_FLT_MAX, ZR_SUBGROUP_ALL, false);
        if (h.hit)
        {
            const WorldTri& T = sc.tris[h.tri];
            ret.hit = true; ret.bary = f2(h.u, h.v); ret.t = h.t;
            ret.geoIdx = T.mesh_idx; ret.insID = 0; ret.primIdx = T.prim_idx;
            const zr_mesh_instance& md = sc.instances[T.mesh_idx];
            if (md.base_emissive_tri_offset == 0xffffffffu) return ret;
            ret.emissiveTriIdx = md.base_emissive_tri_offset + T.prim_idx;
            ret.lightPos = mad3(h.t, wi, adjustedOrigin);
        }
        return ret;
    }
    Hit ToHitInfo(const Scene& sc, bool Curr) const
    {
        Hit hi; hi.hit = hit; hi.t = t;
        if (!hit) return hi;
        FillHitFromTriangle(sc, geoIdx + insID, primIdx, bary, Curr, true, hi);
        return hi;
    }
};

// RayQuery.hlsli:302-334
static inline bool Visibility_Ray(const Scene& sc, float3 origin, float3 wi, float3 normal, bool transmissive)
{
    bool wiBackface = dot(normal, wi) <= 0;
    if (wiBackface)
    {
        if (transmissive) normal = normal * -1.0f;
        else return false;
    }
    float3 adjustedOrigin = RT::OffsetRayRTG(origin, normal);
    float tmin = Math::Lerp(0.0f, 8e-5f, dot(normal, wi));
    sc.counters.n_shadow++;
    Scene::RawHit h = sc.Trace(adjustedOrigin, wi, tmin, ZR_FLT_MAX, ZR_SUBGROUP_ALL, true);
    return !h.hit;
}

// RayQuery.hlsli:337-406
static inline bool Visibility_Segment(const Scene& sc, bool approximate, float3 origin, float3 wi, float rayT, float3 normal,
    uint32_t triID, bool transmissive)
{
    if (triID == 0xffffffffu) return false;
    if (rayT < 1e-6f) return false;
    float ndotwi = dot(normal, wi);
    if (ndotwi == 0) return false;
    bool wiBackface = ndotwi < 0;
    if (wiBackface)
    {
        if (transmissive) normal = normal * -1.0f;
        else return false;
    }
    const float3 adjustedOrigin = RT::OffsetRayRTG(origin, normal);
    sc.counters.n_shadow++;
    const float tminv = 3e-6f;
    Scene::RawHit h;
    if (approximate)
    {
        // RAY_FLAG_ACCEPT_FIRST_HIT_AND_END_SEARCH + "hit ID == light ID -> visible" (RayQuery.hlsli:372-405) depends on which
        // hit the driver finds first.  Pinned order-independently: triangles carrying the target's ID are not occluders,
        // any other hit inside the shortened segment occludes.
        float tmax = Math::PrevFloat32(rayT * 0.999f - Math::NextFloat32(tminv));
        h = sc.Trace(adjustedOrigin, wi, tminv, tmax, ZR_SUBGROUP_NON_EMISSIVE, true, true, triID);
        return !h.hit;
    }
    else
        h = sc.Trace(adjustedOrigin, wi, tminv, rayT, ZR_SUBGROUP_NON_EMISSIVE, false);
    if (h.hit)
    {
        const WorldTri& T = sc.tris[h.tri];
        uint32_t kx = T.mesh_idx, ky = 0, kz = T.prim_idx;
        zr_pcg3d(&kx, &ky, &kz);
        return triID == kx;
    }
    return true;
}

// RayQuery.hlsli:408-450: the two TexSampler policies.  Anisotropic = SampleGrad(samp, uv, ddx, ddy); Isotropic =
// SampleLevel(samp, uv, log2(max(dd.x * w, dd.y * h))) with dd = uv_grads.xy only (used by the reconnection shift,
// Shift.hlsli:519).  Filtering itself is zr_texture.h.
enum class TexSampler { Anisotropic, Isotropic };
static inline void SampleMaterialTex(const Scene& sc, uint32_t tex, TexSampler ts, float2 uv, float4 g, float out[4])
{
    if (ts == TexSampler::Anisotropic) { zr_tex_sample_grad_filter(&sc.tex, tex, sc.texFilter, uv.x, uv.y, g.x, g.y, g.z, g.w, out); return; }
    const zr_texture_desc& d = sc.tex.descs[tex];
    float mip = zr_log2(zr_max(g.x * (float)d.width, g.y * (float)d.height));
    zr_tex_sample_level(&sc.tex, tex, uv.x, uv.y, mip, out);
}

// RayQuery.hlsli:452-524
static inline bool GetMaterialData(const Scene& sc, float3 wo, float eta_curr, float4 uv_grads, Hit& hitInfo,
    BSDF::ShadingData& surface, float& eta, TexSampler ts = TexSampler::Anisotropic)
{
    Mat mat; mat.m = sc.materials[hitInfo.matIdx];
    const bool hitBackface = dot(wo, hitInfo.normal) < 0;
    eta = DEFAULT_ETA_MAT;
    if (!mat.DoubleSided() && hitBackface) return false;
    if (mat.DoubleSided() && hitBackface)
    {
        hitInfo.normal = hitInfo.normal * -1.0f;
        hitInfo.triDiffs.dndu = hitInfo.triDiffs.dndu * -1.0f;
        hitInfo.triDiffs.dndv = hitInfo.triDiffs.dndv * -1.0f;
    }
    float3 baseColor = mat.GetBaseColorFactor();
    float metallic = mat.Metallic() ? 1.0f : 0.0f;
    float roughness = mat.GetSpecularRoughness();
    bool tr = mat.Transmissive();
    eta = mat.GetSpecularIOR();
    float trDepth = tr ? mat.GetTransmissionDepth() : 0;
    const uint32_t baseColorTex = mat.GetBaseColorTex();
    const uint32_t metallicRoughnessTex = mat.GetMetallicRoughnessTex();
    if ((trDepth == 0) && (baseColorTex != ZR_INVALID_TEX))
    {
        float c[4];
        SampleMaterialTex(sc, sc.baseColorMapsOffset + baseColorTex, ts, hitInfo.uv, uv_grads, c);
        baseColor = baseColor * f3(c[0], c[1], c[2]);
    }
    if (metallicRoughnessTex != ZR_INVALID_TEX)
    {
        float c[4];
        SampleMaterialTex(sc, sc.mrMapsOffset + metallicRoughnessTex, ts, hitInfo.uv, uv_grads, c);
        metallic *= c[0];
        roughness *= c[1];
    }
    float eta_next = eta_curr == ETA_AIR ? eta : ETA_AIR;
    float subsurface = mat.ThinWalled() ? zr_round_f16(mat.GetSubsurface()) : 0;
    float coat_weight = mat.GetCoatWeight();
    float3 coat_color = mat.GetCoatColor();
    float coat_roughness = mat.GetCoatRoughness();
    float coat_ior = mat.GetCoatIOR();
    surface = BSDF::ShadingData::Init(hitInfo.normal, wo, metallic >= MIN_METALNESS_METAL, roughness, baseColor,
        eta_curr, eta_next, tr, trDepth, subsurface, coat_weight, coat_color, coat_roughness, coat_ior);
    return true;
}

} // namespace RtRayQuery

namespace Light {

enum class TYPE : uint16_t { NONE = 0, SUN = 1, SKY = 2, EMISSIVE = 3 };

// LightSource.hlsli:72-98
struct AliasTableSample
{
    uint32_t idx; float pdf;
    static AliasTableSample get(const Scene& sc, uint32_t numEmissiveTriangles, RNG& rng)
    {
        AliasTableSample ret;
        uint32_t u0 = rng.UniformUintBounded(numEmissiveTriangles);
        const zr_alias_entry& s = sc.alias[u0];
        if (rng.Uniform() < s.p_curr) { ret.pdf = s.cached_p_orig; ret.idx = u0; return ret; }
        ret.pdf = s.cached_p_alias; ret.idx = s.alias;
        return ret;
    }
};

// LightSource.hlsli:109-137
struct EmissiveTriSample
{
    float3 pos, normal; float2 bary; float pdf;
    static EmissiveTriSample get(float3 pos, const EmTri& tri, RNG& rng, bool reverseNormalIfTwoSided = true)
    {
        EmissiveTriSample ret;
        float2 u = rng.Uniform2D();
        ret.bary = Sampling::UniformSampleTriangle(u);
        const float3 vtx0 = tri.Vtx0();
        const float3 vtx1 = tri.V1();
        const float3 vtx2 = tri.V2();
        ret.pos = (1.0f - ret.bary.x - ret.bary.y) * vtx0 + ret.bary.x * vtx1 + ret.bary.y * vtx2;
        ret.normal = cross(vtx1 - vtx0, vtx2 - vtx0);
        bool normalIs0 = dot(ret.normal, ret.normal) == 0;
        float twoArea = length(ret.normal);
        ret.pdf = normalIs0 ? 0.0f : 2.0f / twoArea;
        ret.normal = normalIs0 ? ret.normal : ret.normal / twoArea;
        ret.normal = reverseNormalIfTwoSided && tri.IsDoubleSided() && dot(pos - ret.pos, ret.normal) < 0 ?
            -ret.normal : ret.normal;
        return ret;
    }
};

// LightSource.hlsli:99-106 + the decode every USE_PRESAMPLED_SETS branch repeats (ReSTIR_GI_NEE.hlsli:68-85, ReSTIR_PT_NEE.hlsli:217-236)
struct PresampledLight { float3 pos, normal, le; float pdf; uint32_t idx, ID; bool twoSided; };
static inline PresampledLight SamplePresampledSet(const Scene& sc, uint32_t sampleSetIdx, float3 shadingPos, RNG& rng)
{
    uint32_t u = rng.UniformUintBounded_Faster(sc.sampleSetSize);
    const zr_presampled_tri& t = sc.sampleSets[(size_t)sampleSetIdx * sc.sampleSetSize + u];
    PresampledLight r;
    r.pos = f3(t.pos); r.normal = Math::DecodeOct32(t.normal);
    r.le = f3(zr_f16_to_f32(t.le[0]), zr_f16_to_f32(t.le[1]), zr_f16_to_f32(t.le[2]));
    r.pdf = t.pdf; r.idx = t.idx; r.ID = t.id; r.twoSided = t.two_sided != 0;
    if (r.twoSided && dot(shadingPos - r.pos, r.normal) < 0) r.normal *= -1.0f;
    return r;
}

// LightSource.hlsli:202-223 (default sampler g_samPointWrap: the mip-0 texel under texUV)
static inline float3 Le_EmissiveTriangle(const Scene& sc, const EmTri& tri, float2 bary)
{
    const float3 emissiveFactor = tri.GetFactor();
    const float emissiveStrength = tri.GetStrength();
    float3 le = emissiveFactor * emissiveStrength;
    if (Math::Luminance(le) == 0) return f3(0.0f);
    const uint32_t emissiveTex = tri.GetTex();
    if (emissiveTex != ZR_INVALID_TEX)
    {
        float2 texUV = (1.0f - bary.x - bary.y) * tri.UV0() + bary.x * tri.UV1() + bary.y * tri.UV2();
        float c[4];
        zr_tex_point(&sc.tex, sc.emissiveMapsOffset + emissiveTex, texUV.x, texUV.y, c);
        le = le * f3(c[0], c[1], c[2]);
    }
    return le;
}

} // namespace Light
} // namespace zro
