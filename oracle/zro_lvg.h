// ORACLE -- test infrastructure only (see zro_math.h header).  PARITY UNPINNED against the reference.
//
// zro_lvg.h: the light voxel grid.
//   K4 PreLighting/BuildLightVoxelGrid.hlsl:21-162 (one 64-thread group per voxel: per-thread RIS over 6 alias-table candidates
//   with target = luminance / distance^2, then pdf = target / (group mean of the RIS weights)); Common/LightVoxelGrid.hlsli:8-70
//   (camera-space grid addressing, jittered lookup); RGI_Util::NEE_Emissive_LVG ReSTIR_GI_NEE.hlsli:121-187.
// Pinned: WaveActiveSum over the 64 threads of a group = the canonical 64-lane butterfly (RPT::WaveSum64); cbLVG.Offset_y, which
// PreLighting.cpp:410-417 leaves uninitialised, = the y offset the lookup uses.
#pragma once
#include "zro_rpt.h"

namespace zro {
namespace LVG {


static inline uint32_t FlattenVoxelIndex(const int v[3], const uint32_t dim[3]) { return (uint32_t)v[2] * dim[0] * dim[1] + (uint32_t)v[1] * dim[0] + (uint32_t)v[0]; }

// LightVoxelGrid.hlsli:15-34
static inline float3 VoxelCenter(const int voxelIdx[3], const uint32_t gridDim[3], float3 voxelExtents, const float* viewInv, float offset_y)
{
    float c[3]; const float e[3] = {voxelExtents.x, voxelExtents.y, voxelExtents.z};
    for (int a = 0; a < 3; a++)
    {
        const int dimDiv2 = (int)gridDim[a] >> 1;
        int cs = voxelIdx[a] - dimDiv2;
        cs += voxelIdx[a] < dimDiv2 ? 1 : 0;
        if (a == 1) cs *= -1;
        const float corner = (float)cs * 2 * e[a];
        const float s = Math::SignNotZero((float)cs);
        c[a] = corner + e[a] * s;
    }
    c[1] += offset_y;
    // mul(viewInv, float4(centerV, 1)): row . (v, 1)
    auto row = [&](int r) { return viewInv[4 * r] * c[0] + viewInv[4 * r + 1] * c[1] + viewInv[4 * r + 2] * c[2] + viewInv[4 * r + 3] * 1.0f; };
    return f3(row(0), row(1), row(2));
}
// LightVoxelGrid.hlsli:36-55
static inline bool MapPosToVoxel(float3 pos, const uint32_t gridDim[3], float3 voxelExtents, const float* view, int idx[3], float offset_y)
{
    auto row = [&](int r) { return view[4 * r] * pos.x + view[4 * r + 1] * pos.y + view[4 * r + 2] * pos.z + view[4 * r + 3] * 1.0f; };
    float pv[3] = {row(0), row(1), row(2)};
    pv[1] -= offset_y;
    const float e[3] = {voxelExtents.x, voxelExtents.y, voxelExtents.z};
    float voxel[3];
    for (int a = 0; a < 3; a++)
    {
        voxel[a] = zr_floor(zr_abs(pv[a]) / (2 * e[a]));
        if (voxel[a] >= (float)((int)gridDim[a] >> 1)) return false;
    }
    for (int a = 0; a < 3; a++) voxel[a] *= Math::SignNotZero(pv[a]);
    voxel[1] *= -1;
    for (int a = 0; a < 3; a++) idx[a] = (int)voxel[a] + ((int)gridDim[a] >> 1);
    idx[0] -= pv[0] < 0 ? 1 : 0; idx[1] -= pv[1] >= 0 ? 1 : 0; idx[2] -= pv[2] < 0 ? 1 : 0;
    return true;
}
// LightVoxelGrid.hlsli:57-70
static inline bool Sample(float3 pos, const Scene& sc, uint32_t numLightsPerVoxel, const float* view, zr_voxel_sample& s, RNG& rng, float3 extents, float offset_y)
{
    const float3 u = rng.Uniform3D();
    float3 posJittered = pos + f3(u.x * 2 - 1, u.y * 2 - 1, u.z * 2 - 1) * extents;
    int v[3];
    if (!MapPosToVoxel(posJittered, sc.lvgDim, extents, view, v, offset_y)) return false;
    const uint32_t start = FlattenVoxelIndex(v, sc.lvgDim) * numLightsPerVoxel;
    const uint32_t k = rng.UniformUintBounded_Faster(numLightsPerVoxel);
    s = sc.lvgData[start + k];
    return true;
}

// K4, BuildLightVoxelGrid.hlsl:57-162: all 64 samples of voxel (vx, vy, vz)
static inline void BuildVoxel(const Scene& sc, const zr_frame_constants& g, const uint32_t dim[3], float3 extents, float offset_y,
    int vx, int vy, int vz, zr_voxel_sample* out)
{
    const int NUM_CANDIDATES = 6;
    const int v[3] = {vx, vy, vz};
    const uint32_t gridStart = FlattenVoxelIndex(v, dim);
    const float3 voxelCenter = VoxelCenter(v, dim, extents, g.curr_view_inv, offset_y);
    float3 corners[8];
    for (int i = 0; i < 8; i++)
        corners[i] = voxelCenter + f3((i & 4) ? 1.0f : -1.0f, (i & 2) ? 1.0f : -1.0f, (i & 1) ? 1.0f : -1.0f) * extents;
    float w_sums[64], targets[64]; uint32_t numLightsGroup = 0;
    for (uint32_t gidx = 0; gidx < 64; gidx++)
    {
        RNG rng = RNG::InitIdx(gridStart * 64 + gidx, g.frame_num);
        zr_voxel_sample r; std::memset(&r, 0, sizeof(r));
        r.pos[0] = r.pos[1] = r.pos[2] = ZR_FLT_MAX; r.id = 0xffffffffu;
        float w_sum = 0, target_z = 0; uint32_t numLights = 0;
        for (int i = 0; i < NUM_CANDIDATES; i++)
        {
            Light::AliasTableSample entry = Light::AliasTableSample::get(sc, g.num_emissive_triangles, rng);
            EmTri tri; tri.t = sc.emissives[entry.idx];
            Light::EmissiveTriSample ls = Light::EmissiveTriSample::get(voxelCenter, tri, rng, false);
            const float3 le = Light::Le_EmissiveTriangle(sc, tri, ls.bary);
            // AdjustLightPos: snap lights inside the voxel to its boundary planes
            const float3 d = abs3(ls.pos - voxelCenter);
            const bool inside = d.x <= extents.x && d.y <= extents.y && d.z <= extents.z;
            float3 lightPos = ls.pos;
            if (inside)
            {
                const int maxIdx = d.x >= d.y ? (d.x >= d.z ? 0 : 2) : (d.y >= d.z ? 1 : 2);
                if (maxIdx == 0) lightPos.x = extents.x; else if (maxIdx == 1) lightPos.y = extents.y; else lightPos.z = extents.z;
            }
            if (!inside && !tri.IsDoubleSided())
            {
                bool back = false;
                for (int c = 0; c < 8; c++) if (dot(corners[c] - ls.pos, ls.normal) <= 0) { back = true; break; }
                if (back) continue;
            }
            const float t = length(lightPos - voxelCenter);
            const float target = Math::Luminance(le) / zr_max(t * t, 1e-6f);
            const float lightPdf = entry.pdf * ls.pdf;
            float w = target / zr_max(lightPdf, 1e-6f);
            w_sum += w;
            if (rng.Uniform() < w / zr_max(w_sum, 1e-6f))
            {
                r.pos[0] = ls.pos.x; r.pos[1] = ls.pos.y; r.pos[2] = ls.pos.z;
                Math::EncodeOct32(ls.normal, r.normal);
                r.le[0] = zr_f32_to_f16(le.x); r.le[1] = zr_f32_to_f16(le.y); r.le[2] = zr_f32_to_f16(le.z);
                r.two_sided = tri.IsDoubleSided() ? 1 : 0;
                r.id = tri.t.id;
                target_z = target;
            }
            numLights++;
        }
        w_sums[gidx] = w_sum; targets[gidx] = target_z; numLightsGroup += numLights;
        out[gidx] = r;
    }
    float w_sum_group = RPT::WaveSum64(w_sums);
    w_sum_group /= (float)(uint16_t)numLightsGroup;
    for (uint32_t gidx = 0; gidx < 64; gidx++) out[gidx].pdf = targets[gidx] / zr_max(w_sum_group, 1e-6f);
}

} // namespace LVG
} // namespace zro
