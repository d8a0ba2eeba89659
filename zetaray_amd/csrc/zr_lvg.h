// zr_lvg.h -- the light voxel grid on the device.
//
// Restates: K4 Source/ZetaRenderPass/PreLighting/BuildLightVoxelGrid.hlsl:21-162 (one 64-thread group per voxel = one wave64 here:
// per-thread RIS over 6 alias-table candidates with target = luminance / distance^2; pdf = target / (group mean of the RIS weights));
// Common/LightVoxelGrid.hlsli:8-70 (camera-space grid addressing around the camera, jittered lookup).
// Pinned: WaveActiveSum over the group = the canonical 64-lane butterfly; cbLVG.Offset_y (left uninitialised by PreLighting.cpp:410-417)
// = the lookup's y offset.  HBM: 32 B written per sample (21 MB for the reference's 32 x 8 x 40 grid), ~6 x (16 + 48) B gathered.
#pragma once
#include "zr_dev_scene.h"

namespace zr {

ZR_HD uint32_t LvgFlatten(const int v[3], const uint32_t dim[3]) { return (uint32_t)v[2] * dim[0] * dim[1] + (uint32_t)v[1] * dim[0] + (uint32_t)v[0]; }

ZR_HD V3 LvgVoxelCenter(const int voxelIdx[3], const uint32_t gridDim[3], V3 ext, const float* viewInv, float offset_y)   // :15-34
{
    float c[3]; const float e[3] = {ext.x, ext.y, ext.z};
    for (int a = 0; a < 3; a++)
    {
        const int dimDiv2 = (int)gridDim[a] >> 1;
        int cs = voxelIdx[a] - dimDiv2;
        cs += voxelIdx[a] < dimDiv2 ? 1 : 0;
        if (a == 1) cs *= -1;
        const float corner = (float)cs * 2 * e[a];
        const float s = SignNotZero((float)cs);
        c[a] = corner + e[a] * s;
    }
    c[1] += offset_y;
    return v3(viewInv[0] * c[0] + viewInv[1] * c[1] + viewInv[2] * c[2] + viewInv[3] * 1.0f,
              viewInv[4] * c[0] + viewInv[5] * c[1] + viewInv[6] * c[2] + viewInv[7] * 1.0f,
              viewInv[8] * c[0] + viewInv[9] * c[1] + viewInv[10] * c[2] + viewInv[11] * 1.0f);
}
ZR_HD bool LvgMapPosToVoxel(V3 pos, const uint32_t gridDim[3], V3 ext, const float* view, int idx[3], float offset_y)   // :36-55
{
    float pv[3] = {view[0] * pos.x + view[1] * pos.y + view[2] * pos.z + view[3] * 1.0f,
                   view[4] * pos.x + view[5] * pos.y + view[6] * pos.z + view[7] * 1.0f,
                   view[8] * pos.x + view[9] * pos.y + view[10] * pos.z + view[11] * 1.0f};
    pv[1] -= offset_y;
    const float e[3] = {ext.x, ext.y, ext.z};
    float voxel[3];
    for (int a = 0; a < 3; a++)
    {
        voxel[a] = zr_floor(zr_abs(pv[a]) / (2 * e[a]));
        if (voxel[a] >= (float)((int)gridDim[a] >> 1)) return false;
    }
    for (int a = 0; a < 3; a++) voxel[a] *= SignNotZero(pv[a]);
    voxel[1] *= -1;
    for (int a = 0; a < 3; a++) idx[a] = (int)voxel[a] + ((int)gridDim[a] >> 1);
    idx[0] -= pv[0] < 0 ? 1 : 0; idx[1] -= pv[1] >= 0 ? 1 : 0; idx[2] -= pv[2] < 0 ? 1 : 0;
    return true;
}
// LVG::Sample, :57-70 (jitter on)
ZR_HD bool LvgSample(const SceneView& sc, V3 pos, const float* view, V3 ext, float offset_y, zr_voxel_sample& s, Rng& rng)
{
    const float ux = rng.Uniform(), uy = rng.Uniform(), uz = rng.Uniform();
    const V3 posJittered = pos + v3(ux * 2 - 1, uy * 2 - 1, uz * 2 - 1) * ext;
    int v[3];
    if (!LvgMapPosToVoxel(posJittered, sc.lvgDim, ext, view, v, offset_y)) return false;
    const uint32_t start = LvgFlatten(v, sc.lvgDim) * ZR_LVG_SAMPLES_PER_VOXEL;
    const uint32_t k = rng.UniformUintBounded_Faster(ZR_LVG_SAMPLES_PER_VOXEL);
    s = sc.lvg[start + k];
    return true;
}

// K4 for thread `gidx` of voxel v: everything before the group reduction.  The caller sums w_sum / numLights over the 64 threads
// and finishes with LvgFinish.
ZR_HD void LvgThread(const SceneView& sc, const zr_frame_constants& g, const uint32_t dim[3], V3 ext, float offset_y, const int v[3],
    uint32_t gidx, zr_voxel_sample& r, float& w_sum, float& target_z, uint32_t& numLights)
{
    const uint32_t gridStart = LvgFlatten(v, dim);
    const V3 voxelCenter = LvgVoxelCenter(v, dim, ext, g.curr_view_inv, offset_y);
    Rng rng; rng.s = zr_pcg(gridStart * ZR_LVG_SAMPLES_PER_VOXEL + gidx + zr_pcg(g.frame_num));
    r.pos[0] = ZR_FLT_MAX; r.pos[1] = ZR_FLT_MAX; r.pos[2] = ZR_FLT_MAX; r.normal[0] = 0; r.normal[1] = 0; r.pdf = 0; r.id = 0xffffffffu;
    r.le[0] = 0; r.le[1] = 0; r.le[2] = 0; r.two_sided = 0;
    w_sum = 0; target_z = 0; numLights = 0;
    for (int i = 0; i < 6; i++)      // NUM_CANDIDATES
    {
        uint32_t u0 = rng.UniformUintBounded(g.num_emissive_triangles);
        const zr_alias_entry ae = sc.alias[u0];
        uint32_t lidx; float lpdfSrc;
        if (rng.Uniform() < ae.p_curr) { lpdfSrc = ae.cached_p_orig; lidx = u0; }
        else { lpdfSrc = ae.cached_p_alias; lidx = ae.alias; }
        const zr_emissive_triangle em = sc.emissives[lidx];
        V2 bary = UniformSampleTriangle(rng.Uniform2D());
        const V3 vtx0 = v3p(em.vtx0), vtx1 = EmV1(em), vtx2 = EmV2(em);
        const V3 lpos = (1.0f - bary.x - bary.y) * vtx0 + bary.x * vtx1 + bary.y * vtx2;
        V3 ln = cross(vtx1 - vtx0, vtx2 - vtx0);
        const bool normalIs0 = dot(ln, ln) == 0;
        const float twoArea = length(ln);
        const float lpdfPos = normalIs0 ? 0.0f : 2.0f / twoArea;
        ln = normalIs0 ? ln : ln / twoArea;
        const V3 le = EmLe(sc, em, bary);
        const V3 d = v3(zr_abs(lpos.x - voxelCenter.x), zr_abs(lpos.y - voxelCenter.y), zr_abs(lpos.z - voxelCenter.z));
        const bool inside = d.x <= ext.x && d.y <= ext.y && d.z <= ext.z;
        V3 lightPos = lpos;
        if (inside)
        {
            const int maxIdx = d.x >= d.y ? (d.x >= d.z ? 0 : 2) : (d.y >= d.z ? 1 : 2);
            if (maxIdx == 0) lightPos.x = ext.x; else if (maxIdx == 1) lightPos.y = ext.y; else lightPos.z = ext.z;
        }
        if (!inside && !EmDoubleSided(em))
        {
            bool back = false;
            for (int c = 0; c < 8; c++)
            {
                const V3 corner = voxelCenter + v3((c & 4) ? 1.0f : -1.0f, (c & 2) ? 1.0f : -1.0f, (c & 1) ? 1.0f : -1.0f) * ext;
                if (dot(corner - lpos, ln) <= 0) back = true;
            }
            if (back) continue;
        }
        const float t = length(lightPos - voxelCenter);
        const float target = Luminance(le) / zr_max(t * t, 1e-6f);
        const float lightPdf = lpdfSrc * lpdfPos;
        const float w = target / zr_max(lightPdf, 1e-6f);
        w_sum += w;
        if (rng.Uniform() < w / zr_max(w_sum, 1e-6f))
        {
            r.pos[0] = lpos.x; r.pos[1] = lpos.y; r.pos[2] = lpos.z;
            const V2 e = EncodeUnitVector(ln);
            r.normal[0] = (uint16_t)FloatToUNorm16(e.x); r.normal[1] = (uint16_t)FloatToUNorm16(e.y);
            r.le[0] = zr_f32_to_f16(le.x); r.le[1] = zr_f32_to_f16(le.y); r.le[2] = zr_f32_to_f16(le.z);
            r.two_sided = EmDoubleSided(em) ? 1 : 0;
            r.id = em.id;
            target_z = target;
        }
        numLights++;
    }
}
ZR_HD void LvgFinish(zr_voxel_sample& r, float target_z, float w_sum_group, uint32_t numLightsGroup)
{
    w_sum_group /= (float)(numLightsGroup & 0xffffu);
    r.pdf = target_z / zr_max(w_sum_group, 1e-6f);
}

} // namespace zr
