// ORACLE tooling -- test infrastructure only.
// Shared host side of the reference PASSES compiled as C++ (ref_pass_*.cpp): scene, descriptor heap, plane registration.
#pragma once
#include <vector>
#include <memory>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include "../../include/zetaray_amd.h"
#include "../../include/zr_srgb_table.h"
#include "../zro_scene.h"           // the oracle's scene container + the ABI's traversal (zr_intersect.h); NOT the oracle's shading code
#include "hlsl_group.h"

#define ZR_GROUPSHARED static

namespace refpass {
using namespace hlsl;

// descriptor-heap slots the drivers use (the reference allocates these dynamically; only the indices in the constant buffers matter)
enum : uint32_t { SLOT_RHO = 0, SLOT_GBUF_CURR = 16, SLOT_GBUF_PREV = 32, SLOT_GBUF_UAV = 48, SLOT_SKY_LUT = 64, SLOT_PASS = 80, SLOT_TEXTURES = 1024 };

static const int kGBufFormats[ZR_GB_COUNT] = { FMT_RGBA8_UNORM, FMT_RG16_UNORM, FMT_RG8_UNORM, FMT_RG16_SNORM, FMT_R11G11B10_FLOAT, FMT_R8_UNORM,
                                               FMT_RGBA16_UINT, FMT_R32_FLOAT, FMT_RGBA32_UINT, FMT_RG32_UINT };

struct RefScene
{
    zro::Scene sc;
    DescriptorHeap heap;
    std::unique_ptr<RefScene> prevHolder;      // the scene as it was before the last zrefp_scene_update_instances (RT_SCENE_BVH_PREV etc.)
};

// EnumToSamplerIdx, IndirectLighting.cpp:21-33: TEXTURE_FILTER (zr_params.tex_filter) -> static sampler index (RendererCore.cpp:547-555)
static inline uint32_t EnumToSamplerIdx(uint32_t f)
{ return f == 0 ? 0u : (f == 1 ? 3u : (f == 2 ? 6u : (f == 3 ? 7u : 5u))); }

static inline void BindPlane(DescriptorHeap& h, uint32_t slot, void* data, uint32_t w, uint32_t ht, int fmt)
{ TexStorage& s = h.table[slot]; s.data = data; s.w = w; s.h = ht; s.d = 1; s.fmt = fmt; }

static inline void BindGBuffer(DescriptorHeap& h, uint32_t base, const zr_gbuffer_planes* p)
{ for (int i = 0; i < ZR_GB_COUNT; i++) BindPlane(h, base + i, p->plane[i], p->width, p->height, kGBufFormats[i]); }

static inline void BindScene(RefScene* r)
{
    g_heapPtr = &r->heap;
    TexStorage& rho = r->heap.table[SLOT_RHO];
    rho.data = (void*)r->sc.rho.data(); rho.w = r->sc.rhoLUT.dim[0]; rho.h = r->sc.rhoLUT.dim[1]; rho.d = r->sc.rhoLUT.dim[2]; rho.fmt = FMT_R16_UNORM;
    for (uint32_t i = 0; i < r->sc.tex.count && SLOT_TEXTURES + i < DescriptorHeap::kSize; i++)
    {
        TexStorage& t = r->heap.table[SLOT_TEXTURES + i];
        t.fmt = FMT_MATERIAL_TEXTURE; t.heap = &r->sc.tex; t.heapIdx = i; t.w = r->sc.tex.descs[i].width; t.h = r->sc.tex.descs[i].height;
    }
    if (!r->sc.skyData.empty()) BindPlane(r->heap, SLOT_SKY_LUT, (void*)r->sc.skyData.data(), r->sc.sky.w, r->sc.sky.h, FMT_R11G11B10_FLOAT);
}
} // namespace refpass

namespace refpass {
// Dispatch(gx x gy thread groups of tx x ty threads): `fn(DTid, Gid, GTid, Gidx)` per thread.  Groups run one after the other; the threads
// of a group run as fibers when the shader has cross-lane operations (wave intrinsics / barriers), else in SV_GroupIndex order.
template<class F> static inline void Dispatch(uint32_t gx, uint32_t gy, uint32_t tx, uint32_t ty, bool fibers, F fn)
{
    static thread_local GroupRunner runner;
    for (uint32_t Gy = 0; Gy < gy; Gy++)
        for (uint32_t Gx = 0; Gx < gx; Gx++)
        {
            auto lane = [&](int i) {
                const uint32_t lx = (uint32_t)i % tx, ly = (uint32_t)i / tx;
                fn(uint3(Gx * tx + lx, Gy * ty + ly, 0), uint3(Gx, Gy, 0), uint3(lx, ly, 0), (uint32_t)i);
            };
            if (fibers) runner.Run((int)(tx * ty), lane);
            else for (int i = 0; i < (int)(tx * ty); i++) { lane(i); FlushPendingRW(); }
        }
}

static inline RefScene* SceneCreate(const zr_scene_desc* d, int force_bvh) { RefScene* r = new RefScene(); r->sc.Build(*d, force_bvh != 0); return r; }
} // namespace refpass

#define ZREFP_SCENE_API \
    extern "C" refpass::RefScene* zrefp_scene_create(const zr_scene_desc* d, int force_bvh) { return refpass::SceneCreate(d, force_bvh); } \
    extern "C" void zrefp_scene_destroy(refpass::RefScene* r) { delete r; } \
    extern "C" int zrefp_scene_update_instances(refpass::RefScene* r, const zr_mesh_instance* inst, const float* xf, uint32_t n) \
    { if (n != r->sc.instances.size()) return -1; r->prevHolder.reset(new refpass::RefScene()); r->prevHolder->sc = r->sc; r->prevHolder->sc.prev = nullptr; \
      r->sc.UpdateInstances(inst, xf, n); return 0; } \
    extern "C" int zrefp_scene_update_emissives(refpass::RefScene* r, const zr_emissive_triangle* t, uint32_t first, uint32_t count) \
    { if ((size_t)first + count > r->sc.emissives.size()) return -1; std::copy(t, t + count, r->sc.emissives.begin() + first); \
      if (r->prevHolder) std::copy(t, t + count, r->prevHolder->sc.emissives.begin() + first); return 0; } \
    extern "C" void zrefp_scene_set_alias_table(refpass::RefScene* r, const zr_alias_entry* e, uint32_t n) { r->sc.alias.assign(e, e + n); } \
    extern "C" void zrefp_scene_set_sample_sets(refpass::RefScene* r, const zr_presampled_tri* e, uint32_t numSets, uint32_t setSize) \
    { r->sc.sampleSets.assign(e, e + (size_t)numSets * setSize); r->sc.sampleSetSize = setSize; } \
    extern "C" void zrefp_scene_set_lvg(refpass::RefScene* r, const zr_voxel_sample* v, const uint32_t* dim, const float* extents, float offsetY) \
    { r->sc.lvgData.assign(v, v + (size_t)dim[0] * dim[1] * dim[2] * 64); for (int a = 0; a < 3; a++) { r->sc.lvgDim[a] = dim[a]; r->sc.lvgExtents[a] = extents[a]; } r->sc.lvgOffsetY = offsetY; } \
    extern "C" void zrefp_scene_set_sky_lut(refpass::RefScene* r, const uint32_t* texels, uint32_t w, uint32_t h) \
    { r->sc.skyData.assign(texels, texels + (size_t)w * h); r->sc.sky.data = r->sc.skyData.data(); r->sc.sky.w = w; r->sc.sky.h = h; }
