// ORACLE tooling -- test infrastructure only.
// The reference's auxiliary passes from its own shaders (compiled by ref_pass_aux.cpp), each driven by a restatement of its host code:
//   PreLighting::Render (PreLighting.cpp:318-449: EstimateTriPower over ceil(n / 8) groups, PresampleEmissives over ceil(samples / 64),
//   BuildLightVoxelGrid over dim.x x dim.y x dim.z groups), Sky::Render (Sky.cpp:121-146), Compositing::Render (Compositing.cpp:82-150:
//   Compositing, then FireflyFilter on the same texture), TAA::Render (TAA.cpp:89-126).  libzref_aux.so
#include "ref_pass_common.h"
#include "ref_dispatch.h"
namespace hlsl {
#include "ZetaRenderPass/Common/FrameConstants.h"
#include "ZetaRenderPass/PreLighting/PreLighting_Common.h"
#include "ZetaRenderPass/Sky/Sky_Common.h"
#include "ZetaRenderPass/Compositing/Compositing_Common.h"
#include "ZetaRenderPass/TAA/TAA_Common.h"
}
using namespace refpass;
ZREFP_SCENE_API
extern "C" void zrefp_shader_estimate_power(const ZrDispatch*);
extern "C" void zrefp_shader_presample(const ZrDispatch*);
extern "C" void zrefp_shader_build_lvg(const ZrDispatch*);
extern "C" void zrefp_shader_sky_lut(const ZrDispatch*);
extern "C" void zrefp_shader_compositing(const ZrDispatch*);
extern "C" void zrefp_shader_firefly(const ZrDispatch*);
extern "C" void zrefp_shader_taa(const ZrDispatch*);

namespace {
enum : uint32_t { AUX_OUT = SLOT_PASS, AUX_IN0, AUX_IN1, AUX_IN2, AUX_IN3 };
zr_frame_constants WithHeap(const zr_frame_constants* cb)
{
    zr_frame_constants g = *cb;
    g.curr_gbuffer_desc_heap_offset = SLOT_GBUF_CURR; g.prev_gbuffer_desc_heap_offset = SLOT_GBUF_PREV; g.env_map_desc_heap_offset = SLOT_SKY_LUT;
    g.base_color_maps_desc_heap_offset += SLOT_TEXTURES; g.normal_maps_desc_heap_offset += SLOT_TEXTURES;
    g.metallic_roughness_maps_desc_heap_offset += SLOT_TEXTURES; g.emissive_maps_desc_heap_offset += SLOT_TEXTURES;
    return g;
}
DescriptorHeap g_heap;      // passes without a scene
}

extern "C" {
// K2.  halton: the 64 Halton(2, 3) points PreLighting::Init uploads (PreLighting.cpp:236-250)
int zrefp_estimate_power(RefScene* r, const zr_frame_constants* cb, const float* halton64x2, float* power)
{
    BindScene(r);
    zr_frame_constants g = WithHeap(cb);
    const uint32_t n = (uint32_t)r->sc.emissives.size();
    g.num_emissive_triangles = n;
    ZrDispatch d; memset(&d, 0, sizeof(d));
    d.scene = r; d.heap = &r->heap; d.frame_cb = &g;
    d.buf[0] = (void*)r->sc.emissives.data(); d.buf_count[0] = n;
    d.buf[1] = (void*)halton64x2; d.buf_count[1] = ESTIMATE_TRI_POWER_NUM_SAMPLES_PER_TRI;
    d.buf[2] = power; d.buf_count[2] = n;
    d.groups_x = (n + ESTIMATE_TRI_POWER_NUM_TRIS_PER_GROUP - 1) / ESTIMATE_TRI_POWER_NUM_TRIS_PER_GROUP; d.groups_y = 1;
    zrefp_shader_estimate_power(&d);
    return 0;
}

// K3 (needs the alias table: zrefp_scene_set_alias_table)
int zrefp_presample(RefScene* r, const zr_frame_constants* cb, uint32_t numSets, uint32_t setSize, zr_presampled_tri* out)
{
    BindScene(r);
    zr_frame_constants g = WithHeap(cb);
    hlsl::cbPresampling L; L.NumTotalSamples = numSets * setSize;
    ZrDispatch d; memset(&d, 0, sizeof(d));
    d.scene = r; d.heap = &r->heap; d.frame_cb = &g; d.local_cb = &L; d.local_cb_bytes = sizeof(L);
    d.buf[0] = (void*)r->sc.emissives.data(); d.buf_count[0] = (uint32_t)r->sc.emissives.size();
    d.buf[1] = (void*)r->sc.alias.data(); d.buf_count[1] = (uint32_t)r->sc.alias.size();
    d.buf[2] = out; d.buf_count[2] = L.NumTotalSamples;
    d.groups_x = (L.NumTotalSamples + PRESAMPLE_EMISSIVE_GROUP_DIM_X - 1) / PRESAMPLE_EMISSIVE_GROUP_DIM_X; d.groups_y = 1;
    zrefp_shader_presample(&d);
    return 0;
}

// K4.  offset_y: PreLighting::Render leaves cbLVG::Offset_y unset (PreLighting.cpp:423-431); the caller says what the register holds
int zrefp_build_lvg(RefScene* r, const zr_frame_constants* cb, const uint32_t* dim, const float* extents, float offset_y, zr_voxel_sample* out)
{
    BindScene(r);
    zr_frame_constants g = WithHeap(cb);
    hlsl::cbLVG L; memset(&L, 0, sizeof(L));
    L.GridDim_x = dim[0]; L.GridDim_y = dim[1]; L.GridDim_z = dim[2];
    L.Extents_x = extents[0]; L.Extents_y = extents[1]; L.Extents_z = extents[2]; L.Offset_y = offset_y;
    L.NumTotalSamples = NUM_SAMPLES_PER_VOXEL * dim[0] * dim[1] * dim[2];
    ZrDispatch d; memset(&d, 0, sizeof(d));
    d.scene = r; d.heap = &r->heap; d.frame_cb = &g; d.local_cb = &L; d.local_cb_bytes = sizeof(L);
    d.buf[0] = (void*)r->sc.emissives.data(); d.buf_count[0] = (uint32_t)r->sc.emissives.size();
    d.buf[1] = (void*)r->sc.alias.data(); d.buf_count[1] = (uint32_t)r->sc.alias.size();
    d.buf[2] = out; d.buf_count[2] = L.NumTotalSamples;
    d.groups_x = dim[0]; d.groups_y = dim[1]; d.groups_z = dim[2];
    zrefp_shader_build_lvg(&d);
    return 0;
}

// K17: the sky-view LUT, R11G11B10_FLOAT texels
int zrefp_sky_lut(const zr_frame_constants* cb, uint32_t w, uint32_t h, uint32_t* out)
{
    g_heapPtr = &g_heap;
    memset(out, 0, (size_t)w * h * 4);
    BindPlane(g_heap, AUX_OUT, out, w, h, FMT_R11G11B10_FLOAT);
    zr_frame_constants g = *cb;
    hlsl::cbSky L; memset(&L, 0, sizeof(L));
    L.LutWidth = w; L.LutHeight = h; L.LutDescHeapIdx = AUX_OUT;
    ZrDispatch d; memset(&d, 0, sizeof(d));
    d.heap = &g_heap; d.frame_cb = &g; d.local_cb = &L; d.local_cb_bytes = sizeof(L);
    d.groups_x = (w + SKY_VIEW_LUT_THREAD_GROUP_SIZE_X - 1) / SKY_VIEW_LUT_THREAD_GROUP_SIZE_X;
    d.groups_y = (h + SKY_VIEW_LUT_THREAD_GROUP_SIZE_Y - 1) / SKY_VIEW_LUT_THREAD_GROUP_SIZE_Y;
    zrefp_shader_sky_lut(&d);
    return 0;
}

// Compositing (+ FireflyFilter when `firefly`): gb = this frame's G-buffer planes; skyDI / emissiveDI / indirect: RGBA32F or null;
// composited: RGBA32F, read (alpha kept) and written.  The sky-view LUT behind miss pixels is the scene's (zrefp_scene_set_sky_lut).
// FireflyFilter.hlsl filters the composited texture IN PLACE -- a data race on a GPU (a thread may read a neighbour before or after that
// neighbour's own store).  The ABI defines it race-free: every thread reads the texture as Compositing left it; the harness gives the
// filter a read snapshot (TexStorage::readData).
int zrefp_composite(RefScene* r, const zr_frame_constants* cb, const zr_gbuffer_planes* gb, const float* skyDI, const float* emissiveDI, const float* indirect,
    uint32_t flags, int firefly, float* composited)
{
    BindScene(r);
    DescriptorHeap& H = r->heap;
    const uint32_t w = gb->width, h = gb->height;
    BindGBuffer(H, SLOT_GBUF_CURR, gb);
    BindPlane(H, AUX_OUT, composited, w, h, FMT_RGBA32_FLOAT);
    if (skyDI) BindPlane(H, AUX_IN0, (void*)skyDI, w, h, FMT_RGBA32_FLOAT);
    if (emissiveDI) BindPlane(H, AUX_IN1, (void*)emissiveDI, w, h, FMT_RGBA32_FLOAT);
    if (indirect) BindPlane(H, AUX_IN2, (void*)indirect, w, h, FMT_RGBA32_FLOAT);
    zr_frame_constants g = WithHeap(cb);
    hlsl::cbCompositing L; memset(&L, 0, sizeof(L));
    L.SkyDIDescHeapIdx = skyDI ? AUX_IN0 : 0; L.EmissiveDIDescHeapIdx = emissiveDI ? AUX_IN1 : 0; L.IndirectDescHeapIdx = indirect ? AUX_IN2 : 0;
    L.OutputUAVDescHeapIdx = AUX_OUT; L.Flags = flags;
    ZrDispatch d; memset(&d, 0, sizeof(d));
    d.scene = r; d.heap = &H; d.frame_cb = &g; d.local_cb = &L; d.local_cb_bytes = sizeof(L);
    d.groups_x = (w + COMPOSITING_THREAD_GROUP_DIM_X - 1) / COMPOSITING_THREAD_GROUP_DIM_X; d.groups_y = (h + COMPOSITING_THREAD_GROUP_DIM_Y - 1) / COMPOSITING_THREAD_GROUP_DIM_Y;
    zrefp_shader_compositing(&d);
    if (firefly)
    {
        std::vector<float> snapshot(composited, composited + (size_t)w * h * 4);
        H.table[AUX_OUT].readData = snapshot.data();
        hlsl::cbFireflyFilter F; F.CompositedUAVDescHeapIdx = AUX_OUT;
        d.local_cb = &F; d.local_cb_bytes = sizeof(F);
        d.groups_x = (w + FIREFLY_FILTER_THREAD_GROUP_DIM_X - 1) / FIREFLY_FILTER_THREAD_GROUP_DIM_X; d.groups_y = (h + FIREFLY_FILTER_THREAD_GROUP_DIM_Y - 1) / FIREFLY_FILTER_THREAD_GROUP_DIM_Y;
        zrefp_shader_firefly(&d);
        H.table[AUX_OUT].readData = nullptr;
    }
    return 0;
}

// TAA: signal RGBA32F (the composited texture), depth R32F + motion RG16_SNORM of the current G-buffer, history / output RGBA16F
int zrefp_taa(const zr_frame_constants* cb, const float* signal, const float* depth, const uint32_t* motion, const uint16_t* prevOut, uint16_t* currOut,
    uint32_t w, uint32_t h, float blendWeight, int temporalValid)
{
    g_heapPtr = &g_heap;
    BindPlane(g_heap, SLOT_GBUF_CURR + ZR_GB_DEPTH, (void*)depth, w, h, FMT_R32_FLOAT);
    BindPlane(g_heap, SLOT_GBUF_CURR + ZR_GB_MOTION_VECTOR, (void*)motion, w, h, FMT_RG16_SNORM);
    BindPlane(g_heap, AUX_IN0, (void*)signal, w, h, FMT_RGBA32_FLOAT);
    BindPlane(g_heap, AUX_IN1, (void*)prevOut, w, h, FMT_RGBA16_FLOAT);
    BindPlane(g_heap, AUX_OUT, currOut, w, h, FMT_RGBA16_FLOAT);
    zr_frame_constants g = *cb;
    g.curr_gbuffer_desc_heap_offset = SLOT_GBUF_CURR;
    hlsl::cbTAA L; memset(&L, 0, sizeof(L));
    L.BlendWeight = blendWeight; L.InputDescHeapIdx = AUX_IN0; L.PrevOutputDescHeapIdx = AUX_IN1; L.CurrOutputDescHeapIdx = AUX_OUT; L.TemporalIsValid = temporalValid ? 1u : 0u;
    ZrDispatch d; memset(&d, 0, sizeof(d));
    d.heap = &g_heap; d.frame_cb = &g; d.local_cb = &L; d.local_cb_bytes = sizeof(L);
    d.groups_x = (w + TAA_THREAD_GROUP_SIZE_X - 1) / TAA_THREAD_GROUP_SIZE_X; d.groups_y = (h + TAA_THREAD_GROUP_SIZE_Y - 1) / TAA_THREAD_GROUP_SIZE_Y;
    zrefp_shader_taa(&d);
    return 0;
}
}
