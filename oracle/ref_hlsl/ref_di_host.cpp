// ORACLE tooling -- test infrastructure only.
// ReSTIR DI from the reference's own shaders, driven by a restatement of the reference's host code.  Compiled twice:
//   -DZR_DI_SKY=0   DirectLighting::Render (RP/DirectLighting/Emissive/DirectLighting.cpp:166-296): ReSTIR_DI_Temporal.hlsl (or _WPS) + ReSTIR_DI_Spatial.hlsl
//   -DZR_DI_SKY=1   SkyDI::Render          (RP/DirectLighting/Sky/SkyDI.cpp:120-250):               SkyDI_Temporal.hlsl + SkyDI_Spatial.hlsl
#include "ref_pass_common.h"
#include "ref_dispatch.h"
namespace hlsl {
#include "ZetaRenderPass/Common/FrameConstants.h"
#if ZR_DI_SKY
#include "ZetaRenderPass/DirectLighting/Sky/SkyDI_Common.h"
#else
#include "ZetaRenderPass/DirectLighting/Emissive/DirectLighting_Common.h"
#endif
}
using namespace refpass;
ZREFP_SCENE_API
extern "C" void zrefp_shader_di_temporal(const ZrDispatch*);
extern "C" void zrefp_shader_di_spatial(const ZrDispatch*);

namespace {
#if ZR_DI_SKY
constexpr int kNumPlanes = 3;
const int kFmt[3] = {FMT_R8_UINT, FMT_RG16_UINT, FMT_RG32_FLOAT};          // SkyDI.h:58-62
typedef hlsl::cb_SkyDI LocalCB;
#else
constexpr int kNumPlanes = 2;
const int kFmt[2] = {FMT_RGBA32_UINT, FMT_RG32_FLOAT};                      // DirectLighting.h:69-72
typedef hlsl::cb_ReSTIR_DI LocalCB;
#endif
// descriptor table: set 0 SRVs, set 0 UAVs, set 1 SRVs, set 1 UAVs, TARGET_UAV, FINAL_UAV (DirectLighting.h:75-91, SkyDI.h:65-84)
constexpr uint32_t R0_SRV = 0, R0_UAV = kNumPlanes, R1_SRV = 2 * kNumPlanes, R1_UAV = 3 * kNumPlanes, TARGET_UAV = 4 * kNumPlanes, FINAL_UAV = 4 * kNumPlanes + 1;
struct DiState { uint32_t w, h; std::vector<uint8_t> res[2][3], target, finalRGBA; int currIdx = 0; bool temporalValid = false; };
uint32_t Slot(uint32_t e) { return SLOT_PASS + e; }
}

extern "C" {
DiState* zrefp_di_create(uint32_t w, uint32_t h)
{
    DiState* S = new DiState(); S->w = w; S->h = h;
    for (int s = 0; s < 2; s++) for (int p = 0; p < kNumPlanes; p++) S->res[s][p].assign((size_t)w * h * FormatBytes(kFmt[p]), 0);
    S->target.assign((size_t)w * h * 8, 0); S->finalRGBA.assign((size_t)w * h * 16, 0);      // TARGET is R16G16B16A16_FLOAT
    return S;
}
void zrefp_di_destroy(DiState* S) { delete S; }
void zrefp_di_reset_temporal(DiState* S) { S->temporalValid = false; S->currIdx = 0; }
// plane 0 .. kNumPlanes - 1 of the set written by the last frame; plane 8 = target (RGBA16F)
int zrefp_di_read_plane(const DiState* S, int plane, void* out)
{
    if (plane == 8) { memcpy(out, S->target.data(), S->target.size()); return 0; }
    if (plane < 0 || plane >= kNumPlanes) return -1;
    const auto& v = S->res[1 - S->currIdx][plane]; memcpy(out, v.data(), v.size()); return 0;
}

int zrefp_di_render(RefScene* r, DiState* S, const zr_frame_constants* cb, const zr_gbuffer_planes* curr, const zr_gbuffer_planes* prev, const zr_params* prm, float* finalOut)
{
    const uint32_t w = S->w, h = S->h;
    BindScene(r);
    DescriptorHeap& H = r->heap;
    BindGBuffer(H, SLOT_GBUF_CURR, curr); BindGBuffer(H, SLOT_GBUF_PREV, prev ? prev : curr);
    for (int s = 0; s < 2; s++) for (int p = 0; p < kNumPlanes; p++)
    {
        BindPlane(H, Slot((s ? R1_SRV : R0_SRV) + p), S->res[s][p].data(), w, h, kFmt[p]);
        BindPlane(H, Slot((s ? R1_UAV : R0_UAV) + p), S->res[s][p].data(), w, h, kFmt[p]);
    }
    BindPlane(H, Slot(TARGET_UAV), S->target.data(), w, h, FMT_RGBA16_FLOAT);
    BindPlane(H, Slot(FINAL_UAV), S->finalRGBA.data(), w, h, FMT_RGBA32_FLOAT);
    zr_frame_constants g = *cb;
    g.curr_gbuffer_desc_heap_offset = SLOT_GBUF_CURR; g.prev_gbuffer_desc_heap_offset = SLOT_GBUF_PREV; g.env_map_desc_heap_offset = SLOT_SKY_LUT;
    g.base_color_maps_desc_heap_offset += SLOT_TEXTURES; g.normal_maps_desc_heap_offset += SLOT_TEXTURES;
    g.metallic_roughness_maps_desc_heap_offset += SLOT_TEXTURES; g.emissive_maps_desc_heap_offset += SLOT_TEXTURES;
    using namespace hlsl;
    LocalCB L; memset(&L, 0, sizeof(L));
    const uint32_t dx = (w + 7) / 8, dy = (h + 7) / 8;
    const bool doTemporal = S->temporalValid && (prm->flags & ZR_IND_TEMPORAL_RESAMPLE);
    const bool doSpatial = doTemporal && (prm->flags & ZR_IND_SPATIAL_RESAMPLE);
    L.TargetDescHeapIdx = Slot(TARGET_UAV); L.FinalDescHeapIdx = Slot(FINAL_UAV);
    L.Alpha_min = prm->alpha_min;
    L.DispatchDimX = (uint16_t)dx; L.DispatchDimY = (uint16_t)dy;
#if ZR_DI_SKY
    L.M_max = prm->m_max_temporal | (prm->m_max_spatial << 16);         // SkyDI.cpp:81: M_max (sky) | M_max (sun) << 16
    L.NumGroupsInTile = (uint16_t)(SKY_DI_TILE_WIDTH * dy);
    L.Flags = (doTemporal ? CB_SKY_DI_FLAGS::TEMPORAL_RESAMPLE : 0u) | (doSpatial ? CB_SKY_DI_FLAGS::SPATIAL_RESAMPLE : 0u) | (!S->temporalValid ? CB_SKY_DI_FLAGS::RESET_TEMPORAL_TEXTURES : 0u);
#else
    L.M_max = prm->m_max_temporal;
    L.NumGroupsInTile = (uint16_t)(RESTIR_DI_TILE_WIDTH * dy);
    L.NumSampleSets = prm->presampling ? prm->num_sample_sets : 0u; L.SampleSetSize = prm->presampling ? prm->sample_set_size : 0u;
    L.Flags = (doTemporal ? CB_RDI_FLAGS::TEMPORAL_RESAMPLE : 0u) | (doSpatial ? CB_RDI_FLAGS::SPATIAL_RESAMPLE : 0u) |
        ((prm->flags & ZR_DI_STOCHASTIC_SPATIAL) ? CB_RDI_FLAGS::STOCHASTIC_SPATIAL : 0u) |
        ((prm->flags & ZR_DI_EXTRA_DISOCCLUSION_SAMPLING) ? CB_RDI_FLAGS::EXTRA_DISOCCLUSION_SAMPLING : 0u) |
        (!S->temporalValid ? CB_RDI_FLAGS::RESET_TEMPORAL_TEXTURES : 0u);
#endif
    const int c = S->currIdx;
    L.PrevReservoir_A_DescHeapIdx = Slot(c == 1 ? R0_SRV : R1_SRV);
    L.CurrReservoir_A_DescHeapIdx = Slot(c == 1 ? R1_UAV : R0_UAV);
    ZrDispatch d; memset(&d, 0, sizeof(d));
    d.scene = r; d.prev_scene = r->prevHolder.get(); d.heap = &H; d.frame_cb = &g; d.local_cb = &L; d.local_cb_bytes = sizeof(L); d.groups_x = dx; d.groups_y = dy;
    zrefp_shader_di_temporal(&d);
    if (doSpatial)
    {
        L.CurrReservoir_A_DescHeapIdx = Slot(c == 1 ? R1_SRV : R0_SRV);
        zrefp_shader_di_spatial(&d);
    }
    S->temporalValid = true; S->currIdx = 1 - S->currIdx;
    if (finalOut) memcpy(finalOut, S->finalRGBA.data(), S->finalRGBA.size());
    return 0;
}
}
