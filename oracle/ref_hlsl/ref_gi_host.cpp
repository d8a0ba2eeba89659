// ORACLE tooling -- test infrastructure only.
// ReSTIR GI (K10) from the reference's own shader (ReSTIR_GI.hlsl, compiled by ref_pass_shader.cpp) driven by a restatement of
// IndirectLighting::RenderReSTIR_GI + the Render() tail (IndirectLighting.cpp:277-368, 1006-1025).  Built per permutation: libzref_gi_{e0,e1,e1p}.so
#include "ref_pass_common.h"
#include "ref_dispatch.h"
namespace hlsl {
#include "ZetaRenderPass/Common/FrameConstants.h"
#include "ZetaRenderPass/IndirectLighting/IndirectLighting_Common.h"
}
using namespace refpass;
ZREFP_SCENE_API
extern "C" void zrefp_shader_gi(const ZrDispatch*);

namespace {
enum DESC_TABLE_RGI : uint32_t { R0_A_SRV, R0_B_SRV, R0_C_SRV, R0_A_UAV, R0_B_UAV, R0_C_UAV, R1_A_SRV, R1_B_SRV, R1_C_SRV, R1_A_UAV, R1_B_UAV, R1_C_UAV, FINAL_UAV };
const int kFmt[3] = {FMT_RGBA32_FLOAT, FMT_RGBA16_FLOAT, FMT_RGBA32_FLOAT};      // IndirectLighting.h:120-126
struct GiState { uint32_t w, h; std::vector<uint8_t> res[2][3], finalRGBA; int currIdx = 0; bool temporalValid = false; };
uint32_t Slot(uint32_t e) { return SLOT_PASS + e; }
}

extern "C" {
GiState* zrefp_gi_create(uint32_t w, uint32_t h)
{
    GiState* S = new GiState(); S->w = w; S->h = h;
    for (int s = 0; s < 2; s++) for (int p = 0; p < 3; p++) S->res[s][p].assign((size_t)w * h * FormatBytes(kFmt[p]), 0);
    S->finalRGBA.assign((size_t)w * h * 16, 0);
    return S;
}
void zrefp_gi_destroy(GiState* S) { delete S; }
void zrefp_gi_reset_temporal(GiState* S) { S->temporalValid = false; S->currIdx = 0; }
// plane 0..2 = A, B, C of the set the NEXT frame reads as "previous"
int zrefp_gi_read_plane(const GiState* S, int plane, void* out)
{ if (plane < 0 || plane > 2) return -1; const auto& v = S->res[1 - S->currIdx][plane]; memcpy(out, v.data(), v.size()); return 0; }

int zrefp_gi_render(RefScene* r, GiState* S, const zr_frame_constants* cb, const zr_gbuffer_planes* curr, const zr_gbuffer_planes* prev, const zr_params* prm, float* finalOut)
{
    const uint32_t w = S->w, h = S->h;
    BindScene(r);
    DescriptorHeap& H = r->heap;
    BindGBuffer(H, SLOT_GBUF_CURR, curr); BindGBuffer(H, SLOT_GBUF_PREV, prev ? prev : curr);
    for (int s = 0; s < 2; s++) for (int p = 0; p < 3; p++)
    {
        BindPlane(H, Slot((s ? R1_A_SRV : R0_A_SRV) + p), S->res[s][p].data(), w, h, kFmt[p]);
        BindPlane(H, Slot((s ? R1_A_UAV : R0_A_UAV) + p), S->res[s][p].data(), w, h, kFmt[p]);
    }
    BindPlane(H, Slot(FINAL_UAV), S->finalRGBA.data(), w, h, FMT_RGBA32_FLOAT);
    zr_frame_constants g = *cb;
    g.curr_gbuffer_desc_heap_offset = SLOT_GBUF_CURR; g.prev_gbuffer_desc_heap_offset = SLOT_GBUF_PREV; g.env_map_desc_heap_offset = SLOT_SKY_LUT;
    g.base_color_maps_desc_heap_offset += SLOT_TEXTURES; g.normal_maps_desc_heap_offset += SLOT_TEXTURES;
    g.metallic_roughness_maps_desc_heap_offset += SLOT_TEXTURES; g.emissive_maps_desc_heap_offset += SLOT_TEXTURES;
    using namespace hlsl;
    cb_ReSTIR_GI L; memset(&L, 0, sizeof(L));
    const int c = S->currIdx;
    L.PrevReservoir_A_DescHeapIdx = Slot(c == 1 ? R0_A_SRV : R1_A_SRV); L.PrevReservoir_B_DescHeapIdx = L.PrevReservoir_A_DescHeapIdx + 1; L.PrevReservoir_C_DescHeapIdx = L.PrevReservoir_A_DescHeapIdx + 2;
    L.CurrReservoir_A_DescHeapIdx = Slot(c == 1 ? R1_A_UAV : R0_A_UAV); L.CurrReservoir_B_DescHeapIdx = L.CurrReservoir_A_DescHeapIdx + 1; L.CurrReservoir_C_DescHeapIdx = L.CurrReservoir_A_DescHeapIdx + 2;
    L.FinalDescHeapIdx = Slot(FINAL_UAV);
    L.Flags = prm->flags & (CB_IND_FLAGS::STOCHASTIC_MULTI_BOUNCE | CB_IND_FLAGS::RUSSIAN_ROULETTE | CB_IND_FLAGS::BOILING_SUPPRESSION | CB_IND_FLAGS::PATH_REGULARIZATION);
    if ((prm->flags & CB_IND_FLAGS::TEMPORAL_RESAMPLE) && S->temporalValid) L.Flags |= CB_IND_FLAGS::TEMPORAL_RESAMPLE;
    if (!S->temporalValid) L.Flags |= CB_IND_FLAGS::RESET_TEMPORAL_TEXTURES;
    const uint32_t dx = (w + 7) / 8, dy = (h + 7) / 8;
    L.DispatchDimX_NumGroupsInTile = ((RESTIR_GI_TEMPORAL_TILE_WIDTH * dy) << 16) | dx;
    L.SampleSetSize_NumSampleSets = prm->presampling ? ((prm->num_sample_sets << 16) | prm->sample_set_size) : 0u;
    L.M_max = prm->m_max_temporal; L.MaxNonTrBounces = prm->max_non_tr_bounces; L.MaxGlossyTrBounces = prm->max_glossy_tr_bounces;
    L.TexFilterDescHeapIdx = EnumToSamplerIdx(prm->tex_filter);
    if (prm->use_lvg)
    {   // IndirectLighting::SetLightVoxelGridParams, IndirectLighting.h:88-102: dimensions packed 16 + 16 | 32, extents + y offset as four halfs
        const uint32_t dx_ = prm->lvg_grid_dim & 1023u, dy_ = (prm->lvg_grid_dim >> 10) & 1023u, dz_ = (prm->lvg_grid_dim >> 20) & 1023u;
        L.GridDim_xy = (dy_ << 16) | dx_; L.GridDim_z = dz_;
        const uint32_t ex = zr_f32_to_f16(prm->lvg_extents[0]), ey = zr_f32_to_f16(prm->lvg_extents[1]), ez = zr_f32_to_f16(prm->lvg_extents[2]), oy = zr_f32_to_f16(prm->lvg_offset_y);
        L.Extents_xy = (ey << 16) | ex; L.Extents_z_Offset_y = (oy << 16) | ez;
    }
    ZrDispatch d; memset(&d, 0, sizeof(d));
    d.scene = r; d.heap = &H; d.frame_cb = &g; d.local_cb = &L; d.local_cb_bytes = sizeof(L); d.groups_x = dx; d.groups_y = dy;
    zrefp_shader_gi(&d);
    S->temporalValid = true; S->currIdx = 1 - S->currIdx;
    if (finalOut) memcpy(finalOut, S->finalRGBA.data(), S->finalRGBA.size());
    return 0;
}
}
