// ORACLE tooling -- test infrastructure only.
//
// ReSTIR PT (K11, K13-K16) from the reference's own shaders compiled as C++ (ref_pass_shader.cpp, one object per shader permutation),
// driven by a restatement of the reference's HOST code: IndirectLighting::RenderReSTIR_PT / ReSTIR_PT_Temporal / ReSTIR_PT_Spatial
// (Source/ZetaRenderPass/IndirectLighting/IndirectLighting.cpp:370-1025) -- constant buffers, descriptor indices (the layout of
// DESC_TABLE_RPT, IndirectLighting.h:167-230), dispatch sizes and order, the ping-pong of the two reservoir sets.  A "wave" is the 64
// consecutive threads of a thread group, which is what the ABI pins wave intrinsics to.  The thread-sort passes (K12,
// ReSTIR_PT_Sort.hlsl x 4) run as in the reference: the temporal pair always, the spatial pair under CB_IND_FLAGS::SORT_SPATIAL; the replay and
// reconnect shaders consume the maps when zr_params.flags carries SORT_TEMPORAL / SORT_SPATIAL.  The group runner steps the waves of a group
// in order, so the LDS InterlockedAdd of the sort (whose arrival order is unspecified on a GPU) resolves in wave order -- the order the ABI fixes.  Built per NEE permutation: libzref_rpt_{e0,e1,e1p}.so.
#include "ref_pass_common.h"
#include "ref_dispatch.h"

namespace hlsl {
#include "ZetaRenderPass/Common/FrameConstants.h"
#include "ZetaRenderPass/IndirectLighting/IndirectLighting_Common.h"
}
using namespace refpass;
ZREFP_SCENE_API

extern "C" {
void zrefp_shader_rpt_pathtrace(const ZrDispatch*);
void zrefp_shader_rpt_replay_ctt(const ZrDispatch*); void zrefp_shader_rpt_replay_ttc(const ZrDispatch*);
void zrefp_shader_rpt_replay_cts(const ZrDispatch*); void zrefp_shader_rpt_replay_stc(const ZrDispatch*);
void zrefp_shader_rpt_reconnect_ctt(const ZrDispatch*); void zrefp_shader_rpt_reconnect_ttc(const ZrDispatch*);
void zrefp_shader_rpt_reconnect_cts(const ZrDispatch*); void zrefp_shader_rpt_reconnect_stc(const ZrDispatch*);
void zrefp_shader_rpt_spatial_search(const ZrDispatch*);
void zrefp_shader_rpt_sort_ctt(const ZrDispatch*); void zrefp_shader_rpt_sort_ttc(const ZrDispatch*);
void zrefp_shader_rpt_sort_cts(const ZrDispatch*); void zrefp_shader_rpt_sort_stc(const ZrDispatch*);
}

namespace {
// IndirectLighting.h:167-230, same order; slot = SLOT_PASS + enum value
enum DESC_TABLE_RPT : uint32_t
{
    RESERVOIR_0_A_SRV, RESERVOIR_0_B_SRV, RESERVOIR_0_C_SRV, RESERVOIR_0_D_SRV, RESERVOIR_0_E_SRV, RESERVOIR_0_F_SRV, RESERVOIR_0_G_SRV,
    RESERVOIR_0_A_UAV, RESERVOIR_0_B_UAV, RESERVOIR_0_C_UAV, RESERVOIR_0_D_UAV, RESERVOIR_0_E_UAV, RESERVOIR_0_F_UAV, RESERVOIR_0_G_UAV,
    RESERVOIR_1_A_SRV, RESERVOIR_1_B_SRV, RESERVOIR_1_C_SRV, RESERVOIR_1_D_SRV, RESERVOIR_1_E_SRV, RESERVOIR_1_F_SRV, RESERVOIR_1_G_SRV,
    RESERVOIR_1_A_UAV, RESERVOIR_1_B_UAV, RESERVOIR_1_C_UAV, RESERVOIR_1_D_UAV, RESERVOIR_1_E_UAV, RESERVOIR_1_F_UAV, RESERVOIR_1_G_UAV,
    RBUFFER_A_CtN_SRV, RBUFFER_B_CtN_SRV, RBUFFER_C_CtN_SRV, RBUFFER_D_CtN_SRV, RBUFFER_A_CtN_UAV, RBUFFER_B_CtN_UAV, RBUFFER_C_CtN_UAV, RBUFFER_D_CtN_UAV,
    RBUFFER_A_NtC_SRV, RBUFFER_B_NtC_SRV, RBUFFER_C_NtC_SRV, RBUFFER_D_NtC_SRV, RBUFFER_A_NtC_UAV, RBUFFER_B_NtC_UAV, RBUFFER_C_NtC_UAV, RBUFFER_D_NtC_UAV,
    THREAD_MAP_CtN_SRV, THREAD_MAP_CtN_UAV, THREAD_MAP_NtC_SRV, THREAD_MAP_NtC_UAV,
    SPATIAL_NEIGHBOR_SRV, SPATIAL_NEIGHBOR_UAV, TARGET_UAV, FINAL_UAV
};
// formats: IndirectLighting.h:303-323 (ResourceFormats_RPT)
const int kResFmt[7] = {FMT_RGBA8_UINT, FMT_RG32_FLOAT, FMT_RGBA32_UINT, FMT_RGBA32_UINT, FMT_R16_FLOAT, FMT_RG32_FLOAT, FMT_RG32_UINT};
const int kRbFmt[4] = {FMT_RGBA16_FLOAT, FMT_RGBA32_UINT, FMT_RGBA32_UINT, FMT_R16_UINT};

struct RptState
{
    uint32_t w, h;
    std::vector<uint8_t> res[2][7], rb[2][4], threadMap[2], neighbor, target, finalRGBA;
    int currTemporalIdx = 0; bool temporalValid = false;
    hlsl::cb_ReSTIR_PT_PathTrace cbPT; hlsl::cb_ReSTIR_PT_Reuse cbReuse;
};
uint32_t Slot(uint32_t e) { return SLOT_PASS + e; }
uint32_t CeilDiv(uint32_t a, uint32_t b) { return (a + b - 1) / b; }
}

extern "C" {

RptState* zrefp_rpt_create(uint32_t w, uint32_t h)
{
    RptState* S = new RptState(); S->w = w; S->h = h;
    const size_t n = (size_t)w * h;
    for (int s = 0; s < 2; s++)
    {
        for (int p = 0; p < 7; p++) S->res[s][p].assign(n * FormatBytes(kResFmt[p]), 0);
        for (int p = 0; p < 4; p++) S->rb[s][p].assign(n * FormatBytes(kRbFmt[p]), 0);
        S->threadMap[s].assign(n * 2, 0);
    }
    S->neighbor.assign(n * 2, 0); S->target.assign(n * 16, 0); S->finalRGBA.assign(n * 16, 0);
    return S;
}
void zrefp_rpt_destroy(RptState* S) { delete S; }
void zrefp_rpt_reset_temporal(RptState* S) { S->temporalValid = false; S->currTemporalIdx = 0; }      // IndirectLighting::ResetTemporal, IndirectLighting.cpp:212-218

// which: 0 = the reservoir set the NEXT frame reads as "previous" (what ZR_OUT_RPT_RESERVOIR_* exposes), 1 = the other set
int zrefp_rpt_read_plane(const RptState* S, int which, int plane, void* out)
{
    // after Render(): with spatial reuse the final reservoirs were written to set 1 - c and the index flipped twice (back to c); without it the
    // final reservoirs are in set c and the index flipped once.  Either way the next frame's "previous" set is 1 - currTemporalIdx.
    const int next_prev = 1 - S->currTemporalIdx;
    const int set = which == 0 ? next_prev : 1 - next_prev;
    if (plane >= 0 && plane < 7) { memcpy(out, S->res[set][plane].data(), S->res[set][plane].size()); return 0; }
    if (plane == 7) { memcpy(out, S->neighbor.data(), S->neighbor.size()); return 0; }
    if (plane == 8) { memcpy(out, S->target.data(), S->target.size()); return 0; }
    if (plane == 9) { memcpy(out, S->finalRGBA.data(), S->finalRGBA.size()); return 0; }
    if (plane == 10 || plane == 11) { memcpy(out, S->threadMap[plane - 10].data(), S->threadMap[plane - 10].size()); return 0; }      // CtN, NtC
    return -1;
}

// IndirectLighting::Render with m_method == ReSTIR_PT (IndirectLighting.cpp:877-1025)
int zrefp_rpt_render(RefScene* r, RptState* S, const zr_frame_constants* cb, const zr_gbuffer_planes* curr, const zr_gbuffer_planes* prev, const zr_params* prm, float* finalOut)
{
    const uint32_t w = S->w, h = S->h;
    BindScene(r);
    DescriptorHeap& H = r->heap;
    BindGBuffer(H, SLOT_GBUF_CURR, curr);
    BindGBuffer(H, SLOT_GBUF_PREV, prev ? prev : curr);
    for (int s = 0; s < 2; s++)
        for (int p = 0; p < 7; p++)
        {
            BindPlane(H, Slot((s ? RESERVOIR_1_A_SRV : RESERVOIR_0_A_SRV) + p), S->res[s][p].data(), w, h, kResFmt[p]);
            BindPlane(H, Slot((s ? RESERVOIR_1_A_UAV : RESERVOIR_0_A_UAV) + p), S->res[s][p].data(), w, h, kResFmt[p]);
        }
    for (int p = 0; p < 4; p++)
    {
        BindPlane(H, Slot(RBUFFER_A_CtN_SRV + p), S->rb[0][p].data(), w, h, kRbFmt[p]); BindPlane(H, Slot(RBUFFER_A_CtN_UAV + p), S->rb[0][p].data(), w, h, kRbFmt[p]);
        BindPlane(H, Slot(RBUFFER_A_NtC_SRV + p), S->rb[1][p].data(), w, h, kRbFmt[p]); BindPlane(H, Slot(RBUFFER_A_NtC_UAV + p), S->rb[1][p].data(), w, h, kRbFmt[p]);
    }
    BindPlane(H, Slot(THREAD_MAP_CtN_SRV), S->threadMap[0].data(), w, h, FMT_R16_UINT); BindPlane(H, Slot(THREAD_MAP_CtN_UAV), S->threadMap[0].data(), w, h, FMT_R16_UINT);
    BindPlane(H, Slot(THREAD_MAP_NtC_SRV), S->threadMap[1].data(), w, h, FMT_R16_UINT); BindPlane(H, Slot(THREAD_MAP_NtC_UAV), S->threadMap[1].data(), w, h, FMT_R16_UINT);
    BindPlane(H, Slot(SPATIAL_NEIGHBOR_SRV), S->neighbor.data(), w, h, FMT_RG8_UINT); BindPlane(H, Slot(SPATIAL_NEIGHBOR_UAV), S->neighbor.data(), w, h, FMT_RG8_UINT);
    BindPlane(H, Slot(TARGET_UAV), S->target.data(), w, h, FMT_RGBA32_FLOAT);
    BindPlane(H, Slot(FINAL_UAV), S->finalRGBA.data(), w, h, FMT_RGBA32_FLOAT);

    zr_frame_constants g = *cb;
    g.curr_gbuffer_desc_heap_offset = SLOT_GBUF_CURR; g.prev_gbuffer_desc_heap_offset = SLOT_GBUF_PREV; g.env_map_desc_heap_offset = SLOT_SKY_LUT;
    g.base_color_maps_desc_heap_offset += SLOT_TEXTURES; g.normal_maps_desc_heap_offset += SLOT_TEXTURES;
    g.metallic_roughness_maps_desc_heap_offset += SLOT_TEXTURES; g.emissive_maps_desc_heap_offset += SLOT_TEXTURES;

    using namespace hlsl;
    // ---- constructor / parameter state: IndirectLighting.cpp:146-165
    cb_ReSTIR_PT_PathTrace& PT = S->cbPT; cb_ReSTIR_PT_Reuse& RU = S->cbReuse;
    memset(&PT, 0, sizeof(PT)); memset(&RU, 0, sizeof(RU));
    const uint32_t texFilter = EnumToSamplerIdx(prm->tex_filter);
    PT.Alpha_min = RU.Alpha_min = prm->alpha_min;
    PT.TexFilterDescHeapIdx = texFilter;
    PT.Packed = RU.Packed = prm->max_non_tr_bounces | (prm->max_glossy_tr_bounces << PACKED_INDEX::NUM_GLOSSY_BOUNCES) |
        ((prm->m_max_temporal & 0xf) << PACKED_INDEX::MAX_TEMPORAL_M) | ((prm->m_max_spatial & 0xf) << PACKED_INDEX::MAX_SPATIAL_M) | (texFilter << PACKED_INDEX::TEX_FILTER);
    const uint32_t userFlags = prm->flags & (CB_IND_FLAGS::STOCHASTIC_MULTI_BOUNCE | CB_IND_FLAGS::RUSSIAN_ROULETTE | CB_IND_FLAGS::BOILING_SUPPRESSION | CB_IND_FLAGS::PATH_REGULARIZATION);
    PT.Flags = RU.Flags = userFlags;
    PT.Flags |= prm->flags & CB_IND_FLAGS::SORT_TEMPORAL;                                            // IndirectLighting.cpp:162-164, 1552-1561
    RU.Flags |= prm->flags & (CB_IND_FLAGS::SORT_TEMPORAL | CB_IND_FLAGS::SORT_SPATIAL);
    PT.SampleSetSize_NumSampleSets = prm->presampling ? ((prm->num_sample_sets << 16) | prm->sample_set_size) : 0u;
    PT.TargetDescHeapIdx = RU.TargetDescHeapIdx = Slot(TARGET_UAV);
    PT.Final = RU.FinalDescHeapIdx = Slot(FINAL_UAV);
    RU.ThreadMap_CtN_DescHeapIdx = Slot(THREAD_MAP_CtN_SRV); RU.ThreadMap_NtC_DescHeapIdx = Slot(THREAD_MAP_NtC_SRV);
    RU.SpatialNeighborHeapIdx = Slot(SPATIAL_NEIGHBOR_SRV);
    if (!S->temporalValid) PT.Flags |= CB_IND_FLAGS::RESET_TEMPORAL_TEXTURES;

    ZrDispatch d; memset(&d, 0, sizeof(d));
    d.scene = r; d.prev_scene = r->prevHolder.get(); d.heap = &H; d.frame_cb = &g;
    auto Run = [&](void (*shader)(const ZrDispatch*), const void* lcb, uint32_t bytes, uint32_t gx, uint32_t gy, bool prevAS) {
        d.local_cb = lcb; d.local_cb_bytes = bytes; d.groups_x = gx; d.groups_y = gy; d.use_prev_scene = prevAS ? 1 : 0; shader(&d); };

    // ---- RenderReSTIR_PT, IndirectLighting.cpp:877-1004
    const int c = S->currTemporalIdx;
    const uint32_t srvAIdx = c == 1 ? RESERVOIR_0_A_SRV : RESERVOIR_1_A_SRV;
    const uint32_t uavAIdx = c == 1 ? RESERVOIR_1_A_UAV : RESERVOIR_0_A_UAV;
    const bool doTemporal = (prm->flags & CB_IND_FLAGS::TEMPORAL_RESAMPLE) && S->temporalValid;
    const uint32_t numSpatialPasses = prm->num_spatial_passes > 2u ? 2u : prm->num_spatial_passes;      // m_numSpatialPasses, 0..2 (IndirectLighting.cpp:1240)
    const bool doSpatial = (prm->flags & CB_IND_FLAGS::SPATIAL_RESAMPLE) && (numSpatialPasses > 0) && doTemporal;       // IndirectLighting.cpp:905-906
    {
        const uint32_t dx = CeilDiv(w, RESTIR_PT_PATH_TRACE_GROUP_DIM_X), dy = CeilDiv(h, RESTIR_PT_PATH_TRACE_GROUP_DIM_Y);
        if (doTemporal) PT.Flags |= CB_IND_FLAGS::TEMPORAL_RESAMPLE;
        if (doSpatial) { PT.Flags |= CB_IND_FLAGS::SPATIAL_RESAMPLE; RU.Flags |= CB_IND_FLAGS::SPATIAL_RESAMPLE; }
        PT.DispatchDimX_NumGroupsInTile = ((RESTIR_PT_TILE_WIDTH * dy) << 16) | dx;
        PT.Reservoir_A_DescHeapIdx = Slot(uavAIdx);
        Run(zrefp_shader_rpt_pathtrace, &PT, sizeof(PT), dx, dy, false);
    }
    if (doTemporal)
    {
        RU.PrevReservoir_A_DescHeapIdx = Slot(srvAIdx);
        RU.Reservoir_A_DescHeapIdx = PT.Reservoir_A_DescHeapIdx;
        // ---- ReSTIR_PT_Temporal, IndirectLighting.cpp:370-596
        RU.Flags |= CB_IND_FLAGS::TEMPORAL_RESAMPLE;
        {   // Sort - TtC, Sort - CtT (IndirectLighting.cpp:383-441; unconditional)
            cb_ReSTIR_PT_Sort so; memset(&so, 0, sizeof(so));
            so.DispatchDimX = CeilDiv(w, RESTIR_PT_SORT_GROUP_DIM_X * 2); so.DispatchDimY = CeilDiv(h, RESTIR_PT_SORT_GROUP_DIM_Y * 2);
            so.Flags = RU.Flags;
            so.Reservoir_A_DescHeapIdx = RU.PrevReservoir_A_DescHeapIdx; so.MapDescHeapIdx = Slot(THREAD_MAP_NtC_UAV);
            Run(zrefp_shader_rpt_sort_ttc, &so, sizeof(so), so.DispatchDimX, so.DispatchDimY, false);
            so.Reservoir_A_DescHeapIdx = PT.Reservoir_A_DescHeapIdx; so.MapDescHeapIdx = Slot(THREAD_MAP_CtN_UAV);
            Run(zrefp_shader_rpt_sort_ctt, &so, sizeof(so), so.DispatchDimX, so.DispatchDimY, false);
        }
        const uint32_t rx = CeilDiv(w, RESTIR_PT_REPLAY_GROUP_DIM_X), ry = CeilDiv(h, RESTIR_PT_REPLAY_GROUP_DIM_Y);
        RU.RBufferA_CtN_DescHeapIdx = Slot(RBUFFER_A_CtN_UAV); RU.RBufferA_NtC_DescHeapIdx = Slot(RBUFFER_A_NtC_UAV);
        Run(zrefp_shader_rpt_replay_ctt, &RU, sizeof(RU), rx, ry, true);
        Run(zrefp_shader_rpt_replay_ttc, &RU, sizeof(RU), rx, ry, false);
        RU.RBufferA_CtN_DescHeapIdx = Slot(RBUFFER_A_CtN_SRV); RU.RBufferA_NtC_DescHeapIdx = Slot(RBUFFER_A_NtC_SRV);
        const uint32_t tx = CeilDiv(w, RESTIR_PT_TEMPORAL_GROUP_DIM_X), ty = CeilDiv(h, RESTIR_PT_TEMPORAL_GROUP_DIM_Y);
        RU.DispatchDimX_NumGroupsInTile = ((RESTIR_PT_TILE_WIDTH * ty) << 16) | tx;
        Run(zrefp_shader_rpt_reconnect_ctt, &RU, sizeof(RU), tx, ty, true);
        Run(zrefp_shader_rpt_reconnect_ttc, &RU, sizeof(RU), tx, ty, false);
    }
    if (doSpatial)
    for (uint32_t pass = 0; pass < numSpatialPasses; pass++)
    {
        // ---- ReSTIR_PT_Spatial, IndirectLighting.cpp:598-875: for (pass < m_numSpatialPasses)
        RU.Packed = (RU.Packed & ~0xf000u) | ((numSpatialPasses << 14) | (pass << 12));
        {
            const uint32_t dx = CeilDiv(w, RESTIR_PT_SPATIAL_SEARCH_GROUP_DIM_X), dy = CeilDiv(h, RESTIR_PT_SPATIAL_SEARCH_GROUP_DIM_Y);
            cb_ReSTIR_PT_SpatialSearch ss; memset(&ss, 0, sizeof(ss));
            ss.DispatchDimX_NumGroupsInTile = ((RESTIR_PT_TILE_WIDTH * dy) << 16) | dx;
            ss.OutputDescHeapIdx = Slot(SPATIAL_NEIGHBOR_UAV);
            ss.Flags = RU.Flags;
            Run(zrefp_shader_rpt_spatial_search, &ss, sizeof(ss), dx, dy, false);
        }
        S->currTemporalIdx = 1 - S->currTemporalIdx;
        if (RU.Flags & CB_IND_FLAGS::SORT_SPATIAL)      // Sort - CtS, Sort - StC (IndirectLighting.cpp:690-742)
        {
            cb_ReSTIR_PT_Sort so; memset(&so, 0, sizeof(so));
            so.DispatchDimX = CeilDiv(w, RESTIR_PT_SORT_GROUP_DIM_X * 2); so.DispatchDimY = CeilDiv(h, RESTIR_PT_SORT_GROUP_DIM_Y * 2);
            so.Flags = RU.Flags; so.Reservoir_A_DescHeapIdx = RU.Reservoir_A_DescHeapIdx;
            so.MapDescHeapIdx = Slot(THREAD_MAP_CtN_UAV);
            Run(zrefp_shader_rpt_sort_cts, &so, sizeof(so), so.DispatchDimX, so.DispatchDimY, false);
            so.SpatialNeighborHeapIdx = Slot(SPATIAL_NEIGHBOR_SRV); so.MapDescHeapIdx = Slot(THREAD_MAP_NtC_UAV);
            Run(zrefp_shader_rpt_sort_stc, &so, sizeof(so), so.DispatchDimX, so.DispatchDimY, false);
        }
        const uint32_t rx = CeilDiv(w, RESTIR_PT_REPLAY_GROUP_DIM_X), ry = CeilDiv(h, RESTIR_PT_REPLAY_GROUP_DIM_Y);
        RU.RBufferA_CtN_DescHeapIdx = Slot(RBUFFER_A_CtN_UAV); RU.RBufferA_NtC_DescHeapIdx = Slot(RBUFFER_A_NtC_UAV);
        Run(zrefp_shader_rpt_replay_cts, &RU, sizeof(RU), rx, ry, false);
        Run(zrefp_shader_rpt_replay_stc, &RU, sizeof(RU), rx, ry, false);
        RU.RBufferA_CtN_DescHeapIdx = Slot(RBUFFER_A_CtN_SRV); RU.RBufferA_NtC_DescHeapIdx = Slot(RBUFFER_A_NtC_SRV);
        const uint32_t sx = CeilDiv(w, RESTIR_PT_SPATIAL_GROUP_DIM_X), sy = CeilDiv(h, RESTIR_PT_SPATIAL_GROUP_DIM_Y);
        RU.DispatchDimX_NumGroupsInTile = ((RESTIR_PT_TILE_WIDTH * sy) << 16) | sx;
        Run(zrefp_shader_rpt_reconnect_cts, &RU, sizeof(RU), sx, sy, false);
        Run(zrefp_shader_rpt_reconnect_stc, &RU, sizeof(RU), sx, sy, false);
        // Prepare for next iteration, IndirectLighting.cpp:860-870: swap input and output reservoirs
        if (pass == 0 && numSpatialPasses == 2u) std::swap(RU.PrevReservoir_A_DescHeapIdx, RU.Reservoir_A_DescHeapIdx);
    }
    // ---- Render() tail, IndirectLighting.cpp:1021-1024
    S->temporalValid = true;
    S->currTemporalIdx = 1 - S->currTemporalIdx;
    if (finalOut) memcpy(finalOut, S->finalRGBA.data(), S->finalRGBA.size());
    return 0;
}

} // extern "C"
