// ORACLE tooling -- test infrastructure only.
// K9 from the reference's own shader: Source/ZetaRenderPass/IndirectLighting/PathTracer/PathTracer.hlsl (+ PathTracing.hlsli, NEE.hlsli,
// ReSTIR_GI_NEE.hlsli, RayQuery.hlsli, LightSource.hlsli, BSDFSampling.hlsli ...) compiled as C++; NEE_EMISSIVE picks the shader permutation
// exactly like the reference's build does (IndirectLighting.h:251-300).
#include "ref_pass_common.h"

namespace hlsl {
#include "ZetaRenderPass/Common/FrameConstants.h"
#include "ZetaRenderPass/IndirectLighting/IndirectLighting_Common.h"
#include "ZetaRenderPass/IndirectLighting/PathTracer/PathTracer.hlsl"
}

using namespace refpass;
ZREFP_SCENE_API

extern "C" int zrefp_nee_emissive() { return NEE_EMISSIVE; }

// one frame of K9 over the G-buffer `planes`; final = RGBA32F plane (read-modify-write when Accumulate && CameraStatic)
extern "C" int zrefp_pathtrace_render(RefScene* r, const zr_frame_constants* cb, const zr_gbuffer_planes* planes, const zr_params* prm, float* finalRGBA)
{
    static_assert(sizeof(hlsl::cbFrameConstants) == sizeof(zr_frame_constants), "cbFrameConstants layout");
    static_assert(sizeof(hlsl::RT::MeshInstance) == sizeof(zr_mesh_instance), "MeshInstance layout");
    BindScene(r);
    const uint32_t w = planes->width, h = planes->height;
    BindGBuffer(r->heap, SLOT_GBUF_CURR, planes);
    BindPlane(r->heap, SLOT_PASS + 0, finalRGBA, w, h, FMT_RGBA32_FLOAT);
    memcpy(&hlsl::g_frame, cb, sizeof(zr_frame_constants));
    hlsl::g_frame.CurrGBufferDescHeapOffset = SLOT_GBUF_CURR; hlsl::g_frame.PrevGBufferDescHeapOffset = SLOT_GBUF_PREV;
    hlsl::g_frame.EnvMapDescHeapOffset = SLOT_SKY_LUT;
    hlsl::g_frame.BaseColorMapsDescHeapOffset += SLOT_TEXTURES; hlsl::g_frame.NormalMapsDescHeapOffset += SLOT_TEXTURES;
    hlsl::g_frame.MetallicRoughnessMapsDescHeapOffset += SLOT_TEXTURES; hlsl::g_frame.EmissiveMapsDescHeapOffset += SLOT_TEXTURES;
    hlsl::cb_ReSTIR_GI& L = hlsl::g_local;
    memset(&L, 0, sizeof(L));
    L.FinalDescHeapIdx = SLOT_PASS + 0;
    L.Flags = prm->flags;
    const uint32_t dimX = (w + 7) / 8, dimY = (h + 7) / 8;
    L.DispatchDimX_NumGroupsInTile = ((RESTIR_GI_TEMPORAL_TILE_WIDTH * dimY) << 16) | dimX;         // IndirectLighting.cpp:247-249
    L.SampleSetSize_NumSampleSets = prm->presampling ? ((prm->num_sample_sets << 16) | prm->sample_set_size) : 0u;   // IndirectLighting.cpp:153-165
    L.MaxNonTrBounces = prm->max_non_tr_bounces; L.MaxGlossyTrBounces = prm->max_glossy_tr_bounces; L.M_max = prm->m_max_temporal;
    L.TexFilterDescHeapIdx = EnumToSamplerIdx(prm->tex_filter);      // IndirectLighting.cpp:21-33, 1565
    hlsl::g_bvh.scene = &r->sc;
    hlsl::g_frameMeshData = StructuredBuffer<hlsl::RT::MeshInstance>((const hlsl::RT::MeshInstance*)r->sc.instances.data(), (uint32_t)r->sc.instances.size());
    hlsl::g_vertices = StructuredBuffer<hlsl::Vertex>((const hlsl::Vertex*)r->sc.vertices.data(), (uint32_t)r->sc.vertices.size());
    hlsl::g_indices = StructuredBuffer<hlsl::uint>(r->sc.indices.data(), (uint32_t)r->sc.indices.size());
    hlsl::g_materials = StructuredBuffer<hlsl::Material>((const hlsl::Material*)r->sc.materials.data(), (uint32_t)r->sc.materials.size());
#if NEE_EMISSIVE == 1
    hlsl::g_emissives = StructuredBuffer<hlsl::RT::EmissiveTriangle>((const hlsl::RT::EmissiveTriangle*)r->sc.emissives.data(), (uint32_t)r->sc.emissives.size());
    hlsl::g_aliasTable = StructuredBuffer<hlsl::RT::EmissiveLumenAliasTableEntry>((const hlsl::RT::EmissiveLumenAliasTableEntry*)r->sc.alias.data(), (uint32_t)r->sc.alias.size());
    hlsl::g_sampleSets = StructuredBuffer<hlsl::RT::PresampledEmissiveTriangle>((const hlsl::RT::PresampledEmissiveTriangle*)r->sc.sampleSets.data(), (uint32_t)r->sc.sampleSets.size());
#endif
    Dispatch(dimX, dimY, 8, 8, true, [](uint3 DTid, uint3 Gid, uint3 GTid, uint32_t) { hlsl::main(DTid, Gid, GTid); });
    return 0;
}
