// ORACLE tooling -- test infrastructure only.
// One reference compute shader compiled as C++, behind a uniform entry point.  Compiled once per shader permutation by oracle/_ref.mk:
//   -DZR_SHADER='"ZetaRenderPass/.../X.hlsl"'  the shader file (its rewritten copy under _ref/gen)
//   -DZR_ENTRY=zrefp_shader_<tag>  -Dhlsl=hlsl_<tag>   a private namespace per shader, so that several shaders link into one library
//   -DZR_LOCAL_CB=<type of g_local>   -DZR_HAS_SCENE=0/1 (g_bvh + geometry buffers)   -DZR_HAS_LIGHTS=0/1 (emissives, alias table, sample sets)
//   plus the shader's own permutation macros (NEE_EMISSIVE, USE_PRESAMPLED_SETS, TEMPORAL_TO_CURRENT, ...), exactly the reference's Variants/*.hlsl
#include "ref_pass_common.h"
#include "ref_dispatch.h"

namespace hlsl {
#include "ZetaRenderPass/Common/FrameConstants.h"
#include ZR_SHADER
}

using namespace refpass;

extern "C" void ZR_ENTRY(const ZrDispatch* d)
{
    static_assert(sizeof(hlsl::cbFrameConstants) == sizeof(zr_frame_constants), "cbFrameConstants layout");
    RefScene* r = (RefScene*)d->scene;
    g_heapPtr = (DescriptorHeap*)d->heap;
    memcpy(&hlsl::g_frame, d->frame_cb, sizeof(zr_frame_constants));
    if (d->local_cb_bytes != sizeof(hlsl::ZR_LOCAL_CB)) { std::fprintf(stderr, "%s: local constant buffer is %u B, shader expects %zu B\n", ZR_SHADER, d->local_cb_bytes, sizeof(hlsl::ZR_LOCAL_CB)); std::abort(); }
    memcpy(&hlsl::g_local, d->local_cb, sizeof(hlsl::ZR_LOCAL_CB));
#ifdef ZR_ROOT_UAV
    hlsl::ZR_ROOT_UAV = RWByteAddressBuffer(d->root_uav);      // e.g. -DZR_ROOT_UAV=g_hist
#endif
#ifdef ZR_DI_GLOBALS
    // DirectLighting / SkyDI root signatures (DirectLighting.cpp:60-98, SkyDI.cpp:48-70): 1 = emissive temporal, 2 = emissive spatial,
    // 3 = sky temporal, 4 = sky spatial
    {
        const zro::Scene& scPrev = d->prev_scene ? ((RefScene*)d->prev_scene)->sc : r->sc;
        hlsl::g_bvh.scene = &r->sc;
#if ZR_DI_GLOBALS == 1 || ZR_DI_GLOBALS == 3
        hlsl::g_bvh_prev.scene = &scPrev;
#endif
#if ZR_DI_GLOBALS == 1 || ZR_DI_GLOBALS == 2
        hlsl::g_emissives = StructuredBuffer<hlsl::RT::EmissiveTriangle>((const hlsl::RT::EmissiveTriangle*)r->sc.emissives.data(), (uint32_t)r->sc.emissives.size());
        hlsl::g_frameMeshData = StructuredBuffer<hlsl::RT::MeshInstance>((const hlsl::RT::MeshInstance*)r->sc.instances.data(), (uint32_t)r->sc.instances.size());
#endif
#if ZR_DI_GLOBALS == 1
        hlsl::g_aliasTable = StructuredBuffer<hlsl::RT::EmissiveLumenAliasTableEntry>((const hlsl::RT::EmissiveLumenAliasTableEntry*)r->sc.alias.data(), (uint32_t)r->sc.alias.size());
#ifdef USE_PRESAMPLED_SETS
        hlsl::g_sampleSets = StructuredBuffer<hlsl::RT::PresampledEmissiveTriangle>((const hlsl::RT::PresampledEmissiveTriangle*)r->sc.sampleSets.data(), (uint32_t)r->sc.sampleSets.size());
#endif
#endif
    }
#endif
#if ZR_HAS_SCENE
    static_assert(sizeof(hlsl::RT::MeshInstance) == sizeof(zr_mesh_instance) && sizeof(hlsl::Vertex) == sizeof(zr_vertex) && sizeof(hlsl::Material) == sizeof(zr_material), "wire layouts");
    const zro::Scene& sc = d->use_prev_scene && d->prev_scene ? ((RefScene*)d->prev_scene)->sc : r->sc;
    hlsl::g_bvh.scene = &sc;
    hlsl::g_frameMeshData = StructuredBuffer<hlsl::RT::MeshInstance>((const hlsl::RT::MeshInstance*)sc.instances.data(), (uint32_t)sc.instances.size());
    hlsl::g_vertices = StructuredBuffer<hlsl::Vertex>((const hlsl::Vertex*)r->sc.vertices.data(), (uint32_t)r->sc.vertices.size());
    hlsl::g_indices = StructuredBuffer<hlsl::uint>(r->sc.indices.data(), (uint32_t)r->sc.indices.size());
    hlsl::g_materials = StructuredBuffer<hlsl::Material>((const hlsl::Material*)r->sc.materials.data(), (uint32_t)r->sc.materials.size());
#endif
#if ZR_HAS_LIGHTS
    static_assert(sizeof(hlsl::RT::EmissiveTriangle) == sizeof(zr_emissive_triangle), "EmissiveTriangle layout");
    hlsl::g_emissives = StructuredBuffer<hlsl::RT::EmissiveTriangle>((const hlsl::RT::EmissiveTriangle*)r->sc.emissives.data(), (uint32_t)r->sc.emissives.size());
    hlsl::g_aliasTable = StructuredBuffer<hlsl::RT::EmissiveLumenAliasTableEntry>((const hlsl::RT::EmissiveLumenAliasTableEntry*)r->sc.alias.data(), (uint32_t)r->sc.alias.size());
    hlsl::g_sampleSets = StructuredBuffer<hlsl::RT::PresampledEmissiveTriangle>((const hlsl::RT::PresampledEmissiveTriangle*)r->sc.sampleSets.data(), (uint32_t)r->sc.sampleSets.size());
#endif
#ifdef USE_LVG
    // Variants/ReSTIR_GI_LVG.hlsl: g_lvg : register(t8), the grid PreLighting built this frame (IndirectLighting.cpp:344-351)
    static_assert(sizeof(hlsl::RT::VoxelSample) == sizeof(zr_voxel_sample), "VoxelSample layout");
    hlsl::g_lvg = StructuredBuffer<hlsl::RT::VoxelSample>((const hlsl::RT::VoxelSample*)r->sc.lvgData.data(), (uint32_t)r->sc.lvgData.size());
#endif
    Dispatch(d->groups_x, d->groups_y, hlsl::zr_numthreads[0], hlsl::zr_numthreads[1], true,
        [](uint3 DTid, uint3 Gid, uint3 GTid, uint32_t Gidx) { hlsl::zr_main_dispatch(DTid, Gid, GTid, Gidx); });
}
