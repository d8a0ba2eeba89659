// ORACLE tooling -- test infrastructure only.
// K1 from the reference's own shader: Source/ZetaRenderPass/GBuffer/GBufferRT_Inline.hlsl (+ GBufferRT.hlsli) compiled as C++.
#include "ref_pass_common.h"

namespace hlsl {
#include "ZetaRenderPass/Common/FrameConstants.h"
#include "ZetaRenderPass/GBuffer/GBufferRT_Common.h"
#include "ZetaRenderPass/GBuffer/GBufferRT_Inline.hlsl"
}

using namespace refpass;

ZREFP_SCENE_API

extern "C" {

// one frame of K1: DispatchThreads(RenderWidth x RenderHeight) of main(DTid); planes are zeroed first like the product / oracle define
// pick_x == 0xffff: no pick pending (GBufferRT.h:36-46); otherwise *picked receives what the shader wrote to g_pick[0] (GBufferRT_Inline.hlsl:241-242)
int zrefp_gbuffer_render_pick(RefScene* r, const zr_frame_constants* cb, zr_gbuffer_planes* planes, uint32_t pick_x, uint32_t pick_y, uint32_t* picked)
{
    static_assert(sizeof(hlsl::cbFrameConstants) == sizeof(zr_frame_constants), "cbFrameConstants layout");
    static_assert(sizeof(hlsl::RT::MeshInstance) == sizeof(zr_mesh_instance), "MeshInstance layout");
    static_assert(sizeof(hlsl::Vertex) == sizeof(zr_vertex), "Vertex layout");
    static_assert(sizeof(hlsl::Material) == sizeof(zr_material), "Material layout");
    BindScene(r);
    r->sc.LatchHeapOffsets(*cb);
    for (int i = 0; i < ZR_GB_COUNT; i++) memset(planes->plane[i], 0, (size_t)planes->width * planes->height * ZR_GB_PLANE_BYTES[i]);
    BindGBuffer(r->heap, SLOT_GBUF_UAV, planes);
    memcpy(&hlsl::g_frame, cb, sizeof(zr_frame_constants));
    hlsl::g_frame.BaseColorMapsDescHeapOffset += SLOT_TEXTURES; hlsl::g_frame.NormalMapsDescHeapOffset += SLOT_TEXTURES;
    hlsl::g_frame.MetallicRoughnessMapsDescHeapOffset += SLOT_TEXTURES; hlsl::g_frame.EmissiveMapsDescHeapOffset += SLOT_TEXTURES;
    hlsl::g_local.UavTableDescHeapIdx = SLOT_GBUF_UAV;
    hlsl::g_local.PickedPixelX = (uint16_t)pick_x; hlsl::g_local.PickedPixelY = (uint16_t)pick_y;
    hlsl::g_bvh.scene = &r->sc;
    hlsl::g_frameMeshData = StructuredBuffer<hlsl::RT::MeshInstance>((const hlsl::RT::MeshInstance*)r->sc.instances.data(), (uint32_t)r->sc.instances.size());
    hlsl::g_sceneVertices = StructuredBuffer<hlsl::Vertex>((const hlsl::Vertex*)r->sc.vertices.data(), (uint32_t)r->sc.vertices.size());
    hlsl::g_sceneIndices = StructuredBuffer<hlsl::uint>(r->sc.indices.data(), (uint32_t)r->sc.indices.size());
    hlsl::g_materials = StructuredBuffer<hlsl::Material>((const hlsl::Material*)r->sc.materials.data(), (uint32_t)r->sc.materials.size());
    static uint32_t pick[4]; pick[0] = 0xfffffffeu; hlsl::g_pick = RWStructuredBuffer<hlsl::uint>(pick, 4);
    for (uint32_t y = 0; y < planes->height; y++)
        for (uint32_t x = 0; x < planes->width; x++)
        { hlsl::main(hlsl::uint3(x, y, 0)); FlushPendingRW(); }
    if (picked) *picked = pick[0];
    return 0;
}
int zrefp_gbuffer_render(RefScene* r, const zr_frame_constants* cb, zr_gbuffer_planes* planes)
{ return zrefp_gbuffer_render_pick(r, cb, planes, 0xffffu, 0xffffu, nullptr); }

} // extern "C"
