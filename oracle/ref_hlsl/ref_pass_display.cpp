// ORACLE tooling -- test infrastructure only.
// The reference's display pixel shader (RP/Display/Display.hlsl mainPS, with Tonemap.hlsli) compiled as C++ and run once per display
// pixel: SV_Position = the pixel centre, as the rasteriser of the full-screen triangle produces it.
#include "ref_pass_common.h"
#include "ref_dispatch.h"

namespace hlsl {
#include "ZetaRenderPass/Common/FrameConstants.h"
#include "ZetaRenderPass/Display/Display.hlsl"
}
using namespace refpass;

// d->root_uav: float4 per display pixel (the render target before the back buffer's format conversion)
extern "C" void zrefp_shader_display(const ZrDispatch* d)
{
    g_heapPtr = (DescriptorHeap*)d->heap;
    memcpy(&hlsl::g_frame, d->frame_cb, sizeof(zr_frame_constants));
    if (d->local_cb_bytes != sizeof(hlsl::cbDisplayPass)) { std::fprintf(stderr, "Display.hlsl: local constant buffer is %u B, shader expects %zu B\n", d->local_cb_bytes, sizeof(hlsl::cbDisplayPass)); std::abort(); }
    memcpy(&hlsl::g_local, d->local_cb, sizeof(hlsl::cbDisplayPass));
    float* out = (float*)d->root_uav;
    const uint32_t w = hlsl::g_frame.DisplayWidth, h = hlsl::g_frame.DisplayHeight;
    for (uint32_t y = 0; y < h; y++) for (uint32_t x = 0; x < w; x++)
    {
        hlsl::VSOut psin;
        psin.PosSS = float4((float)x + 0.5f, (float)y + 0.5f, 0.0f, 1.0f);
        psin.TexCoord = float2(((float)x + 0.5f) / (float)w, ((float)y + 0.5f) / (float)h);
        const float4 c = hlsl::mainPS(psin);
        float* o = out + 4 * ((size_t)y * w + x);
        o[0] = c.x; o[1] = c.y; o[2] = c.z; o[3] = c.w;
    }
}
