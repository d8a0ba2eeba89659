// ORACLE tooling -- test infrastructure only.
// Auto-exposure and display from the reference's own shaders (AutoExposure_Histogram.hlsl, AutoExposure_WeightedAvg.hlsl compiled by
// ref_pass_shader.cpp; Display.hlsl by ref_pass_display.cpp) driven by a restatement of AutoExposure::Render (AutoExposure.cpp:100-143:
// clear the histogram, HISTOGRAM over ceil(w / 16) x ceil(h / 16) groups, WEIGHTED_AVG in one group) and DisplayPass::Render
// (Display.cpp:188-260).  libzref_post.so
#include "ref_pass_common.h"
#include "ref_dispatch.h"
namespace hlsl {
#include "ZetaRenderPass/Common/FrameConstants.h"
#include "ZetaRenderPass/AutoExposure/AutoExposure_Common.h"
#include "ZetaRenderPass/Display/Display_Common.h"
}
using namespace refpass;
extern "C" void zrefp_shader_ae_hist(const ZrDispatch*);
extern "C" void zrefp_shader_ae_avg(const ZrDispatch*);
extern "C" void zrefp_shader_display(const ZrDispatch*);

namespace {
enum : uint32_t { POST_INPUT = SLOT_PASS, POST_EXPOSURE, POST_LUT };
DescriptorHeap g_heap;
float g_dummyDepth = 1.0f;
void BindInput(const void* image, int is_f16, uint32_t w, uint32_t h, std::vector<uint16_t>& tmp)
{
    // an RGBA32F image stands for the R16G16B16A16_FLOAT texture holding its half-rounded values
    const void* data = image;
    if (!is_f16)
    {
        const float* f = (const float*)image;
        tmp.resize((size_t)w * h * 4);
        for (size_t i = 0; i < tmp.size(); i++) tmp[i] = zr_f32_to_f16(f[i]);
        data = tmp.data();
    }
    BindPlane(g_heap, POST_INPUT, (void*)data, w, h, FMT_RGBA16_FLOAT);
}
}

extern "C" {
// prm4 = MinLum, LumRange, LumMapExp, AdaptationRate; exposure2 = the persistent R32G32_FLOAT texel (read + written)
int zrefp_auto_exposure(const void* image, int is_f16, uint32_t w, uint32_t h, const zr_frame_constants* cb, const float* prm4, uint32_t* hist256, float* exposure2)
{
    std::vector<uint16_t> tmp;
    g_heapPtr = &g_heap;
    BindInput(image, is_f16, w, h, tmp);
    BindPlane(g_heap, POST_EXPOSURE, exposure2, 1, 1, FMT_RG32_FLOAT);
    using namespace hlsl;
    cbAutoExposureHist L; memset(&L, 0, sizeof(L));
    L.InputDescHeapIdx = POST_INPUT; L.ExposureDescHeapIdx = POST_EXPOSURE;
    L.MinLum = prm4[0]; L.LumRange = prm4[1]; L.LumMapExp = prm4[2]; L.AdaptationRate = prm4[3];
    L.LowerPercentile = 0.01f; L.UpperPercentile = 0.9f;
    zr_frame_constants g = *cb;
    memset(hist256, 0, HIST_BIN_COUNT * sizeof(uint32_t));      // CopyBufferRegion from the zero buffer
    ZrDispatch d; memset(&d, 0, sizeof(d));
    d.heap = &g_heap; d.frame_cb = &g; d.local_cb = &L; d.local_cb_bytes = sizeof(L); d.root_uav = hist256;
    d.groups_x = (w + THREAD_GROUP_SIZE_HIST_X - 1) / THREAD_GROUP_SIZE_HIST_X; d.groups_y = (h + THREAD_GROUP_SIZE_HIST_Y - 1) / THREAD_GROUP_SIZE_HIST_Y;
    zrefp_shader_ae_hist(&d);
    d.groups_x = 1; d.groups_y = 1;
    zrefp_shader_ae_avg(&d);
    return 0;
}

// image: rw x rh; cb carries the display size; out_rgba: display_width x display_height float4
int zrefp_display(const void* image, int is_f16, uint32_t rw, uint32_t rh, const zr_frame_constants* cb, float* exposure2, uint32_t tonemapper, uint32_t autoExposure,
    float saturation, float agxExp, const uint32_t* lut, uint32_t lutDim, float* out_rgba)
{
    std::vector<uint16_t> tmp;
    g_heapPtr = &g_heap;
    BindInput(image, is_f16, rw, rh, tmp);
    if (exposure2) BindPlane(g_heap, POST_EXPOSURE, exposure2, 1, 1, FMT_RG32_FLOAT);
    if (lut) { TexStorage& t = g_heap.table[POST_LUT]; t.data = (void*)lut; t.w = t.h = t.d = lutDim; t.fmt = FMT_R9G9B9E5; }
    // mainPS fetches the depth plane unconditionally (it only matters for the debug views): a 1 x 1 stand-in
    BindPlane(g_heap, SLOT_GBUF_CURR + 7, &g_dummyDepth, 1, 1, FMT_R32_FLOAT);
    using namespace hlsl;
    cbDisplayPass L; memset(&L, 0, sizeof(L));
    L.DisplayOption = (uint16_t)DisplayOption::DEFAULT; L.Tonemapper = (uint16_t)tonemapper; L.AutoExposure = (uint16_t)autoExposure;
    L.InputDescHeapIdx = POST_INPUT; L.ExposureDescHeapIdx = POST_EXPOSURE; L.LUTDescHeapIdx = POST_LUT;
    L.Saturation = saturation; L.AgXExp = agxExp; L.RoughnessTh = 1.0f;
    zr_frame_constants g = *cb;
    g.curr_gbuffer_desc_heap_offset = SLOT_GBUF_CURR;
    ZrDispatch d; memset(&d, 0, sizeof(d));
    d.heap = &g_heap; d.frame_cb = &g; d.local_cb = &L; d.local_cb_bytes = sizeof(L); d.root_uav = out_rgba;
    zrefp_shader_display(&d);
    return 0;
}
}
