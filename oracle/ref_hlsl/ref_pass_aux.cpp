// ORACLE tooling -- test infrastructure only.
// One of the reference's auxiliary compute shaders compiled as C++ behind the uniform ZrDispatch entry point (like ref_pass_shader.cpp, for
// shaders whose root signatures name their own buffers).  -DZR_AUX selects the shader's bindings:
//   1  PreLighting/EstimateTriEmissivePower.hlsl   buf0 = g_emissvies, buf1 = g_halton (float2), buf2 = g_power (out); [WaveSize(32)]
//   2  PreLighting/PresampleEmissives.hlsl         buf0 = g_emissives, buf1 = g_aliasTable, buf2 = g_sampleSets (out)
//   3  PreLighting/BuildLightVoxelGrid.hlsl        buf0 = g_emissives, buf1 = g_aliasTable, buf2 = g_voxel (out); 3-D dispatch
//   4  Sky/SkyViewLUT.hlsl, 5 Compositing/Compositing.hlsl, 6 Compositing/FireflyFilter.hlsl, 7 TAA/TAA.hlsl: descriptor heap only
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
    g_heapPtr = (DescriptorHeap*)d->heap;
    memcpy(&hlsl::g_frame, d->frame_cb, sizeof(zr_frame_constants));
#ifdef ZR_LOCAL_CB
    if (d->local_cb_bytes != sizeof(hlsl::ZR_LOCAL_CB)) { std::fprintf(stderr, "%s: local constant buffer is %u B, shader expects %zu B\n", ZR_SHADER, d->local_cb_bytes, sizeof(hlsl::ZR_LOCAL_CB)); std::abort(); }
    memcpy(&hlsl::g_local, d->local_cb, sizeof(hlsl::ZR_LOCAL_CB));
#endif
    int waveSize = 64; bool sumAscending = false;
#if ZR_AUX == 1
    static_assert(sizeof(hlsl::RT::EmissiveTriangle) == sizeof(zr_emissive_triangle), "EmissiveTriangle layout");
    hlsl::g_emissvies = StructuredBuffer<hlsl::RT::EmissiveTriangle>((const hlsl::RT::EmissiveTriangle*)d->buf[0], d->buf_count[0]);
    hlsl::g_halton = StructuredBuffer<hlsl::float2>((const hlsl::float2*)d->buf[1], d->buf_count[1]);
    hlsl::g_power = RWStructuredBuffer<float>((float*)d->buf[2], d->buf_count[2]);
    waveSize = ESTIMATE_TRI_POWER_WAVE_LEN;      // [WaveSize(32)]: wave = Gidx / 32, two Halton points per lane
    sumAscending = true;                         // the ABI sums the lane partials in ascending lane order (the order is the driver's on a GPU)
#elif ZR_AUX == 2 || ZR_AUX == 3
    static_assert(sizeof(hlsl::RT::EmissiveTriangle) == sizeof(zr_emissive_triangle) && sizeof(hlsl::RT::EmissiveLumenAliasTableEntry) == sizeof(zr_alias_entry), "light record layouts");
    hlsl::g_emissives = StructuredBuffer<hlsl::RT::EmissiveTriangle>((const hlsl::RT::EmissiveTriangle*)d->buf[0], d->buf_count[0]);
    hlsl::g_aliasTable = StructuredBuffer<hlsl::RT::EmissiveLumenAliasTableEntry>((const hlsl::RT::EmissiveLumenAliasTableEntry*)d->buf[1], d->buf_count[1]);
#if ZR_AUX == 2
    static_assert(sizeof(hlsl::RT::PresampledEmissiveTriangle) == sizeof(zr_presampled_tri), "PresampledEmissiveTriangle layout");
    hlsl::g_sampleSets = RWStructuredBuffer<hlsl::RT::PresampledEmissiveTriangle>((hlsl::RT::PresampledEmissiveTriangle*)d->buf[2], d->buf_count[2]);
#else
    static_assert(sizeof(hlsl::RT::VoxelSample) == sizeof(zr_voxel_sample), "VoxelSample layout");
    hlsl::g_voxel = RWStructuredBuffer<hlsl::RT::VoxelSample>((hlsl::RT::VoxelSample*)d->buf[2], d->buf_count[2]);
#endif
#endif
    static thread_local GroupRunner runner;
    runner.kWave = waveSize; runner.sumAscending = sumAscending;
    const uint32_t tx = hlsl::zr_numthreads[0], ty = hlsl::zr_numthreads[1], gz = d->groups_z ? d->groups_z : 1u;
    for (uint32_t Gz = 0; Gz < gz; Gz++)
        for (uint32_t Gy = 0; Gy < d->groups_y; Gy++)
            for (uint32_t Gx = 0; Gx < d->groups_x; Gx++)
                runner.Run((int)(tx * ty), [&](int i) {
                    const uint32_t lx = (uint32_t)i % tx, ly = (uint32_t)i / tx;
                    hlsl::zr_main_dispatch(uint3(Gx * tx + lx, Gy * ty + ly, Gz), uint3(Gx, Gy, Gz), uint3(lx, ly, 0), (uint32_t)i);
                });
    runner.kWave = 64; runner.sumAscending = false;
}
