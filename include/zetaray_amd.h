/*
 * zetaray_amd.h -- C-ABI of the MI355X-native ReSTIR path-tracing core (libzetaray_amd.so).
 *
 * Drop-in boundary for the hot path of alipbcs/ZetaRay's render passes.  Each entry point replaces one piece of
 * the reference's RenderPass surface (Source/ZetaRenderPass/RenderPass.h:9-80):
 *
 *   zr_scene_*        <- buffers a pass fetches from SharedShaderResources by name
 *                        (Source/ZetaCore/Scene/SceneRenderer.h:17-32; RtAccelerationStructure.cpp:121-200, 318-506)
 *   zr_gbuffer_*      <- GBufferData textures (Source/ZetaRenderer/Default/DefaultRendererImpl.h:80-130,
 *                        Source/ZetaRenderer/Default/GBuffer.cpp:41-76)
 *   zr_pass_create/init/resize/reset_temporal/set_params/render/get_output/destroy
 *                     <- Pass::Pass(), Init(), OnWindowResized(), ResetTemporal(), Set*(), Render(CommandList&),
 *                        GetOutput(SHADER_OUT_RES), Reset()
 *                        (GBufferRT.h:27-47, PreLighting.h:28-58,97-128, IndirectLighting.h:72-108, DirectLighting.h:39-57)
 *
 * Conventions: plain pointers and sizes, no torch / D3D12 types.  All functions return 0 on success or a nonzero
 * zr_status; zr_last_error() returns a thread-local message.  (The reference aborts through Check(); the C++
 * RenderPass-shaped wrapper in zetaray_amd/host converts nonzero into that behaviour.)  zr_pass_render only
 * enqueues work on the given hipStream_t and never synchronises the host -- the analogue of recording into a
 * CommandList.  Distinct handles may be driven from distinct host threads; one handle is externally serialised
 * (same contract as Source/ZetaCore/Core/RenderGraph.cpp:442-541).
 *
 * There is no CPU fallback: every compute entry point fails with ZR_ERR_NO_DEVICE when no HIP device is usable.
 */
#ifndef ZETARAY_AMD_H
#define ZETARAY_AMD_H

#include <stddef.h>
#include <stdint.h>
#include "zr_wire.h"

#ifdef __cplusplus
extern "C" {
#endif

/* 2: zr_params grew (ae_*, display_*, tex_filter); stream-ordered zr_scene_*_async entry points; zr_scene_get_presampled_sets.
 * A caller built against another version must not pass its zr_params to zr_pass_set_params: check zr_abi_version() first. */
#define ZR_ABI_VERSION 3

typedef enum zr_status {
    ZR_OK = 0,
    ZR_ERR_INVALID_ARG = 1,
    ZR_ERR_NO_DEVICE = 2,
    ZR_ERR_HIP = 3,
    ZR_ERR_OOM = 4,
    ZR_ERR_UNSUPPORTED = 5,
    ZR_ERR_NOT_INITIALIZED = 6
} zr_status;

typedef struct zr_scene   zr_scene;
typedef struct zr_gbuffer zr_gbuffer;
typedef struct zr_pass    zr_pass;

/* pass kinds: the four hot-path RenderPass nodes of the reference */
typedef enum zr_pass_kind {
    ZR_PASS_GBUFFER     = 0,   /* GBufferRT          */
    ZR_PASS_PRELIGHTING = 1,   /* PreLighting + EmissiveTriangleAliasTable */
    ZR_PASS_DI_EMISSIVE = 2,   /* DirectLighting     */
    ZR_PASS_DI_SKY      = 3,   /* SkyDI (RP/DirectLighting/Sky/SkyDI.cpp): zr_params.m_max_temporal = M_max (Sky), m_max_spatial = M_max (Sun),
                                  alpha_min = Alpha_min; needs the scene's sky-view LUT (ZR_PASS_SKY) */
    ZR_PASS_INDIRECT    = 4,   /* IndirectLighting   */
    ZR_PASS_COMPOSITING = 5,   /* Compositing: (DI + indirect * !emissive) / NumFramesCameraStatic */
    /* Sky (RP/Sky/Sky.cpp:34-66,120-164; K17 RP/Sky/SkyViewLUT.hlsl): zr_pass_init(pass, LutWidth, LutHeight, 0) (the reference
       uses 256 x 128, DefaultRendererImpl.h:165-166); zr_pass_render writes the sky-view LUT from cbFrameConstants' sun and
       atmosphere fields and binds it to the scene, where Le_Sky of every later pass samples it (the reference does the same
       through EnvMapDescHeapOffset).  Pinned: R11G11B10_FLOAT store rounds to nearest even; the LUT is sampled with fp32 bilinear
       interpolation, texel centres at (i + 0.5) / N, wrap addressing.  Inscattering voxel grid: out of scope (post stack). */
    ZR_PASS_SKY         = 6,
    /* TAA (RP/TAA/TAA.cpp, TAA.hlsl): temporal anti-aliasing of the composited image.  Input: an
       RGBA32F image bound with zr_pass_set_input(ZR_IN_TAA_SIGNAL) -- e.g. the COMPOSITING pass's FINAL -- plus the depth and
       motion-vector planes of the gbuffer passed to zr_pass_render.  Output: ZR_OUT_TAA, R16G16B16A16_FLOAT like the reference's
       two ping-pong targets (TAA.cpp:120-146); zr_params.taa_blend_weight = cbTAA.BlendWeight (default 0.1, TAA.h:72);
       zr_pass_reset_temporal = TemporalIsValid 0 for the next frame. */
    ZR_PASS_TAA         = 7,
    /* AutoExposure (RP/AutoExposure/AutoExposure.cpp:100-143): 256-bin log-luminance histogram of the image bound with
       zr_pass_set_input(ZR_IN_POST_SIGNAL_F16 / _F32), then the bin-weighted mean, exponential adaptation over cbFrameConstants::dt and
       the ISO-100 exposure.  zr_params.ae_*.  Outputs ZR_OUT_EXPOSURE (persistent across frames), ZR_OUT_AE_HISTOGRAM.
       The pass size is the render size; zr_pass_render needs no gbuffer (pass NULL). */
    ZR_PASS_AUTO_EXPOSURE = 8,
    /* DisplayPass (RP/Display/Display.hlsl:41-77, DisplayOption::DEFAULT; Tonemap.hlsli): exposure x tone mapper over the image bound
       with ZR_IN_POST_SIGNAL_*, point-sampled from render to display resolution.  The pass size is the DISPLAY size; the input has
       cb.render_width x render_height texels.  zr_params.display_*; ZR_IN_DISPLAY_EXPOSURE = ZR_OUT_EXPOSURE of the auto-exposure pass;
       the NEUTRAL tone mapper needs zr_pass_set_tonemap_lut.  Outputs ZR_OUT_DISPLAY (the pixel shader's float4) and
       ZR_OUT_DISPLAY_SRGB8 (what the reference's R8G8B8A8_UNORM_SRGB back buffer stores). */
    ZR_PASS_DISPLAY     = 9,
    /* Denoise: a spatiotemporal variance-guided filter (after Schied et al. 2017) over a noisy radiance image -- temporal accumulation of colour and
       luminance moments along the G-buffer's motion vectors, a variance estimate, zr_params.svgf_iterations a-trous iterations with depth / normal /
       luminance edge stopping.  NO REFERENCE COUNTERPART (the reference presents ReSTIR PT through TAA / FSR2): the pass exists for BASELINE.json's
       config 5 ("ReSTIR PT + SVGF denoise tile pass" at 3840 x 2160) and its arithmetic is defined by this library (zetaray_amd/csrc/zr_svgf.h;
       restated in oracle/zro_svgf.h).  Input: an RGBA32F image bound with zr_pass_set_input(ZR_IN_DENOISE_SIGNAL) -- e.g. the INDIRECT pass's
       FINAL -- plus the depth / normal / motion planes of the gbuffer passed to zr_pass_render and the same gbuffer's previous-frame planes.
       Output ZR_OUT_DENOISED; zr_pass_reset_temporal drops the history. */
    ZR_PASS_DENOISE     = 10
} zr_pass_kind;

/* IndirectLighting::INTEGRATOR, reference IndirectLighting.h:40-46 */
typedef enum zr_integrator {
    ZR_INTEGRATOR_PATH_TRACING = 0,
    ZR_INTEGRATOR_RESTIR_GI    = 1,
    ZR_INTEGRATOR_RESTIR_PT    = 2
} zr_integrator;

/* CB_IND_FLAGS, reference IndirectLighting_Common.h:38-49 */
#define ZR_IND_TEMPORAL_RESAMPLE       (1u << 0)
#define ZR_IND_SPATIAL_RESAMPLE        (1u << 1)
#define ZR_IND_STOCHASTIC_MULTI_BOUNCE (1u << 2)
#define ZR_IND_RUSSIAN_ROULETTE        (1u << 3)
#define ZR_IND_BOILING_SUPPRESSION     (1u << 4)
#define ZR_IND_PATH_REGULARIZATION     (1u << 5)
#define ZR_IND_SORT_TEMPORAL           (1u << 6)
#define ZR_IND_SORT_SPATIAL            (1u << 7)
/* ReSTIR DI (ZR_PASS_DI_EMISSIVE) reads TEMPORAL_RESAMPLE / SPATIAL_RESAMPLE above plus (CB_RDI_FLAGS, DirectLighting_Common.h:13-20): */
#define ZR_DI_STOCHASTIC_SPATIAL          (1u << 8)
#define ZR_DI_EXTRA_DISOCCLUSION_SAMPLING (1u << 9)
/* ... and the reference's COMPILE-time switch USE_HALF_VECTOR_COPY_SHIFT (DirectLighting/Emissive/Params.hlsli:12; 0 in the reference's tree) as a run-time flag: BSDF-sampled
   candidates of lobes narrower than alpha_min are reused by copying their half vector in the shading frame and re-tracing the reflected ray (Reservoir.hlsli:56-119, 156-184;
   Resampling.hlsli:138-274; PairwiseMIS.hlsli:60-170); reservoir plane A then carries the flag and lobe in its metadata bits 5..8 and the oct-encoded half vector in
   place of the barycentrics.  Off by default, like the reference.  Pinned against the reference's shaders compiled with the macro at 1 (tests/golden/ref_pass_di_half_vector*.npz). */
#define ZR_DI_HALF_VECTOR_COPY_SHIFT      (1u << 11)
/* Compositing (ZR_PASS_COMPOSITING): run the firefly filter on the composited image (Compositing::SetFireflyFilterEnablement,
   RP/Compositing/FireflyFilter.hlsl).  Pinned: every pixel reads the unfiltered image (the reference filters its UAV in place). */
#define ZR_COMPOSIT_FIREFLY_FILTER        (1u << 10)

/* Pass parameters.  Defaults = the reference's (IndirectLighting.h:231-244, IndirectLighting.cpp:146-165). */
typedef struct zr_params {
    uint32_t flags;                 /* ZR_IND_* */
    uint32_t max_non_tr_bounces;    /* 3 */
    uint32_t max_glossy_tr_bounces; /* 4 */
    uint32_t m_max_temporal;        /* 10 */
    uint32_t m_max_spatial;         /* 8 */
    float    alpha_min;             /* 0.175^2 */
    uint32_t presampling;           /* SetLightPresamplingParams(bool, numSets, setSize) */
    uint32_t num_sample_sets;       /* 128 */
    uint32_t sample_set_size;       /* 512 */
    /* Light voxel grid (K4; PreLighting::SetLightVoxelGridParams / IndirectLighting::SetLightVoxelGridParams, off by default like
       DefaultRendererImpl.h:73-77; needs presampling, IndirectLighting.h:93).  PRELIGHTING builds the grid every frame around the
       camera; ReSTIR GI then samples lights from it on bounces > 0 (the ReSTIR_GI_LVG shader variant). */
    uint32_t use_lvg;               /* 0 */
    uint32_t lvg_grid_dim;          /* x | y << 10 | z << 20; reference default (32, 8, 40) */
    float    lvg_extents[3];        /* voxel half extents; reference default (0.6, 0.45, 0.6) */
    float    lvg_offset_y;          /* 0.1 */
    float    taa_blend_weight;      /* ZR_PASS_TAA: cbTAA.BlendWeight, 0.1 */
    /* ZR_PASS_AUTO_EXPOSURE: cbAutoExposureHist (AutoExposure_Common.h:11-21), defaults AutoExposure.h:73-81 */
    float    ae_min_lum;            /* 5e-3 */
    float    ae_max_lum;            /* 4.0 (LumRange = max - min) */
    float    ae_lum_map_exp;        /* 0.5 */
    float    ae_adaptation_rate;    /* 1.0 */
    /* ZR_PASS_DISPLAY: cbDisplayPass (Display_Common.h:32-46), defaults Display.cpp:69-74 */
    uint32_t display_tonemapper;    /* zr_tonemapper, NEUTRAL */
    uint32_t display_auto_exposure; /* 1 */
    float    display_saturation;    /* 1.0 (NEUTRAL, AgX_CUSTOM) */
    float    display_agx_exp;       /* 1.0 (AgX_CUSTOM) */
    /* ZR_PASS_INDIRECT: enum class TEXTURE_FILTER (IndirectLighting_Common.h:69-77), the sampler of the material maps at path vertices
       (cb_ReSTIR_*::TexFilterDescHeapIdx): ZR_TEX_FILTER_MIP0 / TRI_LINEAR / ANISOTROPIC_2X / ANISOTROPIC_4X / ANISOTROPIC_16X (zr_texture.h) */
    uint32_t tex_filter;            /* ZR_TEX_FILTER_ANISOTROPIC_4X = 3 */
    /* ZR_PASS_DENOISE (no reference counterpart; zr_svgf.h) */
    float    svgf_alpha;            /* 0.2: floor of the colour blend factor (1 / history length above it) */
    float    svgf_alpha_moments;    /* 0.2: the same for the luminance moments */
    float    svgf_sigma_l;          /* 4.0: luminance edge-stopping, in standard deviations */
    float    svgf_sigma_z;          /* 1.0: depth edge-stopping, in units of the pixel's screen-space depth slope */
    uint32_t svgf_normal_power_log2;/* 7: normal weight = max(0, n . n_q) ^ (2 ^ 7) */
    uint32_t svgf_iterations;       /* 5 a-trous iterations (steps 1, 2, 4, 8, 16); 0..8 */
    /* ZR_PASS_INDIRECT, ReSTIR PT (ABI version 3): IndirectLighting::m_numSpatialPasses ("#Spatial Passes", range 0..2, IndirectLighting.cpp:1240,
       default 1 IndirectLighting.h:392).  0: no spatial reuse (the temporal pass writes the frame's radiance); 2: the host loop of
       IndirectLighting.cpp:616-621, 860-870 -- a second search / sort / replay / reconnect round whose inputs are the first round's outputs
       (reservoir sets swapped), the target plane not rewritten in between (ReSTIR_PT_Reconnect_StC.hlsl:328-346).  On tiles the second round is its own
       stage behind one more halo exchange (ZR_STAGE_SPATIAL2). */
    uint32_t num_spatial_passes;    /* 1 */
} zr_params;

/* enum class Tonemapper, Display_Common.h:21-30 */
typedef enum zr_tonemapper {
    ZR_TONEMAP_NONE = 0, ZR_TONEMAP_NEUTRAL = 1, ZR_TONEMAP_AGX_DEFAULT = 2, ZR_TONEMAP_AGX_GOLDEN = 3, ZR_TONEMAP_AGX_PUNCHY = 4,
    ZR_TONEMAP_AGX_CUSTOM = 5
} zr_tonemapper;

/* outputs, GetOutput(SHADER_OUT_RES) */
typedef enum zr_output {
    ZR_OUT_FINAL = 0,              /* RGBA32F, linear radiance (reference: *_FINAL textures) */
    /* ReSTIR PT persistent state, in the reference's texture formats (IndirectLighting.h:128-144, Reservoir.hlsli:267-456):
       the reservoir set the NEXT frame reads as "previous".  Exposed for parity tests and for a renderer that wants to
       checkpoint / migrate temporal history. */
    ZR_OUT_RPT_RESERVOIR_A = 1,    /* RGBA8_UINT   4 B: k | M << 4, lobes | lt_k << 6, lt_k+1 | motion << 2            */
    ZR_OUT_RPT_RESERVOIR_B = 2,    /* RG32F        8 B: w_sum, W                                                      */
    ZR_OUT_RPT_RESERVOIR_C = 3,    /* RGBA32_UINT 16 B                                                               */
    ZR_OUT_RPT_RESERVOIR_D = 4,    /* RGBA32_UINT 16 B                                                               */
    ZR_OUT_RPT_RESERVOIR_E = 5,    /* R16F         2 B                                                               */
    ZR_OUT_RPT_RESERVOIR_F = 6,    /* RG32F        8 B                                                               */
    ZR_OUT_RPT_RESERVOIR_G = 7,    /* RG32_UINT    8 B                                                               */
    ZR_OUT_RPT_TARGET      = 8,    /* RGBA32F     16 B (xyz)                                                         */
    ZR_OUT_RPT_NEIGHBOR    = 9,    /* RG8_UINT     2 B: spatial neighbour offset + 32, 255 = none                    */
    /* replay buffers of the last frame (scratch between K13 and K14/K16; Shift.hlsli:191-358): current-to-neighbour
       and neighbour-to-current, planes A (RGBA16F 8 B), B (RGBA32_UINT), C (RGBA32_UINT), D (R16_UINT) */
    ZR_OUT_RPT_RBUF_CTN_A  = 10, ZR_OUT_RPT_RBUF_CTN_B = 11, ZR_OUT_RPT_RBUF_CTN_C = 12, ZR_OUT_RPT_RBUF_CTN_D = 13,
    ZR_OUT_RPT_RBUF_NTC_A  = 14, ZR_OUT_RPT_RBUF_NTC_B = 15, ZR_OUT_RPT_RBUF_NTC_C = 16, ZR_OUT_RPT_RBUF_NTC_D = 17,
    /* K12 thread maps (ReSTIR_PT_Sort.hlsl; Util.hlsli:20-42), R16_UINT 2 B: x offset + 31 | (y offset + 31) << 7 | error << 15 of the pixel
       the thread at this position shifts; CtN = Sort_CtT then Sort_CtS, NtC = Sort_TtC then Sort_StC (DESC_TABLE_RPT::THREAD_MAP_*).
       Built under ZR_IND_SORT_TEMPORAL / ZR_IND_SORT_SPATIAL; wave order inside a bucket = wave index (the shader leaves it to the
       arrival order of an LDS atomic). */
    ZR_OUT_RPT_THREAD_MAP_CTN = 18, ZR_OUT_RPT_THREAD_MAP_NTC = 19,
    /* ReSTIR DI (ZR_PASS_DI_EMISSIVE) persistent state written by the last frame (Reservoir.hlsli:150-213) */
    ZR_OUT_RDI_RESERVOIR_A = 20,   /* RGBA32_UINT 16 B: bary unorm2, le.xy half2, le.z half | M << 16, lightIdx */
    ZR_OUT_RDI_RESERVOIR_B = 21,   /* RG32F        8 B: w_sum, W */
    ZR_OUT_RDI_TARGET      = 22,   /* RGBA32F     16 B (xyz; negated when the pixel was disoccluded) */
    /* ReSTIR GI persistent state written by the last frame (ReSTIR_GI/Reservoir.hlsli:72-131) */
    ZR_OUT_RGI_RESERVOIR_A = 30,   /* RGBA32F 16 B: sample position, hit ID bits */
    ZR_OUT_RGI_RESERVOIR_B = 31,   /* RGBA16F  8 B: Lo, M */
    ZR_OUT_RGI_RESERVOIR_C = 32,   /* RGBA32F 16 B: w_sum, W, oct32 normal bits, unused */
    /* sun + sky ReSTIR DI (ZR_PASS_DI_SKY) persistent state written by the last frame (DirectLighting/Sky/Reservoir.hlsli:137-164) */
    ZR_OUT_SDI_RESERVOIR_A = 24,   /* R8_UINT    1 B: M | sky << 4 | halfVectorCopyShift << 5 | lobe is coat << 6 | (w_sum > 0) << 7 */
    ZR_OUT_SDI_RESERVOIR_B = 25,   /* RG16_UINT  4 B: oct32 of wi (or of the tangent-frame half vector) */
    ZR_OUT_SDI_RESERVOIR_C = 26,   /* RG32F      8 B: w_sum, W */
    ZR_OUT_SDI_TARGET      = 27,   /* RGBA32F   16 B (xyz) */
    /* Sky (ZR_PASS_SKY) */
    ZR_OUT_SKY_LUT         = 40,   /* R11G11B10_FLOAT 4 B, LutWidth x LutHeight (Sky::SHADER_OUT_RES::SKY_VIEW_LUT) */
    /* TAA (ZR_PASS_TAA) */
    ZR_OUT_TAA             = 41,   /* RGBA16F 8 B: the anti-aliased image written by the last render (TAA::SHADER_OUT_RES::OUTPUT_A / _B) */
    /* AutoExposure (ZR_PASS_AUTO_EXPOSURE) */
    ZR_OUT_EXPOSURE        = 42,   /* RG32F 8 B, 1 x 1: exposure, adapted average luminance (AutoExposure::SHADER_OUT_RES::EXPOSURE) */
    ZR_OUT_AE_HISTOGRAM    = 43,   /* R32_UINT, 256 x 1: the last frame's histogram (bin 0 = luminance <= 1e-4) */
    /* Display (ZR_PASS_DISPLAY) */
    ZR_OUT_DISPLAY         = 44,   /* RGBA32F 16 B: mainPS's return value (linear, before the back buffer's sRGB encode) */
    ZR_OUT_DISPLAY_SRGB8   = 45,   /* RGBA8 4 B: sRGB-encoded, as the reference's R8G8B8A8_UNORM_SRGB back buffer stores it */
    /* Denoise (ZR_PASS_DENOISE) */
    ZR_OUT_DENOISED        = 46,   /* RGBA32F 16 B: filtered radiance, a = its variance estimate */
    ZR_OUT_DENOISE_HISTORY = 47,   /* RGBA32F 16 B: colour history the next frame accumulates into (rgb after the first a-trous iteration), a = history length */
    ZR_OUT_DENOISE_MOMENTS = 48    /* RG32F    8 B: accumulated first / second luminance moments */
} zr_output;

/* G-buffer planes (reference GBufferData::GBUFFER order and DXGI formats, DefaultRendererImpl.h:82-109) */
typedef enum zr_gbuffer_plane {
    ZR_GB_BASE_COLOR = 0,   /* R8G8B8A8_UNORM   4 B  (a = subsurface)                        */
    ZR_GB_NORMAL,           /* R16G16_UNORM     4 B  octahedral                              */
    ZR_GB_METALLIC_ROUGHNESS,/* R8G8_UNORM      2 B  x = flag byte, y = roughness            */
    ZR_GB_MOTION_VECTOR,    /* R16G16_SNORM     4 B                                          */
    ZR_GB_EMISSIVE_COLOR,   /* R11G11B10_FLOAT  4 B  written only when emissive              */
    ZR_GB_IOR,              /* R8_UNORM         1 B  written only when transmissive          */
    ZR_GB_COAT,             /* R16G16B16A16_UINT 8 B written only when coated                */
    ZR_GB_DEPTH,            /* R32_FLOAT        4 B  view z (or t with DoF); miss = FLT_MAX  */
    ZR_GB_TRI_DIFF_GEO_A,   /* R32G32B32A32_UINT 16 B                                        */
    ZR_GB_TRI_DIFF_GEO_B,   /* R32G32_UINT      8 B                                          */
    ZR_GB_COUNT
} zr_gbuffer_plane;

/* bytes per pixel of each plane */
static const uint32_t ZR_GB_PLANE_BYTES[ZR_GB_COUNT] = {4, 4, 2, 4, 4, 1, 8, 4, 16, 8};

/* host view of one frame's G-buffer (oracle input/output and zr_gbuffer_download target); row-major, tightly packed */
typedef struct zr_gbuffer_planes {
    uint32_t width, height;
    void*    plane[ZR_GB_COUNT];
} zr_gbuffer_planes;

/* per-frame ray counters (unit of work = one BVH query) */
typedef struct zr_counters {
    uint64_t n_closest;
    uint64_t n_shadow;
} zr_counters;

/* ---- library ---- */
int         zr_abi_version(void);
const char* zr_last_error(void);
int         zr_device_count(int* count);
/* What the device delivers right now, measured on it (round 6; no reference counterpart -- the reference reads adapter properties through DXGI,
   Source/ZetaCore/Core/Device.cpp, and never times the adapter): static properties + three probes of >= min_ms milliseconds each (<= 0: 50 ms):
   a device-to-device copy (HBM side, read + written bytes per second), an fp32 FMA issue-rate kernel that fills every SIMD (VALU side), and the shader
   clock the probe's waves actually ran at (shader-clock counter over the constant-rate wall clock).  bench.py prints it as `device_state` before its
   timed region, so that a frame time can be told apart from the box it was measured on.  Waits for the device; ~0.2 s. */
typedef struct zr_device_probe {
    char     name[64], arch[32];
    uint32_t compute_units, clock_khz_max, mem_clock_khz_max, mem_bus_bits, l2_bytes, wall_clock_khz;
    uint64_t hbm_bytes;
    float    copy_GBs, copy_ms;              /* device-to-device copy: (read + written) GB/s, device time spent */
    float    fma_tflops, fma_ms;             /* fp32 FMA rate over all CUs (2 flops per FMA), device time spent */
    float    sclk_mhz_under_load;            /* shader clock during the FMA probe */
} zr_device_probe;
int         zr_device_probe_run(int device, float min_ms, zr_device_probe* out);
/* Self-description of the wire formats of zr_wire.h for binding generators and layout checks: one line per struct ("Name size") and per
   field ("Name.Field offset size"), spelled with the REFERENCE's struct / field names (Vertex.h, RtCommon.h, Material.h, FrameConstants.h).
   The test-suite compares this text with offsetof() of the reference's own headers compiled in place (oracle/_ref, tests/test_ref_pins.py).
   Returns the number of bytes written (excluding the terminating 0), or -(required size) when `cap` is too small. */
int         zr_wire_layout(char* buf, size_t cap);

/* ---- scene: device copies of VB/IB/MeshInstance/Material/Emissive/AliasTable + BVH ---- */
int zr_scene_create(int device, const zr_scene_desc* desc, zr_scene** out);
int zr_scene_destroy(zr_scene* scene);
/* Per-frame update of dynamic instances (TLAS::FillMeshInstanceData + the TLAS rebuild, RtAccelerationStructure.cpp:318-506, 708-787;
 * what the reference publishes as RT_FRAME_MESH_INSTANCES_CURR / _PREV and RT_SCENE_BVH_CURR / _PREV): `instances` = the new
 * MeshInstance records (the caller fills PrevRotation / PrevScale / dTranslation like the reference does), `instance_to_world` = the new
 * exact object-to-world matrices (n x 12 floats).  The instance buffer and acceleration structure of the last frame become the
 * "previous" ones that the CtT replay / reconnect passes of ReSTIR PT and the temporal shifts of the DI passes trace against.
 * Host call between frames (waits for the device); n must equal the scene's instance count.
 * Once a scene has been updated, call this every frame -- also with unchanged records on frames where nothing moves -- like the reference
 * rebuilds its instance buffer every frame (RtAccelerationStructure.cpp:382-506): the "previous" buffers are whatever was current before the
 * LAST call, so skipping static frames would leave the temporal passes with a stale previous scene. */
int zr_scene_update_instances(zr_scene* scene, const zr_mesh_instance* instances, const float* instance_to_world, uint32_t n);
/* Emissive triangles of instances that moved (SceneCore::UpdateEmissivePositions, SceneCore.cpp:913-955, then EmissiveBuffer::UpdateTriPositions'
 * upload of [minIdx, maxIdx)): replaces `count` records of the scene's emissive buffer from index `first`.  The caller re-derives them like the
 * reference's CPU side does -- decode the object-space record, transform, re-encode (zrh_emissive_to_world, zetaray_amd/host/zr_scene_io.h).
 * Like the reference, moving a light does not re-estimate powers or rebuild the alias table (that happens when emissive MATERIALS change,
 * PreLighting.cpp:266); presampled sets and the light voxel grid pick the new positions up on the next PRELIGHTING render.
 * Host call between frames (waits for the device). */
int zr_scene_update_emissives(zr_scene* scene, const zr_emissive_triangle* triangles, uint32_t first, uint32_t count);
/* Stream-ordered forms of the three updates above and of zr_scene_set_alias_table -- what the reference does when it records the instance /
 * TLAS update and the buffer uploads on the frame's command list (RtAccelerationStructure.cpp:708-789, SceneCore.cpp:913-955,
 * PreLighting.cpp:556-575): everything is ENQUEUED on `hip_stream` (host records are copied into a pinned staging ring first, so the
 * caller's arrays may be reused at once) and the call returns without waiting for the device.  Ordering against zr_pass_render is the
 * stream's; renders enqueued on OTHER streams are ordered by events inside the library (a render waits for the last update, an update
 * waits for the last render of every stream that used the scene), so no host synchronisation is needed in either direction.  The plain
 * forms are `_async` on the null stream followed by hipDeviceSynchronize().  (ZR_SCENE_UPDATE=rebuild, the host BVH rebuild, stays
 * host-synchronous by construction.) */
int zr_scene_update_instances_async(zr_scene* scene, void* hip_stream, const zr_mesh_instance* instances, const float* instance_to_world, uint32_t n);
/* Background SAH rebuild for dynamic scenes (the reference rebuilds its TLAS every frame, RtAccelerationStructure.cpp:708-789; this library refits one
   world-space tree, whose topology ages as instances move).  Enabled: an update that finds no build in flight snapshots the new transforms and starts
   the host's SAH builder on a thread; the first update after it has finished uploads the new topology into the buffer set that becomes current and
   refits it to that update's transforms -- stream-ordered, no render waits, results never depend on the tree.  stats: builds started / installed,
   building = 0 idle, 1 building, 2 built and waiting for the next update.  ZR_SCENE_UPDATE=refit_sah turns it on for every scene of the process.
   Threading: zr_scene_set_background_rebuild and the update calls of ONE scene must come from one thread at a time (the pass-level contract: a scene
   has one owner); zr_scene_background_rebuild_stats may be called from any thread.  Stream capture: the FIRST in-place update of a scene waits for
   the device once (it cannot know which earlier renders read the buffers it is about to rewrite), so it must not be recorded into a hipGraph
   capture; later updates only enqueue. */
int zr_scene_set_background_rebuild(zr_scene* scene, int enable);
int zr_scene_background_rebuild_stats(zr_scene* scene, uint64_t* started, uint64_t* installed, int* building);
int zr_scene_update_emissives_async(zr_scene* scene, void* hip_stream, const zr_emissive_triangle* triangles, uint32_t first, uint32_t count);
int zr_scene_update_materials_async(zr_scene* scene, void* hip_stream, const zr_material* materials, uint32_t first, uint32_t count);
int zr_scene_set_alias_table_async(zr_scene* scene, void* hip_stream, const zr_alias_entry* entries, uint32_t n);
/* Emissive MATERIALS changed (SceneCore::UpdateEmissiveMaterial: factor / strength rewritten in the records handed to zr_scene_update_emissives;
 * Scene::AreEmissiveMaterialsStale, PreLighting.cpp:266): drops the alias table, so that the next ZR_PASS_PRELIGHTING render re-estimates the
 * triangle powers (K2) and rebuilds it. */
int zr_scene_invalidate_alias_table(zr_scene* scene);
/* ... or, like the reference's steady state (PreLighting.cpp:527-540: the table is rebuilt when the read-back fence has passed, the old one is sampled
 * until then): PRELIGHTING renders only ENQUEUE -- K2 + an asynchronous read-back on the first, the host build + upload once the read-back has
 * landed (checked without waiting) -- so the new table takes effect a frame or two later and no render call blocks. */
int zr_scene_invalidate_alias_table_deferred(zr_scene* scene);
/* Material edits (SceneCore::UpdateMaterial: the material buffer entry is rewritten and re-uploaded): replaces `count` records of the scene's
 * material buffer from index `first`.  Texture indices must stay inside the scene's texture heap.  Host call between frames (waits for the device). */
int zr_scene_update_materials(zr_scene* scene, const zr_material* materials, uint32_t first, uint32_t count);
/* EmissiveTriangleAliasTable::Render (PreLighting.cpp:512-585): upload a host-built table ... */
int zr_scene_set_alias_table(zr_scene* scene, const zr_alias_entry* entries, uint32_t n);
/* ... or build it from per-triangle power exactly like PreLighting.cpp:27-158 (host side, bit-exact) */
int zr_alias_table_build(const float* power, uint32_t n, uint32_t align_phase, zr_alias_entry* out_entries);
int zr_scene_get_alias_table(const zr_scene* scene, zr_alias_entry* out_entries, uint32_t n);
/* K4 output (PreLighting::GetLightVoxelGrid, GlobalResource::LIGHT_VOXEL_GRID): dim.x * dim.y * dim.z * 64 samples, voxel-major */
int zr_scene_get_light_voxel_grid(const zr_scene* scene, void* hip_stream, zr_voxel_sample* out_samples, uint32_t n);
/* K3 output (GlobalResource::PRESAMPLED_EMISSIVE_SETS, PreLighting.cpp:300-315): num_sample_sets * sample_set_size records of the last PRELIGHTING render */
int zr_scene_get_presampled_sets(const zr_scene* scene, void* hip_stream, zr_presampled_tri* out_samples, uint32_t n);
/* BVH introspection for tests / the CPU baseline */
int zr_scene_bvh_info(const zr_scene* scene, uint32_t* num_nodes, uint32_t* num_tris, uint32_t* max_depth);

/* ---- G-buffer ---- */
int zr_gbuffer_create(int device, uint32_t width, uint32_t height, zr_gbuffer** out);
int zr_gbuffer_destroy(zr_gbuffer* gb);
/* Multi-GPU screen-tile split: this G-buffer (and the passes rendering into it) covers the
 * tile whose top-left pixel is (x0, y0) of the full render target described by cbFrameConstants.RenderWidth/Height.
 * Origins are 32-pixel aligned so thread groups, RNG group ids and sort tiles coincide with the single-GPU run. */
int zr_gbuffer_set_tile_origin(zr_gbuffer* gb, uint32_t x0, uint32_t y0);
int zr_gbuffer_download(const zr_gbuffer* gb, void* hip_stream, zr_gbuffer_planes* host_planes);
int zr_gbuffer_device_plane(const zr_gbuffer* gb, int plane, void** dev_ptr);

/* ---- passes ---- */
int zr_params_default(zr_params* p);
int zr_pass_create(int kind, int device, zr_pass** out);
int zr_pass_init(zr_pass* pass, uint32_t width, uint32_t height, int integrator);
int zr_pass_resize(zr_pass* pass, uint32_t width, uint32_t height);   /* OnWindowResized + ResetTemporal */
int zr_pass_reset_temporal(zr_pass* pass);
/* GBUFFER pass: GBufferRT::PickPixel / ClearPick / GetPickReadbackBuffer (GBuffer/GBufferRT.h:36-46).  While a pick is pending every GBUFFER render writes
   the mesh index (GeometryIndex + InstanceID: the index of the instance record) under pixel (x, y) of the render target, UINT32_MAX when the primary ray
   misses (GBufferRT_Inline.hlsl:241-242); on a screen tile only the pass whose tile holds the pixel writes.  zr_pass_read_pick copies the value back on
   `stream` -- the stream the G-buffer was rendered on, or one ordered behind it -- and waits for it; ZR_ERR_NOT_INITIALIZED when no render has covered the pixel since zr_pass_pick_pixel. */
int zr_pass_pick_pixel(zr_pass* pass, uint32_t x, uint32_t y);
int zr_pass_clear_pick(zr_pass* pass);
int zr_pass_read_pick(zr_pass* pass, void* stream, uint32_t* mesh_idx);
int zr_pass_set_params(zr_pass* pass, const zr_params* params);
/* Render(CommandList&): enqueue only.  cb = the 544-byte cbFrameConstants of this frame (host pointer, copied). */
int zr_pass_render(zr_pass* pass, void* hip_stream, const zr_frame_constants* cb, const zr_scene* scene,
                   zr_gbuffer* gbuffer);
/* ---- multi-GPU screen-tile split of a pass with cross-pixel reuse ----
   The G-buffer (and so every plane of the pass) covers this device's tile plus an apron; `owned` is the part this device
   shades.  Apron G-buffer pixels are rendered locally (geometry is replicated); apron reservoirs arrive through
   zr_pass_halo_unpack.  Frame order on every device:
     GBUFFER render -> INDIRECT stage TEMPORAL -> exchange ZR_HALO_POST_TEMPORAL -> INDIRECT stage SPATIAL ->
     exchange ZR_HALO_FINAL (only needed when reprojection can cross tiles, i.e. a moving camera).
   Reference: the two stages are IndirectLighting::ReSTIR_PT_Temporal / ReSTIR_PT_Spatial (IndirectLighting.cpp:370-596, 598-875),
   a renderer registers them as two graph nodes with the exchange between them. */
#define ZR_STAGE_TEMPORAL 1
#define ZR_STAGE_SPATIAL  2
#define ZR_STAGE_ALL      3
/* ReSTIR PT with zr_params.num_spatial_passes = 2 on tiles: the second search / sort / replay / reconnect round as its own stage, with one more exchange of
   ZR_HALO_POST_TEMPORAL (the set the next stage reads) before it; zr_pass_render runs both rounds.  A no-op for every other pass / setting. */
#define ZR_STAGE_SPATIAL2 4
/* ReSTIR PT: the TEMPORAL stage in its two halves (ZR_STAGE_TEMPORAL == both; other passes ignore the bits).
     ZR_STAGE_CANDIDATES      K11 alone: this frame's initial candidates (ReSTIR_PT_PathTrace.hlsl).  Reads only this frame's G-buffer and the scene;
                              writes the "current" reservoir set and the target plane.
     ZR_STAGE_TEMPORAL_REUSE  K12 - K14: Sort_TtC / Sort_CtT, the replays, the temporal reconnection (IndirectLighting.cpp:383-596).
   They are what a renderer puts on two queues to software-pipeline consecutive frames (zr_pass_set_frame_overlap below), the way the reference
   overlaps work between its direct and its async-compute queue (Source/ZetaCore/Core/RenderGraph.cpp:442-541). */
#define ZR_STAGE_CANDIDATES     8
#define ZR_STAGE_TEMPORAL_REUSE 16
/* ZR_PASS_DENOISE only: the steps of the pass one by one (zr_pass_render_stage; ZR_STAGE_SPATIAL / ZR_STAGE_ALL = all of them).  A device of the tile
   split runs them in groups with a halo exchange wherever the next step's stencil would reach beyond what is still exact in its 32-px apron
   (reach: variance 3 px, a-trous iteration i 2 * 2^i px; zetaray_amd/tiling.py denoise_schedule): exchange ZR_HALO_DENOISE_INPUT, TEMPORAL + VARIANCE +
   ATROUS(0..2), exchange ZR_HALO_DENOISE_ITER, ATROUS(3), exchange ZR_HALO_DENOISE_ITER, ATROUS(4) for the default five iterations. */
#define ZR_STAGE_DENOISE_TEMPORAL  (1 << 8)
#define ZR_STAGE_DENOISE_VARIANCE  (1 << 9)
#define ZR_STAGE_DENOISE_ATROUS(i) (1u << (10 + (i)))      /* i = 0 .. svgf_iterations - 1 (<= 8) */
#define ZR_STAGE_DENOISE_MASK      0x3ff00
#define ZR_HALO_POST_TEMPORAL 0   /* valid between the two stages of a frame */
#define ZR_HALO_FINAL         1   /* valid after the frame: the set the next frame reads as "previous" */
/* ZR_PASS_DENOISE: what a tile needs from its neighbours before the temporal step -- this frame's input signal (the RGBA32F plane bound as
   ZR_IN_DENOISE_SIGNAL: the indirect pass shades owned pixels only) + the colour / length history (RGBA32F) + the moment history (RG32F), 40 B per
   pixel -- and between a-trous iterations: the iteration's current colour + variance plane (RGBA32F, 16 B per pixel) */
#define ZR_HALO_DENOISE_INPUT 2
#define ZR_HALO_DENOISE_ITER  3
#define ZR_HALO_BYTES_PER_PIXEL 62  /* planes A..G back to back: 4 + 8 + 16 + 16 + 2 + 8 + 8, each row-major over the rect */
int zr_pass_set_owned_rect(zr_pass* pass, uint32_t x0, uint32_t y0, uint32_t width, uint32_t height);   /* global pixels; width 0 = whole tile */
/* Frame overlap (round 6): K1 + K11 of frame N + 1 beside K15 / K12 / K13 / K16 of frame N.
   K11's candidates depend on nothing but G-buffer(N + 1), so with enable = 1 the ReSTIR PT pass keeps a THIRD reservoir set, a second target plane and
   a second FINAL plane, and its stages may be enqueued on two streams:
       stream A:  GBUFFER(N + 1), PRELIGHTING(N + 1), INDIRECT stage ZR_STAGE_CANDIDATES(N + 1)
       stream B:  INDIRECT stages ZR_STAGE_TEMPORAL_REUSE | ZR_STAGE_SPATIAL [| ZR_STAGE_SPATIAL2](N + 1), then whatever consumes the frame
   in exactly that host order, frame after frame.  The library orders them with events: CANDIDATES(N + 1) waits for the temporal reuse of frame N (the last
   reader of the planes it recycles), TEMPORAL_REUSE(N + 1) waits for CANDIDATES(N + 1); the G-buffer given here is tracked -- a GBUFFER render waits for
   the passes still reading the plane set it is about to overwrite, a pass on another stream waits for the GBUFFER render.  Results do not depend on
   the streams (one stream for everything is the plain order) nor on the switch: tests/test_gpu_parity.py::test_frame_overlap_changes_nothing.
   Outputs: zr_pass_get_output(ZR_OUT_FINAL / ZR_OUT_RPT_*) address the planes of the last frame whose final stage has been enqueued; re-query them every
   frame (ZR_OUT_FINAL alternates between two planes unless the frame accumulates).  Only the ReSTIR PT integrator; call between frames.
   zr_pass_frame_overlap_stream: a non-blocking stream owned by the pass, for callers without streams of their own ("stream A").
   enable = ZR_FRAME_OVERLAP (1): the product mode.  The passes update reservoir records in place and write only the components a record's case uses
   (Reservoir.hlsli:283-456), so the UNUSED bytes of a record -- never read by any pass -- are whatever the set held before: with three sets in
   rotation that is another frame's leftovers than in the plain order.  Every used byte, FINAL and the ray counters are identical.
   enable = ZR_FRAME_OVERLAP_CARRY (2): additionally copies the replaced set and target plane into the set that takes their place before K11 (one
   streaming kernel, 156 B per pixel): every byte of every plane then equals the plain order's -- what the parity tests compare with the oracle. */
#define ZR_FRAME_OVERLAP       1
#define ZR_FRAME_OVERLAP_CARRY 2
int zr_pass_set_frame_overlap(zr_pass* pass, zr_gbuffer* gbuffer, int enable);
int zr_pass_frame_overlap_stream(zr_pass* pass, void** hip_stream);
/* hipDeviceSynchronize for callers that hold no HIP runtime of their own (bindings through ctypes / cgo) */
int zr_device_synchronize(int device);
/* The same protocol for the other passes with cross-pixel reuse (SURVEY 8(e) "Collective"): bytes per pixel of a halo transfer =
   the pass's reservoir planes back to back: ReSTIR PT 62, ReSTIR GI 40 (A, B, C), ReSTIR DI emissive 24 (A, B), sun + sky DI 13.
   DI passes: TEMPORAL stage -> exchange ZR_HALO_POST_TEMPORAL -> SPATIAL stage (the exchanged set is also the one the next
   frame reprojects into, so no ZR_HALO_FINAL is needed).  ReSTIR GI renders in the TEMPORAL stage and needs ZR_HALO_FINAL only. */
int zr_pass_halo_bytes_per_pixel(zr_pass* pass, uint32_t* bytes);
int zr_pass_render_stage(zr_pass* pass, void* hip_stream, const zr_frame_constants* cb, const zr_scene* scene,
                         zr_gbuffer* gbuffer, int stages);
int zr_pass_halo_pack(zr_pass* pass, void* hip_stream, const zr_gbuffer* gbuffer, int which, uint32_t x0, uint32_t y0,
                      uint32_t width, uint32_t height, void* dev_dst, size_t bytes);
int zr_pass_halo_unpack(zr_pass* pass, void* hip_stream, const zr_gbuffer* gbuffer, int which, uint32_t x0, uint32_t y0,
                        uint32_t width, uint32_t height, const void* dev_src, size_t bytes);
/* Fused variant for the multi-device path: ONE kernel moves every reservoir plane of every rect between the planes and one device buffer
   (send or receive buffer of all peers).  Rect i occupies [offset, offset + w * h * bytes_per_pixel) of the buffer, laid out like the block
   zr_pass_halo_pack writes (planes back to back); offsets must be 16-byte aligned and w * h a multiple of 8. */
#define ZR_HALO_MAX_RECTS 16
typedef struct zr_halo_rect { uint32_t x0, y0, w, h; uint64_t offset; } zr_halo_rect;
int zr_pass_halo_pack_all(zr_pass* pass, void* hip_stream, const zr_gbuffer* gbuffer, int which, const zr_halo_rect* rects, uint32_t num_rects,
                          void* dev_buf, size_t bytes);
int zr_pass_halo_unpack_all(zr_pass* pass, void* hip_stream, const zr_gbuffer* gbuffer, int which, const zr_halo_rect* rects, uint32_t num_rects,
                            const void* dev_buf, size_t bytes);
int zr_pass_get_output(const zr_pass* pass, int which, void** dev_ptr, uint32_t* width, uint32_t* height,
                       uint32_t* bytes_per_pixel);
int zr_pass_download_output(const zr_pass* pass, int which, void* hip_stream, void* host_dst, size_t bytes);
/* Compositing inputs (cbCompositing::*DescHeapIdx + CB_COMPOSIT_FLAGS, RP/Compositing/Compositing_Common.h:12-37): device pointers
   to RGBA32F planes of the pass size, or NULL to leave a term out.  Output: zr_pass_get_output(ZR_OUT_FINAL). */
#define ZR_IN_EMISSIVE_DI 0
#define ZR_IN_INDIRECT    1
#define ZR_IN_SKY_DI      2
#define ZR_IN_TAA_SIGNAL  3   /* ZR_PASS_TAA: the RGBA32F image to anti-alias (TAA::SHADER_IN_RES::SIGNAL) */
/* ZR_PASS_AUTO_EXPOSURE / ZR_PASS_DISPLAY: the image to meter / display (AutoExposure::SHADER_IN_DESC::COMPOSITED,
   DisplayPass::SetInput): either an RGBA16F plane (ZR_OUT_TAA) or an RGBA32F plane (ZR_OUT_FINAL of COMPOSITING; read rounded to half,
   which is what the reference's R16G16B16A16_FLOAT composited texture holds).  Binding one clears the other. */
#define ZR_IN_POST_SIGNAL_F16  4
#define ZR_IN_POST_SIGNAL_F32  5
#define ZR_IN_DISPLAY_EXPOSURE 6   /* ZR_PASS_DISPLAY: RG32F 1 x 1 (ZR_OUT_EXPOSURE) */
#define ZR_IN_DENOISE_SIGNAL   7   /* ZR_PASS_DENOISE: the RGBA32F image to filter */
int zr_pass_set_input(zr_pass* pass, int which, const void* dev_plane);
/* ZR_PASS_DISPLAY: the Tony McMapface LUT of the NEUTRAL tone mapper, dim^3 R9G9B9E5_SHAREDEXP texels on the host (the payload of
   Assets/LUT/tony_mc_mapface.dds, 48^3; shipped as zetaray_amd/assets/tony_mc_mapface_rgb9e5.bin).  Display.cpp:196-205. */
int zr_pass_set_tonemap_lut(zr_pass* pass, const uint32_t* rgb9e5, uint32_t dim);
/* ray counters accumulated since the last call (device -> host copy; synchronises the stream) */
int zr_pass_read_counters(zr_pass* pass, void* hip_stream, zr_counters* out, int reset);
/* ReSTIR PT: GPU time per 32 x 32-pixel cell of the pass's planes ((w + 31) / 32 + 1 by (h + 31) / 32 + 1 cells, cell (0, 0) at the plane origin): the summed
 * lifetimes, in units of 16 shader cycles, of the waves of K11 / K14 / K16 that worked on the cell since the last reset: the load signal of the cost-balanced screen split over N devices (tiling.balanced_layout).  No
 * reference counterpart (the reference renders on one GPU).  Costs one atomic per wave while enabled.
 * enable = ZR_COST_MAP_RAYS (2): the cells count the BVH queries issued for their pixels by every kernel of the pass instead (a diagnostic: the
 * per-window ray counts of the at-size parity tests, tests/test_gpu_parity.py::test_atrium_*_windows_*). */
#define ZR_COST_MAP_TIME 1
#define ZR_COST_MAP_RAYS 2
int zr_pass_enable_cost_map(zr_pass* pass, int enable);
int zr_pass_read_cost_map(zr_pass* pass, void* hip_stream, uint32_t* out_cells, uint32_t cells_w, uint32_t cells_h, int reset);
/* diagnostic, ReSTIR PT with ZR_K11=trip in the environment: K11 counts the lanes alive at its bounce boundaries -- what compaction between
 * bounces could win back.  out = {lanes alive at the boundaries, lane slots of the waves that passed them, 32-bit words of path state carried across a
 * boundary}, accumulated since the pass was created.  Waits for the device.  No reference counterpart. */
int zr_pass_debug_trip_stats(zr_pass* pass, uint64_t out[3]);
/* test hook: the node count from which the ReSTIR PT pass launches K11's node-cached instantiation (4 waves per SIMD + the top of the tree in LDS).  Default 1: every
 * scene with a tree (measured faster at every size in round 6; rounds 3 - 5: 16384 nodes = 1 MB) -- a huge count (0x7FFFFFFF) lets the parity tests run the other instantiation.
 * 0 restores the default.  Process-wide. */
int zr_debug_set_large_scene_nodes(uint32_t num_nodes);
/* test hook: the deepest tree the DEVICE builder (ZR_BVH_BUILD=device, ZR_SCENE_UPDATE=rebuild) may produce, in levels.  The default, 21, is what a
 * lane's traversal stack holds (3 entries per level of 64); a node whose key range could not fit below the cap if cut unevenly is cut into four
 * equal parts instead (zr_tu_bvh.hip).  2 .. 21, anything else restores the default.  Process-wide; no reference counterpart. */
int zr_debug_set_bvh_depth_cap(uint32_t levels);
/* test hook: 0 = scenes of the plain material class (every material an opaque, uncoated, non-metallic dielectric; no texture heap) render with the
 * general kernels like every other scene; 1 (default) = they render with the PLAIN permutations of the ReSTIR PT kernels, which have no code for the
 * other lobes.  Results are identical either way (tests/test_gpu_parity.py::test_material_class_kernels_change_nothing).  Process-wide. */
int zr_debug_set_material_class_kernels(int enable);
/* the material class zr_scene_create / zr_scene_update_materials currently derive from the scene's material table: 1 = plain, 0 = general.
 * The lighting passes render a plain scene with the PLAIN kernel permutations when the plane sets of the G-buffer they read were rendered by the GBUFFER
 * pass FROM THAT SCENE while it was plain (the current set; the previous one too from the second render on).  A G-buffer no GBUFFER pass has rendered --
 * planes an engine filled itself through zr_gbuffer_device_plane -- or one rendered from another scene runs the general kernels: its flags (metallic,
 * transmissive, coated, ...) are not the library's to vouch for.  (An engine that OVERWRITES planes of a rendered set must keep them consistent with
 * the material table, as the halo exchange of the tile split does: it moves planes the GBUFFER pass of a neighbouring device rendered from the same scene.) */
int zr_scene_material_class(const zr_scene* scene, uint32_t* out_class);
/* the same counters split by the kernel that issued the queries (not reset; roofline bookkeeping of bench.py) */
int zr_pass_read_kernel_counters(zr_pass* pass, void* hip_stream, uint32_t max_entries, const char** names,
                                 uint64_t* n_closest, uint64_t* n_shadow, uint32_t* count);
/* Device self-test of the arithmetic contract's half conversions: runs the instruction path the kernels use and the portable code of
   zr_detmath.h side by side on the device over all 2^32 fp32 and all 2^16 fp16 bit patterns; returns the number of disagreeing
   patterns (both must be 0).  The second count also covers zr_div255 / zr_div65535 (UNORM decode) against the IEEE division for
   all 65536 integer inputs. */
int zr_selftest_half_conversions(int device, uint64_t* mismatches_f32_to_f16, uint64_t* mismatches_f16_to_f32);

/* GpuTimer analogue (Source/ZetaCore/Core/GpuTimer.h:28-45): per-kernel hipEvent timing of the last render */
int zr_pass_enable_timing(zr_pass* pass, int enable);
int zr_pass_get_timings(zr_pass* pass, uint32_t max_entries, const char** names, float* ms, uint32_t* launches,
                        uint32_t* count);
int zr_pass_destroy(zr_pass* pass);

/* ---- ray-query microbenchmark surface (bench.py roofline leg / parity tests of traversal) ---- */
/* rays: n x 8 floats (ox,oy,oz,tmin,dx,dy,dz,tmax) device pointer; hits: n x 4 uint32 (t bits, u bits, v bits, tri) */
int zr_trace_closest(const zr_scene* scene, void* hip_stream, const float* d_rays, uint32_t n, uint32_t mask,
                     uint32_t* d_hits);
int zr_trace_any(const zr_scene* scene, void* hip_stream, const float* d_rays, uint32_t n, uint32_t mask,
                 uint32_t* d_occluded);

#ifdef __cplusplus
}
#endif

#endif /* ZETARAY_AMD_H */
