// zr_host.h -- C++ host-side mirror of the reference's RenderPass / RenderGraph node API over the C-ABI.
//
// Keeps the surface a ZetaRay-style renderer (Source/ZetaRenderer/Default/PathTracer.cpp:287-294, 474-552) programs
// against, re-implemented over HIP streams and events instead of D3D12 command lists and resource barriers:
//
//   reference                                                     here
//   ------------------------------------------------------------  ---------------------------------------------------
//   Core::CommandList (ZC/Core/CommandList.h)                     Core::CommandList  = one hipStream_t
//   fastdelegate::FastDelegate1<CommandList&> + MakeDelegate      Core::Delegate1<CommandList&> + MakeDelegate
//   Core::RenderGraph (ZC/Core/RenderGraph.h:56-118)              Core::RenderGraph  (same method names/order of use)
//     BeginFrame / RegisterRenderPass / RegisterResource /          resource states -> producer/consumer edges;
//     MoveToPostRegister / AddInput / AddOutput / Build(TaskSet&)   barriers -> hipEventRecord / hipStreamWaitEvent
//   Support::TaskSet (ZC/Support/Task.h:89-149)                   Support::TaskSet   (<= 16 tasks, edges, Run())
//   RenderPass::GBufferRT / PreLighting / IndirectLighting        RenderPass::* : Init / OnWindowResized / ResetTemporal /
//     (RP/GBuffer/GBufferRT.h:27-47, RP/PreLighting/PreLighting.h:28-58,  Set* / GetOutput / Render(CommandList&)
//      RP/IndirectLighting/IndirectLighting.h:72-108)
//   Check(expr, fmt...) aborts (ZC/Utility/Error.h:68-81)         ZR_CHECK: nonzero zr_status -> message + abort
//
// Threading contract is the reference's: Build() runs one task per render node (nodes of one dependency level may
// record concurrently from worker threads, each into its own stream); a pass object is never entered re-entrantly.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>
#include "../../include/zetaray_amd.h"

#define ZR_CHECK(expr) do { int zr_rc_ = (expr); if (zr_rc_ != 0) { std::fprintf(stderr, "Check failed: %s -> %d: %s (%s:%d)\n", #expr, zr_rc_, zr_last_error(), __FILE__, __LINE__); std::abort(); } } while (0)

// ---- multi-device transport (zr_halo.cpp), C linkage so that bench.py / tests can drive it through ctypes
extern "C" {
typedef struct zrh_comm zrh_comm;                     // one RCCL communicator (one rank of the tile split)
typedef struct zrh_halo_exchange zrh_halo_exchange;   // send / receive buffers + the rect lists of one pass's exchange
typedef struct zrh_halo_peer { int peer; uint32_t send_x0, send_y0, send_w, send_h, recv_x0, recv_y0, recv_w, recv_h; } zrh_halo_peer;
const char* zrh_halo_last_error(void);
int zrh_rccl_unique_id(uint8_t* out128);
int zrh_comm_create(int device, int world, int rank, const uint8_t* id128, zrh_comm** out);
void zrh_comm_destroy(zrh_comm* c);
int zrh_halo_exchange_create(zr_pass* pass, zr_gbuffer* gb, zrh_comm* comm, const zrh_halo_peer* peers, uint32_t n, zrh_halo_exchange** out);
/* the same with an explicit size per pixel (0 = the pass's own): ZR_PASS_DENOISE moves 40 B (ZR_HALO_DENOISE_INPUT) or 16 B (ZR_HALO_DENOISE_ITER) */
int zrh_halo_exchange_create_bpp(zr_pass* pass, zr_gbuffer* gb, zrh_comm* comm, const zrh_halo_peer* peers, uint32_t n, uint32_t bytes_per_pixel, zrh_halo_exchange** out);
void zrh_halo_exchange_destroy(zrh_halo_exchange* x);
size_t zrh_halo_exchange_send_bytes(const zrh_halo_exchange* x);
int zrh_halo_exchange_run(zrh_halo_exchange* x, void* hip_stream, int which);
// ---- scene ingestion + per-frame maintenance on the host (zr_scene_io.h) handed to the device scene
struct zrh_scene_data;
// Model::glTF::Load + SceneCore + TLAS build in one call; tex_offsets4 = the four descriptor-table offsets for cbFrameConstants
int zrh_scene_create_from_gltf(int device, const char* path, const uint16_t* rho_lut, const uint32_t* rho_dim3, zr_scene** out, uint32_t* tex_offsets4);
// per frame: upload what zrh_scene_data_begin_frame / zrh_scene_data_set_instance_world changed -- the moved lights' records
// (zr_scene_update_emissives), then the instance buffer + matrices (zr_scene_update_instances)
int zrh_scene_apply_updates(zr_scene* scene, const struct zrh_scene_data* data);                       /* enqueued on the null stream, no host wait */
int zrh_scene_apply_updates_on(zr_scene* scene, const struct zrh_scene_data* data, void* hip_stream); /* enqueued on `hip_stream` */
}

namespace ZetaRayAMD {

namespace Support {
    // Minimal stand-in for ZetaRay's TaskSet: tasks + edges, executed level by level (tasks of one level concurrently).
    struct TaskSet
    {
        static constexpr int MAX_NUM_TASKS = 16;
        using TaskHandle = int;
        TaskHandle EmplaceTask(const char* name, std::function<void()> f);
        void AddOutgoingEdge(TaskHandle a, TaskHandle b);
        void Sort();
        void Finalize() {}
        void Run(bool parallel = true);          // App::Submit(ZetaMove(ts)) + wait
        int GetSize() const { return (int)m_tasks.size(); }
        struct Task { std::string name; std::function<void()> fn; std::vector<int> out; int indeg = 0; int level = 0; };
        std::vector<Task> m_tasks;
        std::vector<std::vector<int>> m_levels;
    };
}

namespace Core {
    class CommandList
    {
    public:
        explicit CommandList(void* hipStream = nullptr) : m_stream(hipStream) {}
        void* Stream() const { return m_stream; }
    private:
        void* m_stream;
    };
    using ComputeCmdList = CommandList;

    // FastDelegate1<CommandList&>-shaped callable
    template<typename Arg> struct Delegate1
    {
        std::function<void(Arg)> fn;
        void operator()(Arg a) const { fn(a); }
        explicit operator bool() const { return (bool)fn; }
    };
    template<typename T, typename Arg>
    Delegate1<Arg> MakeDelegate(T* obj, void (T::*method)(Arg)) { Delegate1<Arg> d; d.fn = [obj, method](Arg a) { (obj->*method)(a); }; return d; }

    enum class RENDER_NODE_TYPE : uint8_t { RENDER, COMPUTE, ASYNC_COMPUTE };
    // stand-ins for D3D12_RESOURCE_STATES: only read vs write matters for ordering
    enum RESOURCE_STATE : uint32_t { STATE_COMMON = 0, STATE_SHADER_READ = 1, STATE_UNORDERED_ACCESS = 2 };

    struct RenderNodeHandle
    {
        static constexpr int INVALID_HANDLE = -1;
        RenderNodeHandle() = default;
        explicit RenderNodeHandle(int u) : Val(u) {}
        bool IsValid() const { return Val != INVALID_HANDLE; }
        int Val = INVALID_HANDLE;
    };

    class RenderGraph
    {
    public:
        static constexpr int MAX_NUM_RENDER_PASSES = 32;
        static constexpr int MAX_NUM_RESOURCES = 64;
        RenderGraph();
        ~RenderGraph();
        RenderGraph(const RenderGraph&) = delete;
        RenderGraph& operator=(const RenderGraph&) = delete;

        void Reset();
        void BeginFrame();
        RenderNodeHandle RegisterRenderPass(const char* name, RENDER_NODE_TYPE t, Delegate1<CommandList&> dlg, bool forceSeparateCmdList = false);
        void RegisterResource(const void* res, uint64_t path, uint32_t initState = STATE_COMMON, bool isWindowSizeDependent = true);
        void RemoveResource(uint64_t path);
        void MoveToPostRegister();
        void AddInput(RenderNodeHandle h, uint64_t path, uint32_t expectedState);
        void AddOutput(RenderNodeHandle h, uint64_t path, uint32_t expectedState);
        // Builds the DAG from the declared producers/consumers, orders nodes by longest path, turns cross-stream
        // dependencies into events, and emits one task per node into `ts`.
        void Build(Support::TaskSet& ts);
        // Host wait for everything submitted by the last Build (the reference's frame completion fence).
        void WaitForFrame();
        // introspection for tests: execution order (node names by batch) of the last Build
        const std::vector<std::vector<std::string>>& Batches() const { return m_batchNames; }
        void* GraphicsStream() const { return m_streams[0]; }
        void* AsyncComputeStream() const { return m_streams[1]; }
        // without a HIP device the graph still builds/orders/executes delegates (streams are null): used by CPU tests
        bool HasDevice() const { return m_hasDevice; }

    private:
        struct Node
        {
            std::string name; RENDER_NODE_TYPE type; Delegate1<CommandList&> dlg;
            std::vector<uint64_t> inputs, outputs; std::vector<int> deps; int batch = 0; void* doneEvent = nullptr;
        };
        std::vector<Node> m_nodes;
        std::vector<uint64_t> m_resources;
        bool m_inPostRegister = false;
        void* m_streams[2] = {nullptr, nullptr};
        bool m_hasDevice = false;
        std::vector<void*> m_eventPool;
        size_t m_eventsUsed = 0;
        std::vector<std::vector<std::string>> m_batchNames;
        void* AcquireEvent();
    };
}

namespace RenderPass {
    // Shared scene + G-buffer objects the passes fetch "by name" in the reference (SharedShaderResources); here they are
    // explicit members of a small context owned by the renderer.
    struct FrameContext
    {
        zr_scene* scene = nullptr;
        zr_gbuffer* gbuffer = nullptr;
        zr_frame_constants frameConstants{};
        uint32_t renderWidth = 0, renderHeight = 0;
        int device = 0;
    };

    struct RenderPassBase
    {
        bool IsInitialized() const { return m_pass != nullptr && m_initialized; }
        void Reset(bool waitForGPU);
        RenderPassBase() = default;
        RenderPassBase(RenderPassBase&&) = delete;
        RenderPassBase& operator=(RenderPassBase&&) = delete;
        ~RenderPassBase();
    protected:
        void InitRenderPass(int kind, FrameContext* ctx, int integrator);
        zr_pass* m_pass = nullptr;
        FrameContext* m_ctx = nullptr;
        bool m_initialized = false;
        int m_integrator = 0;
    };

    struct GBufferRT final : public RenderPassBase
    {
        void Init(FrameContext* ctx);
        void OnWindowResized();
        // GBufferRT.h:36-46: the next Render()s write the mesh index under the pixel into the pick buffer until ClearPick()
        void PickPixel(uint16_t pixelX, uint16_t pixelY);
        bool HasPendingPick() const { return m_pickPending; }
        void ClearPick();
        // the reference hands out its read-back buffer (GetPickReadbackBuffer) and the scene maps it a frame later; here: the value itself, after
        // waiting for `stream` (UINT32_MAX = the primary ray missed)
        uint32_t ReadPick(void* stream = nullptr) const;
        void Render(Core::CommandList& cmdList);
    private:
        bool m_pickPending = false;
    };

    struct PreLighting final : public RenderPassBase
    {
        void Init(FrameContext* ctx);
        void OnWindowResized() {}
        // PreLighting.h:36-52: presampling switches on once the scene holds at least `minToEnable` emissive triangles
        void SetLightPresamplingParams(int minToEnable, int numSampleSets, int sampleSetSize);
        // PreLighting.h:53-58 (dims / extents / y offset of the light voxel grid; needs presampling)
        void SetLightVoxelGridParams(bool enable, uint32_t dimX, uint32_t dimY, uint32_t dimZ, float extX, float extY, float extZ, float offsetY);
        bool IsPresamplingEnabled() const { return m_params.presampling != 0; }
        void Render(Core::CommandList& cmdList);
    private:
        zr_params m_params{};
        int m_minPresample = 0;
        bool m_aliasReady = false;
    };

    // RP/DirectLighting/Emissive/DirectLighting.h:19-60: ReSTIR DI for emissive lights
    struct DirectLighting final : public RenderPassBase
    {
        enum class SHADER_OUT_RES { FINAL, COUNT };
        void Init(FrameContext* ctx);
        void OnWindowResized();
        void ResetTemporal();
        void SetLightPresamplingParams(bool enable, int numSampleSets, int sampleSetSize);
        // The reference's UI parameters of this pass ("Temporal Resample", "Spatial Resample", "M_max", "Extra Sampling (Disocclusion)", "Stochastic
        // Spatial", "Alpha_min (Lobe Selection)": registered in DirectLighting::Init, delivered to DirectLighting.cpp:374-410 *Callback) as plain setters
        void SetTemporalResampling(bool b);
        void SetSpatialResampling(bool b);
        void SetMaxTemporalM(int m);                      // 1..30
        void SetExtraSamplesDisocclusion(bool b);
        void SetStochasticSpatial(bool b);
        void SetAlphaMin(float alphaMin);                 // the constant buffer holds its square (DirectLighting.cpp:404-408)
        void SetHalfVectorCopyShift(bool b);              // USE_HALF_VECTOR_COPY_SHIFT (Emissive/Params.hlsli:12): a compile-time switch in the reference (0 in its tree), a flag here
        void* GetOutput(SHADER_OUT_RES i) const;
        void Render(Core::CommandList& cmdList);
    private:
        void SetFlag(uint32_t bit, bool on);
        zr_params m_params{};
    };

    // RP/Sky/Sky.h:12-112: sky-view LUT (the in-scattering voxel grid is out of scope)
    struct Sky final : public RenderPassBase
    {
        enum class SHADER_OUT_RES { SKY_VIEW_LUT, COUNT };
        void Init(FrameContext* ctx, int lutWidth, int lutHeight);
        void* GetOutput(SHADER_OUT_RES i) const;
        void Render(Core::CommandList& cmdList);
    };

    // RP/DirectLighting/Sky/SkyDI.h:19-137: ReSTIR DI for sun + sky
    struct SkyDI final : public RenderPassBase
    {
        enum class SHADER_OUT_RES { DENOISED, COUNT };       // the reference names its (undenoised) FINAL output DENOISED
        void Init(FrameContext* ctx);
        void OnWindowResized();
        void ResetTemporal();
        // UI parameters (SkyDI.cpp:350-379): "Temporal Resample", "Spatial Resample", "M_max (Sky)", "M_max (Sun)", "Alpha_min (Lobe Selection)"
        void SetTemporalResampling(bool b);
        void SetSpatialResampling(bool b);
        void SetMaxMSky(int m);                           // 1..15
        void SetMaxMSun(int m);                           // 1..15
        void SetAlphaMin(float alphaMin);                 // stored squared
        void* GetOutput(SHADER_OUT_RES i) const;
        void Render(Core::CommandList& cmdList);
    private:
        void SetFlag(uint32_t bit, bool on);
        zr_params m_params{};
    };

    // RP/Compositing/Compositing.h:19-115: (sky DI | emissive DI) + indirect, optional firefly filter
    struct Compositing final : public RenderPassBase
    {
        enum class SHADER_IN_GPU_DESC { SKY_DI, EMISSIVE_DI, INDIRECT, COUNT };
        enum class SHADER_OUT_RES { COMPOSITED, COUNT };
        void Init(FrameContext* ctx);
        void OnWindowResized();
        void SetGpuDescriptor(SHADER_IN_GPU_DESC i, const void* devicePlane);       // RGBA32F FINAL plane of the producing pass
        void SetFireflyFilterEnablement(bool b);
        // "Direct" / "Indirect" toggles of the settings UI (Compositing.cpp:166-179: they clear the CB_COMPOSIT_FLAGS bit of the input; the descriptor stays bound)
        void SetDirectEnablement(bool b);
        void SetIndirectEnablement(bool b);
        void* GetOutput(SHADER_OUT_RES i) const;
        void Render(Core::CommandList& cmdList);
    private:
        void Rebind();
        zr_params m_params{};
        const void* m_desc[3] = {nullptr, nullptr, nullptr};      // by SHADER_IN_GPU_DESC
        bool m_direct = true, m_indirect = true;
    };

    // RP/TAA/TAA.h:20-80: temporal anti-aliasing of the composited image
    struct TAA final : public RenderPassBase
    {
        enum class SHADER_IN_CPU_DESC { SIGNAL, COUNT };
        enum class SHADER_OUT_RES { OUTPUT_A, OUTPUT_B, COUNT };
        void Init(FrameContext* ctx);
        void OnWindowResized();
        void SetCPUDescriptor(SHADER_IN_CPU_DESC i, const void* devicePlane);       // RGBA32F signal
        void SetBlendWeight(float w);                                                // param "BlendWeight", TAA.cpp:150-153
        void ResetTemporal();
        void* GetOutput(SHADER_OUT_RES i) const;      // the library exposes the target written last (both enumerators return it)
        void Render(Core::CommandList& cmdList);
    private:
        zr_params m_params{};
    };

    // Denoise: NO reference counterpart (the reference has no denoiser; BASELINE config 5 asks for one).  Shaped like the reference's other post
    // nodes so that a RenderGraph can schedule it between IndirectLighting and Compositing: input = the indirect pass's radiance, output = the
    // filtered radiance (rgb + variance).  zetaray_amd/csrc/zr_svgf.h defines the filter.
    struct Denoise final : public RenderPassBase
    {
        enum class SHADER_IN_CPU_DESC { SIGNAL, COUNT };
        enum class SHADER_OUT_RES { DENOISED, COUNT };
        void Init(FrameContext* ctx);
        void OnWindowResized();
        void SetCPUDescriptor(SHADER_IN_CPU_DESC i, const void* devicePlane);       // RGBA32F signal
        void SetIterations(uint32_t n);                                              // a-trous iterations, 0..8 (default 5)
        void SetSigmas(float luminance, float depth, uint32_t normalPowerLog2);
        void ResetTemporal();
        void* GetOutput(SHADER_OUT_RES i) const;
        void Render(Core::CommandList& cmdList);
    private:
        zr_params m_params{};
    };

    // RP/AutoExposure/AutoExposure.h:21-100: luminance histogram + adapted exposure of the composited / anti-aliased image
    struct AutoExposure final : public RenderPassBase
    {
        enum class SHADER_IN_DESC { COMPOSITED, COUNT };
        enum class SHADER_OUT_RES { EXPOSURE, COUNT };
        void Init(FrameContext* ctx);
        void OnWindowResized();
        // the image to meter: an RGBA16F plane (TAA output) or an RGBA32F plane (Compositing output)
        void SetDescriptor(SHADER_IN_DESC i, const void* devicePlane, bool rgba16f);
        void SetMinLum(float v); void SetMaxLum(float v); void SetLumMapExp(float v);      // params "Min Lum" / "Max Lum" / "Lum Map Exp", AutoExposure.cpp:64-87
        void* GetOutput(SHADER_OUT_RES i) const;       // RG32F 1 x 1: exposure, adapted luminance
        void Render(Core::CommandList& cmdList);
    private:
        zr_params m_params{};
    };

    // RP/Display/Display.h:35-153: exposure + tone mapping of the final image (DisplayOption::DEFAULT; picking / wireframe overlays are UI)
    struct DisplayPass final : public RenderPassBase
    {
        enum class SHADER_IN_GPU_DESC { COMPOSITED, EXPOSURE, COUNT };
        enum class SHADER_OUT_RES { BACK_BUFFER_LINEAR, BACK_BUFFER_SRGB8, COUNT };       // the reference renders into the swap chain's RTV
        // display size = ctx.frameConstants.display_*; lutRGB9E5 = dim^3 texels of Assets/LUT/tony_mc_mapface.dds (Display.cpp:196-205), may be null
        void Init(FrameContext* ctx, uint32_t displayWidth, uint32_t displayHeight, const uint32_t* lutRGB9E5, uint32_t lutDim);
        void SetGpuDescriptor(SHADER_IN_GPU_DESC i, const void* devicePlane, bool rgba16f = false);
        void SetTonemapper(zr_tonemapper t); void SetAutoExposure(bool b); void SetSaturation(float v); void SetAgXExp(float v);   // Display.cpp:565-619
        void* GetOutput(SHADER_OUT_RES i) const;
        void Render(Core::CommandList& cmdList);
    private:
        zr_params m_params{};
    };

    // Multi-device only (not in the reference): the reservoir halo exchange of the screen-tile split as a graph node between the temporal and
    // spatial stages of a pass (zr_halo.cpp: one pack kernel, grouped ncclSend / ncclRecv over RCCL, one unpack kernel, no host wait)
    struct HaloExchange final
    {
        HaloExchange() = default;
        HaloExchange(const HaloExchange&) = delete;
        ~HaloExchange();
        void Init(FrameContext* ctx, zr_pass* pass, zrh_comm* comm, const zrh_halo_peer* peers, uint32_t numPeers, int which /* ZR_HALO_* */);
        void Render(Core::CommandList& cmdList);
    private:
        FrameContext* m_ctx = nullptr; zrh_halo_exchange* m_x = nullptr; int m_which = 0;
    };

    struct IndirectLighting final : public RenderPassBase
    {
        enum class SHADER_OUT_RES { FINAL, COUNT };
        enum class INTEGRATOR : uint8_t { PATH_TRACING, ReSTIR_GI, ReSTIR_PT, COUNT };
        void Init(FrameContext* ctx, INTEGRATOR method);
        void OnWindowResized();
        void ResetTemporal();
        void SetMethod(INTEGRATOR method);
        void SetLightPresamplingParams(bool enable, int numSampleSets, int sampleSetSize);
        // IndirectLighting.h:83-98 (ReSTIR GI samples lights from the grid on bounces > 0; needs presampling)
        void SetLightVoxelGridParams(bool enabled, uint32_t dimX, uint32_t dimY, uint32_t dimZ, float extX, float extY, float extZ, float offsetY);
        void SetMaxBounces(int nonTr, int glossyTr);
        // The reference's UI parameters of this pass, registered with App::AddParam in SwitchToReSTIR_PT / SwitchToReSTIR_GI / SwitchToPathTracer
        // (IndirectLighting.cpp:1027-1275, 1321-1414) and delivered to the *Callback members (IndirectLighting.cpp:1468-1600): the same knobs as plain setters
        void SetMaxNonTrBounces(int n);                   // "Max Non-Transmissive Bounces", 1..8
        void SetMaxGlossyTrBounces(int n);                // "Max Glossy Transmissive Bounces", 1..8
        void SetStochasticMultibounce(bool b);            // ReSTIR GI / path tracer
        void SetRussianRoulette(bool b);
        void SetTemporalResampling(bool b);
        void SetSpatialResampling(int numPasses);         // "Spatial Resample", 0..2 (m_numSpatialPasses)
        void SetM_maxT(int m);                            // "M_max (Temporal)", 1..15
        void SetM_maxS(int m);                            // "M_max (Spatial)", 1..15
        void SetSortTemporal(bool b);
        void SetSortSpatial(bool b);
        void SetTexFilter(uint32_t zrTexFilter);          // ZR_TEX_FILTER_* (enum class TEXTURE_FILTER, IndirectLighting_Common.h:69-77)
        void SetBoilingSuppression(bool b);
        void SetPathRegularization(bool b);
        void SetAlphaMin(float alphaMin);                 // "Alpha_min (Reconnection)"; the constant buffers hold its square (IndirectLighting.cpp:1593-1600)
        const zr_params& Params() const { return m_params; }
        // device pointer of the FINAL plane (RGBA32F)
        void* GetOutput(SHADER_OUT_RES i) const;
        void Render(Core::CommandList& cmdList);
        // Frame overlap (ReSTIR PT; zetaray_amd.h zr_pass_set_frame_overlap): the pass as TWO graph nodes.  RenderCandidates (K11: this frame's initial candidates)
        // is registered as an ASYNC_COMPUTE node next to GBufferRT and PreLighting, RenderReuse (K12 - K16) as a COMPUTE node that consumes it -- the reference
        // overlaps work between its direct and async-compute queues the same way (Core/RenderGraph.cpp:442-541).  With the frames of a sequence submitted back
        // to back (no WaitForFrame in between) K1 + K11 of frame N + 1 run beside the search / sort / replay / reconnect kernels of frame N; the library
        // orders what crosses frames with events of its own.  GetOutput(FINAL) must be re-queried every frame (two planes alternate).
        void SetFrameOverlap(bool enable, bool carryUnusedBytes = false);
        void RenderCandidates(Core::CommandList& cmdList);
        void RenderReuse(Core::CommandList& cmdList);
    private:
        void SetFlag(uint32_t bit, bool on);
        zr_params m_params{};
    };
}

} // namespace ZetaRayAMD
