// zr_host.cpp -- implementation of the RenderPass / RenderGraph mirror (see zr_host.h).  Plain C++ (g++), links
// libzetaray_amd.so for the passes and libamdhip64 for streams/events (runtime API only; no kernels here).
#include "zr_host.h"
#include "zr_scene_io.h"
#include <algorithm>
#include <cstring>
#include <future>
#include <mutex>
#include <thread>
#include <hip/hip_runtime_api.h>

namespace ZetaRayAMD {

// ------------------------------------------------------------------------------------------------ TaskSet
Support::TaskSet::TaskHandle Support::TaskSet::EmplaceTask(const char* name, std::function<void()> f)
{
    if ((int)m_tasks.size() >= MAX_NUM_TASKS) { std::fprintf(stderr, "TaskSet: too many tasks\n"); std::abort(); }
    Task t; t.name = name; t.fn = std::move(f);
    m_tasks.push_back(std::move(t));
    return (int)m_tasks.size() - 1;
}
void Support::TaskSet::AddOutgoingEdge(TaskHandle a, TaskHandle b) { m_tasks[a].out.push_back(b); m_tasks[b].indeg++; }
void Support::TaskSet::Sort()
{
    // longest-path levels (same ordering rule the reference uses for render nodes, RenderGraph.cpp:561-642)
    std::vector<int> indeg(m_tasks.size());
    for (size_t i = 0; i < m_tasks.size(); i++) { indeg[i] = m_tasks[i].indeg; m_tasks[i].level = 0; }
    std::vector<int> q;
    for (size_t i = 0; i < m_tasks.size(); i++) if (!indeg[i]) q.push_back((int)i);
    size_t done = 0;
    while (done < q.size())
    {
        int u = q[done++];
        for (int v : m_tasks[u].out)
        {
            m_tasks[v].level = std::max(m_tasks[v].level, m_tasks[u].level + 1);
            if (--indeg[v] == 0) q.push_back(v);
        }
    }
    if (done != m_tasks.size()) { std::fprintf(stderr, "TaskSet: cycle detected\n"); std::abort(); }
    int maxL = 0;
    for (auto& t : m_tasks) maxL = std::max(maxL, t.level);
    m_levels.assign(maxL + 1, {});
    for (size_t i = 0; i < m_tasks.size(); i++) m_levels[m_tasks[i].level].push_back((int)i);
}
void Support::TaskSet::Run(bool parallel)
{
    if (m_levels.empty()) Sort();
    for (auto& level : m_levels)
    {
        if (!parallel || level.size() == 1) { for (int i : level) m_tasks[i].fn(); continue; }
        std::vector<std::future<void>> fs;
        for (int i : level) fs.push_back(std::async(std::launch::async, m_tasks[i].fn));
        for (auto& f : fs) f.get();
    }
}

// ------------------------------------------------------------------------------------------------ RenderGraph
namespace Core {

RenderGraph::RenderGraph()
{
    int n = 0;
    if (hipGetDeviceCount(&n) == hipSuccess && n > 0)
    {
        m_hasDevice = true;
        for (auto& s : m_streams) { hipStream_t st; if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { std::fprintf(stderr, "hipStreamCreate failed\n"); std::abort(); } s = st; }
    }
}
RenderGraph::~RenderGraph()
{
    if (!m_hasDevice) return;
    for (void* e : m_eventPool) (void)hipEventDestroy((hipEvent_t)e);
    for (auto& s : m_streams) if (s) (void)hipStreamDestroy((hipStream_t)s);
}
void RenderGraph::Reset() { m_nodes.clear(); m_resources.clear(); m_inPostRegister = false; }
void RenderGraph::BeginFrame() { m_nodes.clear(); m_inPostRegister = false; m_eventsUsed = 0; }

RenderNodeHandle RenderGraph::RegisterRenderPass(const char* name, RENDER_NODE_TYPE t, Delegate1<CommandList&> dlg, bool)
{
    if (m_inPostRegister) { std::fprintf(stderr, "RegisterRenderPass after MoveToPostRegister\n"); std::abort(); }
    if ((int)m_nodes.size() >= MAX_NUM_RENDER_PASSES) { std::fprintf(stderr, "too many render passes\n"); std::abort(); }
    Node n; n.name = name; n.type = t; n.dlg = std::move(dlg);
    m_nodes.push_back(std::move(n));
    return RenderNodeHandle((int)m_nodes.size() - 1);
}
void RenderGraph::RegisterResource(const void*, uint64_t path, uint32_t, bool)
{
    if (m_inPostRegister) { std::fprintf(stderr, "RegisterResource after MoveToPostRegister\n"); std::abort(); }
    if (std::find(m_resources.begin(), m_resources.end(), path) == m_resources.end())
    {
        if ((int)m_resources.size() >= MAX_NUM_RESOURCES) { std::fprintf(stderr, "too many resources\n"); std::abort(); }
        m_resources.push_back(path);
    }
}
void RenderGraph::RemoveResource(uint64_t path) { m_resources.erase(std::remove(m_resources.begin(), m_resources.end(), path), m_resources.end()); }
void RenderGraph::MoveToPostRegister() { m_inPostRegister = true; }
void RenderGraph::AddInput(RenderNodeHandle h, uint64_t path, uint32_t)
{
    if (!m_inPostRegister || !h.IsValid()) { std::fprintf(stderr, "AddInput: invalid call order\n"); std::abort(); }
    if (std::find(m_resources.begin(), m_resources.end(), path) == m_resources.end()) { std::fprintf(stderr, "AddInput: resource %llu was not registered\n", (unsigned long long)path); std::abort(); }
    m_nodes[h.Val].inputs.push_back(path);
}
void RenderGraph::AddOutput(RenderNodeHandle h, uint64_t path, uint32_t)
{
    if (!m_inPostRegister || !h.IsValid()) { std::fprintf(stderr, "AddOutput: invalid call order\n"); std::abort(); }
    if (std::find(m_resources.begin(), m_resources.end(), path) == m_resources.end()) { std::fprintf(stderr, "AddOutput: resource %llu was not registered\n", (unsigned long long)path); std::abort(); }
    m_nodes[h.Val].outputs.push_back(path);
}
void* RenderGraph::AcquireEvent()
{
    if (!m_hasDevice) return nullptr;
    if (m_eventsUsed == m_eventPool.size())
    {
        hipEvent_t e;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { std::fprintf(stderr, "hipEventCreate failed\n"); std::abort(); }
        m_eventPool.push_back(e);
    }
    return m_eventPool[m_eventsUsed++];
}

void RenderGraph::Build(Support::TaskSet& ts)
{
    const int N = (int)m_nodes.size();
    // producer -> consumer edges (a node depends on every earlier-registered node that outputs one of its inputs,
    // and on earlier writers/readers of its outputs: RAW, WAW, WAR), registration order breaks ties like the reference
    for (int i = 0; i < N; i++)
    {
        m_nodes[i].deps.clear();
        for (int j = 0; j < i; j++)
        {
            bool dep = false;
            for (uint64_t in : m_nodes[i].inputs) if (std::find(m_nodes[j].outputs.begin(), m_nodes[j].outputs.end(), in) != m_nodes[j].outputs.end()) dep = true;
            for (uint64_t out : m_nodes[i].outputs)
            {
                if (std::find(m_nodes[j].outputs.begin(), m_nodes[j].outputs.end(), out) != m_nodes[j].outputs.end()) dep = true;
                if (std::find(m_nodes[j].inputs.begin(), m_nodes[j].inputs.end(), out) != m_nodes[j].inputs.end()) dep = true;
            }
            if (dep) m_nodes[i].deps.push_back(j);
        }
    }
    // batch index = longest path from a root (RenderGraph.cpp:561-642)
    int maxBatch = 0;
    for (int i = 0; i < N; i++)
    {
        int b = 0;
        for (int j : m_nodes[i].deps) b = std::max(b, m_nodes[j].batch + 1);
        m_nodes[i].batch = b; maxBatch = std::max(maxBatch, b);
        m_nodes[i].doneEvent = AcquireEvent();
    }
    m_batchNames.assign(N ? maxBatch + 1 : 0, {});
    for (int i = 0; i < N; i++) m_batchNames[m_nodes[i].batch].push_back(m_nodes[i].name);

    // one task per node; task edges mirror the node edges so recording order == dependency order, while the GPU-side
    // order is enforced with events (the reference's barriers + cross-queue fences, RenderGraph.cpp:442-541)
    std::vector<int> handles(N);
    for (int i = 0; i < N; i++)
    {
        handles[i] = ts.EmplaceTask(m_nodes[i].name.c_str(), [this, i]()
        {
            Node& n = m_nodes[i];
            void* stream = m_streams[n.type == RENDER_NODE_TYPE::ASYNC_COMPUTE ? 1 : 0];
            if (m_hasDevice)
                for (int j : n.deps)
                {
                    void* depStream = m_streams[m_nodes[j].type == RENDER_NODE_TYPE::ASYNC_COMPUTE ? 1 : 0];
                    if (depStream != stream) (void)hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)m_nodes[j].doneEvent, 0);
                }
            CommandList cl(stream);
            n.dlg(cl);
            if (m_hasDevice) (void)hipEventRecord((hipEvent_t)n.doneEvent, (hipStream_t)stream);
        });
    }
    for (int i = 0; i < N; i++) for (int j : m_nodes[i].deps) ts.AddOutgoingEdge(handles[j], handles[i]);
    ts.Sort();
    ts.Finalize();
}
void RenderGraph::WaitForFrame()
{
    if (!m_hasDevice) return;
    for (auto& s : m_streams) (void)hipStreamSynchronize((hipStream_t)s);
}

} // namespace Core

// ------------------------------------------------------------------------------------------------ passes
namespace RenderPass {

RenderPassBase::~RenderPassBase() { if (m_pass) zr_pass_destroy(m_pass); }
void RenderPassBase::Reset(bool waitForGPU)
{
    if (!m_pass) return;
    if (waitForGPU) (void)hipDeviceSynchronize();
    zr_pass_destroy(m_pass);
    m_pass = nullptr; m_initialized = false;
}
void RenderPassBase::InitRenderPass(int kind, FrameContext* ctx, int integrator)
{
    if (m_pass) { std::fprintf(stderr, "Attempting to double-init.\n"); std::abort(); }
    m_ctx = ctx; m_integrator = integrator;
    ZR_CHECK(zr_pass_create(kind, ctx->device, &m_pass));
    ZR_CHECK(zr_pass_init(m_pass, ctx->renderWidth, ctx->renderHeight, integrator));
    m_initialized = true;
}

void GBufferRT::Init(FrameContext* ctx) { InitRenderPass(ZR_PASS_GBUFFER, ctx, 0); }
void GBufferRT::OnWindowResized() { ZR_CHECK(zr_pass_resize(m_pass, m_ctx->renderWidth, m_ctx->renderHeight)); }
void GBufferRT::PickPixel(uint16_t x, uint16_t y) { ZR_CHECK(zr_pass_pick_pixel(m_pass, x, y)); m_pickPending = true; }
void GBufferRT::ClearPick() { ZR_CHECK(zr_pass_clear_pick(m_pass)); m_pickPending = false; }
uint32_t GBufferRT::ReadPick(void* stream) const { uint32_t v = 0xffffffffu; ZR_CHECK(zr_pass_read_pick(m_pass, stream, &v)); return v; }
void GBufferRT::Render(Core::CommandList& cl) { ZR_CHECK(zr_pass_render(m_pass, cl.Stream(), &m_ctx->frameConstants, m_ctx->scene, m_ctx->gbuffer)); }

void PreLighting::Init(FrameContext* ctx) { InitRenderPass(ZR_PASS_PRELIGHTING, ctx, 0); ZR_CHECK(zr_params_default(&m_params)); }
void PreLighting::SetLightPresamplingParams(int minToEnable, int numSampleSets, int sampleSetSize)
{
    // PreLighting::Update (PreLighting.cpp:289-297): presampling is used iff the scene has >= minToEnable emissive triangles
    m_minPresample = minToEnable;
    const bool on = numSampleSets > 0 && sampleSetSize > 0 && (int)m_ctx->frameConstants.num_emissive_triangles >= minToEnable;
    m_params.presampling = on ? 1u : 0u;
    m_params.num_sample_sets = (uint32_t)numSampleSets; m_params.sample_set_size = (uint32_t)sampleSetSize;
    if (!on) m_params.use_lvg = 0;
    ZR_CHECK(zr_pass_set_params(m_pass, &m_params));
}
void PreLighting::SetLightVoxelGridParams(bool enable, uint32_t dimX, uint32_t dimY, uint32_t dimZ, float extX, float extY, float extZ, float offsetY)
{
    m_params.use_lvg = (enable && m_params.presampling) ? 1u : 0u;
    m_params.lvg_grid_dim = dimX | (dimY << 10) | (dimZ << 20);
    m_params.lvg_extents[0] = extX; m_params.lvg_extents[1] = extY; m_params.lvg_extents[2] = extZ;
    m_params.lvg_offset_y = offsetY;
    ZR_CHECK(zr_pass_set_params(m_pass, &m_params));
}
void PreLighting::Render(Core::CommandList& cl)
{
    // The alias table only changes when the emissive set changes (EmissiveTriangleAliasTable::Update, PreLighting.cpp:455-510):
    // the library skips that part itself once the scene holds a table.  K3 presampling and the K4 voxel grid are seeded by
    // FrameNum and run every frame (PreLighting.cpp:299-441), so the pass is only skipped when neither is on.
    if (m_aliasReady && !m_params.presampling && !m_params.use_lvg) return;
    ZR_CHECK(zr_pass_render(m_pass, cl.Stream(), &m_ctx->frameConstants, m_ctx->scene, nullptr));
    m_aliasReady = true;
}

void DirectLighting::Init(FrameContext* ctx)
{
    InitRenderPass(ZR_PASS_DI_EMISSIVE, ctx, 0);
    // the library installs the reference defaults for this pass kind (DirectLighting.cpp:100-107); keep a copy to edit
    zr_params_default(&m_params);
    m_params.flags = ZR_IND_TEMPORAL_RESAMPLE | ZR_IND_SPATIAL_RESAMPLE | ZR_DI_STOCHASTIC_SPATIAL | ZR_DI_EXTRA_DISOCCLUSION_SAMPLING;
    m_params.m_max_temporal = 20; m_params.m_max_spatial = 20; m_params.alpha_min = 0.05f * 0.05f;
}
void DirectLighting::OnWindowResized() { ZR_CHECK(zr_pass_resize(m_pass, m_ctx->renderWidth, m_ctx->renderHeight)); }
void DirectLighting::ResetTemporal() { ZR_CHECK(zr_pass_reset_temporal(m_pass)); }
void DirectLighting::SetLightPresamplingParams(bool enable, int numSampleSets, int sampleSetSize)
{
    m_params.presampling = enable ? 1u : 0u; m_params.num_sample_sets = (uint32_t)numSampleSets; m_params.sample_set_size = (uint32_t)sampleSetSize;
    ZR_CHECK(zr_pass_set_params(m_pass, &m_params));
}
void DirectLighting::SetFlag(uint32_t bit, bool on)
{
    m_params.flags = on ? (m_params.flags | bit) : (m_params.flags & ~bit);
    ZR_CHECK(zr_pass_set_params(m_pass, &m_params));
}
void DirectLighting::SetTemporalResampling(bool b) { SetFlag(ZR_IND_TEMPORAL_RESAMPLE, b); }
void DirectLighting::SetSpatialResampling(bool b) { SetFlag(ZR_IND_SPATIAL_RESAMPLE, b); }
void DirectLighting::SetMaxTemporalM(int m) { m_params.m_max_temporal = (uint32_t)m; ZR_CHECK(zr_pass_set_params(m_pass, &m_params)); }
void DirectLighting::SetExtraSamplesDisocclusion(bool b) { SetFlag(ZR_DI_EXTRA_DISOCCLUSION_SAMPLING, b); }
void DirectLighting::SetStochasticSpatial(bool b) { SetFlag(ZR_DI_STOCHASTIC_SPATIAL, b); }
void DirectLighting::SetAlphaMin(float a) { m_params.alpha_min = a * a; ZR_CHECK(zr_pass_set_params(m_pass, &m_params)); }
void DirectLighting::SetHalfVectorCopyShift(bool b) { SetFlag(ZR_DI_HALF_VECTOR_COPY_SHIFT, b); }
void* DirectLighting::GetOutput(SHADER_OUT_RES i) const
{
    if (i != SHADER_OUT_RES::FINAL) { std::fprintf(stderr, "Invalid shader output.\n"); std::abort(); }
    void* dev = nullptr; uint32_t w, h, bpp;
    ZR_CHECK(zr_pass_get_output(m_pass, ZR_OUT_FINAL, &dev, &w, &h, &bpp));
    return dev;
}
void DirectLighting::Render(Core::CommandList& cl) { ZR_CHECK(zr_pass_render(m_pass, cl.Stream(), &m_ctx->frameConstants, m_ctx->scene, m_ctx->gbuffer)); }

void Sky::Init(FrameContext* ctx, int lutWidth, int lutHeight)
{
    // the LUT has its own size: InitRenderPass would use the render size
    m_ctx = ctx;
    ZR_CHECK(zr_pass_create(ZR_PASS_SKY, ctx->device, &m_pass));
    ZR_CHECK(zr_pass_init(m_pass, (uint32_t)lutWidth, (uint32_t)lutHeight, 0));
    m_initialized = true;
}
void* Sky::GetOutput(SHADER_OUT_RES i) const
{
    if (i != SHADER_OUT_RES::SKY_VIEW_LUT) { std::fprintf(stderr, "Invalid shader output.\n"); std::abort(); }
    void* dev = nullptr; uint32_t w, h, bpp;
    ZR_CHECK(zr_pass_get_output(m_pass, ZR_OUT_SKY_LUT, &dev, &w, &h, &bpp));
    return dev;
}
void Sky::Render(Core::CommandList& cl) { ZR_CHECK(zr_pass_render(m_pass, cl.Stream(), &m_ctx->frameConstants, m_ctx->scene, nullptr)); }

void SkyDI::Init(FrameContext* ctx)
{
    InitRenderPass(ZR_PASS_DI_SKY, ctx, 0);       // library defaults = SkyDI.cpp:81-82; a copy to edit (sky M_max = m_max_temporal, sun M_max = m_max_spatial)
    zr_params_default(&m_params);
    m_params.flags = ZR_IND_TEMPORAL_RESAMPLE | ZR_IND_SPATIAL_RESAMPLE;
    m_params.m_max_temporal = 15; m_params.m_max_spatial = 3; m_params.alpha_min = 0.35f * 0.35f;
}
void SkyDI::SetFlag(uint32_t bit, bool on)
{
    m_params.flags = on ? (m_params.flags | bit) : (m_params.flags & ~bit);
    ZR_CHECK(zr_pass_set_params(m_pass, &m_params));
}
void SkyDI::SetTemporalResampling(bool b) { SetFlag(ZR_IND_TEMPORAL_RESAMPLE, b); }
void SkyDI::SetSpatialResampling(bool b) { SetFlag(ZR_IND_SPATIAL_RESAMPLE, b); }
void SkyDI::SetMaxMSky(int m) { m_params.m_max_temporal = (uint32_t)m; ZR_CHECK(zr_pass_set_params(m_pass, &m_params)); }
void SkyDI::SetMaxMSun(int m) { m_params.m_max_spatial = (uint32_t)m; ZR_CHECK(zr_pass_set_params(m_pass, &m_params)); }
void SkyDI::SetAlphaMin(float a) { m_params.alpha_min = a * a; ZR_CHECK(zr_pass_set_params(m_pass, &m_params)); }
void SkyDI::OnWindowResized() { ZR_CHECK(zr_pass_resize(m_pass, m_ctx->renderWidth, m_ctx->renderHeight)); }
void SkyDI::ResetTemporal() { ZR_CHECK(zr_pass_reset_temporal(m_pass)); }
void* SkyDI::GetOutput(SHADER_OUT_RES i) const
{
    if (i != SHADER_OUT_RES::DENOISED) { std::fprintf(stderr, "Invalid shader output.\n"); std::abort(); }
    void* dev = nullptr; uint32_t w, h, bpp;
    ZR_CHECK(zr_pass_get_output(m_pass, ZR_OUT_FINAL, &dev, &w, &h, &bpp));
    return dev;
}
void SkyDI::Render(Core::CommandList& cl) { ZR_CHECK(zr_pass_render(m_pass, cl.Stream(), &m_ctx->frameConstants, m_ctx->scene, m_ctx->gbuffer)); }

void Compositing::Init(FrameContext* ctx) { InitRenderPass(ZR_PASS_COMPOSITING, ctx, 0); ZR_CHECK(zr_params_default(&m_params)); }
void Compositing::OnWindowResized() { ZR_CHECK(zr_pass_resize(m_pass, m_ctx->renderWidth, m_ctx->renderHeight)); }
void Compositing::SetGpuDescriptor(SHADER_IN_GPU_DESC i, const void* dev)
{
    if (i >= SHADER_IN_GPU_DESC::COUNT) { std::fprintf(stderr, "Invalid shader input.\n"); std::abort(); }
    m_desc[(int)i] = dev;
    Rebind();
}
// an input that is switched off is unbound in the library (which composes the planes that are bound); the descriptor is remembered
void Compositing::Rebind()
{
    ZR_CHECK(zr_pass_set_input(m_pass, ZR_IN_SKY_DI, m_direct ? m_desc[(int)SHADER_IN_GPU_DESC::SKY_DI] : nullptr));
    ZR_CHECK(zr_pass_set_input(m_pass, ZR_IN_EMISSIVE_DI, m_direct ? m_desc[(int)SHADER_IN_GPU_DESC::EMISSIVE_DI] : nullptr));
    ZR_CHECK(zr_pass_set_input(m_pass, ZR_IN_INDIRECT, m_indirect ? m_desc[(int)SHADER_IN_GPU_DESC::INDIRECT] : nullptr));
}
void Compositing::SetDirectEnablement(bool b) { m_direct = b; Rebind(); }
void Compositing::SetIndirectEnablement(bool b) { m_indirect = b; Rebind(); }
void Compositing::SetFireflyFilterEnablement(bool b)
{
    m_params.flags = b ? (m_params.flags | ZR_COMPOSIT_FIREFLY_FILTER) : (m_params.flags & ~(uint32_t)ZR_COMPOSIT_FIREFLY_FILTER);
    ZR_CHECK(zr_pass_set_params(m_pass, &m_params));
}
void* Compositing::GetOutput(SHADER_OUT_RES i) const
{
    if (i != SHADER_OUT_RES::COMPOSITED) { std::fprintf(stderr, "Invalid shader output.\n"); std::abort(); }
    void* dev = nullptr; uint32_t w, h, bpp;
    ZR_CHECK(zr_pass_get_output(m_pass, ZR_OUT_FINAL, &dev, &w, &h, &bpp));
    return dev;
}
void Compositing::Render(Core::CommandList& cl) { ZR_CHECK(zr_pass_render(m_pass, cl.Stream(), &m_ctx->frameConstants, m_ctx->scene, m_ctx->gbuffer)); }

void TAA::Init(FrameContext* ctx) { InitRenderPass(ZR_PASS_TAA, ctx, 0); ZR_CHECK(zr_params_default(&m_params)); }
void TAA::OnWindowResized() { ZR_CHECK(zr_pass_resize(m_pass, m_ctx->renderWidth, m_ctx->renderHeight)); }
void TAA::SetCPUDescriptor(SHADER_IN_CPU_DESC, const void* dev) { ZR_CHECK(zr_pass_set_input(m_pass, ZR_IN_TAA_SIGNAL, dev)); }
void TAA::SetBlendWeight(float w) { m_params.taa_blend_weight = w; ZR_CHECK(zr_pass_set_params(m_pass, &m_params)); }
void TAA::ResetTemporal() { ZR_CHECK(zr_pass_reset_temporal(m_pass)); }
void* TAA::GetOutput(SHADER_OUT_RES i) const
{
    if (i >= SHADER_OUT_RES::COUNT) { std::fprintf(stderr, "Invalid shader output.\n"); std::abort(); }
    void* dev = nullptr; uint32_t w, h, bpp;
    ZR_CHECK(zr_pass_get_output(m_pass, ZR_OUT_TAA, &dev, &w, &h, &bpp));
    return dev;
}
void TAA::Render(Core::CommandList& cl) { ZR_CHECK(zr_pass_render(m_pass, cl.Stream(), &m_ctx->frameConstants, m_ctx->scene, m_ctx->gbuffer)); }

void Denoise::Init(FrameContext* ctx) { InitRenderPass(ZR_PASS_DENOISE, ctx, 0); ZR_CHECK(zr_params_default(&m_params)); }
void Denoise::OnWindowResized() { ZR_CHECK(zr_pass_resize(m_pass, m_ctx->renderWidth, m_ctx->renderHeight)); }
void Denoise::SetCPUDescriptor(SHADER_IN_CPU_DESC, const void* dev) { ZR_CHECK(zr_pass_set_input(m_pass, ZR_IN_DENOISE_SIGNAL, dev)); }
void Denoise::SetIterations(uint32_t n) { m_params.svgf_iterations = n; ZR_CHECK(zr_pass_set_params(m_pass, &m_params)); }
void Denoise::SetSigmas(float l, float z, uint32_t nlog2)
{ m_params.svgf_sigma_l = l; m_params.svgf_sigma_z = z; m_params.svgf_normal_power_log2 = nlog2; ZR_CHECK(zr_pass_set_params(m_pass, &m_params)); }
void Denoise::ResetTemporal() { ZR_CHECK(zr_pass_reset_temporal(m_pass)); }
void* Denoise::GetOutput(SHADER_OUT_RES i) const
{
    if (i >= SHADER_OUT_RES::COUNT) { std::fprintf(stderr, "Invalid shader output.\n"); std::abort(); }
    void* dev = nullptr; uint32_t w, h, bpp;
    ZR_CHECK(zr_pass_get_output(m_pass, ZR_OUT_DENOISED, &dev, &w, &h, &bpp));
    return dev;
}
void Denoise::Render(Core::CommandList& cl) { ZR_CHECK(zr_pass_render(m_pass, cl.Stream(), &m_ctx->frameConstants, m_ctx->scene, m_ctx->gbuffer)); }

void AutoExposure::Init(FrameContext* ctx) { InitRenderPass(ZR_PASS_AUTO_EXPOSURE, ctx, 0); ZR_CHECK(zr_params_default(&m_params)); }
void AutoExposure::OnWindowResized() { ZR_CHECK(zr_pass_resize(m_pass, m_ctx->renderWidth, m_ctx->renderHeight)); }
void AutoExposure::SetDescriptor(SHADER_IN_DESC, const void* dev, bool rgba16f) { ZR_CHECK(zr_pass_set_input(m_pass, rgba16f ? ZR_IN_POST_SIGNAL_F16 : ZR_IN_POST_SIGNAL_F32, dev)); }
void AutoExposure::SetMinLum(float v) { m_params.ae_min_lum = v; ZR_CHECK(zr_pass_set_params(m_pass, &m_params)); }
void AutoExposure::SetMaxLum(float v) { m_params.ae_max_lum = v; ZR_CHECK(zr_pass_set_params(m_pass, &m_params)); }
void AutoExposure::SetLumMapExp(float v) { m_params.ae_lum_map_exp = v; ZR_CHECK(zr_pass_set_params(m_pass, &m_params)); }
void* AutoExposure::GetOutput(SHADER_OUT_RES i) const
{
    if (i >= SHADER_OUT_RES::COUNT) { std::fprintf(stderr, "Invalid shader output.\n"); std::abort(); }
    void* dev = nullptr; uint32_t w, h, bpp;
    ZR_CHECK(zr_pass_get_output(m_pass, ZR_OUT_EXPOSURE, &dev, &w, &h, &bpp));
    return dev;
}
void AutoExposure::Render(Core::CommandList& cl) { ZR_CHECK(zr_pass_render(m_pass, cl.Stream(), &m_ctx->frameConstants, m_ctx->scene, nullptr)); }

void DisplayPass::Init(FrameContext* ctx, uint32_t displayWidth, uint32_t displayHeight, const uint32_t* lut, uint32_t lutDim)
{
    m_ctx = ctx;
    ZR_CHECK(zr_pass_create(ZR_PASS_DISPLAY, ctx->device, &m_pass));
    ZR_CHECK(zr_pass_init(m_pass, displayWidth, displayHeight, 0));
    m_initialized = true;
    ZR_CHECK(zr_params_default(&m_params));
    if (lut) ZR_CHECK(zr_pass_set_tonemap_lut(m_pass, lut, lutDim));
    else { m_params.display_tonemapper = ZR_TONEMAP_AGX_DEFAULT; ZR_CHECK(zr_pass_set_params(m_pass, &m_params)); }     // NEUTRAL needs the LUT
}
void DisplayPass::SetGpuDescriptor(SHADER_IN_GPU_DESC i, const void* dev, bool rgba16f)
{
    if (i == SHADER_IN_GPU_DESC::COMPOSITED) ZR_CHECK(zr_pass_set_input(m_pass, rgba16f ? ZR_IN_POST_SIGNAL_F16 : ZR_IN_POST_SIGNAL_F32, dev));
    else if (i == SHADER_IN_GPU_DESC::EXPOSURE) ZR_CHECK(zr_pass_set_input(m_pass, ZR_IN_DISPLAY_EXPOSURE, dev));
    else { std::fprintf(stderr, "out-of-bound access.\n"); std::abort(); }
}
void DisplayPass::SetTonemapper(zr_tonemapper t) { m_params.display_tonemapper = (uint32_t)t; ZR_CHECK(zr_pass_set_params(m_pass, &m_params)); }
void DisplayPass::SetAutoExposure(bool b) { m_params.display_auto_exposure = b ? 1u : 0u; ZR_CHECK(zr_pass_set_params(m_pass, &m_params)); }
void DisplayPass::SetSaturation(float v) { m_params.display_saturation = v; ZR_CHECK(zr_pass_set_params(m_pass, &m_params)); }
void DisplayPass::SetAgXExp(float v) { m_params.display_agx_exp = v; ZR_CHECK(zr_pass_set_params(m_pass, &m_params)); }
void* DisplayPass::GetOutput(SHADER_OUT_RES i) const
{
    if (i >= SHADER_OUT_RES::COUNT) { std::fprintf(stderr, "Invalid shader output.\n"); std::abort(); }
    void* dev = nullptr; uint32_t w, h, bpp;
    ZR_CHECK(zr_pass_get_output(m_pass, i == SHADER_OUT_RES::BACK_BUFFER_LINEAR ? ZR_OUT_DISPLAY : ZR_OUT_DISPLAY_SRGB8, &dev, &w, &h, &bpp));
    return dev;
}
void DisplayPass::Render(Core::CommandList& cl) { ZR_CHECK(zr_pass_render(m_pass, cl.Stream(), &m_ctx->frameConstants, m_ctx->scene, nullptr)); }

void IndirectLighting::Init(FrameContext* ctx, INTEGRATOR method)
{
    zr_params_default(&m_params);
    InitRenderPass(ZR_PASS_INDIRECT, ctx, (int)method);
}
void IndirectLighting::OnWindowResized() { ZR_CHECK(zr_pass_resize(m_pass, m_ctx->renderWidth, m_ctx->renderHeight)); }
void IndirectLighting::ResetTemporal() { ZR_CHECK(zr_pass_reset_temporal(m_pass)); }
void IndirectLighting::SetMethod(INTEGRATOR method)
{
    m_integrator = (int)method;
    ZR_CHECK(zr_pass_init(m_pass, m_ctx->renderWidth, m_ctx->renderHeight, m_integrator));
    ZR_CHECK(zr_pass_set_params(m_pass, &m_params));
}
void IndirectLighting::SetLightPresamplingParams(bool enable, int numSampleSets, int sampleSetSize)
{
    m_params.presampling = enable ? 1u : 0u; m_params.num_sample_sets = (uint32_t)numSampleSets; m_params.sample_set_size = (uint32_t)sampleSetSize;
    ZR_CHECK(zr_pass_set_params(m_pass, &m_params));
}
void IndirectLighting::SetMaxBounces(int nonTr, int glossyTr)
{
    m_params.max_non_tr_bounces = (uint32_t)nonTr; m_params.max_glossy_tr_bounces = (uint32_t)glossyTr;
    ZR_CHECK(zr_pass_set_params(m_pass, &m_params));
}
void IndirectLighting::SetLightVoxelGridParams(bool enabled, uint32_t dimX, uint32_t dimY, uint32_t dimZ, float extX, float extY, float extZ, float offsetY)
{
    if (enabled && !(dimX > 0 && dimY > 0 && dimZ > 0 && extX > 0 && extY > 0 && extZ > 0)) { std::fprintf(stderr, "LVG is enabled, but the dimension is invalid.\n"); std::abort(); }
    if (enabled && !m_params.presampling) { std::fprintf(stderr, "LVG can't be used while light presampling is disabled.\n"); std::abort(); }      // IndirectLighting.h:93
    m_params.use_lvg = enabled ? 1u : 0u; m_params.lvg_grid_dim = dimX | (dimY << 10) | (dimZ << 20);
    m_params.lvg_extents[0] = extX; m_params.lvg_extents[1] = extY; m_params.lvg_extents[2] = extZ; m_params.lvg_offset_y = offsetY;
    ZR_CHECK(zr_pass_set_params(m_pass, &m_params));
}
void IndirectLighting::SetFlag(uint32_t bit, bool on)
{
    m_params.flags = on ? (m_params.flags | bit) : (m_params.flags & ~bit);
    ZR_CHECK(zr_pass_set_params(m_pass, &m_params));
}
void IndirectLighting::SetMaxNonTrBounces(int n) { m_params.max_non_tr_bounces = (uint32_t)n; ZR_CHECK(zr_pass_set_params(m_pass, &m_params)); }
void IndirectLighting::SetMaxGlossyTrBounces(int n) { m_params.max_glossy_tr_bounces = (uint32_t)n; ZR_CHECK(zr_pass_set_params(m_pass, &m_params)); }
void IndirectLighting::SetStochasticMultibounce(bool b) { SetFlag(ZR_IND_STOCHASTIC_MULTI_BOUNCE, b); }
void IndirectLighting::SetRussianRoulette(bool b) { SetFlag(ZR_IND_RUSSIAN_ROULETTE, b); }
void IndirectLighting::SetTemporalResampling(bool b) { SetFlag(ZR_IND_TEMPORAL_RESAMPLE, b); }
// m_numSpatialPasses (IndirectLighting.cpp:1514-1518): doSpatial = (m_numSpatialPasses > 0) && doTemporal (:906)
void IndirectLighting::SetSpatialResampling(int numPasses)
{
    m_params.num_spatial_passes = (uint32_t)numPasses;
    SetFlag(ZR_IND_SPATIAL_RESAMPLE, numPasses > 0);
}
void IndirectLighting::SetM_maxT(int m) { m_params.m_max_temporal = (uint32_t)m; ZR_CHECK(zr_pass_set_params(m_pass, &m_params)); }
void IndirectLighting::SetM_maxS(int m) { m_params.m_max_spatial = (uint32_t)m; ZR_CHECK(zr_pass_set_params(m_pass, &m_params)); }
void IndirectLighting::SetSortTemporal(bool b) { SetFlag(ZR_IND_SORT_TEMPORAL, b); }
void IndirectLighting::SetSortSpatial(bool b) { SetFlag(ZR_IND_SORT_SPATIAL, b); }
void IndirectLighting::SetTexFilter(uint32_t f) { m_params.tex_filter = f; ZR_CHECK(zr_pass_set_params(m_pass, &m_params)); }
void IndirectLighting::SetBoilingSuppression(bool b) { SetFlag(ZR_IND_BOILING_SUPPRESSION, b); }
void IndirectLighting::SetPathRegularization(bool b) { SetFlag(ZR_IND_PATH_REGULARIZATION, b); }
void IndirectLighting::SetAlphaMin(float a) { m_params.alpha_min = a * a; ZR_CHECK(zr_pass_set_params(m_pass, &m_params)); }
void* IndirectLighting::GetOutput(SHADER_OUT_RES i) const
{
    if (i != SHADER_OUT_RES::FINAL) { std::fprintf(stderr, "Invalid shader output.\n"); std::abort(); }
    void* dev = nullptr; uint32_t w, h, bpp;
    ZR_CHECK(zr_pass_get_output(m_pass, ZR_OUT_FINAL, &dev, &w, &h, &bpp));
    return dev;
}
void IndirectLighting::Render(Core::CommandList& cl) { ZR_CHECK(zr_pass_render(m_pass, cl.Stream(), &m_ctx->frameConstants, m_ctx->scene, m_ctx->gbuffer)); }
void IndirectLighting::SetFrameOverlap(bool enable, bool carry)
{ ZR_CHECK(zr_pass_set_frame_overlap(m_pass, m_ctx->gbuffer, enable ? (carry ? ZR_FRAME_OVERLAP_CARRY : ZR_FRAME_OVERLAP) : 0)); }
void IndirectLighting::RenderCandidates(Core::CommandList& cl)
{ ZR_CHECK(zr_pass_render_stage(m_pass, cl.Stream(), &m_ctx->frameConstants, m_ctx->scene, m_ctx->gbuffer, ZR_STAGE_CANDIDATES)); }
void IndirectLighting::RenderReuse(Core::CommandList& cl)
{ ZR_CHECK(zr_pass_render_stage(m_pass, cl.Stream(), &m_ctx->frameConstants, m_ctx->scene, m_ctx->gbuffer, ZR_STAGE_TEMPORAL_REUSE | ZR_STAGE_SPATIAL | ZR_STAGE_SPATIAL2)); }

} // namespace RenderPass
} // namespace ZetaRayAMD

// ------------------------------------------------------------------------------------------------ C test hooks
// Small extern "C" surface so the Python test-suite can exercise the C++ layer (graph ordering without a GPU; a full
// frame through GBufferRT -> PreLighting -> IndirectLighting on the GPU box).
using namespace ZetaRayAMD;

extern "C" {

// Registers the reference's hot-path node set with the reference's resource dependencies (ZR/PathTracer.cpp:325-563,
// ZR/GBuffer.cpp) using dummy delegates and returns the batch layout as a string "a,b|c|d,e".
int zrh_graph_selftest(char* out, int outLen)
{
    Core::RenderGraph g;
    struct Dummy { std::vector<std::string>* log; std::string name; std::mutex* m; void Render(Core::CommandList&) { std::lock_guard<std::mutex> l(*m); log->push_back(name); } };
    std::vector<std::string> log; std::mutex mtx;
    Dummy as{&log, "RT_AS_Build", &mtx}, gb{&log, "GBuffer", &mtx}, sky{&log, "Sky", &mtx}, pre{&log, "PreLighting", &mtx}, alias{&log, "EmissiveAliasTable", &mtx},
        di{&log, "DirectLighting", &mtx}, ind{&log, "Indirect", &mtx}, comp{&log, "Compositing", &mtx};
    enum : uint64_t { R_BVH = 100, R_GBUF, R_SKYLUT, R_POWER, R_ALIAS, R_DI, R_IND, R_HDR };
    g.BeginFrame();
    auto hAS = g.RegisterRenderPass("RT_AS_Build", Core::RENDER_NODE_TYPE::COMPUTE, Core::MakeDelegate(&as, &Dummy::Render));
    auto hGB = g.RegisterRenderPass("GBuffer", Core::RENDER_NODE_TYPE::COMPUTE, Core::MakeDelegate(&gb, &Dummy::Render));
    auto hSky = g.RegisterRenderPass("Sky", Core::RENDER_NODE_TYPE::ASYNC_COMPUTE, Core::MakeDelegate(&sky, &Dummy::Render));
    auto hPre = g.RegisterRenderPass("PreLighting", Core::RENDER_NODE_TYPE::COMPUTE, Core::MakeDelegate(&pre, &Dummy::Render));
    auto hAl = g.RegisterRenderPass("EmissiveAliasTable", Core::RENDER_NODE_TYPE::COMPUTE, Core::MakeDelegate(&alias, &Dummy::Render));
    auto hDI = g.RegisterRenderPass("DirectLighting", Core::RENDER_NODE_TYPE::COMPUTE, Core::MakeDelegate(&di, &Dummy::Render));
    auto hInd = g.RegisterRenderPass("Indirect", Core::RENDER_NODE_TYPE::ASYNC_COMPUTE, Core::MakeDelegate(&ind, &Dummy::Render));
    auto hC = g.RegisterRenderPass("Compositing", Core::RENDER_NODE_TYPE::COMPUTE, Core::MakeDelegate(&comp, &Dummy::Render));
    for (uint64_t r = R_BVH; r <= R_HDR; r++) g.RegisterResource(nullptr, r);
    g.MoveToPostRegister();
    g.AddOutput(hAS, R_BVH, Core::STATE_UNORDERED_ACCESS);
    g.AddInput(hGB, R_BVH, Core::STATE_SHADER_READ); g.AddOutput(hGB, R_GBUF, Core::STATE_UNORDERED_ACCESS);
    g.AddOutput(hSky, R_SKYLUT, Core::STATE_UNORDERED_ACCESS);
    g.AddOutput(hPre, R_POWER, Core::STATE_UNORDERED_ACCESS);
    g.AddInput(hAl, R_POWER, Core::STATE_SHADER_READ); g.AddOutput(hAl, R_ALIAS, Core::STATE_UNORDERED_ACCESS);
    g.AddInput(hDI, R_GBUF, Core::STATE_SHADER_READ); g.AddInput(hDI, R_ALIAS, Core::STATE_SHADER_READ); g.AddInput(hDI, R_BVH, Core::STATE_SHADER_READ); g.AddOutput(hDI, R_DI, Core::STATE_UNORDERED_ACCESS);
    g.AddInput(hInd, R_GBUF, Core::STATE_SHADER_READ); g.AddInput(hInd, R_ALIAS, Core::STATE_SHADER_READ); g.AddInput(hInd, R_BVH, Core::STATE_SHADER_READ); g.AddInput(hInd, R_SKYLUT, Core::STATE_SHADER_READ); g.AddOutput(hInd, R_IND, Core::STATE_UNORDERED_ACCESS);
    g.AddInput(hC, R_DI, Core::STATE_SHADER_READ); g.AddInput(hC, R_IND, Core::STATE_SHADER_READ); g.AddOutput(hC, R_HDR, Core::STATE_UNORDERED_ACCESS);
    Support::TaskSet ts;
    g.Build(ts);
    ts.Run(true);
    g.WaitForFrame();
    std::string s;
    for (size_t b = 0; b < g.Batches().size(); b++) { if (b) s += "|"; for (size_t i = 0; i < g.Batches()[b].size(); i++) { if (i) s += ","; s += g.Batches()[b][i]; } }
    s += "#";
    for (size_t i = 0; i < log.size(); i++) { if (i) s += ","; s += log[i]; }
    std::snprintf(out, outLen, "%s", s.c_str());
    return (int)log.size();
}

// One frame through the C++ pass objects scheduled by the graph; copies FINAL (w*h*4 floats) to `finalOut`.
// Renders `n` consecutive frames (cbs[i] = cbFrameConstants of frame i) through the graph exactly as the reference's
// frame loop does (PathTracer.cpp:474-552: register passes / resources, declare inputs / outputs, Build, submit, fence)
// and copies the FINAL plane of the last frame.  integrator: 0 = PATH_TRACING, 2 = ReSTIR_PT.
int zrh_render_sequence3(const zr_scene_desc* desc, const zr_frame_constants* cbs, uint32_t n, uint32_t w, uint32_t h, int integrator, float* finalOut, float* directOut,
    int presampleSets, int presampleSize);
// the reference's UI parameters of the two lighting passes as one record (tests): every field < 0 leaves the pass's default alone
struct zrh_tuning
{
    int max_non_tr, max_glossy_tr, stochastic_multibounce, russian_roulette, temporal, spatial_passes, m_max_t, m_max_s, sort_temporal, sort_spatial, boiling_suppression,
        path_regularization;
    float alpha_min;
    int di_temporal, di_spatial, di_m_max, di_extra_disocclusion, di_stochastic_spatial;
    float di_alpha_min;
};
static const zrh_tuning* g_tuning = nullptr;
int zrh_render_sequence_tuned(const zr_scene_desc* desc, const zr_frame_constants* cbs, uint32_t n, uint32_t w, uint32_t h, int integrator, const zrh_tuning* t,
    float* finalOut, float* directOut)
{
    g_tuning = t;
    const int r = zrh_render_sequence3(desc, cbs, n, w, h, integrator, finalOut, directOut, 0, 0);
    g_tuning = nullptr;
    return r;
}
int zrh_render_sequence2(const zr_scene_desc* desc, const zr_frame_constants* cbs, uint32_t n, uint32_t w, uint32_t h, int integrator, float* finalOut, float* directOut)
{ return zrh_render_sequence3(desc, cbs, n, w, h, integrator, finalOut, directOut, 0, 0); }

// ... with light presampling (K3) when presampleSets > 0: PreLighting regenerates the sets every frame and the lighting passes
// read them (DefaultRenderer.cpp:257-272: the three SetLightPresamplingParams calls of the reference's settings callback)
int zrh_render_sequence3(const zr_scene_desc* desc, const zr_frame_constants* cbs, uint32_t n, uint32_t w, uint32_t h, int integrator, float* finalOut, float* directOut,
    int presampleSets, int presampleSize)
{
    RenderPass::FrameContext ctx;
    ctx.device = 0; ctx.renderWidth = w; ctx.renderHeight = h;
    ZR_CHECK(zr_scene_create(0, desc, &ctx.scene));
    ZR_CHECK(zr_gbuffer_create(0, w, h, &ctx.gbuffer));
    {
        RenderPass::GBufferRT gb; RenderPass::PreLighting pre; RenderPass::IndirectLighting ind; RenderPass::DirectLighting di;
        gb.Init(&ctx); pre.Init(&ctx); ind.Init(&ctx, (RenderPass::IndirectLighting::INTEGRATOR)integrator);
        if (directOut) di.Init(&ctx);
        if (presampleSets > 0)
        {
            ctx.frameConstants = cbs[0];
            pre.SetLightPresamplingParams(0, presampleSets, presampleSize);
            ind.SetLightPresamplingParams(pre.IsPresamplingEnabled(), presampleSets, presampleSize);
            if (directOut) di.SetLightPresamplingParams(pre.IsPresamplingEnabled(), presampleSets, presampleSize);
        }
        if (const zrh_tuning* t = g_tuning)
        {   // what the reference's settings UI does between frames: one callback per knob (IndirectLighting.cpp:1468-1600, DirectLighting.cpp:374-410)
            if (t->max_non_tr >= 0) ind.SetMaxNonTrBounces(t->max_non_tr);
            if (t->max_glossy_tr >= 0) ind.SetMaxGlossyTrBounces(t->max_glossy_tr);
            if (t->stochastic_multibounce >= 0) ind.SetStochasticMultibounce(t->stochastic_multibounce != 0);
            if (t->russian_roulette >= 0) ind.SetRussianRoulette(t->russian_roulette != 0);
            if (t->temporal >= 0) ind.SetTemporalResampling(t->temporal != 0);
            if (t->spatial_passes >= 0) ind.SetSpatialResampling(t->spatial_passes);
            if (t->m_max_t >= 0) ind.SetM_maxT(t->m_max_t);
            if (t->m_max_s >= 0) ind.SetM_maxS(t->m_max_s);
            if (t->sort_temporal >= 0) ind.SetSortTemporal(t->sort_temporal != 0);
            if (t->sort_spatial >= 0) ind.SetSortSpatial(t->sort_spatial != 0);
            if (t->boiling_suppression >= 0) ind.SetBoilingSuppression(t->boiling_suppression != 0);
            if (t->path_regularization >= 0) ind.SetPathRegularization(t->path_regularization != 0);
            if (t->alpha_min >= 0) ind.SetAlphaMin(t->alpha_min);
            if (directOut)
            {
                if (t->di_temporal >= 0) di.SetTemporalResampling(t->di_temporal != 0);
                if (t->di_spatial >= 0) di.SetSpatialResampling(t->di_spatial != 0);
                if (t->di_m_max >= 0) di.SetMaxTemporalM(t->di_m_max);
                if (t->di_extra_disocclusion >= 0) di.SetExtraSamplesDisocclusion(t->di_extra_disocclusion != 0);
                if (t->di_stochastic_spatial >= 0) di.SetStochasticSpatial(t->di_stochastic_spatial != 0);
                if (t->di_alpha_min >= 0) di.SetAlphaMin(t->di_alpha_min);
            }
        }
        Core::RenderGraph g;
        enum : uint64_t { R_GBUF = 1, R_ALIAS, R_IND, R_DI };
        for (uint32_t f = 0; f < n; f++)
        {
            ctx.frameConstants = cbs[f];
            g.BeginFrame();
            auto hGB = g.RegisterRenderPass("GBuffer", Core::RENDER_NODE_TYPE::COMPUTE, Core::MakeDelegate(&gb, &RenderPass::GBufferRT::Render));
            auto hPre = g.RegisterRenderPass("PreLighting", Core::RENDER_NODE_TYPE::ASYNC_COMPUTE, Core::MakeDelegate(&pre, &RenderPass::PreLighting::Render));
            auto hInd = g.RegisterRenderPass("Indirect", Core::RENDER_NODE_TYPE::COMPUTE, Core::MakeDelegate(&ind, &RenderPass::IndirectLighting::Render));
            // DirectLighting sits in the same dependency level as Indirect (PathTracer.cpp:474-552): both read the G-buffer + alias table
            Core::RenderNodeHandle hDI;
            if (directOut) hDI = g.RegisterRenderPass("DirectLighting", Core::RENDER_NODE_TYPE::COMPUTE, Core::MakeDelegate(&di, &RenderPass::DirectLighting::Render));
            g.RegisterResource(nullptr, R_GBUF); g.RegisterResource(nullptr, R_ALIAS); g.RegisterResource(ind.GetOutput(RenderPass::IndirectLighting::SHADER_OUT_RES::FINAL), R_IND);
            if (directOut) g.RegisterResource(di.GetOutput(RenderPass::DirectLighting::SHADER_OUT_RES::FINAL), R_DI);
            g.MoveToPostRegister();
            g.AddOutput(hGB, R_GBUF, Core::STATE_UNORDERED_ACCESS);
            g.AddOutput(hPre, R_ALIAS, Core::STATE_UNORDERED_ACCESS);
            if (directOut) { g.AddInput(hDI, R_GBUF, Core::STATE_SHADER_READ); g.AddInput(hDI, R_ALIAS, Core::STATE_SHADER_READ); g.AddOutput(hDI, R_DI, Core::STATE_UNORDERED_ACCESS); }
            g.AddInput(hInd, R_GBUF, Core::STATE_SHADER_READ); g.AddInput(hInd, R_ALIAS, Core::STATE_SHADER_READ); g.AddOutput(hInd, R_IND, Core::STATE_UNORDERED_ACCESS);
            Support::TaskSet ts;
            g.Build(ts);
            ts.Run(true);
            g.WaitForFrame();
        }
        if (hipMemcpy(finalOut, ind.GetOutput(RenderPass::IndirectLighting::SHADER_OUT_RES::FINAL), (size_t)w * h * 16, hipMemcpyDeviceToHost) != hipSuccess) return -1;
        if (directOut && hipMemcpy(directOut, di.GetOutput(RenderPass::DirectLighting::SHADER_OUT_RES::FINAL), (size_t)w * h * 16, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    }
    zr_gbuffer_destroy(ctx.gbuffer);
    zr_scene_destroy(ctx.scene);
    return 0;
}

// A ReSTIR PT sequence with the indirect pass as two graph nodes on two queues (IndirectLighting::SetFrameOverlap): GBufferRT, PreLighting and
// Indirect.Candidates are ASYNC_COMPUTE nodes, Indirect.Reuse a COMPUTE node; the frames are submitted back to back and only the last one is waited for,
// so the first half of frame N + 1 runs beside the second half of frame N.  overlapMode 0 = the plain single-node form of zrh_render_sequence3 (for the
// comparison), 1 = ZR_FRAME_OVERLAP, 2 = ZR_FRAME_OVERLAP_CARRY.  Copies FINAL of the last frame; batchesOut (optional): the graph's batches of the last frame.
int zrh_render_sequence_overlap(const zr_scene_desc* desc, const zr_frame_constants* cbs, uint32_t n, uint32_t w, uint32_t h, int overlapMode, float* finalOut,
    int presampleSets, int presampleSize, char* batchesOut, int batchesCap)
{
    RenderPass::FrameContext ctx;
    ctx.device = 0; ctx.renderWidth = w; ctx.renderHeight = h;
    ZR_CHECK(zr_scene_create(0, desc, &ctx.scene));
    ZR_CHECK(zr_gbuffer_create(0, w, h, &ctx.gbuffer));
    {
        RenderPass::GBufferRT gb; RenderPass::PreLighting pre; RenderPass::IndirectLighting ind;
        gb.Init(&ctx); pre.Init(&ctx); ind.Init(&ctx, RenderPass::IndirectLighting::INTEGRATOR::ReSTIR_PT);
        if (presampleSets > 0)
        {
            ctx.frameConstants = cbs[0];
            pre.SetLightPresamplingParams(0, presampleSets, presampleSize);
            ind.SetLightPresamplingParams(pre.IsPresamplingEnabled(), presampleSets, presampleSize);
        }
        if (overlapMode) ind.SetFrameOverlap(true, overlapMode == 2);
        Core::RenderGraph g;
        enum : uint64_t { R_GBUF = 1, R_ALIAS, R_CAND, R_IND };
        for (uint32_t f = 0; f < n; f++)
        {
            ctx.frameConstants = cbs[f];
            g.BeginFrame();
            const auto first = overlapMode ? Core::RENDER_NODE_TYPE::ASYNC_COMPUTE : Core::RENDER_NODE_TYPE::COMPUTE;
            auto hGB = g.RegisterRenderPass("GBuffer", first, Core::MakeDelegate(&gb, &RenderPass::GBufferRT::Render));
            auto hPre = g.RegisterRenderPass("PreLighting", Core::RENDER_NODE_TYPE::ASYNC_COMPUTE, Core::MakeDelegate(&pre, &RenderPass::PreLighting::Render));
            Core::RenderNodeHandle hCand, hInd;
            if (overlapMode)
            {
                hCand = g.RegisterRenderPass("Indirect.Candidates", Core::RENDER_NODE_TYPE::ASYNC_COMPUTE, Core::MakeDelegate(&ind, &RenderPass::IndirectLighting::RenderCandidates));
                hInd = g.RegisterRenderPass("Indirect.Reuse", Core::RENDER_NODE_TYPE::COMPUTE, Core::MakeDelegate(&ind, &RenderPass::IndirectLighting::RenderReuse));
            }
            else hInd = g.RegisterRenderPass("Indirect", Core::RENDER_NODE_TYPE::COMPUTE, Core::MakeDelegate(&ind, &RenderPass::IndirectLighting::Render));
            g.RegisterResource(nullptr, R_GBUF); g.RegisterResource(nullptr, R_ALIAS); g.RegisterResource(nullptr, R_CAND); g.RegisterResource(nullptr, R_IND);
            g.MoveToPostRegister();
            g.AddOutput(hGB, R_GBUF, Core::STATE_UNORDERED_ACCESS);
            g.AddOutput(hPre, R_ALIAS, Core::STATE_UNORDERED_ACCESS);
            if (overlapMode)
            {
                g.AddInput(hCand, R_GBUF, Core::STATE_SHADER_READ); g.AddInput(hCand, R_ALIAS, Core::STATE_SHADER_READ); g.AddOutput(hCand, R_CAND, Core::STATE_UNORDERED_ACCESS);
                g.AddInput(hInd, R_CAND, Core::STATE_SHADER_READ); g.AddInput(hInd, R_GBUF, Core::STATE_SHADER_READ);
            }
            else { g.AddInput(hInd, R_GBUF, Core::STATE_SHADER_READ); g.AddInput(hInd, R_ALIAS, Core::STATE_SHADER_READ); }
            g.AddOutput(hInd, R_IND, Core::STATE_UNORDERED_ACCESS);
            Support::TaskSet ts;
            g.Build(ts);
            ts.Run(true);
            // the alias table's first build (frame 0) is read by both halves of every later frame: one wait, like the reference's first-frame upload fence
            if (!overlapMode || f == 0 || f + 1 == n) g.WaitForFrame();
        }
        if (batchesOut && batchesCap > 0)
        {
            std::string txt;
            for (size_t b = 0; b < g.Batches().size(); b++) { if (b) txt += "|"; for (size_t k = 0; k < g.Batches()[b].size(); k++) { if (k) txt += ","; txt += g.Batches()[b][k]; } }
            std::snprintf(batchesOut, (size_t)batchesCap, "%s", txt.c_str());
        }
        if (hipMemcpy(finalOut, ind.GetOutput(RenderPass::IndirectLighting::SHADER_OUT_RES::FINAL), (size_t)w * h * 16, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    }
    zr_gbuffer_destroy(ctx.gbuffer);
    zr_scene_destroy(ctx.scene);
    return 0;
}

// The reference's default (sun + sky) frame: Sky -> GBuffer -> {SkyDI, Indirect} (PathTracer.cpp:165-181, 311-360, 474-552).
// Copies the FINAL planes of the last frame: Indirect to `finalOut`, SkyDI to `skyDiOut`.
// Scene ingestion on the host (zr_scene_io.cpp) straight into a device scene: Model::glTF::Load + SceneCore + TLAS build of the reference
// collapsed into one call.  tex_offsets4 = the descriptor-table offsets to put into cbFrameConstants (base colour, normal, MR, emissive).
extern "C" int zrh_scene_create_from_gltf(int device, const char* path, const uint16_t* rho_lut, const uint32_t* rho_dim3, zr_scene** out, uint32_t* tex_offsets4)
{
    zrh_scene_data* data = nullptr;
    if (zrh_gltf_load(path, rho_lut, rho_dim3, &data) != 0) { std::fprintf(stderr, "zrh_scene_create_from_gltf: %s\n", zrh_scene_io_last_error()); return -1; }
    const int r = zr_scene_create(device, zrh_scene_data_desc(data), out);
    if (tex_offsets4) zrh_scene_data_tex_offsets(data, tex_offsets4);
    zrh_scene_data_destroy(data);
    return r;
}

// Per-frame hand-over of a host-maintained scene (zr_scene_io.h: begin_frame / set_instance_world) to the device scene: the moved lights' records,
// then the instance buffer + matrices (TLAS update: refit on the device, previous structure kept for the passes that bind it).
// Stream-ordered (zr_scene_update_*_async): enqueued on `stream` like the reference records the TLAS update and the emissive upload on the
// frame's command list; renders the graph issues on its own (non-blocking) streams are ordered behind it by the library's events.
extern "C" int zrh_scene_apply_updates_on(zr_scene* scene, const zrh_scene_data* data, void* stream)
{
    if (!scene || !data) return -1;
    const zr_scene_desc* d = zrh_scene_data_desc(data);
    uint32_t first = 0, count = 0;
    zrh_scene_data_dirty_emissives(data, &first, &count);
    if (count) { const int r = zr_scene_update_emissives_async(scene, stream, d->emissives + first, first, count); if (r) return r; }
    return zr_scene_update_instances_async(scene, stream, d->instances, d->instance_to_world, d->num_instances);
}
extern "C" int zrh_scene_apply_updates(zr_scene* scene, const zrh_scene_data* data) { return zrh_scene_apply_updates_on(scene, data, nullptr); }

int zrh_render_sequence_sky_display(const zr_scene_desc* desc, const zr_frame_constants* cbs, uint32_t n, uint32_t w, uint32_t h, int integrator, float* finalOut, float* skyDiOut,
    float* compositedOut, uint16_t* taaOut, const uint32_t* lutRGB9E5, uint32_t lutDim, int tonemapper, float* exposureOut, float* displayOut, uint8_t* displaySrgbOut);
int zrh_render_sequence_sky_post(const zr_scene_desc* desc, const zr_frame_constants* cbs, uint32_t n, uint32_t w, uint32_t h, int integrator, float* finalOut, float* skyDiOut,
    float* compositedOut, uint16_t* taaOut)
{ return zrh_render_sequence_sky_display(desc, cbs, n, w, h, integrator, finalOut, skyDiOut, compositedOut, taaOut, nullptr, 0, 0, nullptr, nullptr, nullptr); }
int zrh_render_sequence_sky(const zr_scene_desc* desc, const zr_frame_constants* cbs, uint32_t n, uint32_t w, uint32_t h, int integrator, float* finalOut, float* skyDiOut)
{ return zrh_render_sequence_sky_post(desc, cbs, n, w, h, integrator, finalOut, skyDiOut, nullptr, nullptr); }

// ... followed by Compositing -> TAA when compositedOut / taaOut are given (DefaultRenderer's post chain, Compositing.cpp / TAA.cpp):
// compositedOut = RGBA32F of the last frame, taaOut = RGBA16F bits of the last frame
// ... and by AutoExposure -> Display on the TAA output when displayOut is given (DefaultRenderer.cpp: the last two nodes of the frame):
// exposureOut = the (exposure, adapted luminance) texel, displayOut = RGBA32F, displaySrgbOut = RGBA8 of the last frame (display size = w x h)
int zrh_render_sequence_sky_display(const zr_scene_desc* desc, const zr_frame_constants* cbs, uint32_t n, uint32_t w, uint32_t h, int integrator, float* finalOut, float* skyDiOut,
    float* compositedOut, uint16_t* taaOut, const uint32_t* lutRGB9E5, uint32_t lutDim, int tonemapper, float* exposureOut, float* displayOut, uint8_t* displaySrgbOut)
{
    const bool display = displayOut || displaySrgbOut || exposureOut;
    const bool post = compositedOut || taaOut || display;
    RenderPass::FrameContext ctx;
    ctx.device = 0; ctx.renderWidth = w; ctx.renderHeight = h;
    ZR_CHECK(zr_scene_create(0, desc, &ctx.scene));
    ZR_CHECK(zr_gbuffer_create(0, w, h, &ctx.gbuffer));
    {
        RenderPass::Sky sky; RenderPass::GBufferRT gb; RenderPass::IndirectLighting ind; RenderPass::SkyDI sdi;
        sky.Init(&ctx, 256, 128); gb.Init(&ctx); ind.Init(&ctx, (RenderPass::IndirectLighting::INTEGRATOR)integrator); sdi.Init(&ctx);
        RenderPass::Compositing comp; RenderPass::TAA taa;
        if (post)
        {
            comp.Init(&ctx); taa.Init(&ctx);
            comp.SetGpuDescriptor(RenderPass::Compositing::SHADER_IN_GPU_DESC::SKY_DI, sdi.GetOutput(RenderPass::SkyDI::SHADER_OUT_RES::DENOISED));
            comp.SetGpuDescriptor(RenderPass::Compositing::SHADER_IN_GPU_DESC::INDIRECT, ind.GetOutput(RenderPass::IndirectLighting::SHADER_OUT_RES::FINAL));
            taa.SetCPUDescriptor(RenderPass::TAA::SHADER_IN_CPU_DESC::SIGNAL, comp.GetOutput(RenderPass::Compositing::SHADER_OUT_RES::COMPOSITED));
        }
        RenderPass::AutoExposure ae; RenderPass::DisplayPass disp;
        // TAA ping-pongs its two targets: like DefaultRenderer's per-frame Update(), the consumers bind "the TAA target of this frame" when
        // their node records, i.e. after the TAA node has run (they depend on R_TAA)
        struct AeNode { RenderPass::AutoExposure* ae; RenderPass::TAA* taa; void Render(Core::CommandList& cl)
            { ae->SetDescriptor(RenderPass::AutoExposure::SHADER_IN_DESC::COMPOSITED, taa->GetOutput(RenderPass::TAA::SHADER_OUT_RES::OUTPUT_A), true); ae->Render(cl); } } aeNode{&ae, &taa};
        struct DispNode { RenderPass::DisplayPass* d; RenderPass::TAA* taa; void Render(Core::CommandList& cl)
            { d->SetGpuDescriptor(RenderPass::DisplayPass::SHADER_IN_GPU_DESC::COMPOSITED, taa->GetOutput(RenderPass::TAA::SHADER_OUT_RES::OUTPUT_A), true); d->Render(cl); } } dispNode{&disp, &taa};
        if (display)
        {
            ae.Init(&ctx); disp.Init(&ctx, w, h, lutRGB9E5, lutDim);
            disp.SetTonemapper((zr_tonemapper)tonemapper);
            disp.SetGpuDescriptor(RenderPass::DisplayPass::SHADER_IN_GPU_DESC::EXPOSURE, ae.GetOutput(RenderPass::AutoExposure::SHADER_OUT_RES::EXPOSURE));
        }
        Core::RenderGraph g;
        enum : uint64_t { R_LUT = 1, R_GBUF, R_IND, R_SDI, R_COMP, R_TAA, R_EXPOSURE, R_BACKBUFFER };
        for (uint32_t f = 0; f < n; f++)
        {
            ctx.frameConstants = cbs[f];
            g.BeginFrame();
            auto hSky = g.RegisterRenderPass("Sky", Core::RENDER_NODE_TYPE::COMPUTE, Core::MakeDelegate(&sky, &RenderPass::Sky::Render));
            auto hGB = g.RegisterRenderPass("GBuffer", Core::RENDER_NODE_TYPE::COMPUTE, Core::MakeDelegate(&gb, &RenderPass::GBufferRT::Render));
            auto hSdi = g.RegisterRenderPass("SkyDI", Core::RENDER_NODE_TYPE::COMPUTE, Core::MakeDelegate(&sdi, &RenderPass::SkyDI::Render));
            auto hInd = g.RegisterRenderPass("Indirect", Core::RENDER_NODE_TYPE::COMPUTE, Core::MakeDelegate(&ind, &RenderPass::IndirectLighting::Render));
            g.RegisterResource(sky.GetOutput(RenderPass::Sky::SHADER_OUT_RES::SKY_VIEW_LUT), R_LUT); g.RegisterResource(nullptr, R_GBUF);
            g.RegisterResource(ind.GetOutput(RenderPass::IndirectLighting::SHADER_OUT_RES::FINAL), R_IND);
            g.RegisterResource(sdi.GetOutput(RenderPass::SkyDI::SHADER_OUT_RES::DENOISED), R_SDI);
            Core::RenderNodeHandle hComp{}, hTaa{};
            if (post)
            {
                hComp = g.RegisterRenderPass("Compositing", Core::RENDER_NODE_TYPE::COMPUTE, Core::MakeDelegate(&comp, &RenderPass::Compositing::Render));
                hTaa = g.RegisterRenderPass("TAA", Core::RENDER_NODE_TYPE::COMPUTE, Core::MakeDelegate(&taa, &RenderPass::TAA::Render));
                g.RegisterResource(comp.GetOutput(RenderPass::Compositing::SHADER_OUT_RES::COMPOSITED), R_COMP);
                g.RegisterResource(nullptr, R_TAA);       // ping-pong target: identified by its path id, like the reference's dummy resources
            }
            Core::RenderNodeHandle hAe{}, hDisp{};
            if (display)
            {
                hAe = g.RegisterRenderPass("AutoExposure", Core::RENDER_NODE_TYPE::COMPUTE, Core::MakeDelegate(&aeNode, &AeNode::Render));
                hDisp = g.RegisterRenderPass("Display", Core::RENDER_NODE_TYPE::RENDER, Core::MakeDelegate(&dispNode, &DispNode::Render));
                g.RegisterResource(ae.GetOutput(RenderPass::AutoExposure::SHADER_OUT_RES::EXPOSURE), R_EXPOSURE);
                g.RegisterResource(disp.GetOutput(RenderPass::DisplayPass::SHADER_OUT_RES::BACK_BUFFER_LINEAR), R_BACKBUFFER);
            }
            g.MoveToPostRegister();
            g.AddOutput(hSky, R_LUT, Core::STATE_UNORDERED_ACCESS);
            g.AddOutput(hGB, R_GBUF, Core::STATE_UNORDERED_ACCESS);
            g.AddInput(hSdi, R_LUT, Core::STATE_SHADER_READ); g.AddInput(hSdi, R_GBUF, Core::STATE_SHADER_READ); g.AddOutput(hSdi, R_SDI, Core::STATE_UNORDERED_ACCESS);
            g.AddInput(hInd, R_LUT, Core::STATE_SHADER_READ); g.AddInput(hInd, R_GBUF, Core::STATE_SHADER_READ); g.AddOutput(hInd, R_IND, Core::STATE_UNORDERED_ACCESS);
            if (post)
            {
                g.AddInput(hComp, R_SDI, Core::STATE_SHADER_READ); g.AddInput(hComp, R_IND, Core::STATE_SHADER_READ); g.AddInput(hComp, R_GBUF, Core::STATE_SHADER_READ);
                g.AddOutput(hComp, R_COMP, Core::STATE_UNORDERED_ACCESS);
                g.AddInput(hTaa, R_COMP, Core::STATE_SHADER_READ); g.AddInput(hTaa, R_GBUF, Core::STATE_SHADER_READ); g.AddOutput(hTaa, R_TAA, Core::STATE_UNORDERED_ACCESS);
            }
            if (display)
            {
                g.AddInput(hAe, R_TAA, Core::STATE_SHADER_READ); g.AddOutput(hAe, R_EXPOSURE, Core::STATE_UNORDERED_ACCESS);
                g.AddInput(hDisp, R_TAA, Core::STATE_SHADER_READ); g.AddInput(hDisp, R_EXPOSURE, Core::STATE_SHADER_READ); g.AddOutput(hDisp, R_BACKBUFFER, Core::STATE_UNORDERED_ACCESS);
            }
            Support::TaskSet ts;
            g.Build(ts);
            ts.Run(true);
            g.WaitForFrame();
        }
        if (exposureOut && hipMemcpy(exposureOut, ae.GetOutput(RenderPass::AutoExposure::SHADER_OUT_RES::EXPOSURE), 8, hipMemcpyDeviceToHost) != hipSuccess) return -1;
        if (displayOut && hipMemcpy(displayOut, disp.GetOutput(RenderPass::DisplayPass::SHADER_OUT_RES::BACK_BUFFER_LINEAR), (size_t)w * h * 16, hipMemcpyDeviceToHost) != hipSuccess) return -1;
        if (displaySrgbOut && hipMemcpy(displaySrgbOut, disp.GetOutput(RenderPass::DisplayPass::SHADER_OUT_RES::BACK_BUFFER_SRGB8), (size_t)w * h * 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
        if (compositedOut && hipMemcpy(compositedOut, comp.GetOutput(RenderPass::Compositing::SHADER_OUT_RES::COMPOSITED), (size_t)w * h * 16, hipMemcpyDeviceToHost) != hipSuccess) return -1;
        if (taaOut && hipMemcpy(taaOut, taa.GetOutput(RenderPass::TAA::SHADER_OUT_RES::OUTPUT_A), (size_t)w * h * 8, hipMemcpyDeviceToHost) != hipSuccess) return -1;
        if (hipMemcpy(finalOut, ind.GetOutput(RenderPass::IndirectLighting::SHADER_OUT_RES::FINAL), (size_t)w * h * 16, hipMemcpyDeviceToHost) != hipSuccess) return -1;
        if (skyDiOut && hipMemcpy(skyDiOut, sdi.GetOutput(RenderPass::SkyDI::SHADER_OUT_RES::DENOISED), (size_t)w * h * 16, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    }
    zr_gbuffer_destroy(ctx.gbuffer);
    zr_scene_destroy(ctx.scene);
    return 0;
}

// GBuffer -> PreLighting -> IndirectLighting (ReSTIR PT) -> Denoise through the RenderGraph, n frames; finalOut = the indirect pass's FINAL and
// denoisedOut = the Denoise node's output (RGBA32F: rgb + variance) of the last frame
int zrh_render_sequence_denoise(const zr_scene_desc* desc, const zr_frame_constants* cbs, uint32_t n, uint32_t w, uint32_t h, uint32_t iterations, float* finalOut,
    float* denoisedOut)
{
    RenderPass::FrameContext ctx;
    ctx.device = 0; ctx.renderWidth = w; ctx.renderHeight = h;
    ZR_CHECK(zr_scene_create(0, desc, &ctx.scene));
    ZR_CHECK(zr_gbuffer_create(0, w, h, &ctx.gbuffer));
    {
        RenderPass::GBufferRT gb; RenderPass::PreLighting pre; RenderPass::IndirectLighting ind; RenderPass::Denoise dn;
        gb.Init(&ctx); pre.Init(&ctx); ind.Init(&ctx, RenderPass::IndirectLighting::INTEGRATOR::ReSTIR_PT); dn.Init(&ctx);
        dn.SetIterations(iterations);
        dn.SetCPUDescriptor(RenderPass::Denoise::SHADER_IN_CPU_DESC::SIGNAL, ind.GetOutput(RenderPass::IndirectLighting::SHADER_OUT_RES::FINAL));
        Core::RenderGraph g;
        enum : uint64_t { R_GBUF = 1, R_LIGHTS, R_IND, R_DENOISED };
        for (uint32_t f = 0; f < n; f++)
        {
            ctx.frameConstants = cbs[f];
            g.BeginFrame();
            auto hGB = g.RegisterRenderPass("GBuffer", Core::RENDER_NODE_TYPE::COMPUTE, Core::MakeDelegate(&gb, &RenderPass::GBufferRT::Render));
            auto hPre = g.RegisterRenderPass("PreLighting", Core::RENDER_NODE_TYPE::COMPUTE, Core::MakeDelegate(&pre, &RenderPass::PreLighting::Render));
            auto hInd = g.RegisterRenderPass("Indirect", Core::RENDER_NODE_TYPE::COMPUTE, Core::MakeDelegate(&ind, &RenderPass::IndirectLighting::Render));
            auto hDn = g.RegisterRenderPass("Denoise", Core::RENDER_NODE_TYPE::COMPUTE, Core::MakeDelegate(&dn, &RenderPass::Denoise::Render));
            g.RegisterResource(nullptr, R_GBUF); g.RegisterResource(nullptr, R_LIGHTS);
            g.RegisterResource(ind.GetOutput(RenderPass::IndirectLighting::SHADER_OUT_RES::FINAL), R_IND);
            g.RegisterResource(nullptr, R_DENOISED);      // ping-pong target, identified by its path id
            g.MoveToPostRegister();
            g.AddOutput(hGB, R_GBUF, Core::STATE_UNORDERED_ACCESS);
            g.AddOutput(hPre, R_LIGHTS, Core::STATE_UNORDERED_ACCESS);
            g.AddInput(hInd, R_GBUF, Core::STATE_SHADER_READ); g.AddInput(hInd, R_LIGHTS, Core::STATE_SHADER_READ); g.AddOutput(hInd, R_IND, Core::STATE_UNORDERED_ACCESS);
            g.AddInput(hDn, R_IND, Core::STATE_SHADER_READ); g.AddInput(hDn, R_GBUF, Core::STATE_SHADER_READ); g.AddOutput(hDn, R_DENOISED, Core::STATE_UNORDERED_ACCESS);
            Support::TaskSet ts;
            g.Build(ts);
            ts.Run(true);
            g.WaitForFrame();
        }
        if (hipMemcpy(finalOut, ind.GetOutput(RenderPass::IndirectLighting::SHADER_OUT_RES::FINAL), (size_t)w * h * 16, hipMemcpyDeviceToHost) != hipSuccess) return -1;
        if (hipMemcpy(denoisedOut, dn.GetOutput(RenderPass::Denoise::SHADER_OUT_RES::DENOISED), (size_t)w * h * 16, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    }
    zr_gbuffer_destroy(ctx.gbuffer);
    zr_scene_destroy(ctx.scene);
    return 0;
}

int zrh_render_sequence(const zr_scene_desc* desc, const zr_frame_constants* cbs, uint32_t n, uint32_t w, uint32_t h, int integrator, float* finalOut)
{ return zrh_render_sequence2(desc, cbs, n, w, h, integrator, finalOut, nullptr); }

int zrh_render_frame(const zr_scene_desc* desc, const zr_frame_constants* cb, uint32_t w, uint32_t h, float* finalOut)
{ return zrh_render_sequence(desc, cb, 1, w, h, 0, finalOut); }

} // extern "C"
