// zr_tu_probe.hip -- zr_device_probe: what THIS device delivers right now, measured in the process that is about to be timed (round 6).
//
// No reference counterpart (the reference reads adapter properties through DXGI, Source/ZetaCore/Core/Device.cpp, and never benchmarks the adapter).
// Why it exists: the same command line ran 18 % apart on two MI355X boxes of one pool (1.872 ms vs 2.203 ms per Cornell frame, the
// HBM-heavy kernels losing most), and a bench line without the device's own state cannot say whether a number moved because of the code or the box.
// bench.py prints this block as `device_state` before its timed region.
//
// Three probes, each >= `min_ms` of device time, hipEvent-timed on one stream:
//   copy   : device-to-device hipMemcpyAsync of a 512 MiB buffer -- (read + written bytes) / s: the HBM side
//   fma    : a grid that fills every SIMD 8 waves deep, every lane running 8 independent v_fma_f32 chains -- fp32 FMA issue rate: the shader clock under a
//            VALU load (the ReSTIR kernels are VALU-issue bound)
//   sclk   : the same kernel reads the shader-clock counter (s_memtime) and the constant-rate wall clock (s_memrealtime, 100 MHz on gfx9) at its
//            start and end: their ratio is the clock the waves actually ran at
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <string>
#include "../../include/zetaray_amd.h"

namespace zr { int DeviceProbeRun(int device, float min_ms, zr_device_probe* out, std::string& err); }

namespace {

constexpr int kChains = 8, kIters = 4096;

__global__ void __launch_bounds__(256) k_probe_fma(float* sink, unsigned long long* clocks, float a, float b)
{
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    float v[kChains];
    for (int k = 0; k < kChains; k++) v[k] = (float)(threadIdx.x + k) * 1.0e-3f;
    for (int i = 0; i < kIters; i++)
    {
#pragma unroll
        for (int k = 0; k < kChains; k++) v[k] = __builtin_fmaf(v[k], a, b);      // dependent on itself, independent of the other seven
    }
    float s = 0;
    for (int k = 0; k < kChains; k++) s += v[k];
    if (s == 123.456f) sink[0] = s;      // (never true for the a, b passed: keeps the chains alive)
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0) { clocks[0] = c1 - c0; clocks[1] = w1 - w0; }
}

#define PROBE_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { char m_[256]; snprintf(m_, sizeof(m_), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); err = m_; goto done; } } while (0)

} // namespace

int zr::DeviceProbeRun(int device, float min_ms, zr_device_probe* out, std::string& err)
{
    int rc = 1;
    void *a = nullptr, *b = nullptr; float* sink = nullptr; unsigned long long* clocks = nullptr;
    hipStream_t st = nullptr; hipEvent_t e0 = nullptr, e1 = nullptr;
    hipDeviceProp_t prop;
    const size_t bytes = (size_t)512 << 20;
    memset(out, 0, sizeof(*out));
    if (!(min_ms > 0)) min_ms = 50.0f;
    PROBE_TRY(hipSetDevice(device));
    PROBE_TRY(hipGetDeviceProperties(&prop, device));
    snprintf(out->name, sizeof(out->name), "%s", prop.name);
    snprintf(out->arch, sizeof(out->arch), "%s", prop.gcnArchName);
    out->compute_units = (uint32_t)prop.multiProcessorCount; out->clock_khz_max = (uint32_t)prop.clockRate; out->mem_clock_khz_max = (uint32_t)prop.memoryClockRate;
    out->mem_bus_bits = (uint32_t)prop.memoryBusWidth; out->l2_bytes = (uint32_t)prop.l2CacheSize; out->hbm_bytes = (uint64_t)prop.totalGlobalMem;
    out->wall_clock_khz = 100000;
    { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeWallClockRate, device) == hipSuccess && v > 0) out->wall_clock_khz = (uint32_t)v; }
    PROBE_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    PROBE_TRY(hipEventCreate(&e0)); PROBE_TRY(hipEventCreate(&e1));
    PROBE_TRY(hipMalloc(&a, bytes)); PROBE_TRY(hipMalloc(&b, bytes));
    PROBE_TRY(hipMalloc((void**)&sink, 256)); PROBE_TRY(hipMalloc((void**)&clocks, 16));
    PROBE_TRY(hipMemsetAsync(a, 1, bytes, st)); PROBE_TRY(hipMemsetAsync(b, 2, bytes, st));
    // ---- copy: one untimed pass, then passes until min_ms have gone by
    PROBE_TRY(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, st));
    PROBE_TRY(hipStreamSynchronize(st));
    {
        float total = 0; int n = 0;
        while (total < min_ms && n < 4096)
        {
            const int batch = n == 0 ? 4 : 8;
            PROBE_TRY(hipEventRecord(e0, st));
            for (int i = 0; i < batch; i++) PROBE_TRY(hipMemcpyAsync((i & 1) ? a : b, (i & 1) ? b : a, bytes, hipMemcpyDeviceToDevice, st));
            PROBE_TRY(hipEventRecord(e1, st));
            PROBE_TRY(hipEventSynchronize(e1));
            float ms = 0; PROBE_TRY(hipEventElapsedTime(&ms, e0, e1));
            total += ms; n += batch;
        }
        out->copy_GBs = (float)(2.0 * (double)bytes * n / (total * 1e-3) / 1e9); out->copy_ms = total;
    }
    // ---- fma: 8 waves per SIMD on every CU
    {
        const dim3 grid((uint32_t)prop.multiProcessorCount * 8u), block(256);
        hipLaunchKernelGGL(k_probe_fma, grid, block, 0, st, sink, clocks, 0.999f, 1.0e-3f);
        PROBE_TRY(hipGetLastError());
        PROBE_TRY(hipStreamSynchronize(st));
        float total = 0; int n = 0;
        while (total < min_ms && n < 65536)
        {
            const int batch = n == 0 ? 8 : 32;
            PROBE_TRY(hipEventRecord(e0, st));
            for (int i = 0; i < batch; i++) hipLaunchKernelGGL(k_probe_fma, grid, block, 0, st, sink, clocks, 0.999f, 1.0e-3f);
            PROBE_TRY(hipEventRecord(e1, st));
            PROBE_TRY(hipEventSynchronize(e1));
            float ms = 0; PROBE_TRY(hipEventElapsedTime(&ms, e0, e1));
            total += ms; n += batch;
        }
        const double flops = 2.0 * kChains * kIters * 256.0 * grid.x * n;
        out->fma_tflops = (float)(flops / (total * 1e-3) / 1e12); out->fma_ms = total;
        unsigned long long h[2] = {0, 0};
        PROBE_TRY(hipMemcpyAsync(h, clocks, sizeof(h), hipMemcpyDeviceToHost, st));
        PROBE_TRY(hipStreamSynchronize(st));
        if (h[1]) out->sclk_mhz_under_load = (float)((double)h[0] / (double)h[1] * (double)out->wall_clock_khz * 1e-3);
    }
    rc = 0;
done:
    if (a) (void)hipFree(a);
    if (b) (void)hipFree(b);
    if (sink) (void)hipFree(sink);
    if (clocks) (void)hipFree(clocks);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (st) (void)hipStreamDestroy(st);
    return rc;
}
