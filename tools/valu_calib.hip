// VALU / SALU issue-rate calibration for gfx950: what is the peak wave64 instruction rate of a SIMD, and what do
// SQ_INSTS_VALU / SQ_ACTIVE_INST_VALU / SQ_THREAD_CYCLES_VALU read per instruction?  Every kernel is a loop of UNROLL x 8 independent
// chains of ONE instruction (inline asm, so the count per wave is exact), launched with 1, 2, 4 and 8 waves per SIMD.
//   build:  hipcc --offload-arch=gfx950 -O2 -o build/valu_calib tools/valu_calib.hip
//   run:    build/valu_calib            prints one JSON line per (instruction, waves/SIMD): wave-instructions/s from hipEvents
//           rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU ... -- build/valu_calib     the counters per kernel name
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define ITER 2048
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

enum Op { FMA, MUL_F32, ADD_U32, MUL_LO_U32, MAD_U64_U32, RCP, SQRT, CNDMASK, MIN3, CVT_F16, FMA_LO32, FMA_EVEN, FMA_ONE, FMA_LO16, SALU_ADD, DS_RW,
          MOV, CNDMASK_SGPR, CMP_CNDMASK, CMP, AND_B32, LSHL, MAX_F32, BFE, FMA_K, N_OPS };
static const char* kNames[N_OPS] = { "v_fma_f32", "v_mul_f32", "v_add_u32", "v_mul_lo_u32", "v_mad_u64_u32", "v_rcp_f32", "v_sqrt_f32", "v_cndmask_b32",
    "v_min3_f32", "v_cvt_f16_f32", "v_fma_f32@lanes0-31", "v_fma_f32@even_lanes", "v_fma_f32@1lane", "v_fma_f32@lanes0-15", "s_add_u32", "ds_write_b64+ds_read_b64",
    "v_mov_b32", "v_cndmask_b32(sgpr mask)", "v_cmp_lt_f32+v_cndmask_b32", "v_cmp_lt_f32", "v_and_b32", "v_lshlrev_b32", "v_max_f32", "v_bfe_u32", "v_fma_f32(2 vgpr + const)" };

#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)

template<int OP, int W>
__global__ __launch_bounds__(256) void k_calib(float* out, float x, float y)
{
    float a0 = x + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const unsigned lane = threadIdx.x & 63u;
    bool on = true;
    if (OP == FMA_LO32) on = lane < 32;
    if (OP == FMA_EVEN) on = (lane & 1) == 0;
    if (OP == FMA_ONE) on = lane == 0;
    if (OP == FMA_LO16) on = lane < 16;
    if (on)
    {
        if constexpr (OP == SALU_ADD)
        {
            // 32 s_add_u32 on four fixed scalar registers per iteration, in one asm block (the register allocator never sees the chains)
            for (int i = 0; i < ITER; i++)
                asm volatile(
                    "s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n"
                    "s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n"
                    "s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n"
                    "s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n"
                    ::: "s20", "s21", "s22", "s23", "scc");
        }
        else if constexpr (OP == DS_RW)
        {
            __shared__ unsigned long long st[8][256];
            unsigned long long v = (unsigned long long)threadIdx.x;
            for (int i = 0; i < ITER; i++)
            {
#define S(n) st[n][threadIdx.x] = v + n;
                REP8(S)
#undef S
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#define S(n) v += st[n][threadIdx.x];
                REP8(S)
#undef S
            }
            a0 = (float)v;
        }
        else
        {
            unsigned u0 = __float_as_uint(a0), u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3, u4 = u0 + 4, u5 = u0 + 5, u6 = u0 + 6, u7 = u0 + 7;
            unsigned long long q0 = u0, q1 = u1, q2 = u2, q3 = u3, q4 = u4, q5 = u5, q6 = u6, q7 = u7;
            for (int i = 0; i < ITER; i++)
            {
                if constexpr (OP == FMA || OP == FMA_LO32 || OP == FMA_EVEN || OP == FMA_ONE || OP == FMA_LO16) {
#define S(n) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a##n) : "v"(x), "v"(y));
                    REP8(S) REP8(S) REP8(S) REP8(S)
#undef S
                } else if constexpr (OP == MUL_F32) {
#define S(n) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a##n) : "v"(x));
                    REP8(S) REP8(S) REP8(S) REP8(S)
#undef S
                } else if constexpr (OP == ADD_U32) {
#define S(n) asm volatile("v_add_u32 %0, %1, %0" : "+v"(u##n) : "v"(u0 | 1u));
                    REP8(S) REP8(S) REP8(S) REP8(S)
#undef S
                } else if constexpr (OP == MUL_LO_U32) {
#define S(n) asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(u##n) : "v"(lane | 1u));
                    REP8(S) REP8(S) REP8(S) REP8(S)
#undef S
                } else if constexpr (OP == MAD_U64_U32) {
#define S(n) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q##n) : "v"(lane | 1u), "v"(u0) : "vcc");
                    REP8(S) REP8(S) REP8(S) REP8(S)
#undef S
                } else if constexpr (OP == RCP) {
#define S(n) asm volatile("v_rcp_f32 %0, %0" : "+v"(a##n));
                    REP8(S) REP8(S) REP8(S) REP8(S)
#undef S
                } else if constexpr (OP == SQRT) {
#define S(n) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a##n));
                    REP8(S) REP8(S) REP8(S) REP8(S)
#undef S
                } else if constexpr (OP == CNDMASK) {
#define S(n) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a##n) : "v"(x) : );
                    REP8(S) REP8(S) REP8(S) REP8(S)
#undef S
                } else if constexpr (OP == MIN3) {
#define S(n) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a##n) : "v"(x), "v"(y));
                    REP8(S) REP8(S) REP8(S) REP8(S)
#undef S
                } else if constexpr (OP == CVT_F16) {
#define S(n) asm volatile("v_cvt_f16_f32 %0, %0" : "+v"(a##n));
                    REP8(S) REP8(S) REP8(S) REP8(S)
#undef S
                } else if constexpr (OP == MOV) {
#define S(n) asm volatile("v_mov_b32 %0, %1" : "+v"(a##n) : "v"(x));
                    REP8(S) REP8(S) REP8(S) REP8(S)
#undef S
                } else if constexpr (OP == CNDMASK_SGPR) {
#define S(n) asm volatile("v_cndmask_b32 %0, %0, %1, s[20:21]" : "+v"(a##n) : "v"(x) : "s20", "s21");
                    REP8(S) REP8(S) REP8(S) REP8(S)
#undef S
                } else if constexpr (OP == CMP_CNDMASK) {      // 16 compares + 16 selects per 32 instructions
#define S(n) asm volatile("v_cmp_lt_f32 vcc, %1, %0\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a##n) : "v"(x) : "vcc");
                    REP8(S) REP8(S)
#undef S
                } else if constexpr (OP == CMP) {
#define S(n) asm volatile("v_cmp_lt_f32 vcc, %1, %0" : : "v"(a##n), "v"(x) : "vcc");
                    REP8(S) REP8(S) REP8(S) REP8(S)
#undef S
                } else if constexpr (OP == AND_B32) {
#define S(n) asm volatile("v_and_b32 %0, %1, %0" : "+v"(u##n) : "v"(u0 | 0xffff0000u));
                    REP8(S) REP8(S) REP8(S) REP8(S)
#undef S
                } else if constexpr (OP == LSHL) {
#define S(n) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(u##n));
                    REP8(S) REP8(S) REP8(S) REP8(S)
#undef S
                } else if constexpr (OP == MAX_F32) {
#define S(n) asm volatile("v_max_f32 %0, %1, %0" : "+v"(a##n) : "v"(x));
                    REP8(S) REP8(S) REP8(S) REP8(S)
#undef S
                } else if constexpr (OP == BFE) {
#define S(n) asm volatile("v_bfe_u32 %0, %0, 1, 31" : "+v"(u##n));
                    REP8(S) REP8(S) REP8(S) REP8(S)
#undef S
                } else if constexpr (OP == FMA_K) {
#define S(n) asm volatile("v_fma_f32 %0, %0, %1, 1.0" : "+v"(a##n) : "v"(x));
                    REP8(S) REP8(S) REP8(S) REP8(S)
#undef S
                }
            }
            a0 += __uint_as_float(u0 ^ u1 ^ u2 ^ u3 ^ u4 ^ u5 ^ u6 ^ u7) + (float)(q0 + q1 + q2 + q3 + q4 + q5 + q6 + q7);
        }
    }
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678f) out[threadIdx.x] = a0;   // never true; keeps the chains alive
}

// v_fma_f32 / v_mul_f32 with only lanes [0, nlanes) of every wave active (runtime mask): does a partly filled wave issue faster -- or slower?
template<int MUL>
__global__ __launch_bounds__(256) void k_lanes(float* out, float x, float y, unsigned nlanes)
{
    float a0 = x + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    if ((threadIdx.x & 63u) < nlanes)
        for (int i = 0; i < ITER; i++)
        {
            if constexpr (MUL) {
#define S(n) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a##n) : "v"(x));
                REP8(S) REP8(S) REP8(S) REP8(S)
#undef S
            } else {
#define S(n) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a##n) : "v"(x), "v"(y));
                REP8(S) REP8(S) REP8(S) REP8(S)
#undef S
            }
        }
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678f) out[threadIdx.x] = a0;
}
template<int MUL> static void RunLanes(float* out, int cus)
{
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const unsigned lanes[] = {1, 2, 4, 8, 15, 16, 17, 24, 31, 32, 33, 48, 63, 64};
    for (int W : {1, 2, 4, 8})
        for (unsigned nl : lanes)
        {
            const int blocks = cus * W;
            std::vector<float> ms;
            for (int r = 0; r < 5; r++)
            {
                CHK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL((k_lanes<MUL>), dim3(blocks), dim3(256), 0, 0, out, 1.0000001f, 1e-9f, nl);
                CHK(hipEventRecord(e1, 0)); CHK(hipEventSynchronize(e1));
                float t; CHK(hipEventElapsedTime(&t, e0, e1)); ms.push_back(t);
            }
            std::sort(ms.begin(), ms.end());
            const double rate = 32.0 * ITER * blocks * 4.0 / (ms[2] * 1e-3);
            std::printf("{\"inst\": \"%s\", \"active_lanes\": %u, \"waves_per_simd\": %d, \"ms\": %.4f, \"cycles_per_inst_per_simd_at_2p4GHz\": %.3f}\n",
                MUL ? "v_mul_f32" : "v_fma_f32", nl, W, ms[2], 2.4e9 * cus * 4.0 / rate);
        }
}

template<int OP, int W> static void Run(float* out, int cus)
{
    const int blocks = cus * W;      // a 256-thread block = one wave per SIMD of a CU; W blocks per CU
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int i = 0; i < 2; i++) hipLaunchKernelGGL((k_calib<OP, W>), dim3(blocks), dim3(256), 0, 0, out, 1.0000001f, 1e-9f);
    CHK(hipDeviceSynchronize());
    std::vector<float> ms;
    for (int r = 0; r < 5; r++)
    {
        CHK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k_calib<OP, W>), dim3(blocks), dim3(256), 0, 0, out, 1.0000001f, 1e-9f);
        CHK(hipEventRecord(e1, 0)); CHK(hipEventSynchronize(e1));
        float t; CHK(hipEventElapsedTime(&t, e0, e1)); ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    const double perWave = (OP == DS_RW ? 16.0 : 32.0) * ITER;
    const double insts = perWave * blocks * 4.0;
    const double rate = insts / (ms[2] * 1e-3);
    std::printf("{\"inst\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"wave_insts\": %.0f, \"ginst_per_s\": %.1f, \"cycles_per_inst_per_simd_at_2p4GHz\": %.3f}\n",
        kNames[OP], W, ms[2], insts, rate / 1e9, 2.4e9 * cus * 4.0 / rate);
}

template<int OP> static void RunAll(float* out, int cus) { Run<OP, 1>(out, cus); Run<OP, 2>(out, cus); Run<OP, 4>(out, cus); Run<OP, 8>(out, cus); }

int main()
{
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    std::printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d}\n", p.name, cus, p.clockRate / 1000);
    float* out; CHK(hipMalloc(&out, 4096));
    RunAll<FMA>(out, cus); RunAll<MUL_F32>(out, cus); RunAll<ADD_U32>(out, cus); RunAll<MUL_LO_U32>(out, cus); RunAll<MAD_U64_U32>(out, cus);
    RunAll<RCP>(out, cus); RunAll<SQRT>(out, cus); RunAll<CNDMASK>(out, cus); RunAll<MIN3>(out, cus); RunAll<CVT_F16>(out, cus);
    RunAll<FMA_LO32>(out, cus); RunAll<FMA_EVEN>(out, cus); RunAll<FMA_ONE>(out, cus); RunAll<FMA_LO16>(out, cus);
    RunAll<SALU_ADD>(out, cus); RunAll<DS_RW>(out, cus);
    RunAll<MOV>(out, cus); RunAll<CNDMASK_SGPR>(out, cus); RunAll<CMP_CNDMASK>(out, cus); RunAll<CMP>(out, cus); RunAll<AND_B32>(out, cus); RunAll<LSHL>(out, cus);
    RunAll<MAX_F32>(out, cus); RunAll<BFE>(out, cus); RunAll<FMA_K>(out, cus);
    RunLanes<0>(out, cus); RunLanes<1>(out, cus);
    CHK(hipFree(out));
    return 0;
}
