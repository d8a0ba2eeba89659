/*
 * zr_detmath.h -- the arithmetic contract of the zetaray_amd C-ABI.
 *
 * The reference (alipbcs/ZetaRay) evaluates sin/cos/exp/log/rsqrt/normalize/half conversions with whatever the
 * HLSL compiler and the GPU vendor provide; those are not bit-reproducible across devices, and there is no CPU
 * implementation to compare with.  This ABI pins them instead: every transcendental is built here from IEEE-754
 * basic operations (+ - * / sqrt fma, all correctly rounded on x86-64 SSE and on gfx950 with hipcc's default
 * correctly-rounded divide/sqrt) using the public-domain Cephes single-precision kernels (S. Moshier), so the
 * HIP kernels and the CPU oracle produce identical bits.  Both sides must be compiled with -ffp-contract=off
 * (fused multiply-adds appear only where zr_fma is written, which is where the HLSL source says `mad`).
 *
 * This header is part of the interface spec (like zr_wire.h), not of the oracle.
 *
 * Tolerance mode (-DZR_ARITH_FAST, device code only; `make ARITH=fast` builds libzetaray_amd_fast.so): the same names map to the gfx950
 * hardware approximations (v_rcp / v_rsq / v_sqrt / v_exp / v_log / v_sin / v_cos, ~1 ulp, denormal results not guaranteed) and the
 * translation units are compiled with contracted FMAs and the 2.5-ulp divide.  That build is NOT bit-exact against the oracle; its
 * parity bar is the one BASELINE.json's north_star states (per-pixel L2 on radiance, integer reservoir state equal where no decision
 * flips): tests/test_fast_arith.py.  The contract build stays the default and the only one the bit-exact tests load.
 */
#ifndef ZR_DETMATH_H
#define ZR_DETMATH_H

#include <stdint.h>

#if defined(__HIPCC__)
/* always_inline: a real call on the GPU moves every live VGPR and every by-reference argument through scratch memory
   (measured: ~3 GB of scratch traffic per 1080p ReSTIR PT reconnect pass), so kernels are compiled flat */
#define ZR_HD __host__ __device__ inline __attribute__((always_inline))
#define ZR_HDM __host__ __device__ __attribute__((always_inline))          /* member functions */
#define ZR_HD_FLAT ZR_HD
#define ZR_UNROLL _Pragma("unroll")      /* full unrolling: two-slot arrays indexed by the loop counter become registers */
#else
#define ZR_HD static inline
#define ZR_HDM
#define ZR_HD_FLAT static inline
#define ZR_UNROLL
#endif

#define ZR_PI              3.141592654f
#define ZR_TWO_PI          6.283185307f
#define ZR_PI_OVER_2       1.570796327f
#define ZR_PI_OVER_4       0.7853981635f
#define ZR_ONE_OVER_PI     0.318309886f
#define ZR_ONE_OVER_2_PI   0.159154943f
#define ZR_ONE_OVER_4_PI   0.079577472f
#define ZR_FLT_MAX         3.402823466e+38f
#define ZR_FLT16_MAX       65504.0f

ZR_HD uint32_t zr_asuint(float f) { union { float f; uint32_t u; } c; c.f = f; return c.u; }
ZR_HD float    zr_asfloat(uint32_t u) { union { float f; uint32_t u; } c; c.u = u; return c.f; }

ZR_HD float zr_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
#if defined(ZR_ARITH_FAST) && defined(__HIP_DEVICE_COMPILE__)
#define ZR_FAST_DEV 1
#else
#define ZR_FAST_DEV 0
#endif
#if ZR_FAST_DEV
ZR_HD float zr_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
#else
ZR_HD float zr_sqrt(float x) { return __builtin_sqrtf(x); }
#endif
ZR_HD float zr_abs(float x) { return zr_asfloat(zr_asuint(x) & 0x7fffffffu); }
/* HLSL min/max: a comparison that is false for NaN returns the second operand */
ZR_HD float zr_max(float a, float b) { return a > b ? a : b; }
ZR_HD float zr_min(float a, float b) { return a < b ? a : b; }
ZR_HD float zr_saturate(float x) { return x > 0.0f ? (x < 1.0f ? x : 1.0f) : 0.0f; }   /* NaN -> 0 */
ZR_HD float zr_clamp(float x, float lo, float hi) { return zr_min(zr_max(x, lo), hi); }
#if ZR_FAST_DEV
ZR_HD float zr_rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }
ZR_HD float zr_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
#else
ZR_HD float zr_rsqrt(float x) { return 1.0f / zr_sqrt(x); }
ZR_HD float zr_rcp(float x) { return 1.0f / x; }
#endif
ZR_HD float zr_sign(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }
ZR_HD float zr_floor(float x) { return __builtin_floorf(x); }
ZR_HD int   zr_isnan(float x) { return (zr_asuint(x) & 0x7fffffffu) > 0x7f800000u; }
ZR_HD int   zr_isinf(float x) { return (zr_asuint(x) & 0x7fffffffu) == 0x7f800000u; }
/* Math::Lerp (reference Math.hlsli:66-70): mad(t, v1, mad(-t, v0, v0)) */
ZR_HD float zr_lerp_mad(float v0, float v1, float t) { return zr_fma(t, v1, zr_fma(-t, v0, v0)); }
/* HLSL intrinsic lerp */
ZR_HD float zr_lerp(float a, float b, float t) { return a + t * (b - a); }

/* 2^n for integer n, exact; handles the denormal range by splitting */
ZR_HD float zr_ldexp(float x, int n)
{
    if (n > 127) { x *= zr_asfloat(0x7f000000u); n -= 127; if (n > 127) n = 127; }
    else if (n < -126) { x *= zr_asfloat(0x00800000u); n += 126; if (n < -126) n = -126; }
    return x * zr_asfloat((uint32_t)(n + 127) << 23);
}

/* Cephes sinf/cosf kernel.  Valid (|err| ~ 1 ulp) for |x| < 8192; larger arguments never occur on this path. */
#if ZR_FAST_DEV
/* v_sin_f32 / v_cos_f32 take revolutions and are valid on [-256, 256]: reduce with v_fract_f32 first */
ZR_HD void zr_sincos(float xx, float* s, float* c)
{
    const float r = __builtin_amdgcn_fractf(xx * ZR_ONE_OVER_2_PI);
    *s = __builtin_amdgcn_sinf(r);
    *c = __builtin_amdgcn_cosf(r);
}
#else
ZR_HD void zr_sincos(float xx, float* s, float* c)
{
    const float DP1 = 0.78515625f, DP2 = 2.4187564849853515625e-4f, DP3 = 3.77489497744594108e-8f;
    const float FOPI = 1.27323954473516f;
    float x = zr_abs(xx);
    int sgn_s = (xx < 0.0f) ? -1 : 1;
    int sgn_c = 1;
    int j = (int)(FOPI * x);
    float y = (float)j;
    if (j & 1) { j += 1; y += 1.0f; }
    j &= 7;
    if (j > 3) { sgn_s = -sgn_s; sgn_c = -sgn_c; j -= 4; }
    if (j > 1) sgn_c = -sgn_c;
    x = ((x - y * DP1) - y * DP2) - y * DP3;
    float z = x * x;
    float ps = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * x + x;
    float pc = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z
               - 0.5f * z + 1.0f;
    float sv, cv;
    if (j == 1 || j == 2) { sv = pc; cv = ps; } else { sv = ps; cv = pc; }
    *s = sgn_s < 0 ? -sv : sv;
    *c = sgn_c < 0 ? -cv : cv;
}
#endif
ZR_HD float zr_sin(float x) { float s, c; zr_sincos(x, &s, &c); return s; }
ZR_HD float zr_cos(float x) { float s, c; zr_sincos(x, &s, &c); return c; }

#if ZR_FAST_DEV
ZR_HD float zr_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }      /* v_exp_f32 is 2^x */
ZR_HD float zr_log(float x) { return __builtin_amdgcn_logf(x) * 0.693147180559945309f; }      /* v_log_f32 is log2 */
ZR_HD float zr_log2(float x) { return __builtin_amdgcn_logf(x); }
ZR_HD float zr_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
ZR_HD float zr_pow(float x, float y) { return __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x)); }
#else
/* Cephes expf */
ZR_HD float zr_exp(float x)
{
    if (zr_isnan(x)) return x;
    if (x > 88.72283905206835f) return zr_asfloat(0x7f800000u);
    if (x < -103.278929903431851103f) return 0.0f;
    float z = zr_floor(1.44269504088896341f * x + 0.5f);
    x -= z * 0.693359375f;
    x -= z * -2.12194440e-4f;
    int n = (int)z;
    z = x * x;
    z = (((((1.9875691500e-4f * x + 1.3981999507e-3f) * x + 8.3334519073e-3f) * x + 4.1665795894e-2f) * x
          + 1.6666665459e-1f) * x + 5.0000001201e-1f) * z + x + 1.0f;
    return zr_ldexp(z, n);
}

/* Cephes logf (natural log) */
ZR_HD float zr_log(float xx)
{
    if (zr_isnan(xx)) return xx;
    if (xx < 0.0f) return zr_asfloat(0x7fc00000u);
    if (xx == 0.0f) return zr_asfloat(0xff800000u);
    if (zr_isinf(xx)) return xx;
    uint32_t u = zr_asuint(xx);
    int e = 0;
    if ((u & 0x7f800000u) == 0) { xx *= 8388608.0f; u = zr_asuint(xx); e = -23; }   /* denormal */
    e += (int)((u >> 23) & 0xffu) - 126;
    float x = zr_asfloat((u & 0x007fffffu) | 0x3f000000u);                           /* [0.5, 1) */
    if (x < 0.707106781186547524f) { e -= 1; x = x + x - 1.0f; } else { x = x - 1.0f; }
    float z = x * x;
    float y = ((((((((7.0376836292e-2f * x - 1.1514610310e-1f) * x + 1.1676998740e-1f) * x - 1.2420140846e-1f) * x
                 + 1.4249322787e-1f) * x - 1.6668057665e-1f) * x + 2.0000714765e-1f) * x - 2.4999993993e-1f) * x
               + 3.3333331174e-1f) * x * z;
    float fe = (float)e;
    y += -2.12194440e-4f * fe;
    y += -0.5f * z;
    z = x + y;
    z += 0.693359375f * fe;
    return z;
}
ZR_HD float zr_log2(float x) { return zr_log(x) * 1.44269504088896341f; }
ZR_HD float zr_exp2(float x) { return zr_exp(x * 0.693147180559945309f); }
/* HLSL pow(x, y) = exp2(y * log2(x)); x <= 0 follows that definition (log of 0 -> -inf) */
ZR_HD float zr_pow(float x, float y) { return zr_exp(y * zr_log(x)); }
#endif

/* Cephes atanf */
ZR_HD float zr_atan(float xx)
{
    float x = zr_abs(xx), y;
    if (x > 2.414213562373095f) { y = 1.5707963267948966192f; x = -(1.0f / x); }
    else if (x > 0.4142135623730950f) { y = 0.7853981633974483096f; x = (x - 1.0f) / (x + 1.0f); }
    else y = 0.0f;
    float z = x * x;
    y += (((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * x + x;
    return xx < 0.0f ? -y : y;
}
ZR_HD float zr_atan2(float y, float x)
{
    const float PI = 3.14159265358979323846f, PIO2 = 1.5707963267948966192f;
    if (x == 0.0f) { if (y == 0.0f) return 0.0f; return y > 0.0f ? PIO2 : -PIO2; }
    if (y == 0.0f) return x > 0.0f ? 0.0f : PI;
    float w = 0.0f;
    if (x < 0.0f) w = (y < 0.0f) ? -PI : PI;
    return w + zr_atan(y / x);
}

/* unsigned small float (5-bit exponent, `mbits` mantissa bits: the channels of R11G11B10_FLOAT) -> fp32, exact */
ZR_HD float zr_unpack_ufloat(uint32_t bits, int mbits)
{
    const uint32_t e = bits >> mbits, m = bits & ((1u << mbits) - 1u);
    if (e == 0) return (float)m * zr_asfloat((uint32_t)(127 - 14 - mbits) << 23);
    if (e == 31) return zr_asfloat(0x7f800000u | (m << (23 - mbits)));
    return zr_asfloat(((e + 112u) << 23) | (m << (23 - mbits)));
}
/* D3D ftou / ftoi: NaN -> 0, out of range saturates (x86 cvttss2si and C++ casts do something else) */
ZR_HD uint32_t zr_f2u_sat(float f) { if (zr_isnan(f) || f <= 0.0f) return 0u; if (f >= 4294967296.0f) return 0xffffffffu; return (uint32_t)f; }
ZR_HD int32_t zr_f2i_sat(float f) { if (zr_isnan(f)) return 0; if (f >= 2147483648.0f) return 2147483647; if (f <= -2147483648.0f) return (-2147483647 - 1); return (int32_t)f; }

/* fp32 -> fp16, round-to-nearest-even, full denormal/inf/nan handling (== v_cvt_f16_f32 / F16C) */
ZR_HD uint16_t zr_f32_to_f16_portable(float f)
{
    uint32_t x = zr_asuint(f);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t ax = x & 0x7fffffffu;
    if (ax >= 0x7f800000u) return (uint16_t)(sign | (ax > 0x7f800000u ? 0x7e00u : 0x7c00u));
    if (ax >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);           /* rounds to >= 65520 -> inf */
    if (ax < 0x38800000u)                                                /* result is a half denormal (or 0) */
    {
        if (ax < 0x33000000u) return (uint16_t)sign;                     /* < 2^-25 -> 0 */
        uint32_t e = ax >> 23;
        uint32_t m = (ax & 0x007fffffu) | 0x00800000u;
        uint32_t shift = 126u - e;                                       /* 14..24 */
        uint32_t r = m >> shift;
        uint32_t rem = m & ((1u << shift) - 1u);
        uint32_t half = 1u << (shift - 1u);
        if (rem > half || (rem == half && (r & 1u))) r++;
        return (uint16_t)(sign | r);
    }
    uint32_t r = ((ax - 0x38000000u) >> 13);
    uint32_t rem = ax & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) r++;
    return (uint16_t)(sign | r);
}

ZR_HD float zr_f16_to_f32_portable(uint16_t h)
{
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1fu;
    uint32_t m = h & 0x3ffu;
    if (e == 0)
    {
        if (m == 0) return zr_asfloat(sign);
        /* denormal: m * 2^-24, exact in fp32 */
        float v = (float)m * 5.9604644775390625e-8f;
        return zr_asfloat(zr_asuint(v) | sign);
    }
    if (e == 31) return zr_asfloat(sign | 0x7f800000u | (m << 13));
    return zr_asfloat(sign | ((e + 112u) << 23) | (m << 13));
}
/* gfx950 kernels use the conversion instructions (v_cvt_f16_f32 / v_cvt_f32_f16: round-to-nearest-even, fp16 denormals kept) for
   everything but NaN (and Inf on the way up), where the portable code's canonical results are kept.  zr_selftest_half_conversions
   (include/zetaray_amd.h) compares the two paths on the device for every fp32 and every fp16 bit pattern. */
ZR_HD uint16_t zr_f32_to_f16(float f)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if (f != f) return (uint16_t)(((zr_asuint(f) >> 16) & 0x8000u) | 0x7e00u);
    union { _Float16 h; uint16_t u; } c; c.h = (_Float16)f;
    return c.u;
#else
    return zr_f32_to_f16_portable(f);
#endif
}
ZR_HD float zr_f16_to_f32(uint16_t h)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if ((h & 0x7c00u) == 0x7c00u) return zr_asfloat((((uint32_t)h & 0x8000u) << 16) | 0x7f800000u | (((uint32_t)h & 0x3ffu) << 13));
    union { _Float16 h; uint16_t u; } c; c.u = h;
    return (float)c.h;
#else
    return zr_f16_to_f32_portable(h);
#endif
}
/* UNORM8 / UNORM16 -> float: x / 255 and x / 65535 for integer-valued x in [0, 65535].  IEEE division costs ~10 instructions on gfx950;
   device code uses a reciprocal multiply with one fma correction step, which returns the correctly rounded quotient for every such x
   (checked on the device against the division for all 65536 inputs by zr_selftest_half_conversions). */
ZR_HD float zr_div255(float x)
{
#if ZR_FAST_DEV
    return x * (1.0f / 255.0f);
#elif defined(__HIP_DEVICE_COMPILE__)
    const float r = 1.0f / 255.0f;
    const float q = x * r;
    return __builtin_fmaf(__builtin_fmaf(-q, 255.0f, x), r, q);
#else
    return x / 255.0f;
#endif
}
ZR_HD float zr_div65535(float x)
{
#if ZR_FAST_DEV
    return x * (1.0f / 65535.0f);
#elif defined(__HIP_DEVICE_COMPILE__)
    const float r = 1.0f / 65535.0f;
    const float q = x * r;
    return __builtin_fmaf(__builtin_fmaf(-q, 65535.0f, x), r, q);
#else
    return x / 65535.0f;
#endif
}
/* round-trip through half, the effect of an HLSL (half) cast followed by (float) */
ZR_HD float zr_round_f16(float f) { return zr_f16_to_f32(zr_f32_to_f16(f)); }

/* PCG hash family, reference Source/ZetaRenderPass/Common/Sampling.hlsli:12-49 (integer; bit-exact by nature) */
ZR_HD uint32_t zr_pcg(uint32_t x)
{
    uint32_t state = x * 747796405u + 2891336453u;
    uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}
ZR_HD void zr_pcg3d(uint32_t* x, uint32_t* y, uint32_t* z)
{
    uint32_t vx = *x * 1664525u + 1013904223u, vy = *y * 1664525u + 1013904223u, vz = *z * 1664525u + 1013904223u;
    vx += vy * vz; vy += vz * vx; vz += vx * vy;
    vx ^= vx >> 16u; vy ^= vy >> 16u; vz ^= vz >> 16u;
    vx += vy * vz; vy += vz * vx; vz += vx * vy;
    *x = vx; *y = vy; *z = vz;
}
ZR_HD void zr_pcg4d(uint32_t* x, uint32_t* y, uint32_t* z, uint32_t* w)
{
    uint32_t vx = *x * 1664525u + 1013904223u, vy = *y * 1664525u + 1013904223u;
    uint32_t vz = *z * 1664525u + 1013904223u, vw = *w * 1664525u + 1013904223u;
    vx += vy * vw; vy += vz * vx; vz += vx * vy; vw += vy * vz;
    vx ^= vx >> 16u; vy ^= vy >> 16u; vz ^= vz >> 16u; vw ^= vw >> 16u;
    vx += vy * vw; vy += vz * vx; vz += vx * vy; vw += vy * vz;
    *x = vx; *y = vy; *z = vz; *w = vw;
}

#endif /* ZR_DETMATH_H */
