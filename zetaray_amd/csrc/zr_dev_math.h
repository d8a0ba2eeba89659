// zr_dev_math.h -- device-side vector math, packing, RNG and sampling warps for the HIP kernels.
//
// MI355X-native restatement of the shared shader library of the reference:
//   Source/ZetaRenderPass/Common/Math.hlsli   (packing, ONB, TRS, tri differentials; file:line cited per function)
//   Source/ZetaRenderPass/Common/Sampling.hlsli (PCG RNG, warps)
//   Source/ZetaRenderPass/Common/RT.hlsli      (camera rays, self-intersection offset, MIS heuristics)
// Every function is ZR_HD (__host__ __device__) so tests/hostexec can run the per-item stage functions on the CPU
// without a GPU; the product library only ever launches them inside __global__ kernels.
// Arithmetic contract: include/zr_detmath.h, compiled with -ffp-contract=off (fma only where `mad` is written).
#pragma once
#include <stdint.h>
#include "../../include/zr_detmath.h"
#include "../../include/zr_wire.h"

namespace zr {

struct V2      // member-wise copies: see V3
{
    float x, y;
    ZR_HDM V2() = default;
    ZR_HDM V2(const V2& o) : x(o.x), y(o.y) {}
    ZR_HDM V2& operator=(const V2& o) { x = o.x; y = o.y; return *this; }
};
// V3 has user-provided (member-wise) copy operations on purpose: for a trivially copyable struct clang lowers `c ? a : b` on V3
// lvalues to a select of the two ADDRESSES followed by a 12-byte memcpy, which SROA cannot split -- both operands and the result then
// stay in scratch memory (32 such temporaries in k_rpt_temporal / k_rpt_stc).  With member-wise copies the select is over loaded
// scalars and everything is promoted to registers.
struct V3
{
    float x, y, z;
    ZR_HDM V3() = default;
    ZR_HDM V3(const V3& o) : x(o.x), y(o.y), z(o.z) {}
    ZR_HDM V3& operator=(const V3& o) { x = o.x; y = o.y; z = o.z; return *this; }
};
struct V4
{
    float x, y, z, w;
    ZR_HDM V4() = default;
    ZR_HDM V4(const V4& o) : x(o.x), y(o.y), z(o.z), w(o.w) {}
    ZR_HDM V4& operator=(const V4& o) { x = o.x; y = o.y; z = o.z; w = o.w; return *this; }
};

ZR_HD V2 v2(float x, float y) { V2 r; r.x = x; r.y = y; return r; }
ZR_HD V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
ZR_HD V3 v3(float s) { V3 r; r.x = s; r.y = s; r.z = s; return r; }
ZR_HD V3 v3p(const float* p) { V3 r; r.x = p[0]; r.y = p[1]; r.z = p[2]; return r; }
ZR_HD V4 v4(float x, float y, float z, float w) { V4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }

ZR_HD V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
ZR_HD V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
ZR_HD V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
ZR_HD V3 operator/(V3 a, V3 b) { return v3(a.x / b.x, a.y / b.y, a.z / b.z); }
ZR_HD V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
ZR_HD V3 operator*(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
ZR_HD V3 operator/(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
ZR_HD V3 operator-(float s, V3 a) { return v3(s - a.x, s - a.y, s - a.z); }
ZR_HD V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
ZR_HD V2 operator+(V2 a, V2 b) { return v2(a.x + b.x, a.y + b.y); }
ZR_HD V2 operator-(V2 a, V2 b) { return v2(a.x - b.x, a.y - b.y); }
ZR_HD V2 operator*(float s, V2 a) { return v2(s * a.x, s * a.y); }
ZR_HD V2 operator*(V2 a, float s) { return v2(a.x * s, a.y * s); }

ZR_HD float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
ZR_HD float dot(V4 a, V4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
ZR_HD V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
ZR_HD float length(V3 a) { return zr_sqrt(dot(a, a)); }
ZR_HD V3 normalize(V3 a) { float inv = 1.0f / zr_sqrt(dot(a, a)); return a * inv; }
ZR_HD V4 normalize(V4 a) { float inv = 1.0f / zr_sqrt(dot(a, a)); return v4(a.x * inv, a.y * inv, a.z * inv, a.w * inv); }
// mad(s, a, b) per component
ZR_HD V3 mad(float s, V3 a, V3 b) { return v3(zr_fma(s, a.x, b.x), zr_fma(s, a.y, b.y), zr_fma(s, a.z, b.z)); }
ZR_HD V3 saturate(V3 a) { return v3(zr_saturate(a.x), zr_saturate(a.y), zr_saturate(a.z)); }
ZR_HD V3 vmax(V3 a, float s) { return v3(zr_max(a.x, s), zr_max(a.y, s), zr_max(a.z, s)); }
ZR_HD_FLAT V3 vexp(V3 a) { return v3(zr_exp(a.x), zr_exp(a.y), zr_exp(a.z)); }
ZR_HD_FLAT V3 vlog(V3 a) { return v3(zr_log(a.x), zr_log(a.y), zr_log(a.z)); }
ZR_HD bool any_nan(V3 a) { return zr_isnan(a.x) || zr_isnan(a.y) || zr_isnan(a.z); }
ZR_HD V3 reflect(V3 i, V3 n) { return i - 2.0f * dot(n, i) * n; }
ZR_HD V3 refract(V3 i, V3 n, float eta)
{
    float ndoti = dot(n, i);
    float k = 1.0f - eta * eta * (1.0f - ndoti * ndoti);
    if (k < 0.0f) return v3(0.0f);
    return eta * i - (eta * ndoti + zr_sqrt(k)) * n;
}

// ---- Math.hlsli ----
ZR_HD float NextFloat32(float f)     // Math.hlsli:30-39
{
    if (f == -0.0f) f = 0.0f;
    uint32_t u = zr_asuint(f);
    u = f >= 0 ? u + 1 : u - 1;
    return zr_asfloat(u);
}
ZR_HD float PrevFloat32(float f)     // Math.hlsli:42-51
{
    if (f == 0.0f) f = -0.0f;
    uint32_t u = zr_asuint(f);
    u = f > 0 ? u - 1 : u + 1;
    return zr_asfloat(u);
}
ZR_HD float Lerp(float v0, float v1, float t) { return zr_fma(t, v1, zr_fma(-t, v0, v0)); }   // Math.hlsli:66-70
ZR_HD V3 Lerp(V3 a, V3 b, float t) { return v3(Lerp(a.x, b.x, t), Lerp(a.y, b.y, t), Lerp(a.z, b.z, t)); }
ZR_HD float ArcCos(float x)          // Math.hlsli:103-113
{
    float xAbs = zr_abs(x);
    float res = zr_fma(-0.0206453f, xAbs, 0.0764532f);
    res = zr_fma(res, xAbs, -0.21271f);
    res = zr_fma(res, xAbs, 1.57075f);
    res *= zr_sqrt(1.0f - xAbs);
    return (x >= 0) ? res : ZR_PI - res;
}
ZR_HD V2 SphericalFromCartesian(V3 w)  // Math.hlsli:121-134: (theta, phi), phi clockwise from +x in [0, 2 pi)
{
    V2 thetaPhi;
    thetaPhi.x = ArcCos(w.y);
    thetaPhi.y = zr_atan2(-w.z, w.x);
    thetaPhi.y = thetaPhi.y < 0 ? thetaPhi.y + ZR_TWO_PI : thetaPhi.y;
    return thetaPhi;
}
ZR_HD float SignNotZero(float x) { return zr_asfloat(0x3f800000u | (0x80000000u & zr_asuint(x))); }  // :148-155
ZR_HD V2 NDCFromUV(V2 uv) { V2 n = v2(uv.x * 2.0f - 1.0f, uv.y * 2.0f - 1.0f); n.y = -n.y; return n; } // :163-169
ZR_HD V2 UVFromNDC(V2 ndc) { return v2(ndc.x * 0.5f + 0.5f, ndc.y * -0.5f + 0.5f); }                   // :171-174

struct ONB { V3 b1, b2; };
ZR_HD ONB BuildONB(V3 n)             // Math.hlsli:289-302 (Duff et al.)
{
    const float s = SignNotZero(n.z);
    const float a = -1.0f / (s + n.z);
    const float b = n.x * n.y * a;
    ONB r;
    r.b1 = v3(zr_fma(n.x * a, n.x * s, 1.0f), s * b, -s * n.x);
    r.b2 = v3(b, zr_fma(n.y * a, n.y, s), -n.y);
    return r;
}

struct TriDiffs { V3 dpdu, dpdv, dndu, dndv; };
// TriDifferentials::Unpack, Math.hlsli:385-402 (G-buffer planes TRI_DIFF_GEO_A / _B: 12 halfs)
ZR_HD TriDiffs UnpackTriDiffs(const uint32_t* a, const uint32_t* b)
{
    TriDiffs r;
    r.dpdu = v3(zr_f16_to_f32((uint16_t)(a[0] & 0xffff)), zr_f16_to_f32((uint16_t)(a[0] >> 16)), zr_f16_to_f32((uint16_t)(a[1] & 0xffff)));
    r.dpdv = v3(zr_f16_to_f32((uint16_t)(a[1] >> 16)), zr_f16_to_f32((uint16_t)(a[2] & 0xffff)), zr_f16_to_f32((uint16_t)(a[2] >> 16)));
    r.dndu = v3(zr_f16_to_f32((uint16_t)(a[3] & 0xffff)), zr_f16_to_f32((uint16_t)(a[3] >> 16)), zr_f16_to_f32((uint16_t)(b[0] & 0xffff)));
    r.dndv = v3(zr_f16_to_f32((uint16_t)(b[0] >> 16)), zr_f16_to_f32((uint16_t)(b[1] & 0xffff)), zr_f16_to_f32((uint16_t)(b[1] >> 16)));
    return r;
}
ZR_HD TriDiffs ComputeTriDiffs(V3 p0, V3 p1, V3 p2, V3 n0, V3 n1, V3 n2, V2 uv0, V2 uv1, V2 uv2)  // Math.hlsli:339-383
{
    TriDiffs ret;
    V2 duv10 = uv1 - uv0, duv20 = uv2 - uv0;
    float det = duv10.x * duv20.y - duv10.y * duv20.x;
    float invdet = 1.0f / det;
    if (zr_abs(det) < 1e-7f)
    {
        V3 normal = normalize(cross(p1 - p0, p2 - p0));
        ONB onb = BuildONB(normal);
        ret.dpdu = onb.b1; ret.dpdv = onb.b2; ret.dndu = v3(0.0f); ret.dndv = v3(0.0f);
        return ret;
    }
    V3 dp10 = p1 - p0, dp20 = p2 - p0;
    ret.dpdu = (duv20.y * dp10 - duv10.y * dp20) * invdet;
    ret.dpdv = (-duv20.x * dp10 + duv10.x * dp20) * invdet;
    V3 dn10 = n1 - n0, dn20 = n2 - n0;
    ret.dndu = (duv20.y * dn10 - duv10.y * dn20) * invdet;
    ret.dndv = (-duv20.x * dn10 + duv10.x * dn20) * invdet;
    return ret;
}

ZR_HD V3 RotateVector(V3 v, V4 q)    // Math.hlsli:556-565
{
    V3 im = v3(q.x, q.y, q.z);
    V3 t = cross(2.0f * im, v);
    return v + q.w * t + cross(im, t);
}
ZR_HD V3 TransformTRS(V3 pos, V3 tr, V4 rot, V3 scale)  // Math.hlsli:567-574
{
    V3 t = pos * scale;
    t = RotateVector(t, rot);
    return t + tr;
}
ZR_HD V3 InverseTransformTRS(V3 pos, V3 tr, V4 rot, V3 scale)   // Math.hlsli:576-584
{
    V3 t = pos - tr;
    t = RotateVector(t, v4(-rot.x, -rot.y, -rot.z, rot.w));
    return t * v3(1.0f / scale.x, 1.0f / scale.y, 1.0f / scale.z);
}
ZR_HD uint32_t FloatToUNorm8(float f) { f = zr_saturate(f); return (uint32_t)zr_fma(f, 255.0f, 0.5f); }   // :586-590
ZR_HD uint32_t FloatToUNorm16(float f) { f = zr_saturate(f); return (uint32_t)zr_fma(f, 65535.0f, 0.5f); } // :597-601
ZR_HD V4 DecodeNormalized4(const uint16_t* u)   // Math.hlsli:629-634
{
    return v4(zr_fma(zr_div65535((float)u[0]), 2.0f, -1.0f), zr_fma(zr_div65535((float)u[1]), 2.0f, -1.0f),
              zr_fma(zr_div65535((float)u[2]), 2.0f, -1.0f), zr_fma(zr_div65535((float)u[3]), 2.0f, -1.0f));
}
ZR_HD V2 EncodeUnitVector(V3 n)      // Math.hlsli:638-644
{
    float denom = zr_abs(n.x) + zr_abs(n.y) + zr_abs(n.z);
    V2 p = v2(n.x / denom, n.y / denom);
    V2 enc = (n.z <= 0.0f) ? v2((1.0f - zr_abs(p.y)) * SignNotZero(p.x), (1.0f - zr_abs(p.x)) * SignNotZero(p.y)) : p;
    return v2(zr_fma(enc.x, 0.5f, 0.5f), zr_fma(enc.y, 0.5f, 0.5f));
}
ZR_HD V3 DecodeUnitVector(V2 u)      // Math.hlsli:646-658
{
    u = v2(zr_fma(u.x, 2.0f, -1.0f), zr_fma(u.y, 2.0f, -1.0f));
    V3 n = v3(u.x, u.y, 1.0f - zr_abs(u.x) - zr_abs(u.y));
    float t = zr_saturate(-n.z);
    n.x += (n.x >= 0.0f) ? -t : t;
    n.y += (n.y >= 0.0f) ? -t : t;
    return normalize(n);
}
ZR_HD V3 DecodeOct32(const uint16_t* e) { return DecodeUnitVector(v2(zr_div65535((float)e[0]), zr_div65535((float)e[1]))); }
ZR_HD V3 DecodeOct32u(uint32_t e) { return DecodeUnitVector(v2(zr_div65535((float)(e & 0xffffu)), zr_div65535((float)(e >> 16)))); }
ZR_HD float Luminance(V3 c) { return dot(v3(0.2126f, 0.7152f, 0.0722f), c); }   // Math.hlsli:693-696
ZR_HD V3 UnpackRGB8(uint32_t rgb)    // Math.hlsli:733-741
{ return v3(zr_div255((float)(rgb & 0xff)), zr_div255((float)((rgb >> 8) & 0xff)), zr_div255((float)((rgb >> 16) & 0xff))); }
ZR_HD uint32_t Float3ToRGB8(V3 v)    // Math.hlsli:754-761
{
    v = saturate(v);
    return (uint32_t)zr_fma(v.x, 255.0f, 0.5f) | ((uint32_t)zr_fma(v.y, 255.0f, 0.5f) << 8) | ((uint32_t)zr_fma(v.z, 255.0f, 0.5f) << 16);
}
// D3D format conversions pinned by the ABI
ZR_HD uint32_t PackSnorm16(float f)
{
    if (zr_isnan(f)) f = 0;
    f = zr_clamp(f, -1.0f, 1.0f) * 32767.0f;
    int32_t i = (int32_t)(f >= 0 ? f + 0.5f : f - 0.5f);
    return (uint32_t)(uint16_t)(int16_t)i;
}
ZR_HD uint32_t PackUFloat(float f, int mbits)    // unsigned 5-bit-exponent float (R11G11B10_FLOAT channel), RTNE
{
    uint32_t x = zr_asuint(f);
    if (x & 0x80000000u) return 0;
    if (x >= 0x7f800000u) return x > 0x7f800000u ? ((0x1fu << mbits) | 1u) : (0x1fu << mbits);
    const int shift = 23 - mbits;
    if (x >= 0x47800000u) return (0x1eu << mbits) | ((1u << mbits) - 1u);
    if (x < 0x38800000u)
    {
        if (x < 0x33000000u) return 0;
        uint32_t e = x >> 23;
        uint32_t m = (x & 0x007fffffu) | 0x00800000u;
        uint32_t sh = (uint32_t)shift + (113u - e);
        if (sh > 24) return 0;
        uint32_t r = m >> sh;
        uint32_t rem = m & ((1u << sh) - 1u);
        uint32_t half = 1u << (sh - 1u);
        if (rem > half || (rem == half && (r & 1u))) r++;
        return r;
    }
    uint32_t r = (x - 0x38000000u) >> shift;
    uint32_t rem = x & ((1u << shift) - 1u);
    uint32_t half = 1u << (shift - 1);
    if (rem > half || (rem == half && (r & 1u))) r++;
    uint32_t maxv = (0x1eu << mbits) | ((1u << mbits) - 1u);
    return r > maxv ? maxv : r;
}
ZR_HD uint32_t PackR11G11B10F(V3 c) { return PackUFloat(c.x, 6) | (PackUFloat(c.y, 6) << 11) | (PackUFloat(c.z, 5) << 22); }

// ---- Sampling.hlsli:12-159 ----
struct Rng
{
    uint32_t s;
    ZR_HDM Rng() = default;
    ZR_HDM Rng(const Rng& o) : s(o.s) {}
    ZR_HDM Rng& operator=(const Rng& o) { s = o.s; return *this; }
    static ZR_HDM Rng Init(uint32_t px, uint32_t py, uint32_t frame) { Rng r; uint32_t x = px, y = py, z = frame; zr_pcg3d(&x, &y, &z); r.s = x; return r; }
    static ZR_HDM Rng Seed(uint32_t seed) { Rng r; r.s = seed; return r; }
    ZR_HDM uint32_t UniformUint()
    {
        s = s * 747796405u + 2891336453u;
        uint32_t word = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
        return (word >> 22u) ^ word;
    }
    ZR_HDM float Uniform() { return (float)(UniformUint() >> 8) * 5.9604644775390625e-8f; }
    ZR_HDM uint32_t UniformUintBounded(uint32_t bound)
    {
        uint32_t threshold = (~bound + 1u) % bound;
        for (;;) { uint32_t r = UniformUint(); if (r >= threshold) return r % bound; }
    }
    ZR_HDM uint32_t UniformUintBounded_Faster(uint32_t bound) { return (uint32_t)(Uniform() * (float)bound); }
    ZR_HDM V2 Uniform2D() { float a = Uniform(); float b = Uniform(); return v2(a, b); }
};

ZR_HD V3 SampleCosineWeightedHemisphere(V2 u, float* pdf)    // Sampling.hlsli:183-195
{
    const float phi = ZR_TWO_PI * u.y;
    const float sinTheta = zr_sqrt(u.x);
    float s, c; zr_sincos(phi, &s, &c);
    const float z = zr_sqrt(1.0f - u.x);
    *pdf = z * ZR_ONE_OVER_PI;
    return v3(c * sinTheta, s * sinTheta, z);
}
ZR_HD V3 UniformSampleCone(V2 u, float cosThetaMax, float* pdf)   // Sampling.hlsli:198-212
{
    const float phi = ZR_TWO_PI * u.y;
    const float cosTheta = zr_saturate((1.0f - u.x) + u.x * cosThetaMax);
    const float sinTheta = zr_sqrt(1.0f - cosTheta * cosTheta);
    float s, c; zr_sincos(phi, &s, &c);
    *pdf = ZR_ONE_OVER_2_PI / (1.0f - cosThetaMax);
    return v3(c * sinTheta, s * sinTheta, cosTheta);
}
ZR_HD V2 UniformSampleDiskConcentric(V2 u)     // Sampling.hlsli:222-244
{
    float a = 2.0f * u.x - 1.0f, b = 2.0f * u.y - 1.0f;
    if (a == 0 && b == 0) return v2(0, 0);
    float r, phi;
    if (a * a > b * b) { r = a; phi = ZR_PI_OVER_4 * (b / a); }
    else { r = b; phi = ZR_PI_OVER_2 - ZR_PI_OVER_4 * (a / b); }
    float s, c; zr_sincos(phi, &s, &c);
    return v2(r * c, r * s);
}
ZR_HD V2 UniformSampleTriangle(V2 u)           // Sampling.hlsli:270-287
{
    float b1, b2;
    if (u.y > u.x) { b1 = u.x * 0.5f; b2 = u.y - b1; }
    else { b2 = u.y * 0.5f; b1 = u.x - b2; }
    return v2(b1, b2);
}

// ---- RT.hlsli ----
// :233-241.  The reference's parameter is `uint2 pixel`: the auxiliary pixel int2(x, y - 1) of the ray differentials (RT.hlsli:332,
// GBufferRT.hlsli:41) wraps to 4294967295 on the top row of the screen -- the float conversion sees that value (found by the textured
// reference-pass pins: the y-gradient of row 0 is huge there, so its texture fetches land on the coarsest mip)
ZR_HD V3 GeneratePinholeCameraRay_CS(int px, int py, V2 renderDim, float aspectRatio, float tanHalfFOV, V2 jitter)
{
    V2 uv = v2(((float)(uint32_t)px + 0.5f + jitter.x) / renderDim.x, ((float)(uint32_t)py + 0.5f + jitter.y) / renderDim.y);
    V2 ndc = NDCFromUV(uv);
    return v3(ndc.x * aspectRatio * tanHalfFOV, ndc.y * tanHalfFOV, 1);
}
ZR_HD V3 GeneratePinholeCameraRay(int px, int py, V2 renderDim, float aspectRatio, float tanHalfFOV, V3 vbx, V3 vby, V3 vbz, V2 jitter)  // :222-231
{
    V3 dirV = GeneratePinholeCameraRay_CS(px, py, renderDim, aspectRatio, tanHalfFOV, jitter);
    return normalize(mad(dirV.x, vbx, mad(dirV.y, vby, dirV.z * vbz)));
}
ZR_HD V3 OffsetRayRTG(V3 pos, V3 gn)           // RT.hlsli:245-262 (Waechter-Binder)
{
    const float origin = 1.0f / 32.0f, float_scale = 1.0f / 65536.0f, int_scale = 256.0f;
    int32_t ofx = (int32_t)(int_scale * gn.x), ofy = (int32_t)(int_scale * gn.y), ofz = (int32_t)(int_scale * gn.z);
    V3 p_i = v3(zr_asfloat((uint32_t)((int32_t)zr_asuint(pos.x) + ((pos.x < 0) ? -ofx : ofx))),
                zr_asfloat((uint32_t)((int32_t)zr_asuint(pos.y) + ((pos.y < 0) ? -ofy : ofy))),
                zr_asfloat((uint32_t)((int32_t)zr_asuint(pos.z) + ((pos.z < 0) ? -ofz : ofz))));
    return v3(zr_abs(pos.x) < origin ? pos.x + float_scale * gn.x : p_i.x,
              zr_abs(pos.y) < origin ? pos.y + float_scale * gn.y : p_i.y,
              zr_abs(pos.z) < origin ? pos.z + float_scale * gn.z : p_i.z);
}
ZR_HD float BalanceHeuristic3(float p_1, float p_2, float p_3, float f)   // RT.hlsli:285-294 (n_i = 1)
{
    float denom = 1.0f * p_1 + 1.0f * p_2 + 1.0f * p_3;
    if (denom == 0) return 0;
    return (1.0f * f) / denom;
}
ZR_HD V3 PowerHeuristic(float p_1, float p_2, V3 f, float n_1, float n_2)  // RT.hlsli:297-307
{
    float a = n_1 * p_1, b = n_2 * p_2;
    float denom = a * a + b * b;
    if (denom == 0) return v3(0.0f);
    return (n_1 * n_1 * p_1 * f) / denom;
}

// RT::RayDifferentials, RT.hlsli:309-479 (Igehy).  Only texture LODs consume them: kernels carry them when the scene
// has a texture heap and skip them otherwise.
struct RayDiffs
{
    V3 origin_x, dir_x, origin_y, dir_y;
    V4 uv_grads;

    // Init (:323-353).  uv_grads is left uninitialised by the reference and pinned to 0 by the ABI.
    static ZR_HDM RayDiffs Init(int px, int py, V2 renderDim, float tanHalfFOV, float aspectRatio, V2 jitter, V3 vbx, V3 vby, V3 vbz,
        bool thinLens, float focusDepth, V2 lensSample, V3 origin)
    {
        RayDiffs ret;
        V3 dir_cs_x = GeneratePinholeCameraRay_CS(px + 1, py, renderDim, aspectRatio, tanHalfFOV, jitter);
        V3 dir_cs_y = GeneratePinholeCameraRay_CS(px, py - 1, renderDim, aspectRatio, tanHalfFOV, jitter);
        if (thinLens)
        {
            dir_cs_x = focusDepth * dir_cs_x - v3(lensSample.x, lensSample.y, 0);
            dir_cs_y = focusDepth * dir_cs_y - v3(lensSample.x, lensSample.y, 0);
        }
        ret.dir_x = normalize(mad(dir_cs_x.x, vbx, mad(dir_cs_x.y, vby, dir_cs_x.z * vbz)));
        ret.dir_y = normalize(mad(dir_cs_y.x, vbx, mad(dir_cs_y.y, vby, dir_cs_y.z * vbz)));
        ret.origin_x = origin; ret.origin_y = origin;
        ret.uv_grads = v4(0, 0, 0, 0);
        return ret;
    }
    static ZR_HDM RayDiffs Zero()
    { RayDiffs r; r.origin_x = v3(0.0f); r.dir_x = v3(0.0f); r.origin_y = v3(0.0f); r.dir_y = v3(0.0f); r.uv_grads = v4(0, 0, 0, 0); return r; }

    // dpdx_dpdy (:355-378)
    ZR_HDM void dpdx_dpdy(V3 hitPoint, V3 normal, V3& dpdx, V3& dpdy) const
    {
        const float d = dot(normal, hitPoint);
        const float numerator_x = d - dot(normal, origin_x);
        const float denom_x = dot(normal, dir_x);
        const float t_x = numerator_x / denom_x;
        const V3 hitPoint_x = mad(t_x, dir_x, origin_x);
        const float numerator_y = d - dot(normal, origin_y);
        const float denom_y = dot(normal, dir_y);
        const float t_y = numerator_y / denom_y;
        const V3 hitPoint_y = mad(t_y, dir_y, origin_y);
        dpdx = denom_x != 0 ? hitPoint_x - hitPoint : v3(ZR_FLT16_MAX);
        dpdy = denom_y != 0 ? hitPoint_y - hitPoint : v3(ZR_FLT16_MAX);
    }

    // UpdateRays (:380-440)
    ZR_HDM void UpdateRays(V3 p, V3 normal, V3 wi, V3 wo, V3 dndu, V3 dndv, V3 dpdx, V3 dpdy, bool transmitted, float eta)
    {
        origin_x = p + dpdx;
        origin_y = p + dpdy;
        const V3 dwodx = -dir_x - wo;
        const V3 dwody = -dir_y - wo;
        const V3 dndx = dndu * uv_grads.x + dndv * uv_grads.y;
        const V3 dndy = dndu * uv_grads.z + dndv * uv_grads.w;
        const float dndotWodx = dot(dndx, wo) + dot(normal, dwodx);
        const float dndotWody = dot(dndy, wo) + dot(normal, dwody);
        const float ndotwo = dot(normal, wo);
        if (!transmitted)
        {
            dir_x = wi + mad(2.0f, mad(dndotWodx, normal, ndotwo * dndx), -dwodx);
            dir_y = wi + mad(2.0f, mad(dndotWody, normal, ndotwo * dndy), -dwody);
        }
        else
        {
            const float eta_relative = 1.0f / eta;
            const float ndotwi = zr_abs(dot(normal, wi));
            const float q = zr_fma(eta_relative, ndotwo, -ndotwi);
            const float common = zr_fma(eta_relative, -ndotwo / ndotwi, 1.0f);
            const float dqdx = eta_relative * dndotWodx * common;
            const float dqdy = eta_relative * dndotWody * common;
            dir_x = mad(-eta_relative, dwodx, wi) + mad(q, dndx, dqdx * normal);
            dir_y = mad(-eta_relative, dwody, wi) + mad(q, dndy, dqdy * normal);
        }
    }

    // ComputeUVDifferentials (:442-478)
    ZR_HDM void ComputeUVDifferentials(V3 dpdx, V3 dpdy, V3 dpdu, V3 dpdv)
    {
        const float dpduDotdpdu = dot(dpdu, dpdu);
        const float dpdvDotdpdv = dot(dpdv, dpdv);
        const float dpduDotdpdv = dot(dpdu, dpdv);
        const float det = dpduDotdpdu * dpdvDotdpdv - dpduDotdpdv * dpduDotdpdv;
        if (zr_abs(det) < 1e-7f) { uv_grads = v4(0, 0, 0, 0); return; }
        const V2 b_x = v2(dot(dpdu, dpdx), dot(dpdv, dpdx));
        const V2 grads_x = v2((dpdvDotdpdv * b_x.x + -dpduDotdpdv * b_x.y) / det, (-dpduDotdpdv * b_x.x + dpduDotdpdu * b_x.y) / det);
        const V2 b_y = v2(dot(dpdu, dpdy), dot(dpdv, dpdy));
        const V2 grads_y = v2((dpdvDotdpdv * b_y.x + -dpduDotdpdv * b_y.y) / det, (-dpduDotdpdv * b_y.x + dpduDotdpdu * b_y.y) / det);
        const bool invalid_x = (uv_grads.x == ZR_FLT16_MAX) || (dpdx.x == ZR_FLT16_MAX);
        const bool invalid_y = (uv_grads.z == ZR_FLT16_MAX) || (dpdy.x == ZR_FLT16_MAX);
        uv_grads.x = invalid_x ? ZR_FLT16_MAX : grads_x.x;
        uv_grads.y = invalid_x ? ZR_FLT16_MAX : grads_x.y;
        uv_grads.z = invalid_y ? ZR_FLT16_MAX : grads_y.x;
        uv_grads.w = invalid_y ? ZR_FLT16_MAX : grads_y.y;
    }
};

ZR_HD V3 Row3(const float* m, int r) { return v3(m[4 * r], m[4 * r + 1], m[4 * r + 2]); }
ZR_HD V3 Mul3x4(const float* m, V3 p)
{
    return v3(m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3], m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7],
              m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]);
}

} // namespace zr
