// ORACLE -- test infrastructure only.  Nothing under oracle/ is linked into, imported by, or called from the product
// library (zetaray_amd/); only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
//
// zro_math.h: CPU restatement of the reference's shared shader math:
//   Source/ZetaRenderPass/Common/Math.hlsli, Sampling.hlsli, RT.hlsli (file:line cited per function).
// Scalar C++ (g++), one operation per HLSL operation, `mad` -> zr_fma, transcendentals from include/zr_detmath.h
// (the ABI's arithmetic contract).  Compile with -ffp-contract=off.
#pragma once
#include <cstdint>
#include <cstring>
#include "../include/zr_detmath.h"
#include "../include/zr_wire.h"

namespace zro {

struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };

static inline float2 f2(float x, float y) { return {x, y}; }
static inline float3 f3(float x, float y, float z) { return {x, y, z}; }
static inline float3 f3(float s) { return {s, s, s}; }
static inline float4 f4(float x, float y, float z, float w) { return {x, y, z, w}; }
static inline float3 f3(const float* p) { return {p[0], p[1], p[2]}; }

static inline float3 operator+(float3 a, float3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline float3 operator-(float3 a, float3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline float3 operator*(float3 a, float3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
static inline float3 operator/(float3 a, float3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
static inline float3 operator*(float3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
static inline float3 operator*(float s, float3 a) { return {s * a.x, s * a.y, s * a.z}; }
static inline float3 operator/(float3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
static inline float3 operator+(float3 a, float s) { return {a.x + s, a.y + s, a.z + s}; }
static inline float3 operator-(float3 a, float s) { return {a.x - s, a.y - s, a.z - s}; }
static inline float3 operator-(float s, float3 a) { return {s - a.x, s - a.y, s - a.z}; }
static inline float3 operator-(float3 a) { return {-a.x, -a.y, -a.z}; }
static inline float3& operator+=(float3& a, float3 b) { a = a + b; return a; }
static inline float3& operator*=(float3& a, float3 b) { a = a * b; return a; }
static inline float3& operator*=(float3& a, float s) { a = a * s; return a; }
static inline float3& operator/=(float3& a, float s) { a = a / s; return a; }
static inline float2 operator+(float2 a, float2 b) { return {a.x + b.x, a.y + b.y}; }
static inline float2 operator-(float2 a, float2 b) { return {a.x - b.x, a.y - b.y}; }
static inline float2 operator*(float2 a, float s) { return {a.x * s, a.y * s}; }
static inline float2 operator*(float s, float2 a) { return {s * a.x, s * a.y}; }
static inline float2 operator/(float2 a, float s) { return {a.x / s, a.y / s}; }
static inline float2 operator/(float2 a, float2 b) { return {a.x / b.x, a.y / b.y}; }

// HLSL dot(): left-to-right sum of products (no contraction)
static inline float dot(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline float dot(float2 a, float2 b) { return a.x * b.x + a.y * b.y; }
static inline float dot(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
static inline float3 cross(float3 a, float3 b)
{ return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
static inline float length(float3 a) { return zr_sqrt(dot(a, a)); }
// ABI definition of normalize(): v * (1 / sqrt(dot(v, v)))
static inline float3 normalize(float3 a) { float inv = 1.0f / zr_sqrt(dot(a, a)); return a * inv; }
static inline float4 normalize(float4 a)
{ float inv = 1.0f / zr_sqrt(dot(a, a)); return {a.x * inv, a.y * inv, a.z * inv, a.w * inv}; }
static inline float3 mad3(float s, float3 a, float3 b) { return {zr_fma(s, a.x, b.x), zr_fma(s, a.y, b.y), zr_fma(s, a.z, b.z)}; }
static inline float3 mad3(float3 s, float3 a, float3 b) { return {zr_fma(s.x, a.x, b.x), zr_fma(s.y, a.y, b.y), zr_fma(s.z, a.z, b.z)}; }
static inline float3 abs3(float3 a) { return {zr_abs(a.x), zr_abs(a.y), zr_abs(a.z)}; }
static inline float3 max3(float3 a, float s) { return {zr_max(a.x, s), zr_max(a.y, s), zr_max(a.z, s)}; }
static inline float3 saturate3(float3 a) { return {zr_saturate(a.x), zr_saturate(a.y), zr_saturate(a.z)}; }
static inline float3 exp3(float3 a) { return {zr_exp(a.x), zr_exp(a.y), zr_exp(a.z)}; }
static inline float3 log3(float3 a) { return {zr_log(a.x), zr_log(a.y), zr_log(a.z)}; }
static inline bool any_nan(float3 a) { return zr_isnan(a.x) || zr_isnan(a.y) || zr_isnan(a.z); }
// HLSL reflect(i, n) = i - 2 * dot(n, i) * n
static inline float3 reflect(float3 i, float3 n) { return i - 2.0f * dot(n, i) * n; }
// HLSL refract(i, n, eta)
static inline float3 refract(float3 i, float3 n, float eta)
{
    float ndoti = dot(n, i);
    float k = 1.0f - eta * eta * (1.0f - ndoti * ndoti);
    if (k < 0.0f) return f3(0.0f);
    return eta * i - (eta * ndoti + zr_sqrt(k)) * n;
}

namespace Math {
    // Math.hlsli:30-51
    static inline float NextFloat32(float f)
    {
        if (f == -0.0f) f = 0.0f;
        uint32_t u = zr_asuint(f);
        u = f >= 0 ? u + 1 : u - 1;
        return zr_asfloat(u);
    }
    static inline float PrevFloat32(float f)
    {
        if (f == 0.0f) f = -0.0f;
        uint32_t u = zr_asuint(f);
        u = f > 0 ? u - 1 : u + 1;
        return zr_asfloat(u);
    }
    // Math.hlsli:66-70
    static inline float Lerp(float v0, float v1, float t) { return zr_fma(t, v1, zr_fma(-t, v0, v0)); }
    static inline float3 Lerp(float3 v0, float3 v1, float t)
    { return {Lerp(v0.x, v1.x, t), Lerp(v0.y, v1.y, t), Lerp(v0.z, v1.z, t)}; }
    // Math.hlsli:103-113
    static inline float ArcCos(float x)
    {
        float xAbs = zr_abs(x);
        float res = zr_fma(-0.0206453f, xAbs, 0.0764532f);
        res = zr_fma(res, xAbs, -0.21271f);
        res = zr_fma(res, xAbs, 1.57075f);
        res *= zr_sqrt(1.0f - xAbs);
        return (x >= 0) ? res : ZR_PI - res;
    }
    // Math.hlsli:121-134
    static inline float2 SphericalFromCartesian(float3 w)
    {
        float2 thetaPhi;
        thetaPhi.x = ArcCos(w.y);
        thetaPhi.y = zr_atan2(-w.z, w.x);
        thetaPhi.y = thetaPhi.y < 0 ? thetaPhi.y + ZR_TWO_PI : thetaPhi.y;
        return thetaPhi;
    }
    // Math.hlsli:148-155
    static inline float SignNotZero(float x)
    { return zr_asfloat(0x3f800000u | (0x80000000u & zr_asuint(x))); }
    // Math.hlsli:163-174
    static inline float2 NDCFromUV(float2 uv) { float2 ndc = {uv.x * 2.0f - 1.0f, uv.y * 2.0f - 1.0f}; ndc.y = -ndc.y; return ndc; }
    static inline float2 UVFromNDC(float2 ndc) { return {ndc.x * 0.5f + 0.5f, ndc.y * -0.5f + 0.5f}; }

    // Math.hlsli:218-248 (pinhole branch and thin lens branch)
    static inline float3 WorldPosFromScreenSpace2(float2 pos_ss, float2 renderDim, float z_view, float tanHalfFOV,
        float aspectRatio, float2 jitter, float3 viewBasisX, float3 viewBasisY, float3 viewBasisZ,
        bool thinLens, float2 lensSample, float focusDepth, float3& origin)
    {
        float2 uv = {(pos_ss.x + 0.5f + jitter.x) / renderDim.x, (pos_ss.y + 0.5f + jitter.y) / renderDim.y};
        float2 ndc = NDCFromUV(uv);
        float3 dir_w;
        if (!thinLens)
        {
            float3 dir_v = f3(ndc.x * aspectRatio * tanHalfFOV * z_view, ndc.y * tanHalfFOV * z_view, z_view);
            dir_w = mad3(dir_v.x, viewBasisX, mad3(dir_v.y, viewBasisY, dir_v.z * viewBasisZ));
        }
        else
        {
            float3 dir_v = f3(ndc.x * aspectRatio * tanHalfFOV, ndc.y * tanHalfFOV, 1);
            float3 focalPoint = focusDepth * dir_v;
            dir_v = focalPoint - f3(lensSample.x, lensSample.y, 0);
            dir_w = mad3(dir_v.x, viewBasisX, mad3(dir_v.y, viewBasisY, dir_v.z * viewBasisZ));
            dir_w = normalize(dir_w);
            dir_w *= z_view;
            origin += mad3(lensSample.x, viewBasisX, lensSample.y * viewBasisY);
        }
        return origin + dir_w;
    }

    // Math.hlsli:289-306 (Duff et al. ONB)
    struct CoordinateSystem
    {
        float3 b1, b2;
        static CoordinateSystem Build(float3 n)
        {
            const float s = SignNotZero(n.z);
            const float a = -1.0f / (s + n.z);
            const float b = n.x * n.y * a;
            CoordinateSystem ret;
            ret.b1 = f3(zr_fma(n.x * a, n.x * s, 1.0f), s * b, -s * n.x);
            ret.b2 = f3(b, zr_fma(n.y * a, n.y, s), -n.y);
            return ret;
        }
    };

    // Math.hlsli:325-389
    // Math.hlsli:263-284
    static inline float3 TangentSpaceToWorldSpace(float2 bumpNormal2, float3 tangent, float3 normal, float scale)
    {
        float3 bumpNormal = f3(zr_fma(2.0f, bumpNormal2.x, -1.0f), zr_fma(2.0f, bumpNormal2.y, -1.0f), 0.0f);
        bumpNormal.z = zr_sqrt(zr_saturate(1.0f - dot(bumpNormal, bumpNormal)));
        float3 scaledBumpNormal = bumpNormal * f3(scale, scale, 1.0f);
        if (dot(scaledBumpNormal, scaledBumpNormal) < 1e-6f) return normal;
        scaledBumpNormal = normalize(scaledBumpNormal);
        normal = normalize(normal);
        tangent = normalize(tangent - dot(tangent, normal) * normal);
        float3 bitangent = cross(normal, tangent);
        // mul(row vector, float3x3(tangent, bitangent, normal)): sum over rows, left to right
        return scaledBumpNormal.x * tangent + scaledBumpNormal.y * bitangent + scaledBumpNormal.z * normal;
    }

    struct TriDifferentials
    {
        float3 dpdu, dpdv, dndu, dndv;

        static TriDifferentials Unpack(const uint32_t a[4], const uint32_t b[2])
        {
            TriDifferentials r;
            r.dpdu = f3(zr_f16_to_f32(a[0] & 0xffff), zr_f16_to_f32(a[0] >> 16), zr_f16_to_f32(a[1] & 0xffff));
            r.dpdv = f3(zr_f16_to_f32(a[1] >> 16), zr_f16_to_f32(a[2] & 0xffff), zr_f16_to_f32(a[2] >> 16));
            r.dndu = f3(zr_f16_to_f32(a[3] & 0xffff), zr_f16_to_f32(a[3] >> 16), zr_f16_to_f32(b[0] & 0xffff));
            r.dndv = f3(zr_f16_to_f32(b[0] >> 16), zr_f16_to_f32(b[1] & 0xffff), zr_f16_to_f32(b[1] >> 16));
            return r;
        }

        static TriDifferentials Compute(float3 p0, float3 p1, float3 p2, float3 n0, float3 n1, float3 n2,
            float2 uv0, float2 uv1, float2 uv2)
        {
            TriDifferentials ret;
            float2 duv10 = uv1 - uv0;
            float2 duv20 = uv2 - uv0;
            float det = duv10.x * duv20.y - duv10.y * duv20.x;
            float invdet = 1.0f / det;
            if (zr_abs(det) < 1e-7f)
            {
                float3 normal = normalize(cross(p1 - p0, p2 - p0));
                CoordinateSystem onb = CoordinateSystem::Build(normal);
                ret.dpdu = onb.b1; ret.dpdv = onb.b2; ret.dndu = f3(0.0f); ret.dndv = f3(0.0f);
                return ret;
            }
            float3 dp10 = p1 - p0, dp20 = p2 - p0;
            ret.dpdu = (duv20.y * dp10 - duv10.y * dp20) * invdet;
            ret.dpdv = (-duv20.x * dp10 + duv10.x * dp20) * invdet;
            float3 dn10 = n1 - n0, dn20 = n2 - n0;
            ret.dndu = (duv20.y * dn10 - duv10.y * dn20) * invdet;
            ret.dndv = (-duv20.x * dn10 + duv10.x * dn20) * invdet;
            return ret;
        }
    };

    // Math.hlsli:391-408
    static inline float4 RotationQuaternion_Acute(float3 axis, float theta)
    {
        float s = zr_sin(0.5f * theta);
        float c = zr_sqrt(1 - s * s);
        return {s * axis.x, s * axis.y, s * axis.z, c};
    }
    static inline float4 InverseRotationQuaternion(float4 q) { return {-q.x, -q.y, -q.z, q.w}; }
    // Math.hlsli:556-584
    static inline float3 RotateVector(float3 v, float4 q)
    {
        float3 imaginary = f3(q.x, q.y, q.z);
        float real = q.w;
        float3 t = cross(2.0f * imaginary, v);
        return v + real * t + cross(imaginary, t);
    }
    static inline float3 TransformTRS(float3 pos, float3 translation, float4 rotation, float3 scale)
    {
        float3 transformed = pos * scale;
        transformed = RotateVector(transformed, rotation);
        transformed += translation;
        return transformed;
    }
    static inline float3 InverseTransformTRS(float3 pos, float3 translation, float4 rotation, float3 scale)
    {
        float3 transformed = pos - translation;
        float4 qc = {-rotation.x, -rotation.y, -rotation.z, rotation.w};
        transformed = RotateVector(transformed, qc);
        transformed *= f3(1.0f / scale.x, 1.0f / scale.y, 1.0f / scale.z);
        return transformed;
    }
    // Math.hlsli:586-634
    static inline uint32_t FloatToUNorm8(float f) { f = zr_saturate(f); return (uint32_t)zr_fma(f, 255.0f, 0.5f); }
    static inline float UNorm8ToFloat(uint32_t u) { return (float)u / 255.0f; }
    static inline uint16_t FloatToUNorm16(float f) { f = zr_saturate(f); return (uint16_t)zr_fma(f, 65535.0f, 0.5f); }
    static inline float UNorm16ToFloat(uint16_t u) { return (float)u / 65535.0f; }
    static inline float4 DecodeNormalized4(const uint16_t u[4])
    {
        float4 d = {(float)u[0] / 65535.0f, (float)u[1] / 65535.0f, (float)u[2] / 65535.0f, (float)u[3] / 65535.0f};
        return {zr_fma(d.x, 2.0f, -1.0f), zr_fma(d.y, 2.0f, -1.0f), zr_fma(d.z, 2.0f, -1.0f), zr_fma(d.w, 2.0f, -1.0f)};
    }
    // Math.hlsli:636-678 (octahedral)
    static inline float2 EncodeUnitVector(float3 n)
    {
        float denom = zr_abs(n.x) + zr_abs(n.y) + zr_abs(n.z);
        float2 p = {n.x / denom, n.y / denom};
        float2 encoded = (n.z <= 0.0f) ?
            f2((1.0f - zr_abs(p.y)) * SignNotZero(p.x), (1.0f - zr_abs(p.x)) * SignNotZero(p.y)) : p;
        return {zr_fma(encoded.x, 0.5f, 0.5f), zr_fma(encoded.y, 0.5f, 0.5f)};
    }
    static inline float3 DecodeUnitVector(float2 u)
    {
        u = {zr_fma(u.x, 2.0f, -1.0f), zr_fma(u.y, 2.0f, -1.0f)};
        float3 n = f3(u.x, u.y, 1.0f - zr_abs(u.x) - zr_abs(u.y));
        float t = zr_saturate(-n.z);
        n.x += (n.x >= 0.0f) ? -t : t;
        n.y += (n.y >= 0.0f) ? -t : t;
        return normalize(n);
    }
    static inline void EncodeOct32(float3 n, uint16_t out[2])
    {
        float2 u = EncodeUnitVector(n);
        out[0] = FloatToUNorm16(u.x); out[1] = FloatToUNorm16(u.y);
    }
    static inline float3 DecodeOct32(const uint16_t e[2])
    {
        float2 u = {(float)e[0] / 65535.0f, (float)e[1] / 65535.0f};
        return DecodeUnitVector(u);
    }
    static inline float2 DecodeUNorm2(const uint16_t e[2]) { return {(float)e[0] / 65535.0f, (float)e[1] / 65535.0f}; }
    // Math.hlsli:693-761
    static inline float Luminance(float3 c) { return dot(f3(0.2126f, 0.7152f, 0.0722f), c); }
    static inline float3 UnpackRGB8(uint32_t rgb)
    { return f3((float)(rgb & 0xff) / 255.0f, (float)((rgb >> 8) & 0xff) / 255.0f, (float)((rgb >> 16) & 0xff) / 255.0f); }
    static inline float2 UnpackRG(uint32_t rg) { return {(float)(rg & 0xff) / 255.0f, (float)((rg >> 8) & 0xff) / 255.0f}; }
    static inline uint32_t Float3ToRGB8(float3 v)
    {
        v = saturate3(v);
        uint32_t x = (uint32_t)zr_fma(v.x, 255.0f, 0.5f), y = (uint32_t)zr_fma(v.y, 255.0f, 0.5f), z = (uint32_t)zr_fma(v.z, 255.0f, 0.5f);
        return x | (y << 8) | (z << 16);
    }
}

// Sampling.hlsli:12-159
// float -> unsigned small float with `mbits` mantissa bits, 5 exponent bits (R11G11B10_FLOAT), round-to-nearest-even
static inline uint32_t PackUFloat(float f, int mbits)
{
    uint32_t x = zr_asuint(f);
    if (x & 0x80000000u) return 0;                       // negative (and -0) -> 0
    if (x >= 0x7f800000u) return x > 0x7f800000u ? ((0x1fu << mbits) | 1u) : (0x1fu << mbits);   // nan / inf
    const int shift = 23 - mbits;
    if (x >= 0x47800000u) return (0x1eu << mbits) | ((1u << mbits) - 1u);   // >= 65536 -> max finite
    if (x < 0x38800000u)
    {
        // denormal in the target format: value / 2^-14 * 2^mbits
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
static inline uint32_t PackR11G11B10F(float3 c) { return PackUFloat(c.x, 6) | (PackUFloat(c.y, 6) << 11) | (PackUFloat(c.z, 5) << 22); }


struct RNG
{
    uint32_t State;
    static RNG Init(uint32_t px, uint32_t py, uint32_t frame)
    { RNG r; uint32_t x = px, y = py, z = frame; zr_pcg3d(&x, &y, &z); r.State = x; return r; }
    static RNG Init4(uint32_t px, uint32_t py, uint32_t frame, uint32_t idx)
    { RNG r; uint32_t x = px, y = py, z = frame, w = idx; zr_pcg4d(&x, &y, &z, &w); r.State = x; return r; }
    static RNG InitIdx(uint32_t idx, uint32_t frame) { RNG r; r.State = zr_pcg(idx + zr_pcg(frame)); return r; }
    static RNG InitSeed(uint32_t seed) { RNG r; r.State = seed; return r; }
    uint32_t UniformUint()
    {
        State = State * 747796405u + 2891336453u;
        uint32_t word = ((State >> ((State >> 28u) + 4u)) ^ State) * 277803737u;
        return (word >> 22u) ^ word;
    }
    float Uniform() { return (float)(UniformUint() >> 8) * 5.9604644775390625e-8f; }
    uint32_t UniformUintBounded(uint32_t bound)
    {
        uint32_t threshold = (~bound + 1u) % bound;
        for (;;) { uint32_t r = UniformUint(); if (r >= threshold) return r % bound; }
    }
    uint32_t UniformUintBounded_Faster(uint32_t bound) { return (uint32_t)(Uniform() * (float)bound); }
    float2 Uniform2D() { float a = Uniform(); float b = Uniform(); return {a, b}; }
    float3 Uniform3D() { float a = Uniform(); float b = Uniform(); float c = Uniform(); return {a, b, c}; }
};

namespace Sampling {
    // Sampling.hlsli:183-195
    static inline float3 SampleCosineWeightedHemisphere(float2 u, float& pdf)
    {
        const float phi = ZR_TWO_PI * u.y;
        const float sinTheta = zr_sqrt(u.x);
        float s, c; zr_sincos(phi, &s, &c);
        const float x = c * sinTheta;
        const float y = s * sinTheta;
        const float z = zr_sqrt(1.0f - u.x);
        pdf = z * ZR_ONE_OVER_PI;
        return f3(x, y, z);
    }
    // Sampling.hlsli:198-212
    static inline float3 UniformSampleCone(float2 u, float cosThetaMax, float& pdf)
    {
        const float phi = ZR_TWO_PI * u.y;
        const float cosTheta = zr_saturate((1.0f - u.x) + u.x * cosThetaMax);
        const float sinTheta = zr_sqrt(1.0f - cosTheta * cosTheta);
        float s, c; zr_sincos(phi, &s, &c);
        pdf = ZR_ONE_OVER_2_PI / (1.0f - cosThetaMax);
        return f3(c * sinTheta, s * sinTheta, cosTheta);
    }
    // Sampling.hlsli:222-244
    static inline float2 UniformSampleDiskConcentric(float2 u)
    {
        float a = 2.0f * u.x - 1.0f;
        float b = 2.0f * u.y - 1.0f;
        if (a == 0 && b == 0) return {0, 0};
        float r, phi;
        if (a * a > b * b) { r = a; phi = ZR_PI_OVER_4 * (b / a); }
        else { r = b; phi = ZR_PI_OVER_2 - ZR_PI_OVER_4 * (a / b); }
        float s, c; zr_sincos(phi, &s, &c);
        return {r * c, r * s};
    }
    // Sampling.hlsli:270-287 (Heitz low-distortion map)
    static inline float2 UniformSampleTriangle(float2 u)
    {
        float b1, b2;
        if (u.y > u.x) { b1 = u.x * 0.5f; b2 = u.y - b1; }
        else { b2 = u.y * 0.5f; b1 = u.x - b2; }
        return {b1, b2};
    }
}

namespace RT {
    // RT.hlsli:233-241.  The parameter is `uint2 pixel`: the ray differentials' auxiliary pixel int2(x, y - 1) (RT.hlsli:332, GBufferRT.hlsli:41)
    // wraps to 4294967295 for the top row of the screen, and that is the value the float conversion sees
    static inline float3 GeneratePinholeCameraRay_CS(int px, int py, float2 renderDim, float aspectRatio,
        float tanHalfFOV, float2 jitter)
    {
        float2 uv = {((float)(uint32_t)px + 0.5f + jitter.x) / renderDim.x, ((float)(uint32_t)py + 0.5f + jitter.y) / renderDim.y};
        float2 ndc = Math::NDCFromUV(uv);
        return f3(ndc.x * aspectRatio * tanHalfFOV, ndc.y * tanHalfFOV, 1);
    }
    // RT.hlsli:222-231
    static inline float3 GeneratePinholeCameraRay(int px, int py, float2 renderDim, float aspectRatio, float tanHalfFOV,
        float3 viewBasisX, float3 viewBasisY, float3 viewBasisZ, float2 jitter)
    {
        float3 dirV = GeneratePinholeCameraRay_CS(px, py, renderDim, aspectRatio, tanHalfFOV, jitter);
        float3 dirW = mad3(dirV.x, viewBasisX, mad3(dirV.y, viewBasisY, dirV.z * viewBasisZ));
        return normalize(dirW);
    }
    // RT.hlsli:245-262 (Waechter-Binder)
    static inline float3 OffsetRayRTG(float3 pos, float3 geometricNormal)
    {
        const float origin = 1.0f / 32.0f;
        const float float_scale = 1.0f / 65536.0f;
        const float int_scale = 256.0f;
        int32_t ofx = (int32_t)(int_scale * geometricNormal.x);
        int32_t ofy = (int32_t)(int_scale * geometricNormal.y);
        int32_t ofz = (int32_t)(int_scale * geometricNormal.z);
        float3 p_i = f3(zr_asfloat((uint32_t)((int32_t)zr_asuint(pos.x) + ((pos.x < 0) ? -ofx : ofx))),
                        zr_asfloat((uint32_t)((int32_t)zr_asuint(pos.y) + ((pos.y < 0) ? -ofy : ofy))),
                        zr_asfloat((uint32_t)((int32_t)zr_asuint(pos.z) + ((pos.z < 0) ? -ofz : ofz))));
        return f3(zr_abs(pos.x) < origin ? pos.x + float_scale * geometricNormal.x : p_i.x,
                  zr_abs(pos.y) < origin ? pos.y + float_scale * geometricNormal.y : p_i.y,
                  zr_abs(pos.z) < origin ? pos.z + float_scale * geometricNormal.z : p_i.z);
    }
    // RT.hlsli:273-307
    static inline float BalanceHeuristic3(float p_1, float p_2, float p_3, float f)
    {
        float denom = 1.0f * p_1 + 1.0f * p_2 + 1.0f * p_3;
        if (denom == 0) return 0;
        return (1.0f * f) / denom;
    }
    static inline float3 PowerHeuristic(float p_1, float p_2, float3 f, float n_1 = 1, float n_2 = 1)
    {
        float a = n_1 * p_1;
        float b = n_2 * p_2;
        float denom = a * a + b * b;
        if (denom == 0) return f3(0.0f);
        return (n_1 * n_1 * p_1 * f) / denom;
    }

    // RT.hlsli:309-479
    struct RayDifferentials
    {
        float3 origin_x, dir_x, origin_y, dir_y;
        float4 uv_grads;

        static RayDifferentials Init(int px, int py, float2 renderDim, float tanHalfFOV, float aspectRatio,
            float2 jitter, float3 vbx, float3 vby, float3 vbz, bool thinLens, float focusDepth,
            float2 lensSample, float3 origin)
        {
            RayDifferentials ret;
            float3 dir_cs_x = GeneratePinholeCameraRay_CS(px + 1, py, renderDim, aspectRatio, tanHalfFOV, jitter);
            float3 dir_cs_y = GeneratePinholeCameraRay_CS(px, py - 1, renderDim, aspectRatio, tanHalfFOV, jitter);
            if (thinLens)
            {
                dir_cs_x = focusDepth * dir_cs_x - f3(lensSample.x, lensSample.y, 0);
                dir_cs_y = focusDepth * dir_cs_y - f3(lensSample.x, lensSample.y, 0);
            }
            ret.dir_x = normalize(mad3(dir_cs_x.x, vbx, mad3(dir_cs_x.y, vby, dir_cs_x.z * vbz)));
            ret.dir_y = normalize(mad3(dir_cs_y.x, vbx, mad3(dir_cs_y.y, vby, dir_cs_y.z * vbz)));
            ret.origin_x = origin; ret.origin_y = origin;
            // note: the reference leaves uv_grads uninitialised here (RT.hlsli:323-353); it is always written by
            // ComputeUVDifferentials before it is read, except for the FLT16_MAX test inside that function which
            // reads the previous value.  The ABI pins the initial value to 0.
            ret.uv_grads = {0, 0, 0, 0};
            return ret;
        }

        void dpdx_dpdy(float3 hitPoint, float3 normal, float3& dpdx, float3& dpdy) const
        {
            float d = dot(normal, hitPoint);
            float numerator_x = d - dot(normal, origin_x);
            float denom_x = dot(normal, dir_x);
            float t_x = numerator_x / denom_x;
            float3 hitPoint_x = mad3(t_x, dir_x, origin_x);
            float numerator_y = d - dot(normal, origin_y);
            float denom_y = dot(normal, dir_y);
            float t_y = numerator_y / denom_y;
            float3 hitPoint_y = mad3(t_y, dir_y, origin_y);
            dpdx = denom_x != 0 ? hitPoint_x - hitPoint : f3(ZR_FLT16_MAX);
            dpdy = denom_y != 0 ? hitPoint_y - hitPoint : f3(ZR_FLT16_MAX);
        }

        void UpdateRays(float3 p, float3 normal, float3 wi, float3 wo, const Math::TriDifferentials& triDiffs,
            float3 dpdx, float3 dpdy, bool transmitted, float eta)
        {
            origin_x = p + dpdx;
            origin_y = p + dpdy;
            float3 dwodx = -dir_x - wo;
            float3 dwody = -dir_y - wo;
            float3 dndx = triDiffs.dndu * uv_grads.x + triDiffs.dndv * uv_grads.y;
            float3 dndy = triDiffs.dndu * uv_grads.z + triDiffs.dndv * uv_grads.w;
            float dndotWodx = dot(dndx, wo) + dot(normal, dwodx);
            float dndotWody = dot(dndy, wo) + dot(normal, dwody);
            float ndotwo = dot(normal, wo);
            if (!transmitted)
            {
                dir_x = wi + mad3(2.0f, mad3(dndotWodx, normal, ndotwo * dndx), -dwodx);
                dir_y = wi + mad3(2.0f, mad3(dndotWody, normal, ndotwo * dndy), -dwody);
            }
            else
            {
                float eta_relative = 1.0f / eta;
                float ndotwi = zr_abs(dot(normal, wi));
                float q = zr_fma(eta_relative, ndotwo, -ndotwi);
                float common = zr_fma(eta_relative, -ndotwo / ndotwi, 1.0f);
                float dqdx = eta_relative * dndotWodx * common;
                float dqdy = eta_relative * dndotWody * common;
                dir_x = mad3(-eta_relative, dwodx, wi) + mad3(q, dndx, dqdx * normal);
                dir_y = mad3(-eta_relative, dwody, wi) + mad3(q, dndy, dqdy * normal);
            }
        }

        void ComputeUVDifferentials(float3 dpdx, float3 dpdy, float3 dpdu, float3 dpdv)
        {
            float dpduDotdpdu = dot(dpdu, dpdu);
            float dpdvDotdpdv = dot(dpdv, dpdv);
            float dpduDotdpdv = dot(dpdu, dpdv);
            float det = dpduDotdpdu * dpdvDotdpdv - dpduDotdpdv * dpduDotdpdv;
            if (zr_abs(det) < 1e-7f) { uv_grads = {0, 0, 0, 0}; return; }
            float2 b_x = {dot(dpdu, dpdx), dot(dpdv, dpdx)};
            // mul(float2x2(dvv, -duv, -duv, duu), b) / det
            float2 grads_x = {(dpdvDotdpdv * b_x.x + -dpduDotdpdv * b_x.y) / det,
                              (-dpduDotdpdv * b_x.x + dpduDotdpdu * b_x.y) / det};
            float2 b_y = {dot(dpdu, dpdy), dot(dpdv, dpdy)};
            float2 grads_y = {(dpdvDotdpdv * b_y.x + -dpduDotdpdv * b_y.y) / det,
                              (-dpduDotdpdv * b_y.x + dpduDotdpdu * b_y.y) / det};
            bool invalid_x = (uv_grads.x == ZR_FLT16_MAX) || (dpdx.x == ZR_FLT16_MAX);
            bool invalid_y = (uv_grads.z == ZR_FLT16_MAX) || (dpdy.x == ZR_FLT16_MAX);
            uv_grads.x = invalid_x ? ZR_FLT16_MAX : grads_x.x;
            uv_grads.y = invalid_x ? ZR_FLT16_MAX : grads_x.y;
            uv_grads.z = invalid_y ? ZR_FLT16_MAX : grads_y.x;
            uv_grads.w = invalid_y ? ZR_FLT16_MAX : grads_y.y;
        }
    };
}

} // namespace zro
