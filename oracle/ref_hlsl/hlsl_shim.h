// ORACLE tooling -- test infrastructure only.  Nothing under oracle/ is linked into, imported by, or called from the product library.
//
// hlsl_shim.h: the HLSL-2021 language surface (vector / matrix types with swizzles, intrinsics, resource objects) as C++, so
// that g++ can compile the REFERENCE's own shader headers -- rewritten only lexically by hlsl2cpp.py into oracle/_ref/gen/ --
// and the oracle (and the HIP stage functions run on the host) can be pinned against the reference's code instead of against
// a second transcription of it.
//
// Arithmetic contract: every intrinsic maps onto include/zr_detmath.h, i.e. the SAME definitions the ABI pins for the product
// and the oracle (mad = one fused fma; dot summed left to right; normalize(v) = v * (1 / sqrt(dot(v, v))); rsqrt = 1 / sqrt;
// min / max / saturate with HLSL NaN behaviour; Cephes transcendentals; RTNE half conversions).  What this build therefore
// pins is everything the reference's SOURCE determines -- operation order, constants, branches, RNG consumption, packing --
// which is exactly what a restatement can get wrong.  It cannot pin the vendor compiler's transcendental implementations or
// the driver's traversal (unpinnable).
#pragma once
#include <cstdint>
#include <cstddef>
#include <cstring>
#include <type_traits>
#include "../../include/zr_detmath.h"

namespace hlsl {

typedef uint32_t uint;
typedef uint32_t dword;

// ------------------------------------------------------------------------------------------------ half (true fp16: -enable-16bit-types)
struct half
{
    uint16_t bits;
    half() = default;
    half(float f) : bits(zr_f32_to_f16(f)) {}
    operator float() const { return zr_f16_to_f32(bits); }
    half& operator+=(float o) { *this = half((float)*this + o); return *this; }
    half& operator-=(float o) { *this = half((float)*this - o); return *this; }
    half& operator*=(float o) { *this = half((float)*this * o); return *this; }
    half& operator/=(float o) { *this = half((float)*this / o); return *this; }
};
typedef half float16_t;
typedef half min16float;

// element casts: float -> integer follows the D3D rule for ftou / ftoi (D3D11.3 functional spec 22.13: the input is clamped to the target
// range, NaN -> 0) -- what the hardware does and what the ABI's zr_f2u_sat / zr_f2i_sat define; everything else is the C++ conversion
template<class T, class U> struct elem_cast { static T go(const U& u) { return (T)u; } };
template<> struct elem_cast<uint32_t, float> { static uint32_t go(const float& f) { return zr_f2u_sat(f); } };
template<> struct elem_cast<int32_t, float> { static int32_t go(const float& f) { return zr_f2i_sat(f); } };
template<> struct elem_cast<uint16_t, float> { static uint16_t go(const float& f) { const uint32_t v = zr_f2u_sat(f); return (uint16_t)(v > 0xffffu ? 0xffffu : v); } };
template<> struct elem_cast<int16_t, float> { static int16_t go(const float& f) { const int32_t v = zr_f2i_sat(f); return (int16_t)(v > 32767 ? 32767 : (v < -32768 ? -32768 : v)); } };

// ------------------------------------------------------------------------------------------------ vectors with swizzles
template<class T, int N> struct vec;

// element conversions HLSL applies implicitly and that are value-preserving or sign-reinterpreting: half -> float, int <-> uint,
// 16-bit unsigned -> 32-bit.  Everything else (float <-> int) needs the cast the reference writes.
template<class U, class T> struct implicit_vec_conv : std::false_type {};
template<> struct implicit_vec_conv<half, float> : std::true_type {};
template<> struct implicit_vec_conv<int32_t, uint32_t> : std::true_type {};
template<> struct implicit_vec_conv<uint32_t, int32_t> : std::true_type {};
template<> struct implicit_vec_conv<uint16_t, uint32_t> : std::true_type {};
template<> struct implicit_vec_conv<uint16_t, int32_t> : std::true_type {};
template<> struct implicit_vec_conv<bool, int32_t> : std::true_type {};
template<> struct implicit_vec_conv<bool, uint32_t> : std::true_type {};
template<> struct implicit_vec_conv<uint32_t, float> : std::true_type {};      // integer -> float (DXC converts silently; used for pixel coordinates)
template<> struct implicit_vec_conv<int32_t, float> : std::true_type {};
template<> struct implicit_vec_conv<uint16_t, float> : std::true_type {};
template<> struct implicit_vec_conv<int16_t, uint32_t> : std::true_type {};
template<> struct implicit_vec_conv<int16_t, int32_t> : std::true_type {};
template<> struct implicit_vec_conv<int16_t, float> : std::true_type {};

template<class T, int N, int... I> struct Swz
{
    T d[N];
    static constexpr int K = sizeof...(I);
    typedef vec<T, K> V;
    operator V() const { return V(d[I]...); }
    // a half swizzle promotes to a float vector
    template<class T2 = T, class = typename std::enable_if<std::is_same<T2, half>::value>::type> operator vec<float, K>() const { return vec<float, K>((float)d[I]...); }
    Swz& operator=(const V& v) { const int idx[K] = {I...}; V t(v); for (int k = 0; k < K; k++) d[idx[k]] = t.d[k]; return *this; }
    Swz& operator=(const Swz& o) { return *this = (V)o; }
#define HLSL_SWZ_COMPOUND(op) Swz& operator op##=(const V& v) { const int idx[K] = {I...}; V t(v); for (int k = 0; k < K; k++) d[idx[k]] op##= t.d[k]; return *this; }
    HLSL_SWZ_COMPOUND(+) HLSL_SWZ_COMPOUND(-) HLSL_SWZ_COMPOUND(*) HLSL_SWZ_COMPOUND(/)
    HLSL_SWZ_COMPOUND(|) HLSL_SWZ_COMPOUND(&) HLSL_SWZ_COMPOUND(^) HLSL_SWZ_COMPOUND(<<) HLSL_SWZ_COMPOUND(>>)
#undef HLSL_SWZ_COMPOUND
    T operator[](int i) const { const int idx[K] = {I...}; return d[idx[i]]; }
};

#define HLSL_VEC_COMMON(T, NN) \
    static constexpr int N = NN; typedef T elem; \
    vec() { for (int i = 0; i < NN; i++) d[i] = T(); } \
    vec(const vec& o) { for (int i = 0; i < NN; i++) d[i] = o.d[i]; } \
    vec& operator=(const vec& o) { for (int i = 0; i < NN; i++) d[i] = o.d[i]; return *this; } \
    template<class U, class = typename std::enable_if<std::is_convertible<U, T>::value && !std::is_same<U, T>::value && std::is_arithmetic<U>::value>::type> \
    vec(U s) { for (int i = 0; i < NN; i++) d[i] = (T)s; } \
    /* scalar broadcast: implicit, except for half (a half scalar must not turn into a half vector inside scalar arithmetic) */ \
    template<class T2 = T, class = typename std::enable_if<!std::is_same<T2, half>::value>::type> vec(T s) { for (int i = 0; i < NN; i++) d[i] = s; } \
    template<class T2 = T, class = typename std::enable_if<std::is_same<T2, half>::value>::type, class = void> explicit vec(T s) { for (int i = 0; i < NN; i++) d[i] = s; } \
    template<class U, class = typename std::enable_if<!implicit_vec_conv<U, T>::value>::type> \
    explicit vec(const vec<U, NN>& o) { for (int i = 0; i < NN; i++) d[i] = elem_cast<T, U>::go(o.d[i]); } \
    template<class U, class = typename std::enable_if<implicit_vec_conv<U, T>::value>::type, class = void> \
    vec(const vec<U, NN>& o) { for (int i = 0; i < NN; i++) d[i] = (T)o.d[i]; }     /* promotions HLSL applies silently */ \
    template<class U, int M, int... J, class = typename std::enable_if<sizeof...(J) == NN && !std::is_same<U, T>::value>::type> \
    explicit vec(const Swz<U, M, J...>& o) { vec<U, NN> t = o; for (int i = 0; i < NN; i++) d[i] = elem_cast<T, U>::go(t.d[i]); } \
    T& operator[](int i) { return d[i]; } \
    const T& operator[](int i) const { return d[i]; }

template<class T> struct vec<T, 1>
{
    union { T d[1]; struct { T x; }; struct { T r; }; };
    HLSL_VEC_COMMON(T, 1)
    operator T() const { return d[0]; }
};

template<class T> struct vec<T, 2>
{
    union
    {
        T d[2];
        struct { T x, y; };
        struct { T r, g; };
#include "hlsl_swizzles_2.inc"
    };
    HLSL_VEC_COMMON(T, 2)
    vec(T a, T b) { d[0] = a; d[1] = b; }
};

template<class T> struct vec<T, 3>
{
    union
    {
        T d[3];
        struct { T x, y, z; };
        struct { T r, g, b; };
#include "hlsl_swizzles_3.inc"
    };
    HLSL_VEC_COMMON(T, 3)
    vec(T a, T b, T c) { d[0] = a; d[1] = b; d[2] = c; }
    vec(const vec<T, 2>& a, T c) { d[0] = a.d[0]; d[1] = a.d[1]; d[2] = c; }
    vec(T a, const vec<T, 2>& b) { d[0] = a; d[1] = b.d[0]; d[2] = b.d[1]; }
};

template<class T> struct vec<T, 4>
{
    union
    {
        T d[4];
        struct { T x, y, z, w; };
        struct { T r, g, b, a; };
#include "hlsl_swizzles_4.inc"
    };
    HLSL_VEC_COMMON(T, 4)
    vec(T a_, T b_, T c, T e) { d[0] = a_; d[1] = b_; d[2] = c; d[3] = e; }
    vec(const vec<T, 3>& v, T e) { d[0] = v.d[0]; d[1] = v.d[1]; d[2] = v.d[2]; d[3] = e; }
    vec(T a_, const vec<T, 3>& v) { d[0] = a_; d[1] = v.d[0]; d[2] = v.d[1]; d[3] = v.d[2]; }
    vec(const vec<T, 2>& v, T c, T e) { d[0] = v.d[0]; d[1] = v.d[1]; d[2] = c; d[3] = e; }
    vec(const vec<T, 2>& v, const vec<T, 2>& u) { d[0] = v.d[0]; d[1] = v.d[1]; d[2] = u.d[0]; d[3] = u.d[1]; }
    vec(T a_, T b_, const vec<T, 2>& u) { d[0] = a_; d[1] = b_; d[2] = u.d[0]; d[3] = u.d[1]; }
    vec(T a_, const vec<T, 2>& u, T e) { d[0] = a_; d[1] = u.d[0]; d[2] = u.d[1]; d[3] = e; }
    // constructor-style casts with operands of another element type: half4(float3, half), float4(half3, float) ...
    template<class U, class S, class = typename std::enable_if<!std::is_same<U, T>::value && (std::is_arithmetic<S>::value || std::is_same<S, half>::value)>::type>
    vec(const vec<U, 3>& v, S e) { d[0] = (T)v.d[0]; d[1] = (T)v.d[1]; d[2] = (T)v.d[2]; d[3] = (T)e; }
};

#define HLSL_TYPEDEFS(T, name) typedef vec<T, 1> name##1; typedef vec<T, 2> name##2; typedef vec<T, 3> name##3; typedef vec<T, 4> name##4;
HLSL_TYPEDEFS(float, float)
HLSL_TYPEDEFS(uint32_t, uint)
HLSL_TYPEDEFS(int32_t, int)
HLSL_TYPEDEFS(bool, bool)
HLSL_TYPEDEFS(uint16_t, uint16_t)
HLSL_TYPEDEFS(int16_t, int16_t)
HLSL_TYPEDEFS(half, half)
HLSL_TYPEDEFS(half, float16_t)
HLSL_TYPEDEFS(uint32_t, uint32_t)
HLSL_TYPEDEFS(int32_t, int32_t)
#undef HLSL_TYPEDEFS

// ---- operators: plain (non-template) overloads per concrete vector type, so that swizzle proxies and scalars convert implicitly
#define HLSL_BIN(V, R, op) \
    inline R operator op(const V& a, const V& b) { R r; for (int i = 0; i < V::N; i++) r.d[i] = a.d[i] op b.d[i]; return r; }
#define HLSL_CMPD(V, op) \
    inline V& operator op##=(V& a, const V& b) { for (int i = 0; i < V::N; i++) a.d[i] = a.d[i] op b.d[i]; return a; }
#define HLSL_ARITH(V, B) \
    HLSL_BIN(V, V, +) HLSL_BIN(V, V, -) HLSL_BIN(V, V, *) HLSL_BIN(V, V, /) \
    HLSL_CMPD(V, +) HLSL_CMPD(V, -) HLSL_CMPD(V, *) HLSL_CMPD(V, /) \
    HLSL_BIN(V, B, <) HLSL_BIN(V, B, >) HLSL_BIN(V, B, <=) HLSL_BIN(V, B, >=) HLSL_BIN(V, B, ==) HLSL_BIN(V, B, !=) \
    inline V operator-(const V& a) { V r; for (int i = 0; i < V::N; i++) r.d[i] = -a.d[i]; return r; } \
    inline V operator+(const V& a) { return a; }
#define HLSL_BITS(V) \
    HLSL_BIN(V, V, |) HLSL_BIN(V, V, &) HLSL_BIN(V, V, ^) HLSL_BIN(V, V, <<) HLSL_BIN(V, V, >>) HLSL_BIN(V, V, %) \
    HLSL_CMPD(V, |) HLSL_CMPD(V, &) HLSL_CMPD(V, ^) HLSL_CMPD(V, <<) HLSL_CMPD(V, >>) HLSL_CMPD(V, %) \
    inline V operator~(const V& a) { V r; for (int i = 0; i < V::N; i++) r.d[i] = ~a.d[i]; return r; }
HLSL_ARITH(float2, bool2) HLSL_ARITH(float3, bool3) HLSL_ARITH(float4, bool4)
HLSL_ARITH(uint2, bool2) HLSL_ARITH(uint3, bool3) HLSL_ARITH(uint4, bool4)
HLSL_ARITH(int2, bool2) HLSL_ARITH(int3, bool3) HLSL_ARITH(int4, bool4)
HLSL_ARITH(uint16_t2, bool2) HLSL_ARITH(uint16_t3, bool3) HLSL_ARITH(uint16_t4, bool4)
HLSL_BITS(uint2) HLSL_BITS(uint3) HLSL_BITS(uint4) HLSL_BITS(int2) HLSL_BITS(int3) HLSL_BITS(int4)
HLSL_BITS(uint16_t2) HLSL_BITS(uint16_t3) HLSL_BITS(uint16_t4)
// half vectors: arithmetic in float, rounded to fp16 by the assignment back into a half
#define HLSL_HALF_ARITH(V, B) \
    inline V operator+(const V& a, const V& b) { V r; for (int i = 0; i < V::N; i++) r.d[i] = half((float)a.d[i] + (float)b.d[i]); return r; } \
    inline V operator-(const V& a, const V& b) { V r; for (int i = 0; i < V::N; i++) r.d[i] = half((float)a.d[i] - (float)b.d[i]); return r; } \
    inline V operator*(const V& a, const V& b) { V r; for (int i = 0; i < V::N; i++) r.d[i] = half((float)a.d[i] * (float)b.d[i]); return r; } \
    inline V operator/(const V& a, const V& b) { V r; for (int i = 0; i < V::N; i++) r.d[i] = half((float)a.d[i] / (float)b.d[i]); return r; }
HLSL_HALF_ARITH(half2, bool2) HLSL_HALF_ARITH(half3, bool3) HLSL_HALF_ARITH(half4, bool4)
// half vector (op) float scalar / float vector: promotes to floatN
#define HLSL_HALF_MIX(V, F) \
    HLSL_MIX1(V, F, float, F, +) HLSL_MIX1(V, F, float, F, -) HLSL_MIX1(V, F, float, F, *) HLSL_MIX1(V, F, float, F, /) \
    HLSL_MIXV(V, F, +) HLSL_MIXV(V, F, -) HLSL_MIXV(V, F, *) HLSL_MIXV(V, F, /)
// integer vector (op) scalar: an integer scalar keeps the vector's type, a float scalar / float vector promotes to floatN (what DXC does)
#define HLSL_MIX1(V, F, S, R, op) \
    inline R operator op(const V& a, S b) { R r; for (int i = 0; i < V::N; i++) r.d[i] = (typename R::elem)a.d[i] op (typename R::elem)b; return r; } \
    inline R operator op(S a, const V& b) { R r; for (int i = 0; i < V::N; i++) r.d[i] = (typename R::elem)a op (typename R::elem)b.d[i]; return r; }
#define HLSL_MIXV(V, F, op) \
    inline F operator op(const V& a, const F& b) { F r; for (int i = 0; i < V::N; i++) r.d[i] = (float)a.d[i] op b.d[i]; return r; } \
    inline F operator op(const F& a, const V& b) { F r; for (int i = 0; i < V::N; i++) r.d[i] = a.d[i] op (float)b.d[i]; return r; }
#define HLSL_MIX(V, F) \
    HLSL_MIX1(V, F, float, F, +) HLSL_MIX1(V, F, float, F, -) HLSL_MIX1(V, F, float, F, *) HLSL_MIX1(V, F, float, F, /) \
    HLSL_MIX1(V, F, int, V, +) HLSL_MIX1(V, F, int, V, -) HLSL_MIX1(V, F, int, V, *) HLSL_MIX1(V, F, int, V, /) HLSL_MIX1(V, F, int, V, %) \
    HLSL_MIX1(V, F, uint, V, +) HLSL_MIX1(V, F, uint, V, -) HLSL_MIX1(V, F, uint, V, *) HLSL_MIX1(V, F, uint, V, /) HLSL_MIX1(V, F, uint, V, %) \
    HLSL_MIX1(V, F, int, V, &) HLSL_MIX1(V, F, int, V, |) HLSL_MIX1(V, F, int, V, ^) HLSL_MIX1(V, F, int, V, <<) HLSL_MIX1(V, F, int, V, >>) \
    HLSL_MIX1(V, F, uint, V, &) HLSL_MIX1(V, F, uint, V, |) HLSL_MIX1(V, F, uint, V, ^) HLSL_MIX1(V, F, uint, V, <<) HLSL_MIX1(V, F, uint, V, >>) \
    HLSL_MIXV(V, F, +) HLSL_MIXV(V, F, -) HLSL_MIXV(V, F, *) HLSL_MIXV(V, F, /)
#define HLSL_FSCALAR(V) HLSL_MIX1(V, V, float, V, +) HLSL_MIX1(V, V, float, V, -) HLSL_MIX1(V, V, float, V, *) HLSL_MIX1(V, V, float, V, /)
HLSL_FSCALAR(float2) HLSL_FSCALAR(float3) HLSL_FSCALAR(float4)
#define HLSL_HSCALAR(V) HLSL_MIX1(V, V, half, V, +) HLSL_MIX1(V, V, half, V, -) HLSL_MIX1(V, V, half, V, *) HLSL_MIX1(V, V, half, V, /)
HLSL_HSCALAR(float2) HLSL_HSCALAR(float3) HLSL_HSCALAR(float4)
HLSL_MIX(uint2, float2) HLSL_MIX(uint3, float3) HLSL_MIX(uint4, float4)
HLSL_MIX(int2, float2) HLSL_MIX(int3, float3) HLSL_MIX(int4, float4)
HLSL_MIX(uint16_t2, float2) HLSL_MIX(uint16_t3, float3) HLSL_MIX(uint16_t4, float4)
HLSL_HALF_MIX(half2, float2) HLSL_HALF_MIX(half3, float3) HLSL_HALF_MIX(half4, float4)
#define HLSL_BOOLV(V) \
    inline V operator!(const V& a) { V r; for (int i = 0; i < V::N; i++) r.d[i] = !a.d[i]; return r; } \
    inline V hlsl_and(const V& a, const V& b) { V r; for (int i = 0; i < V::N; i++) r.d[i] = a.d[i] && b.d[i]; return r; } \
    inline V hlsl_or(const V& a, const V& b) { V r; for (int i = 0; i < V::N; i++) r.d[i] = a.d[i] || b.d[i]; return r; } \
    inline bool any(const V& a) { bool r = false; for (int i = 0; i < V::N; i++) r = r || a.d[i]; return r; } \
    inline bool all(const V& a) { bool r = true; for (int i = 0; i < V::N; i++) r = r && a.d[i]; return r; } \
    HLSL_BIN(V, V, ==) HLSL_BIN(V, V, !=)
HLSL_BOOLV(bool2) HLSL_BOOLV(bool3) HLSL_BOOLV(bool4)
inline bool any(bool a) { return a; }
inline bool all(bool a) { return a; }
inline bool hlsl_and(bool a, bool b) { return a && b; }
inline bool hlsl_or(bool a, bool b) { return a || b; }

// ------------------------------------------------------------------------------------------------ scalar intrinsics
inline uint asuint(float f) { return zr_asuint(f); }
inline uint asuint(uint u) { return u; }
inline uint asuint(int i) { return (uint)i; }
inline int asint(float f) { return (int)zr_asuint(f); }
inline int asint(uint u) { return (int)u; }
inline float asfloat(uint u) { return zr_asfloat(u); }
inline float asfloat(int i) { return zr_asfloat((uint)i); }
inline float asfloat(float f) { return f; }
inline uint16_t asuint16(half h) { return h.bits; }
inline half asfloat16(uint16_t u) { half h; h.bits = u; return h; }
inline float fma(float a, float b, float c) { return zr_fma(a, b, c); }
inline float sqrt(float x) { return zr_sqrt(x); }
inline float rsqrt(float x) { return zr_rsqrt(x); }
inline float rcp(float x) { return 1.0f / x; }
inline float abs(float x) { return zr_abs(x); }
inline int abs(int x) { return x < 0 ? -x : x; }
inline float saturate(float x) { return zr_saturate(x); }
inline float sin(float x) { return zr_sin(x); }
inline float cos(float x) { return zr_cos(x); }
inline float tan(float x) { float s, c; zr_sincos(x, &s, &c); return s / c; }
inline void sincos(float x, float& s, float& c) { zr_sincos(x, &s, &c); }
inline float exp(float x) { return zr_exp(x); }
inline float exp2(float x) { return zr_exp2(x); }
inline float log(float x) { return zr_log(x); }
inline float log2(float x) { return zr_log2(x); }
inline float pow(float x, float y) { return zr_pow(x, y); }
inline float atan(float x) { return zr_atan(x); }
inline float atan2(float y, float x) { return zr_atan2(y, x); }
inline float floor(float x) { return zr_floor(x); }
inline float ceil(float x) { return -zr_floor(-x); }
inline float frac(float x) { return x - zr_floor(x); }
inline float trunc(float x) { return x < 0.0f ? -zr_floor(-x) : zr_floor(x); }
inline float round(float x) { return __builtin_rintf(x); }          // HLSL round(): to nearest even
inline float fmod(float x, float y) { return x - y * trunc(x / y); }
inline float sign(float x) { return zr_sign(x); }
inline int sign(int x) { return x > 0 ? 1 : (x < 0 ? -1 : 0); }
inline float step(float y, float x) { return x >= y ? 1.0f : 0.0f; }
inline float lerp(float a, float b, float t) { return zr_lerp(a, b, t); }
inline float smoothstep(float a, float b, float x) { float t = zr_saturate((x - a) / (b - a)); return t * t * (3.0f - 2.0f * t); }
inline bool isnan(float x) { return zr_isnan(x) != 0; }
inline bool isinf(float x) { return zr_isinf(x) != 0; }
inline bool isfinite(float x) { return !zr_isnan(x) && !zr_isinf(x); }
inline uint f32tof16(float f) { return (uint)zr_f32_to_f16(f); }
inline float f16tof32(uint u) { return zr_f16_to_f32((uint16_t)(u & 0xffffu)); }
inline uint countbits(uint x) { return (uint)__builtin_popcount(x); }
inline uint firstbithigh(uint x) { return x ? 31u - (uint)__builtin_clz(x) : 0xffffffffu; }
inline uint firstbitlow(uint x) { return x ? (uint)__builtin_ctz(x) : 0xffffffffu; }
inline uint reversebits(uint x) { uint r = 0; for (int i = 0; i < 32; i++) r |= ((x >> i) & 1u) << (31 - i); return r; }

// min / max / clamp / select on scalars of mixed arithmetic types: the usual arithmetic conversions (what DXC applies);
// floats compare with the HLSL / ABI NaN behaviour of zr_min / zr_max (a < b ? a : b)
template<class A, class B, class = void> struct arith2 {};
template<class A, class B> struct arith2<A, B, typename std::enable_if<std::is_arithmetic<A>::value && std::is_arithmetic<B>::value>::type> { typedef typename std::common_type<A, B>::type type; };
template<> struct arith2<half, half, void> { typedef half type; };
template<class B> struct arith2<half, B, typename std::enable_if<std::is_arithmetic<B>::value>::type> { typedef float type; };
template<class A> struct arith2<A, half, typename std::enable_if<std::is_arithmetic<A>::value>::type> { typedef float type; };
template<class T> struct is_scalar_t : std::integral_constant<bool, std::is_arithmetic<T>::value || std::is_same<T, half>::value> {};
#define HLSL_SCALAR2(A, B) typename std::enable_if<is_scalar_t<A>::value && is_scalar_t<B>::value, typename arith2<A, B>::type>::type
template<class A, class B> inline HLSL_SCALAR2(A, B) min(A a, B b) { typedef typename arith2<A, B>::type T; T x = (T)a, y = (T)b; return x < y ? x : y; }
template<class A, class B> inline HLSL_SCALAR2(A, B) max(A a, B b) { typedef typename arith2<A, B>::type T; T x = (T)a, y = (T)b; return x > y ? x : y; }
template<class A, class B, class C> inline typename std::enable_if<is_scalar_t<A>::value && is_scalar_t<B>::value && is_scalar_t<C>::value, typename arith2<typename arith2<A, B>::type, C>::type>::type
clamp(A x, B lo, C hi) { typedef typename arith2<typename arith2<A, B>::type, C>::type T; return min(max((T)x, (T)lo), (T)hi); }
template<class A, class B> inline HLSL_SCALAR2(A, B) select(bool c, A a, B b) { typedef typename arith2<A, B>::type T; return c ? (T)a : (T)b; }
template<class T> inline typename std::enable_if<!is_scalar_t<T>::value, T>::type select(bool c, const T& a, const T& b) { return c ? a : b; }
// mad on scalars of mixed types: floating -> one fused fma in fp32 (the contract), integers -> a * b + c
template<class T> struct mad_impl { static T go(T a, T b, T c) { return (T)(a * b + c); } };
template<> struct mad_impl<float> { static float go(float a, float b, float c) { return zr_fma(a, b, c); } };
template<class A, class B, class C> inline typename std::enable_if<is_scalar_t<A>::value && is_scalar_t<B>::value && is_scalar_t<C>::value, typename arith2<typename arith2<A, B>::type, C>::type>::type
mad(A a, B b, C c) { typedef typename arith2<typename arith2<A, B>::type, C>::type T; return mad_impl<T>::go((T)a, (T)b, (T)c); }

template<int N, class A, class B> inline typename std::enable_if<is_scalar_t<A>::value && is_scalar_t<B>::value, vec<typename arith2<A, B>::type, N>>::type
select(const vec<bool, N>& c, A a, B b) { typedef typename arith2<A, B>::type T; vec<T, N> r; for (int i = 0; i < N; i++) r.d[i] = c.d[i] ? (T)a : (T)b; return r; }
template<int N> inline vec<half, N> asfloat16(const vec<uint16_t, N>& u) { vec<half, N> r; for (int i = 0; i < N; i++) r.d[i] = asfloat16(u.d[i]); return r; }
template<int N> inline vec<uint16_t, N> asuint16(const vec<half, N>& h) { vec<uint16_t, N> r; for (int i = 0; i < N; i++) r.d[i] = h.d[i].bits; return r; }

// ------------------------------------------------------------------------------------------------ vector intrinsics
#define HLSL_MAP1(V, f) inline V f(const V& a) { V r; for (int i = 0; i < V::N; i++) r.d[i] = f(a.d[i]); return r; }
#define HLSL_MAP2(V, f) inline V f(const V& a, const V& b) { V r; for (int i = 0; i < V::N; i++) r.d[i] = f(a.d[i], b.d[i]); return r; }
#define HLSL_MAP3(V, f) inline V f(const V& a, const V& b, const V& c) { V r; for (int i = 0; i < V::N; i++) r.d[i] = f(a.d[i], b.d[i], c.d[i]); return r; }
#define HLSL_FLOATV(V, B, U, I) \
    HLSL_MAP1(V, sqrt) HLSL_MAP1(V, rsqrt) HLSL_MAP1(V, rcp) HLSL_MAP1(V, abs) HLSL_MAP1(V, saturate) HLSL_MAP1(V, sin) HLSL_MAP1(V, cos) \
    HLSL_MAP1(V, exp) HLSL_MAP1(V, exp2) HLSL_MAP1(V, log) HLSL_MAP1(V, log2) HLSL_MAP1(V, floor) HLSL_MAP1(V, ceil) HLSL_MAP1(V, frac) \
    HLSL_MAP1(V, trunc) HLSL_MAP1(V, round) HLSL_MAP1(V, sign) HLSL_MAP1(V, atan) \
    HLSL_MAP2(V, min) HLSL_MAP2(V, max) HLSL_MAP2(V, pow) HLSL_MAP2(V, step) HLSL_MAP2(V, atan2) HLSL_MAP2(V, fmod) \
    HLSL_MAP3(V, mad) HLSL_MAP3(V, clamp) HLSL_MAP3(V, lerp) HLSL_MAP3(V, smoothstep) \
    inline B isnan(const V& a) { B r; for (int i = 0; i < V::N; i++) r.d[i] = isnan(a.d[i]); return r; } \
    inline B isinf(const V& a) { B r; for (int i = 0; i < V::N; i++) r.d[i] = isinf(a.d[i]); return r; } \
    inline U asuint(const V& a) { U r; for (int i = 0; i < V::N; i++) r.d[i] = asuint(a.d[i]); return r; } \
    inline I asint(const V& a) { I r; for (int i = 0; i < V::N; i++) r.d[i] = asint(a.d[i]); return r; } \
    inline V asfloat(const U& a) { V r; for (int i = 0; i < V::N; i++) r.d[i] = asfloat(a.d[i]); return r; } \
    inline V asfloat(const I& a) { V r; for (int i = 0; i < V::N; i++) r.d[i] = asfloat(a.d[i]); return r; } \
    inline U f32tof16(const V& a) { U r; for (int i = 0; i < V::N; i++) r.d[i] = f32tof16(a.d[i]); return r; } \
    inline V f16tof32(const U& a) { V r; for (int i = 0; i < V::N; i++) r.d[i] = f16tof32(a.d[i]); return r; } \
    inline V select(const B& c, const V& a, const V& b) { V r; for (int i = 0; i < V::N; i++) r.d[i] = c.d[i] ? a.d[i] : b.d[i]; return r; } \
    inline float dot(const V& a, const V& b) { float s = a.d[0] * b.d[0]; for (int i = 1; i < V::N; i++) s = s + a.d[i] * b.d[i]; return s; } \
    inline float length(const V& a) { return zr_sqrt(dot(a, a)); } \
    inline float distance(const V& a, const V& b) { return length(a - b); } \
    inline V normalize(const V& a) { float inv = 1.0f / zr_sqrt(dot(a, a)); return a * V(inv); } \
    inline bool any(const V& a) { bool r = false; for (int i = 0; i < V::N; i++) r = r || (a.d[i] != 0.0f); return r; } \
    inline bool all(const V& a) { bool r = true; for (int i = 0; i < V::N; i++) r = r && (a.d[i] != 0.0f); return r; }
HLSL_FLOATV(float2, bool2, uint2, int2) HLSL_FLOATV(float3, bool3, uint3, int3) HLSL_FLOATV(float4, bool4, uint4, int4)
// mad with scalar operands (a float or half scalar broadcasts)
#define HLSL_MADS(V) \
    inline V mad(const V& a, float b, const V& c) { V r; for (int i = 0; i < V::N; i++) r.d[i] = zr_fma(a.d[i], b, c.d[i]); return r; } \
    inline V mad(float a, const V& b, const V& c) { V r; for (int i = 0; i < V::N; i++) r.d[i] = zr_fma(a, b.d[i], c.d[i]); return r; } \
    inline V mad(const V& a, const V& b, float c) { V r; for (int i = 0; i < V::N; i++) r.d[i] = zr_fma(a.d[i], b.d[i], c); return r; } \
    inline V mad(const V& a, float b, float c) { V r; for (int i = 0; i < V::N; i++) r.d[i] = zr_fma(a.d[i], b, c); return r; } \
    inline V mad(float a, const V& b, float c) { V r; for (int i = 0; i < V::N; i++) r.d[i] = zr_fma(a, b.d[i], c); return r; } \
    inline V mad(float a, float b, const V& c) { V r; for (int i = 0; i < V::N; i++) r.d[i] = zr_fma(a, b, c.d[i]); return r; }
HLSL_MADS(float2) HLSL_MADS(float3) HLSL_MADS(float4)
#define HLSL_INTV(V, B) \
    HLSL_MAP2(V, min) HLSL_MAP2(V, max) HLSL_MAP3(V, clamp) HLSL_MAP3(V, mad) \
    inline V select(const B& c, const V& a, const V& b) { V r; for (int i = 0; i < V::N; i++) r.d[i] = c.d[i] ? a.d[i] : b.d[i]; return r; } \
    inline bool any(const V& a) { bool r = false; for (int i = 0; i < V::N; i++) r = r || (a.d[i] != 0); return r; } \
    inline bool all(const V& a) { bool r = true; for (int i = 0; i < V::N; i++) r = r && (a.d[i] != 0); return r; }
HLSL_INTV(uint2, bool2) HLSL_INTV(uint3, bool3) HLSL_INTV(uint4, bool4) HLSL_INTV(int2, bool2) HLSL_INTV(int3, bool3) HLSL_INTV(int4, bool4)
HLSL_INTV(uint16_t2, bool2) HLSL_INTV(uint16_t3, bool3) HLSL_INTV(uint16_t4, bool4)
HLSL_MAP1(int2, abs) HLSL_MAP1(int3, abs) HLSL_MAP1(int4, abs)
HLSL_MAP1(uint2, countbits) HLSL_MAP1(uint3, countbits) HLSL_MAP1(uint4, countbits)
HLSL_MAP1(uint2, firstbitlow) HLSL_MAP1(uint3, firstbitlow) HLSL_MAP1(uint4, firstbitlow)
inline uint2 asuint(const uint2& a) { return a; }
inline uint3 asuint(const uint3& a) { return a; }
inline uint4 asuint(const uint4& a) { return a; }
inline float3 cross(const float3& a, const float3& b) { return float3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
inline float3 reflect(const float3& i, const float3& n) { return i - 2.0f * dot(n, i) * n; }
inline float3 refract(const float3& i, const float3& n, float eta)
{
    float ndoti = dot(n, i);
    float k = 1.0f - eta * eta * (1.0f - ndoti * ndoti);
    if (k < 0.0f) return float3(0.0f);
    return eta * i - (eta * ndoti + zr_sqrt(k)) * n;
}
inline float dot(float a, float b) { return a * b; }
// dot(1, bool4): the number of set components (ReSTIR_PT_Sort.hlsl counts the 2 x 2 pixels of a thread per bucket this way)
inline uint32_t dot(int a, const vec<bool, 4>& b) { return (uint32_t)a * ((uint32_t)b.d[0] + (uint32_t)b.d[1] + (uint32_t)b.d[2] + (uint32_t)b.d[3]); }
// select(bool vector, half vector ...)
inline half2 select(const bool2& c, const half2& a, const half2& b) { half2 r; for (int i = 0; i < 2; i++) r.d[i] = c.d[i] ? a.d[i] : b.d[i]; return r; }
inline half3 select(const bool3& c, const half3& a, const half3& b) { half3 r; for (int i = 0; i < 3; i++) r.d[i] = c.d[i] ? a.d[i] : b.d[i]; return r; }
inline bool2 select(const bool2& c, const bool2& a, const bool2& b) { bool2 r; for (int i = 0; i < 2; i++) r.d[i] = c.d[i] ? a.d[i] : b.d[i]; return r; }
inline bool3 select(const bool3& c, const bool3& a, const bool3& b) { bool3 r; for (int i = 0; i < 3; i++) r.d[i] = c.d[i] ? a.d[i] : b.d[i]; return r; }

// ------------------------------------------------------------------------------------------------ matrices (row-major storage: M.m(r, c), M[r] = row r)
template<int R, int C> struct mat
{
    vec<float, C> rows[R];
    mat() {}
    mat(float s) { for (int r = 0; r < R; r++) rows[r] = vec<float, C>(s); }
    template<class... A, class = typename std::enable_if<sizeof...(A) == R * C && (R * C > 1)>::type>
    mat(A... a) { const float v[R * C] = {(float)a...}; for (int r = 0; r < R; r++) for (int c = 0; c < C; c++) rows[r].d[c] = v[r * C + c]; }
    template<int RR = R, class = typename std::enable_if<RR == 2>::type> mat(const vec<float, C>& a, const vec<float, C>& b) { rows[0] = a; rows[1] = b; }
    template<int RR = R, class = typename std::enable_if<RR == 3>::type> mat(const vec<float, C>& a, const vec<float, C>& b, const vec<float, C>& c) { rows[0] = a; rows[1] = b; rows[2] = c; }
    template<int RR = R, class = typename std::enable_if<RR == 4>::type> mat(const vec<float, C>& a, const vec<float, C>& b, const vec<float, C>& c, const vec<float, C>& e) { rows[0] = a; rows[1] = b; rows[2] = c; rows[3] = e; }
    // (float3x3)M: truncating cast from a larger matrix
    template<int R2, int C2, class = typename std::enable_if<(R2 >= R && C2 >= C && (R2 != R || C2 != C))>::type>
    explicit mat(const mat<R2, C2>& o) { for (int r = 0; r < R; r++) for (int c = 0; c < C; c++) rows[r].d[c] = o.rows[r].d[c]; }
    float& m(int r, int c) { return rows[r].d[c]; }
    float m(int r, int c) const { return rows[r].d[c]; }
    vec<float, C>& operator[](int r) { return rows[r]; }
    const vec<float, C>& operator[](int r) const { return rows[r]; }
};
typedef mat<2, 2> float2x2; typedef mat<3, 3> float3x3; typedef mat<3, 4> float3x4; typedef mat<4, 3> float4x3; typedef mat<4, 4> float4x4;
// mul(M, v): each row dotted with v (left-to-right sum); mul(v, M): sum over rows of v[r] * row r (left to right)
template<int R, int C> inline vec<float, R> mul(const mat<R, C>& M, const vec<float, C>& v)
{ vec<float, R> o; for (int r = 0; r < R; r++) { float s = M.rows[r].d[0] * v.d[0]; for (int c = 1; c < C; c++) s = s + M.rows[r].d[c] * v.d[c]; o.d[r] = s; } return o; }
template<int R, int C> inline vec<float, C> mul(const vec<float, R>& v, const mat<R, C>& M)
{ vec<float, C> o; for (int c = 0; c < C; c++) { float s = v.d[0] * M.rows[0].d[c]; for (int r = 1; r < R; r++) s = s + v.d[r] * M.rows[r].d[c]; o.d[c] = s; } return o; }
template<int R, int K, int C> inline mat<R, C> mul(const mat<R, K>& A, const mat<K, C>& B)
{ mat<R, C> o; for (int r = 0; r < R; r++) for (int c = 0; c < C; c++) { float s = A.rows[r].d[0] * B.rows[0].d[c]; for (int k = 1; k < K; k++) s = s + A.rows[r].d[k] * B.rows[k].d[c]; o.rows[r].d[c] = s; } return o; }
template<int R, int C> inline mat<C, R> transpose(const mat<R, C>& M) { mat<C, R> o; for (int r = 0; r < R; r++) for (int c = 0; c < C; c++) o.rows[c].d[r] = M.rows[r].d[c]; return o; }
template<int R, int C> inline mat<R, C> operator*(const mat<R, C>& M, float s) { mat<R, C> o; for (int r = 0; r < R; r++) o.rows[r] = M.rows[r] * vec<float, C>(s); return o; }
// swizzle proxies as mul operands
template<int R, int C, int M, int... J> inline vec<float, R> mul(const mat<R, C>& A, const Swz<float, M, J...>& v) { return mul(A, (vec<float, C>)v); }
template<int R, int C, int M, int... J> inline vec<float, C> mul(const Swz<float, M, J...>& v, const mat<R, C>& A) { return mul((vec<float, R>)v, A); }

} // namespace hlsl
