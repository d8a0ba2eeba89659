// C entry points over the REFERENCE's own compiled code (Source/ZetaCore/Math/{Common,Sampling}.cpp, Vector.h,
// OctahedralVector.h, Utility/RNG.h), built by oracle/_ref.mk into oracle/_ref/libzref.so.  Used only to pin the
// oracle (tests/test_ref_pins.py) and to generate tests/golden/ref_*.npz (tools/make_ref_goldens.py).
#include <Math/Sampling.h>
#include <Math/Common.h>
#include <Math/Vector.h>
#include <Math/OctahedralVector.h>
#include <Utility/RNG.h>
#include <Utility/Span.h>
#include <vector>

using namespace ZetaRay;

extern "C" {
float zref_kahan_sum(float* d, uint64_t n) { return Math::KahanSum(Util::Span<float>(d, n)); }
void zref_alias_normalize(float* w, uint64_t n) { Math::AliasTable_Normalize(Util::MutableSpan<float>(w, n)); }
// AliasTable_Build (Sampling.cpp:52-140): out = n x (P_Curr, P_Orig, Alias-as-float-bits)
void zref_alias_build(float* w, uint64_t n, float* p_curr, float* p_orig, uint32_t* alias)
{
    std::vector<Math::AliasTableEntry> t(n);
    Math::AliasTable_Build(Util::MutableSpan<float>(w, n), Util::MutableSpan<Math::AliasTableEntry>(t.data(), n));
    for (uint64_t i = 0; i < n; i++) { p_curr[i] = t[i].P_Curr; p_orig[i] = t[i].P_Orig; alias[i] = t[i].Alias; }
}
void zref_oct32_encode(const float* n3, uint16_t* out, uint64_t n)
{ for (uint64_t i = 0; i < n; i++) { Math::oct32 o(n3[3 * i], n3[3 * i + 1], n3[3 * i + 2]); out[2 * i] = o.v.x; out[2 * i + 1] = o.v.y; } }
void zref_oct32_decode(const uint16_t* in, float* n3, uint64_t n)
{ for (uint64_t i = 0; i < n; i++) { Math::oct32 o; o.v.x = in[2 * i]; o.v.y = in[2 * i + 1]; Math::float3 f = o.decode(); n3[3 * i] = f.x; n3[3 * i + 1] = f.y; n3[3 * i + 2] = f.z; } }
void zref_f32_to_f16(const float* x, uint16_t* y, uint64_t n) { for (uint64_t i = 0; i < n; i++) { Math::half h(x[i]); y[i] = h.x; } }
// Math::Halton (Sampling.cpp:160-174): the 64 sample points of the textured branch of EstimateTriEmissivePower (K2)
float zref_halton(int i, int b) { return Math::Halton(i, b); }
// unorm4::FromNormalized (Vector.h:745-769): the rotation quantisation of RT::MeshInstance (RtAccelerationStructure.cpp:343-357)
void zref_unorm4_from_normalized(const float* q4, uint16_t* out, uint64_t n)
{
    for (uint64_t i = 0; i < n; i++)
    {
        Math::float4a v(q4[4 * i], q4[4 * i + 1], q4[4 * i + 2], q4[4 * i + 3]);
        Math::unorm4 u = Math::unorm4::FromNormalized(v);
        out[4 * i] = u.x; out[4 * i + 1] = u.y; out[4 * i + 2] = u.z; out[4 * i + 3] = u.w;
    }
}
uintptr_t zref_align_phase(const float* p) { return ((32 - (reinterpret_cast<uintptr_t>(p) & 31)) & 31) / 4; }
}
