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

// ---- scene-ingestion pins: the reference's own transform math (Math/MatrixFuncs.h) and RT::EmissiveTriangle packing, for the C++ loader
// (zetaray_amd/host/zr_scene_io.cpp).  Matrices cross this boundary as 4 x 3 row-vector matrices (rows 0-2 = basis images, row 3 = translation).
#include <Math/MatrixFuncs.h>
#include <RayTracing/RtCommon.h>
extern "C" {
// world = affineTransformation(s, q, t) x parent  (SceneCore.cpp:871-873)
void zref_compose_world(const float* s3, const float* q4, const float* t3, const float* parent4x3, float* out4x3)
{
    using namespace ZetaRay::Math;
    float3 s(s3[0], s3[1], s3[2]); float4 q(q4[0], q4[1], q4[2], q4[3]); float3 t(t3[0], t3[1], t3[2]);
    v_float4x4 vLocal = affineTransformation(s, q, t);
    float4x3 P;
    for (int i = 0; i < 4; i++) P.m[i] = float3(parent4x3[3 * i], parent4x3[3 * i + 1], parent4x3[3 * i + 2]);
    v_float4x4 vW = mul(vLocal, load4x3(P));
    float4x3 W(store(vW));
    for (int i = 0; i < 4; i++) { out4x3[3 * i] = W.m[i].x; out4x3[3 * i + 1] = W.m[i].y; out4x3[3 * i + 2] = W.m[i].z; }
}
// TLAS::FillMeshInstanceData (RtAccelerationStructure.cpp:318-357): decomposeSRT -> unorm4 rotation, half3 scale, translation
void zref_fill_mesh_instance(const float* M4x3, float* s3, float* q4, float* t3, uint16_t* rot4, uint16_t* scale3)
{
    using namespace ZetaRay::Math;
    float4x3 M;
    for (int i = 0; i < 4; i++) M.m[i] = float3(M4x3[3 * i], M4x3[3 * i + 1], M4x3[3 * i + 2]);
    v_float4x4 vM = load4x3(M);
    float4a t, r, s;
    decomposeSRT(vM, s, r, t);
    s3[0] = s.x; s3[1] = s.y; s3[2] = s.z; q4[0] = r.x; q4[1] = r.y; q4[2] = r.z; q4[3] = r.w; t3[0] = t.x; t3[1] = t.y; t3[2] = t.z;
    unorm4 u = unorm4::FromNormalized(r); rot4[0] = u.x; rot4[1] = u.y; rot4[2] = u.z; rot4[3] = u.w;
    half3 h(s); scale3[0] = h.x; scale3[1] = h.y; scale3[2] = h.z;
}
// RT::EmissiveTriangle ctor (RtCommon.h:73-190); out = the 48-byte record
void zref_emissive_triangle(const float* v0, const float* v1, const float* v2, const float* uv6, uint32_t factorRGB8, uint32_t tex, uint16_t strengthBits,
    uint32_t triIdx, int doubleSided, void* out48)
{
    using namespace ZetaRay; using namespace ZetaRay::Math;
    RT::EmissiveTriangle e(float3(v0[0], v0[1], v0[2]), float3(v1[0], v1[1], v1[2]), float3(v2[0], v2[1], v2[2]),
        float2(uv6[0], uv6[1]), float2(uv6[2], uv6[3]), float2(uv6[4], uv6[5]), factorRGB8, tex, half::asfloat16(strengthBits), triIdx, doubleSided != 0);
    static_assert(sizeof(e) == 48, "EmissiveTriangle size");
    memcpy(out48, &e, 48);
}
// SceneCore::UpdateEmissivePositions / the first-frame emissive transform (SceneCore.cpp:196-236, 913-955): LoadVertices (decode the 16-bit
// octahedral edges and half lengths) -> mul(toWorld, v) -> StoreVertices (re-encode).  M4x3: row-vector matrix as above.
void zref_emissive_to_world(const void* in48, const float* M4x3, void* out48)
{
    using namespace ZetaRay; using namespace ZetaRay::Math;
    RT::EmissiveTriangle e; memcpy(&e, in48, 48);
    float4x3 M;
    for (int i = 0; i < 4; i++) M.m[i] = float3(M4x3[3 * i], M4x3[3 * i + 1], M4x3[3 * i + 2]);
    const v_float4x4 vW = load4x3(M);
    __m128 v0, v1, v2;
    e.LoadVertices(v0, v1, v2);
    v0 = mul(vW, v0); v1 = mul(vW, v1); v2 = mul(vW, v2);
    e.StoreVertices(v0, v1, v2);
    memcpy(out48, &e, 48);
}
}

// ---- layout pins: offsetof / sizeof of the C++ side of the reference's shared C++/HLSL headers, compiled in place ----
#include <RayTracing/RtCommon.h>
#include <Core/Material.h>
#include <Core/Vertex.h>
#include <../ZetaRenderPass/Common/FrameConstants.h>
#include <cstdio>
#include <string>

extern "C" int zref_layout(char* out, int cap)
{
    std::string s;
    char line[160];
#define ZREF_FIELD(T, f) do { std::snprintf(line, sizeof(line), #T "." #f " %zu %zu\n", offsetof(T, f), sizeof(((T*)0)->f)); s += line; } while (0)
#define ZREF_SIZE(T) do { std::snprintf(line, sizeof(line), #T " %zu\n", sizeof(T)); s += line; } while (0)
    using namespace ZetaRay;
    using Core::Vertex;
    ZREF_SIZE(Vertex); ZREF_FIELD(Vertex, Position); ZREF_FIELD(Vertex, TexUV); ZREF_FIELD(Vertex, Normal); ZREF_FIELD(Vertex, Tangent);
    using RT::MeshInstance;
    ZREF_SIZE(MeshInstance); ZREF_FIELD(MeshInstance, BaseVtxOffset); ZREF_FIELD(MeshInstance, BaseIdxOffset); ZREF_FIELD(MeshInstance, Rotation);
    ZREF_FIELD(MeshInstance, Scale); ZREF_FIELD(MeshInstance, MatIdx); ZREF_FIELD(MeshInstance, BaseEmissiveTriOffset); ZREF_FIELD(MeshInstance, Translation);
    ZREF_FIELD(MeshInstance, PrevRotation); ZREF_FIELD(MeshInstance, PrevScale); ZREF_FIELD(MeshInstance, dTranslation); ZREF_FIELD(MeshInstance, BaseColorTex);
    ZREF_FIELD(MeshInstance, AlphaFactor_Cutoff);
    using RT::EmissiveTriangle;
    ZREF_SIZE(EmissiveTriangle); ZREF_FIELD(EmissiveTriangle, Vtx0); ZREF_FIELD(EmissiveTriangle, V0V1); ZREF_FIELD(EmissiveTriangle, V0V2);
    ZREF_FIELD(EmissiveTriangle, EdgeLengths); ZREF_FIELD(EmissiveTriangle, ID); ZREF_FIELD(EmissiveTriangle, PackedA); ZREF_FIELD(EmissiveTriangle, PackedB);
    ZREF_FIELD(EmissiveTriangle, UV0); ZREF_FIELD(EmissiveTriangle, UV1); ZREF_FIELD(EmissiveTriangle, UV2);
    using RT::EmissiveLumenAliasTableEntry;
    ZREF_SIZE(EmissiveLumenAliasTableEntry); ZREF_FIELD(EmissiveLumenAliasTableEntry, CachedP_Orig); ZREF_FIELD(EmissiveLumenAliasTableEntry, CachedP_Alias);
    ZREF_FIELD(EmissiveLumenAliasTableEntry, P_Curr); ZREF_FIELD(EmissiveLumenAliasTableEntry, Alias);
    using RT::PresampledEmissiveTriangle;
    ZREF_SIZE(PresampledEmissiveTriangle); ZREF_FIELD(PresampledEmissiveTriangle, pos); ZREF_FIELD(PresampledEmissiveTriangle, normal); ZREF_FIELD(PresampledEmissiveTriangle, pdf);
    ZREF_FIELD(PresampledEmissiveTriangle, ID); ZREF_FIELD(PresampledEmissiveTriangle, idx); ZREF_FIELD(PresampledEmissiveTriangle, bary); ZREF_FIELD(PresampledEmissiveTriangle, le);
    ZREF_FIELD(PresampledEmissiveTriangle, twoSided);
    using RT::VoxelSample;
    ZREF_SIZE(VoxelSample); ZREF_FIELD(VoxelSample, pos); ZREF_FIELD(VoxelSample, normal); ZREF_FIELD(VoxelSample, pdf); ZREF_FIELD(VoxelSample, ID); ZREF_FIELD(VoxelSample, le);
    ZREF_FIELD(VoxelSample, twoSided);
    ZREF_SIZE(Material); ZREF_FIELD(Material, BaseColorFactor); ZREF_FIELD(Material, BaseColorTex_Subsurf_CoatWeight); ZREF_FIELD(Material, NormalTex_TrDepth);
    ZREF_FIELD(Material, MRTex_SpecRoughness_CoatRoughness); ZREF_FIELD(Material, EmissiveFactor_NormalScale); ZREF_FIELD(Material, EmissiveStrength_IOR);
    ZREF_FIELD(Material, EmissiveTex_AlphaCutoff_CoatIOR); ZREF_FIELD(Material, CoatColor_Flags);
    ZREF_SIZE(cbFrameConstants);
#define ZREF_CB(f) ZREF_FIELD(cbFrameConstants, f)
    ZREF_CB(CurrView); ZREF_CB(PrevView); ZREF_CB(CurrViewInv); ZREF_CB(PrevViewInv); ZREF_CB(CurrViewProj); ZREF_CB(PrevViewProj); ZREF_CB(CameraPos); ZREF_CB(CameraNear);
    ZREF_CB(AspectRatio); ZREF_CB(PixelSpreadAngle); ZREF_CB(TanHalfFOV); ZREF_CB(dt); ZREF_CB(FrameNum); ZREF_CB(CurrGBufferDescHeapOffset); ZREF_CB(PrevGBufferDescHeapOffset);
    ZREF_CB(BaseColorMapsDescHeapOffset); ZREF_CB(NormalMapsDescHeapOffset); ZREF_CB(MetallicRoughnessMapsDescHeapOffset); ZREF_CB(EmissiveMapsDescHeapOffset);
    ZREF_CB(EnvMapDescHeapOffset); ZREF_CB(RenderWidth); ZREF_CB(RenderHeight); ZREF_CB(DisplayWidth); ZREF_CB(DisplayHeight); ZREF_CB(CurrCameraJitter); ZREF_CB(PrevCameraJitter);
    ZREF_CB(PlanetRadius); ZREF_CB(SunCosAngularRadius); ZREF_CB(SunSinAngularRadius); ZREF_CB(pad); ZREF_CB(SunDir); ZREF_CB(SunIlluminance); ZREF_CB(RayleighSigmaSColor);
    ZREF_CB(RayleighSigmaSScale); ZREF_CB(OzoneSigmaAColor); ZREF_CB(OzoneSigmaAScale); ZREF_CB(MieSigmaS); ZREF_CB(MieSigmaA); ZREF_CB(AtmosphereAltitude); ZREF_CB(g);
    ZREF_CB(NumFramesCameraStatic); ZREF_CB(CameraStatic); ZREF_CB(Accumulate); ZREF_CB(SunMoved); ZREF_CB(CameraRayUVGradsScale); ZREF_CB(MipBias);
    ZREF_CB(OneDivNumEmissiveTriangles); ZREF_CB(NumEmissiveTriangles); ZREF_CB(FocusDepth); ZREF_CB(LensRadius); ZREF_CB(DoF); ZREF_CB(pad2);
    if ((int)s.size() + 1 > cap) return -(int)s.size() - 1;
    std::memcpy(out, s.c_str(), s.size() + 1);
    return (int)s.size();
}
