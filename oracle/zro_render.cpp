// ORACLE -- test infrastructure only.  Nothing under oracle/ is linked into, imported by, or called from the product
// library; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
//
// PARITY STATUS: the reference (alipbcs/ZetaRay) has no executable CPU implementation of this path and no golden
// vectors.  What IS pinned against the reference: the alias table (property tests of
// Tests/TestAliasTable.cpp + a build of the reference's own AliasTable/Kahan sources in oracle/_ref when they
// compile) and the octahedral round trip (Tests/TestMath.cpp:485-508).  Radiance, G-buffer and hit results are
// "parity unpinned": this restatement follows the shaders line by line but cannot be checked against them.
//
// zro_render.cpp: CPU restatement of
//   K1  Source/ZetaRenderPass/GBuffer/GBufferRT_Inline.hlsl + GBufferRT.hlsli            (primary-hit G-buffer)
//   K2  Source/ZetaRenderPass/PreLighting/EstimateTriEmissivePower.hlsl                   (emissive power)
//       Source/ZetaRenderPass/PreLighting/PreLighting.cpp:27-158, ZetaCore/Math/Sampling.cpp:13-50,
//       ZetaCore/Math/Common.cpp:72-139                                                   (alias table, Kahan)
//   K9  Source/ZetaRenderPass/IndirectLighting/PathTracer/PathTracer.hlsl,
//       ReSTIR_GI/PathTracing.hlsli, ReSTIR_GI/ReSTIR_GI_NEE.hlsli, NEE.hlsli             (1-spp path tracer)
// and the C entry points the tests bind with ctypes.
#include <cstdio>
#include <memory>
#include <cstdlib>
#include <chrono>
#include "zro_scene.h"
#include "../include/zetaray_amd.h"

using namespace zro;

//--------------------------------------------------------------------------------------
// Alias table (CPU side of the reference hot path)
//--------------------------------------------------------------------------------------

// Math::KahanSum, Common.cpp:72-139.  The reference sums with 8 AVX2 lanes after a scalar head that runs until the
// pointer is 32-byte aligned; `align_phase` = number of floats in that head (0 for the aligned allocations the
// reference uses).  The lane structure is reproduced exactly so the rounding is identical.
static float KahanSumRef(const float* data, int64_t N, int64_t align_phase)
{
    float sum = 0.0f, compensation = 0.0f;
    int64_t start = align_phase < N ? align_phase : N;
    for (int64_t i = 0; i < start; i++)
    {
        float corrected = data[i] - compensation;
        float newSum = sum + corrected;
        compensation = (newSum - sum) - corrected;
        sum = newSum;
    }
    int64_t numToSumSIMD = N - start;
    numToSumSIMD -= numToSumSIMD & 15;
    float vSum[8] = {0}, vComp[8] = {0};
    for (int64_t c = start; c < start + numToSumSIMD; c += 16)
        for (int l = 0; l < 8; l++)
        {
            float vCurr = data[c + l] + data[c + 8 + l];
            float vCorrected = vCurr - vComp[l];
            float vNewSum = vSum[l] + vCorrected;
            vComp[l] = (vNewSum - vSum[l]) - vCorrected;
            vSum[l] = vNewSum;
        }
    for (int i = 0; i < 8; i++)
    {
        float corrected = vSum[i] - compensation - vComp[i];
        float newSum = sum + corrected;
        compensation = (newSum - sum) - corrected;
        sum = newSum;
    }
    for (int64_t i = start + numToSumSIMD; i < N; i++)
    {
        float corrected = data[i] - compensation;
        float newSum = sum + corrected;
        compensation = (newSum - sum) - corrected;
        sum = newSum;
    }
    return sum;
}

// BuildAliasTable, PreLighting.cpp:27-158 (Vose, LIFO index stacks)
static void BuildAliasTableRef(std::vector<float>& probs, zr_alias_entry* table, int64_t align_phase)
{
    const int64_t N = (int64_t)probs.size();
    const float oneDivN = 1.0f / (float)N;
    // AliasTable_Normalize, Sampling.cpp:13-50
    const float sum = KahanSumRef(probs.data(), N, align_phase);
    const float sumRcp = (float)N / sum;
    for (int64_t i = 0; i < N; i++) probs[i] *= sumRcp;

    for (int64_t i = 0; i < N; i++) { table[i].cached_p_orig = probs[i] * oneDivN; table[i].alias = 0xffffffffu; table[i].p_curr = 0; }
    std::vector<uint32_t> larger, smaller;
    larger.reserve(N); smaller.reserve(N);
    for (int64_t i = 0; i < N; i++)
    {
        if (probs[i] < 1.0f) smaller.push_back((uint32_t)i);
        else larger.push_back((uint32_t)i);
    }
    while (!smaller.empty() && !larger.empty())
    {
        const uint32_t smallerIdx = smaller.back(); smaller.pop_back();
        const float smallerProb = probs[smallerIdx];
        const uint32_t largerIdx = larger.back();
        float largerProb = probs[largerIdx];
        table[smallerIdx].alias = largerIdx;
        table[smallerIdx].p_curr = smallerProb;
        largerProb = (smallerProb + largerProb) - 1.0f;
        probs[largerIdx] = largerProb;
        if (largerProb < 1.0f) { larger.pop_back(); smaller.push_back(largerIdx); }
    }
    while (!larger.empty()) { uint32_t idx = larger.back(); larger.pop_back(); table[idx].alias = idx; table[idx].p_curr = 1.0f; }
    while (!smaller.empty()) { uint32_t idx = smaller.back(); smaller.pop_back(); table[idx].alias = idx; table[idx].p_curr = 1.0f; }
    for (int64_t i = 0; i < N; i++) table[i].cached_p_alias = table[table[i].alias].cached_p_orig;
}

//--------------------------------------------------------------------------------------
// Packing helpers for G-buffer planes (D3D format conversion rules pinned by the ABI)
//--------------------------------------------------------------------------------------
static inline uint32_t PackUnorm8(float f) { return Math::FloatToUNorm8(f); }
static inline uint32_t PackUnorm16(float f) { return (uint32_t)Math::FloatToUNorm16(f); }
static inline uint32_t PackSnorm16(float f)
{
    if (zr_isnan(f)) f = 0;
    f = zr_clamp(f, -1.0f, 1.0f);
    f = f * 32767.0f;
    int32_t i = (int32_t)(f >= 0 ? f + 0.5f : f - 0.5f);
    return (uint32_t)(uint16_t)(int16_t)i;
}
// GBuffers.hlsli:52-68
static inline float EncodeMetallic(float metalness, bool isTransmissive, float3 emissive, float trDepth, float subsurface, float coat_weight)
{
    bool isMetal = metalness >= MIN_METALNESS_METAL;
    bool isEmissive = dot(emissive, emissive) > 0;
    uint32_t ret = isTransmissive ? 1u : 0u;
    ret |= ((uint32_t)isEmissive << 1);
    ret |= ((uint32_t)(trDepth > 0) << 3);
    ret |= ((uint32_t)(subsurface > 0) << 4);
    ret |= ((uint32_t)(coat_weight > 0) << 5);
    ret |= ((uint32_t)isMetal << 7);
    return (float)ret / 255.0f;
}
static inline float EncodeIOR(float ior) { return (ior - MIN_IOR) / (MAX_IOR - MIN_IOR); }
static inline float DecodeIOR(float enc) { return zr_fma(enc, MAX_IOR - MIN_IOR, MIN_IOR); }

struct GBView
{
    uint32_t w, h;
    uint32_t* baseColor; uint32_t* normal; uint16_t* mr; uint32_t* motion; uint32_t* emissive; uint8_t* ior;
    uint16_t* coat; float* depth; uint32_t* triA; uint32_t* triB;
    GBView(const zr_gbuffer_planes* p)
    {
        w = p->width; h = p->height;
        baseColor = (uint32_t*)p->plane[ZR_GB_BASE_COLOR]; normal = (uint32_t*)p->plane[ZR_GB_NORMAL];
        mr = (uint16_t*)p->plane[ZR_GB_METALLIC_ROUGHNESS]; motion = (uint32_t*)p->plane[ZR_GB_MOTION_VECTOR];
        emissive = (uint32_t*)p->plane[ZR_GB_EMISSIVE_COLOR]; ior = (uint8_t*)p->plane[ZR_GB_IOR];
        coat = (uint16_t*)p->plane[ZR_GB_COAT]; depth = (float*)p->plane[ZR_GB_DEPTH];
        triA = (uint32_t*)p->plane[ZR_GB_TRI_DIFF_GEO_A]; triB = (uint32_t*)p->plane[ZR_GB_TRI_DIFF_GEO_B];
    }
};

static inline float3 Row(const float* m, int r) { return f3(m[4 * r], m[4 * r + 1], m[4 * r + 2]); }
static inline float3 Mul3x4(const float* m, float3 p)
{
    // mul(float3x4, float4(p, 1)): row . (p, 1), left to right
    return f3(m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3], m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7],
              m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]);
}

// GBufferRT.hlsli:11-100
static float4 UVDifferentials(int px, int py, float3 origin, float3 dir, bool thinLens, float2 lensSample, float focusDepth,
    float t, float3 dpdu, float3 dpdv, const zr_frame_constants& g)
{
    float dpduDotdpdu = dot(dpdu, dpdu);
    float dpdvDotdpdv = dot(dpdv, dpdv);
    float dpduDotdpdv = dot(dpdu, dpdv);
    float det = dpduDotdpdu * dpdvDotdpdv - dpduDotdpdv * dpduDotdpdv;
    if (zr_abs(det) < 1e-7f) return {0, 0, 0, 0};

    const float2 renderDim = {(float)g.render_width, (float)g.render_height};
    const float2 jitter = {g.curr_camera_jitter[0], g.curr_camera_jitter[1]};
    float3 dir_cs_x = RT::GeneratePinholeCameraRay_CS(px + 1, py, renderDim, g.aspect_ratio, g.tan_half_fov, jitter);
    float3 dir_cs_y = RT::GeneratePinholeCameraRay_CS(px, py - 1, renderDim, g.aspect_ratio, g.tan_half_fov, jitter);
    if (thinLens)
    {
        dir_cs_x = focusDepth * dir_cs_x - f3(lensSample.x, lensSample.y, 0);
        dir_cs_y = focusDepth * dir_cs_y - f3(lensSample.x, lensSample.y, 0);
    }
    const float3 vbx = Row(g.curr_view, 0), vby = Row(g.curr_view, 1), vbz = Row(g.curr_view, 2);
    float3 dir_x = normalize(mad3(dir_cs_x.x, vbx, mad3(dir_cs_x.y, vby, dir_cs_x.z * vbz)));
    float3 dir_y = normalize(mad3(dir_cs_y.x, vbx, mad3(dir_cs_y.y, vby, dir_cs_y.z * vbz)));

    float3 faceNormal = normalize(cross(dpdu, dpdv));
    float3 p = origin + t * dir;
    float d = -dot(faceNormal, p);
    float numerator = -dot(faceNormal, origin) - d;

    float denom_x = dot(faceNormal, dir_x);
    denom_x = (denom_x < 0 ? -1.0f : 1.0f) * zr_max(zr_abs(denom_x), 1e-8f);
    float t_x = numerator / denom_x;
    float3 p_x = origin + t_x * dir_x;

    float denom_y = dot(faceNormal, dir_y);
    denom_y = (denom_y < 0 ? -1.0f : 1.0f) * zr_max(zr_abs(denom_y), 1e-8f);
    float t_y = numerator / denom_y;
    float3 p_y = origin + t_y * dir_y;

    // x_hat = (A^T A)^-1 A^T b with A = [dpdu dpdv]; mul(float2x2, float2) = row . vector, left to right
    float3 dpdx = p_x - p;
    float2 bx = {dot(dpdu, dpdx), dot(dpdv, dpdx)};
    float2 grads_x = {(dpdvDotdpdv * bx.x + -dpduDotdpdv * bx.y) / det, (-dpduDotdpdv * bx.x + dpduDotdpdu * bx.y) / det};
    float3 dpdy = p_y - p;
    float2 by = {dot(dpdu, dpdy), dot(dpdv, dpdy)};
    float2 grads_y = {(dpdvDotdpdv * by.x + -dpduDotdpdv * by.y) / det, (-dpduDotdpdv * by.x + dpduDotdpdu * by.y) / det};
    return {grads_x.x, grads_x.y, grads_y.x, grads_y.y};
}

//--------------------------------------------------------------------------------------
// K1: G-buffer (GBufferRT_Inline.hlsl:204-287, TracePrimaryHit :72-198, GBufferRT.hlsli:102-282)
//--------------------------------------------------------------------------------------
// pick (optional): GBufferRT::PickPixel's pixel -> *pick = hitMeshIdx, UINT32_MAX on a miss (GBufferRT_Inline.hlsl:241-242)
// (g_gbRect: the pixels a call renders, x0 y0 x1 y1 -- everything by default; zro_gbuffer_render_rect renders a window of a full-size frame for the
// at-size parity tests, tests/window_parity.py)
static uint32_t g_gbRect[4] = {0u, 0u, 0xffffffffu, 0xffffffffu};
static void RenderGBuffer(const Scene& sc, const zr_frame_constants& g, GBView gb, uint32_t pickX = 0xffffu, uint32_t pickY = 0xffffu, uint32_t* pick = nullptr)
{
    BSDF::g_rho = &sc.rhoLUT;
    const float2 renderDim = {(float)g.render_width, (float)g.render_height};
    const float2 jitter = {g.curr_camera_jitter[0], g.curr_camera_jitter[1]};
    const float3 vbx = Row(g.curr_view, 0), vby = Row(g.curr_view, 1), vbz = Row(g.curr_view, 2);

    for (uint32_t y = 0; y < g.render_height; y++)
    for (uint32_t x = 0; x < g.render_width; x++)
    {
        if (x < g_gbRect[0] || y < g_gbRect[1] || x >= g_gbRect[2] || y >= g_gbRect[3]) continue;
        const size_t px = (size_t)y * g.render_width + x;
        float2 lensSample = {0, 0};
        float3 rayDirCS = RT::GeneratePinholeCameraRay_CS((int)x, (int)y, renderDim, g.aspect_ratio, g.tan_half_fov, jitter);
        float3 rayOrigin = f3(g.camera_pos);
        if (g.dof)
        {
            uint32_t hx = x, hy = y, hz = x; zr_pcg3d(&hx, &hy, &hz);
            RNG rng = RNG::Init(hz, hy, g.frame_num);
            lensSample = Sampling::UniformSampleDiskConcentric(rng.Uniform2D());
            lensSample = lensSample * g.lens_radius;
            rayOrigin += mad3(lensSample.x, vbx, lensSample.y * vby);
            float3 focalPoint = g.focus_depth * rayDirCS;
            rayDirCS = focalPoint - f3(lensSample.x, lensSample.y, 0);
        }
        float3 rayDir = mad3(rayDirCS.x, vbx, mad3(rayDirCS.y, vby, rayDirCS.z * vbz));
        rayDir = normalize(rayDir);

        sc.counters.n_closest++;
        Scene::RawHit h = sc.Trace(rayOrigin, rayDir, 0.0f, ZR_FLT_MAX, ZR_SUBGROUP_ALL, false, false, 0, /*alphaTest*/ true);
        if (pick && x == pickX && y == pickY) *pick = h.hit ? sc.tris[h.tri].mesh_idx : 0xffffffffu;

        if (!h.hit)
        {
            gb.depth[px] = ZR_FLT_MAX;
            // g_metallicRoughness[DTid].x = 4 / 255 (invalid flag); y is left untouched by the reference -> pinned to 0
            gb.mr[px] = (uint16_t)PackUnorm8(4.0f / 255.0f);
            float3 prevCameraPos = f3(g.prev_view_inv[3], g.prev_view_inv[7], g.prev_view_inv[11]);
            float3 motion = f3(g.camera_pos) - prevCameraPos;
            float2 motionNDC = motion.z > 0 ? f2(motion.x / (motion.z * g.tan_half_fov), motion.y / (motion.z * g.tan_half_fov)) : f2(0, 0);
            motionNDC.x /= g.aspect_ratio;
            float2 motionUV = Math::UVFromNDC(motionNDC);
            gb.motion[px] = PackSnorm16(motionUV.x) | (PackSnorm16(motionUV.y) << 16);
            // planes the reference does not write on a miss are pinned to 0
            gb.baseColor[px] = 0; gb.normal[px] = 0; gb.emissive[px] = 0; gb.ior[px] = 0;
            for (int k = 0; k < 4; k++) { gb.coat[4 * px + k] = 0; gb.triA[4 * px + k] = 0; }
            gb.triB[2 * px] = gb.triB[2 * px + 1] = 0;
            continue;
        }

        const WorldTri& T = sc.tris[h.tri];
        const uint32_t meshIdx = T.mesh_idx, primIdx = T.prim_idx;
        const float2 bary = {h.u, h.v};
        const zr_mesh_instance& md = sc.instances[meshIdx];
        uint32_t tri = primIdx * 3 + md.base_idx_offset;
        const zr_vertex& V0 = sc.vertices[sc.indices[tri] + md.base_vtx_offset];
        const zr_vertex& V1 = sc.vertices[sc.indices[tri + 1] + md.base_vtx_offset];
        const zr_vertex& V2 = sc.vertices[sc.indices[tri + 2] + md.base_vtx_offset];

        float4 q = normalize(Math::DecodeNormalized4(md.rotation));
        const float3 scale = f3(zr_f16_to_f32(md.scale[0]), zr_f16_to_f32(md.scale[1]), zr_f16_to_f32(md.scale[2]));
        const float3 translation = f3(md.translation);

        float2 uv0 = {V0.uv[0], V0.uv[1]}, uv1 = {V1.uv[0], V1.uv[1]}, uv2 = {V2.uv[0], V2.uv[1]};
        float2 uv = uv0 + bary.x * (uv1 - uv0) + bary.y * (uv2 - uv0);

        float3 v0_n = Math::DecodeOct32(V0.normal), v1_n = Math::DecodeOct32(V1.normal), v2_n = Math::DecodeOct32(V2.normal);
        float3 normal = v0_n + bary.x * (v1_n - v0_n) + bary.y * (v2_n - v0_n);
        const float3 scaleInv = f3(1.0f / scale.x, 1.0f / scale.y, 1.0f / scale.z);
        normal *= scaleInv;
        normal = Math::RotateVector(normal, q);
        normal = normalize(normal);

        // tangent vector (GBufferRT_Inline.hlsl:147-155)
        float3 v0_t = Math::DecodeOct32(V0.tangent), v1_t = Math::DecodeOct32(V1.tangent), v2_t = Math::DecodeOct32(V2.tangent);
        float3 tangent = v0_t + bary.x * (v1_t - v0_t) + bary.y * (v2_t - v0_t);
        tangent *= scale;
        tangent = Math::RotateVector(tangent, q);
        tangent = normalize(tangent);

        float3 v0W = Math::TransformTRS(f3(V0.pos), translation, q, scale);
        float3 v1W = Math::TransformTRS(f3(V1.pos), translation, q, scale);
        float3 v2W = Math::TransformTRS(f3(V2.pos), translation, q, scale);
        float3 n0W = normalize(Math::RotateVector(v0_n * scaleInv, q));
        float3 n1W = normalize(Math::RotateVector(v1_n * scaleInv, q));
        float3 n2W = normalize(Math::RotateVector(v2_n * scaleInv, q));
        Math::TriDifferentials td = Math::TriDifferentials::Compute(v0W, v1W, v2W, n0W, n1W, n2W, uv0, uv1, uv2);

        // motion vector
        float3 hitPos = mad3(rayDir, f3(h.t), rayOrigin);   // mad(dir, t, origin)
        float3 posL = Math::InverseTransformTRS(hitPos, translation, q, scale);
        float3 prevTranslation = translation - f3(zr_f16_to_f32(md.d_translation[0]), zr_f16_to_f32(md.d_translation[1]), zr_f16_to_f32(md.d_translation[2]));
        float4 q_prev = normalize(Math::DecodeNormalized4(md.prev_rotation));
        float3 prevScale = f3(zr_f16_to_f32(md.prev_scale[0]), zr_f16_to_f32(md.prev_scale[1]), zr_f16_to_f32(md.prev_scale[2]));
        float3 pos_prev = Math::TransformTRS(posL, prevTranslation, q_prev, prevScale);
        float3 posV_prev = Mul3x4(g.prev_view, pos_prev);
        float2 posNDC_prev = {posV_prev.x / (posV_prev.z * g.tan_half_fov), posV_prev.y / (posV_prev.z * g.tan_half_fov)};
        posNDC_prev.x /= g.aspect_ratio;

        float2 currUV = {((float)x + 0.5f) / renderDim.x, ((float)y + 0.5f) / renderDim.y};
        float2 prevUV = Math::UVFromNDC(posNDC_prev) - f2(jitter.x / renderDim.x, jitter.y / renderDim.y);
        float2 motionVec = currUV - prevUV;

        float3 pos = mad3(h.t, rayDir, rayOrigin);
        float3 posV = Mul3x4(g.curr_view, pos);
        float z = g.dof ? h.t : posV.z;
        float3 wo = rayOrigin - pos;

        // ApplyTextureMaps (GBufferRT.hlsli:178-282)
        Mat mat; mat.m = sc.materials[md.mat_idx];
        const bool anyTex = mat.GetBaseColorTex() != ZR_INVALID_TEX || mat.GetNormalTex() != ZR_INVALID_TEX ||
            mat.GetMetallicRoughnessTex() != ZR_INVALID_TEX;
        // (the gradients only feed SampleGrad; they are skipped for untextured materials)
        float4 grads = {0, 0, 0, 0};
        if (anyTex)
            grads = UVDifferentials((int)x, (int)y, rayOrigin, rayDir, g.dof != 0, lensSample, g.focus_depth, h.t, td.dpdu, td.dpdv, g);
        grads = {grads.x * g.camera_ray_uv_grads_scale, grads.y * g.camera_ray_uv_grads_scale,
                 grads.z * g.camera_ray_uv_grads_scale, grads.w * g.camera_ray_uv_grads_scale};
        float3 baseColor = mat.GetBaseColorFactor();
        float3 emissiveColor = mat.GetEmissiveFactor();
        float normalScale = mat.GetNormalScale();
        float metallic = mat.Metallic() ? 1.0f : 0.0f;
        float roughness = mat.GetSpecularRoughness();
        float3 shadingNormal = normal;
        float3 dndu = td.dndu, dndv = td.dndv;
        if (mat.GetBaseColorTex() != ZR_INVALID_TEX)
        {
            float c[4];
            zr_tex_sample_grad_aniso(&sc.tex, g.base_color_maps_desc_heap_offset + mat.GetBaseColorTex(), uv.x, uv.y, grads.x, grads.y, grads.z, grads.w, 16, c);      // g_samAnisotropicWrap: MaxAnisotropy 16 (GBufferRT.hlsli:204-248, RendererCore.cpp:508-521)
            baseColor = baseColor * f3(c[0], c[1], c[2]);
        }
        // avoid normal mapping if tangent = (0, 0, 0), which results in NaN
        if (mat.GetNormalTex() != ZR_INVALID_TEX && zr_abs(dot(tangent, tangent)) > 1e-6f)
        {
            float c[4];
            zr_tex_sample_grad_aniso(&sc.tex, g.normal_maps_desc_heap_offset + mat.GetNormalTex(), uv.x, uv.y, grads.x, grads.y, grads.z, grads.w, 16, c);      // g_samAnisotropicWrap: MaxAnisotropy 16 (GBufferRT.hlsli:204-248, RendererCore.cpp:508-521)
            shadingNormal = Math::TangentSpaceToWorldSpace(f2(c[0], c[1]), tangent, normal, normalScale);
        }
        if (mat.DoubleSided() && dot(wo, normal) < 0) { shadingNormal = shadingNormal * -1.0f; dndu = dndu * -1.0f; dndv = dndv * -1.0f; }
        if (dot(wo, normal) > 0 && dot(wo, shadingNormal) < 0)
        {
            wo = normalize(wo);
            shadingNormal = shadingNormal - dot(shadingNormal, wo) * wo;
            shadingNormal = 1e-4f * wo + shadingNormal;
            shadingNormal = normalize(shadingNormal);
        }
        if (mat.GetMetallicRoughnessTex() != ZR_INVALID_TEX)
        {
            float c[4];
            zr_tex_sample_grad_aniso(&sc.tex, g.metallic_roughness_maps_desc_heap_offset + mat.GetMetallicRoughnessTex(), uv.x, uv.y, grads.x, grads.y, grads.z, grads.w, 16, c);      // g_samAnisotropicWrap: MaxAnisotropy 16 (GBufferRT.hlsli:204-248, RendererCore.cpp:508-521)
            metallic *= c[0];
            roughness *= c[1];
        }
        float emissiveStrength = mat.GetEmissiveStrength();
        if (mat.GetEmissiveTex() != ZR_INVALID_TEX)
        {
            float c[4];
            zr_tex_sample_level(&sc.tex, g.emissive_maps_desc_heap_offset + mat.GetEmissiveTex(), uv.x, uv.y, 0.0f, c);
            emissiveColor = emissiveColor * f3(c[0], c[1], c[2]);
        }
        emissiveColor *= emissiveStrength;
        bool transmissive = mat.Transmissive();
        float ior = mat.GetSpecularIOR();
        float trDepth = transmissive ? mat.GetTransmissionDepth() : 0;
        float subsurface = mat.ThinWalled() ? mat.GetSubsurface() : 0;
        float coat_weight = mat.GetCoatWeight();
        float3 coat_color = mat.GetCoatColor();
        float coat_roughness = mat.GetCoatRoughness();
        float coat_ior = mat.GetCoatIOR();
        float encoded = EncodeMetallic(metallic, transmissive, emissiveColor, trDepth, subsurface, coat_weight);

        // WriteToGBuffers (GBufferRT.hlsli:102-176)
        gb.depth[px] = z;
        float2 en = Math::EncodeUnitVector(shadingNormal);
        gb.normal[px] = PackUnorm16(en.x) | (PackUnorm16(en.y) << 16);
        gb.baseColor[px] = PackUnorm8(baseColor.x) | (PackUnorm8(baseColor.y) << 8) | (PackUnorm8(baseColor.z) << 16) |
            ((subsurface > 0 ? PackUnorm8(subsurface) : 0u) << 24);
        gb.mr[px] = (uint16_t)(PackUnorm8(encoded) | (PackUnorm8(roughness) << 8));
        gb.emissive[px] = dot(emissiveColor, emissiveColor) > 0 ? PackR11G11B10F(max3(emissiveColor, 0.0f)) : 0u;
        gb.ior[px] = transmissive ? (uint8_t)PackUnorm8(EncodeIOR(ior)) : (uint8_t)0;
        if (coat_weight > 0)
        {
            uint32_t c = Math::Float3ToRGB8(coat_color);
            gb.coat[4 * px + 0] = (uint16_t)(c & 0xffff);
            gb.coat[4 * px + 1] = (uint16_t)((c >> 16) | (Math::FloatToUNorm8(coat_weight) << 8));
            gb.coat[4 * px + 2] = (uint16_t)(Math::FloatToUNorm8(coat_roughness) | (Math::FloatToUNorm8(EncodeIOR(coat_ior)) << 8));
            gb.coat[4 * px + 3] = 0;
        }
        else { for (int k = 0; k < 4; k++) gb.coat[4 * px + k] = 0; }
        gb.motion[px] = PackSnorm16(motionVec.x) | (PackSnorm16(motionVec.y) << 16);
        uint32_t dpdu_h[3] = {zr_f32_to_f16(td.dpdu.x), zr_f32_to_f16(td.dpdu.y), zr_f32_to_f16(td.dpdu.z)};
        uint32_t dpdv_h[3] = {zr_f32_to_f16(td.dpdv.x), zr_f32_to_f16(td.dpdv.y), zr_f32_to_f16(td.dpdv.z)};
        uint32_t dndu_h[3] = {zr_f32_to_f16(dndu.x), zr_f32_to_f16(dndu.y), zr_f32_to_f16(dndu.z)};
        uint32_t dndv_h[3] = {zr_f32_to_f16(dndv.x), zr_f32_to_f16(dndv.y), zr_f32_to_f16(dndv.z)};
        gb.triA[4 * px + 0] = dpdu_h[0] | (dpdu_h[1] << 16);
        gb.triA[4 * px + 1] = dpdu_h[2] | (dpdv_h[0] << 16);
        gb.triA[4 * px + 2] = dpdv_h[1] | (dpdv_h[2] << 16);
        gb.triA[4 * px + 3] = dndu_h[0] | (dndu_h[1] << 16);
        gb.triB[2 * px + 0] = dndu_h[2] | (dndv_h[0] << 16);
        gb.triB[2 * px + 1] = dndv_h[1] | (dndv_h[2] << 16);
    }
}

//--------------------------------------------------------------------------------------
// K2: per-triangle emissive power (EstimateTriEmissivePower.hlsl:29-79).  Textured triangles: 32 lanes x 2 Halton(2, 3)
// points (PreLighting.cpp:236-243, Sampling.cpp:160-174), emissive map sampled with g_samLinearWrap at mip 0; the
// WaveActiveSum over the 32 lane partials is pinned to ascending lane order.
//--------------------------------------------------------------------------------------
static float Halton(int i, int b)
{
    float f = 1.0f, r = 0.0f, bf = (float)b;
    while (i > 0) { f /= bf; r = r + f * (float)(i % b); i = (int)((float)i / bf); }
    return r;
}
static void EstimatePower(const Scene& sc, float* out)
{
    for (size_t i = 0; i < sc.emissives.size(); i++)
    {
        EmTri tri; tri.t = sc.emissives[i];
        float3 power = f3(64.0f);    // ESTIMATE_TRI_POWER_NUM_SAMPLES_PER_TRI
        if (tri.GetTex() != ZR_INVALID_TEX)
        {
            power = f3(0.0f);
            for (int lane = 0; lane < 32; lane++)
            {
                float3 lanePower = f3(0.0f);
                for (int k = 0; k < 2; k++)
                {
                    const int si = lane * 2 + k;
                    float2 u = {Halton(si + 1, 2), Halton(si + 1, 3)};
                    float2 bary = Sampling::UniformSampleTriangle(u);
                    float2 texUV = (1.0f - bary.x - bary.y) * tri.UV0() + bary.x * tri.UV1() + bary.y * tri.UV2();
                    float c[4];
                    zr_tex_sample_level(&sc.tex, sc.emissiveMapsOffset + tri.GetTex(), texUV.x, texUV.y, 0.0f, c);
                    lanePower += f3(c[0], c[1], c[2]);
                }
                power += lanePower;
            }
        }
        power = power * tri.GetFactor() * tri.GetStrength();
        const float3 vtx0 = tri.Vtx0(), vtx1 = tri.V1(), vtx2 = tri.V2();
        const float surfaceArea = 0.5f * length(cross(vtx1 - vtx0, vtx2 - vtx0));
        const float pdf = surfaceArea > 0 ? 1.0f / surfaceArea : 0;
        out[i] = pdf > 0 ? Math::Luminance(power) * ZR_PI / (pdf * 64.0f) : 0;
    }
}

//--------------------------------------------------------------------------------------
// K9: path tracer.  NEE: RGI_Util::NEE_Emissive_MIS (ReSTIR_GI_NEE.hlsli:8-118) with PathTracer/Params.hlsli
// (MIS_ALL_BOUNCES 1, MIS_NUM_LIGHT_SAMPLES 1, MIS_NON_DIFFUSE_BSDF_SAMPLING 0, APPROXIMATE_EMISSIVE_SHADOW_RAY 0,
//  ACCOUNT_FOR_TRANSMITTANCE 1)
//--------------------------------------------------------------------------------------
static float3 NEE_Emissive_MIS(const Scene& sc, int NumLightSamples, bool skipDiffuse, float3 pos, float3 normal,
    BSDF::ShadingData surface, uint32_t numEmissives, RNG& rng, bool presampled = false, uint32_t sampleSetIdx = 0, bool approximateShadow = false)
{
    float3 ld = f3(0.0f);
    const bool specular = surface.GlossSpecular() && (surface.metallic || surface.specTr) && (!surface.Coated() || surface.CoatSpecular());
    const int numLightSamples = specular ? 0 : NumLightSamples;

    // BSDF sampling
    {
        BSDF::BSDFSample bsdfSample = skipDiffuse ? BSDF::SampleBSDF_NoDiffuse(normal, surface, rng) : BSDF::SampleBSDF(normal, surface, rng);
        float3 wi = bsdfSample.wi;
        float3 f = bsdfSample.f;
        float wiPdf = bsdfSample.pdf;
        RtRayQuery::Hit_Emissive hitInfo = RtRayQuery::Hit_Emissive::FindClosest(sc, pos, normal, wi, surface.Transmissive());
        if (hitInfo.HitWasEmissive())
        {
            EmTri emissive; emissive.t = sc.emissives[hitInfo.emissiveTriIdx];
            float3 le = Light::Le_EmissiveTriangle(sc, emissive, hitInfo.bary);
            const float3 vtx0 = emissive.Vtx0(), vtx1 = emissive.V1(), vtx2 = emissive.V2();
            float3 lightNormal = cross(vtx1 - vtx0, vtx2 - vtx0);
            float twoArea = length(lightNormal);
            twoArea = zr_max(twoArea, 1e-6f);
            lightNormal = dot(lightNormal, lightNormal) == 0 ? f3(1.0f) : lightNormal / twoArea;
            lightNormal = emissive.IsDoubleSided() && dot(-wi, lightNormal) < 0 ? -lightNormal : lightNormal;
            const float lightSourcePdf = numLightSamples > 0 ? sc.alias[hitInfo.emissiveTriIdx].cached_p_orig : 0;
            const float lightPdf = lightSourcePdf * (2.0f / twoArea);
            float dwdA = hitInfo.t > 0 ? zr_saturate(dot(lightNormal, -wi)) / (hitInfo.t * hitInfo.t) : 0;
            wiPdf *= dwdA;
            le *= f * dwdA;
            ld = RT::PowerHeuristic(wiPdf, lightPdf, le, 1, (float)numLightSamples);
        }
    }
    // Light sampling
    for (int s_l = 0; s_l < numLightSamples; s_l++)
    {
        Light::EmissiveTriSample lightSample; float3 le; float lightPdf; uint32_t lightID;
        if (presampled)     // USE_PRESAMPLED_SETS (ReSTIR_GI_NEE.hlsli:68-85)
        {
            Light::PresampledLight pl = Light::SamplePresampledSet(sc, sampleSetIdx, pos, rng);
            lightSample.pos = pl.pos; lightSample.normal = pl.normal; le = pl.le; lightPdf = pl.pdf; lightID = pl.ID;
        }
        else
        {
            Light::AliasTableSample entry = Light::AliasTableSample::get(sc, numEmissives, rng);
            EmTri tri; tri.t = sc.emissives[entry.idx];
            lightSample = Light::EmissiveTriSample::get(pos, tri, rng);
            le = Light::Le_EmissiveTriangle(sc, tri, lightSample.bary);
            lightPdf = entry.pdf * lightSample.pdf;
            lightID = tri.t.id;
        }
        const float t = length(lightSample.pos - pos);
        const float3 wi = (lightSample.pos - pos) / t;
        if (dot(lightSample.normal, -wi) > 0)
        {
            const float dwdA = zr_saturate(dot(lightSample.normal, -wi)) / (t * t);
            surface.SetWi(wi, normal);
            le *= BSDF::Unified(surface).f * dwdA;
            if (dot(le, le) > 0)
                le *= RtRayQuery::Visibility_Segment(sc, approximateShadow, pos, wi, t, normal, lightID, surface.Transmissive()) ? 1.0f : 0.0f;
            float bsdfPdf = skipDiffuse ? BSDF::BSDFSamplerPdf_NoDiffuse(normal, surface, wi, BSDF::NoOp()) :
                BSDF::BSDFSamplerPdf(normal, surface, wi, BSDF::NoOp(), rng);
            bsdfPdf *= dwdA;
            ld += RT::PowerHeuristic(lightPdf, bsdfPdf, le, (float)numLightSamples);
        }
    }
    return ld;
}

// One path of the 8x8 group ("virtual wave" of 64 lanes, SURVEY.md section 7: the RR reduction runs over the explicit
// 64-pixel block, lanes = pixels of the group in row-major order).
struct PathState
{
    bool active = false;        // still inside ReSTIR_RT::PathTrace's while(true)
    bool atRR = false;          // reached the Russian-roulette point in this iteration
    uint32_t x, y;
    float3 li, throughput, pos, normal;
    float eta_curr; int bounce; bool inTranslucentMedium;
    BSDF::BSDFSample bsdfSample; RtRayQuery::Hit hitInfo; RT::RayDifferentials rd;
    RNG rngThread, rngGroup;
    float3 firstBsdfOverPdf;
    int maxNumBounces; uint32_t sampleSetIdx = 0;
    // per-iteration temporaries carried from phase A to phase B
    BSDF::ShadingData surface; float eta_next; float3 dpdx, dpdy;
};

static const int MIN_NUM_BOUNCES_RUSSIAN_ROULETTE = 3;

//--------------------------------------------------------------------------------------
// Sun / sky next-event estimation (scenes without emissive triangles, NEE_EMISSIVE == 0):
// ReSTIR_Util::NEE_Sun<true> / NEE_Sky<true> (NEE.hlsli:86-152) chosen by RGI_Util::NEE (ReSTIR_GI_NEE.hlsli:194-226)
// with P_SUN_VS_SKY 0.65 and SUN_DISK_SAMPLING 0 (PathTracer/Params.hlsli:8,24; ReSTIR_GI/Params.hlsli:8,28).
//--------------------------------------------------------------------------------------
struct SkyIncidentRadiance { const SkyLUT* lut; float3 operator()(float3 w) const { return Light::Le_Sky(w, *lut); } };

static float3 NEE_Sun(const Scene& sc, const zr_frame_constants& g, float3 pos, float3 normal, BSDF::ShadingData surface)
{
    float3 wi = -f3(g.sun_dir);
    surface.SetWi(wi, normal);
    float3 bsdfxCosTheta = BSDF::Unified(surface).f;
    if (dot(bsdfxCosTheta, bsdfxCosTheta) == 0) return f3(0.0f);
    if (!RtRayQuery::Visibility_Ray(sc, pos, wi, normal, surface.Transmissive())) return f3(0.0f);
    float3 le = Light::Le_Sun(pos, g);
    return bsdfxCosTheta * le;
}
static float3 NEE_Sky(const Scene& sc, float3 pos, float3 normal, const BSDF::ShadingData& surface, RNG& rng)
{
    SkyIncidentRadiance leFunc; leFunc.lut = &sc.sky;
    BSDF::BSDFSample bsdfSample = BSDF::SampleBSDF(normal, surface, leFunc, rng);
    float3 ld = bsdfSample.bsdfOverPdf;
    if (dot(ld, ld) > 0) ld = ld * (RtRayQuery::Visibility_Ray(sc, pos, bsdfSample.wi, normal, surface.Transmissive()) ? 1.0f : 0.0f);
    return ld;
}
static float3 NEE_SunSky(const Scene& sc, const zr_frame_constants& g, float3 pos, float3 normal, const BSDF::ShadingData& surface, RNG& rngThread)
{
    const float P_SUN_VS_SKY = 0.65f;
    float p_sun = rngThread.Uniform();
    if (-g.sun_dir[1] > 0)
    {
        float q = (surface.Transmissive() ? 1.0f : (dot(-f3(g.sun_dir), normal) > 0 ? 1.0f : 0.0f)) * P_SUN_VS_SKY;
        if (p_sun < q) return NEE_Sun(sc, g, pos, normal, surface) / q;
        return NEE_Sky(sc, pos, normal, surface, rngThread) / (1 - q);
    }
    return f3(0.0f);
}

static void RenderPathTracer(const Scene& sc, const zr_frame_constants& g, GBView gb, const zr_params& prm, float* finalRGBA)
{
    BSDF::g_rho = &sc.rhoLUT;
    const uint32_t W = g.render_width, H = g.render_height;
    const float2 renderDim = {(float)W, (float)H};
    const float2 jitter = {g.curr_camera_jitter[0], g.curr_camera_jitter[1]};
    const float3 vbx = Row(g.curr_view, 0), vby = Row(g.curr_view, 1), vbz = Row(g.curr_view, 2);
    const bool russianRoulette = prm.flags & ZR_IND_RUSSIAN_ROULETTE;
    const bool accumulate = g.accumulate && g.camera_static;
    const uint32_t numSampleSets = prm.presampling ? prm.num_sample_sets : 0;

    const uint32_t GX = (W + 7) / 8, GY = (H + 7) / 8;
    std::vector<PathState> lanes(64);
    for (uint32_t gy = 0; gy < GY; gy++)
    for (uint32_t gx = 0; gx < GX; gx++)
    {
        // ---- PathTracer.hlsl main :115-212 + EstimateIndirectLighting :56-109, up to the PathTrace() call ----
        for (uint32_t l = 0; l < 64; l++)
        {
            PathState& P = lanes[l];
            P = PathState();
            const uint32_t x = gx * 8 + (l & 7), y = gy * 8 + (l >> 3);
            P.x = x; P.y = y;
            if (x >= W || y >= H) continue;
            const size_t px = (size_t)y * W + x;
            float* out = finalRGBA + 4 * px;

            const uint16_t mrp = gb.mr[px];
            const float mr_x = (float)(mrp & 0xff) / 255.0f, mr_y = (float)(mrp >> 8) / 255.0f;
            const uint32_t fl = (uint32_t)zr_fma(mr_x, 255.0f, 0.5f);
            const bool f_transmissive = fl & 1, f_emissive = fl & 2, f_invalid = fl & 4, f_trDepthGt0 = fl & 8, f_metallic = fl & 128;
            if (f_invalid || f_emissive)
            {
                if (!accumulate) { out[0] = out[1] = out[2] = 0; }
                continue;
            }
            const float z_view = gb.depth[px];
            float2 lensSample = {0, 0};
            float3 origin = f3(g.camera_pos);
            if (g.dof)
            {
                uint32_t hx = x, hy = y, hz = x; zr_pcg3d(&hx, &hy, &hz);
                RNG rngDoF = RNG::Init(hz, hy, g.frame_num);
                lensSample = Sampling::UniformSampleDiskConcentric(rngDoF.Uniform2D());
                lensSample = lensSample * g.lens_radius;
            }
            const float3 pos = Math::WorldPosFromScreenSpace2(f2((float)x, (float)y), renderDim, z_view, g.tan_half_fov,
                g.aspect_ratio, jitter, vbx, vby, vbz, g.dof, lensSample, g.focus_depth, origin);
            const uint32_t np = gb.normal[px];
            const float3 normal = Math::DecodeUnitVector(f2((float)(np & 0xffff) / 65535.0f, (float)(np >> 16) / 65535.0f));
            const float3 baseColor = Math::UnpackRGB8(gb.baseColor[px]);
            float eta_curr = ETA_AIR, eta_next = DEFAULT_ETA_MAT;
            if (f_transmissive) eta_next = DecodeIOR((float)gb.ior[px] / 255.0f);
            const float3 wo = normalize(origin - pos);
            BSDF::ShadingData surface = BSDF::ShadingData::Init(normal, wo, f_metallic, mr_y, baseColor, eta_curr, eta_next,
                f_transmissive, f_trDepthGt0 ? 1.0f : 0.0f);

            P.rngGroup = RNG::Init(gx ^ 61u, gy ^ 61u, g.frame_num);
            P.rngThread = RNG::Init(x ^ 511u, y ^ 31u, g.frame_num);
            P.maxNumBounces = f_transmissive ? (int)prm.max_glossy_tr_bounces : (int)prm.max_non_tr_bounces;

            // EstimateIndirectLighting
            const uint32_t sampleSetIdx = P.rngGroup.UniformUintBounded_Faster(numSampleSets);
            P.sampleSetIdx = sampleSetIdx;
            P.li = f3(0.0f);
            BSDF::BSDFSample bsdfSample = BSDF::SampleBSDF(normal, surface, P.rngThread);
            bool alive = bsdfSample.pdf != 0;
            RtRayQuery::Hit hitInfo; hitInfo.hit = false;
            if (alive)
            {
                hitInfo = RtRayQuery::FindClosest(sc, true, true, pos, normal, bsdfSample.wi, surface.Transmissive());
                alive = hitInfo.hit;
            }
            if (alive)
            {
                Math::TriDifferentials triDiffs = Math::TriDifferentials::Unpack(&gb.triA[4 * px], &gb.triB[2 * px]);
                RT::RayDifferentials rd = RT::RayDifferentials::Init((int)x, (int)y, renderDim, g.tan_half_fov, g.aspect_ratio,
                    jitter, vbx, vby, vbz, g.dof, g.focus_depth, lensSample, origin);
                float3 dpdx, dpdy;
                rd.dpdx_dpdy(pos, normal, dpdx, dpdy);
                rd.ComputeUVDifferentials(dpdx, dpdy, triDiffs.dpdu, triDiffs.dpdv);
                rd.UpdateRays(pos, normal, bsdfSample.wi, surface.wo, triDiffs, dpdx, dpdy, dot(bsdfSample.wi, normal) < 0, surface.eta);

                // ReSTIR_RT::PathTrace prologue (PathTracing.hlsli:16-21)
                P.active = true;
                P.throughput = f3(1.0f);
                P.pos = pos; P.normal = normal;
                P.eta_curr = dot(normal, bsdfSample.wi) < 0 ? eta_next : ETA_AIR;
                P.bounce = 0;
                P.inTranslucentMedium = dot(normal, bsdfSample.wi) < 0;
                P.bsdfSample = bsdfSample; P.hitInfo = hitInfo; P.rd = rd;
                P.firstBsdfOverPdf = bsdfSample.bsdfOverPdf;
            }
        }

        // ---- ReSTIR_RT::PathTrace loop (PathTracing.hlsli:23-96), lanes in lockstep per iteration ----
        for (;;)
        {
            bool any = false;
            // phase A: up to the Russian-roulette point
            for (uint32_t l = 0; l < 64; l++)
            {
                PathState& P = lanes[l];
                P.atRR = false;
                if (!P.active) continue;
                any = true;
                float3 hitPos = mad3(P.hitInfo.t, P.bsdfSample.wi, P.pos);
                P.rd.dpdx_dpdy(hitPos, P.hitInfo.normal, P.dpdx, P.dpdy);
                P.rd.ComputeUVDifferentials(P.dpdx, P.dpdy, P.hitInfo.triDiffs.dpdu, P.hitInfo.triDiffs.dpdv);
                if (!RtRayQuery::GetMaterialData(sc, -P.bsdfSample.wi, P.eta_curr, P.rd.uv_grads, P.hitInfo, P.surface, P.eta_next))
                { P.active = false; continue; }
                // RGI_Util::NEE, NEE_EMISSIVE == 1, USE_MIS == 1, MIS_ALL_BOUNCES == 1
                if (g.num_emissive_triangles)
                    P.li += P.throughput * NEE_Emissive_MIS(sc, 1, false, hitPos, P.hitInfo.normal, P.surface, g.num_emissive_triangles, P.rngThread,
                        prm.presampling != 0, P.sampleSetIdx);
                else    // NEE_EMISSIVE == 0
                    P.li += P.throughput * NEE_SunSky(sc, g, hitPos, P.hitInfo.normal, P.surface, P.rngThread);
                if (P.inTranslucentMedium && (P.surface.trDepth > 0))
                {
                    float3 extCoeff = -log3(P.surface.baseColor_Fr0_TrCol) / P.surface.trDepth;
                    P.throughput *= exp3(-P.hitInfo.t * extCoeff);
                }
                if (P.bounce >= (P.maxNumBounces - 1)) { P.active = false; continue; }
                P.pos = hitPos;
                P.normal = P.hitInfo.normal;
                P.bounce++;
                P.atRR = russianRoulette && (P.bounce >= MIN_NUM_BOUNCES_RUSSIAN_ROULETTE);
            }
            if (!any) break;
            // WaveActiveMax over the lanes that execute the RR block in this iteration
            float waveThroughput = 0; bool anyRR = false;
            for (uint32_t l = 0; l < 64; l++)
                if (lanes[l].active && lanes[l].atRR)
                {
                    float lum = Math::Luminance(lanes[l].throughput);
                    waveThroughput = anyRR ? zr_max(waveThroughput, lum) : lum; anyRR = true;
                }
            // phase B
            for (uint32_t l = 0; l < 64; l++)
            {
                PathState& P = lanes[l];
                if (!P.active) continue;
                if (P.atRR)
                {
                    float p_terminate = zr_max(0.05f, 1 - waveThroughput);
                    if (P.rngGroup.Uniform() < p_terminate) { P.active = false; continue; }
                    P.throughput /= (1 - p_terminate);
                }
                P.bsdfSample = BSDF::BSDFSample::Init();
                if (P.bounce < P.maxNumBounces) P.bsdfSample = BSDF::SampleBSDF(P.normal, P.surface, P.rngThread);
                if (Math::Luminance(P.bsdfSample.bsdfOverPdf) == 0) { P.active = false; continue; }
                P.hitInfo = RtRayQuery::FindClosest(sc, false, true, P.pos, P.normal, P.bsdfSample.wi, P.surface.Transmissive());
                if (!P.hitInfo.hit) { P.active = false; continue; }
                P.throughput *= P.bsdfSample.bsdfOverPdf;
                bool transmitted = dot(P.normal, P.bsdfSample.wi) < 0;
                P.eta_curr = transmitted ? (P.eta_curr == ETA_AIR ? P.eta_next : ETA_AIR) : P.eta_curr;
                P.inTranslucentMedium = transmitted ? !P.inTranslucentMedium : P.inTranslucentMedium;
                P.rd.UpdateRays(P.pos, P.normal, P.bsdfSample.wi, P.surface.wo, P.hitInfo.triDiffs, P.dpdx, P.dpdy, transmitted, P.surface.eta);
            }
        }

        // ---- epilogue (PathTracer.hlsl:99-108, 199-211) ----
        for (uint32_t l = 0; l < 64; l++)
        {
            PathState& P = lanes[l];
            if (P.x >= W || P.y >= H) continue;
            const size_t px = (size_t)P.y * W + P.x;
            const uint32_t fl = (uint32_t)zr_fma((float)(gb.mr[px] & 0xff) / 255.0f, 255.0f, 0.5f);
            if (fl & (2u | 4u)) continue;
            float3 li = P.li;
            if (dot(li, li) > 0) li *= P.firstBsdfOverPdf;
            li = any_nan(li) ? f3(0.0f) : li;
            float* out = finalRGBA + 4 * px;
            if (accumulate) { out[0] += li.x; out[1] += li.y; out[2] += li.z; }
            else { out[0] = li.x; out[1] = li.y; out[2] = li.z; }
        }
    }
}

#include "zro_rpt.h"
#include "zro_rdi.h"
#include "zro_rgi.h"
#include "zro_sdi.h"
#include "zro_kat.h"
#include "zro_post.h"
#include "zro_svgf.h"

//--------------------------------------------------------------------------------------
// C entry points (ctypes)
//--------------------------------------------------------------------------------------
// ---------------------------------------------------------------- TAA (TAA.hlsl:28-188, Common.hlsli:63-105); ORACLE restatement
namespace TAA {
static float Mitchell1D(float x, float B, float C)
{
    x = zr_abs(2.0f * x);
    const float oneDivSix = 1.0f / 6.0f;
    if (x > 1)
        return ((-B - 6.0f * C) * x * x * x + (6.0f * B + 30.0f * C) * x * x + (-12.0f * B - 48.0f * C) * x + (8.0f * B + 24.0f * C)) * oneDivSix;
    return ((12.0f - 9.0f * B - 6.0f * C) * x * x * x + (-18.0f + 12.0f * B + 6.0f * C) * x * x + (6.0f - 2.0f * B)) * oneDivSix;
}
static float3 ClipAABB(float3 aabbMin, float3 aabbMax, float3 histSample)
{
    float3 center = 0.5f * (aabbMax + aabbMin);
    float3 extents = 0.5f * (aabbMax - aabbMin);
    float3 rayToCenter = histSample - center;
    float3 rayToCenterUnit = abs3(rayToCenter / extents);
    float m = zr_max(rayToCenterUnit.x, zr_max(rayToCenterUnit.y, rayToCenterUnit.z));
    if (m > 1.0f) return center + rayToCenter / m;
    return histSample;
}
struct Tex16 { const uint16_t* p; int w, h;
    float3 Load(int x, int y) const { const uint16_t* t = p + 4 * ((size_t)y * w + x); return f3(zr_f16_to_f32(t[0]), zr_f16_to_f32(t[1]), zr_f16_to_f32(t[2])); }
    // SampleLevel(g_samLinearClamp, uv, 0): software bilinear pinned by the ABI (texel centres at +0.5, clamp addressing)
    float3 Sample(float2 uv) const
    {
        float x = uv.x * (float)w - 0.5f, y = uv.y * (float)h - 0.5f;
        float fx = zr_floor(x), fy = zr_floor(y);
        float tx = x - fx, ty = y - fy;
        auto cl = [](long long v, int hi) { return (int)(v < 0 ? 0 : (v > hi ? hi : v)); };
        long long ix = (long long)zr_f2i_sat(fx), iy = (long long)zr_f2i_sat(fy);
        int x0 = cl(ix, w - 1), x1 = cl(ix == 2147483647LL ? ix : ix + 1, w - 1), y0 = cl(iy, h - 1), y1 = cl(iy == 2147483647LL ? iy : iy + 1, h - 1);
        float3 top = Load(x0, y0) + tx * (Load(x1, y0) - Load(x0, y0));
        float3 bot = Load(x0, y1) + tx * (Load(x1, y1) - Load(x0, y1));
        return top + ty * (bot - top);
    }
};
static float3 SampleTextureCatmullRom(const Tex16& tex, float2 uv, float2 texSize)
{
    float2 samplePos = {uv.x * texSize.x, uv.y * texSize.y};
    float2 texPos1 = {zr_floor(samplePos.x - 0.5f) + 0.5f, zr_floor(samplePos.y - 0.5f) + 0.5f};
    float2 f = samplePos - texPos1;
    auto W0 = [](float f) { return f * (-0.5f + f * (1.0f - 0.5f * f)); };
    auto W1 = [](float f) { return 1.0f + f * f * (-2.5f + 1.5f * f); };
    auto W2 = [](float f) { return f * (0.5f + f * (2.0f - 1.5f * f)); };
    auto W3 = [](float f) { return f * f * (-0.5f + 0.5f * f); };
    float2 w0 = {W0(f.x), W0(f.y)}, w1 = {W1(f.x), W1(f.y)}, w2 = {W2(f.x), W2(f.y)}, w3 = {W3(f.x), W3(f.y)};
    float2 w12 = w1 + w2;
    float2 offset12 = w2 / (w1 + w2);
    float2 texPos0 = (texPos1 - f2(1.0f, 1.0f)) / texSize;
    float2 texPos3 = (texPos1 + f2(2.0f, 2.0f)) / texSize;
    float2 texPos12 = (texPos1 + offset12) / texSize;
    float3 result = f3(0.0f);
    result += tex.Sample(f2(texPos0.x, texPos0.y)) * w0.x * w0.y;
    result += tex.Sample(f2(texPos12.x, texPos0.y)) * w12.x * w0.y;
    result += tex.Sample(f2(texPos3.x, texPos0.y)) * w3.x * w0.y;
    result += tex.Sample(f2(texPos0.x, texPos12.y)) * w0.x * w12.y;
    result += tex.Sample(f2(texPos12.x, texPos12.y)) * w12.x * w12.y;
    result += tex.Sample(f2(texPos3.x, texPos12.y)) * w3.x * w12.y;
    result += tex.Sample(f2(texPos0.x, texPos3.y)) * w0.x * w3.y;
    result += tex.Sample(f2(texPos12.x, texPos3.y)) * w12.x * w3.y;
    result += tex.Sample(f2(texPos3.x, texPos3.y)) * w3.x * w3.y;
    return result;
}
static float2 MotionOf(uint32_t m)      // R16G16_SNORM
{
    float fx = (float)(int16_t)(uint16_t)(m & 0xffff) / 32767.0f, fy = (float)(int16_t)(uint16_t)(m >> 16) / 32767.0f;
    return {fx < -1.0f ? -1.0f : fx, fy < -1.0f ? -1.0f : fy};
}
static void Render(const float* signal, const float* depthPlane, const uint32_t* motion, const uint16_t* prevOut, uint16_t* currOut,
    int W, int H, float blendWeight, bool temporalIsValid)
{
    auto Signal = [&](int x, int y) { const float* c = signal + 4 * ((size_t)y * W + x);
        return f3(c[0], c[1], c[2]); };      // the reference's input is Compositing's R32G32B32A32_FLOAT texture (Compositing.h:96, PostProcessor.cpp:158)
    auto Store = [&](int x, int y, float3 c) { uint16_t* o = currOut + 4 * ((size_t)y * W + x);
        o[0] = zr_f32_to_f16(c.x); o[1] = zr_f32_to_f16(c.y); o[2] = zr_f32_to_f16(c.z); };
    Tex16 prev{prevOut, W, H};
    for (int y = 0; y < H; y++) for (int x = 0; x < W; x++)
    {
        const float depth = depthPlane[(size_t)y * W + x];
        const float3 currColor = Signal(x, y);
        if (!temporalIsValid || depth == ZR_FLT_MAX) { Store(x, y, currColor); continue; }
        float weightSum = Mitchell1D(0, 0.33f, 0.33f) * Mitchell1D(0, 0.33f, 0.33f);
        float3 reconstructed = currColor * weightSum;
        float3 firstMoment = currColor;
        float3 secondMoment = currColor * currColor;
        float closestDepth = depth; int cax = 0, cay = 0;
        int numNeighbors = 1;
        for (int i = -1; i < 2; i++) for (int j = -1; j < 2; j++)
        {
            if (i == 0 && j == 0) continue;
            int nx = x + i, ny = y + j;
            if (nx < 0 || ny < 0 || nx >= W || ny >= H) continue;
            float3 neighborColor = max3(Signal(nx, ny), 0.0f);
            float weight = Mitchell1D((float)i, 0.33f, 0.33f) * Mitchell1D((float)j, 0.33f, 0.33f);
            weight *= 1.0f / (1.0f + Math::Luminance(neighborColor));
            reconstructed += neighborColor * weight;
            weightSum += weight;
            firstMoment += neighborColor;
            secondMoment += neighborColor * neighborColor;
            float neighborDepth = depthPlane[(size_t)ny * W + nx];
            if (neighborDepth < closestDepth) { closestDepth = neighborDepth; cax = i; cay = j; }
            numNeighbors += 1;
        }
        reconstructed /= zr_max(weightSum, 1e-5f);
        const float2 motionVec = MotionOf(motion[(size_t)(y + cay) * W + (x + cax)]);
        const float2 renderDim = {(float)W, (float)H};
        const float2 currUV = {((float)x + 0.5f) / renderDim.x, ((float)y + 0.5f) / renderDim.y};
        const float2 prevUV = currUV - motionVec;
        if (prevUV.x < 0.0f || prevUV.y < 0.0f || prevUV.x > 1.0f || prevUV.y > 1.0f) { Store(x, y, reconstructed); continue; }
        float3 history = SampleTextureCatmullRom(prev, prevUV, renderDim);
        const float3 mean = firstMoment / (float)numNeighbors;
        float3 sd = abs3(secondMoment - (firstMoment * firstMoment) / (float)numNeighbors);
        sd /= ((float)numNeighbors - 1.0f);
        sd = f3(zr_sqrt(sd.x), zr_sqrt(sd.y), zr_sqrt(sd.z));
        const float3 clippedHistory = ClipAABB(mean - sd, mean + sd, history);
        const float currWeight = zr_saturate(blendWeight * (1.0f / (1.0f + Math::Luminance(reconstructed))));
        const float histWeight = zr_saturate((1.0f - blendWeight) * (1.0f / (1.0f + Math::Luminance(clippedHistory))));
        float3 result = (currWeight * reconstructed + histWeight * clippedHistory) / (currWeight + histWeight);
        result = any_nan(result) ? reconstructed : result;
        Store(x, y, result);
    }
}
} // namespace TAA

extern "C" {

struct zro_scene { Scene s; std::unique_ptr<Scene> prevHolder; };
// ray counters of one pass render: queries against the CURRENT structure are tallied on the scene, queries against the PREVIOUS one (the CtT replay /
// reconnect passes of ReSTIR PT and the temporal shifts of the DI passes, once instances have moved) on the previous scene object -- a frame's total
// is their sum (round 4: the second term used to be dropped, which only showed in dynamic frames, where no test compared counters)
static void ResetCounters(const zro_scene* h) { h->s.counters = Counters(); if (h->prevHolder) h->prevHolder->counters = Counters(); }
static void ReadCounters(const zro_scene* h, zr_counters* c)
{
    if (!c) return;
    c->n_closest = h->s.counters.n_closest; c->n_shadow = h->s.counters.n_shadow;
    if (h->prevHolder) { c->n_closest += h->prevHolder->counters.n_closest; c->n_shadow += h->prevHolder->counters.n_shadow; }
}

zro_scene* zro_scene_create(const zr_scene_desc* d, int force_bvh)
{
    zro_scene* h = new zro_scene();
    h->s.Build(*d, force_bvh != 0);
    return h;
}
void zro_scene_destroy(zro_scene* h) { delete h; }
// zr_scene_update_instances: the scene as it was becomes the "previous" one the CtT / temporal-shift passes bind
int zro_scene_update_instances(zro_scene* h, const zr_mesh_instance* instances, const float* instance_to_world, uint32_t n)
{
    if (n != h->s.instances.size()) return -1;
    h->s.prev = nullptr;
    h->prevHolder.reset(new Scene(h->s));
    h->s.UpdateInstances(instances, instance_to_world, n);
    h->s.prev = h->prevHolder.get();
    return 0;
}
// zr_scene_update_emissives: there is one emissive buffer (the reference's too) -- the previous-frame scene view sees the new records as well
int zro_scene_update_emissives(zro_scene* h, const zr_emissive_triangle* tris, uint32_t first, uint32_t count)
{
    if ((size_t)first + count > h->s.emissives.size()) return -1;
    std::copy(tris, tris + count, h->s.emissives.begin() + first);
    if (h->prevHolder) std::copy(tris, tris + count, h->prevHolder->emissives.begin() + first);
    return 0;
}
// zr_scene_update_materials (one material buffer, shared with the previous-frame view)
int zro_scene_update_materials(zro_scene* h, const zr_material* m, uint32_t first, uint32_t count)
{
    if ((size_t)first + count > h->s.materials.size()) return -1;
    std::copy(m, m + count, h->s.materials.begin() + first);
    if (h->prevHolder) std::copy(m, m + count, h->prevHolder->materials.begin() + first);
    return 0;
}
int zro_scene_num_tris(const zro_scene* h) { return (int)h->s.tris.size(); }

int zro_kahan_sum(const float* data, uint64_t n, uint32_t align_phase, float* out)
{ *out = KahanSumRef(data, (int64_t)n, align_phase); return 0; }

int zro_alias_table_build(const float* power, uint32_t n, uint32_t align_phase, zr_alias_entry* out)
{
    std::vector<float> probs(power, power + n);
    BuildAliasTableRef(probs, out, align_phase);
    return 0;
}
int zro_scene_set_alias_table(zro_scene* h, const zr_alias_entry* e, uint32_t n) { h->s.alias.assign(e, e + n); return 0; }
int zro_estimate_power(const zro_scene* h, float* out) { EstimatePower(h->s, out); return 0; }
// zr_texture.h on the scene's heap.  mode 0: point (mip 0); 1: SampleLevel, lod = g[0]; 2: SampleGrad, g = ddx.uv, ddy.uv
int zro_tex_sample(const zro_scene* h, uint32_t tex, int mode, const float* uv, const float* g, uint32_t n, float* out)
{
    for (uint32_t i = 0; i < n; i++)
    {
        float* o = out + 4 * i;
        if (mode == 0) zr_tex_point(&h->s.tex, tex, uv[2 * i], uv[2 * i + 1], o);
        else if (mode == 1) zr_tex_sample_level(&h->s.tex, tex, uv[2 * i], uv[2 * i + 1], g[4 * i], o);
        else zr_tex_sample_grad(&h->s.tex, tex, uv[2 * i], uv[2 * i + 1], g[4 * i], g[4 * i + 1], g[4 * i + 2], g[4 * i + 3], o);
    }
    return 0;
}
// latches the four texture descriptor-table offsets of the frame constants (for the entry points that take no cb)
float zro_halton(int i, int b) { return Halton(i, b); }
int zro_scene_latch_heap_offsets(const zro_scene* h, const zr_frame_constants* cb) { h->s.LatchHeapOffsets(*cb); return 0; }

int zro_gbuffer_render(const zro_scene* h, const zr_frame_constants* cb, zr_gbuffer_planes* planes)
{ h->s.LatchHeapOffsets(*cb); RenderGBuffer(h->s, *cb, GBView(planes)); return 0; }
// ... only the pixels of [x0, x1) x [y0, y1) of the full-size planes (the other pixels keep what the planes held)
int zro_gbuffer_render_rect(const zro_scene* h, const zr_frame_constants* cb, zr_gbuffer_planes* planes, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1)
{
    h->s.LatchHeapOffsets(*cb);
    g_gbRect[0] = x0; g_gbRect[1] = y0; g_gbRect[2] = x1; g_gbRect[3] = y1;
    RenderGBuffer(h->s, *cb, GBView(planes));
    g_gbRect[0] = 0; g_gbRect[1] = 0; g_gbRect[2] = 0xffffffffu; g_gbRect[3] = 0xffffffffu;
    return 0;
}
// ... with GBufferRT::PickPixel(x, y) pending
int zro_gbuffer_render_pick(const zro_scene* h, const zr_frame_constants* cb, zr_gbuffer_planes* planes, uint32_t x, uint32_t y, uint32_t* mesh_idx)
{ h->s.LatchHeapOffsets(*cb); RenderGBuffer(h->s, *cb, GBView(planes), x, y, mesh_idx); return 0; }

int zro_pathtrace_render(const zro_scene* h, const zr_frame_constants* cb, const zr_gbuffer_planes* planes,
    const zr_params* prm, float* final_rgba, zr_counters* counters)
{
    ResetCounters(h); h->s.LatchHeapOffsets(*cb); h->s.texFilter = prm->tex_filter;
    RenderPathTracer(h->s, *cb, GBView(planes), *prm, final_rgba);
    ReadCounters(h, counters);
    return 0;
}

// TAA.hlsl on an RGBA32F signal + the G-buffer's depth / motion planes; prev_out / curr_out: RGBA16F (w * h * 4 halfs)
int zro_taa(const float* signal_rgba, const float* depth, const uint32_t* motion, const uint16_t* prev_out, uint16_t* curr_out,
    uint32_t w, uint32_t h, float blend_weight, int temporal_valid)
{ TAA::Render(signal_rgba, depth, motion, prev_out, curr_out, (int)w, (int)h, blend_weight, temporal_valid != 0); return 0; }

// Denoise pass (zro_svgf.h; no reference counterpart): one frame.  hist_color (RGBA32F: rgb + history length) / hist_moments (2 floats per pixel)
// hold the previous frame's history on entry and this frame's on return; out = RGBA32F (rgb + variance).  params = {alpha, alpha_moments,
// sigma_l, sigma_z}, normal_power_log2, iterations
int zro_svgf(const float* signal_rgba, const float* depth, const uint32_t* normal, const uint32_t* motion, const float* prev_depth, const uint32_t* prev_normal,
    float* hist_color, float* hist_moments, int temporal_valid, const float* params4, uint32_t normal_power_log2, uint32_t iterations, uint32_t w, uint32_t h, float* out)
{
    SVGF::Params prm = {params4[0], params4[1], params4[2], params4[3], normal_power_log2, iterations};
    SVGF::Frame(signal_rgba, depth, normal, motion, prev_depth, prev_normal, hist_color, hist_moments, temporal_valid != 0, prm, (int)w, (int)h, out);
    return 0;
}

// AutoExposure (zro_post.h): histogram of an RGBA16F (is_f16) or RGBA32F image, then the 256-thread weighted average; exposure2 = the
// persistent (exposure, adapted luminance) texel, read and written
static Post::cbAutoExposureHist AeCb(const zr_params* prm)
{ Post::cbAutoExposureHist cb; cb.MinLum = prm->ae_min_lum; cb.LumRange = prm->ae_max_lum - prm->ae_min_lum; cb.LumMapExp = prm->ae_lum_map_exp; cb.AdaptationRate = prm->ae_adaptation_rate; return cb; }
int zro_auto_exposure(const void* image, int is_f16, uint32_t w, uint32_t h, float dt, const zr_params* prm, uint32_t* hist256, float* exposure2)
{
    Post::Image img{image, is_f16 != 0, w, h};
    const Post::cbAutoExposureHist cb = AeCb(prm);
    Post::Histogram(img, cb, hist256);
    Post::WeightedAvg(hist256, w, h, dt, cb, exposure2);
    return 0;
}
// Display.hlsl mainPS over dw x dh display pixels; out_rgba = the float4 return values, out_srgb8 = the back buffer's bytes (may be null)
int zro_display(const void* image, int is_f16, uint32_t rw, uint32_t rh, uint32_t dw, uint32_t dh, const float* exposure2, const zr_params* prm,
    const uint32_t* lut, uint32_t lut_dim, float* out_rgba, uint8_t* out_srgb8)
{
    Post::Image img{image, is_f16 != 0, rw, rh};
    Post::cbDisplayPass cb; cb.Tonemapper = prm->display_tonemapper; cb.AutoExposure = prm->display_auto_exposure; cb.Saturation = prm->display_saturation; cb.AgXExp = prm->display_agx_exp;
    Post::Lut l{lut, lut_dim};
    for (uint32_t y = 0; y < dh; y++) for (uint32_t x = 0; x < dw; x++)
    {
        const float4 c = Post::mainPS(x, y, dw, dh, img, exposure2, cb, l);
        const size_t i = (size_t)y * dw + x;
        out_rgba[4 * i] = c.x; out_rgba[4 * i + 1] = c.y; out_rgba[4 * i + 2] = c.z; out_rgba[4 * i + 3] = c.w;
        if (out_srgb8) { out_srgb8[4 * i] = (uint8_t)Post::LinearToSrgb8(c.x); out_srgb8[4 * i + 1] = (uint8_t)Post::LinearToSrgb8(c.y); out_srgb8[4 * i + 2] = (uint8_t)Post::LinearToSrgb8(c.z); out_srgb8[4 * i + 3] = 255; }
    }
    return 0;
}

// Compositing.hlsl:30-125 (in-scattering off) over w x h pixels: mr = the G-buffer's METALLIC_ROUGHNESS plane (RG8_UNORM, 2 bytes / pixel);
// sky_di / emissive_di / indirect: RGBA32F planes or null (the CB_COMPOSIT_FLAGS); out_rgba is read (alpha kept) and written
int zro_composite(const zro_scene* h, const zr_frame_constants* cb, const uint8_t* mr, const float* sky_di, const float* emissive_di, const float* indirect,
    float* out_rgba, uint32_t w, uint32_t ht)
{
    const zr_frame_constants& g_frame = *cb;
    for (uint32_t y = 0; y < ht; y++) for (uint32_t x = 0; x < w; x++)
    {
        const size_t px = (size_t)y * w + x;
        const RPT::GFlags flags = RPT::DecodeFlags((uint16_t)(mr[2 * px] | (mr[2 * px + 1] << 8)));      // GBuffer::DecodeMetallic of .x
        float* o = out_rgba + 4 * px;
        const bool accumulate = g_frame.accumulate && g_frame.camera_static;
        if (flags.invalid && !accumulate)
        {
            const bool dirLighting = sky_di || emissive_di;
            const float3 c = (dirLighting && h->s.sky.data) ? Light::Le_SkyWithSunDisk(x, y, g_frame, h->s.sky) : f3(0.0f);
            o[0] = c.x; o[1] = c.y; o[2] = c.z;
            continue;
        }
        const uint32_t numFramesAccumulated = accumulate ? g_frame.num_frames_camera_static : 1;
        float3 color = f3(0.0f);
        if (sky_di) color = f3(sky_di + 4 * px);
        else if (emissive_di) color = color + f3(emissive_di + 4 * px);
        if (indirect && !flags.emissive) color = color + f3(indirect + 4 * px);
        color = color / (float)numFramesAccumulated;
        o[0] = color.x; o[1] = color.y; o[2] = color.z;
    }
    return 0;
}

// FireflyFilter.hlsl:33-123 on an RGBA32F image (Jacobi reading of the in-place filter, see include/zetaray_amd.h)
int zro_firefly(const float* in_rgba, const float* depth, float* out_rgba, uint32_t w, uint32_t h)
{
    for (uint32_t y = 0; y < h; y++) for (uint32_t x = 0; x < w; x++)
    {
        const size_t px = (size_t)y * w + x;
        float3 color = f3(in_rgba[4 * px], in_rgba[4 * px + 1], in_rgba[4 * px + 2]);
        out_rgba[4 * px + 3] = in_rgba[4 * px + 3];
        if (depth[px] != ZR_FLT_MAX)
        {
            float minLum = ZR_FLT_MAX, maxLum = 0.0f;
            float3 minColor = color, maxColor = f3(0.0f);
            const float currLum = Math::Luminance(color);
            for (int i = -1; i <= 1; i++) for (int j = -1; j <= 1; j++)
            {
                if (i == 0 && j == 0) continue;
                const int ax = (int)x + j, ay = (int)y + i;
                if (ax < 0 || ay < 0 || ax >= (int)w || ay >= (int)h) continue;
                const size_t np_ = (size_t)ay * w + ax;
                if (depth[np_] == ZR_FLT_MAX) continue;
                const float3 nc = f3(in_rgba[4 * np_], in_rgba[4 * np_ + 1], in_rgba[4 * np_ + 2]);
                const float nl = Math::Luminance(nc);
                if (nl < minLum) { minLum = nl; minColor = nc; }
                else if (nl > maxLum) { maxLum = nl; maxColor = nc; }
            }
            float3 ret = currLum < minLum ? minColor : (currLum > maxLum ? maxColor : color);
            color = minLum <= maxLum ? ret : color;
        }
        out_rgba[4 * px] = color.x; out_rgba[4 * px + 1] = color.y; out_rgba[4 * px + 2] = color.z;
    }
    return 0;
}


// K4 BuildLightVoxelGrid.hlsl: dim.x * dim.y * dim.z voxels x 64 samples, bound to the scene
int zro_build_lvg(zro_scene* h, const zr_frame_constants* cb, const uint32_t* dim, const float* extents, float offset_y, zr_voxel_sample* out)
{
    Scene& sc = h->s;
    sc.LatchHeapOffsets(*cb);
    const size_t nv = (size_t)dim[0] * dim[1] * dim[2];
    sc.lvgData.resize(nv * 64);
    for (int a = 0; a < 3; a++) { sc.lvgDim[a] = dim[a]; sc.lvgExtents[a] = extents[a]; }
    sc.lvgOffsetY = offset_y;
    const float3 e = f3(extents[0], extents[1], extents[2]);
    for (uint32_t z = 0; z < dim[2]; z++) for (uint32_t y = 0; y < dim[1]; y++) for (uint32_t x = 0; x < dim[0]; x++)
    {
        const int v[3] = {(int)x, (int)y, (int)z};
        LVG::BuildVoxel(sc, *cb, dim, e, offset_y, (int)x, (int)y, (int)z, sc.lvgData.data() + (size_t)LVG::FlattenVoxelIndex(v, dim) * 64);
    }
    if (out) std::memcpy(out, sc.lvgData.data(), sc.lvgData.size() * sizeof(zr_voxel_sample));
    return 0;
}

// K17 SkyViewLUT.hlsl: w x h R11G11B10_FLOAT texels; bound to the scene (what Le_Sky samples)
int zro_sky_lut(zro_scene* h, const zr_frame_constants* cb, uint32_t w, uint32_t ht, uint32_t* out)
{
    Scene& sc = h->s;
    sc.skyData.resize((size_t)w * ht);
    for (uint32_t y = 0; y < ht; y++) for (uint32_t x = 0; x < w; x++) sc.skyData[(size_t)y * w + x] = SkyViewLUT_Texel(*cb, x, y, w, ht);
    sc.sky.data = sc.skyData.data(); sc.sky.w = w; sc.sky.h = ht;
    if (out) std::memcpy(out, sc.skyData.data(), sc.skyData.size() * 4);
    return 0;
}
// Le_Sky / Le_Sun probes for tests: n directions (or positions) -> n RGB triples
int zro_le_sky(const zro_scene* h, const float* dirs, uint32_t n, float* out)
{
    for (uint32_t i = 0; i < n; i++) { float3 r = Light::Le_Sky(f3(dirs + 3 * i), h->s.sky); out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z; }
    return 0;
}
int zro_le_sun(const zr_frame_constants* cb, const float* pos, uint32_t n, float* out)
{
    for (uint32_t i = 0; i < n; i++) { float3 r = Light::Le_Sun(f3(pos + 3 * i), *cb); out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z; }
    return 0;
}

// K3 PresampleEmissives.hlsl:20-44: numSets * setSize samples, thread i seeds RNG::Init(i, frame)
int zro_presample(zro_scene* h, uint32_t frame_num, uint32_t num_sets, uint32_t set_size, zr_presampled_tri* out)
{
    Scene& sc = h->s;
    const uint32_t total = num_sets * set_size;
    sc.sampleSets.resize(total); sc.sampleSetSize = set_size;
    for (uint32_t i = 0; i < total; i++)
    {
        RNG rng = RNG::InitIdx(i, frame_num);
        Light::AliasTableSample entry = Light::AliasTableSample::get(sc, (uint32_t)sc.emissives.size(), rng);
        EmTri tri; tri.t = sc.emissives[entry.idx];
        Light::EmissiveTriSample ls = Light::EmissiveTriSample::get(f3(0.0f), tri, rng, false);
        float3 le = Light::Le_EmissiveTriangle(sc, tri, ls.bary);
        zr_presampled_tri& s = sc.sampleSets[i];
        s.pos[0] = ls.pos.x; s.pos[1] = ls.pos.y; s.pos[2] = ls.pos.z;
        Math::EncodeOct32(ls.normal, s.normal);
        s.le[0] = zr_f32_to_f16(le.x); s.le[1] = zr_f32_to_f16(le.y); s.le[2] = zr_f32_to_f16(le.z);
        s.bary[0] = Math::FloatToUNorm16(ls.bary.x); s.bary[1] = Math::FloatToUNorm16(ls.bary.y);
        s.two_sided = tri.IsDoubleSided() ? 1 : 0;
        s.idx = entry.idx; s.id = tri.t.id;
        s.pdf = entry.pdf * ls.pdf;
    }
    if (out) std::memcpy(out, sc.sampleSets.data(), (size_t)total * sizeof(zr_presampled_tri));
    return 0;
}

// ReSTIR PT (zro_rpt.h): stateful (two reservoir sets + r-buffers); prev may be null on the first frame
struct zro_rpt { RPT::State st; };
zro_rpt* zro_rpt_create(uint32_t w, uint32_t h, const uint16_t* sample_set_half2_512)
{
    zro_rpt* r = new zro_rpt(); r->st.Resize(w, h);
    r->st.sampleSet.assign(sample_set_half2_512, sample_set_half2_512 + 1024);
    return r;
}
void zro_rpt_destroy(zro_rpt* r) { delete r; }
void zro_rpt_reset_temporal(zro_rpt* r) { r->st.temporalValid = false; }
int zro_rpt_render(const zro_scene* h, zro_rpt* r, const zr_frame_constants* cb, const zr_gbuffer_planes* curr,
    const zr_gbuffer_planes* prev, const zr_params* prm, float* final_rgba, zr_counters* counters)
{
    ResetCounters(h); h->s.LatchHeapOffsets(*cb); h->s.texFilter = prm->tex_filter;
    RPT::Render(h->s, *cb, curr, prev, *prm, r->st, final_rgba);
    ReadCounters(h, counters);
    return 0;
}
// One stage of a frame over the pixels of rect = {x0, y0, x1, y1} of a full-size frame (zro_rpt.h RenderStage; at-size parity tests).  Several
// disjoint windows share one state: every window but the last of a stage-2 sweep passes commit = 0, which takes back the end-of-frame bookkeeping
// (the set flips, temporalValid) so that the next window runs the same sequence on its own pixels.
int zro_rpt_render_stage(const zro_scene* h, zro_rpt* r, const zr_frame_constants* cb, const zr_gbuffer_planes* curr, const zr_gbuffer_planes* prev,
    const zr_params* prm, float* final_rgba, zr_counters* counters, int stage, const uint32_t* rect, int commit)
{
    ResetCounters(h); h->s.LatchHeapOffsets(*cb); h->s.texFilter = prm->tex_filter;
    const int currIdx = r->st.currIdx; const bool valid = r->st.temporalValid;
    RPT::RenderStage(h->s, *cb, curr, prev, *prm, r->st, final_rgba, stage, rect);
    const int after = r->st.currIdx;      // (returned: where the sets stand once the stage is committed -- the caller addresses them physically meanwhile)
    if (!commit) { r->st.currIdx = currIdx; r->st.temporalValid = valid; }
    ReadCounters(h, counters);
    return after;
}
int zro_rpt_curr_idx(const zro_rpt* r) { return r->st.currIdx; }
// reads (write = 0) or writes rect {x, y, w, h} (global pixel coordinates) of a plane from / into `buf`, a row-major array of buf_w pixels per row
// whose first pixel is (bx0, by0).  which / plane: as zro_rpt_read_plane (planes 0..6 only); which = 2 / 3: reservoir set 0 / 1, whatever is current
int zro_rpt_rw_plane_rect(zro_rpt* r, int which, int plane, void* buf, uint32_t bx0, uint32_t by0, uint32_t buf_w, uint32_t x, uint32_t y, uint32_t w, uint32_t hgt, int write)
{
    RPT::ReservoirPlanes& p = r->st.reservoirs[which >= 2 ? which - 2 : (which == 0 ? 1 - r->st.currIdx : r->st.currIdx)];
    uint8_t* base; size_t bpp;
    switch (plane)
    {
    case 0: base = (uint8_t*)p.A.data(); bpp = 4; break;
    case 1: base = (uint8_t*)p.B.data(); bpp = 8; break;
    case 2: base = (uint8_t*)p.C.data(); bpp = 16; break;
    case 3: base = (uint8_t*)p.D.data(); bpp = 16; break;
    case 4: base = (uint8_t*)p.E.data(); bpp = 2; break;
    case 5: base = (uint8_t*)p.F.data(); bpp = 8; break;
    case 6: base = (uint8_t*)p.G.data(); bpp = 8; break;
    default: return 1;
    }
    const size_t W = r->st.w;
    for (uint32_t j = 0; j < hgt; j++)
    {
        uint8_t* pl = base + ((size_t)(y + j) * W + x) * bpp;
        uint8_t* bf = (uint8_t*)buf + ((size_t)(y + j - by0) * buf_w + (x - bx0)) * bpp;
        if (write) std::memcpy(pl, bf, (size_t)w * bpp); else std::memcpy(bf, pl, (size_t)w * bpp);
    }
    return 0;
}
int zro_rpt_self_shift(const zro_scene* h, zro_rpt* r, const zr_frame_constants* cb, const zr_gbuffer_planes* curr, const zr_params* prm, int which, float* out)
{ RPT::SelfShift(h->s, *cb, curr, *prm, r->st, which, out); return 0; }
// which: 0 = reservoirs the next frame will read as "previous", 1 = the other set.  plane: 0..6 = A..G, 7 = target, 8 = neighbor
int zro_rpt_read_plane(const zro_rpt* r, int which, int plane, void* out)
{
    const RPT::ReservoirPlanes& p = r->st.reservoirs[which == 0 ? 1 - r->st.currIdx : r->st.currIdx];
    auto cp = [&](const void* src, size_t bytes) { std::memcpy(out, src, bytes); return 0; };
    switch (plane)
    {
    case 0: return cp(p.A.data(), p.A.size() * 4);
    case 1: return cp(p.B.data(), p.B.size() * 4);
    case 2: return cp(p.C.data(), p.C.size() * 4);
    case 3: return cp(p.D.data(), p.D.size() * 4);
    case 4: return cp(p.E.data(), p.E.size() * 2);
    case 5: return cp(p.F.data(), p.F.size() * 4);
    case 6: return cp(p.G.data(), p.G.size() * 4);
    case 7: return cp(r->st.target.data(), r->st.target.size() * 4);
    case 8: return cp(r->st.neighbor.data(), r->st.neighbor.size());
    case 10: return cp(r->st.threadMap[0].data(), r->st.threadMap[0].size() * 2);      // K12 thread maps: CtN, NtC
    case 11: return cp(r->st.threadMap[1].data(), r->st.threadMap[1].size() * 2);
    }
    return 1;
}

// ReSTIR DI (zro_rdi.h)
struct zro_rdi { RDI::State st; };
zro_rdi* zro_rdi_create(uint32_t w, uint32_t h, const uint16_t* sample_set_half2_32)
{ zro_rdi* r = new zro_rdi(); r->st.Resize(w, h); r->st.sampleSet.assign(sample_set_half2_32, sample_set_half2_32 + 64); return r; }
void zro_rdi_destroy(zro_rdi* r) { delete r; }
void zro_rdi_reset_temporal(zro_rdi* r) { r->st.temporalValid = false; r->st.currIdx = 0; }
int zro_rdi_render(const zro_scene* h, zro_rdi* r, const zr_frame_constants* cb, const zr_gbuffer_planes* curr, const zr_gbuffer_planes* prev,
    const zr_params* prm, float* final_rgba, zr_counters* counters)
{
    ResetCounters(h); h->s.LatchHeapOffsets(*cb);
    RDI::Render(h->s, *cb, curr, prev, *prm, r->st, final_rgba);
    ReadCounters(h, counters);
    return 0;
}
// plane 0 = reservoir A (4 x u32), 1 = B (2 x f32) of the set written by the last frame, 2 = target (4 x f32)
int zro_rdi_read_plane(const zro_rdi* r, int plane, void* out)
{
    const int last = 1 - r->st.currIdx;
    if (plane == 0) std::memcpy(out, r->st.A[last].data(), r->st.A[last].size() * 4);
    else if (plane == 1) std::memcpy(out, r->st.B[last].data(), r->st.B[last].size() * 4);
    else std::memcpy(out, r->st.target.data(), r->st.target.size() * 4);
    return 0;
}

// ReSTIR DI for sun + sky (zro_sdi.h)
struct zro_sdi { SDI::State st; };
zro_sdi* zro_sdi_create(uint32_t w, uint32_t h) { zro_sdi* r = new zro_sdi(); r->st.Resize(w, h); return r; }
void zro_sdi_destroy(zro_sdi* r) { delete r; }
void zro_sdi_reset_temporal(zro_sdi* r) { r->st.temporalValid = false; r->st.currIdx = 0; }
int zro_sdi_render(const zro_scene* h, zro_sdi* r, const zr_frame_constants* cb, const zr_gbuffer_planes* curr, const zr_gbuffer_planes* prev,
    const zr_params* prm, float* final_rgba, zr_counters* counters)
{
    ResetCounters(h); h->s.LatchHeapOffsets(*cb);
    SDI::Render(h->s, *cb, curr, prev, *prm, r->st, final_rgba);
    ReadCounters(h, counters);
    return 0;
}
// plane 0 = A (u8 metadata), 1 = B (2 x u16 oct32), 2 = C (2 x f32: w_sum, W) of the set written by the last frame, 3 = target (4 x f32)
int zro_sdi_read_plane(const zro_sdi* r, int plane, void* out)
{
    const int last = 1 - r->st.currIdx;
    if (plane == 0) std::memcpy(out, r->st.A[last].data(), r->st.A[last].size());
    else if (plane == 1) std::memcpy(out, r->st.B[last].data(), r->st.B[last].size() * 2);
    else if (plane == 2) std::memcpy(out, r->st.C[last].data(), r->st.C[last].size() * 4);
    else std::memcpy(out, r->st.target.data(), r->st.target.size() * 4);
    return 0;
}

// ReSTIR GI (zro_rgi.h)
struct zro_rgi { RGI::State st; };
zro_rgi* zro_rgi_create(uint32_t w, uint32_t h) { zro_rgi* r = new zro_rgi(); r->st.Resize(w, h); return r; }
void zro_rgi_destroy(zro_rgi* r) { delete r; }
void zro_rgi_reset_temporal(zro_rgi* r) { r->st.temporalValid = false; }
int zro_rgi_render(const zro_scene* h, zro_rgi* r, const zr_frame_constants* cb, const zr_gbuffer_planes* curr, const zr_gbuffer_planes* prev,
    const zr_params* prm, float* final_rgba, zr_counters* counters)
{
    ResetCounters(h); h->s.LatchHeapOffsets(*cb); h->s.texFilter = prm->tex_filter;
    RGI::Render(h->s, *cb, curr, prev, *prm, r->st, final_rgba);
    ReadCounters(h, counters);
    return 0;
}
// plane 0 = A (4 x f32: pos, ID bits), 1 = B (4 x f16: Lo, M), 2 = C (4 x f32: w_sum, W, normal oct32 bits, unused) of the last frame's set
int zro_rgi_read_plane(const zro_rgi* r, int plane, void* out)
{
    const int last = 1 - r->st.currIdx;
    if (plane == 0) std::memcpy(out, r->st.A[last].data(), r->st.A[last].size() * 4);
    else if (plane == 1) std::memcpy(out, r->st.B[last].data(), r->st.B[last].size() * 2);
    else std::memcpy(out, r->st.C[last].data(), r->st.C[last].size() * 4);
    return 0;
}

// rays: n x 8 floats (o, tmin, d, tmax); hits: n x 4 uint32 (t bits, u bits, v bits, tri or 0xffffffff)
int zro_trace_closest(const zro_scene* h, const float* rays, uint32_t n, uint32_t mask, uint32_t* hits)
{
    for (uint32_t i = 0; i < n; i++)
    {
        const float* r = rays + 8 * i;
        Scene::RawHit rh = h->s.Trace(f3(r[0], r[1], r[2]), f3(r[4], r[5], r[6]), r[3], r[7], mask, false);
        hits[4 * i + 0] = zr_asuint(rh.hit ? rh.t : 0.0f); hits[4 * i + 1] = zr_asuint(rh.u); hits[4 * i + 2] = zr_asuint(rh.v);
        hits[4 * i + 3] = rh.hit ? rh.tri : 0xffffffffu;
    }
    return 0;
}
int zro_trace_any(const zro_scene* h, const float* rays, uint32_t n, uint32_t mask, uint32_t* occluded)
{
    for (uint32_t i = 0; i < n; i++)
    {
        const float* r = rays + 8 * i;
        occluded[i] = h->s.Trace(f3(r[0], r[1], r[2]), f3(r[4], r[5], r[6]), r[3], r[7], mask, true).hit ? 1u : 0u;
    }
    return 0;
}
// timed single-thread traversal for bench.py's cpu_baseline leg: returns seconds spent in the traversal loop only
double zro_trace_closest_timed(const zro_scene* h, const float* rays, uint32_t n, uint32_t mask, uint32_t* hits)
{
    auto t0 = std::chrono::steady_clock::now();
    zro_trace_closest(h, rays, n, mask, hits);
    auto t1 = std::chrono::steady_clock::now();
    return std::chrono::duration<double>(t1 - t0).count();
}

// ---- known-answer helpers for the math contract (tests/test_detmath.py) ----
void zro_kat_unary(int fn, const float* x, float* y, uint32_t n)
{
    for (uint32_t i = 0; i < n; i++)
        switch (fn)
        {
        case 0: y[i] = zr_sin(x[i]); break;
        case 1: y[i] = zr_cos(x[i]); break;
        case 2: y[i] = zr_exp(x[i]); break;
        case 3: y[i] = zr_log(x[i]); break;
        case 4: y[i] = zr_atan(x[i]); break;
        case 5: y[i] = zr_round_f16(x[i]); break;
        case 6: y[i] = Math::ArcCos(x[i]); break;
        case 7: y[i] = zr_sqrt(x[i]); break;
        default: y[i] = 0;
        }
}
void zro_kat_f32_to_f16(const float* x, uint16_t* y, uint32_t n) { for (uint32_t i = 0; i < n; i++) y[i] = zr_f32_to_f16(x[i]); }
void zro_kat_f16_to_f32(const uint16_t* x, float* y, uint32_t n) { for (uint32_t i = 0; i < n; i++) y[i] = zr_f16_to_f32(x[i]); }
void zro_kat_pcg3d(const uint32_t* in, uint32_t* out, uint32_t n)
{ for (uint32_t i = 0; i < n; i++) { uint32_t x = in[3 * i], y = in[3 * i + 1], z = in[3 * i + 2]; zr_pcg3d(&x, &y, &z); out[3 * i] = x; out[3 * i + 1] = y; out[3 * i + 2] = z; } }
void zro_kat_oct_encode(const float* n3, uint16_t* out, uint32_t n)
{ for (uint32_t i = 0; i < n; i++) Math::EncodeOct32(f3(n3 + 3 * i), out + 2 * i); }
void zro_kat_oct_decode(const uint16_t* in, float* n3, uint32_t n)
{ for (uint32_t i = 0; i < n; i++) { float3 v = Math::DecodeOct32(in + 2 * i); n3[3 * i] = v.x; n3[3 * i + 1] = v.y; n3[3 * i + 2] = v.z; } }
void zro_kat_rng_stream(uint32_t px, uint32_t py, uint32_t frame, float* out, uint32_t n)
{ RNG r = RNG::Init(px, py, frame); for (uint32_t i = 0; i < n; i++) out[i] = r.Uniform(); }
void zro_kat_uniform_bounded(uint32_t seed, uint32_t bound, uint32_t* out, uint32_t n)
{ RNG r = RNG::InitSeed(seed); for (uint32_t i = 0; i < n; i++) out[i] = r.UniformUintBounded(bound); }
// alias-table draw restated from LightSource.hlsli:72-98 over a standalone table (property tests of TestAliasTable.cpp)
void zro_kat_alias_sample(const zr_alias_entry* table, uint32_t n, uint32_t seed, uint32_t* idx, float* pdf, uint32_t draws)
{
    RNG r = RNG::InitSeed(seed);
    for (uint32_t i = 0; i < draws; i++)
    {
        uint32_t u0 = r.UniformUintBounded(n);
        const zr_alias_entry& s = table[u0];
        if (r.Uniform() < s.p_curr) { idx[i] = u0; pdf[i] = s.cached_p_orig; }
        else { idx[i] = s.alias; pdf[i] = s.cached_p_alias; }
    }
}
// BSDF evaluation probe: one Unified() + SampleBSDF() at a synthetic shading point (tests compare HIP vs oracle)
void zro_kat_bsdf(const zro_scene* h, const float* in /* n x 16 */, float* out /* n x 12 */, uint32_t n)
{
    BSDF::g_rho = &h->s.rhoLUT;
    for (uint32_t i = 0; i < n; i++)
    {
        const float* p = in + 16 * i;
        float3 nrm = normalize(f3(p[0], p[1], p[2]));
        float3 wo = normalize(f3(p[3], p[4], p[5]));
        float3 wi = normalize(f3(p[6], p[7], p[8]));
        bool metallic = p[9] > 0.5f; float roughness = p[10]; float3 base = f3(p[11], p[12], p[13]);
        bool specTr = p[14] > 0.5f; float coat_w = p[15];
        BSDF::ShadingData s = BSDF::ShadingData::Init(nrm, wo, metallic, roughness, base, ETA_AIR, DEFAULT_ETA_MAT, specTr, 0, 0,
            coat_w, f3(0.8f), 0.2f, DEFAULT_ETA_COAT);
        s.SetWi(wi, nrm);
        BSDF::BSDFEval e = BSDF::Unified(s);
        RNG rng = RNG::InitSeed(12345u + i);
        BSDF::BSDFSample bs = BSDF::SampleBSDF(nrm, s, rng);
        float* o = out + 12 * i;
        o[0] = e.f.x; o[1] = e.f.y; o[2] = e.f.z;
        o[3] = bs.wi.x; o[4] = bs.wi.y; o[5] = bs.wi.z; o[6] = bs.pdf;
        o[7] = bs.bsdfOverPdf.x; o[8] = bs.bsdfOverPdf.y; o[9] = bs.bsdfOverPdf.z; o[10] = (float)(int)bs.lobe;
        RNG rng2 = RNG::InitSeed(777u + i);
        o[11] = BSDF::BSDFSamplerPdf(nrm, s, wi, BSDF::NoOp(), rng2);
    }
}

// function-level probes shared with the reference build and the HIP stage functions (zro_kat.h)
void zro_kat2_sampling(const float* in, float* out, uint32_t n) { KAT::Sampling_(in, out, n); }
void zro_kat2_math(const float* in, float* out, uint32_t n) { KAT::Math_(in, out, n); }
void zro_kat2_rt(const float* in, float* out, uint32_t n) { KAT::RT_(in, out, n); }
void zro_kat2_bsdf(const uint16_t* rho, const uint32_t* rho_dim, const float* in, float* out, uint32_t n)
{ RhoLUT lut; lut.data = rho; lut.dim[0] = rho_dim[0]; lut.dim[1] = rho_dim[1]; lut.dim[2] = rho_dim[2]; KAT::BSDF_(&lut, in, out, n); }

} // extern "C"
