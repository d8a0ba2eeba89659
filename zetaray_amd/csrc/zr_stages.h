// zr_stages.h -- per-item stage functions of the G-buffer pass (K1) and the wavefront path tracer (K9).
//
// MI355X design: the reference runs one thread per pixel through the whole path
// (Source/ZetaRenderPass/IndirectLighting/PathTracer/PathTracer.hlsl:115-212 -> ReSTIR_GI/PathTracing.hlsli:10-99, a
// "megakernel" whose inner loop calls the driver's RayQuery).  Here the path is cut at every BVH query into streaming
// stages over SoA queues in HBM:
//
//     gbuffer -> pt_init -> [ trace(C, M, S) -> pt_shade ] x (maxBounces + 1)
//
// * a path's state lives in 16-byte SoA records indexed by its *queue slot* (coalesced dwordx4 loads/stores);
// * every shade step compacts surviving paths into the next queue (wave ballot + one atomic per wave);
// * next-event estimation is deferred: a vertex emits its MIS BSDF ray (M) and its light-segment ray (S) together
//   with the continuation ray (C); the *next* shade step resolves them ("pending NEE") before shading the new vertex.
//   The RNG stream per pixel is consumed in exactly the reference's order, so results do not depend on the cut.
//
// Every function here is ZR_HD: the kernels in zr_kernels.hip are thin wrappers (slot allocation, LDS staging), and
// tests/hostexec runs the same functions serially on the CPU to check them against the oracle without a GPU.
#pragma once
#include "zr_dev_scene.h"
#include "../../include/zetaray_amd.h"

namespace zr {

struct alignas(16) F4 { float x, y, z, w; };
struct alignas(16) U4 { uint32_t x, y, z, w; };
ZR_HD F4 f4(V3 a, float w) { F4 r; r.x = a.x; r.y = a.y; r.z = a.z; r.w = w; return r; }
ZR_HD F4 f4(float x, float y, float z, float w) { F4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
ZR_HD V3 xyz(F4 a) { return v3(a.x, a.y, a.z); }

// ---------------------------------------------------------------- G-buffer planes (device or host pointers)
struct GBuf
{
    // planes cover the screen tile [x0, x0 + w) x [y0, y0 + h) of the full render target (multi-GPU tile split,
    // SURVEY.md section 8(e)); pixel coordinates passed to the stage functions are always global.
    uint32_t w, h, x0, y0;
    uint32_t* baseColor; uint32_t* normal; uint16_t* mr; uint32_t* motion; uint32_t* emissive; uint8_t* ior;
    uint16_t* coat; float* depth; uint32_t* triA; uint32_t* triB;
    uint32_t plain = 0;      // the scene's material class (SceneView::plain) for the surfaces rebuilt from these planes: set by the PLAIN kernel permutations only
};

ZR_HD float EncodeMetallic(float metalness, bool tr, V3 emissive, float trDepth, float subsurface, float coat_weight)  // GBuffers.hlsli:52-68
{
    uint32_t r = tr ? 1u : 0u;
    r |= ((uint32_t)(dot(emissive, emissive) > 0) << 1);
    r |= ((uint32_t)(trDepth > 0) << 3);
    r |= ((uint32_t)(subsurface > 0) << 4);
    r |= ((uint32_t)(coat_weight > 0) << 5);
    r |= ((uint32_t)(metalness >= kMinMetalnessMetal) << 7);
    return zr_div255((float)r);
}
ZR_HD float EncodeIOR(float ior) { return (ior - kMinIOR) / (kMaxIOR - kMinIOR); }   // GBuffers.hlsli:97-105
ZR_HD float DecodeIOR(float e) { return zr_fma(e, kMaxIOR - kMinIOR, kMinIOR); }

// GBufferRT::UVDifferentials, GBufferRT.hlsli:11-100
ZR_HD F4 UVDifferentials(int px, int py, V3 origin, V3 dir, bool thinLens, V2 lensSample, float focusDepth, float t,
    V3 dpdu, V3 dpdv, const zr_frame_constants& g)
{
    const float dpduDotdpdu = dot(dpdu, dpdu);
    const float dpdvDotdpdv = dot(dpdv, dpdv);
    const float dpduDotdpdv = dot(dpdu, dpdv);
    const float det = dpduDotdpdu * dpdvDotdpdv - dpduDotdpdv * dpduDotdpdv;
    if (zr_abs(det) < 1e-7f) return f4(0, 0, 0, 0);

    const V2 renderDim = v2((float)g.render_width, (float)g.render_height);
    const V2 jitter = v2(g.curr_camera_jitter[0], g.curr_camera_jitter[1]);
    V3 dir_cs_x = GeneratePinholeCameraRay_CS(px + 1, py, renderDim, g.aspect_ratio, g.tan_half_fov, jitter);
    V3 dir_cs_y = GeneratePinholeCameraRay_CS(px, py - 1, renderDim, g.aspect_ratio, g.tan_half_fov, jitter);
    if (thinLens)
    {
        dir_cs_x = focusDepth * dir_cs_x - v3(lensSample.x, lensSample.y, 0);
        dir_cs_y = focusDepth * dir_cs_y - v3(lensSample.x, lensSample.y, 0);
    }
    const V3 vbx = Row3(g.curr_view, 0), vby = Row3(g.curr_view, 1), vbz = Row3(g.curr_view, 2);
    const V3 dir_x = normalize(mad(dir_cs_x.x, vbx, mad(dir_cs_x.y, vby, dir_cs_x.z * vbz)));
    const V3 dir_y = normalize(mad(dir_cs_y.x, vbx, mad(dir_cs_y.y, vby, dir_cs_y.z * vbz)));

    const V3 faceNormal = normalize(cross(dpdu, dpdv));
    const V3 p = origin + t * dir;
    const float d = -dot(faceNormal, p);
    const float numerator = -dot(faceNormal, origin) - d;

    float denom_x = dot(faceNormal, dir_x);
    denom_x = (denom_x < 0 ? -1.0f : 1.0f) * zr_max(zr_abs(denom_x), 1e-8f);
    const float t_x = numerator / denom_x;
    const V3 p_x = origin + t_x * dir_x;

    float denom_y = dot(faceNormal, dir_y);
    denom_y = (denom_y < 0 ? -1.0f : 1.0f) * zr_max(zr_abs(denom_y), 1e-8f);
    const float t_y = numerator / denom_y;
    const V3 p_y = origin + t_y * dir_y;

    // least-squares solution x_hat = (A^T A)^-1 A^T b, A = [dpdu dpdv]; mul(float2x2, float2) = row . vector, left to right
    const V3 dpdx = p_x - p;
    const V2 bx = v2(dot(dpdu, dpdx), dot(dpdv, dpdx));
    const V3 dpdy = p_y - p;
    const V2 by = v2(dot(dpdu, dpdy), dot(dpdv, dpdy));
    return f4((dpdvDotdpdv * bx.x + -dpduDotdpdv * bx.y) / det, (-dpduDotdpdv * bx.x + dpduDotdpdu * bx.y) / det,
              (dpdvDotdpdv * by.x + -dpduDotdpdv * by.y) / det, (-dpduDotdpdv * by.x + dpduDotdpdu * by.y) / det);
}

// Math::TangentSpaceToWorldSpace, Math.hlsli:263-284
ZR_HD V3 TangentSpaceToWorldSpace(V2 bumpNormal2, V3 tangent, V3 normal, float scale)
{
    V3 bumpNormal = v3(zr_fma(2.0f, bumpNormal2.x, -1.0f), zr_fma(2.0f, bumpNormal2.y, -1.0f), 0.0f);
    bumpNormal.z = zr_sqrt(zr_saturate(1.0f - dot(bumpNormal, bumpNormal)));
    V3 scaledBumpNormal = bumpNormal * v3(scale, scale, 1.0f);
    if (dot(scaledBumpNormal, scaledBumpNormal) < 1e-6f) return normal;
    scaledBumpNormal = normalize(scaledBumpNormal);
    normal = normalize(normal);
    tangent = normalize(tangent - dot(tangent, normal) * normal);
    const V3 bitangent = cross(normal, tangent);
    // mul(row vector, float3x3(tangent, bitangent, normal)): sum over rows, left to right
    return scaledBumpNormal.x * tangent + scaledBumpNormal.y * bitangent + scaledBumpNormal.z * normal;
}

// K1: one pixel of GBufferRT_Inline.hlsl main (:204-287) + TracePrimaryHit (:72-198) + GBufferRT.hlsli:102-282.
// Primary rays are coherent, so traversal runs inline in this kernel (no queue round trip).
// `picked` (the pixel GBufferRT::PickPixel named, else null): receives hitMeshIdx or UINT32_MAX (GBufferRT_Inline.hlsl:241-242)
ZR_HD void GBufferPixel(const SceneView& sc, const zr_frame_constants& g, const GBuf& gb, uint32_t x, uint32_t y,
    TravStack stack, uint64_t* nClosest, uint32_t* picked = nullptr)
{
    const uint32_t px = (y - gb.y0) * gb.w + (x - gb.x0);
    const V2 renderDim = v2((float)g.render_width, (float)g.render_height);
    const V2 jitter = v2(g.curr_camera_jitter[0], g.curr_camera_jitter[1]);
    const V3 vbx = Row3(g.curr_view, 0), vby = Row3(g.curr_view, 1), vbz = Row3(g.curr_view, 2);

    V2 lens = v2(0, 0);
    V3 dirCS = GeneratePinholeCameraRay_CS((int)x, (int)y, renderDim, g.aspect_ratio, g.tan_half_fov, jitter);
    V3 origin = v3p(g.camera_pos);
    if (g.dof)
    {
        uint32_t hx = x, hy = y, hz = x; zr_pcg3d(&hx, &hy, &hz);
        Rng rng = Rng::Init(hz, hy, g.frame_num);
        lens = UniformSampleDiskConcentric(rng.Uniform2D());
        lens = lens * g.lens_radius;
        origin = origin + mad(lens.x, vbx, lens.y * vby);
        dirCS = g.focus_depth * dirCS - v3(lens.x, lens.y, 0);
    }
    V3 dir = normalize(mad(dirCS.x, vbx, mad(dirCS.y, vby, dirCS.z * vbz)));

    (void)nClosest;
    // TracePrimaryHit: the only ray type without RAY_FLAG_FORCE_OPAQUE -> alpha-tested candidates
    RawHit h = TraverseDyn(sc, origin, dir, 0.0f, ZR_FLT_MAX, ZR_SUBGROUP_ALL, stack, false, false, 0, /*alphaTest*/ true);

    if (h.tri == kInvalidTri)
    {
        if (picked) *picked = 0xffffffffu;
        gb.depth[px] = ZR_FLT_MAX;
        gb.mr[px] = (uint16_t)FloatToUNorm8(4.0f / 255.0f);
        V3 prevCam = v3(g.prev_view_inv[3], g.prev_view_inv[7], g.prev_view_inv[11]);
        V3 motion = v3p(g.camera_pos) - prevCam;
        V2 mndc = motion.z > 0 ? v2(motion.x / (motion.z * g.tan_half_fov), motion.y / (motion.z * g.tan_half_fov)) : v2(0, 0);
        mndc.x /= g.aspect_ratio;
        V2 muv = UVFromNDC(mndc);
        gb.motion[px] = PackSnorm16(muv.x) | (PackSnorm16(muv.y) << 16);
        gb.baseColor[px] = 0; gb.normal[px] = 0; gb.emissive[px] = 0; gb.ior[px] = 0;
        for (int k = 0; k < 4; k++) { gb.coat[4 * px + k] = 0; gb.triA[4 * px + k] = 0; }
        gb.triB[2 * px] = 0; gb.triB[2 * px + 1] = 0;
        return;
    }

    const TriMeta tm = sc.triMeta[h.tri];
    if (picked) *picked = tm.mesh;
    const zr_mesh_instance& md = sc.instances[tm.mesh];
    uint32_t tri = tm.prim * 3 + md.base_idx_offset;
    const zr_vertex& V0 = sc.vertices[sc.indices[tri] + md.base_vtx_offset];
    const zr_vertex& V1 = sc.vertices[sc.indices[tri + 1] + md.base_vtx_offset];
    const zr_vertex& V2_ = sc.vertices[sc.indices[tri + 2] + md.base_vtx_offset];
    V4 q = normalize(DecodeNormalized4(md.rotation));
    const V3 scale = v3(zr_f16_to_f32(md.scale[0]), zr_f16_to_f32(md.scale[1]), zr_f16_to_f32(md.scale[2]));
    const V3 trn = v3p(md.translation);
    const V2 uv0 = v2(V0.uv[0], V0.uv[1]), uv1 = v2(V1.uv[0], V1.uv[1]), uv2 = v2(V2_.uv[0], V2_.uv[1]);

    V3 v0_n = DecodeOct32(V0.normal), v1_n = DecodeOct32(V1.normal), v2_n = DecodeOct32(V2_.normal);
    V3 normal = v0_n + h.u * (v1_n - v0_n) + h.v * (v2_n - v0_n);
    const V3 scaleInv = v3(1.0f / scale.x, 1.0f / scale.y, 1.0f / scale.z);
    normal = normalize(RotateVector(normal * scaleInv, q));

    const V2 uv = uv0 + h.u * (uv1 - uv0) + h.v * (uv2 - uv0);

    // tangent vector (GBufferRT_Inline.hlsl:147-155)
    const V3 v0_t = DecodeOct32(V0.tangent), v1_t = DecodeOct32(V1.tangent), v2_t = DecodeOct32(V2_.tangent);
    V3 tangent = v0_t + h.u * (v1_t - v0_t) + h.v * (v2_t - v0_t);
    tangent = normalize(RotateVector(tangent * scale, q));

    V3 v0W = TransformTRS(v3p(V0.pos), trn, q, scale);
    V3 v1W = TransformTRS(v3p(V1.pos), trn, q, scale);
    V3 v2W = TransformTRS(v3p(V2_.pos), trn, q, scale);
    V3 n0W = normalize(RotateVector(v0_n * scaleInv, q));
    V3 n1W = normalize(RotateVector(v1_n * scaleInv, q));
    V3 n2W = normalize(RotateVector(v2_n * scaleInv, q));
    TriDiffs td = ComputeTriDiffs(v0W, v1W, v2W, n0W, n1W, n2W, uv0, uv1, uv2);

    // motion vector
    V3 hitPos = mad(h.t, dir, origin);
    V3 posL = InverseTransformTRS(hitPos, trn, q, scale);
    V3 prevT = trn - v3(zr_f16_to_f32(md.d_translation[0]), zr_f16_to_f32(md.d_translation[1]), zr_f16_to_f32(md.d_translation[2]));
    V4 qp = normalize(DecodeNormalized4(md.prev_rotation));
    V3 ps = v3(zr_f16_to_f32(md.prev_scale[0]), zr_f16_to_f32(md.prev_scale[1]), zr_f16_to_f32(md.prev_scale[2]));
    V3 posPrev = TransformTRS(posL, prevT, qp, ps);
    V3 pvPrev = Mul3x4(g.prev_view, posPrev);
    V2 ndcPrev = v2(pvPrev.x / (pvPrev.z * g.tan_half_fov), pvPrev.y / (pvPrev.z * g.tan_half_fov));
    ndcPrev.x /= g.aspect_ratio;
    V2 currUV = v2(((float)x + 0.5f) / renderDim.x, ((float)y + 0.5f) / renderDim.y);
    V2 prevUV = UVFromNDC(ndcPrev) - v2(jitter.x / renderDim.x, jitter.y / renderDim.y);
    V2 motionVec = currUV - prevUV;

    V3 pos = mad(h.t, dir, origin);
    V3 posV = Mul3x4(g.curr_view, pos);
    float z = g.dof ? h.t : posV.z;
    V3 wo = origin - pos;

    // ApplyTextureMaps, GBufferRT.hlsli:178-282
    const zr_material mat = sc.materials[md.mat_idx];
    const uint32_t baseColorTex = mat.base_color_tex_subsurf_coat_weight & 0xffffu, normalTex = mat.normal_tex_tr_depth & 0xffffu;
    const uint32_t mrTex = mat.mr_tex_spec_roughness_coat_roughness & 0xffffu, emissiveTex = mat.emissive_tex_alpha_cutoff_coat_ior & 0xffffu;
    // (the gradients only feed SampleGrad; untextured materials skip them)
    F4 grads = f4(0, 0, 0, 0);
    if (baseColorTex != ZR_INVALID_TEX || normalTex != ZR_INVALID_TEX || mrTex != ZR_INVALID_TEX)
        grads = UVDifferentials((int)x, (int)y, origin, dir, g.dof != 0, lens, g.focus_depth, h.t, td.dpdu, td.dpdv, g);
    grads = f4(grads.x * g.camera_ray_uv_grads_scale, grads.y * g.camera_ray_uv_grads_scale,
               grads.z * g.camera_ray_uv_grads_scale, grads.w * g.camera_ray_uv_grads_scale);
    V3 baseColor = UnpackRGB8(mat.base_color_factor);
    V3 emissive = UnpackRGB8(mat.emissive_factor_normal_scale);
    float metallic = MatMetallic(mat) ? 1.0f : 0.0f;
    float roughness = MatRoughness(mat);
    V3 sn = normal;
    V3 dndu = td.dndu, dndv = td.dndv;
    if (baseColorTex != ZR_INVALID_TEX)
    {
        float c[4];
        zr_tex_sample_grad_aniso(&sc.tex, g.base_color_maps_desc_heap_offset + baseColorTex, uv.x, uv.y, grads.x, grads.y, grads.z, grads.w, 16, c);      // g_samAnisotropicWrap: MaxAnisotropy 16 (GBufferRT.hlsli:204-248, RendererCore.cpp:508-521)
        baseColor = baseColor * v3(c[0], c[1], c[2]);
    }
    // avoid normal mapping if tangent = (0, 0, 0), which results in NaN
    if (normalTex != ZR_INVALID_TEX && zr_abs(dot(tangent, tangent)) > 1e-6f)
    {
        float c[4];
        zr_tex_sample_grad_aniso(&sc.tex, g.normal_maps_desc_heap_offset + normalTex, uv.x, uv.y, grads.x, grads.y, grads.z, grads.w, 16, c);      // g_samAnisotropicWrap: MaxAnisotropy 16 (GBufferRT.hlsli:204-248, RendererCore.cpp:508-521)
        sn = TangentSpaceToWorldSpace(v2(c[0], c[1]), tangent, normal, zr_div255((float)(mat.emissive_factor_normal_scale >> 24)));
    }
    if (MatDoubleSided(mat) && dot(wo, normal) < 0) { sn = sn * -1.0f; dndu = dndu * -1.0f; dndv = dndv * -1.0f; }
    if (dot(wo, normal) > 0 && dot(wo, sn) < 0)
    {
        wo = normalize(wo);
        sn = sn - dot(sn, wo) * wo;
        sn = 1e-4f * wo + sn;
        sn = normalize(sn);
    }
    if (mrTex != ZR_INVALID_TEX)
    {
        float c[4];
        zr_tex_sample_grad_aniso(&sc.tex, g.metallic_roughness_maps_desc_heap_offset + mrTex, uv.x, uv.y, grads.x, grads.y, grads.z, grads.w, 16, c);      // g_samAnisotropicWrap: MaxAnisotropy 16 (GBufferRT.hlsli:204-248, RendererCore.cpp:508-521)
        metallic *= c[0];
        roughness *= c[1];
    }
    if (emissiveTex != ZR_INVALID_TEX)
    {
        float c[4];
        zr_tex_sample_level(&sc.tex, g.emissive_maps_desc_heap_offset + emissiveTex, uv.x, uv.y, 0.0f, c);
        emissive = emissive * v3(c[0], c[1], c[2]);
    }
    emissive = emissive * MatEmissiveStrength(mat);
    bool tr = MatTransmissive(mat);
    float ior = MatIOR(mat);
    float trDepth = tr ? MatTrDepth(mat) : 0;
    float subsurface = MatThinWalled(mat) ? MatSubsurface(mat) : 0;
    float coat_w = MatCoatWeight(mat);
    float encoded = EncodeMetallic(metallic, tr, emissive, trDepth, subsurface, coat_w);

    gb.depth[px] = z;
    V2 en = EncodeUnitVector(sn);
    gb.normal[px] = FloatToUNorm16(en.x) | (FloatToUNorm16(en.y) << 16);
    gb.baseColor[px] = FloatToUNorm8(baseColor.x) | (FloatToUNorm8(baseColor.y) << 8) | (FloatToUNorm8(baseColor.z) << 16) |
        ((subsurface > 0 ? FloatToUNorm8(subsurface) : 0u) << 24);
    gb.mr[px] = (uint16_t)(FloatToUNorm8(encoded) | (FloatToUNorm8(roughness) << 8));
    gb.emissive[px] = dot(emissive, emissive) > 0 ? PackR11G11B10F(vmax(emissive, 0.0f)) : 0u;
    gb.ior[px] = tr ? (uint8_t)FloatToUNorm8(EncodeIOR(ior)) : (uint8_t)0;
    if (coat_w > 0)
    {
        uint32_t c = Float3ToRGB8(UnpackRGB8(mat.coat_color_flags));
        gb.coat[4 * px + 0] = (uint16_t)(c & 0xffff);
        gb.coat[4 * px + 1] = (uint16_t)((c >> 16) | (FloatToUNorm8(coat_w) << 8));
        gb.coat[4 * px + 2] = (uint16_t)(FloatToUNorm8(MatCoatRoughness(mat)) | (FloatToUNorm8(EncodeIOR(MatCoatIOR(mat))) << 8));
        gb.coat[4 * px + 3] = 0;
    }
    else { for (int k = 0; k < 4; k++) gb.coat[4 * px + k] = 0; }
    gb.motion[px] = PackSnorm16(motionVec.x) | (PackSnorm16(motionVec.y) << 16);
    uint32_t a0 = zr_f32_to_f16(td.dpdu.x), a1 = zr_f32_to_f16(td.dpdu.y), a2 = zr_f32_to_f16(td.dpdu.z);
    uint32_t b0 = zr_f32_to_f16(td.dpdv.x), b1 = zr_f32_to_f16(td.dpdv.y), b2 = zr_f32_to_f16(td.dpdv.z);
    uint32_t c0 = zr_f32_to_f16(dndu.x), c1 = zr_f32_to_f16(dndu.y), c2 = zr_f32_to_f16(dndu.z);
    uint32_t d0 = zr_f32_to_f16(dndv.x), d1 = zr_f32_to_f16(dndv.y), d2 = zr_f32_to_f16(dndv.z);
    gb.triA[4 * px + 0] = a0 | (a1 << 16);
    gb.triA[4 * px + 1] = a2 | (b0 << 16);
    gb.triA[4 * px + 2] = b1 | (b2 << 16);
    gb.triA[4 * px + 3] = c0 | (c1 << 16);
    gb.triB[2 * px + 0] = c2 | (d0 << 16);
    gb.triB[2 * px + 1] = d1 | (d2 << 16);
}

// ---------------------------------------------------------------- wavefront path state
// flags word (s0.w)
static constexpr uint32_t PF_BOUNCE_MASK = 0xfu;
static constexpr uint32_t PF_IN_MEDIUM = 1u << 4;
static constexpr uint32_t PF_DRAIN = 1u << 5;       // no continuation ray: resolve pending NEE, write the pixel, retire
static constexpr uint32_t PF_PENDING = 1u << 6;     // the previous vertex left NEE contributions to resolve
static constexpr uint32_t PF_NLS = 1u << 7;         // numLightSamples of the pending vertex (0 or 1)
static constexpr uint32_t PF_MAXB_SHIFT = 8;        // bits 8..11 maxNumBounces
static constexpr uint32_t PF_S_RAY = 1u << 12;      // a light-segment ray is in flight for the pending vertex
static constexpr uint32_t PF_PARKED = 1u << 13;     // waiting for the Russian-roulette stage (group max not known yet)
static constexpr uint32_t PF_RD_PENDING = 1u << 14; // textured scenes: RayDifferentials::UpdateRays of the last vertex still to run

struct PathQueue
{
    U4* s0;   // pid, rngThread, rngGroup, flags
    F4* s1;   // li.xyz, eta_curr
    F4* s2;   // throughput.xyz (already multiplied by the continuation's bsdfOverPdf), misPdf
    F4* s3;   // pos.xyz (vertex the continuation ray leaves), sampleSetIdx bits (presampled light sets)
    F4* s4;   // wi.xyz (continuation direction), unused
    F4* s5;   // thrNEE.xyz (throughput at the pending vertex), unused
    F4* s6;   // misF.xyz, unused
    F4* s7;   // misWi.xyz, unused
    F4* s8;   // ldLight.xyz (unshadowed MIS-weighted light-sample contribution), unused
    // Textured scenes only (null otherwise): ray differentials (RT.hlsli:309-479) and what their deferred UpdateRays needs.
    // The reference calls UpdateRays after the continuation ray hit, with the NEW hit's triangle differentials
    // (PathTracing.hlsli:90-95), so it runs at the start of the next shade step (PF_RD_PENDING).
    //   t[0..3] = (origin_x, uv_grads.x), (dir_x, .y), (origin_y, .z), (dir_y, .w)
    //   t[4] = (normal of the vertex the C ray leaves, its surface.eta), t[5] = its surface.wo, t[6] = dpdx, t[7] = dpdy
    F4* t[8];
    // rays of this slot: (origin.xyz, tmin), (dir.xyz, tmax); tmax < 0 => no ray
    F4* rayC_o; F4* rayC_d; F4* rayM_o; F4* rayM_d; F4* rayS_o; F4* rayS_d;
    uint32_t* sLightID;   // emissive triangle ID the S ray is aimed at
    // results written by the trace kernel
    U4* hitC; U4* hitM;   // (t bits, u bits, v bits, global tri or kInvalidTri)
    uint32_t* visS;       // 1 = segment unoccluded
    // device only: compacted list of this queue's rays (slot | type << 30; type 0 = C, 1 = M, 2 = S), any order
    uint32_t* rayList;
};

struct PathOut   // what one shade/init step wants to write into the next queue
{
    bool alive;          // false: the path retired in this step (pixel written)
    U4 s0; F4 s1, s2, s3, s4, s5, s6, s7, s8;
    F4 rayC_o, rayC_d, rayM_o, rayM_d, rayS_o, rayS_d;
    uint32_t sLightID;
    F4 t[8];             // textured scenes only
};

ZR_HD void PackRayDiffs(const RayDiffs& rd, F4* t)
{
    t[0] = f4(rd.origin_x, rd.uv_grads.x); t[1] = f4(rd.dir_x, rd.uv_grads.y);
    t[2] = f4(rd.origin_y, rd.uv_grads.z); t[3] = f4(rd.dir_y, rd.uv_grads.w);
}
ZR_HD RayDiffs UnpackRayDiffs(F4 t0, F4 t1, F4 t2, F4 t3)
{
    RayDiffs rd; rd.origin_x = xyz(t0); rd.dir_x = xyz(t1); rd.origin_y = xyz(t2); rd.dir_y = xyz(t3);
    rd.uv_grads = v4(t0.w, t1.w, t2.w, t3.w);
    return rd;
}

ZR_HD void WritePath(const PathQueue& q, uint32_t slot, const PathOut& p, bool tex = false)
{
    if (tex) { for (int k = 0; k < 8; k++) q.t[k][slot] = p.t[k]; }
    q.s0[slot] = p.s0; q.s1[slot] = p.s1; q.s2[slot] = p.s2; q.s3[slot] = p.s3; q.s4[slot] = p.s4;
    q.s5[slot] = p.s5; q.s6[slot] = p.s6; q.s7[slot] = p.s7; q.s8[slot] = p.s8;
    q.rayC_o[slot] = p.rayC_o; q.rayC_d[slot] = p.rayC_d;
    q.rayM_o[slot] = p.rayM_o; q.rayM_d[slot] = p.rayM_d;
    q.rayS_o[slot] = p.rayS_o; q.rayS_d[slot] = p.rayS_d;
    q.sLightID[slot] = p.sLightID;
    // slots without a C ray read back as a miss (the trace stage only visits listed rays)
    if (p.rayC_d.w < 0) { U4 miss; miss.x = 0; miss.y = 0; miss.z = 0; miss.w = kInvalidTri; q.hitC[slot] = miss; }
}

struct PtParams
{
    uint32_t maxNonTrBounces, maxGlossyTrBounces;
    uint32_t russianRoulette;
    uint32_t numSampleSets;      // 0 when light presampling is off (then NEE draws from the alias table)
    uint32_t accumulate;         // g.accumulate && g.camera_static
    uint32_t tileW, groupsX;     // tile width in pixels / in 8x8 groups (group key of the RR reduction)
};

// closest-hit ray of Hit::FindClosest / Hit_Emissive::FindClosest (RayQuery.hlsli:26-48 / 156-174).
// strictZero: Hit::FindClosest treats ndotwi == 0 as "no ray" and < 0 as backface; Hit_Emissive uses <= 0 as backface.
ZR_HD bool MakeClosestRay(V3 pos, V3 normal, V3 wi, bool transmissive, bool emissiveVariant, F4* ro, F4* rd)
{
    float ndotwi = dot(normal, wi);
    bool back;
    if (emissiveVariant) back = ndotwi <= 0;
    else { if (ndotwi == 0) return false; back = ndotwi < 0; }
    if (back)
    {
        if (!transmissive) return false;
        normal = emissiveVariant ? normal * -1.0f : -normal;
    }
    V3 o = OffsetRayRTG(pos, normal);
    *ro = f4(o, back ? 5e-5f : 1e-6f);
    *rd = f4(wi, ZR_FLT_MAX);
    return true;
}

// Visibility_Segment with APPROXIMATE_EMISSIVE_SHADOW_RAY == 0 (RayQuery.hlsli:337-406): returns 0 = early-out
// "occluded", 1 = ray emitted
ZR_HD int MakeSegmentRay(V3 origin, V3 wi, float rayT, V3 normal, uint32_t triID, bool transmissive, F4* ro, F4* rd)
{
    if (triID == 0xffffffffu) return 0;
    if (rayT < 1e-6f) return 0;
    float ndotwi = dot(normal, wi);
    if (ndotwi == 0) return 0;
    if (ndotwi < 0)
    {
        if (transmissive) normal = normal * -1.0f;
        else return 0;
    }
    V3 o = OffsetRayRTG(origin, normal);
    *ro = f4(o, 3e-6f);
    *rd = f4(wi, rayT);
    return 1;
}

// Visibility_Ray (RayQuery.hlsli:302-334): any hit over ALL geometry to infinity.  Returns 0 = early-out "occluded",
// 1 = ray emitted.  The trace stage recognises it by sLightID == kVisibilityRayID.
static constexpr uint32_t kVisibilityRayID = 0xffffffffu;     // never a segment target: Visibility_Segment rejects it
ZR_HD int MakeVisibilityRay(V3 origin, V3 wi, V3 normal, bool transmissive, F4* ro, F4* rd)
{
    if (dot(normal, wi) <= 0)
    {
        if (transmissive) normal = normal * -1.0f;
        else return 0;
    }
    V3 o = OffsetRayRTG(origin, normal);
    *ro = f4(o, Lerp(0.0f, 8e-5f, dot(normal, wi)));
    *rd = f4(wi, ZR_FLT_MAX);
    return 1;
}

ZR_HD void WriteFinal(float* finalRGBA, uint32_t pid, V3 li, V3 firstBOP, bool accumulate)
{
    if (dot(li, li) > 0) li = li * firstBOP;
    li = any_nan(li) ? v3(0.0f) : li;
    float* o = finalRGBA + 4 * (size_t)pid;
    if (accumulate) { o[0] += li.x; o[1] += li.y; o[2] += li.z; }
    else { o[0] = li.x; o[1] = li.y; o[2] = li.z; }
}

// K9 prologue for one pixel: PathTracer.hlsl main (:115-198) + EstimateIndirectLighting (:56-79) up to the first
// FindClosest.  Writes the pixel directly when no path starts.
ZR_HD void PtInitPixel(const SceneView& sc, const zr_frame_constants& g, const GBuf& gb, const PtParams& prm, uint32_t x, uint32_t y,
    float* finalRGBA, F4* firstBOP, PathOut& out, bool tex = false)
{
    out.alive = false;
    const uint32_t pid = (y - gb.y0) * gb.w + (x - gb.x0);
    const uint16_t mrp = gb.mr[pid];
    const float mr_x = zr_div255((float)(mrp & 0xff)), mr_y = zr_div255((float)(mrp >> 8));
    const uint32_t fl = (uint32_t)zr_fma(mr_x, 255.0f, 0.5f);
    if (fl & (ZR_GBUF_INVALID | ZR_GBUF_EMISSIVE))
    {
        if (!prm.accumulate) { float* o = finalRGBA + 4 * (size_t)pid; o[0] = 0; o[1] = 0; o[2] = 0; }
        return;
    }
    // (gb.plain: the scene's material class, a compile-time constant in the PLAIN kernel permutations -- these flags then are known to be clear)
    const bool f_tr = !gb.plain && (fl & ZR_GBUF_TRANSMISSIVE), f_trDepth = !gb.plain && (fl & ZR_GBUF_TRDEPTH_GT0), f_metal = !gb.plain && (fl & ZR_GBUF_METALLIC);
    const V2 renderDim = v2((float)g.render_width, (float)g.render_height);
    const V2 jitter = v2(g.curr_camera_jitter[0], g.curr_camera_jitter[1]);
    const V3 vbx = Row3(g.curr_view, 0), vby = Row3(g.curr_view, 1), vbz = Row3(g.curr_view, 2);

    const float z_view = gb.depth[pid];
    V2 lens = v2(0, 0);
    V3 origin = v3p(g.camera_pos);
    if (g.dof)
    {
        uint32_t hx = x, hy = y, hz = x; zr_pcg3d(&hx, &hy, &hz);
        Rng r = Rng::Init(hz, hy, g.frame_num);
        lens = UniformSampleDiskConcentric(r.Uniform2D());
        lens = lens * g.lens_radius;
    }
    // Math::WorldPosFromScreenSpace2, Math.hlsli:218-248
    V3 pos;
    {
        V2 uv = v2(((float)x + 0.5f + jitter.x) / renderDim.x, ((float)y + 0.5f + jitter.y) / renderDim.y);
        V2 ndc = NDCFromUV(uv);
        V3 dir_w;
        if (!g.dof)
        {
            V3 dv = v3(ndc.x * g.aspect_ratio * g.tan_half_fov * z_view, ndc.y * g.tan_half_fov * z_view, z_view);
            dir_w = mad(dv.x, vbx, mad(dv.y, vby, dv.z * vbz));
        }
        else
        {
            V3 dv = v3(ndc.x * g.aspect_ratio * g.tan_half_fov, ndc.y * g.tan_half_fov, 1);
            dv = g.focus_depth * dv - v3(lens.x, lens.y, 0);
            dir_w = normalize(mad(dv.x, vbx, mad(dv.y, vby, dv.z * vbz)));
            dir_w = dir_w * z_view;
            origin = origin + mad(lens.x, vbx, lens.y * vby);
        }
        pos = origin + dir_w;
    }
    const V3 normal = DecodeOct32u(gb.normal[pid]);
    const V3 baseColor = UnpackRGB8(gb.baseColor[pid]);
    float eta_curr = kEtaAir, eta_next = kDefaultEtaMat;
    if (f_tr) eta_next = DecodeIOR(zr_div255((float)gb.ior[pid]));
    const V3 wo = normalize(origin - pos);
    Surface surface = InitSurface(normal, wo, f_metal, mr_y, baseColor, eta_curr, eta_next, f_tr, f_trDepth ? 1.0f : 0.0f,
        0.0f, 0.0f, v3(0.0f), 0.0f, kDefaultEtaCoat, gb.plain != 0);

    Rng rngGroup = Rng::Init((x >> 3) ^ 61u, (y >> 3) ^ 61u, g.frame_num);
    Rng rngThread = Rng::Init(x ^ 511u, y ^ 31u, g.frame_num);
    const uint32_t maxB = f_tr ? prm.maxGlossyTrBounces : prm.maxNonTrBounces;
    const uint32_t sampleSetIdx = rngGroup.UniformUintBounded_Faster(prm.numSampleSets);     // one group-RNG draw, always

    PrepareWo(sc.rho, surface);
    BsdfSample bs = SampleBSDF(sc.rho, normal, surface, rngThread);
    F4 ro, rd;
    bool ok = bs.pdf != 0;
    if (ok) ok = MakeClosestRay(pos, normal, bs.wi, surface.Transmissive(), false, &ro, &rd);
    if (!ok)
    {
        WriteFinal(finalRGBA, pid, v3(0.0f), v3(0.0f), prm.accumulate);
        return;
    }
    firstBOP[pid] = f4(bs.bsdfOverPdf, 0.0f);
    const bool tr0 = dot(normal, bs.wi) < 0;
    out.alive = true;
    out.s0.x = pid; out.s0.y = rngThread.s; out.s0.z = rngGroup.s;
    out.s0.w = 0u | (tr0 ? PF_IN_MEDIUM : 0u) | (maxB << PF_MAXB_SHIFT);
    out.s1 = f4(v3(0.0f), tr0 ? eta_next : kEtaAir);
    out.s2 = f4(v3(1.0f), 0.0f);
    out.s3 = f4(pos, zr_asfloat(sampleSetIdx));
    out.s4 = f4(bs.wi, 0.0f);
    out.s5 = f4(v3(0.0f), 0.0f); out.s6 = out.s5; out.s7 = out.s5; out.s8 = out.s5;
    out.rayC_o = ro; out.rayC_d = rd;
    out.rayM_o = f4(v3(0.0f), 0.0f); out.rayM_d = f4(v3(0.0f), -1.0f);
    out.rayS_o = out.rayM_o; out.rayS_d = out.rayM_d;
    out.sLightID = 0xffffffffu;
    if (tex)
    {
        // PathTracer.hlsl:170-190: camera ray differentials -> uv gradients at the primary hit -> differentials of the first
        // bounce (the reference does this only if the first bounce hits; a miss retires the path, so nothing reads them)
        const TriDiffs td = UnpackTriDiffs(&gb.triA[4 * pid], &gb.triB[2 * pid]);
        RayDiffs rdf = RayDiffs::Init((int)x, (int)y, renderDim, g.tan_half_fov, g.aspect_ratio, jitter, vbx, vby, vbz, g.dof != 0,
            g.focus_depth, lens, origin);
        V3 dpdx, dpdy;
        rdf.dpdx_dpdy(pos, normal, dpdx, dpdy);
        rdf.ComputeUVDifferentials(dpdx, dpdy, td.dpdu, td.dpdv);
        rdf.UpdateRays(pos, normal, bs.wi, surface.wo, td.dndu, td.dndv, dpdx, dpdy, dot(bs.wi, normal) < 0, surface.eta);
        PackRayDiffs(rdf, out.t);
        out.t[4] = f4(v3(0.0f), 0.0f); out.t[5] = out.t[4]; out.t[6] = out.t[4]; out.t[7] = out.t[4];
    }
}

ZR_HD void zr_atomic_max_u32(uint32_t* p, uint32_t v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    atomicMax(p, v);
#else
    if (v > *p) *p = v;
#endif
}

// Tail of the PathTrace loop body (PathTracing.hlsli:74-95): sample the continuation direction, emit the C ray,
// update throughput / medium bookkeeping speculatively (they only matter if the ray hits).
ZR_HD void PtContinue(const SceneView& sc, V3 n, const Surface& surface, V3 hitPos, float eta_curr, float eta_next, bool inMedium,
    V3 thr, uint32_t bounce, uint32_t maxB, uint32_t nflags, uint32_t pid, Rng& rngT, Rng& rngG, V3 li, float setIdxBits, PathOut& out)
{
    BsdfSample bs2 = InitBsdfSample();
    F4 cro = f4(v3(0.0f), 0.0f), crd = f4(v3(0.0f), -1.0f);
    if (bounce < maxB) bs2 = SampleBSDF(sc.rho, n, surface, rngT);
    bool cont = !(Luminance(bs2.bsdfOverPdf) == 0);
    if (cont) cont = MakeClosestRay(hitPos, n, bs2.wi, surface.Transmissive(), false, &cro, &crd);
    if (cont)
    {
        thr = thr * bs2.bsdfOverPdf;
        bool transmitted = dot(n, bs2.wi) < 0;
        eta_curr = transmitted ? (eta_curr == kEtaAir ? eta_next : kEtaAir) : eta_curr;
        inMedium = transmitted ? !inMedium : inMedium;
        nflags |= PF_RD_PENDING;        // only textured kernels look at it
    }
    else nflags |= PF_DRAIN;
    out.alive = true;
    out.s0.x = pid; out.s0.y = rngT.s; out.s0.z = rngG.s;
    out.s0.w = nflags | (bounce & PF_BOUNCE_MASK) | (inMedium ? PF_IN_MEDIUM : 0u) | (maxB << PF_MAXB_SHIFT);
    out.s1 = f4(li, eta_curr);
    out.s2 = f4(thr, out.s2.w);
    out.s3 = f4(hitPos, setIdxBits);
    out.s4 = f4(bs2.wi, 0.0f);
    out.rayC_o = cro; out.rayC_d = crd;
}

// One shade step for the path in slot `i` of `in`:
//   (1) resolve the pending NEE of the previous vertex (RGI_Util::NEE_Emissive_MIS tail, ReSTIR_GI_NEE.hlsli:40-64, 98-113)
//   (2) shade the continuation hit: GetMaterialData, NEE setup, Beer-Lambert, bounce bookkeeping, SampleBSDF
//       (ReSTIR_RT::PathTrace loop body, PathTracing.hlsli:25-95)
ZR_HD void PtShadePath(const SceneView& sc, const zr_frame_constants& g, const PtParams& prm, const PathQueue& in, uint32_t i,
    float* finalRGBA, const F4* firstBOP, uint32_t* groupMax, PathOut& out, bool tex = false)
{
    out.alive = false;
    const U4 s0 = in.s0[i];
    const uint32_t pid = s0.x;
    uint32_t flags = s0.w;
    const F4 s1 = in.s1[i];
    V3 li = xyz(s1);
    float eta_curr = s1.w;

    // ---- (1) pending NEE
    if (flags & PF_PENDING)
    {
        const float nls = (flags & PF_NLS) ? 1.0f : 0.0f;
        V3 ld = v3(0.0f);
        const U4 hm = in.hitM[i];
        if (in.rayM_d[i].w >= 0 && hm.w != kInvalidTri)
        {
            const TriMeta tm = sc.triMeta[hm.w];
            const uint32_t base = sc.instances[tm.mesh].base_emissive_tri_offset;
            if (base != 0xffffffffu)
            {
                const uint32_t eidx = base + tm.prim;
                const zr_emissive_triangle em = sc.emissives[eidx];
                V3 le = EmLe(sc, em, v2(zr_asfloat(hm.y), zr_asfloat(hm.z)));
                const V3 vtx0 = v3p(em.vtx0), vtx1 = EmV1(em), vtx2 = EmV2(em);
                V3 ln = cross(vtx1 - vtx0, vtx2 - vtx0);
                float twoArea = length(ln);
                twoArea = zr_max(twoArea, 1e-6f);
                ln = dot(ln, ln) == 0 ? v3(1.0f) : ln / twoArea;
                const V3 wi = xyz(in.s7[i]);
                ln = EmDoubleSided(em) && dot(-wi, ln) < 0 ? -ln : ln;
                const float lightSourcePdf = nls > 0 ? sc.alias[eidx].cached_p_orig : 0;
                const float lightPdf = lightSourcePdf * (2.0f / twoArea);
                const float t = zr_asfloat(hm.x);
                float dwdA = t > 0 ? zr_saturate(dot(ln, -wi)) / (t * t) : 0;
                float wiPdf = in.s2[i].w;
                wiPdf *= dwdA;
                le = le * (xyz(in.s6[i]) * dwdA);
                ld = PowerHeuristic(wiPdf, lightPdf, le, 1.0f, nls);
            }
        }
        bool addLight = true;
        if (flags & PF_S_RAY) addLight = in.visS[i] != 0;
        if (addLight) ld = ld + xyz(in.s8[i]);
        li = li + xyz(in.s5[i]) * ld;
    }

    const V3 fb = xyz(firstBOP[pid]);
    if (flags & PF_DRAIN) { WriteFinal(finalRGBA, pid, li, fb, prm.accumulate); return; }

    // ---- (2) continuation hit
    const U4 hc = in.hitC[i];
    if (hc.w == kInvalidTri) { WriteFinal(finalRGBA, pid, li, fb, prm.accumulate); return; }
    const V3 pos0 = xyz(in.s3[i]);
    const float setIdxBits = in.s3[i].w;
    const V3 wiC = xyz(in.s4[i]);
    V3 thr = xyz(in.s2[i]);
    const float t = zr_asfloat(hc.x);
    const V3 hitPos = mad(t, wiC, pos0);
    const TriMeta tm = sc.triMeta[hc.w];
    HitInfo hit;
    hit.t = t;
    V4 uvGrads = v4(0, 0, 0, 0);
    V3 dpdx = v3(0.0f), dpdy = v3(0.0f);
    if (tex)
    {
        FillHit<true>(sc, tm.mesh, tm.prim, zr_asfloat(hc.y), zr_asfloat(hc.z), false, hit);
        RayDiffs rdf = UnpackRayDiffs(in.t[0][i], in.t[1][i], in.t[2][i], in.t[3][i]);
        if (flags & PF_RD_PENDING)
        {
            // tail of the previous loop iteration (PathTracing.hlsli:90-95), with this hit's triangle differentials
            const F4 t4 = in.t[4][i];
            const V3 nPrev = xyz(t4);
            rdf.UpdateRays(pos0, nPrev, wiC, xyz(in.t[5][i]), hit.dndu, hit.dndv, xyz(in.t[6][i]), xyz(in.t[7][i]), dot(nPrev, wiC) < 0, t4.w);
        }
        rdf.dpdx_dpdy(hitPos, hit.normal, dpdx, dpdy);
        rdf.ComputeUVDifferentials(dpdx, dpdy, hit.dpdu, hit.dpdv);
        uvGrads = rdf.uv_grads;
        PackRayDiffs(rdf, out.t);
    }
    else FillHit<false>(sc, tm.mesh, tm.prim, zr_asfloat(hc.y), zr_asfloat(hc.z), false, hit);
    Surface surface; float eta_mat;
    if (!GetMaterialData(sc, -wiC, eta_curr, hit, surface, eta_mat, uvGrads, tex)) { WriteFinal(finalRGBA, pid, li, fb, prm.accumulate); return; }
    PrepareWo(sc.rho, surface);      // six evaluations of this vertex follow (NEE, its sampler pdf, the continuation's lobes): their wo-only terms once
    const float eta_next = eta_curr == kEtaAir ? eta_mat : kEtaAir;    // as computed inside GetMaterialData
    if (tex) { out.t[4] = f4(hit.normal, surface.eta); out.t[5] = f4(surface.wo, 0.0f); out.t[6] = f4(dpdx, 0.0f); out.t[7] = f4(dpdy, 0.0f); }

    Rng rngT = Rng::Seed(s0.y);
    Rng rngG = Rng::Seed(s0.z);
    uint32_t bounce = flags & PF_BOUNCE_MASK;
    const uint32_t maxB = (flags >> PF_MAXB_SHIFT) & 0xfu;
    bool inMedium = flags & PF_IN_MEDIUM;

    out.s5 = f4(thr, 0.0f);                                          // thrNEE: throughput before Beer-Lambert
    out.rayM_o = f4(v3(0.0f), 0.0f); out.rayM_d = f4(v3(0.0f), -1.0f);
    out.rayS_o = out.rayM_o; out.rayS_d = out.rayM_d;
    out.sLightID = 0xffffffffu;
    uint32_t nflags = PF_PENDING;

    const V3 n = hit.normal;
    V3 ldLight = v3(0.0f);
    if (g.num_emissive_triangles)
    {
        // NEE_Emissive_MIS<1, false> (ReSTIR_GI_NEE.hlsli:8-118), everything that does not need a trace result
        const bool specular = surface.GlossSpecular() && (surface.metallic || surface.specTr) && (!surface.Coated() || surface.CoatSpecular());
        const int numLightSamples = specular ? 0 : 1;
        if (numLightSamples) nflags |= PF_NLS;
        {
            BsdfSample bs = SampleBSDF(sc.rho, n, surface, rngT);
            out.s6 = f4(bs.f, 0.0f);
            out.s7 = f4(bs.wi, 0.0f);
            float misPdf = bs.pdf;
            F4 ro, rd;
            if (MakeClosestRay(hitPos, n, bs.wi, surface.Transmissive(), true, &ro, &rd)) { out.rayM_o = ro; out.rayM_d = rd; }
            out.s2.w = misPdf;
        }
        for (int s_l = 0; s_l < numLightSamples; s_l++)
        {
            V3 lpos, ln, le; float lightPdf; uint32_t lightID;
            if (prm.numSampleSets)      // USE_PRESAMPLED_SETS (ReSTIR_GI_NEE.hlsli:68-85)
            {
                PresampledLight pl = SamplePresampledSet(sc, zr_asuint(setIdxBits), hitPos, rngT);
                lpos = pl.pos; ln = pl.normal; le = pl.le; lightPdf = pl.pdf; lightID = pl.ID;
            }
            else
            {
                // Light::AliasTableSample::get, LightSource.hlsli:72-98
                uint32_t u0 = rngT.UniformUintBounded(g.num_emissive_triangles);
                const zr_alias_entry ae = sc.alias[u0];
                uint32_t lidx; float lpdfSrc;
                if (rngT.Uniform() < ae.p_curr) { lpdfSrc = ae.cached_p_orig; lidx = u0; }
                else { lpdfSrc = ae.cached_p_alias; lidx = ae.alias; }
                const zr_emissive_triangle em = sc.emissives[lidx];
                // Light::EmissiveTriSample::get, LightSource.hlsli:109-137
                V2 u = rngT.Uniform2D();
                V2 bary = UniformSampleTriangle(u);
                const V3 vtx0 = v3p(em.vtx0), vtx1 = EmV1(em), vtx2 = EmV2(em);
                lpos = (1.0f - bary.x - bary.y) * vtx0 + bary.x * vtx1 + bary.y * vtx2;
                ln = cross(vtx1 - vtx0, vtx2 - vtx0);
                bool normalIs0 = dot(ln, ln) == 0;
                float twoArea = length(ln);
                float lpdfPos = normalIs0 ? 0.0f : 2.0f / twoArea;
                ln = normalIs0 ? ln : ln / twoArea;
                ln = EmDoubleSided(em) && dot(hitPos - lpos, ln) < 0 ? -ln : ln;
                le = EmLe(sc, em, bary);
                lightPdf = lpdfSrc * lpdfPos;
                lightID = em.id;
            }
            const float tl = length(lpos - hitPos);
            const V3 wi = (lpos - hitPos) / tl;
            if (dot(ln, -wi) > 0)
            {
                const float dwdA = zr_saturate(dot(ln, -wi)) / (tl * tl);
                surface.SetWi(wi, n);
                const V3 fz = Unified(sc.rho, surface).f;
                le = le * (fz * dwdA);
                bool occludedEarly = false;
                if (dot(le, le) > 0)
                {
                    F4 ro, rd;
                    if (MakeSegmentRay(hitPos, wi, tl, n, lightID, surface.Transmissive(), &ro, &rd))
                    { out.rayS_o = ro; out.rayS_d = rd; out.sLightID = lightID; nflags |= PF_S_RAY; }
                    else occludedEarly = true;
                }
                float bsdfPdf = BSDFSamplerPdf_AtZ(sc.rho, n, surface, wi, fz, rngT);      // (the light direction's SetWi + Unified above, not evaluated again)
                bsdfPdf *= dwdA;
                if (occludedEarly) le = le * 0.0f;
                ldLight = ldLight + PowerHeuristic(lightPdf, bsdfPdf, le, (float)numLightSamples, 1.0f);
            }
        }
    }
    else
    {
        // RGI_Util::NEE with NEE_EMISSIVE == 0 (ReSTIR_GI_NEE.hlsli:194-226): sun with probability q, else the sky through
        // BSDF sampling with Le_Sky as the RIS target (NEE.hlsli:86-152); one visibility ray, resolved by the next step
        out.s6 = f4(v3(0.0f), 0.0f); out.s7 = out.s6; out.s2.w = 0;
        const float p_sun = rngT.Uniform();
        const V3 sunDir = v3p(g.sun_dir);
        if (-sunDir.y > 0)
        {
            const float q = (surface.Transmissive() ? 1.0f : (dot(-sunDir, n) > 0 ? 1.0f : 0.0f)) * 0.65f;   // P_SUN_VS_SKY
            V3 wi = -sunDir; bool wantRay; float invSel;
            if (p_sun < q)
            {
                Surface ss = surface;
                ss.SetWi(wi, n);
                const V3 f = Unified(sc.rho, ss).f;
                wantRay = !(dot(f, f) == 0);
                ldLight = wantRay ? f * Le_Sun(hitPos, g) : v3(0.0f);
                invSel = q;
            }
            else
            {
                SkyIncidentRadiance leFunc; leFunc.lut = sc.sky;
                const BsdfSample bs = SampleBSDF(sc.rho, n, surface, leFunc, rngT);
                ldLight = bs.bsdfOverPdf; wi = bs.wi;
                wantRay = dot(ldLight, ldLight) > 0;
                invSel = 1 - q;
            }
            if (wantRay)
            {
                F4 ro, rd;
                if (MakeVisibilityRay(hitPos, wi, n, surface.Transmissive(), &ro, &rd))
                { out.rayS_o = ro; out.rayS_d = rd; out.sLightID = kVisibilityRayID; nflags |= PF_S_RAY; }
                else ldLight = ldLight * 0.0f;
            }
            ldLight = ldLight / invSel;
        }
    }
    out.s8 = f4(ldLight, 0.0f);

    // Beer-Lambert (ACCOUNT_FOR_TRANSMITTANCE == 1, PathTracing.hlsli:43-50)
    if (inMedium && (surface.trDepth > 0))
    {
        V3 ext = -vlog(surface.base) / surface.trDepth;
        thr = thr * vexp(-t * ext);
    }

    out.alive = true;
    out.s1 = f4(li, eta_curr);
    out.s3 = f4(hitPos, setIdxBits);
    if (bounce >= (maxB - 1))
    {
        // PathTracing.hlsli:53-54: the path ends here; one more round drains the pending NEE of this vertex
        out.s0.x = pid; out.s0.y = rngT.s; out.s0.z = rngG.s;
        out.s0.w = nflags | PF_DRAIN | (bounce & PF_BOUNCE_MASK) | (inMedium ? PF_IN_MEDIUM : 0u) | (maxB << PF_MAXB_SHIFT);
        out.s2 = f4(thr, out.s2.w);
        out.s4 = f4(v3(0.0f), 0.0f);
        out.rayC_o = f4(v3(0.0f), 0.0f); out.rayC_d = f4(v3(0.0f), -1.0f);
        return;
    }
    bounce++;
    if (prm.russianRoulette && bounce >= 3u)
    {
        // Russian roulette (PathTracing.hlsli:62-72) needs WaveActiveMax(luminance(throughput)) over the 8x8 group:
        // publish this lane's value, park the path; PtRussianRoulette (next kernel) finishes the vertex.
        const uint32_t lx = pid % prm.tileW, ly = pid / prm.tileW;
        zr_atomic_max_u32(&groupMax[(ly >> 3) * prm.groupsX + (lx >> 3)], zr_asuint(Luminance(thr)));
        out.s0.x = pid; out.s0.y = rngT.s; out.s0.z = rngG.s;
        out.s0.w = nflags | PF_PARKED | (bounce & PF_BOUNCE_MASK) | (inMedium ? PF_IN_MEDIUM : 0u) | (maxB << PF_MAXB_SHIFT);
        out.s2 = f4(thr, out.s2.w);
        out.s4 = f4(wiC, 0.0f);                                            // incoming direction, needed to rebuild the surface
        out.rayC_o = f4(zr_asfloat(hc.y), zr_asfloat(hc.z), zr_asfloat(hc.w), t);   // hit record (bary, triangle, t)
        out.rayC_d = f4(v3(0.0f), -1.0f);
        return;
    }
    PtContinue(sc, n, surface, hitPos, eta_curr, eta_next, inMedium, thr, bounce, maxB, nflags, pid, rngT, rngG, li, setIdxBits, out);
}

// Russian-roulette stage for the parked path in slot `i` of `q` (in place): PathTracing.hlsli:62-72, then the loop tail.
// `groupMax` holds, per 8x8 group, the max luminance(throughput) over the lanes that reached the RR block this round
// (the reference's WaveActiveMax; lanes = pixels of the 8x8 thread group).
// Returns true when the path continues, i.e. slot i now holds a C ray.
ZR_HD bool PtRussianRoulette(const SceneView& sc, const PtParams& prm, const PathQueue& q, uint32_t i, const uint32_t* groupMax, bool tex = false)
{
    const U4 s0 = q.s0[i];
    if (!(s0.w & PF_PARKED)) return false;
    const uint32_t pid = s0.x;
    const uint32_t lx = pid % prm.tileW, ly = pid / prm.tileW;
    const float waveThroughput = zr_asfloat(groupMax[(ly >> 3) * prm.groupsX + (lx >> 3)]);
    Rng rngT = Rng::Seed(s0.y), rngG = Rng::Seed(s0.z);
    uint32_t flags = s0.w & ~PF_PARKED;
    const uint32_t bounce = flags & PF_BOUNCE_MASK, maxB = (flags >> PF_MAXB_SHIFT) & 0xfu;
    const bool inMedium = flags & PF_IN_MEDIUM;
    const uint32_t nflags = flags & (PF_PENDING | PF_NLS | PF_S_RAY);
    const float p_terminate = zr_max(0.05f, 1 - waveThroughput);
    if (rngG.Uniform() < p_terminate)
    {
        U4 o = s0; o.z = rngG.s; o.w = flags | PF_DRAIN;
        q.s0[i] = o;
        q.rayC_d[i] = f4(v3(0.0f), -1.0f);
        return false;
    }
    const F4 s2 = q.s2[i];
    V3 thr = xyz(s2) / (1 - p_terminate);
    const F4 s1 = q.s1[i];
    const float eta_curr = s1.w;
    const V3 hitPos = xyz(q.s3[i]), wiC = xyz(q.s4[i]);
    const float setIdxBits = q.s3[i].w;
    const F4 rec = q.rayC_o[i];
    const uint32_t tri = zr_asuint(rec.z);
    const TriMeta tm = sc.triMeta[tri];
    HitInfo hit; hit.t = rec.w;
    FillHit<false>(sc, tm.mesh, tm.prim, rec.x, rec.y, false, hit);
    Surface surface; float eta_mat;
    // (succeeded once already in PtShadePath; the uv gradients of this vertex are still in the slot)
    GetMaterialData(sc, -wiC, eta_curr, hit, surface, eta_mat, tex ? v4(q.t[0][i].w, q.t[1][i].w, q.t[2][i].w, q.t[3][i].w) : v4(0, 0, 0, 0), tex);
    const float eta_next = eta_curr == kEtaAir ? eta_mat : kEtaAir;
    PathOut po; po.s2.w = s2.w;
    PtContinue(sc, hit.normal, surface, hitPos, eta_curr, eta_next, inMedium, thr, bounce, maxB, nflags, pid, rngT, rngG, xyz(s1), setIdxBits, po);
    q.s0[i] = po.s0; q.s1[i] = po.s1; q.s2[i] = po.s2; q.s4[i] = po.s4; q.rayC_o[i] = po.rayC_o; q.rayC_d[i] = po.rayC_d;
    return po.rayC_d.w >= 0;
}

// trace stage for one ray of a queue slot
ZR_HD U4 PackRawHit(const RawHit& h)
{ U4 r; r.x = zr_asuint(h.tri == kInvalidTri ? 0.0f : h.t); r.y = zr_asuint(h.u); r.z = zr_asuint(h.v); r.w = h.tri; return r; }
ZR_HD U4 TraceClosestRay(const SceneView& sc, F4 ro, F4 rd, uint32_t mask, TravStack stack)
{ return PackRawHit(Traverse<false>(sc, xyz(ro), xyz(rd), ro.w, rd.w, mask, stack)); }
// S rays.  Segment to an emissive triangle, Visibility_Segment tail (RayQuery.hlsli:392-405): closest hit over NON_EMISSIVE
// geometry, visible iff no hit or the hit's hashed ID equals the light's.  Sun / sky, Visibility_Ray (RayQuery.hlsli:317-333,
// lightID == kVisibilityRayID): any hit over ALL geometry, visible iff none.
ZR_HD uint32_t SegmentVisible(const SceneView& sc, const RawHit& h, uint32_t lightID)
{
    if (h.tri == kInvalidTri) return 1u;
    if (lightID == kVisibilityRayID) return 0u;
    const TriMeta tm = sc.triMeta[h.tri];
    return TriID(tm.mesh, tm.prim) == lightID ? 1u : 0u;
}
ZR_HD uint32_t TraceSegmentRay(const SceneView& sc, F4 ro, F4 rd, uint32_t lightID, TravStack stack)
{
    const bool vis = lightID == kVisibilityRayID;
    return SegmentVisible(sc, TraverseDyn(sc, xyz(ro), xyz(rd), ro.w, rd.w, vis ? ZR_SUBGROUP_ALL : ZR_SUBGROUP_NON_EMISSIVE, stack, vis), lightID);
}

// K3: PresampleEmissives.hlsl:20-44 -- sample i of numSets * setSize
ZR_HD zr_presampled_tri PresampleEmissive(const SceneView& sc, uint32_t i, uint32_t frameNum, uint32_t numEmissives)
{
    Rng rng; rng.s = zr_pcg(i + zr_pcg(frameNum));                 // RNG::Init(idx, frame), Sampling.hlsli:52-60
    uint32_t u0 = rng.UniformUintBounded(numEmissives);
    const zr_alias_entry ae = sc.alias[u0];
    uint32_t lidx; float lpdfSrc;
    if (rng.Uniform() < ae.p_curr) { lpdfSrc = ae.cached_p_orig; lidx = u0; }
    else { lpdfSrc = ae.cached_p_alias; lidx = ae.alias; }
    const zr_emissive_triangle em = sc.emissives[lidx];
    V2 bary = UniformSampleTriangle(rng.Uniform2D());
    const V3 vtx0 = v3p(em.vtx0), vtx1 = EmV1(em), vtx2 = EmV2(em);
    V3 lpos = (1.0f - bary.x - bary.y) * vtx0 + bary.x * vtx1 + bary.y * vtx2;
    V3 ln = cross(vtx1 - vtx0, vtx2 - vtx0);
    bool normalIs0 = dot(ln, ln) == 0;
    float twoArea = length(ln);
    float lpdfPos = normalIs0 ? 0.0f : 2.0f / twoArea;
    ln = normalIs0 ? ln : ln / twoArea;                            // reverseNormalIfTwoSided == false
    V3 le = EmLe(sc, em, bary);
    zr_presampled_tri s;
    s.pos[0] = lpos.x; s.pos[1] = lpos.y; s.pos[2] = lpos.z;
    V2 e = EncodeUnitVector(ln);
    s.normal[0] = (uint16_t)FloatToUNorm16(e.x); s.normal[1] = (uint16_t)FloatToUNorm16(e.y);
    s.le[0] = zr_f32_to_f16(le.x); s.le[1] = zr_f32_to_f16(le.y); s.le[2] = zr_f32_to_f16(le.z);
    s.bary[0] = (uint16_t)FloatToUNorm16(bary.x); s.bary[1] = (uint16_t)FloatToUNorm16(bary.y);
    s.two_sided = EmDoubleSided(em) ? 1 : 0;
    s.idx = lidx; s.id = em.id;
    s.pdf = lpdfSrc * lpdfPos;
    return s;
}

// Compositing.hlsl:30-125 for one pixel (in-scattering off).  Miss pixels show Le_SkyWithSunDisk when a direct-lighting term is bound
// (:43-48); `sky` = the scene's sky-view LUT (data == null -> 0: no ZR_PASS_SKY rendered)
ZR_HD F4 CompositePixel(const zr_frame_constants& g, uint16_t mrp, const F4* skyDI, const F4* emissiveDI, const F4* indirect, size_t px, F4 prevOut,
    const SkyLutView& sky, uint32_t x, uint32_t y)
{
    const uint32_t fl = (uint32_t)zr_fma(zr_div255((float)(mrp & 0xff)), 255.0f, 0.5f);
    const bool accumulate = g.accumulate && g.camera_static;
    if ((fl & ZR_GBUF_INVALID) && !accumulate)
    {
        const bool dirLighting = skyDI || emissiveDI;
        return f4((dirLighting && sky.data) ? Le_SkyWithSunDisk(sky, g, x, y) : v3(0.0f), prevOut.w);
    }
    const uint32_t numFramesAccumulated = accumulate ? g.num_frames_camera_static : 1u;
    V3 color = v3(0.0f);
    if (skyDI) color = xyz(skyDI[px]);
    else if (emissiveDI) color = color + xyz(emissiveDI[px]);
    if (indirect && !(fl & ZR_GBUF_EMISSIVE)) color = color + xyz(indirect[px]);
    color = color / (float)numFramesAccumulated;
    return f4(color, prevOut.w);
}

// FireflyFilter.hlsl:33-85 (the ReLAX firefly clamp): the centre colour is clamped to the [min, max]-luminance colours of its 3 x 3
// neighbours that have geometry (depth != FLT_MAX).  `lum` / `col` / `dep` address a tile with a 1-texel border: (tx, ty) = centre.
// Pinned: the reference filters the composited UAV in place (neighbours may or may not be filtered already); here every pixel
// reads the unfiltered image (Jacobi), which is the only order-independent reading.
template<typename TileC, typename TileD>
ZR_HD V3 FireflyClamp(const TileC& col, const TileD& dep, int tx, int ty, int x, int y, int W, int H, V3 currColor)
{
    float minLum = ZR_FLT_MAX, maxLum = 0.0f;
    V3 minColor = currColor, maxColor = v3(0.0f);
    const float currLum = Luminance(currColor);
    for (int i = -1; i <= 1; i++)
        for (int j = -1; j <= 1; j++)
        {
            if (i == 0 && j == 0) continue;
            const int ax = x + j, ay = y + i;
            if (ax < 0 || ay < 0 || ax >= W || ay >= H) continue;      // (uint compare in the reference: negative addresses fail it too)
            if (dep(tx + j, ty + i) == ZR_FLT_MAX) continue;
            const V3 nc = col(tx + j, ty + i);
            const float nl = Luminance(nc);
            if (nl < minLum) { minLum = nl; minColor = nc; }
            else if (nl > maxLum) { maxLum = nl; maxColor = nc; }
        }
    V3 ret = currLum < minLum ? minColor : (currLum > maxLum ? maxColor : currColor);
    ret = minLum <= maxLum ? ret : currColor;
    return ret;
}

// K2: EstimateTriEmissivePower.hlsl:29-79.  Textured triangles: 32 lanes x 2 Halton(2, 3) points (PreLighting.cpp:236-243,
// Sampling.cpp:160-174), emissive map at mip 0 through g_samLinearWrap; the WaveActiveSum over the lane partials is pinned
// to ascending lane order (one thread per triangle here).
ZR_HD float Halton(int i, int b)
{
    float f = 1.0f, r = 0.0f; const float bf = (float)b;
    while (i > 0) { f /= bf; r = r + f * (float)(i % b); i = (int)((float)i / bf); }
    return r;
}
ZR_HD float EstimateTriPower(const SceneView& sc, const zr_emissive_triangle& em)
{
    V3 power = v3(64.0f);
    const uint32_t emissiveTex = em.packed_b & 0xffffu;
    if (emissiveTex != ZR_INVALID_TEX)
    {
        power = v3(0.0f);
        for (int lane = 0; lane < 32; lane++)
        {
            V3 lanePower = v3(0.0f);
            for (int k = 0; k < 2; k++)
            {
                const int si = lane * 2 + k;
                const V2 bary = UniformSampleTriangle(v2(Halton(si + 1, 2), Halton(si + 1, 3)));
                const V2 texUV = (1.0f - bary.x - bary.y) * EmUV(em.uv0) + bary.x * EmUV(em.uv1) + bary.y * EmUV(em.uv2);
                float c[4];
                zr_tex_sample_level(&sc.tex, sc.emissiveMapsOffset + emissiveTex, texUV.x, texUV.y, 0.0f, c);
                lanePower = lanePower + v3(c[0], c[1], c[2]);
            }
            power = power + lanePower;
        }
    }
    power = power * UnpackRGB8(em.packed_a) * zr_f16_to_f32((uint16_t)(em.packed_b >> 16));
    const V3 vtx0 = v3p(em.vtx0), vtx1 = EmV1(em), vtx2 = EmV2(em);
    const float area = 0.5f * length(cross(vtx1 - vtx0, vtx2 - vtx0));
    const float pdf = area > 0 ? 1.0f / area : 0;
    return pdf > 0 ? Luminance(power) * ZR_PI / (pdf * 64.0f) : 0;
}

} // namespace zr
