// zr_dev_bsdf.h -- device-side OpenPBR-style layered BSDF: evaluation and lobe-RIS sampling.
//
// MI355X-native restatement of Source/ZetaRenderPass/Common/BSDF.hlsli (eval) and BSDFSampling.hlsli (sampling,
// sampler-pdf replay).  Scalar per lane -- there is no dense contraction here, so no MFMA; the kernels that call this
// are bound by HBM state traffic + VALU.  The 3-D reflectance LUT (rho.dds, 64x32x16 R16_UNORM)
// is 64 KiB and is read through a plain pointer: it stays resident in L2 / can be staged in LDS by the caller.
#pragma once
#include "zr_dev_math.h"

namespace zr {

static constexpr float kMinMetalnessMetal = 0.9f;     // Material.h:5-17
static constexpr float kMinIOR = 1.0f;
static constexpr float kMaxIOR = 2.5f;
static constexpr float kDefaultEtaMat = 1.5f;
static constexpr float kDefaultEtaCoat = 1.6f;
static constexpr float kEtaAir = 1.0f;
static constexpr float kMinNdotHSpecular = 0.99998f;  // BSDF.hlsli:31-35
static constexpr float kMaxAlphaSpecular = 0.0016f;

enum Lobe : uint32_t { LOBE_DIFFUSE_R = 0, LOBE_DIFFUSE_T = 1, LOBE_GLOSSY_R = 2, LOBE_GLOSSY_T = 3, LOBE_COAT = 4, LOBE_ALL = 5 };

struct RhoView { const uint16_t* data; uint32_t dx, dy, dz; };

// Texture3D.SampleLevel(g_samLinearClamp, uvw, 0) pinned to fp32 trilinear, texel centres at (i + 0.5) / N, clamp
ZR_HD float SampleRho(const RhoView& lut, float u, float v, float w)
{
    float x = u * (float)lut.dx - 0.5f, y = v * (float)lut.dy - 0.5f, z = w * (float)lut.dz - 0.5f;
    float fx = zr_floor(x), fy = zr_floor(y), fz = zr_floor(z);
    float tx = x - fx, ty = y - fy, tz = z - fz;
    int ix = (int)fx, iy = (int)fy, iz = (int)fz;
    int hx = (int)lut.dx - 1, hy = (int)lut.dy - 1, hz = (int)lut.dz - 1;
    int x0 = ix < 0 ? 0 : (ix > hx ? hx : ix), x1 = ix + 1 < 0 ? 0 : (ix + 1 > hx ? hx : ix + 1);
    int y0 = iy < 0 ? 0 : (iy > hy ? hy : iy), y1 = iy + 1 < 0 ? 0 : (iy + 1 > hy ? hy : iy + 1);
    int z0 = iz < 0 ? 0 : (iz > hz ? hz : iz), z1 = iz + 1 < 0 ? 0 : (iz + 1 > hz ? hz : iz + 1);
    const uint32_t sy = lut.dx, sz = lut.dx * lut.dy;
#define ZR_RHO(X, Y, Z) (zr_div65535((float)lut.data[(uint32_t)(Z) * sz + (uint32_t)(Y) * sy + (uint32_t)(X)]))
    float c00 = zr_lerp(ZR_RHO(x0, y0, z0), ZR_RHO(x1, y0, z0), tx);
    float c10 = zr_lerp(ZR_RHO(x0, y1, z0), ZR_RHO(x1, y1, z0), tx);
    float c01 = zr_lerp(ZR_RHO(x0, y0, z1), ZR_RHO(x1, y0, z1), tx);
    float c11 = zr_lerp(ZR_RHO(x0, y1, z1), ZR_RHO(x1, y1, z1), tx);
#undef ZR_RHO
    return zr_lerp(zr_lerp(c00, c10, ty), zr_lerp(c01, c11, ty), tz);
}

ZR_HD float DielectricF0(float eta) { float f0 = (eta - 1) / (eta + 1); return f0 * f0; }      // BSDF.hlsli:113-117
ZR_HD V3 FresnelSchlick(V3 F0, float whdotwx)                                                  // :122-127
{
    float tmp = 1.0f - whdotwx; float tmpSq = tmp * tmp; float k = tmpSq * tmpSq * tmp;
    return v3(zr_fma(k, 1 - F0.x, F0.x), zr_fma(k, 1 - F0.y, F0.y), zr_fma(k, 1 - F0.z, F0.z));
}
ZR_HD float FresnelSchlick_Dielectric(float F0, float whdotwx)                                 // :130-135
{ float tmp = 1.0f - whdotwx; float tmpSq = tmp * tmp; return zr_fma(tmpSq * tmpSq * tmp, 1 - F0, F0); }
ZR_HD float Fresnel_Dielectric(float ndotwi, float eta, float cosTheta_t)                      // :156-163
{
    float r_par = zr_fma(-eta, cosTheta_t, ndotwi) / zr_fma(eta, cosTheta_t, ndotwi);
    float r_perp = zr_fma(eta, ndotwi, -cosTheta_t) / zr_fma(eta, ndotwi, cosTheta_t);
    return 0.5f * (r_par * r_par + r_perp * r_perp);
}
ZR_HD float GGX(float ndotwh, float alphaSq)                                                   // :169-173
{ float denom = zr_fma(ndotwh * ndotwh, alphaSq - 1.0f, 1.0f); return alphaSq / (ZR_PI * denom * denom); }
ZR_HD float SmithG1(float alphaSq, float ndotx)                                                // :185-190
{
    float ndotxSq = ndotx * ndotx;
    float tanThetaSq = (1.0f - ndotxSq) / ndotxSq;
    return 2.0f / (zr_sqrt(zr_fma(alphaSq, tanThetaSq, 1.0f)) + 1.0f);
}
ZR_HD float SmithG2_Opt(float n, float alphaSq, float ndotwi, float ndotwo)                    // :209-216
{
    float denomWo = ndotwi * zr_sqrt(zr_fma(zr_fma(-ndotwo, alphaSq, ndotwo), ndotwo, alphaSq));
    float denomWi = ndotwo * zr_sqrt(zr_fma(zr_fma(-ndotwi, alphaSq, ndotwi), ndotwi, alphaSq));
    return (0.5f * n) / (denomWo + denomWi);
}
ZR_HD float SmithG2OverG1(float alphaSq, float ndotwi, float ndotwo)                           // :220-226
{
    float G1wi = SmithG1(alphaSq, ndotwi), G1wo = SmithG1(alphaSq, ndotwo);
    return G1wi / (G1wi + G1wo - G1wi * G1wo);
}
ZR_HD float GGXReflectance_Dielectric(const RhoView& rho, float alpha, float ndotwo, float eta) // :279-296
{
    float v = ((alpha - 0.002025f) / (1.0f - 0.002025f));
    float w = ((eta - 0.5f) / (1.99f - 0.5f));
    return zr_saturate(SampleRho(rho, ndotwo, v, w));
}
ZR_HD float E_FON_approx(float cosTheta, float roughness)                                      // :335-345
{
    float mucomp = 1.0f - cosTheta, mucomp2 = mucomp * mucomp;
    float qx = 0.0571085289f * mucomp + 0.491881867f * mucomp2;
    float qy = -0.332181442f * mucomp + 0.0714429953f * mucomp2;
    float GoverPi = qx * 1.0f + qy * mucomp2;
    return zr_fma(roughness, GoverPi, 1.0f) / zr_fma(0.287793398f, roughness, 1.0f);
}
ZR_HD V3 OrenNayar(bool multiScatter, V3 rho, float sigma, float ndotwo, float ndotwi, float wodotwi, float g_wo)  // :354-389
{
    if (sigma == 0) return ZR_ONE_OVER_PI * ndotwi * rho;
    float A = 1.0f / zr_fma(0.287793398f, sigma, 1.0f);
    float B = sigma * A;
    float s_over_t = zr_fma(-ndotwi, ndotwo, wodotwi);
    s_over_t = s_over_t > 0 ? s_over_t / zr_max(ndotwi, ndotwo) : s_over_t;
    V3 f = v3(ZR_ONE_OVER_PI * zr_fma(B, s_over_t, A));
    V3 f_comp = v3(0.0f);
    if (multiScatter)
    {
        float avgR = zr_fma(0.0724882111f, B, A);
        float one_min = 1 - avgR;
        float tmp = ZR_ONE_OVER_PI * (avgR / one_min);
        V3 rho_ms = v3(tmp / zr_fma(-rho.x, one_min, 1.0f), tmp / zr_fma(-rho.y, one_min, 1.0f), tmp / zr_fma(-rho.z, one_min, 1.0f));
        rho_ms = rho_ms * rho;
        float E_wo = g_wo;
        float E_wi = E_FON_approx(ndotwi, sigma);
        f_comp = (1 - E_wo) * (1 - E_wi) * rho_ms;
    }
    return ndotwi * (f + f_comp) * rho;
}
// OrenNayar on a prepared surface (Surface::woMask & WO_DIFFUSE): sigma, A and the multi-scatter albedo term come in, everything else as above
ZR_HD V3 OrenNayarPrepared(bool multiScatter, V3 rho, float sigma, float A, V3 rho_ms, float ndotwo, float ndotwi, float wodotwi, float g_wo)
{
    if (sigma == 0) return ZR_ONE_OVER_PI * ndotwi * rho;
    float B = sigma * A;
    float s_over_t = zr_fma(-ndotwi, ndotwo, wodotwi);
    s_over_t = s_over_t > 0 ? s_over_t / zr_max(ndotwi, ndotwo) : s_over_t;
    V3 f = v3(ZR_ONE_OVER_PI * zr_fma(B, s_over_t, A));
    V3 f_comp = v3(0.0f);
    if (multiScatter)
    {
        float E_wo = g_wo;
        float E_wi = E_FON_approx(ndotwi, sigma);
        f_comp = (1 - E_wo) * (1 - E_wi) * rho_ms;
    }
    return ndotwi * (f + f_comp) * rho;
}
ZR_HD V3 GGXMicrofacetBRDF(float alpha, float ndotwh, float ndotwo, float ndotwi, V3 fr, bool specular)   // :392-413
{
    if (specular) return (ndotwh >= kMinNdotHSpecular ? 1.0f : 0.0f) * fr;
    float alphaSq = alpha * alpha;
    float f = GGX(ndotwh, alphaSq) * SmithG2_Opt(1.0f, alphaSq, ndotwi, ndotwo) * ndotwi;
    return f * fr;
}
// SmithG2_Opt with the wo factor of denomWo prepared (Surface::c_smith_wo)
ZR_HD float SmithG2_OptPrepared(float n, float alphaSq, float ndotwi, float ndotwo, float smith_wo)
{
    float denomWo = ndotwi * smith_wo;
    float denomWi = ndotwo * zr_sqrt(zr_fma(zr_fma(-ndotwi, alphaSq, ndotwi), ndotwi, alphaSq));
    return (0.5f * n) / (denomWo + denomWi);
}
ZR_HD V3 GGXMicrofacetBRDF_Prepared(float alpha, float ndotwh, float ndotwo, float ndotwi, V3 fr, bool specular, float smith_wo)
{
    if (specular) return (ndotwh >= kMinNdotHSpecular ? 1.0f : 0.0f) * fr;
    float alphaSq = alpha * alpha;
    float f = GGX(ndotwh, alphaSq) * SmithG2_OptPrepared(1.0f, alphaSq, ndotwi, ndotwo, smith_wo) * ndotwi;
    return f * fr;
}
ZR_HD float JacobianHalfVecToIncident_Tr(float eta, float whdotwo, float whdotwi)              // :420-427
{
    float denom = zr_fma(whdotwo, 1 / eta, whdotwi);
    denom *= denom;
    return denom > 0 ? whdotwi / denom : 0;
}
ZR_HD float GGXMicrofacetBTDF(float alpha, float ndotwh, float ndotwo, float ndotwi, float whdotwo, float whdotwi,
    float eta, float fr, bool specular)                                                        // :430-458
{
    if (specular) { float f = ndotwh >= kMinNdotHSpecular ? 1.0f : 0.0f; return f * (1 - fr); }
    float alphaSq = alpha * alpha;
    float f = GGX(ndotwh, alphaSq) * SmithG2_Opt(4.0f, alphaSq, ndotwi, ndotwo) * whdotwo;
    f *= JacobianHalfVecToIncident_Tr(eta, whdotwo, whdotwi);
    f *= ndotwi;
    return f * (1 - fr);
}
ZR_HD float GGXMicrofacetBTDF_Prepared(float alpha, float ndotwh, float ndotwo, float ndotwi, float whdotwo, float whdotwi,
    float eta, float fr, bool specular, float smith_wo)
{
    if (specular) { float f = ndotwh >= kMinNdotHSpecular ? 1.0f : 0.0f; return f * (1 - fr); }
    float alphaSq = alpha * alpha;
    float f = GGX(ndotwh, alphaSq) * SmithG2_OptPrepared(4.0f, alphaSq, ndotwi, ndotwo, smith_wo) * whdotwo;
    f *= JacobianHalfVecToIncident_Tr(eta, whdotwo, whdotwi);
    f *= ndotwi;
    return f * (1 - fr);
}
ZR_HD V3 SampleGGXVNDF(V3 wo, float ax, float ay, V2 u)                                         // :464-484
{
    V3 Vh = normalize(v3(ax * wo.x, ay * wo.y, wo.z));
    float phi = ZR_TWO_PI * u.x;
    float z = zr_fma((1.0f - u.y), (1.0f + Vh.z), -Vh.z);
    float sinTheta = zr_sqrt(zr_saturate(1.0f - z * z));
    float s, c; zr_sincos(phi, &s, &c);
    V3 Nh = v3(sinTheta * c, sinTheta * s, z) + Vh;
    return normalize(v3(ax * Nh.x, ay * Nh.y, zr_max(0.0f, Nh.z)));
}
ZR_HD V3 SampleGGXMicrofacet(V3 wo, float alpha, V3 n, V2 u)                                    // :519-538
{
    ONB onb = BuildONB(n);
    V3 woLocal = v3(dot(onb.b1, wo), dot(onb.b2, wo), dot(n, wo));
    V3 wh = SampleGGXVNDF(woLocal, alpha, alpha, u);
    return mad(wh.x, onb.b1, mad(wh.y, onb.b2, wh.z * n));
}
ZR_HD float GGXMicrofacetPdf(float alpha, float ndotwh, float ndotwo)                           // :546-554
{
    float alphaSq = alpha * alpha;
    return (GGX(ndotwh, alphaSq) * SmithG1(alphaSq, ndotwo)) / ndotwo;
}
ZR_HD float GGXMicrofacetPdf_Prepared(float alpha, float ndotwh, float ndotwo, float g1_wo)     // SmithG1(alpha^2, ndotwo) prepared (Surface::c_g1_wo)
{
    float alphaSq = alpha * alpha;
    return (GGX(ndotwh, alphaSq) * g1_wo) / ndotwo;
}

// BSDF.hlsli:560-862
struct Surface
{
    float alpha;
    V3 wo;
    float ndotwi, ndotwo, ndotwh, whdotwi, whdotwo, wodotwi, g_wo;
    V3 base;            // baseColor_Fr0_TrCol
    float eta;
    bool specTr, metallic, backfacing_wo, invalid, reflection;
    float trDepth, subsurface;      // half in the reference: always fp16-representable
    float coat_weight; V3 coat_color; float coat_alpha, coat_eta;
    // Terms of the evaluation that depend on the outgoing direction and the material only -- not on wi.  The reference's BSDF::Unified recomputes
    // them for every incident direction (BSDF.hlsli:1176-1266 is called 6 times per bounce on one ShadingData: two lobe candidates of SampleBSDF,
    // the light sample, three of BSDFSamplerPdf); PrepareWo (below) evaluates them once per surface with the very expressions of the inline code, so
    // every user gets bit-identical values.  woMask: which groups are prepared (WO_*); users of a group that is not evaluate inline (kernels that
    // never call PrepareWo fold the test).  Groups, because a prepared term is a live register while the surface is evaluated: K11, six evaluations
    // per bounce, prepares everything; the reconnection shifts, two or three evaluations per surface at 128 VGPRs, choose (ZR_PREP_SHIFT, zr_rpt.h).
    uint32_t woMask;
    float c_refl_g;      // GGXReflectance_Dielectric(rho, alpha, ndotwo, eta): the rho-LUT sample (8 texel loads + trilinear) of the gloss layer
    float c_refl_c;      // ... of the coat: GGXReflectance_Dielectric(rho, coat_alpha, ndotwo, coat_eta)
    float c_sigma, c_onA; V3 c_rho_ms;      // OrenNayar: sigma = sqrt(alpha), A = 1 / (1 + 0.2878 sigma), the multi-scatter albedo term
    float c_eta_rel, c_f0;                  // Fresnel: 1 / eta, DielectricF0(eta)
    float c_smith_wo;    // SmithG2_Opt: sqrt((ndotwo - ndotwo alpha^2) ndotwo + alpha^2), the factor of denomWo that does not depend on wi
    float c_g1_wo;       // SmithG1(alpha^2, ndotwo)

    ZR_HDM bool ThinWalled() const { return subsurface > 0; }
    ZR_HDM bool Transmissive() const { return specTr || ThinWalled(); }
    ZR_HDM bool Coated() const { return coat_weight != 0; }
    ZR_HDM bool GlossSpecular() const { return alpha <= kMaxAlphaSpecular; }
    ZR_HDM bool CoatSpecular() const { return coat_alpha <= kMaxAlphaSpecular; }
    ZR_HDM V3 TransmissionTint() const { return trDepth > 0 ? v3(1.0f) : base; }

    ZR_HDM void SetWi_Refl(V3 wi, V3 n, V3 wh)
    {
        reflection = true;
        float ndotwi_n = dot(n, wi);
        ndotwh = zr_saturate(dot(n, wh));
        whdotwo = zr_saturate(dot(wh, wo));
        whdotwi = whdotwo;
        bool isInvalid = backfacing_wo || ndotwh == 0 || whdotwo == 0;
        invalid = isInvalid || ndotwi_n <= 0;
        ndotwi = zr_max(ndotwi_n, 1e-5f);
        wodotwi = dot(wo, wi);
    }
    ZR_HDM void SetWi_Refl(V3 wi, V3 n) { SetWi_Refl(wi, n, normalize(wi + wo)); }
    ZR_HDM void SetWi_Tr(V3 wi, V3 n, V3 wh)
    {
        reflection = false;
        float ndotwi_n = dot(n, wi);
        ndotwh = zr_saturate(dot(n, wh));
        whdotwo = zr_saturate(dot(wh, wo));
        whdotwi = zr_abs(dot(wh, wi));
        bool isInvalid = backfacing_wo || (specTr && (ndotwh == 0 || whdotwo == 0));
        invalid = isInvalid || ndotwi_n >= 0 || !Transmissive() || metallic;
        ndotwi = zr_max(zr_abs(ndotwi_n), 1e-5f);
        wodotwi = dot(wo, wi);
    }
    ZR_HDM void SetWi(V3 wi, V3 n, V3 wh)
    {
        float ndotwi_n = dot(n, wi);
        reflection = ndotwi_n >= 0;
        ndotwh = zr_saturate(dot(n, wh));
        whdotwo = zr_saturate(dot(wh, wo));
        bool backfacing_r = ndotwi_n <= 0;
        bool backfacing_t = ndotwi_n >= 0 || !Transmissive() || metallic;
        bool isInvalid = backfacing_wo || (specTr && (ndotwh == 0 || whdotwo == 0));
        invalid = isInvalid || (reflection && backfacing_r) || (!reflection && backfacing_t);
        ndotwi = zr_max(zr_abs(ndotwi_n), 1e-5f);
        whdotwi = zr_abs(dot(wh, wi));
        wodotwi = dot(wo, wi);
    }
    ZR_HDM V3 SetWi(V3 wi, V3 n)
    {
        float ndotwi_n = dot(n, wi);
        reflection = ndotwi_n >= 0;
        float s = reflection ? 1 : eta;
        V3 wh = normalize(mad(s, wi, wo));
        wh = !reflection && eta > 1 ? -wh : wh;
        SetWi(wi, n, wh);
        return wh;
    }
    ZR_HDM float F0() const { return (woMask & 4u) ? c_f0 : DielectricF0(eta); }
    ZR_HDM V3 Fresnel(V3 fr0, bool* tir) const
    {
        float cosTheta_i = whdotwo;
        *tir = false;
        if (metallic) return FresnelSchlick(fr0, cosTheta_i);
        float eta_rel = (woMask & 4u) ? c_eta_rel : 1.0f / eta;
        float sinSq = zr_saturate(zr_fma(-cosTheta_i, cosTheta_i, 1.0f));
        float cosTSq = zr_fma(-eta_rel * eta_rel, sinSq, 1.0f);
        *tir = cosTSq <= 0;
        if (*tir) return v3(1.0f);
        return v3(Fresnel_Dielectric(cosTheta_i, eta_rel, zr_sqrt(cosTSq)));
    }
    ZR_HDM V3 Fresnel() const
    {
        V3 fr0 = metallic ? base : v3(F0());
        bool unused;
        return Fresnel(fr0, &unused);
    }
    ZR_HDM float Fresnel_Coat(float* cosTheta_t) const
    {
        *cosTheta_t = 0;
        float cosTheta_i = whdotwo;
        float eta_rel = 1.0f / coat_eta;
        float sinSq = zr_saturate(zr_fma(-cosTheta_i, cosTheta_i, 1.0f));
        float cosTSq = zr_fma(-eta_rel * eta_rel, sinSq, 1.0f);
        if (cosTSq <= 0) return 1;
        *cosTheta_t = zr_sqrt(cosTSq);
        float Fr0 = DielectricF0(coat_eta);
        float cosTheta = coat_eta > 1 ? cosTheta_i : *cosTheta_t;
        return FresnelSchlick_Dielectric(Fr0, cosTheta);
    }
};

// ShadingData::Init, BSDF.hlsli:584-638
// `plain`: the caller knows the scene's material class -- no metal, no specular transmission, no thin walls, no coat (SceneView::plain, set by the
// host from the material table).  The five inputs below then HAVE these values; writing them as constants lets the PLAIN permutations of the ReSTIR PT
// kernels fold every branch of those lobes away (K11: 24 283 -> 9 529 VALU instructions, 656 -> 592 B of scratch).
ZR_HD Surface InitSurface(V3 n, V3 wo, bool metallic, float roughness, V3 baseColor, float eta_curr, float eta_next,
    bool specTr, float transmissionDepth, float subsurface, float coat_weight, V3 coat_color, float coat_roughness, float eta_coat, bool plain = false)
{
    if (plain) { metallic = false; specTr = false; transmissionDepth = 0; subsurface = 0; coat_weight = 0; coat_roughness = 0; }
    if (coat_weight > 0 && coat_roughness > 0)
    {
        float rx = roughness * roughness, ry = coat_roughness * coat_roughness;
        rx *= rx; ry *= ry;
        float rc = zr_min(rx + 2 * ry, 1.0f);
        rc = zr_rsqrt(zr_rsqrt(rc));
        roughness = Lerp(roughness, rc, coat_weight);
    }
    Surface si;
    si.wo = wo;
    float ndotwo = dot(n, wo);
    si.backfacing_wo = ndotwo <= 0;
    si.ndotwo = zr_max(ndotwo, 1e-5f);
    si.metallic = metallic;
    si.alpha = roughness * roughness;
    si.base = baseColor;
    si.specTr = specTr;
    si.trDepth = zr_round_f16(transmissionDepth);
    si.subsurface = zr_round_f16(subsurface);
    float eta_base = eta_curr == kEtaAir ? eta_next : eta_curr;
    float eta_no_coat = eta_next / eta_curr;
    float eta_coated = eta_base >= eta_coat ? eta_base / eta_coat : eta_coat / eta_base;
    si.eta = Lerp(eta_no_coat, eta_coated, coat_weight);
    si.g_wo = !metallic && !specTr ? E_FON_approx(zr_max(ndotwo, 1e-4f), roughness) : 0;
    si.coat_weight = coat_weight;
    si.coat_color = coat_color;
    si.coat_alpha = coat_roughness * coat_roughness;
    si.coat_eta = eta_curr == kEtaAir ? eta_coat / kEtaAir : kEtaAir / eta_coat;
    si.ndotwi = 0; si.ndotwh = 0; si.whdotwi = 0; si.whdotwo = 0; si.wodotwi = 0; si.invalid = true; si.reflection = true;
    si.woMask = 0; si.c_refl_g = 0; si.c_refl_c = 0; si.c_sigma = 0; si.c_onA = 0; si.c_rho_ms = v3(0.0f); si.c_eta_rel = 0; si.c_f0 = 0; si.c_smith_wo = 0; si.c_g1_wo = 0;
    return si;
}

// The wo-only terms of a surface (see Surface): each one is the expression its inline user evaluates, on the same operands, in the same order.
// Only the terms a later evaluation can reach are computed (a metal never reads the rho LUT, a specular gloss layer has no Smith terms, ...);
// the others stay 0 and are never read, because the users test the same material flags.
ZR_HD float OrenNayarA(float sigma) { return 1.0f / zr_fma(0.287793398f, sigma, 1.0f); }
ZR_HD V3 OrenNayarRhoMs(V3 rho, float A, float B)
{
    float avgR = zr_fma(0.0724882111f, B, A);
    float one_min = 1 - avgR;
    float tmp = ZR_ONE_OVER_PI * (avgR / one_min);
    V3 rho_ms = v3(tmp / zr_fma(-rho.x, one_min, 1.0f), tmp / zr_fma(-rho.y, one_min, 1.0f), tmp / zr_fma(-rho.z, one_min, 1.0f));
    return rho_ms * rho;
}
ZR_HD float SmithWoTerm(float alphaSq, float ndotwo) { return zr_sqrt(zr_fma(zr_fma(-ndotwo, alphaSq, ndotwo), ndotwo, alphaSq)); }
enum : uint32_t { WO_LUT = 1u, WO_DIFFUSE = 2u, WO_FRESNEL = 4u, WO_SMITH = 8u, WO_ALL = 15u };
ZR_HD void PrepareWo(const RhoView& rho, Surface& s, uint32_t groups = WO_ALL)
{
    const float alphaSq = s.alpha * s.alpha;
    if (!s.metallic)
    {
        if (groups & WO_FRESNEL)
        {
            s.c_eta_rel = 1.0f / s.eta;
            s.c_f0 = DielectricF0(s.eta);
        }
        if ((groups & WO_LUT) && !s.GlossSpecular()) s.c_refl_g = GGXReflectance_Dielectric(rho, s.alpha, s.ndotwo, s.eta);
        if ((groups & WO_DIFFUSE) && !s.specTr)      // the diffuse slab: EvalDiffuse is only reached without specular transmission
        {
            s.c_sigma = zr_sqrt(s.alpha);
            if (s.c_sigma != 0)
            {
                s.c_onA = OrenNayarA(s.c_sigma);
                s.c_rho_ms = OrenNayarRhoMs(s.base, s.c_onA, s.c_sigma * s.c_onA);
            }
        }
    }
    if ((groups & WO_LUT) && s.Coated()) s.c_refl_c = GGXReflectance_Dielectric(rho, s.coat_alpha, s.ndotwo, s.coat_eta);
    if ((groups & WO_SMITH) && !s.GlossSpecular())
    {
        s.c_smith_wo = SmithWoTerm(alphaSq, s.ndotwo);
        s.c_g1_wo = SmithG1(alphaSq, s.ndotwo);
    }
    s.woMask = groups;
}
// the two LUT reads of a prepared / unprepared surface
ZR_HD float ReflG(const RhoView& rho, const Surface& s) { return (s.woMask & WO_LUT) ? s.c_refl_g : GGXReflectance_Dielectric(rho, s.alpha, s.ndotwo, s.eta); }
ZR_HD float ReflC(const RhoView& rho, const Surface& s) { return (s.woMask & WO_LUT) ? s.c_refl_c : GGXReflectance_Dielectric(rho, s.coat_alpha, s.ndotwo, s.coat_eta); }

// ---- slabs, BSDF.hlsli:904-1152 ----
ZR_HD V3 EvalDiffuse(bool eon, const Surface& s)
{
    float k = s.subsurface == 0 ? 1 : s.subsurface * 0.5f;
    V3 d = (s.woMask & 2u) ? OrenNayarPrepared(eon, s.base, s.c_sigma, s.c_onA, s.c_rho_ms, s.ndotwo, s.ndotwi, s.wodotwi, s.g_wo)
                     : OrenNayar(eon, s.base, zr_sqrt(s.alpha), s.ndotwo, s.ndotwi, s.wodotwi, s.g_wo);
    return k * d;
}
ZR_HD V3 SampleDiffuse(V3 n, V2 u, float* pdf)
{
    V3 l = SampleCosineWeightedHemisphere(u, pdf);
    ONB onb = BuildONB(n);
    return mad(l.x, onb.b1, mad(l.y, onb.b2, l.z * n));
}
ZR_HD float DiffusePdf(const Surface& s) { return s.ndotwi * ZR_ONE_OVER_PI; }
ZR_HD V3 EvalGloss(const Surface& s, V3 fr)
{
    return (s.woMask & 8u) ? GGXMicrofacetBRDF_Prepared(s.alpha, s.ndotwh, s.ndotwo, s.ndotwi, fr, s.GlossSpecular(), s.c_smith_wo)
                     : GGXMicrofacetBRDF(s.alpha, s.ndotwh, s.ndotwo, s.ndotwi, fr, s.GlossSpecular());
}
// GGXMicrofacetPdf of the gloss layer's half vector
ZR_HD float GlossWhPdf(const Surface& s)
{ return (s.woMask & 8u) && !s.GlossSpecular() ? GGXMicrofacetPdf_Prepared(s.alpha, s.ndotwh, s.ndotwo, s.c_g1_wo) : GGXMicrofacetPdf(s.alpha, s.ndotwh, s.ndotwo); }
ZR_HD V3 SampleGloss(const Surface& s, V3 n, V2 u)
{
    if (s.GlossSpecular()) return reflect(-s.wo, n);
    return reflect(-s.wo, SampleGGXMicrofacet(s.wo, s.alpha, n, u));
}
ZR_HD float GlossPdf(const Surface& s)
{
    if (s.GlossSpecular()) return (s.ndotwh >= kMinNdotHSpecular) ? 1.0f : 0.0f;
    return GlossWhPdf(s) / 4.0f;
}
ZR_HD float EvalTranslucentTr(const Surface& s, float fr)
{
    return (s.woMask & 8u) ? GGXMicrofacetBTDF_Prepared(s.alpha, s.ndotwh, s.ndotwo, s.ndotwi, s.whdotwo, s.whdotwi, s.eta, fr, s.GlossSpecular(), s.c_smith_wo)
                     : GGXMicrofacetBTDF(s.alpha, s.ndotwh, s.ndotwo, s.ndotwi, s.whdotwo, s.whdotwi, s.eta, fr, s.GlossSpecular());
}
ZR_HD float EvalCoat(const Surface& s, float Fr)
{ return s.coat_weight * GGXMicrofacetBRDF(s.coat_alpha, s.ndotwh, s.ndotwo, s.ndotwi, v3(Fr), s.CoatSpecular()).x; }
ZR_HD V3 SampleCoat(const Surface& s, V3 n, V2 u)
{
    V3 wh = s.CoatSpecular() ? n : SampleGGXMicrofacet(s.wo, s.coat_alpha, n, u);
    return reflect(-s.wo, wh);
}
ZR_HD float CoatPdf(const Surface& s)
{
    if (s.CoatSpecular()) return (s.ndotwh >= kMinNdotHSpecular) ? 1.0f : 0.0f;
    return GGXMicrofacetPdf(s.coat_alpha, s.ndotwh, s.ndotwo) / 4.0f;
}
ZR_HD V3 BaseWeight(const RhoView& rho, const Surface& s)
{
    V3 bw = v3(1.0f);
    if (s.Coated())
    {
        float cosT;
        float Fr_coat = s.Fresnel_Coat(&cosT);
        if (cosT <= 0) return v3(0.0f);
        float refl_c = s.CoatSpecular() ? Fr_coat : ReflC(rho, s);
        float c = 0.5f / cosT + 0.5f / s.whdotwo;
        V3 coat_tr = vexp(c * vlog(s.coat_color));
        bw = Lerp(v3(1.0f), (1 - refl_c) * coat_tr, s.coat_weight);
    }
    return bw;
}
ZR_HD V3 TransmittanceToDielectricBaseTr(const RhoView& rho, const Surface& s)
{
    V3 bw = BaseWeight(rho, s);
    float refl_g = s.GlossSpecular() ? 0 : ReflG(rho, s);
    return (1 - refl_g) * bw;
}
ZR_HD V3 DielectricBaseSpecularTr(const RhoView& rho, const Surface& s, float Fr_g)
{
    if (s.invalid || !s.specTr) return v3(0.0f);
    V3 tr = TransmittanceToDielectricBaseTr(rho, s);
    float glossyTr = EvalTranslucentTr(s, Fr_g);
    return glossyTr * s.TransmissionTint() * tr;
}
ZR_HD V3 DielectricBaseDiffuseTr(const RhoView& rho, const Surface& s, float Fr_g)
{
    if (s.invalid) return v3(0.0f);
    V3 bw = BaseWeight(rho, s);
    float refl_g = s.GlossSpecular() ? Fr_g : ReflG(rho, s);
    return (1 - refl_g) * EvalDiffuse(false, s) * bw;
}

struct Eval { V3 f; V3 Fr_g; bool tir; };

// BSDF::Unified, BSDF.hlsli:1176-1266
ZR_HD Eval Unified(const RhoView& rho, const Surface& s)
{
    Eval ret; ret.f = v3(0.0f); ret.Fr_g = v3(0.0f); ret.tir = false;
    if (s.invalid) return ret;
    V3 bw = v3(1.0f);
    if (s.Coated())
    {
        float cosT;
        float Fr_coat = s.Fresnel_Coat(&cosT);
        bool tir_c = cosT <= 0;
        if (!s.reflection && tir_c) return ret;
        if (s.reflection)
        {
            ret.f = v3(EvalCoat(s, Fr_coat));
            if (tir_c) return ret;
        }
        float refl_c = s.CoatSpecular() ? Fr_coat : ReflC(rho, s);
        float c = 1.0f / cosT;
        V3 coat_tr = vexp(c * vlog(s.coat_color));
        bw = Lerp(v3(1.0f), (1 - refl_c) * coat_tr, s.coat_weight);
    }
    V3 fr0 = s.metallic ? s.base : v3(s.F0());
    ret.Fr_g = s.Fresnel(fr0, &ret.tir);
    V3 glossyRefl = EvalGloss(s, ret.Fr_g);
    if (s.metallic || ret.tir) { ret.f = ret.f + bw * glossyRefl; return ret; }
    float refl_g = s.GlossSpecular() ? ret.Fr_g.x : ReflG(rho, s);
    if (!s.specTr)
    {
        V3 diffuse = EvalDiffuse(true, s);
        ret.f = ret.f + bw * ((1 - refl_g) * diffuse + glossyRefl * (s.reflection ? 1.0f : 0.0f));
        return ret;
    }
    if (s.reflection) { ret.f = ret.f + glossyRefl * bw; return ret; }
    refl_g = s.GlossSpecular() ? 0 : refl_g;
    float glossyTr = EvalTranslucentTr(s, ret.Fr_g.x);
    ret.f = ((1 - refl_g) * glossyTr * s.TransmissionTint()) * bw;
    return ret;
}

// ---- BSDFSampling.hlsli ----
struct BsdfSample { V3 wi; uint32_t lobe; float pdf; V3 bsdfOverPdf; V3 f; };
ZR_HD BsdfSample InitBsdfSample()
{ BsdfSample r; r.wi = v3(0.0f); r.lobe = LOBE_ALL; r.pdf = 0; r.bsdfOverPdf = v3(0.0f); r.f = v3(0.0f); return r; }

// target functor of the lobe RIS (BSDFSampling.hlsli:14-22 NoOp; NEE.hlsli:68-84 SkyIncidentRadiance is the other one)
struct NoOpTarget { ZR_HDM V3 operator()(V3) const { return v3(1.0f); } };

// SampleBSDF_NoDiffuse, BSDFSampling.hlsli:59-152
template<typename Func>
ZR_HD BsdfSample SampleBSDF_NoDiffuse(const RhoView& rho, V3 n, Surface s, V2 u_c, V2 u_g, float u_wrs_0, float u_wrs_1, Func func)
{
    BsdfSample ret = InitBsdfSample();
    float pdf_base = 1;
    if (s.Coated())
    {
        float refl_c = ReflC(rho, s);
        float pdf_coat = refl_c * s.coat_weight;
        pdf_base = 1 - pdf_coat;
        if (u_wrs_0 < pdf_coat)
        {
            V3 wi_c = SampleCoat(s, n, u_c);
            s.SetWi_Refl(wi_c, n);
            Eval e = Unified(rho, s);
            ret.wi = wi_c; ret.lobe = LOBE_COAT; ret.f = e.f * func(wi_c);
            ret.pdf = CoatPdf(s) * pdf_coat;
            ret.bsdfOverPdf = ret.f / ret.pdf;
            return ret;
        }
    }
    V3 wh = s.GlossSpecular() ? n : SampleGGXMicrofacet(s.wo, s.alpha, n, u_g);
    V3 wi_r = reflect(-s.wo, wh);
    s.SetWi_Refl(wi_r, n, wh);
    float wh_pdf = GlossWhPdf(s);
    ret.wi = wi_r; ret.lobe = LOBE_GLOSSY_R;
    ret.pdf = s.GlossSpecular() ? 1 : wh_pdf / 4.0f;
    ret.pdf *= pdf_base;
    Eval e = Unified(rho, s);
    const V3 func_r = func(wi_r);
    ret.f = e.f * func_r;
    ret.bsdfOverPdf = ret.f / ret.pdf;
    if (s.metallic || !s.specTr || e.tir) return ret;

    V3 wi_t = refract(-s.wo, wh, 1 / s.eta);
    const V3 func_t = func(wi_t);
    float p_r = e.Fr_g.x * Luminance(func_r);
    p_r = p_r / (p_r + (1 - e.Fr_g.x) * Luminance(func_t));
    if (u_wrs_1 < p_r)
    {
        ret.bsdfOverPdf = ret.bsdfOverPdf / p_r;
        ret.pdf *= p_r;
    }
    else
    {
        s.SetWi_Tr(wi_t, n, wh);
        ret.pdf = (1 - p_r) * pdf_base;
        if (!s.GlossSpecular())
        {
            ret.pdf *= wh_pdf * s.whdotwo;
            ret.pdf *= JacobianHalfVecToIncident_Tr(s.eta, s.whdotwo, s.whdotwi);
        }
        ret.f = DielectricBaseSpecularTr(rho, s, e.Fr_g.x) * func_t;
        ret.bsdfOverPdf = ret.pdf > 0 ? ret.f / ret.pdf : v3(0.0f);
        ret.wi = wi_t; ret.lobe = LOBE_GLOSSY_T;
    }
    return ret;
}

// SampleBSDF_NoSpecTr, BSDFSampling.hlsli:181-296
template<typename Func>
ZR_HD BsdfSample SampleBSDF_NoSpecTr(const RhoView& rho, V3 n, Surface s, V2 u_coat, V2 u_g, V2 u_d, float u_wrs_g,
    float u_wrs_dr, float u_wrs_dt, Func func)
{
    BsdfSample ret = InitBsdfSample();
    float w_sum = 0;
    V3 target = v3(0.0f);
    if (s.Coated())
    {
        V3 wi_c = SampleCoat(s, n, u_coat);
        s.SetWi_Refl(wi_c, n);
        Eval e = Unified(rho, s);
        target = e.f * func(wi_c);
        ret.wi = wi_c; ret.lobe = LOBE_COAT; ret.f = target;
        float pdf_c = CoatPdf(s), pdf_g = GlossPdf(s);
        float pdf_d = !s.metallic ? DiffusePdf(s) : 0;
        w_sum = BalanceHeuristic3(pdf_c, pdf_g, pdf_d, Luminance(target));
    }
    {
        V3 wi_g = SampleGloss(s, n, u_g);
        s.SetWi_Refl(wi_g, n);
        Eval e = Unified(rho, s);
        V3 target_g = e.f * func(wi_g);
        float pdf_g = GlossPdf(s);
        float pdf_d = !s.metallic && !e.tir ? DiffusePdf(s) : 0;
        float pdf_c = s.Coated() ? CoatPdf(s) : 0;
        float w_g = BalanceHeuristic3(pdf_g, pdf_d, pdf_c, Luminance(target_g));
        w_sum += w_g;
        if ((w_sum > 0) && (u_wrs_g < (w_g / w_sum))) { target = target_g; ret.wi = wi_g; ret.lobe = LOBE_GLOSSY_R; ret.f = target_g; }
    }
    if (!s.metallic)
    {
        float pdf_d;
        V3 wi_d = SampleDiffuse(n, u_d, &pdf_d);
        float Fr_g;
        {
            s.SetWi_Refl(wi_d, n);
            Eval e = Unified(rho, s);
            V3 target_dr = e.f * func(wi_d);
            Fr_g = e.Fr_g.x;
            float pdf_g = GlossPdf(s);
            float pdf_c = s.Coated() ? CoatPdf(s) : 0;
            float w_dr = BalanceHeuristic3(pdf_d, pdf_g, pdf_c, Luminance(target_dr));
            w_sum += w_dr;
            if ((w_sum > 0) && (u_wrs_dr < (w_dr / w_sum))) { target = target_dr; ret.wi = wi_d; ret.lobe = LOBE_DIFFUSE_R; ret.f = target_dr; }
        }
        if (s.ThinWalled())
        {
            V3 wi_dt = -wi_d;
            V3 target_dt = DielectricBaseDiffuseTr(rho, s, Fr_g);
            target_dt = target_dt * func(wi_dt);
            float w_dt = Luminance(target_dt) / pdf_d;
            w_sum += w_dt;
            if ((w_sum > 0) && (u_wrs_dt < (w_dt / w_sum))) { target = target_dt; ret.wi = wi_dt; ret.lobe = LOBE_DIFFUSE_T; ret.f = target_dt; }
        }
    }
    float targetLum = Luminance(target);
    ret.bsdfOverPdf = targetLum > 0 ? target * w_sum / targetLum : v3(0.0f);
    ret.pdf = w_sum > 0 ? targetLum / w_sum : 0;
    return ret;
}

// SampleBSDF, BSDFSampling.hlsli:318-338: always consumes 9 uniforms so the RNG stream stays aligned for replay
template<typename Func>
ZR_HD BsdfSample SampleBSDF(const RhoView& rho, V3 n, const Surface& s, Func func, Rng& rng)
{
    V2 u_c = rng.Uniform2D();
    V2 u_g = rng.Uniform2D();
    V2 u_d = rng.Uniform2D();
    float u0 = rng.Uniform(), u1 = rng.Uniform(), u2 = rng.Uniform();
    if (!s.specTr) return SampleBSDF_NoSpecTr(rho, n, s, u_c, u_g, u_d, u0, u1, u2, func);
    return SampleBSDF_NoDiffuse(rho, n, s, u_c, u_g, u0, u1, func);
}
ZR_HD BsdfSample SampleBSDF(const RhoView& rho, V3 n, const Surface& s, Rng& rng) { return SampleBSDF(rho, n, s, NoOpTarget(), rng); }
ZR_HD BsdfSample SampleBSDF_NoDiffuse(const RhoView& rho, V3 n, const Surface& s, V2 u_c, V2 u_g, float u_wrs_0, float u_wrs_1)
{ return SampleBSDF_NoDiffuse(rho, n, s, u_c, u_g, u_wrs_0, u_wrs_1, NoOpTarget()); }

// BSDFSamplerPdf_NoDiffuse, BSDFSampling.hlsli:565-631
template<typename Func>
ZR_HD float BSDFSamplerPdf_NoDiffuse(const RhoView& rho, V3 n, Surface s, V3 wi, Func func)
{
    V3 wh = s.SetWi(wi, n);
    float pdf_base = 1, pdf_c = 0;
    if (s.Coated())
    {
        float refl_c = ReflC(rho, s);
        float pdf_coat = refl_c * s.coat_weight;
        pdf_base = 1 - pdf_coat;
        if (s.reflection) pdf_c = CoatPdf(s) * pdf_coat;
    }
    const float wh_pdf = GlossWhPdf(s);
    if (s.metallic || !s.specTr)
    {
        float pdf_gr = s.GlossSpecular() ? (s.ndotwh >= kMinNdotHSpecular ? 1.0f : 0.0f) : wh_pdf / 4.0f;
        pdf_gr *= pdf_base;
        return s.reflection ? pdf_c + pdf_gr : 0;
    }
    float pdf_g = s.GlossSpecular() ? (s.ndotwh >= kMinNdotHSpecular ? 1.0f : 0.0f) : 1;
    pdf_g *= pdf_base;
    const V3 wi_other = !s.reflection ? reflect(-s.wo, wh) : refract(-s.wo, wh, 1 / s.eta);
    float lumA = Luminance(func(wi)), lumB = Luminance(func(wi_other));
    float Fr_g = s.Fresnel().x;
    float pdf_r = Fr_g * (s.reflection ? lumA : lumB);
    pdf_r = pdf_r / (pdf_r + (1 - Fr_g) * (s.reflection ? lumB : lumA));
    if (s.reflection)
    {
        pdf_g *= s.GlossSpecular() ? 1 : (wh_pdf / 4.0f);
        pdf_g *= pdf_r;
        return pdf_g + pdf_c;
    }
    pdf_g *= 1 - pdf_r;
    if (!s.GlossSpecular())
    {
        pdf_g *= wh_pdf * s.whdotwo;
        pdf_g *= JacobianHalfVecToIncident_Tr(s.eta, s.whdotwo, s.whdotwi);
    }
    return pdf_g;
}

// BSDFSamplerPdf, BSDFSampling.hlsli:639-759
// `haveZ`: the caller has already applied s.SetWi(wi_z, n) to `s` and evaluated f_z = Unified(rho, s).f -- every NEE function of the reference
// evaluates the BSDF towards the light and then asks this function for the sampler's pdf of the same direction, which evaluates it again
// (ReSTIR_PT_NEE.hlsli:262-275, 318-336); SetWi and Unified are pure functions of (s, wi_z, n), so reusing the caller's values is bit-identical.
template<typename Func>
ZR_HD float BSDFSamplerPdf(const RhoView& rho, V3 n, Surface s, V3 wi_z, Func func, Rng& rng, bool haveZ = false, V3 f_z = v3(0.0f))
{
    if (s.specTr) return BSDFSamplerPdf_NoDiffuse(rho, n, s, wi_z, func);
    if (!haveZ) s.SetWi(wi_z, n);
    if (!s.reflection && !s.ThinWalled()) return 0;
    if (!haveZ) f_z = Unified(rho, s).f;
    float targetLum = Luminance(f_z * func(wi_z));
    if (targetLum == 0) return 0;

    float w_sum_c, w_sum_g, w_sum_dr, w_sum_dt;
    {
        float pdf_g = GlossPdf(s);
        float pdf_d = !s.metallic ? DiffusePdf(s) : 0;
        float pdf_c = s.Coated() ? CoatPdf(s) : 0;
        float w = s.reflection ? BalanceHeuristic3(pdf_g, pdf_d, pdf_c, targetLum) : (targetLum / pdf_d) * (!s.metallic ? 1.0f : 0.0f);
        w_sum_g = w; w_sum_dr = w; w_sum_dt = w; w_sum_c = w;
    }
    if (w_sum_g == 0) return 0;

    float pdf_d;
    V3 wi_d = SampleDiffuse(n, rng.Uniform2D(), &pdf_d);
    float Fr_g = 0;
    if (!s.metallic)
    {
        s.SetWi_Refl(wi_d, n);
        Eval e = Unified(rho, s);
        Fr_g = e.Fr_g.x;
        float lum = Luminance(e.f * func(wi_d));
        float pdf_g = GlossPdf(s);
        float pdf_c = s.Coated() ? CoatPdf(s) : 0;
        float w = BalanceHeuristic3(pdf_d, pdf_g, pdf_c, lum);
        w_sum_g += w; w_sum_dt += w; w_sum_c += w;
    }
    if (!s.metallic && s.ThinWalled())
    {
        V3 target_dt = DielectricBaseDiffuseTr(rho, s, Fr_g);
        float w = Luminance(target_dt * func(-wi_d)) / pdf_d;
        w_sum_g += w; w_sum_dr += w; w_sum_c += w;
    }
    {
        V3 wi_g = SampleGloss(s, n, rng.Uniform2D());
        s.SetWi_Refl(wi_g, n);
        V3 target_g = Unified(rho, s).f;
        float lum = Luminance(target_g * func(wi_g));
        float pdf_g = GlossPdf(s);
        float pdf_dd = !s.metallic ? DiffusePdf(s) : 0;
        float pdf_c = s.Coated() ? CoatPdf(s) : 0;
        float w = BalanceHeuristic3(pdf_g, pdf_dd, pdf_c, lum);
        w_sum_dr += w; w_sum_dt += w; w_sum_c += w;
    }
    if (s.Coated())
    {
        V3 wi_c = SampleCoat(s, n, rng.Uniform2D());
        s.SetWi_Refl(wi_c, n);
        V3 target_c = Unified(rho, s).f;
        float lum = Luminance(target_c * func(wi_c));
        float pdf_g = GlossPdf(s);
        float pdf_dd = !s.metallic ? DiffusePdf(s) : 0;
        float pdf_c = CoatPdf(s);
        float w = BalanceHeuristic3(pdf_g, pdf_dd, pdf_c, lum);
        w_sum_g += w; w_sum_dr += w; w_sum_dt += w;
    }
    float pdf = w_sum_g > 0 ? targetLum / w_sum_g : 0;
    pdf += w_sum_dr > 0 ? targetLum / w_sum_dr : 0;
    pdf += w_sum_c > 0 ? targetLum / w_sum_c : 0;
    pdf += s.ThinWalled() && (w_sum_dt > 0) ? targetLum / w_sum_dt : 0;
    return pdf;
}
ZR_HD float BSDFSamplerPdf_NoDiffuse(const RhoView& rho, V3 n, const Surface& s, V3 wi) { return BSDFSamplerPdf_NoDiffuse(rho, n, s, wi, NoOpTarget()); }
ZR_HD float BSDFSamplerPdf(const RhoView& rho, V3 n, const Surface& s, V3 wi_z, Rng& rng) { return BSDFSamplerPdf(rho, n, s, wi_z, NoOpTarget(), rng); }
// s: SetWi(wi_z, n) applied; f_z = Unified(rho, s).f
ZR_HD float BSDFSamplerPdf_AtZ(const RhoView& rho, V3 n, const Surface& s, V3 wi_z, V3 f_z, Rng& rng) { return BSDFSamplerPdf(rho, n, s, wi_z, NoOpTarget(), rng, true, f_z); }

} // namespace zr
