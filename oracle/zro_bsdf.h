// ORACLE -- test infrastructure only (see zro_math.h header).
//
// zro_bsdf.h: CPU restatement of Source/ZetaRenderPass/Common/BSDF.hlsli and BSDFSampling.hlsli
// (OpenPBR-style layered BSDF: coat / metal or dielectric gloss / EON diffuse / diffuse + specular transmission,
// and aggregate-BSDF sampling by streaming RIS over lobes).  file:line cited per function.
#pragma once
#include "zro_math.h"

namespace zro {

// Material.h:5-17
static const float MIN_METALNESS_METAL = 0.9f;
static const float MIN_IOR = 1.0f;
static const float MAX_IOR = 2.5f;
static const float DEFAULT_ETA_MAT = 1.5f;
static const float DEFAULT_ETA_COAT = 1.6f;
static const float ETA_AIR = 1.0f;

// rho.dds: 3-D LUT R16_UNORM sampled with a linear-clamp sampler (BSDF.hlsli:279-296).  The ABI pins the filter to
// fp32 trilinear interpolation with texel centres at (i + 0.5) / N and clamp addressing.
struct RhoLUT { const uint16_t* data; uint32_t dim[3]; };

static inline float SampleRho(const RhoLUT& lut, float u, float v, float w)
{
    const float c[3] = {u, v, w};
    int i0[3], i1[3]; float fr[3];
    for (int a = 0; a < 3; a++)
    {
        float x = c[a] * (float)lut.dim[a] - 0.5f;
        float fl = zr_floor(x);
        fr[a] = x - fl;
        int i = (int)fl;
        int hi = (int)lut.dim[a] - 1;
        i0[a] = i < 0 ? 0 : (i > hi ? hi : i);
        i1[a] = (i + 1) < 0 ? 0 : ((i + 1) > hi ? hi : (i + 1));
    }
    auto T = [&](int x, int y, int z) {
        return (float)lut.data[((size_t)z * lut.dim[1] + y) * lut.dim[0] + x] / 65535.0f; };
    float c00 = zr_lerp(T(i0[0], i0[1], i0[2]), T(i1[0], i0[1], i0[2]), fr[0]);
    float c10 = zr_lerp(T(i0[0], i1[1], i0[2]), T(i1[0], i1[1], i0[2]), fr[0]);
    float c01 = zr_lerp(T(i0[0], i0[1], i1[2]), T(i1[0], i0[1], i1[2]), fr[0]);
    float c11 = zr_lerp(T(i0[0], i1[1], i1[2]), T(i1[0], i1[1], i1[2]), fr[0]);
    float c0 = zr_lerp(c00, c10, fr[1]);
    float c1 = zr_lerp(c01, c11, fr[1]);
    return zr_lerp(c0, c1, fr[2]);
}

namespace BSDF {

static const float MIN_N_DOT_H_SPECULAR = 0.99998f;
static const float MAX_ALPHA_SPECULAR = 0.0016f;

enum class LOBE : uint16_t { DIFFUSE_R = 0, DIFFUSE_T = 1, GLOSSY_R = 2, GLOSSY_T = 3, COAT = 4, ALL = 5 };

// the rho LUT the oracle evaluates against (set by the scene)
static thread_local const RhoLUT* g_rho = nullptr;

// BSDF.hlsli:113-117
static inline float DielectricF0(float eta) { float f0 = (eta - 1) / (eta + 1); return f0 * f0; }
// BSDF.hlsli:122-135
static inline float3 FresnelSchlick(float3 F0, float whdotwx)
{
    float tmp = 1.0f - whdotwx;
    float tmpSq = tmp * tmp;
    float k = tmpSq * tmpSq * tmp;
    return f3(zr_fma(k, 1 - F0.x, F0.x), zr_fma(k, 1 - F0.y, F0.y), zr_fma(k, 1 - F0.z, F0.z));
}
static inline float FresnelSchlick_Dielectric(float F0, float whdotwx)
{
    float tmp = 1.0f - whdotwx;
    float tmpSq = tmp * tmp;
    return zr_fma(tmpSq * tmpSq * tmp, 1 - F0, F0);
}
// BSDF.hlsli:156-163
static inline float Fresnel_Dielectric(float ndotwi, float eta, float cosTheta_t)
{
    float r_parallel = zr_fma(-eta, cosTheta_t, ndotwi) / zr_fma(eta, cosTheta_t, ndotwi);
    float r_perp = zr_fma(eta, ndotwi, -cosTheta_t) / zr_fma(eta, ndotwi, cosTheta_t);
    return 0.5f * (r_parallel * r_parallel + r_perp * r_perp);
}
// BSDF.hlsli:169-173
static inline float GGX(float ndotwh, float alphaSq)
{
    float denom = zr_fma(ndotwh * ndotwh, alphaSq - 1.0f, 1.0f);
    return alphaSq / (ZR_PI * denom * denom);
}
// BSDF.hlsli:185-190
static inline float SmithG1(float alphaSq, float ndotx)
{
    float ndotxSq = ndotx * ndotx;
    float tanThetaSq = (1.0f - ndotxSq) / ndotxSq;
    return 2.0f / (zr_sqrt(zr_fma(alphaSq, tanThetaSq, 1.0f)) + 1.0f);
}
// BSDF.hlsli:209-216
static inline float SmithHeightCorrelatedG2_Opt(int n, float alphaSq, float ndotwi, float ndotwo)
{
    float denomWo = ndotwi * zr_sqrt(zr_fma(zr_fma(-ndotwo, alphaSq, ndotwo), ndotwo, alphaSq));
    float denomWi = ndotwo * zr_sqrt(zr_fma(zr_fma(-ndotwi, alphaSq, ndotwi), ndotwi, alphaSq));
    return (0.5f * (float)n) / (denomWo + denomWi);
}
// BSDF.hlsli:220-226
static inline float SmithHeightCorrelatedG2OverG1(float alphaSq, float ndotwi, float ndotwo)
{
    float G1wi = SmithG1(alphaSq, ndotwi);
    float G1wo = SmithG1(alphaSq, ndotwo);
    return G1wi / (G1wi + G1wo - G1wi * G1wo);
}
// BSDF.hlsli:279-296
static inline float GGXReflectance_Dielectric(float alpha, float ndotwo, float eta)
{
    float u = ndotwo;
    float v = ((alpha - 0.002025f) / (1.0f - 0.002025f));
    float w = ((eta - 0.5f) / (1.99f - 0.5f));
    float rho = SampleRho(*g_rho, u, v, w);
    return zr_saturate(rho);
}
// BSDF.hlsli:335-345
static inline float E_FON_approx(float cosTheta, float roughness)
{
    float mucomp = 1.0f - cosTheta;
    float mucomp2 = mucomp * mucomp;
    // mul(float2x2(0.0571085289, 0.491881867, -0.332181442, 0.0714429953), float2(mucomp, mucomp2))
    float2 q = {0.0571085289f * mucomp + 0.491881867f * mucomp2, -0.332181442f * mucomp + 0.0714429953f * mucomp2};
    float GoverPi = q.x * 1.0f + q.y * mucomp2;
    return zr_fma(roughness, GoverPi, 1.0f) / zr_fma(0.287793398f, roughness, 1.0f);
}
// BSDF.hlsli:354-389 (APPROXIMATE_EON_MULTISCATTER == 1)
static inline float3 OrenNayar(bool AccountForMultiScattering, float3 rho, float sigma, float ndotwo, float ndotwi,
    float wodotwi, float g_wo)
{
    if (sigma == 0) return ZR_ONE_OVER_PI * ndotwi * rho;
    float A = 1.0f / zr_fma(0.287793398f, sigma, 1.0f);
    float B = sigma * A;
    float s_over_t = zr_fma(-ndotwi, ndotwo, wodotwi);
    s_over_t = s_over_t > 0 ? s_over_t / zr_max(ndotwi, ndotwo) : s_over_t;
    float3 f = f3(ZR_ONE_OVER_PI * zr_fma(B, s_over_t, A));
    float3 f_comp = f3(0.0f);
    if (AccountForMultiScattering)
    {
        float avgReflectance = zr_fma(0.0724882111f, B, A);
        float one_min_avgReflectance = 1 - avgReflectance;
        float tmp = ZR_ONE_OVER_PI * (avgReflectance / one_min_avgReflectance);
        float3 rho_ms_over_piSq = f3(tmp / zr_fma(-rho.x, one_min_avgReflectance, 1.0f),
                                     tmp / zr_fma(-rho.y, one_min_avgReflectance, 1.0f),
                                     tmp / zr_fma(-rho.z, one_min_avgReflectance, 1.0f));
        rho_ms_over_piSq *= rho;
        float E_wo = g_wo;
        float E_wi = E_FON_approx(ndotwi, sigma);
        f_comp = (1 - E_wo) * (1 - E_wi) * rho_ms_over_piSq;
    }
    return ndotwi * (f + f_comp) * rho;
}
// BSDF.hlsli:392-413
static inline float3 GGXMicrofacetBRDF(float alpha, float ndotwh, float ndotwo, float ndotwi, float3 fr, bool specular)
{
    if (specular) return (ndotwh >= MIN_N_DOT_H_SPECULAR ? 1.0f : 0.0f) * fr;
    float alphaSq = alpha * alpha;
    float NDF = GGX(ndotwh, alphaSq);
    float G2DivDenom = SmithHeightCorrelatedG2_Opt(1, alphaSq, ndotwi, ndotwo);
    float f = NDF * G2DivDenom * ndotwi;
    return f * fr;
}
// BSDF.hlsli:420-427
static inline float JacobianHalfVecToIncident_Tr(float eta, float whdotwo, float whdotwi)
{
    float denom = zr_fma(whdotwo, 1 / eta, whdotwi);
    denom *= denom;
    return denom > 0 ? whdotwi / denom : 0;
}
// BSDF.hlsli:430-458
static inline float GGXMicrofacetBTDF(float alpha, float ndotwh, float ndotwo, float ndotwi, float whdotwo,
    float whdotwi, float eta, float fr, bool specular)
{
    if (specular) { float f = ndotwh >= MIN_N_DOT_H_SPECULAR ? 1.0f : 0.0f; return f * (1 - fr); }
    float alphaSq = alpha * alpha;
    float NDF = GGX(ndotwh, alphaSq);
    float G2opt = SmithHeightCorrelatedG2_Opt(4, alphaSq, ndotwi, ndotwo);
    float f = NDF * G2opt * whdotwo;
    float dwh_dwi = JacobianHalfVecToIncident_Tr(eta, whdotwo, whdotwi);
    f *= dwh_dwi;
    f *= ndotwi;
    return f * (1 - fr);
}
// BSDF.hlsli:464-484 (Dupuy-Benyoub spherical caps)
static inline float3 SampleGGXVNDF(float3 wo, float alpha_x, float alpha_y, float2 u)
{
    float3 Vh = normalize(f3(alpha_x * wo.x, alpha_y * wo.y, wo.z));
    float phi = ZR_TWO_PI * u.x;
    float z = zr_fma((1.0f - u.y), (1.0f + Vh.z), -Vh.z);
    float sinTheta = zr_sqrt(zr_saturate(1.0f - z * z));
    float s, c; zr_sincos(phi, &s, &c);
    float x = sinTheta * c;
    float y = sinTheta * s;
    float3 Nh = f3(x, y, z) + Vh;
    return normalize(f3(alpha_x * Nh.x, alpha_y * Nh.y, zr_max(0.0f, Nh.z)));
}
// BSDF.hlsli:519-538 (USE_ISOTROPIC_VNDF == 0)
static inline float3 SampleGGXMicrofacet(float3 wo, float alpha, float3 shadingNormal, float2 u)
{
    Math::CoordinateSystem onb = Math::CoordinateSystem::Build(shadingNormal);
    float3 woLocal = f3(dot(onb.b1, wo), dot(onb.b2, wo), dot(shadingNormal, wo));
    float3 whLocal = SampleGGXVNDF(woLocal, alpha, alpha, u);
    return mad3(whLocal.x, onb.b1, mad3(whLocal.y, onb.b2, whLocal.z * shadingNormal));
}
// BSDF.hlsli:546-554
static inline float GGXMicrofacetPdf(float alpha, float ndotwh, float ndotwo)
{
    float alphaSq = alpha * alpha;
    float NDF = GGX(ndotwh, alphaSq);
    float G1 = SmithG1(alphaSq, ndotwo);
    return (NDF * G1) / ndotwo;
}

// BSDF.hlsli:560-862
struct ShadingData
{
    float alpha;
    float3 wo;
    float ndotwi, ndotwo, ndotwh, whdotwi, whdotwo, wodotwi, g_wo;
    float3 baseColor_Fr0_TrCol;
    float eta;
    bool specTr, metallic, backfacing_wo, invalid, reflection;
    float trDepth;      // half in the reference: always holds an fp16-representable value
    float subsurface;   // half
    float coat_weight;
    float3 coat_color;
    float coat_alpha;
    float coat_eta;

    static ShadingData Init(float3 shadingNormal, float3 wo, bool metallic, float roughness, float3 baseColor,
        float eta_curr = ETA_AIR, float eta_next = DEFAULT_ETA_MAT, bool specTr = false,
        float transmissionDepth = 0, float subsurface = 0, float coat_weight = 0, float3 coat_color = {0, 0, 0},
        float coat_roughness = 0, float eta_coat = DEFAULT_ETA_COAT)
    {
        // Coat roughening
        if (coat_weight > 0 && coat_roughness > 0)
        {
            float rx = roughness * roughness, ry = coat_roughness * coat_roughness;
            rx *= rx; ry *= ry;
            float roughness_coated = zr_min(rx + 2 * ry, 1.0f);
            roughness_coated = zr_rsqrt(zr_rsqrt(roughness_coated));
            roughness = Math::Lerp(roughness, roughness_coated, coat_weight);
        }
        ShadingData si;
        si.wo = wo;
        float ndotwo = dot(shadingNormal, wo);
        si.backfacing_wo = ndotwo <= 0;
        si.ndotwo = zr_max(ndotwo, 1e-5f);
        si.metallic = metallic;
        si.alpha = roughness * roughness;
        si.baseColor_Fr0_TrCol = baseColor;
        si.specTr = specTr;
        si.trDepth = zr_round_f16(transmissionDepth);
        si.subsurface = zr_round_f16(subsurface);
        float eta_base = eta_curr == ETA_AIR ? eta_next : eta_curr;
        float eta_no_coat = eta_next / eta_curr;
        float eta_coated = eta_base >= eta_coat ? eta_base / eta_coat : eta_coat / eta_base;
        si.eta = Math::Lerp(eta_no_coat, eta_coated, coat_weight);
        si.g_wo = !metallic && !specTr ? E_FON_approx(zr_max(ndotwo, 1e-4f), roughness) : 0;
        si.coat_weight = coat_weight;
        si.coat_color = coat_color;
        si.coat_alpha = coat_roughness * coat_roughness;
        si.coat_eta = eta_curr == ETA_AIR ? eta_coat / ETA_AIR : ETA_AIR / eta_coat;
        // fields the reference leaves uninitialised until a SetWi*() call
        si.ndotwi = 0; si.ndotwh = 0; si.whdotwi = 0; si.whdotwo = 0; si.wodotwi = 0;
        si.invalid = true; si.reflection = true;
        return si;
    }

    bool ThinWalled() const { return subsurface > 0; }
    bool Transmissive() const { return specTr || ThinWalled(); }
    bool Coated() const { return coat_weight != 0; }
    bool GlossSpecular() const { return alpha <= MAX_ALPHA_SPECULAR; }
    bool CoatSpecular() const { return coat_alpha <= MAX_ALPHA_SPECULAR; }
    float3 TransmissionTint() const { return trDepth > 0 ? f3(1.0f) : baseColor_Fr0_TrCol; }

    void SetWi_Refl(float3 wi, float3 shadingNormal, float3 wh)
    {
        reflection = true;
        float ndotwi_n = dot(shadingNormal, wi);
        ndotwh = zr_saturate(dot(shadingNormal, wh));
        whdotwo = zr_saturate(dot(wh, wo));
        whdotwi = whdotwo;
        bool isInvalid = backfacing_wo || ndotwh == 0 || whdotwo == 0;
        invalid = isInvalid || ndotwi_n <= 0;
        ndotwi = zr_max(ndotwi_n, 1e-5f);
        wodotwi = dot(wo, wi);
    }
    void SetWi_Refl(float3 wi, float3 shadingNormal)
    {
        float3 wh = normalize(wi + wo);
        SetWi_Refl(wi, shadingNormal, wh);
    }
    void SetWi_Tr(float3 wi, float3 shadingNormal, float3 wh)
    {
        reflection = false;
        float ndotwi_n = dot(shadingNormal, wi);
        ndotwh = zr_saturate(dot(shadingNormal, wh));
        whdotwo = zr_saturate(dot(wh, wo));
        whdotwi = zr_abs(dot(wh, wi));
        bool isInvalid = backfacing_wo || (specTr && (ndotwh == 0 || whdotwo == 0));
        invalid = isInvalid || ndotwi_n >= 0 || !Transmissive() || metallic;
        ndotwi = zr_max(zr_abs(ndotwi_n), 1e-5f);
        wodotwi = dot(wo, wi);
    }
    void SetWi(float3 wi, float3 shadingNormal, float3 wh)
    {
        float ndotwi_n = dot(shadingNormal, wi);
        reflection = ndotwi_n >= 0;
        ndotwh = zr_saturate(dot(shadingNormal, wh));
        whdotwo = zr_saturate(dot(wh, wo));
        bool backfacing_r = ndotwi_n <= 0;
        bool backfacing_t = ndotwi_n >= 0 || !Transmissive() || metallic;
        bool isInvalid = backfacing_wo || (specTr && (ndotwh == 0 || whdotwo == 0));
        invalid = isInvalid || (reflection && backfacing_r) || (!reflection && backfacing_t);
        ndotwi = zr_max(zr_abs(ndotwi_n), 1e-5f);
        whdotwi = zr_abs(dot(wh, wi));
        wodotwi = dot(wo, wi);
    }
    float3 SetWi(float3 wi, float3 shadingNormal)
    {
        float ndotwi_n = dot(shadingNormal, wi);
        reflection = ndotwi_n >= 0;
        float s = reflection ? 1 : eta;
        float3 wh = normalize(mad3(s, wi, wo));   // mad(wi, s, wo)
        wh = !reflection && eta > 1 ? -wh : wh;
        SetWi(wi, shadingNormal, wh);
        return wh;
    }

    float3 Fresnel(float3 fr0, bool& tir) const
    {
        float cosTheta_i = whdotwo;
        tir = false;
        if (metallic) return FresnelSchlick(fr0, cosTheta_i);
        float eta_relative = 1.0f / eta;
        float sinTheta_iSq = zr_saturate(zr_fma(-cosTheta_i, cosTheta_i, 1.0f));
        float cosTheta_tSq = zr_fma(-eta_relative * eta_relative, sinTheta_iSq, 1.0f);
        tir = cosTheta_tSq <= 0;
        if (tir) return f3(1.0f);
        float cosTheta_t = zr_sqrt(cosTheta_tSq);
        return f3(Fresnel_Dielectric(cosTheta_i, eta_relative, cosTheta_t));
    }
    float3 Fresnel() const
    {
        float3 fr0 = metallic ? baseColor_Fr0_TrCol : f3(DielectricF0(eta));
        bool unused;
        return Fresnel(fr0, unused);
    }
    float Fresnel_Coat(float& cosTheta_t) const
    {
        cosTheta_t = 0;
        float cosTheta_i = whdotwo;
        float eta_relative = 1.0f / coat_eta;
        float sinTheta_iSq = zr_saturate(zr_fma(-cosTheta_i, cosTheta_i, 1.0f));
        float cosTheta_tSq = zr_fma(-eta_relative * eta_relative, sinTheta_iSq, 1.0f);
        if (cosTheta_tSq <= 0) return 1;
        cosTheta_t = zr_sqrt(cosTheta_tSq);
        float Fr0 = DielectricF0(coat_eta);
        float cosTheta = coat_eta > 1 ? cosTheta_i : cosTheta_t;
        return FresnelSchlick_Dielectric(Fr0, cosTheta);
    }
    void Regularize() { alpha = alpha < 0.25f ? zr_clamp(2.0f * alpha, 0.1f, 0.25f) : alpha; }
};

// BSDF.hlsli:904-922 (USE_OREN_NAYAR == 1)
static inline float3 EvalDiffuse(bool EON, const ShadingData& surface)
{
    float s = surface.subsurface == 0 ? 1 : surface.subsurface * 0.5f;
    float diffuseRoughness = zr_sqrt(surface.alpha);
    float3 diffuse = OrenNayar(EON, surface.baseColor_Fr0_TrCol, diffuseRoughness, surface.ndotwo, surface.ndotwi,
        surface.wodotwi, surface.g_wo);
    return s * diffuse;
}
// BSDF.hlsli:924-944
static inline float3 SampleDiffuse(float3 normal, float2 u, float& pdf)
{
    float3 wiLocal = Sampling::SampleCosineWeightedHemisphere(u, pdf);
    Math::CoordinateSystem onb = Math::CoordinateSystem::Build(normal);
    return mad3(wiLocal.x, onb.b1, mad3(wiLocal.y, onb.b2, wiLocal.z * normal));
}
static inline float DiffusePdf(const ShadingData& s) { return s.ndotwi * ZR_ONE_OVER_PI; }
// BSDF.hlsli:947-985
static inline float3 EvalGloss(const ShadingData& s, float3 fr)
{ return GGXMicrofacetBRDF(s.alpha, s.ndotwh, s.ndotwo, s.ndotwi, fr, s.GlossSpecular()); }
static inline float3 SampleGloss(const ShadingData& s, float3 n, float2 u)
{
    if (s.GlossSpecular()) return reflect(-s.wo, n);
    float3 wh = SampleGGXMicrofacet(s.wo, s.alpha, n, u);
    return reflect(-s.wo, wh);
}
static inline float GlossPdf(const ShadingData& s)
{
    if (s.GlossSpecular()) return (s.ndotwh >= MIN_N_DOT_H_SPECULAR) ? 1.0f : 0.0f;
    return GGXMicrofacetPdf(s.alpha, s.ndotwh, s.ndotwo) / 4.0f;
}
// BSDF.hlsli:987-1034
static inline float EvalTranslucentTr(const ShadingData& s, float fr)
{ return GGXMicrofacetBTDF(s.alpha, s.ndotwh, s.ndotwo, s.ndotwi, s.whdotwo, s.whdotwi, s.eta, fr, s.GlossSpecular()); }
// BSDF.hlsli:1036-1058
static inline float EvalCoat(const ShadingData& s, float Fr)
{ return s.coat_weight * GGXMicrofacetBRDF(s.coat_alpha, s.ndotwh, s.ndotwo, s.ndotwi, f3(Fr), s.CoatSpecular()).x; }
static inline float3 SampleCoat(const ShadingData& s, float3 n, float2 u)
{
    float3 wh = s.CoatSpecular() ? n : SampleGGXMicrofacet(s.wo, s.coat_alpha, n, u);
    return reflect(-s.wo, wh);
}
static inline float CoatPdf(const ShadingData& s)
{
    if (s.CoatSpecular()) return (s.ndotwh >= MIN_N_DOT_H_SPECULAR) ? 1.0f : 0.0f;
    return GGXMicrofacetPdf(s.coat_alpha, s.ndotwh, s.ndotwo) / 4.0f;
}
// BSDF.hlsli:1078-1092
static inline float3 TranslucentTrOverPdf(const ShadingData& s, float fr)
{
    if (s.GlossSpecular()) return (1 - fr) * s.TransmissionTint();
    float alphaSq = s.alpha * s.alpha;
    return SmithHeightCorrelatedG2OverG1(alphaSq, s.ndotwi, s.ndotwo) * (1 - fr) * s.TransmissionTint();
}
// BSDF.hlsli:1094-1119
static inline float3 BaseWeight(const ShadingData& s)
{
    float3 base_weight = f3(1.0f);
    if (s.Coated())
    {
        float cosTheta_t;
        float Fr_coat = s.Fresnel_Coat(cosTheta_t);
        bool tir_c = cosTheta_t <= 0;
        if (tir_c) return f3(0.0f);
        float reflectance_c = s.CoatSpecular() ? Fr_coat : GGXReflectance_Dielectric(s.coat_alpha, s.ndotwo, s.coat_eta);
        float c = 0.5f / cosTheta_t + 0.5f / s.whdotwo;
        float3 coat_tr = exp3(c * log3(s.coat_color));
        base_weight = Math::Lerp(f3(1.0f), (1 - reflectance_c) * coat_tr, s.coat_weight);
    }
    return base_weight;
}
// BSDF.hlsli:1121-1152
static inline float3 TransmittanceToDielectricBaseTr(const ShadingData& s)
{
    float3 base_weight = BaseWeight(s);
    float reflectance_g = s.GlossSpecular() ? 0 : GGXReflectance_Dielectric(s.alpha, s.ndotwo, s.eta);
    return (1 - reflectance_g) * base_weight;
}
static inline float3 DielectricBaseSpecularTr(const ShadingData& s, float Fr_g)
{
    if (s.invalid || !s.specTr) return f3(0.0f);
    float3 transmittance = TransmittanceToDielectricBaseTr(s);
    float glossyTr = EvalTranslucentTr(s, Fr_g);
    return glossyTr * s.TransmissionTint() * transmittance;
}
static inline float3 DielectricBaseDiffuseTr(const ShadingData& s, float Fr_g)
{
    if (s.invalid) return f3(0.0f);
    float3 base_weight = BaseWeight(s);
    float reflectance_g = s.GlossSpecular() ? Fr_g : GGXReflectance_Dielectric(s.alpha, s.ndotwo, s.eta);
    return (1 - reflectance_g) * EvalDiffuse(false, s) * base_weight;
}

struct BSDFEval { float3 f; float3 Fr_g; bool tir; };

// BSDF.hlsli:1176-1266
static inline BSDFEval Unified(const ShadingData& surface)
{
    BSDFEval ret; ret.f = f3(0.0f); ret.Fr_g = f3(0.0f); ret.tir = false;
    if (surface.invalid) return ret;

    float3 base_weight = f3(1.0f);
    if (surface.Coated())
    {
        float cosThetaT_o;
        float Fr_coat = surface.Fresnel_Coat(cosThetaT_o);
        bool tir_c = cosThetaT_o <= 0;
        if (!surface.reflection && tir_c) return ret;
        if (surface.reflection)
        {
            ret.f = f3(EvalCoat(surface, Fr_coat));
            if (tir_c) return ret;
        }
        float reflectance_c = surface.CoatSpecular() ? Fr_coat :
            GGXReflectance_Dielectric(surface.coat_alpha, surface.ndotwo, surface.coat_eta);
        float c = 1.0f / cosThetaT_o;
        float3 coat_tr = exp3(c * log3(surface.coat_color));
        base_weight = Math::Lerp(f3(1.0f), (1 - reflectance_c) * coat_tr, surface.coat_weight);
    }

    float3 fr0 = surface.metallic ? surface.baseColor_Fr0_TrCol : f3(DielectricF0(surface.eta));
    ret.Fr_g = surface.Fresnel(fr0, ret.tir);
    float3 glossyRefl = EvalGloss(surface, ret.Fr_g);

    if (surface.metallic || ret.tir) { ret.f += base_weight * glossyRefl; return ret; }

    float reflectance_g = surface.GlossSpecular() ? ret.Fr_g.x :
        GGXReflectance_Dielectric(surface.alpha, surface.ndotwo, surface.eta);

    if (!surface.specTr)
    {
        float3 diffuse = EvalDiffuse(true, surface);
        ret.f += base_weight * ((1 - reflectance_g) * diffuse + glossyRefl * (surface.reflection ? 1.0f : 0.0f));
        return ret;
    }
    if (surface.reflection) { ret.f += glossyRefl * base_weight; return ret; }

    reflectance_g = surface.GlossSpecular() ? 0 : reflectance_g;
    float glossyTr = EvalTranslucentTr(surface, ret.Fr_g.x);
    ret.f = ((1 - reflectance_g) * glossyTr * surface.TransmissionTint()) * base_weight;
    return ret;
}

//--------------------------------------------------------------------------------------
// BSDFSampling.hlsli
//--------------------------------------------------------------------------------------

struct BSDFSample
{
    float3 wi; LOBE lobe; float pdf; float3 bsdfOverPdf; float3 f;
    static BSDFSample Init()
    { BSDFSample r; r.bsdfOverPdf = f3(0.0f); r.f = f3(0.0f); r.pdf = 0; r.wi = f3(0.0f); r.lobe = LOBE::ALL; return r; }
};
struct BSDFSamplerEval { float pdf; float3 bsdfOverPdf; float3 f; };
struct NoOp { float3 operator()(float3) const { return f3(1.0f); } };

// BSDFSampling.hlsli:59-152
template<typename Func>
static BSDFSample SampleBSDF_NoDiffuse(float3 normal, ShadingData surface, float2 u_c, float2 u_g, float u_wrs_0,
    float u_wrs_1, Func func)
{
    BSDFSample ret = BSDFSample::Init();
    float pdf_base = 1;
    if (surface.Coated())
    {
        float reflectance_c = GGXReflectance_Dielectric(surface.coat_alpha, surface.ndotwo, surface.coat_eta);
        float pdf_coat = reflectance_c * surface.coat_weight;
        pdf_base = 1 - pdf_coat;
        if (u_wrs_0 < pdf_coat)
        {
            float3 wi_c = SampleCoat(surface, normal, u_c);
            surface.SetWi_Refl(wi_c, normal);
            BSDFEval eval = Unified(surface);
            float3 target = eval.f * func(wi_c);
            ret.wi = wi_c; ret.lobe = LOBE::COAT; ret.f = target;
            ret.pdf = CoatPdf(surface) * pdf_coat;
            ret.bsdfOverPdf = ret.f / ret.pdf;
            return ret;
        }
    }
    float3 wh = surface.GlossSpecular() ? normal : SampleGGXMicrofacet(surface.wo, surface.alpha, normal, u_g);
    float3 wi_r = reflect(-surface.wo, wh);
    surface.SetWi_Refl(wi_r, normal, wh);
    float wh_pdf = GGXMicrofacetPdf(surface.alpha, surface.ndotwh, surface.ndotwo);
    ret.wi = wi_r; ret.lobe = LOBE::GLOSSY_R;
    ret.pdf = surface.GlossSpecular() ? 1 : wh_pdf / 4.0f;
    ret.pdf *= pdf_base;
    BSDFEval eval = Unified(surface);
    float3 func_r = func(wi_r);
    ret.f = eval.f * func_r;
    ret.bsdfOverPdf = ret.f / ret.pdf;
    if (surface.metallic || !surface.specTr || eval.tir) return ret;

    float3 wi_t = refract(-surface.wo, wh, 1 / surface.eta);
    float3 func_t = func(wi_t);
    float p_r = eval.Fr_g.x * Math::Luminance(func_r);
    p_r = p_r / (p_r + (1 - eval.Fr_g.x) * Math::Luminance(func_t));
    if (u_wrs_1 < p_r)
    {
        ret.bsdfOverPdf /= p_r;
        ret.pdf *= p_r;
    }
    else
    {
        surface.SetWi_Tr(wi_t, normal, wh);
        ret.pdf = (1 - p_r) * pdf_base;
        if (!surface.GlossSpecular())
        {
            ret.pdf *= wh_pdf * surface.whdotwo;
            float dwh_dwi = JacobianHalfVecToIncident_Tr(surface.eta, surface.whdotwo, surface.whdotwi);
            ret.pdf *= dwh_dwi;
        }
        ret.f = DielectricBaseSpecularTr(surface, eval.Fr_g.x) * func_t;
        ret.bsdfOverPdf = ret.pdf > 0 ? ret.f / ret.pdf : f3(0.0f);
        ret.wi = wi_t; ret.lobe = LOBE::GLOSSY_T;
    }
    return ret;
}

// BSDFSampling.hlsli:181-296
template<typename Func>
static BSDFSample SampleBSDF_NoSpecTr(float3 normal, ShadingData surface, float2 u_coat, float2 u_g, float2 u_d,
    float u_wrs_g, float u_wrs_dr, float u_wrs_dt, Func func)
{
    BSDFSample ret = BSDFSample::Init();
    float w_sum = 0;
    float3 target = f3(0.0f);

    if (surface.Coated())
    {
        float3 wi_c = SampleCoat(surface, normal, u_coat);
        surface.SetWi_Refl(wi_c, normal);
        BSDFEval eval = Unified(surface);
        target = eval.f * func(wi_c);
        ret.wi = wi_c; ret.lobe = LOBE::COAT; ret.f = target;
        float pdf_c = CoatPdf(surface);
        float pdf_g = GlossPdf(surface);
        float pdf_d = !surface.metallic ? DiffusePdf(surface) : 0;
        w_sum = RT::BalanceHeuristic3(pdf_c, pdf_g, pdf_d, Math::Luminance(target));
    }
    {
        float3 wi_g = SampleGloss(surface, normal, u_g);
        surface.SetWi_Refl(wi_g, normal);
        BSDFEval eval = Unified(surface);
        float3 target_g = eval.f * func(wi_g);
        float pdf_g = GlossPdf(surface);
        float pdf_d = !surface.metallic && !eval.tir ? DiffusePdf(surface) : 0;
        float pdf_c = surface.Coated() ? CoatPdf(surface) : 0;
        float w_g = RT::BalanceHeuristic3(pdf_g, pdf_d, pdf_c, Math::Luminance(target_g));
        w_sum += w_g;
        if ((w_sum > 0) && (u_wrs_g < (w_g / w_sum)))
        { target = target_g; ret.wi = wi_g; ret.lobe = LOBE::GLOSSY_R; ret.f = target_g; }
    }
    if (!surface.metallic)
    {
        float pdf_d;
        float3 wi_d = SampleDiffuse(normal, u_d, pdf_d);
        float Fr_g;
        {
            surface.SetWi_Refl(wi_d, normal);
            BSDFEval eval = Unified(surface);
            float3 target_dr = eval.f * func(wi_d);
            Fr_g = eval.Fr_g.x;
            float pdf_g = GlossPdf(surface);
            float pdf_c = surface.Coated() ? CoatPdf(surface) : 0;
            float w_dr = RT::BalanceHeuristic3(pdf_d, pdf_g, pdf_c, Math::Luminance(target_dr));
            w_sum += w_dr;
            if ((w_sum > 0) && (u_wrs_dr < (w_dr / w_sum)))
            { target = target_dr; ret.wi = wi_d; ret.lobe = LOBE::DIFFUSE_R; ret.f = target_dr; }
        }
        if (surface.ThinWalled())
        {
            float3 wi_dt = -wi_d;
            float3 target_dt = DielectricBaseDiffuseTr(surface, Fr_g);
            target_dt *= func(wi_dt);
            float w_dt = Math::Luminance(target_dt) / pdf_d;
            w_sum += w_dt;
            if ((w_sum > 0) && (u_wrs_dt < (w_dt / w_sum)))
            { target = target_dt; ret.wi = wi_dt; ret.lobe = LOBE::DIFFUSE_T; ret.f = target_dt; }
        }
    }
    float targetLum = Math::Luminance(target);
    ret.bsdfOverPdf = targetLum > 0 ? target * w_sum / targetLum : f3(0.0f);
    ret.pdf = w_sum > 0 ? targetLum / w_sum : 0;
    return ret;
}

// BSDFSampling.hlsli:298-338: always burns 9 uniforms so random replay stays aligned
template<typename Func>
static BSDFSample SampleBSDF(float3 normal, const ShadingData& surface, Func func, RNG& rng)
{
    float2 u_c = rng.Uniform2D();
    float2 u_g = rng.Uniform2D();
    float2 u_d = rng.Uniform2D();
    float u_wrs_0 = rng.Uniform();
    float u_wrs_1 = rng.Uniform();
    float u_wrs_2 = rng.Uniform();
    if (!surface.specTr)
        return SampleBSDF_NoSpecTr(normal, surface, u_c, u_g, u_d, u_wrs_0, u_wrs_1, u_wrs_2, func);
    return SampleBSDF_NoDiffuse(normal, surface, u_c, u_g, u_wrs_0, u_wrs_1, func);
}
static inline BSDFSample SampleBSDF(float3 normal, const ShadingData& surface, RNG& rng)
{ return SampleBSDF(normal, surface, NoOp(), rng); }

// BSDFSampling.hlsli:154-165
static inline BSDFSample SampleBSDF_NoDiffuse(float3 normal, const ShadingData& surface, RNG& rng)
{
    float2 u_c = rng.Uniform2D();
    float2 u_g = rng.Uniform2D();
    float u_wrs_0 = rng.Uniform();
    float u_wrs_1 = rng.Uniform();
    return SampleBSDF_NoDiffuse(normal, surface, u_c, u_g, u_wrs_0, u_wrs_1, NoOp());
}

// BSDFSampling.hlsli:565-631
template<typename Func>
static float BSDFSamplerPdf_NoDiffuse(float3 normal, ShadingData surface, float3 wi, Func func)
{
    float3 wh = surface.SetWi(wi, normal);
    float pdf_base = 1;
    float pdf_c = 0;
    if (surface.Coated())
    {
        float reflectance_c = GGXReflectance_Dielectric(surface.coat_alpha, surface.ndotwo, surface.coat_eta);
        float pdf_coat = reflectance_c * surface.coat_weight;
        pdf_base = 1 - pdf_coat;
        if (surface.reflection) pdf_c = CoatPdf(surface) * pdf_coat;
    }
    const float wh_pdf = GGXMicrofacetPdf(surface.alpha, surface.ndotwh, surface.ndotwo);
    if (surface.metallic || !surface.specTr)
    {
        float pdf_gr = surface.GlossSpecular() ? (surface.ndotwh >= MIN_N_DOT_H_SPECULAR ? 1.0f : 0.0f) : wh_pdf / 4.0f;
        pdf_gr *= pdf_base;
        return surface.reflection ? pdf_c + pdf_gr : 0;
    }
    float pdf_g = surface.GlossSpecular() ? (surface.ndotwh >= MIN_N_DOT_H_SPECULAR ? 1.0f : 0.0f) : 1;
    pdf_g *= pdf_base;
    const float3 wi_other = !surface.reflection ? reflect(-surface.wo, wh) : refract(-surface.wo, wh, 1 / surface.eta);
    float targetScaleLum = Math::Luminance(func(wi));
    float targetScaleOtherLum = Math::Luminance(func(wi_other));
    float Fr_g = surface.Fresnel().x;
    float pdf_r = Fr_g * (surface.reflection ? targetScaleLum : targetScaleOtherLum);
    pdf_r = pdf_r / (pdf_r + (1 - Fr_g) * (surface.reflection ? targetScaleOtherLum : targetScaleLum));
    if (surface.reflection)
    {
        pdf_g *= surface.GlossSpecular() ? 1 : (wh_pdf / 4.0f);
        pdf_g *= pdf_r;
        return pdf_g + pdf_c;
    }
    pdf_g *= 1 - pdf_r;
    if (!surface.GlossSpecular())
    {
        pdf_g *= wh_pdf * surface.whdotwo;
        float dwh_dwi = JacobianHalfVecToIncident_Tr(surface.eta, surface.whdotwo, surface.whdotwi);
        pdf_g *= dwh_dwi;
    }
    return pdf_g;
}

// BSDFSampling.hlsli:639-759
template<typename Func>
static float BSDFSamplerPdf(float3 normal, ShadingData surface, float3 wi_z, Func func, RNG& rng)
{
    if (surface.specTr) return BSDFSamplerPdf_NoDiffuse(normal, surface, wi_z, func);

    surface.SetWi(wi_z, normal);
    if (!surface.reflection && !surface.ThinWalled()) return 0;

    BSDFEval eval_z = Unified(surface);
    float targetLum = Math::Luminance(eval_z.f * func(wi_z));
    if (targetLum == 0) return 0;

    float w_sum_c, w_sum_g, w_sum_dr, w_sum_dt;
    {
        float pdf_g = GlossPdf(surface);
        float pdf_d = !surface.metallic ? DiffusePdf(surface) : 0;
        float pdf_c = surface.Coated() ? CoatPdf(surface) : 0;
        float w = surface.reflection ? RT::BalanceHeuristic3(pdf_g, pdf_d, pdf_c, targetLum) :
            (targetLum / pdf_d) * (!surface.metallic ? 1.0f : 0.0f);
        w_sum_g = w; w_sum_dr = w; w_sum_dt = w; w_sum_c = w;
    }
    if (w_sum_g == 0) return 0;

    float pdf_d;
    float3 wi_d = SampleDiffuse(normal, rng.Uniform2D(), pdf_d);
    float Fr_g = 0;
    if (!surface.metallic)
    {
        surface.SetWi_Refl(wi_d, normal);
        BSDFEval eval = Unified(surface);
        Fr_g = eval.Fr_g.x;
        float targetLum_dr = Math::Luminance(eval.f * func(wi_d));
        float pdf_g = GlossPdf(surface);
        float pdf_c = surface.Coated() ? CoatPdf(surface) : 0;
        float w = RT::BalanceHeuristic3(pdf_d, pdf_g, pdf_c, targetLum_dr);
        w_sum_g += w; w_sum_dt += w; w_sum_c += w;
    }
    if (!surface.metallic && surface.ThinWalled())
    {
        float3 target_dt = DielectricBaseDiffuseTr(surface, Fr_g);
        float targetLum_dt = Math::Luminance(target_dt * func(-wi_d));
        float w = targetLum_dt / pdf_d;
        w_sum_g += w; w_sum_dr += w; w_sum_c += w;
    }
    {
        float3 wi_g = SampleGloss(surface, normal, rng.Uniform2D());
        surface.SetWi_Refl(wi_g, normal);
        float3 target_g = Unified(surface).f;
        float targetLum_g = Math::Luminance(target_g * func(wi_g));
        float pdf_g = GlossPdf(surface);
        float pdf_dd = !surface.metallic ? DiffusePdf(surface) : 0;
        float pdf_c = surface.Coated() ? CoatPdf(surface) : 0;
        float w = RT::BalanceHeuristic3(pdf_g, pdf_dd, pdf_c, targetLum_g);
        w_sum_dr += w; w_sum_dt += w; w_sum_c += w;
    }
    if (surface.Coated())
    {
        float3 wi_c = SampleCoat(surface, normal, rng.Uniform2D());
        surface.SetWi_Refl(wi_c, normal);
        float3 target_c = Unified(surface).f;
        float targetLum_c = Math::Luminance(target_c * func(wi_c));
        float pdf_g = GlossPdf(surface);
        float pdf_dd = !surface.metallic ? DiffusePdf(surface) : 0;
        float pdf_c = CoatPdf(surface);
        float w = RT::BalanceHeuristic3(pdf_g, pdf_dd, pdf_c, targetLum_c);
        w_sum_g += w; w_sum_dr += w; w_sum_dt += w;
    }
    float pdf = w_sum_g > 0 ? targetLum / w_sum_g : 0;
    pdf += w_sum_dr > 0 ? targetLum / w_sum_dr : 0;
    pdf += w_sum_c > 0 ? targetLum / w_sum_c : 0;
    pdf += surface.ThinWalled() && (w_sum_dt > 0) ? targetLum / w_sum_dt : 0;
    return pdf;
}

} // namespace BSDF
} // namespace zro
