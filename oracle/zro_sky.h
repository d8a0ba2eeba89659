// zro_sky.h -- TEST INFRASTRUCTURE ONLY (CPU oracle; nothing under zetaray_amd/ may include, link or call this).
//
// Scalar restatement of the reference's sky model:
//   Source/ZetaRenderPass/Common/Volumetric.hlsli:34-229   (phase functions, densities, EstimateTransmittance, EstimateLs)
//   Source/ZetaRenderPass/Sky/SkyViewLUT.hlsl:19-63        (K17: sky-view LUT, non-linear latitude, R11G11B10_FLOAT)
//   Source/ZetaRenderPass/Common/LightSource.hlsli:139-199 (Le_Sun, Le_Sky, Le_SkyWithSunDisk)
// Arithmetic contract as everywhere (zr_detmath.h transcendentals, no contraction).  Pinned where D3D leaves it open:
// the UAV store converts to R11G11B10_FLOAT with round-to-nearest-even (PackUFloat), and g_samLinearWrap is fp32 bilinear
// interpolation with texel centres at (i + 0.5) / N and wrap addressing on both axes (same convention as the rho LUT).
// Parity unpinned against the reference: it has no CPU sky model and no golden LUT.
#pragma once
#include "zro_math.h"
#include "../include/zr_wire.h"

namespace zro {

namespace Volume {

static inline float RayleighPhaseFunction(float cosTheta) { return 0.0596831f * (1.0f + cosTheta * cosTheta); }
// Volumetric.hlsli:52-58
static inline float SchlickPhaseFunction(float cosTheta, float g)
{
    float k = 1.55f * g - 0.55f * g * g * g;
    float denom = 1.0f - k * cosTheta;
    return ZR_ONE_OVER_4_PI * (1.0f - k * k) / (denom * denom);
}
// Volumetric.hlsli:66-88 (altitudes in km)
static inline float DensityRayleigh(float altitude) { return zr_exp(-zr_max(0.0f, altitude / 8.0f)); }
static inline float DensityMie(float altitude) { return zr_exp(-zr_max(0.0f, altitude / 1.2f)); }
static inline float DensityOzone(float altitude) { return zr_max(0.0f, 1 - zr_abs(altitude - 25.0f) / 15.0f); }
static inline float3 AtmosphereDensity(float altitude) { return f3(DensityRayleigh(altitude), DensityMie(altitude), DensityOzone(altitude)); }
static inline float Altitude(float3 pos, float planetRadius) { return length(pos) - planetRadius; }
// Volumetric.hlsli:101-112
static inline float IntersectRayAtmosphere(float radius, float3 rayOrigin, float3 rayDir)
{
    float mDotdir = dot(rayDir, rayOrigin);
    float delta = mDotdir * mDotdir - dot(rayOrigin, rayOrigin) + radius * radius;
    delta = zr_sqrt(delta);
    return -mDotdir + delta;
}
// Volumetric.hlsli:114-132
static inline bool IntersectRayPlanet(float radius, float3 rayOrigin, float3 rayDir, float& t)
{
    float mDotdir = dot(rayDir, rayOrigin);
    float delta = mDotdir * mDotdir - dot(rayOrigin, rayOrigin) + radius * radius;
    if (delta < 0.0f) { t = 0; return false; }
    delta = zr_sqrt(delta);
    t = zr_min(-mDotdir - delta, -mDotdir + delta);
    return t >= 0.0f;
}
// Volumetric.hlsli:137-170
static inline float3 EstimateTransmittance(float planetRadius, float3 rayOrigin, float3 rayDir, float t,
    float3 sigma_t_rayleigh, float sigma_t_mie, float3 sigma_t_ozone, int numSteps)
{
    if (t <= 1e-5f) return f3(1.0f);
    const float stepSize = t / (float)numSteps;
    float3 pos = rayOrigin + 0.5f * stepSize * rayDir;
    float3 opticalThickness = f3(0.0f);
    for (int s = 0; s < numSteps; s++)
    {
        float altitude = Altitude(pos, planetRadius);
        float3 density = AtmosphereDensity(altitude);
        opticalThickness += density;
        pos += stepSize * rayDir;
    }
    opticalThickness = sigma_t_rayleigh * opticalThickness.x + sigma_t_mie * opticalThickness.y + sigma_t_ozone * opticalThickness.z;
    opticalThickness *= stepSize;
    return exp3(-opticalThickness);
}
// Volumetric.hlsli:173-229
static inline float3 EstimateLs(float planetRadius, float3 rayOrigin, float3 rayDir, float3 lightDir, float atmosphereHeight,
    float g, float3 sigma_s_rayleigh, float sigma_s_mie, float sigma_t_mie, float3 sigma_t_ozone, int numSteps)
{
    float t = IntersectRayAtmosphere(planetRadius + atmosphereHeight, rayOrigin, rayDir);
    float tPlanet;
    bool intersectedPlanet = IntersectRayPlanet(planetRadius, rayOrigin, rayDir, tPlanet);
    if (intersectedPlanet) t = tPlanet;
    const float stepSize = t / (float)numSteps;
    float3 pos = rayOrigin + 0.5f * stepSize * rayDir;
    float3 opticalThickness = f3(0.0f), LsRayleigh = f3(0.0f), LsMie = f3(0.0f);
    for (int s = 0; s < numSteps; s++)
    {
        float altitude = Altitude(pos, planetRadius);
        float3 density = AtmosphereDensity(altitude);
        opticalThickness += density * stepSize;
        float3 rayOriginToPosTr = exp3(-(sigma_s_rayleigh * opticalThickness.x + sigma_t_mie * opticalThickness.y +
            sigma_t_ozone * opticalThickness.z));
        const float posToAtmosphereDist = IntersectRayAtmosphere(planetRadius + atmosphereHeight, pos, -lightDir);
        float3 LoTranmittance = EstimateTransmittance(planetRadius, pos, -lightDir, posToAtmosphereDist, sigma_s_rayleigh,
            sigma_t_mie, sigma_t_ozone, 8);
        LsRayleigh += rayOriginToPosTr * density.x * LoTranmittance;
        LsMie += rayOriginToPosTr * density.y * LoTranmittance;
        pos += stepSize * rayDir;
    }
    const float cosTheta = dot(lightDir, -rayDir);
    const float phaseRayleigh = RayleighPhaseFunction(cosTheta);
    const float phaseMie = SchlickPhaseFunction(cosTheta, g);
    float3 Ls = LsRayleigh * sigma_s_rayleigh * phaseRayleigh;
    Ls += LsMie * sigma_s_mie * phaseMie;
    Ls *= stepSize;
    return Ls;
}

} // namespace Volume

// the bound sky-view LUT (R11G11B10_FLOAT texels, row-major)
struct SkyLUT { const uint32_t* data = nullptr; uint32_t w = 0, h = 0; };

// K17, SkyViewLUT.hlsl:19-63: texel (x, y) of a w x h LUT
static inline uint32_t SkyViewLUT_Texel(const zr_frame_constants& g, uint32_t x, uint32_t y, uint32_t w, uint32_t h)
{
    float phi = ((float)x / (float)w);
    phi *= ZR_TWO_PI;
    float v = ((float)y / (float)h);
    float s = v >= 0.5f ? 1.0f : -1.0f;
    float a = v - 0.5f;
    float theta = a * a * ZR_TWO_PI * s + ZR_PI_OVER_2;
    // Math::SphericalToCartesian(1, theta, phi), Math.hlsli:115-119
    float sinTheta = zr_sin(theta);
    float3 wdir = f3(1.0f * sinTheta * zr_cos(phi), 1.0f * zr_cos(theta), -1.0f * sinTheta * zr_sin(phi));
    const float3 sigma_s_rayleigh = f3(g.rayleigh_sigma_s_color) * g.rayleigh_sigma_s_scale;
    const float sigma_t_mie = g.mie_sigma_a + g.mie_sigma_s;
    const float3 sigma_t_ozone = f3(g.ozone_sigma_a_color) * g.ozone_sigma_a_scale;
    float3 rayOrigin = f3(0.0f, g.planet_radius + 0.2f, 0.0f);
    float3 Ls = Volume::EstimateLs(g.planet_radius, rayOrigin, wdir, f3(g.sun_dir), g.atmosphere_altitude, g.g,
        sigma_s_rayleigh, g.mie_sigma_s, sigma_t_mie, sigma_t_ozone, 32);
    Ls *= g.sun_illuminance;
    return PackR11G11B10F(max3(Ls, 0.0f));
}

namespace Light {

// LightSource.hlsli:139-157
static inline float3 Le_Sun(float3 pos, const zr_frame_constants& g)
{
    const float3 sigma_t_rayleigh = f3(g.rayleigh_sigma_s_color) * g.rayleigh_sigma_s_scale;
    const float sigma_t_mie = g.mie_sigma_a + g.mie_sigma_s;
    const float3 sigma_t_ozone = f3(g.ozone_sigma_a_color) * g.ozone_sigma_a_scale;
    float3 temp = pos;
    temp.y += g.planet_radius;
    const float t = Volume::IntersectRayAtmosphere(g.planet_radius + g.atmosphere_altitude, temp, -f3(g.sun_dir));
    const float3 tr = Volume::EstimateTransmittance(g.planet_radius, temp, -f3(g.sun_dir), t, sigma_t_rayleigh, sigma_t_mie,
        sigma_t_ozone, 6);
    return tr * g.sun_illuminance;
}

static inline float3 SkyTexel(const SkyLUT& lut, int x, int y)
{
    const uint32_t v = lut.data[(size_t)y * lut.w + x];
    return f3(zr_unpack_ufloat(v & 0x7ff, 6), zr_unpack_ufloat((v >> 11) & 0x7ff, 6), zr_unpack_ufloat(v >> 22, 5));
}
// g_samLinearWrap.SampleLevel(uv, 0) as pinned in the header
static inline float3 SampleSkyLUT(const SkyLUT& lut, float2 uv)
{
    const float x = uv.x * (float)lut.w - 0.5f, y = uv.y * (float)lut.h - 0.5f;
    const float fx = zr_floor(x), fy = zr_floor(y);
    const float tx = x - fx, ty = y - fy;
    auto wrap = [](int i, int n) { int m = i % n; return m < 0 ? m + n : m; };
    const int x0 = wrap((int)fx, (int)lut.w), x1 = wrap((int)fx + 1, (int)lut.w);
    const int y0 = wrap((int)fy, (int)lut.h), y1 = wrap((int)fy + 1, (int)lut.h);
    const float3 a = SkyTexel(lut, x0, y0), b = SkyTexel(lut, x1, y0), c = SkyTexel(lut, x0, y1), d = SkyTexel(lut, x1, y1);
    const float3 r0 = f3(zr_lerp(a.x, b.x, tx), zr_lerp(a.y, b.y, tx), zr_lerp(a.z, b.z, tx));
    const float3 r1 = f3(zr_lerp(c.x, d.x, tx), zr_lerp(c.y, d.y, tx), zr_lerp(c.z, d.z, tx));
    return f3(zr_lerp(r0.x, r1.x, ty), zr_lerp(r0.y, r1.y, ty), zr_lerp(r0.z, r1.z, ty));
}
// LightSource.hlsli:159-174
static inline float3 Le_Sky(float3 wi, const SkyLUT& lut)
{
    const float2 thetaPhi = Math::SphericalFromCartesian(wi);
    float2 uv = f2(thetaPhi.y * ZR_ONE_OVER_2_PI, thetaPhi.x * ZR_ONE_OVER_PI);
    const float sn = thetaPhi.x >= ZR_PI_OVER_2 ? 1.0f : -1.0f;
    uv.y = zr_fma(0.5f, thetaPhi.x, -ZR_PI_OVER_4);
    uv.y = 0.5f + sn * zr_sqrt(zr_abs(uv.y) * ZR_ONE_OVER_PI);
    return SampleSkyLUT(lut, uv);
}

// Light::Le_SkyWithSunDisk, LightSource.hlsli:176-199 (miss pixels of SkyDI, of the emissive DI pass and of Compositing)
static inline float3 Le_SkyWithSunDisk(uint32_t DTid_x, uint32_t DTid_y, const zr_frame_constants& g_frame, const SkyLUT& envMap)
{
    float3 wc = RT::GeneratePinholeCameraRay((int)DTid_x, (int)DTid_y, f2((float)g_frame.render_width, (float)g_frame.render_height),
        g_frame.aspect_ratio, g_frame.tan_half_fov, f3(g_frame.curr_view), f3(g_frame.curr_view + 4), f3(g_frame.curr_view + 8),
        f2(g_frame.curr_camera_jitter[0], g_frame.curr_camera_jitter[1]));
    float3 rayOrigin = f3(0, 1e-1f, 0);
    rayOrigin.y += g_frame.planet_radius;
    float3 wTemp = wc;
    // cos(a - b) = cos a cos b + sin a sin b
    wTemp.y = wTemp.y * g_frame.sun_cos_angular_radius + zr_sqrt(1 - wc.y * wc.y) * g_frame.sun_sin_angular_radius;
    float t;
    bool intersectedPlanet = Volume::IntersectRayPlanet(g_frame.planet_radius, rayOrigin, wTemp, t);
    if (dot(-wc, f3(g_frame.sun_dir)) >= g_frame.sun_cos_angular_radius && !intersectedPlanet)
        return f3(g_frame.sun_illuminance);
    return Le_Sky(wc, envMap);
}

} // namespace Light
} // namespace zro
