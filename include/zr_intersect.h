/*
 * zr_intersect.h -- the ray/triangle and ray/box arithmetic of the zetaray_amd C-ABI.
 *
 * In the reference, BVH traversal and ray/triangle intersection happen inside the D3D12 driver / RT hardware
 * (DXR 1.1 inline RayQuery: Source/ZetaRenderPass/Common/RayQuery.hlsli:42-53, 168-179, 317-331, 372-396;
 * GBuffer/GBufferRT_Inline.hlsl:72-95); their arithmetic is opaque and unpinned by any reference test.  Only the
 * semantics are defined: closest hit / any hit with TMin < t < TMax over world-space triangles.  This ABI pins the
 * arithmetic instead (SURVEY.md section 7 "Hard parts"): Moeller-Trumbore in the form below, no face culling, hit
 * accepted iff tmin < t < tmax, ties on t broken towards the smaller global triangle index.  Both the HIP kernels
 * and the CPU oracle call these two functions; everything around them (BVH layout, traversal order) is free,
 * because "closest hit with this tie-break" and "any hit" do not depend on traversal order.
 *
 * World-space triangles are stored as (v0, e1 = v1 - v0, e2 = v2 - v0), computed once on the host from the
 * instance's float 3x4 object-to-world matrix (row . (p, 1), summed left to right).
 */
#ifndef ZR_INTERSECT_H
#define ZR_INTERSECT_H

#include "zr_detmath.h"

/* returns 1 on hit and writes t, u (weight of v1), v (weight of v2) -- DXR barycentric convention */
ZR_HD int zr_ray_tri(float ox, float oy, float oz, float dx, float dy, float dz,
                     float v0x, float v0y, float v0z, float e1x, float e1y, float e1z,
                     float e2x, float e2y, float e2z, float tmin, float tmax,
                     float* t_out, float* u_out, float* v_out)
{
    /* p = d x e2 */
    float px = dy * e2z - dz * e2y;
    float py = dz * e2x - dx * e2z;
    float pz = dx * e2y - dy * e2x;
    float det = e1x * px + e1y * py + e1z * pz;
    if (det == 0.0f) return 0;
    float inv = 1.0f / det;
    float tx = ox - v0x, ty = oy - v0y, tz = oz - v0z;
    float u = (tx * px + ty * py + tz * pz) * inv;
    if (!(u >= 0.0f && u <= 1.0f)) return 0;
    /* q = tv x e1 */
    float qx = ty * e1z - tz * e1y;
    float qy = tz * e1x - tx * e1z;
    float qz = tx * e1y - ty * e1x;
    float v = (dx * qx + dy * qy + dz * qz) * inv;
    if (!(v >= 0.0f && (u + v) <= 1.0f)) return 0;
    float t = (e2x * qx + e2y * qy + e2z * qz) * inv;
    if (!(t > tmin && t < tmax)) return 0;
    *t_out = t; *u_out = u; *v_out = v;
    return 1;
}

/* reciprocal direction for the slab test; |d| is clamped to 1e-30 so 0 * inf never produces a NaN */
ZR_HD float zr_safe_rcp_dir(float d)
{
    if (zr_abs(d) < 1e-30f) d = (zr_asuint(d) & 0x80000000u) ? -1e-30f : 1e-30f;
    return 1.0f / d;
}

/*
 * Conservative slab test.  idx/idy/idz = zr_safe_rcp_dir(d).  The exit distance is widened by 2 ulp-ish
 * (pbrt's 1 + 2*gamma(3)) so a box test never rejects a triangle zr_ray_tri would accept.
 * Returns 1 if [tmin, tmax] overlaps the box interval; writes the entry distance (for ordered traversal only).
 */
ZR_HD int zr_ray_box(float ox, float oy, float oz, float idx, float idy, float idz,
                     float bminx, float bminy, float bminz, float bmaxx, float bmaxy, float bmaxz,
                     float tmin, float tmax, float* t_entry)
{
    float t0x = (bminx - ox) * idx, t1x = (bmaxx - ox) * idx;
    float t0y = (bminy - oy) * idy, t1y = (bmaxy - oy) * idy;
    float t0z = (bminz - oz) * idz, t1z = (bmaxz - oz) * idz;
    /* NaN (0 * inf) compares false and drops out of min/max, which is the conservative choice here */
    float nx = zr_min(t0x, t1x), fx = zr_max(t0x, t1x);
    float ny = zr_min(t0y, t1y), fy = zr_max(t0y, t1y);
    float nz = zr_min(t0z, t1z), fz = zr_max(t0z, t1z);
    float tn = zr_max(zr_max(nx, ny), zr_max(nz, tmin));
    float tf = zr_min(zr_min(fx, fy), zr_min(fz, tmax));
    tf *= 1.0000003576278687f;
    *t_entry = tn;
    return tn <= tf;
}

/*
 * The same test with the machine's own min / max (v_min_f32 / v_max_f32 on gfx950, which fuse to the three-operand
 * forms): what BVH traversal uses.  A box test only selects candidates -- closest hit with the index tie-break and any
 * hit do not depend on which conservative test produced them -- so unlike zr_ray_tri its arithmetic is not part of the
 * parity contract; it only has to never reject a box whose triangle zr_ray_tri would accept (same widening as above).
 */
ZR_HD int zr_ray_box_native(float ox, float oy, float oz, float idx, float idy, float idz,
                            float bminx, float bminy, float bminz, float bmaxx, float bmaxy, float bmaxz,
                            float tmin, float tmax, float* t_entry)
{
    float t0x = (bminx - ox) * idx, t1x = (bmaxx - ox) * idx;
    float t0y = (bminy - oy) * idy, t1y = (bmaxy - oy) * idy;
    float t0z = (bminz - oz) * idz, t1z = (bmaxz - oz) * idz;
    float tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(t0x, t1x), __builtin_fminf(t0y, t1y)), __builtin_fmaxf(__builtin_fminf(t0z, t1z), tmin));
    float tf = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(t0x, t1x), __builtin_fmaxf(t0y, t1y)), __builtin_fminf(__builtin_fmaxf(t0z, t1z), tmax));
    tf *= 1.0000003576278687f;
    *t_entry = tn;
    return tn <= tf;
}

#endif /* ZR_INTERSECT_H */
