/*
 * zr_wire.h -- bit-exact wire formats shared by the host caller, the HIP kernels and the CPU oracle.
 *
 * Every struct below restates a GPU data layout of the reference (alipbcs/ZetaRay); the reference file:line each
 * one follows is cited next to it.  These are the formats a ZetaRay-style renderer already holds in its upload
 * buffers, so a drop-in caller hands them over unchanged.
 *
 * Plain C (also valid C++ / HIP).  No torch types, no D3D12 types.
 */
#ifndef ZR_WIRE_H
#define ZR_WIRE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* reference: Source/ZetaCore/Core/Vertex.h:8-14, Source/ZetaRenderPass/Common/Common.hlsli:5-11 (28 B) */
typedef struct zr_vertex {
    float    pos[3];      /* PosL */
    float    uv[2];       /* TexUV */
    uint16_t normal[2];   /* oct32 NormalL  (2 x UNORM16) */
    uint16_t tangent[2];  /* oct32 TangentU (2 x UNORM16) */
} zr_vertex;

/* reference: Source/ZetaCore/RayTracing/RtCommon.h:47-64 (64 B) */
typedef struct zr_mesh_instance {
    uint32_t base_vtx_offset;
    uint32_t base_idx_offset;
    uint16_t rotation[4];       /* unorm4: quaternion mapped [-1,1] -> [0,65535] */
    uint16_t scale[3];          /* half3 */
    uint16_t mat_idx;
    uint32_t base_emissive_tri_offset; /* UINT32_MAX when the instance is not emissive */
    float    translation[3];
    uint16_t prev_rotation[4];
    uint16_t prev_scale[3];
    uint16_t d_translation[3];  /* half3 */
    uint16_t base_color_tex;
    uint16_t alpha_factor_cutoff; /* RG8 */
} zr_mesh_instance;

/* reference: Source/ZetaCore/Core/Material.h:419-427 (32 B, 8 packed dwords) */
typedef struct zr_material {
    uint32_t base_color_factor;                 /* RGBA8 */
    uint32_t base_color_tex_subsurf_coat_weight;/* tex16 | UNORM8 subsurface << 16 | UNORM8 coat weight << 24 */
    uint32_t normal_tex_tr_depth;               /* tex16 | half transmission depth << 16 */
    uint32_t mr_tex_spec_roughness_coat_roughness;
    uint32_t emissive_factor_normal_scale;      /* RGB8 | UNORM8 normal scale << 24 */
    uint32_t emissive_strength_ior;             /* half strength | UNORM16 ior << 16 */
    uint32_t emissive_tex_alpha_cutoff_coat_ior;
    uint32_t coat_color_flags;                  /* RGB8 | flag bits 24.. */
} zr_material;

/* Material flag bits, reference Material.h:31-39 */
#define ZR_MAT_METALLIC_BIT      24
#define ZR_MAT_DOUBLE_SIDED_BIT  25
#define ZR_MAT_TRANSMISSIVE_BIT  26
#define ZR_MAT_ALPHA_1_BIT       27
#define ZR_MAT_ALPHA_2_BIT       28
#define ZR_MAT_THIN_WALLED_BIT   29
#define ZR_INVALID_TEX           0xffffu

/* reference: RtCommon.h:66-131 with ENCODE_EMISSIVE_POS=1, EMISSIVE_UV_HALF=1 (48 B) */
typedef struct zr_emissive_triangle {
    float    vtx0[3];
    uint16_t v0v1[2];        /* oct-encoded unit edge, UNORM16 x2 */
    uint16_t v0v2[2];
    uint16_t edge_lengths[2];/* half2 */
    uint32_t id;             /* PCG3d(geometryIndex, instanceID, primIdx).x */
    uint32_t packed_a;       /* RGB8 emissive factor | bit24 id-patched | bit25 double sided */
    uint32_t packed_b;       /* tex16 | half strength << 16 */
    uint16_t uv0[2];         /* half2 */
    uint16_t uv1[2];
    uint16_t uv2[2];
} zr_emissive_triangle;

/* reference: RtCommon.h:302-310 (16 B) */
typedef struct zr_alias_entry {
    float    cached_p_orig;
    float    cached_p_alias;
    float    p_curr;
    uint32_t alias;
} zr_alias_entry;

/* reference: RtCommon.h:312-322 (40 B) */
typedef struct zr_presampled_tri {
    float    pos[3];
    uint16_t normal[2];
    float    pdf;
    uint32_t id;
    uint32_t idx;
    uint16_t bary[2];
    uint16_t le[3];      /* half3 */
    uint16_t two_sided;
} zr_presampled_tri;

/* reference: RtCommon.h:324-332 (32 B): one of the 64 light samples of a light-voxel-grid voxel */
typedef struct zr_voxel_sample {
    float    pos[3];
    uint16_t normal[2];
    float    pdf;
    uint32_t id;
    uint16_t le[3];      /* half3 */
    uint16_t two_sided;
} zr_voxel_sample;
#define ZR_LVG_SAMPLES_PER_VOXEL 64u   /* NUM_SAMPLES_PER_VOXEL, PreLighting_Common.h:14 */

/* reference: Source/ZetaRenderPass/Common/FrameConstants.h:10-78 (544 B).  Row-major 3x4 matrices. */
typedef struct zr_frame_constants {
    float curr_view[12];
    float prev_view[12];
    float curr_view_inv[12];
    float prev_view_inv[12];
    float curr_view_proj[16];
    float prev_view_proj[16];

    float camera_pos[3];
    float camera_near;

    float aspect_ratio;
    float pixel_spread_angle;
    float tan_half_fov;
    float dt;

    uint32_t frame_num;
    uint32_t curr_gbuffer_desc_heap_offset;  /* unused by the HIP path (no descriptor heap); kept for layout */
    uint32_t prev_gbuffer_desc_heap_offset;
    uint32_t base_color_maps_desc_heap_offset;

    uint32_t normal_maps_desc_heap_offset;
    uint32_t metallic_roughness_maps_desc_heap_offset;
    uint32_t emissive_maps_desc_heap_offset;
    uint32_t env_map_desc_heap_offset;

    uint32_t render_width;
    uint32_t render_height;
    uint32_t display_width;
    uint32_t display_height;

    float curr_camera_jitter[2];
    float prev_camera_jitter[2];

    float planet_radius;
    float sun_cos_angular_radius;
    float sun_sin_angular_radius;
    float pad;

    float sun_dir[3];
    float sun_illuminance;

    float rayleigh_sigma_s_color[3];
    float rayleigh_sigma_s_scale;

    float ozone_sigma_a_color[3];
    float ozone_sigma_a_scale;

    float mie_sigma_s;
    float mie_sigma_a;
    float atmosphere_altitude;
    float g;

    uint32_t num_frames_camera_static;
    uint32_t camera_static;
    uint32_t accumulate;
    uint32_t sun_moved;

    float    camera_ray_uv_grads_scale;
    float    mip_bias;
    float    one_div_num_emissive_triangles;
    uint32_t num_emissive_triangles;

    float    focus_depth;
    float    lens_radius;
    uint32_t dof;
    uint32_t pad2;
} zr_frame_constants;

/* RT_AS_SUBGROUP, reference RtCommon.h:34-39 */
#define ZR_SUBGROUP_EMISSIVE      0x1u
#define ZR_SUBGROUP_NON_EMISSIVE  0x2u
#define ZR_SUBGROUP_ALL           0x3u
/* Extra bit of zr_scene_desc.instance_mask: the instance's BLAS geometry is built WITHOUT
   D3D12_RAYTRACING_GEOMETRY_FLAG_OPAQUE (RtAccelerationStructure.cpp:155-158, RT_Flags::IsOpaque == false), so primary
   rays run the alpha test of GBufferRT_Inline.hlsl:37-70 on its triangles.  All other rays force opaque
   (RayQuery.hlsli:42,168,317-319,372-374,382-383) and ignore the bit. */
#define ZR_INSTANCE_NON_OPAQUE    0x80u

/*
 * Material textures.  The reference binds BC-compressed DDS textures (BC7_UNORM_SRGB base colour / emissive, BC5_UNORM
 * normal / metallic-roughness, Assets.cpp + Tools/BCnCompressglTF) through four descriptor tables and filters them in
 * hardware.  Neither block decompression nor hardware filtering is pinned by the reference, so this ABI takes the decoded
 * texels: every texture is a full mip chain of uncompressed texels (mip m is max(1, w >> m) x max(1, h >> m), rows
 * top-down, mips packed back to back from `offset`, which must be a multiple of 4), and zr_texture.h defines the
 * filtering.  Material / MeshInstance / EmissiveTriangle texture indices address `textures` exactly like the reference
 * addresses its descriptor heap: textures[frame_constants.<kind>_maps_desc_heap_offset + tex16].
 */
#define ZR_TEX_RGBA8_SRGB  0u   /* 4 B/texel, RGB through the sRGB transfer function, A linear (base colour, emissive) */
#define ZR_TEX_RGBA8       1u   /* 4 B/texel, UNORM */
#define ZR_TEX_RG8         2u   /* 2 B/texel, UNORM (normal XY, metallic-roughness); samples return (r, g, 0, 1) */
typedef struct zr_texture_desc {
    uint64_t offset;     /* byte offset of mip 0 inside zr_scene_desc.texels */
    uint16_t width;
    uint16_t height;
    uint8_t  num_mips;   /* >= 1 */
    uint8_t  format;     /* ZR_TEX_* */
    uint16_t pad;
} zr_texture_desc;

/* G-buffer flag byte, reference Source/ZetaRenderPass/Common/GBuffers.hlsli:52-87 */
#define ZR_GBUF_TRANSMISSIVE  (1u << 0)
#define ZR_GBUF_EMISSIVE      (1u << 1)
#define ZR_GBUF_INVALID       (1u << 2)
#define ZR_GBUF_TRDEPTH_GT0   (1u << 3)
#define ZR_GBUF_SUBSURFACE    (1u << 4)
#define ZR_GBUF_COATED        (1u << 5)
#define ZR_GBUF_METALLIC      (1u << 7)

/*
 * Scene description handed to zr_scene_create (and to the oracle).  All arrays are host pointers in the wire
 * formats above -- the same buffers the reference publishes through SharedShaderResources under the names in
 * Source/ZetaCore/Scene/SceneRenderer.h:17-32 (SceneVB, SceneIB, MaterialBuffer, RtFrameMeshInstances,
 * EmissiveTriangles).  `instance_to_world` is the exact float 3x4 row-major object-to-world matrix per mesh
 * instance that the reference feeds to the BLAS build (RtAccelerationStructure.cpp:121-200); geometry for traversal
 * is built from it, shading re-derives positions from the quantised MeshInstance exactly as the reference shaders do.
 */
typedef struct zr_scene_desc {
    const zr_vertex*            vertices;        uint32_t num_vertices;
    const uint32_t*             indices;         uint32_t num_indices;
    const zr_mesh_instance*     instances;       uint32_t num_instances;
    const float*                instance_to_world;   /* num_instances x 12 floats */
    const uint8_t*              instance_mask;       /* num_instances, ZR_SUBGROUP_* */
    const uint32_t*             instance_num_tris;   /* num_instances: triangle count of each instance's mesh */
    const zr_material*          materials;       uint32_t num_materials;
    const zr_emissive_triangle* emissives;       uint32_t num_emissives;
    /* rho.dds payload: R16_UNORM, 64 x 32 x 16 (reference BSDF.hlsli:279-296, Assets/LUT/rho.dds) */
    const uint16_t*             rho_lut;         uint32_t rho_dim[3];
    /* material textures (may be null / 0: every tex16 of the scene must then be ZR_INVALID_TEX) */
    const zr_texture_desc*      textures;        uint32_t num_textures;
    const uint8_t*              texels;          uint64_t texel_bytes;
} zr_scene_desc;

#ifdef __cplusplus
}
#endif

#endif /* ZR_WIRE_H */
