"""numpy dtypes / ctypes structs mirroring include/zr_wire.h and include/zetaray_amd.h (bit-exact layouts).

Host-side plumbing only: these are the buffers a ZetaRay-style caller already owns
(reference: Source/ZetaCore/RayTracing/RtCommon.h:47-332, Source/ZetaCore/Core/Material.h:419-427,
Source/ZetaCore/Core/Vertex.h:8-14, Source/ZetaRenderPass/Common/FrameConstants.h:10-78).
"""
import ctypes as C

import numpy as np

VERTEX = np.dtype([("pos", "<f4", 3), ("uv", "<f4", 2), ("normal", "<u2", 2), ("tangent", "<u2", 2)])
assert VERTEX.itemsize == 28

MESH_INSTANCE = np.dtype([
    ("base_vtx_offset", "<u4"), ("base_idx_offset", "<u4"), ("rotation", "<u2", 4), ("scale", "<u2", 3),
    ("mat_idx", "<u2"), ("base_emissive_tri_offset", "<u4"), ("translation", "<f4", 3),
    ("prev_rotation", "<u2", 4), ("prev_scale", "<u2", 3), ("d_translation", "<u2", 3),
    ("base_color_tex", "<u2"), ("alpha_factor_cutoff", "<u2")])
assert MESH_INSTANCE.itemsize == 64

MATERIAL = np.dtype([
    ("base_color_factor", "<u4"), ("base_color_tex_subsurf_coat_weight", "<u4"), ("normal_tex_tr_depth", "<u4"),
    ("mr_tex_spec_roughness_coat_roughness", "<u4"), ("emissive_factor_normal_scale", "<u4"),
    ("emissive_strength_ior", "<u4"), ("emissive_tex_alpha_cutoff_coat_ior", "<u4"), ("coat_color_flags", "<u4")])
assert MATERIAL.itemsize == 32

EMISSIVE_TRI = np.dtype([
    ("vtx0", "<f4", 3), ("v0v1", "<u2", 2), ("v0v2", "<u2", 2), ("edge_lengths", "<u2", 2), ("id", "<u4"),
    ("packed_a", "<u4"), ("packed_b", "<u4"), ("uv0", "<u2", 2), ("uv1", "<u2", 2), ("uv2", "<u2", 2)])
assert EMISSIVE_TRI.itemsize == 48

ALIAS_ENTRY = np.dtype([("cached_p_orig", "<f4"), ("cached_p_alias", "<f4"), ("p_curr", "<f4"), ("alias", "<u4")])
assert ALIAS_ENTRY.itemsize == 16

PRESAMPLED_TRI = np.dtype([("pos", "<f4", 3), ("normal", "<u2", 2), ("pdf", "<f4"), ("id", "<u4"), ("idx", "<u4"), ("bary", "<u2", 2),
                           ("le", "<u2", 3), ("two_sided", "<u2")])
assert PRESAMPLED_TRI.itemsize == 40
VOXEL_SAMPLE = np.dtype([("pos", "<f4", 3), ("normal", "<u2", 2), ("pdf", "<f4"), ("id", "<u4"), ("le", "<u2", 3), ("two_sided", "<u2")])
assert VOXEL_SAMPLE.itemsize == 32

FRAME_CONSTANTS = np.dtype([
    ("curr_view", "<f4", 12), ("prev_view", "<f4", 12), ("curr_view_inv", "<f4", 12), ("prev_view_inv", "<f4", 12),
    ("curr_view_proj", "<f4", 16), ("prev_view_proj", "<f4", 16),
    ("camera_pos", "<f4", 3), ("camera_near", "<f4"),
    ("aspect_ratio", "<f4"), ("pixel_spread_angle", "<f4"), ("tan_half_fov", "<f4"), ("dt", "<f4"),
    ("frame_num", "<u4"), ("curr_gbuffer_desc_heap_offset", "<u4"), ("prev_gbuffer_desc_heap_offset", "<u4"),
    ("base_color_maps_desc_heap_offset", "<u4"),
    ("normal_maps_desc_heap_offset", "<u4"), ("metallic_roughness_maps_desc_heap_offset", "<u4"),
    ("emissive_maps_desc_heap_offset", "<u4"), ("env_map_desc_heap_offset", "<u4"),
    ("render_width", "<u4"), ("render_height", "<u4"), ("display_width", "<u4"), ("display_height", "<u4"),
    ("curr_camera_jitter", "<f4", 2), ("prev_camera_jitter", "<f4", 2),
    ("planet_radius", "<f4"), ("sun_cos_angular_radius", "<f4"), ("sun_sin_angular_radius", "<f4"), ("pad", "<f4"),
    ("sun_dir", "<f4", 3), ("sun_illuminance", "<f4"),
    ("rayleigh_sigma_s_color", "<f4", 3), ("rayleigh_sigma_s_scale", "<f4"),
    ("ozone_sigma_a_color", "<f4", 3), ("ozone_sigma_a_scale", "<f4"),
    ("mie_sigma_s", "<f4"), ("mie_sigma_a", "<f4"), ("atmosphere_altitude", "<f4"), ("g", "<f4"),
    ("num_frames_camera_static", "<u4"), ("camera_static", "<u4"), ("accumulate", "<u4"), ("sun_moved", "<u4"),
    ("camera_ray_uv_grads_scale", "<f4"), ("mip_bias", "<f4"), ("one_div_num_emissive_triangles", "<f4"),
    ("num_emissive_triangles", "<u4"),
    ("focus_depth", "<f4"), ("lens_radius", "<f4"), ("dof", "<u4"), ("pad2", "<u4")])
assert FRAME_CONSTANTS.itemsize == 544

SUBGROUP_EMISSIVE = 1
SUBGROUP_NON_EMISSIVE = 2
SUBGROUP_ALL = 3
INSTANCE_NON_OPAQUE = 0x80      # extra bit of instance_mask: geometry without the OPAQUE flag (primary-ray alpha test)
TEX_RGBA8_SRGB, TEX_RGBA8, TEX_RG8 = 0, 1, 2
TEXTURE_DESC = np.dtype([("offset", "<u8"), ("width", "<u2"), ("height", "<u2"), ("num_mips", "u1"), ("format", "u1"), ("pad", "<u2")])

GB_PLANE_NAMES = ["base_color", "normal", "metallic_roughness", "motion_vector", "emissive_color", "ior", "coat",
                  "depth", "tri_diff_geo_a", "tri_diff_geo_b"]
GB_PLANE_BYTES = [4, 4, 2, 4, 4, 1, 8, 4, 16, 8]
GB_PLANE_DTYPES = [("<u4", 1), ("<u4", 1), ("<u2", 1), ("<u4", 1), ("<u4", 1), ("u1", 1), ("<u2", 4), ("<f4", 1),
                   ("<u4", 4), ("<u4", 2)]
GB_COUNT = 10


class SceneDesc(C.Structure):
    _fields_ = [
        ("vertices", C.c_void_p), ("num_vertices", C.c_uint32),
        ("indices", C.c_void_p), ("num_indices", C.c_uint32),
        ("instances", C.c_void_p), ("num_instances", C.c_uint32),
        ("instance_to_world", C.c_void_p),
        ("instance_mask", C.c_void_p),
        ("instance_num_tris", C.c_void_p),
        ("materials", C.c_void_p), ("num_materials", C.c_uint32),
        ("emissives", C.c_void_p), ("num_emissives", C.c_uint32),
        ("rho_lut", C.c_void_p), ("rho_dim", C.c_uint32 * 3),
        ("textures", C.c_void_p), ("num_textures", C.c_uint32),
        ("texels", C.c_void_p), ("texel_bytes", C.c_uint64),
    ]


class GBufferPlanes(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("plane", C.c_void_p * GB_COUNT)]


class Params(C.Structure):
    _fields_ = [
        ("flags", C.c_uint32), ("max_non_tr_bounces", C.c_uint32), ("max_glossy_tr_bounces", C.c_uint32),
        ("m_max_temporal", C.c_uint32), ("m_max_spatial", C.c_uint32), ("alpha_min", C.c_float),
        ("presampling", C.c_uint32), ("num_sample_sets", C.c_uint32), ("sample_set_size", C.c_uint32),
        ("use_lvg", C.c_uint32), ("lvg_grid_dim", C.c_uint32), ("lvg_extents", C.c_float * 3), ("lvg_offset_y", C.c_float),
        ("taa_blend_weight", C.c_float),
        ("ae_min_lum", C.c_float), ("ae_max_lum", C.c_float), ("ae_lum_map_exp", C.c_float), ("ae_adaptation_rate", C.c_float),
        ("display_tonemapper", C.c_uint32), ("display_auto_exposure", C.c_uint32), ("display_saturation", C.c_float),
        ("display_agx_exp", C.c_float), ("tex_filter", C.c_uint32),
        ("svgf_alpha", C.c_float), ("svgf_alpha_moments", C.c_float), ("svgf_sigma_l", C.c_float), ("svgf_sigma_z", C.c_float),
        ("svgf_normal_power_log2", C.c_uint32), ("svgf_iterations", C.c_uint32), ("num_spatial_passes", C.c_uint32)]


class Counters(C.Structure):
    _fields_ = [("n_closest", C.c_uint64), ("n_shadow", C.c_uint64)]


IND_TEMPORAL_RESAMPLE = 1 << 0
IND_SPATIAL_RESAMPLE = 1 << 1
IND_STOCHASTIC_MULTI_BOUNCE = 1 << 2
IND_RUSSIAN_ROULETTE = 1 << 3
IND_BOILING_SUPPRESSION = 1 << 4
IND_PATH_REGULARIZATION = 1 << 5
IND_SORT_TEMPORAL = 1 << 6
IND_SORT_SPATIAL = 1 << 7


def default_params() -> Params:
    """Reference defaults: IndirectLighting.h:231-244, IndirectLighting.cpp:146-165."""
    p = Params()
    p.flags = (IND_TEMPORAL_RESAMPLE | IND_SPATIAL_RESAMPLE | IND_RUSSIAN_ROULETTE | IND_BOILING_SUPPRESSION |
               IND_SORT_TEMPORAL | IND_SORT_SPATIAL)
    p.max_non_tr_bounces = 3
    p.max_glossy_tr_bounces = 4
    p.m_max_temporal = 10
    p.m_max_spatial = 8
    p.alpha_min = 0.175 * 0.175
    p.presampling = 0
    p.num_sample_sets = 128
    p.sample_set_size = 512
    p.use_lvg = 0
    p.lvg_grid_dim = 32 | (8 << 10) | (40 << 20)
    p.lvg_extents[:] = (0.6, 0.45, 0.6)
    p.lvg_offset_y = 0.1
    p.taa_blend_weight = 0.1
    set_post_defaults(p)
    return p


TEX_FILTER_MIP0, TEX_FILTER_TRI_LINEAR, TEX_FILTER_ANISOTROPIC_2X, TEX_FILTER_ANISOTROPIC_4X, TEX_FILTER_ANISOTROPIC_16X = range(5)
TONEMAP_NONE, TONEMAP_NEUTRAL, TONEMAP_AGX_DEFAULT, TONEMAP_AGX_GOLDEN, TONEMAP_AGX_PUNCHY, TONEMAP_AGX_CUSTOM = range(6)


def set_post_defaults(p):
    """AutoExposure.h:73-81 and Display.cpp:69-74"""
    p.ae_min_lum, p.ae_max_lum, p.ae_lum_map_exp, p.ae_adaptation_rate = 5e-3, 4.0, 0.5, 1.0
    p.display_tonemapper, p.display_auto_exposure, p.display_saturation, p.display_agx_exp = TONEMAP_NEUTRAL, 1, 1.0, 1.0
    p.tex_filter = TEX_FILTER_ANISOTROPIC_4X      # IndirectLighting.h:243
    # ZR_PASS_DENOISE (no reference counterpart)
    p.svgf_alpha, p.svgf_alpha_moments, p.svgf_sigma_l, p.svgf_sigma_z, p.svgf_normal_power_log2, p.svgf_iterations = 0.2, 0.2, 4.0, 1.0, 7, 5
    p.num_spatial_passes = 1          # IndirectLighting.h:392


COMPOSIT_FIREFLY_FILTER = 1 << 10
DI_STOCHASTIC_SPATIAL = 1 << 8
DI_EXTRA_DISOCCLUSION_SAMPLING = 1 << 9
DI_HALF_VECTOR_COPY_SHIFT = 1 << 11      # the reference's compile-time USE_HALF_VECTOR_COPY_SHIFT (Emissive/Params.hlsli:12) as a run-time flag of the emissive DI pass


def default_params_sky_di():
    """SkyDI defaults (SkyDI.cpp:81-82, SkyDI.h:86-92): M_max sky 15 (m_max_temporal), M_max sun 3 (m_max_spatial), alpha_min 0.35^2"""
    p = default_params()
    p.flags = 0x3          # TEMPORAL_RESAMPLE | SPATIAL_RESAMPLE
    p.m_max_temporal = 15
    p.m_max_spatial = 3
    p.alpha_min = float(np.float32(0.35) * np.float32(0.35))
    return p


def default_params_di() -> Params:
    """ReSTIR DI defaults: DirectLighting.cpp:100-107, DirectLighting.h:93-98 (M_max 20, alpha_min 0.05^2)."""
    p = Params()
    p.flags = IND_TEMPORAL_RESAMPLE | IND_SPATIAL_RESAMPLE | DI_STOCHASTIC_SPATIAL | DI_EXTRA_DISOCCLUSION_SAMPLING
    p.max_non_tr_bounces, p.max_glossy_tr_bounces = 1, 1
    p.m_max_temporal = 20
    p.m_max_spatial = 20
    p.alpha_min = 0.05 * 0.05
    p.presampling = 0
    p.num_sample_sets = 128
    p.sample_set_size = 512
    return p


def alloc_gbuffer_planes(width: int, height: int):
    """Host planes (numpy) + the ctypes view handed to the oracle / zr_gbuffer_download."""
    arrays = []
    planes = GBufferPlanes()
    planes.width, planes.height = width, height
    for i, (dt, n) in enumerate(GB_PLANE_DTYPES):
        shape = (height, width) if n == 1 else (height, width, n)
        a = np.zeros(shape, dtype=dt)
        arrays.append(a)
        planes.plane[i] = a.ctypes.data
    return arrays, planes
