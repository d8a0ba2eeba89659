// zr_scene_io.h -- scene ingestion on the host, in C++: glTF 2.0 -> the wire formats of include/zr_wire.h.
//
// Mirrors the caller side of the hot path in the reference:
//   Source/ZetaCore/Model/glTF.cpp:270-431   mesh primitives: positions / normals / tangents / UVs, RH -> LH (z flip, winding swap)
//   Source/ZetaCore/Model/glTF.cpp:523-643   materials (pbrMetallicRoughness + KHR_materials_emissive_strength / ior / transmission / clearcoat)
//   Source/ZetaCore/Model/glTF.cpp:692-767   emissive instances and triangles (RT::EmissiveTriangle, RtCommon.h:73-190)
//   Source/ZetaCore/Scene/SceneCore.cpp:196-236, 860-900   emissive triangle IDs (PCG3d hash), world = local x parent
//   Source/ZetaCore/RayTracing/RtAccelerationStructure.cpp:318-380   MeshInstance quantisation: decomposeSRT (Math/MatrixFuncs.h:562-610) of
//       the world matrix -> unorm4 rotation, half3 scale, float3 translation
//   Assets.cpp / Tools/BCnCompressglTF       DDS material textures; here BC7 / BC5 / RGBA8 DDS files are decoded once at load to the texel
//       heap of zr_scene_desc (zr_wire.h: the ABI takes decoded texels and defines the filtering)
// No HIP dependency: the library loads on machines without a GPU.
#pragma once
#include <cstddef>
#include <cstdint>
#include "../../include/zr_wire.h"

extern "C" {
typedef struct zrh_scene_data zrh_scene_data;      // owns every array its zr_scene_desc points to

// Loads <path>.gltf (+ its external .bin buffers and .dds images).  rho_lut / rho_dim: the GGX reflectance LUT to attach (Assets/LUT/rho.dds
// payload; copied).  Returns 0 and a handle, or -1 (zrh_scene_io_last_error()).
int zrh_gltf_load(const char* path, const uint16_t* rho_lut, const uint32_t* rho_dim3, zrh_scene_data** out);
const char* zrh_scene_io_last_error(void);
const zr_scene_desc* zrh_scene_data_desc(const zrh_scene_data* s);
// the four descriptor-table offsets for cbFrameConstants (base colour, normal, metallic-roughness, emissive maps) of the scene's texture heap
void zrh_scene_data_tex_offsets(const zrh_scene_data* s, uint32_t* out4);
void zrh_scene_data_destroy(zrh_scene_data* s);

// ---- per-frame maintenance of a loaded scene (SceneCore::Update + TLAS::FillMeshInstanceData for dynamic instances, RtAccelerationStructure.cpp:318-380;
// SceneCore::UpdateEmissivePositions, SceneCore.cpp:913-955).  Per frame: begin_frame (this frame's matrices become the previous ones, every record
// turns static: Prev* = current, dTranslation = 0), set_instance_world for each instance that moves (current + previous S / R / T by decomposeSRT,
// dTranslation = half3(t - t_prev); the EmissiveTriangle records of a light-carrying instance re-derived from the object-space ones), then hand
// zrh_scene_data_desc's instances / instance_to_world and the dirty emissive range to zr_scene_update_emissives / zr_scene_update_instances
// (zrh_scene_apply_updates in zr_host.h does both calls).
void zrh_scene_data_begin_frame(zrh_scene_data* s);
int zrh_scene_data_set_instance_world(zrh_scene_data* s, uint32_t instance, const float* world_3x4);
void zrh_scene_data_dirty_emissives(const zrh_scene_data* s, uint32_t* first, uint32_t* count);

// ---- building blocks, exported for the parity pins (tests/test_scene_io.py) ----
// decomposeSRT + quaternionFromRotationMat1 of a 3 x 4 row-major object-to-world matrix (column-vector convention, zr_scene_desc.instance_to_world)
void zrh_decompose_srt(const float* to_world_3x4, float* scale3, float* quat4, float* translation3);
// affineTransformation(s, q, t) x parent (both 3 x 4 row-major, column-vector convention; parent may be null = identity)
void zrh_compose_world(const float* scale3, const float* quat4, const float* translation3, const float* parent_3x4, float* out_3x4);
// TLAS::FillMeshInstanceData for a static instance: rotation / scale / translation (+ prev_* = current, d_translation = 0)
void zrh_fill_mesh_instance(const float* to_world_3x4, zr_mesh_instance* inst);
// RT::EmissiveTriangle(v0, v1, v2, uv0..2, factor RGB8, texture, half strength bits, id, double sided)
void zrh_pack_emissive_triangle(const float* v0, const float* v1, const float* v2, const float* uv6, uint32_t factor_rgb8, uint32_t tex,
                                uint16_t strength_half, uint32_t id, int double_sided, zr_emissive_triangle* out);
// SceneCore's emissive transform (SceneCore.cpp:196-236, UpdateEmissivePositions :913-955): decode the triangle's vertices (16-bit octahedral edge
// directions, half lengths), transform them by the 3 x 4 object-to-world matrix, encode again; every other field is kept.  Per frame, for a moving
// emissive instance: out[t] = zrh_emissive_to_world(initial[t], new matrix) for its triangles, then zr_scene_update_emissives (zetaray_amd.h)
void zrh_emissive_to_world(const zr_emissive_triangle* in, const float* to_world_3x4, zr_emissive_triangle* out);
// the object-space records of the scene's emissive triangles as loaded (same order as zr_scene_desc.emissives; ID already hashed)
const zr_emissive_triangle* zrh_scene_data_initial_emissives(const zrh_scene_data* s);
// BC7 / BC5 block decompression: w x h texels (multiples of 4 not required) -> RGBA8 / RG8 rows top-down
int zrh_bc7_decode(const uint8_t* blocks, uint32_t w, uint32_t h, uint8_t* rgba8);
int zrh_bc5_decode(const uint8_t* blocks, uint32_t w, uint32_t h, uint8_t* rg8);
}
