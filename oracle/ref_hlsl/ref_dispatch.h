// ORACLE tooling -- test infrastructure only.  One Dispatch() of a reference shader compiled as C++ (ref_pass_shader.cpp).
#pragma once
#include <stdint.h>
typedef struct ZrDispatch
{
    void* scene;              /* refpass::RefScene*: geometry, lights, rho LUT, textures */
    void* prev_scene;         /* previous frame's scene (RT_SCENE_BVH_PREV / RT_FRAME_MESH_INSTANCES_PREV), or null = same */
    int   use_prev_scene;     /* the pass binds the PREVIOUS acceleration structure + mesh instances (CtT replay / reconnect, DI temporal) */
    void* heap;               /* hlsl::DescriptorHeap*: every texture / UAV the constant buffers index */
    const void* frame_cb;     /* cbFrameConstants, 544 B */
    const void* local_cb; uint32_t local_cb_bytes;
    uint32_t groups_x, groups_y;
    void* root_uav;           /* a root-descriptor UAV bound as a global (AutoExposure's g_hist : register(u0)), or null */
    /* ref_pass_aux.cpp (PreLighting / Sky / Compositing / TAA shaders): root SRV / UAV buffers by role, element counts, z extent of the dispatch */
    void* buf[4]; uint32_t buf_count[4];
    uint32_t groups_z;        /* 0 = 1 */
} ZrDispatch;
