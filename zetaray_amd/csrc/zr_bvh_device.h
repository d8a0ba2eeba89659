// zr_bvh_device.h -- interface of the device-side BVH build (zr_tu_bvh.hip): device pointers in, a ready-to-refit BVH4 topology out.
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include <vector>
#include "zr_dev_scene.h"

namespace zr {

struct DeviceBvhInputs
{
    uint32_t numTris;
    const TriMeta* meta;                    // global order: (mesh = GeometryIndex + InstanceID, primitive)
    const zr_mesh_instance* instances; const float* toWorld; const zr_vertex* vertices; const uint32_t* indices;
    const uint8_t* instanceMask;            // ZR_SUBGROUP_* / ZR_INSTANCE_NON_OPAQUE per instance
};
struct DeviceBvhOutputs
{
    BvhTri* tris;                           // numTris, leaf (= sorted) order
    Bvh4Node* nodes; uint32_t nodeCap;      // children set; boxes are computed by the caller's per-level refit (bottom-up)
    uint32_t numNodes = 0, numLevels = 0, stackNeed = 0;
    std::vector<uint32_t> levelOrder, levelOffsets;      // node ids grouped by level, deepest first (what the refit walks)
};
// temporary device buffers, kept between builds of the same scene
struct DeviceBvhScratch
{
    struct Impl; Impl* impl;
    DeviceBvhScratch(); ~DeviceBvhScratch();
    DeviceBvhScratch(const DeviceBvhScratch&) = delete; DeviceBvhScratch& operator=(const DeviceBvhScratch&) = delete;
};
// enqueues the build on `st` and waits for it (the level structure comes back to the host); 0 = ok, else `err` says why
int DeviceBuildBvh4(hipStream_t st, DeviceBvhScratch& scratch, const DeviceBvhInputs& in, DeviceBvhOutputs& out, std::string& err);
// test hook: the deepest tree the builder may produce, in levels (2 .. 21; anything else restores the default 21 = what the traversal stack holds)
void DeviceBvhSetDepthCap(uint32_t levels);

} // namespace zr
