// pt_bvh.h -- host-side BVH construction over the reference's bounding_box() volumes.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/pt_api.h"

namespace pt {

struct HostNode { // mirrors BvhNode (pt_device.h), 64 B: child boxes interleaved per axis
    float lo[3][2];
    float hi[3][2];
    int32_t c[2];
    uint32_t pad[2];
};
static_assert(sizeof(HostNode) == 64, "HostNode must be 64 bytes");

struct Box {
    float lo[3], hi[3];
};

struct BvhBuild {
    std::vector<HostNode> nodes;  // nodes[0] is the root
    std::vector<uint32_t> order;  // list index of the object stored at each leaf slot
    uint32_t max_depth = 0;       // inner nodes on the longest root-to-leaf path
};

// Bounding volume of one object as the reference defines it (sphere.h:112-117, aarect.h:45-50,
// 144-149, 229-234, cone.h:142-147, cylinder.h:133-138), min/max-normalised and padded so that
// rounding in the reference's own hit arithmetic cannot place an accepted hit outside it.
bool prim_box(uint8_t type, const float *param, Box &out);

// Binned SAH (16 bins per axis), leaves of at most leaf_max objects.  Returns "" or an error text.
std::string build_bvh(const pt_scene_desc &desc, uint32_t leaf_max, BvhBuild &out);

} // namespace pt
