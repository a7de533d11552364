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
    float kappa = 0.0f;           // half-angle of the cone a traversal must give its rays (scene_kappa)
};

// How far outside an object's exact volume the reference's fp32 hit arithmetic can still report a hit,
// as a fraction of the distance between ray origin and object: 2^-10 for spheres, cylinders and rects
// (the discriminant of sphere.h:85-101 carries up to 15 * 2^-24 |oc|^2 |d|^2 of rounding error, i.e.
// sqrt(15 * 2^-24) = 9.5e-4 |oc| in position; the largest displacement seen in stress runs is between
// a quarter and a half of that).  A cone's quadratic (cone.h:101-112) mixes in tangens-weighted y
// terms and cancels in its leading coefficient; no closed bound is derived for it, and scenes with
// cones get 2^-9 * max(1, radius/height), capped at 2^-4 (stress runs need less than 2^-12).
float scene_kappa(const pt_scene_desc &desc);

// Bounding volume of one object as the reference defines it (sphere.h:112-117, aarect.h:45-50,
// 144-149, 229-234, cone.h:142-147, cylinder.h:133-138), min/max-normalised and padded so that
// rounding in the reference's own hit arithmetic cannot place an accepted hit outside it: by
// 2^-17 of the largest coordinate, and for the quadrics by kappa times the object's own extent (the
// part of the origin-to-object distance that the traversal's ray cone, which grows with t, cannot see).
bool prim_box(uint8_t type, const float *param, float kappa, Box &out);

// Binned SAH (16 bins per axis), leaves of at most leaf_max objects.  Returns "" or an error text.
std::string build_bvh(const pt_scene_desc &desc, uint32_t leaf_max, BvhBuild &out);

} // namespace pt
