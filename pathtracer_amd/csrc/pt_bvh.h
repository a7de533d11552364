// pt_bvh.h -- host-side BVH construction over the reference's bounding_box() volumes.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/pt_api.h"

namespace pt {

struct HostNode { // mirrors BvhNode (pt_device.h), 64 B: per axis (lo child 0, lo child 1, hi child 0, hi child 1)
    float b[3][4];
    int32_t c[2];
    uint32_t pad[2];
};
static_assert(sizeof(HostNode) == 64, "HostNode must be 64 bytes");

struct Box {
    float lo[3], hi[3];
};

// How wide a cone the traversal's slab test gives a ray (make_box_ray in pt_trace.h).
//
// The reference's fp32 hit arithmetic accepts hits outside an object's exact volume.  For a sphere at
// distance |oc| from the ray origin the discriminant of sphere.h:85-101 carries up to
// 15 * 2^-24 |oc|^2 |d|^2 of rounding error, and the accepted point lies within sqrt(r^2 + eta^2) of
// the centre, eta = k |oc|, k = sqrt(15 * 2^-24) = 9.5e-4 (the largest displacement seen in stress
// runs is between a quarter and a half of that); the cylinder's two-term quadratic (cylinder.h:96-123)
// stays below the same bound, rects only see rounding of order 2^-23 (|o| + t|d|).  The excess over
// the radius, f_r(D) = sqrt(r^2 + k^2 D^2) - r, is convex in the distance D with f_r(0) = 0 and falls
// with r.  So for a ray whose origin is at most Dm away from every object that relies on the cone, a
// cone of slope f_rmin(Dm) / Dm covers them all (chord above a convex function; Dm is taken in the
// L1 norm, which only widens it): rays inside a
// cluster of fair-sized spheres get a cone far narrower than k, rays crossing a field of tiny distant
// spheres get nearly k.  Objects whose excess stays below 1/256 of their radius even across the whole
// scene (a ground sphere of radius 1000) get that excess padded onto their box instead and do not
// enter r_min or the cluster bounds.
//
// A cone's quadratic (cone.h:101-112) mixes in tangens-weighted y terms and cancels in its leading
// coefficient (worst-case analysis gives sqrt(31 * 2^-24 / psi) of the tangens-weighted distance, psi the
// surviving fraction of that coefficient: no useful closed bound).  Scenes with cones use a fixed slope
// of 2^-10 * max(1, radius/height), capped at 2^-4; stress runs need less than a quarter of it.
struct ConeParams {
    float kappa_max = 0.0f;   // slope that is sufficient for any ray
    float kappa_min = 0.0f;   // rounding-level slope every ray gets at least
    float r_min = 1e30f;      // smallest radius among the quadrics that rely on the cone
    float qc[3] = {0, 0, 0};  // centre of the bounds of those quadrics ...
    float qh = 0.0f;          // ... and the sum of the bounds' half extents (L1 reach from the centre)
    uint32_t adaptive = 0;    // 0: every ray uses kappa_max
    // The reduced slope is valid for ray origins no farther than `reach` from any object; a render's
    // origins (camera, lens, points on the objects) lie within elo..ehi, which guarantees it.  Rays
    // handed in from outside (pt_trace_rays) are not checked against it and use kappa_max.
    float elo[3] = {0, 0, 0}, ehi[3] = {0, 0, 0};
};

struct BvhBuild {
    std::vector<HostNode> nodes;  // nodes[0] is the root
    std::vector<uint32_t> order;  // list index of the object stored at each leaf slot
    uint32_t max_depth = 0;       // inner nodes on the longest root-to-leaf path
    ConeParams cone;
};

// Bounding volume of one object as the reference defines it (sphere.h:112-117, aarect.h:45-50,
// 144-149, 229-234, cone.h:142-147, cylinder.h:133-138), min/max-normalised and padded so that
// rounding in the reference's own hit arithmetic cannot place an accepted hit outside it: by
// 2^-17 of the largest coordinate, by `extra`, and for the quadrics by kappa times the object's own
// extent (the part of the origin-to-object distance that a cone growing with t cannot see).
bool prim_box(uint8_t type, const float *param, float kappa, float extra, Box &out);

// Four-wide node with quantised child boxes, 64 B: what k_paths_stream fetches per traversal step.  Made from
// the binary tree by pulling grandchildren up (the child with the largest surface first), so it answers
// exactly what the binary tree answers.  A child's box on axis a is
//   [origin[a] + qlo * 2^(exp[a] - 127),  origin[a] + qhi * 2^(exp[a] - 127)],
// rounded outwards; origin[a] is a multiple of the axis' step small enough (|origin| / step <= 2^23) that
// origin + q * step is exact in fp32, so the decoded planes are exactly these numbers on the device too.
struct WideNode {
    float origin[3];
    uint8_t exp[3], n_children;
    uint8_t qlo[3][4]; // [axis][child]
    uint8_t qhi[3][4];
    int32_t c[4];      // as in the binary node: >= 0 a WideNode, < 0 ~((first << 3) | (count - 1)); unused slots hold kWideAbsent
    uint32_t pad[2];
};
static_assert(sizeof(WideNode) == 64, "WideNode must be 64 bytes");
constexpr int32_t kWideAbsent = INT32_MIN + 1;

struct WideBuild {
    std::vector<WideNode> nodes; // nodes[0] is the root
    uint32_t max_depth = 0;      // wide nodes on the longest root-to-leaf path
};

// The four-wide form of a binary tree (after a build or a refit).  O(N).
void make_wide(const BvhBuild &bvh, WideBuild &out);

// Experiment (-DPT_LDS_WIDE4): the four-wide form with unquantised planes, for scenes traversed from LDS.  A node is
// seven quads of floats: (lo of the four children) and (hi ...) per axis, then the four references (bit patterns);
// an absent child has lo = hi = +inf.  Returns the nodes on the longest root-to-leaf path.
uint32_t make_wide_fp32(const BvhBuild &bvh, std::vector<float> &out);

// Binned SAH (16 bins per axis), leaves of at most leaf_max objects.  Returns "" or an error text.
std::string build_bvh(const pt_scene_desc &desc, uint32_t leaf_max, BvhBuild &out);

// Same topology, new boxes: for a scene whose objects moved or changed size (same count, any types).  O(N);
// the tree's quality follows the motion, its answers do not depend on it.
std::string refit_bvh(const pt_scene_desc &desc, BvhBuild &bvh);

} // namespace pt
