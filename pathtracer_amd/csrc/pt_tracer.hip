// pt_tracer.hip -- gfx950 kernels and the C ABI of libpathtracer_amd.so.
//
// The reference renders with one thread per pixel that loops over samples and recurses through
// virtual hit()/scatter() calls (main.cu:81-100, :46-65).  Here the same path is a wavefront
// pipeline over queues of path records:
//
//   k_raygen      camera::get_ray for every (pixel, sample) of a pass          (camera.h:88-96)
//   k_intersect   hittable_list::hit through a flattened BVH                   (hittable_list.h:87-101)
//   k_shade       material::emitted / scatter, throughput update, compaction   (material.h, main.cu:59-64)
//   k_film        adds the finished paths of a pass to the linear sum buffer   (main.cu:93)
//
// All kernels run a fixed-size grid that strides over a queue whose length lives in device
// memory, so the host never waits for a count between bounces.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "pt_bvh.h"
#include "pt_device.h"

#define PT_BLOCK 256
#define PT_WAVES (PT_BLOCK / 64)

// ---------------------------------------------------------------------------------------------
// Device-side views
// ---------------------------------------------------------------------------------------------
struct SceneDev {
    const BvhNode *nodes;
    const PrimRec *prims;      // leaf (BVH) order
    const PrimRec *prims_list; // list order, for the exact scan
    const float4 *mats;
    uint32_t n_prims, n_nodes;
    uint32_t stack_cap; // entries per lane in the LDS traversal stack
    int32_t root_ref;   // the reference of the scene's only leaf where the tree is one (at most eight objects: the FLAT kernels start there), else 0
    const float4 *wide;      // the four-wide quantised form of the tree (pt::WideNode, 64 B each), for scenes traversed from HBM; or null
    uint32_t wide_stack_cap; // stack entries its traversal may use
    pt::ConeParams cone; // how wide a cone the slab test gives a ray
    DeviceCamera cam;
    float bg[3];
};

// A path record is 40 B in three planes:
//   A = (origin.xyz, path id bits)   B = (direction.xyz, throughput.r)   C = (throughput.g, throughput.b)
struct PathPlanes {
    float4 *A;
    float4 *B;
    float2 *C;
};

struct Counters { // device memory, zeroed per call
    unsigned long long rays, box_tests, prim_tests, exact_fallbacks, exact_replays;
    // Diagnostics of a -DPT_SECTION_CLOCKS build only (k_paths): clock ticks per code section summed
    // over waves, and active-lane / issued-lane slot counts of the three divergent loops.
    unsigned long long sect[8];
    unsigned long long lanes[6], slots[6];
};

#ifdef PT_SECTION_CLOCKS
constexpr bool kProf = true;
#else
constexpr bool kProf = false;
#endif
struct ProfLanes { // per-lane tallies; slots are counted by the first active lane of each issue
    uint32_t lanes[6] = {0, 0, 0, 0, 0, 0}, slots[6] = {0, 0, 0, 0, 0, 0};
    __device__ __forceinline__ void tick(int k)
    {
        if (kProf) {
            const unsigned long long m = __ballot(true);
            lanes[k] += 1;
            if ((threadIdx.x & 63u) == (uint32_t)(__ffsll((long long)m) - 1)) slots[k] += 64;
        }
    }
};

// ---------------------------------------------------------------------------------------------
// Nearest hit
// ---------------------------------------------------------------------------------------------
struct HitResult {
    float t;
    uint32_t slot; // index into the array that was searched; PT_NO_PRIM on a miss
};

// One object under the reference's control flow for a given t_max (the scan's running nearest t).
PT_DI bool hit_any(const PrimRec &q, const Ray &r, float t_min, float t_max, float &t)
{
    switch (q.meta & 7u) {
    case PT_SPHERE: return hit_sphere(q, r, t_min, t_max, t);
    case PT_XY_RECT: return hit_rect<2, 0, 1>(q, r, t_min, t_max, t);
    case PT_XZ_RECT: return hit_rect<1, 0, 2>(q, r, t_min, t_max, t);
    case PT_YZ_RECT: return hit_rect<0, 1, 2>(q, r, t_min, t_max, t);
    case PT_CONE: {
        ConeRoots cr;
        return cone_roots(q, r, cr) && hit_cone_ordered(cr, t_min, t_max, t);
    }
    default: return hit_cylinder(q, r, t_min, t_max, t);
    }
}

// hittable_list::hit (hittable_list.h:87-101) over the list-order records: the definition of the
// answer.  Used by PT_TRACE_LINEAR and to resolve the rare rays whose answer depends on list order.
PT_DI HitResult linear_nearest(const PrimRec *__restrict__ list, uint32_t n, const Ray &r, float t_min, float t_max)
{
    HitResult best;
    best.t = t_max;
    best.slot = PT_NO_PRIM;
    for (uint32_t i = 0; i < n; ++i) {
        PrimRec q = list[i];
        float t;
        if (hit_any(q, r, t_min, best.t, t)) {
            best.t = t;
            best.slot = i;
        }
    }
    return best;
}

// The reference's hit arithmetic is fp32 on unnormalised rays, and what it ACCEPTS is not exactly the
// geometry: a sphere at distance |oc| from the origin is hit when the computed discriminant
// half_b^2 - a*c is >= 0, and that value carries a rounding error of up to 15 * 2^-24 * |oc|^2 |d|^2
// (sphere.h:85-101: three-term dot products, the products, the subtraction).  The accepted root
// therefore puts the ray point within sqrt(r^2 + eta^2) <= r + eta of the centre, eta =
// sqrt(15 * 2^-24) |oc| = 9.5e-4 |oc|: far from the origin, small spheres are hit by rays that
// geometrically miss them, and real hits land visibly outside the exact surface.  The same scaling
// holds for the cylinder's quadratic (cylinder.h:96-123, two-term sums); rects only see rounding of
// order 2^-23 (|o| + t|d|).  A tree that culled by exact geometry would lose such hits, so the slab
// test treats the ray as a cone: a box is entered when some point o + t d comes within
// kappa * |d| * t of it (per axis), kappa = 2^-10 > 9.5e-4 plus the lower-order terms.  (What |oc|
// exceeds t|d| by -- the object's own extent -- is padded onto the boxes at build time.)
//
// Per axis x with w = kappa |d| / |d_x|, the condition  lo - kappa|d|t <= o + t d_x <= hi + kappa|d|t
// is  t >= near / (1 + w)  and, when w < 1,  t <= far / (1 - w)  (no upper bound otherwise), with
// near/far the plane distances (bound - o) / d_x.  Both are still one FMA per plane: the factors are
// folded into per-ray constants  t = bound * A + C,  one pair for the planes the ray meets first on
// that axis and one for the planes it meets last.  Which planes those are (lo or hi) follows from the
// sign of the direction, and a node stores each axis as (lo, lo, hi, hi): the near pair is read from
// byte offset 0 or 8 of the quad, the far pair from the other half, so no per-axis min/max is needed
// (on gfx950 a min/max issues at half the rate of an FMA).
struct BoxRay {
    float anx, any, anz, cnx, cny, cnz; // constants applied to the near planes
    float afx, afy, afz, cfx, cfy, cfz; // constants applied to the far planes
    uint32_t sx, sy, sz;                // byte offset of the near pair within an axis quad: 0 (lo planes near) or 8
    float eps;                          // absolute error bound of a plane distance (rounding of o * inv)
    float eps_far;                      // 2^-22 * the largest ratio of an axis' far constants to its near ones, (1 + w) / (1 - w): see the four-wide step
};

PT_DI void box_ray_axis(float o, float d, float kd, float &a_n, float &c_n, float &a_f, float &c_f, uint32_t &near_off, float &cmax, float &kmax)
{
    // A component of exactly (or nearly) zero would make the reciprocal infinite; the box test sees such
    // a ray as very slightly tilted instead (2^-60), which keeps every distance finite.  Only the pruning
    // test sees this value; hit arithmetic uses the true ray.
    const float tiny = 8.6736174e-19f;
    const float dd = fabsf(d) < tiny ? copysignf(tiny, d) : d;
    const float inv = __builtin_amdgcn_rcpf(dd);   // 1 ulp; the slack below covers it
    const float w = kd * fabsf(inv);               // kappa |d| / |d_x|
    const float g = __builtin_amdgcn_rcpf(1.0f + w) * 0.99999809f; // 1/(1+w), rounded down by 2^-19
    const float no = -(o * inv);
    a_n = inv * g;                                 // near plane
    c_n = no * g;
    if (w < 0.99f) {
        const float f = __builtin_amdgcn_rcpf(1.0f - w) * 1.00000191f; // 1/(1-w), rounded up by 2^-19
        a_f = inv * f;
        c_f = no * f;
        cmax = fmaxf(cmax, fabsf(c_f));
        kmax = fmaxf(kmax, (1.0f + w) * f);
    } else { // the cone is wider than the ray's progress along this axis: no far bound from it
        a_f = 0.0f;
        c_f = 3.0e38f;
    }
    cmax = fmaxf(cmax, fabsf(c_n));
    near_off = dd > 0.0f ? 0u : 8u; // d > 0: the lo planes are met first
}

// Slope of the cone for a ray from origin o (pt::ConeParams): the chord slope f(Dm) / Dm of the excess
// function of the smallest object that relies on the cone, f(D) = sqrt(r^2 + k^2 D^2) - r, that is
// k * eta / (sqrt(r^2 + eta^2) + r) with eta = k Dm.  Dm is the largest distance from o to the bounds
// of those objects, taken in the L1 norm, and the square root is replaced by max(r, eta) below it:
// both only widen the cone.
PT_DI float ray_kappa(const pt::ConeParams &cp, v3 o)
{
    if (!cp.adaptive) return cp.kappa_max;
    const float d1 = (fabsf(o.x - cp.qc[0]) + fabsf(o.y - cp.qc[1])) + (fabsf(o.z - cp.qc[2]) + cp.qh);
    const float eta = cp.kappa_max * d1;
    const float kq = (cp.kappa_max * 1.00001f) * eta * __builtin_amdgcn_rcpf(fmaxf(cp.r_min, eta) + cp.r_min);
    return fminf(fmaxf(kq, cp.kappa_min), cp.kappa_max); // no such objects: r_min is huge and kq = 0
}

PT_DI BoxRay make_box_ray(const Ray &r, float kappa)
{
    BoxRay b;
    // |d| from the hardware sqrt, rounded up: kd only has to be an upper bound of kappa |d|
    const float kd = kappa * __builtin_amdgcn_sqrtf(r.d.x * r.d.x + r.d.y * r.d.y + r.d.z * r.d.z) * 1.000001f;
    float cmax = 0.0f, kmax = 1.0f;
    box_ray_axis(r.o.x, r.d.x, kd, b.anx, b.cnx, b.afx, b.cfx, b.sx, cmax, kmax);
    box_ray_axis(r.o.y, r.d.y, kd, b.any, b.cny, b.afy, b.cfy, b.sy, cmax, kmax);
    box_ray_axis(r.o.z, r.d.z, kd, b.anz, b.cnz, b.afz, b.cfz, b.sz, cmax, kmax);
    b.eps = cmax * (1.0f / 4194304.0f); // 2^-22 * max|C|: twice the rounding bound of C and of bound * A near the origin
    b.eps_far = kmax * (1.00001f / 4194304.0f);
    return b;
}

// Conservative slab test of both child boxes (the role of aabb::hit, aabb.h:78-91) against the ray's
// cone.  It only prunes, so its arithmetic is free-form: one FMA per plane distance.  Rounding moves
// a distance by at most eps absolutely and 2^-22 relatively (reciprocal 2^-23, constant 2^-24, FMA
// 2^-24); the relative part is inside the 2^-19 by which the cone factors were rounded outwards.  A
// box the exact test accepts is therefore never rejected.
PT_DI void box_test2(const float2 &nx, const float2 &ny, const float2 &nz, const float2 &fx, const float2 &fy, const float2 &fz,
                     const BoxRay &b, float t_min, float t_best, bool &h0, bool &h1, float &tn0, float &tn1)
{
    // n* = near planes of (child 0, child 1) per axis, f* = far planes
    const float n0x = __builtin_fmaf(nx.x, b.anx, b.cnx), n1x = __builtin_fmaf(nx.y, b.anx, b.cnx);
    const float n0y = __builtin_fmaf(ny.x, b.any, b.cny), n1y = __builtin_fmaf(ny.y, b.any, b.cny);
    const float n0z = __builtin_fmaf(nz.x, b.anz, b.cnz), n1z = __builtin_fmaf(nz.y, b.anz, b.cnz);
    const float f0x = __builtin_fmaf(fx.x, b.afx, b.cfx), f1x = __builtin_fmaf(fx.y, b.afx, b.cfx);
    const float f0y = __builtin_fmaf(fy.x, b.afy, b.cfy), f1y = __builtin_fmaf(fy.y, b.afy, b.cfy);
    const float f0z = __builtin_fmaf(fz.x, b.afz, b.cfz), f1z = __builtin_fmaf(fz.y, b.afz, b.cfz);
    // An axis without a far bound gives 3e38, or NaN for the infinite planes of an absent child: fminf /
    // fmaxf drop a NaN operand, and such a child is rejected by its infinite near or far distance.
    // (written as instructions: through fminf / fmaxf the compiler first re-canonicalises t_best with an
    // extra v_max per visit, which IEEE minNum asks for and these finite-or-quiet-NaN values do not need)
    float m0, m1, tf0, tf1;
    asm("v_max_f32 %0, %1, %2" : "=v"(m0) : "v"(n0z), "v"(t_min));
    asm("v_max_f32 %0, %1, %2" : "=v"(m1) : "v"(n1z), "v"(t_min));
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(tn0) : "v"(n0x), "v"(n0y), "v"(m0));
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(tn1) : "v"(n1x), "v"(n1y), "v"(m1));
    asm("v_min_f32 %0, %1, %2" : "=v"(m0) : "v"(f0z), "v"(t_best));
    asm("v_min_f32 %0, %1, %2" : "=v"(m1) : "v"(f1z), "v"(t_best));
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(tf0) : "v"(f0x), "v"(f0y), "v"(m0));
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(tf1) : "v"(f1x), "v"(f1y), "v"(m1));
    h0 = tn0 <= tf0 + b.eps;
    h1 = tn1 <= tf1 + b.eps;
}

// A second bound for rays that run almost parallel to a coordinate plane.  On an axis where the cone widens
// faster than the ray advances (w = kappa |d| / |d_x| > 1) the planes the ray "meets last" give no upper
// bound on t (make_box_ray leaves that slot open) but a LOWER one: a box lying behind the ray on that axis
// is only reached once the cone has grown out to it, t >= (far plane distance) / (1 - w) (and never for w < 1,
// which make_box_ray also leaves open from 0.99 on).  Without it such a
// ray is let into every box behind it on the axis -- a quarter of a large scene for a ray along a coordinate
// axis, 300k-500k steps for one ray among the 1M spheres.  Few rays are that steep (two axes: about one in a
// million), so the constants are worked out where they are needed instead of being carried by every ray.
struct SteepBounds {
    float ax, cx, ay, cy, az, cz; // t >= plane * a + c per axis (a = 0, c = -3e38: no such bound)
    float eps;                    // absolute rounding allowance of these distances
};
PT_DI void steep_axis(float o, float d, float kd, float &a2, float &c2, float &cmax)
{
    const float tiny = 8.6736174e-19f;
    const float dd = fabsf(d) < tiny ? copysignf(tiny, d) : d; // as box_ray_axis
    const float inv = __builtin_amdgcn_rcpf(dd);
    const float w = kd * fabsf(inv);
    a2 = 0.0f;
    c2 = -3.0e38f;
    if (!(w < 0.99f)) { // the axes whose far slot make_box_ray leaves open
        // 1 / (w - 1), rounded down (the bound may only come out too low).  Up to w = 1.25 the factor is at least 4
        // and ill-conditioned: 4 serves there, and below w = 1, where a box behind the ray is never reached at all.
        const float h = (w > 1.25f ? __builtin_amdgcn_rcpf(w - 1.0f) : 4.0f) * 0.999995f;
        a2 = -(inv * h);
        c2 = (o * inv) * h;
        cmax = fmaxf(cmax, fabsf(c2));
    }
}
PT_DI SteepBounds make_steep_bounds(const Ray &r, float kappa)
{
    SteepBounds s;
    const float kd = kappa * __builtin_amdgcn_sqrtf(r.d.x * r.d.x + r.d.y * r.d.y + r.d.z * r.d.z) * 1.000001f; // as make_box_ray
    float cmax = 0.0f;
    steep_axis(r.o.x, r.d.x, kd, s.ax, s.cx, cmax);
    steep_axis(r.o.y, r.d.y, kd, s.ay, s.cy, cmax);
    steep_axis(r.o.z, r.d.z, kd, s.az, s.cz, cmax);
    s.eps = cmax * (1.0f / 4194304.0f);
    return s;
}
// true if the box with these far planes (the ones the ray meets last per axis) is out of the cone's reach before t_far
PT_DI bool steep_rejects(const SteepBounds &s, float pfx, float pfy, float pfz, float t_far, float eps)
{
    const float lx = __builtin_fmaf(pfx, s.ax, s.cx), ly = __builtin_fmaf(pfy, s.ay, s.cy), lz = __builtin_fmaf(pfz, s.az, s.cz);
    return fmaxf(fmaxf(lx, ly), lz) > t_far + eps + s.eps; // (a NaN from an infinite plane times zero never rejects)
}

// Nearest hit through the BVH, with the list scan's semantics:
//  * an object's root is accepted under the reference's own test !(root < t_min || t_max < root)
//    with t_max = the running nearest t, so equal t is accepted (hittable_list.h:93-95), and
//  * of two objects at exactly equal t the LATER list index wins, as it would in a scan.
// For every object whose accepted root does not depend on t_max this is order-independent, and the
// traversal order of the tree is irrelevant.  The one exception is a cone whose quadratic has a < 0
// (cone.h:117-122 then tries the larger root first): its nearer root is taken only if an EARLIER
// list entry already produced a hit nearer than the larger root.  Such a candidate is parked and
// decided after the traversal when that is possible from what the traversal knows; otherwise the
// ray is flagged and its scan is replayed exactly over the few objects it can hit (bvh_exact_scan).
// Where the traversal reads nodes and objects from.  GlobalSrc: the 64-B / 32-B records in HBM (cached
// in L2 / Infinity Cache).  LdsSrc: a copy in the workgroup's LDS for scenes that fit, stored as float4
// PLANES (quarter j of every node in plane j) so that the 64 lanes of one ds_read_b128 spread over all
// sixteen 16-byte bank slots instead of the four a 64-byte-strided record array would use.
// near / far plane pairs of a node record (one quad per axis, then the children's references) for a ray
PT_DI void planes_of_record(const float4 &a, const float4 &b, const float4 &c, const float4 &d, const BoxRay &br, float2 &nx, float2 &ny,
                            float2 &nz, float2 &fx, float2 &fy, float2 &fz, int32_t &c0, int32_t &c1)
{
    nx = br.sx ? make_float2(a.z, a.w) : make_float2(a.x, a.y); fx = br.sx ? make_float2(a.x, a.y) : make_float2(a.z, a.w);
    ny = br.sy ? make_float2(b.z, b.w) : make_float2(b.x, b.y); fy = br.sy ? make_float2(b.x, b.y) : make_float2(b.z, b.w);
    nz = br.sz ? make_float2(c.z, c.w) : make_float2(c.x, c.y); fz = br.sz ? make_float2(c.x, c.y) : make_float2(c.z, c.w);
    c0 = __float_as_int(d.x);
    c1 = __float_as_int(d.y);
}

struct GlobalSrc {
    static constexpr bool in_lds = false;
    const BvhNode *nodes;
    const PrimRec *prims;
    pt::ConeParams cone;
    static __device__ __forceinline__ void adapt(BoxRay &) {}
    // near / far plane pairs of node i for this ray, and the children's references
    __device__ __forceinline__ void planes(int32_t i, const BoxRay &br, float2 &nx, float2 &ny, float2 &nz, float2 &fx, float2 &fy,
                                           float2 &fz, int32_t &c0, int32_t &c1) const
    {
        const float4 *np = reinterpret_cast<const float4 *>(nodes + i);
        planes_of_record(np[0], np[1], np[2], np[3], br, nx, ny, nz, fx, fy, fz, c0, c1);
    }
    __device__ __forceinline__ void prim(uint32_t slot, float4 &w0, float4 &w1) const
    {
        const float4 *pp = reinterpret_cast<const float4 *>(prims + slot);
        w0 = pp[0]; w1 = pp[1];
    }
};
// The LDS copy of a node is 80 bytes: per axis the six floats (hi0, hi1, lo0, lo1, hi0, hi1) -- the children's upper
// planes stored twice around their lower ones -- then the two references.  The four floats a ray needs of an axis,
// (near0, near1, far0, far1), are then consecutive whichever way it runs: from byte 8 of the axis for a positive
// direction (lo, lo, hi, hi), from byte 0 for a negative one (hi, hi, lo, lo).  One address per axis (node base + a
// per-ray offset) serves both reads, four address instructions per visit instead of thirteen; and the 80-byte stride
// spreads the lanes of a read over all sixteen 16-byte bank slots, which a 64-byte stride would not.
#define PT_LDS_NODE_BYTES 80u
struct LdsSrc {
    static constexpr bool in_lds = true;
    const char *nodes;   // n_nodes records of PT_LDS_NODE_BYTES
    const float4 *prims; // 2 planes of n_prims
    uint32_t n_nodes, n_prims;
    pt::ConeParams cone;
    // the ray's byte offsets of (near, far) within a record, per axis
    static __device__ __forceinline__ void adapt(BoxRay &br)
    {
        br.sx = br.sx ^ 8u;
        br.sy = 24u + (br.sy ^ 8u);
        br.sz = 48u + (br.sz ^ 8u);
    }
    __device__ __forceinline__ void planes(int32_t i, const BoxRay &br, float2 &nx, float2 &ny, float2 &nz, float2 &fx, float2 &fy,
                                           float2 &fz, int32_t &c0, int32_t &c1) const
    {
        const char *base = nodes + (uint32_t)i * PT_LDS_NODE_BYTES;
        const char *px = base + br.sx, *py = base + br.sy, *pz = base + br.sz;
        nx = *reinterpret_cast<const float2 *>(px); fx = *reinterpret_cast<const float2 *>(px + 8);
        ny = *reinterpret_cast<const float2 *>(py); fy = *reinterpret_cast<const float2 *>(py + 8);
        nz = *reinterpret_cast<const float2 *>(pz); fz = *reinterpret_cast<const float2 *>(pz + 8);
        const int2 c = *reinterpret_cast<const int2 *>(base + 72);
        c0 = c.x;
        c1 = c.y;
    }
    __device__ __forceinline__ void prim(uint32_t slot, float4 &w0, float4 &w1) const
    {
        w0 = prims[slot]; w1 = prims[n_prims + slot];
    }
};

// Copies the scene into the workgroup's LDS (nodes in the 80-byte form, objects as two planes of quads) and returns
// the first byte after it.  The caller synchronises the workgroup before traversing.
template <int BLOCK>
PT_DI char *lds_scene_copy(const SceneDev &sc, float4 *s_mem, LdsSrc &lsrc)
{
    char *s_nodes = reinterpret_cast<char *>(s_mem);
    float4 *s_prims = reinterpret_cast<float4 *>(s_nodes + ((sc.n_nodes * PT_LDS_NODE_BYTES + 15u) & ~15u));
    const float4 *gn = reinterpret_cast<const float4 *>(sc.nodes), *gp = reinterpret_cast<const float4 *>(sc.prims);
    for (uint32_t i = threadIdx.x; i < 4 * sc.n_nodes; i += BLOCK) {
        const float4 q = gn[i]; // quarter i & 3 of node i >> 2: an axis quad (lo0, lo1, hi0, hi1), or the references
        float2 *dst = reinterpret_cast<float2 *>(s_nodes + (i >> 2) * PT_LDS_NODE_BYTES + (i & 3u) * 24u);
        if ((i & 3u) == 3u) {
            dst[0] = make_float2(q.x, q.y);
        } else {
            dst[0] = make_float2(q.z, q.w);
            dst[1] = make_float2(q.x, q.y);
            dst[2] = make_float2(q.z, q.w);
        }
    }
    for (uint32_t i = threadIdx.x; i < 2 * sc.n_prims; i += BLOCK) s_prims[(i & 1u) * sc.n_prims + (i >> 1)] = gp[i];
    lsrc.nodes = s_nodes;
    lsrc.prims = s_prims;
    return reinterpret_cast<char *>(s_prims + 2 * sc.n_prims);
}
static size_t lds_scene_size(uint32_t n_nodes, uint32_t n_prims)
{
    return (((size_t)n_nodes * PT_LDS_NODE_BYTES + 15u) & ~(size_t)15u) + (size_t)n_prims * 32;
}

#define PT_SCAN_MAX 16 // candidates the exact replay keeps per ray

// The list scan replayed exactly, for the rays whose answer depends on list order.  Whatever the
// running t_max, the scan can only ever accept an object one of whose own roots lies in
// [t_min, t_max] and passes its clip; one traversal WITHOUT nearest-hit pruning collects those
// objects (a handful: the surfaces the ray really crosses), they are ordered by list index, and
// hittable_list::hit (hittable_list.h:87-101) is replayed over them.  `overflow` is set when more
// than PT_SCAN_MAX candidates exist (the caller then scans the whole list).
template <int STRIDE, class Src>
static __device__ __attribute__((noinline)) HitResult bvh_exact_scan(const Src &src, uint32_t stack_cap, const Ray &r, const BoxRay &br, float t_min, float t_max,
                               uint32_t *stack, bool &overflow)
{
    uint32_t ids[PT_SCAN_MAX], slots[PT_SCAN_MAX];
    int n = 0;
    overflow = false;
    const bool steep = br.afx == 0.0f || br.afy == 0.0f || br.afz == 0.0f;
    SteepBounds sb = {0.0f, -3.0e38f, 0.0f, -3.0e38f, 0.0f, -3.0e38f, 0.0f};
    if (steep) sb = make_steep_bounds(r, ray_kappa(src.cone, r.o));
    uint32_t sp = 0;
    int32_t cur = 0;
    for (;;) {
        if (cur >= 0) {
            float2 nx, ny, nz, fx, fy, fz;
            int32_t c0, c1;
            src.planes(cur, br, nx, ny, nz, fx, fy, fz, c0, c1);
            float tn0, tn1;
            bool h0, h1;
            box_test2(nx, ny, nz, fx, fy, fz, br, t_min, t_max, h0, h1, tn0, tn1);
            if (steep) { // the bound the open far slot leaves out (see SteepBounds)
                const float tf0 = fminf(fminf(__builtin_fmaf(fx.x, br.afx, br.cfx), __builtin_fmaf(fy.x, br.afy, br.cfy)), fminf(__builtin_fmaf(fz.x, br.afz, br.cfz), t_max));
                const float tf1 = fminf(fminf(__builtin_fmaf(fx.y, br.afx, br.cfx), __builtin_fmaf(fy.y, br.afy, br.cfy)), fminf(__builtin_fmaf(fz.y, br.afz, br.cfz), t_max));
                if (steep_rejects(sb, fx.x, fy.x, fz.x, tf0, br.eps)) h0 = false;
                if (steep_rejects(sb, fx.y, fy.y, fz.y, tf1, br.eps)) h1 = false;
            }
            if (h0 && h1) {
                if (sp < stack_cap) stack[(sp++) * STRIDE] = (uint32_t)c1;
                else overflow = true;
                cur = c0;
                continue;
            }
            if (h0) { cur = c0; continue; }
            if (h1) { cur = c1; continue; }
        } else {
            const uint32_t ref = (uint32_t)~cur, first = ref >> 3, count = (ref & 7u) + 1u;
            for (uint32_t k = 0; k < count; ++k) {
                float4 w0, w1;
                src.prim(first + k, w0, w1);
                PrimRec q;
                q.p[0] = w0.x; q.p[1] = w0.y; q.p[2] = w0.z; q.p[3] = w0.w; q.p[4] = w1.x; q.p[5] = w1.y;
                q.meta = __float_as_uint(w1.z);
                q.id = __float_as_uint(w1.w);
                float t;
                bool possible;
                if ((q.meta & 7u) == PT_CONE) {
                    ConeRoots cr;
                    possible = cone_roots(q, r, cr) &&
                               ((!(cr.r1 < t_min || t_max < cr.r1) && cone_clip_ok(cr, cr.r1)) ||
                                (!(cr.r2 < t_min || t_max < cr.r2) && cone_clip_ok(cr, cr.r2)));
                } else {
                    possible = hit_any(q, r, t_min, t_max, t);
                }
                if (possible) {
                    if (n == PT_SCAN_MAX) { overflow = true; continue; }
                    int at = n++;
                    while (at > 0 && ids[at - 1] > q.id) { ids[at] = ids[at - 1]; slots[at] = slots[at - 1]; --at; }
                    ids[at] = q.id;
                    slots[at] = first + k;
                }
            }
        }
        if (sp == 0) break;
        cur = (int32_t)stack[(--sp) * STRIDE];
    }
    HitResult best;
    best.t = t_max;
    best.slot = PT_NO_PRIM;
    for (int e = 0; e < n; ++e) { // hittable_list.h:92-98 over the candidates, in list order
        float4 w0, w1;
        src.prim(slots[e], w0, w1);
        PrimRec q;
        q.p[0] = w0.x; q.p[1] = w0.y; q.p[2] = w0.z; q.p[3] = w0.w; q.p[4] = w1.x; q.p[5] = w1.y;
        q.meta = __float_as_uint(w1.z);
        q.id = __float_as_uint(w1.w);
        float t;
        if (hit_any(q, r, t_min, best.t, t)) {
            best.t = t;
            best.slot = slots[e];
        }
    }
    return best;
}

// The running answer of one ray's scan: the nearest accepted hit so far and, for scenes with cones,
// the parked candidate (see above).  k_paths_stream keeps one per lane across its steps; bvh_nearest
// below keeps the same state in locals (measured 6 % faster on the small scenes than going through this).
struct Nearest {
    float best_t;
    uint32_t best_slot, best_id;
    uint32_t n_pend, pend_slot, pend_id;
    float pend_r2, pend_r1;
    __device__ __forceinline__ void reset(float t_max)
    {
        best_t = t_max;
        best_slot = PT_NO_PRIM;
        best_id = 0;
        n_pend = 0; pend_slot = 0; pend_id = 0;
        pend_r2 = 0.0f; pend_r1 = 0.0f;
    }
    // equal t: the later list entry wins, as in the scan (hittable_list.h:93-95)
    __device__ __forceinline__ void take(bool h, float t, uint32_t slot, uint32_t id)
    {
        if (h && (t < best_t || best_slot == PT_NO_PRIM || id > best_id)) {
            best_t = t;
            best_slot = slot;
            best_id = id;
        }
    }
};

#define PT_TRAVERSAL_DONE INT32_MIN // not a leaf reference: those are > INT32_MIN for fewer than 2^28 objects

// One object under the scan's rules, against the running nearest hit.  A cone whose nearer root may or
// may not be reachable (see above) is parked in `nh` instead of decided.
template <bool CONES>
PT_DI void object_test(const PrimRec &q, uint32_t slot, const Ray &r, float t_min, float t_max, Nearest &nh)
{
    float t = 0.0f;
    bool h = false;
    switch (q.meta & 7u) {
    case PT_SPHERE: h = hit_sphere(q, r, t_min, nh.best_t, t); break;
    case PT_XY_RECT: h = hit_rect<2, 0, 1>(q, r, t_min, nh.best_t, t); break;
    case PT_XZ_RECT: h = hit_rect<1, 0, 2>(q, r, t_min, nh.best_t, t); break;
    case PT_YZ_RECT: h = hit_rect<0, 1, 2>(q, r, t_min, nh.best_t, t); break;
    case PT_CONE: {
        ConeRoots cr;
        if (CONES && cone_roots(q, r, cr)) {
            if (!(cr.a < 0.0f)) {
                h = hit_cone_ordered(cr, t_min, nh.best_t, t);
            } else {
                // larger root r1 first: an order-independent candidate
                if (!(cr.r1 < t_min || nh.best_t < cr.r1) && cone_clip_ok(cr, cr.r1)) {
                    h = true;
                    t = cr.r1;
                }
                // nearer root r2: only reachable in the scan when its running t_max < r1
                if (!(cr.r2 < t_min || nh.best_t < cr.r2) && !(cr.r1 < t_min) && cone_clip_ok(cr, cr.r2)) {
                    if (t_max < cr.r1) { // no running t_max can reach r1: always the r2 branch
                        h = true;
                        t = cr.r2;
                    } else {
                        if (nh.n_pend == 0) { nh.pend_slot = slot; nh.pend_id = q.id; nh.pend_r2 = cr.r2; nh.pend_r1 = cr.r1; }
                        nh.n_pend++;
                    }
                }
            }
        }
        break;
    }
    default: h = hit_cylinder(q, r, t_min, nh.best_t, t); break;
    }
    nh.take(h, t, slot, q.id);
}

// The same test for a wave whose lanes hold objects of different kinds (scenes with cones are mixed scenes).  Run
// through object_test() such a wave executes the code of every kind present, one after the other: ~230 vector
// instructions when all six are.  Here the three quadrics share one code path and the three rects another:
//  * sphere.h:85-101, cylinder.h:96-123 and cone.h:101-129 are the same computation -- oc, a, half_b, c,
//    discriminant, the roots (-half_b -+ sqrtd) / a tried in that order, then a clip of the hit's y -- with
//    different inputs: the cylinder and the cone drop the y components of oc and d (the reference zeroes them
//    and still goes through dot() and length_squared(), and x*x + 0*0 + z*z rounds as x*x + z*z does), the cone
//    adds its tangens terms, the sphere has no clip (bounds -inf, +inf).  Every product and sum below is the one
//    the reference's expression for that kind performs, in its order.
//  * aarect.h:348-355, :364-371, :380-387 differ only in which components they read.
PT_DI bool quadric_roots(const PrimRec &q, uint32_t type, const Ray &r, ConeRoots &cr)
{
    const bool is_sphere = type == PT_SPHERE, is_cone = type == PT_CONE;
    const v3 center = V(q.p[0], q.p[1], q.p[2]);
    v3 oc = vsub(r.o, center);
    v3 dd = r.d;
    oc.y = is_sphere ? oc.y : 0.0f;
    dd.y = is_sphere ? dd.y : 0.0f;
    float a = vlen2(dd);
    float half_b = vdot(oc, dd);
    const float height = q.p[3]; // (the sphere's radius sits here; its clip bounds below do not use it)
    const float radius = is_sphere ? q.p[3] : q.p[4];
    float k = radius * radius;
    if (is_cone) {
        const float tangens = q.p[5];
        const float D = height - r.o.y + center.y;
        a = a - tangens * r.d.y * r.d.y;
        half_b = half_b + tangens * r.d.y * D;
        k = tangens * D * D;
    }
    const float c = vlen2(oc) - k;
    const float discriminant = half_b * half_b - a * c;
    if (discriminant < 0) return false;
    const float sqrtd = sqrtf(discriminant);
    cr.a = a;
    cr.r1 = (-half_b - sqrtd) / a;
    cr.r2 = (-half_b + sqrtd) / a;
    cr.cy = is_sphere ? -__builtin_inff() : center.y;
    cr.cyh = is_sphere ? __builtin_inff() : center.y + height;
    cr.oy = r.o.y;
    cr.dy = r.d.y;
    return true;
}
PT_DI void object_test_fused(const PrimRec &q, uint32_t slot, const Ray &r, float t_min, float t_max, Nearest &nh)
{
    const uint32_t type = q.meta & 7u;
    float t = 0.0f;
    bool h = false;
    if (type - (uint32_t)PT_XY_RECT < 3u) {
        const bool xy = type == PT_XY_RECT, yz = type == PT_YZ_RECT;
        const float o_k = xy ? r.o.z : (yz ? r.o.x : r.o.y), d_k = xy ? r.d.z : (yz ? r.d.x : r.d.y); // the axis the rect is normal to
        const float o_0 = yz ? r.o.y : r.o.x, d_0 = yz ? r.d.y : r.d.x;
        const float o_1 = xy ? r.o.y : r.o.z, d_1 = xy ? r.d.y : r.d.z;
        t = (q.p[4] - o_k) / d_k;
        if (!(t < t_min || t > nh.best_t)) {
            const float c0 = o_0 + t * d_0;
            const float c1 = o_1 + t * d_1;
            h = !(c0 < q.p[0] || c0 > q.p[1] || c1 < q.p[2] || c1 > q.p[3]);
        }
    } else {
        ConeRoots cr;
        if (quadric_roots(q, type, r, cr)) {
            if (!(cr.a < 0.0f)) { // spheres and cylinders always (their a is a sum of squares, or NaN)
                h = hit_cone_ordered(cr, t_min, nh.best_t, t);
            } else { // as in object_test()
                if (!(cr.r1 < t_min || nh.best_t < cr.r1) && cone_clip_ok(cr, cr.r1)) {
                    h = true;
                    t = cr.r1;
                }
                if (!(cr.r2 < t_min || nh.best_t < cr.r2) && !(cr.r1 < t_min) && cone_clip_ok(cr, cr.r2)) {
                    if (t_max < cr.r1) {
                        h = true;
                        t = cr.r2;
                    } else {
                        if (nh.n_pend == 0) { nh.pend_slot = slot; nh.pend_id = q.id; nh.pend_r2 = cr.r2; nh.pend_r1 = cr.r1; }
                        nh.n_pend++;
                    }
                }
            }
        }
    }
    nh.take(h, t, slot, q.id);
}

// After the traversal: decides the parked cone candidate where the traversal's knowledge suffices,
// otherwise asks for the exact replay.
template <bool CONES>
PT_DI void settle_parked(Nearest &nh, bool &need_exact)
{
    if (!CONES) return;
    if (nh.n_pend > 1) {
        need_exact = true;
    } else if (nh.n_pend == 1 && !(nh.best_t < nh.pend_r2)) {
        // The parked cone is taken by the scan iff some EARLIER entry was accepted with t < pend_r1.
        if (nh.best_slot != PT_NO_PRIM && nh.best_t < nh.pend_r1) {
            if (nh.best_id < nh.pend_id) { // the nearest other hit is earlier in the list: it is such an entry
                nh.best_t = nh.pend_r2;
                nh.best_slot = nh.pend_slot;
            } else {
                need_exact = true; // nearest other hit is later in the list: earlier ones unknown
            }
        }
        // else: nothing at all was hit nearer than pend_r1, so the scan took the larger root: not a hit here
    }
}

// CONES = false compiles the cone case (and with it everything list-order dependent) out: the host
// selects it for scenes that contain no cone.
// KIND says what the scene holds, so that the host can pick a kernel without the code the scene cannot need:
//   PT_KIND_SPHERES  spheres only: no dispatch on the object tag at all, and no NaN roots (a sphere's quadratic has
//                    a = |d|^2 > 0 for every ray a render produces);
//   PT_KIND_NO_CONES any objects but cones: nothing list-order dependent;
//   PT_KIND_ANY      everything.
#define PT_KIND_SPHERES 0
#define PT_KIND_NO_CONES 1
#define PT_KIND_ANY 2
template <bool COUNT, int KIND, int STRIDE, class Src, bool FLAT = false /* the scene is one leaf, `root` its reference: no tree code at all */>
PT_DI HitResult bvh_nearest(const Src &src, uint32_t stack_cap, const Ray &r, float t_min, float t_max,
                            uint32_t *stack /* [sp*STRIDE] */, bool &need_exact, uint32_t &n_box, uint32_t &n_prim, uint32_t &n_replay,
                            ProfLanes *prof = nullptr, int32_t root = 0 /* FLAT: the reference of the scene's only leaf */)
{
    constexpr bool CONES = KIND == PT_KIND_ANY, SPH = KIND == PT_KIND_SPHERES;
    const float kappa = ray_kappa(src.cone, r.o);
    BoxRay br = {};
    if (!FLAT) { // a scene that is one leaf has no box to test: no slab constants either
        br = make_box_ray(r, kappa);
        Src::adapt(br);
    }
    // An axis whose far slot is open (rare).  Only followed up for scenes traversed from HBM, where such a ray
    // otherwise walks a large part of the tree; on the LDS-resident scenes (at most 1000 objects) the test costs
    // the kernel more registers than those rays cost it time (C2: 20.9 -> 22.1 ms with it).
    const bool steep = !Src::in_lds && (br.afx == 0.0f || br.afy == 0.0f || br.afz == 0.0f);

    float best_t = t_max;
    uint32_t best_slot = PT_NO_PRIM, best_id = 0;
    // parked cone candidate
    uint32_t n_pend = 0, pend_slot = 0, pend_id = 0;
    float pend_r2 = 0.0f, pend_r1 = 0.0f;
    need_exact = false;
    bool nan_hit = false;

    const int32_t done = INT32_MIN; // not a leaf reference: those are > INT32_MIN for fewer than 2^28 objects
    // The bottom entry of the stack says "done": a pop needs no test for an empty stack.  (The capacity is the tree's
    // depth + 2: a descent holds one entry per level.)
    if (!FLAT) stack[0] = (uint32_t)done;
    uint32_t sp = 1;
    int32_t cur = FLAT ? root : 0;
    for (;;) {
        // Descend until this lane holds a leaf (or has nothing left); lanes that found theirs wait here, so
        // that the expensive object tests below run with as many lanes of the wave as possible.
        while (!FLAT && cur >= 0) {
            float2 nx, ny, nz, fx, fy, fz;
            int32_t c0, c1;
            src.planes(cur, br, nx, ny, nz, fx, fy, fz, c0, c1);
            float tn0, tn1;
            bool h0, h1;
            box_test2(nx, ny, nz, fx, fy, fz, br, t_min, best_t, h0, h1, tn0, tn1);
            if (COUNT) n_box += 2;
            if (kProf && prof) prof->tick(0);
            if (!Src::in_lds && steep) { // the bound the open far slot leaves out (see SteepBounds)
                const SteepBounds sb = make_steep_bounds(r, kappa);
                const float tf0 = fminf(fminf(__builtin_fmaf(fx.x, br.afx, br.cfx), __builtin_fmaf(fy.x, br.afy, br.cfy)), fminf(__builtin_fmaf(fz.x, br.afz, br.cfz), best_t));
                const float tf1 = fminf(fminf(__builtin_fmaf(fx.y, br.afx, br.cfx), __builtin_fmaf(fy.y, br.afy, br.cfy)), fminf(__builtin_fmaf(fz.y, br.afz, br.cfz), best_t));
                if (steep_rejects(sb, fx.x, fy.x, fz.x, tf0, br.eps)) h0 = false;
                if (steep_rejects(sb, fx.y, fy.y, fz.y, tf1, br.eps)) h1 = false;
            }
            // Which child next, by selects; one masked store where both were hit, one masked load where neither was:
            // a wave spends as long on the scalar instructions and branches of a visit as on its arithmetic, and the
            // four-way if / else this replaces cost twice as many (the spheres frame 4.5 % slower).
            const bool both = h0 && h1;
            const bool pick1 = (both && tn1 < tn0) || (!h0 && h1);
            cur = pick1 ? c1 : c0;
            if (both) stack[(sp++) * STRIDE] = (uint32_t)(pick1 ? c0 : c1);
            if (!(h0 || h1)) cur = (int32_t)stack[(--sp) * STRIDE];
        }
        if (cur == done) break;
        {
            uint32_t ref = (uint32_t)~cur;
            uint32_t first = ref >> 3, count = (ref & 7u) + 1u;
            // Spheres are taken in two steps.  Their discriminant test is cheap and rejects most of a leaf's
            // spheres; the roots (a correctly rounded sqrt and one or two divides) are the expensive part.
            // Computing them where the discriminant is found would run that part once per leaf position at
            // which ANY lane of the wave has a candidate.  Instead a lane holds its candidate back until it
            // finds another one or the leaf ends, so that most lanes compute their roots together after the
            // loop.  The nearest hit does not depend on the order in which spheres, rects and cylinders are
            // tried (equal t goes to the later list entry either way).
            float held_hb = 0.0f, held_disc = 0.0f;
            uint32_t held_slot = PT_NO_PRIM /* nothing held: a flag of its own would live in scalar masks that every pass of the loop below merges */, held_id = 0;
            auto take = [&](bool h, float t, uint32_t slot, uint32_t id) {
                if (h && (t < best_t || best_slot == PT_NO_PRIM || id > best_id)) {
                    best_t = t;
                    best_slot = slot;
                    best_id = id;
                }
            };
            auto release = [&]() {
                float t = 0.0f;
                const bool h = sphere_roots(r, held_hb, held_disc, t_min, best_t, t);
                if (kProf && prof) prof->tick(4);
                take(h, t, held_slot, held_id);
            };
            for (uint32_t k = 0; k < count; ++k) {
                const uint32_t slot = first + k;
                float4 w0, w1;
                src.prim(slot, w0, w1);
                PrimRec q;
                q.p[0] = w0.x; q.p[1] = w0.y; q.p[2] = w0.z; q.p[3] = w0.w; q.p[4] = w1.x; q.p[5] = w1.y;
                q.meta = __float_as_uint(w1.z);
                q.id = __float_as_uint(w1.w);
                if (COUNT) n_prim += 1;
                if (kProf && prof) prof->tick(1);
                float t = 0.0f;
                bool h = false;
                const uint32_t type = q.meta & 7u;
                if (SPH || type == PT_SPHERE) { // the common object first: one compare instead of the switch's search
                    float hb, disc;
                    if (sphere_discriminant(q, r, hb, disc)) {
                        if (held_slot != PT_NO_PRIM) release();
                        held_hb = hb; held_disc = disc; held_slot = slot; held_id = q.id;
                    }
                    continue;
                }
                if (SPH) continue;
                switch (type) {
                case PT_XY_RECT: h = hit_rect<2, 0, 1>(q, r, t_min, best_t, t); break;
                case PT_XZ_RECT: h = hit_rect<1, 0, 2>(q, r, t_min, best_t, t); break;
                case PT_YZ_RECT: h = hit_rect<0, 1, 2>(q, r, t_min, best_t, t); break;
                case PT_CONE: {
                    ConeRoots cr;
                    if (CONES && cone_roots(q, r, cr)) {
                        if (!(cr.a < 0.0f)) {
                            h = hit_cone_ordered(cr, t_min, best_t, t);
                        } else {
                            // larger root r1 first: an order-independent candidate
                            if (!(cr.r1 < t_min || best_t < cr.r1) && cone_clip_ok(cr, cr.r1)) {
                                h = true;
                                t = cr.r1;
                            }
                            // nearer root r2: only reachable in the scan when its running t_max < r1
                            if (!(cr.r2 < t_min || best_t < cr.r2) && !(cr.r1 < t_min) && cone_clip_ok(cr, cr.r2)) {
                                if (t_max < cr.r1) { // no running t_max can reach r1: always the r2 branch
                                    h = true;
                                    t = cr.r2;
                                } else {
                                    if (n_pend == 0) { pend_slot = slot; pend_id = q.id; pend_r2 = cr.r2; pend_r1 = cr.r1; }
                                    n_pend++;
                                }
                            }
                        }
                    }
                    break;
                }
                default: h = hit_cylinder(q, r, t_min, best_t, t); break;
                }
                take(h, t, slot, q.id);
            }
            if (held_slot != PT_NO_PRIM) release();
        }
        // A hit at t = NaN was accepted (the reference's comparisons let it through, and from then on what the scan
        // accepts is a matter of list order: every later entry the ray's line crosses): the list itself answers this ray.
        if (!SPH && best_t != best_t) { nan_hit = true; break; }
        if (FLAT) break; // the one leaf is done
        cur = (int32_t)stack[(--sp) * STRIDE];
        if (cur == done) break;
    }

    if (nan_hit) need_exact = true;
    else if (!CONES) {
    } else if (n_pend > 1) need_exact = true;
    else if (n_pend == 1 && !(best_t < pend_r2)) {
        // The parked cone is taken by the scan iff some EARLIER entry was accepted with t < pend_r1.
        if (best_slot != PT_NO_PRIM && best_t < pend_r1) {
            if (best_id < pend_id) { // the nearest other hit is earlier in the list: it is such an entry
                best_t = pend_r2;
                best_slot = pend_slot;
            } else {
                need_exact = true; // nearest other hit is later in the list: earlier ones unknown
            }
        }
        // else: nothing at all was hit nearer than pend_r1, so the scan took the larger root: not a hit here
    }
    HitResult res;
    res.t = best_t;
    res.slot = best_slot;
    if (CONES && need_exact && !nan_hit) {
        if (FLAT) { // (the replay walks the tree from its root)
            br = make_box_ray(r, kappa);
            Src::adapt(br);
        }
        res = bvh_exact_scan<STRIDE>(src, stack_cap, r, br, t_min, t_max, stack, need_exact);
        n_replay++;
    }
    return res; // need_exact still set: a NaN hit, or more candidates than the replay keeps: the caller scans the list
}

// ---------------------------------------------------------------------------------------------
// Small wave / block helpers
// ---------------------------------------------------------------------------------------------
PT_DI uint32_t lane_prefix(unsigned long long mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

PT_DI unsigned long long wave_sum_u32(uint32_t v)
{
    unsigned long long s = v;
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    return s; // valid in lane 0
}

// hittable_list::hit (hittable_list.h:87-101) for ONE ray by the whole wave: the same answer as
// linear_nearest, 64 list entries at a time.  Every lane tests its entry against the running nearest t
// as it stood at the start of the round; the first accepting lane is what the scan would accept next,
// its t becomes the running value and the lanes after it are tested again.  A round per acceptance,
// so a ray that a scan would update k times costs n/64 + k rounds.  Once the running value is a NaN
// (a hit whose t is NaN is accepted by the reference's comparisons, and so is everything after it)
// no test depends on it any more: all accepting lanes up to the first one with an ordinary t are then
// what the scan accepts, and one round settles them.  All 64 lanes must call this together.
static __device__ __forceinline__ unsigned long long wave_list_scan(const PrimRec *__restrict__ list, uint32_t n, float ox, float oy, float oz, float dx, float dy, float dz,
                                                                 float t_min, float t_max)
{
    Ray r;
    r.o = V(ox, oy, oz);
    r.d = V(dx, dy, dz);
    const uint32_t lane = threadIdx.x & 63u;
    HitResult best; // wave-uniform
    best.t = t_max;
    best.slot = PT_NO_PRIM;
    for (uint32_t base = 0; base < n; base += 64u) {
        const bool valid = base + lane < n;
        const PrimRec q = list[valid ? base + lane : base];
        uint32_t start = 0;
        while (start < 64u) {
            float t = 0.0f;
            const bool acc = valid && lane >= start && hit_any(q, r, t_min, best.t, t);
            const unsigned long long accepted = __ballot(acc);
            if (!accepted) break;
            int j = __ffsll((long long)accepted) - 1;
            bool settled = false;
            if (best.t != best.t) {
                const unsigned long long ordinary = __ballot(acc && t == t);
                if (ordinary) j = __ffsll((long long)ordinary) - 1;
                else { j = 63 - __clzll((long long)accepted); settled = true; }
            }
            best.t = __shfl(t, j, 64);
            best.slot = base + (uint32_t)j;
            if (settled) break;
            start = (uint32_t)j + 1u;
        }
    }
    return ((unsigned long long)best.slot << 32) | __float_as_uint(best.t); // (registers, not memory, across the call)
}
// The same as a call.  Which form costs the calling kernel less is a matter of its register allocation:
// k_paths_stream takes the call where its cone code already fills the register file (100k mixed objects:
// 0.89 against 0.59 Grays/s inlined) and the inlined body otherwise (1M spheres: 1.07 against 0.95).
static __device__ __attribute__((noinline)) unsigned long long wave_list_scan_call(const PrimRec *__restrict__ list, uint32_t n, float ox, float oy, float oz, float dx, float dy,
                                                                                 float dz, float t_min, float t_max)
{
    return wave_list_scan(list, n, ox, oy, oz, dx, dy, dz, t_min, t_max);
}


PT_DI Ray load_ray_xyz(const float *o, const float *d, size_t i)
{
    Ray r;
    r.o = V(o[3 * i], o[3 * i + 1], o[3 * i + 2]);
    r.d = V(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
    return r;
}

// ---------------------------------------------------------------------------------------------
// pt_trace_rays kernel
// ---------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(PT_BLOCK) void k_trace(SceneDev sc, const float *__restrict__ o, const float *__restrict__ d,
                                                    size_t n, float t_min, float t_max, pt_hit *__restrict__ out,
                                                    Counters *cnt)
{
    extern __shared__ uint32_t s_stack[];
    for (size_t base = (size_t)blockIdx.x * PT_BLOCK; base < n; base += (size_t)gridDim.x * PT_BLOCK) {
        size_t i = base + threadIdx.x;
        if (i >= n) continue;
        Ray r = load_ray_xyz(o, d, i);
        HitResult h;
        const PrimRec *arr;
        uint32_t flags = 0;
        if (MODE == PT_TRACE_LINEAR) {
            h = linear_nearest(sc.prims_list, sc.n_prims, r, t_min, t_max);
            arr = sc.prims_list;
        } else {
            bool need_exact;
            uint32_t nb = 0, np = 0, nr = 0;
            GlobalSrc src{sc.nodes, sc.prims, sc.cone};
            if (MODE != PT_TRACE_BVH_IN_SCENE) src.cone.adaptive = 0; // origins from outside the scene's bounds: widest cone
            h = bvh_nearest<false, PT_KIND_ANY, PT_BLOCK>(src, sc.stack_cap, r, t_min, t_max, s_stack + threadIdx.x, need_exact, nb, np, nr);
            arr = sc.prims;
            if (nr) {
                flags = 2;
                atomicAdd(&cnt->exact_replays, 1ull);
            }
            if (need_exact) {
                h = linear_nearest(sc.prims_list, sc.n_prims, r, t_min, t_max);
                arr = sc.prims_list;
                flags |= 1;
                atomicAdd(&cnt->exact_fallbacks, 1ull);
            }
        }
        pt_hit res;
        if (h.slot == PT_NO_PRIM) {
            res.t = 0.0f; res.prim = PT_NO_PRIM; res.n[0] = res.n[1] = res.n[2] = 0.0f; res.front = 0;
            res.p[0] = res.p[1] = res.p[2] = 0.0f;
        } else {
            PrimRec q = arr[h.slot];
            v3 p, nrm;
            bool front;
            surface_at(q, r, h.t, p, nrm, front);
            res.t = h.t; res.prim = q.id;
            res.n[0] = nrm.x; res.n[1] = nrm.y; res.n[2] = nrm.z;
            res.front = front ? 1u : 0u;
            res.p[0] = p.x; res.p[1] = p.y; res.p[2] = p.z;
        }
        res.flags = flags;
        out[i] = res;
    }
}

// ---------------------------------------------------------------------------------------------
// Wavefront integrator kernels
// ---------------------------------------------------------------------------------------------
struct PassDev {
    uint32_t width, height, n_pixels;
    uint32_t samples;       // samples of every pixel in flight in this pass
    uint32_t sample_base;   // absolute index of the pass's first sample
    uint32_t k0, k1;        // Philox key = seed
    int32_t max_depth;
};

// camera::get_ray (camera.h:88-96) with the jitter of main.cu:90-91 for one (pixel, absolute sample).
PT_DI Ray camera_ray(const DeviceCamera &c, const PassDev &ps, uint32_t pixel, uint32_t sample_abs)
{
    uint32_t j = pixel / ps.width, i = pixel - j * ps.width;
    Rng g;
    g.pixel = pixel; g.sample = sample_abs; g.bounce = 0; g.block = 0; g.k0 = ps.k0; g.k1 = ps.k1;
    uint32_t w4[4];
    rng_words(g, w4);
    float u = ((float)i + u32_to_uniform(w4[0])) / (float)ps.width;   // main.cu:90
    float v = ((float)j + u32_to_uniform(w4[1])) / (float)ps.height;  // main.cu:91
    v3 disk = random_in_unit_disk_after_jitter(g, w4);
    v3 rd = vscale(c.lens_radius, disk);                                             // camera.h:89
    v3 offset = vadd(vscale(rd.x, V(c.u[0], c.u[1], c.u[2])), vscale(rd.y, V(c.v[0], c.v[1], c.v[2]))); // :90
    v3 origin = V(c.origin[0], c.origin[1], c.origin[2]);
    Ray r;
    r.o = vadd(origin, offset);                                                      // camera.h:93
    r.d = vsub(vsub(vadd(vadd(V(c.llc[0], c.llc[1], c.llc[2]), vscale(u, V(c.horizontal[0], c.horizontal[1], c.horizontal[2]))),
                         vscale(v, V(c.vertical[0], c.vertical[1], c.vertical[2]))), origin), offset); // camera.h:94
    return r;
}

// Path id = s*n_pixels + pixel.
__global__ __launch_bounds__(PT_BLOCK) void k_raygen(SceneDev sc, PassDev ps, PathPlanes st, uint32_t *n_active)
{
    const uint32_t total = ps.samples * ps.n_pixels;
    if (blockIdx.x == 0 && threadIdx.x == 0) n_active[0] = total;
    for (uint32_t id = blockIdx.x * PT_BLOCK + threadIdx.x; id < total; id += gridDim.x * PT_BLOCK) {
        uint32_t s = id / ps.n_pixels, pixel = id - s * ps.n_pixels;
        Ray r = camera_ray(sc.cam, ps, pixel, ps.sample_base + s);
        st.A[id] = make_float4(r.o.x, r.o.y, r.o.z, __uint_as_float(id));
        st.B[id] = make_float4(r.d.x, r.d.y, r.d.z, 1.0f);
        st.C[id] = make_float2(1.0f, 1.0f);
    }
}

#define PT_ISECT_LDS_BLOCK 1024 // workgroup size of the LDS-resident variant: two of them fill a CU's 32 wave slots

// Nearest hit for every queued ray.  LDS = true: the workgroup first copies the whole tree and object
// array into LDS (scenes up to a few thousand objects) and traverses from there.
template <bool COUNT, bool LDS, int BLOCK, int KIND>
__global__ __launch_bounds__(BLOCK) void k_intersect(SceneDev sc, PathPlanes st, uint2 *__restrict__ hits,
                                                     const uint32_t *__restrict__ n_ptr, Counters *cnt)
{
    extern __shared__ float4 s_mem[];
    const uint32_t n = *n_ptr;
    if (blockIdx.x * BLOCK >= n) return; // uniform per workgroup: nothing queued for it
    uint32_t *s_stack;
    LdsSrc lsrc{nullptr, nullptr, sc.n_nodes, sc.n_prims, sc.cone};
    if (LDS) {
        s_stack = reinterpret_cast<uint32_t *>(lds_scene_copy<BLOCK>(sc, s_mem, lsrc));
        __syncthreads();
    } else {
        s_stack = reinterpret_cast<uint32_t *>(s_mem);
    }
    GlobalSrc gsrc{sc.nodes, sc.prims, sc.cone};
    uint32_t nb = 0, np = 0, nf = 0, nr = 0;
    const float t_min = 0.001f /* main.cu:54 */, t_max = PT_FLT_MAX /* commons.h:11 */;
    for (uint32_t base = blockIdx.x * BLOCK; base < n; base += gridDim.x * BLOCK) {
        uint32_t i = base + threadIdx.x;
        if (i >= n) continue;
        float4 A = st.A[i], B = st.B[i];
        Ray r;
        r.o = V(A.x, A.y, A.z);
        r.d = V(B.x, B.y, B.z);
        bool need_exact;
        HitResult h;
        if (LDS) h = bvh_nearest<COUNT, KIND, BLOCK>(lsrc, sc.stack_cap, r, t_min, t_max, s_stack + threadIdx.x, need_exact, nb, np, nr);
        else h = bvh_nearest<COUNT, KIND, BLOCK>(gsrc, sc.stack_cap, r, t_min, t_max, s_stack + threadIdx.x, need_exact, nb, np, nr);
        uint32_t slot = h.slot;
        if (need_exact) {
            // resolved on the list-order records; the top bit of the slot says which array it indexes
            h = linear_nearest(sc.prims_list, sc.n_prims, r, t_min, t_max);
            slot = h.slot == PT_NO_PRIM ? PT_NO_PRIM : (h.slot | 0x80000000u);
            nf++;
        }
        hits[i] = make_uint2(__float_as_uint(h.t), slot);
    }
    if (COUNT) {
        unsigned long long sb = wave_sum_u32(nb), sp = wave_sum_u32(np);
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&cnt->box_tests, sb);
            atomicAdd(&cnt->prim_tests, sp);
        }
    }
    if (nf) atomicAdd(&cnt->exact_fallbacks, (unsigned long long)nf);
    if (nr) atomicAdd(&cnt->exact_replays, (unsigned long long)nr);
}

// Where material::scatter's random numbers come from: the path's own stream, or values handed in (pt_scatter_batch).
struct StreamRand {
    Rng g;
    __device__ __forceinline__ v3 in_unit_sphere() { return random_in_unit_sphere(g); } // cuda_functions.h:14-20
    __device__ __forceinline__ float uniform()                                           // curand_uniform, material.h:235
    {
        uint32_t w4[4];
        rng_words(g, w4);
        return u32_to_uniform(w4[0]);
    }
};
struct InjectedRand {
    v3 sphere_pt;
    float xi;
    __device__ __forceinline__ v3 in_unit_sphere() { return sphere_pt; }
    __device__ __forceinline__ float uniform() { return xi; }
};

// material::scatter (material.h:90-99 lambertian, :155-162 metal, :223-242 dielectric, :295-299 diffuse_light) for a hit
// with face-forwarded `normal`: the reference's bool, the scattered direction and the attenuation.
template <class Rand>
PT_DI bool material_scatter(uint32_t mtype, const float4 &m, const Ray &r, const v3 &normal, bool front, Rand &rnd, v3 &dir, v3 &atten)
{
    dir = V(0.0f, 0.0f, 0.0f);
    atten = V(0.0f, 0.0f, 0.0f);
    if (mtype == PT_DIFFUSE_LIGHT) return false; // material.h:295-299
    // unit_vector(r_in.direction()) and its reflection, material.h:156 and :228 / :238: worked out once for the lanes of both
    // materials (a wave that shades a metal hit nearly always shades a glass hit too, and would run the code twice)
    v3 unit_direction = V(0.0f, 0.0f, 0.0f), reflected = V(0.0f, 0.0f, 0.0f);
    if (mtype != PT_LAMBERTIAN) {
        unit_direction = vunit(r.d);
        reflected = vreflect(unit_direction, normal);
    }
    if (mtype == PT_DIELECTRIC) { // material.h:223-242
        float ir = m.x;
        atten = V(1.0f, 1.0f, 1.0f);
        float refraction_ratio = front ? (1.0f / ir) : ir;
        float cos_theta = fminf(vdot(vneg(unit_direction), normal), 1.0f);
        float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
        bool reflect_it = refraction_ratio * sin_theta > 1.0f;
        if (!reflect_it) // the uniform is consumed only when refraction is possible (material.h:235)
            reflect_it = reflectance_schlick(cos_theta, refraction_ratio) > rnd.uniform();
        dir = reflect_it ? reflected : vrefract(unit_direction, normal, refraction_ratio);
        return true;
    }
    v3 rs = rnd.in_unit_sphere();
    atten = V(m.x, m.y, m.z);
    if (mtype == PT_LAMBERTIAN) { // material.h:90-99
        dir = vadd(normal, vunit(rs));
        if (vnear_zero(dir)) dir = normal;
        return true;
    }
    // metal, material.h:155-162
    dir = vadd(reflected, vscale(m.w, rs));
    return vdot(dir, normal) > 0.0f; // absorbed paths return emitted() = black (main.cu:61-62)
}

// main.cu:59-64 for one hit: emitted(), scatter(), throughput.  Returns true when the path goes on
// (out_r, out_T are its next ray and throughput); otherwise `result` is the path's final radiance:
// emitted() is non-zero only for diffuse_light, whose scatter() returns false (material.h:295-304),
// so a path's radiance is known exactly when it ends and is T * (emit | background | black).
template <bool SPH = false>
PT_DI bool shade_hit(const float4 *__restrict__ mats, const PassDev &ps, const Ray &r, v3 T, uint32_t pixel, uint32_t sample_abs,
                     int32_t bounce, const PrimRec &q, float t, Ray &out_r, v3 &out_T, v3 &result, ProfLanes *prof = nullptr)
{
    v3 p, normal;
    bool front;
    surface_at<SPH>(q, r, t, p, normal, front);
    const uint32_t mtype = (q.meta >> 3) & 3u;
    const float4 m = mats[q.meta >> 5];
    StreamRand rnd;
    rnd.g.pixel = pixel; rnd.g.sample = sample_abs; rnd.g.bounce = (uint32_t)bounce + 1u; rnd.g.block = 0;
    rnd.g.k0 = ps.k0; rnd.g.k1 = ps.k1;
    v3 dir, atten;
    result = V(0.0f, 0.0f, 0.0f);
    if (mtype == PT_DIFFUSE_LIGHT) result = vmul(T, V(m.x, m.y, m.z)); // material.h:302-304
    const bool scattered = material_scatter(mtype, m, r, normal, front, rnd, dir, atten);
    if (kProf && prof && (mtype == PT_LAMBERTIAN || mtype == PT_METAL)) { // blocks this lane drew against the blocks the wave had to issue
        const uint32_t tr = rnd.g.block;
        uint32_t mx = 1;
        while (__ballot(tr > mx)) ++mx;
        const unsigned long long am = __ballot(true);
        prof->lanes[2] += tr;
        if ((threadIdx.x & 63u) == (uint32_t)(__ffsll((long long)am) - 1)) prof->slots[2] += 64 * mx;
    }
    // main.cu:64: attenuation * ray_color(scattered, depth - 1); at the last allowed bounce the next
    // ray_color call has depth <= 0 and returns black (main.cu:50-51)
    const bool go_on = scattered && (bounce + 1 < ps.max_depth);
    out_r.o = p;
    out_r.d = dir;
    out_T = vmul(T, atten);
    return go_on;
}

// pt_scatter_batch: material::scatter and emitted() for given hits with the random numbers handed in.
struct ScatterOut { uint32_t scattered; float atten[3], o[3], d[3], emitted[3]; };
__global__ __launch_bounds__(PT_BLOCK) void k_scatter(SceneDev sc, const float *__restrict__ o, const float *__restrict__ d, const float *__restrict__ t,
                                                      const uint32_t *__restrict__ prim, const float *__restrict__ sphere_pt, const float *__restrict__ xi, size_t n,
                                                      ScatterOut *__restrict__ out)
{
    for (size_t i = (size_t)blockIdx.x * PT_BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * PT_BLOCK) {
        const Ray r = load_ray_xyz(o, d, i);
        const PrimRec *qp = sc.prims_list + prim[i];
        PrimRec q = *qp;
        v3 p, normal;
        bool front;
        surface_at(q, r, t[i], p, normal, front);
        const uint32_t mtype = (q.meta >> 3) & 3u;
        const float4 m = sc.mats[q.meta >> 5];
        InjectedRand rnd;
        rnd.sphere_pt = V(sphere_pt[3 * i], sphere_pt[3 * i + 1], sphere_pt[3 * i + 2]);
        rnd.xi = xi[i];
        v3 dir, atten;
        const bool ok = material_scatter(mtype, m, r, normal, front, rnd, dir, atten);
        ScatterOut so;
        so.scattered = ok ? 1u : 0u;
        so.atten[0] = atten.x; so.atten[1] = atten.y; so.atten[2] = atten.z;
        so.o[0] = p.x; so.o[1] = p.y; so.o[2] = p.z;
        so.d[0] = dir.x; so.d[1] = dir.y; so.d[2] = dir.z;
        const bool light = mtype == PT_DIFFUSE_LIGHT; // material.h:45-47, :302-304
        so.emitted[0] = light ? m.x : 0.0f; so.emitted[1] = light ? m.y : 0.0f; so.emitted[2] = light ? m.z : 0.0f;
        out[i] = so;
    }
}

PT_DI PrimRec load_prim(const PrimRec *arr, uint32_t slot)
{
    const float4 *pp = reinterpret_cast<const float4 *>(arr + slot);
    float4 w0 = pp[0], w1 = pp[1];
    PrimRec q;
    q.p[0] = w0.x; q.p[1] = w0.y; q.p[2] = w0.z; q.p[3] = w0.w; q.p[4] = w1.x; q.p[5] = w1.y;
    q.meta = __float_as_uint(w1.z);
    q.id = __float_as_uint(w1.w);
    return q;
}

// Shades every path of the queue and compacts the survivors into the next queue.
__global__ __launch_bounds__(PT_BLOCK) void k_shade(SceneDev sc, PassDev ps, PathPlanes in, const uint2 *__restrict__ hits,
                                                    PathPlanes out, const uint32_t *__restrict__ n_in_ptr,
                                                    uint32_t *n_out_ptr, float *__restrict__ rad, int32_t bounce)
{
    __shared__ uint32_t s_cnt[PT_WAVES];
    __shared__ uint32_t s_base;
    const uint32_t n = *n_in_ptr;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t base = blockIdx.x * PT_BLOCK; base < n; base += gridDim.x * PT_BLOCK) {
        uint32_t i = base + threadIdx.x;
        bool survive = false;
        float4 oA, oB;
        float2 oC;
        if (i < n) {
            float4 A = in.A[i], B = in.B[i];
            float2 C = in.C[i];
            uint2 hh = hits[i];
            uint32_t id = __float_as_uint(A.w);
            v3 T = V(B.w, C.x, C.y);
            Ray r;
            r.o = V(A.x, A.y, A.z);
            r.d = V(B.x, B.y, B.z);
            v3 result;
            if (hh.y == PT_NO_PRIM) {
                result = vmul(T, V(sc.bg[0], sc.bg[1], sc.bg[2])); // main.cu:54-55
            } else {
                PrimRec q = load_prim((hh.y & 0x80000000u) ? sc.prims_list : sc.prims, hh.y & 0x7FFFFFFFu);
                uint32_t s = id / ps.n_pixels, pixel = id - s * ps.n_pixels;
                Ray nr;
                v3 nT;
                survive = shade_hit(sc.mats, ps, r, T, pixel, ps.sample_base + s, bounce, q, __uint_as_float(hh.x), nr, nT, result);
                if (survive) {
                    oA = make_float4(nr.o.x, nr.o.y, nr.o.z, A.w);
                    oB = make_float4(nr.d.x, nr.d.y, nr.d.z, nT.x);
                    oC = make_float2(nT.y, nT.z);
                }
            }
            if (!survive) {
                rad[3 * (size_t)id + 0] = result.x;
                rad[3 * (size_t)id + 1] = result.y;
                rad[3 * (size_t)id + 2] = result.z;
            }
        }
        // compaction: one atomic per block chunk
        unsigned long long mask = __ballot(survive);
        if (lane == 0) s_cnt[wave] = (uint32_t)__popcll(mask);
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t tot = 0;
            for (int w = 0; w < PT_WAVES; ++w) { uint32_t c = s_cnt[w]; s_cnt[w] = tot; tot += c; }
            s_base = tot ? atomicAdd(n_out_ptr, tot) : 0u;
        }
        __syncthreads();
        if (survive) {
            uint32_t pos = s_base + s_cnt[wave] + lane_prefix(mask);
            out.A[pos] = oA;
            out.B[pos] = oB;
            out.C[pos] = oC;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// Persistent-workgroup wavefront: the same stages (camera ray -> nearest hit -> scatter/emit ->
// compaction -> film), but a workgroup owns a SLICE of paths (all samples of a run of pixels) and
// takes it through every bounce itself.  Its two ping-pong queues and its radiance buffer are a
// private region of HBM (cache-resident in practice), queue lengths live in LDS, compaction is a
// wave ballot + one LDS atomic per wave, stages are separated by workgroup barriers instead of
// kernel launches, and slices are handed out by a ticket counter.  No global atomics on the path,
// no per-bounce launches, no latency-bound tail launches.
// ---------------------------------------------------------------------------------------------
struct SliceDev {
    float4 *qA;   // [grid][2][cap]
    float4 *qB;
    float2 *qC;
    float *rad;   // [grid][cap*3]
    uint32_t cap; // paths per slice = pix_per_slice * samples
    uint32_t pix_per_slice, n_slices; // the largest slice; slices in the table
    const uint2 *extents;             // [n_slices] (first pixel, pixels): decreasing sizes towards the end of the frame
    uint32_t *ticket;
};

template <bool COUNT, bool LDS, int BLOCK, bool SORT, int KIND, bool SPLIT = false, bool FLAT = false>
__global__ __launch_bounds__(BLOCK, 6) void k_paths(SceneDev sc, PassDev ps, SliceDev sl, float *__restrict__ rgb_sum, Counters *cnt)
{
    constexpr int WAVES = BLOCK / 64;
    extern __shared__ float4 s_mem[];
    __shared__ uint32_t s_slice, s_out[4]; // [parity][0: queued from the front, 1: from the back (SPLIT)]
    __shared__ uint2 s_hit[SORT ? BLOCK : 1];       // (t bits, slot) of the chunk's rays, by thread
    __shared__ uint16_t s_perm[SORT ? BLOCK : 1];   // thread whose ray is shaded at each sorted position
    __shared__ uint16_t s_bin[SORT ? 4 * WAVES : 1]; // per (class, wave) counts, then exclusive offsets
    uint32_t *s_stack;
    LdsSrc lsrc{nullptr, nullptr, sc.n_nodes, sc.n_prims, sc.cone};
    if (LDS) {
        s_stack = reinterpret_cast<uint32_t *>(lds_scene_copy<BLOCK>(sc, s_mem, lsrc));
    } else {
        s_stack = reinterpret_cast<uint32_t *>(s_mem);
    }
    GlobalSrc gsrc{sc.nodes, sc.prims, sc.cone};
    const uint32_t cap = sl.cap;
    float4 *const qA = sl.qA + (size_t)blockIdx.x * 2 * cap;
    float4 *const qB = sl.qB + (size_t)blockIdx.x * 2 * cap;
    float2 *const qC = sl.qC + (size_t)blockIdx.x * 2 * cap;
    float *const rad = sl.rad + (size_t)blockIdx.x * 3 * cap;
    const float t_min = 0.001f /* main.cu:54 */, t_max = PT_FLT_MAX /* commons.h:11 */;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    unsigned long long rays = 0;
    uint32_t nb = 0, np = 0, nf = 0, nr = 0;
    ProfLanes prof;
    unsigned long long sect[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long t_last = kProf ? clock64() : 0;
    auto mark = [&](int k) {
        if (kProf) {
            const long long now = clock64();
            sect[k] += (unsigned long long)(now - t_last);
            t_last = now;
        }
    };

    // nearest hit of one ray; a list-order-resolved hit is returned with the top bit of the slot set
    auto trace = [&](const Ray &r) -> HitResult {
        bool need_exact;
        HitResult h;
        const int32_t root_ref = FLAT ? sc.root_ref : 0; // (a run-time start reference in the kernels with tree code costs them 0.5 %)
        if (LDS) h = bvh_nearest<COUNT, KIND, BLOCK, LdsSrc, FLAT>(lsrc, sc.stack_cap, r, t_min, t_max, s_stack + threadIdx.x, need_exact, nb, np, nr, &prof, root_ref);
        else h = bvh_nearest<COUNT, KIND, BLOCK, GlobalSrc, FLAT>(gsrc, sc.stack_cap, r, t_min, t_max, s_stack + threadIdx.x, need_exact, nb, np, nr, &prof, root_ref);
        if (need_exact) {
            h = linear_nearest(sc.prims_list, sc.n_prims, r, t_min, t_max);
            if (h.slot != PT_NO_PRIM) h.slot |= 0x80000000u;
            nf++;
        }
        return h;
    };

    for (;;) {
        __syncthreads(); // previous slice's film pass is done with rad and the queues
        if (threadIdx.x == 0) s_slice = atomicAdd(sl.ticket, 1u);
        __syncthreads();
        const uint32_t slice = s_slice;
        mark(0);
        if (slice >= sl.n_slices) break;
        const uint2 extent = sl.extents[slice]; // first pixel, pixels
        const uint32_t pix0 = extent.x;
        const uint32_t npix = extent.y;
        uint32_t in_count = npix * ps.samples, in_lo = in_count;
        uint32_t cur = 0;
        for (int32_t b = 0; b < ps.max_depth && in_count > 0; ++b) {
            if (threadIdx.x == 0) s_out[2 * (b & 1)] = s_out[2 * (b & 1) + 1] = 0;
            __syncthreads();
            mark(4);
            const float4 *inA = qA + cur * cap, *inB = qB + cur * cap;
            const float2 *inC = qC + cur * cap;
            float4 *outA = qA + (cur ^ 1u) * cap, *outB = qB + (cur ^ 1u) * cap;
            float2 *outC = qC + (cur ^ 1u) * cap;

            // main.cu:54-64 for one traced ray, then compaction of the wave's survivors into the next queue
            // (ballot + one LDS atomic per wave).  Every lane of the wave calls it; `valid` masks the idle ones.
            auto shade_emit = [&](bool valid, const Ray &r, v3 T, uint32_t lid, HitResult h) {
                bool survive = false;
                float4 oA, oB;
                float2 oC;
                if (valid) {
                    v3 result;
                    if (h.slot == PT_NO_PRIM) {
                        result = vmul(T, V(sc.bg[0], sc.bg[1], sc.bg[2])); // main.cu:54-55
                    } else {
                        PrimRec q;
                        if (LDS && (KIND == PT_KIND_SPHERES || (h.slot & 0x80000000u) == 0)) { // the copy in LDS, unless the list resolved the hit (never in a sphere-only scene)
                            float4 w0, w1;
                            lsrc.prim(h.slot, w0, w1);
                            q.p[0] = w0.x; q.p[1] = w0.y; q.p[2] = w0.z; q.p[3] = w0.w; q.p[4] = w1.x; q.p[5] = w1.y;
                            q.meta = __float_as_uint(w1.z);
                            q.id = __float_as_uint(w1.w);
                        } else {
                            q = load_prim((h.slot & 0x80000000u) ? sc.prims_list : sc.prims, h.slot & 0x7FFFFFFFu);
                        }
                        uint32_t s = lid / npix, pl = lid - s * npix;
                        Ray nr;
                        v3 nT;
                        if (kProf) prof.tick(3);
                        survive = shade_hit<KIND == PT_KIND_SPHERES>(sc.mats, ps, r, T, pix0 + pl, ps.sample_base + s, b, q, h.t, nr, nT, result, &prof);
                        if (survive) {
                            oA = make_float4(nr.o.x, nr.o.y, nr.o.z, __uint_as_float(lid));
                            oB = make_float4(nr.d.x, nr.d.y, nr.d.z, nT.x);
                            oC = make_float2(nT.y, nT.z);
                        }
                    }
                    if (!survive) {
                        rad[3 * lid + 0] = result.x;
                        rad[3 * lid + 1] = result.y;
                        rad[3 * lid + 2] = result.z;
                    }
                }
                if (SPLIT) {
                    // Survivors are queued from both ends of the next queue: rays that head up (by the camera's up vector; in an
                    // open scene most of them leave it after a few node visits) from the back, the others from the front.  A
                    // wave of the next bounce then holds rays of one kind, but for the one chunk where the two runs meet, and
                    // its lanes finish their traversals closer together.
                    const bool up = oB.x * sc.cam.v[0] + oB.y * sc.cam.v[1] + oB.z * sc.cam.v[2] > 0.0f;
                    const unsigned long long m0 = __ballot(survive && !up), m1 = __ballot(survive && up);
                    if (m0 | m1) {
                        const unsigned long long mine = up ? m1 : m0;
                        const uint32_t pre = lane_prefix(mine);
                        uint32_t pos0 = 0;
                        if (survive && pre == 0) pos0 = atomicAdd(&s_out[2 * (b & 1) + (up ? 1 : 0)], (uint32_t)__popcll(mine));
                        const uint32_t p0 = __shfl(pos0, m0 ? __ffsll((long long)m0) - 1 : 0, 64), p1 = __shfl(pos0, m1 ? __ffsll((long long)m1) - 1 : 0, 64);
                        if (survive) {
                            const uint32_t at = up ? cap - 1u - (p1 + pre) : p0 + pre;
                            outA[at] = oA;
                            outB[at] = oB;
                            outC[at] = oC;
                        }
                    }
                    return;
                }
                const unsigned long long mask = __ballot(survive);
                if (mask) {
                    const uint32_t pre = lane_prefix(mask);
                    uint32_t pos0 = 0;
                    if (survive && pre == 0) pos0 = atomicAdd(&s_out[2 * (b & 1)], (uint32_t)__popcll(mask));
                    pos0 = __shfl(pos0, __ffsll((long long)mask) - 1, 64);
                    if (survive) {
                        outA[pos0 + pre] = oA;
                        outB[pos0 + pre] = oB;
                        outC[pos0 + pre] = oC;
                    }
                }
            };

            for (uint32_t base = 0; base < in_count; base += BLOCK) {
                const uint32_t i = base + threadIdx.x;
                if (b == 0) {
                    // camera rays are made on the fly; neighbouring pixels mostly hit the same material
                    Ray r;
                    HitResult h;
                    h.t = 0.0f;
                    h.slot = PT_NO_PRIM;
                    // Camera rays are numbered sample-fast: a wave starts with the samples of one or a few pixels, whose rays
                    // run through the same nodes and mostly end on the same object (C2: 4.5 % faster than one sample of
                    // 64 neighbouring pixels per wave).  A path reports to rad[sample * npix + pixel] either way.
                    uint32_t lid0 = 0;
                    if (i < in_count) {
                        const uint32_t pl = i / ps.samples, s = i - pl * ps.samples;
                        lid0 = s * npix + pl;
                        r = camera_ray(sc.cam, ps, pix0 + pl, ps.sample_base + s);
                        mark(1);
                        h = trace(r);
                    }
                    mark(2);
                    shade_emit(i < in_count, r, V(1.0f, 1.0f, 1.0f), lid0, h);
                    mark(3);
                } else if (!SORT) {
                    Ray r;
                    v3 T = V(0.0f, 0.0f, 0.0f);
                    uint32_t lid = 0;
                    HitResult h;
                    h.t = 0.0f;
                    h.slot = PT_NO_PRIM;
                    if (i < in_count) {
                        const uint32_t qi = SPLIT && i >= in_lo ? cap - 1u - (i - in_lo) : i; // (the rays queued from the back)
                        float4 A = inA[qi], B = inB[qi];
                        float2 C = inC[qi];
                        lid = __float_as_uint(A.w);
                        r.o = V(A.x, A.y, A.z);
                        r.d = V(B.x, B.y, B.z);
                        T = V(B.w, C.x, C.y);
                        mark(1);
                        h = trace(r);
                    }
                    mark(2);
                    shade_emit(i < in_count, r, T, lid, h);
                    mark(3);
                } else {
                    // Stage 1: nearest hit for the chunk's rays, in queue order.
                    uint32_t cls = 3; // 0 ends without random numbers (miss, light) | 1 glass | 2 diffuse, metal | 3 no ray
                    if (i < in_count) {
                        float4 A = inA[i], B = inB[i];
                        Ray r;
                        r.o = V(A.x, A.y, A.z);
                        r.d = V(B.x, B.y, B.z);
                        HitResult h = trace(r);
                        s_hit[threadIdx.x] = make_uint2(__float_as_uint(h.t), h.slot);
                        cls = 0;
                        if (h.slot != PT_NO_PRIM) {
                            const PrimRec *arr = (h.slot & 0x80000000u) ? sc.prims_list : sc.prims;
                            const uint32_t mtype = (arr[h.slot & 0x7FFFFFFFu].meta >> 3) & 3u;
                            cls = mtype == PT_DIFFUSE_LIGHT ? 0u : (mtype == PT_DIELECTRIC ? 1u : 2u);
                        }
                    }
                    // Stage 2: counting sort of the chunk by material class, so that the waves of stage 3 are
                    // (nearly) homogeneous: the rejection-sampling loop runs in fewer, fuller waves.
                    uint32_t rank = 0;
#pragma unroll
                    for (uint32_t k = 0; k < 4; ++k) {
                        const unsigned long long m = __ballot(cls == k);
                        if (cls == k) rank = lane_prefix(m);
                        if (lane == 0) s_bin[k * WAVES + wave] = (uint16_t)__popcll(m);
                    }
                    __syncthreads();
                    if (threadIdx.x == 0) {
                        uint32_t acc = 0;
                        for (int e = 0; e < 4 * WAVES; ++e) { uint32_t c = s_bin[e]; s_bin[e] = (uint16_t)acc; acc += c; }
                    }
                    __syncthreads();
                    s_perm[s_bin[cls * WAVES + wave] + rank] = (uint16_t)threadIdx.x;
                    __syncthreads();
                    // Stage 3: shade in sorted order; the ray is re-read from the (cache-resident) queue.
                    const uint32_t src = s_perm[threadIdx.x];
                    const uint32_t i2 = base + src;
                    Ray r;
                    v3 T = V(0.0f, 0.0f, 0.0f);
                    uint32_t lid = 0;
                    HitResult h;
                    h.t = 0.0f;
                    h.slot = PT_NO_PRIM;
                    if (i2 < in_count) {
                        float4 A = inA[i2], B = inB[i2];
                        float2 C = inC[i2];
                        uint2 hh = s_hit[src];
                        lid = __float_as_uint(A.w);
                        r.o = V(A.x, A.y, A.z);
                        r.d = V(B.x, B.y, B.z);
                        T = V(B.w, C.x, C.y);
                        h.t = __uint_as_float(hh.x);
                        h.slot = hh.y;
                    }
                    shade_emit(i2 < in_count, r, T, lid, h);
                    __syncthreads(); // s_hit / s_perm are reused by the next chunk
                }
            }
            __syncthreads();
            mark(4);
            if (threadIdx.x == 0) rays += in_count;
            in_lo = s_out[2 * (b & 1)];
            in_count = in_lo + (SPLIT ? s_out[2 * (b & 1) + 1] : 0u);
            cur ^= 1u;
        }
        // film: col += ray_color(...) in sample order (main.cu:93) for the slice's pixels
        __syncthreads();
        for (uint32_t pl = threadIdx.x; pl < npix; pl += BLOCK) {
            const size_t px = (size_t)(pix0 + pl) * 3;
            float r = rgb_sum[px], g = rgb_sum[px + 1], bl = rgb_sum[px + 2];
            for (uint32_t s = 0; s < ps.samples; ++s) {
                const uint32_t lid = s * npix + pl;
                r += rad[3 * lid];
                g += rad[3 * lid + 1];
                bl += rad[3 * lid + 2];
            }
            rgb_sum[px] = r;
            rgb_sum[px + 1] = g;
            rgb_sum[px + 2] = bl;
        }
        mark(5);
    }
    if (kProf) {
        for (int k = 0; k < 6; ++k) {
            const unsigned long long a = wave_sum_u32(prof.lanes[k]), b2 = wave_sum_u32(prof.slots[k]);
            if (lane == 0) { atomicAdd(&cnt->lanes[k], a); atomicAdd(&cnt->slots[k], b2); }
        }
        if (lane == 0)
            for (int k = 0; k < 8; ++k) atomicAdd(&cnt->sect[k], sect[k]);
    }
    if (threadIdx.x == 0 && rays) atomicAdd(&cnt->rays, rays);
    if (COUNT) {
        unsigned long long sb = wave_sum_u32(nb), sp = wave_sum_u32(np);
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&cnt->box_tests, sb);
            atomicAdd(&cnt->prim_tests, sp);
        }
    }
    if (nf) atomicAdd(&cnt->exact_fallbacks, (unsigned long long)nf);
    if (nr) atomicAdd(&cnt->exact_replays, (unsigned long long)nr);
}

// ---------------------------------------------------------------------------------------------
// k_paths_ring: k_paths with the slices of a workgroup overlapped.  In k_paths a workgroup takes ONE slice through all
// its bounces before it starts the next, and the bounces after the first few hold a handful of rays each: a bounce
// costs the workgroup at least the latency of one wave's trace and shade whatever it holds, so the forty-odd tail
// bounces of every slice are paid in full (14 % of all wave time at 64 samples per pixel on the random spheres, where a
// slice is 32k paths; 49 % at 8, where a GPU's share of a strong-scaled frame makes slices of 5k).  Here the queue
// never runs dry: whenever fewer than `low_water` rays are queued at the start of a generation, the workgroup takes
// the next slice and its camera rays join the queue's survivors, so every generation but the last few of the launch
// has several full chunks to work on and the stragglers of earlier slices ride along.  A path carries its own bounce
// count and the ring slot of its slice (up to PT_RING slices are under way per workgroup, each with its own radiance
// buffer and a count of unfinished paths in LDS); a slice whose count reaches zero is summed into the film in sample
// order (main.cu:93) at the start of the next generation.  Same paths, same arithmetic, same film order as k_paths:
// bit-identical images.
// ---------------------------------------------------------------------------------------------
#define PT_RING 16           // slices a workgroup may have under way
#define PT_RING_LID_BITS 17  // a path's index within its slice (sample * pixels + pixel)
#define PT_RING_SLOT_BITS 4
#define PT_RING_MAX_DEPTH ((1 << (32 - PT_RING_LID_BITS - PT_RING_SLOT_BITS)) - 1) // the bounce count has the word's other 11 bits
struct RingDev {
    float4 *qA;   // [grid][2][qcap]
    float4 *qB;
    float2 *qC;
    float *rad;   // [grid][PT_RING][cap * 3]
    uint32_t qcap;      // entries per queue: low_water + cap
    uint32_t cap;       // paths of the largest slice
    uint32_t low_water; // a generation that starts with fewer queued rays than this takes the next slice as well
    uint32_t n_slices;
    const uint2 *extents; // [n_slices] (first pixel, pixels)
    uint32_t *ticket;
};

// Adds a finished slice to the film: film[e] += the samples of rad[sample][e], e = 3 * pixel + channel, one after the
// other in sample order (main.cu:93).  A call of its own, so that its registers are not the path loop's.
static __device__ __attribute__((noinline)) void ring_film(const float *__restrict__ rk, float *__restrict__ film, uint32_t n3, uint32_t samples, uint32_t block)
{
    for (uint32_t e = threadIdx.x; e < n3; e += block) {
        float acc = film[e];
        const float *src = rk + e;
        uint32_t s = 0;
        for (; s + 8 <= samples; s += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(s + u) * n3];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
        for (; s < samples; ++s) acc += src[(size_t)s * n3];
        film[e] = acc;
    }
}

template <bool COUNT, bool LDS, int BLOCK, int KIND, bool SPLIT = false, bool FLAT = false>
__global__ __launch_bounds__(BLOCK, 6) void k_paths_ring(SceneDev sc, PassDev ps, RingDev rg, float *__restrict__ rgb_sum, Counters *cnt)
{
    extern __shared__ float4 s_mem[];
    __shared__ uint32_t s_out[4];                                  // [parity][0: queued from the front, 1: from the back (SPLIT)]
    __shared__ uint32_t s_left[PT_RING], s_pix0[PT_RING], s_npix[PT_RING]; // per ring slot: unfinished paths, the slice's extent
    __shared__ uint32_t s_open, s_fresh_slot, s_fresh_n, s_exhausted, s_chunk;
    __shared__ uint32_t s_next_state, s_next_pix0, s_next_npix;    // the slice taken ahead of time: 0 none asked for, 1 a slice, 2 the table is used up
    uint32_t *s_stack;
    LdsSrc lsrc{nullptr, nullptr, sc.n_nodes, sc.n_prims, sc.cone};
    if (LDS) {
        s_stack = reinterpret_cast<uint32_t *>(lds_scene_copy<BLOCK>(sc, s_mem, lsrc));
    } else {
        s_stack = reinterpret_cast<uint32_t *>(s_mem);
    }
    GlobalSrc gsrc{sc.nodes, sc.prims, sc.cone};
    const uint32_t qcap = rg.qcap, cap = rg.cap;
    float4 *const qA = rg.qA + (size_t)blockIdx.x * 2 * qcap;
    float4 *const qB = rg.qB + (size_t)blockIdx.x * 2 * qcap;
    float2 *const qC = rg.qC + (size_t)blockIdx.x * 2 * qcap;
    float *const rad = rg.rad + (size_t)blockIdx.x * PT_RING * 3 * cap;
    const float t_min = 0.001f /* main.cu:54 */, t_max = PT_FLT_MAX /* commons.h:11 */;
    const uint32_t lane = threadIdx.x & 63u;
    unsigned long long rays = 0;
    uint32_t nb = 0, np = 0, nf = 0, nr = 0;
    uint32_t extra_rays = 0; // rays traced from a lane without passing through a queue (the end of the launch)
    ProfLanes prof;
    unsigned long long sect[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long t_last = kProf ? clock64() : 0;
    auto mark = [&](int k) {
        if (kProf) {
            const long long now = clock64();
            sect[k] += (unsigned long long)(now - t_last);
            t_last = now;
        }
    };
    // The next slice is asked for one generation ahead by one lane, which waits for the ticket and the slice's extent
    // while the other waves work; the workgroup never waits for the counter in HBM.
    auto take_ahead = [&]() {
        const uint32_t ticket = atomicAdd(rg.ticket, 1u);
        if (ticket < rg.n_slices) {
            const uint2 e = rg.extents[ticket];
            s_next_pix0 = e.x;
            s_next_npix = e.y;
            s_next_state = 1;
        } else {
            s_next_state = 2;
        }
    };
    if (threadIdx.x == 0) {
        s_out[0] = s_out[1] = s_out[2] = s_out[3] = 0;
        s_open = 0; s_exhausted = 0; s_fresh_n = 0; s_fresh_slot = 0; s_chunk = 0;
        take_ahead();
    }
    constexpr uint32_t LID_MASK = (1u << PT_RING_LID_BITS) - 1u, SLOT_MASK = (1u << PT_RING_SLOT_BITS) - 1u, ALL_OPEN = (1u << PT_RING) - 1u;
    constexpr uint32_t BOUNCE_ONE = 1u << (PT_RING_LID_BITS + PT_RING_SLOT_BITS);
    uint32_t cur = 0;
    uint32_t summed = 0; // slices summed into the film during the last generation: their slots are free from this one on
    long long t_drain = 0;
    for (;;) {
        __syncthreads(); // (A) the last generation's queue entries, radiance and counts are in place (the first time: the scene copy)
        mark(4);
        const uint32_t n_lo = s_out[2 * cur], n_hi = SPLIT ? s_out[2 * cur + 1] : 0u, n_queue = n_lo + n_hi;
        const uint32_t open = s_open;
        uint32_t closing = 0; // slices whose last path ended in the last generation
        for (uint32_t m = open & ~summed; m; m &= m - 1u) {
            const uint32_t k = (uint32_t)__ffs((int)m) - 1u;
            if (s_left[k] == 0) closing |= 1u << k;
        }
        __syncthreads(); // (B) every thread has read what thread 0 changes now
        if (threadIdx.x == 0) {
            uint32_t now_open = open & ~summed, fresh = 0;
            // the next slice joins when the queue runs low, into a slot that was free before this generation
            if (!s_exhausted && n_queue < rg.low_water && open != ALL_OPEN) {
                if (s_next_state == 2) {
                    s_exhausted = 1;
                } else {
                    const uint32_t slot = (uint32_t)__ffs((int)~open) - 1u;
                    s_pix0[slot] = s_next_pix0;
                    s_npix[slot] = s_next_npix;
                    fresh = s_next_npix * ps.samples;
                    s_left[slot] = fresh;
                    s_fresh_slot = slot;
                    now_open |= 1u << slot;
                    s_next_state = 0;
                }
            }
            s_fresh_n = fresh;
            s_open = now_open;
            s_out[2 * (cur ^ 1u)] = s_out[2 * (cur ^ 1u) + 1] = 0;
            s_chunk = 0;
        }
        __syncthreads(); // (C) thread 0's decisions are visible
        mark(0);
        const uint32_t fresh_n = s_fresh_n, fresh_slot = s_fresh_slot;
        const uint32_t fresh_pix0 = s_pix0[fresh_slot], fresh_npix = s_npix[fresh_slot];
        const uint32_t total = n_queue + fresh_n;
        const bool last = total == 0 && s_exhausted != 0; // nothing queued, nothing left to take
        // The end of the launch: no slice left to take and at most one chunk per wave queued.  Every wave then keeps the
        // rays of its chunk in its lanes until the last of them has ended -- no queue, no barrier between their bounces.
        // (A path of fifty bounces is fifty generations; with a handful of rays each, a generation is all latency.)
        const bool drain = s_exhausted != 0 && total <= (uint32_t)BLOCK;
        if (threadIdx.x == BLOCK - 1 && fresh_n) take_ahead(); // (the last wave's: with chunks dealt out on demand it just takes fewer)
        const float4 *inA = qA + cur * qcap, *inB = qB + cur * qcap;
        const float2 *inC = qC + cur * qcap;
        float4 *outA = qA + (cur ^ 1u) * qcap, *outB = qB + (cur ^ 1u) * qcap;
        float2 *outC = qC + (cur ^ 1u) * qcap;
        uint32_t *const out_cnt = &s_out[2 * (cur ^ 1u)];
        // Chunks of 64 rays are dealt out to the waves on demand (one LDS atomic each): rays differ several-fold in cost, and
        // with a fixed share per wave the workgroup would wait at the end of every generation for the wave that drew the dearest.
        for (;;) {
            uint32_t c0 = 0;
            if (lane == 0) c0 = atomicAdd(&s_chunk, 64u);
            c0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)c0);
            if (c0 >= total) break;
            const uint32_t i = c0 + lane;
            const bool valid = i < total;
            Ray r;
            r.o = V(0.0f, 0.0f, 0.0f);
            r.d = V(0.0f, 0.0f, 0.0f);
            v3 T = V(1.0f, 1.0f, 1.0f);
            uint32_t idw = 0; // path index within its slice | ring slot | bounces so far
            HitResult h;
            h.t = 0.0f;
            h.slot = PT_NO_PRIM;
            bool valid_now = valid;
            if (valid) {
                if (i < n_queue) {
                    const uint32_t qi = SPLIT && i >= n_lo ? qcap - 1u - (i - n_lo) : i; // (the rays queued from the back)
                    const float4 A = inA[qi], B = inB[qi];
                    const float2 C = inC[qi];
                    idw = __float_as_uint(A.w);
                    r.o = V(A.x, A.y, A.z);
                    r.d = V(B.x, B.y, B.z);
                    T = V(B.w, C.x, C.y);
                } else {
                    // the new slice's camera rays, numbered sample-fast: a wave starts with the samples of one or a few pixels
                    const uint32_t f = i - n_queue;
                    const uint32_t pl = f / ps.samples, s = f - pl * ps.samples;
                    idw = (s * fresh_npix + pl) | (fresh_slot << PT_RING_LID_BITS);
                    r = camera_ray(sc.cam, ps, fresh_pix0 + pl, ps.sample_base + s);
                }
            }
            mark(1);
            bool survive = false;
            float4 oA, oB;
            float2 oC;
            for (;;) { // (once, except at the end of the launch)
            h.slot = PT_NO_PRIM;
            if (valid_now) {
                bool need_exact;
                const int32_t root_ref = FLAT ? sc.root_ref : 0;
                if (LDS) h = bvh_nearest<COUNT, KIND, BLOCK, LdsSrc, FLAT>(lsrc, sc.stack_cap, r, t_min, t_max, s_stack + threadIdx.x, need_exact, nb, np, nr, &prof, root_ref);
                else h = bvh_nearest<COUNT, KIND, BLOCK, GlobalSrc, FLAT>(gsrc, sc.stack_cap, r, t_min, t_max, s_stack + threadIdx.x, need_exact, nb, np, nr, &prof, root_ref);
                if (need_exact) { // resolved on the list-order records; the top bit of the slot says which array it indexes
                    h = linear_nearest(sc.prims_list, sc.n_prims, r, t_min, t_max);
                    if (h.slot != PT_NO_PRIM) h.slot |= 0x80000000u;
                    nf++;
                }
            }
            mark(2);
            // main.cu:54-64 for the traced ray, then compaction of the wave's survivors into the next queue
            survive = false;
            if (valid_now) {
                const uint32_t lid = idw & LID_MASK, slot = (idw >> PT_RING_LID_BITS) & SLOT_MASK;
                v3 result;
                if (h.slot == PT_NO_PRIM) {
                    result = vmul(T, V(sc.bg[0], sc.bg[1], sc.bg[2])); // main.cu:54-55
                } else {
                    PrimRec q;
                    if (LDS && (KIND == PT_KIND_SPHERES || (h.slot & 0x80000000u) == 0)) {
                        float4 w0, w1;
                        lsrc.prim(h.slot, w0, w1);
                        q.p[0] = w0.x; q.p[1] = w0.y; q.p[2] = w0.z; q.p[3] = w0.w; q.p[4] = w1.x; q.p[5] = w1.y;
                        q.meta = __float_as_uint(w1.z);
                        q.id = __float_as_uint(w1.w);
                    } else {
                        q = load_prim((h.slot & 0x80000000u) ? sc.prims_list : sc.prims, h.slot & 0x7FFFFFFFu);
                    }
                    const uint32_t npix = s_npix[slot];
                    const uint32_t s = lid / npix, pl = lid - s * npix;
                    Ray nray;
                    v3 nT;
                    if (kProf) prof.tick(3);
                    survive = shade_hit<KIND == PT_KIND_SPHERES>(sc.mats, ps, r, T, s_pix0[slot] + pl, ps.sample_base + s, (int32_t)(idw >> (PT_RING_LID_BITS + PT_RING_SLOT_BITS)), q, h.t,
                                                                 nray, nT, result, &prof);
                    if (survive) {
                        oA = make_float4(nray.o.x, nray.o.y, nray.o.z, __uint_as_float(idw + BOUNCE_ONE));
                        oB = make_float4(nray.d.x, nray.d.y, nray.d.z, nT.x);
                        oC = make_float2(nT.y, nT.z);
                    }
                }
                if (!survive) {
                    float *rk = rad + ((size_t)slot * cap + lid) * 3;
                    rk[0] = result.x;
                    rk[1] = result.y;
                    rk[2] = result.z;
                    atomicSub(&s_left[slot], 1u);
                }
            }
            if (!drain || __ballot(survive) == 0ull) break;
            // the end of the launch: the survivors go on from their lanes
            valid_now = survive;
            if (survive) {
                r.o = V(oA.x, oA.y, oA.z);
                r.d = V(oB.x, oB.y, oB.z);
                T = V(oB.w, oC.x, oC.y);
                idw = __float_as_uint(oA.w);
                extra_rays++;
            }
            }
            if (SPLIT) { // as in k_paths: rays that head up (camera's up vector) from the back of the next queue, the others from its front
                const bool up = oB.x * sc.cam.v[0] + oB.y * sc.cam.v[1] + oB.z * sc.cam.v[2] > 0.0f;
                const unsigned long long m0 = __ballot(survive && !up), m1 = __ballot(survive && up);
                if (m0 | m1) {
                    const unsigned long long mine = up ? m1 : m0;
                    const uint32_t pre = lane_prefix(mine);
                    uint32_t pos0 = 0;
                    if (survive && pre == 0) pos0 = atomicAdd(&out_cnt[up ? 1 : 0], (uint32_t)__popcll(mine));
                    const uint32_t p0 = __shfl(pos0, m0 ? __ffsll((long long)m0) - 1 : 0, 64), p1 = __shfl(pos0, m1 ? __ffsll((long long)m1) - 1 : 0, 64);
                    if (survive) {
                        const uint32_t at = up ? qcap - 1u - (p1 + pre) : p0 + pre;
                        outA[at] = oA;
                        outB[at] = oB;
                        outC[at] = oC;
                    }
                }
            } else {
                const unsigned long long mask = __ballot(survive);
                if (mask) {
                    const uint32_t pre = lane_prefix(mask);
                    uint32_t pos0 = 0;
                    if (survive && pre == 0) pos0 = atomicAdd(&out_cnt[0], (uint32_t)__popcll(mask));
                    pos0 = __shfl(pos0, __ffsll((long long)mask) - 1, 64);
                    if (survive) {
                        outA[pos0 + pre] = oA;
                        outB[pos0 + pre] = oB;
                        outC[pos0 + pre] = oC;
                    }
                }
            }
            mark(3);
        }
        // film: col += ray_color(...) in sample order (main.cu:93) for the pixels of the slices that finished in the last
        // generation, by every wave as it runs out of chunks (their slots stay taken until the next generation)
        for (uint32_t m = closing; m; m &= m - 1u) {
            const uint32_t k = (uint32_t)__ffs((int)m) - 1u;
            // one thread per (pixel, colour channel): the samples of a sum are still added one after the other, in sample
            // order, but three times as many threads work and the loads of eight samples are under way together
            ring_film(rad + (size_t)k * 3 * cap, rgb_sum + (size_t)s_pix0[k] * 3, 3u * s_npix[k], ps.samples, BLOCK);
        }
        mark(5);
        if (kProf && s_exhausted) { // diagnostic build: generations after the last slice was taken, and how many rays they held
            if (t_drain == 0) t_drain = clock64();
            if (threadIdx.x == 0) { sect[7] += 1; prof.lanes[5] += total; }
        }
        if (last) break; // (every slice still open had no path left: summed just now or a generation ago)
        if (threadIdx.x == 0) rays += total;
        summed = closing;
        cur ^= 1u;
    }
    if (kProf && t_drain) sect[6] += (unsigned long long)(clock64() - t_drain);
    if (kProf) {
        for (int k = 0; k < 6; ++k) {
            const unsigned long long a = wave_sum_u32(prof.lanes[k]), b2 = wave_sum_u32(prof.slots[k]);
            if (lane == 0) { atomicAdd(&cnt->lanes[k], a); atomicAdd(&cnt->slots[k], b2); }
        }
        if (lane == 0)
            for (int k = 0; k < 8; ++k) atomicAdd(&cnt->sect[k], sect[k]);
    }
    {
        const unsigned long long ex = wave_sum_u32(extra_rays);
        if (lane == 0) rays += ex;
    }
    if (lane == 0 && rays) atomicAdd(&cnt->rays, rays);
    if (COUNT) {
        unsigned long long sb = wave_sum_u32(nb), sp = wave_sum_u32(np);
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&cnt->box_tests, sb);
            atomicAdd(&cnt->prim_tests, sp);
        }
    }
    if (nf) atomicAdd(&cnt->exact_fallbacks, (unsigned long long)nf);
    if (nr) atomicAdd(&cnt->exact_replays, (unsigned long long)nr);
}

// ---------------------------------------------------------------------------------------------
// k_paths_stream: the slice pipeline for scenes traversed from HBM, where every traversal step waits on
// memory and rays differ widely in cost (1..5000 node visits on the 1M-sphere scene).  In k_paths a wave
// traces 64 rays and waits for the slowest of them, and a workgroup waits at the end of every bounce for
// its slowest ray: there 13 % of the lanes of a node visit are active, 60 % of the waves' time is spent at
// that barrier, and a slice's critical path is the SUM over its bounces of the longest ray.  Here a lane
// keeps a path from its camera ray to its end, and nothing is synchronised before the slice is finished:
//  * a traversal step is the same for every lane whatever it holds: fetch the 64 bytes it needs next --
//    an inner node, or the next two objects of its leaf -- then act on them; one memory round trip
//    serves all lanes;
//  * lanes whose traversal has finished wait until `serve_min` of them have (or nothing else is left to
//    do); then they are shaded together -- the path scatters and the lane starts the next traversal, or
//    the path ends and the lane takes the slice's next path (camera ray made on the spot).
// There are no ray queues.  Same paths, same arithmetic, same film order: images are bit-identical to
// k_paths.  (The vector ALUs are mostly idle on these scenes, so the lower lane utilisation of shading
// in small groups costs nothing that matters.)
// ---------------------------------------------------------------------------------------------
#ifndef PT_STREAM_WAVES
#define PT_STREAM_WAVES 4 // waves per SIMD of every k_paths_stream variant (117-128 VGPRs)
#endif
// One pass of k_paths_stream: `samples` samples (PassDev) of the pixels [pix0, pix0 + npix).
struct BandDev {
    uint32_t pix0, npix;
    uint32_t tile; // pixels per tile of the path numbering (a multiple of 64)
    uint32_t run;  // consecutive paths a wave takes at a time
};

template <bool COUNT, int BLOCK, int KIND, bool WIDE>
__global__ __launch_bounds__(BLOCK, PT_STREAM_WAVES) void k_paths_stream(SceneDev sc, PassDev ps, BandDev band, float *__restrict__ rad /* [samples][band pixels][3] of this pass */,
                                                                           uint32_t *__restrict__ ticket, Counters *cnt,
                                                                           uint32_t serve_min /* lanes waiting for shading or a new path before they are served */, uint32_t phase_bias /* 0: node and leaf steps in every iteration */,
                                                                           unsigned long long *dbg = nullptr /* diagnostic build only */, uint32_t dbg_waves = 0)
{
    constexpr bool CONES = KIND == PT_KIND_ANY, SPH = KIND == PT_KIND_SPHERES;
    extern __shared__ float4 s_mem[];
    uint32_t *const stack = reinterpret_cast<uint32_t *>(s_mem) + threadIdx.x;
    if (kProf && dbg && (threadIdx.x & 63u) == 0) dbg[2 * (blockIdx.x * (BLOCK / 64) + threadIdx.x / 64)] = wall_clock64(); // diagnostic build: each wave's start
    GlobalSrc gsrc{sc.nodes, sc.prims, sc.cone};
    const float t_min = 0.001f /* main.cu:54 */, t_max = PT_FLT_MAX /* commons.h:11 */;
    const uint32_t lane = threadIdx.x & 63u;
    unsigned long long rays = 0;
    uint32_t nb = 0, np = 0, nf = 0, nr = 0;
    // Paths are numbered tile by tile (`tile` neighbouring pixels) and within a tile sample by sample.  A
    // wave owns a run of `run` consecutive paths at a time, fills its free lanes from it and takes the next
    // run of the pass from one counter when it is used up; nothing else is shared between waves, and
    // nothing waits for a run to finish.  The lanes of a wave therefore work on the samples of the same few
    // pixels, whose rays meet the same part of the tree: with lanes filled from one pass-wide sequence
    // instead, the 1M-sphere scene ran 25 % slower.
    uint32_t dbg_iter = 0, dbg_steps = 0, dbg_maxsteps = 0, dbg_serves = 0; // diagnostic build: loop iterations, steps of the current / longest ray, serves
    ProfLanes prof; // diagnostic build: issues and active lanes of 0 four-wide steps, 1 two-child steps, 2 leaf steps, 3 shading, 4 new paths
    const uint32_t npix = band.npix;
    const uint32_t n_paths = npix * ps.samples;
    const uint32_t tile_paths = band.tile * ps.samples;
    const uint32_t n_runs = (n_paths + band.run - 1) / band.run;
    uint32_t run_next = 0, run_end = 0; // wave-uniform: what is left of the wave's run
    // pixel within the band and sample within the pass of path p
    auto path_of = [&](uint32_t p, uint32_t &pl, uint32_t &s) {
        const uint32_t t = p / tile_paths, rem = p - t * tile_paths;
        const uint32_t tp = min(band.tile, npix - t * band.tile); // the band's last tile may be short
        s = rem / tp;
        pl = t * band.tile + (rem - s * tp);
    };
    {
        uint32_t lid = PT_NO_PRIM;          // where this lane's path reports: sample * npix + pixel within the band
        int32_t bounce = 0;
        v3 T = V(1.0f, 1.0f, 1.0f);         // throughput of the path so far
        int32_t node = PT_TRAVERSAL_DONE;   // inner node (>= 0), the rest of a leaf (< 0), or no traversal under way
        uint32_t sp = 0;
        Ray r;
        r.o = V(0.0f, 0.0f, 0.0f);
        r.d = V(1.0f, 1.0f, 1.0f);
        BoxRay br = make_box_ray(r, sc.cone.kappa_max);
        Nearest nh;
        nh.reset(t_max);
        bool by_list = false;   // this ray is answered by the list scan: non-finite ray, a hit at t = NaN, or its stack ran out (cannot happen)
        bool exhausted = false; // wave-uniform: the pass has no unstarted path left
        auto start_ray = [&]() {
            br = make_box_ray(r, ray_kappa(sc.cone, r.o));
            nh.reset(t_max);
            by_list = false;
            stack[0] = (uint32_t)PT_TRAVERSAL_DONE; // the bottom entry says "done": a pop needs no test for an empty stack
            sp = 1;
            node = 0; // root
            rays++;
            // A ray with a NaN or infinite component (the reference accepts hits whose t is NaN, and the
            // path goes on from a NaN point): the box test cannot prune for it, and what the scan accepts
            // along it depends on list order throughout.  The list itself answers it.
            if (!(fabsf(r.o.x) + fabsf(r.o.y) + fabsf(r.o.z) + fabsf(r.d.x) + fabsf(r.d.y) + fabsf(r.d.z) < PT_FLT_MAX)) {
                by_list = true;
                node = PT_TRAVERSAL_DONE;
            }
        };
        for (;;) {
            if (kProf) dbg_iter++;
            const bool resting = node == PT_TRAVERSAL_DONE;
            const bool to_shade = resting && lid != PT_NO_PRIM;
            const bool empty = lid == PT_NO_PRIM;
            const uint32_t n_shade = (uint32_t)__popcll(__ballot(to_shade));
            const uint32_t n_empty = exhausted ? 0u : (uint32_t)__popcll(__ballot(empty));
            const bool nobody_traverses = __ballot(!resting) == 0ull;
            if (n_shade + n_empty >= serve_min || nobody_traverses) {
                if (kProf) dbg_serves++;
                if (kProf && to_shade) {
                    if (dbg && dbg_steps > 50000u) { // one of the very long rays, for the report
                        float *f = reinterpret_cast<float *>(dbg + 4 * dbg_waves);
                        f[0] = r.o.x; f[1] = r.o.y; f[2] = r.o.z; f[3] = r.d.x; f[4] = r.d.y; f[5] = r.d.z; f[6] = nh.best_t; f[7] = (float)dbg_steps;
                        f[8] = (float)bounce; f[9] = __uint_as_float(nh.best_slot); f[10] = ray_kappa(sc.cone, r.o); f[11] = br.eps;
                    }
                    dbg_maxsteps = max(dbg_maxsteps, dbg_steps); dbg_steps = 0;
                }
                HitResult h;
                h.t = nh.best_t;
                h.slot = nh.best_slot;
                bool scan_list = false;
                bool fresh = false; // this lane leaves the serve with a new ray (scattered, or the first of a new path)
                if (to_shade) {
                    bool need_exact = by_list;
                    settle_parked<CONES>(nh, need_exact);
                    h.t = nh.best_t;
                    h.slot = nh.best_slot;
                    if (need_exact) {
                        if (CONES && !by_list) {
                            h = bvh_exact_scan<BLOCK>(gsrc, sc.stack_cap, r, br, t_min, t_max, stack, need_exact);
                            nr++;
                        }
                        scan_list = need_exact; // by_list, or more candidates than the replay keeps
                    }
                }
                // the list itself, one such ray at a time by the whole wave
                for (unsigned long long m = __ballot(scan_list); m; m &= m - 1ull) {
                    const int j = __ffsll((long long)m) - 1;
                    const float jox = __shfl(r.o.x, j, 64), joy = __shfl(r.o.y, j, 64), joz = __shfl(r.o.z, j, 64);
                    const float jdx = __shfl(r.d.x, j, 64), jdy = __shfl(r.d.y, j, 64), jdz = __shfl(r.d.z, j, 64);
                    const unsigned long long hj = CONES ? wave_list_scan_call(sc.prims_list, sc.n_prims, jox, joy, joz, jdx, jdy, jdz, t_min, t_max)
                                                        : wave_list_scan(sc.prims_list, sc.n_prims, jox, joy, joz, jdx, jdy, jdz, t_min, t_max);
                    if (lane == (uint32_t)j) {
                        h.t = __uint_as_float((uint32_t)hj);
                        h.slot = (uint32_t)(hj >> 32);
                        if (h.slot != PT_NO_PRIM) h.slot |= 0x80000000u;
                        nf++;
                    }
                }
                if (to_shade) { // main.cu:54-64 for the ray just traced
                    prof.tick(3);
                    bool go_on = false;
                    v3 result;
                    if (h.slot == PT_NO_PRIM) {
                        result = vmul(T, V(sc.bg[0], sc.bg[1], sc.bg[2])); // main.cu:54-55
                    } else {
                        const PrimRec q = load_prim((h.slot & 0x80000000u) ? sc.prims_list : sc.prims, h.slot & 0x7FFFFFFFu);
                        const uint32_t s = lid / npix, pl = lid - s * npix;
                        Ray nray;
                        v3 nT;
                        go_on = shade_hit<SPH>(sc.mats, ps, r, T, band.pix0 + pl, ps.sample_base + s, bounce, q, h.t, nray, nT, result);
                        if (go_on) {
                            r = nray;
                            T = nT;
                            bounce++;
                            if (CONES) start_ray(); else fresh = true;
                        }
                    }
                    if (!go_on) {
                        rad[3 * (size_t)lid + 0] = result.x;
                        rad[3 * (size_t)lid + 1] = result.y;
                        rad[3 * (size_t)lid + 2] = result.z;
                        lid = PT_NO_PRIM;
                    }
                }
                if (!exhausted) { // lanes without a path take the next ones of the wave's run
                    const bool need = lid == PT_NO_PRIM;
                    const unsigned long long m = __ballot(need);
                    if (m) {
                        if (run_next == run_end) { // the wave's run is used up: the next one of the pass
                            uint32_t g = 0;
                            if (lane == (uint32_t)(__ffsll((long long)m) - 1)) g = atomicAdd(ticket, 1u);
                            g = (uint32_t)__builtin_amdgcn_readfirstlane((int)__shfl(g, __ffsll((long long)m) - 1, 64));
                            if (g >= n_runs) {
                                exhausted = true;
                            } else {
                                run_next = g * band.run;
                                run_end = min(run_next + band.run, n_paths);
                            }
                        }
                        const uint32_t k = lane_prefix(m);
                        if (need && k < run_end - run_next) {
                            prof.tick(4);
                            uint32_t s, pl;
                            path_of(run_next + k, pl, s);
                            lid = s * npix + pl;
                            r = camera_ray(sc.cam, ps, band.pix0 + pl, ps.sample_base + s); // main.cu:88-92
                            T = V(1.0f, 1.0f, 1.0f);
                            bounce = 0;
                            if (CONES) start_ray(); else fresh = true;
                        }
                        run_next += min((uint32_t)__popcll(m), run_end - run_next); // (lanes left without a path come back at once)
                    }
                }
                // once for both kinds of new ray: the slab constants are a sixth of a serve's instructions (1M spheres +4.6 %).  The
                // cone variant has no register for the flag across the serve and keeps the two sites (0.7 % slower merged).
                if (!CONES && fresh) start_ray();
                if (__ballot(lid != PT_NO_PRIM) == 0ull) break; // no path left in this wave, none to take
                continue;
            }
            // Traversal steps follow each other in a loop of their own until a serve is due (the same moment the test at the
            // top of the outer loop would find): the outer loop merges the state a serve writes -- ray, slab constants,
            // nearest hit, parked cone -- with the state a step writes, a dozen register moves per iteration that the inner
            // loop does not pay (100k mixed objects: 31 % fewer vector instructions, +15 %; 1M spheres: 10 % fewer, +2 %
            // at four waves per SIMD, where the loop fits the registers without spilling).
            // The record a lane needs next is asked for as soon as its reference is known -- at the end of the step that
            // produced it, before the votes and tests of the next iteration -- and waits in pw0..pw3 for the lane's turn
            // (100k mixed objects +6.6 %).
            float4 pw0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), pw1 = pw0, pw2 = pw0, pw3 = pw0;
            auto fetch = [&]() {
                const float4 *rec = node >= 0 ? (WIDE ? sc.wide + 4 * (size_t)node : reinterpret_cast<const float4 *>(sc.nodes + node))
                                              : reinterpret_cast<const float4 *>(sc.prims + ((uint32_t)~node >> 3));
                pw0 = rec[0]; pw1 = rec[1]; pw2 = rec[2]; pw3 = rec[3]; // (the object array is padded by one record)
            };
            if (node != PT_TRAVERSAL_DONE) fetch();
#pragma nounroll
            for (bool first_step = true;; first_step = false) {
            if (!first_step) {
                const unsigned long long rest = __ballot(node == PT_TRAVERSAL_DONE && (!exhausted || lid != PT_NO_PRIM));
                if ((uint32_t)__popcll(rest) >= serve_min || __ballot(node != PT_TRAVERSAL_DONE) == 0ull) break;
            }
            // One kind of step per iteration, the one more lanes are waiting for: the two kinds share no code, and
            // run in the same iteration each would execute with the lanes of the other masked off.
            bool my_turn = true;
            if (phase_bias) {
                const uint32_t n_inner = (uint32_t)__popcll(__ballot(node >= 0)), n_leaf = (uint32_t)__popcll(__ballot(node < 0 && node != PT_TRAVERSAL_DONE));
                const bool inner_turn = n_inner * phase_bias >= n_leaf * 4u; // phase_bias = 4: plain majority
                my_turn = (node >= 0) == inner_turn;
            }
            if (node != PT_TRAVERSAL_DONE && my_turn) {
                if (kProf) dbg_steps++;
                const bool inner = node >= 0;
                const uint32_t ref = (uint32_t)~node; // leaf: (first object << 3) | (objects left - 1)
                const float4 w0 = pw0, w1 = pw1, w2 = pw2, w3 = pw3;
                if (WIDE && inner) {
                    prof.tick(0);
                    // pt::WideNode: origin, step exponents, the four children's planes as bytes per axis, their references
                    const uint32_t eb = __float_as_uint(w0.w);
                    const float stx = __uint_as_float((eb & 0xFFu) << 23), sty = __uint_as_float((eb & 0xFF00u) << 15), stz = __uint_as_float((eb & 0xFF0000u) << 7);
                    const uint32_t lox = __float_as_uint(w1.x), loy = __float_as_uint(w1.y), loz = __float_as_uint(w1.z);
                    const uint32_t hix = __float_as_uint(w1.w), hiy = __float_as_uint(w2.x), hiz = __float_as_uint(w2.y);
                    // the planes the ray meets first / last on each axis, by its direction signs
                    const uint32_t qnx = br.sx ? hix : lox, qfx = br.sx ? lox : hix;
                    const uint32_t qny = br.sy ? hiy : loy, qfy = br.sy ? loy : hiy;
                    const uint32_t qnz = br.sz ? hiz : loz, qfz = br.sz ? loz : hiz;
                    uint32_t key[4]; // distance at which the ray's cone enters the child's box (its low two bits give way to the child's slot)
                    int32_t cr[4] = {__float_as_int(w2.z), __float_as_int(w2.w), __float_as_int(w3.x), __float_as_int(w3.y)};
#ifdef PT_WIDE_FOLD
                    // The node's grid folded into the slab constants: a plane distance is (origin + q step) A + C = q (step A) + (origin A + C),
                    // one FMA per plane after twelve operations per node instead of two per plane.  step is a power of two, so step A is
                    // exact; origin A + C is rounded once more than in the unfolded form, by at most 2^-24 of its own magnitude (which the
                    // plane distance itself need not share: the node's origin can be far from a plane on the ray's scale), and four times
                    // that much is added to the allowance of the comparison.  The far constants of an axis are its near ones times
                    // (1 + w) / (1 - w) (make_box_ray), so the largest near term and the ray's largest such ratio (eps_far) bound them all;
                    // an axis whose far slot is open has the constants (0, 3e38) and no rounding.
                    const float sanx = stx * br.anx, sany = sty * br.any, sanz = stz * br.anz, safx = stx * br.afx, safy = sty * br.afy, safz = stz * br.afz;
                    const float onx = __builtin_fmaf(w0.x, br.anx, br.cnx), ony = __builtin_fmaf(w0.y, br.any, br.cny), onz = __builtin_fmaf(w0.z, br.anz, br.cnz);
                    const float ofx = __builtin_fmaf(w0.x, br.afx, br.cfx), ofy = __builtin_fmaf(w0.y, br.afy, br.cfy), ofz = __builtin_fmaf(w0.z, br.afz, br.cfz);
                    float omax;
                    asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(omax) : "v"(onx), "v"(ony), "v"(onz));
                    const float eps_node = __builtin_fmaf(omax, br.eps_far, br.eps);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float nx = __builtin_fmaf((float)((qnx >> (8 * k)) & 0xFFu), sanx, onx), fx = __builtin_fmaf((float)((qfx >> (8 * k)) & 0xFFu), safx, ofx);
                        const float ny = __builtin_fmaf((float)((qny >> (8 * k)) & 0xFFu), sany, ony), fy = __builtin_fmaf((float)((qfy >> (8 * k)) & 0xFFu), safy, ofy);
                        const float nz = __builtin_fmaf((float)((qnz >> (8 * k)) & 0xFFu), sanz, onz), fz = __builtin_fmaf((float)((qfz >> (8 * k)) & 0xFFu), safz, ofz);
                        float m, tn, tf;
                        asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(nz), "v"(t_min));
                        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(tn) : "v"(nx), "v"(ny), "v"(m));
                        asm("v_min_f32 %0, %1, %2" : "=v"(m) : "v"(fz), "v"(nh.best_t));
                        asm("v_min3_f32 %0, %1, %2, %3" : "=v"(tf) : "v"(fx), "v"(fy), "v"(m));
                        const bool hit = tn <= tf + eps_node && cr[k] != pt::kWideAbsent;
                        key[k] = ((__float_as_uint(tn) & ~3u) | (uint32_t)k) | (hit ? 0u : 0xFFFFFFFFu); // tn >= t_min > 0: its bits order as it does
                    }
#else
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        // plane = origin + q * step exactly (the builder's grid guarantees it), then the slab test of box_test2
                        const float pnx = __builtin_fmaf((float)((qnx >> (8 * k)) & 0xFFu), stx, w0.x), pfx = __builtin_fmaf((float)((qfx >> (8 * k)) & 0xFFu), stx, w0.x);
                        const float pny = __builtin_fmaf((float)((qny >> (8 * k)) & 0xFFu), sty, w0.y), pfy = __builtin_fmaf((float)((qfy >> (8 * k)) & 0xFFu), sty, w0.y);
                        const float pnz = __builtin_fmaf((float)((qnz >> (8 * k)) & 0xFFu), stz, w0.z), pfz = __builtin_fmaf((float)((qfz >> (8 * k)) & 0xFFu), stz, w0.z);
                        const float nx = __builtin_fmaf(pnx, br.anx, br.cnx), ny = __builtin_fmaf(pny, br.any, br.cny), nz = __builtin_fmaf(pnz, br.anz, br.cnz);
                        const float fx = __builtin_fmaf(pfx, br.afx, br.cfx), fy = __builtin_fmaf(pfy, br.afy, br.cfy), fz = __builtin_fmaf(pfz, br.afz, br.cfz);
                        float m, tn, tf;
                        asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(nz), "v"(t_min));
                        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(tn) : "v"(nx), "v"(ny), "v"(m));
                        asm("v_min_f32 %0, %1, %2" : "=v"(m) : "v"(fz), "v"(nh.best_t));
                        asm("v_min3_f32 %0, %1, %2, %3" : "=v"(tf) : "v"(fx), "v"(fy), "v"(m));
                        const bool hit = tn <= tf + br.eps && cr[k] != pt::kWideAbsent;
                        key[k] = ((__float_as_uint(tn) & ~3u) | (uint32_t)k) | (hit ? 0u : 0xFFFFFFFFu); // tn >= t_min > 0: its bits order as it does
                    }
#endif
                    if (COUNT) nb += 4;
                    if (br.afx == 0.0f || br.afy == 0.0f || br.afz == 0.0f) { // a steep axis (rare): the bound its open far slot leaves out
                        const SteepBounds sb = make_steep_bounds(r, ray_kappa(sc.cone, r.o));
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float pfx = __builtin_fmaf((float)((qfx >> (8 * k)) & 0xFFu), stx, w0.x);
                            const float pfy = __builtin_fmaf((float)((qfy >> (8 * k)) & 0xFFu), sty, w0.y);
                            const float pfz = __builtin_fmaf((float)((qfz >> (8 * k)) & 0xFFu), stz, w0.z);
                            const float tf = fminf(fminf(__builtin_fmaf(pfx, br.afx, br.cfx), __builtin_fmaf(pfy, br.afy, br.cfy)),
                                                   fminf(__builtin_fmaf(pfz, br.afz, br.cfz), nh.best_t));
                            if (steep_rejects(sb, pfx, pfy, pfz, tf, br.eps)) key[k] = 0xFFFFFFFFu;
                        }
                    }
                    // nearest first: a five-exchange sorting network on the keys, the children's references moving with them
                    // (selecting a reference by a sorted key's slot bits afterwards costs more than the ten selects here)
                    auto order2 = [](uint32_t &a, uint32_t &b, int32_t &ra, int32_t &rb) {
                        const bool s = b < a;
                        const uint32_t lo = s ? b : a, hi = s ? a : b;
                        const int32_t rlo = s ? rb : ra, rhi = s ? ra : rb;
                        a = lo; b = hi; ra = rlo; rb = rhi;
                    };
                    order2(key[0], key[1], cr[0], cr[1]); order2(key[2], key[3], cr[2], cr[3]); order2(key[0], key[2], cr[0], cr[2]);
                    order2(key[1], key[3], cr[1], cr[3]); order2(key[1], key[2], cr[1], cr[2]);
                    const uint32_t n_hit = (key[0] != 0xFFFFFFFFu) + (key[1] != 0xFFFFFFFFu) + (key[2] != 0xFFFFFFFFu) + (key[3] != 0xFFFFFFFFu);
                    if (n_hit == 0) {
                        node = (int32_t)stack[(--sp) * BLOCK];
                    } else {
                        if (sp + n_hit - 1u > sc.wide_stack_cap) {
                            by_list = true; // the stack cannot take them (deeper than 31 entries): the list answers this ray
                            node = PT_TRAVERSAL_DONE;
                        } else {
                            if (n_hit > 3) stack[(sp++) * BLOCK] = (uint32_t)cr[3];
                            if (n_hit > 2) stack[(sp++) * BLOCK] = (uint32_t)cr[2];
                            if (n_hit > 1) stack[(sp++) * BLOCK] = (uint32_t)cr[1];
                            node = cr[0];
                        }
                    }
                } else if (inner) {
                    prof.tick(1);
                    float2 nx, ny, nz, fx, fy, fz;
                    int32_t c0, c1;
                    planes_of_record(w0, w1, w2, w3, br, nx, ny, nz, fx, fy, fz, c0, c1);
                    float tn0, tn1;
                    bool h0, h1;
                    box_test2(nx, ny, nz, fx, fy, fz, br, t_min, nh.best_t, h0, h1, tn0, tn1);
                    if (COUNT) nb += 2;
                    if (br.afx == 0.0f || br.afy == 0.0f || br.afz == 0.0f) { // a steep axis (rare): the bound its open far slot leaves out
                        const SteepBounds sb = make_steep_bounds(r, ray_kappa(sc.cone, r.o));
                        const float tf0 = fminf(fminf(__builtin_fmaf(fx.x, br.afx, br.cfx), __builtin_fmaf(fy.x, br.afy, br.cfy)), fminf(__builtin_fmaf(fz.x, br.afz, br.cfz), nh.best_t));
                        const float tf1 = fminf(fminf(__builtin_fmaf(fx.y, br.afx, br.cfx), __builtin_fmaf(fy.y, br.afy, br.cfy)), fminf(__builtin_fmaf(fz.y, br.afz, br.cfz), nh.best_t));
                        if (steep_rejects(sb, fx.x, fy.x, fz.x, tf0, br.eps)) h0 = false;
                        if (steep_rejects(sb, fx.y, fy.y, fz.y, tf1, br.eps)) h1 = false;
                    }
                    // the next child by selects, one masked store, one masked load (as in bvh_nearest; the stack holds the
                    // tree's depth + 2 entries, one per level of a descent and the bottom one)
                    const bool both = h0 && h1;
                    const bool pick1 = (both && tn1 < tn0) || (!h0 && h1);
                    node = pick1 ? c1 : c0;
                    if (both) stack[(sp++) * BLOCK] = (uint32_t)(pick1 ? c0 : c1);
                    if (!(h0 || h1)) node = (int32_t)stack[(--sp) * BLOCK];
                } else {
                    prof.tick(2);
                    const uint32_t first = ref >> 3, left = (ref & 7u) + 1u;
                    PrimRec q;
                    q.p[0] = w0.x; q.p[1] = w0.y; q.p[2] = w0.z; q.p[3] = w0.w; q.p[4] = w1.x; q.p[5] = w1.y;
                    q.meta = __float_as_uint(w1.z);
                    q.id = __float_as_uint(w1.w);
                    if (CONES) { // mixed scenes: one code path per family of objects (and no registers to spare for the split below: 3 % slower with it)
                        object_test_fused(q, first, r, t_min, t_max, nh);
                        if (left > 1) {
                            q.p[0] = w2.x; q.p[1] = w2.y; q.p[2] = w2.z; q.p[3] = w2.w; q.p[4] = w3.x; q.p[5] = w3.y;
                            q.meta = __float_as_uint(w3.z);
                            q.id = __float_as_uint(w3.w);
                            object_test_fused(q, first + 1u, r, t_min, t_max, nh);
                        }
                    } else {
                        PrimRec q2;
                        q2.p[0] = w2.x; q2.p[1] = w2.y; q2.p[2] = w2.z; q2.p[3] = w2.w; q2.p[4] = w3.x; q2.p[5] = w3.y;
                        q2.meta = __float_as_uint(w3.z);
                        q2.id = __float_as_uint(w3.w);
                        // Spheres first, in two halves (sphere.h:85-101): the discriminants of both, then the roots -- a
                        // correctly rounded sqrt and divides, the expensive part -- once for every lane that has a candidate,
                        // and once more only where a lane has two.  (The answer does not depend on the order in which the
                        // objects of a leaf are tried: equal t goes to the later list entry either way.)
                        const bool sph1 = SPH || (q.meta & 7u) == PT_SPHERE, sph2 = left > 1 && (SPH || (q2.meta & 7u) == PT_SPHERE);
                        float hb1 = 0.0f, disc1 = 0.0f, hb2 = 0.0f, disc2 = 0.0f;
                        const bool cand1 = sph1 && sphere_discriminant(q, r, hb1, disc1);
                        const bool cand2 = sph2 && sphere_discriminant(q2, r, hb2, disc2);
                        if (cand1 && cand2) {
                            float t = 0.0f;
                            const bool h = sphere_roots(r, hb1, disc1, t_min, nh.best_t, t);
                            nh.take(h, t, first, q.id);
                        }
                        if (cand1 || cand2) {
                            float t = 0.0f;
                            const bool h = sphere_roots(r, cand2 ? hb2 : hb1, cand2 ? disc2 : disc1, t_min, nh.best_t, t);
                            nh.take(h, t, cand2 ? first + 1u : first, cand2 ? q2.id : q.id);
                        }
                        if (!SPH) {
                            if (!sph1) object_test<CONES>(q, first, r, t_min, t_max, nh);
                            if (left > 1 && !sph2) object_test<CONES>(q2, first + 1u, r, t_min, t_max, nh);
                        }
                    }
                    if (COUNT) np += left > 1 ? 2u : 1u;
                    if (nh.best_t != nh.best_t) { // a hit at t = NaN was accepted: from here on the scan's answer is a matter of list order
                        by_list = true;
                        node = PT_TRAVERSAL_DONE;
                    } else if (left > 2) node = ~(int32_t)(((first + 2u) << 3) | (left - 3u)); // the rest of the leaf
                    else node = (int32_t)stack[(--sp) * BLOCK];
                }
                if (node != PT_TRAVERSAL_DONE) fetch();
            }
            }
        }
    }
    if (kProf && dbg) {
        uint32_t mx = dbg_maxsteps;
        for (int off = 32; off > 0; off >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, off, 64));
        if (lane == 0) {
            const uint32_t w = blockIdx.x * (BLOCK / 64) + threadIdx.x / 64;
            dbg[2 * w + 1] = wall_clock64();
            const size_t nw = dbg_waves;
            dbg[2 * nw + 2 * w] = ((unsigned long long)dbg_iter << 32) | dbg_serves;
            dbg[2 * nw + 2 * w + 1] = ((unsigned long long)mx << 32) | (uint32_t)rays;
        }
    }
    if (kProf) {
        for (int k = 0; k < 5; ++k) {
            const unsigned long long a = wave_sum_u32(prof.lanes[k]), b2 = wave_sum_u32(prof.slots[k]);
            if (lane == 0) { atomicAdd(&cnt->lanes[k], a); atomicAdd(&cnt->slots[k], b2); }
        }
        const unsigned long long it = wave_sum_u32(lane == 0 ? dbg_iter : 0u), sv = wave_sum_u32(lane == 0 ? dbg_serves : 0u);
        if (lane == 0) { atomicAdd(&cnt->sect[0], it); atomicAdd(&cnt->sect[1], sv); }
    }
    rays = wave_sum_u32((uint32_t)rays);
    if (lane == 0 && rays) atomicAdd(&cnt->rays, rays);
    if (COUNT) {
        unsigned long long sb = wave_sum_u32(nb), sp2 = wave_sum_u32(np);
        if (lane == 0) {
            atomicAdd(&cnt->box_tests, sb);
            atomicAdd(&cnt->prim_tests, sp2);
        }
    }
    if (nf) atomicAdd(&cnt->exact_fallbacks, (unsigned long long)nf);
    if (nr) atomicAdd(&cnt->exact_replays, (unsigned long long)nr);
}

// col += ray_color(...) in sample order (main.cu:93), and the pass's ray count.
__global__ __launch_bounds__(PT_BLOCK) void k_film(PassDev ps, const float *__restrict__ rad, float *__restrict__ rgb_sum,
                                                   const uint32_t *__restrict__ n_active, Counters *cnt, uint32_t pix0 = 0, uint32_t npix = 0xFFFFFFFFu)
{
    if (n_active && blockIdx.x == 0 && threadIdx.x == 0) {
        unsigned long long r = 0;
        for (int b = 0; b < ps.max_depth; ++b) r += n_active[b];
        atomicAdd(&cnt->rays, r);
    }
    // rad holds [samples][npix] for the pixels [pix0, pix0 + npix) (all of them by default)
    npix = min(npix, ps.n_pixels - pix0);
    for (uint32_t pl = blockIdx.x * PT_BLOCK + threadIdx.x; pl < npix; pl += gridDim.x * PT_BLOCK) {
        const size_t p = (size_t)pix0 + pl;
        float r = rgb_sum[3 * p], g = rgb_sum[3 * p + 1], b = rgb_sum[3 * p + 2];
        for (uint32_t s = 0; s < ps.samples; ++s) {
            size_t id = (size_t)s * npix + pl;
            r += rad[3 * id];
            g += rad[3 * id + 1];
            b += rad[3 * id + 2];
        }
        rgb_sum[3 * p] = r;
        rgb_sum[3 * p + 1] = g;
        rgb_sum[3 * p + 2] = b;
    }
}

// ---------------------------------------------------------------------------------------------
// Host side
// ---------------------------------------------------------------------------------------------
struct pt_context {
    int device = 0;
    hipStream_t own_stream = nullptr;
    std::string err;
    int n_cu = 256;

    bool has_scene = false;
    SceneDev sc{};
    BvhNode *d_nodes = nullptr;
    float4 *d_wide = nullptr;
    size_t n_wide_alloc = 0;
    uint32_t wide_depth = 0;
    PrimRec *d_prims = nullptr, *d_prims_list = nullptr;
    float4 *d_mats = nullptr;
    uint32_t max_depth_tree = 0;
    pt::BvhBuild tree;      // the installed scene's tree (host copy), refitted by pt_scene_update
    size_t n_mats_alloc = 0;
    bool has_cones = false; // the scene holds at least one cone: the only object whose answer can depend on list order
    int kind = PT_KIND_ANY; // which kernel variants serve the scene (PT_KIND_*)
    bool lds_scene = true; // traverse small scenes from LDS (PT_RENDER_NO_LDS_SCENE turns it off for A/B runs)

    // integrator workspace (grown on demand, kept across calls)
    size_t cap_paths = 0;
    float4 *d_A[2] = {nullptr, nullptr}, *d_B[2] = {nullptr, nullptr};
    float2 *d_C[2] = {nullptr, nullptr};
    uint2 *d_hits = nullptr;
    float *d_rad = nullptr;
    uint32_t *d_nactive = nullptr;
    size_t cap_nactive = 0;
    Counters *d_cnt = nullptr;
    // persistent-workgroup path kernel workspace
    size_t cap_slice_paths = 0; // grid * cap
    float4 *d_sA = nullptr, *d_sB = nullptr;
    float2 *d_sC = nullptr;
    float *d_srad = nullptr;
    uint32_t *d_ticket = nullptr;
    // k_paths_ring workspace: queues [grid][2][qcap], radiance [grid][PT_RING][cap * 3]
    size_t cap_ring_queue = 0, cap_ring_rad = 0; // entries, paths
    float4 *d_gA = nullptr, *d_gB = nullptr;
    float2 *d_gC = nullptr;
    float *d_grad = nullptr;
    uint2 *d_extents = nullptr; // slice table of the last frame shape
    size_t cap_extents = 0;
    uint32_t ext_key[4] = {0, 0, 0, 0}, n_extents = 0;
    // k_paths_stream: finished radiance of one pass, [samples][pixels][3]
    float *d_prad = nullptr;
    size_t cap_prad = 0; // paths
    // pt_trace_rays staging
    float *d_ro = nullptr, *d_rd = nullptr;
    pt_hit *d_rh = nullptr;
    size_t cap_rays = 0;
    std::vector<hipEvent_t> events;
    pt_tuning tune;              // pt_context_configure; negative = the library's own choice
    hipEvent_t last_done = nullptr; // end of the last call that enqueued work using the workspace above
    bool have_last = false;
    struct StreamPlan *stream_plan = nullptr; // launch shape of the streaming pipeline for the installed scene (pt_render_device_noalloc)
    void *scratch = nullptr;     // this GPU's address of the no-allocation scratch array (hipGetSymbolAddress, pt_context_create)
    // pt_render: the device film and a pinned staging buffer, kept across calls
    float *d_film = nullptr, *h_film = nullptr;
    size_t cap_film = 0, cap_hfilm = 0;
};

static int tuned(int32_t v, int dflt) { return v >= 0 ? (int)v : dflt; }

// The scratch of pt_render_device_noalloc: one array per GPU inside the code object.  Host code never uses the
// array's name as an address; a context holds what hipGetSymbolAddress returned with its device current.
#define PT_SCRATCH_HEADER 1024u // the pass counter at byte 0, the Counters record at byte 256
static __device__ __align__(256) unsigned char scratch[PT_SCRATCH_HEADER + PT_NOALLOC_SCRATCH_BYTES];

// Objects per BVH leaf.  Scenes small enough to be traversed from LDS do best with 4 (3 and 4 tie, 2 and 6
// lose 10-20 %).  Scenes traversed from HBM by k_paths_stream, which is bound by instruction issue: 1.  A leaf step
// costs a wave the object code of every kind of object its lanes hold, once per position in the leaf that any lane
// still has an object at; a node visit costs a fifth of that.  (1M spheres 3.04 / 3.20 / 3.27 Grays/s for 3 / 2 / 1,
// 100k mixed objects 2.73 / 2.88 / 3.25 / 3.46 for 4 / 3 / 2 / 1.)  A scene of at most eight objects is one leaf: the
// tree of the Cornell box (7 objects) costs more node visits than it saves object tests (leaves of 1 / 2 / 4 / 8:
// 61.4 / 65.2 / 59.7 / 52.8 ms at 256 spp).  pt_tuning.leaf_max overrides.
static uint32_t leaf_max_for(const pt_scene_desc *desc, const pt_tuning *tune = nullptr)
{
    if (tune && tune->leaf_max > 0) return (uint32_t)tune->leaf_max;
    if (desc->n_prims <= 8u) return 8u;
    return desc->n_prims > 1000u ? 1u : (uint32_t)PT_LEAF_MAX;
}

static pt_status fail(pt_context *ctx, pt_status st, const std::string &msg)
{
    if (ctx) ctx->err = msg;
    return st;
}

#define PT_HIP(ctx, call)                                                                                   \
    do {                                                                                                    \
        hipError_t e__ = (call);                                                                            \
        if (e__ != hipSuccess)                                                                              \
            return fail(ctx, PT_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(e__));          \
    } while (0)

extern "C" const char *pt_version(void) { return "pathtracer_amd 0.1 gfx950"; }

extern "C" pt_status pt_context_create(int device_id, pt_context **out)
{
    if (!out) return PT_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device_id < 0 || device_id >= n) return PT_ERR_DEVICE;
    pt_context *ctx = new pt_context();
    ctx->device = device_id;
    if (hipSetDevice(device_id) != hipSuccess) { delete ctx; return PT_ERR_DEVICE; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) ctx->n_cu = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return PT_ERR_DEVICE; }
    if (hipMalloc(&ctx->d_cnt, sizeof(Counters)) != hipSuccess) { hipStreamDestroy(ctx->own_stream); delete ctx; return PT_ERR_DEVICE; }
    if (hipEventCreateWithFlags(&ctx->last_done, hipEventDisableTiming) != hipSuccess) { hipFree(ctx->d_cnt); hipStreamDestroy(ctx->own_stream); delete ctx; return PT_ERR_DEVICE; }
    pt_tuning_default(&ctx->tune);
    // this GPU's copy of the no-allocation scratch array (each GPU holds its own, at its own address)
    if (hipGetSymbolAddress(&ctx->scratch, HIP_SYMBOL(scratch)) != hipSuccess || !ctx->scratch) {
        hipEventDestroy(ctx->last_done); hipFree(ctx->d_cnt); hipStreamDestroy(ctx->own_stream); delete ctx;
        return PT_ERR_DEVICE;
    }
    *out = ctx;
    return PT_OK;
}

extern "C" void pt_tuning_default(pt_tuning *t)
{
    if (!t) return;
    int32_t *f = reinterpret_cast<int32_t *>(t);
    for (size_t i = 0; i < sizeof(pt_tuning) / sizeof(int32_t); ++i) f[i] = -1;
}

static void drop_stream_plan(pt_context *ctx);
static void make_stream_plan(pt_context *ctx);
static pt_status order_behind_last(pt_context *ctx, hipStream_t stream);
static pt_status mark_last(pt_context *ctx, hipStream_t stream);

extern "C" pt_status pt_context_configure(pt_context *ctx, const pt_tuning *t)
{
    if (!ctx || !t) return PT_ERR_INVALID;
    auto in = [](int32_t v, int lo, int hi) { return v < 0 || (v >= lo && v <= hi); };
    const char *bad = nullptr;
    if (!in(t->leaf_max, 1, 8)) bad = "leaf_max (1..8)";
    else if (!in(t->wide_nodes, 0, 1)) bad = "wide_nodes (0, 1)";
    else if (!in(t->pipeline, 0, 2)) bad = "pipeline (0, 1, 2)";
    else if (!in(t->lds_scene, 0, 1)) bad = "lds_scene (0, 1)";
    else if (!(t->block < 0 || t->block == 256 || t->block == 512)) bad = "block (256, 512)";
    else if (!in(t->workgroups_per_cu, 1, 16)) bad = "workgroups_per_cu (1..16)";
    else if (!in(t->slice_paths, 64, 1 << 22)) bad = "slice_paths (64..2^22)";
    else if (!in(t->slice_taper, 1, 64)) bad = "slice_taper (1..64)";
    else if (!in(t->sort_by_material, 0, 1)) bad = "sort_by_material (0, 1)";
    else if (!in(t->stream_serve_min, 1, 64)) bad = "stream_serve_min (1..64)";
    else if (!in(t->stream_phase, 0, 16)) bad = "stream_phase (0..16)";
    else if (!(t->stream_tile < 0 || (t->stream_tile >= 64 && t->stream_tile <= 4096 && t->stream_tile % 64 == 0))) bad = "stream_tile (64..4096, a multiple of 64)";
    else if (!in(t->stream_run, 64, 65536)) bad = "stream_run (64..65536)";
    else if (!in(t->split_queue, 0, 1)) bad = "split_queue (0, 1)";
    else if (!in(t->ring_low_water, 0, 1 << 22)) bad = "ring_low_water (0..2^22)";
    if (bad) return fail(ctx, PT_ERR_INVALID, std::string("pt_tuning: out of range: ") + bad);
    ctx->tune = *t;
    if (hipSetDevice(ctx->device) == hipSuccess) make_stream_plan(ctx);
    return PT_OK;
}

static void free_scene(pt_context *ctx)
{
    hipFree(ctx->d_nodes); hipFree(ctx->d_prims); hipFree(ctx->d_prims_list); hipFree(ctx->d_mats); hipFree(ctx->d_wide);
    ctx->d_wide = nullptr; ctx->n_wide_alloc = 0;
    ctx->d_nodes = nullptr; ctx->d_prims = ctx->d_prims_list = nullptr; ctx->d_mats = nullptr;
    ctx->has_scene = false;
}

static void free_workspace(pt_context *ctx)
{
    hipFree(ctx->d_prad); ctx->d_prad = nullptr; ctx->cap_prad = 0;
    hipFree(ctx->d_extents); ctx->d_extents = nullptr; ctx->cap_extents = 0; ctx->n_extents = 0;
    for (int k = 0; k < 2; ++k) { hipFree(ctx->d_A[k]); hipFree(ctx->d_B[k]); hipFree(ctx->d_C[k]); ctx->d_A[k] = ctx->d_B[k] = nullptr; ctx->d_C[k] = nullptr; }
    hipFree(ctx->d_hits); hipFree(ctx->d_rad); ctx->d_hits = nullptr; ctx->d_rad = nullptr;
    ctx->cap_paths = 0;
}

extern "C" void pt_context_destroy(pt_context *ctx)
{
    if (!ctx) return;
    hipSetDevice(ctx->device);
    hipDeviceSynchronize();
    free_scene(ctx);
    free_workspace(ctx);
    hipFree(ctx->d_nactive); hipFree(ctx->d_cnt);
    hipFree(ctx->d_sA); hipFree(ctx->d_sB); hipFree(ctx->d_sC); hipFree(ctx->d_srad); hipFree(ctx->d_ticket);
    hipFree(ctx->d_gA); hipFree(ctx->d_gB); hipFree(ctx->d_gC); hipFree(ctx->d_grad);
    hipFree(ctx->d_ro); hipFree(ctx->d_rd); hipFree(ctx->d_rh);
    hipFree(ctx->d_film);
    if (ctx->h_film) hipHostFree(ctx->h_film);
    for (hipEvent_t e : ctx->events) hipEventDestroy(e);
    if (ctx->last_done) hipEventDestroy(ctx->last_done);
    drop_stream_plan(ctx);
    if (ctx->own_stream) hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

extern "C" const char *pt_last_error(const pt_context *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

// camera::camera, camera.h:38-62.  Host fp32, no contraction (this file is built with -ffp-contract=off).
extern "C" pt_status pt_camera_look_at(const float lookfrom[3], const float lookat[3], const float vup[3], float vfov,
                                       float aspect_ratio, float aperture, float focus_dist, pt_camera *out)
{
    if (!lookfrom || !lookat || !vup || !out) return PT_ERR_INVALID;
    struct H { float x, y, z; };
    auto sub = [](H a, H b) { return H{a.x - b.x, a.y - b.y, a.z - b.z}; };
    auto scale = [](float t, H a) { return H{t * a.x, t * a.y, t * a.z}; };
    auto cross = [](H a, H b) { return H{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; };
    auto unit = [&](H a) { float l = sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); return scale(1.0f / l, a); };
    const float PI = 3.1415926535897932385f; // cuda_functions.h:6
    H from{lookfrom[0], lookfrom[1], lookfrom[2]}, at{lookat[0], lookat[1], lookat[2]}, up{vup[0], vup[1], vup[2]};
    const float theta = vfov * PI / 180.0f;
    const float h = tanf(theta / 2.0f);
    const float viewport_height = 2.0f * h;
    const float viewport_width = aspect_ratio * viewport_height;
    H w = unit(sub(from, at));
    H u = unit(cross(up, w));
    H v = cross(w, u);
    H horizontal = scale(focus_dist * viewport_width, u);
    H vertical = scale(focus_dist * viewport_height, v);
    H llc = sub(sub(sub(from, scale(1.0f / 2.0f, horizontal)), scale(1.0f / 2.0f, vertical)), scale(focus_dist, w));
    auto st = [](float *p, H a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; };
    st(out->origin, from); st(out->lower_left_corner, llc); st(out->horizontal, horizontal); st(out->vertical, vertical);
    st(out->u, u); st(out->v, v); st(out->w, w);
    out->lens_radius = aperture / 2.0f;
    return PT_OK;
}

static std::string validate_desc(const pt_scene_desc *d)
{
    if (!d) return "null scene description";
    if (d->n_prims == 0 || !d->prim_type || !d->prim_param || !d->prim_material) return "scene has no objects";
    if (d->n_materials == 0 || !d->mat_type || !d->mat_param) return "scene has no materials";
    if (d->n_materials >= (1u << 27)) return "too many materials (limit 2^27 - 1)";
    for (uint32_t i = 0; i < d->n_materials; ++i)
        if (d->mat_type[i] >= PT_N_MAT_TYPES) return "material " + std::to_string(i) + " has an unknown tag";
    for (uint32_t i = 0; i < d->n_prims; ++i) {
        if (d->prim_type[i] >= PT_N_PRIM_TYPES) return "object " + std::to_string(i) + " has an unknown tag";
        // the reference indexes materials[material_index] unchecked (sphere.h:71)
        if (d->prim_material[i] >= d->n_materials) return "object " + std::to_string(i) + " names a material that does not exist";
    }
    return "";
}

static void make_records(const pt_scene_desc &d, const std::vector<uint32_t> &order, std::vector<PrimRec> &leaf,
                         std::vector<PrimRec> &list, std::vector<float4> &mats)
{
    list.resize(d.n_prims);
    for (uint32_t i = 0; i < d.n_prims; ++i) {
        PrimRec &q = list[i];
        for (int k = 0; k < 6; ++k) q.p[k] = d.prim_param[(size_t)i * 6 + k];
        if (d.prim_type[i] == PT_SPHERE) q.p[4] = 1.0f / q.p[3]; // the factor of sphere.h:105's (p - center) / radius (vec3.h:272-274), once per sphere instead of once per hit
        uint32_t m = d.prim_material[i];
        q.meta = (uint32_t)d.prim_type[i] | ((uint32_t)d.mat_type[m] << 3) | (m << 5);
        q.id = i;
    }
    leaf.resize(d.n_prims);
    for (uint32_t s = 0; s < d.n_prims; ++s) leaf[s] = list[order[s]];
    mats.resize(d.n_materials);
    for (uint32_t i = 0; i < d.n_materials; ++i) {
        const float *p = d.mat_param + (size_t)i * 4;
        float w = p[3];
        if (d.mat_type[i] == PT_METAL) w = w > 1.0f ? 1.0f : w; // material.h:150
        mats[i] = make_float4(p[0], p[1], p[2], w);
    }
}

extern "C" pt_status pt_bvh_build_host(const pt_scene_desc *desc, uint32_t *n_nodes, uint32_t *max_depth, void *nodes_out,
                                       uint32_t *order_out)
{
    if (!desc) return PT_ERR_INVALID;
    return pt_bvh_build_host_leaf(desc, leaf_max_for(desc), n_nodes, max_depth, nodes_out, order_out);
}

extern "C" pt_status pt_bvh_build_host_leaf(const pt_scene_desc *desc, uint32_t leaf_max, uint32_t *n_nodes, uint32_t *max_depth,
                                            void *nodes_out, uint32_t *order_out)
{
    std::string e = validate_desc(desc);
    if (!e.empty() || leaf_max < 1 || leaf_max > 8) return PT_ERR_INVALID;
    pt::BvhBuild b;
    e = pt::build_bvh(*desc, leaf_max, b);
    if (!e.empty()) return PT_ERR_INVALID;
    if (n_nodes) *n_nodes = (uint32_t)b.nodes.size();
    if (max_depth) *max_depth = b.max_depth;
    if (nodes_out) memcpy(nodes_out, b.nodes.data(), b.nodes.size() * sizeof(pt::HostNode));
    if (order_out) memcpy(order_out, b.order.data(), b.order.size() * sizeof(uint32_t));
    return PT_OK;
}

extern "C" pt_status pt_bvh_refit_host(const pt_scene_desc *before, const pt_scene_desc *after, uint32_t *n_nodes, uint32_t *max_depth,
                                       void *nodes_out, uint32_t *order_out)
{
    if (!validate_desc(before).empty() || !validate_desc(after).empty()) return PT_ERR_INVALID;
    pt::BvhBuild b;
    if (!pt::build_bvh(*before, leaf_max_for(before), b).empty()) return PT_ERR_INVALID;
    if (!pt::refit_bvh(*after, b).empty()) return PT_ERR_INVALID;
    if (n_nodes) *n_nodes = (uint32_t)b.nodes.size();
    if (max_depth) *max_depth = b.max_depth;
    if (nodes_out) memcpy(nodes_out, b.nodes.data(), b.nodes.size() * sizeof(pt::HostNode));
    if (order_out) memcpy(order_out, b.order.data(), b.order.size() * sizeof(uint32_t));
    return PT_OK;
}

extern "C" pt_status pt_bvh_wide_host(const pt_scene_desc *desc, uint32_t *n_nodes, uint32_t *max_depth, void *nodes_out)
{
    std::string e = validate_desc(desc);
    if (!e.empty()) return PT_ERR_INVALID;
    pt::BvhBuild b;
    e = pt::build_bvh(*desc, leaf_max_for(desc), b);
    if (!e.empty()) return PT_ERR_INVALID;
    pt::WideBuild w;
    pt::make_wide(b, w);
    if (n_nodes) *n_nodes = (uint32_t)w.nodes.size();
    if (max_depth) *max_depth = w.max_depth;
    if (nodes_out) memcpy(nodes_out, w.nodes.data(), w.nodes.size() * sizeof(pt::WideNode));
    return PT_OK;
}

extern "C" pt_status pt_scene_ray_cone(const pt_scene_desc *desc, pt_ray_cone *out)
{
    if (!out) return PT_ERR_INVALID;
    std::string e = validate_desc(desc);
    if (!e.empty()) return PT_ERR_INVALID;
    pt::BvhBuild b;
    e = pt::build_bvh(*desc, leaf_max_for(desc), b);
    if (!e.empty()) return PT_ERR_INVALID;
    out->kappa_max = b.cone.kappa_max;
    out->kappa_min = b.cone.kappa_min;
    out->r_min = b.cone.r_min;
    for (int a = 0; a < 3; ++a) out->cluster_centre[a] = b.cone.qc[a];
    out->cluster_half_extent_sum = b.cone.qh;
    out->adaptive = b.cone.adaptive;
    return PT_OK;
}

// Copies a built (or refitted) tree and the scene's records to the device and points the context at them.  Everything
// goes into new buffers first; the installed scene (if any) is replaced, and its buffers freed, only after every copy
// has succeeded, so a failure leaves the context rendering what it rendered before.
static pt_status install_scene(pt_context *ctx, const pt_scene_desc *desc, const pt::BvhBuild &b)
{
    if (b.max_depth > 96) return fail(ctx, PT_ERR_LIMIT, "BVH deeper than 96 levels");
    std::vector<PrimRec> leaf, list;
    std::vector<float4> mats;
    make_records(*desc, b.order, leaf, list, mats);

    PT_HIP(ctx, hipSetDevice(ctx->device));
    PT_HIP(ctx, hipDeviceSynchronize());
    const size_t nb = b.nodes.size() * sizeof(BvhNode), pb = leaf.size() * sizeof(PrimRec), mb = mats.size() * sizeof(float4);
    bool any_cone = false, only_spheres = true;
    for (uint32_t i = 0; i < desc->n_prims; ++i) {
        any_cone |= desc->prim_type[i] == PT_CONE;
        only_spheres &= desc->prim_type[i] == PT_SPHERE;
    }
    // Scenes too large for LDS are rendered from the four-wide form of the tree (k_paths_stream): 1.8 times fewer
    // node fetches per ray, 1M spheres 2.82 -> 3.47 Grays/s; with cones (whose kernel variant waits on those fetches
    // with four waves per SIMD) 100k mixed objects 4.54 -> 5.02.  pt_tuning.wide_nodes overrides.
    pt::WideBuild wb;
    const bool want_wide = nb + pb > 48 * 1024 && tuned(ctx->tune.wide_nodes, 1) != 0;
    if (want_wide) pt::make_wide(b, wb);

    BvhNode *n_nodes = nullptr;
    PrimRec *n_prims = nullptr, *n_list = nullptr;
    float4 *n_mats = nullptr, *n_wide = nullptr;
    auto upload = [&]() -> hipError_t {
        hipError_t e;
        if ((e = hipMalloc(&n_nodes, nb)) != hipSuccess) return e;
        if ((e = hipMalloc(&n_prims, pb + sizeof(PrimRec))) != hipSuccess) return e; // one record of padding: k_paths_stream fetches objects in pairs
        if ((e = hipMalloc(&n_list, pb)) != hipSuccess) return e;
        if ((e = hipMalloc(&n_mats, mb)) != hipSuccess) return e;
        if (want_wide && (e = hipMalloc(&n_wide, wb.nodes.size() * sizeof(pt::WideNode))) != hipSuccess) return e;
        if ((e = hipMemset(reinterpret_cast<char *>(n_prims) + pb, 0, sizeof(PrimRec))) != hipSuccess) return e;
        if ((e = hipMemcpy(n_nodes, b.nodes.data(), nb, hipMemcpyHostToDevice)) != hipSuccess) return e;
        if ((e = hipMemcpy(n_prims, leaf.data(), pb, hipMemcpyHostToDevice)) != hipSuccess) return e;
        if ((e = hipMemcpy(n_list, list.data(), pb, hipMemcpyHostToDevice)) != hipSuccess) return e;
        if ((e = hipMemcpy(n_mats, mats.data(), mb, hipMemcpyHostToDevice)) != hipSuccess) return e;
        if (want_wide && (e = hipMemcpy(n_wide, wb.nodes.data(), wb.nodes.size() * sizeof(pt::WideNode), hipMemcpyHostToDevice)) != hipSuccess) return e;
        return hipSuccess;
    };
    const hipError_t ue = upload();
    if (ue != hipSuccess) {
        hipFree(n_nodes); hipFree(n_prims); hipFree(n_list); hipFree(n_mats); hipFree(n_wide);
        return fail(ctx, PT_ERR_DEVICE, std::string("scene upload: ") + hipGetErrorString(ue));
    }
    free_scene(ctx); // the previous scene's buffers
    ctx->d_nodes = n_nodes; ctx->d_prims = n_prims; ctx->d_prims_list = n_list; ctx->d_mats = n_mats; ctx->d_wide = n_wide;
    ctx->n_mats_alloc = mats.size();
    ctx->n_wide_alloc = want_wide ? wb.nodes.size() : 0;

    SceneDev &sc = ctx->sc;
    sc.wide = n_wide;
    // up to three entries per level; 31 is what five workgroups of 256 lanes leave each lane of the CU's LDS
    // (a ray that needs more is answered by the list scan and counted in exact_fallbacks)
    sc.wide_stack_cap = want_wide ? std::min(31u, 3u * wb.max_depth + 2u) : 0u; // (with the bottom entry; 31 x 256 lanes x 4 B lets five workgroups share a CU's LDS)
    ctx->wide_depth = want_wide ? wb.max_depth : 0u;
    sc.nodes = ctx->d_nodes;
    sc.prims = ctx->d_prims;
    sc.prims_list = ctx->d_prims_list;
    sc.mats = ctx->d_mats;
    sc.n_prims = desc->n_prims;
    sc.n_nodes = (uint32_t)b.nodes.size();
    sc.stack_cap = std::max(8u, b.max_depth + 2); // one entry per level of a descent, and the "done" entry at the bottom
    // a scene that is one leaf (at most eight objects): k_paths' FLAT kernels start at the leaf, without the visit to a root whose one box every ray is in
    sc.root_ref = b.nodes.size() == 1 && b.nodes[0].c[0] < 0 && !(b.nodes[0].b[0][1] < INFINITY) ? b.nodes[0].c[0] : 0;
    sc.cone = b.cone;
    const pt_camera &c = desc->camera;
    for (int k = 0; k < 3; ++k) {
        sc.cam.origin[k] = c.origin[k]; sc.cam.llc[k] = c.lower_left_corner[k];
        sc.cam.horizontal[k] = c.horizontal[k]; sc.cam.vertical[k] = c.vertical[k];
        sc.cam.u[k] = c.u[k]; sc.cam.v[k] = c.v[k];
        sc.bg[k] = desc->background[k];
    }
    sc.cam.lens_radius = c.lens_radius;
    ctx->max_depth_tree = b.max_depth;
    ctx->has_cones = any_cone;
    ctx->kind = any_cone ? PT_KIND_ANY : only_spheres ? PT_KIND_SPHERES : PT_KIND_NO_CONES;
    ctx->has_scene = true;
    ctx->have_last = false; // (the device was synchronised above)
    make_stream_plan(ctx);
    return PT_OK;
}

extern "C" pt_status pt_scene_upload(pt_context *ctx, const pt_scene_desc *desc)
{
    if (!ctx) return PT_ERR_INVALID;
    std::string e = validate_desc(desc);
    if (!e.empty()) return fail(ctx, PT_ERR_INVALID, e);
    pt::BvhBuild b;
    e = pt::build_bvh(*desc, leaf_max_for(desc, &ctx->tune), b);
    if (!e.empty()) return fail(ctx, PT_ERR_INVALID, e);
    const pt_status st = install_scene(ctx, desc, b);
    if (st != PT_OK) return st;
    ctx->tree = std::move(b); // kept for pt_scene_update
    return PT_OK;
}

extern "C" pt_status pt_scene_update(pt_context *ctx, const pt_scene_desc *desc)
{
    if (!ctx) return PT_ERR_INVALID;
    if (!ctx->has_scene) return fail(ctx, PT_ERR_NO_SCENE, "no scene uploaded");
    std::string e = validate_desc(desc);
    if (!e.empty()) return fail(ctx, PT_ERR_INVALID, e);
    if (desc->n_prims != ctx->sc.n_prims) return fail(ctx, PT_ERR_INVALID, "pt_scene_update needs the same number of objects as the uploaded scene");
    pt::BvhBuild b = ctx->tree; // refit a copy: a failure leaves the installed scene as it was
    e = pt::refit_bvh(*desc, b);
    if (!e.empty()) return fail(ctx, PT_ERR_INVALID, e);
    const pt_status st = install_scene(ctx, desc, b);
    if (st != PT_OK) return st;
    ctx->tree = std::move(b);
    return PT_OK;
}

static uint32_t grid_for(const pt_context *ctx, size_t n)
{
    size_t blocks = (n + PT_BLOCK - 1) / PT_BLOCK;
    size_t cap = (size_t)ctx->n_cu * 8;
    return (uint32_t)std::max<size_t>(1, std::min(blocks, cap));
}

// LDS bytes of the resident-scene variant; 0 when the scene does not fit beside the traversal stack.
static size_t lds_scene_bytes(const pt_context *ctx)
{
    const SceneDev &sc = ctx->sc;
    size_t b = lds_scene_size(sc.n_nodes, sc.n_prims) + (size_t)sc.stack_cap * PT_ISECT_LDS_BLOCK * 4;
    return b <= 80 * 1024 ? b : 0; // two workgroups per CU must fit in 160 KB
}

static void launch_intersect(pt_context *ctx, const SceneDev &scene, hipStream_t stream, const PathPlanes &cur, const uint32_t *n_ptr, size_t max_paths, bool count)
{
    const size_t lds_res = ctx->lds_scene ? lds_scene_bytes(ctx) : 0;
    if (lds_res) {
        const uint32_t grid = (uint32_t)std::max<size_t>(1, std::min((max_paths + PT_ISECT_LDS_BLOCK - 1) / PT_ISECT_LDS_BLOCK, (size_t)ctx->n_cu * 2));
        if (count)
            hipLaunchKernelGGL((k_intersect<true, true, PT_ISECT_LDS_BLOCK, PT_KIND_ANY>), dim3(grid), dim3(PT_ISECT_LDS_BLOCK), lds_res, stream, scene, cur, ctx->d_hits, n_ptr, ctx->d_cnt);
        else
            hipLaunchKernelGGL((k_intersect<false, true, PT_ISECT_LDS_BLOCK, PT_KIND_ANY>), dim3(grid), dim3(PT_ISECT_LDS_BLOCK), lds_res, stream, scene, cur, ctx->d_hits, n_ptr, ctx->d_cnt);
    } else {
        const uint32_t grid = grid_for(ctx, max_paths);
        const size_t lds = (size_t)ctx->sc.stack_cap * PT_BLOCK * sizeof(uint32_t);
        if (count)
            hipLaunchKernelGGL((k_intersect<true, false, PT_BLOCK, PT_KIND_ANY>), dim3(grid), dim3(PT_BLOCK), lds, stream, scene, cur, ctx->d_hits, n_ptr, ctx->d_cnt);
        else
            hipLaunchKernelGGL((k_intersect<false, false, PT_BLOCK, PT_KIND_ANY>), dim3(grid), dim3(PT_BLOCK), lds, stream, scene, cur, ctx->d_hits, n_ptr, ctx->d_cnt);
    }
}

extern "C" pt_status pt_trace_rays_device(pt_context *ctx, const float *d_o, const float *d_d, size_t n, float t_min,
                                          float t_max, int mode, pt_hit *d_out, void *stream_)
{
    if (!ctx) return PT_ERR_INVALID;
    if (!ctx->has_scene) return fail(ctx, PT_ERR_NO_SCENE, "no scene uploaded");
    if (n == 0) return PT_OK;
    if (!d_o || !d_d || !d_out) return fail(ctx, PT_ERR_INVALID, "null ray or output buffer");
    if (mode != PT_TRACE_BVH && mode != PT_TRACE_LINEAR && mode != PT_TRACE_BVH_IN_SCENE && mode != PT_TRACE_BVH_GEOMETRIC) return fail(ctx, PT_ERR_INVALID, "unknown trace mode");
    hipStream_t stream = (hipStream_t)stream_;
    PT_HIP(ctx, hipSetDevice(ctx->device));
    { const pt_status os = order_behind_last(ctx, stream); if (os != PT_OK) return os; }
    PT_HIP(ctx, hipMemsetAsync(ctx->d_cnt, 0, sizeof(Counters), stream));
    uint32_t grid = grid_for(ctx, n);
    size_t lds = (size_t)ctx->sc.stack_cap * PT_BLOCK * sizeof(uint32_t);
    if (mode == PT_TRACE_LINEAR)
        hipLaunchKernelGGL(k_trace<PT_TRACE_LINEAR>, dim3(grid), dim3(PT_BLOCK), lds, stream, ctx->sc, d_o, d_d, n, t_min, t_max, d_out, ctx->d_cnt);
    else if (mode == PT_TRACE_BVH_IN_SCENE)
        hipLaunchKernelGGL(k_trace<PT_TRACE_BVH_IN_SCENE>, dim3(grid), dim3(PT_BLOCK), lds, stream, ctx->sc, d_o, d_d, n, t_min, t_max, d_out, ctx->d_cnt);
    else if (mode == PT_TRACE_BVH_GEOMETRIC) {
        SceneDev scene = ctx->sc; // no ray cone: boxes culled by exact geometry
        scene.cone.kappa_max = scene.cone.kappa_min = 0.0f;
        scene.cone.adaptive = 0;
        hipLaunchKernelGGL(k_trace<PT_TRACE_BVH_IN_SCENE>, dim3(grid), dim3(PT_BLOCK), lds, stream, scene, d_o, d_d, n, t_min, t_max, d_out, ctx->d_cnt);
    }
    else
        hipLaunchKernelGGL(k_trace<PT_TRACE_BVH>, dim3(grid), dim3(PT_BLOCK), lds, stream, ctx->sc, d_o, d_d, n, t_min, t_max, d_out, ctx->d_cnt);
    PT_HIP(ctx, hipGetLastError());
    return mark_last(ctx, stream);
}

extern "C" pt_status pt_trace_rays(pt_context *ctx, const float *o, const float *d, size_t n, float t_min, float t_max,
                                   int mode, pt_hit *out)
{
    if (!ctx) return PT_ERR_INVALID;
    if (!ctx->has_scene) return fail(ctx, PT_ERR_NO_SCENE, "no scene uploaded");
    if (n == 0) return PT_OK;
    if (!o || !d || !out) return fail(ctx, PT_ERR_INVALID, "null ray or output buffer");
    PT_HIP(ctx, hipSetDevice(ctx->device));
    if (n > ctx->cap_rays) {
        hipFree(ctx->d_ro); hipFree(ctx->d_rd); hipFree(ctx->d_rh);
        ctx->d_ro = ctx->d_rd = nullptr; ctx->d_rh = nullptr; ctx->cap_rays = 0;
        PT_HIP(ctx, hipMalloc(&ctx->d_ro, n * 3 * sizeof(float)));
        PT_HIP(ctx, hipMalloc(&ctx->d_rd, n * 3 * sizeof(float)));
        PT_HIP(ctx, hipMalloc(&ctx->d_rh, n * sizeof(pt_hit)));
        ctx->cap_rays = n;
    }
    PT_HIP(ctx, hipMemcpyAsync(ctx->d_ro, o, n * 3 * sizeof(float), hipMemcpyHostToDevice, ctx->own_stream));
    PT_HIP(ctx, hipMemcpyAsync(ctx->d_rd, d, n * 3 * sizeof(float), hipMemcpyHostToDevice, ctx->own_stream));
    pt_status st = pt_trace_rays_device(ctx, ctx->d_ro, ctx->d_rd, n, t_min, t_max, mode, ctx->d_rh, ctx->own_stream);
    if (st != PT_OK) return st;
    PT_HIP(ctx, hipMemcpyAsync(out, ctx->d_rh, n * sizeof(pt_hit), hipMemcpyDeviceToHost, ctx->own_stream));
    PT_HIP(ctx, hipStreamSynchronize(ctx->own_stream));
    return PT_OK;
}

extern "C" pt_status pt_scatter_batch(pt_context *ctx, const float *o, const float *d, const float *t, const uint32_t *prim, const float *sphere_pt,
                                      const float *xi, size_t n, pt_scatter *out)
{
    static_assert(sizeof(pt_scatter) == sizeof(ScatterOut), "pt_scatter layout");
    if (!ctx) return PT_ERR_INVALID;
    if (!ctx->has_scene) return fail(ctx, PT_ERR_NO_SCENE, "no scene uploaded");
    if (n == 0) return PT_OK;
    if (!o || !d || !t || !prim || !sphere_pt || !xi || !out) return fail(ctx, PT_ERR_INVALID, "null input or output buffer");
    for (size_t i = 0; i < n; ++i)
        if (prim[i] >= ctx->sc.n_prims) return fail(ctx, PT_ERR_INVALID, "hit " + std::to_string(i) + " names an object that does not exist");
    PT_HIP(ctx, hipSetDevice(ctx->device));
    // one staging buffer for the call: 3 + 3 + 1 + 1 + 3 + 1 input words and the output record per hit
    const size_t in_words = 12, bytes = n * (in_words * 4 + sizeof(ScatterOut));
    char *buf = nullptr;
    PT_HIP(ctx, hipMalloc(&buf, bytes));
    float *d_o = reinterpret_cast<float *>(buf), *d_d = d_o + 3 * n, *d_t = d_d + 3 * n;
    uint32_t *d_prim = reinterpret_cast<uint32_t *>(d_t + n);
    float *d_sp = reinterpret_cast<float *>(d_prim + n), *d_xi = d_sp + 3 * n;
    ScatterOut *d_out = reinterpret_cast<ScatterOut *>(d_xi + n);
    hipStream_t st = ctx->own_stream;
    hipError_t e = hipSuccess;
    auto up = [&](void *dst, const void *src, size_t b) { if (e == hipSuccess) e = hipMemcpyAsync(dst, src, b, hipMemcpyHostToDevice, st); };
    up(d_o, o, n * 12); up(d_d, d, n * 12); up(d_t, t, n * 4); up(d_prim, prim, n * 4); up(d_sp, sphere_pt, n * 12); up(d_xi, xi, n * 4);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_scatter, dim3(grid_for(ctx, n)), dim3(PT_BLOCK), 0, st, ctx->sc, (const float *)d_o, (const float *)d_d, (const float *)d_t,
                           (const uint32_t *)d_prim, (const float *)d_sp, (const float *)d_xi, n, d_out);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, n * sizeof(ScatterOut), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    hipFree(buf);
    if (e != hipSuccess) return fail(ctx, PT_ERR_DEVICE, std::string("pt_scatter_batch: ") + hipGetErrorString(e));
    return PT_OK;
}

static pt_status ensure_workspace(pt_context *ctx, size_t paths, size_t depth_slots)
{
    if (paths > ctx->cap_paths) {
        free_workspace(ctx);
        for (int k = 0; k < 2; ++k) {
            PT_HIP(ctx, hipMalloc(&ctx->d_A[k], paths * sizeof(float4)));
            PT_HIP(ctx, hipMalloc(&ctx->d_B[k], paths * sizeof(float4)));
            PT_HIP(ctx, hipMalloc(&ctx->d_C[k], paths * sizeof(float2)));
        }
        PT_HIP(ctx, hipMalloc(&ctx->d_hits, paths * sizeof(uint2)));
        PT_HIP(ctx, hipMalloc(&ctx->d_rad, paths * 3 * sizeof(float)));
        ctx->cap_paths = paths;
    }
    if (depth_slots > ctx->cap_nactive) {
        hipFree(ctx->d_nactive);
        ctx->d_nactive = nullptr; ctx->cap_nactive = 0;
        PT_HIP(ctx, hipMalloc(&ctx->d_nactive, depth_slots * sizeof(uint32_t)));
        ctx->cap_nactive = depth_slots;
    }
    return PT_OK;
}

struct StageTimer {
    pt_context *ctx;
    hipStream_t stream;
    bool on;
    size_t used = 0;
    std::vector<int> stage; // per event pair
    StageTimer(pt_context *c, hipStream_t s, bool o) : ctx(c), stream(s), on(o) {}
    hipEvent_t get()
    {
        if (used == ctx->events.size()) {
            hipEvent_t e;
            hipEventCreate(&e);
            ctx->events.push_back(e);
        }
        return ctx->events[used++];
    }
    void begin(int st) { if (on) { stage.push_back(st); hipEventRecord(get(), stream); } }
    void end() { if (on) hipEventRecord(get(), stream); }
};


template <bool COUNT>
static void launch_paths(pt_context *ctx, const SceneDev &scene, hipStream_t stream, int block, bool lds, bool sort, bool split, uint32_t grid, size_t lds_bytes,
                         const PassDev &ps, const SliceDev &sl, float *d_rgb_sum)
{
    const int kind = ctx->kind;
#define PT_LAUNCH_PATHS(L, B, S, K) hipLaunchKernelGGL((k_paths<COUNT, L, B, S, K>), dim3(grid), dim3(B), lds_bytes, stream, scene, ps, sl, d_rgb_sum, ctx->d_cnt)
#define PT_LAUNCH_PATHS_K(L, B, S)                                               \
    do {                                                                         \
        if (kind == PT_KIND_ANY) PT_LAUNCH_PATHS(L, B, S, PT_KIND_ANY);          \
        else if (kind == PT_KIND_NO_CONES) PT_LAUNCH_PATHS(L, B, S, PT_KIND_NO_CONES); \
        else PT_LAUNCH_PATHS(L, B, S, PT_KIND_SPHERES);                          \
    } while (0)
#define PT_LAUNCH_PATHS_SC(L, B)                                                 \
    do {                                                                         \
        if (sort) PT_LAUNCH_PATHS_K(L, B, true);                                 \
        else PT_LAUNCH_PATHS_K(L, B, false);                                     \
    } while (0)
    if (!COUNT && lds && !sort && scene.root_ref < 0 && kind != PT_KIND_SPHERES) { // a scene of one leaf: the kernels without tree code
        if (kind == PT_KIND_ANY) {
            if (block == 256) hipLaunchKernelGGL((k_paths<false, true, 256, false, PT_KIND_ANY, false, true>), dim3(grid), dim3(256), lds_bytes, stream, scene, ps, sl, d_rgb_sum, ctx->d_cnt);
            else hipLaunchKernelGGL((k_paths<false, true, 512, false, PT_KIND_ANY, false, true>), dim3(grid), dim3(512), lds_bytes, stream, scene, ps, sl, d_rgb_sum, ctx->d_cnt);
        } else {
            if (block == 256) hipLaunchKernelGGL((k_paths<false, true, 256, false, PT_KIND_NO_CONES, false, true>), dim3(grid), dim3(256), lds_bytes, stream, scene, ps, sl, d_rgb_sum, ctx->d_cnt);
            else hipLaunchKernelGGL((k_paths<false, true, 512, false, PT_KIND_NO_CONES, false, true>), dim3(grid), dim3(512), lds_bytes, stream, scene, ps, sl, d_rgb_sum, ctx->d_cnt);
        }
    } else if (lds && split && !sort && kind == PT_KIND_SPHERES) { // (the one shape the two-ended queue is built for)
        if (block == 256) hipLaunchKernelGGL((k_paths<COUNT, true, 256, false, PT_KIND_SPHERES, true>), dim3(grid), dim3(256), lds_bytes, stream, scene, ps, sl, d_rgb_sum, ctx->d_cnt);
        else hipLaunchKernelGGL((k_paths<COUNT, true, 512, false, PT_KIND_SPHERES, true>), dim3(grid), dim3(512), lds_bytes, stream, scene, ps, sl, d_rgb_sum, ctx->d_cnt);
    } else if (lds) {
        if (block == 256) PT_LAUNCH_PATHS_SC(true, 256);
        else PT_LAUNCH_PATHS_SC(true, 512);
    } else {
        if (block == 256) PT_LAUNCH_PATHS_SC(false, 256);
        else PT_LAUNCH_PATHS_SC(false, 512);
    }
#undef PT_LAUNCH_PATHS_SC
#undef PT_LAUNCH_PATHS_K
#undef PT_LAUNCH_PATHS
}

template <bool COUNT>
static void launch_ring(pt_context *ctx, const SceneDev &scene, hipStream_t stream, int block, bool split, uint32_t grid, size_t lds_bytes, const PassDev &ps,
                        const RingDev &rg, float *d_rgb_sum)
{
    const int kind = ctx->kind;
#define PT_LAUNCH_RING(B, K, SP, FL) hipLaunchKernelGGL((k_paths_ring<COUNT, true, B, K, SP, FL>), dim3(grid), dim3(B), lds_bytes, stream, scene, ps, rg, d_rgb_sum, ctx->d_cnt)
#define PT_LAUNCH_RING_B(K, SP, FL)              \
    do {                                         \
        if (block == 256) PT_LAUNCH_RING(256, K, SP, FL); \
        else PT_LAUNCH_RING(512, K, SP, FL);     \
    } while (0)
    if (!COUNT && scene.root_ref < 0 && kind != PT_KIND_SPHERES) { // a scene of one leaf: the kernels without tree code
        if (kind == PT_KIND_ANY) PT_LAUNCH_RING_B(PT_KIND_ANY, false, true);
        else PT_LAUNCH_RING_B(PT_KIND_NO_CONES, false, true);
    } else if (kind == PT_KIND_SPHERES) {
        if (split) PT_LAUNCH_RING_B(PT_KIND_SPHERES, true, false);
        else PT_LAUNCH_RING_B(PT_KIND_SPHERES, false, false);
    } else if (kind == PT_KIND_NO_CONES) {
        PT_LAUNCH_RING_B(PT_KIND_NO_CONES, false, false);
    } else {
        PT_LAUNCH_RING_B(PT_KIND_ANY, false, false);
    }
#undef PT_LAUNCH_RING_B
#undef PT_LAUNCH_RING
}

// The k_paths_stream instance for a launch shape.
static const void *stream_kernel_for(bool count, int block, int kind, bool wide)
{
#define PT_STREAM_PTR(C, B, K, W) reinterpret_cast<const void *>(&k_paths_stream<C, B, K, W>)
#define PT_STREAM_W(C, B, K) (wide ? PT_STREAM_PTR(C, B, K, true) : PT_STREAM_PTR(C, B, K, false))
#define PT_STREAM_K(C, B) (kind == PT_KIND_ANY ? PT_STREAM_W(C, B, PT_KIND_ANY) : kind == PT_KIND_NO_CONES ? PT_STREAM_W(C, B, PT_KIND_NO_CONES) : PT_STREAM_W(C, B, PT_KIND_SPHERES))
#define PT_STREAM_B(C) (block == 256 ? PT_STREAM_K(C, 256) : PT_STREAM_K(C, 512))
    return count ? PT_STREAM_B(true) : PT_STREAM_B(false);
#undef PT_STREAM_B
#undef PT_STREAM_K
#undef PT_STREAM_W
#undef PT_STREAM_PTR
}

// Launch shape of the path kernels for the installed scene: which pipeline, workgroup size, LDS use, resident grid.
#ifndef PT_DEFAULT_LDS_PIPELINE
#define PT_DEFAULT_LDS_PIPELINE 2 // which kernel renders a scene that fits LDS unless pt_tuning.pipeline says otherwise: 0 k_paths, 2 k_paths_ring
#endif
struct PathsShape {
    int block;
    bool lds, sort, stream, wide;
    bool ring;                 // k_paths_ring (slices overlapped within a workgroup) instead of k_paths
    size_t lds_bytes;          // k_paths: traversal stacks (+ the scene copy)
    size_t stream_stack_bytes; // k_paths_stream: traversal stacks
    uint32_t grid_max;         // resident workgroups
};
struct StreamPlan { PathsShape shape; };

static PathsShape paths_shape(pt_context *ctx, uint32_t flags, bool count, bool force_stream, uint64_t total_paths = 0)
{
    const pt_tuning &tu = ctx->tune;
    const SceneDev &sc = ctx->sc;
    PathsShape ps;
    const bool want_lds = (flags & PT_RENDER_NO_LDS_SCENE) == 0 && tuned(tu.lds_scene, 1) != 0;
    ps.sort = !force_stream && (flags & PT_RENDER_NO_SORT) == 0 && tuned(tu.sort_by_material, 0) != 0; // measured slower on the spheres scene: opt-in
    const size_t scene_bytes = lds_scene_size(sc.n_nodes, sc.n_prims);
    int block = tuned(tu.block, 0);
    if (block != 256 && block != 512) {
        // LDS-resident scenes: 512 (one scene copy per 8 waves).  Scenes traversed from HBM: 256 measured best without cones;
        // with cones 512, the shape at which the compiler gets k_paths_stream's cone variant into 128 registers (4 waves per
        // SIMD instead of the 3 its 132 registers at 256 allow: 100k mixed objects 2.40 -> 2.85 Grays/s)
        const bool fits_lds = want_lds && !force_stream && scene_bytes + (size_t)sc.stack_cap * 512 * 4 <= 64 * 1024;
        block = fits_lds || ctx->has_cones ? 512 : 256;
        // A scene so small that six workgroups of 256 share a CU's LDS: the smaller workgroups balance a large frame
        // better (Cornell box at 1024 x 1024 x 1024 spp: 250.7 -> 241.6 ms; a frame of 1.4 M paths loses 5 %).
        if (fits_lds && 6 * (scene_bytes + (size_t)sc.stack_cap * 256 * 4 + 64) <= 160 * 1024 && total_paths >= (16ull << 20)) block = 256;
    }
    ps.block = block;
    const size_t stack_bytes = (size_t)sc.stack_cap * block * 4;
    ps.lds = want_lds && !force_stream && scene_bytes + stack_bytes <= 64 * 1024;
    ps.lds_bytes = stack_bytes + (ps.lds ? scene_bytes : 0);
    // scenes traversed from HBM take the streaming kernel (pt_tuning.pipeline overrides)
    const int pipeline = tuned(tu.pipeline, ps.lds ? PT_DEFAULT_LDS_PIPELINE : 1);
    ps.ring = !force_stream && !ps.sort && ps.lds && pipeline == 2; // (scenes that do not fit LDS: the streaming kernel instead)
    ps.stream = force_stream || (!ps.sort && pipeline != 0 && !ps.ring);
    ps.wide = ps.stream && sc.wide != nullptr;
    // the wide traversal's stack; the exact replay of cone scenes walks the binary tree on the same stack
    ps.stream_stack_bytes = ps.wide ? (size_t)std::max(sc.wide_stack_cap, ctx->has_cones ? sc.stack_cap : 0u) * block * 4 : stack_bytes;
    int per_cu = tuned(tu.workgroups_per_cu, 0);
    if (per_cu <= 0) {
        const size_t static_lds = 64 + (ps.sort ? (size_t)block * 10 + 64 : 0) + (ps.ring ? 12 * PT_RING + 32 : 0); // s_hit, s_perm, s_bin, counters; the ring's slots
        const size_t lds_used = (ps.stream ? ps.stream_stack_bytes : ps.lds_bytes) + static_lds;
        const int waves_per_simd = ps.stream ? PT_STREAM_WAVES : 6; // the kernels' register budgets
        per_cu = std::min<int>(waves_per_simd * 4 * 64 / block, (int)((160 * 1024) / std::max<size_t>(lds_used, 1)));
        if (ps.stream) { // what the runtime will really keep resident (its cone variants take more registers than the budget)
            int fit = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&fit, stream_kernel_for(count, block, ctx->kind, ps.wide), block, ps.stream_stack_bytes) == hipSuccess && fit > 0)
                per_cu = std::min(per_cu, fit);
        }
        per_cu = std::max(per_cu, 1);
    }
    ps.grid_max = (uint32_t)ctx->n_cu * (uint32_t)per_cu;
    return ps;
}

static void drop_stream_plan(pt_context *ctx)
{
    delete ctx->stream_plan;
    ctx->stream_plan = nullptr;
}
// The streaming pipeline's shape for the installed scene, worked out where allocation and queries are allowed
// (pt_scene_upload, pt_scene_update, pt_context_configure) so that pt_render_device_noalloc only launches.
static void make_stream_plan(pt_context *ctx)
{
    drop_stream_plan(ctx);
    if (!ctx->has_scene) return;
    ctx->stream_plan = new StreamPlan{paths_shape(ctx, 0, false, true)};
}

// One frame through k_paths_stream + k_film.  One launch per pass: `spass` samples of a band of pixels; the pass's
// finished radiance goes to rad ([samples][band pixels][3], room for rad_cap_paths paths) and k_film adds it to the
// film in sample order.  A frame that fits the buffer is one pass; otherwise passes of up to 256 samples go band by
// band, so that the samples of a pixel are traced close together in time.  Enqueues only (memsets and launches).
static pt_status stream_passes(pt_context *ctx, const pt_render_params *pr, float *d_rgb_sum, hipStream_t stream, const PathsShape &sh, bool count,
                               SceneDev scene, float *rad, uint64_t rad_cap_paths, uint32_t *ticket, Counters *cnt, struct StageTimer *tm,
                               uint32_t *n_passes_out, unsigned long long *dbg, size_t dbg_waves);

static pt_status render_slices(pt_context *ctx, const pt_render_params *pr, float *d_rgb_sum, hipStream_t stream, pt_stats *stats)
{
    const uint32_t n_pixels = (uint32_t)pr->width * (uint32_t)pr->height;
    const bool count = (pr->flags & PT_RENDER_COUNT_TESTS) != 0;
    const PathsShape sh = paths_shape(ctx, pr->flags, count, false, (uint64_t)pr->width * (uint64_t)pr->height * pr->spp);
    const int block = sh.block;
    const bool lds = sh.lds, sort = sh.sort;
    const size_t lds_bytes = sh.lds_bytes;
    const uint32_t grid_max = sh.grid_max;
    SceneDev scene = ctx->sc;
    if (pr->flags & PT_RENDER_GEOMETRIC_CULL) { // boxes culled by exact geometry: no ray cone
        scene.cone.kappa_max = scene.cone.kappa_min = 0.0f;
        scene.cone.adaptive = 0;
    }

    if (sh.stream) {
        // the pass buffer: 1 GiB at most unless the caller fixes samples_per_pass
        const uint64_t budget_paths = (1ull << 30) / 12;
        const uint32_t tile = (uint32_t)tuned(ctx->tune.stream_tile, 64);
        uint32_t spass = pr->samples_per_pass ? pr->samples_per_pass : std::min(pr->spp, 256u);
        spass = std::min(spass, pr->spp);
        const uint32_t band_pix = (uint32_t)std::min<uint64_t>(n_pixels, std::max<uint64_t>(tile, budget_paths / spass / tile * tile));
        PT_HIP(ctx, hipSetDevice(ctx->device));
        const size_t need = (size_t)spass * band_pix;
        if (need > ctx->cap_prad) {
            hipFree(ctx->d_prad);
            ctx->d_prad = nullptr; ctx->cap_prad = 0;
            PT_HIP(ctx, hipMalloc(&ctx->d_prad, need * 3 * sizeof(float)));
            ctx->cap_prad = need;
        }
        if (!ctx->d_ticket) PT_HIP(ctx, hipMalloc(&ctx->d_ticket, sizeof(uint32_t)));
        StageTimer tm(ctx, stream, (pr->flags & PT_RENDER_STAGE_TIMES) != 0);
        hipEvent_t ev0 = tm.get(), ev1 = tm.get();
        PT_HIP(ctx, hipEventRecord(ev0, stream));
        PT_HIP(ctx, hipMemsetAsync(ctx->d_cnt, 0, sizeof(Counters), stream));
        unsigned long long *dbg = nullptr;
        const size_t dbg_waves = (size_t)grid_max * (block / 64);
        if (kProf) { // diagnostic build: per-wave start / end clocks and step counts, reported after the call
            PT_HIP(ctx, hipMalloc(&dbg, dbg_waves * 32 + 64));
            PT_HIP(ctx, hipMemset(dbg, 0, dbg_waves * 32 + 64));
        }
        uint32_t n_passes = 0;
        const pt_status pst = stream_passes(ctx, pr, d_rgb_sum, stream, sh, count, scene, ctx->d_prad, need, ctx->d_ticket, ctx->d_cnt, &tm, &n_passes, dbg, dbg_waves);
        if (pst != PT_OK) return pst;
        PT_HIP(ctx, hipGetLastError());
        PT_HIP(ctx, hipEventRecord(ev1, stream));
        if (dbg) { // diagnostic build: how long the waves of the last launch lived, and what their longest rays cost
            const size_t nw = dbg_waves;
            std::vector<unsigned long long> hd(nw * 4 + 8);
            PT_HIP(ctx, hipStreamSynchronize(stream));
            PT_HIP(ctx, hipMemcpy(hd.data(), dbg, nw * 32 + 64, hipMemcpyDeviceToHost));
            hipFree(dbg);
            unsigned long long t0 = ~0ull, t1 = 0;
            for (size_t w = 0; w < nw; ++w)
                if (hd[2 * w]) { t0 = std::min(t0, hd[2 * w]); t1 = std::max(t1, hd[2 * w + 1]); }
            const double span = (double)(t1 - t0);
            int late = 0, ran = 0, hist[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            double life = 0;
            std::vector<size_t> idx;
            for (size_t w = 0; w < nw; ++w) {
                if (!hd[2 * w]) continue;
                ran++;
                idx.push_back(w);
                if ((double)(hd[2 * w] - t0) > 0.05 * span) late++;
                life += (double)(hd[2 * w + 1] - hd[2 * w]);
                hist[(int)(9.999 * (double)(hd[2 * w + 1] - t0) / span)]++;
            }
            fprintf(stderr, "[prof] k_paths_stream: %d waves ran, %.1f ms from first start to last end, %d started late, mean life %.2f of that; waves ending per tenth:", ran, span / 1e5 /* 100 MHz */, late, life / std::max(ran, 1) / span);
            for (int k = 0; k < 10; ++k) fprintf(stderr, " %d", hist[k]);
            fprintf(stderr, "\n");
            std::sort(idx.begin(), idx.end(), [&](size_t x, size_t y) { return hd[2 * x + 1] > hd[2 * y + 1]; });
            for (size_t k = 0; k < std::min<size_t>(4, idx.size()); ++k) {
                const size_t w = idx[k];
                fprintf(stderr, "[prof]   wave %zu (among the last to end): life %.1f ms, %llu loop iterations, %llu serves, %llu rays, longest ray %llu steps\n", w, (double)(hd[2 * w + 1] - hd[2 * w]) / 1e5,
                        hd[2 * nw + 2 * w] >> 32, hd[2 * nw + 2 * w] & 0xFFFFFFFFull, hd[2 * nw + 2 * w + 1] & 0xFFFFFFFFull, hd[2 * nw + 2 * w + 1] >> 32);
            }
            const float *f = reinterpret_cast<const float *>(hd.data() + 4 * nw);
            if (f[7] > 0.0f)
                fprintf(stderr, "[prof]   a ray of more than 50000 steps: o %.9g %.9g %.9g d %.9g %.9g %.9g, nearest t %.9g, %.0f steps, bounce %.0f, cone slope %.6g\n", f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7], f[8], f[10]);
        }
        if (stats) {
            memset(stats, 0, sizeof *stats);
            Counters h;
            PT_HIP(ctx, hipMemcpyAsync(&h, ctx->d_cnt, sizeof h, hipMemcpyDeviceToHost, stream));
            PT_HIP(ctx, hipStreamSynchronize(stream));
            if (kProf) { // how often each part of the loop was issued by a wave, and with how many of its lanes
                static const char *const part[5] = {"four-wide step", "two-child step", "leaf step", "shading", "new path"};
                const double rays = (double)std::max<unsigned long long>(h.rays, 1ull);
                fprintf(stderr, "[prof] k_paths_stream: %.3f loop iterations and %.3f serves per 64 rays\n", 64.0 * (double)h.sect[0] / rays, 64.0 * (double)h.sect[1] / rays);
                for (int k = 0; k < 5; ++k)
                    fprintf(stderr, "[prof]   %-15s %8.3f issues per 64 rays, %7.3f per ray by its own lane, lanes active %5.1f %%\n", part[k], (double)h.slots[k] / rays,
                            (double)h.lanes[k] / rays, 100.0 * (double)h.lanes[k] / (double)std::max(h.slots[k], 1ull));
            }
            stats->rays = h.rays;
            stats->primary_rays = (uint64_t)pr->spp * n_pixels;
            stats->box_tests = h.box_tests;
            stats->prim_tests = h.prim_tests;
            stats->exact_fallbacks = h.exact_fallbacks;
            stats->exact_replays = h.exact_replays;
            stats->launches_intersect = n_passes;
            stats->passes = n_passes;
            stats->kernel = PT_KERNEL_PATHS_STREAM;
            float ms = 0.0f;
            PT_HIP(ctx, hipEventElapsedTime(&ms, ev0, ev1));
            stats->ms_total = ms;
            if (tm.on) {
                double acc = 0;
                for (size_t k = 0; k < tm.stage.size(); ++k) {
                    float m = 0.0f;
                    PT_HIP(ctx, hipEventElapsedTime(&m, ctx->events[2 + 2 * k], ctx->events[3 + 2 * k]));
                    acc += m;
                }
                stats->ms_paths = acc;
            }
        }
        return PT_OK;
    }

    // Paths per slice: 32k amortises a slice's ramp-down best, but a frame needs a few slices per resident
    // workgroup to balance (a slice's cost varies several-fold with what its pixels see).
    const uint64_t total_paths = (uint64_t)n_pixels * pr->spp;
    bool ring = sh.ring && pr->max_depth <= PT_RING_MAX_DEPTH;
    uint32_t spass = 0, pix = 0, cap = 0;
    for (;;) {
        // k_paths_ring keeps its queue filled across slices, so its slices only have to be small enough to balance the end
        // of the frame (16 per resident workgroup) and large enough to be worth a ticket
        const uint32_t balanced = ring ? (uint32_t)std::min<uint64_t>(8192, std::max<uint64_t>(2048, total_paths / (16ull * grid_max)))
                                       : (uint32_t)std::min<uint64_t>(32768, std::max<uint64_t>(4096, total_paths / (4ull * grid_max)));
        const uint32_t target = (uint32_t)tuned(ctx->tune.slice_paths, (int)balanced);
        spass = pr->samples_per_pass ? pr->samples_per_pass : std::max(1u, target / 64u);
        spass = std::min(spass, pr->spp);
        pix = std::max(64u, (target / spass) & ~63u);
        if ((uint64_t)pix * spass * 3 >= (1ull << 32)) return fail(ctx, PT_ERR_LIMIT, "samples_per_pass too large for one slice");
        cap = pix * spass;
        if (ring && cap > (1u << PT_RING_LID_BITS)) { ring = false; continue; } // a slice too large for the ring's path index: k_paths
        break;
    }
    PT_HIP(ctx, hipSetDevice(ctx->device));
    // Slices: full-sized ones while plenty of work is left, then sizes falling with what remains (what is
    // left / resident workgroups, never under a quarter of the full size), so that the workgroups finish the
    // frame together: a slice costs its workgroup 0.5-4 ms and several-fold more or less with what its pixels see.
    const uint32_t pix_min = std::max(64u, (pix / (uint32_t)tuned(ctx->tune.slice_taper, 4)) & ~63u);
    if (ctx->ext_key[0] != n_pixels || ctx->ext_key[1] != pix || ctx->ext_key[2] != grid_max || ctx->ext_key[3] != pix_min || !ctx->n_extents) {
        std::vector<uint2> ext;
        for (uint32_t at = 0; at < n_pixels;) {
            const uint32_t left = n_pixels - at;
            uint32_t take = std::min(pix, std::max(pix_min, (left / grid_max) & ~63u));
            take = std::min(take, left);
            ext.push_back(make_uint2(at, take));
            at += take;
        }
        // A previous call on another stream may still read the table (the event wait that orders calls behind each other
        // orders device work only, and the copy below is host-timed): wait for it to finish first.
        if (ctx->have_last) PT_HIP(ctx, hipEventSynchronize(ctx->last_done));
        if (ext.size() > ctx->cap_extents) {
            hipFree(ctx->d_extents);
            ctx->d_extents = nullptr; ctx->cap_extents = 0;
            PT_HIP(ctx, hipMalloc(&ctx->d_extents, ext.size() * sizeof(uint2)));
            ctx->cap_extents = ext.size();
        }
        PT_HIP(ctx, hipMemcpy(ctx->d_extents, ext.data(), ext.size() * sizeof(uint2), hipMemcpyHostToDevice));
        ctx->ext_key[0] = n_pixels; ctx->ext_key[1] = pix; ctx->ext_key[2] = grid_max; ctx->ext_key[3] = pix_min;
        ctx->n_extents = (uint32_t)ext.size();
    }
    const uint32_t n_slices = ctx->n_extents;
    const uint32_t grid = std::min(grid_max, n_slices);

    // k_paths_ring: a new slice joins a queue of fewer than low_water rays
    // (four chunks per wave: random spheres 13.6 ms at 1024 .. 4096 against 13.9 at 512 and 14.5 at 16384, where every generation
    // takes a slice; the Cornell box is flat from 256 to 1024)
    const uint32_t low_water = std::max(1u, (uint32_t)tuned(ctx->tune.ring_low_water, 4 * block));
    const uint32_t qcap = low_water + cap;
    if (ring) {
        const size_t need_q = (size_t)grid * 2 * qcap, need_r = (size_t)grid * PT_RING * cap;
        if (need_q > ctx->cap_ring_queue) {
            hipFree(ctx->d_gA); hipFree(ctx->d_gB); hipFree(ctx->d_gC);
            ctx->d_gA = ctx->d_gB = nullptr; ctx->d_gC = nullptr; ctx->cap_ring_queue = 0;
            PT_HIP(ctx, hipMalloc(&ctx->d_gA, need_q * sizeof(float4)));
            PT_HIP(ctx, hipMalloc(&ctx->d_gB, need_q * sizeof(float4)));
            PT_HIP(ctx, hipMalloc(&ctx->d_gC, need_q * sizeof(float2)));
            ctx->cap_ring_queue = need_q;
        }
        if (need_r > ctx->cap_ring_rad) {
            hipFree(ctx->d_grad);
            ctx->d_grad = nullptr; ctx->cap_ring_rad = 0;
            PT_HIP(ctx, hipMalloc(&ctx->d_grad, need_r * 3 * sizeof(float)));
            ctx->cap_ring_rad = need_r;
        }
    }
    const size_t need = ring ? 0 : (size_t)grid * cap;
    if (need > ctx->cap_slice_paths) {
        hipFree(ctx->d_sA); hipFree(ctx->d_sB); hipFree(ctx->d_sC); hipFree(ctx->d_srad);
        ctx->d_sA = ctx->d_sB = nullptr; ctx->d_sC = nullptr; ctx->d_srad = nullptr; ctx->cap_slice_paths = 0;
        PT_HIP(ctx, hipMalloc(&ctx->d_sA, need * 2 * sizeof(float4)));
        PT_HIP(ctx, hipMalloc(&ctx->d_sB, need * 2 * sizeof(float4)));
        PT_HIP(ctx, hipMalloc(&ctx->d_sC, need * 2 * sizeof(float2)));
        PT_HIP(ctx, hipMalloc(&ctx->d_srad, need * 3 * sizeof(float)));
        ctx->cap_slice_paths = need;
    }
    if (!ctx->d_ticket) PT_HIP(ctx, hipMalloc(&ctx->d_ticket, sizeof(uint32_t)));

    StageTimer tm(ctx, stream, (pr->flags & PT_RENDER_STAGE_TIMES) != 0);
    hipEvent_t ev0 = tm.get(), ev1 = tm.get();
    PT_HIP(ctx, hipEventRecord(ev0, stream));
    PT_HIP(ctx, hipMemsetAsync(ctx->d_cnt, 0, sizeof(Counters), stream));
    PassDev ps;
    ps.width = (uint32_t)pr->width; ps.height = (uint32_t)pr->height; ps.n_pixels = n_pixels;
    ps.k0 = (uint32_t)pr->seed; ps.k1 = (uint32_t)(pr->seed >> 32);
    ps.max_depth = pr->max_depth;
    SliceDev sl;
    sl.qA = ctx->d_sA; sl.qB = ctx->d_sB; sl.qC = ctx->d_sC; sl.rad = ctx->d_srad;
    sl.cap = cap; sl.pix_per_slice = pix; sl.n_slices = n_slices; sl.extents = ctx->d_extents; sl.ticket = ctx->d_ticket;
    // Rays that head up are queued apart (k_paths, SPLIT) where they can leave the scene -- a background that is not black --
    // on the kernel shape the form is built for (spheres in LDS); pt_tuning.split_queue overrides the first condition.
    const bool open_scene = scene.bg[0] != 0.0f || scene.bg[1] != 0.0f || scene.bg[2] != 0.0f;
    const bool split = tuned(ctx->tune.split_queue, open_scene ? 1 : 0) != 0;
    uint32_t n_passes = 0;
    for (uint32_t done = 0; done < pr->spp; done += spass, ++n_passes) {
        ps.samples = std::min(spass, pr->spp - done);
        ps.sample_base = pr->sample_offset + done;
        PT_HIP(ctx, hipMemsetAsync(ctx->d_ticket, 0, sizeof(uint32_t), stream));
        tm.begin(1);
        if (ring) {
            RingDev rg;
            rg.qA = ctx->d_gA; rg.qB = ctx->d_gB; rg.qC = ctx->d_gC; rg.rad = ctx->d_grad;
            rg.qcap = qcap; rg.cap = cap; rg.low_water = low_water; rg.n_slices = n_slices; rg.extents = ctx->d_extents; rg.ticket = ctx->d_ticket;
            if (count) launch_ring<true>(ctx, scene, stream, block, split, grid, lds_bytes, ps, rg, d_rgb_sum);
            else launch_ring<false>(ctx, scene, stream, block, split, grid, lds_bytes, ps, rg, d_rgb_sum);
        } else if (count) launch_paths<true>(ctx, scene, stream, block, lds, sort, split, grid, lds_bytes, ps, sl, d_rgb_sum);
        else launch_paths<false>(ctx, scene, stream, block, lds, sort, split, grid, lds_bytes, ps, sl, d_rgb_sum);
        tm.end();
    }
    PT_HIP(ctx, hipGetLastError());
    PT_HIP(ctx, hipEventRecord(ev1, stream));
    if (stats) {
        memset(stats, 0, sizeof *stats);
        Counters h;
        PT_HIP(ctx, hipMemcpyAsync(&h, ctx->d_cnt, sizeof h, hipMemcpyDeviceToHost, stream));
        PT_HIP(ctx, hipStreamSynchronize(stream));
        stats->rays = h.rays;
        stats->primary_rays = (uint64_t)pr->spp * n_pixels;
        stats->box_tests = h.box_tests;
        stats->prim_tests = h.prim_tests;
        stats->exact_fallbacks = h.exact_fallbacks;
        stats->exact_replays = h.exact_replays;
        stats->launches_intersect = n_passes;
        stats->passes = n_passes;
        stats->kernel = ring ? PT_KERNEL_PATHS_RING : PT_KERNEL_PATHS;
        if (kProf) {
            static const char *names[6] = {"slice setup", "ray fetch / camera", "trace", "shade + emit", "bounce barriers", "film"};
            unsigned long long tot = 0;
            for (int k = 0; k < 6; ++k) tot += h.sect[k];
            for (int k = 0; k < 6; ++k) fprintf(stderr, "[prof] %-20s %5.1f %% of wave time\n", names[k], 100.0 * (double)h.sect[k] / (double)std::max(tot, 1ull));
            if (ring) fprintf(stderr, "[prof] after the last slice was taken: %.1f %% of wave time, %.1f generations per workgroup holding %.0f rays in all\n",
                              100.0 * (double)h.sect[6] / (double)std::max(tot, 1ull), (double)h.sect[7] / (double)grid, (double)h.lanes[5] / (double)grid);
            static const char *ln[6] = {"node visits", "object tests", "sphere-sample tries", "shade calls", "sphere roots", "-"};
            for (int k = 0; k < 5; ++k)
                fprintf(stderr, "[prof] %-20s lanes %llu slots %llu utilisation %.1f %%\n", ln[k], h.lanes[k], h.slots[k], 100.0 * (double)h.lanes[k] / (double)std::max(h.slots[k], 1ull));
        }
        float ms = 0.0f;
        PT_HIP(ctx, hipEventElapsedTime(&ms, ev0, ev1));
        stats->ms_total = ms;
        if (tm.on) {
            double acc = 0;
            for (size_t k = 0; k < tm.stage.size(); ++k) {
                float m = 0.0f;
                PT_HIP(ctx, hipEventElapsedTime(&m, ctx->events[2 + 2 * k], ctx->events[3 + 2 * k]));
                acc += m;
            }
            stats->ms_paths = acc;
        }
    }
    return PT_OK;
}

static pt_status stream_passes(pt_context *ctx, const pt_render_params *pr, float *d_rgb_sum, hipStream_t stream, const PathsShape &sh, bool count,
                               SceneDev scene, float *rad, uint64_t rad_cap_paths, uint32_t *ticket, Counters *cnt, StageTimer *tm,
                               uint32_t *n_passes_out, unsigned long long *dbg, size_t dbg_waves)
{
    const pt_tuning &tu = ctx->tune;
    const uint32_t n_pixels = (uint32_t)pr->width * (uint32_t)pr->height;
    const int block = sh.block;
    const uint32_t tile = (uint32_t)tuned(tu.stream_tile, 64);
    uint32_t spass = pr->samples_per_pass ? pr->samples_per_pass : std::min(pr->spp, 256u);
    spass = std::min(spass, pr->spp);
    // A frame whose pass fits the buffer is one band whatever its pixel count; only a frame that has to be cut is cut
    // at tile boundaries (a band's last tile may be short, a band's first may not).
    uint32_t band_pix = n_pixels;
    if ((uint64_t)spass * n_pixels > rad_cap_paths) {
        if ((uint64_t)spass * tile > rad_cap_paths) return fail(ctx, PT_ERR_LIMIT, "samples_per_pass too large for the pass buffer");
        band_pix = (uint32_t)(rad_cap_paths / spass / tile * tile);
    }
    if ((uint64_t)spass * band_pix >= (1ull << 32) - (1ull << 24)) return fail(ctx, PT_ERR_LIMIT, "samples_per_pass too large for one pass"); // the path counter overshoots by up to 64 per wave
    PassDev ps;
    ps.width = (uint32_t)pr->width; ps.height = (uint32_t)pr->height; ps.n_pixels = n_pixels;
    ps.k0 = (uint32_t)pr->seed; ps.k1 = (uint32_t)(pr->seed >> 32);
    ps.max_depth = pr->max_depth;
    const uint32_t serve_min = (uint32_t)tuned(tu.stream_serve_min, 32);
    // One kind of step per iteration (the two kinds share no code: run in the same iteration each executes with the
    // lanes of the other masked off), leaf steps as soon as a fifth of the traversing lanes wait for one: with the step
    // loop, 1M spheres 4.52 / 4.44 / 4.33 / 4.19 Grays/s for stream_phase 1 / 2 / 3 / 4, 100k mixed objects 5.03 / 5.01 /
    // 4.98 / 4.91 and 4.81 for 0 (both kinds in every iteration).
    const uint32_t phase_bias = (uint32_t)tuned(tu.stream_phase, 1);
    // runs of at most 256 paths: 1M spheres 3.41 -> 3.50 Grays/s, 100k mixed objects 3.56 -> 3.73 against 1024 (64 and 128 tie with 256)
    const uint32_t run_max = (uint32_t)tuned(tu.stream_run, 256);
    const void *kernel = stream_kernel_for(count, block, ctx->kind, sh.wide);
    uint32_t n_passes = 0;
    BandDev band;
    band.tile = tile;
    for (band.pix0 = 0; band.pix0 < n_pixels; band.pix0 += band_pix) {
        band.npix = std::min(band_pix, n_pixels - band.pix0);
        for (uint32_t done = 0; done < pr->spp; done += spass, ++n_passes) {
            ps.samples = std::min(spass, pr->spp - done);
            ps.sample_base = pr->sample_offset + done;
            const uint64_t paths = (uint64_t)ps.samples * band.npix;
            const uint32_t grid = (uint32_t)std::min<uint64_t>(sh.grid_max, (paths + block - 1) / block);
            // a wave's run: long enough to keep its lanes on neighbouring pixels, short enough for 16 runs per wave
            band.run = (uint32_t)std::min<uint64_t>(run_max, std::max<uint64_t>(64, paths / ((uint64_t)grid * (block / 64) * 16) / 64 * 64));
            PT_HIP(ctx, hipMemsetAsync(ticket, 0, sizeof(uint32_t), stream));
            if (tm) tm->begin(1);
            {
                float *a_rad = rad;
                uint32_t *a_ticket = ticket;
                Counters *a_cnt = cnt;
                uint32_t a_serve = serve_min, a_phase = phase_bias, a_dbgw = (uint32_t)dbg_waves;
                void *args[] = {&scene, &ps, &band, &a_rad, &a_ticket, &a_cnt, &a_serve, &a_phase, &dbg, &a_dbgw};
                PT_HIP(ctx, hipLaunchKernel(kernel, dim3(grid), dim3(block), args, sh.stream_stack_bytes, stream));
            }
            if (tm) tm->end();
            hipLaunchKernelGGL(k_film, dim3(grid_for(ctx, band.npix)), dim3(PT_BLOCK), 0, stream, ps, (const float *)rad, d_rgb_sum, (const uint32_t *)nullptr, cnt,
                               band.pix0, band.npix);
        }
    }
    if (n_passes_out) *n_passes_out = n_passes;
    return PT_OK;
}

static pt_status check_render_args(pt_context *ctx, const pt_render_params *pr, const float *d_rgb_sum)
{
    if (!ctx->has_scene) return fail(ctx, PT_ERR_NO_SCENE, "no scene uploaded");
    if (!pr || !d_rgb_sum) return fail(ctx, PT_ERR_INVALID, "null parameters or output buffer");
    if (pr->width <= 0 || pr->height <= 0 || pr->spp == 0 || pr->max_depth <= 0)
        return fail(ctx, PT_ERR_INVALID, "width, height, spp and max_depth must be positive");
    if (pr->max_depth > 4096) return fail(ctx, PT_ERR_LIMIT, "max_depth above 4096");
    if ((uint64_t)pr->width * (uint64_t)pr->height > (1ull << 28)) return fail(ctx, PT_ERR_LIMIT, "image larger than 2^28 pixels");
    if ((uint64_t)pr->sample_offset + pr->spp > 0xFFFFFFFFull) return fail(ctx, PT_ERR_LIMIT, "sample index overflows 32 bits");
    return PT_OK;
}

// pt_render_device without allocation, queries or synchronisation: launches and memsets only, scratch in the
// library's own per-GPU array (see `scratch` above).
extern "C" pt_status pt_render_device_noalloc(pt_context *ctx, const pt_render_params *pr, float *d_rgb_sum, void *stream_)
{
    if (!ctx) return PT_ERR_INVALID;
    const pt_status chk = check_render_args(ctx, pr, d_rgb_sum);
    if (chk != PT_OK) return chk;
    if ((pr->flags & ~(uint32_t)PT_RENDER_GEOMETRIC_CULL) != 0) return fail(ctx, PT_ERR_INVALID, "pt_render_device_noalloc takes no flag but PT_RENDER_GEOMETRIC_CULL");
    if ((uint64_t)pr->width * (uint64_t)pr->height * pr->spp > (1ull << 32)) return fail(ctx, PT_ERR_LIMIT, "pt_render_device_noalloc: more than 2^32 paths (width * height * spp) in one call");
    if (!ctx->scratch) return fail(ctx, PT_ERR_DEVICE, "no scratch array on this device");
    if (!ctx->stream_plan) return fail(ctx, PT_ERR_NO_SCENE, "no launch plan: upload the scene first");
    hipStream_t stream = (hipStream_t)stream_;
    PT_HIP(ctx, hipSetDevice(ctx->device));
    unsigned char *base = static_cast<unsigned char *>(ctx->scratch); // never the array's name: that is a host-side stand-in
    uint32_t *ticket = reinterpret_cast<uint32_t *>(base);
    Counters *cnt = reinterpret_cast<Counters *>(base + 256);
    float *rad = reinterpret_cast<float *>(base + PT_SCRATCH_HEADER);
    static_assert(256 + sizeof(Counters) <= PT_SCRATCH_HEADER, "scratch header too small");
    // the array keeps the previous call's contents: the counters must start at zero (the pass counter is zeroed per pass)
    PT_HIP(ctx, hipMemsetAsync(base, 0, PT_SCRATCH_HEADER, stream));
    SceneDev scene = ctx->sc;
    if (pr->flags & PT_RENDER_GEOMETRIC_CULL) {
        scene.cone.kappa_max = scene.cone.kappa_min = 0.0f;
        scene.cone.adaptive = 0;
    }
    const pt_status st = stream_passes(ctx, pr, d_rgb_sum, stream, ctx->stream_plan->shape, false, scene, rad, PT_NOALLOC_SCRATCH_BYTES / 12u, ticket, cnt,
                                       nullptr, nullptr, nullptr, 0);
    if (st != PT_OK) return st;
    PT_HIP(ctx, hipGetLastError());
    return PT_OK;
}

// A call that enqueues work on a caller's stream and returns shares the context's workspace (counters, queues, pass
// buffer) with the next one: each such call first makes its stream wait for the end of the previous one.
static pt_status order_behind_last(pt_context *ctx, hipStream_t stream)
{
    if (ctx->have_last) PT_HIP(ctx, hipStreamWaitEvent(stream, ctx->last_done, 0));
    return PT_OK;
}
static pt_status mark_last(pt_context *ctx, hipStream_t stream)
{
    PT_HIP(ctx, hipEventRecord(ctx->last_done, stream));
    ctx->have_last = true;
    return PT_OK;
}

static pt_status render_device_impl(pt_context *ctx, const pt_render_params *pr, float *d_rgb_sum, hipStream_t stream, pt_stats *stats);

extern "C" pt_status pt_render_device(pt_context *ctx, const pt_render_params *pr, float *d_rgb_sum, void *stream_,
                                      pt_stats *stats)
{
    if (!ctx) return PT_ERR_INVALID;
    const pt_status chk = check_render_args(ctx, pr, d_rgb_sum);
    if (chk != PT_OK) return chk;
    hipStream_t stream = (hipStream_t)stream_;
    PT_HIP(ctx, hipSetDevice(ctx->device));
    pt_status st = order_behind_last(ctx, stream);
    if (st != PT_OK) return st;
    st = render_device_impl(ctx, pr, d_rgb_sum, stream, stats);
    const pt_status ms = mark_last(ctx, stream); // also after a failure part-way: what was enqueued still uses the workspace
    return st != PT_OK ? st : ms;
}

static pt_status render_device_impl(pt_context *ctx, const pt_render_params *pr, float *d_rgb_sum, hipStream_t stream, pt_stats *stats)
{
    const uint32_t n_pixels = (uint32_t)pr->width * (uint32_t)pr->height;
    if ((pr->flags & PT_RENDER_SPLIT_KERNELS) == 0) return render_slices(ctx, pr, d_rgb_sum, stream, stats);
    SceneDev scene = ctx->sc;
    if (pr->flags & PT_RENDER_GEOMETRIC_CULL) {
        scene.cone.kappa_max = scene.cone.kappa_min = 0.0f;
        scene.cone.adaptive = 0;
    }

    // samples of every pixel in flight per pass: as many as fit the path budget
    const uint64_t budget = 1ull << 27; // 134M paths in flight = 13.4 GB of queues: tail bounces are paid once per pass
    uint32_t spass = pr->samples_per_pass ? pr->samples_per_pass : (uint32_t)std::max<uint64_t>(1, budget / n_pixels);
    spass = std::min(spass, pr->spp);
    if ((uint64_t)spass * n_pixels > (1ull << 31)) return fail(ctx, PT_ERR_LIMIT, "samples_per_pass * pixels above 2^31");
    const size_t paths = (size_t)spass * n_pixels;
    pt_status st = ensure_workspace(ctx, paths, (size_t)pr->max_depth + 1);
    if (st != PT_OK) return st;

    const bool timed = (pr->flags & PT_RENDER_STAGE_TIMES) != 0;
    const bool count = (pr->flags & PT_RENDER_COUNT_TESTS) != 0;
    StageTimer tm(ctx, stream, timed);
    hipEvent_t ev0 = tm.get(), ev1 = tm.get();
    tm.used = 2;
    PT_HIP(ctx, hipEventRecord(ev0, stream));
    PT_HIP(ctx, hipMemsetAsync(ctx->d_cnt, 0, sizeof(Counters), stream));

    PassDev ps;
    ps.width = (uint32_t)pr->width; ps.height = (uint32_t)pr->height; ps.n_pixels = n_pixels;
    ps.k0 = (uint32_t)pr->seed; ps.k1 = (uint32_t)(pr->seed >> 32);
    ps.max_depth = pr->max_depth;
    ctx->lds_scene = (pr->flags & PT_RENDER_NO_LDS_SCENE) == 0;
    uint32_t n_passes = 0, n_isect = 0;
    for (uint32_t done = 0; done < pr->spp; done += spass, ++n_passes) {
        ps.samples = std::min(spass, pr->spp - done);
        ps.sample_base = pr->sample_offset + done;
        const size_t np = (size_t)ps.samples * n_pixels;
        const uint32_t grid = grid_for(ctx, np);
        PT_HIP(ctx, hipMemsetAsync(ctx->d_nactive, 0, ((size_t)pr->max_depth + 1) * sizeof(uint32_t), stream));
        PathPlanes cur{ctx->d_A[0], ctx->d_B[0], ctx->d_C[0]}, nxt{ctx->d_A[1], ctx->d_B[1], ctx->d_C[1]};
        tm.begin(0);
        hipLaunchKernelGGL(k_raygen, dim3(grid), dim3(PT_BLOCK), 0, stream, scene, ps, cur, ctx->d_nactive);
        tm.end();
        for (int32_t b = 0; b < pr->max_depth; ++b) {
            tm.begin(1);
            launch_intersect(ctx, scene, stream, cur, ctx->d_nactive + b, np, count);
            tm.end();
            ++n_isect;
            tm.begin(2);
            hipLaunchKernelGGL(k_shade, dim3(grid), dim3(PT_BLOCK), 0, stream, scene, ps, cur, ctx->d_hits, nxt, ctx->d_nactive + b,
                               ctx->d_nactive + b + 1, ctx->d_rad, b);
            tm.end();
            std::swap(cur, nxt);
        }
        tm.begin(3);
        hipLaunchKernelGGL(k_film, dim3(grid_for(ctx, n_pixels)), dim3(PT_BLOCK), 0, stream, ps, ctx->d_rad, d_rgb_sum, ctx->d_nactive, ctx->d_cnt);
        tm.end();
    }
    PT_HIP(ctx, hipGetLastError());
    PT_HIP(ctx, hipEventRecord(ev1, stream));

    if (stats) {
        memset(stats, 0, sizeof *stats);
        Counters h;
        PT_HIP(ctx, hipMemcpyAsync(&h, ctx->d_cnt, sizeof h, hipMemcpyDeviceToHost, stream));
        PT_HIP(ctx, hipStreamSynchronize(stream));
        stats->rays = h.rays;
        stats->primary_rays = (uint64_t)pr->spp * n_pixels;
        stats->box_tests = h.box_tests;
        stats->prim_tests = h.prim_tests;
        stats->exact_fallbacks = h.exact_fallbacks;
        stats->exact_replays = h.exact_replays;
        stats->launches_intersect = n_isect;
        stats->passes = n_passes;
        stats->kernel = PT_KERNEL_SPLIT;
        float ms = 0.0f;
        PT_HIP(ctx, hipEventElapsedTime(&ms, ev0, ev1));
        stats->ms_total = ms;
        if (timed) {
            double acc[4] = {0, 0, 0, 0};
            for (size_t k = 0; k < tm.stage.size(); ++k) {
                float m = 0.0f;
                PT_HIP(ctx, hipEventElapsedTime(&m, ctx->events[2 + 2 * k], ctx->events[3 + 2 * k]));
                acc[tm.stage[k]] += m;
            }
            stats->ms_raygen = acc[0]; stats->ms_intersect = acc[1]; stats->ms_shade = acc[2]; stats->ms_film = acc[3];
        }
    }
    return PT_OK;
}

// Pinned host memory for the callers' film buffers: a device-to-host copy into it runs at the link's rate (about
// 50 GB/s) instead of the 9 GB/s of a copy into pageable memory, which the runtime stages through its own buffers.
extern "C" void *pt_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}
extern "C" void pt_host_free(void *p)
{
    if (p) hipHostFree(p);
}

// true if [p, p + bytes) is pinned host memory the device can copy into directly
static bool is_pinned_host(const void *p)
{
    hipPointerAttribute_t at;
    memset(&at, 0, sizeof at);
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError(); // an ordinary malloc'ed pointer is reported as an error: not one
        return false;
    }
    return at.type == hipMemoryTypeHost;
}

// Device buffer -> caller's host buffer on `stream`, synchronised.  A pinned destination (pt_host_alloc) is written
// directly.  A pageable one goes through the context's pinned staging buffer in chunks, the copy of chunk k + 1 from
// the device overlapping the host's copy of chunk k out of the staging buffer.
static hipError_t copy_film_to_host(pt_context *ctx, float *dst, const float *d_src, size_t bytes, hipStream_t stream)
{
    hipError_t e;
    if (is_pinned_host(dst)) {
        if ((e = hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, stream)) != hipSuccess) return e;
        return hipStreamSynchronize(stream);
    }
    const size_t chunk = 4u << 20;
    if (ctx->cap_hfilm < 2 * chunk) {
        if (ctx->h_film) hipHostFree(ctx->h_film);
        ctx->h_film = nullptr; ctx->cap_hfilm = 0;
        if ((e = hipHostMalloc(reinterpret_cast<void **>(&ctx->h_film), 2 * chunk, hipHostMallocDefault)) != hipSuccess) return e;
        ctx->cap_hfilm = 2 * chunk;
    }
    if (ctx->events.size() < 2) {
        while (ctx->events.size() < 2) {
            hipEvent_t ev;
            if ((e = hipEventCreate(&ev)) != hipSuccess) return e;
            ctx->events.push_back(ev);
        }
    }
    char *stage[2] = {reinterpret_cast<char *>(ctx->h_film), reinterpret_cast<char *>(ctx->h_film) + chunk};
    const char *src = reinterpret_cast<const char *>(d_src);
    char *out = reinterpret_cast<char *>(dst);
    const size_t n = (bytes + chunk - 1) / chunk;
    auto len = [&](size_t k) { return std::min(chunk, bytes - k * chunk); };
    if ((e = hipMemcpyAsync(stage[0], src, len(0), hipMemcpyDeviceToHost, stream)) != hipSuccess) return e;
    if ((e = hipEventRecord(ctx->events[0], stream)) != hipSuccess) return e;
    for (size_t k = 0; k < n; ++k) {
        if (k + 1 < n) {
            if ((e = hipMemcpyAsync(stage[(k + 1) & 1], src + (k + 1) * chunk, len(k + 1), hipMemcpyDeviceToHost, stream)) != hipSuccess) return e;
            if ((e = hipEventRecord(ctx->events[(k + 1) & 1], stream)) != hipSuccess) return e;
        }
        if ((e = hipEventSynchronize(ctx->events[k & 1])) != hipSuccess) return e;
        memcpy(out + k * chunk, stage[k & 1], len(k));
    }
    return hipSuccess;
}

extern "C" pt_status pt_render(pt_context *ctx, const pt_render_params *pr, float *rgb_sum_host, pt_stats *stats)
{
    if (!ctx) return PT_ERR_INVALID;
    if (!pr || !rgb_sum_host) return fail(ctx, PT_ERR_INVALID, "null parameters or output buffer");
    if (pr->width <= 0 || pr->height <= 0) return fail(ctx, PT_ERR_INVALID, "width and height must be positive");
    PT_HIP(ctx, hipSetDevice(ctx->device));
    const size_t bytes = (size_t)pr->width * pr->height * 3 * sizeof(float);
    if (bytes > ctx->cap_film) { // the device film is kept across calls (the reference's fb lives as long as main, main.cu:167-170)
        PT_HIP(ctx, hipStreamSynchronize(ctx->own_stream));
        hipFree(ctx->d_film);
        ctx->d_film = nullptr; ctx->cap_film = 0;
        PT_HIP(ctx, hipMalloc(&ctx->d_film, bytes));
        ctx->cap_film = bytes;
    }
    PT_HIP(ctx, hipMemsetAsync(ctx->d_film, 0, bytes, ctx->own_stream));
    const pt_status st = pt_render_device(ctx, pr, ctx->d_film, ctx->own_stream, stats);
    if (st != PT_OK) return st;
    const hipError_t e = copy_film_to_host(ctx, rgb_sum_host, ctx->d_film, bytes, ctx->own_stream);
    if (e != hipSuccess) return fail(ctx, PT_ERR_DEVICE, std::string("pt_render: ") + hipGetErrorString(e));
    return PT_OK;
}

// main.cu:95-98 and :192-205 (host).
extern "C" pt_status pt_film_finalize(const float *rgb_sum, int32_t width, int32_t height, uint32_t total_spp, float *rgb_gamma,
                                      uint8_t *rgb8)
{
    if (!rgb_sum || width <= 0 || height <= 0 || total_spp == 0) return PT_ERR_INVALID;
    const float inv = 1.0f / (float)total_spp; // vec3.h:136-138: operator/= multiplies by 1/t
    size_t w = 0;
    for (int32_t j = height - 1; j >= 0; --j) {
        for (int32_t i = 0; i < width; ++i) {
            const size_t px = (size_t)j * width + i;
            for (int c = 0; c < 3; ++c) {
                float g = sqrtf(rgb_sum[px * 3 + c] * inv);
                if (rgb_gamma) rgb_gamma[px * 3 + c] = g;
                if (rgb8) {
                    float cl = g < 0.0f ? 0.0f : (g > 0.999f ? 0.999f : g); // commons.h:58-62
                    rgb8[w] = (uint8_t)(int)(256 * cl);
                }
                ++w;
            }
        }
    }
    return PT_OK;
}
