// pt_trace.h -- device code shared by every kernel of libpathtracer_amd.so: the scene as the kernels see it, the nearest
// hit through the flattened tree with the list scan's semantics (hittable_list.h:87-101), the camera ray
// (camera.h:88-96), material scatter / emitted (material.h) and the records the kernels exchange with the host.
// Everything here is inlined into the kernels of the pt_kernels_*.hip files; pt_context.hip includes it for the
// parameter structs.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "pt_bvh.h"
#include "pt_device.h"

#define PT_BLOCK 256
#define PT_WIDE_STACK_MAX 31u // entries of the four-wide traversal's stack at most (what five workgroups of 256 lanes leave each lane of a CU's LDS)
// what a scene holds (the kernels' KIND parameter; see bvh_nearest)
#define PT_KIND_SPHERES 0
#define PT_KIND_NO_CONES 1
#define PT_KIND_ANY 2
#define PT_WAVES (PT_BLOCK / 64)

// ---------------------------------------------------------------------------------------------
// Device-side views
// ---------------------------------------------------------------------------------------------
// the first words of the table cone_table_hit() searches (layout there)
struct ConeTabHeader { uint32_t n, lo, hi, shift, nb; float flo, fhi; };

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
#ifdef PT_LDS_WIDE4
    uint32_t n_wide4; // experiment: four-wide unquantised nodes (in `wide`) for the LDS-resident kernels
#endif
    const uint32_t *cone_tab; // the cones' distinct slopes, for cone_table_hit(); null for a scene without cones
    ConeTabHeader cone_hdr;   // its header words, as kernel arguments (a scalar load per look-up otherwise)
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
PT_DI HitResult linear_nearest(const PrimRec *__restrict__ list, uint32_t n, const Ray &r, float t_min, float t_max, uint32_t first = 0)
{
    HitResult best;
    best.t = t_max;
    best.slot = PT_NO_PRIM;
    for (uint32_t i = first; i < n; ++i) {
        PrimRec q = list[i];
        float t;
        if (hit_any(q, r, t_min, best.t, t)) {
            best.t = t;
            best.slot = i;
        }
    }
    return best;
}

// Cones that a ray "hits" wherever they stand.  cone.h:108 computes the quadratic's leading coefficient
// a = |d_xz|^2 - tangens d_y d_y.
//  * When the two terms round to the same float a is exactly 0, the discriminant is half_b^2, one of the roots
//    (cone.h:116-120) is 0 / 0 = NaN, and the reference's range and clip tests (all of the form "reject if x < bound") let
//    a NaN through: the cone is accepted at t = NaN by a ray that may pass nowhere near it, and from that list entry on
//    the scan accepts whatever the ray's line crosses, in list order.  About one ray in 16 000 of the 100k mixed objects.
//  * When they differ by a unit or two in the last place, a is a few 2^-24ths of its terms, the roots are the rounding
//    noise of -half_b -+ sqrtd divided by it -- a ratio of small integers times a power of two: 0.5, 6, 768 -- and now and
//    then such a number passes the range test and the clip: the cone is hit at a point that has nothing to do with it.
//    A few rays in 10^9.
// No tree can find such cones by where they stand, but they can be found by their slope: fl(fl(tangens d_y) d_y) is
// monotone in tangens, so the values that (nearly) cancel a given ray lie within a few bit patterns of
// q = |d_xz|^2 / d_y^2, and the scene's distinct tangens values are kept in a table keyed by their bit patterns.
// Layout (uint32 words):
//   [0] n values  [1] bits of the smallest  [2] bits of the largest  [3] shift (>= 7)  [4] cells B  [7] cones kept m
//   [5], [6] the smallest value times (1 - 2^-10) and the largest times (1 + 2^-10), as float bits
//   [8 ..]  B cells of four words: (value, position, value, position) of the at most two values whose
//           (bits - [1]) >> shift is the cell's number, position = the value's rank among the n; unused entries hold
//           PT_CONE_TAB_NONE, and a cell that would need more than two holds PT_CONE_TAB_MORE in its first word
//   then the n values ascending (float bits), the largest list index that has each, n + 1 offsets into the member
//   array, and the m members: the list indices of the cones, grouped by value.
// cone_table_hit() returns bit 0 when some cone's a is exactly 0 for direction d, with the largest list index of such a
// cone (the scan's answer is then the scan's answer from that entry on: the entry is accepted whatever came before it,
// and with the running t a NaN nothing that came before matters), and bit 1 when some cone's a is within 2^-20 of its
// terms without being 0, with the range of positions of such values (the caller then tries those cones themselves:
// cone_near_acceptable).  A range test on the ray alone comes first (no memory; three rays in four end there on the 100k
// mixed objects), then one 16-byte load (two where the candidates straddle a cell boundary).
// (A one-bit-per-16-patterns bitmap in front of the cells, its word asked for when the ray starts and looked at when the
// traversal is over, was measured and is no faster: 100k mixed objects at 8 spp 71.5 against 70.7 ms.)
#define PT_CONE_TAB_HEADER 8u
#define PT_CONE_TAB_NONE 0xFFFFFFFFu
#define PT_CONE_TAB_MORE 0xFFFFFFFEu
#define PT_CONE_TAB_NEAR (1.0f / 1048576.0f) // |a| up to this share of its terms counts as nearly cancelled (eight units in the last place at least)
#define PT_CONE_TAB_SLACK 48u // q is off by at most six bit patterns (d_y^2, the reciprocal, the product and the reference's two products round once each); 2^-20 is 8 to 16 more, and their roundings 2
PT_DI uint32_t cone_table_hit(const uint32_t *__restrict__ tab, const ConeTabHeader &hd, v3 d, uint32_t &last_index, uint32_t &near_lo, uint32_t &near_hi)
{
    const float S = d.x * d.x + d.z * d.z; // |d_xz|^2 as vlen2 of (d_x, 0, d_z) rounds it (cone.h:108, vec3.h:154-156)
    const float dy2 = d.y * d.y;
    last_index = 0;
    near_lo = 0xFFFFFFFFu;
    near_hi = 0;
    if (!(S >= dy2 * hd.flo && S <= dy2 * hd.fhi)) return 0u; // (a NaN ends here too)
    const uint32_t lo = hd.lo, span = hd.hi - lo, shift = hd.shift;
    const uint32_t key = min(max(__float_as_uint(S * __builtin_amdgcn_rcpf(dy2)), lo), hd.hi) - lo;
    const uint32_t b0 = (key > PT_CONE_TAB_SLACK ? key - PT_CONE_TAB_SLACK : 0u) >> shift, b1 = min(key + PT_CONE_TAB_SLACK, span) >> shift;
    const uint4 *cells = reinterpret_cast<const uint4 *>(tab + PT_CONE_TAB_HEADER);
    const uint32_t n = hd.n, nb = hd.nb;
    const float *val = reinterpret_cast<const float *>(tab + PT_CONE_TAB_HEADER + 4u * (size_t)nb);
    const uint32_t *last = reinterpret_cast<const uint32_t *>(val + n);
    const float tol = S * PT_CONE_TAB_NEAR;
    uint32_t found = 0;
    auto check = [&](float v, uint32_t pos) { // tangens * r.d.y * r.d.y against |d_xz|^2, cone.h:108
        const float a = S - (v * d.y) * d.y; // (PT_CONE_TAB_NONE is a NaN: neither branch)
        if (a == 0.0f) {
            found |= 1u;
            last_index = max(last_index, last[pos]);
        } else if (fabsf(a) <= tol) {
            found |= 2u;
            near_lo = min(near_lo, pos);
            near_hi = max(near_hi, pos + 1u);
        }
    };
    const uint4 c0 = cells[b0];
    check(__uint_as_float(c0.x), c0.y); check(__uint_as_float(c0.z), c0.w);
    uint4 c1 = c0;
    if (b1 != b0) { // the candidates straddle a cell boundary (b1 = b0 + 1: a cell is at least 128 bit patterns wide)
        c1 = cells[b1];
        check(__uint_as_float(c1.x), c1.y); check(__uint_as_float(c1.z), c1.w);
    }
    if (c0.x == PT_CONE_TAB_MORE || c1.x == PT_CONE_TAB_MORE) { // a crowded cell: the sorted values
        uint32_t i0 = 0, i1 = n;
        while (i0 < i1) { // the first value whose term is not more than tol below S
            const uint32_t mid = (i0 + i1) >> 1;
            if (S - (val[mid] * d.y) * d.y > tol) i0 = mid + 1u;
            else i1 = mid;
        }
        for (; i0 < n && S - (val[i0] * d.y) * d.y >= -tol; ++i0) check(val[i0], i0);
    }
    return found;
}

// Whether the reference's cone test (cone.h:101-129) can accept one of the cones of the table's values [near_lo, near_hi)
// for this ray at all, under any running t: a cone that cannot is the tree's to ignore, one that can sends the ray to the
// list scan.  Cones whose a is exactly 0 are skipped (they are the other case of cone_table_hit).
static __device__ __attribute__((noinline)) bool cone_near_acceptable(const uint32_t *__restrict__ tab, const PrimRec *__restrict__ list, float ox, float oy, float oz, float dx,
                                                                      float dy, float dz, float t_min, float t_max, uint32_t near_lo, uint32_t near_hi)
{
    Ray r;
    r.o = V(ox, oy, oz);
    r.d = V(dx, dy, dz);
    const uint32_t n = tab[0], nb = tab[4];
    const uint32_t *first = tab + PT_CONE_TAB_HEADER + 4u * (size_t)nb + 2u * (size_t)n, *member = first + n + 1u;
    for (uint32_t j = first[near_lo]; j < first[near_hi]; ++j) {
        const PrimRec q = list[member[j]];
        ConeRoots cr;
        if (!cone_roots(q, r, cr) || cr.a == 0.0f) continue;
        if ((!(cr.r1 < t_min || t_max < cr.r1) && cone_clip_ok(cr, cr.r1)) || (!(cr.r2 < t_min || t_max < cr.r2) && cone_clip_ok(cr, cr.r2))) return true;
    }
    return false;
}

// After the tree walk of a scene with cones: need_exact set and the list entry the scan has to start at, for a ray with
// a cone whose leading coefficient is exactly 0 (from that cone on) or nearly so and acceptable (from the start; with
// both, from the cone on: the scan itself then meets the other); else 0 and need_exact as it was.
template <int KIND>
PT_DI uint32_t cone_scan_from(const SceneDev &sc, const Ray &r, float t_min, float t_max, bool &need_exact)
{
    if (KIND != PT_KIND_ANY || !sc.cone_tab) return 0;
    uint32_t from = 0, near_lo, near_hi;
    const uint32_t found = cone_table_hit(sc.cone_tab, sc.cone_hdr, r.d, from, near_lo, near_hi);
    if (!found) return 0;
    if ((found & 2u) && !cone_near_acceptable(sc.cone_tab, sc.prims_list, r.o.x, r.o.y, r.o.z, r.d.x, r.d.y, r.d.z, t_min, t_max, near_lo, near_hi) && !(found & 1u)) return 0;
    need_exact = true;
    return (found & 1u) ? from : 0u;
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
    static constexpr bool wide4 = false;
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
#ifdef PT_LDS_WIDE4
// Experiment: four-wide nodes with unquantised planes in LDS.  A node is seven quads (112 bytes: the stride spreads the
// lanes of a read over all sixteen 16-byte bank slots): per axis (lo of the four children, hi of the four children), then the
// references.  The quad a ray meets first on an axis is at byte 0 or 16 of the axis' pair by the sign of its direction.
struct LdsSrc {
    static constexpr bool in_lds = true;
    static constexpr bool wide4 = true;
    const char *nodes;
    const float4 *prims;
    uint32_t n_nodes, n_prims;
    pt::ConeParams cone;
    static __device__ __forceinline__ void adapt(BoxRay &br)
    {
        br.sx = br.sx ? 16u : 0u;
        br.sy = 32u + (br.sy ? 16u : 0u);
        br.sz = 64u + (br.sz ? 16u : 0u);
    }
    __device__ __forceinline__ void quads(int32_t i, const BoxRay &br, float4 &nx, float4 &ny, float4 &nz, float4 &fx, float4 &fy, float4 &fz, int4 &c) const
    {
        const char *base = nodes + (uint32_t)i * 112u;
        nx = *reinterpret_cast<const float4 *>(base + br.sx); fx = *reinterpret_cast<const float4 *>(base + (br.sx ^ 16u));
        ny = *reinterpret_cast<const float4 *>(base + br.sy); fy = *reinterpret_cast<const float4 *>(base + (br.sy ^ 16u));
        nz = *reinterpret_cast<const float4 *>(base + br.sz); fz = *reinterpret_cast<const float4 *>(base + (br.sz ^ 16u));
        c = *reinterpret_cast<const int4 *>(base + 96);
    }
    // (the two-wide interface, for code that is compiled but never runs with this source: scenes with cones stay out of LDS in this build)
    __device__ __forceinline__ void planes(int32_t, const BoxRay &, float2 &nx, float2 &ny, float2 &nz, float2 &fx, float2 &fy, float2 &fz, int32_t &c0, int32_t &c1) const
    {
        nx = ny = nz = fx = fy = fz = make_float2(3.0e38f, 3.0e38f);
        c0 = c1 = -1;
    }
    __device__ __forceinline__ void prim(uint32_t slot, float4 &w0, float4 &w1) const
    {
        w0 = prims[slot]; w1 = prims[n_prims + slot];
    }
};
#else
struct LdsSrc {
    static constexpr bool in_lds = true;
    static constexpr bool wide4 = false;
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
#endif

// Copies the scene into the workgroup's LDS (nodes in the 80-byte form, objects as two planes of quads) and returns
// the first byte after it.  The caller synchronises the workgroup before traversing.
template <int BLOCK>
PT_DI char *lds_scene_copy(const SceneDev &sc, float4 *s_mem, LdsSrc &lsrc)
{
    char *s_nodes = reinterpret_cast<char *>(s_mem);
    const float4 *gp = reinterpret_cast<const float4 *>(sc.prims);
#ifdef PT_LDS_WIDE4
    float4 *s_prims = s_mem + 7u * sc.n_wide4;
    for (uint32_t i = threadIdx.x; i < 7u * sc.n_wide4; i += BLOCK) s_mem[i] = sc.wide[i];
#else
    float4 *s_prims = reinterpret_cast<float4 *>(s_nodes + ((sc.n_nodes * PT_LDS_NODE_BYTES + 15u) & ~15u));
    const float4 *gn = reinterpret_cast<const float4 *>(sc.nodes);
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
#endif
    for (uint32_t i = threadIdx.x; i < 2 * sc.n_prims; i += BLOCK) s_prims[(i & 1u) * sc.n_prims + (i >> 1)] = gp[i];
    lsrc.nodes = s_nodes;
    lsrc.prims = s_prims;
    return reinterpret_cast<char *>(s_prims + 2 * sc.n_prims);
}
static size_t lds_scene_size(uint32_t n_nodes /* PT_LDS_WIDE4: four-wide ones */, uint32_t n_prims)
{
#ifdef PT_LDS_WIDE4
    return (size_t)n_nodes * 112u + (size_t)n_prims * 32;
#else
    return (((size_t)n_nodes * PT_LDS_NODE_BYTES + 15u) & ~(size_t)15u) + (size_t)n_prims * 32;
#endif
}

#define PT_SCAN_MAX 16 // candidates the exact replay keeps per ray

// The list scan replayed exactly, for the rays whose answer depends on list order.  Whatever the
// running t_max, the scan can only ever accept an object one of whose own roots lies in
// [t_min, t_max] and passes its clip; one traversal WITHOUT nearest-hit pruning collects those
// objects (a handful: the surfaces the ray really crosses), they are ordered by list index, and
// hittable_list::hit (hittable_list.h:87-101) is replayed over them.  `overflow` is set when more
// than PT_SCAN_MAX candidates exist (the caller then scans the whole list).  `first`: only list entries from this
// index on count (the scan of a ray that a cone accepts at t = NaN whatever came before: cone_table_hit).
template <int STRIDE, class Src>
static __device__ __attribute__((noinline)) HitResult bvh_exact_scan(const Src &src, uint32_t stack_cap, const Ray &r, const BoxRay &br, float t_min, float t_max,
                               uint32_t *stack, bool &overflow, uint32_t first = 0)
{
    uint32_t ids[PT_SCAN_MAX], slots[PT_SCAN_MAX];
    int n = 0;
    overflow = false;
    const bool steep = br.afx == 0.0f || br.afy == 0.0f || br.afz == 0.0f;
    SteepBounds sb = {0.0f, -3.0e38f, 0.0f, -3.0e38f, 0.0f, -3.0e38f, 0.0f};
    if (steep) sb = make_steep_bounds(r, ray_kappa(src.cone, r.o));
    uint32_t sp = 0;
    int32_t cur = 0;
    const uint32_t first_id = first; // (the leaf loop below has a `first` of its own)
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
                if (possible && q.id >= first_id) {
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
#ifdef PT_LDS_WIDE4
        if constexpr (Src::wide4 && !FLAT) {
        while (cur >= 0) { // experiment: four children per visit, nearest first
            float4 qnx, qny, qnz, qfx, qfy, qfz;
            int4 qc;
            src.quads(cur, br, qnx, qny, qnz, qfx, qfy, qfz, qc);
            const float nxs[4] = {qnx.x, qnx.y, qnx.z, qnx.w}, nys[4] = {qny.x, qny.y, qny.z, qny.w}, nzs[4] = {qnz.x, qnz.y, qnz.z, qnz.w};
            const float fxs[4] = {qfx.x, qfx.y, qfx.z, qfx.w}, fys[4] = {qfy.x, qfy.y, qfy.z, qfy.w}, fzs[4] = {qfz.x, qfz.y, qfz.z, qfz.w};
            int32_t cr[4] = {qc.x, qc.y, qc.z, qc.w};
            uint32_t key[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float nx = __builtin_fmaf(nxs[k], br.anx, br.cnx), ny = __builtin_fmaf(nys[k], br.any, br.cny), nz = __builtin_fmaf(nzs[k], br.anz, br.cnz);
                const float fx = __builtin_fmaf(fxs[k], br.afx, br.cfx), fy = __builtin_fmaf(fys[k], br.afy, br.cfy), fz = __builtin_fmaf(fzs[k], br.afz, br.cfz);
                float m, tn, tf;
                asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(nz), "v"(t_min));
                asm("v_max3_f32 %0, %1, %2, %3" : "=v"(tn) : "v"(nx), "v"(ny), "v"(m));
                asm("v_min_f32 %0, %1, %2" : "=v"(m) : "v"(fz), "v"(best_t));
                asm("v_min3_f32 %0, %1, %2, %3" : "=v"(tf) : "v"(fx), "v"(fy), "v"(m));
                const bool hit = tn <= tf + br.eps && cr[k] != pt::kWideAbsent;
                key[k] = ((__float_as_uint(tn) & ~3u) | (uint32_t)k) | (hit ? 0u : 0xFFFFFFFFu); // tn >= t_min > 0: its bits order as it does
            }
            if (COUNT) n_box += 4;
            auto order2 = [](uint32_t &a, uint32_t &b, int32_t &ra, int32_t &rb) {
                const bool s = b < a;
                const uint32_t lo = s ? b : a, hi = s ? a : b;
                const int32_t rlo = s ? rb : ra, rhi = s ? ra : rb;
                a = lo; b = hi; ra = rlo; rb = rhi;
            };
            order2(key[0], key[1], cr[0], cr[1]); order2(key[2], key[3], cr[2], cr[3]); order2(key[0], key[2], cr[0], cr[2]);
            order2(key[1], key[3], cr[1], cr[3]); order2(key[1], key[2], cr[1], cr[2]);
            if (key[3] != 0xFFFFFFFFu) stack[(sp++) * STRIDE] = (uint32_t)cr[3];
            if (key[2] != 0xFFFFFFFFu) stack[(sp++) * STRIDE] = (uint32_t)cr[2];
            if (key[1] != 0xFFFFFFFFu) stack[(sp++) * STRIDE] = (uint32_t)cr[1];
            cur = cr[0];
            if (key[0] == 0xFFFFFFFFu) cur = (int32_t)stack[(--sp) * STRIDE];
        }
        }
#endif
        while (!FLAT && !Src::wide4 && cur >= 0) {
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
static __device__ __forceinline__ unsigned long long wave_list_scan(const PrimRec *__restrict__ list, uint32_t first /* the entry the scan starts at */, uint32_t n, float ox, float oy,
                                                                 float oz, float dx, float dy, float dz, float t_min, float t_max)
{
    Ray r;
    r.o = V(ox, oy, oz);
    r.d = V(dx, dy, dz);
    const uint32_t lane = threadIdx.x & 63u;
    HitResult best; // wave-uniform
    best.t = t_max;
    best.slot = PT_NO_PRIM;
    for (uint32_t base = first; base < n; base += 64u) {
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
static __device__ __attribute__((noinline)) unsigned long long wave_list_scan_call(const PrimRec *__restrict__ list, uint32_t first, uint32_t n, float ox, float oy, float oz, float dx,
                                                                                 float dy, float dz, float t_min, float t_max)
{
    return wave_list_scan(list, first, n, ox, oy, oz, dx, dy, dz, t_min, t_max);
}


PT_DI Ray load_ray_xyz(const float *o, const float *d, size_t i)
{
    Ray r;
    r.o = V(o[3 * i], o[3 * i + 1], o[3 * i + 2]);
    r.d = V(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
    return r;
}

// ---------------------------------------------------------------------------------------------
// What the integrator kernels share
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

#define PT_ISECT_LDS_BLOCK 1024 // workgroup size of the LDS-resident variant: two of them fill a CU's 32 wave slots

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

// pt_scatter_batch: one record per hit
struct ScatterOut { uint32_t scattered; float atten[3], o[3], d[3], emitted[3]; };

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

// A slice table and the private queues of k_paths (pt_kernels_paths.hip)
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

// k_paths_ring (pt_kernels_ring.hip)
#define PT_RING 16           // slices a workgroup may have under way
#define PT_RING_PATH_BITS 18 // a path within its slice: pixel (the low RingDev::pix_bits bits) and sample of the pass (the bits above)
#define PT_RING_SLOT_BITS 4
#define PT_RING_MAX_DEPTH ((1 << (32 - PT_RING_PATH_BITS - PT_RING_SLOT_BITS)) - 1) // the bounce count has the word's other 10 bits
struct RingDev {
    float4 *qA;   // [grid][2][qcap]
    float4 *qB;
    float2 *qC;
    float *rad;   // [grid][PT_RING][cap * 3]
    uint32_t qcap;      // entries per queue: low_water + cap
    uint32_t cap;       // paths of the largest slice
    uint32_t low_water; // a generation that starts with fewer queued rays than this takes the next slice as well
    uint32_t pix_bits;  // bits of a path word that hold the pixel within the slice (every slice has at most 2^pix_bits pixels)
    uint32_t n_slices;
    const uint2 *extents; // [n_slices] (first pixel, pixels)
    uint32_t *ticket;
};

// k_paths_stream (pt_kernels_stream.hip)
#ifndef PT_STREAM_WAVES
#define PT_STREAM_WAVES 4 // waves per SIMD of every k_paths_stream variant (117-128 VGPRs)
#endif
// One pass of k_paths_stream: `samples` samples (PassDev) of the pixels [pix0, pix0 + npix).
struct BandDev {
    uint32_t pix0, npix;
    uint32_t tile; // pixels per tile of the path numbering (a multiple of 64)
    uint32_t run;  // consecutive paths a wave takes at a time
};

