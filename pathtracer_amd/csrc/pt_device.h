// pt_device.h -- device-side data layout and arithmetic of the gfx950 path tracer.
//
// Arithmetic contract: this translation unit is compiled with -ffp-contract=off, IEEE fp32
// divide/sqrt, denormals kept.  Every expression that decides or reports a hit, a normal, a
// scattered ray or a colour is written in the reference's operation order (file:line cited per
// function), so its fp32 result is the reference's, bit for bit.  Only the BVH box test, which
// merely has to be conservative, is free-form.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pt_api.h"

#define PT_FLT_MAX 3.402823466e+38f
#define PT_MAX_TRIES 64    // bound on the reference's unbounded rejection loops (cuda_functions.h:16-18,67-69)
#define PT_LEAF_MAX 4      // objects per BVH leaf for scenes traversed from LDS (1 for larger ones)
#define PT_NO_PRIM 0xFFFFFFFFu

// ---------------------------------------------------------------------------------------------
// Records in HBM.  Both are 16-byte aligned so that one lane moves them with dwordx4 loads.
// ---------------------------------------------------------------------------------------------

// One object, 32 B: six parameters (pt_api.h, pt_scene_desc), a packed word and the list index.
//   meta bits 0-2 object tag, bits 3-4 material tag, bits 5-31 material index.
struct __attribute__((aligned(16))) PrimRec {
    float p[6];
    uint32_t meta;
    uint32_t id;
};
static_assert(sizeof(PrimRec) == 32, "PrimRec must be 32 bytes");

// One inner node, 64 B: the boxes of both children as one quad per axis, b[a] = (lo of child 0, lo of
// child 1, hi of child 0, hi of child 1) on axis a, and their references.  The half of a quad that holds
// the planes a ray meets first is chosen by the sign of the ray's direction on that axis, i.e. by address.
//   ref >= 0: inner node index.  ref < 0: leaf, ~ref = (first_prim << 3) | (count - 1).
//   An absent child has the degenerate box lo = hi = +inf (no ray reaches it) and ref = leaf of count 1 at 0.
struct __attribute__((aligned(16))) BvhNode {
    float b[3][4];
    int32_t c[2];
    uint32_t pad[2];
};
static_assert(sizeof(BvhNode) == 64, "BvhNode must be 64 bytes");

struct DeviceCamera { // camera.h:125-145
    float origin[3], llc[3], horizontal[3], vertical[3], u[3], v[3];
    float lens_radius;
};

// ---------------------------------------------------------------------------------------------
// vec3.h restated
// ---------------------------------------------------------------------------------------------
struct v3 { float x, y, z; };

#define PT_DI static __device__ __forceinline__

PT_DI v3 V(float a, float b, float c) { v3 r; r.x = a; r.y = b; r.z = c; return r; }
PT_DI v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }                                   // vec3.h:73-75
PT_DI v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }               // vec3.h:217-219
PT_DI v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }               // vec3.h:228-230
PT_DI v3 vmul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }               // vec3.h:239-241
PT_DI v3 vscale(float t, v3 a) { return V(t * a.x, t * a.y, t * a.z); }                // vec3.h:250-252
PT_DI v3 vdiv(v3 a, float t) { return vscale(1.0f / t, a); }                           // vec3.h:272-274
PT_DI float vdot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }             // vec3.h:283-287
PT_DI float vlen2(v3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }                  // vec3.h:154-156
PT_DI float vlen(v3 a) { return sqrtf(vlen2(a)); }                                     // vec3.h:145-147
PT_DI v3 vunit(v3 a) { return vdiv(a, vlen(a)); }                                      // vec3.h:308-310
PT_DI bool vnear_zero(v3 a)                                                            // vec3.h:163-166
{
    const float s = 1e-8f;
    return (fabsf(a.x) < s) && (fabsf(a.y) < s) && (fabsf(a.z) < s);
}
PT_DI v3 vreflect(v3 v, v3 n) { return vsub(v, vscale(2.0f * vdot(v, n), n)); }        // vec3.h:319-321
PT_DI v3 vrefract(v3 uv, v3 n, float etai_over_etat)                                   // vec3.h:331-336
{
    float cos_theta = fminf(vdot(vneg(uv), n), 1.0f);
    v3 r_out_perp = vscale(etai_over_etat, vadd(uv, vscale(cos_theta, n)));
    // vec3.h:334 subtracts in double (the literal 1.0), then sqrtf rounds to float first
    v3 r_out_parallel = vscale(-sqrtf((float)fabs(1.0 - (double)vlen2(r_out_perp))), n);
    return vadd(r_out_perp, r_out_parallel);
}
PT_DI v3 ray_at(v3 o, v3 d, float t) { return vadd(o, vscale(t, d)); }                 // ray.h:47-49

// ---------------------------------------------------------------------------------------------
// Counter-based random stream: Philox4x32-7, counter (block, bounce, sample, pixel), key = seed.
// Stands in for the per-pixel curandState (main.cu:34,87); no state buffer exists.  The stream of one
// (pixel, sample, bounce) is the words of blocks 0, 1, 2, ... in order, one word per uniform drawn
// (curand_uniform's role): a unit-sphere try takes three, a unit-disk try two, so four sphere tries come
// out of three blocks.  Seven rounds are the fewest the Philox paper certifies (BigCrush); ten are its default.
// ---------------------------------------------------------------------------------------------
#define PT_PHILOX_ROUNDS 7
struct Rng { uint32_t pixel, sample, bounce, block, k0, k1; };

PT_DI void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4])
{
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < PT_PHILOX_ROUNDS; ++r) {
        // one 32x32->64 multiply per product (v_mad_u64_u32) instead of separate high and low halves
        const unsigned long long p0 = (unsigned long long)M0 * c0, p1 = (unsigned long long)M1 * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        // a ^ b ^ key in one instruction: gfx950's three-input boolean op with truth table 0x96, the
        // (wave-uniform) round key as its scalar operand
        uint32_t n0, n2;
        asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(n0) : "v"(hi1), "v"(c1), "s"(k0));
        asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(n2) : "v"(hi0), "v"(c3), "s"(k1));
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += W0; k1 += W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// the next block of the stream
PT_DI void rng_words(Rng &g, uint32_t o[4])
{
    philox4x32(g.block, g.bounce, g.sample, g.pixel, g.k0, g.k1, o);
    g.block++;
}

// (0,1] in steps of 2^-24: the range of curand_uniform (main.cu:90)
PT_DI float u32_to_uniform(uint32_t x) { return (float)((x >> 8) + 1u) * (1.0f / 16777216.0f); }

// 2u - 1 for the uniform u = ((x >> 8) + 1) * 2^-24 that u32_to_uniform() makes of a random word, as the
// samplers below need it (cuda_functions.h:17,68: 2.0f * vec3(u..) - vec3(1..)).  The reference rounds once
// (2u is exact); so does fma(256 k, 2^-31, 2^-23 - 1) with k = x >> 8, whose product and sum are exact
// before the rounding (256 k = x with its low byte cleared has 24 significant bits): the same float from
// three instructions instead of six, and a mask instead of a shift (shifts issue at half rate on gfx950).
PT_DI float u32_to_signed_unit(uint32_t x)
{
    return __builtin_fmaf((float)(x & 0xFFFFFF00u), 1.0f / 2147483648.0f, 1.0f / 8388608.0f - 1.0f);
}

// cuda_functions.h:14-20.  Try k takes words 3k, 3k+1, 3k+2 of the stream, which must stand at a block
// boundary on entry (it does: the scatter of a bounce draws nothing before it).
PT_DI v3 random_in_unit_sphere(Rng &g)
{
    v3 p;
    int tries = 0;
    uint32_t a[4], b[4];
#define PT_SPHERE_TRY(X, Y, Z)                                                                 \
    p = V(u32_to_signed_unit(X), u32_to_signed_unit(Y), u32_to_signed_unit(Z));                \
    if (!(vlen2(p) >= 1.0f && ++tries < PT_MAX_TRIES)) break;
    for (;;) {
        rng_words(g, a);
        PT_SPHERE_TRY(a[0], a[1], a[2])
        rng_words(g, b);
        PT_SPHERE_TRY(a[3], b[0], b[1])
        rng_words(g, a);
        PT_SPHERE_TRY(b[2], b[3], a[0])
        PT_SPHERE_TRY(a[1], a[2], a[3])
    }
#undef PT_SPHERE_TRY
    return p;
}
// cuda_functions.h:65-71 after the two uniforms of the pixel jitter (main.cu:90-91), which are words 0 and 1
// of the camera stream: `first` is that block, whose words 2 and 3 are the first try.
PT_DI v3 random_in_unit_disk_after_jitter(Rng &g, const uint32_t first[4])
{
    v3 p = V(u32_to_signed_unit(first[2]), u32_to_signed_unit(first[3]), 0.0f);
    int tries = 0;
    while (vlen2(p) >= 1.0f && ++tries < PT_MAX_TRIES) {
        uint32_t o[4];
        rng_words(g, o);
        p = V(u32_to_signed_unit(o[0]), u32_to_signed_unit(o[1]), 0.0f);
        if (!(vlen2(p) >= 1.0f && ++tries < PT_MAX_TRIES)) break;
        p = V(u32_to_signed_unit(o[2]), u32_to_signed_unit(o[3]), 0.0f);
    }
    return p;
}

// ---------------------------------------------------------------------------------------------
// Objects.  hit_*() restate hittable::hit with the reference's own (t_min, t_max) acceptance
// tests; they return the accepted root only -- point and normal are recomputed from (object, t)
// by surface_at(), with the same expressions, once the nearest object is known.
// ---------------------------------------------------------------------------------------------
struct Ray { v3 o, d; };

// sphere.h:85-101
PT_DI bool hit_sphere(const PrimRec &q, const Ray &r, float t_min, float t_max, float &t_out)
{
    v3 center = V(q.p[0], q.p[1], q.p[2]);
    float radius = q.p[3];
    v3 oc = vsub(r.o, center);
    float a = vlen2(r.d);
    float half_b = vdot(oc, r.d);
    float c = vlen2(oc) - radius * radius;
    float discriminant = half_b * half_b - a * c;
    if (discriminant < 0) return false;
    float sqrtd = sqrtf(discriminant);
    float root = (-half_b - sqrtd) / a;
    if (root < t_min || t_max < root) {
        root = (-half_b + sqrtd) / a;
        if (root < t_min || t_max < root) return false;
    }
    t_out = root;
    return true;
}

// hit_sphere() in two halves, for the traversal: the discriminant (sphere.h:85-93; false when negative)
// and the roots with their range tests (sphere.h:94-101).  Same operations in the same order.
PT_DI bool sphere_discriminant(const PrimRec &q, const Ray &r, float &half_b, float &discriminant)
{
    v3 center = V(q.p[0], q.p[1], q.p[2]);
    float radius = q.p[3];
    v3 oc = vsub(r.o, center);
    float a = vlen2(r.d);
    half_b = vdot(oc, r.d);
    float c = vlen2(oc) - radius * radius;
    discriminant = half_b * half_b - a * c;
    return !(discriminant < 0);
}
PT_DI bool sphere_roots(const Ray &r, float half_b, float discriminant, float t_min, float t_max, float &t_out)
{
    float a = vlen2(r.d);
    float sqrtd = sqrtf(discriminant);
    float root = (-half_b - sqrtd) / a;
    if (root < t_min || t_max < root) {
        root = (-half_b + sqrtd) / a;
        if (root < t_min || t_max < root) return false;
    }
    t_out = root;
    return true;
}

PT_DI float comp(v3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

// aarect.h:348-355 (AX=2: xy), :364-371 (AX=1: xz), :380-387 (AX=0: yz)
template <int AX, int A0, int A1>
PT_DI bool hit_rect(const PrimRec &q, const Ray &r, float t_min, float t_max, float &t_out)
{
    float t = (q.p[4] - comp(r.o, AX)) / comp(r.d, AX);
    if (t < t_min || t > t_max) return false;
    float c0 = comp(r.o, A0) + t * comp(r.d, A0);
    float c1 = comp(r.o, A1) + t * comp(r.d, A1);
    if (c0 < q.p[0] || c0 > q.p[1] || c1 < q.p[2] || c1 > q.p[3]) return false;
    t_out = t;
    return true;
}

// cylinder.h:96-123
PT_DI bool hit_cylinder(const PrimRec &q, const Ray &r, float t_min, float t_max, float &t_out)
{
    v3 center = V(q.p[0], q.p[1], q.p[2]);
    float height = q.p[3], radius = q.p[4];
    v3 oc = vsub(r.o, center);
    oc.y = 0.0f;
    v3 dxz = r.d;
    dxz.y = 0.0f;
    float a = vlen2(dxz);
    float half_b = vdot(oc, dxz);
    float c = vlen2(oc) - (radius * radius);
    float discriminant = half_b * half_b - a * c;
    if (discriminant < 0) return false;
    float sqrtd = sqrtf(discriminant);
    float root = (-half_b - sqrtd) / a;
    if (root < t_min || t_max < root) {
        root = (-half_b + sqrtd) / a;
        if (root < t_min || t_max < root) return false;
    }
    float py = r.o.y + root * r.d.y; // ray.h:47-49, y component
    if (py < center.y || py > center.y + height) return false;
    t_out = root;
    return true;
}

// cone.h:101-129.  The quadratic's leading coefficient a can be negative (a ray steeper than the
// cone's side); then (-half_b - sqrtd)/a is the LARGER root and the reference's answer depends on
// the t_max it is called with, i.e. on what earlier list entries were hit.  The two roots are
// therefore reported separately:
//   *t_first  = first-tried root if it lies in [t_min, t_max] (NaN-permissive like the reference)
//               and passes the y clip; first_state: 0 not in range, 1 in range but clipped, 2 hit
//   *t_second = second root if it is >= t_min, passes the y clip; valid only when second_ok.
// With a >= 0 (or NaN) the second root never beats the first (it is the larger one), and calling
// with the running t_max reproduces the list scan exactly, so callers use hit_cone_ordered().
struct ConeRoots { float a, r1, r2, cy, cyh, oy, dy; };

PT_DI bool cone_roots(const PrimRec &q, const Ray &r, ConeRoots &cr)
{
    v3 center = V(q.p[0], q.p[1], q.p[2]);
    float height = q.p[3], tangens = q.p[5];
    v3 oc = vsub(r.o, center);
    oc.y = 0.0f;
    v3 dxz = r.d;
    dxz.y = 0.0f;
    float D = height - r.o.y + center.y;
    float a = vlen2(dxz) - tangens * r.d.y * r.d.y;
    float half_b = vdot(oc, dxz) + tangens * r.d.y * D;
    float c = vlen2(oc) - tangens * D * D;
    float discriminant = half_b * half_b - a * c;
    if (discriminant < 0) return false;
    float sqrtd = sqrtf(discriminant);
    cr.a = a;
    cr.r1 = (-half_b - sqrtd) / a;
    cr.r2 = (-half_b + sqrtd) / a;
    cr.cy = center.y;
    cr.cyh = center.y + height;
    cr.oy = r.o.y;
    cr.dy = r.d.y;
    return true;
}
PT_DI bool cone_clip_ok(const ConeRoots &cr, float root)
{
    float py = cr.oy + root * cr.dy;
    return !(py < cr.cy || py > cr.cyh);
}
// the reference's control flow, verbatim (cone.h:116-129), for a given t_max
PT_DI bool hit_cone_ordered(const ConeRoots &cr, float t_min, float t_max, float &t_out)
{
    float root = cr.r1;
    if (root < t_min || t_max < root) {
        root = cr.r2;
        if (root < t_min || t_max < root) return false;
    }
    if (!cone_clip_ok(cr, root)) return false;
    t_out = root;
    return true;
}

// Point, outward normal, face flag: sphere.h:103-107, aarect.h:356-360 (+2 variants),
// cone.h:125-136, cylinder.h:119-127, hittable.h:29-32.
// (SPH: the scene holds spheres only, the caller knows)
template <bool SPH = false>
PT_DI void surface_at(const PrimRec &q, const Ray &r, float t, v3 &p, v3 &normal, bool &front)
{
    p = ray_at(r.o, r.d, t);
    v3 outward;
    switch (SPH ? (uint32_t)PT_SPHERE : q.meta & 7u) {
    case PT_SPHERE: outward = vscale(q.p[4] /* 1 / radius, as the scene's records hold it */, vsub(p, V(q.p[0], q.p[1], q.p[2]))); break;
    case PT_XY_RECT: outward = V(0.0f, 0.0f, 1.0f); break;
    case PT_XZ_RECT: outward = V(0.0f, 1.0f, 0.0f); break;
    case PT_YZ_RECT: outward = V(1.0f, 0.0f, 0.0f); break;
    case PT_CONE: {
        v3 center = V(q.p[0], q.p[1], q.p[2]);
        float height = q.p[3], radius = q.p[4];
        outward = vdiv(vsub(p, center), radius);
        v3 pc = vsub(p, center);
        pc.y = 0.0f;
        float normal_y = vlen(pc);
        outward.y = normal_y * radius / height;
        break;
    }
    default: { // cylinder
        v3 center = V(q.p[0], q.p[1], q.p[2]);
        outward = vdiv(vsub(p, center), q.p[4]);
        outward.y = p.y; // cylinder.h:126 stores the absolute y
        break;
    }
    }
    front = vdot(r.d, outward) < 0;
    normal = front ? outward : vneg(outward);
}

// ---------------------------------------------------------------------------------------------
// Materials
// ---------------------------------------------------------------------------------------------
// cuda_functions.h:53-57 with pow(x,5) as the multiplication chain ((x*x)*(x*x))*x
PT_DI float reflectance_schlick(float cosine, float ref_idx)
{
    float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
    r0 = r0 * r0;
    float x = 1.0f - cosine;
    float x2 = x * x;
    return r0 + (1.0f - r0) * ((x2 * x2) * x);
}
