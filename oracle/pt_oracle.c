/*
 * pt_oracle.c -- CPU restatement of the reference hot path (see pt_oracle.h for the
 * rules: test infrastructure only; parity unpinned by the reference's own tests).
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC (oracle/Makefile).
 * All arithmetic is fp32 unless the reference itself uses double (vec3.h:334).
 */
#include "pt_oracle.h"
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ vec3.h */
typedef struct { float e[3]; } v3;

static inline v3 V(float a, float b, float c) { v3 r = {{a, b, c}}; return r; }
static inline v3 vload(const float *p) { return V(p[0], p[1], p[2]); }
static inline void vstore(float *p, v3 a) { p[0] = a.e[0]; p[1] = a.e[1]; p[2] = a.e[2]; }
/* vec3.h:73-75 */
static inline v3 vneg(v3 a) { return V(-a.e[0], -a.e[1], -a.e[2]); }
/* vec3.h:217-219 */
static inline v3 vadd(v3 a, v3 b) { return V(a.e[0] + b.e[0], a.e[1] + b.e[1], a.e[2] + b.e[2]); }
/* vec3.h:228-230 */
static inline v3 vsub(v3 a, v3 b) { return V(a.e[0] - b.e[0], a.e[1] - b.e[1], a.e[2] - b.e[2]); }
/* vec3.h:239-241 */
static inline v3 vmul(v3 a, v3 b) { return V(a.e[0] * b.e[0], a.e[1] * b.e[1], a.e[2] * b.e[2]); }
/* vec3.h:250-252 (and :261-263, which forwards to it) */
static inline v3 vscale(float t, v3 a) { return V(t * a.e[0], t * a.e[1], t * a.e[2]); }
/* vec3.h:272-274: reciprocal first, then multiply */
static inline v3 vdiv(v3 a, float t) { return vscale(1 / t, a); }
/* vec3.h:283-287 */
static inline float vdot(v3 a, v3 b) { return a.e[0] * b.e[0] + a.e[1] * b.e[1] + a.e[2] * b.e[2]; }
/* vec3.h:296-300 */
static inline v3 vcross(v3 a, v3 b)
{
    return V(a.e[1] * b.e[2] - a.e[2] * b.e[1], a.e[2] * b.e[0] - a.e[0] * b.e[2], a.e[0] * b.e[1] - a.e[1] * b.e[0]);
}
/* vec3.h:154-156 */
static inline float vlen2(v3 a) { return a.e[0] * a.e[0] + a.e[1] * a.e[1] + a.e[2] * a.e[2]; }
/* vec3.h:145-147 */
static inline float vlen(v3 a) { return sqrtf(vlen2(a)); }
/* vec3.h:308-310 */
static inline v3 vunit(v3 a) { return vdiv(a, vlen(a)); }
/* vec3.h:163-166 */
static inline int vnear_zero(v3 a)
{
    const float s = 1e-8f;
    return (fabsf(a.e[0]) < s) && (fabsf(a.e[1]) < s) && (fabsf(a.e[2]) < s);
}
/* vec3.h:319-321: v - (2*dot(v,n))*n */
static inline v3 vreflect(v3 v, v3 n) { return vsub(v, vscale(2 * vdot(v, n), n)); }
/* vec3.h:331-336: the `1.0 -` at :334 is a double subtraction; fabs stays double; sqrtf rounds to float first */
static inline v3 vrefract(v3 uv, v3 n, float etai_over_etat)
{
    float cos_theta = fminf(vdot(vneg(uv), n), 1.0f);
    v3 r_out_perp = vscale(etai_over_etat, vadd(uv, vscale(cos_theta, n)));
    v3 r_out_parallel = vscale(-sqrtf((float)fabs(1.0 - (double)vlen2(r_out_perp))), n);
    return vadd(r_out_perp, r_out_parallel);
}
/* ray.h:47-49 */
static inline v3 ray_at(v3 o, v3 d, float t) { return vadd(o, vscale(t, d)); }

/* ---------------------------------------------------------------- RNG (not the reference's: see header) */
static const uint32_t PHILOX_M0 = 0xD2511F53u, PHILOX_M1 = 0xCD9E8D57u;
static const uint32_t PHILOX_W0 = 0x9E3779B9u, PHILOX_W1 = 0xBB67AE85u;

void orc_philox4x32(const uint32_t ctr[4], const uint32_t key[2], int rounds, uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < rounds; ++r) {
        uint64_t p0 = (uint64_t)PHILOX_M0 * c0, p1 = (uint64_t)PHILOX_M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += PHILOX_W0; k1 += PHILOX_W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* (0,1] in steps of 2^-24, like curand_uniform's range (main.cu:90) */
float orc_u32_to_uniform(uint32_t x) { return (float)((x >> 8) + 1u) * (1.0f / 16777216.0f); }

/* The stream of one (pixel, sample, bounce): the 32-bit words of Philox4x32-7 blocks 0, 1, 2, ... taken in
 * order, one word per uniform drawn (curand_uniform's role, main.cu:90-91, cuda_functions.h:17,68,
 * material.h:235).  `word` counts the words drawn since the bounce began. */
typedef struct { uint32_t pixel, sample, bounce, word, buf[4], key[2]; } rng_t;

static void rng_start(rng_t *g, uint32_t bounce) { g->bounce = bounce; g->word = 0; }

static float rng_uniform(rng_t *g)
{
    if ((g->word & 3u) == 0) {
        uint32_t ctr[4] = {g->word >> 2, g->bounce, g->sample, g->pixel};
        orc_philox4x32(ctr, g->key, ORC_PHILOX_ROUNDS, g->buf);
    }
    return orc_u32_to_uniform(g->buf[g->word++ & 3u]);
}

#define ORC_MAX_TRIES 64 /* rejection loops are unbounded in the reference; both sides stop here */

/* cuda_functions.h:14-20: three uniforms per try, in stream order */
static v3 random_in_unit_sphere(rng_t *g)
{
    v3 p;
    int tries = 0;
    do {
        float u0 = rng_uniform(g), u1 = rng_uniform(g), u2 = rng_uniform(g);
        p = vsub(vscale(2.0f, V(u0, u1, u2)), V(1, 1, 1));
    } while (vlen2(p) >= 1.0f && ++tries < ORC_MAX_TRIES);
    return p;
}
/* cuda_functions.h:65-71: two uniforms per try */
static v3 random_in_unit_disk(rng_t *g)
{
    v3 p;
    int tries = 0;
    do {
        float u0 = rng_uniform(g), u1 = rng_uniform(g);
        p = vsub(vscale(2.0f, V(u0, u1, 0)), V(1, 1, 0));
    } while (vlen2(p) >= 1.0f && ++tries < ORC_MAX_TRIES);
    return p;
}

/* ------------------------------------------------------------- camera.h */
#define ORC_PI 3.1415926535897932385f /* cuda_functions.h:6 */

void orc_camera_make(const float lookfrom_[3], const float lookat_[3], const float vup_[3], float vfov, float aspect_ratio,
                     float aperture, float focus_dist, orc_scene *out)
{
    /* camera.h:47-61 */
    v3 lookfrom = vload(lookfrom_), lookat = vload(lookat_), vup = vload(vup_);
    const float theta = vfov * ORC_PI / 180.0f;
    const float h = tanf(theta / 2.0f);
    const float viewport_height = 2.0f * h;
    const float viewport_width = aspect_ratio * viewport_height;
    v3 w = vunit(vsub(lookfrom, lookat));
    v3 u = vunit(vcross(vup, w));
    v3 v = vcross(w, u);
    v3 origin = lookfrom;
    v3 horizontal = vscale(focus_dist * viewport_width, u);
    v3 vertical = vscale(focus_dist * viewport_height, v);
    v3 llc = vsub(vsub(vsub(origin, vdiv(horizontal, 2)), vdiv(vertical, 2)), vscale(focus_dist, w));
    vstore(out->cam_origin, origin);
    vstore(out->cam_lower_left, llc);
    vstore(out->cam_horizontal, horizontal);
    vstore(out->cam_vertical, vertical);
    vstore(out->cam_u, u);
    vstore(out->cam_v, v);
    vstore(out->cam_w, w);
    out->cam_lens_radius = aperture / 2;
}

/* camera.h:88-96 */
void orc_get_ray(const orc_scene *s, float sc, float tc, const float disk_pt[2], float out_o[3], float out_d[3])
{
    v3 rd = vscale(s->cam_lens_radius, V(disk_pt[0], disk_pt[1], 0));
    v3 offset = vadd(vscale(rd.e[0], vload(s->cam_u)), vscale(rd.e[1], vload(s->cam_v)));
    v3 origin = vload(s->cam_origin);
    v3 o = vadd(origin, offset);
    v3 d = vsub(vsub(vadd(vadd(vload(s->cam_lower_left), vscale(sc, vload(s->cam_horizontal))),
                          vscale(tc, vload(s->cam_vertical))), origin), offset);
    vstore(out_o, o);
    vstore(out_d, d);
}

/* ------------------------------------------------------------ hittables */
/* hittable.h:29-32 */
static inline void set_face_normal(orc_hit *rec, v3 d, v3 outward)
{
    rec->front = vdot(d, outward) < 0;
    vstore(rec->n, rec->front ? outward : vneg(outward));
}

/* sphere.h:85-110 */
static int sphere_hit(const float *q, v3 o, v3 d, float t_min, float t_max, orc_hit *rec)
{
    v3 center = vload(q);
    float radius = q[3];
    v3 oc = vsub(o, center);
    float a = vlen2(d);
    float half_b = vdot(oc, d);
    float c = vlen2(oc) - radius * radius;
    float discriminant = half_b * half_b - a * c;
    if (discriminant < 0) return 0;
    float sqrtd = sqrtf(discriminant);
    float root = (-half_b - sqrtd) / a;
    if (root < t_min || t_max < root) {
        root = (-half_b + sqrtd) / a;
        if (root < t_min || t_max < root) return 0;
    }
    rec->t = root;
    v3 p = ray_at(o, d, root);
    vstore(rec->p, p);
    set_face_normal(rec, d, vdiv(vsub(p, center), radius));
    return 1;
}

/* aarect.h:348-362 (axis 2), :364-378 (axis 1), :380-394 (axis 0): the plane axis is `ax`,
 * the bounded axes are (a0, a1) = (x,y), (x,z), (y,z). */
static int rect_hit(const float *q, int ax, int a0, int a1, v3 o, v3 d, float t_min, float t_max, orc_hit *rec)
{
    float lo0 = q[0], hi0 = q[1], lo1 = q[2], hi1 = q[3], k = q[4];
    float t = (k - o.e[ax]) / d.e[ax];
    if (t < t_min || t > t_max) return 0;
    float c0 = o.e[a0] + t * d.e[a0];
    float c1 = o.e[a1] + t * d.e[a1];
    if (c0 < lo0 || c0 > hi0 || c1 < lo1 || c1 > hi1) return 0;
    rec->t = t;
    v3 outward = V(0, 0, 0);
    outward.e[ax] = 1;
    set_face_normal(rec, d, outward);
    vstore(rec->p, ray_at(o, d, t));
    return 1;
}

/* cone.h:101-139 */
static int cone_hit(const float *q, v3 o, v3 d, float t_min, float t_max, orc_hit *rec)
{
    v3 center = vload(q);
    float height = q[3], radius = q[4], tangens = q[5];
    v3 oc = vsub(o, center);
    oc.e[1] = 0;
    v3 dxz = d;
    dxz.e[1] = 0;
    float D = height - o.e[1] + center.e[1];
    float a = vlen2(dxz) - tangens * d.e[1] * d.e[1];
    float half_b = vdot(oc, dxz) + tangens * d.e[1] * D;
    float c = vlen2(oc) - tangens * D * D;
    float discriminant = half_b * half_b - a * c;
    if (discriminant < 0) return 0;
    float sqrtd = sqrtf(discriminant);
    float root = (-half_b - sqrtd) / a;
    if (root < t_min || t_max < root) {
        root = (-half_b + sqrtd) / a;
        if (root < t_min || t_max < root) return 0;
    }
    rec->t = root;
    v3 p = ray_at(o, d, root);
    vstore(rec->p, p);
    if (p.e[1] < center.e[1] || p.e[1] > center.e[1] + height) return 0;
    v3 outward = vdiv(vsub(p, center), radius);
    v3 pc = vsub(p, center);
    pc.e[1] = 0;
    float normal_y = vlen(pc);
    outward.e[1] = normal_y * radius / height;
    set_face_normal(rec, d, outward);
    /* A ray nearly parallel to the cone's side cancels the leading coefficient; the root is then
     * rounding noise divided by a tiny a and can lie anywhere along the ray.  Such accepted hits are
     * outside the parity contract (no bound on where they land), like NaN roots. */
    rec->nan_flag = fabsf(a) < 0.015625f * (vlen2(dxz) + tangens * d.e[1] * d.e[1]) ? 2u : 0u;
    return 1;
}

/* cylinder.h:96-130 */
static int cylinder_hit(const float *q, v3 o, v3 d, float t_min, float t_max, orc_hit *rec)
{
    v3 center = vload(q);
    float height = q[3], radius = q[4];
    v3 oc = vsub(o, center);
    oc.e[1] = 0;
    v3 dxz = d;
    dxz.e[1] = 0;
    float a = vlen2(dxz);
    float half_b = vdot(oc, dxz);
    float c = vlen2(oc) - (radius * radius);
    float discriminant = half_b * half_b - a * c;
    if (discriminant < 0) return 0;
    float sqrtd = sqrtf(discriminant);
    float root = (-half_b - sqrtd) / a;
    if (root < t_min || t_max < root) {
        root = (-half_b + sqrtd) / a;
        if (root < t_min || t_max < root) return 0;
    }
    rec->t = root;
    v3 p = ray_at(o, d, root);
    vstore(rec->p, p);
    if (p.e[1] < center.e[1] || p.e[1] > center.e[1] + height) return 0;
    v3 outward = vdiv(vsub(p, center), radius);
    outward.e[1] = p.e[1]; /* cylinder.h:126 assigns the absolute y */
    set_face_normal(rec, d, outward);
    return 1;
}

static int hit_prim(const orc_scene *s, uint32_t i, v3 o, v3 d, float t_min, float t_max, orc_hit *rec)
{
    const float *q = s->prim_param + (size_t)i * 6;
    switch (s->prim_type[i]) {
    case ORC_SPHERE: return sphere_hit(q, o, d, t_min, t_max, rec);
    case ORC_XY_RECT: return rect_hit(q, 2, 0, 1, o, d, t_min, t_max, rec);
    case ORC_XZ_RECT: return rect_hit(q, 1, 0, 2, o, d, t_min, t_max, rec);
    case ORC_YZ_RECT: return rect_hit(q, 0, 1, 2, o, d, t_min, t_max, rec);
    case ORC_CONE: return cone_hit(q, o, d, t_min, t_max, rec);
    case ORC_CYLINDER: return cylinder_hit(q, o, d, t_min, t_max, rec);
    default: return 0;
    }
}

int orc_hit_prim(const orc_scene *s, uint32_t i, const float o[3], const float d[3], float t_min, float t_max,
                 orc_hit *rec)
{
    memset(rec, 0, sizeof *rec);
    rec->prim = 0xFFFFFFFFu;
    int h = hit_prim(s, i, vload(o), vload(d), t_min, t_max, rec);
    if (h) rec->prim = i;
    return h;
}

/* hittable_list.h:87-101 */
static int hit_list(const orc_scene *s, v3 o, v3 d, float t_min, float t_max, orc_hit *rec)
{
    orc_hit temp;
    int hit_anything = 0;
    float closest_so_far = t_max;
    uint32_t nan_flag = 0;
    memset(&temp, 0, sizeof temp);
    for (uint32_t i = 0; i < s->n_prims; ++i) {
        temp.nan_flag = 0;
        if (hit_prim(s, i, o, d, t_min, closest_so_far, &temp)) {
            hit_anything = 1;
            closest_so_far = temp.t;
            if (temp.t != temp.t) nan_flag |= 1;
            nan_flag |= temp.nan_flag;
            *rec = temp;
            rec->prim = i;
        }
    }
    rec->nan_flag = nan_flag;
    return hit_anything;
}

int orc_hit_list(const orc_scene *s, const float o[3], const float d[3], float t_min, float t_max, orc_hit *rec)
{
    memset(rec, 0, sizeof *rec);
    rec->prim = 0xFFFFFFFFu;
    return hit_list(s, vload(o), vload(d), t_min, t_max, rec);
}

/* ------------------------------------------------- tree traversal on the CPU (baseline timing only) */
/* NOT part of the reference (it scans its list, hittable_list.h:87-101) and not a parity checker: a plain
 * stack traversal of the tree the GPU library builds (64-byte nodes: per axis lo of child 0, lo of child 1,
 * hi of child 0, hi of child 1; two child references, negative = leaf ~((first << 3) | (count - 1)) into
 * `order`), so that a CPU figure with the same O(log N) algorithm stands beside the linear-scan one.  Boxes
 * are culled by exact geometry, slightly padded: the few hits the reference's fp32 arithmetic reports outside
 * the exact surfaces (which the GPU traversal keeps) may be lost here, and a cone's list-order dependence is
 * ignored.  Equal t goes to the later list index. */
typedef struct orc_accel {
    const float *nodes; /* 16 words per node */
    const uint32_t *order;
    uint32_t n_nodes;
} orc_accel;
static const orc_accel *g_accel; /* set by orc_render_bvh for the duration of its call */

static int hit_tree(const orc_scene *s, const orc_accel *acc, v3 o, v3 d, float t_min, float t_max, orc_hit *rec)
{
    int32_t stack[128];
    int sp = 0, have = 0;
    float best = t_max;
    uint32_t nan_flag = 0;
    orc_hit temp;
    float inv[3];
    for (int a = 0; a < 3; ++a) inv[a] = 1.0f / d.e[a];
    memset(&temp, 0, sizeof temp);
    int32_t cur = 0;
    for (;;) {
        if (cur >= 0) {
            const float *nd = acc->nodes + (size_t)cur * 16;
            const int32_t *ch = (const int32_t *)(nd + 12);
            int hit[2];
            float tn[2];
            for (int k = 0; k < 2; ++k) {
                float t0 = t_min, t1 = best;
                for (int a = 0; a < 3; ++a) {
                    float x0 = (nd[4 * a + k] - o.e[a]) * inv[a], x1 = (nd[4 * a + 2 + k] - o.e[a]) * inv[a];
                    if (x0 > x1) { float t = x0; x0 = x1; x1 = t; }
                    x0 -= fabsf(x0) * 1e-5f;
                    x1 += fabsf(x1) * 1e-5f;
                    if (x0 > t0) t0 = x0;
                    if (x1 < t1) t1 = x1;
                }
                hit[k] = t0 <= t1;
                tn[k] = t0;
            }
            if (hit[0] && hit[1]) {
                int nearc = tn[1] < tn[0];
                if (sp < 128) stack[sp++] = ch[!nearc];
                cur = ch[nearc];
                continue;
            }
            if (hit[0]) { cur = ch[0]; continue; }
            if (hit[1]) { cur = ch[1]; continue; }
        } else {
            uint32_t ref = (uint32_t)~cur, first = ref >> 3, count = (ref & 7u) + 1u;
            for (uint32_t k = 0; k < count; ++k) {
                uint32_t i = acc->order[first + k];
                temp.nan_flag = 0;
                if (hit_prim(s, i, o, d, t_min, best, &temp) && (!have || temp.t < best || i > rec->prim)) {
                    have = 1;
                    best = temp.t;
                    if (temp.t != temp.t) nan_flag |= 1;
                    nan_flag |= temp.nan_flag;
                    *rec = temp;
                    rec->prim = i;
                }
            }
        }
        if (sp == 0) break;
        cur = stack[--sp];
    }
    rec->nan_flag = nan_flag;
    return have;
}

/* Work per ray on a FIXED tree, for the roofline's algorithmic bytes and flops: the same plain stack traversal as
 * hit_tree (near child first, boxes culled by exact geometry, nearest hit shrinking the ray), counting the box tests
 * (two per inner node visited) and the object tests it performs on every ray of a batch.  Not the reference (it has
 * no tree) and not a checker; the counts do not depend on how any GPU kernel traverses. */
void orc_count_tree_work(const orc_scene *s, const float *nodes, uint32_t n_nodes, const uint32_t *order, const float *o, const float *d,
                         size_t n, float t_min, float t_max, uint64_t *box_tests, uint64_t *prim_tests)
{
    uint64_t nb = 0, np = 0;
    (void)n_nodes;
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : nb, np)
    for (long long r = 0; r < (long long)n; ++r) {
        v3 ro = vload(o + 3 * r), rd = vload(d + 3 * r);
        int32_t stack[128];
        int sp = 0, have = 0;
        float best = t_max, inv[3];
        uint32_t best_id = 0;
        orc_hit temp;
        memset(&temp, 0, sizeof temp);
        for (int a = 0; a < 3; ++a) inv[a] = 1.0f / rd.e[a];
        int32_t cur = 0;
        for (;;) {
            if (cur >= 0) {
                const float *nd = nodes + (size_t)cur * 16;
                const int32_t *ch = (const int32_t *)(nd + 12);
                int hit[2];
                float tn[2];
                nb += 2;
                for (int k = 0; k < 2; ++k) {
                    float t0 = t_min, t1 = best;
                    for (int a = 0; a < 3; ++a) {
                        float x0 = (nd[4 * a + k] - ro.e[a]) * inv[a], x1 = (nd[4 * a + 2 + k] - ro.e[a]) * inv[a];
                        if (x0 > x1) { float t = x0; x0 = x1; x1 = t; }
                        x0 -= fabsf(x0) * 1e-5f;
                        x1 += fabsf(x1) * 1e-5f;
                        if (x0 > t0) t0 = x0;
                        if (x1 < t1) t1 = x1;
                    }
                    hit[k] = t0 <= t1;
                    tn[k] = t0;
                }
                if (hit[0] && hit[1]) {
                    int nearc = tn[1] < tn[0];
                    if (sp < 128) stack[sp++] = ch[!nearc];
                    cur = ch[nearc];
                    continue;
                }
                if (hit[0]) { cur = ch[0]; continue; }
                if (hit[1]) { cur = ch[1]; continue; }
            } else {
                uint32_t ref = (uint32_t)~cur, first = ref >> 3, count = (ref & 7u) + 1u;
                for (uint32_t k = 0; k < count; ++k) {
                    uint32_t i = order[first + k];
                    np += 1;
                    if (hit_prim(s, i, ro, rd, t_min, best, &temp) && (!have || temp.t < best || i > best_id)) {
                        have = 1;
                        best = temp.t;
                        best_id = i;
                    }
                }
            }
            if (sp == 0) break;
            cur = stack[--sp];
        }
    }
    *box_tests = nb;
    *prim_tests = np;
}

/* what ray_color queries: the reference's list scan, or the tree while orc_render_bvh runs */
static int hit_world(const orc_scene *s, v3 o, v3 d, float t_min, float t_max, orc_hit *rec)
{
    const orc_accel *acc = g_accel;
    return acc ? hit_tree(s, acc, o, d, t_min, t_max, rec) : hit_list(s, o, d, t_min, t_max, rec);
}

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void orc_trace_batch(const orc_scene *s, const float *o, const float *d, size_t n, float t_min, float t_max,
                     orc_hit *out, int threads)
{
    if (threads <= 0) threads = orc_num_threads();
#pragma omp parallel for schedule(dynamic, 256) num_threads(threads)
    for (long long i = 0; i < (long long)n; ++i) {
        orc_hit rec;
        memset(&rec, 0, sizeof rec);
        rec.prim = 0xFFFFFFFFu;
        if (!hit_list(s, vload(o + 3 * i), vload(d + 3 * i), t_min, t_max, &rec)) {
            uint32_t nf = rec.nan_flag;
            memset(&rec, 0, sizeof rec);
            rec.prim = 0xFFFFFFFFu;
            rec.nan_flag = nf;
        }
        out[i] = rec;
    }
}

/* ------------------------------------------------------------ materials */
/* cuda_functions.h:53-57.  pow(1-cosine, 5) is evaluated as ((x*x)*(x*x))*x: CUDA's pow(float,int)
 * overload is an integer-power multiplication chain, and the GPU path uses the same chain. */
static inline float reflectance_schlick(float cosine, float ref_idx)
{
    float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
    r0 = r0 * r0;
    float x = 1.0f - cosine;
    float x2 = x * x;
    return r0 + (1.0f - r0) * ((x2 * x2) * x);
}

static int scatter_impl(const orc_scene *s, uint32_t mat, v3 ro, v3 rd, const orc_hit *rec, rng_t *g,
                        const float *sphere_pt, const float *xi_in, v3 *atten, v3 *so, v3 *sd)
{
    const float *m = s->mat_param + (size_t)mat * 4;
    v3 normal = vload(rec->n), p = vload(rec->p);
    (void)ro;
    switch (s->mat_type[mat]) {
    case ORC_LAMBERTIAN: { /* material.h:90-99 */
        v3 rs = sphere_pt ? vload(sphere_pt) : random_in_unit_sphere(g);
        v3 dir = vadd(normal, vunit(rs));
        if (vnear_zero(dir)) dir = normal;
        *so = p;
        *sd = dir;
        *atten = V(m[0], m[1], m[2]);
        return 1;
    }
    case ORC_METAL: { /* material.h:155-162; fuzz already clamped to <= 1 (material.h:150) */
        v3 reflected = vreflect(vunit(rd), normal);
        v3 rs = sphere_pt ? vload(sphere_pt) : random_in_unit_sphere(g);
        *so = p;
        *sd = vadd(reflected, vscale(m[3], rs));
        *atten = V(m[0], m[1], m[2]);
        return vdot(*sd, normal) > 0.0f;
    }
    case ORC_DIELECTRIC: { /* material.h:223-242 */
        float ir = m[0];
        *atten = V(1.0f, 1.0f, 1.0f);
        float refraction_ratio = rec->front ? (1.0f / ir) : ir;
        v3 unit_direction = vunit(rd);
        float cos_theta = fminf(vdot(vneg(unit_direction), normal), 1.0f);
        float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
        int cannot_refract = refraction_ratio * sin_theta > 1.0f;
        v3 direction;
        int reflect_it = cannot_refract;
        if (!reflect_it) { /* the uniform is drawn only when the first operand of || is false (material.h:235) */
            float xi;
            if (xi_in) xi = *xi_in;
            else xi = rng_uniform(g);
            reflect_it = reflectance_schlick(cos_theta, refraction_ratio) > xi;
        }
        if (reflect_it) direction = vreflect(unit_direction, normal);
        else direction = vrefract(unit_direction, normal, refraction_ratio);
        *so = p;
        *sd = direction;
        return 1;
    }
    default: /* diffuse_light, material.h:295-299 */
        *atten = V(0, 0, 0);
        *so = p;
        *sd = V(0, 0, 0);
        return 0;
    }
}

int orc_scatter(const orc_scene *s, uint32_t mat, const float ray_o[3], const float ray_d[3], const orc_hit *rec,
                const float sphere_pt[3], float xi, float atten[3], float out_o[3], float out_d[3])
{
    v3 a, so, sd;
    int r = scatter_impl(s, mat, vload(ray_o), vload(ray_d), rec, NULL, sphere_pt, &xi, &a, &so, &sd);
    vstore(atten, a);
    vstore(out_o, so);
    vstore(out_d, sd);
    return r;
}

/* material.h:45-47 and :302-304 */
static inline v3 emitted(const orc_scene *s, uint32_t mat)
{
    if (s->mat_type[mat] == ORC_DIFFUSE_LIGHT) {
        const float *m = s->mat_param + (size_t)mat * 4;
        return V(m[0], m[1], m[2]);
    }
    return V(0.0f, 0.0f, 0.0f);
}

/* ---------------------------------------------------------- integrator */
/* main.cu:46-65; `bounce` numbers the random stream of the scatter at this depth (1, 2, ...) */
static v3 ray_color(const orc_scene *s, v3 o, v3 d, int depth, rng_t *g, uint32_t bounce, uint64_t *rays)
{
    orc_hit rec;
    if (depth <= 0) return V(0, 0, 0);
    ++*rays;
    memset(&rec, 0, sizeof rec);
    int h = hit_world(s, o, d, 0.001f, FLT_MAX /* commons.h:11 */, &rec);
    static int print_path = -1; /* diagnostic (ORC_PRINT_PATH): the path's rays and hits, one line per bounce; read once */
    if (print_path < 0) print_path = getenv("ORC_PRINT_PATH") != NULL;
    if (print_path)
        fprintf(stderr, "[orc] bounce %u o %.9g %.9g %.9g d %.9g %.9g %.9g hit %d prim %u type %d t %.9g n %.9g %.9g %.9g front %d\n", bounce - 1,
                o.e[0], o.e[1], o.e[2], d.e[0], d.e[1], d.e[2], h, h ? rec.prim : 0u, h ? (int)s->prim_type[rec.prim] : -1, rec.t, rec.n[0],
                rec.n[1], rec.n[2], (int)rec.front);
    if (!h) return vload(s->background);
    uint32_t mat = s->prim_material[rec.prim];
    v3 attenuation, so, sd;
    v3 emit = emitted(s, mat);
    rng_start(g, bounce);
    if (!scatter_impl(s, mat, o, d, &rec, g, NULL, NULL, &attenuation, &so, &sd)) return emit;
    return vadd(emit, vmul(attenuation, ray_color(s, so, sd, depth - 1, g, bounce + 1, rays)));
}

/* main.cu:86-94 for one pixel */
static v3 pixel_sum(const orc_scene *s, int i, int j, int width, int height, uint32_t sample0, uint32_t n_samples,
                    int depth, uint64_t seed, uint64_t *rays)
{
    v3 col = V(0.0f, 0.0f, 0.0f);
    rng_t g;
    g.pixel = (uint32_t)(j * width + i);
    g.key[0] = (uint32_t)seed;
    g.key[1] = (uint32_t)(seed >> 32);
    for (uint32_t k = 0; k < n_samples; ++k) {
        float oo[3], dd[3];
        g.sample = sample0 + k;
        rng_start(&g, 0);
        float u = ((float)i + rng_uniform(&g)) / (float)width;   /* main.cu:90 */
        float v = ((float)j + rng_uniform(&g)) / (float)height;  /* main.cu:91 */
        v3 disk = random_in_unit_disk(&g);
        orc_get_ray(s, u, v, disk.e, oo, dd);
        col = vadd(col, ray_color(s, vload(oo), vload(dd), depth, &g, 1, rays));
    }
    return col;
}

uint64_t orc_render_region(const orc_scene *s, int width, int height, int x0, int y0, int w, int h, uint32_t sample0,
                           uint32_t n_samples, int depth, uint64_t seed, float *rgb_sum, int threads)
{
    uint64_t total = 0;
    if (threads <= 0) threads = orc_num_threads();
    /* pixels are independent: dealt out in runs short enough that a crop of a few pixels still uses every thread */
    const long n_px = (long)w * h;
    long run = n_px / ((long)threads * 16);
    if (run < 1) run = 1;
    if (run > w) run = w;
#pragma omp parallel for schedule(dynamic, run) reduction(+ : total) num_threads(threads)
    for (long px = 0; px < n_px; ++px) {
        const int jj = (int)(px / w), ii = (int)(px - (long)jj * w);
        uint64_t rays = 0;
        v3 c = pixel_sum(s, x0 + ii, y0 + jj, width, height, sample0, n_samples, depth, seed, &rays);
        vstore(rgb_sum + (size_t)px * 3, c);
        total += rays;
    }
    return total;
}

uint64_t orc_render(const orc_scene *s, int width, int height, uint32_t sample0, uint32_t n_samples, int depth,
                    uint64_t seed, float *rgb_sum, int threads)
{
    return orc_render_region(s, width, height, 0, 0, width, height, sample0, n_samples, depth, seed, rgb_sum, threads);
}

/* orc_render with nearest-hit queries answered through a tree (see hit_tree): a timing baseline, not the reference. */
uint64_t orc_render_bvh(const orc_scene *s, const float *nodes, uint32_t n_nodes, const uint32_t *order, int width, int height,
                        uint32_t sample0, uint32_t n_samples, int depth, uint64_t seed, float *rgb_sum, int threads)
{
    orc_accel acc = {nodes, order, n_nodes};
    g_accel = &acc;
    uint64_t rays = orc_render_region(s, width, height, 0, 0, width, height, sample0, n_samples, depth, seed, rgb_sum, threads);
    g_accel = NULL;
    return rays;
}

/* commons.h:58-62 */
static inline float clampf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }

void orc_finalize(const float *rgb_sum, int width, int height, uint32_t spp, float *rgb_gamma, uint8_t *rgb8)
{
    /* main.cu:95-98: col /= float(spp) multiplies by the reciprocal (vec3.h:136-138) */
    float inv = 1 / (float)spp;
    size_t w = 0;
    for (int j = height - 1; j >= 0; --j) { /* main.cu:194: rows written top-down from j = H-1 */
        for (int i = 0; i < width; ++i) {
            size_t px = (size_t)j * width + i;
            for (int c = 0; c < 3; ++c) {
                float g = sqrtf(rgb_sum[px * 3 + c] * inv);
                if (rgb_gamma) rgb_gamma[px * 3 + c] = g;
                if (rgb8) rgb8[w] = (uint8_t)(int)(256 * clampf(g, 0.0f, 0.999f)); /* main.cu:198-200 */
                ++w;
            }
        }
    }
}
