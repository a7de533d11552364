// pt_context.hip -- the host side of libpathtracer_amd.so: the context, the C ABI (include/pt_api.h), scene installation and
// the planning of a render (which pipeline, in how many passes, with what workspace).  The kernels live in the
// pt_kernels_*.hip files and are reached through pt_launch.h.  The only device symbol here is the scratch array of
// pt_render_device_noalloc.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "pt_launch.h"

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
    uint32_t *d_cone_tab = nullptr; // the cones' distinct slopes (cone_table_hit), or null
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
    hipFree(ctx->d_nodes); hipFree(ctx->d_prims); hipFree(ctx->d_prims_list); hipFree(ctx->d_mats); hipFree(ctx->d_wide); hipFree(ctx->d_cone_tab);
    ctx->d_wide = nullptr; ctx->n_wide_alloc = 0; ctx->d_cone_tab = nullptr;
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

// The table cone_table_hit() searches (layout there): the distinct tangens values of the scene's cones, as the list
// records hold them, each with its rank, the largest list index that has it and all the list indices that have it.
// Values that are negative or not finite cancel no ray's |d_xz|^2 and are left out.  Empty when the scene has no such cone.
static std::vector<uint32_t> make_cone_table(const std::vector<PrimRec> &list)
{
    std::vector<std::pair<uint32_t, uint32_t>> v; // (bits of tangens, list index)
    for (size_t i = 0; i < list.size(); ++i) {
        if ((list[i].meta & 7u) != PT_CONE) continue;
        uint32_t bits;
        memcpy(&bits, &list[i].p[5], 4);
        if (bits < 0x7F800000u) v.emplace_back(bits, (uint32_t)i); // +0 .. the largest finite value: their bit patterns order as they do
    }
    std::vector<uint32_t> tab;
    if (v.empty()) return tab;
    std::sort(v.begin(), v.end());
    const uint32_t m = (uint32_t)v.size();
    std::vector<uint32_t> val, last, first;
    for (uint32_t k = 0; k < m; ++k) {
        if (val.empty() || val.back() != v[k].first) { val.push_back(v[k].first); last.push_back(v[k].second); first.push_back(k); }
        else last.back() = v[k].second; // (sorted: the later pair has the larger index)
    }
    first.push_back(m);
    const uint32_t n = (uint32_t)val.size(), lo = val.front(), hi = val.back();
    uint32_t want = 1;
    while (want < 2 * n && want < (1u << 21)) want <<= 1; // (two cells per value, 32 MiB of cells at most: a million slopes)
    uint32_t shift = 7; // a cell spans at least 128 bit patterns, so the candidates of a ray (+-PT_CONE_TAB_SLACK) touch at most two
    while (((hi - lo) >> shift) >= want) ++shift;
    const uint32_t nb = ((hi - lo) >> shift) + 1;
    tab.assign(PT_CONE_TAB_HEADER + 4 * (size_t)nb + 3 * (size_t)n + 1 + m, 0u);
    tab[0] = n; tab[1] = lo; tab[2] = hi; tab[3] = shift; tab[4] = nb; tab[7] = m;
    float flo, fhi;
    memcpy(&flo, &lo, 4); memcpy(&fhi, &hi, 4);
    flo *= 1.0f - 1.0f / 1024.0f; fhi *= 1.0f + 1.0f / 1024.0f; // the bounds of the test that needs no division
    memcpy(&tab[5], &flo, 4); memcpy(&tab[6], &fhi, 4);
    uint32_t *cells = tab.data() + PT_CONE_TAB_HEADER;
    std::fill(cells, cells + 4 * (size_t)nb, PT_CONE_TAB_NONE);
    for (uint32_t i = 0; i < n;) {
        const uint32_t b = (val[i] - lo) >> shift;
        uint32_t j = i;
        while (j < n && ((val[j] - lo) >> shift) == b) ++j;
        if (j - i > 2) {
            cells[4 * b] = PT_CONE_TAB_MORE;
        } else {
            for (uint32_t k = i; k < j; ++k) { cells[4 * b + 2 * (k - i)] = val[k]; cells[4 * b + 2 * (k - i) + 1] = k; }
        }
        i = j;
    }
    uint32_t *out = cells + 4 * (size_t)nb;
    memcpy(out, val.data(), 4 * (size_t)n);
    memcpy(out + n, last.data(), 4 * (size_t)n);
    memcpy(out + 2 * (size_t)n, first.data(), 4 * ((size_t)n + 1));
    for (uint32_t k = 0; k < m; ++k) out[3 * (size_t)n + 1 + k] = v[k].second;
    return tab;
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
#ifdef PT_LDS_WIDE4
    // experiment: scenes of spheres small enough for LDS get four-wide unquantised nodes there (in the `wide` buffer)
    std::vector<float> w32;
    uint32_t w32_depth = 0;
    const bool want_w32 = only_spheres && nb + pb <= 48 * 1024 && b.nodes.size() > 1;
    if (want_w32) w32_depth = pt::make_wide_fp32(b, w32);
    const bool want_wide = want_w32 || (nb + pb > 48 * 1024 && tuned(ctx->tune.wide_nodes, 1) != 0);
    if (want_wide && !want_w32) pt::make_wide(b, wb);
    if (want_w32) wb.nodes.resize((w32.size() * 4 + sizeof(pt::WideNode) - 1) / sizeof(pt::WideNode)); // (sized in WideNodes for the upload below)
#else
    const bool want_wide = nb + pb > 48 * 1024 && tuned(ctx->tune.wide_nodes, 1) != 0;
    if (want_wide) pt::make_wide(b, wb);
#endif

    const std::vector<uint32_t> cone_tab = make_cone_table(list);
    BvhNode *n_nodes = nullptr;
    PrimRec *n_prims = nullptr, *n_list = nullptr;
    float4 *n_mats = nullptr, *n_wide = nullptr;
    uint32_t *n_cone_tab = nullptr;
    auto upload = [&]() -> hipError_t {
        hipError_t e;
        if (!cone_tab.empty()) {
            if ((e = hipMalloc(&n_cone_tab, cone_tab.size() * 4)) != hipSuccess) return e;
            if ((e = hipMemcpy(n_cone_tab, cone_tab.data(), cone_tab.size() * 4, hipMemcpyHostToDevice)) != hipSuccess) return e;
        }
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
#ifdef PT_LDS_WIDE4
        if (want_w32) {
            if ((e = hipMemcpy(n_wide, w32.data(), w32.size() * 4, hipMemcpyHostToDevice)) != hipSuccess) return e;
            return hipSuccess;
        }
#endif
        if (want_wide && (e = hipMemcpy(n_wide, wb.nodes.data(), wb.nodes.size() * sizeof(pt::WideNode), hipMemcpyHostToDevice)) != hipSuccess) return e;
        return hipSuccess;
    };
    const hipError_t ue = upload();
    if (ue != hipSuccess) {
        hipFree(n_nodes); hipFree(n_prims); hipFree(n_list); hipFree(n_mats); hipFree(n_wide); hipFree(n_cone_tab);
        return fail(ctx, PT_ERR_DEVICE, std::string("scene upload: ") + hipGetErrorString(ue));
    }
    free_scene(ctx); // the previous scene's buffers
    ctx->d_nodes = n_nodes; ctx->d_prims = n_prims; ctx->d_prims_list = n_list; ctx->d_mats = n_mats; ctx->d_wide = n_wide;
    ctx->d_cone_tab = n_cone_tab;
    ctx->n_mats_alloc = mats.size();
    ctx->n_wide_alloc = want_wide ? wb.nodes.size() : 0;

    SceneDev &sc = ctx->sc;
    sc.wide = n_wide;
    sc.cone_tab = n_cone_tab;
    sc.cone_hdr = ConeTabHeader{};
    if (!cone_tab.empty()) {
        sc.cone_hdr.n = cone_tab[0]; sc.cone_hdr.lo = cone_tab[1]; sc.cone_hdr.hi = cone_tab[2]; sc.cone_hdr.shift = cone_tab[3]; sc.cone_hdr.nb = cone_tab[4];
        memcpy(&sc.cone_hdr.flo, &cone_tab[5], 4); memcpy(&sc.cone_hdr.fhi, &cone_tab[6], 4);
    }
    // up to three entries per level; 31 is what five workgroups of 256 lanes leave each lane of the CU's LDS
    // (a ray that needs more is answered by the list scan and counted in exact_fallbacks)
    sc.wide_stack_cap = want_wide ? std::min(PT_WIDE_STACK_MAX, 3u * wb.max_depth + 2u) : 0u; // (with the bottom entry; 31 x 256 lanes x 4 B lets five workgroups share a CU's LDS)
    ctx->wide_depth = want_wide ? wb.max_depth : 0u;
    sc.nodes = ctx->d_nodes;
    sc.prims = ctx->d_prims;
    sc.prims_list = ctx->d_prims_list;
    sc.mats = ctx->d_mats;
    sc.n_prims = desc->n_prims;
    sc.n_nodes = (uint32_t)b.nodes.size();
    sc.stack_cap = std::max(8u, b.max_depth + 2); // one entry per level of a descent, and the "done" entry at the bottom
#ifdef PT_LDS_WIDE4
    sc.n_wide4 = want_w32 ? (uint32_t)(w32.size() / 28) : 0u;
    if (want_w32) {
        sc.stack_cap = std::max(8u, 3u * w32_depth + 2u);
        sc.wide_stack_cap = 0;
        ctx->wide_depth = 0;
        ctx->n_wide_alloc = 0;
    }
#endif
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
#ifdef PT_LDS_WIDE4
    if (ctx->kind != PT_KIND_SPHERES || (sc.n_wide4 == 0 && sc.root_ref == 0)) return 0; // the experiment's LDS kernels hold four-wide nodes: spheres only
    size_t b = lds_scene_size(sc.n_wide4, sc.n_prims) + (size_t)sc.stack_cap * PT_ISECT_LDS_BLOCK * 4;
#else
    size_t b = lds_scene_size(sc.n_nodes, sc.n_prims) + (size_t)sc.stack_cap * PT_ISECT_LDS_BLOCK * 4;
#endif
    return b <= 80 * 1024 ? b : 0; // two workgroups per CU must fit in 160 KB
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
    if (mode == PT_TRACE_BVH_GEOMETRIC) {
        SceneDev scene = ctx->sc; // no ray cone: boxes culled by exact geometry
        scene.cone.kappa_max = scene.cone.kappa_min = 0.0f;
        scene.cone.adaptive = 0;
        pt_launch_trace(PT_TRACE_BVH_IN_SCENE, grid, lds, stream, scene, d_o, d_d, n, t_min, t_max, d_out, ctx->d_cnt);
    } else {
        pt_launch_trace(mode, grid, lds, stream, ctx->sc, d_o, d_d, n, t_min, t_max, d_out, ctx->d_cnt);
    }
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
        pt_launch_scatter(grid_for(ctx, n), st, ctx->sc, d_o, d_d, d_t, d_prim, d_sp, d_xi, n, d_out);
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
#ifdef PT_LDS_WIDE4
    const size_t scene_bytes = lds_scene_size(sc.n_wide4, sc.n_prims);
#else
    const size_t scene_bytes = lds_scene_size(sc.n_nodes, sc.n_prims);
#endif
    int block = tuned(tu.block, 0);
    if (block != 256 && block != 512) {
        // LDS-resident scenes: 512 (one scene copy per 8 waves).  Scenes traversed from HBM: 256 measured best without cones;
        // with cones 512, the shape at which the compiler gets k_paths_stream's cone variant into 128 registers (4 waves per
        // SIMD instead of the 3 its 132 registers at 256 allow: 100k mixed objects 2.40 -> 2.85 Grays/s)
        const bool fits_lds = want_lds && !force_stream && scene_bytes + (size_t)sc.stack_cap * 512 * 4 <= 64 * 1024;
        block = fits_lds || ctx->has_cones ? 512 : 256;
        // k_paths on a scene so small that six workgroups of 256 share a CU's LDS: the smaller workgroups balance a large
        // frame better (Cornell box at 1024 x 1024 x 1024 spp: 250.7 -> 241.6 ms; a frame of 1.4 M paths loses 5 %).  Not
        // k_paths_ring, whose chunks go to whichever wave is free: the Cornell box at 256 spp 41.8 ms with 256, 39.1 with 512.
        const bool one_slice_at_a_time = ps.sort || tuned(tu.pipeline, PT_DEFAULT_LDS_PIPELINE) == 0;
        if (fits_lds && one_slice_at_a_time && 6 * (scene_bytes + (size_t)sc.stack_cap * 256 * 4 + 64) <= 160 * 1024 && total_paths >= (16ull << 20)) block = 256;
    }
    ps.block = block;
    const size_t stack_bytes = (size_t)sc.stack_cap * block * 4;
    ps.lds = want_lds && !force_stream && scene_bytes + stack_bytes <= 64 * 1024;
#ifdef PT_LDS_WIDE4
    ps.lds = ps.lds && ctx->kind == PT_KIND_SPHERES && (sc.n_wide4 != 0 || sc.root_ref != 0); // the experiment's LDS kernels: spheres only
#endif
    ps.lds_bytes = stack_bytes + (ps.lds ? scene_bytes : 0);
    // scenes traversed from HBM take the streaming kernel (pt_tuning.pipeline overrides)
    const int pipeline = tuned(tu.pipeline, ps.lds ? PT_DEFAULT_LDS_PIPELINE : 1);
    ps.ring = !force_stream && !ps.sort && ps.lds && pipeline == 2; // (scenes that do not fit LDS: the streaming kernel instead)
    ps.stream = force_stream || (!ps.sort && pipeline != 0 && !ps.ring);
    ps.wide = ps.stream && sc.wide != nullptr;
#ifdef PT_LDS_WIDE4
    if (sc.n_wide4) ps.wide = false; // (`wide` holds the unquantised nodes of the LDS kernels)
#endif
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
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&fit, pt_stream_kernel_for(count, block, ctx->kind, ps.wide), block, ps.stream_stack_bytes) == hipSuccess && fit > 0)
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
    uint32_t spass = 0, pix = 0, cap = 0, ring_pix_bits = 0;
    for (;;) {
        // k_paths_ring keeps its queue filled across slices, so its slices only have to be small enough to balance the end
        // of the frame (16 per resident workgroup) and large enough to be worth a ticket
        // (random spheres at 64 spp: 12.4 ms with slices of 4096 paths, 12.65 with 8192; the Cornell box at 256 spp 38.9 against 39.2)
        const uint32_t balanced = ring ? (uint32_t)std::min<uint64_t>(4096, std::max<uint64_t>(2048, total_paths / (16ull * grid_max)))
                                       : (uint32_t)std::min<uint64_t>(32768, std::max<uint64_t>(4096, total_paths / (4ull * grid_max)));
        const uint32_t target = (uint32_t)tuned(ctx->tune.slice_paths, (int)balanced);
        spass = pr->samples_per_pass ? pr->samples_per_pass : std::max(1u, target / 64u);
        spass = std::min(spass, pr->spp);
        pix = std::max(64u, (target / spass) & ~63u);
        if ((uint64_t)pix * spass * 3 >= (1ull << 32)) return fail(ctx, PT_ERR_LIMIT, "samples_per_pass too large for one slice");
        cap = pix * spass;
        if (ring) { // a slice too large for the ring's path word (pixel and sample side by side in PT_RING_PATH_BITS bits): k_paths
            ring_pix_bits = 6;
            while ((1u << ring_pix_bits) < pix) ++ring_pix_bits;
            uint32_t sample_bits = 0;
            while ((1u << sample_bits) < spass) ++sample_bits;
            if (ring_pix_bits + sample_bits > PT_RING_PATH_BITS || cap > 65536u) { ring = false; continue; } // (and PT_RING radiance buffers of that size per workgroup)
        }
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
    // (two chunks per wave: random spheres 13.6 ms at 1024 .. 4096 against 13.9 at 512 and 14.5 at 16384, where every generation
    // takes a slice)
    // Cornell box at 256 spp (512 threads): 38.9 ms at 1024, 39.1 at 2048, 42.5 at 4096.
    const uint32_t low_water = std::max(1u, (uint32_t)tuned(ctx->tune.ring_low_water, 2 * block));
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
            rg.qcap = qcap; rg.cap = cap; rg.low_water = low_water; rg.pix_bits = ring_pix_bits; rg.n_slices = n_slices; rg.extents = ctx->d_extents; rg.ticket = ctx->d_ticket;
            pt_launch_ring(count, ctx->kind, ctx->d_cnt, scene, stream, block, split, grid, lds_bytes, ps, rg, d_rgb_sum);
        } else {
            pt_launch_paths(count, ctx->kind, ctx->d_cnt, scene, stream, block, lds, sort, split, grid, lds_bytes, ps, sl, d_rgb_sum);
        }
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
    const void *kernel = pt_stream_kernel_for(count, block, ctx->kind, sh.wide);
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
            pt_launch_film(grid_for(ctx, band.npix), stream, ps, rad, d_rgb_sum, nullptr, cnt, band.pix0, band.npix);
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
        pt_launch_raygen(grid, stream, scene, ps, cur, ctx->d_nactive);
        tm.end();
        for (int32_t b = 0; b < pr->max_depth; ++b) {
            tm.begin(1);
            {
                const size_t lds_res = ctx->lds_scene ? lds_scene_bytes(ctx) : 0;
                const uint32_t igrid = lds_res ? (uint32_t)std::max<size_t>(1, std::min((np + PT_ISECT_LDS_BLOCK - 1) / PT_ISECT_LDS_BLOCK, (size_t)ctx->n_cu * 2)) : grid_for(ctx, np);
                pt_launch_intersect(count, lds_res, igrid, (size_t)ctx->sc.stack_cap * PT_BLOCK * sizeof(uint32_t), stream, scene, cur, ctx->d_hits, ctx->d_nactive + b, ctx->d_cnt);
            }
            tm.end();
            ++n_isect;
            tm.begin(2);
            pt_launch_shade(grid, stream, scene, ps, cur, ctx->d_hits, nxt, ctx->d_nactive + b, ctx->d_nactive + b + 1, ctx->d_rad, b);
            tm.end();
            std::swap(cur, nxt);
        }
        tm.begin(3);
        pt_launch_film(grid_for(ctx, n_pixels), stream, ps, ctx->d_rad, d_rgb_sum, ctx->d_nactive, ctx->d_cnt);
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

