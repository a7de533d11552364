/*
 * pt_api.h -- C ABI of libpathtracer_amd.so, the MI355X (gfx950) wavefront path tracer.
 *
 * The reference program has no FFI: its host code reaches the GPU through device-side `new`
 * behind <<<1,1>>> kernels and one `render` launch.  Each entry point below names the reference
 * interface (file:line, relative to the reference tree) whose job it takes over.
 *
 * Conventions: plain C types, opaque context handle, every call returns a pt_status and never
 * exits the process (the reference's checkCudaErrors, commons.h:68-77, prints and exit(99)s).
 * The caller owns host buffers; the library owns device memory except where a parameter is
 * documented as a caller-provided device pointer.  One context serves one GPU and is used from
 * one host thread at a time.  Calls that enqueue work on a stream and return (pt_render_device,
 * pt_trace_rays_device) share the context's workspace: the library orders each such call behind the
 * previous one on the same context (an event wait), whatever streams they use.  Multi-GPU jobs run one
 * context per GPU and combine the linear accumulation buffers with an all-reduce: pt_group_* below does
 * so in one process over RCCL; one process per GPU with pt_render_device works as well.
 */
#ifndef PT_API_H
#define PT_API_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum pt_status {
    PT_OK = 0,
    PT_ERR_INVALID = 1,   /* bad argument or scene description */
    PT_ERR_NO_SCENE = 2,  /* render / trace before pt_scene_upload */
    PT_ERR_DEVICE = 3,    /* a HIP call failed; pt_last_error() has the text */
    PT_ERR_LIMIT = 4      /* a documented size limit was exceeded */
} pt_status;

/* Object tags: the reference's hittable subclasses (sphere.h:11, aarect.h:14,115,207, cone.h:11,
 * cylinder.h:11), dispatched by tag instead of by vtable. */
enum { PT_SPHERE = 0, PT_XY_RECT = 1, PT_XZ_RECT = 2, PT_YZ_RECT = 3, PT_CONE = 4, PT_CYLINDER = 5, PT_N_PRIM_TYPES = 6 };
/* Material tags: material.h:80,141,212,284. */
enum { PT_LAMBERTIAN = 0, PT_METAL = 1, PT_DIELECTRIC = 2, PT_DIFFUSE_LIGHT = 3, PT_N_MAT_TYPES = 4 };

/* camera.h:125-145: the seven basis vectors and the lens radius of a constructed camera. */
typedef struct pt_camera {
    float origin[3], lower_left_corner[3], horizontal[3], vertical[3], u[3], v[3], w[3];
    float lens_radius;
} pt_camera;

/* The flattened world: what scene.h:37-45 builds with one cudaMalloc + one <<<1,1>>> kernel per
 * object.  Borrowed host pointers, copied before pt_scene_upload returns.
 *   prim_param, six floats per object, in list order (hittable_list.h:76):
 *     sphere   cx cy cz radius - -                 (sphere.h:78-82)
 *     xy_rect  x0 x1 y0 y1 k -   xz_rect x0 x1 z0 z1 k -   yz_rect y0 y1 z0 z1 k -   (aarect.h:108,200,291)
 *     cone     cx cy cz height radius tangens      (cone.h:86-98; tangens = (r/h)*(r/h), cone.h:30)
 *     cylinder cx cy cz height radius -            (cylinder.h:85-93)
 *   mat_param, four floats per material:
 *     lambertian r g b - ; metal r g b fuzz ; dielectric ir - - - ; diffuse_light r g b -   (material.h:134,201-205,277,339)
 */
typedef struct pt_scene_desc {
    uint32_t n_prims;
    const uint8_t *prim_type;
    const float *prim_param;
    const uint32_t *prim_material;
    uint32_t n_materials;
    const uint8_t *mat_type;
    const float *mat_param;
    pt_camera camera;
    float background[3]; /* scene.h:311 */
} pt_scene_desc;

/* hit_record (hittable.h:15-33) plus the winning list index. */
typedef struct pt_hit {
    float t;
    uint32_t prim; /* index in list order; 0xFFFFFFFF on a miss */
    float n[3];    /* face-forwarded normal */
    uint32_t front;
    float p[3];
    uint32_t flags; /* bit 1: list-order dependent, replayed over the ray's candidate objects; bit 0: resolved by the full list scan */
} pt_hit;

enum { PT_TRACE_BVH = 0, PT_TRACE_LINEAR = 1, PT_TRACE_BVH_IN_SCENE = 2, PT_TRACE_BVH_GEOMETRIC = 3 };

typedef struct pt_render_params {
    int32_t width, height;
    uint32_t spp;           /* samples this call renders per pixel */
    uint32_t sample_offset; /* absolute index of the first of them: shards of one image use disjoint ranges */
    int32_t max_depth;      /* ray_color's depth argument: 5 in the reference (main.cu:93) */
    uint64_t seed;          /* the reference seeds curand with 1984 + pixel (main.cu:34) */
    uint32_t samples_per_pass; /* 0 = let the library choose how many samples of every pixel are in flight */
    uint32_t flags;            /* PT_RENDER_* */
} pt_render_params;

enum {
    PT_RENDER_STAGE_TIMES = 1u, /* time every kernel with events (slower; fills pt_stats.ms_*) */
    PT_RENDER_NO_SORT = 2u,     /* never use the block-local material sort (it is opt-in: pt_tuning.sort_by_material) */
    PT_RENDER_COUNT_TESTS = 4u, /* count box / object tests per ray (slower; fills pt_stats.box_tests / prim_tests) */
    PT_RENDER_NO_LDS_SCENE = 8u, /* traverse from global memory even when the scene would fit in LDS */
    PT_RENDER_SPLIT_KERNELS = 16u, /* one launch per stage and bounce over global queues instead of the persistent-workgroup kernel */
    /* Cull tree boxes by the objects' exact geometry only (no ray cone, see pt_scene_ray_cone).  The default cone keeps
     * the hits the reference's fp32 quadratics (sphere.h:85-101, cylinder.h:96-123, cone.h:101-129) report outside the exact
     * surfaces far from the ray origin, at up to twice the traversal cost on scenes hundreds of units across.  With this
     * flag a ray may lose such a hit: every lost or changed hit is one whose discriminant is rounding noise
     * (|discriminant| < 1e-4 * half_b^2), everything else stays bit-exact. */
    PT_RENDER_GEOMETRIC_CULL = 32u
};

typedef struct pt_stats {
    uint64_t rays;         /* nearest-hit queries: one per (*world)->hit call at main.cu:54 */
    uint64_t primary_rays;
    uint64_t box_tests, prim_tests; /* only with PT_RENDER_COUNT_TESTS */
    uint64_t exact_fallbacks;       /* rays resolved by a list-order scan of the objects: a hit at t = NaN, a non-finite ray, more than 16
                                       candidate objects, a cone whose quadratic's leading coefficient has all but cancelled and whose
                                       roots pass its tests all the same, or (scenes traversed from on-chip memory) one whose
                                       coefficient is exactly 0 for the ray, cone.h:108 */
    double ms_total;                /* device time of the whole call (events on the call's stream) */
    double ms_raygen, ms_intersect, ms_shade, ms_film; /* only with PT_RENDER_STAGE_TIMES | PT_RENDER_SPLIT_KERNELS */
    uint32_t launches_intersect; /* launches of the dominant kernel (k_intersect, or k_paths) */
    uint32_t passes;
    double ms_paths; /* only with PT_RENDER_STAGE_TIMES: summed k_paths launch time */
    uint64_t exact_replays; /* rays whose list-order dependent answer was replayed over their candidate objects (a cone with a < 0,
                               cone.h:116-120; in the streaming kernel also the rays of the cancelling cones above) */
    uint32_t kernel;        /* which pipeline rendered the call: PT_KERNEL_* */
    uint32_t reserved;
} pt_stats;
enum { PT_KERNEL_PATHS = 1 /* k_paths */, PT_KERNEL_PATHS_STREAM = 2 /* k_paths_stream + k_film */, PT_KERNEL_SPLIT = 3 /* one launch per stage and bounce */,
       PT_KERNEL_PATHS_RING = 4 /* k_paths_ring */ };

typedef struct pt_context pt_context;

/* Scheduling and tree-shape parameters of a context.  They never change what is computed (every setting gives the list
 * scan's answers and, for renders, bit-identical images); they exist for A/B measurements and tests.  A negative field
 * means "the library's own choice".  Nothing in the reference corresponds: it has one kernel shape, main.cu:111-117. */
typedef struct pt_tuning {
    int32_t leaf_max;          /* objects per tree leaf, 1..8; applies from the next pt_scene_upload */
    int32_t wide_nodes;        /* 0 / 1: four-wide quantised nodes for scenes traversed from HBM; next pt_scene_upload */
    int32_t pipeline;          /* 0 queue-based persistent workgroups, a slice at a time (k_paths); 1 streaming lanes (k_paths_stream); 2 queue-based with the
                                  slices of a workgroup overlapped (k_paths_ring; scenes that fit on-chip memory, max_depth <= 1023) */
    int32_t lds_scene;         /* 0: never copy the scene into LDS */
    int32_t block;             /* threads per workgroup: 256 or 512 */
    int32_t workgroups_per_cu; /* 1..16 */
    int32_t slice_paths;       /* paths per slice of k_paths, 64..2^22 */
    int32_t slice_taper;       /* 1..64: the last slices of a frame shrink to 1/taper of a full one */
    int32_t sort_by_material;  /* 0 / 1: block-local material sort before shading in k_paths */
    int32_t stream_serve_min;  /* 1..64 lanes waiting before k_paths_stream shades them */
    int32_t stream_phase;      /* 0..16: 0 both kinds of step per iteration; n: a leaf step once the lanes waiting for one outnumber n/4 of those at inner nodes (default 1) */
    int32_t stream_tile;       /* 64..4096, a multiple of 64: pixels per tile of k_paths_stream's path numbering */
    int32_t stream_run;        /* 64..65536: consecutive paths a wave takes at a time */
    int32_t ring_low_water;    /* 0..2^22: k_paths_ring takes the next slice when fewer rays than this are queued (0: only when none is) */
    int32_t split_queue;       /* 0 / 1: k_paths on an LDS-resident scene of spheres queues the rays that head up (camera's up vector) apart from the others; the library's choice: scenes with a non-black background */
} pt_tuning;
void pt_tuning_default(pt_tuning *t); /* every field negative */
/* PT_ERR_INVALID (and no change) if a non-negative field is out of its range. */
pt_status pt_context_configure(pt_context *ctx, const pt_tuning *t);

/* Replaces the implicit device 0 / cudaDeviceReset life cycle of main.cu:106-220. */
pt_status pt_context_create(int device_id, pt_context **out);
void pt_context_destroy(pt_context *ctx);
/* Text of the last failure on this context ("" if none); valid until the next call on it. */
const char *pt_last_error(const pt_context *ctx);

/* camera::camera (camera.h:38-62), host side, fp32, same operation order. */
pt_status pt_camera_look_at(const float lookfrom[3], const float lookat[3], const float vup[3], float vfov_deg,
                            float aspect_ratio, float aperture, float focus_dist, pt_camera *out);

/* Replaces the upload sequence of scene.h:20-58 (create_material_on_gpu / create_hittable_on_gpu /
 * hittable_list::copy_list_to_gpu / create_camera_on_gpu): flattens to SoA records, builds the BVH
 * over the reference's bounding_box() volumes (aabb.h, sphere.h:112-117, aarect.h:45-50, cone.h:142-147,
 * cylinder.h:133-138) and copies everything to the device. */
pt_status pt_scene_upload(pt_context *ctx, const pt_scene_desc *desc);

/* A changed scene with the same number of objects (moved or resized objects, other types or materials,
 * another camera or background): keeps the uploaded scene's tree and refits its boxes, O(objects), instead
 * of rebuilding it.  The reference has no counterpart -- it rebuilds its whole scene per run (scene.h:20-58,
 * main.cu:157-186) -- and no hierarchy to keep.  Answers are those of a fresh pt_scene_upload of `desc`
 * (the list scan's, bit for bit); only the traversal cost follows how far the objects moved. */
pt_status pt_scene_update(pt_context *ctx, const pt_scene_desc *desc);

/* hittable_list::hit (hittable_list.h:87-101) for n rays.  o_xyz / d_xyz / out are host buffers.
 * mode PT_TRACE_BVH is the production traversal, PT_TRACE_LINEAR the list-order scan on the GPU.
 * PT_TRACE_BVH_IN_SCENE is the traversal exactly as pt_render runs it: the caller asserts that every
 * origin lies within the scene's bounds extended to the camera (as a render's origins do), which lets
 * the traversal narrow its conservative ray cone; with origins outside, hits far from the origin
 * may be lost.  All three give the same answers where their preconditions hold.
 * PT_TRACE_BVH_GEOMETRIC is the traversal of a PT_RENDER_GEOMETRIC_CULL render (boxes culled by the objects' exact
 * geometry, no ray cone): it may lose hits whose discriminant is rounding noise, and only those. */
pt_status pt_trace_rays(pt_context *ctx, const float *o_xyz, const float *d_xyz, size_t n, float t_min, float t_max,
                        int mode, pt_hit *out);
/* Same with device pointers, enqueued on `stream` (a hipStream_t; NULL = the default stream). */
pt_status pt_trace_rays_device(pt_context *ctx, const float *d_o_xyz, const float *d_d_xyz, size_t n, float t_min,
                               float t_max, int mode, pt_hit *d_out, void *stream);

/* material::scatter (material.h:90-99 lambertian, :155-162 metal, :223-242 dielectric, :295-299 diffuse_light) and
 * material::emitted (material.h:45-47, :302-304) for n given hits, with the random numbers handed in instead of drawn:
 * hit i is ray (o, d) meeting object prim[i] (list index) at distance t[i]; sphere_pt is the accepted
 * random_in_unit_sphere point (cuda_functions.h:14-20) the lambertian and metal scatters use, xi the uniform the
 * dielectric compares its reflectance with (material.h:235).  The device code is the render's own (the hit record is
 * rebuilt from (object, t) as the render does it).  A parity entry like pt_trace_rays; host buffers. */
typedef struct pt_scatter {
    uint32_t scattered;   /* scatter()'s return value */
    float attenuation[3];
    float origin[3];      /* the scattered ray: the hit point ... */
    float direction[3];   /* ... and its direction (not normalised, like the reference's) */
    float emitted[3];
} pt_scatter;
pt_status pt_scatter_batch(pt_context *ctx, const float *o_xyz, const float *d_xyz, const float *t, const uint32_t *prim,
                           const float *sphere_pt_xyz, const float *xi, size_t n, pt_scatter *out);

/* render (main.cu:81-100) without its final division and gamma: adds, for every pixel, the sum of
 * ray_color over samples [sample_offset, sample_offset + spp) into rgb_sum (width*height*3 fp32, pixel
 * index j*width+i as main.cu:86).  pt_render zero-fills and writes a host buffer; pt_render_device ADDS
 * into a caller-provided device buffer on `stream`, so that shards of one image on several GPUs can be
 * combined with one all-reduce before pt_film_finalize.  stats may be NULL. */
pt_status pt_render(pt_context *ctx, const pt_render_params *params, float *rgb_sum_host, pt_stats *stats);
pt_status pt_render_device(pt_context *ctx, const pt_render_params *params, float *d_rgb_sum, void *stream,
                           pt_stats *stats);

/* Page-locked host memory for the buffers pt_render and pt_group_render fill (the reference's fb is managed memory read
 * back by page migration, main.cu:167-170, :192-205): a film in such a buffer arrives at the link's rate, a pageable one
 * (malloc, numpy) at about a fifth of it through the library's own staging buffer.  Either kind is accepted.  NULL on
 * failure; free with pt_host_free. */
void *pt_host_alloc(size_t bytes);
void pt_host_free(void *p);

/* pt_render_device for callers that cannot allocate: code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc.  It has no counterpart in the reference.  It calls no hipMalloc / hipFree / event or stream creation
 * and synchronises nothing: the scene must have been uploaded (and pt_context_configure called, if at all) beforehand,
 * and its only device scratch -- the pass buffer of finished path radiance, the pass counter and the ray counters -- is
 * a fixed PT_NOALLOC_SCRATCH_BYTES array inside the library, one copy per GPU, held until the process ends.  The frame
 * is rendered by the streaming pipeline in passes of at most PT_NOALLOC_SCRATCH_BYTES / 12 paths; same image, bit for
 * bit, as pt_render_device.  Limits (PT_ERR_LIMIT): width * height * spp <= 2^32 paths per call, and one 64-pixel
 * tile of samples_per_pass samples (256 if 0) must fit the array.  Every call on a GPU shares that GPU's array: two
 * calls that overlap in time on one GPU (different streams, different contexts) race; calls on one stream do not. */
#define PT_NOALLOC_SCRATCH_BYTES (32u << 20)
pt_status pt_render_device_noalloc(pt_context *ctx, const pt_render_params *params, float *d_rgb_sum, void *stream);

/* One image on several GPUs of a node, in one process.  Nothing in the reference corresponds (one device, main.cu:181-186).
 * A group holds one context per GPU and one RCCL communicator over them (RCCL is loaded on first use); the scene is
 * replicated; pt_group_render cuts params->spp into one contiguous range of absolute sample indices per GPU
 * ([g spp / G, (g + 1) spp / G) from params->sample_offset), renders them concurrently (pt_render_device, one host
 * thread per GPU that lives as long as the group), sums the linear accumulation buffers with one ncclAllReduce over xGMI and copies the sum to the
 * host: feed it to pt_film_finalize with the total spp (the gamma is non-linear and must come after the reduction).
 * The image is that of a single pt_render of the same parameters except for the fp32 association of the per-pixel sum.
 * stats: counters summed over the GPUs, ms_total the whole call on the host's clock. */
typedef struct pt_group pt_group;
pt_status pt_group_create(int n_devices, const int *device_ids, pt_group **out);
void pt_group_destroy(pt_group *g);
int pt_group_size(const pt_group *g);
const char *pt_group_last_error(const pt_group *g);
pt_status pt_group_configure(pt_group *g, const pt_tuning *t);
pt_status pt_group_scene_upload(pt_group *g, const pt_scene_desc *desc);
pt_status pt_group_render(pt_group *g, const pt_render_params *params, float *rgb_sum_host, pt_stats *stats);
/* The same without the copy to the host, for callers that finish on the device: *d_rgb_sum is GPU 0's buffer of the
 * whole image's linear sums (every GPU of the group holds the same sums), owned by the group and valid until the next
 * call on it; all of the group's streams are idle on return. */
pt_status pt_group_render_device(pt_group *g, const pt_render_params *params, const float **d_rgb_sum, pt_stats *stats);
/* The range of absolute sample indices GPU `rank` of `n_ranks` renders in pt_group_render: [sample_offset, sample_offset +
 * spp) cut into contiguous ranges that differ by at most one sample.  Host arithmetic only (no GPU needed);
 * PT_ERR_INVALID if spp < n_ranks (every GPU needs a sample). */
pt_status pt_group_shard(uint32_t spp, uint32_t sample_offset, int n_ranks, int rank, uint32_t *offset, uint32_t *count);

/* main.cu:95-98 and :192-205: mean over total_spp (multiply by the reciprocal, vec3.h:136-138), sqrt
 * gamma, then int(256*clamp(c,0,0.999)) with rows written top-down from j = H-1.  Host buffers;
 * rgb_gamma (bottom-up, like fb) and rgb8 (top-down, like the BMP payload) may each be NULL. */
pt_status pt_film_finalize(const float *rgb_sum, int32_t width, int32_t height, uint32_t total_spp, float *rgb_gamma,
                           uint8_t *rgb8);

/* BVH introspection for host-side tests (no GPU needed): builds the same tree pt_scene_upload
 * builds.  nodes_out receives n_nodes*16 floats/ints (64-byte records), order_out n_prims list
 * indices in leaf order.  Pass NULLs to query the sizes. */
pt_status pt_bvh_build_host(const pt_scene_desc *desc, uint32_t *n_nodes, uint32_t *max_depth, void *nodes_out,
                            uint32_t *order_out);
/* The same with a given leaf size (1..8) instead of the one pt_scene_upload would choose for the scene. */
pt_status pt_bvh_build_host_leaf(const pt_scene_desc *desc, uint32_t leaf_max, uint32_t *n_nodes, uint32_t *max_depth,
                                 void *nodes_out, uint32_t *order_out);

/* The tree pt_scene_update leaves behind (host-side tests, no GPU needed): built for `before`, refitted to
 * `after` (same number of objects).  Outputs as for pt_bvh_build_host. */
pt_status pt_bvh_refit_host(const pt_scene_desc *before, const pt_scene_desc *after, uint32_t *n_nodes, uint32_t *max_depth,
                            void *nodes_out, uint32_t *order_out);

/* The four-wide form of that tree, which renders of scenes too large for on-chip memory traverse (host-side tests,
 * no GPU needed; nothing in the reference: its scan is linear).  64-byte records: float origin[3]; uint8 exp[3],
 * n_children; uint8 qlo[3][4], qhi[3][4] ([axis][child]); int32 child[4] (>= 0 a record, < 0 a leaf reference as in
 * the binary nodes, INT32_MIN + 1 unused); 8 bytes of padding.  A child's box on axis a is
 * origin[a] + q * 2^(exp[a] - 127) for q = qlo, qhi, exactly, and encloses the binary tree's box of that child.
 * Pass nodes_out = NULL to size the buffer; *max_depth counts records on the longest path. */
pt_status pt_bvh_wide_host(const pt_scene_desc *desc, uint32_t *n_nodes, uint32_t *max_depth, void *nodes_out);

/* How wide a cone the traversal gives its rays in this scene (host-side tests, no GPU needed).  The
 * reference's fp32 quadratics (sphere.h:85-101, cylinder.h:96-123, cone.h:101-129) accept hits up to
 * about 2^-10 of the origin-to-object distance outside the exact surfaces; the tree culls against a
 * cone of that slope (kappa_max), narrowed per ray (adaptive != 0) to the chord slope of the excess
 * sqrt(r_min^2 + (kappa_max D)^2) - r_min over the ray origin's L1 reach D = |o - cluster_centre|_1 +
 * cluster_half_extent_sum to the objects that rely on it, never below kappa_min. */
typedef struct pt_ray_cone {
    float kappa_max, kappa_min, r_min;
    float cluster_centre[3];
    float cluster_half_extent_sum;
    uint32_t adaptive;
} pt_ray_cone;
pt_status pt_scene_ray_cone(const pt_scene_desc *desc, pt_ray_cone *out);

/* Library build id, e.g. "pathtracer_amd 0.1 gfx950". */
const char *pt_version(void);

#ifdef __cplusplus
}
#endif
#endif
