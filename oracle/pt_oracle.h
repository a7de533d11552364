/*
 * pt_oracle.h -- CPU restatement of the reference path tracer's hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under pathtracer_amd/ may include, link
 * or load this.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / the CPU baseline.
 *
 * PARITY UNPINNED by the reference's own tests: the reference tree holds no
 * tests, fixtures or golden images, and it cannot be compiled in this image
 * (it needs the CUDA toolkit headers cuda_runtime.h, device_launch_parameters.h,
 * curand_kernel.h, the git-ignored stb_image*.h and nvcc's <<<>>> launch
 * syntax).  The single reference-derived vector that exists -- the centre-ray
 * hit on the default scene recorded at survey time from a host build of the
 * reference headers -- is reproduced bit-for-bit (tests/test_oracle.py).
 *
 * Every function cites the reference file:line (relative to the reference
 * tree) whose arithmetic it restates, operation by operation, in the same
 * order, in fp32, compiled with -ffp-contract=off.
 *
 * The random number source is NOT the reference's (cuRAND XORWOW,
 * main.cu:34,87): it is the counter-based Philox4x32-7 stream the GPU path
 * uses -- the words of blocks 0, 1, 2, ... of the counter (block, bounce, sample,
 * pixel) taken in order, one per uniform drawn -- so oracle and GPU walk the
 * same paths.  The sampler algorithms and the draw order are the reference's.
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_SPHERE = 0, ORC_XY_RECT = 1, ORC_XZ_RECT = 2, ORC_YZ_RECT = 3, ORC_CONE = 4, ORC_CYLINDER = 5 };
enum { ORC_LAMBERTIAN = 0, ORC_METAL = 1, ORC_DIELECTRIC = 2, ORC_DIFFUSE_LIGHT = 3 };

/* A flattened scene: one type tag, six floats and a material index per object.
 *   sphere   : cx cy cz radius - -
 *   xy_rect  : x0 x1 y0 y1 k -        xz_rect: x0 x1 z0 z1 k -      yz_rect: y0 y1 z0 z1 k -
 *   cone     : cx cy cz height radius tangens      (tangens = (r/h)*(r/h), cone.h:30)
 *   cylinder : cx cy cz height radius -
 * materials: lambertian albedo(3) ; metal albedo(3) fuzz(<=1) ; dielectric ir ; diffuse_light emit(3)
 */
typedef struct orc_scene {
    uint32_t n_prims;
    const uint8_t *prim_type;
    const float *prim_param;       /* n_prims * 6 */
    const uint32_t *prim_material; /* n_prims */
    uint32_t n_materials;
    const uint8_t *mat_type;
    const float *mat_param;        /* n_materials * 4 */
    float cam_origin[3], cam_lower_left[3], cam_horizontal[3], cam_vertical[3];
    float cam_u[3], cam_v[3], cam_w[3];
    float cam_lens_radius;
    float background[3];
} orc_scene;

typedef struct orc_hit {
    float t;
    uint32_t prim;      /* list index, 0xFFFFFFFF on a miss */
    float n[3];         /* face-forwarded normal (hit_record::normal) */
    uint32_t front;     /* hit_record::front_face */
    float p[3];
    uint32_t nan_flag;  /* nonzero = outside the parity contract: bit 0 some object reported a NaN t for this ray,
                         * bit 1 the scan accepted a cone hit whose quadratic's leading coefficient had cancelled */
} orc_hit;

/* camera.h:38-62 */
void orc_camera_make(const float lookfrom[3], const float lookat[3], const float vup[3], float vfov_deg,
                     float aspect, float aperture, float focus_dist, orc_scene *out);

/* One object, hittable::hit (sphere.h:85-110, aarect.h:348-394, cone.h:101-139, cylinder.h:96-130). */
int orc_hit_prim(const orc_scene *s, uint32_t i, const float o[3], const float d[3], float t_min, float t_max,
                 orc_hit *rec);
/* hittable_list::hit (hittable_list.h:87-101), with the winning list index. */
int orc_hit_list(const orc_scene *s, const float o[3], const float d[3], float t_min, float t_max, orc_hit *rec);
/* n rays, o/d are n*3 floats; threads <= 0 means all cores. */
void orc_trace_batch(const orc_scene *s, const float *o, const float *d, size_t n, float t_min, float t_max,
                     orc_hit *out, int threads);

/* Philox4x32 (Salmon et al., SC'11), one block of `rounds` rounds (10 is the published default, 7 the
 * fewest rounds the paper certifies as passing BigCrush and what the renderer's stream uses). */
#define ORC_PHILOX_ROUNDS 7
void orc_philox4x32(const uint32_t ctr[4], const uint32_t key[2], int rounds, uint32_t out[4]);
/* The uniform the renderer draws: (0,1], like curand_uniform. */
float orc_u32_to_uniform(uint32_t x);

/* material::scatter with explicit randomness (material.h:90-99,155-162,223-242,295-299):
 * sphere_pt is the accepted random_in_unit_sphere point, xi the dielectric's uniform.
 * Returns the reference's bool; writes attenuation and the scattered ray. */
int orc_scatter(const orc_scene *s, uint32_t mat, const float ray_o[3], const float ray_d[3], const orc_hit *rec,
                const float sphere_pt[3], float xi, float atten[3], float out_o[3], float out_d[3]);

/* camera::get_ray (camera.h:88-96) with an explicit unit-disk point. */
void orc_get_ray(const orc_scene *s, float u, float v, const float disk_pt[2], float out_o[3], float out_d[3]);

/* render (main.cu:81-100) before the division by spp: rgb_sum[pixel*3+c] = sum over samples
 * [sample0, sample0+n_samples) of ray_color (main.cu:46-65) at the given depth, added in sample order.
 * Returns the number of nearest-hit queries (rays).  threads <= 0 means all cores. */
uint64_t orc_render(const orc_scene *s, int width, int height, uint32_t sample0, uint32_t n_samples, int depth,
                    uint64_t seed, float *rgb_sum, int threads);
/* Same walk, but records the k-th ray of each listed path for fixture making: not part of the reference. */
uint64_t orc_render_region(const orc_scene *s, int width, int height, int x0, int y0, int w, int h,
                           uint32_t sample0, uint32_t n_samples, int depth, uint64_t seed, float *rgb_sum,
                           int threads);

/* NOT the reference and not a parity checker: orc_render with its nearest-hit queries answered by a stack traversal
 * of a two-wide tree (the GPU library's host-side build: 64-byte nodes, leaf slots mapped to list indices by `order`)
 * instead of the list scan, so that bench.py can put a CPU figure with the same O(log N) algorithm beside the
 * linear-scan one.  Not reentrant. */
uint64_t orc_render_bvh(const orc_scene *s, const float *nodes, uint32_t n_nodes, const uint32_t *order, int width, int height,
                        uint32_t sample0, uint32_t n_samples, int depth, uint64_t seed, float *rgb_sum, int threads);

/* NOT the reference and not a checker: box tests and object tests a plain near-first stack traversal of the given
 * two-wide tree performs over a batch of rays (see pt_oracle.c), for the roofline's frozen work-per-ray figures. */
void orc_count_tree_work(const orc_scene *s, const float *nodes, uint32_t n_nodes, const uint32_t *order, const float *o, const float *d,
                         size_t n, float t_min, float t_max, uint64_t *box_tests, uint64_t *prim_tests);

/* main.cu:95-98 then :194-205: mean, sqrt gamma, 8-bit quantise, rows top-down from j = H-1. */
void orc_finalize(const float *rgb_sum, int width, int height, uint32_t spp, float *rgb_gamma, uint8_t *rgb8);

int orc_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
