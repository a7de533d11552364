/* The CPU oracle under AddressSanitizer / UBSan: a small scene with every object kind and every material goes through
 * the batch trace (ordinary, axis-parallel and zero-direction rays), the per-call scatter, the render (whole image and
 * a region) and the film finish.  Compiled and run by tests/test_host_sanitizers.py with oracle/pt_oracle.c; the
 * checker is test infrastructure, and a checker that reads out of bounds or relies on undefined behaviour checks nothing. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../oracle/pt_oracle.h"

static uint32_t lcg(uint32_t *s) { return *s = *s * 1664525u + 1013904223u; }
static float unit(uint32_t *s) { return (float)(lcg(s) >> 8) * (1.0f / 16777216.0f); }

int main(void)
{
    enum { N = 14, M = 5 };
    uint8_t type[N];
    float param[N * 6];
    uint32_t mat[N];
    uint8_t mtype[M] = {ORC_LAMBERTIAN, ORC_METAL, ORC_DIELECTRIC, ORC_DIFFUSE_LIGHT, ORC_METAL};
    float mparam[M * 4] = {0.5f, 0.5f, 0.5f, 0, 0.7f, 0.6f, 0.5f, 0.3f, 1.5f, 0, 0, 0, 4, 4, 4, 0, 0.9f, 0.9f, 0.9f, 0.0f};
    uint32_t seed = 12345u;
    memset(param, 0, sizeof param);
    for (int i = 0; i < N; ++i) {
        float *p = param + 6 * i;
        type[i] = (uint8_t)(i % 6);
        mat[i] = (uint32_t)(i % M);
        switch (type[i]) {
        case ORC_SPHERE:
            p[0] = 4 * unit(&seed) - 2; p[1] = 0.5f; p[2] = 4 * unit(&seed) - 2; p[3] = 0.5f;
            break;
        case ORC_XY_RECT: case ORC_XZ_RECT: case ORC_YZ_RECT:
            p[0] = -2; p[1] = 2; p[2] = -2; p[3] = 2; p[4] = type[i] == ORC_XZ_RECT ? 3.0f : -3.0f;
            break;
        case ORC_CONE:
            p[0] = 4 * unit(&seed) - 2; p[1] = 0; p[2] = 4 * unit(&seed) - 2; p[3] = 1.0f; p[4] = 0.4f;
            p[5] = (p[4] / p[3]) * (p[4] / p[3]);
            break;
        default:
            p[0] = 4 * unit(&seed) - 2; p[1] = 0; p[2] = 4 * unit(&seed) - 2; p[3] = 0.8f; p[4] = 0.3f;
            break;
        }
    }
    /* a ground sphere, as in the reference's scenes */
    type[0] = ORC_SPHERE; param[0] = 0; param[1] = -1000; param[2] = 0; param[3] = 1000; mat[0] = 0;

    orc_scene s;
    memset(&s, 0, sizeof s);
    s.n_prims = N; s.prim_type = type; s.prim_param = param; s.prim_material = mat;
    s.n_materials = M; s.mat_type = mtype; s.mat_param = mparam;
    const float from[3] = {6, 2, 5}, at[3] = {0, 0.5f, 0}, up[3] = {0, 1, 0};
    orc_camera_make(from, at, up, 40.0f, 1.5f, 0.1f, 7.0f, &s);
    s.background[0] = 0.7f; s.background[1] = 0.8f; s.background[2] = 1.0f;

    /* batch trace: random rays, rays along the axes, rays from inside objects, a zero direction */
    enum { R = 4096 };
    float *o = malloc(sizeof(float) * 3 * R), *d = malloc(sizeof(float) * 3 * R);
    orc_hit *hits = malloc(sizeof(orc_hit) * R);
    if (!o || !d || !hits) return 2;
    for (int i = 0; i < R; ++i) {
        for (int a = 0; a < 3; ++a) { o[3 * i + a] = 8 * unit(&seed) - 4; d[3 * i + a] = 2 * unit(&seed) - 1; }
        if (i % 64 == 1) { d[3 * i] = 0; d[3 * i + 2] = 0; }                 /* exactly vertical */
        if (i % 64 == 2) { d[3 * i + 1] = 0; o[3 * i + 1] = 3.0f; }           /* in the plane of an xz_rect */
        if (i % 64 == 3) { d[3 * i] = d[3 * i + 1] = d[3 * i + 2] = 0; }      /* no direction at all */
        if (i % 64 == 4) { o[3 * i] = param[6]; o[3 * i + 1] = param[7]; o[3 * i + 2] = param[8]; }
    }
    orc_trace_batch(&s, o, d, R, 0.001f, INFINITY, hits, 2);
    size_t n_hit = 0;
    for (int i = 0; i < R; ++i) {
        if (hits[i].prim != 0xFFFFFFFFu) {
            if (hits[i].prim >= N) { fprintf(stderr, "oracle_san: hit names object %u\n", hits[i].prim); return 1; }
            ++n_hit;
            /* the scatter of every material at this hit */
            for (uint32_t m = 0; m < M; ++m) {
                const float sp[3] = {0.3f, -0.2f, 0.1f};
                float att[3], so[3], sd[3];
                orc_scatter(&s, m, o + 3 * i, d + 3 * i, &hits[i], sp, unit(&seed), att, so, sd);
            }
        }
    }

    enum { W = 36, H = 24 };
    float *sum = calloc((size_t)W * H * 3, sizeof(float)), *sum2 = calloc((size_t)W * H * 3, sizeof(float));
    float *gamma = malloc(sizeof(float) * W * H * 3);
    uint8_t *rgb8 = malloc((size_t)W * H * 3);
    if (!sum || !sum2 || !gamma || !rgb8) return 2;
    const uint64_t rays = orc_render(&s, W, H, 0, 4, 8, 1984u, sum, 2);
    const uint64_t rays_r = orc_render_region(&s, W, H, 5, 7, 11, 9, 0, 4, 8, 1984u, sum2, 2);
    /* the region's pixels are the whole image's pixels */
    for (int y = 0; y < 9; ++y)
        for (int x = 0; x < 11; ++x)
            for (int c = 0; c < 3; ++c) {
                const size_t k = ((size_t)(7 + y) * W + 5 + x) * 3 + c, kr = ((size_t)y * 11 + x) * 3 + c; /* the region is stored packed */
                if (memcmp(&sum[k], &sum2[kr], sizeof(float)) != 0) { fprintf(stderr, "oracle_san: region differs at %d,%d\n", x, y); return 1; }
            }
    orc_finalize(sum, W, H, 4, gamma, rgb8);
    /* a film with values the 8-bit conversion must clamp */
    sum[0] = 1e30f; sum[1] = 0.0f; sum[2] = 4.0f * 0.999f;
    orc_finalize(sum, W, H, 4, gamma, rgb8);

    printf("ok %zu of %d rays hit, %llu + %llu rays rendered\n", n_hit, R, (unsigned long long)rays, (unsigned long long)rays_r);
    free(o); free(d); free(hits); free(sum); free(sum2); free(gamma); free(rgb8);
    return 0;
}
