// pt_launch.h -- what the host side (pt_context.hip) calls to launch the kernels of the pt_kernels_*.hip files.  Each of
// those files instantiates its kernel templates and picks the instance for a launch shape; the host side never names
// a kernel, so a change to one kernel recompiles one file.
#pragma once
#include "pt_trace.h"

// pt_kernels_split.hip
void pt_launch_trace(int mode, uint32_t grid, size_t lds_bytes, hipStream_t stream, const SceneDev &scene, const float *d_o, const float *d_d, size_t n,
                     float t_min, float t_max, pt_hit *d_out, Counters *cnt);
void pt_launch_scatter(uint32_t grid, hipStream_t stream, const SceneDev &scene, const float *d_o, const float *d_d, const float *d_t, const uint32_t *d_prim,
                       const float *d_sphere_pt, const float *d_xi, size_t n, ScatterOut *d_out);
void pt_launch_raygen(uint32_t grid, hipStream_t stream, const SceneDev &scene, const PassDev &ps, const PathPlanes &st, uint32_t *n_active);
// lds_resident_bytes != 0: the variant that copies the scene into LDS (workgroups of PT_ISECT_LDS_BLOCK), else workgroups of PT_BLOCK with lds_bytes of stack
void pt_launch_intersect(bool count, size_t lds_resident_bytes, uint32_t grid, size_t lds_bytes, hipStream_t stream, const SceneDev &scene, const PathPlanes &cur,
                         uint2 *d_hits, const uint32_t *n_ptr, Counters *cnt);
void pt_launch_shade(uint32_t grid, hipStream_t stream, const SceneDev &scene, const PassDev &ps, const PathPlanes &in, const uint2 *d_hits, const PathPlanes &out,
                     const uint32_t *n_in, uint32_t *n_out, float *d_rad, int32_t bounce);
void pt_launch_film(uint32_t grid, hipStream_t stream, const PassDev &ps, const float *d_rad, float *d_rgb_sum, const uint32_t *n_active, Counters *cnt,
                    uint32_t pix0 = 0, uint32_t npix = 0xFFFFFFFFu);

// pt_kernels_paths.hip: kind = PT_KIND_* of the installed scene
void pt_launch_paths(bool count, int kind, Counters *cnt, const SceneDev &scene, hipStream_t stream, int block, bool lds, bool sort, bool split, uint32_t grid,
                     size_t lds_bytes, const PassDev &ps, const SliceDev &sl, float *d_rgb_sum);
// pt_kernels_ring.hip (scenes resident in LDS only)
void pt_launch_ring(bool count, int kind, Counters *cnt, const SceneDev &scene, hipStream_t stream, int block, bool split, uint32_t grid, size_t lds_bytes,
                    const PassDev &ps, const RingDev &rg, float *d_rgb_sum);
// pt_kernels_stream.hip: the k_paths_stream instance for a launch shape, to be launched with hipLaunchKernel (arguments: SceneDev, PassDev, BandDev,
// float *rad, uint32_t *ticket, Counters *, uint32_t serve_min, uint32_t phase_bias, unsigned long long *dbg, uint32_t dbg_waves)
const void *pt_stream_kernel_for(bool count, int block, int kind, bool wide);
