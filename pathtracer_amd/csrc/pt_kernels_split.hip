// pt_kernels_split.hip -- the parity entries and the one-launch-per-stage pipeline.
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
// memory, so the host never waits for a count between bounces.  k_trace (pt_trace_rays) and k_scatter
// (pt_scatter_batch) answer batches handed in by the caller.  The production renders are the persistent kernels of
// pt_kernels_paths.hip, pt_kernels_ring.hip and pt_kernels_stream.hip; k_film also finishes the passes of the last.
#include "pt_launch.h"

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
            const uint32_t scan_from = cone_scan_from<PT_KIND_ANY>(sc, r, t_min, t_max, need_exact);
            if (need_exact) {
                h = linear_nearest(sc.prims_list, sc.n_prims, r, t_min, t_max, scan_from);
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
        const uint32_t scan_from = cone_scan_from<KIND>(sc, r, t_min, t_max, need_exact);
        if (need_exact) {
            // resolved on the list-order records; the top bit of the slot says which array it indexes
            h = linear_nearest(sc.prims_list, sc.n_prims, r, t_min, t_max, scan_from);
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

// pt_scatter_batch: material::scatter and emitted() for given hits with the random numbers handed in.
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
// Launchers (pt_launch.h)
// ---------------------------------------------------------------------------------------------
void pt_launch_trace(int mode, uint32_t grid, size_t lds, hipStream_t stream, const SceneDev &scene, const float *d_o, const float *d_d, size_t n, float t_min,
                     float t_max, pt_hit *d_out, Counters *cnt)
{
    if (mode == PT_TRACE_LINEAR) hipLaunchKernelGGL(k_trace<PT_TRACE_LINEAR>, dim3(grid), dim3(PT_BLOCK), lds, stream, scene, d_o, d_d, n, t_min, t_max, d_out, cnt);
    else if (mode == PT_TRACE_BVH_IN_SCENE) hipLaunchKernelGGL(k_trace<PT_TRACE_BVH_IN_SCENE>, dim3(grid), dim3(PT_BLOCK), lds, stream, scene, d_o, d_d, n, t_min, t_max, d_out, cnt);
    else hipLaunchKernelGGL(k_trace<PT_TRACE_BVH>, dim3(grid), dim3(PT_BLOCK), lds, stream, scene, d_o, d_d, n, t_min, t_max, d_out, cnt);
}

void pt_launch_scatter(uint32_t grid, hipStream_t stream, const SceneDev &scene, const float *d_o, const float *d_d, const float *d_t, const uint32_t *d_prim,
                       const float *d_sphere_pt, const float *d_xi, size_t n, ScatterOut *d_out)
{
    hipLaunchKernelGGL(k_scatter, dim3(grid), dim3(PT_BLOCK), 0, stream, scene, d_o, d_d, d_t, d_prim, d_sphere_pt, d_xi, n, d_out);
}

void pt_launch_raygen(uint32_t grid, hipStream_t stream, const SceneDev &scene, const PassDev &ps, const PathPlanes &st, uint32_t *n_active)
{
    hipLaunchKernelGGL(k_raygen, dim3(grid), dim3(PT_BLOCK), 0, stream, scene, ps, st, n_active);
}

void pt_launch_intersect(bool count, size_t lds_res, uint32_t grid, size_t lds, hipStream_t stream, const SceneDev &scene, const PathPlanes &cur, uint2 *d_hits,
                         const uint32_t *n_ptr, Counters *cnt)
{
    if (lds_res) {
        if (count)
            hipLaunchKernelGGL((k_intersect<true, true, PT_ISECT_LDS_BLOCK, PT_KIND_ANY>), dim3(grid), dim3(PT_ISECT_LDS_BLOCK), lds_res, stream, scene, cur, d_hits, n_ptr, cnt);
        else
            hipLaunchKernelGGL((k_intersect<false, true, PT_ISECT_LDS_BLOCK, PT_KIND_ANY>), dim3(grid), dim3(PT_ISECT_LDS_BLOCK), lds_res, stream, scene, cur, d_hits, n_ptr, cnt);
    } else {
        if (count)
            hipLaunchKernelGGL((k_intersect<true, false, PT_BLOCK, PT_KIND_ANY>), dim3(grid), dim3(PT_BLOCK), lds, stream, scene, cur, d_hits, n_ptr, cnt);
        else
            hipLaunchKernelGGL((k_intersect<false, false, PT_BLOCK, PT_KIND_ANY>), dim3(grid), dim3(PT_BLOCK), lds, stream, scene, cur, d_hits, n_ptr, cnt);
    }
}

void pt_launch_shade(uint32_t grid, hipStream_t stream, const SceneDev &scene, const PassDev &ps, const PathPlanes &in, const uint2 *d_hits, const PathPlanes &out,
                     const uint32_t *n_in, uint32_t *n_out, float *d_rad, int32_t bounce)
{
    hipLaunchKernelGGL(k_shade, dim3(grid), dim3(PT_BLOCK), 0, stream, scene, ps, in, d_hits, out, n_in, n_out, d_rad, bounce);
}

void pt_launch_film(uint32_t grid, hipStream_t stream, const PassDev &ps, const float *d_rad, float *d_rgb_sum, const uint32_t *n_active, Counters *cnt, uint32_t pix0,
                    uint32_t npix)
{
    hipLaunchKernelGGL(k_film, dim3(grid), dim3(PT_BLOCK), 0, stream, ps, d_rad, d_rgb_sum, n_active, cnt, pix0, npix);
}
