// pt_kernels_paths.hip -- k_paths: persistent workgroups, one slice of paths at a time.
#include "pt_launch.h"

// ---------------------------------------------------------------------------------------------
// Persistent-workgroup wavefront: the same stages (camera ray -> nearest hit -> scatter/emit ->
// compaction -> film), but a workgroup owns a SLICE of paths (all samples of a run of pixels) and
// takes it through every bounce itself.  Its two ping-pong queues and its radiance buffer are a
// private region of HBM (cache-resident in practice), queue lengths live in LDS, compaction is a
// wave ballot + one LDS atomic per wave, stages are separated by workgroup barriers instead of
// kernel launches, and slices are handed out by a ticket counter.  No global atomics on the path,
// no per-bounce launches, no latency-bound tail launches.
// ---------------------------------------------------------------------------------------------
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
        const uint32_t scan_from = cone_scan_from<KIND>(sc, r, t_min, t_max, need_exact);
        if (need_exact) {
            h = linear_nearest(sc.prims_list, sc.n_prims, r, t_min, t_max, scan_from);
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

// The k_paths instance for a launch shape (pt_launch.h).
template <bool COUNT>
static void launch_paths(int kind, Counters *cnt, const SceneDev &scene, hipStream_t stream, int block, bool lds, bool sort, bool split, uint32_t grid, size_t lds_bytes,
                         const PassDev &ps, const SliceDev &sl, float *d_rgb_sum)
{
#define PT_LAUNCH_PATHS(L, B, S, K) hipLaunchKernelGGL((k_paths<COUNT, L, B, S, K>), dim3(grid), dim3(B), lds_bytes, stream, scene, ps, sl, d_rgb_sum, cnt)
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
            if (block == 256) hipLaunchKernelGGL((k_paths<false, true, 256, false, PT_KIND_ANY, false, true>), dim3(grid), dim3(256), lds_bytes, stream, scene, ps, sl, d_rgb_sum, cnt);
            else hipLaunchKernelGGL((k_paths<false, true, 512, false, PT_KIND_ANY, false, true>), dim3(grid), dim3(512), lds_bytes, stream, scene, ps, sl, d_rgb_sum, cnt);
        } else {
            if (block == 256) hipLaunchKernelGGL((k_paths<false, true, 256, false, PT_KIND_NO_CONES, false, true>), dim3(grid), dim3(256), lds_bytes, stream, scene, ps, sl, d_rgb_sum, cnt);
            else hipLaunchKernelGGL((k_paths<false, true, 512, false, PT_KIND_NO_CONES, false, true>), dim3(grid), dim3(512), lds_bytes, stream, scene, ps, sl, d_rgb_sum, cnt);
        }
    } else if (lds && split && !sort && kind == PT_KIND_SPHERES) { // (the one shape the two-ended queue is built for)
        if (block == 256) hipLaunchKernelGGL((k_paths<COUNT, true, 256, false, PT_KIND_SPHERES, true>), dim3(grid), dim3(256), lds_bytes, stream, scene, ps, sl, d_rgb_sum, cnt);
        else hipLaunchKernelGGL((k_paths<COUNT, true, 512, false, PT_KIND_SPHERES, true>), dim3(grid), dim3(512), lds_bytes, stream, scene, ps, sl, d_rgb_sum, cnt);
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


void pt_launch_paths(bool count, int kind, Counters *cnt, const SceneDev &scene, hipStream_t stream, int block, bool lds, bool sort, bool split, uint32_t grid,
                     size_t lds_bytes, const PassDev &ps, const SliceDev &sl, float *d_rgb_sum)
{
    if (count) launch_paths<true>(kind, cnt, scene, stream, block, lds, sort, split, grid, lds_bytes, ps, sl, d_rgb_sum);
    else launch_paths<false>(kind, cnt, scene, stream, block, lds, sort, split, grid, lds_bytes, ps, sl, d_rgb_sum);
}
