// pt_kernels_ring.hip -- k_paths_ring: persistent workgroups whose slices overlap.
#include "pt_launch.h"

// ---------------------------------------------------------------------------------------------
// k_paths_ring: k_paths with the slices of a workgroup overlapped.  In k_paths a workgroup takes ONE slice through all
// its bounces before it starts the next, and the bounces after the first few hold a handful of rays each: a bounce
// costs the workgroup at least the latency of one wave's trace and shade whatever it holds, so the forty-odd tail
// bounces of every slice are paid in full (14 % of all wave time at 64 samples per pixel on the random spheres, where a
// slice is 32k paths; 49 % at 8, where a GPU's share of a strong-scaled frame makes slices of 5k).  Here the queue
// never runs dry: whenever fewer than `low_water` rays are queued at the start of a generation, the workgroup takes
// the next slice and its camera rays join the queue's survivors, so every generation but the last few of the launch
// has several full chunks to work on and the stragglers of earlier slices ride along.  A path carries its own bounce
// count and the ring slot of its slice (up to PT_RING slices are under way per workgroup, each with its own radiance
// buffer and a count of unfinished paths in LDS); a slice whose count reaches zero is summed into the film in sample
// order (main.cu:93) at the start of the next generation.  Same paths, same arithmetic, same film order as k_paths:
// bit-identical images.
// ---------------------------------------------------------------------------------------------
// Adds a finished slice to the film: film[e] += the samples of rad[sample][e], e = 3 * pixel + channel, one after the
// other in sample order (main.cu:93).  A call of its own, so that its registers are not the path loop's.
static __device__ __attribute__((noinline)) void ring_film(const float *__restrict__ rk, float *__restrict__ film, uint32_t n3, uint32_t samples, uint32_t block)
{
    for (uint32_t e = threadIdx.x; e < n3; e += block) {
        float acc = film[e];
        const float *src = rk + e;
        uint32_t s = 0;
        for (; s + 8 <= samples; s += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(s + u) * n3];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
        for (; s < samples; ++s) acc += src[(size_t)s * n3];
        film[e] = acc;
    }
}

template <bool COUNT, bool LDS, int BLOCK, int KIND, bool SPLIT = false, bool FLAT = false>
__global__ __launch_bounds__(BLOCK, 6) void k_paths_ring(SceneDev sc, PassDev ps, RingDev rg, float *__restrict__ rgb_sum, Counters *cnt)
{
    extern __shared__ float4 s_mem[];
    __shared__ uint32_t s_out[4];                                  // [parity][0: queued from the front, 1: from the back (SPLIT)]
    __shared__ uint32_t s_left[PT_RING], s_pix0[PT_RING], s_npix[PT_RING]; // per ring slot: unfinished paths, the slice's extent
    __shared__ uint32_t s_open, s_fresh_slot, s_fresh_n, s_exhausted, s_chunk;
    __shared__ uint32_t s_next_state, s_next_pix0, s_next_npix;    // the slice taken ahead of time: 0 none asked for, 1 a slice, 2 the table is used up
    uint32_t *s_stack;
    LdsSrc lsrc{nullptr, nullptr, sc.n_nodes, sc.n_prims, sc.cone};
    if (LDS) {
        s_stack = reinterpret_cast<uint32_t *>(lds_scene_copy<BLOCK>(sc, s_mem, lsrc));
    } else {
        s_stack = reinterpret_cast<uint32_t *>(s_mem);
    }
    GlobalSrc gsrc{sc.nodes, sc.prims, sc.cone};
    const uint32_t qcap = rg.qcap, cap = rg.cap;
    float4 *const qA = rg.qA + (size_t)blockIdx.x * 2 * qcap;
    float4 *const qB = rg.qB + (size_t)blockIdx.x * 2 * qcap;
    float2 *const qC = rg.qC + (size_t)blockIdx.x * 2 * qcap;
    float *const rad = rg.rad + (size_t)blockIdx.x * PT_RING * 3 * cap;
    const float t_min = 0.001f /* main.cu:54 */, t_max = PT_FLT_MAX /* commons.h:11 */;
    const uint32_t lane = threadIdx.x & 63u;
    unsigned long long rays = 0;
    uint32_t nb = 0, np = 0, nf = 0, nr = 0;
    uint32_t my_rays = 0; // rays this lane traced
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
    // The next slice is asked for one generation ahead by one lane, which waits for the ticket and the slice's extent
    // while the other waves work; the workgroup never waits for the counter in HBM.
    auto take_ahead = [&]() {
        const uint32_t ticket = atomicAdd(rg.ticket, 1u);
        if (ticket < rg.n_slices) {
            const uint2 e = rg.extents[ticket];
            s_next_pix0 = e.x;
            s_next_npix = e.y;
            s_next_state = 1;
        } else {
            s_next_state = 2;
        }
    };
    if (threadIdx.x == 0) {
        s_out[0] = s_out[1] = s_out[2] = s_out[3] = 0;
        s_open = 0; s_exhausted = 0; s_fresh_n = 0; s_fresh_slot = 0; s_chunk = 0;
        take_ahead();
    }
    constexpr uint32_t PATH_MASK = (1u << PT_RING_PATH_BITS) - 1u, SLOT_MASK = (1u << PT_RING_SLOT_BITS) - 1u, ALL_OPEN = (1u << PT_RING) - 1u;
    constexpr uint32_t BOUNCE_ONE = 1u << (PT_RING_PATH_BITS + PT_RING_SLOT_BITS);
    const uint32_t pix_bits = rg.pix_bits, pix_mask = (1u << pix_bits) - 1u;
    uint32_t cur = 0;
    uint32_t summed = 0; // slices summed into the film during the last generation: their slots are free from this one on
    long long t_drain = 0;
    for (;;) {
        __syncthreads(); // (A) the last generation's queue entries, radiance and counts are in place (the first time: the scene copy)
        mark(4);
        const uint32_t n_lo = s_out[2 * cur], n_hi = SPLIT ? s_out[2 * cur + 1] : 0u, n_queue = n_lo + n_hi;
        const uint32_t open = s_open;
        uint32_t closing = 0; // slices whose last path ended in the last generation
        for (uint32_t m = open & ~summed; m; m &= m - 1u) {
            const uint32_t k = (uint32_t)__ffs((int)m) - 1u;
            if (s_left[k] == 0) closing |= 1u << k;
        }
        __syncthreads(); // (B) every thread has read what thread 0 changes now
        if (threadIdx.x == 0) {
            uint32_t now_open = open & ~summed, fresh = 0;
            // the next slice joins when the queue runs low, into a slot that was free before this generation
            if (!s_exhausted && n_queue < rg.low_water && open != ALL_OPEN) {
                if (s_next_state == 2) {
                    s_exhausted = 1;
                } else {
                    const uint32_t slot = (uint32_t)__ffs((int)~open) - 1u;
                    s_pix0[slot] = s_next_pix0;
                    s_npix[slot] = s_next_npix;
                    fresh = s_next_npix * ps.samples;
                    s_left[slot] = fresh;
                    s_fresh_slot = slot;
                    now_open |= 1u << slot;
                    s_next_state = 0;
                }
            }
            s_fresh_n = fresh;
            s_open = now_open;
            s_out[2 * (cur ^ 1u)] = s_out[2 * (cur ^ 1u) + 1] = 0;
            s_chunk = 0;
        }
        __syncthreads(); // (C) thread 0's decisions are visible
        mark(0);
        const uint32_t fresh_n = s_fresh_n, fresh_slot = s_fresh_slot;
        const uint32_t fresh_pix0 = s_pix0[fresh_slot];
        const uint32_t total = n_queue + fresh_n;
        const bool last = total == 0 && s_exhausted != 0; // nothing queued, nothing left to take
        // The end of the launch: no slice left to take and at most one chunk per wave queued.  Every wave then keeps the
        // rays of its chunk in its lanes until the last of them has ended -- no queue, no barrier between their bounces.
        // (A path of fifty bounces is fifty generations; with a handful of rays each, a generation is all latency.)
        const bool drain = s_exhausted != 0 && total <= (uint32_t)BLOCK;
        if (threadIdx.x == BLOCK - 1 && fresh_n) take_ahead(); // (the last wave's: with chunks dealt out on demand it just takes fewer)
        const float4 *inA = qA + cur * qcap, *inB = qB + cur * qcap;
        const float2 *inC = qC + cur * qcap;
        float4 *outA = qA + (cur ^ 1u) * qcap, *outB = qB + (cur ^ 1u) * qcap;
        float2 *outC = qC + (cur ^ 1u) * qcap;
        uint32_t *const out_cnt = &s_out[2 * (cur ^ 1u)];
        // Chunks of 64 rays are dealt out to the waves on demand (one LDS atomic each): rays differ several-fold in cost, and
        // with a fixed share per wave the workgroup would wait at the end of every generation for the wave that drew the dearest.
        for (;;) {
            uint32_t c0 = 0;
            if (lane == 0) c0 = atomicAdd(&s_chunk, 64u);
            c0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)c0);
            if (c0 >= total) break;
            const uint32_t i = c0 + lane;
            const bool valid = i < total;
            Ray r;
            r.o = V(0.0f, 0.0f, 0.0f);
            r.d = V(0.0f, 0.0f, 0.0f);
            v3 T = V(1.0f, 1.0f, 1.0f);
            uint32_t idw = 0; // the path word: pixel within its slice | sample within the pass | ring slot | bounces so far
            HitResult h;
            h.t = 0.0f;
            h.slot = PT_NO_PRIM;
            bool valid_now = valid;
            if (valid) {
                if (i < n_queue) {
                    const uint32_t qi = SPLIT && i >= n_lo ? qcap - 1u - (i - n_lo) : i; // (the rays queued from the back)
                    const float4 A = inA[qi], B = inB[qi];
                    const float2 C = inC[qi];
                    idw = __float_as_uint(A.w);
                    r.o = V(A.x, A.y, A.z);
                    r.d = V(B.x, B.y, B.z);
                    T = V(B.w, C.x, C.y);
                } else {
                    // the new slice's camera rays, numbered sample-fast: a wave starts with the samples of one or a few pixels
                    const uint32_t f = i - n_queue;
                    const uint32_t pl = f / ps.samples, s = f - pl * ps.samples;
                    idw = pl | (s << pix_bits) | (fresh_slot << PT_RING_PATH_BITS);
                    r = camera_ray(sc.cam, ps, fresh_pix0 + pl, ps.sample_base + s);
                }
            }
            mark(1);
            bool survive = false;
            float4 oA, oB;
            float2 oC;
            for (;;) { // (once, except at the end of the launch)
            h.slot = PT_NO_PRIM;
            if (valid_now) {
                my_rays++;
                bool need_exact;
                const int32_t root_ref = FLAT ? sc.root_ref : 0;
                if (LDS) h = bvh_nearest<COUNT, KIND, BLOCK, LdsSrc, FLAT>(lsrc, sc.stack_cap, r, t_min, t_max, s_stack + threadIdx.x, need_exact, nb, np, nr, &prof, root_ref);
                else h = bvh_nearest<COUNT, KIND, BLOCK, GlobalSrc, FLAT>(gsrc, sc.stack_cap, r, t_min, t_max, s_stack + threadIdx.x, need_exact, nb, np, nr, &prof, root_ref);
                const uint32_t scan_from = cone_scan_from<KIND>(sc, r, t_min, t_max, need_exact);
                if (need_exact) { // resolved on the list-order records; the top bit of the slot says which array it indexes
                    h = linear_nearest(sc.prims_list, sc.n_prims, r, t_min, t_max, scan_from);
                    if (h.slot != PT_NO_PRIM) h.slot |= 0x80000000u;
                    nf++;
                }
            }
            mark(2);
            // main.cu:54-64 for the traced ray, then compaction of the wave's survivors into the next queue
            survive = false;
            if (valid_now) {
                // pixel within the slice and sample within the pass, as the path word holds them: no division by the slice's width
                const uint32_t pl = idw & pix_mask, s = (idw & PATH_MASK) >> pix_bits, slot = (idw >> PT_RING_PATH_BITS) & SLOT_MASK;
                const uint32_t lid = s * s_npix[slot] + pl; // where the path reports in its slice's radiance buffer
                v3 result;
                if (h.slot == PT_NO_PRIM) {
                    result = vmul(T, V(sc.bg[0], sc.bg[1], sc.bg[2])); // main.cu:54-55
                } else {
                    PrimRec q;
                    if (LDS && (KIND == PT_KIND_SPHERES || (h.slot & 0x80000000u) == 0)) {
                        float4 w0, w1;
                        lsrc.prim(h.slot, w0, w1);
                        q.p[0] = w0.x; q.p[1] = w0.y; q.p[2] = w0.z; q.p[3] = w0.w; q.p[4] = w1.x; q.p[5] = w1.y;
                        q.meta = __float_as_uint(w1.z);
                        q.id = __float_as_uint(w1.w);
                    } else {
                        q = load_prim((h.slot & 0x80000000u) ? sc.prims_list : sc.prims, h.slot & 0x7FFFFFFFu);
                    }
                    Ray nray;
                    v3 nT;
                    if (kProf) prof.tick(3);
                    survive = shade_hit<KIND == PT_KIND_SPHERES>(sc.mats, ps, r, T, s_pix0[slot] + pl, ps.sample_base + s, (int32_t)(idw >> (PT_RING_PATH_BITS + PT_RING_SLOT_BITS)), q, h.t,
                                                                 nray, nT, result, &prof);
                    if (survive) {
                        oA = make_float4(nray.o.x, nray.o.y, nray.o.z, __uint_as_float(idw + BOUNCE_ONE));
                        oB = make_float4(nray.d.x, nray.d.y, nray.d.z, nT.x);
                        oC = make_float2(nT.y, nT.z);
                    }
                }
                if (!survive) {
                    float *rk = rad + ((size_t)slot * cap + lid) * 3;
                    rk[0] = result.x;
                    rk[1] = result.y;
                    rk[2] = result.z;
                    atomicSub(&s_left[slot], 1u);
                }
            }
            if (!drain || __ballot(survive) == 0ull) break;
            // the end of the launch: the survivors go on from their lanes
            valid_now = survive;
            if (survive) {
                r.o = V(oA.x, oA.y, oA.z);
                r.d = V(oB.x, oB.y, oB.z);
                T = V(oB.w, oC.x, oC.y);
                idw = __float_as_uint(oA.w);
            }
            }
            if (SPLIT) { // as in k_paths: rays that head up (camera's up vector) from the back of the next queue, the others from its front
                const bool up = oB.x * sc.cam.v[0] + oB.y * sc.cam.v[1] + oB.z * sc.cam.v[2] > 0.0f;
                const unsigned long long m0 = __ballot(survive && !up), m1 = __ballot(survive && up);
                if (m0 | m1) {
                    const unsigned long long mine = up ? m1 : m0;
                    const uint32_t pre = lane_prefix(mine);
                    uint32_t pos0 = 0;
                    if (survive && pre == 0) pos0 = atomicAdd(&out_cnt[up ? 1 : 0], (uint32_t)__popcll(mine));
                    const uint32_t p0 = __shfl(pos0, m0 ? __ffsll((long long)m0) - 1 : 0, 64), p1 = __shfl(pos0, m1 ? __ffsll((long long)m1) - 1 : 0, 64);
                    if (survive) {
                        const uint32_t at = up ? qcap - 1u - (p1 + pre) : p0 + pre;
                        outA[at] = oA;
                        outB[at] = oB;
                        outC[at] = oC;
                    }
                }
            } else {
                const unsigned long long mask = __ballot(survive);
                if (mask) {
                    const uint32_t pre = lane_prefix(mask);
                    uint32_t pos0 = 0;
                    if (survive && pre == 0) pos0 = atomicAdd(&out_cnt[0], (uint32_t)__popcll(mask));
                    pos0 = __shfl(pos0, __ffsll((long long)mask) - 1, 64);
                    if (survive) {
                        outA[pos0 + pre] = oA;
                        outB[pos0 + pre] = oB;
                        outC[pos0 + pre] = oC;
                    }
                }
            }
            mark(3);
        }
        // film: col += ray_color(...) in sample order (main.cu:93) for the pixels of the slices that finished in the last
        // generation, by every wave as it runs out of chunks (their slots stay taken until the next generation)
        for (uint32_t m = closing; m; m &= m - 1u) {
            const uint32_t k = (uint32_t)__ffs((int)m) - 1u;
            // one thread per (pixel, colour channel): the samples of a sum are still added one after the other, in sample
            // order, but three times as many threads work and the loads of eight samples are under way together
            ring_film(rad + (size_t)k * 3 * cap, rgb_sum + (size_t)s_pix0[k] * 3, 3u * s_npix[k], ps.samples, BLOCK);
        }
        mark(5);
        if (kProf && s_exhausted) { // diagnostic build: generations after the last slice was taken, and how many rays they held
            if (t_drain == 0) t_drain = clock64();
            if (threadIdx.x == 0) { sect[7] += 1; prof.lanes[5] += total; }
        }
        if (last) break; // (every slice still open had no path left: summed just now or a generation ago)
        summed = closing;
        cur ^= 1u;
    }
    if (kProf && t_drain) sect[6] += (unsigned long long)(clock64() - t_drain);
    if (kProf) {
        for (int k = 0; k < 6; ++k) {
            const unsigned long long a = wave_sum_u32(prof.lanes[k]), b2 = wave_sum_u32(prof.slots[k]);
            if (lane == 0) { atomicAdd(&cnt->lanes[k], a); atomicAdd(&cnt->slots[k], b2); }
        }
        if (lane == 0)
            for (int k = 0; k < 8; ++k) atomicAdd(&cnt->sect[k], sect[k]);
    }
    rays = wave_sum_u32(my_rays);
    if (lane == 0 && rays) atomicAdd(&cnt->rays, rays);
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

// The k_paths_ring instance for a launch shape (pt_launch.h).
template <bool COUNT>
static void launch_ring(int kind, Counters *cnt, const SceneDev &scene, hipStream_t stream, int block, bool split, uint32_t grid, size_t lds_bytes, const PassDev &ps,
                        const RingDev &rg, float *d_rgb_sum)
{
#define PT_LAUNCH_RING(B, K, SP, FL) hipLaunchKernelGGL((k_paths_ring<COUNT, true, B, K, SP, FL>), dim3(grid), dim3(B), lds_bytes, stream, scene, ps, rg, d_rgb_sum, cnt)
#define PT_LAUNCH_RING_B(K, SP, FL)              \
    do {                                         \
        if (block == 256) PT_LAUNCH_RING(256, K, SP, FL); \
        else PT_LAUNCH_RING(512, K, SP, FL);     \
    } while (0)
    if (!COUNT && scene.root_ref < 0 && kind != PT_KIND_SPHERES) { // a scene of one leaf: the kernels without tree code
        if (kind == PT_KIND_ANY) PT_LAUNCH_RING_B(PT_KIND_ANY, false, true);
        else PT_LAUNCH_RING_B(PT_KIND_NO_CONES, false, true);
    } else if (kind == PT_KIND_SPHERES) {
        if (split) PT_LAUNCH_RING_B(PT_KIND_SPHERES, true, false);
        else PT_LAUNCH_RING_B(PT_KIND_SPHERES, false, false);
    } else if (kind == PT_KIND_NO_CONES) {
        PT_LAUNCH_RING_B(PT_KIND_NO_CONES, false, false);
    } else {
        PT_LAUNCH_RING_B(PT_KIND_ANY, false, false);
    }
#undef PT_LAUNCH_RING_B
#undef PT_LAUNCH_RING
}

void pt_launch_ring(bool count, int kind, Counters *cnt, const SceneDev &scene, hipStream_t stream, int block, bool split, uint32_t grid, size_t lds_bytes,
                    const PassDev &ps, const RingDev &rg, float *d_rgb_sum)
{
    if (count) launch_ring<true>(kind, cnt, scene, stream, block, split, grid, lds_bytes, ps, rg, d_rgb_sum);
    else launch_ring<false>(kind, cnt, scene, stream, block, split, grid, lds_bytes, ps, rg, d_rgb_sum);
}
