// pt_kernels_stream.hip -- k_paths_stream: persistent lanes for scenes traversed from HBM.
#include "pt_launch.h"

// ---------------------------------------------------------------------------------------------
// k_paths_stream: the slice pipeline for scenes traversed from HBM, where every traversal step waits on
// memory and rays differ widely in cost (1..5000 node visits on the 1M-sphere scene).  In k_paths a wave
// traces 64 rays and waits for the slowest of them, and a workgroup waits at the end of every bounce for
// its slowest ray: there 13 % of the lanes of a node visit are active, 60 % of the waves' time is spent at
// that barrier, and a slice's critical path is the SUM over its bounces of the longest ray.  Here a lane
// keeps a path from its camera ray to its end, and nothing is synchronised before the slice is finished:
//  * a traversal step is the same for every lane whatever it holds: fetch the 64 bytes it needs next --
//    an inner node, or the next two objects of its leaf -- then act on them; one memory round trip
//    serves all lanes;
//  * lanes whose traversal has finished wait until `serve_min` of them have (or nothing else is left to
//    do); then they are shaded together -- the path scatters and the lane starts the next traversal, or
//    the path ends and the lane takes the slice's next path (camera ray made on the spot).
// There are no ray queues.  Same paths, same arithmetic, same film order: images are bit-identical to
// k_paths.  (The vector ALUs are mostly idle on these scenes, so the lower lane utilisation of shading
// in small groups costs nothing that matters.)
// ---------------------------------------------------------------------------------------------
template <bool COUNT, int BLOCK, int KIND, bool WIDE>
__global__ __launch_bounds__(BLOCK, PT_STREAM_WAVES) void k_paths_stream(SceneDev sc, PassDev ps, BandDev band, float *__restrict__ rad /* [samples][band pixels][3] of this pass */,
                                                                           uint32_t *__restrict__ ticket, Counters *cnt,
                                                                           uint32_t serve_min /* lanes waiting for shading or a new path before they are served */, uint32_t phase_bias /* 0: node and leaf steps in every iteration */,
                                                                           unsigned long long *dbg = nullptr /* diagnostic build only */, uint32_t dbg_waves = 0)
{
    constexpr bool CONES = KIND == PT_KIND_ANY, SPH = KIND == PT_KIND_SPHERES;
    extern __shared__ float4 s_mem[];
    uint32_t *const stack = reinterpret_cast<uint32_t *>(s_mem) + threadIdx.x;
    if (kProf && dbg && (threadIdx.x & 63u) == 0) dbg[2 * (blockIdx.x * (BLOCK / 64) + threadIdx.x / 64)] = wall_clock64(); // diagnostic build: each wave's start
    GlobalSrc gsrc{sc.nodes, sc.prims, sc.cone};
    const float t_min = 0.001f /* main.cu:54 */, t_max = PT_FLT_MAX /* commons.h:11 */;
    const uint32_t lane = threadIdx.x & 63u;
    unsigned long long rays = 0;
    uint32_t nb = 0, np = 0, nf = 0, nr = 0;
    // Paths are numbered tile by tile (`tile` neighbouring pixels) and within a tile sample by sample.  A
    // wave owns a run of `run` consecutive paths at a time, fills its free lanes from it and takes the next
    // run of the pass from one counter when it is used up; nothing else is shared between waves, and
    // nothing waits for a run to finish.  The lanes of a wave therefore work on the samples of the same few
    // pixels, whose rays meet the same part of the tree: with lanes filled from one pass-wide sequence
    // instead, the 1M-sphere scene ran 25 % slower.
    uint32_t dbg_iter = 0, dbg_steps = 0, dbg_maxsteps = 0, dbg_serves = 0; // diagnostic build: loop iterations, steps of the current / longest ray, serves
    ProfLanes prof; // diagnostic build: issues and active lanes of 0 four-wide steps, 1 two-child steps, 2 leaf steps, 3 shading, 4 new paths
    const uint32_t npix = band.npix;
    const uint32_t n_paths = npix * ps.samples;
    const uint32_t tile_paths = band.tile * ps.samples;
    const uint32_t n_runs = (n_paths + band.run - 1) / band.run;
    uint32_t run_next = 0, run_end = 0; // wave-uniform: what is left of the wave's run
    // pixel within the band and sample within the pass of path p
    auto path_of = [&](uint32_t p, uint32_t &pl, uint32_t &s) {
        const uint32_t t = p / tile_paths, rem = p - t * tile_paths;
        const uint32_t tp = min(band.tile, npix - t * band.tile); // the band's last tile may be short
        s = rem / tp;
        pl = t * band.tile + (rem - s * tp);
    };
    {
        uint32_t lid = PT_NO_PRIM;          // where this lane's path reports: sample * npix + pixel within the band
        int32_t bounce = 0;
        v3 T = V(1.0f, 1.0f, 1.0f);         // throughput of the path so far
        int32_t node = PT_TRAVERSAL_DONE;   // inner node (>= 0), the rest of a leaf (< 0), or no traversal under way
        uint32_t sp = 0;
        Ray r;
        r.o = V(0.0f, 0.0f, 0.0f);
        r.d = V(1.0f, 1.0f, 1.0f);
        BoxRay br = make_box_ray(r, sc.cone.kappa_max);
        Nearest nh;
        nh.reset(t_max);
        bool by_list = false;   // this ray is answered by the list scan: non-finite ray, a hit at t = NaN, or its stack ran out (cannot happen)
        bool exhausted = false; // wave-uniform: the pass has no unstarted path left
        auto start_ray = [&]() {
            br = make_box_ray(r, ray_kappa(sc.cone, r.o));
            nh.reset(t_max);
            by_list = false;
            stack[0] = (uint32_t)PT_TRAVERSAL_DONE; // the bottom entry says "done": a pop needs no test for an empty stack
            sp = 1;
            node = 0; // root
            rays++;
            // A ray with a NaN or infinite component (the reference accepts hits whose t is NaN, and the
            // path goes on from a NaN point): the box test cannot prune for it, and what the scan accepts
            // along it depends on list order throughout.  The list itself answers it.
            if (!(fabsf(r.o.x) + fabsf(r.o.y) + fabsf(r.o.z) + fabsf(r.d.x) + fabsf(r.d.y) + fabsf(r.d.z) < PT_FLT_MAX)) {
                by_list = true;
                node = PT_TRAVERSAL_DONE;
            }
        };
        for (;;) {
            if (kProf) dbg_iter++;
            const bool resting = node == PT_TRAVERSAL_DONE;
            const bool to_shade = resting && lid != PT_NO_PRIM;
            const bool empty = lid == PT_NO_PRIM;
            const uint32_t n_shade = (uint32_t)__popcll(__ballot(to_shade));
            const uint32_t n_empty = exhausted ? 0u : (uint32_t)__popcll(__ballot(empty));
            const bool nobody_traverses = __ballot(!resting) == 0ull;
            if (n_shade + n_empty >= serve_min || nobody_traverses) {
                if (kProf) dbg_serves++;
                if (kProf && to_shade) {
                    if (dbg && dbg_steps > 50000u) { // one of the very long rays, for the report
                        float *f = reinterpret_cast<float *>(dbg + 4 * dbg_waves);
                        f[0] = r.o.x; f[1] = r.o.y; f[2] = r.o.z; f[3] = r.d.x; f[4] = r.d.y; f[5] = r.d.z; f[6] = nh.best_t; f[7] = (float)dbg_steps;
                        f[8] = (float)bounce; f[9] = __uint_as_float(nh.best_slot); f[10] = ray_kappa(sc.cone, r.o); f[11] = br.eps;
                    }
                    dbg_maxsteps = max(dbg_maxsteps, dbg_steps); dbg_steps = 0;
                }
                HitResult h;
                h.t = nh.best_t;
                h.slot = nh.best_slot;
                bool scan_list = false;
                uint32_t scan_from = 0; // the list entry a scan starts at
                bool fresh = false; // this lane leaves the serve with a new ray (scattered, or the first of a new path)
                if (to_shade) {
                    bool need_exact = by_list;
                    // Some cone's leading coefficient is exactly 0 for this direction (cone_table_hit): the scan accepts that cone
                    // at t = NaN wherever it stands, and its answer is the scan's from that entry on -- the nearest of the later
                    // entries by the scan's own rules, or the cone itself at t = NaN when the ray's line meets none of them.
                    // (A traversal that met such a cone itself and stopped at the NaN, by_list, is answered the same way.)
                    // Or some cone's coefficient has all but cancelled and its roots, which are rounding noise, pass its tests:
                    // then the list itself answers (from the start, or from the cone of the first case on).
                    uint32_t near_lo = 0, near_hi = 0;
                    const uint32_t found = CONES && sc.cone_tab ? cone_table_hit(sc.cone_tab, sc.cone_hdr, r.d, scan_from, near_lo, near_hi) : 0u;
                    const bool near_ok = (found & 2u) && cone_near_acceptable(sc.cone_tab, sc.prims_list, r.o.x, r.o.y, r.o.z, r.d.x, r.d.y, r.d.z, t_min, t_max, near_lo, near_hi);
                    const bool from_cone = (found & 1u) && !near_ok;
                    if (!(found & 1u)) scan_from = 0;
                    if (near_ok) need_exact = true;
                    else if (!from_cone) settle_parked<CONES>(nh, need_exact);
                    h.t = nh.best_t;
                    h.slot = nh.best_slot;
                    if (from_cone || need_exact) {
                        if (CONES && !near_ok && (from_cone || !by_list)) {
                            h = bvh_exact_scan<BLOCK>(gsrc, sc.stack_cap, r, br, t_min, t_max, stack, need_exact, from_cone ? scan_from + 1u : 0u);
                            nr++;
                            if (from_cone && !need_exact && h.slot == PT_NO_PRIM) {
                                h.t = __builtin_nanf("");
                                h.slot = scan_from | 0x80000000u;
                            }
                        }
                        scan_list = need_exact; // by_list, a nearly cancelled cone, or more candidates than the replay keeps
                    }
                    // A ray that is NaN in every component (what follows a hit at t = NaN: the point, and with it the scattered
                    // direction) is accepted by every entry at t = NaN: none of the reference's tests can reject it, so the
                    // scan ends on the list's last entry.  No need to run it.
                    if (scan_list && r.o.x != r.o.x && r.o.y != r.o.y && r.o.z != r.o.z && r.d.x != r.d.x && r.d.y != r.d.y && r.d.z != r.d.z) {
                        scan_list = false;
                        h.t = r.o.x;
                        h.slot = (sc.n_prims - 1u) | 0x80000000u;
                    }
                }
                // the list itself, one such ray at a time by the whole wave
                for (unsigned long long m = __ballot(scan_list); m; m &= m - 1ull) {
                    const int j = __ffsll((long long)m) - 1;
                    const float jox = __shfl(r.o.x, j, 64), joy = __shfl(r.o.y, j, 64), joz = __shfl(r.o.z, j, 64);
                    const float jdx = __shfl(r.d.x, j, 64), jdy = __shfl(r.d.y, j, 64), jdz = __shfl(r.d.z, j, 64);
                    const uint32_t jfrom = CONES ? (uint32_t)__shfl((int)scan_from, j, 64) : 0u;
                    const unsigned long long hj = CONES ? wave_list_scan_call(sc.prims_list, jfrom, sc.n_prims, jox, joy, joz, jdx, jdy, jdz, t_min, t_max)
                                                        : wave_list_scan(sc.prims_list, 0u, sc.n_prims, jox, joy, joz, jdx, jdy, jdz, t_min, t_max);
                    if (lane == (uint32_t)j) {
                        h.t = __uint_as_float((uint32_t)hj);
                        h.slot = (uint32_t)(hj >> 32);
                        if (h.slot != PT_NO_PRIM) h.slot |= 0x80000000u;
                        nf++;
                    }
                }
                if (to_shade) { // main.cu:54-64 for the ray just traced
                    prof.tick(3);
                    bool go_on = false;
                    v3 result;
                    if (h.slot == PT_NO_PRIM) {
                        result = vmul(T, V(sc.bg[0], sc.bg[1], sc.bg[2])); // main.cu:54-55
                    } else {
                        const PrimRec q = load_prim((h.slot & 0x80000000u) ? sc.prims_list : sc.prims, h.slot & 0x7FFFFFFFu);
                        const uint32_t s = lid / npix, pl = lid - s * npix;
                        Ray nray;
                        v3 nT;
                        go_on = shade_hit<SPH>(sc.mats, ps, r, T, band.pix0 + pl, ps.sample_base + s, bounce, q, h.t, nray, nT, result);
                        if (go_on) {
                            r = nray;
                            T = nT;
                            bounce++;
                            if (CONES) start_ray(); else fresh = true;
                        }
                    }
                    if (!go_on) {
                        rad[3 * (size_t)lid + 0] = result.x;
                        rad[3 * (size_t)lid + 1] = result.y;
                        rad[3 * (size_t)lid + 2] = result.z;
                        lid = PT_NO_PRIM;
                    }
                }
                if (!exhausted) { // lanes without a path take the next ones of the wave's run
                    const bool need = lid == PT_NO_PRIM;
                    const unsigned long long m = __ballot(need);
                    if (m) {
                        if (run_next == run_end) { // the wave's run is used up: the next one of the pass
                            uint32_t g = 0;
                            if (lane == (uint32_t)(__ffsll((long long)m) - 1)) g = atomicAdd(ticket, 1u);
                            g = (uint32_t)__builtin_amdgcn_readfirstlane((int)__shfl(g, __ffsll((long long)m) - 1, 64));
                            if (g >= n_runs) {
                                exhausted = true;
                            } else {
                                run_next = g * band.run;
                                run_end = min(run_next + band.run, n_paths);
                            }
                        }
                        const uint32_t k = lane_prefix(m);
                        if (need && k < run_end - run_next) {
                            prof.tick(4);
                            uint32_t s, pl;
                            path_of(run_next + k, pl, s);
                            lid = s * npix + pl;
                            r = camera_ray(sc.cam, ps, band.pix0 + pl, ps.sample_base + s); // main.cu:88-92
                            T = V(1.0f, 1.0f, 1.0f);
                            bounce = 0;
                            if (CONES) start_ray(); else fresh = true;
                        }
                        run_next += min((uint32_t)__popcll(m), run_end - run_next); // (lanes left without a path come back at once)
                    }
                }
                // once for both kinds of new ray: the slab constants are a sixth of a serve's instructions (1M spheres +4.6 %).  The
                // cone variant has no register for the flag across the serve and keeps the two sites (0.7 % slower merged).
                if (!CONES && fresh) start_ray();
                if (__ballot(lid != PT_NO_PRIM) == 0ull) break; // no path left in this wave, none to take
                continue;
            }
            // Traversal steps follow each other in a loop of their own until a serve is due (the same moment the test at the
            // top of the outer loop would find): the outer loop merges the state a serve writes -- ray, slab constants,
            // nearest hit, parked cone -- with the state a step writes, a dozen register moves per iteration that the inner
            // loop does not pay (100k mixed objects: 31 % fewer vector instructions, +15 %; 1M spheres: 10 % fewer, +2 %
            // at four waves per SIMD, where the loop fits the registers without spilling).
            // The record a lane needs next is asked for as soon as its reference is known -- at the end of the step that
            // produced it, before the votes and tests of the next iteration -- and waits in pw0..pw3 for the lane's turn
            // (100k mixed objects +6.6 %).
            float4 pw0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), pw1 = pw0, pw2 = pw0, pw3 = pw0;
            auto fetch = [&]() {
                const float4 *rec = node >= 0 ? (WIDE ? sc.wide + 4 * (size_t)node : reinterpret_cast<const float4 *>(sc.nodes + node))
                                              : reinterpret_cast<const float4 *>(sc.prims + ((uint32_t)~node >> 3));
                pw0 = rec[0]; pw1 = rec[1]; pw2 = rec[2]; pw3 = rec[3]; // (the object array is padded by one record)
            };
            if (node != PT_TRAVERSAL_DONE) fetch();
#pragma nounroll
            for (bool first_step = true;; first_step = false) {
            if (!first_step) {
                const unsigned long long rest = __ballot(node == PT_TRAVERSAL_DONE && (!exhausted || lid != PT_NO_PRIM));
                if ((uint32_t)__popcll(rest) >= serve_min || __ballot(node != PT_TRAVERSAL_DONE) == 0ull) break;
            }
            // One kind of step per iteration, the one more lanes are waiting for: the two kinds share no code, and
            // run in the same iteration each would execute with the lanes of the other masked off.
            bool my_turn = true;
            if (phase_bias) {
                const uint32_t n_inner = (uint32_t)__popcll(__ballot(node >= 0)), n_leaf = (uint32_t)__popcll(__ballot(node < 0 && node != PT_TRAVERSAL_DONE));
                const bool inner_turn = n_inner * phase_bias >= n_leaf * 4u; // phase_bias = 4: plain majority
                my_turn = (node >= 0) == inner_turn;
            }
            if (node != PT_TRAVERSAL_DONE && my_turn) {
                if (kProf) dbg_steps++;
                const bool inner = node >= 0;
                const uint32_t ref = (uint32_t)~node; // leaf: (first object << 3) | (objects left - 1)
                const float4 w0 = pw0, w1 = pw1, w2 = pw2, w3 = pw3;
                if (WIDE && inner) {
                    prof.tick(0);
                    // pt::WideNode: origin, step exponents, the four children's planes as bytes per axis, their references
                    const uint32_t eb = __float_as_uint(w0.w);
                    const float stx = __uint_as_float((eb & 0xFFu) << 23), sty = __uint_as_float((eb & 0xFF00u) << 15), stz = __uint_as_float((eb & 0xFF0000u) << 7);
                    const uint32_t lox = __float_as_uint(w1.x), loy = __float_as_uint(w1.y), loz = __float_as_uint(w1.z);
                    const uint32_t hix = __float_as_uint(w1.w), hiy = __float_as_uint(w2.x), hiz = __float_as_uint(w2.y);
                    // the planes the ray meets first / last on each axis, by its direction signs
                    const uint32_t qnx = br.sx ? hix : lox, qfx = br.sx ? lox : hix;
                    const uint32_t qny = br.sy ? hiy : loy, qfy = br.sy ? loy : hiy;
                    const uint32_t qnz = br.sz ? hiz : loz, qfz = br.sz ? loz : hiz;
                    uint32_t key[4]; // distance at which the ray's cone enters the child's box (its low two bits give way to the child's slot)
                    int32_t cr[4] = {__float_as_int(w2.z), __float_as_int(w2.w), __float_as_int(w3.x), __float_as_int(w3.y)};
#ifdef PT_WIDE_FOLD
                    // The node's grid folded into the slab constants: a plane distance is (origin + q step) A + C = q (step A) + (origin A + C),
                    // one FMA per plane after twelve operations per node instead of two per plane.  step is a power of two, so step A is
                    // exact; origin A + C is rounded once more than in the unfolded form, by at most 2^-24 of its own magnitude (which the
                    // plane distance itself need not share: the node's origin can be far from a plane on the ray's scale), and four times
                    // that much is added to the allowance of the comparison.  The far constants of an axis are its near ones times
                    // (1 + w) / (1 - w) (make_box_ray), so the largest near term and the ray's largest such ratio (eps_far) bound them all;
                    // an axis whose far slot is open has the constants (0, 3e38) and no rounding.
                    const float sanx = stx * br.anx, sany = sty * br.any, sanz = stz * br.anz, safx = stx * br.afx, safy = sty * br.afy, safz = stz * br.afz;
                    const float onx = __builtin_fmaf(w0.x, br.anx, br.cnx), ony = __builtin_fmaf(w0.y, br.any, br.cny), onz = __builtin_fmaf(w0.z, br.anz, br.cnz);
                    const float ofx = __builtin_fmaf(w0.x, br.afx, br.cfx), ofy = __builtin_fmaf(w0.y, br.afy, br.cfy), ofz = __builtin_fmaf(w0.z, br.afz, br.cfz);
                    float omax;
                    asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(omax) : "v"(onx), "v"(ony), "v"(onz));
                    const float eps_node = __builtin_fmaf(omax, br.eps_far, br.eps);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float nx = __builtin_fmaf((float)((qnx >> (8 * k)) & 0xFFu), sanx, onx), fx = __builtin_fmaf((float)((qfx >> (8 * k)) & 0xFFu), safx, ofx);
                        const float ny = __builtin_fmaf((float)((qny >> (8 * k)) & 0xFFu), sany, ony), fy = __builtin_fmaf((float)((qfy >> (8 * k)) & 0xFFu), safy, ofy);
                        const float nz = __builtin_fmaf((float)((qnz >> (8 * k)) & 0xFFu), sanz, onz), fz = __builtin_fmaf((float)((qfz >> (8 * k)) & 0xFFu), safz, ofz);
                        float m, tn, tf;
                        asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(nz), "v"(t_min));
                        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(tn) : "v"(nx), "v"(ny), "v"(m));
                        asm("v_min_f32 %0, %1, %2" : "=v"(m) : "v"(fz), "v"(nh.best_t));
                        asm("v_min3_f32 %0, %1, %2, %3" : "=v"(tf) : "v"(fx), "v"(fy), "v"(m));
                        const bool hit = tn <= tf + eps_node && cr[k] != pt::kWideAbsent;
                        key[k] = ((__float_as_uint(tn) & ~3u) | (uint32_t)k) | (hit ? 0u : 0xFFFFFFFFu); // tn >= t_min > 0: its bits order as it does
                    }
#else
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        // plane = origin + q * step exactly (the builder's grid guarantees it), then the slab test of box_test2
                        const float pnx = __builtin_fmaf((float)((qnx >> (8 * k)) & 0xFFu), stx, w0.x), pfx = __builtin_fmaf((float)((qfx >> (8 * k)) & 0xFFu), stx, w0.x);
                        const float pny = __builtin_fmaf((float)((qny >> (8 * k)) & 0xFFu), sty, w0.y), pfy = __builtin_fmaf((float)((qfy >> (8 * k)) & 0xFFu), sty, w0.y);
                        const float pnz = __builtin_fmaf((float)((qnz >> (8 * k)) & 0xFFu), stz, w0.z), pfz = __builtin_fmaf((float)((qfz >> (8 * k)) & 0xFFu), stz, w0.z);
                        const float nx = __builtin_fmaf(pnx, br.anx, br.cnx), ny = __builtin_fmaf(pny, br.any, br.cny), nz = __builtin_fmaf(pnz, br.anz, br.cnz);
                        const float fx = __builtin_fmaf(pfx, br.afx, br.cfx), fy = __builtin_fmaf(pfy, br.afy, br.cfy), fz = __builtin_fmaf(pfz, br.afz, br.cfz);
                        float m, tn, tf;
                        asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(nz), "v"(t_min));
                        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(tn) : "v"(nx), "v"(ny), "v"(m));
                        asm("v_min_f32 %0, %1, %2" : "=v"(m) : "v"(fz), "v"(nh.best_t));
                        asm("v_min3_f32 %0, %1, %2, %3" : "=v"(tf) : "v"(fx), "v"(fy), "v"(m));
                        const bool hit = tn <= tf + br.eps && cr[k] != pt::kWideAbsent;
                        key[k] = ((__float_as_uint(tn) & ~3u) | (uint32_t)k) | (hit ? 0u : 0xFFFFFFFFu); // tn >= t_min > 0: its bits order as it does
                    }
#endif
                    if (COUNT) nb += 4;
                    if (br.afx == 0.0f || br.afy == 0.0f || br.afz == 0.0f) { // a steep axis (rare): the bound its open far slot leaves out
                        const SteepBounds sb = make_steep_bounds(r, ray_kappa(sc.cone, r.o));
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float pfx = __builtin_fmaf((float)((qfx >> (8 * k)) & 0xFFu), stx, w0.x);
                            const float pfy = __builtin_fmaf((float)((qfy >> (8 * k)) & 0xFFu), sty, w0.y);
                            const float pfz = __builtin_fmaf((float)((qfz >> (8 * k)) & 0xFFu), stz, w0.z);
                            const float tf = fminf(fminf(__builtin_fmaf(pfx, br.afx, br.cfx), __builtin_fmaf(pfy, br.afy, br.cfy)),
                                                   fminf(__builtin_fmaf(pfz, br.afz, br.cfz), nh.best_t));
                            if (steep_rejects(sb, pfx, pfy, pfz, tf, br.eps)) key[k] = 0xFFFFFFFFu;
                        }
                    }
                    // nearest first: a five-exchange sorting network on the keys, the children's references moving with them
                    // (selecting a reference by a sorted key's slot bits afterwards costs more than the ten selects here)
                    auto order2 = [](uint32_t &a, uint32_t &b, int32_t &ra, int32_t &rb) {
                        const bool s = b < a;
                        const uint32_t lo = s ? b : a, hi = s ? a : b;
                        const int32_t rlo = s ? rb : ra, rhi = s ? ra : rb;
                        a = lo; b = hi; ra = rlo; rb = rhi;
                    };
                    order2(key[0], key[1], cr[0], cr[1]); order2(key[2], key[3], cr[2], cr[3]); order2(key[0], key[2], cr[0], cr[2]);
                    order2(key[1], key[3], cr[1], cr[3]); order2(key[1], key[2], cr[1], cr[2]);
                    const uint32_t n_hit = (key[0] != 0xFFFFFFFFu) + (key[1] != 0xFFFFFFFFu) + (key[2] != 0xFFFFFFFFu) + (key[3] != 0xFFFFFFFFu);
                    if (n_hit == 0) {
                        node = (int32_t)stack[(--sp) * BLOCK];
                    } else {
                        // (against the constant, not sc.wide_stack_cap: a stack of fewer entries belongs to a tree too shallow to fill it, three
                        // entries a level, and the compiler re-reads the kernel argument here with a scalar load and a wait in every step)
                        if (sp + n_hit - 1u > PT_WIDE_STACK_MAX) {
                            by_list = true; // the stack cannot take them (deeper than 31 entries): the list answers this ray
                            node = PT_TRAVERSAL_DONE;
                        } else {
                            if (n_hit > 3) stack[(sp++) * BLOCK] = (uint32_t)cr[3];
                            if (n_hit > 2) stack[(sp++) * BLOCK] = (uint32_t)cr[2];
                            if (n_hit > 1) stack[(sp++) * BLOCK] = (uint32_t)cr[1];
                            node = cr[0];
                        }
                    }
                } else if (inner) {
                    prof.tick(1);
                    float2 nx, ny, nz, fx, fy, fz;
                    int32_t c0, c1;
                    planes_of_record(w0, w1, w2, w3, br, nx, ny, nz, fx, fy, fz, c0, c1);
                    float tn0, tn1;
                    bool h0, h1;
                    box_test2(nx, ny, nz, fx, fy, fz, br, t_min, nh.best_t, h0, h1, tn0, tn1);
                    if (COUNT) nb += 2;
                    if (br.afx == 0.0f || br.afy == 0.0f || br.afz == 0.0f) { // a steep axis (rare): the bound its open far slot leaves out
                        const SteepBounds sb = make_steep_bounds(r, ray_kappa(sc.cone, r.o));
                        const float tf0 = fminf(fminf(__builtin_fmaf(fx.x, br.afx, br.cfx), __builtin_fmaf(fy.x, br.afy, br.cfy)), fminf(__builtin_fmaf(fz.x, br.afz, br.cfz), nh.best_t));
                        const float tf1 = fminf(fminf(__builtin_fmaf(fx.y, br.afx, br.cfx), __builtin_fmaf(fy.y, br.afy, br.cfy)), fminf(__builtin_fmaf(fz.y, br.afz, br.cfz), nh.best_t));
                        if (steep_rejects(sb, fx.x, fy.x, fz.x, tf0, br.eps)) h0 = false;
                        if (steep_rejects(sb, fx.y, fy.y, fz.y, tf1, br.eps)) h1 = false;
                    }
                    // the next child by selects, one masked store, one masked load (as in bvh_nearest; the stack holds the
                    // tree's depth + 2 entries, one per level of a descent and the bottom one)
                    const bool both = h0 && h1;
                    const bool pick1 = (both && tn1 < tn0) || (!h0 && h1);
                    node = pick1 ? c1 : c0;
                    if (both) stack[(sp++) * BLOCK] = (uint32_t)(pick1 ? c0 : c1);
                    if (!(h0 || h1)) node = (int32_t)stack[(--sp) * BLOCK];
                } else {
                    prof.tick(2);
                    const uint32_t first = ref >> 3, left = (ref & 7u) + 1u;
                    PrimRec q;
                    q.p[0] = w0.x; q.p[1] = w0.y; q.p[2] = w0.z; q.p[3] = w0.w; q.p[4] = w1.x; q.p[5] = w1.y;
                    q.meta = __float_as_uint(w1.z);
                    q.id = __float_as_uint(w1.w);
                    if (CONES) { // mixed scenes: one code path per family of objects (and no registers to spare for the split below: 3 % slower with it)
                        object_test_fused(q, first, r, t_min, t_max, nh);
                        if (left > 1) {
                            q.p[0] = w2.x; q.p[1] = w2.y; q.p[2] = w2.z; q.p[3] = w2.w; q.p[4] = w3.x; q.p[5] = w3.y;
                            q.meta = __float_as_uint(w3.z);
                            q.id = __float_as_uint(w3.w);
                            object_test_fused(q, first + 1u, r, t_min, t_max, nh);
                        }
                    } else {
                        PrimRec q2;
                        q2.p[0] = w2.x; q2.p[1] = w2.y; q2.p[2] = w2.z; q2.p[3] = w2.w; q2.p[4] = w3.x; q2.p[5] = w3.y;
                        q2.meta = __float_as_uint(w3.z);
                        q2.id = __float_as_uint(w3.w);
                        // Spheres first, in two halves (sphere.h:85-101): the discriminants of both, then the roots -- a
                        // correctly rounded sqrt and divides, the expensive part -- once for every lane that has a candidate,
                        // and once more only where a lane has two.  (The answer does not depend on the order in which the
                        // objects of a leaf are tried: equal t goes to the later list entry either way.)
                        const bool sph1 = SPH || (q.meta & 7u) == PT_SPHERE, sph2 = left > 1 && (SPH || (q2.meta & 7u) == PT_SPHERE);
                        float hb1 = 0.0f, disc1 = 0.0f, hb2 = 0.0f, disc2 = 0.0f;
                        const bool cand1 = sph1 && sphere_discriminant(q, r, hb1, disc1);
                        const bool cand2 = sph2 && sphere_discriminant(q2, r, hb2, disc2);
                        if (cand1 && cand2) {
                            float t = 0.0f;
                            const bool h = sphere_roots(r, hb1, disc1, t_min, nh.best_t, t);
                            nh.take(h, t, first, q.id);
                        }
                        if (cand1 || cand2) {
                            float t = 0.0f;
                            const bool h = sphere_roots(r, cand2 ? hb2 : hb1, cand2 ? disc2 : disc1, t_min, nh.best_t, t);
                            nh.take(h, t, cand2 ? first + 1u : first, cand2 ? q2.id : q.id);
                        }
                        if (!SPH) {
                            if (!sph1) object_test<CONES>(q, first, r, t_min, t_max, nh);
                            if (left > 1 && !sph2) object_test<CONES>(q2, first + 1u, r, t_min, t_max, nh);
                        }
                    }
                    if (COUNT) np += left > 1 ? 2u : 1u;
                    if (nh.best_t != nh.best_t) { // a hit at t = NaN was accepted: from here on the scan's answer is a matter of list order
                        by_list = true;
                        node = PT_TRAVERSAL_DONE;
                    } else if (left > 2) node = ~(int32_t)(((first + 2u) << 3) | (left - 3u)); // the rest of the leaf
                    else node = (int32_t)stack[(--sp) * BLOCK];
                }
                if (node != PT_TRAVERSAL_DONE) fetch();
            }
            }
        }
    }
    if (kProf && dbg) {
        uint32_t mx = dbg_maxsteps;
        for (int off = 32; off > 0; off >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, off, 64));
        if (lane == 0) {
            const uint32_t w = blockIdx.x * (BLOCK / 64) + threadIdx.x / 64;
            dbg[2 * w + 1] = wall_clock64();
            const size_t nw = dbg_waves;
            dbg[2 * nw + 2 * w] = ((unsigned long long)dbg_iter << 32) | dbg_serves;
            dbg[2 * nw + 2 * w + 1] = ((unsigned long long)mx << 32) | (uint32_t)rays;
        }
    }
    if (kProf) {
        for (int k = 0; k < 5; ++k) {
            const unsigned long long a = wave_sum_u32(prof.lanes[k]), b2 = wave_sum_u32(prof.slots[k]);
            if (lane == 0) { atomicAdd(&cnt->lanes[k], a); atomicAdd(&cnt->slots[k], b2); }
        }
        const unsigned long long it = wave_sum_u32(lane == 0 ? dbg_iter : 0u), sv = wave_sum_u32(lane == 0 ? dbg_serves : 0u);
        if (lane == 0) { atomicAdd(&cnt->sect[0], it); atomicAdd(&cnt->sect[1], sv); }
    }
    rays = wave_sum_u32((uint32_t)rays);
    if (lane == 0 && rays) atomicAdd(&cnt->rays, rays);
    if (COUNT) {
        unsigned long long sb = wave_sum_u32(nb), sp2 = wave_sum_u32(np);
        if (lane == 0) {
            atomicAdd(&cnt->box_tests, sb);
            atomicAdd(&cnt->prim_tests, sp2);
        }
    }
    if (nf) atomicAdd(&cnt->exact_fallbacks, (unsigned long long)nf);
    if (nr) atomicAdd(&cnt->exact_replays, (unsigned long long)nr);
}

// The k_paths_stream instance for a launch shape.
const void *pt_stream_kernel_for(bool count, int block, int kind, bool wide)
{
#define PT_STREAM_PTR(C, B, K, W) reinterpret_cast<const void *>(&k_paths_stream<C, B, K, W>)
#define PT_STREAM_W(C, B, K) (wide ? PT_STREAM_PTR(C, B, K, true) : PT_STREAM_PTR(C, B, K, false))
#define PT_STREAM_K(C, B) (kind == PT_KIND_ANY ? PT_STREAM_W(C, B, PT_KIND_ANY) : kind == PT_KIND_NO_CONES ? PT_STREAM_W(C, B, PT_KIND_NO_CONES) : PT_STREAM_W(C, B, PT_KIND_SPHERES))
#define PT_STREAM_B(C) (block == 256 ? PT_STREAM_K(C, 256) : PT_STREAM_K(C, 512))
    return count ? PT_STREAM_B(true) : PT_STREAM_B(false);
#undef PT_STREAM_B
#undef PT_STREAM_K
#undef PT_STREAM_W
#undef PT_STREAM_PTR
}

