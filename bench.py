#!/usr/bin/env python3
"""Benchmark of the hot path: camera ray -> nearest hit -> scatter/emit -> accumulate.

    python bench.py --gpus N --steps K --warmup W [--workload c2]

A step is one full render of the workload's image: each rank renders its own range of absolute
sample indices (--scaling strong, the default: the workload's spp cut into N contiguous ranges, so
the image is the named configuration's whatever N is; --scaling weak: `spp` samples per GPU, the
image gets N*spp), the linear accumulation buffers are summed
with one RCCL all-reduce (on its own stream, overlapped with the next step's render), and every rank
holds the result.  Inputs (scene, BVH) are resident in HBM before the timed region.  Prints ONE JSON
line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from pathtracer_amd import api, distributed, scenes  # noqa: E402

# The benchmark configurations of BASELINE.json.  B / P (mean box and object tests per ray) come from
# profiles/work_per_ray.json: measured once on the CPU by a plain stack traversal of a fixed reference tree (binned
# SAH, 16 bins, leaf <= 4) over the committed ray fixtures (profiles/freeze_work_per_ray.py), and frozen.
WORKLOADS = {
    "c1": dict(name="default scene() 400x225 16spp depth50", scene=scenes.default_scene, w=400, h=225, spp=16, depth=50),
    "c2": dict(name="random-spheres (485 spheres) 1920x1080 64spp depth50", scene=scenes.random_spheres_scene, w=1920, h=1080, spp=64, depth=50),
    "c3": dict(name="cornell box 1024x1024 1024spp depth50", scene=scenes.cornell_scene, w=1024, h=1024, spp=1024, depth=50),
    "c4": dict(name="mixed 100k objects 3840x2160 256spp depth50", scene=scenes.mixed_scene, w=3840, h=2160, spp=256, depth=50),
    "c5": dict(name="1M spheres 3840x2160 4096spp depth50", scene=scenes.sphere_slab_scene, w=3840, h=2160, spp=4096, depth=50),
}
HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak
VALU_PEAK_TFLOPS = 157.3    # fp32 vector peak: 256 CUs x 4 SIMDs x 64 FLOP/clk x 2.4 GHz (v_fma_f32 issues every 2 cycles per SIMD)


def work_per_ray(workload):
    with open(os.path.join(ROOT, "profiles", "work_per_ray.json")) as f:
        w = json.load(f)[workload]
    return float(w["B"]), float(w["P"])


def bytes_per_ray(B, P):
    return 32.0 * B + 32.0 * P + 100.0  # node record share + object record + path-record stream


def flops_per_ray(B, P):
    return 24.0 * B + 30.0 * P + 60.0


def cpu_baseline(scene_builder, depth, seed, W, H, SPP):
    """The CPU oracle (a port of the reference's per-object hit / linear scan / scatter code) on a
    bounded sample of the same workload: same scene, depth and stream, reduced image and spp."""
    from oracle import pt_oracle as O
    from tests import util

    osc = util.oracle_scene(scene_builder)
    shrink = min(1.0, 960.0 / W)  # C2: a quarter of the pixels at the workload's own spp, about 10 s on 16 cores
    w, h, spp = int(W * shrink), int(H * shrink), min(SPP, 64)
    n_prims = len(scene_builder.arrays()["prim_type"])
    if n_prims > 5000:  # the scan is O(objects) per ray: shrink the sample for the big scenes
        w, h, spp = 96, 54, 1
    threads = min(O.num_threads(), 64)
    t0 = time.time()
    _, rays = osc.render(w, h, spp, depth, seed=seed, threads=threads)
    dt = time.time() - t0
    out = {
        "value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
        "sample": "%dx%d %dspp depth%d of the same scene, linear scan as in the reference, %d rays in %.2f s" % (w, h, spp, depth, rays, dt),
    }
    # Beside it, the same walk with nearest hits found through the library's host-built tree on the CPU: the reference has
    # no hierarchy (its scan is O(objects) per ray), so this is the figure to hold the GPU's against if the algorithmic
    # O(N) -> O(log N) gain is not to be credited to the hardware.  Timing only, not a parity reference.
    nodes, order, _ = api.bvh_build_host(scene_builder.desc())
    bw, bh, bspp = (W, H, min(SPP, 2)) if n_prims > 5000 else (W, H, min(SPP, 64))
    t0 = time.time()
    _, brays = osc.render_bvh(nodes, order, bw, bh, bspp, depth, seed=seed, threads=threads)
    bdt = time.time() - t0
    out["bvh"] = {
        "value": brays / bdt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port with a tree (not the reference's algorithm)",
        "sample": "%dx%d %dspp depth%d of the same scene, stack traversal of the library's two-wide tree, %d rays in %.2f s" % (bw, bh, bspp, depth, brays, bdt),
    }
    return out


IN_FLIGHT_BELOW_MS = 8.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--seed", type=int, default=1984)
    ap.add_argument("--no-cpu-baseline", action="store_true", help="the timed region and the roofline only: no CPU baseline, no PCIe-inclusive and no two-in-flight figures (profiling runs)")
    ap.add_argument("--scaling", default=os.environ.get("PT_BENCH_SCALING", "strong"), choices=["weak", "strong"],
                    help="strong (default): the workload's spp is cut into N ranges, the image is the named configuration's; weak: every GPU renders the workload's spp (the image gets N x spp)")
    ap.add_argument("--spp", type=int, default=0, help="override the workload's samples per pixel (profiling runs; not the headline configuration)")
    ap.add_argument("--force-collective", action="store_true", help="with one GPU: still create the process group (one rank) and run the all-reduce on its stream, to rehearse the N > 1 code path")
    ap.add_argument("--in-flight", type=int, default=0, choices=[0, 1, 2],
                    help="steps in flight.  2: consecutive steps alternate between two contexts (each with its own workspace) on two streams, "
                         "so the last bounces of one step's persistent kernel (a few waves, most of the GPU idle) overlap the start of the next.  "
                         "0 (default): 2 when a step alone takes less than %g ms, where that tail is a large part of the step (C2 at 8 / 16 / 32 spp: "
                         "2.15 -> 1.76, 3.66 -> 3.35, 6.60 -> 6.16 ms per step), else 1 (C2 at 48 / 64 spp: 9.48 -> 9.78, 12.37 -> 12.80 ms)" % IN_FLIGHT_BELOW_MS)
    ap.add_argument("--cull", default="cone", choices=["cone", "geometric"], help="geometric: PT_RENDER_GEOMETRIC_CULL (labelled in the output; not the headline configuration)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs one process per GPU: launch with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    # stdout carries the one JSON line and nothing else: what libraries print there while the run lasts (RCCL's version
    # banner at the first collective) goes to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    dist = None
    if world > 1 or args.force_collective:
        import torch.distributed as dist

        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        # RCCL launches the reduction on the process group's own stream: ask for a high-priority one
        opts = None
        try:
            opts = dist.ProcessGroupNCCL.Options()
            opts.is_high_priority_stream = True
        except Exception:  # an older binding: the default stream priority then
            opts = None
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank), pg_options=opts)

    wl = WORKLOADS[args.workload]
    builder = wl["scene"]()
    ctx = api.Context(local_rank)
    ctx.upload(builder.desc())
    ctxs = [ctx]
    W, H, spp, depth = wl["w"], wl["h"], wl["spp"], wl["depth"]
    name = wl["name"]
    if args.spp > 0:
        spp = args.spp
        name += " [spp overridden to %d]" % spp
    base_flags = api.RENDER_GEOMETRIC_CULL if args.cull == "geometric" else 0
    if base_flags:
        name += " [boxes culled by exact geometry]"
    sample_offset, my_spp = distributed.shard(args.scaling, rank, world, spp)
    spp_total = spp * world if args.scaling == "weak" else spp
    # two accumulation buffers: the all-reduce of one step runs on its own stream while the next step renders
    accum = [torch.zeros((H, W, 3), dtype=torch.float32, device="cuda") for _ in range(2)]
    render_stream = torch.cuda.current_stream()
    render_streams = [render_stream]
    # high priority: when a render ends, the all-reduce of its buffer and the next render both become ready, and the
    # reduction's few workgroups should get their slots before the persistent render kernel takes every CU
    comm_stream = torch.cuda.Stream(priority=-1) if dist is not None else None
    params = api.Context.params(W, H, my_spp, depth, seed=args.seed, sample_offset=sample_offset, flags=base_flags)
    rendered = [torch.cuda.Event(), torch.cuda.Event()]
    reduced = [None, None]

    def step(k, want_stats=False, flags=0, ev=None):
        buf = accum[k & 1]
        rs = render_streams[k % len(render_streams)]
        cx = ctxs[k % len(ctxs)]
        with torch.cuda.stream(rs):
            if reduced[k & 1] is not None:
                rs.wait_event(reduced[k & 1])  # the buffer's previous all-reduce has finished
            buf.zero_()
            params.flags = base_flags | flags
            if ev is not None:
                ev[0].record(rs)
            st = cx.render_device(params, buf.data_ptr(), rs.cuda_stream, want_stats=want_stats)
            if ev is not None:
                ev[1].record(rs)
        if comm_stream is not None:  # linear sums; gamma comes after (main.cu:95-98 is non-linear)
            rendered[k & 1].record(rs)
            comm_stream.wait_event(rendered[k & 1])
            with torch.cuda.stream(comm_stream):
                distributed.all_reduce_sum(buf)
                reduced[k & 1] = torch.cuda.Event()
                reduced[k & 1].record(comm_stream)
        return st

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # Set-up, untimed: the ray count of this rank's sample range (deterministic per step), then one step alone to decide
    # how many steps to keep in flight (the slowest rank's time decides for all), then the second context's first step
    # (its workspace is allocated there).
    st0 = step(0, want_stats=True)
    sync()
    t_alone = time.perf_counter()
    step(1)
    sync()
    alone_t = torch.tensor([time.perf_counter() - t_alone], dtype=torch.float64, device="cuda")
    if dist is not None:
        dist.all_reduce(alone_t, op=dist.ReduceOp.MAX)
    ms_alone = alone_t.item() * 1e3
    in_flight = args.in_flight if args.in_flight > 0 else (2 if ms_alone < IN_FLIGHT_BELOW_MS else 1)
    if in_flight == 2:
        ctxs.append(api.Context(local_rank))
        ctxs[-1].upload(builder.desc())
        render_streams.append(torch.cuda.Stream())
        step(1)
        sync()
    for k in range(args.warmup):
        step(k)
    sync()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k, ev=evs[k])
    sync()
    elapsed = time.perf_counter() - t0
    ms_render = sum(a.elapsed_time(b) for a, b in evs) / max(1, args.steps)  # the render's launches, on the stream they run on

    rays_t = torch.tensor([float(st0["rays"])], dtype=torch.float64, device="cuda")
    el_t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    if dist is not None:
        dist.all_reduce(rays_t)
        dist.all_reduce(el_t, op=dist.ReduceOp.MAX)
    rays_step = rays_t.item()
    elapsed = el_t.item()

    # One more step with the library's own events around every launch of the dominant kernel (a cross-check of
    # ms_render, slightly slower), and a short counting pass for the live B / P of the library's own tree.
    prof = step(0, want_stats=True, flags=api.RENDER_STAGE_TIMES)
    sync()
    cnt = None
    if rank == 0:
        cp = api.Context.params(W, H, min(my_spp, 2 if args.workload in ("c4", "c5") else 4), depth, seed=args.seed, flags=base_flags | api.RENDER_COUNT_TESTS)
        accum[0].zero_()
        cnt = ctx.render_device(cp, accum[0].data_ptr(), render_stream.cuda_stream, want_stats=True)

    # The same K steps with two of them in flight: consecutive steps alternate between two contexts (each with its own
    # workspace) on two streams, so that the last rays of one step's persistent kernel -- a handful of paths fifty bounces
    # deep, most of the GPU idle -- overlap the start of the next.  What a caller that renders frame after frame gets;
    # reported beside the headline, never as it (the launches then overlap, and a launch's own duration exceeds the step's).
    pipelined = None
    if world == 1 and in_flight == 1 and not args.no_cpu_baseline and elapsed / args.steps < 0.5:
        ctxs.append(api.Context(local_rank))
        ctxs[-1].upload(builder.desc())
        render_streams.append(torch.cuda.Stream())
        for k in range(4):
            step(k)
        sync()
        n_extra = max(args.steps, 40)  # the two streams settle into their steady state after some ten steps
        t0p = time.perf_counter()
        for k in range(n_extra):
            step(k)
        sync()
        dtp = time.perf_counter() - t0p
        pipelined = {"steps_in_flight": 2, "steps": n_extra, "ms_per_step": dtp / n_extra * 1e3, "Mrays_per_s": rays_step * n_extra / dtp / 1e6,
                     "note": "two contexts on two streams, steps alternating; same work per step"}
        ctxs.pop().close()
        render_streams.pop()
        torch.cuda.synchronize()
    # ... and, when the headline keeps two in flight, the same K steps one at a time
    unpipelined = None
    if world == 1 and in_flight == 2 and not args.no_cpu_baseline:
        sync()
        spare_ctx, spare_stream = ctxs.pop(), render_streams.pop()
        step(0)
        sync()
        t0p = time.perf_counter()
        for k in range(args.steps):
            step(k)
        sync()
        dtp = time.perf_counter() - t0p
        unpipelined = {"steps_in_flight": 1, "ms_per_step": dtp / args.steps * 1e3, "Mrays_per_s": rays_step * args.steps / dtp / 1e6}
        ctxs.append(spare_ctx)
        render_streams.append(spare_stream)

    # The same step through the host-buffer entry point (pt_render: device film cleared, rendered, copied to the caller's
    # buffer), wall time per call: the PCIe-inclusive rate.  Reported beside the headline, never as it.
    pcie = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and elapsed / args.steps < 0.5:
        def host_call(out):
            torch.cuda.synchronize()
            best = None
            for _ in range(3):
                t = time.perf_counter()
                _, hs = ctx.render(W, H, my_spp, depth, seed=args.seed, sample_offset=sample_offset, flags=base_flags, out=out)
                dt = time.perf_counter() - t
                best = dt if best is None else min(best, dt)
            return best, hs["rays"]
        pageable = np.zeros((H, W, 3), dtype=np.float32)
        with api.PinnedFilm(H, W) as film:
            t_pin, r_pin = host_call(film.array)
        t_page, _ = host_call(pageable)
        pcie = {"entry": "pt_render", "film_bytes": int(pageable.nbytes),
                "pinned_host_buffer": {"ms_per_step": t_pin * 1e3, "Mrays_per_s": r_pin / t_pin / 1e6},
                "pageable_host_buffer": {"ms_per_step": t_page * 1e3, "Mrays_per_s": r_pin / t_page / 1e6}}

    if rank == 0:
        B, P = work_per_ray(args.workload)
        bpr, fpr = bytes_per_ray(B, P), flops_per_ray(B, P)
        n_launch = max(1, prof["launches_intersect"])
        kernel = api.KERNEL_NAMES.get(prof["kernel"], "?")
        # with two steps in flight a launch's start event is recorded while the previous step still holds the GPU, so the
        # events of the timed region measure queueing too: the launch duration is then the library's own, from the step
        # that ran alone after the timed region
        ms_launch = (ms_render if in_flight == 1 else prof["ms_paths"]) / n_launch
        rays_launch = st0["rays"] / n_launch
        tflops = fpr * rays_launch / (ms_launch * 1e-3) / 1e12 if ms_launch > 0 else 0.0
        gbs = bpr * rays_launch / (ms_launch * 1e-3) / 1e9 if ms_launch > 0 else 0.0
        pmc = {}
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json"))).get(args.workload, {})
        except Exception:
            pmc = {}
        traffic = pmc.get("hbm_bytes_per_ray")
        roofline = {
            "bound": "valu", "kernel": kernel, "achieved": tflops, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tflops / VALU_PEAK_TFLOPS,
            "traffic": traffic * rays_launch if traffic is not None else None,
            "flops_per_ray": fpr, "bytes_per_ray": bpr, "B": B, "P": P, "work_per_ray_source": "profiles/work_per_ray.json",
            "rays_per_launch": rays_launch, "ms_per_launch": ms_launch, "launches": n_launch,
            "ms_per_launch_library_events": prof["ms_paths"] / n_launch,
            "ms_per_launch_source": "events around the render on its stream, timed region" if in_flight == 1 else "library events of one step alone (launches of the timed region overlap)",
            "hbm": {"algorithmic_GBs": gbs, "peak_GBs": HBM_PEAK_GBS,
                    "measured_GBs": (traffic * rays_launch / (ms_launch * 1e-3) / 1e9) if traffic is not None else None,
                    "measured_frac": (traffic * rays_launch / (ms_launch * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic is not None else None},
            "valu_issue": pmc.get("valu_issue"),
            "note": "achieved = (24 B + 30 P + 60) flop per ray x rays per launch / launch time, against the fp32 vector peak; B, P frozen on a fixed "
                    "CPU-built tree.  Node and object records come from LDS (k_paths) or L2 / Infinity Cache (k_paths_stream), so the "
                    "algorithmic byte rate under `hbm` is not HBM traffic: measured_* is, from the rocprofv3 PMC passes in profiles/ "
                    "(bytes per ray x this run's rays).  valu_issue: share of the vector ALUs' issue cycles in use, from the same passes.",
        }
        if cnt is not None and cnt["rays"]:
            roofline["measured_B"] = cnt["box_tests"] / cnt["rays"]
            roofline["measured_P"] = cnt["prim_tests"] / cnt["rays"]
        out = {
            "metric": "Mrays/s (incl. secondary bounces)", "value": rays_step * args.steps / elapsed / 1e6, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": name + (" [%d steps in flight]" % in_flight if in_flight > 1 else ""), "steps_in_flight": in_flight, "ms_step_alone": ms_alone, "width": W, "height": H, "spp_per_gpu": my_spp, "spp_total": spp_total,
                       "max_depth": depth, "objects": int(len(builder.arrays()["prim_type"])), "rays_per_step": rays_step,
                       "parallelism": "spp-sharded x%d (%s) + all-reduce(sum) of the fp32 accumulation buffer, overlapped with the next step's render" % (world, args.scaling),
                       "seed": args.seed},
            "roofline": roofline,
        }
        if pcie is not None:
            out["pcie_inclusive"] = pcie
        if pipelined is not None:
            out["pipelined"] = pipelined
        if unpipelined is not None:
            out["unpipelined"] = unpipelined
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(builder, depth, args.seed, W, H, spp)
            tree = cb.pop("bvh")
            cb["sample"] += "; the same walk through a CPU-built tree (not the reference's algorithm): %.1f Mrays/s on %d cores" % (tree["value"], tree["cores"])
            out["cpu_baseline"] = cb
            out["cpu_baseline_tree"] = tree
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    for cx in ctxs:
        cx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
