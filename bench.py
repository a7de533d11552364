#!/usr/bin/env python3
"""Benchmark of the hot path: camera ray -> nearest hit -> scatter/emit -> accumulate.

    python bench.py --gpus N --steps K --warmup W [--workload c2]

A step is one full render of the workload's image on every rank: each rank renders `spp` samples
per pixel of its own absolute sample range (weak scaling: the image gets N*spp samples), the linear
accumulation buffers are summed with one RCCL all-reduce, and rank 0 holds the result.  Inputs
(scene, BVH) are resident in HBM before the timed region.  Prints ONE JSON line on rank 0.
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

# The benchmark configurations of BASELINE.json.  B / P: mean box and object tests per ray, measured
# once with PT_RENDER_COUNT_TESTS on the library's tree (binned SAH, 16 bins, leaf <= 4) and frozen.
WORKLOADS = {
    "c1": dict(name="default scene() 400x225 16spp depth50", scene=scenes.default_scene, w=400, h=225, spp=16, depth=50, B=2.00, P=3.09),
    "c2": dict(name="random-spheres (485 spheres) 1920x1080 64spp depth50", scene=scenes.random_spheres_scene, w=1920, h=1080, spp=64, depth=50, B=16.78, P=4.01),
    "c3": dict(name="cornell box 1024x1024 1024spp depth50", scene=scenes.cornell_scene, w=1024, h=1024, spp=1024, depth=50, B=3.98, P=4.85),
    "c4": dict(name="mixed 100k objects 3840x2160 256spp depth50", scene=scenes.mixed_scene, w=3840, h=2160, spp=256, depth=50, B=58.98, P=25.73),
    "c5": dict(name="1M spheres 3840x2160 4096spp depth50", scene=scenes.sphere_slab_scene, w=3840, h=2160, spp=4096, depth=50, B=74.09, P=27.74),
}
HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak
VALU_PEAK_TFLOPS = 78.6     # fp32 vector peak WITHOUT packed v_pk_fma (half of the 157.3 spec figure)


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--seed", type=int, default=1984)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--spp", type=int, default=0, help="override the workload's samples per pixel (profiling runs; not the headline configuration)")
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
    dist = None
    if world > 1:
        import torch.distributed as dist

        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    wl = WORKLOADS[args.workload]
    builder = wl["scene"]()
    ctx = api.Context(local_rank)
    ctx.upload(builder.desc())
    W, H, spp, depth = wl["w"], wl["h"], wl["spp"], wl["depth"]
    name = wl["name"]
    if args.spp > 0:
        spp = args.spp
        name += " [spp overridden to %d]" % spp
    accum = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    sample_offset, _ = distributed.weak_shard(rank, world, spp)
    params = api.Context.params(W, H, spp, depth, seed=args.seed, sample_offset=sample_offset)

    def step(want_stats=False, flags=0):
        accum.zero_()
        params.flags = flags
        st = ctx.render_device(params, accum.data_ptr(), stream, want_stats=want_stats)
        distributed.all_reduce_sum(accum)  # linear sums; gamma comes after (main.cu:95-98 is non-linear)
        return st

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    st0 = step(want_stats=True)  # ray count of this rank's sample range (deterministic per step)
    for _ in range(max(0, args.warmup - 1)):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0

    rays_t = torch.tensor([float(st0["rays"])], dtype=torch.float64, device="cuda")
    el_t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    if dist is not None:
        dist.all_reduce(rays_t)
        dist.all_reduce(el_t, op=dist.ReduceOp.MAX)
    rays_step = rays_t.item()
    elapsed = el_t.item()

    # Dominant kernel (k_paths, the persistent-workgroup wavefront kernel): launch durations from HIP events
    # recorded by the library on the launch stream around every k_paths launch of one more step.
    prof = step(want_stats=True, flags=api.RENDER_STAGE_TIMES)
    torch.cuda.synchronize()
    cnt = None
    if rank == 0 and args.workload in ("c1", "c2"):
        cp = api.Context.params(W, H, min(spp, 4), depth, seed=args.seed, flags=api.RENDER_COUNT_TESTS)
        accum.zero_()
        cnt = ctx.render_device(cp, accum.data_ptr(), stream, want_stats=True)

    if rank == 0:
        bpr, fpr = bytes_per_ray(wl["B"], wl["P"]), flops_per_ray(wl["B"], wl["P"])
        n_launch = max(1, prof["launches_intersect"])
        ms_launch = prof["ms_paths"] / n_launch
        rays_launch = prof["rays"] / n_launch
        achieved = bpr * rays_launch / (ms_launch * 1e-3) / 1e9 if ms_launch > 0 else 0.0
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get(args.workload, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        roofline = {
            "bound": "hbm", "kernel": "k_paths" if len(builder.arrays()["prim_type"]) <= 1000 else "k_paths_stream", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            "bytes_per_ray": bpr, "rays_per_launch": rays_launch, "ms_per_launch": ms_launch, "launches": n_launch,
            "valu": {"achieved_tflops": fpr * rays_launch / (ms_launch * 1e-3) / 1e12 if ms_launch > 0 else 0.0,
                     "peak_tflops": VALU_PEAK_TFLOPS, "peak_kind": "fp32 vector, non-packed", "flops_per_ray": fpr},
            "note": "algorithmic bytes (32B+32P+100 per ray) over event-timed k_paths launches; for cache-resident scenes the "
                    "figure is a rate of useful bytes, not HBM traffic (see `traffic`, from rocprofv3 PMC passes); on the "
                    "LDS-resident scenes the kernel is VALU-issue-bound (profiles/r01_pmc_c2.json), and so is k_paths_stream on the "
                    "large ones (profiles/r01_pmc_c5_8spp.json)",
        }
        if cnt is not None and cnt["rays"]:
            roofline["measured_B"] = cnt["box_tests"] / cnt["rays"]
            roofline["measured_P"] = cnt["prim_tests"] / cnt["rays"]
        out = {
            "metric": "Mrays/s (incl. secondary bounces)", "value": rays_step * args.steps / elapsed / 1e6, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": name, "width": W, "height": H, "spp_per_gpu": spp, "spp_total": spp * world,
                       "max_depth": depth, "objects": int(len(builder.arrays()["prim_type"])), "rays_per_step": rays_step,
                       "parallelism": "spp-sharded x%d + all-reduce(sum) of the fp32 accumulation buffer" % world, "seed": args.seed},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(builder, depth, args.seed, W, H, spp)
        print(json.dumps(out), flush=True)
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
