#!/usr/bin/env python3
"""Freezes B and P, the box tests and object tests per ray that the roofline's algorithmic bytes and flops are
priced on (bytes/ray = 32 B + 32 P + 100, flops/ray = 24 B + 30 P + 60).

They are measured ONCE, on the CPU, by a plain near-first stack traversal (oracle/pt_oracle.c: orc_count_tree_work)
of a FIXED reference tree -- binned SAH, 16 bins, leaves of at most 4 objects, built by the library's host-side
builder -- over the committed ray fixtures tests/golden/rays_*.npz, so that whatever tree or traversal the GPU kernels
use (single-object leaves, four-wide nodes, a widened slab test) cannot inflate the achieved figure.

    python profiles/freeze_work_per_ray.py      # writes profiles/work_per_ray.json; no GPU needed
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pt_oracle as O  # noqa: E402
from pathtracer_amd import api, scenes  # noqa: E402
from tests import util  # noqa: E402

CONFIGS = {
    "c1": ("default", scenes.default_scene),
    "c2": ("random_spheres", scenes.random_spheres_scene),
    "c3": ("cornell", scenes.cornell_scene),
    "c4": ("mixed_100k", scenes.mixed_scene),
    "c5": ("spheres_1m", scenes.sphere_slab_scene),
}


def main():
    out = {}
    for cfg, (fixture, builder) in CONFIGS.items():
        b = builder()
        g = np.load(os.path.join(ROOT, "tests", "golden", "rays_%s.npz" % fixture))
        nodes, order, depth = api.bvh_build_host(b.desc(), leaf_max=4)
        nb, npr = O.count_tree_work(util.oracle_scene(b), nodes, order, g["o"], g["d"])
        n = len(g["o"])
        B, P = nb / n, npr / n
        out[cfg] = {"fixture": "tests/golden/rays_%s.npz" % fixture, "rays": n, "tree": "binned SAH, 16 bins, leaf <= 4", "nodes": int(len(nodes)),
                    "depth": int(depth), "B": round(B, 2), "P": round(P, 2), "bytes_per_ray": round(32 * B + 32 * P + 100, 1),
                    "flops_per_ray": round(24 * B + 30 * P + 60, 1)}
        print(cfg, out[cfg], flush=True)
    json.dump(out, open(os.path.join(ROOT, "profiles", "work_per_ray.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
