#!/usr/bin/env python3
"""Generates the committed fixtures under tests/golden/ from the CPU oracle.

    python tests/golden/make_golden.py

The reference cannot be compiled in this image (it needs the CUDA toolkit headers and nvcc), and it
ships no tests, so these vectors come from oracle/pt_oracle.c -- the plain-C restatement of the
reference's hit / scan / scatter / render arithmetic.  They pin the ORACLE (a regression in it shows
up as a diff here) and they are the fixed ray batches the GPU path is compared on.  The one vector
that does come from the reference itself, recorded at survey time, is in reference_known_answer.json.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from pathtracer_amd import api, scenes  # noqa: E402  (scene definitions only; no GPU call is made)
from tests import util  # noqa: E402

SCENES = {
    "default": scenes.default_scene,
    "cornell": scenes.cornell_scene,
    "random_spheres": scenes.random_spheres_scene,
    "mixed_1k": lambda: scenes.mixed_scene(n=1000, seed=3, extent=20.0, n_materials=64),
}


# the large benchmark scenes: ray batches only (their images are checked on crops at full size by the GPU tests)
LARGE_SCENES = {
    "mixed_100k": scenes.mixed_scene,
    "spheres_1m": scenes.sphere_slab_scene,
}


def main():
    # (b) nearest-hit batches: 4096 rays per scene (camera, two generations of scattered rays, box rays)
    for name, f in list(SCENES.items()) + list(LARGE_SCENES.items()):
        b = f()
        osc = util.oracle_scene(b)
        # (the large scenes without the box rays: rays across the whole of such a scene cost several times what any
        # ray of a render does, and these batches also fix the work per ray the benchmark's roofline is priced on)
        o, d = util.mixed_rays(b, osc, 4096, seed=17, box=name not in LARGE_SCENES)
        h = osc.trace(o, d)
        np.savez_compressed(os.path.join(HERE, "rays_%s.npz" % name), o=o, d=d, t=h["t"], prim=h["prim"], n=h["n"],
                            front=h["front"], p=h["p"], nan_flag=h["nan_flag"])
    # (d) linear-sum images with the shared counter-based stream
    imgs = {}
    for name, (w, hgt, spp, depth) in {"default": (200, 112, 4, 5), "cornell": (64, 64, 4, 5), "random_spheres": (96, 54, 2, 50),
                                       "mixed_1k": (64, 36, 2, 8)}.items():
        osc = util.oracle_scene(SCENES[name]())
        img, rays = osc.render(w, hgt, spp, depth)
        imgs[name + "_img"] = img
        imgs[name + "_meta"] = np.array([w, hgt, spp, depth, rays], dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "images.npz"), **imgs)
    # (c) scatter known answers with injected random numbers
    b = scenes.SceneBuilder()
    mats = [b.lambertian((0.4, 0.2, 0.1)), b.metal((0.7, 0.6, 0.5), 0.3), b.metal((0.9, 0.9, 0.9), 7.0), b.dielectric(1.5),
            b.diffuse_light((4, 4, 4))]
    b.sphere((0, 0, 0), 1.0, mats[0])
    b.camera((0, 0, 5), (0, 0, 0), (0, 1, 0), 40.0, 1.0, 0.0, 5.0)
    osc = util.oracle_scene(b)
    rng = np.random.Generator(np.random.PCG64(23))
    rows = []
    for k in range(64):
        o = (rng.standard_normal(3) * 3).astype(np.float32)
        o *= np.float32(3.0 / max(1e-3, np.linalg.norm(o)))
        if k % 4 == 3:
            o *= np.float32(0.2)  # inside the sphere: back faces, total internal reflection
        d = (-o + rng.standard_normal(3).astype(np.float32) * 0.4).astype(np.float32)
        hit, rec = osc.hit_prim(0, o, d)
        if not hit:
            continue
        sp = rng.random(3).astype(np.float32) * 2 - 1
        sp *= np.float32(0.9 / max(1.0, np.linalg.norm(sp)))
        xi = float(rng.random())
        for m in mats:
            ok, att, so, sd = osc.scatter(m, o, d, rec, sphere_pt=sp, xi=xi)
            rows.append(dict(mat=int(m), o=o.tolist(), d=d.tolist(), sphere_pt=sp.astype(np.float32).tolist(), xi=float(np.float32(xi)),
                             t=float(rec["t"]), ok=bool(ok), atten=att.tolist(), so=so.tolist(), sd=sd.tolist()))
    json.dump(rows, open(os.path.join(HERE, "scatter_kat.json"), "w"))
    print("wrote fixtures to", HERE)


if __name__ == "__main__":
    main()
