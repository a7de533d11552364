"""Fuzz: renders of random small scenes, every pipeline of the library against the CPU oracle.

Scenes of 1 to 20 000 objects of every kind with random materials (all four classes, lights included), sizes and
positions (three in ten far from the origin), a random camera (aperture included; three in ten far from the scene), 24 x 16 pixels, 4 samples, depth 50.  For every scene: the default
pipeline against the oracle (every pixel within 2e-5, equal ray counts) and the other pipelines against the default one
(bit-identical, equal ray counts).  Prints one line per scene; exit status 1 when anything differs.

Usage: python tests/studies/fuzz_render.py <seconds> [seed]
"""
import sys
import time

sys.path.insert(0, ".")
import numpy as np  # noqa: E402

from pathtracer_amd import api, scenes  # noqa: E402
from tests import util  # noqa: E402


def random_scene(rng):
    n = int(rng.choice([1, 2, 7, 9, 40, 150, 600, 3000, 20000], p=[0.1, 0.1, 0.1, 0.1, 0.15, 0.15, 0.15, 0.1, 0.05]))  # (3000 and more: the streaming kernel)
    spread = 10.0 ** rng.uniform(0.3, 1.5)
    bg = (0.7, 0.8, 1.0) if rng.random() < 0.6 else (0.0, 0.0, 0.0)
    b = scenes.SceneBuilder(bg)
    mats = []
    for _ in range(int(rng.integers(1, 9))):
        k = int(rng.integers(0, 4))
        c = rng.random(3)
        if k == 0:
            mats.append(b.lambertian(c))
        elif k == 1:
            mats.append(b.metal(c, rng.random() * 1.2))
        elif k == 2:
            mats.append(b.dielectric(1.0 + rng.random()))
        else:
            mats.append(b.diffuse_light(c * 8.0))
    size = spread * 10.0 ** rng.uniform(-1.5, -0.5)
    centre = rng.standard_normal(3) * 10.0 ** rng.uniform(1.0, 3.5) if rng.random() < 0.3 else np.zeros(3)  # a scene far from the origin
    for _ in range(n):
        c = centre + (rng.random(3) - 0.5) * spread
        r = size * (0.3 + rng.random())
        m = mats[int(rng.integers(0, len(mats)))]
        what = int(rng.integers(0, 6))
        if what == 0:
            b.sphere(tuple(c), r, m)
        elif what == 1:
            b.cylinder2(tuple(c), r * 10.0 ** rng.uniform(-0.7, 0.7), r, m)
        elif what == 2:
            b.xy_rect(c[0] - r, c[0] + r, c[1] - r, c[1] + r, c[2], m)
        elif what == 3:
            b.xz_rect(c[0] - r, c[0] + r, c[2] - r, c[2] + r, c[1], m)
        elif what == 4:
            b.yz_rect(c[1] - r, c[1] + r, c[2] - r, c[2] + r, c[0], m)
        else:
            b.cone(tuple(c), r * 10.0 ** rng.uniform(-0.7, 0.7), r, m)
    if rng.random() < 0.5:  # a ground sphere
        b.sphere(tuple(centre + [0.0, -spread * 50.0 - spread * 0.5, 0.0]), spread * 50.0, mats[0])
    eye = rng.standard_normal(3)
    far = 10.0 ** rng.uniform(0.5, 1.8) if rng.random() < 0.3 else 1.0  # a camera far from the scene, with a lens to match
    eye *= far * spread * rng.uniform(0.8, 2.0) / np.linalg.norm(eye)
    b.camera(tuple(centre + eye), tuple(centre + (rng.random(3) - 0.5) * spread * 0.3), (0, 1, 0), float(rng.uniform(20, 70) / far), 1.5,
             float(rng.choice([0.0, 0.05 * spread])), float(np.linalg.norm(eye)))
    return b, n


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.Generator(np.random.PCG64(seed))
    ctx = api.Context(0)
    w, h, spp, depth = 24, 16, 4, 50
    t0 = time.time()
    done = failed = 0
    while time.time() - t0 < budget:
        b, n = random_scene(rng)
        osc = util.oracle_scene(b)
        ref, rays = osc.render(w, h, spp, depth)
        ctx.upload(b.desc())
        got, st = ctx.render(w, h, spp, depth)
        tol = 2e-5 * np.maximum(1.0, np.abs(ref))
        bad_px = int((np.abs(got - ref) > tol).any(axis=2).sum())
        notes = []
        if bad_px or st["rays"] != rays:
            notes.append("default pipeline (kernel %d): %d pixels off, rays %d against %d" % (st["kernel"], bad_px, st["rays"], rays))
        for tune, flags in (({"pipeline": 0}, 0), ({"pipeline": 1}, 0), ({}, api.RENDER_NO_LDS_SCENE), ({}, api.RENDER_SPLIT_KERNELS)):
            with ctx.tuned(**tune):
                other, so = ctx.render(w, h, spp, depth, flags=flags)
            same = np.array_equal(other.view(np.uint32), got.view(np.uint32)) or np.array_equal(np.isnan(other), np.isnan(got)) and np.allclose(other, got, rtol=0, atol=0, equal_nan=True)
            if not same or so["rays"] != st["rays"]:
                notes.append("%r flags %d (kernel %d): differs from the default, rays %d against %d" % (tune, flags, so["kernel"], so["rays"], st["rays"]))
        done += 1
        failed += bool(notes)
        print("scene %3d: %4d objects kernel %d rays %7d %s | %.0f s" % (done, n, st["kernel"], rays, "ok" if not notes else "; ".join(notes), time.time() - t0), flush=True)
    print("TOTAL scenes %d differing %d" % (done, failed))
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
