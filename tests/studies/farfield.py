"""Ad-hoc: small objects seen from far away with long unnormalised directions -- the regime where the
reference's fp32 quadratics accept hits outside the exact surfaces.  BVH path vs oracle, per object kind."""
import sys
sys.path.insert(0, ".")
import numpy as np
from pathtracer_amd import api, scenes
from tests import util

def scene(kind, n, rng, center):
    b = scenes.SceneBuilder()
    m = b.lambertian((0.5, 0.5, 0.5))
    c = center + (rng.random((n, 3)) - 0.5) * 60
    for k in range(n):
        r = 0.05 + rng.random() * 0.4
        what = kind if kind != "mixed" else ("sphere", "cyl", "cone", "rect")[k % 4]
        if what == "sphere":
            b.sphere(c[k], r, m)
        elif what == "cyl":
            b.cylinder2(c[k], 0.2 + rng.random(), r, m)
        elif what == "cone":
            b.cone(c[k], 0.3 + rng.random(), r, m)
        else:
            ax = k // 4 % 3
            lo = c[k] - r; hi = c[k] + r
            if ax == 0: b.xy_rect(lo[0], hi[0], lo[1], hi[1], c[k][2], m)
            elif ax == 1: b.xz_rect(lo[0], hi[0], lo[2], hi[2], c[k][1], m)
            else: b.yz_rect(lo[1], hi[1], lo[2], hi[2], c[k][0], m)
    b.camera(tuple(center + [0, 0, 100]), tuple(center), (0, 1, 0), 40.0, 1.0, 0.0, 100.0)
    return b, c

ctx = api.Context(0)
rng = np.random.Generator(np.random.PCG64(11))
for kind in ("sphere", "cyl", "rect", "cone", "mixed"):
    for center in (np.zeros(3), np.array([5000.0, 300.0, -2000.0])):
        n = 2000
        b, c = scene(kind, n, rng, center)
        ctx.upload(b.desc())
        osc = util.oracle_scene(b)
        nr = 150000
        tgt = c[rng.integers(0, n, nr)] + (rng.random((nr, 3)) - 0.5) * 1.2
        dist = 10 ** (1 + rng.random(nr) * 2.5)                      # 10 .. 3000 away
        dirn = rng.standard_normal((nr, 3)); dirn /= np.linalg.norm(dirn, axis=1, keepdims=True)
        o = (tgt - dirn * dist[:, None]).astype(np.float32)
        scale = 10 ** (rng.random(nr) * 3.5 - 1)                     # |d| 0.1 .. 300
        d = (dirn * scale[:, None]).astype(np.float32)
        ref = osc.trace(o, d)
        got = ctx.trace(o, d)
        ok = ref["nan_flag"] == 0
        bad = util.hits_equal(got, ref) & ok
        hit = ref["prim"] != api.NO_PRIM
        print("%-6s centre %-22s hits %6d flagged %5d mismatches %d" % (kind, center.tolist(), hit.sum(), (~ok).sum(), bad.sum()), flush=True)
        for i in np.flatnonzero(bad)[:3]:
            print("    ray", i, "oracle", ref["t"][i], ref["prim"][i], "bvh", got["t"][i], got["prim"][i], "dist %.1f |d| %.2f" % (dist[i], scale[i]))
