"""Fuzz: the tree walk against the list scan, both on the GPU, over random scenes and rays.

The list scan (TRACE_LINEAR) is the reference's hittable_list::hit in list order and is what the parity tests pin
against the oracle; the tree walk (TRACE_BVH, and TRACE_BVH_IN_SCENE with the ray cone a render uses) has to return the
same record for every ray, bit for bit.  This script
throws scenes of random size, mix, scale and distance from the origin at both and counts differing records.  Rays: pairs
of random points, rays aimed at objects from far away with unnormalised directions, rays grazing the objects' bounding
spheres, and three generations of rays leaving the hit points found (what a render traces).

Usage: python tools/mb/fuzz_trace.py <seconds> [seed]      (prints one line per scene; exit status 1 on a mismatch)
"""
import sys
import time

sys.path.insert(0, ".")
import numpy as np  # noqa: E402

from pathtracer_amd import api, scenes  # noqa: E402

FIELDS = ("t", "front", "n", "p")


def random_scene(rng):
    n = int(rng.choice([3, 17, 120, 480, 3000, 40000]))
    mix = rng.choice(["sphere", "nocone", "any", "cone"])
    spread = 10.0 ** rng.uniform(0.5, 3.0)
    centre = rng.standard_normal(3) * 10.0 ** rng.uniform(0.0, 4.0) * (rng.random() < 0.6)
    size_lo = 10.0 ** rng.uniform(-2.5, 0.0)
    size_hi = size_lo * 10.0 ** rng.uniform(0.0, 1.5)
    b = scenes.SceneBuilder()
    m = b.lambertian((0.5, 0.5, 0.5))
    pos = centre + (rng.random((n, 3)) - 0.5) * spread
    for k in range(n):
        r = float(np.exp(rng.uniform(np.log(size_lo), np.log(size_hi))))
        if mix == "sphere":
            what = 0
        elif mix == "nocone":
            what = int(rng.integers(0, 5))
        elif mix == "cone":
            what = 5
        else:
            what = int(rng.integers(0, 6))
        c = pos[k]
        if what == 0:
            b.sphere(tuple(c), r, m)
        elif what == 1:
            b.cylinder2(tuple(c), r * 10.0 ** rng.uniform(-1, 1), r, m)
        elif what == 2:
            b.xy_rect(c[0] - r, c[0] + r, c[1] - r * 0.7, c[1] + r * 0.7, c[2], m)
        elif what == 3:
            b.xz_rect(c[0] - r, c[0] + r, c[2] - r * 0.7, c[2] + r * 0.7, c[1], m)
        elif what == 4:
            b.yz_rect(c[1] - r, c[1] + r, c[2] - r * 0.7, c[2] + r * 0.7, c[0], m)
        else:
            b.cone(tuple(c), r * 10.0 ** rng.uniform(-1.3, 1.3), r, m)
    if rng.random() < 0.3:  # a ground sphere that encloses nothing but touches everything's box
        b.sphere(tuple(centre + [0.0, -spread * 20.0 - spread * 0.5, 0.0]), spread * 20.0, m)
    b.camera(tuple(centre + [0, 0, spread]), tuple(centre), (0, 1, 0), 40.0, 1.0, 0.0, spread)
    return b, pos, (n, mix, spread, np.linalg.norm(centre), size_lo, size_hi)


def rays_for(pos, spread, size_hi, n, rng):
    k = n // 3
    lo, hi = pos.min(0) - size_hi, pos.max(0) + size_hi
    a = lo + (hi - lo) * rng.random((k, 3))
    bb = lo + (hi - lo) * rng.random((k, 3))
    o1, d1 = a, bb - a
    tgt = pos[rng.integers(0, len(pos), k)] + (rng.random((k, 3)) - 0.5) * size_hi
    v = rng.standard_normal((k, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    dist = spread * 10.0 ** rng.uniform(-1.0, 2.0, k)
    o2, d2 = tgt - v * dist[:, None], v * 10.0 ** rng.uniform(-1.0, 2.5, k)[:, None]
    # grazing: through a point at distance ~size from an object's centre, perpendicular to the radius
    c = pos[rng.integers(0, len(pos), k)]
    rad = rng.standard_normal((k, 3))
    rad /= np.linalg.norm(rad, axis=1, keepdims=True)
    w = np.cross(rad, rng.standard_normal((k, 3)))
    w /= np.linalg.norm(w, axis=1, keepdims=True)
    q = c + rad * (size_hi * rng.random((k, 1)))
    dist = spread * 10.0 ** rng.uniform(-2.0, 1.5, k)
    o3, d3 = q - w * dist[:, None], w * 10.0 ** rng.uniform(-1.0, 2.5, k)[:, None]
    o = np.concatenate([o1, o2, o3]).astype(np.float32)
    d = np.concatenate([d1, d2, d3]).astype(np.float32)
    return np.ascontiguousarray(o), np.ascontiguousarray(d)


def differing(g, h):
    bad = g["prim"] != h["prim"]
    hit = h["prim"] != api.NO_PRIM
    for f in FIELDS:
        x, y = g[f].view(np.uint32), h[f].view(np.uint32)
        ne = x != y
        bad |= hit & (ne if ne.ndim == 1 else ne.any(axis=1))
    return bad


def explain(arrays, o, d, prim):
    """What the quadratic of the object the scan hit looks like for this ray (fp32, as cone.h:101-112 computes it)."""
    f = np.float32
    t = int(arrays["prim_type"][prim])
    p = arrays["prim_param"][prim].astype(np.float32)
    if t != api.CONE:
        return "object %d type %d params %s" % (prim, t, p.tolist())
    oc = (o - p[:3]).astype(np.float32)
    D = f(f(p[3] - o[1]) + p[1])
    sxz = f(f(d[0] * d[0]) + f(d[2] * d[2]))
    tdy = f(f(p[5] * d[1]) * d[1])
    a = f(sxz - tdy)
    hb = f(f(f(oc[0] * d[0]) + f(oc[2] * d[2])) + f(f(p[5] * d[1]) * D))
    c = f(f(f(oc[0] * oc[0]) + f(oc[2] * oc[2])) - f(f(p[5] * D) * D))
    disc = f(f(hb * hb) - f(a * c))
    return "cone %d height %g radius %g tangens %g: a = %g (its terms %g: %.1f units in the last place), half_b %g c %g discriminant %g (half_b^2 %g)" % (
        prim, p[3], p[4], p[5], a, sxz, abs(float(a)) / float(np.spacing(sxz)), hb, c, disc, f(hb * hb))


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.Generator(np.random.PCG64(seed))
    ctx = api.Context(0)
    t0 = time.time()
    total = bad_total = scenes_done = 0
    while time.time() - t0 < budget:
        b, pos, info = random_scene(rng)
        ctx.upload(b.desc())
        n_rays = 600000 if info[0] <= 3000 else 150000
        o, d = rays_for(pos, info[2], info[5], n_rays, rng)
        box_lo = (pos.min(0) - 0.5 * info[4]).astype(np.float32)  # (the objects' centres: a ground sphere does not widen it)
        box_hi = (pos.max(0) + 0.5 * info[4]).astype(np.float32)
        bad_scene = 0
        hits_scene = 0
        for gen in range(4):
            scan = ctx.trace(o, d, mode=api.TRACE_LINEAR)
            tree = ctx.trace(o, d)
            bad = differing(tree, scan)
            # the cone slope a render's kernels use (narrower near the scene's small objects; valid for origins within the
            # bounds of the objects and the camera, where a render's rays start) must find the same hits for such rays
            tree_in = ctx.trace(o, d, mode=api.TRACE_BVH_IN_SCENE)
            inside = ((o >= box_lo) & (o <= box_hi)).all(axis=1)
            bad_in = differing(tree_in, scan) & ~bad & inside
            if bad_in.any():
                tree = tree_in
                bad |= bad_in
            bad_scene += int(bad.sum())
            total += len(o)
            hit = scan["prim"] != api.NO_PRIM
            hits_scene += int(hit.sum())
            for i in np.flatnonzero(bad)[:3]:
                print("    gen %d ray o=%s d=%s scan t=%r prim=%d flags=%d | tree t=%r prim=%d flags=%d" % (
                    gen, o[i].tolist(), d[i].tolist(), scan["t"][i], scan["prim"][i], scan["flags"][i], tree["t"][i], tree["prim"][i], tree["flags"][i]))
                if scan["prim"][i] != api.NO_PRIM:
                    print("        " + explain(b.arrays(), o[i], d[i], int(scan["prim"][i])))
            if hit.sum() < 100:
                break
            p = scan["p"][hit]
            nrm = scan["n"][hit]
            r = rng.standard_normal((len(p), 3))
            r /= np.linalg.norm(r, axis=1, keepdims=True)
            o = np.ascontiguousarray(p, dtype=np.float32)
            d = np.ascontiguousarray((nrm + r * rng.choice([1.0, 0.3, 3.0])) * 10.0 ** rng.uniform(-1, 1), dtype=np.float32)
        scenes_done += 1
        bad_total += bad_scene
        print("scene %3d: %6d objects %-7s spread %8.1f |centre| %9.1f size %.3g..%.3g | hits %8d differing %d | %.0f s" % (
            (scenes_done,) + info + (hits_scene, bad_scene, time.time() - t0)), flush=True)
    print("TOTAL scenes %d rays %d differing %d" % (scenes_done, total, bad_total))
    return 1 if bad_total else 0


if __name__ == "__main__":
    sys.exit(main())
