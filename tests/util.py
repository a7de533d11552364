"""Shared helpers of the test-suite: the same flattened scene fed to the oracle and to the library."""
import numpy as np

from oracle import pt_oracle as O
from pathtracer_amd import api, scenes


def oracle_scene(builder):
    """The oracle's view of a pathtracer_amd.scenes builder (same arrays, oracle-built camera)."""
    a = builder.arrays()
    s = O.Scene(a["prim_type"], a["prim_param"], a["prim_material"], a["mat_type"], a["mat_param"], builder.background)
    s.set_camera(*builder.camera_args)
    return s


def camera_rays(osc, width, height, n, rng):
    """n pinhole-jittered camera rays through the oracle's get_ray (any fixed disk point)."""
    o = np.zeros((n, 3), np.float32)
    d = np.zeros((n, 3), np.float32)
    for k in range(n):
        u, v = rng.random(), rng.random()
        o[k], d[k] = osc.get_ray(np.float32(u), np.float32(v), (0.0, 0.0))
    return o, d


def secondary_rays(osc, o, d, rng):
    """Rays leaving the first hit points of (o, d) in random directions (what scatter produces)."""
    h = osc.trace(o, d)
    ok = h["prim"] != 0xFFFFFFFF
    p = h["p"][ok]
    n = h["n"][ok]
    r = rng.standard_normal((len(p), 3)).astype(np.float32)
    r /= np.linalg.norm(r, axis=1, keepdims=True).astype(np.float32)
    dirs = (n + r).astype(np.float32)
    return p.astype(np.float32), dirs


def box_rays(lo, hi, n, rng):
    """Rays between random point pairs of a box (incoherent, many misses)."""
    lo, hi = np.asarray(lo, np.float32), np.asarray(hi, np.float32)
    a = (lo + (hi - lo) * rng.random((n, 3), dtype=np.float32)).astype(np.float32)
    b = (lo + (hi - lo) * rng.random((n, 3), dtype=np.float32)).astype(np.float32)
    return a, (b - a).astype(np.float32)


def mixed_rays(builder, osc, n, seed, width=400, height=225, box=True):
    """n rays: camera rays, two generations of rays scattered from their hit points and (box=True) rays between random
    points of the scene's bounds; box=False leaves a third each of the first three (what a render traces)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    n1 = n // 4 if box else (n + 2) // 3
    o1, d1 = camera_rays(osc, width, height, n1, rng)
    o2, d2 = secondary_rays(osc, o1, d1, rng)
    o3, d3 = secondary_rays(osc, o2, d2, rng)
    pp = builder.arrays()["prim_param"]
    pt = builder.arrays()["prim_type"]
    small = np.ones(len(pt), bool)
    if len(pt) > 1:
        small = ~((pt == api.SPHERE) & (np.abs(pp[:, 3]) > 100))
    c = pp[small][:, :3] if small.any() else pp[:, :3]
    lo, hi = c.min(0) - 2, c.max(0) + 2
    n4 = max(0, n - len(o1) - len(o2) - len(o3)) if box else 0
    o4, d4 = box_rays(lo, hi, n4, rng)
    o = np.concatenate([o1, o2, o3, o4]).astype(np.float32)
    d = np.concatenate([d1, d2, d3, d4]).astype(np.float32)
    return np.ascontiguousarray(o), np.ascontiguousarray(d)


def hits_equal(g, h):
    """Bit-level comparison of library hits g with oracle hits h; returns a mismatch mask."""
    bad = g["prim"] != h["prim"]
    hit = h["prim"] != 0xFFFFFFFF
    for f in ("t", "front"):
        bad |= hit & (g[f].view(np.uint32) != h[f].view(np.uint32))
    for f in ("n", "p"):
        bad |= hit & (g[f].view(np.uint32) != h[f].view(np.uint32)).any(axis=1)
    return bad
