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


def hits_equal(g, h, nan_equal=False):
    """Bit-level comparison of library hits g with oracle hits h; returns a mismatch mask.  nan_equal: a NaN equals
    any NaN (the bit pattern of 0 / 0 differs between x86 and gfx950; nothing downstream depends on it)."""
    def differ(x, y):
        ne = x.view(np.uint32) != y.view(np.uint32)
        if nan_equal and x.dtype.kind == "f":
            ne &= ~(np.isnan(x) & np.isnan(y))
        return ne

    bad = g["prim"] != h["prim"]
    hit = h["prim"] != 0xFFFFFFFF
    for f in ("t", "front"):
        bad |= hit & differ(g[f], h[f])
    for f in ("n", "p"):
        bad |= hit & differ(g[f], h[f]).any(axis=1)
    return bad


def cone_field(slopes, n_per, center, rng, spread=60.0, size=(0.3, 1.3)):
    """n_per cones of each radius / height ratio in `slopes`, heights in `size`, scattered over a cube of side `spread`
    around `center`; returns (builder, [(centre, height, radius), ...])."""
    center = np.asarray(center, dtype=np.float64)
    b = scenes.SceneBuilder()
    m = b.lambertian((0.5, 0.5, 0.5))
    cones = []
    for s in slopes:
        for _ in range(n_per):
            c = center + (rng.random(3) - 0.5) * spread
            h = size[0] + rng.random() * (size[1] - size[0])
            b.cone(tuple(c), h, s * h, m)
            cones.append((c, h, s * h))
    b.camera(tuple(center + [0, 0, 100]), tuple(center), (0, 1, 0), 40.0, 1.0, 0.0, 100.0)
    return b, cones


def grazing_cone_rays(cones, n, rng, dist_exp=(1.0, 4.0)):
    """n rays that graze the sides of cones from 10^dist_exp units away: through a point of a cone's side along a
    direction of its tangent plane there -- a third of them along the side's generator line, where the leading
    coefficient of cone.h:108 cancels -- tilted out of the plane by 1e-3 .. 1e-1, with unnormalised directions of length
    0.1 .. 300.  The regime in which the reference's fp32 quadratic accepts roots farthest outside the exact surface."""
    o = np.zeros((n, 3))
    d = np.zeros((n, 3))
    for i in range(n):
        c, h, r = cones[rng.integers(0, len(cones))]
        y = rng.random() * h
        rad = r * (1.0 - y / h)  # the apex is at the top: D = height - o.y + center.y, cone.h:105
        phi = rng.random() * 2.0 * np.pi
        q = c + np.array([rad * np.cos(phi), y, rad * np.sin(phi)])
        nrm = np.array([np.cos(phi), r / h, np.sin(phi)])
        nrm /= np.linalg.norm(nrm)
        if rng.integers(0, 3) == 0:
            v = np.array([-np.cos(phi) * r, h, -np.sin(phi) * r])
            v *= (1.0 if rng.random() < 0.5 else -1.0) / np.linalg.norm(v)
        else:
            v = rng.standard_normal(3)
            v -= nrm * (v @ nrm)
            v /= np.linalg.norm(v)
        v = v + nrm * rng.standard_normal() * 10.0 ** (-3.0 + 2.0 * rng.random())
        v /= np.linalg.norm(v)
        o[i] = q - v * 10.0 ** (dist_exp[0] + rng.random() * (dist_exp[1] - dist_exp[0]))
        d[i] = v * 10.0 ** (rng.random() * 3.5 - 1.0)
    return np.ascontiguousarray(o, dtype=np.float32), np.ascontiguousarray(d, dtype=np.float32)


def needed_cone_slope(arrays, o, d, hits):
    """For every hit of `hits` (oracle records): the slope kappa a tree needs so that the hit point lies within
    kappa * (|d| t + the object's extent) + 2^-17 * (largest box coordinate) of the object's exact bounding volume on every
    axis (the library's box padding and slab test); 0 where the point is inside.  Quadrics only (rects: 0)."""
    prim = hits["prim"]
    need = np.zeros(len(prim))
    ok = prim != 0xFFFFFFFF
    q = arrays["prim_param"][prim[ok]].astype(np.float64)
    t = arrays["prim_type"][prim[ok]]
    p = hits["p"][ok].astype(np.float64)
    sph = t == api.SPHERE
    r = np.where(sph, np.abs(q[:, 3]), np.abs(q[:, 4]))
    h = np.where(sph, 0.0, q[:, 3])
    lo = np.stack([q[:, 0] - r, np.where(sph, q[:, 1] - r, q[:, 1]), q[:, 2] - r], 1)
    hi = np.stack([q[:, 0] + r, np.where(sph, q[:, 1] + r, q[:, 1] + h), q[:, 2] + r], 1)
    m = np.maximum(np.abs(lo), np.abs(hi)).max(1)
    excess = np.maximum(np.maximum(lo - p, p - hi), 0.0).max(1)
    reach = np.linalg.norm(d[ok].astype(np.float64), axis=1) * np.abs(hits["t"][ok].astype(np.float64)) + r + np.abs(h)
    quad = (t == api.SPHERE) | (t == api.CONE) | (t == api.CYLINDER)
    need[ok] = np.where(quad, np.maximum(excess - m * 2.0 ** -17, 0.0) / np.maximum(reach, 1e-30), 0.0)
    return need
