"""GPU parity tests, second part: the committed golden fixtures, the equivalence of the library's own
pipelines and kernel variants, ties, odd shapes and sizes, and the large scenes.

Same bars as tests/test_gpu_parity.py: nearest hit bit-exact; images |diff| <= 2e-5 * max(1, value) on
the linear sums for >= 99.8 % of pixels and relative RMSE <= 2e-3 against the oracle; images of two
library configurations that only schedule the same paths differently are bit-identical.
"""
import os

import numpy as np
import pytest

from pathtracer_amd import api, scenes
from tests import util
from tests.golden.make_golden import LARGE_SCENES, SCENES

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _image_close(got, ref):
    tol = 2e-5 * np.maximum(1.0, np.abs(ref))
    bad_px = (np.abs(got - ref) > tol).any(axis=2)
    rmse = float(np.sqrt(np.mean((got - ref) ** 2)) / max(1e-12, np.sqrt(np.mean(ref ** 2))))
    return bad_px.mean(), rmse


# ------------------------------------------------------------------------------------------------
# committed fixtures (no oracle call: these run from the files alone)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", list(SCENES))
@pytest.mark.parametrize("mode", [api.TRACE_BVH, api.TRACE_LINEAR, api.TRACE_BVH_IN_SCENE])
def test_golden_ray_batches(ctx, name, mode):
    g = np.load(os.path.join(GOLD, "rays_%s.npz" % name))
    ctx.upload(SCENES[name]().desc())
    got = ctx.trace(g["o"], g["d"], mode=mode)
    ref = {k: g[k] for k in ("t", "prim", "n", "front", "p")}
    bad = util.hits_equal(got, ref) & (g["nan_flag"] == 0)
    assert not bad.any(), (name, mode, int(bad.sum()))


@pytest.mark.parametrize("name", list(LARGE_SCENES))
def test_golden_ray_batches_of_the_large_scenes(ctx, name):
    """The 100k-object and 1M-sphere scenes: camera rays and two generations of scattered rays (the batches the
    benchmark's work per ray is frozen on), through the tree with any origin, as a render traverses it, and with boxes
    culled by exact geometry only -- which may lose hits whose discriminant is rounding noise (see
    test_geometric_cull_tolerance) but on these batches loses none."""
    g = np.load(os.path.join(GOLD, "rays_%s.npz" % name))
    ctx.upload(LARGE_SCENES[name]().desc())
    ref = {k: g[k] for k in ("t", "prim", "n", "front", "p")}
    for mode in (api.TRACE_BVH, api.TRACE_BVH_IN_SCENE):
        got = ctx.trace(g["o"], g["d"], mode=mode)
        bad = util.hits_equal(got, ref) & (g["nan_flag"] == 0)
        assert not bad.any(), (name, mode, int(bad.sum()))


@pytest.mark.parametrize("name", list(SCENES))
def test_golden_images(ctx, name):
    g = np.load(os.path.join(GOLD, "images.npz"))
    w, h, spp, depth, rays = (int(v) for v in g[name + "_meta"])
    ctx.upload(SCENES[name]().desc())
    got, st = ctx.render(w, h, spp, depth)
    frac, rmse = _image_close(got, g[name + "_img"])
    assert st["rays"] == rays, (st["rays"], rays)
    assert frac < 2e-3 and rmse < 2e-3, (name, frac, rmse)


# ------------------------------------------------------------------------------------------------
# the library's pipelines and kernel variants walk the same paths: bit-identical images
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,w,h,spp,depth", [("random_spheres", 192, 108, 4, 50), ("mixed_1k", 96, 54, 6, 12), ("cornell", 64, 64, 4, 5)])
def test_pipelines_agree_bitwise(ctx, name, w, h, spp, depth):
    ctx.upload(SCENES[name]().desc())
    base, st = ctx.render(w, h, spp, depth)
    variants = {
        "split kernels": dict(flags=api.RENDER_SPLIT_KERNELS),
        "split kernels, scene in HBM": dict(flags=api.RENDER_SPLIT_KERNELS | api.RENDER_NO_LDS_SCENE),
        "scene in HBM": dict(flags=api.RENDER_NO_LDS_SCENE),
        "counting build": dict(flags=api.RENDER_COUNT_TESTS),
    }
    for tag, kw in variants.items():
        img, s2 = ctx.render(w, h, spp, depth, **kw)
        assert s2["rays"] == st["rays"], tag
        assert np.array_equal(img.view(np.uint32), base.view(np.uint32)), tag
    counts = []
    for pipe in (0, 1, 2):  # the counting variants of the three persistent kernels do the same work per ray on a scene in LDS ...
        with ctx.tuned(pipeline=pipe, leaf_max=4, wide_nodes=0):
            _, sc = ctx.render(w, h, spp, depth, flags=api.RENDER_COUNT_TESTS | (api.RENDER_NO_LDS_SCENE if pipe == 1 else 0))
        assert sc["rays"] == st["rays"] and sc["box_tests"] > st["rays"], pipe
        counts.append((sc["kernel"], sc["box_tests"], sc["prim_tests"]))
    if counts[2][0] == 4:  # (k_paths_stream steps differently, and it is what renders a scene too large for LDS)
        assert counts[0][1:] == counts[2][1:], counts
    for tune in (dict(block=256), dict(sort_by_material=1), dict(slice_paths=4096), dict(slice_paths=1000000),
                 dict(split_queue=0), dict(split_queue=1),  # survivors in one queue / rays that head up queued apart
                 # scene in HBM: the streaming kernel (default there) at several batching thresholds and
                 # workgroup sizes, and the queue-based slice kernel
                 dict(hbm=1, stream_serve_min=1), dict(hbm=1, stream_serve_min=64),
                 dict(hbm=1, block=512), dict(hbm=1, pipeline=0),
                 # node and leaf steps in every iteration / by plain majority; short runs of paths per wave
                 dict(hbm=1, stream_phase=0), dict(hbm=1, stream_phase=4), dict(hbm=1, stream_run=64),
                 dict(pipeline=1),
                 # the queue-based kernel that takes one slice at a time (k_paths; the default, k_paths_ring, overlaps them)
                 dict(pipeline=0), dict(pipeline=0, block=256), dict(pipeline=0, split_queue=0), dict(pipeline=0, split_queue=1),
                 dict(pipeline=0, slice_paths=1000000),
                 # k_paths_ring taking a new slice only when the queue is empty, with every slice it can hold under way at once,
                 # with small workgroups, and with slices too large for its path index (which k_paths renders)
                 dict(pipeline=2), dict(pipeline=2, ring_low_water=0), dict(pipeline=2, ring_low_water=1 << 20, slice_paths=256),
                 dict(pipeline=2, block=256, slice_paths=1024), dict(pipeline=2, split_queue=0), dict(pipeline=2, slice_paths=1 << 18)):
        tune = dict(tune)
        flags = api.RENDER_NO_LDS_SCENE if tune.pop("hbm", None) else 0
        with ctx.tuned(**tune):
            img, s2 = ctx.render(w, h, spp, depth, flags=flags)
        assert s2["rays"] == st["rays"], tune
        assert np.array_equal(img.view(np.uint32), base.view(np.uint32)), tune


@pytest.mark.parametrize("name,w,h,spp,depth", [("mixed_1k", 96, 54, 4, 12), ("random_spheres", 96, 54, 2, 50)])
def test_tree_shape_does_not_change_the_image(ctx, name, w, h, spp, depth):
    """The answer is the list scan's whatever the tree looks like: leaves of 1 to 8 objects, traversed from LDS by
    k_paths and from HBM by k_paths_stream (binary or four-wide nodes), give bit-identical images."""
    ctx.upload(SCENES[name]().desc())
    base, st = ctx.render(w, h, spp, depth)
    for leaf in (1, 2, 3, 8):
        with ctx.tuned(leaf_max=leaf):
            ctx.upload(SCENES[name]().desc())
        for flags in (0, api.RENDER_NO_LDS_SCENE):
            img, s2 = ctx.render(w, h, spp, depth, flags=flags)
            assert s2["rays"] == st["rays"], (leaf, flags)
            assert np.array_equal(img.view(np.uint32), base.view(np.uint32)), (leaf, flags)
    with ctx.tuned(wide_nodes=0):  # k_paths_stream's two-child step (four-wide nodes are the default from HBM)
        ctx.upload(SCENES[name]().desc())
    img, s2 = ctx.render(w, h, spp, depth, flags=api.RENDER_NO_LDS_SCENE)
    assert s2["rays"] == st["rays"]
    assert np.array_equal(img.view(np.uint32), base.view(np.uint32))


def _eight_objects_with_cones():
    """Eight objects, wide cones first in the list and seen from above (rays steeper than their sides: the cone's larger root
    first, cone.h:117), spheres and a cylinder that overlap them later in the list: the answers depend on list order."""
    b = scenes.SceneBuilder(background=(0.7, 0.8, 1.0))
    m = [b.lambertian((0.6, 0.5, 0.4)), b.metal((0.8, 0.8, 0.7), 0.1), b.dielectric(1.5), b.lambertian((0.2, 0.5, 0.7))]
    b.cone((0.0, 0.0, 0.0), 1.0, 2.0, m[0])
    b.cone((0.8, 0.2, 0.4), 1.5, 2.5, m[3])
    b.cone((-1.0, 0.1, -0.6), 0.8, 1.5, m[1])
    b.sphere((0.3, 0.6, 0.2), 0.7, m[2])
    b.sphere((-0.8, 0.5, 0.9), 0.6, m[1])
    b.cylinder2((1.2, 0.0, -0.8), 1.2, 0.5, m[0])
    b.xz_rect(-4, 4, -4, 4, -0.1, m[3])
    b.sphere((0.0, 1.4, 0.0), 0.3, m[0])
    b.camera((0.6, 7.0, 0.9), (0.0, 0.3, 0.0), (0, 0, -1), 40.0, 1.0, 0.05, 7.0)
    return b


def test_scene_of_one_leaf(ctx):
    """A scene of at most eight objects is one leaf, and k_paths starts at it without a root visit or slab constants.
    Same image as with a real tree over the same objects (leaves of 2), from LDS and from HBM, as the oracle's, and the
    exact replay of list-order dependent cone hits -- which walks the tree from its root -- still runs."""
    b = _eight_objects_with_cones()
    w, h, spp, depth = 96, 96, 4, 12
    ctx.upload(b.desc())
    nodes, order, _ = api.bvh_build_host(b.desc())
    assert len(nodes) == 1 and nodes[0]["c"][0] < 0  # one leaf under the root
    base, st = ctx.render(w, h, spp, depth)
    assert st["exact_replays"] > 0
    ref, rays = util.oracle_scene(b).render(w, h, spp, depth)
    assert st["rays"] == rays
    frac, rmse = _image_close(base, ref)
    assert frac < 2e-3 and rmse < 2e-3, (frac, rmse)
    for kw in (dict(flags=api.RENDER_NO_LDS_SCENE), dict(flags=api.RENDER_SPLIT_KERNELS)):
        img, s2 = ctx.render(w, h, spp, depth, **kw)
        assert s2["rays"] == rays and np.array_equal(img.view(np.uint32), base.view(np.uint32)), kw
    with ctx.tuned(leaf_max=2):
        ctx.upload(b.desc())
    for flags in (0, api.RENDER_NO_LDS_SCENE):
        img, s2 = ctx.render(w, h, spp, depth, flags=flags)
        assert s2["rays"] == rays and np.array_equal(img.view(np.uint32), base.view(np.uint32)), flags


def test_depth_beyond_the_ring_kernels_bounce_counter(ctx):
    """k_paths_ring keeps a path's bounce count in 10 bits of its id word; a render with max_depth above 1023 is k_paths'.
    In the closed Cornell box paths do run long (a render at depth 1023 traces more rays than one at depth 20), but none
    reaches a thousand bounces: depth 1023 (k_paths_ring) and depth 2000 (k_paths) give the same image, bit for bit."""
    ctx.upload(scenes.cornell_scene().desc())
    a, sa = ctx.render(48, 48, 8, 1023)
    b, sb = ctx.render(48, 48, 8, 2000)
    c, sc = ctx.render(48, 48, 8, 20)
    assert (sa["kernel"], sb["kernel"], sc["kernel"]) == (4, 1, 4)
    assert sa["rays"] == sb["rays"] > sc["rays"]
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_render_is_deterministic_and_seeded(ctx):
    ctx.upload(scenes.random_spheres_scene().desc())
    a, _ = ctx.render(160, 90, 3, 50)
    b, _ = ctx.render(160, 90, 3, 50)
    c, _ = ctx.render(160, 90, 3, 50, seed=7)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert not np.array_equal(a, c)


# ------------------------------------------------------------------------------------------------
# ties and degenerate objects
# ------------------------------------------------------------------------------------------------
def test_coincident_objects_later_list_entry_wins(ctx):
    """hittable_list.h:93-95 accepts a hit at t == closest_so_far, so of coincident surfaces the LATER
    list entry is reported; the tree must not change that whatever leaf the copies land in."""
    b = scenes.SceneBuilder()
    m = [b.lambertian((0.1 * k, 0.5, 0.5)) for k in range(4)]
    rng = np.random.Generator(np.random.PCG64(5))
    centers = rng.random((40, 3)) * [12, 4, 12]
    for rep in range(3):              # three copies of every sphere, interleaved through the list
        for k, c in enumerate(centers):
            b.sphere(c, 0.6, m[(k + rep) % 4])
    b.xz_rect(-2, 14, -2, 14, -1.0, m[0])
    b.xz_rect(-2, 14, -2, 14, -1.0, m[1])  # two coincident floors
    b.camera((6, 2, -14), (6, 2, 6), (0, 1, 0), 40.0, 1.0, 0.0, 5.0)
    ctx.upload(b.desc())
    osc = util.oracle_scene(b)
    o, d = util.mixed_rays(b, osc, 12000, seed=77)
    ref = osc.trace(o, d)
    got = ctx.trace(o, d)
    ok = ref["nan_flag"] == 0
    assert not (util.hits_equal(got, ref) & ok).any()
    hit = ref["prim"] != api.NO_PRIM
    sph = hit & (ref["prim"] < 120)
    assert sph.sum() > 500 and (ref["prim"][sph] >= 80).all()  # always the last copy
    floor = hit & (ref["prim"] >= 120)
    assert floor.sum() > 100 and (ref["prim"][floor] == 121).all()


def test_degenerate_objects(ctx):
    """Zero-radius spheres, zero-area rects, a zero-height cylinder and a negative-radius sphere (the
    reference's hollow-glass idiom) load and trace exactly like the scan."""
    b = scenes.SceneBuilder()
    m = b.lambertian((0.5, 0.5, 0.5))
    g = b.dielectric(1.5)
    b.sphere((0, 0, 0), 1.0, g)
    b.sphere((0, 0, 0), -0.9, g)
    b.sphere((3, 0, 0), 0.0, m)
    b.xy_rect(1, 1, -1, 1, 2.0, m)
    b.yz_rect(-1, 1, 0.5, 0.5, -2.0, m)
    b.cylinder2((-3, -1, 0), 0.0, 0.7, m)
    b.cylinder2((-3, -1, 3), 2.0, 0.7, m)
    b.xz_rect(-6, 6, -6, 6, -1.0, m)
    b.camera((0, 1, 8), (0, 0, 0), (0, 1, 0), 40.0, 1.0, 0.0, 8.0)
    ctx.upload(b.desc())
    osc = util.oracle_scene(b)
    o, d = util.mixed_rays(b, osc, 8000, seed=31)
    ref = osc.trace(o, d)
    got = ctx.trace(o, d)
    ok = ref["nan_flag"] == 0
    assert not (util.hits_equal(got, ref) & ok).any()
    img_ref, rays = osc.render(64, 48, 4, 8)
    img, st = ctx.render(64, 48, 4, 8)
    frac, rmse = _image_close(img, img_ref)
    assert st["rays"] == rays and frac < 2e-3 and rmse < 2e-3


def _far_field(kind, n, rng, center):
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
            lo, hi = c[k] - r, c[k] + r
            if k // 4 % 3 == 0:
                b.xy_rect(lo[0], hi[0], lo[1], hi[1], c[k][2], m)
            elif k // 4 % 3 == 1:
                b.xz_rect(lo[0], hi[0], lo[2], hi[2], c[k][1], m)
            else:
                b.yz_rect(lo[1], hi[1], lo[2], hi[2], c[k][0], m)
    b.camera(tuple(center + [0, 0, 100]), tuple(center), (0, 1, 0), 40.0, 1.0, 0.0, 100.0)
    return b, c


@pytest.mark.parametrize("kind", ["sphere", "mixed", "sphere enclosed", "cyl enclosed"])
@pytest.mark.parametrize("center", [(0.0, 0.0, 0.0), (5000.0, 300.0, -2000.0)])
def test_far_field_hits_outside_the_exact_surface(ctx, kind, center):
    """Small objects seen from 10..3000 units away along unnormalised directions of length 0.1..300: the
    reference's fp32 quadratics (sphere.h:85-101, cylinder.h:96-123, cone.h:101-129) then accept roots
    whose points lie visibly outside the exact surfaces, or hit objects the ray geometrically misses.
    The tree must not cull them: every hit is still the scan's, bit for bit.  The "enclosed" variants put
    a sphere of radius 4000 around everything, so that the origins lie within the scene's bounds, and trace
    as the render does (ray cone narrowed per origin)."""
    rng = np.random.Generator(np.random.PCG64(11))
    n, nr = 1500, 60000
    enclosed = kind.endswith(" enclosed")
    kind = kind.split()[0]
    b, c = _far_field(kind, n, rng, np.asarray(center))
    if enclosed:
        b.sphere(center, 4000.0, 0)
    ctx.upload(b.desc())
    osc = util.oracle_scene(b)
    tgt = c[rng.integers(0, n, nr)] + (rng.random((nr, 3)) - 0.5) * 1.2
    dist = 10 ** (1 + rng.random(nr) * 2.5)
    dirn = rng.standard_normal((nr, 3))
    dirn /= np.linalg.norm(dirn, axis=1, keepdims=True)
    o = (tgt - dirn * dist[:, None]).astype(np.float32)
    d = (dirn * (10 ** (rng.random(nr) * 3.5 - 1))[:, None]).astype(np.float32)
    ref = osc.trace(o, d)
    got = ctx.trace(o, d, mode=api.TRACE_BVH_IN_SCENE if enclosed else api.TRACE_BVH)
    ok = ref["nan_flag"] == 0
    bad = util.hits_equal(got, ref) & ok
    assert not bad.any(), (kind, center, int(bad.sum()))
    assert ok.mean() > 0.99 and ((ref["prim"] != api.NO_PRIM) & (ref["prim"] < n)).sum() > 5000
    if kind == "sphere":  # the regime is exercised: hit points well outside the exact spheres do occur
        hit = (ref["prim"] != api.NO_PRIM) & (ref["prim"] < n)
        a = b.arrays()["prim_param"][ref["prim"][hit]]
        excess = np.linalg.norm(ref["p"][hit].astype(np.float64) - a[:, :3], axis=1) - a[:, 3]
        assert (excess > 0.05 * a[:, 3]).sum() > 50


@pytest.mark.parametrize("slopes", [(0.05, 0.1, 0.3), (1.0, 3.0), (10.0, 20.0), (0.05, 1.0, 20.0)])
@pytest.mark.parametrize("center", [(0.0, 0.0, 0.0), (5000.0, 300.0, -2000.0)])
def test_rays_grazing_cones_from_afar(ctx, slopes, center):
    """The adversarial case of the cone's quadratic (cone.h:101-129): thin / tall and flat / wide cones (radius / height
    0.05 .. 20), near the origin and 5000 units from it, 10^5 rays that graze their sides from 10 .. 10^4 units away, a
    third of them along the side itself where the leading coefficient cancels.  The rounding error of the
    discriminant is amplified by 1 / sqrt(alpha), alpha = |a| / (|d_xz|^2 + tangens d_y^2), so accepted hits lie farther
    outside the exact surface than for any other object.  Every answer outside the flagged set (alpha < 2^-6) must still
    be the scan's, bit for bit, from the tree, from the GPU's own list scan, and the slope the tree gives its rays must
    cover what these hits need with a margin (measured: at most 0.59 of it)."""
    rng = np.random.Generator(np.random.PCG64(3))
    b, cones = util.cone_field(slopes, 200, center, rng)
    ctx.upload(b.desc())
    osc = util.oracle_scene(b)
    o, d = util.grazing_cone_rays(cones, 100_000, rng)
    ref = osc.trace(o, d)
    ok = ref["nan_flag"] == 0
    hit = ref["prim"] != api.NO_PRIM
    assert (hit & ok).sum() > 40_000 and (hit & ~ok).sum() > 1000  # the regime is exercised, flagged rays included
    for mode in (api.TRACE_BVH, api.TRACE_LINEAR):
        got = ctx.trace(o, d, mode=mode)
        bad = util.hits_equal(got, ref) & ok
        assert not bad.any(), (slopes, center, mode, int(bad.sum()))
    need = util.needed_cone_slope(b.arrays(), o, d, ref)
    kappa = api.scene_ray_cone(b.desc())["kappa_max"]
    assert need[ok].max() > 0.02 * kappa  # hits outside the exact volume do occur ...
    assert need[ok].max() < 0.75 * kappa, (slopes, center, need[ok].max(), kappa)  # ... and the slope covers them with room


def _noise_level_discriminant(arr, prim, o, d):
    """|discriminant| / half_b^2 of object `prim` for ray (o, d), in float64, after sphere.h:85-93, cylinder.h:96-106 and
    cone.h:101-113; inf for rects (no quadratic)."""
    t, q = arr["prim_type"][prim], arr["prim_param"][prim].astype(np.float64)
    o, d = o.astype(np.float64), d.astype(np.float64)
    if t in (api.XY_RECT, api.XZ_RECT, api.YZ_RECT):
        return np.inf
    oc = o - q[:3]
    dd = d.copy()
    if t != api.SPHERE:
        oc[1] = 0.0
        dd[1] = 0.0
    a, hb = dd @ dd, oc @ dd
    if t == api.SPHERE:
        c = oc @ oc - q[3] * q[3]
    elif t == api.CYLINDER:
        c = oc @ oc - q[4] * q[4]
    else:
        D = q[3] - o[1] + q[1]
        a, hb, c = a - q[5] * d[1] * d[1], hb + q[5] * d[1] * D, oc @ oc - q[5] * D * D
    return abs(hb * hb - a * c) / max(hb * hb, 1e-300)


@pytest.mark.parametrize("which", ["mixed_100k", "spheres_1m"])
def test_geometric_cull_tolerance(ctx, which):
    """PT_RENDER_GEOMETRIC_CULL / PT_TRACE_BVH_GEOMETRIC on the two scenes it is for: boxes culled by the objects' exact
    geometry instead of against the ray's cone.  The stated tolerance: hit / miss, list index, t, normal and point
    bit-exact on at least 99.99 % of the rays a render traces (camera rays and two generations of scattered rays of
    the 4K view), and every ray that differs is one whose lost (or changed) object has a discriminant that is rounding
    noise for that ray, |discriminant| < 1e-4 * half_b^2, or whose t differs by less than 4 ulp.  The 100k mixed objects
    (sizes 0.2-1 in a 200-unit box) are within it; the 1M spheres are not (see below).  The exact traversal
    (bit-equal to the oracle's list scan on the committed batches of these scenes) is the reference here: the scan
    itself would take hours on a million rays times a million spheres."""
    b = scenes.mixed_scene() if which == "mixed_100k" else scenes.sphere_slab_scene()
    arr = b.arrays()
    ctx.upload(b.desc())
    rng = np.random.Generator(np.random.PCG64(29))
    cam = api.camera_fields(api.camera_look_at(*b.camera_args))
    n = 400_000
    u, v = rng.random((n, 1), dtype=np.float32), rng.random((n, 1), dtype=np.float32)
    o = np.tile(cam["origin"], (n, 1)).astype(np.float32)
    d = (cam["lower_left_corner"] + u * cam["horizontal"] + v * cam["vertical"] - cam["origin"]).astype(np.float32)
    os_, ds_ = [o], [d]
    for _ in range(2):  # what scatter produces: from the hit point along normal + a random unit vector
        h = ctx.trace(os_[-1], ds_[-1], mode=api.TRACE_BVH_IN_SCENE)
        ok = h["prim"] != api.NO_PRIM
        r = rng.standard_normal((int(ok.sum()), 3)).astype(np.float32)
        r /= np.linalg.norm(r, axis=1, keepdims=True).astype(np.float32)
        os_.append(h["p"][ok].astype(np.float32))
        ds_.append((h["n"][ok] + r).astype(np.float32))
    o, d = np.ascontiguousarray(np.concatenate(os_)), np.ascontiguousarray(np.concatenate(ds_))
    ref = ctx.trace(o, d, mode=api.TRACE_BVH_IN_SCENE)
    got = ctx.trace(o, d, mode=api.TRACE_BVH_GEOMETRIC)
    bad = util.hits_equal(got, ref)
    assert len(o) > 600_000 and (ref["prim"] != api.NO_PRIM).mean() > 0.3
    if which == "mixed_100k":
        assert bad.mean() <= 1e-4, (which, int(bad.sum()), len(o))
    else:
        # Not on the 1M spheres: radius 0.05-0.2 seen from up to 600 units away, where the reference's discriminant noise
        # (about 1e-3 of the distance) exceeds the spheres themselves.  About 1.5 % of the rays lose such a hit: the flag
        # is outside the tolerance on this scene, which is why the ray cone is the default.  What it loses is still noise.
        assert 1e-3 < bad.mean() < 5e-2, (which, int(bad.sum()), len(o))
    for i in np.flatnonzero(bad)[:400]:
        if got["prim"][i] == ref["prim"][i] and abs(int(got["t"][i].view(np.int32)) - int(ref["t"][i].view(np.int32))) < 4:
            continue
        assert ref["prim"][i] != api.NO_PRIM, i  # the geometric traversal only ever loses hits
        noise = _noise_level_discriminant(arr, int(ref["prim"][i]), o[i], d[i])
        assert noise < 1e-4, (which, i, int(ref["prim"][i]), noise)
    # and a render: the same frame through both, equal except where such a hit was lost along a path
    a, sa = ctx.render(640, 360, 2, 50)
    g, sg = ctx.render(640, 360, 2, 50, flags=api.RENDER_GEOMETRIC_CULL)
    differ = (a != g).any(axis=2).mean()
    lim = 2e-3 if which == "mixed_100k" else 0.2
    assert differ <= lim and abs(sa["rays"] - sg["rays"]) <= lim * 0.1 * sa["rays"], (differ, sa["rays"], sg["rays"])


# ------------------------------------------------------------------------------------------------
# shapes and sizes
# ------------------------------------------------------------------------------------------------
def test_empty_ray_batch(ctx):
    ctx.upload(scenes.default_scene().desc())
    got = ctx.trace(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32))
    assert len(got["t"]) == 0


@pytest.mark.parametrize("w,h,spp,depth", [(1, 1, 1, 1), (1, 1, 7, 5), (37, 13, 3, 5), (65, 1, 2, 50), (1, 129, 2, 3), (64, 64, 1, 1)])
def test_odd_image_shapes(ctx, w, h, spp, depth):
    b = scenes.default_scene()
    ctx.upload(b.desc())
    osc = util.oracle_scene(b)
    ref, rays = osc.render(w, h, spp, depth)
    got, st = ctx.render(w, h, spp, depth)
    assert st["rays"] == rays and st["primary_rays"] == w * h * spp
    frac, rmse = _image_close(got, ref)
    assert frac <= (0.0 if w * h < 500 else 2e-3) and rmse < 2e-3


@pytest.mark.parametrize("w,h,spp", [(4, 4, 3), (100, 75, 5), (9, 7, 300)])
def test_small_frames_through_the_streaming_pipeline(ctx, w, h, spp):
    """Frames of fewer pixels than one tile of the streaming pipeline's path numbering (64), and frames whose pixel
    count is not a multiple of it, rendered with the scene in HBM (k_paths_stream): one band per pass whatever the
    pixel count, the default pipeline's image bit for bit."""
    ctx.upload(scenes.random_spheres_scene().desc())
    base, st = ctx.render(w, h, spp, 50)
    img, s2 = ctx.render(w, h, spp, 50, flags=api.RENDER_NO_LDS_SCENE)
    assert s2["kernel"] == 2 and s2["rays"] == st["rays"]
    assert s2["passes"] == (spp + 255) // 256  # one band: a pass per 256 samples, no tail band
    assert np.array_equal(img.view(np.uint32), base.view(np.uint32))
    with ctx.tuned(pipeline=1):  # the same kernel chosen by the tuning field
        img, s3 = ctx.render(w, h, spp, 50, samples_per_pass=2)
    assert s3["passes"] == (spp + 1) // 2 and np.array_equal(img.view(np.uint32), base.view(np.uint32))


def test_render_into_a_pinned_host_buffer(ctx):
    """pt_render fills pageable and page-locked (pt_host_alloc) buffers alike; the latter is written by the device
    directly.  A frame larger than the staging chunks (4 MiB) goes through several of them."""
    ctx.upload(scenes.default_scene().desc())
    w, h = 800, 450  # 4.3 MB: two staging chunks
    want, st = ctx.render(w, h, 2, 5)
    with api.PinnedFilm(h, w) as film:
        film.array[:] = -1.0
        got, s2 = ctx.render(w, h, 2, 5, out=film.array)
        assert s2["rays"] == st["rays"] and np.array_equal(got.view(np.uint32), want.view(np.uint32))
    again, _ = ctx.render(w, h, 2, 5)
    assert np.array_equal(again.view(np.uint32), want.view(np.uint32))
    small, _ = ctx.render(31, 17, 2, 5)  # the kept device film is larger than this frame
    ref, _ = ctx.render(31, 17, 2, 5, flags=api.RENDER_SPLIT_KERNELS)
    assert np.array_equal(small.view(np.uint32), ref.view(np.uint32))


def test_many_samples_span_several_passes_and_slices(ctx):
    """spp larger than one slice holds (and than one pass of the split pipeline at a small budget):
    the per-pixel sum is taken in sample order across passes, like main.cu:88-93."""
    b = scenes.cornell_scene()
    ctx.upload(b.desc())
    osc = util.oracle_scene(b)
    ref, rays = osc.render(24, 16, 700, 5)
    got, st = ctx.render(24, 16, 700, 5)
    assert st["rays"] == rays and st["passes"] >= 2
    frac, rmse = _image_close(got, ref)
    assert frac < 5e-3 and rmse < 1e-3, (frac, rmse)
    again, _ = ctx.render(24, 16, 700, 5, flags=api.RENDER_SPLIT_KERNELS)
    np.testing.assert_allclose(again, got, rtol=1e-5, atol=1e-6)


def test_sample_offset_is_the_absolute_sample_index(ctx):
    b = scenes.default_scene()
    ctx.upload(b.desc())
    osc = util.oracle_scene(b)
    ref, _ = osc.render(80, 45, 2, 5, sample0=1000003)
    got, _ = ctx.render(80, 45, 2, 5, sample_offset=1000003)
    frac, rmse = _image_close(got, ref)
    assert frac < 2e-3 and rmse < 2e-3


def test_render_device_accumulates(ctx):
    """pt_render_device ADDS into the caller's buffer: two calls over disjoint sample ranges leave the
    sum a single call over both ranges gives (the contract the multi-GPU reduction relies on)."""
    import torch

    ctx.upload(scenes.default_scene().desc())
    w, h = 96, 54
    buf = torch.zeros(h, w, 3, dtype=torch.float32, device="cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    ctx.render_device(api.Context.params(w, h, 3, 5, sample_offset=0), buf.data_ptr(), stream=s)
    ctx.render_device(api.Context.params(w, h, 2, 5, sample_offset=3), buf.data_ptr(), stream=s)
    torch.cuda.synchronize()
    one, _ = ctx.render(w, h, 5, 5)
    np.testing.assert_allclose(buf.cpu().numpy(), one, rtol=1e-5, atol=1e-6)


# ------------------------------------------------------------------------------------------------
# large scenes: the tree is deep, the scene lives in HBM
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("which", ["mixed_100k", "spheres_1m"])
def test_large_scene_trace_bit_exact(ctx, which):
    if which == "mixed_100k":
        b, n = scenes.mixed_scene(), 1500          # benchmark configuration 4's scene (cones included)
    else:
        b, n = scenes.sphere_slab_scene(), 300     # benchmark configuration 5's scene
    ctx.upload(b.desc())
    osc = util.oracle_scene(b)
    o, d = util.mixed_rays(b, osc, n, seed=55)
    ref = osc.trace(o, d)
    ok = ref["nan_flag"] == 0
    for mode in (api.TRACE_BVH, api.TRACE_BVH_IN_SCENE):
        got = ctx.trace(o, d, mode=mode)
        bad = util.hits_equal(got, ref) & ok
        assert not bad.any(), (which, mode, int(bad.sum()))
        assert (got["flags"] & 1).sum() == 0  # no ray needed the full list scan
    assert (ref["prim"] != api.NO_PRIM).mean() > 0.2


@pytest.mark.parametrize("which", ["mixed_100k", "spheres_1m"])
def test_large_scene_render_crop(ctx, which):
    """Benchmark configurations 4 and 5: their scenes at 4K, 1 spp; a crop against the oracle (which renders only
    those pixels: the stream is keyed by the absolute pixel index), finite images and sane counters for
    the full frame."""
    if which == "mixed_100k":
        b, (x0, y0, w, h) = scenes.mixed_scene(), (1900, 1100, 24, 12)
    else:
        b, (x0, y0, w, h) = scenes.sphere_slab_scene(), (1912, 1500, 12, 6)
    ctx.upload(b.desc())
    osc = util.oracle_scene(b)
    got, st = ctx.render(3840, 2160, 1, 50)
    ref, _ = osc.render(3840, 2160, 1, 50, region=(x0, y0, w, h))
    frac, rmse = _image_close(got[y0:y0 + h, x0:x0 + w], ref)
    print(which, "crop: fraction of pixels beyond 2e-5", frac, "relative RMSE", rmse, "list scans", st["exact_fallbacks"], "of", st["rays"], "rays")
    assert frac == 0 and rmse < 1e-5, (which, frac, rmse)
    # the list scan answers next to nothing: on the mixed objects the few rays for which a cone's nearly cancelled quadratic
    # yields an acceptable root (a few in 10^8), on the spheres none
    assert np.isfinite(got).all() and st["exact_fallbacks"] <= (1e-6 * st["rays"] if which == "mixed_100k" else 0) and st["rays"] > st["primary_rays"]


@pytest.mark.parametrize("n", [30_000, 1_000_000])
def test_wide_and_binary_nodes_agree_bitwise(ctx, n):
    """The four-wide quantised tree only prunes: a field of spheres rendered through it (the default for cone-free
    scenes in HBM) and through the binary tree it was derived from gives the same image, bit for bit, over some
    million rays; so does a tree with leaves of three objects."""
    b = scenes.sphere_slab_scene(n=n)
    w, h, spp = (640, 360, 3) if n < 100_000 else (960, 540, 1)
    ctx.upload(b.desc())
    base, st = ctx.render(w, h, spp, 50)
    assert st["rays"] > 1.5 * w * h * spp and st["exact_fallbacks"] == 0
    for tune in (dict(wide_nodes=0), dict(leaf_max=3)):
        with ctx.tuned(**tune):
            ctx.upload(b.desc())
        img, s2 = ctx.render(w, h, spp, 50)
        assert s2["rays"] == st["rays"], tune
        assert np.array_equal(img.view(np.uint32), base.view(np.uint32)), tune


def test_path_through_a_nan_hit_is_answered_by_the_list(ctx):
    """A hit at t = NaN: the reference's comparisons accept it (aarect.h:364-371 tests `t < t_min || t > t_max` and
    the bounds the same way, Q9), and from then on hittable_list::hit accepts every later entry the ray's line
    crosses, whatever its distance, until one of them brings the running t back to a number.  No box test can
    follow that, so every kernel hands such a ray to the list scan (counted in exact_fallbacks).  The case is built,
    not found: a camera whose field of view underflows (1e-30 degrees) sends every ray from exactly (15, 15, 40)
    along exactly (0, 0, -10) -- the viewport's extent is absorbed by the coordinates -- and the first list entry is
    an xz_rect in the plane y = 15, in front of everything else: (k - o.y) / d.y = 0 / 0."""
    src = scenes.mixed_scene(n=3000, extent=30.0)
    a = {k: v.copy() for k, v in src.arrays().items()}
    a["prim_type"][0] = api.XZ_RECT
    a["prim_param"][0] = [14, 16, 38, 39, 15, 0]
    b = scenes.ArrayScene(a, src.background, ((15.0, 15.0, 40.0), (15.0, 15.0, 0.0), (0, 1, 0), 1e-30, 1.0, 0.0, 10.0))
    osc = util.oracle_scene(b)
    o = np.array([[15.0, 15.0, 40.0]], dtype=np.float32)
    d = np.array([[0.0, 0.0, -10.0]], dtype=np.float32)
    want = osc.trace(o, d)
    assert want["nan_flag"][0] == 1 and want["prim"][0] not in (0, api.NO_PRIM)  # a later entry took over from the NaN hit
    ctx.upload(b.desc())
    for mode in (api.TRACE_BVH, api.TRACE_BVH_IN_SCENE, api.TRACE_LINEAR):
        got = ctx.trace(o, d, mode=mode)
        assert not util.hits_equal(got, want).any(), mode
        assert mode == api.TRACE_LINEAR or got["flags"][0] & 1
    w, h, spp, depth = 8, 8, 2, 50
    ref, rays = osc.render(w, h, spp, depth)
    got, st = ctx.render(w, h, spp, depth)  # 3000 objects: the streaming kernel
    frac, rmse = _image_close(got, ref)
    assert frac == 0 and rmse < 2e-3 and st["rays"] == rays, (frac, rmse, st["rays"], rays)
    assert st["exact_fallbacks"] >= w * h * spp and np.isfinite(got).all()
    with ctx.tuned(pipeline=0):  # the queue-based kernel, whose traversal is bvh_nearest
        img, s2 = ctx.render(w, h, spp, depth)
    assert s2["rays"] == rays and s2["exact_fallbacks"] >= w * h * spp and np.array_equal(img.view(np.uint32), got.view(np.uint32))


def _with_cancelling_cones(src, positions, tangens=0.25):
    """A copy of a scene's arrays with cones of the given tangens (radius / height squared, cone.h:30) at the given
    list positions, far outside the rest of the scene."""
    a = {k: v.copy() for k, v in src.arrays().items()}
    for k, pos in enumerate(positions):
        a["prim_type"][pos] = api.CONE
        r = np.sqrt(np.float32(tangens)) * np.float32(2.0)
        a["prim_param"][pos] = [500.0 + 10.0 * k, -300.0, 800.0, 2.0, r, tangens]
    return a


@pytest.mark.parametrize("where", ["first", "middle", "last", "two"])
def test_cone_whose_leading_coefficient_cancels(ctx, where):
    """cone.h:108 computes a = |d_xz|^2 - tangens d_y d_y.  For a direction whose two terms are the same float a is
    exactly 0, the quadratic's roots (cone.h:116-120) are -inf and 0 / 0, and the NaN passes every test of the form
    `x < bound` (the range tests, the clip of cone.h:121-123): the cone is hit at t = NaN by a ray that passes nowhere
    near it, and hittable_list::hit (hittable_list.h:87-101) goes on from a running t of NaN.  The answer is then the
    list's from that entry on.  The directions here are (3, 10, 4), (5, 26, 12), (8, 34, 15) with signs, swaps and
    exact scalings: |d_xz|^2 = d_y^2 / 4 exactly, against cones of tangens 1/4 that stand 900 units from everything
    else.  Tree walk, list scan and oracle must agree on every ray, NaN hits included."""
    n = 400
    src = scenes.mixed_scene(n=n, extent=20.0, seed=3)
    positions = {"first": [0], "middle": [n // 2], "last": [n - 1], "two": [5, n - 7]}[where]
    a = _with_cancelling_cones(src, positions)
    b = scenes.ArrayScene(a, src.background, src.camera_args)
    osc = util.oracle_scene(b)
    ctx.upload(b.desc())
    rng = np.random.Generator(np.random.PCG64(17))
    m = 30000
    base = np.array([[3, 10, 4], [5, 26, 12], [8, 34, 15]], dtype=np.float32)[rng.integers(0, 3, m)]
    swap = rng.random(m) < 0.5
    base[swap] = base[swap][:, [2, 1, 0]]
    base *= rng.choice([-1.0, 1.0], (m, 3)).astype(np.float32)
    d = (base * rng.choice([0.125, 0.5, 1.0, 3.0, 64.0], (m, 1)).astype(np.float32)).astype(np.float32)
    o = (rng.random((m, 3)) * 20.0).astype(np.float32)
    # and as many ordinary rays
    o2, d2 = util.box_rays((0, 0, 0), (20, 20, 20), m, rng)
    o, d = np.concatenate([o, o2]), np.concatenate([d, d2])
    want = osc.trace(o, d)
    assert (want["nan_flag"][:m] & 1).all()  # every one of the built rays meets the NaN
    nan_final = np.isnan(want["t"][:m])
    if where == "last":
        assert nan_final.all() and (want["prim"][:m] == n - 1).all()  # whatever else the ray hits: the last entry's NaN stands
    else:
        assert 0 < nan_final.sum() < m  # some rays cross a later entry (which then is the answer), some do not
    for mode in (api.TRACE_LINEAR, api.TRACE_BVH, api.TRACE_BVH_IN_SCENE):
        got = ctx.trace(o, d, mode=mode)
        bad = util.hits_equal(got, want, nan_equal=True)
        assert not bad[:m].any(), (mode, int(bad[:m].sum()))
        assert not (bad[m:] & (want["nan_flag"][m:] == 0)).any(), mode
        assert mode == api.TRACE_LINEAR or (got["flags"][:m] & 1).all()  # answered by the list, from that entry on


def test_cone_whose_leading_coefficient_all_but_cancels(ctx):
    """The neighbour of the case above: direction (0.59999996, 1, 0) against a cone of tangens 0.36 gives
    a = |d_xz|^2 - tangens d_y^2 = -2 units in the last place of its terms.  The roots are then the rounding noise of
    -half_b -+ sqrtd over that -- 32, 64 -- and for a few origins in 10^5 such a number passes the range test and the clip
    (cone.h:116-123): the cone is hit at a point tens of units away from it, outside any box a tree could give it.  The
    rays are found with the oracle on the cone alone (hits farther than 5 units from a cone of height 1), then traced
    through a scene of 400 objects with the cone in the middle of the list: oracle, list scan and tree walk must agree."""
    rng = np.random.Generator(np.random.PCG64(3))
    lone = scenes.SceneBuilder()
    lone.cone((0.0, 0.0, 0.0), 1.0, 0.6, lone.lambertian((0.5, 0.5, 0.5)))
    lone.camera((0, 0, 5), (0, 0, 0), (0, 1, 0), 40.0, 1.0, 0.0, 5.0)
    m = 2000000
    o = ((rng.random((m, 3)) - 0.5) * 200.0).astype(np.float32)
    d = np.tile(np.array([[0.59999996, 1.0, 0.0]], dtype=np.float32), (m, 1)) * rng.choice([-1.0, 1.0], (m, 1)).astype(np.float32)
    h = util.oracle_scene(lone).trace(o, d)
    far = (h["prim"] == 0) & (np.linalg.norm(h["p"], axis=1) > 5.0)
    assert far.sum() >= 10, int(far.sum())
    o, d = np.ascontiguousarray(o[far]), np.ascontiguousarray(d[far])
    n = 400
    src = scenes.mixed_scene(n=n, extent=20.0, seed=3)
    a = {k: v.copy() for k, v in src.arrays().items()}
    a["prim_param"][:, :3] += np.float32(500.0)  # out of the way of these rays (rect parameters are shifted too; any scene will do)
    a["prim_type"][n // 2] = api.CONE
    a["prim_param"][n // 2] = [0.0, 0.0, 0.0, 1.0, 0.6, np.float32(0.6) * np.float32(0.6)]
    b = scenes.ArrayScene(a, src.background, src.camera_args)
    want = util.oracle_scene(b).trace(o, d)
    assert (want["prim"] == n // 2).sum() >= 10 and (np.linalg.norm(want["p"][want["prim"] == n // 2], axis=1) > 5.0).all()
    ctx.upload(b.desc())
    for mode in (api.TRACE_LINEAR, api.TRACE_BVH, api.TRACE_BVH_IN_SCENE):
        got = ctx.trace(o, d, mode=mode)
        bad = util.hits_equal(got, want, nan_equal=True)
        assert not bad.any(), (mode, int(bad.sum()))


@pytest.mark.parametrize("n", [300, 3000])
def test_render_through_a_cancelling_cone(ctx, n):
    """The same in a render: a camera whose field of view underflows sends every primary ray along exactly the same
    direction, and a cone far behind the camera, in the middle of the list, has the tangens that cancels it.  The
    reference's image then shows the later half of the list only (and where a ray meets none of it, a path that goes on
    from a NaN point: every entry accepts such a ray, the last one stands, fifty bounces end in black).  300 objects
    are rendered from LDS (k_paths_ring, k_paths), 3000 by the streaming kernel; all against the oracle."""
    src = scenes.mixed_scene(n=n, extent=30.0, seed=9)
    cam = ((15.0, 15.0, 60.0), (12.0, 11.0, 60.0), (0, 1, 0), 1e-30, 1.0, 0.0, 10.0)  # every ray along exactly (-6, -8, 0)
    b0 = scenes.ArrayScene({k: v.copy() for k, v in src.arrays().items()}, src.background, cam)
    o, d = util.oracle_scene(b0).get_ray(np.float32(0.5), np.float32(0.5), (0.0, 0.0))
    d = np.asarray(d, dtype=np.float32)
    S = np.float32(d[0] * d[0]) + np.float32(d[2] * d[2])
    guess = np.float32(S / np.float32(d[1] * d[1]))
    tangens = None
    for step in range(-8, 9):  # the float whose term cancels |d_xz|^2 exactly, next to the quotient
        t = guess
        for _ in range(abs(step)):
            t = np.nextafter(t, np.float32(np.inf if step > 0 else -np.inf), dtype=np.float32)
        if np.float32(np.float32(t * d[1]) * d[1]) == S:
            tangens = t
    assert tangens is not None
    a = _with_cancelling_cones(src, [n // 2], tangens=float(tangens))
    b = scenes.ArrayScene(a, src.background, cam)
    osc = util.oracle_scene(b)
    w, h, spp, depth = 8, 8, 4, 50
    ref, rays = osc.render(w, h, spp, depth)
    ctx.upload(b.desc())
    first = osc.trace(np.asarray(o, dtype=np.float32).reshape(1, 3), d.reshape(1, 3))
    assert first["nan_flag"][0] & 1
    plain, _ = ctx.render(w, h, spp, depth, flags=0)
    frac, rmse = _image_close(plain, ref)
    assert frac == 0 and rmse < 2e-3, (frac, rmse)
    for tune in ({}, {"pipeline": 0}, {"pipeline": 1}):
        with ctx.tuned(**tune):
            got, st = ctx.render(w, h, spp, depth)
        assert st["rays"] == rays and np.isfinite(got).all(), (tune, st["rays"], rays)
        assert np.array_equal(got.view(np.uint32), plain.view(np.uint32)), tune
        assert st["exact_fallbacks"] + st["exact_replays"] >= w * h * spp, tune  # every primary ray went the special way


def test_random_scenes_tree_equals_scan(ctx):
    """A short run of tools/mb/fuzz_trace.py's generator: scenes of random size, mix, scale and distance from the origin,
    rays between random points, aimed from afar, grazing, and three generations off the hit points; the tree walk
    (both ray cones) must return the list scan's record for every ray.  (The long runs are in the design notes.)"""
    import importlib.util
    import os

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "mb", "fuzz_trace.py")
    spec = importlib.util.spec_from_file_location("fuzz_trace", path)
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    rng = np.random.Generator(np.random.PCG64(2024))
    rays = 0
    for _ in range(24):
        b, pos, info = fz.random_scene(rng)
        ctx.upload(b.desc())
        o, d = fz.rays_for(pos, info[2], info[5], 90000, rng)
        box_lo = (pos.min(0) - 0.5 * info[4]).astype(np.float32)
        box_hi = (pos.max(0) + 0.5 * info[4]).astype(np.float32)
        for gen in range(3):
            scan = ctx.trace(o, d, mode=api.TRACE_LINEAR)
            assert not fz.differing(ctx.trace(o, d), scan).any(), (info, gen)
            inside = ((o >= box_lo) & (o <= box_hi)).all(axis=1)
            assert not (fz.differing(ctx.trace(o, d, mode=api.TRACE_BVH_IN_SCENE), scan) & inside).any(), (info, gen)
            rays += len(o)
            hit = scan["prim"] != api.NO_PRIM
            if hit.sum() < 100:
                break
            r = rng.standard_normal((int(hit.sum()), 3))
            r /= np.linalg.norm(r, axis=1, keepdims=True)
            o = np.ascontiguousarray(scan["p"][hit], dtype=np.float32)
            d = np.ascontiguousarray(scan["n"][hit] + r, dtype=np.float32)
    assert rays > 2000000


def test_random_scenes_render_like_the_oracle(ctx):
    """A short run of tests/studies/fuzz_render.py's generator: random scenes of 1 to 20 000 objects of every kind, all
    four material classes, random cameras (some far away, some scenes far from the origin), 24 x 16 pixels x 4 spp at depth
    50.  The default pipeline against the oracle (every pixel within 2e-5, equal ray counts), the others against it."""
    import importlib.util
    import os

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "studies", "fuzz_render.py")
    spec = importlib.util.spec_from_file_location("fuzz_render", path)
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    rng = np.random.Generator(np.random.PCG64(77))
    w, h, spp, depth = 24, 16, 4, 50
    kernels = set()
    for _ in range(120):
        b, n = fz.random_scene(rng)
        ref, rays = util.oracle_scene(b).render(w, h, spp, depth)
        ctx.upload(b.desc())
        got, st = ctx.render(w, h, spp, depth)
        kernels.add(st["kernel"])
        frac, _ = _image_close(got, ref)
        assert frac == 0 and st["rays"] == rays, (n, st["kernel"], frac, st["rays"], rays)
        for tune, flags in (({"pipeline": 0}, 0), ({"pipeline": 1}, 0), ({}, api.RENDER_SPLIT_KERNELS)):
            with ctx.tuned(**tune):
                other, so = ctx.render(w, h, spp, depth, flags=flags)
            assert so["rays"] == rays and np.array_equal(np.nan_to_num(other, nan=-1.0), np.nan_to_num(got, nan=-1.0)), (n, tune, flags)
    assert {2, 4} <= kernels  # both default kernels were exercised


def test_mixed_objects_tree_equals_scan_on_every_ray(ctx):
    """The 100k mixed objects (25 000 cones of as many slopes): one scattered ray in 16 000 has a direction for which some
    cone's leading coefficient cancels exactly, wherever that cone stands.  The tree walk must return the list scan's
    record for every ray of three generations leaving the hit points of a million camera rays -- no mask."""
    b = scenes.mixed_scene()
    ctx.upload(b.desc())
    rng = np.random.Generator(np.random.PCG64(5))
    m = 1000000
    eye = np.array(b.camera_args[0], dtype=np.float64)
    o = np.ascontiguousarray(np.broadcast_to(eye, (m, 3)), dtype=np.float32)
    d = np.ascontiguousarray(rng.random((m, 3)) * 200.0 - eye, dtype=np.float32)
    nan_hits = 0
    for gen in range(4):
        tree = ctx.trace(o, d)
        scan = ctx.trace(o, d, mode=api.TRACE_LINEAR)
        bad = util.hits_equal(tree, scan, nan_equal=True)
        assert not bad.any(), (gen, int(bad.sum()))
        nan_hits += int(np.isnan(scan["t"]).sum())
        hit = (scan["prim"] != api.NO_PRIM) & ~np.isnan(scan["t"])
        r = rng.standard_normal((int(hit.sum()), 3))
        r /= np.linalg.norm(r, axis=1, keepdims=True)
        o = np.ascontiguousarray(scan["p"][hit], dtype=np.float32)
        d = np.ascontiguousarray(scan["n"][hit] + r, dtype=np.float32)
    assert nan_hits > 20  # (56 with this seed: the case is there)


@pytest.mark.parametrize("which", ["spheres", "mixed"])
def test_views_along_a_coordinate_axis(ctx, which):
    """A camera looking straight down the z axis through a 0.05 degree lens: every primary ray runs almost parallel
    to two coordinate planes, where the slab test's cone widens faster than the ray advances and the planes the
    ray meets last bound t from below instead of from above.  The image must be the oracle's, and the same bit for
    bit from every traversal: binary and four-wide nodes, with and without path queues."""
    if which == "spheres":
        src = scenes.sphere_slab_scene(n=3000, seed=5)
        a = src.arrays()
        a["prim_param"][1:, 0] *= np.float32(0.05)  # a 20 x 50 x 20 column of spheres
        a["prim_param"][1:, 2] *= np.float32(0.05)
        cam = ((0.0, 25.0, 300.0), (0.0, 25.0, 0.0), (0, 1, 0), 0.05, 1.0, 0.0, 300.0)
    else:
        src = scenes.mixed_scene(n=3000, extent=30.0)
        a = src.arrays()
        cam = ((15.0, 15.0, 200.0), (15.0, 15.0, 0.0), (0, 1, 0), 0.05, 1.0, 0.0, 200.0)
    b = scenes.ArrayScene(a, src.background, cam)
    w, h, spp, depth = 48, 48, 2, 8
    ctx.upload(b.desc())
    base, st = ctx.render(w, h, spp, depth)
    ref, rays = util.oracle_scene(b).render(w, h, spp, depth)
    frac, rmse = _image_close(base, ref)
    assert frac <= 0.01 and rmse < 5e-3 and st["rays"] == rays, (frac, rmse, st["rays"], rays)
    assert (ref.sum(axis=2) > 0).mean() > 0.5
    with ctx.tuned(pipeline=0):  # the queue-based kernel on the binary tree in HBM
        img, s2 = ctx.render(w, h, spp, depth)
    assert s2["rays"] == st["rays"] and np.array_equal(img.view(np.uint32), base.view(np.uint32))
    with ctx.tuned(wide_nodes=1 if which == "mixed" else 0):  # the node form that is not this scene's default
        ctx.upload(b.desc())
    img, s2 = ctx.render(w, h, spp, depth)
    assert s2["rays"] == st["rays"] and np.array_equal(img.view(np.uint32), base.view(np.uint32))
    o = np.tile(np.array(cam[0], dtype=np.float32), (6, 1))  # and the same through pt_trace_rays
    d = np.array([[0, 0, -1], [1e-4, 0, -1], [0, -1e-4, -1], [1e-3, 1e-3, -1], [-1e-5, 2e-4, -1], [0, 1e-6, -1]], dtype=np.float32)
    want = util.oracle_scene(b).trace(o, d)
    for mode in (api.TRACE_BVH, api.TRACE_BVH_IN_SCENE):
        assert not util.hits_equal(ctx.trace(o, d, mode=mode), want).any()


def test_group_of_one_gpu_goes_through_rccl(ctx):
    """pt_group on the GPUs this box has (one): contexts, streams, an RCCL communicator (ncclCommInitAll) and the
    in-place ncclAllReduce of the accumulation buffer all run; with a single rank the image is pt_render's, bit for
    bit.  More ranks cannot run here; what they compute is covered by the shard tests (sample ranges add up)."""
    b = scenes.random_spheres_scene()
    ctx.upload(b.desc())
    want, st = ctx.render(320, 180, 5, 50, sample_offset=3)
    with api.Group([0]) as g:
        assert g.size() == 1
        g.upload(b.desc())
        got, sg = g.render(320, 180, 5, 50, sample_offset=3)
        assert sg["rays"] == st["rays"] and sg["kernel"] == st["kernel"]
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        small, _ = g.render(64, 36, 2, 50)  # another frame size: the buffers are reallocated
        ref, _ = ctx.render(64, 36, 2, 50)
        assert np.array_equal(small.view(np.uint32), ref.view(np.uint32))
        for _ in range(3):  # the workers are reused call after call
            again, _ = g.render(64, 36, 2, 50)
            assert np.array_equal(again.view(np.uint32), ref.view(np.uint32))
        with api.PinnedFilm(36, 64) as film:  # a page-locked destination is written directly
            got, _ = g.render(64, 36, 2, 50, out=film.array)
            assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
        import ctypes

        ptr, sd = g.render_device(64, 36, 2, 50)  # the sum left on GPU 0: read it back with the runtime's own copy
        hip = ctypes.CDLL(None)  # the HIP runtime the library uses (loaded into the global namespace by the binding)
        hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        back = np.zeros((36, 64, 3), dtype=np.float32)
        assert ptr and hip.hipMemcpy(back.ctypes.data, ptr, back.nbytes, 2) == 0  # 2: device to host
        assert sd["rays"] > 0 and np.array_equal(back.view(np.uint32), ref.view(np.uint32))
    with pytest.raises(api.PtError):
        api.Group([0, 0])  # one rank per GPU


@pytest.mark.parametrize("seed", range(12))
def test_tiny_random_scenes(ctx, seed):
    """Scenes of 1 to 12 objects of every kind (up to eight objects: one leaf, rendered by the kernels without tree
    code; above: the smallest trees): the oracle's paths ray for ray, traces bit-exact."""
    n = 1 + (seed * 5) % 12
    b = scenes.mixed_scene(n=n, seed=100 + seed, extent=4.0, n_materials=6)
    osc = util.oracle_scene(b)
    ctx.upload(b.desc())
    o, d = util.mixed_rays(b, osc, 4000, seed=seed)
    ref = osc.trace(o, d)
    got = ctx.trace(o, d, mode=api.TRACE_BVH_IN_SCENE)
    assert not (util.hits_equal(got, ref) & (ref["nan_flag"] == 0)).any(), (seed, n)
    w, h, spp, depth = 48, 32, 3, 10
    img, st = ctx.render(w, h, spp, depth)
    want, rays = osc.render(w, h, spp, depth)
    assert st["rays"] == rays, (seed, n, st["rays"], rays)
    frac, rmse = _image_close(img, want)
    assert frac < 2e-3 and rmse < 2e-3, (seed, n, frac, rmse)
    alt, s2 = ctx.render(w, h, spp, depth, flags=api.RENDER_NO_LDS_SCENE)
    assert s2["rays"] == rays and np.array_equal(alt.view(np.uint32), img.view(np.uint32)), (seed, n)
