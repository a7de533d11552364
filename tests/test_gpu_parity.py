"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle.

Bars (fp32 path):
  * nearest hit (hit/miss, t, list index, normal, point, face flag): BIT-EXACT against the oracle's
    list-order scan on every ray of every batch (rays on which the oracle saw a NaN root are outside the
    contract and masked; none occur in these batches unless a test says so);
  * images: the GPU walks the same paths as the oracle (same counter-based stream), and differs only in
    the association of the throughput product (iterative vs the reference's recursion, main.cu:64):
    per-pixel |diff| <= 2e-5 * max(1, value) on the linear sums for >= 99.9 % of pixels (a path whose
    Russian-doll of products rounds differently can flip a rejection-sampling decision further down;
    such pixels are counted, not hidden) and a relative RMSE <= 1e-3.
"""
import os

import numpy as np
import pytest

from pathtracer_amd import api, scenes
from tests import util

pytestmark = pytest.mark.gpu

SCENES = {
    "default": scenes.default_scene,
    "cornell": scenes.cornell_scene,
    "random_spheres": scenes.random_spheres_scene,
    "mixed_1k": lambda: scenes.mixed_scene(n=1000, seed=3, extent=20.0, n_materials=64),
}


@pytest.fixture(scope="module", params=list(SCENES))
def loaded(request, ctx):
    b = SCENES[request.param]()
    ctx.upload(b.desc())
    return request.param, b, util.oracle_scene(b)


@pytest.mark.parametrize("mode", [api.TRACE_BVH, api.TRACE_BVH_IN_SCENE])
def test_trace_bvh_bit_exact(loaded, ctx, mode):
    """TRACE_BVH_IN_SCENE is the traversal as the render runs it (ray cone narrowed by the scene's bounds);
    these batches start at the camera, on the objects or between them, as a render's rays do."""
    name, b, osc = loaded
    o, d = util.mixed_rays(b, osc, 20000, seed=101)
    ref = osc.trace(o, d)
    got = ctx.trace(o, d, mode=mode)
    ok = ref["nan_flag"] == 0
    bad = util.hits_equal(got, ref) & ok
    assert ok.mean() > 0.999
    assert not bad.any(), "%s: %d of %d rays differ, first %s" % (name, bad.sum(), len(o), np.flatnonzero(bad)[:5])
    assert (ref["prim"] != api.NO_PRIM).mean() > 0.2  # the batch does exercise hits


def test_trace_linear_bit_exact(loaded, ctx):
    name, b, osc = loaded
    o, d = util.mixed_rays(b, osc, 4000, seed=202)
    ref = osc.trace(o, d)
    got = ctx.trace(o, d, mode=api.TRACE_LINEAR)
    ok = ref["nan_flag"] == 0
    bad = util.hits_equal(got, ref) & ok
    assert not bad.any(), "%s: %d rays differ" % (name, bad.sum())


def test_trace_t_range(loaded, ctx):
    """A caller-given [t_min, t_max] is honoured exactly like hittable_list::hit's arguments."""
    name, b, osc = loaded
    o, d = util.mixed_rays(b, osc, 3000, seed=303)
    for t_min, t_max in ((0.5, 3.0), (1e-3, 0.9), (2.0, 1e30)):
        ref = osc.trace(o, d, t_min=t_min, t_max=t_max)
        got = ctx.trace(o, d, t_min=t_min, t_max=t_max)
        ok = ref["nan_flag"] == 0
        assert not (util.hits_equal(got, ref) & ok).any(), (name, t_min, t_max)


def _image_close(got, ref):
    tol = 2e-5 * np.maximum(1.0, np.abs(ref))
    bad_px = (np.abs(got - ref) > tol).any(axis=2)
    rmse = float(np.sqrt(np.mean((got - ref) ** 2)) / max(1e-12, np.sqrt(np.mean(ref ** 2))))
    return bad_px.mean(), rmse


@pytest.mark.parametrize("depth", [5, 50])
def test_render_default_scene(ctx, depth):
    """benchmark configuration 1: scene() at 400x225, 16 spp; depth 5 is the reference's own (main.cu:93)."""
    b = scenes.default_scene()
    ctx.upload(b.desc())
    osc = util.oracle_scene(b)
    ref, ref_rays = osc.render(400, 225, 16, depth)
    got, st = ctx.render(400, 225, 16, depth)
    frac, rmse = _image_close(got, ref)
    assert st["rays"] == ref_rays or abs(st["rays"] - ref_rays) < 1e-4 * ref_rays
    assert frac < 1e-3 and rmse < 1e-3, (frac, rmse)


@pytest.mark.parametrize("name,w,h,spp,depth", [("cornell", 96, 96, 8, 5), ("random_spheres", 160, 90, 4, 50), ("mixed_1k", 96, 54, 4, 8)])
def test_render_small(ctx, name, w, h, spp, depth):
    b = SCENES[name]()
    ctx.upload(b.desc())
    osc = util.oracle_scene(b)
    ref, ref_rays = osc.render(w, h, spp, depth)
    got, st = ctx.render(w, h, spp, depth)
    frac, rmse = _image_close(got, ref)
    assert abs(st["rays"] - ref_rays) <= 1e-3 * ref_rays
    assert frac < 2e-3 and rmse < 2e-3, (name, frac, rmse)


def test_render_full_size_crop(ctx):
    """benchmark configuration 2 at its full 1920x1080 size (2 spp): a 48x32 crop against the oracle, which
    renders only those pixels (the stream is keyed by the absolute pixel index)."""
    b = scenes.random_spheres_scene()
    ctx.upload(b.desc())
    osc = util.oracle_scene(b)
    got, st = ctx.render(1920, 1080, 2, 50)
    x0, y0, w, h = 900, 380, 48, 32
    ref, _ = osc.render(1920, 1080, 2, 50, region=(x0, y0, w, h))
    frac, rmse = _image_close(got[y0:y0 + h, x0:x0 + w], ref)
    assert frac < 5e-3 and rmse < 5e-3, (frac, rmse)
    assert np.isfinite(got).all() and st["rays"] > st["primary_rays"]


def test_cornell_full_size_and_its_eight_way_sample_split(ctx):
    """Benchmark configuration 3 at its full size, 1024x1024 at 1024 spp, depth 50 (8 G rays): a 12x8 crop against the
    oracle, which renders only those pixels; and the single-GPU rehearsal of its 8-GPU split: eight calls of 128 spp at
    sample offsets 0, 128, ... (what GPU g of 8 renders) summed in float64 against the one 1024-spp call, with equal
    ray counts (the stream is keyed by the absolute sample index)."""
    b = scenes.cornell_scene()
    ctx.upload(b.desc())
    osc = util.oracle_scene(b)
    full, st = ctx.render(1024, 1024, 1024, 50)
    x0, y0, w, h = 640, 300, 12, 8
    ref, _ = osc.render(1024, 1024, 1024, 50, region=(x0, y0, w, h))
    frac, rmse = _image_close(full[y0:y0 + h, x0:x0 + w], ref)
    assert frac <= 0.02 and rmse < 2e-3, (frac, rmse)
    assert np.isfinite(full).all() and st["rays"] > 5 * st["primary_rays"] and st["kernel"] == 4  # k_paths_ring: the scene fits LDS
    total = np.zeros(full.shape, np.float64)
    rays = 0
    for g in range(8):
        part, sp = ctx.render(1024, 1024, 128, 50, sample_offset=128 * g)
        total += part
        rays += sp["rays"]
    assert rays == st["rays"]
    np.testing.assert_allclose(total, full, rtol=2e-5, atol=1e-4)


def test_sample_shards_add_up(ctx):
    """Sample-range shards (what each GPU of a node renders) sum to the unsharded image: the stream is
    keyed by the absolute sample index, so only the fp32 association of the per-pixel sum changes."""
    b = scenes.default_scene()
    ctx.upload(b.desc())
    full, st = ctx.render(200, 112, 8, 5)
    parts = [ctx.render(200, 112, 2, 5, sample_offset=k * 2) for k in range(4)]
    total = sum(p[0].astype(np.float64) for p in parts)
    assert sum(p[1]["rays"] for p in parts) == st["rays"]
    np.testing.assert_allclose(total, full, rtol=1e-5, atol=1e-6)


def test_pass_size_does_not_matter(ctx):
    """samples_per_pass only changes batching: the image is bit-identical."""
    b = scenes.cornell_scene()
    ctx.upload(b.desc())
    a, sa = ctx.render(64, 64, 6, 5, samples_per_pass=1)
    c, sc = ctx.render(64, 64, 6, 5, samples_per_pass=4)
    assert sa["rays"] == sc["rays"]
    assert np.array_equal(a.view(np.uint32), c.view(np.uint32))


def test_errors(ctx):
    fresh = api.Context(0)
    with pytest.raises(api.PtError) as e:
        fresh.render(8, 8, 1, 5)
    assert e.value.status == 2
    fresh.close()
    b = scenes.default_scene()
    ctx.upload(b.desc())
    with pytest.raises(api.PtError):
        ctx.render(0, 8, 1, 5)
    with pytest.raises(api.PtError):
        ctx.render(8, 8, 0, 5)


def _cone_scene(order):
    b = scenes.SceneBuilder()
    m = b.lambertian((0.5, 0.5, 0.5))
    for what in order:
        if what == "cone":
            b.cone((0, 0, 0), 2.0, 1.0, m)
        elif what == "low":
            b.xz_rect(-5, 5, -5, 5, 2.25, m)   # between the cone's real root (2.5) and mirror root (3.5) along the test ray
        elif what == "high":
            b.xz_rect(-5, 5, -5, 5, 3.0, m)    # beyond the mirror root
        elif what == "pad":
            for k in range(6):
                b.sphere((20 + 3 * k, 0, 0), 1.0, m)
    b.camera((0, 0, 5), (0, 0, 0), (0, 1, 0), 40.0, 1.0, 0.0, 5.0)
    return b


@pytest.mark.parametrize("order", [("cone", "low"), ("low", "cone"), ("cone", "high"), ("high", "cone"), ("pad", "low", "cone", "high"),
                                   ("pad", "cone", "low"), ("low", "pad", "cone", "pad"), ("cone", "pad", "low", "pad")])
def test_cone_list_order_dependence(ctx, order):
    """A steep ray from below through a cone: the reference's answer depends on what EARLIER list entries hit
    (cone.h:117-122 tries the larger root first).  The BVH path must reproduce it for every list order."""
    b = _cone_scene(order)
    ctx.upload(b.desc())
    osc = util.oracle_scene(b)
    rng = np.random.Generator(np.random.PCG64(9))
    o = np.concatenate([[[0.25, -1, 0]], (rng.random((4000, 3), dtype=np.float32) * [2, 1, 2] - [1, 2, 1])]).astype(np.float32)
    d = np.concatenate([[[0, 1, 0]], (rng.random((4000, 3), dtype=np.float32) * [1, 1, 1] - [0.5, 0, 0.5])]).astype(np.float32)
    ref = osc.trace(o, d)
    got = ctx.trace(o, d)
    ok = ref["nan_flag"] == 0
    bad = util.hits_equal(got, ref) & ok
    assert not bad.any(), (order, bad.sum(), np.flatnonzero(bad)[:4])
    lin = ctx.trace(o, d, mode=api.TRACE_LINEAR)
    assert not (util.hits_equal(lin, ref) & ok).any()


def test_cone_field_exercises_the_exact_path(ctx):
    """Many cones, many steep upward rays: the order-dependent cases occur in numbers and stay bit-exact."""
    rng = np.random.Generator(np.random.PCG64(4))
    b = scenes.SceneBuilder()
    mats = [b.lambertian((0.5, 0.5, 0.5)), b.metal((0.8, 0.8, 0.8), 0.1)]
    for k in range(300):
        c = rng.random(3) * [20, 6, 20]
        if k % 3 == 0:
            b.sphere(c, 0.4 + rng.random() * 0.5, mats[k % 2])
        else:
            b.cone(c, 0.6 + rng.random() * 1.5, 0.3 + rng.random() * 0.7, mats[k % 2])
    b.camera((10, 3, -15), (10, 3, 10), (0, 1, 0), 40.0, 1.0, 0.0, 5.0)
    ctx.upload(b.desc())
    osc = util.oracle_scene(b)
    n = 60000
    o = (rng.random((n, 3), dtype=np.float32) * [20, 1, 20] - [0, 2, 0]).astype(np.float32)
    d = (rng.random((n, 3), dtype=np.float32) * [0.6, 1, 0.6] - [0.3, 0, 0.3] + [0, 0.2, 0]).astype(np.float32)
    ref = osc.trace(o, d)
    got = ctx.trace(o, d)
    ok = ref["nan_flag"] == 0
    assert not (util.hits_equal(got, ref) & ok).any()
    assert (got["flags"] & 2).sum() > 10  # list-order dependent rays did occur and were replayed exactly


# ------------------------------------------------------------------------------------------------
# material::scatter with the random numbers handed in (pt_scatter_batch), bit-exact
# ------------------------------------------------------------------------------------------------
def _scatter_scene():
    """The scene of tests/golden/scatter_kat.json with its unit sphere five times over, one per material, so that the
    object index selects the material (pt_scatter_batch scatters off the hit object's own material)."""
    b = scenes.SceneBuilder()
    mats = [b.lambertian((0.4, 0.2, 0.1)), b.metal((0.7, 0.6, 0.5), 0.3), b.metal((0.9, 0.9, 0.9), 7.0), b.dielectric(1.5),
            b.diffuse_light((4, 4, 4))]
    for m in mats:
        b.sphere((0, 0, 0), 1.0, m)
    b.camera((0, 0, 5), (0, 0, 0), (0, 1, 0), 40.0, 1.0, 0.0, 5.0)
    return b


def _check_scatter(got, ok, atten, so, sd, emitted):
    bits = lambda a: np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    assert np.array_equal(got["scattered"] != 0, ok)
    for name, want in (("attenuation", atten), ("origin", so), ("direction", sd), ("emitted", emitted)):
        assert np.array_equal(bits(got[name]), bits(want)), name


def test_scatter_known_answers_bit_exact(ctx):
    """Fixture (c): lambertian, metal (fuzz 0.3 and clamped 7 -> 1), dielectric and light, both faces, absorbed and
    scattered metal, reflection and refraction -- the GPU's material code against the committed answers."""
    import json
    rows = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scatter_kat.json")))
    ctx.upload(_scatter_scene().desc())
    f = lambda k: np.array([r[k] for r in rows], dtype=np.float32)
    got = ctx.scatter(f("o"), f("d"), f("t"), np.array([r["mat"] for r in rows], dtype=np.uint32), f("sphere_pt"), f("xi"))
    emitted = np.array([[4, 4, 4] if r["mat"] == 4 else [0, 0, 0] for r in rows], dtype=np.float32)
    _check_scatter(got, np.array([r["ok"] for r in rows]), f("atten"), f("so"), f("sd"), emitted)
    assert len({(r["mat"], r["ok"]) for r in rows}) >= 6


def test_scatter_random_hits_against_the_oracle(ctx):
    """4000 random hits on all six kinds of object (1k mixed scene), each scattered with random injected numbers."""
    b = scenes.mixed_scene(n=1000, seed=3, extent=20.0, n_materials=64)
    osc = util.oracle_scene(b)
    ctx.upload(b.desc())
    o, d = util.mixed_rays(b, osc, 6000, seed=41)
    h = osc.trace(o, d)
    keep = np.flatnonzero((h["prim"] != api.NO_PRIM) & (h["nan_flag"] == 0))[:4000]
    rng = np.random.Generator(np.random.PCG64(5))
    sp = (rng.random((len(keep), 3), dtype=np.float32) * 2 - 1) * np.float32(0.57)
    xi = rng.random(len(keep), dtype=np.float32)
    a = b.arrays()
    want = [osc.scatter(int(a["prim_material"][h["prim"][i]]), o[i], d[i], h[i], sphere_pt=sp[k], xi=float(xi[k])) for k, i in enumerate(keep)]
    got = ctx.scatter(o[keep], d[keep], h["t"][keep], h["prim"][keep], sp, xi)
    mt = a["mat_type"][a["prim_material"][h["prim"][keep]]]
    emitted = np.where((mt == api.DIFFUSE_LIGHT)[:, None], a["mat_param"][a["prim_material"][h["prim"][keep]], :3], 0).astype(np.float32)
    _check_scatter(got, np.array([w[0] for w in want]), np.array([w[1] for w in want]), np.array([w[2] for w in want]), np.array([w[3] for w in want]), emitted)
    assert len(set(mt.tolist())) == 4 and len(set(a["prim_type"][h["prim"][keep]].tolist())) == 6
