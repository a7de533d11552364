"""pt_scene_update: a changed scene with the same number of objects keeps the uploaded tree and refits its
boxes.  The reference has no counterpart (it rebuilds its scene per run and has no hierarchy); the contract
is that answers are those of a fresh upload of the changed scene -- the list scan's, bit for bit."""
import numpy as np
import pytest

from pathtracer_amd import api, scenes
from tests import util
from tests.test_host import _check_tree


def _moved(builder, seed, types_too=False, with_cones=False):
    """The builder's scene with most objects moved, resized and re-materialed (object 0, the ground, stays)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    a = {k: np.array(v, copy=True) for k, v in builder.arrays().items()}
    n = len(a["prim_type"])
    pp, pt = a["prim_param"], a["prim_type"]
    for i in range(1, n):
        if pt[i] == api.SPHERE:
            pp[i, :3] += (rng.random(3) - 0.5).astype(np.float32) * np.float32(3.0)
            pp[i, 3] *= np.float32(0.5 + rng.random())
            if types_too and i % 7 == 0:    # a sphere becomes a cylinder2(center, height, radius) ...
                pt[i] = api.CYLINDER
                pp[i, 3:6] = (0.6, 0.3, 0.0)
            elif types_too and i % 11 == 0:  # ... or a small xz_rect
                pt[i] = api.XZ_RECT
                c = pp[i, :3].copy()
                pp[i] = (c[0] - 0.4, c[0] + 0.4, c[2] - 0.4, c[2] + 0.4, c[1], 0.0)
            elif with_cones and i % 13 == 0:
                pt[i] = api.CONE
                h, r = np.float32(0.8), np.float32(0.4)
                pp[i, 3:6] = (h, r, (r / h) * (r / h))
    a["prim_material"][1:] = rng.integers(0, len(a["mat_type"]), n - 1).astype(a["prim_material"].dtype)
    return scenes.ArrayScene(a, builder.background, builder.camera_args)


@pytest.mark.parametrize("kw", [dict(), dict(types_too=True), dict(types_too=True, with_cones=True)])
def test_refitted_tree_is_a_valid_tree_of_the_new_scene(kw):
    before = scenes.random_spheres_scene()
    after = _moved(before, 3, **kw)
    nodes, order, depth = api.bvh_refit_host(before.desc(), after.desc())
    nodes0, order0, depth0 = api.bvh_build_host(before.desc())
    assert depth == depth0 and len(nodes) == len(nodes0) and np.array_equal(order, order0)  # same topology
    assert np.array_equal(nodes["c"], nodes0["c"])
    _check_tree(after.desc(), nodes, order)  # every (padded) box encloses its new objects and its children
    with pytest.raises(api.PtError):
        api.bvh_refit_host(before.desc(), scenes.default_scene().desc())  # another object count


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(), dict(types_too=True), dict(types_too=True, with_cones=True)])
def test_update_gives_the_fresh_uploads_answers(ctx, kw):
    before = scenes.random_spheres_scene()
    after = _moved(before, 5, **kw)
    osc = util.oracle_scene(after)
    o, d = util.mixed_rays(after, osc, 20000, seed=77)
    ref = osc.trace(o, d)
    ok = ref["nan_flag"] == 0
    ctx.upload(before.desc())
    ctx.update(after.desc())
    for mode in (api.TRACE_BVH, api.TRACE_BVH_IN_SCENE, api.TRACE_LINEAR):
        got = ctx.trace(o, d, mode=mode)
        assert not (util.hits_equal(got, ref) & ok).any(), (kw, mode)
    img_refit, st_refit = ctx.render(160, 90, 4, 50)
    ctx.upload(after.desc())
    img_fresh, st_fresh = ctx.render(160, 90, 4, 50)
    assert st_refit["rays"] == st_fresh["rays"]
    assert np.array_equal(img_refit.view(np.uint32), img_fresh.view(np.uint32))  # same paths through either tree
    # and back again: updates can be chained
    ctx.update(before.desc())
    img_back, _ = ctx.render(160, 90, 4, 50)
    ctx.upload(before.desc())
    img_orig, _ = ctx.render(160, 90, 4, 50)
    assert np.array_equal(img_back.view(np.uint32), img_orig.view(np.uint32))


@pytest.mark.gpu
def test_update_of_a_large_scene_and_errors(ctx):
    before = scenes.mixed_scene(n=20000, seed=9, extent=60.0, n_materials=64)
    a = {k: np.array(v, copy=True) for k, v in before.arrays().items()}
    rng = np.random.Generator(np.random.PCG64(1))
    sph = a["prim_type"] == api.SPHERE
    a["prim_param"][sph, :3] += (rng.random((int(sph.sum()), 3)) - 0.5).astype(np.float32)
    after = scenes.ArrayScene(a, before.background, before.camera_args)
    ctx.upload(before.desc())
    ctx.update(after.desc())
    osc = util.oracle_scene(after)
    o, d = util.mixed_rays(after, osc, 3000, seed=12)
    ref = osc.trace(o, d)
    got = ctx.trace(o, d)
    assert not (util.hits_equal(got, ref) & (ref["nan_flag"] == 0)).any()
    with pytest.raises(api.PtError):
        ctx.update(scenes.default_scene().desc())  # another object count: refused, the installed scene stays
    again = ctx.trace(o, d)
    assert np.array_equal(again["t"].view(np.uint32), got["t"].view(np.uint32))
    fresh = api.Context(0)
    with pytest.raises(api.PtError) as e:
        fresh.update(after.desc())
    assert e.value.status == 2  # PT_ERR_NO_SCENE
    fresh.close()


@pytest.mark.gpu
def test_update_of_a_one_leaf_scene(ctx):
    """The Cornell box (seven objects: one leaf, rendered by the kernels without tree code) with its cylinder and
    its light moved: the updated scene renders what a fresh upload of it renders."""
    before = scenes.cornell_scene()
    a = {k: np.array(v, copy=True) for k, v in before.arrays().items()}
    cyl = a["prim_type"] == api.CYLINDER
    assert cyl.sum() == 1 and len(a["prim_type"]) <= 8
    a["prim_param"][cyl, 0] -= np.float32(120.0)
    a["prim_param"][cyl, 2] += np.float32(90.0)
    a["prim_param"][-1, 4] -= np.float32(30.0)  # the light, an xz_rect, comes down
    after = scenes.ArrayScene(a, before.background, before.camera_args)
    ctx.upload(before.desc())
    img_before, _ = ctx.render(64, 64, 4, 8)
    ctx.update(after.desc())
    img_refit, st_refit = ctx.render(64, 64, 4, 8)
    ctx.upload(after.desc())
    img_fresh, st_fresh = ctx.render(64, 64, 4, 8)
    assert st_refit["rays"] == st_fresh["rays"]
    assert np.array_equal(img_refit.view(np.uint32), img_fresh.view(np.uint32))
    assert not np.array_equal(img_refit.view(np.uint32), img_before.view(np.uint32))
    ref, rays = util.oracle_scene(after).render(64, 64, 4, 8)
    assert st_fresh["rays"] == rays
