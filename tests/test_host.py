"""CPU tests of the host side: the shared library loads and exports the declared C ABI, and the
host logic behind it (camera, BVH builder, flattening checks, film finish, sharding) is correct.
No compute call is made: there is no GPU here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from oracle import pt_oracle as O
from pathtracer_amd import api, distributed, scenes
from tests import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "pt_api.h")).read()
    declared = set(re.findall(r"\b(pt_[a-z_]+)\s*\(", hdr))
    assert declared == set(api.EXPORTS), declared ^ set(api.EXPORTS)
    L = api.lib()
    for name in declared:
        assert getattr(L, name) is not None
    assert "gfx950" in api.version()


def test_struct_layouts_match_the_header():
    assert C.sizeof(api.PtCamera) == 22 * 4
    assert api.HIT_DTYPE.itemsize == 40 and api.NODE_DTYPE.itemsize == 64
    assert C.sizeof(api.PtRenderParams) == 40 and C.sizeof(api.PtStats) == 112
    assert C.sizeof(api.PtTuning) == 15 * 4 == 4 * len(api.TUNING_FIELDS)


def test_context_create_fails_cleanly_without_a_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(api.PtError):  # status, not exit(99) as the reference's checkCudaErrors
        api.Context(0)


@pytest.mark.parametrize("builder", [scenes.default_scene, scenes.cornell_scene, scenes.random_spheres_scene])
def test_camera_bit_exact_against_oracle(builder):
    b = builder()
    lib_cam = api.camera_fields(api.camera_look_at(*b.camera_args))
    orc_cam = util.oracle_scene(b).camera_fields()
    for k, ok in (("origin", "origin"), ("lower_left_corner", "lower_left"), ("horizontal", "horizontal"), ("vertical", "vertical"),
                  ("u", "u"), ("v", "v"), ("w", "w")):
        assert np.array_equal(lib_cam[k].view(np.uint32), orc_cam[ok].view(np.uint32)), k
    assert lib_cam["lens_radius"] == orc_cam["lens_radius"]


def _check_tree(desc, nodes, order, leaf_max=None):
    # the library's choice: one leaf for a scene of up to eight objects, single objects for scenes traversed from HBM
    leaf_max = leaf_max or (8 if desc.c.n_prims <= 8 else (1 if desc.c.n_prims > 1000 else 4))
    n = desc.c.n_prims
    assert sorted(order.tolist()) == list(range(n))  # every object in exactly one leaf slot
    seen = np.zeros(n, bool)
    pp, pt = desc.prim_param, desc.prim_type

    def prim_bounds(i):
        q, t = pp[i], pt[i]
        if t == api.SPHERE:
            r = abs(q[3]); return q[:3] - r, q[:3] + r
        if t in (api.CONE, api.CYLINDER):
            r = abs(q[4]); return np.array([q[0] - r, q[1], q[2] - r]), np.array([q[0] + r, q[1] + q[3], q[2] + r])
        ax = {api.XY_RECT: (0, 1, 2), api.XZ_RECT: (0, 2, 1), api.YZ_RECT: (1, 2, 0)}[t]
        lo, hi = np.zeros(3), np.zeros(3)
        lo[ax[0]], hi[ax[0]], lo[ax[1]], hi[ax[1]] = q[0], q[1], q[2], q[3]
        lo[ax[2]], hi[ax[2]] = q[4] - 1e-4, q[4] + 1e-4
        return lo, hi

    def walk(ref, lo, hi, depth):
        if ref < 0:
            r = ~ref
            first, count = r >> 3, (r & 7) + 1
            assert count <= leaf_max
            for s in range(first, first + count):
                i = order[s]
                assert not seen[i]
                seen[i] = True
                plo, phi = prim_bounds(i)
                assert (plo >= lo - 1e-6).all() and (phi <= hi + 1e-6).all()  # the (padded) node box encloses the object
            return depth
        nd = nodes[ref]
        kids = [(nd["b"][:, k], nd["b"][:, 2 + k]) for k in range(2)]
        for clo, chi in kids:
            if np.isfinite(clo).all():
                assert (clo >= lo - 1e-6).all() and (chi <= hi + 1e-6).all()  # children inside the parent
        d0 = walk(int(nd["c"][0]), kids[0][0], kids[0][1], depth + 1)
        d1 = walk(int(nd["c"][1]), kids[1][0], kids[1][1], depth + 1) if np.isfinite(kids[1][0]).all() else depth
        return max(d0, d1)

    inf = np.full(3, np.inf)
    depth = walk(0, -inf, inf, 0)
    assert seen.all()
    return depth


@pytest.mark.parametrize("builder", [scenes.default_scene, scenes.cornell_scene, scenes.random_spheres_scene,
                                     lambda: scenes.mixed_scene(n=3000, seed=5, extent=30.0, n_materials=32)])
def test_bvh_invariants(builder):
    d = builder().desc()
    nodes, order, max_depth = api.bvh_build_host(d)
    depth = _check_tree(d, nodes, order)
    assert depth == max_depth and max_depth <= 64
    if d.c.n_prims > 4:
        assert len(nodes) >= d.c.n_prims // (8 if d.c.n_prims > 1000 else 4) - 1


def test_bvh_degenerate_inputs():
    b = scenes.SceneBuilder()
    m = b.lambertian((1, 1, 1))
    for _ in range(37):
        b.sphere((1, 2, 3), 0.5, m)  # coincident centroids: SAH has nothing to cut, the median split must terminate
    b.camera((0, 0, 5), (0, 0, 0), (0, 1, 0), 40.0, 1.0, 0.0, 5.0)
    d = b.desc()
    nodes, order, max_depth = api.bvh_build_host(d)
    _check_tree(d, nodes, order)
    one = scenes.SceneBuilder()
    one.sphere((0, 0, 0), 1, one.lambertian((1, 1, 1)))
    one.camera((0, 0, 5), (0, 0, 0), (0, 1, 0), 40.0, 1.0, 0.0, 5.0)
    nodes, order, max_depth = api.bvh_build_host(one.desc())
    assert len(nodes) == 1 and max_depth == 1 and list(order) == [0] and not np.isfinite(nodes[0]["b"][:, 1]).any()


def test_bad_scenes_are_rejected_not_crashed():
    b = scenes.default_scene()
    a = b.arrays()
    cam = api.camera_look_at(*b.camera_args)
    bad_mat = api.SceneDesc(a["prim_type"], a["prim_param"], np.array([0, 1, 9], np.uint32), a["mat_type"], a["mat_param"], cam)
    with pytest.raises(api.PtError):  # the reference indexes materials[material_index] unchecked (sphere.h:71)
        api.bvh_build_host(bad_mat)
    bad_tag = api.SceneDesc(np.array([0, 0, 7], np.uint8), a["prim_param"], a["prim_material"], a["mat_type"], a["mat_param"], cam)
    with pytest.raises(api.PtError):
        api.bvh_build_host(bad_tag)
    nan = a["prim_param"].copy()
    nan[1, 0] = np.nan
    with pytest.raises(api.PtError):
        api.bvh_build_host(api.SceneDesc(a["prim_type"], nan, a["prim_material"], a["mat_type"], a["mat_param"], cam))
    with pytest.raises(ValueError):
        scenes.SceneBuilder().sphere((0, 0, 0), 1, 0)  # material must exist before the object that uses it
    with pytest.raises(ValueError):
        scenes.SceneBuilder().desc()  # no camera


def test_metal_fuzz_clamp_and_cylinder_argument_order():
    b = scenes.SceneBuilder()
    m = b.metal((1, 1, 1), 3.0)
    assert b.arrays()["mat_param"][m, 3] == 1.0  # material.h:150
    c = scenes.cornell_scene().arrays()
    cyl = c["prim_param"][c["prim_type"] == api.CYLINDER][0]
    assert cyl[3] == 200 and cyl[4] == 64  # scene.h:85 passes (center, 200, 64): height 200, radius 64


def test_film_finalize_bit_exact_against_oracle():
    rng = np.random.Generator(np.random.PCG64(1))
    img = (rng.random((9, 13, 3), dtype=np.float32) * 40).astype(np.float32)
    img[0, 0] = [0, 1e-9, 1e9]
    g1, q1 = api.film_finalize(img, 17)
    g2, q2 = O.finalize(img, 17)
    assert np.array_equal(g1.view(np.uint32), g2.view(np.uint32)) and np.array_equal(q1, q2)
    assert q1.max() == 255 and q1.min() == 0
    # rows are written top-down from j = H-1 (main.cu:194): the first 8-bit row is the last float row
    assert np.array_equal(q1[0], (256 * np.clip(g1[-1], 0, 0.999)).astype(np.int32).astype(np.uint8))


def test_shards():
    assert [distributed.weak_shard(r, 4, 64) for r in range(4)] == [(0, 64), (64, 64), (128, 64), (192, 64)]
    cuts = [distributed.strong_shard(r, 8, 67) for r in range(8)]
    assert sum(n for _, n in cuts) == 67 and cuts[0][0] == 0
    assert all(cuts[k][0] + cuts[k][1] == cuts[k + 1][0] for k in range(7))
    assert max(n for _, n in cuts) - min(n for _, n in cuts) <= 1
    with pytest.raises(ValueError):
        distributed.weak_shard(4, 4, 8)
    with pytest.raises(ValueError):  # fewer samples than ranks: a rank would get none, which pt_render_device rejects
        distributed.strong_shard(0, 8, 7)
    assert distributed.shard("strong", 3, 8, 64) == (24, 8) and distributed.shard("weak", 3, 8, 64) == (192, 64)
    with pytest.raises(ValueError):
        distributed.shard("sideways", 0, 1, 8)


def test_random_spheres_scene_is_the_named_workload():
    a = scenes.random_spheres_scene().arrays()
    assert len(a["prim_type"]) == 485 and (a["prim_type"] == api.SPHERE).all()
    kinds = a["mat_type"][a["prim_material"]]
    assert (kinds == api.LAMBERTIAN).sum() > 350 and (kinds == api.METAL).sum() > 40 and (kinds == api.DIELECTRIC).sum() > 10


def test_ray_cone_parameters():
    """How wide a cone the traversal gives its rays (pt_scene_ray_cone): per-ray for sphere / cylinder scenes,
    rounding-level for rect-only scenes, fixed and slope-scaled for scenes with cones."""
    k = 2.0 ** -10
    c2 = api.scene_ray_cone(scenes.random_spheres_scene().desc())
    assert c2["adaptive"] and c2["kappa_max"] == k and c2["kappa_min"] == 2.0 ** -20
    assert abs(c2["r_min"] - 0.2) < 1e-6                      # the ground sphere (r = 1000) carries its own padding
    assert c2["cluster_half_extent_sum"] < 40 and max(abs(v) for v in c2["cluster_centre"]) < 5
    # a camera ray of that scene: origin (13, 2, 3), reach ~ 50 -> slope far below kappa_max
    d1 = sum(abs(o - c) for o, c in zip((13.0, 2.0, 3.0), c2["cluster_centre"])) + c2["cluster_half_extent_sum"]
    eta = k * d1
    assert k * eta / (max(c2["r_min"], eta) + c2["r_min"]) < 0.2 * k

    box = scenes.SceneBuilder()
    mw = box.lambertian((0.7, 0.7, 0.7))
    box.xz_rect(0, 5, 0, 5, 0, mw)
    box.xy_rect(0, 5, 0, 5, 5, mw)
    box.yz_rect(0, 5, 0, 5, 0, mw)
    box.camera((2, 2, -8), (2, 2, 0), (0, 1, 0), 40.0, 1.0, 0.0, 8.0)
    rects = api.scene_ray_cone(box.desc())
    assert not rects["adaptive"] and rects["kappa_max"] == 2.0 ** -20
    # the Cornell box's one cylinder is large against the room and carries its own padding: rounding-level cone
    cornell = api.scene_ray_cone(scenes.cornell_scene().desc())
    assert cornell["adaptive"] and cornell["r_min"] > 1e29

    b = scenes.SceneBuilder()
    m = b.lambertian((0.5, 0.5, 0.5))
    b.sphere((0, 0, 0), 1.0, m)
    b.cone((3, 0, 0), 0.5, 1.0, m)                            # radius / height = 2
    b.camera((0, 0, 9), (0, 0, 0), (0, 1, 0), 40.0, 1.0, 0.0, 9.0)
    cones = api.scene_ray_cone(b.desc())
    assert not cones["adaptive"] and abs(cones["kappa_max"] - 2 * k) < 1e-9

    tiny = scenes.SceneBuilder()
    tiny.sphere((0, 0, 0), 0.0, tiny.lambertian((1, 1, 1)))   # a zero radius leaves no room to narrow the cone
    tiny.camera((0, 0, 5), (0, 0, 0), (0, 1, 0), 40.0, 1.0, 0.0, 5.0)
    z = api.scene_ray_cone(tiny.desc())
    assert z["adaptive"] and z["r_min"] == 0.0


def test_leaf_boxes_carry_the_objects_share_of_the_cone():
    """An object's box is padded by kappa_max x its own extent (a cone that grows with the distance travelled
    cannot see the part of the origin-to-centre distance that lies inside the object), or by its whole
    worst-case excess when it is large enough to carry that itself."""
    b = scenes.random_spheres_scene()
    d = b.desc()
    nodes, order, _ = api.bvh_build_host(d)
    pp = b.arrays()["prim_param"]
    k = 2.0 ** -10
    found = 0
    for nd in nodes:
        for child in range(2):
            ref = int(nd["c"][child])
            if ref >= 0 or not np.isfinite(nd["b"][:, child]).all():
                continue
            r = ~ref
            first, count = r >> 3, (r & 7) + 1
            for s in range(first, first + count):
                c, rad = pp[order[s]][:3], abs(pp[order[s]][3])
                pad_lo = (c - rad) - nd["b"][:, child]
                pad_hi = nd["b"][:, 2 + child] - (c + rad)
                need = k * rad if rad < 100 else 0.0
                assert (pad_lo >= need).all() and (pad_hi >= need).all()
                found += 1
    assert found == len(pp)


@pytest.mark.parametrize("make", [scenes.default_scene, scenes.cornell_scene, scenes.random_spheres_scene,
                                  lambda: scenes.mixed_scene(n=3000, extent=30.0), lambda: scenes.sphere_slab_scene(n=5000)])
def test_wide_tree_holds_the_binary_tree(make):
    """The four-wide quantised records: the same leaves as the binary tree, each once; every leaf's box (as the
    binary tree holds it) lies inside the decoded box of every record entry above it; the decoding grid is exact
    in fp32 (origin a multiple of the step, |origin| / step <= 2^23)."""
    b = make()
    d = b.desc()
    nodes, order, _ = api.bvh_build_host(d)
    wide, depth = api.bvh_wide_host(d)
    leaf_box = {}
    for nd in nodes:
        for k in range(2):
            if nd["c"][k] < 0 and np.isfinite(nd["b"][:, k]).all():
                leaf_box[int(nd["c"][k])] = (nd["b"][:, k].astype(np.float64), nd["b"][:, 2 + k].astype(np.float64))
    seen = set()

    def walk(i, above, level):
        w = wide[i]
        step = np.ldexp(1.0, w["exp"].astype(int) - 127)
        k0 = w["origin"].astype(np.float64) / step
        assert (k0 == np.round(k0)).all() and (np.abs(k0) <= 2 ** 23).all()
        n = int(w["n_children"])
        assert 1 <= n <= 4 and (w["c"][n:] == api.WIDE_ABSENT).all() and (w["c"][:n] != api.WIDE_ABSENT).all()
        deepest = level
        for k in range(n):
            lo = w["origin"] + w["qlo"][:, k].astype(np.float32) * step.astype(np.float32)  # as the device decodes it
            hi = w["origin"] + w["qhi"][:, k].astype(np.float32) * step.astype(np.float32)
            assert (lo.astype(np.float64) == k0 * step + w["qlo"][:, k] * step).all()      # ... exactly
            ref = int(w["c"][k])
            chain = above + [(lo.astype(np.float64), hi.astype(np.float64))]
            if ref >= 0:
                deepest = max(deepest, walk(ref, chain, level + 1))
            else:
                assert ref not in seen
                seen.add(ref)
                blo, bhi = leaf_box[ref]
                for alo, ahi in chain:
                    assert (alo <= blo).all() and (bhi <= ahi).all()
        return deepest

    assert walk(0, [], 1) == depth
    assert seen == set(leaf_box)
    assert len(wide) <= len(nodes)
