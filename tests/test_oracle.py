"""CPU tests of the oracle: known answers derived by hand from the reference's formulas, the one
vector recorded from the reference itself, and the committed golden fixtures (tests/golden/)."""
import json
import os

import numpy as np
import pytest

from oracle import pt_oracle as O
from pathtracer_amd import scenes
from tests import util

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
f32 = np.float32
FMAX = float(np.finfo(np.float32).max)


def one(b):
    return util.oracle_scene(b)


def simple(add):
    b = scenes.SceneBuilder()
    m = b.lambertian((0.5, 0.5, 0.5))
    add(b, m)
    b.camera((0, 0, 5), (0, 0, 0), (0, 1, 0), 40.0, 1.0, 0.0, 5.0)
    return one(b)


# ---------------------------------------------------------------- random stream
def test_philox_known_answers():
    """Random123 kat_vectors for philox4x32_10 (the published default) and philox4x32_7 (the stream's round count)."""
    assert O.philox([0, 0, 0, 0], [0, 0]) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert O.philox([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert O.philox([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0]) == [
        0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]
    assert O.PHILOX_ROUNDS == 7
    assert O.philox([0, 0, 0, 0], [0, 0], 7) == [0x5F6FB709, 0x0D893F64, 0x4F121F81, 0x4F730A48]
    assert O.philox([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2, 7) == [0x5207DDC2, 0x45165E59, 0x4D8EE751, 0x8C52F662]
    assert O.philox([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0], 7) == [
        0x4DFCCABA, 0x190A87F0, 0xC47362BA, 0xB6B5242A]


def test_uniform_range_is_half_open_like_curand():
    assert O.u32_to_uniform(0) == 2.0 ** -24  # never 0
    assert O.u32_to_uniform(0xFFFFFFFF) == 1.0  # 1 is attainable (main.cu:90 uses curand_uniform: (0,1])
    assert O.u32_to_uniform(0x80000000) == 0.5 + 2.0 ** -24


# ---------------------------------------------------------------- the reference's own vector
def test_reference_known_answer_bit_exact():
    ka = json.load(open(os.path.join(GOLD, "reference_known_answer.json")))
    osc = one(scenes.default_scene())
    h = osc.trace([ka["origin"]], [ka["direction"]])[0]
    assert h["prim"] == ka["prim"] and h["front"] == ka["front"]
    assert h["t"] == f32(ka["t"])  # 9 significant digits identify one fp32 value
    assert all(h["p"][k] == f32(ka["p"][k]) for k in range(3))
    assert all(h["n"][k] == f32(ka["n"][k]) for k in range(3))


# ---------------------------------------------------------------- per-object known answers
def test_sphere_front_back_and_negative_radius():
    osc = simple(lambda b, m: b.sphere((0, 0, 0), 1.0, m))
    hit, r = osc.hit_prim(0, (0, 0, -3), (0, 0, 1))
    assert hit and r["t"] == 2.0 and list(r["p"]) == [0, 0, -1] and list(r["n"]) == [0, 0, -1] and r["front"] == 1
    hit, r = osc.hit_prim(0, (0, 0, 0), (0, 0, 2))  # from inside; direction not normalised: t in units of |d| (Q1)
    assert hit and r["t"] == 0.5 and list(r["n"]) == [0, 0, -1] and r["front"] == 0
    assert not osc.hit_prim(0, (0, 2, -3), (0, 0, 1))[0]
    hollow = simple(lambda b, m: b.sphere((0, 0, 0), -1.0, m))  # sphere.h:106 divides by the signed radius
    hit, r = hollow.hit_prim(0, (0, 0, -3), (0, 0, 1))
    assert hit and list(r["n"]) == [0, 0, -1] and r["front"] == 0  # outward normal flipped: reported as a back face


def test_sphere_range_tests_are_inclusive():
    osc = simple(lambda b, m: b.sphere((0, 0, 0), 1.0, m))
    assert osc.hit_prim(0, (0, 0, -3), (0, 0, 1), t_min=2.0, t_max=2.0)[0]  # root == t_min == t_max accepted (sphere.h:97)
    hit, r = osc.hit_prim(0, (0, 0, -3), (0, 0, 1), t_min=2.5, t_max=10)
    assert hit and r["t"] == 4.0  # near root out of range -> far root
    assert not osc.hit_prim(0, (0, 0, -3), (0, 0, 1), t_min=2.5, t_max=3.5)[0]


def test_rects_and_nan_semantics():
    osc = simple(lambda b, m: (b.xy_rect(0, 2, 0, 2, 1, m), b.xz_rect(0, 2, 0, 2, 1, m), b.yz_rect(0, 2, 0, 2, 1, m)))
    hit, r = osc.hit_prim(0, (1, 1, -1), (0, 0, 1))
    assert hit and r["t"] == 2.0 and list(r["n"]) == [0, 0, -1] and r["front"] == 0  # outward is +z: hit from behind
    hit, r = osc.hit_prim(1, (1, 3, 1), (0, -1, 0))
    assert hit and r["t"] == 2.0 and list(r["n"]) == [0, 1, 0] and r["front"] == 1
    hit, r = osc.hit_prim(2, (4, 1, 1), (-3, 0, 0))
    assert hit and r["t"] == 1.0 and list(r["n"]) == [1, 0, 0] and r["front"] == 1
    assert osc.hit_prim(0, (2, 2, 0), (0, 0, 1))[0]  # bounds are inclusive (aarect.h:354)
    assert not osc.hit_prim(0, (2.0001, 1, 0), (0, 0, 1))[0]
    assert not osc.hit_prim(0, (1, 1, 0), (1, 0, 0))[0]  # parallel: t = +inf is rejected by t > t_max
    hit, r = osc.hit_prim(0, (1, 1, 1), (1, 0, 0))  # in the plane: 0/0 = NaN passes every comparison (Q9)
    assert hit and np.isnan(r["t"])


def test_cylinder_quirks():
    osc = simple(lambda b, m: b.cylinder2((0, 1, 0), 2.0, 0.5, m))  # (center, HEIGHT, RADIUS): Q19
    hit, r = osc.hit_prim(0, (-2, 2, 0), (1, 0, 0))
    assert hit and r["t"] == 1.5
    assert list(r["n"]) == [-1.0, 2.0, 0.0]  # y of the normal is the ABSOLUTE hit height (cylinder.h:126), not unit length
    assert not osc.hit_prim(0, (-2, 3.5, 0), (1, 0, 0))[0]  # above the top
    # near wall hit outside the y range -> false without trying the far wall; no caps (cylinder.h:121-123)
    assert not osc.hit_prim(0, (-1, 3.5, 0), (1, -0.5, 0))[0]  # near wall at y=3.25 (outside), far wall at y=2.75 never tried


def test_cone_quirks():
    osc = simple(lambda b, m: b.cone((0, 0, 0), 2.0, 1.0, m))  # apex at y = 2, base radius 1, tangens 0.25
    hit, r = osc.hit_prim(0, (-3, 1, 0), (1, 0, 0))
    assert hit and r["t"] == 2.5 and r["front"] == 1
    assert list(r["n"]) == [-0.5, 0.25, 0.0]  # ((p-c)/radius).x, then y := |pc_xz| * radius / height (cone.h:131-135)
    # steep ray from above: a < 0, the LARGER root (-half_b - sqrtd)/a is tried first (cone.h:117) and is the real nappe
    hit, r = osc.hit_prim(0, (0.25, 4, 0), (0, -1, 0))
    assert hit and r["t"] == 2.5 and r["p"][1] == 1.5
    # steep ray from below through the open base: first-tried root is the mirror nappe above the apex -> clipped -> false,
    # although the real surface is hit at t = 2.5
    assert not osc.hit_prim(0, (0.25, -1, 0), (0, 1, 0))[0]
    # ... unless t_max already excludes the mirror root: then the nearer, real root is taken
    hit, r = osc.hit_prim(0, (0.25, -1, 0), (0, 1, 0), t_max=3.0)
    assert hit and r["t"] == 2.5


def test_list_scan_order_dependence_of_steep_cone():
    """hittable_list::hit shrinks t_max as it goes (hittable_list.h:93-95), so the cone above is visible from
    below only when an EARLIER list entry was already hit nearer than the mirror root."""
    def scene(cone_first):
        b = scenes.SceneBuilder()
        m = b.lambertian((0.5, 0.5, 0.5))
        if cone_first:
            b.cone((0, 0, 0), 2.0, 1.0, m)
            b.xz_rect(-5, 5, -5, 5, 2.25, m)
        else:
            b.xz_rect(-5, 5, -5, 5, 2.25, m)
            b.cone((0, 0, 0), 2.0, 1.0, m)
        b.camera((0, 0, 5), (0, 0, 0), (0, 1, 0), 40.0, 1.0, 0.0, 5.0)
        return one(b)
    o, d = [[0.25, -1, 0]], [[0, 1, 0]]  # real root 2.5, mirror root 3.5, ceiling at t = 3.25
    h = scene(True).trace(o, d)[0]
    assert h["prim"] == 1 and h["t"] == 3.25  # cone tested with t_max = FLT_MAX: mirror root, clipped; ceiling wins
    h = scene(False).trace(o, d)[0]
    assert h["prim"] == 1 and h["t"] == 2.5  # ceiling first: t_max = 3.25 < 3.5, so the cone's real root is accepted


def test_cone_whose_leading_coefficient_cancels_accepts_every_such_ray():
    """cone.h:108: a = |d_xz|^2 - tangens d_y d_y.  With tangens 1/4 and d = (3, 10, 4) both terms are exactly 25, so
    a = 0 whatever the origin: the discriminant is half_b^2, (-half_b - sqrtd) / 0 is -inf or 0 / 0 and
    (-half_b + sqrtd) / 0 is 0 / 0 or +inf (cone.h:116-120).  -inf fails `root < t_min`; a NaN fails no test, not the
    range tests and not the clip of cone.h:121-123: the cone, 900 units away from the ray, is hit at t = NaN.
    hittable_list::hit (hittable_list.h:87-101) then carries closest_so_far = NaN: a later entry is accepted wherever the
    ray's line crosses it, an earlier hit is overwritten."""
    def scene(order):
        b = scenes.SceneBuilder()
        m = b.lambertian((0.5, 0.5, 0.5))
        for what in order:
            if what == "cone":
                b.cone((500.0, -300.0, 800.0), 2.0, 1.0, m)  # tangens (1 / 2)^2
            else:
                b.sphere((3.0, 10.0, 4.0), 1.0, m)  # on the ray's line, at t = 1 -+ 1 / sqrt(125)
        b.camera((0, 0, 5), (0, 0, 0), (0, 1, 0), 40.0, 1.0, 0.0, 5.0)
        return one(b)
    o, d = [[0.0, 0.0, 0.0]], [[3.0, 10.0, 4.0]]
    hit, r = scene(["cone"]).hit_prim(0, o[0], d[0])
    assert hit and np.isnan(r["t"])
    h = scene(["cone"]).trace(o, d)[0]
    assert h["prim"] == 0 and np.isnan(h["t"]) and h["nan_flag"] & 1
    h = scene(["cone", "sphere"]).trace(o, d)[0]  # the later sphere takes over from the NaN
    assert h["prim"] == 1 and abs(h["t"] - (1.0 - 1.0 / np.sqrt(125.0))) < 1e-6
    h = scene(["sphere", "cone"]).trace(o, d)[0]  # the sphere is hit first -- and overwritten by the cone's NaN
    assert h["prim"] == 1 and np.isnan(h["t"])
    h = scene(["sphere", "cone"]).trace(o, [[3.0, 10.0, 4.5]])[0]  # another direction: a != 0, the far cone is missed
    assert h["prim"] != 1 and not np.isnan(h["t"])


def test_ties_go_to_the_later_list_index():
    b = scenes.SceneBuilder()
    m = b.lambertian((0.5, 0.5, 0.5))
    b.xy_rect(-1, 1, -1, 1, 0, m)
    b.xy_rect(-2, 2, -2, 2, 0, m)
    b.sphere((0, 0, 1), 1.0, m)  # also touches z = 0 at the origin
    b.camera((0, 0, 5), (0, 0, 0), (0, 1, 0), 40.0, 1.0, 0.0, 5.0)
    h = one(b).trace([[0, 0, -2]], [[0, 0, 1]])[0]
    assert h["t"] == 2.0 and h["prim"] == 2


def test_camera_matches_hand_computation():
    b = scenes.SceneBuilder()
    b.sphere((0, 0, 0), 1, b.lambertian((1, 1, 1)))
    b.camera((0, 0, 1), (0, 0, 0), (0, 1, 0), 90.0, 2.0, 0.0, 1.0)
    c = one(b).camera_fields()
    h = np.tan(f32(90.0) * f32(3.1415926535897932385) / f32(180.0) / f32(2.0), dtype=np.float32)
    assert np.allclose(c["horizontal"], [4 * h, 0, 0], rtol=1e-6) and np.allclose(c["vertical"], [0, 2 * h, 0], rtol=1e-6)
    assert list(c["w"]) == [0, 0, 1] and list(c["u"]) == [1, 0, 0] and list(c["v"]) == [0, 1, 0]
    assert c["lens_radius"] == 0
    o, d = one(b).get_ray(0.5, 0.5)
    assert list(o) == [0, 0, 1] and np.allclose(d, [0, 0, -1], atol=1e-6)  # not normalised in general (Q1)


def test_materials_hand_cases():
    b = scenes.SceneBuilder()
    lam, met, fuzzy, glass, light = (b.lambertian((0.4, 0.2, 0.1)), b.metal((0.7, 0.6, 0.5), 0.0), b.metal((1, 1, 1), 7.0),
                                     b.dielectric(1.5), b.diffuse_light((4, 4, 4)))
    b.xz_rect(-5, 5, -5, 5, 0, lam)
    b.camera((0, 0, 5), (0, 0, 0), (0, 1, 0), 40.0, 1.0, 0.0, 5.0)
    osc = one(b)
    o, d = (0, 1, 0), (1, -1, 0)
    hit, rec = osc.hit_prim(0, o, d)
    assert hit and rec["t"] == 1.0 and list(rec["n"]) == [0, 1, 0]
    ok, att, so, sd = osc.scatter(lam, o, d, rec, sphere_pt=(0, 0.5, 0))
    assert ok and list(att) == [f32(0.4), f32(0.2), f32(0.1)] and list(so) == [1, 0, 0] and list(sd) == [0, 2, 0]
    ok, att, so, sd = osc.scatter(lam, o, d, rec, sphere_pt=(0, -0.5, 0))  # normal + unit = 0 -> falls back to the normal
    assert ok and list(sd) == [0, 1, 0]
    ok, att, so, sd = osc.scatter(met, o, d, rec, sphere_pt=(0.3, 0.3, 0.3))  # fuzz 0: mirror direction of unit(d)
    s = f32(1) / np.sqrt(f32(2))
    assert ok and np.allclose(sd, [s, s, 0], rtol=1e-6)
    ok, att, so, sd = osc.scatter(fuzzy, o, d, rec, sphere_pt=(0, -0.9, 0))  # fuzz clamped to 1 (material.h:150): absorbed
    assert not ok and np.allclose(sd, [s, s - 0.9, 0], rtol=1e-6)
    ok, att, so, sd = osc.scatter(glass, o, d, rec, xi=1.0)  # xi = 1 > reflectance: refracts
    assert ok and list(att) == [1, 1, 1] and sd[1] < 0
    ok, att, so, sd = osc.scatter(glass, o, d, rec, xi=1e-6)  # tiny xi: reflects
    assert ok and np.allclose(sd, [s, s, 0], rtol=1e-6)
    ok, att, so, sd = osc.scatter(light, o, d, rec)
    assert not ok


def test_render_black_world_and_background_only():
    b = scenes.SceneBuilder(background=(0.25, 0.5, 1.0))
    b.sphere((0, 0, 100), 1.0, b.lambertian((1, 1, 1)))  # behind the camera
    b.camera((0, 0, 1), (0, 0, 0), (0, 1, 0), 40.0, 1.0, 0.0, 1.0)
    img, rays = one(b).render(8, 6, 3, 5)
    assert rays == 8 * 6 * 3 and np.array_equal(img, np.tile(np.array([0.75, 1.5, 3.0], np.float32), (6, 8, 1)))
    gamma, rgb8 = O.finalize(img, 3)
    assert np.allclose(gamma[0, 0], np.sqrt([0.25, 0.5, 1.0]), rtol=1e-6)
    assert list(rgb8[0, 0]) == [128, int(256 * np.sqrt(f32(0.5))), 255]  # clamp to 0.999 then *256 (main.cu:198-200)
    img0, rays0 = one(b).render(8, 6, 3, 0)  # depth 0: no ray is traced, black (main.cu:50-51)
    assert rays0 == 0 and not img0.any()


# ---------------------------------------------------------------- committed fixtures
@pytest.mark.parametrize("name", ["default", "cornell", "random_spheres", "mixed_1k"])
def test_golden_ray_batches(name):
    from tests.golden.make_golden import SCENES

    g = np.load(os.path.join(GOLD, "rays_%s.npz" % name))
    osc = util.oracle_scene(SCENES[name]())
    h = osc.trace(g["o"], g["d"])
    assert np.array_equal(h["prim"], g["prim"]) and np.array_equal(h["front"], g["front"])
    hit = g["prim"] != 0xFFFFFFFF
    assert hit.sum() > 1000 and (~hit).sum() > 200 and g["nan_flag"].sum() == 0
    for f in ("t", "n", "p"):
        assert np.array_equal(h[f][hit].view(np.uint32), g[f][hit].view(np.uint32)), f


def test_golden_images():
    from tests.golden.make_golden import SCENES

    g = np.load(os.path.join(GOLD, "images.npz"))
    for name in SCENES:
        w, hgt, spp, depth, rays = [int(x) for x in g[name + "_meta"]]
        img, r = util.oracle_scene(SCENES[name]()).render(w, hgt, spp, depth)
        assert r == rays and np.array_equal(img.view(np.uint32), g[name + "_img"].view(np.uint32)), name
        assert np.isfinite(img).all() and img.max() > 0


def test_golden_scatter():
    rows = json.load(open(os.path.join(GOLD, "scatter_kat.json")))
    b = scenes.SceneBuilder()
    mats = [b.lambertian((0.4, 0.2, 0.1)), b.metal((0.7, 0.6, 0.5), 0.3), b.metal((0.9, 0.9, 0.9), 7.0), b.dielectric(1.5),
            b.diffuse_light((4, 4, 4))]
    b.sphere((0, 0, 0), 1.0, mats[0])
    b.camera((0, 0, 5), (0, 0, 0), (0, 1, 0), 40.0, 1.0, 0.0, 5.0)
    osc = util.oracle_scene(b)
    assert len(rows) > 200
    kinds = set()
    for r in rows:
        hit, rec = osc.hit_prim(0, r["o"], r["d"])
        assert hit and rec["t"] == f32(r["t"])
        ok, att, so, sd = osc.scatter(r["mat"], r["o"], r["d"], rec, sphere_pt=r["sphere_pt"], xi=r["xi"])
        assert ok == r["ok"]
        assert np.array_equal(att, np.array(r["atten"], np.float32)) and np.array_equal(so, np.array(r["so"], np.float32))
        assert np.array_equal(sd, np.array(r["sd"], np.float32))
        kinds.add((r["mat"], ok, int(rec["front"])))
    assert len(kinds) >= 8  # both faces, absorbed and scattered metal, reflect and refract


def test_orbit_of_shards_equals_whole():
    """Sample ranges are independent: summing two half renders reproduces the full one exactly in fp64."""
    osc = one(scenes.default_scene())
    full, r = osc.render(64, 36, 4, 5)
    a, ra = osc.render(64, 36, 2, 5, sample0=0)
    c, rc = osc.render(64, 36, 2, 5, sample0=2)
    assert ra + rc == r
    assert np.allclose(a.astype(np.float64) + c, full, rtol=1e-6, atol=1e-7)


def test_tree_traversal_baseline_walks_the_same_paths():
    """bench.py's second CPU figure (nearest hits through the library's host-built tree instead of the list scan) is a
    timing baseline, not the reference's algorithm; on a scene of spheres it must still walk the scan's paths."""
    from pathtracer_amd import api

    b = scenes.random_spheres_scene()
    osc = util.oracle_scene(b)
    nodes, order, _ = api.bvh_build_host(b.desc())
    ref, rays = osc.render(96, 54, 2, 50)
    got, rays_tree = osc.render_bvh(nodes, order, 96, 54, 2, 50)
    assert rays_tree == rays
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))

