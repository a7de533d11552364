"""Ad-hoc GPU diagnostics (not collected by pytest): margins of the parity tests and first timings."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from pathtracer_amd import api, scenes
from tests import util

ctx = api.Context(0)
for name, f in (("default", scenes.default_scene), ("cornell", scenes.cornell_scene), ("spheres", scenes.random_spheres_scene),
                ("mixed_1k", lambda: scenes.mixed_scene(n=1000, seed=3, extent=20.0, n_materials=64))):
    b = f(); ctx.upload(b.desc()); osc = util.oracle_scene(b)
    o, d = util.mixed_rays(b, osc, 20000, seed=101)
    ref = osc.trace(o, d); got = ctx.trace(o, d)
    print(name, "rays", len(o), "hit frac", (ref["prim"] != api.NO_PRIM).mean(), "nan", ref["nan_flag"].sum(),
          "fallback flags", int((got["flags"] & 1).sum()), "mismatch", int(util.hits_equal(got, ref).sum()))
    w, h, spp, depth = (160, 90, 4, 50)
    ref_img, rr = osc.render(w, h, spp, depth); img, st = ctx.render(w, h, spp, depth)
    diff = np.abs(img - ref_img); tol = 2e-5 * np.maximum(1, np.abs(ref_img))
    print("   image: rays", st["rays"], rr, "bad px", int((diff > tol).any(axis=2).sum()), "of", w * h, "max diff", diff.max(),
          "exact-equal frac", (img.view(np.uint32) == ref_img.view(np.uint32)).mean(), "fallbacks", st["exact_fallbacks"])

b = scenes.random_spheres_scene(); ctx.upload(b.desc())
for spp_pass in (0, 1, 4):
    for rep in range(2):
        t = time.time(); img, st = ctx.render(1920, 1080, 64, 50, samples_per_pass=spp_pass); dt = time.time() - t
        print("C2 1080p 64spp depth50 pass=%d: rays %d  device ms %.1f  wall %.3f s  -> %.1f Mrays/s" % (spp_pass, st["rays"], st["ms_total"], dt, st["rays"] / st["ms_total"] / 1e3))
img, st = ctx.render(1920, 1080, 64, 50, flags=api.RENDER_STAGE_TIMES)
print({k: st[k] for k in ("ms_total", "ms_raygen", "ms_intersect", "ms_shade", "ms_film", "launches_intersect", "passes")})
img, st = ctx.render(1920, 1080, 8, 50, flags=api.RENDER_COUNT_TESTS)
print("box tests/ray %.2f prim tests/ray %.2f" % (st["box_tests"] / st["rays"], st["prim_tests"] / st["rays"]))
