"""Ad-hoc: cost of the ray-cone widening on the bench scene (time, box and object tests per ray)."""
import os, sys
sys.path.insert(0, ".")
import numpy as np
from pathtracer_amd import api, scenes
b = scenes.random_spheres_scene()
for k, ad in (("1", "1"), ("1", "0"), ("0", "1"), ("0", "0")):
    os.environ["PT_KAPPA_SCALE"] = k
    os.environ["PT_KAPPA_ADAPTIVE"] = ad
    ctx = api.Context(0)
    ctx.upload(b.desc())
    for _ in range(2):
        ctx.render(1920, 1080, 64, 50)
    ts = [ctx.render(1920, 1080, 64, 50)[1]["ms_total"] for _ in range(5)]
    _, st = ctx.render(1920, 1080, 4, 50, flags=api.RENDER_COUNT_TESTS)
    print("kappa x%s adaptive %s: %.2f ms  B %.2f P %.2f" % (k, ad, float(np.median(ts)), st["box_tests"] / st["rays"], st["prim_tests"] / st["rays"]), flush=True)
    ctx.close()
