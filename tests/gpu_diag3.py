import sys, os
sys.path.insert(0, ".")
from pathtracer_amd import api, scenes
ctx = api.Context(0)
b = scenes.random_spheres_scene(); ctx.upload(b.desc())
for rep in range(2):
    img, st = ctx.render(1920, 1080, 64, 50)
img2, st2 = ctx.render(1920, 1080, 64, 50, flags=api.RENDER_STAGE_TIMES)
print("PT_REFILL=%s: device ms %.2f -> %.1f Mrays/s   intersect %.2f shade %.2f" % (os.environ.get("PT_REFILL"), st["ms_total"], st["rays"] / st["ms_total"] / 1e3, st2["ms_intersect"], st2["ms_shade"]), flush=True)
