import sys, time
import numpy as np
sys.path.insert(0, ".")
from pathtracer_amd import api, scenes
ctx = api.Context(0)
b = scenes.random_spheres_scene(); ctx.upload(b.desc())
for name, fl in (("lds", 0), ("global", api.RENDER_NO_LDS_SCENE), ("lds", 0), ("global", api.RENDER_NO_LDS_SCENE)):
    for rep in range(3):
        img, st = ctx.render(1920, 1080, 64, 50, flags=fl)
    print("%s: device ms %.2f -> %.1f Mrays/s" % (name, st["ms_total"], st["rays"] / st["ms_total"] / 1e3), flush=True)
    img2, st = ctx.render(1920, 1080, 64, 50, flags=fl | api.RENDER_STAGE_TIMES)
    print("   ", {k: round(st[k], 2) for k in ("ms_total", "ms_raygen", "ms_intersect", "ms_shade", "ms_film")}, flush=True)
