import sys, os
sys.path.insert(0, ".")
from pathtracer_amd import api, scenes
ctx = api.Context(0)
b = scenes.random_spheres_scene(); ctx.upload(b.desc())
fl = int(os.environ.get("PT_FLAGS", "0"))
for rep in range(3):
    img, st = ctx.render(1920, 1080, 64, 50, flags=fl)
env = {k: v for k, v in os.environ.items() if k.startswith("PT_")}
print("%s: device ms %.2f -> %.1f Mrays/s  rays %d" % (env, st["ms_total"], st["rays"] / st["ms_total"] / 1e3, st["rays"]), flush=True)
