import sys, os
sys.path.insert(0, ".")
from pathtracer_amd import api, scenes
ctx = api.Context(0)
b = scenes.random_spheres_scene(); ctx.upload(b.desc())
fl = int(os.environ.get("PT_FLAGS", "0"))
for rep in range(3):
    img, st = ctx.render(1920, 1080, 64, 50, flags=fl)
print("flags=%d BLOCK=%s LDS=%s PER_CU=%s SLICE=%s: device ms %.2f -> %.1f Mrays/s  rays %d" % (fl, os.environ.get("PT_PATHS_BLOCK"), os.environ.get("PT_PATHS_LDS"), os.environ.get("PT_PATHS_PER_CU"), os.environ.get("PT_PATHS_SLICE"), st["ms_total"], st["rays"] / st["ms_total"] / 1e3, st["rays"]), flush=True)
