"""Ad-hoc: pt_scene_upload (build) against pt_scene_update (refit) on the large scenes."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from pathtracer_amd import api, scenes
ctx = api.Context(0)
for name, f in (("mixed 100k", scenes.mixed_scene), ("spheres 1M", scenes.sphere_slab_scene)):
    b = f()
    d = b.desc()
    t0 = time.time(); ctx.upload(d); t1 = time.time()
    a = {k: np.array(v, copy=True) for k, v in b.arrays().items()}
    a["prim_param"][1:, 1] += np.float32(0.05)
    b2 = scenes.ArrayScene(a, b.background, b.camera_args)
    d2 = b2.desc()
    t2 = time.time(); ctx.update(d2); t3 = time.time()
    _, st = ctx.render(1920, 1080, 1, 50)
    t4 = time.time(); ctx.upload(d2); t5 = time.time()
    _, st2 = ctx.render(1920, 1080, 1, 50)
    print("%s: upload %.0f ms, update %.0f ms, re-upload %.0f ms; frame after update %.1f ms, after re-upload %.1f ms" % (name, (t1 - t0) * 1e3, (t3 - t2) * 1e3, (t5 - t4) * 1e3, st["ms_total"], st2["ms_total"]), flush=True)
