"""Ad-hoc: the other benchmark workloads at reduced spp (timing, ray statistics, fallbacks)."""
import sys, time
sys.path.insert(0, ".")
from pathtracer_amd import api, scenes
ctx = api.Context(0)
which = sys.argv[1:] or ["c1", "c3", "c4", "c5"]
cfg = {"c1": (scenes.default_scene, 400, 225, 16), "c3": (scenes.cornell_scene, 1024, 1024, 16),
       "c4": (scenes.mixed_scene, 3840, 2160, 1), "c5": (scenes.sphere_slab_scene, 3840, 2160, 2)}
for name in which:
    f, w, h, spp = cfg[name]
    t = time.time(); b = f(); d = b.desc(); t1 = time.time(); ctx.upload(d); t2 = time.time()
    print("%s: build arrays %.2fs upload+BVH %.2fs objects %d" % (name, t1 - t, t2 - t1, d.c.n_prims), flush=True)
    for rep in range(2):
        img, st = ctx.render(w, h, spp, 50)
    print("   %dx%d %dspp depth50: %.2f ms  %.1f Mrays/s  rays/primary %.2f  replays %d fallbacks %d" % (w, h, spp, st["ms_total"], st["rays"] / st["ms_total"] / 1e3, st["rays"] / st["primary_rays"], st["exact_replays"], st["exact_fallbacks"]), flush=True)
    img, st = ctx.render(w, h, 1, 50, flags=api.RENDER_COUNT_TESTS)
    print("   B %.2f  P %.2f" % (st["box_tests"] / st["rays"], st["prim_tests"] / st["rays"]), flush=True)
