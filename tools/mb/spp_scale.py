"""Ad-hoc: time and counters of the large scenes against samples per pixel and slice size."""
import sys, os
sys.path.insert(0, ".")
from pathtracer_amd import api, scenes
which = sys.argv[1] if len(sys.argv) > 1 else "c4"
b = scenes.mixed_scene() if which == "c4" else scenes.sphere_slab_scene()
ctx = api.Context(0)
ctx.upload(b.desc())
W, H = 3840, 2160
ctx.render(W, H, 1, 50)
for spp in [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,2,4,8").split(",")]:
    _, st = ctx.render(W, H, spp, 50, flags=api.RENDER_COUNT_TESTS if os.environ.get("COUNT") else 0)
    print("%s spp %d: %.1f ms, %.1f ms/spp, %.0f Mrays/s, rays/sample %.2f, box %.1f prim %.1f fallbacks %d replays %d passes %d" % (
        which, spp, st["ms_total"], st["ms_total"] / spp, st["rays"] / st["ms_total"] / 1e3, st["rays"] / (W * H * spp),
        st["box_tests"] / max(st["rays"], 1), st["prim_tests"] / max(st["rays"], 1), st["exact_fallbacks"], st["exact_replays"], st["passes"]), flush=True)
