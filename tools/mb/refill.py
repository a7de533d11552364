"""Ad-hoc: k_paths against k_paths_refill (image identity, time per refill threshold)."""
import os, sys  # PT_PATHS_STREAM: 0 = k_paths, 1 = k_paths_stream
sys.path.insert(0, ".")
import numpy as np
from pathtracer_amd import api, scenes
ctx = api.Context(0)
which = sys.argv[1] if len(sys.argv) > 1 else "c2"
if which == "c2":
    b, w, h, spp = scenes.random_spheres_scene(), 1920, 1080, 64
elif which == "c4":
    b, w, h, spp = scenes.mixed_scene(), 3840, 2160, 1
else:
    b, w, h, spp = scenes.sphere_slab_scene(), 3840, 2160, 2
ctx.upload(b.desc())
def run(tag):
    ctx.render(w, h, spp, 50)
    ts = []
    for _ in range(3):
        img, st = ctx.render(w, h, spp, 50)
        ts.append(st["ms_total"])
    print("%-28s %.2f ms (min %.2f)  rays %d replays %d" % (tag, float(np.median(ts)), min(ts), st["rays"], st["exact_replays"]), flush=True)
    return img
os.environ["PT_PATHS_STREAM"] = "0"
ref = run("k_paths")
os.environ["PT_PATHS_STREAM"] = "1"
for m in (sys.argv[2:] or ["1", "8", "16", "32", "48"]):
    os.environ["PT_PATHS_STREAM_MIN"] = m
    img = run("stream, min waiting %s" % m)
    print("   identical to k_paths:", bool(np.array_equal(ref, img)), " max |d| %.3g" % float(np.abs(ref - img).max()), flush=True)
