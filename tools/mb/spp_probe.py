"""Ad-hoc: which sample index of the 100k-object scene is slow."""
import sys, os
sys.path.insert(0, ".")
from pathtracer_amd import api, scenes
b = scenes.mixed_scene()
ctx = api.Context(0)
ctx.upload(b.desc())
W, H = 3840, 2160
ctx.render(W, H, 1, 50)
for k in (3, 4):
    _, st = ctx.render(W, H, 1, 50, sample_offset=k)
    print("sample %d: %.1f ms rays %d" % (k, st["ms_total"], st["rays"]), flush=True)
