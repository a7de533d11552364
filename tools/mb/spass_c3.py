"""Ad-hoc: the Cornell box at 1024 x 1024 x 1024 spp with different samples_per_pass (launches per frame)."""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
from pathtracer_amd import api, scenes
ctx = api.Context(0)
ctx.upload(scenes.cornell_scene().desc())
buf = torch.zeros(1024, 1024, 3, dtype=torch.float32, device="cuda")
for spass in (0, 64, 128, 256, 512, 1024):
    ts = []
    for rep in range(3):
        buf.zero_()
        st = ctx.render_device(api.Context.params(1024, 1024, 1024, 50, samples_per_pass=spass), buf.data_ptr(), 0, want_stats=True)
        ts.append(st["ms_total"])
    print("samples_per_pass %4d: %.2f ms, %d launches, %.1f Grays/s" % (spass, min(ts), st["passes"], st["rays"] / min(ts) / 1e6), flush=True)
