#!/usr/bin/env python3
"""Ad-hoc: the C2 frame through pt_render (host buffer out: device-to-host copy of the film inside the call) against
pt_render_device (film stays in HBM), wall time per call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from pathtracer_amd import api

wl = bench.WORKLOADS["c2"]
ctx = api.Context(0)
ctx.upload(wl["scene"]().desc())
w, h, spp, depth = wl["w"], wl["h"], wl["spp"], wl["depth"]
p = api.Context.params(w, h, spp, depth)
acc = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
for _ in range(2):
    ctx.render(w, h, spp, depth)
    ctx.render_device(p, acc.data_ptr(), 0)
torch.cuda.synchronize()
t = []
for _ in range(10):
    t0 = time.perf_counter(); img, st = ctx.render(w, h, spp, depth); t.append(time.perf_counter() - t0)
d = []
for _ in range(10):
    torch.cuda.synchronize(); t0 = time.perf_counter(); ctx.render_device(p, acc.data_ptr(), 0); torch.cuda.synchronize(); d.append(time.perf_counter() - t0)
rays = st["rays"]
print("pt_render (host buffer out): median %.3f ms  %.1f Mrays/s" % (1e3 * np.median(t), rays / np.median(t) / 1e6))
print("pt_render_device (HBM):      median %.3f ms  %.1f Mrays/s" % (1e3 * np.median(d), rays / np.median(d) / 1e6))
