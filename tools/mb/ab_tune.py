#!/usr/bin/env python3
"""Ad-hoc A/B timing of one workload under different pt_tuning settings, interleaved in one process; also checks that
every variant renders the same image bit for bit.
usage: ab_tune.py <workload> <rounds> field=VAL[,field=VAL]... ...   ('-' = the defaults); AB_SPP overrides the spp"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from pathtracer_amd import api

wl = bench.WORKLOADS[sys.argv[1]]
rounds = int(sys.argv[2])
variants = sys.argv[3:]
spp = int(os.environ.get("AB_SPP", wl["spp"]))
ctx = api.Context(0)
desc = wl["scene"]().desc()
ctx.upload(desc)
acc = torch.zeros((wl["h"], wl["w"], 3), dtype=torch.float32, device="cuda")
p = api.Context.params(wl["w"], wl["h"], spp, wl["depth"])


def fields(v):
    return {} if v == "-" else {k: int(x) for k, x in (kv.split("=") for kv in v.split(","))}


reupload = any(k in ("leaf_max", "wide_nodes") for v in variants for k in fields(v))
res = {v: [] for v in variants}
img = {}
for r in range(rounds + 1):
    for v in variants:
        f = fields(v)
        with ctx.tuned(**f):
            if reupload:
                ctx.upload(desc)  # leaf_max and wide_nodes take effect when a scene is installed
            acc.zero_()
            torch.cuda.synchronize()
            st = ctx.render_device(p, acc.data_ptr(), 0, want_stats=True)
            torch.cuda.synchronize()
        if r:
            res[v].append(st["ms_total"])
        else:
            img[v] = (acc.cpu().numpy().copy(), st["rays"])
base = img[variants[0]]
for v in variants:
    a = np.array(res[v])
    same = np.array_equal(img[v][0].view(np.uint32), base[0].view(np.uint32))
    print("%-44s median %.3f ms  min %.3f  rays %d  %.1f Mrays/s  image==first:%s" % (v, np.median(a), a.min(), img[v][1], img[v][1] / np.median(a) / 1e3, same), flush=True)
