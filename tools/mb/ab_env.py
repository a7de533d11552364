#!/usr/bin/env python3
"""Ad-hoc A/B timing of one workload under different environment settings, interleaved in one process.
usage: ab_env.py <workload> <rounds> NAME=VAL[,NAME=VAL]... ...   ('-' = no setting)"""
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
ctx.upload(wl["scene"]().desc())
acc = torch.zeros((wl["h"], wl["w"], 3), dtype=torch.float32, device="cuda")
p = api.Context.params(wl["w"], wl["h"], spp, wl["depth"])
def setenv(v):
    for kv in v.split(","):
        if kv == "-": continue
        k, x = kv.split("=")
        os.environ[k] = x
def unsetenv(v):
    for kv in v.split(","):
        if kv == "-": continue
        os.environ.pop(kv.split("=")[0], None)
res = {v: [] for v in variants}
img = {}
for r in range(rounds + 1):
    for v in variants:
        setenv(v)
        if os.environ.get("AB_REUPLOAD"): ctx.upload(wl["scene"]().desc())
        acc.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        st = ctx.render_device(p, acc.data_ptr(), 0, want_stats=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        unsetenv(v)
        if r: res[v].append(st["ms_total"])
        else: img[v] = (acc.cpu().numpy().copy(), st["rays"])
base = img[variants[0]]
for v in variants:
    a = np.array(res[v])
    same = np.array_equal(img[v][0].view(np.uint32), base[0].view(np.uint32))
    print("%-40s median %.3f ms  min %.3f  rays %d  %.1f Mrays/s  image==first:%s" % (v, np.median(a), a.min(), img[v][1], img[v][1] / np.median(a) / 1e3, same), flush=True)
