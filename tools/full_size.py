"""Ad-hoc: one pt_render call of a large workload at its FULL size (C5: 4K, 4096 spp, depth 50 on one GPU; about
two minutes), with a heartbeat so that the GPU box does not take the silent call for a hang.  Checks that the sums
are finite, that the ray count is that of eight 512-spp shards' worth of samples, and that the first shard's mean
image agrees statistically with the whole.  Usage: python tools/full_size.py [c5|c4] [spp]"""
import sys
import threading
import time

import numpy as np

sys.path.insert(0, ".")
from pathtracer_amd import api, scenes  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "c5"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else (4096 if which == "c5" else 256)
b = scenes.mixed_scene() if which == "c4" else scenes.sphere_slab_scene()
W, H, depth = 3840, 2160, 50

stop = threading.Event()


def heartbeat():
    t0 = time.time()
    while not stop.wait(30.0):
        print("... %d s" % (time.time() - t0), flush=True)


threading.Thread(target=heartbeat, daemon=True).start()
ctx = api.Context(0)
ctx.upload(b.desc())
shard, st_s = ctx.render(W, H, spp // 8, depth)  # samples [0, spp/8)
print("%s shard %d spp: %.0f ms, %.0f Mrays/s, %d rays" % (which, spp // 8, st_s["ms_total"], st_s["rays"] / st_s["ms_total"] / 1e3, st_s["rays"]), flush=True)
full, st = ctx.render(W, H, spp, depth)
stop.set()
print("%s full %d spp: %.0f ms, %.0f Mrays/s, %d rays, %d primary, %d passes, fallbacks %d" % (
    which, spp, st["ms_total"], st["rays"] / st["ms_total"] / 1e3, st["rays"], st["primary_rays"], st["passes"], st["exact_fallbacks"]), flush=True)
assert np.isfinite(full).all(), "non-finite sums"
assert st["primary_rays"] == W * H * spp, (st["primary_rays"], W * H * spp)
ratio = st["rays"] / (8.0 * st_s["rays"])
print("rays(full) / (8 x rays(shard)) = %.5f" % ratio)
assert abs(ratio - 1.0) < 5e-3
m_full, m_shard = full / spp, shard / (spp // 8)
rel = float(np.sqrt(np.mean((m_full - m_shard) ** 2)) / max(np.sqrt(np.mean(m_full ** 2)), 1e-30))
print("relative RMSE of the first shard's mean image against the whole: %.4f" % rel)
assert rel < 0.2
print("OK")
