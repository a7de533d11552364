"""Ad-hoc: section clocks and lane tallies of k_paths / k_paths_ring on the random-spheres frame at several sample counts
(run with PT_LIB pointing at the diagnostic build, `make -C pathtracer_amd/csrc prof`; the report goes to stderr).
usage: prof_c2.py [field=VAL,...] spp..."""
import sys
sys.path.insert(0, ".")
import torch
from pathtracer_amd import api, scenes

b = scenes.random_spheres_scene()
ctx = api.Context(0)
ctx.upload(b.desc())
args = sys.argv[1:]
if args and "=" in args[0]:
    ctx.configure(**{k: int(v) for k, v in (kv.split("=") for kv in args[0].split(","))})
    args = args[1:]
W, H = 1920, 1080
buf = torch.zeros(H, W, 3, dtype=torch.float32, device="cuda")
for spp in [int(a) for a in args] or [64, 8]:
    for rep in range(2):
        st = ctx.render_device(api.Context.params(W, H, spp, 50), buf.data_ptr(), 0, want_stats=(rep == 1))
    print("spp %d: %.3f ms, %d rays, %.1f Grays/s" % (spp, st["ms_total"], st["rays"], st["rays"] / st["ms_total"] / 1e6), file=sys.stderr, flush=True)
