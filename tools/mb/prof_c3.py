"""Ad-hoc: section clocks and lane tallies of the path kernel on the Cornell box (run with PT_LIB pointing at the
diagnostic build, `make -C pathtracer_amd/csrc prof`; the report goes to stderr).  usage: prof_c3.py [spp]"""
import sys
sys.path.insert(0, ".")
import torch
from pathtracer_amd import api, scenes

b = scenes.cornell_scene()
ctx = api.Context(0)
ctx.upload(b.desc())
W = H = 1024
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 128
buf = torch.zeros(H, W, 3, dtype=torch.float32, device="cuda")
for rep in range(2):
    st = ctx.render_device(api.Context.params(W, H, spp, 50), buf.data_ptr(), 0, want_stats=(rep == 1))
print("spp %d: %.3f ms, %d rays, %.1f Grays/s" % (spp, st["ms_total"], st["rays"], st["rays"] / st["ms_total"] / 1e6), file=sys.stderr, flush=True)
