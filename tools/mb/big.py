"""Ad-hoc: the large scenes under different pipeline settings (time per frame at low spp)."""
import os, sys
sys.path.insert(0, ".")
import numpy as np
from pathtracer_amd import api, scenes
which = sys.argv[1]
b, spp = (scenes.mixed_scene(), 1) if which == "c4" else (scenes.sphere_slab_scene(), 2)
ctx = api.Context(0)
ctx.upload(b.desc())
def run(tag, flags=0, **env):
    for k, v in env.items():
        os.environ[k] = v
    ctx.render(3840, 2160, spp, 50, flags=flags)
    img, st = ctx.render(3840, 2160, spp, 50, flags=flags)
    for k in env:
        del os.environ[k]
    print("%-40s %8.1f ms  %7.1f Mrays/s" % (tag, st["ms_total"], st["rays"] / st["ms_total"] / 1e3), flush=True)
run("k_paths default")
run("split kernels", flags=api.RENDER_SPLIT_KERNELS)
run("k_paths slice 8192", PT_PATHS_SLICE="8192")
run("k_paths block 256", PT_PATHS_BLOCK="256")
run("k_paths block 256 slice 8192", PT_PATHS_BLOCK="256", PT_PATHS_SLICE="8192")
