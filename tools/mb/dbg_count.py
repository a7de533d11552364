"""Ad-hoc: ray / box-test / object-test counts of the counting kernel variants of every pipeline on one small frame (they must agree)."""
import sys
sys.path.insert(0, ".")
from pathtracer_amd import api, scenes
ctx = api.Context(0)
for name, b, args in (("spheres", scenes.random_spheres_scene(), (192, 108, 4, 50)), ("spheres-big", scenes.random_spheres_scene(), (960, 540, 16, 50)), ("cornell", scenes.cornell_scene(), (128, 128, 16, 50))):
    ctx.upload(b.desc())
    for pipe in (0, 2, 0, 2):
        with ctx.tuned(pipeline=pipe):
            img, st = ctx.render(*args, flags=api.RENDER_COUNT_TESTS)
            img2, st2 = ctx.render(*args)
        print(name, pipe, {k: st[k] for k in ("rays", "box_tests", "prim_tests", "kernel")}, st2["rays"], (img == img2).all())
