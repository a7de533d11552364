"""Ad-hoc: find and print the rays of the 1M-sphere scene on which the BVH path and the oracle differ."""
import sys
sys.path.insert(0, ".")
import numpy as np
from pathtracer_amd import api, scenes
from tests import util
ctx = api.Context(0)
b = scenes.sphere_slab_scene()
ctx.upload(b.desc())
osc = util.oracle_scene(b)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
o, d = util.mixed_rays(b, osc, n, seed=55)
ref = osc.trace(o, d)
got = ctx.trace(o, d)
lin = ctx.trace(o, d, mode=api.TRACE_LINEAR)
bad = util.hits_equal(got, ref) & (ref["nan_flag"] == 0)
print("bvh vs oracle mismatches:", int(bad.sum()), " linear vs oracle:", int((util.hits_equal(lin, ref) & (ref["nan_flag"] == 0)).sum()))
a = b.arrays()
for i in np.flatnonzero(bad)[:10]:
    print("ray", i, "o", o[i].tolist(), "d", d[i].tolist())
    print("   oracle t %.9g prim %d   bvh t %.9g prim %d flags %d   linear t %.9g prim %d" % (ref["t"][i], ref["prim"][i], got["t"][i], got["prim"][i], got["flags"][i], lin["t"][i], lin["prim"][i]))
    for p in {int(ref["prim"][i]), int(got["prim"][i])}:
        if p != 0xFFFFFFFF:
            print("   prim", p, "type", a["prim_type"][p], "param", a["prim_param"][p].tolist())
