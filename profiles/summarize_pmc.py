#!/usr/bin/env python3
"""Aggregates rocprofv3 --pmc CSVs (gpurun_out/pmc/<group>/**/ *counter_collection.csv) per kernel:
sum and per-launch mean of every counter.  FETCH_SIZE / WRITE_SIZE are in KiB-like units of 1024 B per the
rocprofv3 derived-metric definition; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x
(TCC_EA0_RDREQ tallied at 64 B for 128-B requests), so both raw and x2 figures are written."""
import csv, glob, json, os, sys, collections

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
workload = sys.argv[2] if len(sys.argv) > 2 else "c2"
agg = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[k][r["Counter_Name"]].add((os.path.relpath(f, root).split(os.sep)[0], r["Dispatch_Id"]))  # (pass, dispatch): a counter may sit in two passes
out = {}
for k in sorted(agg):
    out[k] = {c: {"sum": v, "launches": len(launches[k][c]), "per_launch": v / max(1, len(launches[k][c]))} for c, v in sorted(agg[k].items())}
summary = {"workload": workload, "kernels": out}
ki = next((k for k in out if k.startswith("k_paths_stream<false")), None) or next((k for k in out if k.startswith("k_paths_ring<false")), None) or next((k for k in out if k.startswith("k_paths<false")), None) or next((k for k in out if k.startswith("k_intersect<false")), None)
res = {}
if ki and "FETCH_SIZE" in out[ki] and "WRITE_SIZE" in out[ki]:
    fetch = out[ki]["FETCH_SIZE"]["per_launch"] * 1024.0
    write = out[ki]["WRITE_SIZE"]["per_launch"] * 1024.0
    res = {"kernel": ki, "fetch_bytes_per_launch_raw": fetch, "write_bytes_per_launch": write,
           "hbm_bytes_per_launch": 2.0 * fetch + write,
           "note": "memory-side bytes per launch: 2 x FETCH_SIZE (gfx950 counts 128-B read requests as 64 B) + WRITE_SIZE; Infinity-Cache hits are included"}
summary["derived"] = res
json.dump(summary, open(os.path.join(root, "summary.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
for k in out:
    print(k, {c: round(v["per_launch"], 1) for c, v in out[k].items()})
