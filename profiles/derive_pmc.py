#!/usr/bin/env python3
"""Turns one collect_pmc.sh run (gpurun_out/pmc/summary.json + the bench line in sq.log) into the per-launch
counter file kept under profiles/: counters of the dominant kernel and the ratios derived from them.
Usage: derive_pmc.py <pmc dir> <out.json> "<workload description>" """
import json, os, sys

root, out, desc = sys.argv[1], sys.argv[2], sys.argv[3]
summ = json.load(open(os.path.join(root, "summary.json")))
kern = next(k for k in summ["kernels"] if k.startswith("k_paths"))
c = {n: v["per_launch"] for n, v in summ["kernels"][kern].items()}
bench = None
for line in open(os.path.join(root, "sq.log")):
    if line.startswith("{"):
        bench = json.loads(line)
rays = bench["roofline"]["rays_per_launch"]
cycles = c["GRBM_GUI_ACTIVE"] / 8.0  # summed over the 8 XCDs
mem = 2.0 * c["FETCH_SIZE"] * 1024.0 + c["WRITE_SIZE"] * 1024.0
derived = {
    "launch_ms_at_2.4GHz": cycles / 2.4e6,
    "rays_per_launch": rays,
    "memory_side_bytes_per_launch": mem,
    "memory_side_bytes_per_ray": mem / rays,
    "valu_wave_instructions_per_ray": c["SQ_INSTS_VALU"] / rays,
    "lane_utilisation": c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"]),
    "valu_instructions_x4_over_simd_cycles": 4.0 * c["SQ_ACTIVE_INST_VALU"] / (cycles * 1024.0),
    "waves_resident_fraction_of_launch": 4.0 * c["SQ_WAVE_CYCLES"] / (c["SQ_WAVES"] * cycles),
    "wave_cycles_waiting_fraction": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
    "l2_hit_fraction": c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]),
}
derived["memory_side_GB_per_s"] = mem / (cycles / 2.4e9) / 1e9  # against 8000 GB/s of HBM3E (Infinity-Cache hits included)
derived["memory_side_fraction_of_hbm_peak"] = derived["memory_side_GB_per_s"] / 8000.0
derived["valu_lane_slots_used_fraction"] = derived["valu_instructions_x4_over_simd_cycles"] * derived["lane_utilisation"]  # share of the vector ALUs' lane-cycles doing work
notes = ("FETCH_SIZE/WRITE_SIZE are in KiB on this ROCm (x1024 here); gfx950 counts 128-B read requests as 64 B, hence 2 x FETCH_SIZE; "
         "Infinity-Cache hits are included. GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_WAVE_CYCLES and SQ_ACTIVE_INST_* are in units of "
         "4 cycles. VALU instructions do not all issue in 4 cycles on gfx950 (tools/mb/valu_tput.hip), so the x4 ratio is an index, not a utilisation.")
json.dump({"workload": desc, "kernel": kern, "counters_per_launch": c, "derived": derived, "notes": notes}, open(out, "w"), indent=1)
print(json.dumps(derived, indent=1))
