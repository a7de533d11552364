#!/usr/bin/env python3
"""Turns one collect_pmc.sh run (gpurun_out/pmc/summary.json + the bench line in sq.log) into the per-launch
counter file kept under profiles/ -- counters of the dominant kernel and the ratios derived from them -- and enters
the workload's line into profiles/pmc_latest.json, which bench.py reads.
Usage: derive_pmc.py <pmc dir> <out.json> "<workload description>" <workload key> [isa mix json]"""
import json, os, sys

root, out, desc, key = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4]
isa = json.load(open(sys.argv[5])) if len(sys.argv) > 5 else None
here = os.path.dirname(os.path.abspath(__file__))
summ = json.load(open(os.path.join(root, "summary.json")))
kern = next((k for k in summ["kernels"] if k.startswith("k_paths_stream<false")), None) or next((k for k in summ["kernels"] if k.startswith("k_paths_ring<false")), None) or next(k for k in summ["kernels"] if k.startswith("k_paths<false"))
c = {n: v["per_launch"] for n, v in summ["kernels"][kern].items()}
bench = None
for line in open(os.path.join(root, "sq.log")):
    if line.startswith("{"):
        bench = json.loads(line)
rays = bench["roofline"]["rays_per_launch"]
cycles = c["GRBM_GUI_ACTIVE"] / 8.0  # summed over the 8 XCDs
simd_cycles = cycles * 1024.0        # 256 CUs x 4 SIMDs
mem = 2.0 * c["FETCH_SIZE"] * 1024.0 + c["WRITE_SIZE"] * 1024.0
derived = {
    "launch_ms_at_2.4GHz": cycles / 2.4e6,
    "rays_per_launch": rays,
    "memory_side_bytes_per_launch": mem,
    "memory_side_bytes_per_ray": mem / rays,
    "valu_wave_instructions_per_ray": c["SQ_INSTS_VALU"] / rays,
    "lane_utilisation": c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"]),
    "waves_resident_fraction_of_launch": 4.0 * c["SQ_WAVE_CYCLES"] / (c["SQ_WAVES"] * cycles),
    "wave_cycles_waiting_fraction": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
    "l2_hit_fraction": c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]),
}
derived["memory_side_GB_per_s"] = mem / (cycles / 2.4e9) / 1e9  # against 8000 GB/s of HBM3E (Infinity-Cache hits included)
derived["memory_side_fraction_of_hbm_peak"] = derived["memory_side_GB_per_s"] / 8000.0
# Issue occupancy of the vector ALUs, weighted by instruction mix: dynamic class counts (SQ_INSTS_VALU_*) times the issue
# cost of the class measured by tools/mb/valu_tput.hip (profiles/r02_valu_tput.txt, cycles per wave-instruction per SIMD
# with 8 waves resident).  The 32-bit-integer class and the remainder (moves, selects, compares, min / max, lane ops)
# mix opcodes of different cost: their mean cost comes from the kernel's code (profiles/isa_mix.py).
cost = {"add_f32": 2.4, "mul_f32": 2.4, "fma_f32": 2.4, "trans_f32": 8.0, "cvt": 4.2, "int64": 4.4, "int32": 3.0, "other": 3.6}
if isa:
    for k, v in isa["classes"].items():
        cost[k] = v["mean_issue_cycles"]
if "SQ_INSTS_VALU_FMA_F32" in c:
    dyn = {"add_f32": c["SQ_INSTS_VALU_ADD_F32"], "mul_f32": c["SQ_INSTS_VALU_MUL_F32"], "fma_f32": c["SQ_INSTS_VALU_FMA_F32"],
           "trans_f32": c["SQ_INSTS_VALU_TRANS_F32"], "cvt": c["SQ_INSTS_VALU_CVT"], "int64": c["SQ_INSTS_VALU_INT64"], "int32": c["SQ_INSTS_VALU_INT32"]}
    dyn["other"] = c["SQ_INSTS_VALU"] - sum(dyn.values())
    issue = sum(dyn[k] * cost[k] for k in dyn)
    derived["valu_issue"] = {
        "wave_instructions_per_ray": {k: v / rays for k, v in dyn.items()},
        "issue_cycles_per_instruction": cost,
        "issue_cycles_over_simd_cycles": issue / simd_cycles,
        "mean_issue_cycles_per_instruction": issue / c["SQ_INSTS_VALU"],
        "lane_utilisation": derived["lane_utilisation"],
        "lane_cycles_in_use_fraction": issue / simd_cycles * derived["lane_utilisation"],
    }
notes = ("FETCH_SIZE/WRITE_SIZE are in KiB on this ROCm (x1024 here); gfx950 counts 128-B read requests as 64 B, hence 2 x FETCH_SIZE; "
         "Infinity-Cache hits are included. GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_WAVE_CYCLES and SQ_ACTIVE_INST_* are in units of "
         "4 cycles. valu_issue.issue_cycles_over_simd_cycles is the share of the launch during which the vector ALUs of a SIMD are issuing, "
         "if every instruction costs what it costs back to back in the microbenchmark.")
json.dump({"workload": desc, "kernel": kern, "counters_per_launch": c, "derived": derived, "notes": notes}, open(out, "w"), indent=1)
latest_path = os.path.join(here, "pmc_latest.json")
latest = json.load(open(latest_path)) if os.path.exists(latest_path) else {}
latest[key] = {"kernel": kern, "source": os.path.basename(out), "hbm_bytes_per_ray": mem / rays, "valu_issue": derived.get("valu_issue")}
json.dump(latest, open(latest_path, "w"), indent=1)
print(json.dumps(derived, indent=1))
