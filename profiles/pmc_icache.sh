#!/bin/bash
# One counter pass over a single bench step (GPU box): instruction-cache behaviour of the path kernel.
WL=${1:-c2}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/pmci
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/sq" -- \
   python3 "$REPO/bench.py" --workload "$WL" --steps 1 --warmup 1 --no-cpu-baseline > "$OUT/sq.log" 2>&1 || { tail -5 "$OUT/sq.log"; exit 1; }
python3 "$REPO/profiles/summarize_pmc.py" "$OUT" "$WL" | grep -E "^k_paths<false"
