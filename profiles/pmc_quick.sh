#!/bin/bash
# One quick SQ-counter pass over a single bench step (GPU box).  Usage: bash profiles/pmc_quick.sh [workload]
WL=${1:-c2}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/pmcq
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/sq" -- \
   python3 "$REPO/bench.py" --workload "$WL" --steps 1 --warmup 1 --no-cpu-baseline > "$OUT/sq.log" 2>&1 || { tail -5 "$OUT/sq.log"; exit 1; }
python3 "$REPO/profiles/summarize_pmc.py" "$OUT" "$WL" | grep -E "^k_paths<false|^k_intersect<false|^k_shade"
