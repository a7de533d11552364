#!/bin/bash
# Runs on the GPU box: rocprofv3 --kernel-trace --stats over the default bench command; the per-kernel
# summary lands in gpurun_out/stats/.  Usage: bash profiles/collect_stats.sh [bench.py flags]
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/stats
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- \
    python3 "$REPO/bench.py" --no-cpu-baseline "$@" > "$OUT/bench.log" 2>&1 || { echo "stats pass failed"; tail -5 "$OUT/bench.log"; exit 1; }
tail -1 "$OUT/bench.log" | cut -c1-300
find "$OUT" -name "*kernel_stats.csv" | head -1 | xargs -r cat | cut -c1-220 | head -8
