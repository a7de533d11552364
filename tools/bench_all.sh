#!/bin/bash
# Runs on the GPU box: bench.py on every workload at its full size, one JSON line each into <out.jsonl>.  A heartbeat line
# goes to stdout every 30 s (the 1M-sphere workload is silent for minutes otherwise).
# Usage: bash tools/bench_all.sh <out.jsonl> [workloads...]
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
cd "$R"
OUT=${1:-gpurun_out/bench_workloads.jsonl}; shift
WLS=${*:-c1 c2 c3 c4 c5}
: > "$OUT"
( while true; do sleep 30; echo "[heartbeat] $(date +%T)"; done ) &
HB=$!
trap 'kill $HB 2>/dev/null' EXIT
for w in $WLS; do
  steps=5; extra=""
  [ "$w" = c4 ] && steps=2
  [ "$w" = c5 ] && { steps=1; extra="--spp 512"; }  # one GPU's share of the 8-GPU job (all 4096 spp take 70 s a step)
  timeout -k 10 900 python bench.py --workload $w --steps $steps --warmup 1 $extra 2> gpurun_out/bench_$w.err | tail -1 >> "$OUT" || { echo "bench $w failed"; tail -5 gpurun_out/bench_$w.err; exit 1; }
  tail -1 "$OUT" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', 'ms/step %.3f' % d['ms_per_step'], 'Mrays/s %.0f' % d['value'], 'frac %.4f' % d['roofline']['frac'], d['roofline']['kernel'])"
done
