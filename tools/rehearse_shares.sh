#!/bin/bash
# Runs on the GPU box: one GPU's share of each strong-scaled configuration at N GPUs, with and without the collective
# forced (one rank), so that the fixed cost per step that bounds strong-scaling efficiency is a number.
# Usage: bash tools/rehearse_shares.sh <out.jsonl>
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
cd "$R"
OUT=${1:-gpurun_out/rehearse.jsonl}
: > "$OUT"
run() {
  timeout -k 10 300 python bench.py "$@" --no-cpu-baseline 2> gpurun_out/rehearse.err | tail -1 >> "$OUT" || { echo "failed: $*"; tail -5 gpurun_out/rehearse.err; exit 1; }
  tail -1 "$OUT" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', 'ms/step %.3f' % d['ms_per_step'], 'Mrays/s %.0f' % d['value'], 'kernel ms %.3f' % d['roofline']['ms_per_launch'])"
}
for spp in 64 32 16 8; do
  run --workload c2 --spp $spp --steps 20 --warmup 3 --in-flight 1
  run --workload c2 --spp $spp --steps 20 --warmup 3 --in-flight 1 --force-collective
done
# the bench's own choice of steps in flight (two when a step alone takes less than 8 ms)
for spp in 64 32 16 8; do
  run --workload c2 --spp $spp --steps 20 --warmup 3 --force-collective
done
for spp in 1024 512 256 128; do
  run --workload c3 --spp $spp --steps 4 --warmup 1 --force-collective
done
