#!/bin/bash
# One and two steps in flight and the bench's own choice (0), at the full frame and at an eighth of its samples (one
# GPU's share of eight) with the collective forced: value, ms per step, the launch duration the roofline is priced on.
set -e
out=gpurun_out/inflight_ab.jsonl
: > $out
for spp in 0 8; do
  for fl in 1 2 0; do
    extra=""
    [ $spp -gt 0 ] && extra="--spp $spp --force-collective"
    timeout -k 10 120 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --in-flight $fl $extra 2>/dev/null >> $out
  done
done
python - <<'PY'
import json
for l in open("gpurun_out/inflight_ab.jsonl"):
    d = json.loads(l); r = d["roofline"]
    print(d["config"]["workload"][-40:], "| in flight %d alone %.2f ms | value %.0f ms/step %.3f ms/launch %.3f frac %.4f" % (
        d["config"]["steps_in_flight"], d["config"]["ms_step_alone"], d["value"], d["ms_per_step"], r["ms_per_launch"], r["frac"]))
PY
