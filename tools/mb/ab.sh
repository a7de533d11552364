#!/bin/bash
# Ad-hoc A/B timing of library variants on the default bench workload: ab.sh libA.so libB.so ...
for rep in 1 2; do
  for lib in "$@"; do
    PT_LIB=$PWD/$lib timeout -k 10 120 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', round(d['ms_per_step'],3), 'ms')" || exit 1
  done
done
