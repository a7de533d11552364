#!/bin/bash
# Ad-hoc: times bench.py with each library build under build_ab/ (alternating, two rounds).  usage: ab_lib.sh <workload> [bench flags]
WL=${1:-c2}; shift
for round in 1 2; do
  for lib in build_ab/*.so; do
    PT_LIB=$PWD/$lib timeout -k 10 200 python bench.py --workload $WL --steps 10 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('%-40s %.3f ms/step  %.1f Mrays/s' % ('$lib', j['ms_per_step'], j['value']))
" || exit 1
  done
done
