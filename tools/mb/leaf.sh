#!/bin/bash
# Ad-hoc: bench time per BVH leaf size (PT_LEAF_MAX is read at scene upload).
for lm in 1 2 3 4 6 8; do
  PT_LEAF_MAX=$lm timeout -k 10 120 python bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('leaf_max $lm', round(d['ms_per_step'],3), 'ms')" || exit 1
done
