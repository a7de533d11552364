#!/bin/bash
# Ad-hoc: leaf size on the large scenes (PT_LEAF_MAX is read at scene upload).
for lm in 2 4 6 8; do
  echo "leaf_max $lm"
  PT_LEAF_MAX=$lm timeout -k 10 200 python tools/gpu_diag4.py "$@" 2>&1 | grep -E "depth50|B " || exit 1
done
