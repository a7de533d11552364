#!/bin/bash
# Ad-hoc: the refill kernel under register bounds (library variants) and workgroup sizes on a large scene.
for lib in "$@"; do
  for blk in 256 512; do
    echo "== $lib block $blk"
    PT_LIB=$PWD/$lib PT_PATHS_BLOCK=$blk timeout -k 10 200 python tools/mb/refill.py $SCENE 16 2>&1 | grep -v "amdgpu.ids\|identical" || exit 1
  done
done
