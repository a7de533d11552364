#!/bin/bash
# usage: pmc_ab.sh <workload> "<bench flags>" lib1 lib2 ... : SQ counters of the path kernel's launch for each library build
R=$(cd "$(dirname "$0")/../.." && pwd)
cd /tmp && export TMPDIR=/tmp
WL=$1; FLAGS=$2; shift 2
for lib in "$@"; do
  rm -rf /tmp/pmcab
  PT_LIB=$R/$lib timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pmcab -- python3 $R/bench.py --workload $WL $FLAGS --steps 1 --warmup 1 --no-cpu-baseline > /tmp/pmcab.log 2>&1 || { tail -3 /tmp/pmcab.log; exit 1; }
  python3 - "$lib" <<'PY'
import csv, glob, sys, collections
f = glob.glob('/tmp/pmcab/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r['Kernel_Name'].startswith(('void k_paths<false', 'void k_paths_stream<false', 'void k_paths_ring<false')):
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
print(sys.argv[1], {k: round(sum(v) / len(v) / 1e6, 1) for k, v in sorted(acc.items())})
PY
done
