#!/bin/bash
# Runs on the GPU box: one rocprofv3 pass per counter group (gfx950 cannot mix FETCH_SIZE and WRITE_SIZE in
# one pass), each over one short bench run, CSV output under gpurun_out/pmc/<group>/.  Usage:
#   bash profiles/collect_pmc.sh [workload] ["extra bench.py flags"]
set -u
WL=${1:-c2}
EXTRA=${2:-}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/pmc
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() { # name, counters...
  local name=$1; shift
  timeout -s KILL 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$name" -- \
      python3 "$REPO/bench.py" --workload "$WL" $EXTRA --steps 1 --warmup 1 --no-cpu-baseline > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$OUT/$name.log"; return 1; }
  echo "pass $name ok"
}
run fetch FETCH_SIZE TCC_HIT_sum && \
run write WRITE_SIZE TCC_MISS_sum && \
run sq SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE && \
run sq2 SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS && \
run mix SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT && \
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
