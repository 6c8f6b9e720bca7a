#!/bin/bash
# Kernel-trace statistics and hardware-counter passes of ONE leg of bench.py, one rocprofv3 run per pass (counters
# are never combined with tracing). Run on the GPU box from the repository root:
#   bash profiles/collect_r03.sh <tag> <headline|streaming> [extra env assignments...]
# Results under gpurun_out/r03_<tag>/ ; summarise with profiles/summarize_r03.py
set -u
TAG=${1:-run}
LEG=${2:-headline}
shift 2
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r03_$TAG
mkdir -p "$OUT"
export RT_BENCH_LEG=$LEG
for kv in "$@"; do export "$kv"; done
if [ "$LEG" = headline ]; then ARGS="--steps 50 --warmup 5 --no-cpu-baseline"; else ARGS="--no-cpu-baseline"; fi
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python3 "$ROOT/bench.py" $ARGS > "$OUT/bench.json" 2> "$OUT/bench.err" || { echo "bench failed"; exit 1; }
echo "bench done"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/trace.log" 2>&1 || { echo "kernel trace failed"; exit 1; }
echo "kernel trace done"
pass() {
  name=$1; shift
  timeout -s KILL 300 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/$name.log" 2>&1
  rc=$?
  echo "pass $name rc=$rc"
  return $rc
}
pass sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY || exit 1
pass sq2 SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS || exit 1
pass tcc_rd FETCH_SIZE TCC_HIT_sum GRBM_GUI_ACTIVE || exit 1
pass tcc_wr WRITE_SIZE TCC_MISS_sum TCC_REQ_sum || exit 1
pass tcp TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum TA_BUSY_avr || exit 1
echo "all passes done"
