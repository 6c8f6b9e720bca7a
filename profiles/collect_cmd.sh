#!/bin/bash
# Kernel-trace statistics and the two SQ counter passes (instruction mix, lanes active, wait shares) plus the memory
# passes for any python command, one rocprofv3 run per pass. From the repository root on the GPU box:
#   bash profiles/collect_cmd.sh <tag> <script> [args...]
# Results under gpurun_out/r02_<tag>/ ; summarise with profiles/summarize_r02.py gpurun_out/r02_<tag>
set -u
TAG=$1; shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r02_$TAG
mkdir -p "$OUT"
SCRIPT=$ROOT/$1; shift
ARGS=("$@")
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$SCRIPT" "$@" > "$OUT/trace.log" 2>&1 || { echo "kernel trace failed"; exit 1; }
pass() {
  name=$1; shift
  timeout -s KILL 300 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$SCRIPT" "${ARGS[@]}" > "$OUT/$name.log" 2>&1
  rc=$?; echo "pass $name rc=$rc"; return $rc
}
pass sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY || exit 1
pass sq2 SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS || exit 1
pass tcc_rd FETCH_SIZE TCC_HIT_sum GRBM_GUI_ACTIVE || exit 1
pass tcc_wr WRITE_SIZE TCC_MISS_sum TCC_REQ_sum || exit 1
pass tcp TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum TA_BUSY_avr || exit 1
echo "all passes done"
