#!/bin/bash
# Extra counter passes of the headline leg (instruction cache, instruction fetch, latency levels, LDS), one rocprofv3
# run per pass: bash profiles/collect_extra.sh <tag>   -> gpurun_out/r03_<tag>/
set -u
TAG=${1:-extra}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r03_$TAG
mkdir -p "$OUT"
export RT_BENCH_LEG=headline
ARGS="--steps 50 --warmup 5 --no-cpu-baseline"
cd /tmp && export TMPDIR=/tmp
pass() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/$name.log" 2>&1
  rc=$?; echo "pass $name rc=$rc"; return $rc
}
pass ic SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES || exit 1
pass lvl SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE || exit 1
pass dc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_INST_REQ SQC_TC_DATA_READ_REQ SQC_TC_STALL SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM || exit 1
echo done
