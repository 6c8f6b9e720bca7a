#!/bin/bash
# Lanes active and instruction counts of a development variant of the trace kernel on one scene (two SQ passes).
# From the repository root on the GPU box: bash profiles/collect_variant_lanes.sh <variant> <scene> [frames]
set -u
VAR=$1; SCENE=$2; FRAMES=${3:-3}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/lanes_v${VAR}_${SCENE%%:*}
mkdir -p "$OUT"
export RT_LIB_PATH=$ROOT/ray-tracer_amd/librt_mi355_exp.so RT_VARIANT=$VAR
cd /tmp && export TMPDIR=/tmp
pass() {
  name=$1; shift
  timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$ROOT/tools/frame_loop.py" "$SCENE" "$FRAMES" > "$OUT/$name.log" 2>&1
  rc=$?; echo "pass $name rc=$rc"; return $rc
}
pass sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY || exit 1
pass sq2 SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS || exit 1
echo "done"
