#!/bin/bash
# Kernel-trace statistics and counter passes of the configuration legs of bench.py (tools/config_legs.py), one rocprofv3
# run per pass and leg (counters are never combined with tracing). From the repository root on the GPU box:
#   bash profiles/collect_r03_configs.sh [leg ...]
# Results under gpurun_out/r03_cfg_<leg>/ ; condensed by profiles/summarize_r03_configs.py
set -u
ROOT=$(pwd)
LEGS=${@:-"massive4k_100000_skip massive4k_100000_all massive4k_1000000_skip massive4k_1000000_all massive8k_1000000_skip"}
cd /tmp && export TMPDIR=/tmp
for LEG in $LEGS; do
  OUT=$ROOT/gpurun_out/r03_cfg_$LEG
  mkdir -p "$OUT"
  case $LEG in *1000000_all) REPS=3;; *) REPS=20;; esac
  timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/tools/config_legs.py" $LEG $REPS > "$OUT/trace.log" 2>&1 || { echo "$LEG: kernel trace failed"; exit 1; }
  pass() {
    name=$1; shift
    timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$ROOT/tools/config_legs.py" $LEG $REPS > "$OUT/$name.log" 2>&1
    rc=$?; echo "$LEG pass $name rc=$rc"; return $rc
  }
  pass sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD || exit 1
  pass tcc_rd FETCH_SIZE GRBM_GUI_ACTIVE || exit 1
  pass tcc_wr WRITE_SIZE || exit 1
done
echo "all legs done"
