#!/bin/bash
# Collects what profiles/ holds for a kernel state: the bench line, the rocprofv3 kernel-trace statistics of the same
# command, and the hardware-counter passes. Run on the GPU box from the repository root: bash tools/final_profiles.sh <tag>
set -u
TAG=${1:-run}
ROOT=$(pwd)
mkdir -p gpurun_out
timeout -k 10 300 python3 bench.py --steps 200 --warmup 20 > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err || exit 1
echo "bench done"
( cd /tmp && export TMPDIR=/tmp && timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/prof_$TAG" -- python3 "$ROOT/bench.py" --steps 200 --warmup 20 --no-cpu-baseline > "$ROOT/gpurun_out/prof_$TAG.log" 2>&1 ) || exit 1
echo "kernel trace done"
bash profiles/collect_pmc.sh $TAG || exit 1
python3 profiles/summarize_pmc.py gpurun_out/pmc_$TAG "k_trace_primary<false, true, false, false" > gpurun_out/pmc_$TAG.json || exit 1
echo "pmc done"
