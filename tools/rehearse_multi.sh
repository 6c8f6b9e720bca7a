#!/bin/bash
# Development aid: bench.py's N > 1 path on ONE GPU (every rank on GPU 0, gloo instead of RCCL): checks that the
# sharded path runs end to end; the rates mean nothing (gloo copies through the host).
set -e
# the same route on the real backend, as far as one GPU allows: a one-rank RCCL communicator, groups of 32 frames
RT_BENCH_RCCL_WORLD1=1 timeout -k 10 200 python bench.py --steps 256 --warmup 64 2>&1 | grep -E '^\{|Error|error|Traceback' | cut -c1-260
export RT_BENCH_ONE_GPU=1 RT_BENCH_GROUP=${RT_BENCH_GROUP:-8}
for n in 2 8; do
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29510 + n)) bench.py --gpus $n --steps 40 --warmup 5 --no-cpu-baseline 2>&1 | grep -E '^\{|Error|error|Traceback' | cut -c1-260
done
