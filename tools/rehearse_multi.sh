set -e
timeout -k 10 600 python -u -m pytest tests/test_gpu_render.py -m gpu -x -q --timeout 240 --timeout-method thread 2>&1 | tail -5
export RT_BENCH_ONE_GPU=1
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 100 --warmup 10 --no-cpu-baseline
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 4 --steps 203 --warmup 10 --no-cpu-baseline
