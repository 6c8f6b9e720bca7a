#!/bin/bash
# Development aid: the two randomised GPU parity tests with other seeds and more trials. Run on the GPU box.
set -e
for seed in 1 2 3; do
  echo "seed $seed"
  RT_FUZZ_SEED=$seed RT_FUZZ_TRIALS=${RT_FUZZ_TRIALS:-120} timeout -k 10 900 python -u -m pytest tests/test_gpu_fuzz.py tests/test_gpu_textures.py::test_random_textured_scenes -m gpu -x -q -s --timeout 800 --timeout-method thread 2>&1 | grep -E "random|passed|failed|Error|assert" | head -20
done
