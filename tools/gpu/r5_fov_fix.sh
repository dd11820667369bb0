#!/bin/bash
# after the correctly rounded range in store_fov_kernel: the new test, the file's other GPU tests, the default-flow soak on two seeds
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_global_init.py -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -5
{
  for seed in 1 7; do
    timeout 300 python tools/default_flow_soak.py --seconds 100 --seed $seed --dump gpurun_out/soak_mismatch_$seed.npz
  done
} 2>&1 | tee gpurun_out/r05_default_flow_soak.txt
