cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_icp.py tests/test_gpu_configs.py -q -m gpu --timeout 400 2>&1 | tail -3
timeout 600 python tools/oracle_soak.py --big 1500 --small 4000 --seed 8484 > gpurun_out/r03_oracle_soak_seed8484.json 2>/dev/null; cat gpurun_out/r03_oracle_soak_seed8484.json
timeout 200 python tools/icp_soak.py --seconds 30 2>&1 | tail -1
