cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_configs.py -q -m gpu --timeout 400 -k legs 2>&1 | tail -30 > gpurun_out/r03_t_cfg2.txt
timeout 900 python bench.py --steps 5 --warmup 1 > gpurun_out/r03_bench2.json 2> gpurun_out/r03_bench2.err
tail -n 3 gpurun_out/r03_t_cfg2.txt; tail -c 800 gpurun_out/r03_bench2.err; head -c 300 gpurun_out/r03_bench2.json
