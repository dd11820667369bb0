cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_icp.py tests/test_gpu_configs.py -q -m gpu --timeout 400 -x -k "hires or split or mixed or config4 or large_target" 2>&1 | tail -5 > gpurun_out/r03_t9.txt
timeout 600 python tools/hires_ab.py > gpurun_out/r03_hires9.txt 2>&1
cat gpurun_out/r03_t9.txt gpurun_out/r03_hires9.txt
