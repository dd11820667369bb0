cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_cfar.py -q -m gpu --timeout 400 -x 2>&1 | tail -5
timeout 300 python tools/os_gated_ab.py 2>&1 | tee gpurun_out/r03_os_gated_ab.txt | tail -20
