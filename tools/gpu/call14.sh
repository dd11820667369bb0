cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_icp.py -q -m gpu --timeout 400 -x 2>&1 | tail -5 > gpurun_out/r03_t14.txt
timeout 600 python tools/small_jobs_ab.py 2>&1 | grep -v "workgroup 0\|search kcycles" > gpurun_out/r03_small14.txt
python tools/live_latency.py 2>&1 | tail -1 > gpurun_out/r03_live14.txt
cat gpurun_out/r03_t14.txt gpurun_out/r03_small14.txt gpurun_out/r03_live14.txt
