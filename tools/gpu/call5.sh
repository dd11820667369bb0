cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python tools/small_jobs_ab.py > gpurun_out/r03_small_ab5.txt 2>&1
python tools/live_latency.py 2>&1 | tail -1 > gpurun_out/r03_live5.txt
cat gpurun_out/r03_small_ab5.txt gpurun_out/r03_live5.txt
