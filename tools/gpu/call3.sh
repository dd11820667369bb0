cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python tools/small_jobs_ab.py > gpurun_out/r03_small_ab.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_live -o live -- python $GRAFT_REPO_ROOT/tools/live_latency.py > $GRAFT_REPO_ROOT/gpurun_out/r03_live.txt 2>&1
cd $GRAFT_REPO_ROOT
find /tmp/prof_live -name "*kernel_stats*" | head -3
cp $(find /tmp/prof_live -name "*kernel_stats.csv" | head -1) gpurun_out/r03_live_kernel_stats.csv 2>/dev/null
cat gpurun_out/r03_small_ab.txt | tail -40
