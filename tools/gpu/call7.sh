cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python tools/dbg_unbounded.py > gpurun_out/r03_dbg7.txt 2>&1
cat gpurun_out/r03_dbg7.txt | tail -90
