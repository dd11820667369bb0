cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_icp.py -q -m gpu --timeout 400 -x 2>&1 | tail -5 > gpurun_out/r03_t10.txt
(for v in "SFE_SW_KNN_FAST=1" "SFE_SW_KNN_FAST=0"; do echo "== $v"; env $v python tools/stage_times.py --batch 1024 --icp-variants 0 2>&1 | grep -E "^icp"; done) > gpurun_out/r03_stage10.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
SFE_SW_KNN_FAST=$v timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_b$v -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-legs --no-latency --no-farm --parity-jobs 2 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py /tmp/prof_b$v/b_results.db 2>&1 | head -12 | cut -c1-170 > $GRAFT_REPO_ROOT/gpurun_out/r03_kernels10_fast$v.txt
done
cd $GRAFT_REPO_ROOT
cat gpurun_out/r03_t10.txt gpurun_out/r03_stage10.txt gpurun_out/r03_kernels10_fast1.txt gpurun_out/r03_kernels10_fast0.txt
