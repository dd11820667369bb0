cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu --timeout 400 2>&1 | tail -4 > gpurun_out/r03_t_all23.txt
timeout 900 python bench.py --steps 5 --warmup 1 > gpurun_out/r03_bench23.json 2> gpurun_out/r03_bench23.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_b -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-legs --no-latency --no-farm > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py /tmp/prof_b/b_results.db 2>&1 | head -30 | cut -c1-180 > gpurun_out/r03_run23_bench_kernels.txt
cat gpurun_out/r03_t_all23.txt; tail -c 300 gpurun_out/r03_bench23.err; head -c 200 gpurun_out/r03_bench23.json; echo; head -12 gpurun_out/r03_run23_bench_kernels.txt
