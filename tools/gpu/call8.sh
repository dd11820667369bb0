cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu --timeout 400 2>&1 | tail -8 > gpurun_out/r03_t_all8.txt
timeout 900 python bench.py --steps 5 --warmup 1 > gpurun_out/r03_bench8.json 2> gpurun_out/r03_bench8.err
cat gpurun_out/r03_t_all8.txt; tail -c 500 gpurun_out/r03_bench8.err; head -c 200 gpurun_out/r03_bench8.json
