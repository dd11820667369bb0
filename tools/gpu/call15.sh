cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python bench.py --steps 5 --warmup 1 > gpurun_out/r03_bench15.json 2> gpurun_out/r03_bench15.err
tail -c 400 gpurun_out/r03_bench15.err; head -c 250 gpurun_out/r03_bench15.json
