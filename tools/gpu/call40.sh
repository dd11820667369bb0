#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python bench.py --steps 5 --warmup 1 > gpurun_out/bench_r03_run40.json 2> gpurun_out/bench_r03_run40.err
tail -c 1500 gpurun_out/bench_r03_run40.json; tail -3 gpurun_out/bench_r03_run40.err
