#!/bin/bash
# end of round 5: counters, per-phase ICP table, soaks, then the full check (all GPU tests, smoke, bench with legs, kernel summary)
R=$GRAFT_REPO_ROOT
cd $R
bash tools/gpu/counters.sh r05 2>&1 | tail -12
python tools/stage_times.py --batch 4096 --icp-variants 0 > gpurun_out/r05_stage_times.txt 2>&1; tail -12 gpurun_out/r05_stage_times.txt | cut -c1-400
bash tools/gpu/soaks.sh r05 2>&1 | tail -12
bash tools/gpu/full_check.sh r05_final 2>&1 | tail -30
