#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_icp.py -q -x -k "tiny or small or tiers or rank" 2>&1 | tail -3
timeout 200 python tools/icp_soak.py --seconds 40 2>&1 | tail -2
timeout 300 python tools/small_jobs_ab.py 2>&1 | grep -v "^ *[a-z#-]* *[0-9]* *$" | grep "pts x" > gpurun_out/small_jobs_pk.txt
cat gpurun_out/small_jobs_pk.txt
