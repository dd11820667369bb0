#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_icp.py -q -x 2>&1 | tail -3
timeout 200 python tools/icp_soak.py --seconds 60 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_b43 -- python $GRAFT_REPO_ROOT/bench.py --no-legs --steps 5 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/bench_b43.json 2>/dev/null
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(ls $GRAFT_REPO_ROOT/gpurun_out/prof_b43/*/*.db | head -1) > $GRAFT_REPO_ROOT/gpurun_out/b43_kernels.txt 2>&1
head -14 $GRAFT_REPO_ROOT/gpurun_out/b43_kernels.txt | cut -c1-160
python -c "import json;d=json.load(open('$GRAFT_REPO_ROOT/gpurun_out/bench_b43.json'));print(d['value'],d['ms_per_step'],d['parity_check'])"
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_b43
