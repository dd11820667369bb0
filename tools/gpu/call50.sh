#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
SFE_SW_Q512=1 timeout 900 python -m pytest tests/test_gpu_icp.py -q -x -k "sweep or records or plane" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
SFE_SW_Q512=1 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_b50 -- python $GRAFT_REPO_ROOT/bench.py --no-legs --steps 4 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/bench_b50.json 2>/dev/null
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(ls $GRAFT_REPO_ROOT/gpurun_out/prof_b50/*/*.db | head -1) > $GRAFT_REPO_ROOT/gpurun_out/b50_kernels.txt 2>&1
head -7 $GRAFT_REPO_ROOT/gpurun_out/b50_kernels.txt | cut -c1-160
python -c "import json;d=json.load(open('$GRAFT_REPO_ROOT/gpurun_out/bench_b50.json'));print(d['value'],d['ms_per_step'],d['parity_check']['icp_max_pose_diff_vs_f64_sums'])"
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_b50
