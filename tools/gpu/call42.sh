#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_icp.py -q -x -k "normals or plane or sweep or witness or grid or tiers" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
for g in 24 0 16 32; do
SFE_SW_KNN_CAP=$g rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_prep$g -- python $GRAFT_REPO_ROOT/bench.py --no-legs --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/bench_prep$g.json 2>/dev/null
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(ls $GRAFT_REPO_ROOT/gpurun_out/prof_prep$g/*/*.db | head -1) > $GRAFT_REPO_ROOT/gpurun_out/prep_kernels$g.txt 2>&1
echo "== KNN_CAP=$g"; grep -E "prep_kernel<1024" $GRAFT_REPO_ROOT/gpurun_out/prep_kernels$g.txt | cut -c1-150
python -c "import json;d=json.load(open('$GRAFT_REPO_ROOT/gpurun_out/bench_prep$g.json'));print(d['value'],d['ms_per_step'],d['parity_check']['icp_max_pose_diff_vs_f64_sums'])"
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_prep$g
done
