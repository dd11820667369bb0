#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for inf in 1 2; do
python bench.py --no-legs --steps 6 --warmup 2 --inflight $inf > gpurun_out/bench_inf$inf.json 2>/dev/null
python -c "import json;d=json.load(open('gpurun_out/bench_inf$inf.json'));print($inf, d['value'],d['ms_per_step'],d['parity_check']['icp_max_pose_diff_vs_f64_sums'], d['parity_check']['frames_bit_exact'])"
done
