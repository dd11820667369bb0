#!/bin/bash
# full GPU check of a commit: every -m gpu test, smoke(), the bench line with all legs, and the rocprofv3 kernel summary of
# the same bench command; results under gpurun_out/ (copy what is to be kept into profiles/)
# usage: gpurun --timeout 2400 -- 'bash tools/gpu/full_check.sh <tag>'
tag=${1:-check}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -4 | tee gpurun_out/${tag}_gputests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 1200 python bench.py --steps 5 --warmup 1 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
python - <<PY
import json
d=json.load(open('gpurun_out/${tag}_bench.json'))
print('keyframes/s %.0f  ms/step %.2f  roofline frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))
print(d['stage_ms_per_512_keyframes']); print(d['parity_check']['frames_bit_exact'], d['parity_check']['icp_max_pose_diff_vs_f64_sums'])
print({k: d['live_latency'][k] for k in d['live_latency'] if k.endswith('_us')})
print({k: v for k, v in d['real_size'].items() if not isinstance(v, (dict, list, str))})
print({k: v for k, v in d['configs4_hires'].items() if not isinstance(v, (dict, list, str))})
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -- python $GRAFT_REPO_ROOT/bench.py --no-legs --steps 5 --warmup 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(ls gpurun_out/prof_$tag/*/*.db | head -1) > gpurun_out/${tag}_bench_kernels.txt 2>&1
rm -rf gpurun_out/prof_$tag
head -16 gpurun_out/${tag}_bench_kernels.txt | cut -c1-150
