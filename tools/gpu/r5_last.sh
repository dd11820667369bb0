#!/bin/bash
# the last GPU call of the round: every -m gpu test and the bench line on the final commit (the soaks and the rocprofv3 summary of
# tools/gpu/r5_final2.sh were taken on the same kernels earlier: profiles/r05_soaks.txt, r05_final_bench_kernels.txt)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 200 python bench.py --steps 5 --warmup 1 > gpurun_out/r05_final_bench.json 2> gpurun_out/r05_final_bench.err
python -c "
import json; d=json.load(open('gpurun_out/r05_final_bench.json')); print('keyframes/s %.0f ms/step %.2f frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['frac']), d['config'].get('chained_with_initialization_keyframes_per_s'), d['config'].get('loop_closure_ms_per_search'))"
timeout 200 python -m pytest tests -q -m gpu 2>&1 | tail -3 | tee gpurun_out/r05_final_gputests.txt
