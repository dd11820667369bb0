#!/bin/bash
# round 6 (last bullet): bound the "phase split" with a measurement instead of an estimate.
# A kernel A that ran iterations 0..k of a job on a whole CU (one workgroup per CU, 128 VGPRs, 160 KB of LDS) and handed
# over to today's kernel can at best run those iterations as fast as today's kernel does when it has the CU to itself.
# So: today's kernel, forced point-to-plane chain cut to 1 / 3 / 9 iterations, 4096 jobs, two workgroups per CU
# (SFE_SW_WIDE=0: the bench's shape) against one per CU (SFE_SW_WIDE=1).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for it in 1 3 9 30; do
  for wide in 0 1; do
    echo "== max_iter $it, SFE_SW_WIDE=$wide"
    SFE_SW_WIDE=$wide timeout -s KILL 300 python tools/stage_times.py --batch 4096 --icp-variants 0 --p2plane-only --max-iter $it 2>&1 | grep "^icp" | cut -c1-200
  done
done 2>&1 | tee gpurun_out/r06_phase_split_bound.txt
