#!/bin/bash
# round 6: the staged extraction -> filter hand-over (float32 pairs + bounding boxes instead of float64 points read back):
# its tests, then the front-end stage times with and without it, alternating inside one GPU call
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_extract.py -q -x 2>&1 | tail -4
for r in 1 2 3; do
  for v in 1 0; do
    echo -n "staged=$v  "
    SONARFE_STAGED=$v timeout -s KILL 200 python tools/stage_times.py --batch 4096 --icp-variants 0 --p2plane-only 2>&1 | grep "^cfar\|^extract\|^filter" | tr '\n' ' '; echo
  done
done 2>&1 | tee gpurun_out/r06_staged_ab.txt
