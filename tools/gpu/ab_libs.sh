#!/bin/bash
# round 6: ICP loop kernel A/B inside one GPU call -- the in-tree build against every _ab/libsonarfe_*.so, phase cycles of
# workgroup 0 (PROF build) and the launch time at the bench's batch size
# usage: gpurun --timeout 1500 -- 'bash tools/gpu/ab_libs.sh [rounds] [pytest-file-or-none]'
rounds=${1:-2}; tests=${2:-tests/test_gpu_icp.py}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
if [ "$tests" != none ]; then timeout 1200 python -m pytest $tests -q -x 2>&1 | tail -3; fi
for r in $(seq 1 $rounds); do
  for lib in intree $(ls _ab/libsonarfe_*.so 2>/dev/null); do
    if [ $lib = intree ]; then e=""; else e="SONARFE_LIB=$PWD/$lib"; fi
    echo "== $lib (round $r)"
    env $e timeout -s KILL 200 python tools/stage_times.py --batch 4096 --icp-variants 0 --p2plane-only 2>&1 | grep -A1 "^icp" | cut -c1-260
  done
done 2>&1 | tee gpurun_out/r06_ab_icp.txt
