#!/bin/bash
# round 6: A/B of two environments on the in-tree build inside one GPU call: phase cycles of workgroup 0 + launch time at the
# bench's batch size.  usage: gpurun -- 'bash tools/gpu/ab_env.sh rounds "ENV_A" "ENV_B" [pytest file|none]'
rounds=${1:-2}; ea=$2; eb=$3; tests=${4:-none}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
if [ "$tests" != none ]; then timeout 1200 python -m pytest $tests -q -x 2>&1 | tail -3; fi
for r in $(seq 1 $rounds); do
  for v in a b; do
    if [ $v = a ]; then e=$ea; else e=$eb; fi
    echo "== $v: $e (round $r)"
    env $e timeout -s KILL 200 python tools/stage_times.py --batch 4096 --icp-variants 0 --p2plane-only 2>&1 | grep -A4 "^icp" | cut -c1-330
  done
done 2>&1 | tee gpurun_out/r06_ab_env.txt
