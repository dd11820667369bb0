#!/bin/bash
# A/B two configurations inside ONE gpurun call (box-to-box variance is larger than most kernel changes):
# alternates tools/stage_times.py between environment A and environment B.
# usage (on the GPU box): tools/ab.sh rounds "ENV_A" "ENV_B" [extra stage_times args]
#   e.g. tools/ab.sh 3 "SONARFE_LIB=$PWD/_ab/libsonarfe_a.so" ""      (another build vs the in-tree one)
#        tools/ab.sh 3 "SFE_SW_JUMP=0" "SFE_SW_JUMP=1"
n=${1:-3}; ea=$2; eb=$3; shift 3
for r in $(seq 1 "$n"); do
  for v in a b; do
    if [ "$v" = a ]; then e=$ea; else e=$eb; fi
    env $e timeout -s KILL 90 python tools/stage_times.py --batch 512 --icp-variants 0 "$@" 2>&1 | grep "^icp" | sed "s/^/$v /"
  done
done
