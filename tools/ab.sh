#!/bin/bash
# A/B two builds of the library inside ONE gpurun call (box-to-box variance is larger than most kernel
# changes): alternates `stage_times.py` between _ab/libsonarfe_a.so and the in-tree build.
# usage (on the GPU box): tools/ab.sh [rounds]
n=${1:-3}
for r in $(seq 1 "$n"); do
  for v in a b; do
    if [ "$v" = a ]; then export SONARFE_LIB=$PWD/_ab/libsonarfe_a.so; else unset SONARFE_LIB; fi
    timeout -s KILL 90 python tools/stage_times.py --batch 512 --icp-variants 0 2>&1 | grep "^icp" | sed "s/^/$v /"
  done
done
