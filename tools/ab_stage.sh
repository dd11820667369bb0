#!/bin/bash
# like ab.sh, but prints the front-end stages (cfar / extract / filter) of tools/stage_times.py
n=${1:-3}; ea=$2; eb=$3; shift 3
for r in $(seq 1 "$n"); do
  for v in a b; do
    if [ "$v" = a ]; then e=$ea; else e=$eb; fi
    env $e timeout -s KILL 90 python tools/stage_times.py --batch 512 --icp-variants 0 "$@" 2>&1 | grep "^cfar\|^extract\|^filter" | tr '\n' ' ' | sed "s/^/$v /"; echo
  done
done
