#!/bin/bash
mkdir -p gpurun_out
{
for i in 1 2; do
  echo "-- short last tile"; timeout 200 python tools/cfar_sweep.py --only --bits --frames 1024
  echo "-- full last tile (rounds 1-4)"; SFE_CFAR_FULL_LAST_TILE=1 timeout 200 python tools/cfar_sweep.py --only --bits --frames 1024
done
echo "-- 4096 frames: short / full"; timeout 200 python tools/cfar_sweep.py --only --bits --frames 4096; SFE_CFAR_FULL_LAST_TILE=1 timeout 200 python tools/cfar_sweep.py --only --bits --frames 4096
} 2>&1 | tee gpurun_out/r05_cfar_last_tile_ab.txt
