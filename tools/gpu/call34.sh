#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out


{
for v in "SFE_SG_PIECE=4" "SFE_SG_SLICES=32 SFE_SG_PIECE=3" "SFE_SG_SLICES=32 SFE_SG_PIECE=2" "SFE_SG_SLICES=64 SFE_SG_PIECE=2" "SFE_SG_SLICES=64 SFE_SG_PIECE=0" "SFE_SG_SLICES=32 SFE_SG_PIECE=4" "SFE_SG_SLICES=24 SFE_SG_PIECE=3"; do
  echo "== $v"; env $v python tools/extract_times.py 512 2>&1 | tail -1
done
} > gpurun_out/extract_ab2.txt 2>&1
cat gpurun_out/extract_ab2.txt
