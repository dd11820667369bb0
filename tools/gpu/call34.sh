#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_extract.py -q -x 2>&1 | tail -3
timeout 120 python tools/extract_soak.py --seconds 30 2>&1 | tail -3
{
for v in "" "SFE_SG_SLICES=2" "SFE_SG_SLICES=8" "SFE_SG_SLICES=16" "SFE_SG_SLICES=8 SFE_SG_PIECE=2" "SFE_SG_SLICES=4 SFE_SG_PIECE=2"; do
  echo "== $v"; env $v python tools/extract_times.py 512 2>&1 | tail -1
done
} > gpurun_out/extract_ab2.txt 2>&1
cat gpurun_out/extract_ab2.txt
