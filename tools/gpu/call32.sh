#!/bin/bash
# list-based extraction kernel: parity, soak, A/B of its knobs
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_extract.py -q -x 2>&1 | tail -5
timeout 120 python tools/extract_soak.py --seconds 45 2>&1 | tail -5
{
for v in "" "SFE_SG_LPP=1" "SFE_SG_LPP=2" "SFE_SG_SLICES=2" "SFE_SG_SLICES=8" "SFE_SG_SLICES=8 SFE_SG_LPP=1" "SFE_SG_SLICES=16 SFE_SG_LPP=2" "SFE_SG_SLICES=1"; do
  echo "== $v"; env $v python tools/extract_times.py 512 2>&1 | tail -1
done
} > gpurun_out/extract_ab.txt 2>&1
cat gpurun_out/extract_ab.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_ext -- python $GRAFT_REPO_ROOT/tools/extract_times.py 512 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py gpurun_out/prof_ext > gpurun_out/extract_kernels.txt 2>&1; head -20 gpurun_out/extract_kernels.txt
