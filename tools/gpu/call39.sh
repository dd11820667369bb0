#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_extract.py -q -x 2>&1 | tail -3
{
for v in "" "SFE_EXPAND_WG=1" "SFE_SG_SLICES=32"; do
  echo "== $v"; env $v python tools/extract_times.py 512 2>&1 | tail -1
done
} > gpurun_out/extract_ab5.txt 2>&1
cat gpurun_out/extract_ab5.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_ext5 -- python $GRAFT_REPO_ROOT/tools/extract_times.py 512 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(ls gpurun_out/prof_ext5/*/*.db | head -1) > gpurun_out/extract_kernels5.txt 2>&1; head -10 gpurun_out/extract_kernels5.txt
rm -rf gpurun_out/prof_ext5
