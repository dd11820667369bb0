#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_extract.py tests/test_gpu_configs.py -q -x 2>&1 | tail -3
timeout 120 python tools/extract_soak.py --seconds 20 2>&1 | tail -2
timeout 200 python tools/pipeline_soak.py --seconds 40 2>&1 | tail -2
{
for v in "" "SFE_NO_SELF_CLEAN=1" "SFE_EXPAND_WG=4" "SFE_EXPAND_WG=16"; do
  echo "== $v"; env $v python tools/extract_times.py 512 2>&1 | tail -1
done
} > gpurun_out/extract_ab4.txt 2>&1
cat gpurun_out/extract_ab4.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_ext4 -- python $GRAFT_REPO_ROOT/tools/extract_times.py 512 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(ls gpurun_out/prof_ext4/*/*.db | head -1) > gpurun_out/extract_kernels4.txt 2>&1; head -14 gpurun_out/extract_kernels4.txt
rm -rf gpurun_out/prof_ext4
