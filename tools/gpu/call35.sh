#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/extract_times.py 512 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_ext2 -- python $GRAFT_REPO_ROOT/tools/extract_times.py 512 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(ls gpurun_out/prof_ext2/*/*.db | head -1) > gpurun_out/extract_kernels2.txt 2>&1; head -16 gpurun_out/extract_kernels2.txt
rm -rf gpurun_out/prof_ext2
