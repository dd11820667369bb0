#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_cfar.py -q -x 2>&1 | tail -3
{
echo "== threshold maps: ring kernel for the four ring windows, computed values (default)"; python tools/cfar_variants.py --windows '40,10 32,8 20,4 16,2' 2>&1 | grep "mask+thr\|Ntc" | grep -v " OS "
echo "== SFE_CFAR_NO_RING_THR=1 (sliding-sum kernel with the LDS ring, computed values)"; SFE_CFAR_NO_RING_THR=1 python tools/cfar_variants.py --windows '40,10 32,8 20,4 16,2' 2>&1 | grep "mask+thr" | grep -v " OS "
} > gpurun_out/cfar_thr_ab2.txt 2>&1
cat gpurun_out/cfar_thr_ab2.txt
