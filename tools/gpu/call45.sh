#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_cfar.py -q -x 2>&1 | tail -3
{
echo "== computed threshold maps (default)"; python tools/cfar_variants.py --windows '40,10 32,8 20,4 16,2 24,6 60,16 80,20' 2>&1 | grep -v " OS " 
echo "== table gathers (SFE_CFAR_THR_TABLE=1)"; SFE_CFAR_THR_TABLE=1 python tools/cfar_variants.py --windows '40,10 32,8 20,4 16,2 24,6 60,16 80,20' 2>&1 | grep "mask+thr" | grep -v " OS "
} > gpurun_out/cfar_thr_ab.txt 2>&1
cat gpurun_out/cfar_thr_ab.txt
