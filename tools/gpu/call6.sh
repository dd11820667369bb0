cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_icp.py -q -m gpu --timeout 300 -x 2>&1 | tail -5 > gpurun_out/r03_t_icp6.txt
timeout 600 python tools/small_jobs_ab.py 2>&1 | grep -E "tiers default|1024-thread|workgroup 0|search kcycles" > gpurun_out/r03_small_ab6.txt
(for v in "SFE_SW_UNBOUNDED_COOP=1" "SFE_SW_UNBOUNDED_COOP=0"; do echo "== $v"; env $v python tools/stage_times.py --batch 512 --icp-variants 0 2>&1 | grep -E "^icp|first iteration|per iteration|workgroup 0"; done) > gpurun_out/r03_stage6.txt 2>&1
cat gpurun_out/r03_t_icp6.txt gpurun_out/r03_small_ab6.txt; cut -c1-400 gpurun_out/r03_stage6.txt
