cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_icp.py -q -m gpu --timeout 400 -x 2>&1 | tail -4 > gpurun_out/r03_t17.txt
(for v in "SFE_SW_HOP=1" "SFE_SW_HOP=0"; do echo "== $v"; env $v python tools/stage_times.py --batch 1024 --icp-variants 0 2>&1 | grep -E "^icp|per iteration|whole launch" | cut -c1-420; done) > gpurun_out/r03_stage17.txt 2>&1
(for v in "SFE_SW_HOP=1" "SFE_SW_HOP=0"; do echo "== $v"; env $v SFE_SW_TINY=0 python tools/small_jobs_ab.py 2>&1 | grep -E " default"; done) > gpurun_out/r03_small17.txt 2>&1
cat gpurun_out/r03_t17.txt gpurun_out/r03_stage17.txt gpurun_out/r03_small17.txt
