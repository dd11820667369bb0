cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(for v in "SFE_SW_UNION_ITERS=1" "SFE_SW_UNION_ITERS=0" "SFE_SW_UNION_ITERS=1 SFE_SW_UNION_MAX=768" "SFE_SW_UNION_ITERS=1" "SFE_SW_UNION_ITERS=0"; do echo "== $v"; env $v python tools/stage_times.py --batch 1024 --icp-variants 0 2>&1 | grep -E "^icp|per iteration" | cut -c1-200; done) > gpurun_out/r03_stage18.txt 2>&1
cat gpurun_out/r03_stage18.txt
