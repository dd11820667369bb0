cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_icp.py -q -m gpu --timeout 400 -x 2>&1 | tail -3
(for v in "SFE_SW_LEAN_TRIAGE=1" "SFE_SW_LEAN_TRIAGE=0" "SFE_SW_LEAN_TRIAGE=1" "SFE_SW_LEAN_TRIAGE=0"; do echo "== $v"; env $v python tools/stage_times.py --batch 1024 --icp-variants 0 2>&1 | grep -E "^icp p2plane"; done)
SFE_SW_LEAN_TRIAGE=1 python tools/late_iter_cost.py 2>&1 | tail -2
