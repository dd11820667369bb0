cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_icp.py -q -m gpu --timeout 400 -x 2>&1 | tail -3
timeout 400 python tools/icp_soak.py --seconds 150 2>&1 | cut -c1-300 | tail -8
timeout 300 python tools/small_jobs_ab.py 2>&1 | grep -E " default"
