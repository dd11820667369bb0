cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_icp.py -q -m gpu --timeout 400 -x -k "rank_deficient" 2>&1 | tail -3
