cd $GRAFT_REPO_ROOT
timeout 300 python tools/pipeline_soak.py 2>&1 | tail -4
timeout 200 python tools/extract_soak.py 2>&1 | tail -2
