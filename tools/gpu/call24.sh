cd $GRAFT_REPO_ROOT
python tools/late_iter_cost.py 2>&1 | tail -6
