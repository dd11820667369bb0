cd $GRAFT_REPO_ROOT
python tools/lat_ab.py 2>&1 | tail -12
