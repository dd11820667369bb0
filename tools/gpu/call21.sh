cd $GRAFT_REPO_ROOT
for d in 0 1 2 3; do echo "dbg=$d"; SFE_CF_DBG=$d python tools/extract_times.py 1024 2>&1 | tail -1; done
