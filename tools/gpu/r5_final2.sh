#!/bin/bash
# final commit of round 5: soaks (incl. the default-flow soak) and the full check
R=$GRAFT_REPO_ROOT
cd $R
bash tools/gpu/soaks.sh r05 2>&1 | tail -12
bash tools/gpu/full_check.sh r05_final 2>&1 | tail -30
