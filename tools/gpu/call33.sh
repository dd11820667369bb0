#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
for v in 0 1 2 4 8 16 3 7 23 ; do
  echo "== dbg $v"; SFE_SG_DBG=$v python tools/extract_times.py 512 2>&1 | tail -1
done
} > gpurun_out/extract_dbg.txt 2>&1
cat gpurun_out/extract_dbg.txt
