#!/bin/bash
# kernel summary (rocprofv3 --kernel-trace) of a command: tools/gpu/prof.sh <name> <command...> -> gpurun_out/<name>.txt
name=$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$name
timeout 900 rocprofv3 --kernel-trace -d /tmp/prof_$name -o $name -- "$@" > /tmp/prof_$name.log 2>&1
db=$(find /tmp/prof_$name -name '*.db' | head -1)
python $R/tools/rocpd_summary.py $db > $R/gpurun_out/$name.txt 2>&1
grep -v "^#" $R/gpurun_out/$name.txt | head -${PROF_LINES:-14} | cut -c1-160
