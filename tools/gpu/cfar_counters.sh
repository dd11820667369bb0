#!/bin/bash
# the CFAR passes of tools/gpu/counters.sh alone (kernel summary, FETCH_SIZE, WRITE_SIZE of the bit-stream kernel, 1024 frames per
# launch) -> gpurun_out/<tag>_cfar_bits_*.{txt,db}; then `python tools/make_counter_json.py <tag>` locally
tag=${1:-r05}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
run() {
  local name=$1; shift
  local pmc=()
  while [ "$1" != "--" ]; do pmc+=("$1"); shift; done
  shift
  rm -rf /tmp/prof_$name
  if [ ${#pmc[@]} -gt 0 ]; then
    timeout 600 rocprofv3 --kernel-trace --pmc "${pmc[@]}" -d /tmp/prof_$name -o $name -- "$@" > /tmp/prof_$name.log 2>&1
  else
    timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_$name -o $name -- "$@" > /tmp/prof_$name.log 2>&1
  fi
  echo "== $name rc=$?"
  db=$(find /tmp/prof_$name -name '*.db' | head -1)
  if [ -n "$db" ]; then
    python $R/tools/rocpd_summary.py $db > $R/gpurun_out/${tag}_$name.txt 2>&1
    cp $db $R/gpurun_out/${tag}_$name.db
  else
    tail -5 /tmp/prof_$name.log
  fi
}
run cfar_bits_kernels -- python $R/tools/cfar_sweep.py --only --bits --reps 120
run cfar_bits_fetch FETCH_SIZE -- python $R/tools/cfar_sweep.py --only --bits
run cfar_bits_write WRITE_SIZE -- python $R/tools/cfar_sweep.py --only --bits
head -6 $R/gpurun_out/${tag}_cfar_bits_kernels.txt
