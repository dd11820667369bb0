#!/bin/bash
# kernel summary + FETCH_SIZE / WRITE_SIZE passes of the extraction alone (256 frames per launch) -> gpurun_out/<tag>_extract_*.txt
# and profiles/extract_pmc.json (tools/make_counter_json.py needs the other passes of the same tag to be present as well: run
# tools/gpu/counters.sh for the full set)
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
run extract_kernels -- python $R/tools/extract_times.py 256
run extract_fetch FETCH_SIZE -- python $R/tools/extract_times.py 256
run extract_write WRITE_SIZE -- python $R/tools/extract_times.py 256
cd $R
for B in 512 1024; do for v in 0 2; do EXTRACT_VARIANT=$v timeout 300 python tools/extract_times.py $B; done; done 2>&1 | tee gpurun_out/${tag}_extract_records_ab2.txt
grep -i "extract" gpurun_out/${tag}_extract_kernels.txt | head -12
grep -i "extract" gpurun_out/${tag}_extract_fetch.txt | head -12
grep -i "extract" gpurun_out/${tag}_extract_write.txt | head -12
