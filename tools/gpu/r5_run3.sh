#!/bin/bash
# kernel times + SQ counters of the filter stage on the current build; pruned extraction: its tests
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
run() { # name, pmc args..., -- command
  local name=$1; shift
  local pmc=()
  while [ "$1" != "--" ]; do pmc+=("$1"); shift; done
  shift
  rm -rf /tmp/prof_$name
  if [ ${#pmc[@]} -gt 0 ]; then
    timeout 300 rocprofv3 --kernel-trace --pmc "${pmc[@]}" -d /tmp/prof_$name -o $name -- "$@" > /tmp/prof_$name.log 2>&1
  else
    timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_$name -o $name -- "$@" > /tmp/prof_$name.log 2>&1
  fi
  echo "== $name rc=$?"
  db=$(find /tmp/prof_$name -name '*.db' | head -1)
  if [ -n "$db" ]; then
    python $R/tools/rocpd_summary.py $db > $R/gpurun_out/r05b_$name.txt 2>&1
    cp $db $R/gpurun_out/r05b_$name.db
  else
    tail -5 /tmp/prof_$name.log
  fi
}
mkdir -p $R/gpurun_out
run filters_kernels -- python $R/tools/extract_times.py 512
run filters_sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY -- python $R/tools/extract_times.py 512
run filters_lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS -- python $R/tools/extract_times.py 512
cd $R
grep -i "cf_" gpurun_out/r05b_filters_kernels.txt | cut -c1-150 | head; grep -i "radix" gpurun_out/r05b_filters_sq.txt gpurun_out/r05b_filters_lds.txt | cut -c1-150
timeout 900 python -m pytest tests/test_gpu_extract.py tests/test_gpu_cfar.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -5
timeout 200 python tools/extract_soak.py --seconds 20 2>&1 | tail -2
