#!/bin/bash
# round 5, first GPU call: the new default-flow tests, the store tests, then the filter stage's counters and a short bench
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_global_init.py tests/test_gpu_store.py tests/test_matching_cost.py tests/test_replay.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -150 > gpurun_out/r05_run1_tests.txt
tail -5 gpurun_out/r05_run1_tests.txt
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
    python $R/tools/rocpd_summary.py $db > $R/gpurun_out/r05_$name.txt 2>&1
    cp $db $R/gpurun_out/r05_$name.db
  else
    tail -5 /tmp/prof_$name.log
  fi
}
run filters_kernels -- python $R/tools/extract_times.py 512
run filters_fetch FETCH_SIZE -- python $R/tools/extract_times.py 512
run filters_write WRITE_SIZE -- python $R/tools/extract_times.py 512
run filters_sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY -- python $R/tools/extract_times.py 512
cd $R
grep -i "cf_\|radius\|downsample" gpurun_out/r05_filters_kernels.txt | cut -c1-180 | head -12
timeout 600 python bench.py --no-legs --steps 5 --warmup 1 > gpurun_out/r05_run1_bench.json 2> gpurun_out/r05_run1_bench.err
python -c "
import json; d=json.load(open('gpurun_out/r05_run1_bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('stage_ms_per_step'))"
