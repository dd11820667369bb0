#!/bin/bash
# Every counter the bench line quotes, re-collected on the CURRENT build, one rocprofv3 pass per
# counter set, --kernel-trace only (never combined with the hip / hsa / memory-copy trace domains):
#   CFAR bit-stream kernel, $CFAR_FRAMES (4096: the step's shape) frames per launch: kernel summary (isolated launches), FETCH_SIZE, WRITE_SIZE
#   extraction kernels, 256 frames per launch:       FETCH_SIZE, WRITE_SIZE
#   cloud filters, 512 frames per launch:             kernel summary, FETCH_SIZE, WRITE_SIZE, SQ set, LDS set (round 5)
#   ICP loop + prep kernels, 4096 p2plane30 jobs:    SQ set
# -> gpurun_out/<tag>_*.txt (tools/rocpd_summary.py) and the JSON files bench.py reads (tools/make_counter_json.py).
# usage: gpurun --timeout 1500 -- 'bash tools/gpu/counters.sh r04'
tag=${1:-r06}
CF=${CFAR_FRAMES:-4096}   # frames per CFAR launch: the timed step's shape (round 6; rounds 1-5 collected at 1024)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
run() { # name, pmc args..., -- command
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
run cfar_bits_kernels -- python $R/tools/cfar_sweep.py --only --bits --reps 120 --frames $CF   # (363 launches: the device's sustained clocks, profiles/r06_cfar_series.txt)
run cfar_bits_fetch FETCH_SIZE -- python $R/tools/cfar_sweep.py --only --bits --frames $CF
run cfar_bits_write WRITE_SIZE -- python $R/tools/cfar_sweep.py --only --bits --frames $CF
run extract_fetch FETCH_SIZE -- python $R/tools/extract_times.py 256
run extract_write WRITE_SIZE -- python $R/tools/extract_times.py 256
run filters_kernels -- python $R/tools/extract_times.py 512
run filters_fetch FETCH_SIZE -- python $R/tools/extract_times.py 512
run filters_write WRITE_SIZE -- python $R/tools/extract_times.py 512
run filters_sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY -- python $R/tools/extract_times.py 512
run filters_lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS -- python $R/tools/extract_times.py 512
run icp_sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY -- python $R/tools/stage_times.py --batch 4096 --icp-variants 0 --p2plane-only
cd $R
python tools/make_counter_json.py $tag $CF || true
ls -la gpurun_out/${tag}_*.db | head
head -6 gpurun_out/${tag}_cfar_bits_kernels.txt
