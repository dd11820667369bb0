cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_icp.py -q -m gpu --timeout 300 -x 2>&1 | tail -5 > gpurun_out/r03_t_icp4.txt
(for v in "" "SFE_SW_T1_MIN_JOBS=100000" "SFE_SW_T0_MIN_JOBS=1"; do echo "== $v"; env $v python tools/live_latency.py 2>&1 | tail -1; done) > gpurun_out/r03_live4.txt 2>&1
(for v in "SFE_SW_GRID_DEFER=1" "SFE_SW_GRID_DEFER=0"; do echo "== $v"; env $v python tools/stage_times.py --batch 512 --icp-variants 0 2>&1 | grep -E "^icp|first iteration|per iteration"; done) > gpurun_out/r03_stage4.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_live -o live -- python $GRAFT_REPO_ROOT/tools/live_latency.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py /tmp/prof_live/live_results.db > gpurun_out/r03_live4_kernels.txt 2>&1
python tools/rocpd_timeline.py /tmp/prof_live/live_results.db 2>&1 | tail -60 > gpurun_out/r03_live4_timeline.txt
cat gpurun_out/r03_t_icp4.txt gpurun_out/r03_live4.txt gpurun_out/r03_stage4.txt
