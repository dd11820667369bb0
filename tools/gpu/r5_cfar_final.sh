#!/bin/bash
# CFAR counters at the device's sustained clocks, the launch series, and a short bench run to see the roofline leg
bash tools/gpu/cfar_counters.sh r05 2>&1 | tail -6
cd $GRAFT_REPO_ROOT
timeout 200 python tools/cfar_series.py 1024 400 > gpurun_out/r05_cfar_series.txt 2>&1; tail -4 gpurun_out/r05_cfar_series.txt
timeout 300 python bench.py --no-legs --no-cpu-baseline --no-farm --no-latency --steps 5 --warmup 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('keyframes/s %.0f ms/step %.3f' % (d['value'], d['ms_per_step']), {k: r[k] for k in ('frac','ms_per_launch','warmup_launches','ms_per_launch_first_launches','frac_first_launches','frac_at_step_launch_shape','frac_byte_mask')})"
