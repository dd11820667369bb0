#!/bin/bash
mkdir -p gpurun_out
{
for n in 1 2 3 1 3; do
  timeout 300 python bench.py --no-legs --no-cpu-baseline --no-farm --no-latency --steps 12 --warmup 3 --inflight $n 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('inflight', d['config']['batches_in_flight'], 'keyframes/s %.0f ms/step %.3f frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['frac']), d.get('stage_ms_per_step'))"
done
} 2>&1 | tee gpurun_out/r05_bench_inflight.txt
