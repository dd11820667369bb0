#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_extract.py -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -3
{ timeout 300 python tools/extract_records_stats.py 1024 256; timeout 300 python tools/extract_records_stats.py 512 32; } 2>&1 | tee gpurun_out/r05_extract_records_stats.txt
for v in 0 2; do EXTRACT_VARIANT=$v timeout 300 python tools/extract_times.py 4096; done 2>&1 | tee -a gpurun_out/r05_extract_records_ab.txt
timeout 300 python bench.py --no-legs --no-cpu-baseline --no-farm --no-latency --steps 8 --warmup 2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('keyframes/s %.0f ms/step %.3f frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['frac']), d.get('stage_ms_per_step'), d['roofline_extract']['frac'])"
