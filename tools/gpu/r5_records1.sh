#!/bin/bash
# the record path of the extraction (no canvas bitmap): parity tests, then A/B of the stage against the canvas path
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_extract.py -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -8
{
for B in 512 4096; do
  for v in 0 2 0 2; do EXTRACT_VARIANT=$v timeout 300 python tools/extract_times.py $B; done
done
} 2>&1 | tee gpurun_out/r05_extract_records_ab.txt
