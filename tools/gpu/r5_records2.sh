#!/bin/bash
mkdir -p gpurun_out
{ timeout 300 python tools/extract_records_stats.py 1024 256; timeout 300 python tools/extract_records_stats.py 512 32; } 2>&1 | tee gpurun_out/r05_extract_records_stats.txt
