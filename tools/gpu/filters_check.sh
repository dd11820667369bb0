#!/bin/bash
# parity + time of the resident cloud filters on the current build (run after touching sfe_cloudfilter.hip)
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_extract.py tests/test_gpu_store.py -m gpu -q --tb=short -p no:cacheprovider -x -k "filter or downsample or cloud or store or ping or session" 2>&1 | tail -4
timeout 300 python tools/pipeline_soak.py --seconds 40 2>&1 | tail -2
for i in 1 2; do python tools/extract_times.py 512; done
python tools/extract_times.py 4096
