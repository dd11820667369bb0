#!/bin/bash
# round 5, second GPU call: store-full fix, the chunked radix ranking of the resident downsample (parity + time)
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_store.py tests/test_gpu_extract.py tests/test_gpu_configs.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -40 > gpurun_out/r05_run2_tests.txt
tail -4 gpurun_out/r05_run2_tests.txt
timeout 300 python tools/pipeline_soak.py 2>&1 | tail -3
for i in 1 2; do python tools/extract_times.py 512; done
python tools/extract_times.py 4096
bash tools/gpu/prof.sh r05_run2_filters python tools/extract_times.py 512
