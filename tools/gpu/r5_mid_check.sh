#!/bin/bash
# whole GPU suite + the two soaks that go through the extraction, on the build with the record path and the shgo replay
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -6 | tee gpurun_out/r05_mid_gputests.txt
{
timeout 200 python tools/extract_soak.py --seconds 60
timeout 200 python tools/pipeline_soak.py --seconds 60
} 2>&1 | tee gpurun_out/r05_mid_soaks.txt
