#!/bin/bash
# the CFAR bit-stream kernel with the short last tile: parity tests, then tile heights at 1024 and 4096 frames per launch
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_cfar.py tests/test_reference_cfar.py -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -3
{
timeout 300 python tools/cfar_sweep.py --bits --variants 0 --tiles 0,62,124,186,248,310,372 --frames 1024 --rounds 5
timeout 300 python tools/cfar_sweep.py --bits --variants 0 --tiles 0,124,186,248 --frames 4096 --rounds 3
timeout 300 python tools/cfar_sweep.py --variants 0 --tiles 0,124,186,248 --frames 1024 --rounds 3
} 2>&1 | tee gpurun_out/r05_cfar_tiles.txt
