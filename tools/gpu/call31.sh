#!/bin/bash
# downsample fast path: parity + A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_icp.py -q -k "downsample or feature_extraction or replay" -x 2>&1 | tail -5
timeout 300 python -m pytest tests/test_replay.py tests/test_gpu_extract.py -q -m gpu -x 2>&1 | tail -3
cat > /tmp/ds_ab.py <<'PY'
import numpy as np, time, os, sys
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
from sonar_slam_amd import pcl
rng = np.random.default_rng(1)
for n in (500, 3000, 11000, 16384, 30000, 60000):
    pts = rng.uniform(-30, 30, (n, 2)).astype(np.float32)
    pcl.downsample(pts, 0.5)
    t = time.perf_counter()
    for _ in range(50): out = pcl.downsample(pts, 0.5)
    dt = (time.perf_counter() - t) / 50
    print(os.environ.get('SFE_DS_RANK', 'fast'), n, len(out), '%.1f us' % (dt * 1e6))
PY
python /tmp/ds_ab.py > gpurun_out/ds_ab.txt 2>&1
SFE_DS_RANK=1 python /tmp/ds_ab.py >> gpurun_out/ds_ab.txt 2>&1
cat gpurun_out/ds_ab.txt
