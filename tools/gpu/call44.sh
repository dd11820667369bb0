#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_cfar.py -q -x -k "os or OS or gate" 2>&1 | tail -3
{
echo "== 16-byte streaming (default)"; python tools/os_gated_ab.py 2>&1 | grep "default" 
echo "== 4-byte streaming (SFE_CFAR_OSG_V4=1)"; SFE_CFAR_OSG_V4=1 python tools/os_gated_ab.py 2>&1 | grep "default" | head -3
} > gpurun_out/os_gated_ab2.txt 2>&1
cat gpurun_out/os_gated_ab2.txt
cat > /tmp/lat1.py <<'PY'
import os, sys, time, numpy as np
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
from sonar_slam_amd import _lib, icp_config, pcl, synth
ctx = _lib.default_context(); icp = pcl.ICP(ctx); icp.setParams(icp_config.shipped_params())
for n in (100, 200, 400):
    s, t, g, _ = synth.scan_pair(seed=40 + n, n_src=n, n_tgt=n)
    icp.compute(s, t, g)
    ts = []
    for _ in range(200):
        t0 = time.perf_counter(); icp.compute(s, t, g); ts.append(time.perf_counter() - t0)
    print(os.environ.get('TAG'), n, "median %.1f us  p10 %.1f" % (1e6*np.median(ts), 1e6*np.percentile(ts,10)), flush=True)
PY
TAG=default python /tmp/lat1.py
TAG=notiny SFE_SW_TINY=0 python /tmp/lat1.py
TAG=notiny_t1 SFE_SW_TINY=0 SFE_SW_T1_MIN_JOBS=1 SFE_SW_T0_MIN_JOBS=100000 python /tmp/lat1.py
TAG=notiny_t0 SFE_SW_TINY=0 SFE_SW_T0_MIN_JOBS=1 python /tmp/lat1.py
