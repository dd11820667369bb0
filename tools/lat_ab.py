#!/usr/bin/env python
"""Single scan match latency (host wall clock per pcl.ICP.compute call, shipped chain) of the sweep kernels vs the
brute-force kernel (sfe_icp_set_tuning bit 2) on small clouds, and the GPU time of a resident batch of such jobs."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sonar_slam_amd import _lib, icp_config, pcl, synth  # noqa: E402
from sonar_slam_amd.pipeline import ScanMatchBatch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_legs import timed  # noqa: E402

ctx = _lib.default_context()
icp = pcl.ICP(ctx)
icp.setParams(icp_config.shipped_params())
for n in (100, 200, 400, 1000):
    s, t, g, _ = synth.scan_pair(seed=40 + n, n_src=n, n_tgt=n)
    for variant in (0, 4):
        ctx._check(ctx.lib.sfe_icp_set_tuning(ctx.handle, variant))
        icp.compute(s, t, g)
        ts = []
        for _ in range(30):
            t0 = time.perf_counter()
            icp.compute(s, t, g)
            ts.append(time.perf_counter() - t0)
        pairs = [synth.scan_pair(seed=9000 + i, n_src=n, n_tgt=n) for i in range(256)]
        nj = 4096
        b = ScanMatchBatch(ctx, icp.params, [pairs[j % 256][0] for j in range(nj)], [pairs[j % 256][1] for j in range(nj)],
                           [(j, j) for j in range(nj)], [pairs[j % 256][2] for j in range(nj)])
        ms = timed(ctx, b.run, 3)
        b.free()
        print("%5d points  %-12s single call %7.1f us   batch of %d: %7.3f ms (%.0f jobs/s)"
              % (n, "brute force" if variant else "sweep", 1e6 * np.median(ts), nj, ms, nj / ms * 1e3), flush=True)
ctx._check(ctx.lib.sfe_icp_set_tuning(ctx.handle, 0))
