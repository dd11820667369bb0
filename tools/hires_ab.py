#!/usr/bin/env python
"""A/B of the large-cloud ICP paths (BASELINE configs[4]): 30 guesses x one 20 000 x 20 000 pair alone, split over G
workgroups per job (SFE_SW_MULTI_G), target window in LDS or not; HIP-event times."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sonar_slam_amd import _lib, icp_config, synth  # noqa: E402
from sonar_slam_amd.pipeline import ScanMatchBatch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_legs import pose_diff, timed  # noqa: E402

ctx = _lib.default_context()
NP, NG = 20000, 30
rng = np.random.default_rng(4)
s, t, g, _ = synth.scan_pair(seed=400, n_src=NP, n_tgt=NP)
b0 = synth.pose_of(g)
gs = [synth.pose_matrix(b0[0] + dx, b0[1] + dy, b0[2] + dt).astype(np.float32) for dx, dy, dt in rng.normal(0, [0.3, 0.3, 0.05], (NG, 3))]
for name, kw in (("p2plane30", dict(minimizer=1, use_diff_checker=0, max_iter=30)), ("p2plane1", dict(minimizer=1, use_diff_checker=0, max_iter=1)),
                 ("reference", {})):
    p = icp_config.shipped_params(**kw)
    for ng in (30,):
        one = ScanMatchBatch(ctx, p, [s], [t], [(0, 0)] * ng, gs[:ng])
        for label, env in (("G=1", {"SFE_SW_MULTI": "0"}), ("G=8 one-WG normals", {"SFE_SW_MULTI_G": "8", "SFE_SW_NORMALS_SPLIT": "0"}),
                           ("G=8", {"SFE_SW_MULTI_G": "8"}), ("G=16", {"SFE_SW_MULTI_G": "16"})):
            os.environ.update(env)
            try:
                ms = timed(ctx, one.run, 3)
            finally:
                for k in env:
                    del os.environ[k]
            print("%-9s %2d guesses %-14s %8.3f ms" % (name, ng, label, ms), flush=True)
        one.free()
