#!/usr/bin/env python
"""Per-phase cycles of ONE late (fully recorded) iteration of the 5000 x 5000 point-to-plane chain: profile build with
max_iter = 30 and 24, difference / 6 (workgroup 0; 512 jobs per launch)."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sonar_slam_amd import _lib, icp_config, synth  # noqa: E402
from sonar_slam_amd.pipeline import ScanMatchBatch  # noqa: E402

ctx = _lib.default_context()
NAMES = ["setup", "first pass", "-", "quantile", "reduce", "solve", "later passes", "tier2", "census"]
pairs = [synth.scan_pair(seed=100000 * 0 + j, n_src=5000, n_tgt=5000) for j in range(512)]
prof = {}
for mi in (30, 24, 12):
    p = icp_config.shipped_params(minimizer=1, use_diff_checker=0, max_iter=mi)
    b = ScanMatchBatch(ctx, p, [q[0] for q in pairs], [q[1] for q in pairs], [(j, j) for j in range(512)], [q[2] for q in pairs])
    b.run(); ctx.sync()
    cyc = (ctypes.c_longlong * 96)()
    ctx._check(ctx.lib.sfe_icp_get_profile(ctx.handle, 1, cyc))
    b.run(); ctx.sync()
    ctx._check(ctx.lib.sfe_icp_get_profile(ctx.handle, 0, cyc))
    prof[mi] = [int(cyc[i]) for i in range(9)]
    b.free()
    print(mi, {NAMES[i]: prof[mi][i] for i in range(9) if NAMES[i] != "-"}, flush=True)
print("one iteration of 25..30:", {NAMES[i]: (prof[30][i] - prof[24][i]) // 6 for i in range(9) if NAMES[i] != "-"})
print("one iteration of 13..24:", {NAMES[i]: (prof[24][i] - prof[12][i]) // 12 for i in range(9) if NAMES[i] != "-"})
