#!/usr/bin/env python
"""A/B of the small-job tiers: n x n scan matches (shipped chain) in one resident batch under different launcher knobs
(SFE_SW_*), HIP-event time per launch; optionally the per-phase cycle counters of workgroup 0 (PROF build)."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sonar_slam_amd import _lib, icp_config, synth  # noqa: E402
from sonar_slam_amd.pipeline import ScanMatchBatch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_legs import timed  # noqa: E402

ctx = _lib.default_context()
NAMES = ["setup", "first pass", "-", "quantile", "reduce", "solve", "later passes", "tier2", "census", "#rounds", "#long", "#second-pass"]
for n_pts, n_jobs in ((200, 16384), (500, 8192), (1000, 4096)):
    pairs = [synth.scan_pair(seed=9000 + i, n_src=n_pts, n_tgt=n_pts) for i in range(1024)]
    for mode, kw in (("reference", {}), ("p2plane30", dict(minimizer=1, use_diff_checker=0, max_iter=30))):
        p = icp_config.shipped_params(**kw)
        b = ScanMatchBatch(ctx, p, [pairs[j % 1024][0] for j in range(n_jobs)], [pairs[j % 1024][1] for j in range(n_jobs)],
                           [(j, j) for j in range(n_jobs)], [pairs[j % 1024][2] for j in range(n_jobs)])
        ref = None
        for label, env in (("default", {}), ("no tiny kernel", {"SFE_SW_TINY": "0"}), ("1024-thread only", {"SFE_SW_TIERS": "0", "SFE_SW_TINY": "0"}),
                           ("tiny up to 400k pairs", {"SFE_SW_TINY_PAIRS": "400000"}),
                           ):
            os.environ.update(env)
            try:
                ms = timed(ctx, b.run, 3)
                res = b.results()
            finally:
                for k in env:
                    del os.environ[k]
            same = ref is None or (np.array_equal(ref["T"], res["T"]) and np.array_equal(ref["iters"], res["iters"]))
            ref = ref or res
            print("%5d pts x %5d jobs %-9s %-22s %8.3f ms  %9.0f jobs/s  mean iters %.2f  same results %s"
                  % (n_pts, n_jobs, mode, label, ms, n_jobs / ms * 1e3, res["iters"].mean(), same))
        if mode == "reference" and n_pts == 500:
            os.environ["SFE_SW_T0_SRC"] = "0"
            cyc = (ctypes.c_longlong * 96)()
            ctx._check(ctx.lib.sfe_icp_get_profile(ctx.handle, 1, cyc))
            b.run()
            ctx.sync()
            ctx._check(ctx.lib.sfe_icp_get_profile(ctx.handle, 0, cyc))
            del os.environ["SFE_SW_T0_SRC"]
            print("   four-wave workgroup 0 (cycles): " + ", ".join("%s %d" % (NAMES[i], cyc[i]) for i in range(12) if NAMES[i] != "-"))
            print("   search kcycles per iteration: " + " ".join("%d" % (cyc[16 + 2 * i] // 1000) for i in range(6)))
        b.free()
