#!/usr/bin/env python
"""Sweep the search budgets of the ICP loop kernel (env SFE_SW_BUDGET_A / SFE_SW_BUDGET / SFE_SW_RTRIPS) on the bench
scan pairs: `python tools/icp_knobs.py 6,128,4 6,256,4 ...` (first pass trips, second pass budget, trips per round)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from sonar_slam_amd import _lib, icp_config  # noqa: E402
from sonar_slam_amd.CFAR import CFAR  # noqa: E402
from sonar_slam_amd.feature_extraction import FeatureExtraction, SonarPing, oculus_bearings  # noqa: E402
from sonar_slam_amd.pipeline import KeyframeBatch  # noqa: E402

B = 512
ctx = _lib.default_context()
det = CFAR(40, 10, 0.1, 10)
fe = FeatureExtraction(ctx)
fe.Ntc, fe.Ngc, fe.Pfa, fe.rank, fe.alg, fe.threshold = 40, 10, 0.1, 10, "SOCA", 65
fe.configure()
frames, srcs, tgts, guesses = bench.make_inputs(0, B)
fe.generate_map_xy(SonarPing(frames[0], oculus_bearings(bench.COLS), 30.0 / bench.ROWS))


def timed(fn, reps):
    fn()
    ctx.sync()
    ctx.timer_start()
    for _ in range(reps):
        fn()
    return ctx.timer_stop() / reps


kbs = {}
for mode, p in (("p2plane30", icp_config.shipped_params(minimizer=1, use_diff_checker=0, max_iter=30)),
                ("reference", icp_config.shipped_params())):
    kb = KeyframeBatch(ctx, fe.geometry, det.params["SOCA"], "SOCA", 65, p, B)
    kb.upload_scan_pairs(srcs, tgts, guesses)
    kbs[mode] = kb
combos = [(8, 24), (6, 24), (10, 24), (12, 24), (8, 16), (8, 32), (12, 32), (16, 32)]
if len(sys.argv) > 1:  # e.g. "4,24 6,24 8,40"
    combos = [tuple(int(v) for v in c.split(",")) for c in " ".join(sys.argv[1:]).split()]
for combo in combos:
    refill, budget = combo[:2]
    rtrips = combo[2] if len(combo) > 2 else 4
    os.environ["SFE_SW_BUDGET_A"] = str(refill)
    os.environ["SFE_SW_BUDGET"] = str(budget)
    os.environ["SFE_SW_RTRIPS"] = str(rtrips)
    line = "budget A %2d B %3d R %d:" % (refill, budget, rtrips)
    for mode, kb in kbs.items():
        ms = timed(kb.run_icp, 3)
        cyc = (ctypes.c_longlong * 96)()
        ctx._check(ctx.lib.sfe_icp_get_profile(ctx.handle, 1, cyc))
        kb.run_icp()
        ctx.sync()
        ctx._check(ctx.lib.sfe_icp_get_profile(ctx.handle, 0, cyc))
        line += "  %s %.2f ms (tier1 %dk tier2 %dk long %d trips %d | wave0: fetch %dk walk %dk finish %dk)" % (mode, ms, cyc[6] // 1000, cyc[7] // 1000, cyc[10], cyc[12], cyc[13] // 1000, cyc[14] // 1000, cyc[15] // 1000)
    print(line, flush=True)
