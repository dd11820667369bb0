#!/usr/bin/env python
"""One-knob-at-a-time sweep of the ICP launcher's environment knobs (read per call: DESIGN appendix in LAB_NOTEBOOK.md) on the
bench's 4096 scan pairs, 30-iteration point-to-plane chain: ms per launch (loop + preparation), the default measured again
between the groups.  Every setting returns identical results; this only re-checks the tuning after a kernel change.
usage: python tools/icp_env_sweep.py [NAME=v1,v2,... ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from sonar_slam_amd import _lib, icp_config  # noqa: E402
from sonar_slam_amd.CFAR import CFAR  # noqa: E402
from sonar_slam_amd.feature_extraction import FeatureExtraction, SonarPing, oculus_bearings  # noqa: E402
from sonar_slam_amd.pipeline import KeyframeBatch  # noqa: E402

B = int(os.environ.get("SWEEP_BATCH", "4096"))
groups = sys.argv[1:] or ["SFE_SW_MARGIN=10,25", "SFE_SW_RECM=6,12", "SFE_SW_RECK=2,4", "SFE_SW_BUDGET_A=4,8", "SFE_SW_RTRIPS=3,6",
                          "SFE_SW_BUDGET=64,256", "SFE_SW_UNION_MAX=512,1024", "SFE_SW_UNION_ITERS=0,2", "SFE_SW_STRIP_PTS=64,128"]
ctx = _lib.default_context()
det = CFAR(40, 10, 0.1, 10)
fe = FeatureExtraction(ctx)
fe.Ntc, fe.Ngc, fe.Pfa, fe.rank, fe.alg, fe.threshold = 40, 10, 0.1, 10, "SOCA", 65
fe.configure()
frames, srcs, tgts, guesses = bench.make_inputs(0, B)
fe.generate_map_xy(SonarPing(frames[0], oculus_bearings(bench.COLS), 30.0 / bench.ROWS))
kb = KeyframeBatch(ctx, fe.geometry, det.params["SOCA"], "SOCA", 65,
                   icp_config.shipped_params(minimizer=1, use_diff_checker=0, max_iter=30), B)
kb.upload_scan_pairs(srcs, tgts, guesses)


def timed(reps=3):
    kb.run_icp()
    ctx.sync()
    ctx.timer_start()
    for _ in range(reps):
        kb.run_icp()
    return ctx.timer_stop() / reps


for _ in range(2):
    timed(2)                                        # clocks
print("default                    %.3f ms" % timed(), flush=True)
for g in groups:
    name, vals = g.split("=")
    for v in vals.split(","):
        os.environ[name] = v
        print("%-18s %-7s %.3f ms" % (name, v, timed()), flush=True)
    del os.environ[name]
    print("default                    %.3f ms" % timed(), flush=True)
