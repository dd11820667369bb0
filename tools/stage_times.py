#!/usr/bin/env python
"""Per-stage HIP-event timing of the keyframe pipeline (CFAR / extract / ICP) for quick A/B."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from sonar_slam_amd import _lib, icp_config  # noqa: E402
from sonar_slam_amd.CFAR import CFAR  # noqa: E402
from sonar_slam_amd.feature_extraction import FeatureExtraction, SonarPing, oculus_bearings  # noqa: E402
from sonar_slam_amd.pipeline import KeyframeBatch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--icp-variants", default="0,4")  # 0 = sweep, 4 = brute force
    ap.add_argument("--p2plane-only", action="store_true", help="skip the shipped chain (counter passes)")
    ap.add_argument("--max-iter", type=int, default=30, help="iterations of the forced point-to-plane chain")
    ap.add_argument("--max-points", type=int, default=32768, help="point capacity per frame (the bench's)")
    a = ap.parse_args()
    ctx = _lib.default_context()
    det = CFAR(40, 10, 0.1, 10)
    fe = FeatureExtraction(ctx)
    fe.Ntc, fe.Ngc, fe.Pfa, fe.rank, fe.alg, fe.threshold = 40, 10, 0.1, 10, "SOCA", 65
    fe.configure()
    frames, srcs, tgts, guesses = bench.make_inputs(0, a.batch)
    fe.generate_map_xy(SonarPing(frames[0], oculus_bearings(bench.COLS), 30.0 / bench.ROWS))

    def timed(fn, reps):
        fn()
        ctx.sync()
        ctx.timer_start()
        for _ in range(reps):
            fn()
        return ctx.timer_stop() / reps
    for mode, p in (("p2plane30", icp_config.shipped_params(minimizer=1, use_diff_checker=0, max_iter=a.max_iter)),
                    ("reference", icp_config.shipped_params())):
        if a.p2plane_only and mode != "p2plane30":
            continue
        kb = KeyframeBatch(ctx, fe.geometry, det.params["SOCA"], "SOCA", 65, p, a.batch, max_points=a.max_points)
        kb.upload_frames(frames)
        kb.upload_scan_pairs(srcs, tgts, guesses)
        if mode == "p2plane30":
            print("cfar    %8.3f ms / %d frames" % (timed(kb.run_cfar, 10), a.batch))
            print("extract %8.3f ms / %d frames" % (timed(kb.run_extract, 10), a.batch))
            print("filter  %8.3f ms / %d frames (downsample 0.5 + remove_outlier 1.0/5, device resident)"
                  % (timed(kb.run_filter, 5), a.batch))
        for v in [int(x) for x in a.icp_variants.split(",")]:
            ctx._check(ctx.lib.sfe_icp_set_tuning(ctx.handle, v))
            print("icp %-9s variant %d %8.3f ms / %d jobs" % (mode, v, timed(kb.run_icp, 3), a.batch))
            if not (v & 4):
                import ctypes
                cyc = (ctypes.c_longlong * 96)()
                ctx._check(ctx.lib.sfe_icp_get_profile(ctx.handle, 1, cyc))
                kb.run_icp()
                ctx.sync()
                ctx._check(ctx.lib.sfe_icp_get_profile(ctx.handle, 0, cyc))
                names = ["setup", "first pass", "-", "quantile", "reduce", "solve", "later passes", "tier2", "census",
                         "#rounds", "#long", "#second-pass"]
                it = int(kb.results()["iters"][0])
                print("   workgroup 0, %d iterations: " % it +
                      ", ".join("%s %d" % (n, c) for n, c in zip(names, list(cyc)[:12]) if n != "-"))
                print("   whole launch: %d lane-tier + %d cooperative + %d witness evaluations, %d lower-bound probes, %d "
                      "iterations -> %.1f evaluations per query and iteration"
                      % (cyc[80], cyc[81], cyc[82], cyc[83], cyc[84],
                         (cyc[80] + cyc[81] + cyc[82]) / max(1.0, cyc[84] * float(bench.N_PTS))))
                print("   tier 2: %d trips over %d walks; wave 0 of workgroup 0: fetch %d, walk %d, finish %d cycles"
                      % (cyc[12], cyc[10], cyc[13], cyc[14], cyc[15]))
                if it > 25 and cyc[93] > cyc[85] > 0:
                    st = [int(cyc[85 + k]) for k in range(9)]
                    names2 = ["bounds+reset", "fresh/triage pass", "second pass", "cooperative tier", "census+rounds", "quantile", "sums", "solve"]
                    print("   iteration 25 of workgroup 0, cycles by phase: " + ", ".join("%s %d" % (n, b - a_) for n, a_, b in zip(names2, st[:-1], st[1:]))
                          + " = %d" % (st[8] - st[0]))
                if it <= 10:
                    print("   first iteration by round (queries, second pass, long, kcycles since start): " + " ".join(
                        "%d/%d/%d/%d" % (cyc[16 + 24 + 4 * r], cyc[16 + 25 + 4 * r], cyc[16 + 26 + 4 * r], cyc[16 + 27 + 4 * r] // 1000)
                        for r in range(8)))
                import struct
                print("   per iteration (kcycles search, cap C, n_exact, settled by a clearance record): " + " ".join(
                    "%d/%.3g/%d/%d" % (cyc[16 + 2 * i] // 1000, struct.unpack("f", struct.pack("I", (cyc[17 + 2 * i] >> 32) & 0xFFFFFFFF))[0],
                                       cyc[17 + 2 * i] & 0xFFFF, (cyc[17 + 2 * i] >> 16) & 0xFFFF) for i in range(min(it, 32))))
        ctx._check(ctx.lib.sfe_icp_set_tuning(ctx.handle, 0))
        kb.free()


if __name__ == "__main__":
    main()
