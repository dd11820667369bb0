#!/usr/bin/env python
"""BASELINE configs[4] timing: 2048x1024 frames through CFAR+extract, and the many-to-one ICP batch
(30 guesses x one 20k x 20k pair, slam.yaml:34 / slam.py:346) with the sweep and the brute-force
kernels.  HIP-event times on the library's stream."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sonar_slam_amd import _lib, icp_config, synth  # noqa: E402
from sonar_slam_amd.CFAR import CFAR  # noqa: E402
from sonar_slam_amd.feature_extraction import FeatureExtraction, SonarPing, oculus_bearings  # noqa: E402
from sonar_slam_amd.pipeline import KeyframeBatch  # noqa: E402

ROWS, COLS, NF, NG, NP = 2048, 1024, 64, 30, 20000
ctx = _lib.default_context()
det = CFAR(40, 10, 0.1, 10)
fe = FeatureExtraction(ctx)
fe.Ntc, fe.Ngc, fe.Pfa, fe.rank, fe.alg, fe.threshold = 40, 10, 0.1, 10, "SOCA", 65
fe.configure()
base = [synth.sonar_frame(seed=s, rows=ROWS, cols=COLS, n_blobs=120) for s in range(8)]
frames = np.stack([base[j % 8] for j in range(NF)])
fe.generate_map_xy(SonarPing(frames[0], oculus_bearings(COLS), 30.0 / ROWS))
src, tgt, guess, _ = synth.scan_pair(seed=77, n_src=NP, n_tgt=NP)
b = synth.pose_of(guess)
rng = np.random.default_rng(1)
guesses = [synth.pose_matrix(b[0] + dx, b[1] + dy, b[2] + dt).astype(np.float32)
           for dx, dy, dt in rng.normal(0, [0.3, 0.3, 0.05], (NG, 3))]


def timed(fn, reps):
    fn()
    ctx.sync()
    ctx.timer_start()
    for _ in range(reps):
        fn()
    return ctx.timer_stop() / reps


for mode, p in (("p2plane30", icp_config.shipped_params(minimizer=1, use_diff_checker=0, max_iter=30)),
                ("reference", icp_config.shipped_params())):
    kb = KeyframeBatch(ctx, fe.geometry, det.params["SOCA"], "SOCA", 65, p, NG, max_points=65536)
    if mode == "p2plane30":
        kb2 = KeyframeBatch(ctx, fe.geometry, det.params["SOCA"], "SOCA", 65, p, NF, max_points=65536)
        kb2.upload_frames(frames)
        ms = timed(kb2.run_cfar, 10)
        print("cfar    %8.3f ms / %d frames %dx%d  (%.0f GB/s algorithmic)" % (ms, NF, ROWS, COLS, 2.0 * NF * ROWS * COLS / ms / 1e6))
        print("extract %8.3f ms / %d frames, mean %.0f points" % (timed(kb2.run_extract, 10), NF, kb2.results()["counts"].mean()))
        kb2.free()
    # many-to-one: the same pair under every guess (the library sorts / prepares the shared target once)
    kb.src_off = np.zeros(NG + 1, np.int32)
    kb.tgt_off = np.zeros(NG + 1, np.int32)
    icp = None
    from sonar_slam_amd import pcl
    icp = pcl.ICP(ctx)
    icp.setParams(p)
    for v in (0, 4):
        ctx._check(ctx.lib.sfe_icp_set_tuning(ctx.handle, v))
        import time
        icp.compute_batch(src, tgt, guesses)
        t0 = time.perf_counter()
        msgs, T, it = icp.compute_batch(src, tgt, guesses)
        dt = time.perf_counter() - t0
        print("icp %-9s variant %d: %8.2f ms for %d guesses x %dx%d (host call incl. copies), mean iters %.1f, ok %d"
              % (mode, v, 1e3 * dt, NG, NP, NP, it.mean(), sum(m == "success" for m in msgs)))
    ctx._check(ctx.lib.sfe_icp_set_tuning(ctx.handle, 0))
    kb.free()
