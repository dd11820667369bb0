#!/usr/bin/env python
"""Soak test of the resident per-ping chain (CFAR + gate -> extraction -> downsample -> outlier filter,
KeyframeBatch) against the oracle chain on random frame shapes, CFAR variants and filter parameters."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402
from sonar_slam_amd import _lib, synth  # noqa: E402
from sonar_slam_amd.CFAR import CFAR  # noqa: E402
from sonar_slam_amd.feature_extraction import FeatureExtraction, SonarPing, oculus_bearings  # noqa: E402
from sonar_slam_amd.pipeline import KeyframeBatch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=40)
    a = ap.parse_args()
    ctx = _lib.default_context()
    det = CFAR(40, 10, 0.1, 10)
    rng = np.random.default_rng(int(time.time()))
    t0, n, bad = time.time(), 0, 0
    while time.time() - t0 < a.seconds:
        rows, cols = int(rng.choice([128, 256, 300, 512, 1024])), int(rng.choice([64, 128, 256, 512]))
        alg = str(rng.choice(["CA", "SOCA", "GOCA", "OS"]))
        thr = int(rng.choice([-1, 40, 65, 120]))
        nf = int(rng.integers(1, 12))
        res, rad, mp = float(rng.choice([0.0, 0.25, 0.5, 1.0])), float(rng.choice([0.5, 1.0, 2.0])), int(rng.integers(0, 8))
        frames = np.stack([synth.sonar_frame(seed=int(s), rows=rows, cols=cols, n_blobs=int(rng.integers(0, 40)))
                           for s in rng.integers(0, 1 << 30, nf)])
        fe = FeatureExtraction(ctx)
        fe.generate_map_xy(SonarPing(frames[0], oculus_bearings(cols), 30.0 / rows))
        kb = KeyframeBatch(ctx, fe.geometry, det.params[alg], alg, thr, None, nf, max_points=16384)
        kb.upload_frames(frames)
        kb.run_cfar()
        kb.run_extract()
        kb.run_filter(res, rad, mp)
        ctx.sync()
        p = det.params[alg]
        for j in range(nf):
            k = p[2] if alg == "OS" else 0
            m = oracle.cfar(frames[j], alg, p[0], p[1], p[-1], k=k)
            if thr >= 0:
                m = oracle.gate(frames[j], m, thr)
            rc = oracle.nonzero(oracle.remap_u8(m, fe.map_x, fe.map_y))
            pts = oracle.px_to_m(rc, fe.rows, fe.cols, fe.width, fe.height)
            want = pts.astype(np.float32)
            if len(want) > 16384:
                continue                       # beyond the batch's point capacity: truncated by design
            if len(want) and res > 0:
                want = np.asarray(oracle.downsample(want, res), np.float32)
            if mp > 1 and len(want):
                want = np.asarray(oracle.remove_outlier(want, rad, mp), np.float32)
            ok = (np.array_equal(kb.mask(j), m) and np.array_equal(kb.points(j), pts) and
                  np.array_equal(kb.cloud(j), want.reshape(-1, 2)))
            n += 1
            if not ok:
                bad += 1
                print("MISMATCH", rows, cols, alg, thr, res, rad, mp, flush=True)
        kb.free()
        fe.geometry.close()
    print("pipeline soak: %d pings in %.0f s, %d mismatches" % (n, time.time() - t0, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
