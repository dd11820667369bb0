#!/usr/bin/env python
"""Soak test of the extraction: inverse-map pass vs dense pass vs the oracle on random geometries and
masks of random density (binary and non-binary)."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402
from sonar_slam_amd import _lib  # noqa: E402
from sonar_slam_amd.feature_extraction import FeatureExtraction, SonarPing, oculus_bearings  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=40)
    a = ap.parse_args()
    ctx = _lib.default_context()
    rng = np.random.default_rng(int(time.time()))
    t0, n, bad = time.time(), 0, 0
    while time.time() - t0 < a.seconds:
        beams = int(rng.choice([32, 64, 96, 128, 256, 100]))       # 100: not a multiple of 32 -> dense pass only
        ranges = int(rng.integers(20, 400))
        res = float(rng.uniform(0.05, 0.3))
        fe = FeatureExtraction(ctx)
        fe.generate_map_xy(SonarPing(np.zeros((ranges, beams), np.uint8), oculus_bearings(beams, float(rng.uniform(60, 140))), res))
        for _ in range(6):
            dens = float(rng.choice([0.001, 0.01, 0.1, 0.5, 1.0]))
            mask = (rng.random((ranges, beams)) < dens).astype(np.uint8)
            if rng.random() < 0.2:
                mask *= rng.integers(1, 255, mask.shape, dtype=np.uint8)   # non-binary values
            out = {}
            for v in (0, 1):
                ctx._check(ctx.lib.sfe_extract_set_tuning(ctx.handle, v))
                out[v] = fe.geometry.extract(mask)
            ctx._check(ctx.lib.sfe_extract_set_tuning(ctx.handle, 0))
            rc = oracle.nonzero(oracle.remap_u8(mask, fe.map_x, fe.map_y))
            ok = (all(np.array_equal(out[0][0], out[v][0]) and np.array_equal(out[0][1], out[v][1]) for v in (1,)) and
                  np.array_equal(out[0][0], rc))
            n += 1
            if not ok:
                bad += 1
                print("MISMATCH beams %d ranges %d res %.3f density %.3f" % (beams, ranges, res, dens), flush=True)
        fe.geometry.close()
    print("extract soak: %d masks in %.0f s, %d mismatches" % (n, time.time() - t0, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
