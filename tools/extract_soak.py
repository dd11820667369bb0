#!/usr/bin/env python
"""Soak test of the extraction: inverse-map pass vs dense pass vs the oracle on random geometries and
masks of random density (binary and non-binary); and (round 5) batches of bit streams through the record path -- the default of
the resident pipeline -- with random point capacities and, every other batch, capacities of the record path small enough that
some or all frames are handed back to the canvas kernels."""
import ctypes as C
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402
from sonar_slam_amd import _lib  # noqa: E402
from sonar_slam_amd.feature_extraction import FeatureExtraction, SonarPing, oculus_bearings  # noqa: E402


def bit_stream_batch(ctx, rng, fe, ranges, beams):
    """-> (frames, mismatches): a batch of binary masks as bit streams through sfe_extract_points_bits_batch_dev"""
    nf = int(rng.integers(1, 12))
    px = ranges * beams
    wpf = (px + 31) // 32 + 1
    masks, bits = [], np.zeros((nf, wpf), np.uint32)
    for f in range(nf):
        dens = float(rng.choice([0.0, 0.001, 0.01, 0.05, 0.3, 1.0]))
        m = (rng.random((ranges, beams)) < dens).astype(np.uint8)
        masks.append(m)
        flat = np.zeros(wpf * 32, np.uint8)
        flat[:px] = m.reshape(-1)
        bits[f] = np.packbits(flat.reshape(-1, 32)[:, ::-1], axis=1).view(">u4").reshape(-1).astype(np.uint32)
    want = [oracle.nonzero(oracle.remap_u8(m, fe.map_x, fe.map_y)) for m in masks]
    sizes = sorted(len(w) for w in want)
    cap = int(rng.choice([max(sizes[-1], 1) + 3, max(sizes[len(sizes) // 2], 1), 1 + sizes[-1] // 3 + 5]))
    env = {}
    if rng.random() < 0.5:
        env = {str(rng.choice(["SFE_EXTRACT_CAPW", "SFE_EXTRACT_REC_CAP"])): str(int(rng.choice([1, 7, 60, 300, 1500])))}
    d_bits, d_pts, d_cnt = ctx.alloc(bits.nbytes), ctx.alloc(nf * cap * 16), ctx.alloc(nf * 4)
    bad = 0
    try:
        os.environ.update(env)
        d_bits.upload(bits)
        ctx._check(ctx.lib.sfe_extract_points_bits_batch_dev(ctx.handle, fe.geometry.handle, d_bits.ptr, nf, cap, d_pts.ptr, d_cnt.ptr))
        ctx.sync()
        counts = d_cnt.download(np.int32, nf)
        pts = d_pts.download(np.float64, nf * cap * 2).reshape(nf, cap, 2)
        for f in range(nf):
            k = min(len(want[f]), cap)
            ref = oracle.px_to_m(want[f][:k], fe.rows, fe.cols, fe.width, fe.height)
            if counts[f] != len(want[f]) or not np.array_equal(pts[f, :k], ref):
                bad += 1
                print("MISMATCH bit stream: beams %d ranges %d frame %d of %d, %d points, cap %d, env %r"
                      % (beams, ranges, f, nf, len(want[f]), cap, env), flush=True)
    finally:
        for key in env:
            del os.environ[key]
        for b in (d_bits, d_pts, d_cnt):
            b.free()
    return nf, bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=40)
    a = ap.parse_args()
    ctx = _lib.default_context()
    rng = np.random.default_rng(int(time.time()))
    t0, n, bad, n_bits = time.time(), 0, 0, 0
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
        if beams % 32 == 0:
            nb, nbad = bit_stream_batch(ctx, rng, fe, ranges, beams)
            n_bits += nb
            bad += nbad
        fe.geometry.close()
    print("extract soak: %d masks (+ %d frames as bit streams through the record path) in %.0f s, %d mismatches"
          % (n, n_bits, time.time() - t0, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
