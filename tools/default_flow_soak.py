#!/usr/bin/env python
"""Randomised soak of the round-5 device code behind the reference's default flow: the store-side primitives of the loop-closure
search (keyed get_points of any size, field-of-view gate, compaction, keyed matching) against the oracle / numpy, and the
global-initialisation cost over handles (grids from target handles, both dtypes, many pairs per launch) against the oracle.
python tools/default_flow_soak.py [--seconds 60] [--seed 1]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402
from oracle import chain  # noqa: E402
from sonar_slam_amd import _lib  # noqa: E402
from sonar_slam_amd import matching_cost as mc  # noqa: E402
from sonar_slam_amd import store as st  # noqa: E402
from sonar_slam_amd.pose2 import Pose2  # noqa: E402
from sonar_slam_amd.replay import FrontEnd  # noqa: E402


def cloud(rng, n, spread=30.0):
    if n == 0:
        return np.zeros((0, 2), np.float32)
    kind = rng.integers(3)
    if kind == 0:       # fan-like scatter
        return np.c_[rng.uniform(1, spread, n), rng.uniform(-spread / 1.5, spread / 1.5, n)].astype(np.float32)
    if kind == 1:       # walls
        t = rng.uniform(0, 1, n)
        a, b = rng.uniform(-spread, spread, 2), rng.uniform(-spread, spread, 2)
        return (a[None] + t[:, None] * (b - a)[None] + rng.normal(0, 0.05, (n, 2))).astype(np.float32)
    p = rng.integers(0, int(spread * 4), (n, 2)).astype(np.float32) * 0.25      # a raster: ties, duplicates
    return p


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--dump", help="npz to write the first mismatching round's inputs to")
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    ctx = _lib.default_context()
    s = st.CloudStore(ctx, capacity_points=1 << 21, max_clouds=4096)
    t_end = time.time() + a.seconds
    rounds = bad = n_big = n_amb = 0
    while time.time() < t_end:
        s.truncate(0)
        m = int(rng.integers(1, 14))
        big = rng.random() < 0.08
        sizes = [int(rng.choice([0, 1, 2, 40, 300, 900, 2500])) if not big else int(rng.integers(4000, 12000)) for _ in range(m)]
        clouds = [cloud(rng, n) for n in sizes]
        hs = [s.put(c) for c in clouds]
        poses = [Pose2(*q) for q in np.c_[rng.normal(0, 8, m), rng.normal(0, 8, m), rng.normal(0, 1.0, m)]]
        keys = list(rng.permutation(m + 3)[:m])
        res = float(rng.choice([0.5, 0.25, 1.0]))
        n_big += sum(sizes) > 65536
        # ---- keyed get_points ----
        g = s.get_points_keys(hs, [st.pose_T6(p) for p in poses], keys, res)
        moved = [oracle.transform_points(c, p.matrix(), f64_points=True) for c, p in zip(clouds, poses)]
        allp = np.concatenate(moved) if moved else np.zeros((0, 2), np.float32)
        allk = np.concatenate([np.full(len(x), k, np.int32) for x, k in zip(moved, keys)]) if moved else np.zeros(0, np.int32)
        if len(allp):
            want, idx = oracle.downsample(allp, res, return_index=True)
            wk = allk[idx]
        else:
            want, wk = allp, allk
        gp, gk = s.read(g), s.read_keys(g)
        why = []
        ok = np.array_equal(gp, want) and np.array_equal(gk, wk)
        if not ok:
            why.append("keyed get_points: %d vs %d points, first difference at %s" % (len(gp), len(want), (np.nonzero((gp != want).any(axis=1))[0][:3] if gp.shape == want.shape else "-")))
        # ---- get_points without keys through the batched entry point (big path when > 65 536) ----
        ref = poses[int(rng.integers(m))]
        h2 = s.get_points([hs], [[st.pose_T6(ref.between(p)) for p in poses]], res)[0]
        w2 = oracle.get_points(clouds, [ref.between(p).matrix() for p in poses], res)
        g2 = s.read(h2)
        if not np.array_equal(g2, w2):
            ok = False
            why.append("get_points (batched entry): %d vs %d points, first difference at %s"
                       % (len(g2), len(w2), (np.nonzero((g2 != w2).any(axis=1))[0][:3] if g2.shape == w2.shape else "-")))
        # ---- field-of-view gate + per-key counts + compaction ----
        nf = int(rng.integers(1, 6))
        frames = [Pose2(*q) for q in np.c_[rng.normal(0, 10, nf), rng.normal(0, 10, nf), rng.normal(0, 1.5, nf)]]
        Tinv = [f.inverse() for f in frames]
        rb = list(rng.uniform(5, 40, nf))
        bb = list(rng.uniform(0.2, 3.3, nf))
        nk = m + 3
        sel = FrontEnd._fov_numpy(gp, Tinv, rb, bb) if len(gp) else np.zeros(0, bool)
        hist, n_sel, amb = s.fov_select(g, [st.pose_T6(t) for t in Tinv], rb, bb, nk)
        if amb:
            n_amb += 1
            s.set_selection(g, sel)
        else:
            if not (n_sel == int(sel.sum()) and np.array_equal(hist, np.bincount(gk[sel], minlength=nk))):
                ok = False
                why.append("fov_select: %d selected vs %d" % (n_sel, int(sel.sum())))
        c = s.compact_selected(g)
        if not (np.array_equal(s.read(c), gp[sel]) and np.array_equal(s.read_keys(c), gk[sel])):
            ok = False
            why.append("compact_selected")
        # ---- keyed matching of a moved float32 source ----
        src = cloud(rng, int(rng.choice([0, 1, 50, 700])))
        hsrc = s.put(src)
        est = Pose2(*rng.normal(0, [3, 3, 0.5]))
        h1, ov = s.match_keys(hsrc, st.pose_T6(est), c, 0.5, nk, flags=st.F32_POINTS)
        tsel, ksel = gp[sel], gk[sel]
        if len(src) and len(tsel):
            ids = oracle.match(tsel, oracle.transform_points(src, est.matrix(), f64_points=False), 0.5)[0].reshape(-1)
            if not (ov == int(np.sum(ids != -1)) and np.array_equal(h1, np.bincount(ksel[ids[ids != -1]], minlength=nk))):
                ok = False
                why.append("match_keys: overlap %d vs %d" % (ov, int(np.sum(ids != -1))))
        elif not (ov == 0 and not h1.any()):
            ok = False
            why.append("match_keys on an empty cloud")
        # ---- cost grids over handles: several (source, target) pairs in one launch, both dtypes ----
        pairs = [(i, j) for i in range(m) for j in range(m) if sizes[i] >= 1 and sizes[j] >= 2 and sizes[j] <= 3000][:6]
        if pairs:
            P = int(rng.choice([rng.integers(1, 20), rng.integers(32, 90)]))     # (>= 32: the many-poses kernel)
            for f64 in (True, False):
                T6 = rng.normal(0, 1, (len(pairs), P, 6)).astype(np.float32)
                T6[:, :, [0, 4]] += 1.0
                costs, grids = mc.batch_store(s, [hs[i] for i, _ in pairs], [hs[j] for _, j in pairs], T6, f64_points=f64)
                for q, (i, j) in enumerate(pairs):
                    tgt = clouds[j]
                    xmin, ymin, r_, rows, cols, hsz = chain.grid_geometry(tgt, 0.5)
                    rr = np.clip(np.int32(np.round((tgt[:, 1] - ymin) / r_)), 0, rows - 1)
                    cc = np.clip(np.int32(np.round((tgt[:, 0] - xmin) / r_)), 0, cols - 1)
                    grid = oracle.cost_grid(rr, cc, rows, cols, hsz)
                    if not np.array_equal(costs[q], oracle.matching_cost(grid, clouds[i], T6[q], xmin, ymin, r_, f64_points=f64)):
                        ok = False
                        why.append("matching cost, pair %d (f64 %s)" % (q, f64))
                    if q == 0 and not np.array_equal(grids.download(0), grid):
                        ok = False
                        why.append("cost grid")
                grids.close()
        rounds += 1
        if not ok:
            bad += 1
            print("MISMATCH in round %d (seed %d): sizes %r res %g: %s" % (rounds, a.seed, sizes, res, "; ".join(why)))
            if a.dump:
                np.savez(a.dump, poses=np.array([[p.x(), p.y(), p.theta()] for p in poses]), ref=np.array([ref.x(), ref.y(), ref.theta()]),
                         res=res, fov_frames=np.array([[f.x(), f.y(), f.theta()] for f in frames]), range_bounds=np.array(rb),
                         bearing_bounds=np.array(bb), keys=np.array(keys), **{"cloud%d" % i: c for i, c in enumerate(clouds)})
    print("default-flow soak: %d rounds in %.0f s (%d with a target beyond 65 536 points, %d undecidable gates handed to numpy), %d mismatches"
          % (rounds, a.seconds, n_big, n_amb, bad))
    s.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
