#!/usr/bin/env python
"""Soak test: the sweep ICP kernels against the brute-force kernel on many random problems (bench-size
pairs and small degenerate ones); any difference in transform, status or iteration count is a bug."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sonar_slam_amd import _lib, icp_config, pcl, synth  # noqa: E402
from sonar_slam_amd._lib import IcpParams  # noqa: E402


LAST = {}


def both(ctx, p, srcs, tgts, gs):
    out = []
    for v in (0, 4):
        ctx._check(ctx.lib.sfe_icp_set_tuning(ctx.handle, v))
        icp = pcl.ICP(ctx)
        icp.setParams(p)
        out.append(icp.compute_pairs(srcs, tgts, gs))
    ctx._check(ctx.lib.sfe_icp_set_tuning(ctx.handle, 0))
    a, b = out
    LAST.update(a=a, b=b)
    return a[0] == b[0] and np.array_equal(a[1], b[1], equal_nan=True) and np.array_equal(a[2], b[2])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60)
    a = ap.parse_args()
    ctx = _lib.default_context()
    rng = np.random.default_rng(int(time.time()))
    t0, n_small, n_big, bad = time.time(), 0, 0, 0
    while time.time() - t0 < a.seconds:
        # a batch of bench-size pairs, both shipped chains
        seeds = rng.integers(0, 1 << 30, 16)
        pairs = [synth.scan_pair(seed=int(s), n_src=int(rng.integers(3000, 6000)), n_tgt=int(rng.integers(3000, 8192)))
                 for s in seeds]
        for p in (icp_config.shipped_params(), icp_config.shipped_params(minimizer=1, use_diff_checker=0, max_iter=30)):
            ok = both(ctx, p, [q[0] for q in pairs], [q[1] for q in pairs], [q[2] for q in pairs])
            n_big += len(pairs)
            if not ok:
                bad += 1
                print("MISMATCH big seeds", seeds.tolist(), p.as_dict(), flush=True)
        # small random problems with random parameters
        srcs, tgts, gs = [], [], []
        for _ in range(64):
            ns, nt = int(rng.integers(1, 300)), int(rng.integers(1, 300))
            tgt = rng.uniform(-8, 8, (nt, 2)).astype(np.float32)
            if rng.random() < 0.3:
                tgt[:, 0] = np.round(tgt[:, 0] * 2) / 2
            if rng.random() < 0.3 and nt > 3:
                tgt[nt // 2:] = tgt[:nt - nt // 2]
            src = (tgt[rng.integers(0, nt, ns)] + rng.normal(0, 0.1, (ns, 2))).astype(np.float32)
            if rng.random() < 0.3:
                src[rng.integers(0, ns)] += 50
            srcs.append(src)
            tgts.append(tgt)
            gs.append(synth.pose_matrix(*rng.normal(0, [0.3, 0.3, 0.05])).astype(np.float32))
        p = IcpParams(matcher_max_dist=float(rng.choice([0.5, 3.0, 10.0])), use_max_dist_filter=int(rng.integers(0, 2)),
                      max_dist_filter=float(rng.choice([0.3, 3.0, 20.0])), use_trimmed_filter=int(rng.integers(0, 2)),
                      trim_ratio=float(rng.choice([0.3, 0.8, 1.0])), minimizer=int(rng.integers(0, 2)),
                      max_iter=int(rng.integers(1, 15)), use_diff_checker=int(rng.integers(0, 2)), min_diff_rot=0.001,
                      min_diff_trans=0.01, smooth_len=int(rng.integers(1, 4)), normals_knn=int(rng.integers(2, 17)))
        ok = both(ctx, p, srcs, tgts, gs)
        n_small += len(srcs)
        if not ok:
            bad += 1
            print("MISMATCH small", p.as_dict(), flush=True)
            a_, b_ = LAST["a"], LAST["b"]
            for j in range(len(srcs)):       # which jobs, and by how much
                if a_[0][j] != b_[0][j] or a_[2][j] != b_[2][j] or not np.array_equal(a_[1][j], b_[1][j], equal_nan=True):
                    print("   job %d: %d x %d points, messages %r / %r, iterations %d / %d, max |dT| %.3e"
                          % (j, len(srcs[j]), len(tgts[j]), a_[0][j], b_[0][j], a_[2][j], b_[2][j],
                             float(np.nanmax(np.abs(a_[1][j] - b_[1][j])))), flush=True)
            os.makedirs("gpurun_out", exist_ok=True)
            np.savez("gpurun_out/icp_soak_mismatch_%d.npz" % bad, params=np.array(list(p.as_dict().items()), dtype=object),
                     srcs=np.array(srcs, dtype=object), tgts=np.array(tgts, dtype=object), gs=np.stack(gs), allow_pickle=True)
    print("soak: %d bench-size and %d small scan matches compared in %.0f s, %d mismatching batches"
          % (n_big, n_small, time.time() - t0, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
