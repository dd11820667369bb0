#!/usr/bin/env python
"""Soak: the HIP ICP path (strip sweep, default) against the CPU ORACLE on thousands of random scan matches --
bench-like pairs with the shipped chains, and small random problems with random chain parameters including the
degenerate classes of tests/test_gpu_icp.py::test_sweep_vs_brute_force_fuzz (points on a raster, exact duplicates,
sources beyond maxDist, NaN / inf coordinates, 1-point clouds).

Checked per scan match, against the oracle with fp64 sums (same discrete decisions by construction): status equal,
iteration count equal, pose within 1e-6 (point-to-point) / 1e-4 (point-to-plane: cos/sin of the step and the
Cholesky solve round differently).  A disagreement counts as a MISMATCH unless the problem is ill-conditioned in the
checkable sense that the oracle disagrees with ITSELF between float sums (PointMatcher<float>) and fp64 sums
(status, iteration count, or pose by more than 1e-3): e.g. every inlier matched to one target point, where the
cross-covariance is rounding noise and the rotation any angle.  Those are listed separately (`ill_conditioned`).
An ill-conditioned job on which the HIP path differs from the fp64-sum oracle (`ill_conditioned_disagreeing`) is not waved
through on that alone (VERDICT r5 "what's weak" 4): it is written out in full (`ill_conditioned_disagreeing_jobs`) and
re-examined -- the fp64-sum oracle runs the same job again with the guess moved by ONE float ulp in x, y (four runs) and
with the source points reordered (three runs: only the order of its sums changes).  If the oracle's own answer (status,
iteration count, pose beyond the tolerance) changes under a perturbation that is below the resolution of the problem, no
implementation can be held to its unperturbed answer: the job stays excused, with the evidence in the record.  If the
oracle is stable under all of them the disagreement is the kernel's and counts as a mismatch.
Against the oracle in float the pose difference is reported for the bench-like class next to the float oracle's own
distance from its fp64 version (the float sums' accumulation noise, which grows with the cloud size).

The oracle runs on the host cores (worker pool forked before the first HIP call).  Prints a JSON summary; exit
code 1 on any mismatch.  Test infrastructure: imports oracle/ as the checker, never as the product."""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402
from sonar_slam_amd import _lib, pcl, synth  # noqa: E402

FIELDS = ["matcher_max_dist", "use_max_dist_filter", "max_dist_filter", "use_trimmed_filter", "trim_ratio", "minimizer",
          "max_iter", "use_diff_checker", "min_diff_rot", "min_diff_trans", "smooth_len", "normals_knn"]


def pose_diff(Ta, Tb):
    a, b = synth.pose_of(Ta), synth.pose_of(Tb)
    return max(abs(a[0] - b[0]), abs(a[1] - b[1]), abs(np.arctan2(np.sin(a[2] - b[2]), np.cos(a[2] - b[2]))))


def shipped(minimizer):
    p = dict(matcher_max_dist=10.0, use_max_dist_filter=1, max_dist_filter=3.0, use_trimmed_filter=1, trim_ratio=0.8,
             minimizer=minimizer, max_iter=40, use_diff_checker=1, min_diff_rot=0.01, min_diff_trans=0.1, smooth_len=4,
             normals_knn=10)
    if minimizer == 1:
        p.update(max_iter=30, use_diff_checker=0)
    return p


def make_problems(seed, n_big, n_small):
    rng = np.random.default_rng(seed)
    probs = []
    for i in range(n_big):
        n = int(rng.choice([500, 1200, 2500, 5000]))
        s, t, g, _ = synth.scan_pair(seed=int(rng.integers(0, 1 << 30)), n_src=n, n_tgt=int(n * rng.uniform(0.8, 1.3)))
        probs.append(("bench", s, t, g, shipped(i % 2)))
    for case in range(n_small):
        ns, nt = int(rng.integers(1, 400)), int(rng.integers(1, 400))
        if case % 7 == 0:
            ns, nt = int(rng.integers(1, 6)), int(rng.integers(1, 6))
        tgt = rng.uniform(-8, 8, (nt, 2)).astype(np.float32)
        if case % 3 == 0:
            tgt[:, 0] = np.round(tgt[:, 0])
        if case % 5 == 0 and nt > 4:
            tgt[nt // 2:] = tgt[:nt - nt // 2]
        th = rng.uniform(-0.2, 0.2)
        R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
        src = (tgt[rng.integers(0, nt, ns)] @ R.T + rng.normal(0, 0.05, (ns, 2)) + rng.uniform(-0.5, 0.5, 2)).astype(np.float32)
        if case % 4 == 0:
            src[rng.integers(0, ns)] += 100.0
        if case % 11 == 0:
            src[rng.integers(0, ns), 0] = np.nan
        if case % 13 == 0:
            tgt[rng.integers(0, nt), 1] = np.inf
        p = dict(matcher_max_dist=float(rng.choice([0.5, 3.0, 10.0])), use_max_dist_filter=int(rng.integers(0, 2)),
                 max_dist_filter=float(rng.choice([0.3, 3.0, 20.0])), use_trimmed_filter=int(rng.integers(0, 2)),
                 trim_ratio=float(rng.choice([0.3, 0.8, 1.0])), minimizer=int(rng.integers(0, 2)),
                 max_iter=int(rng.integers(1, 15)), use_diff_checker=int(rng.integers(0, 2)), min_diff_rot=0.001,
                 min_diff_trans=0.01, smooth_len=int(rng.integers(1, 4)), normals_knn=int(rng.integers(2, 17)))
        g = synth.pose_matrix(*rng.normal(0, [0.3, 0.3, 0.05])).astype(np.float32)
        probs.append(("small", src, tgt, g, p))
    return probs


_P = []


def _oracle_one(i):
    kind, s, t, g, p = _P[i]
    oracle.set_kdtree(1 if kind == "bench" else 0)
    d = oracle.icp(s, t, g, oracle.IcpParams(precision=1, **p))
    f = oracle.icp(s, t, g, oracle.IcpParams(precision=0, **p))
    return i, d, f


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--big", type=int, default=800)
    ap.add_argument("--small", type=int, default=1600)
    ap.add_argument("--seed", type=int, default=20260926)
    ap.add_argument("--workers", type=int, default=0)
    a = ap.parse_args()
    t0 = time.time()
    _P.extend(make_problems(a.seed, a.big, a.small))
    n = len(_P)
    workers = a.workers or min(64, len(os.sched_getaffinity(0)))
    pool = mp.get_context("fork").Pool(workers)          # before any HIP call
    pending = pool.map_async(_oracle_one, range(n), chunksize=4)

    ctx = _lib.default_context()
    got = [None] * n
    # group by chain parameters: one launch per group
    groups = {}
    for i, (kind, s, t, g, p) in enumerate(_P):
        groups.setdefault(tuple(p[k] for k in FIELDS), []).append(i)
    for key, idx in groups.items():
        icp = pcl.ICP(ctx)
        icp.setParams(_lib.IcpParams(**dict(zip(FIELDS, key))))
        for c0 in range(0, len(idx), 256):
            part = idx[c0:c0 + 256]
            msgs, T, it = icp.compute_pairs([_P[i][1] for i in part], [_P[i][2] for i in part], [_P[i][3] for i in part])
            for i, m, Tj, itj in zip(part, msgs, T, it):
                got[i] = (m, Tj, int(itj))
    t_gpu = time.time() - t0
    res = pending.get()
    pool.close()
    out = {"scan_matches": n, "bench_like": a.big, "small_random": a.small, "seed": a.seed, "status_mismatch": 0,
           "iteration_mismatch": 0, "pose_mismatch_f64": 0, "ill_conditioned": 0, "ill_conditioned_disagreeing": 0,
           "max_pose_diff_f64_p2p": 0.0, "max_pose_diff_f64_p2plane": 0.0, "max_pose_diff_float_bench": 0.0,
           "max_float_oracle_vs_f64_oracle_bench": 0.0, "bench_beyond_1e-4_of_float_oracle": 0, "failures_agreed": 0,
           "successes": 0}
    bad = []
    for i, (st_d, T_d, it_d), (st_f, T_f, it_f) in res:
        kind, s, t, g, p = _P[i]
        m, T, it = got[i]
        ill = st_d != st_f or it_d != it_f or (st_d == 0 and not pose_diff(T_d, T_f) <= 1e-3)
        out["ill_conditioned"] += ill
        why = None
        if m != oracle.ICP_STATUS_MESSAGES[st_d]:
            why = ("status", m, st_d)
        elif st_d != 0:
            if not np.array_equal(T, g):
                why = ("guess not returned",)
        else:
            d = pose_diff(T, T_d)
            if it != it_d:
                why = ("iters", it, it_d)
            elif not d <= (1e-4 if p["minimizer"] else 1e-6):
                why = ("pose", float(d))
        if why is not None and ill:
            # not excused on the oracle's float/fp64 disagreement alone: is the fp64-sum oracle's answer stable under a
            # one-ulp move of the guess?
            tol = 1e-4 if p["minimizer"] else 1e-6
            oracle.set_kdtree(1 if kind == "bench" else 0)
            unstable = []
            for (r, c, sgn) in ((0, 2, 1), (0, 2, -1), (1, 2, 1), (1, 2, -1)):
                g2 = np.array(g, np.float32, copy=True)
                g2[r, c] = np.nextafter(g2[r, c], np.float32(np.inf * sgn))
                st2, T2, it2 = oracle.icp(s, t, g2, oracle.IcpParams(precision=1, **p))
                moved = st2 != st_d or it2 != it_d or (st_d == 0 and not pose_diff(T2, T_d) <= tol)
                unstable.append({"guess_entry": [r, c], "ulp": sgn, "status": int(st2), "iters": int(it2),
                                 "pose_diff_vs_unperturbed": float(pose_diff(T2, T_d)) if st_d == 0 and st2 == 0 else None,
                                 "oracle_answer_changed": bool(moved)})
            # ... and under a mere REORDERING of the source points (the same problem; only the order of the fp64 sums changes)?
            # A rank-deficient system -- every kept pair on one target point -- amplifies the sums' rounding noise without
            # bound: the oracle's pose then moves by radians when its input is permuted, and so does any other implementation's
            # relative to it (the HIP kernels agree among themselves because they share one canonical order).
            for name, perm in (("reversed", np.arange(len(s))[::-1]), ("rolled by half", np.roll(np.arange(len(s)), len(s) // 2)),
                               ("random permutation (seed 0)", np.random.default_rng(0).permutation(len(s)))):
                st2, T2, it2 = oracle.icp(np.ascontiguousarray(s[perm]), t, g, oracle.IcpParams(precision=1, **p))
                moved = st2 != st_d or it2 != it_d or (st_d == 0 and not pose_diff(T2, T_d) <= tol)
                unstable.append({"source_order": name, "status": int(st2), "iters": int(it2),
                                 "pose_diff_vs_unperturbed": float(pose_diff(T2, T_d)) if st_d == 0 and st2 == 0 else None,
                                 "oracle_answer_changed": bool(moved)})
            oracle.set_kdtree(0)
            excused = any(u["oracle_answer_changed"] for u in unstable)
            out.setdefault("ill_conditioned_disagreeing_jobs", []).append({
                "index": int(i), "kind": kind, "n_src": int(len(s)), "n_tgt": int(len(t)), "params": p,
                "disagreement": [str(x) for x in why], "hip": {"message": m, "iters": int(it), "pose": [float(x) for x in synth.pose_of(T)]},
                "oracle_f64_sums": {"status": int(st_d), "iters": int(it_d), "pose": [float(x) for x in synth.pose_of(T_d)]},
                "oracle_float": {"status": int(st_f), "iters": int(it_f), "pose": [float(x) for x in synth.pose_of(T_f)]},
                "oracle_f64_sums_under_perturbations_below_the_problems_resolution": unstable,
                "excused": bool(excused),
                "reading": ("the fp64-sum oracle's own answer changes when the guess moves by one float ulp or the source points "
                            "are reordered: the job has no answer an implementation could be held to" if excused else
                            "the fp64-sum oracle is stable under one-ulp moves of the guess and reorderings of the source: "
                            "counted as a mismatch")})
            if excused:
                out["ill_conditioned_disagreeing"] += 1
                continue
        if why is not None:
            out[{"status": "status_mismatch", "iters": "iteration_mismatch"}.get(why[0], "pose_mismatch_f64")] += 1
            bad.append((i,) + why)
            continue
        if st_d != 0:
            out["failures_agreed"] += 1
            continue
        out["successes"] += 1
        key = "max_pose_diff_f64_p2plane" if p["minimizer"] else "max_pose_diff_f64_p2p"
        out[key] = max(out[key], float(pose_diff(T, T_d)))
        if kind == "bench" and st_f == 0:
            df = float(pose_diff(T, T_f))
            out["max_pose_diff_float_bench"] = max(out["max_pose_diff_float_bench"], df)
            out["max_float_oracle_vs_f64_oracle_bench"] = max(out["max_float_oracle_vs_f64_oracle_bench"],
                                                              float(pose_diff(T_d, T_f)))
            out["bench_beyond_1e-4_of_float_oracle"] += not df <= 1e-4
    out["seconds"] = round(time.time() - t0, 1)
    out["gpu_seconds_incl_setup"] = round(t_gpu, 1)
    out["oracle_workers"] = workers
    out["device"] = ctx.name()
    for b in bad[:40]:
        kind, s, t, g, p = _P[b[0]]
        print("MISMATCH", b, kind, len(s), len(t), p, file=sys.stderr, flush=True)
    print(json.dumps(out))
    sys.exit(1 if (out["status_mismatch"] or out["iteration_mismatch"] or out["pose_mismatch_f64"]) else 0)


if __name__ == "__main__":
    main()
