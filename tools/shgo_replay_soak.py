#!/usr/bin/env python
"""Soak of sonar_slam_amd/shgo_fast.py against scipy.optimize.shgo on the CPU (no device needed): random bounds, random integer step
functions (ordinary, plateaus, steps finer than SLSQP's finite-difference step), the one-iteration replay (SobolPlan.solve and the C
routine sfe_shgo_sobol_replay) and the several-iteration replay (replay_multi) with the parameters the reference uses and a few
others.  Compared: success, x, fun, and for the several-iteration form the multiset of evaluated points.
usage: shgo_replay_soak.py [--seconds 300] [--seed 1]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scipy.optimize import shgo  # noqa: E402

from sonar_slam_amd import _lib  # noqa: E402
from sonar_slam_amd import shgo_fast as sf  # noqa: E402


def step_function(rng, span, r):
    kind = int(rng.integers(6))
    if kind == 5:
        return sf.piecewise_constant(rng, span * (1e-7 if r % 2 else 1e-5))                     # finer than the step: hand-back
    if kind == 4:
        return sf.piecewise_constant(rng, span, coarse=True, n_planes=int(rng.integers(1, 4)))  # plateaus
    return sf.piecewise_constant(rng, span, coarse=(kind == 0), n_planes=int(rng.integers(4, 40)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300.0)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    lib = _lib.load_library()
    t_end = time.time() + a.seconds
    one = {sf.OK: 0, sf.FAILED: 0, sf.FALLBACK: 0}
    many = {sf.OK: 0, sf.FAILED: 0, sf.FALLBACK: 0}
    bad = r = 0
    plans = {}
    while time.time() < t_end:
        r += 1
        stds = np.array([[rng.uniform(0.05, 3), rng.uniform(0.05, 3), rng.uniform(0.005, 0.3)]]).T
        if r % 3 == 0:
            stds = np.array([[0.2, 0.2, 0.02]]).T            # slam.yaml's odom_sigmas: the sequential scan match's bounds
        bounds = 5.0 * np.c_[-stds, stds]
        span = bounds[:, 1] - bounds[:, 0]
        f = step_function(rng, span, r)
        if r % 2:
            # ---- one iteration ----
            n = int(rng.choice([50, 50, 30, 100]))
            key = (bounds.tobytes(), n)
            if key not in plans:
                if len(plans) > 40:
                    plans.clear()
                plans[key] = sf.SobolPlan(bounds, n, 0.01)
            plan = plans[key]
            table = np.array([[f(p) for p in row] for row in plan.points])
            st, x, fun = plan.solve(table)
            cst, cv = plan.solve_many(lib, table[None])
            if cst[0] != st or (st != sf.FALLBACK and not np.array_equal(plan.X[cv[0]], x)):
                bad += 1
                print("MISMATCH C routine vs definition, round %d" % r)
            one[st] += 1
            if st == sf.FALLBACK:
                continue
            res = shgo(func=f, bounds=bounds, n=n, iters=1, sampling_method="sobol", minimizer_kwargs={"options": {"ftol": 0.01}})
            if not (bool(res.success) == (st == sf.OK) and np.array_equal(res.x, x) and res.fun == fun):
                bad += 1
                print("MISMATCH one iteration, round %d (seed %d): n %d, %r vs %r, %r vs %r" % (r, a.seed, n, res.x, x, res.fun, fun))
        else:
            # ---- several iterations ----
            n, iters = [(100, 5), (100, 5), (50, 2), (100, 3), (64, 4)][int(rng.integers(5))]
            draws, cand, fd = sf.multi_candidates(bounds, n, iters)
            cost = np.array([f(p) for p in cand])
            fd_cost = np.array([[f(p) for p in row] for row in fd])
            st, x, fun, vertices, minimised = sf.replay_multi(bounds, n, iters, draws, cand, cost, fd_cost)
            many[st] += 1
            if st == sf.FALLBACK:
                continue
            asked = []

            def g(p):
                asked.append(tuple(np.asarray(p, float)))
                return f(p)
            res = shgo(func=g, bounds=bounds, n=n, iters=iters, sampling_method="sobol", minimizer_kwargs={"options": {"ftol": 0.01}})
            mine = [tuple(cand[v]) for v in vertices]
            for v in minimised:
                mine.append(tuple(cand[v]))
                mine.extend(tuple(p) for p in fd[v])
            if not (bool(res.success) == (st == sf.OK) and np.array_equal(res.x, x) and res.fun == fun and sorted(mine) == sorted(asked)):
                bad += 1
                print("MISMATCH %d iterations, round %d (seed %d): n %d, %r vs %r, %r vs %r, %d vs %d evaluations"
                      % (iters, r, a.seed, n, res.x, x, res.fun, fun, len(asked), len(mine)))
    print("shgo replay soak (seed %d): %d problems in %.0f s; one iteration: %d replayed, %d failing like shgo, %d handed back; several "
          "iterations: %d replayed, %d failing like shgo, %d handed back; %d mismatches"
          % (a.seed, r, a.seconds, one[sf.OK], one[sf.FAILED], one[sf.FALLBACK], many[sf.OK], many[sf.FAILED], many[sf.FALLBACK], bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
