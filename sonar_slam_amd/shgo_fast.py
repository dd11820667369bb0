"""scipy.optimize.shgo(sampling_method="sobol", iters=1) for a PIECEWISE-CONSTANT cost, without the per-call bookkeeping.

The reference runs `shgo(func=subroutine, n=50, iters=1, sampling_method="sobol", minimizer_kwargs={"options": {"ftol": ...}})`
in front of every sequential scan match (slam.py:692-701).  The cost (`get_matching_cost_subroutine1`, slam.py:529-567) is an
integer count of grid cells: piecewise constant in the pose.  For such a function everything shgo does AFTER the sampling stage is a
function of the cost at a fixed set of points that depends on (bounds, n) only:

* the sampling points, their Delaunay triangulation and the vertex-vertex graph shgo derives from it (`Complex.vf_to_vv` -- including
  its habit of connecting only the first two vertices of every 3-combination of a simplex) do not depend on the cost;
* a vertex is a minimiser when its cost is strictly below every neighbour's (`VertexBase.minimiser`);
* the pool is minimised in the order of `SHGO.minimise_pool`: first the first minimiser in vertex-cache order, then always the one
  FARTHEST (`cdist`) from the last local result;
* the local minimiser is SLSQP with a forward-difference gradient: it asks for the cost at the start x0 and at x0 + 1.49e-8 e_i.  When
  those four values agree the gradient is zero and SLSQP returns x0 itself (bit for bit) with the start's cost;
* the result is the first entry of `np.argsort` over the local results' costs.

`SobolPlan` takes all of that from ONE run of the installed scipy on a constant function (so the qhull triangulation, the graph quirks,
the vertex order and the finite-difference points are scipy's own, whatever its version), and `solve` / `solve_many` replay the rest on
a table of costs [n_vertices x 4] that the caller fills in one launch (`plan.points`).  What the replay cannot decide the way scipy
would -- a finite-difference point whose cost differs from its vertex's (SLSQP then moves), two pool members exactly equally far from
the last result (np.argsort's choice among equal keys is the sort kernel's) -- is REPORTED (`FALLBACK`) and the caller runs
scipy.optimize.shgo itself for that problem; equal lowest costs among the local results are resolved by asking np.argsort itself.  `SobolPlan.self_check` compares the replay with
scipy.optimize.shgo on random piecewise-constant functions; the callers run it once per plan and use shgo for everything if it fails.
"""
import ctypes as _C

import numpy as np

OK, FAILED, FALLBACK, OK_TIED = 0, 1, 2, 3     # SFE_SHGO_* of sonarfe.h
MAX_POOL = 32                 # the C routine's bound on the minimiser pool (beyond: FALLBACK)


def lowest_result(costs):
    """`LMapCache.sort_cache_result`: which local result shgo reports -- np.argsort(...)[0] over the int64 costs in the order the
    pool was minimised.  Among EQUAL lowest costs the choice is numpy's sort kernel's (not the first: the SIMD sorts are not
    stable), so numpy itself is asked whenever there is a tie."""
    low = np.nonzero(costs == costs.min())[0]
    return int(low[0]) if len(low) == 1 else int(np.argsort(costs)[0])


class SobolPlan:
    def __init__(self, bounds, n, ftol):
        from scipy.optimize import minimize
        from scipy.optimize._shgo import SHGO
        self.bounds = np.array(bounds, float)
        self.n, self.ftol = int(n), float(ftol)
        shc = SHGO(lambda x: 0.0, self.bounds, n=self.n, iters=1, sampling_method="sobol",
                   minimizer_kwargs={"options": {"ftol": self.ftol}})
        with shc:
            shc.iterate_all()
        keys = list(shc.HC.V.cache)                       # the order SHGO.minimizers walks
        index = {k: i for i, k in enumerate(keys)}
        self.X = np.array([shc.HC.V[k].x_a for k in keys], float)
        self.V = len(keys)
        nn = [sorted(index[v.x] for v in shc.HC.V[k].nn) for k in keys]
        self.nn_off = np.zeros(self.V + 1, np.int32)
        self.nn_off[1:] = np.cumsum([len(x) for x in nn])
        self.nn_idx = np.array([j for x in nn for j in x], np.int32)
        # the points SLSQP asks for from each vertex when the gradient vanishes: taken from SLSQP itself
        cb = [[b[0], b[1]] for b in self.bounds]
        pts = np.zeros((self.V, 4, 3))
        for v in range(self.V):
            asked = []

            def const(x):
                asked.append(np.array(x, float))
                return -1
            r = minimize(const, self.X[v], method="SLSQP", bounds=cb, options={"ftol": self.ftol})
            if not (len(asked) == 4 and r.success and np.array_equal(r.x, self.X[v]) and np.array_equal(asked[0], self.X[v])):
                raise RuntimeError("SLSQP on a constant function does not stop at its start: this scipy is not the one "
                                   "shgo_fast was written against")
            pts[v] = asked
        self.points = pts                                   # [V x 4 x 3]: vertex, +h e_x, +h e_y, +h e_theta
        self.checked = None

    # ---- one problem, plain Python: the definition the C routine is tested against ----
    def solve(self, table, return_pool=False):
        """table [V x 4] costs at self.points -> (status, x [3], fun[, number of local minimisations]).  FAILED = shgo's "Failed to
        find a feasible minimizer point" (no vertex strictly below all its neighbours): x / fun are the lowest vertex like
        shgo's result then."""
        r = self._solve(table)
        return r if return_pool else r[:3]

    def _solve(self, table):
        table = np.asarray(table)
        f = table[:, 0]
        pool = [v for v in range(self.V)
                if all(f[v] < f[j] for j in self.nn_idx[self.nn_off[v]:self.nn_off[v + 1]])]
        if not pool:
            v = int(np.argmin(f))                            # first lowest in cache order (find_lowest_vertex: strict <)
            return FAILED, self.X[v].copy(), f[v], 0
        if any((table[v, 1:] != f[v]).any() for v in pool):
            return FALLBACK, None, None, 0
        order = [pool[0]]
        rest = pool[1:]
        while rest:
            d = np.sqrt(((self.X[rest] - self.X[order[-1]]) ** 2).sum(axis=1))   # decided by exact ties only: see below
            far = np.nonzero(d == d.max())[0]
            if len(far) > 1:
                return FALLBACK, None, None, 0               # np.argsort's choice among equal distances is the platform's
            order.append(rest.pop(int(far[0])))
        v = order[lowest_result(np.array([f[v] for v in order], np.int64))]
        return OK, self.X[v].copy(), f[v], len(order)

    # ---- many problems at once (C: sfe_shgo_sobol_replay) ----
    def solve_many(self, lib, tables):
        """tables [S x V x 4] int32 -> (status [S] of OK / FAILED / FALLBACK, vertex [S]: index into self.X of the result, -1 for
        FALLBACK).  Equal lowest costs among a problem's local results are resolved here, by np.argsort like shgo."""
        tables = np.ascontiguousarray(tables, np.int32)
        S = tables.shape[0]
        assert tables.shape[1:] == (self.V, 4)
        status = np.zeros(S, np.uint8)
        vertex = np.zeros(S, np.int32)
        n_order = np.zeros(S, np.int32)
        order = np.zeros((S, MAX_POOL), np.int32)
        i32, f64 = _C.POINTER(_C.c_int32), _C.POINTER(_C.c_double)
        X = np.ascontiguousarray(self.X)
        rc = lib.sfe_shgo_sobol_replay(self.V, self.nn_off.ctypes.data_as(i32), self.nn_idx.ctypes.data_as(i32), X.ctypes.data_as(f64),
                                       tables.ctypes.data_as(i32), S, status.ctypes.data_as(_C.POINTER(_C.c_uint8)),
                                       vertex.ctypes.data_as(i32), n_order.ctypes.data_as(i32), order.ctypes.data_as(i32))
        if rc != 0:
            raise RuntimeError("sfe_shgo_sobol_replay: %d" % rc)
        for s in np.nonzero(status == OK_TIED)[0]:
            o = order[s, :n_order[s]]
            vertex[s] = o[lowest_result(tables[s, o, 0].astype(np.int64))]
            status[s] = OK
        return status, vertex

    # ---- the replay against the installed scipy ----
    def self_check(self, rounds=24, seed=0):
        """random piecewise-constant functions: scipy.optimize.shgo vs solve().  -> True when every decided problem agrees"""
        from scipy.optimize import shgo
        rng = np.random.default_rng(seed)
        span = self.bounds[:, 1] - self.bounds[:, 0]
        good, decided = True, 0
        for r in range(rounds):
            f = piecewise_constant(rng, span, coarse=(r % 3 == 0))
            table = np.array([[f(p) for p in row] for row in self.points])
            st, x, fun = self.solve(table)
            res = shgo(func=f, bounds=self.bounds, n=self.n, iters=1, sampling_method="sobol",
                       minimizer_kwargs={"options": {"ftol": self.ftol}})
            if st == FALLBACK:
                continue
            decided += 1
            if st == OK:
                good &= bool(res.success) and np.array_equal(res.x, x) and res.fun == fun
            else:
                good &= (not res.success) and np.array_equal(res.x, x) and res.fun == fun
        self.checked = bool(good and decided >= rounds // 2)
        return self.checked


def piecewise_constant(rng, span, coarse=False, n_planes=24):
    """an integer-valued function made of steps across random planes (the matching cost's kind of function)"""
    a = rng.normal(0, 1, (n_planes, 3)) / span
    b = rng.uniform(0, 1, n_planes)
    w = rng.uniform(0.02, 0.2, n_planes) * (6.0 if coarse else 1.0)
    m = rng.integers(2, 9, n_planes)

    def f(x):
        q = np.floor((a @ np.asarray(x, float) + b) / w).astype(np.int64)
        return np.int64(-np.sum(q % m))
    return f


_PLANS = {}


def plan_for(bounds, n, ftol):
    """plans are a function of (bounds, n, ftol): one per process"""
    key = (np.asarray(bounds, float).tobytes(), int(n), float(ftol))
    p = _PLANS.get(key)
    if p is None:
        p = _PLANS[key] = SobolPlan(bounds, n, ftol)
    return p
