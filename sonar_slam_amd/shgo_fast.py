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
would -- a finite-difference point whose cost differs from its vertex's (SLSQP then moves) -- is REPORTED (`FALLBACK`) and the caller
runs scipy.optimize.shgo itself for that problem.  Where shgo's own choice is numpy's sort kernel's -- equal lowest costs among the
local results, pool members exactly equally far from the last result -- numpy is asked with the same call on the same numbers.  `SobolPlan.self_check` compares the replay with
scipy.optimize.shgo on random piecewise-constant functions; the callers run it once per plan and use shgo for everything if it fails.
"""
import collections
import ctypes as _C
import logging

import numpy as np

OK, FAILED, FALLBACK, OK_TIED = 0, 1, 2, 3     # SFE_SHGO_* of sonarfe.h
DEVELOPED_AGAINST = "1.15.3"   # the scipy whose shgo / SLSQP internals the replays were written against

# ---- what a maintainer must be able to SEE: the replays lean on scipy internals and switch themselves off
# when the installed scipy does not reproduce them -- correct, but ~100 x slower, and silent until round 6.  Every self-check now
# ends in one log line (WARNING when a replay is off, INFO when it is on) and in `status()`, which bench.py prints into its line.
_LOG = logging.getLogger("sonar_slam_amd.shgo_fast")
_STATUS = {}


def _announce(kind, key, ok, why=""):
    import scipy
    _STATUS["%s %s" % (kind, key)] = {"active": bool(ok), "why": why}
    if ok:
        _LOG.info("shgo replay (%s, %s) ACTIVE: reproduces scipy %s's shgo on random step functions (developed against %s)",
                  kind, key, scipy.__version__, DEVELOPED_AGAINST)
    else:
        _LOG.warning("shgo replay (%s, %s) OFF -- scipy %s does not reproduce what shgo_fast.py was written against (%s)%s: every "
                     "global initialisation goes through scipy.optimize.shgo itself (same results, ~100 x slower)",
                     kind, key, scipy.__version__, DEVELOPED_AGAINST, (": " + why) if why else "")


def status():
    """-> {"scipy": installed version, "developed_against": ..., "replays": {name: {"active": bool, "why": str}}} for every replay
    this process has checked so far (none yet: nothing has asked for one)"""
    import scipy
    return {"scipy": scipy.__version__, "developed_against": DEVELOPED_AGAINST, "replays": dict(_STATUS)}
MAX_POOL = 32                 # the C routine's bound on the minimiser pool (beyond: FALLBACK)


def lowest_result(costs):
    """`LMapCache.sort_cache_result`: which local result shgo reports -- np.argsort(...)[0] over the int64 costs in the order the
    pool was minimised.  Among EQUAL lowest costs the choice is numpy's sort kernel's (not the first: the SIMD sorts are not
    stable), so numpy itself is asked whenever there is a tie."""
    low = np.nonzero(costs == costs.min())[0]
    return int(low[0]) if len(low) == 1 else int(np.argsort(costs)[0])


def farthest(x, rest_points):
    """`SHGO.g_topograph`: which of the remaining minimisers is minimised next -- the LAST entry of np.argsort over
    scipy.spatial.distance.cdist from the last local result.  Among exactly equal distances (Sobol points are dyadic: it happens)
    the last one is the sort kernel's choice, so the same two calls are made on the same numbers."""
    from scipy.spatial.distance import cdist
    return int(np.argsort(cdist(np.array([x]), rest_points, "euclidean"), axis=-1)[0, -1])


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
        return r[:4] if return_pool else r[:3]

    def solve_order(self, table):
        """-> (status, x, fun, the vertices shgo minimises from, in its order) -- what the caller needs to enter the evaluations
        of the reference's run into its pose samples: every vertex once, four points per local minimisation"""
        st, x, fun, _, order = self._solve(table)
        return st, x, fun, order

    def _solve(self, table):
        table = np.asarray(table)
        f = table[:, 0]
        pool = [v for v in range(self.V)
                if all(f[v] < f[j] for j in self.nn_idx[self.nn_off[v]:self.nn_off[v + 1]])]
        if not pool:
            v = int(np.argmin(f))                            # first lowest in cache order (find_lowest_vertex: strict <)
            return FAILED, self.X[v].copy(), f[v], 0, []
        if any((table[v, 1:] != f[v]).any() for v in pool):
            return FALLBACK, None, None, 0, []
        order = [pool[0]]
        rest = pool[1:]
        while rest:
            order.append(rest.pop(farthest(self.X[order[-1]], self.X[rest])))
        v = order[lowest_result(np.array([f[v] for v in order], np.int64))]
        return OK, self.X[v].copy(), f[v], len(order), order

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
        # the C routine also hands back exact distance ties: the Python definition decides those with numpy's own argsort
        for s in np.nonzero(status == FALLBACK)[0]:
            st, x, _ = self.solve(tables[s])
            if st != FALLBACK:
                status[s] = st
                vertex[s] = int(np.nonzero((self.X == x).all(axis=1))[0][0])
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
        _announce("one iteration", "n=%d bounds=%s" % (self.n, np.array2string(self.bounds[:, 1], precision=4)), self.checked,
                  "" if self.checked else "%d of %d random problems decided, agreement %s" % (decided, rounds, good))
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


_PLANS = collections.OrderedDict()
MAX_PLANS = 8   # (a caller whose bounds change from call to call -- the loop-closure search with (n, 1) parameters takes them from a
#                 keyframe's covariance -- would otherwise keep every plan it ever built: ADVICE r5)


def plan_for(bounds, n, ftol):
    """plans are a function of (bounds, n, ftol): the most recently used MAX_PLANS of them are kept per process"""
    key = (np.asarray(bounds, float).tobytes(), int(n), float(ftol))
    p = _PLANS.get(key)
    if p is None:
        try:
            p = SobolPlan(bounds, n, ftol)
        except Exception as e:       # a scipy without these internals, or one whose SLSQP behaves differently: no replay, shgo itself
            p = _NoPlan(repr(e))
            _announce("one iteration", "n=%d" % int(n), False, p.why)
        _PLANS[key] = p
    else:
        _PLANS.move_to_end(key)
    while len(_PLANS) > MAX_PLANS:
        _PLANS.popitem(last=False)
    return p


class _NoPlan:
    """what plan_for hands out when the plan cannot be taken from the installed scipy: never `checked`, so every caller runs
    scipy.optimize.shgo itself"""
    checked = False

    def __init__(self, why):
        self.why = why

    def self_check(self, *a, **k):
        return False


# ---------------------------------------------------------------------------------------------------------------------------------
# several iterations (the loop-closure search: shgo(n=100, iters=5), slam.py:952-961)
# ---------------------------------------------------------------------------------------------------------------------------------
# With iters > 1 the triangulation is no longer a function of (bounds, n): every iteration draws 128 k NEW Sobol points, appends the
# local results found so far (duplicates of vertices), and hands `Tri.add_points` the rows behind the previous iteration's row count
# -- so which of the new points enter depends on how many minimisers earlier iterations found.  What stays true for a piecewise-constant
# cost: every point that CAN become a vertex is known beforehand (the rows [128 (k - 1), 128 k) of draw k: 640 points for (100, 5)),
# so their costs and those of their finite-difference points are taken in ONE launch, and the iterations are then replayed on the
# host: scipy.spatial.Delaunay itself for the incremental triangulation (the same calls on the same arrays), the vertex-vertex
# graph, the vertex-cache order, the minimiser pool and its order, the local results and the final argsort restated from
# scipy/optimize/_shgo.py -- vectorised where shgo walks Python objects (its `minimizers` and `vf_to_vv` are 85 % of a 150 ms call).
FD_STEP = 1.4901161193847656e-08        # SLSQP's forward-difference step (scipy/optimize/_slsqp_py.py: _epsilon = sqrt(eps))
_DRAWS = {}


def _sobol_draws(n, iters, dim=3):
    """the unit-cube points draw k of shgo's engine returns (draw k has n2 * k rows; the sequence continues over the draws)"""
    from scipy.stats import qmc
    key = (int(n), int(iters), int(dim))
    if key not in _DRAWS:
        n2 = int(2 ** np.ceil(np.log2(n)))
        engine = qmc.Sobol(d=dim, scramble=False, seed=0)
        _DRAWS[key] = (n2, [engine.random(n2 * (k + 1)) for k in range(iters)])
    return _DRAWS[key]


def multi_candidates(bounds, n, iters):
    """-> (draws: the scaled sample arrays of every iteration, cand [M x 3]: every point that can become a vertex, fd [M x 3 x 3])"""
    bounds = np.array(bounds, float)
    n2, units = _sobol_draws(n, iters, len(bounds))
    draws = []
    for U in units:
        C = U.copy()
        for i in range(len(bounds)):                                   # SHGO.sampling_custom, operation for operation
            C[:, i] = (C[:, i] * (bounds[i][1] - bounds[i][0]) + bounds[i][0])
        draws.append(C)
    cand = np.concatenate([C[n2 * k:] for k, C in enumerate(draws)])
    fd = np.repeat(cand[:, None, :], 3, axis=1)
    for i in range(3):
        fd[:, i, i] = cand[:, i] + FD_STEP
    return draws, cand, fd


def replay_multi(bounds, n, iters, draws, cand, cost, fd_cost):
    """shgo(sobol, iters) after the evaluations: cand / cost [M] / fd_cost [M x 3] as multi_candidates laid them out.
    -> (status, x, fun, vertices: indices into cand of the vertices shgo creates, minimised: indices of the starts it minimises, in
    order).  FALLBACK: a start whose finite-difference points cost something else, a point too close to a bound."""
    from scipy import spatial
    bounds = np.array(bounds, float)
    n2 = len(draws[0])
    cost = np.asarray(cost, np.int64)
    if np.any(bounds[:, 1] - cand.max(axis=0) <= 4 * FD_STEP):
        return FALLBACK, None, None, None, None          # (SLSQP would difference backwards there)
    index = {row.tobytes(): i for i, row in enumerate(cand)}
    M = len(cand)
    created = np.zeros(M, bool)
    cache = []                                           # vertex-cache order (indices into cand)
    nbr_min = np.full(M, np.iinfo(np.int64).max, np.int64)   # lowest cost among a vertex's neighbours (edges only ever add up)
    xl, fl = [], []                                      # local results in the order they were found (indices, costs)
    done = np.zeros(M, bool)                             # vertices that are local results already (SHGO.minimizers skips them)
    tri, n_prc, pid = None, 0, np.zeros(0, np.int64)
    for k in range(iters):
        C = draws[k]
        if xl:
            C = np.vstack((C, cand[xl]))
        if tri is None:
            tri = spatial.Delaunay(C, incremental=True)
        else:
            tri.add_points(C[n_prc:, :])
        n_prc = C.shape[0]
        pts = tri.points
        if len(pts) > len(pid):                          # cand index of every row of the triangulation
            new = [index.get(row.tobytes(), -1) for row in pts[len(pid):]]
            pid = np.concatenate([pid, np.array(new, np.int64)])
            if (pid < 0).any():
                return FALLBACK, None, None, None, None  # a triangulated point outside the candidate set: not this restatement's case
        s = pid[tri.simplices[:, :3]]                    # vf_to_vv touches vertices 0, 1, 2 of a simplex, never the fourth
        flat = s.reshape(-1)
        uniq, first = np.unique(flat, return_index=True)
        fresh = uniq[~created[uniq]]
        fresh = fresh[np.argsort(first[~created[uniq]], kind="stable")]
        created[fresh] = True
        cache.extend(int(v) for v in fresh)
        for a, b in ((0, 1), (0, 2), (1, 2)):
            u, v = s[:, a], s[:, b]
            ok = u != v
            np.minimum.at(nbr_min, u[ok], cost[v[ok]])
            np.minimum.at(nbr_min, v[ok], cost[u[ok]])
        # SHGO.minimizers: vertices strictly below all their neighbours, in cache order, local results skipped
        order = np.array(cache, np.int64)
        is_min = (cost[order] < nbr_min[order]) & ~done[order]
        pool = [int(v) for v in order[is_min]]
        if not pool:
            continue
        if any((fd_cost[v] != cost[v]).any() for v in pool):
            return FALLBACK, None, None, None, None
        seq = [pool[0]]
        rest = pool[1:]
        while rest:
            seq.append(rest.pop(farthest(cand[seq[-1]], cand[rest])))
        for v in seq:
            xl.append(v)
            fl.append(cost[v])
            done[v] = True
    vertices = np.array(cache, np.int64)
    if not xl:
        v = int(vertices[np.argmin(cost[vertices])])     # find_lowest_vertex: the first lowest in cache order
        return FAILED, cand[v].copy(), cost[v], vertices, np.zeros(0, np.int64)
    v = xl[lowest_result(np.array(fl, np.int64))]
    return OK, cand[v].copy(), cost[v], vertices, np.array(xl, np.int64)


_MULTI_OK = {}


def multi_checked(n, iters, ftol, rounds=3, seed=5):
    """replay_multi against the installed scipy.optimize.shgo on a few random step functions, once per process and (n, iters):
    result, value, success and the multiset of evaluated points must agree (problems the replay hands back do not count)"""
    key = (int(n), int(iters), float(ftol))
    if key not in _MULTI_OK:
        why = ""
        try:
            _MULTI_OK[key] = _multi_check(n, iters, ftol, rounds, seed)
        except Exception as e:       # (a scipy without the pieces the replay leans on: shgo itself)
            _MULTI_OK[key] = False
            why = repr(e)
        _announce("%d iterations" % int(iters), "n=%d" % int(n), _MULTI_OK[key], why)
    return _MULTI_OK[key]


def _multi_check(n, iters, ftol, rounds, seed):
    from scipy.optimize import shgo
    rng = np.random.default_rng(seed)
    good, decided = True, 0
    for r in range(rounds + 3):
        if decided >= rounds:
            break
        stds = np.array([[rng.uniform(0.2, 3), rng.uniform(0.2, 3), rng.uniform(0.01, 0.3)]]).T
        bounds = 5.0 * np.c_[-stds, stds]
        f = piecewise_constant(rng, bounds[:, 1] - bounds[:, 0], coarse=(r == 1))
        draws, cand, fd = multi_candidates(bounds, n, iters)
        cost = np.array([f(p) for p in cand])
        fd_cost = np.array([[f(p) for p in row] for row in fd])
        st, x, fun, vertices, minimised = replay_multi(bounds, n, iters, draws, cand, cost, fd_cost)
        if st == FALLBACK:
            continue
        asked = []

        def g(p):
            asked.append(tuple(np.asarray(p, float)))
            return f(p)
        res = shgo(func=g, bounds=bounds, n=n, iters=iters, sampling_method="sobol", minimizer_kwargs={"options": {"ftol": ftol}})
        mine = [tuple(cand[v]) for v in vertices]
        for v in minimised:
            mine.append(tuple(cand[v]))
            mine.extend(tuple(p) for p in fd[v])
        good &= bool(res.success) == (st == OK) and np.array_equal(res.x, x) and res.fun == fun and sorted(mine) == sorted(asked)
        decided += 1
    return bool(good and decided >= 1)
