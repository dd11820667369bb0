"""scipy.optimize.shgo for many scan matches at once, on the host cores.

The global initialisation of a scan match (slam.py:665-716) is ``shgo`` over the matching cost.  The cost function runs on
the GPU (matching_cost.py); shgo itself -- Sobol points, Delaunay triangulation, minimiser pool, SLSQP set-up -- is ~25 ms of
single-threaded Python per call, and ``chained.SessionBatch`` has S of them per step.  They are independent, so they go to a
pool of worker PROCESSES (spawned, not forked: a worker never sees the parent's HIP state and never touches a GPU).

A worker cannot evaluate the cost, so it runs shgo SPECULATIVELY on what the parent already knows: the costs of the first
sampling stage's poses, scored for all sessions in one launch (the same 61 poses for every session: the bounds are the
odometry sigmas).  The few further poses shgo asks for are the local minimiser's finite-difference neighbours of sampled
poses, 1.5e-8 away; the cost is a count of grid cells, piecewise constant, so such a pose is ASSUMED to cost what its nearest
known pose costs, and every assumption is recorded.  The parent then scores all recorded poses of all sessions in ONE launch
and compares: if every assumption holds, the run saw exactly the values a run on the true function would have seen -- shgo is
deterministic -- and its result IS that run's result; a session with a wrong assumption (not seen so far) is simply solved
again in the parent with the true function.  Results are therefore identical to the in-process path
(tests/test_global_init.py).
"""
import numpy as np


def solve(job):
    """one speculative shgo run: job = (bounds, params, X0 [m x 3], costs [m]) -> (success, x, fun, message, asked [k x 3],
    assumed [k])"""
    from scipy.optimize import shgo
    bounds, params, X0, costs = job
    X0 = np.asarray(X0, np.float64)
    table = {X0[i].tobytes(): costs[i] for i in range(len(X0))}
    asked, assumed = [], []

    def f(x):
        x = np.asarray(x, np.float64)
        key = x.tobytes()
        v = table.get(key)
        if v is None:
            d = X0 - x
            v = costs[int(np.argmin(np.einsum("ij,ij->i", d, d)))]
            table[key] = v
            asked.append(x.copy())
            assumed.append(v)
        return np.int64(v)
    res = shgo(func=f, bounds=bounds, n=params[0], iters=params[1], sampling_method="sobol",
               minimizer_kwargs={"options": {"ftol": params[2]}}, workers=lambda _fn, pts: [f(p) for p in pts])
    return (bool(res.success), np.asarray(res.x, np.float64) if res.success else np.zeros(3), float(res.fun) if res.success else 0.0,
            str(res.message), np.array(asked, np.float64).reshape(-1, 3), np.array(assumed, np.int64))


class ShgoPool(object):
    """`workers` long-lived Python processes (``python -m sonar_slam_amd.shgo_pool``), jobs and results as length-prefixed
    pickles over their pipes.  Plain subprocesses rather than a multiprocessing pool: nothing of the parent is inherited or
    re-imported (no HIP state, no ``__main__``), whatever program the parent is."""

    def __init__(self, workers):
        import os
        import subprocess
        import sys
        self.workers = max(1, int(workers))
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        env = dict(os.environ)
        env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
        for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
            env[k] = "1"                                 # one core per worker: the pool is the parallelism
        self.procs = [subprocess.Popen([sys.executable, "-m", "sonar_slam_amd.shgo_pool"], stdin=subprocess.PIPE,
                                       stdout=subprocess.PIPE, env=env) for _ in range(self.workers)]
        for p in self.procs:                             # scipy imported in every worker before anything is timed
            _send(p.stdin, [])
        for p in self.procs:
            _recv(p.stdout)

    def map(self, jobs):
        n, w = len(jobs), self.workers
        parts = [list(range(k, n, w)) for k in range(w)]
        for p, idx in zip(self.procs, parts):
            _send(p.stdin, [jobs[i] for i in idx])
        out = [None] * n
        for p, idx in zip(self.procs, parts):
            for i, r in zip(idx, _recv(p.stdout)):
                out[i] = r
        return out

    def close(self):
        for p in getattr(self, "procs", []):
            try:
                p.stdin.close()
                p.wait(timeout=5)
            except Exception:
                p.kill()
        self.procs = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _send(f, obj):
    import pickle
    import struct
    b = pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL)
    f.write(struct.pack("<q", len(b)))
    f.write(b)
    f.flush()


def _recv(f):
    import pickle
    import struct
    h = f.read(8)
    if len(h) < 8:
        raise EOFError("shgo worker closed its pipe")
    (n,) = struct.unpack("<q", h)
    return pickle.loads(f.read(n))


def _worker():
    import sys
    import scipy.optimize  # noqa: F401
    import scipy.spatial  # noqa: F401
    fin, fout = sys.stdin.buffer, sys.stdout.buffer
    sys.stdout = sys.stderr                             # (nothing but results on the pipe)
    while True:
        try:
            jobs = _recv(fin)
        except EOFError:
            return
        _send(fout, [solve(j) for j in jobs])


if __name__ == "__main__":
    _worker()
