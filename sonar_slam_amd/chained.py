"""Many independent SLAM sessions advanced in lock-step, every cloud device-resident.

One session is what ``replay.FrontEnd`` runs: ping -> FeatureExtraction.callback -> keyframe test -> target cloud
= get_points(last 3 keyframes) -> ICP(source, target, odometry guess) -> sanity checks + overlap -> pose
(slam_ros.py:157-213, slam.py:716-832).  Inside a session the keyframes are strictly sequential -- keyframe k's
target cloud needs the poses the scan matches of k-1, k-2, k-3 produced -- so the batch axis is the SESSION: S
trajectories (robots, bags, replays) step together, step k = keyframe k of every session:

    CFAR + gate -> remap + nonzero + px->m -> downsample -> outlier filter      sfe_*_batch_dev, S pings
    -> append to the keyframe store (slam_ros.py:170 convention)                 sfe_cloud_store_put_batch_dev
    -> S target clouds: transform + concatenate + pcl.downsample                 sfe_cloud_store_get_points
    -> S scan matches over handles                                               sfe_icp_store_compute
    -> S overlap counts                                                          sfe_cloud_store_overlap

The host does what the SLAM node's Python does between those calls -- gtsam.Pose2 algebra in double, the
ssm_min_points / max translation / max rotation / overlap tests, the factor list -- on S sessions at a time, with
exactly the arithmetic of ``pose2.Pose2`` (so a session's records equal those of ``replay.FrontEnd`` on the same
pings bit for bit: tests/test_gpu_store.py).  Per step three small synchronisations (cloud sizes; scan-match
results; overlap counts); no cloud crosses PCIe.

``initialization=True`` adds the reference's default step in front of every scan match (slam.py:77, :665-716): the
global initialisation by ``scipy.optimize.shgo`` over the matching cost.  shgo itself is host Python per session (its
Delaunay / minimiser-pool bookkeeping: tens of milliseconds per call, far more than the scoring); what the device
takes is the cost function: the target grids of all S sessions are built in one launch, shgo's sampling points -- the
same Sobol set for every session, since the bounds are the odometry sigmas -- are scored for all sessions in one more
(``matching_cost.batch_store``), and each session's shgo then runs on that table; the handful of further points its
local minimiser asks for (finite-difference neighbours) are scored one call each.
"""
import math

import numpy as np

from . import _lib as _L
from . import store as _store


class Pose2Batch(object):
    """``pose2.Pose2`` for n poses at once: the same double-precision expressions, element by element (cos / sin /
    atan2 through ``math`` per element, so that no vector math library rounds differently from the scalar class)."""

    __slots__ = ("x", "y", "c", "s")

    def __init__(self, x, y, theta=None, cs=None):
        self.x, self.y = np.array(x, np.float64), np.array(y, np.float64)
        if cs is None:
            th = np.asarray(theta, np.float64)
            self.c = np.array([math.cos(t) for t in th])
            self.s = np.array([math.sin(t) for t in th])
        else:
            c, s = np.array(cs[0], np.float64), np.array(cs[1], np.float64)
            scale = c * c + s * s
            fix = np.abs(scale - 1.0) > 1e-10
            if fix.any():
                with np.errstate(invalid="ignore", divide="ignore"):
                    k = 1.0 / np.sqrt(scale)
                c, s = np.where(fix, c * k, c), np.where(fix, s * k, s)
            self.c, self.s = c, s

    def __len__(self):
        return len(self.x)

    def theta(self):
        return np.array([math.atan2(s, c) for s, c in zip(self.s, self.c)])

    def compose(self, o):
        return Pose2Batch(self.x + self.c * o.x - self.s * o.y, self.y + self.s * o.x + self.c * o.y,
                          cs=(self.c * o.c - self.s * o.s, self.s * o.c + self.c * o.s))

    def inverse(self):
        return Pose2Batch(-(self.c * self.x + self.s * self.y), -(-self.s * self.x + self.c * self.y),
                          cs=(self.c, -self.s))

    def between(self, o):
        return self.inverse().compose(o)

    def T6(self):
        """[n x 6] float32: T00 T01 T02 T10 T11 T12 of ``matrix().astype(np.float32)``"""
        return np.stack([self.c, -self.s, self.x, self.s, self.c, self.y], axis=1).astype(np.float32)

    def matrix32(self):
        """[n x 3 x 3] float32 = ``matrix()`` handed to pybind (the ICP guess, slam.py:316)"""
        M = np.zeros((len(self), 3, 3), np.float64)
        M[:, 0, 0], M[:, 0, 1], M[:, 0, 2] = self.c, -self.s, self.x
        M[:, 1, 0], M[:, 1, 1], M[:, 1, 2] = self.s, self.c, self.y
        M[:, 2, 2] = 1.0
        return M.astype(np.float32)

    def take(self, idx):
        return Pose2Batch(self.x[idx], self.y[idx], cs=(self.c[idx], self.s[idx]))

    def put(self, idx, o):
        self.x[idx], self.y[idx], self.c[idx], self.s[idx] = o.x, o.y, o.c, o.s

    def xytheta(self):
        return np.stack([self.x, self.y, self.theta()], axis=1)


def sample_transforms(lib, target, source, X):
    """T6 [n x len(X) x 6] float32 of target_i.between(source_i.compose(Pose2(*x))) for Pose2Batch target / source (n poses each)
    and deltas X [P x 3]: the cost function's sample transforms (slam.py:548-550) of n sessions at once.  The arithmetic is
    Pose2's, done by the library's host routine (sfe_pose2_sample_transforms); cos / sin of the deltas through ``math`` like Pose2."""
    import ctypes as _C
    X = np.asarray(X, np.float64).reshape(-1, 3)
    n, P = len(target), len(X)
    d4 = np.ascontiguousarray(np.stack([X[:, 0], X[:, 1], [math.cos(t) for t in X[:, 2]], [math.sin(t) for t in X[:, 2]]], axis=1)) \
        if P else np.zeros((0, 4))
    t4 = np.ascontiguousarray(np.stack([target.x, target.y, target.c, target.s], axis=1), np.float64)
    s4 = np.ascontiguousarray(np.stack([source.x, source.y, source.c, source.s], axis=1), np.float64)
    out = np.zeros((n, P, 6), np.float32)
    f64 = _C.POINTER(_C.c_double)
    rc = lib.sfe_pose2_sample_transforms(t4.ctypes.data_as(f64), s4.ctypes.data_as(f64), n, d4.ctypes.data_as(f64), P,
                                         out.ctypes.data_as(_C.POINTER(_C.c_float)))
    if rc != 0:
        raise _L.SonarFEError("sfe_pose2_sample_transforms: %d" % rc)
    return out


class _View(object):
    """a window into a DeviceBuffer (what KeyframeBatch reads as .ptr)"""

    def __init__(self, buf, offset):
        import ctypes as _C
        self.ptr = _C.c_void_p(buf.ptr.value + int(offset))


class SessionBatch(object):
    """S sessions x K pings, all pings resident in HBM.  ``step(k)`` advances every session by its k-th ping."""

    def __init__(self, ctx, geometry, cfar_params, alg, intensity_thr, icp_params, n_sessions, n_steps, dr_poses,
                 max_points=16384, resolution=0.5, outlier_radius=1.0, outlier_min_points=5, point_resolution=0.5,
                 point_noise=0.5, ssm_min_points=50, ssm_max_translation=3.0, ssm_max_rotation=np.deg2rad(30),
                 ssm_target_frames=3, store_points=None, initialization=False, initialization_params=(50, 1, 0.01),
                 odom_sigmas=(0.2, 0.2, 0.02), shgo_workers=1, shgo_replay=True):
        from .pipeline import KeyframeBatch
        self.ctx, self.S, self.K = ctx, int(n_sessions), int(n_steps)
        self.icp_params = icp_params
        self.kb = KeyframeBatch(ctx, geometry, cfar_params, alg, intensity_thr, icp_params, self.S, max_points=max_points)
        self.frame_bytes = self.kb.rows * self.kb.cols
        self._kb_img = self.kb.d_img                     # (kept for free(): the batch's own frame buffer is unused)
        self.d_frames = ctx.alloc(self.K * self.S * self.frame_bytes)
        self.resolution, self.outlier_radius, self.outlier_min_points = resolution, outlier_radius, outlier_min_points
        self.point_resolution, self.point_noise = point_resolution, point_noise
        self.ssm_min_points, self.ssm_max_translation = ssm_min_points, ssm_max_translation
        self.ssm_max_rotation, self.ssm_target_frames = ssm_max_rotation, ssm_target_frames
        # keyframe clouds of K steps + the targets of one step (3 keyframes each before the downsample shrinks them)
        pts = store_points or int(self.S * (self.K + 4) * 2048)
        self.store = _store.CloudStore(ctx, capacity_points=pts, max_clouds=self.S * (self.K + 2))
        dr = np.asarray(dr_poses, np.float64).reshape(self.S, self.K, 3)
        self.dr = [Pose2Batch(dr[:, k, 0], dr[:, k, 1], dr[:, k, 2]) for k in range(self.K)]
        self.max_raw = 0
        self.initialization, self.initialization_params = initialization, tuple(initialization_params)
        self.odom_sigmas = np.array(odom_sigmas, np.float64)
        self._sobol = None
        self.shgo_workers, self._shgo_pool = int(shgo_workers), None    # > 1: shgo_pool.ShgoPool (host processes)
        # shgo_replay: sonar_slam_amd/shgo_fast.py -- what shgo decides after its sampling stage, replayed for all sessions from
        # one table of costs; scipy.optimize.shgo itself only for the sessions the replay reports as undecidable
        self.shgo_replay, self._plan = bool(shgo_replay), None
        self.init_stats = {"shgo_s": 0.0, "cost_calls": 0, "table_hits": 0, "speculated": 0, "speculation_failed": 0,
                           "replayed": 0, "replay_fallbacks": 0, "transforms_s": 0.0, "table_s": 0.0, "grids_s": 0.0}
        self.reset()

    def upload_frames(self, k, frames):
        """pings of step k: [S x rows x cols] uint8"""
        frames = np.ascontiguousarray(frames, np.uint8)
        assert frames.shape == (self.S, self.kb.rows, self.kb.cols)
        self.d_frames.upload(frames, offset=k * self.S * self.frame_bytes)

    def fit_capacity(self):
        """After a first (untimed) run: size the per-ping point capacity by what the pings really hold.  The resident
        downsample's LDS is sized by the capacity -- 128 KB (one frame per CU) at 16 384 points, 64 KB (two per CU) at
        8 192 -- so sessions whose pings stay below 8 192 raw detections run their filters twice as dense."""
        from .pipeline import KeyframeBatch
        if self.max_raw <= 0 or self.max_raw > 7800 or self.kb.cap <= 8192:
            return self.kb.cap
        old = self.kb
        old.d_img = self._kb_img
        self.kb = KeyframeBatch(self.ctx, old.geom, (old.train_hs, old.guard_hs, old.tau), "SOCA", 0, self.icp_params, self.S,
                                max_points=8192)
        self.kb.alg, self.kb.k, self.kb.intensity_thr = old.alg, old.k, old.intensity_thr
        self._kb_img = self.kb.d_img
        old.free()
        return self.kb.cap

    def reset(self):
        self.store.truncate(0)
        self.handles = np.full((self.S, self.K), -1, np.int32)      # keyframe k of session s -> store handle
        self.poses = [None] * self.K                                # Pose2Batch per step (every ping is a keyframe)
        self.records = []

    def step(self, k):
        """-> dict of per-session arrays: status codes, sizes, transforms, overlaps, poses"""
        S, kb, store = self.S, self.kb, self.store
        kb.d_img = _View(self.d_frames, k * S * self.frame_bytes)
        kb.run_cfar()
        kb.run_extract()
        kb.run_filter(self.resolution, self.outlier_radius, self.outlier_min_points)
        src_h = kb.store_clouds(store, stamps=np.arange(S) * self.K + k)
        self.handles[:, k] = src_h
        rec = {"k": k, "status": np.full(S, PRIOR if k == 0 else SUCCESS, np.int8)}
        if k == 0:
            self.poses[0] = self.dr[0]                                      # add_prior (slam.py:426-442)
            rec["n_source"] = store.counts(src_h)
            self._check_raw(k)
            self._check_counts(rec["n_source"], k)
            rec["pose"] = self.poses[0].xytheta()
            self.records.append(rec)
            return rec
        # frame.update(current_keyframe.pose.compose(dr_odom))              slam_ros.py:181-184
        prev = self.poses[k - 1]
        dr_odom = self.dr[k - 1].between(self.dr[k])
        pose = prev.compose(dr_odom)
        # target = get_points(last ssm_target_frames keyframes, target_key = k - 1)   slam.py:740-741
        frames = list(range(k))[-self.ssm_target_frames:]
        m = self.ssm_target_frames
        th = np.full((S, m), -1, np.int32)
        T6 = np.zeros((S, m, 6), np.float32)
        for j, key in enumerate(frames):
            th[:, j] = self.handles[:, key]
            T6[:, j] = prev.between(self.poses[key]).T6()
        n_keep = len(store)
        try:
            return self._scan_match_step(k, rec, src_h, th, T6, prev, pose)
        finally:
            store.truncate(n_keep)                                          # the targets are dropped, the keyframes stay

    def _scan_match_step(self, k, rec, src_h, th, T6, prev, pose):
        S, store = self.S, self.store
        tgt_h = store.get_points(th, T6, self.point_resolution)
        counts = store.counts(np.concatenate([src_h, tgt_h]))
        n_src, n_tgt = counts[:S], counts[S:]
        self._check_raw(k)
        self._check_counts(n_src, k)
        self._check_counts(n_tgt, k)
        rec["n_source"], rec["n_target"] = n_src, n_tgt
        enough = (n_src >= self.ssm_min_points) & (n_tgt >= self.ssm_min_points)
        rec["status"][~enough] = NOT_ENOUGH_POINTS
        dr_between = prev.between(pose)
        initial = prev.between(pose)                                        # target_pose.between(keyframe.pose) slam.py:757
        idx = np.nonzero(enough)[0]
        T = np.zeros((S, 3, 3), np.float32)
        icp_status = np.full(S, -1, np.int32)
        iters = np.zeros(S, np.int32)
        overlap = np.full(S, -1, np.int32)
        est = Pose2Batch(dr_between.x.copy(), dr_between.y.copy(), cs=(dr_between.c.copy(), dr_between.s.copy()))
        if self.initialization and len(idx):
            # slam.py:665-716: ICP starts from the pose shgo found; a failed initialisation leaves the odometry factor
            ok_init, est_src, xs, fs = self._global_init(idx, src_h, tgt_h, pose, prev)
            rec["init_success"] = np.zeros(S, bool)
            rec["init_success"][idx] = ok_init
            rec["init_x"], rec["init_cost"] = np.zeros((S, 3)), np.zeros(S)
            rec["init_x"][idx], rec["init_cost"][idx] = xs, fs
            rec["status"][idx[~ok_init]] = INITIALIZATION_FAILURE
            initial.put(idx[ok_init], prev.take(idx[ok_init]).between(est_src.take(np.nonzero(ok_init)[0])))
            idx = idx[ok_init]
        if len(idx):
            pairs = np.stack([src_h[idx], tgt_h[idx]], axis=1)
            Ti, sti, iti = store.icp(self.icp_params, pairs, initial.take(idx).matrix32())
            T[idx], icp_status[idx], iters[idx] = Ti, sti, iti
            # x, y = T[:2, 2]; theta = np.arctan2(T[1, 0], T[0, 0]); gtsam.Pose2(x, y, theta)     slam.py:319-323
            theta32 = np.arctan2(Ti[:, 1, 0], Ti[:, 0, 0])
            e = Pose2Batch(Ti[:, 0, 2].astype(np.float64), Ti[:, 1, 2].astype(np.float64), theta32.astype(np.float64))
            est.put(idx, e)
            rec["status"][idx[sti != 0]] = NOT_CONVERGED
            delta = initial.take(idx).between(e)
            # (the reference: np.linalg.norm(delta.translation()) per scan match; hypot here, for all sessions at once -- the two can
            #  differ in the last bit, i.e. for a translation within 4e-16 m of the 3 m gate)
            large = (np.hypot(delta.x, delta.y) > self.ssm_max_translation) | (np.abs(delta.theta()) > self.ssm_max_rotation)
            rec["status"][idx[(sti == 0) & large]] = LARGE_TRANSFORMATION
            ok = idx[(sti == 0) & ~large]
            if len(ok):
                ov = store.overlap(np.stack([src_h[ok], tgt_h[ok]], axis=1), est.take(ok).T6(), self.point_noise)
                overlap[ok] = ov
                rec["status"][ok[ov < self.ssm_min_points]] = NOT_ENOUGH_OVERLAP
        good = rec["status"] == SUCCESS
        # keyframe.update(target_pose.compose(estimated))   slam.py:825-827; otherwise the dead-reckoned pose stays
        new_pose = Pose2Batch(pose.x.copy(), pose.y.copy(), cs=(pose.c.copy(), pose.s.copy()))
        gi = np.nonzero(good)[0]
        if len(gi):
            new_pose.put(gi, prev.take(gi).compose(est.take(gi)))
        self.poses[k] = new_pose
        rec.update(T=T, icp_status=icp_status, iters=iters, overlap=overlap, transform=est.xytheta(),
                   pose=new_pose.xytheta())
        self.records.append(rec)
        return rec

    # -- global initialisation (slam.py:665-716) for the sessions `idx` --
    def _sobol_points(self, pose_bounds):
        """the points shgo's first sampling stage evaluates for these bounds and parameters: the same for every session
        and every step, so they are asked for once (a dry run of shgo whose `workers` hook stops at the first pool)"""
        from scipy.optimize import shgo
        if self._sobol is None:
            class _Stop(Exception):
                pass
            got = []

            def pool(_fn, xs):
                got.extend(np.asarray(x, np.float64).copy() for x in xs)
                raise _Stop()
            try:
                shgo(func=lambda x: 0.0, bounds=pose_bounds, n=self.initialization_params[0], iters=self.initialization_params[1],
                     sampling_method="sobol", minimizer_kwargs={"options": {"ftol": self.initialization_params[2]}}, workers=pool)
            except _Stop:
                pass
            self._sobol = np.array(got, np.float64).reshape(-1, 3)
        return self._sobol

    def _replay_plan(self, pose_bounds):
        """shgo_fast.SobolPlan for these bounds, or None (replay switched off, more than one shgo iteration, or the replay does
        not reproduce the installed scipy: then every problem goes through scipy.optimize.shgo)"""
        from . import shgo_fast
        if not self.shgo_replay or self.initialization_params[1] != 1:
            return None
        if self._plan is None:
            self._plan = shgo_fast.plan_for(pose_bounds, self.initialization_params[0], self.initialization_params[2])
            if self._plan.checked is None:
                self._plan.self_check()
        return self._plan if self._plan.checked else None

    def warm_up(self):
        """Build and self-check the shgo replay NOW instead of inside the first step (a few seconds of scipy.optimize.shgo on
        random step functions), and say in the log whether it is active for the installed scipy.  -> shgo_fast.status()"""
        from . import shgo_fast
        pose_stds = np.array([self.odom_sigmas]).T
        self._replay_plan(5.0 * np.c_[-pose_stds, pose_stds])
        return shgo_fast.status()

    def _global_init(self, idx, src_h, tgt_h, pose, prev):
        """-> (success [n], estimated source poses Pose2Batch [n], result.x [n x 3], result.fun [n]) for sessions idx"""
        import time
        from scipy.optimize import shgo
        from . import matching_cost as mc
        from . import shgo_fast
        n = len(idx)
        pose_stds = np.array([self.odom_sigmas]).T
        pose_bounds = 5.0 * np.c_[-pose_stds, pose_stds]
        plan = self._replay_plan(pose_bounds)
        # the poses every session's shgo asks for first: the vertices of its sampling stage (+ with the replay the three
        # forward-difference points SLSQP adds per vertex)
        X0 = plan.points.reshape(-1, 3) if plan is not None else self._sobol_points(pose_bounds)
        src_pose, tgt_pose = pose.take(idx), prev.take(idx)

        def transforms(sel, X):
            """T6 [len(sel) x len(X) x 6] of target_pose.between(source_pose.compose(n2g(x))) (slam.py:548-550)"""
            return sample_transforms(self.ctx.lib, tgt_pose.take(sel), src_pose.take(sel), X)
        t_g = time.perf_counter()
        grids = mc._StoreGrids(self.store, tgt_h[idx], self.point_noise)
        self.init_stats["grids_s"] += time.perf_counter() - t_g
        try:
            t_a = time.perf_counter()
            d4 = np.stack([X0[:, 0], X0[:, 1], [math.cos(t) for t in X0[:, 2]], [math.sin(t) for t in X0[:, 2]]], axis=1)
            t4 = np.stack([tgt_pose.x, tgt_pose.y, tgt_pose.c, tgt_pose.s], axis=1)
            s4 = np.stack([src_pose.x, src_pose.y, src_pose.c, src_pose.s], axis=1)
            t_b = time.perf_counter()
            # [n x len(X0)]: one launch, the sample transforms target.between(source.compose(x)) computed on the device
            table = grids.cost_samples(src_h[idx], t4, s4, d4, f64_points=True)
            self.init_stats["transforms_s"] += t_b - t_a
            self.init_stats["table_s"] += time.perf_counter() - t_b
            self.init_stats["table_hits"] += n * len(X0)
            keys = [x.tobytes() for x in X0]
            ok, xs, fs = np.zeros(n, bool), np.zeros((n, 3)), np.zeros(n)
            t0 = time.perf_counter()
            todo = np.arange(n)
            if plan is not None:
                st, vx = plan.solve_many(self.ctx.lib, table.reshape(n, plan.V, 4))
                done = st != shgo_fast.FALLBACK
                good = st == shgo_fast.OK
                ok[good] = True
                xs[good] = plan.X[vx[good]]
                fs[good] = table.reshape(n, plan.V, 4)[np.nonzero(good)[0], vx[good], 0]
                todo = np.nonzero(~done)[0]
                self.init_stats["replayed"] += int(done.sum())
                self.init_stats["replay_fallbacks"] += len(todo)
            if self.shgo_workers > 1 and len(todo) > 1:
                # speculative runs on the host cores (shgo_pool.py), every assumed cost verified in one launch
                from . import shgo_pool
                if self._shgo_pool is None:
                    t_pool = time.perf_counter()
                    self._shgo_pool = shgo_pool.ShgoPool(self.shgo_workers)
                    t0 += time.perf_counter() - t_pool
                out = self._shgo_pool.map([(pose_bounds, self.initialization_params, X0, table[i].astype(np.int64)) for i in todo])
                who = np.concatenate([np.full(len(o[4]), i, np.int64) for i, o in zip(todo, out)])
                bad = np.zeros(n, bool)
                if len(who):
                    asked = np.concatenate([o[4] for o in out])
                    assumed = np.concatenate([o[5] for o in out])
                    T6 = np.zeros((len(who), 1, 6), np.float32)
                    d = Pose2Batch(asked[:, 0], asked[:, 1], asked[:, 2])
                    T6[:, 0] = tgt_pose.take(who).between(src_pose.take(who).compose(d)).T6()
                    true = grids.cost(src_h[idx[who]], T6, True, grid_index=who)[:, 0]
                    np.logical_or.at(bad, who, true != assumed)
                    self.init_stats["cost_calls"] += len(who)
                self.init_stats["speculated"] += len(todo)
                self.init_stats["speculation_failed"] += int(bad.sum())
                for i, o in zip(todo, out):
                    if not bad[i]:
                        ok[i] = o[0]
                        if o[0]:
                            xs[i], fs[i] = o[1], o[2]
                todo = np.nonzero(bad)[0]
            for i in todo:
                cache = dict(zip(keys, table[i]))

                def f(x, i=i, cache=cache):
                    x = np.asarray(x, np.float64)
                    v = cache.get(x.tobytes())
                    if v is None:
                        v = grids.cost(src_h[idx[i:i + 1]], transforms(np.array([i]), [x]), True, grid_index=[i])[0, 0]
                        cache[x.tobytes()] = v
                        self.init_stats["cost_calls"] += 1
                    else:
                        self.init_stats["table_hits"] += 1
                    return np.int64(v)
                res = shgo(func=f, bounds=pose_bounds, n=self.initialization_params[0], iters=self.initialization_params[1],
                           sampling_method="sobol", minimizer_kwargs={"options": {"ftol": self.initialization_params[2]}},
                           workers=lambda _fn, pts: [f(p) for p in pts])
                ok[i] = bool(res.success)
                if res.success:
                    xs[i], fs[i] = res.x, res.fun
            self.init_stats["shgo_s"] += time.perf_counter() - t0
        finally:
            grids.close()
        est = src_pose.compose(Pose2Batch(xs[:, 0], xs[:, 1], xs[:, 2]))
        return ok, est, xs, fs

    def _check_raw(self, k):
        """a ping with more detections than the batch's point capacity would be truncated silently"""
        raw = self.kb.d_cnt.download(np.int32, self.S)
        self.max_raw = max(self.max_raw, int(raw.max()))
        if int(raw.max()) > self.kb.cap:
            f = int(raw.argmax())
            raise _L.SonarFEError("step %d, session %d: %d points extracted, more than the batch capacity %d "
                                  "(max_points)" % (k, f, raw[f], self.kb.cap))

    def _check_counts(self, counts, k):
        bad = np.nonzero(counts < 0)[0]
        if len(bad):
            raise _L.SonarFEError("step %d, session %d: cloud not stored (count %d: -1 octree deeper than 24 levels, "
                                  "-3 store full)" % (k, int(bad[0]), int(counts[bad[0]])))

    def run(self):
        self.reset()
        for k in range(self.K):
            self.step(k)
        return self.records

    def free(self):
        if self._shgo_pool is not None:
            self._shgo_pool.close()
            self._shgo_pool = None
        self.kb.d_img = self._kb_img
        self.kb.free()
        self.d_frames.free()
        self.store.close()


PRIOR, SUCCESS, NOT_ENOUGH_POINTS, NOT_CONVERGED, LARGE_TRANSFORMATION, NOT_ENOUGH_OVERLAP, INITIALIZATION_FAILURE = range(7)
STATUS_NAMES = ("PRIOR", "SUCCESS", "NOT_ENOUGH_POINTS", "NOT_CONVERGED", "LARGE_TRANSFORMATION", "NOT_ENOUGH_OVERLAP",
                "INITIALIZATION_FAILURE")
