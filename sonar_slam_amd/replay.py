"""ROS-free, gtsam-free replay of the SLAM front end (BASELINE configs[2]: offline replay).

The reference replays a rosbag through ``FeatureExtraction.callback`` and ``SLAMNode.SLAM_callback``
(bruce_slam/src/bruce_slam/utils/io.py:14-33, slam_ros.py:157-213).  rosbag, rospy and gtsam are
not in this image, so this module restates the part of that flow that FEEDS the pose graph -- the
part this project accelerates -- with the same parameters (config/slam.yaml keys) and the same
order of operations, and leaves the graph optimiser pluggable:

    ping -> FeatureExtraction.callback           feature_extraction.py:196-252  (GPU)
         -> wire format (PointCloud2 xyz32)      feature_extraction.py:175-193, slam_ros.py:169-170
         -> keyframe test                        slam.py:1134-1161
         -> target cloud = get_points(last k)    slam.py:229-292, slam_objects.py:178-198  (GPU downsample)
         -> ICP(source, target, odometry guess)  slam.py:294-323 -> pcl.ICP.compute        (GPU)
         -> sanity checks, overlap               slam.py:786-811 -> pcl.match              (GPU)
         -> BetweenFactorPose2(target, source)   slam.py:813-832  -> ``backend.add_between``

With ``FrontEnd(store=CloudStore)`` the same flow runs on device-resident clouds: the feature
extractor leaves each cloud in the store (``FeatureExtraction.callback_store``), a keyframe holds a handle,
``get_points`` / ``compute_icp`` / ``get_overlap`` run on handles (sfe_cloud_store_get_points, sfe_icp_store_compute,
sfe_cloud_store_overlap); the host sees point COUNTS and results only, and the wire bytes are produced on request.
Results are identical to the host-mediated flow (tests/test_gpu_store.py).

The reference's DEFAULT flow runs the global initialisation in front of every scan match (slam.py:77,89:
``ssm_params.initialization = nssm_params.initialization = True``, and nothing in slam_ros.py or slam.yaml turns it off):
``scipy.optimize.shgo`` over ``get_matching_cost_subroutine1`` (slam.py:665-701), ICP then starts from the pose shgo
found, not from the odometry.  ``FrontEnd(ssm_initialization=True)`` -- the default here as well -- does exactly that, with
the cost function on the GPU (matching_cost.py; over store handles when the clouds are device-resident) and shgo's sampling
stage scored in one launch.  ``ssm_initialization=False`` is the path slam.py:665-666 takes when the flag is off.

Loop-closure search (NSSM, slam.py:839-1132): ``FrontEnd(nssm_enable=True)`` -- the default, as config/slam.yaml ships
``nssm/enable: True`` (slam_ros.py:69); sessions shorter than the exclusion zone of 8 keyframes never search --  runs
initialize_nonsequential_scan_matching and the many-guess ICP of compute_icp_with_cov after every keyframe -- aggregated source cloud, keyed global target cloud
(``get_points(..., return_keys=True)``, the descriptor overload of pcl.downsample), field-of-view gate, shgo, target-key
refinement by overlap, <= 30 ICPs on one pair, MinCovDet, the gates -- on host arrays or, with a store, on handles
(sfe_cloud_store_get_points_keys / fov_select / compact_selected / match_keys).  What follows it in the reference -- PCM
(slam.py:1089-1130, 1243-1331) and the ISAM2 update -- is the back end and stays out: an accepted loop closure
is handed to ``backend.add_loop`` and recorded.

``ChainBackend`` composes the accepted between-transforms, which is what ISAM2 returns for a
graph that only holds a prior and sequential between factors.  A gtsam-backed backend can be
dropped in where gtsam exists.
"""
import time as _time

import numpy as np

from . import _lib as _L
from . import icp_config, pcl, wire
from . import store as _store
from .pose2 import Pose2


class CloudRef(object):
    """A keyframe cloud that lives in a ``store.CloudStore``: handle + size (what ``len(points)`` tests read)."""

    __slots__ = ("handle", "n")

    def __init__(self, handle, n):
        self.handle, self.n = int(handle), int(n)

    def __len__(self):
        return self.n


class Keyframe(object):
    """slam_objects.py:120-176 (the fields the front end touches)."""

    def __init__(self, status, time, dr_pose, points=None):
        self.status = status
        self.time = time
        self.dr_pose = dr_pose
        self.pose = None
        self.cov = None
        self.points = points if points is not None else np.zeros((0, 2), np.float32)
        self.transf_points = None

    def update(self, new_pose):
        self.pose = new_pose
        if isinstance(self.points, np.ndarray):     # (a device-resident cloud is transformed where it is used)
            self.transf_points = Keyframe.transform_points(self.points, self.pose)

    update_pose = update

    @staticmethod
    def transform_points(points, pose):
        """slam_objects.py:178-198"""
        if len(points) == 0:
            return np.empty_like(points, np.float32)
        T = np.asarray(pose.matrix()).astype(np.float32)
        return points.dot(T[:2, :2].T) + T[:2, 2]


class ChainBackend(object):
    """prior + sequential between factors: the optimum is the chain of the between transforms"""

    def __init__(self):
        self.factors = []

    def add_prior(self, key, pose):
        self.factors.append(("prior", key, pose))

    def add_between(self, key_a, key_b, transform, kind):
        self.factors.append((kind, key_a, key_b, transform))

    def add_loop(self, key_a, key_b, transform, cov):
        """a loop closure that passed the front end's gates (slam.py:1066-1086); PCM and the graph update are the
        real back end's business"""
        self.factors.append(("loop", key_a, key_b, transform, cov))

    def marginal_covariance(self, key, prev_cov, kind):
        """Stand-in for isam.marginalCovariance(X(key)) (slam.py:1229-1230), which feeds the loop-closure search's
        bounds (slam.py:882-888, 929-932): first-order growth along the chain, the new factor's sigmas added to the
        previous keyframe's covariance.  An INPUT of the front end, not part of it."""
        sig = {"prior": (0.1, 0.1, 0.01), "icp": (0.1, 0.1, 0.01), "odometry": (0.2, 0.2, 0.02)}[kind]
        add = np.diag(np.square(sig))
        return add if prev_cov is None else prev_cov + add


class FrontEnd(object):
    """The sequential-scan-matching half of ``SLAM`` (slam.py), parameters from config/slam.yaml."""

    def __init__(self, ctx=None, icp_params=None, backend=None, keyframe_duration=1.0, keyframe_translation=3.0,
                 keyframe_rotation=np.deg2rad(30), point_resolution=0.5, point_noise=0.5, ssm_min_points=50,
                 ssm_max_translation=3.0, ssm_max_rotation=np.deg2rad(30), ssm_target_frames=3, store=None,
                 ssm_initialization=True, ssm_initialization_params=(50, 1, 0.01), odom_sigmas=(0.2, 0.2, 0.02),
                 nssm_enable=True, nssm_initialization=True, nssm_initialization_params=(100, 5, 0.01), nssm_min_st_sep=8,
                 nssm_min_points=50, nssm_max_translation=10.0, nssm_max_rotation=np.deg2rad(60), nssm_source_frames=5,
                 nssm_cov_samples=30, oculus_max_range=30.0, oculus_horizontal_aperture=np.radians(130.0),
                 mcd_random_state=None, shgo_replay=True):
        self.ctx = ctx
        self.shgo_replay = shgo_replay      # FrontEnd.shgo: replay of shgo's decisions from one table of costs (shgo_fast.py)
        self.store = store          # CloudStore: keyframe clouds stay on the device (feed_handle)
        self.icp = pcl.ICP(ctx)
        self.icp.setParams(icp_params if icp_params is not None else icp_config.shipped_params())
        self.backend = backend or ChainBackend()
        self.keyframe_duration = keyframe_duration
        self.keyframe_translation = keyframe_translation
        self.keyframe_rotation = keyframe_rotation
        self.point_resolution = point_resolution
        self.point_noise = point_noise
        self.ssm_min_points = ssm_min_points
        self.ssm_max_translation = ssm_max_translation
        self.ssm_max_rotation = ssm_max_rotation
        self.ssm_target_frames = ssm_target_frames
        self.icp_odom_sigmas = np.array([0.1, 0.1, 0.01])        # slam.yaml: icp_odom_sigmas
        # global initialisation (slam.py:77-78; slam.yaml has no key for it: the constructor's True stands)
        self.ssm_initialization = ssm_initialization
        self.ssm_initialization_params = ssm_initialization_params
        self.odom_sigmas = np.array(odom_sigmas, np.float64)       # slam.yaml: odom_sigmas
        # loop-closure search (slam.py:88-96 overridden by slam.yaml nssm/*: slam_ros.py:69-75)
        self.nssm_enable = nssm_enable
        self.nssm_initialization = nssm_initialization
        self.nssm_initialization_params = nssm_initialization_params
        self.nssm_min_st_sep, self.nssm_min_points = nssm_min_st_sep, nssm_min_points
        self.nssm_max_translation, self.nssm_max_rotation = nssm_max_translation, nssm_max_rotation
        self.nssm_source_frames, self.nssm_cov_samples = nssm_source_frames, nssm_cov_samples
        assert nssm_source_frames < nssm_min_st_sep                                           # slam.py:158
        assert nssm_cov_samples == 0 or nssm_cov_samples < nssm_initialization_params[0] * nssm_initialization_params[1]
        self.oculus_max_range, self.oculus_horizontal_aperture = oculus_max_range, oculus_horizontal_aperture  # sonar.py:151,158
        self.mcd_random_state = mcd_random_state    # None = the reference's MinCovDet (global numpy RNG)
        self.nssm_log = []
        self.keyframes = []
        self.current_frame = None
        self.log = []

    # -- slam.py:205-227 --
    @property
    def current_key(self):
        return len(self.keyframes)

    @property
    def current_keyframe(self):
        return self.keyframes[-1]

    def is_keyframe(self, frame):
        """slam.py:1134-1161"""
        if not self.keyframes:
            return True
        if frame.time - self.current_keyframe.time < self.keyframe_duration:
            return False
        dr_odom = self.keyframes[-1].dr_pose.between(frame.dr_pose)
        translation = np.linalg.norm(np.array([dr_odom.x(), dr_odom.y()]))       # np.linalg.norm(dr_odom.translation()), the reference's call
        rotation = abs(dr_odom.theta())
        return translation > self.keyframe_translation or rotation > self.keyframe_rotation

    def get_points(self, frames, ref_frame):
        """slam.py:229-292 (no keys): accumulate, move to the reference keyframe, downsample"""
        ref_pose = self.keyframes[ref_frame].pose
        if self.store is not None:
            frames = list(frames)
            T6 = [_store.pose_T6(ref_pose.between(self.keyframes[key].pose)) for key in frames]
            h = self.store.get_points([[self.keyframes[key].points.handle for key in frames]], [T6],
                                      self.point_resolution)[0]
            return CloudRef(h, self.store.counts([h])[0])
        all_points = [np.zeros((0, 2), np.float32)]
        for key in frames:
            transf = ref_pose.between(self.keyframes[key].pose)
            all_points.append(Keyframe.transform_points(self.keyframes[key].points, transf))
        return pcl.downsample(np.concatenate(all_points), self.point_resolution)

    def compute_icp(self, source_points, target_points, guess):
        """slam.py:294-323"""
        if self.store is not None:
            T, st, _ = self.store.icp(self.icp._chain(), [(source_points.handle, target_points.handle)],
                                      [pcl.ICP._guess(guess.matrix())])
            message, T = _L.ICP_STATUS_MESSAGES.get(int(st[0]), "ICP failure %d" % st[0]), T[0]
            x, y = T[:2, 2]
            return message, Pose2(x, y, np.arctan2(T[1, 0], T[0, 0]))
        source_points = np.array(source_points, np.float32)
        target_points = np.array(target_points, np.float32)
        message, T = self.icp.compute(source_points, target_points, guess.matrix())
        x, y = T[:2, 2]
        theta = np.arctan2(T[1, 0], T[0, 0])
        return message, Pose2(x, y, theta)

    def compute_icp_with_cov(self, source_points, target_points, guesses):
        """What slam.py:325-387 computes -- the scatter of the transforms ICP converges to from several initial
        guesses on ONE cloud pair, as a robust covariance in the frame of their centre -- with the one change
        INTEGRATION.md section 3 describes: the guesses go through ONE launch (``pcl.ICP.compute_batch`` /
        ``sfe_icp_store_compute`` over handles: the target is prepared once) instead of a timed Python loop of
        ``compute`` calls (the 2 s budget of :346-358 is never in reach).
        -> (message, centre Pose2, cov 3 x 3, converged transforms [n x 3]); the messages are the reference's."""
        from sklearn.covariance import MinCovDet
        if not len(guesses):
            return "Too few samples for covariance computation", None, None, None
        if self.store is not None:
            Ts, st, _ = self.store.icp(self.icp._chain(), [(source_points.handle, target_points.handle)] * len(guesses),
                                       [pcl.ICP._guess(g.matrix()) for g in guesses])
            ok = np.asarray(st) == 0
        else:
            msgs, Ts, _ = self.icp.compute_batch(np.asarray(source_points, np.float32), np.asarray(target_points, np.float32),
                                                 [g.matrix() for g in guesses])
            ok = np.array([m == "success" for m in msgs], bool)
        Ts = np.asarray(Ts, np.float32)[ok]
        # x, y = T[:2, 2]; theta = np.arctan2(T[1, 0], T[0, 0]) on the float32 matrix; np.array of tuples of np.float32 scalars
        # STAYS float32 (slam.py:352-361), and MinCovDet then takes its location as a float32 mean (its covariance in double):
        # pinned by tests/golden/nssm_pieces.npz -- rounds 1-5 cast to float64 here and were 6e-8 off the reference's centre
        xyt = np.c_[Ts[:, 0, 2], Ts[:, 1, 2], np.arctan2(Ts[:, 1, 0], Ts[:, 0, 0])].astype(np.float32) if len(Ts) else np.zeros((0, 3), np.float32)
        if len(xyt) < 5:
            return "Too few samples for covariance computation", None, None, None
        try:
            est = MinCovDet(store_precision=False, support_fraction=0.8, random_state=self.mcd_random_state).fit(xyt)
        except ValueError:
            return "Failed to calculate covariance", None, None, None
        centre = self._as_pose(est.location_)
        cov = est.covariance_
        R = np.asarray(centre.matrix())[:2, :2]                     # m.rotation().matrix()
        cov[:2, :] = R.T.dot(cov[:2, :])                            # slam.py:378-380
        cov[:, :2] = cov[:, :2].dot(R)
        floor = np.diag(self.icp_odom_sigmas) ** 2                  # never more confident than the configured sigmas
        if np.linalg.det(cov) < np.linalg.det(floor):
            cov = floor
        return "success", centre, cov, xyt

    @staticmethod
    def _as_pose(xytheta):
        return Pose2(float(xytheta[0]), float(xytheta[1]), float(xytheta[2]))

    def get_overlap(self, source_points, target_points, source_pose, f32_source=False):
        """slam.py:389-424 (``f32_source``: the source handle stands for a float32 cloud -- what get_points returned, the
        loop-closure source -- not for a float64 keyframe cloud; host arrays carry their dtype themselves)"""
        if self.store is not None:
            return int(self.store.overlap([(source_points.handle, target_points.handle)], [_store.pose_T6(source_pose)],
                                          self.point_noise, flags=_store.F32_POINTS if f32_source else 0)[0])
        source_points = Keyframe.transform_points(source_points, source_pose)
        indices, _ = pcl.match(target_points, source_points, 1, self.point_noise)
        return int(np.sum(indices != -1))

    # -- global initialisation: slam.py:461-570 (cost), :665-716 / :922-973 (shgo) --
    def matching_cost_subroutine(self, source_points, source_pose, target_points, target_pose, cov, f64_source):
        """get_matching_cost_subroutine1 on whatever the clouds are (host arrays or store handles)"""
        from . import matching_cost as mc
        if self.store is not None:
            return mc.get_matching_cost_subroutine1_store(self.store, source_points.handle, source_pose, target_points.handle,
                                                          target_pose, cov, point_noise=self.point_noise, f64_points=f64_source)
        return mc.get_matching_cost_subroutine1(source_points, source_pose, target_points, target_pose, cov,
                                                point_noise=self.point_noise, ctx=self.ctx)

    @staticmethod
    def shgo(subroutine, pose_bounds, params, replay=True):
        """slam.py:692-701 / :952-961.  ``replay`` (one-iteration calls only, i.e. the sequential scan match's): the cost at shgo's
        sampling points and at the finite-difference points of its local minimiser is taken in ONE launch and what shgo decides
        from there is replayed (shgo_fast.py; checked against the installed scipy once per process).  Otherwise, and whenever the
        replay reports a problem as undecidable, scipy.optimize.shgo itself: the reference's call verbatim, the points of a
        sampling stage scored in one launch (``workers`` is shgo's own hook for evaluating the pool of new vertices: the
        results and their order are those of the one-by-one loop, tests/test_global_init.py)"""
        from scipy.optimize import OptimizeResult, shgo
        from . import shgo_fast
        if replay and params[1] == 1:
            plan = shgo_fast.plan_for(pose_bounds, params[0], params[2])
            if plan.checked is None:
                plan.self_check()
            if plan.checked:
                # (the costs are taken without entering them into the subroutine's pose samples; what the reference's own run
                #  evaluates -- every vertex, then four points per local minimisation -- is entered below: ADVICE r5.  When the
                #  problem goes to scipy instead, scipy's own calls enter theirs.)
                can_record = hasattr(subroutine, "record")
                pts = list(plan.points.reshape(-1, 3))
                table = np.array(subroutine.batch(pts, record=False) if can_record else subroutine.batch(pts), np.int64).reshape(plan.V, 4)
                st, x, fun, order = plan.solve_order(table)
                n_local = len(order)
                if st != shgo_fast.FALLBACK and can_record:
                    subroutine.record(np.concatenate([plan.X, plan.points[order].reshape(-1, 3)]),
                                      np.concatenate([table[:, 0], table[order].reshape(-1)]))
                if st == shgo_fast.OK:
                    return OptimizeResult(x=x, fun=np.int64(fun), success=True, message="Optimization terminated successfully.",
                                          nfev=plan.V + 4 * n_local, nlfev=4 * n_local, replayed=True)
                if st == shgo_fast.FAILED:
                    return OptimizeResult(x=x, fun=np.int64(fun), success=False, nfev=plan.V, replayed=True,
                                          message="Failed to find a feasible minimizer point. Lowest sampling point = %s" % fun)
        elif replay and hasattr(subroutine, "record") and shgo_fast.multi_checked(params[0], params[1], params[2]):
            # several iterations (the loop-closure search): every point that can become a vertex + its finite-difference points in
            # ONE launch, the iterations replayed on the host (shgo_fast.replay_multi); the evaluations the reference's run makes --
            # the vertices shgo creates, four per local minimisation -- are the ones entered into the subroutine's pose_samples
            draws, cand, fd = shgo_fast.multi_candidates(pose_bounds, params[0], params[1])
            M = len(cand)
            costs = np.asarray(subroutine.batch(np.concatenate([cand, fd.reshape(-1, 3)]), record=False), np.int64)
            cost, fd_cost = costs[:M], costs[M:].reshape(M, 3)
            st, x, fun, vertices, minimised = shgo_fast.replay_multi(pose_bounds, params[0], params[1], draws, cand, cost, fd_cost)
            if st != shgo_fast.FALLBACK:
                X = np.concatenate([cand[vertices], cand[minimised], fd[minimised].reshape(-1, 3)])
                subroutine.record(X, np.concatenate([cost[vertices], cost[minimised], fd_cost[minimised].reshape(-1)]))
                if st == shgo_fast.OK:
                    return OptimizeResult(x=x, fun=np.int64(fun), success=True, message="Optimization terminated successfully.",
                                          nfev=len(X), replayed=True)
                return OptimizeResult(x=x, fun=np.int64(fun), success=False, nfev=len(vertices), replayed=True,
                                      message="Failed to find a feasible minimizer point. Lowest sampling point = %s" % fun)

        def pool(_fn, xs):
            xs = [np.asarray(x, np.float64) for x in xs]
            return [np.int64(c) for c in subroutine.batch(xs)] if xs else []
        return shgo(func=subroutine, bounds=pose_bounds, n=params[0], iters=params[1], sampling_method="sobol",
                    minimizer_kwargs={"options": {"ftol": params[2]}}, workers=pool)

    def warm_up(self):
        """Build and self-check the shgo replays this front end will use NOW (tens of scipy.optimize.shgo runs on random step
        functions, a few seconds) instead of inside the first keyframe / loop-closure callback of a live node, and
        say in the log whether they are active for the installed scipy (shgo_fast.status()).  -> shgo_fast.status()"""
        from . import shgo_fast
        if self.shgo_replay:
            if self.ssm_initialization and self.ssm_initialization_params[1] == 1:
                pose_stds = np.array([self.odom_sigmas]).T
                plan = shgo_fast.plan_for(5.0 * np.c_[-pose_stds, pose_stds], self.ssm_initialization_params[0],
                                          self.ssm_initialization_params[2])
                if plan.checked is None:
                    plan.self_check()
            if self.nssm_enable and self.nssm_initialization and self.nssm_initialization_params[1] > 1:
                shgo_fast.multi_checked(*self.nssm_initialization_params)
        return shgo_fast.status()

    def add_sequential_scan_matching(self, keyframe):
        """slam.py:607-832: initialize_sequential_scan_matching (target cloud, the point-count tests, the global
        initialisation when ``ssm_initialization``) + add_sequential_scan_matching"""
        source_key, target_key = self.current_key, self.current_key - 1
        target_pose = self.current_keyframe.pose
        source_points = keyframe.points
        target_frames = range(self.current_key)[-self.ssm_target_frames:]
        target_points = self.get_points(target_frames, target_key)
        rec = {"source_key": source_key, "target_key": target_key, "n_source": len(source_points),
               "n_target": len(target_points)}
        dr_between = self.current_keyframe.pose.between(keyframe.pose)
        if len(source_points) < self.ssm_min_points or len(target_points) < self.ssm_min_points:
            rec["status"] = "NOT_ENOUGH_POINTS"
            self.backend.add_between(target_key, source_key, dr_between, "odometry")
            self._release(target_points)
            return rec
        initial_transform = target_pose.between(keyframe.pose)
        try:
            if self.ssm_initialization:                                         # slam.py:665-716
                pose_stds = np.array([self.odom_sigmas]).T
                pose_bounds = 5.0 * np.c_[-pose_stds, pose_stds]
                subroutine, _ = self.matching_cost_subroutine(source_points, keyframe.pose, target_points, target_pose,
                                                              np.diag(self.odom_sigmas), f64_source=True)
                try:
                    result = self.shgo(subroutine, pose_bounds, self.ssm_initialization_params, replay=self.shgo_replay)
                finally:
                    subroutine.grid.close()
                rec["init_success"] = bool(result.success)
                if not result.success:
                    rec["status"] = "INITIALIZATION_FAILURE"
                    rec["init_message"] = str(result.message)
                    self.backend.add_between(target_key, source_key, dr_between, "odometry")    # slam.py:736-738
                    return rec
                rec["init_x"] = tuple(float(v) for v in result.x)
                rec["init_cost"] = float(result.fun)
                estimated_source_pose = keyframe.pose.compose(Pose2(*result.x))
                initial_transform = target_pose.between(estimated_source_pose)  # slam_objects.py:282-285
            return self._scan_match(keyframe, rec, source_key, target_key, target_pose, source_points, target_points,
                                    initial_transform, dr_between)
        finally:
            self._release(target_points)    # the target cloud get_points built is dropped (it was the newest slot)

    def _scan_match(self, keyframe, rec, source_key, target_key, target_pose, source_points, target_points,
                    initial_transform, dr_between):
        message, estimated = self.compute_icp(source_points, target_points, initial_transform)
        rec["icp"] = message
        status = "SUCCESS"
        if message != "success":
            status = "NOT_CONVERGED"
        if status == "SUCCESS":
            delta = initial_transform.between(estimated)
            if (np.linalg.norm(np.array([delta.x(), delta.y()])) > self.ssm_max_translation or     # (np.linalg.norm(delta.translation()))
                    abs(delta.theta()) > self.ssm_max_rotation):
                status = "LARGE_TRANSFORMATION"
        if status == "SUCCESS":
            overlap = self.get_overlap(source_points, target_points, estimated)
            rec["overlap"] = overlap
            if overlap < self.ssm_min_points:
                status = "NOT_ENOUGH_OVERLAP"
        rec["status"] = status
        if status == "SUCCESS":
            self.backend.add_between(target_key, source_key, estimated, "icp")
            keyframe.update(target_pose.compose(estimated))      # values.insert(X(source), ...) slam.py:825-827
        else:
            self.backend.add_between(target_key, source_key, dr_between, "odometry")   # add_odometry slam.py:444-459
        rec["transform"] = (estimated.x(), estimated.y(), estimated.theta())
        return rec


    # -- loop-closure search: slam.py:839-1132 up to (not including) PCM --
    def get_points_keys(self, frames):
        """SLAM.get_points(frames, None, return_keys=True) (slam.py:229-292): every cloud under its own pose, with its
        key; the descriptor overload of pcl.downsample -> (points, keys) or a keyed CloudRef"""
        frames = list(frames)
        if self.store is not None:
            h = self.store.get_points_keys([self.keyframes[k].points.handle for k in frames],
                                           [_store.pose_T6(self.keyframes[k].pose) for k in frames], frames, self.point_resolution)
            return CloudRef(h, self.store.counts([h])[0]), None
        all_points = [np.zeros((0, 3), np.float32)]
        for key in frames:
            tp = Keyframe.transform_points(self.keyframes[key].points, self.keyframes[key].pose)    # = keyframe.transf_points
            all_points.append(np.c_[tp, key * np.ones((len(tp), 1))])
        all_points = np.concatenate(all_points)
        return pcl.downsample(all_points[:, :2], all_points[:, (2,)], self.point_resolution)

    def _fov_bounds(self, source_frames):
        Tinv, rb, bb = [], [], []
        for f in source_frames:
            pose, cov = self.keyframes[f].pose, self.keyframes[f].cov
            translation_std = np.sqrt(np.max(np.linalg.eigvals(cov[:2, :2])))
            rotation_std = np.sqrt(cov[2, 2])
            rb.append(translation_std * 5.0 + self.oculus_max_range)                    # slam.py:885-888
            bb.append(rotation_std * 5.0 + self.oculus_horizontal_aperture * 0.5)
            Tinv.append(pose.inverse())
        return Tinv, rb, bb

    @staticmethod
    def _fov_numpy(target_points, Tinv, rb, bb):
        """slam.py:877-895 verbatim (np.bool is gone from numpy: bool)"""
        sel = np.zeros(len(target_points), bool)
        for pinv, range_bound, bearing_bound in zip(Tinv, rb, bb):
            local_points = Keyframe.transform_points(target_points, pinv)
            ranges = np.linalg.norm(local_points, axis=1)
            bearings = np.arctan2(local_points[:, 1], local_points[:, 0])
            sel |= (ranges < range_bound) & (abs(bearings) < bearing_bound)
        return sel

    def add_nonsequential_scan_matching(self):
        """slam.py:1003-1087 (+ :839-1001): -> the record of this keyframe's loop-closure search, or None when the
        session is shorter than the exclusion zone"""
        if self.current_key < self.nssm_min_st_sep:
            return None
        n_keep = len(self.store) if self.store is not None else 0
        rec = {"source_key": self.current_key - 1}
        try:
            self._nssm(rec)
        finally:
            if self.store is not None:
                self.store.truncate(n_keep)         # every cloud the search built is dropped again
        self.nssm_log.append(rec)
        return rec

    def _nssm(self, rec):
        K = self.current_key
        source_key = K - 1
        source_pose = self.current_frame.pose               # slam.py:854 (the frame of the PREVIOUS callback: slam_ros.py:211)
        source_frames = list(range(source_key, source_key - self.nssm_source_frames, -1))
        source_points = self.get_points(source_frames, source_key)
        rec["n_source"] = len(source_points)
        if len(source_points) < self.nssm_min_points:
            rec["status"] = "NOT_ENOUGH_POINTS"
            return
        target_frames = list(range(K - self.nssm_min_st_sep))
        Tinv, rb, bb = self._fov_bounds(source_frames)
        if self.store is not None:
            G, _ = self.get_points_keys(target_frames)
            hist, n_sel, n_amb = self.store.fov_select(G.handle, [_store.pose_T6(p) for p in Tinv], rb, bb, K)
            if n_amb:       # a bearing within float32 rounding of its bound: numpy decides (see sfe_cloud_store_fov_select)
                sel = self._fov_numpy(self.store.read(G.handle), Tinv, rb, bb)
                self.store.set_selection(G.handle, sel)
                hist = np.bincount(self.store.read_keys(G.handle)[sel], minlength=K).astype(np.int32)
                n_sel = int(sel.sum())
            rec["fov_ambiguous"] = int(n_amb)
            frames1 = np.nonzero(hist)[0].astype(np.int32)
            counts = hist[frames1]
        else:
            target_points, target_keys = self.get_points_keys(target_frames)
            sel = self._fov_numpy(target_points, Tinv, rb, bb)
            target_points, target_keys = target_points[sel], target_keys[sel]
            n_sel = len(target_points)
            frames1, counts = np.unique(np.int32(target_keys), return_counts=True)
        rec["n_target_global"] = int(n_sel)
        frames1, counts = frames1[counts > 10], counts[counts > 10]
        if len(frames1) == 0 or n_sel < self.nssm_min_points:
            rec["status"] = "NOT_ENOUGH_POINTS"
            return
        target_key = int(frames1[np.argmax(counts)])
        target_pose = self.keyframes[target_key].pose
        rec["target_key_fov"] = target_key
        if self.store is not None:
            S = CloudRef(self.store.compact_selected(G.handle), n_sel)
            tl = self.store.get_points([[S.handle]], [[_store.pose_T6(target_pose.inverse())]], 0.0, flags=_store.F32_POINTS)[0]
            target_local = CloudRef(tl, n_sel)
        else:
            target_local = Keyframe.transform_points(target_points, target_pose.inverse())
        cov = self.keyframes[source_key].cov
        estimated_source_pose, pose_samples = source_pose, None
        if self.nssm_initialization:
            c = self.keyframes[source_frames[-1]].cov       # slam.py:929: `cov` is what the gate's loop left behind
            translation_std = np.sqrt(np.max(np.linalg.eigvals(c[:2, :2])))
            rotation_std = np.sqrt(c[2, 2])
            pose_stds = np.array([[translation_std, translation_std, rotation_std]]).T
            pose_bounds = 5.0 * np.c_[-pose_stds, pose_stds]
            subroutine, pose_samples = self.matching_cost_subroutine(source_points, source_pose, target_local, target_pose, cov,
                                                                     f64_source=False)
            try:
                result = self.shgo(subroutine, pose_bounds, self.nssm_initialization_params, replay=self.shgo_replay)
                rec["init_replayed"] = bool(result.get("replayed", False))
            finally:
                subroutine.grid.close()
            if not result.success:
                rec["status"], rec["init_message"] = "INITIALIZATION_FAILURE", str(result.message)
                return
            rec["init_x"], rec["init_cost"] = tuple(float(v) for v in result.x), float(result.fun)
            estimated_source_pose = source_pose.compose(Pose2(*result.x))
            pose_samples = np.array(pose_samples)
            # refine the target key: the keyframe most of the matched target points came from (slam.py:975-999)
            if self.store is not None:
                hist1, overlap = self.store.match_keys(source_points.handle, _store.pose_T6(estimated_source_pose), S.handle,
                                                       self.point_noise, K, flags=_store.F32_POINTS)
            else:
                moved = Keyframe.transform_points(source_points, estimated_source_pose)
                indices, _ = pcl.match(target_points, moved, 1, self.point_noise)
                hist1 = np.bincount(np.int32(target_keys[indices[indices != -1]]).reshape(-1), minlength=K)
                overlap = int(np.sum(indices != -1))
            rec["overlap_global"] = int(overlap)
            if overlap == 0:
                rec["status"] = "NOT_ENOUGH_OVERLAP"
                return
            target_key = int(np.argmax(hist1))
            target_pose = self.keyframes[target_key].pose
            target_local = self.get_points(target_frames, target_key)
        rec["target_key"], rec["n_target"] = target_key, len(target_local)
        # ICPResult (slam_objects.py:247-300)
        initial_transform = target_pose.between(estimated_source_pose)
        use_samples = self.nssm_cov_samples > 0 and pose_samples is not None
        if self.nssm_initialization and self.nssm_cov_samples > 0:
            guesses = self.initial_transforms(pose_samples, target_pose, limit=self.nssm_cov_samples) if use_samples else []
            rec["n_guesses"] = len(guesses)
            message, odom, cov_icp, samples = self.compute_icp_with_cov(source_points, target_local, guesses)
            rec["icp"] = message
            if message != "success":
                rec["status"] = "NOT_CONVERGED"
                return
            rec["n_converged"] = len(samples)
            rec["sample_transforms"] = samples
            rec["cov"] = cov_icp
        else:
            message, odom = self.compute_icp(source_points, target_local, initial_transform)
            cov_icp = None
            rec["icp"] = message
            if message != "success":
                rec["status"] = "NOT_CONVERGED"
                return
        rec["transform"] = (odom.x(), odom.y(), odom.theta())
        delta = initial_transform.between(odom)                                 # slam.py:1066-1077
        if (np.linalg.norm(np.array([delta.x(), delta.y()])) > self.nssm_max_translation or abs(delta.theta()) > self.nssm_max_rotation):
            rec["status"] = "LARGE_TRANSFORMATION"
            return
        overlap = self.get_overlap(source_points, target_local, odom, f32_source=True)
        rec["overlap"] = overlap
        if overlap < self.nssm_min_points:
            rec["status"] = "NOT_ENOUGH_OVERLAP"
            return
        rec["status"] = "SUCCESS"
        self.backend.add_loop(target_key, source_key, odom, cov_icp)           # -> PCM + ISAM2 (slam.py:1089-1130): back end

    @staticmethod
    def initial_transforms(pose_samples, target_pose, sample_eps=0.01, limit=None):
        """ICPResult.__init__ (slam_objects.py:287-300): the sampled source poses by ascending cost, as transforms from
        the target, near-duplicates dropped.  The reference sorts with ``np.argsort(cost)`` -- an unstable sort of integer
        costs full of ties, over samples whose order is the iteration order of a Python set inside shgo -- so its order
        among equal costs changes from run to run; here ties go by (x, y, theta): one of the orders the reference can
        produce, the same on every run.  ``limit``: stop after that many kept transforms (the caller uses the first
        cov_samples only, slam.py:346: the filter looks at the last kept one alone, so the head of the list is the same)."""
        ps = np.asarray(pose_samples, np.float64)
        idx = np.lexsort((ps[:, 2], ps[:, 1], ps[:, 0], ps[:, 3]))
        filtered = []
        for g in ps[idx, :3]:
            b = target_pose.between(Pose2(*g))
            if filtered:
                d = filtered[-1].between(b)
                if np.linalg.norm([d.x(), d.y(), d.theta()]) < sample_eps:
                    continue
            filtered.append(b)
            if limit is not None and len(filtered) >= limit:
                break
        return filtered

    def _release(self, ref):
        """a cloud that nothing refers to any more and that is the newest slot of the store (stack order)"""
        if self.store is not None and isinstance(ref, CloudRef) and ref.handle >= 0:
            self.store.truncate(ref.handle)

    def feed(self, cloud_bytes, time, dr_pose):
        """``SLAMNode.SLAM_callback`` (slam_ros.py:157-213): one feature message + its odometry."""
        points = wire.unpack_features(cloud_bytes)
        if self.store is not None:      # a message that did arrive over the wire: one upload, then as below
            skipped = wire.is_skipped(points)
            ref = CloudRef(-1, 0) if skipped else CloudRef(self.store.put(points, int(time)), len(points))
            return self._feed(ref, skipped, time, dr_pose)
        return self._feed(points, wire.is_skipped(points), time, dr_pose)

    def feed_handle(self, handle, n_points, time, dr_pose):
        """The same for a cloud ``FeatureExtraction.callback_store`` left in the store (handle < 0: a skipped
        frame).  A frame that does not become a keyframe gives its slot back (it is the newest one)."""
        return self._feed(CloudRef(handle, n_points), handle < 0, time, dr_pose)

    def _feed(self, points, skipped, time, dr_pose):
        frame = Keyframe(False, time, dr_pose)
        if skipped:
            frame.status = False
        else:
            frame.status = self.is_keyframe(frame)
        if self.keyframes:
            dr_odom = self.current_keyframe.dr_pose.between(frame.dr_pose)
            frame.update_pose(self.current_keyframe.pose.compose(dr_odom))
        else:
            frame.update_pose(dr_pose)
        rec = None
        if frame.status:
            frame.points = points                       # slam_ros.py:190 (float64 array of float32 values, or a handle)
            frame.update(frame.pose)
            if not self.keyframes:
                self.backend.add_prior(0, frame.pose)
                rec = {"source_key": 0, "status": "PRIOR", "n_source": len(points)}
            else:
                rec = self.add_sequential_scan_matching(frame)
            # update_factor_graph(frame): append, latest marginal covariance (slam.py:1210-1230; stand-in, see ChainBackend)
            kind = "prior" if not self.keyframes else ("icp" if rec["status"] == "SUCCESS" else "odometry")
            if hasattr(self.backend, "marginal_covariance"):
                frame.cov = self.backend.marginal_covariance(self.current_key, self.keyframes[-1].cov if self.keyframes else None, kind)
            self.keyframes.append(frame)
            rec["pose"] = (frame.pose.x(), frame.pose.y(), frame.pose.theta())
            rec["time"] = time
            self.log.append(rec)
            if self.nssm_enable and self.current_frame is not None:     # slam_ros.py:207
                rec["nssm"] = self.add_nonsequential_scan_matching()
        else:
            self._release(points)
        self.current_frame = frame
        return rec


def replay(pings, stamps, dr_poses, feature_extraction, front_end):
    """Offline loop of utils/io.py:14-33: every ping through the feature extractor, its cloud through
    the wire format into the front end.  -> (per-keyframe records, seconds spent in extraction,
    seconds spent in the SLAM front end)."""
    t_fe = t_slam = 0.0
    for ping, stamp, dr in zip(pings, stamps, dr_poses):
        t0 = _time.perf_counter()
        if front_end.store is not None:
            # same process, same device: the cloud stays in the store; nothing is packed unless somebody subscribes
            h, n, _ = feature_extraction.callback_store(ping, front_end.store, stamp=int(stamp))
            t1 = _time.perf_counter()
            front_end.feed_handle(h, n, stamp, Pose2(*dr))
            t2 = _time.perf_counter()
            t_fe += t1 - t0
            t_slam += t2 - t1
            continue
        pts = feature_extraction.callback(ping)
        data = wire.pack_features(pts)
        t1 = _time.perf_counter()
        front_end.feed(data, stamp, Pose2(*dr))
        t2 = _time.perf_counter()
        t_fe += t1 - t0
        t_slam += t2 - t1
    return front_end.log, t_fe, t_slam
