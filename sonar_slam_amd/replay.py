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

``ChainBackend`` composes the accepted between-transforms, which is what ISAM2 returns for a
graph that only holds a prior and sequential between factors.  A gtsam-backed backend can be
dropped in where gtsam exists.  Loop closures (NSSM + PCM) need the real
optimiser and are outside this harness; their many-guess ICP batch is ``pcl.ICP.compute_batch``.
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


class FrontEnd(object):
    """The sequential-scan-matching half of ``SLAM`` (slam.py), parameters from config/slam.yaml."""

    def __init__(self, ctx=None, icp_params=None, backend=None, keyframe_duration=1.0, keyframe_translation=3.0,
                 keyframe_rotation=np.deg2rad(30), point_resolution=0.5, point_noise=0.5, ssm_min_points=50,
                 ssm_max_translation=3.0, ssm_max_rotation=np.deg2rad(30), ssm_target_frames=3, store=None):
        self.ctx = ctx
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
        translation = float(np.hypot(dr_odom.x(), dr_odom.y()))
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
        INTEGRATION.md section 3 describes: the guesses go through ``pcl.ICP.compute_batch`` in a single launch
        instead of a timed Python loop of ``compute`` calls (the 2 s budget of :346-358 is never in reach).
        -> (message, centre Pose2, cov 3 x 3, converged transforms [n x 3]); the messages are the reference's."""
        from sklearn.covariance import MinCovDet
        msgs, Ts, _ = self.icp.compute_batch(np.asarray(source_points, np.float32), np.asarray(target_points, np.float32),
                                             [g.matrix() for g in guesses])
        ok = np.array([m == "success" for m in msgs], bool)
        Ts = np.asarray(Ts, np.float64)[ok]
        xyt = np.c_[Ts[:, 0, 2], Ts[:, 1, 2], np.arctan2(Ts[:, 1, 0], Ts[:, 0, 0])] if len(Ts) else np.zeros((0, 3))
        if len(xyt) < 5:
            return "Too few samples for covariance computation", None, None, None
        try:
            est = MinCovDet(store_precision=False, support_fraction=0.8).fit(xyt)
        except ValueError:
            return "Failed to calculate covariance", None, None, None
        centre = self._as_pose(est.location_)
        # translation block expressed in the centre pose's own axes: cov_local = B^T cov B, B = blockdiag(R, 1)
        B = np.eye(3)
        B[:2, :2] = [[np.cos(centre.theta()), -np.sin(centre.theta())], [np.sin(centre.theta()), np.cos(centre.theta())]]
        cov = B.T @ est.covariance_ @ B
        floor = np.diag(np.square(self.icp_odom_sigmas))            # never more confident than the configured sigmas
        return "success", centre, (cov if np.linalg.det(cov) >= np.linalg.det(floor) else floor), xyt

    @staticmethod
    def _as_pose(xytheta):
        return Pose2(float(xytheta[0]), float(xytheta[1]), float(xytheta[2]))

    def get_overlap(self, source_points, target_points, source_pose):
        """slam.py:389-424"""
        if self.store is not None:
            return int(self.store.overlap([(source_points.handle, target_points.handle)], [_store.pose_T6(source_pose)],
                                          self.point_noise)[0])
        source_points = Keyframe.transform_points(source_points, source_pose)
        indices, _ = pcl.match(target_points, source_points, 1, self.point_noise)
        return int(np.sum(indices != -1))

    def add_sequential_scan_matching(self, keyframe):
        """slam.py:716-832 without the shgo initialisation (ssm.initialization off, the default path)"""
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
            if (float(np.hypot(delta.x(), delta.y())) > self.ssm_max_translation or
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
            self.keyframes.append(frame)
            rec["pose"] = (frame.pose.x(), frame.pose.y(), frame.pose.theta())
            rec["time"] = time
            self.log.append(rec)
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
