#!/usr/bin/env python
"""Generate the golden fixtures in this directory FROM THE REFERENCE's own Python code.

Run in the build container (needs /root/reference; the GPU box does not have it):

    python tests/golden/make_golden.py

The reference modules cannot be imported as a package here (rospy, cv2, gtsam and the compiled
``cfar`` module are absent), so the two pieces of pure-Python reference logic on the hot path
are executed straight from the reference sources:

  * ``bruce_slam/src/bruce_slam/CFAR.py`` is exec'd with its three package-relative imports
    removed and a stub ``cfar`` module -> the threshold factors tau of CFAR.__init__
    (CFAR.py:17-52,71-121) for several (Ntc, Ngc, Pfa, rank) -> ``cfar_tau.json``
  * ``FeatureExtraction.generate_map_xy`` (feature_extraction.py:134-173) is cut out of the
    class by AST and run on synthetic pings -> ``maps_small.npz`` (full float32 maps of a small
    geometry) and ``maps_digest.json`` (sha256 + samples of the 512x1024 and 1024x2048 ones)

  * ``SLAM.get_matching_cost_subroutine1`` (slam.py:461-570) and ``Keyframe.transform_points``
    (slam_objects.py:178-198) are cut out by AST and run on a synthetic scan pair and candidate
    poses, with stand-ins for the two things they call that are not in this image: ``cv2``
    (getStructuringElement / dilate from the oracle's restatement: those two stay unpinned) and
    ``gtsam.Pose2`` (the 20-line ``Pose2`` below: gtsam's published Rot2 / Pose2 algebra, independent of the
    product's pose2.py; only its matrix reaches the reference functions) -> ``matching_cost.npz`` (target cells, grid shape,
    costs per pose, float32 transformed points of one pose).  Everything numpy does in there
    (bounds, np.arange lengths, rounding, clipping, BLAS dot of transform_points, inside test, sum)
    is the reference's own code on this image's numpy.

  * (round 5) two pieces of the loop-closure search: the field-of-view gate and the per-keyframe counts of
    ``initialize_nonsequential_scan_matching`` (slam.py:875-904: a block inside a long method, cut out by its first and last
    source lines and exec'd on prepared locals), ``ICPResult.__init__`` (slam_objects.py:247-300: which pose samples become
    ICP guesses, in which order), ``get_points(frames, None, return_keys=True)`` (slam.py:229-292: the keyed global target cloud),
    ``get_overlap`` (slam.py:389-424) and ``compute_icp_with_cov`` (slam.py:325-387) with the oracle standing in for pcl.downsample /
    pcl.match / pcl.ICP.compute and sklearn's own MinCovDet -> ``nssm_pieces.npz``.  `python tests/golden/make_golden.py nssm` writes that file alone.

Nothing of the reference is copied into the repository: only the numbers it produces.
"""
import ast
import hashlib
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference/bruce_slam/src/bruce_slam"
HERE = os.path.dirname(os.path.abspath(__file__))
ICP_YAML = "/root/reference/bruce_slam/config/icp.yaml"
sys.path.insert(0, HERE)
import thirdparty  # noqa: E402  (cv2 / bruce_slam.pcl REAL where this machine has them, the oracle's restatement where not;
#                                 every file written below records which stand-ins were used: `stand_ins`)


class Pose2(object):
    """gtsam.Pose2 stand-in for the reference functions cut out below (gtsam is absent from this image).  Written here,
    not imported from the product: a fixture "from the reference" must not depend on the code it pins.  gtsam keeps the
    rotation as (c, s), renormalises a product only when |c^2 + s^2 - 1| > 1e-10 (Rot2::normalize), theta = atan2(s, c)."""

    def __init__(self, x=0.0, y=0.0, theta=0.0, cs=None):
        import math
        self._x, self._y = float(x), float(y)
        c, s = (math.cos(theta), math.sin(theta)) if cs is None else cs
        n = c * c + s * s
        if cs is not None and abs(n - 1.0) > 1e-10:
            c, s = c / math.sqrt(n), s / math.sqrt(n)
        self._c, self._s = c, s

    x = lambda self: self._x
    y = lambda self: self._y

    def theta(self):
        import math
        return math.atan2(self._s, self._c)

    def compose(self, o):
        return Pose2(self._x + self._c * o._x - self._s * o._y, self._y + self._s * o._x + self._c * o._y,
                     cs=(self._c * o._c - self._s * o._s, self._s * o._c + self._c * o._s))

    def inverse(self):
        return Pose2(-(self._c * self._x + self._s * self._y), -(-self._s * self._x + self._c * self._y), cs=(self._c, -self._s))

    def between(self, o):
        return self.inverse().compose(o)

    def matrix(self):
        return np.array([[self._c, -self._s, self._x], [self._s, self._c, self._y], [0.0, 0.0, 1.0]])

    def rotation(self):
        return types.SimpleNamespace(matrix=lambda: np.array([[self._c, -self._s], [self._s, self._c]]))

    def translation(self):
        return np.array([self._x, self._y])


def reference_cfar_class():
    src = open(os.path.join(REF, "CFAR.py")).read()
    keep = [ln for ln in src.splitlines()
            if not ln.startswith(("from .utils", "from .sonar", "from . import cfar"))]
    stub = types.SimpleNamespace(**{n: None for n in
                                    ("ca", "soca", "goca", "os", "ca2", "soca2", "goca2", "os2")})
    ns = {"cfar": stub}
    exec(compile("\n".join(keep), "reference:CFAR.py", "exec"), ns)
    return ns["CFAR"]


def reference_generate_map_xy():
    src = open(os.path.join(REF, "feature_extraction.py")).read()
    tree = ast.parse(src)
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name == "generate_map_xy":
            fn_src = ast.get_source_segment(src, node)
            break
    else:
        raise RuntimeError("generate_map_xy not found")
    import textwrap
    from scipy.interpolate import interp1d
    ns = {"np": np, "interp1d": interp1d}
    exec(compile(textwrap.dedent(fn_src), "reference:feature_extraction.py", "exec"), ns)
    return ns["generate_map_xy"]


def _cut(path, name):
    src = open(os.path.join(REF, path)).read()
    for node in ast.walk(ast.parse(src)):
        if isinstance(node, ast.FunctionDef) and node.name == name:
            import textwrap
            seg = textwrap.dedent(ast.get_source_segment(src, node))
            return seg
    raise RuntimeError("%s not found in %s" % (name, path))


def reference_matching_cost():
    """-> (get_matching_cost_subroutine1 as a plain function of a stub `self`, Keyframe stub)"""
    sys.path.insert(0, os.path.join(HERE, "..", ".."))
    import oracle

    tp_src = _cut("slam_objects.py", "transform_points").replace("@staticmethod", "")
    ns_k = {"np": np, "gtsam": types.SimpleNamespace(Pose2=Pose2)}
    exec(compile(tp_src, "reference:slam_objects.py", "exec"), ns_k)
    Keyframe = types.SimpleNamespace(transform_points=ns_k["transform_points"])

    _cv2 = thirdparty.cv2(oracle)   # (OpenCV is not in this image: then the two calls come from the oracle's restatement)

    def n2g(x, kind):
        assert kind == "Pose2"
        return Pose2(x[0], x[1], x[2])

    def g2n(p):
        return np.array([p.x(), p.y(), p.theta()])

    fn_src = _cut("slam.py", "get_matching_cost_subroutine1")
    from typing import Union
    ns = {"np": np, "cv2": _cv2, "gtsam": types.SimpleNamespace(Pose2=Pose2), "Keyframe": Keyframe, "n2g": n2g,
          "g2n": g2n, "Union": Union}
    exec(compile(fn_src, "reference:slam.py", "exec"), ns)
    return ns["get_matching_cost_subroutine1"], Keyframe, Pose2


class _Self(object):
    """The attributes generate_map_xy reads (feature_extraction.py:58-70)."""

    def __init__(self):
        self.res = self.height = self.rows = self.width = self.cols = None
        self.map_x = self.map_y = None
        self.to_rad = lambda bearing: bearing * np.pi / 18000
        self.REVERSE_Z = 1


class _Ping(object):
    def __init__(self, bearings, res, num_ranges):
        self.bearings = [int(b) for b in bearings]  # ROS delivers a tuple of ints
        self.range_resolution = res
        self.num_ranges = num_ranges


def _cut_block(path, first_marker, last_marker):
    """the source lines of a block inside a function, from the line containing first_marker to the one containing last_marker,
    dedented (for code that is not a function of its own in the reference)"""
    import textwrap
    lines = open(os.path.join(REF, path)).read().split("\n")
    a = next(i for i, l in enumerate(lines) if first_marker in l)
    b = next(i for i, l in enumerate(lines) if last_marker in l and i >= a)
    return textwrap.dedent("\n".join(lines[a:b + 1])), (a + 1, b + 1)


def make_nssm_pieces():
    """-> nssm_pieces.npz: the reference's own lines of the loop-closure search that decide WHICH points and WHICH guesses are used"""
    import textwrap
    out = {}
    _, Keyframe, _ = reference_matching_cost()
    # ---- field-of-view gate + counts per keyframe (slam.py:875-904) ----
    block, span = _cut_block("slam.py", "sel = np.zeros(len(target_points), np.bool)", "counts = counts[counts > 10]")
    np_ns = types.SimpleNamespace(**{k: getattr(np, k) for k in dir(np) if not k.startswith("__")})
    np_ns.bool = bool                        # (np.bool is gone from numpy 2: the alias the reference's numpy had)
    rng = np.random.default_rng(11)
    n_kf = 14
    poses = [Pose2(20.0 * np.cos(a) + rng.normal(0, 0.2), 20.0 * np.sin(a) + rng.normal(0, 0.2), a + np.pi / 2 + rng.normal(0, 0.05))
             for a in np.linspace(0, 2 * np.pi * 14 / 13, n_kf)]
    covs = []
    for k in range(n_kf):
        A = rng.normal(0, 0.2, (3, 3))
        covs.append(A @ A.T + np.diag([0.01, 0.01, 1e-4]) * (k + 1))
    keyframes = [types.SimpleNamespace(pose=p, cov=c) for p, c in zip(poses, covs)]
    target_points = np.c_[rng.uniform(-45, 45, 6000), rng.uniform(-45, 45, 6000)].astype(np.float32)
    target_keys = rng.integers(0, 6, 6000).astype(np.float32)          # (keys travel as a float descriptor row: pcl.cpp:143-159)
    target_keys[rng.random(6000) < 0.004] = 5.0                          # a keyframe with a handful of points: dropped by counts > 10
    target_keys[(target_keys == 5.0) & (rng.random(6000) < 0.995)] = 4.0
    source_frames = [13, 12, 11, 10, 9]
    ns = {"np": np_ns, "Keyframe": Keyframe, "source_frames": source_frames, "target_points": target_points.copy(),
          "target_keys": target_keys.copy(),
          "self": types.SimpleNamespace(keyframes=keyframes, oculus=types.SimpleNamespace(max_range=30.0, horizontal_aperture=np.radians(130.0)))}
    exec(compile(block, "reference:slam.py:%d-%d" % span, "exec"), ns)
    out.update(fov_poses=np.array([[p.x(), p.y(), p.theta()] for p in poses]), fov_covs=np.array(covs),
               fov_source_frames=np.array(source_frames), fov_max_range=30.0, fov_aperture=np.radians(130.0),
               fov_target_points=target_points, fov_target_keys=target_keys, fov_sel=np.asarray(ns["sel"], bool),
               fov_kept_points=np.asarray(ns["target_points"]), fov_kept_keys=np.asarray(ns["target_keys"]),
               fov_frames=np.asarray(ns["target_frames"]), fov_counts=np.asarray(ns["counts"]),
               fov_lines=np.array("slam.py:%d-%d" % span))
    # ---- ICPResult.__init__ (slam_objects.py:247-300): initial transform and the filtered list of sampled transforms ----
    src = open(os.path.join(REF, "slam_objects.py")).read()
    cls = next(n for n in ast.walk(ast.parse(src)) if isinstance(n, ast.ClassDef) and n.name == "ICPResult")
    init = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "__init__")
    fn_src = textwrap.dedent(ast.get_source_segment(src, init))
    ns2 = {"np": np, "InitializationResult": object, "n2g": lambda g, kind: Pose2(*g),
           "g2n": lambda p: np.array([p.x(), p.y(), p.theta()])}
    exec(compile(fn_src, "reference:slam_objects.py", "exec"), ns2)
    source_pose, target_pose = Pose2(3.0, -1.0, 0.4), Pose2(2.2, -0.7, 0.1)
    est = source_pose.compose(Pose2(0.31, -0.12, 0.02))
    base = rng.normal(0, [0.6, 0.6, 0.05], (60, 3))
    deltas, costs = [], []
    for i, b in enumerate(base):                  # near-duplicates NEXT to their original in cost order: the filter's business
        for j in range(int(rng.integers(1, 6))):
            deltas.append(b + (rng.normal(0, [0.003, 0.003, 0.0005]) if j else 0.0))
            costs.append(-1000.0 + 10 * i + j)                                         # distinct costs: argsort has one answer
    order = rng.permutation(len(deltas))          # (the list itself comes in evaluation order, not in cost order)
    deltas, costs = np.array(deltas)[order], np.array(costs)[order]
    sample_poses = [source_pose.compose(Pose2(*d)) for d in deltas]
    samples = np.array([[p.x(), p.y(), p.theta(), c] for p, c in zip(sample_poses, costs)])
    init_ret = types.SimpleNamespace(source_points=None, target_points=None, source_key=13, target_key=2, source_pose=source_pose,
                                     target_pose=target_pose, status=None, estimated_source_pose=est, source_pose_samples=samples)
    res = types.SimpleNamespace()
    ns2["__init__"](res, init_ret, True, 0.01)
    g = lambda p: [p.x(), p.y(), p.theta()]
    out.update(icp_samples=samples, icp_source_pose=np.array(g(source_pose)), icp_target_pose=np.array(g(target_pose)),
               icp_estimated_source_pose=np.array(g(est)), icp_initial_transform=np.array(g(res.initial_transform)),
               icp_initial_transforms=np.array([g(t) for t in res.initial_transforms]), icp_sample_eps=0.01)
    # ---- get_points(frames, None, return_keys=True) (slam.py:229-292): the keyed global target cloud of the search ----
    # pcl.downsample(points, keys, res) = the oracle's octree with the index of every leaf's medoid (that one stays unpinned); the
    # transform to the SLAM frame, the key column, the concatenation order and the float32 rounding at the pybind boundary are the
    # reference's own code.  pcl.match = the oracle's exact nearest neighbour; pcl.ICP.compute = the oracle's chain.
    import oracle as _orc
    from typing import Any, Union
    from sonar_slam_amd import synth

    pcl_mod = thirdparty.pcl(_orc)
    gp_src = _cut("slam.py", "get_points")
    ns_gp = {"np": np, "gtsam": types.SimpleNamespace(Pose2=Pose2), "Keyframe": Keyframe, "Any": Any,
             "pcl": pcl_mod}
    exec(compile(gp_src, "reference:slam.py", "exec"), ns_gp)
    rng = np.random.default_rng(23)
    kf_clouds = [np.c_[rng.uniform(1, 29, n), rng.uniform(-20, 20, n)].astype(np.float32).astype(np.float64) for n in (900, 1, 1400, 0, 650, 1100)]
    kf_poses = [(0.0, 0.0, 0.0), (1.7, -0.2, 0.05), (3.1, 0.4, 0.13), (4.9, 0.1, 0.2), (6.2, -0.7, 0.31), (7.0, -1.6, 0.52)]
    kfs = [types.SimpleNamespace(points=c, pose=Pose2(*q)) for c, q in zip(kf_clouds, kf_poses)]
    for k in kfs:
        k.transf_points = Keyframe.transform_points(k.points, k.pose)          # Keyframe.update (slam_objects.py:160)
    slam = types.SimpleNamespace(keyframes=kfs, current_key=len(kfs), point_resolution=0.5)
    frames_k = [0, 1, 2, 3, 4]
    gpts, gkeys = ns_gp["get_points"](slam, frames_k, None, True)
    out.update({"keyed_cloud%d" % i: c for i, c in enumerate(kf_clouds)})
    out.update(keyed_poses=np.array(kf_poses), keyed_frames=np.array(frames_k), keyed_points=np.asarray(gpts, np.float32),
               keyed_keys=np.asarray(gkeys, np.float32).reshape(-1), keyed_resolution=0.5)
    # ---- get_overlap (slam.py:389-424) ----
    ov_src = _cut("slam.py", "get_overlap")
    ns_ov = {"np": np, "gtsam": types.SimpleNamespace(Pose2=Pose2), "Keyframe": Keyframe,
             "pcl": pcl_mod}
    exec(compile(ov_src, "reference:slam.py", "exec"), ns_ov)
    src_o, tgt_o, guess_o, truth_o = synth.scan_pair(seed=33, n_src=800, n_tgt=900)
    me = types.SimpleNamespace(point_noise=0.5)
    pose_o = Pose2(*synth.pose_of(truth_o))
    n64, idx64 = ns_ov["get_overlap"](me, src_o.astype(np.float64), tgt_o, pose_o, None, True)       # keyframe cloud: float64
    n32 = ns_ov["get_overlap"](me, src_o, tgt_o, pose_o)                                             # aggregated cloud: float32
    n_none = ns_ov["get_overlap"](me, src_o, tgt_o)
    out.update(ov_source=src_o, ov_target=tgt_o, ov_pose=np.array(synth.pose_of(truth_o)), ov_count_f64=int(n64),
               ov_indices_f64=np.asarray(idx64).reshape(-1), ov_count_f32=int(n32), ov_count_no_pose=int(n_none), ov_point_noise=0.5)
    # ---- compute_icp_with_cov (slam.py:325-387): many guesses on one pair, MinCovDet, the covariance in the centre's frame ----
    cov_src = _cut("slam.py", "compute_icp_with_cov")
    import time as time_pkg
    from sklearn.covariance import MinCovDet
    icp_compute = thirdparty.icp_compute(_orc, ICP_YAML)
    ns_cv = {"np": np, "Union": Union, "time_pkg": time_pkg, "MinCovDet": MinCovDet, "n2g": lambda g, kind: Pose2(*g)}
    exec(compile(cov_src, "reference:slam.py", "exec"), ns_cv)
    src_c, tgt_c, guess_c, truth_c = synth.scan_pair(seed=9, n_src=1500, n_tgt=1500)
    gx, gy, gt = synth.pose_of(guess_c)
    rng = np.random.default_rng(1)
    gs = np.array([[gx + dx, gy + dy, gt + dt] for dx, dy, dt in rng.normal(0, [0.2, 0.2, 0.02], (30, 3))])
    me = types.SimpleNamespace(icp=types.SimpleNamespace(compute=icp_compute), icp_odom_sigmas=np.array([0.1, 0.1, 0.01]))
    np.random.seed(0)                       # MinCovDet(random_state=None) draws from numpy's global generator
    msg, m, cov, samples_c = ns_cv["compute_icp_with_cov"](me, src_c, tgt_c, [Pose2(*g) for g in gs])
    np.random.seed(0)
    msg_few, *_ = ns_cv["compute_icp_with_cov"](me, src_c, tgt_c, [Pose2(*g) for g in gs[:3]])
    tiny = types.SimpleNamespace(icp=types.SimpleNamespace(compute=icp_compute), icp_odom_sigmas=np.array([1e-4, 1e-4, 1e-5]))
    np.random.seed(0)
    _, m2, cov2, _ = ns_cv["compute_icp_with_cov"](tiny, src_c, tgt_c, [Pose2(*g) for g in gs])    # sigmas below the scatter: MinCovDet's own
    out.update(cov_source=src_c, cov_target=tgt_c, cov_guesses=gs, cov_message=np.array(msg), cov_message_3_guesses=np.array(msg_few),
               cov_centre=np.array([m.x(), m.y(), m.theta()]), cov_cov=np.asarray(cov), cov_samples=np.asarray(samples_c),
               cov_cov_small_sigmas=np.asarray(cov2), cov_sigmas=np.array([0.1, 0.1, 0.01]), cov_small_sigmas=np.array([1e-4, 1e-4, 1e-5]))
    # ---- is_keyframe (slam.py:1134-1161): which pings become keyframes ----
    kf_src = _cut("slam.py", "is_keyframe")
    ns_kf = {"np": np, "Keyframe": object}
    exec(compile(kf_src, "reference:slam.py", "exec"), ns_kf)
    rng = np.random.default_rng(3)
    last = types.SimpleNamespace(time=10.0, dr_pose=Pose2(4.0, -2.0, 0.7))

    class _S(object):
        keyframes = [last]
        current_keyframe = last
        keyframe_duration, keyframe_translation, keyframe_rotation = 1.0, 3.0, np.radians(30.0)
    cases, flags = [], []
    for _ in range(400):
        kind = rng.integers(4)
        t = 10.0 + float(rng.choice([0.5, 0.999, 1.0, 1.5, 4.0]))
        d = Pose2(*(rng.normal(0, [2.5, 2.5, 0.4]) if kind else (3.0 * np.cos(0.3), 3.0 * np.sin(0.3), 0.0)))   # (on the 3 m gate)
        frame = types.SimpleNamespace(time=t, dr_pose=last.dr_pose.compose(d))
        cases.append([t, frame.dr_pose.x(), frame.dr_pose.y(), frame.dr_pose.theta()])
        flags.append(bool(ns_kf["is_keyframe"](_S(), frame)))
    empty = type("E", (), {"keyframes": []})()
    out.update(kf_last=np.array([10.0, 4.0, -2.0, 0.7]), kf_cases=np.array(cases), kf_flags=np.array(flags),
               kf_first=bool(ns_kf["is_keyframe"](empty, None)))
    out["stand_ins"] = thirdparty.record()
    np.savez_compressed(os.path.join(HERE, "nssm_pieces.npz"), **out)
    print("wrote nssm_pieces.npz (field-of-view gate of %s: %d of %d points kept, keyframes %r; ICPResult: %d of %d sampled transforms kept; "
          "keyed target cloud: %d points from %d; overlap %d / %d / %d; compute_icp_with_cov: %s, %d of 30 guesses converged, det(cov) %.3g / %.3g)"
          % (span, int(ns["sel"].sum()), len(target_points), list(np.asarray(ns["target_frames"])), len(res.initial_transforms), len(samples),
             len(gpts), sum(len(c) for c in kf_clouds[:5]), n64, n32, n_none, msg, len(samples_c), np.linalg.det(cov), np.linalg.det(cov2)))


def _cut_class(path, name):
    import textwrap
    src = open(os.path.join(REF, path)).read()
    for node in ast.walk(ast.parse(src)):
        if isinstance(node, ast.ClassDef) and node.name == name:
            return textwrap.dedent(ast.get_source_segment(src, node))
    raise RuntimeError("class %s not found in %s" % (name, path))


def make_ssm_session():
    """-> ssm_session.npz: the reference's OWN sequential-scan-matching methods (slam.py: initialize_sequential_scan_matching,
    add_sequential_scan_matching, add_odometry, get_points, get_matching_cost_subroutine1, compute_icp, get_overlap; slam_objects.py:
    STATUS, InitializationResult, ICPResult, Keyframe.transform_points) run on a synthetic session of keyframe clouds, by the
    reference's defaults (global initialisation with scipy's shgo ON), with stand-ins only for what is not in this image: pcl
    (downsample / match / ICP.compute = the oracle), cv2 (element / dilate = the oracle), gtsam (Pose2 = the class above; factors and
    noise models = records) and ISAM2 (the new keyframe keeps the initial value its factor was inserted with -- what a chain of
    between factors optimises to).  Recorded per keyframe: status, its description, shgo's estimated source pose, the ICP transform,
    the overlap, whether a scan-matching factor went in, and the pose the keyframe ends with."""
    import contextlib
    from enum import Enum
    from typing import Any, Union
    from scipy.optimize import shgo
    sys.path.insert(0, os.path.join(HERE, "..", ".."))
    import oracle as _orc
    from oracle import chain
    from sonar_slam_amd import synth
    from sonar_slam_amd.CFAR import CFAR
    from sonar_slam_amd.feature_extraction import build_maps, oculus_bearings
    cost_fn, Keyframe, _ = reference_matching_cost()
    n2g = lambda g, kind: Pose2(*g)
    g2n = lambda p: np.array([p.x(), p.y(), p.theta()])
    ns_o = {"np": np, "Enum": Enum, "Any": Any, "Union": Union, "n2g": n2g, "g2n": g2n, "gtsam": types.SimpleNamespace(Pose2=Pose2)}
    for cls in ("STATUS", "InitializationResult", "ICPResult"):
        exec(compile(_cut_class("slam_objects.py", cls), "reference:slam_objects.py", "exec"), ns_o)
    STATUS = ns_o["STATUS"]
    icp_compute = thirdparty.icp_compute(_orc, ICP_YAML)
    factors = []
    gt = types.SimpleNamespace(Pose2=Pose2, BetweenFactorPose2=lambda a, b, t, model: factors.append(("between", a, b, t, model)) or ("between", a, b),
                               PriorFactorPose2=lambda a, p, model: ("prior", a))

    @contextlib.contextmanager
    def CodeTimer(name):
        yield
    ns = {"np": np, "gtsam": gt, "shgo": shgo, "CodeTimer": CodeTimer, "n2g": n2g, "g2n": g2n, "X": lambda k: k, "STATUS": STATUS,
          "InitializationResult": ns_o["InitializationResult"], "ICPResult": ns_o["ICPResult"], "Keyframe": Keyframe, "Any": Any,
          "Union": Union,
          "pcl": thirdparty.pcl(_orc)}
    for name in ("initialize_sequential_scan_matching", "add_sequential_scan_matching", "add_odometry", "get_points", "compute_icp",
                 "get_overlap"):
        exec(compile(_cut("slam.py", name), "reference:slam.py", "exec"), ns)

    class Stamp(float):
        def __sub__(self, o):
            return types.SimpleNamespace(to_sec=lambda d=float(self) - float(o): d)

    class Slam(object):
        current_key = property(lambda self: len(self.keyframes))
        current_keyframe = property(lambda self: self.keyframes[-1])
    S = Slam()
    S.keyframes, S.graph, S.values = [], [], {}
    S.graph = types.SimpleNamespace(add=lambda f: None)
    S.values = types.SimpleNamespace(insert=lambda k, p: inserted.__setitem__(k, p))
    inserted = {}
    S.ssm_params = types.SimpleNamespace(enable=True, min_points=20, max_translation=3.0, max_rotation=np.pi / 6, target_frames=3,
                                         initialization=True, initialization_params=(50, 1, 0.01), cov_samples=0)
    S.odom_sigmas, S.point_resolution, S.point_noise = np.array([0.2, 0.2, 0.02]), 0.5, 0.5
    S.icp = types.SimpleNamespace(compute=icp_compute)
    S.odom_model, S.icp_odom_model = "odometry", "icp"
    S.save_data = S.save_fig = False
    S.create_full_noise_model = lambda cov: ("cov", cov)
    for name in ("initialize_sequential_scan_matching", "add_sequential_scan_matching", "add_odometry", "get_points", "compute_icp",
                 "get_overlap"):
        setattr(S, name, types.MethodType(ns[name], S))
    S.get_matching_cost_subroutine1 = types.MethodType(cost_fn, S)
    last_init = {}
    inner = S.initialize_sequential_scan_matching

    def spy(keyframe):
        ret = inner(keyframe)
        last_init["ret"] = ret
        last_init["status"] = (ret.status.name, ret.status.description)
        return ret
    S.initialize_sequential_scan_matching = spy
    # ---- the session: pings of a synthetic scene -> SLAM-node clouds (oracle feature extraction), drifting odometry ----
    K, rows, beams = 7, 256, 128
    bearings = oculus_bearings(beams)
    res, height, _, width, cols, mx, my = build_maps(bearings, 30.0 / rows, rows)
    fe = types.SimpleNamespace(map_x=mx, map_y=my, rows=rows, cols=cols, width=width, height=height)
    world = synth.world_structure(seed=2, n=5000)
    true, dr = synth.trajectory(n=K, step=1.7, turn=0.04, seed=11, start=(2.0, 0.0, 0.0))
    det = CFAR(40, 10, 0.1, 10)
    clouds = []
    for k in range(K):
        img = synth.render_ping(world, true[k], bearings, rows=rows, seed=k)
        # (float64 arrays of float32 values: ros_numpy's pointcloud2_to_xyz_array hands the SLAM node doubles, slam_ros.py:169-170)
        clouds.append(chain.slam_cloud(chain.feature_cloud(img, det.params["SOCA"], "SOCA", 65, fe)[1]).astype(np.float64))
    clouds[4] = clouds[4][:12]                       # a keyframe with too few points: the odometry factor goes in
    out = {"K": K, "dr": np.array(dr)}
    for k, c in enumerate(clouds):
        out["cloud%d" % k] = c
    summary = []
    for tag, min_points, max_translation in (("a", 20, 3.0), ("b", 20, 0.05), ("c", 600, 3.0)):
        S.keyframes = []
        S.ssm_params.min_points, S.ssm_params.max_translation = min_points, max_translation
        out[tag + "_min_points"], out[tag + "_max_translation"] = min_points, max_translation
        recs = []
        for k in range(K):
            frame = types.SimpleNamespace(time=Stamp(k), dr_pose=Pose2(*dr[k]), points=clouds[k], pose=Pose2(*dr[k]))
            if S.keyframes:                               # slam_ros.py:181-184
                frame.pose = S.current_keyframe.pose.compose(S.current_keyframe.dr_pose.between(frame.dr_pose))
            inserted.clear()
            del factors[:]
            last_init.clear()
            for m in STATUS:                              # (the members carry the description of whoever used them last)
                m.description = None
            overlaps = []
            rec = {"k": k}
            if not S.keyframes:
                inserted[0] = frame.pose                  # add_prior (slam.py:426-436)
                rec["status"] = "PRIOR"
            else:
                S.add_sequential_scan_matching(frame)
                ret = last_init["ret"]
                f = factors[-1]
                rec["init_status"], rec["init_description"] = last_init["status"]
                rec["estimated_source_pose"] = g2n(ret.estimated_source_pose) if ret.estimated_source_pose is not None else None
                rec["n_source"], rec["n_target"] = len(ret.source_points), len(ret.target_points)
                rec["target_points"] = np.asarray(ret.target_points)
                # what went into the graph: the last between factor of this keyframe (scan match or odometry: both run target -> source)
                rec["factor_transform"] = g2n(f[3])
                rec["factor_is_odometry"] = f[4] == "odometry"
                touched = {m.name: m.description for m in STATUS if m.description is not None}
                failed = [n for n in touched if n != "SUCCESS"]
                rec["status"] = failed[0] if rec["factor_is_odometry"] and failed else ("SUCCESS" if not rec["factor_is_odometry"] else rec["init_status"])
                rec["description"] = touched.get(rec["status"])
            # update_factor_graph with a chain of between factors: the keyframe takes the value inserted for it
            frame.pose = inserted[k]
            S.keyframes.append(frame)
            rec["pose"] = g2n(frame.pose)
            recs.append(rec)
        for r in recs:
            k = r["k"]
            out["%s_pose%d" % (tag, k)] = r["pose"]
            if k:
                out["%s_init_status%d" % (tag, k)] = np.array(r["init_status"])
                out["%s_init_description%d" % (tag, k)] = np.array(str(r["init_description"]))
                out["%s_factor_transform%d" % (tag, k)] = r["factor_transform"]
                out["%s_factor_is_odometry%d" % (tag, k)] = r["factor_is_odometry"]
                out["%s_status%d" % (tag, k)] = np.array(r["status"])
                out["%s_description%d" % (tag, k)] = np.array(str(r["description"]))
                out["%s_n_source%d" % (tag, k)], out["%s_n_target%d" % (tag, k)] = r["n_source"], r["n_target"]
                out["%s_target_points%d" % (tag, k)] = r["target_points"]
                if r["estimated_source_pose"] is not None:
                    out["%s_estimated_source_pose%d" % (tag, k)] = r["estimated_source_pose"]
        summary.append((tag, [(r["k"], r.get("status"), r.get("description")) for r in recs]))
    out["stand_ins"] = thirdparty.record()
    np.savez_compressed(os.path.join(HERE, "ssm_session.npz"), **out)
    print("wrote ssm_session.npz:")
    for t in summary:
        print("  ", t)


def make_nssm_session():
    """-> nssm_session.npz: one session on a closed trajectory run by the reference's OWN methods, the loop-closure search included:
    initialize_nonsequential_scan_matching + add_nonsequential_scan_matching + compute_icp_with_cov next to the sequential-scan-
    matching methods of make_ssm_session, by the reference's defaults (slam.yaml's nssm block; both global initialisations ON).
    Stand-ins as there, plus: MinCovDet = sklearn's own with numpy's global generator seeded with 0 before every search (the
    oracle / the product pass random_state=0), PCM = nothing passes (verify_pcm -> []: loop factors never reach the graph, as with
    the chain back end), the marginal covariances of ISAM2 = oracle/chain.py::chain_covariance (an INPUT of the search), and
    self.current_frame moving on only after the search (slam_ros.py:207-211)."""
    import contextlib
    import time as time_pkg
    from enum import Enum
    from typing import Any, Union
    from scipy.optimize import shgo
    from sklearn.covariance import MinCovDet
    sys.path.insert(0, os.path.join(HERE, "..", ".."))
    import oracle as _orc
    from oracle import chain
    from sonar_slam_amd import synth
    from sonar_slam_amd.CFAR import CFAR
    from sonar_slam_amd.feature_extraction import build_maps, oculus_bearings
    cost_fn, Keyframe, _ = reference_matching_cost()
    n2g = lambda g, kind: Pose2(*g)
    g2n = lambda p: np.array([p.x(), p.y(), p.theta()])
    ns_o = {"np": np, "Enum": Enum, "Any": Any, "Union": Union, "n2g": n2g, "g2n": g2n, "gtsam": types.SimpleNamespace(Pose2=Pose2)}
    for cls in ("STATUS", "InitializationResult", "ICPResult"):
        exec(compile(_cut_class("slam_objects.py", cls), "reference:slam_objects.py", "exec"), ns_o)
    STATUS = ns_o["STATUS"]
    icp_compute = thirdparty.icp_compute(_orc, ICP_YAML)
    factors = []
    gt = types.SimpleNamespace(Pose2=Pose2, BetweenFactorPose2=lambda a, b, t, model: factors.append(("between", a, b, t, model)) or ("between", a, b))

    @contextlib.contextmanager
    def CodeTimer(name):
        yield
    np_ns = types.SimpleNamespace(**{k: getattr(np, k) for k in dir(np) if not k.startswith("__")})
    np_ns.bool = bool
    ns = {"np": np_ns, "gtsam": gt, "shgo": shgo, "CodeTimer": CodeTimer, "n2g": n2g, "g2n": g2n, "X": lambda k: k, "STATUS": STATUS,
          "InitializationResult": ns_o["InitializationResult"], "ICPResult": ns_o["ICPResult"], "Keyframe": Keyframe, "Any": Any,
          "Union": Union, "MinCovDet": MinCovDet,
          # (compute_icp_with_cov stops trying guesses after 2 s of wall clock, slam.py:356-358: a real-time guard that
          #  libpointmatcher never reaches with 30 guesses, but the oracle's ICP on this CPU does -- the clock stands still here)
          "time_pkg": types.SimpleNamespace(time=lambda: 0.0),
          "pcl": thirdparty.pcl(_orc)}
    names = ("initialize_sequential_scan_matching", "add_sequential_scan_matching", "add_odometry", "get_points", "compute_icp",
             "compute_icp_with_cov", "get_overlap", "initialize_nonsequential_scan_matching", "add_nonsequential_scan_matching")
    for name in names:
        exec(compile(_cut("slam.py", name), "reference:slam.py", "exec"), ns)

    class Stamp(float):
        def __sub__(self, o):
            return types.SimpleNamespace(to_sec=lambda d=float(self) - float(o): d)

    class Slam(object):
        current_key = property(lambda self: len(self.keyframes))
        current_keyframe = property(lambda self: self.keyframes[-1])
    S = Slam()
    inserted = {}
    S.keyframes, S.current_frame, S.nssm_queue, S.pcm_queue_size, S.min_pcm = [], None, [], 5, 2
    S.graph = types.SimpleNamespace(add=lambda f: None)
    S.values = types.SimpleNamespace(insert=lambda k, p: inserted.__setitem__(k, p))
    S.verify_pcm = lambda queue, min_pcm: []
    S.ssm_params = types.SimpleNamespace(enable=True, min_points=20, max_translation=3.0, max_rotation=np.pi / 6, target_frames=3,
                                         initialization=True, initialization_params=(50, 1, 0.01), cov_samples=0)
    S.nssm_params = types.SimpleNamespace(enable=True, min_st_sep=8, min_points=30, max_translation=10.0, max_rotation=np.pi / 3,
                                          source_frames=5, cov_samples=30, initialization=True, initialization_params=(100, 5, 0.01))
    S.oculus = types.SimpleNamespace(max_range=30.0, horizontal_aperture=np.radians(130.0))
    S.odom_sigmas, S.icp_odom_sigmas, S.point_resolution, S.point_noise = np.array([0.2, 0.2, 0.02]), np.array([0.1, 0.1, 0.01]), 0.5, 0.5
    S.icp = types.SimpleNamespace(compute=icp_compute)
    S.odom_model, S.icp_odom_model = "odometry", "icp"
    S.save_data = S.save_fig = False
    S.create_full_noise_model = lambda cov: ("cov", cov)
    for name in names:
        setattr(S, name, types.MethodType(ns[name], S))
    S.get_matching_cost_subroutine1 = types.MethodType(cost_fn, S)
    last = {}
    inner = S.initialize_nonsequential_scan_matching

    def spy():
        ret = inner()
        last["init"] = ret
        last["init_status"] = (ret.status.name, ret.status.description)
        return ret
    S.initialize_nonsequential_scan_matching = spy
    K, rows, beams = 15, 256, 128
    bearings = oculus_bearings(beams)
    res, height, _, width, cols, mx, my = build_maps(bearings, 30.0 / rows, rows)
    fe = types.SimpleNamespace(map_x=mx, map_y=my, rows=rows, cols=cols, width=width, height=height)
    world = synth.world_structure(seed=2, n=9000)
    true, dr = synth.trajectory(n=K, step=1.7, turn=2 * np.pi / 13, seed=21, start=(20.0, 0.0, 0.0))
    det = CFAR(40, 10, 0.1, 10)
    clouds = []
    for k in range(K):
        img = synth.render_ping(world, true[k], bearings, rows=rows, seed=k)
        clouds.append(chain.slam_cloud(chain.feature_cloud(img, det.params["SOCA"], "SOCA", 65, fe)[1]).astype(np.float64))
    out = {"K": K, "dr": np.array(dr), "ssm_min_points": 20, "nssm_min_points": 30}
    summary = []
    for k in range(K):
        out["cloud%d" % k] = clouds[k]
        frame = types.SimpleNamespace(time=Stamp(k), dr_pose=Pose2(*dr[k]), points=clouds[k], pose=Pose2(*dr[k]), cov=None, constraints=[])
        if S.keyframes:
            frame.pose = S.current_keyframe.pose.compose(S.current_keyframe.dr_pose.between(frame.dr_pose))
        inserted.clear()
        del factors[:]
        if not S.keyframes:
            inserted[0] = frame.pose
            kind = "prior"
        else:
            S.add_sequential_scan_matching(frame)
            kind = "odometry" if factors[-1][4] == "odometry" else "icp"
        frame.pose = inserted[k]
        frame.cov = chain.chain_covariance(S.keyframes[-1].cov if S.keyframes else None, kind)
        frame.transf_points = Keyframe.transform_points(frame.points, frame.pose)       # Keyframe.update (slam_objects.py:160)
        S.keyframes.append(frame)
        out["pose%d" % k] = g2n(frame.pose)
        if S.current_frame is not None:                                                  # slam_ros.py:207
            for m in STATUS:
                m.description = None
            last.clear()
            np.random.seed(0)
            ret2 = S.add_nonsequential_scan_matching()
            if "init" in last:
                ret = last["init"]
                out["search%d" % k] = True
                out["init_status%d" % k] = np.array(last["init_status"][0])
                out["init_description%d" % k] = np.array(str(last["init_status"][1]))
                out["n_source%d" % k] = len(ret.source_points)
                if ret2 is not None:
                    out["status%d" % k] = np.array(ret2.status.name)
                    out["description%d" % k] = np.array(str(ret2.status.description))
                    out["target_key%d" % k] = int(ret2.target_key)
                    out["n_target%d" % k] = len(ret2.target_points)
                    out["initial_transform%d" % k] = g2n(ret2.initial_transform)
                    out["n_guesses%d" % k] = len(ret2.initial_transforms[:30])
                    # the ICP inputs of the search and the guesses in the order THIS run tried them (among pose samples of equal cost
                    # the reference's order is an unstable argsort over shgo's evaluation order: it differs from run to run)
                    out["icp_source%d" % k] = np.asarray(ret2.source_points, np.float32)
                    out["icp_target%d" % k] = np.asarray(ret2.target_points, np.float32)
                    out["guesses%d" % k] = np.array([g2n(t) for t in ret2.initial_transforms[:30]])
                    ps = np.asarray(ret.source_pose_samples)
                    out["pose_samples%d" % k] = ps[np.lexsort((ps[:, 2], ps[:, 1], ps[:, 0], ps[:, 3]))]
                    if ret2.estimated_transform is not None:
                        out["transform%d" % k] = g2n(ret2.estimated_transform)
                        out["cov%d" % k] = np.asarray(ret2.cov)
                        out["sample_transforms%d" % k] = np.asarray(ret2.sample_transforms)
                summary.append((k, last["init_status"], None if ret2 is None else (ret2.status.name, ret2.status.description, int(ret2.target_key))))
        S.current_frame = frame
    out["stand_ins"] = thirdparty.record()
    np.savez_compressed(os.path.join(HERE, "nssm_session.npz"), **out)
    print("wrote nssm_session.npz:")
    for t in summary:
        print("  ", t)


def make_feature_callback():
    """-> feature_callback.npz: the body of FeatureExtraction.callback from the detection to the filtered cloud (feature_extraction.py:
    222-249, cut out by its first and last line) run on synthetic pings with the reference's own CFAR class calling the reference's
    own compiled cfar.cpp (oracle/_ref), its own generate_map_xy maps, and the oracle standing in for cv2.remap and pcl: the intensity
    gate, np.nonzero's order, the pixel -> metre arithmetic and the order of the two filters are the reference's lines."""
    sys.path.insert(0, os.path.join(HERE, "..", ".."))
    import oracle as _orc
    from sonar_slam_amd import synth
    if not _orc.have_ref_cfar():
        sys.exit("oracle/_ref/libcfar_ref.so missing: run `make -C oracle ref` first")
    block, span = _cut_block("feature_extraction.py", "peaks = self.detector.detect(img, self.alg)", "self.outlier_filter_radius, self.outlier_filter_min_points")
    block += "\n    )\n"                     # (the call's closing parenthesis sits on the next source line)
    CFAR = reference_cfar_class()
    gen = reference_generate_map_xy()
    out, summary = {}, []
    for i, (rows, beams, n_blobs, thr, res, radius, min_pts) in enumerate([(256, 128, 10, 65, 0.5, 1.0, 5), (512, 256, 25, 65, 0.5, 1.0, 5),
                                                                           (192, 64, 6, 40, 0.25, 0.5, 3), (256, 128, 10, 65, 0.0, 1.0, 1)]):
        img = synth.sonar_frame(seed=90 + i, rows=rows, cols=beams, n_blobs=n_blobs)
        det = CFAR(40, 10, 0.1, 10)
        for alg in ("CA", "SOCA", "GOCA", "OS"):
            det.detector[alg] = (lambda a: lambda mat, th, gh, tau, k=0: _orc.ref_cfar(mat, a, th, gh, tau, k))(alg)
        me = _Self()
        ping = _Ping(bearings_for(beams), 30.0 / rows, rows)
        gen(me, ping)
        me.detector, me.alg, me.threshold = det, "SOCA", thr
        me.resolution, me.outlier_filter_radius, me.outlier_filter_min_points = res, radius, min_pts
        me.feature_img_pub = types.SimpleNamespace(publish=lambda m: None)
        ns = {"np": np, "self": me, "img": img.copy(), "sonar_msg": ping,
              "cv2": thirdparty.cv2(_orc),
              "ros_numpy": types.SimpleNamespace(image=types.SimpleNamespace(numpy_to_image=lambda a, enc: a)),
              "pcl": thirdparty.pcl(_orc)}
        exec(compile(block, "reference:feature_extraction.py:%d-%d" % span, "exec"), ns)
        out.update({"img%d" % i: img, "bearings%d" % i: np.asarray(ping.bearings, np.int16), "range_resolution%d" % i: ping.range_resolution,
                    "threshold%d" % i: thr, "resolution%d" % i: res, "radius%d" % i: radius, "min_points%d" % i: min_pts,
                    "locs%d" % i: np.asarray(ns["locs"]), "points%d" % i: np.asarray(ns["points"]),
                    "points_dtype%d" % i: np.array(str(np.asarray(ns["points"]).dtype)),
                    "raw_xy%d" % i: np.column_stack((ns["y"], ns["x"]))})
        summary.append((rows, beams, len(ns["locs"]), len(ns["points"]), str(np.asarray(ns["points"]).dtype)))
    out["n"] = len(summary)
    out["lines"] = np.array("feature_extraction.py:%d-%d" % span)
    out["stand_ins"] = thirdparty.record()
    np.savez_compressed(os.path.join(HERE, "feature_callback.npz"), **out)
    print("wrote feature_callback.npz (%s):" % out["lines"], summary)


def bearings_for(n, aperture_deg=130.0):
    half = aperture_deg * 50.0
    return np.round(np.linspace(-half, half, n)).astype(np.int16)


def main():
    if not os.path.isdir(REF):
        sys.exit("reference tree not present; fixtures are committed, nothing to do")
    if sys.argv[1:] == ["nssm"]:
        make_nssm_pieces()
        return
    if sys.argv[1:] == ["ssm"]:
        make_ssm_session()
        return
    if sys.argv[1:] == ["nssm_session"]:
        make_nssm_session()
        return
    if sys.argv[1:] == ["callback"]:
        make_feature_callback()
        return

    # ---- tau ----
    CFAR = reference_cfar_class()
    taus = []
    for Ntc, Ngc, Pfa, rank in [(40, 10, 0.1, 10), (40, 10, 1e-2, 20), (20, 4, 0.05, 5),
                                (32, 8, 1e-3, 16), (16, 2, 0.2, 3)]:
        c = CFAR(Ntc, Ngc, Pfa, rank)
        taus.append({"Ntc": Ntc, "Ngc": Ngc, "Pfa": Pfa, "rank": rank,
                     "CA": float(c.threshold_factor_CA), "SOCA": float(c.threshold_factor_SOCA),
                     "GOCA": float(c.threshold_factor_GOCA), "OS": float(c.threshold_factor_OS),
                     "params_SOCA": [int(c.params["SOCA"][0]), int(c.params["SOCA"][1])],
                     "params_OS_rank": c.params["OS"][2]})
    json.dump(taus, open(os.path.join(HERE, "cfar_tau.json"), "w"), indent=1)

    # ---- maps ----
    gen = reference_generate_map_xy()
    s = _Self()
    small = _Ping(bearings_for(64), 0.25, 96)
    gen(s, small)
    np.savez_compressed(os.path.join(HERE, "maps_small.npz"), map_x=s.map_x, map_y=s.map_y,
                        bearings=np.asarray(small.bearings, np.int16), res=small.range_resolution,
                        num_ranges=small.num_ranges, width=s.width, height=s.height, cols=s.cols)
    digests = []
    for nb, nr, res in [(512, 1024, 30.0 / 1024), (1024, 2048, 30.0 / 2048)]:
        s = _Self()
        p = _Ping(bearings_for(nb), res, nr)
        gen(s, p)
        rng = np.random.default_rng(7)
        idx = rng.integers(0, s.map_x.size, 64)
        digests.append({"beams": nb, "ranges": nr, "res": res, "cols": int(s.cols),
                        "width": float(s.width), "height": float(s.height),
                        "sha256_map_x": hashlib.sha256(s.map_x.tobytes()).hexdigest(),
                        "sha256_map_y": hashlib.sha256(s.map_y.tobytes()).hexdigest(),
                        "sample_idx": idx.tolist(),
                        "sample_map_x": [float(v) for v in s.map_x.ravel()[idx]],
                        "sample_map_y": [float(v) for v in s.map_y.ravel()[idx]]})
    json.dump(digests, open(os.path.join(HERE, "maps_digest.json"), "w"), indent=1)
    # ---- global-initialisation matching cost ----
    fn, Keyframe, Pose2 = reference_matching_cost()
    from sonar_slam_amd import synth
    src, tgt, guess, truth = synth.scan_pair(seed=41, n_src=1200, n_tgt=1300)
    self = types.SimpleNamespace(point_noise=0.5)
    sp, tp = Pose2(*synth.pose_of(truth)), Pose2(0.0, 0.0, 0.0)
    rng = np.random.default_rng(5)
    X = np.c_[rng.uniform(-1, 1, 48), rng.uniform(-1, 1, 48), rng.uniform(-0.2, 0.2, 48)]
    X[0] = 0
    subroutine, samples = fn(self, src, sp, tgt, tp, np.eye(3))
    costs = np.array([subroutine(x) for x in X], np.int64)
    T = tp.between(sp.compose(Pose2(*X[7]))).matrix()
    pts7 = Keyframe.transform_points(src, types.SimpleNamespace(matrix=lambda: T))
    np.savez_compressed(os.path.join(HERE, "matching_cost.npz"), src=src, tgt=tgt, source_pose=np.array(synth.pose_of(truth)),
                        X=X, costs=costs, samples=np.array(samples), points_pose7=pts7, point_noise=0.5,
                        stand_ins=thirdparty.record())
    # ---- Keyframe.transform_points / SLAM.get_points: the scan matcher's inputs ----
    # transform_points on float64 keyframe clouds holding float32 values (what ros_numpy hands the SLAM node,
    # slam_ros.py:169-170) and on float32 ones (sgemm); get_points (slam.py:229-292) cut by AST with
    # pcl.downsample = the oracle's octree (that one stays unpinned) -- the transform, the frame order of the
    # concatenation and the float32 rounding at the pybind boundary are the reference's own code on this numpy.
    from typing import Any
    gp_src = _cut("slam.py", "get_points")
    ns_gp = {"np": np, "gtsam": types.SimpleNamespace(Pose2=Pose2), "Keyframe": Keyframe, "Any": Any,
             "pcl": thirdparty.pcl(__import__("oracle"))}
    exec(compile(gp_src, "reference:slam.py", "exec"), ns_gp)
    rng = np.random.default_rng(77)
    clouds32 = [np.c_[rng.uniform(1, 29, n), rng.uniform(-20, 20, n)].astype(np.float32) for n in (700, 1, 1300, 0, 450)]
    poses = [(0.0, 0.0, 0.0), (1.7, -0.2, 0.05), (3.1, 0.4, 0.13), (4.9, 0.1, 0.2), (6.2, -0.7, 0.31)]
    out_tp = {}
    for name, clouds in (("f64", [c.astype(np.float64) for c in clouds32]), ("f32", clouds32)):
        kfs = [types.SimpleNamespace(points=c, pose=Pose2(*q)) for c, q in zip(clouds, poses)]
        slam = types.SimpleNamespace(keyframes=kfs, current_key=len(kfs), point_resolution=0.5)
        for ref in (4, 2):
            frames = [1, 2, 3] if ref == 4 else [0, 1]
            Ts = [kfs[ref].pose.between(kfs[k].pose).matrix().astype(np.float32) for k in frames]
            moved = [Keyframe.transform_points(kfs[k].points, kfs[ref].pose.between(kfs[k].pose)) for k in frames]
            tgt_cloud = ns_gp["get_points"](slam, frames, ref)
            out_tp["%s_ref%d_T" % (name, ref)] = np.array(Ts)
            out_tp["%s_ref%d_frames" % (name, ref)] = np.array(frames)
            for k, mv in zip(frames, moved):
                out_tp["%s_ref%d_moved%d" % (name, ref, k)] = np.asarray(mv, np.float32)   # the pybind boundary
                out_tp["%s_ref%d_moved%d_dtype" % (name, ref, k)] = np.array(str(np.asarray(mv).dtype))
            out_tp["%s_ref%d_target" % (name, ref)] = np.asarray(tgt_cloud, np.float32)
    for i, c in enumerate(clouds32):
        out_tp["cloud%d" % i] = c
    out_tp["stand_ins"] = thirdparty.record()
    np.savez_compressed(os.path.join(HERE, "transform_points.npz"), **out_tp)
    # ---- CFAR masks / threshold maps from the reference's own cfar.cpp (oracle/_ref, compiled unmodified) ----
    import oracle
    if not oracle.have_ref_cfar():
        sys.exit("oracle/_ref/libcfar_ref.so missing: run `make -C oracle ref` first")
    rng = np.random.default_rng(2026)
    frames = {"sonar": synth.sonar_frame(seed=5, rows=192, cols=64, n_blobs=8),
              "noise": rng.integers(0, 256, (130, 36), dtype=np.uint8),
              "short": rng.integers(0, 256, (50, 12), dtype=np.uint8),        # shorter than the shipped window
              "float": (rng.random((140, 33)) * 300).astype(np.float32)}       # non-uint8 caller
    cases = [(20, 5, {"CA": 2.3701490070915554, "SOCA": 2.749063720096473, "GOCA": 2.121926842646487,
                      "OS": 9.137608674642355}, 10), (8, 2, {"CA": 1.6, "SOCA": 1.9, "GOCA": 1.4, "OS": 3.0}, 5),
             (16, 4, {"CA": 2.0, "SOCA": 2.2, "GOCA": 1.8, "OS": 5.0}, 12)]
    out = {"frame_" + k: v for k, v in frames.items()}
    index = []
    for ci, (th, gh, taus, k) in enumerate(cases):
        for alg in ("CA", "SOCA", "GOCA", "OS"):
            for name, img in frames.items():
                m, t = oracle.ref_cfar(img, alg, th, gh, taus[alg], k, want_threshold=True)
                key = "c%d_%s_%s" % (ci, alg, name)
                out[key + "_mask"] = np.packbits(m, axis=None)
                out[key + "_thr"] = t
                index.append([key, name, alg, th, gh, taus[alg], k])
    out["index"] = np.array(json.dumps(index))
    np.savez_compressed(os.path.join(HERE, "cfar_ref.npz"), **out)
    print("wrote cfar_tau.json, maps_small.npz, maps_digest.json, matching_cost.npz, transform_points.npz, cfar_ref.npz")
    make_nssm_pieces()
    make_ssm_session()
    make_nssm_session()
    make_feature_callback()


if __name__ == "__main__":
    main()
