"""Global-initialisation matching cost (the function ``scipy.optimize.shgo`` minimises).

Mirror of ``SLAM.get_matching_cost_subroutine1`` (bruce_slam/src/bruce_slam/slam.py:461-570; call
sites slam.py:692-701 and :952-961): same arguments, same return value ``(subroutine,
pose_samples)``, ``subroutine(x)`` returns the same cost and logs the same sample row.  The grid
bookkeeping stays numpy exactly as the reference writes it; the two per-cell / per-point stages
run on the GPU:

    cv2.getStructuringElement + cv2.dilate      slam.py:522-527  -> sfe_costgrid_create
    transform_points + round + lookup + sum     slam.py:549-562  -> sfe_matching_cost_batch

``subroutine.batch(X)`` scores many candidate poses in ONE launch (shgo's Sobol sampling stage
evaluates ``n`` points before any local minimisation: slam.py:692-701 passes n =
initialization_params[0]); ``subroutine(x)`` is ``batch([x])[0]``.

Poses: any object with gtsam.Pose2's ``compose / between / matrix / x / y / theta`` is used through
those methods (so with the real gtsam installed the host-side pose algebra IS gtsam's); plain
``(x, y, theta)`` triples go through ``pose2.Pose2``, a restatement of gtsam's Pose2/Rot2 algebra
(gtsam is absent from this image: parity unpinned for that fallback only).
"""
import ctypes as _C

import numpy as np

from . import _lib as _L
from .pose2 import Pose2


def _as_pose(p):
    if hasattr(p, "compose") and hasattr(p, "matrix"):
        return p
    x, y, th = p
    return Pose2(x, y, th)


def _like(pose, x):
    """n2g(x, "Pose2") in the pose class the caller uses."""
    return type(pose)(float(x[0]), float(x[1]), float(x[2]))


def get_matching_cost_subroutine1(source_points, source_pose, target_points, target_pose, source_pose_cov=None,
                                  point_noise=0.5, ctx=None):
    """-> (subroutine, pose_samples), as slam.py:461-570.  ``point_noise`` is ``self.point_noise``
    (slam.yaml; slam.py:73).  ``source_pose_cov`` is accepted and inverted like the reference
    does (slam.py:529; the result is unused there too)."""
    ctx = ctx or _L.default_context()
    source_pose, target_pose = _as_pose(source_pose), _as_pose(target_pose)
    source_points = np.asarray(source_points)
    target_points = np.asarray(target_points)
    pose_samples = []

    # slam.py:507-519, numpy verbatim (host; once per keyframe)
    xmin, ymin = np.min(target_points, axis=0) - 2 * point_noise
    xmax, ymax = np.max(target_points, axis=0) + 2 * point_noise
    resolution = point_noise / 10.0
    xs = np.arange(xmin, xmax, resolution)
    ys = np.arange(ymin, ymax, resolution)
    rows, cols = len(ys), len(xs)
    r = np.int32(np.round((target_points[:, 1] - ymin) / resolution))
    c = np.int32(np.round((target_points[:, 0] - xmin) / resolution))
    r = np.ascontiguousarray(np.clip(r, 0, rows - 1), np.int32)
    c = np.ascontiguousarray(np.clip(c, 0, cols - 1), np.int32)
    dilate_hs = int(np.ceil(point_noise / resolution))          # slam.py:522

    if source_pose_cov is not None:
        np.linalg.inv(source_pose_cov)                          # slam.py:529 (raises like the reference)

    handle = _C.c_void_p()
    with ctx.lock:
        ctx._check(ctx.lib.sfe_costgrid_create(ctx.handle, _L.ptr(r, _C.c_int32), _L.ptr(c, _C.c_int32), len(r),
                                               rows, cols, dilate_hs, _C.byref(handle)))
    grid = _Grid(ctx, handle, rows, cols)
    # the division / subtraction happen in the points' dtype in the reference (float32 clouds)
    f32 = np.float32
    src32 = np.ascontiguousarray(source_points, f32).reshape(-1, 2)
    x0, y0, res32 = f32(xmin), f32(ymin), f32(resolution)

    def batch(X):
        X = np.asarray(X, np.float64).reshape(-1, 3)
        T6 = np.zeros((len(X), 6), f32)
        sample_poses = []
        for i, x in enumerate(X):
            delta = _like(source_pose, x)                       # n2g(x, "Pose2")
            sample_source_pose = source_pose.compose(delta)
            sample_transform = target_pose.between(sample_source_pose)
            T = np.asarray(sample_transform.matrix()).astype(f32)   # Keyframe.transform_points (slam_objects.py:193)
            T6[i] = (T[0, 0], T[0, 1], T[0, 2], T[1, 0], T[1, 1], T[1, 2])
            sample_poses.append(sample_source_pose)
        cost = np.zeros(len(X), np.int32)
        with ctx.lock:
            ctx._check(ctx.lib.sfe_matching_cost_batch(ctx.handle, grid.handle, _L.ptr(src32, _C.c_float), len(src32),
                                                       _L.ptr(T6, _C.c_float), len(T6), x0, y0, res32,
                                                       _L.ptr(cost, _C.c_int32)))
        for sp, cst in zip(sample_poses, cost):
            pose_samples.append(np.r_[[sp.x(), sp.y(), sp.theta()], cst])   # np.r_[g2n(pose), cost]
        return cost

    def subroutine(x):
        return batch([x])[0]

    subroutine.batch = batch
    subroutine.grid = grid            # keeps the device grid alive as long as the closure
    subroutine.geometry = dict(xmin=x0, ymin=y0, resolution=res32, rows=rows, cols=cols, dilate_hs=dilate_hs,
                               target_r=r, target_c=c)
    return subroutine, pose_samples


class _Grid(object):
    def __init__(self, ctx, handle, rows, cols):
        self.ctx, self.handle, self.rows, self.cols = ctx, handle, rows, cols

    def download(self):
        """target_grids after the dilation (rows x cols uint8, 0 / 255)."""
        out = np.zeros((self.rows, self.cols), np.uint8)
        with self.ctx.lock:
            self.ctx._check(self.ctx.lib.sfe_costgrid_download(self.ctx.handle, self.handle, _L.ptr(out, _C.c_uint8)))
        return out

    def close(self):
        if self.handle is not None and self.ctx.handle is not None:
            self.ctx.lib.sfe_costgrid_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
