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

``get_matching_cost_subroutine1_store`` is the same over handles of a ``store.CloudStore``: the target
grid is built on the device from the target handle (only the cloud's bounding box comes down, for the numpy
bookkeeping of slam.py:506-511), the source cloud is read where it lies.  ``batch_store`` builds the grids of MANY
(source, target) pairs at once and scores all their candidate poses in one launch (chained.SessionBatch).

dtypes.  numpy evaluates slam.py:549-562 in the dtype of ``source_points``: float32 clouds (what get_points returns: the
NSSM source) in float32 -- the transform through sgemm -- and the SLAM node's keyframe clouds (float64 arrays of float32
values, slam_ros.py:169-170: the SSM source) in double.  Both are implemented (``SFE_COST_F64_POINTS``) and chosen from
the array's dtype; over handles the caller says which one the cloud stands for (``f64_points``).

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


def _sample_poses(lib, source_pose, target_pose, X):
    """-> (T6 [n x 6] float32 of target_pose.between(source_pose.compose(n2g(x))).matrix() (slam.py:548-550), sample source poses
    [n x 3] = g2n(source_pose.compose(n2g(x)))) for the deltas X [n x 3].  This package's Pose2: the library's host routine and
    vector arithmetic with Pose2's own expressions (cos / sin / atan2 through ``math``); any other pose class (gtsam.Pose2): its
    own methods, pose by pose."""
    import math
    X = np.asarray(X, np.float64).reshape(-1, 3)
    n = len(X)
    if type(source_pose) is Pose2 and type(target_pose) is Pose2 and n:
        from .chained import Pose2Batch, sample_transforms
        tb = Pose2Batch([target_pose._x], [target_pose._y], cs=([target_pose._c], [target_pose._s]))
        sb = Pose2Batch([source_pose._x], [source_pose._y], cs=([source_pose._c], [source_pose._s]))
        T6 = sample_transforms(lib, tb, sb, X)[0]
        c = np.array([math.cos(t) for t in X[:, 2]])
        sn = np.array([math.sin(t) for t in X[:, 2]])
        one = np.zeros(n, np.int64)
        sp = sb.take(one).compose(Pose2Batch(X[:, 0], X[:, 1], cs=(c, sn)))
        return T6, sp.xytheta()
    T6 = np.zeros((n, 6), np.float32)
    poses = np.zeros((n, 3))
    for i, x in enumerate(X):
        sample_source_pose = source_pose.compose(_like(source_pose, x))
        T = np.asarray(target_pose.between(sample_source_pose).matrix()).astype(np.float32)   # Keyframe.transform_points (slam_objects.py:193)
        T6[i] = (T[0, 0], T[0, 1], T[0, 2], T[1, 0], T[1, 1], T[1, 2])
        poses[i] = (sample_source_pose.x(), sample_source_pose.y(), sample_source_pose.theta())
    return T6, poses


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
    # the division / subtraction happen in the points' dtype in the reference: float32 for float32 clouds, double for
    # the SLAM node's float64 keyframe clouds (whose values are float32 numbers: slam_ros.py:169-170)
    f32 = np.float32
    flags = F64_POINTS if source_points.dtype == np.float64 else 0
    src32 = np.ascontiguousarray(source_points, f32).reshape(-1, 2)
    if flags and not np.array_equal(src32.astype(np.float64).reshape(-1), np.asarray(source_points, np.float64).reshape(-1)):
        raise ValueError("float64 source clouds must hold float32 values (slam_ros.py:169-170); cast the cloud to float32 first")
    x0, y0, res32 = f32(xmin), f32(ymin), f32(resolution)

    def batch(X, record=True):
        """costs of the deltas X [n x 3] in one launch; record=False: the evaluations are not entered into pose_samples (the caller
        enters those the reference would have made: ``subroutine.record``)"""
        X = np.asarray(X, np.float64).reshape(-1, 3)
        T6, poses = _sample_poses(ctx.lib, source_pose, target_pose, X)
        cost = np.zeros(len(X), np.int32)
        if len(X):
            with ctx.lock:
                ctx._check(ctx.lib.sfe_matching_cost_batch(ctx.handle, grid.handle, _L.ptr(src32, _C.c_float), len(src32),
                                                           _L.ptr(T6, _C.c_float), len(T6), x0, y0, float(resolution), flags,
                                                           _L.ptr(cost, _C.c_int32)))
        if record:
            pose_samples.extend(np.c_[poses, np.asarray(cost, np.float64)])     # np.r_[g2n(pose), cost] per evaluation
        return cost

    def record(X, costs):
        _, poses = _sample_poses(ctx.lib, source_pose, target_pose, X)
        pose_samples.extend(np.c_[poses, np.asarray(costs, np.float64)])

    def subroutine(x):
        return batch([x])[0]

    subroutine.batch = batch
    subroutine.record = record
    subroutine.grid = grid            # keeps the device grid alive as long as the closure
    subroutine.geometry = dict(xmin=x0, ymin=y0, resolution=res32, rows=rows, cols=cols, dilate_hs=dilate_hs,
                               target_r=r, target_c=c)
    return subroutine, pose_samples


F64_POINTS = 1      # SFE_COST_F64_POINTS


def grid_geometry(bbox, point_noise):
    """slam.py:506-511 + :521 from a cloud's bounding box (min x, min y, max x, max y as the float32 numbers np.min /
    np.max of the float32 cloud give): numpy verbatim -> (xmin, ymin, resolution, rows, cols, dilate_hs)"""
    mn = np.array([bbox[0], bbox[1]], np.float32)
    mx = np.array([bbox[2], bbox[3]], np.float32)
    xmin, ymin = mn - 2 * point_noise
    xmax, ymax = mx + 2 * point_noise
    resolution = point_noise / 10.0
    xs = np.arange(xmin, xmax, resolution)
    ys = np.arange(ymin, ymax, resolution)
    return xmin, ymin, resolution, len(ys), len(xs), int(np.ceil(point_noise / resolution))


_MANY_OK = None       # grid_geometry_many agrees with grid_geometry under the installed numpy (checked on first use)


def grid_geometry_many(bbox, point_noise):
    """grid_geometry for n bounding boxes [n x 4] at once -> (xmin [n], ymin [n], resolution, rows [n], cols [n], dilate_hs).
    np.arange's length is ceil((stop - start) / step) evaluated on the scalars' own types (float32 - float32, divided by the
    Python float): under NEP 50 (numpy >= 2) that is the float32 array arithmetic below; older numpys promote the scalar
    division to double.  So the array form is CHECKED against the scalar function -- every box of the first call, eight boxes of
    every later one -- and the scalar loop takes over for good if they ever differ."""
    global _MANY_OK
    bbox = np.asarray(bbox, np.float32).reshape(-1, 4)
    n = len(bbox)

    def loop():
        geo = [grid_geometry(b, point_noise) for b in bbox]
        return (np.array([g[0] for g in geo], np.float32), np.array([g[1] for g in geo], np.float32), float(geo[0][2]),
                np.array([g[3] for g in geo], np.int32), np.array([g[4] for g in geo], np.int32), int(geo[0][5]))
    if _MANY_OK is None and np.result_type(np.float32(1), 1.0) != np.float32:
        # legacy promotion (numpy < 2, or NPY_PROMOTION_STATE=legacy): `float32 scalar / Python float` is a double division there,
        # and the two forms differ only for near-integer quotients -- too rare for a sampled check to catch.
        # Decided once from the promotion rule itself: the scalar loop, which IS numpy's own np.arange on that numpy.
        _MANY_OK = False
    if _MANY_OK is False or n == 0:
        return loop() if n else (np.zeros(0, np.float32), np.zeros(0, np.float32), point_noise / 10.0, np.zeros(0, np.int32),
                                 np.zeros(0, np.int32), int(np.ceil(point_noise / (point_noise / 10.0))))
    lo = bbox[:, :2] - 2 * point_noise
    hi = bbox[:, 2:] + 2 * point_noise
    resolution = point_noise / 10.0
    length = np.maximum(np.ceil(((hi - lo) / resolution).astype(np.float64)), 0).astype(np.int32)
    out = (np.ascontiguousarray(lo[:, 0]), np.ascontiguousarray(lo[:, 1]), resolution, np.ascontiguousarray(length[:, 1]),
           np.ascontiguousarray(length[:, 0]), int(np.ceil(point_noise / resolution)))
    probe = np.arange(n) if _MANY_OK is None else np.unique(np.linspace(0, n - 1, 8).astype(int))
    for i in probe:
        g = grid_geometry(bbox[i], point_noise)
        if not (g[0] == out[0][i] and g[1] == out[1][i] and g[2] == out[2] and g[3] == out[3][i] and g[4] == out[4][i]
                and g[5] == out[5] and np.asarray(g[0]).dtype == np.float32):
            _MANY_OK = False
            return loop()
    _MANY_OK = True
    return out


def batch_store(store, source_handles, target_handles, T6, point_noise=0.5, f64_points=True):
    """Costs of n_poses candidate transforms for each of n (source, target) pairs of ``store`` in ONE launch: T6
    [n x n_poses x 6] float32 (``store.pose_T6`` of sample_transform).  -> (costs [n x n_poses] int32, grids): ``grids``
    can score further poses of the same pairs (``grids.cost``) and must be closed."""
    grids = _StoreGrids(store, target_handles, point_noise)
    return grids.cost(source_handles, T6, f64_points), grids


class _StoreGrids(object):
    """the dilated target grids of n store clouds (slam.py:505-527), device-resident"""

    def __init__(self, store, target_handles, point_noise):
        self.store, self.ctx = store, store.ctx
        th = np.ascontiguousarray(target_handles, np.int32).reshape(-1)
        self.n = len(th)
        bbox = store.bbox(th)
        if not np.all(np.isfinite(bbox)):
            raise ValueError("matching cost: target cloud %d is empty or holds non-finite points (np.min of an empty cloud raises "
                             "in the reference too: slam.py:506); the caller tests min_points first (slam.py:659)"
                             % int(th[int(np.argmax(~np.isfinite(bbox).all(axis=1)))]))
        self.xmin, self.ymin, self.resolution, self.rows, self.cols, self.dilate_hs = grid_geometry_many(bbox, point_noise)
        self.resolution = float(self.resolution)
        h = _C.c_void_p()
        with self.ctx.lock:
            self.ctx._check(self.ctx.lib.sfe_costgrid_create_store(
                self.ctx.handle, store.handle, _L.ptr(th, _C.c_int32), self.n, _L.ptr(self.xmin, _C.c_float),
                _L.ptr(self.ymin, _C.c_float), np.float32(self.resolution), _L.ptr(self.rows, _C.c_int32),
                _L.ptr(self.cols, _C.c_int32), self.dilate_hs, _C.byref(h)))
        self.handle = h

    def cost(self, source_handles, T6, f64_points=True, grid_index=None):
        """costs [n_jobs x n_poses]: job i = source cloud source_handles[i] against grid grid_index[i] (default: grid i)"""
        sh = np.ascontiguousarray(source_handles, np.int32).reshape(-1)
        gi = None if grid_index is None else np.ascontiguousarray(grid_index, np.int32).reshape(-1)
        assert len(sh) == (self.n if gi is None else len(gi))
        T6 = np.ascontiguousarray(T6, np.float32).reshape(len(sh), -1, 6)
        out = np.zeros((len(sh), T6.shape[1]), np.int32)
        with self.ctx.lock:
            self.ctx._check(self.ctx.lib.sfe_matching_cost_store(
                self.ctx.handle, self.handle, self.store.handle, _L.ptr(sh, _C.c_int32),
                None if gi is None else _L.ptr(gi, _C.c_int32), len(sh), _L.ptr(T6, _C.c_float), T6.shape[1],
                self.resolution, F64_POINTS if f64_points else 0, _L.ptr(out, _C.c_int32)))
        return out

    def cost_samples(self, source_handles, target_xycs, source_xycs, delta_xycs, f64_points=True, grid_index=None):
        """costs [n_jobs x n_deltas] under target_i.between(source_i.compose(delta_j)), the transforms computed on the device from
        the poses ({x, y, cos, sin} rows, float64): sfe_matching_cost_store_samples"""
        sh = np.ascontiguousarray(source_handles, np.int32).reshape(-1)
        gi = None if grid_index is None else np.ascontiguousarray(grid_index, np.int32).reshape(-1)
        assert len(sh) == (self.n if gi is None else len(gi))
        t4 = np.ascontiguousarray(target_xycs, np.float64).reshape(len(sh), 4)
        s4 = np.ascontiguousarray(source_xycs, np.float64).reshape(len(sh), 4)
        d4 = np.ascontiguousarray(delta_xycs, np.float64).reshape(-1, 4)
        out = np.zeros((len(sh), len(d4)), np.int32)
        with self.ctx.lock:
            self.ctx._check(self.ctx.lib.sfe_matching_cost_store_samples(
                self.ctx.handle, self.handle, self.store.handle, _L.ptr(sh, _C.c_int32),
                None if gi is None else _L.ptr(gi, _C.c_int32), len(sh), _L.ptr(t4, _C.c_double), _L.ptr(s4, _C.c_double),
                _L.ptr(d4, _C.c_double), len(d4), self.resolution, F64_POINTS if f64_points else 0, _L.ptr(out, _C.c_int32)))
        return out

    def download(self, index=0):
        out = np.zeros((int(self.rows[index]), int(self.cols[index])), np.uint8)
        with self.ctx.lock:
            self.ctx._check(self.ctx.lib.sfe_costgrid_download(self.ctx.handle, self.handle, int(index), _L.ptr(out, _C.c_uint8)))
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


def get_matching_cost_subroutine1_store(store, source_handle, source_pose, target_handle, target_pose, source_pose_cov=None,
                                        point_noise=0.5, f64_points=True):
    """``get_matching_cost_subroutine1`` (slam.py:461-570) for two clouds of a ``store.CloudStore``: -> (subroutine,
    pose_samples) with ``subroutine.batch``.  ``f64_points``: the SOURCE handle stands for a keyframe cloud of the SLAM node
    (float64 array of float32 values: the sequential scan match, slam.py:683-689); False for a cloud get_points returned
    (float32: the loop-closure search, slam.py:943-949)."""
    source_pose, target_pose = _as_pose(source_pose), _as_pose(target_pose)
    if source_pose_cov is not None:
        np.linalg.inv(source_pose_cov)                          # slam.py:529 (raises like the reference)
    grids = _StoreGrids(store, [target_handle], point_noise)
    pose_samples = []
    f32 = np.float32

    def batch(X, record=True):
        X = np.asarray(X, np.float64).reshape(-1, 3)
        T6, poses = _sample_poses(store.ctx.lib, source_pose, target_pose, X)
        cost = grids.cost([source_handle], T6[None], f64_points)[0] if len(X) else np.zeros(0, np.int32)
        if record:
            pose_samples.extend(np.c_[poses, np.asarray(cost, np.float64)])
        return cost

    def record(X, costs):
        _, poses = _sample_poses(store.ctx.lib, source_pose, target_pose, X)
        pose_samples.extend(np.c_[poses, np.asarray(costs, np.float64)])

    def subroutine(x):
        return batch([x])[0]

    subroutine.batch = batch
    subroutine.record = record
    subroutine.grid = grids
    subroutine.geometry = dict(xmin=grids.xmin[0], ymin=grids.ymin[0], resolution=f32(grids.resolution), rows=int(grids.rows[0]),
                               cols=int(grids.cols[0]), dilate_hs=grids.dilate_hs)
    return subroutine, pose_samples


class _Grid(object):
    def __init__(self, ctx, handle, rows, cols):
        self.ctx, self.handle, self.rows, self.cols = ctx, handle, rows, cols

    def download(self):
        """target_grids after the dilation (rows x cols uint8, 0 / 255)."""
        out = np.zeros((self.rows, self.cols), np.uint8)
        with self.ctx.lock:
            self.ctx._check(self.ctx.lib.sfe_costgrid_download(self.ctx.handle, self.handle, 0, _L.ptr(out, _C.c_uint8)))
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
