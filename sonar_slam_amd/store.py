"""Device-resident keyframe clouds (``sfe_cloud_store``, include/sonarfe.h): SURVEY 8 row f4.

The reference moves every feature cloud through a ROS topic and rebuilds the scan matcher's inputs in
numpy (feature_extraction.py:175-193 -> slam_ros.py:169-170 -> slam.py:229-292, slam_objects.py:178-198).
Here the clouds stay in HBM from the feature extractor to the scan matcher: a cloud is named by an
integer handle; ``get_points`` (transform + concatenate + pcl.downsample), ``icp`` and ``overlap`` take
handles; the host sees sizes (a few bytes per cloud) and results, and the points only when it asks
(``read``: the PointCloud2 for rviz / the mapping node).
"""
import ctypes as _C

import numpy as np

from . import _lib as _L

NEGATE_Y = 1        # SFE_STORE_NEGATE_Y: store (x, -y), what slam_ros.py:170 makes of the feature message
F32_POINTS = 2      # SFE_STORE_F32_POINTS: transform like numpy does for float32 keyframe clouds (sgemm)


def pose_T6(pose_or_matrix):
    """The six float32 numbers Keyframe.transform_points reads from ``pose.matrix().astype(np.float32)``
    (slam_objects.py:192-195): T00 T01 T02 T10 T11 T12."""
    M = pose_or_matrix.matrix() if hasattr(pose_or_matrix, "matrix") else pose_or_matrix
    return np.asarray(M, np.float64).astype(np.float32)[:2, :3].reshape(6)


class CloudStore(object):
    def __init__(self, ctx=None, capacity_points=1 << 22, max_clouds=1 << 16):
        self.ctx = ctx or _L.default_context()
        h = _C.c_void_p()
        with self.ctx.lock:
            self.ctx._check(self.ctx.lib.sfe_cloud_store_create(self.ctx.handle, int(capacity_points), int(max_clouds),
                                                                _C.byref(h)))
        self.handle = h
        self.capacity_points, self.max_clouds = int(capacity_points), int(max_clouds)

    # -- filling --
    def put(self, points, stamp=0):
        """one host cloud (N x 2, rounded to float32 like the pybind boundary) -> handle"""
        pts = np.ascontiguousarray(points, np.float32).reshape(-1, 2)
        h = _C.c_int32(-1)
        with self.ctx.lock:
            self.ctx._check(self.ctx.lib.sfe_cloud_store_put(self.ctx.handle, self.handle, int(stamp),
                                                             _L.ptr(pts, _C.c_float), len(pts), _C.byref(h)))
        return h.value

    def put_batch_dev(self, d_clouds, d_counts, n_frames, cap, stamps=None, flags=0):
        """n_frames clouds straight from the resident cloud filter's outputs (device buffers) -> handles"""
        handles = np.zeros(n_frames, np.int32)
        st = None if stamps is None else np.ascontiguousarray(stamps, np.int64)
        with self.ctx.lock:
            self.ctx._check(self.ctx.lib.sfe_cloud_store_put_batch_dev(
                self.ctx.handle, self.handle, None if st is None else _L.ptr(st, _C.c_int64), d_clouds.ptr, d_counts.ptr,
                int(n_frames), int(cap), int(flags), _L.ptr(handles, _C.c_int32)))
        return handles

    # -- bookkeeping --
    def __len__(self):
        return int(self.ctx.lib.sfe_cloud_store_count(self.handle))

    def meta(self, first=0, n=None):
        """-> (stamps int64, offsets int64, counts int32) of clouds first .. first + n - 1"""
        n = len(self) - first if n is None else int(n)
        st, off, cnt = np.zeros(n, np.int64), np.zeros(n, np.int64), np.zeros(n, np.int32)
        with self.ctx.lock:
            self.ctx._check(self.ctx.lib.sfe_cloud_store_meta(self.ctx.handle, self.handle, int(first), n,
                                                              _L.ptr(st, _C.c_int64), _L.ptr(off, _C.c_int64),
                                                              _L.ptr(cnt, _C.c_int32)))
        return st, off, cnt

    def counts(self, handles):
        handles = np.asarray(handles, np.int64).reshape(-1)
        if len(handles) == 0:
            return np.zeros(0, np.int32)
        lo, hi = int(handles.min()), int(handles.max())
        return self.meta(lo, hi - lo + 1)[2][handles - lo]

    def read(self, handle):
        """the points of one cloud (N x 2 float32), copied to the host"""
        n = int(self.counts([handle])[0])
        out = np.zeros((max(n, 0), 2), np.float32)
        m = _C.c_int(0)
        with self.ctx.lock:
            self.ctx._check(self.ctx.lib.sfe_cloud_store_read(self.ctx.handle, self.handle, int(handle),
                                                              _L.ptr(out, _C.c_float), len(out), _C.byref(m)))
        if m.value < 0:
            raise _L.SonarFEError("cloud %d was not stored (count %d: -1 octree too deep, -3 pool full)" % (handle, m.value))
        return out

    def truncate(self, n_slots):
        with self.ctx.lock:
            self.ctx._check(self.ctx.lib.sfe_cloud_store_truncate(self.ctx.handle, self.handle, int(n_slots)))

    # -- the SLAM node's three uses of a keyframe cloud --
    def get_points(self, handles, T6, resolution, flags=0, stamps=None):
        """SLAM.get_points(frames, ref_frame) for many targets at once: handles [n_jobs x m] (-1 = unused),
        T6 [n_jobs x m x 6] (``pose_T6`` of ref_pose.between(pose)) -> new handles [n_jobs]"""
        handles = np.ascontiguousarray(handles, np.int32)
        if handles.ndim == 1:
            handles = handles[None, :]
        n_jobs, m = handles.shape
        T6 = np.ascontiguousarray(T6, np.float32).reshape(n_jobs, m, 6)
        out = np.zeros(n_jobs, np.int32)
        st = None if stamps is None else np.ascontiguousarray(stamps, np.int64)
        with self.ctx.lock:
            self.ctx._check(self.ctx.lib.sfe_cloud_store_get_points(
                self.ctx.handle, self.handle, _L.ptr(handles, _C.c_int32), _L.ptr(T6, _C.c_float), n_jobs, m,
                float(resolution), int(flags), None if st is None else _L.ptr(st, _C.c_int64), _L.ptr(out, _C.c_int32)))
        return out

    def icp(self, params, pairs, guesses):
        """SLAM.compute_icp over handles: pairs [n x 2] = (source, target), guesses [n x 3 x 3]
        -> (T [n x 3 x 3] float32, status [n], iterations [n])"""
        pairs = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
        n = len(pairs)
        g = np.ascontiguousarray(np.asarray(guesses, np.float32).reshape(n, 9))
        T, st, it = np.zeros((n, 3, 3), np.float32), np.zeros(n, np.int32), np.zeros(n, np.int32)
        with self.ctx.lock:
            self.ctx._check(self.ctx.lib.sfe_icp_store_compute(
                self.ctx.handle, _C.byref(params), self.handle, _L.ptr(pairs, _C.c_int32), _L.ptr(g, _C.c_float), n,
                _L.ptr(T, _C.c_float), _L.ptr(st, _C.c_int32), _L.ptr(it, _C.c_int32)))
        return T, st, it

    def overlap(self, pairs, T6, max_dist, flags=0):
        """SLAM.get_overlap for many (source, target) pairs -> matched source points per pair"""
        pairs = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
        n = len(pairs)
        T6 = np.ascontiguousarray(T6, np.float32).reshape(n, 6)
        out = np.zeros(n, np.int32)
        with self.ctx.lock:
            self.ctx._check(self.ctx.lib.sfe_cloud_store_overlap(
                self.ctx.handle, self.handle, _L.ptr(pairs, _C.c_int32), _L.ptr(T6, _C.c_float), n, float(max_dist),
                int(flags), _L.ptr(out, _C.c_int32)))
        return out

    # -- the loop-closure search (slam.py:839-1001) --
    def bbox(self, handles):
        """-> [n x 4] float32 (min x, min y, max x, max y) of the named clouds"""
        h = np.ascontiguousarray(handles, np.int32).reshape(-1)
        out = np.zeros((len(h), 4), np.float32)
        with self.ctx.lock:
            self.ctx._check(self.ctx.lib.sfe_cloud_store_bbox(self.ctx.handle, self.handle, _L.ptr(h, _C.c_int32), len(h),
                                                              _L.ptr(out, _C.c_float)))
        return out

    def get_points_keys(self, handles, T6, keys, resolution, flags=0, stamp=0):
        """SLAM.get_points(frames, None, return_keys=True) (slam.py:873): cloud k under T6[k] (its pose's matrix) tagged
        keys[k]; the descriptor overload of pcl.downsample.  Any size.  -> the new handle (its keys: ``read_keys``)"""
        handles = np.ascontiguousarray(handles, np.int32).reshape(-1)
        m = len(handles)
        T6 = np.ascontiguousarray(T6, np.float32).reshape(m, 6)
        keys = np.ascontiguousarray(keys, np.int32).reshape(m)
        out = _C.c_int32(-1)
        with self.ctx.lock:
            self.ctx._check(self.ctx.lib.sfe_cloud_store_get_points_keys(
                self.ctx.handle, self.handle, _L.ptr(handles, _C.c_int32), _L.ptr(T6, _C.c_float), _L.ptr(keys, _C.c_int32), m,
                float(resolution), int(flags), int(stamp), _C.byref(out)))
        return out.value

    def read_keys(self, handle):
        n = int(self.counts([handle])[0])
        out = np.zeros(max(n, 0), np.int32)
        m = _C.c_int(0)
        with self.ctx.lock:
            self.ctx._check(self.ctx.lib.sfe_cloud_store_read_keys(self.ctx.handle, self.handle, int(handle),
                                                                   _L.ptr(out, _C.c_int32), len(out), _C.byref(m)))
        return out

    def fov_select(self, handle, Tinv6, range_bounds, bearing_bounds, n_keys):
        """the field-of-view gate of slam.py:875-899 on a keyed cloud -> (per-key counts of the selected points [n_keys],
        number selected, number the device could not decide: then ``set_selection``)"""
        Tinv6 = np.ascontiguousarray(Tinv6, np.float32).reshape(-1, 6)
        rb = np.ascontiguousarray(range_bounds, np.float64).reshape(-1)
        bb = np.ascontiguousarray(bearing_bounds, np.float64).reshape(-1)
        assert len(rb) == len(bb) == len(Tinv6)
        hist = np.zeros(int(n_keys), np.int32)
        n_sel, n_amb = _C.c_int32(0), _C.c_int32(0)
        with self.ctx.lock:
            self.ctx._check(self.ctx.lib.sfe_cloud_store_fov_select(
                self.ctx.handle, self.handle, int(handle), _L.ptr(Tinv6, _C.c_float), _L.ptr(rb, _C.c_double),
                _L.ptr(bb, _C.c_double), len(Tinv6), int(n_keys), _L.ptr(hist, _C.c_int32), _C.byref(n_sel), _C.byref(n_amb)))
        return hist, n_sel.value, n_amb.value

    def set_selection(self, handle, sel):
        sel = np.ascontiguousarray(sel, np.uint8).reshape(-1)
        with self.ctx.lock:
            self.ctx._check(self.ctx.lib.sfe_cloud_store_set_selection(self.ctx.handle, self.handle, int(handle),
                                                                       _L.ptr(sel, _C.c_uint8), len(sel)))

    def compact_selected(self, handle, stamp=0):
        """target_points[sel], target_keys[sel] (slam.py:898-899) -> new keyed handle"""
        out = _C.c_int32(-1)
        with self.ctx.lock:
            self.ctx._check(self.ctx.lib.sfe_cloud_store_compact_selected(self.ctx.handle, self.handle, int(handle), int(stamp),
                                                                          _C.byref(out)))
        return out.value

    def match_keys(self, source, T6, target, max_dist, n_keys, flags=0):
        """slam.py:977-985 -> (per-key counts of the matched targets [n_keys], overlap)"""
        T6 = np.ascontiguousarray(T6, np.float32).reshape(6)
        hist = np.zeros(int(n_keys), np.int32)
        ov = _C.c_int32(0)
        with self.ctx.lock:
            self.ctx._check(self.ctx.lib.sfe_cloud_store_match_keys(
                self.ctx.handle, self.handle, int(source), _L.ptr(T6, _C.c_float), int(target), float(max_dist), int(flags),
                int(n_keys), _L.ptr(hist, _C.c_int32), _C.byref(ov)))
        return hist, ov.value

    def close(self):
        if self.handle is not None and self.ctx.handle is not None:
            self.ctx.lib.sfe_cloud_store_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
