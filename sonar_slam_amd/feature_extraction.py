"""ROS-free core of the reference's feature-extraction node.

Mirror of ``bruce_slam.feature_extraction.FeatureExtraction``
(bruce_slam/src/bruce_slam/feature_extraction.py:27-252): same attribute names, same
``configure()`` / ``generate_map_xy(ping)`` / ``callback(ping)`` flow, same YAML keys
(bruce_slam/config/feature.yaml), but every per-pixel stage runs on the GPU:

    CFAR.detect + intensity gate      feature_extraction.py:223-224  -> sfe_cfar_u8 (fused)
    cv2.remap(peaks) + np.nonzero     feature_extraction.py:231-232  -> sfe_extract_points
    pixel -> metres                   feature_extraction.py:235-238  ->   "      (same launch)
    cv2.remap(img) for the vis image  feature_extraction.py:226      -> sfe_remap_u8
    pcl.downsample / remove_outlier   feature_extraction.py:241-249  -> sonar_slam_amd.pcl

The rospy plumbing (subscriber, PointCloud2 publisher, cv_bridge) is NOT reproduced here: a
rospy node wraps this class by forwarding ``sonar_msg`` to ``callback`` and publishing the
returned points (INTEGRATION.md shows the 10-line wrapper).  ``ping`` is any object with the
OculusPing fields the reference reads: ``ping_id``, ``bearings`` (1/100 deg), ``range_resolution``,
``num_ranges`` and the decoded uint8 image as ``image`` (rows = range bins, cols = beams).
"""
import ctypes as _C

import numpy as np
import yaml
from scipy.interpolate import interp1d

from . import _lib as _L
from . import pcl
from .CFAR import CFAR


COLORMAP_JET = 2        # cv2.COLORMAP_JET, the map of feature_extraction.py:227
PING_VIS_JET = 4        # SFE_PING_VIS_JET


def colormap_lut(colormap=COLORMAP_JET):
    """the 256 x 3 BGR table: cv2.applyColorMap(img, colormap) == lut[img] (sfe_colormap_lut; host only)"""
    lut = np.zeros((256, 3), np.uint8)
    if _L.load_library().sfe_colormap_lut(int(colormap), _L.ptr(lut, _C.c_uint8)) != 0:
        raise ValueError("only cv2.COLORMAP_JET (2) is built (feature_extraction.py:227)")
    return lut


class SonarPing(object):
    """Minimal stand-in for sonar_oculus/OculusPing(Uncompressed) (last row)."""

    def __init__(self, image, bearings, range_resolution, ping_id=0, stamp=None):
        self.image = np.ascontiguousarray(image, np.uint8)
        self.bearings = np.asarray(bearings)
        self.range_resolution = float(range_resolution)
        self.num_ranges = int(self.image.shape[0])
        self.ping_id = int(ping_id)
        self.stamp = stamp


def oculus_bearings(n_beams, aperture_deg=130.0):
    """Evenly spaced beam bearings in 1/100 degree, int16 like OculusPing.bearings."""
    half = aperture_deg * 50.0
    return np.round(np.linspace(-half, half, n_beams)).astype(np.int16)


class Geometry(object):
    """Device-side polar->Cartesian geometry (``sfe_geom``) built from the float maps."""

    def __init__(self, ctx, map_x, map_y, polar_shape, width, height):
        self.ctx = ctx
        self.map_x = np.ascontiguousarray(map_x, np.float32)
        self.map_y = np.ascontiguousarray(map_y, np.float32)
        self.cart_rows, self.cart_cols = self.map_x.shape
        self.polar_rows, self.polar_cols = polar_shape
        self.width, self.height = float(width), float(height)
        h = _C.c_void_p()
        with ctx.lock:
            ctx._check(ctx.lib.sfe_geom_create(
                ctx.handle, _L.ptr(self.map_x, _C.c_float), _L.ptr(self.map_y, _C.c_float),
                self.cart_rows, self.cart_cols, self.polar_rows, self.polar_cols, self.width,
                self.height, _C.byref(h)))
        self.handle = h

    def remap(self, img, colormap=None):
        """cv2.remap(img, map_x, map_y, cv2.INTER_LINEAR) for a uint8 polar image; with ``colormap=COLORMAP_JET`` the
        bgr8 image cv2.applyColorMap(remap, 2) of feature_extraction.py:226-228 in the same pass."""
        img = np.ascontiguousarray(img, np.uint8)
        if img.shape != (self.polar_rows, self.polar_cols):
            raise ValueError("remap: image shape %r does not match the geometry" % (img.shape,))
        if colormap is not None:
            dst = np.zeros((self.cart_rows, self.cart_cols, 3), np.uint8)
            with self.ctx.lock:
                self.ctx._check(self.ctx.lib.sfe_remap_u8_colormap(self.ctx.handle, self.handle, _L.ptr(img, _C.c_uint8),
                                                                   int(colormap), _L.ptr(dst, _C.c_uint8)))
            return dst
        dst = np.zeros((self.cart_rows, self.cart_cols), np.uint8)
        with self.ctx.lock:
            self.ctx._check(self.ctx.lib.sfe_remap_u8(self.ctx.handle, self.handle,
                                                      _L.ptr(img, _C.c_uint8), _L.ptr(dst, _C.c_uint8)))
        return dst

    def extract(self, mask, cap=None):
        """remap(mask) -> nonzero -> px->m.  Returns (locs int64 [N x 2] = (row, col) row-major,
        points float64 [N x 2] = (y_forward, x_lateral) metres)."""
        mask = np.ascontiguousarray(mask, np.uint8)
        if mask.shape != (self.polar_rows, self.polar_cols):
            raise ValueError("extract: mask shape %r does not match the geometry" % (mask.shape,))
        cap = int(cap) if cap is not None else 1 << 16
        while True:
            rc = np.zeros((cap, 2), np.int64)
            pts = np.zeros((cap, 2), np.float64)
            n = _C.c_int64(0)
            with self.ctx.lock:
                ret = self.ctx.lib.sfe_extract_points(self.ctx.handle, self.handle,
                                                      _L.ptr(mask, _C.c_uint8), cap,
                                                      _L.ptr(rc, _C.c_int64), _L.ptr(pts, _C.c_double),
                                                      _C.byref(n))
            if ret == _L.SFE_ERR_CAP:
                cap = int(n.value)
                continue
            self.ctx._check(ret)
            return rc[:n.value].copy(), pts[:n.value].copy()

    def feature_extract(self, img, alg, cfar_params, threshold, resolution, radius, min_points, want_vis=False,
                        cap=16384, colormap=None):
        """One ping through CFAR + gate -> remap + nonzero + px->m -> pcl.downsample -> pcl.remove_outlier in ONE
        library call (sfe_feature_extract_ping: one pinned upload, one download, one synchronisation).
        -> (cloud float32 [N x 2], vis image or None), or None when the cloud's octree is too deep for the
        resident filter (the caller then takes the per-stage entry points).  Bit-identical to the per-stage chain."""
        from .cfar import _gate_u8
        img = np.ascontiguousarray(img, np.uint8)
        if img.shape != (self.polar_rows, self.polar_cols):
            raise ValueError("feature_extract: image shape %r does not match the geometry" % (img.shape,))
        code = _L.ALG[alg]
        if code == 3:
            train_hs, guard_hs, k, tau = cfar_params
        else:
            (train_hs, guard_hs, tau), k = cfar_params, 0
        jet = want_vis and colormap is not None
        if jet and int(colormap) != COLORMAP_JET:
            raise ValueError("only cv2.COLORMAP_JET (2) is built (feature_extraction.py:227)")
        vis = np.zeros((self.cart_rows, self.cart_cols) + ((3,) if jet else ()), np.uint8) if want_vis else None
        cap = int(cap)
        while True:
            cloud = np.zeros((cap, 2), np.float32)
            n, n_raw = _C.c_int32(0), _C.c_int32(0)
            with self.ctx.lock:
                if jet:     # (the entry point with a flags argument, without a store)
                    ret = self.ctx.lib.sfe_feature_extract_ping_store(
                        self.ctx.handle, self.handle, None, 0, _L.ptr(img, _C.c_uint8), code, int(train_hs), int(guard_hs),
                        int(k), float(tau), _gate_u8(threshold), float(resolution), float(radius), int(min_points), cap,
                        PING_VIS_JET, None, _C.byref(n), _C.byref(n_raw), _L.ptr(cloud, _C.c_float), _L.ptr(vis, _C.c_uint8))
                else:
                    ret = self.ctx.lib.sfe_feature_extract_ping(
                        self.ctx.handle, self.handle, _L.ptr(img, _C.c_uint8), code, int(train_hs), int(guard_hs), int(k),
                        float(tau), _gate_u8(threshold), float(resolution), float(radius), int(min_points), cap,
                        _L.ptr(cloud, _C.c_float), _C.byref(n), _C.byref(n_raw),
                        _L.ptr(vis, _C.c_uint8) if want_vis else None)
            if ret == _L.SFE_ERR_CAP and n_raw.value > cap and n_raw.value <= 65536:
                cap = int(n_raw.value)
                continue
            if ret == _L.SFE_ERR_CAP:
                return None                     # more points than the resident filter holds: per-stage path
            self.ctx._check(ret)
            if n.value < 0:
                return None
            return cloud[:n.value].copy(), vis

    def feature_extract_store(self, img, alg, cfar_params, threshold, resolution, radius, min_points, store, stamp=0,
                              flags=1, want_cloud=False, want_vis=False, cap=16384, colormap=None):
        """``feature_extract`` with the cloud left in a ``store.CloudStore`` (sfe_feature_extract_ping_store): the
        filtered cloud never crosses PCIe unless ``want_cloud`` asks for the publishable copy.  flags: store.NEGATE_Y
        (default) keeps it as the SLAM node holds it (slam_ros.py:170).
        -> (handle, n, cloud or None, vis or None), or None when the resident filter cannot take the ping."""
        from .cfar import _gate_u8
        img = np.ascontiguousarray(img, np.uint8)
        if img.shape != (self.polar_rows, self.polar_cols):
            raise ValueError("feature_extract: image shape %r does not match the geometry" % (img.shape,))
        code = _L.ALG[alg]
        if code == 3:
            train_hs, guard_hs, k, tau = cfar_params
        else:
            (train_hs, guard_hs, tau), k = cfar_params, 0
        jet = want_vis and colormap is not None
        if jet and int(colormap) != COLORMAP_JET:
            raise ValueError("only cv2.COLORMAP_JET (2) is built (feature_extraction.py:227)")
        vis = np.zeros((self.cart_rows, self.cart_cols) + ((3,) if jet else ()), np.uint8) if want_vis else None
        flags = int(flags) | (PING_VIS_JET if jet else 0)
        cap = int(cap)
        while True:
            cloud = np.zeros((cap, 2), np.float32) if want_cloud else None
            n, n_raw, h = _C.c_int32(0), _C.c_int32(0), _C.c_int32(-1)
            with self.ctx.lock:
                ret = self.ctx.lib.sfe_feature_extract_ping_store(
                    self.ctx.handle, self.handle, store.handle, int(stamp), _L.ptr(img, _C.c_uint8), code, int(train_hs),
                    int(guard_hs), int(k), float(tau), _gate_u8(threshold), float(resolution), float(radius),
                    int(min_points), cap, int(flags), _C.byref(h), _C.byref(n), _C.byref(n_raw),
                    _L.ptr(cloud, _C.c_float) if want_cloud else None, _L.ptr(vis, _C.c_uint8) if want_vis else None)
            if ret == _L.SFE_ERR_CAP and n_raw.value > cap and n_raw.value <= 65536:
                cap = int(n_raw.value)
                continue
            if ret == _L.SFE_ERR_CAP and n_raw.value > cap:
                return None         # more than 65 536 raw detections: the caller takes the per-stage path
            self.ctx._check(ret)    # (SFE_ERR_CAP with the points inside the capacity: the STORE is full -> raises)
            if n.value < 0:
                return None
            return h.value, n.value, (cloud[:n.value].copy() if want_cloud else None), vis

    def close(self):
        if self.handle is not None:
            self.ctx.lib.sfe_geom_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def build_maps(bearings, range_resolution, num_ranges):
    """The body of FeatureExtraction.generate_map_xy (feature_extraction.py:141-173), host side
    like the reference (numpy float64 math -> float32 maps, scipy interp1d for bearing->column).
    Returns (res, height, rows, width, cols, map_x, map_y)."""
    to_rad = lambda bearing: bearing * np.pi / 18000
    res = range_resolution
    height = num_ranges * res
    rows = num_ranges
    width = np.sin(to_rad(bearings[-1] - bearings[0]) / 2) * height * 2
    cols = int(np.ceil(width / res))

    rad = to_rad(np.asarray(bearings, dtype=np.float32))
    col_of_bearing = interp1d(rad, range(len(rad)), kind="linear", bounds_error=False,
                              fill_value=-1, assume_sorted=True)
    yy = np.arange(rows).reshape(-1, 1)
    xx = np.arange(cols).reshape(1, -1)
    x = res * (rows - yy) + np.zeros_like(xx)           # forward distance of the pixel row
    y = res * (-cols / 2.0 + xx + 0.5) + np.zeros_like(yy)  # lateral offset of the pixel column
    b = np.arctan2(y, x) * 1
    r = np.sqrt(np.square(x) + np.square(y))
    map_y = np.asarray(r / res, dtype=np.float32)
    map_x = np.asarray(col_of_bearing(b), dtype=np.float32)
    return res, height, rows, width, cols, map_x, map_y


class FeatureExtraction(object):
    """Sonar image -> in-plane feature cloud, on the GPU."""

    def __init__(self, ctx=None):
        self.ctx = ctx
        # CFAR defaults (feature_extraction.py:37-45)
        self.Ntc, self.Ngc, self.Pfa, self.rank = 40, 10, 1e-2, None
        self.alg = "SOCA"
        self.detector = None
        self.threshold = 0
        # point-cloud defaults (feature_extraction.py:47-53)
        self.resolution = 0.5
        self.outlier_filter_radius = 1.0
        self.outlier_filter_min_points = 5
        self.skip = 5
        # polar -> Cartesian state (feature_extraction.py:58-70)
        self.res = self.height = self.rows = self.width = self.cols = None
        self.map_x = self.map_y = None
        self.geometry = None
        self.feature_img = None
        self.make_vis_image = False
        self.vis_colormap = None   # COLORMAP_JET: feature_img is the bgr8 image the node publishes (:226-228), one pass
        self.fused = True          # callback() = one sfe_feature_extract_ping call (False: the per-stage calls)

    # ---- configuration: the rosparam keys of init_node (feature_extraction.py:83-110) ----
    def load_yaml(self, path):
        with open(path, "r") as fh:
            cfg = yaml.safe_load(fh)
        self.Ntc = cfg["CFAR"]["Ntc"]
        self.Ngc = cfg["CFAR"]["Ngc"]
        self.Pfa = cfg["CFAR"]["Pfa"]
        self.rank = cfg["CFAR"]["rank"]
        self.alg = cfg["CFAR"].get("alg", "SOCA")
        self.threshold = cfg["filter"]["threshold"]
        self.resolution = cfg["filter"]["resolution"]
        self.outlier_filter_radius = cfg["filter"]["radius"]
        self.outlier_filter_min_points = cfg["filter"]["min_points"]
        self.skip = cfg["filter"]["skip"]
        self.configure()

    def configure(self):
        self.detector = CFAR(self.Ntc, self.Ngc, self.Pfa, self.rank)

    def _context(self):
        if self.ctx is None:
            self.ctx = _L.default_context()
        return self.ctx

    def generate_map_xy(self, ping):
        """feature_extraction.py:134-173; cached until the ping geometry changes (:150-151)."""
        res = ping.range_resolution
        height = ping.num_ranges * res
        rows = ping.num_ranges
        width = np.sin((ping.bearings[-1] - ping.bearings[0]) * np.pi / 18000 / 2) * height * 2
        cols = int(np.ceil(width / res))
        if (self.res, self.height, self.rows, self.width, self.cols) == (res, height, rows, width, cols):
            return
        (self.res, self.height, self.rows, self.width, self.cols,
         self.map_x, self.map_y) = build_maps(ping.bearings, res, rows)
        if self.geometry is not None:
            self.geometry.close()
        self.geometry = Geometry(self._context(), self.map_x, self.map_y,
                                 (rows, len(ping.bearings)), self.width, self.height)

    # ---- stages, exposed separately for tests and for resident pipelines ----
    def detect(self, img):
        """peaks = detector.detect(img, alg); peaks &= img > threshold  (:223-224)."""
        return self.detector.detect_gated(img, self.alg, self.threshold)

    def extract(self, peaks):
        """peaks -> (locs, points) (:231-238)."""
        return self.geometry.extract(peaks)

    def callback(self, ping):
        """feature_extraction.py:196-252 without the ROS publishers.  Returns the N x 2 float
        feature cloud (y_forward, x_lateral); skipped frames return [[nan, nan]] (:201-207)."""
        if ping.ping_id % self.skip != 0:
            self.feature_img = None
            return np.array([[np.nan, np.nan]])
        img = ping.image
        self.generate_map_xy(ping)
        if self.fused and np.asarray(img).dtype == np.uint8:
            # the live path: the whole chain below in one library call (bit-identical to it)
            out = self.geometry.feature_extract(img, self.alg, self.detector.params[self.alg], self.threshold,
                                                self.resolution, self.outlier_filter_radius,
                                                self.outlier_filter_min_points, want_vis=self.make_vis_image,
                                                colormap=self.vis_colormap)
            if out is not None:
                points, vis = out
                if self.make_vis_image:
                    self.feature_img = vis
                return points
        return self._callback_stages(img)

    def callback_store(self, ping, store, stamp=0, publish=False):
        """``callback`` for a SLAM front end that lives in the same process (replay.FrontEnd with a store): the cloud
        stays on the device as a new slot of ``store`` in the SLAM node's convention (forward, -lateral;
        slam_ros.py:170).  -> (handle, n_points, cloud): ``cloud`` is the publishable N x 2 copy
        (feature_extraction.py:175-193) when ``publish`` is set, else None -- the wire bytes exist only on request.
        Skipped frames (:201-207) -> (-1, 0, [[nan, nan]])."""
        if ping.ping_id % self.skip != 0:
            self.feature_img = None
            return -1, 0, np.array([[np.nan, np.nan]])
        img = ping.image
        self.generate_map_xy(ping)
        if np.asarray(img).dtype == np.uint8:
            out = self.geometry.feature_extract_store(img, self.alg, self.detector.params[self.alg], self.threshold,
                                                      self.resolution, self.outlier_filter_radius,
                                                      self.outlier_filter_min_points, store, stamp=stamp,
                                                      want_cloud=publish, want_vis=self.make_vis_image,
                                                      colormap=self.vis_colormap)
            if out is not None:
                h, n, cloud, vis = out
                if self.make_vis_image:
                    self.feature_img = vis
                return h, n, cloud
        # pings the one-call path cannot take (non-uint8 images, clouds beyond the resident filter): the per-stage
        # chain on the host API, then one upload of the finished cloud
        points = self._callback_stages(img)
        pts32 = np.ascontiguousarray(points, np.float32).reshape(-1, 2)
        h = store.put(np.c_[pts32[:, 0], -1 * pts32[:, 1]], stamp)
        if store.counts([h])[0] < 0:        # the pool had no room: no slot is kept, and the caller hears about it
            store.truncate(h)
            raise _L.SonarFEError("libsonarfe error %d: the cloud store is full (%d points did not fit its pool)"
                                  % (_L.SFE_ERR_CAP, len(pts32)))
        return h, len(pts32), (points if publish else None)

    def _callback_stages(self, img):
        peaks = self.detect(img)
        if self.make_vis_image:
            self.feature_img = self.geometry.remap(img, self.vis_colormap)  # :226-227
        _, points = self.extract(peaks)
        if len(points) and self.resolution > 0:
            points = pcl.downsample(points, self.resolution)  # :241-242
        if self.outlier_filter_min_points > 1 and len(points) > 0:
            points = pcl.remove_outlier(points, self.outlier_filter_radius,
                                        self.outlier_filter_min_points)  # :245-249
        return points
