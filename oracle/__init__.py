"""ctypes front-end of the CPU oracle (``sonar_oracle.c``).

TEST INFRASTRUCTURE ONLY -- the product package ``sonar_slam_amd`` never imports
this module.  Allowed importers: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``.

Every function cites the reference lines it restates in ``sonar_oracle.c``.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libsonar_oracle.so")
# the reference's OWN cfar.cpp compiled unmodified (oracle/Makefile `ref`); built only where
# /root/reference exists, travels to the GPU box as a prebuilt .so
_REF_SO = os.path.join(_HERE, "_ref", "libcfar_ref.so")

ALG = {"CA": 0, "SOCA": 1, "GOCA": 2, "OS": 3}

ICP_STATUS_MESSAGES = {
    0: "success",
    1: "no outlier to filter",
    2: "ErrorMnimizer: no point to minimize",
    3: "abs rotation norm not a number",
    4: "abs translation norm not a number",
    5: "point-to-plane system not positive definite",
}


def build(force=False):
    """Compile the oracle with the committed Makefile (gcc, -O3, single thread)."""
    src = os.path.join(_HERE, "sonar_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


class IcpParams(C.Structure):
    _fields_ = [
        ("matcher_max_dist", C.c_float),
        ("use_max_dist_filter", C.c_int),
        ("max_dist_filter", C.c_float),
        ("use_trimmed_filter", C.c_int),
        ("trim_ratio", C.c_float),
        ("minimizer", C.c_int),
        ("max_iter", C.c_int),
        ("use_diff_checker", C.c_int),
        ("min_diff_rot", C.c_float),
        ("min_diff_trans", C.c_float),
        ("smooth_len", C.c_int),
        ("normals_knn", C.c_int),
        ("precision", C.c_int),
    ]


def shipped_icp_params(minimizer=0, precision=1, **over):
    """The chain of bruce_slam/config/icp.yaml:1-31."""
    p = dict(matcher_max_dist=10.0, use_max_dist_filter=1, max_dist_filter=3.0,
             use_trimmed_filter=1, trim_ratio=0.8, minimizer=minimizer, max_iter=40,
             use_diff_checker=1, min_diff_rot=0.01, min_diff_trans=0.1, smooth_len=4,
             normals_knn=10, precision=precision)
    p.update(over)
    return IcpParams(**p)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        u8p, f32p, f64p, i64p, i32p = (C.POINTER(C.c_uint8), C.POINTER(C.c_float),
                                       C.POINTER(C.c_double), C.POINTER(C.c_int64),
                                       C.POINTER(C.c_int32))
        L.orc_cfar_f32.argtypes = [f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_double, u8p, f32p]
        L.orc_cfar_u8.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.c_double, u8p, f32p]
        L.orc_gate_u8.argtypes = [u8p, C.c_size_t, C.c_int, u8p]
        L.orc_remap_u8.argtypes = [u8p, C.c_int, C.c_int, f32p, f32p, C.c_int, C.c_int, u8p]
        L.orc_nonzero.argtypes = [u8p, C.c_int, C.c_int, i64p, C.c_int64]
        L.orc_nonzero.restype = C.c_int64
        L.orc_px_to_m.argtypes = [i64p, C.c_int64, C.c_int, C.c_int, C.c_double, C.c_double, f64p]
        L.orc_match.argtypes = [f32p, C.c_int, f32p, C.c_int, C.c_float, i32p, f32p]
        L.orc_match_knn.argtypes = [f32p, C.c_int, f32p, C.c_int, C.c_int, C.c_float, i32p, f32p]
        L.orc_knn_density.argtypes = [f32p, C.c_int, C.c_int, f32p]
        L.orc_knn_density.restype = C.c_int
        L.orc_normals2d.argtypes = [f32p, C.c_int, C.c_int, f32p]
        L.orc_icp.argtypes = [C.POINTER(IcpParams), f32p, C.c_int, f32p, C.c_int, f32p, f32p,
                              C.POINTER(C.c_int)]
        L.orc_remove_outlier.argtypes = [f32p, C.c_int, C.c_double, C.c_int, f32p]
        L.orc_bilinear_tab.argtypes = [C.POINTER(C.c_int16)]
        L.orc_downsample.argtypes = [f32p, C.c_int, C.c_float, f32p, i32p]
        L.orc_set_kdtree.argtypes = [C.c_int]
        L.orc_ellipse_spans.argtypes = [C.c_int, i32p, i32p]
        L.orc_cost_grid.argtypes = [i32p, i32p, C.c_int, C.c_int, C.c_int, C.c_int, u8p]
        L.orc_matching_cost.argtypes = [u8p, C.c_int, C.c_int, f32p, C.c_int, f32p, C.c_int, C.c_float,
                                        C.c_float, C.c_float, i32p]
        L.orc_transform_points.argtypes = [f32p, C.c_int, f32p, C.c_int, f32p]
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def cfar(img, alg, train_hs, guard_hs, tau, k=0, want_threshold=False):
    """cfar.cpp:10-192 on a 2-D image (uint8 -> cast to float like pybind, else float32)."""
    img = np.asarray(img)
    rows, cols = img.shape
    mask = np.zeros((rows, cols), np.uint8)
    thr = np.zeros((rows, cols), np.float32) if want_threshold else None
    tp = _p(thr, C.c_float) if want_threshold else None
    if img.dtype == np.uint8:
        a = np.ascontiguousarray(img)
        rc = lib().orc_cfar_u8(_p(a, C.c_uint8), rows, cols, ALG[alg], train_hs, guard_hs, int(k),
                               float(tau), _p(mask, C.c_uint8), tp)
    else:
        a = np.ascontiguousarray(img, np.float32)
        rc = lib().orc_cfar_f32(_p(a, C.c_float), rows, cols, ALG[alg], train_hs, guard_hs, int(k),
                                float(tau), _p(mask, C.c_uint8), tp)
    if rc:
        raise ValueError("oracle cfar rc=%d" % rc)
    return (mask, thr) if want_threshold else mask


def gate(img, mask, threshold):
    """feature_extraction.py:224."""
    img = np.ascontiguousarray(img, np.uint8)
    m = np.ascontiguousarray(mask, np.uint8).copy()
    lib().orc_gate_u8(_p(img, C.c_uint8), img.size, int(threshold), _p(m, C.c_uint8))
    return m


def remap_u8(src, map_x, map_y):
    """cv2.remap(src, map_x, map_y, INTER_LINEAR) on uint8 (feature_extraction.py:226,231)."""
    src = np.ascontiguousarray(src, np.uint8)
    mx = np.ascontiguousarray(map_x, np.float32)
    my = np.ascontiguousarray(map_y, np.float32)
    dst = np.zeros(mx.shape, np.uint8)
    lib().orc_remap_u8(_p(src, C.c_uint8), src.shape[0], src.shape[1], _p(mx, C.c_float),
                       _p(my, C.c_float), mx.shape[0], mx.shape[1], _p(dst, C.c_uint8))
    return dst


def nonzero(img):
    """np.c_[np.nonzero(img)] (feature_extraction.py:232)."""
    img = np.ascontiguousarray(img, np.uint8)
    cap = int(img.size)
    rc = np.zeros((max(cap, 1), 2), np.int64)
    n = lib().orc_nonzero(_p(img, C.c_uint8), img.shape[0], img.shape[1], _p(rc, C.c_int64), cap)
    return rc[:n].copy()


def px_to_m(rc, rows, cols, width, height):
    """feature_extraction.py:235-238 -> points [y_fwd, x_lat] float64."""
    rc = np.ascontiguousarray(rc, np.int64).reshape(-1, 2)
    pts = np.zeros((len(rc), 2), np.float64)
    lib().orc_px_to_m(_p(rc, C.c_int64), len(rc), rows, cols, float(width), float(height),
                      _p(pts, C.c_double))
    return pts


def match(ref, pts, max_dist):
    """pcl.match(ref, in, 1, max_dist) (pcl.cpp:161-174) -> (ids [1xN] int32, d2 [1xN] f32)."""
    ref = np.ascontiguousarray(ref, np.float32).reshape(-1, 2)
    pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
    ids = np.zeros(len(pts), np.int32)
    d2 = np.zeros(len(pts), np.float32)
    lib().orc_match(_p(ref, C.c_float), len(ref), _p(pts, C.c_float), len(pts), float(max_dist),
                    _p(ids, C.c_int32), _p(d2, C.c_float))
    return ids[None, :], d2[None, :]


def match_knn(ref, pts, knn, max_dist):
    """pcl.match(ref, in, knn, max_dist) (pcl.cpp:161-174) -> (ids [knn x N] int32, d2 [knn x N] f32), ascending."""
    ref = np.ascontiguousarray(ref, np.float32).reshape(-1, 2)
    pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
    ids = np.zeros((knn, len(pts)), np.int32)
    d2 = np.zeros((knn, len(pts)), np.float32)
    lib().orc_match_knn(_p(ref, C.c_float), len(ref), _p(pts, C.c_float), len(pts), int(knn), float(max_dist),
                        _p(ids, C.c_int32), _p(d2, C.c_float))
    return ids, d2


def knn_density(pts, knn):
    """densities of SurfaceNormalDataPointsFilter{knn, keepDensities} (pcl.cpp:81-88)"""
    pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
    out = np.zeros(len(pts), np.float32)
    if lib().orc_knn_density(_p(pts, C.c_float), len(pts), int(knn), _p(out, C.c_float)):
        raise RuntimeError("Requesting more points than available in cloud")
    return out


def normals2d(tgt, k):
    tgt = np.ascontiguousarray(tgt, np.float32).reshape(-1, 2)
    out = np.zeros_like(tgt)
    lib().orc_normals2d(_p(tgt, C.c_float), len(tgt), int(k), _p(out, C.c_float))
    return out


def icp(src, tgt, guess, params=None):
    """pcl.ICP.compute (pcl.cpp:198-212) -> (status:int, T 3x3 f32, iterations)."""
    params = params or shipped_icp_params()
    src = np.ascontiguousarray(src, np.float32).reshape(-1, 2)
    tgt = np.ascontiguousarray(tgt, np.float32).reshape(-1, 2)
    g = np.ascontiguousarray(guess, np.float32).reshape(3, 3)
    T = np.zeros((3, 3), np.float32)
    it = C.c_int(0)
    st = lib().orc_icp(C.byref(params), _p(src, C.c_float), len(src), _p(tgt, C.c_float), len(tgt),
                       _p(g, C.c_float), _p(T, C.c_float), C.byref(it))
    return st, T, it.value


def remove_outlier(pts, radius, min_points):
    pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
    out = np.zeros_like(pts)
    m = lib().orc_remove_outlier(_p(pts, C.c_float), len(pts), float(radius), int(min_points),
                                 _p(out, C.c_float))
    return out[:m].copy()


def bilinear_tab():
    t = np.zeros((1024, 4), np.int16)
    lib().orc_bilinear_tab(_p(t, C.c_int16))
    return t


def downsample(pts, resolution, return_index=False):
    """pcl.downsample(points, resolution) (pcl.cpp:128-141): octree-grid medoid sampling."""
    pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
    out = np.zeros_like(pts)
    idx = np.zeros(len(pts), np.int32)
    # pcl.cpp passes std::to_string(resolution): six decimals survive
    res = np.float32(float("%f" % np.float32(resolution)))
    m = lib().orc_downsample(_p(pts, C.c_float), len(pts), float(res), _p(out, C.c_float), _p(idx, C.c_int32))
    return (out[:m].copy(), idx[:m].copy()) if return_index else out[:m].copy()


def transform_points(points, T, f64_points=True):
    """Keyframe.transform_points (slam_objects.py:178-198) as the scan matcher sees its result: float32 at the pybind
    boundary.  T: 3 x 3 (or 2 x 3) pose matrix, rounded to float32 like `pose.matrix().astype(np.float32)`;
    f64_points: the keyframe cloud is a float64 array of float32 values (the SLAM node, slam_ros.py:169-170 through
    ros_numpy) -- else float32 (sgemm).  -> N x 2 float32"""
    pts = np.ascontiguousarray(points, np.float32).reshape(-1, 2)
    T6 = np.ascontiguousarray(np.asarray(T)[:2, :3], np.float32)
    out = np.zeros_like(pts)
    lib().orc_transform_points(_p(pts, C.c_float), len(pts), _p(T6, C.c_float), 1 if f64_points else 0, _p(out, C.c_float))
    return out


def get_points(clouds, transforms, resolution, f64_points=True):
    """SLAM.get_points with a reference frame (slam.py:229-292): every keyframe cloud moved by its transform
    (ref_pose.between(pose).matrix()), concatenated in frame order, pcl.downsample -> N x 2 float32"""
    parts = [transform_points(c, T, f64_points) for c, T in zip(clouds, transforms)]
    allp = np.concatenate(parts) if parts else np.zeros((0, 2), np.float32)
    return downsample(allp, resolution) if len(allp) else allp


def overlap(source, target, T, max_dist, f64_points=True):
    """SLAM.get_overlap (slam.py:389-424): transform the source, pcl.match(target, source, 1, max_dist), count the
    matched points"""
    ids, _ = match(target, transform_points(source, T, f64_points), max_dist)
    return int(np.sum(ids != -1))


def colormap_jet_lut():
    """cv2.applyColorMap(x, cv2.COLORMAP_JET)'s table, 256 x 3 uint8 BGR (feature_extraction.py:227).  OpenCV's
    colormap.cpp (class Jet, "equals the GNU Octave colormap jet") samples the ramps r = 4x - 3/2 on [3/8, 5/8), 1 on
    [5/8, 7/8), -4x + 9/2 above; g = 4x - 1/2 on [1/8, 3/8), 1 on [3/8, 5/8), -4x + 7/2 on [5/8, 7/8); b = 4x + 1/2
    below 1/8, 1 on [1/8, 3/8), -4x + 5/2 on [3/8, 5/8) at x = i / 255, scales by 255 and rounds to uint8 (half to
    even).  Here: exact rational arithmetic.  Un-vendored third party, PARITY UNPINNED."""
    from fractions import Fraction as F
    lut = np.zeros((256, 3), np.uint8)
    for i in range(256):
        x = F(i, 255)
        r = (4 * x - F(3, 2)) if F(3, 8) <= x < F(5, 8) else (1 if F(5, 8) <= x < F(7, 8) else ((-4 * x + F(9, 2)) if x >= F(7, 8) else 0))
        g = (4 * x - F(1, 2)) if F(1, 8) <= x < F(3, 8) else (1 if F(3, 8) <= x < F(5, 8) else ((-4 * x + F(7, 2)) if F(5, 8) <= x < F(7, 8) else 0))
        b = (4 * x + F(1, 2)) if x < F(1, 8) else (1 if F(1, 8) <= x < F(3, 8) else ((-4 * x + F(5, 2)) if F(3, 8) <= x < F(5, 8) else 0))
        for c, v in enumerate((b, g, r)):
            v = v * 255
            fl = v.numerator // v.denominator
            rem = v - fl
            q = fl + (1 if (rem > F(1, 2) or (rem == F(1, 2) and fl % 2 == 1)) else 0)
            lut[i, c] = min(max(q, 0), 255)
    return lut


def colormap_jet_lut_float_emulation():
    """The same table through the float32 steps OpenCV takes (as far as its published source is remembered: linspace in
    float, the literal per-channel tables as float, interp1 AT the sample points -- y0 + (x - x0) * (y1 - y0) / (x1 - x0)
    in float -- and convertTo(CV_8U, 255)): where float rounding noise could move a tie."""
    f32 = np.float32
    x = np.arange(256) / 255.0
    r = ((x >= 3 / 8) & (x < 5 / 8)) * (4 * x - 3 / 2) + ((x >= 5 / 8) & (x < 7 / 8)) * 1.0 + (x >= 7 / 8) * (-4 * x + 9 / 2)
    g = ((x >= 1 / 8) & (x < 3 / 8)) * (4 * x - 1 / 2) + ((x >= 3 / 8) & (x < 5 / 8)) * 1.0 + ((x >= 5 / 8) & (x < 7 / 8)) * (-4 * x + 7 / 2)
    b = (x < 1 / 8) * (4 * x + 1 / 2) + ((x >= 1 / 8) & (x < 3 / 8)) * 1.0 + ((x >= 3 / 8) & (x < 5 / 8)) * (-4 * x + 5 / 2)
    step = f32(1.0) / f32(255)
    X = np.array([f32(0) + f32(i) * step for i in range(256)], f32)
    out = np.zeros((256, 3), np.uint8)
    for c, ch in enumerate((b, g, r)):
        Y = ch.astype(f32)
        yi = np.zeros(256, f32)
        for i in range(256):
            xi, low, high = X[i], 0, 255
            while high - low > 1:
                cc = low + ((high - low) >> 1)
                if xi > X[cc]:
                    low = cc
                else:
                    high = cc
            yi[i] = Y[low] + (xi - X[low]) * (Y[high] - Y[low]) / (X[high] - X[low])
        out[:, c] = np.clip(np.rint(yi * f32(255)), 0, 255).astype(np.uint8)
    return out


def ellipse_kernel(hs):
    """cv2.getStructuringElement(cv2.MORPH_ELLIPSE, (2hs+1, 2hs+1), (hs, hs)) as a 0/1 array."""
    size = 2 * hs + 1
    j1 = np.zeros(size, np.int32)
    j2 = np.zeros(size, np.int32)
    lib().orc_ellipse_spans(int(hs), _p(j1, C.c_int32), _p(j2, C.c_int32))
    k = np.zeros((size, size), np.uint8)
    for i in range(size):
        k[i, j1[i]:j2[i]] = 1
    return k


def cost_grid(tgt_r, tgt_c, rows, cols, dilate_hs):
    """target_grids after cv2.dilate (slam.py:515-527): rows x cols uint8 0/255."""
    tr = np.ascontiguousarray(tgt_r, np.int32)
    tc = np.ascontiguousarray(tgt_c, np.int32)
    g = np.zeros((rows, cols), np.uint8)
    lib().orc_cost_grid(_p(tr, C.c_int32), _p(tc, C.c_int32), len(tr), rows, cols, int(dilate_hs), _p(g, C.c_uint8))
    return g


def matching_cost(grid, src, T6, xmin, ymin, resolution, f64_points=False):
    """costs (= -hits, slam.py:549-562) of n_poses float32 transforms [n x 6].  f64_points: the source is a float64
    array of float32 values (the SLAM node's keyframe clouds): numpy evaluates the body in double, with the Python
    float resolution; else a float32 cloud (sgemm transform, float32 cell arithmetic with float32(resolution))."""
    grid = np.ascontiguousarray(grid, np.uint8)
    src = np.ascontiguousarray(src, np.float32).reshape(-1, 2)
    T6 = np.ascontiguousarray(T6, np.float32).reshape(-1, 6)
    out = np.zeros(len(T6), np.int32)
    fn = lib().orc_matching_cost2
    fn.argtypes = [C.POINTER(C.c_uint8), C.c_int, C.c_int, C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_float), C.c_int,
                   C.c_float, C.c_float, C.c_double, C.c_int, C.POINTER(C.c_int32)]
    fn(_p(grid, C.c_uint8), grid.shape[0], grid.shape[1], _p(src, C.c_float), len(src), _p(T6, C.c_float), len(T6),
       np.float32(xmin), np.float32(ymin), float(resolution), 1 if f64_points else 0, _p(out, C.c_int32))
    return out


def set_kdtree(on):
    """Route the NN searches of icp / match / normals2d through the oracle's exact kd-tree (same
    neighbours, O(N log N)); bench.py turns it on for the CPU baseline, the parity tests leave it off."""
    lib().orc_set_kdtree(1 if on else 0)


_ref_lib = None


def have_ref_cfar():
    """True if oracle/_ref/libcfar_ref.so (the reference's own cfar.cpp, compiled unmodified against
    the stand-in Eigen / pybind11 headers in oracle/ref_shim/) is available."""
    return os.path.exists(_REF_SO)


def ref_cfar(img, alg, train_hs, guard_hs, tau, k=0, want_threshold=False):
    """The REFERENCE implementation itself: bruce_slam/src/bruce_slam/cpp/cfar.cpp:10-192 through
    the C wrapper oracle/ref_cfar_wrap.cpp (which does what pybind11 does at the boundary: cast-copy
    the image to a float matrix).  Used to pin ``cfar`` above and the HIP kernels."""
    global _ref_lib
    if _ref_lib is None:
        L = C.CDLL(_REF_SO)
        L.ref_cfar.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                               C.c_int, C.POINTER(C.c_uint8), C.POINTER(C.c_float)]
        _ref_lib = L
    a = np.ascontiguousarray(img, np.float32)     # pybind11's numpy -> MatrixXf cast
    rows, cols = a.shape
    mask = np.zeros((rows, cols), np.uint8)
    thr = np.zeros((rows, cols), np.float32) if want_threshold else None
    rc = _ref_lib.ref_cfar(_p(a, C.c_float), rows, cols, ALG[alg], train_hs, guard_hs, int(k), float(tau),
                           1 if want_threshold else 0, _p(mask, C.c_uint8),
                           _p(thr, C.c_float) if want_threshold else None)
    if rc:
        raise ValueError("ref_cfar rc=%d" % rc)
    return (mask, thr) if want_threshold else mask
