"""ctypes binding of libsonarfe.so (C ABI: include/sonarfe.h).

Host code stays Python, exactly as in the reference (rospy nodes calling native modules);
this module is the only place that touches the shared library.  There is NO CPU fallback:
if the library or a gfx950 device is missing, every compute call raises.
"""
import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SONARFE_LIB: load another build of the same library (kernel A/B runs inside one process pool / one GPU call)
LIB_PATH = os.environ.get("SONARFE_LIB") or os.path.join(_HERE, "libsonarfe.so")

SFE_ERR_CAP = -4

ALG = {"CA": 0, "SOCA": 1, "GOCA": 2, "OS": 3}

ICP_STATUS_MESSAGES = {
    0: "success",
    1: "no outlier to filter",
    2: "ErrorMnimizer: no point to minimize",
    3: "abs rotation norm not a number",
    4: "abs translation norm not a number",
    5: "point-to-plane system not positive definite",
    6: "internal: the workgroups sharing one job were not resident together (sfe_icp_set_tuning bit 4)",
}


class SonarFEError(RuntimeError):
    """Hard error from libsonarfe (bad argument, HIP failure, no device)."""


class IcpParams(C.Structure):
    """Mirror of ``struct sfe_icp_params`` (include/sonarfe.h)."""

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
    ]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


_u8p = C.POINTER(C.c_uint8)
_f32p = C.POINTER(C.c_float)
_f64p = C.POINTER(C.c_double)
_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)
_vp = C.c_void_p

# every symbol include/sonarfe.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "sfe_version": (C.c_char_p, []),
    "sfe_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "sfe_ctx_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "sfe_ctx_destroy": (None, [_vp]),
    "sfe_last_error": (C.c_char_p, [_vp]),
    "sfe_sync": (C.c_int, [_vp]),
    "sfe_device_name": (C.c_int, [_vp, C.c_char_p, C.c_int]),
    "sfe_malloc": (C.c_int, [_vp, C.c_size_t, C.POINTER(_vp)]),
    "sfe_free": (C.c_int, [_vp, _vp]),
    "sfe_memcpy_h2d": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "sfe_memcpy_d2h": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "sfe_memset": (C.c_int, [_vp, _vp, C.c_int, C.c_size_t]),
    "sfe_host_alloc": (C.c_int, [_vp, C.c_size_t, C.POINTER(_vp)]),
    "sfe_host_free": (C.c_int, [_vp, _vp]),
    "sfe_memcpy_h2d_async": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "sfe_stream_fence": (C.c_int, [_vp, C.c_int]),
    "sfe_debug_read_scratch": (C.c_int, [_vp, C.c_int, _vp, C.c_size_t]),
    "sfe_timer_start": (C.c_int, [_vp]),
    "sfe_timer_stop": (C.c_int, [_vp, _f32p]),
    "sfe_cfar_u8": (C.c_int, [_vp, _u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                              C.c_double, C.c_int, _u8p, _f32p]),
    "sfe_cfar_f32": (C.c_int, [_vp, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                               C.c_double, _u8p, _f32p]),
    "sfe_cfar_u8_batch_dev": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_int, C.c_int, C.c_double, C.c_int, _vp, _vp]),
    "sfe_cfar_u8_bits_batch_dev": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                             C.c_int, C.c_int, C.c_double, C.c_int, _vp]),
    "sfe_cfar_set_tuning": (C.c_int, [_vp, C.c_int, C.c_int]),
    "sfe_geom_create": (C.c_int, [_vp, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.c_double, C.c_double, C.POINTER(_vp)]),
    "sfe_geom_destroy": (None, [_vp]),
    "sfe_remap_u8": (C.c_int, [_vp, _vp, _u8p, _u8p]),
    "sfe_remap_u8_dev": (C.c_int, [_vp, _vp, _vp, _vp]),
    "sfe_colormap_lut": (C.c_int, [C.c_int, _u8p]),
    "sfe_remap_u8_colormap": (C.c_int, [_vp, _vp, _u8p, C.c_int, _u8p]),
    "sfe_remap_u8_colormap_dev": (C.c_int, [_vp, _vp, _vp, C.c_int, _vp]),
    "sfe_feature_extract_ping": (C.c_int, [_vp, _vp, _u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_float,
                                           C.c_double, C.c_int, C.c_int64, _f32p, _i32p, _i32p, _u8p]),
    "sfe_extract_points": (C.c_int, [_vp, _vp, _u8p, C.c_int64, _i64p, _f64p, _i64p]),
    "sfe_extract_set_tuning": (C.c_int, [_vp, C.c_int]),
    "sfe_extract_points_batch_dev": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int64, _vp, _vp]),
    "sfe_extract_points_bits_batch_dev": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int64, _vp, _vp]),
    "sfe_extract_points_bits_staged_dev": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int64, _vp, C.c_int, _vp]),
    "sfe_match": (C.c_int, [_vp, _f32p, C.c_int, _f32p, C.c_int, C.c_float, _i32p, _f32p]),
    "sfe_match_knn": (C.c_int, [_vp, _f32p, C.c_int, _f32p, C.c_int, C.c_int, C.c_float, _i32p, _f32p]),
    "sfe_knn_density": (C.c_int, [_vp, _f32p, C.c_int, C.c_int, _f32p]),
    "sfe_remove_outlier": (C.c_int, [_vp, _f32p, C.c_int, C.c_double, C.c_int, _f32p,
                                     C.POINTER(C.c_int)]),
    "sfe_downsample": (C.c_int, [_vp, _f32p, C.c_int, C.c_float, _f32p, _i32p, C.POINTER(C.c_int)]),
    "sfe_icp_compute": (C.c_int, [_vp, C.POINTER(IcpParams), _f32p, C.c_int, _f32p, C.c_int, _f32p,
                                  _f32p, C.POINTER(C.c_int)]),
    "sfe_icp_compute_guesses": (C.c_int, [_vp, C.POINTER(IcpParams), _f32p, C.c_int, _f32p, C.c_int,
                                          _f32p, C.c_int, _f32p, _i32p, _i32p]),
    "sfe_icp_compute_pairs": (C.c_int, [_vp, C.POINTER(IcpParams), _f32p, _i32p, _f32p, _i32p, _f32p, C.c_int,
                                        _f32p, _i32p, _i32p]),
    "sfe_icp_compute_jobs": (C.c_int, [_vp, C.POINTER(IcpParams), _f32p, C.c_int, _f32p, C.c_int, _i32p, _f32p, C.c_int,
                                       _f32p, _i32p, _i32p]),
    "sfe_icp_set_tuning": (C.c_int, [_vp, C.c_int]),
    "sfe_icp_get_profile": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_longlong)]),
    "sfe_icp_batch_dev": (C.c_int, [_vp, C.POINTER(IcpParams), _vp, _i32p, _vp, _i32p, _vp, C.c_int,
                                    _vp, _vp, _vp]),
    "sfe_icp_jobs_dev": (C.c_int, [_vp, C.POINTER(IcpParams), _vp, _vp, _i32p, _vp, C.c_int, _vp, _vp, _vp]),
    "sfe_cloud_filter_batch_dev": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int64, C.c_float, C.c_double, C.c_int,
                                             _vp, _vp]),
    "sfe_cloud_filter_staged_dev": (C.c_int, [_vp, C.c_int, C.c_int64, C.c_float, C.c_double, C.c_int, _vp, _vp]),
    "sfe_cloud_store_create": (C.c_int, [_vp, C.c_int64, C.c_int32, C.POINTER(_vp)]),
    "sfe_cloud_store_destroy": (None, [_vp]),
    "sfe_cloud_store_count": (C.c_int, [_vp]),
    "sfe_cloud_store_put": (C.c_int, [_vp, _vp, C.c_int64, _f32p, C.c_int, _i32p]),
    "sfe_cloud_store_put_batch_dev": (C.c_int, [_vp, _vp, _i64p, _vp, _vp, C.c_int, C.c_int64, C.c_int, _i32p]),
    "sfe_cloud_store_meta": (C.c_int, [_vp, _vp, C.c_int32, C.c_int32, _i64p, _i64p, _i32p]),
    "sfe_cloud_store_read": (C.c_int, [_vp, _vp, C.c_int32, _f32p, C.c_int, C.POINTER(C.c_int)]),
    "sfe_cloud_store_truncate": (C.c_int, [_vp, _vp, C.c_int32]),
    "sfe_cloud_store_get_points": (C.c_int, [_vp, _vp, _i32p, _f32p, C.c_int, C.c_int, C.c_float, C.c_int, _i64p, _i32p]),
    "sfe_icp_store_jobs_dev": (C.c_int, [_vp, C.POINTER(IcpParams), _vp, _i32p, _vp, C.c_int, _vp, _vp, _vp]),
    "sfe_icp_store_compute": (C.c_int, [_vp, C.POINTER(IcpParams), _vp, _i32p, _f32p, C.c_int, _f32p, _i32p, _i32p]),
    "sfe_cloud_store_overlap": (C.c_int, [_vp, _vp, _i32p, _f32p, C.c_int, C.c_float, C.c_int, _i32p]),
    "sfe_feature_extract_ping_store": (C.c_int, [_vp, _vp, _vp, C.c_int64, _u8p, C.c_int, C.c_int, C.c_int, C.c_int,
                                                 C.c_double, C.c_int, C.c_float, C.c_double, C.c_int, C.c_int64, C.c_int,
                                                 _i32p, _i32p, _i32p, _f32p, _u8p]),
    "sfe_costgrid_create": (C.c_int, [_vp, _i32p, _i32p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_vp)]),
    "sfe_costgrid_destroy": (None, [_vp]),
    "sfe_costgrid_download": (C.c_int, [_vp, _vp, C.c_int, _u8p]),
    "sfe_matching_cost_batch": (C.c_int, [_vp, _vp, _f32p, C.c_int, _f32p, C.c_int, C.c_float, C.c_float,
                                          C.c_double, C.c_int, _i32p]),
    "sfe_costgrid_create_store": (C.c_int, [_vp, _vp, _i32p, C.c_int, _f32p, _f32p, C.c_float, _i32p, _i32p, C.c_int,
                                            C.POINTER(_vp)]),
    "sfe_matching_cost_store": (C.c_int, [_vp, _vp, _vp, _i32p, _i32p, C.c_int, _f32p, C.c_int, C.c_double, C.c_int, _i32p]),
    "sfe_shgo_sobol_replay": (C.c_int, [C.c_int, _i32p, _i32p, _f64p, _i32p, C.c_int, _u8p, _i32p, _i32p, _i32p]),
    "sfe_matching_cost_store_samples": (C.c_int, [_vp, _vp, _vp, _i32p, _i32p, C.c_int, _f64p, _f64p, _f64p, C.c_int, C.c_double, C.c_int,
                                                  _i32p]),
    "sfe_pose2_sample_transforms": (C.c_int, [_f64p, _f64p, C.c_int, _f64p, C.c_int, _f32p]),
    "sfe_cloud_store_bbox": (C.c_int, [_vp, _vp, _i32p, C.c_int, _f32p]),
    "sfe_cloud_store_get_points_keys": (C.c_int, [_vp, _vp, _i32p, _f32p, _i32p, C.c_int, C.c_float, C.c_int, C.c_int64, _i32p]),
    "sfe_cloud_store_read_keys": (C.c_int, [_vp, _vp, C.c_int32, _i32p, C.c_int, C.POINTER(C.c_int)]),
    "sfe_cloud_store_fov_select": (C.c_int, [_vp, _vp, C.c_int32, _f32p, _f64p, _f64p, C.c_int, C.c_int, _i32p, _i32p, _i32p]),
    "sfe_cloud_store_set_selection": (C.c_int, [_vp, _vp, C.c_int32, _u8p, C.c_int]),
    "sfe_cloud_store_compact_selected": (C.c_int, [_vp, _vp, C.c_int32, C.c_int64, _i32p]),
    "sfe_cloud_store_match_keys": (C.c_int, [_vp, _vp, C.c_int32, _f32p, C.c_int32, C.c_float, C.c_int, C.c_int, _i32p, _i32p]),
}

_lib = None
_lib_lock = threading.Lock()


def load_library():
    """dlopen libsonarfe.so and attach prototypes.  Raises if it has not been built."""
    global _lib
    with _lib_lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise SonarFEError(
                    "%s not found: build it with `make -C sonar_slam_amd/csrc` "
                    "(or python -c 'import __graft_entry__ as g; g.build()'); "
                    "there is no CPU fallback" % LIB_PATH)
            lib = C.CDLL(LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(lib, name)  # AttributeError = ABI drift, fail loudly
                fn.restype = res
                fn.argtypes = args
            _lib = lib
    return _lib


def ptr(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class DeviceBuffer(object):
    """A device allocation owned by a Context (for resident *_dev pipelines)."""

    def __init__(self, ctx, nbytes):
        self.ctx = ctx
        self.nbytes = int(nbytes)
        p = _vp()
        ctx._check(ctx.lib.sfe_malloc(ctx.handle, self.nbytes, C.byref(p)))
        self.ptr = p

    def upload(self, arr, offset=0):
        arr = np.ascontiguousarray(arr)
        assert offset + arr.nbytes <= self.nbytes
        dst = _vp(self.ptr.value + offset)
        self.ctx._check(self.ctx.lib.sfe_memcpy_h2d(self.ctx.handle, dst, arr.ctypes.data, arr.nbytes))

    def upload_async(self, pinned, offset=0):
        """enqueue-only upload from a ``Context.host_alloc`` array on the context's copy stream (order it against
        the kernels with ``Context.fence``)"""
        assert offset + pinned.nbytes <= self.nbytes
        dst = _vp(self.ptr.value + offset)
        self.ctx._check(self.ctx.lib.sfe_memcpy_h2d_async(self.ctx.handle, dst, pinned.ctypes.data, pinned.nbytes))

    def download(self, dtype, count, offset=0):
        out = np.empty(count, dtype)
        assert offset + out.nbytes <= self.nbytes
        src = _vp(self.ptr.value + offset)
        self.ctx._check(self.ctx.lib.sfe_memcpy_d2h(self.ctx.handle, out.ctypes.data, src, out.nbytes))
        return out

    def zero(self):
        self.ctx._check(self.ctx.lib.sfe_memset(self.ctx.handle, self.ptr, 0, self.nbytes))

    def free(self):
        if self.ptr is not None and self.ptr.value:
            self.ctx.lib.sfe_free(self.ctx.handle, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Context(object):
    """One device + one HIP stream (``sfe_ctx``).  Not re-entrant: calls are serialised."""

    def __init__(self, device=0):
        self.lib = load_library()
        h = _vp()
        rc = self.lib.sfe_ctx_create(int(device), C.byref(h))
        if rc != 0:
            raise SonarFEError("sfe_ctx_create(%d) failed (%d): %s"
                               % (device, rc, self.lib.sfe_last_error(None).decode()))
        self.handle = h
        self.device = int(device)
        self.lock = threading.RLock()

    def _check(self, rc):
        if rc < 0:
            raise SonarFEError("libsonarfe error %d: %s"
                               % (rc, self.lib.sfe_last_error(self.handle).decode()))
        return rc

    def sync(self):
        self._check(self.lib.sfe_sync(self.handle))

    def name(self):
        buf = C.create_string_buffer(256)
        self._check(self.lib.sfe_device_name(self.handle, buf, 256))
        return buf.value.decode()

    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def host_alloc(self, shape, dtype=np.uint8):
        """pinned host array (sfe_host_alloc); free it with ``host_free`` (the array must not be used afterwards)"""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = _vp()
        self._check(self.lib.sfe_host_alloc(self.handle, n, C.byref(p)))
        buf = (C.c_char * max(n, 1)).from_address(p.value)
        arr = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[arr.ctypes.data] = p
        return arr

    def host_free(self, arr):
        p = getattr(self, "_pinned", {}).pop(arr.ctypes.data, None)
        if p is not None:
            self._check(self.lib.sfe_host_free(self.handle, p))

    def fence(self, what):
        """0: kernels behind the uploads enqueued so far; 1: uploads behind the kernels enqueued so far;
        2: the host waits for the uploads (sfe_stream_fence)"""
        self._check(self.lib.sfe_stream_fence(self.handle, int(what)))

    def timer_start(self):
        self._check(self.lib.sfe_timer_start(self.handle))

    def timer_stop(self):
        ms = C.c_float(0)
        self._check(self.lib.sfe_timer_stop(self.handle, C.byref(ms)))
        return ms.value

    def close(self):
        if self.handle is not None:
            # pinned blocks handed out by host_alloc die with the context (sfe_host_free waits for the copy stream);
            # numpy views of them must not be used afterwards
            for p in list(getattr(self, "_pinned", {}).values()):
                self.lib.sfe_host_free(self.handle, p)
            self._pinned = {}
            self.lib.sfe_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = None
_default_lock = threading.Lock()


def default_context():
    """Process-wide context on device $SONARFE_DEVICE (default 0), created on first use."""
    global _default_ctx
    with _default_lock:
        if _default_ctx is None:
            _default_ctx = Context(int(os.environ.get("SONARFE_DEVICE", "0")))
    return _default_ctx


def device_count():
    lib = load_library()
    n = C.c_int(0)
    lib.sfe_device_count(C.byref(n))
    return n.value
