"""Drop-in for the reference's pybind11 module ``bruce_slam.cfar``.

Same names and call signatures as ``PYBIND11_MODULE(cfar, m)``
(bruce_slam/src/bruce_slam/cpp/cfar.cpp:194-204); every call runs the HIP kernels of
libsonarfe.so through ctypes.  ``CFAR.py`` of the reference works unmodified on top of it
(``from . import cfar``, CFAR.py:7).

    ca/soca/goca(img, train_hs, guard_hs, tau)        -> uint8 mask   (cfar.cpp:10,30,53)
    os(img, train_hs, guard_hs, k, tau)               -> uint8 mask   (cfar.cpp:76)
    ca2/soca2/goca2/os2(...)                          -> (mask, float32 threshold map)
                                                                      (cfar.cpp:98-192)

Like the pybind Eigen caster the functions accept any numeric 2-D array: uint8 images (what
the sonar driver delivers, feature_extraction.py:211-217) take the integer-exact fast path,
everything else is cast to float32 (``const MatrixXf&``) and uses the float kernel.

Extension (not in the reference): ``detect_gated`` fuses ``peaks &= img > threshold``
(feature_extraction.py:224) into the same kernel.
"""
import ctypes as _C
import math as _math
import operator as _operator

import numpy as _np

from . import _lib as _L


def _as_index(k):
    try:
        return _operator.index(k)
    except TypeError:
        # pybind11 refuses a float for ``int k`` (CFAR.py:24 makes rank a float when it is None)
        raise TypeError("cfar.os(): k must be an integer, got %r" % (k,))


def _gate_u8(threshold):
    """``img > threshold`` (feature_extraction.py:224) for uint8 pixels as the integer gate of
    sfe_cfar_u8: x > t  <=>  x > floor(t) for integer x; a negative threshold gates nothing (-1 = off),
    a threshold >= 255 gates everything."""
    t = _math.floor(float(threshold))
    return -1 if t < 0 else int(min(t, 255))


def _run(alg, img, train_hs, guard_hs, k, tau, want_thr, gate=None, ctx=None):
    """gate: None = plain cfar.*; else the intensity threshold of feature_extraction.py:224 as given
    (compared as is against float images, floored for uint8 ones)."""
    ctx = ctx or _L.default_context()
    img = _np.asarray(img)
    intensity_thr = -1 if gate is None else _gate_u8(gate)
    if img.ndim != 2:
        raise TypeError("cfar: expected a 2-D array, got shape %r" % (img.shape,))
    rows, cols = img.shape
    mask = _np.zeros((rows, cols), _np.uint8)
    thr = _np.zeros((rows, cols), _np.float32) if want_thr else None
    tp = _L.ptr(thr, _C.c_float) if want_thr else None
    train_hs, guard_hs = _as_index(train_hs), _as_index(guard_hs)
    with ctx.lock:
        if img.dtype == _np.uint8:
            a = _np.ascontiguousarray(img)
            rc = ctx.lib.sfe_cfar_u8(ctx.handle, _L.ptr(a, _C.c_uint8), rows, cols, alg, train_hs,
                                     guard_hs, k, float(tau), int(intensity_thr),
                                     _L.ptr(mask, _C.c_uint8), tp)
        else:
            a = _np.ascontiguousarray(img, _np.float32)
            rc = ctx.lib.sfe_cfar_f32(ctx.handle, _L.ptr(a, _C.c_float), rows, cols, alg, train_hs,
                                      guard_hs, k, float(tau), _L.ptr(mask, _C.c_uint8), tp)
            if gate is not None:
                mask &= (a > gate).astype(_np.uint8)
        ctx._check(rc)
    return (mask, thr) if want_thr else mask


def ca(img, train_hs, guard_hs, tau):
    return _run(0, img, train_hs, guard_hs, 0, tau, False)


def soca(img, train_hs, guard_hs, tau):
    return _run(1, img, train_hs, guard_hs, 0, tau, False)


def goca(img, train_hs, guard_hs, tau):
    return _run(2, img, train_hs, guard_hs, 0, tau, False)


def os(img, train_hs, guard_hs, k, tau):
    return _run(3, img, train_hs, guard_hs, _as_index(k), tau, False)


def ca2(img, train_hs, guard_hs, tau):
    return _run(0, img, train_hs, guard_hs, 0, tau, True)


def soca2(img, train_hs, guard_hs, tau):
    return _run(1, img, train_hs, guard_hs, 0, tau, True)


def goca2(img, train_hs, guard_hs, tau):
    return _run(2, img, train_hs, guard_hs, 0, tau, True)


def os2(img, train_hs, guard_hs, k, tau):
    return _run(3, img, train_hs, guard_hs, _as_index(k), tau, True)


def detect_gated(img, alg, params, threshold, ctx=None):
    """CFAR.detect(img, alg) & (img > threshold) in one kernel.

    ``params`` is the tuple CFAR.params[alg] (CFAR.py:35-40): (train_hs, guard_hs, tau) or
    (train_hs, guard_hs, rank, tau) for "OS".
    """
    code = _L.ALG[alg]
    if code == 3:
        train_hs, guard_hs, k, tau = params
        k = _as_index(k)
    else:
        train_hs, guard_hs, tau = params
        k = 0
    return _run(code, img, train_hs, guard_hs, k, tau, False, gate=threshold, ctx=ctx)
