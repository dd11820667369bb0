"""The threshold maps of the *2 CFAR variants are COMPUTED per pixel on the device (cfar_thr_arith, sfe_cfar.hip) instead
of gathered from a table: thr = (float)(tau * (double)(float)s / D), D = T or 2T (cfar.cpp:27,46,67), with the quotient
formed from the reciprocal of D and two fma residual corrections.  The library checks that sequence against the reference
expression for every window sum of a launch before it uses it; this test restates both on the CPU (libm's fma) over the
shipped windows and a few unfriendly tau values, so that a compiler or refactoring slip in the sequence shows up without
a GPU."""
import ctypes
import ctypes.util

import numpy as np
import pytest

_libm = ctypes.CDLL(ctypes.util.find_library("m"))
_libm.fma.restype = ctypes.c_double
_libm.fma.argtypes = [ctypes.c_double] * 3


def _arith(tau, d, s):
    p = tau * float(np.float32(s))
    rinv = 1.0 / d
    q = p * rinv
    e = _libm.fma(-d, q, p)
    q = _libm.fma(e, rinv, q)
    e = _libm.fma(-d, q, p)
    q = _libm.fma(e, rinv, q)
    return np.float32(q)


@pytest.mark.parametrize("tau", [9.137608674642355, 3.0, 0.1, 1.0 / 3.0, 7.123456789e5, 1e-300])
@pytest.mark.parametrize("train", [40, 32, 20, 16, 24, 60])
def test_computed_threshold_equals_the_reference_expression(tau, train):
    T = train // 2
    rng = np.random.default_rng(train)
    for d, smax in ((float(T), 255 * T), (2.0 * T, 255 * 2 * T)):      # SOCA / GOCA: one window, CA: both
        sums = np.unique(np.r_[0, 1, smax, rng.integers(0, smax + 1, 600)])
        for s in sums:
            want = np.float32(tau * float(np.float32(s)) / d)
            assert _arith(tau, d, int(s)).tobytes() == want.tobytes(), (tau, d, int(s))
