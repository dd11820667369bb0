"""Threshold factors tau(Ntc, Pfa[, rank]) of the four CFAR variants under white Gaussian noise.

The detector compares ``cell > tau * noise_estimate`` (sfe_cfar.hip), so tau has to come out
of the same floating-point iteration the reference runs or masks differ on borderline cells.
What the reference does (bruce_slam/src/bruce_slam/CFAR.py:71-121): CA is closed form; SOCA,
GOCA and OS are roots of a P_fa residual found by scipy.optimize.root (MINPACK hybr) started
from the CA value scaled by ten log-spaced ratios, the first converged start wins.  This
module states those residuals as free functions of (x, ntc, pfa, rank); the class in
``CFAR.py`` binds them.  tests/golden/cfar_tau.json (generated from the reference) pins the
result bit for bit.
"""
from math import exp, lgamma

import numpy as np
from scipy.optimize import root

START_RATIOS = np.logspace(-2, 2, 10)


def _as_float(x):
    # MINPACK passes the unknown as a length-1 array
    return float(np.asarray(x, dtype=float).reshape(-1)[0])


def ca_factor(ntc, pfa):
    return ntc * (pfa ** (-1.0 / ntc) - 1)


def half_window_tail(x, ntc):
    """Sum_{k<N/2} C(N/2-1+k, k) (2+x/(N/2))^-(k+N/2): the term SOCA and GOCA share."""
    x = _as_float(x)
    h = ntc / 2
    base = 2 + x / h
    total = 0.0
    for k in range(int(h)):
        total += exp(lgamma(h + k) - lgamma(k + 1) - lgamma(h)) * base ** (-k)
    return total * base ** (-h)


def residual_soca(x, ntc, pfa):
    return half_window_tail(x, ntc) - pfa / 2


def residual_goca(x, ntc, pfa):
    x = _as_float(x)
    h = ntc / 2
    return (1.0 + x / h) ** (-h) - half_window_tail(x, ntc) - pfa / 2


def residual_os(x, ntc, pfa, rank):
    x = _as_float(x)
    return exp(lgamma(ntc + 1) - lgamma(ntc - rank + 1)
               + lgamma(x + ntc - rank + 1) - lgamma(x + ntc + 1)) - pfa


def first_root(residual, ntc, pfa, label):
    """Root of ``residual`` from the first of the ten scaled CA starts that converges."""
    x_ca = ca_factor(ntc, pfa)
    for s in START_RATIOS:
        sol = root(residual, x_ca * s)
        if sol.success:
            return sol.x[0]
    raise ValueError("Threshold factor of %s not found" % label)
