"""Host-side CFAR front end, mirror of the reference class ``bruce_slam.CFAR.CFAR``.

Same constructor, attributes (``threshold_factor_*``, ``params``, ``detector``,
``detector2``) and methods (``detect``, ``detect2``) as
bruce_slam/src/bruce_slam/CFAR.py:9-133, so FeatureExtraction code written against the
reference runs unchanged; the native module behind it is ``sonar_slam_amd.cfar`` (HIP).

The threshold factors are the white-Gaussian-noise P_fa inversions of CFAR.py:71-121: CA in
closed form, SOCA / GOCA / OS as roots (scipy.optimize.root from ten log-spaced starting
points, first converged one wins -- the iteration must be identical because tau enters the
detector's compare to the last bit).
"""
import math

import numpy as np
from scipy.optimize import root

from . import cfar


class CFAR(object):
    def __init__(self, Ntc, Ngc, Pfa, rank=None):
        if Ntc % 2 or Ngc % 2:  # CFAR.py:19-21
            raise AssertionError("Ntc and Ngc must be even")
        self.Ntc, self.Ngc, self.Pfa = Ntc, Ngc, Pfa
        if rank is None:
            self.rank = self.Ntc / 2  # CFAR.py:24 (a float: cfar.os will refuse it, as pybind does)
        else:
            self.rank = rank
            if not 0 <= self.rank < self.Ntc:
                raise AssertionError("rank out of range")

        self.threshold_factor_CA = self.calc_WGN_threshold_factor_CA()
        self.threshold_factor_SOCA = self._solve(self.calc_WGN_pfa_SOCA, "SOCA")
        self.threshold_factor_GOCA = self._solve(self.calc_WGN_pfa_GOCA, "GOCA")
        self.threshold_factor_OS = self._solve(self.calc_WGN_pfa_OS, "OS")

        hs, gs = self.Ntc // 2, self.Ngc // 2  # CFAR.py:35-40
        self.params = {
            "CA": (hs, gs, self.threshold_factor_CA),
            "SOCA": (hs, gs, self.threshold_factor_SOCA),
            "GOCA": (hs, gs, self.threshold_factor_GOCA),
            "OS": (hs, gs, self.rank, self.threshold_factor_OS),
        }
        self.detector = {"CA": cfar.ca, "SOCA": cfar.soca, "GOCA": cfar.goca, "OS": cfar.os}
        self.detector2 = {"CA": cfar.ca2, "SOCA": cfar.soca2, "GOCA": cfar.goca2, "OS": cfar.os2}

    def __str__(self):
        return ("CFAR Detector Information\n=========================\n"
                "Number of training cells: {}\nNumber of guard cells: {}\n"
                "Probability of false alarm: {}\nOrder statictics rank: {}\n"
                "Threshold factors:\n      CA-CFAR: {:.3f}\n    SOCA-CFAR: {:.3f}\n"
                "    GOCA-CFAR: {:.3f}\n    OSCA-CFAR: {:.3f}\n").format(
                    self.Ntc, self.Ngc, self.Pfa, self.rank, self.threshold_factor_CA,
                    self.threshold_factor_SOCA, self.threshold_factor_GOCA, self.threshold_factor_OS)

    # ---- threshold factors (CFAR.py:71-121) ----
    def calc_WGN_threshold_factor_CA(self):
        return self.Ntc * (self.Pfa ** (-1.0 / self.Ntc) - 1)

    def _solve(self, fun, name):
        x0 = self.calc_WGN_threshold_factor_CA()
        for ratio in np.logspace(-2, 2, 10):
            ret = root(fun, x0 * ratio)
            if ret.success:
                return ret.x[0]
        raise ValueError("Threshold factor of %s not found" % name)

    def calc_WGN_threshold_factor_SOCA(self):
        return self._solve(self.calc_WGN_pfa_SOCA, "SOCA")

    def calc_WGN_threshold_factor_GOCA(self):
        return self._solve(self.calc_WGN_pfa_GOCA, "GOCA")

    def calc_WGN_threshold_factor_OS(self):
        return self._solve(self.calc_WGN_pfa_OS, "OS")

    @staticmethod
    def _scalar(x):
        # scipy.optimize.root hands the residual a length-1 array (the reference does float(x))
        return float(np.asarray(x, dtype=float).reshape(-1)[0])

    def calc_WGN_pfa_GOSOCA_core(self, x):
        x = self._scalar(x)
        half = self.Ntc / 2
        acc = 0.0
        for k in range(int(half)):
            acc += math.exp(math.lgamma(half + k) - math.lgamma(k + 1) - math.lgamma(half)) \
                * (2 + x / half) ** (-k)
        return acc * (2 + x / half) ** (-half)

    def calc_WGN_pfa_SOCA(self, x):
        return self.calc_WGN_pfa_GOSOCA_core(x) - self.Pfa / 2

    def calc_WGN_pfa_GOCA(self, x):
        x = self._scalar(x)
        half = self.Ntc / 2
        return (1.0 + x / half) ** (-half) - self.calc_WGN_pfa_GOSOCA_core(x) - self.Pfa / 2

    def calc_WGN_pfa_OS(self, x):
        x = self._scalar(x)
        n, r = self.Ntc, self.rank
        return math.exp(math.lgamma(n + 1) - math.lgamma(n - r + 1)
                        + math.lgamma(x + n - r + 1) - math.lgamma(x + n + 1)) - self.Pfa

    # ---- detection (CFAR.py:123-133) ----
    def detect(self, mat, alg="CA"):
        """Target mask array."""
        return self.detector[alg](mat, *self.params[alg])

    def detect2(self, mat, alg="CA"):
        """Target mask array and threshold array."""
        return self.detector2[alg](mat, *self.params[alg])

    def detect_gated(self, mat, alg, threshold):
        """detect(mat, alg) & (mat > threshold) fused in one kernel (feature_extraction.py:223-224)."""
        return cfar.detect_gated(mat, alg, self.params[alg], threshold)
