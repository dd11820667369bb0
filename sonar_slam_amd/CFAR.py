"""Host-side CFAR front end, mirror of the reference class ``bruce_slam.CFAR.CFAR``.

Same constructor, attributes (``threshold_factor_*``, ``params``, ``detector``,
``detector2``) and methods (``detect``, ``detect2``) as
bruce_slam/src/bruce_slam/CFAR.py:9-133, so FeatureExtraction code written against the
reference runs unchanged; the native module behind it is ``sonar_slam_amd.cfar`` (HIP).

The threshold factors (CFAR.py:71-121) are computed in ``wgn_threshold``; the ``calc_WGN_*``
methods are kept as thin bindings for code that calls them on the object.
"""
from . import cfar
from . import wgn_threshold as wgn


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
        self.threshold_factor_SOCA = self.calc_WGN_threshold_factor_SOCA()
        self.threshold_factor_GOCA = self.calc_WGN_threshold_factor_GOCA()
        self.threshold_factor_OS = self.calc_WGN_threshold_factor_OS()

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

    # ---- threshold factors: the arithmetic lives in wgn_threshold.py ----
    def calc_WGN_threshold_factor_CA(self):
        return wgn.ca_factor(self.Ntc, self.Pfa)

    def calc_WGN_pfa_GOSOCA_core(self, x):
        return wgn.half_window_tail(x, self.Ntc)

    def calc_WGN_pfa_SOCA(self, x):
        return wgn.residual_soca(x, self.Ntc, self.Pfa)

    def calc_WGN_pfa_GOCA(self, x):
        return wgn.residual_goca(x, self.Ntc, self.Pfa)

    def calc_WGN_pfa_OS(self, x):
        return wgn.residual_os(x, self.Ntc, self.Pfa, self.rank)

    def calc_WGN_threshold_factor_SOCA(self):
        return wgn.first_root(self.calc_WGN_pfa_SOCA, self.Ntc, self.Pfa, "SOCA")

    def calc_WGN_threshold_factor_GOCA(self):
        return wgn.first_root(self.calc_WGN_pfa_GOCA, self.Ntc, self.Pfa, "GOCA")

    def calc_WGN_threshold_factor_OS(self):
        return wgn.first_root(self.calc_WGN_pfa_OS, self.Ntc, self.Pfa, "OS")

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
