"""Threshold-factor arithmetic (sonar_slam_amd/wgn_threshold.py) on its own: the residuals vanish at the
solved factors, the closed form of CA inverts its own P_fa expression, and the class in CFAR.py binds the
same functions (bruce_slam/src/bruce_slam/CFAR.py:71-121; the golden file pins the values bit for bit)."""
import json
import os

import numpy as np
import pytest

from sonar_slam_amd import wgn_threshold as wgn

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("ntc,pfa", [(40, 0.1), (20, 0.05), (8, 1e-3), (64, 1e-2)])
def test_ca_factor_inverts_its_pfa(ntc, pfa):
    tau = wgn.ca_factor(ntc, pfa)
    assert tau > 0
    assert (1.0 + tau / ntc) ** (-ntc) == pytest.approx(pfa, rel=1e-12)


@pytest.mark.parametrize("ntc,pfa,rank", [(40, 0.1, 10), (20, 0.05, 15), (16, 1e-2, 8)])
def test_roots_zero_their_residuals_and_order_as_expected(ntc, pfa, rank):
    so = wgn.first_root(lambda x: wgn.residual_soca(x, ntc, pfa), ntc, pfa, "SOCA")
    go = wgn.first_root(lambda x: wgn.residual_goca(x, ntc, pfa), ntc, pfa, "GOCA")
    osf = wgn.first_root(lambda x: wgn.residual_os(x, ntc, pfa, rank), ntc, pfa, "OS")
    assert abs(wgn.residual_soca(so, ntc, pfa)) < 1e-9
    assert abs(wgn.residual_goca(go, ntc, pfa)) < 1e-9
    assert abs(wgn.residual_os(osf, ntc, pfa, rank)) < 1e-9
    ca = wgn.ca_factor(ntc, pfa)
    assert go < ca < so          # the smaller of two half-window means needs the larger factor
    assert osf > 0


def test_first_root_reports_failure_like_the_reference():
    with pytest.raises(ValueError, match="Threshold factor of X not found"):
        wgn.first_root(lambda x: 1.0 + float(np.asarray(x).reshape(-1)[0]) ** 2, 40, 0.1, "X")


def test_class_binds_the_module_and_matches_the_golden_file():
    from sonar_slam_amd.CFAR import CFAR
    golden = json.load(open(os.path.join(HERE, "golden", "cfar_tau.json")))
    cases = golden["cases"] if isinstance(golden, dict) and "cases" in golden else golden
    case = cases[0]
    det = CFAR(case["Ntc"], case["Ngc"], case["Pfa"], case.get("rank"))
    assert det.threshold_factor_CA == wgn.ca_factor(det.Ntc, det.Pfa)
    assert det.calc_WGN_pfa_SOCA(det.threshold_factor_SOCA) == wgn.residual_soca(det.threshold_factor_SOCA, det.Ntc, det.Pfa)
    assert det.calc_WGN_threshold_factor_GOCA() == det.threshold_factor_GOCA
