"""CFAR pinned to the reference implementation ITSELF: bruce_slam/src/bruce_slam/cpp/cfar.cpp is
compiled unmodified, from where it lies under /root/reference, against a 40-line stand-in for the
two Eigen types it uses (oracle/ref_shim, recipe oracle/Makefile) into oracle/_ref/libcfar_ref.so.
The oracle's C restatement and the HIP kernels must reproduce it bit for bit."""
import numpy as np
import pytest

import oracle
from sonar_slam_amd import synth
from sonar_slam_amd.CFAR import CFAR

need_ref = pytest.mark.skipif(not oracle.have_ref_cfar(), reason="oracle/_ref/libcfar_ref.so not built "
                              "(needs /root/reference at build time)")
ALGS = ["CA", "SOCA", "GOCA", "OS"]


def _params(alg, det):
    p = det.params[alg]
    return (p[0], p[1], p[3], p[2]) if alg == "OS" else (p[0], p[1], p[2], 0)   # train_hs, guard_hs, tau, k


@need_ref
@pytest.mark.parametrize("alg", ALGS)
def test_oracle_restatement_equals_compiled_reference(alg):
    det = CFAR(40, 10, 0.1, 10)                       # feature.yaml:3-7
    th, gh, tau, k = _params(alg, det)
    rng = np.random.default_rng(11)
    frames = [synth.sonar_frame(seed=3, rows=300, cols=96, n_blobs=10),
              rng.integers(0, 256, (120, 40), dtype=np.uint8),
              np.full((90, 17), 255, np.uint8), np.zeros((64, 9), np.uint8),
              rng.integers(0, 256, (50, 12), dtype=np.uint8),      # shorter than the window: all-zero mask
              (rng.random((140, 33)) * 300).astype(np.float32)]    # non-uint8 caller
    for img in frames:
        want, want_thr = oracle.ref_cfar(img, alg, th, gh, tau, k, want_threshold=True)
        got, got_thr = oracle.cfar(img, alg, th, gh, tau, k, want_threshold=True)
        assert np.array_equal(got, want) and np.array_equal(got_thr, want_thr)
        assert np.array_equal(oracle.cfar(img, alg, th, gh, tau, k), oracle.ref_cfar(img, alg, th, gh, tau, k))
    # other windows / taus
    for (t, g, ta) in ((4, 1, 1.3), (10, 0, 2.0), (7, 3, 0.9)):
        img = rng.integers(0, 256, (80, 20), dtype=np.uint8)
        kk = min(k, 2 * t - 1)
        assert np.array_equal(oracle.cfar(img, alg, t, g, ta, kk), oracle.ref_cfar(img, alg, t, g, ta, kk))


@need_ref
@pytest.mark.gpu
@pytest.mark.parametrize("alg", ALGS)
def test_hip_cfar_equals_compiled_reference_full_size(alg):
    """the HIP kernels against the reference's own code on the BASELINE frame shape"""
    from sonar_slam_amd import cfar
    det = CFAR(40, 10, 0.1, 10)
    th, gh, tau, k = _params(alg, det)
    img = synth.sonar_frame(seed=17)                  # 1024 x 512
    want = oracle.ref_cfar(img, alg, th, gh, tau, k)
    fn = getattr(cfar, alg.lower())
    got = fn(img, th, gh, k, tau) if alg == "OS" else fn(img, th, gh, tau)
    assert np.array_equal(np.asarray(got), want)
    fn2 = getattr(cfar, alg.lower() + "2")
    want2 = oracle.ref_cfar(img, alg, th, gh, tau, k, want_threshold=True)
    got2 = fn2(img, th, gh, k, tau) if alg == "OS" else fn2(img, th, gh, tau)
    assert np.array_equal(np.asarray(got2[0]), want2[0]) and np.array_equal(np.asarray(got2[1]), want2[1])
