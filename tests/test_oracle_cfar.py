"""CPU: the C oracle's CFAR against an independent numpy restatement of cfar.cpp and against
hand-computable cases.  (The reference ships no golden vectors: SURVEY 4 / 8c.)"""
import numpy as np
import pytest

import oracle
from np_ref import cfar_np

ALGS = ["CA", "SOCA", "GOCA", "OS"]


@pytest.mark.parametrize("alg", ALGS)
@pytest.mark.parametrize("shape", [(120, 37), (64, 64), (51, 8), (50, 8), (7, 5)])
def test_oracle_matches_numpy_u8(alg, shape, shipped_cfar):
    rng = np.random.default_rng(hash((alg, shape)) % 2**32)
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    p = shipped_cfar.params[alg]
    k = p[2] if alg == "OS" else 0
    m, t = oracle.cfar(img, alg, p[0], p[1], p[-1], k=k, want_threshold=True)
    mr, tr = cfar_np(img, alg, p[0], p[1], p[-1], k=k)
    assert np.array_equal(m, mr)
    assert np.array_equal(t, tr)


@pytest.mark.parametrize("alg", ALGS)
def test_oracle_matches_numpy_float_images(alg):
    rng = np.random.default_rng(3)
    img = rng.gamma(2.0, 11.3, (90, 21)).astype(np.float32)
    m, t = oracle.cfar(img, alg, 6, 2, 1.7, k=4, want_threshold=True)
    mr, tr = cfar_np(img, alg, 6, 2, 1.7, k=4)
    assert np.array_equal(m, mr)
    assert np.array_equal(t, tr)


def test_border_rows_stay_zero_and_short_images(shipped_cfar):
    p = shipped_cfar.params["SOCA"]
    img = np.full((100, 16), 200, np.uint8)
    img[::7] = 3
    m = oracle.cfar(img, "SOCA", *p)
    assert not m[:25].any() and not m[-25:].any()
    # fewer than 2*(train+guard)+1 rows: no valid row at all (cfar.cpp:16 loop bounds)
    assert not oracle.cfar(np.full((50, 16), 255, np.uint8), "SOCA", *p).any()


def test_constant_and_step_images(shipped_cfar):
    th, gh, tau = shipped_cfar.params["SOCA"]
    # constant image: x > tau * x (tau > 1) never fires; all-zero: 0 > 0 never fires
    assert not oracle.cfar(np.full((120, 8), 255, np.uint8), "SOCA", th, gh, tau).any()
    assert not oracle.cfar(np.zeros((120, 8), np.uint8), "SOCA", th, gh, tau).any()
    # a single bright row on a dark floor fires exactly on that row (training cells are all floor)
    img = np.full((120, 8), 10, np.uint8)
    img[60] = 200
    m = oracle.cfar(img, "SOCA", th, gh, tau)
    assert m[60].all() and m.sum() == 8
    # SOCA takes the smaller side: bright block just outside the guard on one side only still fires
    img2 = np.full((120, 8), 10, np.uint8)
    img2[60] = 60
    img2[66:86] = 250  # whole lagging window bright
    assert oracle.cfar(img2, "SOCA", th, gh, tau)[60].all()
    assert not oracle.cfar(img2, "GOCA", th, gh, shipped_cfar.params["GOCA"][2])[60].any()


def test_os_rank_semantics():
    # training cells of row 30 (train_hs=2, guard_hs=1): rows 27,28 and 32,33
    img = np.zeros((61, 1), np.uint8)
    img[[27, 28, 32, 33], 0] = [5, 9, 1, 7]
    img[30, 0] = 8
    for k, kth in enumerate([1, 5, 7, 9]):
        m, t = oracle.cfar(img, "OS", 2, 1, 1.0, k=k, want_threshold=True)
        assert t[30, 0] == kth
        assert m[30, 0] == (8 > kth)


def test_gate():
    img = np.array([[10, 70, 66], [65, 64, 255]], np.uint8)
    mask = np.ones_like(img)
    assert np.array_equal(oracle.gate(img, mask, 65), (img > 65).astype(np.uint8))
