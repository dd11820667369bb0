"""CPU: the oracle's remap / nonzero / px->m / match / ICP restatements against independent
numpy computations and analytic properties."""
import numpy as np
import pytest

import oracle
from sonar_slam_amd import synth
from sonar_slam_amd.feature_extraction import build_maps, oculus_bearings


def test_bilinear_table_matches_opencv_construction():
    t = oracle.bilinear_tab().astype(np.int64)
    assert (t.sum(1) == 32768).all()                 # OpenCV's fix-up guarantees the sum
    assert list(t[0]) == [32767, 0, 0, 1]            # saturate_cast<short>(32768) + fix-up
    i, j = 5, 20
    assert list(t[i * 32 + j]) == [(32 - i) * (32 - j) * 32, (32 - i) * j * 32, i * (32 - j) * 32, i * j * 32]


def _remap_np(src, mx, my):
    """independent float64->fixed-point restatement with numpy (no table)"""
    sx = np.rint(mx.astype(np.float32) * np.float32(32)).astype(np.int64)
    sy = np.rint(my.astype(np.float32) * np.float32(32)).astype(np.int64)
    ix, iy, fx, fy = sx >> 5, sy >> 5, sx & 31, sy & 31
    pad = np.zeros((src.shape[0] + 2, src.shape[1] + 2), np.int64)
    pad[1:-1, 1:-1] = src
    def tap(yy, xx):
        ok = (yy >= -1) & (yy <= src.shape[0]) & (xx >= -1) & (xx <= src.shape[1])
        return np.where(ok, pad[np.clip(yy + 1, 0, pad.shape[0] - 1), np.clip(xx + 1, 0, pad.shape[1] - 1)], 0)
    w00, w01, w10, w11 = (32 - fy) * (32 - fx) * 32, (32 - fy) * fx * 32, fy * (32 - fx) * 32, fy * fx * 32
    z = (fx == 0) & (fy == 0)
    w00 = np.where(z, 32767, w00)
    w11 = np.where(z, 1, w11)
    acc = w00 * tap(iy, ix) + w01 * tap(iy, ix + 1) + w10 * tap(iy + 1, ix) + w11 * tap(iy + 1, ix + 1)
    return ((acc + 16384) >> 15).astype(np.uint8)


def test_remap_oracle_vs_numpy():
    rng = np.random.default_rng(0)
    res, h, rows, w, cols, mx, my = build_maps(oculus_bearings(64), 0.25, 96)
    img = rng.integers(0, 256, (96, 64), dtype=np.uint8)
    assert np.array_equal(oracle.remap_u8(img, mx, my), _remap_np(img, mx, my))
    mask = (rng.random((96, 64)) < 0.05).astype(np.uint8)
    out = oracle.remap_u8(mask, mx, my)
    assert np.array_equal(out, _remap_np(mask, mx, my))
    assert set(np.unique(out)) <= {0, 1}
    # identity map with integer coordinates reproduces the image (table entry {32767,0,0,1})
    yy, xx = np.mgrid[0:96, 0:64].astype(np.float32)
    assert np.array_equal(oracle.remap_u8(img, xx, yy), img)
    # half-pixel shift: rounding (a+b+1)>>1 of fixed point
    half = oracle.remap_u8(img, xx + np.float32(0.5), yy)
    exp = ((img[:, :-1].astype(int) * 16384 + img[:, 1:].astype(int) * 16384 + 16384) >> 15)
    assert np.array_equal(half[:, :-1], exp)


def test_nonzero_and_px_to_m():
    rng = np.random.default_rng(1)
    img = (rng.random((40, 33)) < 0.1).astype(np.uint8)
    rc = oracle.nonzero(img)
    assert np.array_equal(rc, np.c_[np.nonzero(img)])
    rows, cols, width, height = 40, 33, 12.3, 9.87
    pts = oracle.px_to_m(rc, rows, cols, width, height)
    x = rc[:, 1] - cols / 2.
    x = (-1 * ((x / float(cols / 2.)) * (width / 2.)))
    y = (-1 * (rc[:, 0] / float(rows)) * height) + height
    assert np.array_equal(pts, np.column_stack((y, x)))


def test_match_against_numpy():
    rng = np.random.default_rng(2)
    ref = rng.uniform(-10, 10, (300, 2)).astype(np.float32)
    q = rng.uniform(-12, 12, (200, 2)).astype(np.float32)
    ids, d2 = oracle.match(ref, q, 0.5)
    d = ((q[:, None, :] - ref[None, :, :]) ** 2)
    dd = (d[..., 0] + d[..., 1]).astype(np.float32)
    j = dd.argmin(1)
    best = dd[np.arange(len(q)), j]
    ok = best <= np.float32(0.25)
    assert np.array_equal(ids[0], np.where(ok, j, -1))
    assert np.array_equal(d2[0][ok], best[ok]) and np.isinf(d2[0][~ok]).all()
    assert ids.shape == (1, 200) and d2.dtype == np.float32


def test_match_tie_goes_to_lowest_index():
    ref = np.array([[1, 0], [-1, 0], [1, 0]], np.float32)
    ids, _ = oracle.match(ref, np.zeros((1, 2), np.float32), 5.0)
    assert ids[0, 0] == 0


@pytest.mark.parametrize("minimizer", [0, 1])
def test_icp_recovers_motion(minimizer):
    # the shipped differential checker (0.1 m / 0.01 rad over 4 steps) stops after ~5 iterations
    # by design; convergence of the maths is checked with the counter checker alone
    src, tgt, guess, truth = synth.scan_pair(seed=6, n_src=1500, n_tgt=1500, outliers=0.1)
    for prec in (0, 1):
        st, T, it = oracle.icp(src, tgt, guess, oracle.shipped_icp_params(
            minimizer=minimizer, precision=prec, use_diff_checker=0, max_iter=30))
        assert st == 0 and it == 30
        x, y, th = synth.pose_of(T)
        tx, ty, tth = synth.pose_of(truth)
        assert abs(x - tx) < 0.03 and abs(y - ty) < 0.03 and abs(th - tth) < 0.005


def test_icp_float_and_double_accumulation_agree():
    src, tgt, guess, _ = synth.scan_pair(seed=9, n_src=800, n_tgt=700)
    _, Tf, itf = oracle.icp(src, tgt, guess, oracle.shipped_icp_params(precision=0))
    _, Td, itd = oracle.icp(src, tgt, guess, oracle.shipped_icp_params(precision=1))
    assert itf == itd
    pf, pd = synth.pose_of(Tf), synth.pose_of(Td)
    assert max(abs(a - b) for a, b in zip(pf, pd)) < 1e-4


def test_icp_stop_rule_and_failures():
    src, tgt, guess, _ = synth.scan_pair(seed=3, n_src=300, n_tgt=300)
    # counter checker alone: exactly max_iter iterations
    st, _, it = oracle.icp(src, tgt, guess, oracle.shipped_icp_params(use_diff_checker=0, max_iter=7))
    assert st == 0 and it == 7
    # differential checker cannot fire before smoothLength iterations
    st, _, it = oracle.icp(src, tgt, guess, oracle.shipped_icp_params())
    assert it >= 4
    # nothing within the matcher radius: "no outlier to filter", T = guess (pcl.cpp:207-210)
    far = tgt + np.float32(1000.0)
    st, T, it = oracle.icp(src, far, guess, oracle.shipped_icp_params())
    assert st == 1 and np.array_equal(T, guess)
    # without the trimmed filter the same case ends in "no point to minimize"
    st, T, _ = oracle.icp(src, far, guess, oracle.shipped_icp_params(use_trimmed_filter=0))
    assert st == 2 and np.array_equal(T, guess)


def test_icp_identity_on_identical_clouds():
    src, tgt, _, _ = synth.scan_pair(seed=4, n_src=400, n_tgt=400, outliers=0.0)
    st, T, it = oracle.icp(tgt, tgt, np.eye(3, dtype=np.float32), oracle.shipped_icp_params())
    assert st == 0
    assert np.allclose(T, np.eye(3), atol=2e-6)


def test_remove_outlier_semantics():
    pts = np.array([[0, 0], [0.1, 0], [0, 0.1], [5, 5], [0.1, 0.1]], np.float32)
    out = oracle.remove_outlier(pts, 1.0, 3)   # needs > 3 points in radius counting itself
    assert np.array_equal(out, pts[[0, 1, 2, 4]])
    assert len(oracle.remove_outlier(pts, 1.0, 4)) == 0


def test_downsample_oracle_properties():
    rng = np.random.default_rng(11)
    pts = rng.uniform(-10, 10, (2500, 2)).astype(np.float32)
    out, idx = oracle.downsample(pts, 0.5, return_index=True)
    assert np.array_equal(out, pts[idx]) and len(set(idx.tolist())) == len(idx)
    # every input point lies in the same <= 0.5 m cell as some output point
    assert 0.2 * len(pts) < len(out) < len(pts)
    d = np.sqrt(((pts[:, None, :] - out[None, :, :]) ** 2).sum(-1)).min(1)
    assert d.max() <= 0.5 * np.sqrt(2) + 1e-5
    # a single point / identical points collapse to one
    assert len(oracle.downsample(pts[:1], 0.5)) == 1
    assert len(oracle.downsample(np.repeat(pts[:1], 9, axis=0), 0.5)) == 1
    # coarse resolution >= bounding box: one medoid for the whole cloud
    one = oracle.downsample(pts, 100.0)
    c = pts.mean(0)
    assert len(one) == 1 and np.allclose(one[0], pts[np.argmin(((pts - c) ** 2).sum(1))])


def test_kdtree_backend_returns_the_same_neighbours_and_poses():
    """the CPU-baseline accelerator (exact kd-tree) must not change a single decision"""
    import time
    src, tgt, guess, _ = synth.scan_pair(seed=12, n_src=1500, n_tgt=1700)
    tgt = np.concatenate([tgt, tgt[:200]]).astype(np.float32)      # exact duplicates: ties -> lowest index
    out = {}
    for on in (0, 1):
        oracle.set_kdtree(on)
        try:
            t0 = time.perf_counter()
            ids, d2 = oracle.match(tgt, src, 0.5)
            nrm = oracle.normals2d(tgt - tgt.mean(0).astype(np.float32), 10)
            res = [oracle.icp(src, tgt, guess, oracle.shipped_icp_params(minimizer=mz, precision=0,
                                                                         use_diff_checker=1 - mz, max_iter=12))
                   for mz in (0, 1)]
            out[on] = (ids, d2, nrm, res, time.perf_counter() - t0)
        finally:
            oracle.set_kdtree(0)
    a, b = out[0], out[1]
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    for (sa, Ta, ia), (sb, Tb, ib) in zip(a[3], b[3]):
        assert sa == sb and ia == ib and np.array_equal(Ta, Tb)
    assert b[4] < a[4]


def test_jet_colour_map_table():
    """cv2.applyColorMap(img, cv2.COLORMAP_JET) (feature_extraction.py:227) as a 256 x 3 BGR table: the library's
    integer evaluation equals the oracle's rational one; anchor values everybody knows from OpenCV's JET
    (0 -> (128, 0, 0), 255 -> (0, 0, 128), mid-grey green-ish with full G); a float32 emulation of OpenCV's own pipeline
    moves ONE of the 768 entries by one grey level (the table's ramp values are all exact ties): parity unpinned."""
    from sonar_slam_amd.feature_extraction import colormap_lut
    lut = colormap_lut()
    want = oracle.colormap_jet_lut()
    assert lut.shape == (256, 3) and np.array_equal(lut, want)
    assert lut[0].tolist() == [128, 0, 0] and lut[255].tolist() == [0, 0, 128]
    assert lut[1].tolist() == [132, 0, 0] and lut[128, 1] == 255 and lut[64].tolist() == [255, 128, 0]
    emu = oracle.colormap_jet_lut_float_emulation()
    diff = np.argwhere(want != emu)
    assert len(diff) <= 1 and np.abs(want.astype(int) - emu.astype(int)).max() <= 1
    with pytest.raises(ValueError):
        colormap_lut(4)
