"""GPU parity: polar->Cartesian remap, nonzero compaction and px->m vs the oracle, bit-exact."""
import numpy as np
import pytest

import oracle
from sonar_slam_amd import synth
from sonar_slam_amd.feature_extraction import FeatureExtraction, Geometry, SonarPing, build_maps, oculus_bearings

pytestmark = pytest.mark.gpu


def _geom(ctx, beams, ranges, res):
    r, height, rows, width, cols, mx, my = build_maps(oculus_bearings(beams), res, ranges)
    return Geometry(ctx, mx, my, (ranges, beams), width, height), mx, my, width, height


@pytest.mark.parametrize("beams,ranges,res", [(64, 96, 0.25), (512, 1024, 30.0 / 1024), (100, 77, 0.1)])
def test_remap_matches_oracle(ctx, beams, ranges, res):
    g, mx, my, _, _ = _geom(ctx, beams, ranges, res)
    rng = np.random.default_rng(beams)
    img = rng.integers(0, 256, (ranges, beams), dtype=np.uint8)
    assert np.array_equal(g.remap(img), oracle.remap_u8(img, mx, my))
    mask = (rng.random((ranges, beams)) < 0.03).astype(np.uint8)
    assert np.array_equal(g.remap(mask), oracle.remap_u8(mask, mx, my))


@pytest.mark.parametrize("beams,ranges,res", [(64, 96, 0.25), (512, 1024, 30.0 / 1024), (1024, 2048, 30.0 / 2048)])
def test_extract_points_matches_oracle(ctx, beams, ranges, res):
    g, mx, my, width, height = _geom(ctx, beams, ranges, res)
    rng = np.random.default_rng(ranges)
    for density in (0.0, 0.002, 0.05, 1.0):
        mask = (rng.random((ranges, beams)) < density).astype(np.uint8)
        locs, pts = g.extract(mask, cap=4096)      # small cap exercises the grow-and-retry path
        cart = oracle.remap_u8(mask, mx, my)
        want_rc = oracle.nonzero(cart)
        assert np.array_equal(locs, want_rc), density
        want_pts = oracle.px_to_m(want_rc, ranges, mx.shape[1], width, height)
        assert pts.dtype == np.float64 and np.array_equal(pts, want_pts), density


def test_mask_values_other_than_one(ctx):
    g, mx, my, width, height = _geom(ctx, 64, 96, 0.25)
    rng = np.random.default_rng(3)
    mask = (rng.random((96, 64)) < 0.05).astype(np.uint8) * 255
    locs, _ = g.extract(mask)
    assert np.array_equal(locs, oracle.nonzero(oracle.remap_u8(mask, mx, my)))


def test_feature_extraction_stages_on_synthetic_ping(ctx, shipped_cfar):
    """The ROS-free FeatureExtraction mirror, stage by stage, against the oracle chain
    (feature_extraction.py:223-238)."""
    fe = FeatureExtraction(ctx)
    fe.Ntc, fe.Ngc, fe.Pfa, fe.rank, fe.alg, fe.threshold = 40, 10, 0.1, 10, "SOCA", 65
    fe.configure()
    img = synth.sonar_frame(seed=42)
    ping = SonarPing(img, oculus_bearings(512), 30.0 / 1024, ping_id=0)
    fe.generate_map_xy(ping)
    peaks = fe.detect(img)
    th, gh, tau = shipped_cfar.params["SOCA"]
    want_peaks = oracle.gate(img, oracle.cfar(img, "SOCA", th, gh, tau), 65)
    assert np.array_equal(peaks, want_peaks)
    locs, pts = fe.extract(peaks)
    want_rc = oracle.nonzero(oracle.remap_u8(want_peaks, fe.map_x, fe.map_y))
    assert len(want_rc) > 100
    assert np.array_equal(locs, want_rc)
    assert np.array_equal(pts, oracle.px_to_m(want_rc, fe.rows, fe.cols, fe.width, fe.height))
    # geometry cache: same ping geometry does not rebuild the maps (feature_extraction.py:150-151)
    g0 = fe.geometry
    fe.generate_map_xy(ping)
    assert fe.geometry is g0


def test_batched_extract_device_path(ctx, shipped_cfar):
    n, ranges, beams = 5, 1024, 512
    g, mx, my, width, height = _geom(ctx, beams, ranges, 30.0 / 1024)
    th, gh, tau = shipped_cfar.params["SOCA"]
    frames = np.stack([synth.sonar_frame(seed=200 + s) for s in range(n)])
    masks = np.stack([oracle.gate(f, oracle.cfar(f, "SOCA", th, gh, tau), 65) for f in frames])
    cap = 20000
    d_mask, d_pts, d_cnt = ctx.alloc(masks.nbytes), ctx.alloc(n * cap * 16), ctx.alloc(n * 4)
    d_mask.upload(masks)
    ctx._check(ctx.lib.sfe_extract_points_batch_dev(ctx.handle, g.handle, d_mask.ptr, n, cap, d_pts.ptr, d_cnt.ptr))
    ctx.sync()
    cnt = d_cnt.download(np.int32, n)
    pts = d_pts.download(np.float64, n * cap * 2).reshape(n, cap, 2)
    for f in range(n):
        rc = oracle.nonzero(oracle.remap_u8(masks[f], mx, my))
        assert cnt[f] == len(rc)
        assert np.array_equal(pts[f, :cnt[f]], oracle.px_to_m(rc, ranges, mx.shape[1], width, height))
