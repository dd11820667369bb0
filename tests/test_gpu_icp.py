"""GPU parity: pcl.match / pcl.remove_outlier / pcl.ICP through the C ABI vs the CPU oracle.

Tolerances.  Index work (match ids, outlier decisions) is bit-exact.  ICP poses: the HIP path
accumulates in fp64 and rounds each iteration's transform to float; against the oracle's
fp64-accumulation mode it must agree to 1e-6 (same discrete decisions, same float transforms up
to one ulp of a sum); against the oracle's float mode (libpointmatcher's own precision) the bar
is BASELINE.json's 1e-4 m / 1e-4 rad."""
import numpy as np
import pytest

import oracle
from sonar_slam_amd import icp_config, pcl, synth

pytestmark = pytest.mark.gpu

TOL_TIGHT = 1e-6
TOL_REF = 1e-4


@pytest.fixture(autouse=True, params=["tiers", "one-size"])
def job_tiers(request, monkeypatch):
    """Every test of this module runs twice: with the launcher's job classes (the exhaustive one-wave kernel for clouds
    of a few hundred points, one-wave / four-wave workgroups of the strip sweep up to ~2000 points, 1024 threads beyond;
    large many-to-one jobs split over several workgroups) and with every job on the 1024-thread sweep kernels, unsplit
    (what round 2 shipped).  The knobs are read
    per call (sfe_icp_sweep_launch)."""
    if request.param == "one-size":
        monkeypatch.setenv("SFE_SW_TIERS", "0")
        monkeypatch.setenv("SFE_SW_MULTI", "0")
        monkeypatch.setenv("SFE_SW_TINY", "0")
    return request.param


def _pose_diff(Ta, Tb):
    a, b = synth.pose_of(Ta), synth.pose_of(Tb)
    return max(abs(a[0] - b[0]), abs(a[1] - b[1]), abs(np.arctan2(np.sin(a[2] - b[2]), np.cos(a[2] - b[2]))))


def _icp(params, ctx=None):
    icp = pcl.ICP(ctx)
    icp.setParams(params)
    return icp


def test_match_bit_exact():
    rng = np.random.default_rng(1)
    ref = rng.uniform(-15, 15, (3000, 2)).astype(np.float32)
    q = rng.uniform(-16, 16, (2500, 2)).astype(np.float32)
    for md in (0.5, 3.0, 100.0):
        ids, d2 = pcl.match(ref, q, 1, md)
        oi, od = oracle.match(ref, q, md)
        assert ids.shape == (1, 2500) and ids.dtype == np.int32 and d2.dtype == np.float32
        assert np.array_equal(ids, oi) and np.array_equal(d2, od)
    # ties and duplicates -> lowest index; empty query
    dup = np.array([[1, 0], [-1, 0], [1, 0]], np.float32)
    assert pcl.match(dup, np.zeros((1, 2), np.float32), 1, 5.0)[0][0, 0] == 0
    ids, d2 = pcl.match(ref, np.zeros((0, 2), np.float32), 1, 1.0)
    assert ids.shape == (1, 0)


def test_remove_outlier_bit_exact():
    rng = np.random.default_rng(2)
    pts = np.r_[rng.normal(0, 0.6, (400, 2)), rng.uniform(-20, 20, (300, 2))].astype(np.float32)
    for radius, mp in ((1.0, 5), (0.5, 2), (2.0, 40)):
        assert np.array_equal(pcl.remove_outlier(pts, radius, mp), oracle.remove_outlier(pts, radius, mp))
    assert pcl.remove_outlier(np.zeros((0, 2), np.float32), 1.0, 5).shape == (0, 2)


@pytest.mark.parametrize("seed,n", [(0, 300), (1, 1000), (2, 2500), (3, 5000)])
def test_icp_reference_chain_point_to_point(seed, n):
    src, tgt, guess, _ = synth.scan_pair(seed=seed, n_src=n, n_tgt=n - 37)
    icp = _icp(icp_config.shipped_params())
    msg, T = icp.compute(src, tgt, guess)
    assert msg == "success" and T.shape == (3, 3) and T.dtype == np.float32
    st, Td, itd = oracle.icp(src, tgt, guess, oracle.shipped_icp_params(precision=1))
    st, Tf, itf = oracle.icp(src, tgt, guess, oracle.shipped_icp_params(precision=0))
    _, _, it = icp.compute_batch(src, tgt, [guess])
    assert it[0] == itd
    assert _pose_diff(T, Td) < TOL_TIGHT
    assert _pose_diff(T, Tf) < TOL_REF


@pytest.mark.parametrize("seed", [4, 5])
def test_icp_point_to_plane_30_iterations(seed):
    """BASELINE config 2: fixed 30-iteration 2-D point-to-plane."""
    src, tgt, guess, truth = synth.scan_pair(seed=seed, n_src=5000, n_tgt=5000)
    p = icp_config.shipped_params(minimizer=1, use_diff_checker=0, max_iter=30)
    msgs, T, it = _icp(p).compute_batch(src, tgt, [guess])
    assert msgs[0] == "success" and it[0] == 30
    st, To, ito = oracle.icp(src, tgt, guess, oracle.shipped_icp_params(minimizer=1, use_diff_checker=0, max_iter=30))
    assert st == 0 and ito == 30
    assert _pose_diff(T[0], To) < TOL_REF
    assert _pose_diff(T[0], truth) < 0.02


def test_normals_and_large_target_streaming():
    """target larger than the LDS-resident tile (8192 points): streamed tiles must give the
    same matches; also the 20k-point high-res config shape."""
    src, tgt, guess, _ = synth.scan_pair(seed=8, n_src=3000, n_tgt=9000)
    for mz in (0, 1):
        p = icp_config.shipped_params(minimizer=mz, max_iter=6, use_diff_checker=0)
        msgs, T, it = _icp(p).compute_batch(src, tgt, [guess])
        st, To, _ = oracle.icp(src, tgt, guess, oracle.shipped_icp_params(minimizer=mz, max_iter=6, use_diff_checker=0))
        assert msgs[0] == "success" and st == 0
        assert _pose_diff(T[0], To) < (TOL_TIGHT if mz == 0 else TOL_REF)


def test_many_guesses_one_pair():
    """the NSSM loop of compute_icp_with_cov (slam.py:346-358) as one launch"""
    src, tgt, guess, _ = synth.scan_pair(seed=6, n_src=1200, n_tgt=1200)
    rng = np.random.default_rng(0)
    base = synth.pose_of(guess)
    guesses = [synth.pose_matrix(base[0] + dx, base[1] + dy, base[2] + dt).astype(np.float32)
               for dx, dy, dt in rng.normal(0, [0.3, 0.3, 0.05], (30, 3))]
    icp = _icp(icp_config.shipped_params())
    msgs, T, it = icp.compute_batch(src, tgt, guesses)
    assert len(msgs) == 30
    for g, m, Tg, i in zip(guesses, msgs, T, it):
        st, To, ito = oracle.icp(src, tgt, g, oracle.shipped_icp_params(precision=1))
        assert (m == "success") == (st == 0)
        assert i == ito
        assert _pose_diff(Tg, To) < TOL_TIGHT
    # and the single-call API returns the same thing
    m0, T0 = icp.compute(src, tgt, guesses[0])
    assert m0 == msgs[0] and np.array_equal(T0, T[0])


def test_failures_return_the_guess():
    src, tgt, guess, _ = synth.scan_pair(seed=7, n_src=400, n_tgt=400)
    far = tgt + np.float32(1000.0)
    msg, T = _icp(icp_config.shipped_params()).compute(src, far, guess)
    assert msg == "no outlier to filter" and np.array_equal(T, guess)
    msg, T = _icp(icp_config.shipped_params(use_trimmed_filter=0)).compute(src, far, guess)
    assert msg == "ErrorMnimizer: no point to minimize" and np.array_equal(T, guess)


def test_stop_rules():
    src, tgt, guess, _ = synth.scan_pair(seed=3, n_src=700, n_tgt=700)
    _, _, it = _icp(icp_config.shipped_params(use_diff_checker=0, max_iter=7)).compute_batch(src, tgt, [guess])
    assert it[0] == 7
    _, _, it = _icp(icp_config.shipped_params()).compute_batch(src, tgt, [guess])
    _, _, ito = oracle.icp(src, tgt, guess, oracle.shipped_icp_params())
    assert it[0] == ito and it[0] >= 4


def test_loads_shipped_yaml_text(tmp_path):
    from test_host import SHIPPED_ICP_YAML
    f = tmp_path / "icp.yaml"
    f.write_text(SHIPPED_ICP_YAML)
    icp = pcl.ICP()
    icp.loadFromYaml(str(f))
    assert icp.params.as_dict() == icp_config.shipped_params().as_dict()
    src, tgt, guess, _ = synth.scan_pair(seed=12, n_src=500, n_tgt=500)
    msg, T = icp.compute(src, tgt, guess)
    st, To, _ = oracle.icp(src, tgt, guess, oracle.shipped_icp_params(precision=1))
    assert msg == "success" and _pose_diff(T, To) < TOL_TIGHT


def test_tiny_and_degenerate_clouds():
    one = np.array([[1.0, 2.0]], np.float32)
    msg, T = _icp(icp_config.shipped_params()).compute(one, one, np.eye(3, dtype=np.float32))
    st, To, _ = oracle.icp(one, one, np.eye(3, dtype=np.float32), oracle.shipped_icp_params(precision=1))
    assert (msg == "success") == (st == 0)
    assert np.allclose(T, To, atol=1e-6)
    with pytest.raises(RuntimeError):
        _icp(icp_config.shipped_params()).compute(np.zeros((0, 2), np.float32), one, np.eye(3))


def test_duplicated_targets_exercise_the_exact_fallbacks():
    """many coincident target points: the approximate NN filter cannot separate their chunks, so
    the per-query cooperative scan (moderate duplication) and the whole-pass exact scan (queue
    overflow, heavy duplication) must take over and still match the oracle bit for bit"""
    src, tgt, guess, _ = synth.scan_pair(seed=21, n_src=1500, n_tgt=1600)
    rng = np.random.default_rng(0)
    for n_unique in (1200, 40):
        t = tgt[rng.integers(0, n_unique, len(tgt))]          # heavy repetition, random order
        for nn_variant in (0, 1):
            from sonar_slam_amd import _lib
            c = _lib.default_context()
            c._check(c.lib.sfe_icp_set_tuning(c.handle, nn_variant))
            try:
                msgs, T, it = _icp(icp_config.shipped_params(max_iter=8, use_diff_checker=0)).compute_batch(src, t, [guess])
            finally:
                c._check(c.lib.sfe_icp_set_tuning(c.handle, 0))
            st, To, ito = oracle.icp(src, t, guess, oracle.shipped_icp_params(max_iter=8, use_diff_checker=0))
            assert msgs[0] == "success" and st == 0 and it[0] == ito
            assert _pose_diff(T[0], To) < TOL_TIGHT, (n_unique, nn_variant)


def test_downsample_matches_oracle_octree():
    """pcl.downsample: GPU key/rank formulation vs the oracle's recursive quadtree (order, values,
    indices) on uniform, clustered, duplicated and degenerate clouds."""
    rng = np.random.default_rng(5)
    clouds = [rng.uniform(-12, 17, (3000, 2)), rng.normal(0, 0.7, (1500, 2)),
              np.repeat(rng.uniform(-5, 5, (40, 2)), 25, axis=0),        # exact duplicates
              np.c_[np.linspace(0, 30, 800), np.zeros(800)],             # collinear: zero y-extent
              rng.uniform(0, 0.2, (50, 2)), np.array([[3.0, 4.0]]), np.array([[1.0, 1.0], [1.0, 1.0]])]
    for pts in clouds:
        pts = pts.astype(np.float32)
        for res in (0.5, 0.25, 2.0):
            want, widx = oracle.downsample(pts, res, return_index=True)
            got = pcl.downsample(pts, res)
            assert got.dtype == np.float32 and np.array_equal(got, want), (len(pts), res)
            desc = np.arange(len(pts), dtype=np.float32)[:, None]
            g2, d2 = pcl.downsample(pts, desc, res)
            assert np.array_equal(g2, want) and np.array_equal(d2[:, 0].astype(np.int32), widx)
    assert pcl.downsample(np.zeros((0, 2), np.float32), 0.5).shape == (0, 2)


def test_downsample_without_indices_large_and_deep_clouds():
    """pcl.downsample(points, resolution) runs the resident batch path as a batch of one (sort in LDS up to 16 384
    points, in global memory up to 65 536, the rank-counting path above that and for trees deeper than 24 levels);
    the descriptor overload keeps the rank-counting path.  All of them against the oracle's octree."""
    rng = np.random.default_rng(55)
    cases = [(rng.uniform(-30, 30, (11000, 2)), 0.5), (rng.uniform(-30, 30, (16384, 2)), 0.5),
             (rng.uniform(-30, 30, (16385, 2)), 0.3), (rng.uniform(-60, 60, (40000, 2)), 0.5),
             (rng.uniform(-60, 60, (70000, 2)), 1.0),
             (rng.uniform(-30, 30, (600, 2)), 1e-6),     # a tree deeper than 24 levels: falls back
             (np.r_[rng.uniform(-30, 30, (900, 2)), rng.uniform(0, 1e-4, (100, 2))], 1e-5)]
    for pts, res in cases:
        pts = pts.astype(np.float32)
        want, widx = oracle.downsample(pts, res, return_index=True)
        got = pcl.downsample(pts, res)
        assert np.array_equal(got, want), (len(pts), res)
        desc = np.arange(len(pts), dtype=np.float32)[:, None]
        g2, d2 = pcl.downsample(pts, desc, res)
        assert np.array_equal(g2, want) and np.array_equal(d2[:, 0].astype(np.int64), widx), (len(pts), res)


def test_downsample_with_a_non_positive_resolution_splits_down_to_single_points():
    """maxSizeByNode = 0 has no size limit in libpointmatcher's octree: one point per leaf, in path order
    (not "skip the filter", which is what resolution <= 0 means to the batched FeatureExtraction tail)."""
    rng = np.random.default_rng(56)
    pts = rng.uniform(-10, 10, (300, 2)).astype(np.float32)
    for res in (0.0, -1.0):
        want = oracle.downsample(pts, res)
        got = pcl.downsample(pts, res)
        assert len(want) == 300 and not np.array_equal(want, pts)
        assert np.array_equal(got, want), res


def test_feature_extraction_callback_end_to_end(shipped_cfar):
    """FeatureExtraction.callback (feature_extraction.py:196-252 without ROS) vs the oracle chain
    CFAR -> gate -> remap -> nonzero -> px2m -> downsample -> remove_outlier."""
    from sonar_slam_amd.feature_extraction import FeatureExtraction, SonarPing, oculus_bearings
    fe = FeatureExtraction()
    fe.Ntc, fe.Ngc, fe.Pfa, fe.rank, fe.alg, fe.threshold, fe.skip = 40, 10, 0.1, 10, "SOCA", 65, 1
    fe.configure()
    img = synth.sonar_frame(seed=77)
    ping = SonarPing(img, oculus_bearings(512), 30.0 / 1024, ping_id=3)
    pts = fe.callback(ping)
    th, gh, tau = shipped_cfar.params["SOCA"]
    m = oracle.gate(img, oracle.cfar(img, "SOCA", th, gh, tau), 65)
    rc = oracle.nonzero(oracle.remap_u8(m, fe.map_x, fe.map_y))
    p = oracle.px_to_m(rc, fe.rows, fe.cols, fe.width, fe.height)
    p = oracle.downsample(p, 0.5)
    p = oracle.remove_outlier(p, 1.0, 5)
    assert len(p) > 50 and np.array_equal(pts, p)
    fe.skip = 2  # skipped pings publish one NaN point (feature_extraction.py:201-207)
    assert np.isnan(fe.callback(ping)).all()


def _with_variant(ctx, variant, fn):
    ctx._check(ctx.lib.sfe_icp_set_tuning(ctx.handle, variant))
    try:
        return fn()
    finally:
        ctx._check(ctx.lib.sfe_icp_set_tuning(ctx.handle, 0))


@pytest.mark.parametrize("mz", [0, 1])
def test_sweep_search_equals_brute_force_bit_for_bit(ctx, mz):
    """The default strip-sweep search and the brute-force tile scan (tuning bit 2) must take the
    same decisions: identical transforms, iteration counts and statuses, also with duplicated
    target points (ties -> lowest original index) and with far outliers in the source."""
    src, tgt, guess, _ = synth.scan_pair(seed=21, n_src=4000, n_tgt=3500)
    tgt = np.concatenate([tgt, tgt[:700], tgt[100:200]]).astype(np.float32)     # exact duplicates
    src[::97] += 40.0                                                            # nothing within maxDist
    p = icp_config.shipped_params(minimizer=mz, max_iter=12 if mz else 40, use_diff_checker=0 if mz else 1)
    base = synth.pose_of(guess)
    rng = np.random.default_rng(5)
    guesses = [synth.pose_matrix(base[0] + dx, base[1] + dy, base[2] + dt).astype(np.float32)
               for dx, dy, dt in rng.normal(0, [0.4, 0.4, 0.06], (6, 3))]
    a = _with_variant(ctx, 0, lambda: _icp(p, ctx).compute_batch(src, tgt, guesses))
    b = _with_variant(ctx, 4, lambda: _icp(p, ctx).compute_batch(src, tgt, guesses))
    assert a[0] == b[0]
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


def test_hires_many_to_one_batch_sweep_equals_brute_force(ctx, job_tiers):
    """BASELINE configs[4] shape: 20k-point clouds (target beyond the LDS capacity: sorted in HBM
    scratch, walked through L2), several guesses on one pair.  Too large for the CPU oracle in a
    test, so the exhaustive GPU kernel is the checker here (it is oracle-checked at 9000 points
    above).  Unsplit, the sweep is bit-identical to it.  Split over several workgroups (the default for
    this shape) the discrete decisions -- matches, trimmed limit, iteration count -- are the same and
    the fp64 sums are added in another order (per share, then over the shares): statuses and
    iteration counts equal, poses within 1e-6."""
    src, tgt, guess, _ = synth.scan_pair(seed=33, n_src=20000, n_tgt=20000)
    base = synth.pose_of(guess)
    rng = np.random.default_rng(9)
    guesses = [synth.pose_matrix(base[0] + dx, base[1] + dy, base[2] + dt).astype(np.float32)
               for dx, dy, dt in rng.normal(0, [0.3, 0.3, 0.05], (4, 3))]
    for mz, iters in ((0, 8), (1, 6)):
        p = icp_config.shipped_params(minimizer=mz, max_iter=iters, use_diff_checker=0)
        a = _with_variant(ctx, 0, lambda: _icp(p, ctx).compute_batch(src, tgt, guesses))
        b = _with_variant(ctx, 4, lambda: _icp(p, ctx).compute_batch(src, tgt, guesses))
        assert a[0] == b[0] and all(m == "success" for m in a[0])
        assert np.array_equal(a[2], b[2])
        if job_tiers == "one-size":
            assert np.array_equal(a[1], b[1])
        else:
            assert max(_pose_diff(x, y) for x, y in zip(a[1], b[1])) < TOL_TIGHT


@pytest.mark.parametrize("shares", [0, 2, 3, 16])
def test_split_job_takes_the_decisions_of_the_unsplit_one(ctx, monkeypatch, shares):
    """A job shared by several workgroups (sfe_icp_sweep.hip, MULTI): 30-iteration chains of both minimisers and the
    shipped chain with its differential stop, 3 guesses on one 20 000 x 12 000 pair; any number of shares (0 = the
    launcher's choice) must reproduce status and iteration count of the unsplit run and its pose to 1e-6.  The source
    has far outliers and a band without any point, so some shares hold few queries."""
    src, tgt, guess, _ = synth.scan_pair(seed=41, n_src=20000, n_tgt=12000)
    src[::53] += 45.0
    base = synth.pose_of(guess)
    rng = np.random.default_rng(10)
    guesses = [synth.pose_matrix(base[0] + dx, base[1] + dy, base[2] + dt).astype(np.float32)
               for dx, dy, dt in rng.normal(0, [0.3, 0.3, 0.05], (3, 3))]
    for over in (dict(minimizer=1, max_iter=30, use_diff_checker=0), dict(minimizer=0, max_iter=30, use_diff_checker=0), {}):
        p = icp_config.shipped_params(**over)
        monkeypatch.setenv("SFE_SW_MULTI", "0")
        ref = _icp(p, ctx).compute_batch(src, tgt, guesses)
        monkeypatch.setenv("SFE_SW_MULTI", "1")
        monkeypatch.setenv("SFE_SW_TIERS", "1")
        if shares:
            monkeypatch.setenv("SFE_SW_MULTI_G", str(shares))
        got = _icp(p, ctx).compute_batch(src, tgt, guesses)
        assert got[0] == ref[0] and all(m == "success" for m in got[0]), (over, got[0])
        assert np.array_equal(got[2], ref[2]), (over, got[2], ref[2])
        assert max(_pose_diff(x, y) for x, y in zip(got[1], ref[1])) < TOL_TIGHT


def test_split_job_with_empty_shares(ctx, monkeypatch):
    """all queries in two thin bands of strips: most of the 16 shares get no query at all and still take part in
    every exchange"""
    src, tgt, guess, _ = synth.scan_pair(seed=42, n_src=9000, n_tgt=9000)
    big_t = np.concatenate([tgt, tgt + np.float32(0.013)]).astype(np.float32)
    keep = (np.abs(src[:, 1] - np.median(src[:, 1])) < 0.4) | (src[:, 1] > np.quantile(src[:, 1], 0.97))
    band = np.concatenate([src[keep]] * 6)[:8800].astype(np.float32)
    monkeypatch.setenv("SFE_SW_MULTI_MIN_SRC", "4096")
    monkeypatch.setenv("SFE_SW_MULTI_SHARE_MIN", "64")
    p = icp_config.shipped_params(minimizer=1, max_iter=10, use_diff_checker=0)
    monkeypatch.setenv("SFE_SW_MULTI", "0")
    ref = _icp(p, ctx).compute_batch(band, big_t, [guess])
    monkeypatch.setenv("SFE_SW_MULTI", "1")
    monkeypatch.setenv("SFE_SW_TIERS", "1")
    got = _icp(p, ctx).compute_batch(band, big_t, [guess])
    assert got[0] == ref[0] and np.array_equal(got[2], ref[2])
    assert _pose_diff(got[1][0], ref[1][0]) < TOL_TIGHT


def test_small_job_tiers_in_one_batch_equal_brute_force_and_the_oracle(ctx, job_tiers):
    """the job shapes bruce_slam produces (slam.py:769,1032: clouds of 10^2..10^3 points) side by side in one device
    batch: one-wave, four-wave and 1024-thread workgroups, one launch each; bit-identical to the brute-force kernel,
    a sample against the oracle"""
    from sonar_slam_amd.CFAR import CFAR
    from sonar_slam_amd.feature_extraction import FeatureExtraction, SonarPing, oculus_bearings
    from sonar_slam_amd.pipeline import KeyframeBatch
    rng = np.random.default_rng(3)
    sizes = [(int(a), int(b)) for a, b in zip(rng.integers(40, 380, 40), rng.integers(40, 500, 40))]
    sizes += [(int(a), int(b)) for a, b in zip(rng.integers(400, 2000, 24), rng.integers(500, 2040, 24))]
    sizes += [(384, 512), (385, 512), (384, 513), (2048, 2048), (2049, 2048), (3000, 300), (200, 3000), (5000, 5000), (1, 1), (2, 700)]
    pairs = [synth.scan_pair(seed=700 + i, n_src=a, n_tgt=b) for i, (a, b) in enumerate(sizes)]
    fe = FeatureExtraction(ctx)
    fe.Ntc, fe.Ngc, fe.Pfa, fe.rank, fe.alg, fe.threshold = 40, 10, 0.1, 10, "SOCA", 65
    fe.configure()
    fe.generate_map_xy(SonarPing(np.zeros((64, 64), np.uint8), oculus_bearings(64), 0.25))
    for over in ({}, dict(minimizer=1, max_iter=30, use_diff_checker=0)):
        p = icp_config.shipped_params(**over)
        out = {}
        for variant in (0, 4):
            kb = KeyframeBatch(ctx, fe.geometry, CFAR(40, 10, 0.1, 10).params["SOCA"], "SOCA", 65, p, len(pairs))
            kb.upload_scan_pairs([q[0] for q in pairs], [q[1] for q in pairs], [q[2] for q in pairs])
            _with_variant(ctx, variant, lambda: (kb.run_icp(), ctx.sync()))
            out[variant] = kb.results()
            kb.free()
        for k in ("T", "status", "iters"):
            assert np.array_equal(out[0][k], out[4][k], equal_nan=True), (k, over)
        for j in (0, 7, 41, 50, 64, 66, 70):
            st, To, ito = oracle.icp(pairs[j][0], pairs[j][1], pairs[j][2], oracle.shipped_icp_params(precision=1, **over))
            assert st == out[0]["status"][j] and ito == out[0]["iters"][j], (j, over)
            if st == 0:
                assert _pose_diff(out[0]["T"][j], To) < TOL_TIGHT, (j, over)


def test_mixed_size_device_batch(ctx):
    """one launch with LDS-resident and HBM-resident targets side by side (sfe_icp_batch_dev)"""
    from sonar_slam_amd.CFAR import CFAR
    from sonar_slam_amd.feature_extraction import FeatureExtraction, SonarPing, oculus_bearings
    from sonar_slam_amd.pipeline import KeyframeBatch
    sizes = [(700, 900), (3000, 9500), (1500, 1500), (2500, 8193), (64, 8192)]
    pairs = [synth.scan_pair(seed=50 + i, n_src=a, n_tgt=b) for i, (a, b) in enumerate(sizes)]
    fe = FeatureExtraction(ctx)
    fe.Ntc, fe.Ngc, fe.Pfa, fe.rank, fe.alg, fe.threshold = 40, 10, 0.1, 10, "SOCA", 65
    fe.configure()
    fe.generate_map_xy(SonarPing(np.zeros((64, 64), np.uint8), oculus_bearings(64), 0.25))
    p = icp_config.shipped_params()
    out = {}
    for variant in (0, 4):
        kb = KeyframeBatch(ctx, fe.geometry, CFAR(40, 10, 0.1, 10).params["SOCA"], "SOCA", 65, p, len(pairs))
        kb.upload_scan_pairs([q[0] for q in pairs], [q[1] for q in pairs], [q[2] for q in pairs])
        _with_variant(ctx, variant, lambda: (kb.run_icp(), ctx.sync()))
        out[variant] = kb.results()
        kb.free()
    for k in ("T", "status", "iters"):
        assert np.array_equal(out[0][k], out[4][k]), k
    st, To, _ = oracle.icp(pairs[0][0], pairs[0][1], pairs[0][2], oracle.shipped_icp_params(precision=1))
    assert st == 0 and _pose_diff(out[0]["T"][0], To) < TOL_TIGHT


def test_sweep_vs_brute_force_fuzz(ctx):
    """Randomised differential test of the two ICP kernels: cloud sizes from 1 point up, every filter
    on/off, both minimisers, duplicated / collinear / far-away / non-finite points.  Both kernels
    must return identical transforms, statuses and iteration counts."""
    rng = np.random.default_rng(2024)
    from sonar_slam_amd._lib import IcpParams
    n_fail = 0
    for case in range(70):
        ns, nt = int(rng.integers(1, 400)), int(rng.integers(1, 400))
        if case % 7 == 0:
            ns, nt = int(rng.integers(1, 6)), int(rng.integers(1, 6))
        tgt = rng.uniform(-8, 8, (nt, 2)).astype(np.float32)
        if case % 3 == 0:                                   # points on a few lines (x-degenerate windows)
            tgt[:, 0] = np.round(tgt[:, 0])
        if case % 5 == 0 and nt > 4:                        # exact duplicates
            tgt[nt // 2:] = tgt[:nt - nt // 2]
        th = rng.uniform(-0.2, 0.2)
        R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
        src = (tgt[rng.integers(0, nt, ns)] @ R.T + rng.normal(0, 0.05, (ns, 2)) + rng.uniform(-0.5, 0.5, 2)).astype(np.float32)
        if case % 4 == 0:
            src[rng.integers(0, ns)] += 100.0               # nothing within maxDist
        if case % 11 == 0:
            src[rng.integers(0, ns), 0] = np.nan
        if case % 13 == 0:
            tgt[rng.integers(0, nt), 1] = np.inf
        p = IcpParams(matcher_max_dist=float(rng.choice([0.5, 3.0, 10.0])), use_max_dist_filter=int(rng.integers(0, 2)),
                      max_dist_filter=float(rng.choice([0.3, 3.0, 20.0])), use_trimmed_filter=int(rng.integers(0, 2)),
                      trim_ratio=float(rng.choice([0.3, 0.8, 1.0])), minimizer=int(rng.integers(0, 2)),
                      max_iter=int(rng.integers(1, 15)), use_diff_checker=int(rng.integers(0, 2)), min_diff_rot=0.001,
                      min_diff_trans=0.01, smooth_len=int(rng.integers(1, 4)), normals_knn=int(rng.integers(2, 17)))
        guesses = [synth.pose_matrix(*rng.normal(0, [0.3, 0.3, 0.05])).astype(np.float32) for _ in range(3)]
        a = _with_variant(ctx, 0, lambda: _icp(p, ctx).compute_batch(src, tgt, guesses))
        b = _with_variant(ctx, 4, lambda: _icp(p, ctx).compute_batch(src, tgt, guesses))
        same = a[0] == b[0] and np.array_equal(a[1], b[1], equal_nan=True) and np.array_equal(a[2], b[2])
        assert same, (case, ns, nt, p.as_dict(), a[0], b[0], a[2], b[2])
        n_fail += sum(m != "success" for m in a[0])
    assert 0 < n_fail < 200      # the fuzz reaches both the success and the failure paths


def test_rank_deficient_jobs_come_out_the_same_from_every_kernel(ctx):
    """A few hundred source points on a target of 2..7 points (duplicates among them): every inlier is matched to one or
    two points, the point-to-point system is rank-deficient, its sums are rounding noise and the closed-form solve amplifies
    them without bound -- the last bit of an fp64 sum decides the pose.  Every build adds the mean and the minimiser's
    sums in the order of a 1024-thread workgroup, so the one-wave kernels still equal the brute-force kernel bit for bit
    (tools/icp_soak.py found 7 such batches in 113 000 scan matches before that)."""
    from sonar_slam_amd._lib import IcpParams
    rng = np.random.default_rng(4711)
    for rep in range(6):
        srcs, tgts, gs = [], [], []
        for _ in range(48):
            ns, nt = int(rng.integers(150, 320)), int(rng.integers(2, 8))
            tgt = rng.uniform(-8, 8, (nt, 2)).astype(np.float32)
            if rng.random() < 0.5:
                tgt[:, 0] = np.round(tgt[:, 0] * 2) / 2
            if nt > 3 and rng.random() < 0.6:
                tgt[nt // 2:] = tgt[:nt - nt // 2]
            src = (tgt[rng.integers(0, nt, ns)] + rng.normal(0, 0.1, (ns, 2))).astype(np.float32)
            srcs.append(src)
            tgts.append(tgt)
            gs.append(synth.pose_matrix(*rng.normal(0, [0.3, 0.3, 0.05])).astype(np.float32))
        p = IcpParams(matcher_max_dist=float(rng.choice([0.5, 3.0, 10.0])), use_max_dist_filter=int(rng.integers(0, 2)),
                      max_dist_filter=float(rng.choice([0.3, 3.0])), use_trimmed_filter=int(rng.integers(0, 2)),
                      trim_ratio=float(rng.choice([0.3, 1.0])), minimizer=0, max_iter=int(rng.integers(3, 14)),
                      use_diff_checker=int(rng.integers(0, 2)), min_diff_rot=0.001, min_diff_trans=0.01,
                      smooth_len=int(rng.integers(1, 4)), normals_knn=10)
        a = _with_variant(ctx, 0, lambda: _icp(p, ctx).compute_pairs(srcs, tgts, gs))
        b = _with_variant(ctx, 4, lambda: _icp(p, ctx).compute_pairs(srcs, tgts, gs))
        assert a[0] == b[0] and np.array_equal(a[2], b[2]), rep
        assert np.array_equal(a[1], b[1], equal_nan=True), rep


def _sweep_vs_brute(ctx, p, src, tgt, guesses):
    a = _with_variant(ctx, 0, lambda: _icp(p, ctx).compute_batch(src, tgt, guesses))
    b = _with_variant(ctx, 4, lambda: _icp(p, ctx).compute_batch(src, tgt, guesses))
    same = a[0] == b[0] and np.array_equal(a[1], b[1], equal_nan=True) and np.array_equal(a[2], b[2])
    return same, a, b


def test_sweep_iteration_cache_at_the_maxdist_boundary(ctx, monkeypatch):
    """From the second iteration on the sweep reuses what the previous iteration knew: the old
    neighbour as a witness, and for queries without any target within maxDist a clearance that
    proves "still none" while the query has moved less than it.  Source points are placed in a thin
    shell around maxDist from the target and the guess is off, so queries cross the boundary in both
    directions while the pose converges; results must stay identical to the brute-force kernel, with
    the cache on and off, for a finite and for an unbounded matcher."""
    from sonar_slam_amd._lib import IcpParams
    rng = np.random.default_rng(77)
    world = synth._structure(rng, 1500, max_range=12.0)
    tgt = (world + rng.normal(0, 0.02, world.shape)).astype(np.float32)
    inl = world[rng.permutation(len(world))[:900]] + rng.normal(0, 0.02, (900, 2))
    for md in (1.5, float("inf")):
        r = (md if np.isfinite(md) else 3.0) + rng.uniform(-0.08, 0.25, 1200)
        a = rng.uniform(0, 2 * np.pi, 1200)
        shell = world[rng.integers(0, len(world), 1200)] + np.c_[r * np.cos(a), r * np.sin(a)]
        far = rng.uniform(-60, 60, (100, 2))
        src_t = np.concatenate([inl, shell, far])
        T = synth.pose_matrix(0.5, -0.2, 0.06)
        Tinv = np.linalg.inv(T)
        src = (src_t @ Tinv[:2, :2].T + Tinv[:2, 2]).astype(np.float32)
        guesses = [synth.pose_matrix(0.5 + dx, -0.2 + dy, 0.06 + dt).astype(np.float32)
                   for dx, dy, dt in rng.normal(0, [0.15, 0.15, 0.03], (4, 3))]
        for mz in (0, 1):
            p = IcpParams(matcher_max_dist=md, use_max_dist_filter=1, max_dist_filter=1.0, use_trimmed_filter=1,
                          trim_ratio=0.7, minimizer=mz, max_iter=25, use_diff_checker=0, min_diff_rot=0.001,
                          min_diff_trans=0.01, smooth_len=3, normals_knn=10)
            for cache in ("1", "0"):
                monkeypatch.setenv("SFE_SW_CACHE", cache)
                same, x, y = _sweep_vs_brute(ctx, p, src, tgt, guesses)
                assert same, (md, mz, cache, x[0], y[0], x[2], y[2])
                assert all(m == "success" for m in x[0]) and (x[2] == 25).all()
            st, To, ito = oracle.icp(src, tgt, guesses[0], oracle.shipped_icp_params(
                minimizer=mz, precision=1, matcher_max_dist=md, max_dist_filter=1.0, trim_ratio=0.7, max_iter=25,
                use_diff_checker=0, min_diff_rot=0.001, min_diff_trans=0.01, smooth_len=3))
            assert st == 0 and ito == 25 and _pose_diff(x[1][0], To) < TOL_REF


def test_sweep_vs_brute_force_fuzz_many_iterations(ctx):
    """second differential fuzz: larger clouds, 10-25 iterations (the iteration-to-iteration cache is
    exercised over many steps), small / unbounded matcher radii"""
    from sonar_slam_amd._lib import IcpParams
    rng = np.random.default_rng(909)
    for case in range(24):
        ns, nt = int(rng.integers(200, 1500)), int(rng.integers(200, 1500))
        world = synth._structure(rng, max(ns, nt), max_range=float(rng.choice([6.0, 15.0, 30.0])))
        tgt = (world[rng.permutation(len(world))[:nt]] + rng.normal(0, 0.03, (nt, 2))).astype(np.float32)
        if case % 4 == 0:
            tgt[:, case % 8 // 4] = np.round(tgt[:, case % 8 // 4] * 2) / 2      # walls on a half-metre raster
        T = synth.pose_matrix(*rng.normal(0, [0.4, 0.4, 0.05]))
        Tinv = np.linalg.inv(T)
        base = world[rng.permutation(len(world))[:ns]] + rng.normal(0, 0.03, (ns, 2))
        src = base @ Tinv[:2, :2].T + Tinv[:2, 2]
        k = ns // 4
        src[:k] = rng.uniform(-35, 35, (k, 2))                                   # outliers, some beyond everything
        src = src.astype(np.float32)
        if case % 6 == 5:
            tgt[int(rng.integers(0, nt)), int(rng.integers(0, 2))] = np.inf     # a point nothing can match
        p = IcpParams(matcher_max_dist=float(rng.choice([0.4, 1.0, 10.0, np.inf])),
                      use_max_dist_filter=int(rng.integers(0, 2)), max_dist_filter=float(rng.choice([0.3, 3.0])),
                      use_trimmed_filter=int(rng.integers(0, 4) > 0), trim_ratio=float(rng.choice([0.5, 0.8, 0.95])),
                      minimizer=int(rng.integers(0, 2)), max_iter=int(rng.integers(10, 26)), use_diff_checker=0,
                      min_diff_rot=0.001, min_diff_trans=0.01, smooth_len=3, normals_knn=10)
        guesses = [(T @ synth.pose_matrix(*rng.normal(0, [0.2, 0.2, 0.04]))).astype(np.float32) for _ in range(3)]
        same, a, b = _sweep_vs_brute(ctx, p, src, tgt, guesses)
        assert same, (case, ns, nt, p.as_dict(), a[0], b[0], a[2], b[2])


def test_side_stream_preparation_returns_the_same_results(ctx):
    """sfe_icp_set_tuning bit 3: the targets' preparation runs on the library's side stream and only the
    iteration kernel waits for it; launches issued back to back (the next preparation must not overwrite the
    scratch the previous iteration kernel still reads) return exactly what the one-stream order returns"""
    from sonar_slam_amd.CFAR import CFAR
    from sonar_slam_amd.feature_extraction import FeatureExtraction, SonarPing, oculus_bearings
    from sonar_slam_amd.pipeline import KeyframeBatch
    fe = FeatureExtraction(ctx)
    fe.Ntc, fe.Ngc, fe.Pfa, fe.rank, fe.alg, fe.threshold = 40, 10, 0.1, 10, "SOCA", 65
    fe.configure()
    fe.generate_map_xy(SonarPing(np.zeros((64, 64), np.uint8), oculus_bearings(64), 0.25))
    p = icp_config.shipped_params(minimizer=1, use_diff_checker=0, max_iter=12)
    batches = []
    for b in range(3):                                        # three different batches of 24 scan pairs
        pairs = [synth.scan_pair(seed=900 + 31 * b + i, n_src=1500 + 40 * i, n_tgt=1400 + 55 * i) for i in range(24)]
        kb = KeyframeBatch(ctx, fe.geometry, CFAR(40, 10, 0.1, 10).params["SOCA"], "SOCA", 65, p, len(pairs))
        kb.upload_scan_pairs([q[0] for q in pairs], [q[1] for q in pairs], [q[2] for q in pairs])
        batches.append(kb)
    want = []
    for kb in batches:
        kb.run_icp()
        want.append(kb.results())
    ctx._check(ctx.lib.sfe_icp_set_tuning(ctx.handle, 8))
    try:
        for _ in range(2):
            for kb in batches:                                # no sync in between: the launches overlap on the device
                kb.run_icp()
        got = [kb.results() for kb in batches]
    finally:
        ctx._check(ctx.lib.sfe_icp_set_tuning(ctx.handle, 0))
    for w, g in zip(want, got):
        for k in ("T", "status", "iters"):
            assert np.array_equal(w[k], g[k]), k
        assert (g["status"] == 0).all()
    for kb in batches:
        kb.free()


def test_compute_pairs_and_the_farm_equal_single_calls(ctx):
    """ICP.compute_pairs (many independent pairs per launch) and farm.IcpFarm (one worker process per
    device, chunks of pairs per launch) return what one ICP.compute per pair returns"""
    from sonar_slam_amd.farm import IcpFarm
    p = icp_config.shipped_params()
    pairs = [synth.scan_pair(seed=300 + i, n_src=400 + 37 * i, n_tgt=500 + 11 * i) for i in range(7)]
    icp = _icp(p, ctx)
    single = [icp.compute(s, t, g) for s, t, g, _ in pairs]
    msgs, T, it = icp.compute_pairs([q[0] for q in pairs], [q[1] for q in pairs], [q[2] for q in pairs])
    for (m1, T1), m2, T2 in zip(single, msgs, T):
        assert m1 == m2 and np.array_equal(T1, T2)
    jobs = [(s, t, [g, g @ synth.pose_matrix(0.1, 0.0, 0.01).astype(np.float32)]) for s, t, g, _ in pairs[:4]]
    out = IcpFarm(p, devices=[ctx.device]).run(jobs)
    assert len(out) == 4
    for (s, t, gs), (m, Tf, itf) in zip(jobs, out):
        mb, Tb, itb = icp.compute_batch(s, t, gs)
        assert m == mb and np.array_equal(Tf, Tb) and np.array_equal(itf, itb)


def test_match_knn_bit_exact():
    """pcl.match with knn > 1 (pcl.cpp:161-174): the knn nearest in ascending (d2, index) order, -1 / inf where
    fewer lie within max_dist; against the oracle, with duplicated reference points; knn > n_ref throws like libnabo
    (the C entry point itself pads with -1 / inf)"""
    rng = np.random.default_rng(3)
    ref = rng.uniform(-10, 10, (2500, 2)).astype(np.float32)
    ref[100:140] = ref[:40]                                  # exact duplicates: ties by index
    q = rng.uniform(-11, 11, (1700, 2)).astype(np.float32)
    for knn, md in ((2, 0.6), (5, 3.0), (9, 100.0)):
        ids, d2 = pcl.match(ref, q, knn, md)
        oi, od = oracle.match_knn(ref, q, knn, md)
        assert ids.shape == (knn, 1700) and ids.dtype == np.int32 and d2.dtype == np.float32
        assert np.array_equal(ids, oi) and np.array_equal(d2, od)
        assert np.array_equal(ids[0], pcl.match(ref, q, 1, md)[0][0])
    small = ref[:3]
    with pytest.raises(RuntimeError, match="libnabo"):
        pcl.match(small, q[:10], 5, 100.0)
    ids, d2 = pcl.match(small, q[:10], 3, 100.0)
    assert (ids >= 0).all() and np.array_equal(ids, oracle.match_knn(small, q[:10], 3, 100.0)[0])


def test_knn_density_and_max_density_filter():
    """the two stages of the reference's density_filter body (pcl.cpp:81-97): densities against the oracle (bit
    exact), the std::rand thinning against a plain restatement of MaxDensityDataPointsFilter on those densities"""
    rng = np.random.default_rng(8)
    pts = np.r_[rng.normal(0, 0.3, (600, 2)), rng.uniform(-15, 15, (500, 2))].astype(np.float32)
    for knn in (3, 10):
        dens = pcl.knn_density(pts, knn)
        assert np.array_equal(dens, oracle.knn_density(pts, knn))
    with pytest.raises(RuntimeError, match="more points than available"):
        pcl.knn_density(pts[:4], 5)
    dens = oracle.knn_density(pts, 10)
    md = np.float32(np.median(dens))
    g = pcl._GlibcRand(1)
    keep = []
    for d in dens:
        if d > md:
            r = np.float32(g.rand()) / np.float32(2147483647)
            keep.append(bool(r < np.float32(md / d)))      # 1 - nbSaturated / nbPoints == 1 (integer division)
        else:
            keep.append(True)
    pcl.srand(1)
    out = pcl.max_density_filter(pts, 10, md)
    assert np.array_equal(out, pts[np.array(keep)]) and 600 < len(out) < len(pts)
    pcl.srand(1)
    out2, desc2 = pcl.max_density_filter(pts, np.arange(len(pts), dtype=np.float32)[:, None], 10, md)
    assert np.array_equal(out2, out) and np.array_equal(desc2[:, 0].astype(int), np.nonzero(keep)[0])


@pytest.mark.parametrize("mz", [0, 1])
def test_clearance_records_over_long_chains_equal_brute_force(ctx, mz, monkeypatch):
    """Chains with a fixed count of >= 12 iterations run the build with clearance records (a converged query whose old
    neighbour is provably still the nearest is not searched again).  70 iterations: past convergence (every query
    settled by its record), past the 64 iterations the records can number, with duplicated targets (ties never get a
    usable record) and far outliers; transforms, iteration counts and statuses must equal the brute-force kernel's
    bit for bit -- and the same with the records switched off."""
    src, tgt, guess, _ = synth.scan_pair(seed=33, n_src=3000, n_tgt=2800)
    tgt = np.concatenate([tgt, tgt[:300]]).astype(np.float32)
    src[::53] += 35.0
    p = icp_config.shipped_params(minimizer=mz, max_iter=70, use_diff_checker=0)
    base = synth.pose_of(guess)
    rng = np.random.default_rng(8)
    guesses = [synth.pose_matrix(base[0] + dx, base[1] + dy, base[2] + dt).astype(np.float32)
               for dx, dy, dt in rng.normal(0, [0.3, 0.3, 0.04], (4, 3))]
    same, a, b = _sweep_vs_brute(ctx, p, src, tgt, guesses)
    assert same, (a[0], b[0], a[2], b[2])
    assert list(a[2]) == [70] * len(guesses)
    monkeypatch.setenv("SFE_SW_REC", "0")
    off = _with_variant(ctx, 0, lambda: _icp(p, ctx).compute_batch(src, tgt, guesses))
    assert off[0] == a[0] and np.array_equal(off[1], a[1]) and np.array_equal(off[2], a[2])


def test_clearance_records_with_results_in_hbm_scratch(ctx):
    """A source too large for the per-query results to live in LDS next to the target (9000 x 3100 points: the
    LDS_TGT build, results in HBM scratch) through the records build: equal to brute force."""
    src, tgt, guess, _ = synth.scan_pair(seed=34, n_src=9000, n_tgt=2800)
    tgt = np.concatenate([tgt, tgt[:300]]).astype(np.float32)
    p = icp_config.shipped_params(minimizer=1, max_iter=40, use_diff_checker=0)
    base = synth.pose_of(guess)
    guesses = [synth.pose_matrix(base[0] + 0.2, base[1] - 0.1, base[2] + 0.03).astype(np.float32),
               np.asarray(guess, np.float32)]
    same, a, b = _sweep_vs_brute(ctx, p, src, tgt, guesses)
    assert same, (a[0], b[0], a[2], b[2])


@pytest.mark.parametrize("offset", [300.0, 3000.0])
def test_clearance_records_far_from_the_origin(ctx, offset):
    """VERDICT r3 5a / ADVICE r2: pcl.ICP.compute is a general API (mapping.py and any caller in a global frame hand it
    clouds hundreds of metres from the origin).  The clearance-record test allows for the rounding of the two
    transformed positions with a slack that scales with the extent (sfe_icp_sweep.hip, `max(3e-5, 6e-7 (rmax + |t|))`):
    with both clouds offset by +300 m / +3000 m the records build (>= 12 fixed iterations) must still equal the
    brute-force kernel bit for bit and the oracle (fp64 sums) in status, iteration count and pose."""
    src, tgt, guess, _ = synth.scan_pair(seed=35, n_src=2500, n_tgt=2600)
    shift = np.array([offset, -0.7 * offset], np.float32)
    # the guess maps source into target coordinates: T' = S T S^-1 for a common translation S of both clouds
    S = synth.pose_matrix(float(shift[0]), float(shift[1]), 0.0)
    g64 = S @ np.asarray(guess, np.float64) @ np.linalg.inv(S)
    src_o, tgt_o = (src + shift).astype(np.float32), (tgt + shift).astype(np.float32)
    for mz in (0, 1):
        p = icp_config.shipped_params(minimizer=mz, max_iter=24, use_diff_checker=0)
        guesses = [g64.astype(np.float32), (g64 @ synth.pose_matrix(0.05, -0.04, 0.004)).astype(np.float32)]
        same, a, b = _sweep_vs_brute(ctx, p, src_o, tgt_o, guesses)
        assert same, (offset, mz, a[0], b[0], a[2], b[2])
        assert list(a[2]) == [24, 24]
        for g, msg, T, it in zip(guesses, a[0], a[1], a[2]):
            st, To, ito = oracle.icp(src_o, tgt_o, g, oracle.IcpParams(precision=1, **p.as_dict()))
            assert msg == oracle.ICP_STATUS_MESSAGES[st] and it == ito
            # float32 transforms at |t| ~ 3000 m resolve 2.4e-4 m: compare in units of the translation's ulp
            tol = max(TOL_TIGHT, 2.0 * float(np.spacing(np.float32(offset))))
            assert _pose_diff(T, To) <= tol, (offset, mz, _pose_diff(T, To))


def test_two_real_farm_workers_on_one_device(ctx):
    """IcpFarm with two REAL HIP worker processes (two contexts on the one GPU, devices=[0, 0]): two
    batches on the same workers, a bad job between them (ADVICE r2's scenario: nothing may stay in flight), results
    equal to single calls in this process."""
    from sonar_slam_amd.farm import IcpFarm
    p = icp_config.shipped_params()
    pairs = [synth.scan_pair(seed=400 + i, n_src=300 + 41 * i, n_tgt=420 + 13 * i) for i in range(9)]
    icp = _icp(p, ctx)

    def jobs_of(sel):
        return [(pairs[i][0], pairs[i][1], [pairs[i][2], pairs[i][2] @ synth.pose_matrix(0.08, 0.0, 0.01).astype(np.float32)])
                for i in sel]
    with IcpFarm(p, devices=[ctx.device, ctx.device], chunk=3) as f:
        pids = None
        for sel in (range(5), range(9)):
            jobs = jobs_of(sel)
            out = f.run(jobs)
            assert len(out) == len(jobs)
            for (s, t, gs), (m, Tf, itf) in zip(jobs, out):
                mb, Tb, itb = icp.compute_batch(s, t, gs)
                assert list(m) == list(mb) and np.array_equal(Tf, Tb) and np.array_equal(itf, itb)
            now = [w.proc.pid for w in f._workers]
            assert len(now) == 2 and len(set(now)) == 2 and (pids is None or pids == now)
            pids = now
            if sel == range(5):
                bad = jobs_of(range(4))
                bad[1] = (np.zeros((0, 2), np.float32), pairs[1][1], [pairs[1][2]])      # job 1 -> worker 1
                with pytest.raises(RuntimeError, match="empty source"):
                    f.run(bad)


def test_eight_real_farm_workers_on_one_device(ctx):
    """The farm at the target node's width with REAL HIP workers: eight worker processes, eight contexts
    on this box's one GPU (devices=[0] * 8), fewer jobs than workers and more; results equal to single calls here."""
    from sonar_slam_amd.farm import IcpFarm
    p = icp_config.shipped_params()
    pairs = [synth.scan_pair(seed=500 + i, n_src=260 + 31 * i, n_tgt=300 + 17 * i) for i in range(6)]
    icp = _icp(p, ctx)
    with IcpFarm(p, devices=[ctx.device] * 8, chunk=4) as f:
        assert len(f._workers) == 8 and len(set(w.proc.pid for w in f._workers)) == 8
        for n in (3, 19):
            jobs = [(pairs[j % 6][0], pairs[j % 6][1], [pairs[j % 6][2]]) for j in range(n)]
            out = f.run(jobs)
            assert len(out) == n
            for (s_, t_, gs), (m, Tf, itf) in zip(jobs, out):
                mb, Tb, itb = icp.compute_batch(s_, t_, gs)
                assert list(m) == list(mb) and np.array_equal(Tf, Tb) and np.array_equal(itf, itb)
    assert f._workers == []
