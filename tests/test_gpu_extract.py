"""GPU parity: polar->Cartesian remap, nonzero compaction and px->m vs the oracle, bit-exact."""
import os

import numpy as np
import pytest

import oracle
from sonar_slam_amd import synth
from sonar_slam_amd.feature_extraction import FeatureExtraction, Geometry, SonarPing, build_maps, oculus_bearings

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["default", "entries8"])
def extraction_build(request, monkeypatch):
    """Every test of this module runs twice: with the 4-byte inverse-map entries of round 4 (default) and with round 3's
    8-byte entries, which stay as the fallback for geometries whose candidates span more than 127 canvas rows or columns
    (DESIGN 5.2).  The knob is read per call (extract_dev).  (The fused kernel, the second-level flags and round 2's
    row-block kernel ran here until round 5: measured slower, removed, profiles/r05_pruned_variants.txt.)"""
    if request.param == "entries8":
        monkeypatch.setenv("SFE_EXTRACT_NO_COMPACT", "1")
    return request.param


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


@pytest.mark.parametrize("staged", [True, False])
def test_resident_cloud_filters_equal_the_per_cloud_api_and_the_oracle(ctx, shipped_cfar, staged):
    """sfe_cloud_filter_batch_dev (downsample + remove_outlier, device to device, the tail of
    FeatureExtraction.callback) against pcl.downsample / pcl.remove_outlier on the same clouds and
    against the oracle; also with either stage switched off like feature_extraction.py:241,245.
    staged: the round-6 hand-over (sfe_extract_points_bits_staged_dev -> sfe_cloud_filter_staged_dev: float32 pairs +
    bounding boxes instead of float64 points read back and cast) -- the same clouds bit for bit."""
    import oracle
    from sonar_slam_amd import pcl
    from sonar_slam_amd.pipeline import KeyframeBatch
    th, gh, tau = shipped_cfar.params["SOCA"]
    fe = FeatureExtraction(ctx)
    fe.Ntc, fe.Ngc, fe.Pfa, fe.rank, fe.alg, fe.threshold = 40, 10, 0.1, 10, "SOCA", 65
    fe.configure()
    frames = np.stack([synth.sonar_frame(seed=70 + s, rows=512, cols=256, n_blobs=25) for s in range(5)])
    frames[3] = 0                                             # a frame without detections
    fe.generate_map_xy(SonarPing(frames[0], oculus_bearings(256), 30.0 / 512))
    kb = KeyframeBatch(ctx, fe.geometry, (th, gh, tau), "SOCA", 65, None, len(frames), max_points=8192, staged=staged)
    assert kb.staged == staged and kb.points64 == (not staged)
    kb.upload_frames(frames)
    kb.run_cfar()
    kb.run_extract()
    # the last rows aim at the radius filter's cell path: radius far below / around / far above the octree's
    # cell sizes (one cell level .. root cell narrower than the radius -> brute force), a tree deeper than the
    # 8 levels of the narrow sort keys (wide keys, brute-force count), min_points that nothing / everything meets
    for res, rad, mp in ((0.5, 1.0, 5), (0.0, 1.0, 5), (0.5, 1.0, 1), (0.25, 0.6, 3), (0.5, 0.05, 2), (0.5, 0.49, 2),
                         (0.5, 3.7, 40), (0.5, 29.0, 300), (0.5, 500.0, 5), (0.05, 0.3, 3), (2.0, 2.0, 2),
                         (0.5, 1.0, 100000)):
        kb.run_filter(res, rad, mp)
        ctx.sync()
        for j in range(len(frames)):
            pts = kb.points(j)                                # float64, as FeatureExtraction.extract returns them
            want = pts.astype(np.float32)
            want_o = want
            if len(want) and res > 0:
                want = pcl.downsample(want, res, ctx=ctx)
                want_o = oracle.downsample(want_o, res)
            if mp > 1 and len(want):
                want = pcl.remove_outlier(want, rad, mp, ctx=ctx)
                want_o = oracle.remove_outlier(want_o, rad, mp)
            got = kb.cloud(j)
            assert got.dtype == np.float32 and np.array_equal(got, want.reshape(-1, 2)), (res, rad, mp, j)
            assert np.array_equal(got, np.asarray(want_o, np.float32).reshape(-1, 2))
    kb.free()


@pytest.mark.parametrize("points64", [False, True])
def test_staged_hand_over_with_frames_that_leave_the_record_path(ctx, shipped_cfar, points64, monkeypatch):
    """The staged extraction when frames do not fit the record path's per-frame capacities (forced here by shrinking them:
    SFE_EXTRACT_REC_CAP / SFE_EXTRACT_CAPW are read per call): those frames go through the canvas kernels, land as float64
    and are cast by extract_stage_fallback_kernel; the others are staged by the merge kernel itself.  Same clouds as the
    unstaged path and the oracle either way, and points(j) is the float64 cloud in both modes."""
    import oracle
    from sonar_slam_amd.pipeline import KeyframeBatch
    th, gh, tau = shipped_cfar.params["SOCA"]
    fe = FeatureExtraction(ctx)
    fe.Ntc, fe.Ngc, fe.Pfa, fe.rank, fe.alg, fe.threshold = 40, 10, 0.1, 10, "SOCA", 65
    fe.configure()
    # frames of very different density: with a record capacity of 1500 some fit and some do not
    frames = np.stack([synth.sonar_frame(seed=170 + s, rows=512, cols=256, n_blobs=nb) for s, nb in enumerate((2, 25, 4, 40, 0, 12))])
    frames[4] = 0
    fe.generate_map_xy(SonarPing(frames[0], oculus_bearings(256), 30.0 / 512))
    ref = KeyframeBatch(ctx, fe.geometry, (th, gh, tau), "SOCA", 65, None, len(frames), max_points=16384, staged=False)
    ref.upload_frames(frames)
    ref.run_cfar()
    ref.run_extract()
    ref.run_filter(0.5, 1.0, 5)
    ctx.sync()
    want_pts = [ref.points(j) for j in range(len(frames))]
    want_cl = [ref.cloud(j) for j in range(len(frames))]
    ref.free()
    n_rec = sorted(len(p) for p in want_pts)
    for env in ({}, {"SFE_EXTRACT_REC_CAP": "1500"}, {"SFE_EXTRACT_CAPW": "200"}, {"SFE_EXTRACT_REC_CAP": "1"}):
        for k in ("SFE_EXTRACT_REC_CAP", "SFE_EXTRACT_CAPW"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        kb = KeyframeBatch(ctx, fe.geometry, (th, gh, tau), "SOCA", 65, None, len(frames), max_points=16384, staged=True,
                           points64=points64)
        kb.upload_frames(frames)
        kb.run_cfar()
        kb.run_extract()
        kb.run_filter(0.5, 1.0, 5)
        ctx.sync()
        for j in range(len(frames)):
            assert np.array_equal(kb.cloud(j), want_cl[j]), (env, j)
            assert np.array_equal(kb.points(j), want_pts[j]), (env, j)
            o = oracle.remove_outlier(oracle.downsample(want_pts[j].astype(np.float32), 0.5), 1.0, 5) if len(want_pts[j]) else want_cl[j]
            assert np.array_equal(kb.cloud(j), np.asarray(o, np.float32).reshape(-1, 2)), (env, j)
        kb.run_filter(0.25, 0.6, 3)                  # the staged clouds serve a second filter call
        ctx.sync()
        for j in (1, 3):
            o = oracle.remove_outlier(oracle.downsample(want_pts[j].astype(np.float32), 0.25), 0.6, 3)
            assert np.array_equal(kb.cloud(j), o), (env, j)
        kb.free()
    assert n_rec[0] == 0 and n_rec[-1] > 3000


@pytest.mark.parametrize("staged", [True, False])
def test_resident_filters_choose_the_sort_per_frame(ctx, shipped_cfar, staged):
    """A batch whose capacity is beyond the LDS sort (32768 > CF_SORT_CAP) with ONE ping of more than 16384 detections among
    ordinary ones (bench seed 3002: 19 122 points -- the frame that made rank 3 of the 8-rank bench launch raise at the old
    capacity): the dense frame is sorted in HBM scratch, the others in LDS as before; every cloud equals the oracle's."""
    import oracle
    from sonar_slam_amd.pipeline import KeyframeBatch
    th, gh, tau = shipped_cfar.params["SOCA"]
    fe = FeatureExtraction(ctx)
    fe.Ntc, fe.Ngc, fe.Pfa, fe.rank, fe.alg, fe.threshold = 40, 10, 0.1, 10, "SOCA", 65
    fe.configure()
    frames = np.stack([synth.sonar_frame(seed=s) for s in (3001, 3002, 3003)])
    fe.generate_map_xy(SonarPing(frames[0], oculus_bearings(frames.shape[2]), 30.0 / frames.shape[1]))
    kb = KeyframeBatch(ctx, fe.geometry, (th, gh, tau), "SOCA", 65, None, len(frames), max_points=32768, staged=staged)
    kb.upload_frames(frames)
    kb.run_cfar()
    kb.run_extract()
    kb.run_filter(0.5, 1.0, 5)
    counts = kb.results()["counts"]
    assert counts[1] > 16384 and counts[0] <= 16384 and counts[2] <= 16384, counts
    for j in range(len(frames)):
        pts = kb.points(j)
        want = oracle.remove_outlier(oracle.downsample(pts.astype(np.float32), 0.5), 1.0, 5)
        assert np.array_equal(kb.cloud(j), want), j
    kb.free()


def test_staged_filter_call_without_staged_clouds_is_refused(ctx):
    """sfe_cloud_filter_staged_dev checks that the context holds staged clouds of that shape: never stale data"""
    out, cnt = ctx.alloc(4 * 64 * 8), ctx.alloc(4 * 4)
    try:
        ctx.lib.sfe_cloud_filter_batch_dev(ctx.handle, None, None, 0, 64, 0.5, 1.0, 5, None, None)   # (resets nothing: 0 frames)
        rc = ctx.lib.sfe_cloud_filter_staged_dev(ctx.handle, 4, 63, 0.5, 1.0, 5, out.ptr, cnt.ptr)
        assert rc != 0
        with pytest.raises(Exception, match="no staged clouds"):
            ctx._check(rc)
    finally:
        out.free()
        cnt.free()


def test_resident_cloud_filters_hires_frame(ctx, shipped_cfar):
    """BASELINE configs[4] frame shape (2048 x 1024): ~34k detections per ping, beyond the LDS sort
    capacity of the resident downsample -> its HBM-scratch sort path"""
    from sonar_slam_amd import pcl
    from sonar_slam_amd.pipeline import KeyframeBatch
    th, gh, tau = shipped_cfar.params["SOCA"]
    fe = FeatureExtraction(ctx)
    fe.Ntc, fe.Ngc, fe.Pfa, fe.rank, fe.alg, fe.threshold = 40, 10, 0.1, 10, "SOCA", 65
    fe.configure()
    frames = np.stack([synth.sonar_frame(seed=90 + s, rows=2048, cols=1024, n_blobs=120) for s in range(2)])
    fe.generate_map_xy(SonarPing(frames[0], oculus_bearings(1024), 30.0 / 2048))
    kb = KeyframeBatch(ctx, fe.geometry, (th, gh, tau), "SOCA", 65, None, len(frames), max_points=65536)
    kb.upload_frames(frames)
    kb.run_cfar()
    kb.run_extract()
    kb.run_filter(0.5, 1.0, 5)
    ctx.sync()
    for j in range(len(frames)):
        pts = kb.points(j)
        assert len(pts) > 16384
        want = pcl.remove_outlier(pcl.downsample(pts.astype(np.float32), 0.5, ctx=ctx), 1.0, 5, ctx=ctx)
        assert np.array_equal(kb.cloud(j), want)
    kb.free()


@pytest.mark.parametrize("density", [0.002, 0.05, 0.5, 1.0])
def test_inverse_map_extraction_equals_dense_pass_and_oracle(ctx, density):
    """binary masks take the inverse-map (scatter) pass by default; it must produce exactly the points
    of the dense pass (tuning 1) and of the oracle, from almost empty to completely full masks"""
    rng = np.random.default_rng(int(density * 1000))
    beams, ranges, res = 128, 200, 0.15
    fe = FeatureExtraction(ctx)
    fe.generate_map_xy(SonarPing(np.zeros((ranges, beams), np.uint8), oculus_bearings(beams), res))
    for trial in range(3):
        mask = (rng.random((ranges, beams)) < density).astype(np.uint8)
        if trial == 2:
            mask[0, :] = 1
            mask[-1, :] = 1
            mask[:, 0] = 1
            mask[:, -1] = 1                                # image borders: out-of-image taps
        out = {}
        for variant in (0, 1):   # list + lane per set pixel (default), dense pass
            ctx._check(ctx.lib.sfe_extract_set_tuning(ctx.handle, variant))
            try:
                out[variant] = fe.geometry.extract(mask)
            finally:
                ctx._check(ctx.lib.sfe_extract_set_tuning(ctx.handle, 0))
        assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
        rc = oracle.nonzero(oracle.remap_u8(mask, fe.map_x, fe.map_y))
        assert np.array_equal(out[0][0], rc)


def test_fused_ping_call_equals_the_per_stage_chain_and_the_oracle(ctx, shipped_cfar):
    """sfe_feature_extract_ping (one upload, one download, one synchronisation per ping: what
    FeatureExtraction.callback calls) against the per-stage entry points and the oracle chain, on several frames,
    all four CFAR variants, with and without the visualisation image and the filters, and with a too-small
    capacity (retry path)."""
    from sonar_slam_amd.CFAR import CFAR
    from sonar_slam_amd.feature_extraction import FeatureExtraction, SonarPing, oculus_bearings
    det = CFAR(40, 10, 0.1, 10)
    for seed, (rows, cols), alg, res, mp in ((5, (1024, 512), "SOCA", 0.5, 5), (6, (512, 256), "CA", 0.5, 5),
                                             (7, (512, 256), "GOCA", 0.0, 5), (8, (300, 256), "OS", 0.3, 1),
                                             (9, (1024, 512), "SOCA", 0.5, 5)):
        img = synth.sonar_frame(seed=seed, rows=rows, cols=cols)
        ping = SonarPing(img, oculus_bearings(cols), 30.0 / rows, ping_id=0)
        out = {}
        for fused in (True, False):
            fe = FeatureExtraction(ctx)
            fe.Ntc, fe.Ngc, fe.Pfa, fe.rank, fe.alg, fe.threshold, fe.skip = 40, 10, 0.1, 10, alg, 65, 1
            fe.resolution, fe.outlier_filter_min_points = res, mp
            fe.configure()
            fe.fused = fused
            fe.make_vis_image = seed % 2 == 1
            out[fused] = (fe.callback(ping), fe.feature_img)
        a, b = out[True], out[False]
        assert a[0].dtype == np.float32 or len(a[0]) == 0
        assert np.array_equal(np.asarray(a[0], np.float32), np.asarray(b[0], np.float32)), (seed, alg)
        assert (a[1] is None) == (b[1] is None) and (a[1] is None or np.array_equal(a[1], b[1]))
        prm = det.params[alg]
        m = oracle.gate(img, oracle.cfar(img, alg, prm[0], prm[1], prm[-1], k=prm[2] if alg == "OS" else 0), 65)
        p = oracle.px_to_m(oracle.nonzero(oracle.remap_u8(m, fe.map_x, fe.map_y)), fe.rows, fe.cols, fe.width, fe.height)
        p = p.astype(np.float32)
        if len(p) and res > 0:
            p = oracle.downsample(p, res)
        if mp > 1 and len(p):
            p = oracle.remove_outlier(p, 1.0, mp)
        assert np.array_equal(np.asarray(a[0], np.float32), p), (seed, alg)
        if a[1] is not None:
            assert np.array_equal(a[1], oracle.remap_u8(img, fe.map_x, fe.map_y))
    # capacity retry: a first call with room for 64 points only
    got = fe.geometry.feature_extract(img, "SOCA", det.params["SOCA"], 65, 0.5, 1.0, 5, cap=64)
    assert got is not None and np.array_equal(got[0], np.asarray(out[True][0], np.float32))
    # an all-dark ping: no detections, empty cloud
    dark = SonarPing(np.zeros((1024, 512), np.uint8), oculus_bearings(512), 30.0 / 1024, ping_id=0)
    assert fe.callback(dark).shape == (0, 2)


def test_bit_stream_batches_leave_the_canvas_bitmap_clean(ctx, shipped_cfar):
    """The resident path clears the canvas bitmap as it expands it instead of a memset per batch (words with a list entry:
    extract_expand_words_kernel; frames above the point capacity, which get no list: extract_clean_queued_kernel).
    A batch with frames above the capacity, then other frames through the same scratch, a larger batch (the scratch
    grows), the dense pass in between (which leaves its bits behind) -- every batch equal to the oracle."""
    from sonar_slam_amd.pipeline import KeyframeBatch
    th, gh, tau = shipped_cfar.params["SOCA"]
    fe = FeatureExtraction(ctx)
    fe.Ntc, fe.Ngc, fe.Pfa, fe.rank, fe.alg, fe.threshold = 40, 10, 0.1, 10, "SOCA", 65
    fe.configure()
    frames = np.stack([synth.sonar_frame(seed=700 + s, n_blobs=(4 if s % 2 else 40)) for s in range(6)])
    fe.generate_map_xy(SonarPing(frames[0], oculus_bearings(512), 30.0 / 1024))
    want = []
    for f in frames:
        m = oracle.gate(f, oracle.cfar(f, "SOCA", th, gh, tau), 65)
        want.append(oracle.nonzero(oracle.remap_u8(m, fe.map_x, fe.map_y)))
    sizes = sorted(len(w) for w in want)
    small = (sizes[2] + sizes[3]) // 2                    # half of the frames are above it

    def batch(idx, cap, variant=0, env=None):
        kb = KeyframeBatch(ctx, fe.geometry, (th, gh, tau), "SOCA", 65, None, len(idx), max_points=cap, bit_masks=True)
        try:
            os.environ.update(env or {})
            ctx._check(ctx.lib.sfe_extract_set_tuning(ctx.handle, variant))
            kb.upload_frames(frames[idx])
            kb.run_cfar()
            kb.run_extract()
            ctx.sync()
            counts = kb.d_cnt.download(np.int32, len(idx))
            for k, j in enumerate(idx):
                assert counts[k] == len(want[j]), (idx, cap, j)
                if counts[k] <= cap:
                    assert np.array_equal(kb.points(k), oracle.px_to_m(want[j], fe.rows, fe.cols, fe.width, fe.height)), j
        finally:
            for key in env or {}:
                del os.environ[key]
            ctx._check(ctx.lib.sfe_extract_set_tuning(ctx.handle, 0))
            kb.free()

    big = sizes[-1] + 8
    batch([0, 1, 2, 3], big)            # (whatever ran before this test: the scratch may be new or dirty)
    batch([0, 1, 2, 3, 4, 5], small)    # three frames above the capacity: no list, cleared whole
    batch([5, 4, 3], big)               # the same scratch, other frames
    batch([1, 0], big, variant=1)       # dense pass: leaves its bits in the bitmap
    batch([2, 3, 4], big)
    batch(list(range(6)) * 3, big)      # more frames than before: the scratch grows
    batch([4, 1], big)
    # round 5: the record path (no canvas for the frames that fit) hands frames back to the canvas kernels when they exceed its
    # capacities -- the record list, the compact word array -- frame by frame inside one batch; and the canvas path alone
    words = sorted(len(np.unique(w[:, 0] * 4096 + w[:, 1] // 64)) for w in want)
    batch(list(range(6)), big, env={"SFE_EXTRACT_CAPW": str((words[2] + words[3]) // 2)})    # half of the frames handed back
    batch([3, 2, 1], big)
    batch(list(range(6)), big, env={"SFE_EXTRACT_REC_CAP": "64"})                             # every frame handed back
    batch(list(range(6)), small, env={"SFE_EXTRACT_CAPW": str(words[4] + 1)})                 # both kinds of overflow
    batch([0, 5, 2], big, variant=2)    # the canvas path for every frame
    batch([5, 0], big)


@pytest.mark.parametrize("variant", [0, 1, 2])
def test_bit_stream_extraction_equals_the_byte_mask_path(ctx, shipped_cfar, variant):
    """KeyframeBatch hands the detections to the extraction as bit streams (sfe_cfar_u8_bits_batch_dev ->
    sfe_extract_points_bits_batch_dev): same masks and the same points as the 0/1 byte path and the oracle,
    through the inverse map (variant 0: records merged in LDS for bit streams, the canvas bitmap for byte masks; variant 2: the
    canvas bitmap for both) and the dense pass (variant 1)."""
    from sonar_slam_amd.pipeline import KeyframeBatch
    th, gh, tau = shipped_cfar.params["SOCA"]
    fe = FeatureExtraction(ctx)
    fe.Ntc, fe.Ngc, fe.Pfa, fe.rank, fe.alg, fe.threshold = 40, 10, 0.1, 10, "SOCA", 65
    fe.configure()
    frames = np.stack([synth.sonar_frame(seed=300 + s) for s in range(11)])
    frames[3] = 0                                   # an empty frame between full ones
    fe.generate_map_xy(SonarPing(frames[0], oculus_bearings(512), 30.0 / 1024))
    mx, my = fe.map_x, fe.map_y
    res = []
    try:
        ctx._check(ctx.lib.sfe_extract_set_tuning(ctx.handle, variant))
        for bit_masks in (True, False):
            kb = KeyframeBatch(ctx, fe.geometry, (th, gh, tau), "SOCA", 65, None, len(frames), max_points=20000,
                               bit_masks=bit_masks)
            kb.upload_frames(frames)
            kb.run_cfar()
            kb.run_extract()
            ctx.sync()
            res.append(([kb.mask(j) for j in range(kb.n)], [kb.points(j) for j in range(kb.n)]))
            kb.free()
    finally:
        ctx._check(ctx.lib.sfe_extract_set_tuning(ctx.handle, 0))
    for j in range(len(frames)):
        want = oracle.gate(frames[j], oracle.cfar(frames[j], "SOCA", th, gh, tau), 65)
        assert np.array_equal(res[0][0][j], want) and np.array_equal(res[1][0][j], want)
        rc = oracle.nonzero(oracle.remap_u8(want, mx, my))
        pts = oracle.px_to_m(rc, frames.shape[1], mx.shape[1], fe.geometry.width, fe.geometry.height)
        assert np.array_equal(res[0][1][j], pts) and np.array_equal(res[1][1][j], pts)


def test_bit_stream_extraction_refuses_ragged_rows(ctx):
    g, mx, my, width, height = _geom(ctx, 100, 77, 0.1)
    d = ctx.alloc(4096)
    try:
        with pytest.raises(Exception, match="polar_cols"):
            ctx._check(ctx.lib.sfe_extract_points_bits_batch_dev(ctx.handle, g.handle, d.ptr, 1, 16, d.ptr, d.ptr))
    finally:
        d.free()


def test_batch_extraction_above_the_capacity_keeps_the_first_points(ctx, shipped_cfar):
    """sonarfe.h: d_counts[f] is the true count even above cap, and the first cap points are stored.  Frames below
    the capacity take the word-list expansion, frames above it the per-point kernel -- in one call."""
    n, ranges, beams = 6, 1024, 512
    g, mx, my, width, height = _geom(ctx, beams, ranges, 30.0 / 1024)
    th, gh, tau = shipped_cfar.params["SOCA"]
    frames = np.stack([synth.sonar_frame(seed=500 + s, n_blobs=(4 if s % 2 else 40)) for s in range(n)])
    masks = np.stack([oracle.gate(f, oracle.cfar(f, "SOCA", th, gh, tau), 65) for f in frames])
    want = [oracle.px_to_m(oracle.nonzero(oracle.remap_u8(m, mx, my)), ranges, mx.shape[1], width, height)
            for m in masks]
    sizes = sorted(len(w) for w in want)
    cap = (sizes[2] + sizes[3]) // 2                      # half of the frames are above it
    assert sizes[0] < cap < sizes[-1]
    d_mask, d_pts, d_cnt = ctx.alloc(masks.nbytes), ctx.alloc(n * cap * 16), ctx.alloc(n * 4)
    try:
        d_mask.upload(masks)
        d_pts.upload(np.full(n * cap * 2, -7.0))
        ctx._check(ctx.lib.sfe_extract_points_batch_dev(ctx.handle, g.handle, d_mask.ptr, n, cap, d_pts.ptr, d_cnt.ptr))
        ctx.sync()
        cnt = d_cnt.download(np.int32, n)
        pts = d_pts.download(np.float64, n * cap * 2).reshape(n, cap, 2)
    finally:
        for b in (d_mask, d_pts, d_cnt):
            b.free()
    for f in range(n):
        assert cnt[f] == len(want[f])
        k = min(cap, len(want[f]))
        assert np.array_equal(pts[f, :k], want[f][:k]), f
        assert np.all(pts[f, k:] == -7.0)                 # nothing is written behind a frame's points


def test_vis_image_with_the_jet_colour_map_in_one_pass(ctx):
    """feature_extraction.py:226-228: vis_img = cv2.applyColorMap(cv2.remap(img, ...), 2) published as bgr8.  The fused
    kernel (remap value -> BGR table -> packed 3-byte pixels) against table[oracle remap], per-stage call and the
    one-call ping path (with and without a store); odd canvas sizes exercise the tail of the 4-pixel groups."""
    from sonar_slam_amd import store as st
    from sonar_slam_amd.feature_extraction import COLORMAP_JET, FeatureExtraction, SonarPing, oculus_bearings
    lut = oracle.colormap_jet_lut()
    for rows, beams in ((256, 96), (300, 128), (512, 256)):
        fe = FeatureExtraction(ctx)
        fe.Ntc, fe.Ngc, fe.Pfa, fe.rank, fe.alg, fe.threshold, fe.skip = 40, 10, 0.1, 10, "SOCA", 65, 1
        fe.configure()
        img = synth.sonar_frame(seed=rows, rows=rows, cols=beams, n_blobs=10)
        ping = SonarPing(img, oculus_bearings(beams), 30.0 / rows)
        fe.generate_map_xy(ping)
        want = lut[oracle.remap_u8(img, fe.map_x, fe.map_y)]
        got = fe.geometry.remap(img, COLORMAP_JET)
        assert got.shape == want.shape and got.dtype == np.uint8 and np.array_equal(got, want)
        fe.make_vis_image, fe.vis_colormap = True, COLORMAP_JET
        pts = fe.callback(ping)
        assert np.array_equal(fe.feature_img, want)
        s = st.CloudStore(ctx, capacity_points=1 << 16, max_clouds=4)
        h, n, cloud = fe.callback_store(ping, s, publish=True)
        assert np.array_equal(fe.feature_img, want) and np.array_equal(cloud, np.asarray(pts, np.float32))
        fe.fused = False
        fe.callback(ping)
        assert np.array_equal(fe.feature_img, want)
        s.close()


def test_callback_equals_the_reference_callback_lines(ctx):
    """FeatureExtraction.callback (fused one-call path and per-stage path) == the body of the reference's callback run from its own
    lines (tests/golden/feature_callback.npz: feature_extraction.py:223-248 with the reference's CFAR class on its compiled cfar.cpp,
    its own maps, the oracle as cv2.remap / pcl) -- the filtered cloud of every case, the unfiltered fp64 points of the last"""
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "feature_callback.npz"))
    for i in range(int(z["n"])):
        for fused in (True, False):
            fe = FeatureExtraction(ctx)
            fe.Ntc, fe.Ngc, fe.Pfa, fe.rank, fe.alg, fe.threshold = 40, 10, 0.1, 10, "SOCA", int(z["threshold%d" % i])
            fe.resolution, fe.outlier_filter_radius, fe.outlier_filter_min_points, fe.skip = (
                float(z["resolution%d" % i]), float(z["radius%d" % i]), int(z["min_points%d" % i]), 1)
            fe.fused = fused
            fe.configure()
            pts = fe.callback(SonarPing(z["img%d" % i], z["bearings%d" % i], float(z["range_resolution%d" % i])))
            want = z["points%d" % i]
            # (float32: what the filters return and what the feature message carries, feature_extraction.py:182-185; with both
            #  filters off the reference's array is still float64 at this point and the one-call path hands back float32)
            assert np.array_equal(np.asarray(pts, np.float32), want.astype(np.float32)) and len(want) > 400, (i, fused, len(pts), len(want))
            if not fused and want.dtype == np.float64:
                assert np.asarray(pts).dtype == np.float64 and np.array_equal(pts, want)
