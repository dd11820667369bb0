"""GPU parity of the device-resident keyframe store: clouds that stay in HBM from
the feature extractor to the scan matcher give the results of the host-mediated path bit for bit, and those of the
oracle's restatement of slam_objects.py:178-198 / slam.py:229-292,294-323,389-424."""
import numpy as np
import pytest

import oracle
from sonar_slam_amd import _lib, icp_config, pcl, synth
from sonar_slam_amd import store as st
from sonar_slam_amd.pose2 import Pose2

pytestmark = pytest.mark.gpu


def _clouds(rng, sizes):
    return [np.c_[rng.uniform(1, 29, n), rng.uniform(-20, 20, n)].astype(np.float32) for n in sizes]


def test_put_read_meta_truncate(ctx):
    rng = np.random.default_rng(1)
    s = st.CloudStore(ctx, capacity_points=10000, max_clouds=16)
    clouds = _clouds(rng, (700, 0, 1, 1300))
    hs = [s.put(c, stamp=100 + i) for i, c in enumerate(clouds)]
    assert hs == [0, 1, 2, 3] and len(s) == 4
    stamps, off, cnt = s.meta()
    assert stamps.tolist() == [100, 101, 102, 103] and cnt.tolist() == [700, 0, 1, 1300]
    assert off.tolist() == [0, 700, 700, 701]
    for h, c in zip(hs, clouds):
        assert np.array_equal(s.read(h), c)
    s.truncate(2)
    assert len(s) == 2
    h = s.put(clouds[3], stamp=7)
    assert h == 2 and s.meta(2, 1)[1].tolist() == [700] and np.array_equal(s.read(2), clouds[3])
    # the pool is full: the cloud is refused (count -3), nothing is overwritten, and ICP refuses to run on it
    big = _clouds(rng, (9000,))[0]
    hb = s.put(big)
    assert s.counts([hb])[0] == -3 and np.array_equal(s.read(0), clouds[0]) and np.array_equal(s.read(2), clouds[3])
    with pytest.raises(_lib.SonarFEError, match="points"):
        s.icp(icp_config.shipped_params(), [(0, hb)], [np.eye(3)])
    with pytest.raises(_lib.SonarFEError):
        s.read(hb)
    # ... get_points and the overlap count refuse it as well instead of reading it as an empty cloud
    with pytest.raises(_lib.SonarFEError, match="not stored"):
        s.get_points([[0, hb]], [[st.pose_T6(np.eye(3))] * 2], 0.5)
    with pytest.raises(_lib.SonarFEError, match="not stored"):
        s.overlap([(0, hb)], [st.pose_T6(np.eye(3))], 0.5)
    # a cloud that did not fit owns no pool space: the fill level stayed behind the last cloud that did, so a smaller
    # one still goes in behind it, and dropping the failed slot leaves everything else where it was
    hc = s.put(clouds[2])
    assert hc == hb + 1 and s.counts([hc])[0] == 1 and s.meta(hc, 1)[1].tolist() == [2000]
    assert np.array_equal(s.read(hc), clouds[2])
    s.truncate(hb)
    assert s.put(clouds[0]) == hb and s.meta(hb, 1)[1].tolist() == [2000] and np.array_equal(s.read(hb), clouds[0])
    s.truncate(0)
    assert len(s) == 0 and s.put(clouds[0]) == 0 and s.meta()[1].tolist() == [0]
    for _ in range(15):
        s.put(clouds[2])
    with pytest.raises(_lib.SonarFEError, match="slots"):
        s.put(clouds[2])
    s.close()


@pytest.mark.parametrize("flags", [0, st.F32_POINTS])
def test_get_points_matches_the_oracle(ctx, flags):
    """transform (both numpy dtypes of the keyframe clouds) + concatenation in frame order + pcl.downsample, many
    targets per call, unused slots, empty clouds"""
    rng = np.random.default_rng(2)
    s = st.CloudStore(ctx, capacity_points=1 << 18, max_clouds=256)
    clouds = _clouds(rng, (900, 1500, 0, 400, 2500, 1, 3000))
    hs = [s.put(c) for c in clouds]
    poses = [Pose2(*q) for q in rng.normal(0, [3, 3, 0.3], (len(clouds), 3))]
    jobs = [([0, 1, 3], 4), ([4, -1, -1], 0), ([2, 5, -1], 1), ([6, 4, 1], 3), ([2, -1, -1], 0)]
    for res in (0.5, 0.25, 0.0):
        handles = np.array([f for f, _ in jobs], np.int32)
        T6 = np.array([[st.pose_T6(poses[ref].between(poses[max(k, 0)])) for k in f] for f, ref in jobs], np.float32)
        base = len(s)
        out = s.get_points(handles, T6, res, flags=flags)
        assert out.tolist() == list(range(base, base + len(jobs)))
        for (f, ref), h in zip(jobs, out):
            ks = [k for k in f if k >= 0]
            Ts = [poses[ref].between(poses[k]).matrix() for k in ks]
            moved = [oracle.transform_points(clouds[k], T, f64_points=not flags) for k, T in zip(ks, Ts)]
            allp = np.concatenate(moved) if moved else np.zeros((0, 2), np.float32)
            want = oracle.downsample(allp, res) if (len(allp) and res > 0) else allp
            assert np.array_equal(s.read(h), want), (f, ref, res)
        s.truncate(base)
    s.close()


def test_scan_match_and_overlap_over_handles_equal_the_host_calls_and_the_oracle(ctx):
    s = st.CloudStore(ctx, capacity_points=1 << 18, max_clouds=64)
    pairs, guesses, clouds = [], [], []
    for seed, (ns, nt) in enumerate([(300, 350), (900, 1200), (2500, 2500), (5000, 5000), (150, 4000)]):
        src, tgt, guess, _ = synth.scan_pair(seed=seed, n_src=ns, n_tgt=nt)
        pairs.append((s.put(src), s.put(tgt)))
        guesses.append(guess)
        clouds.append((src, tgt))
    for params in (icp_config.shipped_params(), icp_config.shipped_params(minimizer=1, max_iter=30, use_diff_checker=0)):
        T, status, iters = s.icp(params, pairs, guesses)
        icp = pcl.ICP(ctx)
        icp.setParams(params)
        msgs, Th, ith = icp.compute_pairs([c[0] for c in clouds], [c[1] for c in clouds], guesses)
        assert np.array_equal(T, Th) and np.array_equal(iters, ith)
        assert [_lib.ICP_STATUS_MESSAGES[int(x)] for x in status] == list(msgs)
        for (src, tgt), g, Tg, stg, itg in zip(clouds, guesses, T, status, iters):
            so, To, ito = oracle.icp(src, tgt, g, oracle.IcpParams(precision=1, **params.as_dict()))
            assert so == stg and ito == itg
            assert max(abs(a - b) for a, b in zip(synth.pose_of(Tg), synth.pose_of(To))) < 1e-6
    # get_overlap: the estimated pose as float32 matrix, both keyframe dtypes
    poses = [Pose2(*synth.pose_of(t)) for t in T]
    for flags in (0, st.F32_POINTS):
        got = s.overlap(pairs, [st.pose_T6(p) for p in poses], 0.5, flags=flags)
        want = [oracle.overlap(src, tgt, p.matrix(), 0.5, f64_points=not flags) for (src, tgt), p in zip(clouds, poses)]
        assert got.tolist() == want
        # ... and the host call chain of slam.py:389-424 on this library
        for (src, tgt), p, g in zip(clouds, poses, got):
            moved = oracle.transform_points(src, p.matrix(), f64_points=not flags)
            assert int(np.sum(pcl.match(tgt, moved, 1, 0.5, ctx=ctx)[0] != -1)) == g
    s.close()


def _fe(ctx):
    from sonar_slam_amd.feature_extraction import FeatureExtraction
    fe = FeatureExtraction(ctx)
    fe.Ntc, fe.Ngc, fe.Pfa, fe.rank, fe.alg, fe.threshold = 40, 10, 0.1, 10, "SOCA", 65
    fe.resolution, fe.outlier_filter_radius, fe.outlier_filter_min_points, fe.skip = 0.5, 1.0, 5, 1
    fe.configure()
    return fe


def test_ping_into_the_store_equals_the_ping_to_the_host(ctx):
    from sonar_slam_amd.feature_extraction import SonarPing, oculus_bearings
    fe = _fe(ctx)
    fe.make_vis_image = True
    s = st.CloudStore(ctx, capacity_points=1 << 16, max_clouds=8)
    for i, (rows, beams) in enumerate([(512, 256), (1024, 512), (300, 96)]):
        ping = SonarPing(synth.sonar_frame(seed=10 + i, rows=rows, cols=beams, n_blobs=25), oculus_bearings(beams),
                         30.0 / rows, ping_id=i)
        want = np.asarray(fe.callback(ping), np.float32)
        vis_want = fe.feature_img.copy()
        h, n, cloud = fe.callback_store(ping, s, stamp=i, publish=True)
        assert n == len(want) and np.array_equal(cloud, want) and np.array_equal(fe.feature_img, vis_want)
        assert np.array_equal(s.read(h), np.c_[want[:, 0], -1 * want[:, 1]])       # slam_ros.py:170
        h2, n2, none = fe.callback_store(ping, s, stamp=i)                           # nothing published: no copy
        assert none is None and n2 == n and np.array_equal(s.read(h2), s.read(h))
    assert s.meta()[0].tolist() == [0, 0, 1, 1, 2, 2]
    s.close()


def test_ping_into_a_full_store_is_an_error_and_leaves_the_store_usable(ctx):
    """sfe_feature_extract_ping_store reads the slot's own count; a pool that has no room returns
    SFE_ERR_CAP (no handle, no slot), later pings are not poisoned, and a store of another context is refused"""
    from sonar_slam_amd.feature_extraction import SonarPing, oculus_bearings
    fe = _fe(ctx)
    ping = SonarPing(synth.sonar_frame(seed=3, rows=512, cols=256, n_blobs=25), oculus_bearings(256), 30.0 / 512, ping_id=0)
    n = len(fe.callback(ping))
    assert n > 50
    s = st.CloudStore(ctx, capacity_points=2 * n + n // 2, max_clouds=8)
    h0, n0, _ = fe.callback_store(ping, s, stamp=0)
    h1, n1, _ = fe.callback_store(ping, s, stamp=1)
    assert (h0, h1) == (0, 1) and n0 == n1 == n
    with pytest.raises(_lib.SonarFEError, match="full"):
        fe.callback_store(ping, s, stamp=2)
    assert len(s) == 2 and s.meta()[2].tolist() == [n, n]
    with pytest.raises(_lib.SonarFEError, match="full"):          # and again: the first failure did not pin the fill level
        fe.callback_store(ping, s, stamp=3)
    small = _clouds(np.random.default_rng(0), (n // 4,))[0]
    assert s.put(small) == 2 and np.array_equal(s.read(2), small)
    s.truncate(1)
    h, m, _ = fe.callback_store(ping, s, stamp=4)                  # room again after a release
    assert h == 1 and m == n and np.array_equal(s.read(1), s.read(0))
    other = _lib.Context(0)
    try:
        s2 = st.CloudStore(other, capacity_points=1 << 14, max_clouds=4)
        with pytest.raises(_lib.SonarFEError):
            fe.callback_store(ping, s2, stamp=5)
        s2.close()
    finally:
        other.close()
    s.close()


def test_batch_of_filtered_clouds_into_the_store(ctx, shipped_cfar):
    """sfe_cloud_filter_batch_dev -> sfe_cloud_store_put_batch_dev, device to device"""
    from sonar_slam_amd.feature_extraction import Geometry, build_maps, oculus_bearings
    from sonar_slam_amd.pipeline import KeyframeBatch
    rows, beams, n = 512, 256, 12
    res, height, _, width, cols, mx, my = build_maps(oculus_bearings(beams), 30.0 / rows, rows)
    geom = Geometry(ctx, mx, my, (rows, beams), width, height)
    kb = KeyframeBatch(ctx, geom, shipped_cfar.params["SOCA"], "SOCA", 65, icp_config.shipped_params(), n, max_points=8192)
    kb.upload_frames(np.stack([synth.sonar_frame(seed=40 + i, rows=rows, cols=beams, n_blobs=20) for i in range(n)]))
    kb.run_cfar()
    kb.run_extract()
    kb.run_filter()
    s = st.CloudStore(ctx, capacity_points=1 << 17, max_clouds=64)
    s.put(np.ones((5, 2), np.float32))                       # not at the start of the pool
    hs = kb.store_clouds(s, stamps=np.arange(n) + 1000)
    assert hs.tolist() == list(range(1, n + 1)) and s.meta(1, n)[0].tolist() == (np.arange(n) + 1000).tolist()
    for j, h in enumerate(hs):
        c = kb.cloud(j)
        assert len(c) > 20 and np.array_equal(s.read(h), np.c_[c[:, 0], -1 * c[:, 1]])
    kb.free()
    s.close()


def test_front_end_on_the_store_equals_the_host_mediated_front_end(ctx):
    """replay.FrontEnd with device-resident keyframe clouds against the same flow through the wire format and numpy:
    every record of the log (status, sizes, ICP message, transform, overlap, pose) identical, and so are the factors."""
    from sonar_slam_amd.feature_extraction import SonarPing, oculus_bearings
    from sonar_slam_amd.replay import FrontEnd, replay
    world = synth.world_structure(seed=2, n=8000)
    true, dr = synth.trajectory(n=14, step=1.7, turn=0.05, seed=3)
    bearings = oculus_bearings(256)
    pings = [SonarPing(synth.render_ping(world, p, bearings, rows=512, seed=i), bearings, 30.0 / 512, ping_id=i)
             for i, p in enumerate(true)]
    stamps = np.arange(len(pings), dtype=float)
    logs = []
    for use_store in (False, True):
        s = st.CloudStore(ctx, capacity_points=1 << 18, max_clouds=256) if use_store else None
        front = FrontEnd(ctx, keyframe_translation=1.5, keyframe_duration=0.5, store=s, nssm_enable=False)
        log, _, _ = replay(pings, stamps, dr, _fe(ctx), front)
        logs.append((log, [f[:3] + ((f[3].x(), f[3].y(), f[3].theta()),) if len(f) > 3 else f[:2] for f in front.backend.factors]))
        if s is not None:
            # the store holds the keyframes and nothing else: targets and non-keyframes gave their slots back
            assert len(s) == len(front.keyframes)
            assert s.counts(range(len(s))).tolist() == [len(k.points) for k in front.keyframes]
            s.close()
    (a, fa), (b, fb) = logs
    assert len(a) == len(b) >= 6 and sum(r["status"] == "SUCCESS" for r in a) >= len(a) - 2
    for ra, rb in zip(a, b):
        assert ra == rb, (ra, rb)
    assert fa == fb


def _sessions(S, K, rows=512, beams=256, world_seed=2):
    """S trajectories through one scene, K pings each, spaced beyond the keyframe test's translation"""
    from sonar_slam_amd.feature_extraction import oculus_bearings
    world = synth.world_structure(seed=world_seed, n=8000)
    bearings = oculus_bearings(beams)
    frames = np.zeros((K, S, rows, beams), np.uint8)
    dr = np.zeros((S, K, 3))
    true = np.zeros((S, K, 3))
    for s in range(S):
        t, d = synth.trajectory(n=K, step=1.7, turn=0.03 + 0.01 * (s % 4), start=(2.0 + 0.5 * s, 0.3 * s - 1.0, 0.02 * s),
                                seed=100 + s)
        true[s], dr[s] = t, d
        for k in range(K):
            frames[k, s] = synth.render_ping(world, t[k], bearings, rows=rows, seed=1000 * s + k)
    return frames, dr, true, bearings


def test_sessions_in_lock_step_equal_the_front_end_and_the_oracle_chain(ctx, shipped_cfar):
    """chained.SessionBatch (S sessions, every cloud device-resident, host bookkeeping vectorised) against (a)
    replay.FrontEnd on a store, session by session: identical records; (b) the oracle's chain on the same pings:
    sizes / statuses / iteration counts / overlaps equal, transforms and poses <= 1e-6 (fp64-sum oracle)."""
    from oracle import chain
    from sonar_slam_amd import chained
    from sonar_slam_amd.feature_extraction import SonarPing
    from sonar_slam_amd.replay import FrontEnd, replay
    S, K, rows, beams = 5, 6, 512, 256
    frames, dr, true, bearings = _sessions(S, K, rows, beams)
    fe = _fe(ctx)
    fe.generate_map_xy(SonarPing(frames[0, 0], bearings, 30.0 / rows))
    params = icp_config.shipped_params()
    sb = chained.SessionBatch(ctx, fe.geometry, shipped_cfar.params["SOCA"], "SOCA", 65, params, S, K, dr)
    for k in range(K):
        sb.upload_frames(k, frames[k])
    recs = sb.run()
    assert len(recs) == K and len(sb.store) == S * K
    n_success = 0
    for s in range(S):
        # (a) the scalar front end on its own store
        store = st.CloudStore(ctx, capacity_points=1 << 18, max_clouds=64)
        front = FrontEnd(ctx, keyframe_translation=1.5, keyframe_duration=0.5, store=store, ssm_initialization=False, nssm_enable=False)
        pings = [SonarPing(frames[k, s], bearings, 30.0 / rows, ping_id=k) for k in range(K)]
        log, _, _ = replay(pings, np.arange(K, dtype=float), dr[s], fe, front)
        assert len(log) == K, "every ping is meant to be a keyframe"
        # (b) the oracle chain
        clouds = [chain.slam_cloud(chain.feature_cloud(frames[k, s], shipped_cfar.params["SOCA"], "SOCA", 65, fe)[1])
                  for k in range(K)]
        orc = chain.run_session(clouds, dr[s], oracle.IcpParams(precision=1, **params.as_dict()))
        for k in range(K):
            r, a, o = recs[k], log[k], orc[k]
            name = chained.STATUS_NAMES[r["status"][s]]
            assert name == a["status"] == o["status"], (s, k, name, a["status"], o["status"])
            assert r["n_source"][s] == a["n_source"] == o["n_source"]
            assert tuple(r["pose"][s]) == a["pose"]
            assert max(abs(x - y) for x, y in zip(a["pose"], o["pose"])) < 1e-6
            if k == 0:
                continue
            assert r["n_target"][s] == a["n_target"] == o["n_target"]
            if "transform" in a:
                assert tuple(r["transform"][s]) == a["transform"]
                assert r["icp_status"][s] == o["icp_status"] and r["iters"][s] == o["iters"]
                assert max(abs(x - y) for x, y in zip(a["transform"], o["transform"])) < 1e-6
            if "overlap" in a:
                assert r["overlap"][s] == a["overlap"] == o["overlap"]
            n_success += name == "SUCCESS"
        assert np.array_equal(store.counts(range(K)), sb.store.counts(sb.handles[s]))
        for k in (0, K - 1):
            assert np.array_equal(store.read(k), sb.store.read(sb.handles[s, k])) and np.array_equal(store.read(k), clouds[k])
        store.close()
    assert n_success >= S * (K - 1) - 3
    # the chain does its job: against ground truth it beats the odometry it started from
    est = np.stack([r["pose"] for r in recs], axis=1)            # [S x K x 3]
    def rel_err(p):
        e = []
        for s in range(S):
            want = Pose2(*true[s, 0]).between(Pose2(*true[s, -1]))
            got = Pose2(*p[s, 0]).between(Pose2(*p[s, -1]))
            e.append(np.hypot(got.x() - want.x(), got.y() - want.y()))
        return float(np.mean(e))
    assert rel_err(est) < rel_err(dr)
    sb.free()


def test_large_targets_and_many_guesses_over_handles(ctx):
    """get_points on more points than the LDS sort of the resident downsample holds (3 keyframes of 9 000 points: the
    global-memory sort path), and the NSSM shape of slam.py:346-358 over handles: 30 guesses on ONE (source, target) pair
    (the jobs share the target's preparation) -- equal to ICP.compute_batch on the host clouds."""
    rng = np.random.default_rng(9)
    s = st.CloudStore(ctx, capacity_points=1 << 18, max_clouds=64)
    clouds = _clouds(rng, (9000, 9000, 9000))
    hs = [s.put(c) for c in clouds]
    poses = [Pose2(0.0, 0.0, 0.0), Pose2(1.0, -0.5, 0.1), Pose2(2.2, 0.3, -0.07)]
    T6 = [[st.pose_T6(poses[0].between(p)) for p in poses]]
    h = s.get_points([hs], T6, 0.5)[0]
    want = oracle.get_points(clouds, [poses[0].between(p).matrix() for p in poses], 0.5)
    assert np.array_equal(s.read(h), want) and len(want) > 1000
    src, tgt, guess, _ = synth.scan_pair(seed=77, n_src=800, n_tgt=800)
    hsrc, htgt = s.put(src), s.put(tgt)
    base = synth.pose_of(guess)
    guesses = [synth.pose_matrix(base[0] + dx, base[1] + dy, base[2] + dt).astype(np.float32)
               for dx, dy, dt in rng.normal(0, [0.2, 0.2, 0.02], (30, 3))]
    T, status, iters = s.icp(icp_config.shipped_params(), [(hsrc, htgt)] * 30, guesses)
    icp = pcl.ICP(ctx)
    icp.setParams(icp_config.shipped_params())
    msgs, Tb, itb = icp.compute_batch(src, tgt, guesses)
    assert np.array_equal(T, Tb) and np.array_equal(iters, itb)
    assert [_lib.ICP_STATUS_MESSAGES[int(x)] for x in status] == list(msgs)
    s.close()


def test_selection_and_keys_belong_to_the_cloud_they_were_made_for(ctx):
    """(i) a selection is bound to the cloud AND its size -- after a truncate the slot number can come back as a
    larger cloud, and compact_selected on it used to read a stale (and too short) selection buffer; (ii) the keyed entry
    points refuse a cloud that was not built with keys instead of reading whatever the key pool holds at its offset."""
    from sonar_slam_amd import _lib
    from sonar_slam_amd.store import CloudStore, pose_T6
    rng = np.random.default_rng(3)
    st = CloudStore(ctx, capacity_points=1 << 16, max_clouds=64)
    a = st.put(rng.uniform(-20, 20, (500, 2)).astype(np.float32))
    b = st.put(rng.uniform(-20, 20, (700, 2)).astype(np.float32))
    T = np.stack([pose_T6(np.eye(3)), pose_T6(np.eye(3))])
    with pytest.raises(_lib.SonarFEError, match="no keys"):
        st.read_keys(a)                                            # a plain cloud, no keyed cloud built yet
    h = st.get_points_keys([a, b], T, [0, 1], 0.5)
    n_h = int(st.counts([h])[0])
    assert len(st.read_keys(h)) == n_h
    with pytest.raises(_lib.SonarFEError, match="no keys"):
        st.read_keys(a)                                            # still a plain cloud: the pool exists now, its keys do not
    st.set_selection(h, np.ones(n_h, np.uint8))
    st.truncate(2)                                                 # the keyed cloud is dropped ...
    big = st.get_points_keys([a, b, a], np.stack([pose_T6(np.eye(3))] * 3), [0, 1, 2], 0.25)   # ... its slot comes back, larger
    assert big == h and int(st.counts([big])[0]) > n_h
    with pytest.raises(_lib.SonarFEError, match="no selection"):
        st.compact_selected(big)
    st.set_selection(big, np.ones(int(st.counts([big])[0]), np.uint8))
    c = st.compact_selected(big)
    assert int(st.counts([c])[0]) == int(st.counts([big])[0]) and len(st.read_keys(c)) == int(st.counts([c])[0])
    st.close()
