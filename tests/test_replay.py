"""Wire format (host) and the ROS-free front-end replay (GPU): a closed loop
render -> CFAR -> cloud -> ICP on a synthetic trajectory must beat its drifting odometry."""
import numpy as np
import pytest

from sonar_slam_amd import synth, wire
from sonar_slam_amd.pose2 import Pose2


def test_wire_format_is_create_cloud_xyz32_layout():
    pts = np.array([[1.5, -2.0], [3.25, 0.5]])
    data = wire.pack_features(pts)
    assert len(data) == 2 * wire.POINT_STEP
    xyz = np.frombuffer(data, "<f4").reshape(-1, 3)
    assert np.array_equal(xyz, np.array([[1.5, 0, -2.0], [3.25, 0, 0.5]], np.float32))   # (forward, 0, lateral)
    back = wire.unpack_features(data)
    assert np.array_equal(back, np.array([[1.5, 2.0], [3.25, -0.5]], np.float32))          # slam_ros.py:170: [x, -z]
    nan = wire.unpack_features(wire.pack_features(np.array([[np.nan, np.nan]])))
    assert wire.is_skipped(nan) and not wire.is_skipped(back)


def test_pose_chain_matches_matrices():
    a = Pose2(1, 2, 0.4)
    b = Pose2(-0.3, 0.1, -0.2)
    assert np.allclose(a.compose(b).between(a).matrix(), np.linalg.inv(b.matrix()), atol=1e-14)


@pytest.mark.gpu
def test_replay_closed_loop_beats_odometry(ctx):
    from sonar_slam_amd.feature_extraction import FeatureExtraction, SonarPing, oculus_bearings
    from sonar_slam_amd.replay import FrontEnd, replay
    world = synth.world_structure(seed=2, n=8000)
    true, dr = synth.trajectory(n=14, step=1.7, turn=0.05, seed=3)
    bearings = oculus_bearings(256)
    pings = [SonarPing(synth.render_ping(world, p, bearings, rows=512, seed=i), bearings, 30.0 / 512, ping_id=i)
             for i, p in enumerate(true)]
    fe = FeatureExtraction(ctx)
    fe.Ntc, fe.Ngc, fe.Pfa, fe.rank, fe.alg, fe.threshold = 40, 10, 0.1, 10, "SOCA", 65
    fe.resolution, fe.outlier_filter_radius, fe.outlier_filter_min_points, fe.skip = 0.5, 1.0, 5, 1
    fe.configure()
    # (odometry-started scan matches: whether the reference's global initialisation helps is its own business --
    # tests/test_global_init.py checks that flow against the oracle chain, not against ground truth)
    front = FrontEnd(ctx, keyframe_translation=1.5, keyframe_duration=0.5, ssm_initialization=False, nssm_enable=False)
    log, _, _ = replay(pings, np.arange(len(pings), dtype=float), dr, fe, front)
    assert len(log) >= 6 and log[0]["status"] == "PRIOR"
    ssm = [r for r in log[1:]]
    assert sum(r["status"] == "SUCCESS" for r in ssm) >= len(ssm) - 1, [r["status"] for r in ssm]
    # the front-end frame is sensor (forward, +bearing side): compare relative motion, first keyframe = origin
    t0 = Pose2(*true[0])
    est_err, dr_err = [], []
    for r in log:
        k = int(r["time"])
        want = t0.between(Pose2(*true[k]))
        got = Pose2(*dr[0]).between(Pose2(*r["pose"]))
        odo = Pose2(*dr[0]).between(Pose2(*dr[k]))
        est_err.append(np.hypot(got.x() - want.x(), got.y() - want.y()))
        dr_err.append(np.hypot(odo.x() - want.x(), odo.y() - want.y()))
    assert est_err[-1] < 0.5 * max(dr_err[-1], 0.3), (est_err, dr_err)


@pytest.mark.gpu
def test_compute_icp_with_cov_many_guesses_one_launch(ctx):
    """slam.py:325-387 on the batched path: 30 guesses (slam.yaml nssm.cov_samples) on one pair"""
    from sonar_slam_amd.replay import FrontEnd
    src, tgt, guess, truth = synth.scan_pair(seed=9, n_src=2000, n_tgt=2000)
    fe = FrontEnd(ctx)
    rng = np.random.default_rng(1)
    gx, gy, gt = synth.pose_of(guess)
    guesses = [Pose2(gx + dx, gy + dy, gt + dt) for dx, dy, dt in rng.normal(0, [0.2, 0.2, 0.02], (30, 3))]
    msg, odom, cov, samples = fe.compute_icp_with_cov(src, tgt, guesses)
    assert msg == "success" and len(samples) >= 25 and cov.shape == (3, 3)
    tx, ty, tt = synth.pose_of(truth)
    assert abs(odom.x() - tx) < 0.3 and abs(odom.y() - ty) < 0.3 and abs(odom.theta() - tt) < 0.03   # ICP on 20 % outliers
    assert np.linalg.det(cov) >= np.linalg.det(np.diag([0.1, 0.1, 0.01]) ** 2) - 1e-18
    # too few successes -> the reference's message
    msg2, *_ = fe.compute_icp_with_cov(src, tgt, guesses[:3])
    assert msg2 == "Too few samples for covariance computation"


def test_pose2_batch_equals_the_scalar_class_bit_for_bit():
    """chained.Pose2Batch (the SLAM node's gtsam.Pose2 algebra for many sessions at once) must give the bits of
    pose2.Pose2, element by element: the session batch's records are compared with the scalar front end's exactly."""
    from sonar_slam_amd.chained import Pose2Batch
    rng = np.random.default_rng(4)
    n = 257
    A = rng.normal(0, [20, 20, 2.0], (n, 3))
    B = rng.normal(0, [3, 3, 0.4], (n, 3))
    a, b = Pose2Batch(A[:, 0], A[:, 1], A[:, 2]), Pose2Batch(B[:, 0], B[:, 1], B[:, 2])
    sa, sb = [Pose2(*r) for r in A], [Pose2(*r) for r in B]
    chain = a.compose(b).between(a).inverse().compose(b.between(a))
    sc = [x.compose(y).between(x).inverse().compose(y.between(x)) for x, y in zip(sa, sb)]
    got = chain.xytheta()
    want = np.array([[p.x(), p.y(), p.theta()] for p in sc])
    assert np.array_equal(got, want)
    assert np.array_equal(chain.matrix32(), np.stack([p.matrix() for p in sc]).astype(np.float32))
    assert np.array_equal(chain.T6(), np.stack([p.matrix()[:2, :3].reshape(6) for p in sc]).astype(np.float32))
    # the renormalisation branch of Rot2 (|c^2 + s^2 - 1| > 1e-10) on a few elements only
    c, s = np.cos(A[:, 2]), np.sin(A[:, 2])
    c[::7] *= 1.0 + 1e-6
    pb = Pose2Batch(A[:, 0], A[:, 1], cs=(c, s))
    ps = [Pose2(x, y, _cs=(cc, ss)) for x, y, cc, ss in zip(A[:, 0], A[:, 1], c, s)]
    assert np.array_equal(pb.theta(), np.array([p.theta() for p in ps]))
    idx = np.array([3, 50, 200])
    sub = pb.take(idx)
    assert np.array_equal(sub.x, pb.x[idx]) and np.array_equal(sub.c, pb.c[idx])
    pb.put(idx, b.take(idx))
    assert np.array_equal(pb.x[idx], b.x[idx]) and np.array_equal(pb.s[idx], b.s[idx])


def test_oracle_chain_restates_one_session_on_the_cpu():
    """oracle/chain.py (test infrastructure: the CPU checker of the device-resident chained path) on a small synthetic
    session: every keyframe scan-matched against its predecessors, the chain closer to ground truth than the odometry it
    starts from, pose algebra equal to pose2.Pose2 (two independent restatements of gtsam's published Rot2 / Pose2)."""
    import oracle
    from types import SimpleNamespace
    from oracle import chain
    from sonar_slam_amd.CFAR import CFAR
    from sonar_slam_amd.feature_extraction import build_maps, oculus_bearings
    a, b = (1.5, -2.0, 0.7), (0.3, 0.4, -1.2)
    pa, pb = Pose2(*a), Pose2(*b)
    for got, want in ((chain.compose(chain.pose(*a), chain.pose(*b)), pa.compose(pb)),
                      (chain.between(chain.pose(*a), chain.pose(*b)), pa.between(pb)),
                      (chain.inverse(chain.pose(*a)), pa.inverse())):
        assert (got[0], got[1], chain.theta(got)) == (want.x(), want.y(), want.theta())
    rows, beams, K = 256, 128, 5
    bearings = oculus_bearings(beams)
    res, height, _, width, cols, mx, my = build_maps(bearings, 30.0 / rows, rows)
    fe = SimpleNamespace(map_x=mx, map_y=my, rows=rows, cols=cols, width=width, height=height)
    world = synth.world_structure(seed=2, n=5000)
    true, dr = synth.trajectory(n=K, step=1.7, turn=0.04, seed=11)
    det = CFAR(40, 10, 0.1, 10)
    clouds = []
    for k in range(K):
        img = synth.render_ping(world, true[k], bearings, rows=rows, seed=k)
        m, pts = chain.feature_cloud(img, det.params["SOCA"], "SOCA", 65, fe)
        assert m.sum() > 50 and pts.dtype == np.float32
        clouds.append(chain.slam_cloud(pts))
    recs = chain.run_session(clouds, dr, oracle.shipped_icp_params(precision=1), ssm_min_points=20)
    assert [r["status"] for r in recs][0] == "PRIOR" and sum(r["status"] == "SUCCESS" for r in recs) >= K - 2

    def end_err(p):
        want = Pose2(*true[0]).between(Pose2(*true[-1]))
        got = Pose2(*p[0]).between(Pose2(*p[-1]))
        return np.hypot(got.x() - want.x(), got.y() - want.y())
    assert end_err([r["pose"] for r in recs]) < end_err(dr) + 0.05
