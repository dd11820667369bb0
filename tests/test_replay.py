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
    front = FrontEnd(ctx, keyframe_translation=1.5, keyframe_duration=0.5)
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
