"""The reference's DEFAULT scan-matching flow (VERDICT r4 "what's missing" 2, 3): the shgo global initialisation in front of
every scan match (slam.py:77,89, :665-716, :922-973) and the loop-closure search over all older keyframes (slam.py:839-1087).

CPU part: the oracle's restatement (oracle/chain.py) behaves, and the one change the product makes to the shgo call -- the
points of a sampling stage scored in one launch through shgo's own `workers` hook -- gives the results of the one-by-one loop.
GPU part (-m gpu): replay.FrontEnd on host arrays == on store handles == chained.SessionBatch == the oracle chain, record by
record; the store-side kernels of the loop-closure search against numpy / the oracle one by one.
"""
from types import SimpleNamespace

import numpy as np
import pytest

import oracle
from oracle import chain
from sonar_slam_amd import synth
from sonar_slam_amd.pose2 import Pose2


def _session(K, rows=256, beams=128, step=1.7, turn=0.04, seed=11, world_seed=2, n_world=5000, start=(2.0, 0.0, 0.0)):
    """-> (pings [K x rows x beams], SLAM-node clouds, dr, true, bearings, fe namespace for the oracle)"""
    from sonar_slam_amd.CFAR import CFAR
    from sonar_slam_amd.feature_extraction import build_maps, oculus_bearings
    bearings = oculus_bearings(beams)
    res, height, _, width, cols, mx, my = build_maps(bearings, 30.0 / rows, rows)
    fe = SimpleNamespace(map_x=mx, map_y=my, rows=rows, cols=cols, width=width, height=height)
    world = synth.world_structure(seed=world_seed, n=n_world)
    true, dr = synth.trajectory(n=K, step=step, turn=turn, seed=seed, start=start)
    det = CFAR(40, 10, 0.1, 10)
    pings, clouds = [], []
    for k in range(K):
        img = synth.render_ping(world, true[k], bearings, rows=rows, seed=k)
        pings.append(img)
        clouds.append(chain.slam_cloud(chain.feature_cloud(img, det.params["SOCA"], "SOCA", 65, fe)[1]))
    return np.array(pings), clouds, dr, true, bearings, fe


def _rel_err(p, true):
    want = Pose2(*true[0]).between(Pose2(*true[-1]))
    got = Pose2(*p[0]).between(Pose2(*p[-1]))
    return float(np.hypot(got.x() - want.x(), got.y() - want.y()))


def test_pool_hook_of_shgo_gives_the_results_of_the_one_by_one_loop():
    """replay.FrontEnd.shgo hands shgo a `workers` map that scores a whole sampling stage at once; chain.run_shgo is the
    reference's call (slam.py:692-701) as written.  Same minimum, same value, same set of evaluated poses -- for the SSM
    parameters (50, 1) and the NSSM ones (100, 5)."""
    from sonar_slam_amd.replay import FrontEnd
    src, tgt, guess, truth = synth.scan_pair(seed=5, n_src=400, n_tgt=450)
    sp, tp = chain.pose(*synth.pose_of(guess)), chain.pose(0.0, 0.0, 0.0)
    for params, f64 in (((50, 1, 0.01), True), ((100, 5, 0.01), False)):
        sub_a, samples_a = chain.matching_cost_subroutine(src, sp, tgt, tp, 0.5, f64_points=f64)
        sub_b, samples_b = chain.matching_cost_subroutine(src, sp, tgt, tp, 0.5, f64_points=f64)
        sub_b.batch = lambda X: [sub_b(x) for x in X]
        pose_stds = np.array([[0.2, 0.2, 0.02]]).T
        bounds = 5.0 * np.c_[-pose_stds, pose_stds]
        ra = chain.run_shgo(sub_a, bounds, params)
        rb = FrontEnd.shgo(sub_b, bounds, params)
        assert ra.success and rb.success and np.array_equal(ra.x, rb.x) and ra.fun == rb.fun and ra.nfev == rb.nfev
        key = lambda s: tuple(map(tuple, np.array(sorted(map(tuple, s)))))
        assert key(samples_a) == key(samples_b) and len(samples_a) >= params[0]
        # the canonical order of the guesses derived from the samples does not depend on the evaluation order
        ga = chain.initial_transforms(samples_a, tp)
        gb = chain.initial_transforms(samples_b[::-1], tp)
        assert [tuple(g) for g in ga] == [tuple(g) for g in gb] and len(ga) >= 10
        # ... and the initialisation does its job: the pose it finds overlaps the target better than the guess it started from
        assert ra.fun <= sub_a(np.zeros(3)) and ra.fun < -100


def test_f64_and_f32_cost_bodies_follow_numpy():
    """oracle.matching_cost in both dtypes against the reference's numpy expressions (slam.py:554-565) evaluated here on a
    float64 array of float32 values and on the float32 array itself"""
    src, tgt, guess, _ = synth.scan_pair(seed=8, n_src=900, n_tgt=950)
    tgt = oracle.downsample(tgt, 0.5)
    xmin, ymin, resolution, rows, cols, hs = chain.grid_geometry(tgt, 0.5)
    r = np.clip(np.int32(np.round((tgt[:, 1] - ymin) / resolution)), 0, rows - 1)
    c = np.clip(np.int32(np.round((tgt[:, 0] - xmin) / resolution)), 0, cols - 1)
    grid = oracle.cost_grid(r, c, rows, cols, hs)
    rng = np.random.default_rng(3)
    base = synth.pose_of(guess)
    for f64 in (True, False):
        pts_in = src.astype(np.float64) if f64 else src
        for dx, dy, dt in rng.normal(0, [0.5, 0.5, 0.05], (12, 3)):
            T = synth.pose_matrix(base[0] + dx, base[1] + dy, base[2] + dt).astype(np.float32)
            points = pts_in.dot(T[:2, :2].T) + T[:2, 2]                      # Keyframe.transform_points
            rr = np.int32(np.round((points[:, 1] - ymin) / resolution))
            cc = np.int32(np.round((points[:, 0] - xmin) / resolution))
            inside = (0 <= rr) & (rr < rows) & (0 <= cc) & (cc < cols)
            want = -np.sum(grid[rr[inside], cc[inside]] > 0)
            got = oracle.matching_cost(grid, src, T[:2, :3].reshape(1, 6), xmin, ymin, resolution, f64_points=f64)[0]
            assert got == want and want < 0


def test_oracle_chain_with_the_global_initialisation():
    """the default flow on the CPU: every scan match preceded by shgo; the pose shgo finds stays inside its bounds (5 sigma
    of slam.yaml's odom_sigmas), ICP starts from it (the transforms differ from the odometry-started chain's where shgo moved
    the start) and the chain stays sane.  (Whether the initialisation HELPS is the reference's business: on these small
    clouds its grid-overlap maximum is sometimes 0.4 m off and the 5-iteration shipped chain does not come all the way back.)"""
    K = 6
    _, clouds, dr, true, _, _ = _session(K)
    prm = oracle.shipped_icp_params(precision=1)
    plain = chain.run_session(clouds, dr, prm, ssm_min_points=20)
    init = chain.run_session(clouds, dr, prm, ssm_min_points=20, initialization=True)
    assert sum(r["status"] == "SUCCESS" for r in init) >= K - 2
    for r in init[1:]:
        if r["status"] == "NOT_ENOUGH_POINTS":
            continue
        assert r["init_success"] and np.all(np.abs(r["init_x"]) <= 5.0 * np.array([0.2, 0.2, 0.02]) + 1e-12)
        assert r["init_cost"] <= 0
    e_plain, e_init = (_rel_err([r["pose"] for r in x], true) for x in (plain, init))
    assert e_init < 1.0 and e_plain < _rel_err(dr, true)
    moved = [r for r in init[1:] if r["status"] == "SUCCESS" and any(r["init_x"])]
    same = [k for k in range(1, K) if init[k]["status"] == "SUCCESS" and not any(init[k]["init_x"]) and
            all(not any(init[j].get("init_x", (0,))) for j in range(1, k))]
    assert moved, "shgo never moved the start: the test does not exercise the initialisation"
    for k in same:      # until shgo first moves a start the two chains are the same chain
        assert init[k]["transform"] == plain[k]["transform"]


def test_oracle_loop_closure_search_finds_the_revisit():
    """a trajectory that comes back to its start after 13 keyframes: the search of keyframe >= 13 selects old keyframes
    by field of view, refines the target key by overlap, runs the many-guess ICP and passes the gates; the loop transform
    agrees with ground truth to the accuracy of the scan matcher"""
    K = 15
    _, clouds, dr, true, _, _ = _session(K, step=1.7, turn=2 * np.pi / 13, seed=21, n_world=9000, start=(20.0, 0.0, 0.0))
    prm = oracle.shipped_icp_params(precision=1)
    recs = chain.run_session(clouds, dr, prm, ssm_min_points=20, initialization=True,
                             nssm=dict(min_points=30, mcd_random_state=0))
    searches = [r["nssm"] for r in recs[1:] if r.get("nssm") is not None]
    assert len(searches) == K - 8 + 1 and all("status" in n for n in searches)
    good = [n for n in searches if n["status"] == "SUCCESS"]
    assert good, [n["status"] for n in searches]
    for n in good:
        assert n["n_guesses"] >= 5 and n["n_converged"] >= 5 and n["cov"].shape == (3, 3)
        want = Pose2(*true[n["target_key"]]).between(Pose2(*true[n["source_key"]]))
        # (sensor frame = world frame up to the constant mirror of slam_ros.py:170: compare magnitudes of the motion.  The
        # search starts from the pose of the PREVIOUS callback's frame -- slam.py:854 reads self.current_frame, which
        # slam_ros.py:211 only moves on after the search -- so with every ping a keyframe it is one 1.7 m step behind, and
        # the shipped 5-iteration chain does not always make that up: the reference's behaviour, restated, not judged)
        got = n["transform"]
        assert abs(np.hypot(got[0], got[1]) - np.hypot(want.x(), want.y())) < 2.5
