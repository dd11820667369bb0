"""GPU parity at the shapes BASELINE.json names, against the CPU oracle with its exact kd-tree (same neighbours
as its brute-force scan, tests/test_oracle_pipeline.py; 0.3 s per 20k x 20k x 30-iteration job):

  configs[4]  hi-res many-to-one: ONE 20 000 x 20 000 pair, 30 guesses x 30 iterations, both error minimisers
              (the NSSM batch of slam.py:346-358 at slam.yaml's cov_samples = 30)
  configs[3]  the job farm: a few hundred independent 5 000 x 5 000 scan matches through farm.IcpFarm
              (persistent worker, shared-memory job blocks), every one against the oracle
  configs[1]  bench.py end to end with its own parity sample, and the 2-rank launch on one device

Tolerances.  Against the oracle with fp64 sums (same discrete decisions): identical status and iteration counts, pose
within 1e-6 (point-to-point) / 1e-4 (point-to-plane).  Against the oracle in float (PointMatcher<float>): 1e-4 m / rad
at 5 000 points; at 20 000 points the float oracle's OWN sums are the error (sequential float accumulation of 16 000
products: it sits 3e-4 from its fp64 version), so there the statement is that the HIP result is no farther from the
float oracle than the fp64 oracle is (+ 1e-6), and that spread is bounded at 1e-3."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle
from sonar_slam_amd import _lib, icp_config, pcl, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _pose_diff(Ta, Tb):
    a, b = synth.pose_of(Ta), synth.pose_of(Tb)
    return max(abs(a[0] - b[0]), abs(a[1] - b[1]), abs(np.arctan2(np.sin(a[2] - b[2]), np.cos(a[2] - b[2]))))


@pytest.fixture()
def kdtree():
    oracle.set_kdtree(1)
    yield
    oracle.set_kdtree(0)


@pytest.mark.parametrize("minimizer", [0, 1])
def test_config4_hires_many_to_one_30_guesses_30_iterations_vs_oracle(ctx, kdtree, minimizer):
    src, tgt, guess, truth = synth.scan_pair(seed=33, n_src=20000, n_tgt=20000)
    base = synth.pose_of(guess)
    rng = np.random.default_rng(9)
    guesses = [synth.pose_matrix(base[0] + dx, base[1] + dy, base[2] + dt).astype(np.float32)
               for dx, dy, dt in rng.normal(0, [0.3, 0.3, 0.05], (30, 3))]
    over = dict(minimizer=minimizer, max_iter=30, use_diff_checker=0)
    icp = pcl.ICP(ctx)
    icp.setParams(icp_config.shipped_params(**over))
    msgs, T, it = icp.compute_batch(src, tgt, guesses)
    worst_f = worst_d = 0.0
    for g, m, Tg, i in zip(guesses, msgs, T, it):
        st_d, T_d, it_d = oracle.icp(src, tgt, g, oracle.shipped_icp_params(precision=1, **over))
        st_f, T_f, it_f = oracle.icp(src, tgt, g, oracle.shipped_icp_params(precision=0, **over))
        assert m == oracle.ICP_STATUS_MESSAGES[st_d] and i == it_d == 30 and st_f == 0
        worst_d = max(worst_d, _pose_diff(Tg, T_d))
        worst_f = max(worst_f, _pose_diff(Tg, T_f))
        assert _pose_diff(Tg, T_f) <= _pose_diff(T_d, T_f) + 1e-6      # the float oracle's own accumulation noise
    assert worst_d < (1e-6 if minimizer == 0 else 1e-4), worst_d
    assert worst_f < 1e-3, worst_f
    assert min(_pose_diff(Tg, truth) for Tg in T) < 0.02


def test_config4_shipped_chain_on_the_hires_pair_vs_oracle(ctx, kdtree):
    """the chain config/icp.yaml ships (point-to-point, 40-iteration cap, differential stop) on the same pair"""
    src, tgt, guess, _ = synth.scan_pair(seed=34, n_src=20000, n_tgt=20000)
    base = synth.pose_of(guess)
    rng = np.random.default_rng(10)
    guesses = [synth.pose_matrix(base[0] + dx, base[1] + dy, base[2] + dt).astype(np.float32)
               for dx, dy, dt in rng.normal(0, [0.3, 0.3, 0.05], (30, 3))]
    icp = pcl.ICP(ctx)
    icp.setParams(icp_config.shipped_params())
    msgs, T, it = icp.compute_batch(src, tgt, guesses)
    for g, m, Tg, i in zip(guesses, msgs, T, it):
        st, To, ito = oracle.icp(src, tgt, g, oracle.shipped_icp_params(precision=1))
        assert m == oracle.ICP_STATUS_MESSAGES[st] and i == ito
        assert _pose_diff(Tg, To) < 1e-6


def test_config3_job_farm_vs_oracle(ctx, kdtree):
    """256 independent 5000 x 5000 point-to-plane-30 scan matches (64 distinct pairs, every job its own guess)
    through the persistent farm on this device, two batches on the same worker; each job against the oracle."""
    from sonar_slam_amd.farm import IcpFarm
    over = dict(minimizer=1, max_iter=30, use_diff_checker=0)
    p = icp_config.shipped_params(**over)
    pairs = [synth.scan_pair(seed=7000 + s, n_src=5000, n_tgt=5000) for s in range(64)]
    rng = np.random.default_rng(3)
    jobs = []
    for j in range(256):
        s, t, g, _ = pairs[j % 64]
        jobs.append((s, t, [g @ synth.pose_matrix(*rng.normal(0, [0.05, 0.05, 0.005])).astype(np.float32)]))
    with IcpFarm(p, devices=[ctx.device], chunk=96) as farm:
        first = farm.run(jobs[:64])
        out = farm.run(jobs)
        assert [w.proc.is_alive() for w in farm._workers] == [True]
    for a, b in zip(first, out[:64]):
        assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    worst = 0.0
    for (s, t, gs), (msgs, T, it) in zip(jobs, out):
        st_d, T_d, it_d = oracle.icp(s, t, gs[0], oracle.shipped_icp_params(precision=1, **over))
        assert msgs[0] == oracle.ICP_STATUS_MESSAGES[st_d] and it[0] == it_d
        worst = max(worst, _pose_diff(T[0], T_d))
    assert worst < 1e-4, worst


def test_resident_batch_refuses_a_frame_above_its_point_capacity(ctx):
    """the resident path stores only the first `cap` points of a frame; that must be an error, not a
    silently truncated cloud (the per-cloud API retries with a larger buffer instead)."""
    from sonar_slam_amd.CFAR import CFAR
    from sonar_slam_amd.feature_extraction import FeatureExtraction, SonarPing, oculus_bearings
    from sonar_slam_amd.pipeline import KeyframeBatch
    fe = FeatureExtraction(ctx)
    fe.Ntc, fe.Ngc, fe.Pfa, fe.rank, fe.alg, fe.threshold = 40, 10, 0.1, 10, "SOCA", 65
    fe.configure()
    img = synth.sonar_frame(seed=3, rows=256, cols=256, n_blobs=30)
    fe.generate_map_xy(SonarPing(img, oculus_bearings(256), 30.0 / 256))
    locs, _ = fe.extract(fe.detect(img))
    assert len(locs) > 64
    p = icp_config.shipped_params()
    kb = KeyframeBatch(ctx, fe.geometry, CFAR(40, 10, 0.1, 10).params["SOCA"], "SOCA", 65, p, 2, max_points=64)
    kb.upload_frames(np.stack([img, np.zeros_like(img)]))
    kb.run_cfar()
    kb.run_extract()
    kb.run_filter()
    with pytest.raises(_lib.SonarFEError, match="capacity"):
        kb.results()
    with pytest.raises(_lib.SonarFEError, match="capacity"):
        kb.points(0)
    with pytest.raises(_lib.SonarFEError, match="capacity"):
        kb.cloud(0)
    assert len(kb.points(1)) == 0 and len(kb.cloud(1)) == 0      # the empty frame next to it is fine
    kb.free()
    big = KeyframeBatch(ctx, fe.geometry, CFAR(40, 10, 0.1, 10).params["SOCA"], "SOCA", 65, p, 2, max_points=len(locs))
    big.upload_frames(np.stack([img, np.zeros_like(img)]))
    big.run_cfar()
    big.run_extract()
    big.run_filter()
    assert big.results()["counts"][0] == len(locs) and len(big.points(0)) == len(locs)
    big.free()


def _bench(extra, env=None, launcher=(), legs=False):
    cmd = [sys.executable] + list(launcher) + [os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1",
                                               "--batch", "16", "--cfar-frames", "64", "--cfar-launches", "2",
                                               "--no-cpu-baseline"] + ([] if legs else ["--no-legs"]) + extra
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run(cmd, cwd=ROOT, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout.decode()
    return json.loads(lines[0])


def test_bench_line_carries_its_own_parity_sample():
    out = _bench(["--gpus", "1", "--parity-jobs", "4"])
    pc = out["parity_check"]
    assert pc["jobs"] == 4 and pc["frames_bit_exact"] == 4 and pc["icp_max_pose_diff"] <= 1e-4
    assert out["config"]["max_points_per_frame"] <= out["config"]["points_capacity"]
    assert out["roofline"]["frac"] > 0 and out["n_gpus"] == 1


def test_bench_legs_carry_their_own_parity_samples():
    """the legs beyond the timed step (reduced sizes): the shipped chain on the timed pairs, the job shapes of the live
    system on the small-job tiers, BASELINE configs[4] with split jobs, the float-oracle statistic, streamed frames --
    each checks a sample against the oracle and the line is only printed when all of them hold"""
    out = _bench(["--gpus", "1", "--parity-jobs", "2", "--small-legs", "--no-latency", "--no-farm"], legs=True)
    assert out["reference_chain"]["parity"]["max_pose_diff_vs_f64_oracle"] <= 1e-6
    rs = out["real_size"]
    assert rs["ssm"]["all_jobs_on_1024_thread_workgroups"]["bit_identical_results"]
    assert rs["ssm"]["parity"]["jobs"] > 0 and rs["nssm"]["parity"]["jobs"] > 0
    c4 = out["configs4_hires"]
    assert c4["frames"]["frames_bit_exact_vs_oracle"] == 2 and c4["frames"]["roofline"]["frac"] > 0
    for chain in ("p2plane30", "reference_chain"):
        assert c4[chain]["parity"]["max_pose_diff_vs_f64_oracle"] <= 1e-5
        assert c4[chain]["one_batch_split_vs_unsplit_max_pose_diff"] <= 1e-5
    assert out["float_oracle"]["jobs"] >= 2
    sf = out["stream_frames"]
    assert sf["keyframes_per_s_streamed"] > 0 and 0.0 <= sf["overlap_fraction"] <= 1.0 and sf["frame_buffers"] == 3
    # round 5: the reference's default flow -- shgo in front of every scan match, and the loop-closure search over the store
    wi = out["chained"]["with_initialization"]
    assert wi["keyframes_per_s"] > 0 and wi["parity"]["max_pose_diff_vs_oracle_chain"] <= 1e-6 and wi["scipy_only"]["speculative_runs_redone"] == 0
    assert wi["scan_matches_equal_to_the_scipy_only_run"] > 0 and wi["scan_matches_replayed"] + wi["scan_matches_handed_to_scipy"] > 0
    lc = out["loop_closure"]
    assert lc["searches"] >= 1 and lc["parity"]["searches"] == lc["searches"]
    assert out["config"]["chained_with_initialization_keyframes_per_s"] == wi["keyframes_per_s"]
    assert out["roofline_filters"]["frac"] > 0 and out["roofline"]["filters_frac"] == out["roofline_filters"]["frac"]


def test_bench_two_ranks_on_one_device():
    """the driver's N > 1 launch (torch.distributed.run, one rank per GPU, gloo control plane) with both ranks
    pinned to this box's only device: barrier, max-over-ranks time and the aggregate value must come out"""
    port = 29500 + os.getpid() % 1000
    out = _bench(["--gpus", "2", "--parity-jobs", "2"], env={"SONARFE_BENCH_DEVICE": "0"},
                 launcher=["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                           "127.0.0.1", "--master-port", str(port)])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    assert abs(out["value"] - 2 * 16 * 2 / (out["ms_per_step"] * 2e-3)) < 1e-6 * out["value"]
    assert out["parity_check"]["frames_bit_exact"] == 2


@pytest.mark.gpu
def test_bench_eight_ranks_on_one_device():
    """the same launch at the target machine's width: eight ranks, all pinned to this box's only device,
    a tiny batch each -- rendezvous on 127.0.0.1, barrier, max-over-ranks time, the aggregate value over 8 ranks and rank 0's
    one JSON line; the legs that measure one device stay off in a multi-rank launch"""
    port = 29500 + (os.getpid() + 77) % 1000
    out = _bench(["--gpus", "8", "--batch", "8", "--parity-jobs", "2", "--steps", "2", "--warmup", "1"],
                 env={"SONARFE_BENCH_DEVICE": "0"},
                 launcher=["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr",
                           "127.0.0.1", "--master-port", str(port)])
    assert out["n_gpus"] == 8 and out["scaling"] == "weak" and out["value"] > 0 and out["steps"] == 2
    assert abs(out["value"] - 8 * 8 / (out["ms_per_step"] * 1e-3)) < 1e-6 * out["value"]
    assert out["parity_check"]["frames_bit_exact"] == 2
