"""CPU: the host-side pieces of bench.py (input synthesis, usable-core detection, the oracle-based
cpu_baseline leg on a tiny sample).  The GPU legs are exercised by the driver's bench run."""
import os
import sys
from types import SimpleNamespace

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from sonar_slam_amd.CFAR import CFAR  # noqa: E402
from sonar_slam_amd.feature_extraction import build_maps, oculus_bearings  # noqa: E402


def test_inputs_have_the_baseline_shapes():
    frames, srcs, tgts, guesses = bench.make_inputs(0, 3)
    assert frames.shape == (3, bench.ROWS, bench.COLS) and frames.dtype == np.uint8
    assert all(s.shape == (bench.N_PTS, 2) and s.dtype == np.float32 for s in srcs + tgts)
    assert guesses.shape == (3, 3, 3)
    f2, s2, _, _ = bench.make_inputs(1, 3)            # every rank gets its own jobs
    assert not np.array_equal(frames, f2) and not np.array_equal(srcs[0], s2[0])


def test_usable_cores_is_sane():
    n = bench.usable_cores()
    assert 1 <= n <= (os.cpu_count() or 1)


def test_cpu_baseline_leg_runs_and_reports_the_contract_fields():
    det = CFAR(40, 10, 0.1, 10)
    frames, srcs, tgts, guesses = bench.make_inputs(0, 2)
    res_, height_, rows_, width_, cols_, mx, my = build_maps(oculus_bearings(bench.COLS), 30.0 / bench.ROWS, bench.ROWS)
    fe = SimpleNamespace(map_x=mx, map_y=my, rows=rows_, cols=cols_, width=width_, height=height_)
    out = bench.cpu_baseline(frames, srcs, tgts, guesses, 1, det, fe, "reference")
    assert set(out) >= {"value", "unit", "cores", "kind", "sample"}
    assert out["kind"] == "port" and out["unit"] == "keyframes/s" and out["value"] > 0
    assert out["cores"] == bench.usable_cores()


class _OracleBackedBatch(object):
    """stands in for KeyframeBatch in the parity_check test: hands back what a correct device would hold"""

    def __init__(self, frames, det, fe):
        import oracle
        th, gh, tau = det.params["SOCA"]
        self.cap = 16384
        self.m, self.p, self.c = [], [], []
        for f in frames:
            m = oracle.gate(f, oracle.cfar(f, "SOCA", th, gh, tau), 65)
            pts = oracle.px_to_m(oracle.nonzero(oracle.remap_u8(m, fe.map_x, fe.map_y)), fe.rows, fe.cols, fe.width,
                                 fe.height)
            self.m.append(m)
            self.p.append(pts)
            self.c.append(oracle.remove_outlier(oracle.downsample(pts.astype(np.float32), 0.5), 1.0, 5))

    def mask(self, j):
        return self.m[j]

    def points(self, j):
        return self.p[j]

    def cloud(self, j):
        return self.c[j]


def test_parity_check_accepts_correct_outputs_and_raises_on_wrong_ones():
    import pytest

    import oracle
    det = CFAR(40, 10, 0.1, 10)
    frames, srcs, tgts, guesses = bench.make_inputs(0, 2)
    srcs, tgts = [s[:600] for s in srcs], [t[:600] for t in tgts]
    res_, height_, rows_, width_, cols_, mx, my = build_maps(oculus_bearings(bench.COLS), 30.0 / bench.ROWS, bench.ROWS)
    fe = SimpleNamespace(map_x=mx, map_y=my, rows=rows_, cols=cols_, width=width_, height=height_)
    kb = _OracleBackedBatch(frames, det, fe)
    out = [oracle.icp(s, t, g, oracle.shipped_icp_params(precision=1)) for s, t, g in zip(srcs, tgts, guesses)]
    res = {"status": np.array([o[0] for o in out]), "T": np.stack([o[1] for o in out]),
           "iters": np.array([o[2] for o in out])}
    pc = bench.parity_check(kb, res, frames, srcs, tgts, guesses, det, fe, "reference", True, 2)
    assert pc["jobs"] == 2 and pc["frames_bit_exact"] == 2 and pc["icp_max_pose_diff"] <= 1e-4
    bad = dict(res, T=res["T"].copy())
    bad["T"][1, 0, 2] += 1e-3
    with pytest.raises(AssertionError):
        bench.parity_check(kb, bad, frames, srcs, tgts, guesses, det, fe, "reference", True, 2)
    kb.p[0] = kb.p[0][::-1].copy()          # right points, wrong order
    with pytest.raises(AssertionError):
        bench.parity_check(kb, res, frames, srcs, tgts, guesses, det, fe, "reference", True, 2)


def test_leg_checker_accepts_the_oracle_and_raises_on_a_wrong_pose_status_or_iteration_count():
    """tools/bench_legs.check_against_oracle guards every ICP leg of the bench line: status and iteration count equal to
    the oracle with fp64 sums, pose within the tolerance, else AssertionError; the float-oracle statistic rides along."""
    import pytest

    import oracle
    sys.path.insert(0, os.path.join(bench.ROOT, "tools"))
    import bench_legs
    from sonar_slam_amd import synth
    jobs = [synth.scan_pair(seed=60 + i, n_src=300, n_tgt=280)[:3] for i in range(4)]
    ref = [oracle.icp(s, t, g, oracle.shipped_icp_params(precision=1)) for s, t, g in jobs]
    T = np.stack([r[1] for r in ref])
    st = np.array([r[0] for r in ref])
    it = np.array([r[2] for r in ref])
    ok = bench_legs.check_against_oracle("t", jobs, (T, st, it), {}, threads=2)
    assert ok["jobs"] == 4 and ok["max_pose_diff_vs_f64_oracle"] == 0.0 and ok["float_oracle_beyond_1e-4"].endswith("/ 4")
    bad = T.copy()
    bad[2, 1, 2] += 5e-6
    with pytest.raises(AssertionError, match="pose"):
        bench_legs.check_against_oracle("t", jobs, (bad, st, it), {}, threads=2)
    with pytest.raises(AssertionError, match="status/iterations"):
        bench_legs.check_against_oracle("t", jobs, (T, st, it + 1), {}, threads=2)
    with pytest.raises(AssertionError, match="status/iterations"):
        bench_legs.check_against_oracle("t", jobs, (T, st + 1, it), {}, threads=2)
    # the oracle is reentrant: the same jobs on several threads give the same answers (kd-tree on)
    again = bench_legs.oracle_many(jobs * 4, {}, threads=8)
    for j, (r64, r32) in enumerate(again):
        assert r64[0] == ref[j % 4][0] and r64[2] == ref[j % 4][2] and np.array_equal(r64[1], ref[j % 4][1])
