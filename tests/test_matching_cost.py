"""Global-initialisation matching cost (slam.py:461-570): oracle known answers on CPU, HIP path
against the oracle on the GPU.  Integer outputs: bit-exact."""
import numpy as np
import pytest

import oracle
from sonar_slam_amd import matching_cost as mc
from sonar_slam_amd import synth


def test_ellipse_element_matches_opencv_documented_values():
    # cv2.getStructuringElement(cv2.MORPH_ELLIPSE, (5, 5)) as printed in OpenCV's morphology tutorial
    want5 = np.array([[0, 0, 1, 0, 0], [1, 1, 1, 1, 1], [1, 1, 1, 1, 1], [1, 1, 1, 1, 1], [0, 0, 1, 0, 0]], np.uint8)
    assert np.array_equal(oracle.ellipse_kernel(2), want5)
    want7 = np.array([[0, 0, 0, 1, 0, 0, 0], [0, 1, 1, 1, 1, 1, 0], [1, 1, 1, 1, 1, 1, 1], [1, 1, 1, 1, 1, 1, 1],
                      [1, 1, 1, 1, 1, 1, 1], [0, 1, 1, 1, 1, 1, 0], [0, 0, 0, 1, 0, 0, 0]], np.uint8)
    assert np.array_equal(oracle.ellipse_kernel(3), want7)
    assert np.array_equal(oracle.ellipse_kernel(0), np.ones((1, 1), np.uint8))
    k = oracle.ellipse_kernel(10)       # the shipped case: point_noise / (point_noise / 10)
    assert k.shape == (21, 21) and np.array_equal(k, k[::-1]) and np.array_equal(k, k[:, ::-1])


def _numpy_reference(source_points, source_pose, target_points, target_pose, point_noise, X):
    """slam.py:507-562 transcribed with numpy; cv2.dilate replaced by the oracle's restatement."""
    xmin, ymin = np.min(target_points, axis=0) - 2 * point_noise
    xmax, ymax = np.max(target_points, axis=0) + 2 * point_noise
    resolution = point_noise / 10.0
    xs = np.arange(xmin, xmax, resolution)
    ys = np.arange(ymin, ymax, resolution)
    shape = (len(ys), len(xs))
    r = np.int32(np.round((target_points[:, 1] - ymin) / resolution))
    c = np.int32(np.round((target_points[:, 0] - xmin) / resolution))
    r = np.clip(r, 0, shape[0] - 1)
    c = np.clip(c, 0, shape[1] - 1)
    dilate_hs = int(np.ceil(point_noise / resolution))
    grid = oracle.cost_grid(r, c, shape[0], shape[1], dilate_hs)
    costs = []
    for x in X:
        sample_source_pose = source_pose.compose(mc.Pose2(*x))
        T = target_pose.between(sample_source_pose).matrix().astype(np.float32)
        # Keyframe.transform_points; the 2-term dot product written out so that no BLAS FMA decides a cell
        px, py = source_points[:, 0], source_points[:, 1]
        pts_x = (px * T[0, 0] + py * T[0, 1]) + T[0, 2]
        pts_y = (px * T[1, 0] + py * T[1, 1]) + T[1, 2]
        rr = np.int32(np.round((pts_y - ymin) / resolution))
        cc = np.int32(np.round((pts_x - xmin) / resolution))
        inside = (0 <= rr) & (rr < shape[0]) & (0 <= cc) & (cc < shape[1])
        costs.append(-int(np.sum(grid[rr[inside], cc[inside]] > 0)))
    return grid, np.array(costs, np.int32), (np.float32(xmin), np.float32(ymin), np.float32(resolution))


def _scene(seed, n=3000):
    src, tgt, guess, truth = synth.scan_pair(seed=seed, n_src=n, n_tgt=n)
    x, y, th = synth.pose_of(truth)
    rng = np.random.default_rng(seed)
    X = np.c_[rng.uniform(-1, 1, 40), rng.uniform(-1, 1, 40), rng.uniform(-0.2, 0.2, 40)]
    X[0] = 0
    return src, mc.Pose2(x, y, th), tgt, mc.Pose2(0, 0, 0), X


def test_pose2_algebra():
    a, b = mc.Pose2(1.0, -2.0, 0.3), mc.Pose2(-0.5, 0.25, -1.1)
    ab = a.compose(b)
    assert np.allclose(ab.matrix(), a.matrix() @ b.matrix(), atol=1e-15)
    assert np.allclose(a.between(b).matrix(), np.linalg.inv(a.matrix()) @ b.matrix(), atol=1e-15)
    assert abs(ab.theta() - (0.3 - 1.1)) < 1e-15


def test_oracle_cost_equals_numpy_transcription():
    src, sp, tgt, tp, X = _scene(3, n=1500)
    grid, want, (x0, y0, res) = _numpy_reference(src, sp, tgt, tp, 0.5, X)
    T6 = []
    for x in X:
        T = tp.between(sp.compose(mc.Pose2(*x))).matrix().astype(np.float32)
        T6.append([T[0, 0], T[0, 1], T[0, 2], T[1, 0], T[1, 1], T[1, 2]])
    got = oracle.matching_cost(grid, src, np.array(T6, np.float32), x0, y0, res)
    assert np.array_equal(got, want)
    assert want.min() < -100          # the candidates do overlap the target


@pytest.mark.gpu
@pytest.mark.parametrize("seed,noise", [(1, 0.5), (2, 0.5), (4, 0.25)])
def test_gpu_grid_and_costs_bit_exact(ctx, seed, noise):
    src, sp, tgt, tp, X = _scene(seed)
    grid, want, _ = _numpy_reference(src, sp, tgt, tp, noise, X)
    sub, samples = mc.get_matching_cost_subroutine1(src, sp, tgt, tp, np.eye(3), point_noise=noise, ctx=ctx)
    assert np.array_equal(sub.grid.download(), grid)
    got = sub.batch(X)
    assert np.array_equal(got, want)
    assert len(samples) == len(X) and all(s[3] == c for s, c in zip(samples, want))
    # the one-pose-per-call form shgo uses
    assert sub(X[5]) == want[5] and len(samples) == len(X) + 1
    p5 = sp.compose(mc.Pose2(*X[5]))
    assert np.allclose(samples[-1][:3], [p5.x(), p5.y(), p5.theta()])


@pytest.mark.gpu
def test_gpu_many_poses_per_launch_equal_the_oracle_and_the_small_kernel(ctx):
    """launches of >= 32 poses go through matching_cost_many_kernel (16 waves around one staged grid; the float64 body finds the
    cell from a multiplication and divides only next to a half-way point): the same costs as the oracle and as the kernel of the
    small launches, in both dtypes, for pose counts that fill waves and workgroups unevenly -- and on a grid whose resolution is a
    power of two with points ON the half-way points between cells (np.round's half-to-even decides every one of them)"""
    import os
    src, sp, tgt, tp, _ = _scene(3, n=2500)
    rng = np.random.default_rng(8)
    for f64 in (True, False):
        cloud = src.astype(np.float64) if f64 else src
        sub, _ = mc.get_matching_cost_subroutine1(cloud, sp, tgt, tp, None, point_noise=0.5, ctx=ctx)
        geo = sub.geometry
        grid = sub.grid.download()
        for P in (32, 33, 64, 65, 244, 300):
            X = np.c_[rng.uniform(-1, 1, P), rng.uniform(-1, 1, P), rng.uniform(-0.2, 0.2, P)]
            T6 = np.array([np.asarray(tp.between(sp.compose(mc.Pose2(*x))).matrix()).astype(np.float32)[:2].reshape(-1) for x in X])
            want = oracle.matching_cost(grid, src, T6, geo["xmin"], geo["ymin"], 0.05, f64_points=f64)
            got = sub.batch(X)
            os.environ["SFE_COST_NO_MANY"] = "1"
            try:
                small = sub.batch(X)
            finally:
                del os.environ["SFE_COST_NO_MANY"]
            assert np.array_equal(got, want) and np.array_equal(small, want) and (want < -200).any(), (f64, P)
        sub.grid.close()
    # ties: resolution 2^-4, the source points on k + 0.5 cells of the target's grid, the identity as one of the poses
    tgt2 = (np.round(tgt * 16) / 16).astype(np.float32)
    sub, _ = mc.get_matching_cost_subroutine1(np.zeros((4, 2)), tp, tgt2, tp, None, point_noise=0.625, ctx=ctx)
    geo = sub.geometry
    assert float(geo["resolution"]) == 0.0625
    k = rng.integers(0, min(geo["rows"], geo["cols"]) - 1, (3000, 2))
    half = (np.array([geo["xmin"], geo["ymin"]], np.float64) + (k + 0.5) * 0.0625)
    assert np.array_equal(half.astype(np.float32).astype(np.float64), half)
    sub.grid.close()
    sub, _ = mc.get_matching_cost_subroutine1(half, tp, tgt2, tp, None, point_noise=0.625, ctx=ctx)
    X = np.r_[np.zeros((1, 3)), np.c_[rng.uniform(-1, 1, 40), rng.uniform(-1, 1, 40), rng.uniform(-0.2, 0.2, 40)]]
    T6 = np.array([np.asarray(tp.between(tp.compose(mc.Pose2(*x))).matrix()).astype(np.float32)[:2].reshape(-1) for x in X])
    want = oracle.matching_cost(sub.grid.download(), half.astype(np.float32), T6, sub.geometry["xmin"], sub.geometry["ymin"], 0.0625,
                                f64_points=True)
    assert np.array_equal(sub.batch(X), want) and want[0] < 0
    sub.grid.close()


@pytest.mark.gpu
def test_gpu_large_grid_is_read_through_l2(ctx):
    """a grid too large for LDS (point_noise 0.05 m -> 5 mm cells)"""
    src, sp, tgt, tp, X = _scene(6, n=800)
    grid, want, _ = _numpy_reference(src, sp, tgt, tp, 0.05, X[:8])
    assert grid.size // 8 > 96 * 1024
    sub, _ = mc.get_matching_cost_subroutine1(src, sp, tgt, tp, None, point_noise=0.05, ctx=ctx)
    assert np.array_equal(sub.batch(X[:8]), want)
    assert np.array_equal(sub.grid.download(), grid)


@pytest.mark.gpu
def test_gpu_shgo_runs_on_the_subroutine(ctx):
    """the reference's call (slam.py:692-701) works unchanged on the drop-in subroutine"""
    from scipy.optimize import shgo
    src, sp, tgt, tp, _ = _scene(7, n=1000)
    sub, samples = mc.get_matching_cost_subroutine1(src, sp, tgt, tp, np.eye(3), point_noise=0.5, ctx=ctx)
    bounds = 5.0 * np.c_[-np.array([0.2, 0.2, 0.02]), np.array([0.2, 0.2, 0.02])]
    res = shgo(func=sub, bounds=bounds, n=16, iters=1, sampling_method="sobol",
               minimizer_kwargs={"options": {"ftol": 1e-4}})
    assert len(samples) >= 16 and res.fun <= sub([0, 0, 0])


# ---- fixtures produced by the REFERENCE's own get_matching_cost_subroutine1 (tests/golden/make_golden.py) ----
def _golden():
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "matching_cost.npz"))
    return g, mc.Pose2(*g["source_pose"]), mc.Pose2(0, 0, 0)


def test_oracle_reproduces_the_reference_functions_costs():
    """slam.py:461-570 executed from the reference source (cv2 / gtsam stood in, see make_golden.py):
    every numpy step of it -- bounds, np.arange lengths, rounding, clipping, the BLAS dot of
    Keyframe.transform_points, the inside test and the sum -- is the reference's own code."""
    g, sp, tp = _golden()
    grid, costs, (x0, y0, res) = _numpy_reference(g["src"], sp, g["tgt"], tp, float(g["point_noise"]), g["X"])
    assert np.array_equal(costs, g["costs"])
    assert np.array_equal(g["samples"][:, 3], g["costs"])
    # Keyframe.transform_points goes through this image's BLAS (fused multiply-adds): our written-out
    # float32 recipe differs from it by an ulp of the larger product (< 4e-6 m here), and no cell decision flips
    T = tp.between(sp.compose(mc.Pose2(*g["X"][7]))).matrix().astype(np.float32)
    mine = np.c_[(g["src"][:, 0] * T[0, 0] + g["src"][:, 1] * T[0, 1]) + T[0, 2],
                 (g["src"][:, 0] * T[1, 0] + g["src"][:, 1] * T[1, 1]) + T[1, 2]]
    ref = g["points_pose7"]
    assert ref.dtype == np.float32 and np.abs(mine - ref).max() <= 4e-6


@pytest.mark.gpu
def test_gpu_reproduces_the_reference_functions_costs(ctx):
    g, sp, tp = _golden()
    sub, samples = mc.get_matching_cost_subroutine1(g["src"], sp, g["tgt"], tp, np.eye(3),
                                                    point_noise=float(g["point_noise"]), ctx=ctx)
    assert np.array_equal(sub.batch(g["X"]), g["costs"])
    assert np.allclose(np.array(samples), g["samples"], rtol=0, atol=1e-12)
