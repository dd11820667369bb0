"""CPU restatement of one SLAM session's front end around the oracle's primitives.

TEST INFRASTRUCTURE ONLY (like everything under oracle/): the checker of the device-resident chained path
(sonar_slam_amd/chained.py, replay.FrontEnd on a store), never imported by the product.

Flow restated (reference file:line):
    FeatureExtraction.callback     feature_extraction.py:196-252   CFAR + gate, remap, nonzero, px->m, downsample, outliers
    publish_features / unpack      feature_extraction.py:175-193, slam_ros.py:169-170   float32 wire, [x, -z], float64 array
    SLAM_callback                  slam_ros.py:157-213              dead-reckoned pose of the new frame
    add_sequential_scan_matching   slam.py:716-832                  target = get_points(last 3), ICP, checks, overlap
    get_points / transform_points  slam.py:229-292, slam_objects.py:178-198
    compute_icp                    slam.py:294-323
    get_overlap                    slam.py:389-424

Pose algebra: gtsam.Pose2 / Rot2 from gtsam's published source (rotation kept as (c, s), products renormalised only
when |c^2 + s^2 - 1| > 1e-10), written here on plain tuples, independently of sonar_slam_amd.pose2.  gtsam itself is
un-vendored and absent: parity unpinned for that part.  Every ping is taken as a keyframe (the caller spaces them
beyond slam.yaml's keyframe_translation).
"""
import math

import numpy as np

import oracle

STATUS = ("PRIOR", "SUCCESS", "NOT_ENOUGH_POINTS", "NOT_CONVERGED", "LARGE_TRANSFORMATION", "NOT_ENOUGH_OVERLAP")


# ---- gtsam.Pose2 on tuples (x, y, c, s) ----
def pose(x, y, theta):
    return (float(x), float(y), math.cos(theta), math.sin(theta))


def _rot(c, s):
    n = c * c + s * s
    if abs(n - 1.0) > 1e-10:
        k = 1.0 / math.sqrt(n)
        c, s = c * k, s * k
    return c, s


def compose(a, b):
    c, s = _rot(a[2] * b[2] - a[3] * b[3], a[3] * b[2] + a[2] * b[3])
    return (a[0] + a[2] * b[0] - a[3] * b[1], a[1] + a[3] * b[0] + a[2] * b[1], c, s)


def inverse(a):
    c, s = _rot(a[2], -a[3])
    return (-(a[2] * a[0] + a[3] * a[1]), -(-a[3] * a[0] + a[2] * a[1]), c, s)


def between(a, b):
    return compose(inverse(a), b)


def theta(a):
    return math.atan2(a[3], a[2])


def matrix(a):
    return np.array([[a[2], -a[3], a[0]], [a[3], a[2], a[1]], [0.0, 0.0, 1.0]])


# ---- the feature node ----
def feature_cloud(img, cfar_params, alg, threshold, fe, resolution=0.5, radius=1.0, min_points=5):
    """-> (mask, N x 2 float32 (forward, lateral)) of one ping; ``fe``: object with map_x, map_y, rows, cols, width, height"""
    if alg == "OS":
        th, gh, k, tau = cfar_params
    else:
        (th, gh, tau), k = cfar_params, 0
    m = oracle.gate(img, oracle.cfar(img, alg, th, gh, tau, k), threshold)
    rc = oracle.nonzero(oracle.remap_u8(m, fe.map_x, fe.map_y))
    pts = oracle.px_to_m(rc, fe.rows, fe.cols, fe.width, fe.height)
    if len(pts) and resolution > 0:
        pts = oracle.downsample(pts.astype(np.float32), resolution)
    if min_points > 1 and len(pts) > 0:
        pts = oracle.remove_outlier(np.asarray(pts, np.float32), radius, min_points)
    return m, np.asarray(pts, np.float32).reshape(-1, 2)


def slam_cloud(points):
    """the feature message as the SLAM node holds it: float32 on the wire, [x, -z] (slam_ros.py:170)"""
    p = np.asarray(points, np.float32).reshape(-1, 2)
    return np.c_[p[:, 0], -1 * p[:, 1]]


# ---- the SLAM node ----
def run_session(clouds, dr, icp_params, point_resolution=0.5, point_noise=0.5, ssm_min_points=50,
                ssm_max_translation=3.0, ssm_max_rotation=np.deg2rad(30), ssm_target_frames=3):
    """clouds: per keyframe the SLAM node's cloud (``slam_cloud``); dr: [K x 3] dead-reckoned poses; icp_params:
    oracle.IcpParams.  -> list of records (status, sizes, ICP status / iterations / T, overlap, transform, pose)."""
    K = len(clouds)
    drp = [pose(*d) for d in dr]
    poses, recs = [], []
    for k in range(K):
        if k == 0:
            poses.append(drp[0])
            recs.append({"k": 0, "status": "PRIOR", "n_source": len(clouds[0]), "pose": (drp[0][0], drp[0][1], theta(drp[0]))})
            continue
        prev = poses[k - 1]
        cur = compose(prev, between(drp[k - 1], drp[k]))
        frames = list(range(k))[-ssm_target_frames:]
        Ts = [matrix(between(prev, poses[f])) for f in frames]
        target = oracle.get_points([clouds[f] for f in frames], Ts, point_resolution, f64_points=True)
        source = np.asarray(clouds[k], np.float32)
        rec = {"k": k, "n_source": len(source), "n_target": len(target)}
        new_pose = cur
        if len(source) < ssm_min_points or len(target) < ssm_min_points:
            rec["status"] = "NOT_ENOUGH_POINTS"
        else:
            initial = between(prev, cur)
            st, T, it = oracle.icp(source, target, matrix(initial).astype(np.float32), icp_params)
            rec.update(icp_status=st, iters=it, T=T)
            th32 = np.arctan2(T[1, 0], T[0, 0])
            est = pose(T[0, 2], T[1, 2], th32)
            status = "SUCCESS" if st == 0 else "NOT_CONVERGED"
            if status == "SUCCESS":
                d = between(initial, est)
                if float(np.hypot(d[0], d[1])) > ssm_max_translation or abs(theta(d)) > ssm_max_rotation:
                    status = "LARGE_TRANSFORMATION"
            if status == "SUCCESS":
                rec["overlap"] = oracle.overlap(source, target, matrix(est), point_noise, f64_points=True)
                if rec["overlap"] < ssm_min_points:
                    status = "NOT_ENOUGH_OVERLAP"
            rec["status"] = status
            rec["transform"] = (est[0], est[1], theta(est))
            if status == "SUCCESS":
                new_pose = compose(prev, est)
        poses.append(new_pose)
        rec["pose"] = (new_pose[0], new_pose[1], theta(new_pose))
        recs.append(rec)
    return recs
