"""Seeded synthetic inputs of the shapes BASELINE.json names (no rosbag / network here).

  * ``sonar_frame``: uint8 polar sonar image, rows = range bins, cols = beams: Rayleigh
    speckle + bright blobs/arcs + a range gain ramp (SURVEY 8d "Config 1").
  * ``scan_pair``: two in-plane clouds inside a 30 m x 130 deg fan: target = points on random
    polylines/arcs + N(0, 3 cm); source = the same structure seen from a moved pose with a
    fraction of outliers; guess = truth perturbed (SURVEY 8d "Config 2").
"""
import numpy as np


def sonar_frame(seed=0, rows=1024, cols=512, n_blobs=40, speckle=20.0):
    rng = np.random.default_rng(seed)
    img = rng.rayleigh(speckle, size=(rows, cols))
    img *= np.linspace(1.3, 0.7, rows)[:, None]  # range-dependent gain
    for _ in range(n_blobs):
        r0 = int(rng.integers(30, rows - 30))
        c0 = int(rng.integers(5, cols - 5))
        h = int(rng.integers(3, 10))
        w = int(rng.integers(3, 10))
        amp = float(rng.uniform(120, 255))
        if rng.random() < 0.3:  # arc: a wall seen across many beams
            w = int(rng.integers(30, 120))
            h = int(rng.integers(2, 5))
        rr = slice(max(r0 - h // 2, 0), min(r0 + h // 2 + 1, rows))
        cc = slice(max(c0 - w // 2, 0), min(c0 + w // 2 + 1, cols))
        img[rr, cc] += amp * rng.uniform(0.7, 1.0, size=(rr.stop - rr.start, cc.stop - cc.start))
    return np.clip(img, 0, 255).astype(np.uint8)


def _structure(rng, n, max_range=30.0, aperture_deg=130.0):
    """n points on random line segments / arcs inside the sonar fan, (x forward, y lateral)."""
    pts = []
    half = np.deg2rad(aperture_deg / 2)
    remaining = n
    while remaining > 0:
        m = int(min(remaining, rng.integers(40, 400)))
        r = rng.uniform(3.0, max_range - 1.0)
        th = rng.uniform(-half * 0.9, half * 0.9)
        c = np.array([r * np.cos(th), r * np.sin(th)])
        if rng.random() < 0.5:
            d = rng.uniform(0, np.pi)
            L = rng.uniform(1.0, 8.0)
            t = rng.uniform(-0.5, 0.5, m)
            seg = c[None, :] + (t * L)[:, None] * np.array([np.cos(d), np.sin(d)])[None, :]
        else:
            rad = rng.uniform(0.5, 4.0)
            a0 = rng.uniform(0, 2 * np.pi)
            a = a0 + rng.uniform(0, np.pi, m)
            seg = c[None, :] + rad * np.c_[np.cos(a), np.sin(a)]
        pts.append(seg)
        remaining -= m
    return np.concatenate(pts)[:n]


def pose_matrix(x, y, theta):
    c, s = np.cos(theta), np.sin(theta)
    return np.array([[c, -s, x], [s, c, y], [0, 0, 1]], np.float64)


def scan_pair(seed=0, n_src=5000, n_tgt=5000, noise=0.03, outliers=0.2, motion=(0.8, -0.3, 0.087),
              guess_error=(0.3, -0.2, 0.052)):
    """-> (source [n_src x 2] f32, target [n_tgt x 2] f32, guess 3x3 f32, truth 3x3 f64).

    truth maps source coordinates into the target frame: target ~ truth * source."""
    rng = np.random.default_rng(seed)
    world = _structure(rng, max(n_src, n_tgt))
    tgt = world[rng.permutation(len(world))[:n_tgt]] + rng.normal(0, noise, (n_tgt, 2))
    T = pose_matrix(*motion)
    Tinv = np.linalg.inv(T)
    base = world[rng.permutation(len(world))[:n_src]] + rng.normal(0, noise, (n_src, 2))
    src = base @ Tinv[:2, :2].T + Tinv[:2, 2]
    n_out = int(outliers * n_src)
    if n_out:
        idx = rng.permutation(n_src)[:n_out]
        r = rng.uniform(1.0, 30.0, n_out)
        th = rng.uniform(-1.1, 1.1, n_out)
        src[idx] = np.c_[r * np.cos(th), r * np.sin(th)]
    g = pose_matrix(motion[0] + guess_error[0], motion[1] + guess_error[1], motion[2] + guess_error[2])
    return src.astype(np.float32), tgt.astype(np.float32), g.astype(np.float32), T


def pose_of(T):
    """(x, y, theta) of a 3x3 transform, as SLAM.compute_icp parses it (slam.py:319-321)."""
    return float(T[0, 2]), float(T[1, 2]), float(np.arctan2(T[1, 0], T[0, 0]))


# ---- pose-consistent ping sequences for the replay harness (BASELINE configs[2]) --------------
def world_structure(seed=0, n=6000, extent=60.0):
    """Scatterers on random segments / arcs in a (extent x extent) m world, [x, y]."""
    rng = np.random.default_rng(seed)
    pts = []
    remaining = n
    while remaining > 0:
        m = int(min(remaining, rng.integers(60, 400)))
        c = rng.uniform(0.1 * extent, 0.9 * extent, 2) - np.array([0.0, extent / 2])
        if rng.random() < 0.6:
            d = rng.uniform(0, np.pi)
            L = rng.uniform(2.0, 12.0)
            t = rng.uniform(-0.5, 0.5, m)
            seg = c[None, :] + (t * L)[:, None] * np.array([np.cos(d), np.sin(d)])[None, :]
        else:
            rad = rng.uniform(1.0, 5.0)
            a = rng.uniform(0, 2 * np.pi, m)
            seg = c[None, :] + rad * np.c_[np.cos(a), np.sin(a)]
        pts.append(seg)
        remaining -= m
    return np.concatenate(pts)[:n]


def render_ping(world, pose, bearings, rows=512, max_range=30.0, seed=0, speckle=12.0, amp=230.0):
    """A uint8 polar ping (rows = range bins, cols = beams) of the scatterers ``world`` seen from
    ``pose`` = (x, y, theta).  Geometry is the inverse of generate_map_xy + px->m
    (feature_extraction.py:134-173,235-238): range bin = r / res, beam = bearing interpolated in
    ``bearings`` (1/100 deg), so an extracted point comes back at (forward, lateral) = sensor (x, y)
    once it went through the wire format (slam_ros.py:170)."""
    rng = np.random.default_rng(seed)
    cols = len(bearings)
    res = max_range / rows
    x, y, th = pose
    c, s = np.cos(th), np.sin(th)
    d = world - np.array([x, y])
    xs = c * d[:, 0] + s * d[:, 1]
    ys = -s * d[:, 0] + c * d[:, 1]
    r = np.hypot(xs, ys)
    b = np.arctan2(ys, xs)
    brad = np.asarray(bearings, np.float64) * np.pi / 18000.0
    col = np.interp(b, brad, np.arange(cols), left=-1, right=-1)
    ok = (r > 1.0) & (r < max_range - 0.5) & (col >= 0)
    img = rng.rayleigh(speckle, size=(rows, cols))
    rr = np.round(r[ok] / res).astype(int)
    cc = np.round(col[ok]).astype(int)
    for dr in (-1, 0, 1):
        for dc in (-1, 0, 1):
            a = amp * (1.0 if dr == 0 and dc == 0 else 0.55)
            np.maximum.at(img, (np.clip(rr + dr, 0, rows - 1), np.clip(cc + dc, 0, cols - 1)), a)
    return np.clip(img, 0, 255).astype(np.uint8)


def trajectory(n=12, step=1.6, turn=0.06, start=(2.0, 0.0, 0.0), seed=0, odom_sigma=(0.08, 0.08, 0.012)):
    """-> (true poses [n x 3], dead-reckoned poses [n x 3]): a gentle arc; the odometry increments
    carry noise so the dead-reckoned chain drifts."""
    rng = np.random.default_rng(seed)
    true = [np.array(start, np.float64)]
    dr = [np.array(start, np.float64)]
    for _ in range(n - 1):
        inc = np.array([step, 0.0, turn])
        noisy = inc + rng.normal(0, odom_sigma)
        for chain, u in ((true, inc), (dr, noisy)):
            x, y, th = chain[-1]
            chain.append(np.array([x + np.cos(th) * u[0] - np.sin(th) * u[1],
                                   y + np.sin(th) * u[0] + np.cos(th) * u[1], th + u[2]]))
    return np.array(true), np.array(dr)
