"""The neighbourhood rule of the resident radius filter's cell path (sonar_slam_amd/csrc/sfe_cloudfilter.hip,
cf_radius_filter_kernel), checked in float32 on the CPU: the octree path keys are computed with the kernel's
arithmetic (cf_cast_bbox_kernel's root cell, cf_path_key's `p > centre` descents), the cell level is chosen like
the kernel chooses it (deepest level <= 6 whose cells are >= 1.001 radii wide), and every pair of points within
the radius -- by the kernel's float32 distance test -- must lie in cells that differ by at most one in x and in y.
That is all the kernel assumes when it counts inside the 3 x 3 cells around a point."""
import numpy as np

F = np.float32
CELL_LEVELS = 6


def octree_header(p, max_size):
    mn, mx = p.min(0).astype(F), p.max(0).astype(F)
    rx, ry = F(mx[0] - mn[0]), F(mx[1] - mn[1])
    cx, cy = F(mn[0] + F(rx * F(0.5))), F(mn[1] + F(ry * F(0.5)))
    radius = F(max(rx, ry) * F(0.5))
    levels, r = 0, radius
    while not (float(r) * 2.0 <= float(max_size)) and levels < 31:
        r = F(r * F(0.5))
        levels += 1
    return cx, cy, radius, levels


def path_cells(p, cx, cy, radius, levels):
    """(ix, iy) of every point at every level 0..levels, by the kernel's descent"""
    n = len(p)
    ccx, ccy, r = np.full(n, cx, F), np.full(n, cy, F), F(radius)
    ix, iy = np.zeros(n, np.int64), np.zeros(n, np.int64)
    out = [(ix.copy(), iy.copy())]
    for _ in range(levels):
        bx, by = p[:, 0] > ccx, p[:, 1] > ccy
        ix, iy = ix * 2 + bx, iy * 2 + by
        hr = F(r * F(0.5))
        ccx = (ccx + np.where(bx, hr, -hr).astype(F)).astype(F)
        ccy = (ccy + np.where(by, hr, -hr).astype(F)).astype(F)
        r = hr
        out.append((ix.copy(), iy.copy()))
    return out


def test_points_within_the_radius_lie_in_adjacent_cells():
    rng = np.random.default_rng(11)
    checked = 0
    for case in range(300):
        n = int(rng.integers(2, 400))
        scale = float(rng.choice([0.5, 5.0, 30.0, 200.0]))
        off = float(rng.choice([0.0, 0.0, 1000.0]))
        p = (rng.uniform(-scale, scale, (n, 2)) + off).astype(F)
        if case % 3 == 0:                                  # points on walls: many pairs exactly one spacing apart
            p[:, case % 2] = np.round(p[:, case % 2] / 0.25) * 0.25
        res = F(rng.choice([0.05, 0.25, 0.5, 2.0]))
        cx, cy, radius, levels = octree_header(p, res)
        if levels > 8:
            continue                                       # deeper trees take the brute-force count
        radius_f = float(rng.choice([0.05, 0.3, 1.0, 1.0, 3.0, 25.0]))
        r2 = F(radius_f * radius_f)
        need = F(np.sqrt(r2) * F(1.001))
        width, lc = F(radius * F(2.0)), -1
        if need > 0 and width >= need:
            lc = 0
            while lc < CELL_LEVELS and lc < levels and F(width * F(0.5)) >= need:
                width = F(width * F(0.5))
                lc += 1
        if lc < 0:
            continue
        ix, iy = path_cells(p, cx, cy, radius, levels)[lc]
        dx = (p[:, None, 0] - p[None, :, 0]).astype(F)
        dy = (p[:, None, 1] - p[None, :, 1]).astype(F)
        near = ((dx * dx).astype(F) + (dy * dy).astype(F)) <= r2
        assert (np.abs(ix[:, None] - ix[None, :])[near] <= 1).all(), (case, lc, levels)
        assert (np.abs(iy[:, None] - iy[None, :])[near] <= 1).all(), (case, lc, levels)
        checked += int(near.sum())
    assert checked > 10000
