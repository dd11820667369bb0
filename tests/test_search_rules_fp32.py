"""The pruning rules of the strip-sweep ICP search (sonar_slam_amd/csrc/sfe_icp_sweep.hip), checked in
float32 on the CPU with numpy doing exactly the kernel's arithmetic.  The GPU tests compare whole ICP runs with
the brute-force kernel; these fuzz each rule on its own, on inputs built to sit on the decision boundaries:

  d2(p, t) = fl(fl(dx*dx) + fl(dy*dy)),  dx = fl(px - tx), dy = fl(py - ty)          (dist2 of the kernels)

  1. x sweep: once fl(dx*dx) > bound for a point of an x-sorted run, no point further out has d2 <= bound;
  2. strips: if ylb = fl(smin - py) > 0 and fl(ylb*ylb) > bound, no point with y >= smin has d2 <= bound
     (smin = smallest y in the strips above; likewise below with smax);
  3. clearance: a query that stood at p0 with every target at d2 >= best0 > maxDist^2 and has moved to p with
     fl(|p - p0| * 1.00001) < fl(fl(sqrt(best0) * 0.99999) - fl(maxDist * 1.00001)) still has no target within
     maxDist: d2(p, t) > maxDist^2 for every t;
  4. clearance records of matched queries (stated in front of its test below);
  5. the union window of a wave's queries covers every lane's own window (round 3, stated at its test);
  6. the two-level arg-min of the exhaustive one-wave kernel (sfe_icp_tiny.h) picks the index a plain scan picks.
"""
import numpy as np

F = np.float32


def d2(px, py, tx, ty):
    dx = F(px) - np.asarray(tx, F)
    dy = F(py) - np.asarray(ty, F)
    return (dx * dx).astype(F) + (dy * dy).astype(F)          # every product and sum rounded to float32


def _clouds(rng, n, scale):
    kind = rng.integers(0, 4)
    if kind == 0:                                              # uniform
        return rng.uniform(-scale, scale, (n, 2)).astype(F)
    if kind == 1:                                              # on a coarse raster: many equal coordinates
        return (np.round(rng.uniform(-scale, scale, (n, 2)) * 4) / 4).astype(F)
    if kind == 2:                                              # huge offsets: few mantissa bits left for the detail
        return (rng.uniform(-scale, scale, (n, 2)) + 4096.0).astype(F)
    return (rng.normal(0, scale * 1e-3, (n, 2))).astype(F)    # tiny distances (products near the denormals)


def test_x_sweep_stop_rule_never_cuts_off_a_candidate():
    rng = np.random.default_rng(1)
    for case in range(400):
        t = _clouds(rng, int(rng.integers(2, 200)), 10.0)
        t = t[np.argsort(t[:, 0], kind="stable")]
        q = t[rng.integers(0, len(t))] + rng.normal(0, 0.3, 2).astype(F)
        px, py = F(q[0]), F(q[1])
        d = d2(px, py, t[:, 0], t[:, 1])
        e = ((px - t[:, 0]) * (px - t[:, 0])).astype(F)
        for bound in (d.min(), np.nextafter(d.min(), F(np.inf)), F(np.median(d)), F(0.0)):
            lo = int(np.searchsorted(t[:, 0], px, "left"))     # first point with x >= px
            # walk right: stop at the first point with e > bound; nothing beyond may have d2 <= bound
            r = lo
            while r < len(t) and e[r] <= bound:
                r += 1
            assert not (d[r:] <= bound).any()
            l = lo - 1
            while l >= 0 and e[l] <= bound:
                l -= 1
            assert not (d[:l + 1] <= bound).any()


def test_strip_bound_never_prunes_a_candidate():
    rng = np.random.default_rng(2)
    for case in range(400):
        t = _clouds(rng, int(rng.integers(2, 300)), 10.0)
        q = t[rng.integers(0, len(t))] + rng.normal(0, 1.0, 2).astype(F)
        px, py = F(q[0]), F(q[1])
        d = d2(px, py, t[:, 0], t[:, 1])
        ys = np.sort(t[:, 1])
        for smin in (ys[len(ys) // 2], ys[-1], np.nextafter(py, F(np.inf)), py):
            above = t[:, 1] >= smin
            ylb = F(smin) - py
            for bound in (d.min(), F(np.median(d)), F((ylb * ylb)), np.nextafter(F(ylb * ylb), F(-np.inf))):
                if ylb > 0 and F(ylb * ylb) > bound:            # the kernel skips these strips
                    assert not (d[above] <= bound).any()
        for smax in (ys[len(ys) // 2], ys[0], np.nextafter(py, F(-np.inf))):
            below = t[:, 1] <= smax
            ylb = py - F(smax)
            for bound in (d.min(), F(np.median(d)), F(ylb * ylb), np.nextafter(F(ylb * ylb), F(-np.inf))):
                if ylb > 0 and F(ylb * ylb) > bound:
                    assert not (d[below] <= bound).any()


def test_clearance_rule_never_hides_a_target_within_maxdist():
    rng = np.random.default_rng(3)
    skipped = 0
    for case in range(3000):
        md = F(rng.choice([0.3, 1.0, 10.0, 37.5]))
        r2m = F(md * md)
        w2 = F(r2m * F(1.1025))
        n = int(rng.integers(1, 60))
        ang = rng.uniform(0, 2 * np.pi, n)
        # targets just outside maxDist of the query's first position (some far away), optionally at a large offset
        rad = np.where(rng.random(n) < 0.7, md * (1 + rng.uniform(1e-7, 0.06, n)), md * rng.uniform(1.05, 3.0, n))
        off = F(rng.choice([0.0, 0.0, 300.0, 4096.0]))
        p0 = (rng.normal(0, 1, 2) + off).astype(F)
        t = (p0.astype(np.float64) + np.c_[rad * np.cos(ang), rad * np.sin(ang)]).astype(F)
        dd = d2(p0[0], p0[1], t[:, 0], t[:, 1])
        if not (dd > r2m).all():
            continue                                           # not a `none` query at p0
        best0 = F(min(dd.min(), w2))                           # what the search holds: min(nearest, window)
        clearance = F(F(np.sqrt(best0)) * F(0.99999)) - F(md * F(1.00001))
        # move towards the nearest target by a fraction of / about / more than the clearance
        j = int(np.argmin(dd))
        u = (t[j].astype(np.float64) - p0) / max(np.linalg.norm(t[j].astype(np.float64) - p0), 1e-30)
        for frac in (0.0, 0.5, 0.99, 0.9999, 1.0, 1.0001, 1.5):
            step = max(float(clearance), 0.0) * frac
            p = (p0.astype(np.float64) + u * step + rng.normal(0, 1e-7, 2)).astype(F)
            mx, my = F(p[0] - p0[0]), F(p[1] - p0[1])
            mv = F(np.sqrt(F(F(mx * mx) + F(my * my))))
            if F(mv * F(1.00001)) < clearance:                 # the kernel keeps the query `none` without a search
                skipped += 1
                assert (d2(p[0], p[1], t[:, 0], t[:, 1]) > r2m).all(), (case, frac, md, off)
    assert skipped > 2000                                      # the rule does fire on these inputs


# ---------------------------------------------------------------------------------------------------------------
# 4. clearance records (icp_sweep_kernel<..., REC>): a finished search at p0 = Tk x leaves
#        R = fl(fl(sqrt(min(runner-up d2, edge of the searched window))) * 0.99999), low 6 mantissa bits cleared,
#    edge = fl(fl(min(best, C) * M2) + mu2).  Later, at p = Tnow x, with dw = d2(p, old neighbour) and
#        mv = fl(fl(sigma * xr) + ft),  sigma = fl(fl(sigma_max(Anow - Ak)) * 1.001) * 1.0001,
#        xr = fl(|p - tnow| * 1.0001),  ft = fl(fl(|tnow - tk| * 1.0001) + 3e-5),   Ro = fl(R - mv):
#      A. fl(sqrt(dw) * 1.00001) < Ro  =>  the old neighbour is the strict nearest target of p;
#      B. dw > C and fl(sqrt(C) * 1.00001) < Ro  =>  no target of p is within C.
# ---------------------------------------------------------------------------------------------------------------
def _affine(T, x):
    """affine1 of the kernels: fl(fl(fl(a x) + fl(b y)) + c) for both rows of a float32 3x3"""
    px = F(F(F(T[0, 0] * x[0]) + F(T[0, 1] * x[1])) + T[0, 2])
    py = F(F(F(T[1, 0] * x[0]) + F(T[1, 1] * x[1])) + T[1, 2])
    return px, py


def _mat3_mul(a, b):
    """mat3_mul of the kernels: every product and sum rounded to float32, columns left to right"""
    c = np.zeros((3, 3), F)
    for i in range(3):
        for j in range(3):
            s = F(a[i, 0] * b[0, j])
            s = F(s + F(a[i, 1] * b[1, j]))
            s = F(s + F(a[i, 2] * b[2, j]))
            c[i, j] = s
    return c


def _step(theta, tx, ty):
    c, s = F(np.cos(theta)), F(np.sin(theta))
    return np.array([[c, -s, F(tx)], [s, c, F(ty)], [0, 0, 1]], F)


def _movement_bound(Tk, Tn, px, py):
    a0, a1 = F(Tn[0, 0] - Tk[0, 0]), F(Tn[0, 1] - Tk[0, 1])
    a3, a4 = F(Tn[1, 0] - Tk[1, 0]), F(Tn[1, 1] - Tk[1, 1])
    tx, ty = F(Tn[0, 2] - Tk[0, 2]), F(Tn[1, 2] - Tk[1, 2])
    f2 = F(F(F(a0 * a0) + F(a1 * a1)) + F(F(a3 * a3) + F(a4 * a4)))
    det = F(F(a0 * a4) - F(a1 * a3))
    disc = max(F(F(f2 * f2) - F(F(4.0) * F(det * det))), F(0.0))
    fa = F(F(np.sqrt(F(F(0.5) * F(f2 + F(np.sqrt(disc)))))) * F(1.001))
    ft = F(np.sqrt(F(F(tx * tx) + F(ty * ty))))
    mva, mvt = F(fa * F(1.0001)), F(F(ft * F(1.0001)) + F(3e-5))
    ux, uy = F(px - Tn[0, 2]), F(py - Tn[1, 2])
    xr = F(F(np.sqrt(F(F(ux * ux) + F(uy * uy)))) * F(1.0001))
    return F(F(mva * xr) + mvt)


def test_clearance_record_rule_never_keeps_a_wrong_neighbour():
    rng = np.random.default_rng(4)
    hits_a = hits_b = 0
    M2 = F(F(1.08) * F(1.08))
    for case in range(1500):
        n = int(rng.integers(2, 120))
        scale = float(rng.choice([0.2, 1.0, 5.0]))
        centre = rng.uniform(-25, 25, 2)
        t = (centre + rng.normal(0, scale, (n, 2))).astype(F)
        if case % 5 == 0:
            t = (np.round(t * 8) / 8).astype(F)                 # raster: equal distances, duplicates
        # a chain of small rigid steps like T_iter (T <- step * T), float32 all the way
        Tk = _mat3_mul(_step(rng.normal(0, 0.05), *rng.normal(0, 0.3, 2)), np.eye(3, dtype=F))
        x = np.linalg.solve(Tk.astype(np.float64), np.r_[centre + rng.normal(0, scale * 0.3, 2), 1.0])[:2].astype(F)
        p0 = _affine(Tk, x)
        d0 = d2(p0[0], p0[1], t[:, 0], t[:, 1])
        b = int(np.argmin(d0))                                  # (ties: lowest index, as the kernels resolve them)
        others = np.delete(d0, b)
        C = F(rng.choice([d0[b] * 4, d0[b] * 0.5, 0.0056, np.inf]))
        mu = F(rng.choice([0.0, 1e-3, 1e-2]))
        sb = min(d0[b], C) if np.isfinite(C) else d0[b]
        edge = F(F(F(sb) * M2) + F(mu * mu))
        R = F(F(np.sqrt(min(F(others.min()), edge))) * F(0.99999))
        R = np.frombuffer(np.uint32(np.frombuffer(F(R).tobytes(), np.uint32)[0] & ~np.uint32(63)).tobytes(), F)[0]
        gap = float(R) - float(np.sqrt(d0[b]))
        sC = F(F(np.sqrt(C)) * F(1.00001)) if np.isfinite(C) else F(np.inf)
        for frac in (0.0, 0.3, 0.9, 0.999, 1.0, 1.2, 3.0):
            # further steps whose total motion is about frac x the gap the record leaves
            amp = max(gap, 1e-4) * frac
            Tn = Tk
            for _ in range(int(rng.integers(1, 4))):
                Tn = _mat3_mul(_step(rng.normal(0, amp / 60.0), *rng.normal(0, amp / 3.0, 2)), Tn)
            px, py = _affine(Tn, x)
            dn = d2(px, py, t[:, 0], t[:, 1])
            dw = dn[b]
            Ro = F(R - _movement_bound(Tk, Tn, px, py))
            if F(F(np.sqrt(dw)) * F(1.00001)) < Ro:             # A: settled with (dw, b), no search
                hits_a += 1
                assert (np.delete(dn, b) > dw).all(), (case, frac)
            elif np.isfinite(C) and dw > C and sC < Ro:         # B: settled as "beyond the cap", no search
                hits_b += 1
                assert (dn > C).all(), (case, frac)
    assert hits_a > 1500 and hits_b > 50                        # both rules do fire on these inputs


def _strip_of(y, ylo, inv_g, ns):
    v = F(F(F(y) - F(ylo)) * F(inv_g))
    v = min(max(v, F(0.0)), F(ns - 1)) if v == v else F(0.0)
    return int(v)


def test_union_window_covers_every_lanes_own_window():
    """5. union scan (round 3): a wave of neighbouring queries, each holding a bound sb (squared), searches the points whose
    strip lies between strip_of(min py - r) and strip_of(max py + r) and whose x lies in [min px - r, max px + r], with
    r = fl(fl(sqrt(max sb) * 1.0001) + 1e-6).  Every target with d2(p, t) <= sb of ANY of its lanes must be inside:
    strips are assigned by the same strip_of (monotone in y), x by plain comparisons."""
    rng = np.random.default_rng(5)
    for case in range(300):
        scale = float(rng.choice([0.05, 1.0, 30.0]))
        t = _clouds(rng, int(rng.integers(20, 400)), scale)
        ns = int(rng.integers(1, 12))
        ylo, yhi = F(t[:, 1].min()), F(t[:, 1].max())
        inv_g = F(ns) / F(yhi - ylo) if yhi > ylo else F(0.0)
        if not np.isfinite(inv_g):
            inv_g = F(0.0)
        strip = np.array([_strip_of(y, ylo, inv_g, ns) for y in t[:, 1]])
        nq = int(rng.integers(1, 65))
        c = t[rng.integers(0, len(t))]
        q = (c + rng.normal(0, scale * 0.05, (nq, 2))).astype(F)
        d_all = np.stack([d2(q[i, 0], q[i, 1], t[:, 0], t[:, 1]) for i in range(nq)])
        # bounds on the decision boundary: exactly the distance of some target, one ulp above / below it, tiny, zero
        sb = np.array([rng.choice([np.sort(d_all[i])[min(len(t) - 1, int(rng.integers(0, 8)))],
                                   np.nextafter(d_all[i].min(), F(np.inf)), F(d_all[i].min()), F(1e-12), F(0.0)])
                       for i in range(nq)], F)
        r = F(F(np.sqrt(F(sb.max()))) * F(1.0001)) + F(1e-6)
        ux0, ux1 = F(q[:, 0].min()) - r, F(q[:, 0].max()) + r
        uy0, uy1 = F(q[:, 1].min()) - r, F(q[:, 1].max()) + r
        s_lo, s_hi = _strip_of(uy0, ylo, inv_g, ns), _strip_of(uy1, ylo, inv_g, ns)
        inside = (strip >= s_lo) & (strip <= s_hi) & ~(t[:, 0] < ux0) & (t[:, 0] <= ux1)
        for i in range(nq):
            need = d_all[i] <= sb[i]
            assert not (need & ~inside).any(), (case, i)


def test_two_level_arg_min_equals_the_plain_scan():
    """6. exhaustive one-wave kernel (sfe_icp_tiny.h): chunk minima (fminf over 16 points, strict '<' between chunks), then
    the first point of the winning chunk that attains the minimum -- the same index as a plain scan with a strict '<'
    in index order (lowest index on ties), NaN / inf points and padded chunks included."""
    rng = np.random.default_rng(6)
    for case in range(400):
        n = int(rng.integers(1, 200))
        t = _clouds(rng, n, 8.0)
        if case % 3 == 0 and n > 4:
            t[n // 2:] = t[:n - n // 2]                       # exact duplicates: ties
        if case % 5 == 0:
            t[rng.integers(0, n), rng.integers(0, 2)] = np.nan
        if case % 7 == 0:
            t[rng.integers(0, n)] = np.inf
        q = (t[rng.integers(0, n)] + rng.normal(0, 0.2, 2)).astype(F) if case % 11 else np.array([np.nan, 0.0], F)
        with np.errstate(invalid="ignore"):
            d = d2(q[0], q[1], t[:, 0], t[:, 1])
        best, bid = F(np.inf), -1
        for j in range(n):                                     # the reference scan
            if d[j] < best:
                best, bid = d[j], j
        npad = (n + 15) // 16 * 16
        dp = np.full(npad, np.inf, F)
        dp[:n] = d
        b2, bch = F(np.inf), -1
        for c0 in range(0, npad, 16):
            cmin = F(np.inf)
            for v in dp[c0:c0 + 16]:
                cmin = v if (v < cmin) else cmin               # fminf: a NaN never replaces the minimum
            if cmin < b2:
                b2, bch = cmin, c0
        id2 = -1
        if bch >= 0:
            for jj in range(15, -1, -1):
                if dp[bch + jj] == b2:
                    id2 = bch + jj
        assert id2 == bid and (bid < 0 or b2 == best), (case, bid, id2)
