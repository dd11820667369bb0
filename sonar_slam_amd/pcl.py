"""Drop-in for the reference's pybind11 module ``bruce_slam.pcl``.

Same names and call signatures as ``PYBIND11_MODULE(pcl, m)``
(bruce_slam/src/bruce_slam/cpp/pcl.cpp:176-214), backed by the HIP kernels of libsonarfe.so:

    match(ref, in, knn, max_dist) -> (ids int32 [knn x N], dists float32 [knn x N])  pcl.cpp:161
    remove_outlier(points, radius, min_points) -> points'                           pcl.cpp:54
    class ICP: loadFromYaml(path), compute(source, target, guess) -> (message, T),
               getCovariance()                                                      pcl.cpp:184-213
    downsample(points[, descriptors], resolution) -> points'[, descriptors']       pcl.cpp:128-159
    density_filter(points[, descriptors], knn, min_density, max_density)           pcl.cpp:76-126 (raises like
                   the reference does; knn_density / max_density_filter are the working pieces)

Extension (not in the reference): ``ICP.compute_batch(source, target, guesses)`` runs the
many-guesses-one-pair loop of SLAM.compute_icp_with_cov (slam.py:346-358) in one launch.
"""
import ctypes as _C
import os as _os

import numpy as _np

from . import _lib as _L
from . import icp_config as _cfg


def _cloud(a, name):
    a = _np.ascontiguousarray(a, _np.float32)
    if a.ndim != 2 or a.shape[1] not in (2, 3):
        raise TypeError("%s: expected an N x 2 array, got shape %r" % (name, a.shape))
    if a.shape[1] == 3:
        raise NotImplementedError(
            "%s: N x 3 clouds (3-D ICP) are outside the sonar front-end path; the SLAM node only "
            "passes N x 2 in-plane points (slam.py:309-310)" % name)
    return a


def match(ref, pts, knn, max_dist, ctx=None):
    """KDTreeMatcher replacement (pcl.cpp:161-174): the knn exact nearest reference points of every query in
    ascending (d2, index) order, squared distances, -1 / inf beyond max_dist.
    -> (ids int32 [knn x N], dists float32 [knn x N]) like the pybind IntMatrix / Matrix pair."""
    knn = int(knn)
    if knn < 1:
        raise RuntimeError("pcl.match: knn must be >= 1, got %d" % knn)
    ctx = ctx or _L.default_context()
    ref = _cloud(ref, "match(ref)")
    pts = _cloud(pts, "match(in)")
    if knn > len(ref):
        # libnabo: "Requesting more points (k) than available in cloud" (a std::runtime_error -> RuntimeError through
        # pybind); knn_density mirrors the same throw
        raise RuntimeError("pcl.match: knn = %d exceeds the %d points of the reference cloud (libnabo throws)"
                           % (knn, len(ref)))
    ids = _np.full((knn, len(pts)), -1, _np.int32)
    d2 = _np.full((knn, len(pts)), _np.inf, _np.float32)
    if len(pts):
        with ctx.lock:
            if knn == 1:    # the SLAM node's only use (slam.py:418): one pass over the reference
                ctx._check(ctx.lib.sfe_match(ctx.handle, _L.ptr(ref, _C.c_float), len(ref),
                                             _L.ptr(pts, _C.c_float), len(pts), float(max_dist),
                                             _L.ptr(ids, _C.c_int32), _L.ptr(d2, _C.c_float)))
            else:
                ctx._check(ctx.lib.sfe_match_knn(ctx.handle, _L.ptr(ref, _C.c_float), len(ref),
                                                 _L.ptr(pts, _C.c_float), len(pts), knn, float(max_dist),
                                                 _L.ptr(ids, _C.c_int32), _L.ptr(d2, _C.c_float)))
    return ids, d2


def remove_outlier(points, radius, min_points, ctx=None):
    """PCL RadiusOutlierRemoval replacement (pcl.cpp:54-74)."""
    ctx = ctx or _L.default_context()
    pts = _cloud(points, "remove_outlier(points)")
    out = _np.zeros_like(pts)
    n = _C.c_int(0)
    if len(pts):
        with ctx.lock:
            ctx._check(ctx.lib.sfe_remove_outlier(ctx.handle, _L.ptr(pts, _C.c_float), len(pts),
                                                  float(radius), int(min_points),
                                                  _L.ptr(out, _C.c_float), _C.byref(n)))
    return out[:n.value].copy()


def downsample(points, *args, **kw):
    """libpointmatcher OctreeGridDataPointsFilter replacement (pcl.cpp:128-159), both overloads:
    ``downsample(points, resolution)`` -> points' and
    ``downsample(points, descriptors, resolution)`` -> (points', descriptors')."""
    ctx = kw.pop("ctx", None) or _L.default_context()
    if len(args) == 1:
        desc, resolution = None, args[0]
    elif len(args) == 2:
        desc, resolution = _np.ascontiguousarray(args[0], _np.float32), args[1]
    else:
        raise TypeError("downsample(points, resolution) or downsample(points, descriptors, resolution)")
    pts = _cloud(points, "downsample(points)")
    if desc is not None and len(desc) != len(pts):
        raise TypeError("downsample: %d descriptors for %d points" % (len(desc), len(pts)))
    if len(pts) == 0:  # pcl.cpp:130-131,145-146
        return pts if desc is None else (pts, desc)
    out = _np.zeros_like(pts)
    idx = _np.zeros(len(pts), _np.int32) if desc is not None else None   # (only the descriptor overload needs the indices)
    n = _C.c_int(0)
    with ctx.lock:
        ctx._check(ctx.lib.sfe_downsample(ctx.handle, _L.ptr(pts, _C.c_float), len(pts), float(resolution),
                                          _L.ptr(out, _C.c_float), _L.ptr(idx, _C.c_int32) if idx is not None else None,
                                          _C.byref(n)))
    out = out[:n.value].copy()
    return out if desc is None else (out, desc[idx[:n.value]].copy())


class _GlibcRand(object):
    """glibc's rand() (TYPE_3 additive feedback generator, r[i] = r[i-3] + r[i-31]), which is what the std::rand()
    of libpointmatcher's MaxDensityDataPointsFilter resolves to on the reference's platform (Ubuntu / ROS noetic).
    Process-wide state like the C library's: seeded with 1 until somebody calls srand."""
    RAND_MAX = 2147483647

    def __init__(self, seed=1):
        self.srand(seed)

    def srand(self, seed):
        seed = int(seed) & 0xFFFFFFFF or 1
        r = [0] * 34
        r[0] = seed
        for i in range(1, 31):
            hi, lo = divmod(r[i - 1], 127773)                 # 16807 * x mod (2^31 - 1), Schrage
            w = 16807 * lo - 2836 * hi
            r[i] = w + 2147483647 if w < 0 else w
        for i in range(31, 34):
            r[i] = r[i - 31]
        self._r = r
        for _ in range(310):
            self._step()

    def _step(self):
        r = self._r
        o = (r[-31] + r[-3]) & 0xFFFFFFFF
        r.append(o)
        del r[0]
        return o

    def rand(self):
        return self._step() >> 1


_rand = _GlibcRand(1)


def srand(seed):
    """std::srand for the random thinning of max_density_filter (extension: the reference never seeds)."""
    _rand.srand(seed)


def knn_density(points, knn, ctx=None):
    """The ``densities`` descriptor pcl.density_filter computes first (pcl.cpp:81-88): libpointmatcher
    SurfaceNormalDataPointsFilter{knn, keepDensities}: knn / ((4/3) pi r^3), r = the largest distance of one of the
    point's knn nearest points (itself included) from their mean.  -> float32 [N]"""
    ctx = ctx or _L.default_context()
    pts = _cloud(points, "knn_density(points)")
    knn = int(knn)
    if knn < 1:
        raise RuntimeError("knn_density: knn must be >= 1")
    if knn > len(pts):
        raise RuntimeError("Requesting more points than available in cloud")   # libnabo
    dens = _np.zeros(len(pts), _np.float32)
    if len(pts):
        with ctx.lock:
            ctx._check(ctx.lib.sfe_knn_density(ctx.handle, _L.ptr(pts, _C.c_float), len(pts), knn,
                                               _L.ptr(dens, _C.c_float)))
    return dens


def max_density_filter(points, *args, **kw):
    """What the body of the reference's ``density_filter`` does once its stray ``minDensity`` parameter is taken out
    (pcl.cpp:76-126): densities from the knn nearest neighbours (``knn_density``), then libpointmatcher's
    MaxDensityDataPointsFilter: a point with density <= max_density stays; a denser one stays with probability
    max_density / density (scaled by 1 - nbSaturated / nbPoints -- an integer division, i.e. 1 unless every point is
    saturated -- for the points at the maximum density), drawn with std::rand() in input order.
    ``max_density_filter(points, knn, max_density)`` -> points', or with descriptors
    ``max_density_filter(points, descriptors, knn, max_density)`` -> (points', descriptors').  Extension."""
    ctx = kw.pop("ctx", None)
    if len(args) == 2:
        desc, (knn, max_density) = None, args
    elif len(args) == 3:
        desc, knn, max_density = _np.ascontiguousarray(args[0], _np.float32), args[1], args[2]
    else:
        raise TypeError("max_density_filter(points, [descriptors,] knn, max_density)")
    pts = _cloud(points, "max_density_filter(points)")
    if len(pts) == 0:
        return pts if desc is None else (pts, desc)
    dens = knn_density(pts, knn, ctx=ctx)
    max_density = _np.float32(max_density)
    last = dens.max()
    n_sat = int((dens == last).sum())
    keep = _np.ones(len(pts), bool)
    for i in _np.nonzero(dens > max_density)[0]:        # input order; only the dense points draw a number
        r = _np.float32(_rand.rand()) / _np.float32(_GlibcRand.RAND_MAX)
        accept = _np.float32(max_density / dens[i])
        if dens[i] == last:
            accept = _np.float32(accept * _np.float32(1 - n_sat // len(pts)))
        keep[i] = r < accept
    return pts[keep].copy() if desc is None else (pts[keep].copy(), desc[keep].copy())


def density_filter(points, *args):
    """``pcl.density_filter(points[, descriptors], knn, min_density, max_density)`` (pcl.cpp:76-126, both overloads).
    The reference hands libpointmatcher's MaxDensityDataPointsFilter a ``minDensity`` parameter that filter does not
    have (pcl.cpp:90-92, 115-117); its registrar rejects parameters a module never reads, so on a non-empty cloud the
    reference call ends in ``InvalidParameter`` -- a RuntimeError in Python -- which is presumably why its only call
    site is commented out (feature_extraction.py:246).  This drop-in behaves the same: empty clouds come back
    unchanged (:78-79, :103-104), anything else raises with libpointmatcher's message.  ``max_density_filter`` is the
    working form."""
    if len(args) not in (3, 4):
        raise TypeError("density_filter(points, knn, min_density, max_density) or "
                        "density_filter(points, descriptors, knn, min_density, max_density)")
    pts = _np.asarray(points)
    if pts.ndim == 2 and pts.shape[0] == 0:
        return pts if len(args) == 3 else (pts, _np.asarray(args[0]))
    raise RuntimeError("Parameter minDensity for module MaxDensityDataPointsFilter was set but is not used")


class ICP(object):
    """``pcl.ICP`` (PM::ICP bound at pcl.cpp:184-213)."""

    def __init__(self, ctx=None):
        self._ctx = ctx
        # PM::ICP() starts without a chain (pcl.cpp:185); the SLAM node always calls loadFromYaml next (slam.py:99)
        self.params = None

    def loadFromYaml(self, filename, strict=None):
        """pcl.cpp:187-197.  A file that cannot be opened makes the reference print a message and fall back to
        PM::ICP::setDefault(): random sub-sampling of the reading, surface-normal sampling of the reference and a
        3-D point-to-plane chain -- not reproducible and not what any bruce_slam launch file intends.  Default
        (``strict`` True): an error instead of a silently different chain (deviations).

        ``strict=False``, or ``SONARFE_YAML_FALLBACK=shipped`` in the environment when ``strict`` is not given: behave
        like the reference AT THE CALL SITE (slam_ros.py:124-125 sees no exception) -- print the reference's exact
        line (pcl.cpp:192) and carry on with a default chain; the default installed here is the chain of the shipped
        ``config/icp.yaml`` (``icp_config.shipped_params``), since libpointmatcher's ``setDefault`` chain is 3-D and
        random and has no counterpart on this path."""
        if strict is None:
            strict = _os.environ.get("SONARFE_YAML_FALLBACK", "").strip().lower() != "shipped"
        try:
            with open(filename, "r") as fh:
                text = fh.read()
        except (IOError, OSError) as e:
            if not strict:
                print("Failed to load %s. Use default configuration." % filename)      # pcl.cpp:192, verbatim
                self.params = _cfg.shipped_params()
                return
            raise RuntimeError("ICP.loadFromYaml: cannot open %s (%s).  The reference would print 'Failed to load ... Use "
                               "default configuration.' and run libpointmatcher's setDefault() chain (random sampling + "
                               "surface normals), which this front end does not provide: fix the path, install a "
                               "chain with setParams(), or pass strict=False / set SONARFE_YAML_FALLBACK=shipped to "
                               "carry on with the chain of the shipped config/icp.yaml" % (filename, e))
        self.params = _cfg.parse_icp_yaml(text)

    def setParams(self, params):
        """Extension: install an ``IcpParams`` directly."""
        self.params = params

    def _chain(self):
        if self.params is None:
            raise RuntimeError("ICP.compute before loadFromYaml / setParams: the chain is empty (PM::ICP() has no "
                               "default chain either, pcl.cpp:185)")
        return self.params

    @staticmethod
    def _guess(guess):
        g = _np.asarray(guess, _np.float32)
        if g.shape == (4, 4):
            raise NotImplementedError("4 x 4 guesses (3-D ICP) are outside the sonar front-end path")
        if g.shape != (3, 3):
            raise TypeError("ICP.compute: guess must be 3 x 3, got %r" % (g.shape,))
        return _np.ascontiguousarray(g)

    def compute(self, source, target, guess):
        """-> (message, T): ("success", T 3x3 float32) or (ConvergenceError text, guess)."""
        msgs, Ts, _ = self.compute_batch(source, target, [guess])
        return msgs[0], Ts[0]

    def compute_batch(self, source, target, guesses):
        """Many initial guesses on one cloud pair in one launch.
        -> (messages [n], T [n x 3 x 3] float32, iterations [n])"""
        params = self._chain()
        ctx = self._ctx or _L.default_context()
        src = _cloud(source, "ICP.compute(source)")
        tgt = _cloud(target, "ICP.compute(target)")
        g = _np.ascontiguousarray(_np.stack([self._guess(x) for x in guesses]), _np.float32)
        n = len(g)
        if len(src) == 0 or len(tgt) == 0:
            raise RuntimeError("ICP.compute: empty point cloud (libpointmatcher would throw)")
        T = _np.zeros((n, 3, 3), _np.float32)
        st = _np.zeros(n, _np.int32)
        it = _np.zeros(n, _np.int32)
        with ctx.lock:
            ctx._check(ctx.lib.sfe_icp_compute_guesses(
                ctx.handle, _C.byref(params), _L.ptr(src, _C.c_float), len(src),
                _L.ptr(tgt, _C.c_float), len(tgt), _L.ptr(g, _C.c_float), n, _L.ptr(T, _C.c_float),
                _L.ptr(st, _C.c_int32), _L.ptr(it, _C.c_int32)))
        msgs = [_L.ICP_STATUS_MESSAGES.get(int(s), "ICP failure %d" % s) for s in st]
        return msgs, T, it

    def compute_pairs(self, sources, targets, guesses):
        """Many independent (source, target, guess) scan matches in ONE launch (extension; the job
        farm's unit).  -> (messages [n], T [n x 3 x 3] float32, iterations [n])"""
        params = self._chain()
        ctx = self._ctx or _L.default_context()
        n = len(sources)
        if not (len(targets) == n and len(guesses) == n):
            raise TypeError("compute_pairs: %d sources, %d targets, %d guesses" % (n, len(targets), len(guesses)))
        if n == 0:
            return [], _np.zeros((0, 3, 3), _np.float32), _np.zeros(0, _np.int32)
        srcs = [_cloud(s, "ICP.compute_pairs(source)") for s in sources]
        tgts = [_cloud(t, "ICP.compute_pairs(target)") for t in targets]
        if any(len(s) == 0 for s in srcs) or any(len(t) == 0 for t in tgts):
            raise RuntimeError("ICP.compute_pairs: empty point cloud (libpointmatcher would throw)")
        so = _np.zeros(n + 1, _np.int32)
        to = _np.zeros(n + 1, _np.int32)
        so[1:] = _np.cumsum([len(s) for s in srcs])
        to[1:] = _np.cumsum([len(t) for t in tgts])
        src = _np.ascontiguousarray(_np.concatenate(srcs), _np.float32)
        tgt = _np.ascontiguousarray(_np.concatenate(tgts), _np.float32)
        g = _np.ascontiguousarray(_np.stack([self._guess(x) for x in guesses]), _np.float32)
        T = _np.zeros((n, 3, 3), _np.float32)
        st = _np.zeros(n, _np.int32)
        it = _np.zeros(n, _np.int32)
        with ctx.lock:
            ctx._check(ctx.lib.sfe_icp_compute_pairs(
                ctx.handle, _C.byref(params), _L.ptr(src, _C.c_float), _L.ptr(so, _C.c_int32),
                _L.ptr(tgt, _C.c_float), _L.ptr(to, _C.c_int32), _L.ptr(g, _C.c_float), n, _L.ptr(T, _C.c_float),
                _L.ptr(st, _C.c_int32), _L.ptr(it, _C.c_int32)))
        msgs = [_L.ICP_STATUS_MESSAGES.get(int(s), "ICP failure %d" % s) for s in st]
        return msgs, T, it

    def compute_jobs(self, src_all, tgt_all, jobs4, guesses9, out=None):
        """Scan matches named by a job table over two cloud pools (extension; what a farm worker runs per
        chunk): src_all / tgt_all = clouds back to back (N x 2 float32), jobs4 = n x (src_start, n_src,
        tgt_start, n_tgt) in points, guesses9 = n x 9.  Jobs may share clouds; jobs naming the same target slice
        share its preparation.  -> (status int32 [n], T [n x 3 x 3] float32, iterations int32 [n]); the three
        arrays may be handed in preallocated (``out=(status, T, iters)``, e.g. views of shared memory)."""
        params = self._chain()
        ctx = self._ctx or _L.default_context()
        src = _cloud(src_all, "ICP.compute_jobs(src_all)")
        tgt = _cloud(tgt_all, "ICP.compute_jobs(tgt_all)")
        jobs4 = _np.ascontiguousarray(jobs4, _np.int32).reshape(-1, 4)
        n = len(jobs4)
        g = _np.ascontiguousarray(guesses9, _np.float32).reshape(n, 9)
        if out is None:
            st, T, it = _np.zeros(n, _np.int32), _np.zeros((n, 3, 3), _np.float32), _np.zeros(n, _np.int32)
        else:
            st, T, it = out
        if n and ((jobs4[:, 1] <= 0).any() or (jobs4[:, 3] <= 0).any()):
            raise RuntimeError("ICP.compute_jobs: empty point cloud (libpointmatcher would throw)")
        with ctx.lock:
            ctx._check(ctx.lib.sfe_icp_compute_jobs(
                ctx.handle, _C.byref(params), _L.ptr(src, _C.c_float), len(src), _L.ptr(tgt, _C.c_float), len(tgt),
                _L.ptr(jobs4, _C.c_int32), _L.ptr(g, _C.c_float), n, _L.ptr(T, _C.c_float), _L.ptr(st, _C.c_int32),
                _L.ptr(it, _C.c_int32)))
        return st, T, it

    def getCovariance(self):
        """errorMinimizer->getCovariance() (pcl.cpp:213).  For the shipped point-to-point chain libpointmatcher's
        base ``ErrorMinimizer::getCovariance`` logs a warning and returns ``Matrix::Zero(6, 6)`` (restated from its
        published source, unpinned like the rest of pcl.cpp's third-party behaviour): 6 x 6 float32 zeros here.
        ``PointToPlaneErrorMinimizer`` overrides it with Censi's estimate from its 3-D normal equations and
        ``sensorStdDev``; the reference cannot run that chain (icp.yaml configures no normals) and the 2-D form
        is not defined anywhere, so a point-to-plane chain raises instead of returning a made-up matrix.  No
        caller in the reference reads either (grep)."""
        if self._chain().minimizer == 1:
            raise NotImplementedError("ICP.getCovariance for a point-to-plane chain: libpointmatcher's Censi estimate "
                                      "is 3-D only and the reference never runs this chain; use "
                                      "SLAM.compute_icp_with_cov's sampled covariance (slam.py:325-387)")
        return _np.zeros((6, 6), _np.float32)
