"""CPU: host logic -- ICP YAML parsing, the C-ABI surface, loud failure without a GPU,
the job-farm sharding (incl. a 2-process gloo run)."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from sonar_slam_amd import _lib, farm, icp_config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SHIPPED_ICP_YAML = """readingDataPointsFilters:

referenceDataPointsFilters:

matcher:
  KDTreeMatcher:
    knn: 1
    epsilon: 0 
    maxDist: 10.0

outlierFilters:
  - MaxDistOutlierFilter:
      maxDist: 3.0
  - TrimmedDistOutlierFilter:
      ratio: 0.8

errorMinimizer:
  # PointToPlaneErrorMinimizer:
  #   force2D: 1
  PointToPointErrorMinimizer

transformationCheckers:
  - CounterTransformationChecker:
      maxIterationCount: 40
  - DifferentialTransformationChecker:
      minDiffRotErr: 0.01
      minDiffTransErr: 0.1
      smoothLength: 4   

inspector:
  NullInspector
"""


def test_parse_shipped_icp_yaml():
    p = icp_config.parse_icp_yaml(SHIPPED_ICP_YAML)
    assert p.as_dict() == icp_config.shipped_params().as_dict()
    assert (p.minimizer, p.max_iter, p.smooth_len) == (0, 40, 4)
    assert p.trim_ratio == np.float32(0.8) and p.max_dist_filter == 3.0 and p.matcher_max_dist == 10.0


def test_parse_point_to_plane_variant():
    y = SHIPPED_ICP_YAML.replace("  PointToPointErrorMinimizer", "").replace(
        "  # PointToPlaneErrorMinimizer:\n  #   force2D: 1", "  PointToPlaneErrorMinimizer:\n    force2D: 1")
    assert icp_config.parse_icp_yaml(y).minimizer == 1


@pytest.mark.parametrize("bad", [
    "matcher:\n  KDTreeMatcher:\n    knn: 3\n",
    "matcher:\n  NullMatcher\n",
    "outlierFilters:\n  - VarTrimmedDistOutlierFilter:\n      minRatio: 0.1\n",
    "errorMinimizer:\n  PointToPlaneErrorMinimizer\n",
    "referenceDataPointsFilters:\n  - SurfaceNormalDataPointsFilter:\n      knn: 5\n",
    "transformationCheckers:\n  - BoundTransformationChecker:\n      maxRotationNorm: 1\n",
    "somethingElse: 1\n",
])
def test_unknown_modules_are_rejected(bad):
    with pytest.raises(icp_config.IcpConfigError):
        icp_config.parse_icp_yaml(bad)


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "sonarfe.h")).read()
    declared = set(re.findall(r"\b(sfe_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no prototypes found in include/sonarfe.h"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    lib = _lib.load_library()          # dlopen works without a GPU
    for name in declared:
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.sfe_version()


def test_params_struct_matches_header():
    hdr = open(os.path.join(ROOT, "include", "sonarfe.h")).read()
    body = hdr[hdr.index("typedef struct sfe_icp_params {"):hdr.index("} sfe_icp_params;")]
    fields = re.findall(r"^\s*(?:float|int)\s+([a-z_]+);", body, re.M)
    assert fields == [n for n, _ in _lib.IcpParams._fields_]


def test_no_device_fails_loudly_not_silently():
    """On a box without a GPU the product must raise, never fall back to a CPU path."""
    import ctypes as C
    lib = _lib.load_library()
    n = C.c_int(-1)
    rc = lib.sfe_device_count(C.byref(n))
    if rc == 0 and n.value > 0:
        pytest.skip("a HIP device is visible")
    with pytest.raises(_lib.SonarFEError):
        _lib.Context(0)


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "sonar_slam_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", txt, re.M), f
                assert "libsonar_oracle" not in txt, f          # never dlopen'ed
                assert not re.search(r'#include\s+"[^"]*oracle', txt), f  # never compiled in


def test_shard_and_scatter_back():
    for n, w in [(10, 3), (7, 8), (0, 2), (16, 4)]:
        shards = [farm.shard(n, r, w) for r in range(w)]
        assert sorted(sum(shards, [])) == list(range(n))
        res = [[j * j for j in s] for s in shards]
        assert farm.scatter_back(n, w, res) == [j * j for j in range(n)]
    with pytest.raises(ValueError):
        farm.shard(4, 2, 2)


@pytest.mark.parametrize("world", [2, 8])
def test_gloo_farm(world):
    """world_size 2 and 8 (the target node: 8 x MI355X) over gloo on CPU: each rank matches its shard of ICP jobs (with
    the oracle as the stand-in compute, tests may use it) and every rank ends up with all poses in job order --
    rendezvous, shard arithmetic (job j -> rank j mod W) and the gather, with no collective on the data path."""
    script = os.path.join(ROOT, "tests", "gloo_farm_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29533 + world), WORLD_SIZE=str(world),
               PYTHONPATH=ROOT, OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, script], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
        assert "FARM_OK" in o, o


def test_farm_packs_shared_clouds_once():
    from sonar_slam_amd import farm
    rng = np.random.default_rng(0)
    a, b, c = (rng.normal(size=(n, 2)).astype(np.float32) for n in (5, 7, 3))
    g = np.eye(3, dtype=np.float32)
    jobs = [(a, b, [g, g, g]), (c, b, [g]), (a, c, [])]
    sp, tp, ns, nt, rows, gs = farm.pack_jobs(jobs)
    assert [len(x) for x in sp] == [5, 3] and [len(x) for x in tp] == [7, 3] and (ns, nt) == (8, 10)
    assert rows == [(0, 5, 0, 7)] * 3 + [(5, 3, 0, 7)] and len(gs) == 4
    lay = farm._layout(ns, nt, len(rows))
    assert all(lay[k] % 64 == 0 for k in ("src", "tgt", "jobs4", "guess", "T", "status", "iters"))
    with pytest.raises(RuntimeError):
        farm.pack_jobs([(np.zeros((0, 2), np.float32), b, [g])])
    with pytest.raises(TypeError):
        farm.pack_jobs([(a, b, [np.eye(4)])])


def test_persistent_two_worker_farm_over_shared_memory():
    """world size 2 on the CPU: two persistent worker processes, jobs through shared memory, two consecutive
    batches on the same workers (the second one larger: the block grows), results in job order and equal to
    the oracle called directly; a failing worker surfaces as an exception in the parent."""
    import oracle
    from sonar_slam_amd import farm, icp_config, synth
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    p = icp_config.shipped_params()
    pairs = [synth.scan_pair(seed=s, n_src=150 + 10 * s, n_tgt=140) for s in range(3)]
    rng = np.random.default_rng(1)

    def batch(n):
        jobs = []
        for j in range(n):
            s, t, g, _ = pairs[j % 3]
            jobs.append((s, t, [g @ synth.pose_matrix(*rng.normal(0, [0.05, 0.05, 0.005])).astype(np.float32)
                                for _ in range(1 + j % 2)]))
        return jobs
    with farm.IcpFarm(p, devices=[0, 1], chunk=4, _backend="farm_backend:oracle_compute") as f:
        pids = None
        for n in (5, 9):
            jobs = batch(n)
            out = f.run(jobs)
            assert len(out) == n
            for (s, t, gs), (msgs, T, it) in zip(jobs, out):
                assert len(msgs) == len(gs) == len(T) == len(it)
                for g, m, Tj, i in zip(gs, msgs, T, it):
                    st, To, ito = oracle.icp(s, t, g, oracle.IcpParams(precision=1, **p.as_dict()))
                    assert m == oracle.ICP_STATUS_MESSAGES[st] and i == ito and np.array_equal(Tj, To)
            now = [w.proc.pid for w in f._workers]
            assert pids is None or pids == now          # the same processes served both batches
            pids = now
            assert len(set(w.name for w in f._workers)) == 2
    assert f._workers == []
    with farm.IcpFarm(p, devices=[0], _backend="farm_backend:failing_compute") as f:
        with pytest.raises(RuntimeError, match="boom"):
            f.run(batch(2))


def test_persistent_eight_worker_farm_over_shared_memory():
    """the farm at the target machine's width: eight persistent workers (CPU-backend stand-ins for the eight devices), jobs
    dealt j mod 8 through eight shared-memory blocks with distinct names, fewer jobs than workers and more, results in job
    order -- so that the first 8-GPU run cannot fail on process start-up, block naming or shard arithmetic."""
    import oracle
    from sonar_slam_amd import farm, icp_config, synth
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    p = icp_config.shipped_params()
    pairs = [synth.scan_pair(seed=40 + s, n_src=120 + 7 * s, n_tgt=130) for s in range(5)]
    with farm.IcpFarm(p, devices=list(range(8)), chunk=4, _backend="farm_backend:oracle_compute") as f:
        assert len(f._workers) == 8 and len(set(w.name for w in f._workers)) == 8
        for n in (3, 8, 21):
            jobs = [(pairs[j % 5][0], pairs[j % 5][1], [pairs[j % 5][2]]) for j in range(n)]
            out = f.run(jobs)
            assert len(out) == n
            for (s, t, gs), (msgs, T, it) in zip(jobs, out):
                st, To, ito = oracle.icp(s, t, gs[0], oracle.IcpParams(precision=1, **p.as_dict()))
                assert msgs[0] == oracle.ICP_STATUS_MESSAGES[st] and it[0] == ito and np.array_equal(T[0], To)
        assert [farm.shard(21, r, 8) for r in range(8)] == [list(range(r, 21, 8)) for r in range(8)]
    assert f._workers == []


def test_farm_bad_job_leaves_no_request_in_flight():
    """a job that fails validation on rank 1 used to leave rank 0's request unanswered, and the next run()
    read that stale reply and took results the worker had not produced yet.  Now every rank is validated before
    anything is sent, replies carry the request number, and the run after a caught error is correct."""
    import oracle
    from sonar_slam_amd import farm, icp_config, synth
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    p = icp_config.shipped_params()
    good = [synth.scan_pair(seed=s, n_src=120, n_tgt=110) for s in range(4)]
    jobs = [(s, t, [g]) for s, t, g, _ in good]
    bad = list(jobs)
    bad[1] = (np.zeros((0, 2), np.float32), good[1][1], [good[1][2]])      # job 1 -> rank 1: empty source cloud
    with farm.IcpFarm(p, devices=[0, 1], _backend="farm_backend:slow_oracle_compute") as f:
        for _ in range(2):
            with pytest.raises(RuntimeError, match="empty source"):
                f.run(bad)
            out = f.run(jobs)
            for (s, t, gs), (msgs, T, it) in zip(jobs, out):
                st, To, ito = oracle.icp(s, t, gs[0], oracle.IcpParams(precision=1, **p.as_dict()))
                assert msgs[0] == oracle.ICP_STATUS_MESSAGES[st] and it[0] == ito and np.array_equal(T[0], To)
        # a stale reply (older request number) in the pipe is skipped, not taken for the current request
        w = f._workers[0]
        w.conn.send(("run", w.shm.name, farm._layout(0, 0, 0), 1, -7))
        out = f.run(jobs)
        st, To, _ = oracle.icp(*good[0][:3], oracle.IcpParams(precision=1, **p.as_dict()))
        assert np.array_equal(out[0][1][0], To)


def test_farm_run_after_an_interrupted_run_waits_for_the_abandoned_request():
    """a run() interrupted between send and reply (Ctrl-C) leaves its request executing in the workers; the
    next run() must not refill a worker's shared-memory block under it.  The interrupted request is awaited (and its
    reply dropped) per worker before that worker's block is touched."""
    import oracle
    from sonar_slam_amd import farm, icp_config, synth
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    p = icp_config.shipped_params()
    pairs = [synth.scan_pair(seed=20 + s, n_src=130, n_tgt=120) for s in range(6)]
    first = [(s, t, [g]) for s, t, g, _ in pairs[:4]]
    second = [(s, t, [g]) for s, t, g, _ in pairs[2:]]          # other clouds, same layout: the blocks are reused as they are
    with farm.IcpFarm(p, devices=[0, 1], _backend="farm_backend:slow_oracle_compute") as f:
        real = f._recv_reply
        calls = []

        def interrupted(w, seq):
            calls.append(seq)
            raise KeyboardInterrupt()
        f._recv_reply = interrupted
        with pytest.raises(KeyboardInterrupt):
            f.run(first)
        f._recv_reply = real
        assert [w.pending for w in f._workers] == [1, 1]        # both requests are still out
        out = f.run(second)
        assert [w.pending for w in f._workers] == [None, None]
        for (s, t, gs), (msgs, T, it) in zip(second, out):
            st, To, ito = oracle.icp(s, t, gs[0], oracle.IcpParams(precision=1, **p.as_dict()))
            assert msgs[0] == oracle.ICP_STATUS_MESSAGES[st] and it[0] == ito and np.array_equal(T[0], To)


def test_icp_object_has_no_silent_default_chain(tmp_path):
    """PM::ICP() has no chain until loadFromYaml (pcl.cpp:185-197); a missing YAML is an error here, not
    libpointmatcher's setDefault() chain."""
    from sonar_slam_amd import pcl
    icp = pcl.ICP()
    pts = np.zeros((4, 2), np.float32)
    with pytest.raises(RuntimeError, match="chain is empty"):
        icp.compute(pts, pts, np.eye(3))
    with pytest.raises(RuntimeError, match="cannot open"):
        icp.loadFromYaml(str(tmp_path / "no_such_icp.yaml"))
    assert icp.params is None
    f = tmp_path / "icp.yaml"
    f.write_text(SHIPPED_ICP_YAML)
    icp.loadFromYaml(str(f))
    assert icp.params.as_dict() == icp_config.shipped_params().as_dict()
    assert np.array_equal(icp.getCovariance(), np.zeros((6, 6), np.float32))     # ErrorMinimizer's base: Zero(6, 6)
    icp.setParams(icp_config.shipped_params(minimizer=1))
    with pytest.raises(NotImplementedError, match="point-to-plane"):
        icp.getCovariance()
    with pytest.raises(RuntimeError, match="libnabo"):
        pcl.match(np.zeros((2, 2), np.float32), np.zeros((5, 2), np.float32), 3, 1.0, ctx=object())


def test_load_from_yaml_can_behave_like_the_reference_at_the_call_site(tmp_path, capsys, monkeypatch):
    """pcl.cpp:190-194: a launch file with a wrong icp_config path prints one line and carries on (slam_ros.py:124-125
    sees no exception).  Opt-in here (strict=False or SONARFE_YAML_FALLBACK=shipped): the reference's exact line, then
    the chain of the shipped config/icp.yaml; the default stays loud."""
    from sonar_slam_amd import pcl
    missing = str(tmp_path / "no_such_icp.yaml")
    icp = pcl.ICP()
    icp.loadFromYaml(missing, strict=False)
    assert capsys.readouterr().out == "Failed to load %s. Use default configuration.\n" % missing
    assert icp.params.as_dict() == icp_config.shipped_params().as_dict()
    icp2 = pcl.ICP()
    monkeypatch.setenv("SONARFE_YAML_FALLBACK", "shipped")
    icp2.loadFromYaml(missing)                       # the reference's one-argument call
    assert "Failed to load" in capsys.readouterr().out and icp2.params is not None
    with pytest.raises(RuntimeError, match="cannot open"):
        pcl.ICP().loadFromYaml(missing, strict=True)     # an explicit strict wins over the environment
    monkeypatch.delenv("SONARFE_YAML_FALLBACK")
    with pytest.raises(RuntimeError, match="cannot open"):
        pcl.ICP().loadFromYaml(missing)
    # a file that opens but does not parse is an error either way (the reference's loadFromYaml throws on it too)
    bad = tmp_path / "bad.yaml"
    bad.write_text("matcher:\n  NoSuchMatcher:\n    knn: 1\n")
    with pytest.raises(Exception):
        pcl.ICP().loadFromYaml(str(bad), strict=False)


def test_shims_bind_every_name_the_reference_modules_define():
    """cfar.cpp:194-204 and pcl.cpp:176-213 by name and arity: every m.def / .def of the two pybind modules exists in
    the shims and accepts the reference's positional arguments (overloads included)."""
    import inspect

    from sonar_slam_amd import cfar, pcl

    def accepts(fn, n_args):
        sig = inspect.signature(fn)
        pos = [p for p in sig.parameters.values() if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD)]
        var = any(p.kind == p.VAR_POSITIONAL for p in sig.parameters.values())
        required = sum(p.default is p.empty for p in pos)
        return required <= n_args and (var or n_args <= len(pos))
    # cfar.cpp:10,30,53,76,98,120,145,170: (img, train_hs, guard_hs, tau), os adds k before tau
    for name, n in (("ca", 4), ("soca", 4), ("goca", 4), ("os", 5), ("ca2", 4), ("soca2", 4), ("goca2", 4), ("os2", 5)):
        assert accepts(getattr(cfar, name), n), name
    # pcl.cpp:178-184
    assert accepts(pcl.remove_outlier, 3)                       # (points, radius, min_points)
    assert accepts(pcl.density_filter, 4) and accepts(pcl.density_filter, 5)   # (pts, knn, min, max) / (pts, desc, knn, min, max)
    assert accepts(pcl.downsample, 2) and accepts(pcl.downsample, 3)           # (pts, res) / (pts, desc, res)
    assert accepts(pcl.match, 4)                                # (ref, in, knn, max_dist)
    # pcl.cpp:185-213: class ICP
    assert accepts(pcl.ICP, 0)
    for name, n in (("loadFromYaml", 1), ("compute", 3), ("getCovariance", 0)):
        assert accepts(getattr(pcl.ICP(), name), n), name
    # and the reference's own source names nothing else
    ref = os.path.join("/root/reference/bruce_slam/src/bruce_slam/cpp")
    if os.path.isdir(ref):     # (present in the build container only)
        import re
        for fname, mod in (("cfar.cpp", cfar), ("pcl.cpp", pcl)):
            text = open(os.path.join(ref, fname)).read()
            for name in re.findall(r'm\.def\("(\w+)"', text):
                assert hasattr(mod, name), (fname, name)
            for name in re.findall(r'\.def\("(\w+)"', text):
                assert hasattr(mod, name) or hasattr(pcl.ICP, name), (fname, name)


def test_glibc_rand_restatement_and_density_filter_surface():
    """std::rand of the reference's platform: the first outputs after srand(1) / srand(42) are glibc's; and
    pcl.density_filter behaves like the reference call (empty cloud returned, otherwise libpointmatcher's
    'was set but is not used' InvalidParameter for the minDensity it hands to MaxDensityDataPointsFilter)."""
    from sonar_slam_amd import pcl
    g = pcl._GlibcRand(1)
    assert [g.rand() for _ in range(5)] == [1804289383, 846930886, 1681692777, 1714636915, 1957747793]
    g.srand(42)
    assert g.rand() == 71876166
    empty = np.zeros((0, 2), np.float32)
    assert pcl.density_filter(empty, 10, 0.0, 100.0).shape == (0, 2)
    e2, d2 = pcl.density_filter(empty, np.zeros((0, 1), np.float32), 10, 0.0, 100.0)
    assert e2.shape == (0, 2) and d2.shape == (0, 1)
    with pytest.raises(RuntimeError, match="minDensity"):
        pcl.density_filter(np.ones((5, 2), np.float32), 3, 0.0, 100.0)
    with pytest.raises(TypeError):
        pcl.density_filter(np.ones((5, 2), np.float32), 3)
