"""Test-side backend of an IcpFarm worker (farm.IcpFarm(_backend="farm_backend:oracle_compute")): the same
(name, run) contract as farm._hip_compute, computed by the CPU oracle, so that the farm's process / shared-memory
protocol is exercised with world size 2 on a box without a GPU.  Test infrastructure only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def oracle_compute(device, params_dict):
    import oracle
    prm = oracle.IcpParams(precision=1, **params_dict)

    def run(v, chunk):
        for j, (s0, ns, t0, nt) in enumerate(v["jobs4"]):
            st, T, it = oracle.icp(v["src"][s0:s0 + ns], v["tgt"][t0:t0 + nt], v["guess"][j].reshape(3, 3), prm)
            v["status"][j], v["T"][j], v["iters"][j] = st, T, it
    return "oracle worker %d (pid %d)" % (device, os.getpid()), run


def slow_oracle_compute(device, params_dict):
    """oracle_compute with a pause in front: a parent that reads results too early gets stale ones"""
    import time
    name, run = oracle_compute(device, params_dict)

    def slow(v, chunk):
        time.sleep(0.3)
        run(v, chunk)
    return name, slow


def failing_compute(device, params_dict):
    def run(v, chunk):
        raise ValueError("boom on device %d" % device)
    return "failing worker", run
