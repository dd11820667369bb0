#!/usr/bin/env python
"""BASELINE configs[3]: a batch of independent synthetic keyframe scan-match jobs farmed over the
visible GPUs (farm.IcpFarm: one worker process per device, job j -> device j mod G, no collective).
Prints jobs/s.  The clouds are `--distinct` seeded pairs cycled, every job gets its own guess."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sonar_slam_amd import _lib, icp_config, synth  # noqa: E402
from sonar_slam_amd.farm import IcpFarm  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--jobs", type=int, default=10000)
    ap.add_argument("--distinct", type=int, default=128)
    ap.add_argument("--points", type=int, default=5000)
    ap.add_argument("--mode", choices=["p2plane30", "reference"], default="p2plane30")
    ap.add_argument("--devices", type=int, default=0, help="0 = all visible")
    ap.add_argument("--repeats", type=int, default=3, help="batches sent to the same (persistent) workers")
    a = ap.parse_args()
    p = (icp_config.shipped_params(minimizer=1, use_diff_checker=0, max_iter=30) if a.mode == "p2plane30"
         else icp_config.shipped_params())
    pairs = [synth.scan_pair(seed=s, n_src=a.points, n_tgt=a.points) for s in range(a.distinct)]
    rng = np.random.default_rng(0)
    jobs = []
    for j in range(a.jobs):
        s, t, g, _ = pairs[j % a.distinct]
        jobs.append((s, t, [g @ synth.pose_matrix(*rng.normal(0, [0.05, 0.05, 0.005])).astype(np.float32)]))
    n_dev = a.devices or _lib.device_count()
    farm = IcpFarm(p, devices=list(range(n_dev)))
    try:
        t0 = time.perf_counter()
        farm.start()                       # worker processes + their sfe_ctx: paid once per farm, not per batch
        t_start = time.perf_counter() - t0
        times = []
        for rep in range(a.repeats):
            t0 = time.perf_counter()
            out = farm.run(jobs)
            times.append(time.perf_counter() - t0)
    finally:
        farm.close()
    ok = sum(m[0] == "success" for m, _, _ in out)
    print("%d jobs (%dx%d points, %s) on %d device(s): worker start-up %.2f s once; batches %s s wall incl. shared-memory "
          "packing and host<->device copies -> %.0f jobs/s (best), %d converged"
          % (a.jobs, a.points, a.points, a.mode, n_dev, t_start, ["%.3f" % t for t in times], a.jobs / min(times), ok))


if __name__ == "__main__":
    main()
