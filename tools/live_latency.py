#!/usr/bin/env python
"""Latency of the LIVE single-item path (what the ROS nodes call once per ping / per scan match), host wall clock
per call incl. the PCIe copies and the one synchronisation:
  * FeatureExtraction.callback on a 1024 x 512 ping: fused call (sfe_feature_extract_ping) vs the per-stage calls
  * pcl.ICP.compute (shipped chain) on feature-cloud-sized pairs (10^2 .. 10^3 points) and on 5000 points
Prints one JSON object (bench.py embeds it)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sonar_slam_amd import _lib, icp_config, pcl, synth  # noqa: E402
from sonar_slam_amd.feature_extraction import FeatureExtraction, SonarPing, oculus_bearings  # noqa: E402


def measure(ctx=None, pings=30, reps=30):
    ctx = ctx or _lib.default_context()
    fe = FeatureExtraction(ctx)
    fe.Ntc, fe.Ngc, fe.Pfa, fe.rank, fe.alg, fe.threshold, fe.skip = 40, 10, 0.1, 10, "SOCA", 65, 1
    fe.configure()
    frames = [SonarPing(synth.sonar_frame(seed=900 + i), oculus_bearings(512), 30.0 / 1024, ping_id=0) for i in range(8)]
    out = {}
    for name, fused in (("fused", True), ("per_stage", False)):
        fe.fused = fused
        n_pts = [len(fe.callback(p)) for p in frames[:2]]                 # warm-up: maps, geometry, scratch
        t = []
        for i in range(pings):
            t0 = time.perf_counter()
            fe.callback(frames[i % len(frames)])
            t.append(time.perf_counter() - t0)
        out["ping_%s_us" % name] = 1e6 * float(np.median(t))
    out["ping_cloud_points"] = n_pts
    icp = pcl.ICP(ctx)
    icp.setParams(icp_config.shipped_params())
    for n in (200, 1000, 5000):
        s, tg, g, _ = synth.scan_pair(seed=40 + n, n_src=n, n_tgt=n)
        icp.compute(s, tg, g)
        t = []
        for _ in range(reps):
            t0 = time.perf_counter()
            msg, T = icp.compute(s, tg, g)
            t.append(time.perf_counter() - t0)
        assert msg == "success"
        out["scan_match_%d_pts_us" % n] = 1e6 * float(np.median(t))
    return out


if __name__ == "__main__":
    print(json.dumps(measure()))
