#!/bin/bash
mkdir -p gpurun_out
timeout 900 python - <<'PY' 2>&1 | tail -45 | tee gpurun_out/r05_loop_closure_profile.txt
import cProfile, pstats, os, sys, io
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import bench_legs
from sonar_slam_amd import _lib
from sonar_slam_amd.CFAR import CFAR
ctx = _lib.default_context()
det = CFAR(40, 10, 0.1, 10)
bench_legs.loop_closure(ctx, det, 16)          # warm: checks, scratch
import numpy as np, time
from sonar_slam_amd import synth, store as st
from sonar_slam_amd.feature_extraction import FeatureExtraction, SonarPing, oculus_bearings
from sonar_slam_amd.replay import FrontEnd, replay
bearings = oculus_bearings(512)
world = synth.world_structure(seed=2, n=9000)
true, dr = synth.trajectory(n=14, step=1.7, turn=2 * np.pi / 13, seed=21, start=(20.0, 0.0, 0.0))
frames = [synth.render_ping(world, true[k], bearings, rows=1024, seed=7000 + k) for k in range(14)]
fe = FeatureExtraction(ctx)
fe.Ntc, fe.Ngc, fe.Pfa, fe.rank, fe.alg, fe.threshold = 40, 10, 0.1, 10, "SOCA", 65
fe.resolution, fe.outlier_filter_radius, fe.outlier_filter_min_points, fe.skip = 0.5, 1.0, 5, 1
fe.configure()
pings = [SonarPing(f, bearings, 30.0 / 1024, ping_id=k) for k, f in enumerate(frames)]
fe.generate_map_xy(pings[0])
s = st.CloudStore(ctx, capacity_points=1 << 20, max_clouds=1024)
front = FrontEnd(ctx, keyframe_translation=1.5, keyframe_duration=0.5, store=s, nssm_enable=True, mcd_random_state=0)
pr = cProfile.Profile(); pr.enable()
log, t_fe, t_slam = replay(pings, np.arange(14, dtype=float), dr, fe, front)
pr.disable()
print("t_slam %.1f ms" % (1e3 * t_slam))
pstats.Stats(pr).sort_stats("cumulative").print_stats(38)
PY
