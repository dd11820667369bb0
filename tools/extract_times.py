#!/usr/bin/env python
"""HIP-event time of the resident extraction (and CFAR / filters) on the bench frames, for A/B of its knobs."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from sonar_slam_amd import _lib, icp_config  # noqa: E402
from sonar_slam_amd.CFAR import CFAR  # noqa: E402
from sonar_slam_amd.feature_extraction import FeatureExtraction, SonarPing, oculus_bearings  # noqa: E402
from sonar_slam_amd.pipeline import KeyframeBatch  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ctx = _lib.default_context()
det = CFAR(40, 10, 0.1, 10)
fe = FeatureExtraction(ctx)
fe.Ntc, fe.Ngc, fe.Pfa, fe.rank, fe.alg, fe.threshold = 40, 10, 0.1, 10, "SOCA", 65
fe.configure()
frames = bench.make_inputs(0, 1)[0]
import numpy as np
from sonar_slam_amd import synth
base = [synth.sonar_frame(seed=s) for s in range(32)]
frames = np.stack([base[j % 32] for j in range(B)])
fe.generate_map_xy(SonarPing(frames[0], oculus_bearings(bench.COLS), 30.0 / bench.ROWS))
kb = KeyframeBatch(ctx, fe.geometry, det.params["SOCA"], "SOCA", 65, icp_config.shipped_params(), B)
kb.upload_frames(frames)


def timed(fn, reps):
    fn()
    ctx.sync()
    ctx.timer_start()
    for _ in range(reps):
        fn()
    return ctx.timer_stop() / reps


kb.run_cfar()
variant = int(os.environ.get("EXTRACT_VARIANT", "0"))      # 0 = records (default), 2 = canvas bitmap
ctx._check(ctx.lib.sfe_extract_set_tuning(ctx.handle, variant))
print("variant %d: cfar %.3f  extract %.3f  filter %.3f ms / %d frames; points/frame %.0f"
      % (variant, timed(kb.run_cfar, 10), timed(kb.run_extract, 10), timed(kb.run_filter, 5), B, kb.results()["counts"].mean()))
