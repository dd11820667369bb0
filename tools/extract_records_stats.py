#!/usr/bin/env python
"""What the record path of the extraction sees on the bench's frames: records per frame and per gather workgroup, spills, and
how many frames the merge kernel hands back to the canvas kernels (scratch slot 62 after one launch: counts per region, flags).
usage: extract_records_stats.py [frames per launch] [distinct frames]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from sonar_slam_amd import _lib, icp_config, synth  # noqa: E402
from sonar_slam_amd.CFAR import CFAR  # noqa: E402
from sonar_slam_amd.feature_extraction import FeatureExtraction, SonarPing, oculus_bearings  # noqa: E402
from sonar_slam_amd.pipeline import KeyframeBatch  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
D = int(sys.argv[2]) if len(sys.argv) > 2 else 256
ctx = _lib.default_context()
det = CFAR(40, 10, 0.1, 10)
fe = FeatureExtraction(ctx)
fe.Ntc, fe.Ngc, fe.Pfa, fe.rank, fe.alg, fe.threshold = 40, 10, 0.1, 10, "SOCA", 65
fe.configure()
base = [synth.sonar_frame(seed=s) for s in range(D)]
frames = np.stack([base[j % D] for j in range(B)])
fe.generate_map_xy(SonarPing(frames[0], oculus_bearings(bench.COLS), 30.0 / bench.ROWS))
kb = KeyframeBatch(ctx, fe.geometry, det.params["SOCA"], "SOCA", 65, icp_config.shipped_params(), B)
kb.upload_frames(frames)
kb.run_cfar()
kb.run_extract()
ctx.sync()
slices = max(2, min(64, 8192 // B))
if B >= 64:
    slices = max(16, min(32, slices))
raw = np.zeros(B * (slices + 2), np.int32)
ctx._check(ctx.lib.sfe_debug_read_scratch(ctx.handle, 62, raw.ctypes.data, raw.nbytes))
cnt = raw[:B * (slices + 1)].reshape(B, slices + 1)
flags = raw[B * (slices + 1):]
pts = kb.d_cnt.download(np.int32, B)
tot = cnt.sum(axis=1)
print("%d frames per launch (%d distinct), %d gather workgroups per frame" % (B, D, slices))
print("points per frame: mean %.0f max %d" % (pts.mean(), pts.max()))
print("records per frame: mean %.0f max %d;  per workgroup: mean %.0f max %d;  spilled per frame: mean %.1f max %d"
      % (tot.mean(), tot.max(), cnt[:, :slices].mean(), cnt[:, :slices].max(), cnt[:, slices].mean(), cnt[:, slices].max()))
print("frames handed back to the canvas kernels: %d of %d" % (int((flags != 0).sum()), B))
