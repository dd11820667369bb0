#!/usr/bin/env python
"""tools/bench_legs.stream_frames at several (frame buffers, upload chunk) settings on one resident batch: which of them
keeps PCIe busy while the step's kernels run.  python tools/stream_probe.py [batch]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402

import bench  # noqa: E402
import bench_legs  # noqa: E402
from sonar_slam_amd import _lib, icp_config, synth  # noqa: E402
from sonar_slam_amd.CFAR import CFAR  # noqa: E402
from sonar_slam_amd.feature_extraction import FeatureExtraction, SonarPing, oculus_bearings  # noqa: E402
from sonar_slam_amd.pipeline import KeyframeBatch  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ctx = _lib.default_context()
det = CFAR(40, 10, 0.1, 10)
fe = FeatureExtraction(ctx)
fe.Ntc, fe.Ngc, fe.Pfa, fe.rank, fe.alg, fe.threshold = 40, 10, 0.1, 10, "SOCA", 65
fe.configure()
base = [synth.sonar_frame(seed=s) for s in range(32)]
frames = np.stack([base[j % 32] for j in range(B)])
fe.generate_map_xy(SonarPing(frames[0], oculus_bearings(bench.COLS), 30.0 / bench.ROWS))
pairs = [synth.scan_pair(seed=j, n_src=bench.N_PTS, n_tgt=bench.N_PTS) for j in range(64)]
icp_p = icp_config.shipped_params(minimizer=1, use_diff_checker=0, max_iter=30)
kb = KeyframeBatch(ctx, fe.geometry, det.params["SOCA"], "SOCA", 65, icp_p, B)
kb.upload_frames(frames)
kb.upload_scan_pairs([pairs[j % 64][0] for j in range(B)], [pairs[j % 64][1] for j in range(B)],
                     np.stack([pairs[j % 64][2] for j in range(B)]))
ctx.sync()
ctx._check(ctx.lib.sfe_icp_set_tuning(ctx.handle, 8))
kb.run()
ctx.sync()
combos = ((2, 512), (3, 512), (3, 256), (3, 64), (4, 256))
if len(sys.argv) > 2:       # e.g. "3x64,3x256,3x512,2x512,3x512": the order matters for what is being asked (does a setting or the process's age decide?)
    combos = tuple(tuple(int(v) for v in c.split("x")) for c in sys.argv[2].split(","))
keep = {} if os.environ.get("STREAM_PROBE_KEEP") else None     # one pinned pool and one set of frame buffers for every call
for nb, ch in combos:
    r = bench_legs.stream_frames(ctx, kb, True, steps=4, distinct=512, n_buffers=nb, chunk=ch, reuse=keep)
    print(json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items() if k not in ("workload", "note")}))
