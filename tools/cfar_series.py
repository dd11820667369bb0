#!/usr/bin/env python
"""Launch-by-launch time of the CFAR bit-stream kernel (1024 frames per launch), 400 launches back to back, each between its own
pair of HIP events: how the figure of a short run (the bench's roofline leg: 20 launches) relates to a long one (tools/cfar_sweep.py)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sonar_slam_amd import _lib, synth  # noqa: E402
from sonar_slam_amd.CFAR import CFAR  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = int(sys.argv[2]) if len(sys.argv) > 2 else 400
ctx = _lib.default_context()
th, gh, tau = CFAR(40, 10, 0.1, 10).params["SOCA"]
base = np.stack([synth.sonar_frame(seed=s) for s in range(16)])
fb = 1024 * 512
d_in, d_out = ctx.alloc(frames * fb), ctx.alloc(frames * fb)
for f0 in range(0, frames, 16):
    d_in.upload(base[:min(16, frames - f0)], offset=f0 * fb)
ctx.sync()
ms = []
for i in range(n):
    ctx.timer_start()
    ctx._check(ctx.lib.sfe_cfar_u8_bits_batch_dev(ctx.handle, d_in.ptr, frames, 1024, 512, 1, th, gh, 0, float(tau), 65, d_out.ptr))
    ms.append(ctx.timer_stop())
ms = np.array(ms)
print("%d frames per launch, %d launches one after the other, ms per launch:" % (frames, n))
for a in range(0, n, 25):
    print("  launches %3d-%3d: mean %.4f  min %.4f  max %.4f" % (a, min(a + 25, n) - 1, ms[a:a + 25].mean(), ms[a:a + 25].min(), ms[a:a + 25].max()))
