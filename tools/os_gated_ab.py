#!/usr/bin/env python
"""OS-CFAR behind the intensity gate: candidates-only kernel (cfar_u8_os_gated) vs the sliding-histogram kernel, resident
batch of 1024 x 512 frames, HIP-event time per launch and algorithmic GB/s (2 B per pixel)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sonar_slam_amd import _lib, synth  # noqa: E402

ctx = _lib.default_context()
NF, ROWS, COLS = 512, 1024, 512
base = [synth.sonar_frame(seed=s) for s in range(16)]
frames = np.stack([base[j % 16] for j in range(NF)])
d_img = ctx.alloc(frames.nbytes)
d_mask = ctx.alloc(frames.nbytes)
d_img.upload(frames)
for (th, gh, k, tau) in ((20, 5, 10, 9.137608674642355), (10, 2, 8, 3.0), (40, 10, 30, 2.0)):
    for gate in (65, 20, -1):
        for env in ({}, {"SFE_CFAR_NO_OS_GATED": "1"}):
            os.environ.update(env)

            def run():
                ctx._check(ctx.lib.sfe_cfar_u8_batch_dev(ctx.handle, d_img.ptr, NF, ROWS, COLS, 3, th, gh, k, float(tau), gate, d_mask.ptr, None))
            run()
            ctx.sync()
            ctx.timer_start()
            for _ in range(5):
                run()
            ms = ctx.timer_stop() / 5
            for kk in env:
                del os.environ[kk]
            n_det = int(d_mask.download(np.uint8, ROWS * COLS).sum())
            print("OS T=%d G=%d k=%d gate %3d %-22s %8.3f ms / %d frames  %7.0f GB/s algorithmic  (%d detections in frame 0)"
                  % (th, gh, k, gate, "histogram kernel" if env else "default", ms, NF, 2.0 * NF * ROWS * COLS / ms / 1e6, n_det), flush=True)
