#!/usr/bin/env python
"""A/B sweep of the CFAR ring kernel (variant x tile rows) on a batch larger than the Infinity
Cache, timed with HIP events on the library's stream.  Interleaved rounds, reports median/min."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sonar_slam_amd import _lib, synth  # noqa: E402
from sonar_slam_amd.CFAR import CFAR  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=1024)
    ap.add_argument("--rows", type=int, default=1024)
    ap.add_argument("--cols", type=int, default=512)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--variants", default="2,3")
    ap.add_argument("--tiles", default="0,52,104,208,312,520,1024")
    ap.add_argument("--only", action="store_true", help="just run the default config a few times (for rocprof)")
    ap.add_argument("--bits", action="store_true", help="the bit-stream kernel (sfe_cfar_u8_bits_batch_dev)")
    a = ap.parse_args()
    ctx = _lib.default_context()
    det = CFAR(40, 10, 0.1, 10)
    th, gh, tau = det.params["SOCA"]
    base = np.stack([synth.sonar_frame(seed=s, rows=a.rows, cols=a.cols) for s in range(16)])
    fb = a.rows * a.cols
    d_in, d_out = ctx.alloc(a.frames * fb), ctx.alloc(a.frames * fb)
    for f0 in range(0, a.frames, 16):
        n = min(16, a.frames - f0)
        d_in.upload(base[:n], offset=f0 * fb)

    def launch():
        if a.bits:
            ctx._check(ctx.lib.sfe_cfar_u8_bits_batch_dev(ctx.handle, d_in.ptr, a.frames, a.rows, a.cols, 1, th, gh, 0,
                                                          float(tau), 65, d_out.ptr))
            return
        ctx._check(ctx.lib.sfe_cfar_u8_batch_dev(ctx.handle, d_in.ptr, a.frames, a.rows, a.cols, 1, th, gh, 0,
                                                 float(tau), 65, d_out.ptr, None))

    def timed():
        launch()
        ctx.sync()
        ctx.timer_start()
        for _ in range(a.reps):
            launch()
        return ctx.timer_stop() / a.reps

    bytes_ = (1.125 if a.bits else 2.0) * fb * a.frames   # 1 B read + 1 B (1 bit) written per pixel
    if a.only:
        if len(a.tiles.split(",")) == 1:
            ctx._check(ctx.lib.sfe_cfar_set_tuning(ctx.handle, int(a.tiles), 0))
        for _ in range(3):
            ms = timed()
            print("default: %.4f ms  %.0f GB/s" % (ms, bytes_ / ms / 1e6))
        return
    cfgs = [(int(v), int(t)) for v in a.variants.split(",") for t in a.tiles.split(",")]
    res = {c: [] for c in cfgs}
    for _ in range(a.rounds):
        for c in cfgs:
            ctx._check(ctx.lib.sfe_cfar_set_tuning(ctx.handle, c[1], c[0]))
            res[c].append(timed())
    ctx._check(ctx.lib.sfe_cfar_set_tuning(ctx.handle, 0, 0))
    print("frames=%d %dx%d  bytes/launch=%.0f MB" % (a.frames, a.rows, a.cols, bytes_ / 1e6))
    for c in cfgs:
        ms = np.array(res[c])
        print("variant %d tile_rows %4d : median %.4f ms  min %.4f ms  -> %.0f GB/s (median)  %.0f GB/s (best)"
              % (c[0], c[1], np.median(ms), ms.min(), bytes_ / np.median(ms) / 1e6, bytes_ / ms.min() / 1e6))


if __name__ == "__main__":
    main()
