#!/usr/bin/env python
"""Algorithmic GB/s (1 B read + 1 B written per pixel; + 4 B per pixel with the float threshold map of the *2 variants)
of every CFAR configuration CFAR.py can ask for: window sizes (Ntc, Ngc), the four variants, detect2.  Batches of
1024 x 512 frames larger than the Infinity Cache, HIP events on the library's stream.  One line per configuration:
which kernel served it and at what fraction of the 8 TB/s HBM peak."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sonar_slam_amd import _lib, synth  # noqa: E402
from sonar_slam_amd.CFAR import CFAR  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=512)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--windows", default="40,10 20,4 32,8 16,2 80,20 12,0")
    a = ap.parse_args()
    rows, cols = 1024, 512
    ctx = _lib.default_context()
    base = np.stack([synth.sonar_frame(seed=s, rows=rows, cols=cols) for s in range(16)])
    fb = rows * cols
    d_in, d_out = ctx.alloc(a.frames * fb), ctx.alloc(a.frames * fb)
    d_thr = ctx.alloc(a.frames * fb * 4)
    for f0 in range(0, a.frames, 16):
        d_in.upload(base[:min(16, a.frames - f0)], offset=f0 * fb)

    def timed(fn):
        fn()
        ctx.sync()
        ctx.timer_start()
        for _ in range(a.reps):
            fn()
        return ctx.timer_stop() / a.reps

    print("%-10s %-5s %-8s %9s %9s %7s" % ("Ntc,Ngc", "alg", "output", "ms", "GB/s", "of HBM"))
    for w in a.windows.split():
        ntc, ngc = (int(v) for v in w.split(","))
        det = CFAR(ntc, ngc, 0.1, max(1, ntc // 4))
        for alg in ("CA", "SOCA", "GOCA", "OS"):
            p = det.params[alg]
            th, gh, tau = p[0], p[1], p[-1]
            k = p[2] if alg == "OS" else 0
            for want_thr in (False, True):
                if want_thr and alg not in ("SOCA", "OS"):
                    continue
                def launch():
                    ctx._check(ctx.lib.sfe_cfar_u8_batch_dev(ctx.handle, d_in.ptr, a.frames, rows, cols, _lib.ALG[alg], th, gh,
                                                             k, float(tau), 65 if not want_thr else -1, d_out.ptr,
                                                             d_thr.ptr if want_thr else None))
                ms = timed(launch)
                gb = (6.0 if want_thr else 2.0) * fb * a.frames / ms / 1e6
                print("%-10s %-5s %-8s %9.3f %9.0f %6.1f%%" % (w, alg, "mask+thr" if want_thr else "mask", ms, gb, gb / 80.0),
                      flush=True)
                if alg == "OS" and not want_thr:
                    # cfar.os() of the drop-in has NO gate (and feature.yaml's threshold can be low): the pre-filtered candidate
                    # kernel of round 6 against the sliding histogram of rounds 2-5 (SFE_CFAR_NO_OS_PREF=1), same launch
                    for gate in (-1, 20):
                        for pref in (True, False):
                            if pref:
                                os.environ.pop("SFE_CFAR_NO_OS_PREF", None)
                            else:
                                os.environ["SFE_CFAR_NO_OS_PREF"] = "1"

                            def launch_g():
                                ctx._check(ctx.lib.sfe_cfar_u8_batch_dev(ctx.handle, d_in.ptr, a.frames, rows, cols, _lib.ALG[alg], th, gh,
                                                                         k, float(tau), gate, d_out.ptr, None))
                            ms = timed(launch_g)
                            gb = 2.0 * fb * a.frames / ms / 1e6
                            print("%-10s %-5s %-8s %9.3f %9.0f %6.1f%%   %s" % (
                                w, alg, "gate %d" % gate, ms, gb, gb / 80.0,
                                "pre-filtered candidates (cfar_u8_os_gated<PREF>)" if pref else "sliding histogram (cfar_u8_os)"), flush=True)
                    os.environ.pop("SFE_CFAR_NO_OS_PREF", None)


if __name__ == "__main__":
    main()
