#!/usr/bin/env python
"""A/B of the ICP kernel variants on a batch of 5000x5000 jobs (HIP-event timed, interleaved)."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sonar_slam_amd import _lib, icp_config, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--jobs", type=int, default=512)
    ap.add_argument("--n", type=int, default=5000)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--variants", default="0,1")
    a = ap.parse_args()
    ctx = _lib.default_context()
    srcs, tgts, gs = [], [], []
    for j in range(a.jobs):
        s, t, g, _ = synth.scan_pair(seed=j, n_src=a.n, n_tgt=a.n)
        srcs.append(s)
        tgts.append(t)
        gs.append(g.reshape(9))
    src, tgt, g = np.concatenate(srcs), np.concatenate(tgts), np.stack(gs)
    off = (np.arange(a.jobs + 1) * a.n).astype(np.int32)
    d_src, d_tgt, d_g = ctx.alloc(src.nbytes), ctx.alloc(tgt.nbytes), ctx.alloc(g.nbytes)
    d_src.upload(src)
    d_tgt.upload(tgt)
    d_g.upload(g)
    d_T, d_st, d_it = ctx.alloc(a.jobs * 36), ctx.alloc(a.jobs * 4), ctx.alloc(a.jobs * 4)
    modes = {"reference": icp_config.shipped_params(),
             "p2p30": icp_config.shipped_params(use_diff_checker=0, max_iter=30),
             "p2plane30": icp_config.shipped_params(minimizer=1, use_diff_checker=0, max_iter=30),
             "p2plane1": icp_config.shipped_params(minimizer=1, use_diff_checker=0, max_iter=1)}
    variants = [int(v) for v in a.variants.split(",")]

    def run(p):
        ctx._check(ctx.lib.sfe_icp_batch_dev(ctx.handle, C.byref(p), d_src.ptr, _lib.ptr(off, C.c_int32), d_tgt.ptr,
                                             _lib.ptr(off, C.c_int32), d_g.ptr, a.jobs, d_T.ptr, d_st.ptr, d_it.ptr))

    ref = {}
    for name, p in modes.items():
        times = {v: [] for v in variants}
        for _ in range(a.rounds):
            for v in variants:
                ctx._check(ctx.lib.sfe_icp_set_tuning(ctx.handle, v))
                run(p)
                ctx.sync()
                ctx.timer_start()
                run(p)
                times[v].append(ctx.timer_stop())
                T = d_T.download(np.float32, a.jobs * 9)
                it = d_it.download(np.int32, a.jobs)
                if name not in ref:
                    ref[name] = (T, it)
                else:
                    assert np.array_equal(ref[name][0], T) and np.array_equal(ref[name][1], it), \
                        "variant %d changes the result in mode %s" % (v, name)
        iters = int(ref[name][1].sum())
        for v in variants:
            ms = float(np.median(times[v]))
            pairs = float(a.n) * a.n * iters
            print("%-10s variant %d: %8.2f ms  (%d jobs, %.1f iters avg) -> %.2f Tpair/s, %.1f TFLOP/s-equiv"
                  % (name, v, ms, a.jobs, iters / a.jobs, pairs / ms / 1e9, 8 * pairs / ms / 1e9))
    ctx._check(ctx.lib.sfe_icp_set_tuning(ctx.handle, 0))


if __name__ == "__main__":
    main()
