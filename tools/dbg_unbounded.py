import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sonar_slam_amd import _lib, icp_config, pcl, synth
ctx = _lib.default_context()
def run(variant, p, src, tgt, gs):
    ctx._check(ctx.lib.sfe_icp_set_tuning(ctx.handle, variant))
    icp = pcl.ICP(ctx); icp.setParams(p)
    r = icp.compute_batch(src, tgt, gs)
    ctx._check(ctx.lib.sfe_icp_set_tuning(ctx.handle, 0))
    n = len(src)
    d2 = np.zeros(n, np.float32); ix = np.zeros(n, np.int32)
    ctx._check(ctx.lib.sfe_debug_read_scratch(ctx.handle, 5, d2.ctypes.data, d2.nbytes))
    ctx._check(ctx.lib.sfe_debug_read_scratch(ctx.handle, 6, ix.ctypes.data, ix.nbytes))
    return r, d2, ix
src, tgt, guess, _ = synth.scan_pair(seed=2, n_src=2500, n_tgt=2500 - 37)
p = icp_config.shipped_params(max_iter=1)
os.environ["SFE_SW_NO_LDSQ"] = "1"
(b, d2b, ixb) = run(4, p, src, tgt, [guess])
for env in ({}, {"SFE_SW_UNBOUNDED_COOP": "0"}):
    os.environ.update(env)
    (a, d2a, pa) = run(0, p, src, tgt, [guess])
    for k in env: del os.environ[k]
    print(env, "equal T:", bool(np.array_equal(a[1], b[1])))
    # brute force: d2 = inf for no match; sweep: pos >= 0 exact (d2 valid), -1 none, <= -2 inexact (upper bound)
    exact = pa >= 0
    print("  sweep: exact %d none %d inexact %d; brute finite %d" % (exact.sum(), (pa == -1).sum(), (pa <= -2).sum(), np.isfinite(d2b).sum()))
    bad = np.flatnonzero(exact & (d2a != d2b))
    print("  exact queries whose d2 differs from brute force:", len(bad))
    for q in bad[:12]:
        print("    q %d src %s sweep d2 %.6f pos %d  brute d2 %.6f idx %d" % (q, src[q], d2a[q], pa[q], d2b[q], ixb[q]))
    nb = np.flatnonzero((pa == -1) & np.isfinite(d2b))
    print("  none in sweep but matched in brute:", len(nb), [(int(q), float(d2b[q])) for q in nb[:8]])
    ib = np.flatnonzero((pa <= -2) & (d2b > d2a))
    print("  inexact whose upper bound is below the true distance:", len(ib))
