"""The third-party arithmetic behind the reference's hot path -- OpenCV (feature_extraction.py:227,231; slam.py:522-527),
PCL / libpointmatcher / libnabo behind ``bruce_slam.pcl`` (pcl.cpp:54-74,128-174,198-212) -- REAL where this machine has it,
the oracle's restatement where it does not.

Neither library is in the build image or on the GPU box, so every fixture of rounds 1-5 was generated with the
oracle standing in, and the parity of those pieces is "unpinned".  This module is the one place that decides which of the
two a fixture generator gets, and it keeps a record of every stand-in it handed out; `record()` goes into each fixture file
(`stand_ins`), so a file says by itself what it pins:

  * `cv2()`  -> the real module when `import cv2` works, else a namespace with the five calls the reference makes
  * `pcl()`  -> the real compiled `bruce_slam.pcl` when it imports, else a namespace over the oracle
  * `icp_compute(yaml)` -> `pcl.ICP().loadFromYaml(yaml).compute` or the oracle's chain with the shipped parameters

`tools/pin_thirdparty.py` writes `thirdparty_cv2.npz` / `thirdparty_pcl.npz` (inputs and the REAL libraries' outputs) on a
machine that has them; `tests/test_golden.py` then holds the ORACLE to those files.  Test infrastructure: nothing in the
product imports this."""
import json
import types

import numpy as np

_USED = set()


def real_cv2():
    try:
        import cv2 as m
        return m
    except Exception:
        return None


def real_pcl():
    """the reference's compiled pybind module (a sourced catkin workspace puts `bruce_slam` on the path)"""
    try:
        from bruce_slam import pcl as m
        if not hasattr(m, "ICP"):
            return None
        return m
    except Exception:
        return None


def cv2(oracle):
    m = real_cv2()
    if m is not None:
        return m
    _USED.add("cv2.remap / applyColorMap / getStructuringElement / dilate -> oracle.remap_u8 / colormap_jet_lut / "
              "ellipse_kernel / cost_grid")

    def get_structuring_element(shape, ksize, anchor=None):
        assert shape == 2 and ksize[0] == ksize[1] and anchor in (None, (-1, -1), (ksize[0] // 2, ksize[0] // 2))
        return oracle.ellipse_kernel(ksize[0] // 2)

    def dilate(img, kernel):
        hs = kernel.shape[0] // 2
        assert np.array_equal(kernel, oracle.ellipse_kernel(hs))
        r, c = np.nonzero(img)
        return oracle.cost_grid(r, c, img.shape[0], img.shape[1], hs)

    def apply_color_map(img, cmap):
        assert cmap == 2
        return oracle.colormap_jet_lut()[np.asarray(img, np.uint8)]
    return types.SimpleNamespace(INTER_LINEAR=1, MORPH_ELLIPSE=2, COLORMAP_JET=2,
                                 remap=lambda a, mx, my, interp: oracle.remap_u8(np.asarray(a, np.uint8), mx, my),
                                 applyColorMap=apply_color_map, getStructuringElement=get_structuring_element, dilate=dilate)


def pcl(oracle):
    m = real_pcl()
    if m is not None:
        return m
    _USED.add("pcl.downsample / remove_outlier / match -> oracle.downsample / remove_outlier / match")

    def downsample(*a):
        if len(a) == 3:                                   # (points, keys, resolution) -> (points, keys): pcl.cpp:143-159
            p, idx = oracle.downsample(np.asarray(a[0], np.float32), a[2], return_index=True)
            return p, np.asarray(a[1], np.float32)[idx]
        return oracle.downsample(np.asarray(a[0], np.float32), a[1])   # (pybind: Matrix = fp32)
    return types.SimpleNamespace(
        downsample=downsample,
        remove_outlier=lambda pts, r, k: oracle.remove_outlier(np.asarray(pts, np.float32), r, k),
        match=lambda tgt, src, k, r: oracle.match(np.asarray(tgt, np.float32), np.asarray(src, np.float32), r))


def icp_compute(oracle, yaml_path=None, precision=1):
    """-> compute(source, target, guess) -> (message, T) like pcl.ICP.compute (pcl.cpp:198-212)"""
    m = real_pcl()
    if m is not None and yaml_path is not None:
        icp = m.ICP()
        icp.loadFromYaml(yaml_path)
        return icp.compute
    _USED.add("pcl.ICP.compute -> oracle.icp (icp.yaml as shipped, fp64 sums)" if precision else
              "pcl.ICP.compute -> oracle.icp (icp.yaml as shipped, float sums)")
    prm = oracle.shipped_icp_params(precision=precision)

    def compute(src, tgt, g):
        st, T, _ = oracle.icp(np.asarray(src, np.float32), np.asarray(tgt, np.float32), np.asarray(g, np.float32), prm)
        return ("success", T) if st == 0 else ("failure", np.asarray(g, np.float32))
    return compute


def record():
    """what stood in for third-party code since the last record, for the fixture file being written ("[]": nothing did)"""
    out = np.array(json.dumps(sorted(_USED)))
    _USED.clear()
    return out


# ---- the checks tests/test_golden.py runs on thirdparty_cv2.npz / thirdparty_pcl.npz (written by tools/pin_thirdparty.py on a
# machine that has the libraries): the ORACLE against the real libraries' outputs on the inputs stored next to them ----
def check_cv2(fix, oracle):
    """-> number of arrays compared; raises AssertionError on the first difference.  Everything here is integer work: equal."""
    n = 0
    for i in range(2):
        mx, my = fix["map_x%d" % i], fix["map_y%d" % i]
        for kind in ("mask", "img"):                                     # feature_extraction.py:231 / :226
            got = oracle.remap_u8(fix["%s%d" % (kind, i)], mx, my)
            assert np.array_equal(got, fix["remap_%s%d" % (kind, i)]), "cv2.remap(%s%d) differs from the oracle" % (kind, i)
            n += 1
    for hs in range(1, 7):
        k = fix["ellipse%d" % hs]
        assert np.array_equal((k != 0).astype(np.uint8), oracle.ellipse_kernel(hs)), "getStructuringElement(%d) differs" % hs
        n += 1
        for i in range(2):
            g = fix["grid%d" % i]
            r, c = np.nonzero(g)
            got = oracle.cost_grid(r, c, g.shape[0], g.shape[1], hs)    # slam.py:515-527
            assert np.array_equal(got, fix["dilate%d_hs%d" % (i, hs)]), "cv2.dilate(grid%d, ellipse %d) differs" % (i, hs)
            n += 1
    assert np.array_equal(oracle.colormap_jet_lut(), np.asarray(fix["jet"]).reshape(256, 3)), "COLORMAP_JET differs"
    return n + 1


def check_pcl(fix, oracle, pose_tol=1e-4):
    """-> (arrays compared, worst ICP pose difference).  The filters and the matcher: equal, order included (float32 in,
    float32 out, no arithmetic on the values).  ICP: same message, the guess returned on failure (pcl.cpp:203,207-210), pose
    within north_star's 1e-4 m / rad of the oracle in float -- PointMatcher<float> (pcl.cpp:12)."""
    import json as _json
    n = 0
    if "icp_yaml_params" in fix.files:
        want = {k: v for k, v in _json.loads(str(fix["icp_yaml_params"])).items()}
        have = oracle.shipped_icp_params(precision=0)
        for k, v in want.items():
            if hasattr(have, k) and k != "precision":
                assert abs(float(getattr(have, k)) - float(v)) < 1e-12, "the fixture's icp.yaml is not the shipped chain (%s)" % k
    for i in range(int(fix["n_clouds"])):
        c, keys = fix["cloud%d" % i], fix["keys%d" % i]
        for res in (0.5, 0.25):
            tag = "%d_res%03d" % (i, int(res * 100))
            p, idx = oracle.downsample(c, res, return_index=True)
            assert np.array_equal(p, fix["down" + tag]), "pcl.downsample(cloud%d, %g) differs from the oracle" % (i, res)
            assert np.array_equal(p, fix["downk_points" + tag]) and np.array_equal(keys[idx], fix["downk_keys" + tag]), \
                "pcl.downsample(cloud%d, keys, %g) differs from the oracle" % (i, res)
            n += 3
        for radius, min_points in ((1.0, 5), (0.5, 2)):
            got = oracle.remove_outlier(c, radius, min_points)
            assert np.array_equal(got, fix["outlier%d_r%03d_k%d" % (i, int(radius * 100), min_points)]), \
                "pcl.remove_outlier(cloud%d, %g, %d) differs from the oracle" % (i, radius, min_points)
            n += 1
    worst = 0.0
    for i in range(int(fix["n_jobs"])):
        s, t, g = fix["src%d" % i], fix["tgt%d" % i], fix["guess%d" % i]
        for md in (0.5, 3.0):
            ids, d2 = oracle.match(t, s, md)
            assert np.array_equal(ids, fix["match_ids%d_md%03d" % (i, int(md * 100))]), "pcl.match ids (job %d, %g) differ" % (i, md)
            assert np.array_equal(d2, fix["match_d2%d_md%03d" % (i, int(md * 100))]), "pcl.match distances (job %d, %g) differ" % (i, md)
            n += 2
        st, T, _ = oracle.icp(s, t, g, oracle.shipped_icp_params(precision=0))
        msg = str(fix["icp_msg%d" % i])
        assert msg == oracle.ICP_STATUS_MESSAGES[st], "ICP job %d: %r from libpointmatcher, %r from the oracle" % (
            i, msg, oracle.ICP_STATUS_MESSAGES[st])
        Tf = fix["icp_T%d" % i]
        if st != 0:
            assert np.array_equal(Tf, g), "ICP job %d failed but did not return the guess" % i
        else:
            d = max(abs(float(T[0, 2] - Tf[0, 2])), abs(float(T[1, 2] - Tf[1, 2])),
                    abs(float(np.arctan2(T[1, 0], T[0, 0]) - np.arctan2(Tf[1, 0], Tf[0, 0]))))
            assert d <= pose_tol, "ICP job %d: pose %.3e from libpointmatcher's (tolerance %.1e)" % (i, d, pose_tol)
            worst = max(worst, d)
        n += 1
    return n, worst
