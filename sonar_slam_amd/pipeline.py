"""Device-resident batched front end: B keyframes per step, everything stays in HBM.

One keyframe job = one sonar ping through CFAR+gate -> polar->Cartesian extraction -> feature
cloud (feature_extraction.py:223-238), plus one scan-match of a (source, target, guess) cloud
triple (slam.py:294-323 -> pcl.ICP.compute).  Inputs are uploaded once; ``run()`` only enqueues
kernels on the context's stream (no host round trip between the stages) and ``results()``
downloads the small per-job outputs.  This is the unit bench.py times and the farm shards.
"""
import ctypes as _C
import os

import numpy as np

from . import _lib as _L
from .cfar import _gate_u8


class KeyframeBatch(object):
    def __init__(self, ctx, geometry, cfar_params, alg, intensity_thr, icp_params, n_jobs,
                 max_points=16384, bit_masks=None, staged=None, points64=False):
        """bit_masks: the detections go from CFAR to the extraction as bit streams
        (sfe_cfar_u8_bits_batch_dev -> sfe_extract_points_bits_batch_dev) instead of 0/1 bytes; default:
        whenever the geometry allows it (polar_cols % 32 == 0).  Same points either way.
        staged: the extraction hands its clouds to the filters as float32 pairs + bounding boxes
        (sfe_extract_points_bits_staged_dev -> sfe_cloud_filter_staged_dev) instead of float64 points that the
        filters read back and cast; default: with bit masks and max_points <= 65536.  Same clouds either way.
        points64 (staged only): the float64 points (feature_extraction.py:238) are written too; without them
        points(j) extracts frame j again on demand."""
        self.ctx, self.geom = ctx, geometry
        self.alg = _L.ALG[alg]
        if alg == "OS":
            self.train_hs, self.guard_hs, self.k, self.tau = cfar_params
        else:
            self.train_hs, self.guard_hs, self.tau = cfar_params
            self.k = 0
        self.intensity_thr = _gate_u8(intensity_thr)   # img > thr on uint8 pixels (feature_extraction.py:224)
        self.icp_params = icp_params
        self.n = int(n_jobs)
        self.rows, self.cols = geometry.polar_rows, geometry.polar_cols
        self.cap = int(max_points)
        fb = self.n * self.rows * self.cols
        self.bit_masks = (self.cols % 32 == 0) if bit_masks is None else bool(bit_masks)
        if self.bit_masks and self.cols % 32:
            raise ValueError("bit_masks needs polar_cols % 32 == 0")
        self.wpf = (self.rows * self.cols + 31) // 32 + 1      # SFE_BITS_WORDS
        if staged is None and os.environ.get("SONARFE_STAGED") == "0":     # A/B inside one GPU call (tools/ab_stage.sh)
            staged = False
        self.staged = (self.bit_masks and self.cap <= 65536) if staged is None else bool(staged)
        if self.staged and not (self.bit_masks and self.cap <= 65536):
            raise ValueError("staged needs bit masks and max_points <= 65536")
        self.points64 = bool(points64) or not self.staged
        self._staged_ready = False                  # run_extract has left clouds for run_filter on the context
        self.d_img = ctx.alloc(fb)
        self.d_mask = ctx.alloc(self.n * self.wpf * 4 if self.bit_masks else fb)
        self.d_pts = ctx.alloc(self.n * self.cap * 16)
        self.d_cnt = ctx.alloc(self.n * 4)
        self.d_cnt.zero()                           # no extraction run yet = no points (results() checks them)
        self.d_cloud = self.d_cloud_cnt = None      # filtered float32 feature clouds (run_filter)
        self.d_src = self.d_tgt = self.d_guess = None
        self.d_T = ctx.alloc(self.n * 36)
        self.d_status = ctx.alloc(self.n * 4)
        self.d_iters = ctx.alloc(self.n * 4)
        self.src_off = self.tgt_off = None

    def upload_frames(self, frames):
        frames = np.ascontiguousarray(frames, np.uint8)
        assert frames.shape == (self.n, self.rows, self.cols)
        self.d_img.upload(frames)

    def upload_scan_pairs(self, sources, targets, guesses):
        """sources/targets: lists of N_i x 2 float32 clouds; guesses: n x 3 x 3."""
        assert len(sources) == len(targets) == len(guesses) == self.n
        so = np.zeros(self.n + 1, np.int32)
        to = np.zeros(self.n + 1, np.int32)
        so[1:] = np.cumsum([len(s) for s in sources])
        to[1:] = np.cumsum([len(t) for t in targets])
        src = np.ascontiguousarray(np.concatenate(sources), np.float32)
        tgt = np.ascontiguousarray(np.concatenate(targets), np.float32)
        g = np.ascontiguousarray(np.asarray(guesses, np.float32).reshape(self.n, 9))
        self.d_src, self.d_tgt, self.d_guess = (self.ctx.alloc(src.nbytes), self.ctx.alloc(tgt.nbytes),
                                                self.ctx.alloc(g.nbytes))
        self.d_src.upload(src)
        self.d_tgt.upload(tgt)
        self.d_guess.upload(g)
        self.src_off, self.tgt_off = so, to
        self.pair_evals_per_iter = int(sum(len(s) * len(t) for s, t in zip(sources, targets)))

    # ---- stages (enqueue only) ----
    def run_cfar(self):
        c = self.ctx
        if self.bit_masks:
            c._check(c.lib.sfe_cfar_u8_bits_batch_dev(c.handle, self.d_img.ptr, self.n, self.rows, self.cols,
                                                      self.alg, self.train_hs, self.guard_hs, self.k,
                                                      float(self.tau), self.intensity_thr, self.d_mask.ptr))
        else:
            c._check(c.lib.sfe_cfar_u8_batch_dev(c.handle, self.d_img.ptr, self.n, self.rows, self.cols, self.alg,
                                                 self.train_hs, self.guard_hs, self.k, float(self.tau),
                                                 self.intensity_thr, self.d_mask.ptr, None))

    def run_extract(self):
        c = self.ctx
        if self.staged:
            c._check(c.lib.sfe_extract_points_bits_staged_dev(c.handle, self.geom.handle, self.d_mask.ptr, self.n, self.cap,
                                                              self.d_pts.ptr, 1 if self.points64 else 0, self.d_cnt.ptr))
            self._staged_ready = True
            return
        fn = c.lib.sfe_extract_points_bits_batch_dev if self.bit_masks else c.lib.sfe_extract_points_batch_dev
        c._check(fn(c.handle, self.geom.handle, self.d_mask.ptr, self.n, self.cap, self.d_pts.ptr, self.d_cnt.ptr))

    def run_filter(self, resolution=0.5, radius=1.0, min_points=5):
        """pcl.downsample + pcl.remove_outlier on the extracted clouds, device to device
        (feature_extraction.py:241-249; defaults = config/feature.yaml)."""
        c = self.ctx
        if self.d_cloud is None:
            self.d_cloud = c.alloc(self.n * self.cap * 8)
            self.d_cloud_cnt = c.alloc(self.n * 4)
        if self.staged:
            # the staged clouds stay valid until something else on this context stages clouds (another batch, a per-cloud
            # pcl.downsample, the keyframe store): run_filter may repeat; the library refuses stale data (SFE_ERR_ARG),
            # and then the extraction is simply enqueued again in front
            for attempt in (0, 1):
                if not self._staged_ready:
                    self.run_extract()
                rc = c.lib.sfe_cloud_filter_staged_dev(c.handle, self.n, self.cap, float(resolution), float(radius),
                                                       int(min_points), self.d_cloud.ptr, self.d_cloud_cnt.ptr)
                if rc < 0 and attempt == 0 and b"no staged clouds" in c.lib.sfe_last_error(c.handle):
                    self._staged_ready = False
                    continue
                c._check(rc)
                break
            return
        c._check(c.lib.sfe_cloud_filter_batch_dev(c.handle, self.d_pts.ptr, self.d_cnt.ptr, self.n, self.cap,
                                                  float(resolution), float(radius), int(min_points),
                                                  self.d_cloud.ptr, self.d_cloud_cnt.ptr))

    def store_clouds(self, store, stamps=None, flags=1):
        """The filtered clouds of this batch (after run_filter) appended to a ``store.CloudStore``, device to device
        (sfe_cloud_store_put_batch_dev; enqueue only).  flags: store.NEGATE_Y = the SLAM node's convention
        (slam_ros.py:170).  -> handles [n]"""
        if self.d_cloud is None:
            raise _L.SonarFEError("store_clouds before run_filter")
        return store.put_batch_dev(self.d_cloud, self.d_cloud_cnt, self.n, self.cap, stamps=stamps, flags=flags)

    def cloud(self, j):
        """the filtered float32 feature cloud of frame j (after run_filter)"""
        self._check_frame(j)
        n = int(self.d_cloud_cnt.download(np.int32, 1, offset=4 * j)[0])
        if n < 0:
            raise _L.SonarFEError("frame %d: the octree of pcl.downsample is deeper than 24 levels "
                                  "(sfe_cloud_filter_batch_dev reports -1); use the per-cloud pcl.downsample" % j)
        return self.d_cloud.download(np.float32, 2 * n, offset=j * self.cap * 8).reshape(-1, 2)

    def _check_frame(self, j):
        """The resident path stores only the first `cap` points of a frame (sonarfe.h,
        sfe_extract_points_batch_dev): a frame above the cap is an error here, never a silently
        truncated cloud (the per-cloud API, Geometry.extract, retries with a larger buffer instead)."""
        n = int(self.d_cnt.download(np.int32, 1, offset=4 * j)[0])
        if n > self.cap:
            raise _L.SonarFEError("frame %d has %d points, more than the batch capacity %d: construct the "
                                  "KeyframeBatch with max_points >= %d" % (j, n, self.cap, n))
        return n

    def run_icp(self):
        c = self.ctx
        c._check(c.lib.sfe_icp_batch_dev(c.handle, _C.byref(self.icp_params), self.d_src.ptr,
                                         _L.ptr(self.src_off, _C.c_int32), self.d_tgt.ptr,
                                         _L.ptr(self.tgt_off, _C.c_int32), self.d_guess.ptr, self.n,
                                         self.d_T.ptr, self.d_status.ptr, self.d_iters.ptr))

    def run(self, filters=True):
        """One step: all stages of all n keyframes, enqueued back to back on the stream: the whole of
        FeatureExtraction.callback (CFAR + gate, remap + nonzero + px->m, downsample, outlier filter with
        the config/feature.yaml defaults) and one scan match per keyframe."""
        self.run_cfar()
        self.run_extract()
        if filters:
            self.run_filter()
        self.run_icp()

    def results(self):
        """Per-job outputs.  Raises if any frame overflowed the point capacity (its cloud, and everything
        computed from it, would be truncated) or could not be filtered."""
        self.ctx.sync()
        counts = self.d_cnt.download(np.int32, self.n)
        if counts.size and int(counts.max()) > self.cap:
            f = int(counts.argmax())
            raise _L.SonarFEError("frame %d has %d points, more than the batch capacity %d: construct the "
                                  "KeyframeBatch with max_points >= %d" % (f, counts[f], self.cap, counts[f]))
        if self.d_cloud_cnt is not None:
            cc = self.d_cloud_cnt.download(np.int32, self.n)
            if (cc < 0).any():
                raise _L.SonarFEError("frame %d: the octree of pcl.downsample is deeper than 24 levels"
                                      % int(np.argmin(cc)))
        return {
            "counts": counts,
            "cloud_counts": cc if self.d_cloud_cnt is not None else None,
            "T": self.d_T.download(np.float32, self.n * 9).reshape(self.n, 3, 3),
            "status": self.d_status.download(np.int32, self.n),
            "iters": self.d_iters.download(np.int32, self.n),
        }

    def points(self, j):
        """the float64 points of frame j (feature_extraction.py:235-238), np.nonzero order"""
        n = self._check_frame(j)
        if not self.points64:
            # the staged step keeps float32 pairs only: frame j's bit stream goes through the float64 extraction once more
            c = self.ctx
            tmp_p, tmp_n = c.alloc(self.cap * 16), c.alloc(4)
            try:
                c._check(c.lib.sfe_extract_points_bits_batch_dev(c.handle, self.geom.handle,
                                                                 _C.c_void_p(self.d_mask.ptr.value + j * self.wpf * 4),
                                                                 1, self.cap, tmp_p.ptr, tmp_n.ptr))
                c.sync()
                assert int(tmp_n.download(np.int32, 1)[0]) == n
                return tmp_p.download(np.float64, 2 * n).reshape(n, 2)
            finally:
                tmp_p.free()
                tmp_n.free()
        return self.d_pts.download(np.float64, 2 * n, offset=j * self.cap * 16).reshape(n, 2)

    def mask(self, j):
        """the 0/1 detection mask of frame j (peaks of feature_extraction.py:223-224)"""
        sz = self.rows * self.cols
        if self.bit_masks:
            w = self.d_mask.download(np.uint32, self.wpf, offset=j * self.wpf * 4)
            if w[-1] != 0:
                raise _L.SonarFEError("frame %d: the pad word of the bit stream is not 0" % j)
            return np.unpackbits(w.view(np.uint8), bitorder="little")[:sz].reshape(self.rows, self.cols)
        return self.d_mask.download(np.uint8, sz, offset=j * sz).reshape(self.rows, self.cols)

    def free(self):
        for b in (self.d_img, self.d_mask, self.d_pts, self.d_cnt, self.d_cloud, self.d_cloud_cnt, self.d_src,
                  self.d_tgt, self.d_guess,
                  self.d_T, self.d_status, self.d_iters):
            if b is not None:
                b.free()


class ScanMatchBatch(object):
    """Device-resident scan-match jobs with a job table (sfe_icp_jobs_dev): the distinct clouds are uploaded once and
    a job names a (source, target) pair by index, so the <= 30 guesses of ``compute_icp_with_cov`` on one pair
    (slam.py:346-358), or one target matched against many sources, share their clouds -- and the target's preparation."""

    def __init__(self, ctx, icp_params, sources, targets, jobs, guesses):
        """sources / targets: lists of distinct N_i x 2 float32 clouds; jobs: [(source index, target index)];
        guesses: len(jobs) x 3 x 3."""
        self.ctx, self.icp_params = ctx, icp_params
        self.n = len(jobs)
        assert len(guesses) == self.n and self.n > 0
        so = np.concatenate([[0], np.cumsum([len(s) for s in sources])]).astype(np.int64)
        to = np.concatenate([[0], np.cumsum([len(t) for t in targets])]).astype(np.int64)
        self.jobs4 = np.ascontiguousarray([(so[a], so[a + 1] - so[a], to[b], to[b + 1] - to[b]) for a, b in jobs], np.int32)
        src = np.ascontiguousarray(np.concatenate(sources), np.float32)
        tgt = np.ascontiguousarray(np.concatenate(targets), np.float32)
        g = np.ascontiguousarray(np.asarray(guesses, np.float32).reshape(self.n, 9))
        self.d_src, self.d_tgt, self.d_guess = ctx.alloc(src.nbytes), ctx.alloc(tgt.nbytes), ctx.alloc(g.nbytes)
        self.d_src.upload(src)
        self.d_tgt.upload(tgt)
        self.d_guess.upload(g)
        self.d_T = ctx.alloc(self.n * 36)
        self.d_status = ctx.alloc(self.n * 4)
        self.d_iters = ctx.alloc(self.n * 4)

    def run(self):
        """enqueue only"""
        c = self.ctx
        c._check(c.lib.sfe_icp_jobs_dev(c.handle, _C.byref(self.icp_params), self.d_src.ptr, self.d_tgt.ptr,
                                        _L.ptr(self.jobs4, _C.c_int32), self.d_guess.ptr, self.n, self.d_T.ptr,
                                        self.d_status.ptr, self.d_iters.ptr))

    def results(self):
        self.ctx.sync()
        return {"T": self.d_T.download(np.float32, self.n * 9).reshape(self.n, 3, 3),
                "status": self.d_status.download(np.int32, self.n),
                "iters": self.d_iters.download(np.int32, self.n)}

    def free(self):
        for b in (self.d_src, self.d_tgt, self.d_guess, self.d_T, self.d_status, self.d_iters):
            if b is not None:
                b.free()
