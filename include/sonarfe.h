/*
 * sonarfe.h -- C ABI of libsonarfe.so, the MI355X (gfx950) sonar front-end.
 *
 * This is the drop-in boundary for the two pybind11 modules of the reference
 * (there is no C ABI in the reference; these entry points are what a ctypes /
 * cgo / JNI binding of that path binds instead):
 *
 *   bruce_slam.cfar  (bruce_slam/src/bruce_slam/cpp/cfar.cpp:194-204)
 *       ca/soca/goca/os, ca2/soca2/goca2/os2        -> sfe_cfar_u8 / sfe_cfar_f32
 *   bruce_slam.pcl   (bruce_slam/src/bruce_slam/cpp/pcl.cpp:176-214)
 *       match                                       -> sfe_match
 *       remove_outlier                              -> sfe_remove_outlier
 *       downsample (both overloads)                 -> sfe_downsample
 *       ICP.loadFromYaml / compute / getCovariance  -> sfe_icp_* (params parsed host-side)
 *   feature_extraction.py:223-238 (CFAR gate, cv2.remap, nonzero, px->m)
 *                                                   -> sfe_geom_*, sfe_remap_u8, sfe_extract_points
 *
 * Conventions
 *   - plain pointers and sizes only; images are row-major (C order, numpy default):
 *     rows = range bins, cols = beams.  Point clouds are N x 2 float32 row-major.
 *     Transforms are 3 x 3 float32 row-major (Pose2 matrix).
 *   - entry points without a _dev suffix take HOST pointers (they copy in, run the
 *     HIP kernels, copy out) and are what the Python shims call.  *_dev entry points
 *     take DEVICE pointers obtained from sfe_malloc and only enqueue work on the
 *     context's stream (call sfe_sync to wait).
 *   - return value: 0 = success; > 0 = ICP convergence-class status (see
 *     SFE_ICP_*; the reference turns these into (what(), guess)); < 0 = hard error
 *     (bad argument, HIP failure) with text in sfe_last_error().  There is no CPU
 *     fallback anywhere: without a gfx950 device every compute entry point fails.
 *   - the library reads a number of SFE_* environment variables that switch between measured alternatives of the same
 *     computation (every setting returns identical results).  They are measurement tools, not part of this interface:
 *     DESIGN.md, appendix A lists them.
 *   - one sfe_ctx = one device + one HIP stream + its scratch; a ctx is not
 *     re-entrant (the reference's pybind calls hold the GIL and its ICP object is
 *     stateful, SURVEY 8b "Threading"); use one ctx per worker thread/process.
 */
#ifndef SONARFE_H
#define SONARFE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sfe_ctx sfe_ctx;
typedef struct sfe_geom sfe_geom;

/* hard-error codes (negative) */
#define SFE_ERR_ARG (-1)
#define SFE_ERR_HIP (-2)
#define SFE_ERR_NODEV (-3)
#define SFE_ERR_CAP (-4) /* caller-provided output capacity too small */

/* CFAR variants (cfar.cpp:10,30,53,76) */
#define SFE_CFAR_CA 0
#define SFE_CFAR_SOCA 1
#define SFE_CFAR_GOCA 2
#define SFE_CFAR_OS 3

/* ICP convergence-class statuses; messages are libpointmatcher's what() strings */
#define SFE_ICP_OK 0
#define SFE_ICP_NO_OUTLIER 1 /* "no outlier to filter" */
#define SFE_ICP_NO_POINT 2   /* "ErrorMnimizer: no point to minimize" */
#define SFE_ICP_NAN_ROT 3    /* "abs rotation norm not a number" */
#define SFE_ICP_NAN_TRANS 4  /* "abs translation norm not a number" */
#define SFE_ICP_SINGULAR 5   /* point-to-plane normal system not positive definite */
#define SFE_ICP_SPLIT_TIMEOUT 6 /* *_dev entry points only: the workgroups sharing one large job were not resident together (sfe_icp_set_tuning bit 4) */

/* ---- library / context ------------------------------------------------- */
const char *sfe_version(void);
int sfe_device_count(int *count);
int sfe_ctx_create(int device, sfe_ctx **out);
void sfe_ctx_destroy(sfe_ctx *ctx);
const char *sfe_last_error(sfe_ctx *ctx); /* ctx may be NULL: error of the last failed create */
int sfe_sync(sfe_ctx *ctx);
int sfe_device_name(sfe_ctx *ctx, char *buf, int cap);

/* device memory on the ctx's device, for resident pipelines (*_dev entry points) */
int sfe_malloc(sfe_ctx *ctx, size_t bytes, void **dptr);
int sfe_free(sfe_ctx *ctx, void *dptr);
int sfe_memcpy_h2d(sfe_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int sfe_memcpy_d2h(sfe_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);
int sfe_memset(sfe_ctx *ctx, void *dst_dev, int value, size_t bytes);

/* Streamed inputs (feature_extraction.py:196-217: every ping arrives from the host).  Pinned host memory and uploads on
 * the context's copy stream, next to the kernels on its main stream:
 *   sfe_memcpy_h2d_async  enqueue-only upload from a sfe_host_alloc block
 *   sfe_stream_fence(0)   kernels enqueued from now on wait for every upload enqueued so far
 *   sfe_stream_fence(1)   uploads enqueued from now on wait for every kernel enqueued so far (their input buffer is free)
 *   sfe_stream_fence(2)   the host waits for the uploads */
int sfe_host_alloc(sfe_ctx *ctx, size_t bytes, void **hptr);
int sfe_host_free(sfe_ctx *ctx, void *hptr);
int sfe_memcpy_h2d_async(sfe_ctx *ctx, void *dst_dev, const void *src_pinned, size_t bytes);
int sfe_stream_fence(sfe_ctx *ctx, int what);

/* debug: the first `bytes` of the library's scratch buffer `slot` (intermediate results of the kernels) */
int sfe_debug_read_scratch(sfe_ctx *ctx, int slot, void *dst_host, size_t bytes);

/* HIP-event stopwatch on the ctx's stream (used by bench.py for per-kernel time) */
int sfe_timer_start(sfe_ctx *ctx);
int sfe_timer_stop(sfe_ctx *ctx, float *elapsed_ms); /* records, syncs, returns ms */

/* ---- CFAR: replaces bruce_slam.cfar (cfar.cpp:10-192) ------------------- */
/*
 * img: rows x cols.  train_hs/guard_hs/tau/k as passed by CFAR.detect
 * (CFAR.py:35-40,123-127).  intensity_thr >= 0 fuses feature_extraction.py:224
 * (mask &= img > thr); pass -1 for the plain cfar.* result.  thr_out (nullable)
 * receives the float threshold map of the *2 variants (cfar.cpp:98-192).
 * k is only read for SFE_CFAR_OS and must satisfy 0 <= k < 2*train_hs.
 */
int sfe_cfar_u8(sfe_ctx *ctx, const uint8_t *img, int rows, int cols, int alg, int train_hs,
                int guard_hs, int k, double tau, int intensity_thr, uint8_t *mask_out,
                float *thr_out);
/* float images (the pybind caster accepts any numeric dtype): exact float-sum semantics */
int sfe_cfar_f32(sfe_ctx *ctx, const float *img, int rows, int cols, int alg, int train_hs,
                 int guard_hs, int k, double tau, uint8_t *mask_out, float *thr_out);
/* batched, device-resident: n_frames images of rows x cols back to back */
int sfe_cfar_u8_batch_dev(sfe_ctx *ctx, const uint8_t *d_img, int n_frames, int rows, int cols,
                          int alg, int train_hs, int guard_hs, int k, double tau,
                          int intensity_thr, uint8_t *d_mask, float *d_thr);
/* The same detections as a bit stream, the form the batched extraction consumes: per frame
 * SFE_BITS_WORDS(rows*cols) uint32 words, bit (iy*cols + ix) of the stream (LSB first) = mask[iy][ix],
 * the last word of a frame is padding and written as 0.  For the windows of the register-ring kernel and
 * cols % 32 == 0 the kernel stores the bits itself (the 0/1 byte mask of CFAR.detect, feature_extraction.py:223,
 * never exists in memory); every other call runs the byte kernels into scratch and packs them.
 * d_bits: n_frames * SFE_BITS_WORDS(rows*cols) words, 4-byte aligned. */
#define SFE_BITS_WORDS(px) (((px) + 31) / 32 + 1)
int sfe_cfar_u8_bits_batch_dev(sfe_ctx *ctx, const uint8_t *d_img, int n_frames, int rows, int cols,
                               int alg, int train_hs, int guard_hs, int k, double tau,
                               int intensity_thr, uint32_t *d_bits);
/* tuning / A-B knob for the ring kernel: output rows per thread (0 = default) and
 * variant (0 = auto, 1 = force generic kernel, 2 = force ring kernel with prefetch depth 4, 3 = ring kernel depth 13) */
int sfe_cfar_set_tuning(sfe_ctx *ctx, int tile_rows, int variant);

/* ---- polar -> Cartesian: feature_extraction.py:134-173,226-238 --------- */
/*
 * A geometry = the (map_x, map_y) pair generate_map_xy builds (float32,
 * cart_rows x cart_cols, host pointers) for a polar image of polar_rows x
 * polar_cols, plus the metric extent used by the px->m step (width, height in m).
 * Creation uploads the maps and pre-decodes OpenCV's fixed-point coordinates.
 */
int sfe_geom_create(sfe_ctx *ctx, const float *map_x, const float *map_y, int cart_rows,
                    int cart_cols, int polar_rows, int polar_cols, double width, double height,
                    sfe_geom **out);
void sfe_geom_destroy(sfe_geom *g);
/* cv2.remap(src, map_x, map_y, cv2.INTER_LINEAR) for a uint8 image (BORDER_CONSTANT 0) */
int sfe_remap_u8(sfe_ctx *ctx, sfe_geom *g, const uint8_t *src, uint8_t *dst);
/* the same on device pointers (enqueue only): d_src polar_rows x polar_cols, d_dst cart_rows x cart_cols */
int sfe_remap_u8_dev(sfe_ctx *ctx, sfe_geom *g, const uint8_t *d_src, uint8_t *d_dst);
/* cv2.applyColorMap(cv2.remap(src, map_x, map_y, cv2.INTER_LINEAR), colormap) in one pass: the publishable bgr8 feature
 * image of feature_extraction.py:226-228 (dst_bgr: cart_rows x cart_cols x 3, B G R per pixel; the _dev form needs it
 * 4-byte aligned).  Only cv2.COLORMAP_JET (2), the one the node uses.  sfe_colormap_lut: the 256 x 3 BGR table itself
 * (applyColorMap(x, 2)[..] == lut[x]).  OpenCV's table restated, parity unpinned. */
#define SFE_COLORMAP_JET 2
int sfe_colormap_lut(int colormap, uint8_t *lut_bgr);
int sfe_remap_u8_colormap(sfe_ctx *ctx, sfe_geom *g, const uint8_t *src, int colormap, uint8_t *dst_bgr);
int sfe_remap_u8_colormap_dev(sfe_ctx *ctx, sfe_geom *g, const uint8_t *d_src, int colormap, uint8_t *d_dst_bgr);
/*
 * remap(mask) -> np.nonzero -> px->m in one call (feature_extraction.py:231-238).
 * mask: polar_rows x polar_cols uint8 0/1 (host).  Outputs (host, nullable):
 * rc_out int64 [cap x 2] = (row, col) in row-major order, pts_out float64 [cap x 2] =
 * (y_forward, x_lateral) in metres.  *n_out = number of points (may exceed cap -> SFE_ERR_CAP).
 */
int sfe_extract_points(sfe_ctx *ctx, sfe_geom *g, const uint8_t *mask, int64_t cap,
                       int64_t *rc_out, double *pts_out, int64_t *n_out);
/* A-B knob of the extraction: 0 = binary masks go through the inverse map (walk the set polar pixels, evaluate only the
 * canvas pixels that tap them; default) -- bit-stream batches without a canvas bitmap in HBM (records merged in LDS), frames
 * beyond that path's capacities and byte masks through the canvas bitmap; 1 = dense pass over the whole canvas for every
 * frame (what non-binary masks take anyway); 2 = inverse map through the canvas bitmap for every frame.  Identical results. */
int sfe_extract_set_tuning(sfe_ctx *ctx, int variant);
/* device-resident batch: per frame f, points go to d_pts + f*cap*2 (float64), count to d_counts[f]
 * (count is the true number even if it exceeds cap; only the first cap points are stored) */
int sfe_extract_points_batch_dev(sfe_ctx *ctx, sfe_geom *g, const uint8_t *d_mask, int n_frames,
                                 int64_t cap, double *d_pts, int32_t *d_counts);
/* the same from the bit streams of sfe_cfar_u8_bits_batch_dev (polar_cols % 32 == 0 required) */
int sfe_extract_points_bits_batch_dev(sfe_ctx *ctx, sfe_geom *g, const uint32_t *d_bits, int n_frames,
                                      int64_t cap, double *d_pts, int32_t *d_counts);
/* The same, handing the clouds to the resident cloud filters without the float64 round trip (round 6): next to d_counts the
 * call leaves every frame's points as float32 pairs -- the float64 metres of feature_extraction.py:235-238 rounded once, the
 * cast pybind makes when `points` enters pcl.downsample (pcl.cpp:128-131) -- and the frame's bounding box in the context's
 * staging area, where sfe_cloud_filter_staged_dev (below) picks them up; it must be the next filter call on this context with
 * the same n_frames and cap (anything else that stages clouds in between makes it fail with SFE_ERR_ARG, never run on stale
 * data).  d_pts [n_frames][cap][2] float64 is still required: with want_points64 != 0 it receives every frame's points like
 * sfe_extract_points_bits_batch_dev; with 0 only the frames that leave the fast path land there (dense frames beyond its
 * per-frame capacities; their float32 form is made from it) and the rest of it is left untouched -- 8 instead of 24 bytes
 * written per point.  cap <= 65536 (the filters' limit).  Enqueue only. */
int sfe_extract_points_bits_staged_dev(sfe_ctx *ctx, sfe_geom *g, const uint32_t *d_bits, int n_frames,
                                       int64_t cap, double *d_pts, int want_points64, int32_t *d_counts);

/* ---- point clouds: replaces bruce_slam.pcl (pcl.cpp:54-74,161-212) ------ */
typedef struct sfe_icp_params {
    float matcher_max_dist;  /* KDTreeMatcher.maxDist (icp.yaml:9) */
    int use_max_dist_filter; /* MaxDistOutlierFilter listed (icp.yaml:12) */
    float max_dist_filter;   /*   .maxDist (icp.yaml:13) */
    int use_trimmed_filter;  /* TrimmedDistOutlierFilter listed (icp.yaml:14) */
    float trim_ratio;        /*   .ratio (icp.yaml:15) */
    int minimizer;           /* 0 PointToPointErrorMinimizer (icp.yaml:20), 1 2-D PointToPlane (icp.yaml:18-19) */
    int max_iter;            /* CounterTransformationChecker.maxIterationCount (icp.yaml:24) */
    int use_diff_checker;    /* DifferentialTransformationChecker listed (icp.yaml:25) */
    float min_diff_rot;      /*   .minDiffRotErr (icp.yaml:26) */
    float min_diff_trans;    /*   .minDiffTransErr (icp.yaml:27) */
    int smooth_len;          /*   .smoothLength (icp.yaml:28) */
    int normals_knn;         /* point-to-plane only: neighbours (incl. self) for PCA normals */
} sfe_icp_params;

/* pcl.match(ref, in, knn=1, max_dist) (pcl.cpp:161-174): ids int32 [n_in] (-1 = none),
 * d2 float [n_in] squared distance (inf = none).  Ties -> lowest reference index. */
int sfe_match(sfe_ctx *ctx, const float *ref, int n_ref, const float *in, int n_in, float max_dist,
              int32_t *ids, float *d2);
/* the same for knn >= 1 neighbours (pcl.cpp:161-174 hands knn to KDTreeMatcher): ids / d2 are [knn x n_in] row-major
 * (the IntMatrix / Matrix pybind returns), row j = the (j+1)-th nearest reference point in ascending (d2, index)
 * order; -1 / inf where fewer than j+1 reference points lie within max_dist. */
int sfe_match_knn(sfe_ctx *ctx, const float *ref, int n_ref, const float *in, int n_in, int knn, float max_dist,
                  int32_t *ids, float *d2);
/* The densities pcl.density_filter computes (pcl.cpp:76-126: libpointmatcher SurfaceNormalDataPointsFilter{knn,
 * keepDensities}): per point the knn nearest points incl. itself, density = knn / ((4/3) pi r^3), r = the largest
 * distance of a neighbour from the neighbours' mean.  knn > n is an error (libnabo refuses it).  The second stage of
 * the reference's function (MaxDensityDataPointsFilter: sequential, std::rand) stays on the host, see pcl.py. */
int sfe_knn_density(sfe_ctx *ctx, const float *pts, int n, int knn, float *dens_out);
/* pcl.remove_outlier(points, radius, min_points) (pcl.cpp:54-74): order preserved */
int sfe_remove_outlier(sfe_ctx *ctx, const float *pts, int n, double radius, int min_points,
                       float *out, int *n_out);
/* pcl.downsample(points, resolution) (pcl.cpp:128-159): libpointmatcher OctreeGridDataPointsFilter
 * (maxSizeByNode = resolution, medoid per leaf, leaves in depth-first order).  out: up to n points,
 * out_idx (nullable): their indices in the input (the descriptor overload gathers with them). */
int sfe_downsample(sfe_ctx *ctx, const float *pts, int n, float resolution, float *out,
                   int32_t *out_idx, int *n_out);
/* pcl.ICP.compute(source, target, guess) (pcl.cpp:198-212).  Returns SFE_ICP_* (>= 0) or a
 * hard error (< 0).  On a nonzero status T_out = guess (pcl.cpp:203,207-210). */
int sfe_icp_compute(sfe_ctx *ctx, const sfe_icp_params *p, const float *src, int n_src,
                    const float *tgt, int n_tgt, const float *guess9, float *T_out9, int *iters);
/* many guesses on one cloud pair: the loop of SLAM.compute_icp_with_cov (slam.py:346-358) */
int sfe_icp_compute_guesses(sfe_ctx *ctx, const sfe_icp_params *p, const float *src, int n_src,
                            const float *tgt, int n_tgt, const float *guesses9, int n_guesses,
                            float *T_out9, int32_t *status, int32_t *iters);
/* many INDEPENDENT scan pairs in one launch (the job farm's unit; host pointers): clouds concatenated,
 * job j = src[src_off[j]..src_off[j+1]) against tgt[tgt_off[j]..tgt_off[j+1]) with guess 9*j.  Per job:
 * T_out9 (= the guess on a nonzero status, like pcl.cpp:203,207-210), status (SFE_ICP_*), iterations. */
int sfe_icp_compute_pairs(sfe_ctx *ctx, const sfe_icp_params *p, const float *src, const int32_t *src_off,
                          const float *tgt, const int32_t *tgt_off, const float *guesses9, int n_jobs,
                          float *T_out9, int32_t *status, int32_t *iters);
/* the same with an explicit job table, so that jobs may SHARE clouds (many guesses on one pair, one target matched
 * against many sources: the loop of slam.py:346-358 and the job farm's chunks): src / tgt = all clouds back to
 * back (n_src_pts / n_tgt_pts points in total), jobs4 = n_jobs x (src_start, n_src, tgt_start, n_tgt) in points.
 * Jobs naming the same target slice share one target preparation (sort, strip table, normals). */
int sfe_icp_compute_jobs(sfe_ctx *ctx, const sfe_icp_params *p, const float *src, int n_src_pts, const float *tgt,
                         int n_tgt_pts, const int32_t *jobs4, const float *guesses9, int n_jobs, float *T_out9,
                         int32_t *status, int32_t *iters);
/* A-B knob for the ICP kernels.  bit 2: 0 = strip-sweep exact NN search (default; targets beyond 8192
 * points are walked through L2 instead of LDS), 1 = brute-force tile scan for everything.
 * Brute-force only: bit 0: 0 = packed fp32 NN loop, 1 = scalar fp32; bit 1: 0 = 64-VGPR build,
 * 2 workgroups/CU, 1 = 128-VGPR build.  bit 3 (strip sweep only): the clouds and guesses handed to the ICP
 * entry points are final, i.e. no work enqueued earlier on this context still writes them; the preparation of
 * the targets (sort, strip table, normals) then runs on a side stream next to that earlier work and only the
 * iteration kernel waits for both.  Leave it clear when a preceding call on the context produces the clouds.
 * bit 4: never share one large job between several workgroups.  Sharing (jobs of >= 8192 queries on a target beyond
 * 8192 points, when the call holds nothing else and shares x jobs <= CUs) needs every share resident at the same
 * time, which only a device this context has to itself can promise; a share that waits 0.5 s for the others gives up
 * and the job reports SFE_ICP_SPLIT_TIMEOUT.  The host-pointer entry points (sfe_icp_compute*) then run the call again
 * with this bit set, so their callers never see that status; callers of the enqueue-only *_dev entry points that
 * share the device set the bit themselves or repeat the call with it when they read status 6.
 * All variants return identical results. */
int sfe_icp_set_tuning(sfe_ctx *ctx, int variant);
/* profile mode of the strip-sweep ICP kernel: enable/disable, and read the 96 values of the last profiled launch
 * (buffer of 96 long long; batches of more jobs than CUs with LDS-resident targets only).  [80..84] are counted
 * over the WHOLE launch: [80] candidate distance evaluations by the lane-per-query tiers, [81] by the cooperative
 * tier, [82] witness evaluations, [83] lower-bound probes, [84] ICP iterations run (all jobs).  The rest are
 * per-phase cycle counters of workgroup 0, summed over iterations.  Cycles: [0] setup incl. the
 * query sort, [1] first search pass (own strip), [3] trimmed quantile, [4] reduction, [5] solve, [6] later
 * search passes, [7] cooperative tier, [8] finite/exact census.  Counts: [9] search rounds, [10] queries
 * handed to the cooperative tier, [11] queries handed to the second pass, [12] cooperative trips.
 * [16 + 2i], [17 + 2i]: search cycles and (cap bits << 32 | exact matches) of iteration i < 32. */
int sfe_icp_get_profile(sfe_ctx *ctx, int enable, long long *cycles16);
/* independent jobs, device-resident: clouds concatenated, job j uses
 * src[src_off[j]..src_off[j+1]) and tgt[tgt_off[j]..tgt_off[j+1]) (offsets in points, host
 * arrays of n_jobs+1), guess d_guess9 + 9*j; outputs d_T9 (9 floats), d_status, d_iters per job */
int sfe_icp_batch_dev(sfe_ctx *ctx, const sfe_icp_params *p, const float *d_src,
                      const int32_t *src_off, const float *d_tgt, const int32_t *tgt_off,
                      const float *d_guess9, int n_jobs, float *d_T9, int32_t *d_status,
                      int32_t *d_iters);

/* the same with a job table: jobs4 (host, n_jobs x 4) = (src_start, n_src, tgt_start, n_tgt) in points into the resident
 * clouds, so that jobs may share clouds -- the <= 30 guesses of compute_icp_with_cov on one pair (slam.py:346-358), one
 * target matched against many sources; jobs naming the same target slice share its preparation */
int sfe_icp_jobs_dev(sfe_ctx *ctx, const sfe_icp_params *p, const float *d_src, const float *d_tgt,
                     const int32_t *jobs4, const float *d_guess9, int n_jobs, float *d_T9, int32_t *d_status,
                     int32_t *d_iters);

/* Device-resident tail of FeatureExtraction.callback (feature_extraction.py:241-249) for a batch:
 * pcl.downsample(points, resolution) then pcl.remove_outlier(points, radius, min_points) on the
 * float64 clouds sfe_extract_points_batch_dev left in HBM (d_pts [n_frames][cap][2], d_counts),
 * without a host round trip.  resolution <= 0 skips the downsample (:241), min_points <= 1 the
 * outlier filter (:245).  Output: float32 clouds d_out [n_frames][cap][2] (what pybind hands back),
 * d_out_counts[f] points each (-1: the frame's octree is deeper than 24 levels).  cap <= 65536. */
int sfe_cloud_filter_batch_dev(sfe_ctx *ctx, const double *d_pts, const int32_t *d_counts, int n_frames,
                               int64_t cap, float resolution, double radius, int min_points,
                               float *d_out, int32_t *d_out_counts);
/* The same filters (feature_extraction.py:241-249; pcl.cpp:128-141, 54-74) on the clouds sfe_extract_points_bits_staged_dev
 * left staged on this context: same arguments minus the float64 points, same outputs, bit for bit (the float32 points are
 * those the cast of the call above produces; a bounding box does not depend on the order it is taken in). */
int sfe_cloud_filter_staged_dev(sfe_ctx *ctx, int n_frames, int64_t cap, float resolution, double radius,
                                int min_points, float *d_out, int32_t *d_out_counts);

/*
 * ONE ping through the whole of FeatureExtraction.callback (feature_extraction.py:223-249) in one call, for the live
 * node (one ping per callback: latency, not throughput): img (host, polar_rows x polar_cols uint8) goes up once through
 * pinned staging; CFAR + gate (alg / train_hs / guard_hs / k / tau / intensity_thr as for sfe_cfar_u8), remap +
 * nonzero + px->m, pcl.downsample(resolution) and pcl.remove_outlier(radius, min_points) run back to back on the
 * stream (resolution <= 0 / min_points <= 1 skip a filter like :241 / :245); the float32 cloud comes down once:
 * one stream synchronisation per ping.  cloud_out: host [cap x 2] float32 (y_forward, x_lateral), *n_out points;
 * *n_raw_out (nullable) = points before the filters.  vis_out (nullable, host cart_rows x cart_cols) additionally
 * receives cv2.remap(img) for the visualisation (:226).  Returns SFE_ERR_CAP when more than cap points were extracted
 * (*n_raw_out says how many: retry with a larger cap; cap <= 65536); *n_out = -1 when the cloud's octree is deeper
 * than the resident filter's 24 levels (take the per-cloud entry points).  Results are bit-identical to the chain of
 * sfe_cfar_u8 -> sfe_extract_points -> sfe_downsample -> sfe_remove_outlier.
 */
int sfe_feature_extract_ping(sfe_ctx *ctx, sfe_geom *g, const uint8_t *img, int alg, int train_hs, int guard_hs,
                             int k, double tau, int intensity_thr, float resolution, double radius, int min_points,
                             int64_t cap, float *cloud_out, int32_t *n_out, int32_t *n_raw_out, uint8_t *vis_out);

/* ---- device-resident keyframe clouds: the hand-off feature node -> SLAM node without PCIe ------------------
 * Reference flow (every cloud crosses the process boundary and is rebuilt in numpy):
 *   feature_extraction.py:175-193  publish_features           cloud -> PointCloud2 xyz32
 *   slam_ros.py:169-170            SLAM_callback              xyz -> [x, -z] (float64 array of float32 values:
 *                                                             ros_numpy pointcloud2_to_xyz_array)
 *   slam_objects.py:178-198        Keyframe.transform_points  points.dot(T[:2,:2].T) + T[:2,2], T float32
 *   slam.py:229-292                SLAM.get_points            transform, concatenate, pcl.downsample
 *   slam.py:294-323                SLAM.compute_icp           pcl.ICP.compute(source, target, guess)
 *   slam.py:389-424                SLAM.get_overlap           transform, pcl.match, count ids != -1
 * A store keeps the clouds in HBM: one pool of float2 points and a slot table (offset, count) per cloud.  A cloud is
 * named by its handle (slot index, handed out in order); slots are released in stack order (sfe_cloud_store_truncate):
 * keyframe clouds stay for the session, the target clouds get_points builds are dropped after their scan match.
 * Clouds are appended by kernels that read their sizes from device memory, so the sizes reach the host lazily: the
 * entry points that need them (meta, read, get_points, the scan match) copy the new slot-table entries down once
 * (a few bytes per cloud, one stream synchronisation) -- never the points.  A cloud's count is < 0 when its producer
 * failed: -1 the resident downsample refused it (octree deeper than 24 levels), -3 the pool was full. */
typedef struct sfe_cloud_store sfe_cloud_store;
#define SFE_STORE_NEGATE_Y 1   /* put: store (x, -y), what slam_ros.py:170 makes of the feature message */
#define SFE_PING_VIS_JET 4     /* sfe_feature_extract_ping_store: vis_out is the bgr8 image applyColorMap(remap(img), JET)
                                  (cart_rows x cart_cols x 3, feature_extraction.py:226-228) instead of the grey remap */
#define SFE_STORE_F32_POINTS 2 /* get_points / overlap: the caller's keyframe clouds are float32 numpy arrays (sgemm
                                  rounding: fma(p1, r1, p0 * r0) + t in float) instead of the SLAM node's float64 ones
                                  (products and sums in double, rounded to float32 at the pybind boundary) */
int sfe_cloud_store_create(sfe_ctx *ctx, int64_t capacity_points, int32_t max_clouds, sfe_cloud_store **out);
void sfe_cloud_store_destroy(sfe_cloud_store *s);
int sfe_cloud_store_count(sfe_cloud_store *s); /* slots in use = the next handle */
/* one cloud from the host (a feature message that arrived over the wire), enqueue only */
int sfe_cloud_store_put(sfe_ctx *ctx, sfe_cloud_store *s, int64_t stamp, const float *pts, int n, int32_t *handle_out);
/* n_frames clouds straight from sfe_cloud_filter_batch_dev's outputs (d_clouds [n_frames][cap][2] float32, d_counts),
 * device to device, enqueue only; stamps (host, nullable) are kept with the slots; handles_out (host, nullable) */
int sfe_cloud_store_put_batch_dev(sfe_ctx *ctx, sfe_cloud_store *s, const int64_t *stamps, const float *d_clouds,
                                  const int32_t *d_counts, int n_frames, int64_t cap, int flags, int32_t *handles_out);
/* slot table of clouds first .. first + n - 1 (each output nullable); synchronises if entries are new */
int sfe_cloud_store_meta(sfe_ctx *ctx, sfe_cloud_store *s, int32_t first, int32_t n, int64_t *stamps, int64_t *offsets,
                         int32_t *counts);
/* the points of one cloud, for whoever needs them on the host (the PointCloud2 of publish_features, rviz, mapping) */
int sfe_cloud_store_read(sfe_ctx *ctx, sfe_cloud_store *s, int32_t handle, float *out, int cap, int *n_out);
/* release every slot >= n_slots */
int sfe_cloud_store_truncate(sfe_ctx *ctx, sfe_cloud_store *s, int32_t n_slots);
/* SLAM.get_points(frames, ref_frame) for n_jobs target clouds at once (slam.py:229-292): job j takes the clouds
 * handles[j*m .. j*m+m) (-1 = unused), moves cloud k by T6[(j*m+k)*6 ..] = {T00 T01 T02 T10 T11 T12} of
 * ref_pose.between(pose).matrix().astype(float32) like Keyframe.transform_points, concatenates them in that order and
 * runs pcl.downsample(resolution) (resolution <= 0: no downsample); the results become new slots (handles_out, host).
 * Targets of up to 65 536 points run on the resident filters, all jobs of the call in one launch each; a call with a
 * larger one (the loop-closure target of a long session, slam.py:999) takes its jobs one by one through a path without
 * a size limit.  A cloud whose count is < 0 is refused (SFE_ERR_ARG). */
int sfe_cloud_store_get_points(sfe_ctx *ctx, sfe_cloud_store *s, const int32_t *handles, const float *T6, int n_jobs, int m,
                               float resolution, int flags, const int64_t *stamps, int32_t *handles_out);
/* SLAM.compute_icp over handles (slam.py:294-323): pairs = n_jobs x (source handle, target handle); otherwise like
 * sfe_icp_jobs_dev (device guesses / results, enqueue only once the sizes are known) and sfe_icp_compute_jobs (host
 * guesses / results, one synchronisation).  Empty or failed clouds are refused (SFE_ERR_ARG): the caller tests
 * ssm_min_points first like slam.py:745. */
int sfe_icp_store_jobs_dev(sfe_ctx *ctx, const sfe_icp_params *p, sfe_cloud_store *s, const int32_t *pairs,
                           const float *d_guess9, int n_jobs, float *d_T9, int32_t *d_status, int32_t *d_iters);
int sfe_icp_store_compute(sfe_ctx *ctx, const sfe_icp_params *p, sfe_cloud_store *s, const int32_t *pairs,
                          const float *guesses9, int n_jobs, float *T_out9, int32_t *status, int32_t *iters);
/* SLAM.get_overlap (slam.py:389-424) for n_jobs (source, target) pairs: the source moved by T6[j*6 ..] (the estimated
 * pose's matrix as float32), pcl.match(target, source, 1, max_dist), counts_out[j] = matched points (host) */
int sfe_cloud_store_overlap(sfe_ctx *ctx, sfe_cloud_store *s, const int32_t *pairs, const float *T6, int n_jobs,
                            float max_dist, int flags, int32_t *counts_out);
/* ---- loop-closure search over the store: slam.py:839-1001 (initialize_nonsequential_scan_matching) ----
 * get_points(target_frames, None, return_keys=True) (slam.py:873 -> :229-292): m keyframe clouds, cloud k moved by
 * T6[k*6 ..] (its own pose's matrix as float32) and tagged with keys[k]; pcl.downsample(points, keys, resolution), the
 * descriptor overload (pcl.cpp:143-159): a leaf's medoid brings its key along.  No size limit (every keyframe older than
 * k - min_st_sep goes in).  One new slot; its keys stay on the device next to its points. */
int sfe_cloud_store_get_points_keys(sfe_ctx *ctx, sfe_cloud_store *s, const int32_t *handles, const float *T6,
                                    const int32_t *keys, int m, float resolution, int flags, int64_t stamp, int32_t *handle_out);
int sfe_cloud_store_read_keys(sfe_ctx *ctx, sfe_cloud_store *s, int32_t handle, int32_t *out, int cap, int *n_out);
/* The field-of-view gate (slam.py:875-899) on a keyed cloud: source frame f sees a point iff, moved by Tinv6[f*6 ..]
 * (pose.inverse().matrix() as float32, float32 arithmetic like numpy's), its range is < range_bound[f] and |bearing| <
 * bearing_bound[f]; a point is selected iff some frame sees it.  key_counts_out[k] (host, n_keys entries) = selected
 * points with key k -- np.unique(keys[sel], return_counts=True), slam.py:902; *n_selected_out their number.  The
 * selection stays on the device for sfe_cloud_store_compact_selected.  *n_ambiguous_out = points whose bearing lies
 * within float32 rounding of a bound (numpy's float32 arctan2 is not reproduced bit for bit): if it is not 0 the caller
 * evaluates the gate with numpy and hands the selection over with sfe_cloud_store_set_selection. */
int sfe_cloud_store_fov_select(sfe_ctx *ctx, sfe_cloud_store *s, int32_t handle, const float *Tinv6, const double *range_bound,
                               const double *bearing_bound, int n_frames, int n_keys, int32_t *key_counts_out,
                               int32_t *n_selected_out, int32_t *n_ambiguous_out);
int sfe_cloud_store_set_selection(sfe_ctx *ctx, sfe_cloud_store *s, int32_t handle, const uint8_t *sel, int n);
/* target_points[sel], target_keys[sel] (slam.py:898-899): a new slot, order kept, keys kept */
int sfe_cloud_store_compact_selected(sfe_ctx *ctx, sfe_cloud_store *s, int32_t handle, int64_t stamp, int32_t *handle_out);
/* get_overlap(source under T6, target, return_indices=True) + np.unique(target_keys[indices[indices != -1]],
 * return_counts=True) (slam.py:977-985): pcl.match's neighbour (nearest, lowest index among equals, within max_dist) of
 * every source point; key_counts_out[k] = matches whose target carries key k, *overlap_out their number */
int sfe_cloud_store_match_keys(sfe_ctx *ctx, sfe_cloud_store *s, int32_t source, const float *T6, int32_t target, float max_dist,
                               int flags, int n_keys, int32_t *key_counts_out, int32_t *overlap_out);
/* bounding boxes of n clouds: bbox_out[i*4 ..] = {min x, min y, max x, max y} (float32 min / max: what np.min / np.max
 * of the cloud give, slam.py:506-507) */
int sfe_cloud_store_bbox(sfe_ctx *ctx, sfe_cloud_store *s, const int32_t *handles, int n, float *bbox_out);

/* sfe_feature_extract_ping with the cloud left in the store instead of copied to the host: the
 * filtered cloud becomes a new slot (with SFE_STORE_NEGATE_Y in flags: as the SLAM node holds it), *handle_out its
 * handle, *n_out its size; cloud_out (nullable, host [cap x 2], (forward, lateral) as published) receives the points
 * only when the caller wants to publish them.  On SFE_ERR_CAP / *n_out = -1 no slot is kept (*handle_out = -1); a store
 * whose pool has no room for the cloud is SFE_ERR_CAP as well (the slot's own count is read back with the two sizes).
 * s may be NULL (then cloud_out is required and handle_out may be NULL): the same call without a store, for the flags
 * sfe_feature_extract_ping has no argument for (SFE_PING_VIS_JET). */
int sfe_feature_extract_ping_store(sfe_ctx *ctx, sfe_geom *g, sfe_cloud_store *s, int64_t stamp, const uint8_t *img, int alg,
                                   int train_hs, int guard_hs, int k, double tau, int intensity_thr, float resolution,
                                   double radius, int min_points, int64_t cap, int flags, int32_t *handle_out,
                                   int32_t *n_out, int32_t *n_raw_out, float *cloud_out, uint8_t *vis_out);

/* ---- global-initialisation matching cost: slam.py:461-570 ---------------- */
/*
 * get_matching_cost_subroutine1 builds a dilated occupancy grid of the target cloud and hands
 * scipy.optimize.shgo a function that counts the source points falling on set cells under a
 * candidate transform (slam.py:692-701, 952-961 call it 50-500+ times per keyframe, one pose per
 * call).  Here the grid lives on the device and MANY candidate transforms are scored per launch.
 *
 * sfe_costgrid_create: tgt_r / tgt_c = the target cells exactly as slam.py:515-518 computes them
 * (rounded, clipped; host int32), rows x cols = target_grids.shape, dilate_hs = slam.py:522.  The
 * grid is target_grids after cv2.dilate with getStructuringElement(MORPH_ELLIPSE, (2h+1, 2h+1), (h, h)).
 */
typedef struct sfe_costgrid sfe_costgrid;
int sfe_costgrid_create(sfe_ctx *ctx, const int32_t *tgt_r, const int32_t *tgt_c, int n_tgt, int rows,
                        int cols, int dilate_hs, sfe_costgrid **out);
/* The same for n device-resident target clouds at once (one grid each, e.g. one per session): grid i is built from
 * cloud target_handles[i] of the store with the cells of slam.py:514-517 computed on the device in the cloud's dtype
 * (float32: (p - min) / float32(resolution), half to even, clipped); xmin / ymin / rows / cols [n] are what the
 * caller's numpy made of the cloud's bounding box (slam.py:506-511; sfe_cloud_store_bbox brings the box).  Enqueue only. */
int sfe_costgrid_create_store(sfe_ctx *ctx, sfe_cloud_store *s, const int32_t *target_handles, int n, const float *xmin,
                              const float *ymin, float resolution, const int32_t *rows, const int32_t *cols, int dilate_hs,
                              sfe_costgrid **out);
void sfe_costgrid_destroy(sfe_costgrid *g);
/* dilated grid `index` as the reference holds it: rows x cols uint8, 0 / 255 (host buffer) */
int sfe_costgrid_download(sfe_ctx *ctx, sfe_costgrid *g, int index, uint8_t *grid_out);
/*
 * The body of `subroutine` (slam.py:531-564) for n_poses transforms.  src: n_src x 2 float32 source
 * points (host); T6: per pose the float32 entries T00 T01 T02 T10 T11 T12 of
 * sample_transform.matrix(); xmin, ymin (float32, slam.py:506), resolution (slam.py:508);
 * cost_out[p] = -(number of source points on a set cell).
 * flags & SFE_COST_F64_POINTS: the source cloud is a float64 numpy array of float32 values (the SLAM node's keyframe
 * clouds, slam_ros.py:169-170): moved point and cell in double, as numpy promotes them; otherwise a float32 cloud (what
 * get_points returns: sgemm arithmetic, cell in float32 with float32(resolution)).
 */
#define SFE_COST_F64_POINTS 1
int sfe_matching_cost_batch(sfe_ctx *ctx, sfe_costgrid *g, const float *src, int n_src, const float *T6,
                            int n_poses, float xmin, float ymin, double resolution, int flags, int32_t *cost_out);
/* ... over store handles, n_jobs (source cloud, grid) pairs in one launch: job i scores source cloud source_handles[i]
 * against grid grid_index[i] of `g` (NULL: grid i, n_jobs = the number of grids) under the transforms
 * T6[(i * n_poses + p) * 6 ..]; cost_out[i * n_poses + p] (host).  One synchronisation. */
int sfe_matching_cost_store(sfe_ctx *ctx, sfe_costgrid *g, sfe_cloud_store *s, const int32_t *source_handles,
                            const int32_t *grid_index, int n_jobs, const float *T6, int n_poses, double resolution, int flags,
                            int32_t *cost_out);

/* ... the sample transforms computed on the device: job i scores source_handles[i] against its grid under
 * target_i.between(source_i.compose(delta_j)).matrix() (slam.py:548-550) for n_deltas deltas shared by all jobs; poses as
 * {x, y, cos(theta), sin(theta)} doubles (host): target_xycs / source_xycs [n_jobs x 4], delta_xycs [n_deltas x 4];
 * cost_out[i * n_deltas + j].  gtsam.Pose2's double arithmetic as sfe_pose2_sample_transforms below, bit for bit. */
int sfe_matching_cost_store_samples(sfe_ctx *ctx, sfe_costgrid *g, sfe_cloud_store *s, const int32_t *source_handles,
                                    const int32_t *grid_index, int n_jobs, const double *target_xycs, const double *source_xycs,
                                    const double *delta_xycs, int n_deltas, double resolution, int flags, int32_t *cost_out);

/* ---- replaces: everything scipy.optimize.shgo(func, bounds, n, iters=1, sampling_method="sobol", minimizer_kwargs=
 * {"options": {"ftol": ...}}) of slam.py:692-701 does AFTER its sampling stage, for n_problems cost tables at once (host only,
 * no device work).  Holds for the piecewise-constant cost of slam.py:529-567 only.  The vertex graph (CSR nn_off / nn_idx over
 * n_vertices vertices in shgo's vertex-cache order), the vertices x [n_vertices x 3] and the three forward-difference points of
 * SLSQP per vertex come from one run of the installed scipy (sonar_slam_amd/shgo_fast.py, SobolPlan);
 * tables [n_problems x n_vertices x 4]: the cost at each vertex and at its three finite-difference points.
 * status_out: SFE_SHGO_OK (result = vertex_out), SFE_SHGO_OK_TIED (several local results share the lowest cost: the caller asks
 * np.argsort over the costs of order_out[.. n_order_out], as shgo does -- its sort kernel is not stable), SFE_SHGO_FAILED (no
 * vertex strictly below all its neighbours: shgo's success = False, vertex_out = the lowest vertex), SFE_SHGO_FALLBACK (a
 * finite-difference point costs something else than its vertex or more than SFE_SHGO_MAX_POOL minimisers: run
 * scipy.optimize.shgo itself; two pool members equally far from the last result: the order is numpy's sort kernel's choice,
 * shgo_fast.py asks it).  order_out [n_problems x SFE_SHGO_MAX_POOL]: the
 * vertices in the order shgo minimises them. */
#define SFE_SHGO_OK 0
#define SFE_SHGO_FAILED 1
#define SFE_SHGO_FALLBACK 2
#define SFE_SHGO_OK_TIED 3
#define SFE_SHGO_MAX_POOL 32
int sfe_shgo_sobol_replay(int n_vertices, const int32_t *nn_off, const int32_t *nn_idx, const double *x, const int32_t *tables,
                          int n_problems, uint8_t *status_out, int32_t *vertex_out, int32_t *n_order_out, int32_t *order_out);
/* T6_out[(i * n_deltas + j) * 6 ..] = float32 rows of target_i.between(source_i.compose(delta_j)).matrix() (slam.py:548-550:
 * the sample transform of the matching cost) for n_sessions pose pairs x n_deltas deltas; every pose as {x, y, cos(theta),
 * sin(theta)} doubles.  gtsam.Pose2's expressions as sonar_slam_amd/pose2.py states them, in double, in the same order.  Host only. */
int sfe_pose2_sample_transforms(const double *target_xycs, const double *source_xycs, int n_sessions, const double *delta_xycs,
                                int n_deltas, float *T6_out);

#ifdef __cplusplus
}
#endif
#endif /* SONARFE_H */
