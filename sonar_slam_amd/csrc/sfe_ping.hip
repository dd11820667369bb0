// One sonar ping through the whole of FeatureExtraction.callback in ONE call:
//   bruce_slam/src/bruce_slam/feature_extraction.py:223-224  CFAR.detect + intensity gate
//   :226                                                      cv2.remap(img) for the visualisation (optional)
//   :231-238                                                  cv2.remap(peaks) -> np.nonzero -> pixel -> metres
//   :241-249                                                  pcl.downsample, pcl.remove_outlier
// The live ROS node handles one ping per callback, so what counts is latency, not throughput: the per-stage
// host entry points (sfe_cfar_u8, sfe_extract_points, sfe_downsample, sfe_remove_outlier) each pay a pageable
// host->device copy, a device->host copy and a stream synchronisation.  Here the ping goes up once through pinned
// staging, the stages run back to back on the context's stream (the same kernels as the batched resident path, batch
// of one), and the final float32 cloud comes down once: one synchronisation per ping.
#include "sfe_internal.h"

#include <algorithm>
#include <cstring>

// grow-only pinned host buffers for the synchronous single-item entry points (their call ends with a stream
// synchronisation, so a buffer is free again when the call returns)
static void *pinned_io(sfe_ctx *ctx, int slot, size_t bytes)
{
    auto &b = ctx->pin_io[slot];
    if (bytes <= b.cap && b.p)
        return b.p;
    if (b.p)
        (void)hipHostFree(b.p);
    b.p = nullptr;
    b.cap = 0;
    const size_t want = bytes + bytes / 4 + 4096;
    if (hipHostMalloc(&b.p, want, hipHostMallocDefault) != hipSuccess) {
        b.p = nullptr;
        sfe_set_err(ctx, SFE_ERR_HIP, "hipHostMalloc(%zu) for pinned staging failed", want);
        return nullptr;
    }
    b.cap = want;
    return b.p;
}

void *sfe_pinned_io(sfe_ctx *ctx, int slot, size_t bytes) { return pinned_io(ctx, slot, bytes); }

// store != nullptr: the filtered cloud is appended to the store (device to device) and only the two counts come down,
// plus the points when cloud_out is given
static int ping_impl(sfe_ctx *ctx, sfe_geom *g, sfe_cloud_store *store, int64_t stamp, int store_flags, const uint8_t *img,
                     int alg, int train_hs, int guard_hs, int k, double tau, int intensity_thr, float resolution,
                     double radius, int min_points, int64_t cap, float *cloud_out, int32_t *n_out, int32_t *n_raw_out,
                     uint8_t *vis_out, int32_t *handle_out)
{
    const size_t np = (size_t)g->polar_rows * g->polar_cols, nc = (size_t)g->cart_rows * g->cart_cols;
    uint8_t *d_img = (uint8_t *)sfe_scratch(ctx, 32, np);
    // detections: a bit stream when the rows are whole words (SFE_BITS_WORDS(np) words fit into np bytes for
    // np >= 8), the 0/1 byte mask otherwise
    const bool bits = (g->polar_cols & 31) == 0 && np >= 64;
    uint8_t *d_mask = (uint8_t *)sfe_scratch(ctx, 33, np);
    double *d_pts = (double *)sfe_scratch(ctx, 34, (size_t)cap * 16);
    // [0] raw point count, [1] filtered count, then the filtered float32 cloud: one block, one copy back
    int32_t *d_res = (int32_t *)sfe_scratch(ctx, 35, 16 + (size_t)cap * 8);
    const bool jet = (store_flags & SFE_PING_VIS_JET) != 0;
    const size_t nvis = jet ? 3 * nc : nc;
    uint8_t *d_vis = vis_out ? (uint8_t *)sfe_scratch(ctx, 36, nvis + 4) : nullptr;
    uint8_t *h_img = (uint8_t *)pinned_io(ctx, 0, np);
    char *h_res = (char *)pinned_io(ctx, 1, 16 + (size_t)cap * 8 + (vis_out ? nvis : 0));
    if (!d_img || !d_mask || !d_pts || !d_res || (vis_out && !d_vis) || !h_img || !h_res)
        return SFE_ERR_HIP;
    memcpy(h_img, img, np);
    SFE_HIP(ctx, hipMemcpyAsync(d_img, h_img, np, hipMemcpyHostToDevice, ctx->stream));
    if (int rc = bits ? sfe_cfar_u8_bits_batch_dev(ctx, d_img, 1, g->polar_rows, g->polar_cols, alg, train_hs, guard_hs,
                                                   k, tau, intensity_thr, reinterpret_cast<uint32_t *>(d_mask))
                      : sfe_cfar_u8_batch_dev(ctx, d_img, 1, g->polar_rows, g->polar_cols, alg, train_hs, guard_hs, k,
                                              tau, intensity_thr, d_mask, nullptr))
        return rc;
    if (vis_out)
        if (int rc = jet ? sfe_remap_u8_colormap_dev(ctx, g, d_img, SFE_COLORMAP_JET, d_vis) : sfe_remap_u8_dev(ctx, g, d_img, d_vis))
            return rc;
    if (int rc = bits ? sfe_extract_points_bits_batch_dev(ctx, g, reinterpret_cast<const uint32_t *>(d_mask), 1, cap,
                                                          d_pts, d_res)
                      : sfe_extract_points_batch_dev(ctx, g, d_mask, 1, cap, d_pts, d_res))
        return rc;
    float *d_cloud = reinterpret_cast<float *>(d_res + 4);
    if (int rc = sfe_cloud_filter_batch_dev(ctx, d_pts, d_res, 1, cap, resolution, radius, min_points, d_cloud, d_res + 1))
        return rc;
    int32_t handle = -1;
    if (store)
        if (int rc = sfe_store_append_dev(store, &stamp, d_cloud, d_res + 1, 1, cap, store_flags, &handle))
            return rc;
    // the points come down only when somebody wants them on the host
    const size_t b_down = 16 + ((!store || cloud_out) ? (size_t)cap * 8 : 0);
    SFE_HIP(ctx, hipMemcpyAsync(h_res, d_res, b_down, hipMemcpyDeviceToHost, ctx->stream));
    // ... and what the store made of the cloud: the slot's own count (-3: the pool is full), behind the block above in
    // stream order, into the unused third word of its header
    if (store)
        if (int rc = sfe_store_slot_count_async(store, handle, reinterpret_cast<int32_t *>(h_res) + 2))
            return rc;
    if (vis_out)
        SFE_HIP(ctx, hipMemcpyAsync(h_res + 16 + (size_t)cap * 8, d_vis, nvis, hipMemcpyDeviceToHost, ctx->stream));
    SFE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const int32_t n_raw = reinterpret_cast<int32_t *>(h_res)[0], n = reinterpret_cast<int32_t *>(h_res)[1];
    const int32_t n_slot = store ? reinterpret_cast<int32_t *>(h_res)[2] : n;
    if (n_raw_out)
        *n_raw_out = n_raw;
    if (vis_out)
        memcpy(vis_out, h_res + 16 + (size_t)cap * 8, nvis);
    if (handle_out)
        *handle_out = handle;
    if (n_raw > cap || n < 0 || n_slot < 0) {
        if (store) { // no slot for a cloud that is truncated, was refused or did not fit the pool
            if (int rc = sfe_cloud_store_truncate(ctx, store, handle))
                return rc;
            if (handle_out)
                *handle_out = -1;
        }
        if (n_raw > cap) {
            *n_out = 0;
            return sfe_set_err(ctx, SFE_ERR_CAP, "feature_extract_ping: %d points exceed capacity %lld", n_raw, (long long)cap);
        }
        if (n >= 0) { // the filters delivered, the store had no room: a full pool is an error, not an empty cloud
            *n_out = 0;
            return sfe_set_err(ctx, SFE_ERR_CAP, "feature_extract_ping: the cloud store is full (%d points did not fit its pool)", n);
        }
        *n_out = -1; // octree deeper than 24 levels (sfe_cloud_filter_batch_dev): the caller takes the per-cloud path
        return 0;
    }
    *n_out = n;
    if (cloud_out)
        memcpy(cloud_out, h_res + 16, (size_t)n * 8);
    return 0;
}

extern "C" int sfe_feature_extract_ping(sfe_ctx *ctx, sfe_geom *g, const uint8_t *img, int alg, int train_hs,
                                        int guard_hs, int k, double tau, int intensity_thr, float resolution,
                                        double radius, int min_points, int64_t cap, float *cloud_out, int32_t *n_out,
                                        int32_t *n_raw_out, uint8_t *vis_out)
{
    if (int rc = sfe_use(ctx))
        return rc;
    SFE_ARG(ctx, g && img && cloud_out && n_out && g->ctx == ctx && cap > 0 && cap <= 65536);
    return ping_impl(ctx, g, nullptr, 0, 0, img, alg, train_hs, guard_hs, k, tau, intensity_thr, resolution, radius,
                     min_points, cap, cloud_out, n_out, n_raw_out, vis_out, nullptr);
}

extern "C" int sfe_feature_extract_ping_store(sfe_ctx *ctx, sfe_geom *g, sfe_cloud_store *s, int64_t stamp,
                                              const uint8_t *img, int alg, int train_hs, int guard_hs, int k, double tau,
                                              int intensity_thr, float resolution, double radius, int min_points,
                                              int64_t cap, int flags, int32_t *handle_out, int32_t *n_out,
                                              int32_t *n_raw_out, float *cloud_out, uint8_t *vis_out)
{
    if (int rc = sfe_use(ctx))
        return rc;
    SFE_ARG(ctx, g && img && n_out && g->ctx == ctx && cap > 0 && cap <= 65536 && (s ? handle_out != nullptr : cloud_out != nullptr));
    SFE_ARG(ctx, !s || sfe_store_ctx(s) == ctx); // the append runs on the store's stream: it must be the filters' stream
    return ping_impl(ctx, g, s, stamp, flags, img, alg, train_hs, guard_hs, k, tau, intensity_thr, resolution, radius,
                     min_points, cap, cloud_out, n_out, n_raw_out, vis_out, handle_out);
}
