// Internal definitions shared by the libsonarfe translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/sonarfe.h"

#define SFE_NSCRATCH 64
#define SFE_ICP_PROF_N 96 // values sfe_icp_get_profile hands back

struct sfe_ctx {
    int device = -1;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // side stream of the ICP launcher (icp_variant bit 3): the prep kernel of a batch whose inputs are final runs
    // here, next to whatever precedes the ICP call on `stream`; ev_prep orders the loop kernel behind it, ev_loop
    // keeps the next batch's prep off the scratch the loop kernel still reads
    hipStream_t stream2 = nullptr;
    hipEvent_t ev_prep = nullptr, ev_loop = nullptr;
    bool icp_loop_pending = false;
    // copy stream of the streamed-input path (sfe_memcpy_h2d_async): uploads from pinned host memory run next to the
    // kernels on `stream`; ev_copy = behind the last upload, ev_compute = where the kernels stood when the caller last
    // said "everything enqueued so far has consumed its input" (sfe_stream_fence)
    hipStream_t stream_copy = nullptr;
    hipEvent_t ev_copy = nullptr, ev_compute = nullptr;
    std::string err;
    struct Buf {
        void *p = nullptr;
        size_t cap = 0;
    } scratch[SFE_NSCRATCH];
    // pinned host staging for the small tables a launch hands to the device (job records, offsets): two blocks
    // used alternately, each guarded by the event recorded behind its last copy, so that enqueue-only (_dev)
    // entry points never synchronise a stream to keep a pageable vector alive
    struct Pin {
        void *p = nullptr;
        size_t cap = 0;
        hipEvent_t ev = nullptr;
        bool pending = false;
    } pin[2];
    int pin_next = 0;
    void *bm_clean_ptr = nullptr; // extract_dev: the canvas bitmap scratch is known to be zero up to bm_clean_bytes
    size_t bm_clean_bytes = 0;
    Pin pin_io[4]; // grow-only pinned buffers of the synchronous single-item entry points (no events: the call syncs)
    // float threshold table of the sliding-sum CFAR kernel currently on the device (scratch slot 37)
    int thr_tab_alg = -1, thr_tab_T = -1;
    double thr_tab_tau = 0.0;
    const void *thr_tab_ptr = nullptr;
    int tha_alg = -1, tha_T = -1, tha_on = 0; // cfar_thr_arith checked against the reference expression for these
    double tha_tau = 0.0;
    int cfar_tile_rows = 0;
    int cfar_variant = 0;
    int icp_variant = 0;
    int extract_variant = 0;     // 0 = inverse map for binary masks, records instead of a canvas for bit-stream batches (default); 1 = dense pass only; 2 = inverse map through the canvas bitmap always (A/B)
    int icp_prof = 0;            // debug: per-phase cycle counts of workgroup 0 of the sweep kernel
    long long icp_prof_host[SFE_ICP_PROF_N] = {0};
    int n_cu = 256;
    // clouds left in the staging slots by sfe_extract_points_bits_staged_dev, waiting for sfe_cloud_filter_staged_dev
    // (-1: none; anything else that writes those slots resets it)
    int staged_frames = -1;
    long long staged_cap = 0;
};

struct sfe_geom {
    sfe_ctx *ctx = nullptr;
    int cart_rows = 0, cart_cols = 0, polar_rows = 0, polar_cols = 0;
    double width = 0, height = 0;
    int32_t *d_code = nullptr;   // per Cartesian pixel: packed (iy, ix, table index) or -1
    int32_t *d_span = nullptr;   // per Cartesian row: [first, last+1) columns with code != -1
    int words_per_row = 0;       // 64-bit bitmap words per Cartesian row
    unsigned rcp = 0;            // ceil(2^32 / (polar_cols+1))
    int32_t *d_tile_rows = nullptr; // per canvas tile: [ylo, yhi] polar rows tapped by its valid pixels
    int word_groups = 0, tiles_per_frame = 0, lds_bytes = 0;
    // inverse map for sparse binary masks: for every polar pixel the canvas pixels that tap it with a
    // non-zero weight (CSR: offsets [polar_rows * polar_cols + 1], entries = linear canvas indices)
    int32_t *d_inv_off = nullptr;
    uint2 *d_inv_lut = nullptr;     // {canvas index, decision table of the entry} (extract_gather_kernel)
    uint2 *d_inv_ob = nullptr;      // compact form (round 4): per polar pixel {offset into d_inv_c4, base bit index}
    uint32_t *d_inv_c4 = nullptr;   //   4-byte entries {table, tap place, dx, dy}, dead entries dropped
    // px -> m of feature_extraction.py:236-237 per canvas row / column (fp64, the reference's operation order,
    // evaluated once on the host: extract_expand_words_kernel looks the metres up instead of dividing per point)
    double *d_ytab = nullptr, *d_xtab = nullptr;
};

int sfe_set_err(sfe_ctx *ctx, int code, const char *fmt, ...);
int sfe_mask_pack(sfe_ctx *ctx, const uint8_t *d_mask, int n_frames, long long px, uint32_t *d_bits,
                  int32_t *d_nonbin); // sfe_remap.hip: byte mask -> bit stream
void *sfe_scratch(sfe_ctx *ctx, int slot, size_t bytes);  // grow-only device scratch; nullptr on failure
// Pinned staging: sfe_pinned_begin hands out a host block of >= bytes (waiting, if need be, for the copy that last
// read it -- two launches ago); the caller fills it, enqueues its hipMemcpyAsync calls on `s` and then calls
// sfe_pinned_end(ctx, s) so the block is not reused before those copies have run.  nullptr on failure.
void *sfe_pinned_begin(sfe_ctx *ctx, size_t bytes);
int sfe_pinned_end(sfe_ctx *ctx, hipStream_t s);
// grow-only pinned buffer `slot` (0..3) of >= bytes for entry points that end with a stream synchronisation
void *sfe_pinned_io(sfe_ctx *ctx, int slot, size_t bytes);

#define SFE_HIP(ctx, call)                                                                       \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return sfe_set_err((ctx), SFE_ERR_HIP, "%s failed: %s (%s:%d)", #call,               \
                               hipGetErrorString(e_), __FILE__, __LINE__);                       \
    } while (0)

#define SFE_LAUNCH_CHECK(ctx) SFE_HIP(ctx, hipGetLastError())

#define SFE_ARG(ctx, cond)                                                                       \
    do {                                                                                         \
        if (!(cond))                                                                             \
            return sfe_set_err((ctx), SFE_ERR_ARG, "bad argument: %s (%s:%d)", #cond, __FILE__,  \
                               __LINE__);                                                        \
    } while (0)

// sfe_icp_sweep.hip: 0 = launched, 1 = a target exceeds the LDS capacity (use the brute-force kernel), < 0 = error
int sfe_icp_sweep_launch(sfe_ctx *ctx, const sfe_icp_params *p, const float *d_src, const float *d_tgt,
                         const int32_t *jobs4, const float *d_guess9, int n_jobs, float *d_T9, int32_t *d_status,
                         int32_t *d_iters);

// sfe_store.hip: append n_frames clouds ([f][cap] float2 + counts[f], device) to a store; enqueue only
struct sfe_cloud_store;
int sfe_store_append_dev(sfe_cloud_store *s, const int64_t *stamps, const float *d_clouds, const int32_t *d_counts,
                         int n_frames, int64_t cap, int flags, int32_t *handles_out);
// ... the count its commit wrote for a slot (< 0: SFE_STORE_*), copied in stream order to pinned memory; the store's context
int sfe_store_slot_count_async(sfe_cloud_store *s, int32_t handle, int32_t *h_pinned);
sfe_ctx *sfe_store_ctx(sfe_cloud_store *s);

// what a store is made of, for the other translation units that read clouds by handle (sfe_cost.hip); syncs the host
// mirror of the slot table first
struct SfeStoreView {
    const void *d_pool;     // float2 points
    const int64_t *d_off;   // device slot table
    const int32_t *d_cnt;
    const int64_t *off;     // host mirror (valid for slots < n_slots)
    const int32_t *cnt;
    int n_slots;
};
int sfe_store_view(sfe_cloud_store *s, SfeStoreView *v);

// sfe_downsample.hip: pcl.downsample with indices on a device-resident cloud of any size (rank sort in global memory)
struct SfeDsHeader {
    float cx, cy, radius;
    int levels;
    int n_seg;
};
int sfe_ds_run_dev(sfe_ctx *ctx, const float *d_pts, int n, float resolution, float *d_out, int32_t *d_out_idx,
                   SfeDsHeader *d_hdr);

static inline int sfe_use(sfe_ctx *ctx)
{
    if (!ctx)
        return SFE_ERR_ARG;
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess)
        return sfe_set_err(ctx, SFE_ERR_HIP, "hipSetDevice(%d): %s", ctx->device, hipGetErrorString(e));
    return 0;
}
