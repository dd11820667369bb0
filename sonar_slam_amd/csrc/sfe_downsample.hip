// pcl.downsample on gfx950.  Replaces bruce_slam/src/bruce_slam/cpp/pcl.cpp:128-159
// (libpointmatcher OctreeGridDataPointsFilter{maxSizeByNode = resolution, samplingMethod =
// MEDOID, maxPointByNode = 1}), called on every ping (feature_extraction.py:241-242) and on every
// aggregated ICP cloud (slam.py:288-292).
//
// The library builds a quadtree by recursive stable 4-way splits and visits the leaves depth
// first (restated in oracle/sonar_oracle.c, PARITY UNPINNED).  The same result without a tree:
//   * a node stops splitting when radius*2 <= maxSize or it holds <= 1 point, so every leaf
//     with more than one point sits at the fixed depth L where the cell size reaches maxSize;
//   * descending exactly L levels for EVERY point (same float centre updates c +- r/2 as the
//     library) gives a 2L-bit path key; leaves visited depth-first == keys in ascending order,
//     points of a leaf in original order == stable sort by (key, index);
//   * each run of equal keys is one leaf: float centroid in that order, medoid = first point at
//     minimum float distance.
// Sorting is brute-force rank counting (every point against every point, keys tiled through
// LDS): N is a few thousand, the pass is a few microseconds, and it is trivially exact/stable.
#include "sfe_internal.h"

#include <cstdio>
#include <cstdlib>

typedef SfeDsHeader DsHeader; // {cx, cy, radius, levels, n_seg}: sfe_internal.h

#define DS_MAX_LEVELS 31

__global__ __launch_bounds__(1024) void ds_bbox_kernel(const float2 *__restrict__ pts, int n, float max_size,
                                                       DsHeader *__restrict__ hdr)
{
    __shared__ float s_mn[2][16], s_mx[2][16];
    float mnx = INFINITY, mny = INFINITY, mxx = -INFINITY, mxy = -INFINITY;
    for (int i = threadIdx.x; i < n; i += 1024) {
        const float2 p = pts[i];
        mnx = fminf(mnx, p.x);
        mxx = fmaxf(mxx, p.x);
        mny = fminf(mny, p.y);
        mxy = fmaxf(mxy, p.y);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        mnx = fminf(mnx, __shfl_down(mnx, d));
        mxx = fmaxf(mxx, __shfl_down(mxx, d));
        mny = fminf(mny, __shfl_down(mny, d));
        mxy = fmaxf(mxy, __shfl_down(mxy, d));
    }
    if ((threadIdx.x & 63) == 0) {
        s_mn[0][threadIdx.x >> 6] = mnx;
        s_mn[1][threadIdx.x >> 6] = mny;
        s_mx[0][threadIdx.x >> 6] = mxx;
        s_mx[1][threadIdx.x >> 6] = mxy;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w) {
            mnx = fminf(mnx, s_mn[0][w]);
            mny = fminf(mny, s_mn[1][w]);
            mxx = fmaxf(mxx, s_mx[0][w]);
            mxy = fmaxf(mxy, s_mx[1][w]);
        }
        // Octree::build: centre = min + radii*0.5, radius = max(radii)*0.5
        const float rx = mxx - mnx, ry = mxy - mny;
        hdr->cx = mnx + rx * 0.5f;
        hdr->cy = mny + ry * 0.5f;
        float radius = rx;
        if (radius < ry)
            radius = ry;
        radius *= 0.5f;
        hdr->radius = radius;
        int L = 0;
        float r = radius;
        while (!((double)r * 2.0 <= (double)max_size) && L < DS_MAX_LEVELS) {
            r *= 0.5f;
            ++L;
        }
        hdr->levels = L;
        hdr->n_seg = 0;
    }
}

__global__ __launch_bounds__(256) void ds_key_kernel(const float2 *__restrict__ pts, int n,
                                                     const DsHeader *__restrict__ hdr,
                                                     unsigned long long *__restrict__ keys)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n)
        return;
    const float2 p = pts[i];
    float cx = hdr->cx, cy = hdr->cy, r = hdr->radius;
    const int L = hdr->levels;
    unsigned long long key = 0;
    for (int l = 0; l < L; ++l) {
        const unsigned bx = p.x > cx, by = p.y > cy; // Octree::idx: bit i = pt(i) > centre(i)
        key = (key << 2) | (bx | (by << 1));
        const float hr = r * 0.5f;
        cx = cx + (bx ? hr : -hr);
        cy = cy + (by ? hr : -hr);
        r = hr;
    }
    keys[i] = key;
}

// rank of (key_i, i) among all points = position in the stable sort; scatter index and key there
__global__ __launch_bounds__(256) void ds_rank_kernel(const unsigned long long *__restrict__ keys, int n,
                                                      int *__restrict__ sorted_idx,
                                                      unsigned long long *__restrict__ sorted_key)
{
    __shared__ unsigned long long s_k[2048];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const unsigned long long ki = (i < n) ? keys[i] : 0ull;
    int rank = 0;
    for (int tb = 0; tb < n; tb += 2048) {
        const int tn = min(2048, n - tb);
        __syncthreads();
        for (int j = threadIdx.x; j < tn; j += 256)
            s_k[j] = keys[tb + j];
        __syncthreads();
        for (int j = 0; j < tn; ++j) {
            const unsigned long long kj = s_k[j];
            rank += (kj < ki) || (kj == ki && tb + j < i);
        }
    }
    if (i < n) {
        sorted_idx[rank] = i;
        sorted_key[rank] = ki;
    }
}

// one workgroup: mark the first position of every run of equal keys, list the run starts
__global__ __launch_bounds__(1024) void ds_segment_kernel(const unsigned long long *__restrict__ sorted_key, int n,
                                                          int *__restrict__ seg_start, DsHeader *__restrict__ hdr)
{
    __shared__ int s_part[1024];
    const int per = (n + 1023) / 1024;
    const int b = threadIdx.x * per, e = min(b + per, n);
    int c = 0;
    for (int r = b; r < e; ++r)
        c += (r == 0) || (sorted_key[r] != sorted_key[r - 1]);
    s_part[threadIdx.x] = c;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int v = (threadIdx.x >= d) ? s_part[threadIdx.x - d] : 0;
        __syncthreads();
        s_part[threadIdx.x] += v;
        __syncthreads();
    }
    int s = (threadIdx.x == 0) ? 0 : s_part[threadIdx.x - 1];
    for (int r = b; r < e; ++r)
        if ((r == 0) || (sorted_key[r] != sorted_key[r - 1]))
            seg_start[s++] = r;
    if (threadIdx.x == 1023) {
        hdr->n_seg = s_part[1023];
        seg_start[s_part[1023]] = n; // sentinel
    }
}

// one thread per leaf: float centroid in original order, first point at minimum distance
__global__ __launch_bounds__(256) void ds_medoid_kernel(const float2 *__restrict__ pts,
                                                        const int *__restrict__ sorted_idx,
                                                        const int *__restrict__ seg_start,
                                                        const DsHeader *__restrict__ hdr, float2 *__restrict__ out,
                                                        int *__restrict__ out_idx)
{
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= hdr->n_seg)
        return;
    const int r0 = seg_start[s], r1 = seg_start[s + 1];
    float sx = 0.0f, sy = 0.0f;
    for (int r = r0; r < r1; ++r) {
        const float2 p = pts[sorted_idx[r]];
        sx = __fadd_rn(sx, p.x);
        sy = __fadd_rn(sy, p.y);
    }
    const float cnt = (float)(r1 - r0);
    sx = __fdiv_rn(sx, cnt);
    sy = __fdiv_rn(sy, cnt);
    float best = 3.402823466e+38f;
    int bi = sorted_idx[r0];
    for (int r = r0; r < r1; ++r) {
        const int id = sorted_idx[r];
        const float2 p = pts[id];
        const float dx = __fadd_rn(p.x, -sx), dy = __fadd_rn(p.y, -sy);
        // sqrtf, not __fsqrt_rn: the intrinsic lowers to a bare v_sqrt_f32 (1 ulp), sqrtf to the correctly rounded
        // sequence; a 1-ulp tie between two points of a leaf must fall like it does on the CPU
        const float d = sqrtf(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)));
        if (d < best) {
            best = d;
            bi = id;
        }
    }
    out[s] = pts[bi];
    out_idx[s] = bi;
}

// The chain above on a cloud that is already on the device (sfe_store.hip: targets beyond the resident filter's capacity,
// and the descriptor overload of pcl.downsample, which needs the indices).  Enqueue only: d_out / d_out_idx hold the
// medoids and their indices into d_pts, d_hdr->n_seg their number.  Scratch slots 1-4.
int sfe_ds_run_dev(sfe_ctx *ctx, const float *d_pts_, int n, float resolution, float *d_out_, int32_t *d_out_idx,
                   SfeDsHeader *d_hdr_)
{
    const float2 *d_pts = (const float2 *)d_pts_;
    float2 *d_out = (float2 *)d_out_;
    DsHeader *d_hdr = (DsHeader *)d_hdr_;
    char buf[64];
    snprintf(buf, sizeof buf, "%f", (double)resolution); // pcl.cpp:134: std::to_string(float)
    const float max_size = strtof(buf, nullptr);
    unsigned long long *d_keys = (unsigned long long *)sfe_scratch(ctx, 1, 8 * (size_t)n);
    unsigned long long *d_skeys = (unsigned long long *)sfe_scratch(ctx, 2, 8 * (size_t)n);
    int *d_sidx = (int *)sfe_scratch(ctx, 3, 4 * (size_t)n);
    int *d_seg = (int *)sfe_scratch(ctx, 4, 4 * ((size_t)n + 1));
    if (!d_keys || !d_skeys || !d_sidx || !d_seg)
        return SFE_ERR_HIP;
    const unsigned nb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(ds_bbox_kernel, dim3(1), dim3(1024), 0, ctx->stream, d_pts, n, max_size, d_hdr);
    hipLaunchKernelGGL(ds_key_kernel, dim3(nb), dim3(256), 0, ctx->stream, d_pts, n, d_hdr, d_keys);
    hipLaunchKernelGGL(ds_rank_kernel, dim3(nb), dim3(256), 0, ctx->stream, d_keys, n, d_sidx, d_skeys);
    hipLaunchKernelGGL(ds_segment_kernel, dim3(1), dim3(1024), 0, ctx->stream, d_skeys, n, d_seg, d_hdr);
    hipLaunchKernelGGL(ds_medoid_kernel, dim3(nb), dim3(256), 0, ctx->stream, d_pts, d_sidx, d_seg, d_hdr, d_out,
                       d_out_idx);
    SFE_LAUNCH_CHECK(ctx);
    return 0;
}

extern "C" {

int sfe_downsample(sfe_ctx *ctx, const float *pts, int n, float resolution, float *out, int32_t *out_idx,
                   int *n_out)
{
    if (int rc = sfe_use(ctx))
        return rc;
    SFE_ARG(ctx, n >= 0 && n_out && (n == 0 || (pts && out)));
    *n_out = 0;
    if (n == 0)
        return 0; // pcl.cpp:130-131
    // Without the indices (pcl.downsample(points, resolution), the call of every ping and every get_points; only the
    // descriptor overload needs them) the cloud goes through the resident batch path as a batch of one: radix / bitonic
    // sort in LDS instead of the rank counting below (every point against every point: 0.65 ms for an 11 000-point
    // cloud against ~0.1 ms).  Same octree restatement, same medoids (both are checked against the oracle).  The
    // float32 points are widened to float64 on the way up, which sfe_cloud_filter_batch_dev casts back unchanged.
    static const bool no_fast = getenv("SFE_DS_RANK") != nullptr; // A/B
    // (resolution <= 0 / NaN means "no downsample" to the batch entry point -- feature_extraction.py:241 -- but to
    // pcl.downsample it is maxSizeByNode = 0: split down to single points.  Those calls keep the rank-counting path.)
    if (!out_idx && !no_fast && n <= 65536 && resolution > 0.0f) {
        const size_t b_in = sizeof(double) * 2 * (size_t)n + 16;
        double *h64 = (double *)sfe_pinned_io(ctx, 0, b_in);
        char *d_in = (char *)sfe_scratch(ctx, 13, b_in);
        float *d_o = (float *)sfe_scratch(ctx, 18, sizeof(float) * 2 * (size_t)n + 16);
        if (!h64 || !d_in || !d_o)
            return SFE_ERR_HIP;
        for (int i = 0; i < 2 * n; ++i)
            h64[i] = (double)pts[i];
        int32_t *h_cnt = (int32_t *)(h64 + 2 * (size_t)n);
        h_cnt[0] = n;
        SFE_HIP(ctx, hipMemcpyAsync(d_in, h64, b_in, hipMemcpyHostToDevice, ctx->stream));
        int32_t *d_cnt_in = (int32_t *)(d_in + sizeof(double) * 2 * (size_t)n);
        int32_t *d_cnt_out = (int32_t *)(d_o + 2 * (size_t)n);
        if (int rc = sfe_cloud_filter_batch_dev(ctx, (const double *)d_in, d_cnt_in, 1, n, resolution, 0.0, 0, d_o, d_cnt_out))
            return rc;
        int32_t m = 0;
        SFE_HIP(ctx, hipMemcpyAsync(&m, d_cnt_out, sizeof m, hipMemcpyDeviceToHost, ctx->stream));
        SFE_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (m >= 0) {
            SFE_HIP(ctx, hipMemcpyAsync(out, d_o, sizeof(float) * 2 * (size_t)m, hipMemcpyDeviceToHost, ctx->stream));
            SFE_HIP(ctx, hipStreamSynchronize(ctx->stream));
            *n_out = m;
            return 0;
        }
        // (-1: an octree deeper than 24 levels -- the rank counting below handles any depth)
    }
    float2 *d_pts = (float2 *)sfe_scratch(ctx, 0, sizeof(float2) * (size_t)n);
    float2 *d_out = (float2 *)sfe_scratch(ctx, 5, sizeof(float2) * (size_t)n);
    int *d_oidx = (int *)sfe_scratch(ctx, 6, 4 * (size_t)n);
    DsHeader *d_hdr = (DsHeader *)sfe_scratch(ctx, 7, sizeof(DsHeader));
    if (!d_pts || !d_out || !d_oidx || !d_hdr)
        return SFE_ERR_HIP;
    SFE_HIP(ctx, hipMemcpyAsync(d_pts, pts, sizeof(float2) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    if (int rc = sfe_ds_run_dev(ctx, (const float *)d_pts, n, resolution, (float *)d_out, d_oidx, (SfeDsHeader *)d_hdr))
        return rc;
    DsHeader h;
    SFE_HIP(ctx, hipMemcpyAsync(&h, d_hdr, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    SFE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const int m = h.n_seg;
    SFE_HIP(ctx, hipMemcpyAsync(out, d_out, sizeof(float2) * (size_t)m, hipMemcpyDeviceToHost, ctx->stream));
    if (out_idx)
        SFE_HIP(ctx, hipMemcpyAsync(out_idx, d_oidx, 4 * (size_t)m, hipMemcpyDeviceToHost, ctx->stream));
    SFE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *n_out = m;
    return 0;
}

} // extern "C"
