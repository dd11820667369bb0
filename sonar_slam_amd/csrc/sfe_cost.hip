// Global-initialisation matching cost on gfx950: the function scipy.optimize.shgo minimises in
// bruce_slam/src/bruce_slam/slam.py:461-570 (get_matching_cost_subroutine1; call sites
// slam.py:692-701 and :952-961), evaluated for MANY candidate transforms per launch.
//
//   grid   : target cells set to 255, dilated by cv2's (2h+1)^2 MORPH_ELLIPSE element      :515-527
//   cost(T): -#{ source points whose rounded cell under T is inside the grid and set }       :549-562
//
// Layout: the dilated grid is a BIT map (uint32 words, row-major, words_per_row = ceil(cols/32)):
// a 30 m x 30 m scene at 5 cm is 600 x 600 cells = 45 KB, which one workgroup keeps in LDS while it
// streams the source cloud once per candidate pose.  Building it = stamping the element's row
// spans at every target cell with atomicOr (dilation with a symmetric element and a constant
// border that never contributes).  Integer results: bit-exact against the oracle.
// Float recipe per point (numpy float32 arithmetic of the reference, no contraction):
//   x' = fl(fl(fl(px*T00) + fl(py*T01)) + T02);  c = rint(fl(fl(x' - xmin) / res))  (half-even)
#include "sfe_internal.h"

#include <cmath>
#include <vector>

struct sfe_costgrid {
    sfe_ctx *ctx = nullptr;
    int rows = 0, cols = 0, wpr = 0, hs = 0;
    uint32_t *d_bits = nullptr;
};

#define COST_THREADS 256
#define COST_LDS_WORDS (24 * 1024) // 96 KiB of grid bits in LDS; larger grids are read through L2

__global__ __launch_bounds__(256) void costgrid_stamp_kernel(const int32_t *__restrict__ tr,
                                                             const int32_t *__restrict__ tc, int n_tgt, int rows,
                                                             int cols, int wpr, int hs,
                                                             const int32_t *__restrict__ span, // (2hs+1) x [j1, j2)
                                                             uint32_t *__restrict__ bits)
{
    const int size = 2 * hs + 1;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)n_tgt * size)
        return;
    const int p = (int)(gid / size), i = (int)(gid % size);
    const int rr = tr[p] + i - hs;
    if (rr < 0 || rr >= rows)
        return;
    int c0 = tc[p] + span[2 * i] - hs, c1 = tc[p] + span[2 * i + 1] - hs; // [c0, c1)
    c0 = max(c0, 0);
    c1 = min(c1, cols);
    if (c0 >= c1)
        return;
    uint32_t *row = bits + (size_t)rr * wpr;
    for (int w = c0 >> 5; w <= (c1 - 1) >> 5; ++w) {
        const int lo = max(c0 - 32 * w, 0), hi = min(c1 - 32 * w, 32); // bit range [lo, hi) of word w
        const uint32_t m = (hi == 32 ? 0xFFFFFFFFu : ((1u << hi) - 1u)) & ~((1u << lo) - 1u);
        atomicOr(&row[w], m);
    }
}

__global__ __launch_bounds__(256) void costgrid_expand_kernel(const uint32_t *__restrict__ bits, int rows, int cols,
                                                              int wpr, uint8_t *__restrict__ out)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)rows * cols)
        return;
    const int r = (int)(i / cols), c = (int)(i % cols);
    out[i] = (bits[(size_t)r * wpr + (c >> 5)] >> (c & 31)) & 1u ? 255 : 0;
}

template <bool IN_LDS>
__global__ __launch_bounds__(COST_THREADS) void matching_cost_kernel(const uint32_t *__restrict__ bits, int rows,
                                                                     int cols, int wpr,
                                                                     const float2 *__restrict__ src, int n_src,
                                                                     const float *__restrict__ T6, float xmin,
                                                                     float ymin, float res, int32_t *__restrict__ cost)
{
    extern __shared__ uint32_t s_bits[];
    const int nwords = rows * wpr;
    if (IN_LDS) {
        for (int i = threadIdx.x; i < nwords; i += COST_THREADS)
            s_bits[i] = bits[i];
        __syncthreads();
    }
    const uint32_t *__restrict__ B = IN_LDS ? (const uint32_t *)s_bits : bits;
    const float *T = T6 + 6 * (size_t)blockIdx.x;
    const float t00 = T[0], t01 = T[1], t02 = T[2], t10 = T[3], t11 = T[4], t12 = T[5];
    int hits = 0;
    for (int i = threadIdx.x; i < n_src; i += COST_THREADS) {
        const float2 p = src[i];
        const float x = __fadd_rn(__fadd_rn(__fmul_rn(p.x, t00), __fmul_rn(p.y, t01)), t02);
        const float y = __fadd_rn(__fadd_rn(__fmul_rn(p.x, t10), __fmul_rn(p.y, t11)), t12);
        const float qc = __fdiv_rn(__fadd_rn(x, -xmin), res), qr = __fdiv_rn(__fadd_rn(y, -ymin), res);
        const float fc = rintf(qc), fr = rintf(qr); // half-even, exact in float
        if (fr >= 0.0f && fr < (float)rows && fc >= 0.0f && fc < (float)cols) { // NaN fails
            const int r = (int)fr, c = (int)fc;
            hits += (B[r * wpr + (c >> 5)] >> (c & 31)) & 1u;
        }
    }
    // block reduction of the hit count
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
        hits += __shfl_down(hits, d);
    __shared__ int s_part[COST_THREADS / 64];
    if ((threadIdx.x & 63) == 0)
        s_part[threadIdx.x >> 6] = hits;
    __syncthreads();
    if (threadIdx.x == 0) {
        int s = 0;
        for (int w = 0; w < COST_THREADS / 64; ++w)
            s += s_part[w];
        cost[blockIdx.x] = -s;
    }
}

extern "C" {

int sfe_costgrid_create(sfe_ctx *ctx, const int32_t *tgt_r, const int32_t *tgt_c, int n_tgt, int rows, int cols,
                        int dilate_hs, sfe_costgrid **out)
{
    if (int rc = sfe_use(ctx))
        return rc;
    SFE_ARG(ctx, out != nullptr);
    *out = nullptr;
    SFE_ARG(ctx, rows > 0 && cols > 0 && n_tgt >= 0 && dilate_hs >= 0 && (n_tgt == 0 || (tgt_r && tgt_c)));
    SFE_ARG(ctx, (long long)rows * ((cols + 31) / 32) < (1LL << 28));
    sfe_costgrid *g = new sfe_costgrid;
    g->ctx = ctx;
    g->rows = rows;
    g->cols = cols;
    g->wpr = (cols + 31) / 32;
    g->hs = dilate_hs;
    const size_t nbytes = sizeof(uint32_t) * (size_t)rows * g->wpr;
    if (hipMalloc(&g->d_bits, nbytes) != hipSuccess) {
        delete g;
        return sfe_set_err(ctx, SFE_ERR_HIP, "hipMalloc(%zu) for the cost grid failed", nbytes);
    }
    auto fail = [&](int rc) {
        (void)hipFree(g->d_bits);
        delete g;
        return rc;
    };
    if (hipMemsetAsync(g->d_bits, 0, nbytes, ctx->stream) != hipSuccess)
        return fail(sfe_set_err(ctx, SFE_ERR_HIP, "hipMemsetAsync failed"));
    if (n_tgt > 0) {
        // cv2.getStructuringElement(MORPH_ELLIPSE, (2h+1, 2h+1), (h, h)) row spans
        const int size = 2 * dilate_hs + 1;
        std::vector<int32_t> span(2 * (size_t)size);
        const int r = dilate_hs, c = dilate_hs;
        const double inv_r2 = r ? 1.0 / ((double)r * r) : 0.0;
        for (int i = 0; i < size; ++i) {
            const int dy = i - r;
            const int dx = (int)std::lrint(c * std::sqrt(((double)r * r - (double)dy * dy) * inv_r2)); // cvRound
            span[2 * i] = std::max(c - dx, 0);
            span[2 * i + 1] = std::min(c + dx + 1, size);
        }
        int32_t *d_r = (int32_t *)sfe_scratch(ctx, 0, sizeof(int32_t) * (size_t)n_tgt);
        int32_t *d_c = (int32_t *)sfe_scratch(ctx, 1, sizeof(int32_t) * (size_t)n_tgt);
        int32_t *d_span = (int32_t *)sfe_scratch(ctx, 2, sizeof(int32_t) * span.size());
        if (!d_r || !d_c || !d_span)
            return fail(SFE_ERR_HIP);
        if (hipMemcpyAsync(d_r, tgt_r, sizeof(int32_t) * (size_t)n_tgt, hipMemcpyHostToDevice, ctx->stream) !=
                hipSuccess ||
            hipMemcpyAsync(d_c, tgt_c, sizeof(int32_t) * (size_t)n_tgt, hipMemcpyHostToDevice, ctx->stream) !=
                hipSuccess ||
            hipMemcpyAsync(d_span, span.data(), sizeof(int32_t) * span.size(), hipMemcpyHostToDevice, ctx->stream) !=
                hipSuccess)
            return fail(sfe_set_err(ctx, SFE_ERR_HIP, "cost grid upload failed"));
        const long long work = (long long)n_tgt * size;
        hipLaunchKernelGGL(costgrid_stamp_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, ctx->stream, d_r,
                           d_c, n_tgt, rows, cols, g->wpr, dilate_hs, d_span, g->d_bits);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess)
            return fail(sfe_set_err(ctx, SFE_ERR_HIP, "cost grid stamp kernel failed"));
    }
    *out = g;
    return 0;
}

void sfe_costgrid_destroy(sfe_costgrid *g)
{
    if (!g)
        return;
    if (g->ctx && hipSetDevice(g->ctx->device) == hipSuccess) {
        (void)hipStreamSynchronize(g->ctx->stream);
        (void)hipFree(g->d_bits);
    }
    delete g;
}

int sfe_costgrid_download(sfe_ctx *ctx, sfe_costgrid *g, uint8_t *grid_out)
{
    if (int rc = sfe_use(ctx))
        return rc;
    SFE_ARG(ctx, g && g->ctx == ctx && grid_out);
    const size_t n = (size_t)g->rows * g->cols;
    uint8_t *d_out = (uint8_t *)sfe_scratch(ctx, 3, n);
    if (!d_out)
        return SFE_ERR_HIP;
    hipLaunchKernelGGL(costgrid_expand_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, g->d_bits,
                       g->rows, g->cols, g->wpr, d_out);
    SFE_LAUNCH_CHECK(ctx);
    SFE_HIP(ctx, hipMemcpyAsync(grid_out, d_out, n, hipMemcpyDeviceToHost, ctx->stream));
    SFE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int sfe_matching_cost_batch(sfe_ctx *ctx, sfe_costgrid *g, const float *src, int n_src, const float *T6, int n_poses,
                            float xmin, float ymin, float resolution, int32_t *cost_out)
{
    if (int rc = sfe_use(ctx))
        return rc;
    SFE_ARG(ctx, g && g->ctx == ctx && n_src >= 0 && n_poses >= 0 && (n_src == 0 || src) &&
                     (n_poses == 0 || (T6 && cost_out)));
    if (n_poses == 0)
        return 0;
    float *d_src = (float *)sfe_scratch(ctx, 0, sizeof(float) * 2 * (size_t)std::max(n_src, 1));
    float *d_T = (float *)sfe_scratch(ctx, 1, sizeof(float) * 6 * (size_t)n_poses);
    int32_t *d_cost = (int32_t *)sfe_scratch(ctx, 2, sizeof(int32_t) * (size_t)n_poses);
    if (!d_src || !d_T || !d_cost)
        return SFE_ERR_HIP;
    if (n_src)
        SFE_HIP(ctx, hipMemcpyAsync(d_src, src, sizeof(float) * 2 * (size_t)n_src, hipMemcpyHostToDevice, ctx->stream));
    SFE_HIP(ctx, hipMemcpyAsync(d_T, T6, sizeof(float) * 6 * (size_t)n_poses, hipMemcpyHostToDevice, ctx->stream));
    const int nwords = g->rows * g->wpr;
    if (nwords <= COST_LDS_WORDS) {
        const size_t smem = sizeof(uint32_t) * (size_t)nwords;
        SFE_HIP(ctx, hipFuncSetAttribute((const void *)matching_cost_kernel<true>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        hipLaunchKernelGGL(matching_cost_kernel<true>, dim3(n_poses), dim3(COST_THREADS), smem, ctx->stream, g->d_bits,
                           g->rows, g->cols, g->wpr, (const float2 *)d_src, n_src, d_T, xmin, ymin, resolution, d_cost);
    } else {
        hipLaunchKernelGGL(matching_cost_kernel<false>, dim3(n_poses), dim3(COST_THREADS), 0, ctx->stream, g->d_bits,
                           g->rows, g->cols, g->wpr, (const float2 *)d_src, n_src, d_T, xmin, ymin, resolution, d_cost);
    }
    SFE_LAUNCH_CHECK(ctx);
    SFE_HIP(ctx, hipMemcpyAsync(cost_out, d_cost, sizeof(int32_t) * (size_t)n_poses, hipMemcpyDeviceToHost,
                                ctx->stream));
    SFE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

} // extern "C"
