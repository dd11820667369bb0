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
#include "sfe_pose2.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

// One object holds n grids (n = 1 for the host-side entry point; one per session for the store-side one) in a single
// allocation; a device table describes them.
struct CostGridDesc {
    long long word_off; // first word of this grid in d_bits
    int rows, cols, wpr, pad;
    float xmin, ymin;   // slam.py:506 (float32 like the target cloud); unused by the host-side entry point
};
struct CostJob { // what one (job, pose chunk) workgroup scores: a source cloud against a grid
    const float2 *src;
    int n_src, grid;
};

struct sfe_costgrid {
    sfe_ctx *ctx = nullptr;
    int n = 0, hs = 0;
    std::vector<CostGridDesc> desc;
    CostGridDesc *d_desc = nullptr;
    uint32_t *d_bits = nullptr;
    size_t nwords = 0, max_words = 0;
};

#define COST_THREADS 256
#define COST_LDS_WORDS (24 * 1024) // 96 KiB of grid bits in LDS; larger grids are read through L2
#define COST_POSES_PER_BLOCK 8     // poses one workgroup scores against the grid it staged

__device__ __forceinline__ void costgrid_stamp_one(int pr, int pc, int i, int rows, int cols, int wpr, int hs,
                                                   const int32_t *__restrict__ span, uint32_t *__restrict__ bits)
{
    const int rr = pr + i - hs;
    if (rr < 0 || rr >= rows)
        return;
    int c0 = pc + span[2 * i] - hs, c1 = pc + span[2 * i + 1] - hs; // [c0, c1)
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
    costgrid_stamp_one(tr[p], tc[p], i, rows, cols, wpr, hs, span, bits);
}

// The same from a device-resident target cloud (float32 points, what pcl.downsample returned): slam.py:514-517
//     r = np.int32(np.round((target_points[:, 1] - ymin) / resolution)), clipped to the grid
// in the cloud's dtype: float32 subtraction, float32 division by float32(resolution), half-to-even.  grid y = job.
__global__ __launch_bounds__(256) void costgrid_stamp_points_kernel(const CostJob *__restrict__ tgt,
                                                                    const CostGridDesc *__restrict__ desc, int hs,
                                                                    float res, const int32_t *__restrict__ span,
                                                                    uint32_t *__restrict__ bits_all)
{
    const CostJob job = tgt[blockIdx.y];
    const CostGridDesc d = desc[job.grid];
    const int size = 2 * hs + 1;
    uint32_t *bits = bits_all + d.word_off;
    for (long long gid = (long long)blockIdx.x * 256 + threadIdx.x; gid < (long long)job.n_src * size;
         gid += (long long)gridDim.x * 256) {
        const int p = (int)(gid / size), i = (int)(gid % size);
        const float2 q = job.src[p];
        const float fr = rintf(__fdiv_rn(__fadd_rn(q.y, -d.ymin), res)), fc = rintf(__fdiv_rn(__fadd_rn(q.x, -d.xmin), res));
        if (!(fr == fr) || !(fc == fc))
            continue; // (a NaN point: np.int32(nan) is undefined in the reference; no cell here)
        const int pr = (int)fminf(fmaxf(fr, 0.0f), (float)(d.rows - 1)), pc = (int)fminf(fmaxf(fc, 0.0f), (float)(d.cols - 1));
        costgrid_stamp_one(pr, pc, i, d.rows, d.cols, d.wpr, hs, span, bits);
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

// bounding boxes of n clouds (float32 min / max: exact), one workgroup each: out[j] = {min x, min y, max x, max y}
__global__ __launch_bounds__(256) void cost_bbox_kernel(const CostJob *__restrict__ clouds, float4 *__restrict__ out)
{
    __shared__ float s_v[4][4];
    const CostJob job = clouds[blockIdx.x];
    float mnx = INFINITY, mny = INFINITY, mxx = -INFINITY, mxy = -INFINITY;
    for (int i = threadIdx.x; i < job.n_src; i += 256) {
        const float2 p = job.src[i];
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
        const int w = threadIdx.x >> 6;
        s_v[0][w] = mnx;
        s_v[1][w] = mny;
        s_v[2][w] = mxx;
        s_v[3][w] = mxy;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) {
            mnx = fminf(mnx, s_v[0][w]);
            mny = fminf(mny, s_v[1][w]);
            mxx = fmaxf(mxx, s_v[2][w]);
            mxy = fmaxf(mxy, s_v[3][w]);
        }
        out[blockIdx.x] = make_float4(mnx, mny, mxx, mxy);
    }
}

// `subroutine` (slam.py:536-568) for the poses [blockIdx.x * PPB, ...) of job blockIdx.y.  The moved point and its cell
// are computed in the dtype numpy computes them in:
//   F64 (SFE_COST_F64_POINTS): the source is a float64 array of float32 values (the SLAM node's keyframe clouds,
//     slam_ros.py:169-170): products exact in double, one rounding of their sum, translation added in double,
//     (p - float32 xmin) / resolution in double (resolution = point_noise / 10.0, a Python float), np.round, np.int32;
//   else: a float32 cloud (what get_points / pcl.downsample return, the NSSM source): sgemm accumulates over k with fused
//     multiply-adds (Keyframe.transform_points pinned by tests/golden/transform_points.npz), the rest in float32.
template <bool F64, bool IN_LDS>
__global__ __launch_bounds__(COST_THREADS) void matching_cost_kernel(const uint32_t *__restrict__ bits_all,
                                                                     const CostGridDesc *__restrict__ desc,
                                                                     const CostJob *__restrict__ jobs,
                                                                     const float *__restrict__ T6, int n_poses,
                                                                     float res32, double res64, int use_desc_origin,
                                                                     float xmin_arg, float ymin_arg,
                                                                     int32_t *__restrict__ cost)
{
    extern __shared__ uint32_t s_bits[];
    __shared__ int s_part[COST_THREADS / 64];
    const CostJob job = jobs[blockIdx.y];
    const CostGridDesc d = desc[job.grid];
    const int rows = d.rows, cols = d.cols, wpr = d.wpr;
    const float xmin = use_desc_origin ? d.xmin : xmin_arg, ymin = use_desc_origin ? d.ymin : ymin_arg;
    const uint32_t *__restrict__ bits = bits_all + d.word_off;
    const int p0 = blockIdx.x * COST_POSES_PER_BLOCK, p1 = min(p0 + COST_POSES_PER_BLOCK, n_poses);
    if (IN_LDS) {
        const int nwords = rows * wpr;
        for (int i = threadIdx.x; i < nwords; i += COST_THREADS)
            s_bits[i] = bits[i];
        __syncthreads();
    }
    const uint32_t *__restrict__ B = IN_LDS ? (const uint32_t *)s_bits : bits;
    for (int p = p0; p < p1; ++p) {
        const float *T = T6 + 6 * ((size_t)blockIdx.y * n_poses + p);
        const float t00 = T[0], t01 = T[1], t02 = T[2], t10 = T[3], t11 = T[4], t12 = T[5];
        int hits = 0;
        for (int i = threadIdx.x; i < job.n_src; i += COST_THREADS) {
            const float2 q = job.src[i];
            bool inside;
            int r, c;
            if (F64) {
                const double x = __dadd_rn(__dadd_rn(__dmul_rn((double)q.x, (double)t00), __dmul_rn((double)q.y, (double)t01)), (double)t02);
                const double y = __dadd_rn(__dadd_rn(__dmul_rn((double)q.x, (double)t10), __dmul_rn((double)q.y, (double)t11)), (double)t12);
                const double fc = rint(__ddiv_rn(__dadd_rn(x, -(double)xmin), res64)), fr = rint(__ddiv_rn(__dadd_rn(y, -(double)ymin), res64));
                inside = fr >= 0.0 && fr < (double)rows && fc >= 0.0 && fc < (double)cols; // NaN fails
                r = inside ? (int)fr : 0;
                c = inside ? (int)fc : 0;
            } else {
                const float x = __fadd_rn(__fmaf_rn(q.y, t01, __fmul_rn(q.x, t00)), t02);
                const float y = __fadd_rn(__fmaf_rn(q.y, t11, __fmul_rn(q.x, t10)), t12);
                const float fc = rintf(__fdiv_rn(__fadd_rn(x, -xmin), res32)), fr = rintf(__fdiv_rn(__fadd_rn(y, -ymin), res32));
                inside = fr >= 0.0f && fr < (float)rows && fc >= 0.0f && fc < (float)cols;
                r = inside ? (int)fr : 0;
                c = inside ? (int)fc : 0;
            }
            if (inside)
                hits += (B[r * wpr + (c >> 5)] >> (c & 31)) & 1u;
        }
#pragma unroll
        for (int dd = 32; dd >= 1; dd >>= 1)
            hits += __shfl_down(hits, dd);
        __syncthreads(); // (the previous pose's total has been read)
        if ((threadIdx.x & 63) == 0)
            s_part[threadIdx.x >> 6] = hits;
        __syncthreads();
        if (threadIdx.x == 0) {
            int s = 0;
            for (int w = 0; w < COST_THREADS / 64; ++w)
                s += s_part[w];
            cost[(size_t)blockIdx.y * n_poses + p] = -s;
        }
    }
}

// Many poses per job (the 244 poses per session of shgo's replay, shgo_fast.py; round 5): the kernel above gives a workgroup of four
// waves 8 poses and a private copy of the grid in LDS -- at 80 KB per grid that is ONE such workgroup per CU, four waves on a CU that
// holds 32.  Here a workgroup of 16 waves stages the grid once and every wave scores COST_MANY_PW poses of its own against it
// (64 poses per workgroup: a quarter of the stagings, four times the waves; no workgroup-wide reduction: a wave owns its poses).
// F64 only differs from the kernel above in how it finds the cell: rint(fl(d / res)) costs two double divisions per point and
// pose; q = d * fl(1 / res) is within 4e-16 |q| of the correctly rounded quotient, so rint(q) IS the cell unless q lies within
// 1e-9 (|q| + 1) of a half-way point -- only then (and for non-finite q) the division is done.  Same integers, always.
#define COST_MANY_THREADS 1024
#define COST_MANY_PW 4 // poses per wave
__device__ __forceinline__ double cost_cell_f64(double d, double res, double rinv)
{
    const double q = d * rinv, n = rint(q);
    if (fabs(fabs(q - n) - 0.5) > 1e-9 * (fabs(q) + 1.0)) // (NaN: false)
        return n;
    return rint(__ddiv_rn(d, res));
}

template <bool F64>
__global__ __launch_bounds__(COST_MANY_THREADS) void matching_cost_many_kernel(const uint32_t *__restrict__ bits_all,
                                                                               const CostGridDesc *__restrict__ desc,
                                                                               const CostJob *__restrict__ jobs,
                                                                               const float *__restrict__ T6, int n_poses, float res32,
                                                                               double res64, int use_desc_origin, float xmin_arg,
                                                                               float ymin_arg, int32_t *__restrict__ cost)
{
    extern __shared__ uint32_t s_bits[];
    const CostJob job = jobs[blockIdx.y];
    const CostGridDesc d = desc[job.grid];
    const int rows = d.rows, cols = d.cols, wpr = d.wpr;
    const float xmin = use_desc_origin ? d.xmin : xmin_arg, ymin = use_desc_origin ? d.ymin : ymin_arg;
    const uint32_t *__restrict__ bits = bits_all + d.word_off;
    const int nwords = rows * wpr;
    for (int i = threadIdx.x; i < nwords; i += COST_MANY_THREADS)
        s_bits[i] = bits[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p0 = (blockIdx.x * (COST_MANY_THREADS / 64) + wave) * COST_MANY_PW;
    if (p0 >= n_poses)
        return;
    float t[COST_MANY_PW][6];
#pragma unroll
    for (int k = 0; k < COST_MANY_PW; ++k) {
        const float *T = T6 + 6 * ((size_t)blockIdx.y * n_poses + min(p0 + k, n_poses - 1));
#pragma unroll
        for (int e = 0; e < 6; ++e)
            t[k][e] = T[e];
    }
    const double rinv = 1.0 / res64;
    int hits[COST_MANY_PW] = {0, 0, 0, 0};
    static_assert(COST_MANY_PW == 4, "hits initialiser");
    for (int i = lane; i < job.n_src; i += 64) {
        const float2 q = job.src[i];
#pragma unroll
        for (int k = 0; k < COST_MANY_PW; ++k) {
            bool inside;
            int r, c;
            if (F64) {
                const double x = __dadd_rn(__dadd_rn(__dmul_rn((double)q.x, (double)t[k][0]), __dmul_rn((double)q.y, (double)t[k][1])), (double)t[k][2]);
                const double y = __dadd_rn(__dadd_rn(__dmul_rn((double)q.x, (double)t[k][3]), __dmul_rn((double)q.y, (double)t[k][4])), (double)t[k][5]);
                const double fc = cost_cell_f64(__dadd_rn(x, -(double)xmin), res64, rinv), fr = cost_cell_f64(__dadd_rn(y, -(double)ymin), res64, rinv);
                inside = fr >= 0.0 && fr < (double)rows && fc >= 0.0 && fc < (double)cols; // NaN fails
                r = inside ? (int)fr : 0;
                c = inside ? (int)fc : 0;
            } else {
                const float x = __fadd_rn(__fmaf_rn(q.y, t[k][1], __fmul_rn(q.x, t[k][0])), t[k][2]);
                const float y = __fadd_rn(__fmaf_rn(q.y, t[k][4], __fmul_rn(q.x, t[k][3])), t[k][5]);
                const float fc = rintf(__fdiv_rn(__fadd_rn(x, -xmin), res32)), fr = rintf(__fdiv_rn(__fadd_rn(y, -ymin), res32));
                inside = fr >= 0.0f && fr < (float)rows && fc >= 0.0f && fc < (float)cols;
                r = inside ? (int)fr : 0;
                c = inside ? (int)fc : 0;
            }
            if (inside)
                hits[k] += (s_bits[r * wpr + (c >> 5)] >> (c & 31)) & 1u;
        }
    }
#pragma unroll
    for (int k = 0; k < COST_MANY_PW; ++k) {
        int h = hits[k];
#pragma unroll
        for (int dd = 32; dd >= 1; dd >>= 1)
            h += __shfl_down(h, dd);
        if (lane == 0 && p0 + k < n_poses)
            cost[(size_t)blockIdx.y * n_poses + p0 + k] = -h;
    }
}

// cv2.getStructuringElement(MORPH_ELLIPSE, (2h+1, 2h+1), (h, h)) row spans
static std::vector<int32_t> cost_ellipse_spans(int dilate_hs)
{
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
    return span;
}

static void costgrid_free(sfe_costgrid *g)
{
    if (g->d_bits)
        (void)hipFree(g->d_bits);
    if (g->d_desc)
        (void)hipFree(g->d_desc);
    delete g;
}

// allocate n grids of the given shapes (zeroed, descriptors uploaded); enqueue only
static int costgrid_alloc(sfe_ctx *ctx, int n, const int *rows, const int *cols, const float *xmin, const float *ymin,
                          int dilate_hs, sfe_costgrid **out)
{
    sfe_costgrid *g = new sfe_costgrid;
    g->ctx = ctx;
    g->n = n;
    g->hs = dilate_hs;
    g->desc.resize((size_t)n);
    size_t off = 0;
    for (int i = 0; i < n; ++i) {
        CostGridDesc &d = g->desc[i];
        d.rows = rows[i];
        d.cols = cols[i];
        d.wpr = (cols[i] + 31) / 32;
        d.pad = 0;
        d.word_off = (long long)off;
        d.xmin = xmin ? xmin[i] : 0.0f;
        d.ymin = ymin ? ymin[i] : 0.0f;
        const size_t w = (size_t)d.rows * d.wpr;
        g->max_words = std::max(g->max_words, w);
        off += w;
    }
    g->nwords = off;
    if (hipMalloc(&g->d_bits, sizeof(uint32_t) * std::max<size_t>(off, 1)) != hipSuccess ||
        hipMalloc(&g->d_desc, sizeof(CostGridDesc) * (size_t)n) != hipSuccess) {
        costgrid_free(g);
        return sfe_set_err(ctx, SFE_ERR_HIP, "hipMalloc(%zu words) for %d cost grid(s) failed", off, n);
    }
    if (hipMemsetAsync(g->d_bits, 0, sizeof(uint32_t) * std::max<size_t>(off, 1), ctx->stream) != hipSuccess ||
        hipMemcpyAsync(g->d_desc, g->desc.data(), sizeof(CostGridDesc) * (size_t)n, hipMemcpyHostToDevice, ctx->stream) !=
            hipSuccess) { // (desc lives as long as the object; the copy is drained before anybody frees it)
        costgrid_free(g);
        return sfe_set_err(ctx, SFE_ERR_HIP, "cost grid: memset / descriptor upload failed");
    }
    *out = g;
    return 0;
}

// score n_jobs x n_poses transforms (device job table, device T6) -> host costs; one synchronisation
static int cost_launch(sfe_ctx *ctx, sfe_costgrid *g, const CostJob *d_jobs, int n_jobs, const float *d_T, int n_poses,
                       double resolution, int flags, int use_desc_origin, float xmin, float ymin, int32_t *cost_out)
{
    const size_t n_out = (size_t)n_jobs * n_poses;
    int32_t *d_cost = (int32_t *)sfe_scratch(ctx, 2, sizeof(int32_t) * n_out);
    int32_t *h_cost = (int32_t *)sfe_pinned_io(ctx, 3, sizeof(int32_t) * n_out);
    if (!d_cost || !h_cost)
        return SFE_ERR_HIP;
    const bool f64 = (flags & SFE_COST_F64_POINTS) != 0, lds = g->max_words <= COST_LDS_WORDS;
    const size_t smem = lds ? sizeof(uint32_t) * g->max_words : 0;
    auto kern = f64 ? (lds ? matching_cost_kernel<true, true> : matching_cost_kernel<true, false>)
                    : (lds ? matching_cost_kernel<false, true> : matching_cost_kernel<false, false>);
    if (lds)
        SFE_HIP(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    const bool no_many = getenv("SFE_COST_NO_MANY") != nullptr; // A/B (read per call): the 8-poses-per-workgroup kernel for every launch
    if (lds && n_poses >= 32 && !no_many) {
        // many poses per job: 64 per workgroup of 16 waves around one staged grid
        auto many = f64 ? matching_cost_many_kernel<true> : matching_cost_many_kernel<false>;
        SFE_HIP(ctx, hipFuncSetAttribute((const void *)many, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        const int ppb = (COST_MANY_THREADS / 64) * COST_MANY_PW;
        hipLaunchKernelGGL(many, dim3((unsigned)((n_poses + ppb - 1) / ppb), (unsigned)n_jobs), dim3(COST_MANY_THREADS), smem,
                           ctx->stream, (const uint32_t *)g->d_bits, (const CostGridDesc *)g->d_desc, d_jobs, d_T, n_poses,
                           (float)resolution, resolution, use_desc_origin, xmin, ymin, d_cost);
    } else {
        const dim3 grid((unsigned)((n_poses + COST_POSES_PER_BLOCK - 1) / COST_POSES_PER_BLOCK), (unsigned)n_jobs);
        hipLaunchKernelGGL(kern, grid, dim3(COST_THREADS), smem, ctx->stream, (const uint32_t *)g->d_bits,
                           (const CostGridDesc *)g->d_desc, d_jobs, d_T, n_poses, (float)resolution, resolution, use_desc_origin,
                           xmin, ymin, d_cost);
    }
    SFE_LAUNCH_CHECK(ctx);
    SFE_HIP(ctx, hipMemcpyAsync(h_cost, d_cost, sizeof(int32_t) * n_out, hipMemcpyDeviceToHost, ctx->stream));
    SFE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(cost_out, h_cost, sizeof(int32_t) * n_out);
    return 0;
}

// {pool + offset, count, grid index} of n store clouds, uploaded to scratch slot `slot`
static int cost_store_jobs(sfe_ctx *ctx, sfe_cloud_store *s, const int32_t *handles, int n, int slot, const char *what,
                           CostJob **d_out, const int32_t *grid_index = nullptr)
{
    SfeStoreView v;
    if (int rc = sfe_store_view(s, &v))
        return rc;
    CostJob *h = (CostJob *)sfe_pinned_begin(ctx, sizeof(CostJob) * (size_t)n);
    CostJob *d = (CostJob *)sfe_scratch(ctx, slot, sizeof(CostJob) * (size_t)n);
    if (!h || !d)
        return SFE_ERR_HIP;
    for (int i = 0; i < n; ++i) {
        const int hd = handles[i];
        if (hd < 0 || hd >= v.n_slots || v.cnt[hd] < 0) {
            (void)sfe_pinned_end(ctx, ctx->stream);
            return sfe_set_err(ctx, SFE_ERR_ARG, "%s: cloud %d named (job %d), the store holds %d%s", what, hd, i, v.n_slots,
                               (hd >= 0 && hd < v.n_slots) ? " and that one was not stored" : "");
        }
        h[i].src = (const float2 *)v.d_pool + v.off[hd];
        h[i].n_src = v.cnt[hd];
        h[i].grid = grid_index ? grid_index[i] : i;
    }
    SFE_HIP(ctx, hipMemcpyAsync(d, h, sizeof(CostJob) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    if (int rc = sfe_pinned_end(ctx, ctx->stream))
        return rc;
    *d_out = d;
    return 0;
}

// T6 [(i * n_deltas + j) * 6 ..] of target_i.between(source_i.compose(delta_j)): sfe_pose2.h
__global__ __launch_bounds__(256) void cost_sample_transforms_kernel(const double *__restrict__ target, const double *__restrict__ source,
                                                                     const double *__restrict__ delta, int n_jobs, int n_deltas,
                                                                     float *__restrict__ T6)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)n_jobs * n_deltas)
        return;
    const int i = (int)(t / n_deltas), j = (int)(t - (long long)i * n_deltas);
    const SfeP2 tgt{target[4 * i], target[4 * i + 1], target[4 * i + 2], target[4 * i + 3]};
    const SfeP2 src{source[4 * i], source[4 * i + 1], source[4 * i + 2], source[4 * i + 3]};
    const SfeP2 d{delta[4 * j], delta[4 * j + 1], delta[4 * j + 2], delta[4 * j + 3]};
    float out[6];
    sfe_p2_sample_transform(sfe_p2_inverse(tgt), src, d, out);
#pragma unroll
    for (int k = 0; k < 6; ++k)
        T6[6 * t + k] = out[k];
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
    sfe_costgrid *g = nullptr;
    if (int rc = costgrid_alloc(ctx, 1, &rows, &cols, nullptr, nullptr, dilate_hs, &g))
        return rc;
    auto fail = [&](int rc) {
        (void)hipStreamSynchronize(ctx->stream);
        costgrid_free(g);
        return rc;
    };
    if (n_tgt > 0) {
        const int size = 2 * dilate_hs + 1;
        const std::vector<int32_t> span = cost_ellipse_spans(dilate_hs);
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
                           d_c, n_tgt, rows, cols, g->desc[0].wpr, dilate_hs, d_span, g->d_bits);
        if (hipGetLastError() != hipSuccess)
            return fail(sfe_set_err(ctx, SFE_ERR_HIP, "cost grid stamp kernel failed"));
    }
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) // (the pageable uploads above are drained before the call returns)
        return fail(sfe_set_err(ctx, SFE_ERR_HIP, "cost grid stamp kernel failed"));
    *out = g;
    return 0;
}

int sfe_cloud_store_bbox(sfe_ctx *ctx, sfe_cloud_store *s, const int32_t *handles, int n, float *bbox_out)
{
    if (int rc = sfe_use(ctx))
        return rc;
    SFE_ARG(ctx, s && sfe_store_ctx(s) == ctx && n >= 0 && (n == 0 || (handles && bbox_out)));
    if (n == 0)
        return 0;
    CostJob *d_jobs = nullptr;
    if (int rc = cost_store_jobs(ctx, s, handles, n, 0, "bbox", &d_jobs))
        return rc;
    float4 *d_bb = (float4 *)sfe_scratch(ctx, 1, sizeof(float4) * (size_t)n);
    float *h_bb = (float *)sfe_pinned_io(ctx, 3, sizeof(float4) * (size_t)n);
    if (!d_bb || !h_bb)
        return SFE_ERR_HIP;
    hipLaunchKernelGGL(cost_bbox_kernel, dim3((unsigned)n), dim3(256), 0, ctx->stream, (const CostJob *)d_jobs, d_bb);
    SFE_LAUNCH_CHECK(ctx);
    SFE_HIP(ctx, hipMemcpyAsync(h_bb, d_bb, sizeof(float4) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    SFE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(bbox_out, h_bb, sizeof(float4) * (size_t)n);
    return 0;
}

int sfe_costgrid_create_store(sfe_ctx *ctx, sfe_cloud_store *s, const int32_t *target_handles, int n, const float *xmin,
                              const float *ymin, float resolution, const int32_t *rows, const int32_t *cols, int dilate_hs,
                              sfe_costgrid **out)
{
    if (int rc = sfe_use(ctx))
        return rc;
    SFE_ARG(ctx, out != nullptr);
    *out = nullptr;
    SFE_ARG(ctx, s && sfe_store_ctx(s) == ctx && n > 0 && target_handles && xmin && ymin && rows && cols && dilate_hs >= 0 &&
                     resolution > 0.0f);
    long long words = 0;
    for (int i = 0; i < n; ++i) {
        SFE_ARG(ctx, rows[i] > 0 && cols[i] > 0);
        words += (long long)rows[i] * ((cols[i] + 31) / 32);
    }
    SFE_ARG(ctx, words < (1LL << 30));
    CostJob *d_jobs = nullptr;
    if (int rc = cost_store_jobs(ctx, s, target_handles, n, 0, "cost grid", &d_jobs))
        return rc;
    sfe_costgrid *g = nullptr;
    if (int rc = costgrid_alloc(ctx, n, rows, cols, xmin, ymin, dilate_hs, &g))
        return rc;
    const std::vector<int32_t> span = cost_ellipse_spans(dilate_hs);
    int32_t *h_span = (int32_t *)sfe_pinned_begin(ctx, sizeof(int32_t) * span.size());
    int32_t *d_span = (int32_t *)sfe_scratch(ctx, 2, sizeof(int32_t) * span.size());
    auto fail = [&](int rc) {
        (void)hipStreamSynchronize(ctx->stream);
        costgrid_free(g);
        return rc;
    };
    if (!h_span || !d_span)
        return fail(SFE_ERR_HIP);
    memcpy(h_span, span.data(), sizeof(int32_t) * span.size());
    if (hipMemcpyAsync(d_span, h_span, sizeof(int32_t) * span.size(), hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
        return fail(sfe_set_err(ctx, SFE_ERR_HIP, "cost grid upload failed"));
    if (int rc = sfe_pinned_end(ctx, ctx->stream))
        return fail(rc);
    hipLaunchKernelGGL(costgrid_stamp_points_kernel, dim3(32, (unsigned)n), dim3(256), 0, ctx->stream, (const CostJob *)d_jobs,
                       (const CostGridDesc *)g->d_desc, dilate_hs, resolution, (const int32_t *)d_span, g->d_bits);
    if (hipGetLastError() != hipSuccess)
        return fail(sfe_set_err(ctx, SFE_ERR_HIP, "cost grid stamp kernel failed"));
    *out = g; // enqueue only: the scoring calls run on the same stream
    return 0;
}

void sfe_costgrid_destroy(sfe_costgrid *g)
{
    if (!g)
        return;
    if (g->ctx && hipSetDevice(g->ctx->device) == hipSuccess)
        (void)hipStreamSynchronize(g->ctx->stream);
    costgrid_free(g);
}

int sfe_costgrid_download(sfe_ctx *ctx, sfe_costgrid *g, int index, uint8_t *grid_out)
{
    if (int rc = sfe_use(ctx))
        return rc;
    SFE_ARG(ctx, g && g->ctx == ctx && grid_out && index >= 0 && index < g->n);
    const CostGridDesc &dd = g->desc[index];
    const size_t n = (size_t)dd.rows * dd.cols;
    uint8_t *d_out = (uint8_t *)sfe_scratch(ctx, 3, n);
    if (!d_out)
        return SFE_ERR_HIP;
    hipLaunchKernelGGL(costgrid_expand_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const uint32_t *)(g->d_bits + dd.word_off), dd.rows, dd.cols, dd.wpr, d_out);
    SFE_LAUNCH_CHECK(ctx);
    SFE_HIP(ctx, hipMemcpyAsync(grid_out, d_out, n, hipMemcpyDeviceToHost, ctx->stream));
    SFE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int sfe_matching_cost_batch(sfe_ctx *ctx, sfe_costgrid *g, const float *src, int n_src, const float *T6, int n_poses,
                            float xmin, float ymin, double resolution, int flags, int32_t *cost_out)
{
    if (int rc = sfe_use(ctx))
        return rc;
    SFE_ARG(ctx, g && g->ctx == ctx && g->n == 1 && n_src >= 0 && n_poses >= 0 && (n_src == 0 || src) &&
                     (n_poses == 0 || (T6 && cost_out)));
    if (n_poses == 0)
        return 0;
    const size_t b_src = sizeof(float) * 2 * (size_t)std::max(n_src, 1), b_T = sizeof(float) * 6 * (size_t)n_poses;
    char *d_in = (char *)sfe_scratch(ctx, 0, b_src + b_T + sizeof(CostJob));
    char *h_in = (char *)sfe_pinned_begin(ctx, b_src + b_T + sizeof(CostJob));
    if (!d_in || !h_in)
        return SFE_ERR_HIP;
    if (n_src)
        memcpy(h_in, src, sizeof(float) * 2 * (size_t)n_src);
    memcpy(h_in + b_src, T6, b_T);
    CostJob job;
    job.src = (const float2 *)d_in;
    job.n_src = n_src;
    job.grid = 0;
    memcpy(h_in + b_src + b_T, &job, sizeof job);
    SFE_HIP(ctx, hipMemcpyAsync(d_in, h_in, b_src + b_T + sizeof(CostJob), hipMemcpyHostToDevice, ctx->stream));
    if (int rc = sfe_pinned_end(ctx, ctx->stream))
        return rc;
    return cost_launch(ctx, g, (const CostJob *)(d_in + b_src + b_T), 1, (const float *)(d_in + b_src), n_poses, resolution,
                       flags, 0, xmin, ymin, cost_out);
}

int sfe_matching_cost_store(sfe_ctx *ctx, sfe_costgrid *g, sfe_cloud_store *s, const int32_t *source_handles,
                            const int32_t *grid_index, int n_jobs, const float *T6, int n_poses, double resolution, int flags,
                            int32_t *cost_out)
{
    if (int rc = sfe_use(ctx))
        return rc;
    SFE_ARG(ctx, g && g->ctx == ctx && s && sfe_store_ctx(s) == ctx && source_handles && n_poses >= 0 && n_jobs >= 0 &&
                     (grid_index || n_jobs == g->n) && (n_poses == 0 || n_jobs == 0 || (T6 && cost_out)));
    if (n_poses == 0 || n_jobs == 0)
        return 0;
    if (grid_index)
        for (int i = 0; i < n_jobs; ++i)
            SFE_ARG(ctx, grid_index[i] >= 0 && grid_index[i] < g->n);
    CostJob *d_jobs = nullptr;
    if (int rc = cost_store_jobs(ctx, s, source_handles, n_jobs, 0, "matching cost", &d_jobs, grid_index))
        return rc;
    const size_t b_T = sizeof(float) * 6 * (size_t)n_poses * (size_t)n_jobs;
    float *d_T = (float *)sfe_scratch(ctx, 1, b_T);
    float *h_T = (float *)sfe_pinned_begin(ctx, b_T);
    if (!d_T || !h_T)
        return SFE_ERR_HIP;
    memcpy(h_T, T6, b_T);
    SFE_HIP(ctx, hipMemcpyAsync(d_T, h_T, b_T, hipMemcpyHostToDevice, ctx->stream));
    if (int rc = sfe_pinned_end(ctx, ctx->stream))
        return rc;
    return cost_launch(ctx, g, d_jobs, n_jobs, d_T, n_poses, resolution, flags, 1, 0.0f, 0.0f, cost_out);
}

// ... with the sample transforms computed on the device from the poses: job i scores source_handles[i] against its grid under
// target_i.between(source_i.compose(delta_j)) for the n_deltas deltas shared by all jobs (the 244 poses shgo's replay needs per
// session, shgo_fast.py) -- 2 x n_jobs + n_deltas poses go up instead of n_jobs x n_deltas transforms
int sfe_matching_cost_store_samples(sfe_ctx *ctx, sfe_costgrid *g, sfe_cloud_store *s, const int32_t *source_handles,
                                    const int32_t *grid_index, int n_jobs, const double *target_xycs, const double *source_xycs,
                                    const double *delta_xycs, int n_deltas, double resolution, int flags, int32_t *cost_out)
{
    if (int rc = sfe_use(ctx))
        return rc;
    SFE_ARG(ctx, g && g->ctx == ctx && s && sfe_store_ctx(s) == ctx && source_handles && n_deltas >= 0 && n_jobs >= 0 &&
                     (grid_index || n_jobs == g->n) &&
                     (n_deltas == 0 || n_jobs == 0 || (target_xycs && source_xycs && delta_xycs && cost_out)));
    if (n_deltas == 0 || n_jobs == 0)
        return 0;
    if (grid_index)
        for (int i = 0; i < n_jobs; ++i)
            SFE_ARG(ctx, grid_index[i] >= 0 && grid_index[i] < g->n);
    CostJob *d_jobs = nullptr;
    if (int rc = cost_store_jobs(ctx, s, source_handles, n_jobs, 0, "matching cost", &d_jobs, grid_index))
        return rc;
    const size_t b_pose = sizeof(double) * 4 * (size_t)n_jobs, b_delta = sizeof(double) * 4 * (size_t)n_deltas;
    const size_t b_T = sizeof(float) * 6 * (size_t)n_deltas * (size_t)n_jobs;
    float *d_T = (float *)sfe_scratch(ctx, 1, b_T);
    double *d_p = (double *)sfe_scratch(ctx, 3, 2 * b_pose + b_delta);
    double *h_p = (double *)sfe_pinned_begin(ctx, 2 * b_pose + b_delta);
    if (!d_T || !d_p || !h_p)
        return SFE_ERR_HIP;
    memcpy(h_p, target_xycs, b_pose);
    memcpy(h_p + 4 * (size_t)n_jobs, source_xycs, b_pose);
    memcpy(h_p + 8 * (size_t)n_jobs, delta_xycs, b_delta);
    SFE_HIP(ctx, hipMemcpyAsync(d_p, h_p, 2 * b_pose + b_delta, hipMemcpyHostToDevice, ctx->stream));
    if (int rc = sfe_pinned_end(ctx, ctx->stream))
        return rc;
    const long long total = (long long)n_jobs * n_deltas;
    hipLaunchKernelGGL(cost_sample_transforms_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const double *)d_p, (const double *)(d_p + 4 * (size_t)n_jobs), (const double *)(d_p + 8 * (size_t)n_jobs),
                       n_jobs, n_deltas, d_T);
    SFE_LAUNCH_CHECK(ctx);
    return cost_launch(ctx, g, d_jobs, n_jobs, d_T, n_deltas, resolution, flags, 1, 0.0f, 0.0f, cost_out);
}

} // extern "C"
