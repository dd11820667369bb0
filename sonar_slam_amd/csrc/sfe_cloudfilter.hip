// Device-resident tail of FeatureExtraction.callback for a BATCH of pings
// (bruce_slam/src/bruce_slam/feature_extraction.py:241-249):
//     points = pcl.downsample(points, resolution)                   pcl.cpp:128-141
//     points = pcl.remove_outlier(points, radius, min_points)       pcl.cpp:54-74
// Input = what sfe_extract_points_batch_dev leaves in HBM (float64 points, per-frame counts), output =
// the float32 feature clouds the SLAM node receives; nothing visits the host in between (SURVEY 8
// row f4).  Same algorithms, same float recipes and the same results as the single-cloud entry
// points sfe_downsample (sfe_downsample.hip) and sfe_remove_outlier (sfe_icp.hip); every kernel
// takes its cloud size from the per-frame count in device memory and the frame index from
// blockIdx.x.
//
// downsample = libpointmatcher OctreeGridDataPointsFilter restated without a tree (see
// sfe_downsample.hip): 2L-bit path key per point, stable sort by (key, index), one leaf per run of
// equal keys, medoid per leaf.  Here the stable sort is an in-LDS bitonic sort of (key << 16 | index)
// per frame (<= 16384 points, key <= 48 bits); clouds or trees beyond that take the rank-counting
// sort of sfe_downsample.hip.
#include "sfe_cloudfilter.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>

// one workgroup per frame: float64 -> float32 (what pybind does at pcl.cpp's boundary), bounding box,
// octree root and depth
__global__ __launch_bounds__(1024) void cf_cast_bbox_kernel(const double *__restrict__ pts64,
                                                            const int32_t *__restrict__ counts, long long cap,
                                                            float max_size, float2 *__restrict__ p32,
                                                            CfHeader *__restrict__ hdrs)
{
    __shared__ float s_mn[2][16], s_mx[2][16];
    const int f = blockIdx.x;
    const int n = (int)min((long long)max(counts[f], 0), cap);
    const double *src = pts64 + (size_t)f * cap * 2;
    float2 *dst = p32 + (size_t)f * cap;
    float mnx = INFINITY, mny = INFINITY, mxx = -INFINITY, mxy = -INFINITY;
    for (int i = threadIdx.x; i < n; i += 1024) {
        const float2 p = make_float2((float)src[2 * i], (float)src[2 * i + 1]);
        dst[i] = p;
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
        const CfHeader h = cf_make_header(mnx, mny, mxx, mxy, max_size, n);
        hdrs[f] = h;
    }
}

__device__ __forceinline__ unsigned long long cf_path_key(float2 p, const CfHeader &h)
{
    float cx = h.cx, cy = h.cy, r = h.radius;
    unsigned long long key = 0;
    for (int l = 0; l < h.levels; ++l) {
        const unsigned bx = p.x > cx, by = p.y > cy; // Octree::idx: bit i = pt(i) > centre(i)
        key = (key << 2) | (bx | (by << 1));
        const float hr = r * 0.5f;
        cx = cx + (bx ? hr : -hr);
        cy = cy + (by ? hr : -hr);
        r = hr;
    }
    return key;
}

// one workgroup per frame: keys, stable bitonic sort (keys in LDS, or in HBM scratch for frames beyond
// CF_SORT_CAP points: config-B pings yield ~34k detections), leaves, medoids.
// K = sort key type.  The bitonic sort is bound by LDS bandwidth, and a sonar fan at 0.5 m resolution needs a
// 7-level tree = 14 key bits: with K = uint32 (trees of <= 8 levels) the sort moves half the bytes and two
// workgroups share a CU (64 KiB of keys each instead of one with 128 KiB).  A frame whose tree is deeper marks
// itself CF_NEEDS_WIDE and is taken by the uint64 instantiation, which is launched behind it for those
// frames only.
#define CF_NEEDS_WIDE (-2)
// ... and a frame with more points than the launch's LDS holds (capacities beyond CF_SORT_CAP: round 6) marks itself
// CF_NEEDS_GLOBAL: the instantiation that sorts in HBM scratch, launched last, takes whatever is still marked.  The sort
// is chosen per FRAME, not per batch: one dense ping among thousands no longer sends every frame of the batch to the slow path.
#define CF_NEEDS_GLOBAL (-3)
// frames that marked themselves CF_NEEDS_GLOBAL, for the HBM-scratch launch: a list + its length (the marking thread appends)
__device__ __forceinline__ void cf_mark_global(CfHeader *hdrs, int f, int *marked, int *n_marked)
{
    hdrs[f].n_seg = CF_NEEDS_GLOBAL;
    if (marked)
        marked[atomicAdd(n_marked, 1)] = f;
}

// one frame (the whole workgroup); key_slot: which slice of the HBM key scratch this workgroup sorts in (!IN_LDS)
template <bool IN_LDS, typename K>
__device__ __forceinline__ void cf_downsample_frame(const int f, const int key_slot, const float2 *__restrict__ p32, long long cap,
                                                    CfHeader *__restrict__ hdrs, float2 *__restrict__ ds_out,
                                                    int *__restrict__ seg_all, K *__restrict__ gkeys_all, long long n2cap,
                                                    int only_marked, unsigned *__restrict__ leaf_keys_all, int *marked,
                                                    int *n_marked)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[]; // n2 sort keys (IN_LDS)
    __shared__ int s_scan[1024];
    K *s_keys = IN_LDS ? reinterpret_cast<K *>(lds_raw) : gkeys_all + (size_t)key_slot * n2cap;
    const int tid = threadIdx.x;
    const CfHeader h = hdrs[f];
    if (only_marked && h.n_seg != CF_NEEDS_WIDE && h.n_seg != CF_NEEDS_GLOBAL)
        return;
    const int n = h.n;
    const float2 *pts = p32 + (size_t)f * cap;
    float2 *out = ds_out + (size_t)f * cap;
    if (n == 0)
        return;
    if (IN_LDS && n2cap > 0 && n > n2cap) { // more points than this launch's LDS holds: left to the HBM-scratch instantiation
        if (tid == 0 && h.n_seg != CF_NEEDS_GLOBAL)
            cf_mark_global(hdrs, f, marked, n_marked);
        return;
    }
    constexpr int KEY_BITS = (int)sizeof(K) * 8 - 16;
    if (2 * h.levels > KEY_BITS) {
        // (key << 16 | index) holds KEY_BITS key bits: narrow keys hand the frame to the wide instantiation,
        // which refuses cells finer than extent / 2^24
        if (tid == 0)
            hdrs[f].n_seg = sizeof(K) == 4 ? CF_NEEDS_WIDE : -1;
        return;
    }
    unsigned n2 = 2;
    while (n2 < (unsigned)n)
        n2 <<= 1;
    for (unsigned i = tid; i < n2; i += 1024)
        s_keys[i] = i < (unsigned)n ? (K)((cf_path_key(pts[i], h) << 16) | i) : (K)~(K)0;
    __syncthreads();
    for (unsigned k = 2; k <= n2; k <<= 1)
        for (unsigned j = k >> 1; j > 0; j >>= 1) {
            for (unsigned t = tid; t < n2 / 2; t += 1024) {
                const unsigned i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const unsigned l = i | j;
                const K a = s_keys[i], b = s_keys[l];
                if ((a > b) == ((i & k) == 0)) {
                    s_keys[i] = b;
                    s_keys[l] = a;
                }
            }
            __syncthreads();
        }
    // leaf starts: positions whose key differs from the previous one (block scan over per-thread chunks)
    const int per = (n + 1023) / 1024;
    const int b0 = tid * per, e0 = min(b0 + per, n);
    int c = 0;
    for (int r = b0; r < e0; ++r)
        c += (r == 0) || ((s_keys[r] >> 16) != (s_keys[r - 1] >> 16));
    s_scan[tid] = c;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int v = (tid >= d) ? s_scan[tid - d] : 0;
        __syncthreads();
        s_scan[tid] += v;
        __syncthreads();
    }
    const int n_seg = s_scan[1023];
    int *s_seg = seg_all + (size_t)f * (cap + 1); // n_seg + 1 leaf starts (HBM scratch: written once, read once)
    int s = (tid == 0) ? 0 : s_scan[tid - 1];
    for (int r = b0; r < e0; ++r)
        if ((r == 0) || ((s_keys[r] >> 16) != (s_keys[r - 1] >> 16)))
            s_seg[s++] = r;
    if (tid == 0)
        s_seg[n_seg] = n;
    __syncthreads(); // same workgroup: its own stores are visible to it after the barrier
    // one thread per leaf: float centroid in original order, first point at minimum distance
    for (int sg = tid; sg < n_seg; sg += 1024) {
        const int r0 = s_seg[sg], r1 = s_seg[sg + 1];
        float sx = 0.0f, sy = 0.0f;
        for (int r = r0; r < r1; ++r) {
            const float2 p = pts[(int)(s_keys[r] & 0xFFFFu)];
            sx = __fadd_rn(sx, p.x);
            sy = __fadd_rn(sy, p.y);
        }
        const float cnt = (float)(r1 - r0);
        sx = __fdiv_rn(sx, cnt);
        sy = __fdiv_rn(sy, cnt);
        float best = 3.402823466e+38f;
        int bi = (int)(s_keys[r0] & 0xFFFFu);
        for (int r = r0; r < r1; ++r) {
            const int id = (int)(s_keys[r] & 0xFFFFu);
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
        out[sg] = pts[bi];
        if (sizeof(K) == 4 && leaf_keys_all)
            leaf_keys_all[(size_t)f * cap + sg] = (unsigned)(s_keys[r0] >> 16); // the leaf's path = its Morton code
    }
    if (tid == 0) {
        hdrs[f].n_seg = n_seg;
        hdrs[f].n_out = n_seg;
        hdrs[f].zlev = (sizeof(K) == 4 && leaf_keys_all) ? h.levels : -1;
    }
}

// frame_list == nullptr: one workgroup per frame (blockIdx.x).  Else the workgroups stride over the listed frames (the ones
// that marked themselves for the HBM-scratch sort: rare -- a launch of one 1024-thread workgroup per frame that returns at
// once cost 1.9 ms per 4096 frames, more than the whole filter stage; profiles/r06_final_bench_kernels.txt, first pass)
template <bool IN_LDS, typename K>
__global__ __launch_bounds__(1024) void cf_downsample_kernel(const float2 *__restrict__ p32, long long cap,
                                                             CfHeader *__restrict__ hdrs, float2 *__restrict__ ds_out,
                                                             int *__restrict__ seg_all, K *__restrict__ gkeys_all,
                                                             long long n2cap, int only_marked,
                                                             unsigned *__restrict__ leaf_keys_all, int *marked, int *n_marked,
                                                             const int *__restrict__ frame_list)
{
    if (frame_list) {
        const int nl = *n_marked;
        for (int i = blockIdx.x; i < nl; i += gridDim.x) {
            cf_downsample_frame<IN_LDS, K>(frame_list[i], blockIdx.x, p32, cap, hdrs, ds_out, seg_all, gkeys_all, n2cap, only_marked,
                                           leaf_keys_all, nullptr, nullptr);
            __syncthreads(); // (the next frame reuses the shared scan array and this workgroup's key slice)
        }
    } else {
        cf_downsample_frame<IN_LDS, K>(blockIdx.x, blockIdx.x, p32, cap, hdrs, ds_out, seg_all, gkeys_all, n2cap, only_marked,
                                       leaf_keys_all, marked, n_marked);
    }
}

// exclusive prefix sum of one int per thread over the 1024-thread workgroup (wave scan by shuffles, then the 16 wave
// totals by the first wave): two barriers instead of the twenty of a Hillis-Steele scan through LDS.  s_tmp: >= 17 ints.
__device__ __forceinline__ int cf_block_excl_scan(int v, int *s_tmp, int *total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d);
        if (lane >= d)
            incl += t;
    }
    __syncthreads(); // s_tmp may still be read from a previous call
    if (lane == 63)
        s_tmp[wave] = incl;
    __syncthreads();
    if (wave == 0) {
        const int w = lane < 16 ? s_tmp[lane] : 0;
        int wi = w;
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) {
            const int t = __shfl_up(wi, d);
            if (lane >= d)
                wi += t;
        }
        if (lane < 16)
            s_tmp[lane] = wi - w;
        if (lane == 15)
            s_tmp[16] = wi;
    }
    __syncthreads();
    *total = s_tmp[16];
    return incl - v + s_tmp[wave];
}

// The same downsample for trees of <= 8 levels (16 path-key bits: every sonar fan at the shipped 0.5 m), with the stable
// sort done as an LSD radix sort of the point INDICES in LDS instead of a bitonic sort of (key << 16 | index): the
// bitonic network is 105 stages with a workgroup barrier each for 16384 slots; this is two passes of 8 key bits with
// three barriers each.  A pass: every wave owns a contiguous range of the current order and goes through it 64
// indices at a time; the lanes holding the same digit find each other with 8 ballots (match mask), so the wave's digit
// counts and, in the second sweep, every index's rank inside its digit come out of popcounts in lane order -- stable
// by construction (waves in range order, groups in order, lanes in order) without an atomic.  Between the sweeps one
// scan over (digit, wave) turns the counts into output offsets.  The path key of a point is
// computed once and kept in LDS next to the two uint16 index buffers; afterwards the same LDS (128 KiB for 16384
// points: one frame per CU) holds the frame's points in sorted order for the per-leaf loops.  Leaves, medoids and outputs are exactly those of
// cf_downsample_kernel.
// Round 5: the sort's ranking no longer goes through wave ballots.  Rounds 2-4 ranked 64 indices at a time with a match mask
// from 8 ballots per group (stable by construction, no atomics) -- ~200 wave instructions per 64 elements and sweep, 139 k
// VALU wave-instructions per 11 000-point frame at 69 % VALU-active (profiles/filters_pmc.json): the kernel was bound by
// exactly those instructions.  Now every thread owns a CONTIGUOUS chunk of the current order (m <= 17 elements) and a
// private column of 16 digit counters in LDS; it counts its chunk, the counters are scanned digit-major / thread-minor
// (so equal digits keep the order of their positions: stable), and the thread scatters its chunk in order from its own
// running offsets.  A wave instruction now serves 64 elements instead of one group interaction: ~25 per element and
// pass, four passes of 4 bits for the 14 key bits of a sonar fan at 0.5 m.
#define CF_RDX_BITS 4
#define CF_RDX_DIGITS (1 << CF_RDX_BITS)
// elements per sorting thread: CF_RDX_CHUNK = (16384 / 1024) | 1 for capacities up to CF_SORT_CAP; capacities beyond it take the
// build with CF_RDX_CHUNK_BIG, whose LDS (152 of 160 KB) holds a ping of up to CF_LDS_PTS_BIG detections -- the dense pings of the
// bench's synthetic frames (17 494 on rank 0, 19 122 on rank 3) sort in LDS instead of HBM scratch (0.87 ms per step there)
#define CF_RDX_CHUNK 17
#define CF_RDX_CHUNK_BIG 19
#define CF_LDS_PTS_BIG (CF_RDX_CHUNK_BIG * 1024)
template <int CHUNK>
__global__ __launch_bounds__(1024) void cf_downsample_radix_kernel(const float2 *__restrict__ p32, long long cap,
                                                                   CfHeader *__restrict__ hdrs, float2 *__restrict__ ds_out,
                                                                   int *__restrict__ seg_all, int n2cap,
                                                                   unsigned *__restrict__ leaf_keys_all,
                                                                   float2 *__restrict__ spts_all, int lds_bytes, int sort_cols,
                                                                   int *marked, int *n_marked)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[]; // ids[2][n2cap], key[n2cap], cnt[16][sort_cols]: u16
    __shared__ int s_scan[1024];
    const int f = blockIdx.x, tid = threadIdx.x;
    const CfHeader h = hdrs[f];
    const int n = h.n;
    const float2 *pts = p32 + (size_t)f * cap;
    float2 *out = ds_out + (size_t)f * cap;
    if (n == 0)
        return;
    if (n > n2cap) { // more points than the LDS of this launch holds: the HBM-scratch sort launched last takes the frame
        if (tid == 0)
            cf_mark_global(hdrs, f, marked, n_marked);
        return;
    }
    if (2 * h.levels > 16) { // deeper tree: the 64-bit bitonic instantiation launched behind takes the frame
        if (tid == 0)
            hdrs[f].n_seg = CF_NEEDS_WIDE;
        return;
    }
    unsigned short *idA = reinterpret_cast<unsigned short *>(lds_raw);
    unsigned short *idB = idA + n2cap;
    unsigned short *skey = idB + n2cap; // path key of point i (<= 16 bits), computed once: the sort only moves indices
    unsigned short *cnt = skey + n2cap; // [digit][column]
    for (int i = tid; i < n; i += 1024) {
        idA[i] = (unsigned short)i;
        skey[i] = (unsigned short)cf_path_key(pts[i], h);
    }
    // a column's chunk of the current order: m consecutive positions, m odd so that the columns' 16-bit reads spread
    // over the LDS banks (sort_cols = 1024, or 512 for capacities of <= 8192 points, whose LDS share is 64 KB)
    const int m = ((n + sort_cols - 1) / sort_cols) | 1;
    const bool col = tid < sort_cols;
    const int p0 = col ? min(tid * m, n) : n, p1 = min(p0 + m, n);
    const int passes = (2 * h.levels + CF_RDX_BITS - 1) / CF_RDX_BITS;
    if (col)
#pragma unroll
        for (int d = 0; d < CF_RDX_DIGITS; ++d) {
            const int f0 = d * sort_cols + tid;
            cnt[f0 + (f0 >> 4)] = 0;
        }
    // counter (digit d, column t) lives at flat index f = d * sort_cols + t, padded by one slot per 16 so that both access
    // patterns spread over the LDS banks: a column's own counters (consecutive lanes, consecutive slots) and the scan's
    // 16 consecutive counters per thread (stride 17 slots instead of 16: round 5 counters showed 37 % of the LDS cycles
    // lost to bank conflicts with the unpadded layout)
    auto slot = [&](int f) { return f + (f >> 4); };
    for (int pass = 0; pass < passes; ++pass) {
        const int shift = CF_RDX_BITS * pass;
        __syncthreads(); // idA complete (first pass: the identity and the keys; later: the previous scatter)
        // the chunk's indices and digits: independent LDS reads, in flight together
        unsigned id[CHUNK], dg[CHUNK];
#pragma unroll
        for (int k = 0; k < CHUNK; ++k)
            id[k] = p0 + k < p1 ? idA[p0 + k] : 0u;
#pragma unroll
        for (int k = 0; k < CHUNK; ++k)
            dg[k] = p0 + k < p1 ? ((unsigned)skey[id[k]] >> shift) & (CF_RDX_DIGITS - 1) : 0u;
        // (the column's counters were zeroed behind the previous pass's scatter / before the first pass)
#pragma unroll
        for (int k = 0; k < CHUNK; ++k) // count: the column is this thread's own, plain read-modify-write
            if (p0 + k < p1)
                cnt[slot(dg[k] * sort_cols + tid)] += 1;
        // (measured and dropped: the counts in registers -- two 64-bit words of eight 8-bit counters, the value before an
        // increment being the element's rank in the chunk, so that neither sweep reads or writes a counter: 39 % fewer
        // LDS operations, but 128 VGPRs and ~15 64-bit VALU operations per element: 0.262 -> 0.328 ms per 512 frames)
        __syncthreads();
        { // exclusive scan over (digit major, column minor): 16 * sort_cols counters, 16 (or 8) consecutive ones per thread
            const int per_t = CF_RDX_DIGITS * sort_cols / 1024;
            const int base = slot(per_t * tid); // (per_t consecutive flat indices share their padding: per_t divides 16)
            int v[16], sum = 0;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                v[k] = k < per_t ? cnt[base + k] : 0;
                sum += v[k];
            }
            int tot_;
            int run = cf_block_excl_scan(sum, s_scan, &tot_);
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (k < per_t) {
                    cnt[base + k] = (unsigned short)run;
                    run += v[k];
                }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < CHUNK; ++k) // scatter in position order from the column's running offsets
            if (p0 + k < p1) {
                const unsigned a = slot(dg[k] * sort_cols + tid);
                const unsigned off = cnt[a];
                idB[off] = (unsigned short)id[k];
                cnt[a] = (unsigned short)(off + 1);
            }
        if (col && pass + 1 < passes) // the column's own counters again: no barrier needed before the next pass's counts
#pragma unroll
            for (int d = 0; d < CF_RDX_DIGITS; ++d)
                cnt[slot(d * sort_cols + tid)] = 0;
        unsigned short *t_ = idA;
        idA = idB;
        idB = t_;
    }
    __syncthreads();
    // leaf starts: positions whose key differs from the previous one (block scan over per-thread chunks)
    const int per = (n + 1023) / 1024;
    const int b0 = tid * per, e0 = min(b0 + per, n);
    int c = 0;
    {
        unsigned prev = (b0 > 0 && b0 < n) ? skey[idA[b0 - 1]] : 0u;
        for (int r = b0; r < e0; ++r) {
            const unsigned k = skey[idA[r]];
            c += (r == 0) || (k != prev);
            prev = k;
        }
    }
    int n_seg;
    const int seg_base = cf_block_excl_scan(c, s_scan, &n_seg);
    int *s_seg = seg_all + (size_t)f * (cap + 1); // n_seg + 1 leaf starts (HBM scratch: written once, read once)
    // The points in sorted order, gathered by all threads at once, so that a leaf's thread walks consecutive slots
    // instead of one dependent HBM gather per point (twice).  They go into the LDS the sort no longer needs (the launch
    // sizes it for a full frame) -- the per-leaf loops are chains of dependent loads, 100 cycles each from LDS against 500 from L2 --
    // or, for larger frames, into HBM scratch.  (Measured on a 14.7 k-point frame before this: sort 125 k cycles,
    // per-leaf loops 220 k.)
    float2 *spts_g = spts_all + (size_t)f * cap;
    const bool in_lds = (size_t)n * sizeof(float2) <= (size_t)lds_bytes;
    {
        int sidx = seg_base;
        unsigned prev = (b0 > 0 && b0 < n) ? skey[idA[b0 - 1]] : 0u;
        for (int r = b0; r < e0; ++r) {
            const unsigned k = skey[idA[r]];
            if ((r == 0) || (k != prev))
                s_seg[sidx++] = r;
            prev = k;
        }
    }
    if (tid == 0)
        s_seg[n_seg] = n;
    __syncthreads(); // same workgroup: its own stores are visible to it after the barrier
    if (leaf_keys_all) { // the leaves' paths (= Morton codes) while the keys are still there
        for (int sg = tid; sg < n_seg; sg += 1024)
            leaf_keys_all[(size_t)f * cap + sg] = (unsigned)skey[idA[s_seg[sg]]];
    }
    {
        constexpr int NV = CHUNK == CF_RDX_CHUNK ? 16 : CHUNK; // n <= 16384, or <= CHUNK * 1024 in the build for larger capacities
        float2 v[NV];
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int r = tid + k * 1024;
            v[k] = r < n ? pts[idA[r]] : make_float2(0.0f, 0.0f);
        }
        __syncthreads(); // ids and keys have been read: their LDS is free
        float2 *dst = in_lds ? reinterpret_cast<float2 *>(lds_raw) : spts_g;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int r = tid + k * 1024;
            if (r < n)
                dst[r] = v[k];
        }
    }
    __syncthreads(); // same workgroup: its own stores (LDS or HBM) are visible to it after the barrier
    // one thread per leaf: float centroid in original order, first point at minimum distance
    auto medoids = [&](const auto *spts) {
        for (int sg = tid; sg < n_seg; sg += 1024) {
            const int r0 = s_seg[sg], r1 = s_seg[sg + 1];
            float sx = 0.0f, sy = 0.0f;
            for (int r = r0; r < r1; ++r) {
                const float2 p = spts[r];
                sx = __fadd_rn(sx, p.x);
                sy = __fadd_rn(sy, p.y);
            }
            const float cntf = (float)(r1 - r0);
            sx = __fdiv_rn(sx, cntf);
            sy = __fdiv_rn(sy, cntf);
            float best = 3.402823466e+38f;
            int bi = r0;
            for (int r = r0; r < r1; ++r) {
                const float2 p = spts[r];
                const float dx = __fadd_rn(p.x, -sx), dy = __fadd_rn(p.y, -sy);
                // sqrtf, not __fsqrt_rn (see cf_downsample_kernel)
                const float d = sqrtf(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)));
                if (d < best) {
                    best = d;
                    bi = r;
                }
            }
            out[sg] = spts[bi];
        }
    };
    if (in_lds)
        medoids(reinterpret_cast<const float2 *>(lds_raw)); // (LDS address space: ds_read, not flat)
    else
        medoids(static_cast<const float2 *>(spts_g));
    if (tid == 0) {
        hdrs[f].n_seg = n_seg;
        hdrs[f].n_out = n_seg;
        hdrs[f].zlev = leaf_keys_all ? h.levels : -1;
    }
}

// pcl.remove_outlier: keep a point iff more than min_points points (itself included) lie within the
// radius; order preserved.  One workgroup per frame: counts (cloud tiled through LDS), scan, gather.
//
// Fast path for clouds that come out of the downsample (one medoid per octree leaf, in path = Morton order, leaf
// keys kept): the leaves are grouped into the cells of the coarsest tree level whose cells are still at least
// one radius wide (at most level CF_CELL_LEVELS), a cell is a contiguous run of the sorted cloud, and a point
// only counts the points of the 3 x 3 cells around its own -- every point within the radius lies in one of them.
// The same float distance test on a superset of the points that can pass it: identical counts, ~75 tests per
// point instead of ~2500.
#define CF_CELL_LEVELS 6
#define CF_CELL_PTS 8192 // points the fast path holds in LDS
__device__ __forceinline__ unsigned cf_spread(unsigned v) // bit i -> bit 2i (6 bits)
{
    v = (v | (v << 4)) & 0x30Fu;
    v = (v | (v << 2)) & 0x333u;
    v = (v | (v << 1)) & 0x555u;
    return v;
}
__device__ __forceinline__ unsigned cf_squeeze(unsigned v) // bit 2i -> bit i
{
    v &= 0x555u;
    v = (v | (v >> 1)) & 0x333u;
    v = (v | (v >> 2)) & 0x30Fu;
    v = (v | (v >> 4)) & 0x03Fu;
    return v;
}

__global__ __launch_bounds__(1024) void cf_radius_filter_kernel(const float2 *__restrict__ in_all, long long cap,
                                                                CfHeader *__restrict__ hdrs, float r2, int min_points,
                                                                float *__restrict__ out_all,
                                                                int32_t *__restrict__ out_counts, int do_filter,
                                                                const unsigned *__restrict__ leaf_keys_all)
{
    // dynamic LDS: cell path = the cloud, then the cell starts; brute-force path = a tile of 2048 points (the
    // launcher sizes it for whichever is larger, so two workgroups share a CU either way)
    extern __shared__ __attribute__((aligned(16))) unsigned char cf_dyn[];
    float2 *s_p = reinterpret_cast<float2 *>(cf_dyn);
    __shared__ int s_scan[1024];
    const int f = blockIdx.x, tid = threadIdx.x;
    const CfHeader h = hdrs[f];
    const int n = h.n_seg;
    const float2 *pts = in_all + (size_t)f * cap;
    float2 *out = reinterpret_cast<float2 *>(out_all) + (size_t)f * cap;
    if (n < 0) { // the downsample refused this frame (tree deeper than 24 levels)
        if (tid == 0)
            out_counts[f] = -1;
        return;
    }
    // cell level: deepest level (<= CF_CELL_LEVELS, <= the tree's depth) whose cells are >= radius * 1.001 wide
    int lc = -1;
    if (do_filter && leaf_keys_all && h.zlev >= 0 && n <= CF_CELL_PTS && n > 0) {
        const float need = sqrtf(r2) * 1.001f; // the margin dwarfs the rounding of the tree's cell boundaries
        float width = h.radius * 2.0f;          // level 0 = the root cell
        if (need > 0.0f && width >= need) {     // (NaN -> brute force)
            lc = 0;
            while (lc < CF_CELL_LEVELS && lc < h.zlev && width * 0.5f >= need) {
                width *= 0.5f;
                ++lc;
            }
        }
    }
    float2 *c_p = reinterpret_cast<float2 *>(cf_dyn);
    unsigned short *c_start = reinterpret_cast<unsigned short *>(cf_dyn + sizeof(float2) * CF_CELL_PTS);
    const int ncell = 1 << (2 * max(lc, 0));
    const int sh = 2 * (h.zlev - max(lc, 0));
    if (lc >= 0) {
        const unsigned *keys = leaf_keys_all + (size_t)f * cap;
        for (int c = tid; c <= ncell; c += 1024)
            c_start[c] = (unsigned short)n; // "no leaf at or after this cell" until proven otherwise
        __syncthreads();
        // the first leaf of a cell writes its index as the cell's start AND as the start of the empty cells between the
        // previous occupied cell and its own ("the next occupied cell's start": starts grow with the cell index).  Round 6:
        // one barrier instead of the 2 log2(ncell) of a suffix-minimum sweep over the cells (24 at the deepest level).
        for (int i = tid; i < n; i += 1024) {
            c_p[i] = pts[i];
            const unsigned c = keys[i] >> sh;
            const unsigned cp = i == 0 ? 0xFFFFFFFFu : keys[i - 1] >> sh;
            if (cp != c)
                for (unsigned cc = cp + 1u; cc <= c; ++cc) // (cp + 1 wraps to 0 for the first leaf: the cells in front of it)
                    c_start[cc] = (unsigned short)i;
        }
        __syncthreads();
    }
    int carry = 0; // kept points of the previous chunks of 1024
    for (int base = 0; base < n || base == 0; base += 1024) {
        const int i = base + tid;
        float2 p = make_float2(0, 0);
        if (i < n)
            p = pts[i];
        int cnt = 0;
        if (do_filter && lc >= 0) {
            if (i < n) {
                const unsigned ck = (leaf_keys_all + (size_t)f * cap)[i] >> sh;
                const int ix = (int)cf_squeeze(ck), iy = (int)cf_squeeze(ck >> 1), side = 1 << lc;
                for (int dy = -1; dy <= 1; ++dy) {
                    const int cy = iy + dy;
                    if (cy < 0 || cy >= side)
                        continue;
                    for (int dx = -1; dx <= 1; ++dx) {
                        const int cx = ix + dx;
                        if (cx < 0 || cx >= side)
                            continue;
                        const unsigned m = cf_spread((unsigned)cx) | (cf_spread((unsigned)cy) << 1);
                        const int j1 = c_start[m + 1];
                        for (int j = c_start[m]; j < j1; ++j) {
                            const float2 t = c_p[j];
                            const float ddx = __fadd_rn(p.x, -t.x), ddy = __fadd_rn(p.y, -t.y);
                            cnt += __fadd_rn(__fmul_rn(ddx, ddx), __fmul_rn(ddy, ddy)) <= r2;
                        }
                    }
                }
            }
        } else if (do_filter) {
            for (int tb = 0; tb < n; tb += 2048) {
                const int tn = min(2048, n - tb);
                __syncthreads();
                for (int j = tid; j < tn; j += 1024)
                    s_p[j] = pts[tb + j];
                __syncthreads();
                for (int j = 0; j < tn; ++j) {
                    const float2 t = s_p[j];
                    const float dx = __fadd_rn(p.x, -t.x), dy = __fadd_rn(p.y, -t.y);
                    cnt += __fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)) <= r2;
                }
            }
        }
        const int keep = (i < n) && (!do_filter || cnt > min_points);
        int kept_here; // (wave scan + 16 wave totals: 3 barriers instead of the 20 of a Hillis-Steele scan through LDS: round 6)
        const int at = cf_block_excl_scan(keep, s_scan, &kept_here);
        if (keep)
            out[carry + at] = p;
        carry += kept_here;
        if (n == 0)
            break;
    }
    if (tid == 0) {
        hdrs[f].n_out = carry;
        out_counts[f] = carry;
    }
}

// staged hand-over from the extraction: bounding box + count -> octree root and depth (what cf_cast_bbox_kernel's last thread does)
__global__ __launch_bounds__(256) void cf_header_from_bbox_kernel(const CfBBox *__restrict__ bbox, int n_frames, float max_size,
                                                                  CfHeader *__restrict__ hdrs)
{
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f < n_frames) {
        const CfBBox b = bbox[f];
        hdrs[f] = cf_make_header(b.mnx, b.mny, b.mxx, b.mxy, max_size, b.n);
    }
}

float sfe_cf_max_size(float resolution)
{
    char buf[64];
    snprintf(buf, sizeof buf, "%f", (double)resolution);
    return strtof(buf, nullptr);
}

extern "C" int sfe_cloud_filter_batch_dev(sfe_ctx *ctx, const double *d_pts, const int32_t *d_counts, int n_frames,
                                          int64_t cap, float resolution, double radius, int min_points, float *d_out,
                                          int32_t *d_out_counts)
{
    if (int rc = sfe_use(ctx))
        return rc;
    SFE_ARG(ctx, n_frames >= 0 && cap >= 0 && (n_frames == 0 || cap == 0 || (d_pts && d_counts && d_out && d_out_counts)));
    if (n_frames == 0)
        return 0;
    if (resolution > 0.0f && cap > CF_MAX_CAP)
        return sfe_set_err(ctx, SFE_ERR_ARG, "sfe_cloud_filter_batch_dev: cap %lld exceeds %d points per frame",
                           (long long)cap, CF_MAX_CAP);
    const size_t per = (size_t)std::max<int64_t>(cap, 1);
    ctx->staged_frames = -1; // (the staging slots are rewritten)
    float2 *d_p32 = (float2 *)sfe_scratch(ctx, CF_SLOT_P32, sizeof(float2) * per * (size_t)n_frames);
    CfHeader *d_hdr = (CfHeader *)sfe_scratch(ctx, CF_SLOT_HDR, sizeof(CfHeader) * (size_t)n_frames);
    if (!d_p32 || !d_hdr)
        return SFE_ERR_HIP;
    hipLaunchKernelGGL(cf_cast_bbox_kernel, dim3(n_frames), dim3(1024), 0, ctx->stream, d_pts, d_counts, (long long)cap,
                       sfe_cf_max_size(resolution), d_p32, d_hdr);
    return sfe_cf_run_staged(ctx, n_frames, cap, resolution, radius, min_points, d_out, d_out_counts);
}

extern "C" int sfe_cloud_filter_staged_dev(sfe_ctx *ctx, int n_frames, int64_t cap, float resolution, double radius, int min_points,
                                           float *d_out, int32_t *d_out_counts)
{
    if (int rc = sfe_use(ctx))
        return rc;
    SFE_ARG(ctx, n_frames >= 0 && cap > 0 && (n_frames == 0 || (d_out && d_out_counts)));
    if (ctx->staged_frames != n_frames || ctx->staged_cap != (long long)cap)
        return sfe_set_err(ctx, SFE_ERR_ARG, "sfe_cloud_filter_staged_dev: no staged clouds of %d frames x %lld points on this context "
                           "(sfe_extract_points_bits_staged_dev must precede it; found %d x %lld)", n_frames, (long long)cap,
                           ctx->staged_frames, ctx->staged_cap);
    if (n_frames == 0)
        return 0;
    CfBBox *d_bbox = (CfBBox *)sfe_scratch(ctx, CF_SLOT_BBOX, sizeof(CfBBox) * (size_t)n_frames);
    CfHeader *d_hdr = (CfHeader *)sfe_scratch(ctx, CF_SLOT_HDR, sizeof(CfHeader) * (size_t)n_frames);
    if (!d_bbox || !d_hdr)
        return SFE_ERR_HIP;
    hipLaunchKernelGGL(cf_header_from_bbox_kernel, dim3((unsigned)((n_frames + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const CfBBox *)d_bbox, n_frames, sfe_cf_max_size(resolution), d_hdr);
    return sfe_cf_run_staged(ctx, n_frames, cap, resolution, radius, min_points, d_out, d_out_counts);
}

int sfe_cf_run_staged(sfe_ctx *ctx, int n_frames, int64_t cap, float resolution, double radius, int min_points,
                      float *d_out, int32_t *d_out_counts)
{
    const bool do_ds = resolution > 0.0f;            // feature_extraction.py:241
    const bool do_filter = min_points > 1;           // feature_extraction.py:245
    if (do_ds && cap > CF_MAX_CAP)
        return sfe_set_err(ctx, SFE_ERR_ARG, "resident cloud filter: cap %lld exceeds %d points per frame",
                           (long long)cap, CF_MAX_CAP);
    const size_t per = (size_t)std::max<int64_t>(cap, 1);
    float2 *d_p32 = (float2 *)sfe_scratch(ctx, CF_SLOT_P32, sizeof(float2) * per * (size_t)n_frames);
    float2 *d_ds = (float2 *)sfe_scratch(ctx, 26, sizeof(float2) * per * (size_t)n_frames);
    CfHeader *d_hdr = (CfHeader *)sfe_scratch(ctx, CF_SLOT_HDR, sizeof(CfHeader) * (size_t)n_frames);
    if (!d_p32 || !d_ds || !d_hdr)
        return SFE_ERR_HIP;
    const float2 *stage = d_p32;
    unsigned *d_lkeys = nullptr; // leaf keys of the downsampled clouds (radius filter fast path)
    if (do_ds) {
        // 48 key bits + 16 index bits: a frame whose tree is deeper than 24 levels reports count -1
        size_t n2 = 2;
        while (n2 < (size_t)cap)
            n2 <<= 1;
        int *d_seg = (int *)sfe_scratch(ctx, 28, sizeof(int) * (per + 1) * (size_t)n_frames);
        d_lkeys = (unsigned *)sfe_scratch(ctx, 31, sizeof(unsigned) * per * (size_t)n_frames);
        float2 *d_spts = (float2 *)sfe_scratch(ctx, 40, sizeof(float2) * per * (size_t)n_frames);
        if (!d_seg || !d_lkeys || !d_spts)
            return SFE_ERR_HIP;
        {
            // Per FRAME (round 6): frames of <= CF_SORT_CAP points sort in LDS whatever the batch's capacity is -- trees of <= 8
            // levels (every frame of a sonar fan at 0.5 m) by the radix sort of the indices, deeper ones by the bitonic sort with
            // 64-bit keys; a frame with more points (a capacity beyond CF_SORT_CAP allows them) marks itself and is sorted in HBM
            // scratch by the last launch.  (Rounds 2-5 chose by the batch's capacity: one dense ping sent all frames to HBM.)
            const size_t n2l = std::min<size_t>(n2, CF_SORT_CAP); // LDS slots of the bitonic launch (a power of two)
            const size_t n2r = cap <= CF_SORT_CAP ? n2 : (size_t)CF_LDS_PTS_BIG; // ... and of the radix launch
            int *d_marked = nullptr; // capacities beyond the LDS sort: [n_frames] frames that want the HBM sort + [1] how many
            if (cap > CF_SORT_CAP) {
                d_marked = (int *)sfe_scratch(ctx, 52, sizeof(int) * ((size_t)n_frames + 1));
                if (!d_marked)
                    return SFE_ERR_HIP;
                SFE_HIP(ctx, hipMemsetAsync(d_marked + n_frames, 0, sizeof(int), ctx->stream));
            }
            {
                // indices + keys + counters for the sort, then (same bytes) every point of the frame in sorted order.
                // (The LDS is sized by the CAPACITY: 128 KB = one frame per CU at 16 384 points, 64 KB = two per CU at 8 192.
                // Round 4 tried a launch per size class -- frames of <= 8 192 points in a 64 KB launch of their own, the
                // others behind it, a workgroup of the other class returning at once: 0.29 -> 0.42 ms per 512 frames,
                // because dispatching 512 workgroups of 1024 threads costs ~0.12 ms even when they do nothing.  A caller
                // whose pings are small passes a smaller capacity instead: chained.SessionBatch sizes it from its warm-up.)
                // (digit counters: one column of 16 per sorting thread -- 1024 of them, 512 at capacities of <= 8192 points so
                // that indices + keys + counters stay inside the 64 KB that let two frames share a CU)
                const int sort_cols = n2r <= 8192 ? 512 : 1024;
                const size_t n_cnt = CF_RDX_DIGITS * (size_t)sort_cols;
                const bool big = n2r > CF_SORT_CAP;
                // (the sorted points take the same bytes when they fit, else their HBM scratch: the kernel looks at lds_bytes)
                const size_t rdx_smem = big ? 3 * 2 * n2r + 2 * (n_cnt + n_cnt / 16)
                                            : std::max<size_t>(3 * 2 * n2r + 2 * (n_cnt + n_cnt / 16), sizeof(float2) * n2r);
                auto rk = big ? cf_downsample_radix_kernel<CF_RDX_CHUNK_BIG> : cf_downsample_radix_kernel<CF_RDX_CHUNK>;
                SFE_HIP(ctx, hipFuncSetAttribute((const void *)rk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rdx_smem));
                hipLaunchKernelGGL(rk, dim3(n_frames), dim3(1024), rdx_smem, ctx->stream, d_p32,
                                   (long long)cap, d_hdr, d_ds, d_seg, (int)n2r, d_lkeys, d_spts, (int)rdx_smem, sort_cols, d_marked,
                                   d_marked ? d_marked + n_frames : nullptr);
            }
            SFE_HIP(ctx, hipFuncSetAttribute((const void *)cf_downsample_kernel<true, unsigned long long>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)(8 * n2l)));
            hipLaunchKernelGGL((cf_downsample_kernel<true, unsigned long long>), dim3(n_frames), dim3(1024), 8 * n2l,
                               ctx->stream, d_p32, (long long)cap, d_hdr, d_ds, d_seg, (unsigned long long *)nullptr, (long long)n2l, 1,
                               (unsigned *)nullptr, d_marked, d_marked ? d_marked + n_frames : nullptr, (const int *)nullptr);
            if (cap > CF_SORT_CAP) { // the listed frames (> CF_SORT_CAP points): a few workgroups stride over the list
                const int gw = std::min(n_frames, 256); // (one per CU: a batch whose frames ALL want this sort -- 2048 x 1024 pings -- keeps its parallelism)
                unsigned long long *d_gk = (unsigned long long *)sfe_scratch(ctx, 29, 8 * n2 * (size_t)gw);
                if (!d_gk)
                    return SFE_ERR_HIP;
                hipLaunchKernelGGL((cf_downsample_kernel<false, unsigned long long>), dim3(gw), dim3(1024), 0,
                                   ctx->stream, d_p32, (long long)cap, d_hdr, d_ds, d_seg, d_gk, (long long)n2, 1,
                                   (unsigned *)nullptr, (int *)nullptr, d_marked + n_frames, (const int *)d_marked);
            }
        }
        stage = d_ds;
    }
    const size_t cell_smem = sizeof(float2) * CF_CELL_PTS + sizeof(unsigned short) * ((1 << (2 * CF_CELL_LEVELS)) + 2);
    SFE_HIP(ctx, hipFuncSetAttribute((const void *)cf_radius_filter_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)cell_smem));
    hipLaunchKernelGGL(cf_radius_filter_kernel, dim3(n_frames), dim3(1024),
                       d_lkeys ? cell_smem : sizeof(float2) * 2048, ctx->stream, stage,
                       (long long)cap, d_hdr, (float)(radius * radius), min_points, d_out, d_out_counts, do_filter ? 1 : 0,
                       (const unsigned *)d_lkeys);
    SFE_LAUNCH_CHECK(ctx);
    return 0;
}
