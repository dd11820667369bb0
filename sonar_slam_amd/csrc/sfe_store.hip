// Device-resident keyframe clouds: the feature clouds never leave HBM between the feature extractor
// and the scan matcher.
//
//   reference flow (host numpy, every cloud crosses the process boundary as a PointCloud2):
//     feature_extraction.py:175-193  publish_features           cloud -> xyz32 bytes
//     slam_ros.py:169-170            SLAM_callback              bytes -> [x, -z]
//     slam_objects.py:178-198        Keyframe.transform_points  points.dot(R.T) + t      (float32)
//     slam.py:229-292                SLAM.get_points            transform, concatenate, pcl.downsample
//     slam.py:294-323                SLAM.compute_icp           pcl.ICP.compute(source, target, guess)
//     slam.py:389-424                SLAM.get_overlap           transform, pcl.match, count ids != -1
//
//   here: a store = one pool of float2 points + a slot table {offset, count} per cloud, both in HBM.  Clouds are
//   appended by kernels whose sizes come from device memory (the per-frame counts the cloud filter leaves there), so
//   the pool's fill level lives on the device too; the host mirrors the slot table lazily (one small copy when it
//   needs sizes: the SLAM node needs them anyway for its ssm_min_points test, slam.py:745).  get_points, the scan
//   match and the overlap count name clouds by handle (= slot index); the cloud's bytes are copied to the host only
//   when somebody asks for them (sfe_cloud_store_read: the PointCloud2 for rviz / the mapping node).
//
// Slots are handed out in order and freed in stack order (sfe_cloud_store_truncate): keyframes stay for the whole
// session (loop closures read old clouds), the target clouds get_points builds are dropped after the scan match.
#include "sfe_cloudfilter.h"

#include <algorithm>
#include <cstring>
#include <vector>

struct sfe_cloud_store {
    sfe_ctx *ctx = nullptr;
    int64_t capacity = 0; // points
    int32_t max_clouds = 0;
    float2 *d_pool = nullptr;
    int64_t *d_off = nullptr; // [max_clouds] first point of the cloud in the pool
    int32_t *d_cnt = nullptr; // [max_clouds] points (< 0: the producer failed, see SFE_STORE_*)
    int64_t *d_top = nullptr; // [1] points in use
    // loop-closure search (NSSM, slam.py:839-1001): a key (the keyframe a point came from) per pool point, filled for the
    // clouds the keyed entry points produce; a selection byte per point of the cloud the last fov_select looked at
    int32_t *d_key = nullptr; // [capacity], allocated on first use
    uint8_t *d_sel = nullptr;
    size_t sel_cap = 0;
    int32_t sel_handle = -1, sel_count = 0; // the cloud the selection was made for, and its size then (a handle alone can
                                            // be reused by a larger cloud after a truncate)
    std::vector<uint8_t> keyed;             // per slot: built by a keyed entry point (the key pool holds ITS keys)
    // host mirror of the slot table: entries < n_synced are valid
    std::vector<int64_t> stamp, off;
    std::vector<int32_t> cnt;
    int32_t n_slots = 0, n_synced = 0;
};

// ---------------------------------------------------------------------------------------------------------------------
// append: n_frames clouds in the strided layout of the resident filters ([f][cap] float2 + counts[f]) -> packed behind
// the pool's fill level.  Two kernels, no atomics: every block of the first derives its offset from the counts in
// front of it and the fill level (read-only here); the second writes the slot table and moves the fill level.
// flags bit 0: store (x, -y) -- what the SLAM node makes of the feature message, slam_ros.py:170 `np.c_[x, -1 * z]`.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int store_clamp_count(int c, long long cap) { return (int)min((long long)max(c, 0), cap); }

__global__ __launch_bounds__(256) void store_append_kernel(const float2 *__restrict__ clouds,
                                                           const int32_t *__restrict__ counts, int n_frames,
                                                           long long cap, int flags, float2 *__restrict__ pool,
                                                           long long capacity, const int64_t *__restrict__ top)
{
    __shared__ long long s_part[4];
    const int f = blockIdx.x, tid = threadIdx.x;
    long long pre = 0;
    for (int g = tid; g < f; g += 256)
        pre += store_clamp_count(counts[g], cap);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
        pre += __shfl_down(pre, d);
    if ((tid & 63) == 0)
        s_part[tid >> 6] = pre;
    __syncthreads();
    const long long base = *top + s_part[0] + s_part[1] + s_part[2] + s_part[3];
    const int n = store_clamp_count(counts[f], cap);
    if (base + n > capacity)
        return; // the commit kernel marks the slot
    const float2 *src = clouds + (size_t)f * cap;
    float2 *dst = pool + base;
    const bool neg = flags & SFE_STORE_NEGATE_Y;
    for (int i = tid; i < n; i += 256) {
        float2 p = src[i];
        if (neg)
            p.y = -p.y;
        dst[i] = p;
    }
}

#define SFE_STORE_OVERFLOW (-3)
__global__ __launch_bounds__(1024) void store_commit_kernel(const int32_t *__restrict__ counts, int n_frames, long long cap,
                                                            long long capacity, int64_t *__restrict__ top,
                                                            int64_t *__restrict__ off, int32_t *__restrict__ cnt)
{
    __shared__ long long s_wave[16];
    __shared__ long long s_carry;
    __shared__ unsigned long long s_fit_end; // where the clouds that fitted end (they are a prefix of the batch)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) {
        s_carry = *top;
        s_fit_end = ~0ull;
    }
    __syncthreads();
    for (int b = 0; b < n_frames; b += 1024) {
        const int f = b + tid;
        const int raw = f < n_frames ? counts[f] : 0;
        const long long n = f < n_frames ? store_clamp_count(raw, cap) : 0;
        long long incl = n;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const long long t = __shfl_up(incl, d);
            if (lane >= d)
                incl += t;
        }
        if (lane == 63)
            s_wave[wave] = incl;
        __syncthreads();
        long long wpre = 0;
        for (int w = 0; w < wave; ++w)
            wpre += s_wave[w];
        const long long carry = s_carry;
        const long long start = carry + wpre + incl - n;
        if (f < n_frames) {
            const bool fits = start + n <= capacity;
            off[f] = fits ? start : -1;
            cnt[f] = fits ? (raw < 0 ? raw : (int)n) : SFE_STORE_OVERFLOW;
            if (!fits)
                atomicMin(&s_fit_end, (unsigned long long)start);
        }
        __syncthreads();
        if (tid == 1023)
            s_carry = carry + wpre + incl;
        __syncthreads();
    }
    // Clouds are laid out in order, so once one does not fit neither does any behind it (their starts lie beyond the
    // capacity): the fill level stops where the first of them would have begun.  A slot that overflowed (-3, offset -1)
    // therefore owns no pool space, and dropping it (sfe_cloud_store_truncate) leaves the level where it is.
    if (tid == 0)
        *top = s_fit_end != ~0ull ? (long long)s_fit_end : s_carry;
}

// ---------------------------------------------------------------------------------------------------------------------
// get_points (slam.py:229-292 with a reference frame): per job the clouds of <= m keyframes, each moved by its own
// transform exactly like Keyframe.transform_points (slam_objects.py:178-198), concatenated in frame order into the
// staging layout of the resident cloud filters, with the bounding box / octree header pcl.downsample starts from.
//
// points.dot(T[:2, :2].T) + T[:2, 2] with T = pose.matrix().astype(np.float32).  What numpy computes depends on the
// dtype of `points`:
//   F64 (default): the SLAM node's keyframe clouds are float64 arrays holding float32 values
//     (ros_numpy pointcloud2_to_xyz_array -> get_xyz_points(dtype=np.float), slam_ros.py:169-170), so numpy promotes
//     the product to float64: both products are exact in double, their sum is rounded once (FMA or not: the same),
//     the translation is added in double, and the cloud is rounded to float32 at the pybind boundary of
//     pcl.downsample / ICP.compute / match (pcl.cpp:10-16).  Independent of the BLAS underneath.
//   F32 (flag SFE_STORE_F32_POINTS): float32 points go through sgemm, whose x86 kernels accumulate over k with fused
//     multiply-adds: fma(p1, r1, fl(p0 * r0)), then a float add.
// Both pinned by tests/golden/transform_points.npz: the reference's own function on this image's numpy.
// ---------------------------------------------------------------------------------------------------------------------
template <bool F64>
__device__ __forceinline__ float2 store_transform(float2 p, float r00, float r01, float tx, float r10, float r11, float ty)
{
    if (F64) {
        const double x = __dadd_rn(__dadd_rn(__dmul_rn((double)p.x, (double)r00), __dmul_rn((double)p.y, (double)r01)), (double)tx);
        const double y = __dadd_rn(__dadd_rn(__dmul_rn((double)p.x, (double)r10), __dmul_rn((double)p.y, (double)r11)), (double)ty);
        return make_float2((float)x, (float)y);
    }
    const float x = __fadd_rn(__fmaf_rn(p.y, r01, __fmul_rn(p.x, r00)), tx);
    const float y = __fadd_rn(__fmaf_rn(p.y, r11, __fmul_rn(p.x, r10)), ty);
    return make_float2(x, y);
}

template <bool F64>
__global__ __launch_bounds__(1024) void store_gather_kernel(const float2 *__restrict__ pool, const int64_t *__restrict__ off,
                                                            const int32_t *__restrict__ cnt,
                                                            const int32_t *__restrict__ handles, const float *__restrict__ T6,
                                                            int m, long long cap, float max_size, float2 *__restrict__ p32,
                                                            CfHeader *__restrict__ hdrs)
{
    __shared__ float s_mn[2][16], s_mx[2][16];
    const int j = blockIdx.x, tid = threadIdx.x;
    float2 *dst = p32 + (size_t)j * cap;
    float mnx = INFINITY, mny = INFINITY, mxx = -INFINITY, mxy = -INFINITY;
    long long pos = 0;
    for (int k = 0; k < m; ++k) {
        const int h = handles[(size_t)j * m + k];
        if (h < 0)
            continue;
        const int n = max(cnt[h], 0);
        const float2 *src = pool + off[h];
        const float *T = T6 + ((size_t)j * m + k) * 6;
        const float t0 = T[0], t1 = T[1], t2 = T[2], t3 = T[3], t4 = T[4], t5 = T[5]; // {r00, r01, tx, r10, r11, ty}
        for (int i = tid; i < n && pos + i < cap; i += 1024) {
            const float2 q = store_transform<F64>(src[i], t0, t1, t2, t3, t4, t5);
            dst[pos + i] = q;
            mnx = fminf(mnx, q.x);
            mxx = fmaxf(mxx, q.x);
            mny = fminf(mny, q.y);
            mxy = fmaxf(mxy, q.y);
        }
        pos += n;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        mnx = fminf(mnx, __shfl_down(mnx, d));
        mxx = fmaxf(mxx, __shfl_down(mxx, d));
        mny = fminf(mny, __shfl_down(mny, d));
        mxy = fmaxf(mxy, __shfl_down(mxy, d));
    }
    if ((tid & 63) == 0) {
        s_mn[0][tid >> 6] = mnx;
        s_mn[1][tid >> 6] = mny;
        s_mx[0][tid >> 6] = mxx;
        s_mx[1][tid >> 6] = mxy;
    }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 16; ++w) {
            mnx = fminf(mnx, s_mn[0][w]);
            mny = fminf(mny, s_mn[1][w]);
            mxx = fmaxf(mxx, s_mx[0][w]);
            mxy = fmaxf(mxy, s_mx[1][w]);
        }
        hdrs[j] = cf_make_header(mnx, mny, mxx, mxy, max_size, (int)min(pos, cap));
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// get_overlap (slam.py:389-424): transform the source by the estimated pose, pcl.match(target, source, 1, max_dist),
// count the matched points.  One workgroup per job; the same float distance and the same radius test as match_kernel
// (sfe_icp.hip): a query counts iff its nearest target lies within the radius, i.e. iff ANY target does.
// ---------------------------------------------------------------------------------------------------------------------
template <bool F64>
__global__ __launch_bounds__(256) void store_overlap_kernel(const float2 *__restrict__ pool, const int64_t *__restrict__ off,
                                                            const int32_t *__restrict__ cnt, const int32_t *__restrict__ pairs,
                                                            const float *__restrict__ T6, float r2,
                                                            int32_t *__restrict__ out_counts)
{
    __shared__ float2 s_ref[2048];
    __shared__ int s_total;
    const int j = blockIdx.x, tid = threadIdx.x;
    const int hs = pairs[2 * j], ht = pairs[2 * j + 1];
    const int ns = hs >= 0 ? max(cnt[hs], 0) : 0, nt = ht >= 0 ? max(cnt[ht], 0) : 0;
    const float2 *src = pool + (hs >= 0 ? off[hs] : 0), *tgt = pool + (ht >= 0 ? off[ht] : 0);
    const float *Tp = T6 + (size_t)j * 6;
    const float t0 = Tp[0], t1 = Tp[1], t2 = Tp[2], t3 = Tp[3], t4 = Tp[4], t5 = Tp[5];
    if (tid == 0)
        s_total = 0;
    int mine = 0;
    for (int b = 0; b < ns; b += 256) {
        const int i = b + tid;
        float2 q = make_float2(0.0f, 0.0f);
        if (i < ns)
            q = store_transform<F64>(src[i], t0, t1, t2, t3, t4, t5);
        float best = INFINITY;
        for (int tb = 0; tb < nt; tb += 2048) {
            const int tn = min(2048, nt - tb);
            __syncthreads();
            for (int t = tid; t < tn; t += 256)
                s_ref[t] = tgt[tb + t];
            __syncthreads();
            for (int t = 0; t < tn; ++t) {
                const float2 r = s_ref[t];
                const float dx = __fadd_rn(q.x, -r.x), dy = __fadd_rn(q.y, -r.y);
                const float d = __fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy));
                if (d < best)
                    best = d;
            }
        }
        mine += (i < ns) && (best <= r2);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
        mine += __shfl_down(mine, d);
    __syncthreads();
    if ((tid & 63) == 0)
        atomicAdd(&s_total, mine);
    __syncthreads();
    if (tid == 0)
        out_counts[j] = s_total;
}


// ---------------------------------------------------------------------------------------------------------------------
// Loop-closure search over the store (hot loop #4: slam.py:839-1001, 1003-1132).
//
//   get_points(target_frames, None, return_keys=True)   slam.py:873 -> :229-292 -> pcl.downsample(points, keys, res)
//        every keyframe older than k - min_st_sep, each under its own pose, concatenated with its key, descriptor
//        overload of the downsample (pcl.cpp:143-159): the medoid of a leaf brings its key along.  Unbounded in the
//        length of the session, so this path has no 65 536-point limit: the concatenation lives in scratch and goes
//        through the rank sort of sfe_downsample.hip.
//   field-of-view gate                                   slam.py:875-899
//   np.unique(keys[sel], return_counts=True)             slam.py:902-904
//   get_overlap(..., return_indices=True) + unique       slam.py:977-985
// ---------------------------------------------------------------------------------------------------------------------
struct StoreSeg { // one keyframe of a concatenation
    long long src, dst; // first point in the pool / in the concatenation
    int n, key;
    float T[6];
};

template <bool F64>
__global__ __launch_bounds__(256) void store_concat_kernel(const float2 *__restrict__ pool, const StoreSeg *__restrict__ seg,
                                                           int n_seg, long long n_total, float2 *__restrict__ out,
                                                           int32_t *__restrict__ out_key)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_total)
        return;
    int lo = 0, hi = n_seg - 1; // last segment whose dst <= i (segments of zero points share a dst: the last one wins,
    while (lo < hi) {           //  and it is the one that holds the point)
        const int mid = (lo + hi + 1) >> 1;
        if (seg[mid].dst <= i)
            lo = mid;
        else
            hi = mid - 1;
    }
    const StoreSeg sg = seg[lo];
    out[i] = store_transform<F64>(pool[sg.src + (i - sg.dst)], sg.T[0], sg.T[1], sg.T[2], sg.T[3], sg.T[4], sg.T[5]);
    if (out_key)
        out_key[i] = sg.key;
}

// keys of the medoids: key[s] = cat_key[idx[s]]
__global__ __launch_bounds__(256) void store_pick_keys_kernel(const int32_t *__restrict__ cat_key, const int32_t *__restrict__ idx,
                                                              const int32_t *__restrict__ n, int32_t *__restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < *n)
        out[i] = cat_key[idx[i]];
}

// the keys of a freshly committed slot go next to its points (the slot's offset is known on the device only)
__global__ __launch_bounds__(256) void store_commit_keys_kernel(const int32_t *__restrict__ keys, const int64_t *__restrict__ off,
                                                                const int32_t *__restrict__ cnt, int slot,
                                                                int32_t *__restrict__ key_pool)
{
    const long long o = off[slot];
    const int n = cnt[slot];
    if (o < 0)
        return;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
        key_pool[o + i] = keys[i];
}

// Field-of-view gate (slam.py:875-899) for every point of one cloud against n_src source frames:
//     local = transform_points(target_points, pose.inverse())            float32 points: sgemm arithmetic
//     ranges = np.linalg.norm(local, axis=1); bearings = np.arctan2(local[:, 1], local[:, 0])      float32
//     sel |= (ranges < range_bound) & (abs(bearings) < bearing_bound)                              bounds: float64
// The range is exact (float32 multiply, add, correctly rounded sqrt).  numpy's float32 arctan2 is libm's atan2f, whose
// last bit this kernel does not try to reproduce: it evaluates atan2 in double and decides only when |bearing| is
// further from the bound than a float32 result can be from the true angle; a point that no frame selects for certain
// and some frame leaves undecided is counted in *n_ambiguous, and the host then takes the numpy path for this cloud
// (sfe_cloud_store_set_selection).  In practice: never (a bound hit to 1e-6 relative).
struct FovFrame {
    float T[6];
    double range_bound, bearing_bound;
};
__global__ __launch_bounds__(256) void store_fov_kernel(const float2 *__restrict__ pts, const int32_t *__restrict__ keys, int n,
                                                        const FovFrame *__restrict__ frames, int n_frames, int n_keys,
                                                        uint8_t *__restrict__ sel, int32_t *__restrict__ hist,
                                                        int32_t *__restrict__ counters /* [0] selected, [1] ambiguous */)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    bool yes = false, maybe = false;
    if (i < n) {
        const float2 p = pts[i];
        for (int f = 0; f < n_frames; ++f) {
            const FovFrame fr = frames[f];
            const float2 q = store_transform<false>(p, fr.T[0], fr.T[1], fr.T[2], fr.T[3], fr.T[4], fr.T[5]);
            // sqrtf, not __fsqrt_rn: the intrinsic lowers to a bare v_sqrt_f32 (1 ulp), sqrtf to the correctly rounded
            // sequence numpy's float32 sqrt gives (a point 0.95 ulp outside its range bound showed the difference)
            const float range = sqrtf(__fadd_rn(__fmul_rn(q.x, q.x), __fmul_rn(q.y, q.y)));
            if (!((double)range < fr.range_bound))
                continue;
            const double a = fabs(atan2((double)q.y, (double)q.x));
            const double margin = 1e-6 * a + 1e-30;
            if (a < fr.bearing_bound - margin)
                yes = true;
            else if (!(a > fr.bearing_bound + margin))
                maybe = true;
        }
        sel[i] = yes ? 1 : 0;
        if (yes) {
            const int k = keys[i];
            if (k >= 0 && k < n_keys)
                atomicAdd(&hist[k], 1);
        }
    }
    const unsigned long long by = __ballot(yes), bm = __ballot(maybe && !yes);
    if ((threadIdx.x & 63) == 0) {
        if (by)
            atomicAdd(&counters[0], __popcll(by));
        if (bm)
            atomicAdd(&counters[1], __popcll(bm));
    }
}

// selected points of a cloud, order kept, with their keys: one workgroup, chunks of 1024 with a running base
__global__ __launch_bounds__(1024) void store_compact_kernel(const float2 *__restrict__ pts, const int32_t *__restrict__ keys,
                                                             const uint8_t *__restrict__ sel, int n, float2 *__restrict__ out,
                                                             int32_t *__restrict__ out_key, int32_t *__restrict__ n_out)
{
    __shared__ int s_wave[16];
    __shared__ int s_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0)
        s_base = 0;
    __syncthreads();
    for (int b = 0; b < n; b += 1024) {
        const int i = b + tid;
        const bool on = i < n && sel[i];
        const unsigned long long m = __ballot(on);
        if (lane == 0)
            s_wave[wave] = __popcll(m);
        __syncthreads();
        int pre = s_base;
        for (int w = 0; w < wave; ++w)
            pre += s_wave[w];
        if (on) {
            const int pos = pre + __popcll(m & ((1ull << lane) - 1ull));
            out[pos] = pts[i];
            out_key[pos] = keys[i];
        }
        __syncthreads();
        if (tid == 0) {
            int t = 0;
            for (int w = 0; w < 16; ++w)
                t += s_wave[w];
            s_base += t;
        }
        __syncthreads();
    }
    if (tid == 0)
        *n_out = s_base;
}

// get_overlap(source under a pose, target, return_indices=True) + the keys of the matched targets (slam.py:977-985):
// pcl.match's neighbour = nearest target, lowest index among equals, within the radius; hist[key of it] += 1.
template <bool F64>
__global__ __launch_bounds__(256) void store_match_keys_kernel(const float2 *__restrict__ src, int ns, const float2 *__restrict__ tgt,
                                                               const int32_t *__restrict__ tkey, int nt, const float *__restrict__ T6,
                                                               float r2, int n_keys, int32_t *__restrict__ hist,
                                                               int32_t *__restrict__ overlap)
{
    __shared__ float2 s_ref[2048];
    const int i = blockIdx.x * 256 + threadIdx.x;
    float2 q = make_float2(0.0f, 0.0f);
    if (i < ns)
        q = store_transform<F64>(src[i], T6[0], T6[1], T6[2], T6[3], T6[4], T6[5]);
    float best = INFINITY;
    int bi = -1;
    for (int tb = 0; tb < nt; tb += 2048) {
        const int tn = min(2048, nt - tb);
        __syncthreads();
        for (int t = threadIdx.x; t < tn; t += 256)
            s_ref[t] = tgt[tb + t];
        __syncthreads();
        for (int t = 0; t < tn; ++t) {
            const float2 r = s_ref[t];
            const float dx = __fadd_rn(q.x, -r.x), dy = __fadd_rn(q.y, -r.y);
            const float d = __fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy));
            if (d < best) {
                best = d;
                bi = tb + t;
            }
        }
    }
    const bool hit = i < ns && bi >= 0 && best <= r2;
    if (hit) {
        const int k = tkey[bi];
        if (k >= 0 && k < n_keys)
            atomicAdd(&hist[k], 1);
    }
    const unsigned long long m = __ballot(hit);
    if ((threadIdx.x & 63) == 0 && m)
        atomicAdd(overlap, __popcll(m));
}

// fill level := first point of the first dropped slot that holds pool space (stack-order release of slots
// [slot, n_slots)); a slot without an offset never held any, and if none of them did the level stays
__global__ void store_rewind_kernel(int64_t *__restrict__ top, const int64_t *__restrict__ off, int slot, int n_slots)
{
    for (int i = slot; i < n_slots; ++i) {
        const long long o = off[i];
        if (o >= 0) {
            *top = o;
            return;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
static int store_sync_meta(sfe_cloud_store *s)
{
    sfe_ctx *ctx = s->ctx;
    if (s->n_synced >= s->n_slots)
        return 0;
    const int a = s->n_synced, n = s->n_slots - a;
    SFE_HIP(ctx, hipMemcpyAsync(s->off.data() + a, s->d_off + a, sizeof(int64_t) * (size_t)n, hipMemcpyDeviceToHost,
                                ctx->stream));
    SFE_HIP(ctx, hipMemcpyAsync(s->cnt.data() + a, s->d_cnt + a, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost,
                                ctx->stream));
    SFE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    s->n_synced = s->n_slots;
    return 0;
}

int sfe_store_append_dev(sfe_cloud_store *s, const int64_t *stamps, const float *d_clouds, const int32_t *d_counts,
                            int n_frames, int64_t cap, int flags, int32_t *handles_out)
{
    sfe_ctx *ctx = s->ctx;
    if ((int64_t)s->n_slots + n_frames > s->max_clouds)
        return sfe_set_err(ctx, SFE_ERR_CAP, "cloud store: %d slots in use, %d more do not fit its %d", s->n_slots, n_frames,
                           s->max_clouds);
    const int first = s->n_slots;
    hipLaunchKernelGGL(store_append_kernel, dim3(n_frames), dim3(256), 0, ctx->stream, (const float2 *)d_clouds, d_counts,
                       n_frames, (long long)cap, flags, s->d_pool, (long long)s->capacity, (const int64_t *)s->d_top);
    hipLaunchKernelGGL(store_commit_kernel, dim3(1), dim3(1024), 0, ctx->stream, d_counts, n_frames, (long long)cap,
                       (long long)s->capacity, s->d_top, s->d_off + first, s->d_cnt + first);
    SFE_LAUNCH_CHECK(ctx);
    for (int f = 0; f < n_frames; ++f) {
        s->stamp[first + f] = stamps ? stamps[f] : 0;
        if (handles_out)
            handles_out[f] = first + f;
    }
    s->n_slots += n_frames;
    return 0;
}

// the count the commit kernel wrote for `handle` (>= 0 points, < 0: SFE_STORE_*), copied in stream order to a pinned
// location the caller reads after its own synchronisation (sfe_feature_extract_ping_store)
int sfe_store_slot_count_async(sfe_cloud_store *s, int32_t handle, int32_t *h_pinned)
{
    sfe_ctx *ctx = s->ctx;
    SFE_HIP(ctx, hipMemcpyAsync(h_pinned, s->d_cnt + handle, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    return 0;
}

sfe_ctx *sfe_store_ctx(sfe_cloud_store *s) { return s ? s->ctx : nullptr; }


int sfe_store_view(sfe_cloud_store *s, SfeStoreView *v)
{
    if (int rc = store_sync_meta(s))
        return rc;
    v->d_pool = s->d_pool;
    v->d_off = s->d_off;
    v->d_cnt = s->d_cnt;
    v->off = s->off.data();
    v->cnt = s->cnt.data();
    v->n_slots = s->n_slots;
    return 0;
}

static int store_key_pool(sfe_cloud_store *s)
{
    if (s->d_key)
        return 0;
    if (hipMalloc((void **)&s->d_key, sizeof(int32_t) * (size_t)s->capacity) != hipSuccess) {
        s->d_key = nullptr;
        return sfe_set_err(s->ctx, SFE_ERR_HIP, "cloud store: allocating the key pool (%lld points) failed", (long long)s->capacity);
    }
    return 0;
}

// One target cloud of ANY size, optionally with keys (the descriptor overload of pcl.downsample, pcl.cpp:143-159): the
// concatenation is built in scratch, downsampled by the rank-sort chain of sfe_downsample.hip (indices come for free), and
// appended as one new slot (+ its keys).  Enqueue only.
static int store_get_points_big(sfe_cloud_store *s, const int32_t *handles, const float *T6, const int32_t *keys, int m,
                                float resolution, int flags, int64_t stamp, int32_t *handle_out)
{
    sfe_ctx *ctx = s->ctx;
    std::vector<StoreSeg> segs;
    long long total = 0;
    for (int k = 0; k < m; ++k) {
        const int h = handles[k];
        if (h < 0)
            continue;
        if (h >= s->n_slots)
            return sfe_set_err(ctx, SFE_ERR_ARG, "get_points: cloud %d named, the store holds %d", h, s->n_slots);
        if (s->cnt[h] < 0)
            return sfe_set_err(ctx, SFE_ERR_ARG, "get_points: cloud %d was not stored (count %d: -1 octree too deep, -3 pool full)",
                               h, s->cnt[h]);
        if (s->cnt[h] == 0)
            continue;
        StoreSeg sg;
        sg.src = s->off[h];
        sg.dst = total;
        sg.n = s->cnt[h];
        sg.key = keys ? keys[k] : 0;
        memcpy(sg.T, T6 + 6 * (size_t)k, sizeof sg.T);
        segs.push_back(sg);
        total += sg.n;
    }
    if (total >= (1ll << 31) - 1)
        return sfe_set_err(ctx, SFE_ERR_ARG, "get_points: %lld points in one target cloud", total);
    if (keys)
        if (int rc = store_key_pool(s))
            return rc;
    const int n = (int)total;
    const size_t cap = (size_t)std::max(n, 1);
    float2 *d_cat = (float2 *)sfe_scratch(ctx, 53, sizeof(float2) * cap);
    int32_t *d_catkey = keys ? (int32_t *)sfe_scratch(ctx, 54, sizeof(int32_t) * cap) : nullptr;
    float2 *d_out = (float2 *)sfe_scratch(ctx, 55, sizeof(float2) * cap);
    int32_t *d_oidx = (int32_t *)sfe_scratch(ctx, 56, sizeof(int32_t) * cap);
    int32_t *d_okey = keys ? (int32_t *)sfe_scratch(ctx, 57, sizeof(int32_t) * cap) : nullptr;
    SfeDsHeader *d_hdr = (SfeDsHeader *)sfe_scratch(ctx, 58, sizeof(SfeDsHeader));
    if (!d_cat || !d_out || !d_oidx || !d_hdr || (keys && (!d_catkey || !d_okey)))
        return SFE_ERR_HIP;
    const float2 *d_final = d_out;
    const int32_t *d_final_key = d_okey;
    SFE_HIP(ctx, hipMemsetAsync(d_hdr, 0, sizeof(SfeDsHeader), ctx->stream)); // (n_seg = 0: an empty target)
    if (n > 0) {
        const size_t b_seg = sizeof(StoreSeg) * segs.size();
        StoreSeg *h_seg = (StoreSeg *)sfe_pinned_begin(ctx, b_seg);
        StoreSeg *d_seg = (StoreSeg *)sfe_scratch(ctx, 59, b_seg);
        if (!h_seg || !d_seg)
            return SFE_ERR_HIP;
        memcpy(h_seg, segs.data(), b_seg);
        SFE_HIP(ctx, hipMemcpyAsync(d_seg, h_seg, b_seg, hipMemcpyHostToDevice, ctx->stream));
        if (int rc = sfe_pinned_end(ctx, ctx->stream))
            return rc;
        auto cat = (flags & SFE_STORE_F32_POINTS) ? store_concat_kernel<false> : store_concat_kernel<true>;
        hipLaunchKernelGGL(cat, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const float2 *)s->d_pool,
                           (const StoreSeg *)d_seg, (int)segs.size(), (long long)n, d_cat, d_catkey);
        SFE_LAUNCH_CHECK(ctx);
        if (resolution > 0.0f) {
            if (int rc = sfe_ds_run_dev(ctx, (const float *)d_cat, n, resolution, (float *)d_out, d_oidx, d_hdr))
                return rc;
            if (keys) {
                hipLaunchKernelGGL(store_pick_keys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                                   (const int32_t *)d_catkey, (const int32_t *)d_oidx, (const int32_t *)&d_hdr->n_seg, d_okey);
                SFE_LAUNCH_CHECK(ctx);
            }
        } else { // no downsample (SLAM.get_points never asks for this; kept for symmetry with the batched entry point)
            SFE_HIP(ctx, hipMemcpyAsync(&d_hdr->n_seg, &n, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
            d_final = d_cat;
            d_final_key = d_catkey;
        }
    }
    int32_t h_new = -1;
    if (int rc = sfe_store_append_dev(s, &stamp, (const float *)d_final, (const int32_t *)&d_hdr->n_seg, 1, (int64_t)cap, 0, &h_new))
        return rc;
    if (keys && n > 0) {
        hipLaunchKernelGGL(store_commit_keys_kernel, dim3(64), dim3(256), 0, ctx->stream, d_final_key, (const int64_t *)s->d_off,
                           (const int32_t *)s->d_cnt, (int)h_new, s->d_key);
        SFE_LAUNCH_CHECK(ctx);
    }
    if (keys && h_new >= 0) { // (an empty keyed cloud is keyed too: it has no keys to be wrong about)
        if ((int32_t)s->keyed.size() <= h_new)
            s->keyed.resize((size_t)h_new + 1, 0);
        s->keyed[(size_t)h_new] = 1;
    }
    *handle_out = h_new;
    return 0;
}

extern "C" {

int sfe_cloud_store_create(sfe_ctx *ctx, int64_t capacity_points, int32_t max_clouds, sfe_cloud_store **out)
{
    if (int rc = sfe_use(ctx))
        return rc;
    SFE_ARG(ctx, out && capacity_points > 0 && capacity_points < (1ll << 31) && max_clouds > 0);
    *out = nullptr;
    sfe_cloud_store *s = new sfe_cloud_store();
    s->ctx = ctx;
    s->capacity = capacity_points;
    s->max_clouds = max_clouds;
    if (hipMalloc((void **)&s->d_pool, sizeof(float2) * (size_t)capacity_points) != hipSuccess ||
        hipMalloc((void **)&s->d_off, sizeof(int64_t) * (size_t)max_clouds) != hipSuccess ||
        hipMalloc((void **)&s->d_cnt, sizeof(int32_t) * (size_t)max_clouds) != hipSuccess ||
        hipMalloc((void **)&s->d_top, sizeof(int64_t)) != hipSuccess ||
        hipMemsetAsync(s->d_top, 0, sizeof(int64_t), ctx->stream) != hipSuccess) {
        if (s->d_pool)
            (void)hipFree(s->d_pool);
        if (s->d_off)
            (void)hipFree(s->d_off);
        if (s->d_cnt)
            (void)hipFree(s->d_cnt);
        if (s->d_top)
            (void)hipFree(s->d_top);
        delete s;
        return sfe_set_err(ctx, SFE_ERR_HIP, "cloud store: allocating %lld points / %d slots failed",
                           (long long)capacity_points, max_clouds);
    }
    s->stamp.assign((size_t)max_clouds, 0);
    s->off.assign((size_t)max_clouds, 0);
    s->cnt.assign((size_t)max_clouds, 0);
    *out = s;
    return 0;
}

void sfe_cloud_store_destroy(sfe_cloud_store *s)
{
    if (!s)
        return;
    sfe_ctx *ctx = s->ctx;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipStreamSynchronize(ctx->stream2);
    (void)hipFree(s->d_pool);
    (void)hipFree(s->d_off);
    (void)hipFree(s->d_cnt);
    (void)hipFree(s->d_top);
    if (s->d_key)
        (void)hipFree(s->d_key);
    if (s->d_sel)
        (void)hipFree(s->d_sel);
    delete s;
}

int sfe_cloud_store_put_batch_dev(sfe_ctx *ctx, sfe_cloud_store *s, const int64_t *stamps, const float *d_clouds,
                                  const int32_t *d_counts, int n_frames, int64_t cap, int flags, int32_t *handles_out)
{
    if (int rc = sfe_use(ctx))
        return rc;
    SFE_ARG(ctx, s && s->ctx == ctx && n_frames >= 0 && cap >= 0 && (n_frames == 0 || (d_clouds && d_counts)));
    if (n_frames == 0)
        return 0;
    return sfe_store_append_dev(s, stamps, d_clouds, d_counts, n_frames, cap, flags, handles_out);
}

int sfe_cloud_store_put(sfe_ctx *ctx, sfe_cloud_store *s, int64_t stamp, const float *pts, int n, int32_t *handle_out)
{
    if (int rc = sfe_use(ctx))
        return rc;
    SFE_ARG(ctx, s && s->ctx == ctx && n >= 0 && (n == 0 || pts) && handle_out);
    // [count | points] through the event-guarded pinned staging: enqueue only
    const size_t bytes = 16 + sizeof(float) * 2 * (size_t)n;
    char *h = (char *)sfe_pinned_begin(ctx, bytes);
    char *d = (char *)sfe_scratch(ctx, 47, bytes);
    if (!h || !d)
        return SFE_ERR_HIP;
    *(int32_t *)h = n;
    if (n)
        memcpy(h + 16, pts, sizeof(float) * 2 * (size_t)n);
    SFE_HIP(ctx, hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, ctx->stream));
    if (int rc = sfe_pinned_end(ctx, ctx->stream))
        return rc;
    return sfe_store_append_dev(s, &stamp, (const float *)(d + 16), (const int32_t *)d, 1, n, 0, handle_out);
}

int sfe_cloud_store_count(sfe_cloud_store *s) { return s ? s->n_slots : 0; }

int sfe_cloud_store_meta(sfe_ctx *ctx, sfe_cloud_store *s, int32_t first, int32_t n, int64_t *stamps, int64_t *offsets,
                         int32_t *counts)
{
    if (int rc = sfe_use(ctx))
        return rc;
    SFE_ARG(ctx, s && s->ctx == ctx && first >= 0 && n >= 0 && first + n <= s->n_slots);
    if (int rc = store_sync_meta(s))
        return rc;
    for (int i = 0; i < n; ++i) {
        if (stamps)
            stamps[i] = s->stamp[first + i];
        if (offsets)
            offsets[i] = s->off[first + i];
        if (counts)
            counts[i] = s->cnt[first + i];
    }
    return 0;
}

int sfe_cloud_store_read(sfe_ctx *ctx, sfe_cloud_store *s, int32_t handle, float *out, int cap, int *n_out)
{
    if (int rc = sfe_use(ctx))
        return rc;
    SFE_ARG(ctx, s && s->ctx == ctx && handle >= 0 && handle < s->n_slots && n_out && cap >= 0 && (cap == 0 || out));
    if (int rc = store_sync_meta(s))
        return rc;
    const int n = s->cnt[handle];
    *n_out = n;
    if (n <= 0)
        return 0;
    if (n > cap)
        return sfe_set_err(ctx, SFE_ERR_CAP, "cloud store: cloud %d has %d points, the buffer holds %d", handle, n, cap);
    SFE_HIP(ctx, hipMemcpyAsync(out, s->d_pool + s->off[handle], sizeof(float2) * (size_t)n, hipMemcpyDeviceToHost,
                                ctx->stream));
    SFE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int sfe_cloud_store_truncate(sfe_ctx *ctx, sfe_cloud_store *s, int32_t n_slots)
{
    if (int rc = sfe_use(ctx))
        return rc;
    SFE_ARG(ctx, s && s->ctx == ctx && n_slots >= 0 && n_slots <= s->n_slots);
    if (n_slots == s->n_slots)
        return 0;
    // the fill level goes back to where the first dropped cloud began; a dropped cloud that never fitted has no
    // offset (-1) and no pool space (store_commit_kernel), so the level stays where it is
    if (n_slots == 0)
        SFE_HIP(ctx, hipMemsetAsync(s->d_top, 0, sizeof(int64_t), ctx->stream));
    else
        hipLaunchKernelGGL(store_rewind_kernel, dim3(1), dim3(1), 0, ctx->stream, s->d_top, (const int64_t *)s->d_off,
                           (int)n_slots, (int)s->n_slots);
    SFE_LAUNCH_CHECK(ctx);
    s->n_slots = n_slots;
    s->n_synced = std::min(s->n_synced, n_slots);
    if (s->sel_handle >= n_slots) // the cloud the selection belonged to is gone: its slot may come back as another cloud
        s->sel_handle = -1;
    if ((int32_t)s->keyed.size() > n_slots)
        s->keyed.resize((size_t)n_slots);
    return 0;
}

// get_points for n_jobs target clouds at once: handles [n_jobs x m] (-1 = unused), T6 [n_jobs x m x 6] (host)
int sfe_cloud_store_get_points(sfe_ctx *ctx, sfe_cloud_store *s, const int32_t *handles, const float *T6, int n_jobs, int m,
                               float resolution, int flags, const int64_t *stamps, int32_t *handles_out)
{
    if (int rc = sfe_use(ctx))
        return rc;
    SFE_ARG(ctx, s && s->ctx == ctx && n_jobs >= 0 && m >= 1 && (n_jobs == 0 || (handles && T6 && handles_out)));
    if (n_jobs == 0)
        return 0;
    if (int rc = store_sync_meta(s)) // sizes of the named clouds: the scratch of the filters is sized by the largest job
        return rc;
    int64_t cap = 1;
    for (int j = 0; j < n_jobs; ++j) {
        int64_t tot = 0;
        for (int k = 0; k < m; ++k) {
            const int h = handles[(size_t)j * m + k];
            if (h < 0)
                continue;
            if (h >= s->n_slots)
                return sfe_set_err(ctx, SFE_ERR_ARG, "get_points: job %d names cloud %d, the store holds %d", j, h, s->n_slots);
            if (s->cnt[h] < 0)
                return sfe_set_err(ctx, SFE_ERR_ARG, "get_points: job %d names cloud %d, which was not stored (count %d: -1 octree "
                                                     "too deep, -3 pool full)", j, h, s->cnt[h]);
            tot += s->cnt[h];
        }
        cap = std::max(cap, tot);
    }
    if (resolution > 0.0f && cap > CF_MAX_CAP) {
        // beyond the resident filter's capacity (16 index bits in its sort key): every job of the call, in order, through
        // the rank-sort path that knows no size limit (the NSSM target of a long session, slam.py:999)
        for (int j = 0; j < n_jobs; ++j)
            if (int rc = store_get_points_big(s, handles + (size_t)j * m, T6 + (size_t)j * m * 6, nullptr, m, resolution, flags,
                                              stamps ? stamps[j] : 0, handles_out + j))
                return rc;
        return 0;
    }
    const size_t nh = (size_t)n_jobs * m;
    const size_t b_tab = nh * (sizeof(int32_t) + 6 * sizeof(float));
    char *h_tab = (char *)sfe_pinned_begin(ctx, b_tab);
    char *d_tab = (char *)sfe_scratch(ctx, 48, b_tab);
    ctx->staged_frames = -1; // (the staging slots are rewritten)
    float2 *d_p32 = (float2 *)sfe_scratch(ctx, CF_SLOT_P32, sizeof(float2) * (size_t)cap * (size_t)n_jobs);
    CfHeader *d_hdr = (CfHeader *)sfe_scratch(ctx, CF_SLOT_HDR, sizeof(CfHeader) * (size_t)n_jobs);
    float *d_out = (float *)sfe_scratch(ctx, 49, sizeof(float2) * (size_t)cap * (size_t)n_jobs);
    int32_t *d_out_cnt = (int32_t *)sfe_scratch(ctx, 50, sizeof(int32_t) * (size_t)n_jobs);
    if (!h_tab || !d_tab || !d_p32 || !d_hdr || !d_out || !d_out_cnt)
        return SFE_ERR_HIP;
    memcpy(h_tab, T6, nh * 6 * sizeof(float));
    memcpy(h_tab + nh * 6 * sizeof(float), handles, nh * sizeof(int32_t));
    SFE_HIP(ctx, hipMemcpyAsync(d_tab, h_tab, b_tab, hipMemcpyHostToDevice, ctx->stream));
    if (int rc = sfe_pinned_end(ctx, ctx->stream))
        return rc;
    auto gather = (flags & SFE_STORE_F32_POINTS) ? store_gather_kernel<false> : store_gather_kernel<true>;
    hipLaunchKernelGGL(gather, dim3(n_jobs), dim3(1024), 0, ctx->stream, (const float2 *)s->d_pool,
                       (const int64_t *)s->d_off, (const int32_t *)s->d_cnt,
                       (const int32_t *)(d_tab + nh * 6 * sizeof(float)), (const float *)d_tab, m, (long long)cap,
                       sfe_cf_max_size(resolution), d_p32, d_hdr);
    if (int rc = sfe_cf_run_staged(ctx, n_jobs, cap, resolution, 0.0, 0, d_out, d_out_cnt))
        return rc;
    return sfe_store_append_dev(s, stamps, d_out, d_out_cnt, n_jobs, cap, 0, handles_out);
}


// get_points(frames, None, return_keys=True) (slam.py:229-292, :873): m keyframes, each under its own transform, with its
// key; pcl.downsample(points, keys, resolution) (pcl.cpp:143-159).  One new slot; its keys are read by the calls below.
int sfe_cloud_store_get_points_keys(sfe_ctx *ctx, sfe_cloud_store *s, const int32_t *handles, const float *T6,
                                    const int32_t *keys, int m, float resolution, int flags, int64_t stamp, int32_t *handle_out)
{
    if (int rc = sfe_use(ctx))
        return rc;
    SFE_ARG(ctx, s && s->ctx == ctx && m >= 0 && handle_out && (m == 0 || (handles && T6 && keys)));
    if (int rc = store_sync_meta(s))
        return rc;
    return store_get_points_big(s, handles, T6, keys, m, resolution, flags, stamp, handle_out);
}

static int store_keyed_cloud(sfe_cloud_store *s, int32_t handle, const char *what)
{
    sfe_ctx *ctx = s->ctx;
    if (int rc = store_sync_meta(s))
        return rc;
    if (handle < 0 || handle >= s->n_slots || s->cnt[handle] < 0)
        return sfe_set_err(ctx, SFE_ERR_ARG, "%s: cloud %d (the store holds %d) does not exist or was not stored", what, handle,
                           s->n_slots);
    if (!s->d_key || handle >= (int32_t)s->keyed.size() || !s->keyed[(size_t)handle])
        return sfe_set_err(ctx, SFE_ERR_ARG, "%s: cloud %d has no keys (it was not built by a keyed entry point)", what, handle);
    return 0;
}

int sfe_cloud_store_read_keys(sfe_ctx *ctx, sfe_cloud_store *s, int32_t handle, int32_t *out, int cap, int *n_out)
{
    if (int rc = sfe_use(ctx))
        return rc;
    SFE_ARG(ctx, s && s->ctx == ctx && n_out && cap >= 0 && (cap == 0 || out));
    if (int rc = store_keyed_cloud(s, handle, "read_keys"))
        return rc;
    const int n = s->cnt[handle];
    *n_out = n;
    if (n == 0)
        return 0;
    if (n > cap)
        return sfe_set_err(ctx, SFE_ERR_CAP, "cloud store: cloud %d has %d points, the buffer holds %d", handle, n, cap);
    SFE_HIP(ctx, hipMemcpyAsync(out, s->d_key + s->off[handle], sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    SFE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

static int store_sel_buffer(sfe_cloud_store *s, size_t n)
{
    if (s->d_sel && s->sel_cap >= n)
        return 0;
    if (s->d_sel)
        (void)hipFree(s->d_sel);
    s->d_sel = nullptr;
    s->sel_cap = 0;
    const size_t want = n + n / 4 + 4096;
    if (hipMalloc((void **)&s->d_sel, want) != hipSuccess) {
        s->d_sel = nullptr;
        return sfe_set_err(s->ctx, SFE_ERR_HIP, "cloud store: allocating the selection buffer (%zu) failed", want);
    }
    s->sel_cap = want;
    return 0;
}

// slam.py:875-899 + :902: frames = n_frames x {Tinv[6] float32 of pose.inverse().matrix(), range_bound, bearing_bound};
// -> per-key counts of the selected points (key_counts_out[n_keys]), their number, and how many points could not be decided
int sfe_cloud_store_fov_select(sfe_ctx *ctx, sfe_cloud_store *s, int32_t handle, const float *Tinv6, const double *range_bound,
                               const double *bearing_bound, int n_frames, int n_keys, int32_t *key_counts_out,
                               int32_t *n_selected_out, int32_t *n_ambiguous_out)
{
    if (int rc = sfe_use(ctx))
        return rc;
    SFE_ARG(ctx, s && s->ctx == ctx && n_frames >= 0 && n_keys >= 0 && n_selected_out && n_ambiguous_out &&
                     (n_frames == 0 || (Tinv6 && range_bound && bearing_bound)) && (n_keys == 0 || key_counts_out));
    if (int rc = store_keyed_cloud(s, handle, "fov_select"))
        return rc;
    const int n = s->cnt[handle];
    if (int rc = store_sel_buffer(s, (size_t)std::max(n, 1)))
        return rc;
    s->sel_handle = handle;
    s->sel_count = n;
    const size_t b_fr = sizeof(FovFrame) * (size_t)std::max(n_frames, 1), b_out = sizeof(int32_t) * ((size_t)n_keys + 2);
    FovFrame *h_fr = (FovFrame *)sfe_pinned_begin(ctx, b_fr);
    FovFrame *d_fr = (FovFrame *)sfe_scratch(ctx, 59, b_fr);
    int32_t *d_out = (int32_t *)sfe_scratch(ctx, 60, b_out);
    int32_t *h_out = (int32_t *)sfe_pinned_io(ctx, 3, b_out);
    if (!h_fr || !d_fr || !d_out || !h_out)
        return SFE_ERR_HIP;
    for (int f = 0; f < n_frames; ++f) {
        memcpy(h_fr[f].T, Tinv6 + 6 * (size_t)f, sizeof h_fr[f].T);
        h_fr[f].range_bound = range_bound[f];
        h_fr[f].bearing_bound = bearing_bound[f];
    }
    SFE_HIP(ctx, hipMemcpyAsync(d_fr, h_fr, b_fr, hipMemcpyHostToDevice, ctx->stream));
    if (int rc = sfe_pinned_end(ctx, ctx->stream))
        return rc;
    SFE_HIP(ctx, hipMemsetAsync(d_out, 0, b_out, ctx->stream));
    if (n > 0) {
        hipLaunchKernelGGL(store_fov_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                           (const float2 *)(s->d_pool + s->off[handle]), (const int32_t *)(s->d_key + s->off[handle]), n,
                           (const FovFrame *)d_fr, n_frames, n_keys, s->d_sel, d_out + 2, d_out);
        SFE_LAUNCH_CHECK(ctx);
    }
    SFE_HIP(ctx, hipMemcpyAsync(h_out, d_out, b_out, hipMemcpyDeviceToHost, ctx->stream));
    SFE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *n_selected_out = h_out[0];
    *n_ambiguous_out = h_out[1];
    if (n_keys)
        memcpy(key_counts_out, h_out + 2, sizeof(int32_t) * (size_t)n_keys);
    return 0;
}

// the selection of `handle` as the host computed it (the numpy path for a cloud fov_select could not decide)
int sfe_cloud_store_set_selection(sfe_ctx *ctx, sfe_cloud_store *s, int32_t handle, const uint8_t *sel, int n)
{
    if (int rc = sfe_use(ctx))
        return rc;
    SFE_ARG(ctx, s && s->ctx == ctx && n >= 0 && (n == 0 || sel));
    if (int rc = store_keyed_cloud(s, handle, "set_selection"))
        return rc;
    SFE_ARG(ctx, n == s->cnt[handle]);
    if (int rc = store_sel_buffer(s, (size_t)std::max(n, 1)))
        return rc;
    s->sel_handle = handle;
    s->sel_count = n;
    if (n) {
        SFE_HIP(ctx, hipMemcpyAsync(s->d_sel, sel, (size_t)n, hipMemcpyHostToDevice, ctx->stream));
        SFE_HIP(ctx, hipStreamSynchronize(ctx->stream)); // (pageable source)
    }
    return 0;
}

// target_points[sel], target_keys[sel] (slam.py:898-899) as a new slot with keys
int sfe_cloud_store_compact_selected(sfe_ctx *ctx, sfe_cloud_store *s, int32_t handle, int64_t stamp, int32_t *handle_out)
{
    if (int rc = sfe_use(ctx))
        return rc;
    SFE_ARG(ctx, s && s->ctx == ctx && handle_out);
    if (int rc = store_keyed_cloud(s, handle, "compact_selected"))
        return rc;
    if (s->sel_handle != handle || s->sel_count != s->cnt[handle])
        return sfe_set_err(ctx, SFE_ERR_ARG, "compact_selected: cloud %d has no selection (the last one was made for cloud %d of %d "
                           "points)", handle, s->sel_handle, s->sel_count);
    const int n = s->cnt[handle];
    const size_t cap = (size_t)std::max(n, 1);
    float2 *d_out = (float2 *)sfe_scratch(ctx, 55, sizeof(float2) * cap);
    int32_t *d_okey = (int32_t *)sfe_scratch(ctx, 57, sizeof(int32_t) * cap);
    int32_t *d_n = (int32_t *)sfe_scratch(ctx, 58, sizeof(SfeDsHeader));
    if (!d_out || !d_okey || !d_n)
        return SFE_ERR_HIP;
    hipLaunchKernelGGL(store_compact_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const float2 *)(s->d_pool + s->off[handle]),
                       (const int32_t *)(s->d_key + s->off[handle]), (const uint8_t *)s->d_sel, n, d_out, d_okey, d_n);
    SFE_LAUNCH_CHECK(ctx);
    int32_t h_new = -1;
    if (int rc = sfe_store_append_dev(s, &stamp, (const float *)d_out, d_n, 1, (int64_t)cap, 0, &h_new))
        return rc;
    hipLaunchKernelGGL(store_commit_keys_kernel, dim3(64), dim3(256), 0, ctx->stream, (const int32_t *)d_okey,
                       (const int64_t *)s->d_off, (const int32_t *)s->d_cnt, (int)h_new, s->d_key);
    SFE_LAUNCH_CHECK(ctx);
    if (h_new >= 0) {
        if ((int32_t)s->keyed.size() <= h_new)
            s->keyed.resize((size_t)h_new + 1, 0);
        s->keyed[(size_t)h_new] = 1;
    }
    *handle_out = h_new;
    return 0;
}

// slam.py:977-985: the source under T6 matched against a keyed target -> overlap and the per-key counts of the matches
int sfe_cloud_store_match_keys(sfe_ctx *ctx, sfe_cloud_store *s, int32_t source, const float *T6, int32_t target, float max_dist,
                               int flags, int n_keys, int32_t *key_counts_out, int32_t *overlap_out)
{
    if (int rc = sfe_use(ctx))
        return rc;
    SFE_ARG(ctx, s && s->ctx == ctx && T6 && overlap_out && n_keys >= 0 && (n_keys == 0 || key_counts_out));
    if (int rc = store_keyed_cloud(s, target, "match_keys"))
        return rc;
    if (source < 0 || source >= s->n_slots || s->cnt[source] < 0)
        return sfe_set_err(ctx, SFE_ERR_ARG, "match_keys: source cloud %d does not exist or was not stored", source);
    const int ns = s->cnt[source], nt = s->cnt[target];
    const size_t b_out = sizeof(int32_t) * ((size_t)n_keys + 1);
    float *h_T = (float *)sfe_pinned_begin(ctx, 6 * sizeof(float));
    float *d_T = (float *)sfe_scratch(ctx, 59, 6 * sizeof(float));
    int32_t *d_out = (int32_t *)sfe_scratch(ctx, 60, b_out);
    int32_t *h_out = (int32_t *)sfe_pinned_io(ctx, 3, b_out);
    if (!h_T || !d_T || !d_out || !h_out)
        return SFE_ERR_HIP;
    memcpy(h_T, T6, 6 * sizeof(float));
    SFE_HIP(ctx, hipMemcpyAsync(d_T, h_T, 6 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    if (int rc = sfe_pinned_end(ctx, ctx->stream))
        return rc;
    SFE_HIP(ctx, hipMemsetAsync(d_out, 0, b_out, ctx->stream));
    if (ns > 0) {
        auto mk = (flags & SFE_STORE_F32_POINTS) ? store_match_keys_kernel<false> : store_match_keys_kernel<true>;
        hipLaunchKernelGGL(mk, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, ctx->stream,
                           (const float2 *)(s->d_pool + s->off[source]), ns, (const float2 *)(s->d_pool + s->off[target]),
                           (const int32_t *)(s->d_key + s->off[target]), nt, (const float *)d_T, max_dist * max_dist, n_keys,
                           d_out + 1, d_out);
        SFE_LAUNCH_CHECK(ctx);
    }
    SFE_HIP(ctx, hipMemcpyAsync(h_out, d_out, b_out, hipMemcpyDeviceToHost, ctx->stream));
    SFE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *overlap_out = h_out[0];
    if (n_keys)
        memcpy(key_counts_out, h_out + 1, sizeof(int32_t) * (size_t)n_keys);
    return 0;
}

// jobs over handles -> the job table of the ICP launcher (sizes from the mirrored slot table)
static int store_jobs4(sfe_cloud_store *s, const int32_t *pairs, int n_jobs, std::vector<int32_t> &jobs4)
{
    sfe_ctx *ctx = s->ctx;
    if (int rc = store_sync_meta(s))
        return rc;
    jobs4.resize(4 * (size_t)n_jobs);
    for (int j = 0; j < n_jobs; ++j) {
        const int hs = pairs[2 * j], ht = pairs[2 * j + 1];
        if (hs < 0 || ht < 0 || hs >= s->n_slots || ht >= s->n_slots)
            return sfe_set_err(ctx, SFE_ERR_ARG, "scan match %d names clouds (%d, %d), the store holds %d", j, hs, ht,
                               s->n_slots);
        if (s->cnt[hs] <= 0 || s->cnt[ht] <= 0)
            return sfe_set_err(ctx, SFE_ERR_ARG, "scan match %d: cloud %d has %d points, cloud %d has %d (ICP needs "
                                                 "non-empty clouds; the caller tests ssm_min_points first, slam.py:745)",
                               j, hs, s->cnt[hs], ht, s->cnt[ht]);
        jobs4[4 * j] = (int32_t)s->off[hs];
        jobs4[4 * j + 1] = s->cnt[hs];
        jobs4[4 * j + 2] = (int32_t)s->off[ht];
        jobs4[4 * j + 3] = s->cnt[ht];
    }
    return 0;
}

int sfe_icp_store_jobs_dev(sfe_ctx *ctx, const sfe_icp_params *p, sfe_cloud_store *s, const int32_t *pairs,
                           const float *d_guess9, int n_jobs, float *d_T9, int32_t *d_status, int32_t *d_iters)
{
    if (int rc = sfe_use(ctx))
        return rc;
    SFE_ARG(ctx, p && s && s->ctx == ctx && n_jobs >= 0 && (n_jobs == 0 || (pairs && d_guess9 && d_T9 && d_status && d_iters)));
    if (n_jobs == 0)
        return 0;
    std::vector<int32_t> jobs4;
    if (int rc = store_jobs4(s, pairs, n_jobs, jobs4))
        return rc;
    return sfe_icp_jobs_dev(ctx, p, (const float *)s->d_pool, (const float *)s->d_pool, jobs4.data(), d_guess9, n_jobs, d_T9,
                            d_status, d_iters);
}

int sfe_icp_store_compute(sfe_ctx *ctx, const sfe_icp_params *p, sfe_cloud_store *s, const int32_t *pairs,
                          const float *guesses9, int n_jobs, float *T_out9, int32_t *status, int32_t *iters)
{
    if (int rc = sfe_use(ctx))
        return rc;
    SFE_ARG(ctx, p && s && s->ctx == ctx && n_jobs >= 0 && (n_jobs == 0 || (pairs && guesses9 && T_out9 && status)));
    if (n_jobs == 0)
        return 0;
    std::vector<int32_t> jobs4;
    if (int rc = store_jobs4(s, pairs, n_jobs, jobs4))
        return rc;
    const size_t b_g = sizeof(float) * 9 * (size_t)n_jobs, b_out = (sizeof(float) * 9 + 2 * sizeof(int32_t)) * (size_t)n_jobs;
    float *d_g = (float *)sfe_scratch(ctx, 2, b_g);
    char *d_out = (char *)sfe_scratch(ctx, 3, b_out);
    char *h_in = (char *)sfe_pinned_io(ctx, 2, b_g);
    char *h_out = (char *)sfe_pinned_io(ctx, 3, b_out);
    if (!d_g || !d_out || !h_in || !h_out)
        return SFE_ERR_HIP;
    const int was_unsplit = ctx->icp_variant & 16;
    int rc = 0;
    for (int attempt = 0;; ++attempt) {
        memcpy(h_in, guesses9, b_g);
        float *d_T = (float *)d_out;
        int32_t *d_st = (int32_t *)(d_out + sizeof(float) * 9 * (size_t)n_jobs);
        if (hipMemcpyAsync(d_g, h_in, b_g, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
            rc = sfe_set_err(ctx, SFE_ERR_HIP, "scan match over the store: upload of the guesses failed");
            break;
        }
        if ((rc = sfe_icp_jobs_dev(ctx, p, (const float *)s->d_pool, (const float *)s->d_pool, jobs4.data(), d_g, n_jobs, d_T,
                                   d_st, d_st + n_jobs)))
            break;
        if (hipMemcpyAsync(h_out, d_out, b_out, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess) {
            rc = sfe_set_err(ctx, SFE_ERR_HIP, "scan match over the store: download of the results failed");
            break;
        }
        memcpy(T_out9, h_out, sizeof(float) * 9 * (size_t)n_jobs);
        memcpy(status, h_out + sizeof(float) * 9 * (size_t)n_jobs, sizeof(int32_t) * (size_t)n_jobs);
        if (iters)
            memcpy(iters, h_out + (sizeof(float) * 9 + sizeof(int32_t)) * (size_t)n_jobs, sizeof(int32_t) * (size_t)n_jobs);
        // a job shared by several workgroups whose shares were not resident together: once more, unsplit (sonarfe.h,
        // sfe_icp_set_tuning bit 4)
        bool timeout = false;
        for (int j = 0; j < n_jobs; ++j)
            timeout |= status[j] == SFE_ICP_SPLIT_TIMEOUT;
        if (!timeout || attempt == 1 || was_unsplit)
            break;
        ctx->icp_variant |= 16;
    }
    if (!was_unsplit)
        ctx->icp_variant &= ~16;
    return rc;
}

int sfe_cloud_store_overlap(sfe_ctx *ctx, sfe_cloud_store *s, const int32_t *pairs, const float *T6, int n_jobs,
                            float max_dist, int flags, int32_t *counts_out)
{
    if (int rc = sfe_use(ctx))
        return rc;
    SFE_ARG(ctx, s && s->ctx == ctx && n_jobs >= 0 && (n_jobs == 0 || (pairs && T6 && counts_out)));
    if (n_jobs == 0)
        return 0;
    if (int rc = store_sync_meta(s))
        return rc;
    for (int j = 0; j < 2 * n_jobs; ++j) {
        if (pairs[j] >= s->n_slots)
            return sfe_set_err(ctx, SFE_ERR_ARG, "overlap: cloud %d named, the store holds %d", pairs[j], s->n_slots);
        if (pairs[j] >= 0 && s->cnt[pairs[j]] < 0)
            return sfe_set_err(ctx, SFE_ERR_ARG, "overlap: cloud %d was not stored (count %d: -1 octree too deep, -3 pool full)",
                               pairs[j], s->cnt[pairs[j]]);
    }
    const size_t b_in = (size_t)n_jobs * (2 * sizeof(int32_t) + 6 * sizeof(float)), b_out = (size_t)n_jobs * sizeof(int32_t);
    char *h_in = (char *)sfe_pinned_io(ctx, 2, b_in);
    char *h_out = (char *)sfe_pinned_io(ctx, 3, b_out);
    char *d_in = (char *)sfe_scratch(ctx, 48, b_in);
    int32_t *d_out = (int32_t *)sfe_scratch(ctx, 50, b_out);
    if (!h_in || !h_out || !d_in || !d_out)
        return SFE_ERR_HIP;
    memcpy(h_in, T6, (size_t)n_jobs * 6 * sizeof(float));
    memcpy(h_in + (size_t)n_jobs * 6 * sizeof(float), pairs, (size_t)n_jobs * 2 * sizeof(int32_t));
    SFE_HIP(ctx, hipMemcpyAsync(d_in, h_in, b_in, hipMemcpyHostToDevice, ctx->stream));
    auto overlap = (flags & SFE_STORE_F32_POINTS) ? store_overlap_kernel<false> : store_overlap_kernel<true>;
    hipLaunchKernelGGL(overlap, dim3(n_jobs), dim3(256), 0, ctx->stream, (const float2 *)s->d_pool,
                       (const int64_t *)s->d_off, (const int32_t *)s->d_cnt,
                       (const int32_t *)(d_in + (size_t)n_jobs * 6 * sizeof(float)), (const float *)d_in, max_dist * max_dist,
                       d_out);
    SFE_LAUNCH_CHECK(ctx);
    SFE_HIP(ctx, hipMemcpyAsync(h_out, d_out, b_out, hipMemcpyDeviceToHost, ctx->stream));
    SFE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(counts_out, h_out, b_out);
    return 0;
}

} // extern "C"
