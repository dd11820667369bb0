// ICP scan matching with an exact sorted-sweep nearest-neighbour search (default ICP path).
// Replaces bruce_slam/src/bruce_slam/cpp/pcl.cpp:198-212 (ICP.compute -> libpointmatcher chain of
// bruce_slam/config/icp.yaml:1-31); same chain, same decisions and same arithmetic as
// sfe_icp.hip's brute-force kernel (which stays as the A/B
// baseline and checker) -- only the order in which candidate pairs are visited differs.
//
// Why a sweep is exact.  The squared distance everyone on this path compares is
//     d2 = fl( fl(dx*dx) + fl(dy*dy) ),  dx = fl(px - tx), dy = fl(py - ty)      (dist2())
// Rounding is monotone, so d2 >= fl(dx*dx) =: e, and e is non-decreasing in |px - tx|.  With the
// centred target sorted by x, a query walks outwards from its own x position in both directions
// and may stop a direction as soon as e > best: every point further out has d2 >= e > best and
// can neither win nor tie.  All surviving candidates are evaluated with dist2()'s exact
// expression; ties go to the lowest ORIGINAL target index (what the brute-force scan and the
// oracle do), resolved by a rare second walk over the final window.  No kd-tree, no
// approximation, no float re-association: match ids and d2 are bit-identical to brute force.
//
// Work per query drops from n_tgt pair evaluations to the points whose |dx| is within the query's
// own stop bound: ~20 instead of 5000 on converged sonar clouds, a few hundred while the clouds are
// still far apart.  Walks differ wildly in length (a near-vertical wall puts 100+ points in one x
// window; a query with nothing within maxDist walks its whole 10 m window), so the search is tiered
// (details at the loop kernel): short budget for every lane -> survivors compacted into dense waves
// with a longer budget -> what still runs is finished by a whole wave, 256 candidates per trip.
//
// Mapping: prep kernel = one workgroup per distinct target (many guesses on one pair share it):
// mean, centre, bitonic sort of (x-key, index) in LDS (HBM scratch beyond 8192 points), sorted
// cloud + permutation to HBM scratch, PCA normals (k-NN by the same sweep) for point-to-plane.
// Loop kernel = one workgroup per job, all ICP iterations in one launch: sorted target resident in
// LDS (or walked through L2 beyond 8192 points), per iteration: transform + lower bound + capped
// walks (tiers) -> census -> trimmed quantile by exact radix select -> fp64 reduction of the 9(+1)
// sums -> closed-form solve and checkers on one lane.
#include "sfe_icp_common.h"

#include <algorithm>
#include <cstdlib>
#include <map>
#include <type_traits>
#include <utility>

#define SW_TCAP 8192   // target points resident in LDS

struct SweepPrep {
    int tgt_start, n_tgt;
    long long off;     // offset (points) of this target's slice of the sorted-cloud scratch (stride n_tgt + 4)
    long long key_off; // targets beyond the LDS capacity: offset of their sort keys in HBM scratch
};

struct SweepJob {
    int src_start, n_src, n_tgt, prep;
    long long tgt_off; // = SweepPrep.off of its target
    long long q_off;   // offset (points) of this job's slice of the per-query scratch
};

// order-preserving map float -> uint32 (NaN of either sign sorts last)
__device__ __forceinline__ unsigned mono_key(float x)
{
    const unsigned u = __float_as_uint(x);
    if (x != x)
        return 0xFFFFFFFFu;
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float mono_inv(unsigned k)
{
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

// in-LDS bitonic sort of n2 (power of two) 64-bit keys, ascending
__device__ __forceinline__ void bitonic_sort_lds(unsigned long long *keys, unsigned n2)
{
    for (unsigned k = 2; k <= n2; k <<= 1) {
        for (unsigned j = k >> 1; j > 0; j >>= 1) {
            for (unsigned t = threadIdx.x; t < n2 / 2; t += ICP_THREADS) {
                const unsigned i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const unsigned l = i | j;
                const unsigned long long a = keys[i], b = keys[l];
                const bool up = (i & k) == 0;
                if ((a > b) == up) {
                    keys[i] = b;
                    keys[l] = a;
                }
            }
            __syncthreads();
        }
    }
}

struct PrepShared {
    // first the sort keys, then (same bytes) the sorted cloud with one NaN sentinel at each end
    unsigned long long buf[SW_TCAP + 2];
    double red[ICP_WAVES * 2 + 2];
    float mean[2];
};

// ---------------------------------------------------------------------------------------------
// prep: one workgroup per distinct target cloud
// ---------------------------------------------------------------------------------------------
// PCA normals of the centred target: K nearest incl. the point itself, ordered by (d2, original
// index) exactly like the brute-force scan (sfe_icp.hip).  s_tgt = sorted cloud with sentinels
// (1-based positions), in LDS or in HBM scratch.
template <int KM> // capacity of the neighbour list (>= K): its loops are fully unrolled, so a snug KM pays
__device__ __forceinline__ void sweep_knn_normals(const sfe_icp_params &P, const float2 *__restrict__ s_tgt,
                                                  const int *__restrict__ perm, float2 *__restrict__ snrm, int nt)
{
    const int tid = threadIdx.x;
    const int K = min(min(P.normals_knn, KM), nt);
    for (int c = tid; c < nt; c += ICP_THREADS) {
        const float2 q = s_tgt[c + 1];
        float bd[KM];
        int bj[KM];
#pragma unroll
        for (int k = 0; k < KM; ++k) {
            bd[k] = INFINITY;
            bj[k] = 0;
        }
        float kth = INFINITY; // bd[K-1]
        int iL = c, iR = c + 1;  // 1-based positions: the point itself is the first right candidate
        auto consider = [&](float d, int j) {
            if (!(d <= kth) || d == INFINITY)
                return;
            if (K == KM) {
                // full list (the usual case, KM == k): the newcomer replaces the last entry and bubbles up
                // with KM-1 compare-exchanges -- half the work of the count / shift / place form below.
                // Equal distances order by original index (rare: the permutation is only read then).
                if (d == kth && !(perm[j - 1] < perm[bj[KM - 1] - 1]))
                    return;
                bd[KM - 1] = d;
                bj[KM - 1] = j;
#pragma unroll
                for (int k = KM - 1; k >= 1; --k) {
                    bool up = bd[k] < bd[k - 1];
                    if (bd[k] == bd[k - 1] && bj[k - 1] != 0)
                        up = perm[bj[k] - 1] < perm[bj[k - 1] - 1];
                    const float td = up ? bd[k - 1] : bd[k];
                    const int tj = up ? bj[k - 1] : bj[k];
                    bd[k - 1] = up ? bd[k] : bd[k - 1];
                    bj[k - 1] = up ? bj[k] : bj[k - 1];
                    bd[k] = td;
                    bj[k] = tj;
                }
                kth = bd[KM - 1];
                return;
            }
            int p = 0;
            bool eq = false;
#pragma unroll
            for (int k = 0; k < KM; ++k) {
                p += (k < K && bd[k] < d) ? 1 : 0;
                eq |= (k < K && bd[k] == d);
            }
            if (eq) { // ties: lower original index first
                const int o = perm[j - 1];
#pragma unroll
                for (int k = 0; k < KM; ++k)
                    if (k < K && bd[k] == d && perm[bj[k] - 1] < o)
                        ++p;
            }
            if (p >= K)
                return;
#pragma unroll
            for (int k = KM - 1; k >= 1; --k)
                if (k < K && k > p) {
                    bd[k] = bd[k - 1];
                    bj[k] = bj[k - 1];
                }
#pragma unroll
            for (int k = 0; k < KM; ++k) {
                if (k == p) {
                    bd[k] = d;
                    bj[k] = j;
                }
                if (k == K - 1)
                    kth = bd[k];
            }
        };
        while (true) {
            const float2 tl = s_tgt[iL], tr = s_tgt[iR];
            const float dxl = f_add(q.x, -tl.x), el = f_mul(dxl, dxl);
            const float dyl = f_add(q.y, -tl.y), dl = f_add(el, f_mul(dyl, dyl));
            const float dxr = f_add(q.x, -tr.x), er = f_mul(dxr, dxr);
            const float dyr = f_add(q.y, -tr.y), dr = f_add(er, f_mul(dyr, dyr));
            const bool okl = el <= kth, okr = er <= kth; // NaN sentinel -> false
            if (!(okl || okr))
                break;
            if (okr)
                consider(dr, iR);
            if (okl)
                consider(dl, iL);
            iL -= okl ? 1 : 0;
            iR += okr ? 1 : 0;
        }
        double sx = 0, sy = 0;
#pragma unroll
        for (int k = 0; k < KM; ++k)
            if (k < K) {
                const float2 t = s_tgt[bj[k]];
                sx += (double)t.x;
                sy += (double)t.y;
            }
        sx /= K;
        sy /= K;
        double a = 0, b = 0, d = 0;
#pragma unroll
        for (int k = 0; k < KM; ++k)
            if (k < K) {
                const float2 t = s_tgt[bj[k]];
                const double ux = (double)t.x - sx, uy = (double)t.y - sy;
                a += ux * ux;
                b += ux * uy;
                d += uy * uy;
            }
        const double u = a - d, w = 2 * b, h = sqrt(u * u + w * w);
        double tx, ty;
        if (h == 0) {
            tx = 1;
            ty = 0;
        } else if (u >= 0) {
            tx = u + h;
            ty = w;
        } else {
            tx = w;
            ty = h - u;
        }
        double nn = sqrt(tx * tx + ty * ty);
        if (nn == 0) {
            tx = 1;
            ty = 0;
            nn = 1;
        }
        snrm[c] = make_float2((float)(-ty / nn), (float)(tx / nn));
    }
}

// bitonic sort of n2 (power of two) 64-bit keys in HBM scratch by one workgroup (targets that do
// not fit LDS; once per target, the keys stay in L2)
__device__ __forceinline__ void bitonic_sort_global(unsigned long long *keys, unsigned n2)
{
    for (unsigned k = 2; k <= n2; k <<= 1) {
        for (unsigned j = k >> 1; j > 0; j >>= 1) {
            for (unsigned t = threadIdx.x; t < n2 / 2; t += ICP_THREADS) {
                const unsigned i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const unsigned l = i | j;
                const unsigned long long a = keys[i], b = keys[l];
                const bool up = (i & k) == 0;
                if ((a > b) == up) {
                    keys[i] = b;
                    keys[l] = a;
                }
            }
            __syncthreads(); // same workgroup, same CU: its L1 sees its own write-through stores
        }
    }
}

// Sorted-cloud scratch layout per target (stride n_tgt + 4 points): [0] NaN sentinel,
// [1 .. n_tgt] centred points ascending in x, [n_tgt+1], [n_tgt+2] NaN sentinels.  perm / snrm
// use the same stride, entry p-1 belongs to sorted position p.
__global__ __launch_bounds__(ICP_THREADS, 4) void icp_sweep_prep_kernel(sfe_icp_params P,
                                                                        const SweepPrep *__restrict__ preps,
                                                                        const float2 *__restrict__ tgt_all,
                                                                        float2 *__restrict__ stgt_all,
                                                                        int *__restrict__ perm_all,
                                                                        float2 *__restrict__ snrm_all,
                                                                        float *__restrict__ mean_all,
                                                                        unsigned long long *__restrict__ gkeys_all)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    PrepShared &S = *reinterpret_cast<PrepShared *>(smem_raw);
    const SweepPrep J = preps[blockIdx.x];
    const int nt = J.n_tgt, tid = threadIdx.x;
    const float2 *__restrict__ tgt = tgt_all + J.tgt_start;
    float2 *__restrict__ stgt = stgt_all + J.off;
    int *__restrict__ perm = perm_all + J.off;
    const float qnan = __uint_as_float(0x7FC00000u);

    // reference mean (fp64 accumulation, rounded to float), as the brute-force kernel
    {
        double m[2] = {0, 0};
        for (int i = tid; i < nt; i += ICP_THREADS) {
            const float2 t = tgt[i];
            m[0] += t.x;
            m[1] += t.y;
        }
        block_sum<2>(m, S.red);
        if (tid == 0) {
            S.mean[0] = (float)(m[0] / nt);
            S.mean[1] = (float)(m[1] / nt);
            mean_all[2 * blockIdx.x] = S.mean[0];
            mean_all[2 * blockIdx.x + 1] = S.mean[1];
        }
        __syncthreads();
    }
    const float mx = S.mean[0], my = S.mean[1];

    // sort (key(x - mean_x), index)
    unsigned n2 = 2;
    while (n2 < (unsigned)nt)
        n2 <<= 1;
    const bool in_lds = nt <= SW_TCAP;
    unsigned long long *keys = in_lds ? S.buf : gkeys_all + J.key_off;
    for (unsigned i = tid; i < n2; i += ICP_THREADS) {
        unsigned long long k = ~0ull;
        if (i < (unsigned)nt)
            k = ((unsigned long long)mono_key(f_add(tgt[i].x, -mx)) << 32) | i;
        keys[i] = k;
    }
    __syncthreads();
    if (tid == 0) {
        stgt[0] = make_float2(qnan, qnan);
        stgt[nt + 1] = make_float2(qnan, qnan);
        stgt[nt + 2] = make_float2(qnan, qnan);
    }
    if (in_lds) {
        bitonic_sort_lds(S.buf, n2);
        // keys -> sorted centred cloud (registers -> same LDS bytes, shifted by the left sentinel)
        constexpr int PER = SW_TCAP / ICP_THREADS;
        float2 v[PER];
        int id[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int pos = k * ICP_THREADS + tid;
            v[k] = make_float2(0, 0);
            id[k] = 0;
            if (pos < nt) {
                const unsigned long long key = S.buf[pos];
                id[k] = (int)(unsigned)(key & 0xFFFFFFFFu);
                v[k] = make_float2(mono_inv((unsigned)(key >> 32)), f_add(tgt[id[k]].y, -my));
            }
        }
        __syncthreads();
        float2 *s_tgt = reinterpret_cast<float2 *>(S.buf);
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int pos = k * ICP_THREADS + tid;
            if (pos < nt) {
                s_tgt[pos + 1] = v[k];
                stgt[pos + 1] = v[k];
                perm[pos] = id[k];
            }
        }
        if (tid == 0) {
            s_tgt[0] = make_float2(qnan, qnan);
            s_tgt[nt + 1] = make_float2(qnan, qnan);
        }
        __syncthreads();
        if (P.minimizer == 1) {
            if (P.normals_knn <= 8)
                sweep_knn_normals<8>(P, s_tgt, perm, snrm_all + J.off, nt);
            else if (P.normals_knn <= 10)
                sweep_knn_normals<10>(P, s_tgt, perm, snrm_all + J.off, nt);
            else if (P.normals_knn <= 12)
                sweep_knn_normals<12>(P, s_tgt, perm, snrm_all + J.off, nt);
            else
                sweep_knn_normals<ICP_KMAX>(P, s_tgt, perm, snrm_all + J.off, nt);
        }
    } else {
        bitonic_sort_global(keys, n2);
        for (int pos = tid; pos < nt; pos += ICP_THREADS) {
            const unsigned long long key = keys[pos];
            const int id = (int)(unsigned)(key & 0xFFFFFFFFu);
            stgt[pos + 1] = make_float2(mono_inv((unsigned)(key >> 32)), f_add(tgt[id].y, -my));
            perm[pos] = id;
        }
        __syncthreads();
        if (P.minimizer == 1) {
            if (P.normals_knn <= 8)
                sweep_knn_normals<8>(P, stgt, perm, snrm_all + J.off, nt);
            else if (P.normals_knn <= 10)
                sweep_knn_normals<10>(P, stgt, perm, snrm_all + J.off, nt);
            else if (P.normals_knn <= 12)
                sweep_knn_normals<12>(P, stgt, perm, snrm_all + J.off, nt);
            else
                sweep_knn_normals<ICP_KMAX>(P, stgt, perm, snrm_all + J.off, nt);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// loop: one workgroup per job
// ---------------------------------------------------------------------------------------------
// Only pairs that end up with weight 1 need their exact neighbour: d2 <= the trimmed-quantile
// limit (and <= MaxDist^2).  A search is therefore exhaustive only out to a cap C (squared
// radius), and merely keeps going until it has seen SOME target within KDTreeMatcher.maxDist so
// that the count of finite matches is exact.  A query ends as
//   none    : no target within maxDist (exact: its whole maxDist window was walked)
//   exact   : best <= C, every candidate with fl(dx*dx) <= best was evaluated
//   inexact : finite, C < d2_NN <= best            (its walk is suspended, state kept)
// If the exact set holds more than k = floor(n_finite * ratio) values, the k-th smallest of them
// IS the k-th smallest of all (everything else is > C), the limit is exact and so are all
// weight-1 pairs.  Otherwise C grows 4x and the suspended walks resume where they stopped.  C
// starts from the previous iteration's limit, so far outliers cost a handful of steps instead of
// a walk across the whole cloud.  Decisions and results are identical to the exhaustive search.
// control block of a job; the LDS-resident variant places the sorted target right behind it
// wave-uniform float held in an SGPR instead of one VGPR per lane (the loop kernel runs at the
// 64-VGPR budget: every uniform value kept out of the vector file is one spill less)
__device__ __forceinline__ float sw_uniform(float v)
{
    return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v)));
}

__device__ __forceinline__ long long sw_uniform_ll(long long v)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v & 0xFFFFFFFFll));
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)v >> 32));
    return (long long)(((unsigned long long)hi << 32) | lo);
}

struct SweepShared {
    double red[ICP_WAVES * 10 + 10];
    double acc[10];  // the reduced error-minimiser sums (read by the solving lane)
    unsigned hist[256];
    unsigned sel_prefix, sel_k;
    int long_n, long_next, mid_n, wl_n[2];
    int flag_iterate, flag_status;
    float Ti[9];
    float hist_c[ICP_MAX_HIST], hist_s[ICP_MAX_HIST], hist_x[ICP_MAX_HIST], hist_y[ICP_MAX_HIST];
    long long prof_t, prof[16], prof_it[64], prof_b0;
};

#define SW_PROF(k)                                                                               \
    do {                                                                                         \
        if (prof != nullptr && threadIdx.x == 0) {                                               \
            const long long t_ = clock64();                                                      \
            S.prof[k] += t_ - S.prof_t;                                                          \
            S.prof_t = t_;                                                                       \
        }                                                                                        \
    } while (0)

// debug watchdog: a loop that exceeds its bound records a code instead of hanging the device
#define SW_WATCH(cnt, bound, code)                                                               \
    if (++(cnt) > (bound)) {                                                                     \
        if (dbg)                                                                                 \
            atomicMax(dbg + (code), (int)blockIdx.x + 1);                                        \
        break;                                                                                   \
    }
#define SW_NQ 8         // results fetched per lane and batch in the census / quantile / reduction loops
#define SW_BUDGET_A 8   // first-pass trips per walk (4 candidates each)
#define SW_BUDGET 24    // tier-1 trips (4 candidates each) before a walk is handed to the cooperative tier
#define SW_NONE (-1)
// a suspended (inexact) query is stored as pos = -2 - bpos (<= -3): the target it holds doubles as the
// next iteration's witness
#define SW_INEXACT_OF(bpos) (-2 - (bpos))

struct SweepQ { // per-job views of the per-query scratch
    float2 *xy;   // transformed query
    int4 *st;     // suspended walk: x = iL, y = iR, z = bpos | tied << 31; a `none` query keeps its
                  // clearance record here instead: (px, py, clearance) as float bits
    float *d2;    // best so far / final d2
    int *pos;     // >= 0 sorted position of the NN, SW_NONE, <= -3 inexact (SW_INEXACT_OF)
    int *wl[2];   // work lists of suspended queries (ping-pong between rounds)
    int *mid;     // walks that outlived the short first pass (compacted for the second)
    int4 *longe;  // walks handed to the cooperative tier this round: (q, iL, iR, bpos|tied), (px, py, best, -)
    const int *perm;
    int nt;
};

// ties at the final best: lowest original index among the points at distance `best`, found by
// walking the final window once more (rare)
__device__ __forceinline__ int sweep_resolve_tie(const float2 *__restrict__ s_tgt, const SweepQ &Q, float px,
                                                 float py, float best, int iL, int iR)
{
    // the walk window is bounded by the suspended cursors: everything with e <= best lies inside
    int bo = 0x7FFFFFFF, bp = 0;
    int lo = 1, hi = Q.nt + 1; // first position with x >= px
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (s_tgt[mid].x < px)
            lo = mid + 1;
        else
            hi = mid;
    }
    (void)iL;
    (void)iR;
    for (int j = lo - 1; j >= 1; --j) {
        const float2 t = s_tgt[j];
        const float dx = f_add(px, -t.x), e = f_mul(dx, dx);
        if (!(e <= best))
            break;
        const float dy = f_add(py, -t.y);
        if (f_add(e, f_mul(dy, dy)) == best) {
            const int o = Q.perm[j - 1];
            if (o < bo) {
                bo = o;
                bp = j;
            }
        }
    }
    for (int j = lo; j <= Q.nt; ++j) {
        const float2 t = s_tgt[j];
        const float dx = f_add(px, -t.x), e = f_mul(dx, dx);
        if (!(e <= best))
            break;
        const float dy = f_add(py, -t.y);
        if (f_add(e, f_mul(dy, dy)) == best) {
            const int o = Q.perm[j - 1];
            if (o < bo) {
                bo = o;
                bp = j;
            }
        }
    }
    return bp;
}

// LDS_TGT: sorted target resident in LDS (n_tgt <= SW_TCAP) or read from its HBM scratch slice (it
// stays in L2: <= 160 KB for a 20k-point cloud, shared by all guesses of a many-to-one batch)
template <int MINW, bool LDS_TGT>
__global__ __launch_bounds__(ICP_THREADS, MINW) void icp_sweep_kernel(
    sfe_icp_params P, const SweepJob *__restrict__ jobs, const int *__restrict__ job_ids, const float2 *__restrict__ src_all,
    const float *__restrict__ guess_all, const float2 *__restrict__ stgt_all, const int *__restrict__ perm_all,
    const float2 *__restrict__ snrm_all, const float *__restrict__ mean_all, float2 *__restrict__ q_xy_all,
    int4 *__restrict__ q_st_all, int *__restrict__ q_wl_all, int4 *__restrict__ q_long_all, float *__restrict__ nn_d2_all,
    int *__restrict__ nn_pos_all, float *__restrict__ T_out, int *__restrict__ status_out,
    int *__restrict__ iters_out, long long *prof, int *dbg, int sw_budget, int sw_budget_a, int sw_cache)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    SweepShared &S = *reinterpret_cast<SweepShared *>(smem_raw);

    const int jb = __builtin_amdgcn_readfirstlane(job_ids[blockIdx.x]);
    SweepJob J = jobs[jb];
    // the job record is the same for every lane: keep it (and every pointer derived from it) in SGPRs
    J.src_start = __builtin_amdgcn_readfirstlane(J.src_start);
    J.n_src = __builtin_amdgcn_readfirstlane(J.n_src);
    J.n_tgt = __builtin_amdgcn_readfirstlane(J.n_tgt);
    J.prep = __builtin_amdgcn_readfirstlane(J.prep);
    J.tgt_off = sw_uniform_ll(J.tgt_off);
    J.q_off = sw_uniform_ll(J.q_off);
    const int ns = J.n_src, nt = J.n_tgt;
    const float2 *__restrict__ src = src_all + J.src_start;
    const float2 *__restrict__ stgt = stgt_all + J.tgt_off;
    float2 *lds_tgt = reinterpret_cast<float2 *>(smem_raw + ((sizeof(SweepShared) + 15) & ~(size_t)15));
    const float2 *__restrict__ T = LDS_TGT ? (const float2 *)lds_tgt : stgt; // sorted target incl. sentinels
    const float2 *__restrict__ snrm = snrm_all ? snrm_all + J.tgt_off : nullptr;
    SweepQ Q;
    Q.xy = q_xy_all + J.q_off;
    Q.st = q_st_all + J.q_off;
    Q.d2 = nn_d2_all + J.q_off;
    Q.pos = nn_pos_all + J.q_off;
    Q.wl[0] = q_wl_all + 3 * J.q_off;
    Q.wl[1] = Q.wl[0] + ns;
    Q.mid = Q.wl[1] + ns;
    Q.longe = q_long_all + 2 * J.q_off;
    Q.perm = perm_all + J.tgt_off;
    Q.nt = nt;
    const float *guess = guess_all + 9 * (size_t)jb;
    const int tid = threadIdx.x, lane = threadIdx.x & 63;
    const float mx = sw_uniform(mean_all[2 * J.prep]), my = sw_uniform(mean_all[2 * J.prep + 1]);

    if (prof != nullptr && tid == 0) {
        for (int i = 0; i < 16; ++i)
            S.prof[i] = 0;
        S.prof_t = clock64();
    }
    // sorted centred target (with its NaN sentinels: a NaN stops a walk direction) -> LDS
    if (LDS_TGT) {
        for (int i = tid; i < nt + 3; i += ICP_THREADS)
            lds_tgt[i] = stgt[i];
    }

    // ---- T0 = T_refIn_refMean^-1 * guess ; T_iter = I ----
    float T0[9];
    {
        const float Tinv[9] = {1, 0, -mx, 0, 1, -my, 0, 0, 1};
        float g[9];
#pragma unroll
        for (int i = 0; i < 9; ++i)
            g[i] = guess[i];
        mat3_mul(Tinv, g, T0);
#pragma unroll
        for (int i = 0; i < 9; ++i)
            T0[i] = sw_uniform(T0[i]);
    }
    IcpCheck chk = {S.hist_c, S.hist_s, S.hist_x, S.hist_y, 1, 0, 0};
    if (tid == 0) {
        const float I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        for (int i = 0; i < 9; ++i)
            S.Ti[i] = I[i];
        S.flag_iterate = 1;
        S.flag_status = SFE_ICP_OK;
        S.hist_c[0] = 1.0f; // DifferentialTransformationChecker::init pushes the identity
        S.hist_s[0] = 0.0f;
        S.hist_x[0] = 0.0f;
        S.hist_y[0] = 0.0f;
    }
    __syncthreads();

    const float r2_match = sw_uniform(f_mul(P.matcher_max_dist, P.matcher_max_dist));
    const float r2_filter = sw_uniform(f_mul(P.max_dist_filter, P.max_dist_filter));
    // A walk that has not met a target within maxDist yet is bounded by `best`, which starts at W2 = a
    // little MORE than maxDist^2 (found <=> best < r2m_up <=> best <= maxDist^2; with an unbounded
    // matcher: best finite): a query that ends `none` then knows its
    // nearest target is at least sqrt(best) > maxDist away, and that margin lets later iterations prove
    // "still none" from how far the query has moved instead of walking its whole maxDist window again.
    const float r2m_up = sw_uniform((r2_match < INFINITY) ? __uint_as_float(__float_as_uint(r2_match) + 1u) : INFINITY);
    const float W2 = sw_uniform(fmaxf(r2m_up, f_mul(r2_match, 1.1025f)));
    const float md_hi = sw_uniform(f_mul(P.matcher_max_dist, 1.00001f));
    // no pair beyond Cmax can get weight 1
    const float Cmax = P.use_max_dist_filter ? fminf(r2_filter, r2_match) : r2_match;
    float Cinit;
    {
        const float ext = f_add(T[nt].x, -T[1].x); // x extent of the sorted target
        const float h = 8.0f * ext / (float)nt;
        Cinit = h * h;
        if (!(Cinit > 1e-30f) || !(Cinit < Cmax))
            Cinit = Cmax;
        Cinit = sw_uniform(Cinit);
    }
    float Cnext = P.use_trimmed_filter ? Cinit : Cmax; // cap the next iteration starts with
    SW_PROF(0);

    int wd_outer = 0;
    bool use_cache = false; // from the second iteration on: Q.pos / Q.st hold the previous iteration's results
    while (true) {
        SW_WATCH(wd_outer, P.max_iter + 2, 0)
        float Ti[9];
#pragma unroll
        for (int i = 0; i < 9; ++i)
            Ti[i] = sw_uniform(S.Ti[i]);

        // ---- A+B: cur = Ti * (T0 * src); exact NN for every pair that can matter.  Round 0 walks
        // every query (lane i handles queries i, i + 1024, ...: transform, lower bound of cur.x in the
        // sorted x, two-sided walk of at most `sw_budget` trips), later rounds resume the suspended
        // walks with a 4x larger cap.  Walks that exhaust their budget are finished by an exhaustive
        // scan (tier 2), which yields their true neighbour. ----
        if (prof != nullptr && tid == 0)
            S.prof_b0 = clock64();
        float C = Cnext;
        unsigned nfin = 0, nexact = 0, ksel = 0;
        bool limit_inf = false;
        {
            int nwork = ns, cur = 0;
            for (int round = 0;; ++round) {
                if (round > 20) {
                    if (dbg)
                        atomicMax(dbg + 3, (int)blockIdx.x + 1);
                    break;
                }
                if (tid == 0) {
                    S.long_n = 0;
                    S.long_next = 0;
                    S.mid_n = 0;
                    S.wl_n[cur ^ 1] = 0;
                }
                __syncthreads();
                const int *wl = Q.wl[cur];
                int *wl_next = Q.wl[cur ^ 1];
                // -- tier 1: one lane per query, two passes.  Pass A gives every walk a short budget (most
                // finish: the typical window holds ~20 candidates); the rest is COMPACTED into dense
                // waves for pass B with the long budget, so lanes that finished early do not sit idle
                // through the long walks of their neighbours.  What still runs on goes to tier 2. --
                auto walk_pass = [&](const int *list, int n, bool fresh, int budget, bool last) {
                float2 sp_next = make_float2(0, 0); // fresh pass: the next slice's source point is fetched a slice ahead
                int prev_next = 0;                  // ... and so is last iteration's result of that query
                if (fresh && tid < n) {
                    sp_next = src[tid];
                    if (use_cache)
                        prev_next = Q.pos[tid];
                }
                for (int k0 = 0; k0 < n; k0 += ICP_THREADS) {
                    const int slot = k0 + tid;
                    const bool valid = slot < n;
                    const float2 sp_cur = sp_next;
                    const int prev = prev_next;
                    if (fresh && slot + ICP_THREADS < n) {
                        sp_next = src[slot + ICP_THREADS];
                        if (use_cache)
                            prev_next = Q.pos[slot + ICP_THREADS];
                    }
                    int q = 0, iL = 0, iR = 0, bpos = 0;
                    float px = 0, py = 0, best = W2;
                    bool tied = false;
                    if (valid) {
                        if (fresh) {
                            q = slot;
                            const float2 sp = sp_cur;
                            const float rx = affine1(T0[0], T0[1], T0[2], sp.x, sp.y);
                            const float ry = affine1(T0[3], T0[4], T0[5], sp.x, sp.y);
                            px = affine1(Ti[0], Ti[1], Ti[2], rx, ry);
                            py = affine1(Ti[3], Ti[4], Ti[5], rx, ry);
                            Q.xy[q] = make_float2(px, py);
                        } else {
                            q = list[slot];
                            const float2 p = Q.xy[q];
                            const int4 st = Q.st[q];
                            px = p.x;
                            py = p.y;
                            iL = st.x;
                            iR = st.y;
                            bpos = st.z & 0x7FFFFFFF;
                            tied = st.z < 0;
                            best = Q.d2[q];
                        }
                    }
                    if (fresh) { // first 1-based position whose x is not < px (convergent: 14 trips)
                        int lo = 1, hi = nt + 1;
                        while (__ballot(lo < hi)) {
                            const int mid = (lo + hi) >> 1;
                            const bool lt = T[min(mid, nt + 1)].x < px;
                            if (lo < hi) {
                                if (lt)
                                    lo = mid + 1;
                                else
                                    hi = mid;
                            }
                        }
                        iR = lo;
                        iL = lo - 1;
                    }
                    // What the previous iteration knew about this query (the cloud moves little between
                    // iterations).  Its neighbour -- exact or not -- is evaluated first as a WITNESS: a real
                    // target at distance dw, so the walk is "found" at once and bounded by min(dw, C)
                    // instead of running on until it meets some target within maxDist.  The witness counts
                    // as evaluated; the walk skips it when the cursors reach it (`!= bpos` below).
                    // A `none` query stays none as long as it has moved less than its recorded clearance:
                    // |p - t| >= |p0 - t| - |p - p0| > maxDist for every target t (1e-5 relative slop on
                    // each term, two orders above the rounding of the fp32 distances involved).
                    bool skip = false;
                    if (fresh && use_cache && valid) {
                        const int w = prev >= 0 ? prev + 1 : (prev <= -3 ? -2 - prev : 0);
                        if (w) {
                            const float2 t = T[w];
                            const float dxw = f_add(px, -t.x), dyw = f_add(py, -t.y);
                            const float dw = f_add(f_mul(dxw, dxw), f_mul(dyw, dyw));
                            if (dw < best) {
                                best = dw;
                                bpos = w;
                            }
                        } else if (prev == SW_NONE) {
                            const int4 r = Q.st[q];
                            const float mx0 = f_add(px, -__int_as_float(r.x)), my0 = f_add(py, -__int_as_float(r.y));
                            const float mv = sqrtf(f_add(f_mul(mx0, mx0), f_mul(my0, my0)));
                            skip = f_mul(mv, 1.00001f) < __int_as_float(r.z); // NaN -> walk
                        }
                    }
                    bool fin = !valid || skip;
                    for (int trip = 0; trip < budget; ++trip) {
                        if (!fin) {
#pragma unroll
                            for (int s2 = 0; s2 < 2; ++s2) {
                                const float2 tl = T[iL], tr = T[iR];
                                const float dxl = f_add(px, -tl.x), el = f_mul(dxl, dxl);
                                const float dyl = f_add(py, -tl.y), dl = f_add(el, f_mul(dyl, dyl));
                                const float dxr = f_add(px, -tr.x), er = f_mul(dxr, dxr);
                                const float dyr = f_add(py, -tr.y), dr = f_add(er, f_mul(dyr, dyr));
                                const float capv = (best < r2m_up) ? C : best; // nothing within maxDist yet: only `best` bounds the walk
                                const float sb = best < capv ? best : capv;    // stop bound (no NaNs here: plain select)
                                const bool okl = el <= sb, okr = er <= sb;    // NaN sentinel -> false
                                // only consumed candidates (cursor moves past them) may update the state:
                                // nothing is ever evaluated twice, so `tied` flags real ties only
                                tied |= okl && (dl == best) && (iL != bpos);
                                if (okl && dl < best) {
                                    best = dl;
                                    bpos = iL;
                                }
                                tied |= okr && (dr == best) && (iR != bpos);
                                if (okr && dr < best) {
                                    best = dr;
                                    bpos = iR;
                                }
                                iL -= okl ? 1 : 0;
                                iR += okr ? 1 : 0;
                                fin = !(okl || okr);
                            }
                        }
                        if (!__ballot(!fin))
                            break;
                    }
                    // classify: none / exact / suspended (inexact) / long (budget exhausted)
                    const bool is_long = valid && !fin;
                    const bool found = best < r2m_up; // <=> some target with d2 <= maxDist^2 was met (best starts at W2 >= r2m_up)
                    const bool is_none = valid && fin && !found;
                    const bool is_exact = valid && fin && found && best <= C;
                    const bool is_susp = valid && fin && found && !(best <= C);
                    if (is_none) {
                        Q.d2[q] = INFINITY;
                        Q.pos[q] = SW_NONE;
                        if (!skip) // a full walk: every target is at least sqrt(best) away from (px, py)
                            Q.st[q] = make_int4(__float_as_int(px), __float_as_int(py),
                                                __float_as_int(f_add(f_mul(sqrtf(best), 0.99999f), -md_hi)), 0);
                    }
                    if (is_exact) {
                        if (tied)
                            bpos = sweep_resolve_tie(T, Q, px, py, best, iL, iR);
                        Q.d2[q] = best;
                        Q.pos[q] = bpos - 1;
                    }
                    if (is_susp) {
                        Q.d2[q] = best;
                        Q.pos[q] = SW_INEXACT_OF(bpos);
                        Q.st[q] = make_int4(iL, iR, bpos | (tied ? (int)0x80000000 : 0), 0);
                    }
                    { // wave-aggregated appends
                        const unsigned long long ms = __ballot(is_susp), ml = __ballot(is_long);
                        const unsigned long long below = (1ull << lane) - 1ull;
                        if (ms) {
                            int base = 0;
                            if (lane == 0)
                                base = atomicAdd(&S.wl_n[cur ^ 1], __popcll(ms));
                            base = __builtin_amdgcn_readfirstlane(base);
                            if (is_susp)
                                wl_next[base + __popcll(ms & below)] = q;
                        }
                        if (ml && last) {
                            int base = 0;
                            if (lane == 0)
                                base = atomicAdd(&S.long_n, __popcll(ml));
                            base = __builtin_amdgcn_readfirstlane(base);
                            if (is_long) {
                                const int e = base + __popcll(ml & below);
                                Q.longe[2 * e] = make_int4(q, iL, iR, bpos | (tied ? (int)0x80000000 : 0));
                                Q.longe[2 * e + 1] = make_int4(__float_as_int(px), __float_as_int(py),
                                                               __float_as_int(best), 0);
                            }
                        }
                        if (ml && !last) {
                            int base = 0;
                            if (lane == 0)
                                base = atomicAdd(&S.mid_n, __popcll(ml));
                            base = __builtin_amdgcn_readfirstlane(base);
                            if (is_long) {
                                Q.d2[q] = best;
                                Q.st[q] = make_int4(iL, iR, bpos | (tied ? (int)0x80000000 : 0), 0);
                                Q.mid[base + __popcll(ml & below)] = q;
                            }
                        }
                    }
                }
                };
                walk_pass(wl, nwork, round == 0, sw_budget_a, false);
                __syncthreads();
                walk_pass(Q.mid, S.mid_n, false, sw_budget, true);
                __syncthreads();
                SW_PROF(6);
                const int nlong = S.long_n;
                if (prof != nullptr && tid == 0) {
                    S.prof[9] += 1;
                    S.prof[10] += nlong;
                    S.prof[11] += nwork;
                }
                // -- tier 2: one wave per long walk, 32 candidates per side and trip; the walk's state
                // travels in the list entry and the next entry is fetched while this one is walked --
                if (nlong > 0) {
                    const int wave = tid >> 6;
                    const bool left = lane < 32;
                    // walks are handed out dynamically (their lengths differ by orders of magnitude): the first
                    // ICP_WAVES slots are the waves' own, further ones come from the LDS counter
                    (void)wave;
                    auto next_slot = [&]() {
                        int v = 0;
                        if (lane == 0)
                            v = atomicAdd(&S.long_next, 1);
                        return __builtin_amdgcn_readfirstlane(v);
                    };
                    int slot = next_slot();
                    int4 e0 = make_int4(0, 0, 0, 0), e1 = make_int4(0, 0, 0, 0);
                    if (slot < nlong) {
                        e0 = Q.longe[2 * slot];
                        e1 = Q.longe[2 * slot + 1];
                    }
                    while (slot < nlong) {
                        long long tp0 = 0;
                        if (prof != nullptr && tid == 0)
                            tp0 = clock64();
                        const int q = e0.x;
                        int iL = e0.y, iR = e0.z, bpos = e0.w & 0x7FFFFFFF;
                        bool tied = e0.w < 0;
                        const float px = __int_as_float(e1.x), py = __int_as_float(e1.y);
                        float best = __int_as_float(e1.z);
                        slot = next_slot();
                        if (slot < nlong) { // prefetch the next walk
                            e0 = Q.longe[2 * slot];
                            e1 = Q.longe[2 * slot + 1];
                        }
                        if (prof != nullptr && tid == 0) {
                            const long long t_ = clock64();
                            S.prof[13] += t_ - tp0;
                            tp0 = t_;
                        }
                        bool doneL = false, doneR = false;
                        constexpr int U = 4; // candidates per lane and trip: four independent LDS reads in flight
                        const int lo = left ? lane : lane - 32;
                        for (int guard = 0; guard <= nt / (32 * U) + 2; ++guard) { // bounded by construction
                            const float capv = (best < r2m_up) ? C : best;
                            const float sb = best < capv ? best : capv; // stop bound at the start of the trip
                            const bool on = left ? !doneL : !doneR;
                            float d = INFINITY;
                            int jb = 0, nL = 0, nR = 0;
                            bool eqf = false;
#pragma unroll
                            for (int u = 0; u < U; ++u) {
                                const int j = left ? max(iL - lo - 32 * u, 0) : min(iR + lo + 32 * u, nt + 1);
                                const float2 t = T[j];
                                const float dx = f_add(px, -t.x), e = f_mul(dx, dx);
                                const float dy = f_add(py, -t.y);
                                const float du = f_add(e, f_mul(dy, dy));
                                // consumed = within the stop bound (a prefix of each side: e is monotone
                                // outwards); only consumed candidates count, the cursors move past exactly those
                                const bool cons = on && (e <= sb);
                                const unsigned long long mc = __ballot(cons);
                                nL += __popcll(mc & 0xFFFFFFFFull);
                                nR += __popcll(mc >> 32);
                                const bool use = cons && j != bpos; // the witness is already accounted for
                                if (use && du < d) { // NaN never passes
                                    d = du;
                                    jb = j;
                                    eqf = false;
                                } else if (use && du == d && du < INFINITY) {
                                    eqf = true; // two of this lane's candidates at the same distance
                                }
                            }
                            float wmin = INFINITY;
                            if (__ballot(d <= best)) { // rare for far queries: only then pay for the wave reduction
                                wmin = wave_min(d); // d never holds a NaN (only `du < d` updates it)
                            }
                            if (wmin < best) {
                                const unsigned long long who = __ballot(d == wmin);
                                tied = __popcll(who) > 1 || __ballot(eqf && d == wmin) != 0;
                                const int first = __builtin_amdgcn_readfirstlane(__ffsll((long long)who) - 1);
                                bpos = __builtin_amdgcn_readlane(jb, first);
                                best = wmin;
                            } else if (wmin == best && wmin < INFINITY) {
                                tied = true;
                            }
                            iL -= nL;
                            iR += nR;
                            doneL |= nL < 32 * U;
                            doneR |= nR < 32 * U;
                            if (prof != nullptr && lane == 0)
                                atomicAdd((unsigned long long *)&S.prof[12], 1ull);
                            if (doneL && doneR)
                                break;
                        }
                        if (prof != nullptr && tid == 0) {
                            const long long t_ = clock64();
                            S.prof[14] += t_ - tp0;
                            tp0 = t_;
                        }
                        if (lane == 0) {
                            if (!(best < r2m_up)) {
                                Q.d2[q] = INFINITY;
                                Q.pos[q] = SW_NONE;
                                Q.st[q] = make_int4(__float_as_int(px), __float_as_int(py),
                                                    __float_as_int(f_add(f_mul(sqrtf(best), 0.99999f), -md_hi)), 0);
                            } else if (best <= C) {
                                if (tied)
                                    bpos = sweep_resolve_tie(T, Q, px, py, best, iL, iR);
                                Q.d2[q] = best;
                                Q.pos[q] = bpos - 1;
                            } else {
                                Q.d2[q] = best;
                                Q.pos[q] = SW_INEXACT_OF(bpos);
                                Q.st[q] = make_int4(iL, iR, bpos | (tied ? (int)0x80000000 : 0), 0);
                                wl_next[atomicAdd(&S.wl_n[cur ^ 1], 1)] = q;
                            }
                        }
                        if (prof != nullptr && tid == 0)
                            S.prof[15] += clock64() - tp0;
                    }
                }
                __syncthreads();
                SW_PROF(7);
                // -- census: finite matches, and true neighbours within the cap --
                double cnt[2] = {0, 0};
                for (int base = 0; base < ns; base += SW_NQ * ICP_THREADS) { // SW_NQ loads in flight, not a chain
                    int pz[SW_NQ];
                    float dz[SW_NQ];
#pragma unroll
                    for (int k = 0; k < SW_NQ; ++k) {
                        const int i = base + k * ICP_THREADS + tid;
                        pz[k] = i < ns ? Q.pos[i] : SW_NONE;
                        dz[k] = i < ns ? Q.d2[i] : INFINITY;
                    }
#pragma unroll
                    for (int k = 0; k < SW_NQ; ++k) {
                        cnt[0] += (pz[k] != SW_NONE) ? 1.0 : 0.0;
                        cnt[1] += (pz[k] >= 0 && dz[k] <= C) ? 1.0 : 0.0;
                    }
                }
                block_sum<2>(cnt, S.red);
                nfin = (unsigned)cnt[0];
                nexact = (unsigned)cnt[1];
                SW_PROF(8);
                const int nsusp = S.wl_n[cur ^ 1];
                bool done;
                if (P.use_trimmed_filter && nfin > 0) {
                    ksel = (P.trim_ratio >= 1.0f) ? nfin - 1 : (unsigned)f_mul((float)nfin, P.trim_ratio);
                    done = nexact > ksel;
                } else {
                    done = nsusp == 0;
                }
                if (!done && (C >= Cmax || nsusp == 0)) {
                    // the k-th finite distance exceeds MaxDist^2 (or every neighbour is already known)
                    limit_inf = C >= Cmax && nsusp != 0;
                    done = true;
                }
                if (done)
                    break;
                C = sw_uniform((round >= 12) ? Cmax : fminf(fmaxf(4.0f * C, Cinit), Cmax));
                cur ^= 1;
                nwork = nsusp;
            }
        }
        SW_PROF(2);
        if (prof != nullptr && tid == 0 && chk.iters < 32) {
            S.prof_it[2 * chk.iters] = clock64() - S.prof_b0;
            S.prof_it[2 * chk.iters + 1] = ((long long)__float_as_uint(C) << 32) | nexact;
        }

        // ---- C: TrimmedDistOutlierFilter limit: exact order statistic by radix select ----
        float limit = INFINITY;
        bool fail = false;
        if (P.use_trimmed_filter) {
            if (nfin == 0) {
                fail = true; // "no outlier to filter"
                if (tid == 0)
                    S.flag_status = SFE_ICP_NO_OUTLIER;
            } else if (!limit_inf) {
                if (tid == 0) {
                    S.sel_k = ksel;
                    S.sel_prefix = 0;
                }
                __syncthreads();
                for (int shift = 24; shift >= 0; shift -= 8) {
                    if (tid < 256)
                        S.hist[tid] = 0;
                    __syncthreads();
                    const unsigned prefix = S.sel_prefix;
                    const unsigned himask = (shift == 24) ? 0u : (0xFFFFFFFFu << (shift + 8));
                    auto tally = [&](int pz, float dz) { // called wave-uniformly
                        unsigned bin = 0xFFFFFFFFu;      // no contribution
                        if (pz >= 0) {
                            const unsigned u = __float_as_uint(dz); // d >= 0: bit pattern order == value order
                            if ((u & himask) == prefix)
                                bin = (u >> shift) & 255u;
                        }
                        if (shift == 24) {
                            // the exponent byte is the same for nearly every point: aggregate per wave
                            // instead of serialising 64 LDS atomics on one address
                            unsigned long long todo = __ballot(bin != 0xFFFFFFFFu);
                            int wd4 = 0;
                            while (todo) {
                                SW_WATCH(wd4, 64, 4)
                                const int leader = __builtin_amdgcn_readfirstlane(__ffsll((long long)todo) - 1);
                                const unsigned b = (unsigned)__builtin_amdgcn_readlane((int)bin, leader);
                                const unsigned long long same = __ballot(bin == b);
                                if (lane == leader)
                                    atomicAdd(&S.hist[b], (unsigned)__popcll(same));
                                todo &= ~same;
                            }
                        } else if (bin != 0xFFFFFFFFu) {
                            atomicAdd(&S.hist[bin], 1u);
                        }
                    };
                    for (int base = 0; base < ns; base += SW_NQ * ICP_THREADS) {
                        int pz[SW_NQ];
                        float dz[SW_NQ];
#pragma unroll
                        for (int k = 0; k < SW_NQ; ++k) {
                            const int i = base + k * ICP_THREADS + tid;
                            pz[k] = i < ns ? Q.pos[i] : SW_NONE;
                            dz[k] = i < ns ? Q.d2[i] : INFINITY;
                        }
#pragma unroll
                        for (int k = 0; k < SW_NQ; ++k)
                            tally(pz[k], dz[k]);
                    }
                    __syncthreads();
                    if (tid < 64) { // one wave: rank-in-histogram by shuffles instead of a 256-step serial walk
                        const unsigned k = S.sel_k;
                        const unsigned h0 = S.hist[4 * lane], h1 = S.hist[4 * lane + 1], h2 = S.hist[4 * lane + 2],
                                       h3 = S.hist[4 * lane + 3];
                        const unsigned tot = h0 + h1 + h2 + h3;
                        const unsigned incl = wave_inclusive_scan(tot);
                        const unsigned excl = incl - tot;
                        if (k >= excl && k < incl) { // exactly one lane
                            unsigned r = k - excl, b = 4 * lane;
                            if (r >= h0) {
                                r -= h0;
                                ++b;
                                if (r >= h1) {
                                    r -= h1;
                                    ++b;
                                    if (r >= h2) {
                                        r -= h2;
                                        ++b;
                                    }
                                }
                            }
                            S.sel_k = r;
                            S.sel_prefix = prefix | (b << shift);
                        }
                    }
                    __syncthreads();
                }
                limit = sw_uniform(__uint_as_float(S.sel_prefix));
            }
        }
        __syncthreads();
        if (fail)
            break;
        Cnext = P.use_trimmed_filter ? ((limit < Cmax) ? fmaxf(limit, Cinit * 0.0625f) : Cmax) : Cmax;
        SW_PROF(3);

        // ---- D: error minimiser sums over the kept pairs, in two halves of five accumulators: ten fp64
        // accumulators per lane do not fit the 64-VGPR budget next to the loop state (they spilled) ----
        auto sums = [&](auto lo_tag) {
            constexpr int LO = decltype(lo_tag)::value;
            double a5[5] = {0, 0, 0, 0, 0};
            for (int i = tid; i < ns; i += ICP_THREADS) {
                const int id = Q.pos[i];
                const float d = Q.d2[i];
                const bool ok = id >= 0 && (!P.use_max_dist_filter || d <= r2_filter) &&
                                (!P.use_trimmed_filter || d <= limit);
                if (!ok)
                    continue;
                const float2 p = Q.xy[i];
                const double px = p.x, py = p.y;
                const float2 q = T[id + 1];
                const double qx = q.x, qy = q.y;
                double t[10];
                t[0] = 1.0;
                if (P.minimizer == 0) {
                    t[1] = px;
                    t[2] = py;
                    t[3] = qx;
                    t[4] = qy;
                    t[5] = qx * px;
                    t[6] = qx * py;
                    t[7] = qy * px;
                    t[8] = qy * py;
                    t[9] = 0.0;
                } else {
                    const float2 n = snrm[id];
                    const double nx = n.x, ny = n.y;
                    const double a0 = px * ny - py * nx;
                    const double e = nx * (px - qx) + ny * (py - qy);
                    t[1] = a0 * a0;
                    t[2] = a0 * nx;
                    t[3] = a0 * ny;
                    t[4] = nx * nx;
                    t[5] = nx * ny;
                    t[6] = ny * ny;
                    t[7] = -(a0 * e);
                    t[8] = -(nx * e);
                    t[9] = -(ny * e);
                }
#pragma unroll
                for (int k = 0; k < 5; ++k)
                    a5[k] += t[LO + k];
            }
            block_sum<5>(a5, S.red);
            if (tid == 0) {
#pragma unroll
                for (int k = 0; k < 5; ++k)
                    S.acc[LO + k] = a5[k];
            }
        };
        sums(std::integral_constant<int, 0>());
        sums(std::integral_constant<int, 5>());
        SW_PROF(4);

        // ---- E: solve, compose, check (one lane) ----
        if (tid == 0) {
            int status, iterate;
            double acc[10];
            for (int i = 0; i < 10; ++i)
                acc[i] = S.acc[i];
            icp_solve_and_check(P, acc, Ti, S.Ti, chk, status, iterate);
            S.flag_status = status;
            S.flag_iterate = (status == SFE_ICP_OK) ? iterate : 0;
        }
        __syncthreads();
        SW_PROF(5);
        if (!S.flag_iterate)
            break;
        use_cache = sw_cache != 0;
    }

    if (tid == 0) {
        const int status = S.flag_status;
        float *To = T_out + 9 * (size_t)jb;
        if (status == SFE_ICP_OK) {
            const float Tfwd[9] = {1, 0, mx, 0, 1, my, 0, 0, 1};
            float Ti[9], tmp[9], res[9];
            for (int i = 0; i < 9; ++i)
                Ti[i] = S.Ti[i];
            mat3_mul(Ti, T0, tmp);
            mat3_mul(Tfwd, tmp, res);
            for (int i = 0; i < 9; ++i)
                To[i] = res[i];
        } else {
            for (int i = 0; i < 9; ++i) // pcl.cpp:203,207-210: T stays the guess
                To[i] = guess[i];
        }
        status_out[jb] = status;
        iters_out[jb] = chk.iters;
        if (prof != nullptr && jb == 0) {
            for (int i = 0; i < 16; ++i)
                prof[i] = S.prof[i];
            for (int i = 0; i < 64; ++i)
                prof[16 + i] = S.prof_it[i];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host side: job tables, scratch, two launches.  jobs4 = n_jobs x (src_start, n_src, tgt_start,
// n_tgt) in points.
// ---------------------------------------------------------------------------------------------
int sfe_icp_sweep_launch(sfe_ctx *ctx, const sfe_icp_params *p, const float *d_src, const float *d_tgt,
                         const int32_t *jobs4, const float *d_guess9, int n_jobs, float *d_T9, int32_t *d_status,
                         int32_t *d_iters)
{
    std::vector<SweepPrep> preps;
    std::vector<SweepJob> jobs((size_t)n_jobs);
    std::vector<int> ids_lds, ids_glb; // jobs whose target fits LDS / is read from HBM scratch
    std::map<std::pair<int, int>, int> seen; // many guesses on one pair share one prep
    long long toff = 0, qoff = 0, koff = 0;
    for (int j = 0; j < n_jobs; ++j) {
        const int32_t *q = jobs4 + 4 * (size_t)j;
        const auto key = std::make_pair((int)q[2], (int)q[3]);
        auto it = seen.find(key);
        if (it == seen.end()) {
            it = seen.emplace(key, (int)preps.size()).first;
            long long n2 = 0;
            if (q[3] > SW_TCAP) {
                n2 = 2;
                while (n2 < q[3])
                    n2 <<= 1;
            }
            preps.push_back({q[2], q[3], toff, koff});
            toff += q[3] + 4;
            koff += n2;
        }
        const SweepPrep &pr = preps[(size_t)it->second];
        jobs[(size_t)j] = {q[0], q[1], q[3], it->second, pr.off, qoff};
        qoff += q[1];
        (q[3] <= SW_TCAP ? ids_lds : ids_glb).push_back(j);
    }
    const int n_prep = (int)preps.size();
    SweepPrep *d_preps = (SweepPrep *)sfe_scratch(ctx, 12, sizeof(SweepPrep) * (size_t)n_prep);
    SweepJob *d_jobs = (SweepJob *)sfe_scratch(ctx, 13, sizeof(SweepJob) * (size_t)n_jobs);
    float2 *d_stgt = (float2 *)sfe_scratch(ctx, 14, sizeof(float2) * (size_t)toff);
    int *d_perm = (int *)sfe_scratch(ctx, 15, sizeof(int) * (size_t)toff);
    float2 *d_snrm = p->minimizer == 1 ? (float2 *)sfe_scratch(ctx, 16, sizeof(float2) * (size_t)toff) : nullptr;
    float *d_mean = (float *)sfe_scratch(ctx, 17, sizeof(float) * 2 * (size_t)n_prep + sizeof(int) * (size_t)n_jobs);
    int *d_ids = d_mean ? (int *)(d_mean + 2 * (size_t)n_prep) : nullptr;
    unsigned long long *d_gkeys = (unsigned long long *)sfe_scratch(ctx, 24, sizeof(unsigned long long) * (size_t)std::max(koff, 1LL));
    float2 *d_qxy = (float2 *)sfe_scratch(ctx, 18, sizeof(float2) * (size_t)qoff);
    int4 *d_qst = (int4 *)sfe_scratch(ctx, 19, sizeof(int4) * (size_t)qoff);
    int *d_qwl = (int *)sfe_scratch(ctx, 21, sizeof(int) * 3 * (size_t)qoff);
    int4 *d_qlong = (int4 *)sfe_scratch(ctx, 23, sizeof(int4) * 2 * (size_t)qoff);
    float *d_nn_d2 = (float *)sfe_scratch(ctx, 5, sizeof(float) * (size_t)qoff);
    int *d_nn_pos = (int *)sfe_scratch(ctx, 6, sizeof(int) * (size_t)qoff);
    if (!d_preps || !d_jobs || !d_stgt || !d_perm || (p->minimizer == 1 && !d_snrm) || !d_mean || !d_gkeys || !d_qxy || !d_qst || !d_qwl || !d_qlong ||
        !d_nn_d2 || !d_nn_pos)
        return SFE_ERR_HIP;
    SFE_HIP(ctx, hipMemcpyAsync(d_preps, preps.data(), sizeof(SweepPrep) * (size_t)n_prep, hipMemcpyHostToDevice,
                                ctx->stream));
    SFE_HIP(ctx, hipMemcpyAsync(d_jobs, jobs.data(), sizeof(SweepJob) * (size_t)n_jobs, hipMemcpyHostToDevice,
                                ctx->stream));
    const int n_lds = (int)ids_lds.size(), n_glb = (int)ids_glb.size();
    ids_lds.insert(ids_lds.end(), ids_glb.begin(), ids_glb.end()); // [LDS-resident jobs | HBM-resident jobs]
    SFE_HIP(ctx, hipMemcpyAsync(d_ids, ids_lds.data(), sizeof(int) * (size_t)n_jobs, hipMemcpyHostToDevice,
                                ctx->stream));
    // the pageable host vectors must stay alive until the copies have been consumed
    SFE_HIP(ctx, hipStreamSynchronize(ctx->stream));

    SFE_HIP(ctx, hipFuncSetAttribute((const void *)icp_sweep_prep_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)sizeof(PrepShared)));
    hipLaunchKernelGGL(icp_sweep_prep_kernel, dim3(n_prep), dim3(ICP_THREADS), sizeof(PrepShared), ctx->stream, *p,
                       d_preps, (const float2 *)d_tgt, d_stgt, d_perm, d_snrm, d_mean, d_gkeys);
    SFE_LAUNCH_CHECK(ctx);
    static const bool debug = getenv("SFE_ICP_DEBUG") != nullptr;
    const int sw_budget = getenv("SFE_SW_BUDGET") ? atoi(getenv("SFE_SW_BUDGET")) : SW_BUDGET;
    const int sw_budget_a = getenv("SFE_SW_BUDGET_A") ? atoi(getenv("SFE_SW_BUDGET_A")) : SW_BUDGET_A;
    const int sw_cache = getenv("SFE_SW_CACHE") ? atoi(getenv("SFE_SW_CACHE")) : 1; // 0: A/B without the witness / clearance cache
    int *d_dbg = nullptr;
    if (debug) {
        d_dbg = (int *)sfe_scratch(ctx, 22, sizeof(int) * 8);
        if (!d_dbg)
            return SFE_ERR_HIP;
        SFE_HIP(ctx, hipMemsetAsync(d_dbg, 0, sizeof(int) * 8, ctx->stream));
    }
    long long *d_prof = ctx->icp_prof ? (long long *)sfe_scratch(ctx, 20, sizeof(long long) * 80) : nullptr;
    const size_t ctl_bytes = (sizeof(SweepShared) + 15) & ~(size_t)15;
    if (n_lds) {
        const size_t smem = ctl_bytes + sizeof(float2) * (SW_TCAP + 4);
        // up to one job per CU the 128-VGPR build wins (no spills, measured +8 %); beyond that two
        // 64-VGPR workgroups per CU overlap each other's serial phases (measured +14 % at 512 jobs)
        if (n_lds <= ctx->n_cu) {
            SFE_HIP(ctx, hipFuncSetAttribute((const void *)icp_sweep_kernel<4, true>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            hipLaunchKernelGGL((icp_sweep_kernel<4, true>), dim3(n_lds), dim3(ICP_THREADS), smem, ctx->stream, *p, d_jobs,
                               d_ids, (const float2 *)d_src, d_guess9, d_stgt, d_perm, d_snrm, d_mean, d_qxy, d_qst, d_qwl,
                               d_qlong, d_nn_d2, d_nn_pos, d_T9, d_status, d_iters, d_prof, d_dbg, sw_budget, sw_budget_a, sw_cache);
        } else {
            SFE_HIP(ctx, hipFuncSetAttribute((const void *)icp_sweep_kernel<8, true>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            hipLaunchKernelGGL((icp_sweep_kernel<8, true>), dim3(n_lds), dim3(ICP_THREADS), smem, ctx->stream, *p, d_jobs,
                               d_ids, (const float2 *)d_src, d_guess9, d_stgt, d_perm, d_snrm, d_mean, d_qxy, d_qst, d_qwl,
                               d_qlong, d_nn_d2, d_nn_pos, d_T9, d_status, d_iters, d_prof, d_dbg, sw_budget, sw_budget_a, sw_cache);
        }
        SFE_LAUNCH_CHECK(ctx);
    }
    if (n_glb) {
        if (n_glb <= ctx->n_cu)
            hipLaunchKernelGGL((icp_sweep_kernel<4, false>), dim3(n_glb), dim3(ICP_THREADS), ctl_bytes, ctx->stream, *p,
                               d_jobs, d_ids + n_lds, (const float2 *)d_src, d_guess9, d_stgt, d_perm, d_snrm, d_mean,
                               d_qxy, d_qst, d_qwl, d_qlong, d_nn_d2, d_nn_pos, d_T9, d_status, d_iters, d_prof, d_dbg,
                               sw_budget, sw_budget_a, sw_cache);
        else
            hipLaunchKernelGGL((icp_sweep_kernel<8, false>), dim3(n_glb), dim3(ICP_THREADS), ctl_bytes, ctx->stream, *p,
                               d_jobs, d_ids + n_lds, (const float2 *)d_src, d_guess9, d_stgt, d_perm, d_snrm, d_mean,
                               d_qxy, d_qst, d_qwl, d_qlong, d_nn_d2, d_nn_pos, d_T9, d_status, d_iters, d_prof, d_dbg,
                               sw_budget, sw_budget_a, sw_cache);
        SFE_LAUNCH_CHECK(ctx);
    }
    if (debug) {
        int h[8];
        SFE_HIP(ctx, hipMemcpyAsync(h, d_dbg, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
        SFE_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (int i = 0; i < 8; ++i)
            if (h[i])
                fprintf(stderr, "sfe_icp_sweep: watchdog %d tripped (workgroup %d)\n", i, h[i] - 1);
    }
    if (d_prof) {
        SFE_HIP(ctx, hipMemcpyAsync(ctx->icp_prof_host, d_prof, sizeof(long long) * 80, hipMemcpyDeviceToHost,
                                    ctx->stream));
        SFE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return 0;
}

extern "C" int sfe_icp_get_profile(sfe_ctx *ctx, int enable, long long *cycles16)
{
    if (!ctx)
        return SFE_ERR_ARG;
    if (cycles16)
        for (int i = 0; i < 80; ++i)
            cycles16[i] = ctx->icp_prof_host[i];
    ctx->icp_prof = enable;
    return 0;
}
