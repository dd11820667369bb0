// ICP scan matching with an exact sorted-sweep nearest-neighbour search (default ICP path).
// Replaces bruce_slam/src/bruce_slam/cpp/pcl.cpp:198-212 (ICP.compute -> libpointmatcher chain of
// bruce_slam/config/icp.yaml:1-31); same chain, same decisions and same arithmetic as
// sfe_icp.hip's brute-force kernel (which stays for targets that do not fit LDS and as the A/B
// baseline) -- only the order in which candidate pairs are visited differs.
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
// Work per query drops from n_tgt pair evaluations to the points whose |dx| is within the
// query's own NN distance: ~10-400 instead of 5000 on sonar clouds.  Lanes then run out of work at
// very different times (outliers walk far), so a workgroup keeps one shared queue of queries in
// LDS and lanes that finish pull the next query (refill when >= 16 lanes of a wave idle).
//
// Mapping: prep kernel = one workgroup per distinct target: mean, centre, bitonic sort of
// (x-key, index) in LDS, sorted cloud + permutation to HBM scratch, PCA normals (k-NN by the
// same sweep) for point-to-plane.  Loop kernel = one workgroup per job: sorted target resident
// in LDS (<= 8192 points), all ICP iterations in one launch: transform + binary search
// (coalesced) -> sweep (dynamic) -> trimmed quantile by exact radix select -> fp64 reduction of
// the 9(+1) sums -> closed-form solve and checkers on one lane.
#include "sfe_icp_common.h"

#include <algorithm>
#include <map>
#include <utility>

#define SW_TCAP 8192   // target points resident in LDS
#define SW_REFILL 16   // idle lanes per wave that trigger a queue refill

struct SweepPrep {
    int tgt_start, n_tgt;
    long long off; // offset (points) of this target's slice of the sorted-cloud scratch
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
__global__ __launch_bounds__(ICP_THREADS, 4) void icp_sweep_prep_kernel(sfe_icp_params P,
                                                                        const SweepPrep *__restrict__ preps,
                                                                        const float2 *__restrict__ tgt_all,
                                                                        float2 *__restrict__ stgt_all,
                                                                        int *__restrict__ perm_all,
                                                                        float2 *__restrict__ snrm_all,
                                                                        float *__restrict__ mean_all)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    PrepShared &S = *reinterpret_cast<PrepShared *>(smem_raw);
    const SweepPrep J = preps[blockIdx.x];
    const int nt = J.n_tgt, tid = threadIdx.x;
    const float2 *__restrict__ tgt = tgt_all + J.tgt_start;
    float2 *__restrict__ stgt = stgt_all + J.off;
    int *__restrict__ perm = perm_all + J.off;

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
    unsigned long long *keys = S.buf;
    for (unsigned i = tid; i < n2; i += ICP_THREADS) {
        unsigned long long k = ~0ull;
        if (i < (unsigned)nt)
            k = ((unsigned long long)mono_key(f_add(tgt[i].x, -mx)) << 32) | i;
        keys[i] = k;
    }
    __syncthreads();
    bitonic_sort_lds(keys, n2);

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
            const unsigned long long key = keys[pos];
            id[k] = (int)(unsigned)(key & 0xFFFFFFFFu);
            v[k] = make_float2(mono_inv((unsigned)(key >> 32)), f_add(tgt[id[k]].y, -my));
        }
    }
    __syncthreads();
    float2 *s_tgt = reinterpret_cast<float2 *>(S.buf);
    const float qnan = __uint_as_float(0x7FC00000u);
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int pos = k * ICP_THREADS + tid;
        if (pos < nt) {
            s_tgt[pos + 1] = v[k];
            stgt[pos] = v[k];
            perm[pos] = id[k];
        }
    }
    if (tid == 0) {
        s_tgt[0] = make_float2(qnan, qnan);
        s_tgt[nt + 1] = make_float2(qnan, qnan);
    }
    __syncthreads();
    if (P.minimizer != 1)
        return;

    // ---- PCA normals of the centred target: K nearest incl. the point itself, ordered by
    // (d2, original index) exactly like the brute-force scan (sfe_icp.hip) ----
    float2 *__restrict__ snrm = snrm_all + J.off;
    const int K = min(min(P.normals_knn, ICP_KMAX), nt);
    for (int c = tid; c < nt; c += ICP_THREADS) {
        const float2 q = s_tgt[c + 1];
        float bd[ICP_KMAX];
        int bj[ICP_KMAX];
#pragma unroll
        for (int k = 0; k < ICP_KMAX; ++k) {
            bd[k] = INFINITY;
            bj[k] = 0;
        }
        float kth = INFINITY; // bd[K-1]
        int iL = c, iR = c + 1;  // 1-based positions: the point itself is the first right candidate
        auto consider = [&](float d, int j) {
            if (!(d <= kth) || d == INFINITY)
                return;
            int p = 0;
            bool eq = false;
#pragma unroll
            for (int k = 0; k < ICP_KMAX; ++k) {
                p += (k < K && bd[k] < d) ? 1 : 0;
                eq |= (k < K && bd[k] == d);
            }
            if (eq) { // ties: lower original index first
                const int o = perm[j - 1];
#pragma unroll
                for (int k = 0; k < ICP_KMAX; ++k)
                    if (k < K && bd[k] == d && perm[bj[k] - 1] < o)
                        ++p;
            }
            if (p >= K)
                return;
#pragma unroll
            for (int k = ICP_KMAX - 1; k >= 1; --k)
                if (k < K && k > p) {
                    bd[k] = bd[k - 1];
                    bj[k] = bj[k - 1];
                }
#pragma unroll
            for (int k = 0; k < ICP_KMAX; ++k) {
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
        for (int k = 0; k < ICP_KMAX; ++k)
            if (k < K) {
                const float2 t = s_tgt[bj[k]];
                sx += (double)t.x;
                sy += (double)t.y;
            }
        sx /= K;
        sy /= K;
        double a = 0, b = 0, d = 0;
#pragma unroll
        for (int k = 0; k < ICP_KMAX; ++k)
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

// ---------------------------------------------------------------------------------------------
// loop: one workgroup per job
// ---------------------------------------------------------------------------------------------
struct SweepShared {
    float2 tgt[SW_TCAP + 2];
    double red[ICP_WAVES * 10 + 10];
    unsigned hist[256];
    unsigned sel_prefix, sel_k;
    int qnext;
    int flag_iterate, flag_status;
    float Ti[9];
    float hist_c[ICP_MAX_HIST], hist_s[ICP_MAX_HIST], hist_x[ICP_MAX_HIST], hist_y[ICP_MAX_HIST];
    long long prof_t, prof[8];
};

#define SW_PROF(k)                                                                               \
    do {                                                                                         \
        if (prof != nullptr && threadIdx.x == 0) {                                               \
            const long long t_ = clock64();                                                      \
            S.prof[k] += t_ - S.prof_t;                                                          \
            S.prof_t = t_;                                                                       \
        }                                                                                        \
    } while (0)

template <int MINW>
__global__ __launch_bounds__(ICP_THREADS, MINW) void icp_sweep_kernel(
    sfe_icp_params P, const SweepJob *__restrict__ jobs, const float2 *__restrict__ src_all,
    const float *__restrict__ guess_all, const float2 *__restrict__ stgt_all, const int *__restrict__ perm_all,
    const float2 *__restrict__ snrm_all, const float *__restrict__ mean_all, float2 *__restrict__ q_xy_all,
    int *__restrict__ q_start_all, float *__restrict__ nn_d2_all, int *__restrict__ nn_pos_all,
    float *__restrict__ T_out, int *__restrict__ status_out, int *__restrict__ iters_out, long long *prof)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    SweepShared &S = *reinterpret_cast<SweepShared *>(smem_raw);

    const SweepJob J = jobs[blockIdx.x];
    const int ns = J.n_src, nt = J.n_tgt;
    const float2 *__restrict__ src = src_all + J.src_start;
    const float2 *__restrict__ stgt = stgt_all + J.tgt_off;
    const int *__restrict__ perm = perm_all + J.tgt_off;
    const float2 *__restrict__ snrm = snrm_all ? snrm_all + J.tgt_off : nullptr;
    float2 *__restrict__ q_xy = q_xy_all + J.q_off;
    int *__restrict__ q_start = q_start_all + J.q_off;
    float *__restrict__ nn_d2 = nn_d2_all + J.q_off;
    int *__restrict__ nn_pos = nn_pos_all + J.q_off;
    const float *guess = guess_all + 9 * (size_t)blockIdx.x;
    const int tid = threadIdx.x, lane = threadIdx.x & 63;
    const float mx = mean_all[2 * J.prep], my = mean_all[2 * J.prep + 1];

    if (prof != nullptr && tid == 0) {
        for (int i = 0; i < 8; ++i)
            S.prof[i] = 0;
        S.prof_t = clock64();
    }
    // sorted centred target -> LDS, NaN sentinels at both ends (a NaN stops a sweep direction)
    {
        const float qnan = __uint_as_float(0x7FC00000u);
        for (int i = tid; i < nt; i += ICP_THREADS)
            S.tgt[i + 1] = stgt[i];
        if (tid == 0) {
            S.tgt[0] = make_float2(qnan, qnan);
            S.tgt[nt + 1] = make_float2(qnan, qnan);
        }
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

    const float r2_match = f_mul(P.matcher_max_dist, P.matcher_max_dist);
    const float r2_filter = f_mul(P.max_dist_filter, P.max_dist_filter);
    SW_PROF(0);

    while (true) {
        float Ti[9];
#pragma unroll
        for (int i = 0; i < 9; ++i)
            Ti[i] = S.Ti[i];

        // ---- A: cur = Ti * (T0 * src), start position = lower bound of cur.x in the sorted x ----
        for (int i = tid; i < ns; i += ICP_THREADS) {
            const float2 s = src[i];
            const float rx = affine1(T0[0], T0[1], T0[2], s.x, s.y);
            const float ry = affine1(T0[3], T0[4], T0[5], s.x, s.y);
            const float px = affine1(Ti[0], Ti[1], Ti[2], rx, ry);
            const float py = affine1(Ti[3], Ti[4], Ti[5], rx, ry);
            int lo = 1, hi = nt + 1; // first 1-based position whose x is not < px
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (S.tgt[mid].x < px)
                    lo = mid + 1;
                else
                    hi = mid;
            }
            q_xy[i] = make_float2(px, py);
            q_start[i] = lo;
        }
        if (tid == 0)
            S.qnext = 0;
        __syncthreads();
        SW_PROF(1);

        // ---- B: exact 1-NN by the two-sided sweep; lanes pull queries from the workgroup queue ----
        double nfin_d[1] = {0};
        {
            bool active = false, more = true, tied = false;
            float px = 0, py = 0, best = INFINITY;
            int bpos = 0, iL = 0, iR = 0, myq = 0;
            while (true) {
                const unsigned long long im = __ballot(!active);
                if (im) {
                    if (more && (__popcll(im) >= SW_REFILL || im == ~0ull)) {
                        const int cnt = __popcll(im);
                        int base = 0;
                        if (lane == 0)
                            base = atomicAdd(&S.qnext, cnt);
                        base = __builtin_amdgcn_readfirstlane(base);
                        more = base + cnt < ns;
                        if (!active) {
                            const int q = base + __popcll(im & ((1ull << lane) - 1ull));
                            if (q < ns) {
                                const float2 p = q_xy[q];
                                const int st = q_start[q];
                                px = p.x;
                                py = p.y;
                                iR = st;
                                iL = st - 1;
                                best = INFINITY;
                                bpos = 0;
                                tied = false;
                                myq = q;
                                active = true;
                            }
                        }
                    } else if (im == ~0ull) {
                        break;
                    }
                }
                if (active) {
                    bool fin = false;
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const float2 tl = S.tgt[iL], tr = S.tgt[iR];
                        const float dxl = f_add(px, -tl.x), el = f_mul(dxl, dxl);
                        const float dyl = f_add(py, -tl.y), dl = f_add(el, f_mul(dyl, dyl));
                        const float dxr = f_add(px, -tr.x), er = f_mul(dxr, dxr);
                        const float dyr = f_add(py, -tr.y), dr = f_add(er, f_mul(dyr, dyr));
                        const bool okl = el <= best, okr = er <= best; // NaN sentinel -> false
                        tied |= (dl == best);
                        if (dl < best) {
                            best = dl;
                            bpos = iL;
                        }
                        tied |= (dr == best);
                        if (dr < best) {
                            best = dr;
                            bpos = iR;
                        }
                        iL -= okl ? 1 : 0;
                        iR += okr ? 1 : 0;
                        fin = !(okl || okr);
                    }
                    if (fin) {
                        if (tied && best < INFINITY) {
                            // some candidate tied with a running best: walk the final window again and
                            // take the lowest original index among the points at distance `best`
                            const int st = q_start[myq];
                            int bo = 0x7FFFFFFF, bp = 0;
                            for (int j = st - 1; j >= 1; --j) {
                                const float2 t = S.tgt[j];
                                const float dx = f_add(px, -t.x), e = f_mul(dx, dx);
                                if (!(e <= best))
                                    break;
                                const float dy = f_add(py, -t.y);
                                if (f_add(e, f_mul(dy, dy)) == best) {
                                    const int o = perm[j - 1];
                                    if (o < bo) {
                                        bo = o;
                                        bp = j;
                                    }
                                }
                            }
                            for (int j = st; j <= nt; ++j) {
                                const float2 t = S.tgt[j];
                                const float dx = f_add(px, -t.x), e = f_mul(dx, dx);
                                if (!(e <= best))
                                    break;
                                const float dy = f_add(py, -t.y);
                                if (f_add(e, f_mul(dy, dy)) == best) {
                                    const int o = perm[j - 1];
                                    if (o < bo) {
                                        bo = o;
                                        bp = j;
                                    }
                                }
                            }
                            bpos = bp;
                        }
                        float d = best;
                        int id = bpos - 1;
                        if (bpos <= 0 || !(d <= r2_match)) {
                            id = -1;
                            d = INFINITY;
                        } else {
                            nfin_d[0] += 1.0;
                        }
                        nn_d2[myq] = d;
                        nn_pos[myq] = id;
                        active = false;
                    }
                }
            }
        }
        block_sum<1>(nfin_d, S.red); // also orders the nn_d2 / nn_pos stores before the re-reads below
        const unsigned nfin = (unsigned)nfin_d[0];
        SW_PROF(2);

        // ---- C: TrimmedDistOutlierFilter limit: exact order statistic by radix select ----
        float limit = INFINITY;
        bool fail = false;
        if (P.use_trimmed_filter) {
            if (nfin == 0) {
                fail = true; // "no outlier to filter"
                if (tid == 0)
                    S.flag_status = SFE_ICP_NO_OUTLIER;
            } else if (P.trim_ratio >= 1.0f) {
                if (tid == 0)
                    S.sel_k = nfin - 1; // max of the finite distances
            } else if (tid == 0) {
                S.sel_k = (unsigned)f_mul((float)nfin, P.trim_ratio); // values.size()*quantile in float
            }
            if (!fail) {
                if (tid == 0)
                    S.sel_prefix = 0;
                __syncthreads();
                for (int shift = 24; shift >= 0; shift -= 8) {
                    if (tid < 256)
                        S.hist[tid] = 0;
                    __syncthreads();
                    const unsigned prefix = S.sel_prefix;
                    const unsigned himask = (shift == 24) ? 0u : (0xFFFFFFFFu << (shift + 8));
                    for (int i0 = 0; i0 < ns; i0 += ICP_THREADS) {
                        const int i = i0 + tid;
                        unsigned bin = 0xFFFFFFFFu; // no contribution
                        if (i < ns) {
                            const float d = nn_d2[i];
                            const unsigned u = __float_as_uint(d); // d >= 0: bit pattern order == value order
                            if (d != INFINITY && (u & himask) == prefix)
                                bin = (u >> shift) & 255u;
                        }
                        if (shift == 24) {
                            // the exponent byte is the same for nearly every point: aggregate per wave
                            // instead of serialising 64 LDS atomics on one address
                            unsigned long long todo = __ballot(bin != 0xFFFFFFFFu);
                            while (todo) {
                                const int leader = __ffsll((long long)todo) - 1;
                                const unsigned b = (unsigned)__builtin_amdgcn_readlane((int)bin, leader);
                                const unsigned long long same = __ballot(bin == b);
                                if (lane == leader)
                                    atomicAdd(&S.hist[b], (unsigned)__popcll(same));
                                todo &= ~same;
                            }
                        } else if (bin != 0xFFFFFFFFu) {
                            atomicAdd(&S.hist[bin], 1u);
                        }
                    }
                    __syncthreads();
                    if (tid < 64) { // one wave: rank-in-histogram by shuffles instead of a 256-step serial walk
                        const unsigned k = S.sel_k;
                        unsigned h0 = S.hist[4 * lane], h1 = S.hist[4 * lane + 1], h2 = S.hist[4 * lane + 2],
                                 h3 = S.hist[4 * lane + 3];
                        const unsigned tot = h0 + h1 + h2 + h3;
                        unsigned incl = tot;
#pragma unroll
                        for (int d = 1; d < 64; d <<= 1) {
                            const unsigned o = __shfl_up(incl, d);
                            if (lane >= d)
                                incl += o;
                        }
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
                limit = __uint_as_float(S.sel_prefix);
            }
        }
        __syncthreads();
        if (fail)
            break;
        SW_PROF(3);

        // ---- D: error minimiser sums over the kept pairs ----
        double acc[10];
#pragma unroll
        for (int i = 0; i < 10; ++i)
            acc[i] = 0;
        for (int i = tid; i < ns; i += ICP_THREADS) {
            const int id = nn_pos[i];
            const float d = nn_d2[i];
            const bool ok = id >= 0 && (!P.use_max_dist_filter || d <= r2_filter) &&
                            (!P.use_trimmed_filter || d <= limit);
            if (!ok)
                continue;
            const float2 p = q_xy[i];
            const double px = p.x, py = p.y;
            const float2 q = S.tgt[id + 1];
            const double qx = q.x, qy = q.y;
            acc[0] += 1.0;
            if (P.minimizer == 0) {
                acc[1] += px;
                acc[2] += py;
                acc[3] += qx;
                acc[4] += qy;
                acc[5] += qx * px;
                acc[6] += qx * py;
                acc[7] += qy * px;
                acc[8] += qy * py;
            } else {
                const float2 n = snrm[id];
                const double nx = n.x, ny = n.y;
                const double a0 = px * ny - py * nx;
                const double e = nx * (px - qx) + ny * (py - qy);
                acc[1] += a0 * a0;
                acc[2] += a0 * nx;
                acc[3] += a0 * ny;
                acc[4] += nx * nx;
                acc[5] += nx * ny;
                acc[6] += ny * ny;
                acc[7] -= a0 * e;
                acc[8] -= nx * e;
                acc[9] -= ny * e;
            }
        }
        block_sum<10>(acc, S.red);
        SW_PROF(4);

        // ---- E: solve, compose, check (one lane) ----
        if (tid == 0) {
            int status, iterate;
            icp_solve_and_check(P, acc, Ti, S.Ti, chk, status, iterate);
            S.flag_status = status;
            S.flag_iterate = (status == SFE_ICP_OK) ? iterate : 0;
        }
        __syncthreads();
        SW_PROF(5);
        if (!S.flag_iterate)
            break;
    }

    if (tid == 0) {
        const int status = S.flag_status;
        float *To = T_out + 9 * (size_t)blockIdx.x;
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
        status_out[blockIdx.x] = status;
        iters_out[blockIdx.x] = chk.iters;
        if (prof != nullptr && blockIdx.x == 0)
            for (int i = 0; i < 8; ++i)
                prof[i] = S.prof[i];
    }
}

// ---------------------------------------------------------------------------------------------
// host side: job tables, scratch, two launches.  jobs4 = n_jobs x (src_start, n_src, tgt_start,
// n_tgt) in points.  Returns 1 if some target is too large for the sweep (caller falls back).
// ---------------------------------------------------------------------------------------------
int sfe_icp_sweep_launch(sfe_ctx *ctx, const sfe_icp_params *p, const float *d_src, const float *d_tgt,
                         const int32_t *jobs4, const float *d_guess9, int n_jobs, float *d_T9, int32_t *d_status,
                         int32_t *d_iters)
{
    for (int j = 0; j < n_jobs; ++j)
        if (jobs4[4 * (size_t)j + 3] > SW_TCAP)
            return 1;
    std::vector<SweepPrep> preps;
    std::vector<SweepJob> jobs((size_t)n_jobs);
    std::map<std::pair<int, int>, int> seen; // many guesses on one pair share one prep
    long long toff = 0, qoff = 0;
    for (int j = 0; j < n_jobs; ++j) {
        const int32_t *q = jobs4 + 4 * (size_t)j;
        const auto key = std::make_pair((int)q[2], (int)q[3]);
        auto it = seen.find(key);
        if (it == seen.end()) {
            it = seen.emplace(key, (int)preps.size()).first;
            preps.push_back({q[2], q[3], toff});
            toff += q[3];
        }
        const SweepPrep &pr = preps[(size_t)it->second];
        jobs[(size_t)j] = {q[0], q[1], q[3], it->second, pr.off, qoff};
        qoff += q[1];
    }
    const int n_prep = (int)preps.size();
    SweepPrep *d_preps = (SweepPrep *)sfe_scratch(ctx, 12, sizeof(SweepPrep) * (size_t)n_prep);
    SweepJob *d_jobs = (SweepJob *)sfe_scratch(ctx, 13, sizeof(SweepJob) * (size_t)n_jobs);
    float2 *d_stgt = (float2 *)sfe_scratch(ctx, 14, sizeof(float2) * (size_t)toff);
    int *d_perm = (int *)sfe_scratch(ctx, 15, sizeof(int) * (size_t)toff);
    float2 *d_snrm = p->minimizer == 1 ? (float2 *)sfe_scratch(ctx, 16, sizeof(float2) * (size_t)toff) : nullptr;
    float *d_mean = (float *)sfe_scratch(ctx, 17, sizeof(float) * 2 * (size_t)n_prep);
    float2 *d_qxy = (float2 *)sfe_scratch(ctx, 18, sizeof(float2) * (size_t)qoff);
    int *d_qstart = (int *)sfe_scratch(ctx, 19, sizeof(int) * (size_t)qoff);
    float *d_nn_d2 = (float *)sfe_scratch(ctx, 5, sizeof(float) * (size_t)qoff);
    int *d_nn_pos = (int *)sfe_scratch(ctx, 6, sizeof(int) * (size_t)qoff);
    if (!d_preps || !d_jobs || !d_stgt || !d_perm || (p->minimizer == 1 && !d_snrm) || !d_mean || !d_qxy || !d_qstart ||
        !d_nn_d2 || !d_nn_pos)
        return SFE_ERR_HIP;
    SFE_HIP(ctx, hipMemcpyAsync(d_preps, preps.data(), sizeof(SweepPrep) * (size_t)n_prep, hipMemcpyHostToDevice,
                                ctx->stream));
    SFE_HIP(ctx, hipMemcpyAsync(d_jobs, jobs.data(), sizeof(SweepJob) * (size_t)n_jobs, hipMemcpyHostToDevice,
                                ctx->stream));
    // the pageable host vectors must stay alive until the copies have been consumed
    SFE_HIP(ctx, hipStreamSynchronize(ctx->stream));

    SFE_HIP(ctx, hipFuncSetAttribute((const void *)icp_sweep_prep_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)sizeof(PrepShared)));
    hipLaunchKernelGGL(icp_sweep_prep_kernel, dim3(n_prep), dim3(ICP_THREADS), sizeof(PrepShared), ctx->stream, *p,
                       d_preps, (const float2 *)d_tgt, d_stgt, d_perm, d_snrm, d_mean);
    SFE_LAUNCH_CHECK(ctx);
    long long *d_prof = ctx->icp_prof ? (long long *)sfe_scratch(ctx, 20, sizeof(long long) * 8) : nullptr;
    SFE_HIP(ctx, hipFuncSetAttribute((const void *)icp_sweep_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)sizeof(SweepShared)));
    hipLaunchKernelGGL(icp_sweep_kernel<8>, dim3(n_jobs), dim3(ICP_THREADS), sizeof(SweepShared), ctx->stream, *p,
                       d_jobs, (const float2 *)d_src, d_guess9, d_stgt, d_perm, d_snrm, d_mean, d_qxy, d_qstart,
                       d_nn_d2, d_nn_pos, d_T9, d_status, d_iters, d_prof);
    SFE_LAUNCH_CHECK(ctx);
    if (d_prof) {
        SFE_HIP(ctx, hipMemcpyAsync(ctx->icp_prof_host, d_prof, sizeof(long long) * 8, hipMemcpyDeviceToHost,
                                    ctx->stream));
        SFE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return 0;
}

extern "C" int sfe_icp_get_profile(sfe_ctx *ctx, int enable, long long *cycles8)
{
    if (!ctx)
        return SFE_ERR_ARG;
    if (cycles8)
        for (int i = 0; i < 8; ++i)
            cycles8[i] = ctx->icp_prof_host[i];
    ctx->icp_prof = enable;
    return 0;
}
