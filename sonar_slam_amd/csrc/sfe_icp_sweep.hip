// ICP scan matching with an exact strip-sweep nearest-neighbour search (default ICP path).
// Replaces bruce_slam/src/bruce_slam/cpp/pcl.cpp:198-212 (ICP.compute -> libpointmatcher chain of
// bruce_slam/config/icp.yaml:1-31); same chain, same decisions and same arithmetic as
// sfe_icp.hip's brute-force kernel (which stays as the A/B
// baseline and checker) -- only the order in which candidate pairs are visited differs.
//
// Why a sweep is exact.  The squared distance everyone on this path compares is
//     d2 = fl( fl(dx*dx) + fl(dy*dy) ),  dx = fl(px - tx), dy = fl(py - ty)      (dist2())
// Rounding is monotone, so d2 >= fl(dx*dx) =: e and d2 >= fl(dy*dy), and both grow with |dx|, |dy|.
// The centred target is cut into horizontal STRIPS (uniform y intervals, <= 64 of them) and sorted by
// x inside each strip.  A query visits its own strip, then the strips above, then the strips below:
//   * a strip (and every strip beyond it) is skipped once fl(ylb*ylb) > bound, ylb = distance from the
//     query's y to the nearest y any point of those strips has (min / max taken from the data, so no
//     cell-boundary rounding enters);
//   * inside a strip the query walks outwards from its own x position in both directions and stops a
//     direction as soon as e > bound.
// Whatever is skipped has d2 > bound and can neither win nor tie.  All surviving candidates are
// evaluated with dist2()'s exact expression; ties go to the lowest ORIGINAL target index (what the
// brute-force scan and the oracle do), resolved by a rare second pass over the final window.  No
// kd-tree, no approximation, no float re-association: match ids and d2 are bit-identical to brute
// force.
//
// Work per query drops from n_tgt pair evaluations to the points inside a (2r x strip height) box per
// visited strip: ~4 instead of 5000 on converged sonar clouds, a few dozen while the clouds are still
// far apart (a single x-sorted sweep -- the first version of this file -- needed ~17 and ~200: a wall
// along y puts its whole length into one x window).  Walks still differ in length, so the search is
// tiered (details at the loop kernel): own strip with a short budget for every lane -> survivors
// compacted into dense waves that go through all their strips -> what still runs is finished by a whole
// wave, 256 candidates per trip.
//
// Mapping: prep kernel = one workgroup per distinct target (many guesses on one pair share it):
// mean, centre, strip table, bitonic sort of (strip, x-key, index) in LDS (HBM scratch beyond 8192
// points), sorted cloud + permutation to HBM scratch, PCA normals (k-NN by the same strip sweep) for
// point-to-plane.  Loop kernel = one workgroup per job, all ICP iterations in one launch: sorted target
// resident in LDS (or walked through L2 beyond 8192 points), per iteration: transform + capped walks
// (tiers) -> census -> trimmed quantile by exact radix select -> fp64 reduction of the 9(+1) sums ->
// closed-form solve and checkers on one lane.
#include "sfe_icp_common.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <type_traits>
#include <utility>

#define SW_TCAP 8192   // target points resident in LDS
#define SW_NS_MAX 64   // strips per target
#define SW_PAD (SW_NS_MAX + 4) // sentinels: one in front, one behind every strip, two spare behind the last
#define SW_GRID_MAX 8192 // most cells a target's witness grid can have (built in the prep kernel's LDS: 4 B per cell)
// Job tiers (the scan matches bruce_slam itself produces are 10^2..10^3 points, slam.py:769,1032): the
// same kernels instantiated for one-wave and four-wave workgroups, several jobs per CU, chosen by the launcher from
// n_src / n_tgt.  {threads, target capacity of the prep kernel's LDS, witness-grid cells}
#define SW_T0_NT 64
#define SW_T0_TCAP 512
#define SW_T0_GRID 512
#define SW_T1_NT 256
#define SW_T1_TCAP 2048
#define SW_T1_GRID 2048
// Many-to-one batches on large clouds (BASELINE configs[4]: 30 guesses x one 20 000 x 20 000 pair, slam.py:346-358): a
// job is split over up to SW_MG_MAX workgroups (queries by strip band), which meet in a per-job sync area
#define SW_MG_MAX 16
#define SW_MG_WORDS 272 // 32-bit words one exchange can carry (a 256-bin histogram + scalars)

// Strip table of one target (built by the prep kernel, read by every job on that target).
// Sorted-cloud layout (float2 positions): [0] NaN, then for every strip s its points ascending in x
// followed by one NaN sentinel; sbeg[s] = position of the first point of strip s, its points are
// [sbeg[s], sbeg[s+1] - 1), the sentinel behind them sits at sbeg[s+1] - 1 (and is the sentinel in front
// of strip s+1); len = sbeg[ns] = n_tgt + ns + 1, positions len and len+1 hold two more NaNs.  perm /
// snrm use the same positions: entry p-1 belongs to position p.
struct StripTab {
    int ns, len;
    float ylo, inv_g;         // strip(y) = clamp(int((y - ylo) * inv_g), 0, ns - 1)
    float ext_x;              // x extent of the finite points (initial cap of the search)
    // witness grid (iteration 0): cell (ix, iy) = clamp(int((x - gx0) * ginv)), clamp(int((y - gy0) * ginv));
    // grid[iy * gnx + ix] = sorted position of the target point nearest to the cell's centre (0: none)
    float gx0, gy0, ginv;
    int gnx, gny;
    int grid_off64; // this target's slice of the witness-grid scratch starts at int 64 * grid_off64
    int pad_;
    int sbeg[SW_NS_MAX + 1];
    float smin[SW_NS_MAX];    // smallest y of any point in strips >= s (+inf if none)
    float smax[SW_NS_MAX];    // largest y of any point in strips <= s (-inf if none)
};

struct SweepPrep {
    int tgt_start, n_tgt, ns, pad_;
    long long off;     // offset (points) of this target's slice of the sorted-cloud scratch (stride n_tgt + SW_PAD)
    long long key_off; // targets beyond the LDS capacity: offset of their sort keys in HBM scratch
    long long grid_off; // offset (ints, a multiple of 64) of its witness grid
};

struct SweepJob {
    int src_start, n_src, n_tgt, prep;
    long long tgt_off; // = SweepPrep.off of its target
    long long q_off;   // offset (points) of this job's slice of the per-query scratch
    int out;           // index of the caller's job (guess, T_out, status, iterations) this record works for
    int grp, ngrp;     // split jobs: this record is share `grp` of `ngrp` (1: the whole job)
    int sync;          // ... and their sync area is number `sync`
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

__device__ __forceinline__ int strip_of(float y, float ylo, float inv_g, int ns)
{
    float v = f_mul(f_add(y, -ylo), inv_g);
    v = fminf(fmaxf(v, 0.0f), (float)(ns - 1)); // NaN -> 0
    return (int)v;
}

// Sorted target of a job whose cloud does not fit LDS (beyond SW_TCAP points): positions [lo, lo + n) -- the strips its
// queries live in and as many around them as the LDS holds -- are read from LDS, everything else from the HBM scratch
// copy (through L2).  A walk is a chain of dependent reads: one LDS latency per step instead of one L2 round trip.
struct TgtWin {
    const float2 *g; // the whole sorted cloud (HBM scratch)
    const float2 *l; // LDS copy of positions [lo, lo + n)
    int lo;
    unsigned n;
    __device__ __forceinline__ float2 operator[](int j) const
    {
        const unsigned o = (unsigned)(j - lo);
        return o < n ? l[o] : g[j];
    }
};

// first position in [lo, hi) whose x is not < px (hi if there is none; NaN x counts as "not <").
// Convergent form: every lane of the wave must call it, lanes without work pass lo == hi.
template <class TV>
__device__ __forceinline__ int strip_lower_bound(const TV &T, int lo, int hi, float px)
{
    while (__ballot(lo < hi)) {
        const int mid = (lo + hi) >> 1;
        const bool lt = T[mid].x < px;
        if (lo < hi) {
            if (lt)
                lo = mid + 1;
            else
                hi = mid;
        }
    }
    return lo;
}

// the same for one lane on its own (rare paths)
template <class TV>
__device__ __forceinline__ int strip_lower_bound_lane(const TV &T, int lo, int hi, float px)
{
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (T[mid].x < px)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

// Next strip of a search that has covered strips [so, s_up) upwards and (s_dn, so) downwards: upwards
// first (the own strip `so` is the first of them), then downwards; -1 when every remaining strip is
// farther than `sb` in y.  smin / smax are monotone in s, so a direction that is pruned once stays pruned.
__device__ __forceinline__ int next_strip(const StripTab &tab, int ns, int so, int &s_up, int &s_dn, float py, float sb)
{
    bool upok = s_up < ns;
    if (upok && s_up != so) {
        const float yl = f_add(tab.smin[s_up], -py);
        upok = !(yl > 0.0f && f_mul(yl, yl) > sb);
    }
    if (!upok)
        s_up = ns;
    bool dnok = s_dn >= 0;
    if (dnok) {
        const float yl = f_add(py, -tab.smax[s_dn]);
        dnok = !(yl > 0.0f && f_mul(yl, yl) > sb);
    }
    if (!dnok)
        s_dn = -1;
    if (upok)
        return s_up++;
    if (dnok)
        return s_dn--;
    return -1;
}

// in-LDS bitonic sort of n2 (power of two) 64-bit keys, ascending (NT = threads of the workgroup)
template <int NT>
__device__ __forceinline__ void bitonic_sort_lds(unsigned long long *keys, unsigned n2)
{
    for (unsigned k = 2; k <= n2; k <<= 1) {
        for (unsigned j = k >> 1; j > 0; j >>= 1) {
            for (unsigned t = threadIdx.x; t < n2 / 2; t += NT) {
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

template <int NT, int TCAP>
struct PrepShared {
    // first the sort keys, then (same bytes) the sorted cloud with its sentinels
    unsigned long long buf[TCAP + SW_PAD + 4];
    double red[16 * 2 + 2]; // (sized for the 16 waves whose order every build reproduces, see the mean below)
    float mean[2];
    unsigned ykey[2], xkey[2]; // min / max order keys of the finite centred coordinates
    int cnt[SW_NS_MAX];
    unsigned smin_k[SW_NS_MAX], smax_k[SW_NS_MAX];
    StripTab tab;
};

// ---------------------------------------------------------------------------------------------
// prep: one workgroup per distinct target cloud
// ---------------------------------------------------------------------------------------------
// PCA normals of the centred target: K nearest incl. the point itself, ordered by (d2, original
// index) exactly like the brute-force scan (sfe_icp.hip).  s_tgt = sorted cloud in the strip layout
// (in LDS or in HBM scratch).
template <int KM, int NT> // NT = threads of the workgroup; KM = capacity of the neighbour list (>= K): its loops are fully unrolled, so a snug KM pays
__device__ __forceinline__ void sweep_knn_normals(const sfe_icp_params &P, const StripTab &tab,
                                                  const float2 *__restrict__ s_tgt, const int *__restrict__ perm,
                                                  float2 *__restrict__ snrm, int nt, int c_begin = 0, int c_stride = NT)
{ // (c_begin, c_stride: the positions this workgroup takes when several share a target, icp_sweep_normals_kernel)
    // (Tried in round 3 and dropped: a first pass that gives up on a point after 16 / 24 / 32 steps of its walk and a
    // second pass over the listed points packed into whole waves -- a point walks 16 steps on average on a sonar
    // cloud, the longest of 64 neighbours 41 -- 5.5 -> 5.9 ms per 4096 targets: the restarts and the second pass's own
    // longest walks cost more than the waiting lanes of the first.)
    const int tid = threadIdx.x;
    const int K = min(min(P.normals_knn, KM), nt);
    const int ns = tab.ns, len = tab.len;
    for (int c = c_begin + tid + 1; c < len; c += c_stride) { // positions; sentinels are skipped
        const float2 q = s_tgt[c];
        if (q.x != q.x && q.y != q.y)
            continue; // a sentinel (a cloud point that is NaN in both coordinates gets no normal either:
                      // nothing can ever match it)
        float bd[KM];
        int bj[KM];
        // The search runs twice at most.  First without the tie rule: equal distances are only NOTED (where they could
        // change the outcome: at the end of the list when an entry leaves, and in the finished list), the list orders by
        // distance alone.  Two equal distances among a point's candidates are
        // rare (exactly equal fp32 sums of squares); only then the point is searched again with the full rule
        // (equal distances order by original index, which costs a compare, a branch and -- when taken -- two reads
        // of the permutation per exchange step: about half of the instructions of an insertion).
        auto search = [&](auto exact_tag) -> int { // 0: done, 1: a tie the distance-only order cannot settle
        constexpr bool EXACT = decltype(exact_tag)::value;
        bool tie_seen = false;
#pragma unroll
        for (int k = 0; k < KM; ++k) {
            bd[k] = INFINITY;
            bj[k] = 0;
        }
        float kth = INFINITY; // bd[K-1]
        auto consider = [&](float d, int j) {
            if (!(d <= kth) || d == INFINITY)
                return;
            if (K == KM) {
                // full list (the usual case, KM == k): the newcomer replaces the last entry and bubbles up
                // with KM-1 compare-exchanges -- half the work of the count / shift / place form below.
                // Equal distances order by original index (rare: the permutation is only read then).
                if (d == kth) {
                    if (!EXACT) {
                        tie_seen = true;
                        return;
                    }
                    if (!(perm[j - 1] < perm[bj[KM - 1] - 1]))
                        return;
                }
                // (distance-only mode: the entry that leaves must not tie with the one that becomes last -- which of the two
                // stays is the tie rule's call; ties inside the list are looked for once, at the end of the search)
                if (!EXACT && KM >= 2)
                    tie_seen |= bd[KM - 2] == bd[KM - 1] && bd[KM - 1] < INFINITY;
                bd[KM - 1] = d;
                bj[KM - 1] = j;
#pragma unroll
                for (int k = KM - 1; k >= 1; --k) {
                    bool up = bd[k] < bd[k - 1];
                    if (EXACT) {
                        if (bd[k] == bd[k - 1] && bj[k - 1] != 0)
                            up = perm[bj[k] - 1] < perm[bj[k - 1] - 1];
                    }
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
        const int so = strip_of(q.y, tab.ylo, tab.inv_g, ns);
        int s_up = so, s_dn = so - 1;
        for (int guard = 0; guard < 2 * SW_NS_MAX + 2; ++guard) {
            const int s = next_strip(tab, ns, so, s_up, s_dn, q.y, kth);
            if (s < 0)
                break;
            int iR = (s == so) ? c : strip_lower_bound_lane(s_tgt, tab.sbeg[s], tab.sbeg[s + 1] - 1, q.x);
            int iL = iR - 1; // own strip: the point itself is the first right candidate
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
        }
        if (!EXACT && K == KM) {
#pragma unroll
            for (int k = 1; k < KM; ++k)
                tie_seen |= bd[k] == bd[k - 1] && bd[k] < INFINITY;
        }
        return tie_seen ? 1 : 0;
        };
        if (K != KM || search(std::false_type{}) == 1)
            search(std::true_type{});
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
        snrm[c - 1] = make_float2((float)(-ty / nn), (float)(tx / nn));
    }
}

// Witness grid of one target (for the first iteration of every job on it, which has no previous neighbours to start
// from): per cell the sorted position of a target point near the cell's centre -- any real point is a valid upper bound
// of a query's neighbour distance; a near one is a good bound.  Built without searching: every point claims its own
// cell (the point nearest to the centre wins: one atomicMin on (distance bits | position)), then
// a few dilation sweeps hand witnesses to the empty cells around occupied ones (a cell takes, among its 8 neighbours'
// witnesses, the one nearest to its own centre; in place, so a sweep carries them further than one cell).  Cells
// that stay empty are far from every structure: queries there start the first iteration without a witness, as before.
// (A per-cell nearest-neighbour search was tried first: the empty two thirds of a sonar fan's bounding box have their
// nearest point metres away, and those searches cost 2 ms per 512 targets.)
#define SW_GRID_SWEEPS 4
template <int NT>
__device__ __forceinline__ void sweep_grid_witness(const StripTab &tab, const float2 *__restrict__ s_tgt,
                                                   int *__restrict__ grid_out, unsigned *grid)
{ // grid: LDS, SW_GRID_MAX words: (distance to the cell centre, top 16 bits of its float pattern) << 16 | position
    const int gnx = tab.gnx, gny = tab.gny, ncell = gnx * gny, len = tab.len;
    const float cs = tab.ginv > 0.0f ? 1.0f / tab.ginv : 0.0f;
    const float gx0 = tab.gx0, gy0 = tab.gy0, ginv = tab.ginv;
    auto centre_of = [&](int c) {
        const int iy = c / gnx, ix = c - iy * gnx;
        return make_float2(gx0 + ((float)ix + 0.5f) * cs, gy0 + ((float)iy + 0.5f) * cs);
    };
    for (int c = threadIdx.x; c < ncell; c += NT)
        grid[c] = 0xFFFFFFFFu;
    __syncthreads();
    if (len < 65536) { // positions fit 16 bits (always for a target that lives in LDS)
        for (int p = threadIdx.x + 1; p < len; p += NT) {
            const float2 q = s_tgt[p];
            if (!(fabsf(q.x) < INFINITY && fabsf(q.y) < INFINITY))
                continue; // sentinels, non-finite points
            float gxv = f_mul(f_add(q.x, -gx0), ginv), gyv = f_mul(f_add(q.y, -gy0), ginv);
            gxv = fminf(fmaxf(gxv, 0.0f), (float)(gnx - 1));
            gyv = fminf(fmaxf(gyv, 0.0f), (float)(gny - 1));
            const int c = (int)gyv * gnx + (int)gxv;
            const float2 m = centre_of(c);
            // one atomic: the point nearest to the centre (to the 8 mantissa bits kept) wins, its position rides along
            atomicMin(&grid[c], (__float_as_uint(dist2(m.x, m.y, q.x, q.y)) & 0xFFFF0000u) | (unsigned)p);
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < ncell; c += NT)
        grid[c] = (grid[c] == 0xFFFFFFFFu) ? 0u : (grid[c] & 0xFFFFu); // -> position, 0 = empty
    __syncthreads();
    for (int it = 0; it < SW_GRID_SWEEPS; ++it) {
        for (int c = threadIdx.x; c < ncell; c += NT) {
            if (grid[c] != 0)
                continue;
            const int iy = c / gnx, ix = c - iy * gnx;
            const float2 m = centre_of(c);
            float best = INFINITY;
            unsigned bp = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int dx = (k < 3) ? k - 1 : (k == 3 ? -1 : (k == 4 ? 1 : k - 6));
                const int dy = (k < 3) ? -1 : (k < 5 ? 0 : 1);
                const int jx = ix + dx, jy = iy + dy;
                if (jx < 0 || jx >= gnx || jy < 0 || jy >= gny)
                    continue;
                const unsigned w = grid[jy * gnx + jx];
                if (w != 0) {
                    const float2 t = s_tgt[w];
                    const float d = dist2(m.x, m.y, t.x, t.y);
                    if (d < best) {
                        best = d;
                        bp = w;
                    }
                }
            }
            if (bp != 0)
                grid[c] = bp;
        }
        __syncthreads();
    }
    for (int c = threadIdx.x; c < ncell; c += NT)
        grid_out[c] = (int)grid[c];
}

// bitonic sort of n2 (power of two, > CH) 64-bit keys in HBM scratch by one workgroup (targets that do not fit LDS; once
// per target).  Only the exchange steps whose partners lie >= CH keys apart go through memory; every run of steps with
// closer partners is done on CH-key chunks staged in LDS (`chunk`, CH keys): of the 120 steps of a 32 768-key sort 3
// touch HBM, the rest run at LDS speed (0.7 -> ~0.2 ms for a 20 000-point cloud).
template <int NT, int CH>
__device__ __forceinline__ void bitonic_sort_global(unsigned long long *keys, unsigned n2, unsigned long long *chunk)
{
    // all steps (k', j) with k_lo <= k' <= k_hi, j < CH of the network, applied to every CH-aligned chunk: for k' < CH that
    // is the whole sub-network of the chunk, for k' >= CH the tail j = CH/2 .. 1 of merge step k' (k_lo == k_hi then)
    auto chunk_steps = [&](unsigned k_lo, unsigned k_hi) {
        for (unsigned c0 = 0; c0 < n2; c0 += CH) {
            for (unsigned t = threadIdx.x; t < CH; t += NT)
                chunk[t] = keys[c0 + t];
            __syncthreads();
            for (unsigned k = k_lo; k <= k_hi; k <<= 1) {
                for (unsigned j = (k >> 1 < CH ? k >> 1 : CH >> 1); j > 0; j >>= 1) {
                    for (unsigned t = threadIdx.x; t < CH / 2; t += NT) {
                        const unsigned i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                        const unsigned l = i | j;
                        const unsigned long long a = chunk[i], b = chunk[l];
                        const bool up = ((c0 + i) & k) == 0;
                        if ((a > b) == up) {
                            chunk[i] = b;
                            chunk[l] = a;
                        }
                    }
                    __syncthreads();
                }
            }
            for (unsigned t = threadIdx.x; t < CH; t += NT)
                keys[c0 + t] = chunk[t];
            __syncthreads(); // same workgroup, same CU: its L1 sees its own write-through stores
        }
    };
    chunk_steps(2, CH); // every chunk sorted (ascending or descending by its place in the network)
    for (unsigned k = 2 * CH; k <= n2; k <<= 1) {
        for (unsigned j = k >> 1; j >= CH; j >>= 1) {
            for (unsigned t = threadIdx.x; t < n2 / 2; t += NT) {
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
        chunk_steps(k, k);
    }
}

// sort key: strip (8 bits) | order key of x (32 bits) | original index (24 bits)
#define SW_KEY(s, xk, i) (((unsigned long long)(unsigned)(s) << 56) | ((unsigned long long)(xk) << 24) | (unsigned long long)(i))
#define SW_KEY_STRIP(k) ((int)((k) >> 56))
#define SW_KEY_X(k) ((unsigned)(((k) >> 24) & 0xFFFFFFFFull))
#define SW_KEY_ID(k) ((int)((k) & 0xFFFFFFull))

// NT threads; targets of up to TCAP points are sorted (and their normals / witness grid built) in LDS; GM = most cells
// of the witness grid.  prep_ids[blockIdx.x] = the target this workgroup prepares (one launch per tier).
// GTAIL (the 1024-thread build): no LDS of its own for the witness grid -- it is built last, in the part of the key
// buffer the sorted cloud leaves free (a 5 000-point cloud: 5 067 of 8 264 slots; its grid has ~4 300 cells of 4
// bytes), or in place in the output array when that is too small.  66 KB per workgroup instead of 100: two
// workgroups per CU, so that one's k-NN walks fill the other's barriers and LDS waits.
// KMF: capacity of the k-NN list when the launch knows normals_knn (0: all four capacities in one kernel, chosen at run
// time -- whose registers are then those of the largest; the 64-register GTAIL build spilled 705 of them that way).
template <int NT, int TCAP, int GM, bool GTAIL = false, int KMF = 0>
__global__ __launch_bounds__(NT, GTAIL ? 8 : 4) void icp_sweep_prep_kernel(sfe_icp_params P,
                                                                        const SweepPrep *__restrict__ preps,
                                                                        const int *__restrict__ prep_ids,
                                                                        const float2 *__restrict__ tgt_all,
                                                                        float2 *__restrict__ stgt_all,
                                                                        int *__restrict__ perm_all,
                                                                        float2 *__restrict__ snrm_all,
                                                                        float *__restrict__ mean_all,
                                                                        unsigned long long *__restrict__ gkeys_all,
                                                                        StripTab *__restrict__ tab_all,
                                                                        int *__restrict__ grid_all)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    PrepShared<NT, TCAP> &S = *reinterpret_cast<PrepShared<NT, TCAP> *>(smem_raw);
    int *s_grid = reinterpret_cast<int *>(smem_raw + ((sizeof(PrepShared<NT, TCAP>) + 15) & ~(size_t)15)); // witness grid + distances (not GTAIL)
    const int pid = __builtin_amdgcn_readfirstlane(prep_ids[blockIdx.x]);
    const SweepPrep J = preps[pid];
    const int nt = J.n_tgt, ns = J.ns, tid = threadIdx.x, lane = threadIdx.x & 63;
    const float2 *__restrict__ tgt = tgt_all + J.tgt_start;
    float2 *__restrict__ stgt = stgt_all + J.off;
    int *__restrict__ perm = perm_all + J.off;
    const float qnan = __uint_as_float(0x7FC00000u);

    // reference mean (fp64 accumulation, rounded to float), as the brute-force kernel -- and in ITS order whatever NT is:
    // point i belongs to thread i mod 1024 of a 1024-thread workgroup, 64 consecutive threads are a wave (fixed tree), the
    // 16 wave totals are added left to right.  A smaller workgroup plays those waves one after the other (a target of a
    // few points makes the ICP sums rank-deficient, and then the last bit of the mean decides the outcome).
    {
        double m[2] = {0, 0};
        if constexpr (NT == 1024) {
            for (int i = tid; i < nt; i += NT) {
                const float2 t = tgt[i];
                m[0] += t.x;
                m[1] += t.y;
            }
            block_sum<2, NT>(m, S.red);
        } else {
            const int wave = tid >> 6;
            for (int w0 = wave; w0 < 16; w0 += NT / 64) {
                double a0 = 0, a1 = 0;
                for (int i = 64 * w0 + lane; i < nt; i += 1024) {
                    const float2 t = tgt[i];
                    a0 += t.x;
                    a1 += t.y;
                }
                a0 = wave_sum(a0);
                a1 = wave_sum(a1);
                if (lane == 0) {
                    S.red[2 * w0] = a0;
                    S.red[2 * w0 + 1] = a1;
                }
            }
            __syncthreads();
            for (int w = 0; w < 16; ++w) { // (every thread: the same sixteen additions)
                m[0] += S.red[2 * w];
                m[1] += S.red[2 * w + 1];
            }
            __syncthreads();
        }
        if (tid == 0) {
            S.mean[0] = (float)(m[0] / nt);
            S.mean[1] = (float)(m[1] / nt);
            mean_all[2 * pid] = S.mean[0];
            mean_all[2 * pid + 1] = S.mean[1];
            S.ykey[0] = S.xkey[0] = 0xFFFFFFFFu;
            S.ykey[1] = S.xkey[1] = 0u;
        }
        if (tid < SW_NS_MAX) {
            S.cnt[tid] = 0;
            S.smin_k[tid] = 0xFFFFFFFFu;
            S.smax_k[tid] = 0u;
        }
        __syncthreads();
    }
    const float mx = S.mean[0], my = S.mean[1];

    // extent of the finite centred coordinates -> strip geometry
    {
        float ylo = INFINITY, yhi = -INFINITY, xlo = INFINITY, xhi = -INFINITY;
        for (int i = tid; i < nt; i += NT) {
            const float2 t = tgt[i];
            const float x = f_add(t.x, -mx), y = f_add(t.y, -my);
            if (fabsf(y) < INFINITY) {
                ylo = fminf(ylo, y);
                yhi = fmaxf(yhi, y);
            }
            if (fabsf(x) < INFINITY) {
                xlo = fminf(xlo, x);
                xhi = fmaxf(xhi, x);
            }
        }
        ylo = wave_min(ylo);
        yhi = -wave_min(-yhi);
        xlo = wave_min(xlo);
        xhi = -wave_min(-xhi);
        if (lane == 0) {
            atomicMin(&S.ykey[0], mono_key(ylo));
            atomicMax(&S.ykey[1], mono_key(yhi));
            atomicMin(&S.xkey[0], mono_key(xlo));
            atomicMax(&S.xkey[1], mono_key(xhi));
        }
        __syncthreads();
        if (tid == 0) {
            const float y0 = mono_inv(S.ykey[0]), y1 = mono_inv(S.ykey[1]);
            const float x0 = mono_inv(S.xkey[0]), x1 = mono_inv(S.xkey[1]);
            S.tab.ns = ns;
            S.tab.len = nt + ns + 1;
            S.tab.ylo = (y1 >= y0) ? y0 : 0.0f; // no finite point: everything lands in strip 0
            const float inv = (y1 > y0) ? (float)ns / f_add(y1, -y0) : 0.0f;
            S.tab.inv_g = (inv < INFINITY) ? inv : 0.0f;
            S.tab.ext_x = (x1 >= x0) ? f_add(x1, -x0) : 0.0f;
            // witness grid over the bounding box: about one cell per target point, at most GM cells
            const float ex = (x1 >= x0) ? f_add(x1, -x0) : 0.0f, ey = (y1 >= y0) ? f_add(y1, -y0) : 0.0f;
            float cs = sqrtf(fmaxf(ex, 1e-30f) * fmaxf(ey, 1e-30f) / (float)max(nt, 1));
            cs = fmaxf(cs, sqrtf(fmaxf(ex, 1e-30f) * fmaxf(ey, 1e-30f) / (float)(GM / 2)));
            if (!(cs > 0.0f) || !(cs < INFINITY))
                cs = 1.0f;
            int gnx = (int)fminf(ex / cs, 4096.0f) + 1, gny = (int)fminf(ey / cs, 4096.0f) + 1;
            while ((long long)gnx * gny > GM) { // a very elongated box
                if (gnx >= gny)
                    gnx = (gnx + 1) / 2;
                else
                    gny = (gny + 1) / 2;
            }
            S.tab.gx0 = (x1 >= x0) ? x0 : 0.0f;
            S.tab.gy0 = (y1 >= y0) ? y0 : 0.0f;
            S.tab.gnx = gnx;
            S.tab.gny = gny;
            S.tab.grid_off64 = (int)(J.grid_off / 64);
            S.tab.pad_ = 0;
            // one cell size for both axes, large enough that gnx x gny cells cover the box
            const float csx = ex / (float)gnx, csy = ey / (float)gny;
            const float csz = fmaxf(fmaxf(csx, csy), 1e-30f);
            S.tab.ginv = 1.0f / csz;
            if (!(S.tab.ginv < INFINITY))
                S.tab.ginv = 0.0f;
        }
        __syncthreads();
    }
    const float ylo = S.tab.ylo, inv_g = S.tab.inv_g;

    // sort (strip, key(x - mean_x), index); strip population and y range on the way
    unsigned n2 = 2;
    while (n2 < (unsigned)nt)
        n2 <<= 1;
    const bool in_lds = nt <= TCAP;
    unsigned long long *keys = in_lds ? S.buf : gkeys_all + J.key_off;
    for (unsigned i = tid; i < n2; i += NT) {
        unsigned long long k = ~0ull;
        if (i < (unsigned)nt) {
            const float2 t = tgt[i];
            const float x = f_add(t.x, -mx), y = f_add(t.y, -my);
            const int s = strip_of(y, ylo, inv_g, ns);
            k = SW_KEY(s, mono_key(x), i);
            atomicAdd(&S.cnt[s], 1);
            if (y == y) {
                atomicMin(&S.smin_k[s], mono_key(y));
                atomicMax(&S.smax_k[s], mono_key(y));
            }
        }
        keys[i] = k;
    }
    __syncthreads();
    if (tid == 0) {
        int pos = 1;
        for (int s = 0; s < ns; ++s) {
            S.tab.sbeg[s] = pos;
            pos += S.cnt[s] + 1;
        }
        for (int s = ns; s <= SW_NS_MAX; ++s)
            S.tab.sbeg[s] = pos;
        float m = INFINITY;
        for (int s = SW_NS_MAX - 1; s >= 0; --s) {
            if (s < ns && S.smin_k[s] != 0xFFFFFFFFu)
                m = fminf(m, mono_inv(S.smin_k[s]));
            S.tab.smin[s] = m;
        }
        m = -INFINITY;
        for (int s = 0; s < SW_NS_MAX; ++s) {
            if (s < ns && S.smax_k[s] != 0u)
                m = fmaxf(m, mono_inv(S.smax_k[s]));
            S.tab.smax[s] = m;
        }
    }
    __syncthreads();
    { // the table travels to HBM for the loop kernel
        const int *src = reinterpret_cast<const int *>(&S.tab);
        int *dst = reinterpret_cast<int *>(tab_all + pid);
        for (int i = tid; i < (int)(sizeof(StripTab) / sizeof(int)); i += NT)
            dst[i] = src[i];
    }
    const int len = S.tab.len;
    // sentinels of the HBM copy
    for (int s = tid; s <= ns; s += NT)
        stgt[s == 0 ? 0 : S.tab.sbeg[s] - 1] = make_float2(qnan, qnan);
    if (tid < 2)
        stgt[len + tid] = make_float2(qnan, qnan);

    auto grid_witness = [&](const float2 *cloud, int used_slots) { // used_slots: 8-byte slots of S.buf the cloud occupies
        int *out = grid_all + J.grid_off;
        if constexpr (GTAIL) {
            const int room = (TCAP + SW_PAD + 4 - used_slots) * 2, ncell = S.tab.gnx * S.tab.gny;
            unsigned *tail = reinterpret_cast<unsigned *>(S.buf + used_slots);
            sweep_grid_witness<NT>(S.tab, cloud, out, ncell <= room ? tail : reinterpret_cast<unsigned *>(out));
        } else {
            sweep_grid_witness<NT>(S.tab, cloud, out, reinterpret_cast<unsigned *>(s_grid));
        }
    };
    float2 *nrm = snrm_all ? snrm_all + J.off : nullptr;
    if (in_lds) {
        bitonic_sort_lds<NT>(S.buf, n2);
        // keys -> sorted centred cloud (registers -> same LDS bytes, in the strip layout)
        constexpr int PER = TCAP / NT;
        float2 v[PER];
        int id[PER], ps[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int r = k * NT + tid;
            v[k] = make_float2(0, 0);
            id[k] = 0;
            ps[k] = 0;
            if (r < nt) {
                const unsigned long long key = S.buf[r];
                id[k] = SW_KEY_ID(key);
                ps[k] = r + SW_KEY_STRIP(key) + 1;
                v[k] = make_float2(mono_inv(SW_KEY_X(key)), f_add(tgt[id[k]].y, -my));
            }
        }
        __syncthreads();
        float2 *s_tgt = reinterpret_cast<float2 *>(S.buf);
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            if (k * NT + tid < nt) {
                s_tgt[ps[k]] = v[k];
                stgt[ps[k]] = v[k];
                perm[ps[k] - 1] = id[k];
            }
        }
        for (int s = tid; s <= ns; s += NT)
            s_tgt[s == 0 ? 0 : S.tab.sbeg[s] - 1] = make_float2(qnan, qnan);
        if (tid < 2)
            s_tgt[len + tid] = make_float2(qnan, qnan);
        __syncthreads();
        if (P.minimizer == 1) {
            if constexpr (KMF != 0)
                sweep_knn_normals<KMF, NT>(P, S.tab, s_tgt, perm, nrm, nt);
            else if (P.normals_knn <= 8)
                sweep_knn_normals<8, NT>(P, S.tab, s_tgt, perm, nrm, nt);
            else if (P.normals_knn <= 10)
                sweep_knn_normals<10, NT>(P, S.tab, s_tgt, perm, nrm, nt);
            else if (P.normals_knn <= 12)
                sweep_knn_normals<12, NT>(P, S.tab, s_tgt, perm, nrm, nt);
            else
                sweep_knn_normals<ICP_KMAX, NT>(P, S.tab, s_tgt, perm, nrm, nt);
        }
        if (grid_all)
            grid_witness(s_tgt, len + 2);
    } else {
        bitonic_sort_global<NT, TCAP>(keys, n2, S.buf); // (n2 > TCAP here; S.buf holds TCAP + SW_PAD + 4 keys)
        for (int r = tid; r < nt; r += NT) {
            const unsigned long long key = keys[r];
            const int id = SW_KEY_ID(key), pos = r + SW_KEY_STRIP(key) + 1;
            stgt[pos] = make_float2(mono_inv(SW_KEY_X(key)), f_add(tgt[id].y, -my));
            perm[pos - 1] = id;
        }
        __syncthreads();
        if (P.minimizer == 1 && J.pad_ == 0) { // (pad_ = 1: icp_sweep_normals_kernel computes them, many workgroups per target)
            if constexpr (KMF != 0)
                sweep_knn_normals<KMF, NT>(P, S.tab, stgt, perm, nrm, nt);
            else if (P.normals_knn <= 8)
                sweep_knn_normals<8, NT>(P, S.tab, stgt, perm, nrm, nt);
            else if (P.normals_knn <= 10)
                sweep_knn_normals<10, NT>(P, S.tab, stgt, perm, nrm, nt);
            else if (P.normals_knn <= 12)
                sweep_knn_normals<12, NT>(P, S.tab, stgt, perm, nrm, nt);
            else
                sweep_knn_normals<ICP_KMAX, NT>(P, S.tab, stgt, perm, nrm, nt);
        }
        if (grid_all)
            grid_witness(stgt, 0); // (the key buffer was the sort's staging chunk: free now)
    }
}

// PCA normals of the targets that do not fit LDS (sorted in HBM scratch by their prep workgroup): a 20 000-point cloud
// keeps ONE workgroup busy for over a millisecond with them -- longer than a many-to-one batch on that cloud then
// iterates per share -- so they are dealt to gridDim.y workgroups per target here.  Same function, same neighbours.
template <int NT>
__global__ __launch_bounds__(NT, 4) void icp_sweep_normals_kernel(sfe_icp_params P, const SweepPrep *__restrict__ preps,
                                                                  const int *__restrict__ prep_ids,
                                                                  const float2 *__restrict__ stgt_all, const int *__restrict__ perm_all,
                                                                  float2 *__restrict__ snrm_all, const StripTab *__restrict__ tab_all)
{
    __shared__ StripTab s_tab;
    const int pid = __builtin_amdgcn_readfirstlane(prep_ids[blockIdx.x]);
    const SweepPrep J = preps[pid];
    {
        const int *src = reinterpret_cast<const int *>(tab_all + pid);
        int *dst = reinterpret_cast<int *>(&s_tab);
        for (int i = threadIdx.x; i < (int)(sizeof(StripTab) / sizeof(int)); i += NT)
            dst[i] = src[i];
    }
    __syncthreads();
    const float2 *stgt = stgt_all + J.off;
    const int *perm = perm_all + J.off;
    float2 *nrm = snrm_all + J.off;
    const int c0 = blockIdx.y * NT, cs = gridDim.y * NT;
    if (P.normals_knn <= 8)
        sweep_knn_normals<8, NT>(P, s_tab, stgt, perm, nrm, J.n_tgt, c0, cs);
    else if (P.normals_knn <= 10)
        sweep_knn_normals<10, NT>(P, s_tab, stgt, perm, nrm, J.n_tgt, c0, cs);
    else if (P.normals_knn <= 12)
        sweep_knn_normals<12, NT>(P, s_tab, stgt, perm, nrm, J.n_tgt, c0, cs);
    else
        sweep_knn_normals<ICP_KMAX, NT>(P, s_tab, stgt, perm, nrm, J.n_tgt, c0, cs);
}

// ---------------------------------------------------------------------------------------------
// loop: one workgroup per job
// ---------------------------------------------------------------------------------------------
// Only pairs that end up with weight 1 need their exact neighbour: d2 <= the trimmed-quantile
// limit (and <= MaxDist^2).  A search is therefore exhaustive only out to a cap C (squared
// radius), and merely keeps going until it has seen SOME target within KDTreeMatcher.maxDist so
// that the count of finite matches is exact.  A query ends as
//   none    : no target within maxDist (exact: its whole maxDist window was searched)
//   exact   : best <= C, every candidate that could beat or tie `best` was evaluated
//   inexact : finite, C < d2_NN <= best            (suspended: best and its position are kept)
// If the exact set holds more than k = floor(n_finite * ratio) values, the k-th smallest of them
// IS the k-th smallest of all (everything else is > C), the limit is exact and so are all
// weight-1 pairs.  Otherwise the suspended queries search again with C = the k-th smallest of the upper
// bounds all finite queries hold (>= the k-th smallest distance: one repeat suffices), from scratch but
// bounded by the best they already hold; a candidate is never mistaken for a tie with itself because
// the position of the current best is excluded.  C starts from the previous iteration's limit (+ a margin),
// so far outliers cost a handful of steps.  Decisions and results are identical to the exhaustive search.
//
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

// control block of a job; the LDS-resident variant places the sorted target right behind it.  The profile counters
// and the transform history of the clearance records only take room in the builds that use them (the small-job tiers
// run many workgroups per CU: every KB of control block is a job less per CU).
template <int NT, bool PROF, bool REC>
struct SweepShared {
    double red[(NT / 64) * 10 + 10 > 16 * 5 ? (NT / 64) * 10 + 10 : 16 * 5]; // (16 x 5: the canonical order of the sums, below)
    double acc[10];  // the reduced error-minimiser sums (read by the solving lane)
    unsigned hist[256], hist0[256];
    unsigned sel_prefix, sel_k;
    unsigned n_none, n_exact; // census of the iteration, tallied where a query is settled
    int grid_skips;           // first iteration: queries that took a grid witness instead of searching in round 0
    unsigned n_rechit[2];     // queries settled by their clearance record, this iteration / the one before
    int long_n, long_next, mid_n, wl_n[2];
    int flag_iterate, flag_status;
    float Ti[9];
    float thist[REC ? ICP_MAX_HIST : 1][6]; // T_iter of every iteration so far (rows 0 and 1): the movement bounds below
    float mva[REC ? ICP_MAX_HIST : 1], mvt[REC ? ICP_MAX_HIST : 1]; // a query x = T0 * src has moved by at most
                                                // mva[k] |x| + mvt[k] between iteration k and the current one
    unsigned rmax_bits;           // largest |T0 * src| (float bits; >= 0 so the bit patterns order like the values)
    float hist_c[ICP_MAX_HIST], hist_s[ICP_MAX_HIST], hist_x[ICP_MAX_HIST], hist_y[ICP_MAX_HIST];
    long long prof_t, prof[PROF ? 16 : 1], prof_it[PROF ? 64 : 1], prof_b0;
    unsigned xr[16]; // split jobs: the scalars of an exchange between the workgroups of a job
    int xabort;      // ... and its time-out flag
    int win_smin, win_smax, win_lo, win_n; // WIN: strips of this workgroup's queries; the target positions held in LDS
    StripTab tab;
};

#define SW_PROF(k)                                                                               \
    do {                                                                                         \
        if (PROF && threadIdx.x == 0) {                                               \
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
#define SW_BUDGET_A 6   // first pass (own strip): walk trips (4 candidates each) before a query is handed on
#define SW_BUDGET 128   // second pass (all strips): trips + strips before a query is handed to the cooperative tier
#define SW_CAP_MARGIN 15 // percent
#define SW_ROUND_TRIPS 4 // second pass: walk trips between two chances to move on to the next strip
#define SW_NONE (-1)
// an unfinished / suspended (inexact) query is stored as pos = -2 - bpos (<= -2; bpos = 0: nothing met
// yet): the target it holds bounds its next search and doubles as the next iteration's witness
#define SW_INEXACT_OF(bpos) (-2 - (bpos))
#define SW_OWN_DONE 0x80000000u // list entry flags of queries handed from the first to the second pass:
#define SW_TIED 0x40000000u     // own strip finished / a tie with the current best was seen there
#define SW_PARTIAL 0x20000000u  // ... / the runner-up distance of what it has visited so far waits in Q.st[q].z
#define SW_QMASK 0x1FFFFFFFu
// Clearance records (steady-state iterations): a search looks a little further than it has to -- out to (1 + m)^2 x the
// squared bound it would stop at -- and remembers R = distance from the query to the nearest target OTHER than its
// neighbour (min of the runner-up among the visited candidates and the edge of the searched window), with the iteration
// it was taken in.  In a later iteration the query has moved by at most mvb (a bound over all queries from the two
// transforms): if its old neighbour, evaluated first as the witness, is closer than R - mvb, it is still THE nearest
// target and nothing is searched; likewise a query beyond the cap C whose every target is provably beyond C.  Once
// the clouds have converged (a few mm per iteration against neighbour distances of centimetres) almost every query
// takes this path: the iteration costs a transform, one distance and the census.  Decisions and results are those of
// the full search: the skip needs a strict gap (1e-5 relative, two orders above the fp32 rounding of the distances).
#define SW_REC_MIN_ITER 12
#define SW_REC_KAPPA 3.0f
#define SW_REC_MARGIN 8 // percent: the search radius grows by 8 %, ~17 % more candidates

struct SweepQ { // per-job views of the per-query scratch (the transformed query itself is never stored: whoever needs
                // it again recomputes it from the source point, two affine maps with wave-uniform coefficients)
    int4 *st;     // clearance record of a `none` query: (px, py, clearance) as float bits; .z doubles as the runner-up
                  // distance a search carries from the first pass to the second (records build)
    float *d2;    // best so far / final d2
    int *pos;     // >= 0 sorted position - 1 of the NN, SW_NONE, <= -2 inexact (SW_INEXACT_OF)
    int *wl[2];   // work lists of suspended queries (ping-pong between rounds)
    int *mid;     // queries that outlived the first pass (compacted for the second)
    int *lng;     // queries handed to the cooperative tier this round
    int *order;   // all queries, neighbours in space next to each other (see the sort at the kernel start)
    int *slot_of; // ... and the inverse: position of query q in that order
    unsigned *rec; // clearance records, by position in `order` (the fresh pass streams through them)
    float2 *ssrc; // their source points in that order
    const int *perm;
};

// ties at the final best: lowest original index among the points at distance `best`, found by
// searching the final window once more (rare)
template <class TV>
__device__ __forceinline__ int sweep_resolve_tie(const TV &T, const StripTab &tab, const SweepQ &Q,
                                                 float px, float py, float best)
{
    int bo = 0x7FFFFFFF, bp = 0;
    const int ns = tab.ns, so = strip_of(py, tab.ylo, tab.inv_g, ns);
    int s_up = so, s_dn = so - 1;
    for (int guard = 0; guard < 2 * SW_NS_MAX + 2; ++guard) {
        const int s = next_strip(tab, ns, so, s_up, s_dn, py, best);
        if (s < 0)
            break;
        const int lo = strip_lower_bound_lane(T, tab.sbeg[s], tab.sbeg[s + 1] - 1, px);
        for (int dir = 0; dir < 2; ++dir) {
            for (int j = dir ? lo : lo - 1;; j += dir ? 1 : -1) { // the strip's NaN sentinels end both walks
                const float2 t = T[j];
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
        }
    }
    return bp;
}

// LDS_TGT: sorted target resident in LDS (n_tgt <= SW_TCAP) or read from its HBM scratch slice (it
// stays in L2: <= 160 KB for a 20k-point cloud, shared by all guesses of a many-to-one batch).
// LDS_Q: the per-query results (d2 as float, position as int16) live in LDS behind the target instead of HBM
// scratch: with two jobs per CU the scratch of the 64 jobs an XCD runs at a time (~20 MB) does not fit its 4 MB
// L2, so every phase that streams over the results (radix select, error-minimiser sums, the witness lookup of
// the next iteration) otherwise waits for Infinity-Cache / HBM latencies.  Chosen by the launcher when
// control block + 8 (n_tgt + pad) + 6 n_src bytes fit the workgroup's LDS share (5000 x 5000: 76 KB of 80).
// PROF: per-phase cycle counters of workgroup 0 and launch-wide counts of the work done (candidate evaluations,
// lower-bound probes); instantiated for the two-jobs-per-CU builds with an LDS-resident target only.
// REC: the build with clearance records (below); chosen by the launcher for chains that run many iterations.
// MULTI: the job is one of J.ngrp shares of a caller's job (its queries: one band of strips, gathered by
// icp_split_kernel).  Every share runs the whole loop on its own queries; what an iteration decides from ALL queries --
// the census of a search round, the histograms of the radix select, the sums of the error minimiser -- is exchanged
// through the job's sync area (xreduce below) and every share takes the same decisions and solves the same system.
// WIN (targets beyond SW_TCAP points): the part of the sorted target around this workgroup's queries is held in LDS
// (TgtWin), t_cap = its capacity in points.
template <int NT, int MINW, bool LDS_TGT, bool LDS_Q, bool PROF, bool REC, bool MULTI, bool WIN = false>
__global__ __launch_bounds__(NT, MINW) void icp_sweep_kernel(
    sfe_icp_params P, const SweepJob *__restrict__ jobs, const int *__restrict__ job_ids, const float2 *__restrict__ src_all,
    const float *__restrict__ guess_all, const float2 *__restrict__ stgt_all, const int *__restrict__ perm_all,
    const float2 *__restrict__ snrm_all, const float *__restrict__ mean_all, const StripTab *__restrict__ tab_all,
    const int *__restrict__ grid_all, int4 *__restrict__ q_st_all, int *__restrict__ q_wl_all, float2 *__restrict__ q_ssrc_all,
    float *__restrict__ nn_d2_all,
    int *__restrict__ nn_pos_all, float *__restrict__ T_out, int *__restrict__ status_out,
    int *__restrict__ iters_out, long long *prof, int *dbg, int sw_budget, int sw_budget_a, int sw_cache, int t_cap, int q_cap, int sort_chunk, float sw_m, float sw_kappa,
    unsigned long long *__restrict__ sync_all, int sw_cache2)
{
    static_assert(LDS_TGT || !LDS_Q, "LDS_Q needs the LDS-resident target layout");
    static_assert(!MULTI || !PROF, "the profile build runs whole jobs");
    static_assert(!WIN || !LDS_TGT, "a window is for targets that do not fit LDS");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    using Shared = SweepShared<NT, PROF, REC>;
    Shared &S = *reinterpret_cast<Shared *>(smem_raw);

    const int jb = __builtin_amdgcn_readfirstlane(job_ids[blockIdx.x]);
    SweepJob J = jobs[jb];
    // the job record is the same for every lane: keep it (and every pointer derived from it) in SGPRs
    J.src_start = __builtin_amdgcn_readfirstlane(J.src_start);
    J.n_src = __builtin_amdgcn_readfirstlane(J.n_src);
    J.n_tgt = __builtin_amdgcn_readfirstlane(J.n_tgt);
    J.prep = __builtin_amdgcn_readfirstlane(J.prep);
    J.out = __builtin_amdgcn_readfirstlane(J.out);
    J.grp = __builtin_amdgcn_readfirstlane(J.grp);
    J.ngrp = __builtin_amdgcn_readfirstlane(J.ngrp);
    J.sync = __builtin_amdgcn_readfirstlane(J.sync);
    J.tgt_off = sw_uniform_ll(J.tgt_off);
    J.q_off = sw_uniform_ll(J.q_off);
    const int ns = J.n_src, nt = J.n_tgt;
    const float2 *__restrict__ src = src_all + J.src_start;
    const float2 *__restrict__ stgt = stgt_all + J.tgt_off;
    float2 *lds_tgt = reinterpret_cast<float2 *>(smem_raw + ((sizeof(Shared) + 15) & ~(size_t)15));
    using TV = std::conditional_t<WIN, TgtWin, const float2 *>;
    TV T; // sorted target incl. sentinels
    if constexpr (WIN)
        T = TgtWin{stgt, lds_tgt, 0, 0u}; // (the window is chosen and filled behind the query sort)
    else
        T = LDS_TGT ? (const float2 *)lds_tgt : stgt;
    const float2 *__restrict__ snrm = snrm_all ? snrm_all + J.tgt_off : nullptr;
    SweepQ Q;
    Q.st = q_st_all + J.q_off;
    Q.d2 = nn_d2_all + J.q_off;
    Q.pos = nn_pos_all + J.q_off;
    // LDS_Q: [target: t_cap float2][d2: q_cap float][pos: q_cap int16] behind the control block
    float *l_d2 = reinterpret_cast<float *>(lds_tgt + t_cap);
    short *l_pos = reinterpret_cast<short *>(l_d2 + q_cap);
    auto Pz = [&](int i) -> int { // position record of query i: >= 0 exact, SW_NONE, <= -2 inexact
        if constexpr (LDS_Q)
            return (int)l_pos[i];
        else
            return Q.pos[i];
    };
    auto Dz = [&](int i) -> float {
        if constexpr (LDS_Q)
            return l_d2[i];
        else
            return Q.d2[i];
    };
    auto setQ = [&](int i, float d, int pz) {
        if constexpr (LDS_Q) {
            l_d2[i] = d;
            l_pos[i] = (short)pz;
        } else {
            Q.d2[i] = d;
            Q.pos[i] = pz;
        }
    };
    Q.wl[0] = q_wl_all + 7 * J.q_off;
    Q.wl[1] = Q.wl[0] + ns;
    Q.mid = Q.wl[1] + ns;
    Q.lng = Q.mid + ns;
    Q.order = Q.lng + ns;
    Q.slot_of = Q.order + ns;
    Q.rec = reinterpret_cast<unsigned *>(Q.slot_of + ns);
    Q.ssrc = q_ssrc_all + J.q_off;
    Q.perm = perm_all + J.tgt_off;
    const float *guess = guess_all + 9 * (size_t)J.out;
    const int tid = threadIdx.x, lane = threadIdx.x & 63;
    const float mx = sw_uniform(mean_all[2 * J.prep]), my = sw_uniform(mean_all[2 * J.prep + 1]);

    if (PROF && tid == 0) {
        for (int i = 0; i < 16; ++i)
            S.prof[i] = 0;
        for (int i = 0; i < 64; ++i) // (LDS is not zeroed: words no iteration / round writes used to come out as garbage)
            S.prof_it[i] = 0;
        S.prof_t = clock64();
    }
    // wave-uniform work counters (PROF only): candidate distance evaluations of the lane-per-query tiers, of the
    // cooperative tier, witness evaluations, lower-bound probes
    unsigned long long c_eval = 0, c_coop = 0, c_wit = 0, c_lb = 0;
    if (tid == 0) {
        S.rmax_bits = 0u;
        S.xabort = 0;
        S.win_smin = 0x7FFFFFFF;
        S.win_smax = -1;
    }
    { // strip table -> LDS
        const int *tsrc = reinterpret_cast<const int *>(tab_all + J.prep);
        int *tdst = reinterpret_cast<int *>(&S.tab);
        for (int i = tid; i < (int)(sizeof(StripTab) / sizeof(int)); i += NT)
            tdst[i] = tsrc[i];
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
    __syncthreads(); // strip table in place

    // ---- split jobs: what the shares of a job tell each other ----
    // R2 of the hand-off recipe (cdna_hip_programming.md, Guideline 16): the data is the flag.  A share publishes its
    // words as 8-byte granules {epoch, value} (one sc1 store each, no fence) in its row of the job's sync area and
    // reads the same words of EVERY share until their tags show this epoch.  Consecutive exchanges alternate between
    // two banks: a share can be one exchange ahead of the slowest one, never two (it needs everybody's words of the
    // exchange in between).  Every share adds the rows up in share order, so all of them hold the same totals, take
    // the same decisions and run the same number of rounds and iterations.  The area is zeroed before every launch
    // (tag 0 = nothing yet).  A wait of more than ~0.5 s (a share that never became resident: the launcher only splits
    // jobs when all shares fit the device at once) gives up and the job reports SFE_ICP_SPLIT_TIMEOUT.
    typedef __attribute__((address_space(1))) unsigned long long gu64;
    unsigned xepoch = 0;
    gu64 *xsync = MULTI ? (gu64 *)(sync_all + (size_t)J.sync * (size_t)(2 * SW_MG_MAX * SW_MG_WORDS)) : nullptr;
    auto xpoll = [&](gu64 *g, unsigned epoch) -> unsigned {
        unsigned long long x = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(x >> 32) != epoch) {
            const unsigned long long t0 = wall_clock64(); // 100 MHz
            while (true) {
                __builtin_amdgcn_s_sleep(8);
                x = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((unsigned)(x >> 32) == epoch)
                    break;
                if (wall_clock64() - t0 > 50000000ull) { // 0.5 s
                    S.xabort = 1;
                    break;
                }
            }
        }
        return (unsigned)x;
    };
    auto xbail = [&]() { // (all threads, after a barrier)
        if (tid == 0 && J.grp == 0) {
            for (int i = 0; i < 9; ++i)
                T_out[9 * (size_t)J.out + i] = guess[i];
            status_out[J.out] = SFE_ICP_SPLIT_TIMEOUT;
            iters_out[J.out] = 0;
        }
        __builtin_amdgcn_endpgm();
    };
    // vals[0..n) (LDS, n <= SW_MG_WORDS) -> their sums over the shares; called by every thread
    auto xreduce_u32 = [&](unsigned *vals, int n) {
        __syncthreads(); // the words are final
        ++xepoch;
        gu64 *bank = xsync + (size_t)(xepoch & 1u) * (SW_MG_MAX * SW_MG_WORDS);
        for (int t = tid; t < n; t += NT)
            __hip_atomic_store(bank + J.grp * SW_MG_WORDS + t, ((unsigned long long)xepoch << 32) | vals[t], __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        for (int t = tid; t < n; t += NT) {
            unsigned sum = 0;
            for (int w = 0; w < J.ngrp; ++w)
                sum += xpoll(bank + w * SW_MG_WORDS + t, xepoch);
            vals[t] = sum;
        }
        __syncthreads();
        if (S.xabort)
            xbail();
    };
    // S.acc[0..10) -> their sums over the shares (fp64, added in share order: the same bits in every share)
    auto xreduce_acc = [&]() {
        __syncthreads();
        ++xepoch;
        gu64 *bank = xsync + (size_t)(xepoch & 1u) * (SW_MG_MAX * SW_MG_WORDS);
        if (tid < 20)
            __hip_atomic_store(bank + J.grp * SW_MG_WORDS + tid,
                               ((unsigned long long)xepoch << 32) | reinterpret_cast<const unsigned *>(S.acc)[tid],
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        double tot = 0.0;
        if (tid < 10) {
            for (int w = 0; w < J.ngrp; ++w) {
                const unsigned lo = xpoll(bank + w * SW_MG_WORDS + 2 * tid, xepoch);
                const unsigned hi = xpoll(bank + w * SW_MG_WORDS + 2 * tid + 1, xepoch);
                tot += __hiloint2double((int)hi, (int)lo);
            }
        }
        __syncthreads(); // every word has been read from S.acc and published before S.acc is rewritten
        if (tid < 10)
            S.acc[tid] = tot;
        __syncthreads();
        if (S.xabort)
            xbail();
    };
    const int *__restrict__ grid = (grid_all != nullptr && (sw_cache & 4) != 0)
                                       ? grid_all + 64 * (size_t)__builtin_amdgcn_readfirstlane(S.tab.grid_off64) : nullptr;

    // ---- processing order of the queries: sorted by (strip, x) of their position under the guess, so that
    // the 64 lanes of a wave search for neighbours in space: same strips, windows of similar length, same LDS
    // lines -- the lockstep rounds below lose little to their slowest lane.  (A cloud in arbitrary order puts
    // a wall point next to an outlier in the same wave.)  The order only steers which lane searches for which
    // query: results are stored under the query's own index, and the sums of the error minimiser run in the
    // original order.  Sorted once per job, in the LDS that will hold the target (chunks of 8192). ----
    {
        unsigned long long *skeys = reinterpret_cast<unsigned long long *>(lds_tgt);
        float rloc = 0.0f; // largest |T0 * src| among this thread's queries (for the movement bounds of the clearance records)
        int smin_l = 0x7FFFFFFF, smax_l = -1; // WIN: the strips this thread's queries start in
        for (int c0 = 0; c0 < ns; c0 += sort_chunk) { // sort_chunk = the power of two of keys this LDS region holds
            const int n = min(sort_chunk, ns - c0);
            unsigned n2 = 2;
            while (n2 < (unsigned)n)
                n2 <<= 1;
            for (unsigned i = tid; i < n2; i += NT) {
                unsigned long long k = ~0ull;
                if (i < (unsigned)n) {
                    const float2 sp = src[c0 + i];
                    const float rx = affine1(T0[0], T0[1], T0[2], sp.x, sp.y);
                    const float ry = affine1(T0[3], T0[4], T0[5], sp.x, sp.y);
                    const int st_ = strip_of(ry, S.tab.ylo, S.tab.inv_g, S.tab.ns);
                    k = SW_KEY(st_, mono_key(rx), c0 + i);
                    if (WIN) {
                        smin_l = min(smin_l, st_);
                        smax_l = max(smax_l, st_);
                    }
                    const float rr = sqrtf(f_add(f_mul(rx, rx), f_mul(ry, ry)));
                    rloc = (rr > rloc || rr != rr) ? rr : rloc; // (a NaN sticks: no bound, no record is ever used)
                }
                skeys[i] = k;
            }
            __syncthreads();
            bitonic_sort_lds<NT>(skeys, n2);
            for (int i = tid; i < n; i += NT) {
                const int q = SW_KEY_ID(skeys[i]);
                Q.order[c0 + i] = q;
                Q.ssrc[c0 + i] = src[q];
                Q.slot_of[q] = c0 + i;
                Q.rec[c0 + i] = 0u; // no clearance record yet
            }
            __syncthreads();
        }
        atomicMax(&S.rmax_bits, __float_as_uint(rloc)); // rloc >= 0 or NaN (whose pattern is above every finite one)
        if (WIN && smax_l >= 0) {
            atomicMin(&S.win_smin, smin_l);
            atomicMax(&S.win_smax, smax_l);
        }
    }
    // sorted centred target (with its NaN sentinels: a NaN stops a walk direction) -> LDS
    if (LDS_TGT) {
        for (int i = tid; i < nt + SW_PAD; i += NT)
            lds_tgt[i] = stgt[i];
    }
    if constexpr (WIN) {
        // The window: whole strips, in the sorted cloud's own layout (the sentinel in front of the first strip and the one
        // behind the last included).  From the band of strips the queries start in -- shrunk from both ends if the
        // band alone exceeds the capacity (an unsplit job: its queries are everywhere), else grown by whole strips on both
        // sides while they fit.  Reads outside fall back to HBM, so any window is correct; a good one is fast.
        __syncthreads();
        if (tid == 0) {
            const int nst_ = S.tab.ns;
            int a = S.win_smin, b = S.win_smax + 1, lo = 0, n = 0;
            if (S.win_smax >= 0 && t_cap > 0) {
                auto len = [&](int a_, int b_) { return S.tab.sbeg[b_] - S.tab.sbeg[a_] + 1; };
                while (b - a > 1 && len(a, b) > t_cap) {
                    if ((b - a) & 1)
                        --b;
                    else
                        ++a;
                }
                if (len(a, b) <= t_cap) {
                    for (bool grow = true; grow;) {
                        grow = false;
                        if (a > 0 && len(a - 1, b) <= t_cap) {
                            --a;
                            grow = true;
                        }
                        if (b < nst_ && len(a, b + 1) <= t_cap) {
                            ++b;
                            grow = true;
                        }
                    }
                    lo = S.tab.sbeg[a] - 1;
                    n = len(a, b);
                }
            }
            S.win_lo = lo;
            S.win_n = n;
        }
        __syncthreads();
        const int wlo = __builtin_amdgcn_readfirstlane(S.win_lo), wn = __builtin_amdgcn_readfirstlane(S.win_n);
        for (int i = tid; i < wn; i += NT)
            lds_tgt[i] = stgt[wlo + i];
        T.lo = wlo;
        T.n = (unsigned)wn;
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
    const StripTab &tab = S.tab;
    const int nst = __builtin_amdgcn_readfirstlane(tab.ns);
    const float ylo = sw_uniform(tab.ylo), inv_g = sw_uniform(tab.inv_g);

    const float r2_match = sw_uniform(f_mul(P.matcher_max_dist, P.matcher_max_dist));
    const float r2_filter = sw_uniform(f_mul(P.max_dist_filter, P.max_dist_filter));
    // A search that has not met a target within maxDist yet is bounded by `best`, which starts at W2 = a
    // little MORE than maxDist^2 (found <=> best < r2m_up <=> best <= maxDist^2; with an unbounded
    // matcher: best finite): a query that ends `none` then knows its
    // nearest target is at least sqrt(best) > maxDist away, and that margin lets later iterations prove
    // "still none" from how far the query has moved instead of searching its whole maxDist window again.
    const float r2m_up = sw_uniform((r2_match < INFINITY) ? __uint_as_float(__float_as_uint(r2_match) + 1u) : INFINITY);
    const float W2 = sw_uniform(fmaxf(r2m_up, f_mul(r2_match, 1.1025f)));
    const float md_hi = sw_uniform(f_mul(P.matcher_max_dist, 1.00001f));
    // no pair beyond Cmax can get weight 1
    const float Cmax = P.use_max_dist_filter ? fminf(r2_filter, r2_match) : r2_match;
    float Cinit;
    {
        const float h = 8.0f * tab.ext_x / (float)nt; // a few point spacings of a cloud spread along x
        Cinit = h * h;
        if (!(Cinit > 1e-30f) || !(Cinit < Cmax))
            Cinit = Cmax;
        Cinit = sw_uniform(Cinit);
    }
    float Cnext = P.use_trimmed_filter ? Cinit : Cmax; // cap the next iteration starts with
    SW_PROF(0);

    const int sw_rtrips = (sw_cache >> 8) & 255;
    const bool sw_jump = (sw_cache & 2) != 0;
    const float sw_margin = 1.0f + 0.01f * (float)((sw_cache >> 16) & 255);
    int wd_outer = 0;
    // cur = Ti * (T0 * src): the same two roundings wherever a query is (re)computed
    auto xform = [&](const float (&Ti)[9], float2 sp) {
        const float rx = affine1(T0[0], T0[1], T0[2], sp.x, sp.y);
        const float ry = affine1(T0[3], T0[4], T0[5], sp.x, sp.y);
        return make_float2(affine1(Ti[0], Ti[1], Ti[2], rx, ry), affine1(Ti[3], Ti[4], Ti[5], rx, ry));
    };
    bool use_cache = false; // from the second iteration on: Q.pos / Q.st hold the previous iteration's results
    const bool sw_rec = REC && (sw_cache & 16) != 0; // clearance records (0: A/B without them)
    const float M2 = sw_uniform(f_mul(f_add(1.0f, sw_m), f_add(1.0f, sw_m)));
    int rec_epoch = 0;
    const float rmax = sw_uniform(f_mul(__uint_as_float(S.rmax_bits), 1.00001f));
    int it = 0; // iteration index (the records carry it in 6 bits: they are used while it < ICP_MAX_HIST = 64)
    while (true) {
        SW_WATCH(wd_outer, P.max_iter + 2, 0)
        float Ti[9];
#pragma unroll
        for (int i = 0; i < 9; ++i)
            Ti[i] = sw_uniform(S.Ti[i]);
        // movement bounds: |Ti x - Tk x| <= |A - Ak|_F |x| + |t - tk| for every query x = T0 * src, |x| <= rmax, plus
        // the fp32 rounding of the two transformed positions themselves
        if (sw_rec && it < ICP_MAX_HIST) {
            if (tid < it) {
                const float a0 = f_add(Ti[0], -S.thist[tid][0]), a1 = f_add(Ti[1], -S.thist[tid][1]);
                const float a3 = f_add(Ti[3], -S.thist[tid][3]), a4 = f_add(Ti[4], -S.thist[tid][4]);
                const float tx = f_add(Ti[2], -S.thist[tid][2]), ty = f_add(Ti[5], -S.thist[tid][5]);
                // largest singular value of the 2x2 difference: s^2 = (F^2 + sqrt(F^4 - 4 det^2)) / 2 (for two rotations
                // that is F / sqrt 2: the Frobenius norm alone would overstate the turn by 41 %); 1.001 for its rounding
                const float f2 = f_add(f_add(f_mul(a0, a0), f_mul(a1, a1)), f_add(f_mul(a3, a3), f_mul(a4, a4)));
                const float det = f_add(f_mul(a0, a4), -f_mul(a1, a3));
                const float disc = fmaxf(f_add(f_mul(f2, f2), -f_mul(4.0f, f_mul(det, det))), 0.0f);
                const float fa = f_mul(sqrtf(f_mul(0.5f, f_add(f2, sqrtf(disc)))), 1.001f);
                const float ft = sqrtf(f_add(f_mul(tx, tx), f_mul(ty, ty)));
                S.mva[tid] = f_mul(fa, 1.0001f);
                // the two positions themselves are rounded: 3 roundings each, relative to |a x| + |b y| + |c| <= 1.5 (|x| + |t|),
                // so 2 x 3 x 2^-24 x 1.5 (rmax + |t|) = 5.4e-7 (rmax + |t|) -- 3e-5 m up to a 50 m extent, scaled with the
                // data beyond (a fixed constant is only right for sonar-range coordinates)
                const float tmag = fmaxf(f_add(fabsf(Ti[2]), fabsf(Ti[5])), f_add(fabsf(S.thist[tid][2]), fabsf(S.thist[tid][5])));
                S.mvt[tid] = f_add(f_mul(ft, 1.0001f), fmaxf(3e-5f, f_mul(6e-7f, f_add(rmax, tmag))));
            } else if (tid == it) {
#pragma unroll
                for (int i = 0; i < 6; ++i)
                    S.thist[it][i] = Ti[i];
            }
        }
        // Per-iteration modes of the records (all wave-uniform; decided after the barrier below, where mvb is in place):
        //   rec_on: this iteration's searches use the margin and leave records -- only once the last step moved the
        //     queries by less than half of the largest margin a record can have (earlier no record would survive one
        //     iteration, and the margin costs ~17 % more candidates per search);
        //   records older than rec_epoch are ignored (searches of iterations without rec_on did not maintain them);
        //   triage: the fresh pass only tests the records, the misses are searched as dense waves by the second pass --
        //     when most queries of the previous iteration hit (a miss among 64 lanes makes the whole wave search).
        bool rec_on = false, triage = false;
        const int sw_umax = (sw_cache2 >> 8) & 0xFFFF;           // most points of a wave's union window
        const bool union_on = it < (sw_cache2 & 255) && sw_umax > 0; // iterations that use the union scan
        float mu2 = 0.0f; // additive part of the records' margin (squared): searched window = M2 * bound + mu2

        // ---- A+B: cur = Ti * (T0 * src); exact NN for every pair that can matter.  Round 0 searches
        // for every query (lane i handles the queries i, i + 1024, ... of the spatial order), later rounds search again for the
        // suspended ones with a larger cap. ----
        if (PROF && tid == 0)
            S.prof_b0 = clock64();
        // exact order statistic by radix select (4 passes of 8 bits over the distances' bit patterns): the k_sel-th
        // smallest (0-based) d2 among the exact matches, or among all finite ones (exact + inexact: an inexact
        // query holds an upper bound of its neighbour's distance)
        // `top_tallied`: S.hist0 already holds the histogram of the top byte (the census of the final round
        // counts it on the way), so the first of the four passes is skipped
        auto select_kth = [&](unsigned k_sel, bool all_finite, bool top_tallied) -> float {
        if (tid == 0) {
            S.sel_k = k_sel;
            S.sel_prefix = 0;
        }
        __syncthreads();
        for (int shift = 24; shift >= 0; shift -= 8) {
            const bool skip_tally = top_tallied && shift == 24;
            unsigned *hist = skip_tally ? S.hist0 : S.hist;
            if (!skip_tally) {
            for (int b = tid; b < 256; b += NT)
                S.hist[b] = 0;
            __syncthreads();
            }
            const unsigned prefix = S.sel_prefix;
            const unsigned himask = (shift == 24) ? 0u : (0xFFFFFFFFu << (shift + 8));
            auto tally = [&](int pz, float dz) { // called wave-uniformly
                unsigned bin = 0xFFFFFFFFu;      // no contribution
                if (all_finite ? (pz != SW_NONE) : (pz >= 0)) {
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
            if (!skip_tally) {
            for (int base = 0; base < ns; base += SW_NQ * NT) {
                int pz[SW_NQ];
                float dz[SW_NQ];
#pragma unroll
                for (int k = 0; k < SW_NQ; ++k) {
                    const int i = base + k * NT + tid;
                    pz[k] = i < ns ? Pz(i) : SW_NONE;
                    dz[k] = i < ns ? Dz(i) : INFINITY;
                }
#pragma unroll
                for (int k = 0; k < SW_NQ; ++k)
                    tally(pz[k], dz[k]);
            }
            __syncthreads();
            }
            if (MULTI)
                xreduce_u32(hist, 256); // the histogram of ALL shares (every share then picks the same bin)
            if (tid < 64) { // one wave: rank-in-histogram by shuffles instead of a 256-step serial walk
                const unsigned k = S.sel_k;
                const unsigned h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2],
                               h3 = hist[4 * lane + 3];
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
        return sw_uniform(__uint_as_float(S.sel_prefix));
        };
        float C = Cnext;
        unsigned nfin = 0, nexact = 0, ksel = 0;
        bool limit_inf = false;
        // The census of the iteration -- queries without a match, true neighbours within the cap, and the histogram
        // of the top byte of their distances (= the first pass of the radix select) -- is tallied where a query is
        // settled: `none` and `exact` are final for the iteration, so every query is counted once, in whatever round
        // and tier it ends.
        for (int b = tid; b < 256; b += NT)
            S.hist0[b] = 0;
        const unsigned rechit_prev = (it > 0) ? S.n_rechit[(it - 1) & 1] : 0u; // (written last iteration, barriers since)
        if (tid == 0) {
            S.n_none = 0;
            S.n_exact = 0;
            S.grid_skips = 0;
            S.n_rechit[it & 1] = 0u;
        }
        auto tally_settled = [&](bool is_none, bool is_exact, float best) { // called wave-uniformly
            const unsigned long long mn = __ballot(is_none), me = __ballot(is_exact);
            if (lane == 0) {
                if (mn)
                    atomicAdd(&S.n_none, (unsigned)__popcll(mn));
                if (me)
                    atomicAdd(&S.n_exact, (unsigned)__popcll(me));
            }
            // wave-aggregated: the exponent byte is the same for nearly every point
            const unsigned bin = is_exact ? (__float_as_uint(best) >> 24) : 0xFFFFFFFFu;
            unsigned long long todo = me;
            int wd6 = 0;
            while (todo) {
                SW_WATCH(wd6, 64, 6)
                const int leader = __builtin_amdgcn_readfirstlane(__ffsll((long long)todo) - 1);
                const unsigned b = (unsigned)__builtin_amdgcn_readlane((int)bin, leader);
                const unsigned long long same = __ballot(bin == b);
                if (lane == leader)
                    atomicAdd(&S.hist0[b], (unsigned)__popcll(same));
                todo &= ~same;
            }
        };
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
                if (round == 0 && sw_rec && use_cache && it < ICP_MAX_HIST) { // (it >= 1: mvb[it - 1] = the last step)
                    const float mv_last = sw_uniform(f_add(f_mul(S.mva[it - 1], rmax), S.mvt[it - 1]));
                    const float mu = f_mul(sw_kappa, mv_last);
                    rec_on = mu < 0.5f * sqrtf(C); // (NaN -> off)
                    mu2 = sw_uniform(f_mul(mu, mu));
                    // (the first iteration with records has no count yet: the margin was sized for them to hold)
                    triage = rec_on && rec_epoch < it && (rec_epoch == it - 1 || 2u * rechit_prev >= (unsigned)ns) &&
                             (sw_cache & 32) != 0;
                }
                const bool rec_use = rec_on && rec_epoch < it;
                const int *wl = Q.wl[cur];
                int *wl_next = Q.wl[cur ^ 1];
                // -- tier 1: one lane per query.  The first pass (fresh queries only) transforms the query,
                // evaluates last iteration's neighbour as a witness and walks the query's OWN strip with a
                // short budget: on converged clouds that settles ~85 % of the queries.  Whoever needs more
                // strips or more trips is COMPACTED into dense waves for the second pass, which goes through
                // all strips in lockstep rounds (pick a strip -> lower bound -> walk), so lanes that finished
                // early do not sit idle through the long searches of their neighbours.  What exhausts the
                // second budget too goes to tier 2. --
                auto walk_pass = [&](const int *list, int n, bool fresh, int budget, bool last) {
                float2 sp_next = make_float2(0, 0); // fresh pass: the next slice's source point is fetched a slice ahead
                int q_next = 0;                     // ... and so is its index
                if (fresh && tid < n) {
                    sp_next = Q.ssrc[tid];
                    q_next = Q.order[tid];
                }
                const float sC = f_mul(sqrtf(C), 1.00001f); // (the cap of this round as a radius, for the records)
                for (int k0 = 0; k0 < n; k0 += NT) {
                    const int slot = k0 + tid;
                    const bool valid = slot < n;
                    const float2 sp_cur = sp_next;
                    const int q_cur = q_next;
                    int prev = 0;
                    // clearance record of the query (0: none); fetched here, not a slice ahead like the source point:
                    // measured, the register that would carry it costs more than the latency of this coalesced read
                    const unsigned rec = (fresh && rec_use && valid) ? Q.rec[slot] : 0u;
                    if (fresh && use_cache && valid)
                        prev = Pz(q_cur); // last iteration's result of this query (used after the transform)
                    if (fresh && slot + NT < n) {
                        sp_next = Q.ssrc[slot + NT];
                        q_next = Q.order[slot + NT];
                    }
                    int q = 0, bpos = 0;
                    float px = 0, py = 0, best = W2;
                    float second = INFINITY; // smallest distance met so far to a target other than the (then) best one
                    bool tied = false, own_done = false;
                    if (valid) {
                        if (fresh) {
                            q = q_cur;
                            const float2 p = xform(Ti, sp_cur);
                            px = p.x;
                            py = p.y;
                        } else {
                            const unsigned e = (unsigned)list[slot];
                            q = (int)(e & SW_QMASK);
                            own_done = (e & SW_OWN_DONE) != 0;
                            tied = (e & SW_TIED) != 0;
                            const float2 p = xform(Ti, src[q]);
                            px = p.x;
                            py = p.y;
                            bpos = -2 - Pz(q);
                            best = Dz(q);
                            if (e & SW_PARTIAL) // handed on by the first pass: what it had seen
                                second = __int_as_float(Q.st[q].z);
                        }
                    }
                    // What the previous iteration knew about this query (the cloud moves little between
                    // iterations).  Its neighbour -- exact or not -- is evaluated first as a WITNESS: a real
                    // target at distance dw, so the search is "found" at once and bounded by min(dw, C)
                    // instead of running on until it meets some target within maxDist.  The witness counts
                    // as evaluated; the walk skips it when the cursors reach it (`!= bpos` below).
                    // A `none` query stays none as long as it has moved less than its recorded clearance:
                    // |p - t| >= |p0 - t| - |p - p0| > maxDist for every target t (1e-5 relative slop on
                    // each term, two orders above the rounding of the fp32 distances involved).
                    bool skip = false, grid_hit = false, grid_defer = false, rec_hit = false;
                    if (fresh && use_cache && valid) {
                        const int w = prev >= 0 ? prev + 1 : (prev <= -3 ? -2 - prev : 0);
                        if (PROF)
                            c_wit += (unsigned long long)__popcll(__ballot(w != 0));
                        if (w) {
                            const float2 t = T[w];
                            const float dxw = f_add(px, -t.x), dyw = f_add(py, -t.y);
                            const float dw = f_add(f_mul(dxw, dxw), f_mul(dyw, dyw));
                            if (dw < best) {
                                best = dw;
                                bpos = w;
                                // Clearance record: every OTHER target was at least R away when the record was taken and
                                // the query has moved by at most mvb since.  Still closer to its old neighbour than
                                // R - mvb: that one is the nearest target, strictly; or beyond the cap with every target
                                // provably beyond the cap: the search would end with exactly this upper bound.
                                if (rec != 0u && (int)(rec & 63u) >= rec_epoch && dw < r2m_up) {
                                    // |x| of the query's position x before T_iter: T_iter is a rotation + translation, so
                                    // |x| = |p - t| (to the 1e-6 by which its rounded matrix is not orthonormal)
                                    const float ux = f_add(px, -Ti[2]), uy = f_add(py, -Ti[5]);
                                    const float xr = f_mul(sqrtf(f_add(f_mul(ux, ux), f_mul(uy, uy))), 1.0001f);
                                    const float mv = f_add(f_mul(S.mva[rec & 63u], xr), S.mvt[rec & 63u]);
                                    const float Ro = f_add(__uint_as_float(rec & ~63u), -mv);
                                    rec_hit = f_mul(sqrtf(dw), 1.00001f) < Ro || (dw > C && sC < Ro);
                                }
                            }
                        } else if (prev == SW_NONE) {
                            const int4 r = Q.st[q];
                            const float mx0 = f_add(px, -__int_as_float(r.x)), my0 = f_add(py, -__int_as_float(r.y));
                            const float mv = sqrtf(f_add(f_mul(mx0, mx0), f_mul(my0, my0)));
                            skip = f_mul(mv, 1.00001f) < __int_as_float(r.z); // NaN -> search
                        }
                    } else if (fresh && valid && grid != nullptr) {
                        // first iteration: no previous neighbour yet -- the target point nearest to the centre of the
                        // query's grid cell (prep kernel) is the witness: a real target within half a cell diagonal
                        // of the best one, instead of whatever the own strip's x-walk happens to meet first
                        float gxv = f_mul(f_add(px, -tab.gx0), tab.ginv), gyv = f_mul(f_add(py, -tab.gy0), tab.ginv);
                        gxv = fminf(fmaxf(gxv, 0.0f), (float)(tab.gnx - 1)); // NaN -> 0
                        gyv = fminf(fmaxf(gyv, 0.0f), (float)(tab.gny - 1));
                        const int w = grid[(int)gyv * tab.gnx + (int)gxv];
                        if (PROF)
                            c_wit += (unsigned long long)__popcll(__ballot(w != 0));
                        if (w) {
                            const float2 t = T[w];
                            const float dxw = f_add(px, -t.x), dyw = f_add(py, -t.y);
                            const float dw = f_add(f_mul(dxw, dxw), f_mul(dyw, dyw));
                            if (dw < best) {
                                best = dw;
                                bpos = w;
                                // With a trimmed-distance filter the cap of this round is a placeholder (a few point
                                // spacings): hardly anything would be settled within it, so the query goes straight to the
                                // round whose cap comes from the witnesses.  Only while C < Cmax: at Cmax there is no
                                // further round, every query must be searched now.
                                grid_hit = P.use_trimmed_filter && C < Cmax && dw < r2m_up && (sw_cache & 8) != 0; // (a witness just beyond maxDist settles nothing)
                            }
                        }
                        // A query WITHOUT a usable witness (an empty cell far from every structure, a witness beyond maxDist)
                        // would now walk its whole maxDist window -- nothing bounds it until it meets a target -- while the
                        // witnessed queries of its wave, and after the pass the whole workgroup, wait for it.  It is put off
                        // to the same forced next round instead, where it searches next to everybody else.  (Suspended
                        // without any guarantee, like the witnessed ones; it holds no match: best >= r2m_up.)
                        grid_defer = !grid_hit && P.use_trimmed_filter && C < Cmax && (sw_cache & 8) != 0 && (sw_cache & 64) != 0 &&
                                     px == px && py == py;
                    }
                    const int so = strip_of(py, ylo, inv_g, nst);
                    int s_up = own_done ? so + 1 : so, s_dn = so - 1;
                    // a query with a NaN coordinate has no neighbour (every d2 is NaN): nothing to visit
                    bool lane_done = !valid || skip || grid_hit || grid_defer || rec_hit || !(px == px && py == py);
                    bool pending = false; // holds a strip it could not start or finish within the budget
                    bool own_fin = own_done;
                    // ---- union scan (the wide windows of the first iterations) ----
                    // While the clouds are still decimetres apart a query's window holds dozens of candidates and the
                    // lane-private walks run at a third of the lanes (a wave pays for its longest walk, ~200 instructions
                    // per evaluated candidate all told).  The 64 queries of a wave are neighbours in space (sorted by strip
                    // and x), so their windows overlap: the wave takes the UNION -- per strip the positions between
                    // min(px) - r and max(px) + r, r = the largest bound any of its lanes holds -- and every lane evaluates
                    // every point of it against its own query: broadcast LDS reads, no cursors, no divergence, 5.5 VALU per
                    // pair (chunk minima by v_min3, then the chunk that first attained the minimum is looked at again for
                    // the position; equal minima elsewhere = a possible tie, resolved the usual way).  A superset of every
                    // lane's own window, so each lane has searched completely when the scan ends.  Only lanes that hold a
                    // bound take part (a target within maxDist is known); a union beyond sw_umax points falls back to the walks.
                    if (LDS_TGT && union_on && !rec_on) { // (a search that leaves clearance records must know its runner-up: the walks do)
                        const bool part = !lane_done && best < r2m_up;
                        if (__ballot(part)) {
                            const float ru = f_add(f_mul(sqrtf(-wave_min(part ? -fminf(best, C) : 0.0f)), 1.0001f), 1e-6f);
                            const float ux0 = f_add(wave_min(part ? px : INFINITY), -ru), ux1 = f_add(-wave_min(part ? -px : INFINITY), ru);
                            const float uy0 = f_add(wave_min(part ? py : INFINITY), -ru), uy1 = f_add(-wave_min(part ? -py : INFINITY), ru);
                            const int s_lo = __builtin_amdgcn_readfirstlane(strip_of(uy0, ylo, inv_g, nst));
                            const int s_hi = __builtin_amdgcn_readfirstlane(strip_of(uy1, ylo, inv_g, nst));
                            // 64-ary search in [first, sent): first position whose x is not < xq (incl = false) / is > xq (incl = true)
                            auto coop_bound = [&](int first, int sent, float xq, bool incl) {
                                int lo = first, hi = sent;
                                for (int g2 = 0; g2 < 8 && hi - lo > 64; ++g2) {
                                    const int step = (hi - lo + 63) >> 6;
                                    const int pp = lo + lane * step;
                                    const bool inb = pp < hi;
                                    const float x = T[inb ? pp : lo].x;
                                    const int c = __popcll(__ballot(inb && (incl ? x <= xq : x < xq)));
                                    if (c == 0) {
                                        hi = lo;
                                    } else {
                                        const int nlo = lo + (c - 1) * step + 1;
                                        hi = min(lo + c * step, hi);
                                        lo = nlo;
                                    }
                                }
                                const int pp = lo + lane;
                                const bool inb = pp < hi;
                                const float x = T[inb ? pp : lo].x;
                                return lo + __popcll(__ballot(inb && (incl ? x <= xq : x < xq)));
                            };
                            // the union's range in every strip it touches (lane k keeps strip s_lo + k's), and its size
                            int my_lo = 0, my_hi = 0, total = 0;
                            for (int su = s_lo; su <= s_hi; ++su) {
                                const int first = tab.sbeg[su], sent = tab.sbeg[su + 1] - 1;
                                const int a = coop_bound(first, sent, ux0, false), b = coop_bound(first, sent, ux1, true);
                                if (lane == su - s_lo) {
                                    my_lo = a;
                                    my_hi = b;
                                }
                                total += max(b - a, 0);
                            }
                            if (total <= sw_umax) {
                                if (PROF)
                                    c_eval += (unsigned long long)total * (unsigned long long)__popcll(__ballot(part));
                                float bs = __uint_as_float(__float_as_uint(best) + 1u); // (the witness is found again like any other point)
                                int bch = -1, bsent = 0;
                                bool eqc = false;
                                for (int su = s_lo; su <= s_hi; ++su) {
                                    const int a = __builtin_amdgcn_readlane(my_lo, su - s_lo), b = __builtin_amdgcn_readlane(my_hi, su - s_lo);
                                    const int sent = tab.sbeg[su + 1] - 1; // the strip's NaN sentinel pads its last chunk
                                    for (int jb = a; jb < b; jb += 16) {
                                        float cmin = INFINITY;
#pragma unroll
                                        for (int k = 0; k < 16; k += 2) {
                                            const float2 t0 = T[min(jb + k, sent)], t1 = T[min(jb + k + 1, sent)];
                                            cmin = fminf(fminf(cmin, dist2(px, py, t0.x, t0.y)), dist2(px, py, t1.x, t1.y));
                                        }
                                        if (cmin < bs) {
                                            bs = cmin;
                                            bch = jb;
                                            bsent = sent;
                                            eqc = false;
                                        } else if (cmin == bs) {
                                            eqc = true;
                                        }
                                    }
                                }
                                if (part && bch >= 0) {
                                    int hits = 0, pos = 0;
                                    for (int k = 0; k < 16; ++k) {
                                        const int j = min(bch + k, bsent);
                                        const float2 t = T[j];
                                        if (dist2(px, py, t.x, t.y) == bs) {
                                            pos = hits ? pos : j;
                                            ++hits;
                                        }
                                    }
                                    best = bs;
                                    bpos = pos;
                                    tied = eqc || hits > 1;
                                    lane_done = true; // searched completely
                                }
                            }
                        }
                    }
                    // A query that comes to a later pass still WITHOUT any target within maxDist (put off without a
                    // witness, or nothing met within the first pass's budget) has nothing that bounds its search: its
                    // window is the whole maxDist box, hundreds to thousands of candidates, and the other 63 lanes of its
                    // wave would wait while it walks them four at a time until the budget runs out.  It goes to the
                    // cooperative tier at once (256 candidates per trip).
                    if (!fresh && last && !lane_done && !(best < r2m_up) && (sw_cache & 128) != 0) {
                        pending = true;
                        lane_done = true;
                    }
                    int used = 0;
                    auto pick = [&]() { // the lane's next strip, -1 (and lane_done) when nothing is left within its bound
                        int s = -1;
                        if (!lane_done) {
                            const float capv = (best < r2m_up) ? C : best; // nothing within maxDist yet: only `best` bounds the search
                            const float sb = best < capv ? best : capv;
                            s = next_strip(tab, nst, so, s_up, s_dn, py, rec_on ? f_add(f_mul(sb, M2), mu2) : sb);
                            lane_done = s < 0;
                        }
                        return s;
                    };
                    // Rounds: lanes that are between strips pick their next one and find their x position in it
                    // (lower bound), then everybody walks for at most `rtrips` trips; a lane that is not through
                    // its strip by then simply keeps walking in the next round while its neighbours move on to
                    // their next strips -- the wave pays for its slowest LANE (sum over that lane's strips), not
                    // for the slowest lane of every round.
                    const int rtrips = fresh ? budget : sw_rtrips;
                    if (rec_use && fresh) { // (wave-uniform branch)
                        const unsigned long long mh = __ballot(rec_hit);
                        if (mh && lane == 0)
                            atomicAdd(&S.n_rechit[it & 1], (unsigned)__popcll(mh));
                    }
                    int s = -1;
                    if (fresh && triage)
                        pending = !lane_done; // not searched here: the second pass takes the misses as dense waves
                    else
                        s = pick();
                    int iL = 0, iR = 0;
                    bool fin = true; // not inside a strip
                    for (int rnd = 0; rnd < 4096; ++rnd) {
                        if (!__ballot(s >= 0))
                            break;
                        if ((fresh && rnd >= 1) || used >= budget) { // out of budget: whoever still holds a strip is handed on
                            pending = pending || s >= 0; // (a lane sent straight to the cooperative tier holds no strip and stays pending)
                            break;
                        }
                        ++used;
                        const bool start = fin && s >= 0;
                        if (__ballot(start)) {
                            if (PROF) { // ceil(log2(strip population + 1)) probes per starting lane
                                const int len_ = start ? tab.sbeg[s + 1] - 1 - tab.sbeg[s] : 0;
                                int steps_ = len_ > 0 ? 32 - __clz(len_) : 0;
                                for (int o_ = 32; o_ > 0; o_ >>= 1)
                                    steps_ += __shfl_xor(steps_, o_);
                                c_lb += (unsigned long long)__builtin_amdgcn_readfirstlane(steps_);
                            }
                            const int lo = strip_lower_bound(T, start ? tab.sbeg[s] : 0, start ? tab.sbeg[s + 1] - 1 : 0, px);
                            if (start) {
                                iR = lo;
                                iL = lo - 1;
                                fin = false;
                            }
                        }
                        for (int trip = 0; trip < rtrips && __ballot(!fin); ++trip) {
                            ++used;
                            if (PROF)
                                c_eval += 4ull * (unsigned long long)__popcll(__ballot(!fin)); // 2 sub-steps x 2 cursors
                            if (!fin) {
#pragma unroll
                                for (int s2 = 0; s2 < 2; ++s2) {
                                    const float2 tl = T[iL], tr = T[iR];
                                    const float dxl = f_add(px, -tl.x), el = f_mul(dxl, dxl);
                                    const float dyl = f_add(py, -tl.y), dl = f_add(el, f_mul(dyl, dyl));
                                    const float dxr = f_add(px, -tr.x), er = f_mul(dxr, dxr);
                                    const float dyr = f_add(py, -tr.y), dr = f_add(er, f_mul(dyr, dyr));
                                    const float capv = (best < r2m_up) ? C : best;
                                    float sb = best < capv ? best : capv;   // stop bound (no NaNs here: plain select)
                                    if (rec_on)
                                        sb = f_add(f_mul(sb, M2), mu2); // (the records' margin: look a little further than necessary)
                                    const bool okl = el <= sb, okr = er <= sb;   // NaN sentinel -> false
                                    // a candidate at the position of the current best is the best itself (a witness,
                                    // or a point met again by a search that started over): never a tie, never a runner-up
                                    tied |= okl && (dl == best) && (iL != bpos);
                                    if (rec_on && okl && iL != bpos)
                                        second = fminf(second, fmaxf(dl, best));
                                    if (okl && dl < best) {
                                        best = dl;
                                        bpos = iL;
                                    }
                                    tied |= okr && (dr == best) && (iR != bpos);
                                    if (rec_on && okr && iR != bpos)
                                        second = fminf(second, fmaxf(dr, best));
                                    if (okr && dr < best) {
                                        best = dr;
                                        bpos = iR;
                                    }
                                    iL -= okl ? 1 : 0;
                                    iR += okr ? 1 : 0;
                                    fin = !(okl || okr);
                                }
                            }
                        }
                        if (fin && s >= 0) { // through this strip: the next one, or done
                            own_fin |= s == so;
                            s = pick();
                        }
                    }
                    // classify: none / exact / suspended (inexact) / unfinished (handed to the next tier)
                    const bool is_long = valid && pending;
                    const bool settled = valid && !is_long;
                    const bool found = best < r2m_up; // <=> some target with d2 <= maxDist^2 was met (best starts at W2 >= r2m_up)
                    const bool is_none = settled && !found && !grid_defer;
                    // (a grid-witnessed query of the first iteration has not searched anything yet: never exact)
                    const bool is_exact = settled && found && best <= C && !grid_hit;
                    const bool is_susp = settled && ((found && (!(best <= C) || grid_hit)) || grid_defer);
                    if (is_none) {
                        setQ(q, INFINITY, SW_NONE);
                        if (!skip) // a full search: every target is at least sqrt(best) away from (px, py)
                            Q.st[q] = make_int4(__float_as_int(px), __float_as_int(py),
                                                __float_as_int(f_add(f_mul(sqrtf(best), 0.99999f), -md_hi)), 0);
                    }
                    if (is_exact) {
                        if (tied)
                            bpos = sweep_resolve_tie(T, tab, Q, px, py, best);
                        setQ(q, best, bpos - 1);
                    }
                    if (is_susp || is_long)
                        setQ(q, best, SW_INEXACT_OF(bpos));
                    if (rec_on && (is_exact || is_susp) && !rec_hit) {
                        // a finished search: everything within sqrt(M2 x its final bound) has been evaluated
                        unsigned r = 0u;
                        if (!grid_hit && it < ICP_MAX_HIST) {
                            const float capv = (best < r2m_up) ? C : best;
                            const float edge = f_add(f_mul(best < capv ? best : capv, M2), mu2);
                            const float R = f_mul(sqrtf(fminf(second, edge)), 0.99999f);
                            r = (__float_as_uint(R) & ~63u) | (unsigned)it;
                            if (!(R > 0.0f))
                                r = 0u;
                        }
                        Q.rec[fresh ? slot : Q.slot_of[q]] = r;
                    }
                    if (rec_on && is_long) { // the next tier continues from what this one has seen; no record meanwhile
                        Q.st[q].z = __float_as_int(second);
                        Q.rec[fresh ? slot : Q.slot_of[q]] = 0u;
                    }
                    tally_settled(is_none, is_exact, best);
                    if (__ballot(grid_hit || grid_defer) && lane == 0)
                        S.grid_skips = 1;
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
                        if (ml) {
                            int base = 0;
                            if (lane == 0)
                                base = atomicAdd(last ? &S.long_n : &S.mid_n, __popcll(ml));
                            base = __builtin_amdgcn_readfirstlane(base);
                            if (is_long) {
                                // the own strip may be skipped by the next pass only if it was finished here
                                // (everything in it within the then larger bound has been evaluated)
                                const unsigned e = (unsigned)q | (own_fin ? SW_OWN_DONE : 0u) | (tied ? SW_TIED : 0u) |
                                                   ((rec_on && !last) ? SW_PARTIAL : 0u);
                                (last ? Q.lng : Q.mid)[base + __popcll(ml & below)] = (int)e;
                            }
                        }
                    }
                }
                };
                // The fresh pass of an iteration in triage mode (most queries are settled by their clearance records): the same
                // decisions as walk_pass(fresh) takes when it does not search -- transform, witness, record / clearance
                // test, classification, the misses handed to the second pass -- as a loop of its own.  walk_pass carries
                // the whole search state through its body (the 64-VGPR builds spill ~85 registers there); here a lane holds
                // a dozen values.  A late iteration of the 5000 x 5000 job spent 55 k of its 134 k cycles in this pass.
                auto triage_pass = [&]() {
                    const float sC = f_mul(sqrtf(C), 1.00001f);
                    int *wl_next = Q.wl[cur ^ 1];
                    // (query index, source point, record) of the next slice are requested a slice ahead: three words from HBM
                    // scratch whose latency would otherwise stand in front of every slice
                    int q_n = 0;
                    float2 sp_n = make_float2(0, 0);
                    unsigned rec_n = 0u;
                    if (tid < ns) {
                        q_n = Q.order[tid];
                        sp_n = Q.ssrc[tid];
                        rec_n = Q.rec[tid];
                    }
                    for (int k0 = 0; k0 < ns; k0 += NT) {
                        const int slot = k0 + tid;
                        const bool valid = slot < ns;
                        int q = 0, bpos = 0;
                        float px = 0, py = 0, best = W2;
                        bool skip = false, rec_hit = false;
                        const int q_c = q_n;
                        const float2 sp = sp_n;
                        const unsigned rec = rec_n;
                        if (slot + NT < ns) {
                            q_n = Q.order[slot + NT];
                            sp_n = Q.ssrc[slot + NT];
                            rec_n = Q.rec[slot + NT];
                        }
                        if (valid) {
                            q = q_c;
                            const int prev = Pz(q);
                            const float2 p = xform(Ti, sp);
                            px = p.x;
                            py = p.y;
                            const int w = prev >= 0 ? prev + 1 : (prev <= -3 ? -2 - prev : 0);
                            if (w) {
                                const float2 t = T[w];
                                const float dxw = f_add(px, -t.x), dyw = f_add(py, -t.y);
                                const float dw = f_add(f_mul(dxw, dxw), f_mul(dyw, dyw));
                                if (dw < best) {
                                    best = dw;
                                    bpos = w;
                                    if (rec != 0u && (int)(rec & 63u) >= rec_epoch && dw < r2m_up) {
                                        const float ux = f_add(px, -Ti[2]), uy = f_add(py, -Ti[5]);
                                        const float xr = f_mul(sqrtf(f_add(f_mul(ux, ux), f_mul(uy, uy))), 1.0001f);
                                        const float mv = f_add(f_mul(S.mva[rec & 63u], xr), S.mvt[rec & 63u]);
                                        const float Ro = f_add(__uint_as_float(rec & ~63u), -mv);
                                        rec_hit = f_mul(sqrtf(dw), 1.00001f) < Ro || (dw > C && sC < Ro);
                                    }
                                }
                            } else if (prev == SW_NONE) {
                                const int4 r = Q.st[q];
                                const float mx0 = f_add(px, -__int_as_float(r.x)), my0 = f_add(py, -__int_as_float(r.y));
                                const float mv = sqrtf(f_add(f_mul(mx0, mx0), f_mul(my0, my0)));
                                skip = f_mul(mv, 1.00001f) < __int_as_float(r.z); // NaN -> search
                            }
                        }
                        {
                            const unsigned long long mh = __ballot(rec_hit);
                            if (mh && lane == 0)
                                atomicAdd(&S.n_rechit[it & 1], (unsigned)__popcll(mh));
                        }
                        const bool lane_done = !valid || skip || rec_hit || !(px == px && py == py);
                        const bool is_long = valid && !lane_done; // a miss: the second pass searches it
                        const bool settled = valid && lane_done;
                        const bool found = best < r2m_up;
                        const bool is_none = settled && !found;
                        const bool is_exact = settled && found && best <= C;
                        const bool is_susp = settled && found && !(best <= C);
                        if (is_none) {
                            setQ(q, INFINITY, SW_NONE);
                            if (!skip)
                                Q.st[q] = make_int4(__float_as_int(px), __float_as_int(py),
                                                    __float_as_int(f_add(f_mul(sqrtf(best), 0.99999f), -md_hi)), 0);
                        }
                        if (is_exact)
                            setQ(q, best, bpos - 1);
                        if (is_susp || is_long)
                            setQ(q, best, SW_INEXACT_OF(bpos));
                        if (is_long) { // (nothing visited yet: no runner-up, no record meanwhile)
                            Q.st[q].z = __float_as_int(INFINITY);
                            Q.rec[slot] = 0u;
                        }
                        tally_settled(is_none, is_exact, best);
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
                        if (ml) {
                            int base = 0;
                            if (lane == 0)
                                base = atomicAdd(&S.mid_n, __popcll(ml));
                            base = __builtin_amdgcn_readfirstlane(base);
                            if (is_long)
                                Q.mid[base + __popcll(ml & below)] = (int)((unsigned)q | SW_PARTIAL);
                        }
                    }
                };
                if (round == 0) {
                    if (REC && triage && (sw_cache & (1 << 25)) != 0)
                        triage_pass();
                    else
                        walk_pass(wl, nwork, true, sw_budget_a, false);
                    __syncthreads();
                    SW_PROF(1);
                    if (PROF && tid == 0)
                        S.prof[11] += S.mid_n;
                    walk_pass(Q.mid, S.mid_n, false, sw_budget, true);
                } else {
                    walk_pass(wl, nwork, false, sw_budget, true);
                }
                __syncthreads();
                SW_PROF(6);
                const int nlong = S.long_n;
                if (PROF && tid == 0) {
                    S.prof[9] += 1;
                    S.prof[10] += nlong;
                }
                // -- tier 2: one wave per long search, from scratch but bounded by the best it holds: strip
                // by strip (wave-uniform control flow), 64-ary lower bound, then 128 candidates per side and
                // trip; the next query is fetched while this one is searched --
                if (nlong > 0) {
                    const bool left = lane < 32;
                    // queries are handed out dynamically (their lengths differ by orders of magnitude)
                    auto next_slot = [&]() {
                        int v = 0;
                        if (lane == 0)
                            v = atomicAdd(&S.long_next, 1);
                        return __builtin_amdgcn_readfirstlane(v);
                    };
                    int slot = next_slot();
                    int qn = 0, posn = 0;
                    float2 pn = make_float2(0, 0);
                    float bn = 0;
                    auto fetch = [&](int sl) {
                        qn = (int)((unsigned)Q.lng[sl] & SW_QMASK);
                        pn = src[qn];
                        bn = Dz(qn);
                        posn = Pz(qn);
                    };
                    if (slot < nlong)
                        fetch(slot);
                    int wd2 = 0;
                    while (slot < nlong) {
                        SW_WATCH(wd2, ns + 2, 5)
                        long long tp0 = 0;
                        if (PROF && tid == 0)
                            tp0 = clock64();
                        const int q = __builtin_amdgcn_readfirstlane(qn);
                        const float2 pq = xform(Ti, make_float2(sw_uniform(pn.x), sw_uniform(pn.y)));
                        const float px = sw_uniform(pq.x), py = sw_uniform(pq.y);
                        float best = sw_uniform(bn);
                        int bpos = -2 - __builtin_amdgcn_readfirstlane(posn);
                        bool tied = false;
                        slot = next_slot();
                        if (slot < nlong) // prefetch the next query
                            fetch(slot);
                        if (PROF && tid == 0) {
                            const long long t_ = clock64();
                            S.prof[13] += t_ - tp0;
                            tp0 = t_;
                        }
                        constexpr int U = 4; // candidates per lane and trip: four independent LDS reads in flight
                        const int lo32 = left ? lane : lane - 32;
                        const int so = strip_of(py, ylo, inv_g, nst);
                        int s_up = so, s_dn = so - 1;
                        for (int rnd = 0; rnd < 2 * SW_NS_MAX + 4; ++rnd) {
                            int s;
                            {
                                const float capv = (best < r2m_up) ? C : best;
                                const float sb = best < capv ? best : capv;
                                s = __builtin_amdgcn_readfirstlane(next_strip(tab, nst, so, s_up, s_dn, py, sb));
                                s_up = __builtin_amdgcn_readfirstlane(s_up);
                                s_dn = __builtin_amdgcn_readfirstlane(s_dn);
                            }
                            if (s < 0)
                                break;
                            const int first = tab.sbeg[s], sent = tab.sbeg[s + 1] - 1; // points [first, sent)
                            if (sent <= first)
                                continue;
                            // 64-ary lower bound of px in the strip
                            int lo = first, hi = sent;
                            for (int g2 = 0; g2 < 8 && hi - lo > 64; ++g2) {
                                const int step = (hi - lo + 63) >> 6;
                                const int pp = lo + lane * step;
                                const bool inb = pp < hi;
                                const float x = T[inb ? pp : lo].x;
                                const int c = __popcll(__ballot(inb && x < px));
                                if (c == 0) {
                                    hi = lo;
                                } else {
                                    const int nlo = lo + (c - 1) * step + 1;
                                    hi = min(lo + c * step, hi);
                                    lo = nlo;
                                }
                            }
                            {
                                const int pp = lo + lane;
                                const bool inb = pp < hi;
                                const float x = T[inb ? pp : lo].x;
                                lo += __popcll(__ballot(inb && x < px));
                            }
                            int iL = lo - 1, iR = lo;
                            bool doneL = false, doneR = false;
                            for (int guard = 0; guard <= (sent - first) / (32 * U) + 2; ++guard) { // bounded by construction
                                const float capv = (best < r2m_up) ? C : best;
                                const float sb = best < capv ? best : capv; // stop bound at the start of the trip
                                const bool on = left ? !doneL : !doneR;
                                float d = INFINITY;
                                int jbest = 0, nL = 0, nR = 0;
                                bool eqf = false;
#pragma unroll
                                for (int u = 0; u < U; ++u) {
                                    // clamped onto the strip's own sentinels
                                    const int j = left ? max(iL - lo32 - 32 * u, first - 1) : min(iR + lo32 + 32 * u, sent);
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
                                    const bool use = cons && j != bpos; // the best it holds is already accounted for
                                    if (use && du < d) { // NaN never passes
                                        d = du;
                                        jbest = j;
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
                                    const int firstl = __builtin_amdgcn_readfirstlane(__ffsll((long long)who) - 1);
                                    bpos = __builtin_amdgcn_readlane(jbest, firstl);
                                    best = wmin;
                                } else if (wmin == best && wmin < INFINITY) {
                                    tied = true;
                                }
                                iL -= nL;
                                iR += nR;
                                doneL |= nL < 32 * U;
                                doneR |= nR < 32 * U;
                                if (PROF && lane == 0)
                                    atomicAdd((unsigned long long *)&S.prof[12], 1ull);
                                if (PROF)
                                    c_coop += 64ull * U;
                                if (doneL && doneR)
                                    break;
                            }
                        }
                        if (PROF && tid == 0) {
                            const long long t_ = clock64();
                            S.prof[14] += t_ - tp0;
                            tp0 = t_;
                        }
                        if (lane == 0) {
                            if (!(best < r2m_up)) {
                                setQ(q, INFINITY, SW_NONE);
                                Q.st[q] = make_int4(__float_as_int(px), __float_as_int(py),
                                                    __float_as_int(f_add(f_mul(sqrtf(best), 0.99999f), -md_hi)), 0);
                                atomicAdd(&S.n_none, 1u);
                            } else if (best <= C) {
                                if (tied)
                                    bpos = sweep_resolve_tie(T, tab, Q, px, py, best);
                                setQ(q, best, bpos - 1);
                                atomicAdd(&S.n_exact, 1u);
                                atomicAdd(&S.hist0[__float_as_uint(best) >> 24], 1u);
                            } else {
                                setQ(q, best, SW_INEXACT_OF(bpos));
                                wl_next[atomicAdd(&S.wl_n[cur ^ 1], 1)] = q;
                            }
                        }
                        if (PROF && tid == 0)
                            S.prof[15] += clock64() - tp0;
                    }
                }
                __syncthreads();
                SW_PROF(7);
                if (PROF && tid == 0 && chk.iters == 0 && round < 8) { // first iteration, round by round
                    const long long t_ = clock64();
                    S.prof_it[24 + 4 * round] = nwork;
                    S.prof_it[25 + 4 * round] = (round == 0) ? S.mid_n : 0;
                    S.prof_it[26 + 4 * round] = nlong;
                    S.prof_it[27 + 4 * round] = t_ - S.prof_b0;
                }
                // -- census (tallied on the way, see tally_settled) --
                const int nsusp_mine = S.wl_n[cur ^ 1]; // this workgroup's suspended queries: the next round's work list
                int nsusp = nsusp_mine;                 // ... and those of the whole job: what the round decides from
                bool grid_skipped = S.grid_skips != 0;
                if (MULTI) {
                    if (tid == 0) {
                        S.xr[0] = S.n_none;
                        S.xr[1] = S.n_exact;
                        S.xr[2] = (unsigned)nsusp_mine;
                        S.xr[3] = grid_skipped ? 1u : 0u;
                        S.xr[4] = (unsigned)ns;
                    }
                    xreduce_u32(S.xr, 5);
                    nfin = S.xr[4] - S.xr[0];
                    nexact = S.xr[1];
                    nsusp = (int)S.xr[2];
                    grid_skipped = S.xr[3] != 0u;
                } else {
                    nfin = (unsigned)ns - S.n_none;
                    nexact = S.n_exact;
                }
                SW_PROF(8);
                bool done;
                if (P.use_trimmed_filter && nfin > 0) {
                    ksel = (P.trim_ratio >= 1.0f) ? nfin - 1 : (unsigned)f_mul((float)nfin, P.trim_ratio);
                    done = nexact > ksel;
                } else {
                    done = nsusp == 0;
                }
                // queries that took a grid witness instead of searching are suspended WITHOUT the guarantee "neighbour
                // beyond the cap" the other suspended ones carry: the round that searches them must follow, whatever
                // the census says (they exist only in round 0 of the first iteration, and only while C < Cmax)
                if (round == 0 && grid_skipped && nsusp != 0)
                    done = false;
                if (!done && (C >= Cmax || nsusp == 0)) {
                    // the k-th finite distance exceeds MaxDist^2 (or every neighbour is already known)
                    limit_inf = C >= Cmax && nsusp != 0;
                    done = true;
                }
                if (done)
                    break;
                if (P.use_trimmed_filter && sw_jump) {
                    // Every finite query holds an upper bound U of its neighbour's distance (exact ones the distance
                    // itself).  The k-th smallest U is >= the k-th smallest distance, so with C = that value the next
                    // round is the last one: at least k+1 queries have their neighbour within C.
                    const float uk = select_kth(ksel, true, false);
                    C = sw_uniform(fminf(fmaxf(uk, C), Cmax));
                } else {
                    C = sw_uniform((round >= 12) ? Cmax : fminf(fmaxf(4.0f * C, Cinit), Cmax));
                }
                cur ^= 1;
                nwork = nsusp_mine;
            }
        }
        SW_PROF(2);
        if (PROF && tid == 0 && chk.iters < 32) {
            S.prof_it[2 * chk.iters] = clock64() - S.prof_b0;
            S.prof_it[2 * chk.iters + 1] = ((long long)__float_as_uint(C) << 32) | (nexact & 0xFFFFu) |
                                           ((S.n_rechit[it & 1] & 0xFFFFu) << 16); // (clouds of < 65536 points)
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
                limit = select_kth(ksel, false, true);
            }
        }
        __syncthreads();
        if (fail)
            break;
        // the next iteration's cap: this limit plus a margin (the clouds keep moving a little: without it about
        // every fourth converged iteration finds one match too few inside the cap and has to search twice)
        Cnext = P.use_trimmed_filter ? fminf(fmaxf(limit * sw_margin, Cinit * 0.0625f), Cmax) : Cmax;
        SW_PROF(3);

        // ---- D: error minimiser sums over the kept pairs, in two halves of five accumulators: ten fp64
        // accumulators per lane do not fit the 64-VGPR budget next to the loop state (they spilled) ----
        // The order of these sums is that of a 1024-thread workgroup whatever NT is: query i belongs to thread i mod 1024,
        // 64 consecutive threads are a wave (its fixed tree), the 16 wave totals are added left to right.  The smaller
        // builds play those waves one after the other.  (On a rank-deficient problem -- a target of three points -- the
        // sums are rounding noise that the solve amplifies without bound: only the same order gives the same result
        // as the other kernels; tools/icp_soak.py found such jobs at 6 in 100 000 before.)
        auto sums = [&](auto lo_tag) {
            constexpr int LO = decltype(lo_tag)::value;
            auto terms = [&](int i, double (&a5)[5]) {
                const int id = Pz(i);
                const float d = Dz(i);
                const bool ok = id >= 0 && (!P.use_max_dist_filter || d <= r2_filter) &&
                                (!P.use_trimmed_filter || d <= limit);
                if (!ok)
                    return;
                const float2 p = xform(Ti, src[i]);
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
            };
            if constexpr (NT == 1024) {
                double a5[5] = {0, 0, 0, 0, 0};
                for (int i = tid; i < ns; i += NT)
                    terms(i, a5);
                block_sum<5, NT>(a5, S.red);
                if (tid == 0) {
#pragma unroll
                    for (int k = 0; k < 5; ++k)
                        S.acc[LO + k] = a5[k];
                }
            } else {
                const int wave = tid >> 6;
                const int roles = min(16, (ns + 63) >> 6); // (the waves beyond hold no query: their totals are 0.0)
                for (int w0 = wave; w0 < roles; w0 += NT / 64) {
                    double a5[5] = {0, 0, 0, 0, 0};
                    for (int i = 64 * w0 + lane; i < ns; i += 1024)
                        terms(i, a5);
#pragma unroll
                    for (int k = 0; k < 5; ++k) {
                        const double sk = wave_sum(a5[k]);
                        if (lane == 0)
                            S.red[5 * w0 + k] = sk;
                    }
                }
                __syncthreads();
                if (tid < 5) {
                    double sk = 0;
                    for (int w = 0; w < roles; ++w)
                        sk += S.red[5 * w + tid];
                    S.acc[LO + tid] = sk;
                }
                __syncthreads();
            }
        };
        sums(std::integral_constant<int, 0>());
        sums(std::integral_constant<int, 5>());
        if (MULTI)
            xreduce_acc(); // the sums over the queries of every share
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
        use_cache = (sw_cache & 1) != 0;
        if (!rec_on)
            rec_epoch = it + 1;
        ++it;
    }

    if (tid == 0 && (!MULTI || J.grp == 0)) { // (every share of a split job holds the same result: share 0 reports it)
        const int status = S.flag_status;
        float *To = T_out + 9 * (size_t)J.out;
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
        status_out[J.out] = status;
        iters_out[J.out] = chk.iters;
        if (PROF && blockIdx.x == 0) {
            for (int i = 0; i < 16; ++i)
                prof[i] = S.prof[i];
            for (int i = 0; i < 64; ++i)
                prof[16 + i] = S.prof_it[i];
        }
    }
    if (PROF && lane == 0) { // launch-wide work counts: [80] lane-tier evaluations, [81] cooperative tier, [82] witnesses,
                             // [83] lower-bound probes, [84] iterations run
        atomicAdd((unsigned long long *)&prof[80], c_eval);
        atomicAdd((unsigned long long *)&prof[81], c_coop);
        atomicAdd((unsigned long long *)&prof[82], c_wit);
        atomicAdd((unsigned long long *)&prof[83], c_lb);
        if (tid == 0)
            atomicAdd((unsigned long long *)&prof[84], (unsigned long long)chk.iters);
    }
}

#include "sfe_icp_tiny.h"

// ---------------------------------------------------------------------------------------------
// split: one workgroup per job that is shared by several workgroups of the loop kernel (MULTI)
// ---------------------------------------------------------------------------------------------
// The queries are dealt to the job's `ngrp` shares by the strip of their position under the guess -- bands of strips
// holding about the same number of queries -- and every share's source points are gathered into one contiguous slice
// (original order inside a share), so that a share is an ordinary job record on a cloud of its own.  The records of
// the shares (jobs[first .. first + ngrp), filled by the host with the caller's cloud) get their n_src, src_start (in
// the gathered cloud) and q_off here.
template <int NT>
__global__ __launch_bounds__(NT) void icp_split_kernel(SweepJob *__restrict__ jobs, const int *__restrict__ split_first,
                                                       const float2 *__restrict__ src_all, const float *__restrict__ guess_all,
                                                       const float *__restrict__ mean_all, const StripTab *__restrict__ tab_all,
                                                       float2 *__restrict__ gsrc_all)
{
    __shared__ int s_cnt[SW_NS_MAX], s_grp[SW_NS_MAX], s_goff[SW_MG_MAX + 1], s_run[SW_MG_MAX];
    __shared__ int s_w[NT / 64][SW_MG_MAX];
    const int j0 = split_first[blockIdx.x];
    const SweepJob J = jobs[j0];
    const int n = J.n_src, G = min(J.ngrp, SW_MG_MAX), tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float2 *__restrict__ src = src_all + J.src_start;
    const float mx = mean_all[2 * J.prep], my = mean_all[2 * J.prep + 1];
    const StripTab *tab = tab_all + J.prep;
    const float ylo = tab->ylo, inv_g = tab->inv_g;
    const int nst = tab->ns;
    float T0[9];
    {
        const float Tinv[9] = {1, 0, -mx, 0, 1, -my, 0, 0, 1};
        float g[9];
        for (int i = 0; i < 9; ++i)
            g[i] = guess_all[9 * (size_t)J.out + i];
        mat3_mul(Tinv, g, T0);
    }
    if (tid < SW_NS_MAX)
        s_cnt[tid] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += NT) {
        const float2 sp = src[i];
        atomicAdd(&s_cnt[strip_of(affine1(T0[3], T0[4], T0[5], sp.x, sp.y), ylo, inv_g, nst)], 1);
    }
    __syncthreads();
    if (tid == 0) {
        long long cum = 0;
        for (int g = 0; g <= SW_MG_MAX; ++g)
            s_goff[g] = 0;
        for (int st = 0; st < SW_NS_MAX; ++st) { // strip -> share: the share its middle query falls into
            const int c = st < nst ? s_cnt[st] : 0;
            int g = (int)(((2 * cum + c) * G) / (2 * (long long)max(n, 1)));
            g = min(max(g, 0), G - 1);
            s_grp[st] = g;
            s_goff[g + 1] += c;
            cum += c;
        }
        for (int g = 0; g < SW_MG_MAX; ++g) { // counts -> offsets
            s_goff[g + 1] += s_goff[g];
            s_run[g] = 0;
        }
    }
    __syncthreads();
    for (int base = 0; base < n; base += NT) {
        const int i = base + tid;
        const bool valid = i < n;
        float2 sp = make_float2(0, 0);
        int g = -1;
        if (valid) {
            sp = src[i];
            g = s_grp[strip_of(affine1(T0[3], T0[4], T0[5], sp.x, sp.y), ylo, inv_g, nst)];
        }
        unsigned long long mine = 0;
        for (int gg = 0; gg < G; ++gg) {
            const unsigned long long m = __ballot(g == gg);
            if (g == gg)
                mine = m;
            if (lane == 0)
                s_w[wave][gg] = __popcll(m);
        }
        __syncthreads();
        if (tid < G) { // this chunk's queries of share `tid`: where each wave's run starts
            int acc = s_run[tid];
            for (int w = 0; w < NT / 64; ++w) {
                const int t = s_w[w][tid];
                s_w[w][tid] = acc;
                acc += t;
            }
            s_run[tid] = acc;
        }
        __syncthreads();
        if (valid)
            gsrc_all[J.q_off + s_goff[g] + s_w[wave][g] + __popcll(mine & ((1ull << lane) - 1ull))] = sp;
        __syncthreads();
    }
    if (tid < J.ngrp) {
        const int g = min(tid, SW_MG_MAX - 1);
        jobs[j0 + tid].src_start = (int)(J.q_off + s_goff[g]);
        jobs[j0 + tid].n_src = (tid < G) ? s_goff[g + 1] - s_goff[g] : 0;
        jobs[j0 + tid].q_off = J.q_off + s_goff[g];
    }
}

// ---------------------------------------------------------------------------------------------
// host side: job tables, scratch, the prep / split / loop launches.  jobs4 = n_jobs x (src_start, n_src, tgt_start,
// n_tgt) in points.
// ---------------------------------------------------------------------------------------------
namespace {
struct SweepLaunchArgs {
    sfe_ctx *ctx;
    const sfe_icp_params *p;
    const SweepJob *d_jobs;
    const float2 *d_src;
    const float *d_guess9;
    const float2 *d_stgt;
    const int *d_perm;
    const float2 *d_snrm;
    const float *d_mean;
    const StripTab *d_tab;
    const int *d_grid;
    int4 *d_qst;
    int *d_qwl;
    float2 *d_qssrc;
    float *d_nn_d2;
    int *d_nn_pos;
    float *d_T9;
    int32_t *d_status, *d_iters;
    long long *d_prof;
    int *d_dbg;
    int sw_budget, sw_budget_a, sw_cache, sw_cache2;
    float sw_m, sw_kappa;
    unsigned long long *d_sync;
};

int pow2_floor(size_t v)
{
    size_t p = 1;
    while (2 * p <= v)
        p *= 2;
    return (int)p;
}

// bytes of the control block of one instantiation (the dynamic LDS behind it is 16-byte aligned)
template <int NT, bool PROF, bool REC>
constexpr size_t sweep_ctl_bytes()
{
    return (sizeof(SweepShared<NT, PROF, REC>) + 15) & ~(size_t)15;
}

// one launch of the loop kernel: n workgroups, job ids d_ids[0..n), `body` bytes of LDS behind the control block
template <int NT, int MINW, bool LDS_TGT, bool LDS_Q, bool PROF, bool REC, bool MULTI, bool WIN = false>
int sweep_launch_loop(const SweepLaunchArgs &a, int n, const int *d_ids, size_t body, int t_cap, int q_cap)
{
    sfe_ctx *ctx = a.ctx;
    auto kernel = icp_sweep_kernel<NT, MINW, LDS_TGT, LDS_Q, PROF, REC, MULTI, WIN>;
    const size_t smem = sweep_ctl_bytes<NT, PROF, REC>() + body;
    SFE_HIP(ctx, hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(kernel, dim3(n), dim3(NT), smem, ctx->stream, *a.p, a.d_jobs, d_ids, a.d_src, a.d_guess9, a.d_stgt,
                       a.d_perm, a.d_snrm, a.d_mean, a.d_tab, a.d_grid, a.d_qst, a.d_qwl, a.d_qssrc, a.d_nn_d2, a.d_nn_pos,
                       a.d_T9, a.d_status, a.d_iters, a.d_prof, a.d_dbg, a.sw_budget, a.sw_budget_a, a.sw_cache, t_cap, q_cap,
                       pow2_floor(body / 8), a.sw_m, a.sw_kappa, a.d_sync, a.sw_cache2);
    SFE_LAUNCH_CHECK(ctx);
    return 0;
}

int env_int(const char *name, int dflt)
{
    const char *v = getenv(name);
    return v ? atoi(v) : dflt;
}

template <int NT, int TCAP, int GM, bool GTAIL = false, int KMF = 0>
int sweep_launch_prep(sfe_ctx *ctx, hipStream_t ps, const sfe_icp_params *p, int n, const SweepPrep *d_preps, const int *d_pids,
                      const float2 *d_tgt, float2 *d_stgt, int *d_perm, float2 *d_snrm, float *d_mean,
                      unsigned long long *d_gkeys, StripTab *d_tab, int *d_grid)
{
    auto kernel = icp_sweep_prep_kernel<NT, TCAP, GM, GTAIL, KMF>;
    const size_t smem = ((sizeof(PrepShared<NT, TCAP>) + 15) & ~(size_t)15) + (GTAIL ? 0 : 4 * (size_t)GM);
    static_assert(!GTAIL || 8 * (size_t)(TCAP + SW_PAD + 4) >= 4 * (size_t)GM, "the key buffer holds a whole witness grid");
    SFE_HIP(ctx, hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(kernel, dim3(n), dim3(NT), smem, ps, *p, d_preps, d_pids, d_tgt, d_stgt, d_perm, d_snrm, d_mean, d_gkeys,
                       d_tab, d_grid);
    SFE_LAUNCH_CHECK(ctx);
    return 0;
}

} // namespace

int sfe_icp_sweep_launch(sfe_ctx *ctx, const sfe_icp_params *p, const float *d_src, const float *d_tgt,
                         const int32_t *jobs4, const float *d_guess9, int n_jobs, float *d_T9, int32_t *d_status,
                         int32_t *d_iters)
{
    std::vector<SweepPrep> preps;
    std::vector<SweepJob> jobs;
    jobs.reserve((size_t)n_jobs);
    // Job classes, one launch of the loop kernel each:
    //   tiny     clouds of a few hundred points: one wave per job, exhaustive search (sfe_icp_tiny.h)
    //   t0 / t1  small jobs: one-wave / four-wave workgroups, target AND per-query results in LDS, many jobs per CU
    //   q        1024-thread workgroups, target AND per-query results in LDS
    //   lds      ... target in LDS, results in HBM scratch
    //   glb      ... target in HBM scratch (beyond SW_TCAP points)
    //   multi    ... target in HBM scratch, the job shared by several workgroups (many-to-one batches on large clouds)
    std::vector<int> ids_tiny, ids_t0, ids_t1, ids_q, ids_lds, ids_glb, ids_multi, split_first;
    std::vector<int> pids[3]; // targets by prep tier
    std::vector<int> prep_tier; // ... of every target
    std::vector<char> prep_need; // a target only tiny point-to-point jobs use is not prepared (they take its mean themselves)
    std::vector<int> pids_nrm; // ... and those whose normals take a kernel of their own
    std::map<std::pair<int, int>, int> seen; // many guesses on one pair share one prep
    long long toff = 0, qoff = 0, koff = 0, goff = 0;
    // (knobs are read per call, not once per process: the tests switch them between calls)
    // tiers (A/B: SFE_SW_TIERS=0 sends every job to the 1024-thread kernels)
    const int tiers_on = env_int("SFE_SW_TIERS", 1);
    const int t0_src = env_int("SFE_SW_T0_SRC", 384), t1_src = env_int("SFE_SW_T1_SRC", 2048);
    // LDS share of a 1024-thread workgroup: two per CU when the batch has more jobs than CUs, else the whole CU
    const int force_wide = env_int("SFE_SW_WIDE", -1); // A/B: 1 = one 128-VGPR workgroup per CU
    const int force_share = env_int("SFE_SW_SHARE_KB", 0); // A/B: LDS per workgroup
    const bool no_ldsq = getenv("SFE_SW_NO_LDSQ") != nullptr; // A/B
    // A/B: 0 = never split a job; sfe_icp_set_tuning bit 4 = the same per context (a device shared with another
    // context / process cannot promise that all shares of a job are resident together: ADVICE r3)
    const int multi_on = (ctx->icp_variant & 16) ? 0 : env_int("SFE_SW_MULTI", 1);
    const int multi_min_src = env_int("SFE_SW_MULTI_MIN_SRC", 8192);
    const int multi_share_min = env_int("SFE_SW_MULTI_SHARE_MIN", 1024); // fewest queries worth a workgroup
    const int multi_force = env_int("SFE_SW_MULTI_G", 0);     // A/B: shares per job
    // ~96 points per strip on average (measured optimum 96-128 on 5000-point clouds: fewer queries need a
    // second strip, a little more to walk in each), one strip (= a plain x sweep) for small clouds
    const int strip_pts = std::max(1, env_int("SFE_SW_STRIP_PTS", 96));
    // clouds of a few hundred points: the exhaustive one-wave kernel (A/B: SFE_SW_TINY=0)
    const int tiny_on = env_int("SFE_SW_TINY", 1);
    const long long tiny_pairs = env_int("SFE_SW_TINY_PAIRS", (p->use_diff_checker || p->max_iter < SW_REC_MIN_ITER)
                                                                    ? SW_TINY_PAIRS_SHORT : SW_TINY_PAIRS_LONG);
    auto is_tiny = [&](int n_src, int n_tgt) {
        return tiny_on && n_src <= SW_TINY_MAX && n_tgt <= SW_TINY_MAX && (long long)n_src * n_tgt <= tiny_pairs;
    };

    // The small workgroups buy throughput (many jobs per CU), not latency: ONE scan match of 200 points is done sooner by
    // 256 threads (one slice of queries per pass) than by 64 (four slices, one after the other).  So the one-wave tier is
    // only used when the call brings enough such jobs to fill the device with them (the live node's single scan
    // match takes the four-wave kernel).
    const int t0_min_jobs = env_int("SFE_SW_T0_MIN_JOBS", 2 * ctx->n_cu), t1_min_jobs = env_int("SFE_SW_T1_MIN_JOBS", 1);
    auto fits0 = [&](int n_src, int n_tgt) { return tiers_on && n_src <= std::min(t0_src, 4096) && n_tgt <= SW_T0_TCAP; };
    auto fits1 = [&](int n_src, int n_tgt) { return tiers_on && n_src <= std::min(t1_src, 8192) && n_tgt <= SW_T1_TCAP; };
    int n_fit0 = 0, n_fit1 = 0;
    for (int j = 0; j < n_jobs; ++j) {
        const int32_t *q = jobs4 + 4 * (size_t)j;
        if (fits0(q[1], q[3]))
            ++n_fit0;
        else if (fits1(q[1], q[3]))
            ++n_fit1;
    }
    const bool use_t0 = n_fit0 >= t0_min_jobs;
    const bool use_t1 = n_fit1 + (use_t0 ? 0 : n_fit0) >= std::max(1, t1_min_jobs);
    // ... and with only a few small jobs in the call the four-wave kernel takes those of at most one slice of queries
    // (measured, one scan match alone: 200 points 261 us with 256 threads, 282 with 1024, 464 with 64; 1000 points 508 us
    // with 256 threads, 341 with 1024)
    const bool few_small = n_fit0 + n_fit1 < 2 * ctx->n_cu;
    const int t1_few_src = env_int("SFE_SW_T1_FEW_SRC", 320);
    // first pass: sizes -> how many big jobs there are (decides `wide` and whether big jobs are split)
    int n_big = 0, n_t2 = 0;
    auto tier_of = [&](int n_src, int n_tgt) {
        if (use_t0 && fits0(n_src, n_tgt))
            return 0;
        if (use_t1 && fits1(n_src, n_tgt) && (!few_small || n_src <= t1_few_src))
            return 1;
        return 2;
    };
    for (int j = 0; j < n_jobs; ++j) {
        const int32_t *q = jobs4 + 4 * (size_t)j;
        if (!is_tiny(q[1], q[3]) && tier_of(q[1], q[3]) == 2) {
            ++n_t2;
            if (q[3] > SW_TCAP && q[1] >= multi_min_src)
                ++n_big;
        }
    }
    const bool wide = force_wide >= 0 ? force_wide != 0 : n_t2 <= ctx->n_cu;
    const size_t lds_share = (force_share > 0 ? force_share : (wide ? 160 : 80)) * (size_t)1024;
    // Shares per big job: all workgroups of the launch must be resident at once (one 128-VGPR workgroup per CU) -- only
    // when the 1024-thread jobs of this call are the big ones alone and G x n_big fits the CUs
    int mg = 1;
    if (multi_on && n_big > 0 && n_big == n_t2 && wide) {
        mg = std::min(SW_MG_MAX, ctx->n_cu / n_big);
        if (multi_force > 0)
            mg = std::min(std::min(multi_force, SW_MG_MAX), std::max(1, ctx->n_cu / n_big));
    }
    constexpr size_t ctl_q = sweep_ctl_bytes<ICP_THREADS, false, true>(); // (the largest control block of the LDS_Q builds)
    int q_tmax = 0, q_smax = 0, t0_tmax = 0, t0_smax = 0, t1_tmax = 0, t1_smax = 0, tiny_tmax = 0, tiny_smax = 0;
    int n_sync = 0;
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
            const int n_strips = std::max(1, std::min(SW_NS_MAX, (int)q[3] / strip_pts));
            // (the target's preparation follows the same rule: one wave only when the call fills the device with such jobs)
            const int pt = (use_t0 && q[3] <= SW_T0_TCAP) ? 0 : ((use_t1 && q[3] <= SW_T1_TCAP) ? 1 : 2);
            prep_tier.push_back(pt);
            prep_need.push_back(0);
            // (targets sorted in HBM scratch get their normals from icp_sweep_normals_kernel: pad_ = 1)
            const int nrm_later = (q[3] > SW_TCAP && p->minimizer == 1 && env_int("SFE_SW_NORMALS_SPLIT", 1)) ? 1 : 0;
            if (nrm_later)
                pids_nrm.push_back((int)preps.size());
            preps.push_back({q[2], q[3], n_strips, nrm_later, toff, koff, goff});
            toff += q[3] + SW_PAD;
            koff += n2;
            goff += pt == 0 ? SW_T0_GRID : (pt == 1 ? SW_T1_GRID : SW_GRID_MAX);
        }
        const SweepPrep &pr = preps[(size_t)it->second];
        const bool tiny = is_tiny(q[1], q[3]);
        const int tier = tiny ? -1 : tier_of(q[1], q[3]);
        if (!tiny || p->minimizer == 1)
            prep_need[(size_t)it->second] = 1;
        int shares = 1;
        if (tier == 2 && mg > 1 && q[3] > SW_TCAP && q[1] >= multi_min_src)
            shares = std::max(1, std::min(mg, (int)q[1] / std::max(1, multi_share_min)));
        const int rec0 = (int)jobs.size();
        for (int g = 0; g < shares; ++g) // (the shares of a split job: the split kernel fills in their clouds)
            jobs.push_back({q[0], q[1], q[3], it->second, pr.off, qoff, j, g, shares, shares > 1 ? n_sync : 0});
        qoff += q[1];
        if (shares > 1) {
            split_first.push_back(rec0);
            for (int g = 0; g < shares; ++g)
                ids_multi.push_back(rec0 + g);
            ++n_sync;
        } else if (tiny) {
            ids_tiny.push_back(rec0);
            tiny_tmax = std::max(tiny_tmax, (int)q[3]);
            tiny_smax = std::max(tiny_smax, (int)q[1]);
        } else if (tier == 0) {
            ids_t0.push_back(rec0);
            t0_tmax = std::max(t0_tmax, (int)q[3]);
            t0_smax = std::max(t0_smax, (int)q[1]);
        } else if (tier == 1) {
            ids_t1.push_back(rec0);
            t1_tmax = std::max(t1_tmax, (int)q[3]);
            t1_smax = std::max(t1_smax, (int)q[1]);
        } else {
            const bool fits_q = !no_ldsq && q[3] <= SW_TCAP &&
                                ctl_q + 8 * (size_t)(q[3] + SW_PAD) + 6 * (size_t)q[1] + 16 <= lds_share;
            if (fits_q) {
                ids_q.push_back(rec0);
                q_tmax = std::max(q_tmax, (int)q[3]);
                q_smax = std::max(q_smax, (int)q[1]);
            } else {
                (q[3] <= SW_TCAP ? ids_lds : ids_glb).push_back(rec0);
            }
        }
    }
    for (size_t pid = 0; pid < preps.size(); ++pid)
        if (prep_need[pid])
            pids[prep_tier[pid]].push_back((int)pid);
    // the LDS_Q launch is sized by the largest target and the largest source among its jobs
    int t_cap = q_tmax + SW_PAD, q_cap = (q_smax + 3) & ~3;
    if (!ids_q.empty() && ctl_q + 8 * (size_t)t_cap + 6 * (size_t)q_cap > lds_share) {
        ids_lds.insert(ids_lds.end(), ids_q.begin(), ids_q.end()); // odd mix of shapes: keep the results in HBM
        std::sort(ids_lds.begin(), ids_lds.end());
        ids_q.clear();
    }
    const int n_prep = (int)preps.size(), n_rec = (int)jobs.size();
    const int n_tiny = (int)ids_tiny.size();
    const int n_t0 = (int)ids_t0.size(), n_t1 = (int)ids_t1.size(), n_q = (int)ids_q.size(), n_lds = (int)ids_lds.size(),
              n_glb = (int)ids_glb.size(), n_multi = (int)ids_multi.size(), n_split = (int)split_first.size();
    // the tables travel as ONE block: [preps | job records | job ids by class | share-0 records of the split jobs |
    // prep ids by tier]
    const size_t o_jobs = (sizeof(SweepPrep) * (size_t)n_prep + 15) & ~(size_t)15;
    const size_t o_ids = (o_jobs + sizeof(SweepJob) * (size_t)n_rec + 15) & ~(size_t)15;
    const size_t o_split = o_ids + sizeof(int) * (size_t)n_rec;
    const size_t o_pids = o_split + sizeof(int) * (size_t)n_split;
    const size_t o_pnrm = o_pids + sizeof(int) * (size_t)n_prep;
    const size_t tab_bytes = o_pnrm + sizeof(int) * pids_nrm.size();
    char *d_tables = (char *)sfe_scratch(ctx, 12, tab_bytes);
    float2 *d_stgt = (float2 *)sfe_scratch(ctx, 14, sizeof(float2) * (size_t)toff);
    int *d_perm = (int *)sfe_scratch(ctx, 15, sizeof(int) * (size_t)toff);
    float2 *d_snrm = p->minimizer == 1 ? (float2 *)sfe_scratch(ctx, 16, sizeof(float2) * (size_t)toff) : nullptr;
    float *d_mean = (float *)sfe_scratch(ctx, 17, sizeof(float) * 2 * (size_t)n_prep);
    unsigned long long *d_gkeys = (unsigned long long *)sfe_scratch(ctx, 24, sizeof(unsigned long long) * (size_t)std::max(koff, 1LL));
    int4 *d_qst = (int4 *)sfe_scratch(ctx, 19, sizeof(int4) * (size_t)qoff);
    int *d_qwl = (int *)sfe_scratch(ctx, 21, sizeof(int) * 7 * (size_t)qoff);
    float2 *d_qssrc = (float2 *)sfe_scratch(ctx, 30, sizeof(float2) * (size_t)qoff);
    StripTab *d_tab = (StripTab *)sfe_scratch(ctx, 23, sizeof(StripTab) * (size_t)n_prep);
    int *d_grid = (int *)sfe_scratch(ctx, 38, sizeof(int) * (size_t)goff);
    float *d_nn_d2 = (float *)sfe_scratch(ctx, 5, sizeof(float) * (size_t)qoff);
    int *d_nn_pos = (int *)sfe_scratch(ctx, 6, sizeof(int) * (size_t)qoff);
    if (!d_tables || !d_stgt || !d_perm || (p->minimizer == 1 && !d_snrm) || !d_mean || !d_gkeys || !d_qst || !d_qwl || !d_qssrc || !d_tab || !d_grid ||
        !d_nn_d2 || !d_nn_pos)
        return SFE_ERR_HIP;
    // split jobs: the gathered source clouds of the shares and the sync areas
    float2 *d_gsrc = nullptr;
    unsigned long long *d_sync = nullptr;
    const size_t sync_bytes = sizeof(unsigned long long) * 2 * SW_MG_MAX * SW_MG_WORDS * (size_t)n_sync;
    if (n_split) {
        d_gsrc = (float2 *)sfe_scratch(ctx, 46, sizeof(float2) * (size_t)qoff);
        d_sync = (unsigned long long *)sfe_scratch(ctx, 45, sync_bytes);
        if (!d_gsrc || !d_sync)
            return SFE_ERR_HIP;
    }
    SweepPrep *d_preps = (SweepPrep *)d_tables;
    SweepJob *d_jobs = (SweepJob *)(d_tables + o_jobs);
    int *d_ids = (int *)(d_tables + o_ids);
    int *d_split = (int *)(d_tables + o_split);
    int *d_pids = (int *)(d_tables + o_pids);
    // Tuning bit 3: the caller vouches that the clouds and guesses of this batch are final (nothing enqueued on
    // ctx->stream still writes them).  The job tables and the prep kernel then go to the side stream and run next to
    // whatever precedes this call on ctx->stream (a batch pipeline enqueues the front end of the same step there:
    // latency-bound kernels that leave most of a CU idle, like the prep kernel); the loop kernel waits for them.
    const bool side = (ctx->icp_variant & 8) != 0;
    hipStream_t ps = side ? ctx->stream2 : ctx->stream;
    if (side && ctx->icp_loop_pending) // the previous batch's loop kernel still reads the scratch the prep rewrites
        SFE_HIP(ctx, hipStreamWaitEvent(ps, ctx->ev_loop, 0));
    { // pinned staging (no stream synchronisation: this entry point only enqueues)
        char *h = (char *)sfe_pinned_begin(ctx, tab_bytes);
        if (!h)
            return SFE_ERR_HIP;
        memcpy(h, preps.data(), sizeof(SweepPrep) * (size_t)n_prep);
        memcpy(h + o_jobs, jobs.data(), sizeof(SweepJob) * (size_t)n_rec);
        int *hi = (int *)(h + o_ids);
        for (const std::vector<int> *v : {&ids_tiny, &ids_t0, &ids_t1, &ids_q, &ids_lds, &ids_glb, &ids_multi}) {
            if (!v->empty())
                memcpy(hi, v->data(), sizeof(int) * v->size());
            hi += v->size();
        }
        if (n_split)
            memcpy(h + o_split, split_first.data(), sizeof(int) * (size_t)n_split);
        int *hp = (int *)(h + o_pids);
        for (int t = 0; t < 3; ++t) {
            if (!pids[t].empty())
                memcpy(hp, pids[t].data(), sizeof(int) * pids[t].size());
            hp += pids[t].size();
        }
        if (!pids_nrm.empty())
            memcpy(h + o_pnrm, pids_nrm.data(), sizeof(int) * pids_nrm.size());
        SFE_HIP(ctx, hipMemcpyAsync(d_tables, h, tab_bytes, hipMemcpyHostToDevice, ps));
        if (int rc = sfe_pinned_end(ctx, ps))
            return rc;
    }
    {
        const int np0 = (int)pids[0].size(), np1 = (int)pids[1].size(), np2 = (int)pids[2].size();
        if (np0)
            if (int rc = sweep_launch_prep<SW_T0_NT, SW_T0_TCAP, SW_T0_GRID>(ctx, ps, p, np0, d_preps, d_pids, (const float2 *)d_tgt, d_stgt,
                                                                          d_perm, d_snrm, d_mean, d_gkeys, d_tab, d_grid))
                return rc;
        if (np1)
            if (int rc = sweep_launch_prep<SW_T1_NT, SW_T1_TCAP, SW_T1_GRID>(ctx, ps, p, np1, d_preps, d_pids + np0, (const float2 *)d_tgt,
                                                                          d_stgt, d_perm, d_snrm, d_mean, d_gkeys, d_tab, d_grid))
                return rc;
        if (np2) {
            const bool gtail = env_int("SFE_SW_PREP_GTAIL", 1) != 0; // (0: the witness grid in LDS of its own, one workgroup per CU: A/B)
            // (the list capacity the kernel is built for: the smallest of 8 / 10 / 12 / 16 that holds normals_knn)
            const int km = p->minimizer != 1 ? 8 : p->normals_knn <= 8 ? 8 : p->normals_knn <= 10 ? 10 : p->normals_knn <= 12 ? 12 : ICP_KMAX;
            auto go = [&](auto gt, auto kmf) {
                return sweep_launch_prep<ICP_THREADS, SW_TCAP, SW_GRID_MAX, decltype(gt)::value, decltype(kmf)::value>(
                    ctx, ps, p, np2, d_preps, d_pids + np0 + np1, (const float2 *)d_tgt, d_stgt, d_perm, d_snrm, d_mean, d_gkeys, d_tab,
                    d_grid);
            };
            using std::integral_constant;
            // (lists of 12 / 16 neighbours do not fit 64 registers: 184 / 541 spills; those keep one workgroup per CU)
            if (int rc = !gtail || km > 10 ? go(std::false_type{}, integral_constant<int, 0>{})
                         : km == 8          ? go(std::true_type{}, integral_constant<int, 8>{})
                                            : go(std::true_type{}, integral_constant<int, 10>{}))
                return rc;
        }
    }
    if (!pids_nrm.empty()) { // behind the prep (sorted cloud, strip table), 1024 points per workgroup and pass
        int tmax = 0;
        for (int pid : pids_nrm)
            tmax = std::max(tmax, preps[(size_t)pid].n_tgt);
        const int per = std::max(1, std::min(32, std::min((tmax + 2 * ICP_THREADS - 1) / (2 * ICP_THREADS),
                                                          std::max(1, 2 * ctx->n_cu / (int)pids_nrm.size()))));
        hipLaunchKernelGGL(icp_sweep_normals_kernel<ICP_THREADS>, dim3((unsigned)pids_nrm.size(), (unsigned)per), dim3(ICP_THREADS), 0, ps,
                           *p, d_preps, (const int *)(d_tables + o_pnrm), (const float2 *)d_stgt, (const int *)d_perm, d_snrm,
                           (const StripTab *)d_tab);
        SFE_LAUNCH_CHECK(ctx);
    }
    if (n_split) { // behind the prep (it needs the strip tables), in front of the loop
        SFE_HIP(ctx, hipMemsetAsync(d_sync, 0, sync_bytes, ps));
        hipLaunchKernelGGL(icp_split_kernel<ICP_THREADS>, dim3(n_split), dim3(ICP_THREADS), 0, ps, d_jobs, d_split,
                           (const float2 *)d_src, d_guess9, d_mean, d_tab, d_gsrc);
        SFE_LAUNCH_CHECK(ctx);
    }
    if (side) {
        SFE_HIP(ctx, hipEventRecord(ctx->ev_prep, ps));
        SFE_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_prep, 0));
    }
    static const bool debug = getenv("SFE_ICP_DEBUG") != nullptr;
    // bit 0: witness / clearance cache (0: A/B without it); bits 8..15: walk trips per second-pass round
    // bit 1: the cap of a repeated round comes from the queries' upper bounds (0: grows 4x)
    // bit 2: grid witnesses in the first iteration (0: A/B without them)
    const int sw_cache = (env_int("SFE_SW_CACHE", 1) & 1) | (env_int("SFE_SW_GRID", 1) ? 4 : 0) |
                         (env_int("SFE_SW_GRID_SKIP", 1) ? 8 : 0) | // bit 3: witnessed queries skip round 0
                         (env_int("SFE_SW_REC", 1) ? 16 : 0) |      // bit 4: clearance records
                         (env_int("SFE_SW_TRIAGE", 1) ? 32 : 0) |   // bit 5: ... with triage passes
                         (env_int("SFE_SW_GRID_DEFER", 1) ? 64 : 0) | // bit 6: first iteration: queries without a witness wait for round 1
                         (env_int("SFE_SW_UNBOUNDED_COOP", 1) ? 128 : 0) | // bit 7: later passes: unbounded queries -> cooperative tier
                         (env_int("SFE_SW_LEAN_TRIAGE", 1) ? (1 << 25) : 0) |  // bit 25: the triage pass as a loop of its own
                         (env_int("SFE_SW_JUMP", 1) ? 2 : 0) |
                         // bits 16..23: margin (percent) of the next iteration's cap over this iteration's limit
                         (std::max(0, std::min(255, env_int("SFE_SW_MARGIN", SW_CAP_MARGIN))) << 16) |
                         (std::max(1, std::min(255, env_int("SFE_SW_RTRIPS", SW_ROUND_TRIPS))) << 8);
    int *d_dbg = nullptr;
    if (debug) {
        d_dbg = (int *)sfe_scratch(ctx, 22, sizeof(int) * 8);
        if (!d_dbg)
            return SFE_ERR_HIP;
        SFE_HIP(ctx, hipMemsetAsync(d_dbg, 0, sizeof(int) * 8, ctx->stream));
    }
    long long *d_prof = ctx->icp_prof ? (long long *)sfe_scratch(ctx, 20, sizeof(long long) * SFE_ICP_PROF_N) : nullptr;
    if (ctx->icp_prof && !d_prof)
        return SFE_ERR_HIP;
    if (d_prof)
        SFE_HIP(ctx, hipMemsetAsync(d_prof, 0, sizeof(long long) * SFE_ICP_PROF_N, ctx->stream));
    SweepLaunchArgs a;
    a.ctx = ctx;
    a.p = p;
    a.d_jobs = d_jobs;
    a.d_src = (const float2 *)d_src;
    a.d_guess9 = d_guess9;
    a.d_stgt = d_stgt;
    a.d_perm = d_perm;
    a.d_snrm = d_snrm;
    a.d_mean = d_mean;
    a.d_tab = d_tab;
    a.d_grid = d_grid;
    a.d_qst = d_qst;
    a.d_qwl = d_qwl;
    a.d_qssrc = d_qssrc;
    a.d_nn_d2 = d_nn_d2;
    a.d_nn_pos = d_nn_pos;
    a.d_T9 = d_T9;
    a.d_status = d_status;
    a.d_iters = d_iters;
    a.d_prof = d_prof;
    a.d_dbg = d_dbg;
    a.sw_budget = env_int("SFE_SW_BUDGET", SW_BUDGET);
    a.sw_budget_a = env_int("SFE_SW_BUDGET_A", SW_BUDGET_A);
    a.sw_cache = sw_cache;
    // union scan of the first iterations: bits 0..7 = iterations that use it, bits 8..23 = most points of a union window.
    // Measured (1024 jobs of 5000 x 5000, search cycles of workgroup 0 per iteration, walks -> union): iteration 0
    // 997 k -> 828 k, iteration 1 486 k -> 659 k, iteration 2 324 k -> 585 k: only the first iteration's windows (bounds
    // from grid witnesses, ~0.9 m) are wide enough for the union of 64 neighbours' windows (~400 points) to beat the
    // private walks; from the second iteration on a query's old neighbour bounds it to a few dozen candidates.  Whole
    // launch: p2plane30 8.78-8.87 -> 8.64-8.76 ms, shipped chain 4.07 -> 3.93-4.00 ms per 1024 jobs.
    a.sw_cache2 = std::max(0, std::min(255, env_int("SFE_SW_UNION_ITERS", 1))) |
                  (std::max(0, std::min(65535, env_int("SFE_SW_UNION_MAX", 768))) << 8);
    // margin of the clearance records: a search looks this fraction further (in radius) than it has to ...
    a.sw_m = 0.01f * (float)std::max(1, std::min(100, env_int("SFE_SW_RECM", SW_REC_MARGIN)));
    // ... plus sw_kappa x the largest movement of the last step (the steps shrink geometrically once ICP converges: a
    // few times the last one covers all that are still to come)
    a.sw_kappa = getenv("SFE_SW_RECK") ? (float)atof(getenv("SFE_SW_RECK")) : SW_REC_KAPPA;
    a.d_sync = d_sync;
    // The build with clearance records carries more per-lane state (the 64-VGPR budget makes every register count:
    // the same chain runs ~10 % slower in it until the records start to hit), so it only takes chains that are set to
    // run many iterations: a fixed count (no differential checker) of at least SW_REC_MIN_ITER -- and only the builds
    // with LDS-resident targets: for the one-workgroup-per-CU builds with the target in HBM (30 guesses x one
    // 20 000 x 20 000 pair) the records cost more than they save (24.6 -> 30.9 ms, tools/hires_times.py).
    const bool rec_build = (sw_cache & 16) != 0 && p->max_iter >= SW_REC_MIN_ITER && !p->use_diff_checker;
    // A/B: VGPR budget of the LDS_Q build.  A workgroup is 1024 threads = 4 waves per SIMD, so two workgroups per CU
    // only fit at <= 64 VGPRs: measured (4096 jobs of 5000 x 5000, p2plane30) 64 VGPRs 35.4 ms, 72 VGPRs 51.4 ms,
    // 128 VGPRs 48.6 ms (profiles/r02_icp_vgpr_budget.txt) -- one resident job per CU costs more than the 86 spilled
    // VGPRs of the 64-VGPR build
    int rc = 0;
    const int *ids = d_ids;
    if (n_tiny) { // one wave per job, everything in LDS (sfe_icp_tiny.h)
        const int tc = ((tiny_tmax + SW_TINY_CH - 1) / SW_TINY_CH + 1) * SW_TINY_CH, qc = (tiny_smax + 3) & ~3; // (whole chunks + one of padding)
        const size_t smem = ((sizeof(TinyShared) + 15) & ~(size_t)15) + sizeof(float2) * (size_t)tc * (p->minimizer == 1 ? 2 : 1) +
                            6 * (size_t)qc;
        hipLaunchKernelGGL(icp_tiny_kernel, dim3(n_tiny), dim3(SW_TINY_NT), smem, ctx->stream, *p, d_jobs, ids, d_preps,
                           (const float2 *)d_src, (const float2 *)d_tgt, d_guess9, (const int *)d_perm, (const float2 *)d_snrm,
                           (const float *)d_mean, (const StripTab *)d_tab, d_T9, d_status, d_iters, tc, qc);
        SFE_LAUNCH_CHECK(ctx);
        ids += n_tiny;
    }
    if (n_t0) { // one wave per job (no workgroup barrier costs anything), 128 VGPRs, up to 16 jobs per CU
        const int tc = t0_tmax + SW_PAD, qc = (t0_smax + 3) & ~3;
        const size_t body = 8 * (size_t)tc + 6 * (size_t)qc;
        rc = rec_build ? sweep_launch_loop<SW_T0_NT, 4, true, true, false, true, false>(a, n_t0, ids, body, tc, qc)
                       : sweep_launch_loop<SW_T0_NT, 4, true, true, false, false, false>(a, n_t0, ids, body, tc, qc);
        if (rc)
            return rc;
        ids += n_t0;
    }
    if (n_t1) { // four waves per job, 128 VGPRs, up to 4 jobs per CU
        const int tc = t1_tmax + SW_PAD, qc = (t1_smax + 3) & ~3;
        const size_t body = 8 * (size_t)tc + 6 * (size_t)qc;
        const int t1_minw = env_int("SFE_SW_T1_MINW", 4); // A/B: 8 = 64 VGPRs, up to 8 jobs per CU
        if (d_prof && !rec_build)
            rc = sweep_launch_loop<SW_T1_NT, 4, true, true, true, false, false>(a, n_t1, ids, body, tc, qc);
        else if (t1_minw == 8 && !rec_build)
            rc = sweep_launch_loop<SW_T1_NT, 8, true, true, false, false, false>(a, n_t1, ids, body, tc, qc);
        else
        rc = rec_build ? sweep_launch_loop<SW_T1_NT, 4, true, true, false, true, false>(a, n_t1, ids, body, tc, qc)
                       : sweep_launch_loop<SW_T1_NT, 4, true, true, false, false, false>(a, n_t1, ids, body, tc, qc);
        if (rc)
            return rc;
        ids += n_t1;
    }
    // 1024-thread jobs: up to one job per CU the 128-VGPR build wins (no spills, measured +8 %), beyond that two
    // 64-VGPR workgroups per CU overlap each other's serial phases (measured +14 % at 512 jobs)
    if (n_q) {
        const size_t body = 8 * (size_t)t_cap + 6 * (size_t)q_cap;
        if (wide)
            rc = sweep_launch_loop<ICP_THREADS, 4, true, true, false, false, false>(a, n_q, ids, body, t_cap, q_cap);
        else if (d_prof && rec_build)
            rc = sweep_launch_loop<ICP_THREADS, 8, true, true, true, true, false>(a, n_q, ids, body, t_cap, q_cap);
        else if (d_prof)
            rc = sweep_launch_loop<ICP_THREADS, 8, true, true, true, false, false>(a, n_q, ids, body, t_cap, q_cap);
        else if (rec_build)
            rc = sweep_launch_loop<ICP_THREADS, 8, true, true, false, true, false>(a, n_q, ids, body, t_cap, q_cap);
        else
            rc = sweep_launch_loop<ICP_THREADS, 8, true, true, false, false, false>(a, n_q, ids, body, t_cap, q_cap);
        if (rc)
            return rc;
        ids += n_q;
    }
    if (n_lds) {
        const size_t body = sizeof(float2) * (SW_TCAP + SW_PAD);
        if (wide)
            rc = sweep_launch_loop<ICP_THREADS, 4, true, false, false, false, false>(a, n_lds, ids, body, SW_TCAP + SW_PAD, 0);
        else if (d_prof)
            rc = sweep_launch_loop<ICP_THREADS, 8, true, false, true, false, false>(a, n_lds, ids, body, SW_TCAP + SW_PAD, 0);
        else if (rec_build)
            rc = sweep_launch_loop<ICP_THREADS, 8, true, false, false, true, false>(a, n_lds, ids, body, SW_TCAP + SW_PAD, 0);
        else
            rc = sweep_launch_loop<ICP_THREADS, 8, true, false, false, false, false>(a, n_lds, ids, body, SW_TCAP + SW_PAD, 0);
        if (rc)
            return rc;
        ids += n_lds;
    }
    // Targets beyond SW_TCAP points stay in HBM scratch and are read through L2.  SFE_SW_WIN=1 (A/B) holds as much of the
    // sorted target as fits next to the control block in LDS instead (TgtWin: the strips around the workgroup's queries;
    // 19 700 of a 20 000-point cloud's 20 068 positions), one workgroup per CU.  Measured, 30 guesses x one 20 000 x
    // 20 000 pair over 8 shares: 7.97 ms with the window, 7.82 without; 16 batches unsplit: 62.4 / 60.1 ms -- the target
    // is L2-resident and sixteen waves per workgroup hide its latency; every read pays the window test.  Off by default.
    const int win_on = env_int("SFE_SW_WIN", 0);
    constexpr size_t ctl_g = sweep_ctl_bytes<ICP_THREADS, false, false>();
    int glb_tmax = 0;
    for (int j = 0; j < n_jobs; ++j)
        if (jobs4[4 * (size_t)j + 3] > SW_TCAP)
            glb_tmax = std::max(glb_tmax, (int)jobs4[4 * (size_t)j + 3]);
    const int win_cap = std::min(glb_tmax + SW_PAD, (int)((160 * 1024 - ctl_g - 64) / 8));
    const size_t body_win = 8 * (size_t)std::max(win_cap, SW_TCAP); // (the query sort uses the same bytes first)
    if (n_glb) {
        const size_t body = sizeof(unsigned long long) * SW_TCAP; // without a window the LDS behind the control block only serves the query sort
        if (win_on)
            rc = sweep_launch_loop<ICP_THREADS, 4, false, false, false, false, false, true>(a, n_glb, ids, body_win, win_cap, 0);
        else
            rc = wide ? sweep_launch_loop<ICP_THREADS, 4, false, false, false, false, false>(a, n_glb, ids, body, SW_TCAP, 0)
                      : sweep_launch_loop<ICP_THREADS, 8, false, false, false, false, false>(a, n_glb, ids, body, SW_TCAP, 0);
        if (rc)
            return rc;
        ids += n_glb;
    }
    if (n_multi) { // every share on a CU of its own, all of them resident (mg x n_big <= CUs); their clouds were gathered
        const size_t body = sizeof(unsigned long long) * SW_TCAP;
        SweepLaunchArgs am = a;
        am.d_src = d_gsrc;
        rc = win_on ? sweep_launch_loop<ICP_THREADS, 4, false, false, false, false, true, true>(am, n_multi, ids, body_win, win_cap, 0)
                    : sweep_launch_loop<ICP_THREADS, 4, false, false, false, false, true>(am, n_multi, ids, body, SW_TCAP, 0);
        if (rc)
            return rc;
        ids += n_multi;
    }
    if (side) {
        SFE_HIP(ctx, hipEventRecord(ctx->ev_loop, ctx->stream));
        ctx->icp_loop_pending = true;
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
        SFE_HIP(ctx, hipMemcpyAsync(ctx->icp_prof_host, d_prof, sizeof(long long) * SFE_ICP_PROF_N, hipMemcpyDeviceToHost,
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
        for (int i = 0; i < SFE_ICP_PROF_N; ++i)
            cycles16[i] = ctx->icp_prof_host[i];
    ctx->icp_prof = enable;
    return 0;
}
