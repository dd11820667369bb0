#pragma once
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
    unsigned sel_below, sel_n; // predicted quantile: exact values below the window / inside it (their bit patterns: hist[0 .. sel_n))
    unsigned n_none, n_exact; // census of the iteration, tallied where a query is settled
    int grid_skips;           // first iteration: queries that took a grid witness instead of searching in round 0
    unsigned n_rechit[2];     // queries settled by their clearance record, this iteration / the one before
    int long_n, long_next, mid_n, wl_n[2];
    int flag_iterate, flag_status;
    int chk_n[3]; // the transformation checkers' counters (IcpCheck: nhist, counter, iters), thread 0's
    float Ti[9];
    float thist[REC ? ICP_MAX_HIST : 1][6]; // T_iter of every iteration so far (rows 0 and 1): the movement bounds below
    float mva[REC ? ICP_MAX_HIST : 1], mvt[REC ? ICP_MAX_HIST : 1]; // a query x = T0 * src has moved by at most
                                                // mva[k] |x| + mvt[k] between iteration k and the current one
    unsigned rmax_bits;           // largest |T0 * src| (float bits; >= 0 so the bit patterns order like the values)
    float hist_c[ICP_MAX_HIST], hist_s[ICP_MAX_HIST], hist_x[ICP_MAX_HIST], hist_y[ICP_MAX_HIST];
    long long prof_t, prof[PROF ? 16 : 1], prof_it[PROF ? 64 : 1], prof_b0;
    unsigned xr[16]; // split jobs: the scalars of an exchange between the workgroups of a job
    int xabort;      // ... and its time-out flag
    StripTab tab;
};

// (index of a profile counter: the builds without PROF carry a one-entry array and never execute these statements)
#define SW_PI(k) (PROF ? (k) : 0)
#define SW_PROF(k)                                                                               \
    do {                                                                                         \
        if (PROF && threadIdx.x == 0) {                                               \
            const long long t_ = clock64();                                                      \
            S.prof[SW_PI(k)] += t_ - S.prof_t;                                                          \
            S.prof_t = t_;                                                                       \
        }                                                                                        \
    } while (0)

// PROF builds compiled with -DSW_STAMPS: clock stamps of ONE steady-state iteration (SW_STAMP_ITER) of workgroup 0 -> prof[85 ..]:
// where an iteration that the clearance records settle spends its cycles (tools/stage_times.py prints the differences).  Off by
// default: the stamps change the profile build's register allocation (its sums phase reads 1.26 M cycles per job with them,
// 0.76 M without: profiles/r06_icp_stamps_perturb_ab.txt), so the per-phase TOTALS are taken without them.
#define SW_STAMP_ITER 25
#ifndef SW_STAMPS
#define SW_STAMP(k) do { } while (0)
#else
#define SW_STAMP(k)                                                                              \
    do {                                                                                         \
        if (PROF && threadIdx.x == 0 && blockIdx.x == 0 && it == SW_STAMP_ITER)                  \
            prof[85 + (k)] = clock64();                                                          \
    } while (0)
#endif

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


// sort key: strip (8 bits) | order key of x (32 bits) | original index (24 bits)
#define SW_KEY(s, xk, i) (((unsigned long long)(unsigned)(s) << 56) | ((unsigned long long)(xk) << 24) | (unsigned long long)(i))
#define SW_KEY_STRIP(k) ((int)((k) >> 56))
#define SW_KEY_X(k) ((unsigned)(((k) >> 24) & 0xFFFFFFFFull))
#define SW_KEY_ID(k) ((int)((k) & 0xFFFFFFull))

// ---------------------------------------------------------------------------------------------
// The three translation units of the strip-sweep ICP (round 5: one 3 200-line file before):
//   sfe_icp_sweep_prep.hip   target preparation kernels            + their launch functions
//   sfe_icp_sweep_loop.hip   loop kernel, one-wave kernel, split   + their launch functions
//   sfe_icp_sweep.hip        the host side: job classes, scratch, tables, the order of the launches
// The launch functions are templates over the kernels' build parameters, defined and explicitly instantiated where the
// kernels live; the host side only sees these declarations.
// ---------------------------------------------------------------------------------------------
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

// bytes of the control block of one instantiation (the dynamic LDS behind it is 16-byte aligned)
template <int NT, bool PROF, bool REC>
constexpr size_t sweep_ctl_bytes()
{
    return (sizeof(SweepShared<NT, PROF, REC>) + 15) & ~(size_t)15;
}

// one launch of the loop kernel: n workgroups, job ids d_ids[0..n), `body` bytes of LDS behind the control block
template <int NT, int MINW, bool LDS_TGT, bool LDS_Q, bool PROF, bool REC, bool MULTI>
int sweep_launch_loop(const SweepLaunchArgs &a, int n, const int *d_ids, size_t body, int t_cap, int q_cap);

// one launch of the prep kernel: n targets, ids d_pids[0..n) into d_preps, on stream ps
template <int NT, int TCAP, int GM, bool GTAIL = false, int KMF = 0>
int sweep_launch_prep(sfe_ctx *ctx, hipStream_t ps, const sfe_icp_params *p, int n, const SweepPrep *d_preps, const int *d_pids,
                      const float2 *d_tgt, float2 *d_stgt, int *d_perm, float2 *d_snrm, float *d_mean,
                      unsigned long long *d_gkeys, StripTab *d_tab, int *d_grid);

// normals of targets beyond the prep kernel's LDS (n targets x `per` workgroups each)
int sweep_launch_normals(sfe_ctx *ctx, hipStream_t ps, const sfe_icp_params *p, int n, int per, const SweepPrep *d_preps,
                         const int *d_pids, const float2 *d_stgt, const int *d_perm, float2 *d_snrm, const StripTab *d_tab);

// jobs shared by several workgroups: deal their queries to the shares
int sweep_launch_split(sfe_ctx *ctx, hipStream_t ps, int n_split, SweepJob *d_jobs, const int *d_split, const float2 *d_src,
                       const float *d_guess9, const float *d_mean, const StripTab *d_tab, float2 *d_gsrc);

// the one-wave exhaustive kernel (sfe_icp_tiny.h): n jobs, the largest target / source among them (its LDS layout)
int sweep_launch_tiny(sfe_ctx *ctx, const sfe_icp_params *p, int n, const SweepJob *d_jobs, const int *d_ids, const SweepPrep *d_preps,
                      const float2 *d_src, const float2 *d_tgt, const float *d_guess9, const int *d_perm, const float2 *d_snrm,
                      const float *d_mean, const StripTab *d_tab, float *d_T9, int32_t *d_status, int32_t *d_iters, int tiny_tmax,
                      int tiny_smax);
// limits of the one-wave kernel the host side sorts jobs by (sfe_icp_tiny.h)
#define SW_TINY_MAX 512        // most points of either cloud
#define SW_TINY_PAIRS_SHORT 400000
#define SW_TINY_PAIRS_LONG 120000
