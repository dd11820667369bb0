// Strip-sweep ICP, the loop: one workgroup per job, all iterations in one launch (+ the one-wave exhaustive kernel for
// clouds of a few hundred points and the kernel that deals a large job's queries to several workgroups).  Shared
// definitions, the search's exactness argument and the overview: sfe_icp_sweep.h.
#include "sfe_icp_sweep.h"


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
template <int NT, int MINW, bool LDS_TGT, bool LDS_Q, bool PROF, bool REC, bool MULTI>
__global__ __launch_bounds__(NT, MINW) __attribute__((amdgpu_waves_per_eu(MINW, MINW))) void icp_sweep_kernel(
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
    using TV = const float2 *;
    const TV T = LDS_TGT ? (const float2 *)lds_tgt : stgt; // sorted target incl. sentinels (LDS, or HBM scratch through L2)
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
    }
    // sorted centred target (with its NaN sentinels: a NaN stops a walk direction) -> LDS
    if (LDS_TGT) {
        for (int i = tid; i < nt + SW_PAD; i += NT)
            lds_tgt[i] = stgt[i];
    }
    // (the checkers' counters -- history length, iteration counter, iterations run -- are thread 0's alone: they live in
    // LDS between the solves instead of three VGPRs of every lane for the whole kernel)
    if (tid == 0) {
        S.chk_n[0] = 1;
        S.chk_n[1] = 0;
        S.chk_n[2] = 0;
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
    const float Cmax = sw_uniform(P.use_max_dist_filter ? fminf(r2_filter, r2_match) : r2_match);
    float Cinit;
    {
        const float h = 8.0f * tab.ext_x / (float)nt; // a few point spacings of a cloud spread along x
        Cinit = h * h;
        if (!(Cinit > 1e-30f) || !(Cinit < Cmax))
            Cinit = Cmax;
        Cinit = sw_uniform(Cinit);
    }
    float Cnext = sw_uniform(P.use_trimmed_filter ? Cinit : Cmax); // cap the next iteration starts with
    SW_PROF(0);

    const int sw_rtrips = (sw_cache >> 8) & 255;
    const bool sw_jump = (sw_cache & 2) != 0;
    const float sw_margin = sw_uniform(1.0f + 0.01f * (float)((sw_cache >> 16) & 255));
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
    float limit_prev = INFINITY; // the trimmed quantile of the previous iteration (the prediction of this one's, below)
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
        SW_STAMP(0);
        bool rec_on = false, triage = false;
        // predict: the trimmed quantile is looked for in a narrow window around the previous iteration's first (steady state:
        // the limit moves by a fraction of a percent per iteration) -- see "C" below; decided with `triage`
        bool predict = false;
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
            S.sel_below = 0u;
            S.sel_n = 0u;
        }
        auto tally_settled = [&](bool is_none, bool is_exact, float best, bool with_hist = true) { // called wave-uniformly
            const unsigned long long mn = __ballot(is_none), me = __ballot(is_exact);
            if (lane == 0) {
                if (mn)
                    atomicAdd(&S.n_none, (unsigned)__popcll(mn));
                if (me)
                    atomicAdd(&S.n_exact, (unsigned)__popcll(me));
            }
            if (!with_hist) // (a predicted quantile does not need the top-byte histogram: its fallback tallies all four passes)
                return;
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
                if (round == 0)
                    SW_STAMP(1);
                if (round == 0 && sw_rec && use_cache && it < ICP_MAX_HIST) { // (it >= 1: mvb[it - 1] = the last step)
                    const float mv_last = sw_uniform(f_add(f_mul(S.mva[it - 1], rmax), S.mvt[it - 1]));
                    const float mu = f_mul(sw_kappa, mv_last);
                    rec_on = mu < 0.5f * sqrtf(C); // (NaN -> off)
                    mu2 = sw_uniform(f_mul(mu, mu));
                    // (the first iteration with records has no count yet: the margin was sized for them to hold)
                    triage = rec_on && rec_epoch < it && (rec_epoch == it - 1 || 2u * rechit_prev >= (unsigned)ns) &&
                             (sw_cache & 32) != 0;
                    predict = triage && (sw_cache & (1 << 25)) != 0 && (sw_cache & (1 << 26)) != 0 && P.use_trimmed_filter &&
                              limit_prev < INFINITY && limit_prev > 0.0f;
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
                        tally_settled(is_none, is_exact, best, !predict);
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
                // (measured and dropped in round 6: the few dozen misses of a steady-state triage pass straight to the cooperative
                //  tier -- one wave per miss instead of one wave walking them all while fifteen wait: later passes -130 k cycles,
                //  cooperative tier +150 k, the launch unchanged at 26.3-26.4 ms per 4096 jobs (profiles/r06_icp_coop_direct_ab.txt);
                //  and the triage pass's three words per query requested two slices ahead instead of one: 37 k -> 45 k cycles
                //  per steady-state iteration -- the pass is bound by instruction issue next to the other workgroup, not by
                //  those loads)
                if (round == 0) {
                    if (REC && triage && (sw_cache & (1 << 25)) != 0)
                        triage_pass();
                    else
                        walk_pass(wl, nwork, true, sw_budget_a, false);
                    __syncthreads();
                    SW_PROF(1);
                    SW_STAMP(2);
                    if (PROF && tid == 0)
                        S.prof[SW_PI(11)] += S.mid_n;
                    walk_pass(Q.mid, S.mid_n, false, sw_budget, true);
                } else {
                    walk_pass(wl, nwork, false, sw_budget, true);
                }
                __syncthreads();
                SW_PROF(6);
                SW_STAMP(3);
                const int nlong = S.long_n;
                if (PROF && tid == 0) {
                    S.prof[SW_PI(9)] += 1;
                    S.prof[SW_PI(10)] += nlong;
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
                            S.prof[SW_PI(13)] += t_ - tp0;
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
                                    atomicAdd((unsigned long long *)&S.prof[SW_PI(12)], 1ull);
                                if (PROF)
                                    c_coop += 64ull * U;
                                if (doneL && doneR)
                                    break;
                            }
                        }
                        if (PROF && tid == 0) {
                            const long long t_ = clock64();
                            S.prof[SW_PI(14)] += t_ - tp0;
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
                            S.prof[SW_PI(15)] += clock64() - tp0;
                    }
                }
                __syncthreads();
                SW_PROF(7);
                SW_STAMP(4);
                if (PROF && tid == 0 && S.chk_n[2] == 0 && round < 8) { // first iteration, round by round
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
        SW_STAMP(5);
        if (PROF && tid == 0 && S.chk_n[2] < 32) {
            S.prof_it[2 * S.chk_n[2]] = clock64() - S.prof_b0;
            S.prof_it[2 * S.chk_n[2] + 1] = ((long long)__float_as_uint(C) << 32) | (nexact & 0xFFFFu) |
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
            } else if (!limit_inf && predict) {
                // Steady state (round 6): the k-th smallest exact d2 is almost where it was one iteration ago.  ONE pass over the
                // results counts the values below a window of +-2^17 ulps (0.8 .. 1.6 %) around the previous limit and collects
                // those inside it (a few dozen of 4000); when the rank falls into the window -- it does, once the clouds have
                // converged -- one wave picks the value out of the list by counting.  Exact: the window's values are all there
                // and everything below is counted; any other outcome (rank outside, list full) takes the radix select, all four
                // passes (the top-byte histogram was not tallied in this iteration).  15-17 k -> ~6 k cycles per iteration.
                const unsigned pb = __float_as_uint(limit_prev);
                const unsigned w_lo = pb > (1u << 17) ? pb - (1u << 17) : 0u, w_hi = pb + (1u << 17);
                unsigned n_lo = 0; // (wave-uniform)
                for (int base = 0; base < ns; base += NT) {
                    const int i = base + tid;
                    unsigned u = 0xFFFFFFFFu;
                    if (i < ns && Pz(i) >= 0)
                        u = __float_as_uint(Dz(i));
                    n_lo += (unsigned)__popcll(__ballot(u < w_lo));
                    const bool in = u >= w_lo && u <= w_hi;
                    const unsigned long long mi = __ballot(in);
                    if (mi) {
                        unsigned b0 = 0;
                        if (lane == 0)
                            b0 = atomicAdd(&S.sel_n, (unsigned)__popcll(mi));
                        b0 = (unsigned)__builtin_amdgcn_readfirstlane((int)b0);
                        const unsigned at = b0 + (unsigned)__popcll(mi & ((1ull << lane) - 1ull));
                        if (in && at < 256u)
                            S.hist[at] = u;
                    }
                }
                if (lane == 0 && n_lo)
                    atomicAdd(&S.sel_below, n_lo);
                __syncthreads();
                const unsigned n_below = S.sel_below, n_win = S.sel_n;
                const bool hit = n_win <= 256u && ksel >= n_below && ksel - n_below < n_win; // (workgroup-uniform)
                if (hit) {
                    if (tid < 64) {
                        const unsigned r = ksel - n_below;
                        unsigned e[4], less[4] = {0, 0, 0, 0}, leq[4] = {0, 0, 0, 0};
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            e[c] = (unsigned)(64 * c + lane) < n_win ? S.hist[64 * c + lane] : 0xFFFFFFFFu;
                        const int nc = (int)((n_win + 63u) >> 6); // chunks of 64 list entries (<= 4)
                        for (int c = 0; c < nc; ++c) {
                            const unsigned v = c == 0 ? e[0] : c == 1 ? e[1] : c == 2 ? e[2] : e[3]; // chunk c, one entry per lane
                            const int nj = min(64, (int)n_win - 64 * c);
                            for (int j = 0; j < nj; ++j) {
                                const unsigned x = (unsigned)__builtin_amdgcn_readlane((int)v, j);
#pragma unroll
                                for (int k2 = 0; k2 < 4; ++k2)
                                    if (k2 < nc) { // (uniform)
                                        less[k2] += x < e[k2];
                                        leq[k2] += x <= e[k2];
                                    }
                            }
                        }
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            if ((unsigned)(64 * c + lane) < n_win && less[c] <= r && r < leq[c])
                                S.sel_prefix = e[c]; // (every lane that passes holds the same value)
                    }
                    __syncthreads();
                    limit = sw_uniform(__uint_as_float(S.sel_prefix));
                } else {
                    limit = select_kth(ksel, false, false);
                }
            } else if (!limit_inf) {
                limit = select_kth(ksel, false, true);
            }
        }
        __syncthreads();
        if (fail)
            break;
        limit_prev = limit;
        // the next iteration's cap: this limit plus a margin (the clouds keep moving a little: without it about
        // every fourth converged iteration finds one match too few inside the cap and has to search twice)
        Cnext = sw_uniform(P.use_trimmed_filter ? fminf(fmaxf(limit * sw_margin, Cinit * 0.0625f), Cmax) : Cmax);
        SW_PROF(3);
        SW_STAMP(6);

        // ---- D: error minimiser sums over the kept pairs: ONE pass over the results, nine fp64 accumulators per lane.
        // The tenth sum -- the number of kept pairs -- is an integer: counted per wave by ballot + popcount in an SGPR
        // (a sum of 1.0s is exact in any order, so the canonical fp64 total has these very bits).  Rounds 1-5 ran the
        // sums in two halves of five accumulators; each half gathered (source point, neighbour, normal) again -- ten
        // exposed memory round trips per iteration instead of five (profiles/r05_stage_times.txt: 34 k cycles).
        // (The two-pass form left the source after the A/B of round 6: profiles/r06_icp_sums_ab.txt, 1.03 M -> 0.64 M cycles per job.)
        // The order of these sums is that of a 1024-thread workgroup whatever NT is: query i belongs to thread i mod 1024,
        // 64 consecutive threads are a wave (its fixed tree), the 16 wave totals are added left to right.  The smaller
        // builds play those waves one after the other.  (On a rank-deficient problem -- a target of three points -- the
        // sums are rounding noise that the solve amplifies without bound: only the same order gives the same result
        // as the other kernels; tools/icp_soak.py found such jobs at 6 in 100 000 before.)
        {
            asm volatile("; SUMS_BEGIN"); // (markers in the ISA, nothing else: tools/icp_isa.sh cuts the region between them)
            constexpr int NA = 9;
            // kept pair?  (called wave-uniformly; counts the kept pairs of the wave on the way)
            auto kept = [&](int i, bool in, int &id, unsigned &cnt) -> bool {
                float d = 0.0f;
                id = -1;
                if (in) {
                    id = Pz(i);
                    d = Dz(i);
                }
                const bool ok = in && id >= 0 && (!P.use_max_dist_filter || d <= r2_filter) &&
                                (!P.use_trimmed_filter || d <= limit);
                cnt += (unsigned)__popcll(__ballot(ok));
                return ok;
            };
            const int roles = min(16, (ns + 63) >> 6); // (the waves beyond hold no query: their totals are 0.0)
            for (int w0 = tid >> 6; w0 < roles; w0 += NT / 64) { // (NT == 1024: every wave plays itself)
                double a[NA] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
                unsigned cnt = 0;
                if (P.minimizer == 0) { // (a loop per minimiser: eight / nine live accumulators, no merged tails)
                    for (int base = 64 * w0; base < ns; base += 1024) {
                        const int i = base + lane;
                        int id;
                        if (kept(i, i < ns, id, cnt)) {
                            const float2 p = xform(Ti, src[i]);
                            const float2 q = T[id + 1];
                            const double px = p.x, py = p.y, qx = q.x, qy = q.y;
                            a[0] += px;
                            a[1] += py;
                            a[2] += qx;
                            a[3] += qy;
                            a[4] += qx * px;
                            a[5] += qx * py;
                            a[6] += qy * px;
                            a[7] += qy * py;
                        }
                    }
                } else {
                    for (int base = 64 * w0; base < ns; base += 1024) {
                        const int i = base + lane;
                        int id;
                        if (kept(i, i < ns, id, cnt)) {
                            const float2 sp = src[i]; // (both requests leave before either answer is waited for)
                            const float2 n = snrm[id];
                            const float2 q = T[id + 1];
                            const float2 p = xform(Ti, sp);
                            const double px = p.x, py = p.y, nx = n.x, ny = n.y;
                            const double a0 = px * ny - py * nx;
                            const double e = nx * (px - (double)q.x) + ny * (py - (double)q.y);
                            a[0] += a0 * a0;
                            a[1] += a0 * nx;
                            a[2] += a0 * ny;
                            a[3] += nx * nx;
                            a[4] += nx * ny;
                            a[5] += ny * ny;
                            a[6] += -(a0 * e);
                            a[7] += -(nx * e);
                            a[8] += -(ny * e);
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < NA; ++k) {
                    const double sk = wave_sum(a[k]);
                    if (lane == 0)
                        S.red[NA * w0 + k] = sk;
                }
                if (lane == 0)
                    S.xr[w0] = cnt; // (the exchange words of the split jobs are idle between their exchanges)
            }
            __syncthreads();
            if (tid < NA) {
                double sk = 0;
                for (int w = 0; w < roles; ++w)
                    sk += S.red[NA * w + tid];
                S.acc[1 + tid] = sk;
            } else if (tid == NA) {
                unsigned c = 0;
                for (int w = 0; w < roles; ++w)
                    c += S.xr[w];
                S.acc[0] = (double)c;
            }
            __syncthreads();
            asm volatile("; SUMS_END");
        }
        if (MULTI)
            xreduce_acc(); // the sums over the queries of every share
        SW_PROF(4);
        SW_STAMP(7);

        // ---- E: solve, compose, check (one lane) ----
        if (tid == 0) {
            int status, iterate;
            double acc[10];
            for (int i = 0; i < 10; ++i)
                acc[i] = S.acc[i];
            IcpCheck chk = {S.hist_c, S.hist_s, S.hist_x, S.hist_y, S.chk_n[0], S.chk_n[1], S.chk_n[2]};
            icp_solve_and_check(P, acc, Ti, S.Ti, chk, status, iterate);
            S.chk_n[0] = chk.nhist;
            S.chk_n[1] = chk.counter;
            S.chk_n[2] = chk.iters;
            S.flag_status = status;
            S.flag_iterate = (status == SFE_ICP_OK) ? iterate : 0;
        }
        __syncthreads();
        SW_PROF(5);
        SW_STAMP(8);
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
        iters_out[J.out] = S.chk_n[2];
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
            atomicAdd((unsigned long long *)&prof[84], (unsigned long long)S.chk_n[2]);
    }
}

#undef SW_TINY_MAX
#undef SW_TINY_PAIRS_SHORT
#undef SW_TINY_PAIRS_LONG
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

static int pow2_floor(size_t v)
{
    size_t p = 1;
    while (2 * p <= v)
        p *= 2;
    return (int)p;
}

template <int NT, int MINW, bool LDS_TGT, bool LDS_Q, bool PROF, bool REC, bool MULTI>
int sweep_launch_loop(const SweepLaunchArgs &a, int n, const int *d_ids, size_t body, int t_cap, int q_cap)
{
    sfe_ctx *ctx = a.ctx;
    auto kernel = icp_sweep_kernel<NT, MINW, LDS_TGT, LDS_Q, PROF, REC, MULTI>;
    const size_t smem = sweep_ctl_bytes<NT, PROF, REC>() + body;
    SFE_HIP(ctx, hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(kernel, dim3(n), dim3(NT), smem, ctx->stream, *a.p, a.d_jobs, d_ids, a.d_src, a.d_guess9, a.d_stgt,
                       a.d_perm, a.d_snrm, a.d_mean, a.d_tab, a.d_grid, a.d_qst, a.d_qwl, a.d_qssrc, a.d_nn_d2, a.d_nn_pos,
                       a.d_T9, a.d_status, a.d_iters, a.d_prof, a.d_dbg, a.sw_budget, a.sw_budget_a, a.sw_cache, t_cap, q_cap,
                       pow2_floor(body / 8), a.sw_m, a.sw_kappa, a.d_sync, a.sw_cache2);
    SFE_LAUNCH_CHECK(ctx);
    return 0;
}

// the builds the host side asks for: {threads, min waves per EU (the VGPR budget), target in LDS, results in LDS, counted
// profile, clearance records, job shared by several workgroups}
#define SW_LOOP_INST(...) template int sweep_launch_loop<__VA_ARGS__>(const SweepLaunchArgs &, int, const int *, size_t, int, int);
#ifdef SW_INSPECT // (tools/icp_isa.sh: the bench's build alone, for a look at its ISA)
SW_LOOP_INST(ICP_THREADS, 8, true, true, false, true, false)
#else
SW_LOOP_INST(SW_T0_NT, 4, true, true, false, true, false)
SW_LOOP_INST(SW_T0_NT, 4, true, true, false, false, false)
SW_LOOP_INST(SW_T1_NT, 4, true, true, true, false, false)
SW_LOOP_INST(SW_T1_NT, 8, true, true, false, false, false)
SW_LOOP_INST(SW_T1_NT, 4, true, true, false, true, false)
SW_LOOP_INST(SW_T1_NT, 4, true, true, false, false, false)
SW_LOOP_INST(ICP_THREADS, 4, true, true, false, false, false)
SW_LOOP_INST(ICP_THREADS, 8, true, true, true, true, false)
SW_LOOP_INST(ICP_THREADS, 8, true, true, true, false, false)
SW_LOOP_INST(ICP_THREADS, 8, true, true, false, true, false)
SW_LOOP_INST(ICP_THREADS, 8, true, true, false, false, false)
SW_LOOP_INST(ICP_THREADS, 4, true, false, false, false, false)
SW_LOOP_INST(ICP_THREADS, 8, true, false, true, false, false)
SW_LOOP_INST(ICP_THREADS, 8, true, false, false, true, false)
SW_LOOP_INST(ICP_THREADS, 8, true, false, false, false, false)
SW_LOOP_INST(ICP_THREADS, 4, false, false, false, false, false)
SW_LOOP_INST(ICP_THREADS, 8, false, false, false, false, false)
SW_LOOP_INST(ICP_THREADS, 4, false, false, false, false, true)
#endif

int sweep_launch_split(sfe_ctx *ctx, hipStream_t ps, int n_split, SweepJob *d_jobs, const int *d_split, const float2 *d_src,
                       const float *d_guess9, const float *d_mean, const StripTab *d_tab, float2 *d_gsrc)
{
    hipLaunchKernelGGL(icp_split_kernel<ICP_THREADS>, dim3(n_split), dim3(ICP_THREADS), 0, ps, d_jobs, d_split, d_src, d_guess9,
                       d_mean, d_tab, d_gsrc);
    SFE_LAUNCH_CHECK(ctx);
    return 0;
}

int sweep_launch_tiny(sfe_ctx *ctx, const sfe_icp_params *p, int n, const SweepJob *d_jobs, const int *d_ids, const SweepPrep *d_preps,
                      const float2 *d_src, const float2 *d_tgt, const float *d_guess9, const int *d_perm, const float2 *d_snrm,
                      const float *d_mean, const StripTab *d_tab, float *d_T9, int32_t *d_status, int32_t *d_iters, int tiny_tmax,
                      int tiny_smax)
{
    const int tc = ((tiny_tmax + SW_TINY_CH - 1) / SW_TINY_CH + 1) * SW_TINY_CH, qc = (tiny_smax + 3) & ~3; // (whole chunks + one of padding)
    const size_t smem = ((sizeof(TinyShared) + 15) & ~(size_t)15) + sizeof(float2) * (size_t)tc * (p->minimizer == 1 ? 2 : 1) +
                        6 * (size_t)qc;
    hipLaunchKernelGGL(icp_tiny_kernel, dim3(n), dim3(SW_TINY_NT), smem, ctx->stream, *p, d_jobs, d_ids, d_preps, d_src, d_tgt,
                       d_guess9, d_perm, d_snrm, d_mean, d_tab, d_T9, d_status, d_iters, tc, qc);
    SFE_LAUNCH_CHECK(ctx);
    return 0;
}
