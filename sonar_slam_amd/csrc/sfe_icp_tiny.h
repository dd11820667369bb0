// Scan matches on small clouds: ONE WAVE per job, exhaustive nearest-neighbour search.
// (included by sfe_icp_sweep.hip: it shares that file's job tables and the targets' preparation)
//
// bruce_slam's own scan matches are feature clouds of 10^2 .. 10^3 points (slam.py:769,1032).  At a few
// hundred points the strip sweep's machinery -- query sort, witnesses, search rounds and tiers, work lists, the
// suspended-query rounds of the trimmed quantile -- costs more than the search it saves: n_src x n_tgt = 200 x 200
// pair evaluations are 4 400 wave instructions, a few microseconds.  So below SW_TINY_PAIRS pairs a job takes this
// kernel: the centred target, its normals and every per-query result live in LDS, the wave never meets a barrier, and
// an iteration is
//     transform -> all-pairs arg-min (LDS broadcast reads, four queries per lane in registers) -> exact radix select
//     of the trimmed quantile -> fp64 sums of the error minimiser -> closed-form solve and checkers on one lane.
// Same chain, same arithmetic (dist2(), affine1(), the ten sums, icp_solve_and_check) and the same decisions as the
// strip-sweep and the brute-force kernels: the target is scanned in ORIGINAL index order with a strict '<', so ties go
// to the lowest original index like everywhere else; results equal theirs bit for bit (tests/test_gpu_icp.py runs
// every case with and without this kernel).
// Replaces bruce_slam/src/bruce_slam/cpp/pcl.cpp:198-212 for the job sizes the live system produces.
#pragma once

#define SW_TINY_NT 64
#define SW_TINY_MAX 512        // most points of either cloud
// most n_src x n_tgt pairs: chains that stop after a few iterations (the shipped one: differential checker, ~5 iterations,
// clouds still decimetres apart -- the regime where the sweep's windows are wide) / chains forced through many iterations
// (converged clouds: the sweep evaluates ~8 candidates per query there).  Measured, jobs per second tiny vs sweep:
// 200 x 200 shipped 8.1 M vs 4.8 M, 30 forced iterations 2.55 M vs 1.60 M; 500 x 500 shipped 3.0 M vs 2.0 M, 30 forced 0.55 M vs 0.73 M
#define SW_TINY_PAIRS_SHORT 400000
#define SW_TINY_PAIRS_LONG 120000
#define SW_TINY_QPL 4          // queries per lane and pass over the target
#define SW_TINY_CH 16          // target points per chunk of the two-level arg-min

struct TinyShared {
    double red[10];
    unsigned hist[256];
    float hist_c[ICP_MAX_HIST], hist_s[ICP_MAX_HIST], hist_x[ICP_MAX_HIST], hist_y[ICP_MAX_HIST];
    float Ti[9];
    int flag_iterate, flag_status;
    unsigned sel_k, sel_prefix;
};

// dynamic LDS behind TinyShared: [tgt: t_cap float2][nrm: t_cap float2 (point-to-plane)][d2: q_cap float][idx: q_cap int16]
__global__ __launch_bounds__(SW_TINY_NT, 4) void icp_tiny_kernel(
    sfe_icp_params P, const SweepJob *__restrict__ jobs, const int *__restrict__ job_ids, const SweepPrep *__restrict__ preps,
    const float2 *__restrict__ src_all, const float2 *__restrict__ tgt_all, const float *__restrict__ guess_all,
    const int *__restrict__ perm_all, const float2 *__restrict__ snrm_all, const float *__restrict__ mean_all,
    const StripTab *__restrict__ tab_all, float *__restrict__ T_out, int *__restrict__ status_out, int *__restrict__ iters_out,
    int t_cap, int q_cap)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    TinyShared &S = *reinterpret_cast<TinyShared *>(smem_raw);
    float2 *s_tgt = reinterpret_cast<float2 *>(smem_raw + ((sizeof(TinyShared) + 15) & ~(size_t)15));
    float2 *s_nrm = s_tgt + t_cap;
    float *s_d2 = reinterpret_cast<float *>(s_nrm + (P.minimizer == 1 ? t_cap : 0));
    short *s_idx = reinterpret_cast<short *>(s_d2 + q_cap);
    const int jb = __builtin_amdgcn_readfirstlane(job_ids[blockIdx.x]);
    const SweepJob J = jobs[jb];
    const int ns = __builtin_amdgcn_readfirstlane(J.n_src), nt = __builtin_amdgcn_readfirstlane(J.n_tgt);
    const int lane = threadIdx.x;
    const float2 *__restrict__ src = src_all + J.src_start;
    const float2 *__restrict__ tgt = tgt_all + preps[J.prep].tgt_start;
    const float *guess = guess_all + 9 * (size_t)J.out;
    // Reference mean (fp64 accumulation, rounded to float).  A point-to-point chain needs nothing else of the target's
    // preparation (no normals), so its targets are not prepared at all when only jobs of this kernel use them: the
    // mean is taken here, in the summation order of a one-wave preparation (block_sum<2, 64>).
    float mx, my;
    if (P.minimizer == 0) {
        // (64 consecutive points = one wave of the 1024-thread kernels: their tree, then the waves in order)
        double m0 = 0, m1 = 0;
        for (int i0 = 0; i0 < nt; i0 += SW_TINY_NT) {
            const int i = i0 + lane;
            const float2 t = i < nt ? tgt[i] : make_float2(0.0f, 0.0f);
            m0 += wave_sum(i < nt ? (double)t.x : 0.0);
            m1 += wave_sum(i < nt ? (double)t.y : 0.0);
        }
        mx = sw_uniform((float)(m0 / nt));
        my = sw_uniform((float)(m1 / nt));
    } else {
        mx = sw_uniform(mean_all[2 * J.prep]);
        my = sw_uniform(mean_all[2 * J.prep + 1]);
    }

    // centred target in its ORIGINAL order; the normals the preparation left by sorted position go back to that order
    for (int i = lane; i < nt; i += SW_TINY_NT) {
        const float2 t = tgt[i];
        s_tgt[i] = make_float2(f_add(t.x, -mx), f_add(t.y, -my));
    }
    if (lane < SW_TINY_CH) // the arg-min goes through the target in chunks of SW_TINY_CH: the last one is filled up with
        s_tgt[nt + lane] = make_float2(INFINITY, INFINITY); // points nothing can match
    if (P.minimizer == 1) {
        // sorted position p (strip s holds [sbeg[s], sbeg[s + 1] - 1), a sentinel behind it) -> original index perm[p - 1]
        const int *__restrict__ perm = perm_all + J.tgt_off;
        const float2 *__restrict__ snrm = snrm_all + J.tgt_off;
        const StripTab *__restrict__ tab = tab_all + J.prep;
        const int nst = __builtin_amdgcn_readfirstlane(tab->ns);
        for (int st = 0; st < nst; ++st) {
            const int p0 = __builtin_amdgcn_readfirstlane(tab->sbeg[st]), p1 = __builtin_amdgcn_readfirstlane(tab->sbeg[st + 1]) - 1;
            for (int p = p0 + lane; p < p1; p += SW_TINY_NT)
                s_nrm[perm[p - 1]] = snrm[p - 1];
        }
    }
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
    if (lane == 0) {
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
    __syncthreads(); // (one wave: orders the LDS writes above, costs nothing)
    const float r2_match = sw_uniform(f_mul(P.matcher_max_dist, P.matcher_max_dist));
    const float r2_filter = sw_uniform(f_mul(P.max_dist_filter, P.max_dist_filter));
    auto xform = [&](const float (&Ti)[9], float2 sp) { // cur = Ti * (T0 * src): the two roundings of every ICP kernel here
        const float rx = affine1(T0[0], T0[1], T0[2], sp.x, sp.y);
        const float ry = affine1(T0[3], T0[4], T0[5], sp.x, sp.y);
        return make_float2(affine1(Ti[0], Ti[1], Ti[2], rx, ry), affine1(Ti[3], Ti[4], Ti[5], rx, ry));
    };

    for (int guard = 0; guard <= P.max_iter + 1; ++guard) {
        float Ti[9];
#pragma unroll
        for (int i = 0; i < 9; ++i)
            Ti[i] = sw_uniform(S.Ti[i]);
        // ---- match: exact 1-NN of every query among ALL target points (squared distance, lowest index on ties, none
        // beyond KDTreeMatcher.maxDist) ----
        unsigned nfin = 0;
        for (int q0 = 0; q0 < ns; q0 += SW_TINY_NT * SW_TINY_QPL) {
            // Two-level exact arg-min (like the brute-force kernel's): level 1 keeps only the running minimum of each
            // chunk of SW_TINY_CH target points (v_min3, no index bookkeeping: 5.5 VALU operations per pair instead of 8) and
            // remembers the first chunk that lowered it; level 2 rescans that one chunk for the first point that attains the
            // minimum.  Strict '<' between chunks + first hit inside the chunk = lowest index on ties; NaN never wins.
            float px[SW_TINY_QPL], py[SW_TINY_QPL], best[SW_TINY_QPL];
            int bch[SW_TINY_QPL], bid[SW_TINY_QPL];
#pragma unroll
            for (int k = 0; k < SW_TINY_QPL; ++k) {
                const int i = q0 + k * SW_TINY_NT + lane;
                const float2 p = xform(Ti, src[i < ns ? i : 0]);
                px[k] = p.x;
                py[k] = p.y;
                best[k] = INFINITY;
                bch[k] = -1;
                bid[k] = -1;
            }
            for (int c0 = 0; c0 < nt; c0 += SW_TINY_CH) {
                float cmin[SW_TINY_QPL];
#pragma unroll
                for (int k = 0; k < SW_TINY_QPL; ++k)
                    cmin[k] = INFINITY;
#pragma unroll
                for (int jj = 0; jj < SW_TINY_CH; jj += 2) {
                    const float4 t = *reinterpret_cast<const float4 *>(&s_tgt[c0 + jj]); // LDS broadcast
#pragma unroll
                    for (int k = 0; k < SW_TINY_QPL; ++k)
                        cmin[k] = fminf(fminf(cmin[k], dist2(px[k], py[k], t.x, t.y)), dist2(px[k], py[k], t.z, t.w));
                }
#pragma unroll
                for (int k = 0; k < SW_TINY_QPL; ++k)
                    if (cmin[k] < best[k]) {
                        best[k] = cmin[k];
                        bch[k] = c0;
                    }
            }
#pragma unroll
            for (int k = 0; k < SW_TINY_QPL; ++k)
                if (bch[k] >= 0)
                    for (int jj = SW_TINY_CH - 1; jj >= 0; --jj) { // (descending: the lowest index is written last)
                        const float2 t = s_tgt[bch[k] + jj];
                        if (dist2(px[k], py[k], t.x, t.y) == best[k])
                            bid[k] = bch[k] + jj;
                    }
#pragma unroll
            for (int k = 0; k < SW_TINY_QPL; ++k) {
                const int i = q0 + k * SW_TINY_NT + lane;
                // an infinite distance is no match, also for an unbounded matcher
                const bool ok = i < ns && bid[k] >= 0 && best[k] <= r2_match && best[k] != INFINITY;
                if (i < ns) {
                    s_d2[i] = ok ? best[k] : INFINITY;
                    s_idx[i] = (short)(ok ? bid[k] : -1);
                }
                nfin += (unsigned)__popcll(__ballot(ok));
            }
        }
        __syncthreads();
        // ---- TrimmedDistOutlierFilter limit: exact order statistic by radix select on the float bit patterns ----
        float limit = INFINITY;
        if (P.use_trimmed_filter) {
            if (nfin == 0) { // "no outlier to filter"
                if (lane == 0)
                    S.flag_status = SFE_ICP_NO_OUTLIER;
                break;
            }
            unsigned k_sel = (P.trim_ratio >= 1.0f) ? nfin - 1 : (unsigned)f_mul((float)nfin, P.trim_ratio);
            unsigned prefix = 0;
            for (int shift = 24; shift >= 0; shift -= 8) {
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    S.hist[4 * lane + b] = 0;
                __syncthreads();
                const unsigned himask = (shift == 24) ? 0u : (0xFFFFFFFFu << (shift + 8));
                for (int i = lane; i < ns; i += SW_TINY_NT) {
                    const float d = s_d2[i];
                    if (d != INFINITY) {
                        const unsigned u = __float_as_uint(d); // d >= 0: bit pattern order == value order
                        if ((u & himask) == prefix)
                            atomicAdd(&S.hist[(u >> shift) & 255u], 1u);
                    }
                }
                __syncthreads();
                const unsigned h0 = S.hist[4 * lane], h1 = S.hist[4 * lane + 1], h2 = S.hist[4 * lane + 2], h3 = S.hist[4 * lane + 3];
                const unsigned tot = h0 + h1 + h2 + h3;
                const unsigned incl = wave_inclusive_scan(tot), excl = incl - tot;
                if (k_sel >= excl && k_sel < incl) { // exactly one lane
                    unsigned r = k_sel - excl, b = 4 * lane;
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
                __syncthreads();
                k_sel = S.sel_k;
                prefix = S.sel_prefix;
            }
            limit = sw_uniform(__uint_as_float(prefix));
        }
        // ---- error minimiser: sums over the kept pairs, fp64, in the order of the 1024-thread kernels (query i = their
        // thread i, 64 consecutive queries = one of their waves: the wave's fixed tree, then the waves left to right).
        // The sums of a rank-deficient problem (a target of three points) are rounding noise that the solve amplifies
        // without bound: with any other order such jobs would come out differently from the other kernels. ----
        double acc[10];
#pragma unroll
        for (int k = 0; k < 10; ++k)
            acc[k] = 0.0;
        for (int i0 = 0; i0 < ns; i0 += SW_TINY_NT) {
            const int i = i0 + lane;
            double t[10];
#pragma unroll
            for (int k = 0; k < 10; ++k)
                t[k] = 0.0;
            const int id = i < ns ? (int)s_idx[i] : -1;
            const float d = i < ns ? s_d2[i] : INFINITY;
            const bool ok = id >= 0 && (!P.use_max_dist_filter || d <= r2_filter) && (!P.use_trimmed_filter || d <= limit);
            if (ok) {
                const float2 p = xform(Ti, src[i]);
                const double px = p.x, py = p.y;
                const float2 q = s_tgt[id];
                const double qx = q.x, qy = q.y;
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
                } else {
                    const float2 n = s_nrm[id];
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
            }
            const int nk = P.minimizer == 0 ? 9 : 10;
#pragma unroll
            for (int k = 0; k < 10; ++k)
                if (k < nk)
                    acc[k] += wave_sum(0.0 + t[k]);
        }
        // ---- solve, compose, check (one lane) ----
        if (lane == 0) {
            int status, iterate;
            icp_solve_and_check(P, acc, Ti, S.Ti, chk, status, iterate);
            S.flag_status = status;
            S.flag_iterate = (status == SFE_ICP_OK) ? iterate : 0;
        }
        __syncthreads();
        if (!S.flag_iterate)
            break;
    }
    __syncthreads();
    if (lane == 0) {
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
    }
}
