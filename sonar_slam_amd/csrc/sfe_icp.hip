// ICP scan matching, nearest-neighbour match and radius outlier filter on gfx950.
// Replaces bruce_slam/src/bruce_slam/cpp/pcl.cpp:54-74 (remove_outlier), :161-174 (match) and
// :198-212 (ICP.compute -> libpointmatcher chain of bruce_slam/config/icp.yaml:1-31).
//
// The libpointmatcher chain (restated in oracle/sonar_oracle.c):
//   centre the target on its mean, T0 = T_mean^-1 * guess, reading = T0 * source   (once)
//   loop: cur = T_iter * reading
//         exact 1-NN of every cur point in the target (squared distance, lowest index on ties,
//           none beyond KDTreeMatcher.maxDist)
//         weights = [d2 <= MaxDist^2] * [d2 <= quantile_ratio(finite d2)]          (0/1)
//         T_step  = weighted Kabsch (point-to-point) or 2-D point-to-plane normal equations
//         T_iter  = T_step * T_iter ; Counter / Differential checkers
//   result = T_mean * T_iter * T0
//
// Mapping: ONE workgroup (1024 threads, 16 waves) runs the whole ICP of one job in one launch;
// the mean-centred target sits in LDS for all iterations (<= 8192 points; larger targets are
// streamed through the same LDS tile), every lane owns a few source points, the trimmed
// quantile is an exact radix select on the float bit patterns (LDS histogram), and the 9 (or
// 9+1) Gauss-Newton / Kabsch accumulators are reduced in fp64 with wave shuffles + LDS.
// Independent jobs (keyframes, or many initial guesses of one loop-closure pair) are the grid.
// There is no dense contraction here: fp32 VALU + LDS, no MFMA.
#include "sfe_internal.h"

#include <algorithm>
#include <cmath>
#include <cstring>

#include "sfe_icp_common.h"

// Two-level exact arg-min over one LDS tile of centred target points for NP query points per
// lane.  Level 1 keeps only the running minimum of each 16-point chunk (v_min3, no index
// bookkeeping: 5.5 VALU ops per pair instead of 8) and remembers the first chunk that lowered
// the minimum; level 2 (nn_resolve) rescans that one chunk for the first point that attains it.
// Strict '<' between chunks + first hit inside the chunk == lowest index on ties.
// The tile is padded to a multiple of ICP_CH with +inf points (distance inf, never selected).
template <int NP>
__device__ __forceinline__ void nn_scan_tile(const float2 *__restrict__ s_tgt, int tile_n, int chunk_base,
                                             const float (&px)[ICP_PB], const float (&py)[ICP_PB],
                                             float (&best)[ICP_PB], int (&bchunk)[ICP_PB])
{
    const int nchunks = (tile_n + ICP_CH - 1) / ICP_CH;
    for (int c = 0; c < nchunks; ++c) {
        float cmin[NP];
#pragma unroll
        for (int k = 0; k < NP; ++k)
            cmin[k] = INFINITY;
#pragma unroll
        for (int jj = 0; jj < ICP_CH; jj += 2) {
            const float4 t = *reinterpret_cast<const float4 *>(&s_tgt[c * ICP_CH + jj]); // LDS broadcast
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const float d0 = dist2(px[k], py[k], t.x, t.y);
                const float d1 = dist2(px[k], py[k], t.z, t.w);
                cmin[k] = fminf(fminf(cmin[k], d0), d1);
            }
        }
#pragma unroll
        for (int k = 0; k < NP; ++k)
            if (cmin[k] < best[k]) {
                best[k] = cmin[k];
                bchunk[k] = chunk_base + c;
            }
    }
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// Same arg-min, two query points per VALU instruction: (px[k], px[k+1]) live in one 64-bit
// register pair and v_pk_add/mul_f32 evaluate both squared distances with per-component IEEE
// rounding, i.e. bit-identical to dist2().  Measured 7 % faster than the scalar form on MI355X
// (packed fp32 issues at half rate, the gain is fewer instructions to fetch/decode).
template <int NP>
__device__ __forceinline__ void nn_scan_tile_pk(const float2 *__restrict__ s_tgt, int tile_n, int chunk_base,
                                                const float (&px)[ICP_PB], const float (&py)[ICP_PB],
                                                float (&best)[ICP_PB], int (&bchunk)[ICP_PB])
{
    constexpr int NH = (NP + 1) / 2;
    f32x2 qx[NH], qy[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        qx[h] = (f32x2){px[2 * h], px[(2 * h + 1 < NP) ? 2 * h + 1 : 2 * h]};
        qy[h] = (f32x2){py[2 * h], py[(2 * h + 1 < NP) ? 2 * h + 1 : 2 * h]};
    }
    const int nchunks = (tile_n + ICP_CH - 1) / ICP_CH;
    for (int c = 0; c < nchunks; ++c) {
        f32x2 cmin[NH];
#pragma unroll
        for (int h = 0; h < NH; ++h)
            cmin[h] = (f32x2){INFINITY, INFINITY};
#pragma unroll
        for (int jj = 0; jj < ICP_CH; jj += 2) {
            const float4 t = *reinterpret_cast<const float4 *>(&s_tgt[c * ICP_CH + jj]); // LDS broadcast
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                f32x2 dx = qx[h] - t.x, dy = qy[h] - t.y;
                const f32x2 d0 = dx * dx + dy * dy;
                dx = qx[h] - t.z;
                dy = qy[h] - t.w;
                const f32x2 d1 = dx * dx + dy * dy;
                cmin[h].x = fminf(fminf(cmin[h].x, d0.x), d1.x);
                cmin[h].y = fminf(fminf(cmin[h].y, d0.y), d1.y);
            }
        }
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const float cm = (k & 1) ? cmin[k / 2].y : cmin[k / 2].x;
            if (cm < best[k]) {
                best[k] = cm;
                bchunk[k] = chunk_base + c;
            }
        }
    }
}

// load target points [base, base+n) of the job, centred on `mean`, into the LDS tile (+inf pad)
__device__ __forceinline__ void load_tile(float2 *__restrict__ s_tgt, const float2 *__restrict__ tgt, int base,
                                          int n, float mx, float my)
{
    const int npad = (n + ICP_CH - 1) / ICP_CH * ICP_CH;
    for (int i = threadIdx.x; i < npad; i += ICP_THREADS) {
        float2 v = make_float2(INFINITY, INFINITY);
        if (i < n) {
            const float2 t = tgt[base + i];
            v = make_float2(f_add(t.x, -mx), f_add(t.y, -my));
        }
        s_tgt[i] = v;
    }
}

struct IcpShared {
    float2 tgt[ICP_TCAP];
    double red[ICP_WAVES * 10 + 10];
    unsigned hist[256];
    unsigned sel_prefix, sel_k;
    int flag_iterate, flag_status;
    float Ti[9];
    float mean[2];
    float hist_c[ICP_MAX_HIST], hist_s[ICP_MAX_HIST], hist_x[ICP_MAX_HIST], hist_y[ICP_MAX_HIST];
};

template <int MINW>
__global__ __launch_bounds__(ICP_THREADS, MINW) void icp_job_kernel(sfe_icp_params P, int nn_variant,
                                                              const IcpJob *__restrict__ jobs,
                                                              const float2 *__restrict__ src_all,
                                                              const float2 *__restrict__ tgt_all,
                                                              const float *__restrict__ guess_all,
                                                              float *__restrict__ nn_d2_all,
                                                              int *__restrict__ nn_idx_all,
                                                              float2 *__restrict__ nrm_all,
                                                              float *__restrict__ T_out, int *__restrict__ status_out,
                                                              int *__restrict__ iters_out)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    IcpShared &S = *reinterpret_cast<IcpShared *>(smem_raw);

    const IcpJob J = jobs[blockIdx.x];
    const int ns = J.n_src, nt = J.n_tgt;
    const float2 *__restrict__ src = src_all + J.src_start;
    const float2 *__restrict__ tgt = tgt_all + J.tgt_start;
    float *__restrict__ nn_d2 = nn_d2_all + J.scratch_off;
    int *__restrict__ nn_idx = nn_idx_all + J.scratch_off;
    float2 *__restrict__ nrm = nrm_all ? nrm_all + J.nrm_off : nullptr;
    const float *guess = guess_all + 9 * (size_t)blockIdx.x;
    const int tid = threadIdx.x;
    const bool resident = nt <= ICP_TCAP;

    // ---- reference mean (fp64 accumulation, rounded to float): frame refMean ----
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
        }
        __syncthreads();
    }
    const float mx = S.mean[0], my = S.mean[1];
    if (resident) {
        load_tile(S.tgt, tgt, 0, nt, mx, my);
        __syncthreads();
    }

    // ---- point-to-plane only: PCA normals of the centred target (k nearest incl. self) ----
    if (P.minimizer == 1) {
        const int K = min(min(P.normals_knn, ICP_KMAX), nt);
        for (int i0 = 0; i0 < nt; i0 += ICP_THREADS) {
            const int i = i0 + tid;
            const bool act = i < nt;
            float qx = 0, qy = 0;
            if (act) {
                const float2 t = tgt[i];
                qx = f_add(t.x, -mx);
                qy = f_add(t.y, -my);
            }
            float bd[ICP_KMAX];
            int bi[ICP_KMAX];
            float kth = INFINITY;
#pragma unroll
            for (int q = 0; q < ICP_KMAX; ++q) {
                bd[q] = INFINITY;
                bi[q] = 0;
            }
            for (int tb = 0; tb < nt; tb += ICP_TCAP) {
                const int tn = min(ICP_TCAP, nt - tb);
                if (!resident) {
                    __syncthreads();
                    load_tile(S.tgt, tgt, tb, tn, mx, my);
                    __syncthreads();
                }
                if (act) {
                    for (int j = 0; j < tn; ++j) {
                        const float2 t = S.tgt[j];
                        const float dx = f_add(qx, -t.x), dy = f_add(qy, -t.y);
                        const float d = f_add(f_mul(dx, dx), f_mul(dy, dy));
                        if (d < kth) { // kth = current K-th smallest (inf until K points were seen)
                            // position = number of kept entries <= d (ties keep the earlier index)
                            int p = 0;
#pragma unroll
                            for (int q = 0; q < ICP_KMAX; ++q)
                                p += (q < K && bd[q] <= d) ? 1 : 0;
#pragma unroll
                            for (int q = ICP_KMAX - 1; q >= 1; --q) {
                                if (q < K && q > p) {
                                    bd[q] = bd[q - 1];
                                    bi[q] = bi[q - 1];
                                }
                            }
#pragma unroll
                            for (int q = 0; q < ICP_KMAX; ++q) {
                                if (q == p) {
                                    bd[q] = d;
                                    bi[q] = tb + j;
                                }
                                if (q == K - 1)
                                    kth = bd[q];
                            }
                        }
                    }
                }
            }
            if (act) {
                double sx = 0, sy = 0;
#pragma unroll
                for (int q = 0; q < ICP_KMAX; ++q)
                    if (q < K) {
                        const float2 t = tgt[bi[q]];
                        sx += (double)f_add(t.x, -mx);
                        sy += (double)f_add(t.y, -my);
                    }
                sx /= K;
                sy /= K;
                double a = 0, b = 0, d = 0;
#pragma unroll
                for (int q = 0; q < ICP_KMAX; ++q)
                    if (q < K) {
                        const float2 t = tgt[bi[q]];
                        const double ux = (double)f_add(t.x, -mx) - sx, uy = (double)f_add(t.y, -my) - sy;
                        a += ux * ux;
                        b += ux * uy;
                        d += uy * uy;
                    }
                const double u = a - d, v = 2 * b, h = sqrt(u * u + v * v);
                double tx, ty;
                if (h == 0) {
                    tx = 1;
                    ty = 0;
                } else if (u >= 0) {
                    tx = u + h;
                    ty = v;
                } else {
                    tx = v;
                    ty = h - u;
                }
                double nn = sqrt(tx * tx + ty * ty);
                if (nn == 0) {
                    tx = 1;
                    ty = 0;
                    nn = 1;
                }
                nrm[i] = make_float2((float)(-ty / nn), (float)(tx / nn));
            }
        }
        __syncthreads(); // normals visible to the whole workgroup (same CU: L1-coherent stores + barrier)
        __threadfence_block();
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
    if (tid == 0) {
        const float I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        for (int i = 0; i < 9; ++i)
            S.Ti[i] = I[i];
        S.flag_iterate = 1;
        S.flag_status = SFE_ICP_OK;
    }
    __syncthreads();

    // checker state: history in LDS (only thread 0 touches it), counters in thread 0's registers
    IcpCheck chk = {S.hist_c, S.hist_s, S.hist_x, S.hist_y, 1, 0, 0};
    if (tid == 0) {
        S.hist_c[0] = 1.0f; // DifferentialTransformationChecker::init pushes the identity
        S.hist_s[0] = 0.0f;
        S.hist_x[0] = 0.0f;
        S.hist_y[0] = 0.0f;
    }

    const float r2_match = f_mul(P.matcher_max_dist, P.matcher_max_dist);
    const float r2_filter = f_mul(P.max_dist_filter, P.max_dist_filter);

    while (true) {
        float Ti[9];
#pragma unroll
        for (int i = 0; i < 9; ++i)
            Ti[i] = S.Ti[i];

        // ---- match: exact 1-NN of cur = Ti * (T0 * src) in the centred target ----
        double nfin_d[1] = {0};
        for (int base = 0; base < ns; base += ICP_THREADS * ICP_PB) {
            const int np = min(ICP_PB, (ns - base + ICP_THREADS - 1) / ICP_THREADS); // uniform
            float px[ICP_PB], py[ICP_PB], best[ICP_PB];
            int bchunk[ICP_PB];
#pragma unroll
            for (int k = 0; k < ICP_PB; ++k) {
                const int i = base + k * ICP_THREADS + tid;
                float x = 0, y = 0;
                if (k < np && i < ns) {
                    const float2 s = src[i];
                    const float rx = affine1(T0[0], T0[1], T0[2], s.x, s.y);
                    const float ry = affine1(T0[3], T0[4], T0[5], s.x, s.y);
                    x = affine1(Ti[0], Ti[1], Ti[2], rx, ry);
                    y = affine1(Ti[3], Ti[4], Ti[5], rx, ry);
                }
                px[k] = x;
                py[k] = y;
                best[k] = INFINITY;
                bchunk[k] = -1;
            }
            for (int tb = 0; tb < nt; tb += ICP_TCAP) {
                const int tn = min(ICP_TCAP, nt - tb);
                if (!resident) {
                    __syncthreads();
                    load_tile(S.tgt, tgt, tb, tn, mx, my);
                    __syncthreads();
                }
                if (nn_variant == 0) {
                    switch (np) { // one instantiation per points-per-lane count keeps everything in registers
                    case 1: nn_scan_tile_pk<1>(S.tgt, tn, tb / ICP_CH, px, py, best, bchunk); break;
                    case 2: nn_scan_tile_pk<2>(S.tgt, tn, tb / ICP_CH, px, py, best, bchunk); break;
                    case 3: nn_scan_tile_pk<3>(S.tgt, tn, tb / ICP_CH, px, py, best, bchunk); break;
                    case 4: nn_scan_tile_pk<4>(S.tgt, tn, tb / ICP_CH, px, py, best, bchunk); break;
                    case 5: nn_scan_tile_pk<5>(S.tgt, tn, tb / ICP_CH, px, py, best, bchunk); break;
                    case 6: nn_scan_tile_pk<6>(S.tgt, tn, tb / ICP_CH, px, py, best, bchunk); break;
                    case 7: nn_scan_tile_pk<7>(S.tgt, tn, tb / ICP_CH, px, py, best, bchunk); break;
                    default: nn_scan_tile_pk<8>(S.tgt, tn, tb / ICP_CH, px, py, best, bchunk); break;
                    }
                } else {
                    switch (np) {
                    case 1: nn_scan_tile<1>(S.tgt, tn, tb / ICP_CH, px, py, best, bchunk); break;
                    case 2: nn_scan_tile<2>(S.tgt, tn, tb / ICP_CH, px, py, best, bchunk); break;
                    case 3: nn_scan_tile<3>(S.tgt, tn, tb / ICP_CH, px, py, best, bchunk); break;
                    case 4: nn_scan_tile<4>(S.tgt, tn, tb / ICP_CH, px, py, best, bchunk); break;
                    case 5: nn_scan_tile<5>(S.tgt, tn, tb / ICP_CH, px, py, best, bchunk); break;
                    case 6: nn_scan_tile<6>(S.tgt, tn, tb / ICP_CH, px, py, best, bchunk); break;
                    case 7: nn_scan_tile<7>(S.tgt, tn, tb / ICP_CH, px, py, best, bchunk); break;
                    default: nn_scan_tile<8>(S.tgt, tn, tb / ICP_CH, px, py, best, bchunk); break;
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < ICP_PB; ++k) {
                const int i = base + k * ICP_THREADS + tid;
                if (k < np && i < ns) {
                    float d = best[k];
                    int id = -1;
                    if (bchunk[k] >= 0) { // level 2: first point of the winning chunk that attains the minimum
                        const int j0 = bchunk[k] * ICP_CH;
                        for (int jj = ICP_CH - 1; jj >= 0; --jj) {
                            const int j = j0 + jj;
                            if (j < nt) {
                                float2 t;
                                if (resident)
                                    t = S.tgt[j];
                                else {
                                    const float2 g = tgt[j];
                                    t = make_float2(f_add(g.x, -mx), f_add(g.y, -my));
                                }
                                if (dist2(px[k], py[k], t.x, t.y) == d)
                                    id = j;
                            }
                        }
                    }
                    if (id < 0 || !(d <= r2_match) || d == INFINITY) { // an infinite distance is no match, also for an unbounded matcher
                        id = -1;
                        d = INFINITY;
                    } else {
                        nfin_d[0] += 1.0;
                    }
                    nn_d2[i] = d;
                    nn_idx[i] = id;
                }
            }
        }
        block_sum<1>(nfin_d, S.red); // also orders the nn_d2 / nn_idx stores before the re-reads below
        const unsigned nfin = (unsigned)nfin_d[0];

        // ---- TrimmedDistOutlierFilter limit: exact order statistic by radix select ----
        float limit = INFINITY;
        bool fail = false;
        if (P.use_trimmed_filter) {
            if (nfin == 0) {
                fail = true; // "no outlier to filter"
                if (tid == 0)
                    S.flag_status = SFE_ICP_NO_OUTLIER;
            } else if (P.trim_ratio >= 1.0f) {
                // max of the finite distances: select rank nfin-1
                if (tid == 0)
                    S.sel_k = nfin - 1;
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
                    for (int i = tid; i < ns; i += ICP_THREADS) {
                        const float d = nn_d2[i];
                        if (d != INFINITY) {
                            const unsigned u = __float_as_uint(d); // d >= 0: bit pattern order == value order
                            if ((u & himask) == prefix)
                                atomicAdd(&S.hist[(u >> shift) & 255u], 1u);
                        }
                    }
                    __syncthreads();
                    if (tid == 0) {
                        unsigned k = S.sel_k, b = 0;
                        for (; b < 256; ++b) {
                            const unsigned h = S.hist[b];
                            if (k < h)
                                break;
                            k -= h;
                        }
                        S.sel_k = k;
                        S.sel_prefix = prefix | (b << shift);
                    }
                    __syncthreads();
                }
                limit = __uint_as_float(S.sel_prefix);
            }
        }
        __syncthreads();
        if (fail)
            break;

        // ---- error minimiser: accumulate over kept pairs ----
        double acc[10];
#pragma unroll
        for (int i = 0; i < 10; ++i)
            acc[i] = 0;
        for (int i = tid; i < ns; i += ICP_THREADS) {
            const int id = nn_idx[i];
            const float d = nn_d2[i];
            const bool ok = id >= 0 && (!P.use_max_dist_filter || d <= r2_filter) &&
                            (!P.use_trimmed_filter || d <= limit);
            if (!ok)
                continue;
            const float2 s = src[i];
            const float rx = affine1(T0[0], T0[1], T0[2], s.x, s.y);
            const float ry = affine1(T0[3], T0[4], T0[5], s.x, s.y);
            const double px = affine1(Ti[0], Ti[1], Ti[2], rx, ry);
            const double py = affine1(Ti[3], Ti[4], Ti[5], rx, ry);
            float2 q;
            if (resident)
                q = S.tgt[id];
            else {
                const float2 t = tgt[id];
                q = make_float2(f_add(t.x, -mx), f_add(t.y, -my));
            }
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
                const float2 n = nrm[id];
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

        // ---- solve, compose, check (thread 0) ----
        if (tid == 0) {
            int status, iterate;
            icp_solve_and_check(P, acc, Ti, S.Ti, chk, status, iterate);
            S.flag_status = status;
            S.flag_iterate = (status == SFE_ICP_OK) ? iterate : 0;
        }
        __syncthreads();
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
    }
}

// ---------------------------------------------------------------------------------------------
// pcl.match: exact 1-NN with a radius; one lane per query point, reference tiled through LDS
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void match_kernel(const float2 *__restrict__ ref, int nref,
                                                    const float2 *__restrict__ in, int nin, float r2,
                                                    int *__restrict__ ids, float *__restrict__ d2)
{
    __shared__ float2 s_ref[2048];
    const int i = blockIdx.x * 256 + threadIdx.x;
    float px = 0, py = 0;
    if (i < nin) {
        const float2 p = in[i];
        px = p.x;
        py = p.y;
    }
    float best = INFINITY;
    int bi = -1;
    for (int tb = 0; tb < nref; tb += 2048) {
        const int tn = min(2048, nref - tb);
        __syncthreads();
        for (int j = threadIdx.x; j < tn; j += 256)
            s_ref[j] = ref[tb + j];
        __syncthreads();
        for (int j = 0; j < tn; ++j) {
            const float2 t = s_ref[j];
            const float dx = f_add(px, -t.x), dy = f_add(py, -t.y);
            const float d = f_add(f_mul(dx, dx), f_mul(dy, dy));
            if (d < best) {
                best = d;
                bi = tb + j;
            }
        }
    }
    if (i < nin) {
        if (bi < 0 || !(best <= r2)) {
            bi = -1;
            best = INFINITY;
        }
        ids[i] = bi;
        d2[i] = best;
    }
}

// pcl.match with knn > 1 (pcl.cpp:161-174 takes knn): the knn nearest reference points of every query in ascending
// (d2, index) order, -1 / inf where fewer than knn lie within the radius.  One lane per query; neighbour j is the
// lexicographic minimum of (d2, index) above neighbour j-1: knn passes over the reference (tiled through LDS) --
// exact for any knn without a per-lane list; knn is small (1 in the SLAM node).
__global__ __launch_bounds__(256) void match_knn_kernel(const float2 *__restrict__ ref, int nref,
                                                        const float2 *__restrict__ in, int nin, int knn, float r2,
                                                        int *__restrict__ ids, float *__restrict__ d2)
{
    __shared__ float2 s_ref[2048];
    const int i = blockIdx.x * 256 + threadIdx.x;
    float px = 0, py = 0;
    if (i < nin) {
        const float2 p = in[i];
        px = p.x;
        py = p.y;
    }
    float prev_d = -1.0f; // every distance is >= 0
    int prev_i = -1;
    for (int jn = 0; jn < knn; ++jn) {
        float best = INFINITY;
        int bi = -1;
        for (int tb = 0; tb < nref; tb += 2048) {
            const int tn = min(2048, nref - tb);
            __syncthreads();
            for (int j = threadIdx.x; j < tn; j += 256)
                s_ref[j] = ref[tb + j];
            __syncthreads();
            for (int j = 0; j < tn; ++j) {
                const float2 t = s_ref[j];
                const float dx = f_add(px, -t.x), dy = f_add(py, -t.y);
                const float d = f_add(f_mul(dx, dx), f_mul(dy, dy));
                const bool after = d > prev_d || (d == prev_d && tb + j > prev_i); // NaN: never
                if (after && d < best) { // ascending index: the first point attaining the minimum wins
                    best = d;
                    bi = tb + j;
                }
            }
        }
        if (bi < 0 || !(best <= r2)) { // nothing left within the radius: this and all further neighbours are missing
            bi = -1;
            best = INFINITY;
        }
        if (i < nin) {
            ids[(size_t)jn * nin + i] = bi;
            d2[(size_t)jn * nin + i] = best;
        }
        prev_d = bi < 0 ? INFINITY : best;
        prev_i = bi;
    }
}

// Densities of libpointmatcher's SurfaceNormalDataPointsFilter{keepDensities} (pcl.cpp:81-88): per point, the knn
// nearest points incl. itself (ids from match_knn_kernel), their float mean, r = the largest distance of a
// neighbour from that mean, density = knn / ((4/3) pi r^3) with the volume computed in double and rounded to float.
__global__ __launch_bounds__(256) void knn_density_kernel(const float2 *__restrict__ pts, int n, int knn,
                                                          const int *__restrict__ ids, float *__restrict__ dens)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n)
        return;
    float sx = 0.0f, sy = 0.0f;
    int real = 0;
    for (int j = 0; j < knn; ++j) {
        const int id = ids[(size_t)j * n + i];
        if (id >= 0) {
            sx = f_add(sx, pts[id].x);
            sy = f_add(sy, pts[id].y);
            ++real;
        }
    }
    const float mx = __fdiv_rn(sx, (float)real), my = __fdiv_rn(sy, (float)real);
    float rmax = 0.0f;
    for (int j = 0; j < knn; ++j) {
        const int id = ids[(size_t)j * n + i];
        if (id >= 0) {
            const float dx = f_add(pts[id].x, -mx), dy = f_add(pts[id].y, -my);
            rmax = fmaxf(rmax, sqrtf(f_add(f_mul(dx, dx), f_mul(dy, dy))));
        }
    }
    const double r = (double)rmax;
    const float volume = (float)((4. / 3.) * 3.14159265358979323846 * (r * r * r));
    dens[i] = __fdiv_rn((float)real, volume);
}

// pcl.remove_outlier: count points within radius (incl. self); keep iff count > min_points
__global__ __launch_bounds__(256) void radius_count_kernel(const float2 *__restrict__ pts, int n, float r2,
                                                           int min_points, int *__restrict__ keep)
{
    __shared__ float2 s_p[2048];
    const int i = blockIdx.x * 256 + threadIdx.x;
    float px = 0, py = 0;
    if (i < n) {
        const float2 p = pts[i];
        px = p.x;
        py = p.y;
    }
    int cnt = 0;
    for (int tb = 0; tb < n; tb += 2048) {
        const int tn = min(2048, n - tb);
        __syncthreads();
        for (int j = threadIdx.x; j < tn; j += 256)
            s_p[j] = pts[tb + j];
        __syncthreads();
        for (int j = 0; j < tn; ++j) {
            const float2 t = s_p[j];
            const float dx = f_add(px, -t.x), dy = f_add(py, -t.y);
            cnt += f_add(f_mul(dx, dx), f_mul(dy, dy)) <= r2;
        }
    }
    if (i < n)
        keep[i] = cnt > min_points;
}

// ---------------------------------------------------------------------------------------------
static int check_params(sfe_ctx *ctx, const sfe_icp_params *p)
{
    SFE_ARG(ctx, p != nullptr);
    SFE_ARG(ctx, p->minimizer == 0 || p->minimizer == 1);
    SFE_ARG(ctx, p->max_iter >= 1);
    SFE_ARG(ctx, !p->use_diff_checker || (p->smooth_len >= 1 && p->smooth_len < ICP_MAX_HIST));
    SFE_ARG(ctx, !p->use_trimmed_filter || (p->trim_ratio >= 0.0f && p->trim_ratio <= 1.0f));
    SFE_ARG(ctx, p->minimizer == 0 || (p->normals_knn >= 2 && p->normals_knn <= ICP_KMAX));
    SFE_ARG(ctx, p->matcher_max_dist > 0.0f);
    return 0;
}

// jobs4: host array n_jobs x 4 = (src_start, n_src, tgt_start, n_tgt) in points
static int icp_launch(sfe_ctx *ctx, const sfe_icp_params *p, const float *d_src, const float *d_tgt,
                      const int32_t *jobs4, const float *d_guess9, int n_jobs, float *d_T9, int32_t *d_status,
                      int32_t *d_iters)
{
    if (int rc = check_params(ctx, p))
        return rc;
    std::vector<IcpJob> jobs((size_t)n_jobs);
    long long soff = 0, noff = 0;
    for (int j = 0; j < n_jobs; ++j) {
        const int32_t *q = jobs4 + 4 * (size_t)j;
        SFE_ARG(ctx, q[0] >= 0 && q[1] >= 0 && q[2] >= 0 && q[3] >= 0);
        if (q[1] == 0 || q[3] == 0)
            return sfe_set_err(ctx, SFE_ERR_ARG, "ICP job %d has an empty cloud (n_src=%d, n_tgt=%d)", j, q[1], q[3]);
        jobs[j] = {q[0], q[1], q[2], q[3], soff, noff};
        soff += q[1];
        noff += q[3];
    }
    if (!(ctx->icp_variant & 4)) { // default: strip-sweep search (sfe_icp_sweep.hip), same results
        const int rc = sfe_icp_sweep_launch(ctx, p, d_src, d_tgt, jobs4, d_guess9, n_jobs, d_T9, d_status, d_iters);
        if (rc != 1)
            return rc;
    }
    IcpJob *d_jobs = (IcpJob *)sfe_scratch(ctx, 4, sizeof(IcpJob) * (size_t)n_jobs);
    float *d_nn_d2 = (float *)sfe_scratch(ctx, 5, sizeof(float) * (size_t)soff);
    int *d_nn_idx = (int *)sfe_scratch(ctx, 6, sizeof(int) * (size_t)soff);
    float2 *d_nrm = p->minimizer == 1 ? (float2 *)sfe_scratch(ctx, 7, sizeof(float2) * (size_t)noff) : nullptr;
    if (!d_jobs || !d_nn_d2 || !d_nn_idx || (p->minimizer == 1 && !d_nrm))
        return SFE_ERR_HIP;
    { // pinned staging: no stream synchronisation on an enqueue-only path
        void *h = sfe_pinned_begin(ctx, sizeof(IcpJob) * (size_t)n_jobs);
        if (!h)
            return SFE_ERR_HIP;
        memcpy(h, jobs.data(), sizeof(IcpJob) * (size_t)n_jobs);
        SFE_HIP(ctx, hipMemcpyAsync(d_jobs, h, sizeof(IcpJob) * (size_t)n_jobs, hipMemcpyHostToDevice, ctx->stream));
        if (int rc = sfe_pinned_end(ctx, ctx->stream))
            return rc;
    }
    // occupancy A/B: bit 1 of the tuning variant selects the 128-VGPR build (1 workgroup per CU)
    const int nnv = ctx->icp_variant & 1;
    if (!(ctx->icp_variant & 2)) { // default: 64-VGPR build, two workgroups per CU (measured 7 % faster)
        SFE_HIP(ctx, hipFuncSetAttribute((const void *)icp_job_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)sizeof(IcpShared)));
        hipLaunchKernelGGL(icp_job_kernel<8>, dim3(n_jobs), dim3(ICP_THREADS), sizeof(IcpShared), ctx->stream, *p, nnv,
                           d_jobs, (const float2 *)d_src, (const float2 *)d_tgt, d_guess9, d_nn_d2, d_nn_idx, d_nrm,
                           d_T9, d_status, d_iters);
    } else {
        SFE_HIP(ctx, hipFuncSetAttribute((const void *)icp_job_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)sizeof(IcpShared)));
        hipLaunchKernelGGL(icp_job_kernel<4>, dim3(n_jobs), dim3(ICP_THREADS), sizeof(IcpShared), ctx->stream, *p, nnv,
                           d_jobs, (const float2 *)d_src, (const float2 *)d_tgt, d_guess9, d_nn_d2, d_nn_idx, d_nrm,
                           d_T9, d_status, d_iters);
    }
    SFE_LAUNCH_CHECK(ctx);
    return 0;
}

// A job that was shared by several workgroups reports SFE_ICP_SPLIT_TIMEOUT when its shares were not resident
// together (the device is shared with another context or process; sfe_icp_sweep.hip "split jobs").  The host-pointer
// entry points see the statuses and run the call once more with splitting off, so the caller never meets status 6.
static bool icp_any_split_timeout(const int32_t *status, int n)
{
    for (int j = 0; j < n; ++j)
        if (status[j] == SFE_ICP_SPLIT_TIMEOUT)
            return true;
    return false;
}
#define ICP_RETRY_UNSPLIT(call)                                                                  \
    do {                                                                                         \
        if (!(ctx->icp_variant & 16) && icp_any_split_timeout(status, n_retry_)) {               \
            ctx->icp_variant |= 16;                                                              \
            const int rc_ = (call);                                                              \
            ctx->icp_variant &= ~16;                                                             \
            return rc_;                                                                          \
        }                                                                                        \
    } while (0)

extern "C" {

int sfe_icp_set_tuning(sfe_ctx *ctx, int variant)
{
    if (!ctx)
        return SFE_ERR_ARG;
    SFE_ARG(ctx, variant >= 0 && variant <= 31);
    ctx->icp_variant = variant;
    return 0;
}

int sfe_icp_batch_dev(sfe_ctx *ctx, const sfe_icp_params *p, const float *d_src, const int32_t *src_off,
                      const float *d_tgt, const int32_t *tgt_off, const float *d_guess9, int n_jobs, float *d_T9,
                      int32_t *d_status, int32_t *d_iters)
{
    if (int rc = sfe_use(ctx))
        return rc;
    SFE_ARG(ctx, d_src && d_tgt && src_off && tgt_off && d_guess9 && d_T9 && d_status && d_iters && n_jobs >= 0);
    if (n_jobs == 0)
        return 0;
    std::vector<int32_t> jobs4(4 * (size_t)n_jobs);
    for (int j = 0; j < n_jobs; ++j) {
        jobs4[4 * j] = src_off[j];
        jobs4[4 * j + 1] = src_off[j + 1] - src_off[j];
        jobs4[4 * j + 2] = tgt_off[j];
        jobs4[4 * j + 3] = tgt_off[j + 1] - tgt_off[j];
    }
    return icp_launch(ctx, p, d_src, d_tgt, jobs4.data(), d_guess9, n_jobs, d_T9, d_status, d_iters);
}

int sfe_icp_jobs_dev(sfe_ctx *ctx, const sfe_icp_params *p, const float *d_src, const float *d_tgt, const int32_t *jobs4,
                     const float *d_guess9, int n_jobs, float *d_T9, int32_t *d_status, int32_t *d_iters)
{
    if (int rc = sfe_use(ctx))
        return rc;
    SFE_ARG(ctx, d_src && d_tgt && jobs4 && d_guess9 && d_T9 && d_status && d_iters && n_jobs >= 0);
    if (n_jobs == 0)
        return 0;
    return icp_launch(ctx, p, d_src, d_tgt, jobs4, d_guess9, n_jobs, d_T9, d_status, d_iters);
}

int sfe_icp_compute_guesses(sfe_ctx *ctx, const sfe_icp_params *p, const float *src, int n_src, const float *tgt,
                            int n_tgt, const float *guesses9, int n_guesses, float *T_out9, int32_t *status,
                            int32_t *iters)
{
    if (int rc = sfe_use(ctx))
        return rc;
    SFE_ARG(ctx, src && tgt && guesses9 && T_out9 && status && n_src >= 0 && n_tgt >= 0 && n_guesses >= 0);
    if (n_guesses == 0)
        return 0;
    if (n_src == 0 || n_tgt == 0)
        return sfe_set_err(ctx, SFE_ERR_ARG, "ICP needs non-empty clouds (n_src=%d, n_tgt=%d)", n_src, n_tgt);
    // one pinned block up ([source | target | guesses]), one down ([T | status | iterations]): the live node calls this
    // once per scan match with clouds of 10^2..10^3 points, where the copies and their latencies are the cost
    const size_t b_src = sizeof(float) * 2 * (size_t)n_src, b_tgt = sizeof(float) * 2 * (size_t)n_tgt;
    const size_t b_g = sizeof(float) * 9 * (size_t)n_guesses, b_in = b_src + b_tgt + b_g;
    const size_t b_out = (sizeof(float) * 9 + 2 * sizeof(int32_t)) * (size_t)n_guesses;
    char *d_in = (char *)sfe_scratch(ctx, 0, b_in);
    char *d_out = (char *)sfe_scratch(ctx, 3, b_out);
    char *h_in = (char *)sfe_pinned_io(ctx, 2, b_in);
    char *h_out = (char *)sfe_pinned_io(ctx, 3, b_out);
    if (!d_in || !d_out || !h_in || !h_out)
        return SFE_ERR_HIP;
    memcpy(h_in, src, b_src);
    memcpy(h_in + b_src, tgt, b_tgt);
    memcpy(h_in + b_src + b_tgt, guesses9, b_g);
    SFE_HIP(ctx, hipMemcpyAsync(d_in, h_in, b_in, hipMemcpyHostToDevice, ctx->stream));
    float *d_src = (float *)d_in, *d_tgt = (float *)(d_in + b_src), *d_g = (float *)(d_in + b_src + b_tgt);
    float *d_T = (float *)d_out;
    int32_t *d_st = (int32_t *)(d_out + sizeof(float) * 9 * (size_t)n_guesses);
    std::vector<int32_t> jobs4(4 * (size_t)n_guesses);
    for (int j = 0; j < n_guesses; ++j) {
        jobs4[4 * j] = 0;
        jobs4[4 * j + 1] = n_src;
        jobs4[4 * j + 2] = 0;
        jobs4[4 * j + 3] = n_tgt;
    }
    if (int rc = icp_launch(ctx, p, d_src, d_tgt, jobs4.data(), d_g, n_guesses, d_T, d_st, d_st + n_guesses))
        return rc;
    SFE_HIP(ctx, hipMemcpyAsync(h_out, d_out, b_out, hipMemcpyDeviceToHost, ctx->stream));
    SFE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(T_out9, h_out, sizeof(float) * 9 * (size_t)n_guesses);
    memcpy(status, h_out + sizeof(float) * 9 * (size_t)n_guesses, sizeof(int32_t) * (size_t)n_guesses);
    if (iters)
        memcpy(iters, h_out + (sizeof(float) * 9 + sizeof(int32_t)) * (size_t)n_guesses, sizeof(int32_t) * (size_t)n_guesses);
    const int n_retry_ = n_guesses;
    ICP_RETRY_UNSPLIT(sfe_icp_compute_guesses(ctx, p, src, n_src, tgt, n_tgt, guesses9, n_guesses, T_out9, status, iters));
    return 0;
}

int sfe_icp_compute_pairs(sfe_ctx *ctx, const sfe_icp_params *p, const float *src, const int32_t *src_off,
                          const float *tgt, const int32_t *tgt_off, const float *guesses9, int n_jobs, float *T_out9,
                          int32_t *status, int32_t *iters)
{
    if (int rc = sfe_use(ctx))
        return rc;
    SFE_ARG(ctx, n_jobs >= 0 && (n_jobs == 0 || (src && tgt && src_off && tgt_off && guesses9 && T_out9 && status)));
    if (n_jobs == 0)
        return 0;
    const size_t ns = (size_t)src_off[n_jobs], nt = (size_t)tgt_off[n_jobs];
    float *d_src = (float *)sfe_scratch(ctx, 0, sizeof(float) * 2 * std::max<size_t>(ns, 1));
    float *d_tgt = (float *)sfe_scratch(ctx, 1, sizeof(float) * 2 * std::max<size_t>(nt, 1));
    float *d_g = (float *)sfe_scratch(ctx, 2, sizeof(float) * 9 * (size_t)n_jobs);
    float *d_T = (float *)sfe_scratch(ctx, 3, sizeof(float) * 9 * (size_t)n_jobs);
    int32_t *d_st = (int32_t *)sfe_scratch(ctx, 8, sizeof(int32_t) * 2 * (size_t)n_jobs);
    if (!d_src || !d_tgt || !d_g || !d_T || !d_st)
        return SFE_ERR_HIP;
    SFE_HIP(ctx, hipMemcpyAsync(d_src, src, sizeof(float) * 2 * ns, hipMemcpyHostToDevice, ctx->stream));
    SFE_HIP(ctx, hipMemcpyAsync(d_tgt, tgt, sizeof(float) * 2 * nt, hipMemcpyHostToDevice, ctx->stream));
    SFE_HIP(ctx, hipMemcpyAsync(d_g, guesses9, sizeof(float) * 9 * (size_t)n_jobs, hipMemcpyHostToDevice, ctx->stream));
    if (int rc = sfe_icp_batch_dev(ctx, p, d_src, src_off, d_tgt, tgt_off, d_g, n_jobs, d_T, d_st, d_st + n_jobs))
        return rc;
    SFE_HIP(ctx, hipMemcpyAsync(T_out9, d_T, sizeof(float) * 9 * (size_t)n_jobs, hipMemcpyDeviceToHost, ctx->stream));
    SFE_HIP(ctx, hipMemcpyAsync(status, d_st, sizeof(int32_t) * (size_t)n_jobs, hipMemcpyDeviceToHost, ctx->stream));
    if (iters)
        SFE_HIP(ctx, hipMemcpyAsync(iters, d_st + n_jobs, sizeof(int32_t) * (size_t)n_jobs, hipMemcpyDeviceToHost,
                                    ctx->stream));
    SFE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const int n_retry_ = n_jobs;
    ICP_RETRY_UNSPLIT(sfe_icp_compute_pairs(ctx, p, src, src_off, tgt, tgt_off, guesses9, n_jobs, T_out9, status, iters));
    return 0;
}

int sfe_icp_compute_jobs(sfe_ctx *ctx, const sfe_icp_params *p, const float *src, int n_src_pts, const float *tgt,
                         int n_tgt_pts, const int32_t *jobs4, const float *guesses9, int n_jobs, float *T_out9,
                         int32_t *status, int32_t *iters)
{
    if (int rc = sfe_use(ctx))
        return rc;
    SFE_ARG(ctx, n_jobs >= 0 && n_src_pts >= 0 && n_tgt_pts >= 0 &&
                     (n_jobs == 0 || (src && tgt && jobs4 && guesses9 && T_out9 && status)));
    if (n_jobs == 0)
        return 0;
    for (int j = 0; j < n_jobs; ++j) {
        const int32_t *q = jobs4 + 4 * (size_t)j;
        if (q[0] < 0 || q[1] < 0 || q[2] < 0 || q[3] < 0 || (long long)q[0] + q[1] > n_src_pts ||
            (long long)q[2] + q[3] > n_tgt_pts)
            return sfe_set_err(ctx, SFE_ERR_ARG, "ICP job %d (%d+%d, %d+%d) lies outside the clouds (%d, %d points)", j,
                               q[0], q[1], q[2], q[3], n_src_pts, n_tgt_pts);
    }
    float *d_src = (float *)sfe_scratch(ctx, 0, sizeof(float) * 2 * (size_t)std::max(n_src_pts, 1));
    float *d_tgt = (float *)sfe_scratch(ctx, 1, sizeof(float) * 2 * (size_t)std::max(n_tgt_pts, 1));
    float *d_g = (float *)sfe_scratch(ctx, 2, sizeof(float) * 9 * (size_t)n_jobs);
    float *d_T = (float *)sfe_scratch(ctx, 3, sizeof(float) * 9 * (size_t)n_jobs);
    int32_t *d_st = (int32_t *)sfe_scratch(ctx, 8, sizeof(int32_t) * 2 * (size_t)n_jobs);
    if (!d_src || !d_tgt || !d_g || !d_T || !d_st)
        return SFE_ERR_HIP;
    SFE_HIP(ctx, hipMemcpyAsync(d_src, src, sizeof(float) * 2 * (size_t)n_src_pts, hipMemcpyHostToDevice, ctx->stream));
    SFE_HIP(ctx, hipMemcpyAsync(d_tgt, tgt, sizeof(float) * 2 * (size_t)n_tgt_pts, hipMemcpyHostToDevice, ctx->stream));
    SFE_HIP(ctx, hipMemcpyAsync(d_g, guesses9, sizeof(float) * 9 * (size_t)n_jobs, hipMemcpyHostToDevice, ctx->stream));
    if (int rc = icp_launch(ctx, p, d_src, d_tgt, jobs4, d_g, n_jobs, d_T, d_st, d_st + n_jobs))
        return rc;
    SFE_HIP(ctx, hipMemcpyAsync(T_out9, d_T, sizeof(float) * 9 * (size_t)n_jobs, hipMemcpyDeviceToHost, ctx->stream));
    SFE_HIP(ctx, hipMemcpyAsync(status, d_st, sizeof(int32_t) * (size_t)n_jobs, hipMemcpyDeviceToHost, ctx->stream));
    if (iters)
        SFE_HIP(ctx, hipMemcpyAsync(iters, d_st + n_jobs, sizeof(int32_t) * (size_t)n_jobs, hipMemcpyDeviceToHost,
                                    ctx->stream));
    SFE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const int n_retry_ = n_jobs;
    ICP_RETRY_UNSPLIT(sfe_icp_compute_jobs(ctx, p, src, n_src_pts, tgt, n_tgt_pts, jobs4, guesses9, n_jobs, T_out9, status,
                                           iters));
    return 0;
}

int sfe_icp_compute(sfe_ctx *ctx, const sfe_icp_params *p, const float *src, int n_src, const float *tgt, int n_tgt,
                    const float *guess9, float *T_out9, int *iters)
{
    int32_t st = 0, it = 0;
    const int rc = sfe_icp_compute_guesses(ctx, p, src, n_src, tgt, n_tgt, guess9, 1, T_out9, &st, &it);
    if (rc)
        return rc;
    if (iters)
        *iters = it;
    return st;
}

int sfe_match(sfe_ctx *ctx, const float *ref, int n_ref, const float *in, int n_in, float max_dist, int32_t *ids,
              float *d2)
{
    if (int rc = sfe_use(ctx))
        return rc;
    SFE_ARG(ctx, n_ref >= 0 && n_in >= 0 && (n_in == 0 || (in && ids && d2)) && (n_ref == 0 || ref));
    if (n_in == 0)
        return 0;
    float *d_ref = (float *)sfe_scratch(ctx, 0, sizeof(float) * 2 * (size_t)std::max(n_ref, 1));
    float *d_in = (float *)sfe_scratch(ctx, 1, sizeof(float) * 2 * (size_t)n_in);
    int *d_ids = (int *)sfe_scratch(ctx, 2, sizeof(int) * (size_t)n_in);
    float *d_d2 = (float *)sfe_scratch(ctx, 3, sizeof(float) * (size_t)n_in);
    if (!d_ref || !d_in || !d_ids || !d_d2)
        return SFE_ERR_HIP;
    if (n_ref)
        SFE_HIP(ctx, hipMemcpyAsync(d_ref, ref, sizeof(float) * 2 * (size_t)n_ref, hipMemcpyHostToDevice, ctx->stream));
    SFE_HIP(ctx, hipMemcpyAsync(d_in, in, sizeof(float) * 2 * (size_t)n_in, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(match_kernel, dim3((n_in + 255) / 256), dim3(256), 0, ctx->stream, (const float2 *)d_ref, n_ref,
                       (const float2 *)d_in, n_in, max_dist * max_dist, d_ids, d_d2);
    SFE_LAUNCH_CHECK(ctx);
    SFE_HIP(ctx, hipMemcpyAsync(ids, d_ids, sizeof(int) * (size_t)n_in, hipMemcpyDeviceToHost, ctx->stream));
    SFE_HIP(ctx, hipMemcpyAsync(d2, d_d2, sizeof(float) * (size_t)n_in, hipMemcpyDeviceToHost, ctx->stream));
    SFE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int sfe_match_knn(sfe_ctx *ctx, const float *ref, int n_ref, const float *in, int n_in, int knn, float max_dist,
                  int32_t *ids, float *d2)
{
    if (int rc = sfe_use(ctx))
        return rc;
    SFE_ARG(ctx, n_ref >= 0 && n_in >= 0 && knn >= 1 && (n_in == 0 || (in && ids && d2)) && (n_ref == 0 || ref));
    if (n_in == 0)
        return 0;
    float *d_ref = (float *)sfe_scratch(ctx, 0, sizeof(float) * 2 * (size_t)std::max(n_ref, 1));
    float *d_in = (float *)sfe_scratch(ctx, 1, sizeof(float) * 2 * (size_t)n_in);
    int *d_ids = (int *)sfe_scratch(ctx, 2, sizeof(int) * (size_t)n_in * knn);
    float *d_d2 = (float *)sfe_scratch(ctx, 3, sizeof(float) * (size_t)n_in * knn);
    if (!d_ref || !d_in || !d_ids || !d_d2)
        return SFE_ERR_HIP;
    if (n_ref)
        SFE_HIP(ctx, hipMemcpyAsync(d_ref, ref, sizeof(float) * 2 * (size_t)n_ref, hipMemcpyHostToDevice, ctx->stream));
    SFE_HIP(ctx, hipMemcpyAsync(d_in, in, sizeof(float) * 2 * (size_t)n_in, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(match_knn_kernel, dim3((n_in + 255) / 256), dim3(256), 0, ctx->stream, (const float2 *)d_ref, n_ref,
                       (const float2 *)d_in, n_in, knn, max_dist * max_dist, d_ids, d_d2);
    SFE_LAUNCH_CHECK(ctx);
    SFE_HIP(ctx, hipMemcpyAsync(ids, d_ids, sizeof(int) * (size_t)n_in * knn, hipMemcpyDeviceToHost, ctx->stream));
    SFE_HIP(ctx, hipMemcpyAsync(d2, d_d2, sizeof(float) * (size_t)n_in * knn, hipMemcpyDeviceToHost, ctx->stream));
    SFE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int sfe_knn_density(sfe_ctx *ctx, const float *pts, int n, int knn, float *dens_out)
{
    if (int rc = sfe_use(ctx))
        return rc;
    SFE_ARG(ctx, n >= 0 && knn >= 1 && (n == 0 || (pts && dens_out)));
    if (n == 0)
        return 0;
    if (knn > n) // libnabo: "Requesting more points than available in cloud"
        return sfe_set_err(ctx, SFE_ERR_ARG, "knn density: %d neighbours requested from a cloud of %d points", knn, n);
    float *d_pts = (float *)sfe_scratch(ctx, 0, sizeof(float) * 2 * (size_t)n);
    int *d_ids = (int *)sfe_scratch(ctx, 2, sizeof(int) * (size_t)n * knn);
    float *d_d2 = (float *)sfe_scratch(ctx, 3, sizeof(float) * (size_t)n * knn);
    float *d_dens = (float *)sfe_scratch(ctx, 1, sizeof(float) * (size_t)n);
    if (!d_pts || !d_ids || !d_d2 || !d_dens)
        return SFE_ERR_HIP;
    SFE_HIP(ctx, hipMemcpyAsync(d_pts, pts, sizeof(float) * 2 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(match_knn_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, (const float2 *)d_pts, n,
                       (const float2 *)d_pts, n, knn, INFINITY, d_ids, d_d2);
    hipLaunchKernelGGL(knn_density_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, (const float2 *)d_pts, n, knn,
                       d_ids, d_dens);
    SFE_LAUNCH_CHECK(ctx);
    SFE_HIP(ctx, hipMemcpyAsync(dens_out, d_dens, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    SFE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int sfe_remove_outlier(sfe_ctx *ctx, const float *pts, int n, double radius, int min_points, float *out, int *n_out)
{
    if (int rc = sfe_use(ctx))
        return rc;
    SFE_ARG(ctx, n >= 0 && n_out && (n == 0 || (pts && out)));
    *n_out = 0;
    if (n == 0)
        return 0;
    float *d_pts = (float *)sfe_scratch(ctx, 0, sizeof(float) * 2 * (size_t)n);
    int *d_keep = (int *)sfe_scratch(ctx, 2, sizeof(int) * (size_t)n);
    if (!d_pts || !d_keep)
        return SFE_ERR_HIP;
    SFE_HIP(ctx, hipMemcpyAsync(d_pts, pts, sizeof(float) * 2 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(radius_count_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, (const float2 *)d_pts, n,
                       (float)(radius * radius), min_points, d_keep);
    SFE_LAUNCH_CHECK(ctx);
    std::vector<int> keep((size_t)n);
    SFE_HIP(ctx, hipMemcpyAsync(keep.data(), d_keep, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    SFE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    int m = 0;
    for (int i = 0; i < n; ++i) // order-preserving gather of the kept rows (the decision is the GPU's)
        if (keep[i]) {
            out[2 * m] = pts[2 * i];
            out[2 * m + 1] = pts[2 * i + 1];
            ++m;
        }
    *n_out = m;
    return 0;
}

} // extern "C"
