// Strip-sweep ICP, target preparation: mean, centring, strip table, sort, witness grid, PCA normals (one workgroup per
// distinct target; many guesses on one pair share it).  Shared definitions and the overview: sfe_icp_sweep.h.
#include "sfe_icp_sweep.h"

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

template <int NT, int TCAP, int GM, bool GTAIL, int KMF>
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

// the builds the host side asks for (sfe_icp_sweep.hip: job classes t0 / t1 / 1024 threads; the 1024-thread kernel with its
// witness grid in LDS of its own, or inside the key buffer at neighbour-list capacities 8 / 10)
#define SW_PREP_INST(NT, TCAP, GM, GTAIL, KMF)                                                                                   \
    template int sweep_launch_prep<NT, TCAP, GM, GTAIL, KMF>(sfe_ctx *, hipStream_t, const sfe_icp_params *, int, const SweepPrep *, \
                                                             const int *, const float2 *, float2 *, int *, float2 *, float *,     \
                                                             unsigned long long *, StripTab *, int *);
SW_PREP_INST(SW_T0_NT, SW_T0_TCAP, SW_T0_GRID, false, 0)
SW_PREP_INST(SW_T1_NT, SW_T1_TCAP, SW_T1_GRID, false, 0)
SW_PREP_INST(ICP_THREADS, SW_TCAP, SW_GRID_MAX, false, 0)
SW_PREP_INST(ICP_THREADS, SW_TCAP, SW_GRID_MAX, true, 8)
SW_PREP_INST(ICP_THREADS, SW_TCAP, SW_GRID_MAX, true, 10)

int sweep_launch_normals(sfe_ctx *ctx, hipStream_t ps, const sfe_icp_params *p, int n, int per, const SweepPrep *d_preps,
                         const int *d_pids, const float2 *d_stgt, const int *d_perm, float2 *d_snrm, const StripTab *d_tab)
{
    hipLaunchKernelGGL(icp_sweep_normals_kernel<ICP_THREADS>, dim3((unsigned)n, (unsigned)per), dim3(ICP_THREADS), 0, ps, *p, d_preps,
                       d_pids, d_stgt, d_perm, d_snrm, d_tab);
    SFE_LAUNCH_CHECK(ctx);
    return 0;
}
