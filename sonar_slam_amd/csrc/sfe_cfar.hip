// CFAR detectors on gfx950.  Replaces bruce_slam/src/bruce_slam/cpp/cfar.cpp:10-192.
//
// Three kernels:
//   cfar_u8_ring<T,G,ALG>  the hot path (uint8 sonar image, CA/SOCA/GOCA, shipped window):
//                          HBM-bound streaming kernel, 1 B read + 1 B written per pixel.
//   cfar_u8_generic        any window / any variant incl. OS and the *2 threshold maps.
//   cfar_f32_naive         float images: the reference's float sums in the reference's order.
//
// Exactness of the uint8 paths: inputs are integers 0..255, so the reference's
// float window sums (<= 2*train_hs*255 < 2^24) are exact integers whatever the order; we sum
// in integers.  The decision `(double)x > tau*s/train_hs` (cfar.cpp:47) is a function of the
// two integers (x, s) only and monotone in s, so the ring kernel looks up, per pixel value x,
// the number of window sums that pass: lut[x] = #{s : x > f(s)}; pixel fires iff s < lut[x].
// The table is built on the host with exactly the reference's double expression.
#include "sfe_internal.h"

#include <algorithm>
#include <cmath>

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

struct CfarLut {
    uint16_t v[256];
};

__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, a) + __builtin_bit_cast(u16x2, b));
}
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, a) - __builtin_bit_cast(u16x2, b));
}
__device__ __forceinline__ uint32_t pk_min(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(
        uint32_t, __builtin_elementwise_min(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(
        uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ uint32_t pk_sub_sat(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(u16x2, a),
                                                                      __builtin_bit_cast(u16x2, b)));
}
// bytes (b0,b1) / (b2,b3) of x widened to two u16 lanes
__device__ __forceinline__ uint32_t unpack_lo(uint32_t x) { return __builtin_amdgcn_perm(0u, x, 0x0c010c00u); }
__device__ __forceinline__ uint32_t unpack_hi(uint32_t x) { return __builtin_amdgcn_perm(0u, x, 0x0c030c02u); }

// ---------------------------------------------------------------------------------------------
// Register-ring kernel.  One lane owns 4 adjacent beams (one 32-bit load per range row, a wave
// reads 256 contiguous bytes per row) and marches down the range axis.  It keeps the running
// column prefix sums P[i] = sum_{j<i} x[j] of the last R = 2(T+G)+2 rows as packed u16 pairs in
// a register ring (mod-2^16 arithmetic is exact for window sums <= 10200), so
//   lead(r) = P[r-G] - P[r-T-G],  lag(r) = P[r+T+G+1] - P[r+G+1],  x(r) = P[r+1] - P[r]
// cost one packed subtract each, every input byte is loaded exactly once per tile, and the
// window never touches memory again.  LDS holds only the 256-entry decision table.
// ---------------------------------------------------------------------------------------------
// Same as pk_sub but opaque to the optimiser.  lead(r) and lag(r-31) are the same difference of
// ring entries; left to itself the compiler keeps 31 rows of lag values alive to reuse them as
// lead (62 VGPRs, which spills the ring).  Recomputing costs one op per pixel pair.
__device__ __forceinline__ uint32_t pk_sub_opaque(uint32_t a, uint32_t b)
{
    uint32_t d;
    asm("v_pk_sub_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

typedef __amdgpu_buffer_rsrc_t sfe_rsrc_t; // 128-bit buffer resource (SGPRs)

// Tiles and column chunks OVERLAP instead of being predicated: a tile is always a whole number
// of R-row groups (the last tile is shifted up so it ends at the last row) and the last 64-lane
// chunk is shifted left so it ends at the last beam.  Overlapped outputs are recomputed with
// identical values, so the body has no branches, no exec masking and no tail code.
template <int T, int G, int ALG, int D>
__global__ __launch_bounds__(256, 3) void cfar_u8_ring(const uint8_t *__restrict__ img,
                                                    uint8_t *__restrict__ mask, int rows, int cols,
                                                    int n_frames, int groups_per_tile,
                                                    int tiles_per_frame, int chunks_per_row, CfarLut lut)
{
    constexpr int H = T + G;
    constexpr int R = 2 * H + 2;
    static_assert(R % D == 0, "prefetch depth must divide the ring length");

    __shared__ uint16_t s_lut[256];
    s_lut[threadIdx.x] = lut.v[threadIdx.x];
    __syncthreads();

    // wave-uniform work item: (frame f, row tile t, 64-lane column chunk); lane -> 4 beams
    const int lane = threadIdx.x & 63;
    const int wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // XCD-aware block -> tile map.  Workgroup b runs on XCD b % 8 and every XCD has its own L2: all
    // tiles of one frame go to ONE XCD (frame f -> XCD f % 8, its workgroups consecutive in that
    // XCD's dispatch order), so the 2*(T+G) halo rows a tile shares with its neighbours are L2 hits
    // instead of a second trip over the fabric (FETCH_SIZE 1.92x -> ~1.0x of the image bytes).
    const int wpf = tiles_per_frame * chunks_per_row; // waves per frame
    const int bpf = (wpf + 3) >> 2;                    // workgroups per frame
    const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    const int fg = k / bpf;
    const long long f = (long long)fg * 8 + xcd;
    const int wvf = (k - fg * bpf) * 4 + wave_in_block;
    if (f >= n_frames || wvf >= wpf)
        return;
    const int chunk = wvf % chunks_per_row;
    const int t = wvf / chunks_per_row;
    const int lpr = cols >> 2;                       // 4-beam lanes per row (>= 64)
    const int cx0 = min(chunk * 64, lpr - 64);       // last chunk shifted left
    const uint32_t voff = (uint32_t)(cx0 + lane) * 4u;
    const int tile_rows = groups_per_tile * R;       // <= rows
    const int r0 = min(t * tile_rows, rows - tile_rows); // last tile shifted up

    const size_t frame_bytes = (size_t)rows * cols;
    const sfe_rsrc_t src = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(img + (size_t)f * frame_bytes),
                                                             0, (int)frame_bytes, 0x00020000);
    const sfe_rsrc_t dst = __builtin_amdgcn_make_buffer_rsrc(mask + (size_t)f * frame_bytes, 0,
                                                             (int)frame_bytes, 0x00020000);

    // Rows outside the image are clamped instead of zero-filled: a window that touches them
    // belongs to a border row whose output is forced to 0 anyway, and prefix differences of
    // in-image windows do not see them.
    // Even tiles march UP the range axis, odd tiles DOWN (the CA / SOCA / GOCA window is symmetric, so
    // the result is the same): vertically adjacent tiles then touch the halo rows they share at the
    // same time (both at their start, or both at their end) and the second reader hits L2 instead of
    // re-fetching rows that were evicted long ago.  phys(i) = rbase + rs * i maps the logical row
    // of the march to the image row.
    const bool flip = (t & 1) == 0;
    const int rs = flip ? -1 : 1, rbase = flip ? 2 * r0 + tile_rows - 1 : 0;
    auto ld = [&](int i) -> uint32_t {
        const int ic = min(max(rbase + rs * i, 0), rows - 1);
        return __builtin_amdgcn_raw_buffer_load_b32(src, voff, ic * cols, 0);
    };

    // FIFO of rows in flight: row (r0-H+m) lives in pre[m % D]
    uint32_t pre[D];
#pragma unroll
    for (int d = 0; d < D; ++d)
        pre[d] = ld(r0 - H + d);

    uint32_t lo[R], hi[R]; // ring of prefix sums, slot of P[r-H+m] is (j+m)%R at unrolled step j
    lo[0] = 0;
    hi[0] = 0;
#pragma unroll
    for (int m = 0; m < R - 1; ++m) { // warm-up: rows r0-H .. r0+H
        const uint32_t x = pre[m % D];
        pre[m % D] = ld(r0 - H + m + D);
        lo[m + 1] = pk_add(lo[m], unpack_lo(x));
        hi[m + 1] = pk_add(hi[m], unpack_hi(x));
        __builtin_amdgcn_sched_barrier(0);
    }

    for (int g = 0; g < groups_per_tile; ++g) {
        const int rb = r0 + g * R;
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int r = rb + j;
            const uint32_t leadL = pk_sub_opaque(lo[(j + T) % R], lo[j]);
            const uint32_t leadH = pk_sub_opaque(hi[(j + T) % R], hi[j]);
            const uint32_t lagL = pk_sub(lo[(j + R - 1) % R], lo[(j + H + G + 1) % R]);
            const uint32_t lagH = pk_sub(hi[(j + R - 1) % R], hi[(j + H + G + 1) % R]);
            uint32_t sL, sH;
            if (ALG == SFE_CFAR_SOCA) {
                sL = pk_min(leadL, lagL);
                sH = pk_min(leadH, lagH);
            } else if (ALG == SFE_CFAR_GOCA) {
                sL = pk_max(leadL, lagL);
                sH = pk_max(leadH, lagH);
            } else {
                sL = pk_add(leadL, lagL);
                sH = pk_add(leadH, lagH);
            }
            const uint32_t xL = pk_sub(lo[(j + H + 1) % R], lo[(j + H) % R]);
            const uint32_t xH = pk_sub(hi[(j + H + 1) % R], hi[(j + H) % R]);
            const uint32_t l0 = s_lut[xL & 0xffffu], l1 = s_lut[xL >> 16];
            const uint32_t l2 = s_lut[xH & 0xffffu], l3 = s_lut[xH >> 16];
            // s < lut[x]  <=>  bit 15 of the 16-bit difference s - lut[x] (both < 2^15)
            const uint32_t dL = pk_sub(sL, l0 | (l1 << 16));
            const uint32_t dH = pk_sub(sH, l2 | (l3 << 16));
            // sign bits sit in bit 7 of bytes 1 and 3: gather the 4 bytes, shift to bit 0
            uint32_t o = (__builtin_amdgcn_perm(dH, dL, 0x07050301u) >> 7) & 0x01010101u;
            const int pr = rbase + rs * r; // image row of this output
            const uint32_t keep = (pr >= H && pr < rows - H) ? 0xffffffffu : 0u; // cfar.cpp:16,36
            o &= keep;
            __builtin_amdgcn_raw_buffer_store_b32(o, dst, voff, pr * cols, 0);

            const uint32_t x = pre[(j + R - 1) % D]; // row r+H+1
            pre[(j + R - 1) % D] = ld(r + H + 1 + D);
            lo[j] = pk_add(lo[(j + R - 1) % R], unpack_lo(x));
            hi[j] = pk_add(hi[(j + R - 1) % R], unpack_hi(x));
            // keep the scheduler from interleaving many rows (it would blow the register budget
            // the ring needs); the D-deep prefetch FIFO already hides the load latency
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Generic uint8 kernel: one thread per (column, row tile), integer running window sums with the
// taps re-read through L1/L2, decision evaluated directly in fp64 as written in cfar.cpp.
// Handles every variant, any window, the OS order statistic and the *2 threshold maps.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cfar_u8_generic(const uint8_t *__restrict__ img,
                                                       uint8_t *__restrict__ mask,
                                                       float *__restrict__ thr, int rows, int cols,
                                                       int n_frames, int tile_rows, int tiles_per_frame,
                                                       int alg, int T, int G, int k, double tau,
                                                       int intensity_thr)
{
    const long long w = (long long)blockIdx.x * 256 + threadIdx.x;
    const int c = (int)(w % cols);
    const long long tt = w / cols;
    const int t = (int)(tt % tiles_per_frame);
    const long long f = tt / tiles_per_frame;
    if (f >= n_frames)
        return;
    const int H = T + G;
    const uint8_t *__restrict__ in = img + (size_t)f * rows * cols + c;
    uint8_t *__restrict__ mo = mask + (size_t)f * rows * cols + c;
    float *__restrict__ to = thr ? thr + (size_t)f * rows * cols + c : nullptr;
    const int r0 = t * tile_rows, r1 = min(r0 + tile_rows, rows);
    int lead = 0, lag = 0;
    bool primed = false;
    for (int r = r0; r < r1; ++r) {
        uint8_t m = 0;
        float tv = 0.0f;
        if (r >= H && r < rows - H) {
            const int x = in[(size_t)r * cols];
            double tval;
            if (alg == SFE_CFAR_OS) {
                // k-th smallest (0-based) of the 2T training cells: smallest cell value v with
                // #{cells <= v} >= k+1  (std::nth_element value, cfar.cpp:91)
                int vk = 255;
                for (int a = 0; a < 2 * T; ++a) {
                    const int ia = (a < T) ? r - H + a : r + G + 1 + (a - T);
                    const int va = in[(size_t)ia * cols];
                    if (va >= vk)
                        continue;
                    int le = 0;
                    for (int b = 0; b < 2 * T; ++b) {
                        const int ib = (b < T) ? r - H + b : r + G + 1 + (b - T);
                        le += in[(size_t)ib * cols] <= va;
                    }
                    if (le >= k + 1)
                        vk = va;
                }
                tval = tau * (double)(float)vk; // cfar.cpp:92
            } else {
                if (!primed) {
                    lead = lag = 0;
                    for (int i = r - H; i < r - G; ++i)
                        lead += in[(size_t)i * cols];
                    for (int i = r + G + 1; i <= r + H; ++i)
                        lag += in[(size_t)i * cols];
                    primed = true;
                }
                if (alg == SFE_CFAR_CA)
                    tval = tau * (double)(float)(lead + lag) / (2.0 * T); // cfar.cpp:24
                else if (alg == SFE_CFAR_SOCA)
                    tval = tau * (double)(float)min(lead, lag) / T; // cfar.cpp:46-47
                else
                    tval = tau * (double)(float)max(lead, lag) / T; // cfar.cpp:69-70
                if (r + 1 < rows - H) { // slide both windows one row down
                    lead += in[(size_t)(r - G) * cols] - in[(size_t)(r - H) * cols];
                    lag += in[(size_t)(r + H + 1) * cols] - in[(size_t)(r + G + 1) * cols];
                }
            }
            m = (double)(float)x > tval;
            if (intensity_thr >= 0)
                m &= (x > intensity_thr);
            tv = (float)tval;
        }
        mo[(size_t)r * cols] = m;
        if (to)
            to[(size_t)r * cols] = tv;
    }
}

// ---------------------------------------------------------------------------------------------
// Float images: one thread per pixel, float accumulation in the reference's ascending-i order.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cfar_f32_naive(const float *__restrict__ img,
                                                      uint8_t *__restrict__ mask,
                                                      float *__restrict__ thr, int rows, int cols,
                                                      int alg, int T, int G, int k, double tau)
{
    const long long w = (long long)blockIdx.x * 256 + threadIdx.x;
    if (w >= (long long)rows * cols)
        return;
    const int r = (int)(w / cols), c = (int)(w % cols);
    const int H = T + G;
    uint8_t m = 0;
    float tv = 0.0f;
    if (r >= H && r < rows - H) {
        const float *__restrict__ in = img + c;
        double tval;
        if (alg == SFE_CFAR_CA) {
            float s = 0.0f;
            for (int i = r - H; i <= r + H; ++i)
                if (abs(i - r) > G)
                    s = __fadd_rn(s, in[(size_t)i * cols]);
            tval = tau * (double)s / (2.0 * T);
        } else if (alg == SFE_CFAR_SOCA || alg == SFE_CFAR_GOCA) {
            float lead = 0.0f, lag = 0.0f;
            for (int i = r - H; i < r - G; ++i)
                lead = __fadd_rn(lead, in[(size_t)i * cols]);
            for (int i = r + G + 1; i <= r + H; ++i)
                lag = __fadd_rn(lag, in[(size_t)i * cols]);
            const float s = (alg == SFE_CFAR_SOCA) ? (lag < lead ? lag : lead) : (lead < lag ? lag : lead);
            tval = tau * (double)s / T;
        } else {
            // k-th smallest by rank counting: value v with #{< v} <= k < #{<= v}
            float vk = 0.0f;
            for (int a = 0; a < 2 * T; ++a) {
                const int ia = (a < T) ? r - H + a : r + G + 1 + (a - T);
                const float va = in[(size_t)ia * cols];
                int lt = 0, le = 0;
                for (int b = 0; b < 2 * T; ++b) {
                    const int ib = (b < T) ? r - H + b : r + G + 1 + (b - T);
                    const float vb = in[(size_t)ib * cols];
                    lt += vb < va;
                    le += vb <= va;
                }
                if (lt <= k && k < le)
                    vk = va;
            }
            tval = tau * (double)vk;
        }
        m = (double)in[(size_t)r * cols] > tval;
        tv = (float)tval;
    }
    mask[w] = m;
    if (thr)
        thr[w] = tv;
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static bool build_lut(int alg, int T, double tau, int intensity_thr, CfarLut *lut)
{
    if (!(tau >= 0.0) || !std::isfinite(tau))
        return false;
    const int smax = (alg == SFE_CFAR_CA) ? 255 * 2 * T : 255 * T;
    if (smax + 1 > 65535)
        return false;
    auto passes = [&](int x, int s) -> bool {
        const float sf = (float)s; // the reference holds the sum in a float (exact here)
        const double t = (alg == SFE_CFAR_CA) ? tau * sf / (2.0 * T) : tau * sf / T;
        return (double)(float)x > t;
    };
    for (int x = 0; x < 256; ++x) {
        int cnt = 0;
        if (!(intensity_thr >= 0 && x <= intensity_thr) && passes(x, 0)) {
            int lo = 0, hi = smax; // passes(lo) true; find the largest passing s (monotone in s)
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (passes(x, mid))
                    lo = mid;
                else
                    hi = mid - 1;
            }
            cnt = lo + 1;
        }
        lut->v[x] = (uint16_t)cnt;
    }
    return true;
}

template <int T, int G, int D>
static void launch_ring(sfe_ctx *ctx, int alg, const uint8_t *d_img, uint8_t *d_mask, int rows, int cols,
                        int n_frames, int groups, int tiles, const CfarLut &lut)
{
    const int chunks = ((cols >> 2) + 63) / 64;
    const long long bpf = ((long long)tiles * chunks + 3) / 4;             // workgroups per frame
    const unsigned blocks = (unsigned)((((long long)n_frames + 7) / 8) * 8 * bpf); // frames padded to the 8 XCDs
    if (alg == SFE_CFAR_SOCA)
        hipLaunchKernelGGL((cfar_u8_ring<T, G, SFE_CFAR_SOCA, D>), dim3(blocks), dim3(256), 0, ctx->stream, d_img,
                           d_mask, rows, cols, n_frames, groups, tiles, chunks, lut);
    else if (alg == SFE_CFAR_GOCA)
        hipLaunchKernelGGL((cfar_u8_ring<T, G, SFE_CFAR_GOCA, D>), dim3(blocks), dim3(256), 0, ctx->stream, d_img,
                           d_mask, rows, cols, n_frames, groups, tiles, chunks, lut);
    else
        hipLaunchKernelGGL((cfar_u8_ring<T, G, SFE_CFAR_CA, D>), dim3(blocks), dim3(256), 0, ctx->stream, d_img,
                           d_mask, rows, cols, n_frames, groups, tiles, chunks, lut);
}

// R-row groups per tile.  Measured on MI355X (tools/cfar_sweep.py, 1024 frames of 1024x512, XCD-aware
// map + alternating march direction):  1 group/tile 5.1 TB/s with FETCH = 1.30x the image bytes,
// 2 groups 5.1 TB/s with 1.13x, 4 groups 4.9 TB/s with 1.07x, whole column 3.1 TB/s.  Short tiles win
// on time (the kernel is bound by each wave's serial row march, more independent waves hide it);
// 2 groups keep that speed and most of the 2*(T+G) halo rows a tile re-reads are L2 hits.
static int default_groups(const sfe_ctx *, int rows, int, int, int R) { return rows >= 2 * R ? 2 : 1; }

static int cfar_u8_dev(sfe_ctx *ctx, const uint8_t *d_img, int n_frames, int rows, int cols, int alg,
                       int T, int G, int k, double tau, int intensity_thr, uint8_t *d_mask, float *d_thr)
{
    SFE_ARG(ctx, d_img && d_mask);
    SFE_ARG(ctx, n_frames >= 0 && rows >= 0 && cols >= 0);
    SFE_ARG(ctx, alg >= SFE_CFAR_CA && alg <= SFE_CFAR_OS);
    SFE_ARG(ctx, T >= 1 && G >= 0);
    if (alg == SFE_CFAR_OS)
        SFE_ARG(ctx, k >= 0 && k < 2 * T);
    if (n_frames == 0 || rows == 0 || cols == 0)
        return 0;
    CfarLut lut;
    bool ring = (alg != SFE_CFAR_OS) && !d_thr && (cols % 4 == 0) && cols >= 256 && rows >= 52 && (size_t)rows * cols < (1u << 30) &&
                ctx->cfar_variant != 1 &&
                ((reinterpret_cast<uintptr_t>(d_img) | reinterpret_cast<uintptr_t>(d_mask)) % 4 == 0) &&
                (T == 20 && G == 5) && build_lut(alg, T, tau, intensity_thr, &lut);
    if (ctx->cfar_variant >= 2 && !ring)
        return sfe_set_err(ctx, SFE_ERR_ARG, "ring CFAR kernel forced but not applicable to this call");
    if (ring) {
        constexpr int R = 2 * (20 + 5) + 2;
        int groups = ctx->cfar_tile_rows > 0 ? std::max(1, std::min(ctx->cfar_tile_rows / R, rows / R))
                                             : default_groups(ctx, rows, cols, n_frames, R);
        const int tiles = (rows + groups * R - 1) / (groups * R);
        if (ctx->cfar_variant == 3)
            launch_ring<20, 5, 13>(ctx, alg, d_img, d_mask, rows, cols, n_frames, groups, tiles, lut);
        else
            launch_ring<20, 5, 4>(ctx, alg, d_img, d_mask, rows, cols, n_frames, groups, tiles, lut);
    } else {
        const int tr = std::min(rows, 64);
        const int tiles = (rows + tr - 1) / tr;
        const long long threads = (long long)n_frames * tiles * cols;
        const unsigned blocks = (unsigned)((threads + 255) / 256);
        hipLaunchKernelGGL(cfar_u8_generic, dim3(blocks), dim3(256), 0, ctx->stream, d_img, d_mask, d_thr, rows,
                           cols, n_frames, tr, tiles, alg, T, G, k, tau, intensity_thr);
    }
    SFE_LAUNCH_CHECK(ctx);
    return 0;
}

extern "C" {

int sfe_cfar_set_tuning(sfe_ctx *ctx, int tile_rows, int variant)
{
    if (!ctx)
        return SFE_ERR_ARG;
    SFE_ARG(ctx, tile_rows >= 0 && variant >= 0 && variant <= 3);
    ctx->cfar_tile_rows = tile_rows;
    ctx->cfar_variant = variant;
    return 0;
}

int sfe_cfar_u8_batch_dev(sfe_ctx *ctx, const uint8_t *d_img, int n_frames, int rows, int cols, int alg,
                          int train_hs, int guard_hs, int k, double tau, int intensity_thr, uint8_t *d_mask,
                          float *d_thr)
{
    if (int rc = sfe_use(ctx))
        return rc;
    return cfar_u8_dev(ctx, d_img, n_frames, rows, cols, alg, train_hs, guard_hs, k, tau, intensity_thr, d_mask,
                       d_thr);
}

int sfe_cfar_u8(sfe_ctx *ctx, const uint8_t *img, int rows, int cols, int alg, int train_hs, int guard_hs, int k,
                double tau, int intensity_thr, uint8_t *mask_out, float *thr_out)
{
    if (int rc = sfe_use(ctx))
        return rc;
    SFE_ARG(ctx, img && mask_out && rows >= 0 && cols >= 0);
    const size_t n = (size_t)rows * cols;
    if (n == 0)
        return 0;
    uint8_t *d_img = (uint8_t *)sfe_scratch(ctx, 0, n);
    uint8_t *d_mask = (uint8_t *)sfe_scratch(ctx, 1, n);
    float *d_thr = thr_out ? (float *)sfe_scratch(ctx, 2, n * sizeof(float)) : nullptr;
    if (!d_img || !d_mask || (thr_out && !d_thr))
        return SFE_ERR_HIP;
    SFE_HIP(ctx, hipMemcpyAsync(d_img, img, n, hipMemcpyHostToDevice, ctx->stream));
    if (int rc = cfar_u8_dev(ctx, d_img, 1, rows, cols, alg, train_hs, guard_hs, k, tau, intensity_thr, d_mask,
                             d_thr))
        return rc;
    SFE_HIP(ctx, hipMemcpyAsync(mask_out, d_mask, n, hipMemcpyDeviceToHost, ctx->stream));
    if (thr_out)
        SFE_HIP(ctx, hipMemcpyAsync(thr_out, d_thr, n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    SFE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int sfe_cfar_f32(sfe_ctx *ctx, const float *img, int rows, int cols, int alg, int train_hs, int guard_hs, int k,
                 double tau, uint8_t *mask_out, float *thr_out)
{
    if (int rc = sfe_use(ctx))
        return rc;
    SFE_ARG(ctx, img && mask_out && rows >= 0 && cols >= 0);
    SFE_ARG(ctx, alg >= SFE_CFAR_CA && alg <= SFE_CFAR_OS);
    SFE_ARG(ctx, train_hs >= 1 && guard_hs >= 0);
    if (alg == SFE_CFAR_OS)
        SFE_ARG(ctx, k >= 0 && k < 2 * train_hs);
    const size_t n = (size_t)rows * cols;
    if (n == 0)
        return 0;
    float *d_img = (float *)sfe_scratch(ctx, 0, n * sizeof(float));
    uint8_t *d_mask = (uint8_t *)sfe_scratch(ctx, 1, n);
    float *d_thr = thr_out ? (float *)sfe_scratch(ctx, 2, n * sizeof(float)) : nullptr;
    if (!d_img || !d_mask || (thr_out && !d_thr))
        return SFE_ERR_HIP;
    SFE_HIP(ctx, hipMemcpyAsync(d_img, img, n * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    const unsigned blocks = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(cfar_f32_naive, dim3(blocks), dim3(256), 0, ctx->stream, d_img, d_mask, d_thr, rows, cols,
                       alg, train_hs, guard_hs, k, tau);
    SFE_LAUNCH_CHECK(ctx);
    SFE_HIP(ctx, hipMemcpyAsync(mask_out, d_mask, n, hipMemcpyDeviceToHost, ctx->stream));
    if (thr_out)
        SFE_HIP(ctx, hipMemcpyAsync(thr_out, d_thr, n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    SFE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

} // extern "C"
