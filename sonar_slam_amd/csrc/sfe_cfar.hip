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
#include <cstdlib>

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

// The threshold map's value for a window sum s, thr = (float)(tau * (double)(float)s / D) with D = T or 2T (cfar.cpp:27,46,67
// as the host table below restates it), computed instead of fetched: one table gather per pixel is one L1 tag look-up
// per pixel, and that rate -- not HBM -- bounded the map kernels (0.47-0.60 ms per 512 frames whatever the window).
// The quotient is formed with the reciprocal of D and two residual corrections (fma), which is the correctly rounded
// quotient for every operand the host has tried: cfar_u8_dev evaluates this very sequence for every possible sum of the
// launch and compares it with the table bit by bit -- a single difference and the kernel keeps the table (ta.on = 0).
struct CfarThrArith {
    double tau, rinv, d;
    int on;
};
__host__ __device__ __forceinline__ float cfar_thr_arith(const CfarThrArith &ta, uint32_t sv)
{
    const double p = ta.tau * (double)(float)sv;
    double q = p * ta.rinv;
    double e = fma(-ta.d, q, p);
    q = fma(e, ta.rinv, q);
    e = fma(-ta.d, q, p);
    q = fma(e, ta.rinv, q);
    return (float)q;
}


// Tiles and column chunks OVERLAP instead of being predicated: a tile is always a whole number
// of R-row groups (the last tile is shifted up so it ends at the last row) and the last 64-lane
// chunk is shifted left so it ends at the last beam.  Overlapped outputs are recomputed with
// identical values, so the body has no branches, no exec masking and no tail code.
//
// BITS: the detections leave the kernel bit-packed (bit iy*cols+ix of the frame's bit stream, LSB first: the
// layout extract_scatter_kernel reads, sfe_remap.hip) instead of as 0/1 bytes: a lane folds its 4 decisions
// into a nibble (one v_dot4), 8 lanes OR their nibbles into a 32-bit word over three DPP steps and one lane
// of the 8 stores it -- the other 56 lanes aim past the end of the buffer, where the hardware drops the
// store.  Needs cols % 32 == 0 (a row is a whole number of words and every 64-lane chunk starts on one).
// THR (the *2 variants, cfar.cpp:98-192): the float threshold map next to the byte mask, from the window sums this
// kernel holds in registers anyway -- thr = (float)(tau * s / T) computed per pixel (cfar_thr_arith), one float4 per lane and row.
template <int T, int G, int ALG, int D, bool BITS, bool THR = false>
__global__ __launch_bounds__(256, 3) void cfar_u8_ring(const uint8_t *__restrict__ img,
                                                    uint8_t *__restrict__ mask, int rows, int cols,
                                                    int n_frames, int groups_per_tile,
                                                    int tiles_per_frame, int chunks_per_row,
                                                    long long out_frame_bytes, CfarLut lut, float *__restrict__ thr = nullptr,
                                                    CfarThrArith ta = CfarThrArith{0.0, 0.0, 1.0, 0}, int full_last_tile = 0)
{
    static_assert(!(THR && BITS), "the threshold map goes with the byte mask");
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
    // tiles of groups_per_tile * R rows; the LAST tile only runs the groups it needs to reach the end of the image and is
    // shifted up so that it ends there (round 5: it used to run a whole tile -- at 1024 rows and 124-row tiles 92 of its
    // rows were computed twice, 9 % of the launch)
    const int full_rows = groups_per_tile * R;       // <= rows
    const bool last_tile = t == tiles_per_frame - 1;
    const int my_groups = (last_tile && !full_last_tile) ? (rows - t * full_rows + R - 1) / R : groups_per_tile;
    const int tile_rows = my_groups * R;
    const int r0 = last_tile ? rows - tile_rows : t * full_rows;

    const size_t frame_bytes = (size_t)rows * cols;
    const sfe_rsrc_t src = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(img + (size_t)f * frame_bytes),
                                                             0, (int)frame_bytes, 0x00020000);
    const sfe_rsrc_t dst = __builtin_amdgcn_make_buffer_rsrc(mask + (size_t)f * (size_t)out_frame_bytes, 0,
                                                             (int)out_frame_bytes, 0x00020000);
    // BITS: byte offset of this lane's word inside its row (lanes 0, 8, .. store), shift of its nibble
    const uint32_t boff = (lane & 7) == 0 ? (uint32_t)(cx0 + lane) >> 1 : 0x80000000u;
    const uint32_t nsh = (uint32_t)(lane & 7) * 4u;
    const uint32_t nrot = (7u - nsh) & 31u; // BITS: the nibble comes out of the dot product at bits 7..10
    (void)nsh;
    if (BITS && wvf == wpf - 1 && lane == 0) // the pad word behind the last row (read by the extraction's taps)
        __builtin_amdgcn_raw_buffer_store_b32(0u, dst, (uint32_t)(frame_bytes >> 3), 0, 0);

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

    for (int g = 0; g < my_groups; ++g) {
        const int rb = r0 + g * R;
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int r = rb + j;
            const uint32_t xL = pk_sub(lo[(j + H + 1) % R], lo[(j + H) % R]);
            const uint32_t xH = pk_sub(hi[(j + H + 1) % R], hi[(j + H) % R]);
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
            const int pr = rbase + rs * r; // image row of this output
            const bool in_rows = pr >= H && pr < rows - H; // cfar.cpp:16,36
            const uint32_t keep = in_rows ? 0xffffffffu : 0u;
            if (BITS) {
                // s < lut[x]  <=>  bit 15 of the 16-bit difference s - lut[x] (both < 2^15).  (Joining the two 16-bit
                // entries with ds_read_u16_d16 / _d16_hi instead of a v_perm was tried in round 4: with SRAM ECC on, as on
                // MI300 / MI355X, a d16 load zeroes the other half instead of keeping it.)
                const uint32_t l0 = s_lut[xL & 0xffffu], l1 = s_lut[xL >> 16];
                const uint32_t l2 = s_lut[xH & 0xffffu], l3 = s_lut[xH >> 16];
                const uint32_t dL = pk_sub(sL, l0 | (l1 << 16));
                const uint32_t dH = pk_sub(sH, l2 | (l3 << 16));
                // the sign bits sit in bit 7 of bytes 1 and 3: gather the 4 bytes; ONE mask keeps the sign bits and
                // zeroes the rows outside cfar.cpp's loop; the dot product with (1, 2, 4, 8) of bytes that are 0 or 0x80
                // is the nibble << 7, which a rotate puts in its place in the word
                const uint32_t km = in_rows ? 0x80808080u : 0u;
                const uint32_t o = __builtin_amdgcn_perm(dH, dL, 0x07050301u) & km;
                uint32_t v = __builtin_amdgcn_udot4(o, 0x08040201u, 0u, false);
                v = __builtin_amdgcn_alignbit(v, v, nrot);
                v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false);  // quad_perm [1,0,3,2]
                v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false);  // quad_perm [2,3,0,1]
                v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, false); // row_half_mirror
                __builtin_amdgcn_raw_buffer_store_b32(v, dst, boff, pr * (cols >> 3), 0);
            } else {
                const uint32_t l0 = s_lut[xL & 0xffffu], l1 = s_lut[xL >> 16];
                const uint32_t l2 = s_lut[xH & 0xffffu], l3 = s_lut[xH >> 16];
                // s < lut[x]  <=>  bit 15 of the 16-bit difference s - lut[x] (both < 2^15)
                const uint32_t dL = pk_sub(sL, l0 | (l1 << 16));
                const uint32_t dH = pk_sub(sH, l2 | (l3 << 16));
                // sign bits sit in bit 7 of bytes 1 and 3: gather the 4 bytes, shift to bit 0
                uint32_t o = (__builtin_amdgcn_perm(dH, dL, 0x07050301u) >> 7) & 0x01010101u;
                o &= keep;
                __builtin_amdgcn_raw_buffer_store_b32(o, dst, voff, pr * cols, 0);
            }
            if (THR) {
                float4 tv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (keep)
                    tv = make_float4(cfar_thr_arith(ta, sL & 0xffffu), cfar_thr_arith(ta, sL >> 16),
                                     cfar_thr_arith(ta, sH & 0xffffu), cfar_thr_arith(ta, sH >> 16));
                *reinterpret_cast<float4 *>(thr + (size_t)f * frame_bytes + (size_t)pr * cols + voff) = tv;
            }

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
// Sliding-sum kernel: ANY window (train_hs, guard_hs at run time), CA / SOCA / GOCA, optionally the float
// threshold map of the *2 variants (cfar.cpp:98-192).  Same work decomposition and the same exact decision
// table as the ring kernel -- one lane owns 4 adjacent beams, marches down the range axis, a wave reads 256
// contiguous bytes per row -- but the window is kept as two running sums per beam (packed u16 pairs, exact
// mod 2^16 for sums <= 65535) that slide by one row per step:
//     lead += x[r-G] - x[r-T-G],   lag += x[r+T+G+1] - x[r+G+1]
// so a step costs five dword loads per lane: the new row r+T+G+1 from HBM, the four others are rows this
// wave read a few steps ago (L1 / L2 hits: the window of a 256-beam strip is 2(T+G) x 256 B).  No register
// ring, hence no compile-time window: this is what every feature.yaml window other than the shipped
// (Ntc 40, Ngc 10) runs on, at ~2x the instructions per pixel of the ring kernel.
// THR: thr[r][c] = (float)(tau * s / T) looked up in a table over the integer sums s (built on the host
// with the reference's double expression, like the decision table) and stored as one float4 per lane.
// ---------------------------------------------------------------------------------------------
template <int ALG, bool THR>
__global__ __launch_bounds__(256) void cfar_u8_slide(const uint8_t *__restrict__ img, uint8_t *__restrict__ mask,
                                                     float *__restrict__ thr, const float *__restrict__ thr_tab,
                                                     int rows, int cols, int n_frames, int T, int G, int tile_rows,
                                                     int tiles_per_frame, int chunks_per_row, CfarLut lut, CfarThrArith ta)
{
    __shared__ uint16_t s_lut[256];
    s_lut[threadIdx.x] = lut.v[threadIdx.x];
    __syncthreads();
    const int H = T + G;
    const int lane = threadIdx.x & 63;
    const int wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // frame f -> XCD f % 8, its workgroups consecutive in that XCD's dispatch order (as the ring kernel)
    const int wpf = tiles_per_frame * chunks_per_row;
    const int bpf = (wpf + 3) >> 2;
    const int xcd = blockIdx.x & 7, kb = blockIdx.x >> 3;
    const int fg = kb / bpf;
    const long long f = (long long)fg * 8 + xcd;
    const int wvf = (kb - fg * bpf) * 4 + wave_in_block;
    if (f >= n_frames || wvf >= wpf)
        return;
    const int chunk = wvf % chunks_per_row;
    const int t = wvf / chunks_per_row;
    const int lpr = cols >> 2;                                  // 4-beam lanes per row
    const int cx0 = lpr >= 64 ? min(chunk * 64, lpr - 64) : 0;  // last chunk shifted left (outputs recomputed identically)
    if (cx0 + lane >= lpr)
        return;                                                 // images narrower than 256 beams: the spare lanes idle
    const uint32_t voff = (uint32_t)(cx0 + lane) * 4u;
    const int r0 = t * tile_rows, r1 = min(r0 + tile_rows, rows);
    const size_t frame_bytes = (size_t)rows * cols;
    const sfe_rsrc_t src = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(img + (size_t)f * frame_bytes),
                                                             0, (int)frame_bytes, 0x00020000);
    const sfe_rsrc_t dst = __builtin_amdgcn_make_buffer_rsrc(mask + (size_t)f * frame_bytes, 0,
                                                             (int)frame_bytes, 0x00020000);
    // rows outside the image are clamped: they only enter windows of border rows, whose output is forced to 0
    auto ld = [&](int i) -> uint32_t {
        const int ic = min(max(i, 0), rows - 1);
        return __builtin_amdgcn_raw_buffer_load_b32(src, voff, ic * cols, 0);
    };
    uint32_t leadL = 0, leadH = 0, lagL = 0, lagH = 0;
    for (int i = 0; i < T; i += 2) { // the windows of the tile's first row (two rows of each window per trip)
        const uint32_t a0 = ld(r0 - H + i), b0 = ld(r0 + G + 1 + i);
        const uint32_t a1 = i + 1 < T ? ld(r0 - H + i + 1) : 0u, b1 = i + 1 < T ? ld(r0 + G + 2 + i) : 0u;
        leadL = pk_add(pk_add(leadL, unpack_lo(a0)), unpack_lo(a1));
        leadH = pk_add(pk_add(leadH, unpack_hi(a0)), unpack_hi(a1));
        lagL = pk_add(pk_add(lagL, unpack_lo(b0)), unpack_lo(b1));
        lagH = pk_add(pk_add(lagH, unpack_hi(b0)), unpack_hi(b1));
    }
    uint32_t x = ld(r0), a = ld(r0 - G), b = ld(r0 - H), c = ld(r0 + H + 1), d = ld(r0 + G + 1);
    for (int r = r0; r < r1; ++r) {
        // the next row's five loads go out before this row is evaluated
        const uint32_t xn = ld(r + 1), an = ld(r + 1 - G), bn = ld(r + 1 - H), cn = ld(r + H + 2), dn = ld(r + G + 2);
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
        const uint32_t l0 = s_lut[x & 0xffu], l1 = s_lut[(x >> 8) & 0xffu];
        const uint32_t l2 = s_lut[(x >> 16) & 0xffu], l3 = s_lut[x >> 24];
        // s < lut[x] per beam (sums and table entries reach 65535 here: compare, do not subtract)
        const uint32_t s0 = sL & 0xffffu, s1 = sL >> 16, s2 = sH & 0xffffu, s3 = sH >> 16;
        uint32_t o = (s0 < l0 ? 1u : 0u) | (s1 < l1 ? 0x100u : 0u) | (s2 < l2 ? 0x10000u : 0u) | (s3 < l3 ? 0x1000000u : 0u);
        const bool inside = r >= H && r < rows - H; // cfar.cpp:16,36
        o = inside ? o : 0u;
        __builtin_amdgcn_raw_buffer_store_b32(o, dst, voff, r * cols, 0);
        if (THR) {
            float4 tv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (inside)
                tv = ta.on ? make_float4(cfar_thr_arith(ta, s0), cfar_thr_arith(ta, s1), cfar_thr_arith(ta, s2), cfar_thr_arith(ta, s3))
                           : make_float4(thr_tab[s0], thr_tab[s1], thr_tab[s2], thr_tab[s3]);
            *reinterpret_cast<float4 *>(thr + (size_t)f * frame_bytes + (size_t)r * cols + voff) = tv;
        }
        leadL = pk_sub(pk_add(leadL, unpack_lo(a)), unpack_lo(b));
        leadH = pk_sub(pk_add(leadH, unpack_hi(a)), unpack_hi(b));
        lagL = pk_sub(pk_add(lagL, unpack_lo(c)), unpack_lo(d));
        lagH = pk_sub(pk_add(lagH, unpack_hi(c)), unpack_hi(d));
        x = xn;
        a = an;
        b = bn;
        c = cn;
        d = dn;
    }
}

// ---------------------------------------------------------------------------------------------
// The same sliding sums with the window rows staged in LDS: each wave keeps the last R = 2(T+G)+2 rows of its
// 256-beam strip in a private LDS ring (ring[row mod R][lane], one dword per lane and row: conflict-free), so a
// step is ONE global load (the new row, prefetched D rows ahead into registers) + one ds_write + four ds_reads for
// the cells that enter / leave the two sums.  The plain sliding-sum kernel above re-reads those four rows through
// the caches, and with a few thousand waves in flight their windows (R x 256 B each) overflow the 4 MB L2 of an
// XCD: it then moves ~5 B per pixel over the fabric instead of 1 (measured 2.6 TB/s algorithmic; this one keeps
// the traffic of the register-ring kernel for any run-time window).  LDS: R KiB per 4-wave workgroup.
// ---------------------------------------------------------------------------------------------
#define SLIDE_D 4 // rows in flight per lane
template <int ALG, bool THR>
__global__ __launch_bounds__(256) void cfar_u8_slide_lds(const uint8_t *__restrict__ img, uint8_t *__restrict__ mask,
                                                         float *__restrict__ thr, const float *__restrict__ thr_tab,
                                                         int rows, int cols, int n_frames, int T, int G, int tile_rows,
                                                         int tiles_per_frame, int chunks_per_row, CfarLut lut, CfarThrArith ta)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t s_ring_all[]; // [4 waves][R][64]
    __shared__ uint16_t s_lut[256];
    s_lut[threadIdx.x] = lut.v[threadIdx.x];
    __syncthreads();
    const int H = T + G, R = 2 * H + 2;
    const int lane = threadIdx.x & 63;
    const int wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wpf = tiles_per_frame * chunks_per_row;
    const int bpf = (wpf + 3) >> 2;
    const int xcd = blockIdx.x & 7, kb = blockIdx.x >> 3;
    const int fg = kb / bpf;
    const long long f = (long long)fg * 8 + xcd;
    const int wvf = (kb - fg * bpf) * 4 + wave_in_block;
    if (f >= n_frames || wvf >= wpf)
        return;
    const int chunk = wvf % chunks_per_row;
    const int t = wvf / chunks_per_row;
    const int lpr = cols >> 2;
    const int cx0 = lpr >= 64 ? min(chunk * 64, lpr - 64) : 0;
    if (cx0 + lane >= lpr)
        return;
    const uint32_t voff = (uint32_t)(cx0 + lane) * 4u;
    const int r0 = t * tile_rows, r1 = min(r0 + tile_rows, rows);
    const size_t frame_bytes = (size_t)rows * cols;
    const sfe_rsrc_t src = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(img + (size_t)f * frame_bytes),
                                                             0, (int)frame_bytes, 0x00020000);
    const sfe_rsrc_t dst = __builtin_amdgcn_make_buffer_rsrc(mask + (size_t)f * frame_bytes, 0,
                                                             (int)frame_bytes, 0x00020000);
    auto ld = [&](int i) -> uint32_t { // clamped: rows outside the image only enter windows of border rows
        const int ic = min(max(i, 0), rows - 1);
        return __builtin_amdgcn_raw_buffer_load_b32(src, voff, ic * cols, 0);
    };
    uint32_t *ring = s_ring_all + (size_t)wave_in_block * R * 64 + lane; // row slot s lives at ring[s * 64]
    // fill: rows r0-H .. r0+H+1 -> slots 0 .. R-1; the two sums of row r0 on the way
    uint32_t leadL = 0, leadH = 0, lagL = 0, lagH = 0;
    for (int m0 = 0; m0 < R; m0 += SLIDE_D) {
        uint32_t v[SLIDE_D];
#pragma unroll
        for (int u = 0; u < SLIDE_D; ++u)
            v[u] = ld(r0 - H + m0 + u);
#pragma unroll
        for (int u = 0; u < SLIDE_D; ++u) {
            const int m = m0 + u;
            if (m < R) {
                ring[m * 64] = v[u];
                if (m < T) {
                    leadL = pk_add(leadL, unpack_lo(v[u]));
                    leadH = pk_add(leadH, unpack_hi(v[u]));
                }
                if (m > H + G && m <= 2 * H) {
                    lagL = pk_add(lagL, unpack_lo(v[u]));
                    lagH = pk_add(lagH, unpack_hi(v[u]));
                }
            }
        }
    }
    uint32_t pre[SLIDE_D]; // rows r+H+2 .. of the steps to come
#pragma unroll
    for (int u = 0; u < SLIDE_D; ++u)
        pre[u] = ld(r0 + H + 2 + u);
    // slots of the five rows a step touches (wave-uniform counters, wrapped at R)
    int sb = 0, sx = H, sa = H - G, sd = H + G + 1, sc = 2 * H + 1;
    auto wrap = [&](int &v) { v = (v + 1 == R) ? 0 : v + 1; };
    for (int rb = r0; rb < r1; rb += SLIDE_D) {
#pragma unroll
        for (int u = 0; u < SLIDE_D; ++u) {
            const int r = rb + u;
            if (r < r1) { // wave-uniform
                const uint32_t x = ring[sx * 64], a = ring[sa * 64], b = ring[sb * 64], c = ring[sc * 64], d = ring[sd * 64];
                ring[sb * 64] = pre[u]; // row r+H+2 takes the slot of row r-H (read above)
                pre[u] = ld(r + H + 2 + SLIDE_D);
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
                const uint32_t l0 = s_lut[x & 0xffu], l1 = s_lut[(x >> 8) & 0xffu];
                const uint32_t l2 = s_lut[(x >> 16) & 0xffu], l3 = s_lut[x >> 24];
                const uint32_t s0 = sL & 0xffffu, s1 = sL >> 16, s2 = sH & 0xffffu, s3 = sH >> 16;
                uint32_t o = (s0 < l0 ? 1u : 0u) | (s1 < l1 ? 0x100u : 0u) | (s2 < l2 ? 0x10000u : 0u) |
                             (s3 < l3 ? 0x1000000u : 0u);
                const bool inside = r >= H && r < rows - H; // cfar.cpp:16,36
                o = inside ? o : 0u;
                __builtin_amdgcn_raw_buffer_store_b32(o, dst, voff, r * cols, 0);
                if (THR) {
                    float4 tv = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (inside)
                        tv = ta.on ? make_float4(cfar_thr_arith(ta, s0), cfar_thr_arith(ta, s1), cfar_thr_arith(ta, s2), cfar_thr_arith(ta, s3))
                           : make_float4(thr_tab[s0], thr_tab[s1], thr_tab[s2], thr_tab[s3]);
                    *reinterpret_cast<float4 *>(thr + (size_t)f * frame_bytes + (size_t)r * cols + voff) = tv;
                }
                leadL = pk_sub(pk_add(leadL, unpack_lo(a)), unpack_lo(b));
                leadH = pk_sub(pk_add(leadH, unpack_hi(a)), unpack_hi(b));
                lagL = pk_sub(pk_add(lagL, unpack_lo(c)), unpack_lo(d));
                lagH = pk_sub(pk_add(lagH, unpack_hi(c)), unpack_hi(d));
                wrap(sb);
                wrap(sx);
                wrap(sa);
                wrap(sd);
                wrap(sc);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// OS-CFAR (cfar.cpp:76-96, 170-192) for uint8 images: sliding 256-bin histogram + rank walk.
// The order statistic of the 2T training cells is the smallest value v with #{cells <= v} >= k + 1.  One lane owns
// one beam and marches down the range axis; its window's histogram (256 uint8 counts, 2T <= 255) lives in LDS,
// value-major (hist[v][lane]).  Moving one row down replaces two cells (one leaves / one enters each half-window):
// four read-modify-writes, then the lane walks v up or down until
//     below <= k < below + hist[v],        below = #{cells < v}
// holds again -- a step or two, because 2 of the 2T cells changed.  The reference gathers the 2T cells and runs
// nth_element for every pixel; the generic kernel counted ranks in O((2T)^2).  Decision and threshold are table
// look-ups over the 256 possible values of v, built on the host with the reference's double expression.
// ---------------------------------------------------------------------------------------------
struct CfarOsTab {
    uint16_t min_x[256]; // pixel fires iff x >= min_x[v]  (256 = never); the intensity gate is folded in
    float thr[256];      // (float)(tau * v)
};

__global__ __launch_bounds__(64) void cfar_u8_os(const uint8_t *__restrict__ img, uint8_t *__restrict__ mask,
                                                 float *__restrict__ thr, int rows, int cols, int n_frames, int T, int G,
                                                 int k, int tile_rows, int tiles_per_frame, int chunks_per_row,
                                                 CfarOsTab tab)
{
    __shared__ uint8_t s_hist[256 * 64];
    __shared__ uint16_t s_minx[256];
    __shared__ float s_thr[256];
    const int lane = threadIdx.x;
    for (int i = lane; i < 256; i += 64) {
        s_minx[i] = tab.min_x[i];
        s_thr[i] = tab.thr[i];
    }
    for (int i = lane; i < 256 * 64 / 4; i += 64)
        reinterpret_cast<uint32_t *>(s_hist)[i] = 0u;
    __syncthreads();
    const int H = T + G;
    const int wpf = tiles_per_frame * chunks_per_row;
    const long long f = blockIdx.x / wpf;
    const int wvf = blockIdx.x % wpf;
    const int chunk = wvf % chunks_per_row, t = wvf / chunks_per_row;
    const int c = chunk * 64 + lane;
    if (f >= n_frames || c >= cols)
        return; // (no barrier below)
    const int r0 = t * tile_rows, r1 = min(r0 + tile_rows, rows);
    const uint8_t *__restrict__ in = img + (size_t)f * rows * cols + c;
    uint8_t *__restrict__ mo = mask + (size_t)f * rows * cols + c;
    float *__restrict__ to = thr ? thr + (size_t)f * rows * cols + c : nullptr;
    auto ld = [&](int i) -> int { return in[(size_t)min(max(i, 0), rows - 1) * cols]; }; // clamped: border rows only
    uint8_t *h = s_hist + lane;                                                            // h[v * 64]
    for (int i = 0; i < T; ++i) {
        h[ld(r0 - H + i) * 64] += 1;
        h[ld(r0 + G + 1 + i) * 64] += 1;
    }
    int v = 0, below = 0;
    // per row: the pixel, and the four cells that change when the windows move on; fetched one row ahead (the rank
    // walk below does not depend on them, so their latency hides behind it)
    int xn = ld(r0), aln = ld(r0 - G), rln = ld(r0 - H), agn = ld(r0 + H + 1), rgn = ld(r0 + G + 1);
    for (int r = r0; r < r1; ++r) {
        const int x = xn, add_lead = aln, rem_lead = rln, add_lag = agn, rem_lag = rgn;
        xn = ld(r + 1);
        aln = ld(r + 1 - G);
        rln = ld(r + 1 - H);
        agn = ld(r + H + 2);
        rgn = ld(r + G + 2);
        while (below > k) { // restore below <= k < below + hist[v]
            --v;
            below -= h[v * 64];
        }
        int hv = h[v * 64];
        while (below + hv <= k) {
            below += hv;
            ++v;
            hv = h[v * 64];
        }
        const bool inside = r >= H && r < rows - H; // cfar.cpp:82
        mo[(size_t)r * cols] = (inside && x >= (int)s_minx[v]) ? 1 : 0;
        if (to)
            to[(size_t)r * cols] = inside ? s_thr[v] : 0.0f;
        // four bins change; read them together and write each one's final count (bins named twice get the same
        // value from both writes)
        {
            const int c0 = h[rem_lead * 64], c1 = h[add_lead * 64], c2 = h[rem_lag * 64], c3 = h[add_lag * 64];
            auto net = [&](int val) { return (val == add_lead) + (val == add_lag) - (val == rem_lead) - (val == rem_lag); };
            h[rem_lead * 64] = (uint8_t)(c0 + net(rem_lead));
            h[add_lead * 64] = (uint8_t)(c1 + net(add_lead));
            h[rem_lag * 64] = (uint8_t)(c2 + net(rem_lag));
            h[add_lag * 64] = (uint8_t)(c3 + net(add_lag));
        }
        below += (add_lead < v) - (rem_lead < v) + (add_lag < v) - (rem_lag < v);
    }
}

// ---------------------------------------------------------------------------------------------
// OS-CFAR behind an intensity gate (feature_extraction.py:223-224: `peaks = detector.detect(img, alg); peaks &= img >
// threshold`, the only way bruce_slam runs any CFAR): candidates only.
// A pixel at or below the gate is 0 whatever its window holds, and on a sonar image the gate (65 of 255) leaves about
// one pixel in a hundred.  So the order statistic is not tracked for every pixel: a workgroup stages a tile with its
// window halo in LDS, the waves compact the pixels above the gate into lists (wave prefix sums), and one lane per
// CANDIDATE counts its window:   x > tau * train[k]  <=>  train[k] <= L[x]  <=>  at least k + 1 of the 2T training
// cells are <= L[x],   L[x] = the largest value v with x > tau * v in the reference's double expression (a 256-entry
// table built on the host like every other decision table here; -1 = x can never fire).  2T byte reads and compares per
// candidate instead of a sliding 256-bin histogram and a rank walk per pixel; everything else of the tile is a streaming
// copy (1 B in + 1 B out per pixel + the halo).  Same masks as cfar_u8_os / the reference (tests: every OS case with a
// gate runs through this kernel).  Replaces cfar.cpp:76-96 + feature_extraction.py:224.
// ---------------------------------------------------------------------------------------------
#define OSG_TR 128 // tile rows
#define OSG_TC 128 // tile columns (bytes per staged row)
#define OSG_LIST 192 // candidates a wave collects before it takes 64 of them

struct CfarOsGateTab {
    int16_t L[256]; // pixel value x -> largest v with x > tau * v and x above the gate; -1: never fires
    int xc;         // smallest x with L[x] >= 0 (L grows with x: "can fire at all" is one threshold); 257: none
    // PREF (no gate, or a low one: round 6): the level l0 of the pre-filter and what the kernel needs of it
    int x_hi;       // smallest x with L[x] > l0 (257: none)
    int c0;         // l0 + 1: a training cell counts as "above" when it is >= c0
    int m_le;       // 2T - (k + 1): at most that many cells above l0 <=> at least k + 1 cells <= l0
};

// cfar.os() WITHOUT the gate (the drop-in's plain `cfar.os`, `feature.yaml alg: OS` with a low threshold): rounds 2-5 ran the
// sliding 256-bin histogram for every pixel (cfar_u8_os: 4 % of HBM).  PREF turns the image's own statistics into the gate:
//     x fires  <=>  at least k + 1 training cells are <= L[x]  =>  (L[x] > l0)  or  (at least k + 1 cells are <= l0)
// for ANY level l0 (counts grow with the level), so a pixel is a candidate iff x >= x_hi, or x can fire at all and its window
// holds k + 1 cells <= l0.  The second test is one sliding COUNT per column against a constant -- four columns per lane in
// packed bytes (~9 VALU operations per pixel) -- and on sonar images with l0 = L[80] it passes one window in ten thousand
// (background cells are Rayleigh around 20; a cell <= 8 is rare, eleven of them in one window rarer).  Candidates then take
// the exact per-candidate count of the gated kernel.  Exact for every image: a pixel the pre-filter drops cannot fire.
// PREF: 0 = gate only; 1 = pre-filter, any window (the count slides down the column: 16 dependent LDS round trips per thread);
// 2 = pre-filter for train_hs = 20 (the shipped Ntc = 40): the 35 + 35 rows a thread's 16 pixels need are read at once and
// the 16 window counts are differences of two running sums kept in registers -- no dependent chain (the kernel's occupancy is
// set by its LDS, two waves per SIMD: registers are free)
template <bool V16, int PREF> // V16: rows and pointers are 16-byte aligned -- the tile streams through in 16-byte pieces
__global__ __launch_bounds__(256) void cfar_u8_os_gated(const uint8_t *__restrict__ img, uint8_t *__restrict__ mask, int rows,
                                                        int cols, int n_frames, int T, int G, int k, int tiles_y, int tiles_x,
                                                        CfarOsGateTab tab)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t osg_raw[];
    const int H = T + G, SR = OSG_TR + 2 * H; // staged rows
    // (round 6: the tile's output and the pre-filter's flags are BITS in LDS -- 2 KB each instead of 16 KB of bytes: the staged
    //  input alone decides how many workgroups share a CU, 5 instead of 3 for the shipped window)
    uint8_t *s_in = osg_raw;                                  // [SR][OSG_TC]
    uint32_t *s_obits = reinterpret_cast<uint32_t *>(s_in + (size_t)SR * OSG_TC);            // [OSG_TR][OSG_TC / 32]: the mask
    unsigned short *s_list = reinterpret_cast<unsigned short *>(s_obits + OSG_TR * (OSG_TC / 32)); // [4][OSG_LIST + 64 * 4]
    uint32_t *s_flag = reinterpret_cast<uint32_t *>(s_list + 4 * (OSG_LIST + 256)); // PREF: [8 row groups][32 column words][2]: a nibble per row
    __shared__ short s_L[256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tpf = tiles_y * tiles_x;
    const long long f = blockIdx.x / tpf;
    const int tw = blockIdx.x % tpf, ty = tw / tiles_x, tx = tw % tiles_x;
    if (f >= n_frames)
        return;
    s_L[tid] = tab.L[tid];
    const int r0 = ty * OSG_TR, c0 = tx * OSG_TC;
    const int tr = min(OSG_TR, rows - r0), tc = min(OSG_TC, cols - c0); // tc is a multiple of 4
    const uint8_t *__restrict__ in = img + (size_t)f * rows * cols;
    uint8_t *__restrict__ out = mask + (size_t)f * rows * cols;
    // stage rows r0 - H .. r0 + tr + H - 1 (clamped into the image: rows that need a clamped cell are border rows,
    // whose output is 0 anyway, cfar.cpp:82), 4 bytes per thread and load
    const int wpr = OSG_TC / 4; // dwords per staged row
    if constexpr (V16) {
        constexpr int QPR = OSG_TC / 16; // 16-byte pieces per staged row
        const int n16 = (tr + 2 * H) * QPR;
        for (int i0 = tid; i0 < n16; i0 += 4 * 256) { // four loads in flight per thread
            uint4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * 256, rr = i / QPR, cw = i - rr * QPR;
                const int gr = min(max(r0 - H + rr, 0), rows - 1);
                v[u] = make_uint4(0u, 0u, 0u, 0u);
                if (i < n16 && 16 * cw < tc)
                    v[u] = *reinterpret_cast<const uint4 *>(in + (size_t)gr * cols + c0 + 16 * cw);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i0 + u * 256 < n16)
                    reinterpret_cast<uint4 *>(s_in)[i0 + u * 256] = v[u];
        }
    } else {
        for (int i = tid; i < (tr + 2 * H) * wpr; i += 256) {
            const int rr = i / wpr, cw = i - rr * wpr;
            const int gr = min(max(r0 - H + rr, 0), rows - 1);
            uint32_t v = 0u;
            if (4 * cw < tc)
                v = *reinterpret_cast<const uint32_t *>(in + (size_t)gr * cols + c0 + 4 * cw);
            reinterpret_cast<uint32_t *>(s_in)[rr * wpr + cw] = v;
        }
    }
    for (int i = tid; i < OSG_TR * (OSG_TC / 32); i += 256)
        s_obits[i] = 0u;
    __syncthreads();
    if constexpr (PREF != 0) {
        // per pixel: do at least k + 1 of its 2T training cells lie at or below l0?  Thread = (4 columns, 16 rows): the count
        // of cells ABOVE l0 (>= c0) is built once from the 2T rows of its first pixel and then slides down -- four cells in,
        // four out per row, every test on four packed bytes.  Byte-wise v >= c without a carry between bytes: the low
        // seven bits by an addition that cannot leave the byte, bit 7 of the cell decides the rest.
        const int c0 = tab.c0;                                 // 1 .. 255
        const bool hi = c0 > 128;
        const uint32_t kadd = (uint32_t)(128 - (hi ? c0 - 128 : c0)) * 0x01010101u; // low7(v) + kadd has bit 7 <=> low7(v) >= low7(c)
        auto above = [&](uint32_t v) -> uint32_t {             // 0x01 per byte that is >= c0
            const uint32_t g7 = (v & 0x7F7F7F7Fu) + kadd;
            const uint32_t ge = hi ? (g7 & v) : (g7 | v);      // (c0 == 128: kadd = 0, g7 has no bit 7, ge = bit 7 of v)
            return (ge >> 7) & 0x01010101u;
        };
        const int cw = tid & 31, sg = tid >> 5, rr0 = 16 * sg;
        const uint32_t madd = (uint32_t)(127 - min(tab.m_le, 127)) * 0x01010101u; // cnt + madd has bit 7 <=> cnt > m_le
        if constexpr (PREF == 2) {
            if (4 * cw < tc && rr0 < tr) {
                constexpr int TT = 20, NR = 16 + TT - 1;
                const uint32_t *lead = reinterpret_cast<const uint32_t *>(s_in) + rr0 * wpr + cw;   // staged rows rr0 ..
                const uint32_t *lag = lead + (H + G + 1) * wpr;                                     // ... and rr0 + H + G + 1 ..
                uint32_t pl[NR + 1], pg[NR + 1]; // running sums of `above` down the two row ranges (<= 35 per byte)
                pl[0] = pg[0] = 0u;
#pragma unroll
                for (int i = 0; i < NR; ++i) {
                    pl[i + 1] = pl[i] + above(lead[i * wpr]);
                    pg[i + 1] = pg[i] + above(lag[i * wpr]);
                }
                uint32_t fw[2] = {0u, 0u}; // a nibble per row: bit b = column 4 cw + b holds k + 1 cells <= l0
#pragma unroll
                for (int d = 0; d < 16; ++d) {
                    const uint32_t cnt = (pl[d + TT] - pl[d]) + (pg[d + TT] - pg[d]); // (per byte: no borrow, the sums grow)
                    const uint32_t f7 = ~(cnt + madd) & 0x80808080u;
                    fw[d >> 3] |= ((((f7 >> 7) * 0x01020408u) >> 24) & 15u) << (4 * (d & 7));
                }
                s_flag[(sg * 32 + cw) * 2] = fw[0];
                s_flag[(sg * 32 + cw) * 2 + 1] = fw[1];
            }
        } else
        if (4 * cw < tc && rr0 < tr) {
            const uint32_t *col = reinterpret_cast<const uint32_t *>(s_in) + cw; // staged row q: col[q * wpr]
            uint32_t cnt = 0u;                                  // per byte: training cells above l0 (<= 2T <= 255)
            for (int i = 0; i < T; ++i)
                cnt += above(col[(rr0 + i) * wpr]) + above(col[(rr0 + H + G + 1 + i) * wpr]);
            const int rr1 = min(rr0 + 16, tr);
            uint32_t fw[2] = {0u, 0u};
            for (int rr = rr0; rr < rr1; ++rr) {
                // (2T <= 127 on this path: the launcher keeps taller windows on the histogram kernel)
                const uint32_t f7 = ~(cnt + madd) & 0x80808080u; // bit 7: at most m_le cells above l0
                const int d = rr - rr0;
                fw[d >> 3] |= ((((f7 >> 7) * 0x01020408u) >> 24) & 15u) << (4 * (d & 7));
                cnt += above(col[(rr + T) * wpr]) - above(col[rr * wpr]) + above(col[(rr + 2 * H + 1) * wpr]) -
                       above(col[(rr + H + G + 1) * wpr]);
            }
            s_flag[(sg * 32 + cw) * 2] = fw[0];
            s_flag[(sg * 32 + cw) * 2 + 1] = fw[1];
        }
        __syncthreads();
    }
    unsigned short *wl = s_list + wave * (OSG_LIST + 256);
    int nl = 0; // candidates in this wave's list (wave-uniform)
    auto take = [&](int n_take) { // one lane per candidate, the last n_take of the list: count its window
        const bool on = lane < n_take;
        const int e = on ? (int)wl[nl - n_take + lane] : 0;
        const int rr = e >> 7, cc = e & 127; // tile row / column
        const uint8_t *col = s_in + (size_t)rr * OSG_TC + cc; // row rr of the staged tile = image row r - H
        const int Lx = (int)s_L[col[(size_t)H * OSG_TC]];
        int cnt = 0;
        for (int i = 0; i < T; ++i) {
            cnt += (int)col[(size_t)i * OSG_TC] <= Lx;                   // lead: rows r - H .. r - G - 1
            cnt += (int)col[(size_t)(H + G + 1 + i) * OSG_TC] <= Lx;     // lag:  rows r + G + 1 .. r + H
        }
        if (on && cnt > k)
            atomicOr(&s_obits[rr * (OSG_TC / 32) + (cc >> 5)], 1u << (cc & 31));
        nl -= n_take;
    };
    const int xc = tab.xc; // "x can fire at all" is one threshold: x >= xc
    const uint32_t c7 = (uint32_t)(xc & 127) * 0x01010101u;
    for (int rr2 = 2 * wave; rr2 < tr && xc <= 255; rr2 += 8) { // two rows of 128 columns per pass: lane -> (row, 4 columns)
        const int rr = rr2 + (lane >> 5), cw = lane & 31, r = r0 + rr;
        const bool valid = rr < tr && 4 * cw < tc && r >= H && r < rows - H; // (border rows stay 0, cfar.cpp:82)
        const uint32_t px = valid ? reinterpret_cast<const uint32_t *>(s_in)[(rr + H) * wpr + cw] : 0u;
        // per byte: x >= xc.  Low seven bits by a borrow-free subtraction, bit 7 of the pixel decides the rest
        const uint32_t t7 = ((px | 0x80808080u) - c7) & 0x80808080u; // bit 7 of a byte: its low seven bits are >= xc's
        const uint32_t ge = (xc >= 128) ? (px & t7) : ((px | t7) & 0x80808080u);
        unsigned cand = valid ? ((((ge >> 7) & 0x01010101u) * 0x01020408u) >> 24) & 15u : 0u; // bit b: column 4 cw + b
        if constexpr (PREF != 0) { // x >= x_hi, or x >= xc and the window holds k + 1 cells <= l0
            const int xh = tab.x_hi;
            const uint32_t h7 = (uint32_t)(xh & 127) * 0x01010101u;
            const uint32_t u7 = ((px | 0x80808080u) - h7) & 0x80808080u;
            const uint32_t geh = xh > 255 ? 0u : ((xh >= 128) ? (px & u7) : ((px | u7) & 0x80808080u));
            const unsigned hi_n = ((((geh >> 7) & 0x01010101u) * 0x01020408u) >> 24) & 15u;
            const unsigned fl_n = valid ? (s_flag[((rr >> 4) * 32 + cw) * 2 + ((rr >> 3) & 1)] >> (4 * (rr & 7))) & 15u : 0u;
            cand = valid ? (hi_n | (cand & fl_n)) : 0u;
        }
        const int pc = __popc(cand);
        if (!__ballot(pc != 0))
            continue;
        int incl = pc; // wave prefix sum of the candidate counts (<= 4 per lane)
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d);
            if (lane >= d)
                incl += o;
        }
        const int total = __builtin_amdgcn_readlane(incl, 63);
        int pos = nl + incl - pc;
        unsigned cb = cand;
        while (cb) {
            const int b = __ffs((int)cb) - 1;
            wl[pos++] = (unsigned short)((rr << 7) | (4 * cw + b));
            cb &= cb - 1u;
        }
        nl += total;
        while (nl >= 64)
            take(64);
    }
    if (nl > 0)
        take(nl);
    __syncthreads();
    if constexpr (V16) {
        constexpr int QPR = OSG_TC / 16;
        for (int i = tid; i < tr * QPR; i += 256) {
            const int rr = i / QPR, cw = i - rr * QPR;
            if (16 * cw < tc) {
                const uint32_t b16 = (s_obits[rr * (OSG_TC / 32) + (cw >> 1)] >> (16 * (cw & 1))) & 0xFFFFu; // 16 columns
                auto spread = [](uint32_t n4) { return (n4 * 0x00204081u) & 0x01010101u; };                // nibble -> 4 bytes 0 / 1
                *reinterpret_cast<uint4 *>(out + (size_t)(r0 + rr) * cols + c0 + 16 * cw) =
                    make_uint4(spread(b16 & 15u), spread((b16 >> 4) & 15u), spread((b16 >> 8) & 15u), spread(b16 >> 12));
            }
        }
    } else {
        for (int i = tid; i < tr * wpr; i += 256) {
            const int rr = i / wpr, cw = i - rr * wpr;
            if (4 * cw < tc) {
                const uint32_t n4 = (s_obits[rr * (OSG_TC / 32) + (cw >> 3)] >> (4 * (cw & 7))) & 15u;
                *reinterpret_cast<uint32_t *>(out + (size_t)(r0 + rr) * cols + c0 + 4 * cw) = (n4 * 0x00204081u) & 0x01010101u;
            }
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

template <int T, int G, int D, bool BITS, bool THR = false>
static void launch_ring(sfe_ctx *ctx, int alg, const uint8_t *d_img, uint8_t *d_mask, int rows, int cols,
                        int n_frames, int groups, int tiles, long long out_frame_bytes, const CfarLut &lut,
                        float *d_thr = nullptr, CfarThrArith ta = CfarThrArith{0.0, 0.0, 1.0, 0})
{
    const int chunks = ((cols >> 2) + 63) / 64;
    const long long bpf = ((long long)tiles * chunks + 3) / 4;             // workgroups per frame
    const unsigned blocks = (unsigned)((((long long)n_frames + 7) / 8) * 8 * bpf); // frames padded to the 8 XCDs
    static const int full_last = getenv("SFE_CFAR_FULL_LAST_TILE") ? 1 : 0; // A/B: the last tile as long as the others (rounds 1-4)
    if (alg == SFE_CFAR_SOCA)
        hipLaunchKernelGGL((cfar_u8_ring<T, G, SFE_CFAR_SOCA, D, BITS, THR>), dim3(blocks), dim3(256), 0, ctx->stream,
                           d_img, d_mask, rows, cols, n_frames, groups, tiles, chunks, out_frame_bytes, lut, d_thr, ta, full_last);
    else if (alg == SFE_CFAR_GOCA)
        hipLaunchKernelGGL((cfar_u8_ring<T, G, SFE_CFAR_GOCA, D, BITS, THR>), dim3(blocks), dim3(256), 0, ctx->stream,
                           d_img, d_mask, rows, cols, n_frames, groups, tiles, chunks, out_frame_bytes, lut, d_thr, ta, full_last);
    else
        hipLaunchKernelGGL((cfar_u8_ring<T, G, SFE_CFAR_CA, D, BITS, THR>), dim3(blocks), dim3(256), 0, ctx->stream,
                           d_img, d_mask, rows, cols, n_frames, groups, tiles, chunks, out_frame_bytes, lut, d_thr, ta, full_last);
}

// cfar_thr_arith checked against the reference expression for every window sum of (alg, T, tau): on = 1 when each of them
// agrees bit by bit (cached per context)
static CfarThrArith thr_arith_checked(sfe_ctx *ctx, int alg, int T, double tau)
{
    const double d = (alg == SFE_CFAR_CA) ? 2.0 * T : (double)T;
    CfarThrArith ta{tau, 1.0 / d, d, 1};
    if (ctx->tha_alg == alg && ctx->tha_T == T && ctx->tha_tau == tau) {
        ta.on = ctx->tha_on;
        return ta;
    }
    const int smax = (alg == SFE_CFAR_CA) ? 255 * 2 * T : 255 * T;
    for (int sv = 0; sv <= smax && ta.on; ++sv) {
        const float sf = (float)sv;
        const float want = (float)((alg == SFE_CFAR_CA) ? tau * (double)sf / (2.0 * T) : tau * (double)sf / T);
        const float got = cfar_thr_arith(ta, (uint32_t)sv);
        if (__builtin_memcmp(&want, &got, sizeof want) != 0)
            ta.on = 0; // (never seen: the kernels then read the table)
    }
    if (getenv("SFE_CFAR_THR_TABLE"))
        ta.on = 0; // A/B
    ctx->tha_alg = alg;
    ctx->tha_T = T;
    ctx->tha_tau = tau;
    ctx->tha_on = ta.on;
    return ta;
}

// R-row groups per tile.  Measured on MI355X (tools/cfar_sweep.py, 1024 frames of 1024x512, XCD-aware
// map + alternating march direction):  1 group/tile 5.1 TB/s with FETCH = 1.30x the image bytes,
// 2 groups 5.1 TB/s with 1.13x, 4 groups 4.9 TB/s with 1.07x, whole column 3.1 TB/s.  Short tiles win
// on time (the kernel is bound by each wave's serial row march, more independent waves hide it);
// 2 groups keep that speed and most of the 2*(T+G) halo rows a tile re-reads are L2 hits.
static int default_groups(const sfe_ctx *, int rows, int, int, int R) { return rows >= 2 * R ? 2 : 1; }

static int launch_os_hist(sfe_ctx *ctx, const uint8_t *d_img, int n_frames, int rows, int cols, int T, int G, int k,
                          double tau, int intensity_thr, uint8_t *d_mask, float *d_thr)
{
    CfarOsTab tab;
    for (int v = 0; v < 256; ++v) {
        const double t = tau * (double)(float)v; // cfar.cpp:92 / :186
        tab.thr[v] = (float)t;
        int mx = 256;
        for (int x = 255; x >= 0; --x) // (double)x > t is monotone in x
            if ((double)(float)x > t && !(intensity_thr >= 0 && x <= intensity_thr))
                mx = x;
            else
                break;
        tab.min_x[v] = (uint16_t)mx;
    }
    const int tile_rows = std::min(rows, std::max(256, 8 * T));
    const int tiles = (rows + tile_rows - 1) / tile_rows;
    const int chunks = (cols + 63) / 64;
    const long long blocks = (long long)n_frames * tiles * chunks;
    hipLaunchKernelGGL(cfar_u8_os, dim3((unsigned)blocks), dim3(64), 0, ctx->stream, d_img, d_mask, d_thr, rows, cols,
                       n_frames, T, G, k, tile_rows, tiles, chunks, tab);
    return 0;
}

// pref: the pre-filtered form for a missing or low gate (see the kernel); *applied (nullable) = false when it does not apply
// (no level to filter on, a window the packed counters do not hold) and nothing was launched
static int launch_os_gated(sfe_ctx *ctx, const uint8_t *d_img, int n_frames, int rows, int cols, int T, int G, int k, double tau,
                           int intensity_thr, uint8_t *d_mask, bool pref = false, bool *applied = nullptr)
{
    CfarOsGateTab tab;
    for (int x = 0; x < 256; ++x) {
        int L = -1;
        if (!(intensity_thr >= 0 && x <= intensity_thr))
            for (int v = 0; v < 256; ++v) { // (double)x > tau * v is monotone in v: the largest v that still holds
                const double t = tau * (double)(float)v; // cfar.cpp:92
                if ((double)(float)x > t)
                    L = v;
                else
                    break;
            }
        tab.L[x] = (int16_t)L;
    }
    tab.xc = 257;
    for (int x = 255; x >= 0; --x)
        if (tab.L[x] >= 0)
            tab.xc = x;
    tab.x_hi = 257;
    tab.c0 = 1;
    tab.m_le = 0;
    if (pref) {
        // the level: what a pixel of a third of full scale is compared with (L[80]; SFE_CFAR_OS_PREF_X moves it).  Any level is
        // exact; this one keeps both kinds of candidates rare on sonar images
        const int xs = std::min(255, std::max(tab.xc, getenv("SFE_CFAR_OS_PREF_X") ? atoi(getenv("SFE_CFAR_OS_PREF_X")) : 80));
        const int l0 = xs <= 255 ? tab.L[xs] : -1;
        const bool ok = l0 >= 0 && l0 < 255 && 2 * T <= 127 && k + 1 <= 2 * T;
        if (applied)
            *applied = ok;
        if (!ok)
            return 0;
        tab.c0 = l0 + 1;
        tab.m_le = 2 * T - (k + 1);
        for (int x = 255; x >= 0; --x)
            if (tab.L[x] > l0)
                tab.x_hi = x;
    }
    const int tiles_y = (rows + OSG_TR - 1) / OSG_TR, tiles_x = (cols + OSG_TC - 1) / OSG_TC;
    const size_t smem = (size_t)(OSG_TR + 2 * (T + G)) * OSG_TC + (size_t)OSG_TR * (OSG_TC / 8) + sizeof(unsigned short) * 4 * (OSG_LIST + 256) +
                        (pref ? (size_t)8 * 32 * 2 * 4 : 0);
    const bool v16 = cols % 16 == 0 && (((uintptr_t)d_img | (uintptr_t)d_mask) & 15) == 0 && !getenv("SFE_CFAR_OSG_V4");
    const int pv = !pref ? 0 : (T == 20 && !getenv("SFE_CFAR_OS_PREF_SLIDE")) ? 2 : 1;
    auto kernel = pv == 2 ? (v16 ? cfar_u8_os_gated<true, 2> : cfar_u8_os_gated<false, 2>)
                  : pv == 1 ? (v16 ? cfar_u8_os_gated<true, 1> : cfar_u8_os_gated<false, 1>)
                            : (v16 ? cfar_u8_os_gated<true, 0> : cfar_u8_os_gated<false, 0>);
    SFE_HIP(ctx, hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(kernel, dim3((unsigned)((long long)n_frames * tiles_y * tiles_x)), dim3(256), smem, ctx->stream,
                       d_img, d_mask, rows, cols, n_frames, T, G, k, tiles_y, tiles_x, tab);
    return 0;
}

// d_bits != nullptr: the ring kernel writes the bit-packed detections there (BITS variant) and d_mask is not used;
// the caller has checked that the ring kernel applies (ring_bits_applicable).
static int cfar_u8_dev(sfe_ctx *ctx, const uint8_t *d_img, int n_frames, int rows, int cols, int alg,
                       int T, int G, int k, double tau, int intensity_thr, uint8_t *d_mask, float *d_thr,
                       uint32_t *d_bits = nullptr)
{
    if (d_bits)
        d_mask = reinterpret_cast<uint8_t *>(d_bits);
    SFE_ARG(ctx, d_img && d_mask);
    SFE_ARG(ctx, n_frames >= 0 && rows >= 0 && cols >= 0);
    SFE_ARG(ctx, alg >= SFE_CFAR_CA && alg <= SFE_CFAR_OS);
    SFE_ARG(ctx, T >= 1 && G >= 0);
    if (alg == SFE_CFAR_OS)
        SFE_ARG(ctx, k >= 0 && k < 2 * T);
    if (n_frames == 0 || rows == 0 || cols == 0)
        return 0;
    CfarLut lut;
    // register-ring kernel: instantiated for the shipped window (Ntc 40, Ngc 10 -> 20, 5) and for the other windows
    // the reference's feature.yaml comments and tests go through: (32, 8), (20, 4), (16, 2)
    const bool ring_window = (T == 20 && G == 5) || (T == 16 && G == 4) || (T == 10 && G == 2) || (T == 8 && G == 1);
    const int ringR = 2 * (T + G) + 2;
    // (with a threshold map: when it can be computed, cfar_thr_arith, and not next to the bit-stream output)
    const CfarThrArith thr_ta = (d_thr && alg != SFE_CFAR_OS) ? thr_arith_checked(ctx, alg, T, tau) : CfarThrArith{0.0, 0.0, 1.0, 0};
    static const bool no_ring_thr = getenv("SFE_CFAR_NO_RING_THR") != nullptr; // A/B
    bool ring = (alg != SFE_CFAR_OS) && (!d_thr || (thr_ta.on && !d_bits && !no_ring_thr && reinterpret_cast<uintptr_t>(d_thr) % 16 == 0)) && (cols % 4 == 0) && cols >= 256 && rows >= ringR && (size_t)rows * cols < (1u << 30) &&
                ctx->cfar_variant != 1 &&
                ((reinterpret_cast<uintptr_t>(d_img) | reinterpret_cast<uintptr_t>(d_mask)) % 4 == 0) &&
                ring_window && build_lut(alg, T, tau, intensity_thr, &lut);
    if (ctx->cfar_variant >= 2 && !ring)
        return sfe_set_err(ctx, SFE_ERR_ARG, "ring CFAR kernel forced but not applicable to this call");
    // every other window / the threshold maps: sliding-sum kernel (run-time window), then the OS histogram kernel;
    // what is left (odd widths, unaligned buffers, windows beyond the 16-bit sums) takes the generic kernel
    const bool aligned = (cols % 4 == 0) && (size_t)rows * cols < (1u << 30) &&
                         ((reinterpret_cast<uintptr_t>(d_img) | reinterpret_cast<uintptr_t>(d_mask)) % 4 == 0) &&
                         (!d_thr || reinterpret_cast<uintptr_t>(d_thr) % 16 == 0);
    const bool slide = !ring && alg != SFE_CFAR_OS && aligned && ctx->cfar_variant != 1 &&
                       build_lut(alg, T, tau, intensity_thr, &lut);
    const bool os_hist = !ring && alg == SFE_CFAR_OS && ctx->cfar_variant != 1 && 2 * T <= 255 &&
                         (size_t)rows * cols < (1u << 30);
    if (ring) {
        const int R = ringR;
        int groups = ctx->cfar_tile_rows > 0 ? std::max(1, std::min(ctx->cfar_tile_rows / R, rows / R))
                                             : std::max(1, std::min(104 / R, rows / R)); // ~104-row tiles (see default_groups)
        if (T == 20)
            groups = ctx->cfar_tile_rows > 0 ? groups : default_groups(ctx, rows, cols, n_frames, R);
        const int tiles = (rows + groups * R - 1) / (groups * R);
        const long long fb = (long long)rows * cols;
        if (d_bits) {
            const long long bb = (fb / 32 + 1) * 4; // bytes per frame of the bit stream: one pad word (sfe_remap.hip)
            if (T == 20)
                launch_ring<20, 5, 4, true>(ctx, alg, d_img, d_mask, rows, cols, n_frames, groups, tiles, bb, lut);
            else if (T == 16)
                launch_ring<16, 4, 6, true>(ctx, alg, d_img, d_mask, rows, cols, n_frames, groups, tiles, bb, lut);
            else if (T == 10)
                launch_ring<10, 2, 13, true>(ctx, alg, d_img, d_mask, rows, cols, n_frames, groups, tiles, bb, lut);
            else
                launch_ring<8, 1, 5, true>(ctx, alg, d_img, d_mask, rows, cols, n_frames, groups, tiles, bb, lut);
        } else if (d_thr) {
            if (T == 20)
                launch_ring<20, 5, 4, false, true>(ctx, alg, d_img, d_mask, rows, cols, n_frames, groups, tiles, fb, lut, d_thr, thr_ta);
            else if (T == 16)
                launch_ring<16, 4, 6, false, true>(ctx, alg, d_img, d_mask, rows, cols, n_frames, groups, tiles, fb, lut, d_thr, thr_ta);
            else if (T == 10)
                launch_ring<10, 2, 13, false, true>(ctx, alg, d_img, d_mask, rows, cols, n_frames, groups, tiles, fb, lut, d_thr, thr_ta);
            else
                launch_ring<8, 1, 5, false, true>(ctx, alg, d_img, d_mask, rows, cols, n_frames, groups, tiles, fb, lut, d_thr, thr_ta);
        } else if (T == 20 && ctx->cfar_variant == 3)
            launch_ring<20, 5, 13, false>(ctx, alg, d_img, d_mask, rows, cols, n_frames, groups, tiles, fb, lut);
        else if (T == 20)
            launch_ring<20, 5, 4, false>(ctx, alg, d_img, d_mask, rows, cols, n_frames, groups, tiles, fb, lut);
        else if (T == 16)
            launch_ring<16, 4, 6, false>(ctx, alg, d_img, d_mask, rows, cols, n_frames, groups, tiles, fb, lut);
        else if (T == 10)
            launch_ring<10, 2, 13, false>(ctx, alg, d_img, d_mask, rows, cols, n_frames, groups, tiles, fb, lut);
        else
            launch_ring<8, 1, 5, false>(ctx, alg, d_img, d_mask, rows, cols, n_frames, groups, tiles, fb, lut);
    } else if (d_bits) {
        return sfe_set_err(ctx, SFE_ERR_ARG, "internal: bit-packed CFAR output asked of a non-ring window");
    } else if (slide) {
        const float *d_tab = nullptr;
        if (d_thr) { // thr = (float)(tau * s / T) per integer window sum s, with the reference's double expression
            const int smax = (alg == SFE_CFAR_CA) ? 255 * 2 * T : 255 * T;
            float *tab = (float *)sfe_scratch(ctx, 37, sizeof(float) * (size_t)(smax + 1));
            if (!tab)
                return SFE_ERR_HIP;
            if (!(ctx->thr_tab_alg == alg && ctx->thr_tab_T == T && ctx->thr_tab_tau == tau && ctx->thr_tab_ptr == tab)) {
                float *h = (float *)sfe_pinned_begin(ctx, sizeof(float) * (size_t)(smax + 1));
                if (!h)
                    return SFE_ERR_HIP;
                for (int sv = 0; sv <= smax; ++sv) {
                    const float sf = (float)sv;
                    h[sv] = (float)((alg == SFE_CFAR_CA) ? tau * (double)sf / (2.0 * T) : tau * (double)sf / T);
                }
                SFE_HIP(ctx, hipMemcpyAsync(tab, h, sizeof(float) * (size_t)(smax + 1), hipMemcpyHostToDevice, ctx->stream));
                if (int rc = sfe_pinned_end(ctx, ctx->stream))
                    return rc;
                ctx->thr_tab_alg = alg;
                ctx->thr_tab_T = T;
                ctx->thr_tab_tau = tau;
                ctx->thr_tab_ptr = tab;
            }
            d_tab = tab;
        }
        // long tiles: a tile starts with 2T loads per lane to build its first windows
        const int tile_rows = std::min(rows, std::max(128, 8 * T));
        const int tiles = (rows + tile_rows - 1) / tile_rows;
        const int chunks = std::max(1, ((cols >> 2) + 63) / 64);
        const long long bpf = ((long long)tiles * chunks + 3) / 4;
        const unsigned blocks = (unsigned)((((long long)n_frames + 7) / 8) * 8 * bpf);
        // window rows staged in LDS (R KiB per workgroup) unless the window is too tall for it
        const int R = 2 * (T + G) + 2;
        const size_t ring_bytes = (size_t)R * 1024;
        static const bool no_lds = getenv("SFE_CFAR_NO_LDS_RING") != nullptr; // A/B
        // beyond two workgroups per CU (R > 80 rows = 80 KiB) the ring starves the CU of waves and re-reading the four
        // rows through the caches is faster (measured (80, 20): 15 % of HBM with the LDS ring, 25 % without)
        const bool lds_ring = ring_bytes <= 80 * 1024 && !no_lds;
#define SLIDE_LAUNCH(A, THRB)                                                                                          \
    do {                                                                                                               \
        if (lds_ring) {                                                                                                \
            SFE_HIP(ctx, hipFuncSetAttribute((const void *)cfar_u8_slide_lds<A, THRB>,                                 \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)ring_bytes));            \
            hipLaunchKernelGGL((cfar_u8_slide_lds<A, THRB>), dim3(blocks), dim3(256), ring_bytes, ctx->stream, d_img,  \
                               d_mask, d_thr, d_tab, rows, cols, n_frames, T, G, tile_rows, tiles, chunks, lut, thr_ta); \
        } else {                                                                                                       \
            hipLaunchKernelGGL((cfar_u8_slide<A, THRB>), dim3(blocks), dim3(256), 0, ctx->stream, d_img, d_mask,       \
                               d_thr, d_tab, rows, cols, n_frames, T, G, tile_rows, tiles, chunks, lut, thr_ta);       \
        }                                                                                                              \
    } while (0)
        if (alg == SFE_CFAR_SOCA) {
            if (d_thr)
                SLIDE_LAUNCH(SFE_CFAR_SOCA, true);
            else
                SLIDE_LAUNCH(SFE_CFAR_SOCA, false);
        } else if (alg == SFE_CFAR_GOCA) {
            if (d_thr)
                SLIDE_LAUNCH(SFE_CFAR_GOCA, true);
            else
                SLIDE_LAUNCH(SFE_CFAR_GOCA, false);
        } else {
            if (d_thr)
                SLIDE_LAUNCH(SFE_CFAR_CA, true);
            else
                SLIDE_LAUNCH(SFE_CFAR_CA, false);
        }
#undef SLIDE_LAUNCH
    } else if (os_hist && !d_thr && aligned && (size_t)(OSG_TR + 2 * (T + G)) * OSG_TC <= 96 * 1024 &&
               intensity_thr >= (getenv("SFE_CFAR_OS_GATED_MIN") ? atoi(getenv("SFE_CFAR_OS_GATED_MIN")) : 40) &&
               !getenv("SFE_CFAR_NO_OS_GATED")) {
        // OS behind a gate: only the pixels above it are looked at (cfar_u8_os_gated).  It pays when the gate removes most
        // pixels -- measured on 512 sonar frames, (Ntc 40, Ngc 10, k 10): gate 65 0.65 ms against the histogram kernel's
        // 1.65 ms; gate 20 (four pixels in ten pass) 1.84 against 1.65 -- so a low gate keeps the histogram kernel
        // (feature.yaml ships 65; SFE_CFAR_OS_GATED_MIN moves the limit).
        if (int rc = launch_os_gated(ctx, d_img, n_frames, rows, cols, T, G, k, tau, intensity_thr, d_mask))
            return rc;
    } else if (os_hist) {
        // no gate, or a low one: the pre-filtered candidate kernel (round 6) where it applies, else the sliding histogram
        bool done = false;
        if (!d_thr && aligned && (size_t)(OSG_TR + 2 * (T + G)) * OSG_TC <= 96 * 1024 && !getenv("SFE_CFAR_NO_OS_PREF"))
            if (int rc = launch_os_gated(ctx, d_img, n_frames, rows, cols, T, G, k, tau, intensity_thr, d_mask, true, &done))
                return rc;
        if (!done)
        if (int rc = launch_os_hist(ctx, d_img, n_frames, rows, cols, T, G, k, tau, intensity_thr, d_mask, d_thr))
            return rc;
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

// the call the BITS ring kernel takes: a ring window, whole 32-bit words per row, no threshold map
static bool ring_bits_applicable(const sfe_ctx *ctx, const uint8_t *d_img, int rows, int cols, int alg, int T, int G)
{
    const bool ring_window = (T == 20 && G == 5) || (T == 16 && G == 4) || (T == 10 && G == 2) || (T == 8 && G == 1);
    static const bool off = getenv("SFE_CFAR_NO_BITS") != nullptr; // A/B: byte kernel + mask_pack
    return !off && ring_window && alg != SFE_CFAR_OS && cols % 32 == 0 && cols >= 256 && rows >= 2 * (T + G) + 2 &&
           (size_t)rows * cols < (1u << 30) && ctx->cfar_variant == 0 && reinterpret_cast<uintptr_t>(d_img) % 4 == 0;
}

extern "C" {

int sfe_cfar_u8_bits_batch_dev(sfe_ctx *ctx, const uint8_t *d_img, int n_frames, int rows, int cols, int alg,
                               int train_hs, int guard_hs, int k, double tau, int intensity_thr, uint32_t *d_bits)
{
    if (int rc = sfe_use(ctx))
        return rc;
    SFE_ARG(ctx, d_img && d_bits && n_frames >= 0 && rows >= 0 && cols >= 0);
    SFE_ARG(ctx, reinterpret_cast<uintptr_t>(d_bits) % 4 == 0);
    const long long px = (long long)rows * cols, wpf = (px + 31) / 32 + 1;
    if (n_frames == 0 || px == 0)
        return 0;
    CfarLut probe;
    if (ring_bits_applicable(ctx, d_img, rows, cols, alg, train_hs, guard_hs) && train_hs >= 1 &&
        build_lut(alg, train_hs, tau, intensity_thr, &probe))
        return cfar_u8_dev(ctx, d_img, n_frames, rows, cols, alg, train_hs, guard_hs, k, tau, intensity_thr, nullptr,
                           nullptr, d_bits);
    // every other window: the byte kernels into a scratch mask, a bounded number of frames at a time, then packed
    const int chunk = (int)std::max<long long>(1, std::min<long long>(n_frames, (256ll << 20) / px));
    uint8_t *d_tmp = (uint8_t *)sfe_scratch(ctx, 41, (size_t)chunk * px);
    if (!d_tmp)
        return SFE_ERR_HIP;
    for (int f0 = 0; f0 < n_frames; f0 += chunk) {
        const int nf = std::min(chunk, n_frames - f0);
        if (int rc = cfar_u8_dev(ctx, d_img + (size_t)f0 * px, nf, rows, cols, alg, train_hs, guard_hs, k, tau,
                                 intensity_thr, d_tmp, nullptr))
            return rc;
        if (int rc = sfe_mask_pack(ctx, d_tmp, nf, px, d_bits + (size_t)f0 * wpf, nullptr))
            return rc;
    }
    return 0;
}

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
