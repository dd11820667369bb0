// Polar -> Cartesian feature extraction on gfx950.
// Replaces feature_extraction.py:226,231-238: cv2.remap(INTER_LINEAR) of the detection mask,
// np.nonzero (row-major order) and the pixel -> metre conversion.
//
// cv2.remap semantics restated from OpenCV imgproc (see oracle/sonar_oracle.c, PARITY UNPINNED:
// OpenCV is not in the reference tree nor in this image): coordinates are quantised to 1/32 px
// with cvRound (half-to-even), the 2x2 taps use 15-bit fixed-point weights (table entry (0,0)
// is {32767,0,0,1} after OpenCV's sum fix-up), out = (sum w*v + 16384) >> 15, outside = 0.
//
// Layout in HBM: the float maps are decoded ONCE per geometry into one uint32 per Cartesian
// pixel: [31:10] linear index of the top-left tap in a (polar_rows+1) x (polar_cols+1) grid that
// is shifted by one so that -1 is representable, [9:0] = fy*32+fx; 0xFFFFFFFF = no tap inside the
// image.  That halves the per-frame map traffic (4 B instead of 8 B per Cartesian pixel), and a
// per-row [first,last) span skips the pixels outside the sonar fan.
#include "sfe_internal.h"
#include "sfe_cloudfilter.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <algorithm>

#define SFE_CODE_NONE 0xFFFFFFFFu

// Decode one Cartesian pixel.  `rcp` = ceil(2^32 / (pcols+1)) turns the divide of the packed
// linear index into one v_mul_hi (exact for lin < 2^22, checked at geometry creation).
__device__ __forceinline__ int remap_value(const uint8_t *__restrict__ src, int prows, int pcols, unsigned rcp,
                                           uint32_t code)
{
    const unsigned lin = code >> 10;
    const int fy = (int)((code >> 5) & 31u), fx = (int)(code & 31u);
    const unsigned q = __umulhi(lin, rcp);
    const int iy = (int)q - 1, ix = (int)(lin - q * (unsigned)(pcols + 1)) - 1;
    // branch-free taps: clamp the coordinates (always a valid address, all four loads in flight
    // together) and zero the out-of-image ones afterwards (BORDER_CONSTANT 0)
    const int ya = max(iy, 0), yb = min(iy + 1, prows - 1), xa = max(ix, 0), xb = min(ix + 1, pcols - 1);
    const uint8_t *ra = src + (size_t)ya * pcols, *rb = src + (size_t)yb * pcols;
    const int my0 = (iy >= 0) ? 0xff : 0, my1 = (iy + 1 < prows) ? 0xff : 0;
    const int mx0 = (ix >= 0) ? 0xff : 0, mx1 = (ix + 1 < pcols) ? 0xff : 0;
    const int v00 = ra[xa] & my0 & mx0;
    const int v01 = ra[xb] & my0 & mx1;
    const int v10 = rb[xa] & my1 & mx0;
    const int v11 = rb[xb] & my1 & mx1;
    if ((v00 | v01 | v10 | v11) == 0)
        return 0; // sparse detection masks: most taps are empty
    int w00 = (32 - fy) * (32 - fx) * 32, w01 = (32 - fy) * fx * 32;
    int w10 = fy * (32 - fx) * 32, w11 = fy * fx * 32;
    if ((fx | fy) == 0) {
        w00 = 32767;
        w11 = 1;
    }
    const int acc = w00 * v00 + w01 * v01 + w10 * v10 + w11 * v11;
    return (acc + 16384) >> 15; // <= 255 because the weights sum to 32768
}

// full uint8 remap (visualisation image / drop-in cv2.remap)
__global__ __launch_bounds__(256) void remap_u8_kernel(const uint8_t *__restrict__ src,
                                                       const uint32_t *__restrict__ code,
                                                       uint8_t *__restrict__ dst, int prows, int pcols,
                                                       unsigned rcp, long long n_cart, int n_frames)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_cart * n_frames)
        return;
    const long long f = i / n_cart, o = i % n_cart;
    const uint32_t c = code[o];
    dst[i] = (c == SFE_CODE_NONE) ? 0 : (uint8_t)remap_value(src + f * (long long)prows * pcols, prows, pcols, rcp, c);
}

// cv2.applyColorMap(cv2.remap(img, ...), cv2.COLORMAP_JET) in one pass (feature_extraction.py:226-228): the remapped grey
// value goes through a 256-entry BGR table on its way out, so the publishable bgr8 image is written once and the grey
// canvas never exists.  A thread owns 4 consecutive canvas pixels = 12 output bytes = three aligned dword stores.
// lut: 256 x (B | G << 8 | R << 16), staged in LDS.
__global__ __launch_bounds__(256) void remap_u8_lut_kernel(const uint8_t *__restrict__ src,
                                                           const uint32_t *__restrict__ code,
                                                           const uint32_t *__restrict__ lut, uint32_t *__restrict__ dst3,
                                                           int prows, int pcols, unsigned rcp, long long n_cart)
{
    __shared__ uint32_t s_lut[256];
    s_lut[threadIdx.x] = lut[threadIdx.x];
    __syncthreads();
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x; // group of 4 pixels
    const long long o = q * 4;
    if (o >= n_cart)
        return;
    uint32_t px[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint32_t v = 0;
        if (o + k < n_cart) {
            const uint32_t c = code[o + k];
            v = (c == SFE_CODE_NONE) ? 0u : (uint32_t)remap_value(src, prows, pcols, rcp, c);
        }
        px[k] = s_lut[v];
    }
    if (o + 4 <= n_cart) { // b0 g0 r0 b1 | g1 r1 b2 g2 | r2 b3 g3 r3
        dst3[q * 3] = px[0] | (px[1] << 24);
        dst3[q * 3 + 1] = (px[1] >> 8) | (px[2] << 16);
        dst3[q * 3 + 2] = (px[2] >> 16) | (px[3] << 8);
    } else {
        uint8_t *d = reinterpret_cast<uint8_t *>(dst3) + o * 3;
        for (int k = 0; o + k < n_cart; ++k) {
            d[3 * k] = (uint8_t)px[k];
            d[3 * k + 1] = (uint8_t)(px[k] >> 8);
            d[3 * k + 2] = (uint8_t)(px[k] >> 16);
        }
    }
}

// pass 0: pack the uint8 detection mask into bits (bit iy*pcols+ix of the frame's bit stream,
// LSB first) and note whether any byte is > 1 (then the binary shortcut of pass 1 is not valid).
__global__ __launch_bounds__(256) void mask_pack_kernel(const uint8_t *__restrict__ mask,
                                                        uint32_t *__restrict__ bits, int32_t *__restrict__ nonbinary,
                                                        long long px_per_frame, long long words_per_frame)
{
    const long long wi = (long long)blockIdx.x * 256 + threadIdx.x; // word index inside the frame
    const int f = blockIdx.y;
    if (wi >= words_per_frame)
        return;
    const uint8_t *__restrict__ src = mask + (long long)f * px_per_frame + wi * 32;
    const long long left = px_per_frame - wi * 32;
    uint32_t out = 0, big = 0;
    if (left >= 32 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
        const uint4 a = reinterpret_cast<const uint4 *>(src)[0], b = reinterpret_cast<const uint4 *>(src)[1];
        const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t nz = ((w[i] | ((w[i] & 0x7f7f7f7fu) + 0x7f7f7f7fu)) & 0x80808080u) >> 7;
            out |= (((nz * 0x00204081u) >> 21) & 0xfu) << (4 * i);
            big |= w[i] & 0xfefefefeu;
        }
    } else {
        for (int i = 0; i < 32 && i < left; ++i) {
            out |= (uint32_t)(src[i] != 0) << i;
            big |= src[i] & 0xfeu;
        }
    }
    bits[(long long)f * words_per_frame + wi] = out;
    if (big)
        nonbinary[f] = 1;
}

// pass 1: detection bits of the Cartesian canvas as 64-bit ballot words.
// Gathering the four taps of every Cartesian pixel straight from the byte mask is bound by the
// L1 tag pipeline (adjacent Cartesian pixels fall into different 128-byte lines of the polar
// image: ~1 lane per clock, 2.1 ms per 256 frames whatever the loop structure).  So the mask is
// bit-packed (64 KiB per 1024x512 frame) and each workgroup stages the polar rows its tile
// of the canvas needs -- a precomputed [ylo, yhi] range -- into LDS, where random bit reads cost
// a ds_read each.  Tile = 4 ballot words (256 columns) x EXTRACT_RG rows; a wave owns one word
// position and walks the rows, EXTRACT_U rows per batch so the code loads overlap.
// Workgroup -> (frame, row group, word group) with the word group fastest: XCD k keeps the word
// groups k (mod 8), i.e. a vertical strip of the canvas = 1/8 of the code table (L2-resident).
// Non-binary masks (values > 1, where cv2.remap's result depends on the values) take the
// general byte-gather path.
#define EXTRACT_RG 32
#define EXTRACT_U 4
__global__ __launch_bounds__(256) void extract_bits_kernel(const uint8_t *__restrict__ mask,
                                                           const uint32_t *__restrict__ bits,
                                                           const int32_t *__restrict__ nonbinary,
                                                           const uint32_t *__restrict__ code,
                                                           const int32_t *__restrict__ span,
                                                           const int32_t *__restrict__ tile_rows,
                                                           unsigned long long *__restrict__ bitmap, int prows,
                                                           int pcols, unsigned rcp, int crows, int ccols, int wpr,
                                                           int word_groups, int tiles_per_frame,
                                                           long long words_per_frame, int only_general, int n_frames)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t s_bits[];
    // grid = tiles x (frames or fewer): a workgroup takes its tile of frames f0, f0 + stride, ...  When the binary
    // frames were done by extract_gather_kernel (only_general) the launcher keeps the grid small -- a sonar batch
    // normally has no other frame, and 100 000 workgroups that only find that out cost 30 us per launch.
    const int tile = blockIdx.x % tiles_per_frame;
    for (int f = blockIdx.x / tiles_per_frame; f < n_frames; f += gridDim.x / tiles_per_frame) {
    if (only_general && nonbinary[f] == 0) // binary frames were done by extract_gather_kernel (block-uniform)
        continue;
    const int w = (tile % word_groups) * 4 + (threadIdx.x >> 6); // wave-uniform
    const int r0 = (tile / word_groups) * EXTRACT_RG, r1 = min(r0 + EXTRACT_RG, crows);
    const int c = w * 64 + (threadIdx.x & 63);
    const int cc = min(c, ccols - 1);
    const int ylo = tile_rows[2 * tile], yhi = tile_rows[2 * tile + 1];
    // general path: non-binary mask, or rows that are not a whole number of 32-bit words
    const bool general = nonbinary[f] != 0 || (pcols & 31) != 0; // block-uniform
    const int pw = pcols >> 5;  // words per polar row
    const int S = pw | 1;       // LDS row stride: odd, so rows 2 apart do not share a bank
    if (!general && ylo <= yhi) {
        const uint32_t *__restrict__ src = bits + (long long)f * words_per_frame + (long long)ylo * pw;
        const int nw = (yhi - ylo + 1) * pw;
        // 8 independent loads in flight per lane (the staging is otherwise a chain of L2 latencies)
        for (int i0 = threadIdx.x; i0 < nw; i0 += 256 * 8) {
            uint32_t v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * 256;
                v[u] = (i < nw) ? src[i] : 0u;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * 256;
                if (i < nw) {
                    const int r = i / pw, cw = i - r * pw;
                    s_bits[r * S + cw] = v[u];
                }
            }
        }
    }
    __syncthreads();
    const uint8_t *__restrict__ msrc = mask + (long long)f * prows * pcols;
    for (int rb = r0; rb < r1 && w < wpr; rb += EXTRACT_U) {
        uint32_t cd[EXTRACT_U];
#pragma unroll
        for (int i = 0; i < EXTRACT_U; ++i) {
            const int row = min(rb + i, crows - 1);
            cd[i] = code[(long long)row * ccols + cc];
        }
#pragma unroll
        for (int i = 0; i < EXTRACT_U; ++i) {
            const int row = rb + i;
            const int rowc = min(row, crows - 1);
            const int first = span[2 * rowc], last = span[2 * rowc + 1];
            const bool ok = row < r1 && c >= first && c < last && cd[i] != SFE_CODE_NONE;
            bool bit = false;
            if (w * 64 >= last || w * 64 + 64 <= first) {
                // wave-uniform: the whole word lies outside the sonar fan
            } else if (general) {
                if (ok)
                    bit = remap_value(msrc, prows, pcols, rcp, cd[i]) != 0;
            } else if (ylo <= yhi) { // (a tile without any valid pixel has nothing staged)
                const unsigned lin = cd[i] >> 10;
                const int fy = (int)((cd[i] >> 5) & 31u), fx = (int)(cd[i] & 31u);
                const unsigned q = __umulhi(lin, rcp);
                const int iy = (int)q - 1, ix = (int)(lin - q * (unsigned)(pcols + 1)) - 1;
                // clamp into the staged rows (valid pixels are inside by construction), zero the
                // out-of-image taps afterwards
                const int ya = min(max(iy, ylo), yhi), yb = min(max(iy + 1, ylo), yhi);
                const int xa = min(max(ix, 0), pcols - 1), xb = min(max(ix + 1, 0), pcols - 1);
                const int my0 = (iy >= 0 && iy < prows), my1 = (iy + 1 >= 0 && iy + 1 < prows);
                const int mx0 = (ix >= 0 && ix < pcols), mx1 = (ix + 1 >= 0 && ix + 1 < pcols);
                const int ra = (ya - ylo) * S, rb2 = (yb - ylo) * S;
                const int v00 = (s_bits[ra + (xa >> 5)] >> (xa & 31)) & my0 & mx0;
                const int v01 = (s_bits[ra + (xb >> 5)] >> (xb & 31)) & my0 & mx1;
                const int v10 = (s_bits[rb2 + (xa >> 5)] >> (xa & 31)) & my1 & mx0;
                const int v11 = (s_bits[rb2 + (xb >> 5)] >> (xb & 31)) & my1 & mx1;
                int w00 = (32 - fy) * (32 - fx) * 32, w01 = (32 - fy) * fx * 32;
                int w10 = fy * (32 - fx) * 32, w11 = fy * fx * 32;
                if ((fx | fy) == 0) {
                    w00 = 32767;
                    w11 = 1;
                }
                const int acc = w00 * v00 + w01 * v01 + w10 * v10 + w11 * v11;
                bit = ok && ((acc + 16384) >> 15) != 0;
            }
            const unsigned long long word = __ballot(bit);
            if ((threadIdx.x & 63) == 0 && row < r1)
                bitmap[((long long)f * crows + row) * wpr + w] = word;
        }
    }
    __syncthreads(); // the staged rows are replaced by the next frame's
    }
}

// pass 2: one workgroup per frame: row counts from the bitmap, exclusive scan -> row offsets + total.  The bitmap is
// read as one flat stream (lane t takes words t, t + 1024, ...: coalesced); the few non-empty words add their
// popcount to their row's LDS counter.
#define SCAN_THREADS 1024
// inclusive prefix sum over the 64 lanes of a wave: four DPP row shifts inside the rows of 16, then the row totals
__device__ __forceinline__ int scan_wave_incl(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false); // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false); // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false); // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false); // row_shr:8
    const int r0 = __builtin_amdgcn_readlane(v, 15), r1 = __builtin_amdgcn_readlane(v, 31),
              r2 = __builtin_amdgcn_readlane(v, 47);
    const int row = (threadIdx.x & 63) >> 4;
    return v + (row > 0 ? r0 : 0) + (row > 1 ? r1 : 0) + (row > 2 ? r2 : 0);
}
#define SCAN_LDS_LIST 4096 // word-list entries held in LDS until their offsets are known (64 KiB; a sonar frame has 2-3 thousand)
// Word list (list_cap > 0): the non-empty bitmap words of the frame for extract_expand_words_kernel, one int4 each:
// {row << 16 | word of the row, number of the frame's points in front of it, the 64 bits}.  The words are collected
// while the bitmap streams by (wave-aggregated append, any order).  A lane knows the set bits in front of its word
// within its wave's 64 consecutive words from a wave scan of the popcounts; what a row leaves in the preceding 64-word
// chunk (rows are <= 64 words here) goes through s_tail, and the row offsets are added once they stand.  (Counting the
// row's earlier words per list entry afterwards -- the first version -- cost 40 us per 512 frames: every lane of a wave
// reads another row, 64 cache lines per load instruction.)  A frame with more points than `cap` gets no list (a
// word may be missing from it): it is queued in ovf_list for the per-point kernel, which stores its first cap points.
__global__ __launch_bounds__(SCAN_THREADS) void extract_scan_kernel(const unsigned long long *__restrict__ bitmap,
                                                                    int32_t *__restrict__ row_count,
                                                                    int32_t *__restrict__ row_off,
                                                                    int32_t *__restrict__ frame_count, int crows, int wpr,
                                                                    int4 *__restrict__ wlist, int32_t *__restrict__ wlist_n,
                                                                    int list_cap, long long cap,
                                                                    int32_t *__restrict__ ovf_n, int32_t *__restrict__ ovf_list,
                                                                    const int32_t *__restrict__ only)
{
    if (only && !only[blockIdx.x]) // (the record path has done this frame)
        return;
    extern __shared__ __attribute__((aligned(16))) int s_cnt[]; // [SCAN_LDS_LIST int4 entries |] crows row counts |
                                                                // SCAN_THREADS partial sums [| one tail per 64-word chunk]
    __shared__ int s_nlist;
    // the first SCAN_LDS_LIST entries wait in LDS for their offsets and reach the list in one piece; a denser frame's
    // further entries go to the list at once and are completed there (read back by this workgroup: slow, rare)
    int4 *s_list = reinterpret_cast<int4 *>(s_cnt);
    int *s_cntp = s_cnt + (list_cap > 0 ? 4 * SCAN_LDS_LIST : 0);
    int *s_part = s_cntp + crows;
    int *s_tail = s_part + SCAN_THREADS;
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const unsigned long long *__restrict__ bm = bitmap + (long long)f * crows * wpr;
    int4 *__restrict__ wl = wlist ? wlist + (long long)f * list_cap : nullptr;
    const int nw = crows * wpr;
    for (int i = tid; i < crows; i += SCAN_THREADS)
        s_cntp[i] = 0;
    if (list_cap > 0)
        for (int i = tid; i < (nw + 63) / 64; i += SCAN_THREADS)
            s_tail[i] = 0;
    if (tid == 0)
        s_nlist = 0;
    __syncthreads();
    for (int i0 = tid; i0 - lane < nw; i0 += 8 * SCAN_THREADS) { // eight loads in flight per lane; wave-uniform trip count
        unsigned long long wd[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * SCAN_THREADS;
            wd[u] = i < nw ? bm[i] : 0ull;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * SCAN_THREADS;
            const int pc = __popcll(wd[u]);
            if (list_cap <= 0) {
                if (pc)
                    atomicAdd(&s_cntp[i / wpr], pc);
            } else {
                const unsigned long long m = __ballot(pc != 0);
                if (m) { // (wave-uniform)
                    const int row = i / wpr;
                    if (pc)
                        atomicAdd(&s_cntp[row], pc);
                    const int incl = scan_wave_incl(pc);
                    const int excl = incl - pc;
                    const int cstart = __builtin_amdgcn_readfirstlane(i - lane); // first word of the wave's chunk
                    const int rs = row * wpr;                                    // first word of this lane's row
                    // set bits of the row's earlier words inside this chunk (lane 0 holds excl = 0)
                    const int before = excl - __builtin_amdgcn_ds_bpermute(4 * max(rs - cstart, 0), excl);
                    // what the row that runs on into the next chunk has in this one
                    const int rns = ((cstart + 64) / wpr) * wpr;
                    const int tail = __builtin_amdgcn_readlane(incl, 63) -
                                     __builtin_amdgcn_readlane(excl, min(max(rns - cstart, 0), 63));
                    if (lane == 0 && rns < cstart + 64)
                        s_tail[cstart >> 6] = tail;
                    int base = 0;
                    const int leader = __ffsll((long long)m) - 1;
                    if (lane == leader)
                        base = atomicAdd(&s_nlist, __popcll(m));
                    base = __builtin_amdgcn_readlane(base, leader);
                    const int slot = base + __popcll(m & ((1ull << lane) - 1ull));
                    const int4 ent = make_int4((row << 16) | (i - rs), before, (int)(unsigned)(wd[u] & 0xFFFFFFFFull),
                                               (int)(unsigned)(wd[u] >> 32));
                    if (pc && slot < SCAN_LDS_LIST)
                        s_list[slot] = ent;
                    else if (pc && slot < list_cap)
                        wl[slot] = ent;
                }
            }
        }
    }
    __syncthreads();
    int32_t *__restrict__ cnt = row_count + (long long)f * crows;
    int32_t *__restrict__ off = row_off + (long long)f * crows;
    const int per = (crows + SCAN_THREADS - 1) / SCAN_THREADS;
    const int b = tid * per, e = min(b + per, crows);
    int s = 0;
    for (int i = b; i < e; ++i) {
        cnt[i] = s_cntp[i];
        s += s_cntp[i];
    }
    s_part[tid] = s;
    __syncthreads();
    for (int d = 1; d < SCAN_THREADS; d <<= 1) { // Hillis-Steele inclusive scan
        const int v = (tid >= d) ? s_part[tid - d] : 0;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    int run = (tid == 0) ? 0 : s_part[tid - 1];
    for (int i = b; i < e; ++i) {
        off[i] = run;
        const int c = s_cntp[i];
        s_cntp[i] = run; // from here on: the row's offset
        run += c;
    }
    const int total = s_part[SCAN_THREADS - 1];
    if (tid == SCAN_THREADS - 1)
        frame_count[f] = total;
    if (list_cap <= 0)
        return;
    __syncthreads(); // row offsets in s_cnt, the collected words in wl (written by this workgroup)
    const bool listed = total <= cap && s_nlist <= list_cap; // (total <= cap implies the second: a word holds a point)
    if (tid == 0) {
        wlist_n[f] = listed ? s_nlist : 0;
        if (!listed)
            ovf_list[atomicAdd(ovf_n, 1)] = f;
    }
    if (!listed)
        return;
    const int n = s_nlist;
    for (int e2 = tid; e2 < n; e2 += SCAN_THREADS) {
        int4 ent = e2 < SCAN_LDS_LIST ? s_list[e2] : wl[e2];
        const int row = ent.x >> 16, rs = row * wpr, cstart = (rs + (ent.x & 0xFFFF)) & ~63;
        ent.y += s_cntp[row] + (rs < cstart ? s_tail[(cstart >> 6) - 1] : 0);
        wl[e2] = ent;
    }
}


// pass 3, word form: one lane per BYTE of a non-empty bitmap word (extract_scan_kernel's list; 8 lanes share a word):
// the set bits of a word are consecutive points of the frame (np.nonzero order: row-major), the byte's first one
// comes after the word's offset + the set bits of the lower bytes.  A lane walks its <= 8 bits -- next set bit,
// column, metres from the px->m tables (sfe_geom: the fp64 expressions of feature_extraction.py:236-237 evaluated
// once per row / column on the host instead of two fp64 divisions per point; staged in LDS), one 16-byte store; the
// 8 lanes of a word write one contiguous run.  ~25 instructions per point against ~150 of the lane-per-point form
// below (row search, k-th set bit of the row, divisions).  The kernel is a chain of dependent accesses (list length
// -> entry -> tables -> store), so the next entry is fetched while the current one is expanded.
#define EXPAND_WG 8 // workgroups per frame (a sonar frame has 2-3 thousand non-empty words)
// Round 3: a lane per POINT of a batch of words.  A wave takes 64 list entries (a lane each: popcount, wave prefix sum ->
// T points in the batch, ~240), parks them in LDS, and then lane j of every round of 64 finds the entry its point
// belongs to (binary search in the 64 prefix values), clears the lower set bits of the word up to the point's rank and
// stores {y(row), x(col)} at the entry's offset + rank.  The entries of one 64-word chunk of the bitmap follow each other
// in the list with consecutive offsets, so a round's 64 stores of 16 bytes are one or two contiguous kilobytes (the
// lane-per-byte form above left 17 % of the store lanes busy and ran at the line rate of its partial stores: 61 us per
// 512 frames).  The words it has expanded are cleared in the canvas bitmap on the way (clean_bm != nullptr): the next
// batch finds the bitmap zero without a 128 MB memset (extract_dev).
__global__ __launch_bounds__(256) void extract_expand_words_kernel(const int4 *__restrict__ wlist,
                                                                   const int32_t *__restrict__ wlist_n, int list_cap,
                                                                   long long *__restrict__ rc_out,
                                                                   double *__restrict__ pts_out, long long cap, int crows,
                                                                   int ccols, const double *__restrict__ ytab,
                                                                   const double *__restrict__ xtab,
                                                                   unsigned long long *__restrict__ clean_bm, int wpr,
                                                                   const int32_t *__restrict__ only)
{
    extern __shared__ double s_tab[]; // crows y values, ccols x values
    __shared__ int4 s_ent[4][64];
    __shared__ int s_ex[4][64];
    double *s_y = s_tab, *s_x = s_tab + crows;
    const int f = blockIdx.y;
    if (only && !only[f])
        return;
    const int n = wlist_n[f];
    const int4 *__restrict__ wl = wlist + (long long)f * list_cap;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (blockIdx.x * 256 >= n)
        return; // nothing for this workgroup (the list is short): skip the tables too
    if (pts_out) {
        for (int i = threadIdx.x; i < crows; i += 256)
            s_y[i] = ytab[i];
        for (int i = threadIdx.x; i < ccols; i += 256)
            s_x[i] = xtab[i];
    }
    int4 ent_next = make_int4(0, 0, 0, 0);
    {
        const int e0 = blockIdx.x * 256 + wave * 64 + lane;
        if (e0 < n)
            ent_next = wl[e0];
    }
    for (int eb = blockIdx.x * 256; eb < n; eb += gridDim.x * 256) { // (workgroup-uniform)
        const int e = eb + wave * 64 + lane;
        const int4 ent = ent_next;
        {   // the next batch's entry is requested before this one is expanded (the list read was an exposed round trip
            // per batch)
            const int en = e + gridDim.x * 256;
            ent_next = en < n ? wl[en] : make_int4(0, 0, 0, 0);
        }
        const int pc = __popc((unsigned)ent.z) + __popc((unsigned)ent.w); // (0 past the end of the list)
        if (clean_bm && e < n)
            clean_bm[((long long)f * crows + (ent.x >> 16)) * wpr + (ent.x & 0xFFFF)] = 0ull;
        const int incl = scan_wave_incl(pc);
        const int total = __builtin_amdgcn_readlane(incl, 63);
        s_ent[wave][lane] = ent;
        s_ex[wave][lane] = e < n ? incl - pc : 0x7FFFFFFF;
        __syncthreads();
        for (int j = lane; j < total; j += 64) {
            int lo = 0, hi = 63; // the last entry whose first point is <= j
#pragma unroll
            for (int step = 0; step < 6; ++step) {
                const int mid = (lo + hi + 1) >> 1;
                if (s_ex[wave][mid] <= j)
                    lo = mid;
                else
                    hi = mid - 1;
            }
            const int4 en = s_ent[wave][lo];
            const int r = j - s_ex[wave][lo];
            // the r-th set bit of the word, by halving (a loop that clears r bits costs the wave its densest word: 63 steps)
            int rr = r, bit = 0;
            unsigned w32 = (unsigned)en.z;
            {
                const int c = __popc(w32);
                if (rr >= c) {
                    rr -= c;
                    w32 = (unsigned)en.w;
                    bit = 32;
                }
            }
#pragma unroll
            for (int h = 16; h >= 1; h >>= 1) {
                const int c = __popc(w32 & ((1u << h) - 1u));
                if (rr >= c) {
                    rr -= c;
                    w32 >>= h;
                    bit += h;
                }
            }
            const int row = en.x >> 16, col = (en.x & 0xFFFF) * 64 + bit;
            const long long t = (long long)f * cap + en.y + r;
            if (rc_out)
                reinterpret_cast<longlong2 *>(rc_out)[t] = make_longlong2(row, col);
            if (pts_out)
                reinterpret_cast<double2 *>(pts_out)[t] = make_double2(s_y[row], s_x[col]);
        }
        __syncthreads();
    }
}

// the canvas bitmaps of the frames that got no word list (above the point capacity; queued by the scan kernel): cleared
// whole, after the per-point kernel has read them.  Normally no frame is queued and the workgroups leave at once.
__global__ __launch_bounds__(256) void extract_clean_queued_kernel(unsigned long long *__restrict__ bitmap,
                                                                   long long words_per_frame,
                                                                   const int32_t *__restrict__ ovf_n,
                                                                   const int32_t *__restrict__ ovf_list)
{
    const int nq = *ovf_n;
    for (int q = 0; q < nq; ++q) {
        unsigned long long *bm = bitmap + (long long)ovf_list[q] * words_per_frame;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < words_per_frame; i += (long long)gridDim.x * 256)
            bm[i] = 0ull;
    }
}


// pass 3, point form: one lane per POINT: point t of a frame lies in the last row whose offset is <= t (binary
// search in the row offsets; empty rows share their successor's offset, so the last such row is the occupied one) and
// is the (t - offset)-th set bit of that row's words in column order (= np.nonzero order); then metres in fp64,
// operation by operation.  The general form: it stores the first `cap` points of a frame whatever its count.  With
// ovf_list it only serves the frames queued there (frames above the capacity, which the word form leaves alone:
// normally none, the kernel then ends at once); A/B against the word form with SFE_EXPAND_POINTS=1.
__global__ __launch_bounds__(256) void extract_expand_kernel(const unsigned long long *__restrict__ bitmap,
                                                             const int32_t *__restrict__ row_off,
                                                             const int32_t *__restrict__ frame_count,
                                                             long long *__restrict__ rc_out,
                                                             double *__restrict__ pts_out, long long cap,
                                                             int crows, int ccols, int wpr, double width,
                                                             double height, const int32_t *__restrict__ ovf_n,
                                                             const int32_t *__restrict__ ovf_list)
{
    extern __shared__ int32_t s_off[]; // the frame's row offsets: one coalesced fetch instead of a 10-step
                                       // chain of dependent loads per point
    const int nk = ovf_list ? *ovf_n : 1;
    for (int k = 0; k < nk; ++k) {
        const int f = ovf_list ? ovf_list[k] : (int)blockIdx.y;
        const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
        const long long n = min((long long)frame_count[f], cap);
        if ((long long)blockIdx.x * 256 >= n)
            continue; // the whole workgroup is past the frame's last point
        const int32_t *__restrict__ off = row_off + (long long)f * crows;
        __syncthreads(); // (the previous frame's offsets are no longer read)
        for (int i = threadIdx.x; i < crows; i += 256)
            s_off[i] = off[i];
        __syncthreads();
        if (t >= n)
            continue;
        int lo = 0, hi = crows - 1; // largest row with off[row] <= t
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (s_off[mid] <= t)
                lo = mid;
            else
                hi = mid - 1;
        }
        const int row = lo;
        int kk = (int)(t - s_off[row]);
        const unsigned long long *__restrict__ brow = bitmap + ((long long)f * crows + row) * wpr;
        int col = -1;
        for (int w0 = 0; w0 < wpr && col < 0; w0 += 8) { // eight independent loads in flight, then the selection
            unsigned long long wd[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                wd[u] = (w0 + u < wpr) ? brow[w0 + u] : 0ull;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int c = __popcll(wd[u]);
                if (col < 0 && kk < c) {
                    unsigned long long word = wd[u];
                    for (; kk > 0; --kk)
                        word &= word - 1;
                    col = (w0 + u) * 64 + __ffsll((long long)word) - 1;
                }
                if (col < 0)
                    kk -= c;
            }
        }
        if (col < 0)
            continue; // cannot happen: the offsets were counted from these very words
        const long long dsto = ((long long)f * cap + t) * 2;
        if (rc_out) {
            rc_out[dsto] = row;
            rc_out[dsto + 1] = col;
        }
        if (pts_out) {
            // feature_extraction.py:236-237 in float64, operation by operation
            const double half_cols = ccols / 2.;
            const double y = (-1 * ((double)row / (double)crows) * height) + height;
            double x = (double)col - half_cols;
            x = (-1 * ((x / half_cols) * (width / 2.)));
            pts_out[dsto] = y;
            pts_out[dsto + 1] = x;
        }
    }
}

// pass 1 for BINARY masks, inverted: CFAR detections are sparse (~1 % of the polar pixels), so instead of evaluating all
// ~2 M canvas pixels of a frame, walk the set polar pixels and evaluate only the canvas pixels that tap them (precomputed
// inverse map, sfe_geom_create): a few candidates per set pixel, each with exactly the blend of the dense pass.  (Round 2's
// form gave a workgroup a block of polar rows: a sonar frame's ~5 000 detections sit in a few range bands, so most waves
// found a handful of set pixels and paid ~900 instructions of staging, scanning and prefix searches around them;
// profiles/r05_pruned_variants.txt)  List first, then one lane per set pixel: a workgroup takes every
// `slices`-th 64-word piece of the frame's bit stream (interleaved: every workgroup sees every band), collects the
// set pixels of its pieces in ONE LDS list (unordered: the canvas bits are OR-ed), and then all 256 threads draw from
// that list: a lane owns one set pixel, reads the 3 x 3 mask bits around it once from the frame's bit stream (rows
// y-1 .. y+1: every tap of every candidate of this pixel lies there) and walks the pixel's inverse-map range itself --
// neighbouring polar pixels have ranges of about the same length, so the lanes of a wave finish together and no
// search maps candidates to lanes.  No staging of mask rows in LDS.
// The canvas bits: a frame's ~10 000 canvas points lie in ~2 300 bitmap words, and with everything else out of the way
// the global atomics were what the kernel waited for (0.31 -> 0.13 ms per 512 frames without them).  A lane first joins
// the consecutive entries of its pixel that fall into one bitmap word (the entries are sorted by canvas pixel), then
// ORs the word into a direct-mapped LDS table (tag = bitmap word of the frame); a word that finds its slot taken by
// another goes to memory directly, the table is written out once at the end.
// The blend itself is decided when the geometry is built: whether a candidate is a detection and whether this tap reports
// it depends on the entry and the four tap bits only, so the entry carries the 16 answers (sfe_geom_create) and a
// candidate costs a shift, a mask and a bit test instead of the ~50 instructions of the fixed-point blend.
// Identical canvas bitmap (same candidates, same blend, same first-tap rule).
#define SG_LIST 1536  // set pixels a workgroup collects before it expands them (list + table: 18 KB, eight workgroups per CU)
#define SG_BLOCK 1024 // mask words looked at per collection step (4 per thread)
#define SG_TAB 1024   // slots of the canvas-word table
#define SG_SPILL 1024 // record form: slots per frame for words that found no table slot
// COMPACT (round 4): 4-byte entries {decision table [15:0], tap place [17:16], dx [24:18], dy [31:25]} relative to a
// per-pixel base bit index that travels with the pixel's offset ({offset, base} pairs: ONE 16-byte read brings the
// pixel's offset, its base and the next offset), four entries per 16-byte read; entries that can never report (their
// table is 0: a last tap below half weight, ...) are not stored at all.  Half the bytes and fewer reads per set pixel
// than the 8-byte {bit index, table} entries (inv_off / inv_ent of the other instantiation).
// RECORDS (round 5): the canvas words a workgroup has combined do not go to the canvas bitmap at all: they are appended to the
// frame's record list {bitmap word of the frame, 0, its 64 bits} (one allocation per workgroup), and
// extract_merge_expand_kernel merges the records of a frame in LDS and emits the points -- no canvas in HBM, no stream over it.
// `only` (canvas form): the frames to work on (the frames the record path handed back: more records / words / points than its
// capacities); nullptr = all.
template <bool COMPACT, bool RECORDS>
__global__ __launch_bounds__(256, 8) void extract_gather_kernel(const uint32_t *__restrict__ bits,
                                                                const int32_t *__restrict__ nonbinary,
                                                                const int32_t *__restrict__ inv_off,
                                                                const uint2 *__restrict__ inv_ent,
                                                                unsigned long long *__restrict__ bitmap, int prows,
                                                                int pcols, int crows, int wpr, long long words_per_frame,
                                                                int piece_shift, int4 *__restrict__ rec,
                                                                int32_t *__restrict__ rec_n, int rec_cap,
                                                                const int32_t *__restrict__ only)
{
    __shared__ uint32_t s_list[SG_LIST]; // (row << 16 | column) of a set pixel
    __shared__ int s_n, s_want[2]; // (s_want: by step parity -- the other one is cleared while this one is read)
    __shared__ unsigned s_tag[SG_TAB];
    __shared__ unsigned long long s_acc[SG_TAB];
    const int f = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    if (nonbinary[f] != 0 || (only && !only[f]))
        return;
    for (int i = tid; i < SG_TAB; i += 256) {
        s_tag[i] = 0xFFFFFFFFu;
        s_acc[i] = 0ull;
    }
    if (tid == 0) {
        s_n = 0;
        s_want[0] = s_want[1] = 0;
    }
    __syncthreads();
    const int pw = pcols >> 5, nwords = prows * pw;
    const int slices = gridDim.x;
    // rotate the pieces from frame to frame: workgroups are dealt to the 8 XCDs in launch order, and every XCD should
    // see every range band (cf. mode 2 of the kernel above)
    const int sl = (int)((blockIdx.x + 5u * blockIdx.y) % (unsigned)slices);
    const uint32_t *__restrict__ src = bits + (long long)f * words_per_frame;
    unsigned long long *__restrict__ bm = bitmap + (long long)f * crows * wpr;
    // records: this workgroup's region of the frame's list (SG_TAB slots: the table holds no more) and, for the few words that
    // found no table slot, the spill region behind the workgroups' regions (SG_SPILL slots, one returning atomic each)
    const long long rec_stride = (long long)slices * SG_TAB + SG_SPILL;
    int4 *__restrict__ rl = RECORDS ? rec + (long long)f * rec_stride : nullptr;
    int32_t *__restrict__ rcnt = RECORDS ? rec_n + (long long)f * (slices + 1) : nullptr;
    auto or_word = [&](unsigned wd, unsigned long long m) {
        if (RECORDS) {
            const int pos = atomicAdd(&rcnt[slices], 1); // (beyond SG_SPILL: the count says so, the frame is handed back)
            if (pos < SG_SPILL)
                rl[(long long)slices * SG_TAB + pos] = make_int4((int)wd, 0, (int)(unsigned)(m & 0xFFFFFFFFull), (int)(unsigned)(m >> 32));
        } else
            atomicOr(&bm[wd], m);
    };
    // pieces of 64 << piece_shift words (64 words = 4 polar rows of 512 beams); piece p belongs to slice p % slices.  The
    // workgroup's words, piece after piece, are looked at `blk` at a time: thread t takes words t, t + 256, ...
    const int pwords = 64 << piece_shift;
    const int npieces = (nwords + pwords - 1) >> (6 + piece_shift);
    const int my_pieces = (npieces - sl + slices - 1) / slices; // pieces sl, sl + slices, ...
    const int my_words = my_pieces * pwords;
    int v0 = 0, blk = SG_BLOCK, par = 0;
    while (true) {
        // ---- collect: steps of `blk` words while their set pixels fit the list
        while (v0 < my_words) {
            uint32_t w[4];
            int gw[4], pc = 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int v = v0 + u * 256 + tid;
                gw[u] = (sl + (v >> (6 + piece_shift)) * slices) * pwords + (v & (pwords - 1));
                w[u] = (u * 256 + tid < blk && v < my_words && gw[u] < nwords) ? src[gw[u]] : 0u;
                pc += __popc(w[u]);
            }
            const int incl = scan_wave_incl(pc);
            const int wtotal = __builtin_amdgcn_readlane(incl, 63);
            int base = 0;
            if (wtotal != 0 && lane == 63)
                base = atomicAdd(&s_want[par], wtotal); // (one LDS atomic per wave that has any)
            base = __builtin_amdgcn_readlane(base, 63);
            __syncthreads();
            const int want = s_want[par], at = s_n; // set pixels of this step, set pixels already listed
            if (tid == 0)
                s_want[par ^ 1] = 0;
            __syncthreads();
            par ^= 1;
            if (at + want > SG_LIST) {
                // this step waits for the list to be expanded.  Alone it does not fit either (more than SG_LIST set pixels
                // in 1024 words: no sonar mask): 32 words per step from here on, whose 1024 pixels always fit
                if (at == 0)
                    blk = 32;
                break;
            }
            if (want != 0) {
                int pos = at + base + incl - pc;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    uint32_t word = w[u];
                    if (word) {
                        const int row = gw[u] / pw, c0 = (gw[u] - row * pw) * 32;
                        do {
                            const int bb = __ffs((int)word) - 1;
                            s_list[pos++] = ((uint32_t)row << 16) | (uint32_t)(c0 + bb);
                            word &= word - 1u;
                        } while (word);
                    }
                }
                __syncthreads();
                if (tid == 0)
                    s_n = at + want;
            }
            v0 += blk;
        }
        __syncthreads();
        // ---- expand the list: one lane per set pixel
        const int n = s_n;
        for (int j = tid; j < n; j += 256) {
            const uint32_t ent = s_list[j];
            const int py = (int)(ent >> 16), px = (int)(ent & 0xFFFFu);
            const int pi = py * pcols + px;
            int off0, cnt;
            unsigned base_bit = 0;
            if (COMPACT) {
                uint4 ob; // {offset, base} of this pixel, offset of the next one
                __builtin_memcpy(&ob, reinterpret_cast<const uint2 *>(inv_off) + pi, 16);
                off0 = (int)ob.x;
                base_bit = ob.y;
                cnt = (int)ob.z - off0;
            } else {
                off0 = inv_off[pi];
                cnt = inv_off[pi + 1] - off0;
            }
            // the 3 x 3 mask bits around the pixel: bit 3 * (dy + 1) + dx + 1; 0 outside the image.  Per row ONE 8-byte read
            // of two neighbouring words of the bit stream that hold columns px - 1 .. px + 1 (4-byte aligned)
            unsigned nb;
            if (pw >= 2) {
                const int wb = min(max((px >> 5) - ((px & 31) < 16 ? 1 : 0), 0), pw - 2);
                const int rel = px - 1 - 32 * wb; // -1 .. 62: where column px - 1 sits in the pair
                auto bits3 = [&](int yy) -> unsigned {
                    const bool in = yy >= 0 && yy < prows;
                    unsigned long long ww;
                    __builtin_memcpy(&ww, src + (long long)(in ? yy : py) * pw + wb, 8);
                    ww = rel >= 0 ? ww >> rel : ww << 1; // (column -1 and column pcols fall off the ends: 0)
                    return in ? (unsigned)ww & 7u : 0u;
                };
                nb = bits3(py - 1) | (bits3(py) << 3) | (bits3(py + 1) << 6);
            } else { // a 32-beam image: one word per row
                auto bits3 = [&](int yy) -> unsigned {
                    if (yy < 0 || yy >= prows)
                        return 0u;
                    return (unsigned)(((unsigned long long)src[yy] << 1) >> px) & 7u;
                };
                nb = bits3(py - 1) | (bits3(py) << 3) | (bits3(py + 1) << 6);
            }
            unsigned run_w = 0xFFFFFFFFu;
            unsigned long long run_m = 0ull;
            auto emit = [&](unsigned wd, unsigned long long m) {
                unsigned slot = (wd * 0x9E3779B1u) >> (32 - 10);
                static_assert(SG_TAB == 1 << 10, "slot bits");
#pragma unroll 1
                for (int probe = 0; probe < (RECORDS ? 8 : 1); ++probe) { // (records: a taken slot costs a global allocation)
                    const unsigned old = atomicCAS(&s_tag[slot], 0xFFFFFFFFu, wd);
                    if (old == 0xFFFFFFFFu || old == wd) {
                        atomicOr(&s_acc[slot], m);
                        return;
                    }
                    slot = (slot + 1u) & (SG_TAB - 1u);
                }
                or_word(wd, m);
            };
            auto candidate = [&](const uint2 e) { // e.y = 0 (no case reports): the padding of the last round
                const unsigned sh = nb >> (e.y >> 16);
                const unsigned pat = (sh & 3u) | ((sh >> 1) & 12u); // taps 00 01 10 11 of the canvas pixel
                if ((e.y >> pat) & 1u) {
                    const unsigned wd = e.x >> 6;
                    if (wd != run_w) {
                        if (run_m)
                            emit(run_w, run_m);
                        run_w = wd;
                        run_m = 0ull;
                    }
                    run_m |= 1ull << (e.x & 63u);
                }
            };
            if (COMPACT) {
                const unsigned rowbits = (unsigned)wpr * 64u;
                const uint32_t *__restrict__ ent4 = reinterpret_cast<const uint32_t *>(inv_ent);
                auto cand4 = [&](uint32_t e, bool live) {
                    const unsigned sc = (e >> 16) & 3u; // tap place: ry * 2 + rx -> shift ry * 3 + rx
                    const unsigned x = base_bit + (e >> 25) * rowbits + ((e >> 18) & 127u);
                    candidate(make_uint2(x, live ? ((e & 0xFFFFu) | ((sc + (sc >> 1)) << 16)) : 0u));
                };
                for (int k = 0; k < cnt; k += 8) { // eight entries in flight: two 16-byte reads
                    uint4 a, b2;
                    __builtin_memcpy(&a, ent4 + off0 + k, 16);
                    __builtin_memcpy(&b2, ent4 + off0 + min(k + 4, cnt - 1), 16);
                    const bool two = k + 4 < cnt;
                    cand4(a.x, true);
                    cand4(a.y, k + 1 < cnt);
                    cand4(a.z, k + 2 < cnt);
                    cand4(a.w, k + 3 < cnt);
                    cand4(b2.x, two);
                    cand4(b2.y, k + 5 < cnt);
                    cand4(b2.z, k + 6 < cnt);
                    cand4(b2.w, k + 7 < cnt);
                }
            } else
            for (int k = 0; k < cnt; k += 4) { // the loads are what a lane waits for: four entries in flight, two per 16-byte read
                // (past the end of the range: whatever follows in the table -- it ends with two spare entries -- with an empty
                // decision table)
                uint4 a, b2;
                __builtin_memcpy(&a, inv_ent + off0 + k, 16);
                __builtin_memcpy(&b2, inv_ent + off0 + min(k + 2, cnt - 1), 16);
                const bool two = k + 2 < cnt; // (else b2 holds entries cnt - 1, cnt: both done or out of range)
                candidate(make_uint2(a.x, a.y));
                candidate(make_uint2(a.z, (k + 1 < cnt) ? a.w : 0u));
                candidate(make_uint2(b2.x, two ? b2.y : 0u));
                candidate(make_uint2(b2.z, (k + 3 < cnt) ? b2.w : 0u));
            }
            if (run_m)
                emit(run_w, run_m);
        }
        __syncthreads();
        if (v0 >= my_words)
            break;
        if (tid == 0)
            s_n = 0;
        __syncthreads();
    }
    if (!RECORDS) {
        for (int i = tid; i < SG_TAB; i += 256)
            if (s_tag[i] != 0xFFFFFFFFu)
                or_word(s_tag[i], s_acc[i]);
        return;
    }
    // the table as records: thread t holds slots 4 t .. 4 t + 3, the workgroup's records go to its own region of the list
    int mine = 0;
#pragma unroll
    for (int u = 0; u < SG_TAB / 256; ++u)
        mine += s_tag[(SG_TAB / 256) * tid + u] != 0xFFFFFFFFu;
    const int incl = scan_wave_incl(mine);
    if (tid == 0)
        s_want[0] = 0;
    __syncthreads();
    int wbase = 0;
    if (lane == 63 && incl)
        wbase = atomicAdd(&s_want[0], incl);
    wbase = __builtin_amdgcn_readlane(wbase, 63);
    __syncthreads();
    if (tid == 0)
        rcnt[blockIdx.x] = s_want[0];
    int4 *__restrict__ mine_out = rl + (long long)blockIdx.x * SG_TAB + wbase + incl - mine;
#pragma unroll
    for (int u = 0; u < SG_TAB / 256; ++u) {
        const int i = (SG_TAB / 256) * tid + u;
        const unsigned wd = s_tag[i];
        if (wd != 0xFFFFFFFFu) {
            const unsigned long long m = s_acc[i];
            *mine_out++ = make_int4((int)wd, 0, (int)(unsigned)(m & 0xFFFFFFFFull), (int)(unsigned)(m >> 32));
        }
    }
}

// The records of a frame -> its points, one workgroup per frame, everything between in LDS (round 5, VERDICT r4 item 5: the canvas
// bitmap cost 93 KB of atomics + 240 KB of stream + a 50 KB word list per frame to find ~2 300 words).  A presence bit per bitmap
// word of the frame (3.5 KB for config A) is set by every record; the popcount prefix over the presence bits IS the rank of a
// word among the frame's non-empty words in np.nonzero order, so the records OR their bits into a compact array at their word's
// rank -- no sort -- the prefix over the words' popcounts gives the point offsets, and the points are emitted a lane per point like
// extract_expand_words_kernel (binary search in the 64 offsets of a wave's batch, rank-select by halving, metres from the LDS tables).
// A frame that does not fit (more records than rec_cap, more non-empty words than capw, more points than cap) is flagged in
// ovf_flag and left to the canvas kernels, which look at flagged frames only.
#define ME_THREADS 1024
#define ME_RPT 8 // records per thread held in registers (rec_cap <= ME_THREADS * ME_RPT)
// IDX: the type of the compact arrays' word indices and point offsets -- uint16_t when the frame has < 65 536 bitmap words and the
// capacity is < 65 536 points (config A: 78 KB of LDS, two frames per CU), uint32_t otherwise.
template <typename IDX>
__global__ __launch_bounds__(ME_THREADS) void extract_merge_expand_kernel(const int4 *__restrict__ rec, const int32_t *__restrict__ rec_n,
                                                                          int rec_cap, int slices, int32_t *__restrict__ frame_count,
                                                                          int32_t *__restrict__ ovf_flag, long long *__restrict__ rc_out,
                                                                          double *__restrict__ pts_out, long long cap, int crows,
                                                                          int ccols, int wpr, int capw, const double *__restrict__ ytab,
                                                                          const double *__restrict__ xtab, float2 *__restrict__ p32_out,
                                                                          CfBBox *__restrict__ bbox_out)
{
    // p32_out / bbox_out (staged hand-over to the resident cloud filters, round 6): the frame's points once more as float2 --
    // the fp64 metres rounded to float32, what pybind does at pcl.cpp's boundary -- and their bounding box + count, so that the
    // filters start from here instead of reading the float64 points back (cf_cast_bbox_kernel); pts_out may then be null.
    extern __shared__ __attribute__((aligned(16))) unsigned char me_raw[];
    __shared__ int s_wsum[ME_THREADS / 64];
    __shared__ int s_total;
    const int nw = crows * wpr, npw = (nw + 31) >> 5;
    double *s_y = reinterpret_cast<double *>(me_raw), *s_x = s_y + crows;
    unsigned long long *w_bits = reinterpret_cast<unsigned long long *>(s_x + ccols);
    unsigned *pres = reinterpret_cast<unsigned *>(w_bits + capw);
    int *ppre = reinterpret_cast<int *>(pres + npw);
    IDX *w_idx = reinterpret_cast<IDX *>(ppre + npw);
    IDX *w_off = w_idx + capw; // capw + 1
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // the frame's records: `slices` regions of SG_TAB slots (one per workgroup of the gather kernel) + the spill region
    __shared__ int s_rp[66]; // s_rp[r] = records in front of region r
    const int32_t *__restrict__ rcnt = rec_n + (long long)f * (slices + 1);
    if (tid <= slices)
        s_rp[tid + 1] = rcnt[tid];
    if (tid == 0)
        s_rp[0] = 0;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int r = 0; r <= slices; ++r) {
            const int c = s_rp[r + 1];
            s_rp[r] = run;
            run += c;
        }
        s_rp[slices + 1] = run;
        s_total = s_rp[slices + 1] - s_rp[slices] > SG_SPILL ? 0x7FFFFFFF : run;
    }
    __syncthreads();
    const int n = s_total;
    __syncthreads();
    if (n > rec_cap) { // (workgroup-uniform)
        if (tid == 0)
            ovf_flag[f] = 1;
        return;
    }
    const int4 *__restrict__ rl = rec + (long long)f * ((long long)slices * SG_TAB + SG_SPILL);
    int4 ent[ME_RPT];
#pragma unroll
    for (int k = 0; k < ME_RPT; ++k) {
        const int r = tid + k * ME_THREADS;
        int lo = 0, hi = slices; // the region record r lies in: the last one that starts at or before r
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (s_rp[mid] <= r)
                lo = mid;
            else
                hi = mid - 1;
        }
        ent[k] = r < n ? rl[(long long)lo * SG_TAB + (r - s_rp[lo])] : make_int4(-1, 0, 0, 0);
    }
    for (int i = tid; i < npw; i += ME_THREADS)
        pres[i] = 0u;
    for (int i = tid; i < capw; i += ME_THREADS)
        w_bits[i] = 0ull;
    if (pts_out || p32_out) {
        for (int i = tid; i < crows; i += ME_THREADS)
            s_y[i] = ytab[i];
        for (int i = tid; i < ccols; i += ME_THREADS)
            s_x[i] = xtab[i];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < ME_RPT; ++k)
        if (ent[k].x >= 0)
            atomicOr(&pres[ent[k].x >> 5], 1u << (ent[k].x & 31));
    __syncthreads();
    // exclusive prefix of the presence popcounts: thread t owns `per` consecutive presence words
    auto block_excl = [&](int v) -> int { // -> exclusive prefix of v over the workgroup; s_total = the sum
        const int incl = scan_wave_incl(v);
        if (lane == 63)
            s_wsum[wave] = incl;
        __syncthreads();
        int base = 0;
        for (int w = 0; w < wave; ++w)
            base += s_wsum[w];
        if (tid == ME_THREADS - 1)
            s_total = base + incl;
        __syncthreads();
        return base + incl - v;
    };
    {
        const int per = (npw + ME_THREADS - 1) / ME_THREADS;
        const int b = tid * per, e = min(b + per, npw);
        int mine = 0;
        for (int i = b; i < e; ++i)
            mine += __popc(pres[i]);
        int run = block_excl(mine);
        for (int i = b; i < e; ++i) {
            ppre[i] = run;
            run += __popc(pres[i]);
        }
    }
    const int nW = s_total; // non-empty bitmap words of the frame
    __syncthreads();
    if (nW > capw) {
        if (tid == 0)
            ovf_flag[f] = 1;
        return;
    }
#pragma unroll
    for (int k = 0; k < ME_RPT; ++k)
        if (ent[k].x >= 0) {
            const int wd = ent[k].x;
            const int rank = ppre[wd >> 5] + __popc(pres[wd >> 5] & ((1u << (wd & 31)) - 1u));
            atomicOr(&w_bits[rank], ((unsigned long long)(unsigned)ent[k].w << 32) | (unsigned long long)(unsigned)ent[k].z);
            w_idx[rank] = (IDX)wd;
        }
    __syncthreads();
    {
        const int per = (nW + ME_THREADS - 1) / ME_THREADS;
        const int b = min(tid * per, nW), e = min(b + per, nW);
        int mine = 0;
        for (int i = b; i < e; ++i)
            mine += __popcll(w_bits[i]);
        int run = block_excl(mine);
        for (int i = b; i < e; ++i) {
            w_off[i] = (IDX)run; // (a run beyond the capacity may wrap: such a frame is handed back below)
            run += __popcll(w_bits[i]);
        }
    }
    const int total = s_total;
    if (tid == 0)
        frame_count[f] = total;
    if (total > cap) { // the canvas path stores such a frame's first cap points (and reports the same count)
        if (tid == 0)
            ovf_flag[f] = 1;
        return;
    }
    if (tid == 0)
        w_off[nW] = (IDX)total;
    __syncthreads();
    if (cap <= 0)
        return;
    float mnx = INFINITY, mny = INFINITY, mxx = -INFINITY, mxy = -INFINITY; // (staged: bounding box of this thread's points)
    for (int b = wave * 64; b < nW; b += (ME_THREADS / 64) * 64) { // a wave takes 64 consecutive words
        const int hi0 = min(b + 63, nW - 1);
        const int first = (int)w_off[b], npts = (int)w_off[hi0 + 1] - first;
        for (int j = lane; j < npts; j += 64) {
            const int t = first + j;
            int lo = b, hi = hi0; // the last word whose first point is <= t
#pragma unroll
            for (int step = 0; step < 6; ++step) {
                const int mid = (lo + hi + 1) >> 1;
                if ((int)w_off[mid] <= t)
                    lo = mid;
                else
                    hi = mid - 1;
            }
            const unsigned long long wbits = w_bits[lo];
            int rr = t - (int)w_off[lo], bit = 0;
            unsigned w32 = (unsigned)(wbits & 0xFFFFFFFFull);
            {
                const int c = __popc(w32);
                if (rr >= c) {
                    rr -= c;
                    w32 = (unsigned)(wbits >> 32);
                    bit = 32;
                }
            }
#pragma unroll
            for (int h = 16; h >= 1; h >>= 1) {
                const int c = __popc(w32 & ((1u << h) - 1u));
                if (rr >= c) {
                    rr -= c;
                    w32 >>= h;
                    bit += h;
                }
            }
            const int wd = (int)w_idx[lo], row = wd / wpr, col = (wd - row * wpr) * 64 + bit;
            const long long o = (long long)f * cap + t;
            if (rc_out)
                reinterpret_cast<longlong2 *>(rc_out)[o] = make_longlong2(row, col);
            if (pts_out)
                reinterpret_cast<double2 *>(pts_out)[o] = make_double2(s_y[row], s_x[col]);
            if (p32_out) {
                const float2 p = make_float2((float)s_y[row], (float)s_x[col]);
                p32_out[o] = p;
                mnx = fminf(mnx, p.x);
                mxx = fmaxf(mxx, p.x);
                mny = fminf(mny, p.y);
                mxy = fmaxf(mxy, p.y);
            }
        }
    }
    if (bbox_out) { // (min / max do not depend on the order they are taken in: the same box as cf_cast_bbox_kernel's)
        __shared__ float s_bb[4][ME_THREADS / 64];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            mnx = fminf(mnx, __shfl_down(mnx, d));
            mxx = fmaxf(mxx, __shfl_down(mxx, d));
            mny = fminf(mny, __shfl_down(mny, d));
            mxy = fmaxf(mxy, __shfl_down(mxy, d));
        }
        if (lane == 0) {
            s_bb[0][wave] = mnx;
            s_bb[1][wave] = mny;
            s_bb[2][wave] = mxx;
            s_bb[3][wave] = mxy;
        }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < ME_THREADS / 64; ++w) {
                mnx = fminf(mnx, s_bb[0][w]);
                mny = fminf(mny, s_bb[1][w]);
                mxx = fmaxf(mxx, s_bb[2][w]);
                mxy = fmaxf(mxy, s_bb[3][w]);
            }
            CfBBox bb;
            bb.mnx = mnx;
            bb.mny = mny;
            bb.mxx = mxx;
            bb.mxy = mxy;
            bb.n = total;
            bbox_out[f] = bb;
        }
    }
}

// staged hand-over, the frames the record path handed back to the canvas kernels (flagged; or every frame when `flags` is
// null: the record path was not taken): float64 points -> float2 + bounding box, what cf_cast_bbox_kernel does for all frames
// on the unstaged path.  A fixed, small grid whose WAVES stride over the frames: a wave looks at a frame's flag and moves on
// (one workgroup per frame cost 0.03 ms per 4096 frames in launches that almost always return at once); a flagged frame --
// rare -- is cast by the one wave that meets it.
__global__ __launch_bounds__(256) void extract_stage_fallback_kernel(const double *__restrict__ pts64, const int32_t *__restrict__ counts,
                                                                     long long cap, const int32_t *__restrict__ flags,
                                                                     float2 *__restrict__ p32, CfBBox *__restrict__ bbox, int n_frames)
{
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = gridDim.x * 4;
    for (int f = wave; f < n_frames; f += n_waves) {
        if (flags && !flags[f])
            continue;
        const int n = (int)min((long long)max(counts[f], 0), cap);
        const double *src = pts64 + (size_t)f * cap * 2;
        float2 *dst = p32 + (size_t)f * cap;
        float mnx = INFINITY, mny = INFINITY, mxx = -INFINITY, mxy = -INFINITY;
        for (int i = lane; i < n; i += 64) {
            const float2 p = make_float2((float)src[2 * i], (float)src[2 * i + 1]);
            dst[i] = p;
            mnx = fminf(mnx, p.x);
            mxx = fmaxf(mxx, p.x);
            mny = fminf(mny, p.y);
            mxy = fmaxf(mxy, p.y);
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            mnx = fminf(mnx, __shfl_down(mnx, d));
            mxx = fmaxf(mxx, __shfl_down(mxx, d));
            mny = fminf(mny, __shfl_down(mny, d));
            mxy = fmaxf(mxy, __shfl_down(mxy, d));
        }
        if (lane == 0) {
            CfBBox bb;
            bb.mnx = mnx;
            bb.mny = mny;
            bb.mxx = mxx;
            bb.mxy = mxy;
            bb.n = n;
            bbox[f] = bb;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// byte mask -> bit stream (mask_pack_kernel) for n_frames frames of px pixels; d_nonbin (optional, n_frames ints,
// zeroed by the caller) is set for frames holding a byte > 1
int sfe_mask_pack(sfe_ctx *ctx, const uint8_t *d_mask, int n_frames, long long px, uint32_t *d_bits, int32_t *d_nonbin)
{
    const long long wpf = (px + 31) / 32 + 1;
    if (!d_nonbin) {
        d_nonbin = (int32_t *)sfe_scratch(ctx, 11, (size_t)std::max(n_frames, 1024) * 4);
        if (!d_nonbin)
            return SFE_ERR_HIP;
    }
    for (int f0 = 0; f0 < n_frames; f0 += 32768) { // gridDim.y <= 65535
        const int nf = std::min(32768, n_frames - f0);
        hipLaunchKernelGGL(mask_pack_kernel, dim3((unsigned)((wpf + 255) / 256), nf), dim3(256), 0, ctx->stream,
                           d_mask + (size_t)f0 * px, d_bits + (size_t)f0 * wpf, d_nonbin, px, wpf);
    }
    SFE_LAUNCH_CHECK(ctx);
    return 0;
}

// d_bits_in != nullptr: the frames arrive as bit streams (sfe_cfar_u8_bits_batch_dev: binary by construction) and
// d_mask is not read
// d_p32 / d_bbox != nullptr: the staged hand-over to the resident cloud filters (float2 points + bounding boxes, see CfBBox);
// want64 = 0 then leaves d_pts unwritten for the frames the record path handles (the others still land there first)
static int extract_dev(sfe_ctx *ctx, sfe_geom *g, const uint8_t *d_mask, int n_frames, long long cap,
                       long long *d_rc, double *d_pts, int32_t *d_counts, const uint32_t *d_bits_in = nullptr,
                       float2 *d_p32 = nullptr, CfBBox *d_bbox = nullptr, bool want64 = true)
{
    const int crows = g->cart_rows, wpr = g->words_per_row;
    static const int chunk = getenv("SFE_EXTRACT_CHUNK") ? std::max(1, atoi(getenv("SFE_EXTRACT_CHUNK"))) : 1024; // frames per pass: bounds the bitmap scratch (0.25 MB per frame), fewer passes = fewer launches
    const size_t bm_bytes = (size_t)chunk * crows * wpr * sizeof(unsigned long long);
    unsigned long long *d_bm = (unsigned long long *)sfe_scratch(ctx, 39, bm_bytes); // (a slot of its own: its contents outlive the call, see self_clean)
    int32_t *d_rcnt = (int32_t *)sfe_scratch(ctx, 5, (size_t)chunk * crows * 4);
    int32_t *d_roff = (int32_t *)sfe_scratch(ctx, 6, (size_t)chunk * crows * 4);
    if (!d_bm || !d_rcnt || !d_roff)
        return SFE_ERR_HIP;
    const long long px = (long long)g->polar_rows * g->polar_cols;
    const long long wpf = (px + 31) / 32 + 1; // +1 pad word: a tap's word index may be one past the last row
    uint32_t *d_bits_own = d_bits_in ? nullptr : (uint32_t *)sfe_scratch(ctx, 10, (size_t)chunk * wpf * 4);
    int32_t *d_nonbin = (int32_t *)sfe_scratch(ctx, 11, (size_t)chunk * 4);
    if ((!d_bits_in && !d_bits_own) || !d_nonbin)
        return SFE_ERR_HIP;
    SFE_HIP(ctx, hipFuncSetAttribute((const void *)extract_bits_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     g->lds_bytes));
    // word form of the expansion: a list of the non-empty bitmap words per frame (at most one per stored point)
    static const bool points_form = getenv("SFE_EXPAND_POINTS") != nullptr; // A/B: lane-per-point expansion
    static const bool nosc = getenv("SFE_NO_SELF_CLEAN") != nullptr;         // A/B: memset of the bitmap per batch
    const long long words_pf = (long long)crows * wpr;
    const size_t tab_bytes = ((size_t)crows + g->cart_cols) * sizeof(double);
    const bool use_words = !points_form && cap > 0 && std::min(words_pf, cap) <= (1ll << 24) && cap < (1ll << 31) &&
                           wpr <= 64 && crows < 32768 && tab_bytes <= 144 * 1024 && g->d_ytab && g->d_xtab;
    const int list_cap = use_words ? (int)std::min(words_pf, cap) : 0;
    int4 *d_wlist = nullptr;
    int32_t *d_wlist_n = nullptr, *d_ovf = nullptr; // d_ovf: [0] number of queued frames, [1..] their indices
    if (use_words) {
        const int nfc = std::min(chunk, n_frames);
        d_wlist = (int4 *)sfe_scratch(ctx, 42, (size_t)nfc * list_cap * sizeof(int4));
        d_wlist_n = (int32_t *)sfe_scratch(ctx, 43, (size_t)nfc * 4);
        d_ovf = (int32_t *)sfe_scratch(ctx, 44, ((size_t)nfc + 1) * 4);
        if (!d_wlist || !d_wlist_n || !d_ovf)
            return SFE_ERR_HIP;
        SFE_HIP(ctx, hipFuncSetAttribute((const void *)extract_expand_words_kernel,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)tab_bytes));
    }
    // record path (round 5; default for bit-stream batches): no canvas bitmap for the frames that fit its capacities
    const int rec_cap_env = getenv("SFE_EXTRACT_REC_CAP") ? atoi(getenv("SFE_EXTRACT_REC_CAP")) : 0; // (read per call: the tests
    const int capw_env = getenv("SFE_EXTRACT_CAPW") ? atoi(getenv("SFE_EXTRACT_CAPW")) : 0;         //  force the hand-back with them)
    const int rec_cap = std::max(1, std::min(ME_THREADS * ME_RPT, rec_cap_env > 0 ? rec_cap_env : ME_THREADS * ME_RPT));
    const int capw = (int)std::max<long long>(1, std::min<long long>(capw_env > 0 ? capw_env : 4096, std::min(words_pf, std::max<long long>(cap, 1))));
    const bool me_narrow = words_pf < 65536 && cap < 65536;
    const size_t me_lds = tab_bytes + (size_t)capw * 8 + ((size_t)(words_pf + 31) / 32) * 8 + ((size_t)capw * 2 + 1) * (me_narrow ? 2 : 4) + 16;
    const bool records = use_words && d_bits_in && ctx->extract_variant == 0 && g->d_inv_off != nullptr &&
                         (g->polar_cols & 31) == 0 && g->polar_rows < 65536 && g->polar_cols < 65536 && me_lds <= 150 * 1024 &&
                         words_pf < (1ll << 31);
    int4 *d_rec = nullptr;
    int32_t *d_rec_n = nullptr, *d_ovf_flag = nullptr;
    if (records)
        SFE_HIP(ctx, hipFuncSetAttribute(me_narrow ? (const void *)extract_merge_expand_kernel<uint16_t>
                                                   : (const void *)extract_merge_expand_kernel<uint32_t>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)me_lds));
    // bytes of the bitmap scratch known to be zero (the state is dropped for the duration of the call: an error return
    // leaves it unknown)
    size_t clean_bytes = ctx->bm_clean_ptr == (void *)d_bm ? ctx->bm_clean_bytes : 0;
    ctx->bm_clean_ptr = nullptr;
    ctx->bm_clean_bytes = 0;
    for (int f0 = 0; f0 < n_frames; f0 += chunk) {
        const int nf = std::min(chunk, n_frames - f0);
        const uint8_t *m = d_bits_in ? nullptr : d_mask + (size_t)f0 * g->polar_rows * g->polar_cols;
        const uint32_t *d_bits = d_bits_in ? d_bits_in + (size_t)f0 * wpf : d_bits_own;
        const int word_groups = g->word_groups;
        const int tiles = g->tiles_per_frame;
        SFE_HIP(ctx, hipMemsetAsync(d_nonbin, 0, (size_t)nf * 4, ctx->stream));
        if (!d_bits_in)
            hipLaunchKernelGGL(mask_pack_kernel, dim3((unsigned)((wpf + 255) / 256), nf), dim3(256), 0, ctx->stream, m,
                               d_bits_own, d_nonbin, px, wpf);
        // binary frames through the inverse map ((row << 16 | column) list entries: images up to 65 535 x 65 535)
        const bool gather = g->d_inv_off != nullptr && (g->polar_cols & 31) == 0 && ctx->extract_variant != 1 &&
                            g->polar_rows < 65536 && g->polar_cols < 65536;
        // The bitmap cleans up after itself on this path (binary frames through the list kernel, word-list expansion
        // with a capacity: extract_expand_words_kernel clears the words it expands, extract_clean_queued_kernel the
        // frames without a list): the memset is only needed when the scratch is new or another path (or a failed
        // call) has left bits behind.
        const bool self_clean = gather && d_bits_in && use_words && cap > 0 && !nosc;
        const size_t bm_need = (size_t)nf * crows * wpr * sizeof(unsigned long long);
        if (!self_clean)
            clean_bytes = 0; // (this pass leaves its bits in the bitmap)
        if (gather) {
            if (!(self_clean && clean_bytes >= bm_need))
                SFE_HIP(ctx, hipMemsetAsync(d_bm, 0, bm_need, ctx->stream));
            if (self_clean)
                clean_bytes = std::max(clean_bytes, bm_need);
            // workgroups per frame: enough of them to fill the device with a few frames, few enough that a
            // workgroup's list holds several rounds of 256 set pixels when there are many
            static const int sg_slices = getenv("SFE_SG_SLICES") ? atoi(getenv("SFE_SG_SLICES")) : 0;
            static const int sg_piece_env = getenv("SFE_SG_PIECE") ? std::min(8, std::max(0, atoi(getenv("SFE_SG_PIECE")))) : -1;
            // 8192 workgroups per 512 frames measured best (16: 0.259 ms per 512 frames, 8: 0.274, 4: 0.36 -- a workgroup's
            // rounds of 256 set pixels wait for their loads one after the other), in pieces of 1024 words when the frame
            // has that many per workgroup (64 rows of 512 beams: a canvas word collects its bits from neighbouring rows,
            // so whole bands keep the table's words to one workgroup; 0.280 -> 0.259)
            const long long nwords = (long long)g->polar_rows * (g->polar_cols >> 5);
            // (record path: twice the workgroups per frame -- a workgroup's table of 1024 canvas words is its record region, and
            // at 8 workgroups per frame the densest bands of the bench's frames filled it: 77 spilled words per frame, each a
            // returning atomic, 2 % of the frames handed back; profiles/r05_extract_records_stats.txt.
            // Measured: 256 frames per launch 59.1 us with 32 workgroups per frame, 72.7 with 64; 512 frames 0.147 ms with 16, 0.165
            // with 32; 1024 frames 0.256 ms with 16.  So: 8192 workgroups per launch, but between 16 and 32 per frame for batches.)
            int slices = sg_slices > 0 ? sg_slices : std::max(2, std::min(64, 8192 / std::max(nf, 1)));
            if (sg_slices <= 0 && records && nf >= 64)
                slices = std::max(16, std::min(32, slices));
            slices = (int)std::max<long long>(1, std::min<long long>(slices, (nwords + 63) / 64));
            int sg_piece = sg_piece_env >= 0 ? sg_piece_env : 4;
            while (sg_piece_env < 0 && sg_piece > 0 && (nwords >> (6 + sg_piece)) < slices)
                --sg_piece;
            const bool no_compact = getenv("SFE_EXTRACT_NO_COMPACT") != nullptr; // A/B: the 8-byte entries of round 3 (read per call)
            const bool c4 = g->d_inv_c4 && !no_compact;
            const int32_t *p_off = c4 ? reinterpret_cast<const int32_t *>(g->d_inv_ob) : g->d_inv_off;
            const uint2 *p_ent = c4 ? reinterpret_cast<const uint2 *>(g->d_inv_c4) : g->d_inv_lut;
            d_ovf_flag = nullptr;
            if (records && slices <= 64) { // (s_rp of the merge kernel holds 64 regions + the spill region)
                // records -> points without a canvas; the frames that do not fit are flagged and go through the canvas kernels below.
                // Per frame: `slices` regions of SG_TAB record slots + the spill region; counts per region, then the flags
                d_rec = (int4 *)sfe_scratch(ctx, 61, (size_t)nf * ((size_t)slices * SG_TAB + SG_SPILL) * sizeof(int4));
                d_rec_n = (int32_t *)sfe_scratch(ctx, 62, (size_t)nf * ((size_t)slices + 2) * 4);
                if (!d_rec || !d_rec_n)
                    return SFE_ERR_HIP;
                d_ovf_flag = d_rec_n + (size_t)nf * (slices + 1);
                SFE_HIP(ctx, hipMemsetAsync(d_rec_n, 0, (size_t)nf * ((size_t)slices + 2) * 4, ctx->stream));
                if (c4)
                    hipLaunchKernelGGL((extract_gather_kernel<true, true>), dim3((unsigned)slices, nf), dim3(256), 0, ctx->stream,
                                       d_bits, d_nonbin, p_off, p_ent, d_bm, g->polar_rows, g->polar_cols, crows, wpr, wpf, sg_piece,
                                       d_rec, d_rec_n, rec_cap, (const int32_t *)nullptr);
                else
                    hipLaunchKernelGGL((extract_gather_kernel<false, true>), dim3((unsigned)slices, nf), dim3(256), 0, ctx->stream,
                                       d_bits, d_nonbin, p_off, p_ent, d_bm, g->polar_rows, g->polar_cols, crows, wpr, wpf, sg_piece,
                                       d_rec, d_rec_n, rec_cap, (const int32_t *)nullptr);
                long long *rc_f = d_rc ? d_rc + (size_t)f0 * cap * 2 : nullptr;
                double *pts_f = (d_pts && (want64 || !d_p32)) ? d_pts + (size_t)f0 * cap * 2 : nullptr;
                float2 *p32_f = d_p32 ? d_p32 + (size_t)f0 * cap : nullptr;
                CfBBox *bb_f = d_bbox ? d_bbox + f0 : nullptr;
                if (me_narrow)
                    hipLaunchKernelGGL(extract_merge_expand_kernel<uint16_t>, dim3(nf), dim3(ME_THREADS), me_lds, ctx->stream, d_rec,
                                       d_rec_n, rec_cap, slices, d_counts + f0, d_ovf_flag, rc_f, pts_f, cap, crows, g->cart_cols, wpr,
                                       capw, g->d_ytab, g->d_xtab, p32_f, bb_f);
                else
                    hipLaunchKernelGGL(extract_merge_expand_kernel<uint32_t>, dim3(nf), dim3(ME_THREADS), me_lds, ctx->stream, d_rec,
                                       d_rec_n, rec_cap, slices, d_counts + f0, d_ovf_flag, rc_f, pts_f, cap, crows, g->cart_cols, wpr,
                                       capw, g->d_ytab, g->d_xtab, p32_f, bb_f);
            }
            if (c4)
                hipLaunchKernelGGL((extract_gather_kernel<true, false>), dim3((unsigned)slices, nf), dim3(256), 0, ctx->stream, d_bits,
                                   d_nonbin, p_off, p_ent, d_bm, g->polar_rows, g->polar_cols, crows, wpr, wpf, sg_piece,
                                   (int4 *)nullptr, (int32_t *)nullptr, 0, (const int32_t *)d_ovf_flag);
            else
                hipLaunchKernelGGL((extract_gather_kernel<false, false>), dim3((unsigned)slices, nf), dim3(256), 0, ctx->stream, d_bits,
                                   d_nonbin, p_off, p_ent, d_bm, g->polar_rows, g->polar_cols, crows, wpr, wpf, sg_piece,
                                   (int4 *)nullptr, (int32_t *)nullptr, 0, (const int32_t *)d_ovf_flag);
        }
        if (!(gather && d_bits_in)) // bit streams are binary: nothing is left for the general pass
        hipLaunchKernelGGL(extract_bits_kernel, dim3((unsigned)((gather ? std::min(nf, 8) : nf) * tiles)), dim3(256),
                           g->lds_bytes, ctx->stream, m, d_bits, d_nonbin, (const uint32_t *)g->d_code, g->d_span,
                           g->d_tile_rows, d_bm, g->polar_rows, g->polar_cols, g->rcp, crows, g->cart_cols, wpr, word_groups,
                           tiles, wpf, gather ? 1 : 0, nf);
        if (use_words)
            SFE_HIP(ctx, hipMemsetAsync(d_ovf, 0, 4, ctx->stream));
        const size_t scan_lds = sizeof(int) * ((size_t)crows + SCAN_THREADS + (use_words ? (words_pf + 63) / 64 + 4 * SCAN_LDS_LIST : 0));
        if (scan_lds > 48 * 1024)
            SFE_HIP(ctx, hipFuncSetAttribute((const void *)extract_scan_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)scan_lds));
        hipLaunchKernelGGL(extract_scan_kernel, dim3(nf), dim3(SCAN_THREADS), scan_lds,
                           ctx->stream, d_bm, d_rcnt, d_roff, d_counts + f0, crows, wpr, d_wlist, d_wlist_n,
                           use_words ? list_cap : 0, cap, d_ovf, d_ovf ? d_ovf + 1 : nullptr, (const int32_t *)d_ovf_flag);
        if (cap > 0) {
            long long *rc_f = d_rc ? d_rc + (size_t)f0 * cap * 2 : nullptr;
            double *pts_f = d_pts ? d_pts + (size_t)f0 * cap * 2 : nullptr;
            if (use_words) {
                static const int expand_wg_env = getenv("SFE_EXPAND_WG") ? std::max(1, atoi(getenv("SFE_EXPAND_WG"))) : 0;
                // a workgroup first fetches the two metre tables (23 KB for config A): about a thousand workgroups
                // per launch, all resident at once (8 per frame measured 69 us per 512 frames, 2 per frame 53)
                const int expand_wg = expand_wg_env ? expand_wg_env : std::max(1, std::min(EXPAND_WG, 1024 / std::max(nf, 1)));
                hipLaunchKernelGGL(extract_expand_words_kernel, dim3(expand_wg, nf), dim3(256), tab_bytes, ctx->stream,
                                   d_wlist, d_wlist_n, list_cap, rc_f, pts_f, cap, crows, g->cart_cols, g->d_ytab, g->d_xtab,
                                   self_clean ? d_bm : nullptr, wpr, (const int32_t *)d_ovf_flag);
                // frames above the capacity (queued by the scan kernel; normally none): their first cap points
                hipLaunchKernelGGL(extract_expand_kernel, dim3((unsigned)((cap + 255) / 256), 1), dim3(256),
                                   sizeof(int32_t) * (size_t)crows, ctx->stream, d_bm, d_roff, d_counts + f0, rc_f, pts_f, cap,
                                   crows, g->cart_cols, wpr, g->width, g->height, d_ovf, d_ovf + 1);
                if (self_clean)
                    hipLaunchKernelGGL(extract_clean_queued_kernel, dim3(64), dim3(256), 0, ctx->stream, d_bm,
                                       (long long)crows * wpr, d_ovf, d_ovf + 1);
            } else {
                hipLaunchKernelGGL(extract_expand_kernel, dim3((unsigned)((cap + 255) / 256), nf), dim3(256),
                                   sizeof(int32_t) * (size_t)crows, ctx->stream, d_bm, d_roff, d_counts + f0, rc_f, pts_f, cap,
                                   crows, g->cart_cols, wpr, g->width, g->height, (const int32_t *)nullptr,
                                   (const int32_t *)nullptr);
            }
            if (d_p32) // staged: the frames the canvas kernels just expanded (flagged by the record path, or all of them)
                hipLaunchKernelGGL(extract_stage_fallback_kernel, dim3((unsigned)std::min(256, (nf + 3) / 4)), dim3(256), 0, ctx->stream,
                                   (const double *)pts_f, (const int32_t *)(d_counts + f0), cap, (const int32_t *)d_ovf_flag,
                                   d_p32 + (size_t)f0 * cap, d_bbox + f0, nf);
        }
    }
    SFE_LAUNCH_CHECK(ctx);
    ctx->bm_clean_ptr = (void *)d_bm; // every launch went through: that much of the bitmap is zero again when they have run
    ctx->bm_clean_bytes = clean_bytes;
    return 0;
}

static inline int cv_round_f(float v)
{
    return (int)lrintf(v); // nearest-even in the default rounding mode == cvRound
}

extern "C" {

int sfe_extract_set_tuning(sfe_ctx *ctx, int variant)
{
    if (!ctx)
        return SFE_ERR_ARG;
    SFE_ARG(ctx, variant >= 0 && variant <= 2);
    ctx->extract_variant = variant;
    return 0;
}

int sfe_geom_create(sfe_ctx *ctx, const float *map_x, const float *map_y, int cart_rows, int cart_cols,
                    int polar_rows, int polar_cols, double width, double height, sfe_geom **out)
{
    if (int rc = sfe_use(ctx))
        return rc;
    SFE_ARG(ctx, out && map_x && map_y);
    *out = nullptr;
    SFE_ARG(ctx, cart_rows > 0 && cart_cols > 0 && polar_rows > 0 && polar_cols > 0);
    if ((long long)(polar_rows + 1) * (polar_cols + 1) > (1ll << 22))
        return sfe_set_err(ctx, SFE_ERR_ARG, "polar image %dx%d too large for the packed remap code (max 2^22 px)",
                           polar_rows, polar_cols);
    const size_t n = (size_t)cart_rows * cart_cols;
    std::vector<uint32_t> code(n);
    std::vector<int32_t> span(2 * (size_t)cart_rows);
    for (int r = 0; r < cart_rows; ++r) {
        int first = cart_cols, last = 0;
        for (int c = 0; c < cart_cols; ++c) {
            const size_t o = (size_t)r * cart_cols + c;
            const float mx = map_x[o] * 32.0f, my = map_y[o] * 32.0f;
            uint32_t cd = SFE_CODE_NONE;
            // |coordinate| < 2^20 px keeps cvRound and the >>5 well defined; anything larger is far outside
            if (std::fabs(mx) < 3.3e7f && std::fabs(my) < 3.3e7f) {
                const int sx = cv_round_f(mx), sy = cv_round_f(my);
                const int ix = sx >> 5, iy = sy >> 5;
                if (ix >= -1 && ix < polar_cols && iy >= -1 && iy < polar_rows) {
                    const uint32_t lin = (uint32_t)(iy + 1) * (uint32_t)(polar_cols + 1) + (uint32_t)(ix + 1);
                    cd = (lin << 10) | (uint32_t)((sy & 31) << 5) | (uint32_t)(sx & 31);
                    if (c < first)
                        first = c;
                    last = c + 1;
                }
            }
            code[o] = cd;
        }
        if (first > last)
            first = last = 0;
        span[2 * r] = first;
        span[2 * r + 1] = last;
    }
    sfe_geom *g = new sfe_geom();
    g->ctx = ctx;
    g->cart_rows = cart_rows;
    g->cart_cols = cart_cols;
    g->polar_rows = polar_rows;
    g->polar_cols = polar_cols;
    g->width = width;
    g->height = height;
    g->words_per_row = (cart_cols + 63) / 64;
    {   // reciprocal of (polar_cols+1) for the in-kernel divide; verify exactness over the whole range
        const unsigned d = (unsigned)(polar_cols + 1);
        g->rcp = (unsigned)((0x100000000ull + d - 1) / d);
        const unsigned lin_max = (unsigned)(polar_rows + 1) * d;
        for (unsigned lin = 0; lin < lin_max; ++lin)
            if ((unsigned)(((unsigned long long)lin * g->rcp) >> 32) != lin / d) {
                sfe_geom_destroy(g);
                return sfe_set_err(ctx, SFE_ERR_ARG, "reciprocal divide not exact for polar_cols=%d", polar_cols);
            }
    }
    // per tile of the canvas (4 ballot words x EXTRACT_RG rows): the polar rows its valid pixels tap
    g->word_groups = (g->words_per_row + 3) / 4;
    const int row_groups = (cart_rows + EXTRACT_RG - 1) / EXTRACT_RG;
    g->tiles_per_frame = g->word_groups * row_groups;
    std::vector<int32_t> tile_rows(2 * (size_t)g->tiles_per_frame);
    long long max_words = 2;
    for (int t = 0; t < g->tiles_per_frame; ++t) {
        const int wg = t % g->word_groups, rg = t / g->word_groups;
        int ylo = polar_rows, yhi = -1;
        for (int r = rg * EXTRACT_RG; r < std::min((rg + 1) * EXTRACT_RG, cart_rows); ++r)
            for (int c = wg * 256; c < std::min((wg + 1) * 256, cart_cols); ++c) {
                const uint32_t cd = code[(size_t)r * cart_cols + c];
                if (cd == SFE_CODE_NONE)
                    continue;
                const int iy = (int)((cd >> 10) / (uint32_t)(polar_cols + 1)) - 1;
                ylo = std::min(ylo, std::max(iy, 0));
                yhi = std::max(yhi, std::min(iy + 1, polar_rows - 1));
            }
        tile_rows[2 * t] = ylo;
        tile_rows[2 * t + 1] = yhi;
        if (ylo <= yhi)
            max_words = std::max(max_words, (long long)(yhi - ylo + 1) * ((polar_cols >> 5) | 1));
    }
    g->lds_bytes = (int)(max_words * 4);
    if (g->lds_bytes > 150 * 1024) {
        sfe_geom_destroy(g);
        return sfe_set_err(ctx, SFE_ERR_ARG, "geometry needs %d bytes of LDS per canvas tile (max 153600)", g->lds_bytes);
    }
    // inverse map: polar pixel -> canvas pixels tapping it with a non-zero weight (same weights as the kernels)
    std::vector<int32_t> inv_off((size_t)polar_rows * polar_cols + 1, 0);
    std::vector<uint2> inv_ent; // {canvas pixel, code[canvas pixel]}
    if (n < (1ull << 32)) {
        auto each_tap = [&](size_t o, auto &&fn) {
            const uint32_t cd = code[o];
            if (cd == SFE_CODE_NONE)
                return;
            const uint32_t lin = cd >> 10;
            const int fy = (int)((cd >> 5) & 31u), fx = (int)(cd & 31u);
            const int iy = (int)(lin / (uint32_t)(polar_cols + 1)) - 1, ix = (int)(lin % (uint32_t)(polar_cols + 1)) - 1;
            int wgt[4] = {(32 - fy) * (32 - fx), (32 - fy) * fx, fy * (32 - fx), fy * fx};
            if ((fx | fy) == 0)
                wgt[3] = 1;
            for (int t = 0; t < 4; ++t) {
                const int y = iy + (t >> 1), x = ix + (t & 1);
                if (wgt[t] > 0 && y >= 0 && y < polar_rows && x >= 0 && x < polar_cols)
                    fn((size_t)y * polar_cols + x);
            }
        };
        for (size_t o = 0; o < n; ++o)
            each_tap(o, [&](size_t pi) { ++inv_off[pi + 1]; });
        for (size_t i = 1; i < inv_off.size(); ++i)
            inv_off[i] += inv_off[i - 1];
        inv_ent.resize((size_t)inv_off.back());
        std::vector<int32_t> cur(inv_off.begin(), inv_off.end() - 1);
        // entry = {bit index of the canvas pixel inside a frame's bitmap (row * words_per_row * 64 + col), its remap code}:
        // the gather kernel sets that bit without dividing by the canvas width
        const bool bit_index_fits = (unsigned long long)cart_rows * g->words_per_row * 64ull < (1ull << 32) - 1;
        if (!bit_index_fits) {
            sfe_geom_destroy(g);
            return sfe_set_err(ctx, SFE_ERR_ARG, "canvas %dx%d too large for the inverse map's 32-bit bit index", cart_rows, cart_cols);
        }
        for (size_t o = 0; o < n; ++o)
            each_tap(o, [&](size_t pi) {
                const size_t row = o / (size_t)cart_cols, col = o - row * (size_t)cart_cols;
                inv_ent[(size_t)cur[pi]++] = make_uint2((uint32_t)(row * (size_t)g->words_per_row * 64 + col), code[o]);
            });
        // The same entries with the blend decided in advance (extract_gather_kernel).  Whether a canvas pixel reached from
        // one of its set taps is a detection -- and whether THIS tap is the one that reports it -- depends on the entry
        // (the tap's place among the four, the two 5-bit fractions) and on the four mask bits of the taps only: 16 cases,
        // evaluated here with the kernels' arithmetic.  y = table (bit p: taps v00 v01 v10 v11 = bits 0..3 of p) |
        // shift << 16, shift = position of tap 00 inside the 3 x 3 neighbourhood of the set pixel (bit 3 * (dy + 1) + dx + 1).
        std::vector<uint2> inv_lut(inv_ent.size());
        for (size_t pi = 0; pi + 1 < inv_off.size(); ++pi) {
            const int py = (int)(pi / (size_t)polar_cols), px = (int)(pi - (size_t)py * polar_cols);
            for (int32_t j = inv_off[pi]; j < inv_off[pi + 1]; ++j) {
                const uint32_t cd = inv_ent[(size_t)j].y, lin = cd >> 10;
                const int fy = (int)((cd >> 5) & 31u), fx = (int)(cd & 31u);
                const int iy = (int)(lin / (uint32_t)(polar_cols + 1)) - 1, ix = (int)(lin % (uint32_t)(polar_cols + 1)) - 1;
                const int ry = iy - py + 1, rx = ix - px + 1; // 0 or 1
                int w00 = (32 - fy) * (32 - fx) * 32, w01 = (32 - fy) * fx * 32, w10 = fy * (32 - fx) * 32, w11 = fy * fx * 32;
                if ((fx | fy) == 0) {
                    w00 = 32767;
                    w11 = 1;
                }
                const int t_src = (1 - ry) * 2 + (1 - rx);
                uint32_t lut = 0;
                for (int p = 0; p < 16; ++p) {
                    const int v00 = p & 1, v01 = (p >> 1) & 1, v10 = (p >> 2) & 1, v11 = (p >> 3) & 1;
                    const int acc = w00 * v00 + w01 * v01 + w10 * v10 + w11 * v11;
                    const int first = (v00 && w00) ? 0 : (v01 && w01) ? 1 : (v10 && w10) ? 2 : 3;
                    if (((acc + 16384) >> 15) != 0 && t_src == first)
                        lut |= 1u << p;
                }
                inv_lut[(size_t)j] = make_uint2(inv_ent[(size_t)j].x, lut | ((uint32_t)(ry * 3 + rx) << 16));
            }
        }
        // Round 4: the same table in 4 bytes per entry, relative to a base bit index per polar pixel, without the entries
        // whose table is 0 (extract_gather_kernel<true>).  Falls back to the 8-byte entries when a pixel's candidates span
        // more than 127 canvas rows / columns (no sonar fan does).
        {
            const size_t npix = inv_off.size() - 1;
            std::vector<uint2> ob(npix + 1);
            std::vector<uint32_t> c4;
            c4.reserve(inv_lut.size());
            const unsigned long long rowbits = (unsigned long long)g->words_per_row * 64ull;
            bool fits = true;
            for (size_t pi = 0; pi < npix && fits; ++pi) {
                unsigned long long rmin = ~0ull, cmin = ~0ull;
                for (int32_t j = inv_off[pi]; j < inv_off[pi + 1]; ++j)
                    if (inv_lut[(size_t)j].y & 0xFFFFu) {
                        rmin = std::min<unsigned long long>(rmin, inv_lut[(size_t)j].x / rowbits);
                        cmin = std::min<unsigned long long>(cmin, inv_lut[(size_t)j].x % rowbits);
                    }
                if (rmin == ~0ull)
                    rmin = cmin = 0;
                ob[pi] = make_uint2((uint32_t)c4.size(), (uint32_t)(rmin * rowbits + cmin));
                for (int32_t j = inv_off[pi]; j < inv_off[pi + 1]; ++j) {
                    const uint2 e = inv_lut[(size_t)j];
                    if (!(e.y & 0xFFFFu))
                        continue;
                    const unsigned long long dy = e.x / rowbits - rmin, dx = e.x % rowbits - cmin;
                    const unsigned shift = e.y >> 16, sc = shift >= 3 ? shift - 1 : shift; // ry * 3 + rx -> ry * 2 + rx
                    if (dy > 127 || dx > 127 || c4.size() >= 0xFFFFFFF0ull) {
                        fits = false;
                        break;
                    }
                    c4.push_back((e.y & 0xFFFFu) | (sc << 16) | ((uint32_t)dx << 18) | ((uint32_t)dy << 25));
                }
            }
            if (fits) {
                ob[npix] = make_uint2((uint32_t)c4.size(), 0u);
                c4.resize(c4.size() + 8, 0u); // (the kernel reads entries in fours, two reads ahead)
                ob.resize(ob.size() + 1, make_uint2((uint32_t)c4.size(), 0u)); // (16-byte reads of {offset, base} pairs)
                if (hipMalloc((void **)&g->d_inv_ob, ob.size() * sizeof(uint2)) != hipSuccess ||
                    hipMalloc((void **)&g->d_inv_c4, c4.size() * 4) != hipSuccess ||
                    hipMemcpy(g->d_inv_ob, ob.data(), ob.size() * sizeof(uint2), hipMemcpyHostToDevice) != hipSuccess ||
                    hipMemcpy(g->d_inv_c4, c4.data(), c4.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
                    sfe_geom_destroy(g);
                    return sfe_set_err(ctx, SFE_ERR_HIP, "compact inverse remap table upload failed");
                }
            }
        }
        inv_lut.resize(inv_lut.size() + 2, make_uint2(0u, 0u)); // (the kernel reads entries in pairs)
        if (hipMalloc((void **)&g->d_inv_lut, std::max<size_t>(inv_lut.size(), 1) * sizeof(uint2)) != hipSuccess ||
            (!inv_lut.empty() &&
             hipMemcpy(g->d_inv_lut, inv_lut.data(), inv_lut.size() * sizeof(uint2), hipMemcpyHostToDevice) != hipSuccess)) {
            sfe_geom_destroy(g);
            return sfe_set_err(ctx, SFE_ERR_HIP, "inverse remap table upload failed");
        }
        // (the {canvas pixel, remap code} entries themselves stay on the host: only round 2's row-block kernel read them)
        if (hipMalloc((void **)&g->d_inv_off, inv_off.size() * 4) != hipSuccess ||
            hipMemcpy(g->d_inv_off, inv_off.data(), inv_off.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
            sfe_geom_destroy(g);
            return sfe_set_err(ctx, SFE_ERR_HIP, "inverse remap table upload failed");
        }
    }
    {
        std::vector<double> ytab((size_t)cart_rows), xtab((size_t)cart_cols);
        const double half_cols = cart_cols / 2.;
        for (int r = 0; r < cart_rows; ++r)
            ytab[(size_t)r] = (-1 * ((double)r / (double)cart_rows) * height) + height; // feature_extraction.py:237
        for (int c = 0; c < cart_cols; ++c) {
            double x = (double)c - half_cols;                                            // feature_extraction.py:236
            xtab[(size_t)c] = (-1 * ((x / half_cols) * (width / 2.)));
        }
        if (hipMalloc((void **)&g->d_ytab, ytab.size() * 8) != hipSuccess ||
            hipMalloc((void **)&g->d_xtab, xtab.size() * 8) != hipSuccess ||
            hipMemcpy(g->d_ytab, ytab.data(), ytab.size() * 8, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(g->d_xtab, xtab.data(), xtab.size() * 8, hipMemcpyHostToDevice) != hipSuccess) {
            sfe_geom_destroy(g);
            return sfe_set_err(ctx, SFE_ERR_HIP, "px->m table upload failed");
        }
    }
    if (hipMalloc((void **)&g->d_code, n * 4) != hipSuccess ||
        hipMalloc((void **)&g->d_span, span.size() * 4) != hipSuccess ||
        hipMalloc((void **)&g->d_tile_rows, tile_rows.size() * 4) != hipSuccess) {
        sfe_geom_destroy(g);
        return sfe_set_err(ctx, SFE_ERR_HIP, "hipMalloc for geometry failed");
    }
    if (hipMemcpy(g->d_code, code.data(), n * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(g->d_span, span.data(), span.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(g->d_tile_rows, tile_rows.data(), tile_rows.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
        sfe_geom_destroy(g);
        return sfe_set_err(ctx, SFE_ERR_HIP, "hipMemcpy for geometry failed");
    }
    *out = g;
    return 0;
}

void sfe_geom_destroy(sfe_geom *g)
{
    if (!g)
        return;
    if (g->ctx) {
        (void)hipSetDevice(g->ctx->device);
        (void)hipStreamSynchronize(g->ctx->stream);
    }
    if (g->d_code)
        (void)hipFree(g->d_code);
    if (g->d_span)
        (void)hipFree(g->d_span);
    if (g->d_tile_rows)
        (void)hipFree(g->d_tile_rows);
    if (g->d_inv_off)
        (void)hipFree(g->d_inv_off);
    if (g->d_inv_lut)
        (void)hipFree(g->d_inv_lut);
    if (g->d_inv_ob)
        (void)hipFree(g->d_inv_ob);
    if (g->d_inv_c4)
        (void)hipFree(g->d_inv_c4);
    if (g->d_ytab)
        (void)hipFree(g->d_ytab);
    if (g->d_xtab)
        (void)hipFree(g->d_xtab);
    delete g;
}

int sfe_remap_u8(sfe_ctx *ctx, sfe_geom *g, const uint8_t *src, uint8_t *dst)
{
    if (int rc = sfe_use(ctx))
        return rc;
    SFE_ARG(ctx, g && src && dst && g->ctx == ctx);
    const size_t np = (size_t)g->polar_rows * g->polar_cols, nc = (size_t)g->cart_rows * g->cart_cols;
    uint8_t *d_src = (uint8_t *)sfe_scratch(ctx, 0, np);
    uint8_t *d_dst = (uint8_t *)sfe_scratch(ctx, 3, nc);
    if (!d_src || !d_dst)
        return SFE_ERR_HIP;
    SFE_HIP(ctx, hipMemcpyAsync(d_src, src, np, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(remap_u8_kernel, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, ctx->stream, d_src,
                       (const uint32_t *)g->d_code, d_dst, g->polar_rows, g->polar_cols, g->rcp, (long long)nc, 1);
    SFE_LAUNCH_CHECK(ctx);
    SFE_HIP(ctx, hipMemcpyAsync(dst, d_dst, nc, hipMemcpyDeviceToHost, ctx->stream));
    SFE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int sfe_remap_u8_dev(sfe_ctx *ctx, sfe_geom *g, const uint8_t *d_src, uint8_t *d_dst)
{
    if (int rc = sfe_use(ctx))
        return rc;
    SFE_ARG(ctx, g && d_src && d_dst && g->ctx == ctx);
    const size_t nc = (size_t)g->cart_rows * g->cart_cols;
    hipLaunchKernelGGL(remap_u8_kernel, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, ctx->stream, d_src,
                       (const uint32_t *)g->d_code, d_dst, g->polar_rows, g->polar_cols, g->rcp, (long long)nc, 1);
    SFE_LAUNCH_CHECK(ctx);
    return 0;
}

// OpenCV's COLORMAP_JET (imgproc colormap.cpp, class Jet: "equals the GNU Octave colormap jet"): 256 control values per
// channel sampled from the piecewise-linear ramps
//     r = 4x - 3/2 on [3/8, 5/8), 1 on [5/8, 7/8), -4x + 9/2 from 7/8;   g = 4x - 1/2 on [1/8, 3/8), 1, -4x + 7/2 on [5/8, 7/8);
//     b = 4x + 1/2 below 1/8, 1 on [1/8, 3/8), -4x + 5/2 on [3/8, 5/8);   x = i / 255,
// scaled by 255 and rounded to uint8 (cvRound: half to even).  Every ramp value is k + 1/2 exactly, so here the table is
// evaluated in exact integer arithmetic (twice the value) with the tie rule; OpenCV goes through float32 (a literal table,
// an interp1 at the sample points, convertTo(CV_8U, 255)) whose rounding noise may move an entry by one grey level: a
// float32 emulation of that pipeline differs from this table in 1 of 768 entries (tests/test_oracle_pipeline.py).
// OpenCV is un-vendored and absent: PARITY UNPINNED.
static void jet_lut(uint32_t *lut)
{
    auto q = [](int twice) { // round(twice / 2) half to even, clamped to uint8
        int v = twice >> 1;
        if (twice & 1)
            v += v & 1;
        return (uint32_t)(v < 0 ? 0 : v > 255 ? 255 : v);
    };
    for (int i = 0; i < 256; ++i) {
        // 8i vs 255 k: x >= k/8  <=>  8 i >= 255 k
        const int e = 8 * i;
        const int r2 = (e >= 765 && e < 1275) ? 8 * i - 765 : (e >= 1275 && e < 1785) ? 510 : (e >= 1785) ? -8 * i + 2295 : 0;
        const int g2 = (e >= 255 && e < 765) ? 8 * i - 255 : (e >= 765 && e < 1275) ? 510 : (e >= 1275 && e < 1785) ? -8 * i + 1785 : 0;
        const int b2 = (e < 255) ? 8 * i + 255 : (e >= 255 && e < 765) ? 510 : (e >= 765 && e < 1275) ? -8 * i + 1275 : 0;
        lut[i] = q(b2) | (q(g2) << 8) | (q(r2) << 16);
    }
}

int sfe_colormap_lut(int colormap, uint8_t *lut_bgr)
{
    if (colormap != SFE_COLORMAP_JET || !lut_bgr)
        return SFE_ERR_ARG;
    uint32_t lut[256];
    jet_lut(lut);
    for (int i = 0; i < 256; ++i) {
        lut_bgr[3 * i] = (uint8_t)lut[i];
        lut_bgr[3 * i + 1] = (uint8_t)(lut[i] >> 8);
        lut_bgr[3 * i + 2] = (uint8_t)(lut[i] >> 16);
    }
    return 0;
}

int sfe_remap_u8_colormap_dev(sfe_ctx *ctx, sfe_geom *g, const uint8_t *d_src, int colormap, uint8_t *d_dst_bgr)
{
    if (int rc = sfe_use(ctx))
        return rc;
    SFE_ARG(ctx, g && d_src && d_dst_bgr && g->ctx == ctx && (reinterpret_cast<uintptr_t>(d_dst_bgr) & 3) == 0);
    if (colormap != SFE_COLORMAP_JET)
        return sfe_set_err(ctx, SFE_ERR_ARG, "colour map %d: only cv2.COLORMAP_JET (2) is built (feature_extraction.py:227)", colormap);
    uint32_t *d_lut = (uint32_t *)sfe_scratch(ctx, 51, 1024);
    if (!d_lut)
        return SFE_ERR_HIP;
    uint32_t *h_lut = (uint32_t *)sfe_pinned_begin(ctx, 1024);
    if (!h_lut)
        return SFE_ERR_HIP;
    jet_lut(h_lut);
    SFE_HIP(ctx, hipMemcpyAsync(d_lut, h_lut, 1024, hipMemcpyHostToDevice, ctx->stream));
    if (int rc = sfe_pinned_end(ctx, ctx->stream))
        return rc;
    const long long nc = (long long)g->cart_rows * g->cart_cols;
    hipLaunchKernelGGL(remap_u8_lut_kernel, dim3((unsigned)(((nc + 3) / 4 + 255) / 256)), dim3(256), 0, ctx->stream, d_src,
                       (const uint32_t *)g->d_code, (const uint32_t *)d_lut, reinterpret_cast<uint32_t *>(d_dst_bgr),
                       g->polar_rows, g->polar_cols, g->rcp, nc);
    SFE_LAUNCH_CHECK(ctx);
    return 0;
}

int sfe_remap_u8_colormap(sfe_ctx *ctx, sfe_geom *g, const uint8_t *src, int colormap, uint8_t *dst_bgr)
{
    if (int rc = sfe_use(ctx))
        return rc;
    SFE_ARG(ctx, g && src && dst_bgr && g->ctx == ctx);
    const size_t np = (size_t)g->polar_rows * g->polar_cols, nc = (size_t)g->cart_rows * g->cart_cols;
    uint8_t *d_src = (uint8_t *)sfe_scratch(ctx, 0, np);
    uint8_t *d_dst = (uint8_t *)sfe_scratch(ctx, 3, 3 * nc + 4);
    if (!d_src || !d_dst)
        return SFE_ERR_HIP;
    SFE_HIP(ctx, hipMemcpyAsync(d_src, src, np, hipMemcpyHostToDevice, ctx->stream));
    if (int rc = sfe_remap_u8_colormap_dev(ctx, g, d_src, colormap, d_dst))
        return rc;
    SFE_HIP(ctx, hipMemcpyAsync(dst_bgr, d_dst, 3 * nc, hipMemcpyDeviceToHost, ctx->stream));
    SFE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int sfe_extract_points_batch_dev(sfe_ctx *ctx, sfe_geom *g, const uint8_t *d_mask, int n_frames, int64_t cap,
                                 double *d_pts, int32_t *d_counts)
{
    if (int rc = sfe_use(ctx))
        return rc;
    SFE_ARG(ctx, g && d_mask && d_counts && g->ctx == ctx && n_frames >= 0 && cap >= 0);
    if (n_frames == 0)
        return 0;
    return extract_dev(ctx, g, d_mask, n_frames, cap, nullptr, d_pts, d_counts);
}

int sfe_extract_points_bits_batch_dev(sfe_ctx *ctx, sfe_geom *g, const uint32_t *d_bits, int n_frames, int64_t cap,
                                      double *d_pts, int32_t *d_counts)
{
    if (int rc = sfe_use(ctx))
        return rc;
    SFE_ARG(ctx, g && d_bits && d_counts && g->ctx == ctx && n_frames >= 0 && cap >= 0);
    if ((g->polar_cols & 31) != 0)
        return sfe_set_err(ctx, SFE_ERR_ARG, "bit-stream extraction needs polar_cols %% 32 == 0 (got %d)", g->polar_cols);
    if (n_frames == 0)
        return 0;
    return extract_dev(ctx, g, nullptr, n_frames, cap, nullptr, d_pts, d_counts, d_bits);
}

int sfe_extract_points_bits_staged_dev(sfe_ctx *ctx, sfe_geom *g, const uint32_t *d_bits, int n_frames, int64_t cap,
                                       double *d_pts, int want_points64, int32_t *d_counts)
{
    if (int rc = sfe_use(ctx))
        return rc;
    SFE_ARG(ctx, g && d_bits && d_pts && d_counts && g->ctx == ctx && n_frames >= 0 && cap > 0);
    if ((g->polar_cols & 31) != 0)
        return sfe_set_err(ctx, SFE_ERR_ARG, "bit-stream extraction needs polar_cols %% 32 == 0 (got %d)", g->polar_cols);
    if (cap > CF_MAX_CAP)
        return sfe_set_err(ctx, SFE_ERR_ARG, "sfe_extract_points_bits_staged_dev: cap %lld exceeds %d points per frame",
                           (long long)cap, CF_MAX_CAP);
    ctx->staged_frames = -1;
    if (n_frames == 0)
        return 0;
    float2 *d_p32 = (float2 *)sfe_scratch(ctx, CF_SLOT_P32, sizeof(float2) * (size_t)cap * (size_t)n_frames);
    CfBBox *d_bbox = (CfBBox *)sfe_scratch(ctx, CF_SLOT_BBOX, sizeof(CfBBox) * (size_t)n_frames);
    if (!d_p32 || !d_bbox)
        return SFE_ERR_HIP;
    if (int rc = extract_dev(ctx, g, nullptr, n_frames, cap, nullptr, d_pts, d_counts, d_bits, d_p32, d_bbox, want_points64 != 0))
        return rc;
    ctx->staged_frames = n_frames;
    ctx->staged_cap = cap;
    return 0;
}

int sfe_extract_points(sfe_ctx *ctx, sfe_geom *g, const uint8_t *mask, int64_t cap, int64_t *rc_out,
                       double *pts_out, int64_t *n_out)
{
    if (int rc = sfe_use(ctx))
        return rc;
    SFE_ARG(ctx, g && mask && n_out && g->ctx == ctx && cap >= 0);
    const size_t np = (size_t)g->polar_rows * g->polar_cols;
    uint8_t *d_mask = (uint8_t *)sfe_scratch(ctx, 1, np);
    int32_t *d_count = (int32_t *)sfe_scratch(ctx, 7, 64);
    long long *d_rc = rc_out ? (long long *)sfe_scratch(ctx, 8, (size_t)std::max<int64_t>(cap, 1) * 16) : nullptr;
    double *d_pts = pts_out ? (double *)sfe_scratch(ctx, 9, (size_t)std::max<int64_t>(cap, 1) * 16) : nullptr;
    if (!d_mask || !d_count || (rc_out && !d_rc) || (pts_out && !d_pts))
        return SFE_ERR_HIP;
    SFE_HIP(ctx, hipMemcpyAsync(d_mask, mask, np, hipMemcpyHostToDevice, ctx->stream));
    if (int rc = extract_dev(ctx, g, d_mask, 1, cap, d_rc, d_pts, d_count))
        return rc;
    int32_t n = 0;
    SFE_HIP(ctx, hipMemcpyAsync(&n, d_count, 4, hipMemcpyDeviceToHost, ctx->stream));
    SFE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *n_out = n;
    const size_t m = (size_t)std::min<int64_t>(n, cap);
    if (m && rc_out)
        SFE_HIP(ctx, hipMemcpyAsync(rc_out, d_rc, m * 16, hipMemcpyDeviceToHost, ctx->stream));
    if (m && pts_out)
        SFE_HIP(ctx, hipMemcpyAsync(pts_out, d_pts, m * 16, hipMemcpyDeviceToHost, ctx->stream));
    SFE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (n > cap)
        return sfe_set_err(ctx, SFE_ERR_CAP, "extract_points: %d points exceed capacity %lld", n, (long long)cap);
    return 0;
}

} // extern "C"
