// Device helpers shared by the ICP kernels (sfe_icp.hip: brute-force tile scan, sfe_icp_sweep.hip:
// strip-sweep search).  Every float expression that decides a match or a weight is written with
// explicit IEEE roundings (no contraction) so both kernels and the oracle take the same decisions.
#pragma once
#include "sfe_internal.h"

#ifndef ICP_THREADS
#define ICP_THREADS 1024
#endif
#define ICP_WAVES (ICP_THREADS / 64)
#define ICP_TCAP 8192 // target points resident in LDS (64 KiB as float2)
#define ICP_PB 8      // max source points per lane per pass over the target
#define ICP_CH 16     // target points per chunk of the two-level arg-min
#define ICP_KMAX 16   // max neighbours for the PCA normals
#define ICP_MAX_HIST 64 // transformation history kept for the differential checker
#define ICP_PIVOT_RTOL 1e-10 // Cholesky pivot / diagonal entry below which the point-to-plane system counts as singular

struct IcpJob {
    int src_start, n_src, tgt_start, n_tgt;
    long long scratch_off; // offset (in points) of this job's slice of the NN scratch
    long long nrm_off;     // offset (in points) of this job's slice of the normals scratch
};

__device__ __forceinline__ float f_mul(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float f_add(float a, float b) { return __fadd_rn(a, b); }

// x' = (a*x + b*y) + c with every product/sum rounded to float (Eigen's coefficient product)
__device__ __forceinline__ float affine1(float a, float b, float c, float x, float y)
{
    return f_add(f_add(f_mul(a, x), f_mul(b, y)), c);
}

// v + (v of the lane `SHR` places to the left in the same row of 16 lanes, 0.0 where there is none): DPP
// row_shr, no address register and no LDS crossbar (the __shfl form kept six loop-invariant address
// VGPRs alive, which the 64-VGPR ICP kernels spilled and re-loaded around every reduction)
template <int SHR>
__device__ __forceinline__ double dpp_row_shr_add(double v)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x110 + SHR, 0xf, 0xf, false);
    const int hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x110 + SHR, 0xf, 0xf, false);
    return v + __hiloint2double(hi2, lo2);
}

// sum over the 64 lanes, same value in every lane; fixed order (deterministic): inclusive scan inside
// each row of 16 lanes, then the four row totals left to right
__device__ __forceinline__ double wave_sum(double v)
{
    v = dpp_row_shr_add<1>(v);
    v = dpp_row_shr_add<2>(v);
    v = dpp_row_shr_add<4>(v);
    v = dpp_row_shr_add<8>(v);
    const int lo = __double2loint(v), hi = __double2hiint(v);
    double s = __hiloint2double(__builtin_amdgcn_readlane(hi, 15), __builtin_amdgcn_readlane(lo, 15));
    s += __hiloint2double(__builtin_amdgcn_readlane(hi, 31), __builtin_amdgcn_readlane(lo, 31));
    s += __hiloint2double(__builtin_amdgcn_readlane(hi, 47), __builtin_amdgcn_readlane(lo, 47));
    s += __hiloint2double(__builtin_amdgcn_readlane(hi, 63), __builtin_amdgcn_readlane(lo, 63));
    return s;
}

// minimum over the 64 lanes (NaN-free inputs), same value in every lane; DPP like wave_sum
template <int SHR>
__device__ __forceinline__ float dpp_row_shr_min(float v)
{
    const int o = __builtin_amdgcn_update_dpp(0x7F800000 /* +inf where there is no source lane */, __float_as_int(v),
                                              0x110 + SHR, 0xf, 0xf, false);
    return fminf(v, __int_as_float(o));
}
__device__ __forceinline__ float wave_min(float v)
{
    v = dpp_row_shr_min<1>(v);
    v = dpp_row_shr_min<2>(v);
    v = dpp_row_shr_min<4>(v);
    v = dpp_row_shr_min<8>(v);
    const int b = __float_as_int(v);
    return fminf(fminf(__int_as_float(__builtin_amdgcn_readlane(b, 15)), __int_as_float(__builtin_amdgcn_readlane(b, 31))),
                 fminf(__int_as_float(__builtin_amdgcn_readlane(b, 47)), __int_as_float(__builtin_amdgcn_readlane(b, 63))));
}

// inclusive prefix sum over the 64 lanes (unsigned), DPP inside the rows + the row totals of the rows before
template <int SHR>
__device__ __forceinline__ unsigned dpp_row_shr_addu(unsigned v)
{
    return v + (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x110 + SHR, 0xf, 0xf, false);
}
__device__ __forceinline__ unsigned wave_inclusive_scan(unsigned v)
{
    v = dpp_row_shr_addu<1>(v);
    v = dpp_row_shr_addu<2>(v);
    v = dpp_row_shr_addu<4>(v);
    v = dpp_row_shr_addu<8>(v);
    const unsigned r0 = (unsigned)__builtin_amdgcn_readlane((int)v, 15), r1 = (unsigned)__builtin_amdgcn_readlane((int)v, 31),
                   r2 = (unsigned)__builtin_amdgcn_readlane((int)v, 47);
    const int row = (threadIdx.x & 63) >> 4;
    return v + (row > 0 ? r0 : 0u) + (row > 1 ? r1 : 0u) + (row > 2 ? r2 : 0u);
}

// block-wide sum of NV doubles per thread; result valid in every thread (NTH = threads of the workgroup)
template <int NV, int NTH = ICP_THREADS>
__device__ __forceinline__ void block_sum(double (&v)[NV], double *s_red /* (NTH/64)*NV + NV */)
{
    constexpr int NWV = NTH / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const double s = wave_sum(v[i]);
        if (lane == 0)
            s_red[wave * NV + i] = s;
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        double s = 0;
        for (int w = 0; w < NWV; ++w) // fixed order: deterministic
            s += s_red[w * NV + threadIdx.x];
        s_red[NWV * NV + threadIdx.x] = s;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i)
        v[i] = s_red[NWV * NV + i];
    __syncthreads();
}

__device__ __forceinline__ void mat3_mul(const float *a, const float *b, float *c)
{
    float r[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float s = f_mul(a[i * 3], b[j]);
            s = f_add(s, f_mul(a[i * 3 + 1], b[3 + j]));
            s = f_add(s, f_mul(a[i * 3 + 2], b[6 + j]));
            r[i * 3 + j] = s;
        }
#pragma unroll
    for (int i = 0; i < 9; ++i)
        c[i] = r[i];
}

__device__ __forceinline__ float dist2(float px, float py, float tx, float ty)
{
    // fl(fl(dx*dx) + fl(dy*dy)): how the oracle / libnabo accumulate the squared distance
    const float dx = f_add(px, -tx), dy = f_add(py, -ty);
    return f_add(f_mul(dx, dx), f_mul(dy, dy));
}


// Checker state of one job: the history lives in LDS and only thread 0 touches it.
struct IcpCheck {
    float *hist_c, *hist_s, *hist_x, *hist_y;
    int nhist, counter, iters;
};

// One lane: error minimiser solve (closed-form weighted Kabsch, or Cholesky of the 2-D
// point-to-plane normal equations) from the reduced sums `acc`, T_iter = T_step * T_iter, then
// the Counter / Differential transformation checkers (libpointmatcher order).  Ti = current
// T_iter (registers), Ti_lds = where the new one goes.
__device__ __forceinline__ void icp_solve_and_check(const sfe_icp_params &P, const double (&acc)[10],
                                                    const float (&Ti)[9], float *Ti_lds, IcpCheck &C, int &status,
                                                    int &iterate)
{
    float *hist_c = C.hist_c, *hist_s = C.hist_s, *hist_x = C.hist_x, *hist_y = C.hist_y;
    int &nhist = C.nhist, &counter = C.counter, &iters = C.iters;
    status = SFE_ICP_OK;
    iterate = 1;
    float Ts[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (acc[0] == 0.0) {
        status = SFE_ICP_NO_POINT;
    } else if (P.minimizer == 0) {
        const double W = acc[0];
        const double mpx = acc[1] / W, mpy = acc[2] / W, mqx = acc[3] / W, mqy = acc[4] / W;
        const double m00 = acc[5] - acc[3] * mpx, m01 = acc[6] - acc[3] * mpy;
        const double m10 = acc[7] - acc[4] * mpx, m11 = acc[8] - acc[4] * mpy;
        const double Sx = m00 + m11, Kx = m10 - m01;
        const double h = sqrt(Sx * Sx + Kx * Kx);
        const double c = (h == 0) ? 1.0 : Sx / h;
        const double s = (h == 0) ? 0.0 : Kx / h;
        const double tx = mqx - (c * mpx - s * mpy);
        const double ty = mqy - (s * mpx + c * mpy);
        Ts[0] = (float)c;
        Ts[1] = (float)-s;
        Ts[2] = (float)tx;
        Ts[3] = (float)s;
        Ts[4] = (float)c;
        Ts[5] = (float)ty;
    } else {
        // a pivot that has lost ten digits against its diagonal entry = rank-deficient system (in exact arithmetic the
        // pivot is zero, in floating point its sign is summation-order noise): decided by a relative test, the same
        // expression as the oracle's, so that every implementation reports such systems the same way
        const double l00 = sqrt(acc[1]);
        const double l10 = acc[2] / l00, l20 = acc[3] / l00;
        const double p11 = acc[4] - l10 * l10;
        const double l11 = sqrt(p11);
        const double l21 = (acc[5] - l20 * l10) / l11;
        const double p22 = acc[6] - l20 * l20 - l21 * l21;
        const double l22 = sqrt(p22);
        if (!(l00 > 0) || !(p11 > ICP_PIVOT_RTOL * acc[4]) || !(p22 > ICP_PIVOT_RTOL * acc[6])) {
            status = SFE_ICP_SINGULAR;
        } else {
            const double y0 = acc[7] / l00;
            const double y1 = (acc[8] - l10 * y0) / l11;
            const double y2 = (acc[9] - l20 * y0 - l21 * y1) / l22;
            const double x2 = y2 / l22;
            const double x1 = (y1 - l21 * x2) / l11;
            const double x0 = (y0 - l10 * x1 - l20 * x2) / l00;
            const double c = cos(x0), s = sin(x0);
            Ts[0] = (float)c;
            Ts[1] = (float)-s;
            Ts[2] = (float)x1;
            Ts[3] = (float)s;
            Ts[4] = (float)c;
            Ts[5] = (float)x2;
        }
    }
    if (status == SFE_ICP_OK) {
        float Tn[9];
        mat3_mul(Ts, Ti, Tn);
        for (int i = 0; i < 9; ++i)
            Ti_lds[i] = Tn[i];
        ++iters;
        ++counter;
        if (counter >= P.max_iter) {
            iterate = 0; // CounterTransformationChecker: MaxNumIterationsReached
        } else if (P.use_diff_checker) {
            if (nhist < ICP_MAX_HIST) {
                hist_c[nhist] = Tn[0];
                hist_s[nhist] = Tn[3];
                hist_x[nhist] = Tn[2];
                hist_y[nhist] = Tn[5];
                ++nhist;
            } else { // keep a sliding window (only the last smooth_len+1 entries are read)
                for (int i = 1; i < ICP_MAX_HIST; ++i) {
                    hist_c[i - 1] = hist_c[i];
                    hist_s[i - 1] = hist_s[i];
                    hist_x[i - 1] = hist_x[i];
                    hist_y[i - 1] = hist_y[i];
                }
                hist_c[ICP_MAX_HIST - 1] = Tn[0];
                hist_s[ICP_MAX_HIST - 1] = Tn[3];
                hist_x[ICP_MAX_HIST - 1] = Tn[2];
                hist_y[ICP_MAX_HIST - 1] = Tn[5];
            }
            // rotations.size() > smoothLength; size counts the init entry (= iters + 1)
            if (iters + 1 > P.smooth_len) {
                double rsum = 0, tsum = 0;
                for (int i = nhist - 1; i >= nhist - P.smooth_len; --i) {
                    const double c1 = hist_c[i], s1 = hist_s[i], c0 = hist_c[i - 1], s0 = hist_s[i - 1];
                    rsum += fabs(atan2(s1 * c0 - c1 * s0, c1 * c0 + s1 * s0));
                    const double dx = (double)hist_x[i] - hist_x[i - 1];
                    const double dy = (double)hist_y[i] - hist_y[i - 1];
                    tsum += sqrt(dx * dx + dy * dy);
                }
                rsum /= P.smooth_len;
                tsum /= P.smooth_len;
                if (rsum < P.min_diff_rot && tsum < P.min_diff_trans)
                    iterate = 0;
                if (isnan(rsum))
                    status = SFE_ICP_NAN_ROT;
                else if (isnan(tsum))
                    status = SFE_ICP_NAN_TRANS;
            }
        }
    }
}
