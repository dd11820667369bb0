// gtsam.Pose2's arithmetic as sonar_slam_amd/pose2.py states it (rotation kept as (cos, sin), a product renormalised only when
// c^2 + s^2 is off by more than 1e-10: Rot2::normalize), in double, the operations in the order of the Python expressions -- for the
// host routine sfe_pose2_sample_transforms and for the kernel that computes the sample transforms of the matching cost on the
// device (sfe_matching_cost_store_samples).  IEEE double add / multiply / divide / sqrt on both sides (-ffp-contract=off), so the
// float32 rows are the same bits.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>

struct SfeP2 {
    double x, y, c, s;
};

__host__ __device__ inline SfeP2 sfe_p2_make(double x, double y, double c, double s)
{
    const double scale = c * c + s * s;
    if (fabs(scale - 1.0) > 1e-10) {
        const double k = 1.0 / sqrt(scale);
        c = c * k;
        s = s * k;
    }
    return SfeP2{x, y, c, s};
}

__host__ __device__ inline SfeP2 sfe_p2_compose(const SfeP2 &a, const SfeP2 &o)
{
    return sfe_p2_make(a.x + a.c * o.x - a.s * o.y, a.y + a.s * o.x + a.c * o.y, a.c * o.c - a.s * o.s, a.s * o.c + a.c * o.s);
}

__host__ __device__ inline SfeP2 sfe_p2_inverse(const SfeP2 &a)
{
    return sfe_p2_make(-(a.c * a.x + a.s * a.y), -(-a.s * a.x + a.c * a.y), a.c, -a.s);
}

// float32 rows T00 T01 T02 T10 T11 T12 of target.between(source.compose(delta)).matrix()
__host__ __device__ inline void sfe_p2_sample_transform(const SfeP2 &target_inverse, const SfeP2 &source, const SfeP2 &delta, float *out6)
{
    const SfeP2 t = sfe_p2_compose(target_inverse, sfe_p2_compose(source, delta));
    out6[0] = (float)t.c;
    out6[1] = (float)-t.s;
    out6[2] = (float)t.x;
    out6[3] = (float)t.s;
    out6[4] = (float)t.c;
    out6[5] = (float)t.y;
}
