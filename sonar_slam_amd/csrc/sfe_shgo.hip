// Host-only: what scipy.optimize.shgo(sampling_method="sobol", iters=1) decides AFTER its sampling stage when the cost is piecewise
// constant, for many problems at once (slam.py:692-701 calls it in front of every sequential scan match; the cost of
// slam.py:529-567 is an integer count of grid cells).  The graph, the vertex order and the points are scipy's own, taken from one run
// of the installed scipy by sonar_slam_amd/shgo_fast.py (SobolPlan), which also states -- and tests against scipy -- the rules restated
// here: a vertex strictly below all its neighbours is a minimiser (scipy/optimize/_shgo_lib/_vertex.py `minimiser`); the pool is
// minimised first from the first minimiser in vertex order, then always from the one farthest from the last local result
// (`SHGO.minimise_pool` / `g_topograph`); SLSQP from a point whose three forward-difference neighbours cost the same returns the
// point itself; the result is the lowest local result (`LMapCache.sort_cache_result`).  What cannot be decided the way scipy would
// is reported, not guessed: see the status values in sonarfe.h.
#include "../../include/sonarfe.h"
#include "sfe_pose2.h"

#include <cmath>
#include <cstdint>

int sfe_shgo_sobol_replay(int n_vertices, const int32_t *nn_off, const int32_t *nn_idx, const double *x, const int32_t *tables,
                          int n_problems, uint8_t *status_out, int32_t *vertex_out, int32_t *n_order_out, int32_t *order_out)
{
    if (n_vertices <= 0 || n_problems < 0 || !nn_off || !nn_idx || !x || (n_problems && (!tables || !status_out || !vertex_out)) ||
        (n_problems && (!n_order_out || !order_out)))
        return SFE_ERR_ARG;
    const int V = n_vertices;
    for (int s = 0; s < n_problems; ++s) {
        const int32_t *t = tables + (size_t)s * V * 4;
        int32_t *order = order_out + (size_t)s * SFE_SHGO_MAX_POOL;
        int pool[SFE_SHGO_MAX_POOL];
        int n_pool = 0;
        bool overflow = false, moved = false;
        for (int v = 0; v < V; ++v) {
            const int32_t f = t[4 * v];
            bool is_min = true;
            for (int k = nn_off[v]; k < nn_off[v + 1] && is_min; ++k)
                is_min = f < t[4 * nn_idx[k]];
            if (!is_min)
                continue;
            if (n_pool == SFE_SHGO_MAX_POOL) {
                overflow = true;
                break;
            }
            pool[n_pool++] = v;
            moved |= t[4 * v + 1] != f || t[4 * v + 2] != f || t[4 * v + 3] != f;
        }
        n_order_out[s] = 0;
        if (overflow || moved) {
            status_out[s] = SFE_SHGO_FALLBACK;
            vertex_out[s] = -1;
            continue;
        }
        if (n_pool == 0) { // shgo: "Failed to find a feasible minimizer point": the first lowest vertex (find_lowest_vertex: strict <)
            int best = 0;
            for (int v = 1; v < V; ++v)
                if (t[4 * v] < t[4 * best])
                    best = v;
            status_out[s] = SFE_SHGO_FAILED;
            vertex_out[s] = best;
            continue;
        }
        int n_order = 0, n_rest = n_pool - 1;
        order[n_order++] = pool[0];
        int *rest = pool + 1;
        bool tie = false;
        while (n_rest > 0 && !tie) {
            const double *a = x + 3 * (size_t)order[n_order - 1];
            int far = -1;
            double d_far = -1.0, d_second = -1.0;
            for (int i = 0; i < n_rest; ++i) { // scipy.spatial.distance.cdist "euclidean": sqrt of the running sum of squares
                const double *b = x + 3 * (size_t)rest[i];
                double acc = 0.0;
                for (int c = 0; c < 3; ++c) {
                    const double d = a[c] - b[c];
                    acc += d * d;
                }
                const double d = std::sqrt(acc);
                if (d > d_far) {
                    d_second = d_far;
                    d_far = d;
                    far = i;
                } else if (d > d_second)
                    d_second = d;
            }
            // equal -- or within what a fused multiply-add inside scipy's build of cdist could move them -- : not decided here
            const int n_far = (n_rest > 1 && d_far - d_second <= 1e-12 * d_far) ? 2 : 1;
            if (n_far > 1) {
                tie = true; // np.argsort's last among equal distances is the sort kernel's choice: the caller asks numpy (shgo_fast.farthest)
                break;
            }
            order[n_order++] = rest[far];
            for (int i = far; i + 1 < n_rest; ++i)
                rest[i] = rest[i + 1];
            --n_rest;
        }
        if (tie) {
            status_out[s] = SFE_SHGO_FALLBACK;
            vertex_out[s] = -1;
            continue;
        }
        int low = 0, n_low = 1;
        for (int i = 1; i < n_order; ++i) {
            const int32_t f = t[4 * order[i]], fl = t[4 * order[low]];
            if (f < fl) {
                low = i;
                n_low = 1;
            } else if (f == fl)
                ++n_low;
        }
        n_order_out[s] = n_order;
        vertex_out[s] = order[low];
        status_out[s] = n_low > 1 ? SFE_SHGO_OK_TIED : SFE_SHGO_OK; // tied: np.argsort over the costs in `order` picks the result
    }
    return 0;
}

// gtsam.Pose2 arithmetic of the cost function's sample poses (slam.py:548-550): T6 of target_pose.between(source_pose.compose(delta))
// for n_sessions (target, source) pairs x n_deltas deltas, every pose as {x, y, cos, sin} in double -- the expressions of
// sonar_slam_amd/pose2.py (Pose2.compose / inverse / between, rotation renormalised when c^2 + s^2 is off by more than 1e-10), in the
// same order, so the float32 entries are the ones the one-by-one Python path hands to the cost kernel.  Host only.
int sfe_pose2_sample_transforms(const double *target_xycs, const double *source_xycs, int n_sessions, const double *delta_xycs,
                                int n_deltas, float *T6_out)
{
    if (n_sessions < 0 || n_deltas < 0 || (n_sessions && n_deltas && (!target_xycs || !source_xycs || !delta_xycs || !T6_out)))
        return SFE_ERR_ARG;
    for (int i = 0; i < n_sessions; ++i) {
        const SfeP2 tgt{target_xycs[4 * i], target_xycs[4 * i + 1], target_xycs[4 * i + 2], target_xycs[4 * i + 3]};
        const SfeP2 src{source_xycs[4 * i], source_xycs[4 * i + 1], source_xycs[4 * i + 2], source_xycs[4 * i + 3]};
        const SfeP2 inv = sfe_p2_inverse(tgt);
        float *out = T6_out + (size_t)i * n_deltas * 6;
        for (int j = 0; j < n_deltas; ++j) {
            const SfeP2 d{delta_xycs[4 * j], delta_xycs[4 * j + 1], delta_xycs[4 * j + 2], delta_xycs[4 * j + 3]};
            sfe_p2_sample_transform(inv, src, d, out + 6 * j);
        }
    }
    return 0;
}
