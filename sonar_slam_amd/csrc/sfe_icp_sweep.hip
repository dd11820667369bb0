// Strip-sweep ICP, host side: the jobs of a call sorted into classes (one-wave exhaustive kernel, one-wave / four-wave /
// 1024-thread sweep workgroups, targets in LDS or in HBM scratch, jobs shared by several workgroups), their tables and
// scratch, and the order of the launches (preparation -- optionally on the side stream -- normals, split, loop).  The
// kernels live in sfe_icp_sweep_prep.hip and sfe_icp_sweep_loop.hip; what they share, the exactness argument of the search
// and the overview are in sfe_icp_sweep.h.
#include "sfe_icp_sweep.h"

#include <vector>

// ---------------------------------------------------------------------------------------------
// host side: job tables, scratch, the prep / split / loop launches.  jobs4 = n_jobs x (src_start, n_src, tgt_start,
// n_tgt) in points.
// ---------------------------------------------------------------------------------------------
namespace {
int env_int(const char *name, int dflt)
{
    const char *v = getenv(name);
    return v ? atoi(v) : dflt;
}

} // namespace

int sfe_icp_sweep_launch(sfe_ctx *ctx, const sfe_icp_params *p, const float *d_src, const float *d_tgt,
                         const int32_t *jobs4, const float *d_guess9, int n_jobs, float *d_T9, int32_t *d_status,
                         int32_t *d_iters)
{
    std::vector<SweepPrep> preps;
    std::vector<SweepJob> jobs;
    jobs.reserve((size_t)n_jobs);
    // Job classes, one launch of the loop kernel each:
    //   tiny     clouds of a few hundred points: one wave per job, exhaustive search (sfe_icp_tiny.h)
    //   t0 / t1  small jobs: one-wave / four-wave workgroups, target AND per-query results in LDS, many jobs per CU
    //   q        1024-thread workgroups, target AND per-query results in LDS
    //   lds      ... target in LDS, results in HBM scratch
    //   glb      ... target in HBM scratch (beyond SW_TCAP points)
    //   multi    ... target in HBM scratch, the job shared by several workgroups (many-to-one batches on large clouds)
    std::vector<int> ids_tiny, ids_t0, ids_t1, ids_q, ids_lds, ids_glb, ids_multi, split_first;
    std::vector<int> pids[3]; // targets by prep tier
    std::vector<int> prep_tier; // ... of every target
    std::vector<char> prep_need; // a target only tiny point-to-point jobs use is not prepared (they take its mean themselves)
    std::vector<int> pids_nrm; // ... and those whose normals take a kernel of their own
    std::map<std::pair<int, int>, int> seen; // many guesses on one pair share one prep
    long long toff = 0, qoff = 0, koff = 0, goff = 0;
    // (knobs are read per call, not once per process: the tests switch them between calls)
    // tiers (A/B: SFE_SW_TIERS=0 sends every job to the 1024-thread kernels)
    const int tiers_on = env_int("SFE_SW_TIERS", 1);
    const int t0_src = env_int("SFE_SW_T0_SRC", 384), t1_src = env_int("SFE_SW_T1_SRC", 2048);
    // LDS share of a 1024-thread workgroup: two per CU when the batch has more jobs than CUs, else the whole CU
    const int force_wide = env_int("SFE_SW_WIDE", -1); // A/B: 1 = one 128-VGPR workgroup per CU
    const int force_share = env_int("SFE_SW_SHARE_KB", 0); // A/B: LDS per workgroup
    const bool no_ldsq = getenv("SFE_SW_NO_LDSQ") != nullptr; // A/B
    // A/B: 0 = never split a job; sfe_icp_set_tuning bit 4 = the same per context (a device shared with another
    // context / process cannot promise that all shares of a job are resident together: ADVICE r3)
    const int multi_on = (ctx->icp_variant & 16) ? 0 : env_int("SFE_SW_MULTI", 1);
    const int multi_min_src = env_int("SFE_SW_MULTI_MIN_SRC", 8192);
    const int multi_share_min = env_int("SFE_SW_MULTI_SHARE_MIN", 1024); // fewest queries worth a workgroup
    const int multi_force = env_int("SFE_SW_MULTI_G", 0);     // A/B: shares per job
    // ~96 points per strip on average (measured optimum 96-128 on 5000-point clouds: fewer queries need a
    // second strip, a little more to walk in each), one strip (= a plain x sweep) for small clouds
    const int strip_pts = std::max(1, env_int("SFE_SW_STRIP_PTS", 96));
    // clouds of a few hundred points: the exhaustive one-wave kernel (A/B: SFE_SW_TINY=0)
    const int tiny_on = env_int("SFE_SW_TINY", 1);
    const long long tiny_pairs = env_int("SFE_SW_TINY_PAIRS", (p->use_diff_checker || p->max_iter < SW_REC_MIN_ITER)
                                                                    ? SW_TINY_PAIRS_SHORT : SW_TINY_PAIRS_LONG);
    auto is_tiny = [&](int n_src, int n_tgt) {
        return tiny_on && n_src <= SW_TINY_MAX && n_tgt <= SW_TINY_MAX && (long long)n_src * n_tgt <= tiny_pairs;
    };

    // The small workgroups buy throughput (many jobs per CU), not latency: ONE scan match of 200 points is done sooner by
    // 256 threads (one slice of queries per pass) than by 64 (four slices, one after the other).  So the one-wave tier is
    // only used when the call brings enough such jobs to fill the device with them (the live node's single scan
    // match takes the four-wave kernel).
    const int t0_min_jobs = env_int("SFE_SW_T0_MIN_JOBS", 2 * ctx->n_cu), t1_min_jobs = env_int("SFE_SW_T1_MIN_JOBS", 1);
    auto fits0 = [&](int n_src, int n_tgt) { return tiers_on && n_src <= std::min(t0_src, 4096) && n_tgt <= SW_T0_TCAP; };
    auto fits1 = [&](int n_src, int n_tgt) { return tiers_on && n_src <= std::min(t1_src, 8192) && n_tgt <= SW_T1_TCAP; };
    int n_fit0 = 0, n_fit1 = 0;
    for (int j = 0; j < n_jobs; ++j) {
        const int32_t *q = jobs4 + 4 * (size_t)j;
        if (fits0(q[1], q[3]))
            ++n_fit0;
        else if (fits1(q[1], q[3]))
            ++n_fit1;
    }
    const bool use_t0 = n_fit0 >= t0_min_jobs;
    const bool use_t1 = n_fit1 + (use_t0 ? 0 : n_fit0) >= std::max(1, t1_min_jobs);
    // ... and with only a few small jobs in the call the four-wave kernel takes those of at most one slice of queries
    // (measured, one scan match alone: 200 points 261 us with 256 threads, 282 with 1024, 464 with 64; 1000 points 508 us
    // with 256 threads, 341 with 1024)
    const bool few_small = n_fit0 + n_fit1 < 2 * ctx->n_cu;
    const int t1_few_src = env_int("SFE_SW_T1_FEW_SRC", 320);
    // first pass: sizes -> how many big jobs there are (decides `wide` and whether big jobs are split)
    int n_big = 0, n_t2 = 0;
    auto tier_of = [&](int n_src, int n_tgt) {
        if (use_t0 && fits0(n_src, n_tgt))
            return 0;
        if (use_t1 && fits1(n_src, n_tgt) && (!few_small || n_src <= t1_few_src))
            return 1;
        return 2;
    };
    for (int j = 0; j < n_jobs; ++j) {
        const int32_t *q = jobs4 + 4 * (size_t)j;
        if (!is_tiny(q[1], q[3]) && tier_of(q[1], q[3]) == 2) {
            ++n_t2;
            if (q[3] > SW_TCAP && q[1] >= multi_min_src)
                ++n_big;
        }
    }
    const bool wide = force_wide >= 0 ? force_wide != 0 : n_t2 <= ctx->n_cu;
    const size_t lds_share = (force_share > 0 ? force_share : (wide ? 160 : 80)) * (size_t)1024;
    // Shares per big job: all workgroups of the launch must be resident at once (one 128-VGPR workgroup per CU) -- only
    // when the 1024-thread jobs of this call are the big ones alone and G x n_big fits the CUs
    int mg = 1;
    if (multi_on && n_big > 0 && n_big == n_t2 && wide) {
        mg = std::min(SW_MG_MAX, ctx->n_cu / n_big);
        if (multi_force > 0)
            mg = std::min(std::min(multi_force, SW_MG_MAX), std::max(1, ctx->n_cu / n_big));
    }
    constexpr size_t ctl_q = sweep_ctl_bytes<ICP_THREADS, false, true>(); // (the largest control block of the LDS_Q builds)
    int q_tmax = 0, q_smax = 0, t0_tmax = 0, t0_smax = 0, t1_tmax = 0, t1_smax = 0, tiny_tmax = 0, tiny_smax = 0;
    int n_sync = 0;
    for (int j = 0; j < n_jobs; ++j) {
        const int32_t *q = jobs4 + 4 * (size_t)j;
        const auto key = std::make_pair((int)q[2], (int)q[3]);
        auto it = seen.find(key);
        if (it == seen.end()) {
            it = seen.emplace(key, (int)preps.size()).first;
            long long n2 = 0;
            if (q[3] > SW_TCAP) {
                n2 = 2;
                while (n2 < q[3])
                    n2 <<= 1;
            }
            const int n_strips = std::max(1, std::min(SW_NS_MAX, (int)q[3] / strip_pts));
            // (the target's preparation follows the same rule: one wave only when the call fills the device with such jobs)
            const int pt = (use_t0 && q[3] <= SW_T0_TCAP) ? 0 : ((use_t1 && q[3] <= SW_T1_TCAP) ? 1 : 2);
            prep_tier.push_back(pt);
            prep_need.push_back(0);
            // (targets sorted in HBM scratch get their normals from icp_sweep_normals_kernel: pad_ = 1)
            const int nrm_later = (q[3] > SW_TCAP && p->minimizer == 1 && env_int("SFE_SW_NORMALS_SPLIT", 1)) ? 1 : 0;
            if (nrm_later)
                pids_nrm.push_back((int)preps.size());
            preps.push_back({q[2], q[3], n_strips, nrm_later, toff, koff, goff});
            toff += q[3] + SW_PAD;
            koff += n2;
            goff += pt == 0 ? SW_T0_GRID : (pt == 1 ? SW_T1_GRID : SW_GRID_MAX);
        }
        const SweepPrep &pr = preps[(size_t)it->second];
        const bool tiny = is_tiny(q[1], q[3]);
        const int tier = tiny ? -1 : tier_of(q[1], q[3]);
        if (!tiny || p->minimizer == 1)
            prep_need[(size_t)it->second] = 1;
        int shares = 1;
        if (tier == 2 && mg > 1 && q[3] > SW_TCAP && q[1] >= multi_min_src)
            shares = std::max(1, std::min(mg, (int)q[1] / std::max(1, multi_share_min)));
        const int rec0 = (int)jobs.size();
        for (int g = 0; g < shares; ++g) // (the shares of a split job: the split kernel fills in their clouds)
            jobs.push_back({q[0], q[1], q[3], it->second, pr.off, qoff, j, g, shares, shares > 1 ? n_sync : 0});
        qoff += q[1];
        if (shares > 1) {
            split_first.push_back(rec0);
            for (int g = 0; g < shares; ++g)
                ids_multi.push_back(rec0 + g);
            ++n_sync;
        } else if (tiny) {
            ids_tiny.push_back(rec0);
            tiny_tmax = std::max(tiny_tmax, (int)q[3]);
            tiny_smax = std::max(tiny_smax, (int)q[1]);
        } else if (tier == 0) {
            ids_t0.push_back(rec0);
            t0_tmax = std::max(t0_tmax, (int)q[3]);
            t0_smax = std::max(t0_smax, (int)q[1]);
        } else if (tier == 1) {
            ids_t1.push_back(rec0);
            t1_tmax = std::max(t1_tmax, (int)q[3]);
            t1_smax = std::max(t1_smax, (int)q[1]);
        } else {
            const bool fits_q = !no_ldsq && q[3] <= SW_TCAP &&
                                ctl_q + 8 * (size_t)(q[3] + SW_PAD) + 6 * (size_t)q[1] + 16 <= lds_share;
            if (fits_q) {
                ids_q.push_back(rec0);
                q_tmax = std::max(q_tmax, (int)q[3]);
                q_smax = std::max(q_smax, (int)q[1]);
            } else {
                (q[3] <= SW_TCAP ? ids_lds : ids_glb).push_back(rec0);
            }
        }
    }
    for (size_t pid = 0; pid < preps.size(); ++pid)
        if (prep_need[pid])
            pids[prep_tier[pid]].push_back((int)pid);
    // the LDS_Q launch is sized by the largest target and the largest source among its jobs
    int t_cap = q_tmax + SW_PAD, q_cap = (q_smax + 3) & ~3;
    if (!ids_q.empty() && ctl_q + 8 * (size_t)t_cap + 6 * (size_t)q_cap > lds_share) {
        ids_lds.insert(ids_lds.end(), ids_q.begin(), ids_q.end()); // odd mix of shapes: keep the results in HBM
        std::sort(ids_lds.begin(), ids_lds.end());
        ids_q.clear();
    }
    const int n_prep = (int)preps.size(), n_rec = (int)jobs.size();
    const int n_tiny = (int)ids_tiny.size();
    const int n_t0 = (int)ids_t0.size(), n_t1 = (int)ids_t1.size(), n_q = (int)ids_q.size(), n_lds = (int)ids_lds.size(),
              n_glb = (int)ids_glb.size(), n_multi = (int)ids_multi.size(), n_split = (int)split_first.size();
    // the tables travel as ONE block: [preps | job records | job ids by class | share-0 records of the split jobs |
    // prep ids by tier]
    const size_t o_jobs = (sizeof(SweepPrep) * (size_t)n_prep + 15) & ~(size_t)15;
    const size_t o_ids = (o_jobs + sizeof(SweepJob) * (size_t)n_rec + 15) & ~(size_t)15;
    const size_t o_split = o_ids + sizeof(int) * (size_t)n_rec;
    const size_t o_pids = o_split + sizeof(int) * (size_t)n_split;
    const size_t o_pnrm = o_pids + sizeof(int) * (size_t)n_prep;
    const size_t tab_bytes = o_pnrm + sizeof(int) * pids_nrm.size();
    char *d_tables = (char *)sfe_scratch(ctx, 12, tab_bytes);
    float2 *d_stgt = (float2 *)sfe_scratch(ctx, 14, sizeof(float2) * (size_t)toff);
    int *d_perm = (int *)sfe_scratch(ctx, 15, sizeof(int) * (size_t)toff);
    float2 *d_snrm = p->minimizer == 1 ? (float2 *)sfe_scratch(ctx, 16, sizeof(float2) * (size_t)toff) : nullptr;
    float *d_mean = (float *)sfe_scratch(ctx, 17, sizeof(float) * 2 * (size_t)n_prep);
    unsigned long long *d_gkeys = (unsigned long long *)sfe_scratch(ctx, 24, sizeof(unsigned long long) * (size_t)std::max(koff, 1LL));
    int4 *d_qst = (int4 *)sfe_scratch(ctx, 19, sizeof(int4) * (size_t)qoff);
    int *d_qwl = (int *)sfe_scratch(ctx, 21, sizeof(int) * 7 * (size_t)qoff);
    float2 *d_qssrc = (float2 *)sfe_scratch(ctx, 30, sizeof(float2) * (size_t)qoff);
    StripTab *d_tab = (StripTab *)sfe_scratch(ctx, 23, sizeof(StripTab) * (size_t)n_prep);
    int *d_grid = (int *)sfe_scratch(ctx, 38, sizeof(int) * (size_t)goff);
    float *d_nn_d2 = (float *)sfe_scratch(ctx, 5, sizeof(float) * (size_t)qoff);
    int *d_nn_pos = (int *)sfe_scratch(ctx, 6, sizeof(int) * (size_t)qoff);
    if (!d_tables || !d_stgt || !d_perm || (p->minimizer == 1 && !d_snrm) || !d_mean || !d_gkeys || !d_qst || !d_qwl || !d_qssrc || !d_tab || !d_grid ||
        !d_nn_d2 || !d_nn_pos)
        return SFE_ERR_HIP;
    // split jobs: the gathered source clouds of the shares and the sync areas
    float2 *d_gsrc = nullptr;
    unsigned long long *d_sync = nullptr;
    const size_t sync_bytes = sizeof(unsigned long long) * 2 * SW_MG_MAX * SW_MG_WORDS * (size_t)n_sync;
    if (n_split) {
        d_gsrc = (float2 *)sfe_scratch(ctx, 46, sizeof(float2) * (size_t)qoff);
        d_sync = (unsigned long long *)sfe_scratch(ctx, 45, sync_bytes);
        if (!d_gsrc || !d_sync)
            return SFE_ERR_HIP;
    }
    SweepPrep *d_preps = (SweepPrep *)d_tables;
    SweepJob *d_jobs = (SweepJob *)(d_tables + o_jobs);
    int *d_ids = (int *)(d_tables + o_ids);
    int *d_split = (int *)(d_tables + o_split);
    int *d_pids = (int *)(d_tables + o_pids);
    // Tuning bit 3: the caller vouches that the clouds and guesses of this batch are final (nothing enqueued on
    // ctx->stream still writes them).  The job tables and the prep kernel then go to the side stream and run next to
    // whatever precedes this call on ctx->stream (a batch pipeline enqueues the front end of the same step there:
    // latency-bound kernels that leave most of a CU idle, like the prep kernel); the loop kernel waits for them.
    const bool side = (ctx->icp_variant & 8) != 0;
    hipStream_t ps = side ? ctx->stream2 : ctx->stream;
    if (side && ctx->icp_loop_pending) // the previous batch's loop kernel still reads the scratch the prep rewrites
        SFE_HIP(ctx, hipStreamWaitEvent(ps, ctx->ev_loop, 0));
    { // pinned staging (no stream synchronisation: this entry point only enqueues)
        char *h = (char *)sfe_pinned_begin(ctx, tab_bytes);
        if (!h)
            return SFE_ERR_HIP;
        memcpy(h, preps.data(), sizeof(SweepPrep) * (size_t)n_prep);
        memcpy(h + o_jobs, jobs.data(), sizeof(SweepJob) * (size_t)n_rec);
        int *hi = (int *)(h + o_ids);
        for (const std::vector<int> *v : {&ids_tiny, &ids_t0, &ids_t1, &ids_q, &ids_lds, &ids_glb, &ids_multi}) {
            if (!v->empty())
                memcpy(hi, v->data(), sizeof(int) * v->size());
            hi += v->size();
        }
        if (n_split)
            memcpy(h + o_split, split_first.data(), sizeof(int) * (size_t)n_split);
        int *hp = (int *)(h + o_pids);
        for (int t = 0; t < 3; ++t) {
            if (!pids[t].empty())
                memcpy(hp, pids[t].data(), sizeof(int) * pids[t].size());
            hp += pids[t].size();
        }
        if (!pids_nrm.empty())
            memcpy(h + o_pnrm, pids_nrm.data(), sizeof(int) * pids_nrm.size());
        SFE_HIP(ctx, hipMemcpyAsync(d_tables, h, tab_bytes, hipMemcpyHostToDevice, ps));
        if (int rc = sfe_pinned_end(ctx, ps))
            return rc;
    }
    {
        const int np0 = (int)pids[0].size(), np1 = (int)pids[1].size(), np2 = (int)pids[2].size();
        if (np0)
            if (int rc = sweep_launch_prep<SW_T0_NT, SW_T0_TCAP, SW_T0_GRID>(ctx, ps, p, np0, d_preps, d_pids, (const float2 *)d_tgt, d_stgt,
                                                                          d_perm, d_snrm, d_mean, d_gkeys, d_tab, d_grid))
                return rc;
        if (np1)
            if (int rc = sweep_launch_prep<SW_T1_NT, SW_T1_TCAP, SW_T1_GRID>(ctx, ps, p, np1, d_preps, d_pids + np0, (const float2 *)d_tgt,
                                                                          d_stgt, d_perm, d_snrm, d_mean, d_gkeys, d_tab, d_grid))
                return rc;
        if (np2) {
            const bool gtail = env_int("SFE_SW_PREP_GTAIL", 1) != 0; // (0: the witness grid in LDS of its own, one workgroup per CU: A/B)
            // (the list capacity the kernel is built for: the smallest of 8 / 10 / 12 / 16 that holds normals_knn)
            const int km = p->minimizer != 1 ? 8 : p->normals_knn <= 8 ? 8 : p->normals_knn <= 10 ? 10 : p->normals_knn <= 12 ? 12 : ICP_KMAX;
            auto go = [&](auto gt, auto kmf) {
                return sweep_launch_prep<ICP_THREADS, SW_TCAP, SW_GRID_MAX, decltype(gt)::value, decltype(kmf)::value>(
                    ctx, ps, p, np2, d_preps, d_pids + np0 + np1, (const float2 *)d_tgt, d_stgt, d_perm, d_snrm, d_mean, d_gkeys, d_tab,
                    d_grid);
            };
            using std::integral_constant;
            // (lists of 12 / 16 neighbours do not fit 64 registers: 184 / 541 spills; those keep one workgroup per CU)
            if (int rc = !gtail || km > 10 ? go(std::false_type{}, integral_constant<int, 0>{})
                         : km == 8          ? go(std::true_type{}, integral_constant<int, 8>{})
                                            : go(std::true_type{}, integral_constant<int, 10>{}))
                return rc;
        }
    }
    if (!pids_nrm.empty()) { // behind the prep (sorted cloud, strip table), 1024 points per workgroup and pass
        int tmax = 0;
        for (int pid : pids_nrm)
            tmax = std::max(tmax, preps[(size_t)pid].n_tgt);
        const int per = std::max(1, std::min(32, std::min((tmax + 2 * ICP_THREADS - 1) / (2 * ICP_THREADS),
                                                          std::max(1, 2 * ctx->n_cu / (int)pids_nrm.size()))));
        if (int rc = sweep_launch_normals(ctx, ps, p, (int)pids_nrm.size(), per, d_preps, (const int *)(d_tables + o_pnrm),
                                          (const float2 *)d_stgt, (const int *)d_perm, d_snrm, (const StripTab *)d_tab))
            return rc;
    }
    if (n_split) { // behind the prep (it needs the strip tables), in front of the loop
        SFE_HIP(ctx, hipMemsetAsync(d_sync, 0, sync_bytes, ps));
        if (int rc = sweep_launch_split(ctx, ps, n_split, d_jobs, d_split, (const float2 *)d_src, d_guess9, d_mean, d_tab, d_gsrc))
            return rc;
    }
    if (side) {
        SFE_HIP(ctx, hipEventRecord(ctx->ev_prep, ps));
        SFE_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_prep, 0));
    }
    static const bool debug = getenv("SFE_ICP_DEBUG") != nullptr;
    // bit 0: witness / clearance cache (0: A/B without it); bits 8..15: walk trips per second-pass round
    // bit 1: the cap of a repeated round comes from the queries' upper bounds (0: grows 4x)
    // bit 2: grid witnesses in the first iteration (0: A/B without them)
    const int sw_cache = (env_int("SFE_SW_CACHE", 1) & 1) | (env_int("SFE_SW_GRID", 1) ? 4 : 0) |
                         (env_int("SFE_SW_GRID_SKIP", 1) ? 8 : 0) | // bit 3: witnessed queries skip round 0
                         (env_int("SFE_SW_REC", 1) ? 16 : 0) |      // bit 4: clearance records
                         (env_int("SFE_SW_TRIAGE", 1) ? 32 : 0) |   // bit 5: ... with triage passes
                         (env_int("SFE_SW_GRID_DEFER", 1) ? 64 : 0) | // bit 6: first iteration: queries without a witness wait for round 1
                         (env_int("SFE_SW_UNBOUNDED_COOP", 1) ? 128 : 0) | // bit 7: later passes: unbounded queries -> cooperative tier
                         (env_int("SFE_SW_LEAN_TRIAGE", 1) ? (1 << 25) : 0) |  // bit 25: the triage pass as a loop of its own
                         (env_int("SFE_SW_PREDICT_Q", 1) ? (1 << 26) : 0) |    // bit 26: steady state: the trimmed quantile looked for around the previous one
                         (env_int("SFE_SW_JUMP", 1) ? 2 : 0) |
                         // bits 16..23: margin (percent) of the next iteration's cap over this iteration's limit
                         (std::max(0, std::min(255, env_int("SFE_SW_MARGIN", SW_CAP_MARGIN))) << 16) |
                         (std::max(1, std::min(255, env_int("SFE_SW_RTRIPS", SW_ROUND_TRIPS))) << 8);
    int *d_dbg = nullptr;
    if (debug) {
        d_dbg = (int *)sfe_scratch(ctx, 22, sizeof(int) * 8);
        if (!d_dbg)
            return SFE_ERR_HIP;
        SFE_HIP(ctx, hipMemsetAsync(d_dbg, 0, sizeof(int) * 8, ctx->stream));
    }
    long long *d_prof = ctx->icp_prof ? (long long *)sfe_scratch(ctx, 20, sizeof(long long) * SFE_ICP_PROF_N) : nullptr;
    if (ctx->icp_prof && !d_prof)
        return SFE_ERR_HIP;
    if (d_prof)
        SFE_HIP(ctx, hipMemsetAsync(d_prof, 0, sizeof(long long) * SFE_ICP_PROF_N, ctx->stream));
    SweepLaunchArgs a;
    a.ctx = ctx;
    a.p = p;
    a.d_jobs = d_jobs;
    a.d_src = (const float2 *)d_src;
    a.d_guess9 = d_guess9;
    a.d_stgt = d_stgt;
    a.d_perm = d_perm;
    a.d_snrm = d_snrm;
    a.d_mean = d_mean;
    a.d_tab = d_tab;
    a.d_grid = d_grid;
    a.d_qst = d_qst;
    a.d_qwl = d_qwl;
    a.d_qssrc = d_qssrc;
    a.d_nn_d2 = d_nn_d2;
    a.d_nn_pos = d_nn_pos;
    a.d_T9 = d_T9;
    a.d_status = d_status;
    a.d_iters = d_iters;
    a.d_prof = d_prof;
    a.d_dbg = d_dbg;
    a.sw_budget = env_int("SFE_SW_BUDGET", SW_BUDGET);
    a.sw_budget_a = env_int("SFE_SW_BUDGET_A", SW_BUDGET_A);
    a.sw_cache = sw_cache;
    // union scan of the first iterations: bits 0..7 = iterations that use it, bits 8..23 = most points of a union window.
    // Measured (1024 jobs of 5000 x 5000, search cycles of workgroup 0 per iteration, walks -> union): iteration 0
    // 997 k -> 828 k, iteration 1 486 k -> 659 k, iteration 2 324 k -> 585 k: only the first iteration's windows (bounds
    // from grid witnesses, ~0.9 m) are wide enough for the union of 64 neighbours' windows (~400 points) to beat the
    // private walks; from the second iteration on a query's old neighbour bounds it to a few dozen candidates.  Whole
    // launch: p2plane30 8.78-8.87 -> 8.64-8.76 ms, shipped chain 4.07 -> 3.93-4.00 ms per 1024 jobs.
    a.sw_cache2 = std::max(0, std::min(255, env_int("SFE_SW_UNION_ITERS", 1))) |
                  (std::max(0, std::min(65535, env_int("SFE_SW_UNION_MAX", 768))) << 8);
    // margin of the clearance records: a search looks this fraction further (in radius) than it has to ...
    a.sw_m = 0.01f * (float)std::max(1, std::min(100, env_int("SFE_SW_RECM", SW_REC_MARGIN)));
    // ... plus sw_kappa x the largest movement of the last step (the steps shrink geometrically once ICP converges: a
    // few times the last one covers all that are still to come)
    a.sw_kappa = getenv("SFE_SW_RECK") ? (float)atof(getenv("SFE_SW_RECK")) : SW_REC_KAPPA;
    a.d_sync = d_sync;
    // The build with clearance records carries more per-lane state (the 64-VGPR budget makes every register count:
    // the same chain runs ~10 % slower in it until the records start to hit), so it only takes chains that are set to
    // run many iterations: a fixed count (no differential checker) of at least SW_REC_MIN_ITER -- and only the builds
    // with LDS-resident targets: for the one-workgroup-per-CU builds with the target in HBM (30 guesses x one
    // 20 000 x 20 000 pair) the records cost more than they save (24.6 -> 30.9 ms, tools/hires_times.py).
    const bool rec_build = (sw_cache & 16) != 0 && p->max_iter >= SW_REC_MIN_ITER && !p->use_diff_checker;
    // A/B: VGPR budget of the LDS_Q build.  A workgroup is 1024 threads = 4 waves per SIMD, so two workgroups per CU
    // only fit at <= 64 VGPRs: measured (4096 jobs of 5000 x 5000, p2plane30) 64 VGPRs 35.4 ms, 72 VGPRs 51.4 ms,
    // 128 VGPRs 48.6 ms (profiles/r02_icp_vgpr_budget.txt) -- one resident job per CU costs more than the 86 spilled
    // VGPRs of the 64-VGPR build
    int rc = 0;
    const int *ids = d_ids;
    if (n_tiny) { // one wave per job, everything in LDS (sfe_icp_tiny.h)
        if ((rc = sweep_launch_tiny(ctx, p, n_tiny, d_jobs, ids, d_preps, (const float2 *)d_src, (const float2 *)d_tgt, d_guess9,
                                    (const int *)d_perm, (const float2 *)d_snrm, (const float *)d_mean, (const StripTab *)d_tab, d_T9,
                                    d_status, d_iters, tiny_tmax, tiny_smax)))
            return rc;
        ids += n_tiny;
    }
    if (n_t0) { // one wave per job (no workgroup barrier costs anything), 128 VGPRs, up to 16 jobs per CU
        const int tc = t0_tmax + SW_PAD, qc = (t0_smax + 3) & ~3;
        const size_t body = 8 * (size_t)tc + 6 * (size_t)qc;
        rc = rec_build ? sweep_launch_loop<SW_T0_NT, 4, true, true, false, true, false>(a, n_t0, ids, body, tc, qc)
                       : sweep_launch_loop<SW_T0_NT, 4, true, true, false, false, false>(a, n_t0, ids, body, tc, qc);
        if (rc)
            return rc;
        ids += n_t0;
    }
    if (n_t1) { // four waves per job, 128 VGPRs, up to 4 jobs per CU
        const int tc = t1_tmax + SW_PAD, qc = (t1_smax + 3) & ~3;
        const size_t body = 8 * (size_t)tc + 6 * (size_t)qc;
        const int t1_minw = env_int("SFE_SW_T1_MINW", 4); // A/B: 8 = 64 VGPRs, up to 8 jobs per CU
        if (d_prof && !rec_build)
            rc = sweep_launch_loop<SW_T1_NT, 4, true, true, true, false, false>(a, n_t1, ids, body, tc, qc);
        else if (t1_minw == 8 && !rec_build)
            rc = sweep_launch_loop<SW_T1_NT, 8, true, true, false, false, false>(a, n_t1, ids, body, tc, qc);
        else
        rc = rec_build ? sweep_launch_loop<SW_T1_NT, 4, true, true, false, true, false>(a, n_t1, ids, body, tc, qc)
                       : sweep_launch_loop<SW_T1_NT, 4, true, true, false, false, false>(a, n_t1, ids, body, tc, qc);
        if (rc)
            return rc;
        ids += n_t1;
    }
    // 1024-thread jobs: up to one job per CU the 128-VGPR build wins (no spills, measured +8 %), beyond that two
    // 64-VGPR workgroups per CU overlap each other's serial phases (measured +14 % at 512 jobs)
    if (n_q) {
        const size_t body = 8 * (size_t)t_cap + 6 * (size_t)q_cap;
        if (wide)
            rc = sweep_launch_loop<ICP_THREADS, 4, true, true, false, false, false>(a, n_q, ids, body, t_cap, q_cap);
        else if (d_prof && rec_build)
            rc = sweep_launch_loop<ICP_THREADS, 8, true, true, true, true, false>(a, n_q, ids, body, t_cap, q_cap);
        else if (d_prof)
            rc = sweep_launch_loop<ICP_THREADS, 8, true, true, true, false, false>(a, n_q, ids, body, t_cap, q_cap);
        else if (rec_build)
            rc = sweep_launch_loop<ICP_THREADS, 8, true, true, false, true, false>(a, n_q, ids, body, t_cap, q_cap);
        else
            rc = sweep_launch_loop<ICP_THREADS, 8, true, true, false, false, false>(a, n_q, ids, body, t_cap, q_cap);
        if (rc)
            return rc;
        ids += n_q;
    }
    if (n_lds) {
        const size_t body = sizeof(float2) * (SW_TCAP + SW_PAD);
        if (wide)
            rc = sweep_launch_loop<ICP_THREADS, 4, true, false, false, false, false>(a, n_lds, ids, body, SW_TCAP + SW_PAD, 0);
        else if (d_prof)
            rc = sweep_launch_loop<ICP_THREADS, 8, true, false, true, false, false>(a, n_lds, ids, body, SW_TCAP + SW_PAD, 0);
        else if (rec_build)
            rc = sweep_launch_loop<ICP_THREADS, 8, true, false, false, true, false>(a, n_lds, ids, body, SW_TCAP + SW_PAD, 0);
        else
            rc = sweep_launch_loop<ICP_THREADS, 8, true, false, false, false, false>(a, n_lds, ids, body, SW_TCAP + SW_PAD, 0);
        if (rc)
            return rc;
        ids += n_lds;
    }
    // Targets beyond SW_TCAP points stay in HBM scratch and are read through L2.  (Rounds 3-4 carried a build that held the
    // strips around a workgroup's queries in LDS instead -- 7.97 ms with it, 7.82 without on 30 guesses x one 20 000 x 20 000
    // pair: the target is L2-resident and sixteen waves hide its latency, while every read paid the window test; removed in
    // round 5, profiles/r05_pruned_variants.txt.)
    if (n_glb) {
        const size_t body = sizeof(unsigned long long) * SW_TCAP; // without a window the LDS behind the control block only serves the query sort
        rc = wide ? sweep_launch_loop<ICP_THREADS, 4, false, false, false, false, false>(a, n_glb, ids, body, SW_TCAP, 0)
                      : sweep_launch_loop<ICP_THREADS, 8, false, false, false, false, false>(a, n_glb, ids, body, SW_TCAP, 0);
        if (rc)
            return rc;
        ids += n_glb;
    }
    if (n_multi) { // every share on a CU of its own, all of them resident (mg x n_big <= CUs); their clouds were gathered
        const size_t body = sizeof(unsigned long long) * SW_TCAP;
        SweepLaunchArgs am = a;
        am.d_src = d_gsrc;
        rc = sweep_launch_loop<ICP_THREADS, 4, false, false, false, false, true>(am, n_multi, ids, body, SW_TCAP, 0);
        if (rc)
            return rc;
        ids += n_multi;
    }
    if (side) {
        SFE_HIP(ctx, hipEventRecord(ctx->ev_loop, ctx->stream));
        ctx->icp_loop_pending = true;
    }
    if (debug) {
        int h[8];
        SFE_HIP(ctx, hipMemcpyAsync(h, d_dbg, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
        SFE_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (int i = 0; i < 8; ++i)
            if (h[i])
                fprintf(stderr, "sfe_icp_sweep: watchdog %d tripped (workgroup %d)\n", i, h[i] - 1);
    }
    if (d_prof) {
        SFE_HIP(ctx, hipMemcpyAsync(ctx->icp_prof_host, d_prof, sizeof(long long) * SFE_ICP_PROF_N, hipMemcpyDeviceToHost,
                                    ctx->stream));
        SFE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return 0;
}

extern "C" int sfe_icp_get_profile(sfe_ctx *ctx, int enable, long long *cycles16)
{
    if (!ctx)
        return SFE_ERR_ARG;
    if (cycles16)
        for (int i = 0; i < SFE_ICP_PROF_N; ++i)
            cycles16[i] = ctx->icp_prof_host[i];
    ctx->icp_prof = enable;
    return 0;
}
