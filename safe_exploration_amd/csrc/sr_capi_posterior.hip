// sr_capi_posterior.hip -- posterior entry points: per-chunk workspace, the dispatch of one posterior pass over the
// kernels of sr_small / sr_stream / sr_predict (gp_pass), sr_gp_predict, sr_gp_linearize, the GP input transform, the
// completion mailbox and the one-command single query (sr_gp_call1).
#include "sr_handle.h"
using namespace srh;

// ---------------------------------------------------------------------------------------------
// workspace + the three-kernel GP pass over one chunk
// ---------------------------------------------------------------------------------------------
int srh::pick_nsplit(const sr_gp* h, long Tp) {
    const long blocks = ((Tp + 255) / 256) * h->n_out;
    long ns = (768 + blocks - 1) / blocks;
    // down to 16 training rows per workgroup (a 16-long exp chain); beyond SR_FINAL_WAVE_T queries
    // sr_finalize_kernel adds the partial sums serially per thread: never more than max(16, Np/128) there
    // (32 / 64 / 128 rows per workgroup for 128 queries at N = 5000: K* pass 10.9 -> 13.1 / 17.7 / 29.8 us, final stage 10.2 ->
    //  7.4 / 5.9 / 5.5 us: no gain in the sum)
    const long maxs = Tp <= SR_FINAL_WAVE_T ? (long)h->Np / 16
                                            : std::min((long)h->Np / 16, std::max(16L, (long)h->Np / SR_NB));
    if (ns > maxs) ns = maxs;
    if (ns < 1) ns = 1;
    return (int)ns;
}

int srh::ensure_ws(sr_gp* h, long Tp, int nsplit) {
    // two capacities: buffers sized by the padded batch (K*, row-block partials, outputs) and the N-split partials
    // of mu / jac, sized by the PRODUCT nsplit * Tp actually requested (nsplit is largest for tiny batches, Tp for
    // big ones: sizing by max(nsplit) x max(Tp) would hold GBs of dead workspace)
    const long need_part = (long)nsplit * Tp;
    if (Tp <= h->ws_Tp && need_part <= h->ws_part) return SR_OK;
    // a caller that has already handed h->mu / h->var / h->jac on (prepare_ws) must have sized for every route
    SR_CHECK(!h->ws_locked, SR_ESTATE, "internal: workspace of %ld x %ld needs %ld x %d while its buffers are in use",
             h->ws_Tp, h->ws_part, Tp, nsplit);
    const long nTp = std::max(Tp, h->ws_Tp);
    const long npart = std::max(need_part, h->ws_part);
    (void)device_sync();
    free_ws(h);
    const int nrb = h->Np / SR_NB;
    int rc;
    if ((rc = dev_alloc(&h->Ks, (size_t)h->n_out * h->Np * nTp)) ||
        (rc = dev_alloc(&h->mu_part, (size_t)npart * h->n_out)) ||
        (rc = dev_alloc(&h->jac_part, (size_t)npart * h->n_out * h->D)) ||
        (rc = dev_alloc(&h->var_part, (size_t)h->n_out * nrb * nTp)) ||
        (rc = dev_alloc(&h->mu, (size_t)h->n_out * nTp)) ||
        (rc = dev_alloc(&h->var, (size_t)h->n_out * nTp)) ||
        (rc = dev_alloc(&h->jac, (size_t)h->n_out * h->D * nTp)) ||
        (rc = dev_alloc(&h->kxx, (size_t)h->n_out * nTp))) {
        free_ws(h);
        return rc;
    }
    h->ws_Tp = nTp;
    h->ws_part = npart;
    return SR_OK;
}

// Size the workspace for WHATEVER route gp_pass takes with Tc queries, before an entry point resolves h->mu / h->var /
// h->jac: the fused small-batch route wants 2 * ceil(Np / 256) mean partials per query, which exceeds pick_nsplit for
// big models (n_out = 2: Np > 49152) -- sized by pick_nsplit alone, the first T <= 64 reachability call on such a
// model reallocated the workspace under the pointers its caller held.  The lock turns any such growth into an error.
// (65 .. 128 queries also fit the streamed route's 128 columns; measured at N = 5000, T = 128: 188 us against 165 us on
//  the split-K tiles -- the 16-wavefront MFMA kernel reaches 64 % of the matrix pipe there)
static int stream_max_t() { return SR_STREAM_MAX_T; }
int srh::prepare_ws(sr_gp* h, long Tc) {
    const long Tp = round_up(Tc, srt::BN);
    int ns = pick_nsplit(h, Tp);
    if (Tc <= stream_max_t()) ns = std::max(ns, std::max(pick_nsplit(h, srt::BN), 2 * ((h->Np + 255) / 256)));
    return ensure_ws(h, Tp, ns);
}

// ---- fused small-batch route (sr_stream.hip) ----------------------------------------------------------------
// Models beyond the one-launch sizes, up to 128 columns (queries, or [k*, dk*/dx] of one query): U^-1 is streamed
// once, reduction and final stage hang behind tickets.  ARD-RBF with D <= 5 and <= 4 columns: ONE launch (the
// workgroups evaluate their chunk of the columns themselves); otherwise the column pass (K1 / sr_lin_columns) first.
static int stream_buffers(sr_gp* h, int ncols, hipStream_t s) {
    const long need = sr_stream_vp_doubles(h->Np, h->n_out, sr_stream_width(ncols));
    if (need > h->stream_vp_cap) {
        (void)hipStreamSynchronize(s);
        dev_free(h->stream_vp);
        h->stream_vp = nullptr; h->stream_vp_cap = 0;
        SR_TRY(dev_alloc(&h->stream_vp, (size_t)need));
        h->stream_vp_cap = need;
    }
    if (!h->stream_tickets) {
        const int n = sr_stream_tickets(h->Np, h->n_out);
        SR_TRY(dev_alloc(&h->stream_tickets, (size_t)n));
        SR_TRY(dev_zero(h->stream_tickets, sizeof(unsigned) * n));
    }
    return SR_OK;
}

// work items of the run kernel for this model and column count: planned on the host once per (padded size, width), the table
// kept on the device (sr_stream_items)
static int stream_items(sr_gp* h, sr_stream_args& a, int ncols, int width_min, bool can_fuse, hipStream_t s) {
    const int nc = std::max(sr_stream_width(ncols), width_min);
    if (nc <= 4) return SR_OK;
    const long key = (((long)h->Np * 256 + nc) * 64 + h->n_out) * 2 + (can_fuse ? 1 : 0);
    if (h->stream_tab_key != key) {
        if (h->ncu == 0) {
            int cus = 0;
            if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device) != hipSuccess) cus = 256;
            h->ncu = cus;
        }
        std::vector<int> tab;
        int n = 0, nwg = 0;
        const int kr = sr_stream_items(h->Np, h->n_out, nc, h->ncu, can_fuse, tab, &n, &nwg);
        if (kr > 0) {
            if ((long)tab.size() > h->stream_tab_cap) {
                (void)hipStreamSynchronize(s);
                dev_free(h->stream_tab);
                h->stream_tab = nullptr; h->stream_tab_cap = 0;
                SR_TRY(dev_alloc(&h->stream_tab, tab.size()));
                h->stream_tab_cap = (long)tab.size();
            }
            SR_HIP(hipStreamSynchronize(s));                      // (a launch that still reads the old table)
            SR_HIP(hipMemcpy(h->stream_tab, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice));
        }
        h->stream_tab_key = key; h->stream_tab_kr = kr; h->stream_tab_n = n; h->stream_tab_nwg = nwg;
    }
    a.item_tab = h->stream_tab; a.kr = h->stream_tab_kr; a.nitems = h->stream_tab_n; a.nwg = h->stream_tab_nwg;
    return SR_OK;
}

static void stream_common(const sr_gp* h, sr_stream_args& a, int ncols, long Tp) {
    a.Wt = h->Wt; a.Ks = h->Ks; a.Vp = h->stream_vp; a.part = h->var_part; a.tickets = h->stream_tickets;
    a.N = h->N; a.Np = h->Np; a.D = h->D; a.n_out = h->n_out; a.k_lo = h->Np - h->N; a.ncols = ncols; a.ncols_pad = ncols;
    a.Tp = Tp;
    a.Z = h->Z; a.alpha = h->alpha; a.ls = h->ls; a.sf2 = h->sf2;
    a.mu_part_w = h->mu_part; a.jac_part_w = h->jac_part;
    a.host_flag = h->call_flag; a.host_seq = h->call_seq;
}

static int stream_predict(sr_gp* h, long Tc, const double* xa, long lda, int na, const double* xb, long ldb, int nb,
                          double* mu, double* var, double* jac, hipStream_t s) {
    const long Tp = srt::BN;
    const int ncb = (h->Np + 255) / 256;
    // 2 .. 4 queries on a moderate model: the one-launch VALU kernel has only ncb (ncb + 1) n_out workgroups of 256
    // threads there and loses to K1 + MFMA kernel + reduce (N = 700: 27 against 21 us; N = 3000: 34 against 40 us)
    const bool mfma_small = Tc >= 2 && Tc <= 4 && h->Np <= SR_MFMA_SMALL_MAX_NP;
    // 5 .. 32 queries (and 2 .. 4 on the MFMA kernel) where the MFMA route runs one-chunk work items of 16 or 32 columns:
    // the workgroups evaluate their rows of K* themselves too, no K* pass in front (T = 16, N = 1024 / 1500 / 2000 / 3000:
    // 23 / 26 / 28 / 39 -> 21 / 23 / 25 / 34 us; T = 32, N = 1024 / 1500 / 2000: 25 / 28 / 35 -> 22 / 25 / 31 us).  Every
    // column block re-evaluates a chunk's rows, so wide work items on models of many column blocks lose (32 columns at
    // N = 3000: 49 -> 53 us, 64 columns: 76 -> 93 us): those, and the run kernel of the big models, read K* from its pass.
    bool fused_mfma = false;
    if (!h->general && h->D <= SR_STREAM_FUSED_MAX_D && (Tc > 4 || mfma_small) && h->small_path != 0) {
        int g = 0, kc = 0;
        sr_stream_plan(h->Np, h->n_out, std::max(sr_stream_width((int)Tc), mfma_small ? 16 : 0), &g, &kc, true);
        fused_mfma = kc == 1 && (g == 1 || (g == 2 && ncb <= SR_STREAM_FUSED32_MAX_NCB));
    }
    const bool fused = (!h->general && h->D <= SR_STREAM_FUSED_MAX_D && Tc <= 4 && !mfma_small) || fused_mfma;
    const int nsplit = fused ? 2 * ncb : pick_nsplit(h, Tp);
    SR_TRY(ensure_ws(h, Tp, std::max(nsplit, 2 * ncb)));
    SR_TRY(stream_buffers(h, mfma_small ? 16 : (int)Tc, s));
    if (!fused) {
        sr_kstar_args ka;
        ka.Z = h->Z; ka.alpha = h->alpha; ka.ls = h->ls; ka.sf2 = h->sf2;
        ka.kp = h->general ? h->kp : nullptr; ka.kxx = h->kxx;
        ka.xa = xa; ka.lda = lda; ka.na = na; ka.xb = xb; ka.ldb = ldb; ka.nb = nb;
        ka.Ks = h->Ks; ka.mu_part = h->mu_part; ka.jac_part = h->jac_part;
        ka.N = h->N; ka.Np = h->Np; ka.D = h->D; ka.n_out = h->n_out; ka.nsplit = nsplit; ka.T = Tc; ka.Tp = Tp;
        sr_prof_scope ps(&h->prof, SR_K_KSTAR, s);
        SR_TRY(sr_launch_kstar(ka, s));
    }
    sr_stream_args a{};
    stream_common(h, a, (int)Tc, Tp);
    a.mode = 0; a.dot0 = 0; a.width_min = mfma_small ? 16 : 0;
    a.xa = xa; a.lda = lda; a.na = na; a.xb = xb; a.ldb = ldb;
    a.fa.mu_part = h->mu_part; a.fa.jac_part = h->jac_part; a.fa.var_part = h->var_part; a.fa.sf2 = h->sf2;
    a.fa.ls = h->ls; a.fa.kxx = h->general ? h->kxx : nullptr; a.fa.mu = mu; a.fa.var = var; a.fa.jac = jac;
    a.fa.n_out = h->n_out; a.fa.D = h->D; a.fa.nsplit = nsplit; a.fa.nrb = ncb; a.fa.T = Tc; a.fa.Tp = Tp;
    // ONE query evaluated in the kernel: the polling finaliser and its self-validating slots (sr_stream1_kernel) where they
    // fit its LDS; their pattern is "empty" from the allocation on, the finaliser leaves it behind again
    static const bool st1_poll = sr_lab_env("SR_ST1_POLL", 1) != 0;           // (lab build: A/B against the two ticket levels)
    const long nslot = sr_st1_slots(ncb, h->n_out, h->D);
    if (fused && Tc == 1 && !fused_mfma && nslot <= SR_ST1_SLOTS_MAX && 2 * ncb <= 64 && st1_poll) {
        if (h->stream_slots_cap < nslot) {
            (void)hipStreamSynchronize(s);
            dev_free(h->stream_slots);
            h->stream_slots = nullptr; h->stream_slots_cap = 0;
            SR_TRY(dev_alloc(&h->stream_slots, (size_t)SR_ST1_SLOTS_MAX));
            h->stream_slots_cap = SR_ST1_SLOTS_MAX;
            double empty;
            const unsigned long long bits = SR_ST1_EMPTY;
            memcpy(&empty, &bits, sizeof(double));
            SR_TRY(sr_launch_fill(h->stream_slots, (size_t)SR_ST1_SLOTS_MAX, empty, s));
        }
        a.slots = h->stream_slots;
    }
    SR_TRY(stream_items(h, a, (int)Tc, a.width_min, fused, s));
    h->last_streamed = 0;
    sr_prof_scope ps(&h->prof, SR_K_VAR, s);
    return sr_launch_stream(a, fused ? 1 : 0, s);
}

static int stream_linearize(sr_gp* h, const double* x, double* mu, double* var, double* jac_mu, double* jac_var,
                            double* hess_mu, hipStream_t s) {
    const long Tp = srt::BN;
    const int ncb = (h->Np + 255) / 256, ncols = 1 + h->D;
    const bool fused = !h->general && h->D <= SR_LIN_FUSED_MAX_D;
    SR_TRY(ensure_ws(h, Tp, std::max(pick_nsplit(h, Tp), 2 * ncb)));
    SR_TRY(stream_buffers(h, ncols, s));
    const int nblk256 = (h->Np + 255) / 256;
    const size_t need = (size_t)h->n_out * std::max(nblk256, 2 * ncb) * sr_lin_nacc(h->D);
    if (!h->lin_v || h->lin_cap < need) {
        (void)hipStreamSynchronize(s);
        dev_free(h->lin_v);
        h->lin_v = nullptr; h->lin_cap = 0;
        SR_TRY(dev_alloc(&h->lin_v, std::max(need, (size_t)h->n_out * h->Np)));
        h->lin_cap = std::max(need, (size_t)h->n_out * h->Np);
    }
    sr_lin_args la;
    la.Z = h->Z; la.alpha = h->alpha; la.ls = h->ls; la.sf2 = h->sf2; la.Ks = h->Ks; la.g = nullptr;
    la.x = x; la.xb = nullptr; la.na = h->D;
    la.kp = h->general ? h->kp : nullptr;
    la.jac_var = jac_var; la.hess_mu = hess_mu;
    la.N = h->N; la.Np = h->Np; la.D = h->D; la.n_out = h->n_out; la.Tp = Tp;
    if (!fused) {
        sr_prof_scope ps(&h->prof, SR_K_KSTAR, s);
        SR_TRY(sr_launch_lin_columns(la, sr_stream_width(ncols), h->Ks, h->lin_v, s));
    }
    sr_stream_args a{};
    stream_common(h, a, ncols, Tp);
    a.mode = 1; a.dot0 = 1;
    a.la = la; a.lin_part = h->lin_v; a.lin_part_w = h->lin_v;
    a.nblk = fused ? 2 * ncb : nblk256;
    a.lin_dt = h->D <= 3 ? 3 : (h->D <= 5 ? 5 : (h->D <= 8 ? 8 : 12));
    a.lmu = mu; a.lvar = var; a.ljac_mu = jac_mu;
    SR_TRY(stream_items(h, a, ncols, 0, false, s));
    h->last_streamed = 0;
    sr_prof_scope ps(&h->prof, SR_K_VAR, s);
    return sr_launch_stream(a, fused ? 2 : 0, s);
}

// GP posterior of Tc queries x = [xa | xb] into (mu, var, jac) in API layout (jac may be NULL).
// The tile kernels read U^-1 in 16-byte pieces; after an odd number of in-place one-point appends the view is 8 bytes off
// and the model goes back to plain buffers first (unslide: a copy of the factor and a device-wide wait -- this call then
// BLOCKS).  So that a loop of "one append, one big batch" does not pay that every step, the in-place route of the append
// is held off for the next 64 one-point appends, twice as many after every further forced unslide (reset by a refit).
static int tile_route_alignment(sr_gp* h) {
    if (!(h->slide & 1)) return SR_OK;
    SR_TRY(unslide(h));
    const int shift = std::min(h->slide_forced, 14);
    h->slide_hold = 64 << shift;
    ++h->slide_forced;
    return SR_OK;
}

int srh::gp_pass(sr_gp* h, long Tc, const double* xa, long lda, int na, const double* xb, long ldb,
                   int nb, double* mu, double* var, double* jac, hipStream_t s) {
    // (ONE query against Np = 384: the one-launch pass is a single workgroup per output that fetches 590 KB of U^-1 on its
    //  own, 18.8 us; the streamed route spreads them over 6 workgroups per output: 12.4 us.  From 4 queries on the two
    //  are level, and 16 queries share one fetch in the one-launch pass.)
    const bool one_streamed = h->Np == SR_ONE_STREAMED_NP && Tc == 1 && !h->general && h->D <= SR_STREAM_FUSED_MAX_D;
    if (h->small_path == 1 && !h->force_stream && !one_streamed && sr_gp_small_wanted(h->Np, Tc, h->D, h->general != 0)) {
        // small model, few queries: one launch, no workspace (sr_small.hip)
        sr_kstar_args ka{};
        ka.Z = h->Z; ka.alpha = h->alpha; ka.ls = h->ls; ka.sf2 = h->sf2;
        ka.kp = h->general ? h->kp : nullptr;
        ka.xa = xa; ka.lda = lda; ka.na = na; ka.xb = xb; ka.ldb = ldb; ka.nb = nb;
        ka.N = h->N; ka.Np = h->Np; ka.D = h->D; ka.n_out = h->n_out; ka.nsplit = 1;
        ka.T = Tc; ka.Tp = Tc;
        h->last_streamed = 0;
        sr_prof_scope ps(&h->prof, SR_K_SMALL, s);
        return sr_launch_gp_small(ka, h->Wt, mu, var, jac, s);
    }
    if (h->small_path != 0 && !h->force_stream && (h->Np > SR_STREAM_MIN_NP || (one_streamed && h->small_path == 1)) &&
        Tc <= stream_max_t())
        return stream_predict(h, Tc, xa, lda, na, xb, ldb, nb, mu, var, jac, s);   // U^-1 streamed once, 1-3 launches
    const long Tp = round_up(Tc, srt::BN);
    const int nsplit = pick_nsplit(h, Tp);
    SR_TRY(ensure_ws(h, Tp, nsplit));
    sr_kstar_args ka;
    ka.Z = h->Z; ka.alpha = h->alpha; ka.ls = h->ls; ka.sf2 = h->sf2;
    ka.kp = h->general ? h->kp : nullptr; ka.kxx = h->kxx;
    ka.xa = xa; ka.lda = lda; ka.na = na; ka.xb = xb; ka.ldb = ldb; ka.nb = nb;
    ka.Ks = h->Ks; ka.mu_part = h->mu_part; ka.jac_part = h->jac_part;
    ka.N = h->N; ka.Np = h->Np; ka.D = h->D; ka.n_out = h->n_out; ka.nsplit = nsplit;
    ka.T = Tc; ka.Tp = Tp;
    const bool small_var = h->small_path && ((Tc <= SR_SMALL_T && (h->Np > SR_STREAM_MIN_NP || h->force_stream)) ||
                                             (h->small_path == 1 && h->Np > SR_STREAM_MIN_NP &&
                                              Tc <= (long)SR_SMALL_T * sr_var_small_groups_max(h->Np, h->n_out)));
    // (Round 2 could run the K* pass of later column ranges of a big batch on a side stream beside the contraction of the
    //  earlier ones -- the K* pass is bound by its HBM writes, the contraction by the MFMA pipe.  Measured again in round 3,
    //  interleaved on one box, six runs each: 1.3672 +- 0.0012 M evals/s without, 1.3585 +- 0.0011 with 4 ranges: the
    //  contraction's launches stretch by more than the 1.3 ms the overlap hides.  Removed.)
    {
        sr_prof_scope ps(&h->prof, SR_K_KSTAR, s);
        SR_TRY(sr_launch_kstar(ka, s));
    }
    int nrb = h->Np / SR_NB;
    const double* var_part = h->var_part;
    h->last_streamed = 0;
    if (small_var) {
        h->last_streamed = 1;
        // latency regime: stream U^-1 once (HBM-bound) instead of the MFMA tiles
        if (!h->small_vp) SR_TRY(dev_alloc(&h->small_vp, (size_t)sr_var_small_ws(h->Np, h->n_out)));
        sr_prof_scope ps(&h->prof, SR_K_VAR, s);
        SR_TRY(sr_launch_var_small(h->Wt, h->Ks, h->small_vp, h->var_part, h->N, h->Np, Tp, h->n_out, (int)Tc, s));
        nrb = (h->Np + 255) / 256;
    } else if (h->small_path && sr_var_splitk_wanted(h->Np, Tp, h->n_out)) {
        // few query tiles: equal shares of the k-blocks of all tiles, the segments of a tile added by a second launch
        const long need = sr_var_bal_ws(h->Np, Tp, h->n_out);
        if (need > h->splitk_cap) {
            (void)hipStreamSynchronize(s);
            dev_free(h->splitk_vt);
            h->splitk_vt = nullptr; h->splitk_cap = 0;
            SR_TRY(dev_alloc(&h->splitk_vt, (size_t)need));
            h->splitk_cap = need;
        }
        if (!h->splitk_part) SR_TRY(dev_alloc(&h->splitk_part, (size_t)4 * 1024 * srt::BN));
        var_part = h->splitk_part;
        nrb = 4 * (h->Np / SR_NB);
        sr_prof_scope ps(&h->prof, SR_K_VAR, s);
        SR_TRY(tile_route_alignment(h));
        SR_TRY(sr_launch_var_bal(h->Wt, h->Ks, h->splitk_vt, h->splitk_part, h->N, h->Np, Tp, h->n_out, s));
    } else if (h->small_path && sr_var64_wanted(h->Np, Tp, h->n_out)) {
        // small model, few tiles: 64 x 64 workgroup tiles shorten the critical path of the tiny grid
        if (!h->splitk_part) SR_TRY(dev_alloc(&h->splitk_part, (size_t)4 * 1024 * srt::BN));
        var_part = h->splitk_part;             // n_out * (Np/64) * Tp <= 2 * 256 * 128 * 16 doubles
        nrb = h->Np / 64;
        sr_prof_scope ps(&h->prof, SR_K_VAR, s);
        SR_TRY(tile_route_alignment(h));
        SR_TRY(sr_launch_var64(h->Wt, h->Ks, h->splitk_part, h->N, h->Np, Tp, h->n_out, s));
    } else {
        sr_prof_scope ps(&h->prof, SR_K_VAR, s);
        SR_TRY(tile_route_alignment(h));
        SR_TRY(sr_launch_var(h->Wt, h->Ks, h->var_part, h->N, h->Np, Tp, h->n_out, h->var_group, h->var_variant, s));
    }
    sr_final_args fa;
    fa.mu_part = h->mu_part; fa.jac_part = h->jac_part; fa.var_part = var_part; fa.sf2 = h->sf2;
    fa.ls = h->ls; fa.kxx = h->general ? h->kxx : nullptr; fa.mu = mu; fa.var = var; fa.jac = jac;
    fa.n_out = h->n_out; fa.D = h->D; fa.nsplit = nsplit; fa.nrb = nrb; fa.T = Tc; fa.Tp = Tp;
    {
        sr_prof_scope ps(&h->prof, SR_K_FINAL, s);
        SR_TRY(sr_launch_finalize(fa, s));
    }
    return SR_OK;
}

extern "C" int sr_gp_predict(sr_gp_t h, const double* Xq, long T, double* mu, double* var,
                             double* jac, void* stream) {
    SR_CHECK(h != nullptr, SR_EINVAL, "sr_gp_predict: NULL handle");
    SR_CHECK(h->factorized, SR_ESTATE, "sr_gp_predict: model not factorized");
    SR_CHECK(T >= 0, SR_EINVAL, "sr_gp_predict: T=%ld", T);
    if (T == 0) return SR_OK;
    SR_CHECK(Xq && mu && var, SR_EINVAL, "sr_gp_predict: NULL argument");
    hipStream_t s = (hipStream_t)stream;
    SR_DEVICE(h->device);
    for (long t0 = 0; t0 < T; t0 += h->chunk) {
        const long Tc = std::min(h->chunk, T - t0);
        SR_TRY(gp_pass(h, Tc, Xq + t0 * h->D, h->D, h->D, nullptr, 0, 0, mu + t0 * h->n_out,
                       var + t0 * h->n_out, jac ? jac + t0 * h->n_out * h->D : nullptr, s));
    }
    return SR_OK;
}

extern "C" int sr_gp_linearize(sr_gp_t h, const double* x, double* mu, double* var, double* jac_mu,
                               double* jac_var, double* hess_mu, void* stream) {
    SR_CHECK(h != nullptr, SR_EINVAL, "sr_gp_linearize: NULL handle");
    SR_CHECK(h->factorized, SR_ESTATE, "sr_gp_linearize: model not factorized");
    SR_CHECK(x && mu && var && jac_mu && jac_var && hess_mu, SR_EINVAL, "sr_gp_linearize: NULL argument");
    hipStream_t s = (hipStream_t)stream;
    SR_DEVICE(h->device);
    if (h->small_path == 1 && sr_gp_small_lin_wanted(h->Np, h->D, h->general != 0)) {
        // small model: everything in one launch (sr_small.hip, LIN mode; the general family in its own kernel)
        sr_kstar_args ka{};
        ka.Z = h->Z; ka.alpha = h->alpha; ka.ls = h->ls; ka.sf2 = h->sf2;
        ka.kp = h->general ? h->kp : nullptr;
        ka.xa = x; ka.lda = h->D; ka.na = h->D; ka.xb = nullptr; ka.ldb = 0; ka.nb = 0;
        ka.N = h->N; ka.Np = h->Np; ka.D = h->D; ka.n_out = h->n_out; ka.nsplit = 1; ka.T = 1; ka.Tp = 1;
        sr_prof_scope ps(&h->prof, SR_K_SMALL, s);
        return sr_launch_gp_small_lin(ka, h->Wt, mu, var, jac_mu, jac_var, hess_mu, s);
    }
    if (h->small_path != 0)
        return stream_linearize(h, x, mu, var, jac_mu, jac_var, hess_mu, s);
    if (!h->lin_v) { SR_TRY(dev_alloc(&h->lin_v, (size_t)h->n_out * h->Np)); h->lin_cap = (size_t)h->n_out * h->Np; }
    if (!h->lin_g) SR_TRY(dev_alloc(&h->lin_g, (size_t)h->n_out * h->Np));
    h->force_stream = 1;
    const int rc_pass = gp_pass(h, 1, x, h->D, h->D, nullptr, 0, 0, mu, var, jac_mu, s);   // leaves K*(:,0) in the workspace
    h->force_stream = 0;
    SR_TRY(rc_pass);
    const long Tp = srt::BN;
    if (h->last_streamed)    // v = U^-T k* is what the streaming variance pass just accumulated
        SR_TRY(sr_launch_var_small_gather(h->small_vp, h->lin_v, h->Np, h->n_out, 0, 1, s));
    for (int d = 0; d < h->n_out; ++d) {
        const double* Wt = h->Wt + (size_t)d * h->Np * h->Np;
        const double* ks = h->Ks + (size_t)d * h->Np * Tp;
        if (!h->last_streamed)
            SR_TRY(sr_launch_trmv_t(Wt, h->Np, ks, Tp, h->lin_v + (size_t)d * h->Np, h->Np, s));  // v = U^-T k*
        SR_TRY(sr_launch_trmv(Wt, h->Np, h->lin_v + (size_t)d * h->Np, h->lin_g + (size_t)d * h->Np,
                              h->Np, 0, s));                                                       // g = U^-1 v
    }
    sr_lin_args la;
    la.Z = h->Z; la.alpha = h->alpha; la.ls = h->ls; la.sf2 = h->sf2; la.Ks = h->Ks; la.g = h->lin_g; la.x = x;
    la.kp = h->general ? h->kp : nullptr;
    la.jac_var = jac_var; la.hess_mu = hess_mu;
    la.N = h->N; la.Np = h->Np; la.D = h->D; la.n_out = h->n_out; la.Tp = Tp;
    return sr_launch_linearize(la, s);
}

// ---- GP input transform (gp_reachability_casadi.py:60-61,85,94-97; uncertainty_propagation_casadi.py:40-47,60):
// the GP sees x_gp = Tz x (e.g. the cart-pole model without the cart position: D = 4), its Jacobian with respect to
// the state is jac[:, :n_xin] Tz.
__global__ __launch_bounds__(256) void sr_tz_apply_kernel(const double* __restrict__ p, long ldp,
                                                          const double* __restrict__ Tz, double* __restrict__ xbar,
                                                          long T, int n_s, int n_xin) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= T * n_xin) return;
    const long t = e / n_xin;
    const int i = (int)(e % n_xin);
    double v = 0.0;
    for (int j = 0; j < n_s; ++j) v = fma(Tz[i * n_s + j], p[t * ldp + j], v);
    xbar[e] = v;
}

// jacs[t][o][:n_s] = jacg[t][o][:n_xin] Tz ,  jacs[t][o][n_s:] = jacg[t][o][n_xin:]
__global__ __launch_bounds__(256) void sr_tz_jac_kernel(const double* __restrict__ jacg, const double* __restrict__ Tz,
                                                        double* __restrict__ jacs, long T, int n_out, int n_s, int n_xin,
                                                        int n_u) {
    const int Ds = n_s + n_u, Dg = n_xin + n_u;
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= T * n_out * Ds) return;
    const long row = e / Ds;                              // (t, o)
    const int c = (int)(e % Ds);
    const double* g = jacg + row * Dg;
    double v;
    if (c < n_s) {
        v = 0.0;
        for (int i = 0; i < n_xin; ++i) v = fma(g[i], Tz[i * n_s + c], v);
    } else {
        v = g[n_xin + (c - n_s)];
    }
    jacs[e] = v;
}

extern "C" int sr_gp_set_input_transform(sr_gp_t h, const double* Tz, int n_x_in, void* stream) {
    SR_CHECK(h != nullptr, SR_EINVAL, "sr_gp_set_input_transform: NULL handle");
    SR_DEVICE(h->device);
    // (the resident server answers in the GP's input space and never reads Tz: it stays on the device)
    if (Tz == nullptr) { h->n_xin = 0; return SR_OK; }
    SR_CHECK(n_x_in >= 1 && n_x_in < h->D, SR_EINVAL, "sr_gp_set_input_transform: n_x_in=%d with D=%d", n_x_in, h->D);
    if (!h->Tz) SR_TRY(dev_alloc(&h->Tz, (size_t)SR_MAX_D * SR_MAX_NS));
    SR_CHECK(h->n_out <= SR_MAX_NS, SR_EUNSUPPORTED, "sr_gp_set_input_transform: n_out=%d > %d", h->n_out, SR_MAX_NS);
    SR_HIP(hipMemcpyAsync(h->Tz, Tz, sizeof(double) * n_x_in * h->n_out, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    h->n_xin = n_x_in;
    return SR_OK;
}

static int ensure_tz(sr_gp* h, long Tc, int n_s, int n_u) {
    if (h->n_xin == 0 || Tc <= h->tz_cap) return SR_OK;
    (void)device_sync();
    dev_free(h->tz_x); dev_free(h->tz_jac);
    h->tz_x = h->tz_jac = nullptr; h->tz_cap = 0;
    SR_TRY(dev_alloc(&h->tz_x, (size_t)Tc * SR_MAX_D));
    // (sized for any later transform of this handle: a smaller n_x_in means a larger n_u = D - n_x_in)
    SR_TRY(dev_alloc(&h->tz_jac, (size_t)Tc * SR_MAX_NS * (SR_MAX_NS + SR_MAX_D)));
    h->tz_cap = Tc;
    return SR_OK;
}

// GP posterior at the (possibly transformed) states p [ldp] and controls k_ff [ldkff]: mu, var into the given buffers,
// the Jacobian with respect to [state; control] (T x n_s x (n_s + n_u)) behind *jac_out.
int srh::gp_pass_states(sr_gp* h, long Tc, const double* p, long ldp, int n_s, const double* kff, long ldkff, int n_u,
                          double* mu, double* var, const double** jac_out, hipStream_t s) {
    if (h->n_xin == 0) {
        *jac_out = h->jac;
        return gp_pass(h, Tc, p, ldp, n_s, kff, ldkff, n_u, mu, var, h->jac, s);
    }
    SR_TRY(ensure_tz(h, Tc, n_s, n_u));
    hipLaunchKernelGGL(sr_tz_apply_kernel, dim3((unsigned)((Tc * h->n_xin + 255) / 256)), dim3(256), 0, s, p, ldp, h->Tz,
                       h->tz_x, Tc, n_s, h->n_xin);
    SR_HIP(hipGetLastError());
    SR_TRY(gp_pass(h, Tc, h->tz_x, h->n_xin, h->n_xin, kff, ldkff, n_u, mu, var, h->jac, s));
    hipLaunchKernelGGL(sr_tz_jac_kernel, dim3((unsigned)((Tc * h->n_out * (n_s + n_u) + 255) / 256)), dim3(256), 0, s,
                       h->jac, h->Tz, h->tz_jac, Tc, h->n_out, n_s, h->n_xin, n_u);
    SR_HIP(hipGetLastError());
    *jac_out = h->tz_jac;
    return SR_OK;
}

// ---- completion mailbox of the single-query host entry points --------------------------------------------------
// A host that waits for ONE small result (the CasADi / IPOPT callback: state_space_models.py:271-303 calls the model,
// blocks, and returns NumPy arrays) pays for three dependent commands from an idle queue (H2D copy, kernel, D2H copy)
// plus the completion signal of the queue.  sr_publish replaces the D2H copy + hipStreamSynchronize: a kernel copies
// the results from device memory into PINNED host memory with system-scope stores and then writes a sequence number
// next to them; the host spins on that number (sr_wait_flag).  Measured (scripts/call_latency.py): __call__ at N = 200
// 33.7 -> 29.4 us, N = 5000 59.0 -> 55.2 us (the kernel alone, launched back to back: 13 resp. 37.6 us) -- most of the
// rest is dispatch latency of the remaining commands; folding the query into the kernel arguments and the mailbox
// write into the posterior kernels would remove two of the three.
__global__ __launch_bounds__(256) void sr_publish_kernel(const double* __restrict__ src, int n, double* dst,
                                                         unsigned long long* flag, unsigned long long seq) {
    for (int e = threadIdx.x; e < n; e += 256)
        __hip_atomic_store(dst + e, src[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

extern "C" int sr_publish(int device, const double* src_dev, int n, double* dst_host, unsigned long long* flag_host,
                          unsigned long long seq, void* stream) {
    SR_CHECK(src_dev && dst_host && flag_host && n >= 0, SR_EINVAL, "sr_publish: bad argument");
    SR_DEVICE(device);
    double* dst_dev = nullptr;
    unsigned long long* flag_dev = nullptr;
    // pinned (hipHostMalloc / hipHostRegister) memory only: resolves the address the device uses for it
    SR_HIP(hipHostGetDevicePointer((void**)&dst_dev, dst_host, 0));
    SR_HIP(hipHostGetDevicePointer((void**)&flag_dev, flag_host, 0));
    hipLaunchKernelGGL(sr_publish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, src_dev, n, dst_dev, flag_dev, seq);
    SR_HIP(hipGetLastError());
    return SR_OK;
}

// hipStreamSynchronize for a host layer that holds the raw stream (PyTorch builds a Stream object per current_stream() call:
// 2 - 3 us of the 40 a blocking NumPy-level call takes)
// (on `device`: the null stream -- PyTorch's default stream has the raw handle 0 -- is the CURRENT device's)
extern "C" int sr_stream_synchronize(int device, void* stream) {
    SR_DEVICE(device);
    SR_HIP(hipStreamSynchronize((hipStream_t)stream));
    return SR_OK;
}

// 1 where kernels of `device` may be handed the HOST address of this pinned block as it is (the device sees the block at the
// same address), 0 otherwise -- what a host layer asks once before it lets kernels read / write its pinned staging blocks.
// (and where the device's ATOMIC adds reach pinned host memory: the kernels count violated bounds with atomicAdd on a word of
//  the result block, and over a link without PCIe AtomicOps such an add can be dropped silently.  A property of the device
//  and its link, not of the block: probed ONCE per device and process -- 64 adds on a pinned word of the library's own, on
//  the null stream; a stream of its own per probe cost a caller milliseconds every time it pinned a new block.)
__global__ void sr_probe_atomic_kernel(int* w) { atomicAdd(w, 1); }

static int host_atomics_reach(int device) {
    static std::mutex mu;
    static int cached[64];                     // 0 unknown, 1 yes, -1 no
    std::lock_guard<std::mutex> lk(mu);
    if (device < 0 || device >= 64) return 0;
    if (cached[device] != 0) return cached[device] > 0;
    int* w = nullptr;
    int* wd = nullptr;
    bool ok = hipHostMalloc((void**)&w, 64, hipHostMallocMapped) == hipSuccess;
    if (ok) {
        *w = 0;
        ok = hipHostGetDevicePointer((void**)&wd, w, 0) == hipSuccess;
        if (ok) {
            hipLaunchKernelGGL(sr_probe_atomic_kernel, dim3(1), dim3(64), 0, (hipStream_t)nullptr, wd);
            ok = hipGetLastError() == hipSuccess && hipStreamSynchronize((hipStream_t)nullptr) == hipSuccess;
        }
        ok = ok && *(volatile int*)w == 64;
        (void)hipHostFree(w);
    }
    if (!ok) (void)hipGetLastError();
    cached[device] = ok ? 1 : -1;
    return ok ? 1 : 0;
}

extern "C" int sr_host_block_is_device_visible(int device, const void* host_block) {
    if (!host_block) return 0;
    sr_dev_guard guard(device);
    void* dp = nullptr;
    if (hipHostGetDevicePointer(&dp, const_cast<void*>(host_block), 0) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    if (dp != host_block) return 0;
    return host_atomics_reach(device);
}

extern "C" int sr_wait_flag(const unsigned long long* flag_host, unsigned long long seq, double timeout_s) {
    SR_CHECK(flag_host != nullptr, SR_EINVAL, "sr_wait_flag: NULL flag");
    const volatile unsigned long long* f = flag_host;
    const auto t0 = std::chrono::steady_clock::now();
    // The answer of a small model is there within 10 .. 60 us: spin.  A caller that is still waiting after 200 us is
    // behind other work on the device: give the core away between looks, and sleep between them after 5 ms -- a wait
    // that runs into its time-out (seconds) must not burn a core for it.
    for (;;) {
        for (int spin = 0; spin < 2048; ++spin) {
            if (*f == seq) { std::atomic_thread_fence(std::memory_order_acquire); return SR_OK; }
            cpu_relax();
        }
        const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (waited > timeout_s) break;
        if (waited > 5e-3) std::this_thread::sleep_for(std::chrono::microseconds(50));
        else if (waited > 2e-4) std::this_thread::yield();
    }
    sr_set_error("sr_wait_flag: sequence %llu not seen within %.3f s (flag = %llu)", seq, timeout_s, *f);
    return SR_ESTATE;
}

// One blocking single query in ONE command: x (host memory, D doubles, read NOW) travels in the kernel arguments, the
// results go straight to the pinned host block out_host = [mu n | var n | jac_mu n x D (| jac_var n x D | hess n x D x D)]
// and the last workgroup writes `seq` to *flag_host (both pinned; wait with sr_wait_flag).  Only where the one-launch
// posterior of sr_small.hip applies (ARD-RBF, Np <= 384, and 512 with second order); SR_EUNSUPPORTED otherwise -- the
// caller then takes sr_gp_predict / sr_gp_linearize with its own copies.
// replaces the body of SimpleGPModel.__call__ / linearize_predict as CasadiSSMEvaluator drives them
// (/root/reference/safe_exploration/state_space_models.py:271-303, 384-417; ssm_gpy/gaussian_process.py:135-144).
extern "C" int sr_gp_call1(sr_gp_t h, const double* x_host, int second_order, double* out_host,
                           unsigned long long* flag_host, unsigned long long seq, void* stream) {
    SR_CHECK(h != nullptr && x_host && out_host && flag_host, SR_EINVAL, "sr_gp_call1: NULL argument");
    SR_CHECK(h->factorized, SR_ESTATE, "sr_gp_call1: model not factorized");
    // (x is in the GP's input space, as for sr_gp_predict: an input transform of the reachability entry points does not matter)
    const bool one_launch = h->small_path == 1 &&
                            (second_order ? sr_gp_small_lin_wanted(h->Np, h->D, h->general != 0)
                                          : sr_gp_small_wanted(h->Np, 1, h->D, h->general != 0));
    // beyond the one-launch sizes: the streamed route (one to three launches) with the query READ from the pinned input block
    // and outputs and sequence number WRITTEN to the pinned result block by the workgroup that runs the final stage
    const bool streamed = !one_launch && h->small_path == 1 && !h->force_stream && h->Np >= SR_FUSED_NP;
    if (!one_launch && !streamed) {
        sr_set_error("sr_gp_call1: no one-command route for this model (Np=%d, general=%d)", h->Np, h->general);
        return SR_EUNSUPPORTED;
    }
    SR_DEVICE(h->device);
    hipStream_t s = (hipStream_t)stream;
    if (streamed) {
        double *out_d = nullptr, *x_d = nullptr;
        unsigned long long* flag_d = nullptr;
        if (hipHostGetDevicePointer((void**)&out_d, out_host, 0) != hipSuccess ||
            hipHostGetDevicePointer((void**)&flag_d, flag_host, 0) != hipSuccess ||
            hipHostGetDevicePointer((void**)&x_d, const_cast<double*>(x_host), 0) != hipSuccess) {
            (void)hipGetLastError();
            sr_set_error("sr_gp_call1: the query / result / flag block is not device-visible pinned memory");
            return SR_EUNSUPPORTED;
        }
        const int n = h->n_out, D = h->D;
        h->call_flag = flag_d; h->call_seq = seq;
        int rc;
        if (second_order) rc = stream_linearize(h, x_d, out_d, out_d + n, out_d + 2 * n, out_d + 2 * n + n * D, out_d + 2 * n + 2 * n * D, s);
        else rc = stream_predict(h, 1, x_d, D, D, nullptr, 0, 0, out_d, out_d + n, out_d + 2 * n, s);
        h->call_flag = nullptr; h->call_seq = 0;
        return rc;
    }
    if (!h->call_ticket) {
        SR_TRY(dev_alloc(&h->call_ticket, 1));
        SR_TRY(dev_zero(h->call_ticket, sizeof(unsigned)));
    }
    double* out = nullptr;
    unsigned long long* flag = nullptr;
    // a host block that is not device-visible here is a reason to take the launched routes with their own copies (the
    // caller's plain-copy fall-back), not an error of this call
    if (hipHostGetDevicePointer((void**)&out, out_host, 0) != hipSuccess ||
        hipHostGetDevicePointer((void**)&flag, flag_host, 0) != hipSuccess) {
        (void)hipGetLastError();
        sr_set_error("sr_gp_call1: the result / flag block is not device-visible pinned memory");
        return SR_EUNSUPPORTED;
    }
    const int n = h->n_out, D = h->D;
    sr_kstar_args ka{};
    ka.Z = h->Z; ka.alpha = h->alpha; ka.ls = h->ls; ka.sf2 = h->sf2;
    ka.kp = h->general ? h->kp : nullptr;
    ka.xa = nullptr; ka.lda = D; ka.na = D; ka.xb = nullptr; ka.ldb = 0; ka.nb = 0;
    ka.N = h->N; ka.Np = h->Np; ka.D = D; ka.n_out = n; ka.nsplit = 1; ka.T = 1; ka.Tp = 1;
    ka.xv_on = 1;
    for (int j = 0; j < D; ++j) ka.xv[j] = x_host[j];
    ka.done_ticket = h->call_ticket; ka.host_flag = flag; ka.host_seq = seq;
    h->last_streamed = 0;
    sr_prof_scope ps(&h->prof, SR_K_SMALL, s);
    if (second_order)
        return sr_launch_gp_small_lin(ka, h->Wt, out, out + n, out + 2 * n, out + 2 * n + n * D, out + 2 * n + 2 * n * D, s);
    return sr_launch_gp_small(ka, h->Wt, out, out + n, out + 2 * n, s);
}

