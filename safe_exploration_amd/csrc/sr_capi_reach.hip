// sr_capi_reach.hip -- reachability entry points: one-step and multi-step ellipsoid propagation, Gaussian moment
// propagation, the stateless per-query steps (ellipsoid, remainder, safety distance, distance to centre, sampling)
// and the dispatch of the persistent chain kernel (sr_chain.hip K0c) with its process-wide launch gate.
#include "sr_handle.h"
using namespace srh;

static int check_reach_dims(const sr_gp* h, int* n_s, int* n_u) {
    *n_s = h->n_out;
    *n_u = h->D - (h->n_xin ? h->n_xin : h->n_out);
    SR_CHECK(*n_u >= 1, SR_EINVAL, "reachability needs D = n_s + n_u with n_u >= 1 (D=%d, n_out=%d)",
             h->D, h->n_out);
    SR_CHECK(*n_s <= SR_MAX_NS && *n_u <= SR_MAX_NU, SR_EUNSUPPORTED,
             "reachability supports n_s <= %d, n_u <= %d (got %d, %d)", SR_MAX_NS, SR_MAX_NU, *n_s, *n_u);
    return SR_OK;
}

static int try_chain(sr_gp* h, long T, int H, int mode, const double* p0, const double* q0, const double* k_fb0,
                     const double* k_ff, const double* k_fb, const double* a, const double* b, const double* l_mu,
                     const double* l_sigma, double c_safety, double* p_all, double* q_all, double* gp_var_all,
                     int* n_bad, int n_s, int n_u, hipStream_t s, bool* taken);

extern "C" int sr_onestep_reach(sr_gp_t h, long T, const double* p, const double* q,
                                const double* k_ff, const double* k_fb, const double* a,
                                const double* b, const double* l_mu, const double* l_sigma,
                                double c_safety, double* p_out, double* q_out, double* var_out,
                                int* n_bad, void* stream) {
    SR_CHECK(h != nullptr, SR_EINVAL, "sr_onestep_reach: NULL handle");
    SR_CHECK(h->factorized, SR_ESTATE, "sr_onestep_reach: model not factorized");
    SR_CHECK(T >= 0, SR_EINVAL, "sr_onestep_reach: T=%ld", T);
    if (T == 0) return SR_OK;
    SR_CHECK(p && k_ff && a && b && l_mu && l_sigma && p_out && q_out, SR_EINVAL,
             "sr_onestep_reach: NULL argument");
    SR_CHECK(q == nullptr || k_fb != nullptr, SR_EINVAL, "sr_onestep_reach: k_fb required with q");
    int n_s, n_u;
    SR_TRY(check_reach_dims(h, &n_s, &n_u));
    hipStream_t s = (hipStream_t)stream;
    SR_DEVICE(h->device);
    h->last_chain = 0;
    if (n_s <= 2) {
        // small model, few states: posterior and ellipsoid step in one launch (the chain kernel with H = 1): 24.5 ->
        // 21.5 us per call at N = 200 (host-bound from there).  Not for n_s >= 3: the Jacobi rotations of the
        // eigenvalue bound run on one lane per state inside a 512-thread workgroup there (cart-pole: 25 -> 35 us).
        bool chained = false;
        SR_TRY(try_chain(h, T, 1, 0, p, q, k_fb, k_ff, nullptr, a, b, l_mu, l_sigma, c_safety, p_out, q_out, var_out,
                         n_bad, n_s, n_u, s, &chained));
        if (chained) return SR_OK;
    }
    for (long t0 = 0; t0 < T; t0 += h->chunk) {
        const long Tc = std::min(h->chunk, T - t0);
        double* var_dst = var_out ? var_out + t0 * n_s : nullptr;
        // gp_pass must not (re)allocate the workspace once internal pointers are resolved
        SR_TRY(prepare_ws(h, Tc));
        sr_ws_lock lock(h);
        if (!var_dst) var_dst = h->var;
        const double* jac_su = nullptr;
        SR_TRY(gp_pass_states(h, Tc, p + t0 * n_s, n_s, n_s, k_ff + t0 * n_u, n_u, n_u, h->mu, var_dst, &jac_su, s));
        sr_ell_args ea;
        ea.T = Tc; ea.n_s = n_s; ea.n_u = n_u;
        ea.p = p + t0 * n_s; ea.ldp = n_s;
        ea.q = q ? q + t0 * n_s * n_s : nullptr; ea.ldq = (long)n_s * n_s;
        ea.k_ff = k_ff + t0 * n_u; ea.ldkff = n_u;
        ea.k_fb = k_fb ? k_fb + t0 * n_u * n_s : nullptr; ea.ldkfb = (long)n_u * n_s;
        ea.mu = h->mu; ea.var = var_dst; ea.jac = jac_su;
        ea.a = a; ea.b = b; ea.l_mu = l_mu; ea.l_sigma = l_sigma; ea.c_safety = c_safety;
        ea.p_out = p_out + t0 * n_s; ea.ldpo = n_s;
        ea.q_out = q_out + t0 * n_s * n_s; ea.ldqo = (long)n_s * n_s;
        ea.n_bad = n_bad; ea.mode = 0;
        sr_prof_scope ps(&h->prof, SR_K_ELL, s);
        SR_TRY(sr_launch_ellipsoid(ea, s));
    }
    return SR_OK;
}

// Persistent chain launches of one device never overlap, whichever handle or stream they come from: each needs (almost)
// every CU resident at once, two of them side by side would wait for each other's workgroups until the time-out.  Every
// launch waits for the event the previous one recorded (per device, process-wide) and records its own.  Not while the
// caller's stream is being captured into a graph (a cross-stream wait on an uncaptured event is not capturable): a
// captured chain is ordered by its graph.
struct sr_chain_gate { std::mutex m; hipEvent_t ev[32] = {}; hipStream_t last[32] = {}; bool any[32] = {}; };
static sr_chain_gate g_chain_gate;
struct sr_chain_turn {                 // holds the gate from the wait to the record: host threads take turns too
    int device; hipStream_t s; bool active = false;
    sr_chain_turn(int device_, hipStream_t s_) : device(device_), s(s_) {}
    int enter() {
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        const hipError_t ce = hipStreamIsCapturing(s, &st);
        const bool capturing = (ce == hipSuccess && st != hipStreamCaptureStatusNone);
        if (capturing || device < 0 || device >= 32) return SR_OK;
        g_chain_gate.m.lock();
        active = true;
        // (the previous launch on the SAME stream is ordered by the stream itself)
        if (g_chain_gate.ev[device] && !(g_chain_gate.any[device] && g_chain_gate.last[device] == s))
            SR_HIP(hipStreamWaitEvent(s, g_chain_gate.ev[device], 0));
        return SR_OK;
    }
    int leave() {
        if (!active) return SR_OK;
        if (!g_chain_gate.ev[device]) SR_HIP(hipEventCreateWithFlags(&g_chain_gate.ev[device], hipEventDisableTiming));
        SR_HIP(hipEventRecord(g_chain_gate.ev[device], s));
        g_chain_gate.last[device] = s; g_chain_gate.any[device] = true;
        return SR_OK;
    }
    ~sr_chain_turn() { if (active) g_chain_gate.m.unlock(); }
};

// The persistent kernel of sr_chain.hip (K0c) for a chain of H >= 1 steps, where it applies; *taken says whether it ran.
static int try_chain(sr_gp* h, long T, int H, int mode, const double* p0, const double* q0, const double* k_fb0,
                     const double* k_ff, const double* k_fb, const double* a, const double* b, const double* l_mu,
                     const double* l_sigma, double c_safety, double* p_all, double* q_all, double* gp_var_all,
                     int* n_bad, int n_s, int n_u, hipStream_t s, bool* taken) {
    *taken = false;
    const long nss = (long)n_s * n_s, nus = (long)n_u * n_s;
    // small model, few rollouts: the whole chain in one launch (sr_chain.hip K0c).  One launch holds SR_CHAIN_GROUPS
    // workgroups = gmax groups of 16 rollouts (n_s Np / 128 posterior workgroups + the tail workgroup each): 768 rollouts
    // of a pendulum model with N <= 256, 416 of a cart-pole model.  Measured at N = 200, H = 15: 256 rollouts 246
    // (per-step launches) -> 102 us.
    // every workgroup of a launch must be resident (one per CU): leave 16 CUs of whatever this device (or partition of
    // a device) has to other work
    if (h->chain_cap < 0) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device) != hipSuccess) cus = 0;
        h->chain_cap = std::max(0, std::min(SR_CHAIN_GROUPS, cus - 16));
    }
    const int wpg = sr_chain_wgs_per_group(h->Np, n_s);                 // posterior workgroups + the tail workgroup
    if (h->chain_cap < wpg) return SR_OK;                               // not even one group fits: per-step launches
    // A chain launch whose hand-off timed out (SR_CHAIN_TIMEOUT_TICKS: its workgroups were not co-resident in time) has
    // poisoned its outputs with NaN and raised the pinned status word.  The first entry after that reports it -- a
    // caller that never looked at the outputs must not go on with them -- and the handle takes the per-step launches
    // from here on (sr_gp_set_chain(h, 1) re-arms the persistent kernel).
    if (h->chain_status_host && *(volatile int*)h->chain_status_host != 0) {
        *(volatile int*)h->chain_status_host = 0;
        h->chain = 0;
        sr_set_error("a previous persistent multi-step launch timed out (its workgroups did not become co-resident within "
                     "100 ms); its outputs were filled with NaN.  The handle now uses per-step launches; repeat the call.");
        return SR_ESTATE;
    }
    const long xpg = std::max(1l, sr_chain_xels_per_group(h->Np, n_s, n_u, H));
    const int gmax = (int)std::max(1l, std::min((long)(h->chain_cap / wpg), (long)SR_CHAIN_XELS / xpg));   // groups of 16 rollouts per launch
    const long chain_launches = ((T + SR_SMALL_T - 1) / SR_SMALL_T + gmax - 1) / gmax;
    // Several launches in a row still beat the per-step route (N = 200, H = 15: 1024 rollouts 203 against 253 us, 4096
    // rollouts in six launches 608 against 722 us; cart-pole N = 150: 832 rollouts 304 against 411 us) -- except where
    // the per-step route still has its one-launch posterior (T <= SR_FUSED_T) and three launches are needed (cart-pole,
    // 960 rollouts: 454 against 413 us).
    // (one or two steps: a second launch costs what the per-step route's two launches per step cost -- N = 200, H = 1,
    //  960 rollouts: 29 against 24 us)
    const long chain_max_launches = sr_chain_max_launches(T, H);
    if (h->chain && h->small_path == 1 && !h->force_stream && !h->general && h->n_xin == 0 &&
        chain_launches <= chain_max_launches &&
        sr_chain_supported(h->Np, h->D, n_s, n_u, H)) {
        // the kernel must be able to run at all: at least one workgroup per CU (registers, static + dynamic LDS)
        const int occ_key = ((h->Np * 8 + n_s) * 8 + n_u) * 128 + std::min(H, 127);      // (H sizes the dynamic LDS; sr_chain_supported bounds it at 96)
        if (h->chain_occ_key != occ_key) {
            h->chain_occ_blocks = 0;
            SR_TRY(sr_chain_blocks_per_cu(h->Np, n_s, n_u, H, &h->chain_occ_blocks));
            h->chain_occ_key = occ_key;
        }
        if (h->chain_occ_blocks < 1) return SR_OK;                       // per-step launches
        if (!h->chain_xch) {
            // all or nothing: the pointers are committed to the handle only once every allocation and memset is in place
            // (a launch with one of them missing would run on NULL epoch / alive / done words)
            int* st_host = h->chain_status_host; int* st_dev = h->chain_status_dev;
            sr_xel* xch = nullptr; unsigned long long* tickets = nullptr; unsigned* done = nullptr;
            int rc = SR_OK;
            hipError_t he = hipSuccess;
            if (!st_host) {
                he = hipHostMalloc((void**)&st_host, sizeof(int), hipHostMallocMapped);
                if (he == hipSuccess) { *st_host = 0; he = hipHostGetDevicePointer((void**)&st_dev, st_host, 0); }
            }
            if (he == hipSuccess) rc = dev_alloc(&xch, (size_t)SR_CHAIN_XELS);
            if (he == hipSuccess && rc == SR_OK) rc = dev_alloc(&tickets, (size_t)2 * SR_CHAIN_GROUPS);
            if (he == hipSuccess && rc == SR_OK) rc = dev_alloc(&done, (size_t)SR_CHAIN_GROUPS);
            // ON THE CALLER'S STREAM: a memset on the null stream is not ordered with a launch on a non-blocking stream --
            // it wiped tags the first launch had already written (41 MB take 20 us) and that launch timed out
            if (he == hipSuccess && rc == SR_OK) he = hipMemsetAsync(xch, 0, sizeof(sr_xel) * (size_t)SR_CHAIN_XELS, s);     // tag 0 = never written
            if (he == hipSuccess && rc == SR_OK) he = hipMemsetAsync(tickets, 0, sizeof(unsigned long long) * 2 * SR_CHAIN_GROUPS, s);
            if (he == hipSuccess && rc == SR_OK) he = hipMemsetAsync(done, 0, sizeof(unsigned) * SR_CHAIN_GROUPS, s);
            if (he != hipSuccess || rc != SR_OK) {
                if (he != hipSuccess) {
                    (void)hipStreamSynchronize(s);
                    sr_set_error("persistent chain: setting up the exchange buffers -> %s", hipGetErrorString(he));
                    (void)hipGetLastError();
                    rc = SR_EHIP;
                }
                dev_free(xch); dev_free(tickets); dev_free(done);
                if (st_host && !h->chain_status_host) (void)hipHostFree(st_host);
                return rc;
            }
            h->chain_status_host = st_host; h->chain_status_dev = st_dev;
            h->chain_xch = xch; h->chain_tickets = tickets; h->chain_done = done;
        }
        // every workgroup of a chain launch must be resident at once: resident single-query servers (up to n_out Np / 64 CUs
        // each) leave the device first; they come back with their next query
        servers_quiesce_device(h->device);
        sr_prof_scope ps(&h->prof, SR_K_SMALL, s);
        for (long t0 = 0; t0 < T; t0 += (long)gmax * SR_SMALL_T) {
            const long Tc = std::min((long)gmax * SR_SMALL_T, T - t0);
            const int groups = (int)((Tc + SR_SMALL_T - 1) / SR_SMALL_T);
            sr_chain_args ca{};
            ca.k.Z = h->Z; ca.k.alpha = h->alpha; ca.k.ls = h->ls; ca.k.sf2 = h->sf2;
            ca.k.N = h->N; ca.k.Np = h->Np; ca.k.D = h->D; ca.k.n_out = h->n_out; ca.k.na = n_s; ca.k.nb = n_u;
            ca.Wt = h->Wt; ca.T = Tc; ca.H = H; ca.mode = mode;
            ca.p0 = p0 + t0 * n_s; ca.q0 = q0 ? q0 + t0 * nss : nullptr; ca.k_fb0 = k_fb0 ? k_fb0 + t0 * nus : nullptr;
            ca.k_ff = k_ff + t0 * H * n_u; ca.k_fb = k_fb ? k_fb + t0 * (H - 1) * nus : nullptr;
            ca.a = a; ca.b = b; ca.l_mu = l_mu; ca.l_sigma = l_sigma; ca.c_safety = c_safety;
            ca.p_all = p_all + t0 * H * n_s; ca.q_all = q_all + t0 * H * nss;
            ca.gp_var_all = gp_var_all ? gp_var_all + t0 * H * n_s : nullptr;
            ca.n_bad = n_bad; ca.xch = h->chain_xch;
            ca.epoch = h->chain_tickets; ca.alive = h->chain_tickets + SR_CHAIN_GROUPS; ca.done = h->chain_done;
            ca.status = h->chain_status_dev;
            ca.test_drop = h->chain_test_drop;
            sr_chain_turn turn(h->device, s);
            SR_TRY(turn.enter());
            SR_TRY(sr_launch_chain(ca, s));
            SR_TRY(turn.leave());
            (void)groups;
        }
        h->last_chain = 1;
        *taken = true;
        return SR_OK;
    }
    return SR_OK;
}

// shared H-step chain: mode 0 = robust ellipsoids (gp_reachability.py:159-212),
// mode 1/2 = Taylor / mean-equivalent Gaussian moments (uncertainty_propagation_casadi.py:88-190)
static int multistep_impl(sr_gp* h, long T, int H, int mode, const double* p0, const double* q0,
                          const double* k_fb0, const double* k_ff, const double* k_fb, const double* a,
                          const double* b, const double* l_mu, const double* l_sigma, double c_safety,
                          double* p_all, double* q_all, double* gp_var_all, int* n_bad, hipStream_t s) {
    int n_s, n_u;
    SR_TRY(check_reach_dims(h, &n_s, &n_u));
    SR_DEVICE(h->device);
    const long nss = (long)n_s * n_s, nus = (long)n_u * n_s;
    h->last_chain = 0;
    bool chained = false;
    SR_TRY(try_chain(h, T, H, mode, p0, q0, k_fb0, k_ff, k_fb, a, b, l_mu, l_sigma, c_safety, p_all, q_all, gp_var_all,
                     n_bad, n_s, n_u, s, &chained));
    if (chained) return SR_OK;
    for (long t0 = 0; t0 < T; t0 += h->chunk) {
        const long Tc = std::min(h->chunk, T - t0);
        SR_TRY(prepare_ws(h, Tc));
        sr_ws_lock lock(h);
        for (int i = 0; i < H; ++i) {
            // inputs of step i (gp_reachability.py:195-210)
            const double* p_in; long ldp; const double* q_in; long ldq; const double* kfb_in; long ldkfb;
            if (i == 0) {
                p_in = p0 + t0 * n_s; ldp = n_s;
                q_in = q0 ? q0 + t0 * nss : nullptr; ldq = nss;
                kfb_in = k_fb0 ? k_fb0 + t0 * nus : nullptr; ldkfb = nus;
            } else {
                p_in = p_all + (t0 * H + (i - 1)) * n_s; ldp = (long)H * n_s;
                q_in = q_all + (t0 * H + (i - 1)) * nss; ldq = (long)H * nss;
                kfb_in = k_fb + (t0 * (H - 1) + (i - 1)) * nus; ldkfb = (long)(H - 1) * nus;
            }
            const double* kff_in = k_ff + (t0 * H + i) * n_u;
            const long ldkff = (long)H * n_u;
            const double* jac_su = nullptr;
            SR_TRY(gp_pass_states(h, Tc, p_in, ldp, n_s, kff_in, ldkff, n_u, h->mu, h->var, &jac_su, s));
            if (gp_var_all)
                SR_HIP(hipMemcpy2DAsync(gp_var_all + (t0 * H + i) * n_s, sizeof(double) * H * n_s, h->var,
                                        sizeof(double) * n_s, sizeof(double) * n_s, Tc,
                                        hipMemcpyDeviceToDevice, s));
            sr_ell_args ea;
            ea.T = Tc; ea.n_s = n_s; ea.n_u = n_u;
            ea.p = p_in; ea.ldp = ldp; ea.q = q_in; ea.ldq = ldq;
            ea.k_ff = kff_in; ea.ldkff = ldkff; ea.k_fb = kfb_in; ea.ldkfb = ldkfb;
            ea.mu = h->mu; ea.var = h->var; ea.jac = jac_su;
            ea.a = a; ea.b = b; ea.l_mu = l_mu; ea.l_sigma = l_sigma; ea.c_safety = c_safety;
            ea.p_out = p_all + (t0 * H + i) * n_s; ea.ldpo = (long)H * n_s;
            ea.q_out = q_all + (t0 * H + i) * nss; ea.ldqo = (long)H * nss;
            ea.n_bad = n_bad; ea.mode = mode;
            sr_prof_scope ps(&h->prof, SR_K_ELL, s);
            SR_TRY(sr_launch_ellipsoid(ea, s));
        }
    }
    return SR_OK;
}

extern "C" int sr_multistep_reach(sr_gp_t h, long T, int H, const double* p0, const double* q0,
                                  const double* k_fb0, const double* k_ff, const double* k_fb,
                                  const double* a, const double* b, const double* l_mu,
                                  const double* l_sigma, double c_safety, double* p_all,
                                  double* q_all, int* n_bad, void* stream) {
    SR_CHECK(h != nullptr, SR_EINVAL, "sr_multistep_reach: NULL handle");
    SR_CHECK(h->factorized, SR_ESTATE, "sr_multistep_reach: model not factorized");
    SR_CHECK(T >= 0 && H >= 1, SR_EINVAL, "sr_multistep_reach: T=%ld H=%d", T, H);
    if (T == 0) return SR_OK;
    SR_CHECK(p0 && k_ff && a && b && l_mu && l_sigma && p_all && q_all, SR_EINVAL,
             "sr_multistep_reach: NULL argument");
    SR_CHECK(H == 1 || k_fb != nullptr, SR_EINVAL, "sr_multistep_reach: k_fb required for H > 1");
    SR_CHECK(q0 == nullptr || k_fb0 != nullptr, SR_EINVAL, "sr_multistep_reach: k_fb0 required with q0");
    return multistep_impl(h, T, H, 0, p0, q0, k_fb0, k_ff, k_fb, a, b, l_mu, l_sigma, c_safety, p_all, q_all,
                          nullptr, n_bad, (hipStream_t)stream);
}

extern "C" int sr_multistep_moments(sr_gp_t h, long T, int H, int mode, const double* mu0, const double* k_ff,
                                    const double* k_fb, const double* a, const double* b, double* mu_all,
                                    double* sigma_all, double* gp_var_all, void* stream) {
    SR_CHECK(h != nullptr, SR_EINVAL, "sr_multistep_moments: NULL handle");
    SR_CHECK(h->factorized, SR_ESTATE, "sr_multistep_moments: model not factorized");
    SR_CHECK(T >= 0 && H >= 1 && (mode == 1 || mode == 2), SR_EINVAL, "sr_multistep_moments: T=%ld H=%d mode=%d",
             T, H, mode);
    if (T == 0) return SR_OK;
    SR_CHECK(mu0 && k_ff && a && b && mu_all && sigma_all, SR_EINVAL, "sr_multistep_moments: NULL argument");
    SR_CHECK(H == 1 || k_fb != nullptr, SR_EINVAL, "sr_multistep_moments: k_fb required for H > 1");
    // l_mu / l_sigma are unused by the moment modes: any valid device pointer will do
    return multistep_impl(h, T, H, mode, mu0, nullptr, nullptr, k_ff, k_fb, a, b, a, a, 1.0, mu_all, sigma_all,
                          gp_var_all, nullptr, (hipStream_t)stream);
}

extern "C" int sr_moment_step(int device, long T, int n_s, int n_u, int mode, const double* mu_x,
                              const double* sigma_x, const double* k_ff, const double* k_fb, const double* mu_g,
                              const double* var_g, const double* jac_g, const double* a, const double* b,
                              double* mu_out, double* sigma_out, void* stream) {
    SR_CHECK(T >= 0 && (mode == 1 || mode == 2), SR_EINVAL, "sr_moment_step: T=%ld mode=%d", T, mode);
    if (T == 0) return SR_OK;
    SR_CHECK(mu_x && k_ff && mu_g && var_g && a && b && mu_out && sigma_out, SR_EINVAL, "sr_moment_step: NULL argument");
    SR_CHECK(sigma_x == nullptr || (k_fb != nullptr && (mode == 2 || jac_g != nullptr)), SR_EINVAL,
             "sr_moment_step: k_fb (and jac for the Taylor mode) required with sigma_x");
    SR_DEVICE(device);
    sr_ell_args ea;
    ea.T = T; ea.n_s = n_s; ea.n_u = n_u;
    ea.p = mu_x; ea.ldp = n_s; ea.q = sigma_x; ea.ldq = (long)n_s * n_s;
    ea.k_ff = k_ff; ea.ldkff = n_u; ea.k_fb = k_fb; ea.ldkfb = (long)n_u * n_s;
    ea.mu = mu_g; ea.var = var_g; ea.jac = jac_g ? jac_g : mu_g;     // never dereferenced in mode 2
    ea.a = a; ea.b = b; ea.l_mu = a; ea.l_sigma = a; ea.c_safety = 1.0;
    ea.p_out = mu_out; ea.ldpo = n_s; ea.q_out = sigma_out; ea.ldqo = (long)n_s * n_s;
    ea.n_bad = nullptr; ea.mode = mode;
    return sr_launch_ellipsoid(ea, (hipStream_t)stream);
}

extern "C" int sr_ellipsoid_step(int device, long T, int n_s, int n_u, const double* p,
                                 const double* q, const double* k_ff, const double* k_fb,
                                 const double* mu, const double* var, const double* jac,
                                 const double* a, const double* b, const double* l_mu,
                                 const double* l_sigma, double c_safety, double* p_out,
                                 double* q_out, int* n_bad, void* stream) {
    SR_CHECK(T >= 0, SR_EINVAL, "sr_ellipsoid_step: T=%ld", T);
    if (T == 0) return SR_OK;
    SR_CHECK(p && k_ff && mu && var && a && b && l_mu && l_sigma && p_out && q_out, SR_EINVAL,
             "sr_ellipsoid_step: NULL argument");
    SR_CHECK(q == nullptr || (k_fb != nullptr && jac != nullptr), SR_EINVAL,
             "sr_ellipsoid_step: k_fb and jac required with q");
    SR_DEVICE(device);
    sr_ell_args ea;
    ea.T = T; ea.n_s = n_s; ea.n_u = n_u;
    ea.p = p; ea.ldp = n_s; ea.q = q; ea.ldq = (long)n_s * n_s;
    ea.k_ff = k_ff; ea.ldkff = n_u; ea.k_fb = k_fb; ea.ldkfb = (long)n_u * n_s;
    ea.mu = mu; ea.var = var; ea.jac = jac;
    ea.a = a; ea.b = b; ea.l_mu = l_mu; ea.l_sigma = l_sigma; ea.c_safety = c_safety;
    ea.p_out = p_out; ea.ldpo = n_s; ea.q_out = q_out; ea.ldqo = (long)n_s * n_s;
    ea.n_bad = n_bad; ea.mode = 0;
    return sr_launch_ellipsoid(ea, (hipStream_t)stream);
}

extern "C" int sr_remainder_overapprox(int device, long T, int n_s, int n_u, const double* q,
                                       const double* k_fb, const double* l_mu, const double* l_sigma,
                                       double* u_mu, double* u_sigma, void* stream) {
    SR_CHECK(T >= 0, SR_EINVAL, "sr_remainder_overapprox: T=%ld", T);
    if (T == 0) return SR_OK;
    SR_CHECK(q && k_fb && l_mu && l_sigma && u_mu && u_sigma, SR_EINVAL,
             "sr_remainder_overapprox: NULL argument");
    SR_DEVICE(device);
    return sr_launch_remainder(T, n_s, n_u, q, k_fb, l_mu, l_sigma, u_mu, u_sigma, (hipStream_t)stream);
}

extern "C" int sr_safety_distance(int device, long T, int n_s, int m, const double* p,
                                  const double* q, const double* h_mat, const double* h_vec,
                                  double c_safety, double* d, void* stream) {
    SR_CHECK(T >= 0 && n_s >= 1 && m >= 1, SR_EINVAL, "sr_safety_distance: T=%ld n_s=%d m=%d", T, n_s, m);
    if (T == 0) return SR_OK;
    SR_CHECK(p && q && h_mat && h_vec && d, SR_EINVAL, "sr_safety_distance: NULL argument");
    SR_DEVICE(device);
    return sr_launch_safety(T, n_s, m, p, q, h_mat, h_vec, c_safety, d, (hipStream_t)stream);
}

extern "C" int sr_distance_to_center(int device, long T, int K, int n_s, const double* samples, int per_t,
                                     const double* p, const double* q, double* d, void* stream) {
    SR_CHECK(T >= 0 && K >= 0 && n_s >= 1, SR_EINVAL, "sr_distance_to_center: T=%ld K=%d n_s=%d", T, K, n_s);
    if (T == 0 || K == 0) return SR_OK;
    SR_CHECK(samples && p && q && d, SR_EINVAL, "sr_distance_to_center: NULL argument");
    SR_DEVICE(device);
    return sr_launch_distance(T, K, n_s, samples, per_t, p, q, d, (hipStream_t)stream);
}

extern "C" int sr_gp_sample(int device, long T, int size, int n_out, int n_u, const double* mu,
                            const double* var, const double* eps, double* S, const double* k_fb,
                            const double* k_ff, double* z_next, void* stream) {
    SR_CHECK(T >= 0 && size >= 0 && n_out >= 1 && n_out <= SR_MAX_NS && n_u >= 0, SR_EINVAL,
             "sr_gp_sample: T=%ld size=%d n_out=%d n_u=%d", T, size, n_out, n_u);
    if (T == 0 || size == 0) return SR_OK;
    SR_CHECK(mu && var && eps && S, SR_EINVAL, "sr_gp_sample: NULL argument");
    SR_CHECK(!z_next || n_u == 0 || (k_fb && k_ff), SR_EINVAL, "sr_gp_sample: z_next needs k_fb and k_ff");
    SR_DEVICE(device);
    return sr_launch_sample(T, size, n_out, n_u, mu, var, eps, S, k_fb, k_ff, z_next, (hipStream_t)stream);
}

extern "C" int sr_gp_set_chain(sr_gp_t h, int on) {
    SR_CHECK(h != nullptr, SR_EINVAL, "sr_gp_set_chain: NULL handle");
    h->chain = on != 0;
    return SR_OK;
}

extern "C" int sr_gp_last_chain(sr_gp_t h) { return h ? h->last_chain : 0; }

extern "C" int sr_gp_chain_status(sr_gp_t h, int* timed_out) {
    SR_CHECK(h != nullptr && timed_out != nullptr, SR_EINVAL, "sr_gp_chain_status: NULL argument");
    *timed_out = 0;
    if (h->chain_status_host && *(volatile int*)h->chain_status_host != 0) {
        *timed_out = 1;
        *(volatile int*)h->chain_status_host = 0;
        h->chain = 0;                        // per-step launches from here on (sr_gp_set_chain re-arms)
    }
    return SR_OK;
}

// tests: the next persistent multi-step launches are short of `drop` workgroups, i.e. the last group waits for partners
// that never come -- the deterministic way to reach the time-out path (a CU mask of one or two bits does not do it: the
// driver widens such masks, the chain completed on "1 CU")
extern "C" int sr_test_chain_drop(sr_gp_t h, int drop) {
    SR_CHECK(h != nullptr && drop >= 0, SR_EINVAL, "sr_test_chain_drop: bad argument");
    h->chain_test_drop = drop;
    if (drop == 0 && h->chain_tickets) {
        // a launch that was short of a workgroup leaves its group's counters in a state no real launch can produce
        // (in the field every workgroup runs, however late, and the last one to leave resynchronises the group)
        SR_DEVICE(h->device);
        SR_HIP(device_sync());
        SR_TRY(dev_zero(h->chain_xch, sizeof(sr_xel) * (size_t)SR_CHAIN_XELS));
        SR_TRY(dev_zero(h->chain_tickets, sizeof(unsigned long long) * 2 * SR_CHAIN_GROUPS));
        SR_TRY(dev_zero(h->chain_done, sizeof(unsigned) * SR_CHAIN_GROUPS));
    }
    return SR_OK;
}

