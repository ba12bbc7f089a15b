// sr_capi_append.hip -- sr_gp_append: condition the model on up to 128 additional training points without refactorising.
#include "sr_handle.h"
#include <mutex>
using namespace srh;

// ---------------------------------------------------------------------------------------------
// block row append: condition on m <= 128 additional training points without
// refactorising.  K1 = [K B; B^T C]  =>  U1 = [U U12; 0 U22],  U12 = U^-T B,  U22^T U22 = C - U12^T U12,
//                 U1^-1 = [U^-1  -U^-1 U12 U22^-1; 0  U22^-1].   O(N^2 m) instead of O(N^3).
// ---------------------------------------------------------------------------------------------
__global__ void sr_append_queries_kernel(const double* __restrict__ Znew, double* __restrict__ Xq, int m, int D) {
    // 128 query rows, front padded with copies of the first new point (their columns are never used)
    const int t = blockIdx.x, j = threadIdx.x;
    if (j >= D) return;
    const int pf = SR_NB - m;
    Xq[t * D + j] = Znew[(t < pf ? 0 : t - pf) * D + j];
}

// (info != NULL: also clears the per-output failure word of the append that follows; Zdst != NULL: also copies the m x D
//  new inputs behind the old ones -- m D <= 16 x 12 values, the first workgroup does it)
__global__ void sr_append_y_kernel(const double* __restrict__ yT0, int Np0, int N0, const double* __restrict__ Ynew,
                                   int m, double* __restrict__ yT1, int Np1, int n_out, int* __restrict__ info,
                                   const double* __restrict__ Znew, double* __restrict__ Zdst, int mD) {
    const int i = blockIdx.x * 256 + threadIdx.x, d = blockIdx.y;
    if (info && i == 0) info[d] = 0;
    if (Zdst && d == 0 && blockIdx.x == 0 && (int)threadIdx.x < mD) Zdst[threadIdx.x] = Znew[threadIdx.x];
    if (i >= Np1) return;
    const int off1 = Np1 - (N0 + m), off0 = Np0 - N0;
    double v = 0.0;
    if (i >= off1) {
        const int k = i - off1;
        v = (k < N0) ? yT0[(long)d * Np0 + off0 + k] : Ynew[(long)(k - N0) * n_out + d];
    }
    yT1[(long)d * Np1 + i] = v;
}

// m <= 16 new points: U12 = U^-T B through the streaming kernels of the prediction path (the new points are
// the queries), everything else as matrix-vector shaped passes -- see sr_append.hip.  No big allocation while
// the padded size stays the same (U^-1 ping-pongs between two buffers).
// x_host / y_host (sr_gp_append1_host): ONE new point given in host memory -- it travels in the kernel arguments and the
// status words and log-det partials come back through a pinned block the kernel writes (no copy command either way);
// only where the one-launch route applies, SR_EUNSUPPORTED before anything is touched otherwise.
// The grid kernel's workgroups wait for each other on the device, so two of its launches must never be in flight together
// (each could hold a part of the CUs and wait for the rest): one at a time per process, from the launch to the stream
// synchronisation behind it.  (Launches of OTHER processes on the same device are outside this lock.)
// Co-residency is a matter of ONE device: a lock per device.
static std::mutex g_grid_append_mutex[SR_MAX_DEVICES];
static std::mutex& grid_append_mutex(int device) { return g_grid_append_mutex[(unsigned)device % SR_MAX_DEVICES]; }

// 0: the general route (a chain of launches); 1: one launch, one workgroup per output and share of the rows (small models);
// 2: one launch of a grid of workgroups with two device-wide barriers (sr_append1_grid_kernel)
static int append1_route(const sr_gp* h, int m, bool ignore_hold = false) {
    if (m != 1 || h->small_path == 0) return 0;
    const int Np1 = (int)round_up(h->N + m, SR_NB);
    if (h->Np <= SR_APPEND1_MAX_NP0 && Np1 <= SR_APPEND1_MAX_NP0 + SR_NB) return 1;
    static const bool no_grid = sr_lab_on("SR_APPEND_NO_GRID");       // (lab build: A/B measurements)
    if (h->Np <= SR_APPEND1G_MAX_NP0 && h->n_out <= SR_APPEND1_MAX_OUT && !no_grid && (h->grid_hold == 0 || ignore_hold)) return 2;
    return 0;
}
// a grid launch gave its barrier up: leave the route alone for a while (the CUs it wants are held by someone else -- every
// further attempt costs its ~5 ms time-out again), longer after every abort in a row
static void grid_append_aborted(sr_gp* h) {
    ++h->grid_aborts;
    h->grid_hold = 16 << std::min(h->grid_aborts_row, 10);
    ++h->grid_aborts_row;
}
static bool append1_fused(const sr_gp* h, int m) { return append1_route(h, m) != 0; }
// workgroups per output of the grid route: all of them must be resident at once (they wait for each other) and one fits a
// CU, so the grid leaves an eighth of the CUs to whatever else is resident on the device (the single-query servers of other
// models: a few workgroups); and no more workgroups than there is work for
static int append1_grid_w(sr_gp* h) {
    if (h->ncu == 0) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device) != hipSuccess) cus = 64;
        h->ncu = cus;
    }
    const int useful = (h->Np + SR_NB + 15) / 16;             // (16 rows of the new factor per workgroup and pass)
    return std::max(1, std::min(std::min(SR_APPEND1G_MAX_W, useful), (h->ncu - h->ncu / 8) / h->n_out));
}

// The grid did not assemble (status SR_APPG_ABORTED in every output's word): nothing of the model was written; the barrier
// words start from zero again and the caller takes the route of separate launches.
static int grid_append_reset(sr_gp* h, hipStream_t s) {
    SR_HIP(hipMemsetAsync(h->appg_cnt, 0, sizeof(double) * SR_APPEND1_MAX_OUT, s));
    SR_HIP(hipStreamSynchronize(s));
    h->appg_base = 0; h->appg_q = 0;
    return SR_OK;
}

// ONE new point IN PLACE (the padded size stays: a front-padding row is left): the grid kernel computes the new column
// and writes it, the new diagonal entry, alpha and the new target into the memory the model already lives in, and the
// model's buffers become views one step further into their allocations (sr_gp::slide; sr_append1_grid_kernel says why
// that is the appended model).  Nothing is moved, nothing is allocated; a pivot that fails leaves every byte as it was.
static int append1_slide(sr_gp* h, const double* Znew, const double* Ynew, hipStream_t s, int* info, const double* x_host,
                         const double* y_host, bool* aborted) {
    *aborted = false;
    const int N0 = h->N, Np0 = h->Np, D = h->D, n_out = h->n_out;
    // what the in-place view rests on: a front-padding row to give up, slack left behind the last output, buffers that carry it
    SR_CHECK(Np0 - N0 >= 1 && h->slide >= 0 && h->slide < SR_SLIDE_STEPS - 1 && h->slack_ok, SR_ESTATE,
             "append in place: N=%d Np=%d slide=%d slack=%d", N0, Np0, h->slide, h->slack_ok);
    const bool host_new = x_host != nullptr;
    const int W = append1_grid_w(h), nld = n_out * W;
    if (host_new && !h->app_pin) {
        void *p = nullptr, *pd = nullptr;
        SR_HIP(hipHostMalloc(&p, sizeof(double) * (SR_APPEND1_MAX_OUT * SR_APPEND1G_MAX_W + SR_APPEND1_MAX_OUT), hipHostMallocMapped));
        if (hipHostGetDevicePointer(&pd, p, 0) != hipSuccess) {
            (void)hipHostFree(p);
            (void)hipGetLastError();
            sr_set_error("sr_gp_append1_host: pinned host memory is not device-visible here");
            return SR_EUNSUPPORTED;
        }
        h->app_pin = p; h->app_pin_dev = (double*)pd;
    }
    const size_t o_ld = 0, o_info = o_ld + (size_t)nld, o_grid = o_info + (size_t)n_out,
                 need = o_grid + (size_t)sr_append1_grid_ws(Np0, n_out);
    if (h->app_cap < need) {
        (void)device_sync();
        dev_free(h->app_ws);
        h->app_ws = nullptr; h->app_cap = 0;
        SR_TRY(dev_alloc(&h->app_ws, need));
        h->app_cap = need;
    }
    double* ws = h->app_ws;
    if (!h->appg_cnt) {
        SR_TRY(dev_alloc(&h->appg_cnt, (size_t)SR_APPEND1_MAX_OUT));          // (unsigned counters in a block of doubles)
        SR_HIP(hipMemsetAsync(h->appg_cnt, 0, sizeof(double) * SR_APPEND1_MAX_OUT, s));
        h->appg_base = 0; h->appg_q = 0;
    }
    std::unique_lock<std::mutex> grid_lock(grid_append_mutex(h->device));
    SR_TRY(sr_launch_append1_grid(h->Wt, h->alpha, h->yT, h->Z, h->ls, h->sf2, h->noise, h->general ? h->kp : nullptr, Znew, Ynew,
                                  h->Wt + Np0 + 1, h->alpha + 1, h->yT + 1, h->Z + (size_t)N0 * D,
                                  host_new ? h->app_pin_dev : ws + o_ld,
                                  host_new ? reinterpret_cast<int*>(h->app_pin_dev + nld) : reinterpret_cast<int*>(ws + o_info), N0,
                                  Np0, Np0, D, n_out, W, ws + o_grid, reinterpret_cast<unsigned*>(h->appg_cnt), h->appg_base, h->appg_q, s,
                                  x_host, y_host, 1));
    h->appg_base += 2u * (unsigned)W * (unsigned)n_out; h->appg_q += 2u;
    std::vector<double> back(nld + (n_out + 1) / 2, 0.0);         // the log-det partial sums, then n_out ints
    if (host_new) {
        SR_HIP(hipStreamSynchronize(s));
        memcpy(back.data(), h->app_pin, sizeof(double) * nld + sizeof(int) * n_out);
    } else {
        SR_HIP(hipMemcpyAsync(back.data(), ws + o_ld, sizeof(double) * nld + sizeof(int) * n_out, hipMemcpyDeviceToHost, s));
        SR_HIP(hipStreamSynchronize(s));
    }
    std::vector<int> info_h(n_out, 0);
    memcpy(info_h.data(), back.data() + nld, sizeof(int) * n_out);
    if (info_h[0] == SR_APPG_ABORTED) {                           // the grid did not assemble: nothing written
        SR_TRY(grid_append_reset(h, s));
        grid_append_aborted(h);
        *aborted = true;
        sr_set_error("one-point append: the grid of workgroups did not become resident (abort %ld of this handle)", h->grid_aborts);
        return SR_EBUSY;
    }
    h->grid_aborts_row = 0;
    grid_lock.unlock();
    int bad = 0;
    for (int d = 0; d < n_out; ++d) {
        if (info) info[d] = info_h[d];
        if (info_h[d] != 0 && !bad) bad = d + 1;
    }
    if (bad) {               // (the kernel wrote nothing: every workgroup has seen the pivots of all outputs)
        sr_set_error("sr_gp_append: Schur complement not positive definite (output %d, point %d)", bad - 1, info_h[bad - 1]);
        return SR_ENOTPD;
    }
    std::vector<double> ld(n_out, 0.0);
    for (int d = 0; d < n_out; ++d)
        for (int y = 0; y < W; ++y) ld[d] += back[(size_t)d * W + y];
    h->Wt += Np0 + 1; h->alpha += 1; h->yT += 1;
    ++h->slide;
    h->N = N0 + 1;
    h->logdet_host = ld; h->logdet_valid = 1;
    return SR_OK;
}

static int append_small(sr_gp* h, const double* Znew, const double* Ynew, int m, hipStream_t s, int* info,
                        const double* x_host = nullptr, const double* y_host = nullptr, bool no_grid = false) {
    {
        // one point, the padded size stays, the buffers carry their slack: in place
        static const bool no_slide = sr_lab_on("SR_APPEND_NO_SLIDE");      // (lab build: A/B measurements)
        const int np1 = (int)round_up(h->N + m, SR_NB);
        const bool held = m == 1 && (h->grid_hold > 0 || h->slide_hold > 0);
        if (m == 1 && h->slide_hold > 0) --h->slide_hold;       // (a big batch met an odd slide: see tile_route_alignment)
        if (!no_grid && !held && append1_route(h, m) == 2 && np1 == h->Np && h->slack_ok && h->slide < SR_SLIDE_STEPS - 1 &&
            h->N + 1 <= h->z_cap && h->n_out <= SR_APPEND1_MAX_OUT && !no_slide) {
            bool aborted = false;
            const int rc = append1_slide(h, Znew, Ynew, s, info, x_host, y_host, &aborted);
            if (!aborted) return rc;
            if (x_host) return rc;               // (the routes of separate launches want the point in device memory: SR_EBUSY)
            no_grid = true;                      // the grid did not assemble: separate launches
        }
        if (m == 1 && h->grid_hold > 0) { --h->grid_hold; no_grid = true; }
        SR_TRY(unslide(h));                  // everything below works on plain buffers
    }
    const int N0 = h->N, Np0 = h->Np, off0 = Np0 - N0, D = h->D, n_out = h->n_out;
    const int N1 = N0 + m, Np1 = (int)round_up(N1, SR_NB), off1 = Np1 - N1, pf = SR_NB - m;
    const bool host_new = x_host != nullptr;
    if (host_new) {
        if (!append1_fused(h, m) || n_out > SR_APPEND1_MAX_OUT) {
            sr_set_error("sr_gp_append1_host: no one-launch append for this model (Np=%d, n_out=%d)", Np0, n_out);
            return SR_EUNSUPPORTED;
        }
        if (!h->app_pin) {
            void *p = nullptr, *pd = nullptr;
            SR_HIP(hipHostMalloc(&p, sizeof(double) * (SR_APPEND1_MAX_OUT * SR_APPEND1G_MAX_W + SR_APPEND1_MAX_OUT), hipHostMallocMapped));
            if (hipHostGetDevicePointer(&pd, p, 0) != hipSuccess) {
                (void)hipHostFree(p);
                (void)hipGetLastError();
                sr_set_error("sr_gp_append1_host: pinned host memory is not device-visible here");
                return SR_EUNSUPPORTED;
            }
            h->app_pin = p; h->app_pin_dev = (double*)pd;
        }
    }
    const size_t NN0 = (size_t)Np0 * Np0, NN1 = (size_t)Np1 * Np1, BB = (size_t)SR_NB * SR_NB;
    int route = append1_route(h, m);
    if (no_grid && route == 2) route = 0;
    const bool fused1 = route != 0;
    const int nwy = route == 2 ? append1_grid_w(h) : SR_APPEND1_WGS;
    const int nld = n_out * nwy;                                  // log-det partial sums the kernels leave (the status words follow them)
    // scratch layout
    // (Xt, Y2, G, S, S^-1 once per output: every step below is ONE launch over all outputs)
    const size_t s_xt = (size_t)SR_SMALL_T * Np0, s_y2 = (size_t)Np0 * SR_NB;
    const size_t o_u12 = 0, o_xt = o_u12 + (size_t)n_out * SR_SMALL_T * Np0, o_y2 = o_xt + n_out * s_xt,
                 o_g = o_y2 + n_out * s_y2, o_sb = o_g + n_out * BB, o_inv = o_sb + n_out * BB,
                 o_ld = o_inv + n_out * BB, o_info = o_ld + (size_t)nld, o_grid = o_info + (size_t)n_out,
                 need = o_grid + (route == 2 ? (size_t)sr_append1_grid_ws(Np0, n_out) : 0);
    if (h->app_cap < need) {
        (void)device_sync();
        dev_free(h->app_ws);
        h->app_ws = nullptr; h->app_cap = 0;
        SR_TRY(dev_alloc(&h->app_ws, need));
        h->app_cap = need;
    }
    double* ws = h->app_ws;
    double *U12t = ws + o_u12, *Xt = ws + o_xt, *Y2 = ws + o_y2, *G = ws + o_g, *Sb = ws + o_sb, *invS = ws + o_inv;
    int* info_dev = reinterpret_cast<int*>(ws + o_info);
    double *Z1 = nullptr, *yT1 = nullptr, *alpha1 = nullptr, *Wt1 = nullptr;
    const bool reuse_alt = (Np1 == Np0) && h->Wt_alt && h->wt_alt_cap >= (size_t)n_out * NN1;
    // while the padded size stays, nothing is allocated: the new points go behind the old ones in Z (room for Np
    // points), yT / alpha / U^-1 are written into the buffers the state before the previous append lived in
    const bool z_inplace = N1 <= h->z_cap;
    const bool vec_alt = (Np1 == Np0) && h->yT_alt && h->alpha_alt && h->vec_alt_np == Np1;
    int rc = SR_OK;
    auto drop_new = [&]() {
        if (!z_inplace) dev_free(Z1);
        if (!vec_alt) { dev_free(yT1); dev_free(alpha1); }
        if (!reuse_alt) dev_free(Wt1);
        else h->wt_alt_off = -1;       // the spare factor buffer may hold a half-written state now
    };
#define SR_A(expr) do { rc = (expr); if (rc != SR_OK) { drop_new(); return rc; } } while (0)
#define SR_AH(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
        sr_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); drop_new(); return SR_EHIP; } } while (0)
    if (z_inplace) Z1 = h->Z;
    else SR_A(dev_alloc(&Z1, (size_t)Np1 * D));
    if (vec_alt) { yT1 = h->yT_alt; alpha1 = h->alpha_alt; }
    else {
        SR_A(dev_alloc(&yT1, vec_doubles(n_out, Np1)));
        SR_A(dev_alloc(&alpha1, vec_doubles(n_out, Np1)));
        SR_AH(hipMemsetAsync(yT1 + (size_t)n_out * Np1, 0, sizeof(double) * SR_SLIDE_STEPS, s));         // (the slack: sr_gp::slide)
        SR_AH(hipMemsetAsync(alpha1 + (size_t)n_out * Np1, 0, sizeof(double) * SR_SLIDE_STEPS, s));
    }
    if (reuse_alt) Wt1 = h->Wt_alt;
    else {
        SR_A(dev_alloc(&Wt1, wt_doubles(n_out, Np1)));
        SR_AH(hipMemsetAsync(Wt1 + (size_t)n_out * NN1, 0, sizeof(double) * (wt_doubles(n_out, Np1) - (size_t)n_out * NN1), s));
    }
    if (!z_inplace) SR_AH(hipMemcpyAsync(Z1, h->Z, sizeof(double) * N0 * D, hipMemcpyDeviceToDevice, s));
    // (in place: rows N0 .. N1-1 of Z are not read by anything below -- the model keeps N = N0 until the commit)
    // one point on a small model: the whole append is ONE launch (sr_append1_small_kernel)
    std::unique_lock<std::mutex> grid_lock(grid_append_mutex(h->device), std::defer_lock);
    if (route == 2) {
        grid_lock.lock();
        // the target holds zeros below the diagonal and its identity padding, or gets them now (as on the general route)
        if (!(reuse_alt && h->wt_alt_off >= off1)) {
            SR_AH(hipMemsetAsync(Wt1, 0, (size_t)n_out * NN1 * sizeof(double), s));
            SR_A(sr_launch_eye_front(Wt1, Np1, off1, s, n_out));
        }
        if (!h->appg_cnt) {
            SR_A(dev_alloc(&h->appg_cnt, (size_t)SR_APPEND1_MAX_OUT));        // (unsigned counters in a block of doubles)
            SR_AH(hipMemsetAsync(h->appg_cnt, 0, sizeof(double) * SR_APPEND1_MAX_OUT, s));
            h->appg_base = 0; h->appg_q = 0;
        }
        SR_A(sr_launch_append1_grid(h->Wt, h->alpha, h->yT, h->Z, h->ls, h->sf2, h->noise, h->general ? h->kp : nullptr,
                                    Znew, Ynew, Wt1, alpha1, yT1, Z1 + (size_t)N0 * D, host_new ? h->app_pin_dev : ws + o_ld,
                                    host_new ? reinterpret_cast<int*>(h->app_pin_dev + nld) : info_dev, N0, Np0, Np1, D, n_out, nwy,
                                    ws + o_grid, reinterpret_cast<unsigned*>(h->appg_cnt), h->appg_base, h->appg_q, s, x_host, y_host));
        h->appg_base += 2u * (unsigned)nwy * (unsigned)n_out; h->appg_q += 2u;                       // (every workgroup arrives twice, whatever the launch finds)
    } else if (fused1) {
        SR_A(sr_launch_append1_small(h->Wt, h->alpha, h->yT, h->Z, h->ls, h->sf2, h->noise, h->general ? h->kp : nullptr,
                                     Znew, Ynew, Wt1, alpha1, yT1,
                                     Z1 + (size_t)N0 * D, host_new ? h->app_pin_dev : ws + o_ld,
                                     host_new ? reinterpret_cast<int*>(h->app_pin_dev + nld) : info_dev, N0, Np0, Np1, D, n_out, s,
                                     x_host, y_host));
    } else {
    static_assert(SR_SMALL_T * SR_MAX_D <= 256, "the first workgroup of sr_append_y_kernel copies the new inputs");
    hipLaunchKernelGGL(sr_append_y_kernel, dim3((Np1 + 255) / 256, n_out), dim3(256), 0, s, h->yT, Np0, N0, Ynew, m,
                       yT1, Np1, n_out, info_dev, Znew, Z1 + (size_t)N0 * D, m * D);
    SR_AH(hipGetLastError());
    // B = K(Z_old, Z_new) with the new points as queries, then U12 = U^-T B by streaming U^-1 once
    const long Tp = srt::BN;
    const int nsplit = pick_nsplit(h, Tp);
    SR_A(ensure_ws(h, Tp, nsplit));
    sr_kstar_args ka;
    ka.Z = h->Z; ka.alpha = h->alpha; ka.ls = h->ls; ka.sf2 = h->sf2;
    ka.kp = h->general ? h->kp : nullptr; ka.kxx = h->kxx;
    ka.xa = Znew; ka.lda = D; ka.na = D; ka.xb = nullptr; ka.ldb = 0; ka.nb = 0;
    ka.Ks = h->Ks; ka.mu_part = h->mu_part; ka.jac_part = h->jac_part;
    ka.N = N0; ka.Np = Np0; ka.D = D; ka.n_out = n_out; ka.nsplit = nsplit; ka.T = m; ka.Tp = Tp;
    SR_A(sr_launch_kstar(ka, s));
    if (!h->small_vp) SR_A(dev_alloc(&h->small_vp, (size_t)sr_var_small_ws(Np0, n_out)));
    SR_A(sr_launch_var_small(h->Wt, h->Ks, h->small_vp, h->var_part, N0, Np0, Tp, n_out, m, s, 0, false));   // (no norms wanted)
    SR_A(sr_launch_var_small_gather_all(h->small_vp, U12t, Np0, n_out, m, s));
    // All outputs in every launch (round 3; before: a chain of 8 dependent launches PER OUTPUT -- 25 dispatches for one new
    // point on a two-output model, 125 us on the host whatever the model size up to N ~ 1000):
    // G = U12^T U12 (only its m x m corner is ever read: no zero fill), C = K(Z_new, Z_new) + noise, the corner kernel
    // factors and inverts C - G
    SR_A(sr_launch_append_small(U12t, h->Wt, Np0, m, 0, G, nullptr, nullptr, nullptr, s, n_out));
    if (h->general) SR_A(sr_launch_gram_general(Znew, h->kp, 0.0, h->noise, Sb, m, SR_NB, D, s, n_out, (long)BB));
    else SR_A(sr_launch_gram(Znew, h->ls, 0.0, 0.0, h->sf2, h->noise, Sb, m, SR_NB, D, s, n_out, (long)BB));
    SR_A(sr_launch_potrf_corner16(Sb, SR_NB, invS, SR_NB, info_dev, s, G, pf, n_out, (long)BB, (long)BB));   // invS = U22^-1
    // Y2 = -U^-1 U12 U22^-1 and the move of the old factor to its new place in one pass over it; a buffer that
    // did not hold an earlier state of this model is zeroed first (lower triangle, identity padding)
    if (!(reuse_alt && h->wt_alt_off >= off1)) {
        SR_AH(hipMemsetAsync(Wt1, 0, (size_t)n_out * NN1 * sizeof(double), s));
        SR_A(sr_launch_eye_front(Wt1, Np1, off1, s, n_out));
    }
    SR_A(sr_launch_append_move(h->Wt, Np0, off0, N0, U12t, invS, m, Xt, Y2, Wt1, Np1, off1, s, n_out, (long)s_xt, (long)s_y2));
    // alpha1 = [alpha0 + Y2 v2 ; U22^-1 v2],  v2 = U22^-T (y_new - mu_old(z_new)): no pass over U^-1
    SR_A(sr_launch_append_alpha(h->alpha, Np0, N0, Y2, invS, h->mu_part, nsplit, n_out, 0, Tp, Ynew, m, alpha1, Np1, s, 0,
                                n_out, (long)s_y2));
    // log det of the grown model beside the status words: ONE read-back for both (the reference's exploration loop asks
    // for the information gain after every appended point)
    SR_A(sr_launch_logdet(Wt1, Np1, n_out, ws + o_ld, s));
    }
    std::vector<double> back(nld + (n_out + 1) / 2, 0.0);         // the log dets, then n_out ints
    if (host_new) {
        SR_AH(hipStreamSynchronize(s));                           // (the kernel wrote both into the pinned block)
        memcpy(back.data(), h->app_pin, sizeof(double) * nld + sizeof(int) * n_out);
    } else {
        SR_AH(hipMemcpyAsync(back.data(), ws + o_ld, sizeof(double) * nld + sizeof(int) * n_out, hipMemcpyDeviceToHost, s));
        SR_AH(hipStreamSynchronize(s));
    }
    std::vector<int> info_h(n_out, 0);
    memcpy(info_h.data(), back.data() + nld, sizeof(int) * n_out);
    if (route == 2 && info_h[0] == SR_APPG_ABORTED) {            // the grid did not assemble: nothing of the model was written
        const int rr = grid_append_reset(h, s);
        grid_lock.unlock();
        if (reuse_alt) h->wt_alt_off = -1;
        drop_new();
        if (rr != SR_OK) return rr;
        grid_append_aborted(h);
        if (host_new) { sr_set_error("sr_gp_append1_host: the grid of the one-launch append did not assemble"); return SR_EBUSY; }
        return append_small(h, Znew, Ynew, m, s, info, nullptr, nullptr, true);
    }
    if (route == 2) h->grid_aborts_row = 0;
    if (grid_lock.owns_lock()) grid_lock.unlock();
    for (int d = 0; d < n_out; ++d) {
        double t = back[fused1 ? d * nwy : d];
        for (int y = 1; fused1 && y < nwy; ++y) t += back[d * nwy + y];
        back[d] = t;                                              // (d <= d * nwy: nothing unread is overwritten)
    }
    h->logdet_valid = 0;
#undef SR_A
#undef SR_AH
    int bad = 0;
    for (int d = 0; d < n_out; ++d) {
        // the corner kernel reports the pivot inside its front-padded 128-block; the one-launch append reports the
        // training index (N0 + 1) itself
        if (info_h[d] > 0 && !fused1) info_h[d] = N0 + std::max(1, info_h[d] - pf);
        if (info) info[d] = info_h[d];
        if (info_h[d] != 0 && !bad) bad = d + 1;
    }
    if (bad) {
        if (reuse_alt) h->wt_alt_off = -1;     // the spare factor buffer holds a half-written state now
        drop_new();
        sr_set_error("sr_gp_append: Schur complement not positive definite (output %d, point %d)", bad - 1, info_h[bad - 1]);
        return SR_ENOTPD;
    }
    double* old_wt = h->Wt;
    double *old_yT = h->yT, *old_alpha = h->alpha;
    if (!z_inplace) { dev_free(h->Z); h->Z = Z1; h->z_cap = Np1; }
    h->yT = yT1; h->alpha = alpha1; h->Wt = Wt1;
    h->N = N1;
    h->logdet_host.assign(back.begin(), back.begin() + n_out); h->logdet_valid = 1;
    if (Np1 == Np0) {
        if (!vec_alt) { dev_free(h->yT_alt); dev_free(h->alpha_alt); }
        h->yT_alt = old_yT; h->alpha_alt = old_alpha; h->vec_alt_np = Np0;
        // keep the previous buffer for the next append (bounded: not for huge factors)
        if (!reuse_alt) dev_free(h->Wt_alt);
        if ((size_t)n_out * NN0 * sizeof(double) <= SR_FACT_PAR_BYTES * 2) { h->Wt_alt = old_wt; h->wt_alt_cap = (size_t)n_out * NN0; h->wt_alt_off = off0; }
        else { dev_free(old_wt); h->Wt_alt = nullptr; h->wt_alt_cap = 0; h->wt_alt_off = -1; }
    } else {
        dev_free(old_wt);
        dev_free(old_yT); dev_free(old_alpha);
        dev_free(h->yT_alt); dev_free(h->alpha_alt); h->yT_alt = h->alpha_alt = nullptr; h->vec_alt_np = 0;
        dev_free(h->Wt_alt); h->Wt_alt = nullptr; h->wt_alt_cap = 0; h->wt_alt_off = -1;
        h->Np = Np1;
        free_ws(h);
        dev_free(h->lin_v); dev_free(h->lin_g); dev_free(h->small_vp); dev_free(h->splitk_vt); dev_free(h->splitk_part);
        h->lin_v = h->lin_g = h->small_vp = h->splitk_vt = h->splitk_part = nullptr;
        h->splitk_cap = 0;
        dev_free(h->stream_vp); dev_free(h->stream_tickets);
        h->stream_vp = nullptr; h->stream_vp_cap = 0; h->stream_tickets = nullptr;
        dev_free(h->fact_ws); h->fact_ws = nullptr; h->fact_cap = 0;
        dev_free(h->app_ws); h->app_ws = nullptr; h->app_cap = 0;
    }
    return SR_OK;
}

extern "C" int sr_gp_grid_append_aborts(sr_gp_t h, long* n) {
    SR_CHECK(h && n, SR_EINVAL, "sr_gp_grid_append_aborts: NULL argument");
    *n = h->grid_aborts;
    return SR_OK;
}

extern "C" int sr_gp_append1_host(sr_gp_t h, const double* x_host, const double* y_host, void* stream, int* info) {
    SR_CHECK(h != nullptr && x_host && y_host, SR_EINVAL, "sr_gp_append1_host: NULL argument");
    SR_CHECK(h->factorized, SR_ESTATE, "sr_gp_append1_host: model not factorized");
    SR_DEVICE(h->device);
    if (!append1_fused(h, 1) || h->n_out > SR_APPEND1_MAX_OUT) {
        if (append1_route(h, 1, true) == 2 && h->n_out <= SR_APPEND1_MAX_OUT) {
            sr_set_error("sr_gp_append1_host: the grid route is resting after an abort (%d appends to go)", h->grid_hold);
            return SR_EBUSY;                 // (sr_gp_append counts the rest down)
        }
        sr_set_error("sr_gp_append1_host: no one-launch append for this model (Np=%d, n_out=%d)", h->Np, h->n_out);
        return SR_EUNSUPPORTED;
    }
    SR_TRY(server_quiesce(h));
    return append_small(h, nullptr, nullptr, 1, (hipStream_t)stream, info, x_host, y_host);
}

extern "C" int sr_gp_append(sr_gp_t h, const double* Znew, const double* Ynew, int m, void* stream, int* info) {
    SR_CHECK(h != nullptr && Znew && Ynew, SR_EINVAL, "sr_gp_append: NULL argument");
    SR_CHECK(h->factorized, SR_ESTATE, "sr_gp_append: model not factorized");
    SR_CHECK(m >= 1 && m <= SR_NB, SR_EINVAL, "sr_gp_append: m=%d outside 1..%d (append in several calls)", m, SR_NB);
    hipStream_t s = (hipStream_t)stream;
    SR_DEVICE(h->device);
    SR_TRY(server_quiesce(h));            // (it stays armed: the next single query launches it on the grown model)
    if (m <= SR_SMALL_T) return append_small(h, Znew, Ynew, m, s, info);
    SR_TRY(unslide(h));
    // 17 .. 128 new points: the same algebra on the MFMA tile (64 x 64 workgroup tiles: the products are 128 columns wide).
    // Scratch lives with the handle, U^-1 ping-pongs between two buffers while the padded size stays, alpha is updated
    // from the old model's mean at the new points like in the few-points route -- no allocation of the factor's size, no
    // pass over the new U^-1 (first version: 12 hipMallocs, two 210 MB transposes and two triangular mat-vecs per
    // output: 7 ms at N = 5000).
    const int N0 = h->N, Np0 = h->Np, off0 = Np0 - N0, D = h->D, n_out = h->n_out;
    const int N1 = N0 + m, Np1 = (int)round_up(N1, SR_NB), off1 = Np1 - N1;
    const int pf = SR_NB - m;
    const size_t NN0 = (size_t)Np0 * Np0, NN1 = (size_t)Np1 * Np1, BB = (size_t)SR_NB * SR_NB, PB = (size_t)Np0 * SR_NB;
    constexpr int APP_KS = 512;          // K-slice of the thin products (sr_launch_gemm_tn_splitk)
    const size_t o_xq = 0, o_ks = o_xq + (size_t)SR_NB * D, o_u12 = o_ks + (size_t)n_out * PB, o_u12t = o_u12 + PB,
                 o_x = o_u12t + PB, o_y2 = o_x + PB, o_g = o_y2 + PB, o_sb = o_g + BB, o_inv = o_sb + BB, o_wdm = o_inv + BB,
                 o_wtr = o_wdm + BB, o_part = o_wtr + NN0, o_info = o_part + (size_t)((Np0 + APP_KS - 1) / APP_KS) * PB,
                 need = o_info + (size_t)n_out;
    if (h->app_cap < need) {
        (void)device_sync();
        dev_free(h->app_ws);
        h->app_ws = nullptr; h->app_cap = 0;
        SR_TRY(dev_alloc(&h->app_ws, need));
        h->app_cap = need;
    }
    double* ws = h->app_ws;
    double *Xq = ws + o_xq, *Ks = ws + o_ks, *U12 = ws + o_u12, *U12t = ws + o_u12t, *X = ws + o_x, *Y2 = ws + o_y2,
           *G = ws + o_g, *Sb = ws + o_sb, *invS = ws + o_inv, *wdm = ws + o_wdm, *Wtr = ws + o_wtr, *part = ws + o_part;
    int* info_dev = reinterpret_cast<int*>(ws + o_info);
    double *Z1 = nullptr, *yT1 = nullptr, *alpha1 = nullptr, *Wt1 = nullptr;                 // new persistent state
    const bool reuse_alt = (Np1 == Np0) && h->Wt_alt && h->wt_alt_cap >= (size_t)n_out * NN1;
    std::vector<double> sf2(n_out), noise(n_out);
    int rc = SR_OK;
    auto drop_new = [&]() {
        dev_free(Z1); dev_free(yT1); dev_free(alpha1);
        if (!reuse_alt) dev_free(Wt1);
        else h->wt_alt_off = -1;       // the spare factor buffer may hold a half-written state now
    };
#define SR_A(expr) do { rc = (expr); if (rc != SR_OK) { drop_new(); return rc; } } while (0)
#define SR_AH(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
        sr_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); drop_new(); return SR_EHIP; } } while (0)
    SR_A(dev_alloc(&Z1, (size_t)N1 * D));
    SR_A(dev_alloc(&yT1, vec_doubles(n_out, Np1)));
    SR_A(dev_alloc(&alpha1, vec_doubles(n_out, Np1)));
    SR_AH(hipMemsetAsync(yT1 + (size_t)n_out * Np1, 0, sizeof(double) * SR_SLIDE_STEPS, s));             // (the slack: sr_gp::slide)
    SR_AH(hipMemsetAsync(alpha1 + (size_t)n_out * Np1, 0, sizeof(double) * SR_SLIDE_STEPS, s));
    if (reuse_alt) Wt1 = h->Wt_alt;
    else {
        SR_A(dev_alloc(&Wt1, wt_doubles(n_out, Np1)));
        SR_AH(hipMemsetAsync(Wt1 + (size_t)n_out * NN1, 0, sizeof(double) * (wt_doubles(n_out, Np1) - (size_t)n_out * NN1), s));
    }
    SR_AH(hipMemsetAsync(info_dev, 0, sizeof(int) * n_out, s));
    SR_AH(hipMemcpyAsync(sf2.data(), h->sf2, sizeof(double) * n_out, hipMemcpyDeviceToHost, s));
    SR_AH(hipMemcpyAsync(noise.data(), h->noise, sizeof(double) * n_out, hipMemcpyDeviceToHost, s));
    SR_AH(hipMemcpyAsync(Z1, h->Z, sizeof(double) * N0 * D, hipMemcpyDeviceToDevice, s));
    SR_AH(hipMemcpyAsync(Z1 + (size_t)N0 * D, Znew, sizeof(double) * m * D, hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(sr_append_y_kernel, dim3((Np1 + 255) / 256, n_out), dim3(256), 0, s, h->yT, Np0, N0, Ynew, m,
                       yT1, Np1, n_out, (int*)nullptr, (const double*)nullptr, (double*)nullptr, 0);
    hipLaunchKernelGGL(sr_append_queries_kernel, dim3(SR_NB), dim3(64), 0, s, Znew, Xq, m, D);
    SR_AH(hipGetLastError());
    SR_AH(hipStreamSynchronize(s));
    // B = K(Z_old, Z_new): the prediction kernel with the new points as queries (old padded row indexing); its
    // mean partial sums (the old model's mean at the new points) feed the alpha update below
    const int nsplit = pick_nsplit(h, SR_NB);
    SR_A(ensure_ws(h, SR_NB, nsplit));
    sr_kstar_args ka;
    ka.Z = h->Z; ka.alpha = h->alpha; ka.ls = h->ls; ka.sf2 = h->sf2;
    ka.kp = h->general ? h->kp : nullptr; ka.kxx = h->kxx;
    ka.xa = Xq; ka.lda = D; ka.na = D; ka.xb = nullptr; ka.ldb = 0; ka.nb = 0;
    ka.Ks = Ks; ka.mu_part = h->mu_part; ka.jac_part = h->jac_part;
    ka.N = N0; ka.Np = Np0; ka.D = D; ka.n_out = n_out; ka.nsplit = nsplit; ka.T = SR_NB; ka.Tp = SR_NB;
    SR_A(sr_launch_kstar(ka, s));
    for (int d = 0; d < n_out; ++d) {
        const double* Wt0 = h->Wt + (size_t)d * NN0;
        // U12 = U^-T B  (A = U^-1 k-major, upper block triangular)
        // (thin products -- 128 columns, K up to Np -- in K-slices: 227 -> ~50 us, G = U12^T U12 209 -> ~15 us at N = 5000)
        SR_A(sr_launch_gemm_tn_splitk(Wt0, Np0, Ks + (size_t)d * PB, SR_NB, U12, Np0, SR_NB, Np0, APP_KS, 1.0, 3, part, s));
        SR_A(sr_launch_gemm_tn_splitk(U12, SR_NB, U12, SR_NB, G, SR_NB, SR_NB, Np0, APP_KS, 1.0, 0, part, s));
        // S = C - U12^T U12 on the real (front padded) block, C = k(Znew, Znew) + noise I
        if (h->general) SR_A(sr_launch_gram_general(Znew, h->kp + (size_t)d * SR_KP(D), noise[d], nullptr, Sb, m, SR_NB, D, s));
        else SR_A(sr_launch_gram(Znew, h->ls + (size_t)d * D, sf2[d], noise[d], nullptr, nullptr, Sb, m, SR_NB, D, s));
        SR_A(sr_launch_sub_block(Sb, G, pf, s));
        SR_A(sr_launch_potrf_diag(Sb, SR_NB, invS, wdm, SR_NB, 0, info_dev + d, s));       // invS = U22^-1
        // X = U12 U22^-1
        SR_A(sr_launch_transpose_rect(U12, SR_NB, U12t, Np0, Np0, SR_NB, s));
        SR_A(sr_launch_gemm_tn(U12t, Np0, invS, SR_NB, X, SR_NB, Np0, SR_NB, SR_NB, 1.0, 0.0, 0, s));
        // Y2 = -U^-1 X   (A = U^-T = transpose of U^-1, k-major, lower block triangular)
        SR_A(sr_launch_transpose(Wt0, Wtr, Np0, s));
        SR_A(sr_launch_gemm_tn_splitk(Wtr, Np0, X, SR_NB, Y2, Np0, SR_NB, Np0, APP_KS, -1.0, 4, part, s));
        SR_A(sr_launch_append_assemble(Wt0, Np0, off0, N0, Y2, invS, m, Wt1 + (size_t)d * NN1, Np1, off1, s));
        // alpha1 = [alpha0 + Y2 v2 ; U22^-1 v2],  v2 = U22^-T (y_new - mu_old(z_new)): no pass over U^-1
        SR_A(sr_launch_append_alpha(h->alpha + (size_t)d * Np0, Np0, N0, Y2, invS, h->mu_part, nsplit, n_out, d, SR_NB, Ynew,
                                    m, alpha1 + (size_t)d * Np1, Np1, s, pf));
    }
    std::vector<int> info_h(n_out, 0);
    SR_AH(hipMemcpyAsync(info_h.data(), info_dev, sizeof(int) * n_out, hipMemcpyDeviceToHost, s));
    SR_AH(hipStreamSynchronize(s));
#undef SR_A
#undef SR_AH
    int bad = 0;
    for (int d = 0; d < n_out; ++d) {
        if (info_h[d] > 0) info_h[d] = N0 + std::max(1, info_h[d] - pf);
        if (info) info[d] = info_h[d];
        if (info_h[d] != 0 && !bad) bad = d + 1;
    }
    if (bad) {
        drop_new();
        sr_set_error("sr_gp_append: Schur complement not positive definite (output %d, point %d)", bad - 1, info_h[bad - 1]);
        return SR_ENOTPD;
    }
    double* old_wt = h->Wt;
    dev_free(h->Z); dev_free(h->yT); dev_free(h->alpha);
    h->Z = Z1; h->yT = yT1; h->alpha = alpha1; h->Wt = Wt1;
    h->z_cap = N1;
    h->N = N1;
    h->logdet_valid = 0;
    if (Np1 == Np0) {
        // keep the previous buffer for the next append (bounded: not for huge factors)
        if (!reuse_alt) dev_free(h->Wt_alt);
        if ((size_t)n_out * NN0 * sizeof(double) <= SR_FACT_PAR_BYTES * 2) { h->Wt_alt = old_wt; h->wt_alt_cap = (size_t)n_out * NN0; h->wt_alt_off = off0; }
        else { dev_free(old_wt); h->Wt_alt = nullptr; h->wt_alt_cap = 0; h->wt_alt_off = -1; }
    } else {
        // everything sized by Np is dropped and re-created lazily
        dev_free(old_wt);
        dev_free(h->Wt_alt); h->Wt_alt = nullptr; h->wt_alt_cap = 0; h->wt_alt_off = -1;
        h->Np = Np1;
        free_ws(h);
        dev_free(h->lin_v); dev_free(h->lin_g); dev_free(h->small_vp); dev_free(h->splitk_vt); dev_free(h->splitk_part);
        h->lin_v = h->lin_g = h->small_vp = h->splitk_vt = h->splitk_part = nullptr;
        h->splitk_cap = 0;
        dev_free(h->stream_vp); dev_free(h->stream_tickets);
        h->stream_vp = nullptr; h->stream_vp_cap = 0; h->stream_tickets = nullptr;
        dev_free(h->fact_ws); h->fact_ws = nullptr; h->fact_cap = 0;
        dev_free(h->app_ws); h->app_ws = nullptr; h->app_cap = 0;
    }
    return SR_OK;
}
