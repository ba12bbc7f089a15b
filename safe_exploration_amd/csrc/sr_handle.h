// sr_handle.h -- the model handle behind sr_gp_t and the host-side helpers the entry-point files share.
// Private to the library (include/safereach.h is the public face).  The entry points live in
//   sr_capi_handle.hip     handle life cycle, device-memory block cache, data, export / import / packed replication,
//                          switches, per-kernel timing
//   sr_capi_update.hip     model update: Gram -> blocked Cholesky -> U^-1 -> alpha (sr_gp_factorize) and its diagnostics
//   sr_capi_append.hip     block row append (sr_gp_append)
//   sr_capi_posterior.hip  workspace, dispatch of the posterior pass (sr_gp_predict, sr_gp_linearize, sr_gp_call1),
//                          input transform, completion mailbox
//   sr_capi_reach.hip      reachability / moment / sampling entry points and the persistent-chain dispatch
//   sr_capi_server.hip     resident single-query server: start / stop / blocking call
#pragma once
#include "sr_mfma_tile.h"
#include <atomic>
#include <chrono>
#include <mutex>
#include <thread>
#include <vector>
#include <algorithm>

// resident single-query server (sr_capi_server.hip, kernel K0s of sr_server.hip)
struct sr_server {
    int armed = 0;                           // sr_gp_server_start was called: sr_gp_server_call (re)launches as needed
    int running = 0;                         // a launch of this handle may be resident
    void* pinned = nullptr; size_t pinned_bytes = 0;      // one pinned block: mailbox | reply words | reply block
    unsigned long long *mb = nullptr, *reply = nullptr; double* out = nullptr;              // host addresses
    unsigned long long *mb_dev = nullptr, *reply_dev = nullptr; double* out_dev = nullptr;  // the device's addresses of the same
    hipStream_t stream = nullptr;            // non-blocking stream of its own: nothing else is ever ordered behind the kernel
    unsigned long long next_seq = 1, idle_ticks = 500000, epoch = 0;
    unsigned long long cmd = 0;              // command of the request in the mailbox (the line's word 5 also carries the epoch and is
                                             // overwritten when a launch is called off: the host keeps the command)
    long launches = 0, calls = 0;
    int stale = 0;                           // a request was given up (time-out): the next call starts from a fresh launch
    std::mutex mu;                           // one caller at a time: a call, and a quiesce from another thread's entry point
};

struct sr_gp {
    int device = 0, N = 0, Np = 0, D = 0, n_out = 0;
    sr_server srv;
    // persistent device state
    double *Z = nullptr, *yT = nullptr, *ls = nullptr, *sf2 = nullptr, *noise = nullptr,
           *alpha = nullptr, *Wt = nullptr;
    double* kp = nullptr;     // general kernel family: n_out x SR_KP(D) packed parameters (else NULL)
    // GP input transform of the reachability / moment entry points: x_gp = Tz x (Tz n_xin x n_s), NULL = identity
    double* Tz = nullptr; int n_xin = 0;
    double *tz_x = nullptr, *tz_jac = nullptr; long tz_cap = 0;   // transformed inputs / chain-ruled Jacobians (per chunk)
    // persistent multi-step kernel (sr_chain.hip K0c): exchange buffer, per group ticket + epoch + done counter (all of
    // the hand-off state lives on the device), switch
    sr_xel* chain_xch = nullptr; unsigned long long* chain_tickets = nullptr;       // the groups' epochs
    unsigned* chain_done = nullptr; int chain = 1; int last_chain = 0; int chain_cap = -1;
    int* chain_status_host = nullptr; int* chain_status_dev = nullptr;   // pinned: set by a chain launch that timed out
    int chain_occ_key = -1, chain_occ_blocks = 0;                         // occupancy of the kernel last asked about
    int chain_test_drop = 0;                                              // sr_test_chain_drop
    unsigned* call_ticket = nullptr;        // sr_gp_call1: workgroups done (reset by the last one)
    int general = 0;
    int have_data = 0, factorized = 0;
    int import_open = 0;     // between sr_gp_import_begin and sr_gp_import_end
    // per-chunk workspace (grow-only)
    long chunk = 65536, ws_Tp = 0, ws_part = 0;     // ws_part: capacity of mu_part in units of n_out doubles
    int ws_locked = 0;       // internal buffers (mu / var / jac) are referenced by an entry point: growing now is a bug
    double *Ks = nullptr, *mu_part = nullptr, *jac_part = nullptr, *var_part = nullptr,
           *mu = nullptr, *var = nullptr, *jac = nullptr, *kxx = nullptr;
    double *lin_v = nullptr, *lin_g = nullptr, *small_vp = nullptr;   // small-batch scratch
    size_t lin_cap = 0;                                                 // doubles behind lin_v
    double* stream_vp = nullptr; long stream_vp_cap = 0;   // fused small-batch path: partial sums (grow-only)
    unsigned* stream_tickets = nullptr;
    int* stream_tab = nullptr; long stream_tab_cap = 0; long stream_tab_key = -1; int stream_tab_kr = 0, stream_tab_n = 0, stream_tab_nwg = 0;   // work items of the run kernel (+ workgroups they are dealt to)
    double* stream_slots = nullptr; long stream_slots_cap = 0;    // self-validating hand-over slots of the T = 1 kernel's polling finaliser
    double* splitk_vt = nullptr; long splitk_cap = 0;   // partial products of the balanced few-query-tile route (grow-only)
    // log det(K + noise) per output as of the last <= 16-row append (read back with its status words): the blocking read of
    // sr_gp_logdet costs the exploration loop 30 us per step
    std::vector<double> logdet_host; int logdet_valid = 0;
    double* splitk_part = nullptr;                      // n_out x 4 nrb x Tp partial norms (<= 4 MB)     // 2 x (n_out x Np) scratch of sr_gp_linearize
    int var_group = 64;      // query tiles per scheduling group of the variance kernel.  With the diagonal blocks cut short
                             // (variant 2) 64 beats 32: 70.7 against 70.0 TF at C2', fabric-side fetches 61.3 -> 42.9 M KiB per launch
                             // (scripts/pmc_groups.sh); 256 and more lose the sharing of the K* tiles (68.7 TF)
    int small_path = 1;      // latency paths (streaming T <= 16, 64-tiles, split-K) instead of the plain MFMA tiles
    int last_streamed = 0;   // the last gp_pass went through the streaming kernels (their partials hold U^-T k*)
    int force_stream = 0;    // sr_gp_linearize wants those partials whatever the model size
    int var_variant = 4;     // 1: the loop of rounds 1 - 4 (LDS-DMA, barrier on top of a k-tile: 70 TF; the A/B reference), 3: the
                             // pipelined loop of round 5 (barrier under the MFMA stream: 73.1 TF), 4: 3 + diagonal blocks without
                             // their structural zeros (default: 74.5 TF)
    // factorisation: the outputs are independent problems -- below SR_FACT_PAR_BYTES of scratch each gets its own
    // HIP stream (the small-grid kernels of a modest model then overlap) and the scratch stays with the handle
    double* fact_ws = nullptr; size_t fact_cap = 0;      // n_par x (U, W: Np^2 each, v: Np)
    // row append of few points (m <= 16): scratch and a second U^-1 buffer the new factor is assembled into
    // (kept while the padded size does not change: appends then allocate nothing big)
    double* app_ws = nullptr; size_t app_cap = 0;
    unsigned long long* call_flag = nullptr; unsigned long long call_seq = 0;   // set by sr_gp_call1 around a streamed pass
    // In-place one-point appends (sr_capi_append.hip): Wt, alpha and yT may be VIEWS `slide` steps into their allocations
    // (Wt: slide * (Np + 1) doubles, alpha / yT: slide doubles); every entry point that rewrites the model, and every kernel
    // that wants U^-1 aligned to 16 bytes, calls unslide() first.  slack_ok: the three allocations carry the zeroed slack.
    int slide = 0; int slack_ok = 0;
    // A tile route of the posterior pass (16-byte reads of U^-1) that meets an odd slide has to unslide: a full copy of the
    // factor and a device-wide wait.  A loop that alternates one append with one big batch would pay that every step, so such
    // an unslide keeps the next `slide_hold` one-point appends off the in-place route (64, doubling with every forced
    // unslide since the last refit: slide_forced).
    int slide_hold = 0, slide_forced = 0;
    // Grid route of the one-point append: a grid that cannot become resident gives up after ~5 ms.  After such an abort the
    // route is left alone for `grid_hold` one-point appends (16, doubling with every further abort in a row, at most 16384).
    int grid_hold = 0, grid_aborts_row = 0; long grid_aborts = 0;
    double* appg_cnt = nullptr; unsigned appg_base = 0, appg_q = 0;   // (arrivals and barriers of all launches so far)      // barrier counters of the grid append (zero at allocation), their value after the last launch
    void* app_pin = nullptr; double* app_pin_dev = nullptr;   // pinned, mapped: results of sr_gp_append1_host (log det partials, status words)
    double* Wt_alt = nullptr; size_t wt_alt_cap = 0;
    int wt_alt_off = -1;     // front padding of the (complete, well-formed) factor Wt_alt last held; -1 unknown
    // small appends allocate nothing while the padded size stays: Z has room for z_cap points, yT / alpha ping-pong
    long z_cap = 0; double *yT_alt = nullptr, *alpha_alt = nullptr; int vec_alt_np = 0;
    std::vector<double> sf2_host, noise_host;    // host copies of sf2 / noise (filled on first use after set_data)
    // up to SR_FACT_SLOTS outputs are factorised at once as a BATCH (every launch covers all of them).  Streams:
    // CRITICAL (diagonal blocks, panel rows, look-ahead rows, late inversion), BULK (the trailing update behind the
    // look-ahead rows) and INVERSION (the early part of the triangular inversion); the caller's stream waits for them
    hipStream_t fact_stream = nullptr, bulk_stream = nullptr, inv_stream = nullptr;
    // pipelined chain (round 6, chain-bound sizes): the diagonal blocks on a stream of their own BESIDE the rest of the
    // previous block row (row stream); the three streams hand over through counters in device memory (sr_fact_handover_kernel)
    hipStream_t diag_stream = nullptr, row_stream = nullptr;
    unsigned* fact_flags = nullptr; int fact_flags_nb = 0;   // [status | c[nb] | d[nb] | r[nb]]; zero at allocation, values = epochs
    unsigned fact_epoch = 0;
    // tile-flow Cholesky (round 6; sr_flow.h): its counters, their capacity in words, the epoch of the last run
    unsigned* flow_flags = nullptr; long flow_words = 0; unsigned flow_epoch = 0;
    void* flow_segs = nullptr; int flow_segs_key[3] = {0, 0, 0}; long flow_total = 0, flow_total_far = 0, flow_total_upd = 0, flow_total_m = 0;   // task plan on the device for (nb, band, panel)
    int fact_pipe = 0;                                    // sr_gp_set_fact_pipeline: 0 = one chain of launches (default: the pipelined
                                                          // forms measured slower, profiles/r06_fact_pipeline.txt), 1 = three streams, 2 = two,
                                                          // 3 = tile-flow Cholesky (one resident kernel), -1 = never the tile flow
    int last_fact_pipe = 0;                               // the last update ran pipelined (diagnostics / tests)
    hipEvent_t fact_fork = nullptr, fact_join = nullptr;
    hipEvent_t ev_panel[2] = {nullptr, nullptr}, ev_bulk[2] = {nullptr, nullptr};
    hipEvent_t ev_inv[2] = {nullptr, nullptr};            // critical -> inversion stream, back
    int fact_panel = 0;                                  // blocks per Cholesky panel; 0 = by size
    int fact_regime = 0;                                 // how the streams below were made: 0 none, else 1000 x (1 chain-bound, 2 GEMM-bound) + reserved CUs
    int ncu = 0;                                         // compute units of the device (cached)
    // job lists of the level-batched triangular inversion (depend on Np only)
    int* fact_info = nullptr;                            // status words of the factorisation (64 ints)
    size_t mem_total = 0;                                // device memory (cached)
    sr_gemm_job* inv_jobs = nullptr; int inv_jobs_np = 0;
    // jobs of a level in ascending order of their block range [lo, hi); per job: where it ends, where its left child ends
    // (blocks), its 128-tiles -- the inversion is launched in stages as the Cholesky passes those points (sr_capi_update.hip)
    struct inv_level { int off1, off2, count, maxM, maxN; long tiles; int depth; std::vector<int> hi, mid; std::vector<long> tl; };
    std::vector<inv_level> inv_levels;
    sr_prof prof;
};
#define SR_FACT_PAR_BYTES ((size_t)8 << 30)

// every entry point runs on the handle's device and leaves the caller's current device as it found it
// (PyTorch reads its current device from the HIP runtime)
struct sr_dev_guard {
    int prev = -1; hipError_t err = hipSuccess;
    explicit sr_dev_guard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) err = hipSetDevice(dev); else prev = -1;
    }
    ~sr_dev_guard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
#define SR_DEVICE(dev) sr_dev_guard dev_guard_(dev); SR_HIP(dev_guard_.err)

namespace srh {

static inline long round_up(long v, long m) { return (v + m - 1) / m * m; }

// one turn of a host spin loop (the pause hint where the host has one)
static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    std::this_thread::yield();
#endif
}

// device memory through the block cache (sr_capi_handle.hip)
int dev_alloc_bytes(void** p, size_t bytes);
template <typename T>
static inline int dev_alloc(T** p, size_t count) { return dev_alloc_bytes((void**)p, count * sizeof(T)); }
void dev_free(void* p);
int dev_zero(void* p, size_t bytes);
void free_ws(sr_gp* h);
int ensure_wt(sr_gp* h);
// doubles of the three model buffers with their slack (see sr_gp::slide)
static inline size_t wt_doubles(int n_out, int Np) { return (size_t)n_out * Np * Np + (size_t)SR_SLIDE_STEPS * (Np + 1); }
static inline size_t vec_doubles(int n_out, int Np) { return (size_t)n_out * Np + SR_SLIDE_STEPS; }
static inline double* wt_alloc_of(const sr_gp* h) { return h->Wt ? h->Wt - (size_t)h->slide * (h->Np + 1) : nullptr; }
static inline double* alpha_alloc_of(const sr_gp* h) { return h->alpha ? h->alpha - h->slide : nullptr; }
static inline double* yT_alloc_of(const sr_gp* h) { return h->yT ? h->yT - h->slide : nullptr; }
int unslide(sr_gp* h);               // back to plain buffers (fresh allocations, two contiguous copies); no-op when slide == 0

// resident server (sr_capi_server.hip): off the device before the model is written / before a device-wide wait
int server_quiesce(sr_gp* h);
void servers_quiesce_device(int device);
void server_release(sr_gp* h);
// hipDeviceSynchronize for the library: resident servers of the current device leave first (they would keep the wait
// until their idle time-out otherwise)
static inline hipError_t device_sync() {
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) servers_quiesce_device(dev);
    return hipDeviceSynchronize();
}

// posterior pass and its workspace (sr_capi_posterior.hip)
int pick_nsplit(const sr_gp* h, long Tp);
int ensure_ws(sr_gp* h, long Tp, int nsplit);
int prepare_ws(sr_gp* h, long Tc);
struct sr_ws_lock {
    sr_gp* h;
    explicit sr_ws_lock(sr_gp* h_) : h(h_) { h->ws_locked = 1; }
    ~sr_ws_lock() { h->ws_locked = 0; }
};
int gp_pass(sr_gp* h, long Tc, const double* xa, long lda, int na, const double* xb, long ldb,
            int nb, double* mu, double* var, double* jac, hipStream_t s);
int gp_pass_states(sr_gp* h, long Tc, const double* p, long ldp, int n_s, const double* kff, long ldkff, int n_u,
                   double* mu, double* var, const double** jac_out, hipStream_t s);

}  // namespace srh
