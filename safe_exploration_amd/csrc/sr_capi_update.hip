// sr_capi_update.hip -- model update entry points: sr_gp_factorize (Gram -> blocked fp64 Cholesky with MFMA trailing
// updates -> explicit U^-1 -> alpha), its stream set, and the diagnostics that read the factor (marginal likelihood,
// log determinant, explicit K^-1, GEMM / diagonal-block test hooks).
#include "sr_handle.h"
#include "sr_flow.h"
using namespace srh;

// set when a hand-over of the pipelined chain ran into its time-out once (a foreign stream of the chain's priority on one of
// its hardware queues): the process stays on the plain chain from then on
static std::atomic<bool> g_pipe_broken{false};
// ... or a wait of the tile-flow Cholesky did (its diagonal-block workgroups not resident, or on the workers' hardware queue)
// A flow that failed rests: the next g_flow_rest updates that would be tile flows run by launches, then it is tried again; every
// further failure doubles the rest (16, 32, .. 4096 updates).  (Not "never again": the one spurious time-out seen -- a first update
// on fresh scratch of many gigabytes -- would have kept its process on launches for good.)
static std::atomic<long> g_flow_rest{0};
static std::atomic<long> g_flow_rest_len{16};
static std::atomic<int> g_test_flow_fail{0};          // sr_test_flow_fail: the next tile flows' diagonal-block workgroups never get their go

// ---------------------------------------------------------------------------------------------
// factorisation: K = U^T U (right-looking, 128-blocks in panels, look-ahead), U^-T / U^-1 by recursive halving
// with all nodes of a level in one launch, alpha = U^-1 (U^-T y)
// ---------------------------------------------------------------------------------------------
// job lists of the recursive inversion  [L11 0; L21 L22]^-1 = [W11 0; -W22 L21 W11, W22],  L21 = U12^T,
// one list per depth of the halving tree (children before parents):
//   product 1:  Y   = U12^T W11   (A = U12 k-major -- the factor's off-diagonal blocks live in W's upper block
//               triangle --, B = W11 lower-triangular: mode 2) -> parked in the unused strict lower triangle of U
//   product 2:  W21 = -Wt22^T Y   (A = Wt22 = U22^-1 upper-triangular: mode 3), written to W and, transposed,
//               to Wt12 (keeps U^-1 complete for the parent level)
static int ensure_inv_jobs(sr_gp* h) {
    if (h->inv_jobs && h->inv_jobs_np == h->Np) return SR_OK;
    const int Np = h->Np, nb = Np / SR_NB;
    struct Node { int lo, hi, depth; };
    std::vector<Node> nodes, stack;
    stack.push_back({0, nb, 0});
    int max_depth = 0;
    while (!stack.empty()) {
        const Node r = stack.back();
        stack.pop_back();
        if (r.hi - r.lo <= 1) continue;
        nodes.push_back(r);
        max_depth = std::max(max_depth, r.depth);
        const int mid = (r.lo + r.hi) / 2;
        stack.push_back({r.lo, mid, r.depth + 1});
        stack.push_back({mid, r.hi, r.depth + 1});
    }
    std::vector<sr_gemm_job> jobs;
    h->inv_levels.clear();
    std::sort(nodes.begin(), nodes.end(), [](const Node& a, const Node& b) { return a.lo < b.lo; });
    for (int depth = max_depth; depth >= 0; --depth) {
        sr_gp::inv_level lv{};
        lv.depth = depth;
        std::vector<sr_gemm_job> j1, j2;
        for (const Node& r : nodes) {                             // (ascending lo: the jobs the Cholesky has passed are a prefix)
            if (r.depth != depth) continue;
            const int mid = (r.lo + r.hi) / 2;
            const int n1 = (mid - r.lo) * SR_NB, n2 = (r.hi - mid) * SR_NB;
            const long o11 = (long)r.lo * SR_NB * Np + (long)r.lo * SR_NB;
            const long o12 = (long)r.lo * SR_NB * Np + (long)mid * SR_NB;
            const long o21 = (long)mid * SR_NB * Np + (long)r.lo * SR_NB;
            const long o22 = (long)mid * SR_NB * Np + (long)mid * SR_NB;
            j1.push_back({o12, o11, o21, 0, n2, n1, n1, 0});      // Y (in U's lower triangle) = U12^T W11
            j2.push_back({o22, o21, o21, o12, n2, n1, n2, 0});    // W21 = -Wt22^T Y ; Wt12 = W21^T
            lv.maxM = std::max(lv.maxM, n2);
            lv.maxN = std::max(lv.maxN, n1);
            const long tl = (long)(n2 / SR_NB) * (n1 / SR_NB);
            lv.tiles += tl;
            lv.hi.push_back(r.hi); lv.mid.push_back(mid); lv.tl.push_back(tl);
        }
        lv.count = (int)j1.size();
        if (lv.count == 0) continue;
        lv.off1 = (int)jobs.size();
        jobs.insert(jobs.end(), j1.begin(), j1.end());
        lv.off2 = (int)jobs.size();
        jobs.insert(jobs.end(), j2.begin(), j2.end());
        h->inv_levels.push_back(lv);
    }
    dev_free(h->inv_jobs);
    h->inv_jobs = nullptr; h->inv_jobs_np = 0;
    if (jobs.empty()) { h->inv_jobs_np = Np; return SR_OK; }
    SR_TRY(dev_alloc(&h->inv_jobs, jobs.size()));
    SR_HIP(hipMemcpy(h->inv_jobs, jobs.data(), jobs.size() * sizeof(sr_gemm_job), hipMemcpyHostToDevice));
    h->inv_jobs_np = Np;
    return SR_OK;
}

// blocks per Cholesky panel.  Inside a panel a factored block row updates only the panel's remaining rows;
// everything below is updated once per panel with K = panel * 128, which divides the read-modify-write traffic
// of the trailing matrix by `panel` and gives the bulk update K = panel * 128 (a K = 128 update spends most of its
// time in the prologue and the read-modify-write epilogue of its tiles: measured 35 % of the K = 512 rate).
// Measured at N = 1000 ... 10000: 4 beats 1 and 2 everywhere (N = 5000: 6.9 against 7.2-7.9 ms).
static int pick_fact_panel(const sr_gp* h) {
    if (h->fact_panel > 0) return h->fact_panel;
    const int nb = h->Np / SR_NB;
    // measured (scripts/factor_bench.py, n_out = 2): N = 3000 .. 6000 panels of 2 are 2 - 4 % ahead of 4 (3.33 / 3.49,
    // 4.91 / 5.01, 6.82 / 7.01, 9.24 / 9.41 ms), N = 7000 and 10000 panels of 4 (12.99 / 13.04, 28.3 / 30.2 ms)
    // N = 50000 (391 blocks): panels of 4 / 8 / 16 / 32 blocks 2.735 / 2.655 / 2.625 / 2.642 s; N = 30000: 8 and 16 alike
    // round 3 (diagonal block 62 -> 30 us, batched outputs), n_out = 2, panels of 2 / 3 / 4 / 6 / 8 blocks: N = 3000 2.22 / 2.26 /
    // 2.34 / 2.41 / 2.52 ms; N = 5000 5.26 / 5.15 / 5.15 / 5.29 / 5.64; N = 7000 11.0 / 10.5 / 10.2 / 10.3 / 10.5; N = 10000
    // 28.3 / 26.7 / 25.8 / 25.5 / 25.0; N = 20000 (4 / 8 / 12 / 16) 186.8 / 184.3 / 187.4 / 188.6; N = 30000 592 / 578 / 575 /
    // 574; N = 50000 (8 / 12 / 16 / 24) 2.599 / 2.580 / 2.570 / 2.555 s
    // with the in-panel updates of long K on 128-tiles (sr_use_tile64): N = 30000 panels of 12 / 16 / 24 / 32: 565 / 562 / 561 / 562 ms;
    // N = 50000 panels of 24 / 32 / 48: 2.521 / 2.514 / 2.508 s
    return sr_fact_panel(nb);                       // (the table of sr_common.h)
}

// Streams of the factorisation.  The chain of diagonal blocks is latency-bound and must never queue behind the
// (throughput-bound) bulk update; the diagonal-block kernel moreover needs a CU to itself (150 KB of LDS), and a
// workgroup is bound to a shader engine before it waits for a CU.  CU masks are dealt by the driver round-robin over
// the 8 XCDs and, inside an XCD, over its 4 shader engines (scripts/cumask_probe.hip).
//   regime 1 (chain-bound sizes, nb <= 128): critical stream = highest priority, every CU; bulk stream = CU mask
//     without the first 32 bits: one CU per shader engine stays free, wherever the diagonal block lands
//     (measured at N = 5000: 61 us alone, 140-210 us beside an unmasked bulk update, 75 us with the reserve);
//     (a stream of their own for the diagonal blocks, so that a block is factored BESIDE the update of the rest of
//     its block row, was measured here and lost: every hand-over between two hardware queues costs more than the
//     25 us it hides -- N = 5000: 6.9 -> 15.2 ms.  The same holds for a third priority stream per output and for a
//     row-wise inversion running beside the chain: 6.9 -> 12.5 resp. 8.0 -> 10.6 ms.  One critical stream per output.)
//   regime 2 (GEMM-bound sizes): the bulk stream leaves only the first 8 bits out (one CU per XCD, 3 % of the
//     chip), and the diagonal blocks run on a third stream that owns exactly those 8 CUs -- the bulk tiles last
//     200 us there and would otherwise keep a diagonal block waiting for ~1.4 ms (N = 50000: the chain of 391 blocks
//     then IS the critical path of the whole Cholesky).  The critical stream keeps every CU and its priority (a
//     CU-masked stream cannot have one: with all three streams masked the look-ahead rows queued 1:1 with the bulk
//     tiles and the update got 7 % slower); it is idle while a diagonal block runs, so the reserve is free then.
// Few streams on purpose: streams of one priority share a small pool of hardware queues, and two chains on one
// queue serialise (a third high-priority stream per output cost 45 % at N = 5000).
static int make_masked_stream(hipStream_t* st, int ncu, int first_bit, int last_bit) {
    std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
    for (int c = first_bit; c < last_bit; ++c) mask[c / 32] |= 1u << (c % 32);
    SR_HIP(hipExtStreamCreateWithCUMask(st, (uint32_t)mask.size(), mask.data()));
    return SR_OK;
}

// The streams of the model update belong to the PROCESS (per device and (regime, reserved CUs)), not to a handle: creating
// them -- a priority stream and two CU-masked ones -- takes 12 - 18 ms, which every handle of a model whose size changes
// used to pay again (a refit after a change of N: 48 ms, of which 5 were the update).  Updates of different handles that
// share them simply queue behind each other.
// ONE set per device at a time (round 4).  Sets of different keys used to pile up -- a 1000-point model (reserve 32) and
// then a 5000-point one (reserve 64) in the same process: two priority streams and four CU-masked ones -- and the streams
// of one priority share a small pool of hardware queues: the critical chain of the second model then shared a queue with an
// idle stream of the first and its kernels took twice as long in situ (diagonal block 38 -> 75 us, the whole update 4.9 ->
// 7.6 - 8.9 ms; profiles/archive/r04_factor_bench.txt against r03's).  A set of another key is destroyed when a new one is made,
// unless an update is running on it at that moment (`busy`).
struct sr_stream_set { int device, key, busy; hipStream_t fact, bulk, inv, diag, row; int pipe_ok; };
static std::mutex g_stream_mutex;
static std::vector<sr_stream_set> g_stream_sets;
static void release_fact_streams(sr_gp* h) {           // end of an update: the handle forgets the streams, the set is free
    if (!h->fact_stream) return;
    std::lock_guard<std::mutex> lk(g_stream_mutex);
    for (sr_stream_set& c : g_stream_sets)
        if (c.device == h->device && c.fact == h->fact_stream && c.busy > 0) --c.busy;
    h->fact_stream = h->bulk_stream = h->inv_stream = h->diag_stream = h->row_stream = nullptr;
    h->fact_regime = 0;
}

// Do the three streams run on three different hardware queues?  For every ordered pair: a waiter on one, the signal on
// the other; a waiter whose signal sits behind it in the same queue gives up after 3 ms and says so.
static bool own_queues(hipStream_t a, hipStream_t b, hipStream_t c) {
    unsigned* f = nullptr;
    if (hipMalloc((void**)&f, 8 * sizeof(unsigned)) != hipSuccess) { (void)hipGetLastError(); return false; }
    bool ok = hipMemset(f, 0, 8 * sizeof(unsigned)) == hipSuccess;
    hipStream_t st[3] = {a, b, c};
    unsigned v = 0;
    for (int i = 0; i < 3 && ok; ++i)
        for (int j = 0; j < 3 && ok; ++j) {
            if (i == j) continue;
            ++v;
            ok = sr_launch_fact_handover(nullptr, 0, f + 1, v, nullptr, 0, f, 3e-3, st[i]) == SR_OK &&
                 sr_launch_fact_handover(f + 1, v, nullptr, 0, nullptr, 0, f, 3e-3, st[j]) == SR_OK &&
                 hipStreamSynchronize(st[i]) == hipSuccess && hipStreamSynchronize(st[j]) == hipSuccess;
        }
    unsigned status = 1;
    if (ok) ok = hipMemcpy(&status, f, sizeof(unsigned), hipMemcpyDeviceToHost) == hipSuccess && status == 0;
    (void)hipFree(f);
    return ok;
}

// Stream of the resident diagonal-block workgroups of the tile-flow Cholesky: one per device for the life of the process,
// non-blocking (the caller's stream may be the null stream) and of the lowest priority (another pool of hardware queues than
// the update's priority stream).  Its kernel WAITS for the workers on the device, so the two must not share a hardware queue:
// checked once per (device, worker stream) with a pair of hand-overs in either order.
struct sr_flow_stream { hipStream_t srv = nullptr, inv = nullptr; hipStream_t checked_with = nullptr; bool checked = false, ok = false; };
static sr_flow_stream g_flow_streams[64];
static int flow_server_stream(int device, hipStream_t worker, hipStream_t* out, hipStream_t* inv_out) {
    std::lock_guard<std::mutex> lk(g_stream_mutex);
    SR_CHECK(device >= 0 && device < 64, SR_EINVAL, "device index %d", device);
    sr_flow_stream& f = g_flow_streams[device];
    if (!f.srv) {
        int prio_lo = 0, prio_hi = 0;
        SR_HIP(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
        SR_HIP(hipStreamCreateWithPriority(&f.srv, hipStreamNonBlocking, prio_lo));
        // ... and the stream of the inversion's stages that run beside the flow (each behind a gate kernel that waits on the
        // device; everything on it is launched AFTER the diagonal-block workgroups and the workers, so a hardware queue it
        // shares with either only costs the overlap)
        SR_HIP(hipStreamCreateWithPriority(&f.inv, hipStreamNonBlocking, prio_lo));
    }
    if (!f.checked || f.checked_with != worker) {
        unsigned* w = nullptr;
        bool ok = hipMalloc((void**)&w, 8 * sizeof(unsigned)) == hipSuccess && hipMemset(w, 0, 8 * sizeof(unsigned)) == hipSuccess;
        hipStream_t st[2] = {f.srv, worker};
        for (int i = 0; i < 2 && ok; ++i) {
            const unsigned v = (unsigned)(i + 1);
            ok = sr_launch_fact_handover(nullptr, 0, w + 1, v, nullptr, 0, w, 3e-3, st[i]) == SR_OK &&
                 sr_launch_fact_handover(w + 1, v, nullptr, 0, nullptr, 0, w, 3e-3, st[1 - i]) == SR_OK &&
                 hipStreamSynchronize(st[i]) == hipSuccess && hipStreamSynchronize(st[1 - i]) == hipSuccess;
        }
        unsigned status = 1;
        if (ok) ok = hipMemcpy(&status, w, sizeof(unsigned), hipMemcpyDeviceToHost) == hipSuccess && status == 0;
        if (w) (void)hipFree(w);
        (void)hipGetLastError();
        f.checked = true; f.checked_with = worker; f.ok = ok;
    }
    *out = f.ok ? f.srv : nullptr;
    *inv_out = f.ok ? f.inv : nullptr;
    return SR_OK;
}

static int ensure_fact_events(sr_gp* h) {
    if (!h->fact_join) SR_HIP(hipEventCreateWithFlags(&h->fact_join, hipEventDisableTiming));
    for (int e = 0; e < 2; ++e) {
        if (!h->ev_panel[e]) SR_HIP(hipEventCreateWithFlags(&h->ev_panel[e], hipEventDisableTiming));
        if (!h->ev_bulk[e]) SR_HIP(hipEventCreateWithFlags(&h->ev_bulk[e], hipEventDisableTiming));
        if (!h->ev_inv[e]) SR_HIP(hipEventCreateWithFlags(&h->ev_inv[e], hipEventDisableTiming));
    }
    return SR_OK;
}

static int ensure_fact_streams(sr_gp* h, int regime) {
    if (h->ncu == 0) {
        int cus = 0;
        SR_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device));
        h->ncu = cus;
    }
    const int ncu = h->ncu;
    const bool can_mask = ncu >= 64;
    // (giving each output's chain one half of the XCDs through CU masks on ALL of its streams was measured and lost:
    //  N = 5000, two outputs 5.6 -> 8.7 ms -- kernels on a CU-masked queue start late, and the mask costs the priority)
    // CUs the bulk streams leave to the critical chain: sr_fact_reserved_cus (the table of sr_common.h): one per shader
    // engine; two for 36 < Np / 128 <= 52, where the block-row solves and in-panel updates of the chain otherwise queue
    // behind the trailing update's workgroups (50 instead of 12 us each) and the trailing update has the slack (measured,
    // reserve 32 / 64: N = 3500 2.79 / 2.80, N = 5000 5.14 / 4.99, N = 6500 8.79 / 8.74, N = 7500 12.17 / 12.45, N = 10000 25.1 / 26.7 ms)
    const int nblk = h->Np / SR_NB;
    static const int reserve_lab = (int)sr_lab_env("SR_FACT_RESERVE", 0);       // (lab build: CUs the bulk streams leave free)
    const int reserve = reserve_lab > 0 ? reserve_lab : sr_fact_reserved_cus(regime, nblk);
    const bool want_pipe = regime == 1 && (h->fact_pipe == 1 || h->fact_pipe == 2);      // (the prototype's two extra streams only where it is asked for)
    const int key = regime * 1000 + reserve + (want_pipe ? 500 : 0);
    {
        std::lock_guard<std::mutex> lk(g_stream_mutex);
        sr_stream_set* set = nullptr;
        for (sr_stream_set& c : g_stream_sets)
            if (c.device == h->device && c.key == key) set = &c;
        if (!set) {
            // idle sets of other keys on this device go first (their hardware queues are what the new set needs)
            for (size_t i = 0; i < g_stream_sets.size();) {
                sr_stream_set& c = g_stream_sets[i];
                if (c.device == h->device && c.busy == 0) {
                    for (hipStream_t st : {c.fact, c.bulk, c.inv, c.diag, c.row})
                        if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
                    g_stream_sets.erase(g_stream_sets.begin() + i);
                } else ++i;
            }
            sr_stream_set c{h->device, key, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 0};
            int prio_lo = 0, prio_hi = 0;
            SR_HIP(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
            SR_HIP(hipStreamCreateWithPriority(&c.fact, hipStreamNonBlocking, prio_hi));
            if (can_mask) SR_TRY(make_masked_stream(&c.bulk, ncu, reserve, ncu));
            else SR_HIP(hipStreamCreateWithPriority(&c.bulk, hipStreamNonBlocking, prio_lo));
            if (regime == 1) {
                if (can_mask) SR_TRY(make_masked_stream(&c.inv, ncu, reserve, ncu));
                else SR_HIP(hipStreamCreateWithPriority(&c.inv, hipStreamNonBlocking, prio_lo));
                // the pipelined chain: two more streams of the chain's priority.  Their kernels WAIT for each other on the
                // device, so each must sit on a hardware queue of its own (streams of one priority share a small pool of
                // queues): checked once, here -- a waiter that shares its producer's queue runs into its time-out
                if (want_pipe) {
                    SR_HIP(hipStreamCreateWithPriority(&c.diag, hipStreamNonBlocking, prio_hi));
                    SR_HIP(hipStreamCreateWithPriority(&c.row, hipStreamNonBlocking, prio_hi));
                    c.pipe_ok = own_queues(c.fact, c.diag, c.row) ? 1 : 0;
                }
            } else {
                // regime 2: a second bulk stream WITHOUT a mask, for the trailing updates that are long enough to hide a
                // chain that waits for its CUs (below)
                SR_HIP(hipStreamCreateWithPriority(&c.inv, hipStreamNonBlocking, prio_lo));
            }
            g_stream_sets.push_back(c);
            set = &g_stream_sets.back();
        }
        ++set->busy;
        h->fact_stream = set->fact; h->bulk_stream = set->bulk; h->inv_stream = set->inv;
        h->diag_stream = set->pipe_ok ? set->diag : nullptr; h->row_stream = set->pipe_ok ? set->row : nullptr;
    }
    SR_TRY(ensure_fact_events(h));
    h->fact_regime = key;
    return SR_OK;
}

extern "C" int sr_gp_factorize(sr_gp_t h, void* stream, int* info) {
    SR_CHECK(h != nullptr, SR_EINVAL, "sr_gp_factorize: NULL handle");
    SR_CHECK(h->have_data, SR_ESTATE, "sr_gp_factorize: call sr_gp_set_data first");
    SR_DEVICE(h->device);
    SR_TRY(server_quiesce(h));
    SR_TRY(unslide(h));
    h->slide_hold = 0; h->slide_forced = 0;          // (a refit starts the in-place appends' hold-off afresh: sr_capi_posterior.hip)
    const auto t_begin = std::chrono::steady_clock::now();
    static const bool trace_laps = sr_lab_on("SR_FACT_TRACE");
    auto lap = [&](const char* what) { if (trace_laps) fprintf(stderr, "  %s at %.3f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count()); };
    SR_TRY(ensure_wt(h));
    lap("wt");
    SR_TRY(ensure_inv_jobs(h));
    lap("inv_jobs");
    const int Np = h->Np, nb = Np / SR_NB;
    const int P = pick_fact_panel(h);
    const size_t NN = (size_t)Np * Np;
    const size_t per = 2 * NN + (size_t)Np;              // scratch doubles per output in flight: U, W, v
    // outputs in a batch: as many as fit a third of the device's memory in scratch (8 GB at least), at most SR_FACT_SLOTS.
    // (Round 2 capped the scratch at 8 GB, so the outputs of a big model went one after the other; batched, the latency
    // of one output's chain of panel steps is filled with the other's tiles: N = 50000, n_out = 2 -- 80 GB of scratch
    // beside 40 GB of factors.)
    // The scratch stays with the handle (refits allocate nothing): at N = 50000 a hipMalloc / hipFree of 40 GB per update
    // made the first updates of a process take 4.0 - 4.8 s instead of 2.67 s (page-table work inside the timed call).
    // sr_gp_release_scratch hands it back.
    if (h->mem_total == 0) {
        size_t mem_free = 0;
        (void)hipMemGetInfo(&mem_free, &h->mem_total);
    }
    static const size_t par_cap_env = (size_t)sr_lab_env("SR_FACT_PAR_GB", 0) << 30;
    const size_t par_bytes = par_cap_env ? par_cap_env : std::max<size_t>(SR_FACT_PAR_BYTES, h->mem_total / 3);
    int n_par = (int)std::min<size_t>((size_t)std::min(h->n_out, SR_FACT_SLOTS),
                                      std::max<size_t>(1, par_bytes / (per * sizeof(double))));
    const bool keep = per * n_par * sizeof(double) <= par_bytes;
    double* scratch = nullptr;                           // owned here only when it is not kept in the handle
    int rc = SR_OK;
    hipStream_t s0 = (hipStream_t)stream;
    hipStream_t srv = nullptr, finv = nullptr;           // tile flow: the streams of the resident diagonal-block workgroups / of the inversion beside it
    auto cleanup = [&]() {
        // never return with work in flight on the side streams
        for (hipStream_t st : {h->fact_stream, h->bulk_stream, h->inv_stream, h->diag_stream, h->row_stream, srv, finv}) if (st) (void)hipStreamSynchronize(st);
        release_fact_streams(h);
        dev_free(scratch);
    };
#define SR_F(expr) do { rc = (expr); if (rc != SR_OK) { cleanup(); return rc; } } while (0)
#define SR_FH(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
        sr_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); cleanup(); return SR_EHIP; } } while (0)
    double* ws;
    if (keep) {
        if (h->fact_cap < per * n_par) {
            (void)device_sync();
            dev_free(h->fact_ws);
            h->fact_ws = nullptr; h->fact_cap = 0;
            SR_F(dev_alloc(&h->fact_ws, per * n_par));
            h->fact_cap = per * n_par;
        }
        ws = h->fact_ws;
    } else {
        SR_F(dev_alloc(&scratch, per));
        ws = scratch;
    }
    // (nothing is read back or allocated before the first launch: the Gram kernels take the signal variance and the
    //  noise from device memory, the status words live with the handle -- round 2 paid a D2H copy, a stream
    //  synchronisation, a hipMalloc and a hipMemGetInfo here, 0.1 ms before the first kernel started)
    lap("fact_ws");
    if (!h->fact_info) SR_F(dev_alloc(&h->fact_info, (size_t)64));
    int* info_dev = h->fact_info;
    SR_FH(hipMemsetAsync(info_dev, 0, sizeof(int) * h->n_out, s0));
    const int regime = nb <= SR_FACT_CHAIN_MAX_NB ? 1 : 2;
    // One or two blocks (the reference's own model sizes): a handful of kernels with nothing to run beside each other, on the
    // caller's stream -- the hand-over to the critical stream and back cost 25 of the 85 us of such an update.
    // Up to SR_FACT_ONE_STREAM_MAX_NB (15) blocks the WHOLE update stays on the caller's stream (round 6): no fork to the priority
    // stream and no join back (~25 + 12 us of event hand-overs), no side streams for the trailing updates and the inversion's
    // stages, no events -- at these sizes running them beside the chain hides less than the hand-overs cost: N = 300 0.28 -> 0.24 ms,
    // 600 0.43 -> 0.36, 1000 0.59 -> 0.53, 1500 0.85 -> 0.80, 1800 1.07 -> 1.04; level at 16 blocks, 6 % slower at 20
    // (profiles/r06_small_update_streams.txt).  (Only the chain on the caller's stream and the rest beside it: slower from 5
    // blocks on -- the CU-masked side streams synchronise with a caller that is the device's null stream.)
    static const int one_stream_lab = (int)sr_lab_env("SR_FACT_ALL_ON_CALLER", -1);      // (lab build: A/B)
    const int one_stream_nb = one_stream_lab >= 0 ? one_stream_lab : SR_FACT_ONE_STREAM_MAX_NB;
    // Tile flow (round 6; sr_flow.hip): the whole Cholesky as ONE resident kernel of GEMM workgroups on the caller's stream plus
    // one resident diagonal-block workgroup per output on a stream of its own; the inversion behind it, on the caller's stream.
    // One batch of outputs only (the counters are zeroed once per update).
    static const int flow_lab = (int)sr_lab_env("SR_FACT_FLOW", -1);                     // (lab build: 1 = wherever it can run)
    const bool want_flow = h->fact_pipe == 3 || (h->fact_pipe == 0 && (flow_lab == 1 || (nb >= SR_FLOW_MIN_NB && nb <= SR_FLOW_MAX_NB)));
    bool flow = want_flow && regime == 1 && nb >= 3 && n_par >= h->n_out;
    if (flow && g_flow_rest.load() > 0) { --g_flow_rest; flow = false; }
    if (flow) {
        SR_F(flow_server_stream(h->device, s0, &srv, &finv));
        if (!srv) { flow = false; finv = nullptr; }
    }
    const bool own_streams = !flow && !(nb <= 2 && P >= nb) && !(nb <= one_stream_nb && h->fact_pipe <= 0);

    hipStream_t sc = s0, sb = s0, si = nullptr;
    if (own_streams) {
        if (!h->fact_fork) SR_FH(hipEventCreateWithFlags(&h->fact_fork, hipEventDisableTiming));
        SR_FH(hipEventRecord(h->fact_fork, s0));
        SR_F(ensure_fact_streams(h, regime));
        sc = h->fact_stream; sb = h->bulk_stream; si = h->inv_stream;
        SR_FH(hipStreamWaitEvent(sc, h->fact_fork, 0));
    }
    lap("streams");
    static const bool no_early_inv = sr_lab_on("SR_FACT_NO_EARLY_INV");
    // Pipelined chain (round 6; chain-bound sizes): see the block behind `if (pipe)` below.
    const bool pipe = own_streams && regime == 1 && (h->fact_pipe == 1 || h->fact_pipe == 2) && !g_pipe_broken.load() && h->diag_stream && h->row_stream &&
                      nb >= 3;
    const bool two = h->fact_pipe == 2;
    hipStream_t sd = nullptr, sr = nullptr;
    unsigned *fl_status = nullptr, *fl_c = nullptr, *fl_d = nullptr, *fl_r = nullptr;
    if (pipe) {
        if (h->fact_flags_nb < nb) {
            (void)device_sync();
            dev_free(h->fact_flags);
            h->fact_flags = nullptr; h->fact_flags_nb = 0;
            SR_F(dev_alloc(&h->fact_flags, (size_t)(4 + 3 * nb)));
            SR_F(dev_zero(h->fact_flags, sizeof(unsigned) * (size_t)(4 + 3 * nb)));
            h->fact_flags_nb = nb; h->fact_epoch = 0;
        }
        fl_status = h->fact_flags; fl_c = h->fact_flags + 4; fl_d = fl_c + h->fact_flags_nb; fl_r = fl_d + h->fact_flags_nb;
        sd = two ? sc : h->diag_stream; sr = h->row_stream;
        if (!two) SR_FH(hipStreamWaitEvent(sd, h->fact_fork, 0));
        SR_FH(hipStreamWaitEvent(sr, h->fact_fork, 0));
    }
    h->last_fact_pipe = pipe ? 1 : (flow ? 4 : 0);
    unsigned flow_ep = 0;
    // a wait of the tile flow that lasts longer gives up.  It grows with the size of the update: the FIRST update on fresh
    // scratch memory of many gigabytes can stand still for seconds (page-table work inside the first kernels: 4.0 - 4.8 s for
    // the first update at N = 50000 against 2.67), and a wait that gives up keeps the process on launches from then on
    // (one of ten updates at N = 14000 did, once, with the fixed quarter of a second)
    const double flow_timeout_s = SR_FLOW_TIMEOUT_S + 1e-12 * (double)h->n_out * (double)Np * (double)Np * (double)Np;
    static const int flow_band_lab = (int)sr_lab_env("SR_FLOW_BAND", -1), flow_acq = (int)sr_lab_env("SR_FLOW_ACQ", 0);
    const int flow_band = flow_band_lab >= 0 ? flow_band_lab : SR_FLOW_BAND;
    static const int flow_panel_lab = (int)sr_lab_env("SR_FLOW_PANEL", 0);
    const int flow_panel = flow_panel_lab > 0 ? flow_panel_lab : (h->fact_panel > 0 ? h->fact_panel : sr_flow_panel(nb));
    if (flow) {
        const long words = SR_FLOW_HDR + (long)h->n_out * sr_flow_words(nb);
        if (h->flow_words < words) {
            (void)device_sync();
            dev_free(h->flow_flags);
            h->flow_flags = nullptr; h->flow_words = 0;
            SR_F(dev_alloc(&h->flow_flags, (size_t)words));
            SR_F(dev_zero(h->flow_flags, sizeof(unsigned) * (size_t)words));
            h->flow_words = words; h->flow_epoch = 0;
        }
        flow_ep = ++h->flow_epoch;
        if (!h->flow_segs || h->flow_segs_key[0] != nb || h->flow_segs_key[1] != flow_band || h->flow_segs_key[2] != flow_panel) {
            std::vector<sr_flow_seg> segs((size_t)nb + 1);
            h->flow_total = sr_flow_plan(nb, flow_band, flow_panel, segs.data(), &h->flow_total_far, &h->flow_total_upd, &h->flow_total_m);
            (void)device_sync();
            if (h->flow_segs) (void)hipFree(h->flow_segs);
            h->flow_segs = nullptr;
            SR_FH(hipMalloc(&h->flow_segs, sizeof(sr_flow_seg) * ((size_t)nb + 1)));
            SR_FH(hipMemcpy(h->flow_segs, segs.data(), sizeof(sr_flow_seg) * ((size_t)nb + 1), hipMemcpyHostToDevice));
            h->flow_segs_key[0] = nb; h->flow_segs_key[1] = flow_band; h->flow_segs_key[2] = flow_panel;
        }
    }

    // Outputs are processed in rounds of n_par as a BATCH: one chain of launches, every kernel works on the n_par
    // problems at once (grid dimension = output; operands `per` resp. NN doubles apart).  Round 2 ran one chain of
    // launches per output on streams of their own; the chains got in each other's way -- a diagonal-block kernel needs
    // a CU to itself (136 KB of LDS) and waited for the other chain's GEMM workgroups to drain, every GEMM of one chain
    // slowed the other's: N = 5000, one output alone 4.0 ms, two outputs 5.6 ms.
    // Buffers per output in flight:
    //   U  = Gram matrix, updated in place; its diagonal blocks end up factored; its strict lower block triangle
    //        is scratch of the inversion (Y).
    //   W  = strict upper block triangle: the FINAL off-diagonal block rows of the factor (the block row solve
    //        writes them here, out of place: with the 64 x 64 tile two workgroups share the 128 rows of a column
    //        block, so an in-place solve would race); diagonal + lower: U^-T.  Every block of W is written before
    //        it is read: no memset.
    //   Wt = U^-1 (diagonal blocks from the diagonal-block kernel, upper blocks from the inversion; the strict
    //        lower triangle is zero from allocation and never written).
    for (int d0 = 0; d0 < h->n_out; d0 += n_par) {
        const int nd = std::min(n_par, h->n_out - d0);
        double* U = ws;                                   // batch member b: + b * per
        double* W = U + NN;
        double* Wt = h->Wt + (size_t)d0 * NN;             // batch member b: + b * NN
        const long sP = (long)per, sN = (long)NN;
        const sr_batch b_ppp{nd, sP, sP, sP, 0};          // all operands in the scratch
        const sr_batch b_diag{nd, sP, sN, sP, 0};         // A = U, wt = Wt, w = W
        const sr_batch b_solve{nd, sN, sP, sP, 0};        // A = Wt (U_kk^-1), B = U, C = W
        const sr_batch b_inv1{nd, sP, sP, sP, 0};         // inversion, product 1: (W, W) -> U
        const sr_batch b_inv2{nd, sN, sP, sP, sN};        // inversion, product 2: (Wt, U) -> W, Wt
        int n_bulk = 0;                                   // bulk updates issued so far (event ping-pong)
        // Early inversion (chain-bound sizes): once the chain has passed the middle block, the root's LEFT subtree of the
        // inversion and the root's first product need nothing the Cholesky still writes (factor rows above the middle,
        // their diagonal-block inverses; results go to W / Wt inside the left half and to U's unused lower-left block).
        // They run on a low-priority stream of their own beside the second half of the chain, which leaves most of the
        // chip idle -- 5/8 of the inversion's flops.  Afterwards: right subtree, root's second product.
        const int root_mid = nb / 2;
        const bool early_inv = regime == 1 && si != nullptr && nb >= 8 && !no_early_inv;
        bool early_done = false;
        // Staged inversion (round 6: more than one stage).  A node [lo, hi) of the halving tree needs nothing the Cholesky
        // still writes once the chain has passed hi (both products), its FIRST product once the chain has passed its middle.
        // Stages are launched on the inversion stream when the chain passes the middles of the rightmost spine of the tree
        // -- nb / 2 (round 3: the root's left subtree and first product), then 3 nb / 4, 7 nb / 8, .. -- children before
        // parents; what is left runs behind the chain.  inv_done[l]: jobs of level l that are complete (a prefix: ascending
        // block ranges); inv_p1[l]: the first product of the job behind them is launched too.
        std::vector<int> inv_done(h->inv_levels.size(), 0), inv_p1(h->inv_levels.size(), 0);
        auto inv_stage = [&](int X, hipStream_t st, bool all) -> int {
            for (size_t li = 0; li < h->inv_levels.size(); ++li) {       // deepest level first
                const sr_gp::inv_level& lv = h->inv_levels[li];
                const int a = inv_done[li];
                int b = a;
                while (b < lv.count && (all || lv.hi[b] <= X)) ++b;
                const int a1 = a + inv_p1[li];                           // first job whose first product is still to come
                const bool half = b < lv.count && lv.mid[b] <= X && !(b == a && inv_p1[li]);   // job b: left child complete
                const int b1 = half ? b + 1 : ((b == a && inv_p1[li]) ? a1 : b);
                sr_prof_scope ps(&h->prof, SR_K_TRINV, st);
                if (b1 > a1) {
                    long tl = 0;
                    for (int j = a1; j < b1; ++j) tl += lv.tl[j];
                    const int rc1 = sr_launch_gemm_tn_jobs(W, W, U, nullptr, Np, h->inv_jobs + lv.off1 + a1, b1 - a1, lv.maxM, lv.maxN, tl,
                                                           1.0, 2, st, &b_inv1);
                    if (rc1 != SR_OK) return rc1;
                }
                if (b > a) {
                    long tl = 0;
                    for (int j = a; j < b; ++j) tl += lv.tl[j];
                    const int rc2 = sr_launch_gemm_tn_jobs(Wt, U, W, Wt, Np, h->inv_jobs + lv.off2 + a, b - a, lv.maxM, lv.maxN, tl, -1.0, 3,
                                                           st, &b_inv2);
                    if (rc2 != SR_OK) return rc2;
                }
                if (b > a) inv_p1[li] = half ? 1 : 0;
                else if (half) inv_p1[li] = 1;
                inv_done[li] = b;
            }
            return SR_OK;
        };
        int inv_trigger = root_mid;                       // next point of the chain at which a stage is launched
        // further stages while at least this many blocks remain -- up to 24 blocks only: beyond, whatever runs beside the chain
        // costs it more than the shorter tail saves (same box, stages / one stage: N = 2000 1.142 / 1.156 ms, 3000 1.85 / 1.88,
        // 5000 4.70 / 4.41, 10000 24.0 / 23.7; profiles/r06_fact_pipeline.txt)
        static const int inv_rest_lab = (int)sr_lab_env("SR_FACT_INV_REST", 0);      // (lab build: 1000 = one stage everywhere)
        const int inv_min_rest = inv_rest_lab > 0 ? inv_rest_lab : (nb <= 24 ? 3 : 1000);
        (void)root_mid;
        if (flow) {
            // the diagonal-block workgroups first -- they must be resident when the workers fill the chip; the caller's stream
            // is drained so that they do not wait for their go longer than the Gram kernel takes
            SR_FH(hipStreamSynchronize(s0));
            // (test hook: an epoch nobody publishes -- they leave after their time-out, the workers' waits run into theirs, the
            //  status word is raised and the update is repeated by launches)
            const unsigned srv_ep = g_test_flow_fail.load() > 0 ? flow_ep + 0x40000000u : flow_ep;
            if (g_test_flow_fail.load() > 0) --g_test_flow_fail;
            SR_F(sr_launch_flow_diag_server(U, Np, Wt, W, Np, nb, flow_panel, info_dev + d0, h->flow_flags, srv_ep, flow_timeout_s,
                                            flow_timeout_s, srv, &b_diag));
            SR_FH(hipMemsetAsync(h->flow_flags + SR_FLOW_STATUS, 0,
                                 sizeof(unsigned) * (size_t)(SR_FLOW_HDR - SR_FLOW_STATUS + (long)h->n_out * sr_flow_words(nb)), sc));
        }
        {
            sr_prof_scope ps(&h->prof, SR_K_GRAM, sc);
            if (h->general)
                SR_F(sr_launch_gram_general(h->Z, h->kp + (size_t)d0 * SR_KP(h->D), 0.0, h->noise + d0, U, h->N, Np, h->D, sc,
                                            nd, sP));
            else
                SR_F(sr_launch_gram(h->Z, h->ls + (size_t)d0 * h->D, 0.0, 0.0, h->sf2 + d0, h->noise + d0, U, h->N, Np, h->D,
                                    sc, nd, sP));
        }
        // --- Cholesky K = U^T U, right-looking in panels of P blocks with look-ahead: after a panel is factored
        // (critical stream, per block: update of its row by the panel's rows above, diagonal block, block row solve),
        // the trailing update is split -- the rows of the NEXT panel on the critical stream, so that its factorisation can start at
        // once, everything behind them on the bulk stream, which may lag one panel behind.
        // Panel boundaries: equal panels of P blocks.  (Round 4, N = 50000: panels that START narrow and double up to P -- 8,
        // 16, 32, 48, .. -- so that the first panel's chain, which has no trailing update to run beside, is short: 65.9 /
        // 65.7 / 65.8 TF with a first panel of 8 / 4 / 16 blocks against 66.0 with equal panels.  The first chain is not idle
        // time: 53 of its 60 ms are the in-panel products, 13 % of the Cholesky's flops, and the narrow panels' trailing
        // updates with their shorter K lose what the hidden chain gains.)
        std::vector<int> pb;
        for (int p = 0; p < nb; p += P) pb.push_back(p);
        pb.push_back(nb);
        if (flow) {
            const long total = h->flow_total;
            static const int flow_prio = (int)sr_lab_env("SR_FLOW_PRIO", 1);
            static const int flow_keep_lab = (int)sr_lab_env("SR_FLOW_KEEP", -1), flow_exit_lab = (int)sr_lab_env("SR_FLOW_EXIT_PCT", -1);
            const int flow_keep = flow_keep_lab >= 0 ? flow_keep_lab : SR_FLOW_KEEP_WGS;
            const int flow_exit_row = (finv && nb >= 8) ? nb * (flow_exit_lab >= 0 ? flow_exit_lab : (nb <= 64 ? SR_FLOW_EXIT_PCT : SR_FLOW_EXIT_PCT_BIG)) / 100 : nb;
            const sr_flow_params fp{U, W, Wt, sP, sN, Np, nb, nd, flow_band, flow_panel, total, h->flow_total_far, h->flow_total_upd, h->flow_total_m, flow_prio, flow_keep, flow_exit_row, (const sr_flow_seg*)h->flow_segs,
                                    h->flow_flags, flow_ep, (unsigned long long)(flow_timeout_s * 1e8), flow_acq};
            if (h->ncu == 0) {
                int cus = 0;
                SR_FH(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device));
                h->ncu = cus;
            }
            // two workgroups per CU (72 KB of LDS each), none on the CUs the diagonal-block workgroups hold
            const long wgs = std::min<long>((total + h->flow_total_far + h->flow_total_upd) * nd, 2L * std::max(1, h->ncu - nd));
            {
                sr_prof_scope ps(&h->prof, SR_K_GEMM, sc);
                SR_F(sr_launch_flow_workers(fp, (int)wgs, sc));
            }
            // the inversion's stages beside the flow: what lies above block row X needs nothing the Cholesky still writes once
            // the chain has passed X (inv_stage) -- at X = nb / 2, then halfway to the end again while three blocks remain;
            // each stage behind a gate kernel that waits for the flow's counters on the device
            static const bool flow_no_inv = sr_lab_on("SR_FLOW_NO_EARLY_INV");
            if (finv && nb >= 8 && !flow_no_inv) {
                SR_F(ensure_fact_events(h));
                for (int X = nb / 2; X < nb;) {
                    SR_F(sr_launch_flow_gate(h->flow_flags, flow_ep, nd, nb, X, flow_timeout_s, finv));
                    SR_F(inv_stage(X, finv, false));
                    const int nxt = (X + nb) / 2;
                    if (nb - nxt < 3 || nxt <= X) break;
                    X = nxt;
                }
                SR_FH(hipEventRecord(h->ev_inv[1], finv));
                early_done = true;
            }
        } else if (pipe) {
            // ---- The same factorisation with the block step CUT at its dependencies and dealt to three streams:
            //   critical (sc):  Sc(k) = the ONE block U[k][k+1] of the block row solve, then Ud(k+1) = the update of the next
            //                   diagonal block -- all the next diagonal block needs;
            //   diagonal (sd):  D(k+1), the diagonal-block kernel, as soon as Ud(k+1) is through;
            //   row (sr):       Sr(k) = the rest of the block row solve, Ur(k+1) = the rest of the next row's update, BESIDE
            //                   D(k+1) (24 us in which the plain chain ran one workgroup per output and nothing else);
            //                   at a panel boundary the look-ahead rows and the hand-over to the bulk stream.
            // Step k of the plain chain was D + S + U one after the other (~55 us alone, ~100 beside the trailing update);
            // here it is max(D + Sc + Ud + two hand-overs, Sr + Ur).  Hand-overs: counters c[k] (diagonal block k updated),
            // d[k] (factored), r[k] (row k updated; everything older on the row stream done), published and awaited by
            // sr_fact_handover_kernel -- 2.7 us each, where an event pair between two hardware queues costs 12 - 13
            // (scripts/xqueue_handoff.hip, profiles/r06_xqueue_handoff.txt) and lost the overlap in round 2.
            // Same tiles, same k order as the plain chain: the same numbers.
            const unsigned ep = ++h->fact_epoch;
            const double tmo = 2.0;
            auto hand = [&](hipStream_t st, unsigned* set, const unsigned* w0, const unsigned* w1) {
                return sr_launch_fact_handover(set, ep, w0, ep, w1, ep, fl_status, tmo, st);
            };
            int pi = 0;                                    // panel of block kb
            bool early_pending = false;
            for (int kb = 0; kb < nb; ++kb) {
                while (kb >= pb[pi + 1]) ++pi;
                const int p0 = pb[pi], p1 = pb[pi + 1];
                const size_t dg = (size_t)kb * SR_NB * Np + (size_t)kb * SR_NB;
                const int ncols = Np - (kb + 1) * SR_NB;
                // ---- diagonal stream: D(kb) once its block is updated (block 0: once the Gram matrix is there)
                // (two-stream form, fact_pipe == 2: D(kb) stays on the critical stream behind Ud(kb); only the rest of the
                //  rows runs beside it -- one hardware queue and two hand-overs per step fewer)
                if (!two) SR_F(hand(sd, kb > 0 ? fl_d + kb - 1 : nullptr, fl_c + kb, nullptr));
                {
                    sr_prof_scope ps(&h->prof, SR_K_POTRF, sd);
                    SR_F(sr_launch_potrf_diag(U, Np, Wt + dg, W + dg, Np, kb, info_dev + d0, sd, 0, &b_diag));
                }
                // ---- critical stream: publishes c[kb] (Gram / Ud(kb) are in front of it), waits for D(kb) and row kb
                if (two) SR_F(hand(sc, fl_d + kb, kb > 0 ? fl_r + kb : nullptr, nullptr));
                else SR_F(hand(sc, fl_c + kb, fl_d + kb, kb > 0 ? fl_r + kb : nullptr));
                if (early_pending) {
                    // every factor row above kb is final, its diagonal blocks inverted (r[kb] stands for the rest of row
                    // kb - 1's solve as well): the stage of the inversion that lies above it
                    early_pending = false; early_done = true;
                    SR_FH(hipEventRecord(h->ev_inv[0], sc));
                    SR_FH(hipStreamWaitEvent(si, h->ev_inv[0], 0));
                    SR_F(inv_stage(kb, si, false));
                    SR_FH(hipEventRecord(h->ev_inv[1], si));
                    inv_trigger = nb + 1;                              // (the prototype keeps to one stage)
                }
                // ---- row stream: publishes r[kb] (Ur(kb) is in front of it), waits for D(kb)
                SR_F(hand(sr, kb > 0 ? fl_r + kb : nullptr, fl_d + kb, nullptr));
                if (ncols <= 0) break;                     // last block: factored, nothing to its right
                const int nxt = kb + 1;                    // the row the updates below make ready
                const bool boundary = nxt == p1;           // ... is the first of the next panel: look-ahead of the panel [p0, p1)
                const size_t dg1 = (size_t)nxt * SR_NB * Np + (size_t)nxt * SR_NB;
                const int right = ncols - SR_NB;           // columns right of block nxt
                // operands of the update of row nxt: the factor rows [p0, nxt) of this panel at columns >= nxt
                const double* Ua = W + (size_t)p0 * SR_NB * Np + (size_t)nxt * SR_NB;
                const int Ku = (nxt - p0) * SR_NB;
                {
                    sr_prof_scope ps(&h->prof, SR_K_GEMM, sc);
                    SR_F(sr_launch_gemm_tn(Wt + dg, Np, U + dg + SR_NB, Np, W + dg + SR_NB, Np, SR_NB, SR_NB, SR_NB, 1.0, 0.0, 0, sc, 1,
                                           &b_solve));                                   // Sc(kb)
                }
                if (two) SR_F(hand(sc, fl_c + nxt, nullptr, nullptr));      // (here c[nxt] says: U[kb][nxt] is there)
                // the previous trailing update wrote the look-ahead rows too: it has to be through (here and on the row stream)
                if (boundary && n_bulk > 0) SR_FH(hipStreamWaitEvent(sc, h->ev_bulk[(n_bulk - 1) & 1], 0));
                {
                    sr_prof_scope ps(&h->prof, SR_K_GEMM, sc);
                    SR_F(sr_launch_gemm_tn_upper(Ua, Np, Ua, Np, U + dg1, Np, SR_NB, SR_NB, Ku, -1.0, 1.0, sc, 1, -1, &b_ppp));   // Ud(nxt)
                }
                if (right > 0) {
                    sr_prof_scope ps(&h->prof, SR_K_GEMM, sr);
                    SR_F(sr_launch_gemm_tn(Wt + dg, Np, U + dg + 2 * SR_NB, Np, W + dg + 2 * SR_NB, Np, SR_NB, right, SR_NB, 1.0, 0.0, 0,
                                           sr, 1, &b_solve));                            // Sr(kb)
                }
                // the updates of row nxt read U[kb][nxt] (Sc): c[nxt] is published behind Ud(nxt), i.e. behind Sc(kb)
                if (right > 0 || boundary) SR_F(hand(sr, nullptr, fl_c + nxt, nullptr));
                if (boundary && n_bulk > 0) SR_FH(hipStreamWaitEvent(sr, h->ev_bulk[(n_bulk - 1) & 1], 0));
                if (right > 0) {
                    sr_prof_scope ps(&h->prof, SR_K_GEMM, sr);
                    SR_F(sr_launch_gemm_tn(Ua, Np, Ua + SR_NB, Np, U + dg1 + SR_NB, Np, SR_NB, right, Ku, -1.0, 1.0, 0, sr, 1,
                                           &b_ppp));                                     // Ur(nxt)
                }
                if (boundary) {
                    const int next_w = pi + 2 < (int)pb.size() ? pb[pi + 2] - p1 : 0;
                    const int rest = Np - p1 * SR_NB;
                    const int la = std::min(next_w * SR_NB, rest);
                    const int bulk = rest - la;
                    if (la > SR_NB) {                      // the other look-ahead rows (behind Ur(nxt): same stream)
                        sr_prof_scope ps(&h->prof, SR_K_GEMM, sr);
                        SR_F(sr_launch_gemm_tn_upper(Ua + SR_NB, Np, Ua + SR_NB, Np, U + dg1 + (size_t)SR_NB * Np + SR_NB, Np, la - SR_NB,
                                                     rest - SR_NB, Ku, -1.0, 1.0, sr, 1, -1, &b_ppp));
                    }
                    if (bulk > 0) {                        // the trailing update behind them, on the bulk stream
                        SR_FH(hipEventRecord(h->ev_panel[pi & 1], sr));
                        SR_FH(hipStreamWaitEvent(sb, h->ev_panel[pi & 1], 0));
                        {
                            sr_prof_scope ps(&h->prof, SR_K_GEMM, sb);
                            SR_F(sr_launch_gemm_tn_upper(Ua + la, Np, Ua + la, Np, U + dg1 + (size_t)la * Np + la, Np, bulk, bulk, Ku, -1.0, 1.0,
                                                         sb, 0, -1, &b_ppp));
                        }
                        SR_FH(hipEventRecord(h->ev_bulk[n_bulk & 1], sb));
                        ++n_bulk;
                    }
                    if (early_inv && !early_done && p1 >= inv_trigger && p1 < nb) early_pending = true;
                }
            }
            // the last hand-over of the critical stream has waited for d[nb-1] and r[nb-1]; the diagonal stream still owes
            // the publication of d[nb-1]
            if (!two) SR_F(hand(sd, fl_d + nb - 1, nullptr, nullptr));
        } else
        for (int pi = 0; pi + 1 < (int)pb.size(); ++pi) {
            const int p0 = pb[pi], p1 = pb[pi + 1];
            const int next_w = pi + 2 < (int)pb.size() ? pb[pi + 2] - p1 : 0;      // blocks of the next panel
            for (int kb = p0; kb < p1; ++kb) {
                const size_t dg = (size_t)kb * SR_NB * Np + (size_t)kb * SR_NB;
                const int ncols = Np - (kb + 1) * SR_NB;
                // left-looking INSIDE the panel: block row kb takes the updates of the panel's rows above it in
                // one product with K = (kb - p0) * 128, right before it is needed -- each row block of the
                // panel is read-modify-written once (eager rank-128 updates of all remaining panel rows
                // touched it up to P - 1 times with K = 128, where prologue and epilogue dominate a tile).
                // (Round 2 ran the diagonal block of the big sizes on a stream of its own beside the rest of its row's
                // update; with the block down from 62 - 180 us to 35 the two event hand-offs per block cost more than the
                // overlap bought: N = 50000 2.616 -> 2.594 s, N = 20000 193 -> 189 ms without it.)
                const double* Upr = W + (size_t)p0 * SR_NB * Np + (size_t)kb * SR_NB;       // rows p0..kb-1, cols >= kb
                const int Kin = (kb - p0) * SR_NB;
                if (kb > p0) {
                    sr_prof_scope ps(&h->prof, SR_K_GEMM, sc);
                    SR_F(sr_launch_gemm_tn_upper(Upr, Np, Upr, Np, U + dg, Np, SR_NB, Np - kb * SR_NB, Kin, -1.0, 1.0, sc, 1, -1,
                                                 &b_ppp));
                }
                {
                    sr_prof_scope ps(&h->prof, SR_K_POTRF, sc);
                    SR_F(sr_launch_potrf_diag(U, Np, Wt + dg, W + dg, Np, kb, info_dev + d0, sc, 0, &b_diag));
                }
                if (ncols > 0) {
                    const double* Arow = U + dg + SR_NB;          // updated Gram rows right of the block
                    double* Urow = W + dg + SR_NB;                // factor rows U[kb][cols right of the block]
                    sr_prof_scope ps(&h->prof, SR_K_GEMM, sc);
                    // U_k,: = U_kk^-T A_k,:   (A operand = U_kk^-1, k-major)
                    SR_F(sr_launch_gemm_tn(Wt + dg, Np, Arow, Np, Urow, Np, SR_NB, ncols, SR_NB, 1.0, 0.0, 0, sc, 1, &b_solve));
                }
            }
            if (early_inv && p1 >= inv_trigger && p1 < nb) {
                // every factor row above p1 is final (and its diagonal block inverted): what of the inversion lies above it
                if (early_done) SR_FH(hipStreamWaitEvent(si, h->ev_inv[1], 0));      // (stages in order; same stream anyway)
                early_done = true;
                SR_FH(hipEventRecord(h->ev_inv[0], sc));
                SR_FH(hipStreamWaitEvent(si, h->ev_inv[0], 0));
                SR_F(inv_stage(p1, si, false));
                SR_FH(hipEventRecord(h->ev_inv[1], si));
                while (inv_trigger <= p1) {
                    const int nxt_t = (inv_trigger + nb) / 2;
                    inv_trigger = (nb - nxt_t >= inv_min_rest && nxt_t > inv_trigger) ? nxt_t : nb + 1;
                }
            }
            const int rest = Np - p1 * SR_NB;
            if (rest <= 0) continue;
            const int Kp = (p1 - p0) * SR_NB;
            const double* Upan = W + (size_t)p0 * SR_NB * Np + (size_t)p1 * SR_NB;    // factor rows of the panel
            double* Cnext = U + (size_t)p1 * SR_NB * Np + (size_t)p1 * SR_NB;
            const int la = std::min(next_w * SR_NB, rest);    // rows of the next panel
            const int bulk = rest - la;
            // regime 1 (chain-bound sizes): the bulk update starts only AFTER the look-ahead rows are done -- started
            // together, its workgroups fill the CUs first and the look-ahead (on the critical path) takes 60 - 70 us
            // instead of 20; regime 2: when the panel's rows are final
            const bool bulk_after_la = (regime == 1);
            const bool side = sb != sc;                   // (everything on one stream: no events)
            if (bulk > 0 && !bulk_after_la && side) SR_FH(hipEventRecord(h->ev_panel[pi & 1], sc));
            // the previous bulk update wrote the look-ahead rows too: it has to be through
            if (n_bulk > 0 && side) SR_FH(hipStreamWaitEvent(sc, h->ev_bulk[(n_bulk - 1) & 1], 0));
            {
                sr_prof_scope ps(&h->prof, SR_K_GEMM, sc);
                SR_F(sr_launch_gemm_tn_upper(Upan, Np, Upan, Np, Cnext, Np, la, rest, Kp, -1.0, 1.0, sc, 1, -1, &b_ppp));
            }
            if (bulk > 0) {
                // regime 2: the CUs the masked bulk stream leaves to the chain (3 % of the chip) idle for most of a long
                // trailing update.  Where the update is SR_FACT_FREE_RATIO times longer than a chain that has to wait for its
                // CUs (a diagonal block then takes milliseconds instead of 0.26, a block step ~1.8 ms on average), it runs on the
                // unmasked stream and takes the whole chip: the first three of the nine panels at N = 50000, 66.8 -> 67.2 TF
                // (ratio 1: 67.2, 4: 66.8, 0.5 and below: 66.9 - 66.7 -- the chain's tail is exposed).
                hipStream_t sbp = sb;
                if (regime == 2 && si != nullptr) {
                    const double t_bulk = (double)nd * (double)bulk * (double)bulk * (double)Kp / 65e12;      // s
                    const double t_chain = (double)next_w * 1.8e-3;
                    static const double free_ratio = sr_lab_envf("SR_FACT_FREE_RATIO", SR_FACT_FREE_RATIO);
                    if (free_ratio > 0.0 && t_bulk > free_ratio * t_chain) sbp = si;
                }
                if (bulk_after_la && side) SR_FH(hipEventRecord(h->ev_panel[pi & 1], sc));
                if (side) SR_FH(hipStreamWaitEvent(sbp, h->ev_panel[pi & 1], 0));
                // (two bulk streams: the previous trailing update may have run on the other one)
                if (regime == 2 && n_bulk > 0) SR_FH(hipStreamWaitEvent(sbp, h->ev_bulk[(n_bulk - 1) & 1], 0));
                {
                    sr_prof_scope ps(&h->prof, SR_K_GEMM, sbp);
                    SR_F(sr_launch_gemm_tn_upper(Upan + la, Np, Upan + la, Np, Cnext + (size_t)la * Np + la, Np,
                                                 bulk, bulk, Kp, -1.0, 1.0, sbp, 0, -1, &b_ppp));
                }
                if (side) SR_FH(hipEventRecord(h->ev_bulk[n_bulk & 1], sbp));
                ++n_bulk;
            }
        }
        if (n_bulk > 0 && sb != sc) SR_FH(hipStreamWaitEvent(sc, h->ev_bulk[(n_bulk - 1) & 1], 0));
        // --- W = U^-T (lower) and Wt = U^-1 (upper) by recursive halving of the block range, level by level
        // (ensure_inv_jobs).  The diagonal blocks of W / Wt were written by sr_potrf_diag_kernel.
        if (early_done) SR_FH(hipStreamWaitEvent(sc, h->ev_inv[1], 0));
        SR_F(inv_stage(nb, sc, true));                    // whatever the stages have left
        // alpha = Wt (W y)   (v behind W in the scratch)
        SR_F(sr_launch_trmv(W, Np, h->yT + (size_t)d0 * Np, W + NN, Np, 1, sc, nd, sP, Np, sP));
        SR_F(sr_launch_trmv(Wt, Np, W + NN, h->alpha + (size_t)d0 * Np, Np, 0, sc, nd, sN, sP, Np));
    }
    if (own_streams) {
        SR_FH(hipEventRecord(h->fact_join, sc));
        SR_FH(hipStreamWaitEvent(s0, h->fact_join, 0));
    }
    static const bool trace = sr_lab_on("SR_FACT_TRACE");
    const auto t_enq = std::chrono::steady_clock::now();
    std::vector<int> info_h(h->n_out, 0);
    SR_FH(hipMemcpyAsync(info_h.data(), info_dev, sizeof(int) * h->n_out, hipMemcpyDeviceToHost, s0));
    // (the tile flow's status word travels with it: a blocking copy of its own behind the synchronisation cost 0.1 - 0.4 ms)
    unsigned flow_status = 0;
    if (flow) SR_FH(hipMemcpyAsync(&flow_status, h->flow_flags + SR_FLOW_STATUS, sizeof(unsigned), hipMemcpyDeviceToHost, s0));
    // (no log determinant here: a launch and a copy on the critical path of every refit -- 0.3 % at N = 5000 -- for a number
    //  only the exploration loop asks for, and there the appends keep the host copy current)
    h->logdet_valid = 0;
    SR_FH(hipStreamSynchronize(s0));
    if (trace) {
        const auto t_end = std::chrono::steady_clock::now();
        fprintf(stderr, "sr_gp_factorize: Np=%d enqueue %.3f ms, total %.3f ms\n", Np,
                std::chrono::duration<double, std::milli>(t_enq - t_begin).count(),
                std::chrono::duration<double, std::milli>(t_end - t_begin).count());
    }
    unsigned pipe_status = 0;
    if (pipe && hipMemcpy(&pipe_status, fl_status, sizeof(unsigned), hipMemcpyDeviceToHost) != hipSuccess) pipe_status = 1;
    if (flow) pipe_status = flow_status;
    cleanup();
#undef SR_F
#undef SR_FH
    if (pipe_status != 0) {
        // a hand-over gave up: whatever was computed behind it is void.  Once more, on the plain chain.
        if (flow) {
            const long len = g_flow_rest_len.load();
            g_flow_rest.store(len);
            g_flow_rest_len.store(std::min<long>(4096, 2 * len));
        } else {
            g_pipe_broken.store(true);
            (void)hipMemset(h->fact_flags, 0, sizeof(unsigned) * (size_t)(4 + 3 * h->fact_flags_nb));
            h->fact_epoch = 0;
        }
        return sr_gp_factorize(h, stream, info);
    }
    int bad = 0;
    for (int d = 0; d < h->n_out; ++d) {
        if (info_h[d] > 0) info_h[d] = std::max(1, info_h[d] - (h->Np - h->N));   // padded -> training index
        if (info) info[d] = info_h[d];
        if (info_h[d] != 0 && !bad) bad = d + 1;
    }
    if (bad) {
        h->factorized = 0; h->logdet_valid = 0;
        sr_set_error("Cholesky breakdown: output %d, pivot %d not positive", bad - 1, info_h[bad - 1]);
        return SR_ENOTPD;
    }
    h->factorized = 1;
    return SR_OK;
}

extern "C" int sr_gp_set_fact_pipeline(sr_gp_t h, int on) {
    SR_CHECK(h != nullptr && on >= -1 && on <= 3, SR_EINVAL, "sr_gp_set_fact_pipeline: bad argument");
    h->fact_pipe = on;
    return SR_OK;
}

extern "C" int sr_gp_fact_pipelined(sr_gp_t h) { return h ? h->last_fact_pipe : 0; }

extern "C" int sr_test_flow_fail(int n) {
    // n > 0: the next n tile-flow updates of this process fail on the device (see sr_gp_factorize); n = 0: forget that one did
    // (a failed flow rests for 16, 32, .. updates of the process)
    SR_CHECK(n >= 0, SR_EINVAL, "sr_test_flow_fail: bad argument");
    g_test_flow_fail.store(n);
    if (n == 0) { g_flow_rest.store(0); g_flow_rest_len.store(16); }
    return SR_OK;
}

extern "C" int sr_test_flow_plan(int nb, int band, int panel, int* segs, long* totals) {
    SR_CHECK(nb >= 1 && nb <= 4096 && band >= 0 && panel >= 1 && segs && totals, SR_EINVAL, "sr_test_flow_plan: bad argument");
    std::vector<sr_flow_seg> sg((size_t)nb + 1);
    totals[0] = sr_flow_plan(nb, band, panel, sg.data(), &totals[1], &totals[2], &totals[3]);
    for (int i = 0; i <= nb; ++i) {
        segs[4 * i + 0] = sg[i].start_c; segs[4 * i + 1] = sg[i].start_f; segs[4 * i + 2] = sg[i].start_b; segs[4 * i + 3] = sg[i].start_m;
    }
    return SR_OK;
}

extern "C" int sr_gp_flow_stats(sr_gp_t h, unsigned* out, int cap) {
    SR_CHECK(h != nullptr && out != nullptr, SR_EINVAL, "sr_gp_flow_stats: NULL argument");
    SR_CHECK(h->last_fact_pipe == 4 && h->flow_flags, SR_ESTATE, "sr_gp_flow_stats: the last model update was not a tile flow");
    SR_DEVICE(h->device);
    const int nb = h->Np / SR_NB;
    const int need = 24 + h->n_out * nb;
    SR_CHECK(cap >= need, SR_EINVAL, "sr_gp_flow_stats: %d words needed", need);
    SR_HIP(hipMemcpy(out, h->flow_flags + SR_FLOW_STATS, 24 * sizeof(unsigned), hipMemcpyDeviceToHost));
    for (int d = 0; d < h->n_out; ++d)
        SR_HIP(hipMemcpy(out + 24 + d * nb, h->flow_flags + SR_FLOW_HDR + (long)d * sr_flow_words(nb) + nb + 9L * nb * nb,
                         nb * sizeof(unsigned), hipMemcpyDeviceToHost));
    return need;
}

extern "C" int sr_gp_set_fact_panel(sr_gp_t h, int panel) {
    SR_CHECK(h != nullptr && panel >= 0 && panel <= 64, SR_EINVAL, "sr_gp_set_fact_panel: bad argument");
    h->fact_panel = panel;
    return SR_OK;
}

extern "C" int sr_gp_mll(sr_gp_t h, double* nll, double* grad, void* stream) {
    SR_CHECK(h && nll && grad, SR_EINVAL, "sr_gp_mll: NULL argument");
    SR_CHECK(h->factorized, SR_ESTATE, "sr_gp_mll: model not factorized");
    SR_CHECK(h->general, SR_ESTATE, "sr_gp_mll: set the data with sr_gp_set_data_general (packed parameters)");
    hipStream_t s = (hipStream_t)stream;
    SR_DEVICE(h->device);
    const int Np = h->Np, D = h->D;
    const size_t NN = (size_t)Np * Np;
    const int ng = SR_KP(D);                        // [v, c0, s[D], a[D], b[D], noise]
    double *W = nullptr, *Kinv = nullptr, *partial = nullptr, *ld = nullptr;
    int rc;
    if ((rc = dev_alloc(&W, NN)) || (rc = dev_alloc(&Kinv, NN)) || (rc = dev_alloc(&partial, (size_t)sr_mll_ws(h->N))) ||
        (rc = dev_alloc(&ld, (size_t)h->n_out))) {
        dev_free(W); dev_free(Kinv); dev_free(partial); dev_free(ld);
        return rc;
    }
    rc = sr_launch_logdet(h->Wt, Np, h->n_out, ld, s);
    for (int d = 0; d < h->n_out && rc == SR_OK; ++d) {
        // K_y^-1 = U^-1 U^-T = sum_k W[k][i] W[k][j]  (W = Wt^T, k-major)
        rc = sr_launch_transpose(h->Wt + (size_t)d * NN, W, Np, s);
        if (rc == SR_OK) rc = sr_launch_gemm_tn(W, Np, W, Np, Kinv, Np, Np, Np, Np, 1.0, 0.0, 0, s);
        if (rc == SR_OK)
            rc = sr_launch_mll(Kinv, Np, h->N, h->alpha + (size_t)d * Np, h->yT + (size_t)d * Np, h->Z,
                               h->kp + (size_t)d * SR_KP(D), D, ld + d, partial, nll + d, grad + (size_t)d * ng, s);
    }
    const hipError_t e = hipStreamSynchronize(s);
    dev_free(W); dev_free(Kinv); dev_free(partial); dev_free(ld);
    if (rc != SR_OK) return rc;
    SR_HIP(e);
    return SR_OK;
}

// host copy of the same numbers if the last model update left one (SR_ESTATE otherwise: call sr_gp_logdet)
extern "C" int sr_gp_logdet_cached(sr_gp_t h, double* logdet_host) {
    SR_CHECK(h && logdet_host, SR_EINVAL, "sr_gp_logdet_cached: NULL argument");
    SR_CHECK(h->factorized, SR_ESTATE, "sr_gp_logdet_cached: model not factorized");
    if (!h->logdet_valid || (int)h->logdet_host.size() != h->n_out) {
        sr_set_error("sr_gp_logdet_cached: no host copy");
        return SR_ESTATE;
    }
    for (int d = 0; d < h->n_out; ++d) logdet_host[d] = h->logdet_host[d];
    return SR_OK;
}

extern "C" int sr_gp_logdet(sr_gp_t h, double* logdet, void* stream) {
    SR_CHECK(h && logdet, SR_EINVAL, "sr_gp_logdet: NULL argument");
    SR_CHECK(h->factorized, SR_ESTATE, "sr_gp_logdet: model not factorized");
    SR_DEVICE(h->device);
    return sr_launch_logdet(h->Wt, h->Np, h->n_out, logdet, (hipStream_t)stream);
}

extern "C" int sr_gp_inv_k(sr_gp_t h, int d, double* inv_k, void* stream) {
    SR_CHECK(h && inv_k, SR_EINVAL, "sr_gp_inv_k: NULL argument");
    SR_CHECK(h->factorized, SR_ESTATE, "sr_gp_inv_k: model not factorized");
    SR_CHECK(d >= 0 && d < h->n_out, SR_EINVAL, "sr_gp_inv_k: d=%d out of range", d);
    hipStream_t s = (hipStream_t)stream;
    SR_DEVICE(h->device);
    const int Np = h->Np;
    const size_t NN = (size_t)Np * Np;
    double *W = nullptr, *out = nullptr;
    int rc;
    if ((rc = dev_alloc(&W, NN)) || (rc = dev_alloc(&out, NN))) { dev_free(W); dev_free(out); return rc; }
    // K^-1 = U^-1 U^-T = sum_k W[k][i] W[k][j]  (W = Wt^T, k-major)
    rc = sr_launch_transpose(h->Wt + (size_t)d * NN, W, Np, s);
    if (rc == SR_OK) rc = sr_launch_gemm_tn(W, Np, W, Np, out, Np, Np, Np, Np, 1.0, 0.0, 0, s);
    hipError_t e = hipSuccess;
    if (rc == SR_OK)
        e = hipMemcpy2DAsync(inv_k, sizeof(double) * h->N, out + (size_t)(Np - h->N) * Np + (Np - h->N),
                             sizeof(double) * Np, sizeof(double) * h->N, h->N, hipMemcpyDeviceToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    dev_free(W); dev_free(out);
    if (rc != SR_OK) return rc;
    SR_HIP(e);
    return SR_OK;
}

extern "C" int sr_test_gemm_tn(int device, const double* A, long lda, const double* B, long ldb,
                               double* C, long ldc, int M, int N, int K, double alpha, double beta,
                               int mode, void* stream) {
    SR_CHECK(A && B && C, SR_EINVAL, "sr_test_gemm_tn: NULL argument");
    SR_DEVICE(device);
    return sr_launch_gemm_tn(A, lda, B, ldb, C, ldc, M, N, K, alpha, beta, mode, (hipStream_t)stream);
}

extern "C" int sr_test_gemm_tn_upper(int device, const double* A, long lda, const double* B, long ldb, double* C,
                                     long ldc, int M, int N, int K, double alpha, double beta, int order, void* stream) {
    SR_CHECK(A && B && C, SR_EINVAL, "sr_test_gemm_tn_upper: NULL argument");
    SR_DEVICE(device);
    return sr_launch_gemm_tn_upper(A, lda, B, ldb, C, ldc, M, N, K, alpha, beta, (hipStream_t)stream, 0, order);
}

extern "C" int sr_test_potrf_diag(int device, double* A, long lda, double* wt, double* w, long ldw, int* info,
                                  int skip, void* stream) {
    SR_CHECK(A && wt && w && info, SR_EINVAL, "sr_test_potrf_diag: NULL argument");
    SR_DEVICE(device);
    return sr_launch_potrf_diag(A, lda, wt, w, ldw, 0, info, (hipStream_t)stream, skip);
}

