// sr_capi.hip -- the extern "C" boundary declared in include/safereach.h.
// Host-side orchestration only: owns the handle, its persistent state (Z, alpha, Wt, hyper-
// parameters) and the per-chunk workspace; enqueues the kernels of sr_factor / sr_predict /
// sr_ellipsoid on the caller's stream.
#include "sr_mfma_tile.h"
#include <atomic>
#include <chrono>
#include <mutex>
#include <thread>
#include <vector>
#include <algorithm>

static thread_local char g_err[1024] = "";

void sr_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

struct sr_gp {
    int device = 0, N = 0, Np = 0, D = 0, n_out = 0;
    // persistent device state
    double *Z = nullptr, *yT = nullptr, *ls = nullptr, *sf2 = nullptr, *noise = nullptr,
           *alpha = nullptr, *Wt = nullptr;
    double* kp = nullptr;     // general kernel family: n_out x SR_KP(D) packed parameters (else NULL)
    // GP input transform of the reachability / moment entry points: x_gp = Tz x (Tz n_xin x n_s), NULL = identity
    double* Tz = nullptr; int n_xin = 0;
    double *tz_x = nullptr, *tz_jac = nullptr; long tz_cap = 0;   // transformed inputs / chain-ruled Jacobians (per chunk)
    // persistent multi-step kernel (sr_small.hip K0c): exchange buffer, per group ticket + epoch + done counter (all of
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
    double* splitk_vt = nullptr; long splitk_cap = 0;   // split-K partial tiles (grow-only)
    int balanced = 1;                                   // few query tiles: balanced shares (K2b) or chunks (K2k): A/B switch
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
    int var_variant = 2;     // 0: register-staged tiles, 1: LDS-DMA (global_load_lds) tiles, 2: 1 + diagonal blocks
                             // without their structural zeros (default; 69.8 -> 70.05 TF at C2')
    // factorisation: the outputs are independent problems -- below SR_FACT_PAR_BYTES of scratch each gets its own
    // HIP stream (the small-grid kernels of a modest model then overlap) and the scratch stays with the handle
    double* fact_ws = nullptr; size_t fact_cap = 0;      // n_par x (U, W: Np^2 each, v: Np)
    // row append of few points (m <= 16): scratch and a second U^-1 buffer the new factor is assembled into
    // (kept while the padded size does not change: appends then allocate nothing big)
    double* app_ws = nullptr; size_t app_cap = 0;
    double* Wt_alt = nullptr; size_t wt_alt_cap = 0;
    int wt_alt_off = -1;     // front padding of the (complete, well-formed) factor Wt_alt last held; -1 unknown
    // small appends allocate nothing while the padded size stays: Z has room for z_cap points, yT / alpha ping-pong
    long z_cap = 0; double *yT_alt = nullptr, *alpha_alt = nullptr; int vec_alt_np = 0;
    std::vector<double> sf2_host, noise_host;    // host copies of sf2 / noise (filled on first use after set_data)
    // up to SR_FACT_SLOTS outputs are factorised at once as a BATCH (every launch covers all of them).  Streams:
    // CRITICAL (diagonal blocks, panel rows, look-ahead rows, late inversion), BULK (the trailing update behind the
    // look-ahead rows) and INVERSION (the early part of the triangular inversion); the caller's stream waits for them
    hipStream_t fact_stream = nullptr, bulk_stream = nullptr, inv_stream = nullptr;
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
    struct inv_level { int off1, off2, count, maxM, maxN; long tiles; int depth, n_left; long tiles_left; };   // jobs of the root's left subtree first
    std::vector<inv_level> inv_levels;
    sr_prof prof;
};
#define SR_FACT_PAR_BYTES ((size_t)8 << 30)
#define SR_FACT_RESERVED_CUS 32     // CU-mask bits the bulk streams of the factorisation leave out (1 CU per shader engine)

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

static inline long round_up(long v, long m) { return (v + m - 1) / m * m; }

// ---- device memory with a block cache ---------------------------------------------------------------------------
// A model that grows (update_model with new points: every call a new padded size) or is replaced (a new handle per
// refit) used to take its big buffers from hipMalloc every time -- and the first touch of a fresh allocation is what
// costs: zeroing 1.7 GB of new memory took 24 ms, a refit after a change of N 48 ms against 5.7 ms in place.  Blocks of
// >= 1 MB are therefore rounded up to a size class (steps of 1/8 of the power of two below the size: <= 12.5 % over) and
// on release kept for the next request of their class (<= an eighth of device memory in all, oldest out first).  The
// contents of a block are unspecified, as hipMalloc's are: every buffer that must start from zeros is zeroed by its owner,
// and the whole GPU suite passes with new buffers filled with NaN patterns (SR_GUARD=1 SR_POISON=1).
// sr_release_cached_memory() hands everything back to the driver; so does a failed hipMalloc,
// once, before it is reported.
struct sr_block { void* p; size_t bytes; int device; unsigned long long stamp; };
struct sr_block_cache {
    std::mutex m;
    std::vector<sr_block> idle;                    // cached blocks
    std::vector<sr_block> live;                    // big blocks handed out (p -> class size)
    size_t idle_bytes = 0, cap_bytes = 0;
    unsigned long long clock = 0;
};
static sr_block_cache g_blocks;
#define SR_CACHE_MIN ((size_t)1 << 20)

static size_t sr_size_class(size_t bytes) {
    size_t pw = (size_t)1 << 20;
    while ((pw << 1) <= bytes) pw <<= 1;
    const size_t step = pw / 8;
    return (bytes + step - 1) / step * step;
}
static void sr_cache_drop_locked(size_t keep_bytes) {                  // oldest first until at most keep_bytes stay
    while (g_blocks.idle_bytes > keep_bytes && !g_blocks.idle.empty()) {
        size_t o = 0;
        for (size_t i = 1; i < g_blocks.idle.size(); ++i)
            if (g_blocks.idle[i].stamp < g_blocks.idle[o].stamp) o = i;
        const sr_block b = g_blocks.idle[o];
        g_blocks.idle.erase(g_blocks.idle.begin() + o);
        g_blocks.idle_bytes -= b.bytes;
        sr_dev_guard guard(b.device);
        (void)hipFree(b.p);
    }
}
// SR_GUARD=1 (diagnostics): every allocation ends at the end of its own 2 MiB-granular hipMalloc, so that a read or write
// past a buffer leaves the mapping (a GPU memory fault) instead of landing silently in a neighbour.  With SR_POISON=1 as
// well the new buffer is filled with NaN bit patterns instead of zeros: code that relies on fresh memory being zero shows
static std::vector<std::pair<void*, void*>> g_guard_map;      // user pointer -> base
static int dev_alloc_bytes(void** p, size_t bytes) {
    *p = nullptr;
    if (bytes == 0) return SR_OK;
    static const bool guard = getenv("SR_GUARD") != nullptr;
    if (guard) {
        const size_t gran = (size_t)2 << 20, al = 16;
        const size_t need = (bytes + al - 1) / al * al;
        const size_t tot = (need + gran - 1) / gran * gran;
        void* base = nullptr;
        SR_HIP(hipMalloc(&base, tot));
        static const bool poison = getenv("SR_POISON") != nullptr;      // SR_POISON=1: new buffers hold NaN, not zeros
        SR_HIP(hipMemset(base, poison ? 0xFF : 0, tot));
        SR_HIP(hipStreamSynchronize(nullptr));
        *p = (char*)base + (tot - need);
        std::lock_guard<std::mutex> lk(g_blocks.m);
        g_guard_map.push_back({*p, base});
        return SR_OK;
    }
    if (bytes < SR_CACHE_MIN) { SR_HIP(hipMalloc(p, bytes)); return SR_OK; }
    int device = 0;
    SR_HIP(hipGetDevice(&device));
    const size_t cls = sr_size_class(bytes);
    std::lock_guard<std::mutex> lk(g_blocks.m);
    for (size_t i = 0; i < g_blocks.idle.size(); ++i)
        if (g_blocks.idle[i].device == device && g_blocks.idle[i].bytes == cls) {
            sr_block b = g_blocks.idle[i];
            g_blocks.idle.erase(g_blocks.idle.begin() + i);
            g_blocks.idle_bytes -= b.bytes;
            g_blocks.live.push_back(b);
            *p = b.p;
            return SR_OK;
        }
    size_t got = cls;                                // what the block really holds (its class, unless memory is short)
    hipError_t e = hipMalloc(p, cls);
    if (e != hipSuccess) {                           // out of memory: everything cached goes back first, then the exact size
        (void)hipGetLastError();
        sr_cache_drop_locked(0);
        e = hipMalloc(p, cls);
        if (e != hipSuccess) { (void)hipGetLastError(); got = bytes; e = hipMalloc(p, bytes); }
        if (e != hipSuccess) {
            sr_set_error("hipMalloc of %zu bytes -> %s", bytes, hipGetErrorString(e));
            (void)hipGetLastError();
            *p = nullptr;
            return SR_EHIP;
        }
    }
    g_blocks.live.push_back({*p, got, device, 0});
    return SR_OK;
}
template <typename T>
static int dev_alloc(T** p, size_t count) { return dev_alloc_bytes((void**)p, count * sizeof(T)); }
static void dev_free(void* p) {
    if (!p) return;
    std::lock_guard<std::mutex> lk(g_blocks.m);
    for (size_t i = 0; i < g_guard_map.size(); ++i)
        if (g_guard_map[i].first == p) {
            void* base = g_guard_map[i].second;
            g_guard_map.erase(g_guard_map.begin() + i);
            (void)hipFree(base);
            return;
        }
    for (size_t i = 0; i < g_blocks.live.size(); ++i)
        if (g_blocks.live[i].p == p) {
            sr_block b = g_blocks.live[i];
            g_blocks.live.erase(g_blocks.live.begin() + i);
            if (g_blocks.cap_bytes == 0) {
                size_t mem_free = 0, mem_total = 0;
                (void)hipMemGetInfo(&mem_free, &mem_total);
                g_blocks.cap_bytes = mem_total / 8;
            }
            static const bool no_cache = getenv("SR_NO_BLOCK_CACHE") != nullptr;      // diagnostics: every release goes to the driver
            if (b.bytes > g_blocks.cap_bytes / 2 || no_cache) { (void)hipFree(p); return; }
            (void)hipDeviceSynchronize();              // what hipFree implied: nothing in flight still uses the block
            b.stamp = ++g_blocks.clock;
            g_blocks.idle.push_back(b);
            g_blocks.idle_bytes += b.bytes;
            sr_cache_drop_locked(g_blocks.cap_bytes);
            return;
        }
    for (const sr_block& b : g_blocks.idle)
        if (b.p == p) { fprintf(stderr, "libsafereach: block %p released twice\n", p); return; }
    (void)hipFree(p);
}
extern "C" int sr_release_cached_memory(void) {
    std::lock_guard<std::mutex> lk(g_blocks.m);
    sr_cache_drop_locked(0);
    return SR_OK;
}
// Zero a freshly allocated buffer and WAIT: a memset on the null stream is not ordered with the launches that follow on a
// caller's non-blocking stream (first-use paths only; never inside a stream capture).
static int dev_zero(void* p, size_t bytes) {
    SR_HIP(hipMemset(p, 0, bytes));
    SR_HIP(hipStreamSynchronize(nullptr));
    return SR_OK;
}

extern "C" int sr_version(void) { return 100; }
extern "C" const char* sr_last_error(void) { return g_err; }

extern "C" int sr_device_count(int* n) {
    SR_CHECK(n != nullptr, SR_EINVAL, "sr_device_count: n is NULL");
    *n = 0;
    int c = 0;
    SR_HIP(hipGetDeviceCount(&c));
    *n = c;
    SR_CHECK(c > 0, SR_EHIP, "no HIP device visible");
    return SR_OK;
}

extern "C" int sr_gp_create(sr_gp_t* out, int device, int N, int D, int n_out) {
    SR_CHECK(out != nullptr, SR_EINVAL, "sr_gp_create: handle pointer is NULL");
    *out = nullptr;
    SR_CHECK(N >= 1 && D >= 1 && n_out >= 1, SR_EINVAL, "sr_gp_create: N=%d D=%d n_out=%d", N, D, n_out);
    SR_CHECK(D <= SR_MAX_D, SR_EUNSUPPORTED, "sr_gp_create: D=%d > %d", D, SR_MAX_D);
    SR_CHECK(n_out <= 64, SR_EUNSUPPORTED, "sr_gp_create: n_out=%d > 64", n_out);
    SR_DEVICE(device);
    sr_gp* h = new sr_gp();
    h->device = device; h->N = N; h->D = D; h->n_out = n_out;
    h->Np = (int)round_up(N, SR_NB);
    int rc = SR_OK;
    h->z_cap = h->Np;
    if ((rc = dev_alloc(&h->Z, (size_t)h->Np * D)) || (rc = dev_alloc(&h->yT, (size_t)n_out * h->Np)) ||
        (rc = dev_alloc(&h->ls, (size_t)n_out * D)) || (rc = dev_alloc(&h->sf2, n_out)) ||
        (rc = dev_alloc(&h->noise, n_out)) || (rc = dev_alloc(&h->alpha, (size_t)n_out * h->Np))) {
        sr_gp_destroy(h);
        return rc;
    }
    *out = h;
    return SR_OK;
}

static void free_ws(sr_gp* h) {
    dev_free(h->Ks); dev_free(h->mu_part); dev_free(h->jac_part); dev_free(h->var_part);
    dev_free(h->mu); dev_free(h->var); dev_free(h->jac); dev_free(h->kxx);
    h->Ks = h->mu_part = h->jac_part = h->var_part = h->mu = h->var = h->jac = h->kxx = nullptr;
    h->ws_Tp = 0; h->ws_part = 0;
}

extern "C" int sr_gp_destroy(sr_gp_t h) {
    if (!h) return SR_OK;
    sr_dev_guard guard(h->device);
    (void)hipDeviceSynchronize();
    dev_free(h->Z); dev_free(h->yT); dev_free(h->ls); dev_free(h->sf2); dev_free(h->noise);
    dev_free(h->alpha); dev_free(h->Wt); dev_free(h->kp); dev_free(h->lin_v); dev_free(h->lin_g); dev_free(h->small_vp); dev_free(h->splitk_vt); dev_free(h->splitk_part);
    dev_free(h->stream_vp); dev_free(h->stream_tickets);
    dev_free(h->Tz); dev_free(h->tz_x); dev_free(h->tz_jac);
    dev_free(h->chain_xch); dev_free(h->chain_tickets); dev_free(h->chain_done); dev_free(h->call_ticket);
    if (h->chain_status_host) (void)hipHostFree(h->chain_status_host);
    dev_free(h->yT_alt); dev_free(h->alpha_alt);
    free_ws(h);
    dev_free(h->fact_ws); dev_free(h->app_ws); dev_free(h->Wt_alt);
    for (hipEvent_t e : {h->fact_join, h->ev_panel[0], h->ev_panel[1], h->ev_bulk[0], h->ev_bulk[1], h->ev_inv[0], h->ev_inv[1]})
        if (e) (void)hipEventDestroy(e);
    if (h->fact_fork) (void)hipEventDestroy(h->fact_fork);
    dev_free(h->inv_jobs); dev_free(h->fact_info);
    h->prof.destroy();
    delete h;
    return SR_OK;
}

__global__ void sr_pack_y_kernel(const double* __restrict__ Y, double* __restrict__ yT, int N, int Np,
                                 int n_out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int d = blockIdx.y;
    const int off = Np - N;                   // front padding
    if (i < Np) yT[(long)d * Np + i] = (i >= off) ? Y[(long)(i - off) * n_out + d] : 0.0;
}

extern "C" int sr_gp_set_data(sr_gp_t h, const double* Z, const double* Y, const double* ls,
                              const double* sf2, const double* noise, void* stream) {
    SR_CHECK(h && Z && Y && ls && sf2 && noise, SR_EINVAL, "sr_gp_set_data: NULL argument");
    hipStream_t s = (hipStream_t)stream;
    SR_DEVICE(h->device);
    SR_HIP(hipMemcpyAsync(h->Z, Z, sizeof(double) * h->N * h->D, hipMemcpyDeviceToDevice, s));
    SR_HIP(hipMemcpyAsync(h->ls, ls, sizeof(double) * h->n_out * h->D, hipMemcpyDeviceToDevice, s));
    SR_HIP(hipMemcpyAsync(h->sf2, sf2, sizeof(double) * h->n_out, hipMemcpyDeviceToDevice, s));
    SR_HIP(hipMemcpyAsync(h->noise, noise, sizeof(double) * h->n_out, hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(sr_pack_y_kernel, dim3((h->Np + 255) / 256, h->n_out), dim3(256), 0, s, Y,
                       h->yT, h->N, h->Np, h->n_out);
    SR_HIP(hipGetLastError());
    h->general = 0;
    h->have_data = 1;
    h->factorized = 0; h->logdet_valid = 0;
    h->sf2_host.clear(); h->noise_host.clear();
    return SR_OK;
}

extern "C" int sr_gp_set_data_general(sr_gp_t h, const double* Z, const double* Y, const double* kparams,
                                      const double* noise, void* stream) {
    SR_CHECK(h && Z && Y && kparams && noise, SR_EINVAL, "sr_gp_set_data_general: NULL argument");
    hipStream_t s = (hipStream_t)stream;
    SR_DEVICE(h->device);
    if (!h->kp) SR_TRY(dev_alloc(&h->kp, (size_t)h->n_out * SR_KP(h->D)));
    SR_HIP(hipMemcpyAsync(h->Z, Z, sizeof(double) * h->N * h->D, hipMemcpyDeviceToDevice, s));
    SR_HIP(hipMemcpyAsync(h->kp, kparams, sizeof(double) * h->n_out * SR_KP(h->D), hipMemcpyDeviceToDevice, s));
    SR_HIP(hipMemcpyAsync(h->noise, noise, sizeof(double) * h->n_out, hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(sr_pack_y_kernel, dim3((h->Np + 255) / 256, h->n_out), dim3(256), 0, s, Y,
                       h->yT, h->N, h->Np, h->n_out);
    SR_HIP(hipGetLastError());
    h->general = 1;
    h->sf2_host.clear(); h->noise_host.clear();
    h->have_data = 1;
    h->factorized = 0; h->logdet_valid = 0;
    return SR_OK;
}

extern "C" int sr_gp_dims(sr_gp_t h, int* N, int* D, int* n_out, long* Np) {
    SR_CHECK(h != nullptr, SR_EINVAL, "sr_gp_dims: NULL handle");
    if (N) *N = h->N;
    if (D) *D = h->D;
    if (n_out) *n_out = h->n_out;
    if (Np) *Np = h->Np;
    return SR_OK;
}

extern "C" int sr_gp_padded_n(sr_gp_t h, long* Np) {
    SR_CHECK(h && Np, SR_EINVAL, "sr_gp_padded_n: NULL argument");
    *Np = h->Np;
    return SR_OK;
}

static int ensure_wt(sr_gp* h) {
    if (!h->Wt) {
        SR_TRY(dev_alloc(&h->Wt, (size_t)h->n_out * h->Np * h->Np));
        // the strict lower triangle of U^-1 is never written by the factorisation: zero it once
        SR_TRY(dev_zero(h->Wt, sizeof(double) * h->n_out * h->Np * h->Np));
    }
    return SR_OK;
}

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
    const int root_mid = nb / 2;
    for (int depth = max_depth; depth >= 0; --depth) {
        sr_gp::inv_level lv{};
        lv.depth = depth;
        std::vector<sr_gemm_job> j1, j2;
        for (int side = 0; side < 2; ++side)
        for (const Node& r : nodes) {
            if (r.depth != depth) continue;
            const bool left = depth > 0 && r.hi <= root_mid;          // inside the root's left half
            if (left != (side == 0)) continue;
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
            lv.tiles += (long)(n2 / SR_NB) * (n1 / SR_NB);
            if (left) { ++lv.n_left; lv.tiles_left += (long)(n2 / SR_NB) * (n1 / SR_NB); }
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
    return nb <= 28 ? 2 : (nb <= 44 ? 3 : (nb <= 64 ? 4 : (nb <= 200 ? 8 : (nb <= 300 ? 24 : 48))));
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
struct sr_stream_set { int device, key; hipStream_t fact, bulk, inv; };
static std::mutex g_stream_mutex;
static std::vector<sr_stream_set> g_stream_sets;
static void drop_fact_streams(sr_gp* h) {
    // (the handle only forgets them; work it has in flight on them is waited for)
    for (hipStream_t* st : {&h->fact_stream, &h->bulk_stream, &h->inv_stream})
        if (*st) { (void)hipStreamSynchronize(*st); *st = nullptr; }
    h->fact_regime = 0;
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
    // CUs the bulk streams leave to the critical chain: one per shader engine; two for 36 < Np / 128 <= 52, where the
    // block-row solves and in-panel updates of the chain otherwise queue behind the trailing update's workgroups (50
    // instead of 12 us each) and the trailing update has the slack (measured, reserve 32 / 64: N = 3500 2.79 / 2.80,
    // N = 5000 5.14 / 4.99, N = 6500 8.79 / 8.74, N = 7500 12.17 / 12.45, N = 10000 25.1 / 26.7 ms)
    const int nblk = h->Np / SR_NB;
    const int reserve = regime == 2 ? 8 : ((nblk > 36 && nblk <= 52) ? 2 * SR_FACT_RESERVED_CUS : SR_FACT_RESERVED_CUS);
    const int key = regime * 1000 + reserve;
    if (h->fact_regime != key) drop_fact_streams(h);
    if (!h->fact_stream) {
        std::lock_guard<std::mutex> lk(g_stream_mutex);
        sr_stream_set* set = nullptr;
        for (sr_stream_set& c : g_stream_sets)
            if (c.device == h->device && c.key == key) set = &c;
        if (!set) {
            sr_stream_set c{h->device, key, nullptr, nullptr, nullptr};
            int prio_lo = 0, prio_hi = 0;
            SR_HIP(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
            SR_HIP(hipStreamCreateWithPriority(&c.fact, hipStreamNonBlocking, prio_hi));
            if (can_mask) SR_TRY(make_masked_stream(&c.bulk, ncu, reserve, ncu));
            else SR_HIP(hipStreamCreateWithPriority(&c.bulk, hipStreamNonBlocking, prio_lo));
            if (regime == 1) {
                if (can_mask) SR_TRY(make_masked_stream(&c.inv, ncu, reserve, ncu));
                else SR_HIP(hipStreamCreateWithPriority(&c.inv, hipStreamNonBlocking, prio_lo));
            }
            g_stream_sets.push_back(c);
            set = &g_stream_sets.back();
        }
        h->fact_stream = set->fact; h->bulk_stream = set->bulk; h->inv_stream = set->inv;
    }
    if (!h->fact_join) SR_HIP(hipEventCreateWithFlags(&h->fact_join, hipEventDisableTiming));
    for (int e = 0; e < 2; ++e) {
        if (!h->ev_panel[e]) SR_HIP(hipEventCreateWithFlags(&h->ev_panel[e], hipEventDisableTiming));
        if (!h->ev_bulk[e]) SR_HIP(hipEventCreateWithFlags(&h->ev_bulk[e], hipEventDisableTiming));
        if (!h->ev_inv[e]) SR_HIP(hipEventCreateWithFlags(&h->ev_inv[e], hipEventDisableTiming));
    }
    h->fact_regime = key;
    return SR_OK;
}

extern "C" int sr_gp_factorize(sr_gp_t h, void* stream, int* info) {
    SR_CHECK(h != nullptr, SR_EINVAL, "sr_gp_factorize: NULL handle");
    SR_CHECK(h->have_data, SR_ESTATE, "sr_gp_factorize: call sr_gp_set_data first");
    SR_DEVICE(h->device);
    const auto t_begin = std::chrono::steady_clock::now();
    static const bool trace_laps = getenv("SR_FACT_TRACE") != nullptr;
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
    static const size_t par_cap_env = getenv("SR_FACT_PAR_GB") ? (size_t)atol(getenv("SR_FACT_PAR_GB")) << 30 : 0;
    const size_t par_bytes = par_cap_env ? par_cap_env : std::max<size_t>(SR_FACT_PAR_BYTES, h->mem_total / 3);
    int n_par = (int)std::min<size_t>((size_t)std::min(h->n_out, SR_FACT_SLOTS),
                                      std::max<size_t>(1, par_bytes / (per * sizeof(double))));
    const bool keep = per * n_par * sizeof(double) <= par_bytes;
    double* scratch = nullptr;                           // owned here only when it is not kept in the handle
    int rc = SR_OK;
    hipStream_t s0 = (hipStream_t)stream;
    auto cleanup = [&]() {
        // never return with work in flight on the side streams
        for (hipStream_t st : {h->fact_stream, h->bulk_stream, h->inv_stream}) if (st) (void)hipStreamSynchronize(st);
        dev_free(scratch);
    };
#define SR_F(expr) do { rc = (expr); if (rc != SR_OK) { cleanup(); return rc; } } while (0)
#define SR_FH(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
        sr_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); cleanup(); return SR_EHIP; } } while (0)
    double* ws;
    if (keep) {
        if (h->fact_cap < per * n_par) {
            (void)hipDeviceSynchronize();
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
    if (!h->fact_fork) SR_FH(hipEventCreateWithFlags(&h->fact_fork, hipEventDisableTiming));
    SR_FH(hipEventRecord(h->fact_fork, s0));
    const int regime = nb <= 128 ? 1 : 2;
    SR_F(ensure_fact_streams(h, regime));
    lap("streams");
    static const bool no_early_inv = getenv("SR_FACT_NO_EARLY_INV") != nullptr;
    hipStream_t sc = h->fact_stream, sb = h->bulk_stream, si = h->inv_stream;
    SR_FH(hipStreamWaitEvent(sc, h->fact_fork, 0));

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
        int pi = 0;
        for (int p0 = 0; p0 < nb; p0 += P, ++pi) {
            const int p1 = std::min(nb, p0 + P);
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
            if (early_inv && !early_done && p1 >= root_mid && p1 < nb) {
                // every factor row above the middle is final (and its diagonal block inverted): left subtree + root product 1
                early_done = true;
                SR_FH(hipEventRecord(h->ev_inv[0], sc));
                SR_FH(hipStreamWaitEvent(si, h->ev_inv[0], 0));
                for (const sr_gp::inv_level& lv : h->inv_levels) {
                    sr_prof_scope ps(&h->prof, SR_K_TRINV, si);
                    if (lv.depth == 0) {
                        SR_F(sr_launch_gemm_tn_jobs(W, W, U, nullptr, Np, h->inv_jobs + lv.off1, 1, lv.maxM, lv.maxN, lv.tiles, 1.0, 2,
                                                    si, &b_inv1));
                    } else if (lv.n_left > 0) {
                        SR_F(sr_launch_gemm_tn_jobs(W, W, U, nullptr, Np, h->inv_jobs + lv.off1, lv.n_left, lv.maxM, lv.maxN,
                                                    lv.tiles_left, 1.0, 2, si, &b_inv1));
                        SR_F(sr_launch_gemm_tn_jobs(Wt, U, W, Wt, Np, h->inv_jobs + lv.off2, lv.n_left, lv.maxM, lv.maxN,
                                                    lv.tiles_left, -1.0, 3, si, &b_inv2));
                    }
                }
                SR_FH(hipEventRecord(h->ev_inv[1], si));
            }
            const int rest = Np - p1 * SR_NB;
            if (rest <= 0) continue;
            const int Kp = (p1 - p0) * SR_NB;
            const double* Upan = W + (size_t)p0 * SR_NB * Np + (size_t)p1 * SR_NB;    // factor rows of the panel
            double* Cnext = U + (size_t)p1 * SR_NB * Np + (size_t)p1 * SR_NB;
            const int la = std::min(P * SR_NB, rest);         // rows of the next panel
            const int bulk = rest - la;
            // regime 1 (chain-bound sizes): the bulk update starts only AFTER the look-ahead rows are done -- started
            // together, its workgroups fill the CUs first and the look-ahead (on the critical path) takes 60 - 70 us
            // instead of 20; regime 2: when the panel's rows are final
            const bool bulk_after_la = (regime == 1);
            if (bulk > 0 && !bulk_after_la) SR_FH(hipEventRecord(h->ev_panel[pi & 1], sc));
            // the previous bulk update wrote the look-ahead rows too: it has to be through
            if (n_bulk > 0) SR_FH(hipStreamWaitEvent(sc, h->ev_bulk[(n_bulk - 1) & 1], 0));
            {
                sr_prof_scope ps(&h->prof, SR_K_GEMM, sc);
                SR_F(sr_launch_gemm_tn_upper(Upan, Np, Upan, Np, Cnext, Np, la, rest, Kp, -1.0, 1.0, sc, 1, -1, &b_ppp));
            }
            if (bulk > 0) {
                if (bulk_after_la) SR_FH(hipEventRecord(h->ev_panel[pi & 1], sc));
                SR_FH(hipStreamWaitEvent(sb, h->ev_panel[pi & 1], 0));
                {
                    sr_prof_scope ps(&h->prof, SR_K_GEMM, sb);
                    SR_F(sr_launch_gemm_tn_upper(Upan + la, Np, Upan + la, Np, Cnext + (size_t)la * Np + la, Np,
                                                 bulk, bulk, Kp, -1.0, 1.0, sb, 0, -1, &b_ppp));
                }
                SR_FH(hipEventRecord(h->ev_bulk[n_bulk & 1], sb));
                ++n_bulk;
            }
        }
        if (n_bulk > 0) SR_FH(hipStreamWaitEvent(sc, h->ev_bulk[(n_bulk - 1) & 1], 0));
        // --- W = U^-T (lower) and Wt = U^-1 (upper) by recursive halving of the block range, level by level
        // (ensure_inv_jobs).  The diagonal blocks of W / Wt were written by sr_potrf_diag_kernel.
        if (early_done) SR_FH(hipStreamWaitEvent(sc, h->ev_inv[1], 0));
        for (const sr_gp::inv_level& lv : h->inv_levels) {
            sr_prof_scope ps(&h->prof, SR_K_TRINV, sc);
            const int skip = early_done ? lv.n_left : 0;          // jobs of the left subtree already ran
            const int cnt = lv.count - skip;
            const long tiles = lv.tiles - (early_done ? lv.tiles_left : 0);
            if (cnt > 0 && !(early_done && lv.depth == 0))
                SR_F(sr_launch_gemm_tn_jobs(W, W, U, nullptr, Np, h->inv_jobs + lv.off1 + skip, cnt, lv.maxM, lv.maxN, tiles, 1.0, 2,
                                            sc, &b_inv1));
            if (cnt > 0)
                SR_F(sr_launch_gemm_tn_jobs(Wt, U, W, Wt, Np, h->inv_jobs + lv.off2 + skip, cnt, lv.maxM, lv.maxN, tiles, -1.0, 3,
                                            sc, &b_inv2));
        }
        // alpha = Wt (W y)   (v behind W in the scratch)
        SR_F(sr_launch_trmv(W, Np, h->yT + (size_t)d0 * Np, W + NN, Np, 1, sc, nd, sP, Np, sP));
        SR_F(sr_launch_trmv(Wt, Np, W + NN, h->alpha + (size_t)d0 * Np, Np, 0, sc, nd, sN, sP, Np));
    }
    SR_FH(hipEventRecord(h->fact_join, sc));
    SR_FH(hipStreamWaitEvent(s0, h->fact_join, 0));
    static const bool trace = getenv("SR_FACT_TRACE") != nullptr;
    const auto t_enq = std::chrono::steady_clock::now();
    std::vector<int> info_h(h->n_out, 0);
    SR_FH(hipMemcpyAsync(info_h.data(), info_dev, sizeof(int) * h->n_out, hipMemcpyDeviceToHost, s0));
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
    cleanup();
#undef SR_F
#undef SR_FH
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

extern "C" int sr_gp_set_fact_panel(sr_gp_t h, int panel) {
    SR_CHECK(h != nullptr && panel >= 0 && panel <= 64, SR_EINVAL, "sr_gp_set_fact_panel: bad argument");
    h->fact_panel = panel;
    return SR_OK;
}

extern "C" int sr_gp_export(sr_gp_t h, double* alpha, double* Wt, void* stream) {
    SR_CHECK(h != nullptr, SR_EINVAL, "sr_gp_export: NULL handle");
    SR_CHECK(h->factorized, SR_ESTATE, "sr_gp_export: model not factorized");
    hipStream_t s = (hipStream_t)stream;
    SR_DEVICE(h->device);
    if (alpha)
        SR_HIP(hipMemcpy2DAsync(alpha, sizeof(double) * h->N, h->alpha + (h->Np - h->N),
                                sizeof(double) * h->Np, sizeof(double) * h->N, h->n_out,
                                hipMemcpyDeviceToDevice, s));
    if (Wt)
        SR_HIP(hipMemcpyAsync(Wt, h->Wt, sizeof(double) * h->n_out * h->Np * h->Np,
                              hipMemcpyDeviceToDevice, s));
    return SR_OK;
}

extern "C" int sr_gp_import(sr_gp_t h, const double* alpha, const double* Wt, void* stream) {
    SR_CHECK(h != nullptr, SR_EINVAL, "sr_gp_import: NULL handle");
    SR_CHECK(h->have_data, SR_ESTATE, "sr_gp_import: call sr_gp_set_data first (Z and hyper-parameters)");
    SR_CHECK(alpha && Wt, SR_EINVAL, "sr_gp_import: alpha and Wt are both required");
    hipStream_t s = (hipStream_t)stream;
    SR_DEVICE(h->device);
    SR_TRY(ensure_wt(h));
    SR_HIP(hipMemsetAsync(h->alpha, 0, sizeof(double) * h->n_out * h->Np, s));
    SR_HIP(hipMemcpy2DAsync(h->alpha + (h->Np - h->N), sizeof(double) * h->Np, alpha,
                            sizeof(double) * h->N, sizeof(double) * h->N, h->n_out,
                            hipMemcpyDeviceToDevice, s));
    SR_HIP(hipMemcpyAsync(h->Wt, Wt, sizeof(double) * h->n_out * h->Np * h->Np,
                          hipMemcpyDeviceToDevice, s));
    h->factorized = 1; h->logdet_valid = 0;
    return SR_OK;
}

// ---- packed posterior state for the one-time replication ------------------------------------------
// U^-1 is upper triangular: of the Np^2 doubles sr_gp_export hands out only the N (N + 1) / 2 on and above the
// diagonal of the real (unpadded) rows carry information -- 100 MB instead of 210 MB per output at N = 5000, 10 GB
// instead of 20 GB at N = 50000.  Training row i (0 <= i < N) contributes its N - i entries U^-1[i][i..N-1]; rows
// [row0, row1) are packed back to back, so a replication can travel in bounded pieces through a small staging buffer.
static inline long packed_rows_count(long N, long row0, long row1) {
    return (row1 - row0) * N - (row1 * (row1 - 1) - row0 * (row0 - 1)) / 2;
}

template <bool PACK>
__global__ __launch_bounds__(256) void sr_pack_rows_kernel(double* __restrict__ Wt, double* __restrict__ buf, int N,
                                                           int Np, long row0) {
    const long i = row0 + blockIdx.x;                                    // training row
    const long off = Np - N;
    const long base = (i - row0) * N - (i * (i - 1) - row0 * (row0 - 1)) / 2;   // packed offset of row i's first entry
    double* row = Wt + (i + off) * Np + off + i;
    const int len = N - (int)i;
    for (int c = blockIdx.y * 256 + threadIdx.x; c < len; c += gridDim.y * 256) {
        if (PACK) buf[base + c] = row[c];
        else row[c] = buf[base + c];
    }
}

extern "C" long sr_gp_packed_count(sr_gp_t h, long row0, long row1) {
    if (!h || row0 < 0 || row1 < row0 || row1 > h->N) return -1;
    return packed_rows_count(h->N, row0, row1);
}

extern "C" int sr_gp_export_packed(sr_gp_t h, int d, long row0, long row1, double* buf, void* stream) {
    SR_CHECK(h != nullptr && buf != nullptr, SR_EINVAL, "sr_gp_export_packed: NULL argument");
    SR_CHECK(h->factorized, SR_ESTATE, "sr_gp_export_packed: model not factorized");
    SR_CHECK(d >= 0 && d < h->n_out && row0 >= 0 && row0 <= row1 && row1 <= h->N, SR_EINVAL,
             "sr_gp_export_packed: d=%d rows [%ld, %ld) outside the model (n_out=%d, N=%d)", d, row0, row1, h->n_out, h->N);
    if (row1 == row0) return SR_OK;
    SR_DEVICE(h->device);
    const int gy = std::max(1, std::min(8, (h->N - (int)row0 + 2047) / 2048));
    hipLaunchKernelGGL(sr_pack_rows_kernel<true>, dim3((unsigned)(row1 - row0), gy), dim3(256), 0, (hipStream_t)stream,
                       h->Wt + (size_t)d * h->Np * h->Np, buf, h->N, h->Np, row0);
    SR_HIP(hipGetLastError());
    return SR_OK;
}

// Receiver side: sr_gp_import_begin (alpha, identity padding, structural zeros), then the packed rows of every output
// in any order and in any number of pieces (sr_gp_import_packed), then sr_gp_import_end marks the handle factorized.
extern "C" int sr_gp_import_begin(sr_gp_t h, const double* alpha, void* stream) {
    SR_CHECK(h != nullptr && alpha != nullptr, SR_EINVAL, "sr_gp_import_begin: NULL argument");
    SR_CHECK(h->have_data, SR_ESTATE, "sr_gp_import_begin: call sr_gp_set_data first (Z and hyper-parameters)");
    hipStream_t s = (hipStream_t)stream;
    SR_DEVICE(h->device);
    SR_TRY(ensure_wt(h));             // zero below the diagonal from allocation on; nothing ever writes there
    h->factorized = 0; h->logdet_valid = 0;
    h->import_open = 1;
    SR_HIP(hipMemsetAsync(h->alpha, 0, sizeof(double) * h->n_out * h->Np, s));
    SR_HIP(hipMemcpy2DAsync(h->alpha + (h->Np - h->N), sizeof(double) * h->Np, alpha, sizeof(double) * h->N,
                            sizeof(double) * h->N, h->n_out, hipMemcpyDeviceToDevice, s));
    for (int d = 0; d < h->n_out; ++d)
        SR_TRY(sr_launch_eye_front(h->Wt + (size_t)d * h->Np * h->Np, h->Np, h->Np - h->N, s));
    return SR_OK;
}

extern "C" int sr_gp_import_packed(sr_gp_t h, int d, long row0, long row1, const double* buf, void* stream) {
    SR_CHECK(h != nullptr && buf != nullptr, SR_EINVAL, "sr_gp_import_packed: NULL argument");
    SR_CHECK(h->import_open, SR_ESTATE, "sr_gp_import_packed: call sr_gp_import_begin first");
    SR_CHECK(d >= 0 && d < h->n_out && row0 >= 0 && row0 <= row1 && row1 <= h->N, SR_EINVAL,
             "sr_gp_import_packed: d=%d rows [%ld, %ld) outside the model (n_out=%d, N=%d)", d, row0, row1, h->n_out, h->N);
    if (row1 == row0) return SR_OK;
    SR_DEVICE(h->device);
    const int gy = std::max(1, std::min(8, (h->N - (int)row0 + 2047) / 2048));
    hipLaunchKernelGGL(sr_pack_rows_kernel<false>, dim3((unsigned)(row1 - row0), gy), dim3(256), 0, (hipStream_t)stream,
                       h->Wt + (size_t)d * h->Np * h->Np, const_cast<double*>(buf), h->N, h->Np, row0);
    SR_HIP(hipGetLastError());
    return SR_OK;
}

extern "C" int sr_gp_import_end(sr_gp_t h) {
    SR_CHECK(h != nullptr, SR_EINVAL, "sr_gp_import_end: NULL handle");
    SR_CHECK(h->import_open, SR_ESTATE, "sr_gp_import_end: no import in progress");
    h->import_open = 0;
    h->factorized = 1; h->logdet_valid = 0;
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

// ---------------------------------------------------------------------------------------------
// workspace + the three-kernel GP pass over one chunk
// ---------------------------------------------------------------------------------------------
static int pick_nsplit(const sr_gp* h, long Tp) {
    const long blocks = ((Tp + 255) / 256) * h->n_out;
    long ns = (768 + blocks - 1) / blocks;
    // down to 16 training rows per workgroup (a 16-long exp chain); beyond SR_FINAL_WAVE_T queries
    // sr_finalize_kernel adds the partial sums serially per thread: never more than max(16, Np/128) there
    const long maxs = Tp <= SR_FINAL_WAVE_T ? (long)h->Np / 16
                                            : std::min((long)h->Np / 16, std::max(16L, (long)h->Np / SR_NB));
    if (ns > maxs) ns = maxs;
    if (ns < 1) ns = 1;
    return (int)ns;
}

static int ensure_ws(sr_gp* h, long Tp, int nsplit) {
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
    (void)hipDeviceSynchronize();
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
static int prepare_ws(sr_gp* h, long Tc) {
    const long Tp = round_up(Tc, srt::BN);
    int ns = pick_nsplit(h, Tp);
    if (Tc <= stream_max_t()) ns = std::max(ns, std::max(pick_nsplit(h, srt::BN), 2 * ((h->Np + 255) / 256)));
    return ensure_ws(h, Tp, ns);
}
struct sr_ws_lock {
    sr_gp* h;
    explicit sr_ws_lock(sr_gp* h_) : h(h_) { h->ws_locked = 1; }
    ~sr_ws_lock() { h->ws_locked = 0; }
};

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

static void stream_common(const sr_gp* h, sr_stream_args& a, int ncols, long Tp) {
    a.Wt = h->Wt; a.Ks = h->Ks; a.Vp = h->stream_vp; a.part = h->var_part; a.tickets = h->stream_tickets;
    a.N = h->N; a.Np = h->Np; a.D = h->D; a.n_out = h->n_out; a.k_lo = h->Np - h->N; a.ncols = ncols; a.ncols_pad = ncols;
    a.Tp = Tp;
    a.Z = h->Z; a.alpha = h->alpha; a.ls = h->ls; a.sf2 = h->sf2;
    a.mu_part_w = h->mu_part; a.jac_part_w = h->jac_part;
}

static int stream_predict(sr_gp* h, long Tc, const double* xa, long lda, int na, const double* xb, long ldb, int nb,
                          double* mu, double* var, double* jac, hipStream_t s) {
    const long Tp = srt::BN;
    const int ncb = (h->Np + 255) / 256;
    // 2 .. 4 queries on a moderate model: the one-launch VALU kernel has only ncb (ncb + 1) n_out workgroups of 256
    // threads there and loses to K1 + MFMA kernel + reduce (N = 700: 27 against 21 us; N = 3000: 34 against 40 us)
    const bool mfma_small = Tc >= 2 && Tc <= 4 && h->Np <= 2048;
    const bool fused = !h->general && h->D <= 5 && Tc <= 4 && !mfma_small;
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
    h->last_streamed = 0;
    sr_prof_scope ps(&h->prof, SR_K_VAR, s);
    return sr_launch_stream(a, fused ? 1 : 0, s);
}

static int stream_linearize(sr_gp* h, const double* x, double* mu, double* var, double* jac_mu, double* jac_var,
                            double* hess_mu, hipStream_t s) {
    const long Tp = srt::BN;
    const int ncb = (h->Np + 255) / 256, ncols = 1 + h->D;
    const bool fused = !h->general && h->D <= 3;
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
    h->last_streamed = 0;
    sr_prof_scope ps(&h->prof, SR_K_VAR, s);
    return sr_launch_stream(a, fused ? 2 : 0, s);
}

// GP posterior of Tc queries x = [xa | xb] into (mu, var, jac) in API layout (jac may be NULL).
static int gp_pass(sr_gp* h, long Tc, const double* xa, long lda, int na, const double* xb, long ldb,
                   int nb, double* mu, double* var, double* jac, hipStream_t s) {
    // (ONE query against Np = 384: the one-launch pass is a single workgroup per output that fetches 590 KB of U^-1 on its
    //  own, 18.8 us; the streamed route spreads them over 6 workgroups per output: 12.4 us.  From 4 queries on the two
    //  are level, and 16 queries share one fetch in the one-launch pass.)
    const bool one_streamed = h->Np == 384 && Tc == 1 && !h->general && h->D <= 5;
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
    } else if (h->small_path && h->balanced && sr_var_splitk_wanted(h->Np, Tp, h->n_out) && sr_var_bal_wanted(h->Np, Tp, h->n_out)) {
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
        SR_TRY(sr_launch_var_bal(h->Wt, h->Ks, h->splitk_vt, h->splitk_part, h->N, h->Np, Tp, h->n_out, s));
    } else if (h->small_path && sr_var_splitk_wanted(h->Np, Tp, h->n_out)) {
        // few query tiles: split the K range so that no workgroup serialises a whole row block
        const long need = sr_var_splitk_ws(h->Np, Tp, h->n_out);
        if (need > h->splitk_cap) {
            (void)hipStreamSynchronize(s);
            dev_free(h->splitk_vt);
            h->splitk_vt = nullptr; h->splitk_cap = 0;
            SR_TRY(dev_alloc(&h->splitk_vt, (size_t)need));
            h->splitk_cap = need;
        }
        if (!h->splitk_part) SR_TRY(dev_alloc(&h->splitk_part, (size_t)4 * 1024 * srt::BN));  // wgs <= 1024
        var_part = h->splitk_part;
        nrb = 4 * (h->Np / SR_NB);
        sr_prof_scope ps(&h->prof, SR_K_VAR, s);
        SR_TRY(sr_launch_var_splitk(h->Wt, h->Ks, h->splitk_vt, h->splitk_part, h->N, h->Np, Tp, h->n_out, s));
    } else if (h->small_path && sr_var64_wanted(h->Np, Tp, h->n_out)) {
        // small model, few tiles: 64 x 64 workgroup tiles shorten the critical path of the tiny grid
        if (!h->splitk_part) SR_TRY(dev_alloc(&h->splitk_part, (size_t)4 * 1024 * srt::BN));
        var_part = h->splitk_part;             // n_out * (Np/64) * Tp <= 2 * 256 * 128 * 16 doubles
        nrb = h->Np / 64;
        sr_prof_scope ps(&h->prof, SR_K_VAR, s);
        SR_TRY(sr_launch_var64(h->Wt, h->Ks, h->splitk_part, h->N, h->Np, Tp, h->n_out, s));
    } else {
        sr_prof_scope ps(&h->prof, SR_K_VAR, s);
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
    if (h->small_path == 1 && !h->general && sr_gp_small_wanted(h->Np, SR_SMALL_T, h->D, false)) {
        // small ARD-RBF model: everything in one launch (sr_small.hip, LIN mode)
        sr_kstar_args ka{};
        ka.Z = h->Z; ka.alpha = h->alpha; ka.ls = h->ls; ka.sf2 = h->sf2;
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
    (void)hipDeviceSynchronize();
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
static int gp_pass_states(sr_gp* h, long Tc, const double* p, long ldp, int n_s, const double* kff, long ldkff, int n_u,
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

// The persistent kernel of sr_small.hip (K0c) for a chain of H >= 1 steps, where it applies; *taken says whether it ran.
static int try_chain(sr_gp* h, long T, int H, int mode, const double* p0, const double* q0, const double* k_fb0,
                     const double* k_ff, const double* k_fb, const double* a, const double* b, const double* l_mu,
                     const double* l_sigma, double c_safety, double* p_all, double* q_all, double* gp_var_all,
                     int* n_bad, int n_s, int n_u, hipStream_t s, bool* taken) {
    *taken = false;
    const long nss = (long)n_s * n_s, nus = (long)n_u * n_s;
    // small model, few rollouts: the whole chain in one launch (sr_small.hip K0c).  One launch holds SR_CHAIN_GROUPS
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
    const long chain_max_launches = H <= 2 ? 1 : (T > SR_FUSED_T ? 6 : 2);
    if (h->chain && h->small_path == 1 && !h->force_stream && !h->general && h->n_xin == 0 &&
        chain_launches <= chain_max_launches &&
        sr_chain_supported(h->Np, h->D, n_s, n_u, H)) {
        // the kernel must be able to run at all: at least one workgroup per CU (registers, static + dynamic LDS)
        const int occ_key = ((h->Np * 8 + n_s) * 8 + n_u) * 64 + std::min(H, 63);
        if (h->chain_occ_key != occ_key) {
            h->chain_occ_blocks = 0;
            SR_TRY(sr_chain_blocks_per_cu(h->Np, n_s, n_u, H, &h->chain_occ_blocks));
            h->chain_occ_key = occ_key;
        }
        if (h->chain_occ_blocks < 1) return SR_OK;                       // per-step launches
        if (!h->chain_xch) {
            SR_HIP(hipHostMalloc((void**)&h->chain_status_host, sizeof(int), hipHostMallocMapped));
            *h->chain_status_host = 0;
            SR_HIP(hipHostGetDevicePointer((void**)&h->chain_status_dev, h->chain_status_host, 0));
            SR_TRY(dev_alloc(&h->chain_xch, (size_t)SR_CHAIN_XELS));
            SR_TRY(dev_alloc(&h->chain_tickets, (size_t)2 * SR_CHAIN_GROUPS));
            SR_TRY(dev_alloc(&h->chain_done, (size_t)SR_CHAIN_GROUPS));
            // ON THE CALLER'S STREAM: a memset on the null stream is not ordered with a launch on a non-blocking stream --
            // it wiped tags the first launch had already written (41 MB take 20 us) and that launch timed out
            SR_HIP(hipMemsetAsync(h->chain_xch, 0, sizeof(sr_xel) * (size_t)SR_CHAIN_XELS, s));     // tag 0 = never written
            SR_HIP(hipMemsetAsync(h->chain_tickets, 0, sizeof(unsigned long long) * 2 * SR_CHAIN_GROUPS, s));
            SR_HIP(hipMemsetAsync(h->chain_done, 0, sizeof(unsigned) * SR_CHAIN_GROUPS, s));
        }
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

extern "C" int sr_gp_set_chunk(sr_gp_t h, long chunk) {
    SR_CHECK(h != nullptr && chunk >= 1, SR_EINVAL, "sr_gp_set_chunk: bad argument");
    h->chunk = chunk;
    return SR_OK;
}

extern "C" int sr_gp_set_var_group(sr_gp_t h, int group) {
    SR_CHECK(h != nullptr && group >= 1, SR_EINVAL, "sr_gp_set_var_group: bad argument");
    h->var_group = group;
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
            __builtin_ia32_pause();
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
    if (!(h->small_path == 1 && !h->general && h->n_xin == 0 &&
          sr_gp_small_wanted(h->Np, second_order ? SR_SMALL_T : 1, h->D, false))) {
        sr_set_error("sr_gp_call1: no one-launch posterior for this model (Np=%d, general=%d)", h->Np, h->general);
        return SR_EUNSUPPORTED;
    }
    SR_DEVICE(h->device);
    hipStream_t s = (hipStream_t)stream;
    if (!h->call_ticket) {
        SR_TRY(dev_alloc(&h->call_ticket, 1));
        SR_TRY(dev_zero(h->call_ticket, sizeof(unsigned)));
    }
    double* out = nullptr;
    unsigned long long* flag = nullptr;
    SR_HIP(hipHostGetDevicePointer((void**)&out, out_host, 0));
    SR_HIP(hipHostGetDevicePointer((void**)&flag, flag_host, 0));
    const int n = h->n_out, D = h->D;
    sr_kstar_args ka{};
    ka.Z = h->Z; ka.alpha = h->alpha; ka.ls = h->ls; ka.sf2 = h->sf2;
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

extern "C" int sr_gp_release_scratch(sr_gp_t h) {
    SR_CHECK(h != nullptr, SR_EINVAL, "sr_gp_release_scratch: NULL handle");
    SR_DEVICE(h->device);
    SR_HIP(hipDeviceSynchronize());
    dev_free(h->fact_ws); h->fact_ws = nullptr; h->fact_cap = 0;
    dev_free(h->Wt_alt); h->Wt_alt = nullptr; h->wt_alt_cap = 0; h->wt_alt_off = -1;
    dev_free(h->app_ws); h->app_ws = nullptr; h->app_cap = 0;
    dev_free(h->yT_alt); dev_free(h->alpha_alt); h->yT_alt = h->alpha_alt = nullptr; h->vec_alt_np = 0;
    // the caller wants the memory back: what these releases left in the block cache goes to the driver too
    return sr_release_cached_memory();
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

extern "C" int sr_gp_set_small_path(sr_gp_t h, int on) {
    SR_CHECK(h != nullptr, SR_EINVAL, "sr_gp_set_small_path: NULL handle");
    h->small_path = on & 3;  // 0: plain three-kernel pass only; 1: all latency paths; 2: all but the fused K0
    h->balanced = (on & 4) ? 0 : 1;      // + 4: the chunks of K2k instead of the balanced shares of K2b (A/B measurements)
    return SR_OK;
}

extern "C" int sr_gp_set_var_variant(sr_gp_t h, int variant) {
    SR_CHECK(h != nullptr && (variant >= 0 && variant <= 2), SR_EINVAL, "sr_gp_set_var_variant: bad argument");
    h->var_variant = variant;
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
        SR_HIP(hipDeviceSynchronize());
        SR_TRY(dev_zero(h->chain_xch, sizeof(sr_xel) * (size_t)SR_CHAIN_XELS));
        SR_TRY(dev_zero(h->chain_tickets, sizeof(unsigned long long) * 2 * SR_CHAIN_GROUPS));
        SR_TRY(dev_zero(h->chain_done, sizeof(unsigned) * SR_CHAIN_GROUPS));
    }
    return SR_OK;
}

extern "C" int sr_prof_enable(sr_gp_t h, int on) {
    SR_CHECK(h != nullptr, SR_EINVAL, "sr_prof_enable: NULL handle");
    SR_DEVICE(h->device);
    if (!on) h->prof.resolve();
    h->prof.enabled = on ? 1 : 0;
    return SR_OK;
}
extern "C" int sr_prof_reset(sr_gp_t h) {
    SR_CHECK(h != nullptr, SR_EINVAL, "sr_prof_reset: NULL handle");
    SR_DEVICE(h->device);
    h->prof.resolve();
    for (int i = 0; i < SR_K_COUNT; ++i) { h->prof.ms[i] = 0.0; h->prof.launches[i] = 0; }
    return SR_OK;
}
extern "C" int sr_prof_get(sr_gp_t h, int kernel_id, double* ms_total, long* launches) {
    SR_CHECK(h != nullptr && kernel_id >= 0 && kernel_id < SR_K_COUNT, SR_EINVAL,
             "sr_prof_get: bad argument");
    SR_DEVICE(h->device);
    h->prof.resolve();
    if (ms_total) *ms_total = h->prof.ms[kernel_id];
    if (launches) *launches = h->prof.launches[kernel_id];
    return SR_OK;
}

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
// the queries), everything else as matrix-vector shaped passes -- see sr_factor.hip.  No big allocation while
// the padded size stays the same (U^-1 ping-pongs between two buffers).
static int append_small(sr_gp* h, const double* Znew, const double* Ynew, int m, hipStream_t s, int* info) {
    const int N0 = h->N, Np0 = h->Np, off0 = Np0 - N0, D = h->D, n_out = h->n_out;
    const int N1 = N0 + m, Np1 = (int)round_up(N1, SR_NB), off1 = Np1 - N1, pf = SR_NB - m;
    const size_t NN0 = (size_t)Np0 * Np0, NN1 = (size_t)Np1 * Np1, BB = (size_t)SR_NB * SR_NB;
    // scratch layout
    // (Xt, Y2, G, S, S^-1 once per output: every step below is ONE launch over all outputs)
    const size_t s_xt = (size_t)SR_SMALL_T * Np0, s_y2 = (size_t)Np0 * SR_NB;
    const size_t o_u12 = 0, o_xt = o_u12 + (size_t)n_out * SR_SMALL_T * Np0, o_y2 = o_xt + n_out * s_xt,
                 o_g = o_y2 + n_out * s_y2, o_sb = o_g + n_out * BB, o_inv = o_sb + n_out * BB,
                 o_ld = o_inv + n_out * BB, o_info = o_ld + (size_t)n_out * SR_APPEND1_WGS, need = o_info + (size_t)n_out;
    if (h->app_cap < need) {
        (void)hipDeviceSynchronize();
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
        SR_A(dev_alloc(&yT1, (size_t)n_out * Np1));
        SR_A(dev_alloc(&alpha1, (size_t)n_out * Np1));
    }
    if (reuse_alt) Wt1 = h->Wt_alt;
    else SR_A(dev_alloc(&Wt1, (size_t)n_out * NN1));
    if (!z_inplace) SR_AH(hipMemcpyAsync(Z1, h->Z, sizeof(double) * N0 * D, hipMemcpyDeviceToDevice, s));
    // (in place: rows N0 .. N1-1 of Z are not read by anything below -- the model keeps N = N0 until the commit)
    // one point on a small model: the whole append is ONE launch (sr_append1_small_kernel)
    const bool fused1 = m == 1 && Np0 <= 512 && Np1 <= 640 && h->small_path != 0;
    if (fused1) {
        SR_A(sr_launch_append1_small(h->Wt, h->alpha, h->yT, h->Z, h->ls, h->sf2, h->noise, h->general ? h->kp : nullptr,
                                     Znew, Ynew, Wt1, alpha1, yT1,
                                     Z1 + (size_t)N0 * D, ws + o_ld, info_dev, N0, Np0, Np1, D, n_out, s));
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
    const int nld = n_out * SR_APPEND1_WGS;                       // (the one-launch route leaves partial sums)
    std::vector<double> back(nld + (n_out + 1) / 2, 0.0);         // the log dets, then n_out ints
    SR_AH(hipMemcpyAsync(back.data(), ws + o_ld, sizeof(double) * nld + sizeof(int) * n_out, hipMemcpyDeviceToHost, s));
    SR_AH(hipStreamSynchronize(s));
    std::vector<int> info_h(n_out, 0);
    memcpy(info_h.data(), back.data() + nld, sizeof(int) * n_out);
    for (int d = 0; d < n_out; ++d) {
        double t = back[fused1 ? d * SR_APPEND1_WGS : d];
        for (int y = 1; fused1 && y < SR_APPEND1_WGS; ++y) t += back[d * SR_APPEND1_WGS + y];
        back[d] = t;                                              // (d <= d * SR_APPEND1_WGS: nothing unread is overwritten)
    }
    h->logdet_valid = 0;
#undef SR_A
#undef SR_AH
    int bad = 0;
    for (int d = 0; d < n_out; ++d) {
        if (info_h[d] > 0) info_h[d] = N0 + std::max(1, info_h[d] - pf);
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

extern "C" int sr_gp_append(sr_gp_t h, const double* Znew, const double* Ynew, int m, void* stream, int* info) {
    SR_CHECK(h != nullptr && Znew && Ynew, SR_EINVAL, "sr_gp_append: NULL argument");
    SR_CHECK(h->factorized, SR_ESTATE, "sr_gp_append: model not factorized");
    SR_CHECK(m >= 1 && m <= SR_NB, SR_EINVAL, "sr_gp_append: m=%d outside 1..%d (append in several calls)", m, SR_NB);
    hipStream_t s = (hipStream_t)stream;
    SR_DEVICE(h->device);
    if (m <= SR_SMALL_T) return append_small(h, Znew, Ynew, m, s, info);
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
        (void)hipDeviceSynchronize();
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
    SR_A(dev_alloc(&yT1, (size_t)n_out * Np1));
    SR_A(dev_alloc(&alpha1, (size_t)n_out * Np1));
    if (reuse_alt) Wt1 = h->Wt_alt;
    else SR_A(dev_alloc(&Wt1, (size_t)n_out * NN1));
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
