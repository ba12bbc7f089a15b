// sr_capi_handle.hip -- the extern "C" boundary declared in include/safereach.h, part 1: the handle, its device memory
// (block cache), the training data, export / import of the posterior state (dense and packed), switches and the
// per-kernel timing.  Host-side orchestration only.
#include "sr_handle.h"
using namespace srh;

static thread_local char g_err[1024] = "";

void sr_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}


// ---- device memory with a block cache ---------------------------------------------------------------------------
// A model that grows (update_model with new points: every call a new padded size) or is replaced (a new handle per
// refit) used to take its big buffers from hipMalloc every time -- and the first touch of a fresh allocation is what
// costs: zeroing 1.7 GB of new memory took 24 ms, a refit after a change of N 48 ms against 5.7 ms in place.  Blocks of
// >= 1 MB are therefore rounded up to a size class (steps of 1/8 of the power of two below the size: <= 12.5 % over) and
// on release kept for the next request of their class.  Bounds (PyTorch's caching allocator shares the device and cannot
// reclaim what sits here): per DEVICE at most min(1/8 of its memory, 8 GB) -- SR_BLOCK_CACHE_MB overrides, 0 switches the
// cache off --, oldest out first; a block larger than half the cap, or one that fell back to an exact (non-class) size
// because memory was short, goes straight back to the driver.  The contents of a block are unspecified, as hipMalloc's
// are: every buffer that must start from zeros is zeroed by its owner, and the whole GPU suite passes with new buffers
// filled with NaN patterns (SR_GUARD=1 SR_POISON=1).  sr_release_cached_memory() hands everything back to the driver
// (the Python layer calls it when a torch allocation fails, and retries); so does a failed hipMalloc in here, once,
// before it is reported.
struct sr_block { void* p; size_t bytes; int device; unsigned long long stamp; };
struct sr_block_cache {
    std::mutex m;
    std::vector<sr_block> idle;                    // cached blocks
    std::vector<sr_block> live;                    // big blocks handed out (p -> class size)
    size_t idle_bytes[32] = {0}, cap_bytes[32] = {0};     // per device
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
// oldest first until at most keep_bytes stay on `device` (device < 0: on every device)
static void sr_cache_drop_locked(int device, size_t keep_bytes) {
    for (;;) {
        size_t o = g_blocks.idle.size();
        for (size_t i = 0; i < g_blocks.idle.size(); ++i) {
            const sr_block& c = g_blocks.idle[i];
            if (device >= 0 && c.device != device) continue;
            if (g_blocks.idle_bytes[c.device & 31] <= keep_bytes) continue;
            if (o == g_blocks.idle.size() || c.stamp < g_blocks.idle[o].stamp) o = i;
        }
        if (o == g_blocks.idle.size()) return;
        const sr_block b = g_blocks.idle[o];
        g_blocks.idle.erase(g_blocks.idle.begin() + o);
        g_blocks.idle_bytes[b.device & 31] -= b.bytes;
        sr_dev_guard guard(b.device);
        (void)hipFree(b.p);
    }
}
// SR_GUARD=1 (diagnostics): every allocation ends at the end of its own 2 MiB-granular hipMalloc, so that a read or write
// past a buffer leaves the mapping (a GPU memory fault) instead of landing silently in a neighbour.  With SR_POISON=1 as
// well the new buffer is filled with NaN bit patterns instead of zeros: code that relies on fresh memory being zero shows
static std::vector<std::pair<void*, void*>> g_guard_map;      // user pointer -> base
int srh::dev_alloc_bytes(void** p, size_t bytes) {
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
            g_blocks.idle_bytes[device & 31] -= b.bytes;
            g_blocks.live.push_back(b);
            *p = b.p;
            return SR_OK;
        }
    size_t got = cls;                                // what the block really holds (its class, unless memory is short)
    hipError_t e = hipMalloc(p, cls);
    if (e != hipSuccess) {                           // out of memory: everything cached goes back first, then the exact size
        (void)hipGetLastError();
        sr_cache_drop_locked(-1, 0);
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
void srh::dev_free(void* p) {
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
            size_t& cap = g_blocks.cap_bytes[b.device & 31];
            if (cap == 0) {
                size_t mem_free = 0, mem_total = 0;
                sr_dev_guard guard(b.device);
                (void)hipMemGetInfo(&mem_free, &mem_total);
                const char* env = getenv("SR_BLOCK_CACHE_MB");          // (0: every release goes to the driver)
                cap = env ? ((size_t)atol(env) << 20) + 1 : std::min(mem_total / 8, (size_t)8 << 30);   // (+ 1: "asked" marker)
            }
            const bool no_cache = cap <= 1;
            // (a block that fell back to its exact size when memory was short would never match a request again)
            if (b.bytes > cap / 2 || b.bytes != sr_size_class(b.bytes) || no_cache) {
                sr_dev_guard guard(b.device);
                (void)hipFree(p);
                return;
            }
            {
                sr_dev_guard guard(b.device);
                (void)device_sync();          // what hipFree implied: nothing in flight still uses the block
            }
            b.stamp = ++g_blocks.clock;
            g_blocks.idle.push_back(b);
            g_blocks.idle_bytes[b.device & 31] += b.bytes;
            sr_cache_drop_locked(b.device, cap);
            return;
        }
    for (const sr_block& b : g_blocks.idle)
        if (b.p == p) { fprintf(stderr, "libsafereach: block %p released twice\n", p); return; }
    (void)hipFree(p);
}
extern "C" int sr_release_cached_memory(void) {
    std::lock_guard<std::mutex> lk(g_blocks.m);
    sr_cache_drop_locked(-1, 0);
    return SR_OK;
}
// Zero a freshly allocated buffer and WAIT: a memset on the null stream is not ordered with the launches that follow on a
// caller's non-blocking stream (first-use paths only; never inside a stream capture).
int srh::dev_zero(void* p, size_t bytes) {
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
    if ((rc = dev_alloc(&h->Z, (size_t)h->Np * D)) || (rc = dev_alloc(&h->yT, vec_doubles(n_out, h->Np))) ||
        (rc = dev_alloc(&h->ls, (size_t)n_out * D)) || (rc = dev_alloc(&h->sf2, n_out)) ||
        (rc = dev_alloc(&h->noise, n_out)) || (rc = dev_alloc(&h->alpha, vec_doubles(n_out, h->Np))) ||
        (rc = dev_zero(h->yT, sizeof(double) * vec_doubles(n_out, h->Np))) ||
        (rc = dev_zero(h->alpha, sizeof(double) * vec_doubles(n_out, h->Np)))) {
        sr_gp_destroy(h);
        return rc;
    }
    *out = h;
    return SR_OK;
}

void srh::free_ws(sr_gp* h) {
    dev_free(h->Ks); dev_free(h->mu_part); dev_free(h->jac_part); dev_free(h->var_part);
    dev_free(h->mu); dev_free(h->var); dev_free(h->jac); dev_free(h->kxx);
    h->Ks = h->mu_part = h->jac_part = h->var_part = h->mu = h->var = h->jac = h->kxx = nullptr;
    h->ws_Tp = 0; h->ws_part = 0;
}

extern "C" int sr_gp_destroy(sr_gp_t h) {
    if (!h) return SR_OK;
    sr_dev_guard guard(h->device);
    server_release(h);
    (void)device_sync();
    dev_free(h->Z); dev_free(yT_alloc_of(h)); dev_free(h->ls); dev_free(h->sf2); dev_free(h->noise);
    dev_free(alpha_alloc_of(h)); dev_free(wt_alloc_of(h)); dev_free(h->kp); dev_free(h->lin_v); dev_free(h->lin_g); dev_free(h->small_vp); dev_free(h->splitk_vt); dev_free(h->splitk_part);
    dev_free(h->stream_vp); dev_free(h->stream_tickets); dev_free(h->stream_slots); dev_free(h->stream_tab);
    dev_free(h->Tz); dev_free(h->tz_x); dev_free(h->tz_jac);
    dev_free(h->chain_xch); dev_free(h->chain_tickets); dev_free(h->chain_done); dev_free(h->call_ticket);
    if (h->chain_status_host) (void)hipHostFree(h->chain_status_host);
    dev_free(h->yT_alt); dev_free(h->alpha_alt);
    free_ws(h);
    dev_free(h->fact_ws); dev_free(h->app_ws); dev_free(h->appg_cnt); dev_free(h->Wt_alt); dev_free(h->fact_flags); dev_free(h->flow_flags); if (h->flow_segs) (void)hipFree(h->flow_segs);
    if (h->app_pin) (void)hipHostFree(h->app_pin);
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
    SR_TRY(server_quiesce(h));            // the resident server reads the model: off the device before it changes
    SR_TRY(unslide(h));
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
    SR_TRY(server_quiesce(h));
    SR_TRY(unslide(h));
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

int srh::ensure_wt(sr_gp* h) {
    if (!h->Wt) {
        SR_TRY(dev_alloc(&h->Wt, wt_doubles(h->n_out, h->Np)));
        // the strict lower triangle of U^-1 is never written by the factorisation: zero it once (and the slack behind it)
        SR_TRY(dev_zero(h->Wt, sizeof(double) * wt_doubles(h->n_out, h->Np)));
        h->slack_ok = 1;                 // (alpha and yT carry theirs from sr_gp_create / the append routes on)
    }
    return SR_OK;
}

// The model back in plain buffers: the views are contiguous ranges of their allocations, so each moves with ONE copy into
// a fresh allocation (a copy onto its own allocation would overlap).  Rare: a refit, new data, an import, an append of
// several points or a big batch (whose tile kernels read U^-1 in 16-byte pieces) after in-place one-point appends.
int srh::unslide(sr_gp* h) {
    if (h->slide == 0) return SR_OK;
    sr_dev_guard guard(h->device);
    SR_TRY(server_quiesce(h));           // a resident server of this model reads the views that are about to be freed
    const size_t nw = (size_t)h->n_out * h->Np * h->Np, nv = (size_t)h->n_out * h->Np;
    double *w = nullptr, *a = nullptr, *y = nullptr;
    int rc = SR_OK;
    if ((rc = dev_alloc(&w, wt_doubles(h->n_out, h->Np))) || (rc = dev_alloc(&a, vec_doubles(h->n_out, h->Np))) ||
        (rc = dev_alloc(&y, vec_doubles(h->n_out, h->Np)))) {
        dev_free(w); dev_free(a); dev_free(y);
        return rc;
    }
    hipError_t e = hipMemcpy(w, h->Wt, sizeof(double) * nw, hipMemcpyDeviceToDevice);
    if (e == hipSuccess) e = hipMemset(w + nw, 0, sizeof(double) * (wt_doubles(h->n_out, h->Np) - nw));
    if (e == hipSuccess) e = hipMemcpy(a, h->alpha, sizeof(double) * nv, hipMemcpyDeviceToDevice);
    if (e == hipSuccess) e = hipMemset(a + nv, 0, sizeof(double) * SR_SLIDE_STEPS);
    if (e == hipSuccess) e = hipMemcpy(y, h->yT, sizeof(double) * nv, hipMemcpyDeviceToDevice);
    if (e == hipSuccess) e = hipMemset(y + nv, 0, sizeof(double) * SR_SLIDE_STEPS);
    // the old buffers go back to the block cache below: nothing on ANY stream may still be reading them (a prediction the
    // caller launched asynchronously, say) -- the device-wide wait of the library (resident servers leave first)
    if (e == hipSuccess) e = device_sync();
    if (e != hipSuccess) {
        dev_free(w); dev_free(a); dev_free(y);
        sr_set_error("unslide: %s", hipGetErrorString(e));
        return SR_EHIP;
    }
    dev_free(wt_alloc_of(h)); dev_free(alpha_alloc_of(h)); dev_free(yT_alloc_of(h));
    h->Wt = w; h->alpha = a; h->yT = y;
    h->slide = 0;
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
    SR_TRY(server_quiesce(h));
    SR_TRY(unslide(h));
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
    SR_TRY(server_quiesce(h));
    SR_TRY(unslide(h));
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

extern "C" int sr_gp_release_scratch(sr_gp_t h) {
    SR_CHECK(h != nullptr, SR_EINVAL, "sr_gp_release_scratch: NULL handle");
    SR_DEVICE(h->device);
    SR_HIP(device_sync());
    dev_free(h->fact_ws); h->fact_ws = nullptr; h->fact_cap = 0;
    dev_free(h->Wt_alt); h->Wt_alt = nullptr; h->wt_alt_cap = 0; h->wt_alt_off = -1;
    dev_free(h->app_ws); h->app_ws = nullptr; h->app_cap = 0;
    dev_free(h->yT_alt); dev_free(h->alpha_alt); h->yT_alt = h->alpha_alt = nullptr; h->vec_alt_np = 0;
    // the caller wants the memory back: what these releases left in the block cache goes to the driver too
    return sr_release_cached_memory();
}

extern "C" int sr_gp_set_small_path(sr_gp_t h, int on) {
    SR_CHECK(h != nullptr, SR_EINVAL, "sr_gp_set_small_path: NULL handle");
    h->small_path = on & 3;  // 0: plain three-kernel pass only; 1: all latency paths; 2: all but the fused K0
    return SR_OK;
}

extern "C" int sr_gp_set_var_variant(sr_gp_t h, int variant) {
    SR_CHECK(h != nullptr && (variant == 1 || variant == 3 || variant == 4 || variant == 5), SR_EINVAL, "sr_gp_set_var_variant: 1, 3, 4 or 5");
#ifndef SR_LAB
    if (variant != 4) { sr_set_error("sr_gp_set_var_variant: variant %d exists in the lab build only (make lab)", variant); return SR_EUNSUPPORTED; }
#endif
    h->var_variant = variant;
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

