// sr_capi_server.hip -- the resident single-query server (kernel K0s, sr_server.hip): start / stop / blocking call, and the
// hooks the other entry points use to take it off the device before they write the model or wait for the whole device.
//
// replaces the blocking evaluation inside CasadiSSMEvaluator.eval / JacFun.eval / BackFun.eval
// (/root/reference/safe_exploration/state_space_models.py:278-303, 384-417, 534-562 -> SimpleGPModel.__call__ /
// linearize_predict, ssm_gpy/gaussian_process.py:135-144) where the model has a one-launch posterior.
#include "sr_handle.h"
using namespace srh;

namespace {

std::mutex g_srv_mutex;
std::vector<sr_gp*> g_srv_running;            // handles whose server kernel may be resident (any device)

constexpr size_t MB_WORDS = 16;               // mailbox: 128 bytes (one line); reply: 2 x SR_SERVER_ALIVE words
constexpr size_t REPLY_WORDS = 3 * SR_SERVER_ALIVE;

size_t out_doubles(const sr_gp* h) { return (size_t)2 * h->n_out + (size_t)2 * h->n_out * h->D + (size_t)h->n_out * h->D * h->D; }
int parts_of(const sr_gp* h) { return sr_gp_server_parts(h->Np, h->general != 0); }
int slots_of(const sr_gp* h) { return h->n_out * parts_of(h); }                      // reply words: one per (output, part)
size_t rec_doubles(const sr_gp* h) { return (size_t)SR_SERVER_ALIVE * SR_SERVER_REC; }     // (room for any model of the handle)

// Every kernel identifier (ARD-RBF; mat52 / lin_rbf / lin_mat52 through the general family) up to 512 padded rows, D <= 5.
// Queries are in the GP's INPUT space, as for sr_gp_predict: an input transform set for the reachability entry points
// (sr_gp_set_input_transform) neither concerns nor disturbs the server.
bool servable(const sr_gp* h) {
    return h->factorized && h->small_path == 1 && slots_of(h) <= SR_SERVER_ALIVE && sr_gp_server_supported(h->Np, h->D);
}

// ---- the mailbox line (layout: sr_server_args in sr_common.h) -------------------------------------------------------------
// HOST MEMORY MODEL.  The host side publishes with ordinary stores in program order -- payload words, check word, sequence
// number last -- separated by release fences.  On x86-64 (TSO: stores are not reordered with older stores) the fences
// cost nothing and the device, which validates every fetch of the line against the check word anyway, sees the request
// with its first look after the sequence number lands.  On a weaker host memory model the fences are what orders the
// stores; correctness does not depend on the order either way (a half-written line fails the check and is fetched again).
inline void mb_store(unsigned long long* p, unsigned long long v) { __atomic_store_n(p, v, __ATOMIC_RELAXED); }
inline unsigned long long mb_load(const unsigned long long* p) { return __atomic_load_n(p, __ATOMIC_RELAXED); }
inline unsigned long long mb_check(const unsigned long long* mb) {
    unsigned long long c = SR_SERVER_CHK;
    for (int i = 0; i < 7; ++i) c ^= mb_load(mb + i);
    return c;
}
// epoch word of the line for this launch with the command of the request that stands in the line (kept on the host: a
// launch that was called off has ~0 in this word); the check word follows it
inline void mb_set_epoch(sr_server& sv, unsigned long long epoch) {
    mb_store(sv.mb + 5, (epoch << 8) | (sv.cmd & 0xffull));
    mb_store(sv.mb + 7, mb_check(sv.mb));
    std::atomic_thread_fence(std::memory_order_seq_cst);
}
// call the launch off: every workgroup leaves at its next look, whatever request it is waiting for
inline void mb_call_off(sr_server& sv) {
    mb_store(sv.mb + 5, ~0ull);
    std::atomic_thread_fence(std::memory_order_seq_cst);
}

void registry_add(sr_gp* h) {
    std::lock_guard<std::mutex> lk(g_srv_mutex);
    if (std::find(g_srv_running.begin(), g_srv_running.end(), h) == g_srv_running.end()) g_srv_running.push_back(h);
}
void registry_remove(sr_gp* h) {
    std::lock_guard<std::mutex> lk(g_srv_mutex);
    g_srv_running.erase(std::remove(g_srv_running.begin(), g_srv_running.end(), h), g_srv_running.end());
}

// launch the kernel for the requests from first_seq on (the previous launch, if any, has left the device)
int server_launch(sr_gp* h, unsigned long long first_seq) {
    sr_server& sv = h->srv;
    for (int d = 0; d < slots_of(h); ++d) sv.reply[SR_SERVER_ALIVE + d] = 1ull;
    ++sv.epoch;
    mb_set_epoch(sv, sv.epoch);
    sr_kstar_args ka{};
    ka.Z = h->Z; ka.alpha = h->alpha; ka.ls = h->ls; ka.sf2 = h->sf2;
    ka.kp = h->general ? h->kp : nullptr;
    ka.xa = nullptr; ka.lda = h->D; ka.na = h->D; ka.xb = nullptr; ka.ldb = 0; ka.nb = 0;
    ka.N = h->N; ka.Np = h->Np; ka.D = h->D; ka.n_out = h->n_out; ka.nsplit = 1; ka.T = 1; ka.Tp = 1;
    sr_server_args sa{};
    sa.mb = sv.mb_dev; sa.out = sv.out_dev; sa.reply = sv.reply_dev;
    sa.first_seq = first_seq; sa.idle_ticks = sv.idle_ticks; sa.epoch = sv.epoch;
    SR_TRY(sr_launch_gp_server(ka, h->Wt, sa, sv.stream));
    sv.running = 1;
    ++sv.launches;
    registry_add(h);
    return SR_OK;
}

bool any_left(const sr_gp* h) {
    const volatile unsigned long long* alive = h->srv.reply + SR_SERVER_ALIVE;
    for (int d = 0; d < slots_of(h); ++d)
        if (alive[d] == 0ull) return true;
    return false;
}

}  // namespace

namespace {
int quiesce_locked(sr_gp* h) {
    sr_server& sv = h->srv;
    if (!sv.running) return SR_OK;
    mb_call_off(sv);
    sr_dev_guard guard(h->device);
    const hipError_t e = hipStreamSynchronize(sv.stream);
    sv.running = 0;
    sv.stale = 0;
    registry_remove(h);
    SR_HIP(e);
    return SR_OK;
}
}  // namespace

// Take the server of this handle off the device (it stays armed: the next sr_gp_server_call launches it again).
int srh::server_quiesce(sr_gp* h) {
    std::lock_guard<std::mutex> lk(h->srv.mu);
    return quiesce_locked(h);
}

// Before a device-wide wait (hipDeviceSynchronize inside the library): every resident server of this device leaves now
// instead of after its idle time-out.
void srh::servers_quiesce_device(int device) {
    std::vector<sr_gp*> hs;
    {
        std::lock_guard<std::mutex> lk(g_srv_mutex);
        for (sr_gp* h : g_srv_running)
            if (h->device == device) hs.push_back(h);
    }
    for (sr_gp* h : hs) (void)server_quiesce(h);
}

void srh::server_release(sr_gp* h) {
    (void)server_quiesce(h);
    sr_server& sv = h->srv;
    sv.armed = 0;
    if (sv.stream) { (void)hipStreamDestroy(sv.stream); sv.stream = nullptr; }
    if (sv.pinned) { (void)hipHostFree(sv.pinned); sv.pinned = nullptr; }
    sv.mb = sv.reply = nullptr; sv.out = nullptr;
}

extern "C" int sr_gp_server_start(sr_gp_t h, double idle_timeout_s) {
    SR_CHECK(h != nullptr, SR_EINVAL, "sr_gp_server_start: NULL handle");
    SR_CHECK(h->factorized, SR_ESTATE, "sr_gp_server_start: model not factorized");
    SR_CHECK(idle_timeout_s > 0.0 && idle_timeout_s <= 10.0, SR_EINVAL, "sr_gp_server_start: idle time-out %g s outside (0, 10]",
             idle_timeout_s);
    if (!servable(h)) {
        sr_set_error("sr_gp_server_start: no resident server for this model (Np <= %d, D <= 5; Np=%d D=%d)", SR_FUSED_NP,
                     h->Np, h->D);
        return SR_EUNSUPPORTED;
    }
    SR_DEVICE(h->device);
    sr_server& sv = h->srv;
    std::lock_guard<std::mutex> lk(sv.mu);
    SR_TRY(quiesce_locked(h));
    const size_t bytes = (MB_WORDS + REPLY_WORDS) * sizeof(unsigned long long) + rec_doubles(h) * sizeof(double);
    if (sv.pinned && sv.pinned_bytes < bytes) { (void)hipHostFree(sv.pinned); sv.pinned = nullptr; }
    if (!sv.pinned) {
        void* p = nullptr;
        // COHERENT (fine-grained): the device must read the mailbox from host memory every time it looks -- through a
        // cacheable mapping a request became visible to the polling workgroups only after ~180 us
        SR_HIP(hipHostMalloc(&p, bytes, hipHostMallocMapped | hipHostMallocCoherent));
        void* pd = nullptr;
        const hipError_t e = hipHostGetDevicePointer(&pd, p, 0);
        if (e != hipSuccess) {
            (void)hipHostFree(p);
            (void)hipGetLastError();
            sr_set_error("sr_gp_server_start: pinned host memory is not device-visible here (%s)", hipGetErrorString(e));
            return SR_EUNSUPPORTED;
        }
        memset(p, 0, bytes);
        sv.pinned = p; sv.pinned_bytes = bytes;
        sv.mb = (unsigned long long*)p; sv.reply = sv.mb + MB_WORDS; sv.out = (double*)(sv.reply + REPLY_WORDS);
        sv.mb_dev = (unsigned long long*)pd; sv.reply_dev = sv.mb_dev + MB_WORDS; sv.out_dev = (double*)(sv.reply_dev + REPLY_WORDS);
        sv.next_seq = 1;
    }
    if (!sv.stream) SR_HIP(hipStreamCreateWithFlags(&sv.stream, hipStreamNonBlocking));
    sv.idle_ticks = (unsigned long long)(idle_timeout_s * 1e8);           // 100 MHz wall clock
    sv.armed = 1;
    return server_launch(h, sv.next_seq);
}

extern "C" int sr_gp_server_stop(sr_gp_t h) {
    SR_CHECK(h != nullptr, SR_EINVAL, "sr_gp_server_stop: NULL handle");
    const int rc = server_quiesce(h);
    h->srv.armed = 0;
    return rc;
}

extern "C" int sr_gp_server_state(sr_gp_t h, int* armed, int* resident, long* launches, long* calls) {
    SR_CHECK(h != nullptr, SR_EINVAL, "sr_gp_server_state: NULL handle");
    std::lock_guard<std::mutex> lk(h->srv.mu);            // (a call or a quiesce on another thread writes these)
    if (armed) *armed = h->srv.armed;
    if (resident) *resident = (h->srv.running && !any_left(h)) ? 1 : 0;
    if (launches) *launches = h->srv.launches;
    if (calls) *calls = h->srv.calls;
    return SR_OK;
}

extern "C" int sr_gp_server_call(sr_gp_t h, const double* x_host, int second_order, double* out_host, double timeout_s) {
    SR_CHECK(h != nullptr && x_host && out_host, SR_EINVAL, "sr_gp_server_call: NULL argument");
    SR_CHECK(timeout_s == timeout_s, SR_EINVAL, "sr_gp_server_call: time-out is NaN");
    sr_server& sv = h->srv;
    std::lock_guard<std::mutex> lk(sv.mu);
    if (!sv.armed) { sr_set_error("sr_gp_server_call: no server armed (sr_gp_server_start)"); return SR_EUNSUPPORTED; }
    if (!servable(h)) {                                   // the model has changed under the armed server
        (void)quiesce_locked(h);
        sv.armed = 0;
        sr_set_error("sr_gp_server_call: the model no longer has a resident server (Np=%d)", h->Np);
        return SR_EUNSUPPORTED;
    }
    const unsigned long long seq = sv.next_seq;
    if (!sv.running || sv.stale || any_left(h)) {
        // never launched for this model state, (partly) gone on its idle time-out, or called off after a request that was
        // given up: wait for the rest to leave, launch anew
        SR_DEVICE(h->device);
        if (sv.running) {
            mb_call_off(sv);                                  // workgroups still polling leave at once
            SR_HIP(hipStreamSynchronize(sv.stream));
        }
        SR_TRY(server_launch(h, seq));
        sv.stale = 0;
    }
    const int D = h->D, n = h->n_out, parts = parts_of(h), nslot = n * parts;
    {
        // the request: query and command, the check word over the line as it will stand, the sequence number last
        unsigned long long w[8];
        for (int j = 0; j < 5; ++j) {
            const double xv = j < D ? x_host[j] : 0.0;
            memcpy(&w[j], &xv, sizeof(double));
        }
        sv.cmd = second_order == 2 ? SR_SERVER_CMD_PING : (second_order ? SR_SERVER_CMD_SECOND : SR_SERVER_CMD_FIRST);   // (2: diagnostics)
        w[5] = (sv.epoch << 8) | sv.cmd;
        w[6] = seq;
        w[7] = SR_SERVER_CHK;
        for (int i = 0; i < 7; ++i) w[7] ^= w[i];
        for (int i = 0; i < 6; ++i) mb_store(sv.mb + i, w[i]);
        mb_store(sv.mb + 7, w[7]);
        std::atomic_thread_fence(std::memory_order_release);
        mb_store(sv.mb + 6, w[6]);
    }
    const volatile unsigned long long* reply = sv.reply;
    const auto t0 = std::chrono::steady_clock::now();
    long spins = 0;
    for (;;) {
        bool all = true;
        for (int d = 0; d < nslot; ++d) all = all && (reply[d] == seq);
        if (all) break;
        srh::cpu_relax();
        if ((++spins & 1023) == 0) {
            if (any_left(h)) {
                // a workgroup left on its idle time-out between our look at `alive` and the request: relaunch for this
                // sequence number (the request is still in the mailbox; answers are idempotent)
                bool done = true;
                for (int d = 0; d < nslot; ++d) done = done && (reply[d] == seq);
                if (done) break;
                SR_DEVICE(h->device);
                mb_call_off(sv);
                SR_HIP(hipStreamSynchronize(sv.stream));
                SR_TRY(server_launch(h, seq));
            }
            const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (waited > timeout_s) {
                // give the request up: its sequence number is never used again (workgroups that did answer it hold answers to
                // THIS query in their reply words), and the launch is called off so that the next call starts a fresh one
                mb_call_off(sv);
                sv.stale = 1;
                ++sv.next_seq;
                sr_set_error("sr_gp_server_call: no answer to request %llu within %.3f s", seq, timeout_s);
                return SR_ESTATE;
            }
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    if (second_order == 2) {
        out_host[0] = (double)sv.reply[2 * SR_SERVER_ALIVE] * 1e-2;      // us of the last evaluation on the device
    } else {
        // per-(output, part) records -> the API layout [mu n | var n | jac_mu n x D | jac_var n x D | hess n x D x D].  A model
        // served in parts: part 0 carries mu and the mean's derivatives, every part its strips' share of |U^-T k*|^2 and of
        // the dot products behind d var/dx -- added here in ascending part order
        const int DD = D * D;
        for (int d = 0; d < n; ++d) {
            const double* r = sv.out + (size_t)d * parts * SR_SERVER_REC;
            out_host[d] = r[0];
            for (int j = 0; j < D; ++j) out_host[2 * n + d * D + j] = r[2 + j];
            if (parts == 1) {
                out_host[n + d] = r[1];
                if (second_order) for (int j = 0; j < D; ++j) out_host[2 * n + n * D + d * D + j] = r[2 + D + j];
            } else {
                double q0 = 0.0;
                for (int pp = 0; pp < parts; ++pp) q0 += r[(size_t)pp * SR_SERVER_REC + 1];
                double v = r[SR_SERVER_REC - 1] - q0;                   // sf2 - |U^-T k*|^2
                if (!(v > SR_VAR_CLIP)) v = SR_VAR_CLIP;
                out_host[n + d] = v;
                if (second_order)
                    for (int j = 0; j < D; ++j) {
                        double qj = 0.0;
                        for (int pp = 0; pp < parts; ++pp) qj += r[(size_t)pp * SR_SERVER_REC + 2 + D + j];
                        out_host[2 * n + n * D + d * D + j] = -2.0 * qj;
                    }
            }
            if (second_order) for (int q = 0; q < DD; ++q) out_host[2 * n + 2 * n * D + d * DD + q] = r[2 + 2 * D + q];
        }
    }
    ++sv.next_seq;
    ++sv.calls;
    return SR_OK;
}
