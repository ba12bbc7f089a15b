// sr_comm.hip -- libsafereach_comm.so: model replication over RCCL for hosts WITHOUT PyTorch.
//
// The path shards over queries and needs exactly one exchange: after a model update the training set's posterior
// state (alpha, U^-1) travels ONCE from the GPU that factorised to the others (one ncclBroadcast over
// xGMI).  The Python mirror does that with torch.distributed (backend "nccl" = RCCL); this library offers the same
// step to a plain C / C++ host driving several GPUs from one process (ncclCommInitAll: one communicator per device).
// It is a SEPARATE shared object on purpose: PyTorch bundles its own RCCL, and libsafereach.so must stay loadable
// next to it without a second copy of the collective runtime in the process.
//
// Built only from the public C-ABI of include/safereach.h (sr_gp_dims / sr_gp_export / sr_gp_import).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <cstdarg>
#include <cstdio>
#include <vector>
#include "../../include/safereach_comm.h"

static thread_local char g_cerr[512] = "";
static std::vector<ncclComm_t> g_comms;
static std::vector<int> g_devs;
static std::vector<hipStream_t> g_streams;

static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_cerr, sizeof(g_cerr), fmt, ap);
    va_end(ap);
    return code;
}
#define CH(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(SR_EHIP, "%s -> %s", #call, hipGetErrorString(e_)); } while (0)
#define CN(call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) return fail(SR_EHIP, "%s -> %s", #call, ncclGetErrorString(r_)); } while (0)

extern "C" const char* sr_comm_last_error(void) { return g_cerr; }

extern "C" int sr_comm_destroy(void) {
    for (size_t i = 0; i < g_comms.size(); ++i) {
        (void)hipSetDevice(g_devs[i]);
        if (g_streams[i]) (void)hipStreamDestroy(g_streams[i]);
        (void)ncclCommDestroy(g_comms[i]);
    }
    g_comms.clear(); g_devs.clear(); g_streams.clear();
    return SR_OK;
}

extern "C" int sr_comm_init_all(int ndev, const int* devices) {
    if (ndev < 1 || devices == nullptr) return fail(SR_EINVAL, "sr_comm_init_all: ndev=%d", ndev);
    sr_comm_destroy();
    int prev = 0;
    (void)hipGetDevice(&prev);
    g_comms.assign(ndev, nullptr);
    g_devs.assign(devices, devices + ndev);
    g_streams.assign(ndev, nullptr);
    CN(ncclCommInitAll(g_comms.data(), ndev, devices));
    for (int i = 0; i < ndev; ++i) {
        CH(hipSetDevice(devices[i]));
        CH(hipStreamCreateWithFlags(&g_streams[i], hipStreamNonBlocking));
    }
    (void)hipSetDevice(prev);
    return SR_OK;
}

// handles[i] lives on the i-th device of sr_comm_init_all; handles[root] is factorised, the others have their data
// set (sr_gp_set_data[_general]: Z, targets, hyper-parameters -- a few KB the host has anyway).  On return every handle
// holds the root's posterior (alpha, U^-1) and is ready to predict.  What travels: alpha and the PACKED upper triangle
// of U^-1 -- n_out * (N + N (N + 1) / 2) doubles, half of the dense buffer -- in pieces of <= 64 MB through ONE staging
// buffer per device (pack on the root, broadcast, unpack on the receivers, all in order on the device's stream): no
// copy of the factor's size anywhere (the first version staged the dense n_out * Np^2 buffer on every device).
static const long SR_COMM_PIECE = 8L << 20;       // doubles per staging piece

extern "C" int sr_comm_bcast(sr_gp_t* handles, int ndev, int root) {
    if ((int)g_comms.size() != ndev || ndev < 1) return fail(SR_ESTATE, "sr_comm_bcast: call sr_comm_init_all(%d, ...) first", ndev);
    if (handles == nullptr || root < 0 || root >= ndev) return fail(SR_EINVAL, "sr_comm_bcast: bad argument");
    int N = 0, D = 0, n_out = 0;
    long Np = 0;
    if (sr_gp_dims(handles[root], &N, &D, &n_out, &Np) != SR_OK) return fail(SR_EINVAL, "sr_comm_bcast: %s", sr_last_error());
    for (int i = 0; i < ndev; ++i) {
        int n = 0, d = 0, o = 0;
        long np_ = 0;
        if (sr_gp_dims(handles[i], &n, &d, &o, &np_) != SR_OK || n != N || d != D || o != n_out || np_ != Np)
            return fail(SR_EINVAL, "sr_comm_bcast: handle %d does not match the root's shape", i);
    }
    int prev = 0;
    (void)hipGetDevice(&prev);
    // row ranges of the packed triangle, each at most SR_COMM_PIECE doubles (a single row may exceed it)
    struct piece { long r0, r1, cnt; };
    std::vector<piece> pieces;
    long cap = 0;
    for (long r0 = 0; r0 < N;) {
        long r1 = r0 + 1, cnt = N - r0;
        while (r1 < N && cnt + (N - r1) <= SR_COMM_PIECE) { cnt += N - r1; ++r1; }
        pieces.push_back({r0, r1, cnt});
        cap = cnt > cap ? cnt : cap;
        r0 = r1;
    }
    const size_t na = (size_t)n_out * N;
    std::vector<double*> buf(ndev, nullptr);                  // [alpha | one staging piece]
    int rc = SR_OK;
    for (int i = 0; i < ndev && rc == SR_OK; ++i) {
        if (hipSetDevice(g_devs[i]) != hipSuccess || hipMalloc((void**)&buf[i], (na + (size_t)cap) * sizeof(double)) != hipSuccess)
            rc = fail(SR_EHIP, "sr_comm_bcast: allocation of %zu bytes on device %d failed", (na + (size_t)cap) * sizeof(double), g_devs[i]);
    }
    auto bcast = [&](size_t off, size_t count) -> int {
        ncclResult_t r = ncclGroupStart();
        for (int i = 0; i < ndev && r == ncclSuccess; ++i)
            r = ncclBroadcast(buf[i] + off, buf[i] + off, count, ncclDouble, root, g_comms[i], g_streams[i]);
        const ncclResult_t r2 = ncclGroupEnd();
        if (r != ncclSuccess || r2 != ncclSuccess)
            return fail(SR_EHIP, "sr_comm_bcast: RCCL broadcast failed: %s", ncclGetErrorString(r != ncclSuccess ? r : r2));
        return SR_OK;
    };
    if (rc == SR_OK && sr_gp_export(handles[root], buf[root], nullptr, g_streams[root]) != SR_OK)
        rc = fail(SR_ESTATE, "sr_comm_bcast: %s", sr_last_error());
    if (rc == SR_OK) rc = bcast(0, na);
    for (int i = 0; i < ndev && rc == SR_OK; ++i)
        if (i != root && sr_gp_import_begin(handles[i], buf[i], g_streams[i]) != SR_OK)
            rc = fail(SR_ESTATE, "sr_comm_bcast: %s", sr_last_error());
    for (int d = 0; d < n_out && rc == SR_OK; ++d) {
        for (const piece& pc : pieces) {
            if (sr_gp_export_packed(handles[root], d, pc.r0, pc.r1, buf[root] + na, g_streams[root]) != SR_OK) {
                rc = fail(SR_ESTATE, "sr_comm_bcast: %s", sr_last_error());
                break;
            }
            if ((rc = bcast(na, (size_t)pc.cnt)) != SR_OK) break;
            for (int i = 0; i < ndev && rc == SR_OK; ++i)
                if (i != root && sr_gp_import_packed(handles[i], d, pc.r0, pc.r1, buf[i] + na, g_streams[i]) != SR_OK)
                    rc = fail(SR_ESTATE, "sr_comm_bcast: %s", sr_last_error());
            if (rc != SR_OK) break;
        }
    }
    for (int i = 0; i < ndev; ++i) {
        (void)hipSetDevice(g_devs[i]);
        if (g_streams[i]) (void)hipStreamSynchronize(g_streams[i]);
        if (rc == SR_OK && i != root && sr_gp_import_end(handles[i]) != SR_OK)
            rc = fail(SR_ESTATE, "sr_comm_bcast: %s", sr_last_error());
        if (buf[i]) (void)hipFree(buf[i]);
    }
    (void)hipSetDevice(prev);
    return rc;
}
