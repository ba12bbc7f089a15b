// comm_demo.cpp -- replicate a factorised model over RCCL from a plain HIP host (no Python, no PyTorch) and shard a
// query batch over the GPUs of one process:
//
//   hipcc --offload-arch=gfx950 -O2 examples/comm_demo.cpp -Iinclude -Lsafe_exploration_amd -lsafereach -lsafereach_comm \
//         -Wl,-rpath,$PWD/safe_exploration_amd -o /tmp/comm_demo && /tmp/comm_demo
//
// Device 0 factorises; sr_comm_bcast sends alpha and U^-1 once to every other device; each device predicts its
// contiguous slice of the queries; the gathered result must equal device 0's prediction of the whole batch exactly.
// With a single GPU the program still goes through init / broadcast / destroy (a one-rank communicator).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "safereach_comm.h"

#define CK(call) do { int rc_ = (call); if (rc_ != SR_OK) { std::fprintf(stderr, "%s -> %d: %s | %s\n", #call, rc_, sr_last_error(), sr_comm_last_error()); return 1; } } while (0)
#define HK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { std::fprintf(stderr, "%s -> %s\n", #call, hipGetErrorString(e_)); return 1; } } while (0)

static double* to_dev(const std::vector<double>& v) {
    double* d = nullptr;
    if (hipMalloc((void**)&d, v.size() * sizeof(double)) != hipSuccess) std::abort();
    if (hipMemcpy(d, v.data(), v.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) std::abort();
    return d;
}

int main() {
    int ndev = 0;
    CK(sr_device_count(&ndev));
    if (ndev > 8) ndev = 8;
    const int N = 700, D = 3, n_out = 2, T = 1003;
    std::vector<double> Z(N * D), Y(N * n_out), X(T * D);
    unsigned s = 2024u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (s >> 8) * (1.0 / 16777216.0) * 2.0 - 1.0; };
    for (auto& z : Z) z = rnd();
    for (auto& x : X) x = 0.8 * rnd();
    for (int i = 0; i < N; ++i)
        for (int d = 0; d < n_out; ++d) Y[i * n_out + d] = std::sin(2.0 * (Z[i * D] + (d + 1) * Z[i * D + 1] - Z[i * D + 2]));
    const std::vector<double> ls = {0.7, 0.9, 1.1, 1.2, 0.8, 0.6}, sf2 = {1.0, 0.5}, noise = {1e-2 + 1e-5 + 1e-8, 2e-2 + 1e-5 + 1e-8};

    std::vector<int> devs(ndev);
    std::vector<sr_gp_t> h(ndev, nullptr);
    for (int i = 0; i < ndev; ++i) {
        devs[i] = i;
        HK(hipSetDevice(i));
        CK(sr_gp_create(&h[i], i, N, D, n_out));
        // every device gets the (tiny) data and hyper-parameters from the host; only device 0 factorises
        CK(sr_gp_set_data(h[i], to_dev(Z), to_dev(Y), to_dev(ls), to_dev(sf2), to_dev(noise), nullptr));
        HK(hipDeviceSynchronize());
    }
    HK(hipSetDevice(0));
    std::vector<int> info(n_out, 0);
    CK(sr_gp_factorize(h[0], nullptr, info.data()));
    CK(sr_comm_init_all(ndev, devs.data()));
    CK(sr_comm_bcast(h.data(), ndev, 0));

    // reference: the whole batch on device 0
    std::vector<double> mu0(T * n_out), var0(T * n_out), mu(T * n_out), var(T * n_out);
    {
        double *dx = to_dev(X), *dm = nullptr, *dv = nullptr;
        HK(hipMalloc((void**)&dm, mu0.size() * sizeof(double)));
        HK(hipMalloc((void**)&dv, var0.size() * sizeof(double)));
        CK(sr_gp_predict(h[0], dx, T, dm, dv, nullptr, nullptr));
        HK(hipDeviceSynchronize());
        HK(hipMemcpy(mu0.data(), dm, mu0.size() * sizeof(double), hipMemcpyDeviceToHost));
        HK(hipMemcpy(var0.data(), dv, var0.size() * sizeof(double), hipMemcpyDeviceToHost));
    }
    // shards: contiguous slices, sizes differ by at most one
    for (int i = 0; i < ndev; ++i) {
        const long base = T / ndev, rem = T % ndev;
        const long lo = i * base + (i < rem ? i : rem), cnt = base + (i < rem ? 1 : 0);
        HK(hipSetDevice(i));
        std::vector<double> xs(X.begin() + lo * D, X.begin() + (lo + cnt) * D);
        double *dx = to_dev(xs), *dm = nullptr, *dv = nullptr;
        HK(hipMalloc((void**)&dm, cnt * n_out * sizeof(double)));
        HK(hipMalloc((void**)&dv, cnt * n_out * sizeof(double)));
        CK(sr_gp_predict(h[i], dx, cnt, dm, dv, nullptr, nullptr));
        HK(hipDeviceSynchronize());
        HK(hipMemcpy(mu.data() + lo * n_out, dm, cnt * n_out * sizeof(double), hipMemcpyDeviceToHost));
        HK(hipMemcpy(var.data() + lo * n_out, dv, cnt * n_out * sizeof(double), hipMemcpyDeviceToHost));
    }
    double dmu = 0, dvar = 0;
    for (size_t k = 0; k < mu.size(); ++k) {
        dmu = std::fmax(dmu, std::fabs(mu[k] - mu0[k]));
        dvar = std::fmax(dvar, std::fabs(var[k] - var0[k]));
    }
    std::printf("%d device(s): max |mu - mu0| = %.3e  max |var - var0| = %.3e\n", ndev, dmu, dvar);
    CK(sr_comm_destroy());
    for (int i = 0; i < ndev; ++i) { HK(hipSetDevice(i)); CK(sr_gp_destroy(h[i])); }
    if (dmu > 1e-12 || dvar > 1e-12) return 2;       // (different batch sizes take different kernel paths: rounding only)
    std::printf("comm_demo OK\n");
    return 0;
}
