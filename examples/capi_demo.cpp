// capi_demo.cpp -- the C-ABI of include/safereach.h used from a plain HIP host program: no Python, no PyTorch.
//
//   hipcc --offload-arch=gfx950 -O2 examples/capi_demo.cpp -Iinclude -Lsafe_exploration_amd -lsafereach \
//         -Wl,-rpath,$PWD/safe_exploration_amd -o /tmp/capi_demo && /tmp/capi_demo
//
// 1. sr_ellipsoid_step on the worked anchor of SURVEY.md 8(c) (GP outputs given as constants; the expected
//    numbers were produced by the reference's own onestep_reachability).
// 2. a 2-output GP on N = 300 synthetic points: factorise, predict at the training inputs and verify the exact
//    posterior identity  mu(z_i) + s2n * alpha_i == y_i ; then one fused one-step reachability batch.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "safereach.h"

#define CK(call) do { int rc_ = (call); if (rc_ != SR_OK) { std::fprintf(stderr, "%s -> %d: %s\n", #call, rc_, sr_last_error()); return 1; } } while (0)
#define HK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { std::fprintf(stderr, "%s -> %s\n", #call, hipGetErrorString(e_)); return 1; } } while (0)

static double* to_dev(const std::vector<double>& v) {
    double* d = nullptr;
    if (hipMalloc((void**)&d, v.size() * sizeof(double)) != hipSuccess) std::abort();
    if (hipMemcpy(d, v.data(), v.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) std::abort();
    return d;
}
static std::vector<double> to_host(const double* d, size_t n) {
    std::vector<double> v(n);
    if (hipMemcpy(v.data(), d, n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) std::abort();
    return v;
}

int main() {
    int ndev = 0;
    CK(sr_device_count(&ndev));
    std::printf("libsafereach version %d, %d device(s)\n", sr_version(), ndev);

    // ---- 1. anchor: n_s = 2, n_u = 1
    double *p = to_dev({0.1, -0.2}), *q = to_dev({0.1, 0.04, 0.04, 0.13}), *kff = to_dev({0.3});
    double *kfb = to_dev({0.4, -0.1}), *mu = to_dev({0.05, -0.02}), *var = to_dev({0.01, 0.04});
    double *jac = to_dev({0.1, 0.2, 0.3, -0.1, 0.05, 0.2}), *a = to_dev({1, 0, 0, 1}), *b = to_dev({0, 0});
    double *l = to_dev({0.05, 0.02}), *p1 = to_dev({0, 0}), *q1 = to_dev({0, 0, 0, 0});
    CK(sr_ellipsoid_step(0, 1, 2, 1, p, q, kff, kfb, mu, var, jac, a, b, l, l, 2.0, p1, q1, nullptr, nullptr));
    HK(hipDeviceSynchronize());
    std::vector<double> hq = to_host(q1, 4), hp = to_host(p1, 2);
    const double ref[4] = {0.605462201964151, 0.158620529130949, 0.158620529130949, 0.943186225924461};
    double err = 0;
    for (int i = 0; i < 4; ++i) err = std::fmax(err, std::fabs(hq[i] - ref[i]) / ref[i]);
    std::printf("anchor: p1 = [%.3f %.3f]  Q1 = [[%.12f %.12f] [%.12f %.12f]]  max rel err %.2e\n", hp[0], hp[1],
                hq[0], hq[1], hq[2], hq[3], err);
    if (err > 1e-12 || std::fabs(hp[0] - 0.15) > 1e-14 || std::fabs(hp[1] + 0.22) > 1e-14) return 2;

    // ---- 2. GP model through the handle API
    const int N = 300, D = 3, n_out = 2, T = 300;
    std::vector<double> Z(N * D), Y(N * n_out);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (s >> 8) * (1.0 / 16777216.0) * 2.0 - 1.0; };
    for (auto& z : Z) z = rnd();
    for (int i = 0; i < N; ++i)
        for (int d = 0; d < n_out; ++d) Y[i * n_out + d] = 0.1 * std::sin(2.0 * (Z[i * D] + (d + 1) * Z[i * D + 1] - Z[i * D + 2]));
    const double s2n = 1e-4 + 1e-5 + 1e-8;
    double *dZ = to_dev(Z), *dY = to_dev(Y), *dls = to_dev({0.8, 1.0, 1.2, 1.1, 0.7, 0.9});
    double *dsf = to_dev({0.01, 0.01}), *dsn = to_dev({s2n, s2n});
    sr_gp_t h = nullptr;
    CK(sr_gp_create(&h, 0, N, D, n_out));
    CK(sr_gp_set_data(h, dZ, dY, dls, dsf, dsn, nullptr));
    int info[2] = {0, 0};
    CK(sr_gp_factorize(h, nullptr, info));
    double *dmu = to_dev(std::vector<double>(T * n_out)), *dvar = to_dev(std::vector<double>(T * n_out));
    double* dalpha = to_dev(std::vector<double>(n_out * N));
    CK(sr_gp_predict(h, dZ, T, dmu, dvar, nullptr, nullptr));
    CK(sr_gp_export(h, dalpha, nullptr, nullptr));
    HK(hipDeviceSynchronize());
    std::vector<double> hmu = to_host(dmu, T * n_out), hal = to_host(dalpha, n_out * N), hvar = to_host(dvar, T * n_out);
    double res = 0, vmin = 1e300;
    for (int i = 0; i < N; ++i)
        for (int d = 0; d < n_out; ++d) {
            res = std::fmax(res, std::fabs(hmu[i * n_out + d] + s2n * hal[d * N + i] - Y[i * n_out + d]));
            vmin = std::fmin(vmin, hvar[i * n_out + d]);
        }
    std::printf("GP N=%d: max |mu(z_i) + s2n alpha_i - y_i| = %.2e, min var = %.2e\n", N, res, vmin);
    if (res > 1e-9 || !(vmin > 0)) return 3;

    // fused one-step reachability for the T training inputs as query states (k_ff = z[:, 2], Q = 0.01 I)
    std::vector<double> P(T * 2), KFF(T), Q(T * 4, 0.0), KFB(T * 2, 0.05);
    for (int t = 0; t < T; ++t) { P[2 * t] = Z[t * D]; P[2 * t + 1] = Z[t * D + 1]; KFF[t] = Z[t * D + 2]; Q[4 * t] = Q[4 * t + 3] = 0.01; }
    double *dP = to_dev(P), *dKFF = to_dev(KFF), *dQ = to_dev(Q), *dKFB = to_dev(KFB);
    double *dP1 = to_dev(std::vector<double>(T * 2)), *dQ1 = to_dev(std::vector<double>(T * 4));
    int* dbad = nullptr;
    HK(hipMalloc((void**)&dbad, sizeof(int)));
    HK(hipMemset(dbad, 0, sizeof(int)));
    CK(sr_onestep_reach(h, T, dP, dQ, dKFF, dKFB, a, b, l, l, 2.0, dP1, dQ1, nullptr, dbad, nullptr));
    HK(hipDeviceSynchronize());
    int bad = -1;
    HK(hipMemcpy(&bad, dbad, sizeof(int), hipMemcpyDeviceToHost));
    std::vector<double> hQ1 = to_host(dQ1, T * 4);
    double asym = 0, dmin = 1e300;
    for (int t = 0; t < T; ++t) {
        asym = std::fmax(asym, std::fabs(hQ1[4 * t + 1] - hQ1[4 * t + 2]));
        dmin = std::fmin(dmin, hQ1[4 * t] * hQ1[4 * t + 3] - hQ1[4 * t + 1] * hQ1[4 * t + 2]);
    }
    std::printf("one-step reachability T=%d: n_bad = %d, max |Q-Q^T| = %.1e, min det Q1 = %.3e\n", T, bad, asym, dmin);
    if (bad != 0 || asym > 1e-15 || !(dmin > 0)) return 4;

    // ---- 3. blocking single queries through the resident server (what an MPC's NLP solver does per iteration): no
    // launch per query; plain host buffers in and out
    if (sr_gp_server_start(h, 0.005) == 0) {
        std::vector<double> out(2 * n_out + 2 * n_out * D + n_out * D * D);
        double worst = 0;
        for (int t = 0; t < 50; ++t) {
            CK(sr_gp_server_call(h, &Z[t * D], 0, out.data(), 1.0));
            for (int d = 0; d < n_out; ++d) {
                worst = std::fmax(worst, std::fabs(out[d] - hmu[t * n_out + d]));
                worst = std::fmax(worst, std::fabs(out[n_out + d] - hvar[t * n_out + d]));
            }
        }
        const auto t0 = std::chrono::steady_clock::now();
        for (int t = 0; t < 2000; ++t) CK(sr_gp_server_call(h, &Z[(t % N) * D], 1, out.data(), 1.0));
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 2000;
        std::printf("resident server: 50 queries against sr_gp_predict max |diff| = %.1e; second-order call %.1f us\n", worst, us);
        CK(sr_gp_server_stop(h));
        if (worst > 1e-12) return 5;
    } else {
        std::printf("resident server: not available here (%s)\n", sr_last_error());
    }
    CK(sr_gp_destroy(h));
    std::printf("capi_demo OK\n");
    return 0;
}
