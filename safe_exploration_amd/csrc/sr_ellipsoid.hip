// sr_ellipsoid.hip -- per-query Jacobian-linearised ellipsoid propagate / sum.
//
// One thread per query; everything lives in registers (n_s <= 8, n_u <= 4 are template parameters
// so every small-matrix index is a compile-time constant).
//
// replaces the algebra of
//   /root/reference/safe_exploration/gp_reachability.py:65-88   (point branch)
//   /root/reference/safe_exploration/gp_reachability.py:89-156  (ellipsoid branch)
//   /root/reference/safe_exploration/utils.py:108-144           (compute_remainder_overapproximations)
//   /root/reference/safe_exploration/utils_ellipsoid.py:63-94, 197-233
//   /root/reference/safe_exploration/gp_reachability.py:215-250 (lin_ellipsoid_safety_distance)
#include "sr_common.h"
#include "sr_ellipsoid_dev.h"

template <int NS, int NU>
__global__ __launch_bounds__(256) void sr_ellipsoid_kernel(sr_ell_args a) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= a.T) return;
    sr_ellipsoid_one<NS, NU>(a, t);
}

template <int NS, int NU>
__global__ __launch_bounds__(256) void sr_remainder_kernel(long T, const double* __restrict__ qg,
                                                           const double* __restrict__ kg,
                                                           const double* __restrict__ l_mu,
                                                           const double* __restrict__ l_sigma,
                                                           double* __restrict__ u_mu,
                                                           double* __restrict__ u_sigma) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    double q[NS][NS], kfb[NU][NS];
#pragma unroll
    for (int i = 0; i < NS; ++i)
#pragma unroll
        for (int j = 0; j < NS; ++j) q[i][j] = qg[(t * NS + i) * NS + j];
#pragma unroll
    for (int k = 0; k < NU; ++k)
#pragma unroll
        for (int j = 0; j < NS; ++j) kfb[k][j] = kg[(t * NU + k) * NS + j];
    const double r2 = sr_lambda_max_qb<NS, NU>(q, kfb);
    const double r1 = sqrt(r2);
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        u_mu[t * NS + i] = l_mu[i] * r2;
        u_sigma[t * NS + i] = l_sigma[i] * r1;
    }
}

// d[t][m] = h_m . p + c sqrt(h_m Q h_m^T) - h_vec[m]           (gp_reachability.py:245-248)
__global__ __launch_bounds__(256) void sr_safety_kernel(long T, int n_s, int m,
                                                        const double* __restrict__ p,
                                                        const double* __restrict__ q,
                                                        const double* __restrict__ h_mat,
                                                        const double* __restrict__ h_vec, double c,
                                                        double* __restrict__ d) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    const double* pt = p + t * n_s;
    const double* qt = q + t * n_s * n_s;
    for (int r = 0; r < m; ++r) {
        const double* h = h_mat + r * n_s;
        double dc = 0.0, quad = 0.0;
        for (int i = 0; i < n_s; ++i) {
            dc = fma(h[i], pt[i], dc);
            double s = 0.0;
            for (int j = 0; j < n_s; ++j) s = fma(qt[i * n_s + j], h[j], s);
            quad = fma(s, h[i], quad);
        }
        d[t * m + r] = dc + c * sqrt(quad) - h_vec[r];
    }
}

// d[t][k] = (s_k - p_t)^T Q_t^-1 (s_k - p_t): T ellipsoids x K samples (samples shared, or one set per
// ellipsoid when per_t != 0).  Cholesky of Q_t per thread, then a triangular solve: |L^-1 (s-p)|^2.
// replaces utils_ellipsoid.distance_to_center / sample_inside_ellipsoid  utils_ellipsoid.py:16-60
template <int NS>
__global__ __launch_bounds__(256) void sr_distance_kernel(long T, int K, const double* __restrict__ samples,
                                                          int per_t, const double* __restrict__ p,
                                                          const double* __restrict__ q, double* __restrict__ d) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= T * K) return;
    const long t = idx / K;
    const int k = (int)(idx % K);
    double L[NS][NS], y[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        double sjj = q[(t * NS + j) * NS + j];
#pragma unroll
        for (int c = 0; c < j; ++c) sjj -= L[j][c] * L[j][c];
        const double ljj = sqrt(sjj);
        L[j][j] = ljj;
#pragma unroll
        for (int i = 0; i < NS; ++i)
            if (i > j) {
                double v = q[(t * NS + i) * NS + j];
#pragma unroll
                for (int c = 0; c < j; ++c) v -= L[i][c] * L[j][c];
                L[i][j] = v / ljj;
            }
    }
    const double* sm = samples + (per_t ? (t * K + k) : (long)k) * NS;
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        double v = sm[i] - p[t * NS + i];
#pragma unroll
        for (int c = 0; c < i; ++c) v -= L[i][c] * y[c];
        y[i] = v / L[i][i];
        acc = fma(y[i], y[i], acc);
    }
    d[idx] = acc;
}

int sr_launch_distance(long T, int K, int n_s, const double* samples, int per_t, const double* p,
                       const double* q, double* d, hipStream_t s) {
    if (T <= 0 || K <= 0) return SR_OK;
    dim3 grid((unsigned)((T * K + 255) / 256));
#define SR_DIST_CASE(NS) case NS: hipLaunchKernelGGL(sr_distance_kernel<NS>, grid, dim3(256), 0, s, T, K, samples, per_t, p, q, d); break
    switch (n_s) {
        SR_DIST_CASE(1); SR_DIST_CASE(2); SR_DIST_CASE(3); SR_DIST_CASE(4);
        SR_DIST_CASE(5); SR_DIST_CASE(6); SR_DIST_CASE(7); SR_DIST_CASE(8);
        default: sr_set_error("distance: n_s=%d outside 1..%d", n_s, SR_MAX_NS); return SR_EUNSUPPORTED;
    }
#undef SR_DIST_CASE
    SR_HIP(hipGetLastError());
    return SR_OK;
}

// ---------------------------------------------------------------------------------------------
// Monte-Carlo propagation step (sampling_models.py:66-80 + ssm_gpy/gaussian_process.py:598-619):
//   S[t][j][:]  = mu[t][:] + sqrt(var[t][:]) * eps[t][j][:]          (marginal posterior samples)
//   z[t][j][:]  = [ S[t][j][:],  k_fb S[t][j][:] + k_ff ]            (next GP inputs, optional)
// One thread per sample; pure streaming (eps in, S/z out).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sr_sample_kernel(long T, int size, int n_out, int n_u,
                                                        const double* __restrict__ mu,
                                                        const double* __restrict__ var,
                                                        const double* __restrict__ eps,
                                                        double* __restrict__ S,
                                                        const double* __restrict__ k_fb,
                                                        const double* __restrict__ k_ff,
                                                        double* __restrict__ z) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= T * size) return;
    const long t = idx / size;
    double sv[SR_MAX_NS];
    for (int d = 0; d < n_out; ++d) {
        const double v = mu[t * n_out + d] + sqrt(var[t * n_out + d]) * eps[idx * n_out + d];
        sv[d] = v;
        S[idx * n_out + d] = v;
    }
    if (z) {
        double* zr = z + idx * (n_out + n_u);
        for (int d = 0; d < n_out; ++d) zr[d] = sv[d];
        for (int u = 0; u < n_u; ++u) {
            double a = k_ff[u];
            for (int d = 0; d < n_out; ++d) a = fma(k_fb[u * n_out + d], sv[d], a);
            zr[n_out + u] = a;
        }
    }
}

int sr_launch_sample(long T, int size, int n_out, int n_u, const double* mu, const double* var,
                     const double* eps, double* S, const double* k_fb, const double* k_ff, double* z,
                     hipStream_t s) {
    if (T <= 0 || size <= 0) return SR_OK;
    dim3 grid((unsigned)((T * size + 255) / 256));
    hipLaunchKernelGGL(sr_sample_kernel, grid, dim3(256), 0, s, T, size, n_out, n_u, mu, var, eps, S, k_fb,
                       k_ff, z);
    SR_HIP(hipGetLastError());
    return SR_OK;
}

// ---------------------------------------------------------------------------------------------
// dispatch on (n_s, n_u)
// ---------------------------------------------------------------------------------------------
template <int NS>
static int launch_ell_ns(const sr_ell_args& a, hipStream_t s) {
    dim3 grid((unsigned)((a.T + 255) / 256));
    switch (a.n_u) {
        case 1: hipLaunchKernelGGL((sr_ellipsoid_kernel<NS, 1>), grid, dim3(256), 0, s, a); break;
        case 2: hipLaunchKernelGGL((sr_ellipsoid_kernel<NS, 2>), grid, dim3(256), 0, s, a); break;
        case 3: hipLaunchKernelGGL((sr_ellipsoid_kernel<NS, 3>), grid, dim3(256), 0, s, a); break;
        case 4: hipLaunchKernelGGL((sr_ellipsoid_kernel<NS, 4>), grid, dim3(256), 0, s, a); break;
        default: sr_set_error("ellipsoid: n_u=%d outside 1..%d", a.n_u, SR_MAX_NU); return SR_EUNSUPPORTED;
    }
    SR_HIP(hipGetLastError());
    return SR_OK;
}

int sr_launch_ellipsoid(const sr_ell_args& a, hipStream_t s) {
    if (a.T <= 0) return SR_OK;
    switch (a.n_s) {
        case 1: return launch_ell_ns<1>(a, s);
        case 2: return launch_ell_ns<2>(a, s);
        case 3: return launch_ell_ns<3>(a, s);
        case 4: return launch_ell_ns<4>(a, s);
        case 5: return launch_ell_ns<5>(a, s);
        case 6: return launch_ell_ns<6>(a, s);
        case 7: return launch_ell_ns<7>(a, s);
        case 8: return launch_ell_ns<8>(a, s);
        default: sr_set_error("ellipsoid: n_s=%d outside 1..%d", a.n_s, SR_MAX_NS); return SR_EUNSUPPORTED;
    }
}

template <int NS>
static int launch_rem_ns(long T, int n_u, const double* q, const double* k, const double* lm,
                         const double* lsg, double* um, double* us, hipStream_t s) {
    dim3 grid((unsigned)((T + 255) / 256));
    switch (n_u) {
        case 1: hipLaunchKernelGGL((sr_remainder_kernel<NS, 1>), grid, dim3(256), 0, s, T, q, k, lm, lsg, um, us); break;
        case 2: hipLaunchKernelGGL((sr_remainder_kernel<NS, 2>), grid, dim3(256), 0, s, T, q, k, lm, lsg, um, us); break;
        case 3: hipLaunchKernelGGL((sr_remainder_kernel<NS, 3>), grid, dim3(256), 0, s, T, q, k, lm, lsg, um, us); break;
        case 4: hipLaunchKernelGGL((sr_remainder_kernel<NS, 4>), grid, dim3(256), 0, s, T, q, k, lm, lsg, um, us); break;
        default: sr_set_error("remainder: n_u=%d outside 1..%d", n_u, SR_MAX_NU); return SR_EUNSUPPORTED;
    }
    SR_HIP(hipGetLastError());
    return SR_OK;
}

int sr_launch_remainder(long T, int n_s, int n_u, const double* q, const double* k_fb,
                        const double* l_mu, const double* l_sigma, double* u_mu, double* u_sigma,
                        hipStream_t s) {
    if (T <= 0) return SR_OK;
    switch (n_s) {
        case 1: return launch_rem_ns<1>(T, n_u, q, k_fb, l_mu, l_sigma, u_mu, u_sigma, s);
        case 2: return launch_rem_ns<2>(T, n_u, q, k_fb, l_mu, l_sigma, u_mu, u_sigma, s);
        case 3: return launch_rem_ns<3>(T, n_u, q, k_fb, l_mu, l_sigma, u_mu, u_sigma, s);
        case 4: return launch_rem_ns<4>(T, n_u, q, k_fb, l_mu, l_sigma, u_mu, u_sigma, s);
        case 5: return launch_rem_ns<5>(T, n_u, q, k_fb, l_mu, l_sigma, u_mu, u_sigma, s);
        case 6: return launch_rem_ns<6>(T, n_u, q, k_fb, l_mu, l_sigma, u_mu, u_sigma, s);
        case 7: return launch_rem_ns<7>(T, n_u, q, k_fb, l_mu, l_sigma, u_mu, u_sigma, s);
        case 8: return launch_rem_ns<8>(T, n_u, q, k_fb, l_mu, l_sigma, u_mu, u_sigma, s);
        default: sr_set_error("remainder: n_s=%d outside 1..%d", n_s, SR_MAX_NS); return SR_EUNSUPPORTED;
    }
}

int sr_launch_safety(long T, int n_s, int m, const double* p, const double* q, const double* h_mat,
                     const double* h_vec, double c, double* d, hipStream_t s) {
    if (T <= 0) return SR_OK;
    hipLaunchKernelGGL(sr_safety_kernel, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, s, T, n_s,
                       m, p, q, h_mat, h_vec, c, d);
    SR_HIP(hipGetLastError());
    return SR_OK;
}
