// sr_linearize.hip -- second-order outputs of the single-query CasADi boundary:
// d var/dx and the Hessian of mu, the extra outputs of linearize_predict(..., jacobians=True)
//   contract: /root/reference/safe_exploration/state_space_models.py:106-138, consumed at :402-415
//   reference implementation (fp32, autograd): ssm_pytorch/gaussian_process.py:333-385
// RBF closed forms (g = K_y^-1 k* = U^-1 (U^-T k*)):
//   d var/dx_j      = -2 sum_i g_i k*_i (z_ij - x_j) / l_j^2
//   d2 mu/dx_j dx_k = sum_i alpha_i k*_i [ (z_ij-x_j)(z_ik-x_k)/(l_j^2 l_k^2) - delta_jk / l_j^2 ]
// Latency path (T = 1): three small launches per output, HBM-bound on the 2 x N^2/2 factor reads.
// The other kernel identifiers (mat52, lin_rbf, lin_mat52) go through sr_linearize_general_kernel below.
#include "sr_common.h"
#include "sr_final_dev.h"

// y[i] = sum_{k <= i} M[k][i] * x[k * xs]   (M upper triangular, row-major): v = U^-T k*
__global__ __launch_bounds__(256) void sr_trmv_t_kernel(const double* __restrict__ M, long ld,
                                                        const double* __restrict__ x, long xs,
                                                        double* __restrict__ y, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double s0 = 0.0, s1 = 0.0;
    int k = 0;
    for (; k + 1 <= i; k += 2) {
        s0 = fma(M[(long)k * ld + i], x[(long)k * xs], s0);
        s1 = fma(M[(long)(k + 1) * ld + i], x[(long)(k + 1) * xs], s1);
    }
    if (k <= i) s0 = fma(M[(long)k * ld + i], x[(long)k * xs], s0);
    y[i] = s0 + s1;
}

int sr_launch_trmv_t(const double* M, long ld, const double* x, long xs, double* y, int n,
                     hipStream_t s) {
    hipLaunchKernelGGL(sr_trmv_t_kernel, dim3((n + 255) / 256), dim3(256), 0, s, M, ld, x, xs, y, n);
    SR_HIP(hipGetLastError());
    return SR_OK;
}

template <int DT>
__global__ __launch_bounds__(256) void sr_linearize_kernel(sr_lin_args a) {
    constexpr int NH = DT * (DT + 1) / 2;
    constexpr int NACC = DT + NH + 1;
    __shared__ double red[4][NACC];
    const int d = blockIdx.x;
    const int off = a.Np - a.N;
    const double* ks = a.Ks + (long)d * a.Np * a.Tp;       // column t = 0, stride Tp
    const double* g = a.g + (long)d * a.Np;
    const double* al = a.alpha + (long)d * a.Np;
    double il2[DT], x[DT], acc[NACC];
#pragma unroll
    for (int j = 0; j < DT; ++j) {
        const double l = (j < a.D) ? a.ls[d * a.D + j] : 1.0;
        il2[j] = (j < a.D) ? 1.0 / (l * l) : 0.0;
        x[j] = (j < a.D) ? a.x[j] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < NACC; ++q) acc[q] = 0.0;
    for (int i = threadIdx.x; i < a.N; i += 256) {
        const double k = ks[(long)(i + off) * a.Tp];
        const double gk = g[i + off] * k;
        const double w = al[i + off] * k;
        double df[DT];
#pragma unroll
        for (int j = 0; j < DT; ++j) df[j] = (j < a.D) ? (a.Z[(long)i * a.D + j] - x[j]) * il2[j] : 0.0;
        int q = DT;
#pragma unroll
        for (int j = 0; j < DT; ++j) {
            acc[j] = fma(gk, df[j], acc[j]);
#pragma unroll
            for (int c = 0; c < DT; ++c)
                if (c >= j) { acc[q] = fma(w * df[j], df[c], acc[q]); ++q; }
        }
        acc[NACC - 1] += w;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < NACC; ++q) {
        double v = acc[q];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) red[wave][q] = v;
    }
    __syncthreads();
    // one thread per accumulator (thread 0 holding all NACC totals cost the D <= 12 instantiation 980 B of scratch per lane)
    if (threadIdx.x < NACC - 1) {
        const int t = threadIdx.x;
        const double tot = red[0][t] + red[1][t] + red[2][t] + red[3][t];
        if (t < DT) {
            if (t < a.D) a.jac_var[d * a.D + t] = -2.0 * tot;
        } else {
            int e = t - DT, j = 0;
            while (e >= DT - j) { e -= DT - j; ++j; }          // entry (j, c = j + e) of the DT-wide upper triangle
            const int c = j + e;
            if (j < a.D && c < a.D) {
                double hv = tot;
                if (c == j) {
                    const double l = a.ls[d * a.D + j];
                    hv -= (red[0][NACC - 1] + red[1][NACC - 1] + red[2][NACC - 1] + red[3][NACC - 1]) * (1.0 / (l * l));
                }
                a.hess_mu[((long)d * a.D + j) * a.D + c] = hv;
                a.hess_mu[((long)d * a.D + c) * a.D + j] = hv;
            }
        }
    }
}

// General kernel family of sr_common.h (Matern-5/2, linear x stationary + linear; the kernels the reference's
// journal experiments use, ssm_gpy/gp_models_utils_casadi.py:43-157), differentiated by hand -- the reference
// leaves these derivatives to CasADi's AD.  With u_j = s_j^2 (x_j - z_j), c = c0 + sum a_j x_j z_j,
// g = kappa'(r)/r, h = g'(r)/r   (RBF: g = -kappa, h = kappa;  Matern-5/2: g = -5/3 (1 + sqrt5 r) e, h = 25/3 e,
// e = exp(-sqrt5 r)):
//   d k/dx_j       = a_j z_j v kappa + c v g u_j + b_j z_j
//   d2 k/dx_j dx_l = v g (a_j z_j u_l + a_l z_l u_j) + c v (h u_j u_l + g s_j^2 delta_jl)
//   d var/dx_j     = 2 (a_j v + b_j) x_j - 2 sum_i G_i d k_i/dx_j ,   G = K_y^-1 k*
template <int DT>
__global__ __launch_bounds__(256) void sr_linearize_general_kernel(sr_lin_args a) {
    constexpr int NH = DT * (DT + 1) / 2;
    constexpr int NACC = DT + NH;
    __shared__ double red[4][NACC];
    const int d = blockIdx.x;
    const int off = a.Np - a.N;
    const double* kp = a.kp + (long)d * SR_KP(a.D);
    const int kind = (int)kp[0];
    const double var = kp[1], c0 = kp[2];
    const double* G = a.g + (long)d * a.Np;
    const double* al = a.alpha + (long)d * a.Np;
    double s2[DT], av[DT], bv[DT], x[DT], acc[NACC];
#pragma unroll
    for (int j = 0; j < DT; ++j) {
        const double sj = (j < a.D) ? kp[3 + j] : 0.0;
        s2[j] = sj * sj;
        av[j] = (j < a.D) ? kp[3 + a.D + j] : 0.0;
        bv[j] = (j < a.D) ? kp[3 + 2 * a.D + j] : 0.0;
        x[j] = (j < a.D) ? a.x[j] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < NACC; ++q) acc[q] = 0.0;
    for (int i = threadIdx.x; i < a.N; i += 256) {
        double z[DT], u[DT], r2 = 0.0, la = 0.0;
#pragma unroll
        for (int j = 0; j < DT; ++j) {
            z[j] = (j < a.D) ? a.Z[(long)i * a.D + j] : 0.0;
            const double df = x[j] - z[j];
            u[j] = s2[j] * df;
            r2 = fma(u[j], df, r2);
            la = fma(av[j] * x[j], z[j], la);
        }
        double kap, g, h;
        if (kind == 0) {
            kap = exp(-0.5 * r2);
            g = -kap;
            h = kap;
        } else {
            const double rr = sqrt(r2);
            const double e = exp(-2.23606797749978969641 * rr);
            kap = (1.0 + 2.23606797749978969641 * rr + (5.0 / 3.0) * r2) * e;
            g = -(5.0 / 3.0) * (1.0 + 2.23606797749978969641 * rr) * e;
            h = (25.0 / 3.0) * e;
        }
        const double pre = (c0 + la) * var;
        const double Gi = G[i + off], w = al[i + off];
        const double vk = var * kap, pg = pre * g, ph = pre * h, vg = var * g;
        int q = DT;
#pragma unroll
        for (int j = 0; j < DT; ++j) {
            const double azj = av[j] * z[j];
            acc[j] = fma(Gi, fma(vk, azj, fma(pg, u[j], bv[j] * z[j])), acc[j]);
#pragma unroll
            for (int c = 0; c < DT; ++c)
                if (c >= j) {
                    double hv = fma(vg, fma(azj, u[c], av[c] * z[c] * u[j]), ph * u[j] * u[c]);
                    if (c == j) hv = fma(pg, s2[j], hv);
                    acc[q] = fma(w, hv, acc[q]);
                    ++q;
                }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < NACC; ++q) {
        double v = acc[q];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) red[wave][q] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int q = DT;
#pragma unroll
        for (int j = 0; j < DT; ++j) {
            if (j < a.D)
                a.jac_var[d * a.D + j] = 2.0 * (av[j] * var + bv[j]) * x[j] -
                                         2.0 * (red[0][j] + red[1][j] + red[2][j] + red[3][j]);
#pragma unroll
            for (int c = 0; c < DT; ++c)
                if (c >= j) {
                    if (j < a.D && c < a.D) {
                        const double hv = red[0][q] + red[1][q] + red[2][q] + red[3][q];
                        a.hess_mu[((long)d * a.D + j) * a.D + c] = hv;
                        a.hess_mu[((long)d * a.D + c) * a.D + j] = hv;
                    }
                    ++q;
                }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Streamed route (any model size, every kernel identifier): ONE pass over U^-1 instead of two.
//   d var/dx_j = d k(x,x)/dx_j - 2 (U^-T dk*/dx_j) . (U^-T k*)
// so the right-hand sides [k*, dk*/dx_1 .. dk*/dx_D] go through the streaming contraction of the prediction
// path (sr_var_small_*, 1 + D <= 16 columns) whose reduce pass forms the dot products with column 0; mean,
// mean-Jacobian and mean-Hessian are plain reductions over the training points done while the columns are written.
//   sr_lin_columns_kernel : grid (Np/256, n_out) -- columns into the K* workspace + partial sums per workgroup
//   sr_lin_final_kernel   : grid (n_out)         -- adds the partials, writes the five outputs
// ------------------------------------------------------------------------------------------------
template <int DT>
__global__ __launch_bounds__(256) void sr_lin_columns_kernel(sr_lin_args a, int tq, double* __restrict__ Ks,
                                                             double* __restrict__ lin_part) {
    constexpr int NH = DT * (DT + 1) / 2;
    constexpr int NACC = 1 + DT + NH;                  // mu, d mu/dx, upper triangle of the Hessian
    __shared__ double red[4][NACC];
    const int d = blockIdx.y;
    const int ip = blockIdx.x * 256 + threadIdx.x;       // padded training index
    const int off = a.Np - a.N;
    const bool valid = ip < a.Np && ip >= off;
    double acc[NACC];
#pragma unroll
    for (int q = 0; q < NACC; ++q) acc[q] = 0.0;
    double col[DT + 1];
#pragma unroll
    for (int c = 0; c <= DT; ++c) col[c] = 0.0;
    if (valid) {
        const int i = ip - off;
        const double w = a.alpha[(long)d * a.Np + ip];
        double z[DT], u[DT], x[DT], r2 = 0.0;
        if (a.kp == nullptr) {                           // ARD-RBF: u_j = (z_j - x_j) / l_j^2
            double il2[DT];
#pragma unroll
            for (int j = 0; j < DT; ++j) {
                const double l = (j < a.D) ? a.ls[d * a.D + j] : 1.0;
                il2[j] = (j < a.D) ? 1.0 / (l * l) : 0.0;
                z[j] = (j < a.D) ? a.Z[(long)i * a.D + j] : 0.0;
                x[j] = (j < a.D) ? sr_lin_x(a, j) : 0.0;
                u[j] = (z[j] - x[j]) * il2[j];
                r2 = fma(u[j], z[j] - x[j], r2);
            }
            const double k = a.sf2[d] * exp(-0.5 * r2);
            col[0] = k;
            acc[0] = w * k;
            int q = 1 + DT;
#pragma unroll
            for (int j = 0; j < DT; ++j) {
                col[1 + j] = k * u[j];
                acc[1 + j] = w * k * u[j];
#pragma unroll
                for (int c = 0; c < DT; ++c)
                    if (c >= j) {
                        double hv = u[j] * u[c];
                        if (c == j) hv -= il2[j];
                        acc[q] = w * k * hv;
                        ++q;
                    }
            }
        } else {                                         // general family, formulas of sr_linearize_general_kernel
            const double* kp = a.kp + (long)d * SR_KP(a.D);
            const int kind = (int)kp[0];
            const double var = kp[1], c0 = kp[2];
            double s2[DT], av[DT], bv[DT], la = 0.0, lb = 0.0;
#pragma unroll
            for (int j = 0; j < DT; ++j) {
                const double sj = (j < a.D) ? kp[3 + j] : 0.0;
                s2[j] = sj * sj;
                av[j] = (j < a.D) ? kp[3 + a.D + j] : 0.0;
                bv[j] = (j < a.D) ? kp[3 + 2 * a.D + j] : 0.0;
                z[j] = (j < a.D) ? a.Z[(long)i * a.D + j] : 0.0;
                x[j] = (j < a.D) ? sr_lin_x(a, j) : 0.0;
                const double df = x[j] - z[j];
                u[j] = s2[j] * df;
                r2 = fma(u[j], df, r2);
                la = fma(av[j] * x[j], z[j], la);
                lb = fma(bv[j] * x[j], z[j], lb);
            }
            double kap, g, h;
            if (kind == 0) {
                kap = exp(-0.5 * r2);
                g = -kap;
                h = kap;
            } else {
                const double rr = sqrt(r2);
                const double e = exp(-2.23606797749978969641 * rr);
                kap = (1.0 + 2.23606797749978969641 * rr + (5.0 / 3.0) * r2) * e;
                g = -(5.0 / 3.0) * (1.0 + 2.23606797749978969641 * rr) * e;
                h = (25.0 / 3.0) * e;
            }
            const double pre = (c0 + la) * var;
            const double k = fma(pre, kap, lb);
            col[0] = k;
            acc[0] = w * k;
            const double vk = var * kap, pg = pre * g, ph = pre * h, vg = var * g;
            int q = 1 + DT;
#pragma unroll
            for (int j = 0; j < DT; ++j) {
                const double azj = av[j] * z[j];
                const double dk = fma(vk, azj, fma(pg, u[j], bv[j] * z[j]));
                col[1 + j] = dk;
                acc[1 + j] = w * dk;
#pragma unroll
                for (int c = 0; c < DT; ++c)
                    if (c >= j) {
                        double hv = fma(vg, fma(azj, u[c], av[c] * z[c] * u[j]), ph * u[j] * u[c]);
                        if (c == j) hv = fma(pg, s2[j], hv);
                        acc[q] = w * hv;
                        ++q;
                    }
            }
        }
    }
    if (ip < a.Np) {
        double* dst = Ks + ((long)d * a.Np + ip) * a.Tp;
#pragma unroll
        for (int c = 0; c <= DT; ++c)
            if (c < tq) dst[c] = (c <= a.D) ? col[c] : 0.0;
        for (int c = DT + 1; c < tq; ++c) dst[c] = 0.0;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < NACC; ++q) {
        double v = acc[q];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) red[wave][q] = v;
    }
    __syncthreads();
    if (threadIdx.x < NACC)
        lin_part[((long)d * gridDim.x + blockIdx.x) * NACC + threadIdx.x] =
            red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

__global__ __launch_bounds__(256) void sr_lin_final_kernel(sr_lin_args a, const double* __restrict__ lin_part,
                                                          int nblk, int DT, const double* __restrict__ dots, int ncb,
                                                          double* __restrict__ mu, double* __restrict__ var,
                                                          double* __restrict__ jac_mu) {
    __shared__ double sh[91 + 13];
    sr_lin_final_dev(a, lin_part, nblk, DT, dots, ncb, mu, var, jac_mu, blockIdx.x, threadIdx.x, sh);
}

int sr_launch_lin_columns(const sr_lin_args& a, int tq, double* Ks, double* lin_part, hipStream_t s) {
    dim3 grid((a.Np + 255) / 256, a.n_out);
#define SR_LC_CASE(DT) hipLaunchKernelGGL(sr_lin_columns_kernel<DT>, grid, dim3(256), 0, s, a, tq, Ks, lin_part)
    if (a.D <= 3) SR_LC_CASE(3);
    else if (a.D <= 5) SR_LC_CASE(5);
    else if (a.D <= 8) SR_LC_CASE(8);
    else if (a.D <= 12) SR_LC_CASE(12);
    else { sr_set_error("linearize: D=%d > %d", a.D, SR_MAX_D); return SR_EUNSUPPORTED; }
#undef SR_LC_CASE
    SR_HIP(hipGetLastError());
    return SR_OK;
}

int sr_lin_nacc(int D) {
    const int DT = D <= 3 ? 3 : (D <= 5 ? 5 : (D <= 8 ? 8 : 12));
    return 1 + DT + DT * (DT + 1) / 2;
}

int sr_launch_lin_final(const sr_lin_args& a, const double* lin_part, const double* dots, int ncb, double* mu,
                        double* var, double* jac_mu, hipStream_t s) {
    const int nblk = (a.Np + 255) / 256;
    // NACC <= 91 (D = 12) and D*D <= 144 threads are needed: one block of 256 covers every case
    const int DT = a.D <= 3 ? 3 : (a.D <= 5 ? 5 : (a.D <= 8 ? 8 : 12));
    hipLaunchKernelGGL(sr_lin_final_kernel, dim3(a.n_out), dim3(256), 0, s, a, lin_part, nblk, DT, dots, ncb, mu, var,
                       jac_mu);
    SR_HIP(hipGetLastError());
    return SR_OK;
}

int sr_launch_linearize(const sr_lin_args& a, hipStream_t s) {
    dim3 grid(a.n_out);
    if (a.kp) {
#define SR_LING_CASE(DT) hipLaunchKernelGGL(sr_linearize_general_kernel<DT>, grid, dim3(256), 0, s, a)
        if (a.D <= 3) SR_LING_CASE(3);
        else if (a.D <= 5) SR_LING_CASE(5);
        else if (a.D <= 8) SR_LING_CASE(8);
        else if (a.D <= 12) SR_LING_CASE(12);
        else { sr_set_error("linearize: D=%d > %d", a.D, SR_MAX_D); return SR_EUNSUPPORTED; }
#undef SR_LING_CASE
        SR_HIP(hipGetLastError());
        return SR_OK;
    }
#define SR_LIN_CASE(DT) hipLaunchKernelGGL(sr_linearize_kernel<DT>, grid, dim3(256), 0, s, a)
    if (a.D <= 3) SR_LIN_CASE(3);
    else if (a.D <= 5) SR_LIN_CASE(5);
    else if (a.D <= 8) SR_LIN_CASE(8);
    else if (a.D <= 12) SR_LIN_CASE(12);
    else { sr_set_error("linearize: D=%d > %d", a.D, SR_MAX_D); return SR_EUNSUPPORTED; }
#undef SR_LIN_CASE
    SR_HIP(hipGetLastError());
    return SR_OK;
}
