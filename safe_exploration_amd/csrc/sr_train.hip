// sr_train.hip -- negative log marginal likelihood of the factorised model and its gradient with respect to
// the packed kernel parameters and the noise: the objective `model_gp.optimize()` minimises inside
// SimpleGPModel.train(opt_hyp=True)  (/root/reference/safe_exploration/ssm_gpy/gaussian_process.py:249-250;
// the optimiser itself lives in GPy and is not restated -- the host drives L-BFGS-B over these values).
//
//   nll   = 1/2 y^T alpha + 1/2 log det K_y + N/2 log 2 pi
//   d nll / d theta = 1/2 sum_ij M_ij dK_ij/d theta ,   M = K_y^-1 - alpha alpha^T
// with the general kernel family of sr_common.h, k = (c0 + sum a x y) v kappa(r) + sum b x y:
//   dk/dv = c kappa,  dk/dc0 = v kappa,  dk/ds_j = c v g s_j (x_j - y_j)^2  (g = kappa'(r)/r),
//   dk/da_j = x_j y_j v kappa,  dk/db_j = x_j y_j,  dK_y/dnoise = I.
// One thread per (i, j) pair of a 16 x 16 tile, deterministic two-stage reduction.
#include "sr_common.h"

template <int DT>
__global__ __launch_bounds__(256) void sr_mll_grad_kernel(const double* __restrict__ Kinv, int Np, int N,
                                                          const double* __restrict__ alpha,
                                                          const double* __restrict__ Z,
                                                          const double* __restrict__ kp, int D,
                                                          double* __restrict__ partial) {
    constexpr int NACC = 3 + 3 * DT;               // v, c0, s[DT], a[DT], b[DT], noise
    __shared__ double red[4][NACC];
    const int off = Np - N;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int i = blockIdx.y * 16 + ty, j = blockIdx.x * 16 + tx;
    const int kind = (int)kp[0];
    const double var = kp[1], c0 = kp[2];
    double acc[NACC];
#pragma unroll
    for (int q = 0; q < NACC; ++q) acc[q] = 0.0;
    if (i < N && j < N) {
        const double m = 0.5 * (Kinv[(long)(i + off) * Np + (j + off)] - alpha[i + off] * alpha[j + off]);
        double r2 = 0.0, la = 0.0;
        double xy[DT], sd2[DT];
#pragma unroll
        for (int c = 0; c < DT; ++c) {
            const double x = (c < D) ? Z[(long)i * D + c] : 0.0;
            const double y = (c < D) ? Z[(long)j * D + c] : 0.0;
            const double s = (c < D) ? kp[3 + c] : 0.0;
            const double a = (c < D) ? kp[3 + D + c] : 0.0;
            const double df = x - y;
            xy[c] = x * y;
            sd2[c] = s * df * df;                      // s_c (x_c - y_c)^2
            r2 = fma(s * sd2[c], 1.0, r2);
            la = fma(a, xy[c], la);
        }
        double kap, g;
        if (kind == 0) {
            kap = exp(-0.5 * r2);
            g = -kap;
        } else {
            const double rr = sqrt(r2);
            const double e = exp(-2.23606797749978969641 * rr);
            kap = (1.0 + 2.23606797749978969641 * rr + (5.0 / 3.0) * r2) * e;
            g = -(5.0 / 3.0) * (1.0 + 2.23606797749978969641 * rr) * e;
        }
        const double cc = c0 + la;
        acc[0] = m * cc * kap;
        acc[1] = m * var * kap;
        const double mpg = m * cc * var * g, mvk = m * var * kap;
#pragma unroll
        for (int c = 0; c < DT; ++c) {
            acc[2 + c] = mpg * sd2[c];
            acc[2 + DT + c] = mvk * xy[c];
            acc[2 + 2 * DT + c] = m * xy[c];
        }
        acc[NACC - 1] = (i == j) ? m : 0.0;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < NACC; ++q) {
        double v = acc[q];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) red[wave][q] = v;
    }
    __syncthreads();
    if (threadIdx.x < NACC) {
        const long b = (long)blockIdx.y * gridDim.x + blockIdx.x;
        partial[b * NACC + threadIdx.x] =
            red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    }
}

// grad[q] = sum_blocks partial[b][q] (fixed order);  nll = 1/2 y^T alpha + 1/2 logdet + N/2 log 2 pi
__global__ __launch_bounds__(256) void sr_mll_reduce_kernel(const double* __restrict__ partial, long nblocks,
                                                            int nacc, int DT, int D,
                                                            const double* __restrict__ yT,
                                                            const double* __restrict__ alpha, int Np, int N,
                                                            const double* __restrict__ logdet,
                                                            double* __restrict__ nll, double* __restrict__ grad) {
    __shared__ double red[4];
    const int q = blockIdx.x;                      // 0 .. nacc-1: gradient entries; nacc: y^T alpha
    double v = 0.0;
    if (q < nacc) {
        for (long b = threadIdx.x; b < nblocks; b += 256) v += partial[b * nacc + q];
    } else {
        const int off = Np - N;
        for (int i = threadIdx.x; i < N; i += 256) v = fma(yT[i + off], alpha[i + off], v);
    }
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double tot = red[0] + red[1] + red[2] + red[3];
        if (q == nacc) {
            *nll = 0.5 * tot + 0.5 * logdet[0] + 0.5 * N * 1.83787706640934548356;   // log(2 pi)
        } else {
            // compact the DT-strided accumulators to the D-strided API layout [v, c0, s[D], a[D], b[D], noise]
            int dst = -1;
            if (q < 2) dst = q;
            else if (q == nacc - 1) dst = 2 + 3 * D;
            else {
                const int grp = (q - 2) / DT, c = (q - 2) % DT;
                if (c < D) dst = 2 + grp * D + c;
            }
            if (dst >= 0) grad[dst] = tot;
        }
    }
}

long sr_mll_ws(int N) {
    const long nb = ((long)N + 15) / 16;
    return nb * nb * (3 + 3 * 12);
}

int sr_launch_mll(const double* Kinv, int Np, int N, const double* alpha, const double* yT, const double* Z,
                  const double* kp, int D, const double* logdet, double* partial, double* nll, double* grad,
                  hipStream_t s) {
    const int nb = (N + 15) / 16;
    dim3 grid(nb, nb);
    int DT;
#define SR_MLL_CASE(DTV) DT = DTV; hipLaunchKernelGGL(sr_mll_grad_kernel<DTV>, grid, dim3(256), 0, s, Kinv, Np, N, alpha, Z, kp, D, partial)
    if (D <= 3) { SR_MLL_CASE(3); }
    else if (D <= 5) { SR_MLL_CASE(5); }
    else if (D <= 8) { SR_MLL_CASE(8); }
    else if (D <= 12) { SR_MLL_CASE(12); }
    else { sr_set_error("mll: D=%d > %d", D, SR_MAX_D); return SR_EUNSUPPORTED; }
#undef SR_MLL_CASE
    SR_HIP(hipGetLastError());
    const int nacc = 3 + 3 * DT;
    hipLaunchKernelGGL(sr_mll_reduce_kernel, dim3(nacc + 1), dim3(256), 0, s, partial, (long)nb * nb, nacc, DT, D,
                       yT, alpha, Np, N, logdet, nll, grad);
    SR_HIP(hipGetLastError());
    return SR_OK;
}
