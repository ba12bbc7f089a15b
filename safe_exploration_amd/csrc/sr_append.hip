// sr_append.hip -- kernels of the block row append of U^-1 (sr_gp_append, sr_capi_append.hip):
// update_model(x, y, opt_hyp=False, replace_old=False), /root/reference/safe_exploration/ssm_gpy/gaussian_process.py:347-419
// (the reference refactorises; its row-append sketch: ssm_pytorch/utilities.py:74-117).
#include "sr_mfma_tile.h"
#include <atomic>
#include "sr_pivot_dev.h"
#include "sr_final_dev.h"

// ---- helpers of the block row-append update (sr_gp_append) --------------------------------------
// S[r][c] -= G[r][c] on the real block r, c >= pf of a front-padded 128 x 128 tile
__global__ __launch_bounds__(256) void sr_sub_block_kernel(double* __restrict__ S, const double* __restrict__ G,
                                                           int pf) {
    for (int idx = threadIdx.x; idx < SR_NB * SR_NB; idx += 256) {
        const int r = idx >> 7, c = idx & 127;
        if (r >= pf && c >= pf) S[idx] -= G[idx];
    }
}

int sr_launch_sub_block(double* S, const double* G, int pf, hipStream_t s) {
    hipLaunchKernelGGL(sr_sub_block_kernel, dim3(1), dim3(256), 0, s, S, G, pf);
    SR_HIP(hipGetLastError());
    return SR_OK;
}

// new U^-1 (Np1 x Np1, front padding off1) from the old one, the new off-diagonal columns
// Y2 (= -U^-1 U12 U22^-1, rows in OLD padded indexing, 128 front-padded columns) and U22^-1 (invS)
__global__ __launch_bounds__(256) void sr_append_assemble_kernel(const double* __restrict__ Wt0, int Np0,
                                                                 int off0, int N0, const double* __restrict__ Y2,
                                                                 const double* __restrict__ invS, int m,
                                                                 double* __restrict__ Wt1, int Np1, int off1) {
    const int r = blockIdx.y;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= Np1) return;
    const int pf = SR_NB - m;
    double v;
    if (r < off1 || c < off1) {
        v = (r == c) ? 1.0 : 0.0;
    } else {
        const int i = r - off1, j = c - off1;
        if (i < N0 && j < N0) v = Wt0[(long)(off0 + i) * Np0 + off0 + j];
        else if (i < N0) v = Y2[(long)(off0 + i) * SR_NB + pf + (j - N0)];
        else if (j >= N0) v = invS[(pf + i - N0) * SR_NB + pf + (j - N0)];
        else v = 0.0;
    }
    Wt1[(long)r * Np1 + c] = v;
}

int sr_launch_append_assemble(const double* Wt0, int Np0, int off0, int N0, const double* Y2,
                              const double* invS, int m, double* Wt1, int Np1, int off1, hipStream_t s) {
    hipLaunchKernelGGL(sr_append_assemble_kernel, dim3((Np1 + 255) / 256, Np1), dim3(256), 0, s, Wt0, Np0, off0,
                       N0, Y2, invS, m, Wt1, Np1, off1);
    SR_HIP(hipGetLastError());
    return SR_OK;
}

// ---- row append with FEW new points (m <= 16): matrix-vector shaped kernels, every pass over U^-1 is one
// coalesced stream (the GEMM route pads the m columns to a 128-wide tile and serialises 40 workgroups).
// U12t[a][i] (a < m, i < Np0, padded row indexing) holds U12 = U^-T K(Z_old, Z_new) column by column.

// G[pf+a][pf+b] = sum_i U12t[a][i] U12t[b][i] inside a zeroed 128 x 128 block; grid (m, m)
__global__ __launch_bounds__(256) void sr_append_gsmall_kernel(const double* __restrict__ U12t, int Np0, int m,
                                                               double* __restrict__ G) {
    __shared__ double red[4];
    const int a = blockIdx.x, b = blockIdx.y, pf = SR_NB - m;
    U12t += (long)blockIdx.z * m * Np0; G += (long)blockIdx.z * SR_NB * SR_NB;      // batch member (output)
    double v = 0.0;
    for (int i = threadIdx.x; i < Np0; i += 256) v = fma(U12t[(long)a * Np0 + i], U12t[(long)b * Np0 + i], v);
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) G[(pf + a) * SR_NB + pf + b] = red[0] + red[1] + red[2] + red[3];
}

// Xt[c][i] = sum_{a <= c} U12t[a][i] invS[pf+a][pf+c]   (X = U12 U22^-1, U22^-1 upper triangular)
__global__ __launch_bounds__(256) void sr_append_xt_kernel(const double* __restrict__ U12t,
                                                           const double* __restrict__ invS, int Np0, int m,
                                                           double* __restrict__ Xt, long sXt) {
    const int i = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y, pf = SR_NB - m;
    U12t += (long)blockIdx.z * m * Np0; invS += (long)blockIdx.z * SR_NB * SR_NB; Xt += (long)blockIdx.z * sXt;
    if (i >= Np0) return;
    double v = 0.0;
    for (int a = 0; a <= c; ++a) v = fma(U12t[(long)a * Np0 + i], invS[(pf + a) * SR_NB + pf + c], v);
    Xt[(long)c * Np0 + i] = v;
}

// Y2[i][pf+c] = -sum_{k >= i} Wt0[i][k] Xt[c][k]  (Y2 = -U^-1 X): one wavefront per row, lanes over k, MC columns
template <int MC>
__global__ __launch_bounds__(256) void sr_append_y2_kernel(const double* __restrict__ Wt0, int Np0,
                                                           const double* __restrict__ Xt, int m,
                                                           double* __restrict__ Y2) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63, pf = SR_NB - m;
    if (row >= Np0) return;
    double acc[MC];
#pragma unroll
    for (int c = 0; c < MC; ++c) acc[c] = 0.0;
    for (int k = row + lane; k < Np0; k += 64) {
        const double w = Wt0[(long)row * Np0 + k];
#pragma unroll
        for (int c = 0; c < MC; ++c)
            if (c < m) acc[c] = fma(w, Xt[(long)c * Np0 + k], acc[c]);
    }
#pragma unroll
    for (int c = 0; c < MC; ++c) {
        double v = acc[c];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0 && c < m) Y2[(long)row * SR_NB + pf + c] = -v;
    }
}

// The new columns and the move of the old factor in ONE pass: wavefront per OLD padded row `row` (real index i):
//   acc[c]      = sum_{k >= row} Wt0[row][k] Xt[c][k]                       (Y2 = -U^-1 X, as sr_append_y2_kernel)
//   Wt1[r][..]  = the same row, shifted to the new padding (r = off1 + i), upper part only, + the m new entries -acc
// and one extra workgroup writes the rows of the new points (U22^-1).  Everything below the diagonal of Wt1 and its
// identity padding must already be in place (a buffer that held an earlier state of the same model, or zeroed +
// sr_launch_eye_front): the copy through sr_append_assemble_kernel read and wrote the full square, zeros included --
// 525 MB per output and append at N = 5000 against 210 MB here.
template <int MC>
__global__ __launch_bounds__(256) void sr_append_move_kernel(const double* __restrict__ Wt0, int Np0, int off0, int N0,
                                                             const double* __restrict__ Xt,
                                                             const double* __restrict__ invS, int m,
                                                             double* __restrict__ Y2, double* __restrict__ Wt1,
                                                             int Np1, int off1, long sXt, long sY2,
                                                             const double* __restrict__ U12one) {
    // U12one != NULL (MC == 1, one new point): Xt[0][k] = U12t[0][k] U22^-1 is formed here, Xt is not read
    const int lane = threadIdx.x & 63, pf = SR_NB - m;
    {                                                       // batch member (output)
        const long b = blockIdx.y;
        Wt0 += b * Np0 * Np0; Xt += b * sXt; invS += b * SR_NB * SR_NB; Y2 += b * sY2; Wt1 += b * Np1 * Np1;
        if (U12one) U12one += b * Np0;
    }
    const double inv00 = (MC == 1 && U12one) ? invS[pf * SR_NB + pf] : 0.0;
    const int nrow_blocks = (Np0 + 3) / 4;
    if ((int)blockIdx.x >= nrow_blocks) {
        // rows of the new points: Wt1[off1 + N0 + q][off1 + N0 + c] = U22^-1[q][c]
        for (int e = threadIdx.x; e < m * m; e += 256) {
            const int q = e / m, c = e % m;
            if (c >= q) Wt1[(long)(off1 + N0 + q) * Np1 + off1 + N0 + c] = invS[(pf + q) * SR_NB + pf + c];
        }
        return;
    }
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= Np0 || row < off0) return;                   // old padding rows carry nothing
    const int shift = off1 - off0;                          // new column index = old + shift
    double* dst = Wt1 + (long)(row + shift) * Np1 + shift;
    double acc[MC];
#pragma unroll
    for (int c = 0; c < MC; ++c) acc[c] = 0.0;
    for (int k = row + lane; k < Np0; k += 64) {
        const double w = Wt0[(long)row * Np0 + k];
        dst[k] = w;
        if (MC == 1 && U12one) {
            acc[0] = fma(w, U12one[k] * inv00, acc[0]);
        } else {
#pragma unroll
            for (int c = 0; c < MC; ++c)
                if (c < m) acc[c] = fma(w, Xt[(long)c * Np0 + k], acc[c]);
        }
    }
#pragma unroll
    for (int c = 0; c < MC; ++c) {
        double v = acc[c];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0 && c < m) {
            Y2[(long)row * SR_NB + pf + c] = -v;
            dst[Np0 + c] = -v;                               // column off1 + N0 + c of the new matrix
        }
    }
}

// ones on the first n diagonal entries (identity padding of a zeroed matrix)
__global__ __launch_bounds__(256) void sr_eye_front_kernel(double* __restrict__ W, int ld, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    W += (long)blockIdx.y * ld * ld;                        // batch member: ld x ld matrices back to back
    if (i < n) W[(long)i * ld + i] = 1.0;
}

int sr_launch_eye_front(double* W, int ld, int n, hipStream_t s, int nbatch) {
    if (n <= 0) return SR_OK;
    hipLaunchKernelGGL(sr_eye_front_kernel, dim3((n + 255) / 256, nbatch), dim3(256), 0, s, W, ld, n);
    SR_HIP(hipGetLastError());
    return SR_OK;
}

// nbatch outputs in one launch each: U12t (m x Np0 each), invS (128 x 128), Wt0 / Wt1 (full squares) back to back, Xt and Y2
// with the strides given
int sr_launch_append_move(const double* Wt0, int Np0, int off0, int N0, const double* U12t, const double* invS, int m,
                          double* Xt, double* Y2, double* Wt1, int Np1, int off1, hipStream_t s, int nbatch, long sXt,
                          long sY2) {
    const dim3 grid((Np0 + 3) / 4 + 1, nbatch);
    if (m <= 1) {                                           // one new point: X = U12 U22^-1 is a scaling, done in the move
        hipLaunchKernelGGL(sr_append_move_kernel<1>, grid, dim3(256), 0, s, Wt0, Np0, off0, N0, Xt, invS, m, Y2, Wt1, Np1, off1, sXt,
                           sY2, U12t);
        SR_HIP(hipGetLastError());
        return SR_OK;
    }
    hipLaunchKernelGGL(sr_append_xt_kernel, dim3((Np0 + 255) / 256, m, nbatch), dim3(256), 0, s, U12t, invS, Np0, m, Xt, sXt);
    SR_HIP(hipGetLastError());
    if (m <= 4)
        hipLaunchKernelGGL(sr_append_move_kernel<4>, grid, dim3(256), 0, s, Wt0, Np0, off0, N0, Xt, invS, m, Y2, Wt1, Np1, off1, sXt,
                           sY2, (const double*)nullptr);
    else
        hipLaunchKernelGGL(sr_append_move_kernel<16>, grid, dim3(256), 0, s, Wt0, Np0, off0, N0, Xt, invS, m, Y2, Wt1, Np1, off1, sXt,
                           sY2, (const double*)nullptr);
    SR_HIP(hipGetLastError());
    return SR_OK;
}

// ------------------------------------------------------------------------------------------------
// ONE new point on a SMALL model (ARD-RBF or the general kernel family; old padded size <= 512, new <= 640: the reference's own regime, a transition
// appended after every step of its exploration loop, exploration_runner.py:186-188) -- the whole append in ONE launch,
// one workgroup of 16 wavefronts per output:
//   b = K(Z_old, z_new), mu_old = b . alpha0, u12 = U^-T b (thread = column, 4 k-slices), s = sf2 + noise - |u12|^2,
//   u22^-1 = 1 / sqrt(s), X = u12 u22^-1;
//   then the new factor row by row (wavefront = row), written IN FULL (zeros below the diagonal, identity padding: the
//   target buffer needs no preparation): the old row moved to the new padding, its new last entry
//   y2 = -sum_{k >= row} U^-1[row][k] X[k] from the same pass, alpha1 = alpha0 + y2 v2 (v2 = u22^-1 (y_new - mu_old)),
//   the shifted targets, the new point's row (u22^-1), log det of the new factor (SR_APPEND1_WGS partial sums per
//   output), the failure word.
// The general route does this in 10 launches (65 us inside sr_gp_append at any size up to N ~ 1000); the arithmetic is
// the same (same sums in another order: the tests compare both routes with the refit and the CPU restatement).
// ------------------------------------------------------------------------------------------------
struct sr_append1_args {
    const double* Wt0; const double* alpha0; const double* yT0; const double* Z;     // old state (Z: N0 x D)
    const double* ls; const double* sf2; const double* noise;                         // n_out x D, n_out, n_out
    const double* kp;                                                                 // general kernels: n_out x SR_KP(D), else NULL
    const double* znew; const double* ynew;                                           // D, n_out
    double* Wt1; double* alpha1; double* yT1; double* Zdst;                           // new state (Zdst: row N0 of Z, or NULL)
    double* logdet; int* info;                                                        // n_out each
    int N0, Np0, Np1, D, n_out;
    // the new point in the kernel arguments (a host caller: no H2D copy command); znew / ynew are NULL then
    int inl; double xin[SR_MAX_D]; double yin[SR_APPEND1_MAX_OUT];
};

template <int NPMAX>   // 256 or 512: the old padded size it serves
__global__ __launch_bounds__(1024) void sr_append1_small_kernel(sr_append1_args a) {
    __shared__ double b[NPMAX], u12[NPMAX], X[NPMAX], part[4][NPMAX], red[16];
    __shared__ double s_mu, s_inv, s_v2;
    __shared__ double zn[SR_MAX_D];                          // the new input (from memory or from the kernel arguments)
    const int d = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N0 = a.N0, Np0 = a.Np0, Np1 = a.Np1, D = a.D;
    if (tid < D) zn[tid] = a.inl ? a.xin[tid] : a.znew[tid];
    const double y_new = a.inl ? a.yin[d] : a.ynew[d];
    __syncthreads();
    auto sr_wave_sum = [](double v) {
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        return v;
    };
    const int off0 = Np0 - N0, off1 = Np1 - (N0 + 1), shift = off1 - off0;
    const double* Wt0 = a.Wt0 + (long)d * Np0 * Np0;
    const double* alpha0 = a.alpha0 + (long)d * Np0;
    double* Wt1 = a.Wt1 + (long)d * Np1 * Np1;
    // gridDim.y workgroups per output share the rows of the new factor (each of them repeats the cheap first part: one
    // workgroup alone writes a 256-row factor in 25 us, four take 8); workgroup y = 0 also reports failure and copies z_new
    const int wy = blockIdx.y, nwy = gridDim.y;
    if (d == 0 && wy == 0 && a.Zdst && tid < D) a.Zdst[tid] = zn[tid];
    // ---- b = K(Z_old, z_new) in padded row indexing, mu_old = b . alpha0
    double mu_t = 0.0;
#pragma unroll 1
    for (int row = tid; row < NPMAX; row += 1024) {
        double v = 0.0;
        if (row < Np0 && row >= off0) {
            const double* z = a.Z + (long)(row - off0) * D;
            if (a.kp) {                                      // general family (sr_common.h), as sr_gram_general_kernel
                const double* kp = a.kp + (long)d * SR_KP(D);
                const double *sv = kp + 3, *av = kp + 3 + D, *bv = kp + 3 + 2 * D;
                double r2 = 0.0, la = 0.0, lb = 0.0;
                for (int c = 0; c < D; ++c) {
                    const double t = (z[c] - zn[c]) * sv[c];
                    r2 = fma(t, t, r2);
                    la = fma(av[c] * z[c], zn[c], la);
                    lb = fma(bv[c] * z[c], zn[c], lb);
                }
                v = (kp[2] + la) * kp[1] * sr_kappa((int)kp[0], r2) + lb;
            } else {
                double r2 = 0.0;
                for (int c = 0; c < D; ++c) {
                    const double t = (z[c] - zn[c]) / a.ls[d * D + c];
                    r2 = fma(t, t, r2);
                }
                v = a.sf2[d] * exp(-0.5 * r2);
            }
            mu_t += v * alpha0[row];
        }
        b[row] = v;
    }
    auto block_sum = [&](double v) {                         // fixed order: wavefront sums, then wavefront 0 .. 15
        const double w = sr_wave_sum(v);
        __syncthreads();                                     // (red may still be read from the previous sum)
        if (lane == 0) red[wave] = w;
        __syncthreads();
        double t = 0.0;
        for (int k = 0; k < 16; ++k) t += red[k];
        return t;
    };
    const double mu_old = block_sum(mu_t);
    if (tid == 0) s_mu = mu_old;
    // ---- u12[i] = sum_{k <= i} U^-1[k][i] b[k]: work item (slice q of NPMAX / 4 rows, column i), 16 loads in flight
    constexpr int SLICE = NPMAX / 4;
#pragma unroll 1
    for (int wi = tid; wi < 4 * NPMAX; wi += 1024) {
        const int q = wi / NPMAX, i = wi % NPMAX;
        double acc = 0.0;
        if (i < Np0) {
            constexpr int UB = (NPMAX == 256) ? 16 : 8;      // loads in flight (the 512 form would spill with 16)
            const int k_end = min(q * SLICE + SLICE - 1, i);
            for (int k0 = q * SLICE; k0 <= k_end; k0 += UB) {
                double w[UB];
#pragma unroll
                for (int u = 0; u < UB; ++u) w[u] = (k0 + u <= k_end) ? Wt0[(long)(k0 + u) * Np0 + i] : 0.0;
#pragma unroll
                for (int u = 0; u < UB; ++u) acc = fma(w[u], b[min(k0 + u, NPMAX - 1)], acc);
            }
        }
        part[q][i] = acc;
    }
    __syncthreads();
    double g_t = 0.0;
    for (int i = tid; i < NPMAX; i += 1024) {
        const double v = (part[0][i] + part[1][i]) + (part[2][i] + part[3][i]);
        u12[i] = v;
        g_t = fma(v, v, g_t);
    }
    const double g = block_sum(g_t);
    if (tid == 0) {
        double prior;                                        // k(z_new, z_new)
        if (a.kp) {
            const double* kp = a.kp + (long)d * SR_KP(D);
            double la = 0.0, lb = 0.0;
            for (int c = 0; c < D; ++c) {
                la = fma(kp[3 + D + c] * zn[c], zn[c], la);
                lb = fma(kp[3 + 2 * D + c] * zn[c], zn[c], lb);
            }
            prior = (kp[2] + la) * kp[1] + lb;               // kappa(0) = 1
        } else {
            prior = a.sf2[d];
        }
        double sch = prior + a.noise[d] - g;                 // Schur complement of the new point
        if (!(sch > 0.0)) {                                  // also catches NaN
            if (wy == 0) a.info[d] = N0 + 1;
            sch = 1.0;
        } else if (wy == 0) {
            a.info[d] = 0;
        }
        double sd, inv;
        sr_sqrt_rsqrt(sch, sd, inv);
        s_inv = inv;
        s_v2 = inv * (y_new - s_mu);
    }
    __syncthreads();
    const double inv = s_inv, v2 = s_v2;
    for (int i = tid; i < NPMAX; i += 1024) X[i] = u12[i] * inv;
    __syncthreads();
    // ---- the new factor, alpha and targets, row by row
    double ld = 0.0;                                         // sum of log(diagonal) over this wavefront's rows (lane 0)
    const double* yT0 = a.yT0 + (long)d * Np0;
    double* alpha1 = a.alpha1 + (long)d * Np1;
    double* yT1 = a.yT1 + (long)d * Np1;
    const int Rlast = Np1 - 1;                               // row of the new point
    for (int R = wy * 16 + wave; R < Np1; R += 16 * nwy) {
        double* dst = Wt1 + (long)R * Np1;
        if (R < off1 || R == Rlast) {
            const double dg = (R == Rlast) ? inv : 1.0;
            for (int C = lane; C < Np1; C += 64) dst[C] = (C == R) ? dg : 0.0;
            if (lane == 0) {
                alpha1[R] = (R == Rlast) ? inv * v2 : 0.0;
                yT1[R] = (R == Rlast) ? y_new : 0.0;
                if (R == Rlast) ld += log(inv);
            }
            continue;
        }
        const int r0 = R - shift;                            // old padded row
        const double* src = Wt0 + (long)r0 * Np0;
        double acc = 0.0, dgv = 1.0;
        for (int C = lane; C < Rlast; C += 64) {
            double v = 0.0;
            if (C >= R) {
                v = src[C - shift];
                acc = fma(v, X[C - shift], acc);
                if (C == R) dgv = v;
            }
            dst[C] = v;
        }
        acc = sr_wave_sum(acc);
        dgv = __shfl(dgv, R & 63);                           // the lane that held the diagonal entry
        if (lane == 0) {
            dst[Rlast] = -acc;
            alpha1[R] = fma(-acc, v2, alpha0[r0]);
            yT1[R] = yT0[r0];
            ld += log(dgv);
        }
    }
    __syncthreads();
    if (lane == 0) red[wave] = ld;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int w = 0; w < 16; ++w) t += red[w];
        a.logdet[d * nwy + wy] = -2.0 * t;                  // partial sums: the host adds the gridDim.y of an output
    }
}

int sr_launch_append1_small(const double* Wt0, const double* alpha0, const double* yT0, const double* Z, const double* ls,
                            const double* sf2, const double* noise, const double* kp, const double* znew, const double* ynew, double* Wt1,
                            double* alpha1, double* yT1, double* Zdst, double* logdet, int* info, int N0, int Np0, int Np1,
                            int D, int n_out, hipStream_t s, const double* x_host, const double* y_host) {
    SR_CHECK(Np0 <= 512 && Np1 <= 640 && N0 >= 1 && N0 <= Np0, SR_EINVAL, "append1_small: Np0 = %d, Np1 = %d", Np0, Np1);
    sr_append1_args a{Wt0, alpha0, yT0, Z, ls, sf2, noise, kp, znew, ynew, Wt1, alpha1, yT1, Zdst, logdet, info, N0, Np0, Np1, D, n_out,
                      0, {}, {}};
    if (x_host) {                                            // the new point travels in the kernel arguments
        SR_CHECK(y_host && D <= SR_MAX_D && n_out <= SR_APPEND1_MAX_OUT, SR_EINVAL, "append1_small: D = %d, n_out = %d", D, n_out);
        a.inl = 1; a.znew = nullptr; a.ynew = nullptr;
        for (int c = 0; c < D; ++c) a.xin[c] = x_host[c];
        for (int d = 0; d < n_out; ++d) a.yin[d] = y_host[d];
    }
    if (Np0 <= 256) hipLaunchKernelGGL(sr_append1_small_kernel<256>, dim3(n_out, SR_APPEND1_WGS), dim3(1024), 0, s, a);
    else hipLaunchKernelGGL(sr_append1_small_kernel<512>, dim3(n_out, SR_APPEND1_WGS), dim3(1024), 0, s, a);
    SR_HIP(hipGetLastError());
    return SR_OK;
}

// ------------------------------------------------------------------------------------------------
// ONE new point on a model of up to SR_APPEND1G_MAX_NP0 padded rows, in ONE launch of a GRID of workgroups (the general
// route below takes 10 dependent launches: 93 us of mostly launch latency at N = 1000, where the arithmetic needs 15).
// gridDim = (n_out, W); the W workgroups of an output meet twice at a device-wide barrier (a counter in device memory that
// only grows: the arrivals of all launches so far + W, + 2 W as targets).  Every workgroup of the grid must be resident at
// once: one fits a CU (73 KB of LDS, 1024 threads, ~100 registers), the host keeps n_out W below 7/8 of the CUs.
//   every workgroup:  b = K(Z_old, z_new) and mu_old = b . alpha0 (N kernel evaluations: cheaper than handing them over)
//   phase 1  workgroup w: vp[w][c] = sum over ITS rows k = off0 + w, + W, .. <= c of U^-1[k][c] b[k]   (rows read as rows)
//   -- barrier --
//   phase 2  256 columns at a time: u12[c] = sum_w vp[w][c] in the order of w, gpart[cb] = sum_c u12[c]^2
//   -- barrier --
//   every workgroup:  s = k(z, z) + noise - sum_cb gpart[cb], u22^-1 = 1 / sqrt(s), X = u12 u22^-1 (LDS), v2
//   phase 3  the rows of the new factor, wavefront per row as in sr_append_move_kernel: the old row moved to the new
//            padding (upper part only: the target holds zeros below the diagonal and its identity padding, see the host),
//            its new last entry y2 = -sum_{k >= row} U^-1[row][k] X[k], alpha1 = alpha0 + y2 v2, the shifted targets,
//            the new point's row, sum of log(diagonal) per workgroup.
// What one workgroup hands to another goes out with agent-scope stores and comes in with agent-scope loads (the L2s of
// the XCDs are not coherent with each other); the barrier itself is a relaxed agent-scope counter.
// IN PLACE (`inplace`, the padded size stays): the new factor is the OLD MEMORY seen from Wt0 + Np0 + 1 with the same leading
// dimension -- element (r, c) of that view is element (r + 1, c + 1) of the old one, which is exactly where the move would
// put it -- so phase 3 only READS the rows (for y2) and writes the new column, the new point's diagonal entry, alpha and
// the new target; nothing is written unless the pivots of ALL outputs are positive (the barriers are grid-wide for that).
// The last column of the view wraps into column 0 of the rows two below (lower triangle: zero, never read again), its last
// row into the front-padding row the next output has just dropped (zeros right of its diagonal) or into the slack behind
// the last output.  alpha and yT slide by one element.
// ------------------------------------------------------------------------------------------------
struct sr_append1g_args {
    sr_append1_args a;
    double* vp; double* u12; double* gpart;      // n_out x W x Np0, n_out x Np0, n_out x ncb
    unsigned extra;                              // tests: arrivals that never come (the barrier is given up)
    unsigned* cnt; unsigned base; unsigned q0;   // cnt[0]: arrivals, cnt[1]: barrier state (both zero at allocation); arrivals and
                                                 // barriers of all launches so far
    int ncb, npairs;
    int inplace;                                 // Wt1 = Wt0 + Np0 + 1, alpha1 = alpha0 + 1, yT1 = yT0 + 1: see the kernel
};

// Device-wide barrier of the grid.  cnt[0] counts arrivals (it only grows: `target` = arrivals of all barriers so far +
// this grid), cnt[1] is the STATE of the barriers: 2 q after barrier number q has opened, 2 q - 1 after it was given up.  The
// last arriver opens (compare-and-swap 2 (q - 1) -> 2 q); a workgroup that has polled SR_APPG_SPINS times (~5 ms: some of the
// grid never became resident -- another process holding the CUs with a grid of its own, say) gives up with the same
// compare-and-swap towards 2 q - 1: whichever lands first decides for the whole grid.  Returns false when the barrier was
// given up: the caller reports SR_APPG_ABORTED for every output and leaves the kernel, before anything of the model is
// written; the host resets both words and takes the route of separate launches.
#define SR_APPG_SPINS 8192
__device__ __forceinline__ bool sr_appg_barrier(unsigned* cnt, unsigned target, unsigned q, int* s_flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned* state = cnt + 1;
        const unsigned before = 2u * (q - 1u), open = 2u * q, given_up = 2u * q - 1u;
        if (__hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == target - 1u) {
            unsigned e = before;
            __hip_atomic_compare_exchange_strong(state, &e, open, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        unsigned st = before;
        for (int it = 0;; ++it) {
            st = __hip_atomic_load(state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (st != before) break;
            if (it >= SR_APPG_SPINS) {
                unsigned e = before;
                __hip_atomic_compare_exchange_strong(state, &e, given_up, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                st = __hip_atomic_load(state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            __builtin_amdgcn_s_sleep(16);
        }
        *s_flag = (st == open);
    }
    __syncthreads();
    return *s_flag != 0;
}

__global__ __launch_bounds__(1024) void sr_append1_grid_kernel(sr_append1g_args g) {
    const sr_append1_args& a = g.a;
    __shared__ double b[SR_APPEND1G_MAX_NP0];                // b, then X
    __shared__ double part[4][256];
    __shared__ double red[16];
    __shared__ double s_mu, s_inv, s_v2;
    __shared__ int s_ok, s_bar;
    __shared__ double zn[SR_MAX_D];
    const int d = blockIdx.x, w = blockIdx.y, W = gridDim.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N0 = a.N0, Np0 = a.Np0, Np1 = a.Np1, D = a.D;
    if (tid < D) zn[tid] = a.inl ? a.xin[tid] : a.znew[tid];
    const double y_new = a.inl ? a.yin[d] : a.ynew[d];
    __syncthreads();
    const int off0 = Np0 - N0, off1 = Np1 - (N0 + 1), shift = off1 - off0;
    const double* Wt0 = a.Wt0 + (long)d * Np0 * Np0;
    const double* alpha0 = a.alpha0 + (long)d * Np0;
    double* Wt1 = a.Wt1 + (long)d * Np1 * Np1;
    if (d == 0 && w == 0 && a.Zdst && tid < D) a.Zdst[tid] = zn[tid];
    // ---- b = K(Z_old, z_new) in padded row indexing, mu_old = b . alpha0 (as sr_append1_small_kernel)
    double mu_t = 0.0;
#pragma unroll 2
    for (int row = tid; row < Np0; row += 1024) {
        double v = 0.0;
        if (row >= off0) {
            const double* z = a.Z + (long)(row - off0) * D;
            if (a.kp) {
                const double* kp = a.kp + (long)d * SR_KP(D);
                const double *sv = kp + 3, *av = kp + 3 + D, *bv = kp + 3 + 2 * D;
                double r2 = 0.0, la = 0.0, lb = 0.0;
                for (int c = 0; c < D; ++c) {
                    const double t = (z[c] - zn[c]) * sv[c];
                    r2 = fma(t, t, r2);
                    la = fma(av[c] * z[c], zn[c], la);
                    lb = fma(bv[c] * z[c], zn[c], lb);
                }
                v = (kp[2] + la) * kp[1] * sr_kappa((int)kp[0], r2) + lb;
            } else {
                double r2 = 0.0;
                for (int c = 0; c < D; ++c) {
                    const double t = (z[c] - zn[c]) / a.ls[d * D + c];
                    r2 = fma(t, t, r2);
                }
                v = a.sf2[d] * exp(-0.5 * r2);
            }
            mu_t += v * alpha0[row];
        }
        b[row] = v;
    }
    auto wave_sum = [](double v) {
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        return v;
    };
    auto block_sum = [&](double v) {                         // fixed order: wavefront sums, then wavefront 0 .. 15
        const double ws = wave_sum(v);
        __syncthreads();
        if (lane == 0) red[wave] = ws;
        __syncthreads();
        double t = 0.0;
        for (int k = 0; k < 16; ++k) t += red[k];
        return t;
    };
    const double mu_old = block_sum(mu_t);
    if (tid == 0) s_mu = mu_old;
    // ---- phase 1: workgroup w takes the old rows k = off0 + w, + W, .. (interleaved: the rows of a triangle differ in
    // length; runs of 2 .. 8 neighbouring rows per workgroup instead measured the same) and adds b[k] times row k into ITS
    // partial of u12 = U^-T b -- thread = column (+ 1024 i), the rows of a column
    // eight at a time in flight, every load part of an 8 KB run of its row
    {
        const int ncol_t = (Np0 + 1023) / 1024;
        for (int i = 0; i < ncol_t; ++i) {
            // (odd column groups mirrored: a wavefront with the short columns of one group has the long ones of the next --
            //  the wavefronts of a workgroup wait for each other at the barrier)
            const int c = 1024 * i + ((i & 1) ? 1023 - tid : tid);
            if (c >= Np0) continue;                          // (Np0 is a multiple of 128: whole wavefronts skip)
            const int cw = min(__builtin_amdgcn_readfirstlane(c) | 63, Np0 - 1);   // last column of this wavefront: rows beyond it hold
            const double* col = Wt0 + c;                                // zeros for all 64 (rows between c and cw: zeros read)
            double acc = 0.0;
            // sixteen rows in flight, every load unconditional (a row beyond the last one wanted is read once more with a zero
            // multiplier: behind a load that may not have been issued the compiler waits for each one)
            for (int k = off0 + w; k <= cw; k += 16 * W) {
                double v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = col[(long)min(k + u * W, cw) * Np0];
#pragma unroll
                for (int u = 0; u < 16; ++u) acc = fma(v[u], (k + u * W <= cw) ? b[k + u * W] : 0.0, acc);
            }
            sr_st_agent(g.vp + ((long)d * W + w) * Np0 + c, acc);
        }
    }
    unsigned* cnt = g.cnt;                               // ONE counter for the whole grid: every output waits for every other
    const unsigned nwg = (unsigned)W * gridDim.x + g.extra;   // (in place, no output may write unless the pivots of ALL are positive)
    if (!sr_appg_barrier(cnt, g.base + nwg, g.q0 + 1u, &s_bar)) {
        if (tid < (int)gridDim.x) a.info[tid] = SR_APPG_ABORTED;
        return;
    }
    // ---- phase 2: u12 = sum over the workgroups' partials in the order of w, and the squares, 256 columns at a time
    for (int cb = w; cb < g.ncb; cb += W) {
        const int cl = tid & 255, q = tid >> 8, c = cb * 256 + cl;
        double v = 0.0;
        if (c < Np0) {
            const double* src = g.vp + (long)d * W * Np0 + c;
            for (int w0 = q; w0 < W; w0 += 32) {             // eight partials in flight (same order of additions)
                double x[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) x[u] = (w0 + 4 * u < W) ? sr_ld<true>(src + (long)(w0 + 4 * u) * Np0) : 0.0;
#pragma unroll
                for (int u = 0; u < 8; ++u) v += x[u];
            }
        }
        part[q][cl] = v;
        __syncthreads();
        v = 0.0;
        if (tid < 256) {
            v = (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
            if (cb * 256 + tid < Np0) sr_st_agent(g.u12 + (long)d * Np0 + cb * 256 + tid, v);
            else v = 0.0;
        }
        const double gs = block_sum(v * v);
        if (tid == 0) sr_st_agent(g.gpart + (long)d * g.ncb + cb, gs);
    }
    if (!sr_appg_barrier(cnt, g.base + 2u * nwg, g.q0 + 2u, &s_bar)) {
        if (tid < (int)gridDim.x) a.info[tid] = SR_APPG_ABORTED;
        return;
    }
    // ---- the new point's pivot (every workgroup; workgroup 0 of an output reports); in place, the pivots of ALL outputs
    if (tid == 0) {
        int ok_all = 1;
        for (int dd = 0; dd < (int)gridDim.x; ++dd) {
            if (!g.inplace && dd != d) continue;
            double gsum = 0.0;
            for (int cb = 0; cb < g.ncb; ++cb) gsum += sr_ld<true>(g.gpart + (long)dd * g.ncb + cb);
            double prior;                                    // k(z_new, z_new)
            if (a.kp) {
                const double* kp = a.kp + (long)dd * SR_KP(D);
                double la = 0.0, lb = 0.0;
                for (int c = 0; c < D; ++c) {
                    la = fma(kp[3 + D + c] * zn[c], zn[c], la);
                    lb = fma(kp[3 + 2 * D + c] * zn[c], zn[c], lb);
                }
                prior = (kp[2] + la) * kp[1] + lb;           // kappa(0) = 1
            } else {
                prior = a.sf2[dd];
            }
            double sch = prior + a.noise[dd] - gsum;         // Schur complement of the new point
            const bool pos = sch > 0.0;                      // false for NaN too
            if (!pos) { ok_all = 0; sch = 1.0; }
            if (dd == d) {
                if (w == 0) a.info[d] = pos ? 0 : N0 + 1;
                double sd, inv;
                sr_sqrt_rsqrt(sch, sd, inv);
                s_inv = inv;
                s_v2 = inv * (y_new - s_mu);
            }
        }
        s_ok = ok_all;
    }
    __syncthreads();
    const double inv = s_inv, v2 = s_v2;
    const bool wr = !g.inplace || s_ok != 0;               // in place: a failed pivot (of any output) leaves the model as it is
    for (int i = tid; i < Np0; i += 1024) b[i] = sr_ld<true>(g.u12 + (long)d * Np0 + i) * inv;     // X
    __syncthreads();
    // ---- phase 3: the new factor, alpha and targets, row by row (upper part: the rest of the target is in place)
    double ld = 0.0;
    const double* yT0 = a.yT0 + (long)d * Np0;
    double* alpha1 = a.alpha1 + (long)d * Np1;
    double* yT1 = a.yT1 + (long)d * Np1;
    const int Rlast = Np1 - 1;                               // row of the new point
    for (int R = w * 16 + wave; R < Np1; R += 16 * W) {
        double* dst = Wt1 + (long)R * Np1;
        if (R < off1 || R == Rlast) {
            if (lane == 0) {
                if (R == Rlast) { if (wr) dst[Rlast] = inv; ld += log(inv); }
                if (wr && (!g.inplace || R == Rlast)) {      // (in place the padding entries of alpha and yT are the old ones)
                    alpha1[R] = (R == Rlast) ? inv * v2 : 0.0;
                    yT1[R] = (R == Rlast) ? y_new : 0.0;
                }
            }
            continue;
        }
        const int r0 = R - shift;                            // old padded row
        const double* src = Wt0 + (long)r0 * Np0 - shift;    // src[C] = old entry of new column C
        double acc = 0.0;
        int C0 = R;
        if (g.inplace) {                                     // dst[C] IS src[C]: read only
            for (; C0 + 512 <= Rlast; C0 += 512) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = src[C0 + 64 * u + lane];
#pragma unroll
                for (int u = 0; u < 8; ++u) acc = fma(v[u], b[C0 + 64 * u + lane - shift], acc);
            }
            for (; C0 < Rlast; C0 += 64) {
                const int C = C0 + lane;
                if (C < Rlast) acc = fma(src[C], b[C - shift], acc);
            }
        } else {
            for (; C0 + 512 <= Rlast; C0 += 512) {           // eight loads in flight per lane, no predicates
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = src[C0 + 64 * u + lane];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int C = C0 + 64 * u + lane;
                    dst[C] = v[u];
                    acc = fma(v[u], b[C - shift], acc);
                }
            }
            for (; C0 < Rlast; C0 += 64) {
                const int C = C0 + lane;
                if (C < Rlast) {
                    const double v = src[C];
                    dst[C] = v;
                    acc = fma(v, b[C - shift], acc);
                }
            }
        }
        acc = wave_sum(acc);
        if (lane == 0) {
            const double dg = src[R];                        // the old diagonal entry
            if (wr) {
                dst[Rlast] = -acc;
                alpha1[R] = fma(-acc, v2, alpha0[r0]);
                if (!g.inplace) yT1[R] = yT0[r0];
            }
            ld += log(dg);
        }
    }
    __syncthreads();
    if (lane == 0) red[wave] = ld;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int k = 0; k < 16; ++k) t += red[k];
        a.logdet[d * W + w] = -2.0 * t;                      // partial sums: the host adds the W of an output
    }
}

static std::atomic<int> g_test_grid_abort{0};
// tests: the next n launches of the grid kernel wait for a workgroup that does not exist and give up
extern "C" int sr_test_grid_append_abort(int n) { g_test_grid_abort.store(n); return SR_OK; }

long sr_append1_grid_ws(int Np0, int n_out) {                 // doubles of scratch: vp (W partials of u12), u12, gpart
    const int ncb = (Np0 + 255) / 256;
    return (long)n_out * ((long)SR_APPEND1G_MAX_W * Np0 + Np0 + ncb);
}

int sr_launch_append1_grid(const double* Wt0, const double* alpha0, const double* yT0, const double* Z, const double* ls,
                           const double* sf2, const double* noise, const double* kp, const double* znew, const double* ynew, double* Wt1,
                           double* alpha1, double* yT1, double* Zdst, double* logdet, int* info, int N0, int Np0, int Np1,
                           int D, int n_out, int W, double* ws, unsigned* cnt, unsigned base, unsigned q0, hipStream_t s,
                           const double* x_host, const double* y_host, int inplace) {
    SR_CHECK(Np0 <= SR_APPEND1G_MAX_NP0 && Np0 % 128 == 0 && N0 >= 1 && N0 <= Np0 && W >= 1, SR_EINVAL, "append1_grid: Np0 = %d, W = %d", Np0, W);
    sr_append1g_args g;
    g.a = sr_append1_args{Wt0, alpha0, yT0, Z, ls, sf2, noise, kp, znew, ynew, Wt1, alpha1, yT1, Zdst, logdet, info, N0, Np0, Np1, D, n_out,
                          0, {}, {}};
    if (x_host) {                                            // the new point travels in the kernel arguments
        SR_CHECK(y_host && D <= SR_MAX_D && n_out <= SR_APPEND1_MAX_OUT, SR_EINVAL, "append1_grid: D = %d, n_out = %d", D, n_out);
        g.a.inl = 1; g.a.znew = nullptr; g.a.ynew = nullptr;
        for (int c = 0; c < D; ++c) g.a.xin[c] = x_host[c];
        for (int d = 0; d < n_out; ++d) g.a.yin[d] = y_host[d];
    }
    SR_CHECK(W <= SR_APPEND1G_MAX_W, SR_EINVAL, "append1_grid: W = %d", W);
    g.ncb = (Np0 + 255) / 256;
    g.npairs = 0;
    g.vp = ws; g.u12 = ws + (long)n_out * SR_APPEND1G_MAX_W * Np0; g.gpart = g.u12 + (long)n_out * Np0;
    g.cnt = cnt; g.base = base; g.q0 = q0;
    g.extra = 0;
    if (g_test_grid_abort.load() > 0 && g_test_grid_abort.fetch_sub(1) > 0) g.extra = 1;
    g.inplace = inplace;
    SR_CHECK(!inplace || (Np1 == Np0 && Wt1 == Wt0 + Np0 + 1 && alpha1 == alpha0 + 1 && yT1 == yT0 + 1), SR_EINVAL,
             "append1_grid: in place needs the slid views of the same buffers");
    hipLaunchKernelGGL(sr_append1_grid_kernel, dim3(n_out, W), dim3(1024), 0, s, g);
    SR_HIP(hipGetLastError());
    return SR_OK;
}

// alpha of the grown model without another pass over U^-1:  with r = y_new - mu_old(z_new) (the old model's
// mean at the new points, which the K* pass has just produced) and v2 = U22^-T r,
//   alpha1 = [alpha0 + Y2 v2 ; U22^-1 v2].     One workgroup recomputes v2 (m <= 16), grid over the rows.
__global__ __launch_bounds__(256) void sr_append_alpha_kernel(const double* __restrict__ alpha0, int Np0, int N0,
                                                              const double* __restrict__ Y2,
                                                              const double* __restrict__ invS,
                                                              const double* __restrict__ mu_part, int nsplit,
                                                              int n_out, int d, long Tp,
                                                              const double* __restrict__ Ynew, int m,
                                                              double* __restrict__ alpha1, int Np1, int qoff, long sY2) {
    // qoff: position of the first new point among the queries of the K* pass (front-padded query block: 128 - m)
    __shared__ double r[SR_NB], v2[SR_NB], red[4][16];
    const int pf = SR_NB - m, off0 = Np0 - N0, off1 = Np1 - (N0 + m);
    {                                                       // batch: output d + blockIdx.y, the pointers given belong to d
        const long b = blockIdx.y;
        d += (int)b; alpha0 += b * Np0; Y2 += b * sY2; invS += b * SR_NB * SR_NB; alpha1 += b * Np1;
    }
    // mean of the old model at the new points: the N-split partials of the K* pass, summed by the whole workgroup
    // (one thread per point walking up to Np / 16 partials was 119 us of dependent-latency at N = 5000)
    for (int q0 = 0; q0 < m; q0 += 16) {
        double acc[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0.0;
        for (int sp = threadIdx.x; sp < nsplit; sp += 256) {
            const double* src = mu_part + ((long)sp * n_out + d) * Tp + qoff + q0;
#pragma unroll
            for (int q = 0; q < 16; ++q)
                if (q0 + q < m) acc[q] += src[q];
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            double v = acc[q];
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][q] = v;
        }
        __syncthreads();
        if (threadIdx.x < 16 && q0 + (int)threadIdx.x < m) {
            const int q = q0 + threadIdx.x;
            const double mu = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
            r[q] = Ynew[(long)q * n_out + d] - mu;
        }
        __syncthreads();
    }
    if ((int)threadIdx.x < m) {
        double v = 0.0;
        for (int b = 0; b <= (int)threadIdx.x; ++b) v = fma(invS[(pf + b) * SR_NB + pf + threadIdx.x], r[b], v);
        v2[threadIdx.x] = v;
    }
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;           // index in the new padded vector
    if (i >= Np1) return;
    double a = 0.0;
    if (i >= off1) {
        const int k = i - off1;
        if (k < N0) {
            a = alpha0[off0 + k];
            for (int c = 0; c < m; ++c) a = fma(Y2[(long)(off0 + k) * SR_NB + pf + c], v2[c], a);
        } else {
            const int q = k - N0;
            for (int c = q; c < m; ++c) a = fma(invS[(pf + q) * SR_NB + pf + c], v2[c], a);
        }
    }
    alpha1[i] = a;
}

int sr_launch_append_alpha(const double* alpha0, int Np0, int N0, const double* Y2, const double* invS,
                           const double* mu_part, int nsplit, int n_out, int d, long Tp, const double* Ynew, int m,
                           double* alpha1, int Np1, hipStream_t s, int qoff, int nbatch, long sY2) {
    hipLaunchKernelGGL(sr_append_alpha_kernel, dim3((Np1 + 255) / 256, nbatch), dim3(256), 0, s, alpha0, Np0, N0, Y2, invS,
                       mu_part, nsplit, n_out, d, Tp, Ynew, m, alpha1, Np1, qoff, sY2);
    SR_HIP(hipGetLastError());
    return SR_OK;
}

int sr_launch_append_small(const double* U12t, const double* Wt0, int Np0, int m, int stage, double* G,
                           const double* invS, double* Xt, double* Y2, hipStream_t s, int nbatch) {
    if (stage == 0) {
        hipLaunchKernelGGL(sr_append_gsmall_kernel, dim3(m, m, nbatch), dim3(256), 0, s, U12t, Np0, m, G);
    } else {
        hipLaunchKernelGGL(sr_append_xt_kernel, dim3((Np0 + 255) / 256, m), dim3(256), 0, s, U12t, invS, Np0, m, Xt, 0L);
        SR_HIP(hipGetLastError());
        if (m <= 1)
            hipLaunchKernelGGL(sr_append_y2_kernel<1>, dim3((Np0 + 3) / 4), dim3(256), 0, s, Wt0, Np0, Xt, m, Y2);
        else if (m <= 4)
            hipLaunchKernelGGL(sr_append_y2_kernel<4>, dim3((Np0 + 3) / 4), dim3(256), 0, s, Wt0, Np0, Xt, m, Y2);
        else
            hipLaunchKernelGGL(sr_append_y2_kernel<16>, dim3((Np0 + 3) / 4), dim3(256), 0, s, Wt0, Np0, Xt, m, Y2);
    }
    SR_HIP(hipGetLastError());
    return SR_OK;
}

