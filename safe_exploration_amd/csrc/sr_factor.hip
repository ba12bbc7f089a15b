// sr_factor.hip -- model-update path: Gram matrix, the diagonal-block kernel of the blocked fp64 Cholesky
// K = U^T U (factor and inverse of a 128 x 128 block in one workgroup), triangular matrix-vector products, log det.
// The products of the update run through sr_gemm.hip; launch plan: sr_capi_update.hip.
//
// replaces (numerically) what GPy computes for SimpleGPModel.train / update_model:
//   /root/reference/safe_exploration/ssm_gpy/gaussian_process.py:238-275, 398-419
#include "sr_mfma_tile.h"
#include "sr_flow.h"
#include "sr_pivot_dev.h"
// ------------------------------------------------------------------------------------------------
// Gram matrix K[i][j] = sf2 exp(-0.5 |(z_i - z_j)/l|^2) + noise (i==j); identity on the padding.
// (kernel spec: ssm_gpy/gp_models_utils_casadi.py:17-40; noise on the diagonal:
//  ssm_gpy/gaussian_process.py:252-253)
// ------------------------------------------------------------------------------------------------
// sf2_dev / noise_dev (device scalars) take precedence over the by-value arguments when not NULL: the model update
// then needs no host copy of the hyper-parameters before its first launch.  Thread = 4 rows x 1 column: the column's
// scaled coordinates are loaded once per 4 entries, and the scaling is a multiplication by 1 / l (computed once per
// thread; the division per entry and dimension was a third of the kernel's time).
#define SR_GRAM_ROWS 32
__global__ __launch_bounds__(256) void sr_gram_kernel(const double* __restrict__ Z,
                                                      const double* __restrict__ ls, double sf2,
                                                      double noise, const double* __restrict__ sf2_dev,
                                                      const double* __restrict__ noise_dev,
                                                      double* __restrict__ K, int N, int Np, int D, long strideK) {
    __shared__ double zs[SR_GRAM_ROWS][SR_MAX_D];
    const int i0 = blockIdx.y * SR_GRAM_ROWS;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if ((int)((blockIdx.x * 256 + 255) | 127) < i0) return;   // blocks left of the diagonal are never read (upper factorisation)
    const int b = blockIdx.z;                // batch member (output)
    ls += (long)b * D;
    K += (long)b * strideK;
    if (sf2_dev) sf2 = sf2_dev[b];
    if (noise_dev) noise = noise_dev[b];
    const int off = Np - N;                  // front padding
    double il[SR_MAX_D], zj[SR_MAX_D];
    for (int c = 0; c < D; ++c) {
        il[c] = 1.0 / ls[c];
        zj[c] = (j < Np && j >= off) ? Z[(long)(j - off) * D + c] * il[c] : 0.0;
    }
    // the rows of this tile, scaled, in LDS (broadcast reads)
    for (int e = threadIdx.x; e < SR_GRAM_ROWS * D; e += 256) {
        const int r = e / D, c = e % D, i = i0 + r;
        zs[r][c] = (i < Np && i >= off) ? Z[(long)(i - off) * D + c] * (1.0 / ls[c]) : 0.0;
    }
    __syncthreads();
    if (j >= Np || (j | 127) < i0) return;
    for (int u = 0; u < SR_GRAM_ROWS; ++u) {
        const int i = i0 + u;
        if (i >= Np) break;
        double v;
        if (i < off || j < off) {
            v = (i == j) ? 1.0 : 0.0;
        } else if (i == j) {
            v = sf2 + noise;
        } else {
            double r2 = 0.0;
            for (int c = 0; c < D; ++c) {
                const double t = zs[u][c] - zj[c];
                r2 = fma(t, t, r2);
            }
            v = sf2 * exp(-0.5 * r2);
        }
        K[(long)i * Np + j] = v;
    }
}

int sr_launch_gram(const double* Z, const double* ls, double sf2, double noise, const double* sf2_dev,
                   const double* noise_dev, double* K, int N, int Np, int D, hipStream_t s, int nbatch, long strideK) {
    dim3 grid((Np + 255) / 256, (Np + SR_GRAM_ROWS - 1) / SR_GRAM_ROWS, nbatch);
    hipLaunchKernelGGL(sr_gram_kernel, grid, dim3(256), 0, s, Z, ls, sf2, noise, sf2_dev, noise_dev, K, N, Np, D, strideK);
    SR_HIP(hipGetLastError());
    return SR_OK;
}

// general kernel family (see sr_common.h; kappa: sr_pivot_dev.h); diagonal = k(z_i, z_i) + noise

__global__ __launch_bounds__(256) void sr_gram_general_kernel(const double* __restrict__ Z,
                                                              const double* __restrict__ kp, double noise,
                                                              const double* __restrict__ noise_dev,
                                                              double* __restrict__ K, int N, int Np, int D,
                                                              long strideK) {
    const int i = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= Np) return;
    if ((j | 127) < i) return;               // blocks left of the diagonal are never read (upper factorisation)
    const int b = blockIdx.z;                // batch member (output)
    kp += (long)b * SR_KP(D);
    K += (long)b * strideK;
    if (noise_dev) noise = noise_dev[b];
    const int off = Np - N;
    double v;
    if (i < off || j < off) {
        v = (i == j) ? 1.0 : 0.0;
    } else {
        const double* zi = Z + (long)(i - off) * D;
        const double* zj = Z + (long)(j - off) * D;
        const int kind = (int)kp[0];
        const double var = kp[1], c0 = kp[2];
        const double *sv = kp + 3, *av = kp + 3 + D, *bv = kp + 3 + 2 * D;
        double r2 = 0.0, la = 0.0, lb = 0.0;
        for (int c = 0; c < D; ++c) {
            const double t = (zi[c] - zj[c]) * sv[c];
            r2 = fma(t, t, r2);
            la = fma(av[c] * zi[c], zj[c], la);
            lb = fma(bv[c] * zi[c], zj[c], lb);
        }
        v = (c0 + la) * var * sr_kappa(kind, (i == j) ? 0.0 : r2) + lb;
        if (i == j) v += noise;
    }
    K[(long)i * Np + j] = v;
}

int sr_launch_gram_general(const double* Z, const double* kp, double noise, const double* noise_dev, double* K, int N,
                           int Np, int D, hipStream_t s, int nbatch, long strideK) {
    dim3 grid((Np + 255) / 256, Np, nbatch);
    hipLaunchKernelGGL(sr_gram_general_kernel, grid, dim3(256), 0, s, Z, kp, noise, noise_dev, K, N, Np, D, strideK);
    SR_HIP(hipGetLastError());
    return SR_OK;
}

// ------------------------------------------------------------------------------------------------
// Diagonal block: A_kk = U_kk^T U_kk (upper Cholesky) and the inverse of U_kk, all in LDS, by one workgroup of 16
// wavefronts (sr_potrf_diag_kernel below; the helpers here are shared with the one-wavefront corner kernel of the row
// append).  It sits on the critical path of the blocked factorisation (one launch per 128 rows): its serial core is the
// chain of 128 pivots (sqrt -> reciprocal -> rank-1 update -> next pivot), taken 16 at a time by ONE wavefront entirely
// in registers -- lane c holds column c, pivots and multipliers travel through v_readlane -- no LDS round trip, no fence.
// (Round 2's form -- panel rows by forward substitution, sub-block inverses on a spare wavefront, U_kk^-1 assembled by
// [A B; 0 C]^-1 steps at block sizes 16, 32, 64: 62 us -- and the first one, 88 us, are in the history of the repository.)
// ------------------------------------------------------------------------------------------------
#define SR_PD_LD 144     // 128 + 16: the four row groups (lane >> 4) of an MFMA operand read land 16 doubles apart modulo 32,
                         // i.e. on disjoint LDS banks (129 = 128 + 1 made every such read a two-way conflict)
#define SR_PD_THREADS 1024
#define SR_PD_XLD 17

// upper Cholesky of the 16 x 16 sub-block at (j0, j0) of S by one wavefront, in registers.
// invd[j0 + j] receives 1 / U[j][j]; *fail the 1-based index of the first non-positive pivot.
__device__ __forceinline__ void sr_factor16(double* S, int j0, double* invd, int* fail, int lane) {
    const int c = lane & 15;                       // lanes 16 .. 63 mirror lanes 0 .. 15 (same addresses: broadcast)
    double a[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = S[(j0 + r) * SR_PD_LD + j0 + c];
    double myinv = 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        double d = sr_readlane_f64(a[j], j);       // pivot: wavefront-uniform
        if (!(d > 0.0)) {                          // also catches NaN
            if (lane == 0 && *fail == 0) *fail = j0 + j + 1;
            d = 1.0;
        }
        double sd, inv;
        sr_sqrt_rsqrt(d, sd, inv);
        if (c == j) myinv = inv;
        a[j] = (c == j) ? sd : a[j] * inv;         // row j of U (meaningful on lanes c >= j)
#pragma unroll
        for (int r = j + 1; r < 16; ++r) {
            const double ujr = sr_readlane_f64(a[j], r);     // U[j][r]: wavefront-uniform
            a[r] = fma(-ujr, a[j], a[r]);
        }
    }
    if (lane < 16) {
        invd[j0 + c] = myinv;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (r <= c) S[(j0 + r) * SR_PD_LD + j0 + c] = a[r];
    }
}

// X = U^-1 of the factored 16 x 16 sub-block at (j0, j0) by one wavefront: lane i holds row i of X,
//   X[i][j] = -(sum_{k < j} X[i][k] U[k][j]) / U[j][j],  X[j][j] = 1 / U[j][j];   U broadcast from LDS.
__device__ __forceinline__ void sr_invert16(const double* S, int j0, const double* invd, double* X, int lane) {
    const int i = lane & 15;
    double x[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < j; ++k) s = fma(x[k], S[(j0 + k) * SR_PD_LD + j0 + j], s);
        const double ij = invd[j0 + j];
        x[j] = (i == j) ? ij : ((i < j) ? -s * ij : 0.0);
    }
    if (lane < 16) {
#pragma unroll
        for (int j = 0; j < 16; ++j) X[i * SR_PD_XLD + j] = x[j];
    }
}

// The Schur complement of a row append of m <= 16 points lives in the LAST 16 x 16 pivot of an otherwise identity
// 128 x 128 block: one wavefront factors and inverts that corner (3 us; the full kernel takes 62 us on the way of every
// appended point).  Writes rows / columns 112 .. 127 of wt_diag (= U22^-1, upper) only.
// G != NULL: the block factored is A - G on rows / columns >= pf (the Schur complement C - U12^T U12 of the append, G a
// 128 x 128 block with leading dimension 128); blockIdx.x: batch member (strides sA of A, sW of wt_diag, 128 x 128 of G).
__global__ __launch_bounds__(64) void sr_potrf_corner16_kernel(double* A, long lda, double* wt_diag, long ldw,
                                                               int* info, const double* __restrict__ G, int pf,
                                                               long sA, long sW) {
    __shared__ double S[16 * SR_PD_LD];
    __shared__ double X[16 * SR_PD_XLD];
    __shared__ double invd[16];
    __shared__ int fail;
    A += (long)blockIdx.x * sA; wt_diag += (long)blockIdx.x * sW; info += blockIdx.x;
    if (G) G += (long)blockIdx.x * SR_NB * SR_NB;
    const int lane = threadIdx.x, c0 = SR_NB - 16;
    if (lane == 0) fail = 0;
    for (int idx = lane; idx < 256; idx += 64) {
        const int r = idx >> 4, c = idx & 15;
        double v = (c >= r) ? A[(long)(c0 + r) * lda + c0 + c] : 0.0;
        if (G && c >= r && c0 + r >= pf) v -= G[(c0 + r) * SR_NB + c0 + c];
        S[r * SR_PD_LD + c] = v;
    }
    __syncthreads();
    sr_factor16(S, 0, invd, &fail, lane);
    __syncthreads();
    sr_invert16(S, 0, invd, X, lane);
    __syncthreads();
    const bool bad = fail != 0;
    if (bad && lane == 0 && *info == 0) *info = c0 + fail;
    for (int idx = lane; idx < 256; idx += 64) {
        const int r = idx >> 4, c = idx & 15;
        const double eye = (r == c) ? 1.0 : 0.0;
        A[(long)(c0 + r) * lda + c0 + c] = bad ? eye : ((c >= r) ? S[r * SR_PD_LD + c] : 0.0);
        wt_diag[(long)(c0 + r) * ldw + c0 + c] = bad ? eye : X[r * SR_PD_XLD + c];
    }
}

int sr_launch_potrf_corner16(double* A, long lda, double* wt_diag, long ldw, int* info_dev, hipStream_t s,
                             const double* G, int pf, int nbatch, long sA, long sW) {
    hipLaunchKernelGGL(sr_potrf_corner16_kernel, dim3(nbatch), dim3(64), 0, s, A, lda, wt_diag, ldw, info_dev, G, pf, sA, sW);
    SR_HIP(hipGetLastError());
    return SR_OK;
}

// ------------------------------------------------------------------------------------------------
// Diagonal block: factor AND inverse in ONE sweep over the 8 panels of 16 columns.
// The elimination that turns A_kk into U_kk (U^-T A = U) is applied to an augmented identity at the same time, so that
// V = U_kk^-T is complete the moment the factor is:
//   * sr_factor16_aug: the 16 x 16 pivot tile in the registers of ONE wavefront (wavefront 0), 16 of its lanes carrying
//     the columns of the identity through the same row operations: T_p = U_pp^-T at no extra instruction;
//   * the panel row  U[p][c] = T_p A[p][c] (c > p)  and  V[p][c'] = T_p V[p][c'] (c' < p): seven 16 x 16 tiles, 4 MFMAs
//     each, one wavefront per tile, in place in LDS and straight to global memory;
//   * trailing updates  A[r][c] -= U[p][r]^T U[p][c]  (p < r <= c)  and  V[r][c'] -= U[p][r]^T V[p][c']  (c' <= p < r)
//     as independent tile jobs of 12 worker wavefronts;
//   * wavefront 0 -- the critical path: 128 dependent pivots -- computes the ONE tile it needs (U[p][p+1]) itself,
//     updates the next pivot tile with it (the MFMA result layout IS the operand layout of the next product: row
//     (lane >> 4) + 4 reg, column lane & 15) and factors it at once; it never waits for the rest of the panel row.
//   V lives in the strict lower block triangle of S (A needs the upper one only): no second LDS matrix.
// fp64 MFMA and fp64 VALU share the DP pipe of a SIMD: the wavefronts that sit on wavefront 0's SIMD (4, 8, 12) take no
// MFMA job -- they copy the pivot stage to global memory and write the structural zeros of the outputs.
// Synchronisation: ONE LDS-only workgroup barrier per panel (s_waitcnt lgkmcnt(0); s_barrier -- __syncthreads() would
// also wait for the global stores in flight, which nobody reads back) and a tile count in LDS that tells the workers when
// the panel row is complete.
// History (profiles/archive/r05_diag_kernel.txt has the cycle counts): 88 us (round 1) -> 62 (sub-block inverses on a spare
// wavefront) -> 43 (one sweep) -> 30 (round 3: pivots by v_readlane, two barriers per panel) -> 23 (this form: pivot
// chain written for the smallest fp64 instruction count, one barrier, loads of the upper triangle only and in flight
// together, wavefront 0 starting on the first tile while the other 15 load).  What is left: wavefront 0's 4.8k cycles per
// panel (3.1k of them the 16 pivots) and, in the first four panels, the 35 .. 26 trailing tiles, whose 16 LDS operations
// per tile keep the LDS busier than the pivots keep wavefront 0 (tiles resident in worker registers halve that traffic but
// measured slower: 24.5 us, profiles/archive/r05_diag_kernel.txt).
// ------------------------------------------------------------------------------------------------
#define SR_PD_TLD 33      // pivot stage: 16 rows of [U_pp (16 columns) | T_p = U_pp^-T (16 columns)], padded

// (A square-root-free variant that takes R_j[r] from lane r with the DPP row broadcast of the fp64 ALU -- one
//  v_fmac_f64_dpp ... row_newbcast:r per updated entry, reciprocal instead of rsqrt on the chain, the 16 square roots
//  at the end -- was built and measured in round 3: 38 us per block against 30 with v_readlane.  DPP on the fp64 ALU is
//  slow on this part.)
// upper Cholesky of a 16 x 16 tile (src, ld) by one wavefront in registers.  X (LDS, 16 x SR_PD_TLD) receives
// [U | T], T = U^-T (lower triangular, exact zeros above; the part of U below its diagonal is scratch).
// src, ld: the tile (LDS), its leading dimension.  *fail: 1-based index of the first non-positive pivot (j0 + ...).
// The copies to global memory are another wavefront's job.
__device__ __forceinline__ void sr_factor16_aug(const double* src, long ld, int j0, double* X, int* fail, int lane) {
    // A single wavefront is bound by instruction ISSUE here, not by latency: an fp64 operation holds the issue port of its
    // wavefront for 8 cycles, anything else for 4, and filling the latency slots of the pivot chain with independent work
    // buys nothing (scripts/pivot_chain.hip: chain alone 186 cycles per pivot, with the row updates 262 = 186 + their
    // issue time).  So the 16 pivots are written for the smallest fp64 instruction count:
    //   * lanes 32 .. 63, which only mirrored lanes 0 .. 31 (16 columns of the tile, 16 of the augmented identity), now own
    //     the ODD rows and lanes 0 .. 31 the EVEN ones: a rank-1 update is one FMA per PAIR of rows (71 instead of 120 per
    //     tile), its multipliers U[j][r] read back from the pivot stage in LDS -- the finished row has to go there anyway
    //     -- with one ds_read per pair of rows, the address differing by half (two v_readlane, a hazard nop per row before);
    //   * the finished row crosses to the other half by v_permlane32_swap (two instructions);
    //   * pivot j = a[j][j] - U[j-1][j]^2 is taken from lane j, which has both terms, before the pending update of row j;
    //   * 1 / sqrt(d) by two coupled Goldschmidt steps from the v_rsq_f64 seed (halving / doubling by integer adds on the
    //     exponent), U[j][j] = d / sqrt(d) falls out of the row scaling: no select, no second residual step.
    // Scheduling barriers keep the order (left to itself the compiler builds a left-looking form whose row j + 1 waits for
    // a chain of j dependent FMAs, and any branch in here makes it sink the updates to their uses: 5450 -- 6100 cycles per
    // tile against 2.7k).  A non-positive pivot takes no branch: it leaves NaN in every row after it.
    const int c = lane & 15;
    const bool aug = (lane & 16) != 0;              // lanes 16..31, 48..63: columns of the identity
    const int hf = lane >> 5;                       // rows 2 k + hf
    double b[8], u[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) b[k] = aug ? ((2 * k + hf == c) ? 1.0 : 0.0) : src[(2 * k + hf) * ld + c];
    const double* Xh = X + hf;
    double sp = 0.0;                                // row j - 1 of [U | U^-T], in both halves
#define SR_PV_SB __builtin_amdgcn_sched_barrier(0)
#define SR_PV_U(k_) if (j > 0 && (k_) >= (j >> 1)) u[k_] = Xh[(j > 0 ? j - 1 : 0) * SR_PD_TLD + 2 * (k_)];
#define SR_PV_F(i_) if (j > 0 && (j >> 1) + (i_) < 8) b[((j >> 1) + (i_)) & 7] = fma(-u[((j >> 1) + (i_)) & 7], sp, b[((j >> 1) + (i_)) & 7]);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int kj = j >> 1, ho = j & 1;
        SR_PV_U(0) SR_PV_U(1) SR_PV_U(2) SR_PV_U(3) SR_PV_U(4) SR_PV_U(5) SR_PV_U(6) SR_PV_U(7)
        SR_PV_SB;
        const double t = (j > 0) ? fma(-sp, sp, b[kj]) : b[kj];
        const double d = sr_readlane_f64(t, j + 32 * ho);      // pivot: wavefront-uniform
        SR_PV_SB;
        const double y = __builtin_amdgcn_rsq(d);
        SR_PV_SB;
        double g = d * y;
        double h = __hiloint2double(__double2hiint(y) - 0x00100000, __double2loint(y));
        SR_PV_SB;
        const double r = fma(-h, g, 0.5);
        SR_PV_SB;
        g = fma(g, r, g);
        h = fma(h, r, h);
        SR_PV_SB;
        const double r2 = fma(-h, g, 0.5);
        const double hh = __hiloint2double(__double2hiint(h) + 0x00100000, __double2loint(h));
        SR_PV_F(0) SR_PV_F(1) SR_PV_F(2)
        SR_PV_SB;
        const double inv = fma(hh, r2, hh);
        SR_PV_F(3) SR_PV_F(4) SR_PV_F(5)
        SR_PV_SB;
        SR_PV_F(6) SR_PV_F(7)
        const double so = b[kj] * inv;               // row j, in the half that owns it
        SR_PV_SB;
        const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(so), (unsigned)__double2loint(so), false, false);
        const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(so), (unsigned)__double2hiint(so), false, false);
        sp = __hiloint2double((int)hi[ho], (int)lo[ho]);
        X[j * SR_PD_TLD + (lane & 31)] = sp;         // both halves: same value, same address
        SR_PV_SB;
    }
#undef SR_PV_F
#undef SR_PV_U
#undef SR_PV_SB
    // a bad pivot leaves NaN in every row after it: the last diagonal entry tells (wavefront-uniform branch)
    const double last = sr_readlane_f64(sp, 15);
    if (!(last > 0.0 && last < __builtin_inf())) {
        const double sd = X[c * SR_PD_TLD + c];
        const unsigned long long okm = __ballot(sd > 0.0 && sd < __builtin_inf());
        if (lane == 0 && *fail == 0) *fail = j0 + __builtin_ctzll(~okm | (1ull << 16)) + 1;
    }
}

// P = T X for the 16 x 16 tile X at rows j0.., columns c0.. of S: result in the MFMA layout (reg q: row (lane>>4) + 4 q,
// column lane & 15), which is also the A- and B-operand layout of a product that contracts over P's rows
__device__ __forceinline__ d4_t sr_pd_panel_tile(const double* S, const double* X, int j0, int c0, int lk, int ln) {
    d4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const double af = X[ln * SR_PD_TLD + 16 + 4 * kk + lk];
        const double bf = S[(j0 + 4 * kk + lk) * SR_PD_LD + c0 + ln];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af, bf, acc, 0, 0, 0);
    }
    return acc;
}

// LDS-only workgroup barrier: __syncthreads() also waits for the global stores in flight (vmcnt(0)), and this kernel
// streams its results to global memory all the way through without ever reading them back
__device__ __forceinline__ void sr_pd_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// count of panel tiles in LDS (monotonic over the kernel): a wavefront adds one when its tile of panel row p is written,
// the wavefronts that need the whole row wait for 7 (p + 1).  LDS executes the operations of a wavefront in order, so
// whoever sees the count sees the tile; the compiler is held by the asm statements.
__device__ __forceinline__ void sr_pd_signal(int* cnt, int lane) {
    asm volatile("" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void sr_pd_wait(int* cnt, int target) {
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
}

// The block at Ag (leading dimension lda; k0 = its first row, for the breakdown report) -> U_kk in place, U_kk^-1 to wt_diag,
// U_kk^-T to w_diag (leading dimension ldw).  S, Xb, fail, done: the workgroup's LDS.  Shared by the launched kernel (one block
// per launch) and the resident workgroup of the tile-flow Cholesky below.
__device__ __forceinline__ void sr_pd_block(double* Ag, long lda, double* wt_diag, double* w_diag, long ldw, long k0,
                                            int* info, bool storeA, double* S, double (*Xb)[16 * SR_PD_TLD], int& fail,
                                            int& done) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lk = lane >> 4, ln = lane & 15;
    if (tid == 0) { fail = 0; done = 0; }
    // wavefront -> role: 0 the pivots; 1 .. 6 a tile of the panel row each, then trailing-update jobs; 4, 8, 12 (the SIMD
    // of wavefront 0: fp64 MFMA and fp64 VALU share its DP pipe, MFMA jobs there stall the pivot chain) stores only
    // (4: one panel tile too); the other 12 are the MFMA workers of the trailing update.
    const bool simd0 = (wave & 3) == 0;
    const int worker = wave - 1 - (wave >> 2);             // 0 .. 11 for the MFMA wavefronts
    constexpr int NWORK = 12;
    if (wave != 0) {
        // the 36 upper tiles come from global memory -- every load in flight before the first LDS write, and issued a
        // little AFTER the eight loads of wavefront 0, which would otherwise queue behind these 150 --, the 28 strictly
        // lower ones (V = 0 off its diagonal) are zero-filled while the loads fly
        constexpr int NT = SR_NB / 16, NUP = NT * (NT + 1) / 2 * 256, NLO = NT * (NT - 1) / 2 * 256;
        constexpr int NTH = SR_PD_THREADS - 64, NLD = (NUP + NTH - 1) / NTH, NZ = (NLO + NTH - 1) / NTH;
        const auto tri = [](int t) { return (t >= 1) + (t >= 3) + (t >= 6) + (t >= 10) + (t >= 15) + (t >= 21) + (t >= 28); };
        __builtin_amdgcn_s_sleep(4);
        double v[NLD];
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int m = tid - 64 + i * NTH, tile = m >> 8, e = m & 255;
            const int tc = tri(tile), tr = tile - tc * (tc + 1) / 2;         // column-major upper triangle
            const int r = 16 * tr + (e >> 4), c = 16 * tc + (e & 15);
            v[i] = (m < NUP && c >= r) ? Ag[(long)r * lda + c] : 0.0;
        }
#pragma unroll
        for (int i = 0; i < NZ; ++i) {
            const int m = tid - 64 + i * NTH, tile = m >> 8, e = m & 255;
            const int tr = tri(tile) + 1, tc = tile - tr * (tr - 1) / 2;     // row-major strict lower triangle
            if (m < NLO) S[(16 * tr + (e >> 4)) * SR_PD_LD + 16 * tc + (e & 15)] = 0.0;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int m = tid - 64 + i * NTH, tile = m >> 8, e = m & 255;
            const int tc = tri(tile), tr = tile - tc * (tc + 1) / 2;
            if (m < NUP) S[(16 * tr + (e >> 4)) * SR_PD_LD + 16 * tc + (e & 15)] = v[i];
        }
    }
    // Panel p = -1 is the prologue: wavefront 0 factors the first pivot tile straight from global memory while the other
    // 15 bring the block into LDS (one copy of the pivot code in the kernel: it starts cold in the instruction cache of
    // whatever CU the launch lands on).
    // ONE workgroup barrier per panel.  Wavefront 0 never waits for the panel row: it computes the one tile it needs
    // (U[p][p+1]) itself, updates the next pivot tile with it and factors that -- 16 dependent pivots, the critical path
    // -- while the others compute the rest of the row, count their tiles in, and run the trailing update of panel p once
    // the count says the row is complete.
    for (int p = -1; p < SR_NB / 16; ++p) {
        const int j0 = 16 * p;
        const double* X = Xb[p & 1];
        const int nbt = SR_NB / 16 - 1 - p;
        {
            const int target = (SR_NB / 16 - 1) * (p + 1);
            if (wave == 0 && nbt > 0 && p < 0) {
                // first pivot tile: global memory -> LDS, then the one copy of the pivot code below (a second copy reading
                // global memory would start cold in the instruction cache once more)
                int z = 0;
                asm volatile("" : "+v"(z));              // keeps these addresses out of the registers of the whole loop
#pragma unroll
                for (int q = 0; q < 4; ++q) S[(lk + 4 * q) * SR_PD_LD + ln] = Ag[(long)(lk + 4 * q + z) * lda + ln];
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                sr_factor16_aug(&S[0], SR_PD_LD, 0, Xb[0], &fail, lane);
            } else if (wave == 0 && nbt > 0) {
                // U[p][p+1] = T_p A[p][p+1]; next pivot tile -= U[p][p+1]^T U[p][p+1]; factor it.  All operands in one
                // LDS round trip; the MFMA result layout IS the operand layout of the next product.
                const int c0 = j0 + 16;
                double af[4], bf[4];
                d4_t cc, mine = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    af[kk] = X[ln * SR_PD_TLD + 16 + 4 * kk + lk];
                    bf[kk] = S[(j0 + 4 * kk + lk) * SR_PD_LD + c0 + ln];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) cc[q] = S[(c0 + lk + 4 * q) * SR_PD_LD + c0 + ln];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) mine = __builtin_amdgcn_mfma_f64_16x16x4f64(af[kk], bf[kk], mine, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) S[(j0 + lk + 4 * q) * SR_PD_LD + c0 + ln] = mine[q];   // row p + 1 of the trailing update
                sr_pd_signal(&done, lane);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) cc = __builtin_amdgcn_mfma_f64_16x16x4f64(-mine[kk], mine[kk], cc, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) S[(c0 + lk + 4 * q) * SR_PD_LD + c0 + ln] = cc[q];
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");     // the tile update above, then its reads below
                sr_factor16_aug(&S[c0 * SR_PD_LD + c0], SR_PD_LD, c0, Xb[(p + 1) & 1], &fail, lane);
            } else if (p >= 0 && wave < SR_NB / 16 - 1) {
                // the other tiles of panel row p, in place: wavefront w takes tile w
                const int ct = (wave < nbt) ? p + 1 + wave : wave - nbt;     // tile column: A part c > p, then V part c' < p
                const int c0 = 16 * ct;
                const d4_t mine = sr_pd_panel_tile(S, X, j0, c0, lk, ln);
#pragma unroll
                for (int q = 0; q < 4; ++q) S[(j0 + lk + 4 * q) * SR_PD_LD + c0 + ln] = mine[q];
                sr_pd_signal(&done, lane);
                if (ct > p) {
                    if (storeA) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) Ag[(long)(j0 + lk + 4 * q) * lda + c0 + ln] = mine[q];
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        w_diag[(long)(j0 + lk + 4 * q) * ldw + c0 + ln] = mine[q];
                        wt_diag[(long)(c0 + ln) * ldw + j0 + lk + 4 * q] = mine[q];
                    }
                }
            }
            if (p >= 0 && wave != 0) {
                sr_pd_wait(&done, target);
                if (!simd0) {
                    // ---- trailing updates  A[r][c] -= U[p][r]^T U[p][c]  (p < r <= c)  and  V[r][c'] -= U[p][r]^T V[p][c']
                    const int nA = nbt * (nbt + 1) / 2;        // trailing tiles of A; tile 0 is wavefront 0's
                    const int nV = nbt * (p + 1);              // trailing tiles of V
                    // Each job: operands AND the tile to be updated in ONE LDS round trip (left to the compiler: read, wait,
                    // MFMA, four times over, then the read-modify-write: 1800 cycles per job instead of 1100).  Job -> tile
                    // without loops or divisions (a quarter of the instructions of a job were its index arithmetic).
                    // (Prefetching the next job's operands under the MFMAs of this one was measured slower.)
                    const int ne = nA + nV;
                    const int rcp = (65536 + p) / (p + 1);                 // v / (p + 1) = (v * rcp) >> 16 for v < 64
                    for (int e = 1 + worker; e < ne; e += NWORK) {
                        int r0, c0;
                        bool useT = false;
                        if (e < nA) {
                            // row ti of the triangle has nbt - ti tiles; f counts from the far end, where row 7-.. has 1, 2, 3 ...
                            const int f = nA - 1 - e;                      // 0 .. nA - 1, last tile first
                            const int g = (f >= 1) + (f >= 3) + (f >= 6) + (f >= 10) + (f >= 15) + (f >= 21);   // rows from the end
                            const int ti = nbt - 1 - g, rem = (nbt - ti) - 1 - (f - g * (g + 1) / 2);
                            r0 = 16 * (p + 1 + ti);
                            c0 = r0 + 16 * rem;
                        } else {
                            const int v = e - nA, ti = (v * rcp) >> 16, cq = v - ti * (p + 1);
                            r0 = 16 * (p + 1 + ti);
                            c0 = 16 * cq;
                            useT = (cq == p);           // V[p][p] after the panel step is T itself (kept in the pivot stage)
                        }
                        const double* bsrc = useT ? &X[lk * SR_PD_TLD + 16 + ln] : &S[(j0 + lk) * SR_PD_LD + c0 + ln];
                        const int bld = useT ? 4 * SR_PD_TLD : 4 * SR_PD_LD;
                        double af[4], bf[4];
                        d4_t cc;
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) {
                            af[kk] = S[(j0 + 4 * kk + lk) * SR_PD_LD + r0 + ln];
                            bf[kk] = bsrc[kk * bld];
                        }
#pragma unroll
                        for (int q = 0; q < 4; ++q) cc[q] = S[(r0 + lk + 4 * q) * SR_PD_LD + c0 + ln];
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) cc = __builtin_amdgcn_mfma_f64_16x16x4f64(-af[kk], bf[kk], cc, 0, 0, 0);
#pragma unroll
                        for (int q = 0; q < 4; ++q) S[(r0 + lk + 4 * q) * SR_PD_LD + c0 + ln] = cc[q];
                    }
                } else {
                    const int sw = (wave >> 2) - 1;            // 0, 1, 2
                    if (sw == 0) {
                        // the pivot stage of this panel: U_pp (zeros below its diagonal), T_p (lower) and T_p^T (upper)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int r = lk + 4 * q;
                            const double u = X[r * SR_PD_TLD + ln], t = X[r * SR_PD_TLD + 16 + ln];
                            if (storeA) Ag[(long)(j0 + r) * lda + j0 + ln] = (r <= ln) ? u : 0.0;
                            w_diag[(long)(j0 + r) * ldw + j0 + ln] = t;
                            wt_diag[(long)(j0 + ln) * ldw + j0 + r] = t;
                        }
                        if (nbt > 0 && storeA) {                // wavefront 0 left the store of U[p][p+1] to us
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                Ag[(long)(j0 + lk + 4 * q) * lda + j0 + 16 + ln] = S[(j0 + lk + 4 * q) * SR_PD_LD + j0 + 16 + ln];
                        }
                    }
                    // structural zeros: 28 pairs (strictly-lower tile (zr, zc) of A and U^-1, its mirror of U^-T), 4 per
                    // panel, wavefronts 8 and 12 two each (wavefront 4 has the pivot stage)
                    if (sw > 0 && p < 7) {
                        for (int i = 0; i < 2; ++i) {
                            int z = 4 * p + 2 * (sw - 1) + i, zr = 1;
                            while (z >= zr) { z -= zr; ++zr; }
                            const int zc = z;
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const long rr = 16 * zr + lk + 4 * q, cc = 16 * zc + ln;
                                if (storeA) Ag[rr * lda + cc] = 0.0;
                                wt_diag[rr * ldw + cc] = 0.0;
                                w_diag[(long)(16 * zc + lk + 4 * q) * ldw + 16 * zr + ln] = 0.0;
                            }
                        }
                    }
                }
            }
        }
        sr_pd_barrier();
    }
    if (fail) {
        __syncthreads();                                // every store above has landed before the block is replaced
        if (tid == 0 && *info == 0) *info = (int)k0 + fail;
        // keep downstream kernels finite: identity block
        for (int idx = tid; idx < SR_NB * SR_NB; idx += SR_PD_THREADS) {
            const int r = idx >> 7, c = idx & 127;
            const double v = (r == c) ? 1.0 : 0.0;
            if (storeA) Ag[(long)r * lda + c] = v;
            wt_diag[(long)r * ldw + c] = v;
            w_diag[(long)r * ldw + c] = v;
        }
    }
}

__global__ __launch_bounds__(SR_PD_THREADS, 1) void sr_potrf_diag_kernel(double* A, long lda,
                                                                         double* wt_diag, double* w_diag,
                                                                         long ldw, int kb, int* info, int skip,
                                                                         sr_batch bt) {
    // skip (sr_test_potrf_diag; 0 in production): 64 = leave A untouched (back-to-back timing on one input)
    __shared__ double S[SR_NB * SR_PD_LD];
    __shared__ double Xb[2][16 * SR_PD_TLD];
    __shared__ int fail, done;
    A += (long)blockIdx.x * bt.sA; wt_diag += (long)blockIdx.x * bt.sB; w_diag += (long)blockIdx.x * bt.sC;
    info += blockIdx.x;
    const long k0 = (long)kb * SR_NB;
    __builtin_amdgcn_s_setprio(3);
    sr_pd_block(A + k0 * lda + k0, lda, wt_diag, w_diag, ldw, k0, info, !(skip & 64), S, Xb, fail, done);
}

// ------------------------------------------------------------------------------------------------
// RESIDENT form: the diagonal blocks of the tile-flow Cholesky (round 6; sr_flow.hip has the picture, sr_flow.h the counters).
// ONE workgroup per output is launched in front of everything else, keeps its CU for the whole factorisation and factors
// block after block: it waits until the three upper 64 x 64 tiles of block kb have taken their last update (ac[kb][2 kb] >= 1,
// ac[kb][2 kb + 1] >= 2; first row of a panel: ap[kb][kb]; block 0: until the Gram matrix is there, SR_FLOW_GO), factors and inverts, and publishes dd[kb] = 1,
// which the block-row solves of the worker kernel wait for.  What comes in is read behind an agent-scope acquire, what goes
// out leaves in front of an agent-scope release (the L2s of the XCDs are not coherent with each other).  A wait beyond its
// time-out, or a raised status word, ends the workgroup (the host then repeats the update by launches).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(SR_PD_THREADS, 1) void sr_flow_diag_server_kernel(double* A, long lda, double* Wt, double* W,
                                                                               long ldw, int nb, int panel, int* info, sr_batch bt,
                                                                               unsigned* flags, unsigned epoch,
                                                                               unsigned long long timeout_go,
                                                                               unsigned long long timeout) {
    __shared__ double S[SR_NB * SR_PD_LD];
    __shared__ double Xb[2][16 * SR_PD_TLD];
    __shared__ int fail, done, go;
    const int d = blockIdx.x;
    A += (long)d * bt.sA; Wt += (long)d * bt.sB; W += (long)d * bt.sC;
    info += d;
    unsigned* dd = flags + SR_FLOW_HDR + (long)d * sr_flow_words(nb);
    const unsigned* ac = dd + nb;
    const int nt = 2 * nb;
    const unsigned* ap = ac + 2L * nb * nt;
    unsigned* tk = dd + nb + 9L * nb * nb;
    unsigned long long t_go = 0;
    unsigned* status = flags + SR_FLOW_STATUS;
    __builtin_amdgcn_s_setprio(3);
    if (threadIdx.x == 0) __hip_atomic_store(flags + SR_FLOW_ALIVE + d, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll 1
    for (int kb = 0; kb < nb; ++kb) {
        if (threadIdx.x == 0) {
            int ok = 1;
            const unsigned long long t0 = wall_clock64();
            unsigned spins = 0;
            if (kb == 0) {
                // (the status word may still hold the previous run's value until the memset in front of this run's Gram
                //  kernel has run: not looked at here, and a time-out just leaves)
                while (__hip_atomic_load(flags + SR_FLOW_GO, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
                    __builtin_amdgcn_s_sleep(4);
                    if ((++spins & 63) == 0 && wall_clock64() - t0 > timeout_go) { ok = 0; break; }
                }
                t_go = wall_clock64();
            } else {
                // first row of a panel: the block is what the updates by the panels in front left (ap counts them); otherwise
                // its three upper tiles have taken the panel's rows above it
                const bool first = false && kb % panel == 0;   // (every block's band is updated in 64-tiles now: sr_flow.hip)
                const unsigned* a0 = first ? ap + (long)kb * nb + kb : ac + (long)kb * nt + 2 * kb;
                const unsigned* a1 = first ? a0 : a0 + 1;
                const unsigned v0 = first ? (unsigned)(kb / panel) : 1u, v1 = first ? v0 : 2u;
                while (__hip_atomic_load(a0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < v0 ||
                       __hip_atomic_load(a1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < v1) {
                    __builtin_amdgcn_s_sleep(1);
                    if ((++spins & 63) != 0) continue;
                    if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { ok = 0; break; }
                    if (wall_clock64() - t0 > timeout) {
                        __hip_atomic_store(status, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ok = 0;
                        break;
                    }
                }
            }
            go = ok;
        }
        __syncthreads();
        if (!go) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const long k0 = (long)kb * SR_NB;
        // (the leading dimensions are made opaque per block: otherwise every per-lane address of the block routine is computed
        //  once in front of the loop and kept -- 128 VGPRs at 16 wavefronts, 84 bytes of scratch per lane)
        long lda_i = lda, ldw_i = ldw;
        double *A_i = A, *Wt_i = Wt, *W_i = W;
        asm volatile("" : "+s"(lda_i), "+s"(ldw_i), "+s"(A_i), "+s"(Wt_i), "+s"(W_i));
        const long dg = k0 * ldw_i + k0;
        sr_pd_block(A_i + k0 * lda_i + k0, lda_i, Wt_i + dg, W_i + dg, ldw_i, k0, info, true, S, Xb, fail, done);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");       // every thread: its stores are out before the word below
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_store(dd + kb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            tk[kb] = (unsigned)(wall_clock64() - t_go);
        }
    }
}

int sr_launch_flow_diag_server(double* A, long lda, double* Wt, double* W, long ldw, int nb, int panel, int* info_dev,
                               unsigned* flags, unsigned epoch, double timeout_go_s, double timeout_s, hipStream_t s,
                               const sr_batch* btp) {
    const sr_batch bt = btp ? *btp : sr_batch{};
    hipLaunchKernelGGL(sr_flow_diag_server_kernel, dim3(bt.n), dim3(SR_PD_THREADS), 0, s, A, lda, Wt, W, ldw, nb, panel, info_dev, bt,
                       flags, epoch, (unsigned long long)(timeout_go_s * 1e8), (unsigned long long)(timeout_s * 1e8));
    SR_HIP(hipGetLastError());
    return SR_OK;
}

int sr_launch_potrf_diag(double* A, long lda, double* wt_diag, double* w_diag, long ldw, int kb,
                         int* info_dev, hipStream_t s, int skip, const sr_batch* btp) {
    const sr_batch bt = btp ? *btp : sr_batch{};
    hipLaunchKernelGGL(sr_potrf_diag_kernel, dim3(bt.n), dim3(SR_PD_THREADS), 0, s, A, lda, wt_diag, w_diag, ldw,
                       kb, info_dev, skip, bt);
    SR_HIP(hipGetLastError());
    return SR_OK;
}

// ------------------------------------------------------------------------------------------------
// Hand-over between the hardware queues of the pipelined model update (round 6; sr_capi_update.hip): ONE thread that first
// publishes a counter value (what the kernels in front of it on ITS stream have finished) and then waits until up to two
// other counters -- published the same way from other streams -- have reached theirs.  Whatever is behind it on its
// stream starts with the acquire every kernel start carries, so the data those counters stand for is visible to it.
// 2.7 us per hand-over measured (scripts/xqueue_handoff.hip) against 12 - 13 us through hipEventRecord /
// hipStreamWaitEvent.  A wait that exceeds `timeout` ticks of the 100 MHz wall clock (a producer that shares the waiter's
// hardware queue would never start) sets *status and lets everything through: the host then repeats the update on the
// plain route.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void sr_fact_handover_kernel(unsigned* set_flag, unsigned set_v, const unsigned* w0, unsigned v0,
                                                              const unsigned* w1, unsigned v1, unsigned* status,
                                                              unsigned long long timeout) {
    if (threadIdx.x != 0) return;
    if (set_flag) __hip_atomic_store(set_flag, set_v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long t0 = wall_clock64();
    for (int i = 0; i < 2; ++i) {
        const unsigned* w = i ? w1 : w0;
        const unsigned v = i ? v1 : v0;
        if (!w) continue;
        // counters only grow (the epoch of the update is the value waited for); signed difference: wrap-safe
        unsigned spins = 0;
        while ((int)(__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - v) < 0) {
            // a polite poller: ~0.4 us between looks (single-workgroup kernels land on the same first CUs as this wave; a
            // tight loop competes with them for instruction issue)
            __builtin_amdgcn_s_sleep(16);
            if ((++spins & 63) != 0) continue;
            if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
            if (wall_clock64() - t0 > timeout) {
                __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
        }
    }
}

int sr_launch_fact_handover(unsigned* set_flag, unsigned set_v, const unsigned* w0, unsigned v0, const unsigned* w1,
                            unsigned v1, unsigned* status, double timeout_s, hipStream_t s) {
    hipLaunchKernelGGL(sr_fact_handover_kernel, dim3(1), dim3(64), 0, s, set_flag, set_v, w0, v0, w1, v1, status,
                       (unsigned long long)(timeout_s * 1e8));
    SR_HIP(hipGetLastError());
    return SR_OK;
}

// ------------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------------
// dst[c][r] = src[r][c] for an (rows x cols) block; rows, cols multiples of 32
__global__ __launch_bounds__(256) void sr_transpose_kernel(const double* __restrict__ src, long lds_,
                                                           double* __restrict__ dst, long ldd) {
    __shared__ double t[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int r = ty; r < 32; r += 8) t[r][tx] = src[(long)(by + r) * lds_ + bx + tx];
    __syncthreads();
    for (int r = ty; r < 32; r += 8) dst[(long)(bx + r) * ldd + by + tx] = t[tx][r];
}

int sr_launch_transpose_rect(const double* src, long lds_, double* dst, long ldd, int rows, int cols,
                             hipStream_t s) {
    SR_CHECK(rows % 32 == 0 && cols % 32 == 0 && rows > 0 && cols > 0, SR_EINVAL,
             "transpose: %d x %d not multiples of 32", rows, cols);
    hipLaunchKernelGGL(sr_transpose_kernel, dim3(cols / 32, rows / 32), dim3(256), 0, s, src, lds_, dst, ldd);
    SR_HIP(hipGetLastError());
    return SR_OK;
}

int sr_launch_transpose(const double* src, double* dst, int n, hipStream_t s) {
    return sr_launch_transpose_rect(src, n, dst, n, n, n, s);
}

// one wavefront per row, eight independent loads per lane in flight; shuffle reduction.  blockIdx.y = batch member
// (matrix, x and y `sM`, `sx`, `sy` doubles apart).  (One load per lane and iteration: 46 us per 105 MB triangle; the four
// products behind a model update with two outputs were 0.18 ms at its very end.)
__global__ __launch_bounds__(256) void sr_trmv_kernel(const double* __restrict__ M, long ld,
                                                      const double* __restrict__ x,
                                                      double* __restrict__ y, int n, int lower, long sM, long sx,
                                                      long sy) {
    // longest rows first (a row is one wavefront's chain of dependent load batches: scheduled last, the 5120-element
    // rows of a lower triangle were the tail of the launch)
    const int slot = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (slot >= n) return;
    const int row = lower ? n - 1 - slot : slot;
    M += (long)blockIdx.y * sM; x += (long)blockIdx.y * sx; y += (long)blockIdx.y * sy;
    const int c0 = lower ? 0 : row, c1 = lower ? row + 1 : n;
    const double* m = M + (long)row * ld;
    double acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = 0.0;
    int c = c0 + lane;
    for (; c + 7 * 64 < c1; c += 8 * 64) {
        double mv[8], xv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { mv[u] = m[c + 64 * u]; xv[u] = x[c + 64 * u]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] = fma(mv[u], xv[u], acc[u]);
    }
    for (; c < c1; c += 64) acc[0] = fma(m[c], x[c], acc[0]);
    double s = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) y[row] = s;
}

int sr_launch_trmv(const double* M, long ld, const double* x, double* y, int n, int lower,
                   hipStream_t s, int nbatch, long sM, long sx, long sy) {
    hipLaunchKernelGGL(sr_trmv_kernel, dim3((n + 3) / 4, nbatch), dim3(256), 0, s, M, ld, x, y, n, lower, sM, sx, sy);
    SR_HIP(hipGetLastError());
    return SR_OK;
}

__global__ __launch_bounds__(256) void sr_fill_kernel(double* p, size_t n, double v) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) p[i] = v;
}

int sr_launch_fill(double* p, size_t n, double v, hipStream_t s) {
    if (n == 0) return SR_OK;
    size_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(sr_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p, n, v);
    SR_HIP(hipGetLastError());
    return SR_OK;
}

// ------------------------------------------------------------------------------------------------
// log det(K_y) per output from the diagonal of U^-1:  log det = -2 sum_i log (U^-1)_ii.
// One workgroup per output; the padding block has unit diagonal and contributes nothing.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sr_logdet_kernel(const double* __restrict__ Wt, int Np,
                                                        double* __restrict__ out) {
    __shared__ double red[4];
    const double* W = Wt + (size_t)blockIdx.x * Np * Np;
    double acc = 0.0;
    for (int i = threadIdx.x; i < Np; i += 256) acc += log(W[(size_t)i * Np + i]);
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = -2.0 * (red[0] + red[1] + red[2] + red[3]);
}

int sr_launch_logdet(const double* Wt, int Np, int n_out, double* out, hipStream_t s) {
    hipLaunchKernelGGL(sr_logdet_kernel, dim3(n_out), dim3(256), 0, s, Wt, Np, out);
    SR_HIP(hipGetLastError());
    return SR_OK;
}
