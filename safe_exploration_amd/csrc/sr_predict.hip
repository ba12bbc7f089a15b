// sr_predict.hip -- batched GP posterior: the per-control-step inference kernels.
//
//   K1 sr_kstar_kernel : RBF cross-covariance K*(Z, Xq) written once to HBM (k-major, queries
//                        contiguous) fused with mu = k* alpha and d mu/dx      [VALU/exp bound]
//   K2 sr_var_kernel   : |W k*|^2 = column norms of the block-triangular product Wt^T K*, on the
//                        fp64 matrix cores, never storing the product          [MFMA bound, dominant]
//   K3 sr_finalize     : var = sf2 - sum_rb part, clip 1e-15, transpose to the API layout
//
// formulas: /root/reference/safe_exploration/ssm_gpy/gp_models_utils_casadi.py:17-40,160-197
// shapes:   /root/reference/safe_exploration/ssm_gpy/gaussian_process.py:546-596
#include "sr_mfma_tile.h"
#include "sr_final_dev.h"
#include <cstdlib>

// ------------------------------------------------------------------------------------------------
// K1: one thread per (query, output); training inputs staged through LDS in tiles of 256 rows,
// pre-scaled by 1/lengthscale so a pair costs D sub + D fma + exp; all lanes read the same LDS
// address (broadcast).  K* stores: 64 consecutive queries per wavefront = 512 B contiguous.
// ------------------------------------------------------------------------------------------------
#define SR_ZT 256
// QPT queries per thread (t, t + 256, ..): the workgroup's stores of a row are one contiguous run of QPT x 2 KiB, the
// LDS reads of the training row are shared by QPT kernel evaluations, and QPT independent exp chains interleave.
// ADJ (round 6, QPT == 2 only): the two queries of a thread are NEIGHBOURS (t, t + 1) and a row of K* leaves as one 16-byte
// store per lane (1 KiB contiguous per wavefront) instead of two 8-byte stores 2 KiB apart; ADJ == 2: non-temporal (what the
// big batches use).
template <int DT, int QPT, int ADJ = 0>
__global__ __launch_bounds__(256) void sr_kstar_kernel(sr_kstar_args a) {
    __shared__ double zs[SR_ZT * DT];
    __shared__ double al[SR_ZT];
    static_assert(ADJ == 0 || QPT == 2, "adjacent queries come in pairs");
    constexpr int QS = ADJ ? 1 : 256;                     // distance of a thread's queries
    const int d = blockIdx.y, sp = blockIdx.z;
    const long tb = (long)blockIdx.x * 256 * QPT + (ADJ ? 2 * threadIdx.x : threadIdx.x);
    const long tpad = a.Tw ? a.Tw : a.Tp;
    bool live[QPT], inpad[QPT], any = false;
#pragma unroll
    for (int q = 0; q < QPT; ++q) {
        live[q] = tb + QS * q < a.T;
        inpad[q] = tb + QS * q < tpad;
        any |= inpad[q];
    }

    double inv_l[DT], xs[QPT][DT], g[QPT][DT], mu[QPT];
#pragma unroll
    for (int j = 0; j < DT; ++j) inv_l[j] = (j < a.D) ? 1.0 / a.ls[d * a.D + j] : 0.0;
#pragma unroll
    for (int q = 0; q < QPT; ++q) {
        const long t = tb + QS * q;
        mu[q] = 0.0;
#pragma unroll
        for (int j = 0; j < DT; ++j) {
            double x = 0.0;
            if (live[q] && j < a.D)
                x = (j < a.na) ? a.xa[t * a.lda + j] : a.xb[t * a.ldb + (j - a.na)];
            xs[q][j] = x * inv_l[j];
            g[q][j] = 0.0;
        }
    }
    const double sf2 = a.sf2[d];

    const int off = a.Np - a.N;
    const int rows_per = (a.Np + a.nsplit - 1) / a.nsplit;
    const int i_beg = sp * rows_per;
    const int i_end = min(a.Np, i_beg + rows_per);
    double* ks_col = a.Ks + (long)d * a.Np * a.Tp + tb;

    for (int i0 = i_beg; i0 < i_end; i0 += SR_ZT) {
        const int nrow = min(SR_ZT, i_end - i0);
        __syncthreads();
        {
            const int r = threadIdx.x;
            const int i = i0 + r - off;       // training index (front padding)
            const bool ok = (r < nrow) && (i >= 0);
#pragma unroll
            for (int j = 0; j < DT; ++j)
                zs[r * DT + j] = (ok && j < a.D) ? a.Z[(long)i * a.D + j] * inv_l[j] : 0.0;
            al[r] = ok ? a.alpha[(long)d * a.Np + i0 + r] : 0.0;
        }
        __syncthreads();
        const int rpad = min(nrow, max(0, off - i0));     // leading padding rows of this tile
        if (any) {
            for (int r = 0; r < rpad; ++r)
#pragma unroll
                for (int q = 0; q < QPT; ++q)
                    if (inpad[q]) ks_col[(long)(i0 + r) * a.Tp + QS * q] = 0.0;
            for (int r = rpad; r < nrow; ++r) {
                double z[DT];
#pragma unroll
                for (int j = 0; j < DT; ++j) z[j] = zs[r * DT + j];
                const double alr = al[r];
                double kq[QPT];
#pragma unroll
                for (int q = 0; q < QPT; ++q) {
                    double diff[DT];
                    double r2 = 0.0;
#pragma unroll
                    for (int j = 0; j < DT; ++j) {
                        diff[j] = xs[q][j] - z[j];
                        r2 = fma(diff[j], diff[j], r2);
                    }
                    const double k = live[q] ? sf2 * exp(-0.5 * r2) : 0.0;
                    kq[q] = k;
                    if (!ADJ && inpad[q]) ks_col[(long)(i0 + r) * a.Tp + 256 * q] = k;
                    const double w = k * alr;
                    mu[q] += w;
#pragma unroll
                    for (int j = 0; j < DT; ++j) g[q][j] = fma(-w, diff[j], g[q][j]);
                }
                if (ADJ) {                        // (tpad is a multiple of 128: a pair is inside the padded range or outside)
                    typedef double sr_d2 __attribute__((ext_vector_type(2)));
                    sr_d2* dst = reinterpret_cast<sr_d2*>(ks_col + (long)(i0 + r) * a.Tp);
                    const sr_d2 v = {kq[0], kq[QPT - 1]};
                    if (inpad[0]) { if (ADJ == 2) __builtin_nontemporal_store(v, dst); else *dst = v; }
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < QPT; ++q)
        if (inpad[q]) {
            const long t = tb + QS * q;
            a.mu_part[((long)sp * a.n_out + d) * a.Tp + t] = mu[q];
#pragma unroll
            for (int j = 0; j < DT; ++j)
                if (j < a.D)
                    a.jac_part[(((long)sp * a.n_out + d) * a.D + j) * a.Tp + t] = g[q][j] * inv_l[j];
        }
}

// K1g: the same pass for the general kernel family of sr_common.h (Matern-5/2, linear x stationary +
// linear).  Inputs stay unscaled in LDS because the linear parts need x_j z_j; per pair
//   k = (c0 + sum a_j x_j z_j) v kappa(r) + sum b_j x_j z_j
//   dk/dx_j = a_j z_j v kappa + (c0 + ...) v g (x_j - z_j) s_j^2 + b_j z_j ,  g = kappa'(r)/r
// and the prior variance k(x,x) = (c0 + sum a_j x_j^2) v + sum b_j x_j^2 is emitted per query.
template <int DT>
__global__ __launch_bounds__(256) void sr_kstar_general_kernel(sr_kstar_args a) {
    __shared__ double zs[SR_ZT * DT];
    __shared__ double al[SR_ZT];
    const int d = blockIdx.y, sp = blockIdx.z;
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const bool live = t < a.T;
    const bool inpad = t < (a.Tw ? a.Tw : a.Tp);
    const double* kp = a.kp + (long)d * SR_KP(a.D);
    const int kind = (int)kp[0];
    const double var = kp[1], c0 = kp[2];
    double s2[DT], ax[DT], bx[DT], av[DT], bv[DT], x[DT], g[DT];
    double kxx = c0 * var;
#pragma unroll
    for (int j = 0; j < DT; ++j) {
        const double sj = (j < a.D) ? kp[3 + j] : 0.0;
        av[j] = (j < a.D) ? kp[3 + a.D + j] : 0.0;
        bv[j] = (j < a.D) ? kp[3 + 2 * a.D + j] : 0.0;
        double xv = 0.0;
        if (live && j < a.D) xv = (j < a.na) ? a.xa[t * a.lda + j] : a.xb[t * a.ldb + (j - a.na)];
        x[j] = xv;
        s2[j] = sj * sj;
        ax[j] = av[j] * xv;
        bx[j] = bv[j] * xv;
        kxx = fma(ax[j] * var + bx[j], xv, kxx);
        g[j] = 0.0;
    }
    double mu = 0.0;
    const int off = a.Np - a.N;
    const int rows_per = (a.Np + a.nsplit - 1) / a.nsplit;
    const int i_beg = sp * rows_per;
    const int i_end = min(a.Np, i_beg + rows_per);
    double* ks_col = a.Ks + (long)d * a.Np * a.Tp + t;
    for (int i0 = i_beg; i0 < i_end; i0 += SR_ZT) {
        const int nrow = min(SR_ZT, i_end - i0);
        __syncthreads();
        {
            const int r = threadIdx.x;
            const int i = i0 + r - off;
            const bool ok = (r < nrow) && (i >= 0);
#pragma unroll
            for (int j = 0; j < DT; ++j) zs[r * DT + j] = (ok && j < a.D) ? a.Z[(long)i * a.D + j] : 0.0;
            al[r] = ok ? a.alpha[(long)d * a.Np + i0 + r] : 0.0;
        }
        __syncthreads();
        const int rpad = min(nrow, max(0, off - i0));
        if (inpad) {
            for (int r = 0; r < rpad; ++r) ks_col[(long)(i0 + r) * a.Tp] = 0.0;
            for (int r = rpad; r < nrow; ++r) {
                double diff[DT];
                double r2 = 0.0, la = 0.0, lb = 0.0;
#pragma unroll
                for (int j = 0; j < DT; ++j) {
                    const double z = zs[r * DT + j];
                    diff[j] = x[j] - z;
                    r2 = fma(diff[j] * s2[j], diff[j], r2);
                    la = fma(ax[j], z, la);
                    lb = fma(bx[j], z, lb);
                }
                double kap, gk;              // kappa and kappa'(r)/r
                if (kind == 0) {
                    kap = exp(-0.5 * r2);
                    gk = -kap;
                } else {
                    const double rr = sqrt(r2);
                    const double e = exp(-2.23606797749978969641 * rr);
                    kap = (1.0 + 2.23606797749978969641 * rr + (5.0 / 3.0) * r2) * e;
                    gk = -(5.0 / 3.0) * (1.0 + 2.23606797749978969641 * rr) * e;
                }
                const double pre = (c0 + la) * var;
                const double k = live ? fma(pre, kap, lb) : 0.0;
                ks_col[(long)(i0 + r) * a.Tp] = k;
                const double w = live ? al[r] : 0.0;
                mu = fma(w, k, mu);
                const double wk = w * var * kap, wg = w * pre * gk;
#pragma unroll
                for (int j = 0; j < DT; ++j) {
                    const double z = zs[r * DT + j];
                    g[j] = fma(wk * av[j] + w * bv[j], z, fma(wg * s2[j], diff[j], g[j]));
                }
            }
        }
    }
    if (inpad) {
        a.mu_part[((long)sp * a.n_out + d) * a.Tp + t] = mu;
        if (sp == 0) a.kxx[(long)d * a.Tp + t] = kxx;
#pragma unroll
        for (int j = 0; j < DT; ++j)
            if (j < a.D) a.jac_part[(((long)sp * a.n_out + d) * a.D + j) * a.Tp + t] = g[j];
    }
}

int sr_launch_kstar(const sr_kstar_args& a, hipStream_t s) {
    dim3 grid((unsigned)(((a.Tw ? a.Tw : a.Tp) + 255) / 256), a.n_out, a.nsplit);
    if (a.kp) {
#define SR_KSTARG_CASE(DT) hipLaunchKernelGGL(sr_kstar_general_kernel<DT>, grid, dim3(256), 0, s, a)
        if (a.D <= 3) SR_KSTARG_CASE(3);
        else if (a.D <= 5) SR_KSTARG_CASE(5);
        else if (a.D <= 8) SR_KSTARG_CASE(8);
        else if (a.D <= 12) SR_KSTARG_CASE(12);
        else { sr_set_error("kstar: D=%d > %d", a.D, SR_MAX_D); return SR_EUNSUPPORTED; }
#undef SR_KSTARG_CASE
        SR_HIP(hipGetLastError());
        return SR_OK;
    }
#define SR_KSTAR_CASE(DT) \
    hipLaunchKernelGGL((sr_kstar_kernel<DT, 1>), grid, dim3(256), 0, s, a)
    // big batches: two queries per thread (65536 queries at N = 5000: 1.32 -> 1.19 ms; four: 1.32 -- the registers
    // of four exp chains cost the occupancy what the sharing buys)
    if (a.D <= 5 && (a.Tw ? a.Tw : a.Tp) >= 8192) {
        const long tpad = a.Tw ? a.Tw : a.Tp;
        dim3 g2((unsigned)((tpad + 511) / 512), a.n_out, a.nsplit);
        // round 6: the two queries of a thread are neighbours and a row of K* leaves as ONE non-temporal 16-byte store per lane
        // (same box, 65536 queries at N = 5000: 1.49 ms with two 8-byte stores 2 KiB apart, 1.47 with plain 16-byte stores,
        // 1.28 non-temporal -- the 5.2 GB pass through the caches once; profiles/r06_headline_sweeps.txt)
        if (a.D <= 3) hipLaunchKernelGGL((sr_kstar_kernel<3, 2, 2>), g2, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((sr_kstar_kernel<5, 2, 2>), g2, dim3(256), 0, s, a);
        SR_HIP(hipGetLastError());
        return SR_OK;
    }
    if (a.D <= 3) SR_KSTAR_CASE(3);
    else if (a.D <= 5) SR_KSTAR_CASE(5);
    else if (a.D <= 8) SR_KSTAR_CASE(8);
    else if (a.D <= 12) SR_KSTAR_CASE(12);
    else { sr_set_error("kstar: D=%d > %d", a.D, SR_MAX_D); return SR_EUNSUPPORTED; }
#undef SR_KSTAR_CASE
    SR_HIP(hipGetLastError());
    return SR_OK;
}

// ------------------------------------------------------------------------------------------------
// K2: for row block rb (128 rows of V = W K*) and query tile x (128 queries):
//        V[i][t] = sum_{k <= i} Wt[k][i] * Ks[k][t]     (Wt = U^-1, upper triangular, k-major)
//        part[rb][t] = sum_{i in rb} V[i][t]^2
// K range is [0, (rb+1)*128): the block-triangular structure halves the flops of a dense product.
// Work items are ordered heavy-first (rb descending) inside groups of `group` query tiles; blocks
// b, b+8, ... share an XCD (dispatch is round-robin over the 8 XCDs), so with the query-tile index
// fastest each XCD's L2 sees group/8 query tiles x all row blocks: W tiles are shared between the
// query tiles, K* tiles between the row blocks.
// ------------------------------------------------------------------------------------------------
template <int VARIANT>
__global__ __launch_bounds__(256, 2) void sr_var_kernel(const double* __restrict__ Wt,
                                                        const double* __restrict__ Ks,
                                                        double* __restrict__ part, int Np, long Tp,
                                                        int nrb, int ntq, int group, int k_beg) {
    __shared__ double smem[srt::SMEM_DOUBLES];
    // VARIANT 5 (round 6): a workgroup takes the PAIR of row blocks (nrb - 1 - p, p) of its query tile, one after the other.
    // Every pair contracts over (nrb + 1) * 128 rows in all, so all workgroups of the launch last the same: the 512 that are
    // resident together start together, walk k together and are replaced together.  With single tiles (variants 1 - 4) a
    // row of 64 equal workgroups is replaced as a whole while the seven other resident rows are somewhere else in k: the
    // K* tiles of a query tile are then fetched once per row (L2 hit rate 0.58 at C2': the hits are the U^-1 tiles shared
    // by the eight query tiles of a row on an XCD).  Same tiles, same arithmetic per tile: same bits.
    constexpr bool PAIR = VARIANT == 5;
    const int nrow = PAIR ? (nrb + 1) / 2 : nrb;          // work items per query tile
    const int ngrp = (ntq + group - 1) / group;
    const long per_d_padded = (long)ngrp * nrow * group;
    const long b = blockIdx.x;
    const int d = (int)(b / per_d_padded);
    long rem = b % per_d_padded;
    const int xg = (int)(rem / ((long)nrow * group));
    rem = rem % ((long)nrow * group);
    const int item = (int)(rem / group);
    const int x = xg * group + (int)(rem % group);
    if (x >= ntq) return;
    const double* B = Ks + (long)d * Np * Tp + (long)x * srt::BN;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;

#pragma unroll 1
    for (int pass = 0; pass < (PAIR ? 2 : 1); ++pass) {
        const int rb = PAIR ? (pass == 0 ? nrb - 1 - item : item) : nrb - 1 - item;
        if (PAIR && pass == 1 && rb == nrb - 1 - item) break;            // odd number of row blocks: the middle one is alone
        const double* A = Wt + (long)d * Np * Np + (long)rb * srt::BM;
        srt::Acc acc;
        acc.zero();
        // VARIANT 4 / 5 leave out the structural zeros of the diagonal block (row tiles interleaved over the two wavefront
        // rows; the epilogue below sums over all rows of the block, so the row assignment does not show).
        // VARIANT 3 / 4 / 5: the pipelined loop of round 5 (barrier under the MFMA stream)
        if (VARIANT >= 4) srt::mainloop_tn_pipe<true>(A, Np, B, Tp, k_beg, (rb + 1) * srt::BM, smem, acc);
        else if (VARIANT == 3) srt::mainloop_tn_pipe<false>(A, Np, B, Tp, k_beg, (rb + 1) * srt::BM, smem, acc);
        else srt::mainloop_tn_glds<16>(A, Np, B, Tp, k_beg, (rb + 1) * srt::BM, smem, acc);      // (1: the loop of rounds 1 - 4)

        double s[4];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            double v = 0.0;
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int r = 0; r < 4; ++r) v = fma(acc.v[mi][ni][r], acc.v[mi][ni][r], v);
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            s[ni] = v;
        }
        // the main loop ended with a barrier: smem is free.  red[wm][128]
        double* red = smem;
        if (lane < 16) {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) red[wm * 128 + wn * 64 + ni * 16 + lane] = s[ni];
        }
        __syncthreads();
        if (threadIdx.x < 128) {
            const double v = red[threadIdx.x] + red[128 + threadIdx.x];
            part[((long)d * nrb + rb) * Tp + (long)x * srt::BN + threadIdx.x] = v;
        }
        if (PAIR) __syncthreads();                                       // `red` is read: the next tile's DMA may land
    }
}

int sr_launch_var(const double* Wt, const double* Ks, double* part, int N, int Np, long Tp, int n_out,
                  int group, int variant, hipStream_t s, long Tw) {
    // Tw != 0: only the first Tw columns behind the (pre-offset) Ks / part pointers; Tp stays the row stride
    const int k_beg = ((Np - N) / srt::BK) * srt::BK;     // rows k < Np-N are padding: K* is zero there
    const int nrb = Np / srt::BM;
    const int ntq = (int)((Tw ? Tw : Tp) / srt::BN);
    if (group < 1) group = 1;
    if (group > ntq) group = ntq;
    const int ngrp = (ntq + group - 1) / group;
    const long blocks = (long)n_out * ngrp * (variant == 5 ? (nrb + 1) / 2 : nrb) * group;
    SR_CHECK(blocks < 2147483647L, SR_EINVAL, "var: grid too large (%ld blocks)", blocks);
#ifdef SR_LAB
    // the A/B forms (sr_gp_set_var_variant): 1 the loop of rounds 1 - 4, 3 the pipelined loop without the diagonal-block walk,
    // 5 pairs of row blocks per workgroup (round 6: profiles/r06_headline_sweeps.txt)
    if (variant == 5) {
        hipLaunchKernelGGL(sr_var_kernel<5>, dim3((unsigned)blocks), dim3(256), 0, s, Wt, Ks, part, Np, Tp, nrb, ntq, group, k_beg);
    } else if (variant == 3) {
        hipLaunchKernelGGL(sr_var_kernel<3>, dim3((unsigned)blocks), dim3(256), 0, s, Wt, Ks, part, Np, Tp, nrb, ntq, group, k_beg);
    } else if (variant == 1) {
        hipLaunchKernelGGL(sr_var_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, s, Wt, Ks, part, Np, Tp, nrb, ntq, group, k_beg);
    } else
#endif
    {
        (void)variant;
        hipLaunchKernelGGL(sr_var_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, s, Wt, Ks, part, Np, Tp, nrb, ntq, group, k_beg);
    }
    SR_HIP(hipGetLastError());
    return SR_OK;
}


// ------------------------------------------------------------------------------------------------
// K2m: 64 x 64 workgroup tile for SMALL models (Np <= 1024, few workgroups): the regime the reference's
// own experiments run in (N = 25 .. 150 inducing points, a few hundred candidate states per step).
// Same contraction as K2; 4 wavefronts of 32 x 32 (2 x 2 MFMA tiles), register-staged double-buffered
// LDS tiles, triangular K range at 64-row granularity.  A k-step costs 16 MFMAs per wavefront instead of
// 64, so the critical path of the (tiny) grid is 4x shorter.
// ------------------------------------------------------------------------------------------------
#define SR_T64 64
#define SR_LD64 80      // 64 + 16 doubles: row stride 160 dwords == 32 mod 64 (conflict-free ds_read_b64)
__global__ __launch_bounds__(256) void sr_var64_kernel(const double* __restrict__ Wt,
                                                       const double* __restrict__ Ks,
                                                       double* __restrict__ part, int Np, long Tp, int k_beg) {
    __shared__ double As[2][16][SR_LD64];
    __shared__ double Bs[2][16][SR_LD64];
    __shared__ double red[2][SR_T64];
    const int x = blockIdx.x, rb = blockIdx.y, d = blockIdx.z;      // 64-query tile, 64-row block, output
    const int nrb = gridDim.y;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const double* A = Wt + (long)d * Np * Np + (long)rb * SR_T64;
    const double* B = Ks + (long)d * Np * Tp + (long)x * SR_T64;
    const int k_end = (rb + 1) * SR_T64;
    // staging map: 16 rows x 32 double2 per operand = 512 double2; thread takes rows r0, r0 + 8
    const int c2 = tid & 31, r0 = tid >> 5;
    double2 ra0, ra1, rb0, rb1;
    d4_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = d4_t{0.0, 0.0, 0.0, 0.0};
#define SR64_GLOAD(k0)                                                                              \
    do {                                                                                             \
        ra0 = *reinterpret_cast<const double2*>(A + (long)((k0) + r0) * Np + 2 * c2);               \
        ra1 = *reinterpret_cast<const double2*>(A + (long)((k0) + r0 + 8) * Np + 2 * c2);           \
        rb0 = *reinterpret_cast<const double2*>(B + (long)((k0) + r0) * Tp + 2 * c2);               \
        rb1 = *reinterpret_cast<const double2*>(B + (long)((k0) + r0 + 8) * Tp + 2 * c2);           \
    } while (0)
#define SR64_SSTORE(buf)                                                                             \
    do {                                                                                             \
        *reinterpret_cast<double2*>(&As[buf][r0][2 * c2]) = ra0;                                     \
        *reinterpret_cast<double2*>(&As[buf][r0 + 8][2 * c2]) = ra1;                                 \
        *reinterpret_cast<double2*>(&Bs[buf][r0][2 * c2]) = rb0;                                     \
        *reinterpret_cast<double2*>(&Bs[buf][r0 + 8][2 * c2]) = rb1;                                 \
    } while (0)
    if (k_beg < k_end) {
        SR64_GLOAD(k_beg);
        SR64_SSTORE(0);
    }
    __syncthreads();
    int buf = 0;
    for (int k0 = k_beg; k0 < k_end; k0 += 16) {
        const bool more = (k0 + 16) < k_end;
        if (more) SR64_GLOAD(k0 + 16);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            double af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[i] = As[buf][kk * 4 + (lane >> 4)][wm * 32 + i * 16 + (lane & 15)];
                bf[i] = Bs[buf][kk * 4 + (lane >> 4)][wn * 32 + i * 16 + (lane & 15)];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (more) SR64_SSTORE(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
#undef SR64_GLOAD
#undef SR64_SSTORE
    // column sums of squares over the 64 rows: in-lane, across the 4 row groups, across the 2 wavefront rows
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        double v = 0.0;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) v = fma(acc[mi][ni][r], acc[mi][ni][r], v);
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (lane < 16) red[wm][wn * 32 + ni * 16 + lane] = v;
    }
    __syncthreads();
    if (tid < SR_T64) part[((long)d * nrb + rb) * Tp + (long)x * SR_T64 + tid] = red[0][tid] + red[1][tid];
}

int sr_launch_var64(const double* Wt, const double* Ks, double* part, int N, int Np, long Tp, int n_out,
                    hipStream_t s) {
    const int k_beg = ((Np - N) / 16) * 16;
    hipLaunchKernelGGL(sr_var64_kernel, dim3((unsigned)(Tp / SR_T64), Np / SR_T64, n_out), dim3(256), 0, s, Wt, Ks,
                       part, Np, Tp, k_beg);
    SR_HIP(hipGetLastError());
    return SR_OK;
}

// ------------------------------------------------------------------------------------------------
// K2b: few query tiles (<= 1024 plain workgroups on a model of more than two row blocks), BALANCED ("stream-K" shares).
// (K2k -- every tile cut into chunks of 1 / 2 / 4 / 8 k-blocks, rounds 1 - 4 -- lost to this kernel everywhere once the main
//  loop was the pipelined one, also below 256 cells where it used to be the default: 46 against 60 us at N = 512, T = 1100;
//  removed in round 5, profiles/archive/r05_bal_ab.txt.)  The work of the launch is the list of
// k-blocks (128 k-rows of one 128 x 128 output tile), tiles ordered (output, query tile, row block DEScending: heavy tiles
// first), blocks ascending inside a tile; workgroup g of G takes the contiguous share [g U / G, (g + 1) U / G) of its U
// entries -- the same number of MFMAs for everybody, whatever the triangular k range of a tile (K2k cuts every tile into
// chunks of 1 / 2 / 4 / 8 blocks: at N = 5000, T = 128 that is 440 workgroups of up to 4 blocks on 512 slots, CUs with two
// of them take twice as long as CUs with one).  A share covers at most two partial tiles (its first and its last) plus
// whole tiles in between: a whole tile is squared and reduced on the spot, a partial product goes to the workgroup's slot
// (accumulator layout, coalesced), and sr_var_bal_reduce_kernel -- four workgroups per tile, one per row of MFMA tiles of
// the accumulator layout -- adds a tile's segments in ascending k order: deterministic, whoever ran first.
// (First form of round 3: ONE launch, a ticket per tile electing the last arriving share to add the segments.  With 40 row
//  blocks under one or two query tiles a heavy tile has a dozen segments of 128 KB and its last arriver adds them alone at
//  the tail of the launch: N = 5000, T = 128 / 256: 192 / 297 us against 168 / 287 of K2k; with the second launch instead
//  158 / 257, and it is faster everywhere else too -- N = 2000 T = 256 / 512: 96 / 136 -> 83 / 118 us, N = 3000 T = 512:
//  207 -> 198, N = 4000 T = 128 / 512: 167 / 441 -> 120 / 323, N = 5000 T = 512 / 1024: 483 / 896 -> 451 / 844;
//  profiles/archive/r03_streamk.txt.)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ long sr_sk_bound(long g, long U, long G) { return (g * U) / G; }
__device__ __forceinline__ long sr_sk_owner(long u, long U, long G) {        // the g with bound(g) <= u < bound(g + 1)
    long g = (u * G) / U;
    while (g + 1 <= G && sr_sk_bound(g + 1, U, G) <= u) ++g;
    while (g > 0 && sr_sk_bound(g, U, G) > u) --g;
    return g;
}

__global__ __launch_bounds__(256, 2) void sr_var_bal_kernel(const double* __restrict__ Wt, const double* __restrict__ Ks,
                                                            double* Vt, double* __restrict__ part, int Np, long Tp,
                                                            int nrb, int ntq, int k_beg, long U) {
    __shared__ double smem[srt::SMEM_DOUBLES];
    const long G = gridDim.x;
    const long S = (long)nrb * (nrb + 1) / 2;                 // blocks of one (output, query tile)
    // Which share this workgroup takes.  Workgroups b, b + 8, .. run on the same XCD and share its L2: they take the SAME
    // positions inside different (output, query tile) ranges, so that the U^-1 tiles one of them streams are the tiles
    // the others of the same output need at about the same time (identity mapping: every workgroup of an XCD streams
    // its own part of U^-1).  Measured, n_out = 2, T = 1024: N = 2000 205 -> 195 us, N = 3000 365 -> 358, N = 5000 905 -> 895;
    // with 64 ranges (T = 4096) it loses 2 %: only while a range has at least 16 shares.
    long g = blockIdx.x;
    {
        const long ndx = U / S;                                // (output, query tile) ranges, S entries each
        if (ndx > 0 && G % (8 * ndx) == 0 && G / ndx >= 16) {
            const long per = G / ndx, c = g & 7, i = g >> 3;
            g = (i % ndx) * per + c * (per / 8) + i / ndx;
        }
    }
    long u = sr_sk_bound(g, U, G);
    const long u1 = sr_sk_bound(g + 1, U, G);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    while (u < u1) {
        // tile of entry u: (d, x) = u / S; inside: descending row blocks, tile j (rb = nrb - 1 - j) holds nrb - j blocks
        const long dx = u / S;
        const long r = u - dx * S;
        int j = 0;
        long c = 0;                                            // first entry of tile j
        while (c + (nrb - j) <= r) { c += nrb - j; ++j; }
        const int rb = nrb - 1 - j, n = nrb - j;
        const int o = (int)(r - c);                            // first k-block of this segment
        const int len = (int)((n - o < u1 - u) ? n - o : u1 - u);
        const int d = (int)(dx / ntq), x = (int)(dx % ntq);
        const int k0 = (o == 0) ? k_beg : o * srt::BM;
        const int k1 = (o + len) * srt::BM;
        const double* A = Wt + (long)d * Np * Np + (long)rb * srt::BM;
        const double* B = Ks + (long)d * Np * Tp + (long)x * srt::BN;
        srt::Acc acc;
        acc.zero();
        srt::mainloop_tn_pipe<false>(A, Np, B, Tp, k0, k1, smem, acc);
        if (len != n) {
            // slot of a segment: its workgroup's slot 1 if the tile starts in that workgroup's share, else 0
            // (16-byte stores: the 128 KB of a partial product are store-issue bound -- half the instructions of the
            //  8-byte form; element (mi, ni, half h) of lane tid at ((mi 4 + ni) 2 + h) 256 + tid double2's)
            double2* slot = reinterpret_cast<double2*>(Vt + ((g * 2) + (o == 0 ? 1 : 0)) * (long)(srt::BM * srt::BN));
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh)
                        slot[((mi * 4 + ni) * 2 + hh) * 256 + tid] = double2{acc.v[mi][ni][2 * hh], acc.v[mi][ni][2 * hh + 1]};
        } else {                                               // the whole tile was ours
            double sq[4];
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                double v = 0.0;
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int q = 0; q < 4; ++q) v = fma(acc.v[mi][ni][q], acc.v[mi][ni][q], v);
                v += __shfl_xor(v, 16);
                v += __shfl_xor(v, 32);
                sq[ni] = v;
            }
            __syncthreads();                                   // (the main loop's LDS is reused)
            double* red = smem;
            if (lane < 16) {
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) red[wm * 128 + wn * 64 + ni * 16 + lane] = sq[ni];
            }
            __syncthreads();
            // four partial norms per tile (the layout of the reduce pass): the whole norm and three zeros
            double* pt = part + ((long)d * 4 * nrb + rb * 4) * Tp + (long)x * srt::BN;
            if (tid < 128) pt[tid] = red[tid] + red[128 + tid];
            else for (int q = 1; q < 4; ++q) pt[q * Tp + tid - 128] = 0.0;
            __syncthreads();
        }
        u += len;
    }
}

__global__ __launch_bounds__(256) void sr_var_bal_reduce_kernel(const double* __restrict__ Vt, double* __restrict__ part,
                                                                long Tp, int nrb, int ntq, long U, long G) {
    __shared__ double red[256];
    const int mi = blockIdx.x;
    const long dx = blockIdx.y / nrb;
    const int rb = blockIdx.y % nrb, j = nrb - 1 - rb, n = rb + 1;
    const long S = (long)nrb * (nrb + 1) / 2;
    const long t0 = dx * S + (long)j * nrb - (long)j * (j - 1) / 2;        // the tile's first entry (heavy tiles first)
    const long g_first = sr_sk_owner(t0, U, G), g_last = sr_sk_owner(t0 + n - 1, U, G);
    if (g_first == g_last) return;                             // one segment: finished by its workgroup
    const int nseg = (int)(g_last - g_first + 1);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    double2 v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = double2{0.0, 0.0};
    // four segments' loads in flight (a heavy tile has 7 - 13 segments: one dependent round trip each cost 18 us at N = 5000)
    for (int sg0 = 0; sg0 < nseg; sg0 += 4) {
        double2 w[4][8];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int sg = sg0 + b;
            const double2* src = reinterpret_cast<const double2*>(Vt + (((g_first + sg) * 2) + (sg == 0 ? 1 : 0)) * (long)(srt::BM * srt::BN)) + (long)(mi * 8) * 256 + tid;
#pragma unroll
            for (int e = 0; e < 8; ++e) w[b][e] = (sg < nseg) ? src[e * 256] : double2{0.0, 0.0};
        }
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int e = 0; e < 8; ++e) { v[e].x += w[b][e].x; v[e].y += w[b][e].y; }
    }
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        double q = 0.0;
        q = fma(v[ni * 2].x, v[ni * 2].x, q);
        q = fma(v[ni * 2].y, v[ni * 2].y, q);
        q = fma(v[ni * 2 + 1].x, v[ni * 2 + 1].x, q);
        q = fma(v[ni * 2 + 1].y, v[ni * 2 + 1].y, q);
        q += __shfl_xor(q, 16);
        q += __shfl_xor(q, 32);
        if (lane < 16) red[wm * 128 + wn * 64 + ni * 16 + lane] = q;
    }
    __syncthreads();
    const int d = (int)(dx / ntq), x = (int)(dx % ntq);
    if (tid < 128) part[((long)d * 4 * nrb + rb * 4 + mi) * Tp + (long)x * srt::BN + tid] = red[tid] + red[128 + tid];
}

long sr_var_bal_ws(int Np, long Tp, int n_out) {
    const long nrb = Np / srt::BM;
    return sr_var_bal_wgs((long)n_out * (Tp / srt::BN) * nrb * (nrb + 1) / 2) * 2 * (long)(srt::BM * srt::BN);
}
int sr_launch_var_bal(const double* Wt, const double* Ks, double* Vt, double* part, int N, int Np, long Tp, int n_out,
                      hipStream_t s) {
    const int k_beg = ((Np - N) / srt::BK) * srt::BK;
    const int nrb = Np / srt::BM, ntq = (int)(Tp / srt::BN);
    const long U = (long)n_out * ntq * nrb * (nrb + 1) / 2;
    const long G = sr_var_bal_wgs(U);
    hipLaunchKernelGGL(sr_var_bal_kernel, dim3((unsigned)G), dim3(256), 0, s, Wt, Ks, Vt, part, Np, Tp, nrb, ntq, k_beg, U);
    SR_HIP(hipGetLastError());
    hipLaunchKernelGGL(sr_var_bal_reduce_kernel, dim3(4, n_out * ntq * nrb), dim3(256), 0, s, Vt, part, Tp, nrb, ntq, U, G);
    SR_HIP(hipGetLastError());
    return SR_OK;
}

// ------------------------------------------------------------------------------------------------
// K2s: small-batch variance (T <= SR_TS queries, the CasADi/IPOPT callback regime).  With one query
// tile the MFMA kernel is serialised on its longest row block; here U^-1 is streamed exactly once at
// HBM rate instead: thread = one column i of Wt, workgroup = 256 columns x 128 k-rows, K* rows
// broadcast from LDS.  HBM-bound: n_out * Np^2/2 * 8 bytes per call (">= 25 us per eval").
//   pass 1: Vp[chunk][t][i] = sum_{k in chunk, k <= i} Wt[k][i] K*[k][t]
//   pass 2: part[cb][t]     = sum_{i in column block cb} (sum_chunks Vp)^2     (layout of sr_finalize)
// ------------------------------------------------------------------------------------------------
#define SR_TS 16
template <int TQ>   // live queries handled: 1, 4 or SR_TS; Vp holds TQ rows per (column block, k-chunk) pair
__global__ __launch_bounds__(256) void sr_var_small_partial_kernel(const double* __restrict__ Wt,
                                                                   const double* __restrict__ Ks,
                                                                   double* __restrict__ Vp, int Np,
                                                                   long Tp, int npairs, int k_lo) {
    __shared__ double ks[128][TQ];
    const int d = blockIdx.y;
    const int p = blockIdx.x;                       // pair index: column block cb, k-chunk j <= 2 cb + 1
    int cb = (int)((sqrt(4.0 * p + 1.0) - 1.0) * 0.5);
    while ((cb + 1) * (cb + 2) <= p) ++cb;
    while (cb * (cb + 1) > p) --cb;
    const int j = p - cb * (cb + 1);
    const int k0 = j * 128;
    const int i = cb * 256 + threadIdx.x;
    const double* ksrc = Ks + (long)d * Np * Tp + (long)k0 * Tp;
    for (int e = threadIdx.x; e < 128 * TQ; e += 256) {
        const int r = e / TQ, t = e % TQ;
        ks[r][t] = (k0 + r < Np) ? ksrc[(long)r * Tp + t] : 0.0;
    }
    __syncthreads();
    double acc[TQ];
#pragma unroll
    for (int t = 0; t < TQ; ++t) acc[t] = 0.0;
    const double* w = Wt + (long)d * Np * Np + (long)k0 * Np + i;
    const int kmax = (i < Np) ? min(127, i - k0) : -1;   // rows k0 .. k0 + kmax have k <= i
    const int kbeg = max(0, k_lo - k0);             // leading padding rows carry K* == 0
    int r = kbeg;
    for (; r + 15 <= kmax; r += 16) {               // 16 independent 2 KiB row segments in flight per wavefront
        double wv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) wv[u] = w[(long)(r + u) * Np];
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int t = 0; t < TQ; ++t) acc[t] = fma(wv[u], ks[r + u][t], acc[t]);
    }
    for (; r + 3 <= kmax; r += 4) {
        double wv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) wv[u] = w[(long)(r + u) * Np];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int t = 0; t < TQ; ++t) acc[t] = fma(wv[u], ks[r + u][t], acc[t]);
    }
    for (; r <= kmax; ++r) {
        const double wv = w[(long)r * Np];
#pragma unroll
        for (int t = 0; t < TQ; ++t) acc[t] = fma(wv, ks[r][t], acc[t]);
    }
    double* out = Vp + ((long)d * npairs + p) * TQ * 256;
#pragma unroll
    for (int t = 0; t < TQ; ++t) out[t * 256 + threadIdx.x] = acc[t];
}

// 16 live queries: the VALU kernel above spends 16 FMAs and 8 broadcast LDS reads per loaded element (LDS-issue
// bound at ~5 TB/s, and 7 us of serial work per workgroup); here the same (column block, k-chunk) pair is one
// workgroup of 16 wavefronts, wavefront w owning the 16-column strip w on the MFMA 16x16x4 tile: A-fragments
// straight from global (the 16 strips of a row are one contiguous 2 KiB segment), B-fragments = the K* rows in
// LDS, one ds_read_b64 per MFMA.  Same Vp layout as the VALU kernel (the reduce / gather kernels are shared).
__global__ __launch_bounds__(1024) void sr_var_small_partial_mfma_kernel(const double* __restrict__ Wt,
                                                                         const double* __restrict__ Ks,
                                                                         double* __restrict__ Vp, int Np,
                                                                         long Tp, int npairs, int k_lo) {
    __shared__ double ks[128][SR_TS];
    const int d = blockIdx.y;
    const int p = blockIdx.x;
    const int grp = blockIdx.z;                          // group of 16 queries (columns grp*16 .. grp*16+15 of K*)
    int cb = (int)((sqrt(4.0 * p + 1.0) - 1.0) * 0.5);
    while ((cb + 1) * (cb + 2) <= p) ++cb;
    while (cb * (cb + 1) > p) --cb;
    const int j = p - cb * (cb + 1);
    const int k0 = j * 128;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lk = lane >> 4, ln = lane & 15;
    const double* ksrc = Ks + (long)d * Np * Tp + (long)k0 * Tp + grp * SR_TS;
    for (int e = threadIdx.x; e < 128 * SR_TS; e += 1024) {
        const int r = e / SR_TS, t = e % SR_TS;
        ks[r][t] = (k0 + r < Np) ? ksrc[(long)r * Tp + t] : 0.0;
    }
    __syncthreads();
    const int i0 = cb * 256 + 16 * wave;                 // first column of this strip
    d4_t acc = {0.0, 0.0, 0.0, 0.0};
    if (i0 < Np) {
        // k-steps of 4 rows: rows k0 + 4u + lk.  Rows beyond the strip's last column hold zeros of U^-1 (skipped),
        // rows in front of k_lo carry K* == 0 (skipped at k-step granularity).
        // (a chunk may lie beyond the strip and, for an odd multiple of 128 rows, beyond the matrix: (-1) / 4 + 1 == 1)
        const int u_end = (i0 + 15 < k0) ? 0 : min(min(32, (Np - k0) / 4), (i0 + 15 - k0) / 4 + 1);
        int u = max(0, (k_lo - k0) / 4);
        const double* w = Wt + (long)d * Np * Np + (long)(k0 + lk) * Np + i0 + ln;
        for (; u + 16 <= u_end; u += 16) {
            double af[16], bf[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) af[q] = w[(long)(4 * (u + q)) * Np];
#pragma unroll
            for (int q = 0; q < 16; ++q) bf[q] = ks[4 * (u + q) + lk][ln];
#pragma unroll
            for (int q = 0; q < 16; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af[q], bf[q], acc, 0, 0, 0);
        }
        for (; u + 4 <= u_end; u += 4) {
            double af[4], bf[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) af[q] = w[(long)(4 * (u + q)) * Np];
#pragma unroll
            for (int q = 0; q < 4; ++q) bf[q] = ks[4 * (u + q) + lk][ln];
#pragma unroll
            for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af[q], bf[q], acc, 0, 0, 0);
        }
        for (; u < u_end; ++u)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(w[(long)(4 * u) * Np], ks[4 * u + lk][ln], acc, 0, 0, 0);
    }
    // acc[r] = V[column i0 + lk + 4r][query ln]
    double* out = Vp + (((long)grp * gridDim.y + d) * npairs + p) * SR_TS * 256;
#pragma unroll
    for (int r = 0; r < 4; ++r) out[ln * 256 + 16 * wave + lk + 4 * r] = acc[r];
}

// part[cb][t] = sum_{i in column block cb} V_t[i] * (dot0 ? V_0[i] : V_t[i]),  V_t = sum_chunks Vp ;
// grid (ncb, n_out, tq): one query (column) per workgroup.  dot0 serves sr_gp_linearize, whose columns are
// [k*, dk*/dx_j] and whose outputs are the dot products with the k* column.
__global__ __launch_bounds__(256) void sr_var_small_reduce_kernel(const double* __restrict__ Vp,
                                                                  double* __restrict__ part, long Tp,
                                                                  int npairs, int ncb, int tq, int dot0) {
    __shared__ double red[4];
    const int cb = blockIdx.x, d = blockIdx.y, t = blockIdx.z % tq, grp = blockIdx.z / tq;
    const int p0 = cb * (cb + 1), nch = 2 * cb + 2;
    Vp += (long)grp * gridDim.y * npairs * tq * 256;     // query groups of the MFMA streaming kernel
    const double* src = Vp + (((long)d * npairs + p0) * tq + t) * 256 + threadIdx.x;
    double v0 = 0.0, v1 = 0.0;
    for (int j = 0; j + 1 < nch; j += 2) {
        v0 += src[(long)j * tq * 256];
        v1 += src[(long)(j + 1) * tq * 256];
    }
    double w = v0 + v1;
    if (dot0 && t != 0) {
        const double* s0 = Vp + (((long)d * npairs + p0) * tq) * 256 + threadIdx.x;
        double w0 = 0.0, w1 = 0.0;
        for (int j = 0; j + 1 < nch; j += 2) {
            w0 += s0[(long)j * tq * 256];
            w1 += s0[(long)(j + 1) * tq * 256];
        }
        w = w0 + w1;
    }
    double q = (v0 + v1) * w;
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = q;
    __syncthreads();
    if (threadIdx.x == 0) part[((long)d * ncb + cb) * Tp + grp * tq + t] = red[0] + red[1] + red[2] + red[3];
}

// v[d][i] = sum_chunks Vp[..][t][i]  (= (U^-T k*_t)[i]) -- reused by sr_gp_linearize
__global__ __launch_bounds__(256) void sr_var_small_gather_kernel(const double* __restrict__ Vp,
                                                                  double* __restrict__ v, int Np, int npairs,
                                                                  int t, int tq) {
    const int cb = blockIdx.x, d = blockIdx.y;
    const int i = cb * 256 + threadIdx.x;
    if (i >= Np) return;
    const int p0 = cb * (cb + 1), nch = 2 * cb + 2;
    const double* src = Vp + (((long)d * npairs + p0) * tq + t) * 256 + threadIdx.x;
    double v0 = 0.0, v1 = 0.0;
    for (int j = 0; j + 1 < nch; j += 2) {          // same association as the reduce kernel
        v0 += src[(long)j * tq * 256];
        v1 += src[(long)(j + 1) * tq * 256];
    }
    v[(long)d * Np + i] = v0 + v1;
}

static inline int small_tq(int T) { return T <= 1 ? 1 : (T <= 4 ? 4 : SR_TS); }

// all live queries at once: v[(d * tq_out + t) * Np + i], grid (ncb, n_out, nt)
__global__ __launch_bounds__(256) void sr_var_small_gather_all_kernel(const double* __restrict__ Vp,
                                                                      double* __restrict__ v, int Np, int npairs,
                                                                      int tq, int nt) {
    const int cb = blockIdx.x, d = blockIdx.y, t = blockIdx.z;
    const int i = cb * 256 + threadIdx.x;
    if (i >= Np) return;
    const int p0 = cb * (cb + 1), nch = 2 * cb + 2;
    const double* src = Vp + (((long)d * npairs + p0) * tq + t) * 256 + threadIdx.x;
    double v0 = 0.0, v1 = 0.0;
    for (int j = 0; j + 1 < nch; j += 2) {
        v0 += src[(long)j * tq * 256];
        v1 += src[(long)(j + 1) * tq * 256];
    }
    v[((long)d * nt + t) * Np + i] = v0 + v1;
}

int sr_launch_var_small_gather_all(const double* Vp, double* v, int Np, int n_out, int T, hipStream_t s) {
    const int ncb = (Np + 255) / 256;
    hipLaunchKernelGGL(sr_var_small_gather_all_kernel, dim3(ncb, n_out, T), dim3(256), 0, s, Vp, v, Np,
                       ncb * (ncb + 1), small_tq(T), T);
    SR_HIP(hipGetLastError());
    return SR_OK;
}

int sr_launch_var_small_gather(const double* Vp, double* v, int Np, int n_out, int t, int T, hipStream_t s) {
    const int ncb = (Np + 255) / 256;
    hipLaunchKernelGGL(sr_var_small_gather_kernel, dim3(ncb, n_out), dim3(256), 0, s, Vp, v, Np,
                       ncb * (ncb + 1), t, small_tq(T));
    SR_HIP(hipGetLastError());
    return SR_OK;
}

// workspace need of the small path: n_out * npairs * SR_TS * 256 doubles for Vp
long sr_var_small_ws(int Np, int n_out) {
    const int ncb = (Np + 255) / 256;
    return (long)n_out * ncb * (ncb + 1) * SR_TS * 256 * sr_var_small_groups_max(Np, n_out);
}

int sr_launch_var_small(const double* Wt, const double* Ks, double* Vp, double* part, int N, int Np,
                        long Tp, int n_out, int T, hipStream_t s, int dot0, bool reduce) {
    const int ncb = (Np + 255) / 256;              // Np is a multiple of 128: the last block may be half empty
    const int npairs = ncb * (ncb + 1);
    const int k_lo = Np - N;
    const int groups = T > SR_TS ? (T + SR_TS - 1) / SR_TS : 1;
    // plain (cached) loads: at N = 5000 the 210 MB of U^-1 stay in the 256 MB Infinity Cache between calls --
    // measured 6.8 TB/s effective; non-temporal loads were 14 % slower
#define SR_SMALL_LAUNCH(TQ)                                                                              \
    hipLaunchKernelGGL(sr_var_small_partial_kernel<TQ>, dim3(npairs, n_out), dim3(256), 0, s, Wt, Ks, Vp, Np, \
                       Tp, npairs, k_lo)
    if (T <= 1) SR_SMALL_LAUNCH(1);
    else if (T <= 4) SR_SMALL_LAUNCH(4);
    else
        hipLaunchKernelGGL(sr_var_small_partial_mfma_kernel, dim3(npairs, n_out, groups), dim3(1024), 0, s, Wt, Ks, Vp,
                           Np, Tp, npairs, k_lo);
#undef SR_SMALL_LAUNCH
    SR_HIP(hipGetLastError());
    if (!reduce) return SR_OK;                     // the caller gathers the partial products itself (sr_gp_append)
    const int tq = small_tq(T);
    hipLaunchKernelGGL(sr_var_small_reduce_kernel, dim3(ncb, n_out, tq * groups), dim3(256), 0, s, Vp, part, Tp,
                       npairs, ncb, tq, dot0);
    SR_HIP(hipGetLastError());
    return SR_OK;
}

// ------------------------------------------------------------------------------------------------
// K3: reduce partials, clip, write API layout (T x n_out [x D]).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sr_finalize_kernel(sr_final_args a) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const int d = blockIdx.y;
    if (t >= a.T) return;
    double m = 0.0;
    for (int s = 0; s < a.nsplit; ++s) m += a.mu_part[((long)s * a.n_out + d) * a.Tp + t];
    double q = 0.0;
    for (int rb = 0; rb < a.nrb; ++rb) q += a.var_part[((long)d * a.nrb + rb) * a.Tp + t];
    double v = (a.kxx ? a.kxx[(long)d * a.Tp + t] : a.sf2[d]) - q;
    if (!(v > SR_VAR_CLIP)) v = SR_VAR_CLIP;
    a.mu[t * a.n_out + d] = m;
    a.var[t * a.n_out + d] = v;
    if (a.jac) {
        for (int j = 0; j < a.D; ++j) {
            double gj = 0.0;
            for (int s = 0; s < a.nsplit; ++s)
                gj += a.jac_part[(((long)s * a.n_out + d) * a.D + j) * a.Tp + t];
            a.jac[(t * a.n_out + d) * a.D + j] = gj;
        }
    }
}

// Few queries: one wavefront per (query, output), lanes stride over the partial sums and combine
// with a butterfly -- the serial chain of the thread-per-query form (nsplit * (1 + D) + nrb dependent
// loads) dominates the latency of the small-batch path otherwise.  (sr_final_dev.h)
__global__ __launch_bounds__(256) void sr_finalize_wave_kernel(sr_final_args a) {
    const long t = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= a.T) return;
    sr_final_query_wave(a, t, blockIdx.y, threadIdx.x & 63);
}

int sr_launch_finalize(const sr_final_args& a, hipStream_t s) {
    if (a.T <= SR_FINAL_WAVE_T) {
        dim3 grid((unsigned)((a.T + 3) / 4), a.n_out);
        hipLaunchKernelGGL(sr_finalize_wave_kernel, grid, dim3(256), 0, s, a);
        SR_HIP(hipGetLastError());
        return SR_OK;
    }
    dim3 grid((unsigned)((a.T + 255) / 256), a.n_out);
    hipLaunchKernelGGL(sr_finalize_kernel, grid, dim3(256), 0, s, a);
    SR_HIP(hipGetLastError());
    return SR_OK;
}
