// sr_small_dev.h -- device code shared by the one-launch posterior of small models (K0, sr_small.hip), the resident
// single-query server (K0s, sr_server.hip) and the persistent multi-step kernel (K0c, sr_chain.hip): phase A (k* in
// A-fragment layout, mean and mean-Jacobian on the MFMA tile), phase B (contraction with 16-column strips of U^-1),
// the LDS layout of a workgroup and the output stage.  See sr_small.hip for the algorithm.
#pragma once
#include "sr_final_dev.h"
#include "sr_ellipsoid_dev.h"

#define SR_FQ 16         // queries per workgroup == N of the MFMA tile
typedef double sr_d4 __attribute__((ext_vector_type(4)));

// Phases B and C, shared by the kernels below: V = U^-T [columns of ks] on the MFMA tile, then per column c
// redC[strip][c] = sum_{rows of the strip} V[i][c] * (DOT0 ? V[i][0] : V[i][c]).  Ends with a barrier.
template <int NP, bool DOT0>
__device__ __forceinline__ void sr_small_contract(const double* __restrict__ Wd, const double (*ks)[SR_FQ],
                                                  double* pB, double (*redC)[SR_FQ], int wave, int lane) {
    constexpr int NSTRIP = NP / 16, NPAIR = NSTRIP / 2;
    constexpr int NSPLIT = (16 / NPAIR) > 0 ? 16 / NPAIR : 1;
    const int lk = lane >> 4, ln = lane & 15;
    const int pr = wave / NSPLIT, h = wave % NSPLIT;      // part h of strips pr and NSTRIP-1-pr
    sr_d4 accB[2];
    accB[0] = sr_d4{0.0, 0.0, 0.0, 0.0};
    accB[1] = sr_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int which = 0; which < 2; ++which) {
        if (pr >= NPAIR) break;                                   // Np = 384: 12 pairs, 4 wavefronts idle here
        const int sidx = which ? NSTRIP - 1 - pr : pr;
        const int chunk = 4 * (sidx + 1) / NSPLIT;               // k-steps (of 4 rows) of this part
        int st = h * chunk;
        const int st_end = st + chunk;
        const double* wcol = Wd + (long)lk * NP + 16 * sidx + ln;
        sr_d4 acc = {0.0, 0.0, 0.0, 0.0};
        for (; st + 16 <= st_end; st += 16) {
            double af[16], bf[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) af[u] = wcol[(long)(4 * (st + u)) * NP];
#pragma unroll
            for (int u = 0; u < 16; ++u) bf[u] = ks[4 * (st + u) + lk][ln];
#pragma unroll
            for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af[u], bf[u], acc, 0, 0, 0);
        }
        for (; st + 4 <= st_end; st += 4) {
            double af[4], bf[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) af[u] = wcol[(long)(4 * (st + u)) * NP];
#pragma unroll
            for (int u = 0; u < 4; ++u) bf[u] = ks[4 * (st + u) + lk][ln];
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af[u], bf[u], acc, 0, 0, 0);
        }
        for (; st < st_end; ++st) {
            const double af = wcol[(long)(4 * st) * NP];
            const double bf = ks[4 * st + lk][ln];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af, bf, acc, 0, 0, 0);
        }
        accB[which] = acc;
        if (NSPLIT > 1 && h > 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) pB[((h - 1) * NSTRIP + sidx) * 256 + r * 64 + lane] = acc[r];
        }
    }
    __syncthreads();
    if (h == 0 && pr < NPAIR) {
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            const int sidx = which ? NSTRIP - 1 - pr : pr;
            double q = 0.0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double v = accB[which][r];
#pragma unroll
                for (int hh = 0; hh < NSPLIT - 1; ++hh) v += pB[(hh * NSTRIP + sidx) * 256 + r * 64 + lane];
                const double w = DOT0 ? __shfl(v, lane & 48) : v;    // DOT0: dot with column 0 of the same row
                q = fma(v, w, q);
            }
            q += __shfl_xor(q, 16);
            q += __shfl_xor(q, 32);
            if (lane < 16) redC[sidx][lane] = q;
        }
    }
    __syncthreads();
}

// LIN: single query with second-order outputs (sr_gp_linearize).  The 16 MFMA columns then carry
// [k*, dk*/dx_1 .. dk*/dx_D] instead of 16 queries (dk*_i/dx_j = k*_i (z_ij - x_j)/l_j^2):
//   R[c][:] = sum_i col_c[i] M[i][:]   ->  mu = R[0][0],  d mu/dx_j = R[1+j][0],
//                                          d2 mu/dx_j dx_l = (R[1+j][1+l] - x_l/l_l R[1+j][0])/l_l - delta_jl mu/l_j^2
//   V_c = U^-T col_c                   ->  var = sf2 - V_0.V_0,  d var/dx_j = -2 V_j.V_0
// -- the same two phases, no second pass over U^-1.
// LDS of one posterior evaluation (arrays live in the calling kernel)
template <int NP, int DT>
struct sr_small_lds {
    double (*ks)[SR_FQ];        // [NP]      k*[k][t]
    double (*xq)[DT];           // [SR_FQ]   queries of this tile, scaled by 1/lengthscale
    double (*pA)[256];          // [16]      phase A: per-wavefront partial R (accumulator layout)
    double (*Rs)[16];           // [SR_FQ]   R[t][c]
    double* pB;                 // phase B: partial V tiles, parts h > 0
    double (*redC)[SR_FQ];      // [NP/16]   per strip and column: sum of squares (or dots with column 0)
};
#define SR_SMALL_LDS_DECL(NP, DT)                                                                          \
    __shared__ double ks_[NP][SR_FQ];                                                                      \
    __shared__ double xq_[SR_FQ][DT];                                                                      \
    __shared__ double pA_[16][256];                                                                        \
    __shared__ double Rs_[SR_FQ][16];                                                                      \
    __shared__ double pB_[((16 / (NP / 32)) > 1 ? (16 / (NP / 32)) - 1 : 1) * ((16 / (NP / 32)) > 1 ? NP / 16 : 1) * 256]; \
    __shared__ double redC_[NP / 16][SR_FQ];                                                               \
    sr_small_lds<NP, DT> L{ks_, xq_, pA_, Rs_, pB_, redC_}

// Phases A - C for output d and the (up to) SR_FQ queries x_t = [xa[t*lda ..], xb[t*ldb ..]], t < nq, whose pointers
// may be global or LDS.  Leaves R in L.Rs, the scaled queries in L.xq and the strip sums in L.redC; ends with a
// barrier.  Must be called by all 1024 threads.
// The training rows of a lane's phase-A fragments: phase A loads them itself, HC steps at a time, unless KEEP:
// The persistent chain kernel keeps them in LDS for all its steps (sr_small_rows: Np x (DT + 1) doubles, z_ij / l_j and
// alpha_i; round 2 kept them in registers where they fit -- 64 to 96 VGPRs beside the U^-1 fragments, the main reason
// for the kernel's scratch use -- and re-read them from L2 every step where they did not).
template <int NP, int DT>
struct sr_small_rows {
    const double (*r)[DT + 1];        // [NP]: z_i0 / l_0 .. , alpha_i (0 on padding rows)
    const double* il;                 // [DT]: 1 / l_j (0 beyond D) -- read per step, not held in registers across steps
};

// all threads of the workgroup; the caller's barrier publishes the rows
template <int NP, int DT>
__device__ __forceinline__ void sr_small_rows_fill(const sr_kstar_args& a, int d, double (*dst)[DT + 1], int nthreads) {
    const int off = NP - a.N;
    for (int e = threadIdx.x; e < NP * (DT + 1); e += nthreads) {
        const int i = e / (DT + 1), j = e % (DT + 1);
        const bool valid = i >= off;
        double v = 0.0;
        if (valid && j == DT) v = a.alpha[(long)d * NP + i];
        else if (valid && j < a.D) v = a.Z[(long)(i - off) * a.D + j] * (1.0 / a.ls[d * a.D + j]);   // as phase A forms it
        dst[i][j] = v;
    }
}
template <int DT>
__device__ __forceinline__ void sr_small_il_fill(const sr_kstar_args& a, int d, double* il) {
    if (threadIdx.x < DT) il[threadIdx.x] = ((int)threadIdx.x < a.D) ? 1.0 / a.ls[d * a.D + threadIdx.x] : 0.0;
}

// Phase A alone: k* into L.ks, R = k*^T M into L.Rs (valid for threads < 256 right away, for everybody after the next
// barrier), the scaled queries into L.xq.
template <int NP, int DT, bool LIN, bool KEEP = false, int NW = 16>
__device__ __forceinline__ void sr_small_phase_a(const sr_kstar_args& a, int d,
                                                 const double* xa, long lda, const double* xb, long ldb, long nq,
                                                 const sr_small_lds<NP, DT>& L,
                                                 const sr_small_rows<NP, DT>* rows = nullptr, int tid_in = -1) {
    constexpr int RPW = NP / NW;             // training rows per wavefront in phase A
    constexpr int KSA = RPW / 4;             // phase-A k-steps per wavefront
#ifndef SR_CHAIN_HC
#define SR_CHAIN_HC 2
#endif
    // k-steps whose global loads are hoisted together.  The persistent kernel (NW = 8) holds its U^-1 fragments in
    // registers throughout: there at most SR_CHAIN_HC steps (HC (DT + 1) doubles per lane) -- with half of the 16 steps
    // of Np = 512 hoisted the kernel needed 184 - 520 B of scratch per lane
    constexpr int HC0 = KSA <= 4 ? KSA : KSA / 2;
    // (query width 8 in the one-launch kernels: 1024 threads = 128 registers per lane; four hoisted steps of 9 doubles each
    //  beside xs / il made the LIN instantiations spill: 20 .. 116 B per lane)
    constexpr int HC1 = (NW == 16 && DT >= 8 && HC0 > 2 && KSA % 2 == 0) ? 2 : HC0;
    // (the resident server -- KEEP with 16 wavefronts -- reads its rows from LDS: one step at a time costs nothing there
    //  and keeps the D = 5 instantiations inside 128 registers)
    constexpr int HC = (KEEP && NW == 16) ? 1
                     : ((NW == 16 || HC1 <= SR_CHAIN_HC) ? HC1 : ((KSA % SR_CHAIN_HC == 0) ? SR_CHAIN_HC : (KSA % 3 == 0 ? 3 : 2)));
    static_assert(DT + 1 <= 16, "the mean/Jacobian right-hand side must fit the 16 MFMA columns");
    static_assert(NP % 128 == 0 && NP <= 512 && KSA % HC == 0 && KSA >= 1, "Np in {128, 256, 384, 512}");
    double (*ks)[SR_FQ] = L.ks;
    double (*xq)[DT] = L.xq;
    double (*pA)[256] = L.pA;
    double (*Rs)[16] = L.Rs;

    const int tid = tid_in >= 0 ? tid_in : (int)threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lk = lane >> 4, ln = lane & 15;             // fragment coordinates: k offset, m/n index
    const int off = NP - a.N;                             // front padding

    const double sf2 = a.sf2[d];
    const int qt = LIN ? 0 : ln;                          // query index of this lane's column
    const bool live = LIN ? true : (ln < nq);

    // ---- phase A ------------------------------------------------------------------------------
    {
        // global loads of HC k-steps first (one round trip), then the arithmetic
        double xs[DT], il[DT];
#pragma unroll
        for (int j = 0; j < DT; ++j) {
            il[j] = (!KEEP && j < a.D) ? a.ls[d * a.D + j] : 1.0;
            xs[j] = 0.0;
            if (live && j < a.D) xs[j] = a.xv_on ? a.xv[j] : ((j < a.na) ? xa[qt * lda + j] : xb[qt * ldb + (j - a.na)]);
        }
#pragma unroll
        for (int j = 0; j < DT; ++j) {
            il[j] = KEEP ? rows->il[j] : ((j < a.D) ? 1.0 / il[j] : 0.0);
            xs[j] *= il[j];
        }
        sr_d4 accA = {0.0, 0.0, 0.0, 0.0};
        if constexpr (LIN) {
            // ONE query: the 16 lanes of a fragment row share the training row, so the row's k* is evaluated once -- lane r
            // of the wavefront owns row r of its RPW rows (round 1) -- and handed to the fragment layout through LDS (round 2:
            // each lane then needs one coordinate of its row, jc = ln - 1).  In the fragment layout every lane repeated the
            // exp: KSA serial evaluations per wavefront, times the wavefronts of a SIMD.  Same arithmetic, same bits.
            static_assert(RPW <= 64, "one lane per training row of the wavefront");
            double k1, al1;
            {
                const int i1 = wave * RPW + (lane < RPW ? lane : RPW - 1);
                const bool valid = i1 >= off;
                double r2 = 0.0;
                al1 = KEEP ? rows->r[i1][DT] : (valid ? a.alpha[(long)d * NP + i1] : 0.0);
#pragma unroll
                for (int j = 0; j < DT; ++j) {
                    const double zs = KEEP ? rows->r[i1][j] : (((valid && j < a.D) ? a.Z[(long)(i1 - off) * a.D + j] : 0.0) * il[j]);
                    const double df = xs[j] - zs;
                    r2 = fma(df, df, r2);
                }
                k1 = valid ? sf2 * exp(-0.5 * r2) : 0.0;
                // to the 16 lanes of the row's fragment row through LDS: columns 14, 15 of its own K* row (1 + D <= 9 columns
                // carry numbers); a ds_bpermute_b32 per dword and step cost ~64 cycles each
                if (lane < RPW) { ks[i1][14] = k1; ks[i1][15] = al1; }
            }
            const int jc = (ln >= 1 && ln <= DT) ? ln - 1 : -1;
            double xc = 0.0, ilc = 0.0;
#pragma unroll
            for (int j = 0; j < DT; ++j)
                if (jc == j) { xc = xs[j]; ilc = il[j]; }
#pragma unroll
            for (int st = 0; st < KSA; ++st) {
                const int i = wave * RPW + 4 * st + lk;
                double zc = 0.0;
                if (jc >= 0) {
                    if (KEEP) zc = rows->r[i][jc];
                    else if (i >= off && jc < a.D) zc = a.Z[(long)(i - off) * a.D + jc] * ilc;
                }
                const double ki = ks[i][14], ali = ks[i][15];
                const double df = xc - zc;
                const double bfrag = (ln == 0) ? ali : ((jc >= 0) ? ali * zc : 0.0);
                const double scale = (ln == 0) ? 1.0 : ((jc >= 0) ? -df * ilc : 0.0);      // (z_j - x_j) / l_j^2
                const double k = ki * scale;                   // column c: k* (c = 0), dk*/dx_{c-1}, 0 beyond D
                if (ln < 14) ks[i][ln] = k;
                accA = __builtin_amdgcn_mfma_f64_16x16x4f64(k, bfrag, accA, 0, 0, 0);
            }
        } else
#pragma unroll
        for (int c0 = 0; c0 < KSA; c0 += HC) {
            double zv[HC][DT], al[HC];
#pragma unroll
            for (int st = 0; st < HC; ++st) {
                const int i = wave * RPW + 4 * (c0 + st) + lk;
                const bool valid = i >= off;
                if (KEEP) {
                    al[st] = rows->r[i][DT];
#pragma unroll
                    for (int j = 0; j < DT; ++j) zv[st][j] = rows->r[i][j];
                } else {
                    al[st] = valid ? a.alpha[(long)d * NP + i] : 0.0;
#pragma unroll
                    for (int j = 0; j < DT; ++j) zv[st][j] = (valid && j < a.D) ? a.Z[(long)(i - off) * a.D + j] : 0.0;
                }
            }
#pragma unroll
            for (int st = 0; st < HC; ++st) {
                const int i = wave * RPW + 4 * (c0 + st) + lk;
                double r2 = 0.0, bfrag = (ln == 0) ? al[st] : 0.0, scale = (ln == 0) ? 1.0 : 0.0;
#pragma unroll
                for (int j = 0; j < DT; ++j) {
                    const double zs = KEEP ? zv[st][j] : zv[st][j] * il[j];
                    const double df = xs[j] - zs;
                    r2 = fma(df, df, r2);
                    if (ln == j + 1) {
                        bfrag = al[st] * zs;
                        scale = -df * il[j];                    // (z_j - x_j) / l_j^2
                    }
                }
                double k = (i >= off && live) ? sf2 * exp(-0.5 * r2) : 0.0;
                if (LIN) k *= scale;                            // column c: k* (c = 0), dk*/dx_{c-1}, 0 beyond D
                ks[i][ln] = k;
                accA = __builtin_amdgcn_mfma_f64_16x16x4f64(k, bfrag, accA, 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) pA[wave][r * 64 + lane] = accA[r];
        if (wave == 0 && lk == 0) {
#pragma unroll
            for (int j = 0; j < DT; ++j) xq[ln][j] = xs[j];
        }
    }
    __syncthreads();
    if (tid < 256) {
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) v += pA[w][tid];
        const int l2 = tid & 63, r = tid >> 6;
        Rs[(l2 >> 4) + 4 * r][l2 & 15] = v;
    }
}

template <int NP, int DT, bool LIN, bool KEEP = false>
__device__ __forceinline__ void sr_small_posterior(const sr_kstar_args& a, const double* __restrict__ Wt, int d,
                                                   const double* xa, long lda, const double* xb, long ldb, long nq,
                                                   const sr_small_lds<NP, DT>& L,
                                                   const sr_small_rows<NP, DT>* rows = nullptr, int tid_in = -1) {
    // (tid_in: the thread index as the resident server hands it in -- through an empty asm every round, so that nothing
    //  derived from it counts as loop-invariant there)
    const int tid = tid_in >= 0 ? tid_in : (int)threadIdx.x;
    sr_small_phase_a<NP, DT, LIN, KEEP, 16>(a, d, xa, lda, xb, ldb, nq, L, rows, tid_in);
    // ---- phases B, C ---------------------------------------------------------------------------
    sr_small_contract<NP, LIN>(Wt + (long)d * NP * NP, L.ks, L.pB, L.redC, tid >> 6, tid & 63);
}

// Outputs of one posterior evaluation (after sr_small_posterior), straight to the API layout; all threads call.
template <int NP, int DT, bool LIN>
__device__ __forceinline__ void sr_small_outputs(const sr_kstar_args& a, const sr_small_lds<NP, DT>& L, int d, long t0, double sf2,
                                                 double* __restrict__ mu, double* __restrict__ var, double* __restrict__ jac,
                                                 double* __restrict__ jac_var, double* __restrict__ hess, int tid_in = -1) {
    constexpr int NSTRIP = NP / 16;          // 16-column strips of U^-1
    double (*xq)[DT] = L.xq;
    double (*Rs)[16] = L.Rs;
    double (*redC)[SR_FQ] = L.redC;
    const int tid = tid_in >= 0 ? tid_in : (int)threadIdx.x;
    if (LIN) {
        const double m = Rs[0][0];
        if (tid == 0) mu[d] = m;
        if (tid < a.D) jac[d * a.D + tid] = Rs[1 + tid][0];
        if (tid < a.D * a.D) {
            const int j = min(tid / a.D, tid % a.D), l = max(tid / a.D, tid % a.D);
            const double ilj = 1.0 / a.ls[d * a.D + j], ill = 1.0 / a.ls[d * a.D + l];
            double hv = (Rs[1 + j][1 + l] - xq[0][l] * Rs[1 + j][0]) * ill;
            if (j == l) hv -= m * ilj * ilj;
            hess[(long)d * a.D * a.D + tid] = hv;
        }
    } else if (tid < SR_FQ * (DT + 1)) {
        const int t = tid / (DT + 1), j = tid % (DT + 1);
        if (t0 + t < a.T) {
            const double m = Rs[t][0];
            if (j == DT) mu[(t0 + t) * a.n_out + d] = m;
            else if (jac && j < a.D)
                jac[((t0 + t) * a.n_out + d) * a.D + j] = (Rs[t][1 + j] - xq[t][j] * m) / a.ls[d * a.D + j];
        }
    }

    if (LIN) {
        if (tid <= a.D) {
            double qn = 0.0;
#pragma unroll
            for (int sidx = 0; sidx < NSTRIP; ++sidx) qn += redC[sidx][tid];
            if (tid == 0) {
                double v = sf2 - qn;
                if (!(v > SR_VAR_CLIP)) v = SR_VAR_CLIP;
                var[d] = v;
            } else {
                jac_var[d * a.D + tid - 1] = -2.0 * qn;
            }
        }
    } else if (tid < SR_FQ && t0 + tid < a.T) {
        double qn = 0.0;
#pragma unroll
        for (int sidx = 0; sidx < NSTRIP; ++sidx) qn += redC[sidx][tid];
        double v = sf2 - qn;
        if (!(v > SR_VAR_CLIP)) v = SR_VAR_CLIP;
        var[(t0 + tid) * a.n_out + d] = v;
    }

}

// ------------------------------------------------------------------------------------------------------------------
// GENERAL kernel family (sr_common.h: k = (c0 + sum a x z) v kappa(r) + sum b x z, kappa in {RBF, Matern-5/2} -- the
// reference's mat52 / lin_rbf / lin_mat52, ssm_gpy/gp_models_utils_casadi.py:43-157, which its journal experiments run:
// experiments/journal_experiment_configs/defaultconfig_episode.py:39), ONE query with second-order outputs, on the same
// MFMA tile as the ARD-RBF form above.  With u_j = s_j^2 (x_j - z_j), c = c0 + sum a x z, g = kappa'/r, h = g'/r
// (sr_linearize.hip):
//   dk/dx_j        = a_j z_j v kappa + c v g u_j + b_j z_j
//   d2k/dx_j dx_l  = v g (a_j z_j u_l + a_l z_l u_j) + c v (h u_j u_l + g s_j^2 delta_jl)
// Two products against the SAME right-hand side M[i][:] = alpha_i [1, z_i - x] (16 columns, 1 + D used):
//   P1 rows m:  0: k    1+j: dk/dx_j    8: c v g    9+j: c v h u_j          P2 rows m:  1+l: v g u_l
//   mu = P1[0][0],  d mu/dx_j = P1[1+j][0],
//   d2 mu/dx_j dx_l = a_j (P2[1+l][1+j] + x_j P2[1+l][0]) + a_l (P2[1+j][1+l] + x_l P2[1+j][0]) - s_l^2 P1[9+j][1+l]
//                     + delta_jl s_j^2 P1[8][0]
// and rows 0 .. D of P1's left operand are the columns [k*, dk*/dx] phase B contracts with U^-1:
//   var = k(x,x) - V_0.V_0,  d var/dx_j = 2 (a_j v + b_j) x_j - 2 V_j.V_0,   k(x,x) = (c0 + sum a x^2) v + sum b x^2.
// D <= 5 (rows 9 + j <= 15 would allow 7; the D = 8 instantiations of the sixteen-wavefront kernels spill).
// ------------------------------------------------------------------------------------------------------------------
template <int NP, int DT>
struct sr_gen_lds {
    sr_small_lds<NP, DT> s;     // ks, xq (row 0: the query, UNSCALED), pA, Rs (P1), pB, redC
    double (*pA2)[256];         // per-wavefront partial P2
    double (*Rs2)[16];          // P2
};
// packed parameters of one output (SR_KP(D) doubles: kappa id, v, c0, s[D], a[D], b[D]) -> what the evaluation uses
template <int DT>
struct sr_gen_par {
    int kind; double v, c0, s2[DT], a[DT], b[DT];
    __device__ __forceinline__ void load(const double* kp, int D) {
        kind = (int)kp[0]; v = kp[1]; c0 = kp[2];
#pragma unroll
        for (int j = 0; j < DT; ++j) {
            const double sj = (j < D) ? kp[3 + j] : 0.0;
            s2[j] = sj * sj;
            a[j] = (j < D) ? kp[3 + D + j] : 0.0;
            b[j] = (j < D) ? kp[3 + 2 * D + j] : 0.0;
        }
    }
};
// rows for KEEP: z_ij UNSCALED, alpha_i (0 on padding rows)
template <int NP, int DT>
__device__ __forceinline__ void sr_small_rows_fill_raw(const sr_kstar_args& a, int d, double (*dst)[DT + 1], int nthreads) {
    const int off = NP - a.N;
    for (int e = threadIdx.x; e < NP * (DT + 1); e += nthreads) {
        const int i = e / (DT + 1), j = e % (DT + 1);
        const bool valid = i >= off;
        double v = 0.0;
        if (valid && j == DT) v = a.alpha[(long)d * NP + i];
        else if (valid && j < a.D) v = a.Z[(long)(i - off) * a.D + j];
        dst[i][j] = v;
    }
}

// Phase A of the general family for ONE query x (D doubles at xsrc: global, LDS or the kernel arguments a.xv): the columns
// [k*, dk*/dx] into L.s.ks, P1 into L.s.Rs and P2 into L.Rs2 (both valid after the NEXT barrier), the query into L.s.xq[0].
// kp: the output's packed parameters (global or LDS).  All NW wavefronts call.
template <int NP, int DT, bool KEEP, int NW>
__device__ __forceinline__ void sr_small_phase_a_gen(const sr_kstar_args& a, int d, const double* xsrc, const double* kp,
                                                     const sr_gen_lds<NP, DT>& L,
                                                     const sr_small_rows<NP, DT>* rows = nullptr, int tid_in = -1) {
    constexpr int RPW = NP / NW, KSA = RPW / 4;
    static_assert(DT <= 5, "rows 9 + j of P1 and the registers of the sixteen-wavefront kernels: D <= 5");
    double (*ks)[SR_FQ] = L.s.ks;
    const int tid = tid_in >= 0 ? tid_in : (int)threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lk = lane >> 4, ln = lane & 15;
    const int off = NP - a.N;
    sr_gen_par<DT> P;
    P.load(kp, a.D);
    double x[DT];
#pragma unroll
    for (int j = 0; j < DT; ++j) x[j] = (j < a.D) ? (a.xv_on ? a.xv[j] : xsrc[j]) : 0.0;
    static_assert(RPW <= 64, "one lane per training row of the wavefront");
    // Round 1 -- lane r of the wavefront owns training row r of its RPW rows and evaluates the row's radial terms ONCE (in
    // the fragment layout of round 2 the 16 lanes of a row would each repeat the exp and the square root: with eight
    // wavefronts on four SIMDs that was 16 serial evaluations per SIMD, 4.8 of the 10.6 us of a 256-row model).
    double k0, vk, pg, ph, vg, al1;
    {
        const int i1 = wave * RPW + (lane < RPW ? lane : RPW - 1);
        const bool valid = i1 >= off;
        double z[DT];
        if (KEEP) {
            al1 = rows->r[i1][DT];
#pragma unroll
            for (int j = 0; j < DT; ++j) z[j] = rows->r[i1][j];
        } else {
            al1 = valid ? a.alpha[(long)d * NP + i1] : 0.0;
#pragma unroll
            for (int j = 0; j < DT; ++j) z[j] = (valid && j < a.D) ? a.Z[(long)(i1 - off) * a.D + j] : 0.0;
        }
        double r2 = 0.0, la = 0.0, lb = 0.0;
#pragma unroll
        for (int j = 0; j < DT; ++j) {
            const double df = x[j] - z[j];
            r2 = fma(P.s2[j] * df, df, r2);
            la = fma(P.a[j] * x[j], z[j], la);
            lb = fma(P.b[j] * x[j], z[j], lb);
        }
        double kap, g, hh;
        if (P.kind == 0) {
            kap = exp(-0.5 * r2);
            g = -kap;
            hh = kap;
        } else {
            const double rr = sqrt(r2);
            const double e = exp(-2.23606797749978969641 * rr);
            kap = (1.0 + 2.23606797749978969641 * rr + (5.0 / 3.0) * r2) * e;
            g = -(5.0 / 3.0) * (1.0 + 2.23606797749978969641 * rr) * e;
            hh = (25.0 / 3.0) * e;
        }
        const double pre = (P.c0 + la) * P.v;
        k0 = valid ? fma(pre, kap, lb) : 0.0;
        vk = valid ? P.v * kap : 0.0;
        pg = valid ? pre * g : 0.0;
        ph = valid ? pre * hh : 0.0;
        vg = valid ? P.v * g : 0.0;
    }
    // The row's terms go to the 16 lanes of its fragment row through LDS: columns 10 .. 15 of the row's own K* row (columns
    // beyond 1 + D <= 6 of ks carry nothing: phase B multiplies them along and nobody reads the result).  A shuffle per term
    // and step was the first form: ds_bpermute_b32 costs ~64 cycles on this part, twelve of them per step 900 cycles --
    // 3.0 of the 7.5 us a 256-row model took on the device (timestamps inside the kernel); a broadcast ds_read_b64 costs 2.
    if (lane < RPW) {
        double* rs = &ks[wave * RPW + lane][10];
        rs[0] = k0; rs[1] = vk; rs[2] = pg; rs[3] = ph; rs[4] = vg; rs[5] = al1;
    }
    // Round 2 -- fragment layout: this lane's row m = ln of the two left operands and its column n = ln of the right-hand
    // side need ONE coordinate of the training row: jc = ln - 1 (rows 1 + j, column 1 + j) or ln - 9 (rows 9 + j).
    // What a lane forms from the row's terms is written with per-lane 0 / 1 factors instead of branches on its role, and the
    // terms of step st + 1 are requested before the arithmetic of step st:
    //   A1 = c0 k0 + c8 pg + zc (vk a_c + b_c) + (lo pg + hi ph) u_c ,   A2 = lo vg u_c ,   B = alpha (c0 + lo (zc - x_c))
    const int jc = (ln >= 1 && ln <= a.D) ? ln - 1 : ((ln >= 9 && ln < 9 + a.D) ? ln - 9 : -1);
    const bool low = ln >= 1 && ln <= a.D, high = ln >= 9 && ln < 9 + a.D;
    double xc = 0.0, s2c = 0.0, ac = 0.0, bc = 0.0;
#pragma unroll
    for (int j = 0; j < DT; ++j)
        if (jc == j) { xc = x[j]; s2c = P.s2[j]; if (low) { ac = P.a[j]; bc = P.b[j]; } }
    const double c0 = (ln == 0) ? 1.0 : 0.0, c8 = (ln == 8) ? 1.0 : 0.0, lo = low ? 1.0 : 0.0, hi = high ? 1.0 : 0.0;
    const double ksm = (ln <= a.D) ? 1.0 : 0.0;
    const double cx = s2c * xc;
    struct RowTerms { double k0, vk, pg, ph, vg, al, zc; };
    auto fetch = [&](int st) -> RowTerms {
        const int i = wave * RPW + 4 * st + lk;
        RowTerms t;
        t.zc = 0.0;
        if (jc >= 0) {
            if (KEEP) t.zc = rows->r[i][jc];
            else if (i >= off) t.zc = a.Z[(long)(i - off) * a.D + jc];
        }
        const double* rs = &ks[i][10];
        t.k0 = rs[0]; t.vk = rs[1]; t.pg = rs[2]; t.ph = rs[3]; t.vg = rs[4]; t.al = rs[5];
        return t;
    };
    sr_d4 acc1 = {0.0, 0.0, 0.0, 0.0}, acc2 = acc1;
    RowTerms nx = fetch(0);
#pragma unroll
    for (int st = 0; st < KSA; ++st) {
        const int i = wave * RPW + 4 * st + lk;
        const RowTerms t = nx;
        if (st + 1 < KSA) nx = fetch(st + 1);
        const double uc = fma(-s2c, t.zc, cx);                               // s_c^2 (x_c - z_c)
        const double A1 = fma(c0, t.k0, fma(c8, t.pg, fma(t.zc, fma(t.vk, ac, bc), fma(lo, t.pg, hi * t.ph) * uc)));
        const double A2 = lo * t.vg * uc;
        const double bfrag = t.al * fma(lo, t.zc - xc, c0);
        if (ln < 10) ks[i][ln] = ksm * A1;                                  // (columns 10 .. 15: the row's terms stay)
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(A1, bfrag, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(A2, bfrag, acc2, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        L.s.pA[wave][r * 64 + lane] = acc1[r];
        L.pA2[wave][r * 64 + lane] = acc2[r];
    }
    if (tid == 0) {
#pragma unroll
        for (int j = 0; j < DT; ++j) L.s.xq[0][j] = x[j];
    }
    __syncthreads();
    if (tid < 512) {
        const int t2 = tid & 255;
        double v = 0.0;
        if (tid < 256) {
#pragma unroll
            for (int w = 0; w < NW; ++w) v += L.s.pA[w][t2];
        } else {
#pragma unroll
            for (int w = 0; w < NW; ++w) v += L.pA2[w][t2];
        }
        const int l2 = t2 & 63, r = t2 >> 6;
        (tid < 256 ? L.s.Rs : L.Rs2)[(l2 >> 4) + 4 * r][l2 & 15] = v;
    }
}

// Element e of the record [mu, var (here: the prior variance k(x,x); the caller subtracts V_0.V_0), d mu/dx (D),
// d k(x,x)/dx (D; the caller subtracts 2 V_j.V_0), d2 mu/dx2 (D x D)] of the general family after sr_small_phase_a_gen
// (+ a barrier).  kp as there.
template <int NP, int DT>
__device__ __forceinline__ double sr_gen_record_elem(int e, int D, const sr_gen_lds<NP, DT>& L, const double* kp) {
    const double (*R1)[16] = L.s.Rs;
    const double (*R2)[16] = L.Rs2;
    const double* x = L.s.xq[0];
    const double v = kp[1], c0 = kp[2];
    if (e == 0) return R1[0][0];
    if (e == 1) {
        double kxx = c0 * v;
        for (int j = 0; j < D; ++j) kxx = fma((kp[3 + D + j] * v + kp[3 + 2 * D + j]) * x[j], x[j], kxx);
        return kxx;
    }
    if (e < 2 + D) return R1[1 + (e - 2)][0];
    if (e < 2 + 2 * D) {
        const int j = e - (2 + D);
        return 2.0 * (kp[3 + D + j] * v + kp[3 + 2 * D + j]) * x[j];
    }
    if (e < 2 + 2 * D + D * D) {
        const int q = e - (2 + 2 * D);
        const int j = min(q / D, q % D), l = max(q / D, q % D);       // (j <= l: the matrix comes out exactly symmetric)
        const double aj = kp[3 + D + j], al = kp[3 + D + l], sl = kp[3 + l];
        const double s1jl = fma(x[j], R2[1 + l][0], R2[1 + l][1 + j]);
        const double s1lj = fma(x[l], R2[1 + j][0], R2[1 + j][1 + l]);
        double hv = aj * s1jl + al * s1lj - sl * sl * R1[9 + j][1 + l];
        if (j == l) hv = fma(kp[3 + j] * kp[3 + j], R1[8][0], hv);
        return hv;
    }
    return 0.0;
}
