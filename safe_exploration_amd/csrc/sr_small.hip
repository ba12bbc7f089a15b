// sr_small.hip -- K0: the whole GP posterior of a SMALL model in one launch.
//
// Regime: the reference's own experiments (25 .. 150 inducing points; Np <= 512 here)
// driven with one query (CasADi/IPOPT callback) up to ~1000 candidate states per step.  There the
// three-kernel pass K1 -> K2m -> K3 is three dependent launches of tiny grids, each bound by its own chain
// of global-load round trips (6 + 13 + 5 us at N = 200); nothing is bound by flops or bytes.
// One workgroup of 16 wavefronts evaluates SR_FQ = 16 queries of ONE output, everything on the fp64 MFMA
// 16x16x4 tile so that no operand is ever broadcast through LDS:
//
//   phase A  wavefront w owns Np/16 training rows: each lane evaluates k*[i][t] for its (i, t) of the
//            MFMA A-fragment, stores it to LDS and multiplies it on the spot with the B-fragment
//            M[i][:] = alpha_i [1, z_i/l]:  R[t][:] = sum_i k*[i][t] M[i][:]  gives
//            mu_t = R[t][0]  and  d mu_t/dx_j = (R[t][1+j] - x_tj/l_j R[t][0]) / l_j
//   phase B  V[i][t] = sum_{k<=i} Wt[k][i] k*[k][t]: 16-column strips of U^-1, strips s and Np/16-1-s paired
//            and their k range cut into 32/(Np/16) parts so that all 16 wavefronts carry the same number of
//            MFMAs (34 at Np = 256); A-fragments straight from L2, up to 16 loads in flight per lane
//   phase C  parts added in a fixed order, var_t = max(sf2 - sum_i V[i][t]^2, 1e-15)
//
// Outputs go straight to the API layout: no partial buffers in HBM, no finalize launch.
// formulas: /root/reference/safe_exploration/ssm_gpy/gp_models_utils_casadi.py:17-40,160-197
// shapes:   /root/reference/safe_exploration/ssm_gpy/gaussian_process.py:546-596
#include "sr_common.h"
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

template <int NP, int DT, bool LIN>
__global__ __launch_bounds__(1024) void sr_gp_small_kernel(sr_kstar_args a, const double* __restrict__ Wt,
                                                           double* __restrict__ mu, double* __restrict__ var,
                                                           double* __restrict__ jac, double* __restrict__ jac_var,
                                                           double* __restrict__ hess) {
    SR_SMALL_LDS_DECL(NP, DT);
    const int tid = threadIdx.x;
    const int d = blockIdx.y;
    const long t0 = (long)blockIdx.x * SR_FQ;
    const double sf2 = a.sf2[d];
    sr_small_posterior<NP, DT, LIN>(a, Wt, d, a.xa + t0 * a.lda, a.lda, a.xb + t0 * a.ldb, a.ldb, a.T - t0, L);

    sr_small_outputs<NP, DT, LIN>(a, L, d, t0, sf2, mu, var, jac, jac_var, hess);

    if (a.host_flag) {
        // blocking single query: the outputs above went to pinned host memory; once every workgroup's stores are
        // out (system-scope fence), the last one to arrive publishes the sequence number
        __threadfence_system();
        __syncthreads();
        if (tid == 0) {
            const unsigned old = __hip_atomic_fetch_add(a.done_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old == gridDim.x * gridDim.y - 1u) {
                __hip_atomic_store(a.done_ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(a.host_flag, a.host_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K0s: RESIDENT single-query server ("latency-optimised single-query kernel ... persistent kernel, pinned
// host buffers").  The production loop evaluates the model once per IPOPT callback (CasadiSSMEvaluator.eval / JacFun.eval,
// /root/reference/safe_exploration/state_space_models.py:278-303, 384-417): one query, the host blocks.  Launched per
// call, K0 costs the dispatch latency of a 1024-thread workgroup on both sides of ~5 us of work (blocking call 22 - 30 us
// at N = 100 .. 200).  Here one workgroup per output STAYS on its CU and polls a mailbox in pinned host memory:
//   host:   x (D doubles) and the command word into the mailbox, then the sequence number (one cache line, written in
//           this order by ordinary stores); spins on the n_out reply words
//   device: lane 0 of each workgroup polls the sequence word with system-scope loads; on a hit the workgroup runs phases
//           A - C of K0 (first order, or LIN for the second-order outputs), stores the results to the pinned reply block,
//           fences (system scope) and stores the sequence number into ITS reply word
// No launch, no copy command and no completion interrupt on the path: one PCIe read to see the request, one posted
// write to answer it.  The workgroup leaves on the STOP command or when no request arrived for idle_ticks (its `alive`
// word in the reply block then reads 0 and the host relaunches it with the next request): a device-wide synchronisation
// elsewhere in the process waits at most that long.  The model is read through the L2 like K0 does (it cannot change
// while the server runs: every entry point that writes it stops the server first).
// ------------------------------------------------------------------------------------------------
// Phase B with the wavefront's fragments of U^-1 held in REGISTERS across requests (NP = 128: 9 doubles per lane; the
// strips of a wavefront and their k ranges as in sr_small_contract).  The run length of strip A is wavefront-uniform but
// not a compile-time constant: one straight-line body per length (a branch per MFMA would serialise the LDS reads).
template <int NP, int HELD_ = -1>
struct sr_srv_frag {
    static constexpr int NSTRIP = NP / 16, NPAIR = NSTRIP / 2, NSPLIT = 16 / NPAIR;
    static constexpr int TOT = 4 * (NSTRIP + 1) / NSPLIT;            // k-steps (of 4 rows) of a wavefront, both strips
    static constexpr int HELD = HELD_ < 0 ? TOT : (HELD_ < TOT ? HELD_ : TOT);      // how many of them this object holds
    double w[HELD];
    __device__ __forceinline__ static const double* addr(const double* __restrict__ Wd, int wave, int lane, int u) {
        const int lk = lane >> 4, ln = lane & 15;
        const int pr = wave / NSPLIT, h = wave % NSPLIT;
        const int nA = 4 * (pr + 1) / NSPLIT;
        const bool inA = u < nA;
        const int sidx = inA ? pr : NSTRIP - 1 - pr;
        const int chunk = 4 * (sidx + 1) / NSPLIT;
        const int st = h * chunk + (inA ? u : u - nA);
        return Wd + (long)(4 * st + lk) * NP + 16 * sidx + ln;
    }
    __device__ __forceinline__ void load(const double* __restrict__ Wd, int wave, int lane) {
#pragma unroll
        for (int u = 0; u < HELD; ++u) w[u] = *addr(Wd, wave, lane, u);
    }
};
template <int NP, int NA, class F>
__device__ __forceinline__ void sr_srv_mfma(const F& f, const double* __restrict__ Wd, int wave, int lane,
                                            const double (*ks)[SR_FQ], int stA, int stB, int lk, int ln, sr_d4 (&acc)[2]) {
    constexpr int TOT = F::TOT, HELD = F::HELD;
    double bf[TOT], rest[TOT - HELD > 0 ? TOT - HELD : 1];
#pragma unroll
    for (int u = HELD; u < TOT; ++u) rest[u - HELD] = *F::addr(Wd, wave, lane, u);      // (what the object does not hold)
#pragma unroll
    for (int u = 0; u < TOT; ++u) bf[u] = ks[4 * ((u < NA) ? stA + u : stB + (u - NA)) + lk][ln];
    sr_d4 a = {0.0, 0.0, 0.0, 0.0}, b = a;
#pragma unroll
    for (int u = 0; u < TOT; ++u) {
        const double wv = u < HELD ? f.w[u < HELD ? u : 0] : rest[u >= HELD ? u - HELD : 0];
        if (u < NA) a = __builtin_amdgcn_mfma_f64_16x16x4f64(wv, bf[u], a, 0, 0, 0);
        else b = __builtin_amdgcn_mfma_f64_16x16x4f64(wv, bf[u], b, 0, 0, 0);
    }
    acc[0] = a; acc[1] = b;
}
// same contract as sr_small_contract<NP, true> (DOT0: columns dotted with column 0)
template <int NP, class F>
__device__ __forceinline__ void sr_srv_contract(const F& f, const double* __restrict__ Wd, const double (*ks)[SR_FQ], double* pB,
                                                double (*redC)[SR_FQ], int wave, int lane) {
    constexpr int NSTRIP = F::NSTRIP, NSPLIT = F::NSPLIT;
    static_assert(NSPLIT >= 2 && NSTRIP <= 16, "register-held fragments: Np <= 256");
    const int lk = lane >> 4, ln = lane & 15;
    const int pr = wave / NSPLIT, h = wave % NSPLIT;
    const int nA = 4 * (pr + 1) / NSPLIT, nB = 4 * (NSTRIP - pr) / NSPLIT;
    sr_d4 accB[2];
    const int stA = h * nA, stB = h * nB;
    switch (nA) {
#define SRV_CASE(NA_) case NA_: if constexpr (NA_ < F::TOT) sr_srv_mfma<NP, NA_>(f, Wd, wave, lane, ks, stA, stB, lk, ln, accB); break;
        SRV_CASE(1) SRV_CASE(2) SRV_CASE(3) SRV_CASE(4) SRV_CASE(6) SRV_CASE(8) SRV_CASE(10) SRV_CASE(12) SRV_CASE(14) SRV_CASE(16)
#undef SRV_CASE
        default: accB[0] = accB[1] = sr_d4{0.0, 0.0, 0.0, 0.0}; break;
    }
#pragma unroll
    for (int which = 0; which < 2; ++which) {
        const int sidx = which ? NSTRIP - 1 - pr : pr;
        if (h > 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) pB[((h - 1) * NSTRIP + sidx) * 256 + r * 64 + lane] = accB[which][r];
        }
    }
    __syncthreads();
    if (h == 0) {
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            const int sidx = which ? NSTRIP - 1 - pr : pr;
            double q = 0.0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double v = accB[which][r];
#pragma unroll
                for (int hh = 0; hh < NSPLIT - 1; ++hh) v += pB[(hh * NSTRIP + sidx) * 256 + r * 64 + lane];
                const double w = __shfl(v, lane & 48);               // dot with column 0 of the same row
                q = fma(v, w, q);
            }
            q += __shfl_xor(q, 16);
            q += __shfl_xor(q, 32);
            if (lane < 16) redC[sidx][lane] = q;
        }
    }
    __syncthreads();
}

// (the model travels as the few words the evaluation needs -- sr_server_model -- and the argument block of phase A is
//  rebuilt from them every round: with the whole sr_kstar_args live across the loop's back edge the scalar registers run
//  out, spill into vector lanes and those into scratch: 12 .. 380 B per lane)
struct sr_server_model { const double *Z, *alpha, *ls, *sf2, *Wt; int N, D, n_out; };
template <int NP, int DT>
__global__ __launch_bounds__(1024) void sr_gp_server_kernel(sr_server_model m, sr_server_args sv) {
    // U^-1 fragments of the wavefront (9 doubles per lane at Np = 128) in registers ACROSS requests with D <= 3 (REGS); with
    // D = 5 that costs 20 B of scratch: there six of the nine are fetched at the START of an evaluation, so that their L2
    // round trip runs under phase A instead of after it (EARLY).  Np = 256 has 34 per lane: neither fits beside the
    // working set of the straight-line contraction (fetching 8 .. 16 of them early: 20 - 84 B of scratch); it reads them
    // after phase A like the launched kernel.
    constexpr bool REGS = NP <= 128 && DT <= 3;
    constexpr bool EARLY = !REGS && NP <= 128;
    constexpr int HELD = REGS ? -1 : 6;
    SR_SMALL_LDS_DECL(NP, DT);
    __shared__ double rows_[NP][DT + 1];      // the training rows of phase A, pre-scaled, with alpha: fetched once
    __shared__ double il_[DT];
    __shared__ double xreq[8];
    __shared__ unsigned long long req_cmd;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int d = blockIdx.y;
    const int D = m.D, n = m.n_out;
    double* out = sv.out;
    unsigned long long expect = sv.first_seq;
    sr_srv_frag<(REGS || EARLY) ? NP : 128, HELD> frag;
    {
        sr_kstar_args a0{};
        a0.Z = m.Z; a0.alpha = m.alpha; a0.ls = m.ls; a0.N = m.N; a0.Np = NP; a0.D = D; a0.n_out = n;
        sr_small_rows_fill<NP, DT>(a0, d, rows_, 1024);
        sr_small_il_fill<DT>(a0, d, il_);
        if (REGS) frag.load(m.Wt + (long)d * NP * NP, wave, lane);
    }
    const double sf2 = m.sf2[d];
    __syncthreads();
    const sr_small_rows<NP, DT> rows{rows_, il_};
    for (;;) {
        if (wave == 0) {
            // the mailbox is ONE 64-byte line [x0 .. x4 | launch epoch | command | sequence number]: lanes 0 .. 7 fetch it with
            // one request, so a hit on the sequence number (written last by the host) comes with the query it belongs to; an
            // epoch other than this launch's means STOP, whatever request a workgroup is waiting for
            unsigned long long cmd = SR_SERVER_CMD_IDLE;
            const unsigned long long t_last = wall_clock64();             // 100 MHz
            for (;;) {
                unsigned long long wv = 0;
                if (lane < 8) wv = __hip_atomic_load(sv.mb + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                const unsigned long long s = __shfl(wv, 7);
                if (__shfl(wv, 5) != sv.epoch) { cmd = SR_SERVER_CMD_STOP; break; }      // the host called this launch off
                if (s == expect) {
                    cmd = __shfl(wv, 6);
                    if (lane < 6) xreq[lane] = __longlong_as_double((long long)wv);
                    break;
                }
                if (wall_clock64() - t_last > sv.idle_ticks) break;
                __builtin_amdgcn_s_sleep(2);
            }
            if (lane == 0) req_cmd = cmd;
        }
        __syncthreads();
        const unsigned long long cmd = req_cmd;
        if (cmd == SR_SERVER_CMD_STOP || cmd == SR_SERVER_CMD_IDLE) break; // stop command or idle time-out
        const unsigned long long t_seen = wall_clock64();
        if (cmd == SR_SERVER_CMD_PING) {                                   // diagnostics: answer without evaluating
            if (tid == 0) __hip_atomic_store(sv.reply + d, expect, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            ++expect;
            __syncthreads();
            continue;
        }
        // Always the second-order evaluation: its outputs contain the first-order ones (mu, var, d mu/dx lead the reply block),
        // it costs ~2 us more than the first-order pass, and ONE code path inside the loop keeps the kernel within its 128
        // registers per lane (both paths inlined: 60 - 508 B of scratch per lane).
        {
            // (the pointers and the thread index pass through an empty asm every round: what the compiler can prove
            //  loop-invariant -- the address arithmetic of the U^-1 fragments, 38 pointers per lane -- it hoists in front
            //  of the polling loop and spills)
            const double* pW = m.Wt;
            int tq = tid;
            asm volatile("" : "+s"(pW), "+v"(tq));
            sr_kstar_args a{};
            a.sf2 = m.sf2;
            a.lda = D; a.na = D; a.N = m.N; a.Np = NP; a.D = D; a.n_out = n; a.nsplit = 1; a.T = 1; a.Tp = 1;
            if constexpr (EARLY) frag.load(pW + (long)d * NP * NP, __builtin_amdgcn_readfirstlane(tq >> 6), tq & 63);
            sr_small_phase_a<NP, DT, true, true, 16>(a, d, xreq, D, xreq, D, 1, L, &rows, tq);
            if constexpr (REGS || EARLY) sr_srv_contract<NP>(frag, pW + (long)d * NP * NP, L.ks, L.pB, L.redC, __builtin_amdgcn_readfirstlane(tq >> 6), tq & 63);
            else sr_small_contract<NP, true>(pW + (long)d * NP * NP, L.ks, L.pB, L.redC, tq >> 6, tq & 63);
        }
        // The answer of this output is ONE record [mu, var, d mu/dx (D), d var/dx (D), d2 mu/dx2 (D x D)] of 2 + 2 D + D^2 <= 37
        // doubles: lane e of the first wavefront forms element e and ONE store instruction carries the record to the pinned
        // reply block (scattered over the API layout, five store instructions to host memory cost 4 us of the 8).
        if (wave == 0) {
            constexpr int NSTRIP = NP / 16;
            const int e = lane, R = 2 + 2 * D + D * D;
            const double mval = L.Rs[0][0];
            double val = 0.0;
            if (e == 0) val = mval;
            else if (e < 2 + D) {
                if (e >= 2) val = L.Rs[1 + (e - 2)][0];
            } else if (e < 2 + 2 * D) {
            } else if (e < R) {
                const int q = e - (2 + 2 * D);
                const int j = min(q / D, q % D), l = max(q / D, q % D);
                val = (L.Rs[1 + j][1 + l] - L.xq[0][l] * L.Rs[1 + j][0]) * il_[l];
                if (j == l) val -= mval * il_[j] * il_[j];
            }
            // var (element 1) and d var/dx_j (elements 2 + D + j): column c = 0 resp. 1 + j of the strip sums
            const int c = (e == 1) ? 0 : ((e >= 2 + D && e < 2 + 2 * D) ? e - (2 + D) + 1 : -1);
            if (c >= 0) {
                double qn = 0.0;
#pragma unroll
                for (int sidx = 0; sidx < NSTRIP; ++sidx) qn += L.redC[sidx][c];
                if (c == 0) {
                    val = sf2 - qn;
                    if (!(val > SR_VAR_CLIP)) val = SR_VAR_CLIP;
                } else val = -2.0 * qn;
            }
            if (e < R) out[(long)d * SR_SERVER_REC + e] = val;
            __threadfence_system();
            if (lane == 0) {
                // (diagnostics: ticks of the 100 MHz clock this evaluation took on the device, request seen -> results fenced)
                sv.reply[2 * SR_SERVER_ALIVE + d] = wall_clock64() - t_seen;
                __hip_atomic_store(sv.reply + d, expect, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        ++expect;
    }
    if (tid == 0) __hip_atomic_store(sv.reply + SR_SERVER_ALIVE + d, 0ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- K0s in PARTS: Np = 256, 384, 512 ------------------------------------------------------------------------------------
// One 16-wavefront workgroup cannot keep the U^-1 fragments of a 256-row model on chip (34 doubles per lane against a budget
// of 128 registers).  Here an output is served by Np / 64 workgroups of EIGHT wavefronts (256 registers per lane), each
// owning two strip pairs (strips s and Np/16 - 1 - s: whole COLUMNS of U^-1, so no partial products cross workgroups) with
// the k range of a pair cut over four wavefronts: Np / 16 + 1 fragments per lane, in registers across requests.  Every part
// polls the mailbox itself and evaluates phase A in full (it needs all of k*); part 0 answers with mu, d mu/dx, d2 mu/dx2,
// every part with its strips' share of |U^-T k*|^2 and of the dot products with the dk*/dx columns; the HOST adds the
// Np / 64 shares in a fixed order (sr_gp_server_call).  
template <int NP>
struct sr_part_frag {
    static constexpr int NSTRIP = NP / 16, TOT = NSTRIP + 1;         // k-steps (of 4 rows) of a wavefront, both strips
    double w[TOT];
    // pair pr (strips pr and NSTRIP - 1 - pr), quarter h of their k ranges: pr + 1 resp. NSTRIP - pr steps
    __device__ __forceinline__ void load(const double* __restrict__ Wd, int pr, int h, int lane) {
        const int lk = lane >> 4, ln = lane & 15;
        const int nA = pr + 1;
#pragma unroll
        for (int u = 0; u < TOT; ++u) {
            const bool inA = u < nA;
            const int sidx = inA ? pr : NSTRIP - 1 - pr;
            const int st = h * (sidx + 1) + (inA ? u : u - nA);
            w[u] = Wd[(long)(4 * st + lk) * NP + 16 * sidx + ln];
        }
    }
};
template <int NP, int NA>
__device__ __forceinline__ void sr_part_mfma(const sr_part_frag<NP>& f, const double (*ks)[SR_FQ], int stA, int stB, int lk, int ln,
                                             sr_d4 (&acc)[2]) {
    constexpr int TOT = sr_part_frag<NP>::TOT;
    sr_d4 a0 = {0.0, 0.0, 0.0, 0.0}, a1 = a0, b0 = a0, b1 = a0;
#pragma unroll
    for (int u = 0; u < TOT; ++u) {
        const double bf = ks[4 * ((u < NA) ? stA + u : stB + (u - NA)) + lk][ln];
        if (u < NA) {
            if (u & 1) a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(f.w[u], bf, a1, 0, 0, 0);
            else a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(f.w[u], bf, a0, 0, 0, 0);
        } else {
            if (u & 1) b1 = __builtin_amdgcn_mfma_f64_16x16x4f64(f.w[u], bf, b1, 0, 0, 0);
            else b0 = __builtin_amdgcn_mfma_f64_16x16x4f64(f.w[u], bf, b0, 0, 0, 0);
        }
    }
    acc[0] = a0 + a1; acc[1] = b0 + b1;
}

template <int NP, int DT>
__global__ __launch_bounds__(512) void sr_gp_server_parts_kernel(sr_server_model m, sr_server_args sv) {
    constexpr int NSTRIP = NP / 16, NPAIR = NSTRIP / 2, PARTS = NP / 64;
    static_assert(NPAIR == 2 * PARTS, "two strip pairs per part");
    __shared__ double ks_[NP][SR_FQ];
    __shared__ double xq_[SR_FQ][DT];
    __shared__ double pA_[8][256];
    __shared__ double Rs_[SR_FQ][16];
    __shared__ double pB_[3 * 4 * 256];        // quarters h = 1 .. 3 of the part's four strips
    __shared__ double redC_[NP / 16][SR_FQ];
    sr_small_lds<NP, DT> L{ks_, xq_, pA_, Rs_, pB_, redC_};
    __shared__ double rows_[NP][DT + 1];
    __shared__ double il_[DT];
    __shared__ double xreq[8];
    __shared__ unsigned long long req_cmd;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int part = blockIdx.x, d = blockIdx.y;
    const int D = m.D, n = m.n_out;
    const int pr = 2 * part + (wave >> 2), h = wave & 3;      // this wavefront's strip pair and quarter
    const int lp = wave >> 2;                                  // local pair: local strips 2 lp (A) and 2 lp + 1 (B)
    double* out = sv.out + ((long)d * PARTS + part) * SR_SERVER_REC;
    unsigned long long expect = sv.first_seq;
    sr_part_frag<NP> frag;
    {
        sr_kstar_args a0{};
        a0.Z = m.Z; a0.alpha = m.alpha; a0.ls = m.ls; a0.N = m.N; a0.Np = NP; a0.D = D; a0.n_out = n;
        sr_small_rows_fill<NP, DT>(a0, d, rows_, 512);
        sr_small_il_fill<DT>(a0, d, il_);
        frag.load(m.Wt + (long)d * NP * NP, pr, h, lane);
    }
    const double sf2 = m.sf2[d];
    __syncthreads();
    const sr_small_rows<NP, DT> rows{rows_, il_};
    for (;;) {
        if (wave == 0) {
            unsigned long long cmd = SR_SERVER_CMD_IDLE;
            const unsigned long long t_last = wall_clock64();
            for (;;) {
                unsigned long long wv = 0;
                if (lane < 8) wv = __hip_atomic_load(sv.mb + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                const unsigned long long s = __shfl(wv, 7);
                if (__shfl(wv, 5) != sv.epoch) { cmd = SR_SERVER_CMD_STOP; break; }      // the host called this launch off
                if (s == expect) {
                    cmd = __shfl(wv, 6);
                    if (lane < 6) xreq[lane] = __longlong_as_double((long long)wv);
                    break;
                }
                if (wall_clock64() - t_last > sv.idle_ticks) break;
                __builtin_amdgcn_s_sleep(2);
            }
            if (lane == 0) req_cmd = cmd;
        }
        __syncthreads();
        const unsigned long long cmd = req_cmd;
        if (cmd == SR_SERVER_CMD_STOP || cmd == SR_SERVER_CMD_IDLE) break;
        const unsigned long long t_seen = wall_clock64();
        const int slot = d * PARTS + part;
        if (cmd == SR_SERVER_CMD_PING) {
            if (tid == 0) __hip_atomic_store(sv.reply + slot, expect, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            ++expect;
            __syncthreads();
            continue;
        }
        {
            int tq = tid;
            asm volatile("" : "+v"(tq));                       // (nothing derived from the thread index is loop-invariant)
            sr_kstar_args a{};
            a.sf2 = m.sf2;
            a.lda = D; a.na = D; a.N = m.N; a.Np = NP; a.D = D; a.n_out = n; a.nsplit = 1; a.T = 1; a.Tp = 1;
            sr_small_phase_a<NP, DT, true, true, 8>(a, d, xreq, D, xreq, D, 1, L, &rows, tq);
            // phase B on the part's strips
            const int lk = (tq & 63) >> 4, ln = tq & 15;
            sr_d4 acc[2];
            const int nA = pr + 1, nB = NSTRIP - pr;
            switch (nA) {
#define SRP_CASE(NA_) case NA_: if constexpr (NA_ <= NPAIR) sr_part_mfma<NP, NA_>(frag, ks_, h * nA, h * nB, lk, ln, acc); break;
                SRP_CASE(1) SRP_CASE(2) SRP_CASE(3) SRP_CASE(4) SRP_CASE(5) SRP_CASE(6) SRP_CASE(7) SRP_CASE(8)
                SRP_CASE(9) SRP_CASE(10) SRP_CASE(11) SRP_CASE(12) SRP_CASE(13) SRP_CASE(14) SRP_CASE(15) SRP_CASE(16)
#undef SRP_CASE
                default: acc[0] = acc[1] = sr_d4{0.0, 0.0, 0.0, 0.0}; break;
            }
            if (h > 0) {
#pragma unroll
                for (int which = 0; which < 2; ++which)
#pragma unroll
                    for (int r = 0; r < 4; ++r) pB_[((h - 1) * 4 + 2 * lp + which) * 256 + r * 64 + (tq & 63)] = acc[which][r];
            }
            __syncthreads();
            if (h == 0) {
#pragma unroll
                for (int which = 0; which < 2; ++which) {
                    double q = 0.0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        double v = acc[which][r];
#pragma unroll
                        for (int hh = 0; hh < 3; ++hh) v += pB_[(hh * 4 + 2 * lp + which) * 256 + r * 64 + (tq & 63)];
                        const double w = __shfl(v, (tq & 63) & 48);           // dot with column 0 of the same row
                        q = fma(v, w, q);
                    }
                    q += __shfl_xor(q, 16);
                    q += __shfl_xor(q, 32);
                    if ((tq & 63) < 16) redC_[2 * lp + which][tq & 63] = q;       // (local strip index: 0 .. 3)
                }
            }
            __syncthreads();
        }
        // the part's record: [mu, share of q_0, d mu/dx (D), shares of q_{1+j} (D), d2 mu/dx2 (D x D), .., sf2 at the end]; mu,
        // the mean's derivatives and sf2 from part 0 only
        if (wave == 0) {
            const int e = lane, R = 2 + 2 * D + D * D;
            const double mval = Rs_[0][0];
            double val = 0.0;
            if (part == 0) {
                if (e == 0) val = mval;
                else if (e >= 2 && e < 2 + D) val = Rs_[1 + (e - 2)][0];
                else if (e >= 2 + 2 * D && e < R) {
                    const int q = e - (2 + 2 * D);
                    const int j = min(q / D, q % D), l = max(q / D, q % D);
                    val = (Rs_[1 + j][1 + l] - xq_[0][l] * Rs_[1 + j][0]) * il_[l];
                    if (j == l) val -= mval * il_[j] * il_[j];
                } else if (e == SR_SERVER_REC - 1) val = sf2;
            }
            const int c = (e == 1) ? 0 : ((e >= 2 + D && e < 2 + 2 * D) ? e - (2 + D) + 1 : -1);
            if (c >= 0) val = redC_[0][c] + redC_[1][c] + redC_[2][c] + redC_[3][c];
            if (e < SR_SERVER_REC) out[e] = val;
            __threadfence_system();
            if (lane == 0) {
                sv.reply[2 * SR_SERVER_ALIVE + slot] = wall_clock64() - t_seen;
                __hip_atomic_store(sv.reply + slot, expect, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        ++expect;
    }
    if (tid == 0) __hip_atomic_store(sv.reply + SR_SERVER_ALIVE + d * PARTS + part, 0ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

template <int NP>
static int launch_server_np(const sr_kstar_args& a, const double* Wt, const sr_server_args& sv, hipStream_t s) {
    const sr_server_model m{a.Z, a.alpha, a.ls, a.sf2, Wt, a.N, a.D, a.n_out};
    SR_CHECK(sr_gp_server_supported(NP, a.D), SR_EUNSUPPORTED, "gp_server: Np=%d D=%d not built", NP, a.D);
    if constexpr (NP >= 256) {
        static_assert(sr_gp_server_parts(NP) == NP / 64, "parts");
        dim3 grid(NP / 64, a.n_out);
        if (a.D <= 3) hipLaunchKernelGGL((sr_gp_server_parts_kernel<NP, 3>), grid, dim3(512), 0, s, m, sv);
        else hipLaunchKernelGGL((sr_gp_server_parts_kernel<NP, 5>), grid, dim3(512), 0, s, m, sv);
    } else {
        dim3 grid(1, a.n_out);
        if (a.D <= 3) hipLaunchKernelGGL((sr_gp_server_kernel<NP, 3>), grid, dim3(1024), 0, s, m, sv);
        else hipLaunchKernelGGL((sr_gp_server_kernel<NP, 5>), grid, dim3(1024), 0, s, m, sv);
    }
    SR_HIP(hipGetLastError());
    return SR_OK;
}

int sr_launch_gp_server(const sr_kstar_args& a, const double* Wt, const sr_server_args& sv, hipStream_t s) {
    if (a.Np == 128) return launch_server_np<128>(a, Wt, sv, s);
    if (a.Np == 256) return launch_server_np<256>(a, Wt, sv, s);
    if (a.Np == 384) return launch_server_np<384>(a, Wt, sv, s);
    if (a.Np == 512) return launch_server_np<512>(a, Wt, sv, s);
    sr_set_error("gp_server: Np=%d not supported", a.Np);
    return SR_EUNSUPPORTED;
}

// ------------------------------------------------------------------------------------------------
// K0c: the H-step reachability chain of a small model in ONE launch (multi_step_reachability,
// /root/reference/safe_exploration/gp_reachability.py:159-212; moment chains of
// uncertainty_propagation_casadi.py:88-190 through `mode`).
//
// Launched per step the chain costs two dependent launches per step (posterior 10.7 us + ellipsoid 4.7 us at N = 200:
// 0.23 ms for H = 15), much of it launch latency.  Here a group of 16 rollouts is served, for ALL steps, by
//   n_out x P POSTERIOR workgroups (g, d, part): output d, P = Np / 128 of them sharing the contraction with U^-1 of one
//        (g, d).  Fetched once: the wavefront's fragments of U^-1 into REGISTERS (2 (Np / 16 + 1) doubles per lane; 8
//        wavefronts per workgroup = 256 VGPRs per lane), the training rows of phase A and 1 / l into LDS, the group's
//        feed-forward controls into LDS.  Per step i:
//            phase A at [p_i, k_ff_i] (all parts: k* is needed in full)
//            part 0: (mu, d mu/dx)[d] -> exchange buffer      (BEFORE the contraction: they do not depend on it)
//            phase B on the part's 4 strip pairs (strips s and Np/16-1-s; 2 wavefronts per pair, each half of the k range:
//                2 (Np / 16 + 1) MFMAs per wavefront whatever the pair), squared and summed per rollout
//            the part's share of |U^-T k*|^2 -> exchange buffer
//            poll the means of ALL outputs of step i, move the centres: p_{i+1} = a p_i + b u_i + mu
//   one TAIL workgroup: polls (mu, d mu/dx, shares of |U^-T k*|^2) of every step as they appear, runs the ellipsoid step of
//        the 16 rollouts (sr_ellipsoid_one, one lane each, state (p, Q) in LDS), writes p_all / q_all.
// The centres do not depend on the shape matrices, so the posterior workgroups never wait for an ellipsoid step (n_s = 4:
// 8 us of Jacobi rotations): the Q chain trails the chain of centres.
// Exchange: every element is 16 bytes (value, bits(value) ^ mix(tag)), tag = the group's epoch + step + 1, written by ONE
// agent-scope store and read by one agent-scope load; a reader polls until value and check word agree for this step's
// tag.  No ticket, no fence, no wait for the stores to be acknowledged; every (step, output) has its own slot, so
// nothing is overwritten inside a launch; the epoch lives on the device (a captured launch can be replayed).
// All groups x (n_out P + 1) workgroups of a launch must be resident at once (<= SR_CHAIN_GROUPS, one per CU; other work
// on the device only delays them); a poll that does not end within SR_CHAIN_TIMEOUT_TICKS (100 ms) raises the status
// word and poisons the group's outputs with NaN instead of hanging the device.
// (n_out = 1 with Np = 128 needs no exchange: one workgroup does everything.)
// History, per 15-step chain of 256 rollouts at N = 200: per-step launches 241 us; one workgroup per (g, d), U^-1 from L2
// every step 188; fragments in registers, tickets, ellipsoid step in every workgroup 148; tagged elements 128; tail
// workgroup + mean published before the contraction 102.
// ------------------------------------------------------------------------------------------------
#define SR_CHAIN_PARTS(NP) ((NP) / 128)
#define SR_CHAIN_NW 8                /* wavefronts per workgroup: 256 registers per lane, room for the U^-1 fragments */
#define SR_CHAIN_TOT(NP) (2 * ((NP) / 16 + 1))

// strips and k-ranges of a wavefront in the split contraction: pair pr = 4 part + wave / 2, half h = wave % 2
template <int NP>
struct sr_flat_geo {
    int sA, sB, nA, stA, stB;
    __device__ __forceinline__ sr_flat_geo(int part, int wave) {
        const int pr = 4 * part + (wave >> 1), h = wave & 1;
        sA = pr; sB = NP / 16 - 1 - pr;
        nA = 2 * (sA + 1);                     // k-steps (of 4 rows) of this half of strip A; strip B: 2 (sB + 1)
        stA = h * nA; stB = h * 2 * (sB + 1);
    }
};

template <int NP>
__device__ __forceinline__ void sr_flat_load(const double* __restrict__ Wd, int part, int wave, int lane,
                                             double (&w)[SR_CHAIN_TOT(NP)]) {
    const sr_flat_geo<NP> g(part, wave);
    const int lk = lane >> 4, ln = lane & 15;
#pragma unroll
    for (int u = 0; u < SR_CHAIN_TOT(NP); ++u) {
        const bool inA = u < g.nA;
        const int st = inA ? g.stA + u : g.stB + (u - g.nA);
        const int strip = inA ? g.sA : g.sB;
        w[u] = Wd[(long)(4 * st + lk) * NP + 16 * strip + ln];
    }
}

// The MFMAs of one wavefront with the length NA of its strip-A run known at compile time: straight-line code, so that
// the scheduler batches the LDS reads of the B-fragments (with a wavefront-uniform branch per MFMA every read waited for
// its own latency: 4.1 us per step at Np = 256 instead of the 1.8 us the MFMA pipe needs).  Two accumulators per strip
// break the dependent chain.
template <int NP, int NA>
__device__ __forceinline__ void sr_flat_mfma(const double (&w)[SR_CHAIN_TOT(NP)], const double (*ks)[SR_FQ], int stA,
                                             int stB, int lk, int ln, sr_d4 (&acc)[2]) {
    constexpr int TOT = SR_CHAIN_TOT(NP);
    sr_d4 a0 = {0.0, 0.0, 0.0, 0.0}, a1 = a0, b0 = a0, b1 = a0;
#pragma unroll
    for (int u = 0; u < NA; u += 2) {
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(w[u], ks[4 * (stA + u) + lk][ln], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(w[u + 1], ks[4 * (stA + u + 1) + lk][ln], a1, 0, 0, 0);
    }
#pragma unroll
    for (int u = NA; u < TOT; u += 2) {
        b0 = __builtin_amdgcn_mfma_f64_16x16x4f64(w[u], ks[4 * (stB + u - NA) + lk][ln], b0, 0, 0, 0);
        b1 = __builtin_amdgcn_mfma_f64_16x16x4f64(w[u + 1], ks[4 * (stB + u + 1 - NA) + lk][ln], b1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        acc[0][r] = a0[r] + a1[r];
        acc[1][r] = b0[r] + b1[r];
    }
}

// redP[2 q + which][t] = sum over the rows of strip (pair q of the part, which) of V[i][t]^2.  Ends with a barrier.
template <int NP>
__device__ __forceinline__ void sr_flat_contract(const double (&w)[SR_CHAIN_TOT(NP)], const double (*ks)[SR_FQ],
                                                 double* pB, double (*redP)[SR_FQ], int part, int wave, int lane) {
    const sr_flat_geo<NP> g(part, wave);
    const int lk = lane >> 4, ln = lane & 15, q = wave >> 1, h = wave & 1;
    sr_d4 acc[2];
    // nA = 2 (pair + 1), pair = 0 .. Np / 32 - 1 (both NA and TOT - NA are even)
#define SR_FLAT_CASE(PR) case PR: if (PR < NP / 32) sr_flat_mfma<NP, (PR < NP / 32) ? 2 * (PR + 1) : 2>(w, ks, g.stA, g.stB, lk, ln, acc); break;
    switch (g.sA) {
        SR_FLAT_CASE(0) SR_FLAT_CASE(1) SR_FLAT_CASE(2) SR_FLAT_CASE(3) SR_FLAT_CASE(4) SR_FLAT_CASE(5) SR_FLAT_CASE(6)
        SR_FLAT_CASE(7) SR_FLAT_CASE(8) SR_FLAT_CASE(9) SR_FLAT_CASE(10) SR_FLAT_CASE(11) SR_FLAT_CASE(12)
        SR_FLAT_CASE(13) SR_FLAT_CASE(14) SR_FLAT_CASE(15)
    }
#undef SR_FLAT_CASE
    if (h > 0) {
#pragma unroll
        for (int which = 0; which < 2; ++which)
#pragma unroll
            for (int r = 0; r < 4; ++r) pB[((2 * q + which) * 4 + r) * 64 + lane] = acc[which][r];
    }
    __syncthreads();
    if (h == 0) {
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            double sq = 0.0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double v = acc[which][r] + pB[((2 * q + which) * 4 + r) * 64 + lane];
                sq = fma(v, v, sq);
            }
            sq += __shfl_xor(sq, 16);
            sq += __shfl_xor(sq, 32);
            if (lane < 16) redP[2 * q + which][lane] = sq;
        }
    }
    __syncthreads();
}

// One element of the exchange buffer of the chain kernel: the value and a check word = bits(value) ^ mix(tag) travel in
// ONE 16-byte store / load.  A reader accepts an element when value and check word agree for THIS step's tag, so what
// it accepts is this step's value even if the two 8-byte halves of an element should ever become visible separately
// (an old half next to a new one fails the check unless the two values are equal).
__device__ __forceinline__ unsigned long long sr_xel_mix(unsigned long long tag) { return tag * 0x9E3779B97F4A7C15ull; }
__device__ __forceinline__ void sr_xel_store(sr_xel* p, double v, unsigned long long tag) {
    typedef unsigned sr_u4 __attribute__((ext_vector_type(4)));
    const unsigned long long vb = (unsigned long long)__double_as_longlong(v), cb = vb ^ sr_xel_mix(tag);
    const sr_u4 x = {(unsigned)vb, (unsigned)(vb >> 32), (unsigned)cb, (unsigned)(cb >> 32)};
    // (the s_nop: a store of more than 8 bytes per lane must not be followed directly by a write to its data registers
    //  -- the compiler's hazard recogniser does not look inside inline assembly; without it the last lanes of the store
    //  left with the next instruction's values)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(x) : "memory");
}
// polls n elements p[0], p[stride], ... (all loads in flight together) until every one carries `tag`; false = timed out
template <int N>
__device__ __forceinline__ bool sr_xel_poll(const sr_xel* p, long stride, unsigned long long tag, double (&v)[N]) {
    typedef unsigned sr_u4 __attribute__((ext_vector_type(4)));
    const unsigned long long want = sr_xel_mix(tag);
    unsigned long long t_start = 0;
    for (int spin = 0;; ++spin) {
        sr_u4 x[N];
#pragma unroll
        for (int e = 0; e < N; ++e)
            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(x[e]) : "v"(p + e * stride) : "memory");
        bool ok = true;
#pragma unroll
        for (int e = 0; e < N; ++e) {
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(x[e]) : : "memory");
            const unsigned long long vb = (unsigned long long)x[e][1] << 32 | x[e][0];
            const unsigned long long cb = (unsigned long long)x[e][3] << 32 | x[e][2];
            ok = ok && ((vb ^ cb) == want);
        }
        if (ok) {
#pragma unroll
            for (int e = 0; e < N; ++e) v[e] = __longlong_as_double((long long)((unsigned long long)x[e][1] << 32 | x[e][0]));
            return true;
        }
        // 100 ms: a workgroup of the group never came (a launch on a stream whose CU mask holds fewer CUs than the grid
        // has workgroups would wait for ever; other work on the device only delays it)
        if ((spin & 63) == 0) {
            const unsigned long long now = wall_clock64();             // 100 MHz
            if (spin == 0) t_start = now;
            else if (now - t_start > SR_CHAIN_TIMEOUT_TICKS) {
#pragma unroll
                for (int e = 0; e < N; ++e) v[e] = 0.0;
                return false;
            }
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

// centre of the next step: p1 = a p + b u + mu (gp_reachability.py:82-83 / :115) -- the operation order of
// sr_ellipsoid_one, so the workgroups that only follow the centres and the one that runs the full step agree bit for bit
template <int NS, int NU>
__device__ __forceinline__ double sr_center_next(const double* a_row, const double* b_row, const double* p,
                                                 const double* u, double mu) {
    double s = mu;
#pragma unroll
    for (int j = 0; j < NS; ++j) s = fma(a_row[j], p[j], s);
#pragma unroll
    for (int k = 0; k < NU; ++k) s = fma(b_row[k], u[k], s);
    return s;
}

#define SR_CHAIN_TAIL(NP, NS) ((NS) > 1 || SR_CHAIN_PARTS(NP) > 1)
#define SR_CHAIN_WPG(NP, NS) ((NS) * SR_CHAIN_PARTS(NP) + (SR_CHAIN_TAIL(NP, NS) ? 1 : 0))      /* workgroups per group */
#define SR_CHAIN_BS(NP, D) (SR_FQ * ((D) + 1) + SR_FQ * SR_CHAIN_PARTS(NP))      /* exchange elements per (group, step, output) */

template <int NP, int DT, int NS, int NU>
__global__ __launch_bounds__(64 * SR_CHAIN_NW) void sr_chain_kernel(sr_chain_args c) {
    constexpr int D = NS + NU;
    constexpr int P = SR_CHAIN_PARTS(NP);
    constexpr int NW = SR_CHAIN_NW, NT = 64 * NW;
    constexpr int TOT = SR_CHAIN_TOT(NP);          // U^-1 fragments (doubles) per lane
    constexpr bool TAIL = SR_CHAIN_TAIL(NP, NS);   // a group has a workgroup of its own for the shape matrices
    constexpr int WPG = SR_CHAIN_WPG(NP, NS);
    constexpr int BS = SR_CHAIN_BS(NP, D);         // per output: [j <= D][16] from part 0 (d mu/dx_j, mu), then [part][16]
    static_assert(D <= DT, "query width");
    static_assert(NS * SR_FQ <= 64, "the centres are moved by one wavefront");
    __shared__ double ks_[NP][SR_FQ];
    __shared__ double xq_[SR_FQ][DT];
    __shared__ double Rs_[SR_FQ][16];
    __shared__ double big_[8 * 256];               // phase A: pA[8][256]; phase B: pB[8 strips][256] (16 KiB)
    __shared__ double redP[8][SR_FQ];
    sr_small_lds<NP, DT> L{ks_, xq_, reinterpret_cast<double (*)[256]>(big_), Rs_, big_, nullptr};
    __shared__ double ps[SR_FQ][NS];               // centres of the 16 rollouts
    __shared__ double qs[SR_FQ][NS * NS];          // shape matrices
    __shared__ double mus[SR_FQ][NS], vars_[SR_FQ][NS], jacs[SR_FQ][NS * D];
    __shared__ double cst[NS * NS + NS * NU + 3 * NS];     // a, b, l_mu, l_sigma, sf2
    __shared__ double rows_[NP][DT + 1];                   // training rows of output d: z_i / l, alpha_i
    __shared__ double il_[DT];                             // 1 / lengthscale of output d
    extern __shared__ double ctl[];                        // k_ff [16][H][NU], then k_fb [16][H-1][NU][NS] of the group
    __shared__ int fail;
    __shared__ unsigned long long base_s;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_out = NS;
    const int wg = blockIdx.x % WPG, g = blockIdx.x / WPG;
    const bool tail = TAIL && wg == NS * P;
    const int part = tail ? 0 : wg % P, d = tail ? 0 : wg / P;
    const long t0 = (long)g * SR_FQ;
    const long nq = c.T - t0 < SR_FQ ? c.T - t0 : SR_FQ;
    const long nss = NS * NS, nus = NU * NS;
    const bool writer = TAIL ? tail : true;
    if (tid == 0) {
        fail = 0;
        // the group's epoch: the tags of this launch are epoch + 1 .. epoch + H (the previous launch's last workgroup
        // to leave moved it past its own)
        base_s = TAIL ? __hip_atomic_load(c.epoch + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
    }

    // everything that does not change from step to step is fetched once
    double* kffs = ctl;
    double* kfbs = ctl + (long)SR_FQ * c.H * NU;
    for (long e = tid; e < nq * c.H * NU; e += NT) kffs[e] = c.k_ff[t0 * c.H * NU + e];
    if (writer)
        for (long e = tid; e < nq * (c.H - 1) * nus; e += NT) kfbs[e] = c.k_fb[t0 * (c.H - 1) * nus + e];
    constexpr int C_B = NS * NS, C_LM = C_B + NS * NU, C_LS = C_LM + NS, C_SF = C_LS + NS;
    if (tid < C_B) cst[tid] = c.a[tid];
    else if (tid < C_LM) cst[tid] = c.b[tid - C_B];
    else if (tid < C_LS) cst[tid] = c.l_mu[tid - C_LM];
    else if (tid < C_SF) cst[tid] = c.l_sigma[tid - C_LS];
    else if (tid < C_SF + NS) cst[tid] = c.k.sf2[tid - C_SF];

    // one ellipsoid step of the group's rollouts in LDS + its results to the caller (the tail workgroup; the only
    // workgroup of a group without exchange)
    auto shape_step = [&](int i) {
        if (tid < nq) {
            sr_ell_args ea;
            ea.T = nq; ea.n_s = NS; ea.n_u = NU;
            if (i == 0) {
                ea.p = c.p0 + t0 * NS; ea.ldp = NS;
                ea.q = c.q0 ? c.q0 + t0 * nss : nullptr; ea.ldq = nss;
                ea.k_fb = c.k_fb0 ? c.k_fb0 + t0 * nus : nullptr; ea.ldkfb = nus;
            } else {
                ea.p = &ps[0][0]; ea.ldp = NS;
                ea.q = &qs[0][0]; ea.ldq = nss;
                ea.k_fb = kfbs + (i - 1) * nus; ea.ldkfb = (long)(c.H - 1) * nus;
            }
            ea.k_ff = kffs + i * NU; ea.ldkff = (long)c.H * NU;
            ea.mu = &mus[0][0]; ea.var = &vars_[0][0]; ea.jac = &jacs[0][0];
            ea.a = cst; ea.b = cst + C_B; ea.l_mu = cst + C_LM; ea.l_sigma = cst + C_LS;
            ea.c_safety = c.c_safety;
            ea.p_out = &ps[0][0]; ea.ldpo = NS;
            ea.q_out = &qs[0][0]; ea.ldqo = nss;
            ea.n_bad = c.n_bad;
            ea.mode = c.mode;
            sr_ellipsoid_one<NS, NU>(ea, tid);
        }
        __syncthreads();
        if (tid < nq * NS) {
            const int t = tid / NS, j = tid % NS;
            c.p_all[((t0 + t) * c.H + i) * NS + j] = ps[t][j];
            if (c.gp_var_all) c.gp_var_all[((t0 + t) * c.H + i) * NS + j] = vars_[t][j];
        }
        if (tid < nq * nss) {
            const int t = tid / (int)nss, j = tid % (int)nss;
            c.q_all[((t0 + t) * c.H + i) * nss + j] = qs[t][j];
        }
    };

    if (tail) {
        // ---- the group's shape matrices: the Q chain trails the chain of centres ------------------------------
        // The centres p_i do not depend on the shape matrices (p_{i+1} = a p_i + b u_i + mu(p_i, u_i)), so the workgroups
        // that evaluate the posterior never wait for an ellipsoid step: this workgroup takes every step's (mu, d mu/dx,
        // sigma^2) from the exchange buffer as it appears, runs the step (one lane per rollout) and writes the results.
        __syncthreads();
        if (tid == 0) __hip_atomic_store(c.alive + g, base_s + 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int i = 0; i < c.H; ++i) {
            const unsigned long long tag = base_s + (unsigned long long)i + 1ull;
            const sr_xel* xi = c.xch + ((long)g * c.H + i) * n_out * BS;
            if (tid < n_out * SR_FQ * (D + 2)) {
                const int o = tid / (SR_FQ * (D + 2)), r = tid % (SR_FQ * (D + 2));
                const int j = r >> 4, t = r & 15;
                if (j <= D) {
                    double v[1];
                    if (!sr_xel_poll<1>(xi + (long)o * BS + r, 0, tag, v)) fail = 1;
                    if (j < D) jacs[t][o * D + j] = v[0]; else mus[t][o] = v[0];
                } else {
                    double sh[P];
                    if (!sr_xel_poll<P>(xi + (long)o * BS + SR_FQ * (D + 1) + t, SR_FQ, tag, sh)) fail = 1;
                    double qn = 0.0;
#pragma unroll
                    for (int pp = 0; pp < P; ++pp) qn += sh[pp];
                    const double v = cst[C_SF + o] - qn;
                    vars_[t][o] = (v > SR_VAR_CLIP) ? v : SR_VAR_CLIP;
                }
            }
            __syncthreads();
            if (fail) break;
            shape_step(i);
            __syncthreads();                 // the results were read from LDS before the next step's values arrive
        }
    } else {
        double wreg[TOT];
        sr_flat_load<NP>(c.Wt + (long)d * NP * NP, part, wave, lane, wreg);
        // training rows of phase A (pre-scaled) in LDS
        constexpr bool KEEP = true;
        sr_small_rows_fill<NP, DT>(c.k, d, rows_, NT);
        sr_small_il_fill<DT>(c.k, d, il_);
        const sr_small_rows<NP, DT> rows{rows_, il_};
        __syncthreads();

        for (int i = 0; i < c.H; ++i) {
            // ---- posterior of output d at the centres of step i ------------------------------------
            const double* xa = (i == 0) ? c.p0 + t0 * NS : &ps[0][0];
            const double* xb = kffs + i * NU;
            sr_small_phase_a<NP, DT, false, KEEP, NW>(c.k, d, xa, NS, xb, (long)c.H * NU, nq, L, &rows);
            __syncthreads();                                   // R complete: the partial-R buffer becomes the partial-V buffer
            // Every 16-byte element of the exchange buffer is (value, tag) written by ONE store, tag = the group's epoch
            // + step + 1: a reader polls the elements it needs until they carry this step's tag -- no ticket, no wait for
            // the stores to be acknowledged.  (mu, d mu/dx)[d] leave before the contraction with U^-1 starts.
            const unsigned long long tag = base_s + (unsigned long long)i + 1ull;
            sr_xel* xo = c.xch + (((long)g * c.H + i) * n_out + d) * BS;
            if (TAIL && part == 0 && tid < SR_FQ * (D + 1)) {
                const int j = tid >> 4, t = tid & 15;
                const double v = (j < D) ? (Rs_[t][1 + j] - xq_[t][j] * Rs_[t][0]) * il_[j] : Rs_[t][0];
                sr_xel_store(xo + tid, v, tag);
            }
            sr_flat_contract<NP>(wreg, ks_, big_, redP, part, wave, lane);
            if (TAIL) {
                if (tid < SR_FQ) {
                    double v = 0.0;
#pragma unroll
                    for (int sidx = 0; sidx < 8; ++sidx) v += redP[sidx][tid];       // this part's share of |U^-T k*|^2
                    sr_xel_store(xo + SR_FQ * (D + 1) + part * SR_FQ + tid, v, tag);
                }
                // ---- the means of all outputs move the centres (first wavefront: reads before writes in lockstep)
                if (tid < NS * SR_FQ) {
                    const int o = tid >> 4, t = tid & 15;
                    double m[1];
                    if (!sr_xel_poll<1>(c.xch + (((long)g * c.H + i) * n_out + o) * BS + SR_FQ * D + t, 0, tag, m)) fail = 1;
                    double pc[NS], uc[NU];
#pragma unroll
                    for (int j = 0; j < NS; ++j) pc[j] = (t < nq) ? xa[t * NS + j] : 0.0;
#pragma unroll
                    for (int k = 0; k < NU; ++k) uc[k] = (t < nq) ? xb[t * (long)c.H * NU + k] : 0.0;
                    const double pn = sr_center_next<NS, NU>(cst + o * NS, cst + C_B + o * NU, pc, uc, m[0]);
                    __builtin_amdgcn_wave_barrier();
                    ps[t][o] = pn;
                }
                __syncthreads();
                if (fail) break;
            } else {
                if (tid < SR_FQ * (D + 2)) {
                    const int j = tid >> 4, t = tid & 15;
                    if (j < D) jacs[t][j] = (Rs_[t][1 + j] - xq_[t][j] * Rs_[t][0]) * il_[j];
                    else if (j == D) mus[t][0] = Rs_[t][0];
                    else {
                        double v = 0.0;
#pragma unroll
                        for (int sidx = 0; sidx < 8; ++sidx) v += redP[sidx][t];
                        v = cst[C_SF] - v;
                        vars_[t][0] = (v > SR_VAR_CLIP) ? v : SR_VAR_CLIP;
                    }
                }
                __syncthreads();
                shape_step(i);
                // (the next phase A starts by reading ps and writes none of the arrays read above before its first barrier)
            }
        }
        // the tail workgroup writes this group's results: it must have been there
        if (TAIL && wg == 0 && !fail) {
            if (tid == 0) {
                const unsigned long long t_start = wall_clock64();
                while (__hip_atomic_load(c.alive + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != base_s + 1ull) {
                    __builtin_amdgcn_s_sleep(8);
                    if (wall_clock64() - t_start > SR_CHAIN_TIMEOUT_TICKS) { fail = 2; break; }
                }
            }
            __syncthreads();
        }
    }
    if (fail && tid == 0 && c.status)     // not silent: the host finds this after synchronising (sr_gp_chain_status)
        __hip_atomic_fetch_or(c.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if ((fail && writer) || fail == 2) {
        const double nan = __builtin_nan("");
        for (long e = tid; e < nq * c.H * NS; e += NT) c.p_all[t0 * c.H * NS + e] = nan;
        for (long e = tid; e < nq * c.H * nss; e += NT) c.q_all[t0 * c.H * nss + e] = nan;
    }
    if (TAIL && tid == 0) {
        // the last workgroup of the group to leave moves the epoch past this launch's tags: the next launch starts from
        // there (whatever happened in this one -- a timed-out group resynchronises itself this way)
        const unsigned old = __hip_atomic_fetch_add(c.done + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == (unsigned)WPG - 1u) {
            __hip_atomic_store(c.done + g, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(c.epoch + g, base_s + (unsigned long long)c.H, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template <int NP, int NS, int NU>
static int launch_chain_np(const sr_chain_args& a, hipStream_t s) {
    constexpr int DT = (NS + NU <= 3) ? 3 : (NS + NU <= 5 ? 5 : 8);
    const unsigned groups = (unsigned)((a.T + SR_FQ - 1) / SR_FQ);
    const size_t ctl_bytes = sizeof(double) * SR_FQ * ((size_t)a.H * NU + (size_t)(a.H - 1) * NU * NS);
    hipLaunchKernelGGL((sr_chain_kernel<NP, DT, NS, NU>), dim3(groups * SR_CHAIN_WPG(NP, NS) - a.test_drop), dim3(64 * SR_CHAIN_NW), ctl_bytes, s, a);
    SR_HIP(hipGetLastError());
    return SR_OK;
}

template <int NP, int NS, int NU>
static int chain_occupancy_np(int H, int* blocks) {
    constexpr int DT = (NS + NU <= 3) ? 3 : (NS + NU <= 5 ? 5 : 8);
    const size_t ctl_bytes = sizeof(double) * SR_FQ * ((size_t)H * NU + (size_t)(H - 1) * NU * NS);
    SR_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks, sr_chain_kernel<NP, DT, NS, NU>, 64 * SR_CHAIN_NW, ctl_bytes));
    return SR_OK;
}
template <int NS, int NU>
static int chain_occupancy_su(int Np, int H, int* blocks) {
    switch (Np) {
        case 128: return chain_occupancy_np<128, NS, NU>(H, blocks);
        case 256: return chain_occupancy_np<256, NS, NU>(H, blocks);
        case 384: return chain_occupancy_np<384, NS, NU>(H, blocks);
        case 512: return chain_occupancy_np<512, NS, NU>(H, blocks);
    }
    *blocks = 0;
    return SR_OK;
}
int sr_chain_blocks_per_cu(int Np, int n_s, int n_u, int H, int* blocks) {
    *blocks = 0;
    if (n_u == 1) {
        if (n_s == 1) return chain_occupancy_su<1, 1>(Np, H, blocks);
        if (n_s == 2) return chain_occupancy_su<2, 1>(Np, H, blocks);
        if (n_s == 3) return chain_occupancy_su<3, 1>(Np, H, blocks);
        if (n_s == 4) return chain_occupancy_su<4, 1>(Np, H, blocks);
    } else if (n_u == 2) {
        if (n_s == 2) return chain_occupancy_su<2, 2>(Np, H, blocks);
        if (n_s == 3) return chain_occupancy_su<3, 2>(Np, H, blocks);
    }
    return SR_OK;
}

template <int NS, int NU>
static int launch_chain_su(const sr_chain_args& a, hipStream_t s) {
    switch (a.k.Np) {
        case 128: return launch_chain_np<128, NS, NU>(a, s);
        case 256: return launch_chain_np<256, NS, NU>(a, s);
        case 384: return launch_chain_np<384, NS, NU>(a, s);
        case 512: return launch_chain_np<512, NS, NU>(a, s);
    }
    sr_set_error("chain: Np=%d not supported", a.k.Np);
    return SR_EUNSUPPORTED;
}

// the systems of the reference's experiments (pendulum 2 + 1, cart-pole 4 + 1) and their neighbours; anything else
// runs the per-step launches
// Every instantiation is scratch-free (profiles/r03_kernel_resources.txt: 147 .. 254 VGPRs, no spills) since the
// ellipsoid step moved to the group's tail workgroup: the posterior workgroups hold their U^-1 fragments (36 .. 132
// registers of the 256 per lane) without the live ranges of sr_ellipsoid_one beside them, and the tail workgroup holds no
// fragments.  (Round 2 / early round 3: one body did both -- up to 328 B of scratch per lane, and the dispatcher had
// to leave n_s = 4 at Np >= 384 and n_s >= 3 at Np = 512 to the per-step launches.)
static bool sr_chain_dispatched(int Np, int n_s, int n_u) {
    (void)Np; (void)n_s; (void)n_u;
    return true;
}

int sr_chain_wgs_per_group(int Np, int n_s) { return n_s * (Np / 128) + ((n_s > 1 || Np > 128) ? 1 : 0); }
long sr_chain_xels_per_group(int Np, int n_s, int n_u, int H) {
    return (long)H * n_s * (SR_FQ * (n_s + n_u + 1) + SR_FQ * (Np / 128));
}

bool sr_chain_supported(int Np, int D, int n_s, int n_u, int H) {
    if (!(Np % 128 == 0 && Np <= SR_FUSED_NP && D == n_s + n_u)) return false;
    if ((long)H * (n_u + n_u * n_s) * SR_FQ * 8 > 24576) return false;      // the group's control sequence lives in LDS
    if (!((n_u == 1 && n_s >= 1 && n_s <= 4) || (n_u == 2 && (n_s == 2 || n_s == 3)))) return false;
    return sr_chain_dispatched(Np, n_s, n_u);
}

int sr_launch_chain(const sr_chain_args& a, hipStream_t s) {
    const int n_s = a.k.n_out, n_u = a.k.D - a.k.n_out;
    SR_CHECK((a.T + SR_FQ - 1) / SR_FQ * sr_chain_wgs_per_group(a.k.Np, n_s) <= SR_CHAIN_GROUPS, SR_EINVAL,
             "chain: %ld rollouts x %d outputs x %d parts do not fit one launch", a.T, n_s, a.k.Np / 128);
    SR_CHECK((a.T + SR_FQ - 1) / SR_FQ * sr_chain_xels_per_group(a.k.Np, n_s, n_u, a.H) <= (long)SR_CHAIN_XELS, SR_EINVAL,
             "chain: exchange buffer too small for %ld rollouts x %d steps", a.T, a.H);
    if (n_u == 1) {
        if (n_s == 1) return launch_chain_su<1, 1>(a, s);
        if (n_s == 2) return launch_chain_su<2, 1>(a, s);
        if (n_s == 3) return launch_chain_su<3, 1>(a, s);
        if (n_s == 4) return launch_chain_su<4, 1>(a, s);
    } else if (n_u == 2) {
        if (n_s == 2) return launch_chain_su<2, 2>(a, s);
        if (n_s == 3) return launch_chain_su<3, 2>(a, s);
    }
    sr_set_error("chain: n_s=%d n_u=%d not instantiated", n_s, n_u);
    return SR_EUNSUPPORTED;
}

// General kernel family (Matern-5/2, linear x stationary + linear: sr_common.h; the kernels of the reference's
// journal experiments) through the same one-launch pass.  k = c v kappa + l with c = c0 + sum a x z, l = sum b x z:
//   dk/dx_j = a_j z_j v kappa + c v g s_j^2 (x_j - z_j) + b_j z_j ,  g = kappa'(r)/r
// so the mean-Jacobian needs four products against M[i] = alpha_i [1, z_i] instead of one:
//   Rk = k^T M (mu), Rq = kappa^T M, Rg = (c g)^T M, R1 = 1^T M:
//   d mu/dx_j = a_j v Rq[t][1+j] + v s_j^2 (x_tj Rg[t][0] - Rg[t][1+j]) + b_j R1[t][1+j]
// and the prior variance is k(x,x) = (c0 + sum a x^2) v + sum b x^2.
template <int NP, int DT>
__global__ __launch_bounds__(1024) void sr_gp_small_general_kernel(sr_kstar_args a, const double* __restrict__ Wt,
                                                                   double* __restrict__ mu,
                                                                   double* __restrict__ var,
                                                                   double* __restrict__ jac) {
    constexpr int NSTRIP = NP / 16, NPAIR = NSTRIP / 2;
    constexpr int NSPLIT = (16 / NPAIR) > 0 ? 16 / NPAIR : 1;
    constexpr int RPW = NP / 16, KSA = RPW / 4;
    __shared__ double ks[NP][SR_FQ];
    __shared__ double xq[SR_FQ][DT];                     // queries of this tile (unscaled)
    __shared__ double pA[16][256];
    __shared__ double Rs[4][SR_FQ][16];                  // Rk, Rq, Rg, R1
    __shared__ double pB[(NSPLIT > 1 ? NSPLIT - 1 : 1) * (NSPLIT > 1 ? NSTRIP : 1) * 256];
    __shared__ double redC[NSTRIP][SR_FQ];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lk = lane >> 4, ln = lane & 15;
    const int d = blockIdx.y;
    const long t0 = (long)blockIdx.x * SR_FQ;
    const int off = NP - a.N;
    const double* kp = a.kp + (long)d * SR_KP(a.D);
    const int kind = (int)kp[0];
    const double vv = kp[1], c0 = kp[2];
    const bool live = t0 + ln < a.T;

    // per lane: the query; the kernel parameters s_j^2, a_j, b_j are wavefront-uniform
    double x[DT];
    double s2[DT], av[DT], bv[DT];
#pragma unroll
    for (int j = 0; j < DT; ++j) {
        x[j] = 0.0;
        if (live && j < a.D) x[j] = (j < a.na) ? a.xa[(t0 + ln) * a.lda + j] : a.xb[(t0 + ln) * a.ldb + (j - a.na)];
        const double sj = (j < a.D) ? kp[3 + j] : 0.0;
        s2[j] = sj * sj;
        av[j] = (j < a.D) ? kp[3 + a.D + j] : 0.0;
        bv[j] = (j < a.D) ? kp[3 + 2 * a.D + j] : 0.0;
    }
    sr_d4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = sr_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int st = 0; st < KSA; ++st) {
        const int i = wave * RPW + 4 * st + lk;
        const bool valid = i >= off;
        const double al = valid ? a.alpha[(long)d * NP + i] : 0.0;
        double r2 = 0.0, la = 0.0, lb = 0.0, bfrag = (ln == 0) ? al : 0.0;
#pragma unroll
        for (int j = 0; j < DT; ++j) {
            const double z = (valid && j < a.D) ? a.Z[(long)(i - off) * a.D + j] : 0.0;
            const double df = x[j] - z;
            r2 = fma(df * s2[j], df, r2);
            la = fma(av[j] * x[j], z, la);
            lb = fma(bv[j] * x[j], z, lb);
            if (ln == j + 1) bfrag = al * z;
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
        const bool on = valid && live;
        const double cc = c0 + la;
        const double k = on ? fma(cc * vv, kap, lb) : 0.0;
        ks[i][ln] = k;
        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(k, bfrag, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(on ? kap : 0.0, bfrag, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(on ? cc * g : 0.0, bfrag, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(on ? 1.0 : 0.0, bfrag, acc[3], 0, 0, 0);
    }
    if (wave == 0 && lk == 0) {
#pragma unroll
        for (int j = 0; j < DT; ++j) xq[ln][j] = x[j];
    }
    // the four partial products are summed over the wavefronts one after the other through the same buffer
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int r = 0; r < 4; ++r) pA[wave][r * 64 + lane] = acc[q][r];
        __syncthreads();
        if (tid < 256) {
            double v = 0.0;
#pragma unroll
            for (int w = 0; w < 16; ++w) v += pA[w][tid];
            const int l2 = tid & 63, r = tid >> 6;
            Rs[q][(l2 >> 4) + 4 * r][l2 & 15] = v;
        }
        __syncthreads();
    }

    sr_small_contract<NP, false>(Wt + (long)d * NP * NP, ks, pB, redC, wave, lane);

    if (tid < SR_FQ * (DT + 1)) {
        const int t = tid / (DT + 1), j = tid % (DT + 1);
        if (t0 + t < a.T) {
            if (j == DT) {
                mu[(t0 + t) * a.n_out + d] = Rs[0][t][0];
            } else if (jac && j < a.D) {
                const double sj = kp[3 + j], aj = kp[3 + a.D + j], bj = kp[3 + 2 * a.D + j];
                jac[((t0 + t) * a.n_out + d) * a.D + j] =
                    aj * vv * Rs[1][t][1 + j] + vv * sj * sj * (xq[t][j] * Rs[2][t][0] - Rs[2][t][1 + j]) +
                    bj * Rs[3][t][1 + j];
            }
        }
    }
    if (tid < SR_FQ && t0 + tid < a.T) {
        double qn = 0.0;
#pragma unroll
        for (int sidx = 0; sidx < NSTRIP; ++sidx) qn += redC[sidx][tid];
        double kxx = c0 * vv;
        for (int j = 0; j < a.D; ++j) {
            const double xv = xq[tid][j];
            kxx = fma((kp[3 + a.D + j] * vv + kp[3 + 2 * a.D + j]) * xv, xv, kxx);
        }
        double v = kxx - qn;
        if (!(v > SR_VAR_CLIP)) v = SR_VAR_CLIP;
        var[(t0 + tid) * a.n_out + d] = v;
    }
}


template <int NP>
static int launch_small_np(const sr_kstar_args& a, const double* Wt, double* mu, double* var, double* jac,
                           double* jac_var, double* hess, hipStream_t s) {
    if (hess) {                                            // single query with second-order outputs
        dim3 grid(1, a.n_out);
#define SR_SMALL_LIN(DT) hipLaunchKernelGGL((sr_gp_small_kernel<NP, DT, true>), grid, dim3(1024), 0, s, a, Wt, mu, var, jac, jac_var, hess)
        if (a.D <= 3) SR_SMALL_LIN(3);
        else if (a.D <= 5) SR_SMALL_LIN(5);
        else SR_SMALL_LIN(8);
#undef SR_SMALL_LIN
        SR_HIP(hipGetLastError());
        return SR_OK;
    }
    dim3 grid((unsigned)((a.T + SR_FQ - 1) / SR_FQ), a.n_out);
    if (a.kp) {                                            // general kernel family
#define SR_SMALL_GEN(DT) hipLaunchKernelGGL((sr_gp_small_general_kernel<NP, DT>), grid, dim3(1024), 0, s, a, Wt, mu, var, jac)
        if (a.D <= 3) SR_SMALL_GEN(3);
        else if (a.D <= 5) SR_SMALL_GEN(5);
        else SR_SMALL_GEN(8);
#undef SR_SMALL_GEN
        SR_HIP(hipGetLastError());
        return SR_OK;
    }
#define SR_SMALL_CASE(DT) hipLaunchKernelGGL((sr_gp_small_kernel<NP, DT, false>), grid, dim3(1024), 0, s, a, Wt, mu, var, jac, nullptr, nullptr)
    if (a.D <= 3) SR_SMALL_CASE(3);
    else if (a.D <= 5) SR_SMALL_CASE(5);
    else SR_SMALL_CASE(8);
#undef SR_SMALL_CASE
    SR_HIP(hipGetLastError());
    return SR_OK;
}

int sr_launch_gp_small(const sr_kstar_args& a, const double* Wt, double* mu, double* var, double* jac,
                       hipStream_t s) {
    if (a.Np == 128) return launch_small_np<128>(a, Wt, mu, var, jac, nullptr, nullptr, s);
    if (a.Np == 256) return launch_small_np<256>(a, Wt, mu, var, jac, nullptr, nullptr, s);
    if (a.Np == 384) return launch_small_np<384>(a, Wt, mu, var, jac, nullptr, nullptr, s);
    if (a.Np == 512) return launch_small_np<512>(a, Wt, mu, var, jac, nullptr, nullptr, s);
    sr_set_error("gp_small: Np=%d not supported", a.Np);
    return SR_EUNSUPPORTED;
}

int sr_launch_gp_small_lin(const sr_kstar_args& a, const double* Wt, double* mu, double* var, double* jac_mu,
                           double* jac_var, double* hess_mu, hipStream_t s) {
    if (a.Np == 128) return launch_small_np<128>(a, Wt, mu, var, jac_mu, jac_var, hess_mu, s);
    if (a.Np == 256) return launch_small_np<256>(a, Wt, mu, var, jac_mu, jac_var, hess_mu, s);
    if (a.Np == 384) return launch_small_np<384>(a, Wt, mu, var, jac_mu, jac_var, hess_mu, s);
    if (a.Np == 512) return launch_small_np<512>(a, Wt, mu, var, jac_mu, jac_var, hess_mu, s);
    sr_set_error("gp_small: Np=%d not supported", a.Np);
    return SR_EUNSUPPORTED;
}
