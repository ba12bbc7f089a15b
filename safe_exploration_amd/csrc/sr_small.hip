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
#include "sr_small_dev.h"

// blocking single query (sr_gp_call1): the outputs went to pinned host memory; once every workgroup's stores are out
// (system-scope fence), the last one to arrive publishes the sequence number the host spins on
__device__ __forceinline__ void sr_small_publish(const sr_kstar_args& a, int tid) {
    if (!a.host_flag) return;
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

    sr_small_publish(a, tid);
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
        if (live && j < a.D) x[j] = a.xv_on ? a.xv[j] : ((j < a.na) ? a.xa[(t0 + ln) * a.lda + j] : a.xb[(t0 + ln) * a.ldb + (j - a.na)]);
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
    sr_small_publish(a, tid);
}

// The same family, ONE query with second-order outputs (sr_gp_linearize / sr_gp_call1 of a small mat52 / lin_* model: the
// call CasadiSSMEvaluator makes per IPOPT iteration with the journal experiments' kernels, state_space_models.py:384-417):
// phase A of sr_small_dev.h (general form), the contraction of the columns [k*, dk*/dx] with U^-1, outputs in the API layout.
template <int NP, int DT>
__global__ __launch_bounds__(1024) void sr_gp_small_gen_lin_kernel(sr_kstar_args a, const double* __restrict__ Wt,
                                                                   double* __restrict__ mu, double* __restrict__ var,
                                                                   double* __restrict__ jac, double* __restrict__ jac_var,
                                                                   double* __restrict__ hess) {
    constexpr int NSTRIP = NP / 16;
    __shared__ double ks_[NP][SR_FQ];
    __shared__ double xq_[1][DT];
    __shared__ double pA_[16][256];
    __shared__ double pA2_[16][256];
    __shared__ double Rs_[SR_FQ][16];
    __shared__ double Rs2_[SR_FQ][16];
    __shared__ double pB_[((16 / (NP / 32)) > 1 ? (16 / (NP / 32)) - 1 : 1) * ((16 / (NP / 32)) > 1 ? NP / 16 : 1) * 256];
    __shared__ double redC_[NP / 16][SR_FQ];
    const sr_gen_lds<NP, DT> L{{ks_, xq_, pA_, Rs_, pB_, redC_}, pA2_, Rs2_};
    const int tid = threadIdx.x;
    const int d = blockIdx.y, D = a.D;
    const double* kp = a.kp + (long)d * SR_KP(D);
    sr_small_phase_a_gen<NP, DT, false, 16>(a, d, a.xa, kp, L);
    sr_small_contract<NP, true>(Wt + (long)d * NP * NP, ks_, pB_, redC_, tid >> 6, tid & 63);
    const int R = 2 + 2 * D + D * D;
    if (tid < R) {
        double val = sr_gen_record_elem<NP, DT>(tid, D, L, kp);
        const int c = (tid == 1) ? 0 : ((tid >= 2 + D && tid < 2 + 2 * D) ? tid - (2 + D) + 1 : -1);
        if (c >= 0) {
            double qn = 0.0;
#pragma unroll
            for (int sidx = 0; sidx < NSTRIP; ++sidx) qn += redC_[sidx][c];
            if (c == 0) {
                val -= qn;
                if (!(val > SR_VAR_CLIP)) val = SR_VAR_CLIP;
            } else val = fma(-2.0, qn, val);
        }
        if (tid == 0) mu[d] = val;
        else if (tid == 1) var[d] = val;
        else if (tid < 2 + D) jac[d * D + tid - 2] = val;
        else if (tid < 2 + 2 * D) jac_var[d * D + tid - (2 + D)] = val;
        else hess[(long)d * D * D + tid - (2 + 2 * D)] = val;
    }
    sr_small_publish(a, tid);
}


template <int NP>
static int launch_small_np(const sr_kstar_args& a, const double* Wt, double* mu, double* var, double* jac,
                           double* jac_var, double* hess, hipStream_t s) {
    if (hess && a.kp) {                                    // the same for the general kernel family (D <= 5)
        dim3 grid(1, a.n_out);
        SR_CHECK(a.D <= 5, SR_EUNSUPPORTED, "gp_small: second order of a general kernel in one launch needs D <= 5 (D=%d)", a.D);
        if (a.D <= 3) hipLaunchKernelGGL((sr_gp_small_gen_lin_kernel<NP, 3>), grid, dim3(1024), 0, s, a, Wt, mu, var, jac, jac_var, hess);
        else hipLaunchKernelGGL((sr_gp_small_gen_lin_kernel<NP, 5>), grid, dim3(1024), 0, s, a, Wt, mu, var, jac, jac_var, hess);
        SR_HIP(hipGetLastError());
        return SR_OK;
    }
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
