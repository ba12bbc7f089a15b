// sr_ellipsoid_dev.h -- the per-query ellipsoid step as a device function (shared by sr_ellipsoid.hip and the
// persistent multi-step kernel of sr_chain.hip).
//
// replaces the algebra of
//   /root/reference/safe_exploration/gp_reachability.py:65-88   (point branch)
//   /root/reference/safe_exploration/gp_reachability.py:89-156  (ellipsoid branch)
//   /root/reference/safe_exploration/utils.py:108-144           (compute_remainder_overapproximations)
//   /root/reference/safe_exploration/utils_ellipsoid.py:63-94
#pragma once
#include "sr_common.h"

// largest eigenvalue of Q * B, Q symmetric PSD, B = I + K^T K SPD.
// eig(Q B) == eig(L^T Q L) with B = L L^T (similarity by L^T); the latter is symmetric, so a cyclic
// Jacobi iteration gives all eigenvalues to full fp64 accuracy.  The reference calls
// scipy.linalg.eig on the non-symmetric product and keeps np.max (utils.py:133-134).
template <int NS, int NU>
__device__ __forceinline__ double sr_lambda_max_qb(const double (&q)[NS][NS],
                                                   const double (&kfb)[NU][NS]) {
    // B[i][j] = delta_ij + sum_u K[u][i] K[u][j], formed where the factorisation consumes it (never stored)
    auto Bij = [&](int i, int j) {
        double s = (i == j) ? 1.0 : 0.0;
#pragma unroll
        for (int u = 0; u < NU; ++u) s = fma(kfb[u][i], kfb[u][j], s);
        return s;
    };
    if (NS == 1) return q[0][0] * Bij(0, 0);
    // lower Cholesky of B
    double L[NS][NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        double s = Bij(j, j);
#pragma unroll
        for (int k = 0; k < j; ++k) s -= L[j][k] * L[j][k];
        const double ljj = sqrt(s);
        L[j][j] = ljj;
        const double inv = 1.0 / ljj;
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            if (i > j) {
                double v = Bij(i, j);
#pragma unroll
                for (int k = 0; k < j; ++k) v -= L[i][k] * L[j][k];
                L[i][j] = v * inv;
            } else if (i < j) {
                L[i][j] = 0.0;
            }
        }
    }
    // M = L^T Q L, upper triangle only (M[i][j], j >= i: the entries below the diagonal are never formed -- half the
    // registers and half the work of the rotations below; at n_s = 8 the full matrix beside Q, L, Q L and the caller's
    // H Q H^T was what spilled)
    double QL[NS][NS], M[NS][NS];
#pragma unroll
    for (int i = 0; i < NS; ++i)
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < NS; ++k)
                if (k >= j) s = fma(q[i][k], L[k][j], s);
            QL[i][j] = s;
        }
#pragma unroll
    for (int i = 0; i < NS; ++i)
#pragma unroll
        for (int j = 0; j < NS; ++j)
            if (j >= i) {
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < NS; ++k)
                    if (k >= i) s = fma(L[k][i], QL[k][j], s);
                M[i][j] = s;
            }
    if (NS == 2) {
        const double tr = M[0][0] + M[1][1];
        const double df = M[0][0] - M[1][1];
        return 0.5 * (tr + sqrt(fma(df, df, 4.0 * M[0][1] * M[0][1])));
    }
    // cyclic Jacobi (eigenvalues only) on the upper triangle.  Stops when the squared off-diagonal norm is below 1e-26 of
    // the squared diagonal norm: the eigenvalue error is bounded by off^2 / (2 gap), i.e. far below one ulp unless two
    // eigenvalues agree to 13 digits -- and then by their difference.  One rotation costs one sqrt, one division and one
    // rsqrt:   t = sgn(d) 2 a / (|d| + sqrt(d^2 + 4 a^2)),  d = M_rr - M_pp, a = M_pr;   c = 1 / sqrt(1 + t^2),  s = t c,
    //          M_pp -= t a,  M_rr += t a,  M_pr = 0,  (M_kp, M_kr) <- (c M_kp - s M_kr, s M_kp + c M_kr)  for k != p, r
    // (the textbook form through theta = d / 2a takes two square roots and three divisions; on one lane per query the
    // rotations are the whole cost of the step for n_s >= 3).
#define SR_SYM(i, j) M[(i) < (j) ? (i) : (j)][(i) < (j) ? (j) : (i)]
    for (int sweep = 0; sweep < 24; ++sweep) {
        double off = 0.0, dia = 0.0;
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            dia = fma(M[i][i], M[i][i], dia);
#pragma unroll
            for (int j = 0; j < NS; ++j)
                if (j > i) off = fma(M[i][j], M[i][j], off);
        }
        if (off <= 1e-26 * dia || off == 0.0) break;
#pragma unroll
        for (int p = 0; p < NS; ++p)
#pragma unroll
            for (int r = 0; r < NS; ++r) {
                if (r > p) {
                    const double apr = M[p][r];
                    if (apr != 0.0) {
                        const double dd = M[r][r] - M[p][p];
                        const double two_a = 2.0 * apr;
                        const double den = fabs(dd) + sqrt(fma(dd, dd, two_a * two_a));
                        const double tt = ((dd >= 0.0) ? two_a : -two_a) / den;
                        const double c = rsqrt(fma(tt, tt, 1.0));
                        const double s = tt * c;
                        M[p][p] = fma(-tt, apr, M[p][p]);
                        M[r][r] = fma(tt, apr, M[r][r]);
                        M[p][r] = 0.0;
#pragma unroll
                        for (int k = 0; k < NS; ++k) {
                            if (k != p && k != r) {
                                const double mkp = SR_SYM(k, p), mkr = SR_SYM(k, r);
                                SR_SYM(k, p) = c * mkp - s * mkr;
                                SR_SYM(k, r) = s * mkp + c * mkr;
                            }
                        }
                    }
                }
            }
    }
#undef SR_SYM
    double lam = M[0][0];
#pragma unroll
    for (int i = 1; i < NS; ++i) lam = fmax(lam, M[i][i]);
    return lam;
}

// One ellipsoid (or Gaussian-moment) step of query t; every pointer of `a` may be global or LDS (flat addressing):
// the persistent chain kernel of sr_chain.hip keeps the state of its 16 rollouts in LDS between steps.
// p / q are read completely before p_out / q_out are written, so the step may run in place.
template <int NS, int NU>
__device__ __forceinline__ void sr_ellipsoid_one(const sr_ell_args& a, long t) {
    int* n_bad = a.n_bad;
    constexpr int D = NS + NU;

    double p[NS], u[NU], mu[NS], var[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        p[i] = a.p[t * a.ldp + i];
        mu[i] = a.mu[t * NS + i];
        var[i] = a.var[t * NS + i];
    }
#pragma unroll
    for (int k = 0; k < NU; ++k) u[k] = a.k_ff[t * a.ldkff + k];

    // p_lin = a p + b u + mu        (gp_reachability.py:82-83 / :115)
    double p1[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        double s = mu[i];
#pragma unroll
        for (int j = 0; j < NS; ++j) s = fma(a.a[i * NS + j], p[j], s);
#pragma unroll
        for (int k = 0; k < NU; ++k) s = fma(a.b[i * NU + k], u[k], s);
        p1[i] = s;
        a.p_out[t * a.ldpo + i] = s;
    }
    double* qo = a.q_out + t * a.ldqo;
    bool bad = false;

    if (a.q == nullptr) {
        // point branch: Q1 = diag(n_s (c sqrt(var))^2)     (gp_reachability.py:78-80)
        // moment modes: Sigma1 = diag(var)                  (uncertainty_propagation_casadi.py:50-55, 253-258)
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const double ub = a.c_safety * sqrt(var[i]);
            bad |= !(ub > 0.0);
            const double dv = (a.mode == 0) ? NS * ub * ub : var[i];
#pragma unroll
            for (int j = 0; j < NS; ++j) qo[i * NS + j] = (i == j) ? dv : 0.0;
        }
        if (bad && n_bad) atomicAdd(n_bad, 1);
        return;
    }

    double q[NS][NS], H[NS][NS];
#pragma unroll
    for (int i = 0; i < NS; ++i)
#pragma unroll
        for (int j = 0; j < NS; ++j) q[i][j] = a.q[t * a.ldq + i * NS + j];

    // remainder radius first: lambda_max(Q (I + K^T K)) needs Q, L, Q L and M at once -- computed before H Q H^T exists,
    // the step stays in registers up to n_s = 7 (n_s = 8: see profiles/archive/r03_kernel_resources.txt)   (:125-137, utils.py:129-142)
    // (the feedback matrix is fetched twice -- here and for H below -- rather than held across the eigenvalue iteration)
    double r2 = 0.0;
    if (a.mode == 0) {
        double kfb0[NU][NS];
#pragma unroll
        for (int k = 0; k < NU; ++k)
#pragma unroll
            for (int j = 0; j < NS; ++j) kfb0[k][j] = a.k_fb[t * a.ldkfb + k * NS + j];
        r2 = sr_lambda_max_qb<NS, NU>(q, kfb0);
    }
    double kfb[NU][NS];
#pragma unroll
    for (int k = 0; k < NU; ++k)
#pragma unroll
        for (int j = 0; j < NS; ++j) kfb[k][j] = a.k_fb[t * a.ldkfb + k * NS + j];
    // H = a + a_mu + (b_mu + b) k_fb                          (gp_reachability.py:110-114)
    // (mean-equivalent propagation drops the Jacobian terms: H = a + b k_fb)
    const double* jac = a.jac + t * NS * D;
    const double jw = (a.mode == 2) ? 0.0 : 1.0;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        double bm[NU];
#pragma unroll
        for (int k = 0; k < NU; ++k) bm[k] = (a.mode == 2 ? 0.0 : jac[i * D + NS + k]) + a.b[i * NU + k];
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            double s = a.a[i * NS + j] + (a.mode == 2 ? 0.0 : jw * jac[i * D + j]);
#pragma unroll
            for (int k = 0; k < NU; ++k) s = fma(bm[k], kfb[k][j], s);
            H[i][j] = s;
        }
    }
    // Q0 = H Q H^T, one row of H Q at a time                   (:117)
    double Q0[NS][NS];
    double trQ0 = 0.0;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        double hq[NS];
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < NS; ++k) s = fma(H[i][k], q[k][j], s);
            hq[j] = s;
        }
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < NS; ++k) s = fma(hq[k], H[j][k], s);
            Q0[i][j] = s;
            if (i == j) trQ0 += s;
        }
    }
    if (a.mode != 0) {
        // Gaussian moment propagation: [a b I] Sigma_all [a b I]^T collapses to H Sigma H^T + diag(var)
        // (uncertainty_propagation_casadi.py:57-87 with the Jacobian cross terms, :260-283 without)
#pragma unroll
        for (int i = 0; i < NS; ++i)
#pragma unroll
            for (int j = 0; j < NS; ++j) qo[i * NS + j] = Q0[i][j] + ((i == j) ? var[i] : 0.0);
        return;
    }
    // remainder boxes                                          (:125-137, utils.py:129-142)
    const double r1 = sqrt(r2);
    double dL[NS], dM[NS];
    double trS = 0.0, trM = 0.0;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const double ub_mu = a.l_mu[i] * r2;
        const double ub_sg = a.c_safety * (sqrt(var[i]) + a.l_sigma[i] * r1);
        bad |= !(ub_mu > 0.0) || !(ub_sg > 0.0);
        const double ds = NS * ub_sg * ub_sg;
        const double dm = NS * ub_mu * ub_mu;
        trS += ds;
        trM += dm;
        dL[i] = ds;
        dM[i] = dm;
    }
    // trace-optimal sums                                        (:143-148, utils_ellipsoid.py:88-92)
    const double c1 = sqrt(trS / trM);
    double trL = 0.0;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        dL[i] = (1.0 + 1.0 / c1) * dL[i] + (1.0 + c1) * dM[i];
        trL += dL[i];
    }
    const double c2 = sqrt(trL / trQ0);
#pragma unroll
    for (int i = 0; i < NS; ++i)
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            double v = (1.0 + c2) * Q0[i][j];
            if (i == j) v += (1.0 + 1.0 / c2) * dL[i];
            qo[i * NS + j] = v;
        }
    if (bad && n_bad) atomicAdd(n_bad, 1);
    (void)p1;
}

