// sr_final_dev.h -- final stages of the posterior as DEVICE functions, shared by the stand-alone kernels
// (sr_finalize_wave_kernel, sr_lin_final_kernel) and by the fused small-batch kernels of sr_stream.hip, whose last
// workgroup runs them in place of a further launch.
#pragma once
#include "sr_common.h"

__device__ __forceinline__ double sr_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// AG: the value was written by ANOTHER workgroup of the same launch with an agent-scope (write-through) store:
// read it with an agent-scope load (served by L2 / memory, never by this CU's L1) -- cdna_hip_programming.md G16.
template <bool AG>
__device__ __forceinline__ double sr_ld(const double* p) {
    if (AG) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}
__device__ __forceinline__ void sr_st_agent(double* p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// one wavefront: query t, output d.  Lanes stride over the partial sums and combine with a butterfly.
template <bool AG = false>
__device__ __forceinline__ void sr_final_query_wave(const sr_final_args& a, long t, int d, int lane) {
    if (a.nsplit <= 64 && a.nrb <= 64 && a.D <= SR_MAX_D) {
        // few partial sums: every lane fetches its share of ALL quantities first (independent loads: one memory
        // latency for the whole query), the butterflies follow
        const bool hs = lane < a.nsplit, hr = lane < a.nrb;
        double m = hs ? sr_ld<AG>(a.mu_part + ((long)lane * a.n_out + d) * a.Tp + t) : 0.0;
        double q = hr ? sr_ld<AG>(a.var_part + ((long)d * a.nrb + lane) * a.Tp + t) : 0.0;
        double g[SR_MAX_D];
#pragma unroll
        for (int j = 0; j < SR_MAX_D; ++j)
            g[j] = (a.jac && hs && j < a.D) ? sr_ld<AG>(a.jac_part + (((long)lane * a.n_out + d) * a.D + j) * a.Tp + t) : 0.0;
        m = sr_wave_sum(m);
        q = sr_wave_sum(q);
        double v = (a.kxx ? a.kxx[(long)d * a.Tp + t] : a.sf2[d]) - q;
        if (!(v > SR_VAR_CLIP)) v = SR_VAR_CLIP;
        if (lane == 0) {
            a.mu[t * a.n_out + d] = m;
            a.var[t * a.n_out + d] = v;
        }
        if (a.jac) {
#pragma unroll
            for (int j = 0; j < SR_MAX_D; ++j)
                if (j < a.D) {
                    const double gj = sr_wave_sum(g[j]);
                    if (lane == 0) a.jac[(t * a.n_out + d) * a.D + j] = gj;
                }
        }
        return;
    }
    // many partial sums (the K* pass splits the training points down to 16 rows per workgroup for small batches: 320
    // partials per quantity at N = 5000): lanes stride over them, EIGHT loads per lane and quantity in flight, mean and
    // variance together, then the Jacobian three components at a time -- a plain strided loop pays one memory latency
    // per partial and quantity (20 dependent round trips = 9 us of the 12 us this stage took at N = 5000).
    double m = 0.0, q = 0.0;
    for (int s0 = 0; s0 < a.nsplit || s0 < a.nrb; s0 += 512) {
        double xm[8], xq[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int s = s0 + lane + 64 * u;
            xm[u] = (s < a.nsplit) ? sr_ld<AG>(a.mu_part + ((long)s * a.n_out + d) * a.Tp + t) : 0.0;
            xq[u] = (s < a.nrb) ? sr_ld<AG>(a.var_part + ((long)d * a.nrb + s) * a.Tp + t) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { m += xm[u]; q += xq[u]; }
    }
    m = sr_wave_sum(m);
    q = sr_wave_sum(q);
    double v = (a.kxx ? a.kxx[(long)d * a.Tp + t] : a.sf2[d]) - q;
    if (!(v > SR_VAR_CLIP)) v = SR_VAR_CLIP;
    if (lane == 0) {
        a.mu[t * a.n_out + d] = m;
        a.var[t * a.n_out + d] = v;
    }
    if (a.jac) {
        for (int j0 = 0; j0 < a.D; j0 += 3) {
            double g[3] = {0.0, 0.0, 0.0};
            for (int s0 = 0; s0 < a.nsplit; s0 += 512) {
                double x[3][8];
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int s = s0 + lane + 64 * u;
                        x[c][u] = (s < a.nsplit && j0 + c < a.D)
                                      ? sr_ld<AG>(a.jac_part + (((long)s * a.n_out + d) * a.D + j0 + c) * a.Tp + t) : 0.0;
                    }
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                    for (int u = 0; u < 8; ++u) g[c] += x[c][u];
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const double gj = sr_wave_sum(g[c]);
                if (lane == 0 && j0 + c < a.D) a.jac[(t * a.n_out + d) * a.D + j0 + c] = gj;
            }
        }
    }
}

// query coordinate j of the single-query entry points: the first na from x, the rest from xb
__device__ __forceinline__ double sr_lin_x(const sr_lin_args& a, int j) {
    return (a.xb && j >= a.na) ? a.xb[j - a.na] : a.x[j];
}

// Final stage of the streamed single-query route for output d, by one workgroup (>= 160 threads; all of them call):
// adds the partial sums [mu, d mu/dx, upper triangle of the Hessian] of the column pass (lin_part: nblk blocks of
// NACC = 1 + DT + DT (DT+1)/2 entries per output, DT the width the column pass was instantiated with) and the dot
// products of the streamed columns with column 0 (dots: ncb partial sums per column), writes the five outputs.
// sh: >= 91 + 13 doubles of shared memory.
template <bool AG = false>
__device__ __forceinline__ void sr_lin_final_dev(const sr_lin_args& a, const double* __restrict__ lin_part, int nblk,
                                                 int DT, const double* __restrict__ dots, int ncb, double* mu,
                                                 double* var, double* jac_mu, int d, int t, double* sh) {
    const int NACC = 1 + DT + DT * (DT + 1) / 2;
    double* tot = sh;
    double* dt = sh + 91;
    __syncthreads();
    if (t < NACC) {
        double v = 0.0;
        for (int b = 0; b < nblk; ++b) v += sr_ld<AG>(lin_part + ((long)d * nblk + b) * NACC + t);
        tot[t] = v;
    }
    if (t >= 128 && t - 128 <= DT) {
        const int c = t - 128;
        double v = 0.0;
        if (c <= a.D)
            for (int cb = 0; cb < ncb; ++cb) v += sr_ld<AG>(dots + ((long)d * ncb + cb) * a.Tp + c);
        dt[c] = v;
    }
    __syncthreads();
    if (t == 0) {
        double kxx, x2a = 0.0;
        if (a.kp == nullptr) kxx = a.sf2[d];
        else {
            const double* kp = a.kp + (long)d * SR_KP(a.D);
            for (int j = 0; j < a.D; ++j)
                x2a = fma((kp[3 + a.D + j] * kp[1] + kp[3 + 2 * a.D + j]) * sr_lin_x(a, j), sr_lin_x(a, j), x2a);
            kxx = kp[2] * kp[1] + x2a;
        }
        double v = kxx - dt[0];
        if (!(v > SR_VAR_CLIP)) v = SR_VAR_CLIP;
        mu[d] = tot[0];
        var[d] = v;
    }
    if (t < a.D) {
        if (jac_mu) jac_mu[d * a.D + t] = tot[1 + t];
        double dkxx = 0.0;
        if (a.kp != nullptr) {
            const double* kp = a.kp + (long)d * SR_KP(a.D);
            dkxx = 2.0 * (kp[3 + a.D + t] * kp[1] + kp[3 + 2 * a.D + t]) * sr_lin_x(a, t);
        }
        if (a.jac_var) a.jac_var[d * a.D + t] = dkxx - 2.0 * dt[1 + t];
    }
    if (a.hess_mu && t < a.D * a.D) {
        const int j = min(t / a.D, t % a.D), c = max(t / a.D, t % a.D);
        // position of (j, c), j <= c, in the DT-wide upper-triangle enumeration
        const int q = 1 + DT + j * DT - j * (j - 1) / 2 + (c - j);
        a.hess_mu[(long)d * a.D * a.D + t] = tot[q];
    }
}

// The same final stage by ONE wavefront (output d), so that several outputs are finished side by side by the last
// workgroup of a fused launch.  shw: >= 120 doubles private to the calling wavefront.  Partial sums are fetched 8 at a
// time (the loads are independent; a plain loop pays one memory latency per partial).
template <bool AG>
__device__ __forceinline__ double sr_sum_strided(const double* p, long stride, int n) {
    double v = 0.0;
    for (int b = 0; b < n; b += 8) {
        double x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = (b + u < n) ? sr_ld<AG>(p + (long)(b + u) * stride) : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) v += x[u];
    }
    return v;
}

template <bool AG = false>
__device__ __forceinline__ void sr_lin_final_wave(const sr_lin_args& a, const double* __restrict__ lin_part, int nblk,
                                                  int DT, const double* __restrict__ dots, int ncb, double* mu,
                                                  double* var, double* jac_mu, int d, int lane, double* shw) {
    const int NACC = 1 + DT + DT * (DT + 1) / 2;
    double* tot = shw;
    double* dt = shw + 91;
    if (nblk <= 64 && ncb <= 64 && NACC <= 28) {
        // lane b holds block b's share of every accumulator: all loads in flight at once, then the butterflies
        double x[28], y[6];
#pragma unroll
        for (int q = 0; q < 28; ++q)
            x[q] = (q < NACC && lane < nblk) ? sr_ld<AG>(lin_part + ((long)d * nblk + lane) * NACC + q) : 0.0;
#pragma unroll
        for (int c = 0; c < 6; ++c)
            y[c] = (c <= a.D && lane < ncb) ? sr_ld<AG>(dots + ((long)d * ncb + lane) * a.Tp + c) : 0.0;
#pragma unroll
        for (int q = 0; q < 28; ++q)
            if (q < NACC) {
                const double v = sr_wave_sum(x[q]);
                if (lane == 0) tot[q] = v;
            }
#pragma unroll
        for (int c = 0; c < 6; ++c)
            if (c <= DT) {
                const double v = sr_wave_sum(y[c]);
                if (lane == 0) dt[c] = v;
            }
    } else {
        for (int q = lane; q < NACC; q += 64) tot[q] = sr_sum_strided<AG>(lin_part + (long)d * nblk * NACC + q, NACC, nblk);
        if (lane <= DT) dt[lane] = (lane <= a.D) ? sr_sum_strided<AG>(dots + (long)d * ncb * a.Tp + lane, a.Tp, ncb) : 0.0;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");      // LDS writes above, reads by other lanes below
    if (lane == 0) {
        double kxx, x2a = 0.0;
        if (a.kp == nullptr) kxx = a.sf2[d];
        else {
            const double* kp = a.kp + (long)d * SR_KP(a.D);
            for (int j = 0; j < a.D; ++j)
                x2a = fma((kp[3 + a.D + j] * kp[1] + kp[3 + 2 * a.D + j]) * sr_lin_x(a, j), sr_lin_x(a, j), x2a);
            kxx = kp[2] * kp[1] + x2a;
        }
        double v = kxx - dt[0];
        if (!(v > SR_VAR_CLIP)) v = SR_VAR_CLIP;
        mu[d] = tot[0];
        var[d] = v;
    }
    if (lane < a.D) {
        if (jac_mu) jac_mu[d * a.D + lane] = tot[1 + lane];
        double dkxx = 0.0;
        if (a.kp != nullptr) {
            const double* kp = a.kp + (long)d * SR_KP(a.D);
            dkxx = 2.0 * (kp[3 + a.D + lane] * kp[1] + kp[3 + 2 * a.D + lane]) * sr_lin_x(a, lane);
        }
        if (a.jac_var) a.jac_var[d * a.D + lane] = dkxx - 2.0 * dt[1 + lane];
    }
    if (a.hess_mu)
        for (int t = lane; t < a.D * a.D; t += 64) {
            const int j = min(t / a.D, t % a.D), c = max(t / a.D, t % a.D);
            const int q = 1 + DT + j * DT - j * (j - 1) / 2 + (c - j);
            a.hess_mu[(long)d * a.D * a.D + t] = tot[q];
        }
}
