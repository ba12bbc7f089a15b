// sr_chain.hip -- K0c: the H-step reachability / moment chain of a small model in ONE persistent launch.
#include "sr_small_dev.h"

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
// Every instantiation is scratch-free (profiles/archive/r03_kernel_resources.txt: 147 .. 254 VGPRs, no spills) since the
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

