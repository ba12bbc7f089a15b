// sr_server.hip -- K0s: the RESIDENT single-query server of small models (host side: sr_capi_server.hip).
#include "sr_small_dev.h"

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

// The mailbox is ONE 64-byte line in pinned host memory:
//   [x0 .. x4 | (launch epoch << 8) | command | sequence number | check word = x0 ^ .. ^ seq ^ SR_SERVER_CHK]
// Lanes 0 .. 7 of the first wavefront fetch it with one request.  A request is taken when the sequence number is the
// expected one AND the eight words xor to SR_SERVER_CHK: a fetch that reached host memory as several transactions and
// caught the line half written (new sequence number, old query) fails the check and is simply repeated -- nothing is
// assumed about the atomicity of the 64-byte read.  An epoch other than this launch's (one 8-byte word: never torn) means
// STOP, whatever request a workgroup is waiting for.  Returns the command (SR_SERVER_CMD_IDLE after idle_ticks without a
// request); on a hit x0 .. x4 are in xreq.
__device__ __forceinline__ unsigned long long sr_srv_poll(const sr_server_args& sv, unsigned long long expect, int lane,
                                                          double* xreq) {
    unsigned long long cmd = SR_SERVER_CMD_IDLE;
    const unsigned long long t_last = wall_clock64();             // 100 MHz
    for (;;) {
        unsigned long long wv = 0;
        if (lane < 8) wv = __hip_atomic_load(sv.mb + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        // (words read into scalar registers lane by lane: the check costs the polling wavefront no vector registers)
        const unsigned wlo = (unsigned)wv, whi = (unsigned)(wv >> 32);
#define SR_MB_WORD(l) (((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)whi, l) << 32) | (unsigned)__builtin_amdgcn_readlane((int)wlo, l))
        const unsigned long long w5 = SR_MB_WORD(5);
        if ((w5 >> 8) != sv.epoch) { cmd = SR_SERVER_CMD_STOP; break; }      // the host called this launch off
        if (SR_MB_WORD(6) == expect) {
            const unsigned long long c = SR_MB_WORD(0) ^ SR_MB_WORD(1) ^ SR_MB_WORD(2) ^ SR_MB_WORD(3) ^ SR_MB_WORD(4) ^ w5 ^
                                         expect ^ SR_MB_WORD(7);
#undef SR_MB_WORD
            if (c == SR_SERVER_CHK) {
                cmd = w5 & 0xffull;
                if (lane < 5) xreq[lane] = __longlong_as_double((long long)wv);
                break;
            }
        }
        if (wall_clock64() - t_last > sv.idle_ticks) break;
        __builtin_amdgcn_s_sleep(2);
    }
    return cmd;
}

// (the model travels as the few words the evaluation needs -- sr_server_model -- and the argument block of phase A is
//  rebuilt from them every round: with the whole sr_kstar_args live across the loop's back edge the scalar registers run
//  out, spill into vector lanes and those into scratch: 12 .. 380 B per lane)
struct sr_server_model { const double *Z, *alpha, *ls, *sf2, *Wt, *kp; int N, D, n_out; };
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
            const unsigned long long cmd = sr_srv_poll(sv, expect, lane, xreq);
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

// GEN: the general kernel family (mat52 / lin_rbf / lin_mat52: sr_small_dev.h, second part) -- the training rows stay
// unscaled, the output's packed parameters sit in LDS, phase A forms two products, the record carries k(x,x) instead of
// sf2 and part 0 folds the prior variance's gradient into its shares.  Also at Np = 128 (two parts).
template <int NP, int DT, bool GEN>
__global__ __launch_bounds__(512) void sr_gp_server_parts_kernel(sr_server_model m, sr_server_args sv) {
    constexpr int NSTRIP = NP / 16, NPAIR = NSTRIP / 2, PARTS = NP / 64;
    static_assert(NPAIR == 2 * PARTS, "two strip pairs per part");
    __shared__ double ks_[NP][SR_FQ];
    __shared__ double xq_[GEN ? 1 : SR_FQ][DT];
    __shared__ double pA_[8][256];
    __shared__ double pA2_[GEN ? 8 : 1][256];
    __shared__ double Rs_[SR_FQ][16];
    __shared__ double Rs2_[GEN ? SR_FQ : 1][16];
    __shared__ double pB_[3 * 4 * 256];        // quarters h = 1 .. 3 of the part's four strips
    __shared__ double redC_[NP / 16][SR_FQ];
    sr_small_lds<NP, DT> L{ks_, xq_, pA_, Rs_, pB_, redC_};
    const sr_gen_lds<NP, DT> LG{L, pA2_, Rs2_};
    __shared__ double rows_[NP][DT + 1];
    __shared__ double il_[DT];
    __shared__ double gp_[3 + 3 * DT];         // GEN: packed parameters of this output
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
        if constexpr (GEN) {
            sr_small_rows_fill_raw<NP, DT>(a0, d, rows_, 512);
            if (tid < SR_KP(D)) gp_[tid] = m.kp[(long)d * SR_KP(D) + tid];
        } else {
            sr_small_rows_fill<NP, DT>(a0, d, rows_, 512);
            sr_small_il_fill<DT>(a0, d, il_);
        }
        frag.load(m.Wt + (long)d * NP * NP, pr, h, lane);
    }
    const double sf2 = GEN ? 0.0 : m.sf2[d];
    __syncthreads();
    const sr_small_rows<NP, DT> rows{rows_, il_};
    for (;;) {
        if (wave == 0) {
            const unsigned long long cmd = sr_srv_poll(sv, expect, lane, xreq);
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
            if constexpr (GEN) sr_small_phase_a_gen<NP, DT, true, 8>(a, d, xreq, gp_, LG, &rows, tq);
            else sr_small_phase_a<NP, DT, true, true, 8>(a, d, xreq, D, xreq, D, 1, L, &rows, tq);
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
            const int c = (e == 1) ? 0 : ((e >= 2 + D && e < 2 + 2 * D) ? e - (2 + D) + 1 : -1);
            if constexpr (GEN) {
                // (the host forms var = rec[REC - 1] - sum of the shares of q_0 and d var/dx_j = -2 sum of the shares of q_j:
                //  part 0 hands k(x,x) over in place of sf2 and takes half the prior variance's gradient off its share)
                if (part == 0) {
                    if (e == SR_SERVER_REC - 1) val = sr_gen_record_elem<NP, DT>(1, D, LG, gp_);
                    else if (e < R && e != 1) val = sr_gen_record_elem<NP, DT>(e, D, LG, gp_);
                }
                if (c >= 0) {
                    const double share = redC_[0][c] + redC_[1][c] + redC_[2][c] + redC_[3][c];
                    val = (c == 0) ? share : fma(-0.5, val, share);
                }
            } else {
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
                if (c >= 0) val = redC_[0][c] + redC_[1][c] + redC_[2][c] + redC_[3][c];
            }
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
    const sr_server_model m{a.Z, a.alpha, a.ls, a.sf2, Wt, a.kp, a.N, a.D, a.n_out};
    SR_CHECK(sr_gp_server_supported(NP, a.D), SR_EUNSUPPORTED, "gp_server: Np=%d D=%d not built", NP, a.D);
    if (a.kp) {                                            // general kernel family: in parts at every size
        static_assert(sr_gp_server_parts(NP, true) == NP / 64, "parts");
        dim3 grid(NP / 64, a.n_out);
        if (a.D <= 3) hipLaunchKernelGGL((sr_gp_server_parts_kernel<NP, 3, true>), grid, dim3(512), 0, s, m, sv);
        else hipLaunchKernelGGL((sr_gp_server_parts_kernel<NP, 5, true>), grid, dim3(512), 0, s, m, sv);
    } else if constexpr (NP >= 256) {
        static_assert(sr_gp_server_parts(NP, false) == NP / 64, "parts");
        dim3 grid(NP / 64, a.n_out);
        if (a.D <= 3) hipLaunchKernelGGL((sr_gp_server_parts_kernel<NP, 3, false>), grid, dim3(512), 0, s, m, sv);
        else hipLaunchKernelGGL((sr_gp_server_parts_kernel<NP, 5, false>), grid, dim3(512), 0, s, m, sv);
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

