// sr_stream.hip -- the small-batch posterior in as few launches as the data flow allows (the
// CasADi / IPOPT callback evaluates ONE query per iteration, state_space_models.py:278-303, 384-417; candidate
// batches of the MPC are a few dozen queries).  At N = 5000 the floor of such a call is the 210 MB of U^-1 that
// have to be streamed once (26 us at 8 TB/s); the first version spent as long again in three auxiliary launches
// (K* pass 8.7 + reduce 6.5 + finalize 9.8 us of mostly fixed cost).
//
//   sr_stream1_kernel<TQ, SRC, DT>  up to 4 columns, ONE launch: every workgroup owns a (256-column block, 128-row
//       chunk) pair of U^-1 as before, but (SRC 1, 2) evaluates its chunk of the ARD-RBF cross-covariance columns
//       itself -- [k*] per query, or [k*, dk*/dx_1..D] of the single query in linearize mode -- instead of reading
//       them from a pass of their own, stores its partial sums, and takes a ticket: the LAST workgroup of a column
//       block adds that block's partials (fixed order: deterministic), squares and reduces; the last column block
//       runs the final stage (mean, variance, Jacobian[, d var/dx, Hessian]).  SRC 0 takes the columns from the
//       workspace (general kernel family, D > 5).
//   sr_stream_mfma_kernel<G>        5 .. 128 columns on the MFMA 16x16x4 tile: 16 G columns per workgroup, U^-1
//       streamed ONCE for all of them (the first version ran one workgroup per group of 16 queries, each re-reading
//       U^-1 from L2 / Infinity Cache, and fell back to split-K tiles at 17 queries: 70 -> 165 us).
//   sr_stream_reduce_kernel         their reduction, one workgroup per (column block, output, column), fused with the
//       final stage through a ticket per query.
//
// Inter-workgroup hand-off (cdna_hip_programming.md, Guideline 16): everything one workgroup hands to another goes out
// with agent-scope (write-through) stores and comes in with agent-scope loads; every storing wave drains
// (s_waitcnt vmcnt(0)) -> __syncthreads -> lane 0: relaxed agent atomic on the ticket -> __syncthreads.  No release fence: an L2 write-back per workgroup (buffer_wbl2) costs microseconds
// and serialises per XCD (first version: 243 us for the 5120 workgroups of the T = 128 reduction).  Tickets are
// zeroed at allocation and reset by the last arriver.
#include "sr_mfma_tile.h"
#include "sr_final_dev.h"
#include <algorithm>
#include <vector>

#define SR_ST_ROWS 128
#define SR_ST_COLS 256

// pair index p -> column block cb, k-chunk j <= 2 cb + 1
__device__ __forceinline__ void sr_pair_decode(int p, int& cb, int& j) {
    cb = (int)((sqrt(4.0 * p + 1.0) - 1.0) * 0.5);
    while ((cb + 1) * (cb + 2) <= p) ++cb;
    while (cb * (cb + 1) > p) --cb;
    j = p - cb * (cb + 1);
}

// returns true in exactly one workgroup: the one whose arrival completes `expected` on *ticket (which it resets).
// All threads of the workgroup must call; on return in the elected workgroup every other arriver's stores are visible.
__device__ __forceinline__ bool sr_ticket_last(unsigned* ticket, unsigned expected, int* s_flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = (old == expected - 1u);
        // (no acquire fence: whatever the elected workgroup reads of the others' results it reads with agent-scope
        //  loads, which never hit this CU's L1 -- the fence costs 1.7 us on the tail of the launch, twice)
        if (last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *s_flag = last;
    }
    __syncthreads();
    return *s_flag != 0;
}

// sum of one double per thread over the workgroup (256 or 1024 threads); result valid in thread 0
__device__ __forceinline__ double sr_block_sum(double v, double* red) {
    v = sr_wave_sum(v);
    const int nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
    if (threadIdx.x == 0)
        for (int w = 0; w < nw; ++w) s += red[w];
    return s;
}

// the final stage, by the elected last workgroup (any block size that is a multiple of 64, >= 256)
__device__ __forceinline__ void sr_stream_final(const sr_stream_args& a, double* sh) {
    if (a.mode == 0) {
        const int lane = threadIdx.x & 63, nw = blockDim.x >> 6;
        for (long q = threadIdx.x >> 6; q < a.fa.T * a.fa.n_out; q += nw)
            sr_final_query_wave<true>(a.fa, q / a.fa.n_out, (int)(q % a.fa.n_out), lane);
    } else {
        // one wavefront per output (the first four wavefronts: sh holds 4 x 120 doubles)
        const int wave = threadIdx.x >> 6;
        if (wave < 4)
            for (int d = wave; d < a.n_out; d += 4)
                sr_lin_final_wave<true>(a.la, a.lin_part, a.nblk, a.lin_dt, a.part, a.ncb, a.lmu, a.lvar, a.ljac_mu, d,
                                        threadIdx.x & 63, sh + wave * 120);
    }
    if (a.host_flag) {                                   // (workgroup-uniform; every thread of the workgroup is here)
        __threadfence_system();                          // this thread's outputs, in pinned host memory, before the flag
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(a.host_flag, a.host_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ------------------------------------------------------------------------------------------------
// <= 4 columns, one launch.  256 threads: thread = one column i of U^-1 inside the column block, 16 independent row
// loads in flight.  The first batch of rows is requested BEFORE the columns are evaluated: the prologue hides behind
// its latency.  (Two adjacent columns per thread with 16-byte loads measured the same at one column and 25 % slower
// at four: 66 against 52 us at N = 5000.)
// ------------------------------------------------------------------------------------------------
#define SR_ST1_THREADS 256
template <int TQ, int SRC, int DT>
__global__ __launch_bounds__(SR_ST1_THREADS) void sr_stream1_kernel(sr_stream_args a) {
    constexpr int NH = DT * (DT + 1) / 2;
    constexpr int NACC = (SRC == 2) ? 1 + DT + NH : TQ * (1 + DT);
    __shared__ double ks[SR_ST_ROWS][TQ];
    __shared__ double sh[4 * 120];
    __shared__ int s_flag;
    // pairs in DESCENDING order: the column blocks with many chunks (expensive reductions) are dispatched first and
    // their reductions overlap with the streaming of the rest; the tail of the launch reduces the short ones
    const int d = blockIdx.y, p = (int)gridDim.x - 1 - (int)blockIdx.x;
    int cb, j;
    sr_pair_decode(p, cb, j);
    const int k0 = j * SR_ST_ROWS;
    const int tid = threadIdx.x;

    // ---- this thread's share of U^-1: column i, rows k0 + r0 .. k0 + kmax (k <= i)
    const int i = cb * SR_ST_COLS + tid;
    const double* w = a.Wt + (long)d * a.Np * a.Np + (long)k0 * a.Np + i;
    const int kmax = (i < a.Np) ? min(SR_ST_ROWS - 1, i - k0) : -1;
    const int r0 = max(0, a.k_lo - k0);                    // leading padding rows carry zeros
    double wv[16];
    const bool first16 = r0 + 15 <= kmax;
    if (first16) {
#pragma unroll
        for (int u = 0; u < 16; ++u) wv[u] = w[(long)(r0 + u) * a.Np];
    }

    if (SRC == 0) {
        const double* ksrc = a.Ks + (long)d * a.Np * a.Tp + (long)k0 * a.Tp;
        for (int e = tid; e < SR_ST_ROWS * TQ; e += SR_ST1_THREADS) {
            const int r = e / TQ, t = e % TQ;
            ks[r][t] = (k0 + r < a.Np) ? ksrc[(long)r * a.Tp + t] : 0.0;
        }
    } else {
        // this chunk of the ARD-RBF columns, evaluated here: threads 0 .. 127, one training row each.  The workgroup
        // with the smallest column block of the chunk (cb == j / 2) also owns the chunk's share of the sums over the
        // training points (mean, mean-Jacobian [, Hessian]): slot j of the N-split partials.
        const bool owner = (cb == (j >> 1));
        const int off = a.Np - a.N;
        double acc[NACC];
#pragma unroll
        for (int q = 0; q < NACC; ++q) acc[q] = 0.0;
        if (tid < SR_ST_ROWS) {
            const int ip = k0 + tid, it = ip - off;
            const bool valid = ip < a.Np && it >= 0;
            double col[TQ];
#pragma unroll
            for (int t = 0; t < TQ; ++t) col[t] = 0.0;
            if (valid) {
                const double wgt = a.alpha[(long)d * a.Np + ip];
                const double sf2 = a.sf2[d];
                double z[DT], il2[DT];
#pragma unroll
                for (int c = 0; c < DT; ++c) {
                    const double l = (c < a.D) ? a.ls[d * a.D + c] : 1.0;
                    il2[c] = (c < a.D) ? 1.0 / (l * l) : 0.0;
                    z[c] = (c < a.D) ? a.Z[(long)it * a.D + c] : 0.0;
                }
                if (SRC == 1) {
#pragma unroll
                    for (int t = 0; t < TQ; ++t) {
                        if (t < a.ncols) {
                            double u[DT], r2 = 0.0;
#pragma unroll
                            for (int c = 0; c < DT; ++c) {
                                double x = 0.0;
                                if (c < a.D) x = (c < a.na) ? a.xa[(long)t * a.lda + c] : a.xb[(long)t * a.ldb + (c - a.na)];
                                u[c] = (z[c] - x) * il2[c];
                                r2 = fma(u[c], z[c] - x, r2);
                            }
                            const double k = sf2 * exp(-0.5 * r2);
                            col[t] = k;
                            acc[t * (1 + DT)] = wgt * k;
#pragma unroll
                            for (int c = 0; c < DT; ++c) acc[t * (1 + DT) + 1 + c] = wgt * k * u[c];
                        }
                    }
                } else {
                    double u[DT], r2 = 0.0;
#pragma unroll
                    for (int c = 0; c < DT; ++c) {
                        const double x = (c < a.D) ? sr_lin_x(a.la, c) : 0.0;
                        u[c] = (z[c] - x) * il2[c];
                        r2 = fma(u[c], z[c] - x, r2);
                    }
                    const double k = sf2 * exp(-0.5 * r2);
                    col[0] = k;
                    acc[0] = wgt * k;
                    int q = 1 + DT;
#pragma unroll
                    for (int c = 0; c < DT; ++c) {
                        if (1 + c < TQ) col[1 + c] = k * u[c];
                        acc[1 + c] = wgt * k * u[c];
#pragma unroll
                        for (int e = 0; e < DT; ++e)
                            if (e >= c) {
                                double hv = u[c] * u[e];
                                if (e == c) hv -= il2[c];
                                acc[q] = wgt * k * hv;
                                ++q;
                            }
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < TQ; ++t) ks[tid][t] = col[t];
        }
        if (owner) {                                  // workgroup-uniform
            // rows live in wavefronts 0 and 1: butterfly per wavefront, one pass through shared memory
#pragma unroll
            for (int q = 0; q < NACC; ++q) {
                const double v = sr_wave_sum(acc[q]);
                if ((tid & 63) == 0 && tid < SR_ST_ROWS) sh[(tid >> 6) * NACC + q] = v;
            }
            __syncthreads();
            if (tid < NACC) {
                const double sum = sh[tid] + sh[NACC + tid];
                if (SRC == 2) {
                    sr_st_agent(a.lin_part_w + ((long)d * a.nblk + j) * NACC + tid, sum);
                } else {
                    const int t = tid / (1 + DT), c = tid % (1 + DT);
                    if (TQ == 1 && SRC == 1 && a.slots) {       // polling finaliser: the chunk's slot [j][d][1 + D]
                        if (c <= a.D) sr_st_agent(a.slots + (long)a.n_out * a.ncb + ((long)j * a.n_out + d) * (1 + a.D) + c, sum);
                    } else if (t < a.ncols) {
                        if (c == 0) sr_st_agent(a.mu_part_w + ((long)j * a.n_out + d) * a.Tp + t, sum);
                        else if (c - 1 < a.D) sr_st_agent(a.jac_part_w + (((long)j * a.n_out + d) * a.D + (c - 1)) * a.Tp + t, sum);
                    }
                }
            }
        }
    }
    __syncthreads();

    // ---- this (column block, k-chunk) of V = U^-T [columns]
    double acc[TQ];
#pragma unroll
    for (int t = 0; t < TQ; ++t) acc[t] = 0.0;
    {
        int r = r0;
        if (first16) {
#pragma unroll
            for (int u = 0; u < 16; ++u)
#pragma unroll
                for (int t = 0; t < TQ; ++t) acc[t] = fma(wv[u], ks[r0 + u][t], acc[t]);
            r += 16;
        }
        for (; r + 15 <= kmax; r += 16) {
#pragma unroll
            for (int u = 0; u < 16; ++u) wv[u] = w[(long)(r + u) * a.Np];
#pragma unroll
            for (int u = 0; u < 16; ++u)
#pragma unroll
                for (int t = 0; t < TQ; ++t) acc[t] = fma(wv[u], ks[r + u][t], acc[t]);
        }
        for (; r + 3 <= kmax; r += 4) {
            double w4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) w4[u] = w[(long)(r + u) * a.Np];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int t = 0; t < TQ; ++t) acc[t] = fma(w4[u], ks[r + u][t], acc[t]);
        }
        for (; r <= kmax; ++r) {
            const double w1 = w[(long)r * a.Np];
#pragma unroll
            for (int t = 0; t < TQ; ++t) acc[t] = fma(w1, ks[r][t], acc[t]);
        }
    }
    double* out = a.Vp + ((long)d * a.npairs + p) * TQ * SR_ST_COLS;
#pragma unroll
    for (int t = 0; t < TQ; ++t) sr_st_agent(out + t * SR_ST_COLS + tid, acc[t]);

    if (a.probe == 1) return;
    // Polling finaliser (round 6; one ARD-RBF query, a.slots): the LAST workgroup of the grid in dispatch order -- everybody
    // else is resident or gone when it starts -- stays behind and gathers the self-validating slots: no second ticket
    // (store -> drain -> atomic -> loads were four dependent round trips behind the last column block: 5.7 of the 38 us of
    // the call at N = 5000).
    const bool poll = TQ == 1 && SRC == 1 && a.slots != nullptr;
    const bool finaliser = poll && blockIdx.x == gridDim.x - 1 && blockIdx.y == gridDim.y - 1;
    // ---- last workgroup of this column block: add the chunks, square (or multiply with column 0), reduce
    const int nch = 2 * cb + 2;
    const bool last1 = sr_ticket_last(a.tickets + d * a.ncb + cb, (unsigned)nch, &s_flag);
    if (!last1 && !finaliser) return;
    if (a.probe == 2) return;
    if (last1) {
        const double* src = a.Vp + ((long)d * a.npairs + cb * (cb + 1)) * TQ * SR_ST_COLS + tid;
        double v[TQ];
        // same association as the stand-alone reduce kernel (even chunks, odd chunks, ascending).  Round 6: the loads of ALL
        // columns and of up to 40 chunks are in flight together -- this reduction sits on the tail of the launch (the last
        // arriver of the longest column block runs it while the chip waits), and it is a chain of dependent round trips,
        // not of bytes: batches of 8 chunks per column were 5 round trips at N = 5000 for one column and 20 for four.
        constexpr int CB = (TQ == 1) ? 40 : 10;              // chunks per batch (TQ x CB loads in flight)
        double v0[TQ], v1[TQ];
#pragma unroll
        for (int t = 0; t < TQ; ++t) { v0[t] = 0.0; v1[t] = 0.0; }
        for (int c = 0; c < nch; c += CB) {
            double x[CB][TQ];
#pragma unroll
            for (int u = 0; u < CB; ++u)
#pragma unroll
                for (int t = 0; t < TQ; ++t)
                    x[u][t] = (c + u < nch) ? sr_ld<true>(src + ((long)(c + u) * TQ + t) * SR_ST_COLS) : 0.0;
#pragma unroll
            for (int u = 0; u < CB; u += 2)
#pragma unroll
                for (int t = 0; t < TQ; ++t) { v0[t] += x[u][t]; v1[t] += x[u + 1][t]; }
        }
#pragma unroll
        for (int t = 0; t < TQ; ++t) v[t] = v0[t] + v1[t];
#pragma unroll
        for (int t = 0; t < TQ; ++t) {
            const double q = sr_block_sum(v[t] * ((a.dot0 && t != 0) ? v[0] : v[t]), sh);
            if (tid == 0) {
                if (poll) sr_st_agent(a.slots + (long)d * a.ncb + cb, q);
                else sr_st_agent(a.part + ((long)d * a.ncb + cb) * a.Tp + t, q);
            }
        }
    }
    if (TQ == 1 && SRC == 1 && poll) {
        if (!finaliser) return;
        // ---- gather: every thread polls its share of the slots until none of them is empty (one load round trip per
        // look; everything the final stage needs arrives with the look that succeeds), keeps them in LDS, empties them
        // again for the next call
        __shared__ double slot_s[SR_ST1_SLOTS_MAX];
        const int nslot = (int)sr_st1_slots(a.ncb, a.n_out, a.D);
        constexpr int PER = SR_ST1_SLOTS_MAX / SR_ST1_THREADS;
        double val[PER];
        for (int it = 0; it < 4000000; ++it) {
            int missing = 0;
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const int i = tid + SR_ST1_THREADS * u;
                val[u] = (i < nslot) ? sr_ld<true>(a.slots + i) : 0.0;
                missing |= (i < nslot && (unsigned long long)__double_as_longlong(val[u]) == SR_ST1_EMPTY);
            }
            if (!__syncthreads_or(missing)) break;
        }
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int i = tid + SR_ST1_THREADS * u;
            if (i < nslot) {
                slot_s[i] = val[u];
                a.slots[i] = __longlong_as_double((long long)SR_ST1_EMPTY);
            }
        }
        __syncthreads();
        // ---- final stage from LDS, one wavefront per output: the loads and butterflies of sr_final_query_wave (few partial
        // sums: lane l holds partial l of every quantity), so the result is the ticket route's to the last bit
        const int lane = tid & 63;
        const int nsplit = 2 * a.ncb;
        for (int dd = tid >> 6; dd < a.n_out; dd += SR_ST1_THREADS / 64) {
            const double* ms = slot_s + (long)a.n_out * a.ncb + (long)dd * (1 + a.D);
            const int mstride = a.n_out * (1 + a.D);
            double m = 0.0, q = 0.0;
            for (int l = lane; l < nsplit; l += 64) m += ms[(long)l * mstride];          // (nsplit <= 64: one term per lane)
            for (int l = lane; l < a.ncb; l += 64) q += slot_s[(long)dd * a.ncb + l];
            m = sr_wave_sum(m);
            q = sr_wave_sum(q);
            double v = a.fa.sf2[dd] - q;
            if (!(v > SR_VAR_CLIP)) v = SR_VAR_CLIP;
            if (lane == 0) {
                a.fa.mu[dd] = m;
                a.fa.var[dd] = v;
            }
            if (a.fa.jac) {
                for (int c = 0; c < a.D; ++c) {
                    double g = 0.0;
                    for (int l = lane; l < nsplit; l += 64) g += ms[(long)l * mstride + 1 + c];
                    g = sr_wave_sum(g);
                    if (lane == 0) a.fa.jac[(long)dd * a.D + c] = g;
                }
            }
        }
        if (a.host_flag) {                                   // (as sr_stream_final)
            __threadfence_system();
            __syncthreads();
            if (tid == 0) __hip_atomic_store(a.host_flag, a.host_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    // ---- last column block of the call: final stage
    if (!sr_ticket_last(a.tickets + a.n_out * a.ncb, (unsigned)(a.n_out * a.ncb), &s_flag)) return;
    sr_stream_final(a, sh);
}

// Partial product of an MFMA work item: Vp[item][query][256 columns].  The accumulator of v_mfma_f64_16x16x4 holds, in lane
// (lk, ln), rows lk + 4 r (r = 0 .. 3) of query ln.  Round 6: the A-fragment lanes load the strip's columns PERMUTED (MFMA row
// m <-> column 4 (m % 4) + m / 4: the same 128-byte line, another lane), so that a lane's four rows are the neighbouring
// columns 4 lk .. 4 lk + 3 and leave as ONE 32-byte run: the four lanes of a query fill a whole 128-byte line per store.
// (N = 5000, T = 64: 102.5 -> 99.3 us, N = 3000: 74.2 -> 68.9, T = 32 45 from 48; non-temporal on top: no difference --
// profiles/r06_stream_items.txt)
__device__ __forceinline__ int sr_st_strip_col(int ln, int epi) { return epi ? 4 * (ln & 3) + (ln >> 2) : ln; }
template <int G>
__device__ __forceinline__ void sr_st_store_partial(double* out, const d4_t (&acc)[G], int wave, int lk, int ln, bool live, int epi) {
    if (epi == 0) {
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(long)(16 * g + ln) * SR_ST_COLS + 16 * wave + lk + 4 * r] = live ? acc[g][r] : 0.0;
        return;
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
        d4_t* dst = reinterpret_cast<d4_t*>(out + (long)(16 * g + ln) * SR_ST_COLS + 16 * wave + 4 * lk);
        const d4_t v = live ? acc[g] : d4_t{0.0, 0.0, 0.0, 0.0};
        if (epi == 2) __builtin_nontemporal_store(v, dst); else *dst = v;
    }
}

// ------------------------------------------------------------------------------------------------
// ONE 128-row k-chunk per workgroup (models whose grid would not cover the chip with longer runs: N < ~4000; there
// this form is faster than the run kernel below with KC = 1 -- N = 3000, T = 32: 26.0 + 14.6 against 33.3 + 17.0 us for
// kernel + reduction: its wavefronts skip the k-steps below their strip's diagonal).
// 16 G columns per workgroup on the MFMA 16x16x4 tile.  The (column block, k-chunk) pair is one workgroup of 16
// wavefronts, wavefront w owning the 16-column strip w: A-fragments straight from global (the 16 strips of a row are
// one contiguous 2 KiB segment), B-fragments = the K* rows in LDS (row stride 16 G + 16 doubles for G > 1: the two
// k-rows of a 32-lane ds_read_b64 group then sit on opposite bank halves), one ds_read_b64 per MFMA, G MFMAs per loaded
// A-fragment.  Vp[(d, pair)][column][256].
// ------------------------------------------------------------------------------------------------
// SRC 1 (ARD-RBF, D <= DT <= 5): the workgroup EVALUATES its 128 rows of K* itself instead of reading them from a K*
// pass of their own (a launch of 8 - 10 us in front of a kernel of 10 - 40): thread (row r = tid % 128, query group
// tid / 128) holds its training row and evaluates NC / 8 queries against it; the workgroup with the smallest column block
// of the chunk (cb == j / 2) also owns the chunk's share of the sums over the training points (mean, mean-Jacobian):
// slot j of the 2 ncb N-split partials the final stage adds.
template <int G, int SRC = 0, int DT = 1>
__global__ __launch_bounds__(1024) void sr_stream_mfma1_kernel(sr_stream_args a) {
    constexpr int NC = 16 * G;
    constexpr int LDK = (G == 1) ? 16 : NC + 16;
    __shared__ double ks[SR_ST_ROWS * LDK];
    // (G > 1: the owner's partial sums live in the 16 padding columns of the K* rows -- a buffer of their own would cost the
    //  G = 4 kernel its second workgroup per CU)
    __shared__ double own_[(SRC == 1 && G == 1) ? 2 * NC * (1 + DT) : 1];
    auto own = [&](int half, int t, int c) -> double& {
        return (G == 1) ? own_[(half * NC + t) * (1 + DT) + c] : ks[t * LDK + NC + half * (1 + DT) + c];
    };
    const int d = blockIdx.y, p = blockIdx.x;
    int cb, j;
    sr_pair_decode(p, cb, j);
    const int k0 = j * SR_ST_ROWS;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lk = lane >> 4, ln = lane & 15;
    const int c0 = blockIdx.z * NC;                      // first column of this workgroup (grid.z: column blocks)
    if (SRC == 0) {
        const double* ksrc = a.Ks + (long)d * a.Np * a.Tp + (long)k0 * a.Tp + c0;
        for (int e = threadIdx.x; e < SR_ST_ROWS * NC; e += 1024) {
            const int r = e / NC, t = e % NC;
            ks[r * LDK + t] = (k0 + r < a.Np && c0 + t < a.ncols_pad) ? ksrc[(long)r * a.Tp + t] : 0.0;
        }
    } else {
        constexpr int TPG = NC / 8;                      // queries per thread
        const int r = threadIdx.x & 127, tg = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 7);
        const int ip = k0 + r, it = ip - (a.Np - a.N);
        const bool valid = ip < a.Np && it >= 0;
        const bool owner = (cb == (j >> 1));             // workgroup-uniform
        const double wgt = valid ? a.alpha[(long)d * a.Np + ip] : 0.0;
        const double sf2 = a.sf2[d];
        double z[DT], il2[DT];
#pragma unroll
        for (int c = 0; c < DT; ++c) {
            const double l = (c < a.D) ? a.ls[d * a.D + c] : 1.0;
            il2[c] = (c < a.D) ? 1.0 / (l * l) : 0.0;
            z[c] = (valid && c < a.D) ? a.Z[(long)it * a.D + c] : 0.0;
        }
#pragma unroll
        for (int tt = 0; tt < TPG; ++tt) {
            const int t = tg * TPG + tt;
            const long tq = c0 + t;
            const bool live = tq < a.ncols;              // wavefront-uniform (a wavefront = 64 rows of one query group)
            double u[DT], r2 = 0.0;
#pragma unroll
            for (int c = 0; c < DT; ++c) {
                double x = 0.0;
                if (live && c < a.D) x = (c < a.na) ? a.xa[tq * a.lda + c] : a.xb[tq * a.ldb + (c - a.na)];
                u[c] = (z[c] - x) * il2[c];
                r2 = fma(u[c], z[c] - x, r2);
            }
            const double k = (valid && live) ? sf2 * exp(-0.5 * r2) : 0.0;
            ks[r * LDK + t] = k;
            if (owner) {
                const double w = wgt * k;
                const double m = sr_wave_sum(w);
                if (lane == 0) own(wave & 1, t, 0) = m;
#pragma unroll
                for (int c = 0; c < DT; ++c) {
                    const double gc = sr_wave_sum(w * u[c]);
                    if (lane == 0) own(wave & 1, t, 1 + c) = gc;
                }
            }
        }
        if (owner) {
            __syncthreads();
            const int t = (int)threadIdx.x / (1 + DT), c = (int)threadIdx.x % (1 + DT);
            if (t < NC && c0 + t < a.ncols) {
                const double sum = own(0, t, c) + own(1, t, c);
                if (c == 0) a.mu_part_w[((long)j * a.n_out + d) * a.Tp + c0 + t] = sum;
                else if (c - 1 < a.D) a.jac_part_w[(((long)j * a.n_out + d) * a.D + (c - 1)) * a.Tp + c0 + t] = sum;
            }
        }
    }
    __syncthreads();
    const int i0 = cb * SR_ST_COLS + 16 * wave;          // first column of this strip
    d4_t acc[G];
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = d4_t{0.0, 0.0, 0.0, 0.0};
    if (i0 < a.Np) {
        // k-steps of 4 rows: rows k0 + 4u + lk.  Rows beyond the strip's last column hold zeros of U^-1 (skipped),
        // rows in front of k_lo carry K* == 0 (skipped at k-step granularity).
        // (k0 may lie beyond the strip -- and, in the last column block of a model whose padded size is an odd multiple
        //  of 128, beyond the matrix: integer division truncates towards zero, (-1) / 4 + 1 used to give one k-step that read
        //  four rows past U^-1 of the last output)
        const int u_end = (i0 + 15 < k0) ? 0 : min(min(32, (a.Np - k0) / 4), (i0 + 15 - k0) / 4 + 1);
        int u = max(0, (a.k_lo - k0) / 4);
        const double* w = a.Wt + (long)d * a.Np * a.Np + (long)(k0 + lk) * a.Np + i0 + sr_st_strip_col(ln, a.epi);
        constexpr int UB = (G <= 2) ? 16 : 8;            // A-fragments in flight per batch
        for (; u + UB <= u_end; u += UB) {
            double af[UB];
#pragma unroll
            for (int q = 0; q < UB; ++q) af[q] = w[(long)(4 * (u + q)) * a.Np];
#pragma unroll
            for (int q = 0; q < UB; ++q)
#pragma unroll
                for (int g = 0; g < G; ++g)
                    acc[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[q], ks[(4 * (u + q) + lk) * LDK + 16 * g + ln], acc[g], 0, 0, 0);
        }
        for (; u < u_end; ++u) {
            const double af = w[(long)(4 * u) * a.Np];
#pragma unroll
            for (int g = 0; g < G; ++g)
                acc[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(af, ks[(4 * u + lk) * LDK + 16 * g + ln], acc[g], 0, 0, 0);
        }
    }
    // acc[g][r] = V[column i0 + 4 lk + r][query 16 g + ln]   (sr_st_store_partial)
    double* out = a.Vp + (((long)d * a.npairs + p) * (NC * gridDim.z) + c0) * SR_ST_COLS;
    sr_st_store_partial<G>(out, acc, wave, lk, ln, true, a.epi);
}

// part[d][cb][t] = sum_{i in column block cb} V_t[i] (dot0 ? V_0[i] : V_t[i]),  V_t = sum_chunks Vp;
// grid (ncb, n_out, live columns).  The last workgroup of a COLUMN t (predict: query t) runs that query's final stage;
// in linearize mode the last workgroup of the whole grid runs the single query's.
__global__ __launch_bounds__(256) void sr_stream_reduce1_kernel(sr_stream_args a, int nc) {
    __shared__ double sh[4 * 120];
    __shared__ int s_flag;
    const int cb = blockIdx.x, d = blockIdx.y, t = blockIdx.z;
    const int p0 = cb * (cb + 1), nch = 2 * cb + 2;
    const double* src = a.Vp + (((long)d * a.npairs + p0) * nc + t) * SR_ST_COLS + threadIdx.x;
    const long cs = (long)nc * SR_ST_COLS;               // chunk stride
    double v0 = 0.0, v1 = 0.0;
    for (int c = 0; c < nch; c += 8) {                    // (plain loads: Vp comes from the previous launch)
        double x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = (c + u < nch) ? src[(c + u) * cs] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; u += 2) { v0 += x[u]; v1 += x[u + 1]; }
    }
    double w = v0 + v1;
    if (a.dot0 && t != 0) {
        const double* s0 = a.Vp + (((long)d * a.npairs + p0) * nc) * SR_ST_COLS + threadIdx.x;
        double w0 = 0.0, w1 = 0.0;
        for (int c = 0; c < nch; c += 8) {
            double x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = (c + u < nch) ? s0[(c + u) * cs] : 0.0;
#pragma unroll
            for (int u = 0; u < 8; u += 2) { w0 += x[u]; w1 += x[u + 1]; }
        }
        w = w0 + w1;
    }
    const double q = sr_block_sum((v0 + v1) * w, sh);
    if (threadIdx.x == 0) sr_st_agent(a.part + ((long)d * a.ncb + cb) * a.Tp + t, q);
    if (a.mode == 0) {
        if (!sr_ticket_last(a.tickets + t, (unsigned)(a.n_out * a.ncb), &s_flag)) return;
        for (int dd = threadIdx.x >> 6; dd < a.n_out; dd += 4) sr_final_query_wave<true>(a.fa, t, dd, threadIdx.x & 63);
    } else {
        if (!sr_ticket_last(a.tickets, (unsigned)(a.n_out * a.ncb * gridDim.z), &s_flag)) return;
        sr_stream_final(a, sh);
    }
}

// ------------------------------------------------------------------------------------------------
// 16 G columns per workgroup on the MFMA 16x16x4 tile.  A workgroup of 16 wavefronts owns a column block of 256
// columns of U^-1 and a run of KC consecutive 128-row k-chunks of it (work item (cb, J), enumerated column block by
// column block); wavefront w owns the 16-column strip w: A-fragments straight from global (the 16 strips of a row are one
// contiguous 2 KiB segment), the next batch requested before the MFMAs of the current one are issued; B-fragments = the
// K* rows, staged through LDS SUB rows at a time, double-buffered (row stride 16 G + 16 doubles for G > 1: the two k-rows
// of a 32-lane ds_read_b64 group then sit on opposite bank halves), one ds_read_b64 per MFMA, G MFMAs per loaded
// A-fragment.  The accumulators stay in registers over the whole run: one partial result per (work item, column).
// (Round 2: one 128-row chunk per workgroup -- 840 workgroups of two load batches each at N = 5000, their prologues and
// epilogues exposed, and four times the partial sums: T = 16 / 32 / 64: 41 + 21 / 55 + 26 / 96 + 38 us for this kernel
// + its reduction.)
// Vp[(d, work item)][column][256].
// ------------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ int sr_st_items_of(int cb, int kc) { return (2 * cb + 2 + kc - 1) / kc; }
// work items of column block cb when an item is a run of `kr` rows (round 6: kr a multiple of the kernel's LDS stage, not of
// the 128-row chunk; the last item of a column block is shorter)
__host__ __device__ __forceinline__ int sr_st_items_rows(int cb, int kr) { return ((2 * cb + 2) * SR_ST_ROWS + kr - 1) / kr; }

// Work items come from a TABLE (round 6): entry p = (column block, run J of a.kr rows, slot of the partial result), longest
// runs first -- the launch is one workgroup per CU and as long as its longest resident sequence; with runs of whole chunks in
// column-block order, N = 5000 / T = 64 was 220 workgroups of up to 8 LDS stages on 256 CUs (1680 stages: 6.6 per CU).
#ifdef SR_LAB
// lab build: per-workgroup time stamps of the run kernel (100 MHz wall clock): start, first stage staged, loop done, end, stages
__device__ unsigned long long sr_lab_st_trace[8 * 4096];
extern "C" int sr_lab_stream_trace(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(sr_lab_st_trace), sizeof(unsigned long long) * (size_t)std::min(n, 8 * 4096));
}
#define SR_LAB_STAMP(i_) do { if (threadIdx.x == 0) { const unsigned wg_ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z); \
    if (wg_ < 4096) sr_lab_st_trace[8 * wg_ + (i_)] = wall_clock64(); } } while (0)
#define SR_LAB_NOTE(i_, v_) do { if (threadIdx.x == 0) { const unsigned wg_ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z); \
    if (wg_ < 4096) sr_lab_st_trace[8 * wg_ + (i_)] = (unsigned long long)(v_); } } while (0)
#else
#define SR_LAB_STAMP(i_) do {} while (0)
#define SR_LAB_NOTE(i_, v_) do {} while (0)
#endif
template <int G, bool MULTI = false>
__global__ __launch_bounds__(1024) void sr_stream_mfma_kernel(sr_stream_args a) {
    constexpr int NC = 16 * G;
    constexpr int LDK = (G == 1) ? 16 : NC + 16;
    constexpr int SUB = (G <= 2) ? 128 : (G == 4 ? 64 : 32);      // K* rows per LDS stage
    constexpr int UB = (G <= 2) ? 16 : (G == 4 ? 8 : 4);          // A-fragments (k-steps of 4 rows) per batch
    constexpr int BPS = SUB / 4 / UB;                             // batches per stage: 2
    constexpr int PF = SUB * NC / 1024;                           // K* doubles a thread moves per stage
    static_assert(BPS == 2 && BPS * UB * 4 == SUB && PF * 1024 == SUB * NC, "stage geometry");
    __shared__ double ks[2][SUB * LDK];
    const int d = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lk = lane >> 4, ln = lane & 15;
    const int c0 = blockIdx.z * NC;                      // first column of this workgroup (grid.z: column blocks)
    // The workgroup's work items, one after the other (round 6, second half: a launch with a few more items than CUs -- N = 5000,
    // T = 64: 258 runs of <= 7 stages -- gives two short ones to the same workgroup instead of taking longer runs: sr_stream_items)
    // (MULTI is a variant of its own: with the loop the compiler keeps the strips' addresses in registers across the items --
    //  98 -> 122 VGPRs at G = 4, spills at G = 2 and G = 8 -- so the one-item form stays exactly the kernel it was)
    const int* wg_tab = a.item_tab + 3 * a.nitems;
    const int it0 = MULTI ? wg_tab[blockIdx.x] : (int)blockIdx.x, it1 = MULTI ? wg_tab[blockIdx.x + 1] : (int)blockIdx.x + 1;
#pragma unroll 1
    for (int it = it0; it < it1; ++it) {
    const int cb = a.item_tab[3 * it], J = a.item_tab[3 * it + 1], p = a.item_tab[3 * it + 2];
    const int k0 = J * a.kr, k1 = min(min(k0 + a.kr, (2 * cb + 2) * SR_ST_ROWS), a.Np);
    const int nsub = (k1 - k0 + SUB - 1) / SUB;
    SR_LAB_STAMP(0);
    SR_LAB_NOTE(4, nsub);
    SR_LAB_NOTE(5, __smid());
    if (k0 >= a.Np) {
        // an EMPTY run (the chunk beyond Np of an odd padded size) reports zeros and is gone.  (Not a branch around the
        // prologue below: with one, the G = 2 kernel went from 112 VGPRs to 128 + 76 B of scratch, T = 32 at N = 5000 75 ->
        // 101 us; a run that starts at row 0 instead cost the same registers.)
        double* out = a.Vp + (((long)d * a.nitems + p) * (NC * gridDim.z) + blockIdx.z * NC) * SR_ST_COLS;
        for (int e = threadIdx.x; e < NC * SR_ST_COLS; e += 1024) out[e] = 0.0;
        continue;
    }
    const double* ksrc = a.Ks + (long)d * a.Np * a.Tp + (long)k0 * a.Tp + c0;

    // K* rows of stage `sub` that this thread moves (element e = tid + 1024 m: row e / NC, column e % NC)
    // (Every global load of the main loop is UNCONDITIONAL -- a clamped address and a select instead of a predicate, the
    //  current stage fetched once more instead of `if (more)`: behind a load that may or may not have been issued the
    //  compiler's s_waitcnt has to assume it was not, and the wavefronts drained their whole queue, the next batch
    //  included, at the end of every stage: T = 64 at N = 5000 86 us with the MFMA pipe 50 % busy.)
    auto ks_fetch = [&](int sub, double (&pf)[PF]) {
#pragma unroll
        for (int m = 0; m < PF; ++m) {
            const int e = tid + 1024 * m, r = sub * SUB + e / NC, t = e % NC;
            const bool in = c0 + t < a.ncols_pad;
            const double v = ksrc[(long)r * a.Tp + (in ? t : 0)];
            pf[m] = in ? v : 0.0;
        }
    };
    auto ks_put = [&](int buf, const double (&pf)[PF]) {
#pragma unroll
        for (int m = 0; m < PF; ++m) {
            const int e = tid + 1024 * m;
            ks[buf][(e / NC) * LDK + e % NC] = pf[m];
        }
    };

    const int i0 = min(cb * SR_ST_COLS + 16 * wave, a.Np - 16);      // first column of this strip (Np is a multiple of 128)
    // Every wavefront runs every batch of the work item -- uniform control flow, so that the requests of the next batch
    // stay in flight under the MFMAs of the current one.  No masks: rows below a strip's diagonal hold the zeros of U^-1,
    // rows in front of k_lo meet K* == 0.  (A per-wavefront range -- skipping the batches below the diagonal -- made the
    // compiler wait for every single load: 116 against 71 us at N = 5000, T = 16.  What it would save is half of the
    // two diagonal chunks of a column block: 5 % of the MFMAs at N = 5000.)
    const double* w = a.Wt + (long)d * a.Np * a.Np + (long)(k0 + lk) * a.Np + i0 + sr_st_strip_col(ln, a.epi);
    d4_t acc[G];
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = d4_t{0.0, 0.0, 0.0, 0.0};
    double pf[PF];
    double A0[UB], A1[UB];
    ks_fetch(0, pf);
#pragma unroll
    for (int q = 0; q < UB; ++q) A0[q] = w[(long)(4 * q) * a.Np];
    ks_put(0, pf);
    __syncthreads();
    SR_LAB_STAMP(1);
    // One stage = 2 UB k-steps of 4 rows: A-fragment of the step (A0: first half, A1: second half, each requested half a
    // stage ahead, ONE load per step), G B-fragments out of the LDS stage, G MFMAs.  The B-fragments run PD steps ahead of
    // their MFMAs in a ring of registers: read, wait, multiply in turn -- as the compiler arranges it -- ties every pair of
    // MFMAs to an LDS round trip; a wavefront then fills 40 % of the pipe, the oldest one of a SIMD is served first and
    // the last one runs alone at the end of every stage (measured at T = 64, N = 5000: wavefront 0 through its 64 MFMAs
    // after 11.3k cycles, at the barrier for 8.8k more; 20.9k per stage against 16.4k of MFMA time).  Scheduling barriers
    // pin the order.  The stage barrier is LDS-only: __syncthreads() would drain the A-fragments in flight.
    constexpr int PD = (G == 8) ? 0 : (G == 4 ? 1 : (G == 2 ? 2 : 4)), NR = PD + 1;   // (G = 8: 64 accumulator registers, no room for a ring)
    for (int sub = 0; sub < nsub; ++sub) {
        const double* kb = ks[sub & 1] + lk * LDK + ln;
        const bool more = sub + 1 < nsub;
        const int sn = more ? sub + 1 : sub;                  // the last stage fetches itself again (unused)
        double Bq[NR][G];
#pragma unroll
        for (int i = 0; i < PD; ++i)
#pragma unroll
            for (int g = 0; g < G; ++g) Bq[i][g] = kb[4 * i * LDK + 16 * g];
#pragma unroll
        for (int qq = 0; qq < 2 * UB; ++qq) {
            // the further a wavefront is into its stage, the lower its priority: the SIMD serves its OLDEST wavefront first,
            // and the youngest used to be left with half its stage to run alone (at 82 % of the pipe) behind the others
            if (qq == 0) __builtin_amdgcn_s_setprio(3);
            if (qq == UB / 2) __builtin_amdgcn_s_setprio(2);
            if (qq == UB) __builtin_amdgcn_s_setprio(1);
            if (qq == UB + UB / 2) __builtin_amdgcn_s_setprio(0);
            if (qq == 0) ks_fetch(sn, pf);
            if (qq < UB) A1[qq] = w[(long)(4 * ((2 * sub + 1) * UB + qq)) * a.Np];
            else A0[qq - UB] = w[(long)(4 * ((2 * sn) * UB + qq - UB)) * a.Np];
            if (qq + PD < 2 * UB) {
#pragma unroll
                for (int g = 0; g < G; ++g) Bq[(qq + PD) % NR][g] = kb[4 * (qq + PD) * LDK + 16 * g];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < G; ++g)
                acc[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(qq < UB ? A0[qq] : A1[qq - UB], Bq[qq % NR][g], acc[g], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (more) ks_put((sub + 1) & 1, pf);                  // (that buffer was last read in stage sub - 1)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    // acc[g][r] = V[column i0 + 4 lk + r][query 16 g + ln] (sr_st_store_partial); strips beyond the matrix (last column block) report zeros
    SR_LAB_STAMP(2);
    const bool live = cb * SR_ST_COLS + 16 * wave < a.Np;
    double* out = a.Vp + (((long)d * a.nitems + p) * (NC * gridDim.z) + c0) * SR_ST_COLS;
    sr_st_store_partial<G>(out, acc, wave, lk, ln, live, a.epi);
#ifdef SR_LAB
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SR_LAB_STAMP(3);
#endif
    }
}

// Reduction + final stage of the MFMA kernel's partial results, ONE workgroup of 512 threads per (column t, output d) and
// no hand-off between workgroups in predict mode: thread (sub, col) adds, for the column blocks cb = sub, sub + 2, ..,
// the work items of column `col` of the block in their fixed order, squares (or multiplies with column 0 in linearize
// mode) and accumulates; the workgroup sum is the query's |U^-T k*|^2, and its first wavefront runs the query's final
// stage on the spot.  (Round 2: one workgroup of 256 threads per (column block, output, column) + a ticket per query:
// 20 x n_out x T workgroups of two dependent round trips each -- 19 / 26 / 38 us for T = 16 / 32 / 64 at N = 5000,
// whatever the amount of partial sums.)  Linearize mode: the last workgroup of the grid (ticket) runs the final stage.
__global__ __launch_bounds__(512) void sr_stream_reduce_kernel(sr_stream_args a, int nc, int KR, int nitems) {
    __shared__ double sh[4 * 120];
    __shared__ double red[8];
    __shared__ int s_flag;
    const int t = blockIdx.x, d = blockIdx.y;
    const int tid = threadIdx.x, sub = tid >> 8, col = tid & 255;
    const long cs = (long)nc * SR_ST_COLS;               // item stride
    const double* base = a.Vp + (long)d * nitems * cs + col;
    const bool dot = a.dot0 && t != 0;
    // predict mode: the first wavefront requests the query's partial sums of mean and mean-Jacobian (written by the K*
    // pass: they do not depend on this reduction) BEFORE its share of the reduction, and holds them in registers -- the
    // final stage below then costs two butterflies instead of two more dependent round trips (16.9 -> 12 us at N = 5000).
    // Same loads, same order of additions as sr_final_query_wave (the results are the same to the last bit).
    const bool pre = a.mode == 0 && tid < 64 && a.fa.nsplit <= 512 && a.fa.D <= 3;
    double xm[8], xg[3][8];
    if (pre) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int sp = tid + 64 * u;
            const bool in = sp < a.fa.nsplit;
            xm[u] = in ? a.fa.mu_part[((long)sp * a.fa.n_out + d) * a.fa.Tp + t] : 0.0;
#pragma unroll
            for (int c = 0; c < 3; ++c)
                xg[c][u] = (in && a.fa.jac && c < a.fa.D) ? a.fa.jac_part[(((long)sp * a.fa.n_out + d) * a.fa.D + c) * a.fa.Tp + t] : 0.0;
        }
    }
    // first work item of this thread's first column block
    int p0 = 0;
    for (int c = 0; c < sub; ++c) p0 += sr_st_items_rows(c, KR);
    double acc = 0.0;
    if (!dot && sr_st_items_rows(a.ncb - 1, KR) <= 12) {
        // at most 12 work items per column block (N = 5000: KC = 4, 1 .. 10): the items of FIVE column blocks are
        // requested together -- the reduction is a chain of dependent round trips, not of bytes
        for (int cb = sub; cb < a.ncb; cb += 10) {
            double x[5][12];
            int pp = p0;
#pragma unroll
            for (int b = 0; b < 5; ++b) {
                const int cbb = cb + 2 * b;
                const int nch = cbb < a.ncb ? sr_st_items_rows(cbb, KR) : 0;
#pragma unroll
                for (int u = 0; u < 12; ++u) x[b][u] = (u < nch) ? base[(long)(pp + u) * cs + (long)t * SR_ST_COLS] : 0.0;
                for (int c = cbb; c < cbb + 2 && c < a.ncb; ++c) pp += sr_st_items_rows(c, KR);
            }
            p0 = pp;
#pragma unroll
            for (int b = 0; b < 5; ++b) {
                double v = 0.0;
#pragma unroll
                for (int u = 0; u < 12; ++u) v += x[b][u];
                acc = fma(v, v, acc);
            }
        }
    } else {
        for (int cb = sub; cb < a.ncb; cb += 2) {
            const int nch = sr_st_items_rows(cb, KR);
            double v = 0.0, v0 = 0.0;
            for (int c = 0; c < nch; c += 12) {               // (plain loads: Vp comes from the previous launch)
                double x[12], y[12];
#pragma unroll
                for (int u = 0; u < 12; ++u) {
                    x[u] = (c + u < nch) ? base[(long)(p0 + c + u) * cs + (long)t * SR_ST_COLS] : 0.0;
                    y[u] = (dot && c + u < nch) ? base[(long)(p0 + c + u) * cs] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 12; ++u) { v += x[u]; v0 += y[u]; }
            }
            acc = fma(v, dot ? v0 : v, acc);
            for (int c = cb; c < cb + 2 && c < a.ncb; ++c) p0 += sr_st_items_rows(c, KR);
        }
    }
    // sum over the workgroup in a fixed order: wavefront butterflies, then the 16 wavefront sums
    const double wsum = sr_wave_sum(acc);
    if ((tid & 63) == 0) red[tid >> 6] = wsum;
    __syncthreads();
    double q = 0.0;
    if (tid == 0) {
#pragma unroll
        for (int w = 0; w < 8; ++w) q += red[w];
        // (agent scope: in linearize mode another workgroup reads it; the final stage below reads it back the same way)
        sr_st_agent(a.part + (long)d * a.Tp + t, q);
    }
    if (pre) {
        double m = 0.0, g[3] = {0.0, 0.0, 0.0};
#pragma unroll
        for (int u = 0; u < 8; ++u) m += xm[u];
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int u = 0; u < 8; ++u) g[c] += xg[c][u];
        m = sr_wave_sum(m);
#pragma unroll
        for (int c = 0; c < 3; ++c) g[c] = sr_wave_sum(g[c]);
        if (tid == 0) {
            double v = (a.fa.kxx ? a.fa.kxx[(long)d * a.fa.Tp + t] : a.fa.sf2[d]) - q;
            if (!(v > SR_VAR_CLIP)) v = SR_VAR_CLIP;
            a.fa.mu[(long)t * a.fa.n_out + d] = m;
            a.fa.var[(long)t * a.fa.n_out + d] = v;
            if (a.fa.jac)
                for (int c = 0; c < a.fa.D; ++c) a.fa.jac[((long)t * a.fa.n_out + d) * a.fa.D + c] = g[c];
        }
        return;
    }
    if (a.mode == 0) {
        if (tid < 64) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            sr_final_args fa = a.fa;
            fa.nrb = 1;                                   // part[d][0][t] holds the whole sum
            sr_final_query_wave<true>(fa, t, d, tid);
        }
    } else {
        if (!sr_ticket_last(a.tickets, (unsigned)(a.n_out * gridDim.x), &s_flag)) return;
        sr_stream_args b = a;
        b.ncb = 1;
        sr_stream_final(b, sh);
    }
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
long sr_stream_vp_doubles(int Np, int n_out, int ncols) {
    const int ncb = (Np + SR_ST_COLS - 1) / SR_ST_COLS;
    return (long)n_out * ncb * (ncb + 1) * ncols * SR_ST_COLS;
}
int sr_stream_tickets(int Np, int n_out) {
    const int ncb = (Np + SR_ST_COLS - 1) / SR_ST_COLS;
    return std::max(n_out * ncb + 1, 128 + 1);
}
// columns the accumulate kernel works on for `ncols` live ones: 1, 4, or a multiple of 16 (<= 128)
int sr_stream_width(int ncols) {
    if (ncols <= 1) return 1;
    if (ncols <= 4) return 4;
    if (ncols <= 16) return 16;
    if (ncols <= 32) return 32;
    if (ncols <= 64) return 64;
    return 128;
}

// The plan of the MFMA route for nc (16 .. 128) columns: g = groups of 16 columns per workgroup, kc = k-chunks per work item.
void sr_stream_plan(int Np, int n_out, int nc, int* g_out, int* kc_out, bool can_fuse) {
    const int ncb = (Np + SR_ST_COLS - 1) / SR_ST_COLS;
    const long npairs = (long)ncb * (ncb + 1);
    // columns per workgroup: as many as possible (U^-1 is then read once for all of them) while the grid still
    // covers the chip; a small model keeps its column groups side by side in grid.z (they re-read U^-1 from L2)
    int g = nc / 16;
    while (g > 1 && npairs * n_out * (nc / (16 * g)) < 256) g >>= 1;
    const int gz = nc / (16 * g);
    // k-chunks per workgroup: as long a run as still leaves about one workgroup per CU (N = 5000, T = 16 / 64 / 128
    // columns, whole call: runs of 1 / 2 / 3 / 4 chunks 79 / 70 / 81 / 66, 169 / 138 / 166 / 125, 288 / 245 / 300 / 218 us)
    auto items = [&](int kc) { long n = 0; for (int cb = 0; cb < ncb; ++cb) n += sr_st_items_of(cb, kc); return n; };
    int kc = 1;
    static const long min_items = sr_lab_env("SR_ST_MIN_ITEMS", 200);       // (lab build: 1000000 = one-chunk work items everywhere)
    for (int c : {2, 3, 4, 6, 8})
        if (items(c) * n_out * gz >= min_items) kc = c;
    // 16 columns: one-chunk work items that evaluate their rows of K* themselves (no K* pass in front) stay ahead of the run
    // kernel up to ~7000 rows (round 6 scan, same box, T = 8 / 16, one-chunk against runs: N = 3400 32 / 35 against 45 / 45 us,
    // 4200 43 / 48 against 52 / 52, 5000 53 / 60 against 63 / 62, 7000 108 / 115 against 113 / 113; with 32 columns the two are
    // level up to 3800 rows and the runs win beyond: profiles/r06_stream_items.txt)
    if (can_fuse && nc == 16 && ncb <= 28 && min_items == 200) kc = 1;
    *g_out = g; *kc_out = kc;
}


// Work items of the run kernel (kc > 1 in sr_stream_plan) for nc columns: the run length kr (rows; a multiple of the kernel's LDS
// stage, at least a chunk) that gives the shortest launch, and the table [column block, run, slot] x items, longest runs first.
// "Shortest": the workgroups are dealt to the CUs in grid order (x fastest, then output, then column group), one per CU, each to
// the CU that is free first; a run costs its rows plus ~32 rows' worth of prologue and epilogue.  (Round 5 took runs of whole
// chunks in column-block order, as long as >= 200 workgroups remained: N = 5000, T = 64: 220 workgroups of up to 512 rows =
// 8 stages on the longest CU where 1680 stages over 256 CUs are 6.6.)
int sr_stream_items(int Np, int n_out, int nc, int n_cu, bool can_fuse, std::vector<int>& tab, int* nitems_out, int* nwg_out) {
    int g, kc;
    sr_stream_plan(Np, n_out, nc, &g, &kc, can_fuse);
    tab.clear();
    *nitems_out = 0; *nwg_out = 0;
    if (kc <= 1) return 0;
    const int ncb = (Np + SR_ST_COLS - 1) / SR_ST_COLS;
    const int gz = nc / (16 * g);
    const int sub = (g <= 2) ? 128 : (g == 4 ? 64 : 32);          // rows per LDS stage (sr_stream_mfma_kernel)
    const int cus = n_cu > 0 ? n_cu : 256;
    static const int forced = (int)sr_lab_env("SR_ST_KR", 0);     // (lab build: rows per run, 0 = planned; -1 = round 5's runs)
    // several items per workgroup only with 64 columns per workgroup (g = 4: the MFMA-bound width; the narrower ones are bound by
    // the stream and their kernels would spill with the loop over the items)
    static const int no_pack_lab = (int)sr_lab_env("SR_ST_NO_PACK", 0);   // (lab build: 1 = one work item per workgroup everywhere)
    const bool no_pack = no_pack_lab != 0 || g != 4;
    struct item { int rows, cb, J, slot; };
    auto build = [&](int kr) {                                    // items sorted by length, longest first (stable)
        std::vector<item> v;
        int slot = 0;
        for (int cb = 0; cb < ncb; ++cb) {
            const int n = sr_st_items_rows(cb, kr), top = std::min((2 * cb + 2) * SR_ST_ROWS, Np);
            for (int J = 0; J < n; ++J, ++slot) v.push_back({std::max(0, std::min(J * kr + kr, top) - J * kr), cb, J, slot});
        }
        std::stable_sort(v.begin(), v.end(), [](const item& x, const item& y) { return x.rows > y.rows; });
        return v;
    };
    // The launch is ONE round: at most one workgroup per CU (a workgroup that has to wait for a CU measured far worse than its
    // length says -- N = 5000, T = 64: 258 workgroups of <= 448 rows 112 us, 220 of <= 512 rows 102).  W workgroups per (output,
    // column group); with more items than that, the items are dealt longest first to the workgroup that is shortest so far, and a
    // workgroup runs its items one after the other.  An item costs its rows plus ~32 rows' worth of prologue and epilogue (96 behind
    // another item).
    const int W = std::max(1, cus / (n_out * gz));
    auto pack = [&](const std::vector<item>& v, std::vector<std::vector<int>>& bins) {
        const int nb = no_pack ? (int)v.size() : std::min((int)v.size(), W);
        bins.assign(nb, {});
        std::vector<long> load(nb, 0);
        for (int i = 0; i < (int)v.size(); ++i) {
            const int b = (int)(std::min_element(load.begin(), load.end()) - load.begin());
            // (a further item of a workgroup costs more than the first: its prologue starts cold behind the stores of the item in
            //  front.  With 32 rows for every item and ties going to the shortest runs the plan of N = 5400 was 125 workgroups of
            //  two 4-stage items each -- 125 - 140 us against 113 with one 9-stage item: profiles/r06_stream_pack.txt)
            load[b] += v[i].rows + (bins[b].empty() ? 32 : 96);
            bins[b].push_back(i);
        }
        long worst = 0;
        if (no_pack) {                                           // the old estimate: dealt in grid order to the CU that is free first
            std::vector<long> cu(cus, 0);
            for (int z = 0; z < gz * n_out; ++z)
                for (const item& e : v) { auto it = std::min_element(cu.begin(), cu.end()); *it += e.rows + 32; worst = std::max(worst, *it); }
            if ((long)v.size() * gz * n_out > cus) worst += 1000000;      // (more than one round: last resort)
        } else
            worst = *std::max_element(load.begin(), load.end());
        return worst;
    };
    int best_kr = kc * SR_ST_ROWS;
    if (forced > 0) best_kr = (forced + sub - 1) / sub * sub;
    else if (forced == 0) {
        long best = -1;
        for (int kr = std::max(SR_ST_ROWS, 2 * sub); kr <= 32 * SR_ST_ROWS; kr += sub) {
            std::vector<std::vector<int>> bins;
            const long m = pack(build(kr), bins);
            if (best < 0 || m < best || (!no_pack && m == best)) { best = m; best_kr = kr; }     // (ties: the longer runs when packing)
        }
    }
    const std::vector<item> v = build(best_kr);
    std::vector<std::vector<int>> bins;
    if (forced < 0) {                                              // round 5: column-block order (no sorting), one item each
        bins.clear();
        std::vector<int> order(v.size());
        for (size_t i = 0; i < v.size(); ++i) order[v[i].slot] = (int)i;
        for (size_t sidx = 0; sidx < v.size(); ++sidx) bins.push_back({order[sidx]});
    } else
        (void)pack(v, bins);
    // table: [column block, run, slot] x items in workgroup order, then the workgroups' first entries (nwg + 1)
    std::vector<int> first;
    for (const auto& b : bins) {
        first.push_back((int)tab.size() / 3);
        for (int i : b) { tab.push_back(v[i].cb); tab.push_back(v[i].J); tab.push_back(v[i].slot); }
    }
    first.push_back((int)tab.size() / 3);
    tab.insert(tab.end(), first.begin(), first.end());
    *nitems_out = (int)v.size();
    *nwg_out = (int)bins.size();
    return best_kr;
}

// src: 0 columns from a.Ks, 1 ARD-RBF predict columns evaluated in the kernel, 2 ARD-RBF linearize columns
int sr_launch_stream(sr_stream_args a, int src, hipStream_t s) {
    a.ncb = (a.Np + SR_ST_COLS - 1) / SR_ST_COLS;
    a.npairs = a.ncb * (a.ncb + 1);
    const int nc = max(sr_stream_width(a.ncols), a.width_min);
    SR_CHECK(a.ncols >= 1 && a.ncols <= 128, SR_EINVAL, "stream: %d columns", a.ncols);
    static const int probe_env = (int)sr_lab_env("SR_ST1_PROBE", 0);      // (lab build: scripts/t1_probe.py)
    a.probe = probe_env;
    static const int epi_env = (int)sr_lab_env("SR_ST_EPI", 1);           // (lab build: A/B of the partial-product stores)
    a.epi = epi_env;
    dim3 grid(a.npairs, a.n_out);
    if (nc <= 4) {
        SR_CHECK(src == 0 || (a.D <= 5 && (src == 1 || a.D + 1 <= 4)), SR_EINVAL, "stream: src %d with D = %d", src, a.D);
#define SR_ST1(TQ, SRC, DT) hipLaunchKernelGGL((sr_stream1_kernel<TQ, SRC, DT>), grid, dim3(SR_ST1_THREADS), 0, s, a)
        if (src == 0) { if (nc == 1) SR_ST1(1, 0, 1); else SR_ST1(4, 0, 1); }
        else if (src == 1) {
            if (a.D <= 3) { if (nc == 1) SR_ST1(1, 1, 3); else SR_ST1(4, 1, 3); }
            else { if (nc == 1) SR_ST1(1, 1, 5); else SR_ST1(4, 1, 5); }
        } else SR_ST1(4, 2, 3);
#undef SR_ST1
        SR_HIP(hipGetLastError());
        return SR_OK;
    }
    SR_CHECK(src == 0 || (src == 1 && a.D <= 5), SR_EINVAL, "stream: src %d with %d columns, D = %d", src, a.ncols, a.D);
    a.ncols_pad = a.ncols;
    int g, kc;
    sr_stream_plan(a.Np, a.n_out, nc, &g, &kc, src == 1);
    grid.z = nc / (16 * g);
    SR_CHECK(src == 0 || kc == 1, SR_EINVAL, "stream: columns evaluated in the kernel only for one-chunk work items (runs of %d)", kc);
    if (kc == 1) {
        // (columns evaluated in the kernel: 16 or 32 per workgroup only -- every column block re-evaluates the chunk's rows,
        //  and from 64 columns on that costs a workgroup more than the K* pass it saves: N = 3000, T = 64 76 -> 93 us)
        SR_CHECK(src == 0 || g <= 2, SR_EINVAL, "stream: %d columns per workgroup are not evaluated in the kernel", 16 * g);
#define SR_M1(G_) do { if (src == 0) hipLaunchKernelGGL((sr_stream_mfma1_kernel<G_, 0, 1>), grid, dim3(1024), 0, s, a); \
                       else if (a.D <= 3) hipLaunchKernelGGL((sr_stream_mfma1_kernel<G_, 1, 3>), grid, dim3(1024), 0, s, a); \
                       else hipLaunchKernelGGL((sr_stream_mfma1_kernel<G_, 1, 5>), grid, dim3(1024), 0, s, a); } while (0)
        switch (g) {
            case 1: SR_M1(1); break;
            case 2: SR_M1(2); break;
            case 4: hipLaunchKernelGGL((sr_stream_mfma1_kernel<4, 0, 1>), grid, dim3(1024), 0, s, a); break;
            default: hipLaunchKernelGGL((sr_stream_mfma1_kernel<8, 0, 1>), grid, dim3(1024), 0, s, a); break;
        }
#undef SR_M1
        SR_HIP(hipGetLastError());
        hipLaunchKernelGGL(sr_stream_reduce1_kernel, dim3(a.ncb, a.n_out, a.ncols), dim3(256), 0, s, a, nc);
        SR_HIP(hipGetLastError());
        return SR_OK;
    }
    SR_CHECK(a.item_tab != nullptr && a.kr >= SR_ST_ROWS && a.nitems > 0 && a.nwg > 0 && a.nwg <= a.nitems &&
             (a.nwg == a.nitems || g == 4), SR_EINVAL, "stream: no work-item table (sr_stream_items)");
    const int nitems = a.nitems;
    grid.x = a.nwg;
    switch (g) {
        case 1: hipLaunchKernelGGL(sr_stream_mfma_kernel<1>, grid, dim3(1024), 0, s, a); break;
        case 2: hipLaunchKernelGGL(sr_stream_mfma_kernel<2>, grid, dim3(1024), 0, s, a); break;
        case 4:
            if (a.nwg < nitems) hipLaunchKernelGGL((sr_stream_mfma_kernel<4, true>), grid, dim3(1024), 0, s, a);
            else hipLaunchKernelGGL(sr_stream_mfma_kernel<4>, grid, dim3(1024), 0, s, a);
            break;
        default: hipLaunchKernelGGL(sr_stream_mfma_kernel<8>, grid, dim3(1024), 0, s, a); break;
    }
    SR_HIP(hipGetLastError());
    hipLaunchKernelGGL(sr_stream_reduce_kernel, dim3(a.ncols, a.n_out), dim3(512), 0, s, a, nc, a.kr, nitems);
    SR_HIP(hipGetLastError());
    return SR_OK;
}
