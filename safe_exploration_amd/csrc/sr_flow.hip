// sr_flow.hip -- tile-flow Cholesky of the model update at the chain-bound sizes (round 6).
//
// The launched form (sr_capi_update.hip) walks the block chain with three dependent launches per 128 rows -- diagonal block,
// block-row solve, update of the next rows -- beside a trailing update on another stream, and every one of the chain's kernels
// then waits for CUs the trailing update holds (block step ~100 us in situ against ~55 alone).  Here the whole
// factorisation K = U^T U is ONE resident kernel of GEMM workgroups plus one resident diagonal-block workgroup per output
// (sr_flow_diag_server_kernel, sr_factor.hip), and the dependencies travel through counters in device memory (sr_flow.h):
//
//   * the block rows go in PANELS (sr_flow.h).  Behind a panel every 128 x 128 block of the trailing matrix takes the panel's
//     factor rows in one product of K = 128 * panel, read-modify-write ("UPD" tasks) -- that is where the flops are, and any
//     workgroup can do any of them the moment the panel's rows exist.  INSIDE a panel a row is finished left-looking: its
//     tiles take the rows of the panel above them one by one as they become final (tr[k][..] says so), the tile stays in the
//     accumulators in between -- by the time the chain reaches block row i its tiles lack one K = 128 step.
//     (First form of this file, everything left-looking from row 0: a tile of row r is r sequential K-steps of ONE workgroup,
//     and 512 resident workgroups hold three block rows of tiles -- N = 5000 3.2 ms for the Cholesky against 0.73 at N = 2000.)
//   * "solve": U[i][tile] = U_ii^-T A'[i][tile] as a product with the inverse the diagonal-block workgroup leaves (dd[i]), out
//     of place into W, as in the launched form.  Far blocks: update and solve are ONE task (the tile goes to U and comes back as
//     the product's B operand: half as many waiting workgroups).
//   * near the diagonal (`band` blocks right of it) tiles are 64 x 64 -- the two products between two diagonal blocks are on
//     the critical path and a 64-tile's K = 128 step is 3.4 us of a CU against 14 -- further out 128 x 128 (16 flop per operand
//     byte instead of 8).
//   * the band next to the diagonal never waits for a UPD of the panel just finished: its tiles take the previous panel's
//     rows too, left-looking, so that a panel boundary costs the chain nothing.
//   * tasks are handed out in ONE order by one fetch-and-add -- per block row: [its share of the UPD tasks of the previous
//     panel; with a panel's first row all those the panel's own rows need] [critical: solves of the row, then the updates of the NEXT
//     row's band -- held while the diagonal block is being factored, they lack one K-step when the solves are through] [far blocks].  A task that cannot start yet waits inside with its tile in the accumulators.  Every task depends
//     only on tasks in front of it in that order (and on the diagonal-block workgroup, which depends on such tasks), and whoever
//     holds a task is running: no deadlock whatever the number of resident workgroups.
//   * past the middle block row a quarter of the workgroups keep up with the chain; the others leave, and the stages of the
//     triangular inversion that wait behind their gates on another stream (sr_flow_gate_kernel) take their CUs.
//   Measured and not kept (profiles/r06_flow.txt): (a) two queues -- rows / UPD -- each served by one of the two workgroups a CU
//   holds: the UPD workgroup then computes alone on its CU, and a lone workgroup reaches 74 % of the fp64 matrix pipe; (b) three
//   queues and a workgroup that takes the head of the first queue whose head can START, else looks again: by compare-and-swap
//   500 workgroups fight for one head (block steps of 500 us), by fetch-and-add the search costs every task several
//   dependent atomics -- slower than waiting inside at every size.
//
// Memory model.  A producer's tile leaves behind an agent-scope release before its counter is raised.  Factor rows (W) and
// diagonal-block inverses are written exactly once per run and only after the kernel started: no line of them can sit in a
// cache before its counter says so (tiles are 512-byte aligned rows: no line is shared between tiles), so the loads behind the
// counter need no invalidate.  Tiles of the Gram matrix (U) are read-modify-written by several workgroups in turn, on
// different XCDs with L2s of their own: those loads and stores are agent-scope (sc1: they go to the memory side), and so
// are the LDS-DMA loads of a solve that reads a tile other workgroups wrote.  `acq` = 1 (lab
// switch SR_FLOW_ACQ) puts that acquire behind every wait.
#include "sr_mfma_tile.h"
#include "sr_flow.h"
#include <algorithm>
#include <vector>

namespace {

struct fl_tile64 {
    using Acc = srt64::Acc;
    static constexpr int T = 64, NI = 2;
    static __device__ __forceinline__ void mainloop(const double* A, long lda, const double* B, long ldb, int k0, int k1,
                                                    double* smem, Acc& acc) {
        srt64::mainloop_tn_pipe<false>(A, lda, B, ldb, k0, k1, smem, acc);
    }
    static __device__ __forceinline__ void mainloop_coh(const double* A, long lda, const double* B, long ldb, int k0, int k1,
                                                        double* smem, Acc& acc) {
        srt64::mainloop_tn_pipe<true>(A, lda, B, ldb, k0, k1, smem, acc);
    }
    static __device__ __forceinline__ int row(int wm, int mi, int lane, int r) { return srt64::acc_row(wm, mi, lane, r); }
    static __device__ __forceinline__ int col(int wn, int ni, int lane) { return srt64::acc_col(wn, ni, lane); }
};
struct fl_tile128 {
    using Acc = srt::Acc;
    static constexpr int T = 128, NI = 4;
    static __device__ __forceinline__ void mainloop(const double* A, long lda, const double* B, long ldb, int k0, int k1,
                                                    double* smem, Acc& acc) {
        srt::mainloop_tn_pipe<false, false>(A, lda, B, ldb, k0, k1, smem, acc);
    }
    static __device__ __forceinline__ void mainloop_coh(const double* A, long lda, const double* B, long ldb, int k0, int k1,
                                                        double* smem, Acc& acc) {
        srt::mainloop_tn_pipe<false, true>(A, lda, B, ldb, k0, k1, smem, acc);
    }
    static __device__ __forceinline__ int row(int wm, int mi, int lane, int r) { return srt::acc_row(wm, mi, lane, r); }
    static __device__ __forceinline__ int col(int wn, int ni, int lane) { return srt::acc_col(wn, ni, lane); }
};
static_assert(srt::SMEM_DOUBLES == srt64::SMEM_DOUBLES, "one LDS buffer for both tiles");

struct fl_ctx {
    unsigned* status;
    unsigned long long timeout;
    int* sh;                     // LDS words: what wavefront 0 found ([0] fl_rows_ready, [1] fl_wait4 / start-up), [2] ticks it waited
    int acq;
};

// wavefront 0: every 64th look at the counters also looks at the status word and the clock (wavefront-uniform result)
__device__ __forceinline__ bool fl_give_up(const fl_ctx& cx, unsigned& spins, unsigned long long t0) {
    if ((++spins & 63) != 0) return false;
    if (__hip_atomic_load(cx.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return true;
    if (wall_clock64() - t0 > cx.timeout) {
        __hip_atomic_store(cx.status, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return true;
    }
    return false;
}

// ALL threads call.  Waits until the factor's block row k0 is final at the 64-columns c[0..3] (tr >= 2 each), then returns
// how many CONSECUTIVE block rows k0, k0 + 1, .. < kmax are (at most 16); -1: give up.
__device__ __forceinline__ int fl_rows_ready(const fl_ctx& cx, const unsigned* tr, int nt, int k0, int kmax, int c0, int c1,
                                             int c2, int c3) {
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        const int kk = lane >> 2, f = lane & 3;
        const int c = f == 0 ? c0 : (f == 1 ? c1 : (f == 2 ? c2 : c3));
        const int k = k0 + kk;
        const bool valid = k < kmax;
        const unsigned* p = tr + (long)(valid ? k : k0) * nt + c;
        const unsigned long long t0 = wall_clock64();
        unsigned spins = 0;
        int n;
        for (;;) {
            const unsigned v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long m = __ballot(valid && v >= 2u);
            const unsigned long long r = m & (m >> 1) & (m >> 2) & (m >> 3) & 0x1111111111111111ull;
            const unsigned long long x = ~r & 0x1111111111111111ull;
            n = x ? (__builtin_ctzll(x) >> 2) : 16;
            if (n > 0) break;
            __builtin_amdgcn_s_sleep(1);
            if (fl_give_up(cx, spins, t0)) { n = -1; break; }
        }
        if (lane == 0) { *cx.sh = n; cx.sh[2] += (int)(wall_clock64() - t0); }
    }
    __syncthreads();
    const int n = *cx.sh;
    if (n > 0 && cx.acq) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return n;
}

// ALL threads call.  Waits for up to four counters p[f] >= t[f] (f < nf); false: give up.
__device__ __forceinline__ bool fl_wait4(const fl_ctx& cx, const unsigned* p0, unsigned t0_, const unsigned* p1, unsigned t1_,
                                         const unsigned* p2, unsigned t2_, int nf) {
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        const unsigned* p = lane == 0 ? p0 : (lane == 1 ? p1 : p2);
        const unsigned t = lane == 0 ? t0_ : (lane == 1 ? t1_ : t2_);
        const bool mine = lane < nf;
        const unsigned long long t0 = wall_clock64();
        unsigned spins = 0;
        int ok = 1;
        for (;;) {
            const unsigned v = mine ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            if (__ballot(mine && v < t) == 0ull) break;
            __builtin_amdgcn_s_sleep(1);
            if (fl_give_up(cx, spins, t0)) { ok = 0; break; }
        }
        if (lane == 0) { cx.sh[1] = ok; cx.sh[2] += (int)(wall_clock64() - t0); }
    }
    __syncthreads();
    const bool ok = cx.sh[1] != 0;
    if (ok && cx.acq) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return ok;
}

// the tile is out: release, then the counters (64-tile: one row half of one 64-column; 128-tile: both halves of two)
template <class TL>
__device__ __forceinline__ void fl_publish(unsigned* cnt) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (threadIdx.x == 0) {
        if (TL::T == 64) {
            __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            __hip_atomic_fetch_add(cnt, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(cnt + 1, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
__device__ __forceinline__ void fl_publish1(unsigned* cnt) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// C_tile -= sum_{k in [k_lo, k_hi) block rows} W[k][r0 ..]^T W[k][n0 ..] for the tile of U at rows r0, columns n0 -- once block
// (r0 / 128, n0 / 128) has taken `apv` panels (*apw >= apv), the factor rows as they become final.  The tile is read and
// written with agent-scope accesses (other workgroups, on other XCDs, wrote it before and will read it after).
template <class TL>
__device__ __forceinline__ bool fl_accum(const fl_ctx& cx, double* U, const double* W, long Np, long r0, int n0, int k_lo,
                                         int k_hi, const unsigned* tr, int nt, const unsigned* apw, unsigned apv, double* smem) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    if (apv > 0 && !fl_wait4(cx, apw, apv, apw, apv, apw, apv, 1)) return false;
    double* C = U + r0 * Np + n0;
    typename TL::Acc acc;
#pragma unroll
    for (int mi = 0; mi < TL::NI; ++mi)
#pragma unroll
        for (int ni = 0; ni < TL::NI; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                acc.v[mi][ni][r] = -__hip_atomic_load(C + (long)TL::row(wm, mi, lane, r) * Np + TL::col(wn, ni, lane),
                                                      __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int ca = (int)(r0 >> 6), cb = n0 >> 6;
    const int ca1 = TL::T == 128 ? ca + 1 : ca, cb1 = TL::T == 128 ? cb + 1 : cb;
    int k = k_lo;
    while (k < k_hi) {
        const int n = fl_rows_ready(cx, tr, nt, k, k_hi, ca, ca1, cb, cb1);
        if (n < 0) return false;
        TL::mainloop(W + r0, Np, W + n0, Np, k * SR_NB, (k + n) * SR_NB, smem, acc);
        k += n;
    }
#pragma unroll
    for (int mi = 0; mi < TL::NI; ++mi)
#pragma unroll
        for (int ni = 0; ni < TL::NI; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                __hip_atomic_store(C + (long)TL::row(wm, mi, lane, r) * Np + TL::col(wn, ni, lane), -acc.v[mi][ni][r],
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
}

// block-row solve of the tile at rows 128 i + moff, columns n0:  W_tile = (U_ii^-1)[:, moff ..]^T A'[i-rows][n0 ..], once the
// diagonal block is factored (dd[i]) and the counters w1 >= v1, w2 >= v2 (nf - 1 of them) say that A' is complete.
// foreign: other workgroups wrote A' (agent-scope DMA reads); false: this workgroup did, a moment ago.
template <class TL>
__device__ __forceinline__ bool fl_solve(const fl_ctx& cx, const double* U, double* W, const double* Wt, long Np, int i,
                                         int moff, int n0, const unsigned* dd, const unsigned* w1, unsigned v1,
                                         const unsigned* w2, unsigned v2, int nf, bool foreign, unsigned* tr, int nt,
                                         double* smem) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    if (!fl_wait4(cx, dd + i, 1u, w1, v1, w2, v2, nf)) return false;
    const long k0 = (long)i * SR_NB;
    typename TL::Acc acc;
    acc.zero();
    // U_ii^-1 is upper triangular: column m < 64 has nothing below row 64
    const int kend = TL::T == 128 ? SR_NB : moff + 64;
    // (foreign: the tile of U was last written by other workgroups, with agent-scope stores -- read it the same way)
    if (foreign) TL::mainloop_coh(Wt + k0 * Np + k0 + moff, Np, U + k0 * Np + n0, Np, 0, kend, smem, acc);
    else TL::mainloop(Wt + k0 * Np + k0 + moff, Np, U + k0 * Np + n0, Np, 0, kend, smem, acc);
    double* C = W + (k0 + moff) * Np + n0;
#pragma unroll
    for (int mi = 0; mi < TL::NI; ++mi)
#pragma unroll
        for (int ni = 0; ni < TL::NI; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                C[(long)TL::row(wm, mi, lane, r) * Np + TL::col(wn, ni, lane)] = acc.v[mi][ni][r];
    fl_publish<TL>(tr + (long)i * nt + (n0 >> 6));
    return true;
}


// what a workgroup is to do next
struct fl_task {
    int op;                      // 0 update of a 64-tile (-> ac), 1 solve of a 64-tile, 2 far block: update + solve, 3 far block: solve only, 4 UPD
    int kind;                    // diagnostics (sr_gp_flow_stats)
    int i, j;                    // block row / block column
    int moff, n0;                // row offset inside the block row (64-tiles), first column
    int k_lo, k_hi;              // factor block rows the update takes
    unsigned apt;                // panels block (i, j) must have taken before
};

// decoding state of one workgroup: tasks of a queue come in ascending order
struct fl_state { int sg_c = 0, sg_f = 0, uq = 0, ubase = 0, sg_m = 0; };

__device__ __forceinline__ int fl_upd_row_count(int nb, int PW, int band, int q, int bi) {
    // blocks of row bi that take panel q through a UPD task: the rows of the NEXT panel keep their band out of it
    const int c = bi < (q + 2) * PW ? nb - bi - band - 1 : nb - bi;
    return c > 0 ? c : 0;
}

// queue 0 (critical: diagonal tiles, near updates, near solves), 1 (far blocks), 2 (panel updates); qi-th task of the queue
// for one output
__device__ __forceinline__ fl_task fl_decode(const sr_flow_params& P, fl_state& st, int queue, int qi) {
    const int nb = P.nb, band = P.band, PW = P.panel;
    fl_task T{};
    if (queue == 2) {
        for (;;) {
            int cnt = 0;
            for (int bi = (st.uq + 1) * PW; bi < nb; ++bi) cnt += fl_upd_row_count(nb, PW, band, st.uq, bi);
            if (qi < st.ubase + cnt) break;
            st.ubase += cnt;
            ++st.uq;
        }
        const int q = st.uq;
        int idx = qi - st.ubase;
        int bi = (q + 1) * PW;
        for (;;) {
            const int c = fl_upd_row_count(nb, PW, band, q, bi);
            if (idx < c) break;
            idx -= c;
            ++bi;
        }
        const bool next = bi < (q + 2) * PW;
        T.op = 4; T.kind = next ? 0 : 1;
        T.i = bi; T.j = (next ? bi + band + 1 : bi) + idx;
        T.moff = 0; T.n0 = T.j * SR_NB;
        T.k_lo = q * PW; T.k_hi = (q + 1) * PW;
        T.apt = (unsigned)q;
        return T;
    }
    if (queue == 0) {
        // segment s: the solves of row s, then the updates of row s + 1 (which take the factor rows that exist as soon as a
        // workgroup holds them, and lack one K-step when the solves of row s are through)
        while (st.sg_c + 1 < nb && qi >= P.segs[st.sg_c + 1].start_c) ++st.sg_c;
        const int s = st.sg_c;
        int e = qi - P.segs[s].start_c;
        const int n_sol = 4 * sr_flow_near(nb, s, band);
        const int iu = s + 1;                                // the row whose updates live in this segment
        const bool solve = e < n_sol;
        if (!solve) e -= n_sol;
        if (solve) {
            T.i = s;
            T.op = 1; T.kind = 4; T.j = s + 1 + (e >> 2); T.moff = (e & 2) ? 64 : 0; T.n0 = T.j * SR_NB + (e & 1) * 64;
            return T;
        }
        const int i = iu, p = i / PW;
        T.i = i;
        // the band takes the previous panel's rows AND this panel's rows above row i left-looking, in 64-tiles
        T.k_lo = p >= 1 ? (p - 1) * PW : 0; T.k_hi = i;
        T.apt = p >= 1 ? (unsigned)(p - 1) : 0u;
        T.op = 0;
        if (e < 3) {                                         // the diagonal block's upper tiles: (0, 0), (0, 1), (1, 1)
            T.kind = 2; T.j = i; T.moff = e == 2 ? 64 : 0; T.n0 = i * SR_NB + (e == 0 ? 0 : 64);
            return T;
        }
        e -= 3;
        T.kind = 3; T.j = i + 1 + (e >> 2); T.moff = (e & 2) ? 64 : 0; T.n0 = T.j * SR_NB + (e & 1) * 64;
        return T;
    }
    while (st.sg_f + 1 < nb && qi >= P.segs[st.sg_f + 1].start_f) ++st.sg_f;
    const int i = st.sg_f, p = i / PW, r = i - p * PW;
    const int e = qi - P.segs[i].start_f;
    T.i = i; T.j = i + 1 + sr_flow_near(nb, i, band) + e; T.moff = 0; T.n0 = T.j * SR_NB;
    T.k_lo = p * PW; T.k_hi = i; T.apt = (unsigned)p;
    T.op = r == 0 ? 3 : 2; T.kind = 5;
    return T;
}

}  // namespace

__global__ __launch_bounds__(256, 2) void sr_flow_worker_kernel(sr_flow_params P) {
    __shared__ double smem[srt::SMEM_DOUBLES];
    __shared__ int sh_word[3], sh_task[2];      // (sh_task[1]: the position taken)
    fl_ctx cx{P.flags + SR_FLOW_STATUS, P.timeout, sh_word, P.acq};
    const int nb = P.nb, nt = 2 * nb;
    const long Np = P.Np;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // the Gram matrix was there before this kernel started: the diagonal-block workgroups may go
    if (threadIdx.x == 0) __hip_atomic_store(P.flags + SR_FLOW_GO, P.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // ... and they must be resident before anybody waits for them inside a task
    {
        const unsigned* a0 = P.flags + SR_FLOW_ALIVE;
        if (wave == 0) {
            const bool mine = lane < P.n_out;
            const unsigned long long t0 = wall_clock64();
            unsigned spins = 0;
            int ok = 1;
            for (;;) {
                const unsigned v = mine ? __hip_atomic_load(a0 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : P.epoch;
                if (__ballot(v != P.epoch) == 0ull) break;
                __builtin_amdgcn_s_sleep(4);
                if (fl_give_up(cx, spins, t0)) { ok = 0; break; }
            }
            if (lane == 0) sh_word[1] = ok;
        }
        __syncthreads();
        if (sh_word[1] == 0) return;
        __syncthreads();
    }
    fl_state st;
    unsigned* ctr = P.flags + SR_FLOW_TASK;
#pragma unroll 1
    for (;;) {
        // ---- ONE order, handed out by fetch-and-add (sr_flow.h).  A task that cannot start yet waits inside.
        if (threadIdx.x == 0) {
            sh_task[1] = (int)__hip_atomic_fetch_add(ctr + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sh_word[2] = 0;
        }
        __syncthreads();
        int queue, t = __builtin_amdgcn_readfirstlane(sh_task[1]);
        if (t >= (int)P.total_m * P.n_out) {
            // Nothing left to hand out.  The LAST diagonal block has no task behind it: workgroup 0 keeps the kernel alive until
            // the diagonal-block workgroups have written it (whatever is behind this kernel on the stream -- the triangular
            // inversion -- reads its inverse).  (Found by the first update of a fresh handle: every later one read the
            // previous update's identical block.)
            if (blockIdx.x == 0) {
                unsigned* last = P.flags + SR_FLOW_HDR + (long)(nb - 1);
                for (int d = 0; d < P.n_out; ++d) {
                    const unsigned* w = last + (long)d * sr_flow_words(nb);
                    if (!fl_wait4(cx, w, 1u, w, 1u, w, 1u, 1)) return;
                }
            }
            return;
        }
        {
            // position in the one order -> (kind of task, its number among those)
            const int d = t % P.n_out, qm = t / P.n_out;
            while (st.sg_m + 1 < nb && qm >= P.segs[st.sg_m + 1].start_m) ++st.sg_m;
            const sr_flow_seg S = P.segs[st.sg_m], S1 = P.segs[st.sg_m + 1];
            // [panel updates] [solves of the row] [far blocks of the row] [updates of the next row's band]: the last need the
            // row's factor at every column of their blocks, the outermost of which is a far block of this row
            const int e = qm - S.start_m, n_b = S1.start_b - S.start_b, n_f = S1.start_f - S.start_f;
            const int n_sol = 4 * sr_flow_near(nb, st.sg_m, P.band);
            if (e < n_b) { queue = 2; t = S.start_b + e; }
            else if (e < n_b + n_sol) { queue = 0; t = S.start_c + (e - n_b); }
            else if (e < n_b + n_sol + n_f) { queue = 1; t = S.start_f + (e - n_b - n_sol); }
            else { queue = 0; t = S.start_c + (e - n_b - n_f); }
            t = t * P.n_out + d;
        }
        const unsigned long long t_task = wall_clock64();
        const int d = __builtin_amdgcn_readfirstlane(t % P.n_out);
        fl_task T = fl_decode(P, st, queue, t / P.n_out);
        // (wavefront-uniform by construction; said so that the tile pointers live in SGPRs, as the LDS-DMA code needs them)
        T.op = __builtin_amdgcn_readfirstlane(T.op); T.kind = __builtin_amdgcn_readfirstlane(T.kind);
        T.i = __builtin_amdgcn_readfirstlane(T.i); T.j = __builtin_amdgcn_readfirstlane(T.j);
        T.moff = __builtin_amdgcn_readfirstlane(T.moff); T.n0 = __builtin_amdgcn_readfirstlane(T.n0);
        T.k_lo = __builtin_amdgcn_readfirstlane(T.k_lo); T.k_hi = __builtin_amdgcn_readfirstlane(T.k_hi);
        T.apt = (unsigned)__builtin_amdgcn_readfirstlane((int)T.apt);
        double* U = P.U + (long)d * P.sU;
        double* W = P.W + (long)d * P.sU;
        const double* Wt = P.Wt + (long)d * P.sWt;
        unsigned* dd = P.flags + SR_FLOW_HDR + (long)d * sr_flow_words(nb);
        unsigned* ac = dd + nb;
        unsigned* tr = ac + (long)nb * nt;
        unsigned* ap = tr + (long)nb * nt;
        unsigned* apw = ap + (long)T.i * nb + T.j;
        const long r0 = (long)T.i * SR_NB + T.moff;
        bool ok;
        // Issue priority on a CU shared with another workgroup: the band (between two diagonal blocks) first, then whatever the
        // NEXT block row waits for column by column -- a far block's update + solve is a link of a chain over the rows of its
        // column, and at the matrix pipe's half rate (beside a UPD task) that link is longer than a block step -- then the UPD
        // tasks of the next panel's rows, last the bulk.  (lab: SR_FLOW_PRIO=0 only the band)
        if (T.kind >= 2 && T.kind <= 4) __builtin_amdgcn_s_setprio(3);
        else if (T.kind == 5 && P.prio) __builtin_amdgcn_s_setprio(2);
        else if (T.kind == 0 && P.prio) __builtin_amdgcn_s_setprio(1);
        if (T.op == 0) {
            ok = fl_accum<fl_tile64>(cx, U, W, Np, r0, T.n0, T.k_lo, T.k_hi, tr, nt, apw, T.apt, smem);
            if (ok) fl_publish1(ac + (long)T.i * nt + (T.n0 >> 6));
        } else if (T.op == 1) {
            const unsigned* a0 = ac + (long)T.i * nt + (T.n0 >> 6);
            ok = fl_solve<fl_tile64>(cx, U, W, Wt, Np, T.i, T.moff, T.n0, dd, a0, 2u, a0, 2u, T.i > 0 ? 2 : 1, true, tr, nt, smem);
        } else if (T.op == 2) {
            ok = fl_accum<fl_tile128>(cx, U, W, Np, r0, T.n0, T.k_lo, T.k_hi, tr, nt, apw, T.apt, smem);
            if (ok) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the tile is out (sc1 stores) before the DMA reads it back
                ok = fl_solve<fl_tile128>(cx, U, W, Wt, Np, T.i, 0, T.n0, dd, dd + T.i, 1u, dd + T.i, 1u, 1, false, tr, nt, smem);
            }
        } else if (T.op == 3) {
            ok = fl_solve<fl_tile128>(cx, U, W, Wt, Np, T.i, 0, T.n0, dd, apw, T.apt, apw, T.apt, T.apt > 0 ? 2 : 1, true, tr, nt, smem);
        } else {
            ok = fl_accum<fl_tile128>(cx, U, W, Np, r0, T.n0, T.k_lo, T.k_hi, tr, nt, apw, T.apt, smem);
            if (ok) fl_publish1(apw);
        }
        __builtin_amdgcn_s_setprio(0);
        if (!ok) return;
        // diagnostics (sr_gp_flow_stats): per kind of task -- 0 update of the next panel's rows, 1 of the rows behind it, 2
        // diagonal tiles, 3 near updates, 4 near solves, 5 far blocks -- how many, ticks inside, ticks of those spent waiting
        // (including the search for the task)
        if (threadIdx.x == 0) {
            unsigned* stt = P.flags + SR_FLOW_STATS + 4 * T.kind;
            __hip_atomic_fetch_add(stt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(stt + 1, (unsigned)(wall_clock64() - t_task), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(stt + 2, (unsigned)sh_word[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();                          // (sh_task / sh_word[2] are rewritten by the next search)
        // Past block row `exit_row` the chain bounds the factorisation and a quarter of the workgroups keep up with it: the
        // others leave, and their CUs take the stages of the triangular inversion that wait behind their gates
        // (sr_capi_update.hip) -- those kernels cannot start on a CU that holds two of these workgroups (LDS).
        if (queue != 2 && T.i >= P.exit_row && (int)blockIdx.x >= P.keep) return;
    }
}

long sr_flow_plan(int nb, int band, int panel, sr_flow_seg* segs, long* total_far, long* total_upd, long* total_m) {
    // the update queue's index range per panel q: [ub[q], ub[q + 1]), its first na[q] tasks the far blocks of the next panel's rows
    std::vector<long> ub(1, 0), na;
    for (int q = 0; (q + 1) * panel < nb; ++q) {
        long a = 0, b = 0;
        for (int bi = (q + 1) * panel; bi < nb; ++bi) {
            const bool next = bi < (q + 2) * panel;
            const int c = next ? nb - bi - band - 1 : nb - bi;
            (next ? a : b) += c > 0 ? c : 0;
        }
        na.push_back(a);
        ub.push_back(ub.back() + a + b);
    }
    long sc = 0, sf = 0, sm = 0;
    for (int i = 0; i <= nb; ++i) {
        const int p = i / panel, r = i % panel;
        // updates by panel p - 1 handed out with row i: all of those the rows of panel p need with its first row, of the
        // others an equal share with every row
        long sb = i < nb ? 0 : ub.back();
        if (i < nb && p >= 1) {
            const int q = p - 1;
            const int rows = std::min(nb, (p + 1) * panel) - p * panel;
            const long nbk = ub[q + 1] - ub[q] - na[q];
            sb = ub[q] + (r == 0 ? 0 : na[q] + nbk * r / rows);
        }
        segs[i].start_c = (int)sc; segs[i].start_f = (int)sf; segs[i].start_b = (int)sb; segs[i].start_m = 0;
        if (i == nb) break;
        const int near = sr_flow_near(nb, i, band);
        sc += 4 * near + (i + 1 < nb ? 3 + 4 * sr_flow_near(nb, i + 1, band) : 0);       // solves of row i, updates of row i + 1
        sf += nb - 1 - i - near;
    }
    for (int i = 0; i < nb; ++i) {
        segs[i].start_m = (int)sm;
        sm += (segs[i + 1].start_b - segs[i].start_b) + (segs[i + 1].start_c - segs[i].start_c) + (segs[i + 1].start_f - segs[i].start_f);
    }
    segs[nb].start_m = (int)sm;
    *total_far = sf;
    *total_upd = ub.back();
    *total_m = sm;
    return sc;
}

// ------------------------------------------------------------------------------------------------
// Gate in front of a stage of the triangular inversion that runs BESIDE the tile flow (sr_capi_update.hip): one workgroup
// that waits until the run `epoch` has started and every factor row above block row X is final for every output -- row X - 1
// solved at all its columns (tr >= 2; rows above it were final before it could be), or, for X = nb, the last diagonal block
// factored.  The stage's kernels behind it on the stream then start with everything they read visible.  A time-out or a
// raised status word lets it through (the host repeats the update by launches).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sr_flow_gate_kernel(unsigned* flags, unsigned epoch, int n_out, int nb, int X,
                                                           unsigned long long timeout) {
    const int nt = 2 * nb;
    const unsigned long long t0 = wall_clock64();
    unsigned* status = flags + SR_FLOW_STATUS;
    unsigned spins = 0;
    while (__hip_atomic_load(flags + SR_FLOW_GO, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
        __builtin_amdgcn_s_sleep(32);
        if ((++spins & 63) == 0 && wall_clock64() - t0 > timeout) return;
    }
    for (;;) {
        int ok = 1;
        for (int d = 0; d < n_out; ++d) {
            const unsigned* dd = flags + SR_FLOW_HDR + (long)d * sr_flow_words(nb);
            const unsigned* tr = dd + nb + (long)nb * nt;
            if (X >= nb) {
                if (threadIdx.x == 0 && __hip_atomic_load(dd + nb - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 1u) ok = 0;
            } else {
                for (int c = 2 * X + (int)threadIdx.x; c < nt; c += 256)
                    if (__hip_atomic_load(tr + (long)(X - 1) * nt + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 2u) ok = 0;
            }
        }
        if (__syncthreads_and(ok)) return;
        __builtin_amdgcn_s_sleep(64);
        ++spins;
        int stop = 0;
        if (threadIdx.x == 0 && (spins & 15) == 0) {
            if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) stop = 1;
            else if (wall_clock64() - t0 > timeout) {
                __hip_atomic_store(status, 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                stop = 1;
            }
        }
        if (__syncthreads_or(stop)) return;
    }
}

int sr_launch_flow_gate(unsigned* flags, unsigned epoch, int n_out, int nb, int X, double timeout_s, hipStream_t s) {
    hipLaunchKernelGGL(sr_flow_gate_kernel, dim3(1), dim3(256), 0, s, flags, epoch, n_out, nb, X, (unsigned long long)(timeout_s * 1e8));
    SR_HIP(hipGetLastError());
    return SR_OK;
}

int sr_launch_flow_workers(const sr_flow_params& p, int wgs, hipStream_t s) {
    hipLaunchKernelGGL(sr_flow_worker_kernel, dim3(wgs), dim3(256), 0, s, p);
    SR_HIP(hipGetLastError());
    return SR_OK;
}
