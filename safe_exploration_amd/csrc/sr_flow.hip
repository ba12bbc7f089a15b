// sr_flow.hip -- tile-flow Cholesky of the model update at the chain-bound sizes (round 6).
//
// The launched form (sr_capi_update.hip) walks the block chain with three dependent launches per 128 rows -- diagonal block,
// block-row solve, update of the next rows -- beside a trailing update on another stream, and every one of the chain's kernels
// then waits for CUs the trailing update holds (block step ~100 us in situ against ~55 alone).  Here the whole
// factorisation K = U^T U is ONE resident kernel of GEMM workgroups plus one resident diagonal-block workgroup per output
// (sr_flow_diag_server_kernel, sr_factor.hip), and the dependencies travel through counters in device memory (sr_flow.h):
//
//   * LEFT-LOOKING by tile: the task "update (i, tile)" owns one tile of block row i of the Gram matrix for its whole life:
//     acc = A_tile - sum_{k < i} U[k][rows]^T U[k][cols], one MFMA main loop per run of factor rows that are final (tr[k][..]
//     says so), the tile stays in the accumulators in between and is written ONCE.  Every factor row is therefore applied as
//     soon as it exists; by the time the chain reaches block row i its tiles lack one K = 128 step.
//   * "solve (i, tile)": U[i][tile] = U_ii^-T A'[i][tile] as a product with the inverse the diagonal-block workgroup leaves
//     (dd[i]), out of place into W, as in the launched form.
//   * near the diagonal (`band` blocks right of it) tiles are 64 x 64 -- the two products between two diagonal blocks are on
//     the critical path and a 64-tile's K = 128 step is 3.4 us of a CU against 14 -- further out 128 x 128 (16 flop per operand
//     byte instead of 8).
//   * tasks are handed out in ONE order -- block row by block row, updates before solves -- through an atomic counter.  A task
//     depends only on tasks in front of it in that order (and on the diagonal-block workgroup, which depends on such tasks),
//     and whoever holds a task is running: no deadlock whatever the number of resident workgroups.  The workgroups that are
//     ahead of the chain wait inside their tasks with the tile in registers; that is the look-ahead.
//
// Memory model.  A producer's tile leaves behind an agent-scope release (L2 write-back) before its counter is raised.  Every
// word a consumer reads is written exactly once per run, and only AFTER the kernel started; no line of it can sit in a cache
// before its counter says so (tiles are 512-byte aligned rows: no line is shared between tiles), so the consumer's loads
// behind the counter need no invalidate -- `acq` = 0; `acq` = 1 puts the agent-scope acquire of the language model behind every
// wait (an L2 invalidate of the whole XCD each time; lab switch SR_FLOW_ACQ for the A/B).
#include "sr_mfma_tile.h"
#include "sr_flow.h"

namespace {

struct fl_tile64 {
    using Acc = srt64::Acc;
    static constexpr int T = 64, NI = 2;
    static __device__ __forceinline__ void mainloop(const double* A, long lda, const double* B, long ldb, int k0, int k1,
                                                    double* smem, Acc& acc) {
        srt64::mainloop_tn_pipe(A, lda, B, ldb, k0, k1, smem, acc);
    }
    static __device__ __forceinline__ int row(int wm, int mi, int lane, int r) { return srt64::acc_row(wm, mi, lane, r); }
    static __device__ __forceinline__ int col(int wn, int ni, int lane) { return srt64::acc_col(wn, ni, lane); }
};
struct fl_tile128 {
    using Acc = srt::Acc;
    static constexpr int T = 128, NI = 4;
    static __device__ __forceinline__ void mainloop(const double* A, long lda, const double* B, long ldb, int k0, int k1,
                                                    double* smem, Acc& acc) {
        srt::mainloop_tn_pipe<false>(A, lda, B, ldb, k0, k1, smem, acc);
    }
    static __device__ __forceinline__ int row(int wm, int mi, int lane, int r) { return srt::acc_row(wm, mi, lane, r); }
    static __device__ __forceinline__ int col(int wn, int ni, int lane) { return srt::acc_col(wn, ni, lane); }
};
static_assert(srt::SMEM_DOUBLES == srt64::SMEM_DOUBLES, "one LDS buffer for both tiles");

struct fl_ctx {
    unsigned* status;
    unsigned long long timeout;
    int* sh;                     // one LDS word: what wavefront 0 found
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
        if (lane == 0) *cx.sh = n;
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
        if (lane == 0) *cx.sh = ok;
    }
    __syncthreads();
    const bool ok = *cx.sh != 0;
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

// left-looking update of the tile at rows 128 i + moff, columns n0 of U (in place), from the factor rows in W
template <class TL>
__device__ __forceinline__ bool fl_update(const fl_ctx& cx, double* U, const double* W, long Np, int i, int moff, int n0,
                                          const unsigned* tr, unsigned* ac, int nt, double* smem) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const long r0 = (long)i * SR_NB + moff;
    double* C = U + r0 * Np + n0;
    typename TL::Acc acc;
#pragma unroll
    for (int mi = 0; mi < TL::NI; ++mi)
#pragma unroll
        for (int ni = 0; ni < TL::NI; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                acc.v[mi][ni][r] = -C[(long)TL::row(wm, mi, lane, r) * Np + TL::col(wn, ni, lane)];
    const int ca = (int)(r0 >> 6), cb = n0 >> 6;
    const int ca1 = TL::T == 128 ? ca + 1 : ca, cb1 = TL::T == 128 ? cb + 1 : cb;
    int k = 0;
    while (k < i) {
        const int n = fl_rows_ready(cx, tr, nt, k, i, ca, ca1, cb, cb1);
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
                C[(long)TL::row(wm, mi, lane, r) * Np + TL::col(wn, ni, lane)] = -acc.v[mi][ni][r];
    fl_publish<TL>(ac + (long)i * nt + cb);
    return true;
}

// block-row solve of the tile at rows 128 i + moff, columns n0:  W_tile = (U_ii^-1)[:, moff ..]^T A'[i-rows][n0 ..]
template <class TL>
__device__ __forceinline__ bool fl_solve(const fl_ctx& cx, const double* U, double* W, const double* Wt, long Np, int i,
                                         int moff, int n0, const unsigned* dd, const unsigned* ac, unsigned* tr, int nt,
                                         double* smem) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int cb = n0 >> 6;
    const unsigned* a0 = ac + (long)i * nt + cb;
    const int nf = i == 0 ? 1 : (TL::T == 128 ? 3 : 2);
    if (!fl_wait4(cx, dd + i, 1u, a0, 2u, a0 + 1, 2u, nf)) return false;
    const long k0 = (long)i * SR_NB;
    typename TL::Acc acc;
    acc.zero();
    // U_ii^-1 is upper triangular: column m < 64 has nothing below row 64
    const int kend = TL::T == 128 ? SR_NB : moff + 64;
    TL::mainloop(Wt + k0 * Np + k0 + moff, Np, U + k0 * Np + n0, Np, 0, kend, smem, acc);
    double* C = W + (k0 + moff) * Np + n0;
#pragma unroll
    for (int mi = 0; mi < TL::NI; ++mi)
#pragma unroll
        for (int ni = 0; ni < TL::NI; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                C[(long)TL::row(wm, mi, lane, r) * Np + TL::col(wn, ni, lane)] = acc.v[mi][ni][r];
    fl_publish<TL>(tr + (long)i * nt + cb);
    return true;
}

}  // namespace

__global__ __launch_bounds__(256, 2) void sr_flow_worker_kernel(sr_flow_params P) {
    __shared__ double smem[srt::SMEM_DOUBLES];
    __shared__ int sh_word, sh_task;
    fl_ctx cx{P.flags + SR_FLOW_STATUS, P.timeout, &sh_word, P.acq};
    const int nb = P.nb, nt = 2 * nb, band = P.band;
    const long Np = P.Np;
    // the diagonal-block workgroups must be resident before anybody waits for them inside a task
    {
        const unsigned* a0 = P.flags + SR_FLOW_ALIVE;
        if (threadIdx.x < 64) {
            const int lane = threadIdx.x;
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
            if (lane == 0) sh_word = ok;
        }
        __syncthreads();
        if (sh_word == 0) return;
        __syncthreads();
    }
    int row = 0;
    long base = 0;               // first task of block row `row`
    const long all = P.total * P.n_out;
#pragma unroll 1
    for (;;) {
        if (threadIdx.x == 0)
            sh_task = (int)__hip_atomic_fetch_add(P.flags + SR_FLOW_TASK, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const long t = sh_task;
        if (t >= all) return;
        if (t == 0 && threadIdx.x == 0)          // the Gram matrix was there before this kernel started
            __hip_atomic_store(P.flags + SR_FLOW_GO, P.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int d = (int)(t % P.n_out);
        long q = t / P.n_out;
        while (q >= base + sr_flow_nacc(nb, row, band) + sr_flow_ntr(nb, row, band)) {
            base += sr_flow_nacc(nb, row, band) + sr_flow_ntr(nb, row, band);
            ++row;
        }
        const int i = row;
        int e = (int)(q - base);
        double* U = P.U + (long)d * P.sU;
        double* W = P.W + (long)d * P.sU;
        const double* Wt = P.Wt + (long)d * P.sWt;
        unsigned* dd = P.flags + SR_FLOW_HDR + (long)d * sr_flow_words(nb);
        unsigned* ac = dd + nb;
        unsigned* tr = ac + (long)nb * nt;
        const int nacc = sr_flow_nacc(nb, i, band), near = sr_flow_near(nb, i, band);
        bool ok;
        if (e < nacc) {
            if (e < 3) {
                // the diagonal block's upper tiles: (0, 0), (0, 1), (1, 1)
                __builtin_amdgcn_s_setprio(2);
                ok = fl_update<fl_tile64>(cx, U, W, Np, i, e == 2 ? 64 : 0, i * SR_NB + (e == 0 ? 0 : 64), tr, ac, nt, smem);
                __builtin_amdgcn_s_setprio(0);
            } else if ((e -= 3) < 4 * near) {
                const int j = i + 1 + (e >> 2);
                ok = fl_update<fl_tile64>(cx, U, W, Np, i, (e & 2) ? 64 : 0, j * SR_NB + (e & 1) * 64, tr, ac, nt, smem);
            } else {
                const int j = i + 1 + near + (e - 4 * near);
                ok = fl_update<fl_tile128>(cx, U, W, Np, i, 0, j * SR_NB, tr, ac, nt, smem);
            }
        } else {
            e -= nacc;
            if (e < 4 * near) {
                const int j = i + 1 + (e >> 2);
                if (e < 4) __builtin_amdgcn_s_setprio(2);
                ok = fl_solve<fl_tile64>(cx, U, W, Wt, Np, i, (e & 2) ? 64 : 0, j * SR_NB + (e & 1) * 64, dd, ac, tr, nt, smem);
                __builtin_amdgcn_s_setprio(0);
            } else {
                const int j = i + 1 + near + (e - 4 * near);
                ok = fl_solve<fl_tile128>(cx, U, W, Wt, Np, i, 0, j * SR_NB, dd, ac, tr, nt, smem);
            }
        }
        if (!ok) return;
    }
}

int sr_launch_flow_workers(const sr_flow_params& p, int wgs, hipStream_t s) {
    hipLaunchKernelGGL(sr_flow_worker_kernel, dim3(wgs), dim3(256), 0, s, p);
    SR_HIP(hipGetLastError());
    return SR_OK;
}
