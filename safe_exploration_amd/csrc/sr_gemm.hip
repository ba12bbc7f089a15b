// sr_gemm.hip -- the TN GEMMs of the model update and of the row append on the fp64 matrix cores (both workgroup tiles of
// sr_mfma_tile.h): plain, split-K, upper-block-triangle (trailing update of the Cholesky) and job-table (one launch per
// level of the recursive triangular inversion).  Launch plan: sr_capi_update.hip / sr_capi_append.hip.
#include "sr_mfma_tile.h"
#include <cstdlib>
#ifndef SR_T64_PIPE
#define SR_T64_PIPE 1     /* 64 x 64 tile: the pipelined loop of round 5 (0: the loop of round 3, for A/B builds) */
#endif

// ------------------------------------------------------------------------------------------------
// TN GEMMs on the fp64 matrix cores, on either workgroup tile of sr_mfma_tile.h
// ------------------------------------------------------------------------------------------------
struct sr_tile128 {
    using Acc = srt::Acc;
    static constexpr int T = 128, NI = 4, SMEM = srt::SMEM_DOUBLES, WPS = 2;
    static __device__ __forceinline__ void mainloop(const double* A, long lda, const double* B, long ldb, int k0,
                                                    int k1, double* smem, Acc& acc) {
        // LDS-DMA staging (+5 % over register staging).  (Four stages of 8 k-rows in the same LDS -- three k-tiles in
        // flight, hand-placed vmcnt -- measured the same: 42.9 TF at K = 256, 51.4 at K = 1024, C4 63.2 against 64.2.
        // What these products lose, they lose to the tail of the grid, not to the pipeline of a tile.)
        // Round 5: the pipelined loop (barrier under the MFMA stream; sr_mfma_tile.h).
        srt::mainloop_tn_pipe<false>(A, lda, B, ldb, k0, k1, smem, acc);
    }
    static __device__ __forceinline__ int row(int wm, int mi, int lane, int r) { return srt::acc_row(wm, mi, lane, r); }
    static __device__ __forceinline__ int col(int wn, int ni, int lane) { return srt::acc_col(wn, ni, lane); }
};
struct sr_tile64 {
    using Acc = srt64::Acc;
    static constexpr int T = 64, NI = 2, SMEM = srt64::SMEM_DOUBLES, WPS = 2;
    static __device__ __forceinline__ void mainloop(const double* A, long lda, const double* B, long ldb, int k0,
                                                    int k1, double* smem, Acc& acc) {
        static_assert(true, "");
        if (SR_T64_PIPE) srt64::mainloop_tn_pipe(A, lda, B, ldb, k0, k1, smem, acc);
        else srt64::mainloop_tn(A, lda, B, ldb, k0, k1, smem, acc);
    }
    static __device__ __forceinline__ int row(int wm, int mi, int lane, int r) { return srt64::acc_row(wm, mi, lane, r); }
    static __device__ __forceinline__ int col(int wn, int ni, int lane) { return srt64::acc_col(wn, ni, lane); }
};

// C = alpha A^T B + beta C on one tile at (m0, n0), k in [k_beg, k_end)
template <class TL>
__device__ __forceinline__ void sr_gemm_tile(const double* __restrict__ A, long lda, const double* __restrict__ B,
                                             long ldb, double* C, long ldc, int m0, int n0, int k_beg, int k_end,
                                             double alpha, double beta, double* smem) {
    typename TL::Acc acc;
    acc.zero();
    TL::mainloop(A + m0, lda, B + n0, ldb, k_beg, k_end, smem, acc);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    if (beta != 0.0) {
        // read-modify-write: all loads of a row of MFMA tiles first, then the arithmetic and the stores (with the
        // test on beta inside the element loop every element was a load -> s_waitcnt vmcnt(0) -> store round trip of
        // its own: 16 resp. 64 dependent global-memory latencies per tile)
#pragma unroll
        for (int mi = 0; mi < TL::NI; ++mi) {
            double old[TL::NI][4];
#pragma unroll
            for (int ni = 0; ni < TL::NI; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    old[ni][r] = C[(m0 + TL::row(wm, mi, lane, r)) * ldc + n0 + TL::col(wn, ni, lane)];
#pragma unroll
            for (int ni = 0; ni < TL::NI; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    C[(m0 + TL::row(wm, mi, lane, r)) * ldc + n0 + TL::col(wn, ni, lane)] =
                        fma(alpha, acc.v[mi][ni][r], beta * old[ni][r]);
        }
    } else {
#pragma unroll
        for (int mi = 0; mi < TL::NI; ++mi)
#pragma unroll
            for (int ni = 0; ni < TL::NI; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    C[(m0 + TL::row(wm, mi, lane, r)) * ldc + n0 + TL::col(wn, ni, lane)] = alpha * acc.v[mi][ni][r];
    }
}

// rectangular grid; mode as documented in sr_common.h (k ranges at the tile's own granularity)
template <class TL>
__global__ __launch_bounds__(256, TL::WPS) void sr_gemm_tn_kernel(
    const double* __restrict__ A, long lda, const double* __restrict__ B, long ldb, double* C, long ldc, int K,
    double alpha, double beta, int mode, int prio, sr_batch bt) {
    __shared__ double smem[TL::SMEM];
    if (prio) __builtin_amdgcn_s_setprio(3);       // critical-path product: win the issue arbitration on a shared SIMD
    A += (long)blockIdx.z * bt.sA; B += (long)blockIdx.z * bt.sB; C += (long)blockIdx.z * bt.sC;
    const int m0 = blockIdx.y * TL::T;
    const int n0 = blockIdx.x * TL::T;
    if (mode == 1 && (n0 & ~127) < (m0 & ~127)) return;        // triangular structure is defined on 128-blocks
    const int k_beg = (mode == 2) ? (n0 & ~127) : ((mode == 4) ? (m0 & ~127) : 0);
    const int k_end = (mode == 3) ? min(K, (m0 & ~127) + 128) : K;
    sr_gemm_tile<TL>(A, lda, B, ldb, C, ldc, m0, n0, k_beg, k_end, alpha, beta, smem);
}

// One fp64 MFMA holds its SIMD for 64 cycles: a 128 x 128 tile with K = 128 is 14 us of one CU, whatever else
// happens.  Products of few tiles are therefore latency-bound (the block row and the look-ahead row of the
// Cholesky sit on its critical path) or balance-bound (triangular k ranges); they take the 64 x 64 tile.
static long sr_env_long(const char* name, long dflt) { return sr_lab_env(name, dflt); }     // (lab build: scripts/refit_ab.py)
static inline bool sr_use_tile64(long tiles128, int K = 0) {
    static const long thr = sr_env_long("SR_T64_THR", 1024);           // (measurements)
    // ... unless K is long: then a grid that occupies the chip at least once is throughput-bound and the 64-tile's 8 flop
    // per operand byte is the limit (the in-panel updates of the N = 50000 factorisation -- 128 rows x 50000 columns, K up
    // to 2944: 21 TF on 64-tiles)
    if (K >= 768 && tiles128 >= 256) return false;
    return tiles128 < thr;
}
// ... but a 64 x 64 tile moves 8 bytes of operands per 8 flop (K-independent): a grid of them that fills the chip is
// bound by L2 / fabric bandwidth (N = 5000, K = 256 bulk update of two outputs: 1.7 GB in 254 us = 6.8 TB/s, 21 TF per
// output).  THROUGHPUT-bound products (bulk trailing update, the big levels of the inversion) therefore take the
// 128-tile (16 flop per byte) as soon as there are enough of them to occupy the chip once.
static inline bool sr_use_tile64_bulk(long tiles128) {
    static const long thr = sr_env_long("SR_T64_BULK_THR", 192);
    return tiles128 < thr;
}
static inline bool sr_use_tile64_jobs(long tiles128) {
    static const long thr = sr_env_long("SR_T64_JOBS_THR", 1024);
    return tiles128 < thr;
}

int sr_launch_gemm_tn(const double* A, long lda, const double* B, long ldb, double* C, long ldc,
                      int M, int N, int K, double alpha, double beta, int mode, hipStream_t s, int prio,
                      const sr_batch* btp) {
    SR_CHECK(M % srt::BM == 0 && N % srt::BN == 0 && K % srt::BK == 0 && M > 0 && N > 0, SR_EINVAL,
             "gemm_tn: M=%d N=%d K=%d must be tile multiples", M, N, K);
    const sr_batch bt = btp ? *btp : sr_batch{};
    if (sr_use_tile64((long)(M / 128) * (N / 128) * bt.n, K))
        hipLaunchKernelGGL(sr_gemm_tn_kernel<sr_tile64>, dim3(N / 64, M / 64, bt.n), dim3(256), 0, s, A, lda, B, ldb, C, ldc,
                           K, alpha, beta, mode, prio, bt);
    else
        hipLaunchKernelGGL(sr_gemm_tn_kernel<sr_tile128>, dim3(N / 128, M / 128, bt.n), dim3(256), 0, s, A, lda, B, ldb, C,
                           ldc, K, alpha, beta, mode, prio, bt);
    SR_HIP(hipGetLastError());
    return SR_OK;
}

// ------------------------------------------------------------------------------------------------
// Split-K form for THIN products (the row append: M = Np rows, N = 128 columns, K up to Np): the plain kernel has
// Np / 64 * 2 workgroups there, the longest of which walks all of K (227 us at Np = 5120); G = U12^T U12 is ONE
// 128 x 128 tile with K = Np (209 us).  grid.z = K-slices of `ks` rows; slice z writes its (possibly empty: zeros)
// contribution to part + z * M * ldc, sr_sum_slices_kernel adds the slices in order (deterministic).
// ------------------------------------------------------------------------------------------------
template <class TL>
__global__ __launch_bounds__(256, TL::WPS) void sr_gemm_tn_splitk_kernel(
    const double* __restrict__ A, long lda, const double* __restrict__ B, long ldb, double* part, long ldc, int M, int K,
    int ks, double alpha, int mode) {
    __shared__ double smem[TL::SMEM];
    const int m0 = blockIdx.y * TL::T;
    const int n0 = blockIdx.x * TL::T;
    int k_beg = (mode == 2) ? (n0 & ~127) : ((mode == 4) ? (m0 & ~127) : 0);
    int k_end = (mode == 3) ? min(K, (m0 & ~127) + 128) : K;
    k_beg = max(k_beg, (int)blockIdx.z * ks);
    k_end = min(k_end, ((int)blockIdx.z + 1) * ks);
    if (k_end < k_beg) k_end = k_beg;
    sr_gemm_tile<TL>(A, lda, B, ldb, part + (long)blockIdx.z * M * ldc, ldc, m0, n0, k_beg, k_end, alpha, 0.0, smem);
}

__global__ __launch_bounds__(256) void sr_sum_slices_kernel(const double* __restrict__ part, long stride, int nsl,
                                                            double* __restrict__ out, long n) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    double v = 0.0;
    for (int z = 0; z < nsl; ++z) v += part[(long)z * stride + e];
    out[e] = v;
}

// C (M x N, ldc == N: the slices are contiguous copies of it) = alpha A^T B restricted by `mode` as in sr_launch_gemm_tn
int sr_launch_gemm_tn_splitk(const double* A, long lda, const double* B, long ldb, double* C, int M, int N, int K,
                             int ks, double alpha, int mode, double* part, hipStream_t s) {
    SR_CHECK(M % srt::BM == 0 && N % srt::BN == 0 && K % srt::BK == 0 && ks % 128 == 0 && ks > 0, SR_EINVAL,
             "gemm_tn_splitk: M=%d N=%d K=%d ks=%d", M, N, K, ks);
    const int nsl = (K + ks - 1) / ks;
    hipLaunchKernelGGL(sr_gemm_tn_splitk_kernel<sr_tile64>, dim3(N / 64, M / 64, nsl), dim3(256), 0, s, A, lda, B, ldb,
                       part, (long)N, M, K, ks, alpha, mode);
    SR_HIP(hipGetLastError());
    const long n = (long)M * N;
    hipLaunchKernelGGL(sr_sum_slices_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, part, n, nsl, C, n);
    SR_HIP(hipGetLastError());
    return SR_OK;
}

// ------------------------------------------------------------------------------------------------
// Upper block triangle only (mode 1 above) on a LINEAR grid: tile b -> (m, n >= m), rows of tn - m tiles.
// The rectangular grid of the trailing update starts (and retires) tm*tn/2 empty workgroups; at N = 50000
// that is 47 000 of them per panel.  C (op)= alpha A^T B + beta C on the tiles n0 >= m0.
// ------------------------------------------------------------------------------------------------
// tile (m, n) of linear index b in the row-major enumeration of the upper triangle: row m holds tiles
// [c(m), c(m+1)), c(m) = m tn - m (m - 1) / 2, columns n = m .. tn - 1
__device__ __forceinline__ void sr_upper_index(long b, int tn, int& m, int& n) {
    m = (int)((2.0 * tn + 1.0 - sqrt((2.0 * tn + 1.0) * (2.0 * tn + 1.0) - 8.0 * (double)b)) * 0.5);
    if (m < 0) m = 0;
    if (m > tn) m = tn;
    while (m < tn && (long)(m + 1) * tn - (long)(m + 1) * m / 2 <= b) ++m;      // b past the end: m = tn
    while (m > 0 && (long)m * tn - (long)m * (m - 1) / 2 > b) --m;
    n = m + (int)(b - ((long)m * tn - (long)m * (m - 1) / 2));
}

// order 0: tiles in row-major order of the upper triangle (small grids).
// order 1: XCD-aware super-tiles.  Workgroup b runs on XCD b % 8 (dispatch is round-robin over the XCDs), each XCD has
//   its own 4 MiB L2.  XCD x therefore takes the super-tiles s = x, x + 8, ... of 8 x 8 tiles, 64 consecutive
//   workgroups of ITS sequence b / 8 per super-tile -- just the 64 workgroups its 32 CUs hold at a time: they walk k
//   together, every A tile row is shared by 8 of them and every B tile row by 8.  Without it a K = 1024 update
//   streams 2 MB of operands per 33.5 MFlop tile (16 flop/B: 3.3 TB/s at the measured 55 TF, i.e. bound by the
//   fabric, not by the matrix cores).
template <class TL>
__global__ __launch_bounds__(256, TL::WPS) void sr_gemm_tn_upper_kernel(
    const double* __restrict__ A, long lda, const double* __restrict__ B, long ldb, double* C, long ldc,
    int K, int tm, int tn, double alpha, double beta, int prio, int order, sr_batch bt) {
    __shared__ double smem[TL::SMEM];
    if (prio) __builtin_amdgcn_s_setprio(3);
    A += (long)blockIdx.y * bt.sA; B += (long)blockIdx.y * bt.sB; C += (long)blockIdx.y * bt.sC;
    int m, n;
    if (order == 0) {
        sr_upper_index(blockIdx.x, tn, m, n);
    } else {
        const long b = blockIdx.x;
        const int xcd = (int)(b & 7);
        const long l = b >> 3;
        const long st = (l >> 6) * 8 + xcd;              // super-tile of this workgroup
        const int stn = (tn + 7) >> 3;
        int sm, sn;
        sr_upper_index(st, stn, sm, sn);
        const int w = (int)(l & 63);
        m = sm * 8 + (w >> 3);
        n = sn * 8 + (w & 7);
        if (sm >= ((tm + 7) >> 3) || m >= tm || n >= tn || n < m) return;
    }
    sr_gemm_tile<TL>(A, lda, B, ldb, C, ldc, m * TL::T, n * TL::T, 0, K, alpha, beta, smem);
}

// C: M x N with only the tiles n0 >= m0 touched (M <= N, both multiples of 128).  With the 64 x 64 tile the
// lower-left quarter of every diagonal 128-block stays untouched as well: nothing reads it (the diagonal-block
// kernel loads the upper triangle only).
int sr_launch_gemm_tn_upper(const double* A, long lda, const double* B, long ldb, double* C, long ldc,
                            int M, int N, int K, double alpha, double beta, hipStream_t s, int prio, int order,
                            const sr_batch* btp) {
    const sr_batch bt = btp ? *btp : sr_batch{};
    SR_CHECK(M % srt::BM == 0 && N % srt::BN == 0 && K % srt::BK == 0 && M > 0 && N >= M, SR_EINVAL,
             "gemm_tn_upper: M=%d N=%d K=%d", M, N, K);
    const long tm128 = M / 128, tn128 = N / 128;
    const long tiles128 = (tm128 * tn128 - tm128 * (tm128 - 1) / 2) * bt.n;
    const bool t64 = prio ? sr_use_tile64(tiles128, K) : sr_use_tile64_bulk(tiles128);
    const long tm = t64 ? M / 64 : tm128, tn = t64 ? N / 64 : tn128;
    if (order < 0) order = tiles128 >= 4096 ? 1 : 0;     // super-tiles pay once the grid is many times the chip
    long blocks;
    if (order == 0) {
        blocks = tm * tn - tm * (tm - 1) / 2;
    } else {
        const long stm = (tm + 7) / 8, stn = (tn + 7) / 8;
        const long nst = stm * stn - stm * (stm - 1) / 2;      // super-tiles (sm, sn >= sm)
        blocks = ((nst + 7) / 8) * 8 * 64;
    }
    SR_CHECK(blocks < 2147483647L, SR_EINVAL, "gemm_tn_upper: grid too large");
    if (t64)
        hipLaunchKernelGGL(sr_gemm_tn_upper_kernel<sr_tile64>, dim3((unsigned)blocks, bt.n), dim3(256), 0, s, A, lda, B, ldb,
                           C, ldc, K, (int)tm, (int)tn, alpha, beta, prio, order, bt);
    else
        hipLaunchKernelGGL(sr_gemm_tn_upper_kernel<sr_tile128>, dim3((unsigned)blocks, bt.n), dim3(256), 0, s, A, lda, B, ldb,
                           C, ldc, K, (int)tm, (int)tn, alpha, beta, prio, order, bt);
    SR_HIP(hipGetLastError());
    return SR_OK;
}

// ------------------------------------------------------------------------------------------------
// A LIST of independent TN products in one launch (blockIdx.z = job): the nodes of one level of the recursive
// triangular inversion.  Per job C = alpha A^T B with the operands at job-specific offsets of common base
// pointers; optionally the transpose of C is written as well (through LDS, coalesced both ways) -- the second
// product of a node yields W21 and U^-1's block Wt12 = W21^T at once, so no transpose pass is needed.
//   mode 2: B lower-triangular (k starts at n0);  mode 3: A upper-triangular (k ends at m0 + tile).
// (first version: one launch per product and node -- 2 x 39 GEMM + 39 transpose launches at N = 5000, most of them
//  a handful of workgroups wide and serialised on one stream.)
// ------------------------------------------------------------------------------------------------
template <class TL>
__global__ __launch_bounds__(256, TL::WPS) void sr_gemm_tn_jobs_kernel(
    const double* __restrict__ Ab, const double* __restrict__ Bb, double* Cb, double* CTb, long ld,
    const sr_gemm_job* __restrict__ jobs, double alpha, int mode, int njobs, int stf, int sts, sr_batch bt) {
    __shared__ double smem[TL::SMEM];
    const int bz = (int)blockIdx.z / njobs;               // batch member, job
    const sr_gemm_job jb = jobs[(int)blockIdx.z - bz * njobs];
    Ab += (long)bz * bt.sA; Bb += (long)bz * bt.sB; Cb += (long)bz * bt.sC;
    if (CTb) CTb += (long)bz * bt.sCT;
    const int tm = jb.M / TL::T;
    // heavy tiles first, so that the tail of the grid consists of the SHORT k ranges: the slow grid index (y) walks
    // the dimension that sets the k range -- mode 2: n ascending (k starts at n0), mode 3: m descending (k ends at m0 + T)
    int mt, nt;
    if (stf == 0) {
        mt = (mode == 2) ? (int)blockIdx.x : tm - 1 - (int)blockIdx.y;
        nt = (mode == 2) ? (int)blockIdx.y : (int)blockIdx.x;
    } else {
        // XCD-aware super-tiles (grids many times the chip; linear grid.x, a multiple of 8 x 64), as in
        // sr_gemm_tn_upper_kernel: workgroup b runs on XCD b % 8, which takes the super-tiles s = x, x + 8, .. of 8 x 8 tiles,
        // 64 consecutive workgroups of its own sequence each -- the 64 its 32 CUs hold at a time share 8 operand rows of A
        // and 8 of B and start at (nearly) the same k.  Super-tiles in the order of the plain grid: `stf` of them along the
        // fast dimension, `sts` along the one that sets the k range (heavy first).
        const long b = blockIdx.x;
        const int xcd = (int)(b & 7);
        const long l = b >> 3;
        const long st = (l >> 6) * 8 + xcd;
        const int w = (int)(l & 63);
        const int ss = (int)(st / stf), sf = (int)(st - (long)ss * stf);
        if (ss >= sts) return;
        if (mode == 2) { mt = sf * 8 + (w & 7); nt = ss * 8 + (w >> 3); }
        else           { mt = tm - 1 - (ss * 8 + (w >> 3)); nt = sf * 8 + (w & 7); }
    }
    if (nt * TL::T >= jb.N || mt >= tm || mt < 0) return;
    const int m0 = mt * TL::T;
    const int n0 = nt * TL::T;
    const int k_beg = (mode == 2) ? n0 : 0;
    const int k_end = (mode == 3) ? min(jb.K, m0 + TL::T) : jb.K;

    typename TL::Acc acc;
    acc.zero();
    TL::mainloop(Ab + jb.a + m0, ld, Bb + jb.b + n0, ld, k_beg, k_end, smem, acc);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    double* C = Cb + jb.c;
#pragma unroll
    for (int mi = 0; mi < TL::NI; ++mi)
#pragma unroll
        for (int ni = 0; ni < TL::NI; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc.v[mi][ni][r] *= alpha;
                C[(long)(m0 + TL::row(wm, mi, lane, r)) * ld + n0 + TL::col(wn, ni, lane)] = acc.v[mi][ni][r];
            }
    if (CTb == nullptr) return;
    // CT[n][m] = C[m][n] through T[tile n][65]: 64 rows (m) at a time (one pass for the 64-tile, two for 128)
    double* CT = CTb + jb.ct;
    double* T = smem;
    constexpr int TLD = 65;
    constexpr int HALVES = TL::T / 64;           // wavefront rows per pass: all (64-tile) or one of two (128-tile)
#pragma unroll 1
    for (int h = 0; h < HALVES; ++h) {
        if (HALVES == 1 || wm == h) {
#pragma unroll
            for (int mi = 0; mi < TL::NI; ++mi)
#pragma unroll
                for (int ni = 0; ni < TL::NI; ++ni)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        T[TL::col(wn, ni, lane) * TLD + (TL::row(wm, mi, lane, r) & 63)] = acc.v[mi][ni][r];
        }
        __syncthreads();
        for (int nrow = wave; nrow < TL::T; nrow += 4)        // one wavefront = one 512 B row segment of CT
            CT[(long)(n0 + nrow) * ld + m0 + h * 64 + lane] = T[nrow * TLD + lane];
        __syncthreads();
    }
}

int sr_launch_gemm_tn_jobs(const double* Ab, const double* Bb, double* Cb, double* CTb, long ld,
                           const sr_gemm_job* jobs_dev, int njobs, int maxM, int maxN, long tiles128, double alpha,
                           int mode, hipStream_t s, const sr_batch* btp) {
    const sr_batch bt = btp ? *btp : sr_batch{};
    SR_CHECK(njobs > 0 && (long)njobs * bt.n <= 65535 && maxM % srt::BM == 0 && maxN % srt::BN == 0 && (mode == 2 || mode == 3),
             SR_EINVAL, "gemm_tn_jobs: njobs=%d maxM=%d maxN=%d mode=%d", njobs, maxM, maxN, mode);
    const int T = sr_use_tile64_jobs(tiles128 * bt.n) ? 64 : 128;
    dim3 grid = (mode == 2) ? dim3(maxM / T, maxN / T, njobs * bt.n) : dim3(maxN / T, maxM / T, njobs * bt.n);
    // super-tiles pay once a job's grid is MANY times the chip -- the root of a big inversion (N = 50000: 195 x 196 tiles,
    // the whole update 2.507 -> 2.469 s on one box, 2.478 -> 2.429 on another); the order of the plain grid otherwise: a
    // workgroup goes to XCD b % 8, an XCD takes whole super-tiles, and with few of them per XCD the shares differ -- from 8192
    // tiles on (the second level at N = 50000, 98 x 98, the root at N = 30000) 2.469 -> 2.482 s and 557 -> 561 ms, from 4096
    // N = 20000 (root 78 x 79) 173.3 -> 176.8 ms, from 1024 N = 10000 23.7 -> 26.3 ms (profiles/r06_jobs_supertiles.txt)
    static const long st_thr = sr_lab_env("SR_JOBS_ST_THR", 16384);      // (lab build: A/B)
    int stf = 0, sts = 0;
    if (T == 128 && (long)grid.x * grid.y >= st_thr) {
        stf = ((int)grid.x + 7) / 8; sts = ((int)grid.y + 7) / 8;
        const long nst = (long)stf * sts;
        const long blocks = ((nst + 7) / 8) * 8 * 64;
        SR_CHECK(blocks < 2147483647L, SR_EINVAL, "gemm_tn_jobs: grid too large");
        grid = dim3((unsigned)blocks, 1, njobs * bt.n);
    }
    if (T == 64)
        hipLaunchKernelGGL(sr_gemm_tn_jobs_kernel<sr_tile64>, grid, dim3(256), 0, s, Ab, Bb, Cb, CTb, ld, jobs_dev, alpha,
                           mode, njobs, stf, sts, bt);
    else
        hipLaunchKernelGGL(sr_gemm_tn_jobs_kernel<sr_tile128>, grid, dim3(256), 0, s, Ab, Bb, Cb, CTb, ld, jobs_dev,
                           alpha, mode, njobs, stf, sts, bt);
    SR_HIP(hipGetLastError());
    return SR_OK;
}

