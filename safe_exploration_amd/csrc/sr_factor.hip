// sr_factor.hip -- model-update path: Gram matrix, blocked fp64-MFMA Cholesky
// K = U^T U, explicit triangular inverse W = U^-T by blocked forward substitution, helpers.
//
// replaces (numerically) what GPy computes for SimpleGPModel.train / update_model:
//   /root/reference/safe_exploration/ssm_gpy/gaussian_process.py:238-275, 398-419
#include "sr_mfma_tile.h"

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
        srt::mainloop_tn_glds<16>(A, lda, B, ldb, k0, k1, smem, acc);
    }
    static __device__ __forceinline__ int row(int wm, int mi, int lane, int r) { return srt::acc_row(wm, mi, lane, r); }
    static __device__ __forceinline__ int col(int wn, int ni, int lane) { return srt::acc_col(wn, ni, lane); }
};
struct sr_tile64 {
    using Acc = srt64::Acc;
    static constexpr int T = 64, NI = 2, SMEM = srt64::SMEM_DOUBLES, WPS = 2;
    static __device__ __forceinline__ void mainloop(const double* A, long lda, const double* B, long ldb, int k0,
                                                    int k1, double* smem, Acc& acc) {
        srt64::mainloop_tn(A, lda, B, ldb, k0, k1, smem, acc);
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
static inline bool sr_use_tile64(long tiles128, int K = 0) {
    // ... unless K is long: then a grid that occupies the chip at least once is throughput-bound and the 64-tile's 8 flop
    // per operand byte is the limit (the in-panel updates of the N = 50000 factorisation -- 128 rows x 50000 columns, K up
    // to 2944: 21 TF on 64-tiles)
    if (K >= 768 && tiles128 >= 256) return false;
    return tiles128 < 1024;
}
// ... but a 64 x 64 tile moves 8 bytes of operands per 8 flop (K-independent): a grid of them that fills the chip is
// bound by L2 / fabric bandwidth (N = 5000, K = 256 bulk update of two outputs: 1.7 GB in 254 us = 6.8 TB/s, 21 TF per
// output).  THROUGHPUT-bound products (bulk trailing update, the big levels of the inversion) therefore take the
// 128-tile (16 flop per byte) as soon as there are enough of them to occupy the chip once.
static inline bool sr_use_tile64_bulk(long tiles128) {
    return tiles128 < 192;
}
static inline bool sr_use_tile64_jobs(long tiles128) {
    return tiles128 < 1024;
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
    const sr_gemm_job* __restrict__ jobs, double alpha, int mode, int njobs, sr_batch bt) {
    __shared__ double smem[TL::SMEM];
    const int bz = (int)blockIdx.z / njobs;               // batch member, job
    const sr_gemm_job jb = jobs[(int)blockIdx.z - bz * njobs];
    Ab += (long)bz * bt.sA; Bb += (long)bz * bt.sB; Cb += (long)bz * bt.sC;
    if (CTb) CTb += (long)bz * bt.sCT;
    const int tm = jb.M / TL::T;
    // heavy tiles first, so that the tail of the grid consists of the SHORT k ranges: the slow grid index (y) walks
    // the dimension that sets the k range -- mode 2: n ascending (k starts at n0), mode 3: m descending (k ends at m0 + T)
    const int mt = (mode == 2) ? (int)blockIdx.x : tm - 1 - (int)blockIdx.y;
    const int nt = (mode == 2) ? (int)blockIdx.y : (int)blockIdx.x;
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
    const dim3 grid = (mode == 2) ? dim3(maxM / T, maxN / T, njobs * bt.n) : dim3(maxN / T, maxM / T, njobs * bt.n);
    if (T == 64)
        hipLaunchKernelGGL(sr_gemm_tn_jobs_kernel<sr_tile64>, grid, dim3(256), 0, s, Ab, Bb, Cb, CTb, ld, jobs_dev, alpha,
                           mode, njobs, bt);
    else
        hipLaunchKernelGGL(sr_gemm_tn_jobs_kernel<sr_tile128>, grid, dim3(256), 0, s, Ab, Bb, Cb, CTb, ld, jobs_dev,
                           alpha, mode, njobs, bt);
    SR_HIP(hipGetLastError());
    return SR_OK;
}

// ------------------------------------------------------------------------------------------------
// Gram matrix K[i][j] = sf2 exp(-0.5 |(z_i - z_j)/l|^2) + noise (i==j); identity on the padding.
// (kernel spec: ssm_gpy/gp_models_utils_casadi.py:17-40; noise on the diagonal:
//  ssm_gpy/gaussian_process.py:252-253)
// ------------------------------------------------------------------------------------------------
// sf2_dev / noise_dev (device scalars) take precedence over the by-value arguments when not NULL: the model update
// then needs no host copy of the hyper-parameters before its first launch.  Thread = 4 rows x 1 column: the column's
// scaled coordinates are loaded once per 4 entries, and the scaling is a multiplication by 1 / l (computed once per
// thread; the division per entry and dimension was a third of the kernel's time).
#define SR_GRAM_ROWS 32
__global__ __launch_bounds__(256) void sr_gram_kernel(const double* __restrict__ Z,
                                                      const double* __restrict__ ls, double sf2,
                                                      double noise, const double* __restrict__ sf2_dev,
                                                      const double* __restrict__ noise_dev,
                                                      double* __restrict__ K, int N, int Np, int D, long strideK) {
    __shared__ double zs[SR_GRAM_ROWS][SR_MAX_D];
    const int i0 = blockIdx.y * SR_GRAM_ROWS;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if ((int)((blockIdx.x * 256 + 255) | 127) < i0) return;   // blocks left of the diagonal are never read (upper factorisation)
    const int b = blockIdx.z;                // batch member (output)
    ls += (long)b * D;
    K += (long)b * strideK;
    if (sf2_dev) sf2 = sf2_dev[b];
    if (noise_dev) noise = noise_dev[b];
    const int off = Np - N;                  // front padding
    double il[SR_MAX_D], zj[SR_MAX_D];
    for (int c = 0; c < D; ++c) {
        il[c] = 1.0 / ls[c];
        zj[c] = (j < Np && j >= off) ? Z[(long)(j - off) * D + c] * il[c] : 0.0;
    }
    // the rows of this tile, scaled, in LDS (broadcast reads)
    for (int e = threadIdx.x; e < SR_GRAM_ROWS * D; e += 256) {
        const int r = e / D, c = e % D, i = i0 + r;
        zs[r][c] = (i < Np && i >= off) ? Z[(long)(i - off) * D + c] * (1.0 / ls[c]) : 0.0;
    }
    __syncthreads();
    if (j >= Np || (j | 127) < i0) return;
    for (int u = 0; u < SR_GRAM_ROWS; ++u) {
        const int i = i0 + u;
        if (i >= Np) break;
        double v;
        if (i < off || j < off) {
            v = (i == j) ? 1.0 : 0.0;
        } else if (i == j) {
            v = sf2 + noise;
        } else {
            double r2 = 0.0;
            for (int c = 0; c < D; ++c) {
                const double t = zs[u][c] - zj[c];
                r2 = fma(t, t, r2);
            }
            v = sf2 * exp(-0.5 * r2);
        }
        K[(long)i * Np + j] = v;
    }
}

int sr_launch_gram(const double* Z, const double* ls, double sf2, double noise, const double* sf2_dev,
                   const double* noise_dev, double* K, int N, int Np, int D, hipStream_t s, int nbatch, long strideK) {
    dim3 grid((Np + 255) / 256, (Np + SR_GRAM_ROWS - 1) / SR_GRAM_ROWS, nbatch);
    hipLaunchKernelGGL(sr_gram_kernel, grid, dim3(256), 0, s, Z, ls, sf2, noise, sf2_dev, noise_dev, K, N, Np, D, strideK);
    SR_HIP(hipGetLastError());
    return SR_OK;
}

// general kernel family (see sr_common.h); diagonal = k(z_i, z_i) + noise
__device__ __forceinline__ double sr_kappa(int kind, double r2) {
    if (kind == 0) return exp(-0.5 * r2);
    const double r = sqrt(r2);
    return (1.0 + 2.23606797749978969641 * r + (5.0 / 3.0) * r2) * exp(-2.23606797749978969641 * r);
}

__global__ __launch_bounds__(256) void sr_gram_general_kernel(const double* __restrict__ Z,
                                                              const double* __restrict__ kp, double noise,
                                                              const double* __restrict__ noise_dev,
                                                              double* __restrict__ K, int N, int Np, int D,
                                                              long strideK) {
    const int i = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= Np) return;
    if ((j | 127) < i) return;               // blocks left of the diagonal are never read (upper factorisation)
    const int b = blockIdx.z;                // batch member (output)
    kp += (long)b * SR_KP(D);
    K += (long)b * strideK;
    if (noise_dev) noise = noise_dev[b];
    const int off = Np - N;
    double v;
    if (i < off || j < off) {
        v = (i == j) ? 1.0 : 0.0;
    } else {
        const double* zi = Z + (long)(i - off) * D;
        const double* zj = Z + (long)(j - off) * D;
        const int kind = (int)kp[0];
        const double var = kp[1], c0 = kp[2];
        const double *sv = kp + 3, *av = kp + 3 + D, *bv = kp + 3 + 2 * D;
        double r2 = 0.0, la = 0.0, lb = 0.0;
        for (int c = 0; c < D; ++c) {
            const double t = (zi[c] - zj[c]) * sv[c];
            r2 = fma(t, t, r2);
            la = fma(av[c] * zi[c], zj[c], la);
            lb = fma(bv[c] * zi[c], zj[c], lb);
        }
        v = (c0 + la) * var * sr_kappa(kind, (i == j) ? 0.0 : r2) + lb;
        if (i == j) v += noise;
    }
    K[(long)i * Np + j] = v;
}

int sr_launch_gram_general(const double* Z, const double* kp, double noise, const double* noise_dev, double* K, int N,
                           int Np, int D, hipStream_t s, int nbatch, long strideK) {
    dim3 grid((Np + 255) / 256, Np, nbatch);
    hipLaunchKernelGGL(sr_gram_general_kernel, grid, dim3(256), 0, s, Z, kp, noise, noise_dev, K, N, Np, D, strideK);
    SR_HIP(hipGetLastError());
    return SR_OK;
}

// ------------------------------------------------------------------------------------------------
// Diagonal block: A_kk = U_kk^T U_kk (upper Cholesky) and in-place inverse of U_kk, all in LDS.
// One workgroup of 16 wavefronts; 128 x 129 doubles of LDS + the eight inverted 16 x 16 diagonal sub-blocks.
// This kernel sits on the critical path of the blocked factorisation (one launch per 128 rows): its serial core
// is the chain of 128 pivots (sqrt -> reciprocal -> rank-1 update -> next pivot).  Structure, per 16-column panel p:
//   (1) the 16 x 16 diagonal sub-block is factored by ONE wavefront entirely in registers: lane c holds column c,
//       pivots and multipliers travel through v_readlane (compile-time lanes) -- no LDS round trip, no fence;
//   (2) the panel row is solved by forward substitution with the stored pivot reciprocals (thread = column, the
//       factor broadcast from LDS) while a spare wavefront inverts the sub-block (rows in lanes) on the side;
//   (3) the trailing sub-matrix takes its rank-16 update on the MFMA tile; the wavefront that owns the next
//       diagonal tile goes straight on to factor it (step (1) of panel p + 1) while the others finish the update.
// Two block-wide barriers per panel.  Inverse: [A B; 0 C]^-1 = [A^-1, -A^-1 B C^-1; 0, C^-1] at block sizes
// 16, 32, 64 with both products on the MFMA tile; the intermediate A^-1 B is parked in the lower-left block.
// The strict lower triangle of S is scratch throughout and masked on every read that means "U" or "U^-1".
// (first version: sub-block steps through LDS in lock step, 3 barriers per panel, division per element: 88 us.)
// ------------------------------------------------------------------------------------------------
#define SR_PD_LD 129
#define SR_PD_THREADS 1024
#define SR_PD_XLD 17

__device__ __forceinline__ double sr_readlane_f64(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

// sd = sqrt(d), inv = 1 / sqrt(d) for a positive, normal d: v_rsq_f64 seed, one coupled Goldschmidt step
// (g -> sqrt, h -> 1/(2 sqrt), both to ~2^-52) and one residual correction each.  The library sqrt() + 1.0 / x
// pair costs ~32 dependent instructions (range scaling, v_div_scale / v_div_fmas / v_div_fixup); this chain is
// 9, and it sits 128 times on the critical path of every diagonal block.  |sd^2 - d| <= 1 ulp(d), |inv sd - 1| <= 2^-52.
__device__ __forceinline__ void sr_sqrt_rsqrt(double d, double& sd, double& inv) {
    const double y = __builtin_amdgcn_rsq(d);
    double g = d * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    const double e = fma(-g, g, d);
    g = fma(e, h, g);
    const double r2 = fma(-h, g, 0.5);
    h = fma(h, r2, h);
    sd = g;
    inv = h + h;
}

// upper Cholesky of the 16 x 16 sub-block at (j0, j0) of S by one wavefront, in registers.
// invd[j0 + j] receives 1 / U[j][j]; *fail the 1-based index of the first non-positive pivot.
__device__ __forceinline__ void sr_factor16(double* S, int j0, double* invd, int* fail, int lane) {
    const int c = lane & 15;                       // lanes 16 .. 63 mirror lanes 0 .. 15 (same addresses: broadcast)
    double a[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = S[(j0 + r) * SR_PD_LD + j0 + c];
    double myinv = 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        double d = sr_readlane_f64(a[j], j);       // pivot: wavefront-uniform
        if (!(d > 0.0)) {                          // also catches NaN
            if (lane == 0 && *fail == 0) *fail = j0 + j + 1;
            d = 1.0;
        }
        double sd, inv;
        sr_sqrt_rsqrt(d, sd, inv);
        if (c == j) myinv = inv;
        a[j] = (c == j) ? sd : a[j] * inv;         // row j of U (meaningful on lanes c >= j)
#pragma unroll
        for (int r = j + 1; r < 16; ++r) {
            const double ujr = sr_readlane_f64(a[j], r);     // U[j][r]: wavefront-uniform
            a[r] = fma(-ujr, a[j], a[r]);
        }
    }
    if (lane < 16) {
        invd[j0 + c] = myinv;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (r <= c) S[(j0 + r) * SR_PD_LD + j0 + c] = a[r];
    }
}

// X = U^-1 of the factored 16 x 16 sub-block at (j0, j0) by one wavefront: lane i holds row i of X,
//   X[i][j] = -(sum_{k < j} X[i][k] U[k][j]) / U[j][j],  X[j][j] = 1 / U[j][j];   U broadcast from LDS.
__device__ __forceinline__ void sr_invert16(const double* S, int j0, const double* invd, double* X, int lane) {
    const int i = lane & 15;
    double x[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < j; ++k) s = fma(x[k], S[(j0 + k) * SR_PD_LD + j0 + j], s);
        const double ij = invd[j0 + j];
        x[j] = (i == j) ? ij : ((i < j) ? -s * ij : 0.0);
    }
    if (lane < 16) {
#pragma unroll
        for (int j = 0; j < 16; ++j) X[i * SR_PD_XLD + j] = x[j];
    }
}

__global__ __launch_bounds__(SR_PD_THREADS, 1) void sr_potrf_diag_v2_kernel(double* A, long lda,
                                                                         double* wt_diag, double* w_diag,
                                                                         long ldw, int kb, int* info, int skip) {
    // skip: ablation bits of sr_test_potrf_diag (0 in production): 1 pivots, 2 panel rows, 4 trailing update,
    // 8 sub-block inverses, 16 inverse combination, 32 global loads / stores
    __shared__ double S[SR_NB * SR_PD_LD];
    __shared__ double Xd[(SR_NB / 16) * 16 * SR_PD_XLD];
    __shared__ double invd[SR_NB];
    __shared__ int fail;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lk = lane >> 4, ln = lane & 15;
    const long k0 = (long)kb * SR_NB;
    __builtin_amdgcn_s_setprio(3);
    if (tid == 0) fail = 0;
    for (int idx = tid; idx < SR_NB * SR_NB; idx += SR_PD_THREADS) {
        const int r = idx >> 7, c = idx & 127;
        S[r * SR_PD_LD + c] = (c >= r) ? ((skip & 32) ? (r == c ? 2.0 : 0.01) : A[(k0 + r) * lda + k0 + c]) : 0.0;
    }
    if (tid < SR_NB) invd[tid] = 1.0;
    __syncthreads();

    // ---- blocked right-looking upper Cholesky ------------------------------------------------------
    if (wave == 0 && !(skip & 1)) sr_factor16(S, 0, invd, &fail, lane);
    __syncthreads();
    for (int p = 0; p < SR_NB / 16; ++p) {
        const int j0 = 16 * p;
        // panel row: U[j0 .. j0+15][c] = D^-T A[j0 .. j0+15][c] for the columns right of the panel
        if (tid < SR_NB && tid >= j0 + 16 && !(skip & 2)) {
            double v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                double x = S[(j0 + i) * SR_PD_LD + tid];
#pragma unroll
                for (int k = 0; k < 16; ++k)
                    if (k < i) x = fma(-S[(j0 + k) * SR_PD_LD + j0 + i], v[k], x);
                v[i] = x * invd[j0 + i];
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) S[(j0 + i) * SR_PD_LD + tid] = v[i];
        }
        if (wave == SR_PD_THREADS / 64 - 1 && !(skip & 8)) sr_invert16(S, j0, invd, Xd + p * 16 * SR_PD_XLD, lane);
        __syncthreads();
        // trailing rank-16 update on the MFMA tile: tiles (ti <= tj) of the blocks right of / below the panel.
        // Tile 0 is the next diagonal sub-block: its owner (wavefront 0) factors it right away.
        const int nbt = SR_NB / 16 - 1 - p;
        const int ntile = (skip & 4) ? 0 : nbt * (nbt + 1) / 2;
        for (int e = (wave == 0) ? 0 : wave; e < ntile; e += (wave == 0) ? ntile : SR_PD_THREADS / 64 - 1) {
            int ti = 0, rem = e;
            while (rem >= nbt - ti) { rem -= nbt - ti; ++ti; }
            const int r0 = 16 * (p + 1 + ti), c0 = 16 * (p + 1 + ti + rem);
            d4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const double af = S[(j0 + 4 * kk + lk) * SR_PD_LD + r0 + ln];
                const double bf = S[(j0 + 4 * kk + lk) * SR_PD_LD + c0 + ln];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af, bf, acc, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) S[(r0 + lk + 4 * q) * SR_PD_LD + c0 + ln] -= acc[q];
        }
        if (wave == 0 && nbt > 0 && !(skip & 1)) {
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");     // the tile update above, then its reads below
            sr_factor16(S, j0 + 16, invd, &fail, lane);
        }
        __syncthreads();
    }
    if (fail) {
        if (tid == 0 && *info == 0) *info = (int)k0 + fail;
        // keep downstream kernels finite: identity block
        for (int idx = tid; idx < SR_NB * SR_NB; idx += SR_PD_THREADS) {
            const int r = idx >> 7, c = idx & 127;
            const double v = (r == c) ? 1.0 : 0.0;
            A[(k0 + r) * lda + k0 + c] = v;
            wt_diag[(long)r * ldw + c] = v;
            w_diag[(long)r * ldw + c] = v;
        }
        return;
    }
    if (!(skip & 32))
        for (int idx = tid; idx < SR_NB * SR_NB; idx += SR_PD_THREADS) {
            const int r = idx >> 7, c = idx & 127;
            A[(k0 + r) * lda + k0 + c] = (c >= r) ? S[r * SR_PD_LD + c] : 0.0;
        }
    __syncthreads();                                     // S is overwritten from here on

    // ---- inverse of the upper-triangular block -----------------------------------------------------
    // (a) the eight inverted 16 x 16 diagonal sub-blocks (computed on the side above) take their places
    for (int idx = tid; idx < (SR_NB / 16) * 256; idx += SR_PD_THREADS) {
        const int b = idx >> 8, i = (idx >> 4) & 15, j = idx & 15;
        S[(16 * b + i) * SR_PD_LD + 16 * b + j] = Xd[(b * 16 + i) * SR_PD_XLD + j];
    }
    __syncthreads();
    // (b) combine at block sizes 16, 32, 64
    for (int n = 16; n < ((skip & 16) ? 0 : SR_NB); n *= 2) {
        const int tpb = n / 16;                          // 16-tiles per block edge
        const int items = (SR_NB / (2 * n)) * tpb * tpb;
        // T^T = (A^-1 B)^T into the lower-left block
        for (int e = wave; e < items; e += SR_PD_THREADS / 64) {
            const int pq = e / (tpb * tpb), t2 = e % (tpb * tpb);
            const int a0 = 2 * n * pq, m0 = 16 * (t2 / tpb), n0 = 16 * (t2 % tpb);
            d4_t acc = {0.0, 0.0, 0.0, 0.0};
            for (int kb4 = m0 / 4; kb4 < n / 4; ++kb4) {     // A^-1 is upper triangular: k >= m
                const int k = 4 * kb4 + lk, m = m0 + ln;
                const double af = (k >= m) ? S[(a0 + m) * SR_PD_LD + a0 + k] : 0.0;
                const double bf = S[(a0 + k) * SR_PD_LD + a0 + n + n0 + ln];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af, bf, acc, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) S[(a0 + n + n0 + ln) * SR_PD_LD + a0 + m0 + lk + 4 * q] = acc[q];
        }
        __syncthreads();
        // X12 = -T C^-1 into the upper-right block
        for (int e = wave; e < items; e += SR_PD_THREADS / 64) {
            const int pq = e / (tpb * tpb), t2 = e % (tpb * tpb);
            const int a0 = 2 * n * pq, m0 = 16 * (t2 / tpb), c0 = 16 * (t2 % tpb);
            d4_t acc = {0.0, 0.0, 0.0, 0.0};
            for (int kb4 = 0; kb4 < (c0 + 16) / 4; ++kb4) {  // C^-1 is upper triangular: k <= c
                const int k = 4 * kb4 + lk, c = c0 + ln;
                const double af = S[(a0 + n + k) * SR_PD_LD + a0 + m0 + ln];
                const double bf = (k <= c) ? S[(a0 + n + k) * SR_PD_LD + a0 + n + c] : 0.0;
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af, bf, acc, 0, 0, 0);
            }
            // every wavefront must have read T and C^-1 of this block pair before X12 lands on top of B:
            // B is only read by the first product (barrier above), so the write is safe here
#pragma unroll
            for (int q = 0; q < 4; ++q) S[(a0 + m0 + lk + 4 * q) * SR_PD_LD + a0 + n + c0 + ln] = -acc[q];
        }
        __syncthreads();
    }
    if (skip & 32) {
        if (tid == 0) wt_diag[0] = S[5 * SR_PD_LD + 7];
        return;
    }
    for (int idx = tid; idx < SR_NB * SR_NB; idx += SR_PD_THREADS) {
        const int r = idx >> 7, c = idx & 127;
        wt_diag[(long)r * ldw + c] = (c >= r) ? S[r * SR_PD_LD + c] : 0.0;    // U_kk^-1   (upper)
        w_diag[(long)r * ldw + c] = (r >= c) ? S[c * SR_PD_LD + r] : 0.0;     // U_kk^-T   (lower)
    }
}

// The Schur complement of a row append of m <= 16 points lives in the LAST 16 x 16 pivot of an otherwise identity
// 128 x 128 block: one wavefront factors and inverts that corner (3 us; the full kernel takes 62 us on the way of every
// appended point).  Writes rows / columns 112 .. 127 of wt_diag (= U22^-1, upper) only.
// G != NULL: the block factored is A - G on rows / columns >= pf (the Schur complement C - U12^T U12 of the append, G a
// 128 x 128 block with leading dimension 128); blockIdx.x: batch member (strides sA of A, sW of wt_diag, 128 x 128 of G).
__global__ __launch_bounds__(64) void sr_potrf_corner16_kernel(double* A, long lda, double* wt_diag, long ldw,
                                                               int* info, const double* __restrict__ G, int pf,
                                                               long sA, long sW) {
    __shared__ double S[16 * SR_PD_LD];
    __shared__ double X[16 * SR_PD_XLD];
    __shared__ double invd[16];
    __shared__ int fail;
    A += (long)blockIdx.x * sA; wt_diag += (long)blockIdx.x * sW; info += blockIdx.x;
    if (G) G += (long)blockIdx.x * SR_NB * SR_NB;
    const int lane = threadIdx.x, c0 = SR_NB - 16;
    if (lane == 0) fail = 0;
    for (int idx = lane; idx < 256; idx += 64) {
        const int r = idx >> 4, c = idx & 15;
        double v = (c >= r) ? A[(long)(c0 + r) * lda + c0 + c] : 0.0;
        if (G && c >= r && c0 + r >= pf) v -= G[(c0 + r) * SR_NB + c0 + c];
        S[r * SR_PD_LD + c] = v;
    }
    __syncthreads();
    sr_factor16(S, 0, invd, &fail, lane);
    __syncthreads();
    sr_invert16(S, 0, invd, X, lane);
    __syncthreads();
    const bool bad = fail != 0;
    if (bad && lane == 0 && *info == 0) *info = c0 + fail;
    for (int idx = lane; idx < 256; idx += 64) {
        const int r = idx >> 4, c = idx & 15;
        const double eye = (r == c) ? 1.0 : 0.0;
        A[(long)(c0 + r) * lda + c0 + c] = bad ? eye : ((c >= r) ? S[r * SR_PD_LD + c] : 0.0);
        wt_diag[(long)(c0 + r) * ldw + c0 + c] = bad ? eye : X[r * SR_PD_XLD + c];
    }
}

int sr_launch_potrf_corner16(double* A, long lda, double* wt_diag, long ldw, int* info_dev, hipStream_t s,
                             const double* G, int pf, int nbatch, long sA, long sW) {
    hipLaunchKernelGGL(sr_potrf_corner16_kernel, dim3(nbatch), dim3(64), 0, s, A, lda, wt_diag, ldw, info_dev, G, pf, sA, sW);
    SR_HIP(hipGetLastError());
    return SR_OK;
}

// ------------------------------------------------------------------------------------------------
// Diagonal block, round 3: factor AND inverse in ONE sweep over the 8 panels of 16 columns.
// The elimination that turns A_kk into U_kk (U^-T A = U) is applied to an augmented identity at the same time, so that
// V = U_kk^-T is complete the moment the factor is -- the separate inversion phase of the kernel above (sub-block
// inverses + three combination levels, 5 more barriers) is gone, and so are the VALU forward substitutions:
//   * sr_factor16_aug: the 16 x 16 pivot block in registers as before; lanes 16..31, which only mirrored lanes 0..15,
//     now carry the columns of the identity through the same row operations: T_p = U_pp^-T at no extra instruction;
//   * phase 1 of panel p: the panel row  U[p][c] = T_p A[p][c] (c > p)  and  V[p][c'] = T_p V[p][c'] (c' < p): seven
//     16 x 16 tiles, 4 MFMAs each, one wavefront per tile, in place in LDS and straight to global memory;
//   * phase 2: trailing updates  A[r][c] -= U[p][r]^T U[p][c]  (p < r <= c)  and  V[r][c'] -= U[p][r]^T V[p][c']
//     (c' <= p < r) as independent tile jobs; wavefront 0 updates the next pivot tile from the panel tile it still
//     holds in registers (the MFMA result layout IS the operand layout of the next product: row (lane>>4) + 4 reg,
//     column lane & 15) and factors it at once.
//   V lives in the strict lower block triangle of S (A needs the upper one only): no second LDS matrix.
// fp64 MFMA and fp64 VALU share the DP pipe of a SIMD: the wavefronts that sit on wavefront 0's SIMD (4, 8, 12) take no
// MFMA job in phase 2 -- they copy the pivot stage to global memory and write the structural zeros of the outputs --,
// else the pivot chain (the critical path: 16 dependent rsqrt / FMA sequences per panel) stalls behind their MFMAs
// (first version of this kernel, every wavefront computing the panel tiles it needs itself: 43 us).
// Two barriers per panel.
// ------------------------------------------------------------------------------------------------
#define SR_PD_TLD 33      // pivot stage: 16 rows of [U_pp (16 columns) | T_p = U_pp^-T (16 columns)], padded

// (A square-root-free variant that takes R_j[r] from lane r with the DPP row broadcast of the fp64 ALU -- one
//  v_fmac_f64_dpp ... row_newbcast:r per updated entry, reciprocal instead of rsqrt on the chain, the 16 square roots
//  at the end -- was built and measured: 38 us per block against 30 with the v_readlane form below, software-pipelined
//  or not.  DPP on the fp64 ALU is slow on this part.)
// upper Cholesky of the 16 x 16 tile at (j0, j0) of S by one wavefront in registers.  X (LDS, 16 x SR_PD_TLD) receives
// [U | T], T = U^-T (lower triangular, exact zeros above; the part of U below its diagonal is scratch).
// *fail: 1-based index of the first non-positive pivot.  Nothing but the 16 LDS writes follows the pivot chain: the
// copies to global memory are another wavefront's job.
__device__ __forceinline__ void sr_factor16_aug(const double* S, int j0, double* X, int* fail, int lane) {
    const int c = lane & 15;
    const bool aug = (lane & 16) != 0;              // lanes 16..31 (and their mirrors 48..63): columns of the identity
    double a[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = aug ? ((r == c) ? 1.0 : 0.0) : S[(j0 + r) * SR_PD_LD + j0 + c];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        double d = sr_readlane_f64(a[j], j);       // pivot: wavefront-uniform
        if (!(d > 0.0)) {                          // also catches NaN
            if (lane == 0 && *fail == 0) *fail = j0 + j + 1;
            d = 1.0;
        }
        double sd, inv;
        sr_sqrt_rsqrt(d, sd, inv);
        a[j] = (!aug && c == j) ? sd : a[j] * inv; // row j of [U | U^-T]
#pragma unroll
        for (int r = j + 1; r < 16; ++r) {
            const double ujr = sr_readlane_f64(a[j], r);     // U[j][r]: wavefront-uniform
            a[r] = fma(-ujr, a[j], a[r]);
        }
    }
    if (lane < 32) {
#pragma unroll
        for (int r = 0; r < 16; ++r) X[r * SR_PD_TLD + lane] = a[r];
    }
}

// P = T X for the 16 x 16 tile X at rows j0.., columns c0.. of S: result in the MFMA layout (reg q: row (lane>>4) + 4 q,
// column lane & 15), which is also the A- and B-operand layout of a product that contracts over P's rows
__device__ __forceinline__ d4_t sr_pd_panel_tile(const double* S, const double* X, int j0, int c0, int lk, int ln) {
    d4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const double af = X[ln * SR_PD_TLD + 16 + 4 * kk + lk];
        const double bf = S[(j0 + 4 * kk + lk) * SR_PD_LD + c0 + ln];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af, bf, acc, 0, 0, 0);
    }
    return acc;
}

__global__ __launch_bounds__(SR_PD_THREADS, 1) void sr_potrf_diag_kernel(double* A, long lda,
                                                                         double* wt_diag, double* w_diag,
                                                                         long ldw, int kb, int* info, int skip,
                                                                         sr_batch bt) {
    // skip (sr_test_potrf_diag; 0 in production): 64 = leave A untouched (back-to-back timing on one input)
    __shared__ double S[SR_NB * SR_PD_LD];
    __shared__ double Xb[2][16 * SR_PD_TLD];
    __shared__ int fail;
    A += (long)blockIdx.x * bt.sA; wt_diag += (long)blockIdx.x * bt.sB; w_diag += (long)blockIdx.x * bt.sC;
    info += blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lk = lane >> 4, ln = lane & 15;
    const long k0 = (long)kb * SR_NB;
    const bool storeA = !(skip & 64);
    double* Ag = A + k0 * lda + k0;
    __builtin_amdgcn_s_setprio(3);
    if (tid == 0) fail = 0;
    for (int idx = tid; idx < SR_NB * SR_NB; idx += SR_PD_THREADS) {
        const int r = idx >> 7, c = idx & 127;
        S[r * SR_PD_LD + c] = (c >= r) ? Ag[(long)r * lda + c] : 0.0;       // strict lower triangle: V = 0 off its diagonal
    }
    __syncthreads();
    if (wave == 0) sr_factor16_aug(S, 0, Xb[0], &fail, lane);
    __syncthreads();
    // wavefront -> role in phase 2: 0 pivots; 4, 8, 12 (same SIMD as 0) stores only; the other 12 the MFMA jobs
    const bool simd0 = (wave & 3) == 0;
    const int worker = wave - 1 - (wave >> 2);             // 0 .. 11 for the MFMA wavefronts
    constexpr int NWORK = 12;
    for (int p = 0; p < SR_NB / 16; ++p) {
        const int j0 = 16 * p;
        const double* X = Xb[p & 1];
        const int nbt = SR_NB / 16 - 1 - p;
        // ---- phase 1: the seven tiles of panel row p, in place: wavefront w takes tile w (0: the next pivot's column)
        d4_t mine = {0.0, 0.0, 0.0, 0.0};
        if (wave < SR_NB / 16 - 1) {
            const int ct = (wave < nbt) ? p + 1 + wave : wave - nbt;         // tile column: A part c > p, then V part c' < p
            const int c0 = 16 * ct;
            mine = sr_pd_panel_tile(S, X, j0, c0, lk, ln);
            if (wave != 0 || nbt == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) S[(j0 + lk + 4 * q) * SR_PD_LD + c0 + ln] = mine[q];
            }
            if (ct > p) {
                if (storeA && wave != 0) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) Ag[(long)(j0 + lk + 4 * q) * lda + c0 + ln] = mine[q];
                }
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    w_diag[(long)(j0 + lk + 4 * q) * ldw + c0 + ln] = mine[q];
                    wt_diag[(long)(c0 + ln) * ldw + j0 + lk + 4 * q] = mine[q];
                }
            }
        }
        if (wave == 0 && nbt > 0) {
            // the panel tile of the next pivot's column is needed by the others too (row p + 1 of the trailing update)
#pragma unroll
            for (int q = 0; q < 4; ++q) S[(j0 + lk + 4 * q) * SR_PD_LD + j0 + 16 + ln] = mine[q];
        }
        __syncthreads();
        // ---- phase 2
        if (wave == 0) {
            if (nbt > 0) {
                d4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(mine[kk], mine[kk], acc, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) S[(j0 + 16 + lk + 4 * q) * SR_PD_LD + j0 + 16 + ln] -= acc[q];
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");     // the tile update above, then its reads below
                sr_factor16_aug(S, j0 + 16, Xb[(p + 1) & 1], &fail, lane);
            }
        } else if (!simd0) {
            const int nA = nbt * (nbt + 1) / 2;        // trailing tiles of A; tile 0 is wavefront 0's
            const int nV = nbt * (p + 1);              // trailing tiles of V
            for (int e = 1 + worker; e < nA + nV; e += NWORK) {
                int r0, c0;
                bool useT = false;
                if (e < nA) {
                    int ti = 0, rem = e;
                    while (rem >= nbt - ti) { rem -= nbt - ti; ++ti; }
                    r0 = 16 * (p + 1 + ti);
                    c0 = r0 + 16 * rem;
                } else {
                    const int v = e - nA, ti = v / (p + 1), cq = v % (p + 1);
                    r0 = 16 * (p + 1 + ti);
                    c0 = 16 * cq;
                    useT = (cq == p);                   // V[p][p] after the panel step is T itself (kept in the pivot stage)
                }
                d4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const double af = S[(j0 + 4 * kk + lk) * SR_PD_LD + r0 + ln];
                    const double bf = useT ? X[(4 * kk + lk) * SR_PD_TLD + 16 + ln] : S[(j0 + 4 * kk + lk) * SR_PD_LD + c0 + ln];
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af, bf, acc, 0, 0, 0);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) S[(r0 + lk + 4 * q) * SR_PD_LD + c0 + ln] -= acc[q];
            }
        } else {
            const int sw = (wave >> 2) - 1;            // 0, 1, 2
            if (sw == 0) {
                // the pivot stage of this panel: U_pp (zeros below its diagonal), T_p (lower) and T_p^T (upper)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r = lk + 4 * q;
                    const double u = X[r * SR_PD_TLD + ln], t = X[r * SR_PD_TLD + 16 + ln];
                    if (storeA) Ag[(long)(j0 + r) * lda + j0 + ln] = (r <= ln) ? u : 0.0;
                    w_diag[(long)(j0 + r) * ldw + j0 + ln] = t;
                    wt_diag[(long)(j0 + ln) * ldw + j0 + r] = t;
                }
                if (nbt > 0 && storeA) {                // wavefront 0 left the store of U[p][p+1] to us
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        Ag[(long)(j0 + lk + 4 * q) * lda + j0 + 16 + ln] = S[(j0 + lk + 4 * q) * SR_PD_LD + j0 + 16 + ln];
                }
            }
            // structural zeros: 28 pairs (strictly-lower tile (zr, zc) of A and U^-1, its mirror of U^-T), 4 per panel,
            // wavefronts 8 and 12 two each (wavefront 4 has the pivot stage)
            if (sw > 0 && p < 7) {
                for (int i = 0; i < 2; ++i) {
                    int z = 4 * p + 2 * (sw - 1) + i, zr = 1;
                    while (z >= zr) { z -= zr; ++zr; }
                    const int zc = z;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const long rr = 16 * zr + lk + 4 * q, cc = 16 * zc + ln;
                        if (storeA) Ag[rr * lda + cc] = 0.0;
                        wt_diag[rr * ldw + cc] = 0.0;
                        w_diag[(long)(16 * zc + lk + 4 * q) * ldw + 16 * zr + ln] = 0.0;
                    }
                }
            }
        }
        __syncthreads();
    }
    if (fail) {
        if (tid == 0 && *info == 0) *info = (int)k0 + fail;
        // keep downstream kernels finite: identity block
        for (int idx = tid; idx < SR_NB * SR_NB; idx += SR_PD_THREADS) {
            const int r = idx >> 7, c = idx & 127;
            const double v = (r == c) ? 1.0 : 0.0;
            if (storeA) Ag[(long)r * lda + c] = v;
            wt_diag[(long)r * ldw + c] = v;
            w_diag[(long)r * ldw + c] = v;
        }
    }
}

int sr_launch_potrf_diag(double* A, long lda, double* wt_diag, double* w_diag, long ldw, int kb,
                         int* info_dev, hipStream_t s, int skip, const sr_batch* btp) {
    const sr_batch bt = btp ? *btp : sr_batch{};
    if (skip & 128)      // the round-2 kernel (A/B timing through sr_test_potrf_diag only)
        hipLaunchKernelGGL(sr_potrf_diag_v2_kernel, dim3(1), dim3(SR_PD_THREADS), 0, s, A, lda, wt_diag, w_diag, ldw,
                           kb, info_dev, skip & 63);
    else
        hipLaunchKernelGGL(sr_potrf_diag_kernel, dim3(bt.n), dim3(SR_PD_THREADS), 0, s, A, lda, wt_diag, w_diag, ldw,
                           kb, info_dev, skip, bt);
    SR_HIP(hipGetLastError());
    return SR_OK;
}

// ------------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------------
// dst[c][r] = src[r][c] for an (rows x cols) block; rows, cols multiples of 32
__global__ __launch_bounds__(256) void sr_transpose_kernel(const double* __restrict__ src, long lds_,
                                                           double* __restrict__ dst, long ldd) {
    __shared__ double t[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int r = ty; r < 32; r += 8) t[r][tx] = src[(long)(by + r) * lds_ + bx + tx];
    __syncthreads();
    for (int r = ty; r < 32; r += 8) dst[(long)(bx + r) * ldd + by + tx] = t[tx][r];
}

int sr_launch_transpose_rect(const double* src, long lds_, double* dst, long ldd, int rows, int cols,
                             hipStream_t s) {
    SR_CHECK(rows % 32 == 0 && cols % 32 == 0 && rows > 0 && cols > 0, SR_EINVAL,
             "transpose: %d x %d not multiples of 32", rows, cols);
    hipLaunchKernelGGL(sr_transpose_kernel, dim3(cols / 32, rows / 32), dim3(256), 0, s, src, lds_, dst, ldd);
    SR_HIP(hipGetLastError());
    return SR_OK;
}

int sr_launch_transpose(const double* src, double* dst, int n, hipStream_t s) {
    return sr_launch_transpose_rect(src, n, dst, n, n, n, s);
}

// ---- helpers of the block row-append update (sr_gp_append) --------------------------------------
// S[r][c] -= G[r][c] on the real block r, c >= pf of a front-padded 128 x 128 tile
__global__ __launch_bounds__(256) void sr_sub_block_kernel(double* __restrict__ S, const double* __restrict__ G,
                                                           int pf) {
    for (int idx = threadIdx.x; idx < SR_NB * SR_NB; idx += 256) {
        const int r = idx >> 7, c = idx & 127;
        if (r >= pf && c >= pf) S[idx] -= G[idx];
    }
}

int sr_launch_sub_block(double* S, const double* G, int pf, hipStream_t s) {
    hipLaunchKernelGGL(sr_sub_block_kernel, dim3(1), dim3(256), 0, s, S, G, pf);
    SR_HIP(hipGetLastError());
    return SR_OK;
}

// new U^-1 (Np1 x Np1, front padding off1) from the old one, the new off-diagonal columns
// Y2 (= -U^-1 U12 U22^-1, rows in OLD padded indexing, 128 front-padded columns) and U22^-1 (invS)
__global__ __launch_bounds__(256) void sr_append_assemble_kernel(const double* __restrict__ Wt0, int Np0,
                                                                 int off0, int N0, const double* __restrict__ Y2,
                                                                 const double* __restrict__ invS, int m,
                                                                 double* __restrict__ Wt1, int Np1, int off1) {
    const int r = blockIdx.y;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= Np1) return;
    const int pf = SR_NB - m;
    double v;
    if (r < off1 || c < off1) {
        v = (r == c) ? 1.0 : 0.0;
    } else {
        const int i = r - off1, j = c - off1;
        if (i < N0 && j < N0) v = Wt0[(long)(off0 + i) * Np0 + off0 + j];
        else if (i < N0) v = Y2[(long)(off0 + i) * SR_NB + pf + (j - N0)];
        else if (j >= N0) v = invS[(pf + i - N0) * SR_NB + pf + (j - N0)];
        else v = 0.0;
    }
    Wt1[(long)r * Np1 + c] = v;
}

int sr_launch_append_assemble(const double* Wt0, int Np0, int off0, int N0, const double* Y2,
                              const double* invS, int m, double* Wt1, int Np1, int off1, hipStream_t s) {
    hipLaunchKernelGGL(sr_append_assemble_kernel, dim3((Np1 + 255) / 256, Np1), dim3(256), 0, s, Wt0, Np0, off0,
                       N0, Y2, invS, m, Wt1, Np1, off1);
    SR_HIP(hipGetLastError());
    return SR_OK;
}

// ---- row append with FEW new points (m <= 16): matrix-vector shaped kernels, every pass over U^-1 is one
// coalesced stream (the GEMM route pads the m columns to a 128-wide tile and serialises 40 workgroups).
// U12t[a][i] (a < m, i < Np0, padded row indexing) holds U12 = U^-T K(Z_old, Z_new) column by column.

// G[pf+a][pf+b] = sum_i U12t[a][i] U12t[b][i] inside a zeroed 128 x 128 block; grid (m, m)
__global__ __launch_bounds__(256) void sr_append_gsmall_kernel(const double* __restrict__ U12t, int Np0, int m,
                                                               double* __restrict__ G) {
    __shared__ double red[4];
    const int a = blockIdx.x, b = blockIdx.y, pf = SR_NB - m;
    U12t += (long)blockIdx.z * m * Np0; G += (long)blockIdx.z * SR_NB * SR_NB;      // batch member (output)
    double v = 0.0;
    for (int i = threadIdx.x; i < Np0; i += 256) v = fma(U12t[(long)a * Np0 + i], U12t[(long)b * Np0 + i], v);
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) G[(pf + a) * SR_NB + pf + b] = red[0] + red[1] + red[2] + red[3];
}

// Xt[c][i] = sum_{a <= c} U12t[a][i] invS[pf+a][pf+c]   (X = U12 U22^-1, U22^-1 upper triangular)
__global__ __launch_bounds__(256) void sr_append_xt_kernel(const double* __restrict__ U12t,
                                                           const double* __restrict__ invS, int Np0, int m,
                                                           double* __restrict__ Xt, long sXt) {
    const int i = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y, pf = SR_NB - m;
    U12t += (long)blockIdx.z * m * Np0; invS += (long)blockIdx.z * SR_NB * SR_NB; Xt += (long)blockIdx.z * sXt;
    if (i >= Np0) return;
    double v = 0.0;
    for (int a = 0; a <= c; ++a) v = fma(U12t[(long)a * Np0 + i], invS[(pf + a) * SR_NB + pf + c], v);
    Xt[(long)c * Np0 + i] = v;
}

// Y2[i][pf+c] = -sum_{k >= i} Wt0[i][k] Xt[c][k]  (Y2 = -U^-1 X): one wavefront per row, lanes over k, MC columns
template <int MC>
__global__ __launch_bounds__(256) void sr_append_y2_kernel(const double* __restrict__ Wt0, int Np0,
                                                           const double* __restrict__ Xt, int m,
                                                           double* __restrict__ Y2) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63, pf = SR_NB - m;
    if (row >= Np0) return;
    double acc[MC];
#pragma unroll
    for (int c = 0; c < MC; ++c) acc[c] = 0.0;
    for (int k = row + lane; k < Np0; k += 64) {
        const double w = Wt0[(long)row * Np0 + k];
#pragma unroll
        for (int c = 0; c < MC; ++c)
            if (c < m) acc[c] = fma(w, Xt[(long)c * Np0 + k], acc[c]);
    }
#pragma unroll
    for (int c = 0; c < MC; ++c) {
        double v = acc[c];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0 && c < m) Y2[(long)row * SR_NB + pf + c] = -v;
    }
}

// The new columns and the move of the old factor in ONE pass: wavefront per OLD padded row `row` (real index i):
//   acc[c]      = sum_{k >= row} Wt0[row][k] Xt[c][k]                       (Y2 = -U^-1 X, as sr_append_y2_kernel)
//   Wt1[r][..]  = the same row, shifted to the new padding (r = off1 + i), upper part only, + the m new entries -acc
// and one extra workgroup writes the rows of the new points (U22^-1).  Everything below the diagonal of Wt1 and its
// identity padding must already be in place (a buffer that held an earlier state of the same model, or zeroed +
// sr_launch_eye_front): the copy through sr_append_assemble_kernel read and wrote the full square, zeros included --
// 525 MB per output and append at N = 5000 against 210 MB here.
template <int MC>
__global__ __launch_bounds__(256) void sr_append_move_kernel(const double* __restrict__ Wt0, int Np0, int off0, int N0,
                                                             const double* __restrict__ Xt,
                                                             const double* __restrict__ invS, int m,
                                                             double* __restrict__ Y2, double* __restrict__ Wt1,
                                                             int Np1, int off1, long sXt, long sY2,
                                                             const double* __restrict__ U12one) {
    // U12one != NULL (MC == 1, one new point): Xt[0][k] = U12t[0][k] U22^-1 is formed here, Xt is not read
    const int lane = threadIdx.x & 63, pf = SR_NB - m;
    {                                                       // batch member (output)
        const long b = blockIdx.y;
        Wt0 += b * Np0 * Np0; Xt += b * sXt; invS += b * SR_NB * SR_NB; Y2 += b * sY2; Wt1 += b * Np1 * Np1;
        if (U12one) U12one += b * Np0;
    }
    const double inv00 = (MC == 1 && U12one) ? invS[pf * SR_NB + pf] : 0.0;
    const int nrow_blocks = (Np0 + 3) / 4;
    if ((int)blockIdx.x >= nrow_blocks) {
        // rows of the new points: Wt1[off1 + N0 + q][off1 + N0 + c] = U22^-1[q][c]
        for (int e = threadIdx.x; e < m * m; e += 256) {
            const int q = e / m, c = e % m;
            if (c >= q) Wt1[(long)(off1 + N0 + q) * Np1 + off1 + N0 + c] = invS[(pf + q) * SR_NB + pf + c];
        }
        return;
    }
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= Np0 || row < off0) return;                   // old padding rows carry nothing
    const int shift = off1 - off0;                          // new column index = old + shift
    double* dst = Wt1 + (long)(row + shift) * Np1 + shift;
    double acc[MC];
#pragma unroll
    for (int c = 0; c < MC; ++c) acc[c] = 0.0;
    for (int k = row + lane; k < Np0; k += 64) {
        const double w = Wt0[(long)row * Np0 + k];
        dst[k] = w;
        if (MC == 1 && U12one) {
            acc[0] = fma(w, U12one[k] * inv00, acc[0]);
        } else {
#pragma unroll
            for (int c = 0; c < MC; ++c)
                if (c < m) acc[c] = fma(w, Xt[(long)c * Np0 + k], acc[c]);
        }
    }
#pragma unroll
    for (int c = 0; c < MC; ++c) {
        double v = acc[c];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0 && c < m) {
            Y2[(long)row * SR_NB + pf + c] = -v;
            dst[Np0 + c] = -v;                               // column off1 + N0 + c of the new matrix
        }
    }
}

// ones on the first n diagonal entries (identity padding of a zeroed matrix)
__global__ __launch_bounds__(256) void sr_eye_front_kernel(double* __restrict__ W, int ld, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    W += (long)blockIdx.y * ld * ld;                        // batch member: ld x ld matrices back to back
    if (i < n) W[(long)i * ld + i] = 1.0;
}

int sr_launch_eye_front(double* W, int ld, int n, hipStream_t s, int nbatch) {
    if (n <= 0) return SR_OK;
    hipLaunchKernelGGL(sr_eye_front_kernel, dim3((n + 255) / 256, nbatch), dim3(256), 0, s, W, ld, n);
    SR_HIP(hipGetLastError());
    return SR_OK;
}

// nbatch outputs in one launch each: U12t (m x Np0 each), invS (128 x 128), Wt0 / Wt1 (full squares) back to back, Xt and Y2
// with the strides given
int sr_launch_append_move(const double* Wt0, int Np0, int off0, int N0, const double* U12t, const double* invS, int m,
                          double* Xt, double* Y2, double* Wt1, int Np1, int off1, hipStream_t s, int nbatch, long sXt,
                          long sY2) {
    const dim3 grid((Np0 + 3) / 4 + 1, nbatch);
    if (m <= 1) {                                           // one new point: X = U12 U22^-1 is a scaling, done in the move
        hipLaunchKernelGGL(sr_append_move_kernel<1>, grid, dim3(256), 0, s, Wt0, Np0, off0, N0, Xt, invS, m, Y2, Wt1, Np1, off1, sXt,
                           sY2, U12t);
        SR_HIP(hipGetLastError());
        return SR_OK;
    }
    hipLaunchKernelGGL(sr_append_xt_kernel, dim3((Np0 + 255) / 256, m, nbatch), dim3(256), 0, s, U12t, invS, Np0, m, Xt, sXt);
    SR_HIP(hipGetLastError());
    if (m <= 4)
        hipLaunchKernelGGL(sr_append_move_kernel<4>, grid, dim3(256), 0, s, Wt0, Np0, off0, N0, Xt, invS, m, Y2, Wt1, Np1, off1, sXt,
                           sY2, (const double*)nullptr);
    else
        hipLaunchKernelGGL(sr_append_move_kernel<16>, grid, dim3(256), 0, s, Wt0, Np0, off0, N0, Xt, invS, m, Y2, Wt1, Np1, off1, sXt,
                           sY2, (const double*)nullptr);
    SR_HIP(hipGetLastError());
    return SR_OK;
}

// ------------------------------------------------------------------------------------------------
// ONE new point on a SMALL model (ARD-RBF or the general kernel family; old padded size <= 512, new <= 640: the reference's own regime, a transition
// appended after every step of its exploration loop, exploration_runner.py:186-188) -- the whole append in ONE launch,
// one workgroup of 16 wavefronts per output:
//   b = K(Z_old, z_new), mu_old = b . alpha0, u12 = U^-T b (thread = column, 4 k-slices), s = sf2 + noise - |u12|^2,
//   u22^-1 = 1 / sqrt(s), X = u12 u22^-1;
//   then the new factor row by row (wavefront = row), written IN FULL (zeros below the diagonal, identity padding: the
//   target buffer needs no preparation): the old row moved to the new padding, its new last entry
//   y2 = -sum_{k >= row} U^-1[row][k] X[k] from the same pass, alpha1 = alpha0 + y2 v2 (v2 = u22^-1 (y_new - mu_old)),
//   the shifted targets, the new point's row (u22^-1), log det of the new factor (SR_APPEND1_WGS partial sums per
//   output), the failure word.
// The general route does this in 10 launches (65 us inside sr_gp_append at any size up to N ~ 1000); the arithmetic is
// the same (same sums in another order: the tests compare both routes with the refit and the CPU restatement).
// ------------------------------------------------------------------------------------------------
struct sr_append1_args {
    const double* Wt0; const double* alpha0; const double* yT0; const double* Z;     // old state (Z: N0 x D)
    const double* ls; const double* sf2; const double* noise;                         // n_out x D, n_out, n_out
    const double* kp;                                                                 // general kernels: n_out x SR_KP(D), else NULL
    const double* znew; const double* ynew;                                           // D, n_out
    double* Wt1; double* alpha1; double* yT1; double* Zdst;                           // new state (Zdst: row N0 of Z, or NULL)
    double* logdet; int* info;                                                        // n_out each
    int N0, Np0, Np1, D, n_out;
    // the new point in the kernel arguments (a host caller: no H2D copy command); znew / ynew are NULL then
    int inl; double xin[SR_MAX_D]; double yin[SR_APPEND1_MAX_OUT];
};

template <int NPMAX>   // 256 or 512: the old padded size it serves
__global__ __launch_bounds__(1024) void sr_append1_small_kernel(sr_append1_args a) {
    __shared__ double b[NPMAX], u12[NPMAX], X[NPMAX], part[4][NPMAX], red[16];
    __shared__ double s_mu, s_inv, s_v2;
    __shared__ double zn[SR_MAX_D];                          // the new input (from memory or from the kernel arguments)
    const int d = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N0 = a.N0, Np0 = a.Np0, Np1 = a.Np1, D = a.D;
    if (tid < D) zn[tid] = a.inl ? a.xin[tid] : a.znew[tid];
    const double y_new = a.inl ? a.yin[d] : a.ynew[d];
    __syncthreads();
    auto sr_wave_sum = [](double v) {
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        return v;
    };
    const int off0 = Np0 - N0, off1 = Np1 - (N0 + 1), shift = off1 - off0;
    const double* Wt0 = a.Wt0 + (long)d * Np0 * Np0;
    const double* alpha0 = a.alpha0 + (long)d * Np0;
    double* Wt1 = a.Wt1 + (long)d * Np1 * Np1;
    // gridDim.y workgroups per output share the rows of the new factor (each of them repeats the cheap first part: one
    // workgroup alone writes a 256-row factor in 25 us, four take 8); workgroup y = 0 also reports failure and copies z_new
    const int wy = blockIdx.y, nwy = gridDim.y;
    if (d == 0 && wy == 0 && a.Zdst && tid < D) a.Zdst[tid] = zn[tid];
    // ---- b = K(Z_old, z_new) in padded row indexing, mu_old = b . alpha0
    double mu_t = 0.0;
#pragma unroll 1
    for (int row = tid; row < NPMAX; row += 1024) {
        double v = 0.0;
        if (row < Np0 && row >= off0) {
            const double* z = a.Z + (long)(row - off0) * D;
            if (a.kp) {                                      // general family (sr_common.h), as sr_gram_general_kernel
                const double* kp = a.kp + (long)d * SR_KP(D);
                const double *sv = kp + 3, *av = kp + 3 + D, *bv = kp + 3 + 2 * D;
                double r2 = 0.0, la = 0.0, lb = 0.0;
                for (int c = 0; c < D; ++c) {
                    const double t = (z[c] - zn[c]) * sv[c];
                    r2 = fma(t, t, r2);
                    la = fma(av[c] * z[c], zn[c], la);
                    lb = fma(bv[c] * z[c], zn[c], lb);
                }
                v = (kp[2] + la) * kp[1] * sr_kappa((int)kp[0], r2) + lb;
            } else {
                double r2 = 0.0;
                for (int c = 0; c < D; ++c) {
                    const double t = (z[c] - zn[c]) / a.ls[d * D + c];
                    r2 = fma(t, t, r2);
                }
                v = a.sf2[d] * exp(-0.5 * r2);
            }
            mu_t += v * alpha0[row];
        }
        b[row] = v;
    }
    auto block_sum = [&](double v) {                         // fixed order: wavefront sums, then wavefront 0 .. 15
        const double w = sr_wave_sum(v);
        __syncthreads();                                     // (red may still be read from the previous sum)
        if (lane == 0) red[wave] = w;
        __syncthreads();
        double t = 0.0;
        for (int k = 0; k < 16; ++k) t += red[k];
        return t;
    };
    const double mu_old = block_sum(mu_t);
    if (tid == 0) s_mu = mu_old;
    // ---- u12[i] = sum_{k <= i} U^-1[k][i] b[k]: work item (slice q of NPMAX / 4 rows, column i), 16 loads in flight
    constexpr int SLICE = NPMAX / 4;
#pragma unroll 1
    for (int wi = tid; wi < 4 * NPMAX; wi += 1024) {
        const int q = wi / NPMAX, i = wi % NPMAX;
        double acc = 0.0;
        if (i < Np0) {
            constexpr int UB = (NPMAX == 256) ? 16 : 8;      // loads in flight (the 512 form would spill with 16)
            const int k_end = min(q * SLICE + SLICE - 1, i);
            for (int k0 = q * SLICE; k0 <= k_end; k0 += UB) {
                double w[UB];
#pragma unroll
                for (int u = 0; u < UB; ++u) w[u] = (k0 + u <= k_end) ? Wt0[(long)(k0 + u) * Np0 + i] : 0.0;
#pragma unroll
                for (int u = 0; u < UB; ++u) acc = fma(w[u], b[min(k0 + u, NPMAX - 1)], acc);
            }
        }
        part[q][i] = acc;
    }
    __syncthreads();
    double g_t = 0.0;
    for (int i = tid; i < NPMAX; i += 1024) {
        const double v = (part[0][i] + part[1][i]) + (part[2][i] + part[3][i]);
        u12[i] = v;
        g_t = fma(v, v, g_t);
    }
    const double g = block_sum(g_t);
    if (tid == 0) {
        double prior;                                        // k(z_new, z_new)
        if (a.kp) {
            const double* kp = a.kp + (long)d * SR_KP(D);
            double la = 0.0, lb = 0.0;
            for (int c = 0; c < D; ++c) {
                la = fma(kp[3 + D + c] * zn[c], zn[c], la);
                lb = fma(kp[3 + 2 * D + c] * zn[c], zn[c], lb);
            }
            prior = (kp[2] + la) * kp[1] + lb;               // kappa(0) = 1
        } else {
            prior = a.sf2[d];
        }
        double sch = prior + a.noise[d] - g;                 // Schur complement of the new point
        if (!(sch > 0.0)) {                                  // also catches NaN
            if (wy == 0) a.info[d] = N0 + 1;
            sch = 1.0;
        } else if (wy == 0) {
            a.info[d] = 0;
        }
        double sd, inv;
        sr_sqrt_rsqrt(sch, sd, inv);
        s_inv = inv;
        s_v2 = inv * (y_new - s_mu);
    }
    __syncthreads();
    const double inv = s_inv, v2 = s_v2;
    for (int i = tid; i < NPMAX; i += 1024) X[i] = u12[i] * inv;
    __syncthreads();
    // ---- the new factor, alpha and targets, row by row
    double ld = 0.0;                                         // sum of log(diagonal) over this wavefront's rows (lane 0)
    const double* yT0 = a.yT0 + (long)d * Np0;
    double* alpha1 = a.alpha1 + (long)d * Np1;
    double* yT1 = a.yT1 + (long)d * Np1;
    const int Rlast = Np1 - 1;                               // row of the new point
    for (int R = wy * 16 + wave; R < Np1; R += 16 * nwy) {
        double* dst = Wt1 + (long)R * Np1;
        if (R < off1 || R == Rlast) {
            const double dg = (R == Rlast) ? inv : 1.0;
            for (int C = lane; C < Np1; C += 64) dst[C] = (C == R) ? dg : 0.0;
            if (lane == 0) {
                alpha1[R] = (R == Rlast) ? inv * v2 : 0.0;
                yT1[R] = (R == Rlast) ? y_new : 0.0;
                if (R == Rlast) ld += log(inv);
            }
            continue;
        }
        const int r0 = R - shift;                            // old padded row
        const double* src = Wt0 + (long)r0 * Np0;
        double acc = 0.0, dgv = 1.0;
        for (int C = lane; C < Rlast; C += 64) {
            double v = 0.0;
            if (C >= R) {
                v = src[C - shift];
                acc = fma(v, X[C - shift], acc);
                if (C == R) dgv = v;
            }
            dst[C] = v;
        }
        acc = sr_wave_sum(acc);
        dgv = __shfl(dgv, R & 63);                           // the lane that held the diagonal entry
        if (lane == 0) {
            dst[Rlast] = -acc;
            alpha1[R] = fma(-acc, v2, alpha0[r0]);
            yT1[R] = yT0[r0];
            ld += log(dgv);
        }
    }
    __syncthreads();
    if (lane == 0) red[wave] = ld;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int w = 0; w < 16; ++w) t += red[w];
        a.logdet[d * nwy + wy] = -2.0 * t;                  // partial sums: the host adds the gridDim.y of an output
    }
}

int sr_launch_append1_small(const double* Wt0, const double* alpha0, const double* yT0, const double* Z, const double* ls,
                            const double* sf2, const double* noise, const double* kp, const double* znew, const double* ynew, double* Wt1,
                            double* alpha1, double* yT1, double* Zdst, double* logdet, int* info, int N0, int Np0, int Np1,
                            int D, int n_out, hipStream_t s, const double* x_host, const double* y_host) {
    SR_CHECK(Np0 <= 512 && Np1 <= 640 && N0 >= 1 && N0 <= Np0, SR_EINVAL, "append1_small: Np0 = %d, Np1 = %d", Np0, Np1);
    sr_append1_args a{Wt0, alpha0, yT0, Z, ls, sf2, noise, kp, znew, ynew, Wt1, alpha1, yT1, Zdst, logdet, info, N0, Np0, Np1, D, n_out,
                      0, {}, {}};
    if (x_host) {                                            // the new point travels in the kernel arguments
        SR_CHECK(y_host && D <= SR_MAX_D && n_out <= SR_APPEND1_MAX_OUT, SR_EINVAL, "append1_small: D = %d, n_out = %d", D, n_out);
        a.inl = 1; a.znew = nullptr; a.ynew = nullptr;
        for (int c = 0; c < D; ++c) a.xin[c] = x_host[c];
        for (int d = 0; d < n_out; ++d) a.yin[d] = y_host[d];
    }
    if (Np0 <= 256) hipLaunchKernelGGL(sr_append1_small_kernel<256>, dim3(n_out, SR_APPEND1_WGS), dim3(1024), 0, s, a);
    else hipLaunchKernelGGL(sr_append1_small_kernel<512>, dim3(n_out, SR_APPEND1_WGS), dim3(1024), 0, s, a);
    SR_HIP(hipGetLastError());
    return SR_OK;
}

// alpha of the grown model without another pass over U^-1:  with r = y_new - mu_old(z_new) (the old model's
// mean at the new points, which the K* pass has just produced) and v2 = U22^-T r,
//   alpha1 = [alpha0 + Y2 v2 ; U22^-1 v2].     One workgroup recomputes v2 (m <= 16), grid over the rows.
__global__ __launch_bounds__(256) void sr_append_alpha_kernel(const double* __restrict__ alpha0, int Np0, int N0,
                                                              const double* __restrict__ Y2,
                                                              const double* __restrict__ invS,
                                                              const double* __restrict__ mu_part, int nsplit,
                                                              int n_out, int d, long Tp,
                                                              const double* __restrict__ Ynew, int m,
                                                              double* __restrict__ alpha1, int Np1, int qoff, long sY2) {
    // qoff: position of the first new point among the queries of the K* pass (front-padded query block: 128 - m)
    __shared__ double r[SR_NB], v2[SR_NB], red[4][16];
    const int pf = SR_NB - m, off0 = Np0 - N0, off1 = Np1 - (N0 + m);
    {                                                       // batch: output d + blockIdx.y, the pointers given belong to d
        const long b = blockIdx.y;
        d += (int)b; alpha0 += b * Np0; Y2 += b * sY2; invS += b * SR_NB * SR_NB; alpha1 += b * Np1;
    }
    // mean of the old model at the new points: the N-split partials of the K* pass, summed by the whole workgroup
    // (one thread per point walking up to Np / 16 partials was 119 us of dependent-latency at N = 5000)
    for (int q0 = 0; q0 < m; q0 += 16) {
        double acc[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0.0;
        for (int sp = threadIdx.x; sp < nsplit; sp += 256) {
            const double* src = mu_part + ((long)sp * n_out + d) * Tp + qoff + q0;
#pragma unroll
            for (int q = 0; q < 16; ++q)
                if (q0 + q < m) acc[q] += src[q];
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            double v = acc[q];
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][q] = v;
        }
        __syncthreads();
        if (threadIdx.x < 16 && q0 + (int)threadIdx.x < m) {
            const int q = q0 + threadIdx.x;
            const double mu = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
            r[q] = Ynew[(long)q * n_out + d] - mu;
        }
        __syncthreads();
    }
    if ((int)threadIdx.x < m) {
        double v = 0.0;
        for (int b = 0; b <= (int)threadIdx.x; ++b) v = fma(invS[(pf + b) * SR_NB + pf + threadIdx.x], r[b], v);
        v2[threadIdx.x] = v;
    }
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;           // index in the new padded vector
    if (i >= Np1) return;
    double a = 0.0;
    if (i >= off1) {
        const int k = i - off1;
        if (k < N0) {
            a = alpha0[off0 + k];
            for (int c = 0; c < m; ++c) a = fma(Y2[(long)(off0 + k) * SR_NB + pf + c], v2[c], a);
        } else {
            const int q = k - N0;
            for (int c = q; c < m; ++c) a = fma(invS[(pf + q) * SR_NB + pf + c], v2[c], a);
        }
    }
    alpha1[i] = a;
}

int sr_launch_append_alpha(const double* alpha0, int Np0, int N0, const double* Y2, const double* invS,
                           const double* mu_part, int nsplit, int n_out, int d, long Tp, const double* Ynew, int m,
                           double* alpha1, int Np1, hipStream_t s, int qoff, int nbatch, long sY2) {
    hipLaunchKernelGGL(sr_append_alpha_kernel, dim3((Np1 + 255) / 256, nbatch), dim3(256), 0, s, alpha0, Np0, N0, Y2, invS,
                       mu_part, nsplit, n_out, d, Tp, Ynew, m, alpha1, Np1, qoff, sY2);
    SR_HIP(hipGetLastError());
    return SR_OK;
}

int sr_launch_append_small(const double* U12t, const double* Wt0, int Np0, int m, int stage, double* G,
                           const double* invS, double* Xt, double* Y2, hipStream_t s, int nbatch) {
    if (stage == 0) {
        hipLaunchKernelGGL(sr_append_gsmall_kernel, dim3(m, m, nbatch), dim3(256), 0, s, U12t, Np0, m, G);
    } else {
        hipLaunchKernelGGL(sr_append_xt_kernel, dim3((Np0 + 255) / 256, m), dim3(256), 0, s, U12t, invS, Np0, m, Xt, 0L);
        SR_HIP(hipGetLastError());
        if (m <= 1)
            hipLaunchKernelGGL(sr_append_y2_kernel<1>, dim3((Np0 + 3) / 4), dim3(256), 0, s, Wt0, Np0, Xt, m, Y2);
        else if (m <= 4)
            hipLaunchKernelGGL(sr_append_y2_kernel<4>, dim3((Np0 + 3) / 4), dim3(256), 0, s, Wt0, Np0, Xt, m, Y2);
        else
            hipLaunchKernelGGL(sr_append_y2_kernel<16>, dim3((Np0 + 3) / 4), dim3(256), 0, s, Wt0, Np0, Xt, m, Y2);
    }
    SR_HIP(hipGetLastError());
    return SR_OK;
}

// one wavefront per row, eight independent loads per lane in flight; shuffle reduction.  blockIdx.y = batch member
// (matrix, x and y `sM`, `sx`, `sy` doubles apart).  (One load per lane and iteration: 46 us per 105 MB triangle; the four
// products behind a model update with two outputs were 0.18 ms at its very end.)
__global__ __launch_bounds__(256) void sr_trmv_kernel(const double* __restrict__ M, long ld,
                                                      const double* __restrict__ x,
                                                      double* __restrict__ y, int n, int lower, long sM, long sx,
                                                      long sy) {
    // longest rows first (a row is one wavefront's chain of dependent load batches: scheduled last, the 5120-element
    // rows of a lower triangle were the tail of the launch)
    const int slot = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (slot >= n) return;
    const int row = lower ? n - 1 - slot : slot;
    M += (long)blockIdx.y * sM; x += (long)blockIdx.y * sx; y += (long)blockIdx.y * sy;
    const int c0 = lower ? 0 : row, c1 = lower ? row + 1 : n;
    const double* m = M + (long)row * ld;
    double acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = 0.0;
    int c = c0 + lane;
    for (; c + 7 * 64 < c1; c += 8 * 64) {
        double mv[8], xv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { mv[u] = m[c + 64 * u]; xv[u] = x[c + 64 * u]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] = fma(mv[u], xv[u], acc[u]);
    }
    for (; c < c1; c += 64) acc[0] = fma(m[c], x[c], acc[0]);
    double s = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) y[row] = s;
}

int sr_launch_trmv(const double* M, long ld, const double* x, double* y, int n, int lower,
                   hipStream_t s, int nbatch, long sM, long sx, long sy) {
    hipLaunchKernelGGL(sr_trmv_kernel, dim3((n + 3) / 4, nbatch), dim3(256), 0, s, M, ld, x, y, n, lower, sM, sx, sy);
    SR_HIP(hipGetLastError());
    return SR_OK;
}

__global__ __launch_bounds__(256) void sr_fill_kernel(double* p, size_t n, double v) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) p[i] = v;
}

int sr_launch_fill(double* p, size_t n, double v, hipStream_t s) {
    if (n == 0) return SR_OK;
    size_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(sr_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p, n, v);
    SR_HIP(hipGetLastError());
    return SR_OK;
}

// ------------------------------------------------------------------------------------------------
// log det(K_y) per output from the diagonal of U^-1:  log det = -2 sum_i log (U^-1)_ii.
// One workgroup per output; the padding block has unit diagonal and contributes nothing.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sr_logdet_kernel(const double* __restrict__ Wt, int Np,
                                                        double* __restrict__ out) {
    __shared__ double red[4];
    const double* W = Wt + (size_t)blockIdx.x * Np * Np;
    double acc = 0.0;
    for (int i = threadIdx.x; i < Np; i += 256) acc += log(W[(size_t)i * Np + i]);
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = -2.0 * (red[0] + red[1] + red[2] + red[3]);
}

int sr_launch_logdet(const double* Wt, int Np, int n_out, double* out, hipStream_t s) {
    hipLaunchKernelGGL(sr_logdet_kernel, dim3(n_out), dim3(256), 0, s, Wt, Np, out);
    SR_HIP(hipGetLastError());
    return SR_OK;
}
