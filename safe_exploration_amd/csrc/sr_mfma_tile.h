// sr_mfma_tile.h -- fp64 MFMA "TN" tile main loop for gfx950 (v_mfma_f64_16x16x4_f64).
//
//   acc[m][n] += sum_{k in [k_beg, k_end)} A[k][m] * B[k][n]
//
// Both operands are k-major (row k holds contiguous m resp. n), which is how this library stores
// every matrix it multiplies (U, W, Wt, K*): a BK x 128 tile is BK fully coalesced 1 KiB rows and
// lands in LDS unchanged; MFMA fragments are then read with conflict-free ds_read_b64
// (row stride 144 doubles = 288 dwords == 32 mod 64 -> the two k-rows a 32-lane group touches sit on
// opposite bank halves).
//
// Workgroup: 256 threads = 4 wavefronts (2 x 2), each wavefront owns a 64 x 64 sub-tile
// = 4 x 4 MFMA tiles of 16 x 16 (64 accumulator doubles / lane).  Global->LDS is register-staged and
// double-buffered: the loads for k-tile t+1 are issued before the MFMAs of k-tile t.
#pragma once
#include "sr_common.h"

namespace srt {

constexpr int BM = 128, BN = 128, BK = 16;
constexpr int LDT = 144;                     // padded LDS row (doubles)
constexpr int STAGE = BK * LDT;              // doubles per operand per stage
constexpr int SMEM_DOUBLES = 4 * STAGE;      // A,B x 2 stages  (73,728 B)

struct Acc {
    d4_t v[4][4];
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) v[i][j] = d4_t{0.0, 0.0, 0.0, 0.0};
    }
};

// element (mi, ni, r) of the accumulator is C[row][col] with
//   row = wm*64 + mi*16 + (lane>>4) + 4*r ,  col = wn*64 + ni*16 + (lane&15)
__device__ __forceinline__ int acc_row(int wm, int mi, int lane, int r) {
    return wm * 64 + mi * 16 + (lane >> 4) + 4 * r;
}
__device__ __forceinline__ int acc_col(int wn, int ni, int lane) {
    return wn * 64 + ni * 16 + (lane & 15);
}

// Contract of the main loops below: A points at A[0][m0], B at B[0][n0] (row strides lda / ldb in doubles); k range
// [k_beg, k_end), both multiples of BK.  smem: SMEM_DOUBLES doubles.  All 256 threads must call.
// (The register-staged loop of round 1 -- global_load -> VGPR -> ds_write, 205 VGPRs, 66.5 TF -- is gone since round 5.)

// Rounds 1 - 4: global->LDS through the LDS-DMA path
// (global_load_lds_dwordx4: no staging VGPRs, no ds_write pass).  Each wavefront-instruction moves one
// 1 KiB tile row (lane-linear destination == our row layout; the padding sits between rows).
// Two LDS stages; the barrier at the top of a k-step carries the vmcnt(0) that retires the DMA of the
// tile about to be read, and frees the other stage for the next DMA.
#define SRT_AS1(p) ((const __attribute__((address_space(1))) void*)(p))
#define SRT_AS3(p) ((__attribute__((address_space(3))) void*)(p))
__device__ __forceinline__ int acc_row_ilv(int wm, int mi, int lane, int r) {
    return (2 * mi + wm) * 16 + (lane >> 4) + 4 * r;
}

// (kept as the A/B reference of the pipelined loop below: variant 1 of sr_var_kernel, same bits)
template <int BKT>
__device__ __forceinline__ void mainloop_tn_glds(const double* __restrict__ A, long lda,
                                                 const double* __restrict__ B, long ldb,
                                                 int k_beg, int k_end, double* smem, Acc& acc) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    double* As = smem;
    constexpr int STG = BKT * LDT;             // doubles per operand per stage
    double* Bs = smem + 2 * STG;
    if (k_beg >= k_end) return;

    const double* ga = A + (long)wave * lda + 2 * lane;   // rows wave, wave+4, wave+8, wave+12
    const double* gb = B + (long)wave * ldb + 2 * lane;
    const int srow = wave * LDT;

#define SRT_DMA(k0, buf)                                                                             \
    do {                                                                                             \
        const double* pa_ = ga + (long)(k0) * lda;                                                   \
        const double* pb_ = gb + (long)(k0) * ldb;                                                   \
        double* sa_ = As + (buf) * STG + srow;                                                       \
        double* sb_ = Bs + (buf) * STG + srow;                                                       \
        _Pragma("unroll") for (int j_ = 0; j_ < BKT / 4; ++j_) {                                     \
            __builtin_amdgcn_global_load_lds(SRT_AS1(pa_ + 4 * j_ * lda), SRT_AS3(sa_ + 4 * j_ * LDT), 16, 0, 0); \
            __builtin_amdgcn_global_load_lds(SRT_AS1(pb_ + 4 * j_ * ldb), SRT_AS3(sb_ + 4 * j_ * LDT), 16, 0, 0); \
        }                                                                                            \
    } while (0)

    SRT_DMA(k_beg, 0);
    // A-fragment of row tile mi: columns (rows of the product) wm*64 + mi*16 .. , or (2 mi + wm)*16 .. when interleaved
    const int fa = (lane >> 4) * LDT + wm * 64 + (lane & 15);
    constexpr int FAS = 16;                    // distance of consecutive row tiles of one wavefront
    const int fb = (lane >> 4) * LDT + wn * 64 + (lane & 15);
    int buf = 0;
    for (int k0 = k_beg; k0 < k_end; k0 += BKT) {
        __syncthreads();                       // vmcnt(0) + s_barrier: tile k0 landed, other stage is free
        if (k0 + BKT < k_end) SRT_DMA(k0 + BKT, buf ^ 1);
        const double* as = As + buf * STG + fa;
        const double* bs = Bs + buf * STG + fb;
#pragma unroll
        for (int kk = 0; kk < BKT / 4; ++kk) {
            double af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                af[i] = as[kk * 4 * LDT + i * FAS];
                bf[i] = bs[kk * 4 * LDT + i * 16];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc.v[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[j], acc.v[i][j], 0, 0, 0);
        }
        buf ^= 1;
    }
    __syncthreads();                           // callers reuse smem after the main loop
#undef SRT_DMA
}

// ------------------------------------------------------------------------------------------------------------------
// Round 5: the same contract with the k-tile boundary crossed UNDER the MFMA stream (scripts/mfma_pipe_tile.hip,
// profiles/archive/r05_mfma_pipe_tile.txt: 69.7 TF for the skeleton of mainloop_tn_glds, 77.8 TF for this one on the same box).
// What the old loop lost: after its barrier at the top of a k-tile a wavefront issued its DMA burst, then its first
// fragment reads, waited for them, and only then its first MFMA; and hipcc's s_waitcnt pass, which stops counting LDS
// reads in order once a global_load_lds builtin is in the loop, waited lgkmcnt(0) before EVERY block of 16 MFMAs, i.e. for
// reads it had just issued.  While a wavefront sits in such a gap the other wavefront of its SIMD runs alone, and a lone
// wavefront reaches 74 % of the fp64 matrix pipe whatever it does (57 TF with one workgroup per CU, with or without
// double-buffered fragments).  Here:
//   * fragments are double-buffered in registers: the reads of k-step kk + 1 are issued before the MFMAs of kk;
//   * the barrier of a tile sits in front of its LAST block of 16 MFMAs: by then the wavefront has issued (and waited
//     for) all its reads of this stage, so behind the barrier it issues the first reads of the NEXT tile and the DMA of the
//     tile after that into the stage just freed -- under 16 MFMAs that are already its own;
//   * the eight DMA instructions are spread over those MFMAs (one per two), SGPR base + one loop-invariant VGPR offset,
//     through inline asm so that the compiler's waitcnt pass keeps counting (lgkmcnt(4) / (6) instead of (0));
//   * two k-tiles per trip, so the stage is a compile-time constant: LDS offsets are immediates, no VALU between MFMAs.
// One barrier per k-tile and two LDS stages, as before.  Same order of accumulation: identical results bit for bit.
// ------------------------------------------------------------------------------------------------------------------
// DIAG: the LAST 128 k of the range multiply a diagonal block of an upper-triangular A (A[k][m] == 0 for k > m inside
// the block).  In 16 x 16 sub-blocks only the pairs (k-tile kt <= row tile it) carry numbers -- 36 of 64 -- and the plain
// loop spends a full MFMA on each of the other 28 (2.6 % of all MFMAs of the triangular contraction at N = 5000).  With
// DIAG the two wavefront rows own INTERLEAVED row tiles (wm = 0: it = 0, 2, 4, 6; wm = 1: it = 1, 3, 5, 7) instead of
// the upper and the lower half, so that both lose work at the same pace (16 resp. 20 of 32 pairs each), and the eight
// diagonal k-tiles run fully unrolled with the row tiles kt > it left out at compile time.  A workgroup advances at the
// pace of its slower wavefront row: 20 / 32 of the diagonal block's time.  (Skipping with the half / half assignment --
// 10 resp. 26 of 32 -- buys nothing: measured in round 1.)  Callers must use acc_row_ilv for the row of an accumulator.
// Measured at C2' (N = 5000, 65536 queries): 73.1 -> 74.5 TFLOP/s with the pipelined loop (round 4's loop: 69.8 -> 70.05).
struct Frag { double a[4], b[4]; };

// one LDS-DMA instruction: 1 KiB tile row from gbase (SGPR pair) + voff (bytes, per lane) to LDS byte address lds0 + IMM
// (M0 is a reserved register: the compiler loads it in front of each of its own uses and keeps nothing in it)
// (COH, a template parameter of the loops below: the loads are agent-scope, sc1 -- for operands that other workgroups, on
//  other XCDs, have rewritten since this XCD's L2 may last have seen them: the tile-flow Cholesky, sr_flow.hip)
#define SRT_DMA1(lds0_, IMM_, voff_, gbase_)                                                        \
    do {                                                                                            \
        if constexpr (COH)                                                                          \
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 sc1"            \
                         :: "s"((lds0_) + (unsigned)(IMM_)), "v"(voff_), "s"(gbase_) : "memory");     \
        else                                                                                        \
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"                \
                         :: "s"((lds0_) + (unsigned)(IMM_)), "v"(voff_), "s"(gbase_) : "memory");     \
    } while (0)

template <bool DIAG = false, bool COH = false>
__device__ __forceinline__ void mainloop_tn_pipe(const double* __restrict__ A, long lda,
                                                 const double* __restrict__ B, long ldb,
                                                 int k_beg, int k_end, double* smem, Acc& acc) {
    constexpr int STG = BK * LDT;              // doubles per operand per stage; layout [A st0 | A st1 | B st0 | B st1]
    if (k_beg >= k_end) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const double* As = smem;
    const double* Bs = smem + 2 * STG;
    const int fa = (lane >> 4) * LDT + (DIAG ? wm * 16 : wm * 64) + (lane & 15);
    constexpr int FAS = DIAG ? 32 : 16;        // distance of consecutive row tiles of one wavefront (interleaved with DIAG)
    const int fb = (lane >> 4) * LDT + wn * 64 + (lane & 15);
    // DMA: wavefront w moves rows w, w + 4, w + 8, w + 12 of both operand tiles
    const unsigned voff = (unsigned)lane * 16u;
    const unsigned lds0 = (unsigned)(size_t)(smem + wave * LDT);
    const double* ga = A + (long)(k_beg + wave) * lda;       // row `wave` of the current DMA tile (uniform)
    const double* gb = B + (long)(k_beg + wave) * ldb;
    const long sa4 = 4 * lda, sb4 = 4 * ldb;

#define SRT_RD(F, st, kk)                                                                           \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                              \
        (F).a[i_] = As[(st) * STG + (kk) * 4 * LDT + i_ * FAS + fa];                                \
        (F).b[i_] = Bs[(st) * STG + (kk) * 4 * LDT + i_ * 16 + fb];                                 \
    }
    // MI0: first live row tile (DIAG: row tiles below it hold structural zeros in this k-tile)
#define SRT_MF(F, MI0)                                                                              \
    _Pragma("unroll") for (int i_ = (MI0); i_ < 4; ++i_)                                            \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                            \
            acc.v[i_][j_] = __builtin_amdgcn_mfma_f64_16x16x4f64((F).a[i_], (F).b[j_], acc.v[i_][j_], 0, 0, 0);
#define SRT_SB __builtin_amdgcn_sched_barrier(0)
    // the whole DMA of the tile at (ga, gb) into stage st, as a burst (prologue only)
#define SRT_DMA_TILE(st)                                                                            \
    _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_) {                                              \
        SRT_DMA1(lds0, ((st) * STG + 4 * r_ * LDT) * 8, voff, ga + r_ * sa4);                       \
        SRT_DMA1(lds0, ((2 + (st)) * STG + 4 * r_ * LDT) * 8, voff, gb + r_ * sb4);                 \
    }
    // last MFMA block of a tile (fragments F, rows >= MI0); with `dma` (wavefront-uniform) the DMA of the tile at (ga, gb)
    // into stage st is spread over it -- the branches go around the asm statements only, the MFMAs are unconditional
#define SRT_MF_DMA(F, MI0, st, dma)                                                                 \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                              \
        if (i_ >= (MI0)) {                                                                          \
            acc.v[i_][0] = __builtin_amdgcn_mfma_f64_16x16x4f64((F).a[i_], (F).b[0], acc.v[i_][0], 0, 0, 0); \
            acc.v[i_][1] = __builtin_amdgcn_mfma_f64_16x16x4f64((F).a[i_], (F).b[1], acc.v[i_][1], 0, 0, 0); \
        }                                                                                           \
        SRT_SB;                                                                                     \
        if (dma) SRT_DMA1(lds0, ((st) * STG + 4 * i_ * LDT) * 8, voff, ga + i_ * sa4);              \
        SRT_SB;                                                                                     \
        if (i_ >= (MI0)) {                                                                          \
            acc.v[i_][2] = __builtin_amdgcn_mfma_f64_16x16x4f64((F).a[i_], (F).b[2], acc.v[i_][2], 0, 0, 0); \
            acc.v[i_][3] = __builtin_amdgcn_mfma_f64_16x16x4f64((F).a[i_], (F).b[3], acc.v[i_][3], 0, 0, 0); \
        }                                                                                           \
        SRT_SB;                                                                                     \
        if (dma) SRT_DMA1(lds0, ((2 + (st)) * STG + 4 * i_ * LDT) * 8, voff, gb + i_ * sb4);        \
        SRT_SB;                                                                                     \
    }
    // A tile on stage st with the fragments of its k-step 0 already in F0; leaves the next tile's k-step 0 in F0 (read in any
    // case: without a next tile it is a harmless read of the other stage).  next: another tile follows (barrier); dma: and
    // one after that, whose DMA goes into this stage.  Both wavefront-uniform.
#define SRT_TILE(st, MI0, next, dma)                                                                \
    do {                                                                                            \
        SRT_RD(F1, st, 1); SRT_SB; SRT_MF(F0, MI0); SRT_SB;                                         \
        SRT_RD(F0, st, 2); SRT_SB; SRT_MF(F1, MI0); SRT_SB;                                         \
        SRT_RD(F1, st, 3); SRT_SB; SRT_MF(F0, MI0); SRT_SB;                                         \
        if (next) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");           \
        SRT_RD(F0, (st) ^ 1, 0); SRT_SB;                                                            \
        if (dma) { ga += 16 * lda; gb += 16 * ldb; }                                                \
        SRT_MF_DMA(F1, MI0, st, dma);                                                               \
    } while (0)

    Frag F0, F1;
    const int nt = (k_end - k_beg) / BK;
    const bool diag = DIAG && nt >= 8;         // the masked walk needs the whole diagonal block inside the range
    // Tiles go in PAIRS (stage 0, stage 1); an odd count starts with one tile on stage 1.  A pair at tile t: its first tile
    // always has a successor; `more` = the pair has a successor pair = everything else it needs to know.
    const int s0 = nt & 1;
    // prologue: the DMAs of tiles 0 and 1 (both stages are free), wait for tile 0 only
    if (s0) { SRT_DMA_TILE(1); } else { SRT_DMA_TILE(0); }
    if (nt > 1) {
        ga += 16 * lda; gb += 16 * ldb;
        if (s0) { SRT_DMA_TILE(0); } else { SRT_DMA_TILE(1); }
        asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    }
    // (ga, gb) point at the LAST tile whose DMA was issued; SRT_TILE advances them before it issues the next
    int t = 0;
    if (s0) {
        SRT_RD(F0, 1, 0);
        const bool nx = nt > 1, dm = nt > 2;
        SRT_TILE(1, 0, nx, dm);
        t = 1;
    } else {
        SRT_RD(F0, 0, 0);
    }
    const int t_gen = diag ? nt - 8 : nt;      // pairs before the diagonal block (nt - 8 - s0 is even)
    for (; t < t_gen; t += 2) {
        const bool more = t + 2 < nt;
        SRT_TILE(0, 0, true, more);
        SRT_TILE(1, 0, more, more);
    }
    if (diag) {
        // k-tile kt of the diagonal block: row tile it = 2 mi + wm carries numbers iff it >= kt; both wavefront rows use the
        // live set mi >= kt / 2 (exact for wm = 1, one block of zeros too many per odd kt for wm = 0)
        SRT_TILE(0, 0, true, true);
        SRT_TILE(1, 0, true, true);
        SRT_TILE(0, 1, true, true);
        SRT_TILE(1, 1, true, true);
        SRT_TILE(0, 2, true, true);
        SRT_TILE(1, 2, true, true);
        SRT_TILE(0, 3, true, false);
        SRT_TILE(1, 3, false, false);
    }
    __syncthreads();                           // callers reuse smem after the main loop
#undef SRT_RD
#undef SRT_MF
#undef SRT_SB
#undef SRT_DMA_TILE
#undef SRT_MF_DMA
#undef SRT_TILE
}

// Round 3, measured and NOT kept (scripts/mfma_lds_tile.hip, profiles/archive/r03_mfma_lds_tile.txt): where the 9 % between this
// loop (70.5 TF inside sr_var_kernel) and the matrix pipe (77.5 TF) go.  An LDS-fed loop of the same fragment reads and
// MFMAs runs at 77.0 - 77.7 TF for EVERY wavefront tile from 32 x 32 to 96 x 64 at two wavefronts per SIMD: the LDS -> VGPR
// traffic costs nothing, so a 96 x 64 wavefront tile has nothing to win (and its 41 KB stages would leave one workgroup per
// CU: 58 TF).  A barrier per k-tile: 77.0.  The LDS-DMA of the next k-tile, no barrier: 74.5 (-3.5 %); DMA + barrier:
// 72.0 (-3 % more), the same with one, two or three k-tiles in flight (s_waitcnt vmcnt(0 / 8 / 16)) -- it is not the DMA's
// latency.  72.0 is what this skeleton can reach; the kernel is at 0.98 of it (the rest: structural zeros of the
// diagonal blocks, prologue, epilogue).  Two rewrites of the loop confirmed it: four stages of 8 k-rows with three tiles
// in flight (equal), and the barrier moved to the middle of a tile with the next tile's first fragments read across the
// tile boundary (66.5 TF: twice the barriers, and hipcc hoists the barrier back to the top of the tile).
}  // namespace srt

// ------------------------------------------------------------------------------------------------
// 64 x 64 workgroup tile, same contract (k-major operands): 4 wavefronts (2 x 2) of 32 x 32 = 2 x 2 MFMA tiles.
// One fp64 MFMA occupies its SIMD for 64 cycles, so a 128 x 128 x 128 tile product is 14 us of one CU however
// it is scheduled; products with FEW 128-tiles (the block row / look-ahead row of the Cholesky, the small and the
// triangular levels of the inversion) are latency- and balance-bound, and a 4 x finer tile cuts both by 4.
// LDS-DMA staging: one global_load_lds_dwordx4 moves 1 KiB = TWO 64-double tile rows (lanes 0-31: row r,
// lanes 32-63: row r + 2); the pair is contiguous in LDS, pairs are 144 doubles apart.  With rows (r, r + 2) paired
// the k-rows lk = 0, 1 of a 32-lane ds_read_b64 group sit in different pairs, i.e. on opposite bank halves
// (pair stride 288 dwords == 32 mod 64): conflict-free fragment reads.  36 KiB of LDS: 4 workgroups per CU.
// ------------------------------------------------------------------------------------------------
namespace srt64 {

constexpr int BM = 64, BN = 64, BK = 16;
constexpr int LDP = 144;                     // doubles between row pairs
constexpr int STAGE = (BK / 2) * LDP;        // doubles per operand per stage
// Round 3: FOUR stages, three k-tiles in flight.  A k-tile of this tile is 16 MFMAs per wavefront (0.43 us); with two
// stages the single DMA in flight exposed the whole global -> LDS latency on every k-step (1.5 us per step measured:
// the K = 128 block-row solve of the Cholesky took 14 us for 3.4 us of MFMAs).  The barrier of a k-step waits for
// exactly the DMAs of the stage about to be read (s_waitcnt vmcnt(4 x stages still in flight), by hand: the
// __syncthreads() of the two-stage loop carries a vmcnt(0)).  72 KiB of LDS: 2 workgroups per CU.
constexpr int NS = 4;
constexpr int SMEM_DOUBLES = 2 * NS * STAGE; // A, B x NS stages (73,728 B)

struct Acc {
    d4_t v[2][2];
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) v[i][j] = d4_t{0.0, 0.0, 0.0, 0.0};
    }
};
// element (mi, ni, r): row = wm*32 + mi*16 + (lane>>4) + 4*r ,  col = wn*32 + ni*16 + (lane&15)
__device__ __forceinline__ int acc_row(int wm, int mi, int lane, int r) { return wm * 32 + mi * 16 + (lane >> 4) + 4 * r; }
__device__ __forceinline__ int acc_col(int wn, int ni, int lane) { return wn * 32 + ni * 16 + (lane & 15); }

// LDS offset (doubles) of tile row r inside a stage: rows (r, r + 2) share a pair
__device__ __forceinline__ int row_off(int r) { return (((r >> 2) << 1) + (r & 1)) * LDP + ((r >> 1) & 1) * 64; }

// wait until at most `stages` stages' DMAs (4 instructions per wavefront each) are still in flight, then the barrier;
// "memory": LDS reads stay behind it, the DMA of the stage it frees stays behind it too
__device__ __forceinline__ void wait_stage_barrier(int stages) {
    if (stages >= 2) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
    else if (stages == 1) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ void mainloop_tn(const double* __restrict__ A, long lda, const double* __restrict__ B,
                                            long ldb, int k_beg, int k_end, double* smem, Acc& acc) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    double* As = smem;
    double* Bs = smem + NS * STAGE;
    if (k_beg >= k_end) return;
    // wavefront w moves pairs w and w + 4 of each operand: pair q = rows (4 (q >> 1) + (q & 1), that + 2)
    const int half = lane >> 5, l32 = lane & 31;
    const int r0 = 4 * (wave >> 1) + (wave & 1) + 2 * half;            // row of pair `wave` this lane loads
    const double* ga = A + (long)r0 * lda + 2 * l32;
    const double* gb = B + (long)r0 * ldb + 2 * l32;
    const int so = wave * LDP;                                          // pair `wave`; pair wave + 4 is 8 rows further
#define SRT64_DMA(k0, buf)                                                                             \
    do {                                                                                               \
        const double* pa_ = ga + (long)(k0) * lda;                                                     \
        const double* pb_ = gb + (long)(k0) * ldb;                                                     \
        double* sa_ = As + (buf) * STAGE + so;                                                         \
        double* sb_ = Bs + (buf) * STAGE + so;                                                         \
        __builtin_amdgcn_global_load_lds(SRT_AS1(pa_), SRT_AS3(sa_), 16, 0, 0);                        \
        __builtin_amdgcn_global_load_lds(SRT_AS1(pb_), SRT_AS3(sb_), 16, 0, 0);                        \
        __builtin_amdgcn_global_load_lds(SRT_AS1(pa_ + 8 * lda), SRT_AS3(sa_ + 4 * LDP), 16, 0, 0);    \
        __builtin_amdgcn_global_load_lds(SRT_AS1(pb_ + 8 * ldb), SRT_AS3(sb_ + 4 * LDP), 16, 0, 0);    \
    } while (0)
    const int nsteps = (k_end - k_beg) / BK;
    // everything of this wavefront that is older than the DMAs below (a caller's loads) must not count against them
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int s_ = 0; s_ < NS - 1; ++s_)
        if (s_ < nsteps) SRT64_DMA(k_beg + s_ * BK, s_);
    const int lk = lane >> 4, ln = lane & 15;
    int buf = 0, nbuf = NS - 1;                // stage read by step t, stage filled by step t (t + NS - 1's data)
    for (int t = 0; t < nsteps; ++t) {
        // issued so far: min(nsteps, t + NS - 1) stages; stage t must have landed
        const int issued = (t + NS - 1 < nsteps) ? t + NS - 1 : nsteps;
        wait_stage_barrier(issued - (t + 1));
        if (t + NS - 1 < nsteps) SRT64_DMA(k_beg + (t + NS - 1) * BK, nbuf);
        const double* as = As + buf * STAGE + wm * 32 + ln;
        const double* bs = Bs + buf * STAGE + wn * 32 + ln;
#pragma unroll
        for (int kk = 0; kk < BK / 4; ++kk) {
            const int ro = row_off(4 * kk + lk);
            double af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[i] = as[ro + i * 16];
                bf[i] = bs[ro + i * 16];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc.v[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[j], acc.v[i][j], 0, 0, 0);
        }
        buf = (buf + 1 == NS) ? 0 : buf + 1;
        nbuf = (nbuf + 1 == NS) ? 0 : nbuf + 1;
    }
    __syncthreads();                           // callers reuse smem after the main loop
#undef SRT64_DMA
}

// Round 5: the 64 x 64 loop in the pipelined form of srt::mainloop_tn_pipe (see there): fragments of k-step kk + 1 read
// under the four MFMAs of kk, the barrier of a 16-row k-tile in front of its LAST four MFMAs, behind it the first reads of
// the next tile and the DMA of the tile four ahead (FOUR stages: three tiles in flight, as before), the four DMA
// instructions spread over those MFMAs, inline asm so that the compiler's waitcnt pass keeps counting, four k-tiles per trip
// so that the stage is a compile-time constant.  Same order of accumulation as mainloop_tn: identical bits.
struct Frag { double a[2], b[2]; };

template <bool COH = false>
__device__ __forceinline__ void mainloop_tn_pipe(const double* __restrict__ A, long lda, const double* __restrict__ B,
                                                 long ldb, int k_beg, int k_end, double* smem, Acc& acc) {
    if (k_beg >= k_end) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const double* As = smem;
    const double* Bs = smem + NS * STAGE;
    const int nsteps = (k_end - k_beg) / BK;
    // DMA: wavefront w moves pairs w and w + 4 of each operand; lanes 0 - 31 one row of a pair, lanes 32 - 63 the row two below
    const int half = lane >> 5, l32 = lane & 31;
    const unsigned voa = (unsigned)((2L * half * lda + 2 * l32) * 8);
    const unsigned vob = (unsigned)((2L * half * ldb + 2 * l32) * 8);
    const int r0 = 4 * (wave >> 1) + (wave & 1);                      // first row of pair `wave` (uniform)
    const double* ga = A + (long)(k_beg + r0) * lda;                  // tile whose DMA is issued next
    const double* gb = B + (long)(k_beg + r0) * ldb;
    const unsigned lds0 = (unsigned)(size_t)(smem + wave * LDP);
    // fragment reads: row_off(4 kk + lk) = 2 kk LDP + [(lk & 1) LDP + ((lk >> 1) & 1) 64]
    const int lk = lane >> 4, ln = lane & 15;
    const int fo = (lk & 1) * LDP + ((lk >> 1) & 1) * 64 + ln;
    const int fa = fo + wm * 32, fb = fo + wn * 32;

#define S64_DMA1(IMM_, voff_, gbase_)                                                               \
    do {                                                                                            \
        if constexpr (COH)                                                                          \
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 sc1"            \
                         :: "s"(lds0 + (unsigned)(IMM_)), "v"(voff_), "s"(gbase_) : "memory");        \
        else                                                                                        \
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"                \
                         :: "s"(lds0 + (unsigned)(IMM_)), "v"(voff_), "s"(gbase_) : "memory");        \
    } while (0)
    // instruction q (0 .. 3) of the DMA of the tile at (ga, gb) into stage st: A pair w, B pair w, A pair w + 4, B pair w + 4
#define S64_DMAQ(st, q)                                                                             \
    do {                                                                                            \
        if ((q) == 0) S64_DMA1(((st) * STAGE) * 8, voa, ga);                                        \
        else if ((q) == 1) S64_DMA1(((NS + (st)) * STAGE) * 8, vob, gb);                            \
        else if ((q) == 2) S64_DMA1(((st) * STAGE + 4 * LDP) * 8, voa, ga + 8 * lda);               \
        else S64_DMA1(((NS + (st)) * STAGE + 4 * LDP) * 8, vob, gb + 8 * ldb);                      \
    } while (0)
#define S64_RD(F, st, kk)                                                                           \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                              \
        (F).a[i_] = As[(st) * STAGE + 2 * (kk) * LDP + i_ * 16 + fa];                               \
        (F).b[i_] = Bs[(st) * STAGE + 2 * (kk) * LDP + i_ * 16 + fb];                               \
    }
#define S64_MF(F)                                                                                   \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_)                                                \
        _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                            \
            acc.v[i_][j_] = __builtin_amdgcn_mfma_f64_16x16x4f64((F).a[i_], (F).b[j_], acc.v[i_][j_], 0, 0, 0);
#define S64_SB __builtin_amdgcn_sched_barrier(0)
    // k-tile t on stage st, fragments of its k-step 0 in F0.  Behind its barrier: tile t + 1 has landed (the tiles t + 2,
    // t + 3 -- as far as they exist -- may still be in flight), stage st is free for tile t + 4.
#define S64_STEP(st, t)                                                                             \
    do {                                                                                            \
        S64_RD(F1, st, 1); S64_SB; S64_MF(F0); S64_SB;                                              \
        S64_RD(F0, st, 2); S64_SB; S64_MF(F1); S64_SB;                                              \
        S64_RD(F1, st, 3); S64_SB; S64_MF(F0); S64_SB;                                              \
        const int left_ = nsteps - 1 - (t);            /* tiles behind this one */                   \
        if (left_ >= 3) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");    \
        else if (left_ == 2) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory"); \
        else if (left_ == 1) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); \
        S64_RD(F0, ((st) + 1) & 3, 0); S64_SB;                                                      \
        const bool dma_ = left_ >= 4;                                                               \
        if (dma_) { ga += 16 * lda; gb += 16 * ldb; }                                               \
        acc.v[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(F1.a[0], F1.b[0], acc.v[0][0], 0, 0, 0); \
        S64_SB; if (dma_) S64_DMAQ(st, 0); S64_SB;                                                  \
        acc.v[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(F1.a[0], F1.b[1], acc.v[0][1], 0, 0, 0); \
        S64_SB; if (dma_) S64_DMAQ(st, 1); S64_SB;                                                  \
        acc.v[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(F1.a[1], F1.b[0], acc.v[1][0], 0, 0, 0); \
        S64_SB; if (dma_) S64_DMAQ(st, 2); S64_SB;                                                  \
        acc.v[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(F1.a[1], F1.b[1], acc.v[1][1], 0, 0, 0); \
        S64_SB; if (dma_) S64_DMAQ(st, 3); S64_SB;                                                  \
    } while (0)

    // everything of this wavefront that is older than the DMAs below (a caller's loads) must not count against them
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // prologue: the first four tiles (all stages are free), wait for the first
    const int npro = nsteps < NS ? nsteps : NS;
    _Pragma("unroll") for (int s_ = 0; s_ < NS; ++s_) {
        if (s_ < npro) {
            if (s_ > 0) { ga += 16 * lda; gb += 16 * ldb; }
            S64_DMAQ(s_, 0); S64_DMAQ(s_, 1); S64_DMAQ(s_, 2); S64_DMAQ(s_, 3);
        }
    }
    if (npro == 4) asm volatile("s_waitcnt vmcnt(12)\n\ts_barrier" ::: "memory");
    else if (npro == 3) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
    else if (npro == 2) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    Frag F0, F1;
    S64_RD(F0, 0, 0);
    for (int t = 0; t < nsteps; t += 4) {
        S64_STEP(0, t);
        if (t + 1 >= nsteps) break;
        S64_STEP(1, t + 1);
        if (t + 2 >= nsteps) break;
        S64_STEP(2, t + 2);
        if (t + 3 >= nsteps) break;
        S64_STEP(3, t + 3);
    }
    __syncthreads();                           // callers reuse smem after the main loop
#undef S64_DMA1
#undef S64_DMAQ
#undef S64_RD
#undef S64_MF
#undef S64_SB
#undef S64_STEP
}

}  // namespace srt64
