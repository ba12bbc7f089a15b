// Microbenchmark (round 5): a 128 x 128 x 16 fp64 MFMA k-tile loop whose per-k-tile workgroup barrier sits INSIDE the MFMA
// stream instead of in front of it.
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma_pipe_tile.hip -o /tmp/pipe && /tmp/pipe
// The shipped loop (mainloop_tn_glds, skeleton = mode 2 of scripts/mfma_lds_tile.hip, 72.1 TF) does per k-tile:
//   s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier; 8 x global_load_lds (burst); ds_reads of kk = 0; s_waitcnt lgkmcnt(0); 64 MFMAs
// so after every barrier a wavefront issues no MFMA until its DMA burst is out and its first fragments are back (~300
// cycles), and inside the tile every kk block waits for its fragments too; while one wavefront of a SIMD sits in such a gap
// the other runs alone, and a lone wavefront of this code reaches 75 % of the pipe (mode 0 with one workgroup per CU: 58.6 TF).
// Here: fragments are double-buffered in registers (the reads of kk + 1 are issued before the MFMAs of kk), and the tile
// boundary is crossed under the last MFMA block of a tile:
//   kk = 0 .. 2:  issue reads(kk + 1);  16 MFMAs(kk)
//   kk = 3:       s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier        (everyone's reads of this stage are done, the next tile landed)
//                 DMA(tile t + 2 -> this stage);  issue reads(tile t + 1, kk = 0);  16 MFMAs(kk = 3)
// One barrier per k-tile as before, two LDS stages as before; the MFMA stream never waits for LDS latency.
// MODE 0: as described, DMA as a burst of 8.  1: DMA spread over the MFMA block (sched_group_barrier).  2: no DMA, no barrier
// (ceiling of the fragment pipeline).  3: barrier, no DMA.  4: the shipped skeleton for reference (barrier at the top).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define AS1(p) ((const __attribute__((address_space(1))) void*)(p))
#define AS3(p) ((__attribute__((address_space(3))) void*)(p))
constexpr int LDT = 144, STG = 16 * LDT;

struct Frag { double a[4], b[4]; };

template <int MODE, int WPS>
__global__ __launch_bounds__(256, WPS) void kpipe(double* out, int iters, const double* src) {
    __shared__ double S[2 * 2 * STG];          // [operand][stage][16][LDT]
    for (int e = threadIdx.x; e < 4 * STG; e += 256) S[e] = 1e-3 * (e % 7);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wm = wave >> 1, wn = wave & 1;
    const double* As = S;
    const double* Bs = S + 2 * STG;
    const int fa = (lane >> 4) * LDT + wm * 64 + (lane & 15);
    const int fb = (lane >> 4) * LDT + wn * 64 + (lane & 15);
    const double* gsrc = src + (long)(blockIdx.x % 20) * 128 + (long)wave * 5120 + 2 * lane;
    d4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = d4{0, 0, 0, 0};

#define RD(F, st, kk)                                                                                  \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                 \
        (F).a[i_] = As[(st) * STG + (kk) * 4 * LDT + i_ * 16 + fa];                                    \
        (F).b[i_] = Bs[(st) * STG + (kk) * 4 * LDT + i_ * 16 + fb];                                    \
    }
#define MF(F)                                                                                          \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                   \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                               \
            acc[i_][j_] = __builtin_amdgcn_mfma_f64_16x16x4f64((F).a[i_], (F).b[j_], acc[i_][j_], 0, 0, 0);
#define SB __builtin_amdgcn_sched_barrier(0)
// The LDS-DMA goes through inline asm: with __builtin_amdgcn_global_load_lds in the loop the compiler's s_waitcnt pass gives
// up counting LDS reads in order and waits lgkmcnt(0) before every MFMA block -- i.e. for the fragment reads it has JUST
// issued (seen in the ISA of the first version of this file), which is exactly the stall the double buffer is there to hide.
#define DMA1(gp_, lp_)                                                                                 \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"                      \
                 :: "s"((unsigned)(size_t)(lp_)), "v"(gp_) : "memory")
#define DMA(it_, st)                                                                                   \
    do {                                                                                               \
        const double* g_ = gsrc + (long)((it_) & 511) * (16 * 5120);                                   \
        double* da_ = S + (st) * STG + wave * LDT;                                                     \
        double* db_ = S + 2 * STG + (st) * STG + wave * LDT;                                           \
        _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_) {                                             \
            DMA1(g_ + (4 * r_) * 5120L, da_ + 4 * r_ * LDT);                                           \
            DMA1(g_ + 2560 + (4 * r_) * 5120L, db_ + 4 * r_ * LDT);                                    \
        }                                                                                              \
    } while (0)

    Frag F0, F1;
    if (MODE == 5 || MODE == 6) {
        // two k-tiles per trip: the stage is a compile-time constant (LDS offsets are immediates, no VALU address
        // arithmetic between the MFMAs), the DMA takes an SGPR base + one loop-invariant VGPR offset per lane; MODE 5 spreads
        // the eight DMAs of a tile over the last MFMA block, MODE 6 issues them as a burst
        const unsigned voff = (unsigned)(((long)wave * 5120 + 2 * lane) * 8);
        const double* sbase = src + (long)(blockIdx.x % 20) * 128;
        const unsigned lds0 = (unsigned)(size_t)(S + wave * LDT);
#define DMAS(row_, bsel_, st_, tile_)                                                                  \
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"                       \
                     :: "s"(lds0 + (unsigned)(((bsel_) * 2 * STG + (st_) * STG + 4 * (row_) * LDT) * 8)), "v"(voff), \
                        "s"(sbase + (long)((tile_) & 511) * (16 * 5120) + (bsel_) * 2560 + (4 * (row_)) * 5120L) : "memory")
#define TILE(st_, it_)                                                                                 \
        RD(F1, st_, 1); SB; MF(F0); SB;                                                                \
        RD(F0, st_, 2); SB; MF(F1); SB;                                                                \
        RD(F1, st_, 3); SB; MF(F0); SB;                                                                \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");                       \
        RD(F0, (st_) ^ 1, 0); SB;                                                                      \
        if (MODE == 6) {                                                                               \
            _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_) { DMAS(r_, 0, st_, (it_) + 2); DMAS(r_, 1, st_, (it_) + 2); } \
            SB; MF(F1); SB;                                                                            \
        } else {                                                                                       \
            _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                         \
                acc[i_][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(F1.a[i_], F1.b[0], acc[i_][0], 0, 0, 0); \
                acc[i_][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(F1.a[i_], F1.b[1], acc[i_][1], 0, 0, 0); \
                SB; DMAS(i_, 0, st_, (it_) + 2); SB;                                                   \
                acc[i_][2] = __builtin_amdgcn_mfma_f64_16x16x4f64(F1.a[i_], F1.b[2], acc[i_][2], 0, 0, 0); \
                acc[i_][3] = __builtin_amdgcn_mfma_f64_16x16x4f64(F1.a[i_], F1.b[3], acc[i_][3], 0, 0, 0); \
                SB; DMAS(i_, 1, st_, (it_) + 2); SB;                                                   \
            }                                                                                          \
        }
        DMA(0, 0);
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        DMA(1, 1);
        RD(F0, 0, 0);
        for (int it = 0; it < iters; it += 2) {
            TILE(0, it)
            TILE(1, it + 1)
        }
    } else if (MODE == 4) {
        // the shipped skeleton: barrier at the top of the tile, DMA burst, reads, MFMAs (compiler-scheduled)
        for (int it = 0; it < iters; ++it) {
            const int st = it & 1;
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            DMA(it + 1, st ^ 1);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                RD(F0, st, kk);
                MF(F0);
            }
        }
    } else {
        if (MODE <= 1) DMA(0, 0);
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        if (MODE <= 1) DMA(1, 1);
        RD(F0, 0, 0);
        for (int it = 0; it < iters; ++it) {
            const int st = it & 1;
            RD(F1, st, 1); SB; MF(F0); SB;
            RD(F0, st, 2); SB; MF(F1); SB;
            RD(F1, st, 3); SB; MF(F0); SB;
            if (MODE == 0 || MODE == 1) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else if (MODE == 3) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else asm volatile("" ::: "memory");
            if (MODE == 0) {
                DMA(it + 2, st);
                RD(F0, st ^ 1, 0); SB; MF(F1); SB;
            } else if (MODE == 1) {
                // one DMA after every second MFMA of the block
                RD(F0, st ^ 1, 0); SB;
                const double* g_ = gsrc + (long)((it + 2) & 511) * (16 * 5120);
                double* da_ = S + st * STG + wave * LDT;
                double* db_ = S + 2 * STG + st * STG + wave * LDT;
#pragma unroll
                for (int i_ = 0; i_ < 4; ++i_) {
                    acc[i_][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(F1.a[i_], F1.b[0], acc[i_][0], 0, 0, 0);
                    acc[i_][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(F1.a[i_], F1.b[1], acc[i_][1], 0, 0, 0);
                    SB;
                    DMA1(g_ + (4 * i_) * 5120L, da_ + 4 * i_ * LDT);
                    SB;
                    acc[i_][2] = __builtin_amdgcn_mfma_f64_16x16x4f64(F1.a[i_], F1.b[2], acc[i_][2], 0, 0, 0);
                    acc[i_][3] = __builtin_amdgcn_mfma_f64_16x16x4f64(F1.a[i_], F1.b[3], acc[i_][3], 0, 0, 0);
                    SB;
                    DMA1(g_ + 2560 + (4 * i_) * 5120L, db_ + 4 * i_ * LDT);
                    SB;
                }
            } else {
                RD(F0, st ^ 1, 0); SB; MF(F1); SB;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    double s = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * 256 + threadIdx.x] = s + F0.a[0];
}

template <int MODE, int WPS>
void run(const double* src, const char* what) {
    const int blocks = 256 * WPS, iters = 4000;
    double* out; (void)hipMalloc(&out, sizeof(double) * blocks * 256);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    kpipe<MODE, WPS><<<blocks, 256>>>(out, 50, src);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        kpipe<MODE, WPS><<<blocks, 256>>>(out, iters, src);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double flops = 2.0 * 16 * 16 * 4 * 16 * 4.0 * iters * blocks * 4;
    hipFuncAttributes fa; (void)hipFuncGetAttributes(&fa, (const void*)kpipe<MODE, WPS>);
    printf("mode %d  %d workgroup(s)/CU  %-72s %6.2f TFLOP/s   regs %d  scratch %zu B\n", MODE, WPS, what, flops / best / 1e9, fa.numRegs,
           (size_t)fa.localSizeBytes);
    (void)hipFree(out);
}

int main() {
    const size_t nsrc = (size_t)512 * 16 * 5120 + (1 << 20);
    double* src; (void)hipMalloc(&src, sizeof(double) * nsrc);
    (void)hipMemset(src, 0, sizeof(double) * nsrc);
    run<4, 2>(src, "shipped skeleton: barrier on top, DMA burst, reads, MFMAs");
    run<2, 2>(src, "fragments double-buffered, no DMA, no barrier");
    run<2, 1>(src, "fragments double-buffered, no DMA, no barrier");
    run<3, 2>(src, "fragments double-buffered, barrier inside the MFMA stream, no DMA");
    run<0, 2>(src, "fragments double-buffered, barrier inside the MFMA stream, DMA burst behind it");
    run<1, 2>(src, "the same, DMA spread over the last MFMA block");
    run<6, 2>(src, "two k-tiles per trip (immediate LDS offsets), DMA saddr form, burst");
    run<5, 2>(src, "two k-tiles per trip (immediate LDS offsets), DMA saddr form, spread");
    run<0, 1>(src, "fragments double-buffered, barrier inside, DMA burst (one workgroup per CU)");
    run<4, 1>(src, "shipped skeleton (one workgroup per CU)");
    return 0;
}
