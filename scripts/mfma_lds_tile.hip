// Microbenchmark: what does the wavefront-tile SHAPE buy the fp64 MFMA contraction on gfx950?
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma_lds_tile.hip -o /tmp/tile && /tmp/tile
// The inner loop of sr_var_kernel without global loads and barriers: per k-step of 4 a wavefront reads NA A-fragments
// and NB B-fragments (ds_read_b64 each, the tile's conflict-free LDS layout) and issues NA x NB MFMAs 16x16x4.
// LDS -> VGPR fragment traffic per MFMA: (NA + NB) / (NA NB): 0.5 at 4 x 4 (the shipped 64 x 64 wavefront tile),
// 0.417 at 6 x 4 (96 x 64), 0.375 at 8 x 4.  The measured rate of each shape is the ceiling a kernel built on it can reach
// before DMA, barriers, prologue and epilogue take their share.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int LDT_BIG = 208;  // doubles per LDS row for tiles beyond 4 x 4: >= 16 * max(NA, NB) * 2 wavefront columns, == 32 mod 64 dwords like the tile's 144

// MODE 0: fragments + MFMAs only.  1: + one s_barrier per k-tile of 16 (64 MFMAs at 4 x 4).  2: + the LDS-DMA of the next
// k-tile (8 global_load_lds_dwordx4 per wavefront from a 64 MB buffer) with s_waitcnt vmcnt(0) before the barrier -- the
// skeleton of mainloop_tn_glds.  3: DMA without the barrier.  4 / 5: as 2, but the barrier waits only for the k-tile issued
// one / two tiles earlier (a deeper pipeline: s_waitcnt vmcnt(8) / vmcnt(16)).
template <int NA, int NB, int WPS, int MODE = 0>
__global__ __launch_bounds__(256, WPS) void k(double* out, int iters, const double* src = nullptr) {
    constexpr int LDT = (NA <= 4 && NB <= 4) ? 144 : LDT_BIG;      // 4 x 4: the tile's own layout, two workgroups per CU even with the DMA stage
    __shared__ double As[16 * LDT], Bs[16 * LDT];
    __shared__ double Ad[(MODE >= 2) ? 16 * 144 : 1], Bd[(MODE >= 2) ? 16 * 144 : 1];      // DMA target stage
    for (int e = threadIdx.x; e < 16 * LDT; e += 256) { As[e] = 1e-3 * (e % 7); Bs[e] = 1e-3 * (e % 5); }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wm = wave >> 1, wn = wave & 1;
    const double* as = As + (lane >> 4) * LDT + wm * 16 * NA + (lane & 15);
    const double* bs = Bs + (lane >> 4) * LDT + wn * 16 * NB + (lane & 15);
    const double* gsrc = (MODE >= 2) ? src + (long)(blockIdx.x % 20) * 128 + (long)wave * 5120 + 2 * lane : nullptr;
    d4 acc[NA][NB];
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = d4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            double af[NA], bf[NB];
#pragma unroll
            for (int i = 0; i < NA; ++i) af[i] = as[kk * 4 * LDT + i * 16];
#pragma unroll
            for (int j = 0; j < NB; ++j) bf[j] = bs[kk * 4 * LDT + j * 16];
#pragma unroll
            for (int i = 0; i < NA; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        asm volatile("" ::: "memory");            // the LDS contents "change": fragments are re-read every k-tile
        if (MODE == 2) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        if (MODE == 4) { asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }       // only the OLDER of two k-tiles in flight
        if (MODE == 5) { asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); }      // the oldest of three
        if (MODE == 1 || MODE == 2 || MODE >= 4) __builtin_amdgcn_s_barrier();
        if (MODE >= 2) {
            // next k-tile: 16 rows x 128 doubles per operand, one 1 KiB row per wavefront instruction (into a dummy stage)
            // (row stride 5120 doubles like the headline model; the walk wraps inside a 64 MB window)
            const double* g = gsrc + (long)(it & 511) * (16 * 5120);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + (4 * r) * 5120L),
                                                 (__attribute__((address_space(3))) void*)(Ad + (wave + 4 * r) * 144), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + 2560 + (4 * r) * 5120L),
                                                 (__attribute__((address_space(3))) void*)(Bd + (wave + 4 * r) * 144), 16, 0, 0);
            }
        }
    }
    if (MODE >= 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    double s = 0;
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// MODE 6: a FIFTH wavefront issues the whole DMA of a k-tile (32 loads) and nothing else; the four compute wavefronts
// only meet it at the barrier.  Tells whether the DMA's cost is its issue inside the MFMA streams or the LDS write port.
__global__ __launch_bounds__(320, 2) void kp(double* out, int iters, const double* src) {
    constexpr int LDT = 144;
    __shared__ double As[16 * LDT], Bs[16 * LDT];
    __shared__ double Ad[16 * 144], Bd[16 * 144];
    for (int e = threadIdx.x; e < 16 * LDT; e += 320) { As[e] = 1e-3 * (e % 7); Bs[e] = 1e-3 * (e % 5); }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave == 4) {
        const double* gsrc = src + (long)(blockIdx.x % 20) * 128 + 2 * lane;
        for (int it = 0; it < iters; ++it) {
            const double* g = gsrc + (long)(it & 511) * (16 * 5120);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + r * 5120L),
                                                 (__attribute__((address_space(3))) void*)(Ad + r * 144), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + 2560 + r * 5120L),
                                                 (__attribute__((address_space(3))) void*)(Bd + r * 144), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        return;
    }
    const int wm = wave >> 1, wn = wave & 1;
    const double* as = As + (lane >> 4) * LDT + wm * 64 + (lane & 15);
    const double* bs = Bs + (lane >> 4) * LDT + wn * 64 + (lane & 15);
    d4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = d4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            double af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = as[kk * 4 * LDT + i * 16];
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[j] = bs[kk * 4 * LDT + j * 16];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

void run_producer(const double* src) {
    const int blocks = 512, iters = 4000;
    double* out; (void)hipMalloc(&out, sizeof(double) * blocks * 256);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    kp<<<blocks, 320>>>(out, 50, src);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        kp<<<blocks, 320>>>(out, iters, src);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double flops = 2.0 * 16 * 16 * 4 * 16 * 4.0 * iters * blocks * 4;
    hipFuncAttributes fa; (void)hipFuncGetAttributes(&fa, (const void*)kp);
    printf("mode 6  producer wavefront + 4 compute wavefronts of 64 x 64, 2 workgroups/CU: %6.2f TFLOP/s   regs %d\n",
           flops / best / 1e9, fa.numRegs);
    (void)hipFree(out);
}

template <int NA, int NB, int WPS, int MODE = 0>
void run(const double* src = nullptr) {
    const int blocks = 256 * WPS, iters = 4000;
    double* out; (void)hipMalloc(&out, sizeof(double) * blocks * 256);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<NA, NB, WPS, MODE><<<blocks, 256>>>(out, 50, src);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        k<NA, NB, WPS, MODE><<<blocks, 256>>>(out, iters, src);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double flops = 2.0 * 16 * 16 * 4 * NA * NB * 4.0 * iters * blocks * 4;
    hipFuncAttributes fa; (void)hipFuncGetAttributes(&fa, (const void*)k<NA, NB, WPS, MODE>);
    printf("mode %d  wavefront tile %3d x %3d (%d x %d MFMA tiles), %d workgroup(s)/CU: %6.2f TFLOP/s   LDS reads / MFMA %.3f   regs %d   scratch %zu B\n",
           MODE, 16 * NA, 16 * NB, NA, NB, WPS, flops / best / 1e9, (double)(NA + NB) / (NA * NB), fa.numRegs, (size_t)fa.localSizeBytes);
    (void)hipFree(out);
}


// MODE 7: ONE workgroup of EIGHT wavefronts per CU on a 128 x 256 tile (2 x 4 wavefronts of 64 x 64; A tile shared by the
// four wavefront columns): the same 2 wavefronts per SIMD, but 48 KiB of DMA per k-tile for 8 x 64 MFMAs instead of 32 KiB for
// 4 x 64 -- 6 instead of 8 global_load_lds per wavefront and k-tile -- and a barrier of eight wavefronts.
__global__ __launch_bounds__(512, 1) void k8(double* out, int iters, const double* src) {
    constexpr int LDA = 144, LDB = 272;
    __shared__ double As[16 * LDA], Bs[16 * LDB];
    __shared__ double Ad[16 * LDA], Bd[16 * LDB];
    for (int e = threadIdx.x; e < 16 * LDA; e += 512) As[e] = 1e-3 * (e % 7);
    for (int e = threadIdx.x; e < 16 * LDB; e += 512) Bs[e] = 1e-3 * (e % 5);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wm = wave >> 2, wn = wave & 3;
    const double* as = As + (lane >> 4) * LDA + wm * 64 + (lane & 15);
    const double* bs = Bs + (lane >> 4) * LDB + wn * 64 + (lane & 15);
    const double* gsrc = src + (long)(blockIdx.x % 20) * 128 + 2 * lane;
    d4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = d4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            double af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = as[kk * 4 * LDA + i * 16];
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[j] = bs[kk * 4 * LDB + j * 16];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        asm volatile("" ::: "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const double* g = gsrc + (long)(it & 511) * (16 * 5120);
        // 48 rows of 1 KiB: A rows 0 .. 15, B rows as two halves of 1 KiB each; wavefront w takes rows w, w + 8, ..
#pragma unroll
        for (int r = 0; r < 2; ++r)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + (wave + 8 * r) * 5120L),
                                             (__attribute__((address_space(3))) void*)(Ad + (wave + 8 * r) * LDA), 16, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = (wave + 8 * r) >> 1, half = (wave + 8 * r) & 1;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + 2560 + half * 128 + row * 5120L),
                                             (__attribute__((address_space(3))) void*)(Bd + row * LDB + half * 128), 16, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    double s = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

void run_k8(const double* src) {
    const int blocks = 256, iters = 4000;
    double* out; (void)hipMalloc(&out, sizeof(double) * blocks * 512);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k8<<<blocks, 512>>>(out, 50, src);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        k8<<<blocks, 512>>>(out, iters, src);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double flops = 2.0 * 16 * 16 * 4 * 16 * 4.0 * iters * blocks * 8;
    hipFuncAttributes fa; (void)hipFuncGetAttributes(&fa, (const void*)k8);
    printf("mode 7  one workgroup of 8 wavefronts (128 x 256 tile) per CU, DMA + vmcnt(0) + barrier: %6.2f TFLOP/s   regs %d\n",
           flops / best / 1e9, fa.numRegs);
    (void)hipFree(out);
}

int main() {
    const size_t nsrc = (size_t)512 * 16 * 5120 + (1 << 20);
    double* src; (void)hipMalloc(&src, sizeof(double) * nsrc);
    (void)hipMemset(src, 0, sizeof(double) * nsrc);
    run<4, 4, 2, 1>(); run<4, 4, 2, 2>(src); run<4, 4, 2, 3>(src); run<4, 4, 2, 4>(src); run<4, 4, 2, 5>(src);
    run_producer(src);
    run_k8(src);
    run<6, 4, 2, 1>(); run<6, 4, 2, 2>(src);
    run<4, 4, 2>(); run<4, 4, 1>();
    run<6, 4, 2>(); run<4, 6, 2>(); run<6, 4, 1>();
    run<5, 4, 2>(); run<5, 5, 2>();
    run<8, 4, 1>(); run<6, 6, 1>(); run<8, 6, 1>();
    run<2, 2, 2>(); run<3, 3, 2>();
    return 0;
}
