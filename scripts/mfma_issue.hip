// Microbenchmark: fp64 MFMA 16x16x4 issue rate against the number of wavefronts per SIMD and of independent accumulators
// per wavefront (no memory traffic at all).  One workgroup per CU, NW wavefronts each; every wavefront repeats
//   for i < NACC: acc[i] = mfma(a, b, acc[i])
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma_issue.hip -o scripts/_bin/mfma_issue && scripts/_bin/mfma_issue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC, int NT, int BAR = 0>
__global__ __launch_bounds__(NT) void kissue(double* out, int iters, double a0, double b0) {
    d4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
    const double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 32 / NACC; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        if (BAR && (it % BAR) == BAR - 1) __syncthreads();       // a workgroup barrier every 32 BAR MFMAs of a wavefront
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[(long)blockIdx.x * NT + threadIdx.x] = s;
}

template <int NACC, int NT, int BAR = 0>
void run(double* out, int ncu) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((kissue<NACC, NT, BAR>), dim3(ncu), dim3(NT), 0, 0, out, 10, 1.0, 1.0);
    hipEventRecord(e0);
    hipLaunchKernelGGL((kissue<NACC, NT, BAR>), dim3(ncu), dim3(NT), 0, 0, out, iters, 1.0, 1.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)ncu * (NT / 64) * iters * 32.0 * 2048.0;
    printf("%2d wavefronts per SIMD, %2d accumulators per wavefront, barrier every %3d MFMAs: %6.2f TFLOP/s (%.0f %% of 78.6)\n", NT / 256, NACC, 32 * BAR,
           flops / (ms * 1e-3) / 1e12, 100.0 * flops / (ms * 1e-3) / 78.6e12);
}

int main() {
    int ncu = 0;
    hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    double* out; hipMalloc(&out, (size_t)ncu * 1024 * 8);
    run<1, 256>(out, ncu); run<2, 256>(out, ncu); run<4, 256>(out, ncu); run<8, 256>(out, ncu); run<16, 256>(out, ncu);
    run<1, 512>(out, ncu); run<2, 512>(out, ncu); run<4, 512>(out, ncu); run<16, 512>(out, ncu);
    run<1, 1024>(out, ncu); run<2, 1024>(out, ncu); run<4, 1024>(out, ncu); run<8, 1024>(out, ncu); run<16, 1024>(out, ncu);
    run<4, 1024, 1>(out, ncu); run<4, 1024, 2>(out, ncu); run<4, 1024, 4>(out, ncu); run<4, 512, 2>(out, ncu); run<4, 256, 2>(out, ncu);
    return 0;
}
