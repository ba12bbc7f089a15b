// Microbenchmark: sustained rate of v_mfma_f64_16x16x4_f64 on gfx950 (the ceiling the variance
// kernel is priced against).  hipcc --offload-arch=gfx950 -O3 scripts/mfma_f64_peak.hip -o /tmp/peak && /tmp/peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(double* out, int iters, double a0, double b0) {
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks_per_cu) {
    int blocks = 256 * blocks_per_cu, iters = 20000;
    double* out; hipMalloc(&out, sizeof(double) * blocks * 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<blocks, 256>>>(out, 100, 1.0, 1e-3);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC><<<blocks, 256>>>(out, iters, 1.0, 1e-3);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = 2.0 * 16 * 16 * 4 * (double)NACC * iters * 4.0 * blocks;
    printf("NACC=%d blocks/CU=%d  %.3f ms  %.2f TFLOP/s  (%.1f cycles/MFMA/SIMD at 2.4 GHz)\n", NACC,
           blocks_per_cu, ms, flops / ms / 1e9, ms * 1e-3 * 2.4e9 / ((double)NACC * iters * blocks_per_cu));
    hipFree(out);
}
int main() {
    run<4>(1); run<8>(1); run<16>(1); run<8>(2); run<16>(2);
    return 0;
}
