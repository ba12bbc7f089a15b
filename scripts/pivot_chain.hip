// Microbenchmark of the 16-pivot chain of the diagonal-block kernel (sr_factor16_aug of sr_factor.hip) on ONE wavefront:
// cycles per call (s_memtime) and the result against a plain fp64 Cholesky of the same tile, so that variants of the chain
// can be compared without the rest of the kernel.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I safe_exploration_amd/csrc scripts/pivot_chain.hip -o scripts/_bin/pivot
#include "../safe_exploration_amd/csrc/sr_factor.hip"
#include <cstdio>
#include <cmath>
#include <vector>
#include <cstdarg>

void sr_set_error(const char*, ...) {}

__global__ __launch_bounds__(1024, 1) void pivot_bench(const double* A16, double* out, long long* ticks, int reps) {
    __shared__ double S[SR_NB * SR_PD_LD];
    __shared__ double X[16 * SR_PD_TLD];
    __shared__ int fail;
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid == 0) fail = 0;
    for (int i = tid; i < 256; i += 1024) S[(i >> 4) * SR_PD_LD + (i & 15)] = A16[i];
    __syncthreads();
    if (tid < 64) {
        sr_factor16_aug(S, SR_PD_LD, 0, X, &fail, lane);
        __builtin_amdgcn_s_waitcnt(0);
        const long long t0 = __builtin_readcyclecounter();
        for (int r = 0; r < reps; ++r) {
            sr_factor16_aug(S, SR_PD_LD, 0, X, &fail, lane);
            __builtin_amdgcn_s_waitcnt(0);
        }
        const long long t1 = __builtin_readcyclecounter();
        if (lane == 0) { ticks[0] = t1 - t0; ticks[1] = fail; }
        for (int i = lane; i < 16 * 32; i += 64) out[i] = X[(i >> 5) * SR_PD_TLD + (i & 31)];
    }
}

int main() {
    std::vector<double> A(256), M(16 * 24);
    unsigned s = 12345u;
    for (auto& v : M) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.0 - 0.5; }
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            double t = (i == j) ? 0.5 : 0.0;
            for (int k = 0; k < 24; ++k) t += M[i * 24 + k] * M[j * 24 + k];
            A[i * 16 + j] = t;
        }
    std::vector<double> U(A);                       // plain upper Cholesky, row by row
    for (int j = 0; j < 16; ++j) {
        const double d = std::sqrt(U[j * 16 + j]);
        for (int c = j; c < 16; ++c) U[j * 16 + c] = (c == j) ? d : U[j * 16 + c] / d;
        for (int r = j + 1; r < 16; ++r)
            for (int c = r; c < 16; ++c) U[r * 16 + c] -= U[j * 16 + r] * U[j * 16 + c];
    }
    double *dA, *dO; long long* dT;
    hipMalloc(&dA, 256 * 8); hipMalloc(&dO, 512 * 8); hipMalloc(&dT, 16);
    hipMemcpy(dA, A.data(), 256 * 8, hipMemcpyHostToDevice);
    const int reps = 200;
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(pivot_bench, dim3(1), dim3(1024), 0, 0, dA, dO, dT, reps);
    hipDeviceSynchronize();
    std::vector<double> O(512); long long T[2];
    hipMemcpy(O.data(), dO, 512 * 8, hipMemcpyDeviceToHost); hipMemcpy(T, dT, 16, hipMemcpyDeviceToHost);
    double eu = 0, et = 0;
    for (int i = 0; i < 16; ++i)
        for (int j = i; j < 16; ++j) eu = std::fmax(eu, std::fabs(O[i * 32 + j] - U[i * 16 + j]));
    for (int i = 0; i < 16; ++i)                    // T = U^-T:  T U^T = I
        for (int j = 0; j < 16; ++j) {
            double t = 0;
            for (int k = j; k < 16; ++k) t += O[i * 32 + 16 + k] * U[j * 16 + k];
            et = std::fmax(et, std::fabs(t - (i == j ? 1.0 : 0.0)));
        }
    printf("sr_factor16_aug: %.0f cycles per 16 pivots (%.1f per pivot)   max|U - chol| = %.2e   max|T U^T - I| = %.2e   fail=%lld\n",
           (double)T[0] / reps, (double)T[0] / reps / 16, eu, et, T[1]);
    return 0;
}
