// Microbenchmark: do fp64 MFMA and fp64 VALU FMA overlap on gfx950?  Each workgroup has 8 wavefronts:
// wavefronts 0-3 issue v_mfma_f64_16x16x4_f64, wavefronts 4-7 issue v_fma_f64 chains.  mode 1: MFMA only,
// mode 2: VALU only, mode 3: both.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CLOB "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127"
__global__ __launch_bounds__(512) void k(double* out, int iters, int mode, double a0, double b0) {
    const int wave = threadIdx.x >> 6;
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    if (wave < 4) {
        if (!(mode & 1)) return;
        asm volatile("v_accvgpr_write_b32 a0, 0\n\tv_accvgpr_write_b32 a1, 0\n\tv_accvgpr_write_b32 a2, 0\n\tv_accvgpr_write_b32 a3, 0\n\tv_accvgpr_write_b32 a4, 0\n\tv_accvgpr_write_b32 a5, 0\n\tv_accvgpr_write_b32 a6, 0\n\tv_accvgpr_write_b32 a7, 0\n\tv_accvgpr_write_b32 a8, 0\n\tv_accvgpr_write_b32 a9, 0\n\tv_accvgpr_write_b32 a10, 0\n\tv_accvgpr_write_b32 a11, 0\n\tv_accvgpr_write_b32 a12, 0\n\tv_accvgpr_write_b32 a13, 0\n\tv_accvgpr_write_b32 a14, 0\n\tv_accvgpr_write_b32 a15, 0\n\tv_accvgpr_write_b32 a16, 0\n\tv_accvgpr_write_b32 a17, 0\n\tv_accvgpr_write_b32 a18, 0\n\tv_accvgpr_write_b32 a19, 0\n\tv_accvgpr_write_b32 a20, 0\n\tv_accvgpr_write_b32 a21, 0\n\tv_accvgpr_write_b32 a22, 0\n\tv_accvgpr_write_b32 a23, 0\n\tv_accvgpr_write_b32 a24, 0\n\tv_accvgpr_write_b32 a25, 0\n\tv_accvgpr_write_b32 a26, 0\n\tv_accvgpr_write_b32 a27, 0\n\tv_accvgpr_write_b32 a28, 0\n\tv_accvgpr_write_b32 a29, 0\n\tv_accvgpr_write_b32 a30, 0\n\tv_accvgpr_write_b32 a31, 0\n\tv_accvgpr_write_b32 a32, 0\n\tv_accvgpr_write_b32 a33, 0\n\tv_accvgpr_write_b32 a34, 0\n\tv_accvgpr_write_b32 a35, 0\n\tv_accvgpr_write_b32 a36, 0\n\tv_accvgpr_write_b32 a37, 0\n\tv_accvgpr_write_b32 a38, 0\n\tv_accvgpr_write_b32 a39, 0\n\tv_accvgpr_write_b32 a40, 0\n\tv_accvgpr_write_b32 a41, 0\n\tv_accvgpr_write_b32 a42, 0\n\tv_accvgpr_write_b32 a43, 0\n\tv_accvgpr_write_b32 a44, 0\n\tv_accvgpr_write_b32 a45, 0\n\tv_accvgpr_write_b32 a46, 0\n\tv_accvgpr_write_b32 a47, 0\n\tv_accvgpr_write_b32 a48, 0\n\tv_accvgpr_write_b32 a49, 0\n\tv_accvgpr_write_b32 a50, 0\n\tv_accvgpr_write_b32 a51, 0\n\tv_accvgpr_write_b32 a52, 0\n\tv_accvgpr_write_b32 a53, 0\n\tv_accvgpr_write_b32 a54, 0\n\tv_accvgpr_write_b32 a55, 0\n\tv_accvgpr_write_b32 a56, 0\n\tv_accvgpr_write_b32 a57, 0\n\tv_accvgpr_write_b32 a58, 0\n\tv_accvgpr_write_b32 a59, 0\n\tv_accvgpr_write_b32 a60, 0\n\tv_accvgpr_write_b32 a61, 0\n\tv_accvgpr_write_b32 a62, 0\n\tv_accvgpr_write_b32 a63, 0\n\tv_accvgpr_write_b32 a64, 0\n\tv_accvgpr_write_b32 a65, 0\n\tv_accvgpr_write_b32 a66, 0\n\tv_accvgpr_write_b32 a67, 0\n\tv_accvgpr_write_b32 a68, 0\n\tv_accvgpr_write_b32 a69, 0\n\tv_accvgpr_write_b32 a70, 0\n\tv_accvgpr_write_b32 a71, 0\n\tv_accvgpr_write_b32 a72, 0\n\tv_accvgpr_write_b32 a73, 0\n\tv_accvgpr_write_b32 a74, 0\n\tv_accvgpr_write_b32 a75, 0\n\tv_accvgpr_write_b32 a76, 0\n\tv_accvgpr_write_b32 a77, 0\n\tv_accvgpr_write_b32 a78, 0\n\tv_accvgpr_write_b32 a79, 0\n\tv_accvgpr_write_b32 a80, 0\n\tv_accvgpr_write_b32 a81, 0\n\tv_accvgpr_write_b32 a82, 0\n\tv_accvgpr_write_b32 a83, 0\n\tv_accvgpr_write_b32 a84, 0\n\tv_accvgpr_write_b32 a85, 0\n\tv_accvgpr_write_b32 a86, 0\n\tv_accvgpr_write_b32 a87, 0\n\tv_accvgpr_write_b32 a88, 0\n\tv_accvgpr_write_b32 a89, 0\n\tv_accvgpr_write_b32 a90, 0\n\tv_accvgpr_write_b32 a91, 0\n\tv_accvgpr_write_b32 a92, 0\n\tv_accvgpr_write_b32 a93, 0\n\tv_accvgpr_write_b32 a94, 0\n\tv_accvgpr_write_b32 a95, 0\n\tv_accvgpr_write_b32 a96, 0\n\tv_accvgpr_write_b32 a97, 0\n\tv_accvgpr_write_b32 a98, 0\n\tv_accvgpr_write_b32 a99, 0\n\tv_accvgpr_write_b32 a100, 0\n\tv_accvgpr_write_b32 a101, 0\n\tv_accvgpr_write_b32 a102, 0\n\tv_accvgpr_write_b32 a103, 0\n\tv_accvgpr_write_b32 a104, 0\n\tv_accvgpr_write_b32 a105, 0\n\tv_accvgpr_write_b32 a106, 0\n\tv_accvgpr_write_b32 a107, 0\n\tv_accvgpr_write_b32 a108, 0\n\tv_accvgpr_write_b32 a109, 0\n\tv_accvgpr_write_b32 a110, 0\n\tv_accvgpr_write_b32 a111, 0\n\tv_accvgpr_write_b32 a112, 0\n\tv_accvgpr_write_b32 a113, 0\n\tv_accvgpr_write_b32 a114, 0\n\tv_accvgpr_write_b32 a115, 0\n\tv_accvgpr_write_b32 a116, 0\n\tv_accvgpr_write_b32 a117, 0\n\tv_accvgpr_write_b32 a118, 0\n\tv_accvgpr_write_b32 a119, 0\n\tv_accvgpr_write_b32 a120, 0\n\tv_accvgpr_write_b32 a121, 0\n\tv_accvgpr_write_b32 a122, 0\n\tv_accvgpr_write_b32 a123, 0\n\tv_accvgpr_write_b32 a124, 0\n\tv_accvgpr_write_b32 a125, 0\n\tv_accvgpr_write_b32 a126, 0\n\tv_accvgpr_write_b32 a127, 0" ::: CLOB);
        for (int it = 0; it < iters; ++it) asm volatile("v_mfma_f64_16x16x4_f64 a[0:7], %0, %1, a[0:7]\n\tv_mfma_f64_16x16x4_f64 a[8:15], %0, %1, a[8:15]\n\tv_mfma_f64_16x16x4_f64 a[16:23], %0, %1, a[16:23]\n\tv_mfma_f64_16x16x4_f64 a[24:31], %0, %1, a[24:31]\n\tv_mfma_f64_16x16x4_f64 a[32:39], %0, %1, a[32:39]\n\tv_mfma_f64_16x16x4_f64 a[40:47], %0, %1, a[40:47]\n\tv_mfma_f64_16x16x4_f64 a[48:55], %0, %1, a[48:55]\n\tv_mfma_f64_16x16x4_f64 a[56:63], %0, %1, a[56:63]\n\tv_mfma_f64_16x16x4_f64 a[64:71], %0, %1, a[64:71]\n\tv_mfma_f64_16x16x4_f64 a[72:79], %0, %1, a[72:79]\n\tv_mfma_f64_16x16x4_f64 a[80:87], %0, %1, a[80:87]\n\tv_mfma_f64_16x16x4_f64 a[88:95], %0, %1, a[88:95]\n\tv_mfma_f64_16x16x4_f64 a[96:103], %0, %1, a[96:103]\n\tv_mfma_f64_16x16x4_f64 a[104:111], %0, %1, a[104:111]\n\tv_mfma_f64_16x16x4_f64 a[112:119], %0, %1, a[112:119]\n\tv_mfma_f64_16x16x4_f64 a[120:127], %0, %1, a[120:127]" :: "v"(a), "v"(b) : CLOB);
        double s; asm volatile("v_accvgpr_read_b32 %0, a0" : "=v"(((int*)&s)[0]) :: CLOB);
        out[blockIdx.x * 512 + threadIdx.x] = s;
    } else {
        if (!(mode & 2)) return;
        double x[16];
        for (int i = 0; i < 16; ++i) x[i] = a + i;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) x[i] = __builtin_fma(x[i], b, a);   // 64 FMAs per iteration
        }
        double s = 0; for (int i = 0; i < 16; ++i) s += x[i];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    }
}
int main() {
    int blocks = 256, iters = 20000;
    double* out; (void)hipMalloc(&out, sizeof(double) * blocks * 512);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int mode = 1; mode <= 3; ++mode) {
        k<<<blocks, 512>>>(out, 100, mode, 1.0, 0.999);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        k<<<blocks, 512>>>(out, iters, mode, 1.0, 0.999);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        double waves = blocks * 4.0;
        double fm = (mode & 1) ? 2.0 * 16 * 16 * 4 * 16.0 * iters * waves : 0;
        double fv = (mode & 2) ? 2.0 * 64 * 64.0 * iters * waves : 0;
        printf("mode %d: %.3f ms  MFMA %.1f TF  VALU %.1f TF  total %.1f TF\n", mode, ms, fm / ms / 1e9, fv / ms / 1e9,
               (fm + fv) / ms / 1e9);
    }
    return 0;
}
