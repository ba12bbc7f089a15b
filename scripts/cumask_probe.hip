// Which physical CUs does a CU-masked stream use?  hipcc --offload-arch=gfx950 -O2 scripts/cumask_probe.hip -o /tmp/cumask_probe
// Each workgroup records (XCC id, SE, SH, CU) of where it ran; the host prints the set per mask.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <set>
#include <map>

__global__ void where_kernel(unsigned* out, int spin) {
    __shared__ double pad[8192];   // 64 KB: two workgroups per CU at most
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    pad[threadIdx.x] = hw;
    long t0 = clock64();
    while (clock64() - t0 < spin) {}
    if (threadIdx.x == 0) out[blockIdx.x] = (hw & 0xffff) | ((xcc & 0xf) << 16) | (pad[1] > 1e300 ? 1u << 31 : 0);
}

static void run(const char* name, hipStream_t s, unsigned* d, int n) {
    std::vector<unsigned> h(n);
    hipLaunchKernelGGL(where_kernel, dim3(n), dim3(256), 0, s, d, 40000);
    hipStreamSynchronize(s);
    hipMemcpy(h.data(), d, n * sizeof(unsigned), hipMemcpyDeviceToHost);
    std::map<unsigned, std::set<unsigned>> per_xcc;
    for (unsigned v : h) per_xcc[(v >> 16) & 0xf].insert(v & 0xff00);   // cu_id[11:8], sh[12], se[15:13]
    size_t tot = 0;
    printf("%-28s", name);
    for (auto& kv : per_xcc) { printf(" xcc%u:%zu", kv.first, kv.second.size()); tot += kv.second.size(); }
    printf("  total distinct CUs %zu\n", tot);
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int ncu = p.multiProcessorCount, n = 4096;
    unsigned* d; hipMalloc(&d, n * sizeof(unsigned));
    hipStream_t s0; hipStreamCreate(&s0);
    run("no mask", s0, d, n);
    for (int res : {8, 16, 32}) {
        std::vector<uint32_t> m((ncu + 31) / 32, 0u);
        for (int c = res; c < ncu; ++c) m[c / 32] |= 1u << (c % 32);
        hipStream_t s; hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)m.size(), m.data());
        char nm[64]; snprintf(nm, 64, "exclude first %d bits (%s)", res, hipGetErrorString(e));
        if (e == hipSuccess) run(nm, s, d, n);
        std::vector<uint32_t> m2((ncu + 31) / 32, 0u);
        for (int c = 0; c < res; ++c) m2[c / 32] |= 1u << (c % 32);
        hipStream_t s2; e = hipExtStreamCreateWithCUMask(&s2, (uint32_t)m2.size(), m2.data());
        snprintf(nm, 64, "ONLY first %d bits (%s)", res, hipGetErrorString(e));
        if (e == hipSuccess) run(nm, s2, d, n);
    }
    // a single-workgroup kernel: where does it land, repeatedly?
    std::vector<unsigned> h(1);
    printf("single-WG launches land on:");
    for (int i = 0; i < 12; ++i) {
        hipLaunchKernelGGL(where_kernel, dim3(1), dim3(256), 0, s0, d, 100);
        hipStreamSynchronize(s0);
        hipMemcpy(h.data(), d, sizeof(unsigned), hipMemcpyDeviceToHost);
        printf(" x%u/%04x", (h[0] >> 16) & 0xf, h[0] & 0xff00);
    }
    printf("\n");
    return 0;
}
