// What does it cost to hand work from one hardware queue to another?  (round 6: can the diagonal-block kernel of the
// Cholesky run BESIDE the rest of its block row?)
//   hipcc --offload-arch=gfx950 -O3 scripts/xqueue_handoff.hip -o scripts/_bin/xq && ./scripts/_bin/xq
// Ping-pong of K hops between two streams, everything enqueued up front, `work` us of spinning per hop:
//   event : hipEventRecord on one stream, hipStreamWaitEvent on the other            (what round 2 measured as too slow)
//   flagk : a one-thread SIGNAL kernel stores a counter, a one-thread GATE kernel on the other stream polls it
//   flagi : the worker kernel itself polls in its prologue and stores in its epilogue (release fence before the store)
// and one stream alone (`same`) as the floor; then: does hipExtAnyOrderLaunch let two kernels of ONE stream overlap?
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <chrono>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ void spin_us(double us) {
    const long long t0 = wall_clock64();
    while ((double)(wall_clock64() - t0) < us * 100.0) __builtin_amdgcn_s_sleep(2);
}
__global__ void work_kernel(double us, long long* stamp, int slot) {
    if (threadIdx.x == 0 && stamp) stamp[2 * slot] = wall_clock64();
    spin_us(us);
    if (threadIdx.x == 0 && stamp) stamp[2 * slot + 1] = wall_clock64();
}
__global__ void signal_kernel(unsigned* flag, unsigned v) {
    __hip_atomic_store(flag, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void gate_kernel(const unsigned* flag, unsigned v) {
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < v) __builtin_amdgcn_s_sleep(1);
}
// worker with the gate in its prologue and the signal in its epilogue; `payload` doubles are written before the signal
// and checked by the consumer after its acquire (stale reads counted in *bad)
__global__ void work_flag_kernel(double us, const unsigned* wait_flag, unsigned wait_v, unsigned* set_flag, unsigned set_v,
                                 double* payload_out, const double* payload_in, int n, unsigned* bad, long long* stamp,
                                 int slot) {
    if (wait_flag) {
        if (threadIdx.x == 0)
            while (__hip_atomic_load(wait_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < wait_v) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __syncthreads();
    }
    if (threadIdx.x == 0 && stamp) stamp[2 * slot] = wall_clock64();
    if (payload_in) {
        unsigned b = 0;
        for (int i = threadIdx.x; i < n; i += blockDim.x) b += payload_in[i] != (double)wait_v;
        if (b) atomicAdd(bad, b);
    }
    spin_us(us);
    if (payload_out)
        for (int i = threadIdx.x; i < n; i += blockDim.x) payload_out[i] = (double)set_v;
    if (threadIdx.x == 0 && stamp) stamp[2 * slot + 1] = wall_clock64();
    if (set_flag) {
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(set_flag, set_v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// L2-resident reader: every workgroup sweeps ITS 64 KB slice `iters` times (the slices of a launch fit the L2s: 1024 x 64 KB
// = 64 MB is too much -- the launch uses 256 workgroups x 64 KB = 16 MB over 8 x 4 MB of L2, i.e. 2 MB per XCD).
__global__ void l2_reader_kernel(const double* __restrict__ src, int iters, double* sink) {
    const double* p = src + (size_t)blockIdx.x * 8192;
    double acc = 0.0;
    for (int it = 0; it < iters; ++it)
        for (int i = threadIdx.x; i < 8192; i += 256) acc += p[i];
    if (acc == 12345.678) sink[0] = acc;
}
__global__ void empty_kernel() {}
__global__ void writer_kernel(double* dst, int n, double v) {     // n doubles per workgroup
    double* p = dst + (size_t)blockIdx.x * 32768;
    for (int i = threadIdx.x; i < n; i += 256) p[i] = v;
}
__global__ void slow_gate_kernel(const unsigned* flag, unsigned v) {
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < v) { __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127); }
}
__global__ void fma_kernel(int iters, double* sink) {
    double a = threadIdx.x * 1e-3, b = 1.0000001, c = 1e-9;
    for (int i = 0; i < iters; ++i) { a = fma(a, b, c); b = fma(b, 0.9999999, c); }
    if (a + b == 12345.678) sink[0] = a;
}

static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
    hipStream_t sa, sb;
    int lo = 0, hi = 0;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CK(hipStreamCreateWithPriority(&sa, hipStreamNonBlocking, hi));
    CK(hipStreamCreateWithPriority(&sb, hipStreamNonBlocking, hi));
    unsigned *fa, *fb, *bad;
    double* pay;
    const int NP = 128 * 128;                       // one diagonal block
    CK(hipMalloc(&fa, 256)); CK(hipMalloc(&fb, 256)); CK(hipMalloc(&bad, 256)); CK(hipMalloc(&pay, 2 * NP * sizeof(double)));
    CK(hipMemset(bad, 0, 4));
    const int K = 200;
    unsigned* sig = nullptr;                        // signal memory for hipStreamWaitValue32 (two words, 8 bytes apart)
    {
        int can = 0;
        (void)hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
        if (!can || hipExtMallocWithFlags((void**)&sig, 16, hipMallocSignalMemory) != hipSuccess) { (void)hipGetLastError(); sig = nullptr; }
        printf("hipStreamWaitValue32: %s\n", sig ? "available" : "not available on this device / runtime");
    }
    std::vector<hipEvent_t> ev(2 * K);
    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (double work : {2.0, 10.0, 25.0}) {
        double t[5] = {0, 0, 0, 0, 0};
        for (int mode = 0; mode < 5; ++mode) {
            if (mode == 4 && !sig) continue;
            std::vector<double> reps;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipMemset(fa, 0, 4)); CK(hipMemset(fb, 0, 4));
                if (sig) CK(hipMemset(sig, 0, 16));
                CK(hipDeviceSynchronize());
                const double t0 = now_us();
                for (int i = 0; i < K; ++i) {
                    if (mode == 0) {                // one stream
                        hipLaunchKernelGGL(work_kernel, dim3(1), dim3(256), 0, sa, work, nullptr, 0);
                        hipLaunchKernelGGL(work_kernel, dim3(1), dim3(256), 0, sa, work, nullptr, 0);
                    } else if (mode == 1) {         // events
                        if (i > 0) CK(hipStreamWaitEvent(sa, ev[2 * i - 1], 0));
                        hipLaunchKernelGGL(work_kernel, dim3(1), dim3(256), 0, sa, work, nullptr, 0);
                        CK(hipEventRecord(ev[2 * i], sa));
                        CK(hipStreamWaitEvent(sb, ev[2 * i], 0));
                        hipLaunchKernelGGL(work_kernel, dim3(1), dim3(256), 0, sb, work, nullptr, 0);
                        CK(hipEventRecord(ev[2 * i + 1], sb));
                    } else if (mode == 2) {         // signal / gate kernels
                        if (i > 0) hipLaunchKernelGGL(gate_kernel, dim3(1), dim3(1), 0, sa, fb, (unsigned)i);
                        hipLaunchKernelGGL(work_kernel, dim3(1), dim3(256), 0, sa, work, nullptr, 0);
                        hipLaunchKernelGGL(signal_kernel, dim3(1), dim3(1), 0, sa, fa, (unsigned)(i + 1));
                        hipLaunchKernelGGL(gate_kernel, dim3(1), dim3(1), 0, sb, fa, (unsigned)(i + 1));
                        hipLaunchKernelGGL(work_kernel, dim3(1), dim3(256), 0, sb, work, nullptr, 0);
                        hipLaunchKernelGGL(signal_kernel, dim3(1), dim3(1), 0, sb, fb, (unsigned)(i + 1));
                    } else if (mode == 4) {         // stream memory operations: the command processor writes / waits, no kernel
                        if (i > 0) CK(hipStreamWaitValue32(sa, sig + 2, (unsigned)i, hipStreamWaitValueGte, 0xFFFFFFFFu));
                        hipLaunchKernelGGL(work_kernel, dim3(1), dim3(256), 0, sa, work, nullptr, 0);
                        CK(hipStreamWriteValue32(sa, sig, (unsigned)(i + 1), 0));
                        CK(hipStreamWaitValue32(sb, sig, (unsigned)(i + 1), hipStreamWaitValueGte, 0xFFFFFFFFu));
                        hipLaunchKernelGGL(work_kernel, dim3(1), dim3(256), 0, sb, work, nullptr, 0);
                        CK(hipStreamWriteValue32(sb, sig + 2, (unsigned)(i + 1), 0));
                    } else {                        // gate and signal inside the workers, a 128 KB payload handed over
                        hipLaunchKernelGGL(work_flag_kernel, dim3(1), dim3(256), 0, sa, work, i > 0 ? fb : nullptr, (unsigned)i, fa,
                                           (unsigned)(i + 1), pay, i > 0 ? pay + NP : nullptr, NP, bad, nullptr, 0);
                        hipLaunchKernelGGL(work_flag_kernel, dim3(1), dim3(256), 0, sb, work, fa, (unsigned)(i + 1), fb,
                                           (unsigned)(i + 1), pay + NP, pay, NP, bad, nullptr, 0);
                    }
                }
                CK(hipDeviceSynchronize());
                reps.push_back((now_us() - t0) / (2.0 * K));
            }
            std::sort(reps.begin(), reps.end());
            t[mode] = reps[2];
        }
        unsigned hbad = 0;
        CK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
        printf("work %5.1f us per hop | per hop: same stream %6.2f | events %6.2f | signal+gate kernels %6.2f | in-kernel flags %6.2f | stream write/wait value %6.2f us"
               "  (stale payload words %u)\n", work, t[0], t[1], t[2], t[3], t[4], hbad);
    }
    // ---- does a POLLING one-thread kernel on another queue slow a kernel down?  (20 launches of a fixed 20 us of fp64 FMAs)
    {
        double* sink2; CK(hipMalloc(&sink2, 64));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipStream_t sc; CK(hipStreamCreateWithPriority(&sc, hipStreamNonBlocking, hi));
        for (int spin : {0, 1, 2}) {               // nothing beside it / gate_kernel (s_sleep 1) / slow poller (s_sleep 127)
            CK(hipMemset(fa, 0, 4));
            CK(hipDeviceSynchronize());
            if (spin == 1) hipLaunchKernelGGL(gate_kernel, dim3(1), dim3(1), 0, sb, fa, 1u);
            if (spin == 2) hipLaunchKernelGGL(slow_gate_kernel, dim3(1), dim3(1), 0, sb, fa, 1u);
            CK(hipEventRecord(e0, sa));
            for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(fma_kernel, dim3(16), dim3(256), 0, sa, 6000, sink2);
            CK(hipEventRecord(e1, sa));
            CK(hipStreamSynchronize(sa));
            hipLaunchKernelGGL(signal_kernel, dim3(1), dim3(1), 0, sc, fa, 1u);
            CK(hipDeviceSynchronize());
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("20 launches of a fixed-work FMA kernel (16 WGs): %.1f us each with %s\n", ms * 1e3 / 20,
                   spin == 0 ? "nothing beside it" : (spin == 1 ? "a polling gate kernel (s_sleep 1) on another queue" : "a slow poller (s_sleep 127) on another queue"));
        }
    }
    // ---- does ANY resident kernel on another queue slow the BOUNDARIES of kernels that write memory?  (a kernel's end carries a
    // release: its stores written back from the L2s)  40 launches of a kernel that writes `kb` KB per workgroup, 64 workgroups
    {
        double* dst; CK(hipMalloc(&dst, 64 * 32768 * sizeof(double)));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipStream_t sc; CK(hipStreamCreateWithPriority(&sc, hipStreamNonBlocking, hi));
        for (int kb : {0, 16, 256}) {
            for (int beside : {0, 1, 2}) {         // nothing / an ALU-only resident kernel (one workgroup) / the polling gate kernel
                CK(hipMemset(fa, 0, 4));
                CK(hipDeviceSynchronize());
                if (beside == 1) hipLaunchKernelGGL(work_kernel, dim3(1), dim3(64), 0, sb, 3000.0, nullptr, 0);
                if (beside == 2) hipLaunchKernelGGL(gate_kernel, dim3(1), dim3(1), 0, sb, fa, 1u);
                CK(hipEventRecord(e0, sa));
                for (int i = 0; i < 40; ++i) hipLaunchKernelGGL(writer_kernel, dim3(64), dim3(256), 0, sa, dst, kb * 128, (double)i);
                CK(hipEventRecord(e1, sa));
                CK(hipStreamSynchronize(sa));
                hipLaunchKernelGGL(signal_kernel, dim3(1), dim3(1), 0, sc, fa, 1u);
                CK(hipDeviceSynchronize());
                float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
                printf("40 launches of a kernel writing %3d KB per workgroup (64 WGs): %6.1f us each with %s\n", kb, ms * 1e3 / 40,
                       beside == 0 ? "nothing beside it" : (beside == 1 ? "a resident ALU-only kernel on another queue" : "the polling gate kernel on another queue"));
            }
        }
    }
    // ---- what do kernel boundaries on ANOTHER queue cost a kernel that lives from its L2?  (every kernel start invalidates the
    // L2s, every kernel end writes them back: the caches are per XCD and not coherent with each other)
    {
        double *src, *sink;
        CK(hipMalloc(&src, 256 * 8192 * sizeof(double))); CK(hipMalloc(&sink, 64));
        CK(hipMemset(src, 0, 256 * 8192 * sizeof(double)));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int iters = 400;
        for (int rate : {0, 1, 4}) {          // empty kernels on the other stream per reader launch: none, back to back (1 queue, 4 launches deep)
            float ms_best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0, sa));
                hipLaunchKernelGGL(l2_reader_kernel, dim3(256), dim3(256), 0, sa, src, iters, sink);
                CK(hipEventRecord(e1, sa));
                if (rate) for (int i = 0; i < 150 * rate; ++i) hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, sb);
                CK(hipDeviceSynchronize());
                float ms = 0;
                CK(hipEventElapsedTime(&ms, e0, e1));
                ms_best = ms < ms_best ? ms : ms_best;
            }
            printf("L2-resident reader (256 WGs x 64 KB x %d sweeps = %.1f GB): %.1f us = %.2f TB/s with %s\n", iters,
                   256.0 * 65536 * iters / 1e9, ms_best * 1e3, 256.0 * 65536 * iters / (ms_best * 1e-3) / 1e12,
                   rate == 0 ? "nothing beside it" : (rate == 1 ? "150 empty kernels on another stream" : "600 empty kernels on another stream"));
        }
    }
    // ---- hipExtAnyOrderLaunch on one stream: does the second kernel start before the first one ends?
    long long* stamp;
    CK(hipMalloc(&stamp, 64 * sizeof(long long)));
    for (int flags : {0, (int)hipExtAnyOrderLaunch}) {
        CK(hipMemset(stamp, 0, 64 * sizeof(long long)));
        hipLaunchKernelGGL(work_kernel, dim3(1), dim3(256), 0, sa, 1.0, nullptr, 0);      // warm
        CK(hipDeviceSynchronize());
        hipLaunchKernelGGL(work_kernel, dim3(1), dim3(256), 0, sa, 100.0, stamp, 0);
        hipExtLaunchKernelGGL(work_kernel, dim3(1), dim3(256), 0, sa, nullptr, nullptr, flags, 5.0, stamp, 1);
        CK(hipDeviceSynchronize());
        long long h[4];
        CK(hipMemcpy(h, stamp, sizeof(h), hipMemcpyDeviceToHost));
        printf("one stream, second launch with flags=%d: first runs [0, %.1f] us, second starts at %.1f us -> %s\n", flags,
               (h[1] - h[0]) / 100.0, (h[2] - h[0]) / 100.0, h[2] < h[1] ? "OVERLAP" : "in order");
    }
    return 0;
}
