// Micro-benchmark: does a LOW-priority HIP stream stay out of the way of a latency-critical chain?
// Chain: n dependent kernels on stream A, each `wgA` workgroups x 512 threads x 140 KB LDS holding for `usA` microseconds (the level
// kernels of the chart).  Background: ONE launch on stream B of `wgB` workgroups x 256 threads x 90 KB LDS holding `usB` each (a
// weight-gradient GEMM cut into short slices).  Measured: the chain alone, the background alone, both with B at default priority,
// both with B created at the lowest priority (hipStreamCreateWithPriority) -- the chain's time and the time until both are done.
// build: hipcc --offload-arch=gfx950 -O3 -o priority_bench priority_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
template <int T>
__global__ __launch_bounds__(T) void work(float* p, int ticks) {
    extern __shared__ float sm[];
    const long long t0 = wall_clock64();
    if (threadIdx.x == 0) sm[0] = p[blockIdx.x & 1023];
    __syncthreads();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 1 && sm[0] == 12345.f) p[blockIdx.x & 1023] = 1.f;
}
int main() {
    float* p; CK(hipMalloc(&p, 1 << 20)); CK(hipMemset(p, 0, 1 << 20));
    CK(hipFuncSetAttribute((const void*)work<512>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)work<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    int lo = 0, hi = 0; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    printf("stream priority range: least %d, greatest %d\n", lo, hi);
    hipStream_t sa, sb, sl, sh;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    CK(hipStreamCreateWithPriority(&sl, hipStreamNonBlocking, lo)); CK(hipStreamCreateWithPriority(&sh, hipStreamNonBlocking, hi));
    hipEvent_t e0, eA, eB; CK(hipEventCreate(&e0)); CK(hipEventCreate(&eA)); CK(hipEventCreate(&eB));
    int clk = 0; CK(hipDeviceGetAttribute(&clk, hipDeviceAttributeWallClockRate, 0));
    const int n = 60;
    auto tk = [&](int us) { return (int)((long long)us * clk / 1000); };
    auto run = [&](hipStream_t A, hipStream_t B, bool chain, bool bg, int wgA, int usA, int wgB, int usB, float* tA, float* tAll) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, A)); CK(hipStreamWaitEvent(B, e0, 0));
            if (bg) hipLaunchKernelGGL(work<256>, dim3(wgB), dim3(256), 90 << 10, B, p, tk(usB));
            if (chain) for (int i = 0; i < n; ++i) hipLaunchKernelGGL(work<512>, dim3(wgA), dim3(512), 140 << 10, A, p, tk(usA));
            CK(hipEventRecord(eA, A)); CK(hipEventRecord(eB, B));
            CK(hipEventSynchronize(eA)); CK(hipEventSynchronize(eB));
            float a, b; CK(hipEventElapsedTime(&a, e0, eA)); CK(hipEventElapsedTime(&b, e0, eB));
            *tA = a; *tAll = a > b ? a : b;
        }
    };
    printf("chain: %d kernels; times in ms: chain alone | background alone | default priority: chain, all | low-priority background: chain, all | high-priority chain + low background: chain, all\n", n);
    for (int wgA : {120, 240})
        for (int usA : {10, 25})
            for (int usB : {10, 40, 150}) {
                const int wgB = 256 * 400 / usB;          // ~0.4 ms of whole-chip background work in every case
                float c0, x, b0, c1, a1, c2, a2, c3, a3;
                run(sa, sb, true, false, wgA, usA, wgB, usB, &c0, &x);
                run(sa, sb, false, true, wgA, usA, wgB, usB, &x, &b0);
                run(sa, sb, true, true, wgA, usA, wgB, usB, &c1, &a1);
                run(sa, sl, true, true, wgA, usA, wgB, usB, &c2, &a2);
                run(sh, sl, true, true, wgA, usA, wgB, usB, &c3, &a3);
                printf("chain %3d wg x %2d us, background %5d wg x %3d us | %.3f | %.3f | %.3f %.3f | %.3f %.3f | %.3f %.3f\n", wgA, usA, wgB, usB, c0, b0, c1, a1,
                       c2, a2, c3, a3);
            }
    return 0;
}
