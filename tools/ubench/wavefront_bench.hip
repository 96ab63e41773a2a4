// Micro-benchmark: two dependent kernel chains on two HIP streams, run in lockstep with cross-stream events
// (the inside / outside wavefront: step k of chain B needs step k-1 of chain A), against the same 2n kernels on one stream.
// Each kernel holds `wg` workgroups of 512 threads with `lds` KB of LDS for `us` microseconds (spin on the wall clock).
// build: hipcc --offload-arch=gfx950 -O3 -o wavefront_bench wavefront_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ __launch_bounds__(512) void work(float* p, int ticks) {
    extern __shared__ float sm[];
    const long long t0 = wall_clock64();
    if (threadIdx.x == 0) sm[0] = p[blockIdx.x];
    __syncthreads();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 1 && sm[0] == 12345.f) p[blockIdx.x] = 1.f;
}
int main() {
    float* p; CK(hipMalloc(&p, 64 << 20)); CK(hipMemset(p, 0, 64 << 20));
    CK(hipFuncSetAttribute((const void*)work, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    const int n = 100;
    hipEvent_t ev[n], e0, e1, ej; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
    for (int i = 0; i < n; ++i) CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
    int clk = 0; CK(hipDeviceGetAttribute(&clk, hipDeviceAttributeWallClockRate, 0));   // kHz
    printf("wall clock %d kHz\n", clk);
    for (int us : {8, 15, 30})
        for (int wg : {60, 125, 250})
            for (int lds : {0, 140 << 10}) {
                const int ticks = (int)((long long)us * clk / 1000);
                float ms1, ms2, ms3;
                // (1) one stream, 2n kernels
                for (int rep = 0; rep < 2; ++rep) {
                    CK(hipEventRecord(e0, sa));
                    for (int i = 0; i < 2 * n; ++i) hipLaunchKernelGGL(work, dim3(wg), dim3(512), lds, sa, p, ticks);
                    CK(hipEventRecord(e1, sa)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms1, e0, e1));
                }
                // (2) two streams, n kernels each, independent
                for (int rep = 0; rep < 2; ++rep) {
                    CK(hipEventRecord(e0, sa)); CK(hipStreamWaitEvent(sb, e0, 0));
                    for (int i = 0; i < n; ++i) {
                        hipLaunchKernelGGL(work, dim3(wg), dim3(512), lds, sa, p, ticks);
                        hipLaunchKernelGGL(work, dim3(wg), dim3(512), lds, sb, p + 4096, ticks);
                    }
                    CK(hipEventRecord(ej, sb)); CK(hipStreamWaitEvent(sa, ej, 0));
                    CK(hipEventRecord(e1, sa)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms2, e0, e1));
                }
                // (3) two streams in lockstep: kernel i of chain B waits for kernel i-1 of chain A
                for (int rep = 0; rep < 2; ++rep) {
                    CK(hipEventRecord(e0, sa)); CK(hipStreamWaitEvent(sb, e0, 0));
                    for (int i = 0; i < n; ++i) {
                        hipLaunchKernelGGL(work, dim3(wg), dim3(512), lds, sa, p, ticks);
                        if (i > 0) CK(hipStreamWaitEvent(sb, ev[i - 1], 0));
                        hipLaunchKernelGGL(work, dim3(wg), dim3(512), lds, sb, p + 4096, ticks);
                        CK(hipEventRecord(ev[i], sa));
                    }
                    CK(hipEventRecord(ej, sb)); CK(hipStreamWaitEvent(sa, ej, 0));
                    CK(hipEventRecord(e1, sa)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms3, e0, e1));
                }
                printf("kernel %2d us, %3d WG, %3d KB LDS: one stream %.2f us/kernel | two streams %.2f us/pair | lockstep with events %.2f us/pair\n",
                       us, wg, lds >> 10, ms1 * 1000 / (2 * n), ms2 * 1000 / n, ms3 * 1000 / n);
            }
    return 0;
}
