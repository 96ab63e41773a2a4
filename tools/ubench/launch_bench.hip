// Micro-benchmark: cost of a dependent kernel boundary on one stream for the launch shapes of the level kernels.
// build: hipcc --offload-arch=gfx950 -O3 -o launch_bench launch_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void tiny(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }
__global__ __launch_bounds__(512) void biglds(float* p) {
    extern __shared__ float sm[];
    if (threadIdx.x == 0) sm[0] = p[blockIdx.x];
    __syncthreads();
    if (threadIdx.x == 1 && sm[0] == 12345.f) p[blockIdx.x] = 1.f;
}
// touches `bytes` of memory per block (stream read) to emulate a kernel with a little real work
__global__ __launch_bounds__(256) void reader(const float4* src, float* out, int n4) {
    float4 a = make_float4(0, 0, 0, 0);
    for (int i = threadIdx.x; i < n4; i += 256) { float4 v = src[(size_t)blockIdx.x * n4 + i]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
    if (a.x + a.y + a.z + a.w == 12345.f) out[blockIdx.x] = 1.f;
}
template <class F> static float chain(F f, int n) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i) f(i);
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < n; ++i) f(i);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1000.f / n;
}
int main() {
    float* p; CK(hipMalloc(&p, 64 << 20)); CK(hipMemset(p, 0, 64 << 20));
    CK(hipFuncSetAttribute((const void*)biglds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    printf("tiny 256x256:            %.2f us\n", chain([&](int) { hipLaunchKernelGGL(tiny, dim3(256), dim3(256), 0, 0, p); }, 400));
    printf("tiny 1x64:               %.2f us\n", chain([&](int) { hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, 0, p); }, 400));
    printf("tiny 1280x256:           %.2f us\n", chain([&](int) { hipLaunchKernelGGL(tiny, dim3(1280), dim3(256), 0, 0, p); }, 400));
    for (int lds : {0, 64 << 10, 156 << 10})
        for (int wg : {50, 250})
            printf("biglds %3d WG x 512, %3d KB LDS: %.2f us\n", wg, lds >> 10, chain([&](int) { hipLaunchKernelGGL(biglds, dim3(wg), dim3(512), lds, 0, p); }, 400));
    printf("alternating tiny / biglds(250, 156 KB): %.2f us per launch\n",
           chain([&](int i) { if (i & 1) hipLaunchKernelGGL(biglds, dim3(250), dim3(512), 156 << 10, 0, p); else hipLaunchKernelGGL(tiny, dim3(640), dim3(256), 0, 0, p); }, 400));
    for (int kb : {16, 128}) {
        const int n4 = kb * 1024 / 16;
        printf("reader 256 WG x %3d KB:  %.2f us\n", kb, chain([&](int) { hipLaunchKernelGGL(reader, dim3(256), dim3(256), 0, 0, (const float4*)p, p + (60 << 18), n4); }, 400));
    }
    // hipGraph replay of a 100-kernel chain
    {
        hipStream_t s; CK(hipStreamCreate(&s));
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int i = 0; i < 100; ++i) { if (i & 1) hipLaunchKernelGGL(biglds, dim3(250), dim3(512), 156 << 10, s, p); else hipLaunchKernelGGL(tiny, dim3(640), dim3(256), 0, s, p); }
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < 10; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("graph replay, 100-kernel alternating chain: %.2f us per kernel\n", ms * 1000.f / 1000);
        // two parallel branches of 100 kernels each in one graph
        hipStream_t s2; CK(hipStreamCreate(&s2));
        hipEvent_t f, j; CK(hipEventCreate(&f)); CK(hipEventCreate(&j));
        hipGraph_t g2; hipGraphExec_t ge2;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        CK(hipEventRecord(f, s)); CK(hipStreamWaitEvent(s2, f, 0));
        for (int i = 0; i < 100; ++i) {
            hipLaunchKernelGGL(biglds, dim3(100), dim3(512), 156 << 10, s, p);
            hipLaunchKernelGGL(biglds, dim3(100), dim3(512), 156 << 10, s2, p + 4096);
        }
        CK(hipEventRecord(j, s2)); CK(hipStreamWaitEvent(s, j, 0));
        CK(hipStreamEndCapture(s, &g2));
        CK(hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0));
        for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge2, s));
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < 10; ++i) CK(hipGraphLaunch(ge2, s));
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("graph replay, two parallel 100-kernel branches (100 WG x 156 KB each): %.2f us per graph-level step (pair of kernels)\n", ms * 1000.f / 1000);
    }
    return 0;
}
