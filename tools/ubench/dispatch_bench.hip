// Micro-benchmark: how fast does the chip START workgroups?  The per-level projection launches (level_project / rows_gemm_ksplit) are
// 5 000 - 7 500 workgroups of 256 threads that each do a few microseconds of work; this measures an empty and a one-round-trip kernel
// of that shape, per launch on one stream, by grid size, block size, VGPR budget and LDS size.
// build: hipcc --offload-arch=gfx950 -O3 -o dispatch_bench dispatch_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
template <int T> __global__ __launch_bounds__(T) void empty_k(float* p) { if (p == nullptr) p[threadIdx.x] = 1.f; }
// one dependent global round trip per thread (index -> row), then a store: the minimal "latency chain" block
template <int T> __global__ __launch_bounds__(T) void trip_k(const int* idx, const float4* rows, float4* out, int n) {
    const int b = blockIdx.x;
    const int r = idx[b % n];
    float4 v = rows[(size_t)r * 64 + (threadIdx.x & 63)];
    out[(size_t)b * T + threadIdx.x] = v;
}
template <int T> __global__ __launch_bounds__(T) void lds_k(float* p) {
    __shared__ float sm[1088];
    sm[threadIdx.x] = (float)threadIdx.x;
    __syncthreads();
    if (sm[(threadIdx.x + 1) % T] == -1.f) p[blockIdx.x] = 1.f;
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
    float* p; CK(hipMalloc(&p, 256 << 20)); CK(hipMemset(p, 0, 256 << 20));
    int* idx; CK(hipMalloc(&idx, 1 << 20)); CK(hipMemset(idx, 0, 1 << 20));
    float4* rows = (float4*)(p + (32 << 20));
    float4* out = (float4*)p;
    for (int g : {256, 1280, 2560, 5120, 7680, 10240, 20480}) {
        const float e256 = chain([&](int) { hipLaunchKernelGGL(empty_k<256>, dim3(g), dim3(256), 0, 0, p); }, 300);
        const float e64 = chain([&](int) { hipLaunchKernelGGL(empty_k<64>, dim3(g * 4), dim3(64), 0, 0, p); }, 300);
        const float e512 = chain([&](int) { hipLaunchKernelGGL(empty_k<512>, dim3(g / 2), dim3(512), 0, 0, p); }, 300);
        const float e1024 = chain([&](int) { hipLaunchKernelGGL(empty_k<1024>, dim3(g / 4), dim3(1024), 0, 0, p); }, 300);
        const float l256 = chain([&](int) { hipLaunchKernelGGL(lds_k<256>, dim3(g), dim3(256), 0, 0, p); }, 300);
        const float t256 = chain([&](int) { hipLaunchKernelGGL(trip_k<256>, dim3(g), dim3(256), 0, 0, idx, rows, out, 1024); }, 300);
        printf("%6d x 256 threads (%7d waves): empty %6.2f us | as x64 blocks %6.2f | as x512 %6.2f | as x1024 %6.2f | 4 KB LDS + barrier %6.2f | one round trip + store %6.2f\n",
               g, g * 4, e256, e64, e512, e1024, l256, t256);
    }
    return 0;
}
