// Micro-benchmark of the weight-stationary rows GEMM variants (debug aid, not shipped).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I cliora_amd/csrc tools/kbench.hip -o /tmp/kbench && /tmp/kbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
#include "chart_kernels.hpp"
using namespace cliora;

struct ConstA {
    struct Ctx { float v; };
    using Raw = float4;
    __device__ Ctx row(int r) const { return Ctx{(float)(r & 7)}; }
    __device__ Raw fetch(const Ctx& c, int k) const { return make_float4(c.v, 1.f, 2.f, (float)k); }
    __device__ float4 finish(const Ctx&, const Raw& v) const { return v; }
};
struct NullE {
    float* sink;
    struct RCtx { int r; };
    __device__ RCtx row(int r) const { return RCtx{r}; }
    __device__ void store4(const RCtx& rc, int col, float4 v) const { if (v.x == 12345.678f) sink[0] = v.y; }
};

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int CT, int SC, int WAVES, class AP, class EP>
float run(const char* name, const float* W, int K, int nrows, AP ap, EP ep, int gx_cap = 51) {
    const size_t lds = (size_t)CT * 16 * (K + WS_LDS_PAD) * sizeof(float);
    CK(hipFuncSetAttribute((const void*)rows_gemm_ws<CT, SC, WAVES, AP, EP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int ntiles = (nrows + 15) / 16, gy = (K / 16) / CT;
    int gx = (ntiles + WAVES - 1) / WAVES; if (gx > gx_cap) gx = gx_cap; if (gx < 1) gx = 1;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((rows_gemm_ws<CT, SC, WAVES, AP, EP>), dim3(gx, gy), dim3(WAVES * 64), lds, 0, W, K, K, 1, nrows, ap, ep);
    CK(hipDeviceSynchronize());
    const int it = 20;
    CK(hipEventRecord(a));
    for (int w = 0; w < it; ++w) hipLaunchKernelGGL((rows_gemm_ws<CT, SC, WAVES, AP, EP>), dim3(gx, gy), dim3(WAVES * 64), lds, 0, W, K, K, 1, nrows, ap, ep);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const float us = ms * 1e3f / it;
    printf("%-34s rows %6d grid %3dx%d waves %d : %8.1f us  %6.1f TF\n", name, nrows, gx, gy, WAVES, us, 2.0 * nrows * K * K / us * 1e-6);
    return us;
}

int main() {
    const int D = 400, K = 400, B = 64, C = 210, ldpi = 3 * D;
    const int maxrows = 24320;
    float *W, *PI, *Y, *sink; int32_t *arow, *brow;
    CK(hipMalloc(&W, K * K * 4)); CK(hipMalloc(&PI, (size_t)maxrows * ldpi * 4)); CK(hipMalloc(&Y, (size_t)maxrows * D * 4));
    CK(hipMalloc(&sink, 64)); CK(hipMalloc(&arow, maxrows * 4)); CK(hipMalloc(&brow, maxrows * 4));
    std::vector<float> h((size_t)maxrows * ldpi); for (auto& v : h) v = (rand() % 2001 - 1000) * 1e-3f;
    CK(hipMemcpy(PI, h.data(), h.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(W, h.data(), K * K * 4, hipMemcpyHostToDevice));
    std::vector<int32_t> ia(maxrows), ib(maxrows);
    for (int r = 0; r < maxrows; ++r) { ia[r] = rand() % (B * C); ib[r] = rand() % (B * C); }
    CK(hipMemcpy(arow, ia.data(), maxrows * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(brow, ib.data(), maxrows * 4, hipMemcpyHostToDevice));
    for (int nrows : {0, 1216, 6400, 24320}) {
        ComposeXA cx{arow, brow, 0, PI, ldpi, PI + D, ldpi};
        StoreRowsE se{Y, D, W, 2, D};
        PlainRowsA pa{PI, ldpi};
        run<5, 5, 4>("ComposeXA+Store w4", W, K, nrows, cx, se);
        run<5, 5, 8>("ComposeXA+Store w8", W, K, nrows, cx, se);
        run<5, 5, 4>("PlainRows+Store w4", W, K, nrows, pa, se);
        run<5, 5, 4>("Const+Store w4", W, K, nrows, ConstA{}, se);
        run<5, 5, 4>("Const+Null w4", W, K, nrows, ConstA{}, NullE{sink});
        run<5, 5, 8>("Const+Null w8", W, K, nrows, ConstA{}, NullE{sink});
        run<5, 1, 4>("ComposeXA+Store SC1 w4", W, K, nrows, cx, se);
        run<1, 5, 4>("ComposeXA+Store CT1 w4", W, K, nrows, cx, se, 10);
    }
    return 0;
}
