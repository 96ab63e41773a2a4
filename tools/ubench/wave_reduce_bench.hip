// Wave-wide sum / max: the ds_bpermute butterfly (__shfl_xor, six dependent LDS-crossbar round trips) against the same butterfly
// on v_permlane32_swap / v_permlane16_swap (gfx950) + DPP row moves -- the SAME pairs in the SAME order, so the results must be
// equal to the bit; prints mismatches and the time of a chain of dependent reductions.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/wave_reduce_bench.hip -o /tmp/wrb && /tmp/wrb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>

template <int CTRL, int BANK>
__device__ __forceinline__ float mdpp(float old, float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, 0xf, BANK, false));
}
template <class Op>
__device__ __forceinline__ float fast_reduce(float v, Op op) {
    {   // lane ^ 32: the halves of the wave swapped
        auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = op(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
    {   // lane ^ 16: odd and even rows swapped
        auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = op(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
    v = op(v, mdpp<0x128, 0xf>(v, v));                                           // row_ror:8  = lane ^ 8
    { float t = mdpp<0x104, 0x5>(v, v); t = mdpp<0x114, 0xA>(t, v); v = op(v, t); }   // row_shl:4 / row_shr:4 by bank = lane ^ 4
    v = op(v, mdpp<0x4E, 0xf>(v, v));                                            // quad_perm [2,3,0,1] = lane ^ 2
    v = op(v, mdpp<0xB1, 0xf>(v, v));                                            // quad_perm [1,0,3,2] = lane ^ 1
    return v;
}
template <class Op>
__device__ __forceinline__ float ref_reduce(float v, Op op) {
    for (int o = 32; o > 0; o >>= 1) v = op(v, __shfl_xor(v, o));
    return v;
}
struct Add { __device__ float operator()(float a, float b) const { return a + b; } };
struct Max { __device__ float operator()(float a, float b) const { return fmaxf(a, b); } };

__global__ void check(const float* x, float* a, float* b, float* c, float* d) {
    const int i = threadIdx.x + blockIdx.x * 64;
    a[i] = fast_reduce(x[i], Add()); b[i] = ref_reduce(x[i], Add());
    c[i] = fast_reduce(x[i], Max()); d[i] = ref_reduce(x[i], Max());
}
template <bool FAST>
__global__ void chain(const float* x, float* out, int n) {
    float v = x[threadIdx.x];
    for (int i = 0; i < n; ++i) v = (FAST ? fast_reduce(v, Add()) : ref_reduce(v, Add())) * 0.015625f + x[threadIdx.x];
    out[threadIdx.x] = v;
}
int main() {
    const int N = 64 * 4096;
    float *x, *r[4];
    (void)hipMallocManaged(&x, N * 4);
    for (auto& p : r) (void)hipMallocManaged(&p, N * 4);
    srand(1);
    for (int i = 0; i < N; ++i) x[i] = ((float)rand() / RAND_MAX * 2.f - 1.f) * (i % 7 == 0 ? 1e3f : 1.f);
    hipLaunchKernelGGL(check, dim3(N / 64), dim3(64), 0, 0, x, r[0], r[1], r[2], r[3]);
    (void)hipDeviceSynchronize();
    int bad_s = 0, bad_m = 0;
    for (int i = 0; i < N; ++i) { bad_s += memcmp(&r[0][i], &r[1][i], 4) != 0; bad_m += memcmp(&r[2][i], &r[3][i], 4) != 0; }
    printf("sum mismatches %d, max mismatches %d of %d\n", bad_s, bad_m, N);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int n = 20000;
    for (int fast = 0; fast < 2; ++fast) {
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0);
            if (fast) hipLaunchKernelGGL(chain<true>, dim3(1), dim3(64), 0, 0, x, r[0], n);
            else hipLaunchKernelGGL(chain<false>, dim3(1), dim3(64), 0, 0, x, r[0], n);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("%s: %.1f ns per dependent wave reduction\n", fast ? "permlane swap + DPP" : "ds_bpermute butterfly", ms * 1e6 / n);
        }
    }
    return (bad_s | bad_m) != 0;
}
