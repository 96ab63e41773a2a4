// Micro-benchmark: how fast can the compose kernels' operand gather pull rows into a CU?
// Emulates rows_gemm_ws3's access: grid (gx, GY) workgroups of 8 waves, a wave walks 16-row tiles, per 32-deep k-step it
// fetches 2 x 16 B of the PL(a) row and of the PR(b) row; GY column blocks re-read the same rows (same XCD by id % 8).
//   V0  r01 pattern: one instruction = 16 rows x 64 B (half cache lines)
//   V1  full lines:  one instruction = 8 rows x 128 B, lanes i and i+8 swap one float4 afterwards (DPP row_ror:8)
//   V2  V1 without the swap (upper bound of V1)
//   V3  whole-row pieces: one instruction = 1 KiB of ONE row (rows handled one after the other), no MFMA-ready layout
// build: hipcc --offload-arch=gfx950 -O3 -o gather_bench gather_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 relu_add(float4 a, float4 b) {
    return make_float4(fmaxf(a.x + b.x, 0.f), fmaxf(a.y + b.y, 0.f), fmaxf(a.z + b.z, 0.f), fmaxf(a.w + b.w, 0.f));
}
__device__ __forceinline__ float4 ror8(float4 v) {   // lanes i <-> i+8 inside each row of 16 lanes
    float4 r;
    r.x = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v.x), 0x128, 0xf, 0xf, false));
    r.y = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v.y), 0x128, 0xf, 0xf, false));
    r.z = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v.z), 0x128, 0xf, 0xf, false));
    r.w = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v.w), 0x128, 0xf, 0xf, false));
    return r;
}

template <int V, int PD>
__global__ __launch_bounds__(512) void gather_kernel(const float* __restrict__ T, int ld, const int* __restrict__ arow,
                                                     const int* __restrict__ brow, int nrows, int K, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int ntiles = (nrows + 15) >> 4;
    const int stride = gridDim.x * 8;
    const int nsteps = (K + 31) / 32;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int tile = blockIdx.x * 8 + wave; tile < ntiles; tile += stride) {
        if (V == 3) {
            // whole rows: 16 rows x 2 operands, each 1600 B = 100 float4: lanes 0..63 then 0..35
            for (int r = 0; r < 16; r += 4) {
                float4 v[4][4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int row = min(tile * 16 + r + j, nrows - 1);
                    const float* pa = T + (size_t)arow[row] * ld;
                    const float* pb = T + (size_t)brow[row] * ld + 400;
                    v[j][0] = ld4(pa + 4 * lane);
                    v[j][1] = ld4(pa + 4 * min(lane + 64, 99));
                    v[j][2] = ld4(pb + 4 * lane);
                    v[j][3] = ld4(pb + 4 * min(lane + 64, 99));
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 s0 = relu_add(v[j][0], v[j][2]), s1 = relu_add(v[j][1], v[j][3]);
                    acc.x += s0.x + s1.x; acc.y += s0.y + s1.y; acc.z += s0.z + s1.z; acc.w += s0.w + s1.w;
                }
            }
            continue;
        }
        const float *pa, *pb;
        int o0, o1;
        if (V == 0) {
            const int row = min(tile * 16 + i, nrows - 1);
            pa = T + (size_t)arow[row] * ld;
            pb = T + (size_t)brow[row] * ld + 400;
            o0 = 4 * g; o1 = 16 + 4 * g;
        } else {
            // instruction A: rows (i&7), instruction B: rows 8 + (i&7); byte offset in the 128-B line = 16*(g + 4*(i>>3))
            const int rA = min(tile * 16 + (i & 7), nrows - 1), rB = min(tile * 16 + 8 + (i & 7), nrows - 1);
            pa = T + (size_t)arow[rA] * ld;            // V1/V2 keep two row pointers per operand
            pb = T + (size_t)brow[rA] * ld + 400;
            o0 = 4 * (g + 4 * (i >> 3));
            o1 = (int)(((size_t)arow[rB] - (size_t)arow[rA]) * ld);   // element delta to row B of the a-operand
            // b-operand delta
            acc.w += 0.f;
            // store the b delta in a second variable through the loop below
            const int db = (int)(((size_t)brow[rB] - (size_t)brow[rA]) * ld);
            float4 ra[PD][4];
            auto issue = [&](int sl, int s) {
                const int k = 32 * s + o0;
                ra[sl][0] = ld4(pa + k); ra[sl][1] = ld4(pa + o1 + k);
                ra[sl][2] = ld4(pb + k); ra[sl][3] = ld4(pb + db + k);
            };
#pragma unroll
            for (int sl = 0; sl < PD; ++sl) issue(sl, min(sl, nsteps - 1));
            for (int base = 0; base < nsteps; base += PD) {
#pragma unroll
                for (int sl = 0; sl < PD; ++sl) {
                    const int st = base + sl;
                    if (st < nsteps) {
                        float4 sA = relu_add(ra[sl][0], ra[sl][2]);     // rows 0-7 of the tile, this lane's piece
                        float4 sB = relu_add(ra[sl][1], ra[sl][3]);     // rows 8-15
                        if (V == 1) {
                            // lane i < 8 keeps sA (its own first half) and needs lane i+8's sA (its second half);
                            // lane i >= 8 keeps sB and needs lane i-8's sB
                            const bool lo = (i < 8);
                            const float4 give = lo ? sB : sA;
                            const float4 got = ror8(give);
                            const float4 keep = lo ? sA : sB;
                            sA = keep; sB = got;
                        }
                        acc.x += sA.x + sB.x; acc.y += sA.y + sB.y; acc.z += sA.z + sB.z; acc.w += sA.w + sB.w;
                    }
                    issue(sl, min(st + PD, nsteps - 1));
                }
            }
            continue;
        }
        float4 ra[PD][4];
        auto issue = [&](int sl, int s) {
            const int k = 32 * s;
            ra[sl][0] = ld4(pa + k + o0); ra[sl][1] = ld4(pa + k + (k + 16 < K ? o1 : o0));
            ra[sl][2] = ld4(pb + k + o0); ra[sl][3] = ld4(pb + k + (k + 16 < K ? o1 : o0));
        };
#pragma unroll
        for (int sl = 0; sl < PD; ++sl) issue(sl, min(sl, nsteps - 1));
        for (int base = 0; base < nsteps; base += PD) {
#pragma unroll
            for (int sl = 0; sl < PD; ++sl) {
                const int st = base + sl;
                if (st < nsteps) {
                    const float4 sA = relu_add(ra[sl][0], ra[sl][2]), sB = relu_add(ra[sl][1], ra[sl][3]);
                    acc.x += sA.x + sB.x; acc.y += sA.y + sB.y; acc.z += sA.z + sB.z; acc.w += sA.w + sB.w;
                }
                issue(sl, min(st + PD, nsteps - 1));
            }
        }
    }
    out[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 512 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

__global__ void fill(float* p, size_t n, unsigned seed) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)e * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[e] = (float)(x & 0xffff) / 65536.f - 0.5f;
    }
}
// trivial kernels for the launch-boundary measurement
__global__ void tiny(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }

template <int V, int PD>
static float run(const float* T, int ld, const int* a, const int* b, int nrows, int gx, int gy, float* out, int reps) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((gather_kernel<V, PD>), dim3(gx, gy), dim3(512), 0, 0, T, ld, a, b, nrows, 400, out);
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((gather_kernel<V, PD>), dim3(gx, gy), dim3(512), 0, 0, T, ld, a, b, nrows, 400, out);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1000.f / reps;
}

int main() {
    const int B = 64, L = 20, C = 210, ld = 1200;
    const size_t ncell = (size_t)B * C;
    float* T; CK(hipMalloc(&T, ncell * ld * sizeof(float)));
    hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, 0, T, ncell * ld, 7u);
    float* out; CK(hipMalloc(&out, 256 * 8 * 512 * sizeof(float)));
    // inside level lv (Lc = L - lv cells, N = lv splits), row order [b][pos][n]; and the tile-major order [b][n][pos-chunk] is the
    // same set of rows, so only this order is timed
    for (int lv : {10, 19, 1}) {
        for (int outside = 0; outside < 2; ++outside) {
            if (outside && lv != 1) continue;
            std::vector<int> a, b;
            auto cell = [&](int level, int pos) { return C - (L - level) * (L - level + 1) / 2 + pos; };
            if (!outside) {
                for (int s = 0; s < B; ++s) for (int pos = 0; pos < L - lv; ++pos) for (int n = 0; n < lv; ++n) {
                    a.push_back(s * C + cell(n, pos)); b.push_back(s * C + cell(lv - n - 1, pos + n + 1));
                }
            } else {   // outside level 0: 19 (sibling, parent) pairs per leaf
                for (int s = 0; s < B; ++s) for (int pos = 0; pos < L; ++pos) for (int n = 0; n < L - 1; ++n) {
                    int sib, par;
                    if (n < pos) { par = cell(pos - n, n); sib = cell(pos - 1 - n, n); }
                    else { const int r = pos + 1 + (n - pos); par = cell(r - pos, pos); sib = cell(r - pos - 1, pos + 1); }
                    a.push_back(s * C + sib); b.push_back(s * C + par);
                }
            }
            const int nrows = (int)a.size();
            int *da, *db; CK(hipMalloc(&da, nrows * 4)); CK(hipMalloc(&db, nrows * 4));
            CK(hipMemcpy(da, a.data(), nrows * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), nrows * 4, hipMemcpyHostToDevice));
            const int ntiles = (nrows + 15) / 16;
            for (int gy : {5, 1}) {
                const int cap = 256 / gy;
                const int passes = (ntiles + 8 * cap - 1) / (8 * cap);
                int gx = (ntiles + 8 * passes - 1) / (8 * passes);
                if (gx >= 8 && (gx + 7) / 8 * 8 <= cap) gx = (gx + 7) / 8 * 8;
                const double mb = (double)nrows * gy * 3200.0 / 1e6;
                const float t0 = run<0, 4>(T, ld, da, db, nrows, gx, gy, out, 50);
                const float t1 = run<1, 4>(T, ld, da, db, nrows, gx, gy, out, 50);
                const float t2 = run<2, 4>(T, ld, da, db, nrows, gx, gy, out, 50);
                const float t3 = run<3, 4>(T, ld, da, db, nrows, gx, gy, out, 50);
                const float t08 = run<0, 8>(T, ld, da, db, nrows, gx, gy, out, 50);
                const float t18 = run<1, 8>(T, ld, da, db, nrows, gx, gy, out, 50);
                printf("%s lv=%2d rows=%6d grid=(%d,%d) ingest=%.1f MB | V0 %.1f us (%.0f GB/s, %.1f GB/s/CU) | V1 %.1f us (%.0f) | V2 %.1f us (%.0f) | V3 %.1f us (%.0f) | PD8: V0 %.1f V1 %.1f\n",
                       outside ? "out" : "in ", lv, nrows, gx, gy, mb, t0, mb / t0 * 1e3, mb / t0 * 1e3 / (gx * gy), t1, mb / t1 * 1e3, t2, mb / t2 * 1e3, t3, mb / t3 * 1e3, t08, t18);
            }
            CK(hipFree(da)); CK(hipFree(db));
        }
    }
    // launch boundary: 200 dependent tiny kernels
    {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int w = 0; w < 10; ++w) hipLaunchKernelGGL(tiny, dim3(256), dim3(256), 0, 0, out);
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < 200; ++r) hipLaunchKernelGGL(tiny, dim3(256), dim3(256), 0, 0, out);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("tiny kernel chain: %.2f us per launch\n", ms * 1000.f / 200);
    }
    return 0;
}
