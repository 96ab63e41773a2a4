// Micro-benchmark 4 (from gather_bench3): does the operand ring ingest faster when a load instruction touches FULL 128-byte lines?
//   M1  quad-coalesced (what the compose kernels do): lane l reads row (l >> 2), 16 B at k = 32s + 4(l & 3) and at +16: an instruction = 16 rows x 64 B
//   M3  line-coalesced: lane l reads row (l >> 3) [second load: row 8 + (l >> 3)], 16 B at k = 32s + 4(l & 7): an instruction = 8 rows x 128 B
//   each with the row pitch as in the library (1200 floats, PR at +400: a k-step's 128 B piece straddles two lines on every other row) and with a
//   line-aligned layout (pitch 1248 floats, PR at +416: every piece is one line)
// Micro-benchmark 3: the r01 operand ring (4 k-steps in flight, counted vmcnt) with two lane -> address maps
//   M0  r01 / MFMA-native: lane l reads row (l & 15), 16 B at k = 32s + 4(l >> 4) and at +16   (a quad = 4 rows x 16 B)
//   M1  quad-coalesced:     lane l reads row (l >> 2), 16 B at k = 32s + 4(l & 3) and at +16   (a quad = 64 contiguous B)
//   M2  M1 + ds_bpermute of the two summed float4 back to the MFMA-native lanes (8 bpermutes per k-step)
// build: hipcc --offload-arch=gfx950 -O3 -o gather_bench3 gather_bench3.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 relu_add(float4 a, float4 b) {
    return make_float4(fmaxf(a.x + b.x, 0.f), fmaxf(a.y + b.y, 0.f), fmaxf(a.z + b.z, 0.f), fmaxf(a.w + b.w, 0.f));
}
__device__ __forceinline__ float bperm(int addr, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr, __builtin_bit_cast(int, v)));
}
template <int M, int PD>
__global__ __launch_bounds__(512) void gk(const float* __restrict__ T, int ld, int offb, const int* __restrict__ arow,
                                          const int* __restrict__ brow, int nrows, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rsel = M == 0 ? (lane & 15) : (M == 3 ? (lane >> 3) : (lane >> 2));
    const int psel = M == 0 ? (lane >> 4) : (M == 3 ? (lane & 7) : (lane & 3));
    const int paddr = 4 * (4 * (lane & 15) + (lane >> 4));      // bpermute source lane (byte address) for MFMA-native lane `lane`
    const int ntiles = (nrows + 15) >> 4;
    const int stride = gridDim.x * 8;
    constexpr int nsteps = 13;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int tile = blockIdx.x * 8 + wave;
    if (tile >= ntiles) return;
    auto rowp = [&](int t, const int* tab, int off) { return T + (size_t)tab[min(t * 16 + rsel, nrows - 1)] * ld + off; };
    const float *pa = rowp(tile, arow, 0), *pb = rowp(tile, brow, offb);
    const size_t r8 = M == 3 ? (size_t)8 : 0;       // M3: the second load of a k-step is the row 8 further down the tile (same k piece)
    auto row2 = [&](int t, const int* tab, int off) { return T + (size_t)tab[min(t * 16 + rsel + 8, nrows - 1)] * ld + off; };
    const float *pa2 = M == 3 ? row2(tile, arow, 0) : pa, *pb2 = M == 3 ? row2(tile, brow, offb) : pb;
    (void)r8;
    float4 ra[PD][4];
    auto issue = [&](int sl, const float* a, const float* b, const float* a2, const float* b2, int s) {
        const int k = 32 * s + 4 * psel;
        const int k2 = M == 3 ? k : k + (32 * s + 16 < 400 ? 16 : 0);
        ra[sl][0] = ld4(a + k); ra[sl][1] = ld4(a2 + k2); ra[sl][2] = ld4(b + k); ra[sl][3] = ld4(b2 + k2);
    };
#pragma unroll
    for (int sl = 0; sl < PD; ++sl) issue(sl, pa, pb, pa2, pb2, sl);
    while (true) {
        const int ntile = tile + stride;
        const bool has_next = ntile < ntiles;
        const float *pan = rowp(has_next ? ntile : tile, arow, 0), *pbn = rowp(has_next ? ntile : tile, brow, offb);
        const float *pan2 = M == 3 ? row2(has_next ? ntile : tile, arow, 0) : pan, *pbn2 = M == 3 ? row2(has_next ? ntile : tile, brow, offb) : pbn;
#pragma unroll
        for (int base = 0; base < 16; base += PD) {
#pragma unroll
            for (int sl = 0; sl < PD; ++sl) {
                const int st = base + sl;
                if (st < nsteps) {
                    float4 sA = relu_add(ra[sl][0], ra[sl][2]), sB = relu_add(ra[sl][1], ra[sl][3]);
                    if (M == 2) {
                        sA = make_float4(bperm(paddr, sA.x), bperm(paddr, sA.y), bperm(paddr, sA.z), bperm(paddr, sA.w));
                        sB = make_float4(bperm(paddr, sB.x), bperm(paddr, sB.y), bperm(paddr, sB.z), bperm(paddr, sB.w));
                    }
                    acc.x += sA.x + sB.x; acc.y += sA.y + sB.y; acc.z += sA.z + sB.z; acc.w += sA.w + sB.w;
                }
                const int nst = st + PD;
                const bool in_cur = nst < nsteps;
                issue(sl, in_cur ? pa : pan, in_cur ? pb : pbn, in_cur ? pa2 : pan2, in_cur ? pb2 : pbn2, in_cur ? nst : (sl < nsteps ? sl : 0));
            }
        }
        if (!has_next) break;
        pa = pan; pb = pbn; pa2 = pan2; pb2 = pbn2; tile = ntile;
    }
    out[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 512 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}
__global__ void fill(float* p, size_t n, unsigned seed) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)e * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[e] = (float)(x & 0xffff) / 65536.f - 0.5f;
    }
}
template <int M, int PD>
static float run(const float* T, int ld, int offb, const int* a, const int* b, int nrows, int gx, int gy, float* out) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((gk<M, PD>), dim3(gx, gy), dim3(512), 0, 0, T, ld, offb, a, b, nrows, out);
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < 50; ++r) hipLaunchKernelGGL((gk<M, PD>), dim3(gx, gy), dim3(512), 0, 0, T, ld, offb, a, b, nrows, out);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 20.f;
}
int main() {
    const int B = 64, L = 20, C = 210;
    const size_t ncell = (size_t)B * C;
    float* T; CK(hipMalloc(&T, ncell * 1248 * sizeof(float)));
    hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, 0, T, ncell * 1248, 7u);
    float* out; CK(hipMalloc(&out, 256 * 8 * 512 * sizeof(float)));
    auto cell = [&](int level, int pos) { return C - (L - level) * (L - level + 1) / 2 + pos; };
    for (int cfg = 0; cfg < 3; ++cfg) {
        std::vector<int> a, b;
        if (cfg == 0) for (int s = 0; s < B; ++s) for (int pos = 0; pos < 10; ++pos) for (int n = 0; n < 10; ++n) { a.push_back(s * C + cell(n, pos)); b.push_back(s * C + cell(9 - n, pos + n + 1)); }
        if (cfg == 1) for (int s = 0; s < B; ++s) for (int pos = 0; pos < L; ++pos) for (int n = 0; n < L - 1; ++n) {
            int sib, par;
            if (n < pos) { par = cell(pos - n, n); sib = cell(pos - 1 - n, n); } else { const int r = pos + 1 + (n - pos); par = cell(r - pos, pos); sib = cell(r - pos - 1, pos + 1); }
            a.push_back(s * C + sib); b.push_back(s * C + par);
        }
        if (cfg == 2) for (int s = 0; s < B; ++s) for (int pos = 0; pos < 1; ++pos) for (int n = 0; n < 19; ++n) { a.push_back(s * C + cell(n, pos)); b.push_back(s * C + cell(18 - n, pos + n + 1)); }
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
            const float a1 = run<1, 4>(T, 1200, 400, da, db, nrows, gx, gy, out), a3 = run<3, 4>(T, 1200, 400, da, db, nrows, gx, gy, out);
            const float b1 = run<1, 4>(T, 1248, 416, da, db, nrows, gx, gy, out), b3 = run<3, 4>(T, 1248, 416, da, db, nrows, gx, gy, out);
            const float c1 = run<1, 6>(T, 1248, 416, da, db, nrows, gx, gy, out), c3 = run<3, 6>(T, 1248, 416, da, db, nrows, gx, gy, out);
            const int ncu = gx * gy;
            printf("cfg%d rows=%6d grid=(%d,%d) %.1f MB | pitch 1200: quad %.1f us (%.0f GB/s, %.0f per CU)  line %.1f (%.0f, %.0f) | pitch 1248 aligned: quad %.1f (%.0f, %.0f)  line %.1f (%.0f, %.0f) | aligned, ring 6: quad %.1f line %.1f\n",
                   cfg, nrows, gx, gy, mb, a1, mb / a1 * 1e3, mb / a1 * 1e3 / ncu, a3, mb / a3 * 1e3, mb / a3 * 1e3 / ncu, b1, mb / b1 * 1e3, mb / b1 * 1e3 / ncu, b3, mb / b3 * 1e3, mb / b3 * 1e3 / ncu, c1, c3);
        }
        CK(hipFree(da)); CK(hipFree(db));
    }
    return 0;
}
