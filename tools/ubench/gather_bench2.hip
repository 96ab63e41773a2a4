// Micro-benchmark 2: what sets the cost of a gathered row load on gfx950?
// Every variant moves the same bytes per 16-row tile (16 rows x 1600 B of operand a + the same of operand b) and issues its
// loads in batches of 16 wave-instructions (issue 16, wait, add up).  What changes is the SHAPE of one wave-instruction:
//   R rows x (1024 / R) contiguous bytes, R = 1, 2, 4, 8, 16.
// Also: 16 instead of 8 waves per workgroup; a compact table (row stride = 1600 B) with CONSECUTIVE rows per tile.
// build: hipcc --offload-arch=gfx950 -O3 -o gather_bench2 gather_bench2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

template <int R, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void gk(const float* __restrict__ T, int ld, int boff, const int* __restrict__ arow,
                                                 const int* __restrict__ brow, int nrows, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int LPR = 64 / R;              // lanes per row
    constexpr int CB = 1024 / R;             // contiguous bytes per row per instruction
    constexpr int NCH = (1600 + CB - 1) / CB; // chunks per row
    constexpr int NSUB = 16 / R;
    const int r = lane / LPR, piece = lane % LPR;
    const int ntiles = (nrows + 15) >> 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int tile = blockIdx.x * WAVES + wave; tile < ntiles; tile += gridDim.x * WAVES) {
        // list of (sub, chunk) instruction pairs; 8 pairs (16 loads) per batch
        constexpr int NP = NSUB * NCH;
        for (int p0 = 0; p0 < NP; p0 += 8) {
            float4 v[8][2];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int p = min(p0 + j, NP - 1);
                const int sub = p / NCH, ch = p - sub * NCH;
                const int row = min(tile * 16 + sub * R + r, nrows - 1);
                const int off = min(ch * (CB / 4) + piece * 4, 396);
                v[j][0] = ld4(T + (size_t)arow[row] * ld + off);
                v[j][1] = ld4(T + (size_t)brow[row] * ld + boff + off);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                acc.x += fmaxf(v[j][0].x + v[j][1].x, 0.f); acc.y += fmaxf(v[j][0].y + v[j][1].y, 0.f);
                acc.z += fmaxf(v[j][0].z + v[j][1].z, 0.f); acc.w += fmaxf(v[j][0].w + v[j][1].w, 0.f);
            }
        }
    }
    out[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * (WAVES * 64) + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}
// plain streaming reference: every wave reads `per_wave` contiguous bytes
__global__ __launch_bounds__(512) void stream_k(const float* __restrict__ T, size_t per_wave_f4, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const size_t w = (size_t)blockIdx.x * 8 + (threadIdx.x >> 6);
    const float4* p = reinterpret_cast<const float4*>(T) + w * per_wave_f4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (size_t e = lane; e < per_wave_f4; e += 64 * 8) {
        float4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = p[min(e + 64 * j, per_wave_f4 - 1)];
#pragma unroll
        for (int j = 0; j < 8; ++j) { acc.x += v[j].x; acc.y += v[j].y; acc.z += v[j].z; acc.w += v[j].w; }
    }
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}
__global__ void fill(float* p, size_t n, unsigned seed) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)e * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[e] = (float)(x & 0xffff) / 65536.f - 0.5f;
    }
}
template <int R, int WAVES>
static float run(const float* T, int ld, int boff, const int* a, const int* b, int nrows, int gx, int gy, float* out) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((gk<R, WAVES>), dim3(gx, gy), dim3(WAVES * 64), 0, 0, T, ld, boff, a, b, nrows, out);
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < 50; ++r) hipLaunchKernelGGL((gk<R, WAVES>), dim3(gx, gy), dim3(WAVES * 64), 0, 0, T, ld, boff, a, b, nrows, out);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 20.f;
}
int main() {
    const int B = 64, L = 20, C = 210;
    const size_t ncell = (size_t)B * C;
    float* T; CK(hipMalloc(&T, ncell * 1200 * sizeof(float)));
    hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, 0, T, ncell * 1200, 7u);
    float* out; CK(hipMalloc(&out, 256 * 8 * 1024 * sizeof(float)));
    auto cell = [&](int level, int pos) { return C - (L - level) * (L - level + 1) / 2 + pos; };
    for (int cfg = 0; cfg < 4; ++cfg) {
        // cfg 0: inside level 10, chart order [b][pos][n] (r01 rows); cfg 1: outside level 0 (24320 rows);
        // cfg 2: inside level 10 in TILE order [b][n][pos] (16 consecutive cells per tile where Lc allows)
        // cfg 3: like 2 but compact tables (ld = 400, PL and PR tables apart)
        std::vector<int> a, b;
        if (cfg == 0) for (int s = 0; s < B; ++s) for (int pos = 0; pos < 10; ++pos) for (int n = 0; n < 10; ++n) { a.push_back(s * C + cell(n, pos)); b.push_back(s * C + cell(9 - n, pos + n + 1)); }
        if (cfg == 1) for (int s = 0; s < B; ++s) for (int pos = 0; pos < L; ++pos) for (int n = 0; n < L - 1; ++n) {
            int sib, par;
            if (n < pos) { par = cell(pos - n, n); sib = cell(pos - 1 - n, n); } else { const int r = pos + 1 + (n - pos); par = cell(r - pos, pos); sib = cell(r - pos - 1, pos + 1); }
            a.push_back(s * C + sib); b.push_back(s * C + par);
        }
        if (cfg >= 2) for (int s = 0; s < B; ++s) for (int n = 0; n < 10; ++n) for (int pos = 0; pos < 10; ++pos) { a.push_back(s * C + cell(n, pos)); b.push_back(s * C + cell(9 - n, pos + n + 1)); }
        const int ld = cfg == 3 ? 400 : 1200;
        const int boff = cfg == 3 ? (int)(ncell * 400) : 400;      // compact: PR table after the PL table
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
            const float t1 = run<1, 8>(T, ld, boff, da, db, nrows, gx, gy, out), t2 = run<2, 8>(T, ld, boff, da, db, nrows, gx, gy, out),
                        t4 = run<4, 8>(T, ld, boff, da, db, nrows, gx, gy, out), t8 = run<8, 8>(T, ld, boff, da, db, nrows, gx, gy, out),
                        t16 = run<16, 8>(T, ld, boff, da, db, nrows, gx, gy, out);
            const int gx16 = (gx + 1) / 2;
            const float w1 = run<1, 16>(T, ld, boff, da, db, nrows, gx16, gy, out), w16 = run<16, 16>(T, ld, boff, da, db, nrows, gx16, gy, out);
            printf("cfg%d rows=%6d grid=(%d,%d) %.1f MB | us by rows/instr: R1 %.1f (%.0f GB/s)  R2 %.1f  R4 %.1f  R8 %.1f  R16 %.1f (%.0f GB/s) | 16 waves/WG, grid (%d,%d): R1 %.1f R16 %.1f\n",
                   cfg, nrows, gx, gy, mb, t1, mb / t1 * 1e3, t2, t4, t8, t16, mb / t16 * 1e3, gx16, gy, w1, w16);
        }
        CK(hipFree(da)); CK(hipFree(db));
    }
    for (size_t kb : {64, 512, 4096}) {     // contiguous streaming: 2048 waves x kb KiB
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const size_t f4 = kb * 1024 / 16;
        if (256 * 8 * f4 * 16 > ncell * 1200 * 4) continue;
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(stream_k, dim3(256), dim3(512), 0, 0, T, f4, out);
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(stream_k, dim3(256), dim3(512), 0, 0, T, f4, out);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("stream %zu KiB per wave (%.1f MB): %.1f us, %.0f GB/s\n", kb, 2048.0 * kb / 1024, ms * 50.f, 2048.0 * kb * 1024 / (ms * 50.f) / 1e3);
    }
    return 0;
}
