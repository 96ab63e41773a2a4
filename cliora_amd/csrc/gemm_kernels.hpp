// MFMA building blocks for the chart engine (gfx950, wave64, fp32-in/fp32-acc MFMA).
//
// Two shapes cover every matmul on the path:
//
//  rows_gemm_ws   out[r][j] = sum_k A(r,k) * W[j][k]          ("weight stationary")
//      A block of CT*16 weight rows stays resident in LDS for the life of the
//      workgroup; every wave walks 16-row tiles of A on its own, producing its
//      A fragments straight from global memory through a functor (gather + add +
//      ReLU for the compose layer), so the main loop has no barrier at all.
//
//  tn_gemm        C[i][j]  = sum_r A(r,i) * B(r,j)             (weight gradients)
//      split-K over waves; each wave owns a (TI*16 x TJ*16) block of C for one
//      slice of rows and writes its partial block to a slab; a second kernel sums
//      the slabs in a fixed order (bitwise reproducible, no atomics).
//
// v_mfma_f32_16x16x4_f32 operand maps (cdna_hip_programming.md section 3):
//   A[i = lane&15][k = lane>>4], B[k = lane>>4][j = lane&15],
//   D: col = lane&15, row = (lane>>4)*4 + reg.
// The k index inside a 16-deep chunk is permuted (lane group q takes k = 4q..4q+3
// over four MFMAs) so that each lane fetches one 16-byte vector per operand per
// chunk; A and B use the same permutation, the sum over k is unchanged.
#pragma once
#include <hip/hip_runtime.h>

namespace cliora {

using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

constexpr int WS_THREADS = 256;   // tn_gemm workgroup size
constexpr int WS_LDS_PAD = 0;   // the LDS image is the plain row-major block (written by LDS-DMA, lane-linear)

// ---------------------------------------------------------------------------------
// rows_gemm_ws
//   W      : [ncols_total][ldw] row-major (the nn.Linear layout: out x in); the reduction runs over
//            nseg segments of Kseg columns (Kseg % (16*SC) == 0); one segment is LDS-resident at a time
//   grid.y : column blocks of CT*16 output columns;  grid.x: row-tile walkers
//   WAVES  : wavefronts per workgroup (4 = one per SIMD, 8 = two per SIMD to hide load latency)
//   SC     : 16-deep k-chunks fetched per prefetch stage; the A fragments of stage s+1 are
//            in flight while the MFMAs of stage s issue (register double buffer, no barrier)
//   AProd  : Ctx row(int r);  Raw fetch(const Ctx&, int k)  (issues the loads);
//            float4 finish(const Ctx&, const Raw&)          (ALU part: add / ReLU / mask)
//   Epi    : RCtx row(int r);  void store4(const RCtx&, int col, float4 v)   (4 consecutive columns)
//   The weight fragment is the MFMA's A operand and the row fragment its B operand, i.e. the tile is
//   computed transposed: a lane ends up with 4 CONSECUTIVE output columns of ONE row per accumulator,
//   so the epilogue is one 16-byte access per lane per 16x16 tile.
// ---------------------------------------------------------------------------------
template <int CT, int SC, int WAVES, class AProd, class Epi>
__global__ __launch_bounds__(WAVES * 64) void rows_gemm_ws(const float* __restrict__ W, int ldw, int Kseg, int nseg, int nrows,
                                                           AProd ap, Epi epi) {
    extern __shared__ __attribute__((aligned(16))) float lds_w[];
    constexpr int T = WAVES * 64;
    const int ldb = Kseg + WS_LDS_PAD;
    const int col0 = blockIdx.y * (CT * 16);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int ntiles = (nrows + 15) >> 4;
    const float* bbase = lds_w + i * ldb + 4 * q;
    using Raw = typename AProd::Raw;

    // stage the CT*16 x Kseg weight block of K-segment `seg` into LDS with LDS-DMA: every wave
    // instruction moves 1 KiB (lane -> 16 B), all of a wave's pieces are in flight together
    auto stage = [&](int seg) {
        const float* Wb = W + (size_t)col0 * ldw + (size_t)seg * Kseg;
        const int k4 = Kseg >> 2, n4 = CT * 16 * k4;
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        for (int e0 = wv * 64; e0 < n4; e0 += T) {
            const int e = e0 + lane;
            if (e < n4) {
                const int r = e / k4, c = e - r * k4;
                __builtin_amdgcn_global_load_lds((const void*)(Wb + (size_t)r * ldw + c * 4),
                                                 (__attribute__((address_space(3))) void*)(lds_w + e0 * 4), 16, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    if (nseg == 1) { stage(0); __syncthreads(); }

    for (int tile0 = blockIdx.x * WAVES; tile0 < ntiles; tile0 += gridDim.x * WAVES) {
        const int tile = tile0 + wave;
        const bool active = tile < ntiles;       // inactive waves still take part in the barriers
        int row = tile * 16 + i;
        if (row >= nrows) row = nrows - 1;       // clamp: computed, never stored
        const auto ctx = ap.row(row);
        f32x4 acc[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int seg = 0; seg < nseg; ++seg) {
            if (nseg > 1) { __syncthreads(); stage(seg); __syncthreads(); }
            if (!active) continue;
            const int kbase = seg * Kseg;
            Raw cur[SC], nxt[SC];
#pragma unroll
            for (int j = 0; j < SC; ++j) cur[j] = ap.fetch(ctx, kbase + 16 * j + 4 * q);
            for (int ks = 0; ks < Kseg; ks += 16 * SC) {
                const bool more = ks + 16 * SC < Kseg;
                if (more) {
#pragma unroll
                    for (int j = 0; j < SC; ++j) nxt[j] = ap.fetch(ctx, kbase + ks + 16 * (SC + j) + 4 * q);
                }
#pragma unroll
                for (int j = 0; j < SC; ++j) {
                    const float4 a = ap.finish(ctx, cur[j]);
                    float4 b[CT];
#pragma unroll
                    for (int c = 0; c < CT; ++c) b[c] = *reinterpret_cast<const float4*>(bbase + c * 16 * ldb + ks + 16 * j);
                    // k-step outermost: consecutive MFMAs hit different accumulators (the 16x16x4 f32 MFMA
                    // has a 40-cycle dependent latency against a 32-cycle issue interval)
#pragma unroll
                    for (int c = 0; c < CT; ++c) acc[c] = mfma16(b[c].x, a.x, acc[c]);
#pragma unroll
                    for (int c = 0; c < CT; ++c) acc[c] = mfma16(b[c].y, a.y, acc[c]);
#pragma unroll
                    for (int c = 0; c < CT; ++c) acc[c] = mfma16(b[c].z, a.z, acc[c]);
#pragma unroll
                    for (int c = 0; c < CT; ++c) acc[c] = mfma16(b[c].w, a.w, acc[c]);
                }
                if (more) {
#pragma unroll
                    for (int j = 0; j < SC; ++j) cur[j] = nxt[j];
                }
            }
        }
        if (active && tile * 16 + i < nrows) {
            const auto rc = epi.row(tile * 16 + i);
#pragma unroll
            for (int c = 0; c < CT; ++c)
                epi.store4(rc, col0 + c * 16 + 4 * q, make_float4(acc[c][0], acc[c][1], acc[c][2], acc[c][3]));
        }
    }
}

// ---------------------------------------------------------------------------------
// tn_gemm:  C[i][j] = sum_r A(r,i) B(r,j),  i < Mi, j < Nj  (both multiples of 16*T)
//   grid.x = (Mi/(TI*16)) * (Nj/(TJ*16)) blocks of C; grid.y*4 + wave = row slice.
//   AProd/BProd: Ctx row(int r) const; float val(const Ctx&, int col) const;
//   slab  : [nslices][Mi][Nj];  colsum (optional, COLSUM): [nslices][Mi] = sum_r A(r,i)
// ---------------------------------------------------------------------------------
template <int TI, int TJ, bool COLSUM, class AProd, class BProd>
__global__ __launch_bounds__(WS_THREADS) void tn_gemm(int nrows, int rows_per_slice, int Mi, int Nj,
                                                      AProd ap, BProd bp,
                                                      float* __restrict__ slab, float* __restrict__ colsum) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int njb = Nj / (TJ * 16);
    const int ib = blockIdx.x / njb, jb = blockIdx.x - ib * njb;
    const int i0 = ib * TI * 16, j0 = jb * TJ * 16;
    const int slice = blockIdx.y * 4 + wave;
    const int rbeg = slice * rows_per_slice;
    int rend = rbeg + rows_per_slice;
    if (rend > nrows) rend = nrows;
    f32x4 acc[TI][TJ];
    float csum[TI];
#pragma unroll
    for (int a = 0; a < TI; ++a) {
        csum[a] = 0.f;
#pragma unroll
        for (int b = 0; b < TJ; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll 2
    for (int r0 = rbeg; r0 < rend; r0 += 4) {
        const int r = r0 + q;
        const bool ok = r < rend;
        const int rc = ok ? r : rend - 1;
        const auto ca = ap.row(rc);
        const auto cb = bp.row(rc);
        float av[TI], bv[TJ];
#pragma unroll
        for (int a = 0; a < TI; ++a) {
            const float v = ap.val(ca, i0 + a * 16 + i);
            av[a] = ok ? v : 0.f;
        }
#pragma unroll
        for (int b = 0; b < TJ; ++b) bv[b] = bp.val(cb, j0 + b * 16 + i);
#pragma unroll
        for (int a = 0; a < TI; ++a) {
            if (COLSUM) csum[a] += av[a];
#pragma unroll
            for (int b = 0; b < TJ; ++b) acc[a][b] = mfma16(av[a], bv[b], acc[a][b]);
        }
    }
    float* out = slab + (size_t)slice * Mi * Nj;
#pragma unroll
    for (int a = 0; a < TI; ++a)
#pragma unroll
        for (int b = 0; b < TJ; ++b)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)
                out[(size_t)(i0 + a * 16 + q * 4 + reg) * Nj + j0 + b * 16 + i] = acc[a][b][reg];
    if (COLSUM && jb == 0) {
#pragma unroll
        for (int a = 0; a < TI; ++a) {
            float v = csum[a];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (q == 0) colsum[(size_t)slice * Mi + i0 + a * 16 + i] = v;
        }
    }
}

// out[e] = sum_s slab[s][e], fixed order.
__global__ void slab_reduce(const float* __restrict__ slab, int nslices, size_t n, float* __restrict__ out) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    float v = 0.f;
    for (int s = 0; s < nslices; ++s) v += slab[(size_t)s * n + e];
    out[e] = v;
}

}  // namespace cliora
