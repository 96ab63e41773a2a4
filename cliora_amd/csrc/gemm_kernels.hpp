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
#include <stdint.h>

namespace cliora {

using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Wave-wide sum / max, the butterfly lane ^ 32, ^ 16, ... ^ 1.  Rounds 1-2 ran it on __shfl_xor (ds_bpermute: six dependent
// LDS-crossbar round trips, 180 ns per reduction on MI355X); the same pairs in the same order on gfx950's v_permlane32_swap /
// v_permlane16_swap (the halves / the odd and even rows of the wave exchanged) and DPP row moves take 69 ns and give the SAME BITS
// (tools/ubench/wave_reduce_bench.hip, profiles/r03_ubench_wave_reduce.txt: 0 mismatches in 262 144 waves) -- every score, norm and
// softmax kernel is a chain of these.  All 64 lanes must be active (true at every call site: the callers branch per wave).
template <int CTRL, int BANK>
__device__ __forceinline__ float wave_dpp(float old, float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, 0xf, BANK, false));
}
template <class Op>
__device__ __forceinline__ float wave_butterfly(float v, Op op) {
    {   // lane ^ 32
        auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = op(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
    {   // lane ^ 16
        auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = op(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
    v = op(v, wave_dpp<0x128, 0xf>(v, v));                                                   // row_ror:8 = lane ^ 8
    { float t = wave_dpp<0x104, 0x5>(v, v); t = wave_dpp<0x114, 0xA>(t, v); v = op(v, t); }   // row_shl:4 | row_shr:4 by bank = lane ^ 4
    v = op(v, wave_dpp<0x4E, 0xf>(v, v));                                                    // quad_perm [2,3,0,1] = lane ^ 2
    v = op(v, wave_dpp<0xB1, 0xf>(v, v));                                                    // quad_perm [1,0,3,2] = lane ^ 1
    return v;
}
struct WaveAdd { __device__ __forceinline__ float operator()(float a, float b) const { return a + b; } };
struct WaveMax { __device__ __forceinline__ float operator()(float a, float b) const { return fmaxf(a, b); } };
__device__ __forceinline__ float wave_sum(float v) { return wave_butterfly(v, WaveAdd()); }
__device__ __forceinline__ float wave_max(float v) { return wave_butterfly(v, WaveMax()); }

// Lane maps of a 16-row operand tile.  The MFMA wants lane l to hold row (l & 15), k-piece (l >> 4) -- but gfx950 coalesces
// the addresses of CONSECUTIVE lanes, so a gather issued in that map is 64 separate 16-byte requests per instruction and runs at
// about half the rate of the same bytes fetched with four consecutive lanes on 64 contiguous bytes of one row
// (tools/ubench/gather_bench3.hip on MI355X: 9.1 -> 4.5 us for a 1 216-row level, 34.9 -> 23.1 us for 24 320 rows).
// So operand rows are FETCHED in the quad-coalesced map (lane l: row l >> 2, piece l & 3), run through the producer's ALU
// part there, and moved to the MFMA lanes with one ds_bpermute per dword (LDS crossbar, no LDS memory).
__device__ __forceinline__ int fetch_row_of(int lane) { return lane >> 2; }
__device__ __forceinline__ int fetch_piece_of(int lane) { return lane & 3; }
__device__ __forceinline__ int mfma_src_addr(int lane) { return 4 * (4 * (lane & 15) + (lane >> 4)); }   // byte address of the source lane
__device__ __forceinline__ float4 to_mfma_lanes(int src_addr, float4 v) {
    float4 r;
    r.x = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src_addr, __builtin_bit_cast(int, v.x)));
    r.y = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src_addr, __builtin_bit_cast(int, v.y)));
    r.z = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src_addr, __builtin_bit_cast(int, v.z)));
    r.w = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src_addr, __builtin_bit_cast(int, v.w)));
    return r;
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
static __global__ __launch_bounds__(WAVES * 64) void rows_gemm_ws(const float* __restrict__ W, int ldw, int Kseg, int nseg, int nrows,
                                                           AProd ap, Epi epi) {
    extern __shared__ __attribute__((aligned(16))) float lds_w[];
    constexpr int T = WAVES * 64;
    const int ldb = Kseg + WS_LDS_PAD;
    const int col0 = blockIdx.y * (CT * 16);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int li = fetch_row_of(lane), lq = fetch_piece_of(lane), psrc = mfma_src_addr(lane);   // fetch-side lane map
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

    if (nseg == 1) {
        // Single weight segment (the compose layer): no barrier in the loop, and the wave pipelines
        // ACROSS its tiles -- the row indices of the next tile are fetched during the second-to-last
        // stage of the current one and its first A stage during the last, so a tile starts with its
        // operands already in flight instead of paying two dependent load latencies.
        const int stride = gridDim.x * WAVES;
        int tile = blockIdx.x * WAVES + wave;
        if (tile >= ntiles) return;
        const int nstages = Kseg / (16 * SC);
        auto rowof = [&](int t) { const int r = t * 16 + li; return r < nrows ? r : nrows - 1; };   // clamp: computed, never stored
        auto ctx = ap.row(rowof(tile));
        auto ctxn = ctx;
        Raw cur[SC], nxt[SC];
#pragma unroll
        for (int j = 0; j < SC; ++j) cur[j] = ap.fetch(ctx, 16 * j + 4 * lq);
        while (true) {
            const int ntile = tile + stride;
            const bool has_next = ntile < ntiles;
            f32x4 acc[CT];
#pragma unroll
            for (int c = 0; c < CT; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int stg = 0; stg < nstages; ++stg) {
                const int ks = stg * 16 * SC;
                if (stg + 1 < nstages) {
#pragma unroll
                    for (int j = 0; j < SC; ++j) nxt[j] = ap.fetch(ctx, ks + 16 * (SC + j) + 4 * lq);
                    if (has_next && (stg + 2 == nstages || nstages == 1)) ctxn = ap.row(rowof(ntile));
                } else if (has_next) {
                    if (nstages == 1) ctxn = ap.row(rowof(ntile));
#pragma unroll
                    for (int j = 0; j < SC; ++j) nxt[j] = ap.fetch(ctxn, 16 * j + 4 * lq);
                }
#pragma unroll
                for (int j = 0; j < SC; ++j) {
                    const float4 af = ap.finish(ctx, cur[j]);
                    // optional side output of the A fragment (materialises x / dz for the weight-gradient
                    // GEMM): each column block writes its 1/gridDim.y share of the k-chunks
                    if (AProd::kSide && ((ks >> 4) + j) % (int)gridDim.y == (int)blockIdx.y) ap.side(ctx, ks + 16 * j + 4 * lq, af);
                    const float4 a = to_mfma_lanes(psrc, af);
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
#pragma unroll
                for (int j = 0; j < SC; ++j) cur[j] = nxt[j];
            }
            if (tile * 16 + i < nrows) {
                const auto rc = epi.row(tile * 16 + i);
#pragma unroll
                for (int c = 0; c < CT; ++c)
                    epi.store4(rc, col0 + c * 16 + 4 * q, make_float4(acc[c][0], acc[c][1], acc[c][2], acc[c][3]));
            }
            if (!has_next) break;
            ctx = ctxn;
            tile = ntile;
        }
        return;
    }

    for (int tile0 = blockIdx.x * WAVES; tile0 < ntiles; tile0 += gridDim.x * WAVES) {
        const int tile = tile0 + wave;
        const bool active = tile < ntiles;       // inactive waves still take part in the barriers
        int row = tile * 16 + li;
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
            for (int j = 0; j < SC; ++j) cur[j] = ap.fetch(ctx, kbase + 16 * j + 4 * lq);
            for (int ks = 0; ks < Kseg; ks += 16 * SC) {
                const bool more = ks + 16 * SC < Kseg;
                if (more) {
#pragma unroll
                    for (int j = 0; j < SC; ++j) nxt[j] = ap.fetch(ctx, kbase + ks + 16 * (SC + j) + 4 * lq);
                }
#pragma unroll
                for (int j = 0; j < SC; ++j) {
                    const float4 af = ap.finish(ctx, cur[j]);
                    // optional side output of the A fragment (materialises x / dz for the weight-gradient
                    // GEMM): each column block writes its 1/gridDim.y share of the k-chunks
                    if (AProd::kSide && ((ks >> 4) + j) % (int)gridDim.y == (int)blockIdx.y) ap.side(ctx, kbase + ks + 16 * j + 4 * lq, af);
                    const float4 a = to_mfma_lanes(psrc, af);
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
// Split-bf16 ("bf16x3") arithmetic: an fp32 value v is carried as hi = bf16(v) and lo = bf16(v - hi)
// (both round-to-nearest, v_cvt_pk_bf16_f32), and a product a*b is accumulated in fp32 as
// a_hi*b_hi + a_lo*b_hi + a_hi*b_lo on v_mfma_f32_16x16x32_bf16.  The dropped terms are below
// 2^-16 of |a*b| per product (fp32 itself keeps 2^-24), the accumulator stays fp32, and bf16 has the
// fp32 exponent range, so there is no scaling to manage.  Three such MFMAs cover k = 32 in 48 cycles
// where the fp32-input MFMA takes 256: the matrix core stops being the limit of the compose layer.
//
// v_mfma_f32_16x16x32_bf16 operand maps: A[i = lane&15][k = 8*(lane>>4) + j], B[k = 8*(lane>>4) + j][n = lane&15],
// j = 0..7 in the lane's four operand registers; D as for the fp32 form.  Which k of the 32-deep step a slot holds is
// free as long as both operands agree (the sum over k is unchanged): see split_weight_image.
// ---------------------------------------------------------------------------------
constexpr int WS3_PAD = 8;        // pad dwords per weight-image row: S = roundup32(K) + 8
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;

__device__ __forceinline__ f32x4 mfma32bf(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {   // a -> low half, b -> high half
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    const bf16x2 r = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(uint32_t, r);
}
// eight consecutive-k fp32 values -> hi and lo operand registers
__device__ __forceinline__ void split_bf16x8(const float4 a, const float4 b, u32x4& hi, u32x4& lo) {
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const uint32_t h = pack_bf16(v[2 * p], v[2 * p + 1]);
        hi[p] = h;
        lo[p] = pack_bf16(v[2 * p] - __uint_as_float(h << 16), v[2 * p + 1] - __uint_as_float(h & 0xffff0000u));
    }
}

// Weight image for rows_gemm_ws3: row j (one output column) = [Kp/2 dwords of hi pairs | Kp/2 dwords of lo pairs | 8 pad
// dwords], Kp = K rounded up to 32, zero beyond K.  Row stride S = Kp + 8 dwords: S/4 = 2 (mod 4), so in each of
// ds_read_b128's four 16-lane groups ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ... : eight rows of lane group g and the
// other eight rows of g+1, MI355X_MICROARCH.md LDS table) the 16-byte bank quads (2*row + g) mod 16 are all different.
// grid = (ceil(S/256), nrows, nmat); the matrices are [nrows][ldw] fp32 and are laid out one after the other.
constexpr int MAX_IMAGES = 12;
struct SplitImageTab { const float* src[MAX_IMAGES]; uint32_t* dst[MAX_IMAGES]; int nrows[MAX_IMAGES], ldw[MAX_IMAGES], K[MAX_IMAGES]; };
// grid = (ceil(max S / 256), max nrows, nmat); image m: row stride S = roundup32(K[m]) + WS3_PAD dwords
// one dword of the image: row `row`, dword p < S of image m
__device__ __forceinline__ void split_image_dword(const SplitImageTab& tab, int m, int row, int p) {
    const int K = tab.K[m], Kp = (K + 31) / 32 * 32, S = Kp + WS3_PAD;
    const float* W = tab.src[m] + (size_t)row * tab.ldw[m];
    uint32_t* img = tab.dst[m] + (size_t)row * S;
    const int half = Kp >> 1;
    if (p >= Kp) { img[p] = 0u; return; }
    // operand dword pr of the row = k-step pr/16, lane group g = (pr%16)/4, register q = pr%4, holding the k pair
    // 32*step + 4g + 2q (q < 2) or 32*step + 16 + 4g + 2(q-2): a lane's eight k are two runs of four, 16 apart, so
    // that the row operand's two 16-byte fetches per k-step each read 64 contiguous bytes of a row across g
    const int pr = p < half ? p : p - half;
    const int stp = pr >> 4, g = (pr >> 2) & 3, q = pr & 3;
    const int k0 = 32 * stp + 4 * g + (q < 2 ? 2 * q : 16 + 2 * (q - 2));
    const float v0 = k0 < K ? W[k0] : 0.f, v1 = k0 + 1 < K ? W[k0 + 1] : 0.f;
    const uint32_t h = pack_bf16(v0, v1);
    img[p] = p < half ? h : pack_bf16(v0 - __uint_as_float(h << 16), v1 - __uint_as_float(h & 0xffff0000u));
}
static __global__ __launch_bounds__(256) void split_weight_image(SplitImageTab tab) {
    const int m = blockIdx.z;
    const int S = (tab.K[m] + 31) / 32 * 32 + WS3_PAD;
    if ((int)blockIdx.y >= tab.nrows[m]) return;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= S) return;
    split_image_dword(tab, m, blockIdx.y, p);
}

// fp32 fragment image for rows_gemm_ksplit<.., FRAG = true>: float4 index ((ct * K/16 + ch) * 64 + lane) holds
// W[ct*16 + (lane & 15)][ch*16 + 4*(lane >> 4) .. +3].  Same table as split_weight_image; grid = (ceil(max nrows*K/4 / 256), 1, nmat).
__device__ __forceinline__ void frag_image_body(const SplitImageTab& tab, int m) {
    const int K = tab.K[m], nch = K >> 4;
    const size_t n4 = (size_t)tab.nrows[m] * K / 4;
    float4* img = reinterpret_cast<float4*>(tab.dst[m]);
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n4; e += (size_t)gridDim.x * 256) {
        const int lane = (int)(e & 63);
        const size_t blk = e >> 6;
        const int ch = (int)(blk % nch), ct = (int)(blk / nch);
        const float* src = tab.src[m] + (size_t)(ct * 16 + (lane & 15)) * tab.ldw[m] + ch * 16 + 4 * (lane >> 4);
        img[e] = *reinterpret_cast<const float4*>(src);
    }
}
static __global__ __launch_bounds__(256) void frag_weight_image(SplitImageTab tab) { frag_image_body(tab, blockIdx.z); }

// per-lane select of a small POD (row context) on a wave-uniform condition: v_cndmask, no branch
template <class T>
__device__ __forceinline__ T pick_pod(bool c, const T& a, const T& b) {
    static_assert(sizeof(T) % 4 == 0, "dword-sized contexts only");
    uint32_t wa[sizeof(T) / 4], wb[sizeof(T) / 4];
    __builtin_memcpy(wa, &a, sizeof(T));
    __builtin_memcpy(wb, &b, sizeof(T));
#pragma unroll
    for (size_t k = 0; k < sizeof(T) / 4; ++k) wa[k] = c ? wa[k] : wb[k];
    T r;
    __builtin_memcpy(&r, wa, sizeof(T));
    return r;
}


// ---------------------------------------------------------------------------------
// rows_gemm_ws3: rows_gemm_ws (single weight segment) in split-bf16 arithmetic (the span-region scorers of CLIORA in the default
// arithmetic mode: 24.8 GFLOP per step at c3, MFMA-bound on the fp32-input MFMA).
//   Wimg   : split_weight_image() of the [ncols][K] weight; block blockIdx.y's CT*16 image rows are one
//            contiguous piece and are copied to LDS as they are (LDS-DMA, lane-linear)
//   a wave walks 16-row tiles; per 32-deep k-step a lane fetches its row's 8 consecutive k through the
//   same AProd functors as the fp32 kernel (two fetches), splits them once, and issues 3*CT MFMAs.
//   PD k-steps of operand loads are in flight per wave (register ring); the ring runs on across the wave's
//   tiles, so the next tile's first loads are issued while the current tile's last steps compute.
// ---------------------------------------------------------------------------------
template <int CT, int WAVES, int PD, int K16, class AProd, class Epi>
static __global__ __launch_bounds__(WAVES * 64) void rows_gemm_ws3(const uint32_t* __restrict__ Wimg, int S_, int K_, int nrows, AProd ap, Epi epi) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_img[];
    constexpr int T = WAVES * 64;
    // K16 > 0: the reduction length K = 16*K16 is a compile-time constant -- the k-step loop unrolls completely and every
    // LDS / global offset, ring slot and "second run inside the row" test becomes an immediate (d = 400: K16 = 25);
    // K16 == 0: the same code with run-time K
    constexpr bool KS = K16 > 0;
    constexpr int UNROLL_STEPS = KS ? 64 : 1;      // the k-step loop: fully unrolled when K is a compile-time constant
    const int K = KS ? K16 * 16 : K_;
    const int S = KS ? (K16 + 1) / 2 * 32 + WS3_PAD : S_;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int li = fetch_row_of(lane), lg = fetch_piece_of(lane), psrc = mfma_src_addr(lane);   // fetch-side lane map
    const int Kp = S - WS3_PAD, half = Kp >> 1;
    const int col0 = blockIdx.y * (CT * 16);
    {
        const uint32_t* src = Wimg + (size_t)col0 * S;
        const int n16 = CT * 16 * S / 4;
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        for (int e0 = wv * 64; e0 < n16; e0 += T) {
            const int e = e0 + lane;
            if (e < n16)
                __builtin_amdgcn_global_load_lds((const void*)(src + (size_t)e * 4),
                                                 (__attribute__((address_space(3))) void*)(lds_img + e0 * 4), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    const int ntiles = (nrows + 15) >> 4;
    const int stride = gridDim.x * WAVES;
    int tile = blockIdx.x * WAVES + wave;
    if (tile >= ntiles) return;
    const int nsteps = Kp >> 5;
    const int nsteps_p = (nsteps + PD - 1) / PD * PD;     // the ring's slot of k-step s is s % PD in every tile
    int wfrag_off = i * S + 4 * g;
    const int gy = gridDim.y, by = blockIdx.y;
    using Raw = typename AProd::Raw;
    auto rowof = [&](int t) { const int r = t * 16 + li; return r < nrows ? r : nrows - 1; };   // clamp: computed, never stored
    // A lane's k at step s: 32s + 4g .. +3 and 32s + 16 + 4g .. +3 (the image's permutation; g = the lane's k-piece).  K is a multiple of 16, so
    // the first run is always inside the row and the second is inside for every lane or for none; when it is not, the
    // first run is fetched twice and the duplicate is discarded.  EVERY ring slot issues exactly two fetches per turn,
    // whatever the step: the loads then stand in a fixed order, the waits before a step count the loads issued after its
    // own (vmcnt is in order) and never drain the ring -- a fetch behind a branch forces a full drain at the join.
    Raw ra[PD][2];
    auto issue = [&](int slot, const decltype(ap.row(0))& c, int s) {
        const int k = 32 * s + 4 * lg;
        ra[slot][0] = ap.fetch(c, k);
        ra[slot][1] = ap.fetch(c, k + (32 * s + 16 < K ? 16 : 0));
    };
    auto ctx = ap.row(rowof(tile));
#pragma unroll
    for (int sl = 0; sl < PD; ++sl) issue(sl, ctx, sl < nsteps ? sl : 0);
    while (true) {
        const int ntile = tile + stride;
        const bool has_next = ntile < ntiles;
        const auto ctxn = ap.row(rowof(has_next ? ntile : tile));
        f32x4 acc[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        // the weight fragments do not change from tile to tile: keep the compiler from hoisting their LDS reads out of
        // the tile loop (hundreds of registers once the k-steps are unrolled)
        asm volatile("" : "+v"(wfrag_off));
        const uint32_t* wfrag = lds_img + wfrag_off;
        int side_turn = 0;                       // k-step modulo the number of column blocks (side output shared out)
#pragma unroll UNROLL_STEPS
        for (int base = 0; base < nsteps_p; base += PD) {
#pragma unroll
            for (int sl = 0; sl < PD; ++sl) {
                const int st = base + sl;
                if (st < nsteps) {
                    const bool second = 32 * st + 16 < K;
                    const float4 f0 = ap.finish(ctx, ra[sl][0]);
                    // beyond K this is a second copy of the first run: finite, and its weights in the image are zero
                    const float4 f1 = ap.finish(ctx, ra[sl][1]);
                    // optional side output of the fp32 row fragment (x / dz for the weight-gradient GEMM)
                    if (AProd::kSide && side_turn == by) {
                        const int k = 32 * st + 4 * lg;
                        ap.side(ctx, k, f0);
                        if (second) ap.side(ctx, k + 16, f1);
                    }
                    side_turn = side_turn + 1 == gy ? 0 : side_turn + 1;
                    const float4 a0 = to_mfma_lanes(psrc, f0), a1 = to_mfma_lanes(psrc, f1);
                    u32x4 xh, xl;
                    split_bf16x8(a0, a1, xh, xl);
                    u32x4 wh[CT], wl[CT];
#pragma unroll
                    for (int c = 0; c < CT; ++c) {
                        wh[c] = *reinterpret_cast<const u32x4*>(wfrag + c * 16 * S + 16 * st);
                        wl[c] = *reinterpret_cast<const u32x4*>(wfrag + c * 16 * S + 16 * st + half);
                    }
                    // term by term over the CT accumulators: dependent MFMAs are CT issues apart
#pragma unroll
                    for (int c = 0; c < CT; ++c) acc[c] = mfma32bf(wl[c], xh, acc[c]);
#pragma unroll
                    for (int c = 0; c < CT; ++c) acc[c] = mfma32bf(wh[c], xl, acc[c]);
#pragma unroll
                    for (int c = 0; c < CT; ++c) acc[c] = mfma32bf(wh[c], xh, acc[c]);
                }
                // refill the slot: this tile's step st+PD, or -- in the tile's last ring turn -- the next tile's step sl
                const int nst = st + PD;
                const bool in_cur = nst < nsteps;
                issue(sl, pick_pod(in_cur, ctx, ctxn), in_cur ? nst : (sl < nsteps ? sl : 0));
            }
        }
        if (tile * 16 + i < nrows) {
            const auto rc = epi.row(tile * 16 + i);
#pragma unroll
            for (int c = 0; c < CT; ++c)
                epi.store4(rc, col0 + c * 16 + 4 * g, make_float4(acc[c][0], acc[c][1], acc[c][2], acc[c][3]));
        }
        if (!has_next) break;
        ctx = ctxn;
        tile = ntile;
    }
}

// ---------------------------------------------------------------------------------
// rows_gemm_ksplit: same contract as rows_gemm_ws for SMALL row counts (the per-level cell
// projections and their backward: a few hundred rows, 39 dependent levels per pass).  There the
// weight-stationary kernel is all fixed cost (128 KiB of weights staged per workgroup for one
// tile per wave), so this one keeps no weights in LDS and instead maximises parallelism:
//   * a workgroup (4 waves) owns one (RT*16 rows) x (CT*16 cols) output block;
//   * the four waves split the reduction (k) range between them and each streams its A and W
//     fragments straight from global/L2 with a one-chunk register prefetch;
//   * partial accumulators meet in LDS and the epilogue is shared out over the waves.
// W: [ncols][K] row-major (K = whole reduction length, a multiple of 16); with FRAG, W is frag_weight_image() of
// that matrix: the 64 lanes' 16-byte weight fragments of one (column tile, 16-deep k-chunk) are 1 KiB contiguous, so a
// weight load is one fully used run of cache lines instead of sixteen 64-byte pieces 4*K bytes apart (the per-level
// launches are bound by how fast ONE CU can pull its block's operands through its texture-address path).
// ---------------------------------------------------------------------------------
template <int RT, int CT, bool FRAG, class AProd, class Epi>
static __global__ __launch_bounds__(256) void rows_gemm_ksplit(const float* __restrict__ W, int K, int nrg, int nrgp, int ncolblocks, int nrows,
                                                        AProd ap, Epi epi) {
    __shared__ float4 part[4][RT * CT][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 15, q = lane >> 4;
    const int li = fetch_row_of(lane), lq = fetch_piece_of(lane), psrc = mfma_src_addr(lane);   // fetch-side lane map
    const int cb = blockIdx.x / nrgp, rg = blockIdx.x - cb * nrgp;   // row group fastest: see the launcher (XCD L2 reuse)
    if (rg >= nrg) return;
    const int col0 = cb * (CT * 16);
    const int tile0 = rg * RT;
    const int nchunks = K >> 4;
    const int cbase = nchunks / 4, crem = nchunks % 4;
    const int ch0 = wave * cbase + min(wave, crem);
#ifdef CLIORA_DIAG_GEMMK                               // wrong-result timing diagnostic (tools/ab/anatomy.sh): 1/N of each wave's chunks
    const int nch = max(1, (cbase + (wave < crem ? 1 : 0)) / CLIORA_DIAG_GEMMK);
#else
    const int nch = cbase + (wave < crem ? 1 : 0);
#endif
    using Raw = typename AProd::Raw;
    using Ctx = decltype(ap.row(0));
    Ctx ctx[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) ctx[r] = ap.row(min((tile0 + r) * 16 + li, nrows - 1));
    const float* wrow = W + (size_t)(col0 + i) * K + 4 * q;
    f32x4 acc[RT][CT];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    // register ring of PD chunks: the loads of chunk n+PD are issued right after chunk n has been
    // consumed, so PD-1 chunks of MFMA work (not one) stand between a load and its use -- the
    // per-level launches are short and latency-bound, not bandwidth-bound
    constexpr int PD = 4;
    Raw ra[PD][RT];
    float4 rw[PD][CT];
    auto load = [&](int slot, int ch) {
        const int k = 16 * (ch0 + ch);
#pragma unroll
        for (int r = 0; r < RT; ++r) ra[slot][r] = ap.fetch(ctx[r], k + 4 * lq);
#pragma unroll
        for (int c = 0; c < CT; ++c)
            rw[slot][c] = FRAG ? reinterpret_cast<const float4*>(W)[((size_t)(cb * CT + c) * nchunks + ch0 + ch) * 64 + lane]
                               : *reinterpret_cast<const float4*>(wrow + (size_t)c * 16 * K + k);
    };
#pragma unroll
    for (int sl = 0; sl < PD; ++sl)
        if (sl < nch) load(sl, sl);
    for (int base = 0; base < nch; base += PD) {
#pragma unroll
        for (int sl = 0; sl < PD; ++sl) {
            if (base + sl < nch) {
                float4 a[RT];
#pragma unroll
                for (int r = 0; r < RT; ++r) a[r] = ap.finish(ctx[r], ra[sl][r]);
                if (AProd::kSide && (ch0 + base + sl) % ncolblocks == cb) {     // side output, shared out over the column blocks
#pragma unroll
                    for (int r = 0; r < RT; ++r) ap.side(ctx[r], 16 * (ch0 + base + sl) + 4 * lq, a[r]);
                }
#pragma unroll
                for (int r = 0; r < RT; ++r) a[r] = to_mfma_lanes(psrc, a[r]);
#pragma unroll
                for (int r = 0; r < RT; ++r)
#pragma unroll
                    for (int c = 0; c < CT; ++c) acc[r][c] = mfma16(rw[sl][c].x, a[r].x, acc[r][c]);
#pragma unroll
                for (int r = 0; r < RT; ++r)
#pragma unroll
                    for (int c = 0; c < CT; ++c) acc[r][c] = mfma16(rw[sl][c].y, a[r].y, acc[r][c]);
#pragma unroll
                for (int r = 0; r < RT; ++r)
#pragma unroll
                    for (int c = 0; c < CT; ++c) acc[r][c] = mfma16(rw[sl][c].z, a[r].z, acc[r][c]);
#pragma unroll
                for (int r = 0; r < RT; ++r)
#pragma unroll
                    for (int c = 0; c < CT; ++c) acc[r][c] = mfma16(rw[sl][c].w, a[r].w, acc[r][c]);
                if (base + sl + PD < nch) load(sl, base + sl + PD);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int c = 0; c < CT; ++c)
            part[wave][r * CT + c][lane] = make_float4(acc[r][c][0], acc[r][c][1], acc[r][c][2], acc[r][c][3]);
    __syncthreads();
    // wave w finishes output tiles w, w+4, ...: fixed summation order over the four k-slices
    for (int t = wave; t < RT * CT; t += 4) {
        const float4 p0 = part[0][t][lane], p1 = part[1][t][lane], p2 = part[2][t][lane], p3 = part[3][t][lane];
        const float4 v = make_float4(((p0.x + p1.x) + p2.x) + p3.x, ((p0.y + p1.y) + p2.y) + p3.y,
                                     ((p0.z + p1.z) + p2.z) + p3.z, ((p0.w + p1.w) + p2.w) + p3.w);
        const int r = t / CT, c = t - r * CT;
        const int row = (tile0 + r) * 16 + i;
        if (row < nrows) epi.store4(epi.row(row), col0 + c * 16 + 4 * q, v);
    }
}


// ---------------------------------------------------------------------------------
// rows_gemm_ksplit3 (round 4): rows_gemm_ksplit on split-bf16 operands (three v_mfma_f32_16x16x32_bf16 per product, fp32 accumulate:
// the arithmetic of the compose GEMMs) for the TreeLSTM's per-cell gate projections and their backward -- 10 of its 11 projection
// blocks feed sigmoid / tanh gates, not scores: [cells x 400] x [400 x 4000] per level on the fp32-input MFMA was 40 % of the c5
// step.  A 32-deep k-step of a 16 x 16 tile is 48 MFMA cycles instead of 256, so a workgroup owns 16 rows x 80 columns.
// Weights: frag_weight_image3 -- per (column tile, k-step) the 64 lanes' hi registers (1 KiB) then their lo registers, k in the
// operand order of split_weight_image, zero beyond K.  (Measured NOT to pay for DioraMLP's K = 1200 backward GEMM, which is
// ingest-bound at its size: profiles/r04_notes.md.)  Score projections (QL) stay on exact fp32 products.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ void frag_image3_body(const SplitImageTab& tab, int m) {
    const int K = tab.K[m], nst = (K + 31) >> 5;
    const size_t n = (size_t)(tab.nrows[m] >> 4) * nst * 512;      // dwords: 16 columns x 32 k per (tile, step)
    uint32_t* img = tab.dst[m];
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
        const int q = (int)(e & 3), lane = (int)((e >> 2) & 63), plane = (int)((e >> 8) & 1);
        const size_t blk = e >> 9;
        const int st = (int)(blk % nst), ct = (int)(blk / nst);
        const int g = lane >> 4;
        const int k0 = 32 * st + 4 * g + (q < 2 ? 2 * q : 16 + 2 * (q - 2));
        const float* src = tab.src[m] + (size_t)(ct * 16 + (lane & 15)) * tab.ldw[m];
        const float v0 = k0 < K ? src[k0] : 0.f, v1 = k0 + 1 < K ? src[k0 + 1] : 0.f;
        const uint32_t h = pack_bf16(v0, v1);
        img[e] = plane == 0 ? h : pack_bf16(v0 - __uint_as_float(h << 16), v1 - __uint_as_float(h & 0xffff0000u));
    }
}
static __global__ __launch_bounds__(256) void frag_weight_image3(SplitImageTab tab) { frag_image3_body(tab, blockIdx.z); }
// every weight image of a call in ONE launch (three launches in a row opened the forward: 17 us): blockIdx.z runs over the split images,
// then the fp32 fragment images, then the bf16 fragment images; grid = (256, 1, ns + nf + n3)
static __global__ __launch_bounds__(256) void weight_images_all(SplitImageTab split, int ns, SplitImageTab frag, int nf, SplitImageTab frag3) {
    int m = blockIdx.z;
    if (m < ns) {
        const int S = (split.K[m] + 31) / 32 * 32 + WS3_PAD;
        const size_t n = (size_t)split.nrows[m] * S;
        for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) split_image_dword(split, m, (int)(e / S), (int)(e % S));
        return;
    }
    m -= ns;
    if (m < nf) { frag_image_body(frag, m); return; }
    frag_image3_body(frag3, m - nf);
}

template <int CT, class AProd, class Epi>
static __global__ __launch_bounds__(256) void rows_gemm_ksplit3(const uint32_t* __restrict__ Wimg, int K, int nrg, int nrgp, int nrows,
                                                         AProd ap, Epi epi) {
    __shared__ float4 part[4][CT][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 15, q = lane >> 4;
    const int li = fetch_row_of(lane), lq = fetch_piece_of(lane), psrc = mfma_src_addr(lane);
    const int cb = blockIdx.x / nrgp, rg = blockIdx.x - cb * nrgp;   // row group fastest (XCD L2 reuse of the rows)
    if (rg >= nrg) return;
    const int col0 = cb * (CT * 16);
    const int nst = (K + 31) >> 5;
    const int sbase = nst / 4, srem = nst % 4;
    const int s0 = wave * sbase + min(wave, srem);
    const int ns = sbase + (wave < srem ? 1 : 0);
    using Ctx = decltype(ap.row(0));
    const Ctx ctx = ap.row(min(rg * 16 + li, nrows - 1));
    const u32x4* wfrag = reinterpret_cast<const u32x4*>(Wimg) + ((size_t)(cb * CT) * nst) * 128 + lane;
    f32x4 acc[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int PD = 3;
    float4 ra[PD][2];
    u32x4 rh[PD][CT], rl[PD][CT];
    auto load = [&](int slot, int s) {
        const int st = s0 + s;
        const int k = 32 * st + 4 * lq;
        ra[slot][0] = ap.finish(ctx, ap.fetch(ctx, k));
        ra[slot][1] = 32 * st + 16 < K ? ap.finish(ctx, ap.fetch(ctx, k + 16)) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            rh[slot][c] = wfrag[((size_t)c * nst + st) * 128];
            rl[slot][c] = wfrag[((size_t)c * nst + st) * 128 + 64];
        }
    };
#pragma unroll
    for (int sl = 0; sl < PD; ++sl)
        if (sl < ns) load(sl, sl);
    for (int base = 0; base < ns; base += PD) {
#pragma unroll
        for (int sl = 0; sl < PD; ++sl) {
            if (base + sl < ns) {
                u32x4 xh, xl;
                split_bf16x8(to_mfma_lanes(psrc, ra[sl][0]), to_mfma_lanes(psrc, ra[sl][1]), xh, xl);
#pragma unroll
                for (int c = 0; c < CT; ++c) acc[c] = mfma32bf(rl[sl][c], xh, acc[c]);
#pragma unroll
                for (int c = 0; c < CT; ++c) acc[c] = mfma32bf(rh[sl][c], xl, acc[c]);
#pragma unroll
                for (int c = 0; c < CT; ++c) acc[c] = mfma32bf(rh[sl][c], xh, acc[c]);
                if (base + sl + PD < ns) load(sl, base + sl + PD);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < CT; ++c) part[wave][c][lane] = make_float4(acc[c][0], acc[c][1], acc[c][2], acc[c][3]);
    __syncthreads();
    for (int t = wave; t < CT; t += 4) {                 // fixed summation order over the four k-slices
        const float4 p0 = part[0][t][lane], p1 = part[1][t][lane], p2 = part[2][t][lane], p3 = part[3][t][lane];
        const float4 v = make_float4(((p0.x + p1.x) + p2.x) + p3.x, ((p0.y + p1.y) + p2.y) + p3.y,
                                     ((p0.z + p1.z) + p2.z) + p3.z, ((p0.w + p1.w) + p2.w) + p3.w);
        const int row = rg * 16 + i;
        if (row < nrows) epi.store4(epi.row(row), col0 + t * 16 + 4 * q, v);
    }
}

// rows_gemm_ksplit3x (round 5): rows_gemm_ksplit3 with RT x CT sixteen-wide tiles per workgroup -- the form for DioraMLP's per-level
// projection GEMMs, which sit on two limits at once (profiles/r05_notes.md section 14): what a CU can pull from L2 (a 16 x 16 block streams
// 150 KB of rows and fragments for one tile) and the fp32-input MFMA (16 x 16 x 4 in 32 cycles).  Split-bf16 products take the MFMA limit
// away (3 x 16 cycles per 16 x 16 x 32), a square tile halves the bytes per output; each alone lost.  Ragged last column block (nt column
// tiles in all); every output element is summed in the same order for any tile shape (four k-slices, fixed tree).
template <int RT, int CT, class AProd, class Epi>
static __global__ __launch_bounds__(256) void rows_gemm_ksplit3x(const uint32_t* __restrict__ Wimg, int K, int nrg, int nrgp, int nt, int nrows,
                                                          AProd ap, Epi epi) {
    __shared__ float4 part[4][RT * CT][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 15, q = lane >> 4;
    const int li = fetch_row_of(lane), lq = fetch_piece_of(lane), psrc = mfma_src_addr(lane);
    const int cb = blockIdx.x / nrgp, rg = blockIdx.x - cb * nrgp;   // row group fastest (XCD L2 reuse of the rows)
    if (rg >= nrg) return;
    const int ct0 = cb * CT;
    const int nct = min(CT, nt - ct0);                    // workgroup-uniform
    const int nst = (K + 31) >> 5;
    const int sbase = nst / 4, srem = nst % 4;
    const int s0 = wave * sbase + min(wave, srem);
    const int ns = sbase + (wave < srem ? 1 : 0);
    using Ctx = decltype(ap.row(0));
    Ctx ctx[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) ctx[r] = ap.row(min((rg * RT + r) * 16 + li, nrows - 1));
    const u32x4* wfrag = reinterpret_cast<const u32x4*>(Wimg) + ((size_t)ct0 * nst) * 128 + lane;
    f32x4 acc[RT][CT];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int PD = RT * CT > 4 ? 2 : 3;
    float4 ra[PD][RT][2];
    u32x4 rh[PD][CT], rl[PD][CT];
    auto load = [&](int slot, int s) {
        const int st = s0 + s;
        const int k = 32 * st + 4 * lq;
        const bool second = 32 * st + 16 < K;
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            ra[slot][r][0] = ap.finish(ctx[r], ap.fetch(ctx[r], k));
            ra[slot][r][1] = second ? ap.finish(ctx[r], ap.fetch(ctx[r], k + 16)) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int c = 0; c < CT; ++c)
            if (c < nct) {
                rh[slot][c] = wfrag[((size_t)c * nst + st) * 128];
                rl[slot][c] = wfrag[((size_t)c * nst + st) * 128 + 64];
            }
    };
#pragma unroll
    for (int sl = 0; sl < PD; ++sl)
        if (sl < ns) load(sl, sl);
    for (int base = 0; base < ns; base += PD) {
#pragma unroll
        for (int sl = 0; sl < PD; ++sl) {
            if (base + sl < ns) {
                u32x4 xh[RT], xl[RT];
#pragma unroll
                for (int r = 0; r < RT; ++r) split_bf16x8(to_mfma_lanes(psrc, ra[sl][r][0]), to_mfma_lanes(psrc, ra[sl][r][1]), xh[r], xl[r]);
#pragma unroll
                for (int r = 0; r < RT; ++r)
#pragma unroll
                    for (int c = 0; c < CT; ++c) if (c < nct) acc[r][c] = mfma32bf(rl[sl][c], xh[r], acc[r][c]);
#pragma unroll
                for (int r = 0; r < RT; ++r)
#pragma unroll
                    for (int c = 0; c < CT; ++c) if (c < nct) acc[r][c] = mfma32bf(rh[sl][c], xl[r], acc[r][c]);
#pragma unroll
                for (int r = 0; r < RT; ++r)
#pragma unroll
                    for (int c = 0; c < CT; ++c) if (c < nct) acc[r][c] = mfma32bf(rh[sl][c], xh[r], acc[r][c]);
                if (base + sl + PD < ns) load(sl, base + sl + PD);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int c = 0; c < CT; ++c) part[wave][r * CT + c][lane] = make_float4(acc[r][c][0], acc[r][c][1], acc[r][c][2], acc[r][c][3]);
    __syncthreads();
    for (int t = wave; t < RT * CT; t += 4) {            // fixed summation order over the four k-slices
        const int r = t / CT, c = t - r * CT;
        if (c >= nct) continue;
        const float4 p0 = part[0][t][lane], p1 = part[1][t][lane], p2 = part[2][t][lane], p3 = part[3][t][lane];
        const float4 v = make_float4(((p0.x + p1.x) + p2.x) + p3.x, ((p0.y + p1.y) + p2.y) + p3.y,
                                     ((p0.z + p1.z) + p2.z) + p3.z, ((p0.w + p1.w) + p2.w) + p3.w);
        const int row = (rg * RT + r) * 16 + i;
        if (row < nrows) epi.store4(epi.row(row), (ct0 + c) * 16 + 4 * q, v);
    }
}

// ---------------------------------------------------------------------------------
// tn_gemm:  C[i][j] = sum_r A(r,i) B(r,j),  i < Mi, j < Nj  (both multiples of 16*T)
//   grid.x = (Mi/(TI*16)) * (Nj/(TJ*16)) blocks of C; grid.y*4 + wave = row slice.
//   AProd/BProd: Ctx row(int r) const; float val(const Ctx&, int col) const;
//   slab  : [nslices][Mi][Nj];  colsum (optional, COLSUM): [nslices][Mi] = sum_r A(r,i)
// ---------------------------------------------------------------------------------
template <int TI, int TJ, bool COLSUM, class AProd, class BProd>
static __global__ __launch_bounds__(WS_THREADS) void tn_gemm(int nrows, int rows_per_slice, int Mi, int Nj,
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

// ---------------------------------------------------------------------------------
// tn_gemm_direct: the same product for SMALL row counts (configs[0]-sized plans: a few hundred to a few thousand rows) in ONE
// launch: a workgroup of eight waves owns one 16 x 16 block of C, the waves split the rows, their partial blocks meet in LDS in wave
// order and wave 0 writes (or accumulates into) the result -- no slab, no slab_reduce launches (tn_gemm + two reductions were three
// launches per weight gradient, and a step of such a plan is bound by the host's launch rate).  Exact fp32 MFMA.
// ---------------------------------------------------------------------------------
constexpr int TND_WAVES = 8;
template <bool COLSUM, class AProd, class BProd>
static __global__ __launch_bounds__(TND_WAVES * 64) void tn_gemm_direct(int nrows, int Mi, int Nj, AProd ap, BProd bp, float* __restrict__ out,
                                                                    float* __restrict__ colsum_out, int accumulate) {
    __shared__ float red[TND_WAVES - 1][4][64];
    __shared__ float redc[TND_WAVES - 1][16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int njb = Nj / 16;
    const int ib = blockIdx.x / njb, jb = blockIdx.x - ib * njb;
    const int i0 = ib * 16, j0 = jb * 16;
    const int rps = ((nrows + TND_WAVES - 1) / TND_WAVES + 3) / 4 * 4;
    const int rbeg = wave * rps;
    const int rend = min(rbeg + rps, nrows);
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    float csum = 0.f;
#pragma unroll 8
    for (int r0 = rbeg; r0 < rend; r0 += 4) {
        const int r = r0 + q;
        const bool ok = r < rend;
        const int rc = ok ? r : rend - 1;
        const float a = ap.val(ap.row(rc), i0 + i);
        const float b = bp.val(bp.row(rc), j0 + i);
        const float av = ok ? a : 0.f;
        if (COLSUM) csum += av;
        acc = mfma16(av, b, acc);
    }
    if (COLSUM) {
        csum += __shfl_xor(csum, 16);
        csum += __shfl_xor(csum, 32);
    }
    if (wave > 0) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) red[wave - 1][reg][lane] = acc[reg];
        if (COLSUM && q == 0) redc[wave - 1][i] = csum;
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int w = 0; w < TND_WAVES - 1; ++w) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) acc[reg] += red[w][reg][lane];
        if (COLSUM && q == 0) csum += redc[w][i];
    }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        float* o = out + (size_t)(i0 + q * 4 + reg) * Nj + j0 + i;
        *o = accumulate ? *o + acc[reg] : acc[reg];
    }
    if (COLSUM && jb == 0 && q == 0) {
        float* o = colsum_out + i0 + i;
        *o = accumulate ? *o + csum : csum;
    }
}

// ---------------------------------------------------------------------------------
// tn_gemm_dma:  C[i][j] = sum_r A[r][i] B[r][j]  for the big weight gradient (rows = span pairs).
//   A (dz) and B (x) are dense row-major matrices the compose kernels materialised.  One
//   workgroup (4 waves, one per SIMD) owns ALL Mi rows of C x one block of up to NJT*16 columns
//   for one slice of the pair rows.  16 pair rows per stage are brought into LDS by LDS-DMA
//   (global_load_lds, 1 KiB per wave instruction, no VGPR staging, no VALU), double buffered:
//   stage s+1 lands while stage s feeds the MFMAs.  The accumulators take most of the register
//   file (NIT*NJT 16x16 tiles per wave) -- that is what bounds the HBM traffic per pair row at
//   Mi + nkb*... bytes; see DESIGN.md.
//   NIT = i-tiles per wave (every wave runs the same count: a short share re-computes its last
//   tile and skips the duplicate store -- the workgroup waits for its longest wave anyway);
//   NJT = j-tiles per column block (short blocks likewise).
//   slab: [nslices][Mi][Nj];  colsum (COLSUM): [nslices][Mi] = sum_r A[r][i]
// ---------------------------------------------------------------------------------
constexpr int TN_RS = 16;       // pair rows per stage (4 MFMA k-steps)
constexpr int TN_NP = 12;       // LDS-DMA pieces per wave per stage, upper bound: (Mi/16 + NJT + 3) / 4

template <int NIT, int NJT, bool COLSUM>
static __global__ __launch_bounds__(256) void tn_gemm_dma(const float* __restrict__ A, const float* __restrict__ B, int nrows,
                                                   int rows_per_slice, int Mi, int Nj, int nkb,
                                                   float* __restrict__ slab, float* __restrict__ colsum) {
    extern __shared__ __attribute__((aligned(16))) float lds_t[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, q = lane >> 4;
    const int kb = blockIdx.x % nkb, slice = blockIdx.x / nkb;
    const int NTI = Mi >> 4, NTJ = Nj >> 4;
    const int jbase = NTJ / nkb, jrem = NTJ % nkb;
    const int jt0 = kb * jbase + min(kb, jrem);
    const int njt = jbase + (kb < jrem ? 1 : 0);            // j-tiles of this block (<= NJT)
    const int ibase = NTI / 4, irem = NTI % 4;
    const int it0 = wave * ibase + min(wave, irem);
    const int nit = max(1, ibase + (wave < irem ? 1 : 0));  // i-tiles of this wave (<= NIT)
    const bool has_tiles = ibase + (wave < irem ? 1 : 0) > 0;
    const int ldA = Mi, ldX = NJT * 16;                     // Mi % 32 == 16 and ldX % 32 == 16 read conflict-free
    const int UA = TN_RS * (Mi >> 2), UX = TN_RS * (ldX >> 2);   // 16-byte units per stage: A region, X region
    const int bufsz = TN_RS * (ldA + ldX);
    const int x4 = ldX >> 2, xv4 = njt * 4;                 // float4 columns per X row: allocated, valid

    const int rbeg = slice * rows_per_slice;
    const int rend = min(nrows, rbeg + rows_per_slice);
    const int nstages = rend > rbeg ? (rend - rbeg + TN_RS - 1) / TN_RS : 0;

    f32x4 acc[NIT][NJT];
#pragma unroll
    for (int a = 0; a < NIT; ++a)
#pragma unroll
        for (int b = 0; b < NJT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    float cs0 = 0.f, cs1 = 0.f;

    // LDS-DMA of one stage: the A region is the contiguous image of rows [r0, r0+16) of A; the X
    // region holds columns [jt0*16, +ldX) of the same rows of B.  A stage is NTI + NJT pieces of
    // 1 KiB (64 lanes x 16 B), each entirely A or entirely X; wave w moves pieces w, w+4, ...
    // The lane -> (row, column) map of every piece is stage-invariant and computed once here.
    // Rows past the slice end are re-read from the last valid row (finite data) and masked out of
    // the MFMAs below.
    const int npieces = (UA + UX) >> 6;
    int p_rr[TN_NP], p_off[TN_NP];          // row within the stage, element offset inside that row (incl. column base)
#pragma unroll
    for (int k = 0; k < TN_NP; ++k) {
        const int e = (wave + 4 * k) * 64 + lane;
        if (e < UA) {
            p_rr[k] = e / (Mi >> 2);
            p_off[k] = 4 * (e - p_rr[k] * (Mi >> 2));
        } else {
            const int f = e - UA;
            p_rr[k] = f / x4;
            p_off[k] = jt0 * 16 + 4 * min(f - p_rr[k] * x4, xv4 - 1);
        }
    }
    auto issue = [&](int stage) {
        const int r0 = rbeg + stage * TN_RS;
        const int rmax = rend - 1 - r0;      // last valid row of the stage
        float* buf = lds_t + (stage & 1) * bufsz;
#pragma unroll
        for (int k = 0; k < TN_NP; ++k) {
            const int piece = wave + 4 * k;
            if (piece < npieces) {
                const bool isA = piece * 64 < UA;
                const float* base = isA ? A : B;
                const int ld = isA ? Mi : Nj;
                const float* src = base + (size_t)(r0 + min(p_rr[k], rmax)) * ld + p_off[k];
                __builtin_amdgcn_global_load_lds((const void*)src, (__attribute__((address_space(3))) void*)(buf + piece * 256), 16, 0, 0);
            }
        }
    };

    // LDS offsets of this lane's fragments: tile indices are clamped to the wave's / block's last tile
    int aoff[NIT], xoff[NJT];
#pragma unroll
    for (int t = 0; t < NIT; ++t) aoff[t] = (min(it0, NTI - 1) + min(t, nit - 1)) * 16 + i;
#pragma unroll
    for (int u = 0; u < NJT; ++u) xoff[u] = TN_RS * ldA + min(u, njt - 1) * 16 + i;

    if (nstages > 0) issue(0);
    for (int st = 0; st < nstages; ++st) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                 // stage st has landed for every wave; buffer (st+1)&1 is free again
        if (st + 1 < nstages) issue(st + 1);
        const float* cur = lds_t + (st & 1) * bufsz;
        const int r0 = rbeg + st * TN_RS;
        // fragments of k-step ks+1 are read from LDS while the MFMAs of k-step ks issue (register
        // double buffer; sched_barrier pins "reads first" so the LDS latency is not exposed per tile)
        float av[2][NIT], bv[2][NJT];
        auto frag = [&](int ks, float (&a)[NIT], float (&b)[NJT]) {
            const int rr = 4 * ks + q;
            const bool valid = r0 + rr < rend;
#pragma unroll
            for (int u = 0; u < NJT; ++u) b[u] = cur[rr * ldX + xoff[u]];
#pragma unroll
            for (int t = 0; t < NIT; ++t) { const float v = cur[rr * ldA + aoff[t]]; a[t] = valid ? v : 0.f; }
        };
        frag(0, av[0], bv[0]);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks < 3) frag(ks + 1, av[(ks + 1) & 1], bv[(ks + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < NIT; ++t)
#pragma unroll
                for (int u = 0; u < NJT; ++u) acc[t][u] = mfma16(av[ks & 1][t], bv[ks & 1][u], acc[t][u]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (COLSUM && kb == 0) {                 // column sums of A (bias gradient), branch-free
            const int c0 = min(tid, Mi - 1), c1 = min(tid + 256, Mi - 1);
#pragma unroll
            for (int rr = 0; rr < TN_RS; ++rr) {
                const float m = (r0 + rr < rend) ? 1.f : 0.f;
                cs0 = fmaf(m, cur[rr * ldA + c0], cs0);
                cs1 = fmaf(m, cur[rr * ldA + c1], cs1);
            }
        }
    }
    float* out = slab + (size_t)slice * Mi * Nj;
    if (has_tiles) {
#pragma unroll
        for (int t = 0; t < NIT; ++t)
            if (t < nit)
#pragma unroll
                for (int u = 0; u < NJT; ++u)
                    if (u < njt)
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg)
                            out[(size_t)((it0 + t) * 16 + q * 4 + reg) * Nj + (jt0 + u) * 16 + i] = acc[t][u][reg];
    }
    if (COLSUM && kb == 0) {
        if (tid < Mi) colsum[(size_t)slice * Mi + tid] = cs0;
        if (tid + 256 < Mi) colsum[(size_t)slice * Mi + tid + 256] = cs1;
    }
}

// ---------------------------------------------------------------------------------
// tn_gemm_dma3: tn_gemm_dma in split-bf16 arithmetic (see the bf16x3 note above).
//   Same blocks, slices, slab and LDS-DMA staging, but a stage is 32 pair rows = ONE k-step of
//   v_mfma_f32_16x16x32_bf16.  The fp32 stage in LDS is turned into operands on the way to the registers:
//   a lane's eight k of a fragment are rows g, 4+g, ..., 28+g of the stage (g = lane>>4; any assignment works
//   as long as both operands use it, and this one keeps the 32-lane halves of a ds_read_b32 on rows that are
//   one apart = 16 banks apart for the 400- and 144-float row lengths), read as eight dwords and split once.
//   The nine column fragments are kept for the whole step; row fragment t+1 is assembled while the 3*NJT
//   MFMAs of row t issue.
// ---------------------------------------------------------------------------------
constexpr int TN3_RS = 32;
constexpr int TN3_NP = 20;      // LDS-DMA pieces per wave per stage, upper bound: (Mi + NJT*16) / 32
template <int NIT, int NJT, bool COLSUM>
static __global__ __launch_bounds__(256) void tn_gemm_dma3(const float* __restrict__ A, const float* __restrict__ B, int nrows,
                                                    int rows_per_slice, int nslices, int Mi, int Nj, int nkb,
                                                    float* __restrict__ slab, float* __restrict__ colsum) {
    extern __shared__ __attribute__((aligned(16))) float lds_t[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;
    // workgroups are dealt round-robin over the 8 XCDs by block id: the nkb column blocks of a slice re-read the same
    // A rows, so they get ids that are equal mod 8 and close together -- one XCD's L2 then serves the re-reads
    // (grid = 8 * nkb * ceil(nslices / 8); ids whose slice does not exist leave at once)
    const int xcd = blockIdx.x & 7, wq = blockIdx.x >> 3;
    const int kb = wq % nkb, slice = (wq / nkb) * 8 + xcd;
    if (slice >= nslices) return;
    const int NTI = Mi >> 4, NTJ = Nj >> 4;
    const int jbase = NTJ / nkb, jrem = NTJ % nkb;
    const int jt0 = kb * jbase + min(kb, jrem);
    const int njt = jbase + (kb < jrem ? 1 : 0);            // j-tiles of this block (<= NJT)
    const int ibase = NTI / 4, irem = NTI % 4;
    const int it0 = wave * ibase + min(wave, irem);
    const int nit = max(1, ibase + (wave < irem ? 1 : 0));  // i-tiles of this wave (<= NIT)
    const bool has_tiles = ibase + (wave < irem ? 1 : 0) > 0;
    const int ldA = Mi, ldX = NJT * 16;
    const int UA = TN3_RS * (Mi >> 2), UX = TN3_RS * (ldX >> 2);
    const int bufsz = TN3_RS * (ldA + ldX);
    const int x4 = ldX >> 2, xv4 = njt * 4;

    const int rbeg = slice * rows_per_slice;
    const int rend = min(nrows, rbeg + rows_per_slice);
    const int nstages = rend > rbeg ? (rend - rbeg + TN3_RS - 1) / TN3_RS : 0;

    f32x4 acc[NIT][NJT];
#pragma unroll
    for (int a = 0; a < NIT; ++a)
#pragma unroll
        for (int b = 0; b < NJT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    float cs0 = 0.f, cs1 = 0.f;
    // Column sums of A (the bias gradient): when some column block has a spare j-tile slot (its njt < NJT; at d = 400 the
    // blocks hold 9, 8, 8 of the 25 tiles) that slot's B fragment is set to ones and the MFMAs the slot costs anyway
    // deliver sum_r A[r][i] in its accumulator; otherwise block 0 adds the stage's rows up from LDS.
    const int kb_ones = jrem > 0 && jbase < NJT ? jrem : (jbase < NJT ? 0 : -1);    // first block with njt == jbase < NJT
    const bool ones_here = COLSUM && kb == kb_ones;

    const int npieces = (UA + UX) >> 6;
    int p_rr[TN3_NP], p_off[TN3_NP];
#pragma unroll
    for (int k = 0; k < TN3_NP; ++k) {
        const int e = (wave + 4 * k) * 64 + lane;
        if (e < UA) {
            p_rr[k] = e / (Mi >> 2);
            p_off[k] = 4 * (e - p_rr[k] * (Mi >> 2));
        } else {
            const int f = e - UA;
            p_rr[k] = f / x4;
            p_off[k] = jt0 * 16 + 4 * min(f - p_rr[k] * x4, xv4 - 1);
        }
    }
    auto issue = [&](int stage) {
        const int r0 = rbeg + stage * TN3_RS;
        const int rmax = rend - 1 - r0;      // last valid row of the stage: rows past it re-read it (finite) and are masked below
        float* buf = lds_t + (stage & 1) * bufsz;
#pragma unroll
        for (int k = 0; k < TN3_NP; ++k) {
            const int piece = wave + 4 * k;
            if (piece < npieces) {
                const bool isA = piece * 64 < UA;
                const float* base = isA ? A : B;
                const int ld = isA ? Mi : Nj;
                const float* src = base + (size_t)(r0 + min(p_rr[k], rmax)) * ld + p_off[k];
                __builtin_amdgcn_global_load_lds((const void*)src, (__attribute__((address_space(3))) void*)(buf + piece * 256), 16, 0, 0);
            }
        }
    };

    int aoff[NIT];
#pragma unroll
    for (int t = 0; t < NIT; ++t) aoff[t] = g * ldA + (min(it0, NTI - 1) + min(t, nit - 1)) * 16 + i;

    if (nstages > 0) issue(0);
    for (int st = 0; st < nstages; ++st) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                 // stage st has landed for every wave; buffer (st+1)&1 is free again
        if (st + 1 < nstages) issue(st + 1);
        const float* cur = lds_t + (st & 1) * bufsz;
        const int r0 = rbeg + st * TN3_RS;
        const int nvalid = rend - r0;    // rows of this stage that exist (>= 32 except in the slice's last stage)
        auto frag = [&](int off, int ld, u32x4& hi, u32x4& lo) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = cur[off + 4 * j * ld];
            split_bf16x8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), hi, lo);
        };
        // the slice's last stage may be short: its missing A rows (DMA re-read the last valid row there) are zeroed
        // in LDS, so the operand reads below need no mask
        if (nvalid < TN3_RS) {
            float* cw = lds_t + (st & 1) * bufsz;
            for (int e = nvalid * ldA + tid; e < TN3_RS * ldA; e += 256) cw[e] = 0.f;
        }
        // the NJT column fragments are the same for all four waves: each wave splits its share once and leaves the
        // operand registers of all 64 lanes in LDS ([fragment][hi|lo][lane] x 16 B, read back conflict-free)
        u32x4* bconv = reinterpret_cast<u32x4*>(lds_t + 2 * bufsz);
        for (int u = wave; u < NJT; u += 4) {
            u32x4 h, l;
            frag(TN3_RS * ldA + g * ldX + min(u, njt - 1) * 16 + i, ldX, h, l);
            bconv[(2 * u) * 64 + lane] = h;
            bconv[(2 * u + 1) * 64 + lane] = l;
        }
        __syncthreads();
        u32x4 bh[NJT], bl[NJT];
#pragma unroll
        for (int u = 0; u < NJT; ++u) { bh[u] = bconv[(2 * u) * 64 + lane]; bl[u] = bconv[(2 * u + 1) * 64 + lane]; }
        if (ones_here) {                 // bf16 1.0 = 0x3F80 in all eight k of the spare fragment
            bh[NJT - 1] = u32x4{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
            bl[NJT - 1] = u32x4{0u, 0u, 0u, 0u};
        }
        // three-deep pipeline over the wave's row tiles: LDS reads of tile t+2, split of tile t+1 and the 3*NJT MFMAs of
        // tile t are issued interleaved (sched_group_barrier), so neither the LDS latency nor the split is exposed
        float raw[2][8];
        u32x4 ah[2], al[2];
        auto loadraw = [&](int t, float (&r)[8]) {
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] = cur[aoff[t] + 4 * j * ldA];
        };
        auto conv = [&](const float (&r)[8], u32x4& hi, u32x4& lo) {
            split_bf16x8(make_float4(r[0], r[1], r[2], r[3]), make_float4(r[4], r[5], r[6], r[7]), hi, lo);
        };
        loadraw(0, raw[0]);
        conv(raw[0], ah[0], al[0]);
        if (NIT > 1) loadraw(1, raw[1]);
#pragma unroll
        for (int t = 0; t < NIT; ++t) {
            if (t + 1 < NIT) conv(raw[(t + 1) & 1], ah[(t + 1) & 1], al[(t + 1) & 1]);
            if (t + 2 < NIT) loadraw(t + 2, raw[t & 1]);
#pragma unroll
            for (int u = 0; u < NJT; ++u) acc[t][u] = mfma32bf(al[t & 1], bh[u], acc[t][u]);
#pragma unroll
            for (int u = 0; u < NJT; ++u) acc[t][u] = mfma32bf(ah[t & 1], bl[u], acc[t][u]);
#pragma unroll
            for (int u = 0; u < NJT; ++u) acc[t][u] = mfma32bf(ah[t & 1], bh[u], acc[t][u]);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 LDS read
                __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);   // 1 VALU
            }
#pragma unroll
            for (int k = 8; k < 3 * NJT; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
            }
        }
        if (COLSUM && kb_ones < 0 && kb == 0) {  // column sums of A (bias gradient), branch-free
            const int c0 = min(tid, Mi - 1), c1 = min(tid + 256, Mi - 1);
#pragma unroll 8
            for (int rr = 0; rr < TN3_RS; ++rr) {
                const float m = (rr < nvalid) ? 1.f : 0.f;
                cs0 = fmaf(m, cur[rr * ldA + c0], cs0);
                cs1 = fmaf(m, cur[rr * ldA + c1], cs1);
            }
        }
    }
    float* out = slab + (size_t)slice * Mi * Nj;
    if (has_tiles) {
#pragma unroll
        for (int t = 0; t < NIT; ++t)
            if (t < nit)
#pragma unroll
                for (int u = 0; u < NJT; ++u)
                    if (u < njt)
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg)
                            out[(size_t)((it0 + t) * 16 + g * 4 + reg) * Nj + (jt0 + u) * 16 + i] = acc[t][u][reg];
    }
    if (COLSUM && kb_ones < 0 && kb == 0) {
        if (tid < Mi) colsum[(size_t)slice * Mi + tid] = cs0;
        if (tid + 256 < Mi) colsum[(size_t)slice * Mi + tid + 256] = cs1;
    }
    if (ones_here && has_tiles && i == 0) {      // every column of the ones tile holds the same sums: lane column 0 writes
#pragma unroll
        for (int t = 0; t < NIT; ++t)
            if (t < nit)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) colsum[(size_t)slice * Mi + (it0 + t) * 16 + g * 4 + reg] = acc[t][NJT - 1][reg];
    }
}

// tn_gemm_dma3x: tn_gemm_dma3 with EIGHT waves (two per SIMD).  With one wave per SIMD everything a stage needs besides its 189 MFMAs
// -- 17 LDS-DMA pieces (100-250 issue cycles each, the price measured with round 3's rows-stationary kernel, profiles/r03_rows_stationary.txt), the operand reads and splits -- sits in
// that wave's own instruction stream: 8 500 cycles per 32-row stage for 3 024 cycles of MFMA (28 % MFMA-busy by PMC).  Here wave w takes
// the i-tiles of wave w & 3 of the four-wave kernel and HALF of the block's j-tiles (NJW = ceil(NJT / 2); the second half has a spare
// slot, which carries the ones-tile of the bias gradient in block 0), so that one wave's DMA issue and operand forming run under its
// SIMD partner's MFMAs.  Every accumulator tile is still produced by one wave in the same stage and k order: the slab is bitwise the
// four-wave kernel's.
template <int NIT, int NJT, int NJW, bool COLSUM>
static __global__ __launch_bounds__(512) void tn_gemm_dma3x(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb, int nrows,
                                                     int rows_per_slice, int nslices, int Mi, int Nj, int nkb,
                                                     float* __restrict__ slab, float* __restrict__ colsum,
                                                     int slice2 = 0x7fffffff, int nrows2 = 0, long long shift2 = 0,
                                                     int rm_hi = 0, int rm_C = 0, int rm_off = 0, long long a_prob_stride = 0) {
    // gridDim.y > 1: that many problems in one launch -- problem y reads A + y * a_prob_stride (the projection blocks of dPI: column
    // offsets of one row-major matrix) against the same B, and its slab / column-sum slices are interleaved [slice][problem] so that
    // ONE reduction finishes all of them into consecutive outputs (round 4: three launches + three reductions ended the step)
    // rm_hi > 0: the rows are the cells [rm_off, rm_off + rm_hi) of every sentence's chart (rm_C cells per sentence) of BOTH
    // matrices -- row r is chart row (r / rm_hi) * rm_C + rm_off + r % rm_hi: the projections' weight gradient over the levels
    // that are final (round 4: the fp32 element-load tn_gemm took 233 us for them at d 400 and ended the step)
    // slices >= slice2 walk a SECOND row range of the same matrices: nrows2 rows starting shift2 rows further down (the two ends of the
    // pair rows around the part whose weight gradient started early: one launch, one slab, one reduction for both)
    static_assert(2 * NJW > NJT, "the second half of the j-tiles needs a spare slot for the ones-tile");
    extern __shared__ __attribute__((aligned(16))) float lds_t[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int iw = wave & 3, jw = wave >> 2;
    const int i = lane & 15, g = lane >> 4;
    const int xcd = blockIdx.x & 7, wq = blockIdx.x >> 3;
    const int kb = wq % nkb, slice = (wq / nkb) * 8 + xcd;
    if (slice >= nslices) return;
    const int NTI = Mi >> 4, NTJ = Nj >> 4;
    const int jbase = NTJ / nkb, jrem = NTJ % nkb;
    const int jt0 = kb * jbase + min(kb, jrem);
    const int njt = jbase + (kb < jrem ? 1 : 0);            // j-tiles of this block (<= NJT)
    const int ju0 = jw * NJW;                               // first j-tile (block-relative) of this wave
    const int njw = max(0, min(njt - ju0, NJW));            // j-tiles of this wave
    const int ibase = NTI / 4, irem = NTI % 4;
    const int it0 = iw * ibase + min(iw, irem);
    const int nit = max(1, ibase + (iw < irem ? 1 : 0));    // i-tiles of this wave (<= NIT)
    const bool has_tiles = ibase + (iw < irem ? 1 : 0) > 0;
    const int ldA = Mi, ldX = NJT * 16;
    const int UA = TN3_RS * (Mi >> 2), UX = TN3_RS * (ldX >> 2);
    const int bufsz = TN3_RS * (ldA + ldX);
    const int x4 = ldX >> 2, xv4 = njt * 4;

    const bool second = slice >= slice2;
    if (second) { A += shift2 * lda; B += shift2 * ldb; }
    const int prob = blockIdx.y, nprob = gridDim.y;
    A += (size_t)prob * a_prob_stride;
    const int rbeg = (second ? slice - slice2 : slice) * rows_per_slice;
    const int rend = min(second ? nrows2 : nrows, rbeg + rows_per_slice);
    const int nstages = rend > rbeg ? (rend - rbeg + TN3_RS - 1) / TN3_RS : 0;

    f32x4 acc[NIT][NJW];
#pragma unroll
    for (int a = 0; a < NIT; ++a)
#pragma unroll
        for (int b = 0; b < NJW; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool ones_here = COLSUM && kb == 0 && jw == 1;    // the spare slot NJW - 1 of the second half

    constexpr int NPW = (TN3_NP + 1) / 2;                   // pieces per wave per stage, upper bound
    const int npieces = (UA + UX) >> 6;
    const float rm_inv = rm_hi ? 1.0f / (float)rm_hi : 0.f;   // (r + 0.5) * rm_inv floors to r / rm_hi exactly for r < 4e6
    int p_rr[NPW], p_off[NPW];
#pragma unroll
    for (int k = 0; k < NPW; ++k) {
        const int e = (wave + 8 * k) * 64 + lane;
        if (e < UA) {
            p_rr[k] = e / (Mi >> 2);
            p_off[k] = 4 * (e - p_rr[k] * (Mi >> 2));
        } else {
            const int f = e - UA;
            p_rr[k] = f / x4;
            p_off[k] = jt0 * 16 + 4 * min(f - p_rr[k] * x4, xv4 - 1);
        }
    }
    // pieces [k0, k1) of this wave's share of a stage
    auto issue = [&](int stage, int k0, int k1) {
        const int r0 = rbeg + stage * TN3_RS;
        const int rmax = rend - 1 - r0;
        float* buf = lds_t + (stage & 1) * bufsz;
#pragma unroll
        for (int k = 0; k < NPW; ++k) {
            if (k < k0 || k >= k1) continue;
            const int piece = wave + 8 * k;
            if (piece < npieces) {
                const bool isA = piece * 64 < UA;
                const float* base = isA ? A : B;
                const int ld = isA ? lda : ldb;        // row strides of the two operands (Mi / Nj when they are plain matrices)
                int row = r0 + min(p_rr[k], rmax);
                if (rm_hi) { const int sb_ = (int)(((float)row + 0.5f) * rm_inv); row = sb_ * rm_C + rm_off + (row - sb_ * rm_hi); }
                const float* src = base + (size_t)row * ld + p_off[k];
                __builtin_amdgcn_global_load_lds((const void*)src, (__attribute__((address_space(3))) void*)(buf + piece * 256), 16, 0, 0);
            }
        }
    };

    int aoff[NIT];
#pragma unroll
    for (int t = 0; t < NIT; ++t) aoff[t] = g * ldA + (min(it0, NTI - 1) + min(t, nit - 1)) * 16 + i;

    if (nstages > 0) issue(0, 0, NPW);
    constexpr int PPT = (NPW + NIT - 1) / NIT;           // pieces of the NEXT stage issued beside each row tile's MFMAs
    for (int st = 0; st < nstages; ++st) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                 // stage st has landed for every wave; buffer (st+1)&1 is free again
        const bool more = st + 1 < nstages;      // the next stage's LDS-DMA is issued piecewise inside the MFMA loop below (its issue cost,
                                                 // 100-250 cycles a piece, is what a stage waited for when all pieces went out up front)
        const float* cur = lds_t + (st & 1) * bufsz;
        const int r0 = rbeg + st * TN3_RS;
        const int nvalid = rend - r0;
        auto frag = [&](int off, int ld, u32x4& hi, u32x4& lo) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = cur[off + 4 * j * ld];
            split_bf16x8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), hi, lo);
        };
        if (nvalid < TN3_RS) {
            float* cw = lds_t + (st & 1) * bufsz;
            for (int e = nvalid * ldA + tid; e < TN3_RS * ldA; e += 512) cw[e] = 0.f;
        }
        // the NJT column fragments: each wave splits its share once and leaves the operand registers of all 64 lanes in LDS
        u32x4* bconv = reinterpret_cast<u32x4*>(lds_t + 2 * bufsz);
        for (int u = wave; u < NJT; u += 8) {
            u32x4 h, l;
            frag(TN3_RS * ldA + g * ldX + min(u, njt - 1) * 16 + i, ldX, h, l);
            bconv[(2 * u) * 64 + lane] = h;
            bconv[(2 * u + 1) * 64 + lane] = l;
        }
        __syncthreads();
        u32x4 bh[NJW], bl[NJW];
#pragma unroll
        for (int u = 0; u < NJW; ++u) {
            const int uu = min(ju0 + u, NJT - 1);
            bh[u] = bconv[(2 * uu) * 64 + lane]; bl[u] = bconv[(2 * uu + 1) * 64 + lane];
        }
        if (ones_here) {                 // bf16 1.0 = 0x3F80 in all eight k of the spare fragment
            bh[NJW - 1] = u32x4{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
            bl[NJW - 1] = u32x4{0u, 0u, 0u, 0u};
        }
        float raw[2][8];
        u32x4 ah[2], al[2];
        auto loadraw = [&](int t, float (&r)[8]) {
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] = cur[aoff[t] + 4 * j * ldA];
        };
        auto conv = [&](const float (&r)[8], u32x4& hi, u32x4& lo) {
            split_bf16x8(make_float4(r[0], r[1], r[2], r[3]), make_float4(r[4], r[5], r[6], r[7]), hi, lo);
        };
        loadraw(0, raw[0]);
        conv(raw[0], ah[0], al[0]);
        if (NIT > 1) loadraw(1, raw[1]);
#pragma unroll
        for (int t = 0; t < NIT; ++t) {
            if (t + 1 < NIT) conv(raw[(t + 1) & 1], ah[(t + 1) & 1], al[(t + 1) & 1]);
            if (t + 2 < NIT) loadraw(t + 2, raw[t & 1]);
            if (more) issue(st + 1, PPT * t, PPT * t + PPT);
#pragma unroll
            for (int u = 0; u < NJW; ++u) acc[t][u] = mfma32bf(al[t & 1], bh[u], acc[t][u]);
#pragma unroll
            for (int u = 0; u < NJW; ++u) acc[t][u] = mfma32bf(ah[t & 1], bl[u], acc[t][u]);
#pragma unroll
            for (int u = 0; u < NJW; ++u) acc[t][u] = mfma32bf(ah[t & 1], bh[u], acc[t][u]);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 LDS read
                __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);   // 1 VALU
            }
#pragma unroll
            for (int k = 8; k < 3 * NJW; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                if (k - 8 < PPT) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);     // 1 VMEM: a DMA piece of the next stage
            }
        }
    }
    float* out = slab + ((size_t)slice * nprob + prob) * Mi * Nj;
    if (has_tiles) {
#pragma unroll
        for (int t = 0; t < NIT; ++t)
            if (t < nit)
#pragma unroll
                for (int u = 0; u < NJW; ++u)
                    if (u < njw)
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg)
                            out[(size_t)((it0 + t) * 16 + g * 4 + reg) * Nj + (jt0 + ju0 + u) * 16 + i] = acc[t][u][reg];
    }
    if (ones_here && has_tiles && i == 0) {      // every column of the ones tile holds the same sums: lane column 0 writes
#pragma unroll
        for (int t = 0; t < NIT; ++t)
            if (t < nit)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) colsum[((size_t)slice * nprob + prob) * Mi + (it0 + t) * 16 + g * 4 + reg] = acc[t][NJW - 1][reg];
    }
}

// out[e] (+)= sum_s slab[s][e] and -- the blocks past the matrix, same launch -- cs_out[i] (+)= sum_s csl[s][i].  Fixed order: a
// workgroup owns 64 * V consecutive elements (V = 4: one float4 per thread); its four waves take the slices s = w, w + 4, ... in
// order, eight loads in flight each (the first form of this kernel was one thread per element walking the slices one dependent
// 4-byte load after the other: 56 us for the 48 slices of a d = 400 pair-row weight gradient, 30 us for its 400-element bias
// gradient, both on the step's critical tail), and the four partial sums meet in LDS as (w0 + w1) + (w2 + w3).
template <int V>
static __global__ __launch_bounds__(256) void slab_reduce_k(const float* __restrict__ slab, int nslices, size_t n, float* __restrict__ out,
                                                            int accumulate, const float* __restrict__ csl, size_t ncs,
                                                            float* __restrict__ cs_out, unsigned nblk_main) {
    __shared__ float sh[3][64 * V];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const bool main = blockIdx.x < nblk_main;
    const float* src = main ? slab : csl;
    float* dst = main ? out : cs_out;
    const size_t len = main ? n : ncs;
    const size_t e = ((size_t)(main ? blockIdx.x : blockIdx.x - nblk_main) * 64 + lane) * V;
    const bool ok = e < len;
    float acc[V];
#pragma unroll
    for (int v = 0; v < V; ++v) acc[v] = 0.f;
    if (ok) {
        for (int s0 = w; s0 < nslices; s0 += 32) {
            float t[8][V];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int s = s0 + 4 * j;
                const float* p = src + (size_t)min(s, nslices - 1) * len + e;
                if constexpr (V == 4) {
                    const float4 q = *reinterpret_cast<const float4*>(p);
                    t[j][0] = q.x; t[j][1] = q.y; t[j][2] = q.z; t[j][3] = q.w;
                } else {
                    t[j][0] = p[0];
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (s0 + 4 * j < nslices)
#pragma unroll
                    for (int v = 0; v < V; ++v) acc[v] += t[j][v];
        }
    }
    if (w > 0)
#pragma unroll
        for (int v = 0; v < V; ++v) sh[w - 1][lane * V + v] = acc[v];
    __syncthreads();
    if (w != 0 || !ok) return;
#pragma unroll
    for (int v = 0; v < V; ++v) {
        float r = (acc[v] + sh[0][lane * V + v]) + (sh[1][lane * V + v] + sh[2][lane * V + v]);
        if (accumulate) r += dst[e + v];
        acc[v] = r;
    }
    if constexpr (V == 4) *reinterpret_cast<float4*>(dst + e) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    else dst[e] = acc[0];
}

// one launch for a slab and (optionally) its column-sum slab; false on a launch error (hipGetLastError holds it)
static inline void launch_slab_reduce(hipStream_t st, const float* slab, int nslices, size_t n, float* out, int accumulate,
                                      const float* csl = nullptr, size_t ncs = 0, float* cs_out = nullptr) {
    if (!cs_out) ncs = 0;
    const bool vec = (n % 4 == 0) && (ncs % 4 == 0) && ((reinterpret_cast<uintptr_t>(slab) | reinterpret_cast<uintptr_t>(out) |
                                                        reinterpret_cast<uintptr_t>(csl) | reinterpret_cast<uintptr_t>(cs_out)) % 16 == 0);
    const int per = vec ? 256 : 64;
    const unsigned nb = (unsigned)((n + per - 1) / per), nc = (unsigned)((ncs + per - 1) / per);
    if (vec) hipLaunchKernelGGL(slab_reduce_k<4>, dim3(nb + nc), dim3(256), 0, st, slab, nslices, n, out, accumulate, csl, ncs, cs_out, nb);
    else hipLaunchKernelGGL(slab_reduce_k<1>, dim3(nb + nc), dim3(256), 0, st, slab, nslices, n, out, accumulate, csl, ncs, cs_out, nb);
}

}  // namespace cliora
