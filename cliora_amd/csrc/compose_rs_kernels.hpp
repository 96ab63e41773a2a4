// Rows-stationary forward compose of one chart level (gfx950): the big-level counterpart of level_compose_fwd.
//
// level_compose_fwd is weight-stationary: a workgroup owns one block of 80 output columns (its split-bf16 W2 image stays in LDS)
// and every one of the five column blocks gathers every operand row again -- 16 kB per pair row through L2.  For levels of a few
// thousand pair rows that is the optimum (the weights are re-streamed by nobody); for levels of 10^4 rows and more the gathers
// are what the launch waits for (66-79 GB/s per CU of ingest, 16.5 us per tile at L = 40 against 7 at L = 20).
//
// Here the ROWS stay and the weights stream.  A wave gathers the two operand rows of its 16-row tile ONCE, forms
// x = relu(PL(a) + PR(b)) and keeps it as MFMA operands in registers (13 k-steps x 8 registers at K = 400); the workgroup then
// walks the column blocks, the W2 image of a block passing through LDS in thirds of the reduction (two buffers, LDS-DMA of
// third t+1 under the MFMAs of third t; one barrier per third).  Per pair row: 3.2 kB gathered instead of 16 kB; per workgroup
// task (8 waves x 16 rows): the whole 665 kB image from L2, the same bytes for every workgroup of the chip.
//
// Same tile order, same arithmetic: a task is (TG cell tiles) x (part s of SP of the split range), the 8 waves are dealt
// WPG = 8 / TG to a cell tile, wave r takes split n0 + r -- ONE split per wave (the operand registers hold one tile), so the
// launch geometry must have ceil(N / SP) <= WPG (plan.cpp: compose_geom_rs).  k-steps accumulate in the order of
// level_compose_fwd, the epilogue and the LDS tree over the waves of a cell tile are its code: with the same (TG, SP) the two
// kernels agree to the BIT (tests/test_gpu_rows_stationary.py).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "level_kernels.hpp"

namespace cliora {

constexpr int RS_STRIDE = 168;        // dwords per image row of a stage buffer: 2 x 80 (hi | lo, or 160 fp32) + 8; 168/4 = 2 (mod 4): conflict-free
constexpr int RS_HALF = 80;           // dword offset of the lo pieces inside a row
constexpr int RS_NT = 3;              // thirds of the reduction
// LDS: two stage buffers + the reduction slots
template <int CT> constexpr size_t rs_lds_bytes() { return (size_t)2 * CT * 16 * RS_STRIDE * 4 + (size_t)LC_SLOTS * CT * 64 * sizeof(float4); }

// One third of a column block's image -> a stage buffer.  LDS-DMA writes lane-linear, so every lane fills the 16-byte slots
// q = it * 512 + thread of the linear buffer; slot q is piece w = q % 42 of row q / 42: w < 20 the hi piece (k-steps s0 ..),
// 20 <= w < 40 the lo piece, the rest padding (exact-fp32 mode: 40 slots of fp32 per row, 32 per k-step).  What a slot reads does
// not depend on the block or the third except for a uniform base, so each thread keeps its RS_NIT slot descriptors in registers
// (source offset in dwords | piece index << 16 | valid << 31): the per-lane address arithmetic was half of the 0.85 us a third's
// issue cost (tools/rs_trace.py).
constexpr int RS_NIT = 7;             // ceil(80 rows * 42 slots / 512 threads)
template <int CT, bool F32>
__device__ __forceinline__ void rs_describe(int S, int half, uint32_t (&desc)[RS_NIT]) {
    constexpr int SLOTS_ROW = RS_STRIDE / 4, NSLOT = CT * 16 * SLOTS_ROW;
    static_assert((NSLOT + 511) / 512 <= RS_NIT, "descriptor count");
#pragma unroll
    for (int it = 0; it < RS_NIT; ++it) {
        const int q = it * 512 + (int)threadIdx.x;
        const int c = q / SLOTS_ROW, w = q - c * SLOTS_ROW;
        uint32_t d;
        if constexpr (F32) {
            const bool ok = q < NSLOT && w < 2 * (RS_HALF / 4);
            d = (uint32_t)(c * S + 4 * w) | ((uint32_t)w << 16) | (ok ? 0x80000000u : 0u);
        } else {
            const bool lo = w >= RS_HALF / 4;
            const int wp = lo ? w - RS_HALF / 4 : w;
            const bool ok = q < NSLOT && w < 2 * (RS_HALF / 4);
            d = (uint32_t)(c * S + (lo ? half : 0) + 4 * wp) | ((uint32_t)wp << 16) | (ok ? 0x80000000u : 0u);
        }
        desc[it] = d;
    }
}
template <int CT, bool F32>
__device__ __forceinline__ void rs_stage(const uint32_t* __restrict__ Wimg, int S, int K, int col0, int s0, int ns, uint32_t* buf,
                                         const uint32_t (&desc)[RS_NIT], int wave, int it0 = 0, int it1 = RS_NIT) {
    const uint32_t* base = Wimg + (size_t)col0 * S + (F32 ? 32 : 16) * s0;
    const int lim = (F32 ? 8 : 4) * ns;
#pragma unroll
    for (int it = 0; it < RS_NIT; ++it) {
        if (it < it0 || it >= it1) continue;          // a wave that computes issues its pieces two per k-step, beside the MFMAs
        const uint32_t d = desc[it];
        const int wp = (int)((d >> 16) & 0xffu);
        bool ok = (d >> 31) != 0 && wp < lim;
        if constexpr (F32) ok = ok && 32 * s0 + 4 * wp < K;
        if (ok)
            __builtin_amdgcn_global_load_lds((const void*)(base + (d & 0xffffu)),
                                             (__attribute__((address_space(3))) void*)(buf + (it * 512 + wave * 64) * 4), 16, 0, 0);
    }
}

// Weight fragments of one k-step in registers (split mode: hi and lo halves; exact-fp32 mode: the two runs of four k), read one
// k-step AHEAD of the MFMAs that consume them.  Left to itself hipcc reads one fragment, waits for it, issues one to three
// MFMAs and reads the next (rs_trace: 0.55 us per k-step where the MFMAs alone take 0.2): the scheduling barrier between two
// k-steps keeps "read step s+1, multiply step s" together.  The wait for step s sits BEFORE the reads of step s+1: with both in flight
// (20 > the 15 that lgkmcnt can count) hipcc falls back to lgkmcnt(0) after the reads, which exposes their latency again.
template <int CT> struct WFrag { u32x4 a[CT], b[CT]; };
template <int CT, bool F32>
__device__ __forceinline__ WFrag<CT> rs_read_frag(const uint32_t* wimg, int i, int g, int st, bool second) {
    WFrag<CT> w;
    if constexpr (F32) {
        const uint32_t* wf = wimg + i * RS_STRIDE + 4 * g + 32 * st;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            w.a[c] = *reinterpret_cast<const u32x4*>(wf + c * 16 * RS_STRIDE);
            w.b[c] = second ? *reinterpret_cast<const u32x4*>(wf + c * 16 * RS_STRIDE + 16) : u32x4{0u, 0u, 0u, 0u};
        }
    } else {
        const uint32_t* wf = wimg + i * RS_STRIDE + 4 * g + 16 * st;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            w.a[c] = *reinterpret_cast<const u32x4*>(wf + c * 16 * RS_STRIDE);
            w.b[c] = *reinterpret_cast<const u32x4*>(wf + c * 16 * RS_STRIDE + RS_HALF);
        }
    }
    return w;
}
// the MFMAs of one k-step, in the order of kstep_mfma (level_kernels.hpp)
template <int CT, bool F32>
__device__ __forceinline__ void rs_mfma_step(const WFrag<CT>& w, const StepOperand& x, bool second, f32x4 (&acc)[CT]) {
    if constexpr (F32) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int c = 0; c < CT; ++c) acc[c] = mfma16(__uint_as_float(w.a[c][q]), __uint_as_float(x.h[q]), acc[c]);
        if (second) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int c = 0; c < CT; ++c) acc[c] = mfma16(__uint_as_float(w.b[c][q]), __uint_as_float(x.l[q]), acc[c]);
        }
    } else {
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[c] = mfma32bf(w.b[c], x.h, acc[c]);
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[c] = mfma32bf(w.a[c], x.l, acc[c]);
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[c] = mfma32bf(w.a[c], x.h, acc[c]);
    }
}

template <int CT, int K16, bool F32>
__global__ __launch_bounds__(512) void level_compose_fwd_rs(const uint32_t* __restrict__ Wimg, int S_, PairLevel lv, const float* __restrict__ PA, int lda,
                                                            const float* __restrict__ PB, int ldb, const float* __restrict__ bias,
                                                            const float* __restrict__ Pp, int TG, int SP, int ntask, int ncb, float* __restrict__ HP,
                                                            size_t hp_stride, int Dp, uint32_t* __restrict__ ymask, float* __restrict__ Y,
                                                            unsigned long long* __restrict__ trace) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_rs[];
#ifdef CLIORA_RS_STAMPS     // diagnostic build: 100 MHz stamps of each workgroup's first task (tools/rs_trace.py)
    int stamp_k = 0;
#define RS_STAMP() do { if (trace && threadIdx.x == 0 && task == (int)blockIdx.x && stamp_k < 32) trace[(size_t)blockIdx.x * 32 + stamp_k++] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define RS_STAMP() do {} while (0)
#endif
    constexpr int WAVES = 8, K = K16 * 16;
    constexpr int NSTEPS = (K + 31) / 32;
    constexpr int T0 = (NSTEPS + 2) / 3, T1 = (NSTEPS - T0 + 1) / 2;          // 13 -> 5, 4, 4
    constexpr int SB[RS_NT + 1] = {0, T0, T0 + T1, NSTEPS};
    static_assert(T0 * 16 <= RS_HALF && T0 * 32 <= RS_STRIDE - 8, "a third must fit a stage row");
    const int S = F32 ? K : S_;                                              // row stride of the global image in dwords
    const int half = F32 ? 0 : (S_ - WS3_PAD) >> 1;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 15, g = lane >> 4;
    const int li = fetch_row_of(lane), lg = fetch_piece_of(lane), psrc = mfma_src_addr(lane);
    uint32_t* buf0 = lds_rs;
    uint32_t* buf1 = lds_rs + CT * 16 * RS_STRIDE;
    float4* red = reinterpret_cast<float4*>(lds_rs + 2 * CT * 16 * RS_STRIDE);     // [LC_SLOTS][CT][64]
    const int WPG = WAVES / TG;
    const int j = wave / WPG, r = wave - j * WPG;
    const int G = (lv.ncell + 15) >> 4;
    const int Ns = (lv.N + SP - 1) / SP;
    const int nstage = RS_NT * ncb;
    uint32_t desc[RS_NIT];
    rs_describe<CT, F32>(S, half, desc);

    for (int task = blockIdx.x; task < ntask; task += gridDim.x) {
        RS_STAMP();
        // every wave is past its last read of both buffers (barrier at the end of the previous task): the first third may land
        rs_stage<CT, F32>(Wimg, S, K, 0, SB[0], SB[1] - SB[0], buf0, desc, wave);
        const int gg = task / SP, s = task - gg * SP;
        const int gt = gg * TG + j;
        const bool have = gt < G;
        const int n0 = s * Ns, n1 = min(lv.N, n0 + Ns);
        const int n = n0 + r;
        const bool work = have && n < n1;
        // ---- the tile's operand: gathered once, kept as MFMA operands ----
        StepOperand xop[NSTEPS];
        {
            const int t = min(min(gt, G - 1) * 16 + li, lv.ncell - 1);        // clamp: computed, masked out by p = 0 and never stored
            const int b = t / lv.Lc, p = t - b * lv.Lc;
            const int idx = p * lv.N + (work ? n : 0);
            const float* pa = PA + ((size_t)b * lv.C + lv.pa[idx]) * lda;
            const float* pb = PB + ((size_t)b * lv.C + lv.pb[idx]) * ldb;
            if (work) {
                // GB k-steps of raw rows per batch, TWO batches in flight: the next batch is on its way while this one is formed
                constexpr int GB = 4, NBATCH = (NSTEPS + GB - 1) / GB;
                float4 ua[2][GB][2], ub[2][GB][2];
                auto fetch = [&](int slot, int base) {
#pragma unroll
                    for (int q = 0; q < GB; ++q) {
                        const int st = base + q < NSTEPS ? base + q : NSTEPS - 1;
                        const int k = 32 * st + 4 * lg;
                        const int k2 = k + (32 * st + 16 < K ? 16 : 0);
                        ua[slot][q][0] = ld4(pa + k); ub[slot][q][0] = ld4(pb + k);
                        ua[slot][q][1] = ld4(pa + k2); ub[slot][q][1] = ld4(pb + k2);
                    }
                };
                fetch(0, 0);
#pragma unroll
                for (int bi = 0; bi < NBATCH; ++bi) {
                    if (bi + 1 < NBATCH) fetch((bi + 1) & 1, (bi + 1) * GB);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int q = 0; q < GB; ++q) {
                        if (bi * GB + q < NSTEPS) {
                            const float4 (&A)[2] = ua[bi & 1][q];
                            const float4 (&Bv)[2] = ub[bi & 1][q];
                            const float4 f0 = make_float4(fmaxf(A[0].x + Bv[0].x, 0.f), fmaxf(A[0].y + Bv[0].y, 0.f),
                                                          fmaxf(A[0].z + Bv[0].z, 0.f), fmaxf(A[0].w + Bv[0].w, 0.f));
                            const float4 f1 = make_float4(fmaxf(A[1].x + Bv[1].x, 0.f), fmaxf(A[1].y + Bv[1].y, 0.f),
                                                          fmaxf(A[1].z + Bv[1].z, 0.f), fmaxf(A[1].w + Bv[1].w, 0.f));
                            xop[bi * GB + q] = make_operand<F32>(psrc, f0, f1);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int st = 0; st < NSTEPS; ++st) xop[st] = StepOperand{u32x4{0u, 0u, 0u, 0u}, u32x4{0u, 0u, 0u, 0u}};
            }
        }
        RS_STAMP();
        // MFMA-lane view of the tile: row i is target cell ti, pair row prow
        const int ti = gt * 16 + i;
        const bool ok = work && ti < lv.ncell;
        const size_t prow = (size_t)lv.rowbase + (size_t)min(max(ti, 0), lv.ncell - 1) * lv.N + (work ? n : 0);
        const float pn = ok ? Pp[prow] : 0.f;

        for (int cb = 0; cb < ncb; ++cb) {
            const int col0 = cb * (CT * 16);
            f32x4 acc[CT];
#pragma unroll
            for (int c = 0; c < CT; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t3 = 0; t3 < RS_NT; ++t3) {
                const int sidx = cb * RS_NT + t3;
                // third sidx has landed everywhere, and every wave is done with third sidx - 1: its buffer takes third sidx + 1
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                const bool more = sidx + 1 < nstage;
                const int nt3 = (t3 + 1) % RS_NT, ncol0 = t3 + 1 < RS_NT ? col0 : col0 + CT * 16;
                uint32_t* nbuf = (sidx & 1) ? buf0 : buf1;
                // the next third's LDS-DMA: an idle wave issues its share now; a wave that computes spreads it over its k-steps (an LDS-DMA
                // costs 100-250 issue cycles; up front that was 0.3-0.85 us of every third, beside the MFMAs it hides)
                if (more && !work) rs_stage<CT, F32>(Wimg, S, K, ncol0, SB[nt3], SB[nt3 + 1] - SB[nt3], nbuf, desc, wave);
                const uint32_t* wimg = (sidx & 1) ? buf1 : buf0;
                if (work) {
                    WFrag<CT> w = rs_read_frag<CT, F32>(wimg, i, g, 0, 32 * SB[t3] + 16 < K);
#pragma unroll
                    for (int st = SB[t3]; st < SB[t3 + 1]; ++st) {
                        WFrag<CT> wn = w;
                        __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0): w has landed; at most one k-step of reads is ever in flight
                        if (st + 1 < SB[t3 + 1]) wn = rs_read_frag<CT, F32>(wimg, i, g, st + 1 - SB[t3], 32 * (st + 1) + 16 < K);
                        __builtin_amdgcn_sched_barrier(0);
                        if (more) rs_stage<CT, F32>(Wimg, S, K, ncol0, SB[nt3], SB[nt3 + 1] - SB[nt3], nbuf, desc, wave, 2 * (st - SB[t3]), 2 * (st - SB[t3]) + 2);
                        rs_mfma_step<CT, F32>(w, xop[st], 32 * st + 16 < K, acc);
                        __builtin_amdgcn_sched_barrier(0);
                        w = wn;
                    }
                }
                RS_STAMP();
            }
            // epilogue of the block: y = relu(acc + b2); g = p_n y; ReLU bits; optional y rows      (as level_compose_fwd)
            f32x4 hacc[CT];
            uint32_t bits = 0;
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                const float4 bvc = ld4(bias + col0 + c * 16 + 4 * g);
                const float y0 = fmaxf(acc[c][0] + bvc.x, 0.f), y1 = fmaxf(acc[c][1] + bvc.y, 0.f);
                const float y2 = fmaxf(acc[c][2] + bvc.z, 0.f), y3 = fmaxf(acc[c][3] + bvc.w, 0.f);
                hacc[c][0] = fmaf(pn, y0, 0.f); hacc[c][1] = fmaf(pn, y1, 0.f);
                hacc[c][2] = fmaf(pn, y2, 0.f); hacc[c][3] = fmaf(pn, y3, 0.f);
                bits |= ((y0 > 0.f ? 1u : 0u) | (y1 > 0.f ? 2u : 0u) | (y2 > 0.f ? 4u : 0u) | (y3 > 0.f ? 8u : 0u)) << (4 * c);
                if (Y && ok) st4(Y + prow * Dp + col0 + c * 16 + 4 * g, make_float4(y0, y1, y2, y3));
            }
            if (ymask && ok) ymask[(prow * ncb + cb) * 4 + g] = bits;
            // sum over the WPG waves of a cell tile: the fixed tree of level_compose_fwd
#pragma unroll
            for (int stride = 4; stride >= 1; stride >>= 1) {
                if (WPG >= 2 * stride) {
                    const bool holding = r < 2 * stride;
                    const bool writer = holding && r >= stride;
                    if (writer) {
                        const int slot = j * stride + (r - stride);
#pragma unroll
                        for (int c = 0; c < CT; ++c) red[(slot * CT + c) * 64 + lane] = make_float4(hacc[c][0], hacc[c][1], hacc[c][2], hacc[c][3]);
                    }
                    __syncthreads();
                    if (holding && !writer) {
                        const int slot = j * stride + r;
#pragma unroll
                        for (int c = 0; c < CT; ++c) {
                            const float4 v = red[(slot * CT + c) * 64 + lane];
                            hacc[c][0] += v.x; hacc[c][1] += v.y; hacc[c][2] += v.z; hacc[c][3] += v.w;
                        }
                    }
                    __syncthreads();
                }
            }
            if (have && r == 0) {
                if (ti < lv.ncell) {
                    const int b = ti / lv.Lc, p = ti - b * lv.Lc;
                    float* o = HP + (size_t)s * hp_stride + ((size_t)b * lv.C + lv.off + p) * Dp + col0 + 4 * g;
#pragma unroll
                    for (int c = 0; c < CT; ++c) st4(o + c * 16, make_float4(hacc[c][0], hacc[c][1], hacc[c][2], hacc[c][3]));
                }
            }
        }
        RS_STAMP();
        __syncthreads();          // both buffers are free for the next task's first third
    }
}

}  // namespace cliora
