// Fused per-level kernels of the DioraMLP chart recursion (gfx950).
//
// One chart level of one pass (cliora/net/diora.py:295-310 inside_func, :358-376 outside_func) is TWO launches:
//
//   level_compose_fwd    for every split: x = relu(PL(a) + PR(b)), y = relu(W2 x + b2)   [split-bf16 MFMA, weights in LDS]
//                        and, in the SAME kernel, the softmax-weighted sum over the splits of a cell
//                            g = sum_n p_n y_n                                            (diora.py:137-146)
//                        -- the softmax weights do not depend on the compose output, so the per-split rows y_n (and x_n) never
//                        reach HBM; what is kept for the backward is one ReLU bit per element of y
//   level_project        h = g / max(||g||, eps)  (utils.py:11-14), the projections of the new cells
//                            [PL | PR | QL] = h Wcat^T + bias                             (factored first compose layer + bilinear)
//                        and -- first blocks of the same grid -- the split scores s_n = QL(a).h_b + s_a + s_b of the NEXT level,
//                        their softmax p_n and the cell score (score_cell; diora.py:125-134): a pair has at most one operand on
//                        the newest level and its partner is then a leaf, so the scoring needs none of the new projections
//   (level_scores: the scoring alone, for the first level of a pass; level_finish: norm + chart rows of a level without projections)
//
//
// Tile order.  A 16-row MFMA tile is (16 consecutive target cells t = b*Lc + p of the level) x (ONE split n): the N tiles of
// a "cell tile" differ only in n, so the weighted sum over the splits is an element-wise FMA into a register accumulator --
// no cross-lane reduction, no atomics, fixed summation order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "chart_kernels.hpp"
#include "wgrad_tiles.hpp"

namespace cliora {

struct PairLevel {
    const int32_t *pa, *pb;   // operand cells (ids inside one sentence's chart) of the level's pairs: index p*N + n
    int Lc, N, C, ncell;      // cells per sentence at the level, splits per cell, cells per chart, B*Lc
    int rowbase;              // global pair row of (t, n) = rowbase + t*N + n, t = b*Lc + p
    int off;                  // chart offset of the level: chart row of t = b*C + off + p
    long long tilebase;       // level_compose_bwd, tiled operands: storage tile of the level's wave tile 0 (Plan::tile_base_*)
};

constexpr int LC_SLOTS = 4;   // LDS slots of the cross-wave reduction (one per writer of a round)

// ---------------------------------------------------------------------------------
// level_compose_fwd
//   grid.y = column blocks of CT*16 output columns (the block's split-bf16 weight image stays in LDS);
//   grid.x walks TASKS = (group of TG cell tiles) x (part s of SP of the split range).  The 8 waves of the workgroup are
//   dealt WPG = 8 / TG waves per cell tile; wave r of a cell tile takes the splits n0 + r, n0 + r + WPG, ... and keeps
//   sum p_n y_n in registers; the WPG partial sums meet in LDS in a fixed tree order.  Output: HP[s][chart row][Dp] (partial
//   aggregates, summed over s by level_project), the ReLU bits of y, and on request the y rows (hooks).
//   Operand rows are fetched in the quad-coalesced lane map and moved to the MFMA lanes by ds_bpermute (gemm_kernels.hpp).
// ---------------------------------------------------------------------------------
// F32 = true: the exact-fp32 arithmetic mode (cliora_set_mfma_mode): the LDS image is the plain fp32 weight block ([CT*16][K],
// passed through Wimg with S_ = K) and a 32-deep k-step is eight v_mfma_f32_16x16x4_f32 per column tile instead of three bf16 ones.
// MFMA operand of one 32-deep k-step of a 16-row tile, in the MFMA lanes: split-bf16 (hi, lo) -- or, in the exact-fp32 mode,
// the two fp32 runs of four k as they are.  Prepared one k-step ahead of the MFMAs that consume it.
struct StepOperand { u32x4 h, l; };
template <bool F32>
__device__ __forceinline__ StepOperand make_operand(int psrc, const float4& f0, const float4& f1) {
    const float4 a0 = to_mfma_lanes(psrc, f0), a1 = to_mfma_lanes(psrc, f1);
    StepOperand o;
    if constexpr (F32) {
        o.h = u32x4{__float_as_uint(a0.x), __float_as_uint(a0.y), __float_as_uint(a0.z), __float_as_uint(a0.w)};
        o.l = u32x4{__float_as_uint(a1.x), __float_as_uint(a1.y), __float_as_uint(a1.z), __float_as_uint(a1.w)};
    } else {
        split_bf16x8(a0, a1, o.h, o.l);
    }
    return o;
}

template <int CT, bool F32>
__device__ __forceinline__ void kstep_mfma(const uint32_t* wimg, int i, int g, int S, int half, int st, bool second, const StepOperand& x,
                                           f32x4 (&acc)[CT]) {
    if constexpr (F32) {
        const float* wf = reinterpret_cast<const float*>(wimg) + i * S + 4 * g + 32 * st;
        const float a0[4] = {__uint_as_float(x.h[0]), __uint_as_float(x.h[1]), __uint_as_float(x.h[2]), __uint_as_float(x.h[3])};
        const float a1[4] = {__uint_as_float(x.l[0]), __uint_as_float(x.l[1]), __uint_as_float(x.l[2]), __uint_as_float(x.l[3])};
        float4 b0[CT], b1[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            b0[c] = *reinterpret_cast<const float4*>(wf + c * 16 * S);
            b1[c] = second ? *reinterpret_cast<const float4*>(wf + c * 16 * S + 16) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[c] = mfma16(b0[c].x, a0[0], acc[c]);
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[c] = mfma16(b0[c].y, a0[1], acc[c]);
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[c] = mfma16(b0[c].z, a0[2], acc[c]);
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[c] = mfma16(b0[c].w, a0[3], acc[c]);
        if (second) {
#pragma unroll
            for (int c = 0; c < CT; ++c) acc[c] = mfma16(b1[c].x, a1[0], acc[c]);
#pragma unroll
            for (int c = 0; c < CT; ++c) acc[c] = mfma16(b1[c].y, a1[1], acc[c]);
#pragma unroll
            for (int c = 0; c < CT; ++c) acc[c] = mfma16(b1[c].z, a1[2], acc[c]);
#pragma unroll
            for (int c = 0; c < CT; ++c) acc[c] = mfma16(b1[c].w, a1[3], acc[c]);
        }
    } else {
        const uint32_t* wfrag = wimg + i * S + 4 * g;
        u32x4 wh[CT], wl[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            wh[c] = *reinterpret_cast<const u32x4*>(wfrag + c * 16 * S + 16 * st);
            wl[c] = *reinterpret_cast<const u32x4*>(wfrag + c * 16 * S + 16 * st + half);
        }
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[c] = mfma32bf(wl[c], x.h, acc[c]);
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[c] = mfma32bf(wh[c], x.l, acc[c]);
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[c] = mfma32bf(wh[c], x.h, acc[c]);
    }
}

// the block's weight image -> LDS (LDS-DMA, lane-linear); the caller waits (vmcnt(0) + barrier) before the first use
__device__ __forceinline__ void stage_weight_image(const uint32_t* src, uint32_t* lds, int ndwords, int wave, int lane, int nthreads) {
    const int n16 = ndwords / 4;
    for (int e0 = wave * 64; e0 < n16; e0 += nthreads) {
        const int e = e0 + lane;
        if (e < n16)
            __builtin_amdgcn_global_load_lds((const void*)(src + (size_t)e * 4), (__attribute__((address_space(3))) void*)(lds + e0 * 4), 16, 0, 0);
    }
}

// The tasks bx, bx + gx, ... of one level's compose launch, by one workgroup (the whole body of level_compose_fwd; the two-segment
// kernel level_compose_fwd2 runs it on the segment its block index falls into: same tasks, same order, same bits).
template <int CT, int K16, bool F32>
__device__ __forceinline__ void compose_fwd_tasks(uint32_t* lds_img, const int bx, const int gx, const uint32_t* __restrict__ Wimg, int S_, int K_,
                                                  const PairLevel& lv, const float* __restrict__ PA, int lda, const float* __restrict__ PB, int ldb,
                                                  const float* __restrict__ bias, const float* __restrict__ Pp,
                                                  int TG, int SP, int ntask, float* __restrict__ HP, size_t hp_stride, int Dp,
                                                  uint32_t* __restrict__ ymask, float* __restrict__ Y) {
    constexpr int WAVES = 8, T = WAVES * 64, PD = 4;
    constexpr bool KS = K16 > 0;
    constexpr int UNROLL_STEPS = KS ? 64 : 1;
    const int K = KS ? K16 * 16 : K_;
    const int S = F32 ? K : (KS ? (K16 + 1) / 2 * 32 + WS3_PAD : S_);     // row stride of the LDS image in dwords
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 15, g = lane >> 4;
    const int li = fetch_row_of(lane), lg = fetch_piece_of(lane), psrc = mfma_src_addr(lane);
    const int Kp = F32 ? (K + 31) / 32 * 32 : S - WS3_PAD, half = Kp >> 1;
    const int by = blockIdx.y, gy = gridDim.y;
    const int col0 = by * (CT * 16);
    // waited for below, after the first row contexts are on their way
    stage_weight_image(Wimg + (size_t)col0 * S, lds_img, CT * 16 * S, wave, lane, T);
    float4* red = reinterpret_cast<float4*>(lds_img + CT * 16 * S);      // [LC_SLOTS][CT][64]
    float4 bv[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) bv[c] = ld4(bias + col0 + c * 16 + 4 * g);
    const int nsteps = Kp >> 5;
    const int nsteps_p = (nsteps + PD - 1) / PD * PD;
    int wimg_off = 0;
    const int WPG = WAVES / TG;                      // waves per cell tile (1, 2, 4 or 8)
    const int j = wave / WPG, r = wave - j * WPG;
    const int G = (lv.ncell + 15) >> 4;
    const int Ns = (lv.N + SP - 1) / SP;

    struct Ctx { const float *pa, *pb; };
    auto rowctx = [&](int gt, int n) {               // fetch-lane view of tile (gt, n)
        const int t = min(gt * 16 + li, lv.ncell - 1);            // clamp: computed, masked out by p = 0 and never stored
        const int b = t / lv.Lc, p = t - b * lv.Lc;
        const int idx = p * lv.N + n;
        const size_t ca = (size_t)b * lv.C + lv.pa[idx], cb = (size_t)b * lv.C + lv.pb[idx];
        return Ctx{PA + ca * lda, PB + cb * ldb};
    };
    Raw2 ra[PD][2];
    auto issue = [&](int slot, const Ctx& c, int s) {
        const int k = 32 * s + 4 * lg;
        const int k2 = k + (32 * s + 16 < K ? 16 : 0);
        ra[slot][0] = Raw2{ld4(c.pa + k), ld4(c.pb + k)};
        ra[slot][1] = Raw2{ld4(c.pa + k2), ld4(c.pb + k2)};
    };
    auto relu_add = [](const Raw2& q) {
        return make_float4(fmaxf(q.u.x + q.v.x, 0.f), fmaxf(q.u.y + q.v.y, 0.f), fmaxf(q.u.z + q.v.z, 0.f), fmaxf(q.u.w + q.v.w, 0.f));
    };

    bool staged = false;
    for (int task = bx; task < ntask; task += gx) {
        const int gg = task / SP, s = task - gg * SP;
        const int gt = gg * TG + j;                  // this wave's cell tile
        const bool have = gt < G;
        const int n0 = s * Ns, n1 = min(lv.N, n0 + Ns);
        f32x4 hacc[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) hacc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        const bool work = have && n0 + r < n1;
        Ctx ctx = rowctx(min(gt, G - 1), work ? n0 + r : n0);        // index loads overlap the weight staging
        if (!staged) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            staged = true;
        }
        if (work) {
            int n = n0 + r;
#pragma unroll
            for (int sl = 0; sl < PD; ++sl) issue(sl, ctx, sl < nsteps ? sl : 0);
            while (true) {
                const int nn = n + WPG;
                const bool has_next = nn < n1;
                const Ctx ctxn = rowctx(gt, has_next ? nn : n);
                // MFMA-lane view of the tile: row i is target cell ti, pair row prow
                const int ti = gt * 16 + i;
                const bool ok = ti < lv.ncell;
                const size_t prow = (size_t)lv.rowbase + (size_t)min(ti, lv.ncell - 1) * lv.N + n;
                const float pn = ok ? Pp[prow] : 0.f;
                f32x4 acc[CT];
#pragma unroll
                for (int c = 0; c < CT; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
                asm volatile("" : "+v"(wimg_off));           // keep the weight-fragment LDS reads inside the tile loop
                const uint32_t* wimg = lds_img + wimg_off;
                // Software pipeline over the k-steps: the operand of step st+1 (add, ReLU, lane move, split: VALU + LDS crossbar) is
                // prepared beside the MFMAs of step st; the ring slot of step st (consumed one iteration earlier) is refilled with
                // step st+PD -- or, in the tile's last PD steps, with the next tile's step of the same slot.  The scheduling barrier keeps
                // each refill where it is written: hipcc otherwise sinks the loads to their use (one exposed latency per k-step).
                StepOperand cur = make_operand<F32>(psrc, relu_add(ra[0][0]), relu_add(ra[0][1]));
#pragma unroll UNROLL_STEPS
                for (int base = 0; base < nsteps_p; base += PD) {
#pragma unroll
                    for (int sl = 0; sl < PD; ++sl) {
                        const int st = base + sl;
                        if (st < nsteps) {
                            StepOperand nxt = cur;
                            if (st + 1 < nsteps) nxt = make_operand<F32>(psrc, relu_add(ra[(sl + 1) % PD][0]), relu_add(ra[(sl + 1) % PD][1]));
                            kstep_mfma<CT, F32>(wimg, i, g, S, half, st, 32 * st + 16 < K, cur, acc);
                            const int nst = st + PD;
                            const bool in_cur = nst < nsteps;
                            issue(sl, pick_pod(in_cur, ctx, ctxn), in_cur ? nst : (sl < nsteps ? sl : 0));
                            __builtin_amdgcn_sched_barrier(0);
                            cur = nxt;
                        }
                    }
                }
                // epilogue of the tile: y = relu(acc + b2); g += p_n y; ReLU bits; optional y rows
                uint32_t bits = 0;
#pragma unroll
                for (int c = 0; c < CT; ++c) {
                    const float y0 = fmaxf(acc[c][0] + bv[c].x, 0.f), y1 = fmaxf(acc[c][1] + bv[c].y, 0.f);
                    const float y2 = fmaxf(acc[c][2] + bv[c].z, 0.f), y3 = fmaxf(acc[c][3] + bv[c].w, 0.f);
                    hacc[c][0] = fmaf(pn, y0, hacc[c][0]); hacc[c][1] = fmaf(pn, y1, hacc[c][1]);
                    hacc[c][2] = fmaf(pn, y2, hacc[c][2]); hacc[c][3] = fmaf(pn, y3, hacc[c][3]);
                    bits |= ((y0 > 0.f ? 1u : 0u) | (y1 > 0.f ? 2u : 0u) | (y2 > 0.f ? 4u : 0u) | (y3 > 0.f ? 8u : 0u)) << (4 * c);
                    if (Y && ok) st4(Y + prow * Dp + col0 + c * 16 + 4 * g, make_float4(y0, y1, y2, y3));
                }
                if (ymask && ok) ymask[(prow * gy + by) * 4 + g] = bits;
                if (!has_next) break;
                ctx = ctxn;
                n = nn;
            }
        }
        // ---- sum over the WPG waves of a cell tile: fixed tree.  In the round of `stride` the waves r < 2*stride still hold
        // data; r >= stride park their accumulators in LDS (slot j*stride + r - stride < 4), r < stride add their partner's.
        // WPG is the same for every wave of the launch, so every wave takes the same barriers.
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
            const int ti = gt * 16 + i;
            if (ti < lv.ncell) {
                const int b = ti / lv.Lc, p = ti - b * lv.Lc;
                float* o = HP + (size_t)s * hp_stride + ((size_t)b * lv.C + lv.off + p) * Dp + col0 + 4 * g;
#pragma unroll
                for (int c = 0; c < CT; ++c) st4(o + c * 16, make_float4(hacc[c][0], hacc[c][1], hacc[c][2], hacc[c][3]));
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// compose_fwd_tasks_dual (round 5): the same tasks with TWO splits of a cell tile in flight per wave.
//
// What bounds level_compose_fwd on the big levels is the CU's LDS pipe, not the MFMA or the VALU: per 32-deep k-step a wave reads the
// column block's weight fragments (ten ds_read_b128 = 10 KB) and moves its operand rows to the MFMA lanes (eight ds_bpermute = 2 KB) for
// fifteen MFMAs; eight waves x 12 KB = 96 KB per k-step at 128 B / clock = 750 of the ~1 150 clocks a k-step takes (profiles/r05_notes.md).
// A wave's splits n, n + WPG, n + 2 WPG, ... of one cell tile share the block's weights: here the wave takes them TWO at a time, reads
// each weight fragment once per k-step and feeds it to both splits' accumulators (30 MFMAs per 10 KB of weights).  Each accumulator still
// sees its products in the order (w_lo x_hi, w_hi x_lo, w_hi x_hi) per k-step and the weighted sum g += p_n y_n still runs in the
// wave's split order (A = n before B = n + WPG), so the results are bitwise those of compose_fwd_tasks; an odd last split runs alone.
// The bias sits in LDS behind the reduction slots (20 registers the second accumulator set needs); PD ring slots per split.
// ---------------------------------------------------------------------------------
template <int CT, bool F32>
__device__ __forceinline__ void kstep_mfma2(const uint32_t* wimg, int i, int g, int S, int half, int st, bool second, const StepOperand& xa,
                                            const StepOperand& xb, f32x4 (&acca)[CT], f32x4 (&accb)[CT]) {
    if constexpr (F32) {
        kstep_mfma<CT, F32>(wimg, i, g, S, half, st, second, xa, acca);
        kstep_mfma<CT, F32>(wimg, i, g, S, half, st, second, xb, accb);
    } else {
        const uint32_t* wfrag = wimg + i * S + 4 * g;
        u32x4 wh[CT], wl[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            wh[c] = *reinterpret_cast<const u32x4*>(wfrag + c * 16 * S + 16 * st);
            wl[c] = *reinterpret_cast<const u32x4*>(wfrag + c * 16 * S + 16 * st + half);
        }
#pragma unroll
        for (int c = 0; c < CT; ++c) { acca[c] = mfma32bf(wl[c], xa.h, acca[c]); accb[c] = mfma32bf(wl[c], xb.h, accb[c]); }
#pragma unroll
        for (int c = 0; c < CT; ++c) { acca[c] = mfma32bf(wh[c], xa.l, acca[c]); accb[c] = mfma32bf(wh[c], xb.l, accb[c]); }
#pragma unroll
        for (int c = 0; c < CT; ++c) { acca[c] = mfma32bf(wh[c], xa.h, acca[c]); accb[c] = mfma32bf(wh[c], xb.h, accb[c]); }
    }
}

template <int CT, int K16, bool F32, int PD>
__device__ __forceinline__ void compose_fwd_tasks_dual(uint32_t* lds_img, const int bx, const int gx, const uint32_t* __restrict__ Wimg, int S_, int K_,
                                                       const PairLevel& lv, const float* __restrict__ PA, int lda, const float* __restrict__ PB, int ldb,
                                                       const float* __restrict__ bias, const float* __restrict__ Pp,
                                                       int TG, int SP, int ntask, float* __restrict__ HP, size_t hp_stride, int Dp,
                                                       uint32_t* __restrict__ ymask, float* __restrict__ Y) {
    constexpr int WAVES = 8, T = WAVES * 64;
    constexpr bool KS = K16 > 0;
    constexpr int UNROLL_STEPS = KS ? 64 : 1;
    const int K = KS ? K16 * 16 : K_;
    const int S = F32 ? K : (KS ? (K16 + 1) / 2 * 32 + WS3_PAD : S_);     // row stride of the LDS image in dwords
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 15, g = lane >> 4;
    const int li = fetch_row_of(lane), lg = fetch_piece_of(lane), psrc = mfma_src_addr(lane);
    const int Kp = F32 ? (K + 31) / 32 * 32 : S - WS3_PAD, half = Kp >> 1;
    const int by = blockIdx.y, gy = gridDim.y;
    const int col0 = by * (CT * 16);
    stage_weight_image(Wimg + (size_t)col0 * S, lds_img, CT * 16 * S, wave, lane, T);
    float4* red = reinterpret_cast<float4*>(lds_img + CT * 16 * S);      // [LC_SLOTS][CT][64]
    float* b2s = reinterpret_cast<float*>(red + LC_SLOTS * CT * 64);     // the block's bias (CT * 16 floats), read in the tile epilogues
    for (int c = threadIdx.x; c < CT * 16; c += T) b2s[c] = bias[col0 + c];
    const int nsteps = Kp >> 5;
    const int nsteps_p = (nsteps + PD - 1) / PD * PD;
    int wimg_off = 0;
    const int WPG = WAVES / TG;
    const int j = wave / WPG, r = wave - j * WPG;
    const int G = (lv.ncell + 15) >> 4;
    const int Ns = (lv.N + SP - 1) / SP;

    struct Ctx { const float *pa, *pb; };
    auto rowctx = [&](int gt, int n) {
        const int t = min(gt * 16 + li, lv.ncell - 1);
        const int b = t / lv.Lc, p = t - b * lv.Lc;
        const int idx = p * lv.N + n;
        const size_t ca = (size_t)b * lv.C + lv.pa[idx], cb = (size_t)b * lv.C + lv.pb[idx];
        return Ctx{PA + ca * lda, PB + cb * ldb};
    };
    Raw2 raA[PD][2], raB[PD][2];
    auto issue = [&](Raw2 (&ra)[PD][2], int slot, const Ctx& c, int s) {
        const int k = 32 * s + 4 * lg;
        const int k2 = k + (32 * s + 16 < K ? 16 : 0);
        ra[slot][0] = Raw2{ld4(c.pa + k), ld4(c.pb + k)};
        ra[slot][1] = Raw2{ld4(c.pa + k2), ld4(c.pb + k2)};
    };
    auto relu_add = [](const Raw2& q) {
        return make_float4(fmaxf(q.u.x + q.v.x, 0.f), fmaxf(q.u.y + q.v.y, 0.f), fmaxf(q.u.z + q.v.z, 0.f), fmaxf(q.u.w + q.v.w, 0.f));
    };

    bool staged = false;
    for (int task = bx; task < ntask; task += gx) {
        const int gg = task / SP, s = task - gg * SP;
        const int gt = gg * TG + j;
        const bool have = gt < G;
        const int n0 = s * Ns, n1 = min(lv.N, n0 + Ns);
        f32x4 hacc[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) hacc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        const bool work = have && n0 + r < n1;
        const int gtc = min(gt, G - 1);
        int n = work ? n0 + r : n0;
        Ctx ctxA = rowctx(gtc, n);
        Ctx ctxB = rowctx(gtc, (work && n + WPG < n1) ? n + WPG : n);
        if (!staged) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            staged = true;
        }
        if (work) {
#pragma unroll
            for (int sl = 0; sl < PD; ++sl) { issue(raA, sl, ctxA, sl < nsteps ? sl : 0); issue(raB, sl, ctxB, sl < nsteps ? sl : 0); }
            // epilogue of one (tile, split): y = relu(acc + b2); g += p_n y; ReLU bits; optional y rows
            auto epilogue = [&](int nn_, f32x4 (&acc)[CT]) {
                const int ti = gt * 16 + i;
                const bool ok = ti < lv.ncell;
                const size_t prow = (size_t)lv.rowbase + (size_t)min(ti, lv.ncell - 1) * lv.N + nn_;
                const float pn = ok ? Pp[prow] : 0.f;
                uint32_t bits = 0;
#pragma unroll
                for (int c = 0; c < CT; ++c) {
                    const float4 bvc = *reinterpret_cast<const float4*>(b2s + c * 16 + 4 * g);
                    const float y0 = fmaxf(acc[c][0] + bvc.x, 0.f), y1 = fmaxf(acc[c][1] + bvc.y, 0.f);
                    const float y2 = fmaxf(acc[c][2] + bvc.z, 0.f), y3 = fmaxf(acc[c][3] + bvc.w, 0.f);
                    hacc[c][0] = fmaf(pn, y0, hacc[c][0]); hacc[c][1] = fmaf(pn, y1, hacc[c][1]);
                    hacc[c][2] = fmaf(pn, y2, hacc[c][2]); hacc[c][3] = fmaf(pn, y3, hacc[c][3]);
                    bits |= ((y0 > 0.f ? 1u : 0u) | (y1 > 0.f ? 2u : 0u) | (y2 > 0.f ? 4u : 0u) | (y3 > 0.f ? 8u : 0u)) << (4 * c);
                    if (Y && ok) st4(Y + prow * Dp + col0 + c * 16 + 4 * g, make_float4(y0, y1, y2, y3));
                }
                if (ymask && ok) ymask[(prow * gy + by) * 4 + g] = bits;
            };
            while (true) {
                const bool dual = n + WPG < n1;                          // wave-uniform
                const int nB = n + WPG;
                const int nnA = n + (dual ? 2 : 1) * WPG;                // the wave's next split(s)
                const bool nextA = nnA < n1, nextB = nnA + WPG < n1;
                const Ctx ctxnA = rowctx(gt, nextA ? nnA : n);
                const Ctx ctxnB = rowctx(gt, nextB ? nnA + WPG : n);
                asm volatile("" : "+v"(wimg_off));
                const uint32_t* wimg = lds_img + wimg_off;
                f32x4 accA[CT], accB[CT];
#pragma unroll
                for (int c = 0; c < CT; ++c) { accA[c] = f32x4{0.f, 0.f, 0.f, 0.f}; accB[c] = f32x4{0.f, 0.f, 0.f, 0.f}; }
                if (dual) {
                    StepOperand curA = make_operand<F32>(psrc, relu_add(raA[0][0]), relu_add(raA[0][1]));
                    StepOperand curB = make_operand<F32>(psrc, relu_add(raB[0][0]), relu_add(raB[0][1]));
#pragma unroll UNROLL_STEPS
                    for (int base = 0; base < nsteps_p; base += PD) {
#pragma unroll
                        for (int sl = 0; sl < PD; ++sl) {
                            const int st = base + sl;
                            if (st < nsteps) {
                                StepOperand nxtA = curA, nxtB = curB;
                                if (st + 1 < nsteps) {
                                    nxtA = make_operand<F32>(psrc, relu_add(raA[(sl + 1) % PD][0]), relu_add(raA[(sl + 1) % PD][1]));
                                    nxtB = make_operand<F32>(psrc, relu_add(raB[(sl + 1) % PD][0]), relu_add(raB[(sl + 1) % PD][1]));
                                }
                                kstep_mfma2<CT, F32>(wimg, i, g, S, half, st, 32 * st + 16 < K, curA, curB, accA, accB);
                                const int nst = st + PD;
                                const bool in_cur = nst < nsteps;
                                const int step = in_cur ? nst : (sl < nsteps ? sl : 0);
                                issue(raA, sl, pick_pod(in_cur, ctxA, ctxnA), step);
                                issue(raB, sl, pick_pod(in_cur, ctxB, ctxnB), step);
                                __builtin_amdgcn_sched_barrier(0);
                                curA = nxtA; curB = nxtB;
                            }
                        }
                    }
                    epilogue(n, accA);
                    epilogue(nB, accB);
                } else {
                    StepOperand cur = make_operand<F32>(psrc, relu_add(raA[0][0]), relu_add(raA[0][1]));
#pragma unroll UNROLL_STEPS
                    for (int base = 0; base < nsteps_p; base += PD) {
#pragma unroll
                        for (int sl = 0; sl < PD; ++sl) {
                            const int st = base + sl;
                            if (st < nsteps) {
                                StepOperand nxt = cur;
                                if (st + 1 < nsteps) nxt = make_operand<F32>(psrc, relu_add(raA[(sl + 1) % PD][0]), relu_add(raA[(sl + 1) % PD][1]));
                                kstep_mfma<CT, F32>(wimg, i, g, S, half, st, 32 * st + 16 < K, cur, accA);
                                const int nst = st + PD;
                                const bool in_cur = nst < nsteps;
                                issue(raA, sl, pick_pod(in_cur, ctxA, ctxnA), in_cur ? nst : (sl < nsteps ? sl : 0));
                                __builtin_amdgcn_sched_barrier(0);
                                cur = nxt;
                            }
                        }
                    }
                    epilogue(n, accA);
                }
                if (!nextA) break;
                ctxA = ctxnA; ctxB = ctxnB;
                n = nnA;
            }
        }
        // ---- sum over the WPG waves of a cell tile: the fixed tree of compose_fwd_tasks
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
            const int ti = gt * 16 + i;
            if (ti < lv.ncell) {
                const int b = ti / lv.Lc, p = ti - b * lv.Lc;
                float* o = HP + (size_t)s * hp_stride + ((size_t)b * lv.C + lv.off + p) * Dp + col0 + 4 * g;
#pragma unroll
                for (int c = 0; c < CT; ++c) st4(o + c * 16, make_float4(hacc[c][0], hacc[c][1], hacc[c][2], hacc[c][3]));
            }
        }
    }
}

template <int CT, int K16, bool F32>
__global__ __launch_bounds__(512) void level_compose_fwd(const uint32_t* __restrict__ Wimg, int S_, int K_, PairLevel lv,
                                                         const float* __restrict__ PA, int lda, const float* __restrict__ PB, int ldb,
                                                         const float* __restrict__ bias, const float* __restrict__ Pp,
                                                         int TG, int SP, int ntask, float* __restrict__ HP, size_t hp_stride, int Dp,
                                                         uint32_t* __restrict__ ymask, float* __restrict__ Y) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_img[];
    compose_fwd_tasks<CT, K16, F32>(lds_img, blockIdx.x, gridDim.x, Wimg, S_, K_, lv, PA, lda, PB, ldb, bias, Pp, TG, SP, ntask, HP, hp_stride, Dp, ymask, Y);
}

// ---------------------------------------------------------------------------------
// level_compose_fwd2 (round 5): the compose launches of BOTH passes of one wavefront step -- inside level k and outside level L-k
//  -- as ONE grid on ONE queue.  grid.x = seg0.gx + seg1.gx (+ padding to a multiple of 8: the column blocks of
// one x then share an XCD): the blocks [0, seg0.gx) run segment 0's tasks with stride seg0.gx, the next seg1.gx blocks segment 1's --
// each segment with the geometry (TG, SP, ntask) the plan gives its level, which is sized for its share of the compose workgroups
// (plan.cpp), so a workgroup does exactly what a workgroup of the level's own launch did and the results are the same bits.  What goes
// away is the second queue: no cross-stream event per step (a barrier packet on both queues, ~5 us each), no staggered start.
// ---------------------------------------------------------------------------------
struct ComposeSeg {
    const uint32_t* Wimg;     // the pass's W2 image (split-bf16) or plain fp32 weight (exact mode)
    PairLevel lv;
    const float *PA, *PB, *bias;
    int lda, ldb;
    int TG, SP, ntask, gx;    // gx = 0: no such segment in this step
    float* HP;
};
// DUALPD > 0: compose_fwd_tasks_dual with that ring depth (two splits of a cell tile in flight per wave); 0: compose_fwd_tasks
template <int CT, int K16, bool F32, int DUALPD = 0>
__global__ __launch_bounds__(512) void level_compose_fwd2(ComposeSeg s0, ComposeSeg s1, int S_, int K_, const float* __restrict__ Pp, size_t hp_stride,
                                                          int Dp, uint32_t* __restrict__ ymask, float* __restrict__ Y) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_img[];
    const bool second = (int)blockIdx.x >= s0.gx;                  // workgroup-uniform: scalar selects
    const int bx = (int)blockIdx.x - (second ? s0.gx : 0);
    const int gx = second ? s1.gx : s0.gx;
    if (bx >= gx) return;                                           // padding blocks
#define SEG(f) (second ? s1.f : s0.f)
    PairLevel lv;
    lv.pa = SEG(lv.pa); lv.pb = SEG(lv.pb); lv.Lc = SEG(lv.Lc); lv.N = SEG(lv.N); lv.C = SEG(lv.C); lv.ncell = SEG(lv.ncell);
    lv.rowbase = SEG(lv.rowbase); lv.off = SEG(lv.off); lv.tilebase = SEG(lv.tilebase);
    if constexpr (DUALPD > 0)
        compose_fwd_tasks_dual<CT, K16, F32, DUALPD>(lds_img, bx, gx, SEG(Wimg), S_, K_, lv, SEG(PA), SEG(lda), SEG(PB), SEG(ldb), SEG(bias), Pp, SEG(TG),
                                                     SEG(SP), SEG(ntask), SEG(HP), hp_stride, Dp, ymask, Y);
    else
        compose_fwd_tasks<CT, K16, F32>(lds_img, bx, gx, SEG(Wimg), S_, K_, lv, SEG(PA), SEG(lda), SEG(PB), SEG(ldb), SEG(bias), Pp, SEG(TG), SEG(SP),
                                        SEG(ntask), SEG(HP), hp_stride, Dp, ymask, Y);
#undef SEG
}

// sum of the SP partial aggregates of one chart row, fixed order
template <int SP>
__device__ __forceinline__ float4 sum_parts(const float* hp, size_t hp_stride, int k) {
    float4 a = ld4(hp + k);
#pragma unroll
    for (int s = 1; s < SP; ++s) a = f4add(a, ld4(hp + (size_t)s * hp_stride + k));
    return a;
}

// ---------------------------------------------------------------------------------
// level_project: unit norm of the level's aggregates + projection of the new cells, one launch.
//   rows = the level's cells (r = b*Lc + p), A(r, :) = sum_s HP[s][chart row]  (the un-normalised g of level_compose_fwd);
//   out[chart row][col] = (sum_k A(r,k) W[col][k]) / max(||A(r,:)||, eps) + bias[col]      -- the projection is linear, so the
//   norm is applied to the accumulators: ||A|| is summed while the row streams through as the MFMA operand (no extra pass).
//   The column-block-0 workgroups also write H = A / max(||A||, eps) (the chart output) and the raw norm.
//   Same split-K structure as rows_gemm_ksplit<1, CT, FRAG>: one workgroup = 16 rows x CT*16 columns, the four waves split
//   the reduction, weights from the fragment image (frag_weight_image), exact fp32 MFMA.
// ---------------------------------------------------------------------------------
//
// The split scores of the NEXT level ride in the same launch (ScoreArgs, blocks [0, sc.nscore)): they need the new cells' h
// but none of their projections -- a pair has at most one operand on the newest level, and its partner is then a leaf:
//   newest cell is the right child b:   s = QL(leaf a) . h_b                 (QL of the leaves is long there)
//   newest cell is the left child a:    s = h_a . QR(leaf b),  QR = M h      (the leaves' QR is projected once per forward)
// with h = g / max(||g||, eps) formed on the fly from the partial aggregates -- so the scoring leaves the critical path
// (compose -> [projection || next scores] -> compose) instead of being a launch of its own between two compose kernels.
struct ScoreArgs {
    int nscore;                     // target cells of the level to score (0: no scoring in this launch)
    LevelArgs g;                    // that level
    const int32_t *pa, *pb;         // operand cells (ids inside one sentence's chart) of the level's pairs: index p*N + n
    const float* QA; int ldA;       // QL table of the a operands (row stride ldA)
    const float* HB;                // H chart of the b operands
    const float *SA, *SB;           // chart scores of the a / b operands
    float *Sp, *Pp, *Sout;          // per-split score, softmax weight; the target cells' scores
    int new_lo, new_hi;             // cells [new_lo, new_hi) of a sentence's chart are the newest level (not yet in HB / QA)
    int a_can_be_new;               // inside pass: both operands live in the inside chart; outside pass: only the parent (b)
    const float* HA;                // H chart of the a operands (inside pass: = HB)
    const float* HPn; size_t hp_stride; int SPn; int normalize;   // partial aggregates of the newest cells; nullptr: HA / HB already hold them
    const float* QRleaf;            // (B*L, Dp): M h of the leaves
    int L;
};

// score + softmax of one target cell (pair_scores_fwd with the newest-level operands taken from the partial aggregates)
//
// Wave w scores the splits n = w, w + 4, ...  Round 5: as a pipeline instead of one split after the other -- the ISA of the first form
// was, per split, index loads -> wait -> operand rows -> wait -> the two chart scores -> wait: three dependent global round trips x five
// splits per wave at L = 20, 10 us for the score blocks of a launch (profiles/r05_notes.md).  Now one vector load fetches the operand
// cells of ALL of the wave's splits (lane j: split w + 4 j), the "plain" splits (both operands final: rows of QA / HB) run SCORE_KF at a
// time with all their rows and chart scores in flight together, and the (at most two per cell) splits with an operand on the newest
// level follow.  Same formula per split, so the same bits.
#ifndef CLIORA_SCORE_KF
#define CLIORA_SCORE_KF 2
#endif
constexpr int SCORE_KF = CLIORA_SCORE_KF;
__device__ __forceinline__ void score_cell(const ScoreArgs& sc, int t, float* sh_s) {
    const LevelArgs& g = sc.g;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = t / g.Lc, p = t - b * g.Lc;
    const int row0 = g.rowbase + t * g.N;
    const int Dp = g.Dp, nv = Dp >> 2;
    const bool a0 = lane < nv, a1 = lane + 64 < nv;
    const int c0 = 4 * lane, c1 = 4 * (lane + 64);
    const int bC = b * g.C;
    int my_ca = 0, my_cb = 0;                       // operand cells of this wave's split j = lane
    {
        const int n = wave + 4 * lane;
        if (n < g.N) { my_ca = sc.pa[p * g.N + n]; my_cb = sc.pb[p * g.N + n]; }
    }
    auto is_new = [&](int ca, int cb) { return (sc.a_can_be_new && ca >= sc.new_lo && ca < sc.new_hi) || (cb >= sc.new_lo && cb < sc.new_hi); };
    struct Plain { float4 u0, u1, v0, v1; float sa, sb; };
    auto fetch = [&](int ca, int cb) {
        const int ar = bC + ca, br = bC + cb;
        Plain o{f4zero(), f4zero(), f4zero(), f4zero(), 0.f, 0.f};
        const float* qa = sc.QA + (size_t)ar * sc.ldA;
        const float* hb = sc.HB + (size_t)br * Dp;
        if (a0) { o.u0 = ld4(qa + c0); o.v0 = ld4(hb + c0); }
        if (a1) { o.u1 = ld4(qa + c1); o.v1 = ld4(hb + c1); }
        o.sa = sc.SA[ar]; o.sb = sc.SB[br];
        return o;
    };
    auto finish = [&](int n, const Plain& o) {
        const float s = wave_sum(f4dot(o.u0, o.v0) + f4dot(o.u1, o.v1)) / 1.f + o.sa + o.sb;
        if (lane == 0) sh_s[n] = s;
    };
    {
        // SCORE_KF plain splits in flight at a time: their rows and chart scores are all issued before the first is reduced (distinct
        // registers, fully unrolled -- a rotating "pending" copy made hipcc wait for the loads before the copy)
        int j = 0;
        auto next_plain = [&](int& n, int& ca, int& cb) {             // wave-uniform walk over this wave's splits
            while (wave + 4 * j < g.N) {
                ca = __builtin_amdgcn_readlane(my_ca, j); cb = __builtin_amdgcn_readlane(my_cb, j);
                n = wave + 4 * j;
                ++j;
                if (!is_new(ca, cb)) return true;
            }
            return false;
        };
        while (true) {
            Plain o[SCORE_KF];
            int nn[SCORE_KF];
            bool have[SCORE_KF];
#pragma unroll
            for (int q = 0; q < SCORE_KF; ++q) {
                int ca = 0, cb = 0;
                nn[q] = 0;
                have[q] = next_plain(nn[q], ca, cb);
                if (have[q]) o[q] = fetch(ca, cb);
            }
#pragma unroll
            for (int q = 0; q < SCORE_KF; ++q)
                if (have[q]) finish(nn[q], o[q]);
            if (!have[SCORE_KF - 1]) break;
        }
    }
    for (int j = 0, n = wave; n < g.N; ++j, n += 4) {                // the splits with an operand on the newest level
        const int ca = __builtin_amdgcn_readlane(my_ca, j), cb = __builtin_amdgcn_readlane(my_cb, j);
        if (!is_new(ca, cb)) continue;
        const int ar = bC + ca, br = bC + cb;
        const bool a_new = sc.a_can_be_new && ca >= sc.new_lo && ca < sc.new_hi;
        float4 u0 = f4zero(), u1 = f4zero(), v0 = f4zero(), v1 = f4zero();
        float den = 1.f;
        auto newest = [&](int crow, float4& x0, float4& x1) {       // h of a newest-level cell (un-normalised) and its norm
            if (sc.HPn) {
                for (int sp = 0; sp < sc.SPn; ++sp) {
                    const float* src = sc.HPn + (size_t)sp * sc.hp_stride + (size_t)crow * Dp;
                    if (a0) x0 = f4add(x0, ld4(src + c0));
                    if (a1) x1 = f4add(x1, ld4(src + c1));
                }
                const float nr = sqrtf(wave_sum(f4dot(x0, x0) + f4dot(x1, x1)));
                den = sc.normalize ? fmaxf(nr, UNIT_EPS) : 1.f;
            } else {
                const float* src = (a_new ? sc.HA : sc.HB) + (size_t)crow * Dp;
                if (a0) x0 = ld4(src + c0);
                if (a1) x1 = ld4(src + c1);
            }
        };
        const float sa = sc.SA[ar], sb = sc.SB[br];
        if (a_new) {                               // partner is a leaf: QR(leaf) = M h_b
            const float* qr = sc.QRleaf + ((size_t)b * sc.L + cb) * Dp;
            if (a0) v0 = ld4(qr + c0);
            if (a1) v1 = ld4(qr + c1);
            newest(ar, u0, u1);
        } else {
            const float* qa = sc.QA + (size_t)ar * sc.ldA;
            if (a0) u0 = ld4(qa + c0);
            if (a1) u1 = ld4(qa + c1);
            newest(br, v0, v1);
        }
        const float s = wave_sum(f4dot(u0, v0) + f4dot(u1, v1)) / den + sa + sb;
        if (lane == 0) sh_s[n] = s;
    }
    __syncthreads();
    if (wave != 0) return;
    const float my_s = lane < g.N ? sh_s[lane] : -INFINITY;
    const float m = wave_max(my_s);
    const float e = lane < g.N ? expf(my_s - m) : 0.f;
    const float pn = e / wave_sum(e);
    if (lane < g.N) { sc.Sp[row0 + lane] = my_s; sc.Pp[row0 + lane] = pn; }
    const float st = wave_sum(lane < g.N ? pn * my_s : 0.f);
    if (lane == 0) sc.Sout[(size_t)bC + g.off + p] = st;
}

// the scoring alone (first level of a pass: every operand is already final)
static __global__ __launch_bounds__(256) void level_scores(ScoreArgs sc) {
    __shared__ float sh_s[64];
    score_cell(sc, cell_of_block(blockIdx.x, sc.g.B, sc.g.Lc, sc.g.affine), sh_s);
}

// one projection block (16 rows x CT*16 columns) of level_project: block `bid` of nrgp * ncolblocks
template <int CT, int SP>
__device__ __forceinline__ void project_block(float4 (*part)[CT][64], float (*sh_ss)[16], const int bid, const float* __restrict__ Wfrag, int K, int nrg,
                                              int nrgp, int ncell, int Lc, int C, int off, const float* __restrict__ HP, size_t hp_stride,
                                              int normalize, const float* __restrict__ bias, float* __restrict__ P, int ldp,
                                              float* __restrict__ H, float* __restrict__ nrm) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 15, q = lane >> 4;
    const int li = fetch_row_of(lane), lq = fetch_piece_of(lane), psrc = mfma_src_addr(lane);
    const int cb = bid / nrgp, rg = bid - cb * nrgp;
    if (rg >= nrg) return;
    const int col0 = cb * (CT * 16);
    const int nchunks = K >> 4;
    const int cbase = nchunks / 4, crem = nchunks % 4;
    const int ch0 = wave * cbase + min(wave, crem);
    const int nch = cbase + (wave < crem ? 1 : 0);
    auto crow_of = [&](int r) { const int rc = min(r, ncell - 1); const int b = rc / Lc; return (size_t)b * C + off + (rc - b * Lc); };
    const float* hp = HP + crow_of(rg * 16 + li) * K;
    f32x4 acc[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int PD = 4;
    float4 ra[PD];
    float4 rw[PD][CT];
    auto load = [&](int slot, int ch) {
        ra[slot] = sum_parts<SP>(hp, hp_stride, 16 * (ch0 + ch) + 4 * lq);
#pragma unroll
        for (int c = 0; c < CT; ++c) rw[slot][c] = reinterpret_cast<const float4*>(Wfrag)[((size_t)(cb * CT + c) * nchunks + ch0 + ch) * 64 + lane];
    };
#pragma unroll
    for (int sl = 0; sl < PD; ++sl)
        if (sl < nch) load(sl, sl);
    float ss = 0.f;
    for (int base = 0; base < nch; base += PD) {
#pragma unroll
        for (int sl = 0; sl < PD; ++sl) {
            if (base + sl < nch) {
                ss += f4dot(ra[sl], ra[sl]);
                const float4 a = to_mfma_lanes(psrc, ra[sl]);
#pragma unroll
                for (int c = 0; c < CT; ++c) acc[c] = mfma16(rw[sl][c].x, a.x, acc[c]);
#pragma unroll
                for (int c = 0; c < CT; ++c) acc[c] = mfma16(rw[sl][c].y, a.y, acc[c]);
#pragma unroll
                for (int c = 0; c < CT; ++c) acc[c] = mfma16(rw[sl][c].z, a.z, acc[c]);
#pragma unroll
                for (int c = 0; c < CT; ++c) acc[c] = mfma16(rw[sl][c].w, a.w, acc[c]);
                if (base + sl + PD < nch) load(sl, base + sl + PD);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < CT; ++c) part[wave][c][lane] = make_float4(acc[c][0], acc[c][1], acc[c][2], acc[c][3]);
    ss += __shfl_xor(ss, 1);                 // the four fetch lanes of a row
    ss += __shfl_xor(ss, 2);
    if (lq == 0) sh_ss[wave][li] = ss;
    __syncthreads();
    auto den_of = [&](int r16, float* raw) {
        const float nr = sqrtf(((sh_ss[0][r16] + sh_ss[1][r16]) + sh_ss[2][r16]) + sh_ss[3][r16]);
        if (raw) *raw = nr;
        return normalize ? fmaxf(nr, UNIT_EPS) : 1.f;
    };
    const float den = den_of(i, nullptr);
    const int row = rg * 16 + i;
    for (int t = wave; t < CT; t += 4) {
        const float4 p0 = part[0][t][lane], p1 = part[1][t][lane], p2 = part[2][t][lane], p3 = part[3][t][lane];
        float4 v = make_float4(((p0.x + p1.x) + p2.x) + p3.x, ((p0.y + p1.y) + p2.y) + p3.y,
                               ((p0.z + p1.z) + p2.z) + p3.z, ((p0.w + p1.w) + p2.w) + p3.w);
        v = make_float4(v.x / den, v.y / den, v.z / den, v.w / den);
        const int col = col0 + t * 16 + 4 * q;
        if (bias) v = f4add(v, ld4(bias + col));
        if (row < ncell) st4(P + crow_of(row) * ldp + col, v);
    }
    if (cb == 0 && H) {                       // chart output of the level: wave w writes rows w, w+4, ...
        // all of the wave's rows are fetched before the first is written: one row after the other (fetch, divide, store, next) made
        // these blocks -- one in ncolblocks -- a chain of eight dependent round trips, and the launch waits for its slowest block
        const int nv = K >> 2;
        float4 a[4][2];
        size_t crow[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = rg * 16 + wave + 4 * k;
            crow[k] = crow_of(r);
            const float* src = HP + crow[k] * K;
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int v4 = lane + 64 * it;
                a[k][it] = (r < ncell && v4 < nv) ? sum_parts<SP>(src, hp_stride, 4 * v4) : f4zero();
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int rr = wave + 4 * k, r = rg * 16 + rr;
            if (r >= ncell) break;
            float nr;
            const float d = den_of(rr, &nr);
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int v4 = lane + 64 * it;
                if (v4 < nv) st4(H + crow[k] * K + 4 * v4, make_float4(a[k][it].x / d, a[k][it].y / d, a[k][it].z / d, a[k][it].w / d));
            }
            if (lane == 0) nrm[crow[k]] = nr;
        }
    }
}

template <int CT, int SP>
__global__ __launch_bounds__(256) void level_project(const float* __restrict__ Wfrag, int K, int nrg, int nrgp, int ncolblocks,
                                                     int ncell, int Lc, int C, int off, const float* __restrict__ HP, size_t hp_stride,
                                                     int normalize, const float* __restrict__ bias, float* __restrict__ P, int ldp,
                                                     float* __restrict__ H, float* __restrict__ nrm, ScoreArgs sc) {
    __shared__ float4 part[4][CT][64];
    __shared__ float sh_ss[4][16];
    (void)ncolblocks;
    if ((int)blockIdx.x < sc.nscore) {               // the next level's scores: first in the grid, so they start at once
        score_cell(sc, cell_of_block(blockIdx.x, sc.g.B, sc.g.Lc, sc.g.affine), &sh_ss[0][0]);
        return;
    }
    project_block<CT, SP>(part, sh_ss, (int)blockIdx.x - sc.nscore, Wfrag, K, nrg, nrgp, ncell, Lc, C, off, HP, hp_stride, normalize, bias, P, ldp, H, nrm);
}

// levels whose cells need no projection (inside root, outside leaves): sum the partial aggregates, unit norm, chart row.
// Four cells per block (one per wave): block `blk` of cells_grid(ncell).
__device__ __forceinline__ void finish_cells(const int blk, int ncell, int Lc, int C, int off, int Dp, const float* __restrict__ HP,
                                             size_t hp_stride, int SP, int normalize, float* __restrict__ H, float* __restrict__ nrm) {
    const int lane = threadIdx.x & 63;
    const int r = blk * 4 + (threadIdx.x >> 6);
    if (r >= ncell) return;
    const int b = r / Lc;
    const size_t crow = (size_t)b * C + off + (r - b * Lc);
    const int nv = Dp >> 2;
    float4 v0 = f4zero(), v1 = f4zero();
    for (int s = 0; s < SP; ++s) {
        const float* src = HP + (size_t)s * hp_stride + crow * Dp;
        if (lane < nv) v0 = f4add(v0, ld4(src + 4 * lane));
        if (lane + 64 < nv) v1 = f4add(v1, ld4(src + 4 * (lane + 64)));
    }
    const float nr = sqrtf(wave_sum(f4dot(v0, v0) + f4dot(v1, v1)));
    const float den = normalize ? fmaxf(nr, UNIT_EPS) : 1.f;
    float* h = H + crow * Dp;
    if (lane < nv) st4(h + 4 * lane, make_float4(v0.x / den, v0.y / den, v0.z / den, v0.w / den));
    if (lane + 64 < nv) st4(h + 4 * (lane + 64), make_float4(v1.x / den, v1.y / den, v1.z / den, v1.w / den));
    if (lane == 0) nrm[crow] = nr;
}
static __global__ __launch_bounds__(256) void level_finish(int ncell, int Lc, int C, int off, int Dp, const float* __restrict__ HP,
                                                    size_t hp_stride, int SP, int normalize, float* __restrict__ H,
                                                    float* __restrict__ nrm) {
    finish_cells(blockIdx.x, ncell, Lc, C, off, Dp, HP, hp_stride, SP, normalize, H, nrm);
}

// project_block with RT x CT sixteen-wide tiles per workgroup (round 5).  The per-level projections are bound by what a block pulls through
// L2 -- a 16 x 16 block streams 25.6 KB of rows and 25.6 KB of weight fragments for ONE tile (292 MB per launch at level 1 of c2, measured
// 16.7 us per launch with the scores off, profiles/r05_notes.md) -- so a block of RT row tiles x CT column tiles moves (RT + CT) / (RT CT)
// of that per tile.  Every output element is still summed in the same order (four waves split the reduction, fixed LDS tree): the
// same bits for any tile shape.  Ragged last column block (ntc column tiles in all); sh_ss: [4][RT * 16].
template <int RT, int CT, int SP, int PD>
__device__ __forceinline__ void project_tile(float4 (*part)[RT * CT][64], float (*sh_ss)[RT * 16], const int bid, const float* __restrict__ Wfrag, int K,
                                             int nrg, int nrgp, int ntc, int ncell, int Lc, int C, int off, const float* __restrict__ HP,
                                             size_t hp_stride, int normalize, const float* __restrict__ bias, float* __restrict__ P, int ldp,
                                             float* __restrict__ H, float* __restrict__ nrm) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 15, q = lane >> 4;
    const int li = fetch_row_of(lane), lq = fetch_piece_of(lane), psrc = mfma_src_addr(lane);
    const int cb = bid / nrgp, rg = bid - cb * nrgp;
    if (rg >= nrg) return;
    const int ct0 = cb * CT;                               // first column tile of the block
    const int nct = min(CT, ntc - ct0);                    // ragged last block (workgroup-uniform)
    const int nchunks = K >> 4;
    const int cbase = nchunks / 4, crem = nchunks % 4;
    const int ch0 = wave * cbase + min(wave, crem);
#ifdef CLIORA_DIAG_GEMMK                               // wrong-result timing diagnostic (tools/ab/anatomy.sh): 1/N of each wave's chunks
    const int nch = max(1, (cbase + (wave < crem ? 1 : 0)) / CLIORA_DIAG_GEMMK);
#else
    const int nch = cbase + (wave < crem ? 1 : 0);
#endif
    auto crow_of = [&](int r) { const int rc = min(r, ncell - 1); const int b = rc / Lc; return (size_t)b * C + off + (rc - b * Lc); };
    const float* hp[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) hp[r] = HP + crow_of((rg * RT + r) * 16 + li) * K;
    f32x4 acc[RT][CT];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 ra[PD][RT];
    float4 rw[PD][CT];
    auto load = [&](int slot, int ch) {
#pragma unroll
        for (int r = 0; r < RT; ++r) ra[slot][r] = sum_parts<SP>(hp[r], hp_stride, 16 * (ch0 + ch) + 4 * lq);
#pragma unroll
        for (int c = 0; c < CT; ++c)
            if (c < nct) rw[slot][c] = reinterpret_cast<const float4*>(Wfrag)[((size_t)(ct0 + c) * nchunks + ch0 + ch) * 64 + lane];
    };
#pragma unroll
    for (int sl = 0; sl < PD; ++sl)
        if (sl < nch) load(sl, sl);
    float ss[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) ss[r] = 0.f;
    for (int base = 0; base < nch; base += PD) {
#pragma unroll
        for (int sl = 0; sl < PD; ++sl) {
            if (base + sl < nch) {
                float4 a[RT];
#pragma unroll
                for (int r = 0; r < RT; ++r) { ss[r] += f4dot(ra[sl][r], ra[sl][r]); a[r] = to_mfma_lanes(psrc, ra[sl][r]); }
#pragma unroll
                for (int r = 0; r < RT; ++r)
#pragma unroll
                    for (int c = 0; c < CT; ++c) if (c < nct) acc[r][c] = mfma16(rw[sl][c].x, a[r].x, acc[r][c]);
#pragma unroll
                for (int r = 0; r < RT; ++r)
#pragma unroll
                    for (int c = 0; c < CT; ++c) if (c < nct) acc[r][c] = mfma16(rw[sl][c].y, a[r].y, acc[r][c]);
#pragma unroll
                for (int r = 0; r < RT; ++r)
#pragma unroll
                    for (int c = 0; c < CT; ++c) if (c < nct) acc[r][c] = mfma16(rw[sl][c].z, a[r].z, acc[r][c]);
#pragma unroll
                for (int r = 0; r < RT; ++r)
#pragma unroll
                    for (int c = 0; c < CT; ++c) if (c < nct) acc[r][c] = mfma16(rw[sl][c].w, a[r].w, acc[r][c]);
                if (base + sl + PD < nch) load(sl, base + sl + PD);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < RT; ++r) {
#pragma unroll
        for (int c = 0; c < CT; ++c) part[wave][r * CT + c][lane] = make_float4(acc[r][c][0], acc[r][c][1], acc[r][c][2], acc[r][c][3]);
        float s_ = ss[r];
        s_ += __shfl_xor(s_, 1);                 // the four fetch lanes of a row
        s_ += __shfl_xor(s_, 2);
        if (lq == 0) sh_ss[wave][r * 16 + li] = s_;
    }
    __syncthreads();
    auto den_of = [&](int r16, float* raw) {
        const float nr = sqrtf(((sh_ss[0][r16] + sh_ss[1][r16]) + sh_ss[2][r16]) + sh_ss[3][r16]);
        if (raw) *raw = nr;
        return normalize ? fmaxf(nr, UNIT_EPS) : 1.f;
    };
    for (int t = wave; t < RT * CT; t += 4) {
        const int r = t / CT, c = t - r * CT;
        if (c >= nct) continue;
        const float den = den_of(r * 16 + i, nullptr);
        const int row = (rg * RT + r) * 16 + i;
        const float4 p0 = part[0][t][lane], p1 = part[1][t][lane], p2 = part[2][t][lane], p3 = part[3][t][lane];
        float4 v = make_float4(((p0.x + p1.x) + p2.x) + p3.x, ((p0.y + p1.y) + p2.y) + p3.y,
                               ((p0.z + p1.z) + p2.z) + p3.z, ((p0.w + p1.w) + p2.w) + p3.w);
        v = make_float4(v.x / den, v.y / den, v.z / den, v.w / den);
        const int col = (ct0 + c) * 16 + 4 * q;
        if (bias) v = f4add(v, ld4(bias + col));
        if (row < ncell) st4(P + crow_of(row) * ldp + col, v);
    }
    if (cb == 0 && H) {                       // chart output of the level (see project_block): wave w writes rows w, w+4, ... of each row tile
        const int nv = K >> 2;
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            float4 a[4][2];
            size_t crow[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int rr = (rg * RT + r) * 16 + wave + 4 * k;
                crow[k] = crow_of(rr);
                const float* src = HP + crow[k] * K;
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int v4 = lane + 64 * it;
                    a[k][it] = (rr < ncell && v4 < nv) ? sum_parts<SP>(src, hp_stride, 4 * v4) : f4zero();
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r16 = wave + 4 * k, rr = (rg * RT + r) * 16 + r16;
                if (rr >= ncell) break;
                float nr;
                const float d = den_of(r * 16 + r16, &nr);
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int v4 = lane + 64 * it;
                    if (v4 < nv) st4(H + crow[k] * K + 4 * v4, make_float4(a[k][it].x / d, a[k][it].y / d, a[k][it].z / d, a[k][it].w / d));
                }
                if (lane == 0) nrm[crow[k]] = nr;
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// level_project2 (round 5): the second launch of a wavefront step for BOTH passes -- norm + projection of inside level k and of outside
// level L-k, the chart rows of a level that needs no projection (level_finish: inside root, outside leaves), and the split scores +
// softmax of the next level of either pass -- as ONE grid on one queue.  Block order: the scores first (they are the latency chains of
// the launch), then segment 0's projection blocks, segment 1's, then the finish blocks.  16 x 16 tiles (CT = 1: the variant every
// sweep of the block size preferred, profiles/r04_notes.md); SP0 / SP1 = parts of the two levels' partial aggregates.  Every block runs
// the code of level_project / level_finish on its own segment: the same bits as the per-pass launches.
// ---------------------------------------------------------------------------------
struct ProjSeg {
    const float* Wfrag; int K, nrg, nrgp, ntc, ncell, Lc, C, off;      // nrg row groups of P2_RT row tiles (nrgp: padded to 8), ntc column tiles
    const float* HP; size_t hp_stride; int normalize; const float* bias; float* P; int ldp; float* H; float* nrm;
    int nproj;            // projection blocks (nrgp * ceil(ntc / column tiles per block)); 0: none
    int nfin, SPfin;      // level_finish blocks (cells_grid(ncell)) and the level's parts; 0: none
    ScoreArgs sc;         // sc.nscore = 0: no scoring
};
// tile shape of level_project2's projection blocks: RT row tiles x CTA (segment 0: inside, 75 column tiles at d 400) / CTB (segment 1:
// outside, 25) column tiles, PD chunks of the reduction in flight per wave.  Build-time (tools/ab: -DCLIORA_P2_RT=.. etc.)
#ifndef CLIORA_P2_RT
#define CLIORA_P2_RT 1
#endif
#ifndef CLIORA_P2_CTA
#define CLIORA_P2_CTA 1
#endif
#ifndef CLIORA_P2_CTB
#define CLIORA_P2_CTB 1
#endif
#ifndef CLIORA_P2_PD
#define CLIORA_P2_PD 4
#endif
constexpr int P2_RT = CLIORA_P2_RT, P2_CTA = CLIORA_P2_CTA, P2_CTB = CLIORA_P2_CTB, P2_PD = CLIORA_P2_PD;
constexpr int P2_CTM = P2_CTA > P2_CTB ? P2_CTA : P2_CTB;
template <int SP0, int SP1>
__global__ __launch_bounds__(256) void level_project2(ProjSeg a, ProjSeg b, int order) {
    __shared__ float4 part[4][P2_RT * P2_CTM][64];
    __shared__ float sh_ss[4][P2_RT * 16];
    int bid = blockIdx.x;
    {   // order 1: the projection blocks first, the score blocks behind them (order 0: scores first); 2: scores in the middle
        const int ns = a.sc.nscore + b.sc.nscore, np = a.nproj + b.nproj;
        if (order == 1 && bid < ns + np) bid = bid < np ? bid + ns : bid - np;
        else if (order == 2 && bid < ns + np) { const int h = np / 2; bid = bid < h ? bid + ns : (bid < h + ns ? bid - h : bid); }
    }
    if (bid < a.sc.nscore) { score_cell(a.sc, cell_of_block(bid, a.sc.g.B, a.sc.g.Lc, a.sc.g.affine), &sh_ss[0][0]); return; }
    bid -= a.sc.nscore;
    if (bid < b.sc.nscore) { score_cell(b.sc, cell_of_block(bid, b.sc.g.B, b.sc.g.Lc, b.sc.g.affine), &sh_ss[0][0]); return; }
    bid -= b.sc.nscore;
    if (bid < a.nproj) {
        project_tile<P2_RT, P2_CTA, SP0, P2_PD>(reinterpret_cast<float4 (*)[P2_RT * P2_CTA][64]>(&part[0][0][0]), sh_ss, bid, a.Wfrag, a.K, a.nrg, a.nrgp, a.ntc,
                                                a.ncell, a.Lc, a.C, a.off, a.HP, a.hp_stride, a.normalize, a.bias, a.P, a.ldp, a.H, a.nrm);
        return;
    }
    bid -= a.nproj;
    if (bid < b.nproj) {
        project_tile<P2_RT, P2_CTB, SP1, P2_PD>(reinterpret_cast<float4 (*)[P2_RT * P2_CTB][64]>(&part[0][0][0]), sh_ss, bid, b.Wfrag, b.K, b.nrg, b.nrgp, b.ntc,
                                                b.ncell, b.Lc, b.C, b.off, b.HP, b.hp_stride, b.normalize, b.bias, b.P, b.ldp, b.H, b.nrm);
        return;
    }
    bid -= b.nproj;
    if (bid < a.nfin) { finish_cells(bid, a.ncell, a.Lc, a.C, a.off, a.K, a.HP, a.hp_stride, a.SPfin, a.normalize, a.H, a.nrm); return; }
    bid -= a.nfin;
    if (bid < b.nfin) finish_cells(bid, b.ncell, b.Lc, b.C, b.off, b.K, b.HP, b.hp_stride, b.SPfin, b.normalize, b.H, b.nrm);
}

// ---------------------------------------------------------------------------------
// Backward of one level's compose layer.  Per tile (16 target cells t, one split n), with dG = d loss / d g (the aggregate):
//   A rows    dzu = dG[t] masked by the ReLU bits of y_n            (K = the z dimension; p_n is applied afterwards)
//   GEMM      u = dzu W2                                              (W2^T block stays in LDS)
//   epilogue  xs = PL(a) + PR(b) for the block's columns (re-gathered: x is not kept by the forward), x = relu(xs)
//             DA  = p_n u [xs > 0]      -> the per-pair gradient the cell gathers sum (cell_gather_bwd_*)
//             X   = x, DZ = p_n dzu     -> operands of the weight gradient dW2 = DZ^T X (DZ from the fetch lanes, 1/gridDim.y each)
//             DPP[row][block] = sum over the block's columns of u x, DPB[row] = dzu . b2 (block 0, from the fetch lanes):
//                                since y = relu(z), dG . y_n = (dG masked) . z_n = u . x_n + (dG masked) . b2
//                                -- the softmax backward gets its dp_n without y_n (cell_dsoftmax)
// Tiles are independent here (no reduction over the splits): waves walk the level's tiles (g-major) with a stride.
// ---------------------------------------------------------------------------------
// TILED: X and DZ leave as 16-row tiles of bf16 hi / lo planes (wgrad_tiles.hpp: the operand image of tn_gemm_tiles; storage tile =
// lv.tilebase + the wave tile's index; rows past the level's last cell are zeros) instead of fp32 rows.
template <int CT, int K16, bool F32, bool TILED = false>
__global__ __launch_bounds__(512) void level_compose_bwd(const uint32_t* __restrict__ Wimg, int S_, int K_, PairLevel lv,
                                                         const float* __restrict__ dG, const uint32_t* __restrict__ ymask,
                                                         const float* __restrict__ Pp, const float* __restrict__ PA, int lda,
                                                         const float* __restrict__ PB, int ldb, const float* __restrict__ b2, int Dp,
                                                         float* __restrict__ DA, float* __restrict__ DZ, float* __restrict__ X,
                                                         float* __restrict__ DPP, float* __restrict__ DPB) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_img[];
#ifndef CLIORA_CB_PD
#define CLIORA_CB_PD 4
#endif
    constexpr int WAVES = 8, T = WAVES * 64, PD = CLIORA_CB_PD;       // k-steps of operand rows in flight per wave
    constexpr bool KS = K16 > 0;
    constexpr int UNROLL_STEPS = KS ? 64 : 1;
    const int K = KS ? K16 * 16 : K_;
    const int S = F32 ? K : (KS ? (K16 + 1) / 2 * 32 + WS3_PAD : S_);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 15, g = lane >> 4;
    const int li = fetch_row_of(lane), lg = fetch_piece_of(lane), psrc = mfma_src_addr(lane);
    const int fsrc = 4 * (li + 16 * lg);             // ds_bpermute address of the MFMA lane (row li, piece lg) for the fetch lane 4 li + lg
    const int Kp = F32 ? (K + 31) / 32 * 32 : S - WS3_PAD, half = Kp >> 1;
    const int by = blockIdx.y, gy = gridDim.y;
    const int col0 = by * (CT * 16);
    stage_weight_image(Wimg + (size_t)col0 * S, lds_img, CT * 16 * S, wave, lane, T);
    // b2 next to the image (Kp floats, zero beyond K): block 0 reads it per k-step for the bias term of dG . y_n -- from LDS, so
    // that no fresh global load sits inside the loop (vmcnt is in order: waiting for one would drain the operand ring)
    float* b2s = reinterpret_cast<float*>(lds_img + CT * 16 * S);
    for (int k = threadIdx.x; k < Kp; k += T) b2s[k] = k < K ? b2[k] : 0.f;
    const int nsteps = Kp >> 5;
    const int nsteps_p = (nsteps + PD - 1) / PD * PD;
    int wimg_off = 0;
    const int G = (lv.ncell + 15) >> 4;
    const int ntiles = G * lv.N;
    const int stride = gridDim.x * WAVES;

    struct Ctx { const float* gp; const uint32_t* mp; float* zo; float pn; float* bo; };
    const int NT = Dp >> 4;
    auto rowctx = [&](int tile) {                    // fetch-lane view of tile = gt * N + n
        const int gt = tile / lv.N, n = tile - gt * lv.N;
        const int t = min(gt * 16 + li, lv.ncell - 1);
        const int b = t / lv.Lc, p = t - b * lv.Lc;
        const size_t prow = (size_t)lv.rowbase + (size_t)t * lv.N + n;
        if (TILED)       // the lane's 8 bytes of the tile's hi planes (row li, columns 4 lg ..): lane-linear; zero rows past the last cell
            return Ctx{dG + ((size_t)b * lv.C + lv.off + p) * Dp, ymask + prow * gy * 4 + lg,
                       DZ + (size_t)(lv.tilebase + tile) * NT * 256 + 2 * lane, gt * 16 + li < lv.ncell ? Pp[prow] : 0.f, DPB + prow};
        return Ctx{dG + ((size_t)b * lv.C + lv.off + p) * Dp, ymask + prow * gy * 4 + lg, DZ + prow * Dp, Pp[prow], DPB + prow};
    };
    struct Slot { float4 g0, g1; uint32_t m0, m1; };
    Slot ra[PD];
    auto issue = [&](int slot, const Ctx& c, int s) {
        const int k = 32 * s + 4 * lg;
        const bool second = 32 * s + 16 < K;
        const int k2 = k + (second ? 16 : 0);
        ra[slot].g0 = ld4(c.gp + k);
        ra[slot].g1 = ld4(c.gp + k2);
        // column k of z lives in column block k / (CT*16), 16-column tile (k / 16) % CT, word g = (k % 16) / 4 = lg
        ra[slot].m0 = c.mp[((2 * s) / CT) * 4];
        ra[slot].m1 = c.mp[((2 * s + (second ? 1 : 0)) / CT) * 4];
    };
    auto masked = [](const float4& v, uint32_t nib) {
        return make_float4((nib & 1u) ? v.x : 0.f, (nib & 2u) ? v.y : 0.f, (nib & 4u) ? v.z : 0.f, (nib & 8u) ? v.w : 0.f);
    };

    int tile = blockIdx.x * WAVES + wave;
    const bool work = tile < ntiles;
    Ctx ctx = rowctx(work ? tile : 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (!work) return;
#pragma unroll
    for (int sl = 0; sl < PD; ++sl) issue(sl, ctx, sl < nsteps ? sl : 0);
    while (true) {
        const int ntile = tile + stride;
        const bool has_next = ntile < ntiles;
        const Ctx ctxn = rowctx(has_next ? ntile : tile);
        // The epilogue runs in the FETCH-lane view (lane = 4 li + lg: row li = target cell ti, columns 4 lg .. of each column tile), like
        // the operand ring: its re-gather of xs = PL(a) + PR(b), the DA row stores and the X tile stores are then 64 contiguous bytes per
        // four lanes (the X tile: lane-linear) where the MFMA-lane view (row = lane & 15) made every one of them 64 separate requests per
        // instruction -- the re-gather alone was 4.7 us of the c2 launch's 31 (tools/ab/anatomy.sh).  The accumulators come over with
        // five ds_bpermute quadruples per tile (the k-loop has 26).
        const int gt = tile / lv.N, n = tile - gt * lv.N;
        const int ti = gt * 16 + li;
        const bool ok = ti < lv.ncell;
        const int tc = min(ti, lv.ncell - 1);
        const int eb = tc / lv.Lc, ep = tc - eb * lv.Lc;
        const size_t prow = (size_t)lv.rowbase + (size_t)tc * lv.N + n;
        const float pn = Pp[prow];
        const float* xa = PA + ((size_t)eb * lv.C + lv.pa[ep * lv.N + n]) * lda + col0 + 4 * lg;
        const float* xb = PB + ((size_t)eb * lv.C + lv.pb[ep * lv.N + n]) * ldb + col0 + 4 * lg;
        float4 ea[CT], ebv[CT];                     // the epilogue's operands, on their way while the K loop runs
#pragma unroll
        for (int c = 0; c < CT; ++c) { ea[c] = ld4(xa + c * 16); ebv[c] = ld4(xb + c * 16); }
        __builtin_amdgcn_sched_barrier(0);
        f32x4 acc[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        asm volatile("" : "+v"(wimg_off));
        const uint32_t* wimg = lds_img + wimg_off;
        float bdot = 0.f;                           // block 0: (dG masked by y_n) . b2 over this lane's k pieces
        // operand of k-step st from its ring slot: mask by the ReLU bits, this block's share of the DZ rows, block 0's bias dot
        auto prep = [&](int st, const Slot& q) {
            const bool second = 32 * st + 16 < K;
            const float4 f0 = masked(q.g0, q.m0 >> (4 * ((2 * st) % CT)));
            const float4 f1 = masked(q.g1, q.m1 >> (4 * ((2 * st + (second ? 1 : 0)) % CT)));
            const int k = 32 * st + 4 * lg;
#ifdef CLIORA_DIAG_NOXZ      // timing proxy (WRONG weight gradients): no X / DZ tile stores (profiles/r06_notes.md section 4)
            if (false) {
#else
            if (st % gy == by) {
#endif
                if (TILED) {
                    uint32_t* zt = reinterpret_cast<uint32_t*>(ctx.zo) + (2 * st) * 256;
                    store_split_tile(zt, make_float4(ctx.pn * f0.x, ctx.pn * f0.y, ctx.pn * f0.z, ctx.pn * f0.w));
                    if (second) store_split_tile(zt + 256, make_float4(ctx.pn * f1.x, ctx.pn * f1.y, ctx.pn * f1.z, ctx.pn * f1.w));
                } else {
                    st4(ctx.zo + k, make_float4(ctx.pn * f0.x, ctx.pn * f0.y, ctx.pn * f0.z, ctx.pn * f0.w));
                    if (second) st4(ctx.zo + k + 16, make_float4(ctx.pn * f1.x, ctx.pn * f1.y, ctx.pn * f1.z, ctx.pn * f1.w));
                }
            }
            if (by == 0) {
                bdot += f4dot(f0, *reinterpret_cast<const float4*>(b2s + k));
                if (second) bdot += f4dot(f1, *reinterpret_cast<const float4*>(b2s + k + 16));
            }
            return make_operand<F32>(psrc, f0, f1);
        };
        // software pipeline over the k-steps, as in level_compose_fwd
        StepOperand cur = prep(0, ra[0]);
#pragma unroll UNROLL_STEPS
        for (int base = 0; base < nsteps_p; base += PD) {
#pragma unroll
            for (int sl = 0; sl < PD; ++sl) {
                const int st = base + sl;
                if (st < nsteps) {
                    StepOperand nxt = cur;
                    if (st + 1 < nsteps) nxt = prep(st + 1, ra[(sl + 1) % PD]);
                    kstep_mfma<CT, F32>(wimg, i, g, S, half, st, 32 * st + 16 < K, cur, acc);
                    const int nst = st + PD;
                    const bool in_cur = nst < nsteps;
                    issue(sl, pick_pod(in_cur, ctx, ctxn), in_cur ? nst : (sl < nsteps ? sl : 0));
                    __builtin_amdgcn_sched_barrier(0);
                    cur = nxt;
                }
            }
        }
        // epilogue
        if (by == 0) {                               // the four fetch lanes of a row hold its k pieces
            bdot += __shfl_xor(bdot, 1);
            bdot += __shfl_xor(bdot, 2);
            if (lg == 0 && gt * 16 + li < lv.ncell) *ctx.bo = bdot;
        }
        float dp = 0.f;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            // u for (row li, columns 4 lg ..): MFMA lane li + 16 lg holds it
            const float4 u = to_mfma_lanes(fsrc, make_float4(acc[c][0], acc[c][1], acc[c][2], acc[c][3]));
            const float4 pa4 = ea[c], pb4 = ebv[c];
            const float xs0 = pa4.x + pb4.x, xs1 = pa4.y + pb4.y, xs2 = pa4.z + pb4.z, xs3 = pa4.w + pb4.w;
            const float4 x = make_float4(fmaxf(xs0, 0.f), fmaxf(xs1, 0.f), fmaxf(xs2, 0.f), fmaxf(xs3, 0.f));
            dp = fmaf(u.x, x.x, dp); dp = fmaf(u.y, x.y, dp); dp = fmaf(u.z, x.z, dp); dp = fmaf(u.w, x.w, dp);
            if (ok) {
                const size_t o = prow * Dp + col0 + c * 16 + 4 * lg;
                st4(DA + o, make_float4(x.x > 0.f ? pn * u.x : 0.f, x.y > 0.f ? pn * u.y : 0.f,
                                        x.z > 0.f ? pn * u.z : 0.f, x.w > 0.f ? pn * u.w : 0.f));
                if (!TILED) st4(X + o, x);
            }
#ifndef CLIORA_DIAG_NOXZ
            if (TILED)          // the tile's lane slot 8 (row) + 2 (piece) = 2 lane: lane-linear
#else
            if (false)
#endif
                store_split_tile(reinterpret_cast<uint32_t*>(X) + ((size_t)(lv.tilebase + tile) * NT + by * CT + c) * 256 + 2 * lane,
                                 ok ? x : f4zero());
        }
        dp += __shfl_xor(dp, 1);                    // the four lanes of a row: (g0 + g1) + (g2 + g3), the order the MFMA-lane view summed in
        dp += __shfl_xor(dp, 2);
        if (ok && lg == 0) DPP[prow * gy + by] = dp;
        if (!has_next) break;
        ctx = ctxn;
        tile = ntile;
    }
}

// dG = unit-norm backward of the level's cells: the gradient with respect to the aggregate g (H = g / max(||g||, eps)).
// One wave per cell.  CLIORA passes u = unit(g) and ||g|| (the attention residual sits between g and H).
static __global__ __launch_bounds__(256) void cell_dnorm(LevelArgs g, const float* __restrict__ VH, const float* __restrict__ H,
                                                         const float* __restrict__ nrm, int normalize, float* __restrict__ dG) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= g.B * g.Lc) return;
    const int b = t / g.Lc, p = t - b * g.Lc;
    const size_t crow = (size_t)b * g.C + g.off + p;
    const int Dp = g.Dp, nv = Dp >> 2;
    const bool a0 = lane < nv, a1 = lane + 64 < nv;
    float4 v0 = f4zero(), v1 = f4zero(), h0 = f4zero(), h1 = f4zero();
    if (a0) { v0 = ld4(VH + crow * Dp + 4 * lane); h0 = ld4(H + crow * Dp + 4 * lane); }
    if (a1) { v1 = ld4(VH + crow * Dp + 4 * (lane + 64)); h1 = ld4(H + crow * Dp + 4 * (lane + 64)); }
    unit_norm_bwd(v0, v1, h0, h1, nrm[crow], normalize);
    if (a0) st4(dG + crow * Dp + 4 * lane, v0);
    if (a1) st4(dG + crow * Dp + 4 * (lane + 64), v1);
}

// Softmax / score backward of the level's cells (diora.py:125-149 differentiated), one wave per cell:
//   dp_n = sum_blocks DPP[row][block] + DPB[row]                        (= dG . y_n, see level_compose_bwd)
//   ds_n = p_n [ (dp_n - sum_m p_m dp_m) + dS_tot (1 + s_n - S) ]
static __global__ __launch_bounds__(256) void cell_dsoftmax(LevelArgs g, int ncb, const float* __restrict__ DPP, const float* __restrict__ DPB,
                                                            const float* __restrict__ Sp, const float* __restrict__ Pp,
                                                            const float* __restrict__ Schart, const float* __restrict__ dStot,
                                                            float* __restrict__ DS) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= g.B * g.Lc || g.N == 0) return;
    const int b = t / g.Lc, p = t - b * g.Lc;
    const size_t crow = (size_t)b * g.C + g.off + p;
    const size_t row0 = (size_t)g.rowbase + (size_t)t * g.N;
    const bool an = lane < g.N;
    float dp = an ? DPB[row0 + lane] : 0.f;
    if (an)
        for (int cb = 0; cb < ncb; ++cb) dp += DPP[(row0 + lane) * ncb + cb];
    const float pn = an ? Pp[row0 + lane] : 0.f;
    const float sn = an ? Sp[row0 + lane] : 0.f;
    const float mean = wave_sum(pn * dp);
    // 1 + (s_n - S), in that order: the scores reach 1e8 without unit normalisation, where (1 + s_n) - S loses the 1 (found by
    // tools/fuzz_parity.py: every gradient through outside_s of a cell with |S| > 2^24 vanished)
    const float ds = pn * ((dp - mean) + dStot[crow] * (1.f + (sn - Schart[crow])));
    if (an) DS[row0 + lane] = ds;
}

}  // namespace cliora
