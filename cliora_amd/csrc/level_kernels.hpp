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
// The same device code runs as phases of ONE persistent launch in persist_kernels.hpp (cliora_set_persistent), bitwise equal.
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

template <int CT, int K16, bool F32>
__global__ __launch_bounds__(512) void level_compose_fwd(const uint32_t* __restrict__ Wimg, int S_, int K_, PairLevel lv,
                                                         const float* __restrict__ PA, int lda, const float* __restrict__ PB, int ldb,
                                                         const float* __restrict__ bias, const float* __restrict__ Pp,
                                                         int TG, int SP, int ntask, float* __restrict__ HP, size_t hp_stride, int Dp,
                                                         uint32_t* __restrict__ ymask, float* __restrict__ Y) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_img[];
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
    for (int task = blockIdx.x; task < ntask; task += gridDim.x) {
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
__device__ __forceinline__ void score_cell(const ScoreArgs& sc, int t, float* sh_s) {
    const LevelArgs& g = sc.g;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = t / g.Lc, p = t - b * g.Lc;
    const int row0 = g.rowbase + t * g.N;
    const int Dp = g.Dp, nv = Dp >> 2;
    const bool a0 = lane < nv, a1 = lane + 64 < nv;
    const int c0 = 4 * lane, c1 = 4 * (lane + 64);
    const int bC = b * g.C;
    for (int n0 = wave; n0 < g.N; n0 += 16) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + 4 * j;
            if (n >= g.N) break;                     // wave-uniform
            const int ca = sc.pa[p * g.N + n], cb = sc.pb[p * g.N + n];
            const int ar = bC + ca, br = bC + cb;
            const bool a_new = sc.a_can_be_new && ca >= sc.new_lo && ca < sc.new_hi;
            const bool b_new = cb >= sc.new_lo && cb < sc.new_hi;
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
            if (a_new) {                               // partner is a leaf: QR(leaf) = M h_b
                newest(ar, u0, u1);
                const float* qr = sc.QRleaf + ((size_t)b * sc.L + cb) * Dp;
                if (a0) v0 = ld4(qr + c0);
                if (a1) v1 = ld4(qr + c1);
            } else {
                const float* qa = sc.QA + (size_t)ar * sc.ldA;
                if (a0) u0 = ld4(qa + c0);
                if (a1) u1 = ld4(qa + c1);
                if (b_new) newest(br, v0, v1);
                else {
                    const float* hb = sc.HB + (size_t)br * Dp;
                    if (a0) v0 = ld4(hb + c0);
                    if (a1) v1 = ld4(hb + c1);
                }
            }
            const float s = wave_sum(f4dot(u0, v0) + f4dot(u1, v1)) / den + sc.SA[ar] + sc.SB[br];
            if (lane == 0) sh_s[n] = s;
        }
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
    score_cell(sc, blockIdx.x, sh_s);
}

template <int CT, int SP>
__global__ __launch_bounds__(256) void level_project(const float* __restrict__ Wfrag, int K, int nrg, int nrgp, int ncolblocks,
                                                     int ncell, int Lc, int C, int off, const float* __restrict__ HP, size_t hp_stride,
                                                     int normalize, const float* __restrict__ bias, float* __restrict__ P, int ldp,
                                                     float* __restrict__ H, float* __restrict__ nrm, ScoreArgs sc) {
    __shared__ float4 part[4][CT][64];
    __shared__ float sh_ss[4][16];
    if ((int)blockIdx.x < sc.nscore) {               // the next level's scores: first in the grid, so they start at once
        score_cell(sc, blockIdx.x, &sh_ss[0][0]);
        return;
    }
    const int bid = blockIdx.x - sc.nscore;
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

// levels whose cells need no projection (inside root, outside leaves): sum the partial aggregates, unit norm, chart row.
static __global__ __launch_bounds__(256) void level_finish(int ncell, int Lc, int C, int off, int Dp, const float* __restrict__ HP,
                                                    size_t hp_stride, int SP, int normalize, float* __restrict__ H,
                                                    float* __restrict__ nrm) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
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
    constexpr int WAVES = 8, T = WAVES * 64, PD = 4;
    constexpr bool KS = K16 > 0;
    constexpr int UNROLL_STEPS = KS ? 64 : 1;
    const int K = KS ? K16 * 16 : K_;
    const int S = F32 ? K : (KS ? (K16 + 1) / 2 * 32 + WS3_PAD : S_);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 15, g = lane >> 4;
    const int li = fetch_row_of(lane), lg = fetch_piece_of(lane), psrc = mfma_src_addr(lane);
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
        // MFMA-lane view: row i = target cell ti; its operand cells for the epilogue re-gather of xs = PL(a) + PR(b)
        const int gt = tile / lv.N, n = tile - gt * lv.N;
        const int ti = gt * 16 + i;
        const bool ok = ti < lv.ncell;
        const int tc = min(ti, lv.ncell - 1);
        const int eb = tc / lv.Lc, ep = tc - eb * lv.Lc;
        const size_t prow = (size_t)lv.rowbase + (size_t)tc * lv.N + n;
        const float pn = Pp[prow];
        const float* xa = PA + ((size_t)eb * lv.C + lv.pa[ep * lv.N + n]) * lda + col0 + 4 * g;
        const float* xb = PB + ((size_t)eb * lv.C + lv.pb[ep * lv.N + n]) * ldb + col0 + 4 * g;
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
            if (st % gy == by) {
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
            const float4 pa4 = ea[c], pb4 = ebv[c];
            const float xs0 = pa4.x + pb4.x, xs1 = pa4.y + pb4.y, xs2 = pa4.z + pb4.z, xs3 = pa4.w + pb4.w;
            const float4 x = make_float4(fmaxf(xs0, 0.f), fmaxf(xs1, 0.f), fmaxf(xs2, 0.f), fmaxf(xs3, 0.f));
            dp = fmaf(acc[c][0], x.x, dp); dp = fmaf(acc[c][1], x.y, dp); dp = fmaf(acc[c][2], x.z, dp); dp = fmaf(acc[c][3], x.w, dp);
            if (ok) {
                const size_t o = prow * Dp + col0 + c * 16 + 4 * g;
                st4(DA + o, make_float4(x.x > 0.f ? pn * acc[c][0] : 0.f, x.y > 0.f ? pn * acc[c][1] : 0.f,
                                        x.z > 0.f ? pn * acc[c][2] : 0.f, x.w > 0.f ? pn * acc[c][3] : 0.f));
                if (!TILED) st4(X + o, x);
            }
            if (TILED)
                store_split_tile(reinterpret_cast<uint32_t*>(X) + ((size_t)(lv.tilebase + tile) * NT + by * CT + c) * 256 + 8 * i + 2 * g,
                                 ok ? x : f4zero());
        }
        dp += __shfl_xor(dp, 16);
        dp += __shfl_xor(dp, 32);
        if (ok && g == 0) DPP[prow * gy + by] = dp;
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
