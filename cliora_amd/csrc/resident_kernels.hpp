// Sentence-resident chart kernels for small hidden sizes (Dp <= 64; BASELINE configs[0]: d 50, batch 8, length 10) (gfx950).
//
// At these sizes a chart level is a few hundred pair rows for the whole batch: the launch-per-level path (level_kernels.hpp) is
// bound by what separates two levels -- a launch boundary (or, in round 3's one-launch level loop, a grid-wide barrier:
// profiles/r03_persist_ab.txt) -- not by the work.  The sentences of a batch never exchange anything inside the recursion (diora.py:295-331, 358-398
// index one sentence's chart only), so here ONE WORKGROUP owns ONE SENTENCE and walks every level of both passes by itself:
// the only thing between two levels is a workgroup barrier.  Sixteen waves; a wave owns a chart cell of the level:
//
//   forward   split scores s_n = QL(a).h_b + s_a + s_b and their softmax (lane n holds split n), the compose MLP of every split
//             y_n = relu(W2 relu(PL(a) + PR(b)) + b2), the aggregate g = sum_n p_n y_n, its unit norm, the projections of the new cell
//   backward  the gather of every use of the cell (cell_gather_bwd_*), the projections' transposed product, the unit-norm backward,
//             and per split the compose backward + the softmax / score backward (level_compose_bwd, cell_dsoftmax)
//
// A row of Dp <= 64 floats is ONE VALUE PER LANE; a Dp x Dp layer is 64 steps of (one LDS read of the weight row, conflict-free,
// + one v_readlane broadcast of the operand + FMA), four rows at a time so that the weight read is shared; weights stay in LDS for
// the whole kernel (shared: 80 KB, unshared: 128 KB at Dp = 64).  Exact fp32 FMA arithmetic in either cliora_set_mfma_mode.
//
// Every buffer is read and written in the launch path's own format (charts, projections, per-split score / weight, ReLU bits,
// DA / DS / DZ / X pair rows, dPI / dPO), so either direction can run on either path (tests/test_gpu_resident.py mixes them) and
// the weight-gradient GEMMs of the backward's tail are the launch path's.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "level_kernels.hpp"

namespace cliora {

constexpr int RES_WAVES = 12;
constexpr int RES_THREADS = RES_WAVES * 64;
constexpr int RES_R = 8;                     // pair rows per pass over a weight matrix

struct ResArgs {
    const int32_t* tabs;                     // the plan's int32 tables
    uint32_t pa_in, pb_in, pa_out, pb_out;   // offsets (ints) of the per-sentence pair tables
    uint32_t lvl_in, lvl_out;                // ... of the per-level pair counts before a level (Plan::lvl_base_*)
    UseTab use_ina, use_inb, use_outa, use_outb;
    int B, L, C, D, Dp, ldpi, nblk, blk_plo, blk_qlo, share, normalize, run_outside;
    int ct, gy;                              // geometry of the ReLU-bit words (FwdLayout::ct3, ncb3)
    long long R_in;
    // forward state
    float *IH, *OH, *IS, *OS, *PI, *PO, *nrmi, *nrmo, *Sp, *Pp;
    uint32_t* ymask;                         // nullptr: no backward will follow
    const float *wcatT, *bcat, *w2iT, *b2i, *w2oT, *b2o, *w1roT, *rootp;      // forward: transposed weights ([k][col])
    const float *wcat, *w2i, *w2o, *w1ro;                                      // backward: the weights as they are ([col][k])
    // backward
    const float *dIH, *dIS, *dOH, *dOS;      // cotangents of the four outputs (row stride D)
    float* T;                                // tanh output of the leaf layer (B*L x Dp): written by the forward, read by the backward
    const float *X, *wlT, *bl, *wl;          // leaf layer (diora.py:58-63): input rows (B*L x Dp), Wl^T ([k][col]), bias, Wl ([col][k])
    float* dX;                               // d x_span (row stride D) or nullptr
    float *outIH, *outOH;                    // forward, D != Dp only: the caller's (B, C, D) charts, written beside the padded working copies
    float *VHo, *dPI, *dPO, *DA, *DS, *DZ, *Xrows, *dU;
    unsigned long long* trace;               // diagnostics (CLIORA_RES_TRACE=1): wall-clock stamps of workgroup 0, wave 0
};

#define RES_STAMPW(i) do { if (a.trace && blockIdx.x == 0 && lane == 0) a.trace[(i)] = wall_clock64(); } while (0)
#define RES_STAMP(i) do { if (a.trace && blockIdx.x == 0 && threadIdx.x == 0) a.trace[(i)] = wall_clock64(); } while (0)

__device__ __forceinline__ float res_bcast(float v, int k) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), k)); }
__device__ __forceinline__ int res_bcast_i(int v, int k) { return __builtin_amdgcn_readlane(v, k); }

// Wave reductions on the DPP network instead of ds_bpermute (wave_sum of gemm_kernels.hpp: six dependent LDS-crossbar round trips,
// ~0.25 us each here, and a cell's routine is a chain of N + 5 of them): the butterfly's first four steps as quad_perm / row_half_mirror /
// row_mirror moves inside a 16-lane row, the four row sums through v_readlane.  Same pairs in the same order as wave_sum: bitwise equal.
template <int CTRL>
__device__ __forceinline__ float res_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float res_wave_sum(float v) {
    v += res_dpp<0xB1>(v);          // quad_perm [1,0,3,2]: lane ^ 1
    v += res_dpp<0x4E>(v);          // quad_perm [2,3,0,1]: lane ^ 2
    v += res_dpp<0x141>(v);         // row_half_mirror: the other quad of the 8 (quads are uniform by now)
    v += res_dpp<0x140>(v);         // row_mirror: the other half of the row
    const float r0 = res_bcast(v, 0), r1 = res_bcast(v, 16), r2 = res_bcast(v, 32), r3 = res_bcast(v, 48);
    return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ float res_wave_max(float v) {
    v = fmaxf(v, res_dpp<0xB1>(v));
    v = fmaxf(v, res_dpp<0x4E>(v));
    v = fmaxf(v, res_dpp<0x141>(v));
    v = fmaxf(v, res_dpp<0x140>(v));
    const float r0 = res_bcast(v, 0), r1 = res_bcast(v, 16), r2 = res_bcast(v, 32), r3 = res_bcast(v, 48);
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}

// A Dp x Dp layer for rows held ONE VALUE PER LANE:  out[lane] += sum_k Wt[k][lane] * x[k].
// The weight row Wt[k][.] is one conflict-free LDS read; the operand x[k] must reach every lane: the wave parks its rows in its LDS
// scratch and every lane reads the SAME address back (a broadcast read, 16 bytes = four values at a time), so the arithmetic is
// v_pk_fma_f32 on register pairs with no cross-lane VALU work.  hipcc left to itself emits ds_read -> s_waitcnt lgkmcnt(0) -> FMA
// per k (one LDS latency per k: 2.6 us for one 64 x 64 layer, measured with the stamps below); the loads are therefore issued in
// blocks of eight k, double-buffered, with scheduling barriers that keep each block where it is written.
// One wave's LDS instructions execute in order: the broadcast reads see the parked values.
typedef float res_f2 __attribute__((ext_vector_type(2)));
constexpr int RES_SCR = 576;                 // floats of LDS scratch per wave: eight parked rows [k][8] (or four [k][4] + one more row [k] at 256)

__device__ __forceinline__ void res_park_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void res_park4(float* xs, int lane, bool act, const float (&x)[4]) {
    if (act) *reinterpret_cast<float4*>(xs + 4 * lane) = make_float4(x[0], x[1], x[2], x[3]);
    res_park_sync();
}
__device__ __forceinline__ void res_park1(float* xs, int lane, bool act, float x) {
    if (act) xs[lane] = x;
    res_park_sync();
}

// eight rows at once: out[r][lane] += sum_k W[k][lane] * xs[k][r].  The weight sits in LDS interleaved by four k
// (Wq[((k >> 2) * Dp + lane) * 4 + (k & 3)]): one 16-byte read gives a lane its weights of four k, eight broadcast reads the operands.
__device__ __forceinline__ void res_park8(float* xs, int lane, bool act, const float (&x)[RES_R]) {
    if (act) {
        *reinterpret_cast<float4*>(xs + 8 * lane) = make_float4(x[0], x[1], x[2], x[3]);
        *reinterpret_cast<float4*>(xs + 8 * lane + 4) = make_float4(x[4], x[5], x[6], x[7]);
    }
    res_park_sync();
}
__device__ __forceinline__ void res_matvec8(const float* Wq, int Dp, int lc, const float* xs, float (&out)[RES_R]) {
    res_f2 o0 = {out[0], out[1]}, o1 = {out[2], out[3]}, o2 = {out[4], out[5]}, o3 = {out[6], out[7]};
    // phases of two k (four broadcast reads, eight packed FMAs), double-buffered; the lane's weights of four k arrive as one read
    float4 w, wn, xA[4], xB[4];
    auto load = [&](float4 (&x)[4], int k0) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            x[2 * kk] = *reinterpret_cast<const float4*>(xs + 8 * (k0 + kk));
            x[2 * kk + 1] = *reinterpret_cast<const float4*>(xs + 8 * (k0 + kk) + 4);
        }
    };
    auto mac = [&](float w0, float w1, const float4 (&x)[4]) {
        const float wv[2] = {w0, w1};
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const res_f2 ww = {wv[kk], wv[kk]};
            o0 = __builtin_elementwise_fma(ww, res_f2{x[2 * kk].x, x[2 * kk].y}, o0);
            o1 = __builtin_elementwise_fma(ww, res_f2{x[2 * kk].z, x[2 * kk].w}, o1);
            o2 = __builtin_elementwise_fma(ww, res_f2{x[2 * kk + 1].x, x[2 * kk + 1].y}, o2);
            o3 = __builtin_elementwise_fma(ww, res_f2{x[2 * kk + 1].z, x[2 * kk + 1].w}, o3);
        }
    };
    w = *reinterpret_cast<const float4*>(Wq + (size_t)lc * 4);
    load(xA, 0);
    for (int k0 = 0; k0 < Dp; k0 += 4) {
        load(xB, k0 + 2);
        wn = *reinterpret_cast<const float4*>(Wq + ((size_t)(min(k0 + 4, Dp - 4) >> 2) * Dp + lc) * 4);
        __builtin_amdgcn_sched_barrier(0);
        mac(w.x, w.y, xA);
        __builtin_amdgcn_sched_barrier(0);
        if (k0 + 4 < Dp) load(xA, k0 + 4);
        __builtin_amdgcn_sched_barrier(0);
        mac(w.z, w.w, xB);
        __builtin_amdgcn_sched_barrier(0);
        w = wn;
    }
    out[0] = o0.x; out[1] = o0.y; out[2] = o1.x; out[3] = o1.y; out[4] = o2.x; out[5] = o2.y; out[6] = o3.x; out[7] = o3.y;
    __builtin_amdgcn_wave_barrier();
}

// one row: returns out + sum_k Wt[k][lane] * xs[k]
__device__ __forceinline__ float res_matvec1(const float* Wt, int ld, int Dp, int lc, const float* xs, float out) {
    float o2 = 0.f;
    float wA[8], wB[8];
    float4 xA[2], xB[2];
    auto load = [&](float (&w)[8], float4 (&x)[2], int k0) {
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) w[kk] = Wt[(k0 + kk) * ld + lc];
        x[0] = *reinterpret_cast<const float4*>(xs + k0);
        x[1] = *reinterpret_cast<const float4*>(xs + k0 + 4);
    };
    auto mac = [&](const float (&w)[8], const float4 (&x)[2]) {
        out = fmaf(w[0], x[0].x, out); o2 = fmaf(w[1], x[0].y, o2); out = fmaf(w[2], x[0].z, out); o2 = fmaf(w[3], x[0].w, o2);
        out = fmaf(w[4], x[1].x, out); o2 = fmaf(w[5], x[1].y, o2); out = fmaf(w[6], x[1].z, out); o2 = fmaf(w[7], x[1].w, o2);
    };
    load(wA, xA, 0);
    for (int k0 = 0; k0 < Dp; k0 += 16) {
        load(wB, xB, k0 + 8);
        __builtin_amdgcn_sched_barrier(0);
        mac(wA, xA);
        __builtin_amdgcn_sched_barrier(0);
        if (k0 + 16 < Dp) load(wA, xA, k0 + 16);
        __builtin_amdgcn_sched_barrier(0);
        mac(wB, xB);
        __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_wave_barrier();
    return out + o2;
}

// The cell projections.  LDS holds the blocks INTERLEAVED: W4[(k * Dp + lane) * 4 + blk] for the blocks 0..3 (one 16-byte read per
// k gives a lane its four block weights) and W5[k * Dp + lane] for the fifth block of an unshared plan.
//   forward   o[blk][lane] = bias + sum_k Wcat[blk*Dp + lane][k] h[k]          W4 indexed (k, lane): h parked as one row
//   backward  v[lane]     += sum_col sum_blk dP[blk][col] Wcat[blk*Dp + col][lane]   W4 indexed (col, lane): dP parked [col][4] (+ [col])
template <bool FIVE>
__device__ __forceinline__ void res_project_fwd(const float* W4, const float* W5, int Dp, int lc, const float* xs, float (&o)[5]) {
    res_f2 o01 = {o[0], o[1]}, o23 = {o[2], o[3]};
    float o4 = o[4];
    float4 wA[4], wB[4], xA, xB;
    float vA[4], vB[4];
    auto load = [&](float4 (&w)[4], float (&v)[4], float4& x, int k0) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            w[kk] = *reinterpret_cast<const float4*>(W4 + ((size_t)(k0 + kk) * Dp + lc) * 4);
            v[kk] = FIVE ? W5[(k0 + kk) * Dp + lc] : 0.f;
        }
        x = *reinterpret_cast<const float4*>(xs + k0);
    };
    auto mac = [&](const float4 (&w)[4], const float (&v)[4], const float4& x) {
        const float xv[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const res_f2 xx = {xv[kk], xv[kk]};
            o01 = __builtin_elementwise_fma(res_f2{w[kk].x, w[kk].y}, xx, o01);
            o23 = __builtin_elementwise_fma(res_f2{w[kk].z, w[kk].w}, xx, o23);
            if (FIVE) o4 = fmaf(v[kk], xv[kk], o4);
        }
    };
    load(wA, vA, xA, 0);
    for (int k0 = 0; k0 < Dp; k0 += 8) {
        load(wB, vB, xB, k0 + 4);
        __builtin_amdgcn_sched_barrier(0);
        mac(wA, vA, xA);
        __builtin_amdgcn_sched_barrier(0);
        if (k0 + 8 < Dp) load(wA, vA, xA, k0 + 8);
        __builtin_amdgcn_sched_barrier(0);
        mac(wB, vB, xB);
        __builtin_amdgcn_sched_barrier(0);
    }
    o[0] = o01.x; o[1] = o01.y; o[2] = o23.x; o[3] = o23.y; o[4] = o4;
    __builtin_amdgcn_wave_barrier();
}
template <bool FIVE>
__device__ __forceinline__ float res_project_bwd(const float* W4, const float* W5, int Dp, int lc, const float* xs4, const float* xs5) {
    res_f2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
    float a4 = 0.f;
    float4 wA[4], wB[4], xA[4], xB[4], yA, yB;
    float vA[4], vB[4];
    auto load = [&](float4 (&w)[4], float (&v)[4], float4 (&x)[4], float4& y, int c0) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            w[kk] = *reinterpret_cast<const float4*>(W4 + ((size_t)(c0 + kk) * Dp + lc) * 4);
            v[kk] = FIVE ? W5[(c0 + kk) * Dp + lc] : 0.f;
            x[kk] = *reinterpret_cast<const float4*>(xs4 + 4 * (c0 + kk));
        }
        y = FIVE ? *reinterpret_cast<const float4*>(xs5 + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto mac = [&](const float4 (&w)[4], const float (&v)[4], const float4 (&x)[4], const float4& y) {
        const float yv[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            a01 = __builtin_elementwise_fma(res_f2{w[kk].x, w[kk].y}, res_f2{x[kk].x, x[kk].y}, a01);
            a23 = __builtin_elementwise_fma(res_f2{w[kk].z, w[kk].w}, res_f2{x[kk].z, x[kk].w}, a23);
            if (FIVE) a4 = fmaf(v[kk], yv[kk], a4);
        }
    };
    load(wA, vA, xA, yA, 0);
    for (int c0 = 0; c0 < Dp; c0 += 8) {
        load(wB, vB, xB, yB, c0 + 4);
        __builtin_amdgcn_sched_barrier(0);
        mac(wA, vA, xA, yA);
        __builtin_amdgcn_sched_barrier(0);
        if (c0 + 8 < Dp) load(wA, vA, xA, yA, c0 + 8);
        __builtin_amdgcn_sched_barrier(0);
        mac(wB, vB, xB, yB);
        __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_wave_barrier();
    return ((a01.x + a01.y) + (a23.x + a23.y)) + a4;
}

// One pass's view of a level for the per-cell routines below
struct ResPass {
    const int32_t *pa, *pb;                  // the level's operand cells: index p*N + n  (a: left child / sibling, b: right child / parent)
    const float* QA; int ldq;                // QL block of the a operands
    const float* XA; int ldxa;               // first-layer block of the a operands (PL, bias included)
    const float* XB; int ldxb;               // first-layer block of the b operands (PR / PRo)
    const float* HB;                         // chart of the b operands (row stride Dp)
    const float *SA, *SB;                    // chart scores of the a / b operands
    const float* W2;                         // LDS: forward W2^T ([k][col]); backward W2 ([z][x])
    float b2;                                // this lane's element of the second bias
    float *H, *nrm, *S;                      // the target chart, its norms and scores
    float* Hout;                             // forward: the caller's un-padded chart (or nullptr)
    int Lc, N, off; long long rowbase;
};

// ---------------------------------------------------------------------------------------------------------------- forward
// One target cell (wave-wide): scores -> softmax -> compose of every split -> aggregate -> unit norm.  Returns h (one value per lane).
__device__ __forceinline__ float res_cell_fwd(const ResArgs& a, const ResPass& q, int b, int p, int lane, bool act, int lc, float* xs, int tr = 0) {
    const int Dp = a.Dp, N = q.N;
    const size_t bC = (size_t)b * a.C;
    const size_t crow = bC + q.off + p;
    const size_t row0 = (size_t)q.rowbase + ((size_t)b * q.Lc + p) * N;
    const int nl = min(lane, N - 1);
    const int ca = q.pa[p * N + nl], cb = q.pb[p * N + nl];
    // the first chunk's compose operands leave with the score operands: one round trip for both
    float xa[RES_R], xb[RES_R];
    auto load_x = [&](int n0, float (&ua)[RES_R], float (&ub)[RES_R]) {
#pragma unroll
        for (int r = 0; r < RES_R; ++r) {
            const int n = min(n0 + r, N - 1);
            const int can = res_bcast_i(ca, n), cbn = res_bcast_i(cb, n);
            ua[r] = act ? q.XA[(bC + can) * q.ldxa + lane] : 0.f;
            ub[r] = act ? q.XB[(bC + cbn) * q.ldxb + lane] : 0.f;
        }
    };
    RES_STAMP(tr + 0);
    load_x(0, xa, xb);
    const float sa = q.SA[bC + ca], sb = q.SB[bC + cb];
    float my_s = -INFINITY;
    for (int n0 = 0; n0 < N; n0 += RES_R) {                // eight splits' operand rows in flight
        float d[RES_R];
#pragma unroll
        for (int r = 0; r < RES_R; ++r) {
            const int n = min(n0 + r, N - 1);
            const int can = res_bcast_i(ca, n), cbn = res_bcast_i(cb, n);
            d[r] = act ? q.QA[(bC + can) * q.ldq + lane] * q.HB[(bC + cbn) * Dp + lane] : 0.f;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < RES_R; ++r) {
            if (n0 + r >= N) break;                            // wave-uniform
            const float s = res_wave_sum(d[r]);
            if (lane == n0 + r) my_s = (s + sa) + sb;
        }
    }
    RES_STAMP(tr + 1);
    const float m = res_wave_max(my_s);
    const float e = lane < N ? expf(my_s - m) : 0.f;
    const float pn = e / res_wave_sum(e);
    if (lane < N) { a.Sp[row0 + lane] = my_s; a.Pp[row0 + lane] = pn; }
    const float st = res_wave_sum(lane < N ? pn * my_s : 0.f);
    if (lane == 0) q.S[crow] = st;
    RES_STAMP(tr + 2);

    float hagg = 0.f;
    for (int n0 = 0; n0 < N; n0 += RES_R) {
        float x[RES_R], z[RES_R];
#pragma unroll
        for (int r = 0; r < RES_R; ++r) {
            x[r] = fmaxf(xa[r] + xb[r], 0.f);
            z[r] = q.b2;
        }
        res_park8(xs, lane, act, x);
        if (n0 + RES_R < N) load_x(n0 + RES_R, xa, xb);      // the next chunk's rows travel under this chunk's layer
        __builtin_amdgcn_sched_barrier(0);
        res_matvec8(q.W2, Dp, lc, xs, z);
#pragma unroll
        for (int r = 0; r < RES_R; ++r) {
            const int n = n0 + r;
            if (n >= N) break;                                 // wave-uniform
            const float y = act ? fmaxf(z[r], 0.f) : 0.f;
            hagg = fmaf(res_bcast(pn, n), y, hagg);
            if (a.ymask) {
                // the launch path's words: column k -> block k / (ct*16), word (k % 16) / 4 of the block, bit 4 * ((k % (ct*16)) / 16) + k % 4
                const unsigned long long mk = __ballot(y > 0.f);
                const int nw = a.gy * 4;
                if (lane < nw) {
                    const int by = lane >> 2, g = lane & 3;
                    uint32_t word = 0;
                    for (int c = 0; c < a.ct; ++c) word |= (uint32_t)((mk >> (by * a.ct * 16 + c * 16 + 4 * g)) & 0xFull) << (4 * c);
                    a.ymask[(row0 + n) * nw + lane] = word;
                }
            }
        }
    }
    RES_STAMP(tr + 3);
    const float nr = sqrtf(res_wave_sum(hagg * hagg));
    const float den = a.normalize ? fmaxf(nr, UNIT_EPS) : 1.f;
    const float h = hagg / den;
    if (act) q.H[crow * Dp + lane] = h;
    if (q.Hout && lane < a.D) q.Hout[crow * a.D + lane] = h;
    if (lane == 0) q.nrm[crow] = nr;
    RES_STAMP(tr + 4);
    return h;
}

static __global__ __launch_bounds__(RES_THREADS) void resident_fwd(ResArgs a) {
    extern __shared__ __attribute__((aligned(16))) float res_lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int Dp = a.Dp, ldpi = a.ldpi, L = a.L, C = a.C;
    const bool act = lane < Dp;
    const int lc = min(lane, Dp - 1);
    float* sW4 = res_lds;                                     // [k][col][4]: projection blocks 0..3, interleaved
    float* sW2i = sW4 + 4 * Dp * Dp;                          // [k][col]
    float* sW1ro = sW2i + Dp * Dp;
    float* sW2o = a.share ? sW2i : sW1ro + Dp * Dp;
    float* sW5 = sW1ro + 2 * Dp * Dp;                         // [k][col]: the fifth block (unshared plans only)
    float* xs = sW1ro + (a.share ? 1 : 3) * Dp * Dp + wave * RES_SCR;            // this wave's operand scratch
    // the leaf layer's weight: beside the scratch when there is room (shared plans: 139 KB in all), from global memory otherwise
    const float* sWl = a.wlT;
    if (a.share) {
        float* dst = sW1ro + Dp * Dp + RES_WAVES * RES_SCR;
        for (int i = threadIdx.x; i < Dp * Dp; i += RES_THREADS) dst[i] = a.wlT[i];
        sWl = dst;
    }
    const float bl = act ? a.bl[lane] : 0.f;
    for (int i = threadIdx.x; i < 4 * Dp * Dp; i += RES_THREADS) {
        const int blk = i & 3, kc = i >> 2, k = kc / Dp, col = kc - k * Dp;
        sW4[i] = blk < a.nblk ? a.wcatT[(size_t)k * ldpi + blk * Dp + col] : 0.f;
    }
    for (int i = threadIdx.x; i < Dp * Dp; i += RES_THREADS) {
        const int qi = (((i / Dp) >> 2) * Dp + i % Dp) * 4 + ((i / Dp) & 3);      // interleaved by four k (res_matvec8)
        sW2i[qi] = a.w2iT[i];
        sW1ro[i] = a.w1roT[i];
        if (!a.share) {
            sW2o[qi] = a.w2oT[i];
            const int k = i / Dp, col = i - k * Dp;
            sW5[i] = a.wcatT[(size_t)k * ldpi + 4 * Dp + col];
        }
    }
    const float b2i = act ? a.b2i[lane] : 0.f, b2o = act ? a.b2o[lane] : 0.f;
    float bc[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) bc[k] = (act && k < a.nblk) ? a.bcat[k * Dp + lane] : 0.f;
    __syncthreads();
    const int32_t* lvl_in = a.tabs + a.lvl_in;
    const int32_t* lvl_out = a.tabs + a.lvl_out;

    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        const size_t bC = (size_t)b * C;
        auto inside_cell = [&](int level, int p) {           // diora.py:295-331 for one cell; the leaves' rows and projections are there already
            // the lane index made opaque per cell: hipcc otherwise hoists one 64-bit per-lane address per array out of the whole level loop
            // (~50 VGPRs of kernel-invariant pointers) and spills them at the 168-VGPR cap of twelve waves
            int lane_o = lane; asm volatile("" : "+v"(lane_o));
            const int lane = lane_o; const bool act = lane < Dp; const int lc = min(lane, Dp - 1);
            ResPass q;
            q.Lc = L - level; q.N = level; q.off = C - (L - level) * (L - level + 1) / 2;
            q.rowbase = (long long)a.B * lvl_in[level];
            q.pa = a.tabs + a.pa_in + lvl_in[level]; q.pb = a.tabs + a.pb_in + lvl_in[level];
            q.QA = a.PI + 2 * Dp; q.ldq = ldpi; q.XA = a.PI; q.ldxa = ldpi; q.XB = a.PI + Dp; q.ldxb = ldpi;
            q.HB = a.IH; q.SA = a.IS; q.SB = a.IS; q.W2 = sW2i; q.b2 = b2i; q.H = a.IH; q.nrm = a.nrmi; q.S = a.IS; q.Hout = a.outIH;
            const float h = res_cell_fwd(a, q, b, p, lane, act, lc, xs, 256 + 8 * level);
            RES_STAMP(4 * level + 1);
            if (level < L - 1) {                              // [PL | PR | QL | (PLo | QLo)] = h Wcat^T + bias
                float o[5];
#pragma unroll
                for (int k = 0; k < 5; ++k) o[k] = bc[k];
                res_park1(xs, lane, act, h);
                if (a.nblk > 4) res_project_fwd<true>(sW4, sW5, Dp, lc, xs, o);
                else res_project_fwd<false>(sW4, sW5, Dp, lc, xs, o);
                float* dst = a.PI + (bC + q.off + p) * ldpi + lane;
#pragma unroll
                for (int k = 0; k < 5; ++k)
                    if (act && k < a.nblk) dst[k * Dp] = o[k];
            }
            RES_STAMP(4 * level + 2);
        };
        auto outside_cell = [&](int level, int p) {          // diora.py:358-398 for one cell
            int lane_o = lane; asm volatile("" : "+v"(lane_o));
            const int lane = lane_o; const bool act = lane < Dp; const int lc = min(lane, Dp - 1);
            ResPass q;
            q.Lc = L - level; q.N = L - 1 - level; q.off = C - (L - level) * (L - level + 1) / 2;
            q.rowbase = a.R_in + (long long)a.B * lvl_out[level];
            q.pa = a.tabs + a.pa_out + lvl_out[level]; q.pb = a.tabs + a.pb_out + lvl_out[level];
            q.QA = a.PI + (size_t)a.blk_qlo * Dp; q.ldq = ldpi; q.XA = a.PI + (size_t)a.blk_plo * Dp; q.ldxa = ldpi; q.XB = a.PO; q.ldxb = Dp;
            q.HB = a.OH; q.SA = a.IS; q.SB = a.OS; q.W2 = sW2o; q.b2 = b2o; q.H = a.OH; q.nrm = a.nrmo; q.S = a.OS; q.Hout = a.outOH;
            const float h = res_cell_fwd(a, q, b, p, lane, act, lc, xs);
            if (level >= 1) {                                 // PRo of the new parents
                res_park1(xs, lane, act, h);
                const float o = res_matvec1(sW1ro, Dp, Dp, lc, xs, 0.f);
                if (act) a.PO[(bC + q.off + p) * Dp + lane] = o;
            }
        };
        // leaves (diora.py:58-63, 283-292): h = unit(tanh(x Wl^T + bl)), score 0, and their projections
        for (int p = wave; p < L; p += RES_WAVES) {
            const size_t r = (size_t)b * L + p, crow = bC + p;
            res_park1(xs, lane, act, act ? a.X[r * Dp + lane] : 0.f);
            const float u = res_matvec1(sWl, Dp, Dp, lc, xs, bl);
            const float t = act ? tanhf(u) : 0.f;
            if (act) a.T[r * Dp + lane] = t;
            const float nr = sqrtf(res_wave_sum(t * t));
            const float den = a.normalize ? fmaxf(nr, UNIT_EPS) : 1.f;
            const float h = t / den;
            if (act) a.IH[crow * Dp + lane] = h;
            if (a.outIH && lane < a.D) a.outIH[crow * a.D + lane] = h;
            if (lane == 0) { a.nrmi[crow] = nr; a.IS[crow] = 0.f; }
            if (L > 1) {
                float o[5];
#pragma unroll
                for (int k = 0; k < 5; ++k) o[k] = bc[k];
                res_park1(xs, lane, act, h);
                if (a.nblk > 4) res_project_fwd<true>(sW4, sW5, Dp, lc, xs, o);
                else res_project_fwd<false>(sW4, sW5, Dp, lc, xs, o);
                float* dst = a.PI + crow * ldpi + lane;
#pragma unroll
                for (int k = 0; k < 5; ++k)
                    if (act && k < a.nblk) dst[k * Dp] = o[k];
            }
        }
        // root of the outside chart (diora.py:337-356): the last wave, beside the leaves
        if (a.run_outside && wave == RES_WAVES - 1) {
            const float v = act ? a.rootp[lane] : 0.f;
            const float nr = sqrtf(res_wave_sum(v * v));
            const float den = a.normalize ? fmaxf(nr, UNIT_EPS) : 1.f;
            const float h = v / den;
            const size_t crow = bC + C - 1;
            if (act) a.OH[crow * Dp + lane] = h;
            if (a.outOH && lane < a.D) a.outOH[crow * a.D + lane] = h;
            if (lane == 0) { a.nrmo[crow] = nr; a.OS[crow] = 0.f; }
            res_park1(xs, lane, act, h);
            const float o = res_matvec1(sW1ro, Dp, Dp, lc, xs, 0.f);
            if (act) a.PO[crow * Dp + lane] = o;
        }
        __syncthreads();
        // The two passes as a wavefront (cliora_chart_forward): outside level t composes parents of the outside
        // levels above it with siblings of the inside levels <= L-2-t, so step k runs inside level k AND outside level L-k: L steps
        // instead of 2 (L-1), and the cells of a step are (L-k) + k = L whatever k -- every step fills the same number of waves.
        for (int k = 1; k <= L; ++k) {
            const int nin = k <= L - 1 ? L - k : 0;
            const int nout = (a.run_outside && k >= 2) ? k : 0;
            RES_STAMP(4 * k + 0);
            for (int t = wave; t < nin + nout; t += RES_WAVES) {
                if (t < nin) inside_cell(k, t);
                else outside_cell(L - k, t - nin);
            }
            __syncthreads();
            RES_STAMP(4 * k + 3);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------- backward
// one use list of a cell: da += DA[row], x += ds[row] * SRC[partner], vS += ds[row]     (gather_uses, one wave, one value per lane)
__device__ __forceinline__ void res_gather(const UseTab& ut, int c, int b, size_t bC, const float* DA, const float* DS, int Dp, const float* SRC,
                                           int ldsrc, int lane, bool act, float& da, float& x, float& vS) {
    const int beg = ut.off[c], end = ut.off[c + 1];
    for (int u0 = beg; u0 < end; u0 += 64) {
        const int uu = min(u0 + lane, end - 1);
        const size_t rl = (size_t)ut.row[uu] + (size_t)b * ut.stride[uu];
        const int pl = ut.partner[uu];
        const float dsl = (u0 + lane < end) ? DS[rl] : 0.f;
        const int cnt = min(64, end - u0);
        for (int j0 = 0; j0 < cnt; j0 += 4) {
            float av[4], sv[4], dv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int jj = min(j0 + j, cnt - 1);
                const unsigned rlo = __builtin_amdgcn_readlane((unsigned)(rl & 0xffffffffu), jj);
                const unsigned rhi = __builtin_amdgcn_readlane((unsigned)(rl >> 32), jj);
                const size_t r = ((size_t)rhi << 32) | rlo;
                const int pc = res_bcast_i(pl, jj);
                dv[j] = (j0 + j < cnt) ? res_bcast(dsl, jj) : 0.f;
                av[j] = (act && j0 + j < cnt) ? DA[r * Dp + lane] : 0.f;
                sv[j] = act ? SRC[(bC + pc) * ldsrc + lane] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                vS += dv[j];
                da += av[j];
                x = fmaf(dv[j], sv[j], x);
            }
        }
    }
}

__device__ __forceinline__ float res_dnorm(float v, float h, float nr, int normalize) {
    if (!normalize) return v;
    if (nr > UNIT_EPS) {
        const float dot = res_wave_sum(v * h);
        const float inv = 1.f / nr;
        return (v - h * dot) * inv;
    }
    return v * (1.f / UNIT_EPS);
}

// the splits of one target cell: compose backward + softmax / score backward (level_compose_bwd + cell_dsoftmax); dG: gradient at the aggregate
__device__ __forceinline__ void res_cell_pairs_bwd(const ResArgs& a, const ResPass& q, int b, int p, int lane, bool act, int lc, float dG,
                                                   float dStot, float* scr) {
    const int Dp = a.Dp, N = q.N;
    const size_t bC = (size_t)b * a.C;
    const size_t crow = bC + q.off + p;
    const size_t row0 = (size_t)q.rowbase + ((size_t)b * q.Lc + p) * N;
    const int nl = min(lane, N - 1);
    const int ca = q.pa[p * N + nl], cb = q.pb[p * N + nl];
    const float pn = lane < N ? a.Pp[row0 + lane] : 0.f;
    const float sn = lane < N ? a.Sp[row0 + lane] : 0.f;
    // this lane's ReLU bit position (column = lane)
    const int cw = a.ct * 16;
    const int by = lc / cw, cc = (lc % cw) >> 4, g = (lc & 15) >> 2, e = lc & 3;
    const int nw = a.gy * 4;
    float dp_l = 0.f;
    for (int n0 = 0; n0 < N; n0 += RES_R) {
        float dzu[RES_R], u[RES_R], xs[RES_R];
#pragma unroll
        for (int r = 0; r < RES_R; ++r) {
            const int n = min(n0 + r, N - 1);
            const uint32_t word = a.ymask[(row0 + n) * nw + by * 4 + g];
            const int can = res_bcast_i(ca, n), cbn = res_bcast_i(cb, n);
            xs[r] = act ? q.XA[(bC + can) * q.ldxa + lane] + q.XB[(bC + cbn) * q.ldxb + lane] : 0.f;
            dzu[r] = (act && ((word >> (4 * cc + e)) & 1u)) ? dG : 0.f;
            u[r] = 0.f;
        }
        res_park8(scr, lane, act, dzu);
        res_matvec8(q.W2, Dp, lc, scr, u);
#pragma unroll
        for (int r = 0; r < RES_R; ++r) {
            const int n = n0 + r;
            if (n >= N) break;                                 // wave-uniform
            const float pr = res_bcast(pn, n);
            const float x = fmaxf(xs[r], 0.f);
            const float uu = act ? u[r] : 0.f;
            // dG . y_n = u . x_n + (dG masked) . b2   (y = relu(z))
            const float dp = res_wave_sum(fmaf(uu, x, dzu[r] * q.b2));
            if (lane == n) dp_l = dp;
            if (act) {
                const size_t o = (row0 + n) * Dp + lane;
                a.DZ[o] = pr * dzu[r];
                a.Xrows[o] = x;
                a.DA[o] = xs[r] > 0.f ? pr * uu : 0.f;
            }
        }
    }
    const float mean = res_wave_sum(pn * dp_l);
    const float ds = pn * ((dp_l - mean) + dStot * (1.f + (sn - q.S[crow])));
    if (lane < N) a.DS[row0 + lane] = ds;
}

static __global__ __launch_bounds__(RES_THREADS) void resident_bwd(ResArgs a) {
    extern __shared__ __attribute__((aligned(16))) float res_lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int Dp = a.Dp, ldpi = a.ldpi, L = a.L, C = a.C, D = a.D;
    const bool act = lane < Dp;
    float* sW4 = res_lds;                                     // [col][k][4]: projection blocks 0..3, interleaved
    float* sW2i = sW4 + 4 * Dp * Dp;                          // [z][x]
    float* sW1ro = sW2i + Dp * Dp;                            // [col][k]
    float* sW2o = a.share ? sW2i : sW1ro + Dp * Dp;
    float* sW5 = sW1ro + 2 * Dp * Dp;                         // [col][k]: the fifth block (unshared plans only)
    float* scr = sW1ro + (a.share ? 1 : 3) * Dp * Dp + wave * RES_SCR;           // this wave's operand scratch
    const float* sWl = a.wl;                                  // leaf weight [col][k]: in LDS when there is room (shared plans)
    if (a.share) {
        float* dst = sW1ro + Dp * Dp + RES_WAVES * RES_SCR;
        for (int i = threadIdx.x; i < Dp * Dp; i += RES_THREADS) dst[i] = a.wl[i];
        sWl = dst;
    }
    for (int i = threadIdx.x; i < 4 * Dp * Dp; i += RES_THREADS) {
        const int blk = i & 3, ck = i >> 2, col = ck / Dp, k = ck - col * Dp;
        sW4[i] = blk < a.nblk ? a.wcat[((size_t)blk * Dp + col) * Dp + k] : 0.f;
    }
    for (int i = threadIdx.x; i < Dp * Dp; i += RES_THREADS) {
        const int qi = (((i / Dp) >> 2) * Dp + i % Dp) * 4 + ((i / Dp) & 3);      // interleaved by four z (res_matvec8)
        sW2i[qi] = a.w2i[i];
        sW1ro[i] = a.w1ro[i];
        if (!a.share) {
            sW2o[qi] = a.w2o[i];
            sW5[i] = a.wcat[(size_t)4 * Dp * Dp + i];
        }
    }
    const float b2i = act ? a.b2i[lane] : 0.f, b2o = act ? a.b2o[lane] : 0.f;
    __syncthreads();
    const int32_t* lvl_in = a.tabs + a.lvl_in;
    const int32_t* lvl_out = a.tabs + a.lvl_out;

    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        const size_t bC = (size_t)b * C;
        auto outside_cell = [&](int level, int p) {          // a parent's uses are the pairs of the outside levels below it
            int lane_o = lane; asm volatile("" : "+v"(lane_o));
            const int lane = lane_o; const bool act = lane < Dp; const int lc = min(lane, Dp - 1);
            const bool ext = lane < D;
            ResPass q;
            q.Lc = L - level; q.N = L - 1 - level; q.off = C - (L - level) * (L - level + 1) / 2;
            q.rowbase = a.R_in + (long long)a.B * lvl_out[level];          // (the root level has no pairs: N = 0)
            q.pa = a.tabs + a.pa_out + lvl_out[level]; q.pb = a.tabs + a.pb_out + lvl_out[level];
            q.XA = a.PI + (size_t)a.blk_plo * Dp; q.ldxa = ldpi; q.XB = a.PO; q.ldxb = Dp;
            q.W2 = sW2o; q.b2 = b2o; q.S = a.OS;
            const int c = q.off + p;
            const size_t crow = bC + c;
            const int tb = 512 + 16 * level;
            if (p == 0) RES_STAMPW(tb + 0);
            float dpo = 0.f, v = 0.f, vS = 0.f;
            res_gather(a.use_outb, c, b, bC, a.DA, a.DS, Dp, a.PI + (size_t)a.blk_qlo * Dp, ldpi, lane, act, dpo, v, vS);
            v += (ext && a.dOH) ? a.dOH[crow * D + lane] : 0.f;
            if (level < L - 1 && a.dOS) vS += a.dOS[crow];
            if (p == 0) RES_STAMPW(tb + 1);
            if (act) a.dPO[crow * Dp + lane] = dpo;
            if (level >= 1) {                                  // vH += dPRo . W1R_out
                res_park1(scr, lane, act, dpo);
                const float o = res_matvec1(sW1ro, Dp, Dp, lc, scr, 0.f);
                v += act ? o : 0.f;
            }
            if (level == L - 1) {                              // the root: root_bwd sums its unit-norm backward over the batch
                if (act) a.VHo[crow * Dp + lane] = v;
                return;
            }
            if (p == 0) RES_STAMPW(tb + 2);
            const float h = act ? a.OH[crow * Dp + lane] : 0.f;
            const float dG = res_dnorm(v, h, a.nrmo[crow], a.normalize);
            if (p == 0) RES_STAMPW(tb + 3);
            res_cell_pairs_bwd(a, q, b, p, lane, act, lc, dG, vS, scr);
            if (p == 0) RES_STAMPW(tb + 4);
        };
        auto inside_cell = [&](int level, int p) {
            int lane_o = lane; asm volatile("" : "+v"(lane_o));
            const int lane = lane_o; const bool act = lane < Dp; const int lc = min(lane, Dp - 1);
            const bool ext = lane < D;
            ResPass q;
            q.Lc = L - level; q.N = level; q.off = C - (L - level) * (L - level + 1) / 2;
            q.rowbase = (long long)a.B * lvl_in[level];
            q.pa = a.tabs + a.pa_in + lvl_in[level]; q.pb = a.tabs + a.pb_in + lvl_in[level];
            q.XA = a.PI; q.ldxa = ldpi; q.XB = a.PI + Dp; q.ldxb = ldpi;
            q.W2 = sW2i; q.b2 = b2i; q.S = a.IS;
            const int c = q.off + p;
            const size_t crow = bC + c;
            const int tb = 512 + 16 * (L - 1 - level) + 8;
            if (p == 0) RES_STAMPW(tb + 0);
            float dPL = 0.f, dPR = 0.f, dQL = 0.f, dPLo = 0.f, dQLo = 0.f, v = 0.f, vS = 0.f;
            res_gather(a.use_inb, c, b, bC, a.DA, a.DS, Dp, a.PI + 2 * Dp, ldpi, lane, act, dPR, v, vS);      // right-child uses
            res_gather(a.use_ina, c, b, bC, a.DA, a.DS, Dp, a.IH, Dp, lane, act, dPL, dQL, vS);                // left-child uses
            if (a.run_outside) res_gather(a.use_outa, c, b, bC, a.DA, a.DS, Dp, a.OH, Dp, lane, act, dPLo, dQLo, vS);   // sibling uses
            if (p == 0) RES_STAMPW(tb + 1);
            if (a.share) { dPL += dPLo; dQL += dQLo; }
            float blk[5] = {dPL, dPR, dQL, dPLo, dQLo};
            float* o = a.dPI + crow * ldpi + lane;
#pragma unroll
            for (int k = 0; k < 5; ++k)
                if (act && k < a.nblk) o[k * Dp] = blk[k];
            v += (ext && a.dIH) ? a.dIH[crow * D + lane] : 0.f;
            if (level >= 1 && a.dIS) vS += a.dIS[crow];
            if (level <= L - 2) {                              // vH += dP . Wcat
                const float b4[4] = {blk[0], blk[1], blk[2], a.nblk > 3 ? blk[3] : 0.f};
                if (act) scr[256 + lane] = a.nblk > 4 ? blk[4] : 0.f;
                res_park4(scr, lane, act, b4);
                const float acc = a.nblk > 4 ? res_project_bwd<true>(sW4, sW5, Dp, lc, scr, scr + 256) : res_project_bwd<false>(sW4, sW5, Dp, lc, scr, scr + 256);
                v += act ? acc : 0.f;
            }
            if (p == 0) RES_STAMPW(tb + 2);
            const float h = act ? a.IH[crow * Dp + lane] : 0.f;
            const float dG = res_dnorm(v, h, a.nrmi[crow], a.normalize);
            if (p == 0) RES_STAMPW(tb + 3);
            if (level == 0) {                                  // leaves: H = unit(T), T = tanh(U)  (leaf_bwd_pre)
                const size_t r = (size_t)b * L + p;
                float du = 0.f;
                if (act) { const float t = a.T[r * Dp + lane]; du = dG * (1.f - t * t); a.dU[r * Dp + lane] = du; }
                if (a.dX) {                                    // d x = dU Wl
                    res_park1(scr, lane, act, du);
                    const float dx = res_matvec1(sWl, Dp, Dp, lc, scr, 0.f);
                    if (ext) a.dX[r * D + lane] = dx;
                }
                return;
            }
            res_cell_pairs_bwd(a, q, b, p, lane, act, lc, dG, vS, scr);
            if (p == 0) RES_STAMPW(tb + 4);
        };
        // The two chains as a wavefront (cliora_chart_backward): the backward of outside level t only feeds inside cells of the levels
        // <= L-2-t (its siblings), so step j runs outside level j AND inside level L-1-j -- whose gathers read the outside pair rows
        // of the levels <= j-1, written in earlier steps: L steps of (L-j) + (j+1) = L+1 cells each.
        for (int j = 0; j <= L - 1; ++j) {
            const int nout = a.run_outside ? L - j : 0;
            const int nin = j + 1;
            for (int t = wave; t < nout + nin; t += RES_WAVES) {
                if (t < nout) outside_cell(j, t);
                else inside_cell(L - 1 - j, t - nout);
            }
            __syncthreads();
            RES_STAMP(512 + 16 * j + 7);
        }
    }
}

}  // namespace cliora
