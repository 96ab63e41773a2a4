// DioraTreeLSTM kernels (gfx950).  PARITY UNPINNED: the reference ships this composition only
// as commented-out text (cliora/net/vg.py:28-76):
//   leaf     [u,i,o]       = chunk3(x W^T + B[:3D]);      c = sig(i) tanh(u);  h = sig(o) tanh(c)
//   compose  [u,i,o,f0,f1] = chunk5([a;b] U^T + B);
//            c = sig(f0+k) c_a + sig(f1+k) c_b + sig(i) tanh(u);  h = sig(o) tanh(c)      (k = 1 inside, 0 outside)
// Factored like the MLP: every CELL is projected once (PL = U[:, :D] h + B and PR = U[:, D:] h,
// five gate blocks each, plus QL = mat^T h), so a span PAIR needs no matmul at all -- only the
// gate arithmetic on PL(a) + PR(b) and the two child cell states.  The whole TreeLSTM pair path is
// therefore HBM-bound: 12 D floats read and (when a backward or a hook follows) 2 D written per pair in the forward.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "chart_kernels.hpp"

namespace cliora {

// Gate transcendentals on the hardware exp2 / rcp forms (round 4): sigmoid = rcp(1 + exp(-x)), tanh = 1 - 2 rcp(1 + exp(2x)) -- both
// saturate correctly through exp -> inf / 0, absolute error ~1e-7 (the cell kernels recompute every gate twice per pair in the
// backward: c5 at L 40 34.0 -> 31.4 ms, L 20 6.33 -> 5.90 against libm expf / tanhf; fixtures and oracle checks unchanged).
__device__ __forceinline__ float sigm(float x) { return __frcp_rn(1.f + __expf(-x)); }
__device__ __forceinline__ float lstm_tanh(float x) { return 1.f - 2.f * __frcp_rn(1.f + __expf(2.f * x)); }
__device__ __forceinline__ float4 f4map(float4 a, float (*f)(float)) { return make_float4(f(a.x), f(a.y), f(a.z), f(a.w)); }
__device__ __forceinline__ float4 f4sig(float4 a, float k) { return make_float4(sigm(a.x + k), sigm(a.y + k), sigm(a.z + k), sigm(a.w + k)); }
__device__ __forceinline__ float4 f4tanh(float4 a) { return make_float4(lstm_tanh(a.x), lstm_tanh(a.y), lstm_tanh(a.z), lstm_tanh(a.w)); }
__device__ __forceinline__ float4 f4mul(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 f4scale(float s, float4 a) { return make_float4(s * a.x, s * a.y, s * a.z, s * a.w); }
// g (1 - g)  and  (1 - t^2)
__device__ __forceinline__ float4 f4dsig(float4 g) { return make_float4(g.x * (1.f - g.x), g.y * (1.f - g.y), g.z * (1.f - g.z), g.w * (1.f - g.w)); }
__device__ __forceinline__ float4 f4dtanh(float4 t) { return make_float4(1.f - t.x * t.x, 1.f - t.y * t.y, 1.f - t.z * t.z, 1.f - t.w * t.w); }

// ---- leaves: ACT = x W^T + B[:3D] (3 blocks of Dp per row) -> h, c -> unit norm of both
static __global__ __launch_bounds__(256) void lstm_leaf_fwd(int B, int L, int C, int Dp, const float* __restrict__ ACT, int normalize,
                                                     float* __restrict__ H, float* __restrict__ Cc, float* __restrict__ nrmH,
                                                     float* __restrict__ nrmC, float* __restrict__ S) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= B * L) return;
    const int b = r / L, p = r - b * L;
    const size_t crow = (size_t)b * C + p;
    const float* a = ACT + (size_t)r * 3 * Dp;
    const int nv = Dp >> 2;
    float4 h[2], c[2];
    float sh = 0.f, sc = 0.f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int v = lane + 64 * k;
        h[k] = f4zero(); c[k] = f4zero();
        if (v < nv) {
            const float4 u = f4tanh(ld4(a + 4 * v)), i = f4sig(ld4(a + Dp + 4 * v), 0.f), o = f4sig(ld4(a + 2 * Dp + 4 * v), 0.f);
            c[k] = f4mul(i, u);
            h[k] = f4mul(o, f4tanh(c[k]));
            sh += f4dot(h[k], h[k]); sc += f4dot(c[k], c[k]);
        }
    }
    const float nh = sqrtf(wave_sum(sh)), nc = sqrtf(wave_sum(sc));
    const float dh = normalize ? fmaxf(nh, UNIT_EPS) : 1.f, dc = normalize ? fmaxf(nc, UNIT_EPS) : 1.f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int v = lane + 64 * k;
        if (v < nv) {
            st4(H + crow * Dp + 4 * v, make_float4(h[k].x / dh, h[k].y / dh, h[k].z / dh, h[k].w / dh));
            st4(Cc + crow * Dp + 4 * v, make_float4(c[k].x / dc, c[k].y / dc, c[k].z / dc, c[k].w / dc));
        }
    }
    if (lane == 0) { nrmH[crow] = nh; nrmC[crow] = nc; S[crow] = 0.f; }
}

// ---- backward of the cells of one level, cell-centric: one workgroup per cell, one THREAD per 16-byte column, walking the
// cell's use lists (the span pairs it is an operand of) and RECOMPUTING each pair's gate gradients on the fly:
//   pair (a, b) -> target t, split weight p:   [u,i,o,f0,f1] = gates(PL(a) + PR(b)),  c = f0 c_a + f1 c_b + i u,  tc = tanh(c)
//       dh = p dGh(t),   dc = p dGc(t) + dh o (1 - tc^2)
//       d act = [dc i (1-u^2), dc u i(1-i), dh tc o(1-o), dc c_a f0(1-f0), dc c_b f1(1-f1)]     (the same five rows for a and for b)
//       d c_a = dc f0,   d c_b = dc f1
//   inside cell:  dPL(5Dp) = sum_{left + sibling uses} d act;  dPR(5Dp) = sum_{right uses} d act;
//                 dQL = sum_{left} ds H(right) + sum_{sibling} ds OH(parent);
//                 vH = ext + sum_{right} ds QL(left);  vC = ext + sum_{left,sibling} d c_a + sum_{right} d c_b
//   outside cell: dPRo(5Dp) = sum_{parent uses} d act;  vH = ext + sum ds QL(sibling);  vC = ext + sum d c_b
// Round 2 wrote the five d act rows and d c_a / d c_b of every pair to HBM (lstm_pair_bwd: 24 kB read + 11 kB written per pair) and
// gathered them per cell (19 kB per pair); a use costs the same 12.8 kB here (the partner's five gate rows, its cell state, the
// target's two gradient rows) and nothing is written per pair: 57.6 -> 28.8 kB per pair in the backward, and the 16 GB pair-gradient
// buffer of B 64 / L 40 is gone.  The gate arithmetic is recomputed twice per pair (once from each operand's side): ~2 ms of
// VALU per step at that size against ~15 ms of HBM time saved.
// W floats per thread: W = 4 is 16-byte accesses by Dp/4 threads with ~170 registers each, W = 1 a thread per column with a quarter
// of the registers and four times the waves.  c5 at L = 40 times the same with 1, 2 and 4 (35.8 / 36.5 / 35.6 ms per step): the
// kernels are bound by the bytes of their use lists (nine rows per use) and the gate transcendentals, not by latency hiding.
template <int W> struct VecW { float v[W]; };
template <int W> __device__ __forceinline__ VecW<W> vzero() { VecW<W> r; for (int q = 0; q < W; ++q) r.v[q] = 0.f; return r; }
template <int W> __device__ __forceinline__ VecW<W> vload(const float* p) {
    VecW<W> r;
    if constexpr (W == 4) { const float4 t = ld4(p); r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w; }
    else if constexpr (W == 2) { const float2 t = *reinterpret_cast<const float2*>(p); r.v[0] = t.x; r.v[1] = t.y; }
    else r.v[0] = p[0];
    return r;
}
template <int W> __device__ __forceinline__ void vstore(float* p, const VecW<W>& a) {
    if constexpr (W == 4) st4(p, make_float4(a.v[0], a.v[1], a.v[2], a.v[3]));
    else if constexpr (W == 2) *reinterpret_cast<float2*>(p) = make_float2(a.v[0], a.v[1]);
    else p[0] = a.v[0];
}
template <int W> __device__ __forceinline__ VecW<W> vload_ext(const float* base, int D, int col) {      // caller's stride D, maybe unaligned
    VecW<W> r;
#pragma unroll
    for (int q = 0; q < W; ++q) r.v[q] = col + q < D ? base[col + q] : 0.f;
    return r;
}

template <int W> struct LstmPairGrad { VecW<W> da[5]; VecW<W> dca, dcb; };
template <int W>
__device__ __forceinline__ LstmPairGrad<W> lstm_pair_grad(const VecW<W> (&x)[5], const VecW<W>& cA, const VecW<W>& cB, float kf, float pn,
                                                          const VecW<W>& dGh, const VecW<W>& dGc) {
    LstmPairGrad<W> r;
#pragma unroll
    for (int q = 0; q < W; ++q) {
        const float u = lstm_tanh(x[0].v[q]), i = sigm(x[1].v[q] + 0.f), o = sigm(x[2].v[q] + 0.f), f0 = sigm(x[3].v[q] + kf), f1 = sigm(x[4].v[q] + kf);
        const float c = (f0 * cA.v[q] + f1 * cB.v[q]) + i * u;          // as lstm_cell_fwd forms it
        const float tc = lstm_tanh(c);
        const float dh = pn * dGh.v[q];
        const float dc = pn * dGc.v[q] + (dh * o) * (1.f - tc * tc);
        r.da[0].v[q] = (dc * i) * (1.f - u * u);
        r.da[1].v[q] = (dc * u) * (i * (1.f - i));
        r.da[2].v[q] = (dh * tc) * (o * (1.f - o));
        r.da[3].v[q] = (dc * cA.v[q]) * (f0 * (1.f - f0));
        r.da[4].v[q] = (dc * cB.v[q]) * (f1 * (1.f - f1));
        r.dca.v[q] = dc * f0;
        r.dcb.v[q] = dc * f1;
    }
    return r;
}

// One use list of the cell.  OWN_IS_A: the cell is the pair's a operand (left child / sibling) and the partner its b operand.
//   own[5]: the cell's own gate rows at this column; PP: the partner's gate rows (row stride ldp, first block at PP);
//   CP: the partner's cell-state chart; SC: the chart / projection the score term multiplies ds with (row stride lds)
template <bool OWN_IS_A, int W>
__device__ __forceinline__ void lstm_walk_uses(const UseTab& ut, int c, int b, int bC, int c4, int Dp, const VecW<W> (&own)[5], const VecW<W>& cown,
                                               const float* __restrict__ PP, int ldp, const float* __restrict__ CP, float kf,
                                               const int32_t* __restrict__ trow, const float* __restrict__ Pp, const float* __restrict__ DS,
                                               const float* __restrict__ dGh, const float* __restrict__ dGc, const float* __restrict__ SC, int lds,
                                               VecW<W> (&dact)[5], VecW<W>& vc, VecW<W>& sterm, float& vs) {
    const int beg = ut.off[c], end = ut.off[c + 1];
    constexpr int NB = W == 4 ? 2 : 4;                 // uses in flight
    for (int u0 = beg; u0 < end; u0 += NB) {
        VecW<W> pg[NB][5], pc[NB], gh[NB], gc[NB], sc[NB];
        float pn[NB], ds[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int uu = min(u0 + j, end - 1);
            const size_t r = (size_t)ut.row[uu] + (size_t)b * ut.stride[uu];
            const size_t prow = (size_t)(bC + ut.partner[uu]);
            const size_t tr = (size_t)trow[r];
            const bool live = u0 + j < end;
            pn[j] = live ? Pp[r] : 0.f;                // a dead slot re-reads the last use with weight zero: adds exact zeros
            ds[j] = live ? DS[r] : 0.f;
#pragma unroll
            for (int k = 0; k < 5; ++k) pg[j][k] = vload<W>(PP + prow * ldp + (size_t)k * Dp + c4);
            pc[j] = vload<W>(CP + prow * Dp + c4);
            gh[j] = vload<W>(dGh + tr * Dp + c4);
            gc[j] = vload<W>(dGc + tr * Dp + c4);
            sc[j] = vload<W>(SC + prow * lds + c4);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            VecW<W> x[5];
#pragma unroll
            for (int k = 0; k < 5; ++k)
#pragma unroll
                for (int q = 0; q < W; ++q) x[k].v[q] = OWN_IS_A ? own[k].v[q] + pg[j][k].v[q] : pg[j][k].v[q] + own[k].v[q];     // PL(a) + PR(b), in that order
            const LstmPairGrad<W> g = lstm_pair_grad<W>(x, OWN_IS_A ? cown : pc[j], OWN_IS_A ? pc[j] : cown, kf, pn[j], gh[j], gc[j]);
#pragma unroll
            for (int q = 0; q < W; ++q) {
#pragma unroll
                for (int k = 0; k < 5; ++k) dact[k].v[q] += g.da[k].v[q];
                vc.v[q] += OWN_IS_A ? g.dca.v[q] : g.dcb.v[q];
                sterm.v[q] = fmaf(ds[j], sc[j].v[q], sterm.v[q]);
            }
            vs += ds[j];
        }
    }
}

template <int W>
static __global__ __launch_bounds__(512) void lstm_cell_bwd_in(LevelArgs g, int D, const float* __restrict__ dH_ext, const float* __restrict__ dC_ext,
                                                        const float* __restrict__ dS_ext, UseTab ina, UseTab inb, UseTab outa, int with_outside,
                                                        const int32_t* __restrict__ trow, const float* __restrict__ Pp, const float* __restrict__ DS,
                                                        const float* __restrict__ PI, int ldpi, int blk_plo, int blk_qlo, const float* __restrict__ PO, int ldpo,
                                                        const float* __restrict__ IH, const float* __restrict__ IC, const float* __restrict__ OH,
                                                        const float* __restrict__ OC, const float* __restrict__ dGi, const float* __restrict__ dGci,
                                                        const float* __restrict__ dGo, const float* __restrict__ dGco, float* __restrict__ dPI,
                                                        float* __restrict__ VH, float* __restrict__ VC, float* __restrict__ dStot) {
    const int t = cell_of_block(blockIdx.x, g.B, g.Lc, g.affine), v = threadIdx.x;      // sentence-affine block order (chart_kernels.hpp)
    const int b = t / g.Lc, p = t - b * g.Lc;
    const int c = g.off + p;
    const size_t crow = (size_t)b * g.C + c;
    const int Dp = g.Dp, bC = b * g.C, nv = Dp / W;
    if (v >= nv) return;
    const int c4 = W * v;
    const float* mine = PI + crow * ldpi;
    float* o = dPI + crow * ldpi;
    const VecW<W> cown = vload<W>(IC + crow * Dp + c4);
    VecW<W> vh = dH_ext ? vload_ext<W>(dH_ext + crow * D, D, c4) : vzero<W>();
    VecW<W> vc = dC_ext ? vload_ext<W>(dC_ext + crow * D, D, c4) : vzero<W>();
    float vs = 0.f;
    {   // the cell as a operand: left child in the inside pass (its PL rows), then sibling in the outside pass (the same rows when the
        // outside functions are the inside ones, else its PLo rows of the outside weights); one role's rows and sums are live at a time
        VecW<W> pl[5], dpl[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) { pl[k] = vload<W>(mine + (size_t)k * Dp + c4); dpl[k] = vzero<W>(); }
        VecW<W> dql = vzero<W>();
        // left child: partner = right child (PR blocks), score term ds * H(right)
        lstm_walk_uses<true, W>(ina, c, b, bC, c4, Dp, pl, cown, PI + (size_t)5 * Dp, ldpi, IC, 1.0f, trow, Pp, DS, dGi, dGci, IH, Dp, dpl, vc, dql, vs);
        if (blk_plo != 0) {                 // unshared: the inside blocks are done, the sibling role has its own
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                vstore<W>(o + (size_t)k * Dp + c4, dpl[k]);
                pl[k] = vload<W>(mine + (size_t)(blk_plo + k) * Dp + c4); dpl[k] = vzero<W>();
            }
            vstore<W>(o + (size_t)10 * Dp + c4, dql);
            dql = vzero<W>();
        }
        // sibling in the outside pass: partner = parent (outside cell, PRo blocks), targets are outside cells, constant 0 (diora.py:174)
        if (with_outside)
            lstm_walk_uses<true, W>(outa, c, b, bC, c4, Dp, pl, cown, PO, ldpo, OC, 0.0f, trow, Pp, DS, dGo, dGco, OH, Dp, dpl, vc, dql, vs);
#pragma unroll
        for (int k = 0; k < 5; ++k) vstore<W>(o + (size_t)(blk_plo + k) * Dp + c4, dpl[k]);
        vstore<W>(o + (size_t)blk_qlo * Dp + c4, dql);
    }
    {   // right child: partner = left child (PL blocks), score term ds * QL(left)
        VecW<W> pr[5], dpr[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) { pr[k] = vload<W>(mine + (size_t)(5 + k) * Dp + c4); dpr[k] = vzero<W>(); }
        lstm_walk_uses<false, W>(inb, c, b, bC, c4, Dp, pr, cown, PI, ldpi, IC, 1.0f, trow, Pp, DS, dGi, dGci, PI + (size_t)10 * Dp, ldpi, dpr, vc, vh, vs);
#pragma unroll
        for (int k = 0; k < 5; ++k) vstore<W>(o + (size_t)(5 + k) * Dp + c4, dpr[k]);
    }
    vstore<W>(VH + crow * Dp + c4, vh);
    vstore<W>(VC + crow * Dp + c4, vc);
    if (v == 0) dStot[crow] = (dS_ext ? dS_ext[crow] : 0.f) + vs;
}

template <int W>
static __global__ __launch_bounds__(512) void lstm_cell_bwd_out(LevelArgs g, int D, const float* __restrict__ dH_ext, const float* __restrict__ dC_ext,
                                                         const float* __restrict__ dS_ext, UseTab outb, const int32_t* __restrict__ trow,
                                                         const float* __restrict__ Pp, const float* __restrict__ DS, const float* __restrict__ PI,
                                                         int ldpi, int blk_plo, int blk_qlo, const float* __restrict__ PO, int ldpo, const float* __restrict__ IC,
                                                         const float* __restrict__ OC, const float* __restrict__ dGo, const float* __restrict__ dGco,
                                                         float* __restrict__ dPO, float* __restrict__ VH, float* __restrict__ VC,
                                                         float* __restrict__ dStot) {
    const int t = cell_of_block(blockIdx.x, g.B, g.Lc, g.affine), v = threadIdx.x;      // sentence-affine block order (chart_kernels.hpp)
    const int b = t / g.Lc, p = t - b * g.Lc;
    const int c = g.off + p;
    const size_t crow = (size_t)b * g.C + c;
    const int Dp = g.Dp, bC = b * g.C, nv = Dp / W;
    if (v >= nv) return;
    const int c4 = W * v;
    VecW<W> po[5], dpo[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) { po[k] = vload<W>(PO + crow * ldpo + (size_t)k * Dp + c4); dpo[k] = vzero<W>(); }
    const VecW<W> cown = vload<W>(OC + crow * Dp + c4);
    VecW<W> vh = dH_ext ? vload_ext<W>(dH_ext + crow * D, D, c4) : vzero<W>();
    VecW<W> vc = dC_ext ? vload_ext<W>(dC_ext + crow * D, D, c4) : vzero<W>();
    float vs = 0.f;
    // parent in the outside pass: partner = sibling (inside cell; its PL / QL blocks of the outside functions), score term ds * QL(sibling)
    lstm_walk_uses<false, W>(outb, c, b, bC, c4, Dp, po, cown, PI + (size_t)blk_plo * Dp, ldpi, IC, 0.0f, trow, Pp, DS, dGo, dGco,
                             PI + (size_t)blk_qlo * Dp, ldpi, dpo, vc, vh, vs);
#pragma unroll
    for (int k = 0; k < 5; ++k) vstore<W>(dPO + crow * ldpo + (size_t)k * Dp + c4, dpo[k]);
    vstore<W>(VH + crow * Dp + c4, vh);
    vstore<W>(VC + crow * Dp + c4, vc);
    if (v == 0) dStot[crow] = (dS_ext ? dS_ext[crow] : 0.f) + vs;
}

// ---- forward of the cells of one level, cell-centric: one workgroup per target cell,
// one THREAD per W columns; the splits of the cell are walked NB at a time, each split's gates formed from PL(a) + PR(b) and the
// two child cell states, and h = sum_n p_n h_n, c = sum_n p_n c_n accumulate in registers in split order.  The per-split rows h_n, c_n go to Y / X only when the backward (its softmax term needs
// dG . h_n + dGc . c_n) or a hook will read them: the forward moves 12 rows per pair without them, 14 with (16 as the pair + aggregate
// kernels of round 2).  Measured on MI355X at B 64 / L 40: the training step does not move (36.3-37.0 ms against 35.6-36.8, one, two
// or four floats per thread alike) -- the backward's use lists set it; the evaluation forward is what gains.
//   PA / PB: the gate blocks of the a / b operand (row strides ldA / ldB), CA / CB their cell-state charts, kf the forget constant
template <int W>
static __global__ __launch_bounds__(512) void lstm_cell_fwd(LevelArgs g, const int32_t* __restrict__ arow, const int32_t* __restrict__ brow,
                                                     const float* __restrict__ PA, int ldA, const float* __restrict__ PB, int ldB,
                                                     const float* __restrict__ CA, const float* __restrict__ CB, float kf,
                                                     const float* __restrict__ Pp, int normalize, float* __restrict__ Y, float* __restrict__ X,
                                                     float* __restrict__ H, float* __restrict__ Cc, float* __restrict__ nrmH,
                                                     float* __restrict__ nrmC) {
    __shared__ float sh_n[2][8];
    const int t = cell_of_block(blockIdx.x, g.B, g.Lc, g.affine), v = threadIdx.x;      // sentence-affine block order (chart_kernels.hpp)
    const int b = t / g.Lc, p = t - b * g.Lc;
    const int Dp = g.Dp, nv = Dp / W;
    const bool act = v < nv;
    const int c4 = W * (act ? v : 0);
    const size_t row0 = (size_t)g.rowbase + (size_t)t * g.N;
    VecW<W> vh = vzero<W>(), vc = vzero<W>();
    constexpr int NB = W == 4 ? 2 : 4;
    for (int n0 = 0; n0 < g.N; n0 += NB) {
        VecW<W> ga[NB][5], gb[NB][5], ca[NB], cb[NB];
        float pn[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const size_t r = row0 + min(n0 + j, g.N - 1);
            const size_t ar = (size_t)arow[r], br = (size_t)brow[r];
            pn[j] = n0 + j < g.N ? Pp[r] : 0.f;
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                ga[j][k] = vload<W>(PA + ar * ldA + (size_t)k * Dp + c4);
                gb[j][k] = vload<W>(PB + br * ldB + (size_t)k * Dp + c4);
            }
            ca[j] = vload<W>(CA + ar * Dp + c4);
            cb[j] = vload<W>(CB + br * Dp + c4);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            if (n0 + j < g.N) {
                VecW<W> hh, cc;
#pragma unroll
                for (int q = 0; q < W; ++q) {
                    const float u = lstm_tanh(ga[j][0].v[q] + gb[j][0].v[q]);
                    const float i = sigm((ga[j][1].v[q] + gb[j][1].v[q]) + 0.f);
                    const float o = sigm((ga[j][2].v[q] + gb[j][2].v[q]) + 0.f);
                    const float f0 = sigm((ga[j][3].v[q] + gb[j][3].v[q]) + kf);
                    const float f1 = sigm((ga[j][4].v[q] + gb[j][4].v[q]) + kf);
                    const float c = (f0 * ca[j].v[q] + f1 * cb[j].v[q]) + i * u;
                    cc.v[q] = c;
                    hh.v[q] = o * lstm_tanh(c);
                    vh.v[q] = fmaf(pn[j], hh.v[q], vh.v[q]);
                    vc.v[q] = fmaf(pn[j], c, vc.v[q]);
                }
                if (Y && act) {
                    const size_t r = row0 + n0 + j;
                    vstore<W>(Y + r * Dp + c4, hh);
                    vstore<W>(X + r * Dp + c4, cc);
                }
            }
        }
    }
    // norms of both vectors over the workgroup: wave sums, then the waves in order
    float sh = 0.f, sc = 0.f;
    if (act) {
#pragma unroll
        for (int q = 0; q < W; ++q) { sh = fmaf(vh.v[q], vh.v[q], sh); sc = fmaf(vc.v[q], vc.v[q], sc); }
    }
    sh = wave_sum(sh); sc = wave_sum(sc);
    const int wave = v >> 6, nw = (blockDim.x + 63) >> 6;
    if ((v & 63) == 0) { sh_n[0][wave] = sh; sh_n[1][wave] = sc; }
    __syncthreads();
    float th = 0.f, tc = 0.f;
    for (int w = 0; w < nw; ++w) { th += sh_n[0][w]; tc += sh_n[1][w]; }
    const float nh = sqrtf(th), nc = sqrtf(tc);
    const float dh = normalize ? fmaxf(nh, UNIT_EPS) : 1.f, dc = normalize ? fmaxf(nc, UNIT_EPS) : 1.f;
    const size_t crow = (size_t)b * g.C + g.off + p;
    if (act) {
        VecW<W> oh, oc;
#pragma unroll
        for (int q = 0; q < W; ++q) { oh.v[q] = vh.v[q] / dh; oc.v[q] = vc.v[q] / dc; }
        vstore<W>(H + crow * Dp + c4, oh);
        vstore<W>(Cc + crow * Dp + c4, oc);
    }
    if (v == 0) { nrmH[crow] = nh; nrmC[crow] = nc; }
}

// ---- backward, unit-norm of both vectors + softmax/score backward (one workgroup per cell)
//   dp_n = dGh . y_n + dGc . x_n
static __global__ __launch_bounds__(256) void lstm_scores_bwd(LevelArgs g, const float* __restrict__ VH, const float* __restrict__ VC,
                                                       const float* __restrict__ H, const float* __restrict__ Cc,
                                                       const float* __restrict__ nrmH, const float* __restrict__ nrmC, int normalize,
                                                       const float* __restrict__ Y, const float* __restrict__ X,
                                                       const float* __restrict__ Sp, const float* __restrict__ Pp,
                                                       const float* __restrict__ Schart, const float* __restrict__ dStot,
                                                       float* __restrict__ dGh, float* __restrict__ dGc, float* __restrict__ DS) {
    __shared__ float sh_dp[64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t = cell_of_block(blockIdx.x, g.B, g.Lc, g.affine);      // sentence-affine block order (chart_kernels.hpp)
    const int b = t / g.Lc, p = t - b * g.Lc;
    const size_t crow = (size_t)b * g.C + g.off + p;
    const int Dp = g.Dp, nv = Dp >> 2;
    const bool a0 = lane < nv, a1 = lane + 64 < nv;
    float4 v0 = f4zero(), v1 = f4zero(), h0 = f4zero(), h1 = f4zero(), w0 = f4zero(), w1 = f4zero(), c0 = f4zero(), c1 = f4zero();
    if (a0) { v0 = ld4(VH + crow * Dp + 4 * lane); h0 = ld4(H + crow * Dp + 4 * lane); w0 = ld4(VC + crow * Dp + 4 * lane); c0 = ld4(Cc + crow * Dp + 4 * lane); }
    if (a1) { v1 = ld4(VH + crow * Dp + 4 * (lane + 64)); h1 = ld4(H + crow * Dp + 4 * (lane + 64)); w1 = ld4(VC + crow * Dp + 4 * (lane + 64)); c1 = ld4(Cc + crow * Dp + 4 * (lane + 64)); }
    unit_norm_bwd(v0, v1, h0, h1, nrmH[crow], normalize);
    unit_norm_bwd(w0, w1, c0, c1, nrmC[crow], normalize);
    if (wave == 0) {
        if (a0) { st4(dGh + crow * Dp + 4 * lane, v0); st4(dGc + crow * Dp + 4 * lane, w0); }
        if (a1) { st4(dGh + crow * Dp + 4 * (lane + 64), v1); st4(dGc + crow * Dp + 4 * (lane + 64), w1); }
    }
    if (g.N == 0) return;
    const int row0 = g.rowbase + t * g.N;
    for (int n0 = wave; n0 < g.N; n0 += 8) {
        float d[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const size_t r = (size_t)(row0 + min(n0 + 4 * j, g.N - 1));
            float s = 0.f;
            if (a0) s = f4dot(v0, ld4(Y + r * Dp + 4 * lane)) + f4dot(w0, ld4(X + r * Dp + 4 * lane));
            if (a1) s += f4dot(v1, ld4(Y + r * Dp + 4 * (lane + 64))) + f4dot(w1, ld4(X + r * Dp + 4 * (lane + 64)));
            d[j] = s;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float r = wave_sum(d[j]);
            if (lane == 0 && n0 + 4 * j < g.N) sh_dp[n0 + 4 * j] = r;
        }
    }
    __syncthreads();
    if (wave != 0) return;
    const bool an = lane < g.N;
    const float dp = an ? sh_dp[lane] : 0.f;
    const float pn = an ? Pp[row0 + lane] : 0.f;
    const float sn = an ? Sp[row0 + lane] : 0.f;
    const float mean = wave_sum(pn * dp);
    // 1 + (s_n - S), in that order: the scores reach 1e8 without unit normalisation, where (1 + s_n) - S loses the 1 (found by
    // tools/fuzz_parity.py: every gradient through outside_s of a cell with |S| > 2^24 vanished)
    const float ds = pn * ((dp - mean) + dStot[crow] * (1.f + (sn - Schart[crow])));
    if (an) DS[row0 + lane] = ds;
}

// ---- leaves backward: unit-norm of h and c, then the leaf gates -> dACT (3 blocks)
static __global__ __launch_bounds__(256) void lstm_leaf_bwd(int B, int L, int C, int Dp, const float* __restrict__ VH, const float* __restrict__ VC,
                                                     const float* __restrict__ H, const float* __restrict__ Cc,
                                                     const float* __restrict__ nrmH, const float* __restrict__ nrmC, int normalize,
                                                     const float* __restrict__ ACT, float* __restrict__ dACT) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= B * L) return;
    const int b = r / L, p = r - b * L;
    const size_t crow = (size_t)b * C + p;
    const int nv = Dp >> 2;
    const bool a0 = lane < nv, a1 = lane + 64 < nv;
    float4 v0 = f4zero(), v1 = f4zero(), h0 = f4zero(), h1 = f4zero(), w0 = f4zero(), w1 = f4zero(), c0 = f4zero(), c1 = f4zero();
    if (a0) { v0 = ld4(VH + crow * Dp + 4 * lane); h0 = ld4(H + crow * Dp + 4 * lane); w0 = ld4(VC + crow * Dp + 4 * lane); c0 = ld4(Cc + crow * Dp + 4 * lane); }
    if (a1) { v1 = ld4(VH + crow * Dp + 4 * (lane + 64)); h1 = ld4(H + crow * Dp + 4 * (lane + 64)); w1 = ld4(VC + crow * Dp + 4 * (lane + 64)); c1 = ld4(Cc + crow * Dp + 4 * (lane + 64)); }
    unit_norm_bwd(v0, v1, h0, h1, nrmH[crow], normalize);     // -> dh
    unit_norm_bwd(w0, w1, c0, c1, nrmC[crow], normalize);     // -> dc (direct part)
    const float* a = ACT + (size_t)r * 3 * Dp;
    float* d = dACT + (size_t)r * 3 * Dp;
    const float4 dh[2] = {v0, v1}, dcd[2] = {w0, w1};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int v = lane + 64 * k;
        if (v < nv) {
            const float4 u = f4tanh(ld4(a + 4 * v)), i = f4sig(ld4(a + Dp + 4 * v), 0.f), o = f4sig(ld4(a + 2 * Dp + 4 * v), 0.f);
            const float4 tc = f4tanh(f4mul(i, u));
            const float4 dc = f4add(dcd[k], f4mul(f4mul(dh[k], o), f4dtanh(tc)));
            st4(d + 4 * v, f4mul(f4mul(dc, i), f4dtanh(u)));
            st4(d + Dp + 4 * v, f4mul(f4mul(dc, u), f4dsig(i)));
            st4(d + 2 * Dp + 4 * v, f4mul(f4mul(dh[k], tc), f4dsig(o)));
        }
    }
}

}  // namespace cliora
