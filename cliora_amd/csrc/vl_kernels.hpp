// CLIORA (vision-language) additions to the chart engine (gfx950).
//
//   AttentionHead.forward            cliora/net/cliora.py:28-42
//       score_k = q . o_k  (k over the R image regions of the SAME sentence -- the reference
//       computes all B x B sentence/image pairs and keeps the diagonal, :37-39),
//       prob = softmax_k(score), dropout(0.1) in training, context = sum_k prob_k o_k
//   VLComposeMLP.leaf_transform      cliora.py:71-80, 290-301   h = unit(unit(tanh(fc x)) + ctx), c = unit(ctx)
//   inside_aggregate                 cliora.py:140-157          h = unit(unit(sum_n p_n y_n) + ctx)
//   span-region / word-region scorers  cliora.py:453-468       einsum('abx,cdx->acbd', ...)
//
// One workgroup (4 waves) per chart cell.  A wave's share of the R region vectors of the sentence (regions w, w+4, ...: NRW of
// them, 9 at R = 36) is fetched into registers at the START of the kernel -- the loads do not depend on the cell's own row -- and
// serves both the scores and the context.  The kernels are latency chains (22 us per launch for 64 cells as for 1 280 when the
// regions were fetched four at a time, twice: round 3 trace), not bandwidth: one round trip instead of seven.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "chart_kernels.hpp"

namespace cliora {

constexpr int VL_MAXR = 64;     // regions per image (one per lane in the softmax)

// ---------------------------------------------------------------------------------
// Forward for the inside cells of one level (N > 0: sum the partial aggregates first; N == 0: leaves,
// source = tanh output T).  Writes H = unit(u + ctx), U = u, P = prob (before dropout), the two
// norms, and for the leaves inside_c = unit(ctx).
// ---------------------------------------------------------------------------------
// The aggregate g = sum_n p_n y_n of the level's cells comes from level_compose_fwd as SP partial rows (HP).
template <int NRW>
static __global__ __launch_bounds__(256) void cell_attend_fwd(LevelArgs g, int L, const float* __restrict__ HP, size_t hp_stride, int SP,
                                                       const float* __restrict__ T, const float* __restrict__ OBJ, int R,
                                                       const float* __restrict__ mask, int normalize,
                                                       float* __restrict__ H, float* __restrict__ nrmV, float* __restrict__ U,
                                                       float* __restrict__ nrmU, float* __restrict__ PK, float* __restrict__ IC,
                                                       int icD, float* __restrict__ S) {
    // one workgroup (4 waves) per cell: the splits of the aggregate and the regions of the attention are dealt over the
    // waves (4 rows in flight each), partial vectors meet in LDS and are added in wave order (fixed summation order)
    __shared__ float4 sh_v[4][128];
    __shared__ float sh_sc[VL_MAXR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t = cell_of_block(blockIdx.x, g.B, g.Lc, g.affine);      // sentence-affine block order (chart_kernels.hpp)
    const int b = t / g.Lc, p = t - b * g.Lc;
    const size_t crow = (size_t)b * g.C + g.off + p;
    const int Dp = g.Dp, nv = Dp >> 2;
    const bool a0 = lane < nv, a1 = lane + 64 < nv;
    const int c0 = 4 * lane, c1 = 4 * (lane + 64);
    auto sum_waves = [&](float4& x0, float4& x1) {          // every wave leaves with the sum of the four partial vectors
        __syncthreads();                                    // (the previous round's readers are done)
        if (a0) sh_v[wave][lane] = x0;
        if (a1) sh_v[wave][lane + 64] = x1;
        __syncthreads();
        if (a0) x0 = f4add(f4add(f4add(sh_v[0][lane], sh_v[1][lane]), sh_v[2][lane]), sh_v[3][lane]);
        if (a1) x1 = f4add(f4add(f4add(sh_v[0][lane + 64], sh_v[1][lane + 64]), sh_v[2][lane + 64]), sh_v[3][lane + 64]);
    };
    // this wave's regions and this lane's dropout factor, on their way before anything else
    const float mk_pre = (mask && lane < R) ? mask[crow * R + lane] : 1.f;
    const float* ob = OBJ + (size_t)b * R * Dp;
    float4 o0[NRW], o1[NRW];
#pragma unroll
    for (int m = 0; m < NRW; ++m) {
        const float* o = ob + (size_t)min(wave + 4 * m, R - 1) * Dp;
        o0[m] = a0 ? ld4(o + c0) : f4zero();
        o1[m] = a1 ? ld4(o + c1) : f4zero();
    }
    float4 v0 = f4zero(), v1 = f4zero();
    if (g.N == 0) {
        const float* s = T + ((size_t)b * L + p) * Dp;
        if (a0) v0 = ld4(s + c0);
        if (a1) v1 = ld4(s + c1);
        if (tid == 0) S[crow] = 0.f;
    } else {                                          // every wave reads the whole (pre-aggregated) row: parts added in order
        for (int sp = 0; sp < SP; ++sp) {
            const float* src = HP + (size_t)sp * hp_stride + crow * Dp;
            if (a0) v0 = f4add(v0, ld4(src + c0));
            if (a1) v1 = f4add(v1, ld4(src + c1));
        }
    }
    // u = unit(v)   (every wave holds the whole row)
    const float nu = sqrtf(wave_sum(f4dot(v0, v0) + f4dot(v1, v1)));
    const float du = normalize ? fmaxf(nu, UNIT_EPS) : 1.f;
    const float4 u0 = make_float4(v0.x / du, v0.y / du, v0.z / du, v0.w / du);
    const float4 u1 = make_float4(v1.x / du, v1.y / du, v1.z / du, v1.w / du);
    // scores over the regions of this sentence: wave w takes regions w, w+4, ...
    {
        float d[NRW];
#pragma unroll
        for (int m = 0; m < NRW; ++m) d[m] = f4dot(u0, o0[m]) + f4dot(u1, o1[m]);
#pragma unroll
        for (int m = 0; m < NRW; ++m) {
            const float s = wave_sum(d[m]);
            if (lane == 0 && wave + 4 * m < R) sh_sc[wave + 4 * m] = s;
        }
    }
    __syncthreads();
    const float my_sc = lane < R ? sh_sc[lane] : -INFINITY;
    const float mx = wave_max(my_sc);
    const float e = lane < R ? expf(my_sc - mx) : 0.f;
    const float pk = e / wave_sum(e);
    const float pm = (mask && lane < R) ? pk * mk_pre : pk;   // pre-scaled dropout mask (0 or 1/(1-p))
    if (wave == 0 && lane < R) PK[crow * VL_MAXR + lane] = pk;
    // context = sum_k pm_k o_k, the wave's regions first (from the registers the scores used)
    float4 x0 = f4zero(), x1 = f4zero();
#pragma unroll
    for (int m = 0; m < NRW; ++m) {
        const int k = wave + 4 * m;
        const float pj = __shfl(pm, min(k, R - 1));
        const float w = k < R ? pj : 0.f;
        x0 = f4fma(w, o0[m], x0); x1 = f4fma(w, o1[m], x1);
    }
    sum_waves(x0, x1);
    if (wave != 0) return;
    // h = unit(u + ctx)
    const float4 w0 = f4add(u0, x0), w1 = f4add(u1, x1);
    const float nw = sqrtf(wave_sum(f4dot(w0, w0) + f4dot(w1, w1)));
    const float dw = normalize ? fmaxf(nw, UNIT_EPS) : 1.f;
    if (a0) { st4(H + crow * Dp + c0, make_float4(w0.x / dw, w0.y / dw, w0.z / dw, w0.w / dw)); st4(U + crow * Dp + c0, u0); }
    if (a1) { st4(H + crow * Dp + c1, make_float4(w1.x / dw, w1.y / dw, w1.z / dw, w1.w / dw)); st4(U + crow * Dp + c1, u1); }
    if (lane == 0) { nrmU[crow] = nu; nrmV[crow] = nw; }
    if (IC && g.N == 0) {               // leaves: c = unit(ctx)  (cliora.py:79, 299)
        const float nc = sqrtf(wave_sum(f4dot(x0, x0) + f4dot(x1, x1)));
        const float dc = normalize ? fmaxf(nc, UNIT_EPS) : 1.f;
        float* ic = IC + crow * icD;
        const float xs[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (c0 + j < icD) ic[c0 + j] = xs[j] / dc;
            if (a1 && c1 + j < icD) ic[c1 + j] = xs[4 + j] / dc;
        }
    }
}

// ---------------------------------------------------------------------------------
// Backward of the attention residual for the cells of one level.  In: VH = dL/dH (complete).
//   dv = normbwd(VH; H, |v|);  dctx = dv;  dpm_k = dctx . o_k;  dp = dpm * mask
//   dsc_k = p_k (dp_k - sum_j p_j dp_j);  du = dv + sum_k dsc_k o_k
// Out: VH := du (the unit-norm backward through u = unit(g) is done by the kernels that follow,
// which are handed U / |g| instead of H / |v|), DCTX = dv, and per region PMo = p*mask, DSC = dsc
// for the per-sentence reduction of d obj (obj_grad_reduce).
// ---------------------------------------------------------------------------------
template <int NRW>
static __global__ __launch_bounds__(256) void cell_attend_bwd(LevelArgs g, float* __restrict__ VH, const float* __restrict__ H,
                                                       const float* __restrict__ nrmV, int normalize, const float* __restrict__ OBJ,
                                                       int R, const float* __restrict__ mask, const float* __restrict__ PK,
                                                       float* __restrict__ DCTX, float* __restrict__ PMo, float* __restrict__ DSC,
                                                       const float* __restrict__ U, const float* __restrict__ nrmU, float* __restrict__ dG) {
    // dG != nullptr (round 4): the unit-norm backward through u = unit(g) that cell_dnorm did in a launch of its own behind this one --
    // same loads, same formulas (unit_norm_bwd on the du this kernel just formed), one launch less on the inside chain of every level
    // one workgroup (4 waves) per cell, the regions dealt over the waves as in cell_attend_fwd
    __shared__ float4 sh_v[4][128];
    __shared__ float sh_d[VL_MAXR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t = cell_of_block(blockIdx.x, g.B, g.Lc, g.affine);      // sentence-affine block order (chart_kernels.hpp)
    const int b = t / g.Lc, p = t - b * g.Lc;
    const size_t crow = (size_t)b * g.C + g.off + p;
    const int Dp = g.Dp, nv = Dp >> 2;
    const bool a0 = lane < nv, a1 = lane + 64 < nv;
    const int c0 = 4 * lane, c1 = 4 * (lane + 64);
    const float* ob = OBJ + (size_t)b * R * Dp;
    const bool ak = lane < R;
    const float pk = ak ? PK[crow * VL_MAXR + lane] : 0.f;          // the softmax and the dropout factor of this lane's region: no later round trip
    const float mk = (mask && ak) ? mask[crow * R + lane] : 1.f;
    float4 o0[NRW], o1[NRW];                              // this wave's regions (w, w+4, ...), fetched once for both uses
#pragma unroll
    for (int m = 0; m < NRW; ++m) {
        const float* o = ob + (size_t)min(wave + 4 * m, R - 1) * Dp;
        o0[m] = a0 ? ld4(o + c0) : f4zero();
        o1[m] = a1 ? ld4(o + c1) : f4zero();
    }
    float4 v0 = f4zero(), v1 = f4zero(), h0 = f4zero(), h1 = f4zero();
    if (a0) { v0 = ld4(VH + crow * Dp + c0); h0 = ld4(H + crow * Dp + c0); }
    if (a1) { v1 = ld4(VH + crow * Dp + c1); h1 = ld4(H + crow * Dp + c1); }
    unit_norm_bwd(v0, v1, h0, h1, nrmV[crow], normalize);        // v = dL/d(u + ctx), the same in every wave
    {
        float d[NRW];
#pragma unroll
        for (int m = 0; m < NRW; ++m) d[m] = f4dot(v0, o0[m]) + f4dot(v1, o1[m]);
#pragma unroll
        for (int m = 0; m < NRW; ++m) {
            const float s = wave_sum(d[m]);
            if (lane == 0 && wave + 4 * m < R) sh_d[wave + 4 * m] = s;
        }
    }
    __syncthreads();
    const float dpm = ak ? sh_d[lane] : 0.f;
    const float dp = ak ? dpm * mk : 0.f;
    const float mean = wave_sum(pk * dp);
    const float dsc = pk * (dp - mean);
    if (wave == 0 && ak) { PMo[crow * VL_MAXR + lane] = pk * mk; DSC[crow * VL_MAXR + lane] = dsc; }
    float4 u0 = wave == 0 ? v0 : f4zero(), u1 = wave == 0 ? v1 : f4zero();
#pragma unroll
    for (int m = 0; m < NRW; ++m) {
        const int k = wave + 4 * m;
        const float dj = __shfl(dsc, min(k, R - 1));
        const float w = k < R ? dj : 0.f;
        u0 = f4fma(w, o0[m], u0); u1 = f4fma(w, o1[m], u1);
    }
    if (a0) sh_v[wave][lane] = u0;
    if (a1) sh_v[wave][lane + 64] = u1;
    __syncthreads();
    if (wave != 0) return;
    float4 d0 = f4zero(), d1 = f4zero();
    if (a0) { d0 = f4add(f4add(f4add(sh_v[0][lane], sh_v[1][lane]), sh_v[2][lane]), sh_v[3][lane]); st4(DCTX + crow * Dp + c0, v0); st4(VH + crow * Dp + c0, d0); }
    if (a1) { d1 = f4add(f4add(f4add(sh_v[0][lane + 64], sh_v[1][lane + 64]), sh_v[2][lane + 64]), sh_v[3][lane + 64]); st4(DCTX + crow * Dp + c1, v1); st4(VH + crow * Dp + c1, d1); }
    if (dG) {
        float4 hu0 = f4zero(), hu1 = f4zero();
        if (a0) hu0 = ld4(U + crow * Dp + c0);
        if (a1) hu1 = ld4(U + crow * Dp + c1);
        unit_norm_bwd(d0, d1, hu0, hu1, nrmU[crow], normalize);
        if (a0) st4(dG + crow * Dp + c0, d0);
        if (a1) st4(dG + crow * Dp + c1, d1);
    }
}

// d obj[b][k][:] = sum over the inside cells of sentence b of  pm[cell][k] * dctx[cell][:] + dsc[cell][k] * u[cell][:]
// One workgroup (8 waves) per (sentence, region group y of 4): it owns the regions y + 4 m, m < NRW.  Wave w takes the float4 columns
// lane + 64 (w & 1) and the quarter w >> 1 of the sentence's cells, with NRW accumulators: a cell's two rows are fetched ONCE for all of the
// wave's regions (round 6; one wave per region re-read both rows of every cell 36 times -- 1.5 GB through L2 per launch at c3 -- and walked
// all 210 cells in one dependent chain: 159 us on the stream that ends the CLIORA backward).  The region weights of a cell come as one
// lane-indexed load (lane k holds region k) and are broadcast by readlane.  The four quarters meet in LDS and are added in quarter order:
// a fixed order, the same bits every run (not the bits of the one-chain sum it replaces).
constexpr int OGR_WAVES = 8;
template <int NRW>
static __global__ __launch_bounds__(512) void obj_grad_reduce(int B, int C, int Dp, int R, const float* __restrict__ DCTX, const float* __restrict__ U,
                                                       const float* __restrict__ PMo, const float* __restrict__ DSC,
                                                       float* __restrict__ dOBJ) {
    extern __shared__ __attribute__((aligned(16))) float4 ogr_part[];      // [3 quarters][2 halves][NRW][64]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = wave & 1, q = wave >> 1;
    const int b = blockIdx.x, k0 = blockIdx.y;
    const int nv = Dp >> 2;
    const int v = lane + 64 * half;
    const bool act = v < nv;
    const int c0 = 4 * (act ? v : 0);
    const int per = (C + 3) / 4, cbeg = q * per, cend = min(C, cbeg + per);
    float4 acc[NRW];
#pragma unroll
    for (int m = 0; m < NRW; ++m) acc[m] = f4zero();
    constexpr int NB = 8;                 // cells in flight
    for (int cc = cbeg; cc < cend; cc += NB) {
        float pmv[NB], dsv[NB];
        float4 d[NB], u[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const size_t crow = (size_t)b * C + min(cc + j, C - 1);
            const bool ok = cc + j < cend;
            pmv[j] = ok ? PMo[crow * VL_MAXR + lane] : 0.f;        // lane k: region k (VL_MAXR = 64 = the wave)
            dsv[j] = ok ? DSC[crow * VL_MAXR + lane] : 0.f;
            d[j] = act ? ld4(DCTX + crow * Dp + c0) : f4zero();
            u[j] = act ? ld4(U + crow * Dp + c0) : f4zero();
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
#pragma unroll
            for (int m = 0; m < NRW; ++m) {
                const int k = min(k0 + 4 * m, VL_MAXR - 1);
                const float pm = __shfl(pmv[j], k), ds = __shfl(dsv[j], k);
                acc[m] = f4fma(pm, d[j], acc[m]);
                acc[m] = f4fma(ds, u[j], acc[m]);
            }
        }
    }
    if (q > 0) {
#pragma unroll
        for (int m = 0; m < NRW; ++m) ogr_part[(((q - 1) * 2 + half) * NRW + m) * 64 + lane] = acc[m];
    }
    __syncthreads();
    if (q != 0 || !act) return;
#pragma unroll
    for (int m = 0; m < NRW; ++m) {
        float4 s = acc[m];
#pragma unroll
        for (int qq = 0; qq < 3; ++qq) s = f4add(s, ogr_part[((qq * 2 + half) * NRW + m) * 64 + lane]);
        const int k = k0 + 4 * m;
        if (k < R) st4(dOBJ + ((size_t)b * R + k) * Dp + c0, s);
    }
}

// ---------------------------------------------------------------------------------
// Functors for the span-region / word-region scorers  einsum('abx,cdx->acbd', Q, O)  (cliora.py:457-466)
//   Q rows r = (sentence a, cell b), O rows = (image c, region d);  out[a][c][b][d]
// ---------------------------------------------------------------------------------
// A rows = sum of two charts (inside_h + outside_h), or a single matrix when Q1 == nullptr
struct SumRowsA {
    const float *Q0, *Q1; int ld;
    struct Ctx { const float *p, *q; };
    using Raw = Raw2;
    __device__ Ctx row(int r) const { return Ctx{Q0 + (size_t)r * ld, Q1 ? Q1 + (size_t)r * ld : nullptr}; }
    __device__ Raw fetch(const Ctx& c, int k) const { return Raw{ld4(c.p + k), c.q ? ld4(c.q + k) : f4zero()}; }
    __device__ float4 finish(const Ctx&, const Raw& r) const { return f4add(r.u, r.v); }
    static constexpr bool kSide = false;
    __device__ void side(const Ctx&, int, float4) const {}
    __device__ float val(const Ctx& c, int col) const { return c.p[col] + (c.q ? c.q[col] : 0.f); }
};
// out[((a*B + c)*Cq + b)*R + d] = v (+ add[...same index in a (B,B,Cadd,R) tensor])
struct ScoreStoreE {
    float* out; int B, Cq, R; const float* add; int Cadd;
    struct RCtx { int a, b; };
    __device__ RCtx row(int r) const { const int a = r / Cq; return RCtx{a, r - a * Cq}; }
    __device__ void put(const RCtx& rc, int col, float v) const {
        if (col >= B * R) return;
        const int c = col / R, d = col - c * R;
        const size_t o = (((size_t)rc.a * B + c) * Cq + rc.b) * R + d;
        if (add) v += add[(((size_t)rc.a * B + c) * Cadd + rc.b) * R + d];
        out[o] = v;
    }
    __device__ void store4(const RCtx& rc, int col, float4 v) const {
        if ((R & 3) == 0 && col + 3 < B * R && !add) {
            const int c = col / R, d = col - c * R;
            st4(out + (((size_t)rc.a * B + c) * Cq + rc.b) * R + d, v);
            return;
        }
        put(rc, col, v.x); put(rc, col + 1, v.y); put(rc, col + 2, v.z); put(rc, col + 3, v.w);
    }
};
// Region-max scorer (ContrastiveLoss, trainer.py:101): key = (order-preserving bits of the score) << 32 | ~region, merged by a
// 64-bit atomic max -- exact and order-independent: the largest score wins, the smallest region index among equal scores
// (torch.max).  keys[(a*B + c)*Cq + b], zeroed before the launch.
__device__ __forceinline__ unsigned long long region_key(float v, int d) {
    const uint32_t u = __float_as_uint(v);
    const uint32_t o = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((unsigned long long)o << 32) | (uint32_t)(0xFFFFFFFFu - (uint32_t)d);
}
struct ScoreMaxE {
    unsigned long long* keys; int B, Cq, R;
    struct RCtx { int a, b; };
    __device__ RCtx row(int r) const { const int a = r / Cq; return RCtx{a, r - a * Cq}; }
    __device__ void put(const RCtx& rc, int col, float v) const {
        if (col >= B * R) return;
        const int c = col / R, d = col - c * R;
        atomicMax(keys + ((size_t)rc.a * B + c) * Cq + rc.b, region_key(v, d));
    }
    __device__ void store4(const RCtx& rc, int col, float4 v) const {
        if ((R & 3) == 0 && col + 3 < B * R) {       // the four columns are four regions of one image: one atomic
            const int c = col / R, d = col - c * R;
            unsigned long long k = region_key(v.x, d);
            const unsigned long long k1 = region_key(v.y, d + 1), k2 = region_key(v.z, d + 2), k3 = region_key(v.w, d + 3);
            k = k1 > k ? k1 : k; k = k2 > k ? k2 : k; k = k3 > k ? k3 : k;
            atomicMax(keys + ((size_t)rc.a * B + c) * Cq + rc.b, k);
            return;
        }
        put(rc, col, v.x); put(rc, col + 1, v.y); put(rc, col + 2, v.z); put(rc, col + 3, v.w);
    }
};
static __global__ void region_keys_decode(const unsigned long long* __restrict__ keys, size_t n, float* __restrict__ vmax, int32_t* __restrict__ arg) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const unsigned long long k = keys[e];
    const uint32_t o = (uint32_t)(k >> 32);
    const uint32_t u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
    vmax[e] = __uint_as_float(u);
    if (arg) arg[e] = (int32_t)(0xFFFFFFFFu - (uint32_t)k);
}
// Backward of the region max, sparse: one region per (sentence a, image c, span b) carries the cotangent.
//   d_sum_h[a][b][:] = sum_c g[a][c][b] * O[c][arg[a][c][b]][:]            one workgroup per (a, b), waves over c, fixed order
// O = the padded region matrix (B*R rows of stride Dp, 16-byte aligned rows), out rows have stride D.
static __global__ __launch_bounds__(256) void region_max_bwd_rows(int B, int Cq, int R, int D, int Dp, const float* __restrict__ G,
                                                                  const int32_t* __restrict__ arg, const float* __restrict__ O, int ldo,
                                                                  float* __restrict__ out) {
    __shared__ float4 sh[3][2][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = blockIdx.x, a = r / Cq, b = r - a * Cq;
    const int nv = Dp >> 2;
    const bool act0 = lane < nv, act1 = lane + 64 < nv;
    const int col0 = 4 * lane, col1 = 4 * (lane + 64);
    float4 acc0 = f4zero(), acc1 = f4zero();
    const size_t base = (size_t)a * B * Cq + b;
    for (int c0 = wave; c0 < B; c0 += 16) {
        float gv[4]; const float* row[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = c0 + 4 * j;
            const bool ok = c < B;
            const size_t o = base + (size_t)(ok ? c : 0) * Cq;
            gv[j] = ok ? G[o] : 0.f;
            row[j] = O + ((size_t)(ok ? c : 0) * R + arg[o]) * ldo;
        }
        float4 v0[4], v1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {       // a zero cotangent (every span beyond the first C / 2 under the contrastive loss, trainer.py:126) adds nothing: no row fetch
            const bool live = gv[j] != 0.f;
            v0[j] = (act0 && live) ? ld4(row[j] + col0) : f4zero(); v1[j] = (act1 && live) ? ld4(row[j] + col1) : f4zero();
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc0 = f4fma(gv[j], v0[j], acc0); acc1 = f4fma(gv[j], v1[j], acc1); }
    }
    if (wave > 0) { sh[wave - 1][0][lane] = acc0; sh[wave - 1][1][lane] = acc1; }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int w = 0; w < 3; ++w) { acc0 = f4add(acc0, sh[w][0][lane]); acc1 = f4add(acc1, sh[w][1][lane]); }
    float* o = out + (size_t)r * D;
    const float a0[4] = {acc0.x, acc0.y, acc0.z, acc0.w}, a1[4] = {acc1.x, acc1.y, acc1.z, acc1.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (col0 + j < D) o[col0 + j] = a0[j];
        if (col1 + j < D) o[col1 + j] = a1[j];
    }
}
// S = P + Q for two aligned (n4 float4) arrays: the summed chart rows of the region-max backward when no padding is needed
static __global__ void add_rows4(const float4* __restrict__ P, const float4* __restrict__ Q, size_t n4, float4* __restrict__ S) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n4) { const float4 a = P[e], b = Q[e]; S[e] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
}
//   d_obj[c][d][:] = sum over (a, b) with arg[a][c][b] == d of g[a][c][b] * S[a][b][:]     one workgroup per (c, d) and CHUNK of
// sentences a (blockIdx.y): the arg max of an image is heavily skewed towards a few regions, so a region's matches are cut
// into nchunk independent partial sums (part[chunk][c*R + d][:], rows of stride Dp, added in chunk order by slab_reduce).  The waves
// scan the chunk's (sentence, span) entries in a fixed order and add the matching rows of S = inside_h + outside_h (stride Dp).
static __global__ __launch_bounds__(256) void region_max_bwd_obj(int B, int Cq, int R, int Dp, int a_per_chunk, const float* __restrict__ G,
                                                                 const int32_t* __restrict__ arg, const float* __restrict__ S,
                                                                 float* __restrict__ part) {
    __shared__ float4 sh[3][2][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x / R, d = blockIdx.x - c * R;
    const int nv = Dp >> 2;
    const bool act0 = lane < nv, act1 = lane + 64 < nv;
    const int col0 = 4 * lane, col1 = 4 * (lane + 64);
    float4 acc0 = f4zero(), acc1 = f4zero();
    const int a_lo = blockIdx.y * a_per_chunk, a_hi = min(B, a_lo + a_per_chunk);
    for (int a = a_lo + wave; a < a_hi; a += 4) {
        const size_t base = ((size_t)a * B + c) * Cq;
        for (int b0 = 0; b0 < Cq; b0 += 64) {
            const int b = b0 + lane;
            const float gq = b < Cq ? G[base + b] : 0.f;
            const bool hit = gq != 0.f && arg[base + b] == d;      // a zero cotangent adds nothing (trainer.py:126: only the first C / 2 spans carry one)
            const float gl = hit ? gq : 0.f;
            unsigned long long m = __ballot(hit);
            while (m) {                              // up to four matches in flight, in span order
                int bb[4]; float gv[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int l = m ? __ffsll((long long)m) - 1 : 0;
                    gv[j] = m ? __shfl(gl, l) : 0.f;
                    bb[j] = b0 + l;
                    if (m) m &= m - 1;
                }
                float4 v0[4], v1[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float* row = S + ((size_t)a * Cq + bb[j]) * Dp;
                    v0[j] = act0 ? ld4(row + col0) : f4zero();
                    v1[j] = act1 ? ld4(row + col1) : f4zero();
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) { acc0 = f4fma(gv[j], v0[j], acc0); acc1 = f4fma(gv[j], v1[j], acc1); }
            }
        }
    }
    if (wave > 0) { sh[wave - 1][0][lane] = acc0; sh[wave - 1][1][lane] = acc1; }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int w = 0; w < 3; ++w) { acc0 = f4add(acc0, sh[w][0][lane]); acc1 = f4add(acc1, sh[w][1][lane]); }
    float* o = part + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * Dp;
    if (act0) st4(o + col0, acc0);
    if (act1) st4(o + col1, acc1);
}

// ContrastiveLoss on the region maxima (trainer.py:103-128) with its gradient, one workgroup per span b < C/2.
//   part[b] = sum_x marg[x][b] vl[x][b];  last[b][x] = marg[x][b] vl[x][b] (the in_s[x][C-1] term of the marginal's gradient)
constexpr float CONTRASTIVE_FLOOR = 1e-8f;   // trainer.py:86 min_val
static __global__ __launch_bounds__(256) void contrastive_spans(int B, int Cq, const float* __restrict__ M, const float* __restrict__ IS,
                                                                const float* __restrict__ OS, float margin, float k,
                                                                float* __restrict__ part, float* __restrict__ last,
                                                                float* __restrict__ dM, float* __restrict__ dIS, float* __restrict__ dOS) {
    extern __shared__ float sm[];
    float* S = sm;                       // [B][B+1]: s[a][c] of this span (padded rows: column walks stay conflict-free)
    float* diag = S + B * (B + 1);       // [B]
    float* marg = diag + B;              // [B]
    float* vl = marg + B;                // [B]
    float* dd = vl + B;                  // [B] gradient reaching the diagonal through the subtracted positives
    float* red = dd + B;                 // [256]
    const int b = blockIdx.x, tid = threadIdx.x, ld = B + 1;
    for (int e = tid; e < B * B; e += 256) {
        const int a = e / B, c = e - a * B;
        S[a * ld + c] = M[((size_t)a * B + c) * Cq + b];
    }
    __syncthreads();
    if (tid < B) {
        diag[tid] = S[tid * ld + tid];
        marg[tid] = expf(IS[(size_t)tid * Cq + b] + OS[(size_t)tid * Cq + b] - IS[(size_t)tid * Cq + Cq - 1]);
    }
    __syncthreads();
    const float invB = 1.f / (float)B;
    if (tid < B) {                       // row x = tid: text hinge over the images; column x: image hinge over the texts
        const int x = tid;
        float lt = 0.f, li = 0.f;
        for (int j = 0; j < B; ++j) {
            if (j == x) continue;
            lt += fmaxf(margin + S[x * ld + j] - diag[x], CONTRASTIVE_FLOOR);
            li += fmaxf(margin + S[j * ld + x] - diag[x], CONTRASTIVE_FLOOR);
        }
        vl[x] = (lt + li) * invB;
    }
    __syncthreads();
    // gradient: d txt[a][c] = k marg[a] / B, d img[a][c] = k marg[c] / B, through the clamp where its argument >= the floor
    if (tid < B) {
        const int x = tid;
        float g = 0.f;
        for (int j = 0; j < B; ++j) {
            if (j == x) continue;
            if (margin + S[x * ld + j] - diag[x] >= CONTRASTIVE_FLOOR) g -= k * marg[x] * invB;      // txt row x subtracts diag[x]
            if (margin + S[j * ld + x] - diag[x] >= CONTRASTIVE_FLOOR) g -= k * marg[x] * invB;      // img column x (-> vl[x]) subtracts diag[x]
        }
        dd[x] = g;
    }
    __syncthreads();
    for (int e = tid; e < B * B; e += 256) {
        const int a = e / B, c = e - a * B;
        float g;
        if (a == c) g = dd[a];
        else {
            const float s = S[a * ld + c];
            g = (margin + s - diag[a] >= CONTRASTIVE_FLOOR ? k * marg[a] * invB : 0.f) + (margin + s - diag[c] >= CONTRASTIVE_FLOOR ? k * marg[c] * invB : 0.f);
        }
        dM[((size_t)a * B + c) * Cq + b] = g;
    }
    float p = 0.f;
    if (tid < B) {
        const float mv = marg[tid] * vl[tid];
        p = mv;
        dIS[(size_t)tid * Cq + b] = k * mv;          // d marg = k vl, d (in_s + out_s) = d marg * marg
        dOS[(size_t)tid * Cq + b] = k * mv;
        last[(size_t)b * B + tid] = mv;
    }
    red[tid] = p;
    __syncthreads();
    for (int s2 = 128; s2 > 0; s2 >>= 1) {
        if (tid < s2) red[tid] += red[tid + s2];
        __syncthreads();
    }
    if (tid == 0) part[b] = red[0];
}
// loss = k sum_b part[b];  d in_s[x][C-1] = -k sum_b last[b][x]   (fixed order: the same bits as one lane adding b = 0, 1, ...)
// 256 threads: wave 0 prefetches the partial losses (lane l holds part[l], part[l + 64], ...) and lane 0 adds them in b order out of LDS;
// waves 1..3 take the columns x < B, four loads in flight (the round-5 form was one dependent global load per addend on a single lane: 29 us
// at C = 210).
static __global__ __launch_bounds__(256) void contrastive_finish(int B, int Cq, int nb, const float* __restrict__ part, const float* __restrict__ last, float k,
                                                                 float* __restrict__ loss, float* __restrict__ dIS) {
    __shared__ float sp[1024];
    const int tid = threadIdx.x;
    if (tid < 64) {
        const int nbc = min(nb, 1024);
        for (int b = tid; b < nbc; b += 64) sp[b] = part[b];
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0);
        if (tid == 0) {
            float s = 0.f;
            for (int b = 0; b < nbc; ++b) s += sp[b];
            for (int b = nbc; b < nb; ++b) s += part[b];
            loss[0] = k * s;
        }
        return;
    }
    for (int x = tid - 64; x < B; x += 192) {
        float s = 0.f;
        int b = 0;
        for (; b + 4 <= nb; b += 4) {
            const float v0 = last[(size_t)b * B + x], v1 = last[(size_t)(b + 1) * B + x], v2 = last[(size_t)(b + 2) * B + x], v3 = last[(size_t)(b + 3) * B + x];
            s += v0; s += v1; s += v2; s += v3;
        }
        for (; b < nb; ++b) s += last[(size_t)b * B + x];
        dIS[(size_t)x * Cq + Cq - 1] -= k * s;
    }
}

// A(r, k) = dScore[a][c][b][d] with r = (a, b), k = (c, d); k >= B*R reads as zero
struct ScoreGradA {
    const float* G; int B, Cq, R;
    struct Ctx { const float* base; };
    using Raw = float4;
    __device__ Ctx row(int r) const { const int a = r / Cq, b = r - a * Cq; return Ctx{G + ((size_t)a * B * Cq + b) * R}; }
    __device__ float val(const Ctx& c, int k) const {
        if (k >= B * R) return 0.f;
        const int cc = k / R, d = k - cc * R;
        return c.base[(size_t)cc * Cq * R + d];
    }
    __device__ Raw fetch(const Ctx& c, int k) const {
        if ((R & 3) == 0 && k + 3 < B * R) {
            const int cc = k / R, d = k - cc * R;
            return ld4(c.base + (size_t)cc * Cq * R + d);
        }
        return make_float4(val(c, k), val(c, k + 1), val(c, k + 2), val(c, k + 3));
    }
    __device__ float4 finish(const Ctx&, const Raw& v) const { return v; }
    static constexpr bool kSide = false;
    __device__ void side(const Ctx&, int, float4) const {}
};
// A(r, k) = dAll[a][c][b][d] + (b < L2 ? dVg[a][c][b][d] : 0): eval-mode vg_atten = all_atten[:, :, :L] + ...
struct ScoreGrad2A {
    const float* G1; const float* G2; int B, C1, C2, R;
    struct Ctx { const float *b1, *b2; };
    using Raw = Raw2;
    __device__ Ctx row(int r) const {
        const int a = r / C1, b = r - a * C1;
        return Ctx{G1 ? G1 + ((size_t)a * B * C1 + b) * R : nullptr, (G2 && b < C2) ? G2 + ((size_t)a * B * C2 + b) * R : nullptr};
    }
    __device__ float one(const float* base, int Cq, int k) const {
        if (!base || k >= B * R) return 0.f;
        const int cc = k / R, d = k - cc * R;
        return base[(size_t)cc * Cq * R + d];
    }
    __device__ float val(const Ctx& c, int k) const { return one(c.b1, C1, k) + one(c.b2, C2, k); }
    __device__ float4 four(const float* base, int Cq, int k) const {
        if (base && (R & 3) == 0 && k + 3 < B * R) {
            const int cc = k / R, d = k - cc * R;
            return ld4(base + (size_t)cc * Cq * R + d);
        }
        return make_float4(one(base, Cq, k), one(base, Cq, k + 1), one(base, Cq, k + 2), one(base, Cq, k + 3));
    }
    __device__ Raw fetch(const Ctx& c, int k) const { return Raw{four(c.b1, C1, k), four(c.b2, C2, k)}; }
    __device__ float4 finish(const Ctx&, const Raw& r) const { return f4add(r.u, r.v); }
    static constexpr bool kSide = false;
    __device__ void side(const Ctx&, int, float4) const {}
};

// out[r][:D] = unit-norm backward of V[r] through Hn[r] = X[r] / max(|X[r]|, eps)
static __global__ __launch_bounds__(256) void rows_unit_bwd(int nrows, int Dp, int D, const float* __restrict__ V, const float* __restrict__ Hn,
                                                     const float* __restrict__ nrm, int normalize, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= nrows) return;
    const int nv = Dp >> 2;
    const bool a0 = lane < nv, a1 = lane + 64 < nv;
    const int c0 = 4 * lane, c1 = 4 * (lane + 64);
    float4 v0 = f4zero(), v1 = f4zero(), h0 = f4zero(), h1 = f4zero();
    if (a0) { v0 = ld4(V + (size_t)r * Dp + c0); h0 = ld4(Hn + (size_t)r * Dp + c0); }
    if (a1) { v1 = ld4(V + (size_t)r * Dp + c1); h1 = ld4(Hn + (size_t)r * Dp + c1); }
    unit_norm_bwd(v0, v1, h0, h1, nrm[r], normalize);
    float* o = out + (size_t)r * D;
    const float vs[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (a0 && c0 + j < D) o[c0 + j] = vs[j];
        if (a1 && c1 + j < D) o[c1 + j] = vs[4 + j];
    }
}

// adds the optional second matrix row-wise:  out[r][col] = v (+ prev)   plain (rows x ld) store with accumulate flag
struct StoreAccE {
    float* out; int ld, ncols, accumulate;
    struct RCtx { float* o; };
    __device__ RCtx row(int r) const { return RCtx{out + (size_t)r * ld}; }
    __device__ void store4(const RCtx& rc, int col, float4 v) const {
        if ((ld & 3) == 0 && col + 3 < ncols) {
            if (accumulate) v = f4add(v, ld4(rc.o + col));
            st4(rc.o + col, v);
            return;
        }
        const float vs[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (col + j < ncols) rc.o[col + j] = vs[j] + (accumulate ? rc.o[col + j] : 0.f);
    }
};

}  // namespace cliora
