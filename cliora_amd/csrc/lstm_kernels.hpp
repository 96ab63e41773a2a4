// DioraTreeLSTM kernels (gfx950).  PARITY UNPINNED: the reference ships this composition only
// as commented-out text (cliora/net/vg.py:28-76):
//   leaf     [u,i,o]       = chunk3(x W^T + B[:3D]);      c = sig(i) tanh(u);  h = sig(o) tanh(c)
//   compose  [u,i,o,f0,f1] = chunk5([a;b] U^T + B);
//            c = sig(f0+k) c_a + sig(f1+k) c_b + sig(i) tanh(u);  h = sig(o) tanh(c)      (k = 1 inside, 0 outside)
// Factored like the MLP: every CELL is projected once (PL = U[:, :D] h + B and PR = U[:, D:] h,
// five gate blocks each, plus QL = mat^T h), so a span PAIR needs no matmul at all -- only the
// gate arithmetic on PL(a) + PR(b) and the two child cell states.  The whole TreeLSTM pair path is
// therefore HBM-bound: 12 D floats read and 2 D written per pair in the forward.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "chart_kernels.hpp"

namespace cliora {

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float4 f4map(float4 a, float (*f)(float)) { return make_float4(f(a.x), f(a.y), f(a.z), f(a.w)); }
__device__ __forceinline__ float4 f4sig(float4 a, float k) { return make_float4(sigm(a.x + k), sigm(a.y + k), sigm(a.z + k), sigm(a.w + k)); }
__device__ __forceinline__ float4 f4tanh(float4 a) { return make_float4(tanhf(a.x), tanhf(a.y), tanhf(a.z), tanhf(a.w)); }
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

// ---- one span pair per wave: gates from PL(a) + PR(b), child cell states, -> Y = h, X = c
static __global__ __launch_bounds__(256) void lstm_pair_fwd(int rowbase, int nrows, int Dp, const int32_t* __restrict__ arow, const int32_t* __restrict__ brow,
                                                     const float* __restrict__ PA, int ldA, const float* __restrict__ PB, int ldB,
                                                     const float* __restrict__ CA, const float* __restrict__ CB, float kf,
                                                     float* __restrict__ Y, float* __restrict__ X) {
    const int lane = threadIdx.x & 63;
    const int rl = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (rl >= nrows) return;
    const size_t r = (size_t)rowbase + rl;
    const int ar = arow[r], br = brow[r];
    const float* pa = PA + (size_t)ar * ldA;
    const float* pb = PB + (size_t)br * ldB;
    const float* ca = CA + (size_t)ar * Dp;
    const float* cb = CB + (size_t)br * Dp;
    const int nv = Dp >> 2;
    for (int v = lane; v < nv; v += 64) {
        const int c4 = 4 * v;
        const float4 u = f4tanh(f4add(ld4(pa + c4), ld4(pb + c4)));
        const float4 i = f4sig(f4add(ld4(pa + Dp + c4), ld4(pb + Dp + c4)), 0.f);
        const float4 o = f4sig(f4add(ld4(pa + 2 * Dp + c4), ld4(pb + 2 * Dp + c4)), 0.f);
        const float4 f0 = f4sig(f4add(ld4(pa + 3 * Dp + c4), ld4(pb + 3 * Dp + c4)), kf);
        const float4 f1 = f4sig(f4add(ld4(pa + 4 * Dp + c4), ld4(pb + 4 * Dp + c4)), kf);
        const float4 c = f4add(f4add(f4mul(f0, ld4(ca + c4)), f4mul(f1, ld4(cb + c4))), f4mul(i, u));
        st4(X + r * Dp + c4, c);
        st4(Y + r * Dp + c4, f4mul(o, f4tanh(c)));
    }
}

// ---- softmax-weighted sums of h and c over the splits + unit norm of both (one workgroup per cell)
static __global__ __launch_bounds__(256) void lstm_aggregate_fwd(LevelArgs g, const float* __restrict__ Y, const float* __restrict__ X,
                                                          const float* __restrict__ Pp, int normalize, float* __restrict__ H,
                                                          float* __restrict__ Cc, float* __restrict__ nrmH, float* __restrict__ nrmC) {
    // waves 0,1 aggregate h (columns split lane / lane+64 as elsewhere), waves 2,3 do the same for c
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t = blockIdx.x;
    const int b = t / g.Lc, p = t - b * g.Lc;
    if (wave & 1) return;                       // one wave per vector is enough (row = 2 float4 per lane)
    const bool isC = wave >= 2;
    const float* SRC = isC ? X : Y;
    const int row0 = g.rowbase + t * g.N;
    const int nv = g.Dp >> 2;
    const bool a0 = lane < nv, a1 = lane + 64 < nv;
    float4 v0 = f4zero(), v1 = f4zero();
    for (int n0 = 0; n0 < g.N; n0 += 4) {
        float pn[4];
        float4 y0[4], y1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = min(n0 + j, g.N - 1);
            pn[j] = (n0 + j < g.N) ? Pp[row0 + n] : 0.f;
            const float* y = SRC + (size_t)(row0 + n) * g.Dp;
            y0[j] = a0 ? ld4(y + 4 * lane) : f4zero();
            y1[j] = a1 ? ld4(y + 4 * (lane + 64)) : f4zero();
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { v0 = f4fma(pn[j], y0[j], v0); v1 = f4fma(pn[j], y1[j], v1); }
    }
    const float nr = sqrtf(wave_sum(f4dot(v0, v0) + f4dot(v1, v1)));
    const float den = normalize ? fmaxf(nr, UNIT_EPS) : 1.f;
    const size_t crow = (size_t)b * g.C + g.off + p;
    float* h = (isC ? Cc : H) + crow * g.Dp;
    if (a0) st4(h + 4 * lane, make_float4(v0.x / den, v0.y / den, v0.z / den, v0.w / den));
    if (a1) st4(h + 4 * (lane + 64), make_float4(v1.x / den, v1.y / den, v1.z / den, v1.w / den));
    if (lane == 0) (isC ? nrmC : nrmH)[crow] = nr;
}

// ---- backward, gather for the cells of one level.  One workgroup per cell, THREADS over the
// output columns (the projection rows are 5*Dp wide), every thread walks the cell's use lists.
//   inside cell:  dPL(5Dp) = sum_{left uses + sibling uses} DA;  dPR(5Dp) = sum_{right uses} DA;
//                 dQL = sum_{left} ds H(right) + sum_{sibling} ds OH(parent);
//                 vH = ext + sum_{right} ds QL(left);  vC = ext + sum_{left,sibling} DCA + sum_{right} DCB
//   outside cell: dPRo(5Dp) = sum_{parent uses} DA;  vH = ext + sum ds QL(sibling);  vC = ext + sum DCB
__device__ __forceinline__ float4 sum_rows(const UseTab& ut, int c, int b, const float* __restrict__ SRC, int ld, int col) {
    float4 acc = f4zero();
    const int beg = ut.off[c], end = ut.off[c + 1];
    for (int u0 = beg; u0 < end; u0 += 8) {            // eight rows in flight; accumulation stays in use order
        float4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int uu = min(u0 + j, end - 1);
            const size_t r = (size_t)ut.row[uu] + (size_t)b * ut.stride[uu];
            v[j] = ld4(SRC + r * ld + col);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) if (u0 + j < end) acc = f4add(acc, v[j]);
    }
    return acc;
}
__device__ __forceinline__ float4 sum_scaled(const UseTab& ut, int c, int b, int bC, const float* __restrict__ DS, const float* __restrict__ SRC,
                                             int ld, int col) {
    float4 acc = f4zero();
    const int beg = ut.off[c], end = ut.off[c + 1];
    for (int u0 = beg; u0 < end; u0 += 8) {
        float4 v[8];
        float ds[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int uu = min(u0 + j, end - 1);
            const size_t r = (size_t)ut.row[uu] + (size_t)b * ut.stride[uu];
            ds[j] = (u0 + j < end) ? DS[r] : 0.f;
            v[j] = ld4(SRC + (size_t)(bC + ut.partner[uu]) * ld + col);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc = f4fma(ds[j], v[j], acc);
    }
    return acc;
}
__device__ __forceinline__ float sum_ds(const UseTab& ut, int c, int b, const float* __restrict__ DS) {
    float acc = 0.f;
    for (int u = ut.off[c]; u < ut.off[c + 1]; ++u) acc += DS[(size_t)ut.row[u] + (size_t)b * ut.stride[u]];
    return acc;
}

static __global__ __launch_bounds__(256) void lstm_gather_bwd_in(LevelArgs g, int D, const float* __restrict__ dH_ext, const float* __restrict__ dC_ext,
                                                          const float* __restrict__ dS_ext, UseTab ina, UseTab inb, UseTab outa, int with_outside,
                                                          const float* __restrict__ DA, const float* __restrict__ DCA, const float* __restrict__ DCB,
                                                          const float* __restrict__ DS, const float* __restrict__ PI, int ldpi,
                                                          const float* __restrict__ IH, const float* __restrict__ OH,
                                                          float* __restrict__ dPI, float* __restrict__ VH, float* __restrict__ VC,
                                                          float* __restrict__ dStot) {
    const int t = blockIdx.x, tid = threadIdx.x;
    const int b = t / g.Lc, p = t - b * g.Lc;
    const int c = g.off + p;
    const size_t crow = (size_t)b * g.C + c;
    const int Dp = g.Dp, bC = b * g.C;
    const int n5 = (5 * Dp) >> 2, n1 = Dp >> 2;
    float* o = dPI + crow * ldpi;
    // grid.y deals the 13*Dp/4 output float4 columns out in chunks of 256: a thread walks the use lists once
    for (int v = blockIdx.y * 256 + tid; v < 2 * n5 + 3 * n1; v += 256 * gridDim.y) {
        if (v < n5) {                                  // dPL
            float4 a = sum_rows(ina, c, b, DA, 5 * Dp, 4 * v);
            if (with_outside) a = f4add(a, sum_rows(outa, c, b, DA, 5 * Dp, 4 * v));
            st4(o + 4 * v, a);
        } else if (v < 2 * n5) {                       // dPR
            const int col = 4 * (v - n5);
            st4(o + 5 * Dp + col, sum_rows(inb, c, b, DA, 5 * Dp, col));
        } else if (v < 2 * n5 + n1) {                  // dQL
            const int col = 4 * (v - 2 * n5);
            float4 a = sum_scaled(ina, c, b, bC, DS, IH, Dp, col);
            if (with_outside) a = f4add(a, sum_scaled(outa, c, b, bC, DS, OH, Dp, col));
            st4(o + 10 * Dp + col, a);
        } else if (v < 2 * n5 + 2 * n1) {              // vH
            const int col = 4 * (v - 2 * n5 - n1);
            float4 a = dH_ext ? ld_ext(dH_ext + crow * D, D, col) : f4zero();
            a = f4add(a, sum_scaled(inb, c, b, bC, DS, PI + 10 * Dp, ldpi, col));
            st4(VH + crow * Dp + col, a);
        } else {                                       // vC
            const int col = 4 * (v - 2 * n5 - 2 * n1);
            float4 a = dC_ext ? ld_ext(dC_ext + crow * D, D, col) : f4zero();
            a = f4add(a, sum_rows(ina, c, b, DCA, Dp, col));
            if (with_outside) a = f4add(a, sum_rows(outa, c, b, DCA, Dp, col));
            a = f4add(a, sum_rows(inb, c, b, DCB, Dp, col));
            st4(VC + crow * Dp + col, a);
        }
    }
    if (tid == 0 && blockIdx.y == 0) {
        float vs = dS_ext ? dS_ext[crow] : 0.f;
        vs += sum_ds(inb, c, b, DS) + sum_ds(ina, c, b, DS);
        if (with_outside) vs += sum_ds(outa, c, b, DS);
        dStot[crow] = vs;
    }
}

static __global__ __launch_bounds__(256) void lstm_gather_bwd_out(LevelArgs g, int D, const float* __restrict__ dH_ext, const float* __restrict__ dC_ext,
                                                           const float* __restrict__ dS_ext, UseTab outb, const float* __restrict__ DA,
                                                           const float* __restrict__ DCB, const float* __restrict__ DS,
                                                           const float* __restrict__ PI, int ldpi, float* __restrict__ dPO,
                                                           float* __restrict__ VH, float* __restrict__ VC, float* __restrict__ dStot) {
    const int t = blockIdx.x, tid = threadIdx.x;
    const int b = t / g.Lc, p = t - b * g.Lc;
    const int c = g.off + p;
    const size_t crow = (size_t)b * g.C + c;
    const int Dp = g.Dp, bC = b * g.C;
    const int n5 = (5 * Dp) >> 2, n1 = Dp >> 2;
    for (int v = blockIdx.y * 256 + tid; v < n5 + 2 * n1; v += 256 * gridDim.y) {
        if (v < n5) {
            st4(dPO + crow * 5 * Dp + 4 * v, sum_rows(outb, c, b, DA, 5 * Dp, 4 * v));
        } else if (v < n5 + n1) {
            const int col = 4 * (v - n5);
            float4 a = dH_ext ? ld_ext(dH_ext + crow * D, D, col) : f4zero();
            st4(VH + crow * Dp + col, f4add(a, sum_scaled(outb, c, b, bC, DS, PI + 10 * Dp, ldpi, col)));
        } else {
            const int col = 4 * (v - n5 - n1);
            float4 a = dC_ext ? ld_ext(dC_ext + crow * D, D, col) : f4zero();
            st4(VC + crow * Dp + col, f4add(a, sum_rows(outb, c, b, DCB, Dp, col)));
        }
    }
    if (tid == 0 && blockIdx.y == 0) dStot[crow] = (dS_ext ? dS_ext[crow] : 0.f) + sum_ds(outb, c, b, DS);
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
    const int t = blockIdx.x;
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
    const float ds = pn * ((dp - mean) + dStot[crow] * (1.f + sn - Schart[crow]));
    if (an) DS[row0 + lane] = ds;
}

// ---- backward of one span pair (one wave per pair row): gates recomputed from PL(a) + PR(b)
//   dh = p dGh(target), dc_in = p dGc(target);  out: DA (5 gate pre-activation grads), DCA = dc f0, DCB = dc f1
static __global__ __launch_bounds__(256) void lstm_pair_bwd(int rowbase, int nrows, int Dp, const int32_t* __restrict__ arow, const int32_t* __restrict__ brow,
                                                     const int32_t* __restrict__ trow, const float* __restrict__ PA, int ldA,
                                                     const float* __restrict__ PB, int ldB, const float* __restrict__ CA,
                                                     const float* __restrict__ CB, float kf, const float* __restrict__ X,
                                                     const float* __restrict__ Pp, const float* __restrict__ dGh, const float* __restrict__ dGc,
                                                     float* __restrict__ DA, float* __restrict__ DCA, float* __restrict__ DCB) {
    const int lane = threadIdx.x & 63;
    const int rl = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (rl >= nrows) return;
    const size_t r = (size_t)rowbase + rl;
    const int ar = arow[r], br = brow[r], tr = trow[r];
    const float pn = Pp[r];
    const float* pa = PA + (size_t)ar * ldA;
    const float* pb = PB + (size_t)br * ldB;
    const float* ca = CA + (size_t)ar * Dp;
    const float* cb = CB + (size_t)br * Dp;
    const int nv = Dp >> 2;
    for (int v = lane; v < nv; v += 64) {
        const int c4 = 4 * v;
        const float4 u = f4tanh(f4add(ld4(pa + c4), ld4(pb + c4)));
        const float4 i = f4sig(f4add(ld4(pa + Dp + c4), ld4(pb + Dp + c4)), 0.f);
        const float4 o = f4sig(f4add(ld4(pa + 2 * Dp + c4), ld4(pb + 2 * Dp + c4)), 0.f);
        const float4 f0 = f4sig(f4add(ld4(pa + 3 * Dp + c4), ld4(pb + 3 * Dp + c4)), kf);
        const float4 f1 = f4sig(f4add(ld4(pa + 4 * Dp + c4), ld4(pb + 4 * Dp + c4)), kf);
        const float4 cA = ld4(ca + c4), cB = ld4(cb + c4);
        const float4 tc = f4tanh(ld4(X + r * Dp + c4));
        const float4 dh = f4scale(pn, ld4(dGh + (size_t)tr * Dp + c4));
        const float4 dc = f4add(f4scale(pn, ld4(dGc + (size_t)tr * Dp + c4)), f4mul(f4mul(dh, o), f4dtanh(tc)));
        float* da = DA + r * 5 * Dp + c4;
        st4(da, f4mul(f4mul(dc, i), f4dtanh(u)));                 // d act_u
        st4(da + Dp, f4mul(f4mul(dc, u), f4dsig(i)));             // d act_i
        st4(da + 2 * Dp, f4mul(f4mul(dh, tc), f4dsig(o)));        // d act_o
        st4(da + 3 * Dp, f4mul(f4mul(dc, cA), f4dsig(f0)));       // d act_f0
        st4(da + 4 * Dp, f4mul(f4mul(dc, cB), f4dsig(f1)));       // d act_f1
        st4(DCA + r * Dp + c4, f4mul(dc, f0));
        st4(DCB + r * Dp + c4, f4mul(dc, f1));
    }
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
