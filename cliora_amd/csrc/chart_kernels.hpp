// Chart-level kernels for the DIORA inside-outside recursion (gfx950).
//
// The compose MLP of the reference (cliora/net/diora.py:65-72)
//     h = relu(W2 relu(W1 [a;b] + b1) + b2)
// is evaluated in factored form: every chart CELL is projected once
//     PL = W1[:, :D] h + b1,   PR = W1[:, D:] h,   QL = mat^T h
// and every span PAIR then costs one add+ReLU, one D x D layer (MFMA) and one D-dot
// for the bilinear score a^T mat b = QL(a) . b  (diora.py:89-97, 125-134).
//
// Memory-bound kernels here give one wavefront to one chart cell; rows are read
// with 16-byte vector loads (Dp is a multiple of 16 floats, pad columns are zero).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemm_kernels.hpp"

namespace cliora {

constexpr float UNIT_EPS = 1e-8f;   // cliora/net/utils.py:10

// Block index -> target cell t = b * Lc + p of a level, sentence-affine (round 5): workgroups are dealt round-robin over the 8 XCDs by
// linear block id, and a cell's operand / partner rows all belong to its own sentence -- with t = block id, the cells of one sentence
// spread over all eight L2s and each of them fetches the sentence's rows; here the blocks with id = x (mod 8) take the sentences
// b = x (mod 8), so a sentence's chart rows are fetched by ONE L2.  A permutation of the blocks only: the results are the same bits.
// CLIORA_XCD_AFFINE=0 (api_common.hpp: xcd_affine()) restores t = block id.
__device__ __forceinline__ int cell_of_block(int bid, int B, int Lc, int affine) {
    const int n8 = (B >> 3) * 8 * Lc;                  // blocks of the sentences below the last multiple of 8
    if (!affine || bid >= n8) return bid;
    const int x = bid & 7, q = bid >> 3;
    const int bb = q / Lc, p = q - bb * Lc;
    return (bb * 8 + x) * Lc + p;
}

struct LevelArgs {
    int B, C, Dp, Lc, N, off;   // target cells of this level: chart row b*C + off + p, p < Lc; N splits each
    int rowbase;                // first global pair row of the level; row = rowbase + (b*Lc + p)*N + n
    int affine;                 // one-workgroup-per-cell kernels: sentence-affine block order (cell_of_block)
};

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 f4fma(float s, float4 a, float4 c) {
    return make_float4(fmaf(s, a.x, c.x), fmaf(s, a.y, c.y), fmaf(s, a.z, c.z), fmaf(s, a.w, c.w));
}
__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float f4dot(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

// ---------------------------------------------------------------------------------
// Generic padded 2-D copy / transpose / sum, driven by a descriptor table:
//   dst[r][c] = sum_s src_s[(r0_s + (T_s ? c : r))][c0_s + (T_s ? r : c)]   (0 outside the source)
// Used to pack parameters into the padded/concatenated layouts, to pad inputs,
// un-pad outputs, and to scatter packed gradients back to the reference shapes.
// ---------------------------------------------------------------------------------
struct CopySrc { const float* p; int ld, rows, cols, r0, c0, T; };
struct CopyDesc { float* dst; int ldd, drows, dcols; CopySrc s[2]; };
constexpr int MAX_COPY = 32;
struct CopyTable { CopyDesc d[MAX_COPY]; int n; };

static __global__ void copy2d_multi(CopyTable tab) {
    const CopyDesc& d = tab.d[blockIdx.y];
    const long long total = (long long)d.drows * d.dcols;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(e / d.dcols), c = (int)(e - (long long)r * d.dcols);
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const CopySrc& s = d.s[k];
            if (s.p) {
                const int sr = s.T ? c : r, sc = s.T ? r : c;
                if (sr < s.rows && sc < s.cols) v += s.p[(size_t)(s.r0 + sr) * s.ld + s.c0 + sc];
            }
        }
        d.dst[(size_t)r * d.ldd + c] = v;
    }
}

// ---------------------------------------------------------------------------------
// dst[b*C + off + p] = src_row / max(||src_row||, eps)   (utils.py:11-14)
// src row = (b*rows_per_b + p) * src_ld (src_ld = 0 broadcasts one row: the root vector, diora.py:337-356)
// also writes the raw norm and zeroes the cell's score.
// ---------------------------------------------------------------------------------
static __global__ __launch_bounds__(256) void unit_norm_rows(const float* __restrict__ src, int src_ld, int nrows, int rows_per_b,
                                                      int C, int off, int Dp, int normalize,
                                                      float* __restrict__ H, float* __restrict__ nrm, float* __restrict__ S) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= nrows) return;
    const int b = r / rows_per_b, p = r - b * rows_per_b;
    const size_t crow = (size_t)b * C + off + p;
    const float* s = src + (size_t)r * src_ld;
    const int nv = Dp >> 2;
    float4 v0 = f4zero(), v1 = f4zero();
    if (lane < nv) v0 = ld4(s + 4 * lane);
    if (lane + 64 < nv) v1 = ld4(s + 4 * (lane + 64));
    const float nr = sqrtf(wave_sum(f4dot(v0, v0) + f4dot(v1, v1)));
    const float den = normalize ? fmaxf(nr, UNIT_EPS) : 1.f;
    float* h = H + crow * Dp;
    if (lane < nv) st4(h + 4 * lane, make_float4(v0.x / den, v0.y / den, v0.z / den, v0.w / den));
    if (lane + 64 < nv) st4(h + 4 * (lane + 64), make_float4(v1.x / den, v1.y / den, v1.z / den, v1.w / den));
    if (lane == 0) { nrm[crow] = nr; S[crow] = 0.f; }
}

// ---------------------------------------------------------------------------------
// Per-split scores and their softmax for one level (diora.py:125-134 / 177-185):
//   s_n = QL(a_n) . H(b_n) + S(a_n) + S(b_n);  p = softmax_n(s);  S(target) = sum_n p_n s_n
// ---------------------------------------------------------------------------------
static __global__ __launch_bounds__(256) void pair_scores_fwd(LevelArgs g, const int32_t* __restrict__ arow, const int32_t* __restrict__ brow,
                                                       const float* __restrict__ QA, int ldA, const float* __restrict__ HB,
                                                       const float* SA, const float* SB,
                                                       float* __restrict__ Sp, float* __restrict__ Pp, float* Sout) {
    // one workgroup (4 waves) per target cell: wave w scores the splits n = w, w+4, ... (all of them
    // in flight together for L <= 16, four at a time beyond), wave 0 then does the softmax
    __shared__ float sh_s[64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t = cell_of_block(blockIdx.x, g.B, g.Lc, g.affine);
    const int b = t / g.Lc, p = t - b * g.Lc;
    const int row0 = g.rowbase + t * g.N;
    const int nv = g.Dp >> 2;
    const bool a0 = lane < nv, a1 = lane + 64 < nv;
    for (int n0 = wave; n0 < g.N; n0 += 16) {
        int ar[4], br[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = min(n0 + 4 * j, g.N - 1);
            ar[j] = arow[row0 + n];
            br[j] = brow[row0 + n];
        }
        float d[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float* qa = QA + (size_t)ar[j] * ldA;
            const float* hb = HB + (size_t)br[j] * g.Dp;
            float v = 0.f;
            if (a0) v = f4dot(ld4(qa + 4 * lane), ld4(hb + 4 * lane));
            if (a1) v += f4dot(ld4(qa + 4 * (lane + 64)), ld4(hb + 4 * (lane + 64)));
            d[j] = v;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float s = wave_sum(d[j]) + SA[ar[j]] + SB[br[j]];
            if (lane == 0 && n0 + 4 * j < g.N) sh_s[n0 + 4 * j] = s;
        }
    }
    __syncthreads();
    if (wave != 0) return;
    const float my_s = lane < g.N ? sh_s[lane] : -INFINITY;
    const float m = wave_max(my_s);
    const float e = lane < g.N ? expf(my_s - m) : 0.f;
    const float pn = e / wave_sum(e);
    if (lane < g.N) { Sp[row0 + lane] = my_s; Pp[row0 + lane] = pn; }
    const float st = wave_sum(lane < g.N ? pn * my_s : 0.f);
    if (lane == 0) Sout[(size_t)b * g.C + g.off + p] = st;
}



// ---------------------------------------------------------------------------------
// Backward, step 1 for the cells of one level: gather every use of the cell.
//   inside cell c (as left child a / right child b in the inside pass, as sibling in the outside pass):
//     dPL  = sum_{a-uses} DA[row]            dPR = sum_{b-uses} DA[row]
//     dQL  = sum_{a-uses} ds[row] * H(partner)
//     vH   = dH_ext + sum_{b-uses} ds[row] * QL(partner)       (then += dP . Wcat by the GEMM that follows)
//     vS   = dS_ext + sum_{all uses} ds[row]
// ---------------------------------------------------------------------------------
struct UseTab { const int32_t *off, *row, *stride, *partner; };

__device__ __forceinline__ float4 ld_ext(const float* base, int D, int col) {
    // cotangent rows keep the caller's stride D (may be unaligned): element loads with a bound
    float4 v = f4zero();
    if (col + 0 < D) v.x = base[col + 0];
    if (col + 1 < D) v.y = base[col + 1];
    if (col + 2 < D) v.z = base[col + 2];
    if (col + 3 < D) v.w = base[col + 3];
    return v;
}

// One use list of one cell.  The four waves of the cell's workgroup take the uses u = w, w+4, ...
// of the list, four in flight each (index loads, then all row loads, then the accumulation):
//   acc_da += DA[row];   acc_x += ds[row] * SRC[partner];   vS += ds[row]
__device__ __forceinline__ void gather_uses(const UseTab& ut, int c, int b, int bC, int wave, const float* __restrict__ DA,
                                            const float* __restrict__ DS, int Dp, const float* __restrict__ SRC, int ldsrc,
                                            int col0, int col1, bool act0, bool act1,
                                            float4& da0, float4& da1, float4& x0, float4& x1, float& vS) {
    const int beg = ut.off[c], end = ut.off[c + 1];
    for (int u0 = beg + wave; u0 < end; u0 += 16) {
        size_t r[4];
        float ds[4], m[4];
        const float* sp[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int uu = min(u0 + 4 * j, end - 1);
            m[j] = (u0 + 4 * j < end) ? 1.f : 0.f;
            r[j] = (size_t)ut.row[uu] + (size_t)b * ut.stride[uu];
            sp[j] = SRC + (size_t)(bC + ut.partner[uu]) * ldsrc;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) ds[j] = DS[r[j]] * m[j];
        float4 a0[4], a1[4], s0[4], s1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            a0[j] = act0 ? ld4(DA + r[j] * Dp + col0) : f4zero();
            s0[j] = act0 ? ld4(sp[j] + col0) : f4zero();
            a1[j] = act1 ? ld4(DA + r[j] * Dp + col1) : f4zero();
            s1[j] = act1 ? ld4(sp[j] + col1) : f4zero();
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            vS += ds[j];
            da0 = f4fma(m[j], a0[j], da0);
            da1 = f4fma(m[j], a1[j], da1);
            x0 = f4fma(ds[j], s0[j], x0);
            x1 = f4fma(ds[j], s1[j], x1);
        }
    }
}

// fixed-order sum of the four waves' partial float4 pairs through LDS, as a two-level tree ((w0 + w2) + (w1 + w3)) over a
// two-wave buffer: 20 KB instead of 40, so that a gather workgroup fits on a CU beside a level_compose_bwd workgroup (the two
// chains of the backward wavefront overlap there).  Wave 0 gets the total.
constexpr int GATHER_SLOTS = 10;
__device__ __forceinline__ void wg_sum_pairs(float4 (*sh)[GATHER_SLOTS][64], int wave, int lane, int nslots, float4* v) {
    if (wave >= 2)
        for (int k = 0; k < nslots; ++k) sh[wave - 2][k][lane] = v[k];
    __syncthreads();
    if (wave < 2)
        for (int k = 0; k < nslots; ++k) v[k] = f4add(v[k], sh[wave][k][lane]);
    __syncthreads();
    if (wave == 1)
        for (int k = 0; k < nslots; ++k) sh[0][k][lane] = v[k];
    __syncthreads();
    if (wave == 0)
        for (int k = 0; k < nslots; ++k) v[k] = f4add(v[k], sh[0][k][lane]);
}

// sib_pl / sib_ql / sib_s (round 4): the sums over the cell's sibling uses in the OUTSIDE pass (dPLo, dQLo, their ds), formed by the
// outside chain one step earlier (cell_gather_bwd_sib) -- half of an inside cell's uses leave the inside chain, the longer of the two.
// Shared weights: rows of two scratch charts, added into the PL / QL blocks here; unshared: the kernel wrote blocks 3 / 4 of dPI itself.
// with_outside: 0 no outside pass; 1 the sibling sums come from cell_gather_bwd_sib (sib_pl / sib_ql / sib_s); 2 this kernel walks the
// sibling use list itself (the steps in which the OUTSIDE chain is the longer one: api_mlp.hip picks per step) and -- unshared weights --
// leaves its sums in sib_pl / sib_ql, which then are blocks 3 / 4 of dPI.
static __global__ __launch_bounds__(256) void cell_gather_bwd_in(LevelArgs g, int D, const float* __restrict__ dH_ext, const float* __restrict__ dS_ext,
                                                          UseTab ina, UseTab inb, UseTab outa, int with_outside,
                                                          float* __restrict__ sib_pl, float* __restrict__ sib_ql, int ldsib,
                                                          const float* __restrict__ sib_s,
                                                          const float* __restrict__ DA, const float* __restrict__ DS,
                                                          const float* __restrict__ PI, int ldpi, int share,
                                                          const float* __restrict__ IH, const float* __restrict__ OH,
                                                          float* __restrict__ dPI, float* __restrict__ VH, float* __restrict__ dStot,
                                                          const float* __restrict__ bcat, float* __restrict__ dots) {
    __shared__ float4 sh[2][GATHER_SLOTS][64];
    __shared__ float sh_s[4], sh_s2[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t = cell_of_block(blockIdx.x, g.B, g.Lc, g.affine);
    const int b = t / g.Lc, p = t - b * g.Lc;
    const int c = g.off + p;
    const size_t crow = (size_t)b * g.C + c;
    const int Dp = g.Dp, nv = Dp >> 2;
    const int bC = b * g.C;
    const bool act0 = lane < nv, act1 = lane + 64 < nv;
    const int col0 = 4 * lane, col1 = 4 * (lane + 64);
    float vS = 0.f;
    // slots: 0,1 vh | 2,3 dPL | 4,5 dPR | 6,7 dQL | 8,9 (outside-pass uses) handled below
    float4 v[GATHER_SLOTS];
#pragma unroll
    for (int k = 0; k < GATHER_SLOTS; ++k) v[k] = f4zero();
    // the outside pass's share, on its way while the lists below are walked (shared weights; wave 0 adds it at the end)
    const bool sib = with_outside == 1 && share && wave == 0;
    float4 sp0 = f4zero(), sp1 = f4zero(), sq0 = f4zero(), sq1 = f4zero();
    if (sib && act0) { sp0 = ld4(sib_pl + crow * Dp + col0); sq0 = ld4(sib_ql + crow * Dp + col0); }
    if (sib && act1) { sp1 = ld4(sib_pl + crow * Dp + col1); sq1 = ld4(sib_ql + crow * Dp + col1); }
    const float ss = (with_outside == 1 && wave == 0) ? sib_s[crow] : 0.f;
    // right-child uses: partner = left child; dH += ds * QL(left);  dPR += DA
    // (CLIORA_DIAG_GATHER: wrong-result timing builds of tools/ab/anatomy.sh -- 1: no list, 2: the right-child list only, 3: no list, no dots)
#if !defined(CLIORA_DIAG_GATHER) || CLIORA_DIAG_GATHER == 2
    gather_uses(inb, c, b, bC, wave, DA, DS, Dp, PI + 2 * Dp, ldpi, col0, col1, act0, act1, v[4], v[5], v[0], v[1], vS);
#endif
    // left-child uses: partner = right child; dQL += ds * H(right);  dPL += DA
#if !defined(CLIORA_DIAG_GATHER)
    gather_uses(ina, c, b, bC, wave, DA, DS, Dp, IH, Dp, col0, col1, act0, act1, v[2], v[3], v[6], v[7], vS);
#endif
    wg_sum_pairs(sh, wave, lane, 8, v);
    if (with_outside == 2) {           // sibling uses in the outside pass, walked here: partner = parent (outside chart); own reduction round
        float4 w[4] = {f4zero(), f4zero(), f4zero(), f4zero()};
        __syncthreads();
        float vSo = 0.f;               // summed on its own, in cell_gather_bwd_sib's order: the result does not depend on who sums
        gather_uses(outa, c, b, bC, wave, DA, DS, Dp, OH, Dp, col0, col1, act0, act1, w[0], w[1], w[2], w[3], vSo);
        wg_sum_pairs(sh, wave, lane, 4, w);
        if (lane == 0) sh_s2[wave] = vSo;
        sp0 = w[0]; sp1 = w[1]; sq0 = w[2]; sq1 = w[3];
        if (!share && wave == 0) {     // unshared weights: blocks 3 / 4 of dPI (what cell_gather_bwd_sib writes in the other mode)
            if (act0) { st4(sib_pl + crow * ldsib + col0, sp0); st4(sib_ql + crow * ldsib + col0, sq0); }
            if (act1) { st4(sib_pl + crow * ldsib + col1, sp1); st4(sib_ql + crow * ldsib + col1, sq1); }
        }
        if (!share) { sp0 = sp1 = sq0 = sq1 = f4zero(); }
    }
    if (lane == 0) sh_s[wave] = vS;    // every lane of a wave holds the same vS
    __syncthreads();
    if (wave != 0) return;
    const float sso = with_outside == 2 ? ((sh_s2[0] + sh_s2[1]) + sh_s2[2]) + sh_s2[3] : ss;
    const float vs = ((sh_s[0] + sh_s[1]) + sh_s[2]) + sh_s[3] + sso + (dS_ext ? dS_ext[crow] : 0.f);
    float* o = dPI + crow * ldpi;
    if (act0) {
        const float4 e = dH_ext ? ld_ext(dH_ext + crow * D, D, col0) : f4zero();
        st4(o + col0, f4add(v[2], sp0)); st4(o + Dp + col0, v[4]); st4(o + 2 * Dp + col0, f4add(v[6], sq0));
        st4(VH + crow * Dp + col0, f4add(v[0], e));
    }
    if (act1) {
        const float4 e = dH_ext ? ld_ext(dH_ext + crow * D, D, col1) : f4zero();
        st4(o + col1, f4add(v[3], sp1)); st4(o + Dp + col1, v[5]); st4(o + 2 * Dp + col1, f4add(v[7], sq1));
        st4(VH + crow * Dp + col1, f4add(v[1], e));
    }
    if (lane == 0) dStot[crow] = vs;
#if defined(CLIORA_DIAG_GATHER) && CLIORA_DIAG_GATHER == 3
    dots = nullptr;
#endif
    if (dots) {      // H . vH of the cell without the finished vH (NormBwdLevelE): H . vHg + dP . (P - bias) over the cell's projection blocks
        float d = 0.f;
        const float* hr = IH + crow * Dp;
        const float* pr = PI + crow * ldpi;
        if (act0) {
            const float4 e = dH_ext ? ld_ext(dH_ext + crow * D, D, col0) : f4zero();
            d += f4dot(ld4(hr + col0), f4add(v[0], e));
            const float4 b = ld4(bcat + col0), pl = ld4(pr + col0);
            d += f4dot(f4add(v[2], sp0), make_float4(pl.x - b.x, pl.y - b.y, pl.z - b.z, pl.w - b.w));
            d += f4dot(v[4], ld4(pr + Dp + col0)) + f4dot(f4add(v[6], sq0), ld4(pr + 2 * Dp + col0));
        }
        if (act1) {
            const float4 e = dH_ext ? ld_ext(dH_ext + crow * D, D, col1) : f4zero();
            d += f4dot(ld4(hr + col1), f4add(v[1], e));
            const float4 b = ld4(bcat + col1), pl = ld4(pr + col1);
            d += f4dot(f4add(v[3], sp1), make_float4(pl.x - b.x, pl.y - b.y, pl.z - b.z, pl.w - b.w));
            d += f4dot(v[5], ld4(pr + Dp + col1)) + f4dot(f4add(v[7], sq1), ld4(pr + 2 * Dp + col1));
        }
        d = wave_sum(d);
        if (lane == 0) dots[crow] = d;
    }
}

// Sibling uses of the inside cells of one level in the outside pass (partner = the parent, outside chart):
//   dPLo = sum DA[row];  dQLo = sum ds[row] * OH(parent);  sib_s = sum ds[row]
// They are final once the outside backward has passed level L-2-s for an inside level s -- one step before the inside chain gets to
// that level -- so the OUTSIDE chain's stream sums them (after its cell_dsoftmax of that step) and cell_gather_bwd_in only adds the
// result: in the backward wavefront the inside chain is the longer one, and these are half of its cells' uses.
// out_pl / out_ql: row stride ldo (shared weights: scratch charts of stride Dp; unshared: blocks 3 / 4 of dPI, stride ldpi).
static __global__ __launch_bounds__(256) void cell_gather_bwd_sib(LevelArgs g, UseTab outa, const float* __restrict__ DA, const float* __restrict__ DS,
                                                           const float* __restrict__ OH, float* __restrict__ out_pl, float* __restrict__ out_ql,
                                                           int ldo, float* __restrict__ sib_s) {
    __shared__ float4 sh[2][GATHER_SLOTS][64];
    __shared__ float sh_s[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t = cell_of_block(blockIdx.x, g.B, g.Lc, g.affine);
    const int b = t / g.Lc, p = t - b * g.Lc;
    const int c = g.off + p;
    const size_t crow = (size_t)b * g.C + c;
    const int Dp = g.Dp, nv = Dp >> 2;
    const int bC = b * g.C;
    const bool act0 = lane < nv, act1 = lane + 64 < nv;
    const int col0 = 4 * lane, col1 = 4 * (lane + 64);
    float vS = 0.f;
    float4 v[GATHER_SLOTS];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = f4zero();
    gather_uses(outa, c, b, bC, wave, DA, DS, Dp, OH, Dp, col0, col1, act0, act1, v[2], v[3], v[0], v[1], vS);
    wg_sum_pairs(sh, wave, lane, 4, v);
    if (lane == 0) sh_s[wave] = vS;
    __syncthreads();
    if (wave != 0) return;
    if (act0) { st4(out_pl + crow * ldo + col0, v[2]); st4(out_ql + crow * ldo + col0, v[0]); }
    if (act1) { st4(out_pl + crow * ldo + col1, v[3]); st4(out_ql + crow * ldo + col1, v[1]); }
    if (lane == 0) sib_s[crow] = ((sh_s[0] + sh_s[1]) + sh_s[2]) + sh_s[3];
}

//   outside cell c (as parent in the outside pass):
//     dPRo = sum DA[row];  vH = dH_ext + sum ds[row] * QLo(sibling);  vS = dS_ext + sum ds[row]
static __global__ __launch_bounds__(256) void cell_gather_bwd_out(LevelArgs g, int D, const float* __restrict__ dH_ext, const float* __restrict__ dS_ext,
                                                           UseTab outb, const float* __restrict__ DA, const float* __restrict__ DS,
                                                           const float* __restrict__ PI, int ldpi, int blk_qlo,
                                                           float* __restrict__ dPO, float* __restrict__ VH, float* __restrict__ dStot,
                                                           const float* __restrict__ OHc, const float* __restrict__ PO, float* __restrict__ dots) {
    __shared__ float4 sh[2][GATHER_SLOTS][64];
    __shared__ float sh_s[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t = cell_of_block(blockIdx.x, g.B, g.Lc, g.affine);
    const int b = t / g.Lc, p = t - b * g.Lc;
    const int c = g.off + p;
    const size_t crow = (size_t)b * g.C + c;
    const int Dp = g.Dp, nv = Dp >> 2;
    const int bC = b * g.C;
    const bool act0 = lane < nv, act1 = lane + 64 < nv;
    const int col0 = 4 * lane, col1 = 4 * (lane + 64);
    float vS = 0.f;
    float4 v[GATHER_SLOTS];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = f4zero();
    gather_uses(outb, c, b, bC, wave, DA, DS, Dp, PI + (size_t)blk_qlo * Dp, ldpi, col0, col1, act0, act1, v[2], v[3], v[0], v[1], vS);
    wg_sum_pairs(sh, wave, lane, 4, v);
    if (lane == 0) sh_s[wave] = vS;
    __syncthreads();
    if (wave != 0) return;
    const float vs = ((sh_s[0] + sh_s[1]) + sh_s[2]) + sh_s[3] + (dS_ext ? dS_ext[crow] : 0.f);
    if (act0) {
        const float4 e = dH_ext ? ld_ext(dH_ext + crow * D, D, col0) : f4zero();
        st4(dPO + crow * Dp + col0, v[2]); st4(VH + crow * Dp + col0, f4add(v[0], e));
    }
    if (act1) {
        const float4 e = dH_ext ? ld_ext(dH_ext + crow * D, D, col1) : f4zero();
        st4(dPO + crow * Dp + col1, v[3]); st4(VH + crow * Dp + col1, f4add(v[1], e));
    }
    if (lane == 0) dStot[crow] = vs;
    if (dots) {      // H . vH = H . vHg + dPO . PO (the outside projection has no bias): see NormBwdLevelE
        float d = 0.f;
        if (act0) {
            const float4 e = dH_ext ? ld_ext(dH_ext + crow * D, D, col0) : f4zero();
            d += f4dot(ld4(OHc + crow * Dp + col0), f4add(v[0], e)) + f4dot(v[2], ld4(PO + crow * Dp + col0));
        }
        if (act1) {
            const float4 e = dH_ext ? ld_ext(dH_ext + crow * D, D, col1) : f4zero();
            d += f4dot(ld4(OHc + crow * Dp + col1), f4add(v[1], e)) + f4dot(v[3], ld4(PO + crow * Dp + col1));
        }
        d = wave_sum(d);
        if (lane == 0) dots[crow] = d;
    }
}

// unit-norm backward for one row held as two float4 per lane: H = g / max(||g||, eps)
__device__ __forceinline__ void unit_norm_bwd(float4& v0, float4& v1, float4 h0, float4 h1, float nr, int normalize) {
    if (!normalize) return;
    if (nr > UNIT_EPS) {
        const float dot = wave_sum(f4dot(v0, h0) + f4dot(v1, h1));
        const float inv = 1.f / nr;
        v0 = make_float4((v0.x - h0.x * dot) * inv, (v0.y - h0.y * dot) * inv, (v0.z - h0.z * dot) * inv, (v0.w - h0.w * dot) * inv);
        v1 = make_float4((v1.x - h1.x * dot) * inv, (v1.y - h1.y * dot) * inv, (v1.z - h1.z * dot) * inv, (v1.w - h1.w * dot) * inv);
    } else {
        const float inv = 1.f / UNIT_EPS;   // clamp(min=eps) passes no gradient to the norm
        v0 = make_float4(v0.x * inv, v0.y * inv, v0.z * inv, v0.w * inv);
        v1 = make_float4(v1.x * inv, v1.y * inv, v1.z * inv, v1.w * inv);
    }
}


// leaves: H = unit(T), T = tanh(U)  (diora.py:58-63, 283-292):  dU = normbwd(vH) * (1 - T^2)
static __global__ __launch_bounds__(256) void leaf_bwd_pre(int B, int L, int C, int Dp, const float* __restrict__ VH, const float* __restrict__ H,
                                                    const float* __restrict__ nrm, int normalize, const float* __restrict__ T,
                                                    float* __restrict__ dU) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= B * L) return;
    const int b = r / L, p = r - b * L;
    const size_t crow = (size_t)b * C + p;
    const int nv = Dp >> 2;
    const bool a0 = lane < nv, a1 = lane + 64 < nv;
    float4 v0 = f4zero(), v1 = f4zero(), h0 = f4zero(), h1 = f4zero();
    if (a0) { v0 = ld4(VH + crow * Dp + 4 * lane); h0 = ld4(H + crow * Dp + 4 * lane); }
    if (a1) { v1 = ld4(VH + crow * Dp + 4 * (lane + 64)); h1 = ld4(H + crow * Dp + 4 * (lane + 64)); }
    unit_norm_bwd(v0, v1, h0, h1, nrm[crow], normalize);
    if (a0) {
        const float4 t = ld4(T + (size_t)r * Dp + 4 * lane);
        st4(dU + (size_t)r * Dp + 4 * lane, make_float4(v0.x * (1.f - t.x * t.x), v0.y * (1.f - t.y * t.y), v0.z * (1.f - t.z * t.z), v0.w * (1.f - t.w * t.w)));
    }
    if (a1) {
        const float4 t = ld4(T + (size_t)r * Dp + 4 * (lane + 64));
        st4(dU + (size_t)r * Dp + 4 * (lane + 64), make_float4(v1.x * (1.f - t.x * t.x), v1.y * (1.f - t.y * t.y), v1.z * (1.f - t.z * t.z), v1.w * (1.f - t.w * t.w)));
    }
}

// outside root: OH[root] = unit(root_vector) broadcast over the batch (diora.py:337-356);
// d root_vector = sum_b normbwd(vH[b, root]).  One workgroup of 16 waves: wave w takes sentences w, w+16, ... (a lane
// owns columns lane, lane+64, ...; the row dot product is a wave reduction), the 16 partial vectors meet in LDS and are
// added in wave order -- a fixed summation order, so the result is bitwise reproducible.
constexpr int ROOT_WAVES = 16;
static __global__ __launch_bounds__(ROOT_WAVES * 64) void root_bwd(int B, int C, int Dp, const float* __restrict__ VH, const float* __restrict__ H,
                                                            const float* __restrict__ nrm, int normalize, float* __restrict__ groot) {
    __shared__ float part[ROOT_WAVES][512];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
    for (int b = wave; b < B; b += ROOT_WAVES) {
        const size_t crow = (size_t)b * C + C - 1;
        float v[8], h[8];
        float dot = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int col = lane + 64 * k;
            v[k] = col < Dp ? VH[crow * Dp + col] : 0.f;
            h[k] = col < Dp ? H[crow * Dp + col] : 0.f;
            dot = fmaf(v[k], h[k], dot);
        }
        if (!normalize) {
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] += v[k];
            continue;
        }
        const float nr = nrm[crow];
        if (nr > UNIT_EPS) {
            dot = wave_sum(dot);
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] += (v[k] - h[k] * dot) / nr;
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] += v[k] / UNIT_EPS;
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) part[wave][lane + 64 * k] = acc[k];
    __syncthreads();
    for (int col = threadIdx.x; col < Dp; col += ROOT_WAVES * 64) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < ROOT_WAVES; ++w) t += part[w][col];
        groot[col] = t;
    }
}

// ---------------------------------------------------------------------------------
// Functors for the MFMA kernels
// ---------------------------------------------------------------------------------
struct Raw2 { float4 u, v; };   // two 16-byte loads in flight (shared by the gather producers)

// A rows = plain rows of a matrix
struct PlainRowsA {
    const float* p; int ld;
    struct Ctx { const float* r; };
    using Raw = float4;
    __device__ Ctx row(int r) const { return Ctx{p + (size_t)r * ld}; }
    __device__ Raw fetch(const Ctx& c, int k) const { return ld4(c.r + k); }
    __device__ float4 finish(const Ctx&, const Raw& v) const { return v; }
    static constexpr bool kSide = false;
    __device__ void side(const Ctx&, int, float4) const {}
    __device__ float val(const Ctx& c, int col) const { return c.r[col]; }
};
// A rows = the cells of one chart level, all sentences: row r = b*Lc + p -> chart row b*C + off + p
struct LevelRowsA {
    const float* p; int ld, C, off, Lc;
    struct Ctx { const float* r; };
    using Raw = float4;
    __device__ Ctx row(int r) const { const int b = r / Lc; return Ctx{p + ((size_t)b * C + off + (r - b * Lc)) * ld}; }
    __device__ Raw fetch(const Ctx& c, int k) const { return ld4(c.r + k); }
    __device__ float4 finish(const Ctx&, const Raw& v) const { return v; }
    static constexpr bool kSide = false;
    __device__ void side(const Ctx&, int, float4) const {}
    __device__ float val(const Ctx& c, int col) const { return c.r[col]; }
};

// epilogues
struct StoreRowsE {            // out[r*ld + col] = act(v + bias[col]); ACT 0 none, 1 tanh, 2 relu; cols >= ncols skipped
    float* out; int ld; const float* bias; int act; int ncols;   // vector path when ld % 4 == 0 (16-B aligned rows)
    struct RCtx { float* o; };
    __device__ RCtx row(int r) const { return RCtx{out + (size_t)r * ld}; }
    __device__ float f(float v) const { return act == 1 ? tanhf(v) : (act == 2 ? fmaxf(v, 0.f) : v); }
    __device__ void store4(const RCtx& rc, int col, float4 v) const {
        if (col >= ncols) return;
        if (bias) { const float4 b = ld4(bias + col); v = f4add(v, b); }
        v = make_float4(f(v.x), f(v.y), f(v.z), f(v.w));
        if ((ld & 3) == 0 && col + 3 < ncols) { st4(rc.o + col, v); return; }
        rc.o[col] = v.x;
        if (col + 1 < ncols) rc.o[col + 1] = v.y;
        if (col + 2 < ncols) rc.o[col + 2] = v.z;
        if (col + 3 < ncols) rc.o[col + 3] = v.w;
    }
};
struct StoreLevelE {           // level row r -> chart row; out[crow*ld + col] = v + bias[col]   (or += when accumulate)
    float* out; int ld, C, off, Lc; const float* bias; int accumulate;
    struct RCtx { float* o; };
    __device__ RCtx row(int r) const { const int b = r / Lc; return RCtx{out + ((size_t)b * C + off + (r - b * Lc)) * ld}; }
    __device__ void store4(const RCtx& rc, int col, float4 v) const {
        if (bias) v = f4add(v, ld4(bias + col));
        if (accumulate) v = f4add(v, ld4(rc.o + col));
        st4(rc.o + col, v);
    }
};

// Projection-backward GEMM with the unit-norm backward in its epilogue (round 4): the GEMM's accumulator is dP Wcat for the level's
// rows; with vH = vHg + dP Wcat the gradient w.r.t. the aggregate g (H = g / max(||g||, eps)) is
//     dG = (vH - H (H . vH)) / ||g||        and   H . vH = H . vHg + dP . (P - bias)      (P = H Wcat^T + bias is the stored projection)
// -- the right-hand side needs no finished vH, so the gather kernels, which hold vHg and dP of the cell in registers, leave the dot
// product in `dots` and this epilogue turns each 16 x 16 tile of the product straight into dG: the cell_dnorm launch of every level
// (but the root / leaf ones, which have no GEMM) leaves both backward chains.  Same formulas as unit_norm_bwd, per element.
struct NormBwdLevelE {
    float* dG; const float* VHg; const float* H; const float* nrm; const float* dots; int ld, C, off, Lc, normalize;
    struct RCtx { float* o; const float* v; const float* h; float dot, inv; int plain; };
    __device__ RCtx row(int r) const {
        const int b = r / Lc;
        const size_t crow = (size_t)b * C + off + (r - b * Lc);
        const float nr = nrm[crow];
        const int small = !(nr > UNIT_EPS);
        return RCtx{dG + crow * ld, VHg + crow * ld, H + crow * ld, (normalize && !small) ? dots[crow] : 0.f,
                    !normalize ? 1.f : (small ? 1.f / UNIT_EPS : 1.f / nr), 0};
    }
    __device__ void store4(const RCtx& rc, int col, float4 v) const {
        const float4 g = ld4(rc.v + col), h = ld4(rc.h + col);
        st4(rc.o + col, make_float4(((v.x + g.x) - h.x * rc.dot) * rc.inv, ((v.y + g.y) - h.y * rc.dot) * rc.inv,
                                    ((v.z + g.z) - h.z * rc.dot) * rc.inv, ((v.w + g.w) - h.w * rc.dot) * rc.inv));
    }
};

}  // namespace cliora
