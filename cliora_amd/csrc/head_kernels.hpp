// Kernels of the callers either side of the chart path: the producers of the chart's
// inputs and the loss heads that read its outputs, so that a training step stays on hand-written kernels end to end.
//
//   Embed.forward / ImageEncoder.forward   cliora/net/trainer.py:204-224, cliora/net/utils.py:37-55
//       y = gather(x, index) W^T + bias        -- the fp32-input MFMA GEMMs of gemm_kernels.hpp (rows_gemm_ksplit forward and for
//                                                 the gradient of the gathered rows, tn_gemm for dW / dbias) behind a gather functor
//   ReconstructionSoftmaxLoss.forward      cliora/net/trainer.py:46-78
//       logits = [proj_pos . cell | cell proj_neg^T],  loss = mean_r (logsumexp(logits_r) - logits_r0)
//   VGLoss.forward                         cliora/net/trainer.py:139-171
//       logits[a][c] = sum_l max_r vg[a][c][l][r] / L,  loss = alpha * CE(logits, diag)
// Every reduction has a fixed order (no float atomics): results are bitwise reproducible.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "chart_kernels.hpp"

namespace cliora {

// A rows = rows of a table picked by a 64-bit index (nn.Embedding lookup), or the table's own rows (index == nullptr)
struct GatherRowsA {
    const float* p; int ld; const int64_t* idx;
    struct Ctx { const float* r; };
    using Raw = float4;
    __device__ Ctx row(int r) const { return Ctx{p + (size_t)(idx ? idx[r] : (int64_t)r) * ld}; }
    __device__ Raw fetch(const Ctx& c, int k) const { return ld4(c.r + k); }
    __device__ float4 finish(const Ctx&, const Raw& v) const { return v; }
    static constexpr bool kSide = false;
    __device__ void side(const Ctx&, int, float4) const {}
    __device__ float val(const Ctx& c, int col) const { return c.r[col]; }
};
// A rows = plain rows whose valid width is not a multiple of 16: columns >= ncols read as zero (no access beyond the row)
struct BoundedRowsA {
    const float* p; int ld, ncols;
    struct Ctx { const float* r; };
    using Raw = float4;
    __device__ Ctx row(int r) const { return Ctx{p + (size_t)r * ld}; }
    __device__ Raw fetch(const Ctx& c, int k) const {
        if (k + 3 < ncols && (ld & 3) == 0) return ld4(c.r + k);
        return make_float4(k < ncols ? c.r[k] : 0.f, k + 1 < ncols ? c.r[k + 1] : 0.f, k + 2 < ncols ? c.r[k + 2] : 0.f, k + 3 < ncols ? c.r[k + 3] : 0.f);
    }
    __device__ float4 finish(const Ctx&, const Raw& v) const { return v; }
    static constexpr bool kSide = false;
    __device__ void side(const Ctx&, int, float4) const {}
    __device__ float val(const Ctx& c, int col) const { return col < ncols ? c.r[col] : 0.f; }
};
// A rows = the leaf cells of a (B, C, D) chart: row r = b*L + l -> chart row b*C + l, bounded like BoundedRowsA
struct LeafRowsA {
    const float* p; int ld, ncols, L, C;
    struct Ctx { const float* r; };
    using Raw = float4;
    __device__ Ctx row(int r) const { const int b = r / L; return Ctx{p + ((size_t)b * C + (r - b * L)) * ld}; }
    __device__ Raw fetch(const Ctx& c, int k) const {
        if (k + 3 < ncols && (ld & 3) == 0) return ld4(c.r + k);
        return make_float4(k < ncols ? c.r[k] : 0.f, k + 1 < ncols ? c.r[k + 1] : 0.f, k + 2 < ncols ? c.r[k + 2] : 0.f, k + 3 < ncols ? c.r[k + 3] : 0.f);
    }
    __device__ float4 finish(const Ctx&, const Raw& v) const { return v; }
    static constexpr bool kSide = false;
    __device__ void side(const Ctx&, int, float4) const {}
    __device__ float val(const Ctx& c, int col) const { return col < ncols ? c.r[col] : 0.f; }
};
// epilogue: out[r*ld + col] = v + extra[r] * X[r*ldx + col]   (cols >= ncols skipped; element stores: ld may be unaligned)
struct StoreAxpyRowsE {
    float* out; int ld; const float* extra; const float* X; int ldx; int ncols;
    struct RCtx { float* o; const float* x; float e; };
    __device__ RCtx row(int r) const { return RCtx{out + (size_t)r * ld, X + (size_t)r * ldx, extra[r]}; }
    __device__ void store4(const RCtx& rc, int col, float4 v) const {
        const float a[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (col + j < ncols) rc.o[col + j] = a[j] + rc.e * rc.x[col + j];
    }
};

// ---------------------------------------------------------------------------------------------------------------------------
// VGLoss (trainer.py:139-171).  One wave per (sentence a, image c): logits[a][c] = sum_l max_r vg[a][c][l][r] / L and the region
// that attains each maximum (smallest index on ties, as torch.max on the CPU).
// ---------------------------------------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(256) void vg_logits_fwd(int B, int L, int R, const float* __restrict__ vg, float* __restrict__ logits,
                                                            int32_t* __restrict__ arg) {
    const int lane = threadIdx.x & 63;
    const int pair = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pair >= B * B) return;
    const float* src = vg + (size_t)pair * L * R;
    float sum = 0.f;
    for (int l = 0; l < L; ++l) {
        float best = -INFINITY; int bi = 0x7fffffff;
        for (int r = lane; r < R; r += 64) { const float v = src[l * R + r]; if (v > best) { best = v; bi = r; } }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best, o); const int oi = __shfl_xor(bi, o);
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        sum += best;
        if (lane == 0) arg[(size_t)pair * L + l] = bi;
    }
    if (lane == 0) logits[pair] = sum / (float)L;
}
// One workgroup: cross entropy of the (B, B) logits against the diagonal; row a on wave a % 16 (sixteen waves: a row is three dependent
// passes with a wave reduction each, 20 us on four waves at B 64).  d_logits for an upstream cotangent of 1; the per-row losses are
// added in row order by one lane.
static __global__ __launch_bounds__(1024) void vg_ce(int B, float alpha, const float* __restrict__ logits, float* __restrict__ row_loss,
                                                     float* __restrict__ d_logits, float* __restrict__ loss) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    for (int a = wave; a < B; a += nwave) {
        float m = -INFINITY;
        for (int c = lane; c < B; c += 64) m = fmaxf(m, logits[a * B + c]);
        m = wave_max(m);
        float s = 0.f;
        for (int c = lane; c < B; c += 64) s += expf(logits[a * B + c] - m);
        s = wave_sum(s);
        const float lse = m + logf(s);
        for (int c = lane; c < B; c += 64) d_logits[a * B + c] = alpha / (float)B * (expf(logits[a * B + c] - lse) - (c == a ? 1.f : 0.f));
        if (lane == 0) row_loss[a] = lse - logits[a * B + a];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int a = 0; a < B; ++a) t += row_loss[a];
        loss[0] = alpha * t / (float)B;
    }
}
static __global__ __launch_bounds__(256) void vg_scatter_bwd(size_t n, int L, int R, const float* __restrict__ d_logits, const int32_t* __restrict__ arg,
                                                             float* __restrict__ d_vg) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const size_t pl = e / R;                  // (pair, l)
    const int r = (int)(e - pl * R);
    const size_t pair = pl / L;
    d_vg[e] = (arg[pl] == r) ? d_logits[pair] / (float)L : 0.f;
}

// ---------------------------------------------------------------------------------------------------------------------------
// ReconstructionSoftmaxLoss (trainer.py:46-78), after the GEMMs: one wave per word position r.
//   logits_r = [xp_r, xn_r0 .. xn_r(K-1)],  xp_r = P_pos[r] . cell[r]  (computed here),  xn = cell P_neg^T  (a GEMM, given)
//   row_loss[r] = logsumexp(logits_r) - xp_r;   G[r][k] = softmax(logits_r)[1 + k] / N  (zero in the pad columns k >= K);
//   gxp[r] = (softmax(logits_r)[0] - 1) / N
// ---------------------------------------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(256) void recon_ce(int N, int K, int Kp, int D, int L, int C, const float* __restrict__ P, int ldp,
                                                       const float* __restrict__ OH, float* __restrict__ XN /* in: xn, out: G */,
                                                       float* __restrict__ gxp, float* __restrict__ row_loss) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= N) return;
    const int b = r / L;
    const float* cell = OH + ((size_t)b * C + (r - b * L)) * D;
    const float* pp = P + (size_t)r * ldp;
    float xp = 0.f;
    for (int k = lane; k < D; k += 64) xp += pp[k] * cell[k];
    xp = wave_sum(xp);
    float* xn = XN + (size_t)r * Kp;
    float m = xp;
    for (int k = lane; k < K; k += 64) m = fmaxf(m, xn[k]);
    m = wave_max(m);
    float s = 0.f;
    for (int k = lane; k < K; k += 64) s += expf(xn[k] - m);
    s = wave_sum(s) + expf(xp - m);
    const float lse = m + logf(s);
    const float inv = 1.f / (float)N;
    for (int k = lane; k < Kp; k += 64) xn[k] = k < K ? expf(xn[k] - lse) * inv : 0.f;
    if (lane == 0) { gxp[r] = (expf(xp - lse) - 1.f) * inv; row_loss[r] = lse - xp; }
}
// mean of n values in index order (one workgroup, fixed tree over 256 strided partial sums)
static __global__ __launch_bounds__(256) void mean_in_order(int n, const float* __restrict__ v, float* __restrict__ out) {
    __shared__ float sh[256];
    float t = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) t += v[i];
    sh[threadIdx.x] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < 256; ++i) s += sh[i];
        out[0] = s / (float)n;
    }
}
// rows *= scale[0] (the upstream cotangent, a device scalar); dst may alias src
static __global__ void scale_by_scalar(size_t n, const float* __restrict__ src, const float* __restrict__ scale, float* __restrict__ dst) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e < n) dst[e] = src[e] * scale[0];
}
// dP[r] = g[r] * cell[r] for the N word positions (the positive projection's gradient), zero-padded to ldp columns
static __global__ __launch_bounds__(256) void recon_dpos(int N, int D, int ldp, int L, int C, const float* __restrict__ gxp, const float* __restrict__ OH,
                                                         float* __restrict__ dP) {
    const int r = blockIdx.x;
    const int b = r / L;
    const float* cell = OH + ((size_t)b * C + (r - b * L)) * D;
    const float g = gxp[r];
    for (int k = threadIdx.x; k < ldp; k += 256) dP[(size_t)r * ldp + k] = k < D ? g * cell[k] : 0.f;
}
// concatenated lookup index: tokens then negatives
static __global__ void concat_index(int n0, const int64_t* __restrict__ a, int n1, const int64_t* __restrict__ b, int64_t* __restrict__ out) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < n0) out[e] = a[e];
    else if (e < n0 + n1) out[e] = b[e - n0];
}

// ---------------------------------------------------------------------------------------------------------------------------
// Gradient of an embedding table from the gradients of its looked-up rows (the backward of F.embedding, trainer.py:219 / :54-58):
//   table_grad[index[i]] += rows[i]   for i < n,  every other row of table_grad zero (the caller clears it first).
// The looked-up rows come in up to SCATTER_MAX_SEGS segments (round 6: every producer of a step -- the reconstruction loss's positives
// and negatives, Embed's span and word projections -- in ONE launch instead of a scatter + a dense add each); row v of the launch is
// row v - first[s] of segment s.  One workgroup per row; the FIRST occurrence of a token owns its table row and adds the later
// occurrences in ascending v: no atomics on floats, the same bits every run -- torch's index_add_ adds with float atomics in arrival
// order.  All 256 threads scan the index list (a few thousand entries, L2-resident) once: ownership from the entries before v, the list
// of later occurrences (collected with an LDS counter, then sorted: a handful at most) from those behind it.  K a multiple of 4.
constexpr int SCATTER_MAX_SEGS = 4;
constexpr int SCATTER_LIST = 128;
struct ScatterSegs {
    const float* rows[SCATTER_MAX_SEGS];
    const long long* index[SCATTER_MAX_SEGS];
    int first[SCATTER_MAX_SEGS + 1];       // first[s] = rows before segment s; first[nseg] = n
    int nseg;
};
__device__ __forceinline__ int scatter_seg_of(const ScatterSegs& g, int v) {
    int s = 0;
#pragma unroll
    for (int k = 1; k < SCATTER_MAX_SEGS; ++k) s += (k < g.nseg && v >= g.first[k]) ? 1 : 0;
    return s;
}
__device__ __forceinline__ long long scatter_tok(const ScatterSegs& g, int v) {
    const int s = scatter_seg_of(g, v);
    return g.index[s][v - g.first[s]];
}
__device__ __forceinline__ const float* scatter_row(const ScatterSegs& g, int v, int K) {
    const int s = scatter_seg_of(g, v);
    return g.rows[s] + (size_t)(v - g.first[s]) * K;
}
static __global__ __launch_bounds__(256) void rows_scatter_add(ScatterSegs g, int K, float* __restrict__ table_grad, long long V) {
    __shared__ int owner, cnt;
    __shared__ int later[SCATTER_LIST];
    const int v = blockIdx.x, n = g.first[g.nseg];
    const long long tok = scatter_tok(g, v);
    if (tok < 0 || tok >= V) return;
    if (threadIdx.x == 0) { owner = 1; cnt = 0; }
    __syncthreads();
    for (int j = threadIdx.x; j < n; j += 256) {
        if (j == v || scatter_tok(g, j) != tok) continue;
        if (j < v) owner = 0;                      // an earlier occurrence owns the row (benign race: every writer stores 0)
        else {
            const int k = atomicAdd(&cnt, 1);      // integer counter in LDS: the ORDER of the list is fixed by the sort below
            if (k < SCATTER_LIST) later[k] = j;
        }
    }
    __syncthreads();
    if (!owner) return;
    const int m = cnt;
    if (m > SCATTER_LIST) {                        // a token repeated more than SCATTER_LIST times: the plain ordered scan
        for (int c = 4 * threadIdx.x; c < K; c += 1024) {
            float4 acc = ld4(scatter_row(g, v, K) + c);
            for (int j = v + 1; j < n; ++j)
                if (scatter_tok(g, j) == tok) acc = f4add(acc, ld4(scatter_row(g, j, K) + c));
            st4(table_grad + (size_t)tok * K + c, acc);
        }
        return;
    }
    if (threadIdx.x == 0)                          // ascending v: insertion sort of a handful of entries
        for (int a = 1; a < m; ++a) {
            const int x = later[a];
            int b = a - 1;
            for (; b >= 0 && later[b] > x; --b) later[b + 1] = later[b];
            later[b + 1] = x;
        }
    __syncthreads();
    for (int c = 4 * threadIdx.x; c < K; c += 1024) {
        float4 acc = ld4(scatter_row(g, v, K) + c);
        for (int q = 0; q < m; ++q) acc = f4add(acc, ld4(scatter_row(g, later[q], K) + c));
        st4(table_grad + (size_t)tok * K + c, acc);
    }
}

// Trainer.gradient_update (trainer.py:450-455): clip_grad_norm_(params, max_norm) + Adam step over ONE flat parameter / gradient
// buffer: partial sums of g^2 per block -> total in block order (one lane) -> clip coefficient -> the update.
// ---------------------------------------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(256) void sumsq_partial(size_t n, const float* __restrict__ g, float* __restrict__ part) {
    __shared__ float sh[256];
    float t = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) t += g[i] * g[i];
    sh[threadIdx.x] = t;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = sh[0];
}
static __global__ __launch_bounds__(256) void clip_coef(int nparts, const float* __restrict__ part, float max_norm, float* __restrict__ out /* [0] norm, [1] coef */) {
    __shared__ float sh[256];
    float t = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 256) t += part[i];
    sh[threadIdx.x] = t;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {                          // fixed tree: the same total for the same partial sums, run after run
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float norm = sqrtf(sh[0]);
        out[0] = norm;
        out[1] = fminf(1.f, max_norm / (norm + 1e-6f));          // torch.nn.utils.clip_grad_norm_: clamp(max_norm / (norm + 1e-6), max=1)
    }
}
static __global__ __launch_bounds__(256) void adam_step(size_t n, float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                        const float* __restrict__ coef, float lr, float b1, float b2, float eps, float bc1, float bc2) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i] * coef[1];
    g[i] = gi;                                                   // the clipped gradient stays readable, as after clip_grad_norm_
    const float mi = m[i] + (gi - m[i]) * (1.f - b1);              // exp_avg.lerp_(grad, 1 - beta1)
    const float vi = v[i] * b2 + (1.f - b2) * gi * gi;              // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
    m[i] = mi; v[i] = vi;
    // torch.optim.Adam (no amsgrad): p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
    p[i] = p[i] - (lr / bc1) * mi / (sqrtf(vi) / sqrtf(bc2) + eps);
}

}  // namespace cliora
