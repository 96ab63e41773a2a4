// C ABI, heads unit: the callers either side of the chart path as hand-written kernels:
// Embed / ImageEncoder projections, ReconstructionSoftmaxLoss, VGLoss and the clip + Adam update.
// See include/cliora_chart.h for the contract and the reference lines each entry point replaces.
#include "api_common.hpp"
#include "head_kernels.hpp"

static inline size_t al64(size_t x) { return (x + 63) & ~size_t(63); }
static inline int ceil16(int x) { return (x + 15) / 16 * 16; }

// C[i][j] (+ colsum[i]) = sum_r A(r,i) B(r,j) for a rectangular C (Mi x Nj, both multiples of 16): tile counts per dimension
template <int TI, int TJ, class AP, class BP>
static int launch_tn_rect_t(hipStream_t st, int nrows, int Mi, int Nj, AP ap, BP bp, float* slab, size_t slab_floats, float* out, float* colsum_out) {
    const int blocks = (Mi / (TI * 16)) * (Nj / (TJ * 16));
    const size_t per_slice = (size_t)Mi * Nj + (colsum_out ? Mi : 0);
    int nsl = (int)std::min<size_t>(slab_floats / per_slice, (size_t)std::max(4, 2048 / std::max(1, blocks)));
    nsl = std::min(nsl, (nrows + 31) / 32);
    nsl = std::max(4, nsl / 4 * 4);
    if ((size_t)nsl * per_slice > slab_floats) return fail(CLIORA_ENOMEM, "workspace too small for the weight-gradient GEMM");
    int rps = (nrows + nsl - 1) / nsl;
    rps = (rps + 3) / 4 * 4;
    float* csl = slab + (size_t)nsl * Mi * Nj;
    if (colsum_out) hipLaunchKernelGGL((tn_gemm<TI, TJ, true, AP, BP>), dim3(blocks, nsl / 4), dim3(WS_THREADS), 0, st, nrows, rps, Mi, Nj, ap, bp, slab, csl);
    else hipLaunchKernelGGL((tn_gemm<TI, TJ, false, AP, BP>), dim3(blocks, nsl / 4), dim3(WS_THREADS), 0, st, nrows, rps, Mi, Nj, ap, bp, slab, csl);
    LAUNCHOK("tn_gemm(rect)");
    const size_t n = (size_t)Mi * Nj;
    launch_slab_reduce(st, slab, nsl, n, out, 0, csl, (size_t)Mi, colsum_out);
    LAUNCHOK("slab_reduce");
    return CLIORA_OK;
}
template <class AP, class BP>
static int launch_tn_rect(hipStream_t st, int nrows, int Mi, int Nj, AP ap, BP bp, float* slab, size_t slab_floats, float* out, float* colsum_out) {
    const int ti = (Mi / 16) % 5 == 0 ? 5 : ((Mi / 16) % 4 == 0 ? 4 : 1);
    const int tj = (Nj / 16) % 4 == 0 ? 4 : 1;
#define TNR(a, b) return launch_tn_rect_t<a, b>(st, nrows, Mi, Nj, ap, bp, slab, slab_floats, out, colsum_out)
    if (ti == 5 && tj == 4) TNR(5, 4);
    if (ti == 5) TNR(5, 1);
    if (ti == 4 && tj == 4) TNR(4, 4);
    if (ti == 4) TNR(4, 1);
    if (tj == 4) TNR(1, 4);
    TNR(1, 1);
#undef TNR
}
// slab floats for a (Mi x Nj) product over nrows rows: at most 16 slices, at most 64 MB
static size_t tn_rect_slab_floats(int nrows, int Mi, int Nj) {
    const size_t per = (size_t)Mi * Nj + Mi;
    size_t nsl = std::min<size_t>(16, std::max<size_t>(4, ((size_t)nrows + 31) / 32 / 4 * 4));
    while (nsl > 4 && nsl * per * 4 > ((size_t)64 << 20)) nsl -= 4;
    return nsl * per;
}

// ------------------------------------------------------------------ VGLoss
extern "C" size_t cliora_vg_workspace_bytes(int B, int L) {
    return (al64((size_t)B * B) * 2 + al64((size_t)B) + al64((size_t)B * B * L)) * sizeof(float);
}
extern "C" int cliora_vg_loss(int B, int L, int R, const float* vg_atten, float alpha, float* loss, float* d_vg_atten, void* ws, size_t ws_bytes, void* stream) {
    if (!vg_atten || !loss || !ws || B < 1 || L < 1 || R < 1) return fail(CLIORA_EINVAL, "bad argument");
    if (ws_bytes < cliora_vg_workspace_bytes(B, L)) return fail(CLIORA_ENOMEM, "workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float* w = (float*)ws;
    float* logits = w; float* dlog = w + al64((size_t)B * B); float* row_loss = dlog + al64((size_t)B * B);
    int32_t* arg = reinterpret_cast<int32_t*>(row_loss + al64((size_t)B));
    hipLaunchKernelGGL(vg_logits_fwd, dim3((B * B + 3) / 4), dim3(256), 0, st, B, L, R, vg_atten, logits, arg);
    LAUNCHOK("vg_logits_fwd");
    hipLaunchKernelGGL(vg_ce, dim3(1), dim3(1024), 0, st, B, alpha, logits, row_loss, dlog, loss);
    LAUNCHOK("vg_ce");
    if (d_vg_atten) {
        const size_t n = (size_t)B * B * L * R;
        hipLaunchKernelGGL(vg_scatter_bwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, L, R, dlog, arg, d_vg_atten);
        LAUNCHOK("vg_scatter_bwd");
    }
    return CLIORA_OK;
}

// ------------------------------------------------------------------ y = gather(x, index) W^T + bias
struct ProjLayout { size_t wp, bp, wT, gw, slab, slab_floats, total; int Dp; };
static ProjLayout proj_layout(int nrows, int K, int D) {
    ProjLayout l; size_t o = 0;
    auto take = [&](size_t n) { size_t at = o; o = al64(o + n); return at; };
    l.Dp = ceil16(D);
    l.wp = take((size_t)l.Dp * K); l.bp = take(l.Dp);
    l.wT = take((size_t)K * l.Dp);
    l.gw = take((size_t)l.Dp * K + l.Dp);
    l.slab_floats = tn_rect_slab_floats(nrows, l.Dp, K);
    l.slab = take(l.slab_floats);
    l.total = o;
    return l;
}
extern "C" size_t cliora_proj_workspace_bytes(int nrows, int K, int D) { return proj_layout(nrows, K, D).total * sizeof(float); }

extern "C" int cliora_proj_forward(const float* x, const int64_t* index, int nrows, int K, const float* w, const float* bias, int D, float* y, void* ws,
                                   size_t ws_bytes, void* stream) {
    if (!x || !w || !y || !ws || nrows < 1 || D < 1) return fail(CLIORA_EINVAL, "bad argument");
    if (K < 16 || K % 16) return fail(CLIORA_EINVAL, "the input width must be a multiple of 16");
    const ProjLayout l = proj_layout(nrows, K, D);
    if (ws_bytes < l.total * sizeof(float)) return fail(CLIORA_ENOMEM, "workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float* wsf = (float*)ws;
    const float* W = w; const float* Bv = bias;
    if (D != l.Dp) {        // rows D .. Dp-1 of the weight read as zero
        CopyTable t; t.n = 0;
        add_copy(t, wsf + l.wp, K, l.Dp, K, w, K, D, K, 0, 0, 0);
        if (bias) add_copy(t, wsf + l.bp, l.Dp, 1, l.Dp, bias, D, 1, D, 0, 0, 0);
        OKR(run_copies(st, t));
        W = wsf + l.wp; Bv = bias ? wsf + l.bp : nullptr;
    }
    return launch_rows_direct(st, W, nullptr, IMG_NONE, K, l.Dp, nrows, GatherRowsA{x, K, index}, StoreRowsE{y, D, Bv, 0, D});
}

extern "C" int cliora_proj_backward(const float* x, const int64_t* index, int nrows, int K, const float* w, const float* d_y, int D, float* d_w,
                                    float* d_bias, float* d_rows, void* ws, size_t ws_bytes, void* stream) {
    if (!x || !w || !d_y || !ws || nrows < 1 || D < 1) return fail(CLIORA_EINVAL, "bad argument");
    if (K < 16 || K % 16) return fail(CLIORA_EINVAL, "the input width must be a multiple of 16");
    const ProjLayout l = proj_layout(nrows, K, D);
    if (ws_bytes < l.total * sizeof(float)) return fail(CLIORA_ENOMEM, "workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float* wsf = (float*)ws;
    const int Dp = l.Dp;
    if (d_w || d_bias) {     // dW = dY^T X_gathered, dbias = column sums of dY
        float* gw = (D == Dp && d_w) ? d_w : wsf + l.gw;
        float* gb = d_bias ? ((D == Dp) ? d_bias : wsf + l.gw + (size_t)Dp * K) : nullptr;
        OKR(launch_tn_rect(st, nrows, Dp, K, BoundedRowsA{d_y, D, D}, GatherRowsA{x, K, index}, wsf + l.slab, l.slab_floats, gw, gb));
        if (D != Dp) {
            CopyTable t; t.n = 0;
            if (d_w) add_copy(t, d_w, K, D, K, gw, K, D, K, 0, 0, 0);
            if (d_bias) add_copy(t, d_bias, D, 1, D, gb, Dp, 1, D, 0, 0, 0);
            OKR(run_copies(st, t));
        }
    }
    if (d_rows) {            // dX_gathered = dY W: the reduction runs over D (padded to Dp with zeros on both sides)
        CopyTable t; t.n = 0;
        add_copy(t, wsf + l.wT, Dp, K, Dp, w, K, D, K, 0, 0, 1);
        OKR(run_copies(st, t));
        OKR(launch_rows_direct(st, wsf + l.wT, nullptr, IMG_NONE, Dp, K, nrows, BoundedRowsA{d_y, D, D}, StoreRowsE{d_rows, K, nullptr, 0, K}));
    }
    return CLIORA_OK;
}

// ------------------------------------------------------------------ ReconstructionSoftmaxLoss
struct ReconLayout { size_t idx, P, XN, gxp, row_loss, gs, gxs, PnT, dP, matT, gm, slab, slab_floats, total; int Dp, Knp, NR; };
static ReconLayout recon_layout(int N, int Kn, int E, int D) {
    ReconLayout l; size_t o = 0;
    auto take = [&](size_t n) { size_t at = o; o = al64(o + n); return at; };
    l.Dp = ceil16(D); l.Knp = ceil16(Kn); l.NR = N + l.Knp;            // projection rows: N positives, Kn negatives, zero rows up to Knp
    l.idx = take((size_t)2 * (N + Kn));                                // int64
    l.P = take((size_t)l.NR * l.Dp);
    l.XN = take((size_t)N * l.Knp); l.gxp = take(N); l.row_loss = take(N);
    l.gs = take((size_t)N * l.Knp); l.gxs = take(N);                   // the same, scaled by the upstream cotangent
    l.PnT = take((size_t)l.Dp * l.Knp);
    l.dP = take((size_t)l.NR * l.Dp);
    l.matT = take((size_t)E * l.Dp);
    l.gm = take((size_t)l.Dp * E);
    l.slab_floats = std::max(tn_rect_slab_floats(N + Kn, l.Dp, E), tn_rect_slab_floats(N, l.Knp, l.Dp));
    l.slab = take(l.slab_floats);
    l.total = o;
    return l;
}
extern "C" size_t cliora_recon_workspace_bytes(int N, int Kn, int E, int D) { return recon_layout(N, Kn, E, D).total * sizeof(float); }

extern "C" int cliora_recon_forward(const int64_t* tokens, const int64_t* neg, int B, int L, int C, int Kn, const float* emb, int E, const float* mat, int D,
                                    const float* outside_h, float* loss, void* ws, size_t ws_bytes, void* stream) {
    if (!tokens || !neg || !emb || !mat || !outside_h || !loss || !ws || B < 1 || L < 1 || Kn < 1) return fail(CLIORA_EINVAL, "bad argument");
    if (E < 16 || E % 16) return fail(CLIORA_EINVAL, "the embedding width must be a multiple of 16");
    const int N = B * L;
    const ReconLayout l = recon_layout(N, Kn, E, D);
    if (ws_bytes < l.total * sizeof(float)) return fail(CLIORA_ENOMEM, "workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float* w = (float*)ws;
    int64_t* idx = reinterpret_cast<int64_t*>(w + l.idx);
    const int Dp = l.Dp, Knp = l.Knp;
    hipLaunchKernelGGL(concat_index, dim3((N + Kn + 255) / 256), dim3(256), 0, st, N, tokens, Kn, neg, idx);
    LAUNCHOK("concat_index");
    HIPOK(hipMemsetAsync(w + l.P, 0, (size_t)l.NR * Dp * sizeof(float), st));
    // P = E[tokens; negatives] mat^T   (trainer.py:59-60); the weight's rows D .. Dp-1 read as zero through a padded copy
    const float* W = mat;
    if (D != Dp) {
        CopyTable t; t.n = 0;
        add_copy(t, w + l.gm, E, Dp, E, mat, E, D, E, 0, 0, 0);
        OKR(run_copies(st, t));
        W = w + l.gm;
    }
    OKR(launch_rows_direct(st, W, nullptr, IMG_NONE, E, Dp, N + Kn, GatherRowsA{emb, E, idx}, StoreRowsE{w + l.P, Dp, nullptr, 0, D}));
    // xn = cell P_neg^T   (trainer.py:67)
    OKR(launch_rows_direct(st, w + l.P + (size_t)N * Dp, nullptr, IMG_NONE, Dp, Knp, N, LeafRowsA{outside_h, D, D, L, C},
                           StoreRowsE{w + l.XN, Knp, nullptr, 0, Knp}));
    hipLaunchKernelGGL(recon_ce, dim3((N + 3) / 4), dim3(256), 0, st, N, Kn, Knp, D, L, C, w + l.P, Dp, outside_h, w + l.XN, w + l.gxp, w + l.row_loss);
    LAUNCHOK("recon_ce");
    hipLaunchKernelGGL(mean_in_order, dim3(1), dim3(256), 0, st, N, w + l.row_loss, loss);
    LAUNCHOK("mean_in_order");
    return CLIORA_OK;
}

// Backward of the call above on the SAME workspace.  gscale: the upstream cotangent (device scalar).  d_cell (N, D): gradient of
// outside_h[:, :L]; d_mat (D, E); d_rows (N + Kn, E): gradient of the looked-up embedding rows (tokens, then negatives), or NULL.
extern "C" int cliora_recon_backward(const int64_t* tokens, const int64_t* neg, int B, int L, int C, int Kn, const float* emb, int E, const float* mat, int D,
                                     const float* outside_h, const float* gscale, float* d_cell, float* d_mat, float* d_rows, void* ws, size_t ws_bytes,
                                     void* stream) {
    (void)tokens; (void)neg;
    if (!emb || !mat || !outside_h || !gscale || !ws) return fail(CLIORA_EINVAL, "NULL argument");
    const int N = B * L;
    const ReconLayout l = recon_layout(N, Kn, E, D);
    if (ws_bytes < l.total * sizeof(float)) return fail(CLIORA_ENOMEM, "workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float* w = (float*)ws;
    const int64_t* idx = reinterpret_cast<const int64_t*>(w + l.idx);
    const int Dp = l.Dp, Knp = l.Knp;
    const size_t ng = (size_t)N * Knp;
    hipLaunchKernelGGL(scale_by_scalar, dim3((unsigned)((ng + 255) / 256)), dim3(256), 0, st, ng, w + l.XN, gscale, w + l.gs);
    hipLaunchKernelGGL(scale_by_scalar, dim3((N + 255) / 256), dim3(256), 0, st, (size_t)N, w + l.gxp, gscale, w + l.gxs);
    LAUNCHOK("scale_by_scalar");
    const float* Ppos = w + l.P;
    const float* Pneg = w + l.P + (size_t)N * Dp;
    if (d_cell) {            // d cell = G P_neg + gxp * P_pos
        CopyTable t; t.n = 0;
        add_copy(t, w + l.PnT, Knp, Dp, Knp, Pneg, Dp, Knp, Dp, 0, 0, 1);
        OKR(run_copies(st, t));
        OKR(launch_rows_direct(st, w + l.PnT, nullptr, IMG_NONE, Knp, Dp, N, PlainRowsA{w + l.gs, Knp},
                               StoreAxpyRowsE{d_cell, D, w + l.gxs, Ppos, Dp, D}));
    }
    if (d_mat || d_rows) {
        // dP: positives gxp * cell, negatives G^T cell
        hipLaunchKernelGGL(recon_dpos, dim3(N), dim3(256), 0, st, N, D, Dp, L, C, w + l.gxs, outside_h, w + l.dP);
        LAUNCHOK("recon_dpos");
        OKR(launch_tn_rect(st, N, Knp, Dp, PlainRowsA{w + l.gs, Knp}, LeafRowsA{outside_h, D, D, L, C}, w + l.slab, l.slab_floats,
                           w + l.dP + (size_t)N * Dp, (float*)nullptr));
    }
    if (d_mat) {             // d mat = dP^T E[tokens; negatives]
        float* gm = D == Dp ? d_mat : w + l.gm;
        OKR(launch_tn_rect(st, N + Kn, Dp, E, PlainRowsA{w + l.dP, Dp}, GatherRowsA{emb, E, idx}, w + l.slab, l.slab_floats, gm, (float*)nullptr));
        if (D != Dp) {
            CopyTable t; t.n = 0;
            add_copy(t, d_mat, E, D, E, gm, E, D, E, 0, 0, 0);
            OKR(run_copies(st, t));
        }
    }
    if (d_rows) {            // d E rows = dP mat
        CopyTable t; t.n = 0;
        add_copy(t, w + l.matT, Dp, E, Dp, mat, E, D, E, 0, 0, 1);
        OKR(run_copies(st, t));
        OKR(launch_rows_direct(st, w + l.matT, nullptr, IMG_NONE, Dp, E, N + Kn, PlainRowsA{w + l.dP, Dp}, StoreRowsE{d_rows, E, nullptr, 0, E}));
    }
    return CLIORA_OK;
}

// ------------------------------------------------------------------ clip_grad_norm_ + Adam on one flat buffer
// table_grad (V, K) = zeros, then table_grad[index[i]] += rows[i]: the scatter F.embedding's backward does (the last ATen op of the
// training step in round 3: zeros_like + index_add_), deterministic
extern "C" int cliora_rows_scatter_add_segments(const float* const* rows, const int64_t* const* index, const int* n, int nseg, int K, float* table_grad,
                                               int64_t V, void* stream) {
    if (!rows || !index || !n || !table_grad) return fail(CLIORA_EINVAL, "NULL argument");
    if (nseg < 0 || nseg > SCATTER_MAX_SEGS) return fail(CLIORA_EINVAL, "rows_scatter_add: at most " + std::to_string(SCATTER_MAX_SEGS) + " segments");
    if (K <= 0 || K % 4 != 0 || V <= 0) return fail(CLIORA_EINVAL, "rows_scatter_add: K a positive multiple of 4, V > 0");
    hipStream_t st = (hipStream_t)stream;
    ScatterSegs g{};
    long long total = 0;
    for (int s = 0; s < nseg; ++s) {
        if (n[s] < 0 || (n[s] > 0 && (!rows[s] || !index[s]))) return fail(CLIORA_EINVAL, "rows_scatter_add: bad segment");
        if (n[s] == 0) continue;                       // empty segments are dropped
        g.rows[g.nseg] = rows[s]; g.index[g.nseg] = reinterpret_cast<const long long*>(index[s]); g.first[g.nseg] = (int)total;
        total += n[s]; ++g.nseg;
    }
    if (total > 0x7fffffffLL) return fail(CLIORA_EINVAL, "rows_scatter_add: too many rows");
    g.first[g.nseg] = (int)total;
    HIPOK(hipMemsetAsync(table_grad, 0, (size_t)V * K * sizeof(float), st));
    if (total == 0) return CLIORA_OK;
    hipLaunchKernelGGL(rows_scatter_add, dim3((unsigned)total), dim3(256), 0, st, g, K, table_grad, (long long)V);
    LAUNCHOK("rows_scatter_add");
    return CLIORA_OK;
}
extern "C" int cliora_rows_scatter_add(const float* rows, const int64_t* index, int n, int K, float* table_grad, int64_t V, void* stream) {
    if (!rows || !index || !table_grad) return fail(CLIORA_EINVAL, "NULL argument");
    if (n < 0) return fail(CLIORA_EINVAL, "rows_scatter_add: n >= 0");
    return cliora_rows_scatter_add_segments(&rows, &index, &n, 1, K, table_grad, V, stream);
}

extern "C" size_t cliora_clip_adam_workspace_bytes(void) { return (1024 + 64) * sizeof(float); }
extern "C" int cliora_clip_adam(float* params, float* grads, float* exp_avg, float* exp_avg_sq, size_t n, float max_norm, float lr, float beta1, float beta2,
                                float eps, int step, void* ws, size_t ws_bytes, void* stream) {
    if (!params || !grads || !exp_avg || !exp_avg_sq || !ws || n == 0 || step < 1) return fail(CLIORA_EINVAL, "bad argument");
    if (ws_bytes < cliora_clip_adam_workspace_bytes()) return fail(CLIORA_ENOMEM, "workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float* w = (float*)ws;
    const int nparts = (int)std::min<size_t>(1024, (n + 255) / 256);
    hipLaunchKernelGGL(sumsq_partial, dim3(nparts), dim3(256), 0, st, n, grads, w);
    hipLaunchKernelGGL(clip_coef, dim3(1), dim3(256), 0, st, nparts, w, max_norm, w + 1024);
    const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
    hipLaunchKernelGGL(adam_step, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, params, grads, exp_avg, exp_avg_sq, w + 1024, lr, beta1, beta2, eps,
                       bc1, bc2);
    LAUNCHOK("clip_adam");
    return CLIORA_OK;
}
