// C ABI, CLIORA span-region / word-region scorer unit.
#include "api_common.hpp"
#include "vl_kernels.hpp"

// ------------------------------------------------------------------ CLIORA span-region / word-region scorers
extern "C" size_t cliora_plan_vl_workspace_bytes(const cliora_plan* plan) { return plan ? plan->p.vl.total * sizeof(float) : 0; }

// C[i][j] = sum_r A(r,i) B(r,j) with different tile counts per side (register-only split-K kernel)
template <int TI, int TJ, class AP, class BP>
static int launch_tn_ij(hipStream_t st, int nrows, int Mi, int Nj, AP ap, BP bp, float* slab, size_t slab_floats, float* out) {
    const int blocks = (Mi / (TI * 16)) * (Nj / (TJ * 16));
    const size_t per_slice = (size_t)Mi * Nj;
    int nsl = (int)std::min<size_t>(slab_floats / per_slice, (size_t)std::max(1, 2048 / blocks));
    nsl = std::min(nsl, (nrows + 15) / 16);
    nsl = std::max(4, nsl / 4 * 4);
    if ((size_t)nsl * per_slice > slab_floats) return fail(CLIORA_ENOMEM, "slab too small for the region-gradient GEMM");
    int rps = (nrows + nsl - 1) / nsl;
    rps = (rps + 3) / 4 * 4;
    hipLaunchKernelGGL((tn_gemm<TI, TJ, false, AP, BP>), dim3(blocks, nsl / 4), dim3(WS_THREADS), 0, st, nrows, rps, Mi, Nj, ap, bp, slab,
                       (float*)nullptr);
    LAUNCHOK("tn_gemm(ij)");
    launch_slab_reduce(st, slab, nsl, per_slice, out, 0);
    LAUNCHOK("slab_reduce");
    return CLIORA_OK;
}
template <class AP, class BP>
static int launch_tn_regions(hipStream_t st, int nrows, int Mi, int Nj, AP ap, BP bp, float* slab, size_t slab_floats, float* out) {
    const int ti = (Mi / 16) % 4 == 0 ? 4 : ((Mi / 16) % 2 == 0 ? 2 : 1);
    const int tj = pick_tiles(Nj / 16);
#define RG_CASE(a, b) if (ti == a && tj == b) return launch_tn_ij<a, b>(st, nrows, Mi, Nj, ap, bp, slab, slab_floats, out)
    RG_CASE(4, 5); RG_CASE(4, 4); RG_CASE(4, 2); RG_CASE(4, 1);
    RG_CASE(2, 5); RG_CASE(2, 4); RG_CASE(2, 2); RG_CASE(2, 1);
    RG_CASE(1, 5); RG_CASE(1, 4); RG_CASE(1, 2); RG_CASE(1, 1);
#undef RG_CASE
    return fail(CLIORA_EINVAL, "unsupported region-gradient tile shape");
}

// reduction over the (padded) region axis: split it into LDS-sized segments
static void region_segments(int NRp, int ncols, int* Kseg, int* nseg) {
    const int ct = pick_tiles(ncols / 16);
    for (int n = 1; n <= NRp / 16; ++n) {
        if (NRp % n || (NRp / n) % 16) continue;
        if ((size_t)ct * 16 * (NRp / n) * sizeof(float) <= 150 * 1024) { *Kseg = NRp / n; *nseg = n; return; }
    }
    *Kseg = 16; *nseg = NRp / 16;
}

// The span-region scorer GEMM  out[r][c*R + d] = A(r, :) . O[c][d][:]  (cliora.py:457-466) in the arithmetic mode in force: the
// fp32-input MFMA (exact products) or, by default, split-bf16 products on the bf16 MFMA with the region matrix as a split image
// (rows_gemm_ws3; 24.8 GFLOP at c3: 267 us on the fp32 MFMA).  O: (NRp x Dp) region matrix, row stride ldo.
template <class AP, class EP>
static int launch_scorer(hipStream_t st, const Plan& p, void* vl_ws, const float* O, int ldo, int nrows, AP ap, EP ep) {
    const int Dp = p.Dp, NRp = p.vl.NRp;
    if (!split_bf16() || ldo != Dp) return launch_rows(st, O, Dp, 1, NRp, nrows, ap, ep);
    float* img = (float*)vl_ws + p.vl.oimg;
    ImageList im;
    im.add(O, img, NRp, ldo, Dp);
    OKR(build_weight_images(st, im));
    return launch_rows3(st, reinterpret_cast<const uint32_t*>(img), image_stride(Dp), Dp, NRp, nrows, ap, ep);
}

struct VlViews { float *oall, *oallT, *wall, *wallT, *sump, *xwp, *xwn, *dxn, *nrm, *gobj, *slab; };
static VlViews vl_views(const Plan& p, void* ws) {
    float* w = (float*)ws;
    const auto& v = p.vl;
    return VlViews{w + v.oall, w + v.oallT, w + v.wall, w + v.wallT, w + v.sump, w + v.xwp, w + v.xwn, w + v.dxn, w + v.nrm, w + v.gobj, w + v.slab};
}

extern "C" int cliora_vl_scores_forward(cliora_plan* plan, const float* inside_h, const float* outside_h, const float* obj_span,
                                        const float* x_word, const float* obj_word, int training, float* all_atten,
                                        float* vg_atten, void* vl_ws, size_t vl_ws_bytes, void* stream) {
    if (!plan || !inside_h || !outside_h || !obj_span || !vl_ws) return fail(CLIORA_EINVAL, "NULL argument");
    if (!all_atten && !(training && vg_atten)) return fail(CLIORA_EINVAL, "all_atten may only be NULL in training mode with vg_atten given");
    const Plan& p = plan->p;
    if (p.R <= 0) return fail(CLIORA_EINVAL, "the scorers need a CLIORA plan (R > 0)");
    if (vl_ws_bytes < p.vl.total * sizeof(float)) return fail(CLIORA_ENOMEM, "VL workspace too small");
    if (vg_atten && (!x_word || !obj_word)) return fail(CLIORA_EINVAL, "vg_atten needs x_word and obj_word");
    hipStream_t st = (hipStream_t)stream;
    const int B = p.B, L = p.L, D = p.D, Dp = p.Dp, C = p.C, R = p.R, NRp = p.vl.NRp;
    const bool padded = D != Dp;
    const VlViews v = vl_views(p, vl_ws);
    const bool inplace = !padded && B * R == NRp;        // the region matrices are already (NRp x Dp) weights: no packed copies
    {
        CopyTable t; t.n = 0;
        if (all_atten && !inplace) add_copy(t, v.oall, Dp, NRp, Dp, obj_span, D, B * R, D, 0, 0, 0);
        if (vg_atten && !inplace) add_copy(t, v.wall, Dp, NRp, Dp, obj_word, D, B * R, D, 0, 0, 0);
        if (padded && all_atten) add_copy(t, v.sump, Dp, B * C, Dp, inside_h, D, B * C, D, 0, 0, 0, outside_h, D, B * C, D, 0, 0, 0);
        if (vg_atten && (padded || !training)) add_copy(t, v.xwp, Dp, B * L, Dp, x_word, D, B * L, D, 0, 0, 0);
        OKR(run_copies(st, t));
    }
    const SumRowsA sumA = padded ? SumRowsA{v.sump, nullptr, Dp} : SumRowsA{inside_h, outside_h, Dp};
    if (all_atten) OKR(launch_scorer(st, p, vl_ws, inplace ? obj_span : v.oall, Dp, B * C, sumA, ScoreStoreE{all_atten, B, C, R, nullptr, 0}));
    if (vg_atten) {
        if (training) {
            const SumRowsA xw = SumRowsA{padded ? v.xwp : x_word, nullptr, Dp};
            OKR(launch_rows(st, inplace ? obj_word : v.wall, Dp, 1, NRp, B * L, xw, ScoreStoreE{vg_atten, B, L, R, nullptr, 0}));
        } else {   // eval: all_atten[:, :, :L] + unit(x_word) . obj_word   (cliora.py:462-464)
            hipLaunchKernelGGL(unit_norm_rows, dim3(cells_grid(B * L)), dim3(256), 0, st, v.xwp, Dp, B * L, B * L, 0, 0, Dp, p.normalize,
                               v.xwn, v.nrm, v.nrm + B * L);
            LAUNCHOK("unit_norm_rows(x_word)");
            OKR(launch_rows(st, inplace ? obj_word : v.wall, Dp, 1, NRp, B * L, SumRowsA{v.xwn, nullptr, Dp}, ScoreStoreE{vg_atten, B, L, R, all_atten, C}));
        }
    }
    return CLIORA_OK;
}

extern "C" int cliora_vl_scores_max_forward(cliora_plan* plan, const float* inside_h, const float* outside_h, const float* obj_span,
                                            float* all_max, int32_t* all_arg, void* vl_ws, size_t vl_ws_bytes, void* stream) {
    if (!plan || !inside_h || !outside_h || !obj_span || !all_max || !vl_ws) return fail(CLIORA_EINVAL, "NULL argument");
    const Plan& p = plan->p;
    if (p.R <= 0) return fail(CLIORA_EINVAL, "the scorers need a CLIORA plan (R > 0)");
    if (vl_ws_bytes < p.vl.total * sizeof(float)) return fail(CLIORA_ENOMEM, "VL workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int B = p.B, D = p.D, Dp = p.Dp, C = p.C, R = p.R, NRp = p.vl.NRp;
    const bool padded = D != Dp;
    const VlViews v = vl_views(p, vl_ws);
    unsigned long long* keys = reinterpret_cast<unsigned long long*>((float*)vl_ws + p.vl.keys);
    const size_t nkeys = (size_t)B * B * C;
    HIPOK(hipMemsetAsync(keys, 0, nkeys * sizeof(unsigned long long), st));
    const bool inplace = !padded && B * R == NRp;        // the region matrix is already a (NRp x Dp) weight: no packed copy
    if (!inplace) {
        CopyTable t; t.n = 0;
        add_copy(t, v.oall, Dp, NRp, Dp, obj_span, D, B * R, D, 0, 0, 0);
        if (padded) add_copy(t, v.sump, Dp, B * C, Dp, inside_h, D, B * C, D, 0, 0, 0, outside_h, D, B * C, D, 0, 0, 0);
        OKR(run_copies(st, t));
    }
    const SumRowsA sumA = padded ? SumRowsA{v.sump, nullptr, Dp} : SumRowsA{inside_h, outside_h, Dp};
    OKR(launch_scorer(st, p, vl_ws, inplace ? obj_span : v.oall, Dp, B * C, sumA, ScoreMaxE{keys, B, C, R}));
    hipLaunchKernelGGL(region_keys_decode, dim3((unsigned)((nkeys + 255) / 256)), dim3(256), 0, st, keys, nkeys, all_max, all_arg);
    LAUNCHOK("region_keys_decode");
    return CLIORA_OK;
}

extern "C" int cliora_vl_scores_max_backward(cliora_plan* plan, const float* inside_h, const float* outside_h, const float* obj_span,
                                             const float* d_all_max, const int32_t* all_arg, float* d_sum_h, float* d_obj_span,
                                             void* vl_ws, size_t vl_ws_bytes, void* stream) {
    if (!plan || !inside_h || !outside_h || !obj_span || !d_all_max || !all_arg || !vl_ws) return fail(CLIORA_EINVAL, "NULL argument");
    const Plan& p = plan->p;
    if (p.R <= 0) return fail(CLIORA_EINVAL, "the scorers need a CLIORA plan (R > 0)");
    if (vl_ws_bytes < p.vl.total * sizeof(float)) return fail(CLIORA_ENOMEM, "VL workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int B = p.B, D = p.D, Dp = p.Dp, C = p.C, R = p.R, NRp = p.vl.NRp;
    const VlViews v = vl_views(p, vl_ws);
    if (Dp > 512) return fail(CLIORA_EINVAL, "region-max backward: D > 512 is not supported");
    const bool padded = D != Dp;
    if (padded) {   // 16-byte aligned operand rows: the region matrix and S = inside_h + outside_h (D % 16 == 0: read in place)
        CopyTable t; t.n = 0;
        if (d_sum_h) add_copy(t, v.oall, Dp, NRp, Dp, obj_span, D, B * R, D, 0, 0, 0);
        if (d_obj_span) add_copy(t, v.sump, Dp, B * C, Dp, inside_h, D, B * C, D, 0, 0, 0, outside_h, D, B * C, D, 0, 0, 0);
        OKR(run_copies(st, t));
    }
    if (d_sum_h) {
        hipLaunchKernelGGL(region_max_bwd_rows, dim3(B * C), dim3(256), 0, st, B, C, R, D, Dp, d_all_max, all_arg, padded ? v.oall : obj_span, Dp, d_sum_h);
        LAUNCHOK("region_max_bwd_rows");
    }
    if (d_obj_span) {
        // partial sums over chunks of sentences (at most 8: the slab holds 8 region matrices), then the fixed-order sum and the unpad copy
        const int a_per_chunk = std::max(4, (B + 7) / 8), nchunk = (B + a_per_chunk - 1) / a_per_chunk;
        const size_t n = (size_t)B * R * Dp;
        if (!padded) {
            const size_t n4 = (size_t)B * C * Dp / 4;
            hipLaunchKernelGGL(add_rows4, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, reinterpret_cast<const float4*>(inside_h),
                               reinterpret_cast<const float4*>(outside_h), n4, reinterpret_cast<float4*>(v.sump));
            LAUNCHOK("add_rows4");
        }
        hipLaunchKernelGGL(region_max_bwd_obj, dim3(B * R, nchunk), dim3(256), 0, st, B, C, R, Dp, a_per_chunk, d_all_max, all_arg, v.sump, v.slab);
        LAUNCHOK("region_max_bwd_obj");
        launch_slab_reduce(st, v.slab, nchunk, n, v.gobj, 0);
        LAUNCHOK("slab_reduce");
        CopyTable t; t.n = 0;
        add_copy(t, d_obj_span, D, B * R, D, v.gobj, Dp, B * R, D, 0, 0, 0);
        OKR(run_copies(st, t));
    }
    return CLIORA_OK;
}

extern "C" int cliora_vl_scores_backward(cliora_plan* plan, const float* inside_h, const float* outside_h, const float* obj_span,
                                         const float* x_word, const float* obj_word, int training, const float* d_all,
                                         const float* d_vg, float* d_sum_h, float* d_obj_span, float* d_x_word, float* d_obj_word,
                                         void* vl_ws, size_t vl_ws_bytes, void* stream) {
    if (!plan || !inside_h || !outside_h || !obj_span || !vl_ws) return fail(CLIORA_EINVAL, "NULL argument");
    const Plan& p = plan->p;
    if (p.R <= 0) return fail(CLIORA_EINVAL, "the scorers need a CLIORA plan (R > 0)");
    if (vl_ws_bytes < p.vl.total * sizeof(float)) return fail(CLIORA_ENOMEM, "VL workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int B = p.B, L = p.L, D = p.D, Dp = p.Dp, C = p.C, R = p.R, NRp = p.vl.NRp;
    const bool padded = D != Dp;
    const VlViews v = vl_views(p, vl_ws);
    const bool eval_vg = !training && d_vg;      // eval: vg = all_atten[:, :, :L] + unit(x_word) . obj_word  (cliora.py:462-464)
    {   // transposed region matrices: W[j][k] = O[k][j], k padded with zero rows
        CopyTable t; t.n = 0;
        const bool all_live = d_all || eval_vg;              // no cotangent reaches all_atten (training with the region-max scorer): skip its operands
        if (all_live && d_sum_h) add_copy(t, v.oallT, NRp, Dp, NRp, obj_span, D, B * R, D, 0, 0, 1);
        if (d_vg) add_copy(t, v.wallT, NRp, Dp, NRp, obj_word, D, B * R, D, 0, 0, 1);
        if (padded && all_live) add_copy(t, v.sump, Dp, B * C, Dp, inside_h, D, B * C, D, 0, 0, 0, outside_h, D, B * C, D, 0, 0, 0);
        if (d_vg && (padded || eval_vg)) add_copy(t, v.xwp, Dp, B * L, Dp, x_word, D, B * L, D, 0, 0, 0);
        OKR(run_copies(st, t));
    }
    if (eval_vg) {
        hipLaunchKernelGGL(unit_norm_rows, dim3(cells_grid(B * L)), dim3(256), 0, st, v.xwp, Dp, B * L, B * L, 0, 0, Dp, p.normalize, v.xwn,
                           v.nrm, v.nrm + B * L);
        LAUNCHOK("unit_norm_rows(x_word)");
    }
    const SumRowsA sumA = padded ? SumRowsA{v.sump, nullptr, Dp} : SumRowsA{inside_h, outside_h, Dp};
    // gradient reaching all_atten: its own cotangent, plus (eval) the vg cotangent on the leaf cells
    const ScoreGrad2A gall{d_all, eval_vg ? d_vg : nullptr, B, C, L, R};
    const bool any_all = d_all || eval_vg;
    int Kseg = NRp, nseg = 1;
    region_segments(NRp, Dp, &Kseg, &nseg);
    if (d_sum_h) {
        if (any_all) OKR(launch_rows(st, v.oallT, Kseg, nseg, Dp, B * C, gall, StoreAccE{d_sum_h, D, D, 0}));
        else HIPOK(hipMemsetAsync(d_sum_h, 0, (size_t)B * C * D * sizeof(float), st));
    }
    if (d_obj_span) {
        if (any_all) {
            OKR(launch_tn_regions(st, B * C, NRp, Dp, gall, sumA, v.slab, p.vl.slab_floats, v.gobj));
            CopyTable t; t.n = 0;
            add_copy(t, d_obj_span, D, B * R, D, v.gobj, Dp, B * R, D, 0, 0, 0);
            OKR(run_copies(st, t));
        } else HIPOK(hipMemsetAsync(d_obj_span, 0, (size_t)B * R * D * sizeof(float), st));
    }
    if (d_x_word) {
        if (d_vg && training) OKR(launch_rows(st, v.wallT, Kseg, nseg, Dp, B * L, ScoreGradA{d_vg, B, L, R}, StoreAccE{d_x_word, D, D, 0}));
        else if (d_vg) {      // through unit(x_word)
            OKR(launch_rows(st, v.wallT, Kseg, nseg, Dp, B * L, ScoreGradA{d_vg, B, L, R}, StoreAccE{v.dxn, Dp, Dp, 0}));
            hipLaunchKernelGGL(rows_unit_bwd, dim3(cells_grid(B * L)), dim3(256), 0, st, B * L, Dp, D, v.dxn, v.xwn, v.nrm, p.normalize, d_x_word);
            LAUNCHOK("rows_unit_bwd");
        } else HIPOK(hipMemsetAsync(d_x_word, 0, (size_t)B * L * D * sizeof(float), st));
    }
    if (d_obj_word) {
        if (d_vg) {
            const SumRowsA xw = eval_vg ? SumRowsA{v.xwn, nullptr, Dp} : SumRowsA{padded ? v.xwp : x_word, nullptr, Dp};
            OKR(launch_tn_regions(st, B * L, NRp, Dp, ScoreGradA{d_vg, B, L, R}, xw, v.slab, p.vl.slab_floats, v.gobj));
            CopyTable t; t.n = 0;
            add_copy(t, d_obj_word, D, B * R, D, v.gobj, Dp, B * R, D, 0, 0, 0);
            OKR(run_copies(st, t));
        } else HIPOK(hipMemsetAsync(d_obj_word, 0, (size_t)B * R * D * sizeof(float), st));
    }
    return CLIORA_OK;
}

// ------------------------------------------------------------------ ContrastiveLoss on the region maxima
extern "C" size_t cliora_contrastive_workspace_bytes(int B, int C) { return ((size_t)(C / 2) * (B + 1) + 64) * sizeof(float); }

extern "C" int cliora_contrastive_loss(int B, int C, const float* all_max, const float* inside_s, const float* outside_s, float margin,
                                       float alpha, float* loss, float* d_all_max, float* d_inside_s, float* d_outside_s, void* ws,
                                       size_t ws_bytes, void* stream) {
    if (!all_max || !inside_s || !outside_s || !loss || !d_all_max || !d_inside_s || !d_outside_s || !ws) return fail(CLIORA_EINVAL, "NULL argument");
    if (B < 1 || B > 128 || C < 1) return fail(CLIORA_EINVAL, "contrastive loss kernel: 1 <= B <= 128");
    if (ws_bytes < cliora_contrastive_workspace_bytes(B, C)) return fail(CLIORA_ENOMEM, "contrastive workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int nb = C / 2;                              // trainer.py:126: only the first span_length // 2 spans enter the loss
    float* part = (float*)ws;
    float* last = part + ((nb + 63) / 64) * 64;
    HIPOK(hipMemsetAsync(d_all_max, 0, (size_t)B * B * C * sizeof(float), st));
    HIPOK(hipMemsetAsync(d_inside_s, 0, (size_t)B * C * sizeof(float), st));
    HIPOK(hipMemsetAsync(d_outside_s, 0, (size_t)B * C * sizeof(float), st));
    const float k = alpha / (float)B;
    if (nb > 0) {
        const size_t lds = ((size_t)B * (B + 1) + 4 * B + 256) * sizeof(float);
        OKR(cliora_ensure_max_lds((const void*)contrastive_spans));
        hipLaunchKernelGGL(contrastive_spans, dim3(nb), dim3(256), lds, st, B, C, all_max, inside_s, outside_s, margin, k, part, last,
                           d_all_max, d_inside_s, d_outside_s);
        LAUNCHOK("contrastive_spans");
    }
    hipLaunchKernelGGL(contrastive_finish, dim3(1), dim3(256), 0, st, B, C, nb, part, last, k, loss, d_inside_s);
    LAUNCHOK("contrastive_finish");
    return CLIORA_OK;
}
