// C ABI, DioraTreeLSTM unit (parity unpinned, see lstm_kernels.hpp).
#include "api_common.hpp"
#include "lstm_kernels.hpp"

#ifndef LSTM_FW
#define LSTM_FW 4         // floats per thread in the cell-centric forward kernel
#endif
#ifndef LSTM_W
#define LSTM_W 4          // floats per thread in the cell-centric backward kernels (lstm_kernels.hpp: VecW); 1, 2 and 4 time the same on MI355X
#endif

// The gate projections (10 of the 11 / 15 of the 17 blocks per inside cell, all 5 per outside cell) and the projection backward on
// split-bf16 MFMA (gemm_kernels.hpp: rows_gemm_ksplit3) in the default arithmetic mode -- their outputs feed sigmoid / tanh, which the
// fixtures and the oracle hold to 1e-4; the score blocks (QL = mat^T h) stay on exact fp32 products.  CLIORA_LSTM_PROJ=f32: the fp32 kernels.
static bool lstm_proj_split() {
    static const bool off = [] { const char* e = getenv("CLIORA_LSTM_PROJ"); return e && !strcmp(e, "f32"); }();
    return !off && split_bf16();
}

// two streams pay under the same rule as for DioraMLP (api_mlp.hip: wavefront_pays)
static bool wavefront_pays_lstm(const Plan& p, int mode) {
    if (p.L <= 2 || mode == 0) return false;
    if (mode >= 1) return true;      // (2 = merged: DioraMLP only; here two streams)
    return (double)(p.R_in + p.R_out) / (2.0 * (p.L - 1)) * p.Dp >= 100e3;
}

// ------------------------------------------------------------------ DioraTreeLSTM (parity unpinned, see lstm_kernels.hpp)
extern "C" int cliora_lstm_forward(cliora_plan* plan, const cliora_params* P, const float* x_span, float* inside_h, float* inside_c,
                                   float* inside_s, float* outside_h, float* outside_c, float* outside_s, void* fwd_ws,
                                   size_t fwd_ws_bytes, int run_outside, void* stream) {
    if (!plan || !P || !x_span || !inside_h || !inside_c || !inside_s || !outside_h || !outside_c || !outside_s || !fwd_ws)
        return fail(CLIORA_EINVAL, "NULL argument");
    const Plan& p = plan->p;
    if (p.arch != 1) return fail(CLIORA_EINVAL, "not a TreeLSTM plan (create it with cliora_plan_create_ex(..., arch = 1))");
    if (!P->lstm_w || !P->lstm_u || !P->lstm_b || !P->in_mat || !P->root_h || !P->root_c) return fail(CLIORA_EINVAL, "missing TreeLSTM parameter");
    if (!p.share && (!P->lstm_u_out || !P->lstm_b_out || !P->out_mat))
        return fail(CLIORA_EINVAL, "share = 0: missing outside TreeLSTM parameter (lstm_u_out, lstm_b_out, out_mat)");
    if (fwd_ws_bytes < p.fwd.total * sizeof(float)) return fail(CLIORA_ENOMEM, "forward workspace too small");
    hipStream_t st = (hipStream_t)stream;
    OKR(cliora_plan_ready(plan, st));
    std::lock_guard<std::mutex> lanes_lock(*plan->lanes_mu);
    ForkGuard fork_guard(st);
    const Dev dv = dev_views(p);
    float* ws = (float*)fwd_ws;
    const FwdLayout& f = p.fwd;
    const int B = p.B, L = p.L, D = p.D, Dp = p.Dp, C = p.C, ldpi = p.nblk * Dp, ldpo = 5 * Dp;
    const size_t DD = (size_t)Dp * Dp;
    const bool padded = D != Dp;
    // outside functions: the inside ones (diora.py:459-461) or their own U, B, mat (share = 0)
    const float* Uo = p.share ? P->lstm_u : P->lstm_u_out;
    float *IH = padded ? ws + f.ihp : inside_h, *OH = padded ? ws + f.ohp : outside_h;
    float *IC = padded ? ws + f.icp : inside_c, *OC = padded ? ws + f.ocp : outside_c;
    float *IS = inside_s, *OS = outside_s;
    const float* X = padded ? ws + f.xp : x_span;
    {
        CopyTable t; t.n = 0;
        for (int g = 0; g < 3; ++g) {
            add_copy(t, ws + f.wl + g * DD, Dp, Dp, Dp, P->lstm_w, D, D, D, g * D, 0, 0);
            add_copy(t, ws + f.wlT + g * Dp, 3 * Dp, Dp, Dp, P->lstm_w, D, D, D, g * D, 0, 1);
            add_copy(t, ws + f.bl + g * Dp, Dp, 1, Dp, P->lstm_b, 5 * D, 1, D, 0, g * D, 0);
        }
        for (int g = 0; g < 5; ++g) {
            add_copy(t, ws + f.wcat + g * DD, Dp, Dp, Dp, P->lstm_u, 2 * D, D, D, g * D, 0, 0);          // PL gate g: U[:, :D]
            add_copy(t, ws + f.wcat + (5 + g) * DD, Dp, Dp, Dp, P->lstm_u, 2 * D, D, D, g * D, D, 0);    // PR gate g: U[:, D:]
            add_copy(t, ws + f.bcat + g * Dp, Dp, 1, Dp, P->lstm_b, 5 * D, 1, D, 0, g * D, 0);
        }
        add_copy(t, ws + f.wcat + 10 * DD, Dp, Dp, Dp, P->in_mat, D, D, D, 0, 0, 1);
        add_copy(t, ws + f.bcat + 5 * Dp, Dp, 1, 6 * Dp, nullptr, 0, 0, 0, 0, 0, 0);
        if (!p.share) add_copy(t, ws + f.bcat + 16 * Dp, Dp, 1, Dp, nullptr, 0, 0, 0, 0, 0, 0);
        add_copy(t, ws + f.rootp, Dp, 1, Dp, P->root_h, D, 1, D, 0, 0, 0);
        add_copy(t, ws + f.rootc, Dp, 1, Dp, P->root_c, D, 1, D, 0, 0, 0);
        if (padded) add_copy(t, ws + f.xp, Dp, B * L, Dp, x_span, D, B * L, D, 0, 0, 0);
        OKR(run_copies(st, t));
        CopyTable u; u.n = 0;
        for (int g = 0; g < 5; ++g) {
            add_copy(u, ws + f.wcatT + g * Dp, ldpi, Dp, Dp, P->lstm_u, 2 * D, D, D, g * D, 0, 1);
            add_copy(u, ws + f.wcatT + (5 + g) * Dp, ldpi, Dp, Dp, P->lstm_u, 2 * D, D, D, g * D, D, 1);
            add_copy(u, ws + f.w1ro + g * DD, Dp, Dp, Dp, Uo, 2 * D, D, D, g * D, D, 0);           // PRo gate g: the outside U[:, D:]
            add_copy(u, ws + f.w1roT + g * Dp, ldpo, Dp, Dp, Uo, 2 * D, D, D, g * D, D, 1);
        }
        add_copy(u, ws + f.wcatT + 10 * Dp, ldpi, Dp, Dp, P->in_mat, D, D, D, 0, 0, 0);
        OKR(run_copies(st, u));
        if (!p.share) {        // blocks 11..15: PLo = U_out[:, :D] h + B_out, block 16: QLo = mat_out^T h, of every inside cell
            CopyTable v; v.n = 0;
            for (int g = 0; g < 5; ++g) {
                add_copy(v, ws + f.wcat + (11 + g) * DD, Dp, Dp, Dp, P->lstm_u_out, 2 * D, D, D, g * D, 0, 0);
                add_copy(v, ws + f.wcatT + (11 + g) * Dp, ldpi, Dp, Dp, P->lstm_u_out, 2 * D, D, D, g * D, 0, 1);
                add_copy(v, ws + f.bcat + (11 + g) * Dp, Dp, 1, Dp, P->lstm_b_out, 5 * D, 1, D, 0, g * D, 0);
            }
            add_copy(v, ws + f.wcat + 16 * DD, Dp, Dp, Dp, P->out_mat, D, D, D, 0, 0, 1);
            add_copy(v, ws + f.wcatT + 16 * Dp, ldpi, Dp, Dp, P->out_mat, D, D, D, 0, 0, 0);
            OKR(run_copies(st, v));
        }
        {
            ImageList pj;
            pj.add(ws + f.wl, ws + f.wl3, 3 * Dp, Dp, Dp); pj.add(ws + f.wlT, ws + f.wlT3, Dp, 3 * Dp, 3 * Dp);
            pj.add(ws + f.wcat, ws + f.wcat3, ldpi, Dp, Dp); pj.add(ws + f.wcatT, ws + f.wcatT3, Dp, ldpi, ldpi);
            pj.add(ws + f.w1ro, ws + f.w1ro3, ldpo, Dp, Dp); pj.add(ws + f.w1roT, ws + f.w1roT3, Dp, ldpo, ldpo);
            ImageList p3;       // built in either mode: the backward call may run under the other one
            p3.add(ws + f.wcat, ws + f.wcat3s, ldpi, Dp, Dp); p3.add(ws + f.wcatT, ws + f.wcatT3s, Dp, ldpi, ldpi);
            p3.add(ws + f.w1ro, ws + f.w1ro3s, ldpo, Dp, Dp); p3.add(ws + f.w1roT, ws + f.w1roT3s, Dp, ldpo, ldpo);
            OKR(build_all_images(st, ImageList{}, pj, p3));      // one launch
        }
    }
    const bool proj3 = lstm_proj_split();
    // projections of the cells [off, off + Lc) of every sentence: every block on split-bf16 products, then the score blocks (10, and 16
    // with unshared outside functions) again on exact fp32 products -- their fragment image is a column-tile range of the whole one
    auto project_inside = [&](hipStream_t s_, int ncell, int off, int Lc) -> int {
        const LevelRowsA rows{IH, Dp, C, off, Lc};
        if (!proj3)
            return launch_rows_direct(s_, ws + f.wcat, PROJ_IMG(f.wcat3), Dp, ldpi, ncell, rows, StoreLevelE{ws + f.pi, ldpi, C, off, Lc, ws + f.bcat, 0});
        OKR(launch_rows_direct3(s_, ws + f.wcat3s, Dp, ldpi, ncell, rows, StoreLevelE{ws + f.pi, ldpi, C, off, Lc, ws + f.bcat, 0}));
        for (int blk : {10, 16}) {
            if (blk >= p.nblk) break;
            OKR(launch_rows_direct(s_, ws + f.wcat + (size_t)blk * DD, ws + f.wcat3 + (size_t)blk * DD, IMG_FRAG_F32, Dp, Dp, ncell, rows,
                                   StoreLevelE{ws + f.pi + (size_t)blk * Dp, ldpi, C, off, Lc, ws + f.bcat + (size_t)blk * Dp, 0}));
        }
        return CLIORA_OK;
    };
    auto project_outside = [&](hipStream_t s_, int ncell, int off, int Lc) -> int {
        const LevelRowsA rows{OH, Dp, C, off, Lc};
        if (!proj3)
            return launch_rows_direct(s_, ws + f.w1ro, PROJ_IMG(f.w1ro3), Dp, ldpo, ncell, rows, StoreLevelE{ws + f.po, ldpo, C, off, Lc, nullptr, 0});
        return launch_rows_direct3(s_, ws + f.w1ro3s, Dp, ldpo, ncell, rows, StoreLevelE{ws + f.po, ldpo, C, off, Lc, nullptr, 0});
    };
    // leaves
    OKR(launch_rows_direct(st, ws + f.wl, PROJ_IMG(f.wl3), Dp, 3 * Dp, B * L, PlainRowsA{X, Dp}, StoreRowsE{ws + f.t, 3 * Dp, ws + f.bl, 0, 3 * Dp}));
    hipLaunchKernelGGL(lstm_leaf_fwd, dim3(cells_grid(B * L)), dim3(256), 0, st, B, L, C, Dp, ws + f.t, p.normalize, IH, IC, ws + f.nrmi,
                       ws + f.nrmic, IS);
    LAUNCHOK("lstm_leaf_fwd");
    if (L > 1) OKR(project_inside(st, B * L, 0, L));
    // The two passes as a wavefront (see cliora_chart_forward): step k runs inside level k on the caller's stream and outside level
    // L-k on the side stream, which needs the inside projections of the levels <= k-2 (its siblings).
    // run_outside carries the flag bits of cliora_chart_forward: without a backward to come and without a hook the per-split rows
    // h_n, c_n are not written at all
    const int flags = run_outside;
    run_outside = flags & 1;
    const bool keep_pairs = (flags & CLIORA_FWD_NO_BACKWARD) == 0 || (flags & CLIORA_FWD_PAIR_STATES) != 0;
    const bool two_streams = wavefront_pays_lstm(p, g_cliora_wavefront) && run_outside;
    hipStream_t sa = st, sb = two_streams ? plan->side : st;
    auto inside_step = [&](int level) -> int {
        const LevelArgs g = level_args(p, level, false);
        const int ncell = B * g.Lc;
        hipLaunchKernelGGL(pair_scores_fwd, dim3(ncell), dim3(256), 0, sa, g, dv.arow, dv.brow, ws + f.pi + 10 * Dp, ldpi, IH, IS, IS,
                           ws + f.sp, ws + f.pp, IS);
        LAUNCHOK("pair_scores_fwd");
        hipLaunchKernelGGL(lstm_cell_fwd<LSTM_FW>, dim3(ncell), dim3((Dp / LSTM_FW + 63) / 64 * 64), 0, sa, g, dv.arow, dv.brow, ws + f.pi, ldpi,
                           ws + f.pi + 5 * Dp, ldpi, IC, IC, 1.0f, ws + f.pp, p.normalize, keep_pairs ? ws + f.y : nullptr, ws + f.x, IH, IC,
                           ws + f.nrmi, ws + f.nrmic);
        LAUNCHOK("lstm_cell_fwd");
        if (level < L - 1) OKR(project_inside(sa, ncell, g.off, g.Lc));
        return CLIORA_OK;
    };
    auto outside_step = [&](int level) -> int {
        const LevelArgs g = level_args(p, level, true);
        const int ncell = B * g.Lc;
        hipLaunchKernelGGL(pair_scores_fwd, dim3(ncell), dim3(256), 0, sb, g, dv.arow, dv.brow, ws + f.pi + (size_t)p.blk_qlo * Dp, ldpi, OH, IS, OS,
                           ws + f.sp, ws + f.pp, OS);
        LAUNCHOK("pair_scores_fwd(out)");
        hipLaunchKernelGGL(lstm_cell_fwd<LSTM_FW>, dim3(ncell), dim3((Dp / LSTM_FW + 63) / 64 * 64), 0, sb, g, dv.arow, dv.brow,
                           ws + f.pi + (size_t)p.blk_plo * Dp, ldpi, ws + f.po, ldpo, IC, OC, 0.0f, ws + f.pp, p.normalize,
                           keep_pairs ? ws + f.y : nullptr, ws + f.x, OH, OC, ws + f.nrmo, ws + f.nrmoc);
        LAUNCHOK("lstm_cell_fwd(out)");
        if (level >= 1) OKR(project_outside(sb, ncell, g.off, g.Lc));
        return CLIORA_OK;
    };
    if (two_streams) {
        HIPOK(hipEventRecord(plan->ev_fork[0], st));
        HIPOK(hipStreamWaitEvent(sb, plan->ev_fork[0], 0));
        fork_guard.arm(0, sb, plan->ev_join[0]);
    }
    if (run_outside) {
        hipLaunchKernelGGL(unit_norm_rows, dim3(cells_grid(B)), dim3(256), 0, sb, ws + f.rootp, 0, B, 1, C, C - 1, Dp, p.normalize, OH, ws + f.nrmo, OS);
        hipLaunchKernelGGL(unit_norm_rows, dim3(cells_grid(B)), dim3(256), 0, sb, ws + f.rootc, 0, B, 1, C, C - 1, Dp, p.normalize, OC, ws + f.nrmoc, OS);
        LAUNCHOK("unit_norm_rows(root)");
        if (L > 1) OKR(project_outside(sb, B, C - 1, 1));
    } else {
        HIPOK(hipMemsetAsync(OH, 0, (size_t)B * C * Dp * sizeof(float), st));
        HIPOK(hipMemsetAsync(OC, 0, (size_t)B * C * Dp * sizeof(float), st));
        HIPOK(hipMemsetAsync(OS, 0, (size_t)B * C * sizeof(float), st));
    }
    for (int k = 1; k <= L; ++k) {
        if (k <= L - 1) {
            OKR(inside_step(k));
            if (two_streams) HIPOK(hipEventRecord(plan->ev_level[k], sa));
        }
        if (run_outside && k >= 2) {
            if (two_streams && k >= 3) HIPOK(hipStreamWaitEvent(sb, plan->ev_level[k - 2], 0));
            OKR(outside_step(L - k));
        }
    }
    if (two_streams) {
        HIPOK(hipEventRecord(plan->ev_join[0], sb));
        HIPOK(hipStreamWaitEvent(st, plan->ev_join[0], 0));
        fork_guard.disarm();
    }
    if (padded) {
        CopyTable t; t.n = 0;
        add_copy(t, inside_h, D, B * C, D, IH, Dp, B * C, D, 0, 0, 0);
        add_copy(t, outside_h, D, B * C, D, OH, Dp, B * C, D, 0, 0, 0);
        add_copy(t, inside_c, D, B * C, D, IC, Dp, B * C, D, 0, 0, 0);
        add_copy(t, outside_c, D, B * C, D, OC, Dp, B * C, D, 0, 0, 0);
        OKR(run_copies(st, t));
    }
    return CLIORA_OK;
}

extern "C" int cliora_lstm_backward(cliora_plan* plan, const cliora_params* P, const float* x_span, const float* inside_h,
                                    const float* inside_c, const float* inside_s, const float* outside_h, const float* outside_c,
                                    const float* outside_s, const float* d_ih, const float* d_ic, const float* d_is, const float* d_oh,
                                    const float* d_oc, const float* d_os, void* fwd_ws, size_t fwd_ws_bytes, void* bwd_ws,
                                    size_t bwd_ws_bytes, float* d_x_span, const cliora_params* G, int ran_outside, void* stream) {
    (void)P;
    if (!plan || !x_span || !inside_h || !inside_c || !inside_s || !outside_h || !outside_c || !outside_s || !fwd_ws || !bwd_ws || !G)
        return fail(CLIORA_EINVAL, "NULL argument");
    const Plan& p = plan->p;
    if (p.arch != 1) return fail(CLIORA_EINVAL, "not a TreeLSTM plan");
    if (fwd_ws_bytes < p.fwd.total * sizeof(float)) return fail(CLIORA_ENOMEM, "forward workspace too small");
    if (bwd_ws_bytes < p.bwd.total * sizeof(float)) return fail(CLIORA_ENOMEM, "backward workspace too small");
    if (!plan->uploaded) return fail(CLIORA_EINVAL, "backward called before forward");
    hipStream_t st = (hipStream_t)stream;
    OKR(cliora_plan_ready(plan, st));
    std::lock_guard<std::mutex> lanes_lock(*plan->lanes_mu);
    ForkGuard fork_guard(st);
    const Dev dv = dev_views(p);
    float* ws = (float*)fwd_ws;
    float* wb = (float*)bwd_ws;
    const FwdLayout& f = p.fwd;
    const BwdLayout& bw = p.bwd;
    const int B = p.B, L = p.L, D = p.D, Dp = p.Dp, C = p.C, ldpi = p.nblk * Dp, ldpo = 5 * Dp;
    const size_t DD = (size_t)Dp * Dp;
    const bool padded = D != Dp;
    const float *IH = padded ? ws + f.ihp : inside_h, *OH = padded ? ws + f.ohp : outside_h;
    const float *IC = padded ? ws + f.icp : inside_c, *OC = padded ? ws + f.ocp : outside_c;
    const float *IS = inside_s, *OS = outside_s;
    const float* X = padded ? ws + f.xp : x_span;
    float *VH = wb + bw.vh, *VC = wb + bw.vc, *dG = wb + bw.dg, *dGc = wb + bw.dgc, *dStot = wb + bw.dstot;
    float *DS = wb + bw.ds, *dPI = wb + bw.dpi, *dPO = wb + bw.dpo, *dU = wb + bw.du;
    const float *Y = ws + f.y, *Xc = ws + f.x, *Sp = ws + f.sp, *Pp = ws + f.pp, *PI = ws + f.pi, *PO = ws + f.po;

    const bool two_streams = wavefront_pays_lstm(p, g_cliora_wavefront) && ran_outside;
    const bool proj3 = lstm_proj_split();
    hipStream_t sa = st, sb = two_streams ? plan->side : st;
    float *VHo = wb + bw.vh_o, *VCo = wb + bw.vc_o, *dGo = wb + bw.dg_o, *dGco = wb + bw.dgc_o, *dStoto = wb + bw.dstot_o;
    auto outside_bwd_step = [&](int level) -> int {
        const LevelArgs g = level_args(p, level, true);
        const int ncell = B * g.Lc;
        hipLaunchKernelGGL(lstm_cell_bwd_out<LSTM_W>, dim3(ncell), dim3((Dp / LSTM_W + 63) / 64 * 64), 0, sb, g, D, d_oh, d_oc, level == L - 1 ? nullptr : d_os, dv.use[ROLE_OUTB], dv.trow,
                           Pp, DS, PI, ldpi, p.blk_plo, p.blk_qlo, PO, ldpo, IC, OC, dGo, dGco, dPO, VHo, VCo, dStoto);
        LAUNCHOK("lstm_cell_bwd_out");
        if (level >= 1 && proj3)
            OKR(launch_rows_direct3(sb, ws + f.w1roT3s, ldpo, Dp, ncell, LevelRowsA{dPO, ldpo, C, g.off, g.Lc}, StoreLevelE{VHo, Dp, C, g.off, g.Lc, nullptr, 1}));
        else if (level >= 1)
            OKR(launch_rows_direct(sb, ws + f.w1roT, PROJ_IMG(f.w1roT3), ldpo, Dp, ncell, LevelRowsA{dPO, ldpo, C, g.off, g.Lc},
                                   StoreLevelE{VHo, Dp, C, g.off, g.Lc, nullptr, 1}));
        if (level == L - 1) {
            hipLaunchKernelGGL(root_bwd, dim3(1), dim3(ROOT_WAVES * 64), 0, sb, B, C, Dp, VHo, OH, ws + f.nrmo, p.normalize, wb + bw.groot);
            hipLaunchKernelGGL(root_bwd, dim3(1), dim3(ROOT_WAVES * 64), 0, sb, B, C, Dp, VCo, OC, ws + f.nrmoc, p.normalize, wb + bw.grootc);
            LAUNCHOK("root_bwd");
            return CLIORA_OK;
        }
        hipLaunchKernelGGL(lstm_scores_bwd, dim3(ncell), dim3(256), 0, sb, g, VHo, VCo, OH, OC, ws + f.nrmo, ws + f.nrmoc, p.normalize, Y, Xc,
                           Sp, Pp, OS, dStoto, dGo, dGco, DS);
        LAUNCHOK("lstm_scores_bwd(out)");
        return CLIORA_OK;
    };
    auto inside_bwd_step = [&](int level) -> int {
        const LevelArgs g = level_args(p, level, false);
        const int ncell = B * g.Lc;
        hipLaunchKernelGGL(lstm_cell_bwd_in<LSTM_W>, dim3(ncell), dim3((Dp / LSTM_W + 63) / 64 * 64), 0, sa, g, D, d_ih, d_ic, level == 0 ? nullptr : d_is, dv.use[ROLE_INA], dv.use[ROLE_INB],
                           dv.use[ROLE_OUTA], ran_outside, dv.trow, Pp, DS, PI, ldpi, p.blk_plo, p.blk_qlo, PO, ldpo, IH, IC, OH, OC, dG, dGc, dGo, dGco, dPI, VH, VC, dStot);
        LAUNCHOK("lstm_cell_bwd_in");
        if (level <= L - 2 && proj3)
            OKR(launch_rows_direct3(sa, ws + f.wcatT3s, ldpi, Dp, ncell, LevelRowsA{dPI, ldpi, C, g.off, g.Lc}, StoreLevelE{VH, Dp, C, g.off, g.Lc, nullptr, 1}));
        else if (level <= L - 2)
            OKR(launch_rows_direct(sa, ws + f.wcatT, PROJ_IMG(f.wcatT3), ldpi, Dp, ncell, LevelRowsA{dPI, ldpi, C, g.off, g.Lc},
                                   StoreLevelE{VH, Dp, C, g.off, g.Lc, nullptr, 1}));
        if (level == 0) return CLIORA_OK;
        hipLaunchKernelGGL(lstm_scores_bwd, dim3(ncell), dim3(256), 0, sa, g, VH, VC, IH, IC, ws + f.nrmi, ws + f.nrmic, p.normalize, Y, Xc, Sp,
                           Pp, IS, dStot, dG, dGc, DS);
        LAUNCHOK("lstm_scores_bwd(in)");
        return CLIORA_OK;
    };
    if (two_streams) {
        HIPOK(hipEventRecord(plan->ev_fork[0], st));
        HIPOK(hipStreamWaitEvent(sb, plan->ev_fork[0], 0));
        fork_guard.arm(0, sb, plan->ev_join[0]);
    }
    if (!ran_outside) {
        HIPOK(hipMemsetAsync(wb + bw.gw1ro, 0, 5 * DD * sizeof(float), st));
        HIPOK(hipMemsetAsync(wb + bw.groot, 0, (size_t)Dp * sizeof(float), st));
        HIPOK(hipMemsetAsync(wb + bw.grootc, 0, (size_t)Dp * sizeof(float), st));
    }
    for (int j = 0; j <= L - 1; ++j) {                   // outside level j beside inside level L-1-j (see cliora_chart_backward)
        if (ran_outside) {
            OKR(outside_bwd_step(j));
            if (two_streams) HIPOK(hipEventRecord(plan->ev_level[j], sb));
        }
        if (two_streams && j >= 1) HIPOK(hipStreamWaitEvent(sa, plan->ev_level[j - 1], 0));
        OKR(inside_bwd_step(L - 1 - j));
    }
    if (two_streams) {
        HIPOK(hipEventRecord(plan->ev_join[0], sb));
        HIPOK(hipStreamWaitEvent(st, plan->ev_join[0], 0));
        fork_guard.disarm();
    }
    // Weight gradients of the cell projections: one Dp x Dp block of dP^T H per gate block.  At d = 400 in the default arithmetic every block
    // runs on the LDS-DMA-fed eight-wave kernel of the DioraMLP pair rows (tn_gemm_dma3x; the blocks of dP are its strided A operand):
    // 52 480 chart rows x 11 + 5 blocks at L = 40 were 4 ms of register-fed tn_gemm at the end of the step.
    const bool blockwise = tn_pairs_strided_ok(Dp);
    if (ran_outside) {
        if (blockwise) {
            for (int gI = 0; gI < 5; ++gI)
                OKR(launch_tn_pairs(st, dPO + (size_t)gI * Dp, OH, B * C, Dp, wb + bw.slab, bw.slab_floats, wb + bw.gw1ro + gI * DD, (float*)nullptr, 0, 0, ldpo, Dp));
        } else
            OKR(launch_tn(st, B * C, ldpo, Dp, Dp, PlainRowsA{dPO, ldpo}, PlainRowsA{OH, Dp}, wb + bw.slab, bw.slab_floats, wb + bw.gw1ro,
                          (float*)nullptr));
    }
    hipLaunchKernelGGL(lstm_leaf_bwd, dim3(cells_grid(B * L)), dim3(256), 0, st, B, L, C, Dp, VH, VC, IH, IC, ws + f.nrmi, ws + f.nrmic,
                       p.normalize, ws + f.t, dU);
    LAUNCHOK("lstm_leaf_bwd");
    if (d_x_span)
        OKR(launch_rows_direct(st, ws + f.wlT, PROJ_IMG(f.wlT3), 3 * Dp, Dp, B * L, PlainRowsA{dU, 3 * Dp}, StoreRowsE{d_x_span, D, nullptr, 0, D}));
    if (blockwise) {
        for (int gI = 0; gI < p.nblk; ++gI)
            OKR(launch_tn_pairs(st, dPI + (size_t)gI * Dp, IH, B * C, Dp, wb + bw.slab, bw.slab_floats, wb + bw.gwcat + gI * DD, wb + bw.gbcat + gI * Dp, 0, 0,
                                ldpi, Dp));
    } else
        OKR(launch_tn(st, B * C, ldpi, Dp, Dp, PlainRowsA{dPI, ldpi}, PlainRowsA{IH, Dp}, wb + bw.slab, bw.slab_floats, wb + bw.gwcat, wb + bw.gbcat));
    OKR(launch_tn(st, B * L, 3 * Dp, Dp, Dp, PlainRowsA{dU, 3 * Dp}, PlainRowsA{X, Dp}, wb + bw.slab, bw.slab_floats, wb + bw.gwl, wb + bw.gbl));
    {
        CopyTable t; t.n = 0;
        for (int g = 0; g < 5; ++g) {
            if (G->lstm_u) {
                add_copy(t, G->lstm_u + (size_t)g * D * 2 * D, 2 * D, D, D, wb + bw.gwcat + g * DD, Dp, D, D, 0, 0, 0);
                if (p.share)      // U[:, D:] serves the inside right child and the outside parent
                    add_copy(t, G->lstm_u + (size_t)g * D * 2 * D + D, 2 * D, D, D, wb + bw.gwcat + (5 + g) * DD, Dp, D, D, 0, 0, 0,
                             wb + bw.gw1ro + g * DD, Dp, D, D, 0, 0, 0);
                else
                    add_copy(t, G->lstm_u + (size_t)g * D * 2 * D + D, 2 * D, D, D, wb + bw.gwcat + (5 + g) * DD, Dp, D, D, 0, 0, 0);
            }
            if (G->lstm_b) {
                if (g < 3) add_copy(t, G->lstm_b + g * D, D, 1, D, wb + bw.gbcat + g * Dp, Dp, 1, D, 0, 0, 0, wb + bw.gbl + g * Dp, Dp, 1, D, 0, 0, 0);
                else add_copy(t, G->lstm_b + g * D, D, 1, D, wb + bw.gbcat + g * Dp, Dp, 1, D, 0, 0, 0);
            }
        }
        for (int g = 0; g < 3; ++g)
            if (G->lstm_w) add_copy(t, G->lstm_w + (size_t)g * D * D, D, D, D, wb + bw.gwl + g * DD, Dp, D, D, 0, 0, 0);
        if (G->in_mat) add_copy(t, G->in_mat, D, D, D, wb + bw.gwcat + 10 * DD, Dp, D, D, 0, 0, 1);
        if (G->root_h) add_copy(t, G->root_h, D, 1, D, wb + bw.groot, Dp, 1, D, 0, 0, 0);
        if (G->root_c) add_copy(t, G->root_c, D, 1, D, wb + bw.grootc, Dp, 1, D, 0, 0, 0);
        OKR(run_copies(st, t));
        if (!p.share) {
            CopyTable v; v.n = 0;
            for (int g = 0; g < 5; ++g) {
                if (G->lstm_u_out) {
                    add_copy(v, G->lstm_u_out + (size_t)g * D * 2 * D, 2 * D, D, D, wb + bw.gwcat + (11 + g) * DD, Dp, D, D, 0, 0, 0);
                    add_copy(v, G->lstm_u_out + (size_t)g * D * 2 * D + D, 2 * D, D, D, wb + bw.gw1ro + g * DD, Dp, D, D, 0, 0, 0);
                }
                if (G->lstm_b_out) add_copy(v, G->lstm_b_out + g * D, D, 1, D, wb + bw.gbcat + (11 + g) * Dp, Dp, 1, D, 0, 0, 0);
            }
            if (G->out_mat) add_copy(v, G->out_mat, D, D, D, wb + bw.gwcat + 16 * DD, Dp, D, D, 0, 0, 1);
            OKR(run_copies(st, v));
        }
    }
    return CLIORA_OK;
}
