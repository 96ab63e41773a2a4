// C ABI, DioraMLP / CLIORA chart unit: forward / backward sequencing of the level kernels.
// See include/cliora_chart.h for the contract and the reference lines it replaces.
#include "api_common.hpp"
#include <hip/hip_ext.h>
#include "level_kernels.hpp"
#include "resident_kernels.hpp"
#include "vl_kernels.hpp"

// the attention kernels keep a wave's regions in registers: 9 per wave cover R <= 36 (the reference's 36 boxes), 16 cover R <= 64
#define ATTEND_LAUNCH(kern, R_, grid, stream, ...)                                                              \
    do {                                                                                                      \
        if ((R_) <= 36) hipLaunchKernelGGL(kern<9>, grid, dim3(256), 0, stream, __VA_ARGS__);                  \
        else hipLaunchKernelGGL(kern<16>, grid, dim3(256), 0, stream, __VA_ARGS__);                            \
    } while (0)

// ------------------------------------------------------------------ fused level kernels (level_kernels.hpp)
// Geometry of a level's compose launch: the plan's tasks (plan.cpp compose_geom) on at most `cap` workgroups per column block.
struct ComposeLaunch { int TG, SP, ntask, gx; };
static ComposeLaunch compose_launch(const cliora_plan* plan, int level, bool outside_pass) {
    const Plan& p = plan->p;
    const int32_t* e = p.level_geom.data() + ((size_t)(outside_pass ? p.L : 0) + level) * PLEVEL_INTS;
    ComposeLaunch q{e[5], e[6], e[7], std::min(e[7], p.compose_cap)};
    // workgroups are dealt round-robin over the 8 XCDs by linear id (x + y*gx): with gx a multiple of 8 the column blocks
    // that gather the same operand rows share one XCD's L2 (speed only, never correctness)
    if (q.gx >= 8 && (q.gx + 7) / 8 * 8 <= p.compose_cap) q.gx = (q.gx + 7) / 8 * 8;
    return q;
}

// LDS of the compose kernels: the column block's weight image (split-bf16: S dwords per column; exact fp32: K) + reduction slots
static size_t compose_lds_bytes(int ct, int S, bool with_slots) {
    return (size_t)ct * 16 * S * sizeof(uint32_t) + (with_slots ? (size_t)LC_SLOTS * ct * 64 * sizeof(float4) : 0);
}

template <int CT, int K16, bool F32>
static int launch_level_compose_inst(hipStream_t st, const uint32_t* Wimg, int S, int K, int ncb, const PairLevel& lv, const float* PA, int lda,
                                     const float* PB, int ldb, const float* bias, const float* Pp, float* HP, size_t hp_stride, int Dp,
                                     uint32_t* ymask, float* Y, const ComposeLaunch& q) {
    OKR(cliora_ensure_max_lds((const void*)level_compose_fwd<CT, K16, F32>));
    hipLaunchKernelGGL((level_compose_fwd<CT, K16, F32>), dim3(q.gx, ncb), dim3(512), compose_lds_bytes(CT, S, true), st, Wimg, S, K, lv, PA, lda,
                       PB, ldb, bias, Pp, q.TG, q.SP, q.ntask, HP, hp_stride, Dp, ymask, Y);
    LAUNCHOK("level_compose_fwd");
    return CLIORA_OK;
}

// y = relu(W2 relu(PL(a) + PR(b)) + b2) for every pair of the level and g = sum_n p_n y_n per cell, into HP (SP parts).
// W: the plain fp32 weight (exact mode), Wimg: its split-bf16 image.
static int launch_level_compose(hipStream_t st, const float* W, const float* Wimg, int S3, int Dp, int ct, int ncb, const PairLevel& lv,
                                const float* PA, int lda, const float* PB, int ldb, const float* bias, const float* Pp, float* HP,
                                size_t hp_stride, uint32_t* ymask, float* Y, const ComposeLaunch& q) {
    const bool f32 = !split_bf16();
    const uint32_t* I = reinterpret_cast<const uint32_t*>(f32 ? W : Wimg);
    const int S = f32 ? Dp : S3;
#define LC_ARGS st, I, S, Dp, ncb, lv, PA, lda, PB, ldb, bias, Pp, HP, hp_stride, Dp, ymask, Y, q
#define LC_CASE(c, k16) return f32 ? launch_level_compose_inst<c, k16, true>(LC_ARGS) : launch_level_compose_inst<c, k16, false>(LC_ARGS)
    if (ct == 5 && Dp == 400) LC_CASE(5, 25);
    switch (ct) {
        case 5: LC_CASE(5, 0);
        case 4: LC_CASE(4, 0);
        case 2: LC_CASE(2, 0);
        default: LC_CASE(1, 0);
    }
#undef LC_CASE
#undef LC_ARGS
}

template <int CT, int K16, bool F32, bool TILED = false>
static int launch_level_compose_bwd_inst(hipStream_t st, const uint32_t* Wimg, int S, int K, int ncb, const PairLevel& lv, const float* dG,
                                         const uint32_t* ymask, const float* Pp, const float* PA, int lda, const float* PB, int ldb, const float* b2,
                                         float* DA, float* DZ, float* X, float* DPP, float* DPB, int cus) {
    OKR(cliora_ensure_max_lds((const void*)level_compose_bwd<CT, K16, F32, TILED>));
    const int ntiles = (lv.ncell + 15) / 16 * lv.N;
    const int cap = std::max(1, cus / ncb);
    const int passes = (ntiles + 8 * cap - 1) / (8 * cap);
    int gx = (ntiles + 8 * passes - 1) / (8 * passes);
    if (gx >= 8 && (gx + 7) / 8 * 8 <= cap) gx = (gx + 7) / 8 * 8;      // column blocks of the same tiles on one XCD
    hipLaunchKernelGGL((level_compose_bwd<CT, K16, F32, TILED>), dim3(gx, ncb), dim3(512), compose_lds_bytes(CT, S, false) + (size_t)((K + 31) / 32 * 32) * sizeof(float), st, Wimg, S, K, lv, dG, ymask,
                       Pp, PA, lda, PB, ldb, b2, K, DA, DZ, X, DPP, DPB);
    LAUNCHOK("level_compose_bwd");
    return CLIORA_OK;
}

// backward of the level's compose layer: DA, DZ, X rows and the partial dG.y_n (see level_compose_bwd).  WT: plain fp32 W2^T.
static int launch_level_compose_bwd(hipStream_t st, const float* WT, const float* WTimg, int S3, int Dp, int ct, int ncb, const PairLevel& lv,
                                    const float* dG, const uint32_t* ymask, const float* Pp, const float* PA, int lda, const float* PB,
                                    int ldb, const float* b2, float* DA, float* DZ, float* X, float* DPP, float* DPB, bool tiled = false, int cus = 256) {
    if (lv.N <= 0 || lv.ncell <= 0) return CLIORA_OK;
    const bool f32 = !split_bf16();
    if (tiled) {         // X, DZ as tiled split-bf16 operands of tn_gemm_tiles (pair_tiles_ok: d = 400, split mode)
        if (f32 || ct != 5 || Dp != 400) return fail(CLIORA_EINVAL, "tiled pair-row operands: d = 400 in split-bf16 mode only");
        return launch_level_compose_bwd_inst<5, 25, false, true>(st, reinterpret_cast<const uint32_t*>(WTimg), S3, Dp, ncb, lv, dG, ymask, Pp, PA, lda,
                                                                   PB, ldb, b2, DA, DZ, X, DPP, DPB, cus);
    }
    const uint32_t* I = reinterpret_cast<const uint32_t*>(f32 ? WT : WTimg);
    const int S = f32 ? Dp : S3;
#define LB_ARGS st, I, S, Dp, ncb, lv, dG, ymask, Pp, PA, lda, PB, ldb, b2, DA, DZ, X, DPP, DPB, cus
#define LB_CASE(c, k16) return f32 ? launch_level_compose_bwd_inst<c, k16, true>(LB_ARGS) : launch_level_compose_bwd_inst<c, k16, false>(LB_ARGS)
    if (ct == 5 && Dp == 400) LB_CASE(5, 25);
    switch (ct) {
        case 5: LB_CASE(5, 0);
        case 4: LB_CASE(4, 0);
        case 2: LB_CASE(2, 0);
        default: LB_CASE(1, 0);
    }
#undef LB_CASE
#undef LB_ARGS
}

// `stop`: an event to be signalled by the level_project launch itself (hipExtLaunchKernelGGL's stop event: the dispatch packet's own
// completion signal) instead of by a hipEventRecord behind it -- the record is a barrier packet of its own on the chain's queue, and the
// kernel behind it started 6-7 us late at every level of the forward's inside chain (profiles/r04_notes.md).  nullptr: none.
template <int CT, int SP>
static int launch_level_project_inst(hipStream_t st, const float* Wfrag, int K, int ncols, int ncell, int Lc, int C, int off, const float* HP,
                                     size_t hp_stride, int normalize, const float* bias, float* P, int ldp, float* H, float* nrm,
                                     const ScoreArgs& sc, hipEvent_t stop) {
    const int nrg = (ncell + 15) / 16;
    const int nrgp = nrg >= 8 ? (nrg + 7) / 8 * 8 : nrg;          // column blocks of one row group share an XCD
    const int ncb = ncols / (16 * CT);
    LAUNCH_SIGNALLING(stop, (level_project<CT, SP>), dim3(sc.nscore + nrgp * ncb), dim3(256), 0, st, Wfrag, K, nrg, nrgp, ncb, ncell, Lc,
                      C, off, HP, hp_stride, normalize, bias, P, ldp, H, nrm, sc);
    LAUNCHOK("level_project");
    return CLIORA_OK;
}
template <int CT>
static int launch_level_project_sp(hipStream_t st, int SP, const float* Wfrag, int K, int ncols, int ncell, int Lc, int C, int off,
                                   const float* HP, size_t hp_stride, int normalize, const float* bias, float* P, int ldp, float* H, float* nrm,
                                   const ScoreArgs& sc, hipEvent_t stop) {
    switch (SP) {
        case 1: return launch_level_project_inst<CT, 1>(st, Wfrag, K, ncols, ncell, Lc, C, off, HP, hp_stride, normalize, bias, P, ldp, H, nrm, sc, stop);
        case 2: return launch_level_project_inst<CT, 2>(st, Wfrag, K, ncols, ncell, Lc, C, off, HP, hp_stride, normalize, bias, P, ldp, H, nrm, sc, stop);
        case 3: return launch_level_project_inst<CT, 3>(st, Wfrag, K, ncols, ncell, Lc, C, off, HP, hp_stride, normalize, bias, P, ldp, H, nrm, sc, stop);
        default: return launch_level_project_inst<CT, 4>(st, Wfrag, K, ncols, ncell, Lc, C, off, HP, hp_stride, normalize, bias, P, ldp, H, nrm, sc, stop);
    }
}
// h = unit(sum of the SP parts) for the level's cells (chart rows of H, raw norms) and P = h W^T + bias; sc: the next level's
// scoring, in the same launch
static int launch_level_project(hipStream_t st, int SP, const float* Wfrag, int K, int ncols, int ncell, int Lc, int C, int off, const float* HP,
                                size_t hp_stride, int normalize, const float* bias, float* P, int ldp, float* H, float* nrm, const ScoreArgs& sc,
                                hipEvent_t stop = nullptr) {
    const int nt = ncols / 16, nrt = (ncell + 15) / 16;
    if (g_ksplit_min_blocks < 0) { const char* e = getenv("CLIORA_KSPLIT_MIN_BLOCKS"); g_ksplit_min_blocks = e ? atoi(e) : 1000; }
    // a block's MFMA work and operand bytes are fixed by its tile: wide tiles only when the launch still covers the chip
    if (nt % 5 == 0 && nrt * (nt / 5) >= g_ksplit_min_blocks)
        return launch_level_project_sp<5>(st, SP, Wfrag, K, ncols, ncell, Lc, C, off, HP, hp_stride, normalize, bias, P, ldp, H, nrm, sc, stop);
    return launch_level_project_sp<1>(st, SP, Wfrag, K, ncols, ncell, Lc, C, off, HP, hp_stride, normalize, bias, P, ldp, H, nrm, sc, stop);
}

// ---- the two passes' launches of one wavefront step as ONE grid each (level_compose_fwd2 / level_project2), on the caller's stream
// CLIORA_COMPOSE_DUAL = ring depth of the two-splits-per-wave compose (level_kernels.hpp: compose_fwd_tasks_dual), 0: one split per wave
static int compose_dual_pd() { static const int d = [] { const char* e = getenv("CLIORA_COMPOSE_DUAL"); return e ? atoi(e) : 2; }(); return d; }
template <int CT, int K16, bool F32>
static int launch_level_compose2_inst(hipStream_t st, const ComposeSeg& a, const ComposeSeg& b, int S, int K, int ncb, const float* Pp,
                                      size_t hp_stride, int Dp, uint32_t* ymask, float* Y) {
    int gx = a.gx + b.gx;
    if (gx >= 8) gx = (gx + 7) / 8 * 8;            // padding blocks return at once; column blocks of one x share an XCD (linear id = x + y * gx)
    if constexpr (K16 == 25 && !F32) {             // the d = 400 split-bf16 instance only
        const size_t lds_dual = compose_lds_bytes(CT, S, true) + (size_t)CT * 16 * sizeof(float);
        if (compose_dual_pd() == 2) {
            OKR(cliora_ensure_max_lds((const void*)level_compose_fwd2<CT, K16, F32, 2>));
            hipLaunchKernelGGL((level_compose_fwd2<CT, K16, F32, 2>), dim3(gx, ncb), dim3(512), lds_dual, st, a, b, S, K, Pp, hp_stride, Dp, ymask, Y);
            LAUNCHOK("level_compose_fwd2(dual 2)");
            return CLIORA_OK;
        }
        if (compose_dual_pd() == 3) {
            OKR(cliora_ensure_max_lds((const void*)level_compose_fwd2<CT, K16, F32, 3>));
            hipLaunchKernelGGL((level_compose_fwd2<CT, K16, F32, 3>), dim3(gx, ncb), dim3(512), lds_dual, st, a, b, S, K, Pp, hp_stride, Dp, ymask, Y);
            LAUNCHOK("level_compose_fwd2(dual 3)");
            return CLIORA_OK;
        }
    }
    OKR(cliora_ensure_max_lds((const void*)level_compose_fwd2<CT, K16, F32>));
    hipLaunchKernelGGL((level_compose_fwd2<CT, K16, F32>), dim3(gx, ncb), dim3(512), compose_lds_bytes(CT, S, true), st, a, b, S, K, Pp, hp_stride, Dp, ymask, Y);
    LAUNCHOK("level_compose_fwd2");
    return CLIORA_OK;
}
// a.Wimg / b.Wimg: the pass's plain fp32 W2 in the exact mode, its split-bf16 image otherwise (the caller picks by split_bf16())
static int launch_level_compose2(hipStream_t st, const ComposeSeg& a, const ComposeSeg& b, int S3, int Dp, int ct, int ncb, const float* Pp,
                                 size_t hp_stride, uint32_t* ymask, float* Y) {
    if (a.gx + b.gx <= 0) return CLIORA_OK;
    const bool f32 = !split_bf16();
    const int S = f32 ? Dp : S3;
#define LC2_ARGS st, a, b, S, Dp, ncb, Pp, hp_stride, Dp, ymask, Y
#define LC2_CASE(c, k16) return f32 ? launch_level_compose2_inst<c, k16, true>(LC2_ARGS) : launch_level_compose2_inst<c, k16, false>(LC2_ARGS)
    if (ct == 5 && Dp == 400) LC2_CASE(5, 25);
    switch (ct) {
        case 5: LC2_CASE(5, 0);
        case 4: LC2_CASE(4, 0);
        case 2: LC2_CASE(2, 0);
        default: LC2_CASE(1, 0);
    }
#undef LC2_CASE
#undef LC2_ARGS
}
// diagnostics (timing experiments, WRONG results; only in a build with -DCLIORA_DIAG_P2, tools/ab/build_variant.sh -- the shipped library
// ignores the variable): CLIORA_P2_DIAG bit 0: no score blocks, bit 1: no projection blocks
#ifdef CLIORA_DIAG_P2
static int p2_diag() { static const int d = [] { const char* e = getenv("CLIORA_P2_DIAG"); return e ? atoi(e) : 0; }(); return d; }
#else
static constexpr int p2_diag() { return 0; }
#endif
template <int SP0, int SP1>
static int launch_level_project2_inst(hipStream_t st, const ProjSeg& a_, const ProjSeg& b_) {
    ProjSeg a = a_, b = b_;
    if (p2_diag() & 1) { a.sc.nscore = 0; b.sc.nscore = 0; }
    if (p2_diag() & 2) { a.nproj = 0; b.nproj = 0; }
    const int n = a.sc.nscore + b.sc.nscore + a.nproj + b.nproj + a.nfin + b.nfin;
    if (n <= 0) return CLIORA_OK;
    // block order: 1 = the projection blocks first, the score blocks behind them (measured 24.05 -> 22.97 us per launch at c2; 0 = scores first)
    static const int order = [] { const char* e = getenv("CLIORA_P2_ORDER"); return e ? atoi(e) : 1; }();
    hipLaunchKernelGGL((level_project2<SP0, SP1>), dim3(n), dim3(256), 0, st, a, b, order);
    LAUNCHOK("level_project2");
    return CLIORA_OK;
}
static bool project2_parts_ok(int SP) { return SP == 1 || SP == 2 || SP == 4; }      // what plan.cpp compose_geom deals (powers of two <= HP_PARTS)
static int launch_level_project2(hipStream_t st, int SP0, int SP1, const ProjSeg& a, const ProjSeg& b) {
#define P2_ROW(s0) \
    switch (SP1) { case 1: return launch_level_project2_inst<s0, 1>(st, a, b); case 2: return launch_level_project2_inst<s0, 2>(st, a, b); \
                   default: return launch_level_project2_inst<s0, 4>(st, a, b); }
    if (!project2_parts_ok(SP0) || !project2_parts_ok(SP1)) return fail(CLIORA_EINVAL, "level_project2: parts per level must be 1, 2 or 4");
    switch (SP0) {
        case 1: P2_ROW(1)
        case 2: P2_ROW(2)
        default: P2_ROW(4)
    }
#undef P2_ROW
}

// Two streams pay when a level's launches carry enough work to outweigh the event per step (3 us): measured on MI355X from
// B 64 / L 8 / d 400 (1.17 -> 0.90 ms) to B 64 / L 40 / d 400 (23.1 -> 20.9 ms); B 8 / L 10 / d 50 loses (0.76 -> 0.95 ms).
// cliora_set_wavefront / CLIORA_WAVEFRONT=0|1 force it off / on.
static bool wavefront_pays(const Plan& p, int env) {
    if (p.L <= 2 || env == 0) return false;
    if (env >= 1) return true;
    const double row_floats_per_level = (double)(p.R_in + p.R_out) / (2.0 * (p.L - 1)) * p.Dp;
    return row_floats_per_level >= 100e3;
}

// The two chains of the wavefront on ONE queue (round 5): step k = ONE compose grid over inside level k and outside level L-k
// (level_compose_fwd2) and ONE projection / score grid for both (level_project2), on the caller's stream.  The two-stream form pays
// a cross-stream event per step (a barrier packet on both queues: the launch behind it starts ~5 us late, profiles/r04_timeline.txt)
// and staggers the chains; one queue needs neither.  Same tasks, same summation order: bitwise the two-stream (and the sequential)
// results.  DioraMLP and CLIORA plans; cliora_set_wavefront(CLIORA_WAVEFRONT_MERGED) / CLIORA_WAVEFRONT=2 forces it, AUTO takes it
// wherever the two-stream wavefront would pay, ON (1) keeps the two streams.
static bool merged_pays(const Plan& p, bool vl) {
    (void)vl;                 // CLIORA plans too: the attention residual of the inside cells sits between the two grids (cliora.py:140-157)
    if (p.arch != 0 || p.L <= 2) return false;
    if (g_cliora_wavefront != 2 && !(g_cliora_wavefront < 0 && wavefront_pays(p, -1))) return false;
    for (size_t e = 0; e < (size_t)2 * p.L; ++e) {
        const int sp = p.level_geom[e * PLEVEL_INTS + 6];
        if (p.level_geom[e * PLEVEL_INTS + 1] > 0 && !project2_parts_ok(sp)) return false;
    }
    return true;
}

// One workgroup per sentence for every level of both passes (resident_kernels.hpp) when a row fits a wavefront: text-only DioraMLP,
// Dp <= 64, no per-pair hook states.  Measured on MI355X (tools/resident_ab.py, profiles/r03_resident_ab.txt), forward / forward +
// backward ms, launches -> resident:
//   D 50 / B 8 / L 10 (configs[0])   0.198 -> 0.196 / 0.80 -> 0.56        D 50 / B 256 / L 10   0.31 -> 0.23 / 1.59 -> 1.25
//   D 32 / B 64 / L 12               0.246 -> 0.211 / 1.07 -> 0.86        D 16 / B 128 / L 8    0.159 -> 0.094 / 0.75 -> 0.51
//   D 64 / B 64 / L 16               0.36 -> 0.46 / 1.40 -> 1.81          D 64 / B 8 / L 40     1.19 -> 3.84 / 3.0 -> 11.7
//   D 64 / B 256 / L 16              0.54 -> 0.39 / 3.02 -> 2.35          D 64 / B 256 / L 40   4.29 -> 3.49 / 20.8 -> 13.4
// one workgroup (12 waves) per sentence is a chain of L steps of 10-16 us each whatever the batch, so AUTO takes it while a
// sentence's chart is short (span pairs per sentence, both passes, <= g_cliora_resident_max_pairs = 1000: L <= 12) or when the
// batch gives at least every other CU a sentence (B >= CUs / 2 = 128).  The two directions share every buffer format with the
// launch path, so either can run on either (tests/test_gpu_resident.py).
// cliora_set_resident / CLIORA_RESIDENT=0|1 force it off / on (on is still refused for shapes the kernels do not cover).
static size_t resident_lds_bytes(const Plan& p) {
    return ((size_t)(p.share ? 7 : 8) * p.Dp * p.Dp + (size_t)RES_WAVES * RES_SCR) * sizeof(float);      // shared plans: + the leaf weight
}
static bool resident_pays(const cliora_plan* plan, bool vl, bool compress, bool /*backward*/) {
    const Plan& p = plan->p;
    if (g_cliora_resident == 0 || vl || compress || p.arch != 0 || p.L < 2 || p.Dp > 64) return false;
    if (resident_lds_bytes(p) > 160 * 1024) return false;
    if (g_cliora_resident < 0) {
        // short sentences (the chain of L steps beats the launch boundaries), or enough sentences to give every other CU its own
        // (the kernels are then throughput-bound and win at every length: D 64 / B 256 / L 40 20.8 -> 13.4 ms)
        if (p.P_in + p.P_out > g_cliora_resident_max_pairs && p.B * 2 < plan->ncu) return false;
    }
    return true;
}
static ResArgs resident_args(const cliora_plan* plan, float* ws, float* IH, float* OH, float* IS, float* OS, int run_outside) {
    const Plan& p = plan->p;
    const FwdLayout& f = p.fwd;
    const Dev dv = dev_views(p);
    ResArgs a{};
    a.tabs = p.d_tables;
    a.pa_in = (uint32_t)p.dev.pair_a_in; a.pb_in = (uint32_t)p.dev.pair_b_in; a.pa_out = (uint32_t)p.dev.pair_a_out; a.pb_out = (uint32_t)p.dev.pair_b_out;
    a.lvl_in = (uint32_t)p.dev.lvl_base_in; a.lvl_out = (uint32_t)p.dev.lvl_base_out;
    a.use_ina = dv.use[ROLE_INA]; a.use_inb = dv.use[ROLE_INB]; a.use_outa = dv.use[ROLE_OUTA]; a.use_outb = dv.use[ROLE_OUTB];
    a.B = p.B; a.L = p.L; a.C = p.C; a.D = p.D; a.Dp = p.Dp; a.ldpi = p.nblk * p.Dp; a.nblk = p.nblk; a.blk_plo = p.blk_plo; a.blk_qlo = p.blk_qlo;
    a.share = p.share; a.normalize = p.normalize; a.run_outside = run_outside; a.ct = f.ct3; a.gy = f.ncb3; a.R_in = p.R_in;
    a.IH = IH; a.OH = OH; a.IS = IS; a.OS = OS; a.PI = ws + f.pi; a.PO = ws + f.po; a.nrmi = ws + f.nrmi; a.nrmo = ws + f.nrmo;
    a.Sp = ws + f.sp; a.Pp = ws + f.pp; a.ymask = reinterpret_cast<uint32_t*>(ws + f.ymask);
    a.wcatT = ws + f.wcatT; a.bcat = ws + f.bcat; a.w2iT = ws + f.w2iT; a.b2i = ws + f.b2i; a.w2oT = ws + f.w2oT; a.b2o = ws + f.b2o;
    a.w1roT = ws + f.w1roT; a.rootp = ws + f.rootp;
    a.wcat = ws + f.wcat; a.w2i = ws + f.w2i; a.w2o = ws + f.w2o; a.w1ro = ws + f.w1ro;
    a.T = ws + f.t; a.wlT = ws + f.wlT; a.wl = ws + f.wl; a.bl = ws + f.bl;
    return a;
}

// split-bf16 / fragment images of the packed weights, built in either arithmetic mode (two small launches): the backward call may run
// under the other one
static int build_chart_images(const cliora_plan* plan, float* ws, bool compress, hipStream_t st) {
    const Plan& p = plan->p;
    const FwdLayout& f = p.fwd;
    const int Dp = p.Dp, ldpi = p.nblk * p.Dp;
    ImageList im;
    im.add(ws + f.w2i, ws + f.w2i3, Dp, Dp, Dp); im.add(ws + f.w2iT, ws + f.w2iT3, Dp, Dp, Dp);
    if (!p.share) { im.add(ws + f.w2o, ws + f.w2o3, Dp, Dp, Dp); im.add(ws + f.w2oT, ws + f.w2oT3, Dp, Dp, Dp); }
    ImageList pj;
    pj.add(ws + f.wl, ws + f.wl3, Dp, Dp, Dp); pj.add(ws + f.wlT, ws + f.wlT3, Dp, Dp, Dp);
    pj.add(ws + f.wcat, ws + f.wcat3, ldpi, Dp, Dp); pj.add(ws + f.wcatT, ws + f.wcatT3, Dp, ldpi, ldpi);
    pj.add(ws + f.w1ro, ws + f.w1ro3, Dp, Dp, Dp); pj.add(ws + f.w1roT, ws + f.w1roT3, Dp, Dp, Dp);
    pj.add(ws + f.matp, ws + f.matq3, Dp, Dp, Dp);
    if (compress) { pj.add(ws + f.rootw, ws + f.rootw3, Dp, Dp, Dp); pj.add(ws + f.rootwT, ws + f.rootwT3, Dp, Dp, Dp); }
    ImageList p3;               // split-bf16 fragment images of the projection transposes (rows_gemm_ksplit3x: the backward's per-level GEMMs)
    if (p.arch == 0) { p3.add(ws + f.wcatT, ws + f.wcatT3s, Dp, ldpi, ldpi); p3.add(ws + f.w1roT, ws + f.w1roT3s, Dp, Dp, Dp); }
    OKR(build_all_images(st, im, pj, p3));
    return CLIORA_OK;
}

// ------------------------------------------------------------------ forward
extern "C" int cliora_chart_forward(cliora_plan* plan, const cliora_params* P, const float* x_span, const float* obj_span,
                                    const float* drop_mask, float* inside_h, float* inside_s, float* outside_h,
                                    float* outside_s, float* inside_c, void* fwd_ws, size_t fwd_ws_bytes, int run_outside,
                                    void* stream) {
    if (!plan || !P || !x_span || !inside_h || !inside_s || !outside_h || !outside_s || !fwd_ws)
        return fail(CLIORA_EINVAL, "NULL argument");
    const Plan& p = plan->p;
    const bool vl = p.R > 0;
    if (p.arch != 0) return fail(CLIORA_EINVAL, "TreeLSTM plan: use cliora_lstm_forward");
    if (vl && !obj_span) return fail(CLIORA_EINVAL, "a CLIORA plan (R > 0) needs obj_span");
    if (!vl && obj_span) return fail(CLIORA_EINVAL, "obj_span given to a text-only plan (R = 0)");
    if (fwd_ws_bytes < p.fwd.total * sizeof(float)) return fail(CLIORA_ENOMEM, "forward workspace too small");
    if ((run_outside & CLIORA_FWD_PAIR_STATES) && fwd_ws_bytes < (p.fwd.total + p.fwd.pair_h_floats) * sizeof(float))
        return fail(CLIORA_ENOMEM, "forward workspace has no room for the pair states (cliora_plan_pair_states_bytes)");
    if (!p.share && (!P->out_w1 || !P->out_b1 || !P->out_w2 || !P->out_b2 || !P->out_mat))
        return fail(CLIORA_EINVAL, "share=0 needs the out_* parameters");
    // compress = True (diora.py:342-343): the outside root is a projection of the inside root, so the passes run one after the other
    const bool compress = P->root_mat != nullptr;
    if (!compress && !P->root_h) return fail(CLIORA_EINVAL, "root_h (or root_mat with compress = True) is NULL");
    hipStream_t st = (hipStream_t)stream;
    OKR(cliora_plan_ready(plan, st));
    std::lock_guard<std::mutex> lanes_lock(*plan->lanes_mu);
    ForkGuard fork_guard(st);                          // declared after the lock: runs (joins the side streams) before it is released
    float* ws = (float*)fwd_ws;
    const FwdLayout& f = p.fwd;
    const int B = p.B, L = p.L, D = p.D, Dp = p.Dp, C = p.C, nb = p.nblk, ldpi = nb * Dp;
    const bool padded = D != Dp;
    float* IH = padded ? ws + f.ihp : inside_h;
    float* OH = padded ? ws + f.ohp : outside_h;
    float* IS = inside_s;
    float* OS = outside_s;
    const float* X = padded ? ws + f.xp : x_span;
    const float* OBJ = vl ? (padded ? ws + f.objp : obj_span) : nullptr;
    const float* w1o = p.share ? P->in_w1 : P->out_w1;

    const bool resident = !(run_outside & CLIORA_FWD_PAIR_STATES) && resident_pays(plan, vl, compress, false);     // its workgroups do the leaves too

    // ---- pack parameters into padded / concatenated / transposed layouts ----
    {
        CopyTable t; t.n = 0;
        add_copy(t, ws + f.wl, Dp, Dp, Dp, P->leaf_w, D, D, D, 0, 0, 0);
        add_copy(t, ws + f.wlT, Dp, Dp, Dp, P->leaf_w, D, D, D, 0, 0, 1);
        add_copy(t, ws + f.bl, Dp, 1, Dp, P->leaf_b, D, 1, D, 0, 0, 0);
        // Wcat rows: [W1L_in ; W1R_in ; mat_in^T ; (W1L_out ; mat_out^T)]
        add_copy(t, ws + f.wcat + 0 * (size_t)Dp * Dp, Dp, Dp, Dp, P->in_w1, 2 * D, D, D, 0, 0, 0);
        add_copy(t, ws + f.wcat + 1 * (size_t)Dp * Dp, Dp, Dp, Dp, P->in_w1, 2 * D, D, D, 0, D, 0);
        add_copy(t, ws + f.wcat + 2 * (size_t)Dp * Dp, Dp, Dp, Dp, P->in_mat, D, D, D, 0, 0, 1);
        add_copy(t, ws + f.wcatT + 0 * Dp, ldpi, Dp, Dp, P->in_w1, 2 * D, D, D, 0, 0, 1);
        add_copy(t, ws + f.wcatT + 1 * Dp, ldpi, Dp, Dp, P->in_w1, 2 * D, D, D, 0, D, 1);
        add_copy(t, ws + f.wcatT + 2 * Dp, ldpi, Dp, Dp, P->in_mat, D, D, D, 0, 0, 0);
        add_copy(t, ws + f.bcat, Dp, 1, Dp, P->in_b1, D, 1, D, 0, 0, 0);
        add_copy(t, ws + f.bcat + Dp, Dp, 1, 2 * Dp, nullptr, 0, 0, 0, 0, 0, 0);
        if (!p.share) {
            add_copy(t, ws + f.wcat + 3 * (size_t)Dp * Dp, Dp, Dp, Dp, P->out_w1, 2 * D, D, D, 0, 0, 0);
            add_copy(t, ws + f.wcat + 4 * (size_t)Dp * Dp, Dp, Dp, Dp, P->out_mat, D, D, D, 0, 0, 1);
            add_copy(t, ws + f.wcatT + 3 * Dp, ldpi, Dp, Dp, P->out_w1, 2 * D, D, D, 0, 0, 1);
            add_copy(t, ws + f.wcatT + 4 * Dp, ldpi, Dp, Dp, P->out_mat, D, D, D, 0, 0, 0);
            add_copy(t, ws + f.bcat + 3 * Dp, Dp, 1, Dp, P->out_b1, D, 1, D, 0, 0, 0);
            add_copy(t, ws + f.bcat + 4 * Dp, Dp, 1, Dp, nullptr, 0, 0, 0, 0, 0, 0);
            add_copy(t, ws + f.w2o, Dp, Dp, Dp, P->out_w2, D, D, D, 0, 0, 0);
            add_copy(t, ws + f.w2oT, Dp, Dp, Dp, P->out_w2, D, D, D, 0, 0, 1);
            add_copy(t, ws + f.b2o, Dp, 1, Dp, P->out_b2, D, 1, D, 0, 0, 0);
        }
        add_copy(t, ws + f.w1ro, Dp, Dp, Dp, w1o, 2 * D, D, D, 0, D, 0);
        add_copy(t, ws + f.w1roT, Dp, Dp, Dp, w1o, 2 * D, D, D, 0, D, 1);
        add_copy(t, ws + f.w2i, Dp, Dp, Dp, P->in_w2, D, D, D, 0, 0, 0);
        add_copy(t, ws + f.w2iT, Dp, Dp, Dp, P->in_w2, D, D, D, 0, 0, 1);
        add_copy(t, ws + f.b2i, Dp, 1, Dp, P->in_b2, D, 1, D, 0, 0, 0);
        if (!compress) add_copy(t, ws + f.rootp, Dp, 1, Dp, P->root_h, D, 1, D, 0, 0, 0);
        else {
            add_copy(t, ws + f.rootw, Dp, Dp, Dp, P->root_mat, D, D, D, 0, 0, 1);      // the projection h -> h M as a Linear weight: M^T
            add_copy(t, ws + f.rootwT, Dp, Dp, Dp, P->root_mat, D, D, D, 0, 0, 0);
        }
        add_copy(t, ws + f.matp, Dp, Dp, Dp, P->in_mat, D, D, D, 0, 0, 0);
        if (padded) add_copy(t, ws + f.xp, Dp, B * L, Dp, x_span, D, B * L, D, 0, 0, 0);
        if (padded && vl) add_copy(t, ws + f.objp, Dp, B * p.R, Dp, obj_span, D, B * p.R, D, 0, 0, 0);
        OKR(run_copies(st, t));
        // the weight images (split-bf16 / fragment layouts of the packed weights) feed the launch path's GEMMs only: a resident forward
        // leaves them out, and a launch-path backward on such a workspace builds them first (cliora_chart_backward)
        if (!resident) OKR(build_chart_images(plan, ws, compress, st));
        if (resident) plan->note_imageless(fwd_ws);
        else (void)plan->take_imageless(fwd_ws);
    }

    // ---- leaves: h = unit(tanh(x Wl^T + bl))  (diora.py:58-63, 283-292) ----
    if (!resident) OKR(launch_rows_direct(st, ws + f.wl, PROJ_IMG(f.wl3), Dp, Dp, B * L, PlainRowsA{X, Dp}, StoreRowsE{ws + f.t, Dp, ws + f.bl, 1, Dp}));
    if (resident) {
    } else if (vl) {   // h = unit(unit(tanh) + attention(...)), c = unit(context)   (cliora.py:71-80, 290-301)
        LevelArgs g0 = level_args(p, 0, false);
        ATTEND_LAUNCH(cell_attend_fwd, p.R, dim3(B * L), st, g0, L, (const float*)nullptr, (size_t)0, 0,
                           ws + f.t, OBJ, p.R, drop_mask, p.normalize, IH, ws + f.nrmi, ws + f.att_u, ws + f.att_nrmu, ws + f.att_pk,
                           inside_c, D, IS);
        LAUNCHOK("cell_attend_fwd(leaves)");
    } else {
        hipLaunchKernelGGL(unit_norm_rows, dim3(cells_grid(B * L)), dim3(256), 0, st, ws + f.t, Dp, B * L, L, C, 0, Dp, p.normalize,
                           IH, ws + f.nrmi, IS);
        LAUNCHOK("unit_norm_rows");
    }
    if (L > 1 && !resident)
        OKR(launch_rows_direct(st, ws + f.wcat, PROJ_IMG(f.wcat3), Dp, ldpi, B * L, LevelRowsA{IH, Dp, C, 0, L},
                        StoreLevelE{ws + f.pi, ldpi, C, 0, L, ws + f.bcat, 0}));

    // Every level runs the fused kernels of level_kernels.hpp: split scores -> compose + aggregate -> norm + projection.
    // No per-pair row reaches HBM; the backward gets one ReLU bit per element of y (skipped when no backward will follow).
    const int flags = run_outside;
    run_outside = flags & 1;
    const bool keep = (flags & CLIORA_FWD_NO_BACKWARD) == 0;
    float* PH = nullptr;                               // per-pair compose outputs, only for the hooks
    if (flags & CLIORA_FWD_PAIR_STATES) PH = ws + f.pair_h;      // room checked before the first launch
    const size_t hp_stride = (size_t)B * C * Dp;
    float* HPi = ws + f.hp;                            // partial aggregates of the level being composed, one buffer per pass
    float* HPo = ws + f.hp_o;
    uint32_t* YM = keep ? reinterpret_cast<uint32_t*>(ws + f.ymask) : nullptr;
    auto pair_level = [&](int level, bool outside_pass) {
        const LevelArgs g = level_args(p, level, outside_pass);
        const int32_t* t = p.d_tables;
        const size_t base = outside_pass ? p.lvl_base_out[level] : p.lvl_base_in[level];
        PairLevel lv;
        lv.pa = t + (outside_pass ? p.dev.pair_a_out : p.dev.pair_a_in) + base;
        lv.pb = t + (outside_pass ? p.dev.pair_b_out : p.dev.pair_b_in) + base;
        lv.Lc = g.Lc; lv.N = g.N; lv.C = C; lv.ncell = B * g.Lc; lv.rowbase = g.rowbase; lv.off = g.off;
        lv.tilebase = outside_pass ? p.tile_base_out(level) : p.tile_base_in(level);
        return lv;
    };

    // Scoring of target level T of a pass.  newest >= 0: the cells of that level are still partial aggregates (HP, SPn parts) --
    // the scoring then rides in the projection launch of that level (level_project); newest < 0: every operand is final.
    auto score_args = [&](int T, bool outside_pass, int newest, int SPn) {
        ScoreArgs sc{};
        sc.g = level_args(p, T, outside_pass);
        sc.nscore = B * sc.g.Lc;
        {
            const size_t base = outside_pass ? p.lvl_base_out[T] : p.lvl_base_in[T];
            sc.pa = p.d_tables + (outside_pass ? p.dev.pair_a_out : p.dev.pair_a_in) + base;
            sc.pb = p.d_tables + (outside_pass ? p.dev.pair_b_out : p.dev.pair_b_in) + base;
        }
        sc.QA = ws + f.pi + (size_t)(outside_pass ? p.blk_qlo : 2) * Dp; sc.ldA = ldpi;
        sc.HB = outside_pass ? OH : IH; sc.HA = IH;
        sc.SA = IS; sc.SB = outside_pass ? OS : IS;
        sc.Sp = ws + f.sp; sc.Pp = ws + f.pp; sc.Sout = outside_pass ? OS : IS;
        sc.a_can_be_new = outside_pass ? 0 : 1;
        sc.new_lo = sc.new_hi = 0;
        if (newest >= 0) {
            sc.new_lo = p.level_offset[newest]; sc.new_hi = sc.new_lo + (L - newest);
            sc.HPn = outside_pass ? HPo : HPi; sc.hp_stride = hp_stride; sc.SPn = SPn;
        }
        sc.normalize = p.normalize;
        sc.QRleaf = ws + f.qrleaf; sc.L = L;
        return sc;
    };
    auto launch_scores = [&](hipStream_t s, const ScoreArgs& sc) {
        hipLaunchKernelGGL(level_scores, dim3(sc.nscore), dim3(256), 0, s, sc);
        LAUNCHOK("level_scores");
        return CLIORA_OK;
    };

    // ---- the two passes as a wavefront ------------------------------------------------------------------------------------
    // Outside target level t composes (parent outside cell, sibling inside cell) pairs whose siblings sit at inside levels
    // <= L-2-t (diora.py:358-398, outside_index.py:39-127): the outside pass does not have to wait for the inside pass to finish,
    // only to stay behind it.  Step k runs inside level k on the caller's stream and outside level L-k on the plan's side stream
    // (which waits for the event of inside step k-1: its riding scores of level L-k-1 read inside level k-1).  The 2(L-1)
    // dependent levels of the reference become L steps of two concurrent, latency-bound launches (tools/ubench/wavefront_bench.hip:
    // two streams run such kernels side by side at the cost of one; an event dependency per step adds 3 us).
    // round 5: the same wavefront on ONE queue -- the two passes' launches of a step as one grid each (level_compose_fwd2 /
    // level_project2): no side stream, no event per step (merged_pays)
    const bool merged = run_outside && !compress && !resident && merged_pays(p, vl);
    const bool two_streams = wavefront_pays(p, g_cliora_wavefront) && run_outside && !compress && !resident && !merged;
    hipStream_t sa = st, sb = two_streams ? plan->side : st;

    auto inside_step = [&](int level, hipEvent_t project_stop) -> int {          // diora.py:295-331 for one level
        const LevelArgs g = level_args(p, level, false);
        const int ncell = B * g.Lc;
        const ComposeLaunch cq = compose_launch(plan, level, false);
        const int SP = cq.SP;
        {
            ProfScope ps(CLIORA_KCLASS_COMPOSE_FWD, sa);
            OKR(launch_level_compose(sa, ws + f.w2i, ws + f.w2i3, f.S3, Dp, f.ct3, f.ncb3, pair_level(level, false), ws + f.pi, ldpi,
                                     ws + f.pi + Dp, ldpi, ws + f.b2i, ws + f.pp, HPi, hp_stride, YM, PH, cq));
        }
        if (vl) {   // cliora.py:140-157: attention residual between the aggregate and the second unit norm, then the projections
            ATTEND_LAUNCH(cell_attend_fwd, p.R, dim3(ncell), sa, g, L, HPi, hp_stride, SP, (const float*)nullptr, OBJ, p.R, drop_mask, p.normalize, IH,
                               ws + f.nrmi, ws + f.att_u, ws + f.att_nrmu, ws + f.att_pk, (float*)nullptr, D, IS);
            LAUNCHOK("cell_attend_fwd");
            if (level < L - 1) {
                // projections of the finished rows (the attention kernel wrote H and its norms: nothing to normalise, no chart output)
                // and the next level's scores in the same launch, the newest operands read from H as one "part"
                ScoreArgs sc = score_args(level + 1, false, level, 1);
                sc.HPn = nullptr;                  // the newest cells are final rows of H already
                OKR(launch_level_project(sa, 1, ws + f.wcat3, Dp, ldpi, ncell, g.Lc, C, g.off, IH, 0, 0, ws + f.bcat, ws + f.pi, ldpi,
                                         (float*)nullptr, (float*)nullptr, sc));
            }
        } else if (level < L - 1) {      // norm + projection of this level, and the next level's scores in the same launch
            OKR(launch_level_project(sa, SP, ws + f.wcat3, Dp, ldpi, ncell, g.Lc, C, g.off, HPi, hp_stride, p.normalize, ws + f.bcat,
                                     ws + f.pi, ldpi, IH, ws + f.nrmi, score_args(level + 1, false, level, SP), project_stop));
        } else {
            hipLaunchKernelGGL(level_finish, dim3(cells_grid(ncell)), dim3(256), 0, sa, ncell, g.Lc, C, g.off, Dp, HPi, hp_stride, SP,
                               p.normalize, IH, ws + f.nrmi);
            LAUNCHOK("level_finish");
        }
        return CLIORA_OK;
    };
    auto outside_step = [&](int level) -> int {         // diora.py:358-398 for one level
        const LevelArgs g = level_args(p, level, true);
        const int ncell = B * g.Lc;
        const ComposeLaunch cq = compose_launch(plan, level, true);
        const int SP = cq.SP;
        {
            ProfScope ps(CLIORA_KCLASS_COMPOSE_FWD, sb);
            OKR(launch_level_compose(sb, ws + f.w2o, ws + f.w2o3, f.S3, Dp, f.ct3, f.ncb3, pair_level(level, true),
                                     ws + f.pi + (size_t)p.blk_plo * Dp, ldpi, ws + f.po, Dp, ws + f.b2o, ws + f.pp, HPo, hp_stride, YM, PH, cq));
        }
        if (level >= 1)      // the level below is scored in the same launch: its newest parents are this level's cells
            OKR(launch_level_project(sb, SP, ws + f.w1ro3, Dp, Dp, ncell, g.Lc, C, g.off, HPo, hp_stride, p.normalize, nullptr,
                                     ws + f.po, Dp, OH, ws + f.nrmo, score_args(level - 1, true, level, SP)));
        else {
            hipLaunchKernelGGL(level_finish, dim3(cells_grid(ncell)), dim3(256), 0, sb, ncell, g.Lc, C, g.off, Dp, HPo, hp_stride, SP,
                               p.normalize, OH, ws + f.nrmo);
            LAUNCHOK("level_finish(out)");
        }
        return CLIORA_OK;
    };

    if (L > 1 && !resident) {
        // QR = M h of the leaves: the partner of a newest-level LEFT child is always a leaf (see level_project)
        OKR(launch_rows_direct(st, ws + f.matp, PROJ_IMG(f.matq3), Dp, Dp, B * L, LevelRowsA{IH, Dp, C, 0, L},
                               StoreRowsE{ws + f.qrleaf, Dp, nullptr, 0, Dp}));
        OKR(launch_scores(st, score_args(1, false, -1, 0)));
    }
    if (two_streams) {
        HIPOK(hipEventRecord(plan->ev_fork[0], st));
        HIPOK(hipStreamWaitEvent(sb, plan->ev_fork[0], 0));
        fork_guard.arm(0, sb, plan->ev_join[0]);
    }
    // root of the outside chart (diora.py:337-356) and the scores of the level below it: parents = the root only.  compress = True:
    // the root of sentence b is unit(inside_h[b, root] @ root_mat_out) -- one row per sentence, after the inside pass.
    auto init_root = [&]() -> int {
        if (run_outside && resident) return CLIORA_OK;      // the sentence's workgroup forms its own root row
        if (run_outside) {
            if (compress)
                OKR(launch_rows_direct(sb, ws + f.rootw, PROJ_IMG(f.rootw3), Dp, Dp, B, LevelRowsA{IH, Dp, C, C - 1, 1},
                                       StoreRowsE{ws + f.rootpb, Dp, nullptr, 0, Dp}));
            hipLaunchKernelGGL(unit_norm_rows, dim3(cells_grid(B)), dim3(256), 0, sb, compress ? ws + f.rootpb : ws + f.rootp, compress ? Dp : 0, B, 1,
                               C, C - 1, Dp, p.normalize, OH, ws + f.nrmo, OS);
            LAUNCHOK("unit_norm_rows(root)");
            if (L > 1) {
                OKR(launch_rows_direct(sb, ws + f.w1ro, PROJ_IMG(f.w1ro3), Dp, Dp, B, LevelRowsA{OH, Dp, C, C - 1, 1}, StoreLevelE{ws + f.po, Dp, C, C - 1, 1, nullptr, 0}));
                OKR(launch_scores(sb, score_args(L - 2, true, -1, 0)));
            }
        } else {
            HIPOK(hipMemsetAsync(OH, 0, (size_t)B * C * Dp * sizeof(float), st));
            HIPOK(hipMemsetAsync(OS, 0, (size_t)B * C * sizeof(float), st));
        }
        return CLIORA_OK;
    };
    if (!compress) OKR(init_root());
    if (resident) {
        // ---- one workgroup per sentence, every level of both passes (resident_kernels.hpp) ----
        ResArgs a = resident_args(plan, ws, IH, OH, IS, OS, run_outside);
        a.X = X;
        if (padded) { a.outIH = inside_h; a.outOH = run_outside ? outside_h : nullptr; }     // the un-padded charts straight from the kernel
        if (!keep) a.ymask = nullptr;
        static const bool res_trace = [] { const char* e = getenv("CLIORA_RES_TRACE"); return e && atoi(e) != 0; }();
        a.trace = res_trace ? reinterpret_cast<unsigned long long*>(plan->trace_words) : nullptr;
        OKR(cliora_ensure_max_lds((const void*)resident_fwd));
        ProfScope ps(CLIORA_KCLASS_COMPOSE_FWD, st);
        hipLaunchKernelGGL(resident_fwd, dim3(std::min(B, std::max(1, plan->ncu))), dim3(RES_THREADS), resident_lds_bytes(p), st, a);
        LAUNCHOK("resident_fwd");
    }
    if (merged) {
        const bool f32w = !split_bf16();
        auto compose_seg = [&](int level, bool outside_pass) {
            ComposeSeg sg{};
            const int32_t* e = p.level_geom.data() + ((size_t)(outside_pass ? L : 0) + level) * PLEVEL_INTS;
            const float* w = outside_pass ? (f32w ? ws + f.w2o : ws + f.w2o3) : (f32w ? ws + f.w2i : ws + f.w2i3);
            sg.Wimg = reinterpret_cast<const uint32_t*>(w);
            sg.lv = pair_level(level, outside_pass);
            sg.PA = outside_pass ? ws + f.pi + (size_t)p.blk_plo * Dp : ws + f.pi; sg.lda = ldpi;
            sg.PB = outside_pass ? ws + f.po : ws + f.pi + Dp; sg.ldb = outside_pass ? Dp : ldpi;
            sg.bias = outside_pass ? ws + f.b2o : ws + f.b2i;
            sg.TG = e[5]; sg.SP = e[6]; sg.ntask = e[7];
            // the workgroups the level's geometry was sized for (its share of the chip in this step): both segments together fit the CUs once
            sg.gx = std::max(1, std::min(e[7], compose_cap_share(p, outside_pass ? 1 : 0, level)));
            sg.HP = outside_pass ? HPo : HPi;
            return sg;
        };
        auto project_seg = [&](int level, bool outside_pass, int SP) {
            ProjSeg q{};
            const LevelArgs g = level_args(p, level, outside_pass);
            q.ncell = B * g.Lc; q.Lc = g.Lc; q.C = C; q.off = g.off; q.K = Dp;
            q.HP = outside_pass ? HPo : HPi; q.hp_stride = hp_stride; q.normalize = p.normalize;
            q.H = outside_pass ? OH : IH; q.nrm = ws + (outside_pass ? f.nrmo : f.nrmi);
            const bool projects = outside_pass ? level >= 1 : level < L - 1;      // else: chart rows only (level_finish)
            const bool attended = vl && !outside_pass;     // CLIORA inside cells: cell_attend_fwd has written the final rows of H and their norms
            if (attended) { q.HP = IH; q.hp_stride = 0; q.normalize = 0; q.H = nullptr; q.nrm = nullptr; }
            if (!projects) {
                if (!attended) { q.nfin = (int)cells_grid(q.ncell); q.SPfin = SP; }
                return q;
            }
            q.nrg = ((q.ncell + 15) / 16 + P2_RT - 1) / P2_RT;          // row groups of P2_RT sixteen-row tiles
            q.nrgp = q.nrg >= 8 ? (q.nrg + 7) / 8 * 8 : q.nrg;
            const int ncols = outside_pass ? Dp : ldpi;
            const int ctb = outside_pass ? P2_CTB : P2_CTA;                // column tiles per block of this segment
            q.ntc = ncols / 16;
            q.nproj = q.nrgp * ((q.ntc + ctb - 1) / ctb);
            q.Wfrag = ws + (outside_pass ? f.w1ro3 : f.wcat3);
            q.bias = outside_pass ? nullptr : ws + f.bcat;
            q.P = ws + (outside_pass ? f.po : f.pi); q.ldp = ncols;
            q.sc = score_args(outside_pass ? level - 1 : level + 1, outside_pass, level, attended ? 1 : SP);   // the next level of the pass, scored in the same launch
            if (attended) q.sc.HPn = nullptr;              // the newest operands are final rows of H already
            return q;
        };
        for (int k = 1; k <= L; ++k) {
            ComposeSeg ca{}, cb{};
            ProjSeg qa{}, qb{};
            if (k <= L - 1) { ca = compose_seg(k, false); qa = project_seg(k, false, ca.SP); }
            if (k >= 2) { cb = compose_seg(L - k, true); qb = project_seg(L - k, true, cb.SP); }
            {
                ProfScope ps(CLIORA_KCLASS_COMPOSE_FWD, st);
                OKR(launch_level_compose2(st, ca, cb, f.S3, Dp, f.ct3, f.ncb3, ws + f.pp, hp_stride, YM, PH));
            }
#ifdef CLIORA_DIAG_BG
            // Timing proxy for the "old / new pairs" pipeline (profiles/r06_notes.md section 5): what does a BACKGROUND compose grid cost the
            // critical chain?  Step k's compose is launched a second time on the side stream, behind the first (same inputs, the same values
            // written again: results unchanged), so that it runs beside this step's projection / score grid and the next step's compose --
            // where the background composition of the next level's old splits would run.  CLIORA_BG=1: every step; 2: the share (k-1)/(k+1)
            // of its tasks, which is what the old splits are.
            {
                static const int bg = [] { const char* e = getenv("CLIORA_BG"); return e ? atoi(e) : 0; }();
                if (bg && k >= 2 && k <= L - 1) {
                    HIPOK(hipEventRecord(plan->ev_level[k], st));
                    HIPOK(hipStreamWaitEvent(plan->side, plan->ev_level[k], 0));
                    ComposeSeg ba = ca, bb = cb;
                    if (bg == 2) { ba.ntask = std::max(1, ba.ntask * (k - 1) / (k + 1)); ba.gx = std::min(ba.gx, ba.ntask); if (bb.gx) { bb.ntask = std::max(1, bb.ntask * (k - 1) / (k + 1)); bb.gx = std::min(bb.gx, bb.ntask); } }
                    OKR(launch_level_compose2(plan->side, ba, bb, f.S3, Dp, f.ct3, f.ncb3, ws + f.pp, hp_stride, YM, PH));
                    fork_guard.arm(0, plan->side, plan->ev_join[0]);
                }
            }
#endif
            if (vl && k <= L - 1) {   // cliora.py:140-157: the attention residual between the aggregate and the second unit norm (inside cells only)
                const LevelArgs g = level_args(p, k, false);
                ATTEND_LAUNCH(cell_attend_fwd, p.R, dim3(B * g.Lc), st, g, L, HPi, hp_stride, ca.SP, (const float*)nullptr, OBJ, p.R, drop_mask, p.normalize, IH,
                                   ws + f.nrmi, ws + f.att_u, ws + f.att_nrmu, ws + f.att_pk, (float*)nullptr, D, IS);
                LAUNCHOK("cell_attend_fwd");
            }
            OKR(launch_level_project2(st, (ca.gx && !vl) ? ca.SP : 1, cb.gx ? cb.SP : 1, qa, qb));
        }
    }
    for (int k = 1; k <= L && !resident && !merged; ++k) {
        if (k <= L - 1) {
            const bool by_kernel = two_streams && stop_events_on() && !vl && k < L - 1;      // the step ends with a level_project launch
            OKR(inside_step(k, by_kernel ? plan->ev_level[k] : nullptr));
            if (two_streams && !by_kernel) HIPOK(hipEventRecord(plan->ev_level[k], sa));
        }
        if (run_outside && k >= 2 && !compress) {
            const int level = L - k;
            // compose reads the siblings' projections of inside levels <= k-2, the riding scores (level >= 1) those of level k-1
            const int need = level >= 1 ? k - 1 : k - 2;
            if (two_streams && need >= 1) HIPOK(hipStreamWaitEvent(sb, plan->ev_level[need], 0));
            OKR(outside_step(level));
        }
    }
    if (compress) {           // the outside pass after the inside pass, on the caller's stream
        OKR(init_root());
        for (int level = L - 2; level >= 0 && run_outside; --level) OKR(outside_step(level));
    }
    if (two_streams) {
        HIPOK(hipEventRecord(plan->ev_join[0], sb));
        HIPOK(hipStreamWaitEvent(st, plan->ev_join[0], 0));
        fork_guard.disarm();
    }
    if (padded && !(resident && run_outside)) {
        CopyTable t; t.n = 0;
        if (!resident) add_copy(t, inside_h, D, B * C, D, IH, Dp, B * C, D, 0, 0, 0);
        add_copy(t, outside_h, D, B * C, D, OH, Dp, B * C, D, 0, 0, 0);      // (inside-only resident call: the zeroed outside chart)
        OKR(run_copies(st, t));
    }
    return CLIORA_OK;
}

// ------------------------------------------------------------------ backward
extern "C" int cliora_chart_backward(cliora_plan* plan, const cliora_params* P, const float* x_span, const float* obj_span,
                                     const float* drop_mask, const float* inside_h, const float* inside_s,
                                     const float* outside_h, const float* outside_s, const float* d_inside_h,
                                     const float* d_inside_s, const float* d_outside_h, const float* d_outside_s,
                                     void* fwd_ws, size_t fwd_ws_bytes, void* bwd_ws, size_t bwd_ws_bytes, float* d_x_span,
                                     float* d_obj_span, const cliora_params* G, int ran_outside, void* stream) {
    if (!plan || !x_span || !inside_h || !inside_s || !outside_h || !outside_s || !fwd_ws || !bwd_ws || !G)
        return fail(CLIORA_EINVAL, "NULL argument");
    const Plan& p = plan->p;
    const bool vl = p.R > 0;
    if (p.arch != 0) return fail(CLIORA_EINVAL, "TreeLSTM plan: use cliora_lstm_backward");
    if (vl && !obj_span) return fail(CLIORA_EINVAL, "a CLIORA plan (R > 0) needs obj_span");
    if (fwd_ws_bytes < p.fwd.total * sizeof(float)) return fail(CLIORA_ENOMEM, "forward workspace too small");
    if (bwd_ws_bytes < p.bwd.total * sizeof(float)) return fail(CLIORA_ENOMEM, "backward workspace too small");
    if (!plan->uploaded) return fail(CLIORA_EINVAL, "backward called before forward");
    hipStream_t st = (hipStream_t)stream;
    OKR(cliora_plan_ready(plan, st));
    std::lock_guard<std::mutex> lanes_lock(*plan->lanes_mu);
    ForkGuard fork_guard(st);
    fork_guard.arm(1, plan->side2, plan->ev_join[1]);     // the weight-gradient stream takes work at several points of the call
    const Dev dv = dev_views(p);
    float* ws = (float*)fwd_ws;
    float* wb = (float*)bwd_ws;
    const FwdLayout& f = p.fwd;
    const BwdLayout& bw = p.bwd;
    const int B = p.B, L = p.L, D = p.D, Dp = p.Dp, C = p.C, nb = p.nblk, ldpi = nb * Dp;
    const bool padded = D != Dp;
    const float* IH = padded ? ws + f.ihp : inside_h;
    const float* OH = padded ? ws + f.ohp : outside_h;
    const float* IS = inside_s;
    const float* OS = outside_s;
    const float* X = padded ? ws + f.xp : x_span;
    float *VH = wb + bw.vh, *dG = wb + bw.dg, *dStot = wb + bw.dstot, *DA = wb + bw.da, *DS = wb + bw.ds;
    float *dPI = wb + bw.dpi, *dPO = wb + bw.dpo, *dU = wb + bw.du;
    const float *Sp = ws + f.sp, *Pp = ws + f.pp, *PI = ws + f.pi;
    float *DZ = wb + bw.dz, *Xp = wb + bw.x, *DPP = wb + bw.dpp, *DPB = wb + bw.dpb;
    const uint32_t* YM = reinterpret_cast<const uint32_t*>(ws + f.ymask);
    auto pair_level = [&](int level, bool outside_pass) {
        const LevelArgs g = level_args(p, level, outside_pass);
        const int32_t* t = p.d_tables;
        const size_t base = outside_pass ? p.lvl_base_out[level] : p.lvl_base_in[level];
        PairLevel lv;
        lv.pa = t + (outside_pass ? p.dev.pair_a_out : p.dev.pair_a_in) + base;
        lv.pb = t + (outside_pass ? p.dev.pair_b_out : p.dev.pair_b_in) + base;
        lv.Lc = g.Lc; lv.N = g.N; lv.C = C; lv.ncell = B * g.Lc; lv.rowbase = g.rowbase; lv.off = g.off;
        lv.tilebase = outside_pass ? p.tile_base_out(level) : p.tile_base_in(level);
        return lv;
    };
    const float* OBJ = vl ? (padded ? ws + f.objp : obj_span) : nullptr;
    // CLIORA: the unit-norm / softmax backward of the inside cells works on u = unit(aggregate), not on h
    const float* IHn = vl ? ws + f.att_u : IH;
    const float* nrmIn = vl ? ws + f.att_nrmu : ws + f.nrmi;

    // ---- the two backward chains as a wavefront ---------------------------------------------------------------------------
    // The backward of outside level t only feeds inside cells of levels <= L-2-t (its siblings), so the backward of the inside
    // pass does not have to wait for the whole outside backward: step j runs outside level j on the plan's side stream and
    // inside level L-1-j on the caller's stream, which waits for the event of outside step j-1 (see cliora_chart_forward).
    // compress = True: the outside root's gradient flows into the inside root, so the outside backward ends before the inside one starts
    const bool compress = P && P->root_mat;
    const bool resident = resident_pays(plan, vl, compress, true);
    if (!resident && plan->take_imageless(fwd_ws))            // the forward ran on the resident kernels and left the weight images out
        OKR(build_chart_images(plan, ws, compress, st));
    const bool two_streams = wavefront_pays(p, g_cliora_wavefront) && ran_outside && !compress && !resident;
    hipStream_t sa = st, sb = two_streams ? plan->side : st, sw = plan->side2;
    float *VHo = wb + bw.vh_o, *dGo = wb + bw.dg_o, *dStoto = wb + bw.dstot_o;
    // unit-norm backward in the projection-backward GEMM's epilogue (chart_kernels.hpp: NormBwdLevelE): text-only charts without compress
    // (CLIORA normalises around the attention residual; compress keeps per-sentence root gradients in dGo).  CLIORA_FUSE_DNORM=0: off.
    static const bool fuse_off = [] { const char* e = getenv("CLIORA_FUSE_DNORM"); return e && atoi(e) == 0; }();
    // (the OUTSIDE chart has no attention residual: its chain keeps the fusion in CLIORA plans too -- round 4, c3 chart step)
    const bool fuse_dnorm = !fuse_off && !compress && !resident;
    // the projection-backward GEMMs on split-bf16 products with RT x CT tiles (rows_gemm_ksplit3x): CLIORA_BWD_GEMM3 = 10 RT + CT
    // Round 5: these GEMMs sat on two limits at once -- what a CU pulls from L2 (a 16 x 16 block streams 150 KB for one tile) and the
    // fp32-input MFMA (16 x 16 x 4 in 32 cycles) -- and each remedy alone lost (profiles/r05_notes.md section 14); split-bf16 products on
    // 32 x 48 tiles take both: c2 3.005 -> 2.905 ms, L 40 16.37 -> 16.19.  Shapes 22 / 32 / 24: 2.95-2.96; 42 / 33 / 25 / 13 / 15: 3.0-3.14.
    // CLIORA_BWD_GEMM3=0: the fp32 16 x 16 kernel (always in the exact-fp32 arithmetic mode).
    static const int bwd_gemm3 = gemm3_shape_env("CLIORA_BWD_GEMM3", true);
    static const int bwd_gemm3_min = [] { const char* e = getenv("CLIORA_BWD_GEMM3_MIN"); return e ? atoi(e) : 0; }();     // 0 / 128 / 256 / 512 / 768 cells: 2.905 / 2.91 / 2.92 / 2.92 / 2.94
    auto gemm3_pays = [&](int ncell) { return bwd_gemm3 > 0 && split_bf16() && ncell >= bwd_gemm3_min; };
    // X / DZ of the pair rows as tiled split-bf16 operands (wgrad_tiles.hpp) instead of fp32 rows.  CLIORA_PAIR_TILES=0: off.
    const bool tiled = pair_tiles_ok(Dp) && !resident;

    // Sibling uses of inside level s in the outside pass (cell_gather_bwd_sib), on the OUTSIDE chain's stream: they are complete once
    // the outside backward has done level L-2-s, one step before the inside chain reaches level s (which already waits for that
    // step's event).  Shared weights: two scratch charts that cell_gather_bwd_in adds; unshared: blocks 3 / 4 of dPI directly.
    float *sibPL = p.share ? wb + bw.sib_pl : dPI + (size_t)3 * Dp, *sibQL = p.share ? wb + bw.sib_ql : dPI + (size_t)4 * Dp;
    const int ldsib = p.share ? Dp : ldpi;
    // Which chain sums them: in the first steps of the backward the OUTSIDE chain is the longer one (outside levels 0, 1, ... hold the
    // most pair rows, the inside levels L-1, L-2, ... the fewest cells), later the inside chain (its low levels' cells have the most
    // uses): the inside levels below `sib_split` (reached in the later steps) get their sibling sums from the outside chain, the others
    // walk the list themselves.  Measured per step at c2 (profiles/r04_notes.md): inside / outside chain 57-84 / 75-103 us in steps
    // 0-11 with everything on the outside chain, 100 / 60-99 in steps 12-19.  CLIORA_SIB_SPLIT = 0 (never) .. L - 1 (always).
    static const int sib_env = [] { const char* e = getenv("CLIORA_SIB_SPLIT"); return e ? atoi(e) : -1; }();
    // (at most L - 1: the root level L - 1 has no sibling use, and no cell_gather_bwd_sib launch ever writes its rows)
    const int sib_split = !two_streams ? 0 : sib_env >= 0 ? std::min(sib_env, L - 1) : (L * 2) / 5;
    auto sib_on_outside_chain = [&](int s_level) { return s_level < sib_split; };
    auto sibling_runs = [&](int s_level) { return s_level >= 0 && s_level <= L - 1 && ran_outside && sib_on_outside_chain(s_level); };
    auto sibling_gather = [&](int s_level, hipEvent_t done) -> int {
        if (!sibling_runs(s_level)) return CLIORA_OK;
        const LevelArgs gi = level_args(p, s_level, false);
        LAUNCH_SIGNALLING(done, cell_gather_bwd_sib, dim3(B * gi.Lc), dim3(256), 0, sb, gi, dv.use[ROLE_OUTA], DA, DS, OH, sibPL, sibQL, ldsib, wb + bw.sib_s);
        LAUNCHOK("cell_gather_bwd_sib");
        return CLIORA_OK;
    };
    // `done`: an event that the step's LAST launch signals itself (LAUNCH_SIGNALLING) -- nullptr: the caller records
    auto outside_bwd_step = [&](int level, hipEvent_t done) -> int {
        const LevelArgs g = level_args(p, level, true);     // N == 0 at the root level
        const int ncell = B * g.Lc;
        // levels 1 .. L-2: the unit-norm backward rides in the projection-backward GEMM's epilogue (NormBwdLevelE; the gather leaves
        // H . vH behind), no cell_dnorm launch
        const bool fused_norm = fuse_dnorm && level >= 1 && level <= L - 2;
        hipLaunchKernelGGL(cell_gather_bwd_out, dim3(ncell), dim3(256), 0, sb, g, D, d_outside_h,
                           level == L - 1 ? nullptr : d_outside_s, dv.use[ROLE_OUTB], DA, DS, PI, ldpi, p.blk_qlo, dPO, VHo, dStoto,
                           OH, ws + f.po, fused_norm ? wb + bw.dots_o : nullptr);
        LAUNCHOK("cell_gather_bwd_out");
        if (fused_norm && gemm3_pays(ncell))
            OKR(launch_rows_direct3x(sb, ws + f.w1roT3s, Dp, Dp, ncell, LevelRowsA{dPO, Dp, C, g.off, g.Lc},
                              NormBwdLevelE{dGo, VHo, OH, ws + f.nrmo, wb + bw.dots_o, Dp, C, g.off, g.Lc, p.normalize}, bwd_gemm3));
        else if (fused_norm)
            OKR(launch_rows_direct(sb, ws + f.w1roT, PROJ_IMG(f.w1roT3), Dp, Dp, ncell, LevelRowsA{dPO, Dp, C, g.off, g.Lc},
                            NormBwdLevelE{dGo, VHo, OH, ws + f.nrmo, wb + bw.dots_o, Dp, C, g.off, g.Lc, p.normalize}));
        else if (level >= 1 && gemm3_pays(ncell))
            OKR(launch_rows_direct3x(sb, ws + f.w1roT3s, Dp, Dp, ncell, LevelRowsA{dPO, Dp, C, g.off, g.Lc},
                              StoreLevelE{VHo, Dp, C, g.off, g.Lc, nullptr, 1}, bwd_gemm3));
        else if (level >= 1)
            OKR(launch_rows_direct(sb, ws + f.w1roT, PROJ_IMG(f.w1roT3), Dp, Dp, ncell, LevelRowsA{dPO, Dp, C, g.off, g.Lc},
                            StoreLevelE{VHo, Dp, C, g.off, g.Lc, nullptr, 1}));
        if (level == L - 1) {
            if (compress) {      // per-sentence roots: the unit-norm backward of each, kept in dGo for the inside root and d root_mat_out
                hipLaunchKernelGGL(cell_dnorm, dim3(cells_grid(ncell)), dim3(256), 0, sb, g, VHo, OH, ws + f.nrmo, p.normalize, dGo);
                LAUNCHOK("cell_dnorm(root)");
                return CLIORA_OK;
            }
            LAUNCH_SIGNALLING(done, root_bwd, dim3(1), dim3(ROOT_WAVES * 64), 0, sb, B, C, Dp, VHo, OH, ws + f.nrmo, p.normalize, wb + bw.groot);
            LAUNCHOK("root_bwd");
            return CLIORA_OK;
        }
        if (!fused_norm) {
            hipLaunchKernelGGL(cell_dnorm, dim3(cells_grid(ncell)), dim3(256), 0, sb, g, VHo, OH, ws + f.nrmo, p.normalize, dGo);
            LAUNCHOK("cell_dnorm(out)");
        }
        {
            ProfScope ps(CLIORA_KCLASS_COMPOSE_BWD, sb);
            OKR(launch_level_compose_bwd(sb, ws + f.w2oT, ws + f.w2oT3, f.S3, Dp, f.ct3, f.ncb3, pair_level(level, true), dGo, YM, Pp,
                                         PI + (size_t)p.blk_plo * Dp, ldpi, ws + f.po, Dp, ws + f.b2o, DA, DZ, Xp, DPP, DPB, tiled));
        }
        const bool sib_last = sibling_runs(L - 2 - level);
        LAUNCH_SIGNALLING(sib_last ? nullptr : done, cell_dsoftmax, dim3(cells_grid(ncell)), dim3(256), 0, sb, g, f.ncb3, DPP, DPB, Sp, Pp, OS, dStoto, DS);
        LAUNCHOK("cell_dsoftmax(out)");
        OKR(sibling_gather(L - 2 - level, done));      // the inside level whose sibling uses are final with this outside level
        return CLIORA_OK;
    };

    // The projections' weight gradient dWcat = sum over cells dP^T h is split in two: the rows of the levels >= ksplit are final once that
    // level's gather has run, and their share runs on the GEMM stream beside the remaining (latency-bound, small) levels; only the low
    // levels' rows are left for the tail, where the pair weight gradient owns the chip.
    static const int ksplit_env = [] { const char* e = getenv("CLIORA_WGRAD_SPLIT_LEVEL"); return e ? atoi(e) : -1; }();
    // fp32 element-load kernel (exact mode, other widths): level 2 (round 2, L = 20: 4.52 -> 4.42 ms, higher levels leave more for the tail).
    // Split mode at d = 400 (round 4: the remapped LDS-DMA kernel, a launch per projection block): the early part is cheap enough to be
    // worth starting in the MIDDLE of the chain, so that the GEMM stream is idle again when the low levels' share arrives -- c2 with
    // level 2 / 4 / 8 / 10 / 12: 3.36 / 3.35 / 3.30-3.32 / 3.29 / 3.32 ms
    const bool wcat_split_path = split_bf16() && tn_pairs_strided_ok(Dp);
    // (with all blocks in one launch the part pays a little earlier still: level 8 / 10 at L 20: 3.146 / 3.172 ms)
    const int ksplit = ksplit_env >= 0 ? ksplit_env : wcat_split_path ? std::max(1, std::min(2 * L / 5, L - 1)) : std::min(2, L - 1);
    // ... optionally a second part (the levels below the first part down to ksplit2, CLIORA_WGRAD_SPLIT_LEVEL2): three more launches
    // beside the inside chain's heaviest steps cost more than they take from the tail
    static const int ksplit2_env = [] { const char* e = getenv("CLIORA_WGRAD_SPLIT_LEVEL2"); return e ? atoi(e) : -1; }();
    const int ksplit2 = ksplit2_env > 0 ? ksplit2_env : -1;         // measured at c2: 3.32 (none) vs 3.37 (levels 2-5): off
    int tail_cells = C;                                         // chart rows per sentence still to be covered by the tail launch
    bool dpi_done_recorded = false;                             // ev_fork[2] marks "every row of dPI is final" on the caller's stream
    // d Wcat (+)= dPI^T IH over the cells [off, off + hi) of every sentence.  Split mode at d = 400 (round 4): one launch per projection
    // block on the eight-wave split-bf16 kernel with the rows remapped inside it; else the fp32 element-load kernel over all blocks.
    static const bool wcat_f32 = [] { const char* e = getenv("CLIORA_WCAT_GRAD"); return e && !strcmp(e, "f32"); }();
    static const int wcat_slices = [] { const char* e = getenv("CLIORA_WCAT_SLICES"); return e ? atoi(e) : 48; }();
    auto wcat_grad = [&](hipStream_t s_, int off, int hi, int accumulate) -> int {
        // (the kernel's row remap floors (r + 0.5) / hi in fp32: exact below 2^22 rows -- beyond, the element-load kernel)
        if (!wcat_f32 && tn_pairs_strided_ok(Dp) && (long long)B * hi >= 512 && (long long)B * hi < (1LL << 22)) {
            // all projection blocks in one launch + one reduction (CLIORA_WCAT_FUSED=0: a launch and a reduction per block, 48 slices each):
            // the three launches + three reductions of the low levels' part ended the step ~120 us after the inside chain.  Slices per block
            // (x 3 column blocks x nb workgroups), c2: 8 / 12 / 16 / 20 / 24 / 28 -> 3.17 / 3.14 / 3.11 / 3.11 / 3.11 / 3.15 ms (unfused 3.195)
            static const bool fused = [] { const char* e = getenv("CLIORA_WCAT_FUSED"); return !e || atoi(e) != 0; }();
            static const int fused_slices = [] { const char* e = getenv("CLIORA_WCAT_FUSED_SLICES"); return e ? atoi(e) : 0; }();
            if (fused)
                return launch_tn_level_block(s_, dPI, ldpi, 0, IH, B, C, off, hi, Dp, wb + bw.slab2, bw.slab_floats, wb + bw.gwcat, wb + bw.gbcat, accumulate,
                                             fused_slices > 0 ? fused_slices : std::max(8, (180 / (3 * nb)) / 4 * 4), nb);
            for (int blk = 0; blk < nb; ++blk)
                OKR(launch_tn_level_block(s_, dPI, ldpi, blk * Dp, IH, B, C, off, hi, Dp, wb + bw.slab2, bw.slab_floats,
                                          wb + bw.gwcat + (size_t)blk * Dp * Dp, wb + bw.gbcat + (size_t)blk * Dp, accumulate, wcat_slices));
            return CLIORA_OK;
        }
        return launch_tn(s_, B * hi, ldpi, Dp, Dp, LevelRowsA{dPI, ldpi, C, off, hi}, LevelRowsA{IH, Dp, C, off, hi}, wb + bw.slab2,
                         bw.slab_floats, wb + bw.gwcat, wb + bw.gbcat, accumulate);
    };
    auto inside_bwd_step = [&](int level) -> int {
        const LevelArgs g = level_args(p, level, false);        // N == 0 at the leaves
        const int ncell = B * g.Lc;
        const bool fused_norm = fuse_dnorm && p.share && !vl && level >= 1 && level <= L - 2;      // as in outside_bwd_step
        hipLaunchKernelGGL(cell_gather_bwd_in, dim3(ncell), dim3(256), 0, sa, g, D, d_inside_h,
                           level == 0 ? nullptr : d_inside_s, dv.use[ROLE_INA], dv.use[ROLE_INB], dv.use[ROLE_OUTA],
                           !ran_outside ? 0 : (sib_on_outside_chain(level) ? 1 : 2),
                           sibPL, sibQL, ldsib, wb + bw.sib_s, DA, DS, PI, ldpi, p.share, IH, OH, dPI, VH, dStot,
                           ws + f.bcat, fused_norm ? wb + bw.dots : nullptr);
        LAUNCHOK("cell_gather_bwd_in");
        if (level == 0) {       // dPI is complete: the low levels' share of the projections' weight gradient can start (GEMM stream)
            HIPOK(hipEventRecord(plan->ev_fork[2], sa));
            dpi_done_recorded = true;
        }
        if (fused_norm && gemm3_pays(ncell))
            OKR(launch_rows_direct3x(sa, ws + f.wcatT3s, ldpi, Dp, ncell, LevelRowsA{dPI, ldpi, C, g.off, g.Lc},
                              NormBwdLevelE{dG, VH, IH, ws + f.nrmi, wb + bw.dots, Dp, C, g.off, g.Lc, p.normalize}, bwd_gemm3));
        else if (fused_norm)
            OKR(launch_rows_direct(sa, ws + f.wcatT, PROJ_IMG(f.wcatT3), ldpi, Dp, ncell, LevelRowsA{dPI, ldpi, C, g.off, g.Lc},
                            NormBwdLevelE{dG, VH, IH, ws + f.nrmi, wb + bw.dots, Dp, C, g.off, g.Lc, p.normalize}));
        else if (level <= L - 2 && gemm3_pays(ncell))
            OKR(launch_rows_direct3x(sa, ws + f.wcatT3s, ldpi, Dp, ncell, LevelRowsA{dPI, ldpi, C, g.off, g.Lc},
                              StoreLevelE{VH, Dp, C, g.off, g.Lc, nullptr, 1}, bwd_gemm3));
        else if (level <= L - 2)
            OKR(launch_rows_direct(sa, ws + f.wcatT, PROJ_IMG(f.wcatT3), ldpi, Dp, ncell, LevelRowsA{dPI, ldpi, C, g.off, g.Lc},
                            StoreLevelE{VH, Dp, C, g.off, g.Lc, nullptr, 1}));
        else if (compress && ran_outside)       // d inside_h[root] += d root @ root_mat_out^T   (diora.py:342-343)
            OKR(launch_rows_direct(sa, ws + f.rootwT, PROJ_IMG(f.rootwT3), Dp, Dp, ncell, LevelRowsA{dGo, Dp, C, g.off, g.Lc},
                            StoreLevelE{VH, Dp, C, g.off, g.Lc, nullptr, 1}));
        if (vl) {
            // (levels >= 1: with the unit-norm backward through u = unit(g) at its end, cell_dnorm's launch of round 3)
            ATTEND_LAUNCH(cell_attend_bwd, p.R, dim3(ncell), sa, g, VH, IH, ws + f.nrmi, p.normalize, OBJ, p.R,
                               drop_mask, ws + f.att_pk, wb + bw.dctx, wb + bw.pmo, wb + bw.dsc, IHn, nrmIn, level >= 1 ? dG : (float*)nullptr);
            LAUNCHOK("cell_attend_bwd");
        }
        if ((level == ksplit || level == ksplit2) && level >= 1) {
            // cells per sentence at the levels from this one up to the part already on its way (a level's cells are contiguous)
            const int hi = tail_cells - g.off;
            HIPOK(hipEventRecord(plan->ev_fork[2], sa));
            HIPOK(hipStreamWaitEvent(sw, plan->ev_fork[2], 0));
            OKR(wcat_grad(sw, g.off, hi, tail_cells < C));
            tail_cells = g.off;
        }
        if (level == 0) return CLIORA_OK;
        if (!fused_norm && !vl) {
            hipLaunchKernelGGL(cell_dnorm, dim3(cells_grid(ncell)), dim3(256), 0, sa, g, VH, IHn, nrmIn, p.normalize, dG);
            LAUNCHOK("cell_dnorm(in)");
        }
        {
            ProfScope ps(CLIORA_KCLASS_COMPOSE_BWD, sa);
            OKR(launch_level_compose_bwd(sa, ws + f.w2iT, ws + f.w2iT3, f.S3, Dp, f.ct3, f.ncb3, pair_level(level, false), dG, YM, Pp, PI, ldpi,
                                         PI + Dp, ldpi, ws + f.b2i, DA, DZ, Xp, DPP, DPB, tiled));
        }
        hipLaunchKernelGGL(cell_dsoftmax, dim3(cells_grid(ncell)), dim3(256), 0, sa, g, f.ncb3, DPP, DPB, Sp, Pp, IS, dStot, DS);
        LAUNCHOK("cell_dsoftmax(in)");
        if (level == 1 && two_streams) HIPOK(hipEventRecord(plan->ev_join[2], sa));     // the last pair rows (DZ, X) are final
        return CLIORA_OK;
    };

    if (two_streams) {
        HIPOK(hipEventRecord(plan->ev_fork[0], st));
        HIPOK(hipStreamWaitEvent(sb, plan->ev_fork[0], 0));
        fork_guard.arm(0, sb, plan->ev_join[0]);
    }
    if (!ran_outside && !p.share)      // no outside pass: nobody writes the outside blocks of dPI (cell_gather_bwd_sib does otherwise)
        HIPOK(hipMemsetAsync(dPI, 0, (size_t)B * C * ldpi * sizeof(float), st));
    if (!ran_outside) {
        HIPOK(hipMemsetAsync(wb + bw.gw2o, 0, (size_t)Dp * Dp * sizeof(float), st));
        HIPOK(hipMemsetAsync(wb + bw.gb2o, 0, (size_t)Dp * sizeof(float), st));
        HIPOK(hipMemsetAsync(wb + bw.gw1ro, 0, (size_t)Dp * Dp * sizeof(float), st));
        HIPOK(hipMemsetAsync(wb + bw.groot, 0, (size_t)Dp * sizeof(float), st));
    }
    if (compress) {          // the whole outside backward (down to the per-sentence roots), then the inside backward
        for (int j = 0; j <= L - 1 && ran_outside; ++j) OKR(outside_bwd_step(j, nullptr));
        for (int j = 0; j <= L - 1; ++j) OKR(inside_bwd_step(L - 1 - j));
        if (ran_outside)     // d root_mat_out^T = sum_b d root[b]^T inside_h[b, root]
            OKR(launch_tn(st, B, Dp, Dp, Dp, LevelRowsA{dGo, Dp, C, C - 1, 1}, LevelRowsA{IH, Dp, C, C - 1, 1}, wb + bw.slab, bw.slab_floats,
                          wb + bw.groot_mat, (float*)nullptr));
        else
            HIPOK(hipMemsetAsync(wb + bw.groot_mat, 0, (size_t)Dp * Dp * sizeof(float), st));
    }
    // The pair rows' weight gradient, early part.  After step J the rows of outside levels 0..J and of inside levels L-1-J..L-1 are final,
    // and they are ONE contiguous range of the pair rows (the inside levels end where the outside levels begin): its dW2 runs on the GEMM
    // stream beside the last steps of the chains, whose kernels are small (the outside levels above J and the inside levels below L-1-J
    // have few pair rows), on a part of the chip; the tail then holds the two short end ranges only.
    // Measured on MI355X at d 400 / B 64 with J = (L-1)/2 and 32 slices (96 workgroups): L 20 3.519 -> 3.485 ms, L 30 9.39 -> 9.21,
    // L 40 19.88 -> 19.36, L 12 unchanged; later steps or wider launches lose (the part must end before the chains do, and it takes
    // from them nearly what it removes from the tail: profiles/r03_ubench_priority.txt).  CLIORA_WGRAD_EARLY_STEP = -1 turns it off.
    static const int early_env = [] { const char* e = getenv("CLIORA_WGRAD_EARLY_STEP"); return e ? atoi(e) : -2; }();
    static const int early_slices = [] { const char* e = getenv("CLIORA_WGRAD_EARLY_SLICES"); return e ? atoi(e) : 32; }();
    // round 4 (the two ends now run as ONE launch with one reduction, so a shorter tail pays more than it did): c2 J = 9 / 10 / 11 / 12:
    // 3.41 / 3.39 / 3.33 / 3.42 ms per step; L 40 J = 19 / 21 / 23: 19.25 / 19.22 / 19.60
    const int early_auto = L >= 16 ? L / 2 + 1 : -1;
    const int early_pick = early_env == -2 ? early_auto : early_env;
    const int J_early = (p.share && ran_outside && !compress && !resident && early_pick >= 0 && early_pick <= L - 3) ? early_pick : -1;
    static const int early2_env = [] { const char* e = getenv("CLIORA_WGRAD_EARLY_STEP2"); return e ? atoi(e) : -2; }();      // -1: off
    const int early2_auto = L >= 28 ? (J_early + L) / 2 : -1;
    const int early2_pick = early2_env == -2 ? early2_auto : early2_env;
    const int J_early2 = (J_early >= 0 && early2_pick > J_early && early2_pick <= L - 3) ? early2_pick : -1;
    long long early_r0 = 0, early_r1 = 0, early_t0 = 0, early_t1 = 0;       // ... as pair rows and as 16-row tiles
    if (resident) {
        // ---- both chains and the leaves' pre-activation gradient: one workgroup per sentence (resident_kernels.hpp) ----
        ResArgs a = resident_args(plan, ws, const_cast<float*>(IH), const_cast<float*>(OH), const_cast<float*>(IS), const_cast<float*>(OS), ran_outside);
        a.dIH = d_inside_h; a.dIS = d_inside_s; a.dOH = d_outside_h; a.dOS = d_outside_s;
        a.VHo = VHo; a.dPI = dPI; a.dPO = dPO; a.DA = DA; a.DS = DS; a.DZ = DZ; a.Xrows = Xp; a.dU = dU; a.dX = d_x_span;
        static const bool res_trace = [] { const char* e = getenv("CLIORA_RES_TRACE"); return e && atoi(e) != 0; }();
        a.trace = res_trace ? reinterpret_cast<unsigned long long*>(plan->trace_words) : nullptr;
        OKR(cliora_ensure_max_lds((const void*)resident_bwd));
        {
            ProfScope ps(CLIORA_KCLASS_COMPOSE_BWD, st);
            hipLaunchKernelGGL(resident_bwd, dim3(std::min(B, std::max(1, plan->ncu))), dim3(RES_THREADS), resident_lds_bytes(p), st, a);
            LAUNCHOK("resident_bwd");
        }
        if (ran_outside) {
            hipLaunchKernelGGL(root_bwd, dim3(1), dim3(ROOT_WAVES * 64), 0, st, B, C, Dp, VHo, OH, ws + f.nrmo, p.normalize, wb + bw.groot);
            LAUNCHOK("root_bwd");
        }
    }
    for (int j = 0; j <= L - 1 && !compress && !resident; ++j) {
        if (ran_outside) {
            // (the compress root step ends in a launch that takes no event; it never runs on two streams)
            const bool by_kernel = two_streams && stop_events_on();
            OKR(outside_bwd_step(j, by_kernel ? plan->ev_level[j] : nullptr));
            if (two_streams && !by_kernel) HIPOK(hipEventRecord(plan->ev_level[j], sb));
        }
        // the gather of inside level L-1-j reads the outside pairs whose sibling it is: outside levels <= j-1
        if (two_streams && j >= 1) HIPOK(hipStreamWaitEvent(sa, plan->ev_level[j - 1], 0));
        OKR(inside_bwd_step(L - 1 - j));
        if (j == J_early) {
            early_r0 = p.row_base_in(L - 1 - j);
            early_r1 = p.row_base_out(j + 1);
            HIPOK(hipEventRecord(plan->ev_fork[2], sa));
            HIPOK(hipStreamWaitEvent(sw, plan->ev_fork[2], 0));
            if (two_streams) HIPOK(hipStreamWaitEvent(sw, plan->ev_level[j], 0));
            early_t0 = p.tile_base_in(L - 1 - j);
            early_t1 = p.tile_base_out(j + 1);
            if (tiled)
                OKR(launch_tn_tiles(sw, DZ, Xp, early_t0, early_t1 - early_t0, 0, 0, Dp, wb + bw.slab2, bw.slab_floats, wb + bw.gw2o, wb + bw.gb2o, 0,
                                    early_slices));
            else
                OKR(launch_tn_pairs(sw, DZ + (size_t)early_r0 * Dp, Xp + (size_t)early_r0 * Dp, (int)(early_r1 - early_r0), Dp, wb + bw.slab2, bw.slab_floats,
                                    wb + bw.gw2o, wb + bw.gb2o, 0, early_slices));
        }
        // Long charts, tiled operands: a SECOND early part.  At L 40 the first part (the middle 60 % of the pair rows) is done a millisecond
        // before the chains are, the GEMM stream idles, and the tail -- the two ends, 0.5 ms with the whole chip -- then meets the chains' last
        // kernels (the level-0 projection backward took 450 us beside it instead of 40).  The rows that became final since the first part are two
        // ranges (inside levels L-1-J2 .. L-2-J, outside levels J+1 .. J2): one two-range launch, accumulated onto the first part's result.
        if (tiled && J_early >= 0 && j == J_early2) {
            const long long t0 = p.tile_base_in(L - 1 - j), t1 = p.tile_base_out(j + 1);
            HIPOK(hipEventRecord(plan->ev_fork[2], sa));
            HIPOK(hipStreamWaitEvent(sw, plan->ev_fork[2], 0));
            if (two_streams) HIPOK(hipStreamWaitEvent(sw, plan->ev_level[j], 0));
            OKR(launch_tn_tiles(sw, DZ, Xp, t0, early_t0 - t0, early_t1, t1 - early_t1, Dp, wb + bw.slab2, bw.slab_floats, wb + bw.gw2o, wb + bw.gb2o, 1,
                                early_slices));
            early_t0 = t0; early_t1 = t1;
            early_r0 = p.row_base_in(L - 1 - j); early_r1 = p.row_base_out(j + 1);
        }
    }
    // leaves
    if (!resident) {
        hipLaunchKernelGGL(leaf_bwd_pre, dim3(cells_grid(B * L)), dim3(256), 0, st, B, L, C, Dp, VH, IHn, nrmIn, p.normalize, ws + f.t, dU);
        LAUNCHOK("leaf_bwd_pre");
    }
    if (d_x_span && !resident)
        OKR(launch_rows_direct(st, ws + f.wlT, PROJ_IMG(f.wlT3), Dp, Dp, B * L, PlainRowsA{dU, Dp}, StoreRowsE{d_x_span, D, nullptr, 0, D}));

    // ---- weight gradients: the cell projections' and the leaf layer's on the GEMM stream, the pair rows' dW2 (tn_gemm_dma3 fills
    //      every CU's LDS) on the caller's stream once both chains are done ----
    HIPOK(hipEventRecord(plan->ev_fork[1], st));
    // The pair rows' tail (the rows whose dW2 did not start early).  Two streams (round 4): on the OUTSIDE chain's stream -- that chain
    // ends a step before the inside chain, and the last pair rows are final after inside level 1 (ev_join[2]), so the tail runs beside the
    // inside chain's last gather, projection backward and leaf layer instead of after them.  One stream: on the caller's stream, last.
    auto pair_rows_tail = [&](hipStream_t s_) -> int {
        ProfScope ps(CLIORA_KCLASS_WGRAD, s_);
        if (tiled) {
            if (ran_outside && !p.share)
                OKR(launch_tn_tiles(s_, DZ, Xp, p.T_in, p.T_out, 0, 0, Dp, wb + bw.slab, bw.slab_floats, wb + bw.gw2o, wb + bw.gb2o, 0, 0));
            const long long ntl = (p.share && ran_outside) ? p.T_in + p.T_out : p.T_in;
            static const int tail_slices2 = [] { const char* e = getenv("CLIORA_WGRAD_TAIL_SLICES2"); return e ? atoi(e) : 72; }();
            if (J_early >= 0)
                OKR(launch_tn_tiles(s_, DZ, Xp, 0, early_t0, early_t1, ntl - early_t1, Dp, wb + bw.slab, bw.slab_floats, wb + bw.gw2i, wb + bw.gb2i, 0,
                                    tail_slices2));
            else
                OKR(launch_tn_tiles(s_, DZ, Xp, 0, ntl, 0, 0, Dp, wb + bw.slab, bw.slab_floats, wb + bw.gw2i, wb + bw.gb2i, 0, 0));
            return CLIORA_OK;
        }
        if (ran_outside && !p.share)
            OKR(launch_tn_pairs(s_, DZ + (size_t)p.R_in * Dp, Xp + (size_t)p.R_in * Dp, (int)p.R_out, Dp, wb + bw.slab, bw.slab_floats,
                                wb + bw.gw2o, wb + bw.gb2o));
        // shared weights: inside and outside pair rows are one contiguous range -> one launch
        const long long nr = (p.share && ran_outside) ? p.R_in + p.R_out : p.R_in;
        if (J_early >= 0) {          // the middle of the range is on its way on the GEMM stream (into gw2o / gb2o): the two ends here
            static const int tail_slices = [] { const char* e = getenv("CLIORA_WGRAD_TAIL_SLICES"); return e ? atoi(e) : 48; }();
            static const int tail_slices2 = [] { const char* e = getenv("CLIORA_WGRAD_TAIL_SLICES2"); return e ? atoi(e) : 72; }();
            if (tn_pairs_two_ranges_ok(Dp))          // both ends in one launch (round 4): one slab, one reduction
                OKR(launch_tn_pairs_two_ranges(s_, DZ, Xp, (int)early_r0, early_r1, (int)(nr - early_r1), Dp, wb + bw.slab, bw.slab_floats,
                                               wb + bw.gw2i, wb + bw.gb2i, tail_slices2));
            else {
                OKR(launch_tn_pairs(s_, DZ, Xp, (int)early_r0, Dp, wb + bw.slab, bw.slab_floats, wb + bw.gw2i, wb + bw.gb2i, 0, tail_slices));
                OKR(launch_tn_pairs(s_, DZ + (size_t)early_r1 * Dp, Xp + (size_t)early_r1 * Dp, (int)(nr - early_r1), Dp, wb + bw.slab, bw.slab_floats,
                                    wb + bw.gw2i, wb + bw.gb2i, 1, tail_slices));
            }
        } else
            OKR(launch_tn_pairs(s_, DZ, Xp, (int)nr, Dp, wb + bw.slab, bw.slab_floats, wb + bw.gw2i, wb + bw.gb2i));
        // shared weights without the early part: gw2o / gb2o hold nothing of this call -- the scatter below leaves them out
        return CLIORA_OK;
    };
    if (ran_outside) {
        if (two_streams) {
            // d W1R_out = dPO^T OH needs the outside chain only: on ITS stream, behind its last kernel (round 4; on the GEMM stream it
            // queued behind the projections' weight gradient and ended the step) -- on the split-bf16 LDS-DMA kernel where there is one
            // (plain contiguous rows: 100 + 31 us of fp32 element-load GEMM + reduction -> one short launch)
            if (tn_pairs_strided_ok(Dp))
                OKR(launch_tn_pairs(sb, dPO, OH, B * C, Dp, wb + bw.slab, bw.slab_floats, wb + bw.gw1ro, (float*)nullptr, 0, 72));
            else
                OKR(launch_tn(sb, B * C, Dp, Dp, Dp, PlainRowsA{dPO, Dp}, PlainRowsA{OH, Dp}, wb + bw.slab, bw.slab_floats, wb + bw.gw1ro,
                              (float*)nullptr));
            if (L >= 2) HIPOK(hipStreamWaitEvent(sb, plan->ev_join[2], 0));
            OKR(pair_rows_tail(sb));
            HIPOK(hipEventRecord(plan->ev_join[0], sb));
        } else {           // one chain: the same kernel (the two schedules agree to the bit), on the GEMM stream
            HIPOK(hipStreamWaitEvent(sw, plan->ev_fork[1], 0));
            if (tn_pairs_strided_ok(Dp))
                OKR(launch_tn_pairs(sw, dPO, OH, B * C, Dp, wb + bw.slab2, bw.slab_floats, wb + bw.gw1ro, (float*)nullptr, 0, 72));
            else
                OKR(launch_tn(sw, B * C, Dp, Dp, Dp, PlainRowsA{dPO, Dp}, PlainRowsA{OH, Dp}, wb + bw.slab2, bw.slab_floats, wb + bw.gw1ro,
                              (float*)nullptr));
        }
    }
    if (dpi_done_recorded) {     // the low levels' d Wcat needs the last gather only, not the leaf layer's backward behind it
        HIPOK(hipStreamWaitEvent(sw, plan->ev_fork[2], 0));
        OKR(wcat_grad(sw, 0, tail_cells, tail_cells < C));
    }
    HIPOK(hipStreamWaitEvent(sw, plan->ev_fork[1], 0));
    if (vl && d_obj_span) {      // d obj: independent of the weight gradients, so beside the pair rows' tail on the GEMM stream
        float* dO = padded ? wb + bw.dobjp : d_obj_span;
        if (Dp > 512) return fail(CLIORA_EINVAL, "obj_grad_reduce: D > 512 is not supported");
        if (p.R <= 36) {
            const size_t lds = (size_t)3 * 2 * 9 * 64 * sizeof(float4);
            OKR(cliora_ensure_max_lds((const void*)obj_grad_reduce<9>));
            hipLaunchKernelGGL(obj_grad_reduce<9>, dim3(B, 4), dim3(512), lds, sw, B, C, Dp, p.R, wb + bw.dctx, ws + f.att_u, wb + bw.pmo, wb + bw.dsc, dO);
        } else {
            const size_t lds = (size_t)3 * 2 * 16 * 64 * sizeof(float4);
            OKR(cliora_ensure_max_lds((const void*)obj_grad_reduce<16>));
            hipLaunchKernelGGL(obj_grad_reduce<16>, dim3(B, 4), dim3(512), lds, sw, B, C, Dp, p.R, wb + bw.dctx, ws + f.att_u, wb + bw.pmo, wb + bw.dsc, dO);
        }
        LAUNCHOK("obj_grad_reduce");
    }
    if (!dpi_done_recorded) OKR(wcat_grad(sw, 0, tail_cells, tail_cells < C));
    OKR(launch_tn(sw, B * L, Dp, Dp, Dp, PlainRowsA{dU, Dp}, PlainRowsA{X, Dp}, wb + bw.slab2, bw.slab_floats, wb + bw.gwl, wb + bw.gbl));
    HIPOK(hipEventRecord(plan->ev_join[1], sw));
    if (!two_streams) OKR(pair_rows_tail(st));
    else HIPOK(hipStreamWaitEvent(st, plan->ev_join[0], 0));
    HIPOK(hipStreamWaitEvent(st, plan->ev_join[1], 0));
    fork_guard.disarm();                                   // both side streams have been joined above

    // ---- scatter packed gradients back to the reference parameter shapes ----
    {
        CopyTable t; t.n = 0;
        const size_t DD = (size_t)Dp * Dp;
        if (vl && d_obj_span && padded) add_copy(t, d_obj_span, D, B * p.R, D, wb + bw.dobjp, Dp, B * p.R, D, 0, 0, 0);
        if (G->leaf_w) add_copy(t, G->leaf_w, D, D, D, wb + bw.gwl, Dp, D, D, 0, 0, 0);
        if (G->leaf_b) add_copy(t, G->leaf_b, D, 1, D, wb + bw.gbl, Dp, 1, D, 0, 0, 0);
        if (G->root_h && !compress) add_copy(t, G->root_h, D, 1, D, wb + bw.groot, Dp, 1, D, 0, 0, 0);
        if (G->root_mat && compress) add_copy(t, G->root_mat, D, D, D, wb + bw.groot_mat, Dp, D, D, 0, 0, 1);
        if (p.share) {
            if (G->in_w1) {
                add_copy(t, G->in_w1, 2 * D, D, D, wb + bw.gwcat, Dp, D, D, 0, 0, 0);
                add_copy(t, G->in_w1 + D, 2 * D, D, D, wb + bw.gwcat + DD, Dp, D, D, 0, 0, 0, wb + bw.gw1ro, Dp, D, D, 0, 0, 0);
            }
            if (G->in_b1) add_copy(t, G->in_b1, D, 1, D, wb + bw.gbcat, Dp, 1, D, 0, 0, 0);
            if (G->in_mat) add_copy(t, G->in_mat, D, D, D, wb + bw.gwcat + 2 * DD, Dp, D, D, 0, 0, 1);
            const float* gw2o_part = J_early >= 0 ? wb + bw.gw2o : nullptr;      // the early middle range's share (else: not written)
            const float* gb2o_part = J_early >= 0 ? wb + bw.gb2o : nullptr;
            if (G->in_w2) add_copy(t, G->in_w2, D, D, D, wb + bw.gw2i, Dp, D, D, 0, 0, 0, gw2o_part, Dp, D, D, 0, 0, 0);
            if (G->in_b2) add_copy(t, G->in_b2, D, 1, D, wb + bw.gb2i, Dp, 1, D, 0, 0, 0, gb2o_part, Dp, 1, D, 0, 0, 0);
        } else {
            if (G->in_w1) {
                add_copy(t, G->in_w1, 2 * D, D, D, wb + bw.gwcat, Dp, D, D, 0, 0, 0);
                add_copy(t, G->in_w1 + D, 2 * D, D, D, wb + bw.gwcat + DD, Dp, D, D, 0, 0, 0);
            }
            if (G->in_b1) add_copy(t, G->in_b1, D, 1, D, wb + bw.gbcat, Dp, 1, D, 0, 0, 0);
            if (G->in_mat) add_copy(t, G->in_mat, D, D, D, wb + bw.gwcat + 2 * DD, Dp, D, D, 0, 0, 1);
            if (G->in_w2) add_copy(t, G->in_w2, D, D, D, wb + bw.gw2i, Dp, D, D, 0, 0, 0);
            if (G->in_b2) add_copy(t, G->in_b2, D, 1, D, wb + bw.gb2i, Dp, 1, D, 0, 0, 0);
            if (G->out_w1) {
                add_copy(t, G->out_w1, 2 * D, D, D, wb + bw.gwcat + 3 * DD, Dp, D, D, 0, 0, 0);
                add_copy(t, G->out_w1 + D, 2 * D, D, D, wb + bw.gw1ro, Dp, D, D, 0, 0, 0);
            }
            if (G->out_b1) add_copy(t, G->out_b1, D, 1, D, wb + bw.gbcat + 3 * Dp, Dp, 1, D, 0, 0, 0);
            if (G->out_mat) add_copy(t, G->out_mat, D, D, D, wb + bw.gwcat + 4 * DD, Dp, D, D, 0, 0, 1);
            if (G->out_w2) add_copy(t, G->out_w2, D, D, D, wb + bw.gw2o, Dp, D, D, 0, 0, 0);
            if (G->out_b2) add_copy(t, G->out_b2, D, 1, D, wb + bw.gb2o, Dp, 1, D, 0, 0, 0);
        }
        OKR(run_copies(st, t));
    }
    return CLIORA_OK;
}
