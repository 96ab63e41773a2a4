#include "plan.hpp"

#include <algorithm>

namespace cliora {

static inline int ncells(int L) { return L * (L + 1) / 2; }

static size_t align64(size_t x) { return (x + 63) & ~size_t(63); }

// Measured on MI355X (round 2, rocprof per-level traces): a round of tasks costs a ~5.8 us latency chain (index loads, ring
// fill, epilogue, reduction) plus ~2.8 us of MFMA / VALU / LDS issue per tile on the busiest SIMD (waves w and w+4 share
// one); rounds do not overlap.  Pick the cheapest geometry, fewer parts on a tie.
ComposeGeom compose_geom(int ncell, int N, int cap) {
    const int G = (ncell + 15) / 16;
    ComposeGeom best{1, 1, G};
    double best_t = 1e30;
    for (int TG : {1, 2, 4, 8}) {
        const int wpg = 8 / TG;
        const int groups = (G + TG - 1) / TG;
        for (int SP = 1; SP <= HP_PARTS && SP <= std::max(1, N); SP *= 2) {
            const int np = (N + SP - 1) / SP;
            const int ntask = groups * SP;
            const int rounds = (ntask + cap - 1) / cap;
            const int depth = (np + wpg - 1) / wpg;
            const int busy = TG * std::min(wpg, np);
            int simd = depth * (busy > 4 ? 2 : 1);
            if (busy > 4 && wpg == 8 && np % 8 != 0 && np % 8 <= 4) simd -= 1;     // the last sweep reaches one wave of each pair only
            const double t = rounds * (5.8 + 2.8 * simd) + 0.3 * (SP - 1);
            if (t < best_t - 1e-9) { best_t = t; best = ComposeGeom{TG, SP, ntask}; }
        }
    }
    return best;
}

// Compose workgroups per column block that level `lv` of a pass (0 inside, 1 outside) may count on: step k of the forward composes
// inside level k and outside level L-k side by side, so each gets its share of compose_cap -- by pair rows
// (shared weights: any workgroup serves either pass), or the half that holds its weight image (unshared).
int compose_cap_share(const Plan& p, int pass, int lv) {
    const int L = p.L;
    const int k = pass ? L - lv : lv;                      // the step this level runs in
    const bool has_in = k >= 1 && k <= L - 1, has_out = k >= 2 && k <= L;
    int cap = p.compose_cap;
    if (has_in && has_out && cap >= 2) {
        if (!p.share) cap = cap / 2;
        else {
            const double rows_in = (double)(L - k) * k, rows_out = (double)k * (k - 1);
            const int cap_in = std::min(cap - 1, std::max(1, (int)(cap * rows_in / (rows_in + rows_out) + 0.5)));
            cap = pass ? cap - cap_in : cap_in;
        }
    }
    return cap;
}

std::string build_plan(Plan& p, int B, int L, int D, int share, int normalize, int R, int arch) {
    if (B < 1 || L < 1 || D < 1) return "B, L, D must be positive";
    if (D > 512) return "D > 512 is not supported by the weight-stationary kernels";
    if (L > CLIORA_MAX_L) return "L > 64 is not supported (one split per lane in the score kernels)";
    if (R < 0 || R > 64) return "R (image regions) must be in [0, 64]";
    if (normalize != 0 && normalize != 1) return "normalize must be 0 (none) or 1 (unit)";
    p.B = B; p.L = L; p.D = D; p.Dp = (D + 15) / 16 * 16; p.C = ncells(L);
    p.share = share ? 1 : 0; p.normalize = normalize; p.R = R;
    if (arch != 0 && arch != 1) return "arch must be 0 (MLP) or 1 (TreeLSTM)";
    if (arch == 1 && R != 0) return "TreeLSTM plans are text-only (R = 0)";
    p.arch = arch;
    if (arch == 0) {
        p.nblk = p.share ? 3 : 5; p.npo = 1; p.nleaf = 1; p.off_pr = 1; p.off_ql = 2;
        p.blk_plo = p.share ? 0 : 3;
        p.blk_qlo = p.share ? 2 : 4;
    } else {   // [PL (5 gates: u,i,o,f0,f1) | PR (5) | QL] and, with unshared outside functions, [PLo (5) | QLo] of the outside weights
        p.nblk = p.share ? 11 : 17; p.npo = 5; p.nleaf = 3; p.off_pr = 5; p.off_ql = 10;
        p.blk_plo = p.share ? 0 : 11; p.blk_qlo = p.share ? 10 : 16;
    }
    p.P_in = (L - 1) * L * (L + 1) / 6;
    p.P_out = (L - 1) * L * (L + 1) / 3;
    p.R_in = (long long)B * p.P_in;
    p.R_out = (long long)B * p.P_out;
    if (p.R_in + p.R_out > 0x7fffffffLL / 2) return "batch too large for 32-bit pair rows";
    p.tile_base_in_.assign(L + 1, 0); p.tile_base_out_.assign(L + 1, 0);
    for (int lv = 0; lv < L; ++lv) {
        const long long g16 = ((long long)B * (L - lv) + 15) / 16;
        p.tile_base_in_[lv + 1] = (int32_t)(p.tile_base_in_[lv] + g16 * lv);
        p.tile_base_out_[lv + 1] = (int32_t)(p.tile_base_out_[lv] + g16 * (L - lv - 1));
    }
    p.T_in = p.tile_base_in_[L]; p.T_out = p.tile_base_out_[L];

    const int C = p.C;
    p.level_offset.resize(L);
    for (int lv = 0; lv < L; ++lv) p.level_offset[lv] = C - ncells(L - lv);
    auto cell = [&](int level, int pos) { return p.level_offset[level] + pos; };

    // ---- per-level pair tables -------------------------------------------------
    p.lvl_base_in.assign(L, 0);
    p.lvl_base_out.assign(L, 0);
    p.pair_a_in.clear(); p.pair_b_in.clear(); p.pair_a_out.clear(); p.pair_b_out.clear();
    std::vector<std::vector<std::vector<int32_t>>> tmp(N_ROLES, std::vector<std::vector<int32_t>>(C));
    // each use is pushed as three ints: row, stride, partner
    auto push_use = [&](int role, int c, long long row, int stride, int partner) {
        tmp[role][c].push_back((int32_t)row);
        tmp[role][c].push_back(stride);
        tmp[role][c].push_back(partner);
    };
    int acc = 0;
    for (int lv = 1; lv < L; ++lv) {          // inside: target (lv, pos), split n: left (n,pos), right (lv-n-1, pos+n+1)
        p.lvl_base_in[lv] = acc;
        const int Lc = L - lv, N = lv;
        for (int pos = 0; pos < Lc; ++pos)
            for (int n = 0; n < N; ++n) {
                const int a = cell(n, pos), b = cell(lv - n - 1, pos + n + 1);
                p.pair_a_in.push_back(a);
                p.pair_b_in.push_back(b);
                const long long row = (long long)B * acc + (pos * N + n);
                push_use(ROLE_INA, a, row, Lc * N, b);
                push_use(ROLE_INB, b, row, Lc * N, a);
            }
        acc += Lc * N;
    }
    acc = 0;
    for (int lv = 0; lv + 1 < L; ++lv) {      // outside: target (lv, pos); our split order: parents that START left of
        p.lvl_base_out[lv] = acc;             // the target first (target = right child), then parents that END right of it
        const int Lc = L - lv, N = L - lv - 1;
        for (int pos = 0; pos < Lc; ++pos)
            for (int n = 0; n < N; ++n) {
                int sib, par;
                if (n < pos) {                 // parent [q, pos+lv], sibling [q, pos-1]
                    const int q = n;
                    par = cell(pos + lv - q, q);
                    sib = cell(pos - 1 - q, q);
                } else {                       // parent [pos, r], sibling [pos+lv+1, r]
                    const int r = pos + lv + 1 + (n - pos);
                    par = cell(r - pos, pos);
                    sib = cell(r - pos - lv - 1, pos + lv + 1);
                }
                p.pair_a_out.push_back(sib);
                p.pair_b_out.push_back(par);
                const long long row = p.R_in + (long long)B * acc + (pos * N + n);
                push_use(ROLE_OUTA, sib, row, Lc * N, par);
                push_use(ROLE_OUTB, par, row, Lc * N, sib);
            }
        acc += Lc * N;
    }
    for (int r = 0; r < N_ROLES; ++r) {
        UseList& u = p.uses[r];
        u.off.assign(C + 1, 0);
        u.row.clear(); u.stride.clear(); u.partner.clear();
        for (int c = 0; c < C; ++c) {
            const auto& v = tmp[r][c];
            for (size_t i = 0; i < v.size(); i += 3) {
                u.row.push_back(v[i]); u.stride.push_back(v[i + 1]); u.partner.push_back(v[i + 2]);
            }
            u.off[c + 1] = (int32_t)u.row.size();
        }
    }

    // ---- workspace layouts -----------------------------------------------------
    const size_t Dp = p.Dp, nb = p.nblk, BC = (size_t)B * C, BL = (size_t)B * L;
    const size_t npo = p.npo, nlf = p.nleaf, lstm = arch == 1 ? 1 : 0;
    const size_t Rt = (size_t)(p.R_in + p.R_out);
    const bool padded = (p.D != p.Dp);
    {
        FwdLayout& f = p.fwd;
        size_t o = 0;
        auto take = [&](size_t n) { size_t at = o; o = align64(o + n); return at; };
        f.wl = take(nlf * Dp * Dp); f.bl = take(nlf * Dp); f.wlT = take(nlf * Dp * Dp);
        f.wcat = take(nb * Dp * Dp); f.bcat = take(nb * Dp); f.wcatT = take(nb * Dp * Dp);
        f.w1ro = take(npo * Dp * Dp); f.w1roT = take(npo * Dp * Dp);
        f.w2i = take(Dp * Dp); f.b2i = take(Dp); f.w2iT = take(Dp * Dp);
        if (p.share) { f.w2o = f.w2i; f.b2o = f.b2i; f.w2oT = f.w2iT; }
        else { f.w2o = take(Dp * Dp); f.b2o = take(Dp); f.w2oT = take(Dp * Dp); }
        f.Kp3 = (int)((Dp + 31) / 32 * 32); f.S3 = f.Kp3 + 8;
        f.w2i3 = take(Dp * f.S3); f.w2iT3 = take(Dp * f.S3);
        if (p.share) { f.w2o3 = f.w2i3; f.w2oT3 = f.w2iT3; }
        else { f.w2o3 = take(Dp * f.S3); f.w2oT3 = take(Dp * f.S3); }
        auto img = [&](size_t rows, size_t K) { return take(rows * ((K + 31) / 32 * 32 + 8)); };
        f.wl3 = img(nlf * Dp, Dp); f.wlT3 = img(Dp, nlf * Dp);
        f.wcat3 = img(nb * Dp, Dp); f.wcatT3 = img(Dp, nb * Dp);
        f.w1ro3 = img(npo * Dp, Dp); f.w1roT3 = img(Dp, npo * Dp);
        {   // frag_weight_image3: output columns x roundup32(K) dwords (TreeLSTM plans only)
            auto img3 = [&](size_t cols, size_t K) { return take(cols * ((K + 31) / 32 * 32)); };     // (DioraMLP: rows_gemm_ksplit3x)
            f.wcat3s = img3(nb * Dp, Dp); f.wcatT3s = img3(Dp, nb * Dp); f.w1ro3s = img3(npo * Dp, Dp); f.w1roT3s = img3(Dp, npo * Dp);
        }
        f.rootp = take(Dp);
        f.matp = take(arch == 0 ? Dp * Dp : 0); f.matq3 = take(arch == 0 ? Dp * Dp : 0); f.qrleaf = take(arch == 0 ? BL * Dp : 0);
        f.xp = take(padded ? BL * Dp : 0);
        f.ihp = take(padded ? BC * Dp : 0);
        f.ohp = take(padded ? BC * Dp : 0);
        f.objp = take((padded && R > 0) ? (size_t)B * R * Dp : 0);
        f.t = take(BL * nlf * Dp);
        f.pi = take(BC * nb * Dp);
        f.po = take(BC * npo * Dp);
        f.y = take(lstm * Rt * Dp);
        f.x = take(lstm * Rt * Dp);
        f.sp = take(Rt);
        f.pp = take(Rt);
        {   // column block of the weight-stationary compose kernels: the most 16-column tiles whose split-bf16 image
            // (+ the cross-wave reduction slots) fits the CU's 160 KiB of LDS
            const int nt = (int)(Dp / 16);
            f.ct3 = 1;
            for (int ct : {5, 4, 2, 1})
                if (nt % ct == 0 && (size_t)ct * 16 * f.S3 * 4 + (size_t)4 * ct * 1024 <= 160 * 1024) { f.ct3 = ct; break; }
            f.ncb3 = nt / f.ct3;
        }
        f.hp = take(arch == 0 ? (size_t)HP_PARTS * BC * Dp : 0);
        f.hp_o = take(arch == 0 ? (size_t)HP_PARTS * BC * Dp : 0);
        f.ymask = take(arch == 0 ? Rt * f.ncb3 * 4 : 0);
        f.nrmi = take(BC);
        f.nrmo = take(BC);
        f.icp = take(lstm * BC * Dp); f.ocp = take(lstm * BC * Dp);
        f.nrmic = take(lstm * BC); f.nrmoc = take(lstm * BC); f.rootc = take(lstm * Dp);
        f.att_u = take(R > 0 ? BC * Dp : 0);
        f.att_pk = take(R > 0 ? BC * 64 : 0);
        f.att_nrmu = take(R > 0 ? BC : 0);
        f.rootw = take(arch == 0 ? Dp * Dp : 0); f.rootwT = take(arch == 0 ? Dp * Dp : 0);
        f.rootw3 = arch == 0 ? img(Dp, Dp) : o; f.rootwT3 = arch == 0 ? img(Dp, Dp) : o;
        f.rootpb = take(arch == 0 ? (size_t)B * Dp : 0);
        f.total = o;
        // per-pair compose outputs for the hooks: the TreeLSTM keeps them anyway (y rows), DioraMLP writes them into an
        // optional tail of the workspace only when a hook is overridden
        f.pair_h = arch == 0 ? o : f.y;
        f.pair_h_floats = arch == 0 ? align64(Rt * Dp) : 0;
    }
    {
        BwdLayout& b = p.bwd;
        size_t o = 0;
        auto take = [&](size_t n) { size_t at = o; o = align64(o + n); return at; };
        b.vh = take(BC * Dp); b.dg = take(BC * Dp); b.dstot = take(BC);
        b.vh_o = take(BC * Dp); b.dg_o = take(BC * Dp); b.dstot_o = take(BC);
        b.vc_o = take(lstm * BC * Dp); b.dgc_o = take(lstm * BC * Dp);
        // TreeLSTM keeps no per-pair gradient rows: its cell-centric backward recomputes them per use (lstm_kernels.hpp)
        b.da = take(lstm ? 0 : Rt * Dp); b.ds = take(Rt);
        // (DioraMLP: both also hold the tiled form, whole 16-row tiles per level and split)
        const size_t Rtile = std::max(Rt, (size_t)(p.T_in + p.T_out) * 16);
        b.dz = take(lstm ? 0 : Rtile * Dp);
        b.x = take(arch == 0 ? Rtile * Dp : 0);
        b.dpp = take(arch == 0 ? Rt * p.fwd.ncb3 : 0);
        b.dpb = take(arch == 0 ? Rt : 0);
        b.dcb = take(0); b.vc = take(lstm * BC * Dp); b.dgc = take(lstm * BC * Dp); b.grootc = take(lstm * Dp);
        b.dpi = take(BC * nb * Dp); b.dpo = take(BC * npo * Dp);
        b.dots = take(arch == 0 ? BC : 0); b.dots_o = take(arch == 0 ? BC : 0);
        b.sib_pl = take((arch == 0 && p.share) ? BC * Dp : 0); b.sib_ql = take((arch == 0 && p.share) ? BC * Dp : 0); b.sib_s = take(arch == 0 ? BC : 0);
        b.du = take(BL * nlf * Dp); b.dxp = take(padded ? BL * Dp : 0);
        // split-K slabs: at most 1024 wave-sized partial blocks of 80x80 per weight-gradient GEMM
        b.slab_floats = std::max((size_t)2048 * 80 * 80, (size_t)128 * (Dp * Dp + Dp));
        b.slab = take(b.slab_floats);
        b.slab2 = take(arch == 0 ? b.slab_floats : 0);
        b.gwcat = take(nb * Dp * Dp); b.gbcat = take(nb * Dp); b.gw1ro = take(npo * Dp * Dp);
        b.gw2i = take(Dp * Dp); b.gb2i = take(Dp); b.gw2o = take(Dp * Dp); b.gb2o = take(Dp);
        b.gwl = take(nlf * Dp * Dp); b.gbl = take(nlf * Dp); b.groot = take(Dp);
        b.dctx = take(R > 0 ? BC * Dp : 0);
        b.pmo = take(R > 0 ? BC * 64 : 0);
        b.dsc = take(R > 0 ? BC * 64 : 0);
        b.dobjp = take(R > 0 ? (size_t)B * R * Dp : 0);
        b.groot_mat = take(arch == 0 ? Dp * Dp : 0);
        b.total = o;
    }
    {
        auto& v = p.vl;
        size_t o = 0;
        auto take = [&](size_t n) { size_t at = o; o = align64(o + n); return at; };
        v.NRp = ((B * R + 15) / 16) * 16;
        const size_t NR = (size_t)v.NRp;
        const size_t on = R > 0 ? 1 : 0;
        v.oall = take(on * NR * Dp); v.oallT = take(on * NR * Dp);
        v.wall = take(on * NR * Dp); v.wallT = take(on * NR * Dp);
        v.sump = take(on * BC * Dp); v.xwp = take(on * BL * Dp); v.xwn = take(on * BL * Dp); v.dxn = take(on * BL * Dp); v.nrm = take(on * 4 * BL);
        v.gobj = take(on * NR * Dp);
        v.oimg = take(on * NR * ((Dp + 31) / 32 * 32 + 8));      // split-bf16 image of the region matrix (the scorers' weight in the default mode)
        v.keys = take(on * 2 * (size_t)B * BC);          // (B,B,C) 64-bit (score, region) keys of the region-max scorer
        v.slab_floats = on * (8 * (NR * Dp) + 64);
        v.slab = take(v.slab_floats);
        v.total = o;
    }
    // ---- per-level shapes and compose geometry (one entry per pass and level) ----
    // sized for one workgroup per CU of a 256-CU part, the column blocks of a task on one XCD: ((256 / 8) / ncb) * 8 per block
    p.compose_cap = std::max(1, (32 / std::max(1, p.fwd.ncb3))) * 8;
    if (p.fwd.ncb3 > 32) p.compose_cap = std::max(1, 256 / p.fwd.ncb3);
    p.level_geom.assign((size_t)2 * L * PLEVEL_INTS, 0);
    // Step k of the forward composes inside level k and outside level L-k side by side: each level's geometry is sized for its share of the compose workgroups, by pair rows
    // (shared weights), or for the half that holds its weight image (unshared).
    for (int pass = 0; pass < 2; ++pass)
        for (int lv = 0; lv < L; ++lv) {
            int32_t* e = p.level_geom.data() + ((size_t)pass * L + lv) * PLEVEL_INTS;
            const int Lc = L - lv, N = pass ? L - lv - 1 : lv;
            e[0] = Lc; e[1] = N; e[2] = p.level_offset[lv];
            e[3] = (int32_t)(pass ? p.row_base_out(lv) : p.row_base_in(lv));
            e[4] = pass ? p.lvl_base_out[lv] : p.lvl_base_in[lv];
            const ComposeGeom q = compose_geom(B * Lc, N, compose_cap_share(p, pass, lv));
            e[5] = q.TG; e[6] = q.SP; e[7] = q.ntask;
        }
    return "";
}

// Batch-expanded operand / target chart rows of every pair row.  Only the TreeLSTM kernels read them on the device (the DioraMLP
// kernels index the per-sentence pair tables), so they are built on demand: at the TreeLSTM upload, or when a caller asks for the
// tables by name (tests).  3 x (R_in + R_out) ints: 25 MB at B 64 / L 40.
void build_row_maps(Plan& p) {
    const size_t Rt = (size_t)(p.R_in + p.R_out);
    if (p.arow.size() == Rt) return;
    const int B = p.B, L = p.L, C = p.C;
    auto cell = [&](int level, int pos) { return p.level_offset[level] + pos; };
    p.arow.resize(Rt); p.brow.resize(Rt); p.trow.resize(Rt);
    for (int pass = 0; pass < 2; ++pass) {
        const auto& ta = pass ? p.pair_a_out : p.pair_a_in;
        const auto& tb = pass ? p.pair_b_out : p.pair_b_in;
        const auto& base = pass ? p.lvl_base_out : p.lvl_base_in;
        const int lv0 = pass ? 0 : 1, lv1 = pass ? L - 1 : L;
        for (int lv = lv0; lv < lv1; ++lv) {
            const int Lc = L - lv, N = pass ? L - lv - 1 : lv;
            const long long rb = pass ? p.row_base_out(lv) : p.row_base_in(lv);
            for (int b = 0; b < B; ++b)
                for (int pos = 0; pos < Lc; ++pos)
                    for (int n = 0; n < N; ++n) {
                        const int loc = pos * N + n;
                        const size_t r = (size_t)(rb + (long long)b * Lc * N + loc);
                        p.arow[r] = b * C + ta[base[lv] + loc];
                        p.brow[r] = b * C + tb[base[lv] + loc];
                        p.trow[r] = b * C + cell(lv, pos);
                    }
        }
    }
}

std::vector<int32_t> flatten_tables(Plan& p) {
    std::vector<int32_t> flat;
    auto put = [&](const std::vector<int32_t>& v) {
        size_t at = flat.size();
        flat.insert(flat.end(), v.begin(), v.end());
        while (flat.size() % 4) flat.push_back(0);
        return at;
    };
    p.dev.pair_a_in = put(p.pair_a_in); p.dev.pair_b_in = put(p.pair_b_in);
    p.dev.pair_a_out = put(p.pair_a_out); p.dev.pair_b_out = put(p.pair_b_out);
    for (int r = 0; r < N_ROLES; ++r) {
        p.dev.use_off[r] = put(p.uses[r].off);
        p.dev.use_row[r] = put(p.uses[r].row);
        p.dev.use_stride[r] = put(p.uses[r].stride);
        p.dev.use_partner[r] = put(p.uses[r].partner);
    }
    p.dev.lvl_base_in = put(p.lvl_base_in);
    p.dev.lvl_base_out = put(p.lvl_base_out);
    p.dev.level_geom = put(p.level_geom);
    if (p.arch == 1) build_row_maps(p);
    p.dev.arow = put(p.arow); p.dev.brow = put(p.brow); p.dev.trow = put(p.trow);
    if (flat.empty()) flat.push_back(0);
    return flat;
}

const std::vector<int32_t>* find_table(Plan& p, const std::string& name) {
    if (name == "level_offset") return &p.level_offset;
    if (name == "pair_a_in") return &p.pair_a_in;
    if (name == "pair_b_in") return &p.pair_b_in;
    if (name == "pair_a_out") return &p.pair_a_out;
    if (name == "pair_b_out") return &p.pair_b_out;
    if (name == "pair_lvl_base_in") return &p.lvl_base_in;
    if (name == "pair_lvl_base_out") return &p.lvl_base_out;
    if (name == "level_geom") return &p.level_geom;
    if (name == "tile_base_in") return &p.tile_base_in_;          // 16-row tiles of the pair rows (wgrad_tiles.hpp): first tile of a level, per pass
    if (name == "tile_base_out") return &p.tile_base_out_;
    if (name == "arow" || name == "brow" || name == "trow") build_row_maps(p);
    if (name == "arow") return &p.arow;
    if (name == "brow") return &p.brow;
    if (name == "trow") return &p.trow;
    static const char* roles[N_ROLES] = {"ina", "inb", "outa", "outb"};
    for (int r = 0; r < N_ROLES; ++r) {
        const std::string s = roles[r];
        if (name == "use_off_" + s) return &p.uses[r].off;
        if (name == "use_row_" + s) return &p.uses[r].row;
        if (name == "use_stride_" + s) return &p.uses[r].stride;
        if (name == "use_partner_" + s) return &p.uses[r].partner;
    }
    return nullptr;
}

}  // namespace cliora
