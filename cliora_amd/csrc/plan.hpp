// Chart plan: host-side index tables, use lists and workspace layout for one
// (B, L, D, share, normalize, R) chart shape.  Pure C++ (no HIP calls) so that
// it can be exercised on a box without a GPU.
//
// Reference behaviour restated here (tables only, closed form):
//   cliora/net/offset_cache.py:1-7        level offsets
//   cliora/net/inside_index.py:131-197    inside (left,right) per split
//   cliora/net/outside_index.py:39-62     outside (parent,sibling) per split
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace cliora {

constexpr int ROLE_INA = 0;   // inside cell used as LEFT child in the inside pass
constexpr int ROLE_INB = 1;   // inside cell used as RIGHT child in the inside pass
constexpr int ROLE_OUTA = 2;  // inside cell used as SIBLING in the outside pass
constexpr int ROLE_OUTB = 3;  // outside cell used as PARENT in the outside pass
constexpr int N_ROLES = 4;
constexpr int HP_PARTS = 4;        // a cell's split range is cut into at most this many parts (level_compose_fwd tasks)
constexpr int CLIORA_MAX_L = 64;   // sentence length bound: one split per lane in the score kernels
constexpr int PLEVEL_INTS = 8;     // ints per entry of Plan::level_geom {Lc, N, level offset, pair-row base, pair-table base, TG, SP, ntask}

struct UseList {
    std::vector<int32_t> off;      // C+1, CSR offsets per cell
    std::vector<int32_t> row;      // batch-independent part of the pair row
    std::vector<int32_t> stride;   // per-sentence row stride of that level (Lc*N)
    std::vector<int32_t> partner;  // the other cell of the pair
};

// Offsets are in floats from the start of the workspace; every region starts 64-float aligned.
struct FwdLayout {
    size_t wl, bl, wlT;                 // leaf fc (Dp x Dp), bias, transpose
    size_t wcat, bcat, wcatT;           // inside-cell projection: (nblk*Dp x Dp), bias, transpose (Dp x nblk*Dp)
    size_t w1ro, w1roT;                 // outside-cell projection W1[:, D:] of the outside compose
    size_t w2i, b2i, w2iT;              // second compose layer, inside weights
    size_t w2o, b2o, w2oT;              // outside weights (alias of inside when shared)
    size_t w2i3, w2iT3, w2o3, w2oT3;    // split-bf16 LDS images of the four (Dp rows x S3 dwords each; see split_weight_image)
    int Kp3, S3;                        // image geometry for K = Dp: k rounded up to 32, row stride in dwords
    size_t wl3, wlT3, wcat3, wcatT3, w1ro3, w1roT3;   // images of the leaf / projection weights and their transposes
    size_t wcat3s, wcatT3s, w1ro3s, w1roT3s;          // TreeLSTM: split-bf16 fragment images of the projection weights (rows_gemm_ksplit3)
    size_t rootp;                       // root vector, padded
    size_t rootw, rootwT, rootw3, rootwT3, rootpb;   // compress = True (diora.py:342-343): root_mat_out^T / root_mat_out (padded), their fragment images, per-sentence root rows (B x Dp)
    size_t matp, matq3, qrleaf;         // inside score matrix padded (Dp x Dp), its fragment image, and QR = M h of the leaves (B*L x Dp)
    size_t xp, ihp, ohp;                // padded copies (only when D != Dp; else unused)
    size_t objp;                        // padded obj (CLIORA, D != Dp)
    size_t t;                           // leaf tanh output (B*L x Dp)
    size_t pi, po;                      // projections of inside cells (B*C x nblk*Dp) / outside cells (B*C x Dp)
    size_t y, x;                        // TreeLSTM only: per-pair h and c (R x Dp each).  DioraMLP keeps no per-pair rows
    size_t pair_h;                      // DioraMLP hooks: per-pair compose outputs (R x Dp) in the OPTIONAL tail of the workspace (beyond `total`)
    size_t pair_h_floats;               // size of that tail
    size_t sp, pp;                      // per-pair score / softmax weight (R)
    size_t hp, hp_o;                    // partial aggregates of level_compose_fwd: HP_PARTS x (B*C x Dp), summed by level_project;
                                        // one per pass (the two passes run as a wavefront on two streams)
    size_t ymask;                       // ReLU bits of the compose output y: R x ncb3 x 4 words (word g of a column block: 4 bits per 16-column tile)
    int ct3, ncb3;                      // column tiles per weight-stationary column block / number of such blocks (Dp = 16 * ct3 * ncb3)
    size_t nrmi, nrmo;                  // per-cell pre-normalisation norm (B*C)
    size_t icp, ocp, nrmic, nrmoc, rootc;   // TreeLSTM: cell-state charts (B*C x Dp), their norms, padded root c
    size_t att_u, att_pk, att_nrmu;     // CLIORA per inside cell: u = unit(aggregate) (B*C x Dp), region probabilities (B*C x 64), |aggregate| (B*C)
    size_t total;
};

struct BwdLayout {
    size_t vh, dg, dstot;               // per-cell backward state (B*C x Dp), (B*C): inside chart
    size_t vh_o, dg_o, dstot_o;         // the same for the outside chart (the two backward chains run as a wavefront on two streams)
    size_t vc_o, dgc_o;                 // TreeLSTM: the cell-state halves of the outside chart's backward state
    size_t da, ds;                      // per-pair grads (R x Dp), (R)
    size_t dz;                          // per-pair grad at the second pre-activation (R x Dp); TreeLSTM: d c_a per pair
    size_t x;                           // DioraMLP: per-pair first-layer activation relu(PL+PR) (R x Dp), re-formed by level_compose_bwd for the weight gradient
    size_t dpp, dpb;                    // DioraMLP: partial dG.y_n per pair row and column block (R x ncb3), and its bias term (R)
    size_t dcb, vc, dgc, grootc;        // TreeLSTM: d c_b per pair (R x Dp), cell-state grads per cell (B*C x Dp) x2, d root c
    size_t dpi, dpo;                    // grads of the projections
    size_t dots, dots_o;                // DioraMLP: H . vH per cell, left by the gathers for the unit-norm backward in the GEMM epilogue (NormBwdLevelE)
    size_t sib_pl, sib_ql, sib_s;       // DioraMLP: sums over the inside cells' sibling uses in the outside pass (cell_gather_bwd_sib): two charts (shared weights only), one scalar per cell
    size_t du, dxp;                     // leaf pre-activation grad, padded dx
    size_t slab, slab2;                 // split-K partial sums for the weight-gradient GEMMs (slab2: the side stream's)
    size_t gwcat, gbcat, gw1ro, gw2i, gb2i, gw2o, gb2o, gwl, gbl, groot, groot_mat;   // packed parameter grads (groot_mat: d root_mat_out^T, compress = True)
    size_t dctx, pmo, dsc, dobjp;       // CLIORA: d context (B*C x Dp), p*mask and d score per region (B*C x 64 each), d obj (B*R x Dp)
    size_t total;
    size_t slab_floats;
};

struct Plan {
    int B, L, D, Dp, C, share, normalize, R;
    int arch;                 // 0 = DioraMLP, 1 = DioraTreeLSTM (composition of cliora/net/vg.py:28-76, commented out upstream)
    int npo, nleaf;           // Dp-wide blocks per outside-cell projection (1 | 5) and per leaf pre-activation (1 | 3)
    int off_pr, off_ql;       // block index of PR and QL inside an inside-cell projection row (1,2 | 5,10)
    int nblk;                 // projection blocks per inside cell: PL, PR, QL (+ PLo, QLo when not shared)
    int blk_plo, blk_qlo;     // which block the outside pass reads for sibling PL / QL
    int P_in, P_out;          // span pairs per sentence
    long long R_in, R_out;    // pair rows in the batch
    std::vector<int32_t> level_offset;                 // L
    std::vector<int32_t> lvl_base_in, lvl_base_out;    // L: pairs per sentence before this level
    std::vector<int32_t> pair_a_in, pair_b_in;         // P_in   (left, right) cell per local pair p*N+n
    std::vector<int32_t> pair_a_out, pair_b_out;       // P_out  (sibling, parent)
    UseList uses[N_ROLES];
    // per global pair row (inside rows first, then outside rows): chart row (b*C + cell) of the
    // a-operand cell, the b-operand cell and the target cell
    std::vector<int32_t> arow, brow, trow;
    // per (pass, level): {Lc, N, chart offset, first pair row, offset into the pass's pair tables, TG, SP, ntask} -- the level's
    // shape and the task geometry of its compose kernel (compose_geom), inside levels first; 2 * L entries of PLEVEL_INTS ints
    std::vector<int32_t> level_geom;
    int compose_cap;          // compose workgroups per column block that geometry is sized for
    FwdLayout fwd;
    BwdLayout bwd;
    // span-region scorer workspace (floats): padded/transposed region matrices, padded sum rows, slabs
    struct VlLayout { size_t oall, oallT, wall, wallT, sump, xwp, xwn, dxn, nrm, gobj, oimg, keys, slab, slab_floats, total; int NRp; } vl;

    // device copies (filled lazily by the HIP side)
    int32_t* d_tables = nullptr;
    size_t d_tables_count = 0;
    struct DevOff { size_t pair_a_in, pair_b_in, pair_a_out, pair_b_out; size_t use_off[N_ROLES], use_row[N_ROLES], use_stride[N_ROLES], use_partner[N_ROLES]; size_t arow, brow, trow, lvl_base_in, lvl_base_out, level_geom; } dev;

    int Lc(int level) const { return L - level; }
    int Nin(int level) const { return level; }
    int Nout(int level) const { return L - level - 1; }
    long long row_base_in(int level) const { return (long long)B * lvl_base_in[level]; }
    long long row_base_out(int level) const { return R_in + (long long)B * lvl_base_out[level]; }
    // The pair rows as 16-row tiles (one tile = 16 target cells of one level x one split: a wave tile of level_compose_bwd), in the
    // pair rows' order (inside levels, then outside levels): the storage unit of the tiled split-bf16 operands of the pair rows'
    // weight gradient (gemm_kernels.hpp: tn_gemm_tiles).  A level with B * Lc not a multiple of 16 carries zero rows in its last tiles.
    std::vector<int32_t> tile_base_in_, tile_base_out_;        // L + 1 each (the last entry: the pass's end); fewer tiles than pair rows / 16 + L^2, which fit 32 bits
    long long T_in = 0, T_out = 0;
    long long tile_base_in(int level) const { return tile_base_in_[level]; }
    long long tile_base_out(int level) const { return T_in + tile_base_out_[level]; }
};

// Tasks of level_compose_fwd for a level of `ncell` target cells with N splits each: TG cell tiles per workgroup (8 / TG waves
// each), the split range cut in SP parts, ntask = ceil(G / TG) * SP workgroup tasks; `cap` workgroups per column block.
struct ComposeGeom { int TG, SP, ntask; };
ComposeGeom compose_geom(int ncell, int N, int cap);
int compose_cap_share(const Plan& p, int pass, int lv);   // the cap a level's geometry is sized for: its share of Plan::compose_cap in its wavefront step

// Builds every host table and the workspace layouts.  Returns "" or an error message.
std::string build_plan(Plan& p, int B, int L, int D, int share, int normalize, int R, int arch = 0);

// Flattens all int32 tables into one array (to upload once); fills p.dev offsets.
std::vector<int32_t> flatten_tables(Plan& p);

const std::vector<int32_t>* find_table(Plan& p, const std::string& name);     // builds the row maps when they are asked for
void build_row_maps(Plan& p);

}  // namespace cliora
