/*
 * cliora_chart.h -- C ABI of the MI355X-native DIORA/CLIORA chart engine.
 *
 * Drop-in boundary for ONE path of bobwan1995/cliora: the chart-based
 * inside-outside recursion behind DioraMLP.forward().  The reference has no
 * native code and no FFI; what a maintainer replaces is the body of
 *     cliora/net/diora.py:424-450   DioraBase.forward  (text-only DIORA)
 *     cliora/net/cliora.py:438-468  DioraBase.forward  (CLIORA, vision-language)
 * and the autograd graph torch builds under it (driven from
 *     cliora/net/trainer.py:288     self.diora(embed_span, embed_word, obj_span, obj_word)
 *     cliora/net/trainer.py:450-455 Trainer.gradient_update -> loss.backward()).
 * INTEGRATION.md shows the ctypes stub that binds these entry points.
 *
 * Conventions: plain pointers and sizes only, no torch types.  Every data
 * pointer is a DEVICE pointer to contiguous row-major fp32 unless it says
 * otherwise.  The caller owns every buffer (allocates, frees); the library only
 * reads/writes them and keeps no per-call state outside the plan and the
 * caller's workspaces.  All work is enqueued on `stream` (a hipStream_t passed
 * as void*) and returns without synchronising.  Every function returns 0 on
 * success, a negative CLIORA_E* code otherwise; it never throws.
 * cliora_last_error() gives the message for the calling thread.
 *
 * Chart layout (cliora/net/offset_cache.py:1-7): cell (level, pos) = span of
 * words [pos, pos+level]; id = C - (L-level)(L-level+1)/2 + pos, C = L(L+1)/2.
 */
#ifndef CLIORA_CHART_H
#define CLIORA_CHART_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CLIORA_OK 0
#define CLIORA_EINVAL (-1)   /* bad argument / unsupported shape */
#define CLIORA_EHIP (-2)     /* a HIP runtime call failed */
#define CLIORA_ENOMEM (-3)   /* workspace too small */

#define CLIORA_NORM_NONE 0   /* cliora/net/utils.py:17-27 NormalizeFunc('none') */
#define CLIORA_NORM_UNIT 1   /* NormalizeFunc('unit'): x / max(||x||, 1e-8)     */

typedef struct cliora_plan cliora_plan;

/* Parameters of DioraMLP in the reference's own shapes (cliora/net/diora.py:453-471):
 *   leaf_w (D,D), leaf_b (D)          inside_compose_func.leaf_fc.{weight,bias}
 *   w1 (D,2D), b1 (D)                 *_compose_func.h_fcs.0.{weight,bias}
 *   w2 (D,D),  b2 (D)                 *_compose_func.h_fcs.2.{weight,bias}
 *   mat (D,D)                         *_score_func.mat
 *   root_h (D)                        root_vector_out_h   (NULL with compress = True: see root_mat below)
 * With share=1 the out_* pointers are ignored (the reference aliases the modules,
 * diora.py:459-461).  The same struct carries gradients (same shapes, written,
 * not accumulated). */
typedef struct cliora_params {
    float *leaf_w, *leaf_b;
    float *in_w1, *in_b1, *in_w2, *in_b2, *in_mat;
    float *out_w1, *out_b1, *out_w2, *out_b2, *out_mat;
    float *root_h;
    /* DioraTreeLSTM only (composition of cliora/net/vg.py:28-76):
     * lstm_w (3D,D) leaf, lstm_u (5D,2D), lstm_b (5D), root_c (D) = root_vector_out_c;
     * share = 0: lstm_u_out (5D,2D), lstm_b_out (5D) = outside_compose_func.{U,B}, and out_mat = outside_score_func.mat
     * (diora.py:459-464 builds a second compose / score module when the functions are not shared) */
    float *lstm_w, *lstm_u, *lstm_b, *root_c;
    float *lstm_u_out, *lstm_b_out;
    /* compress = True (diora.py:342-343, 466-467): root_mat (D,D) = root_mat_out replaces root_h -- the outside root of a
     * sentence is unit(inside_h[root] @ root_mat_out), so the outside pass starts only after the inside pass has ended */
    float *root_mat;
} cliora_params;

/* A plan fixes (batch B, length L, size D, share, normalize, number of image
 * regions R; R = 0 selects text-only DIORA) and owns the small device index
 * tables for that chart shape: per-level (left,right) / (sibling,parent) cell
 * tables (cliora/net/inside_index.py:182-197, outside_index.py:93-127) and the
 * per-cell use lists the backward gathers over.  Creating a plan builds the
 * tables on the host (no GPU needed); they are uploaded by the FIRST forward
 * call that uses the plan, on the device that is current then (one hipMalloc
 * + copy + stream synchronise; cliora_plan_device_bytes tells how much), and
 * every later call must run on that device.  Create and warm plans outside the
 * step loop, or cache them (the reference caches the same tables in Index,
 * cliora/net/utils.py:67-134; cliora_amd/_lib.py keeps an LRU of plans under
 * a byte budget). */
int cliora_plan_create(int B, int L, int D, int share, int normalize, int R, cliora_plan** out);
/* arch: 0 = DioraMLP (as above), 1 = DioraTreeLSTM (text-only: R = 0) */
int cliora_plan_create_ex(int B, int L, int D, int share, int normalize, int R, int arch, cliora_plan** out);
void cliora_plan_destroy(cliora_plan* plan);

/* Bytes the caller must provide.  fwd workspace: written by forward, must stay
 * untouched until the matching backward.  bwd workspace: scratch. */
size_t cliora_plan_fwd_workspace_bytes(const cliora_plan* plan);
size_t cliora_plan_bwd_workspace_bytes(const cliora_plan* plan);

/* Host view of a plan table (int32), for tests and tooling.  Names:
 * "level_offset", "pair_a_in", "pair_b_in", "pair_a_out", "pair_b_out",
 * "pair_lvl_base_in", "pair_lvl_base_out", "use_off_<role>", "use_row_<role>",
 * "use_stride_<role>", "use_partner_<role>" with role in {ina, inb, outa, outb};
 * "tile_base_in", "tile_base_out" (L + 1 each): first 16-row tile of a level's pair rows in the
 * tiled split-bf16 operand form of the pair rows' weight gradient (csrc/wgrad_tiles.hpp), per pass.
 * Works without a GPU. */
int cliora_plan_table(const cliora_plan* plan, const char* name, const int32_t** data, size_t* count);

/* Forward: DioraBase.forward -- leaf transform, inside pass, and (run_outside != 0)
 * root init + outside pass (diora.py:283-398).
 *   x_span     (B,L,D)   in
 *   obj_span   (B,R,D)   in, CLIORA only (else NULL)
 *   drop_mask  CLIORA training only: pre-scaled dropout mask (0 or 1/(1-p)) of the
 *              AttentionHead (cliora.py:32,40) for every inside cell, laid out like the chart:
 *              (B,C,R), leaves first; NULL = no dropout (eval)
 *   inside_h/outside_h (B,C,D), inside_s/outside_s (B,C)   out
 *   inside_c   (B,C,D)   out, CLIORA only (else NULL): unit(context) at the leaves, 0 above
 * For DioraMLP the c charts are identically zero (diora.py:70); the caller zero-fills
 * them once, they are not touched here.
 * run_outside is a flag word: bit 0 = run the outside pass; CLIORA_FWD_NO_BACKWARD = no backward call will follow
 * (torch.no_grad / eval): the per-pair state the backward needs is not written (hooks need the default). */
#define CLIORA_FWD_NO_BACKWARD 2
/* CLIORA_FWD_PAIR_STATES: also write the un-aggregated compose outputs of every pair (what inside_hook / outside_hook receive,
 * diora.py:295-334, 364-398) into the TAIL of the workspace: the caller then passes a workspace of
 * cliora_plan_fwd_workspace_bytes() + cliora_plan_pair_states_bytes() bytes.  Without it no per-pair vector is stored. */
#define CLIORA_FWD_PAIR_STATES 4
size_t cliora_plan_pair_states_bytes(const cliora_plan* plan);
int cliora_chart_forward(cliora_plan* plan, const cliora_params* params,
                         const float* x_span, const float* obj_span, const float* drop_mask,
                         float* inside_h, float* inside_s, float* outside_h, float* outside_s,
                         float* inside_c,
                         void* fwd_workspace, size_t fwd_workspace_bytes,
                         int run_outside, void* stream);

/* Backward of the call above (what torch.autograd replays through diora.py:295-398).
 * d_* are the cotangents of the four chart outputs (any may be NULL = zero).
 * Writes d_x_span (B,L,D), d_obj_span (B,R,D, CLIORA) and every field of `grads`
 * that is non-NULL (shared weights: inside+outside contributions summed into in_*). */
int cliora_chart_backward(cliora_plan* plan, const cliora_params* params,
                          const float* x_span, const float* obj_span, const float* drop_mask,
                          const float* inside_h, const float* inside_s,
                          const float* outside_h, const float* outside_s,
                          const float* d_inside_h, const float* d_inside_s,
                          const float* d_outside_h, const float* d_outside_s,
                          void* fwd_workspace, size_t fwd_workspace_bytes,
                          void* bwd_workspace, size_t bwd_workspace_bytes,
                          float* d_x_span, float* d_obj_span, const cliora_params* grads,
                          int ran_outside, void* stream);

/* DioraTreeLSTM (BASELINE config 5).  PARITY UNPINNED: the reference ships this composition only
 * as commented-out text (cliora/net/vg.py:28-76); what is implemented is that text on the
 * DioraBase skeleton (cliora/net/diora.py:283-398), constant = 1 inside / 0 outside (diora.py:174),
 * root_vector_out_c a parameter (the hint at diora.py:470-471).  Same conventions as
 * cliora_chart_forward / _backward; here the cell-state charts inside_c / outside_c (B,C,D) are
 * real outputs with cotangents.  Uses lstm_w, lstm_u, lstm_b, in_mat, root_h, root_c of `params`. */
int cliora_lstm_forward(cliora_plan* plan, const cliora_params* params, const float* x_span,
                        float* inside_h, float* inside_c, float* inside_s,
                        float* outside_h, float* outside_c, float* outside_s,
                        void* fwd_workspace, size_t fwd_workspace_bytes, int run_outside, void* stream);
int cliora_lstm_backward(cliora_plan* plan, const cliora_params* params, const float* x_span,
                         const float* inside_h, const float* inside_c, const float* inside_s,
                         const float* outside_h, const float* outside_c, const float* outside_s,
                         const float* d_inside_h, const float* d_inside_c, const float* d_inside_s,
                         const float* d_outside_h, const float* d_outside_c, const float* d_outside_s,
                         void* fwd_workspace, size_t fwd_workspace_bytes,
                         void* bwd_workspace, size_t bwd_workspace_bytes,
                         float* d_x_span, const cliora_params* grads, int ran_outside, void* stream);

/* CLIORA span-region / word-region scorers (cliora/net/cliora.py:453-468):
 *   all_atten (B,B,C,R) = einsum('abx,cdx->acbd', inside_h + outside_h, obj_span)
 *   training != 0:  vg_atten (B,B,L,R) = einsum('abx,cdx->acbd', x_word, obj_word)
 *   training == 0:  vg_atten = all_atten[:, :, :L] + einsum(unit(x_word), obj_word)
 * (atten_score is the a == c diagonal of vg_atten: a view, left to the caller.)
 * Backward: d_sum_h (B,C,D) is the gradient w.r.t. (inside_h + outside_h) -- it goes to both
 * charts; d_obj_span, d_obj_word (B,R,D), d_x_word (B,L,D; training mode only).  NULL cotangents
 * are zero; NULL outputs are skipped. */
size_t cliora_plan_vl_workspace_bytes(const cliora_plan* plan);
/* all_atten may be NULL in training mode (only vg_atten is wanted: the contrastive loss takes cliora_vl_scores_max_forward). */
int cliora_vl_scores_forward(cliora_plan* plan, const float* inside_h, const float* outside_h,
                             const float* obj_span, const float* x_word, const float* obj_word,
                             int training, float* all_atten, float* vg_atten,
                             void* vl_workspace, size_t vl_workspace_bytes, void* stream);
int cliora_vl_scores_backward(cliora_plan* plan, const float* inside_h, const float* outside_h,
                              const float* obj_span, const float* x_word, const float* obj_word,
                              int training, const float* d_all_atten, const float* d_vg_atten,
                              float* d_sum_h, float* d_obj_span, float* d_x_word, float* d_obj_word,
                              void* vl_workspace, size_t vl_workspace_bytes, void* stream);

/* The same span-region scores reduced over the R regions of each image, as ContrastiveLoss consumes them
 * (cliora/net/trainer.py:101 `all_atten_score.max(-1).values`): the (B,B,C,R) tensor (124 MB at B 64, L 20, R 36) is never
 * written -- the GEMM's epilogue keeps, per (sentence a, image c, span b), the largest score and the smallest region index
 * that attains it (what torch.max returns):
 *   all_max (B,B,C) float, all_arg (B,B,C) int32
 * Backward: the cotangent of all_max flows to the arg-max region only (torch's max backward); d_sum_h (B,C,D) as above,
 * d_obj_span (B,R,D).  NULL outputs are skipped.  Uses the same workspace as cliora_vl_scores_forward. */
int cliora_vl_scores_max_forward(cliora_plan* plan, const float* inside_h, const float* outside_h, const float* obj_span,
                                 float* all_max, int32_t* all_arg, void* vl_workspace, size_t vl_workspace_bytes, void* stream);
int cliora_vl_scores_max_backward(cliora_plan* plan, const float* inside_h, const float* outside_h, const float* obj_span,
                                  const float* d_all_max, const int32_t* all_arg, float* d_sum_h, float* d_obj_span,
                                  void* vl_workspace, size_t vl_workspace_bytes, void* stream);

/* ContrastiveLoss after the region max (cliora/net/trainer.py:103-128), one launch per direction:
 *   s[b][a][c] = all_max[a][c][b];  txt = clamp(margin + s - s[b][a][a], min=1e-8), img = clamp(margin + s - s[b][c][c], min=1e-8),
 *   both zero on a == c;  vl[a][b] = mean_c txt[b][a][c] + mean_a' img[b][a'][a];  marg[a][b] = exp(in_s + out_s - in_s[a][C-1]);
 *   loss = alpha * mean_a sum_{b < C/2} marg[a][b] * vl[a][b]
 * all_max (B,B,C), inside_s / outside_s (B,C).  The forward also leaves the gradient of the loss w.r.t. its three inputs (for an
 * upstream cotangent of 1) in d_all_max (B,B,C), d_inside_s, d_outside_s (B,C): the caller scales them by the cotangent it
 * receives.  B <= 128.  workspace: cliora_contrastive_workspace_bytes(B, C). */
size_t cliora_contrastive_workspace_bytes(int B, int C);
int cliora_contrastive_loss(int B, int C, const float* all_max, const float* inside_s, const float* outside_s, float margin,
                            float alpha, float* loss, float* d_all_max, float* d_inside_s, float* d_outside_s, void* workspace,
                            size_t workspace_bytes, void* stream);

/* ---- the callers either side of the chart, as kernels of this library ----
 *
 * Projection of (optionally looked-up) rows: y (nrows, D) = gather(x, index) w^T + bias.
 *   Embed.forward (cliora/net/trainer.py:219-224): x = embeddings.weight (V, K), index = the batch's token ids (nrows = B*L,
 *     int64), w = mat or mat1 (D, K), bias = NULL;
 *   ImageEncoder.forward (cliora/net/utils.py:52-55): x = obj_feats (nrows = B*R, K = 2048), index = NULL, w / bias = fc or fc_vis.
 * K must be a multiple of 16.  Backward: d_w (D, K), d_bias (D), d_rows (nrows, K) = gradient of the looked-up rows (the
 * caller scatters it into the table's gradient); any of them may be NULL.  Exact fp32 products (fp32-input MFMA). */
size_t cliora_proj_workspace_bytes(int nrows, int K, int D);
int cliora_proj_forward(const float* x, const int64_t* index, int nrows, int K, const float* w, const float* bias, int D,
                        float* y, void* workspace, size_t workspace_bytes, void* stream);
int cliora_proj_backward(const float* x, const int64_t* index, int nrows, int K, const float* w, const float* d_y, int D,
                         float* d_w, float* d_bias, float* d_rows, void* workspace, size_t workspace_bytes, void* stream);

/* Gradient of an embedding table from the gradients of its looked-up rows -- what autograd does behind F.embedding
 * (cliora/net/trainer.py:219, :54-58; the table trains when emb = none: cliora/data/embeddings.py:164):
 *   table_grad (V, K) = 0;  table_grad[index[i]] += rows[i], i < n      (index int64, K a multiple of 4; ids outside [0, V) are skipped)
 * Repeated ids add up in ascending i, without atomics: bitwise reproducible. */
int cliora_rows_scatter_add(const float* rows, const int64_t* index, int n, int K, float* table_grad, int64_t V, void* stream);
/* The same with the looked-up rows in up to 4 segments (rows[s] (n[s], K), index[s] (n[s])), as ONE zero-fill + ONE launch: a training
 * step reaches the table through several lookups (trainer.py:54-58 positives and negatives of the reconstruction loss, :219 Embed), and
 * autograd would scatter each into its own dense (V, K) tensor and add them.  Row order = segment order, then row order inside a
 * segment; repeated ids add up in that order, bitwise reproducible.  (Host arrays of device pointers.) */
int cliora_rows_scatter_add_segments(const float* const* rows, const int64_t* const* index, const int* n, int nseg, int K, float* table_grad,
                                     int64_t V, void* stream);

/* ReconstructionSoftmaxLoss.forward (cliora/net/trainer.py:46-78): tokens (B*L) and neg (Kn) int64 ids into emb (V, E), mat (D, E),
 * outside_h (B, C, D) of which the leaf cells [:, :L] are read:
 *   proj = emb[ids] mat^T;  logits_r = [proj_pos_r . cell_r | cell_r proj_neg^T];  loss = mean_r CE(logits_r, 0)
 * E must be a multiple of 16.  The backward runs on the SAME workspace (it keeps the projections and the softmax): gscale is the
 * upstream cotangent (a device scalar); d_cell (B*L, D) is the gradient of outside_h[:, :L], d_mat (D, E), d_rows (B*L + Kn, E)
 * the gradient of the looked-up embedding rows (tokens first, then negatives; NULL when the embeddings are frozen). */
size_t cliora_recon_workspace_bytes(int n_tokens, int Kn, int E, int D);
int cliora_recon_forward(const int64_t* tokens, const int64_t* neg, int B, int L, int C, int Kn, const float* emb, int E,
                         const float* mat, int D, const float* outside_h, float* loss, void* workspace, size_t workspace_bytes,
                         void* stream);
int cliora_recon_backward(const int64_t* tokens, const int64_t* neg, int B, int L, int C, int Kn, const float* emb, int E,
                          const float* mat, int D, const float* outside_h, const float* gscale, float* d_cell, float* d_mat,
                          float* d_rows, void* workspace, size_t workspace_bytes, void* stream);

/* VGLoss.forward (cliora/net/trainer.py:139-171, variant V1): vg_atten (B, B, L, R);
 *   logits[a][c] = sum_l max_r vg_atten[a][c][l][r] / L;  loss = alpha * cross_entropy(logits, arange(B))
 * The forward also leaves the gradient of the loss w.r.t. vg_atten (for an upstream cotangent of 1; it flows to the arg-max region
 * of each word, the smallest index on ties) in d_vg_atten (B, B, L, R), or skips it when NULL. */
size_t cliora_vg_workspace_bytes(int B, int L);
int cliora_vg_loss(int B, int L, int R, const float* vg_atten, float alpha, float* loss, float* d_vg_atten, void* workspace,
                   size_t workspace_bytes, void* stream);

/* Trainer.gradient_update (cliora/net/trainer.py:450-455): torch.nn.utils.clip_grad_norm_(params, max_norm) followed by one
 * torch.optim.Adam step (no weight decay, no amsgrad), over ONE flat fp32 buffer holding every parameter (and one holding every
 * gradient, e.g. the buffer the data-parallel all-reduce works on): three launches whatever the number of parameters.  `step`
 * counts from 1.  grads are left clipped, as clip_grad_norm_ leaves them. */
size_t cliora_clip_adam_workspace_bytes(void);
int cliora_clip_adam(float* params, float* grads, float* exp_avg, float* exp_avg_sq, size_t n, float max_norm, float lr, float beta1,
                     float beta2, float eps, int step, void* workspace, size_t workspace_bytes, void* stream);

/* Un-aggregated per-split tensors the reference hands to inside_hook (diora.py:295-334)
 * for `level`: scores = (B, L-level, level) laid out exactly like the reference's
 * s.view(B, Lc, N, 1); h = the compose outputs, `rows` = B*(L-level)*level rows of D
 * valid floats with row stride `ldh`, same row order as the reference's h (M, D).
 * Both are device pointers into the fwd workspace (h: into its CLIORA_FWD_PAIR_STATES tail; the forward must have run
 * with that flag). */
int cliora_inside_pair_states(const cliora_plan* plan, void* fwd_workspace, int level,
                              const float** scores, const float** h, size_t* rows, size_t* ldh);

/* The same for the outside pass (what the reference hands to outside_hook, diora.py:364-398), for target `level`
 * (0 <= level <= L-2): `rows` = B*(L-level)*(L-level-1) rows in THIS library's split order
 * [sentence][target position j][split n]; the reference's order is [sentence][split i][target position j] with
 *   n = L-2-i-level  if j < (L-level-1) - i,   else n = (L-level-1) - i - 1
 * (cliora/net/outside_index.py:39-62; cliora_amd/index.py::Index.get_outside_index applies it). */
int cliora_outside_pair_states(const cliora_plan* plan, void* fwd_workspace, int level,
                               const float** scores, const float** h, size_t* rows, size_t* ldh);

/* CKY decode (cliora/analysis/cky.py:31-99 + analysis/utils.py:78-95): best binary
 * tree per sentence from the inside per-split scores of the last forward; first
 * maximum wins.  split_out (B, C) int32 device: chosen split n per cell (leaves -1);
 * the tree is rebuilt from it on the host. */
int cliora_cky_decode(cliora_plan* plan, void* fwd_workspace, int32_t* split_out, void* stream);
/* The same decode, with the tree's constituent spans emitted on the device: spans_out (B, L-1, 2) int32 device = (start, end) word
 * positions, children before parents, left subtree first, the root last -- the list the reference builds on the host with
 * get_spans(get_actions(tree)) (cliora/analysis/utils.py:3-49; what scripts/train.py:184-204 scores F1 on): evaluation needs ONE
 * device-to-host copy of B*(L-1)*2 ints and no recursion over a split table.  split_out: as cliora_cky_decode, or NULL. */
int cliora_cky_spans(cliora_plan* plan, void* fwd_workspace, int32_t* split_out, int32_t* spans_out, void* stream);

/* Optional HIP-event timing of one kernel class (used by bench.py for the roofline
 * line): enable, run steps, then read the accumulated device time and launch count. */
int cliora_prof_enable(int kernel_class, int on);
int cliora_prof_read(int kernel_class, double* total_ms, long long* launches, void* stream);
#define CLIORA_KCLASS_COMPOSE_FWD 0
#define CLIORA_KCLASS_COMPOSE_BWD 1
#define CLIORA_KCLASS_WGRAD 2
#define CLIORA_KCLASS_COUNT 3

const char* cliora_last_error(void);
/* Arithmetic of the compose-layer GEMMs (y = W2 x, dx = W2^T dz) and of the pair weight gradient dW2 = dz^T x:
 *   CLIORA_MFMA_SPLIT_BF16 (default): every fp32 operand is carried as two bf16 (hi + lo) and a product is three
 *     v_mfma_f32_16x16x32_bf16 with fp32 accumulation (operand rounding 2^-18; outputs stay within 1e-4 of the reference --
 *     2e-5 measured; gradient elements ~1e-5 of their tensor's scale in the median, single elements up to a few 1e-2 where
 *     a ReLU pre-activation within ~1e-5 of zero changes sign: DESIGN.md section 5);
 *   CLIORA_MFMA_F32: the fp32-input MFMA (products exact in fp32) -- the reference's arithmetic (cliora/net/diora.py:65-68
 *     nn.Linear in fp32), ~1.25x slower per step.
 * Everything else on the path (cell projections, scores, softmax, norms) is fp32 in both modes.
 * Process-wide; the environment variable CLIORA_MFMA=f32|bf16x3 sets the initial value.  Returns the previous mode. */
#define CLIORA_MFMA_F32 0
#define CLIORA_MFMA_SPLIT_BF16 1
int cliora_set_mfma_mode(int mode);

/* Scheduling of the two passes.  Outside level t only reads inside levels <= L-2-t (cliora/net/outside_index.py:39-127), so the
 * outside pass (and, in the backward, the inside pass's backward) can run one step behind the other pass instead of after it:
 * two dependent chains on two HIP streams (the caller's and one owned by the library, forked and joined by events inside every
 * call, so the caller's stream semantics are unchanged).  CLIORA_WAVEFRONT_AUTO (default; the environment variable
 * CLIORA_WAVEFRONT=0|1 sets the initial value) uses two streams when a level carries enough work to pay for the events;
 * OFF runs the reference's order on the caller's stream alone.  Results are bitwise identical either way.  Process-wide;
 * returns the previous mode. */
#define CLIORA_WAVEFRONT_AUTO (-1)
#define CLIORA_WAVEFRONT_OFF 0
#define CLIORA_WAVEFRONT_ON 1
/* MERGED (DioraMLP and CLIORA plans, R >= 0): the same wavefront on ONE queue -- step k of the FORWARD runs ONE compose grid over
 * inside level k and outside level L-k and ONE projection / score grid for both, on the caller's stream (CLIORA plans: the region
 * attention of the new inside cells, cliora.py:140-157, between the two grids): no side stream and no cross-stream event per step.
 * AUTO takes it wherever the two-stream form would pay; ON keeps the two streams.  TreeLSTM plans and compress = True treat MERGED
 * as ON (two streams); the backward always runs its two chains on two streams.  Bitwise the same results again
 * (CLIORA_WAVEFRONT=2). */
#define CLIORA_WAVEFRONT_MERGED 2
int cliora_set_wavefront(int mode);

/* One workgroup per sentence.  For a text-only DioraMLP plan whose rows fit a wavefront (D <= 64; BASELINE configs[0]) the level
 * loops of the forward (cliora/net/diora.py:312-331, 378-398) and of the backward run inside ONE launch each, a workgroup walking
 * every level of one sentence's chart with workgroup barriers between them (csrc/resident_kernels.hpp): sentences exchange nothing
 * inside the recursion.  AUTO (the default; CLIORA_RESIDENT=0|1 sets the initial value) takes it for short sentences
 * (CLIORA_RESIDENT_MAX_PAIRS span pairs per sentence, both passes) or batches of at least half as many sentences as the device has
 * compute units, OFF never, ON for every shape the kernels cover.  Same buffers
 * and formats as the launch-per-level path, exact fp32 FMA arithmetic, results equal to fp32 rounding.  Process-wide; returns the
 * previous mode. */
#define CLIORA_RESIDENT_AUTO (-1)
#define CLIORA_RESIDENT_OFF 0
#define CLIORA_RESIDENT_ON 1
int cliora_set_resident(int mode);

/* Float offset of a named region of the forward workspace ("pi", "po", "hp", "hp_o", "sp", "pp", "ymask", "nrmi", "nrmo", "t",
 * "qrleaf", "total"), for tests and tooling that compare two runs region by region; (size_t)-1 for an unknown name. */
size_t cliora_plan_fwd_offset(const cliora_plan* plan, const char* name);

/* Bytes of device index tables the plan uploads at its first use. */
size_t cliora_plan_device_bytes(const cliora_plan* plan);

/* A stream of the library's, on the current device, for work of the CALLER that is independent of the chart call it surrounds (the
 * word branch of cliora/net/cliora.py:459-461 and trainer.py:139-171: word projections, word-region scorer, VG loss gradient; the
 * region-matrix half of the span-region scorer's backward): chosen once per device so that it runs concurrently with `caller_stream`
 * and with the two streams the chart calls fork onto.  The caller orders it against its own streams with events (torch:
 * torch.cuda.ExternalStream(handle), wait_stream); the library never waits on it.  The handle stays valid for the life of the process. */
int cliora_device_side_stream(void* caller_stream, void** out);

/* Diagnostics: the wall-clock stamps (100 MHz) the sentence-resident kernels leave when CLIORA_RES_TRACE=1 (tools/resident_trace.py):
 * copies `count` 64-bit words of the device's stamp buffer to `out` (host) after synchronising `stream`. */
int cliora_resident_trace(cliora_plan* plan, unsigned long long* out, size_t count, void* stream);

const char* cliora_version(void);

#ifdef __cplusplus
}
#endif
#endif /* CLIORA_CHART_H */
