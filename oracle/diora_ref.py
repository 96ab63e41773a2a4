"""Torch-CPU restatement of the DIORA / CLIORA chart recursion and its callers.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Differentiable (plain torch
ops), so ``torch.autograd`` on it is the gradient oracle for the hand-written
HIP backward.  It keeps the reference's op sequence (index_select gathers, cat,
two Linear+ReLU, matmul+bmm bilinear, softmax, broadcast multiply + sum,
norm/clamp/div, slice writes into freshly zeroed charts, and the all-zero ``c``
chart traffic) so that timing it is a fair stand-in for timing the reference on
the same host ("port" CPU baseline in bench.py).

Parameter dictionaries use the reference's state_dict names
(cliora/net/diora.py:453-471), e.g. ``inside_compose_func.h_fcs.0.weight``.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import chart_layout as CL

EPS = 1e-8  # cliora/net/utils.py:10


def unit_norm(x):
    """utils.py:11-14: x / max(||x||_2, 1e-8) over the last dim."""
    return x / x.norm(p=2, dim=-1, keepdim=True).clamp(min=EPS)


def _normalizer(mode):
    # utils.py:17-27
    if mode == 'unit':
        return unit_norm
    if mode == 'none':
        return lambda x: x
    raise ValueError(mode)


class Charts:
    """diora.py:7-23 (cliora.py:6-25): six zero-initialised chart tensors."""

    def __init__(self, B, L, D, dtype=torch.float32):
        C = CL.n_cells(L)
        z = lambda w: torch.full((B, C, w), 0, dtype=dtype)     # the reference is fp32; fp64 runs serve as the accuracy yardstick
        self.inside_h, self.inside_c, self.inside_s = z(D), z(D), z(1)
        self.outside_h, self.outside_c, self.outside_s = z(D), z(D), z(1)


def init_params(D, share=True, seed=0, compress=False):
    """All parameters ~ N(0,1): diora.py:234-237 with the shapes of :453-471."""
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=g)
    # the reference registers the root parameter first; compress = True: root_mat_out (D, D) instead of root_vector_out_h (diora.py:466-467)
    P = {'root_mat_out': rn(D, D)} if compress else {'root_vector_out_h': rn(D)}
    P.update({
        'inside_score_func.mat': rn(D, D),
        'inside_compose_func.leaf_fc.weight': rn(D, D),
        'inside_compose_func.leaf_fc.bias': rn(D),
        'inside_compose_func.h_fcs.0.weight': rn(D, 2 * D),
        'inside_compose_func.h_fcs.0.bias': rn(D),
        'inside_compose_func.h_fcs.2.weight': rn(D, D),
        'inside_compose_func.h_fcs.2.bias': rn(D),
    })
    if not share:
        P.update({
            'outside_score_func.mat': rn(D, D),
            'outside_compose_func.h_fcs.0.weight': rn(D, 2 * D),
            'outside_compose_func.h_fcs.0.bias': rn(D),
            'outside_compose_func.h_fcs.2.weight': rn(D, D),
            'outside_compose_func.h_fcs.2.bias': rn(D),
        })
    return P


def init_params_treelstm(D, seed=0, share=True):
    """DioraTreeLSTM parameters ~ N(0,1), in module order:
    root_vector_out_h, root_vector_out_c, inside_score_func.mat, inside_compose_func.{W (3D,D), U (5D,2D), B (5D)}, and with
    share=False (diora.py:462-464) outside_score_func.mat, outside_compose_func.{U, B}.
    RECONSTRUCTION: the class exists in the reference only as commented-out text (cliora/net/vg.py:28-76)."""
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=g)
    P = {
        'root_vector_out_h': rn(D), 'root_vector_out_c': rn(D),
        'inside_score_func.mat': rn(D, D),
        'inside_compose_func.W': rn(3 * D, D), 'inside_compose_func.U': rn(5 * D, 2 * D), 'inside_compose_func.B': rn(5 * D),
    }
    if not share:
        P.update({'outside_score_func.mat': rn(D, D), 'outside_compose_func.U': rn(5 * D, 2 * D), 'outside_compose_func.B': rn(5 * D)})
    return P


def treelstm_leaf(P, x):
    """vg.py:50-61 (commented): [u,i,o] = chunk3(x W^T + B[:3D]); c = sigmoid(i) tanh(u); h = sigmoid(o) tanh(c)."""
    D = x.shape[-1]
    act = torch.matmul(x, P['inside_compose_func.W'].t()) + P['inside_compose_func.B'][:3 * D]
    a = torch.chunk(act, 3, dim=-1)
    u, i, o = torch.tanh(a[0]), torch.sigmoid(a[1]), torch.sigmoid(a[2])
    c = i * u
    return o * torch.tanh(c), c


def treelstm_compose(P, hs, cs, constant=1.0, pre='inside'):
    """vg.py:63-76 (commented): [u,i,o,f0,f1] = chunk5([a;b] U^T + B);
    c = sigmoid(f0+const) c_a + sigmoid(f1+const) c_b + sigmoid(i) tanh(u); h = sigmoid(o) tanh(c).
    pre: which module's U, B ('outside' for the outside pass of an unshared model)."""
    act = torch.matmul(torch.cat(hs, 1), P[pre + '_compose_func.U'].t()) + P[pre + '_compose_func.B']
    a = torch.chunk(act, 5, dim=1)
    u, i, o = torch.tanh(a[0]), torch.sigmoid(a[1]), torch.sigmoid(a[2])
    f0, f1 = torch.sigmoid(a[3] + constant), torch.sigmoid(a[4] + constant)
    c = f0 * cs[0] + f1 * cs[1] + i * u
    return o * torch.tanh(c), c


def _side(P, side, share):
    pre = 'inside' if (share or side == 'inside') else 'outside'
    return dict(
        W1=P[pre + '_compose_func.h_fcs.0.weight'], b1=P[pre + '_compose_func.h_fcs.0.bias'],
        W2=P[pre + '_compose_func.h_fcs.2.weight'], b2=P[pre + '_compose_func.h_fcs.2.bias'],
        M=P[pre + '_score_func.mat'])


def compose_mlp(W, a, b):
    """ComposeMLP.forward, diora.py:65-72: relu(W2 relu(W1 [a;b] + b1) + b2), c = 0."""
    x = torch.cat([a, b], 1)
    h = F.relu(F.linear(F.relu(F.linear(x, W['W1'], W['b1'])), W['W2'], W['b2']))
    c = torch.full(h.shape, 0, dtype=h.dtype)
    return h, c


def bilinear(M, a, b):
    """Bilinear.forward, diora.py:89-97: row-wise a^T M b via matmul + batched 1xD.Dx1."""
    t = torch.matmul(a, M).unsqueeze(1)
    return torch.matmul(t, b.unsqueeze(2)).view(-1, 1)


def attention_head(q, k, v, training, p_drop=0.1, temp=1.0):
    """AttentionHead.forward, cliora.py:35-42.

    Scores every (sentence a, image c) pair, keeps the diagonal, softmax over the
    regions, dropout (active in training mode), context = prob @ v.
    """
    full = torch.einsum('abx,cdx->acbd', q, k)
    score = torch.diagonal(full / temp, 0, 0, 1).permute(2, 0, 1)
    prob = F.dropout(torch.softmax(score, dim=-1), p_drop, training)
    return torch.bmm(prob, v)


def _idx(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def diora_forward(P, x_span, x_word=None, obj_span=None, obj_word=None, *, outside=True,
                  normalize='unit', share=True, training=False, keep_pairs=False, arch='mlp'):
    """DioraBase.forward for DioraMLP: diora.py:424-450 (text) / cliora.py:438-468 (VL).

    Returns a dict with the six charts, and -- when ``keep_pairs`` -- the
    un-aggregated per-level scores the hooks receive (diora.py:331, :398).
    ``obj_span is not None`` selects the CLIORA variant.
    """
    B, L, D = x_span.shape
    vl = obj_span is not None
    nrm = _normalizer(normalize)
    lstm = arch == 'treelstm'           # RECONSTRUCTION (parity unpinned): vg.py:28-76 on the DioraBase skeleton
    if lstm:
        assert not vl
        Win = dict(M=P['inside_score_func.mat'])
        Wout = Win if share else dict(M=P['outside_score_func.mat'])
    else:
        Win, Wout = _side(P, 'inside', share), _side(P, 'outside', share)
    off = CL.level_offsets(L)
    ch = Charts(B, L, D, x_span.dtype)
    pair_s_in, pair_s_out, pair_h_out = {}, {}, {}

    # ---- leaves: diora.py:58-63,283-292 / cliora.py:71-80,290-301
    if lstm:
        h, c = treelstm_leaf(P, x_span)
    else:
        h = torch.tanh(F.linear(x_span, P['inside_compose_func.leaf_fc.weight'],
                                P['inside_compose_func.leaf_fc.bias']))
    if lstm:
        pass
    elif vl:
        h = nrm(h)
        cxt = attention_head(h, obj_span, obj_span, training)
        h = h + cxt
        c = cxt
    else:
        c = torch.full(h.shape, 0, dtype=h.dtype)
    h, c = nrm(h.view(B, L, D)), nrm(c.view(B, L, D))
    ch.inside_h[:, :L] = h
    ch.inside_c[:, :L] = c

    # ---- inside pass: diora.py:295-331
    for level in range(1, L):
        Lc, N = L - level, level
        lidx, ridx = map(_idx, CL.inside_pairs(L, level))
        g = lambda t, i, w: t.index_select(index=i, dim=1).view(-1, w)
        lh, rh = g(ch.inside_h, lidx, D), g(ch.inside_h, ridx, D)
        lc, rc = g(ch.inside_c, lidx, D), g(ch.inside_c, ridx, D)
        ls, rs = g(ch.inside_s, lidx, 1), g(ch.inside_s, ridx, 1)
        ph, pc = treelstm_compose(P, [lh, rh], [lc, rc], 1.0) if lstm else compose_mlp(Win, lh, rh)
        s = (bilinear(Win['M'], lh, rh) + ls + rs).view(B, Lc, N, 1)
        p = torch.softmax(s, dim=2)
        h_agg = torch.sum(ph.view(B, Lc, N, -1) * p, 2)
        c_agg = torch.sum(pc.view(B, Lc, N, -1) * p, 2)
        s_agg = torch.sum(s * p, 2)
        h_agg = nrm(h_agg)
        if vl:  # cliora.py:140-157
            h_agg = nrm(h_agg + attention_head(h_agg, obj_span, obj_span, training))
        c_agg = nrm(c_agg)
        o = int(off[level])
        ch.inside_h[:, o:o + Lc] = h_agg
        ch.inside_c[:, o:o + Lc] = c_agg
        ch.inside_s[:, o:o + Lc] = s_agg
        if keep_pairs:
            pair_s_in[level] = s

    # ---- outside pass: diora.py:337-398
    if outside:
        if 'root_mat_out' in P:      # compress = True (diora.py:342-343)
            rh_ = nrm(torch.matmul(ch.inside_h[:, -1:], P['root_mat_out']))
        else:
            rh_ = nrm(P['root_vector_out_h'].view(1, 1, D).expand(B, 1, D))
        if lstm:      # diora.py:346-350 with a root_vector_out_c parameter (the commented hint at diora.py:470-471)
            rc_ = nrm(P['root_vector_out_c'].view(1, 1, D).expand(B, 1, D))
        else:
            rc_ = nrm(torch.full((B, 1, D), 0, dtype=x_span.dtype))
        ch.outside_h[:, -1:] = rh_
        ch.outside_c[:, -1:] = rc_
        for level in range(L - 2, -1, -1):
            Lc = L - level
            pidx, sidx = map(_idx, CL.outside_pairs(L, level))
            gp = lambda t, w: t.index_select(index=pidx, dim=1).view(-1, w)
            gs = lambda t, w: t.index_select(index=sidx, dim=1).view(-1, w)
            par_h, sib_h = gp(ch.outside_h, D), gs(ch.inside_h, D)
            par_c, sib_c = gp(ch.outside_c, D), gs(ch.inside_c, D)
            par_s, sib_s = gp(ch.outside_s, 1), gs(ch.inside_s, 1)
            if lstm:   # outside_compose passes constant=0 (diora.py:174)
                ph, pc = treelstm_compose(P, [sib_h, par_h], [sib_c, par_c], 0.0, 'inside' if share else 'outside')
            else:
                ph, pc = compose_mlp(Wout, sib_h, par_h)       # order [sibling, parent] :366-368
            s = (bilinear(Wout['M'], sib_h, par_h) + sib_s + par_s).view(B, -1, Lc, 1)
            p = torch.softmax(s, dim=1)
            N = s.shape[1]
            h_agg = nrm(torch.sum(ph.view(B, N, Lc, -1) * p, 1))
            c_agg = nrm(torch.sum(pc.view(B, N, Lc, -1) * p, 1))
            s_agg = torch.sum(s * p, 1)
            o = int(off[level])
            ch.outside_h[:, o:o + Lc] = h_agg
            ch.outside_c[:, o:o + Lc] = c_agg
            ch.outside_s[:, o:o + Lc] = s_agg
            if keep_pairs:
                pair_s_out[level] = s
                pair_h_out[level] = ph.view(B, N, Lc, -1)       # what outside_hook receives as h (diora.py:398)

    out = dict(inside_h=ch.inside_h, inside_c=ch.inside_c, inside_s=ch.inside_s,
               outside_h=ch.outside_h, outside_c=ch.outside_c, outside_s=ch.outside_s,
               pair_s_in=pair_s_in, pair_s_out=pair_s_out, pair_h_out=pair_h_out,
               all_atten_score=None, vg_atten_score=None, atten_score=None)

    # ---- CLIORA tail: cliora.py:453-468
    if vl:
        all_att = torch.einsum('abx,cdx->acbd', ch.inside_h + ch.outside_h, obj_span)
        if training:
            vg = torch.einsum('abx,cdx->acbd', x_word, obj_word)
        else:
            vg = all_att[:, :, :L] + torch.einsum('abx,cdx->acbd', nrm(x_word), obj_word)
        out['all_atten_score'] = all_att
        out['vg_atten_score'] = vg
        out['atten_score'] = torch.diagonal(vg, 0, 0, 1).permute(2, 0, 1)
    return out


# --------------------------------------------------------------------------
# Callers either side of the path (SURVEY.md section 8c "counterparts")
# --------------------------------------------------------------------------

def embed_forward(emb_weight, mat, mat1, sentences):
    """Embed.forward, trainer.py:219-224."""
    B, L = sentences.shape
    e = F.embedding(sentences.view(-1), emb_weight)
    return (torch.mm(e, mat.t()).view(B, L, -1), torch.mm(e, mat1.t()).view(B, L, -1))


def image_encoder_forward(Wf, bf, Wv, bv, obj_feats):
    """ImageEncoder.forward, utils.py:52-55."""
    x = obj_feats.float()
    return F.linear(x, Wf, bf), F.linear(x, Wv, bv)


def reconstruction_loss(emb_weight, mat, sentences, neg_samples, outside_h):
    """ReconstructionSoftmaxLoss.forward, trainer.py:46-78."""
    B, L = sentences.shape
    K = neg_samples.shape[0]
    emb_pos = F.embedding(sentences, emb_weight)
    emb_neg = F.embedding(neg_samples.unsqueeze(0), emb_weight)
    cell = outside_h[:, :L].view(B, L, 1, -1)
    proj_pos = torch.matmul(emb_pos, mat.t())
    proj_neg = torch.matmul(emb_neg, mat.t())
    xp = torch.einsum('abc,abxc->abx', proj_pos, cell)
    xn = torch.einsum('zec,abxc->abe', proj_neg, cell)
    score = torch.cat([xp, xn], 2).view(B * L, K + 1)
    target = torch.full((B * L,), 0, dtype=torch.int64)
    return F.cross_entropy(score, target)


def contrastive_loss(inside_s, outside_s, all_atten_score, margin=0.2, alpha=1.0, min_val=1e-8):
    """ContrastiveLoss.forward, trainer.py:91-128."""
    ins, outs = inside_s.squeeze(-1), outside_s.squeeze(-1)
    B, C = ins.shape
    sc = all_atten_score.max(-1).values.permute(2, 0, 1)         # C,B,B
    diag = torch.diagonal(sc, 0, -1).unsqueeze(-1)
    d1 = diag.expand_as(sc)
    d2 = diag.transpose(1, 2).expand_as(sc)
    lt = (margin + sc - d1).clamp(min=min_val)
    li = (margin + sc - d2).clamp(min=min_val)
    eye = (torch.eye(B, device=sc.device) > 0.5).unsqueeze(0).expand_as(sc)
    lt = lt.masked_fill(eye, 0).mean(2)
    li = li.masked_fill(eye, 0).mean(1)
    vl = (lt + li).t()
    marg = torch.exp(ins + outs - ins[:, [-1]])
    return (marg * vl)[:, :(C // 2)].sum(-1).mean() * alpha


def vg_loss(vg_atten_score, alpha=1.0):
    """VGLoss.forward, trainer.py:139-171 (variant V1)."""
    B, _, L, _ = vg_atten_score.shape
    logits = vg_atten_score.max(-1).values.sum(-1) / L
    return alpha * F.cross_entropy(logits, torch.arange(B, device=logits.device))


# --------------------------------------------------------------------------
# Tree recovery: analysis/utils.py:78-95 (hook) + analysis/cky.py:15-109
# --------------------------------------------------------------------------

def cky_trees(pair_s_in, B, L):
    """Batched CKY over the inside per-split scores.

    ``pair_s_in[level]`` is (B, Lc, N, 1).  The hook stores s - max_n s
    (utils.py:89-93); chart cells start at 1 (cky.py:24-25, :39); the best
    split is the FIRST maximum (argmax, cky.py:86).  Trees are nested tuples of
    word positions.
    """
    val = [np.ones((L - lv, B), dtype=np.float32) for lv in range(L)]
    bp = [[[None] * (L - lv) for lv in range(L)] for _ in range(B)]
    for level in range(1, L):
        s = pair_s_in[level].detach()
        s = (s - s.max(2, keepdim=True)[0]).squeeze(-1).numpy()     # B,Lc,N
        for pos in range(L - level):
            cand = np.stack([val[n][pos] + val[level - n - 1][pos + n + 1] + s[:, pos, n]
                             for n in range(level)], 1)              # B,N  (fp32 adds, l+r then +s)
            best = torch.from_numpy(cand).argmax(1).numpy()
            val[level][pos] = cand[np.arange(B), best]
            for b in range(B):
                bp[b][level][pos] = int(best[b])

    def build(b, level, pos):
        if level == 0:
            return pos
        n = bp[b][level][pos]
        return (build(b, n, pos), build(b, level - n - 1, pos + n + 1))
    return [build(b, L - 1, 0) for b in range(B)]


def tree_score(pair_s_in, b, tree):
    """CKY objective (cky.py:83) of a GIVEN tree of sentence b under the per-split scores: leaves count 1, a node adds
    s_n - max_n s of its split.  cky_trees returns the maximiser; two trees whose scores differ by rounding noise are a tie."""
    def walk(t):
        if isinstance(t, int):
            return t, t, 1.0
        l0, l1, vl = walk(t[0])
        r0, r1, vr = walk(t[1])
        level, pos, n = r1 - l0, l0, l1 - l0
        s = pair_s_in[level][b, pos, :, 0].detach().double()
        return l0, r1, vl + vr + float(s[n] - s.max())
    return walk(tree)[2]


def tree_spans(tree):
    """Constituent spans (start, end) of a nested-tuple tree, children before parents."""
    spans = []

    def walk(t):
        if isinstance(t, int):
            return t, t
        l0, _ = walk(t[0])
        _, r1 = walk(t[1])
        spans.append((l0, r1))
        return l0, r1
    walk(tree)
    return spans
