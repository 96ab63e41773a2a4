"""BASELINE.json configurations at their own sizes (VERDICT r01 "configs not exercised at their own size"):

  c5  DioraTreeLSTM d=400 at L=40                 vs the CPU oracle (forward, gradients, trees)      [parity unpinned, see treelstm.py]
  c3  CLIORA d=400, B=64, L=20, 36 x 2048-d regions through ImageEncoder, reconstruction + VG + contrastive losses
      vs the CPU oracle on the SAME batch (charts, scores, every loss value, gradients)
  c2  full-size gradients against an fp64 run of the oracle: the HIP path's distance to fp64 is held against the distance of
      the reference arithmetic (the fp32 oracle) to fp64, per tensor, at the median, the 99th percentile and the maximum --
      the statement behind the "ReLU kink" tolerance of test_gpu_parity._grad_ok.  The measured ratios are written to
      gpurun_out/accuracy_fp64.json (copied to profiles/).
"""
import json
import os

import numpy as np
import pytest
import torch

from conftest import grad_check

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _err(a, b):
    a = a.detach().float().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().float().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    return float(np.abs(a - b).max()) if a.size else 0.0


def _scale(b):
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    return max(1.0, float(np.abs(b).max())) if b.size else 1.0


def _grad_full(t, ref, mode, what):
    """Full-size gradient check: exact-fp32 mode element-wise at the 99th percentile (2e-4 of scale) with the kink allowance of
    test_gpu_parity._grad_ok at the maximum; split-bf16 mode the statistical check of conftest.grad_check."""
    if mode != 'f32':
        return grad_check(t, ref, mode, 2e-4, what, full_size=True)
    d = (t.detach().cpu().double() - ref.detach().double()).abs().flatten()
    sc = _scale(ref)
    assert float(torch.quantile(d[:: max(1, d.numel() // 200000)], 0.99)) <= 2e-4 * sc, what
    assert float(d.max()) <= 2e-2 * sc, (what, float(d.max()), sc)


def test_c5_treelstm_length_40():
    """d = 400, L = 40 (820 cells, 31 980 pairs per sentence), B = 2: every chart, every gradient and the trees."""
    from cliora_amd.treelstm import DioraTreeLSTM
    from oracle import diora_ref as R
    D, B, L = 400, 2, 40
    keys = ('inside_h', 'inside_c', 'inside_s', 'outside_h', 'outside_c', 'outside_s')
    P = R.init_params_treelstm(D, seed=11)
    gen = torch.Generator().manual_seed(12)
    x = torch.randn(B, L, D, generator=gen)
    m = DioraTreeLSTM(D)
    sd = m.state_dict()
    for k in sd:
        sd[k] = P[k if k in P else 'inside_' + k[len('outside_'):]].detach().clone()
    m.load_state_dict(sd)
    m = m.cuda()
    xg = x.clone().cuda().requires_grad_(True)
    m(xg, xg)
    for v in P.values():
        v.requires_grad_(True)
    xc = x.clone().requires_grad_(True)
    ref = R.diora_forward(P, xc, xc, arch='treelstm', keep_pairs=True)
    C = L * (L + 1) // 2
    cot = {k: torch.randn(B, C, 1 if k.endswith('_s') else D, generator=gen) for k in keys}
    sum((ref[k] * cot[k]).sum() for k in keys).backward()
    torch.autograd.backward([getattr(m, k) for k in keys], [cot[k].cuda() for k in keys])
    torch.cuda.synchronize()
    for k in keys:
        assert _err(getattr(m, k), ref[k]) <= 1e-4 * _scale(ref[k]), k
    named = dict(m.named_parameters())
    for k, p in P.items():           # everything on this path is exact fp32: element-wise bound, kink-tolerant at the maximum
        d = (named[k].grad.detach().cpu().double() - p.grad.double()).abs().flatten()
        sc = _scale(p.grad)
        assert float(torch.quantile(d[:: max(1, d.numel() // 200000)], 0.99)) <= 2e-4 * sc, k
        assert float(d.max()) <= 2e-2 * sc, (k, float(d.max()), sc)
    d = (xg.grad.cpu().double() - xc.grad.double()).abs()
    assert float(d.max()) <= 2e-2 * _scale(xc.grad) and float(torch.quantile(d.flatten()[::4], 0.99)) <= 2e-4 * _scale(xc.grad)
    m.eval()
    with torch.no_grad():
        m(x.cuda(), x.cuda())
    assert [str(t) for t in m.cky()] == [str(t) for t in R.cky_trees(ref['pair_s_in'], B, L)]


def test_c5_treelstm_full_batch_properties():
    """BASELINE configs[4] at the bench's own size -- DioraTreeLSTM d = 400, B = 64, L = 40 (2.05 M pair rows) -- through the
    size-independent properties (the CPU oracle needs minutes per sentence pair at this length; it checks B = 2 above):
    unit-length chart vectors, zero leaf / root scores, sentences independent of their batch (the first two against the oracle run on them
    alone), bitwise equality of two runs, linearity of the backward in the cotangent.  [TreeLSTM parity is unpinned: treelstm.py]"""
    from cliora_amd.treelstm import DioraTreeLSTM
    from oracle import diora_ref as R
    D, B, L = 400, 64, 40
    C = L * (L + 1) // 2
    keys = ('inside_h', 'inside_c', 'inside_s', 'outside_h', 'outside_c', 'outside_s')
    P = R.init_params_treelstm(D, seed=41)
    m = DioraTreeLSTM(D)
    sd = m.state_dict()
    for k in sd:
        sd[k] = P[k if k in P else 'inside_' + k[len('outside_'):]].detach().clone()
    m.load_state_dict(sd)
    m = m.cuda()
    gen = torch.Generator().manual_seed(42)
    x = torch.randn(B, L, D, generator=gen)
    cot = {k: torch.randn(B, C, 1 if k.endswith('_s') else D, generator=gen).cuda() for k in keys}

    def run(scale=1.0):
        for p_ in m.parameters():
            p_.grad = None
        xg = x.clone().cuda().requires_grad_(True)
        m(xg, xg)
        outs = {k: getattr(m, k).detach().clone() for k in keys}
        torch.autograd.backward([getattr(m, k) for k in keys], [scale * cot[k] for k in keys])
        torch.cuda.synchronize()
        return outs, {n: p_.grad.clone() for n, p_ in m.named_parameters()}, xg.grad.clone()
    o1, g1, dx1 = run()
    assert float((o1['inside_h'].norm(dim=-1) - 1).abs().max()) < 1e-5 and float((o1['outside_h'].norm(dim=-1) - 1).abs().max()) < 1e-5
    assert float(o1['inside_s'][:, :L].abs().max()) == 0.0 and float(o1['outside_s'][:, -1].abs().max()) == 0.0
    for k in keys:
        assert bool(torch.isfinite(o1[k]).all()), k
    with torch.no_grad():
        ref = R.diora_forward(P, x[:2], x[:2], arch='treelstm')
    for k in keys:
        assert _err(o1[k][:2], ref[k]) <= 1e-4 * _scale(ref[k]), k
    o2, g2, dx2 = run()
    for k in keys:
        assert torch.equal(o1[k], o2[k]), k
    for n in g1:
        assert torch.equal(g1[n], g2[n]), n
    assert torch.equal(dx1, dx2)
    _, g3, dx3 = run(2.0)
    for n in g1:
        assert _err(g3[n], 2.0 * g1[n]) <= 1e-4 * _scale(g1[n]), n
    assert _err(dx3, 2.0 * dx1) <= 1e-4 * _scale(dx1)


def test_c3_cliora_full_size(mfma_mode):
    """BASELINE configs[2]: CLIORA d=400, batch 64, length 20, 36 regions x 2048-d features, all three losses."""
    from cliora_amd import harness as H
    from oracle import diora_ref as R
    B, L, D, Rg, V, E, K = 64, 20, 400, 36, 10000, 1024, 100      # BASELINE's own sizes: 1024-d embeddings, V 10 000, k_neg 100
    torch.manual_seed(21)
    net = H.build_net(D, torch.nn.Embedding(V, E), obj_feats=True, img_dim=2048, k_neg=K, vg_loss=True, use_contr=True,
                      vl_margin=0.2, alpha_contr=1.0, alpha_vg=1.0)
    for p in net.img_encoder.parameters():
        torch.nn.init.normal_(p, std=0.02)             # the reference's zero init makes every VL score 0
    g = torch.Generator().manual_seed(22)
    sentences = torch.randint(0, V, (B, L), generator=g)
    neg = torch.randperm(V, generator=g)[:K]
    obj = torch.relu(torch.randn(B, Rg, 2048, generator=g))     # region features are post-ReLU in the real data
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    net = net.cuda().eval()                                       # dropout off (the fixtures' convention), gradients on
    out = net(sentences.cuda(), obj.cuda(), neg.cuda())
    out['total_loss'].mean(0).sum().backward()
    torch.cuda.synchronize()
    diora = net.diora

    # ---- the CPU oracle on the same batch
    P = {k[len('diora.'):]: v.clone().requires_grad_(True) for k, v in sd.items() if k.startswith('diora.')}
    emb_w = sd['embed.embeddings.weight'].clone().requires_grad_(True)
    mat, mat1 = sd['embed.mat'].clone().requires_grad_(True), sd['embed.mat1'].clone().requires_grad_(True)
    iw = {k: sd['img_encoder.' + k].clone().requires_grad_(True) for k in ('fc.weight', 'fc.bias', 'fc_vis.weight', 'fc_vis.bias')}
    rmat = sd['reconstruct_softmax_loss.mat'].clone().requires_grad_(True)
    xs, xw = R.embed_forward(emb_w, mat, mat1, sentences)
    os_, ow = R.image_encoder_forward(iw['fc.weight'], iw['fc.bias'], iw['fc_vis.weight'], iw['fc_vis.bias'], obj)
    ref = R.diora_forward(P, xs, xw, os_, ow, training=False, keep_pairs=True)
    l_rec = R.reconstruction_loss(emb_w, rmat, sentences, neg, ref['outside_h'])
    l_vg = R.vg_loss(ref['vg_atten_score'], 1.0)
    l_con = R.contrastive_loss(ref['inside_s'], ref['outside_s'], ref['all_atten_score'], 0.2, 1.0)
    (l_rec + l_vg + l_con).backward()

    for k in ('inside_h', 'inside_s', 'outside_h', 'outside_s', 'inside_c', 'all_atten_score', 'vg_atten_score', 'atten_score'):
        assert _err(getattr(diora, k), ref[k]) <= 1e-4 * _scale(ref[k]), k
    assert float((diora.inside_h.norm(dim=-1) - 1).abs().max()) < 1e-5 and float((diora.outside_h.norm(dim=-1) - 1).abs().max()) < 1e-5
    for name, want in (('reconstruct_softmax_loss', l_rec), ('vg_loss', l_vg), ('contrastive_loss', l_con)):
        got = float(out[name].sum())
        assert abs(got - float(want)) <= 1e-4 * max(1.0, abs(float(want))), (name, got, float(want))
    named = dict(net.named_parameters())
    for k, p in P.items():
        if p.grad is not None:           # shared weights: the state_dict's outside_* aliases take no part
            _grad_full(named['diora.' + k].grad, p.grad, mfma_mode, k)
    for k, p in iw.items():
        _grad_full(named['img_encoder.' + k].grad, p.grad, mfma_mode, k)
    _grad_full(named['embed.mat'].grad, mat.grad, mfma_mode, 'embed.mat')
    # trees of the whole batch
    with torch.no_grad():
        net(sentences.cuda(), obj.cuda(), neg.cuda())
    assert [str(t) for t in diora.cky()] == [str(t) for t in R.cky_trees(ref['pair_s_in'], B, L)]


def test_c3_cliora_full_size_training_mode(mfma_mode):
    """BASELINE configs[2] as the bench runs it: train() -- a recorded (B, C, R) dropout mask on the attention probabilities
    (cliora.py:35-42) and the lazy region-max scorer behind ContrastiveLoss (trainer.py:101) -- against the oracle on the same
    batch with the mask replayed."""
    from cliora_amd import harness as H
    from oracle import diora_ref as R
    B, L, D, Rg, V, E, K = 64, 20, 400, 36, 10000, 1024, 100
    C = L * (L + 1) // 2
    torch.manual_seed(31)
    net = H.build_net(D, torch.nn.Embedding(V, E), obj_feats=True, img_dim=2048, k_neg=K, vg_loss=True, use_contr=True,
                      vl_margin=0.2, alpha_contr=1.0, alpha_vg=1.0)
    for p in net.img_encoder.parameters():
        torch.nn.init.normal_(p, std=0.02)
    g = torch.Generator().manual_seed(32)
    sentences = torch.randint(0, V, (B, L), generator=g)
    neg = torch.randperm(V, generator=g)[:K]
    obj = torch.relu(torch.randn(B, Rg, 2048, generator=g))
    mask = torch.nn.functional.dropout(torch.ones(B, C, Rg), 0.1, True)          # pre-scaled keep mask, chart order (leaves first)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    net = net.cuda().train()
    net.diora.dropout_mask = mask.cuda()
    assert net.diora.lazy_region_scores                                       # the fused region-max path is the one under test
    out = net(sentences.cuda(), obj.cuda(), neg.cuda())
    out['total_loss'].mean(0).sum().backward()
    torch.cuda.synchronize()
    diora = net.diora

    P = {k[len('diora.'):]: v.clone().requires_grad_(True) for k, v in sd.items() if k.startswith('diora.')}
    emb_w = sd['embed.embeddings.weight'].clone().requires_grad_(True)
    mat, mat1 = sd['embed.mat'].clone().requires_grad_(True), sd['embed.mat1'].clone().requires_grad_(True)
    iw = {k: sd['img_encoder.' + k].clone().requires_grad_(True) for k in ('fc.weight', 'fc.bias', 'fc_vis.weight', 'fc_vis.bias')}
    rmat = sd['reconstruct_softmax_loss.mat'].clone().requires_grad_(True)
    xs, xw = R.embed_forward(emb_w, mat, mat1, sentences)
    os_, ow = R.image_encoder_forward(iw['fc.weight'], iw['fc.bias'], iw['fc_vis.weight'], iw['fc_vis.bias'], obj)
    off = [C - (L - lv) * (L - lv + 1) // 2 for lv in range(L)] + [C]
    calls = {'i': 0}
    orig = R.F.dropout

    def replay(x, p, training):            # one call per level, leaves first: the level's slice of the recorded mask
        i = calls['i']
        calls['i'] += 1
        return x * mask[:, off[i]:off[i + 1]]
    R.F.dropout = replay
    try:
        ref = R.diora_forward(P, xs, xw, os_, ow, training=True)
    finally:
        R.F.dropout = orig
    assert calls['i'] == L
    l_rec = R.reconstruction_loss(emb_w, rmat, sentences, neg, ref['outside_h'])
    l_vg = R.vg_loss(ref['vg_atten_score'], 1.0)
    l_con = R.contrastive_loss(ref['inside_s'], ref['outside_s'], ref['all_atten_score'], 0.2, 1.0)
    (l_rec + l_vg + l_con).backward()

    for k in ('inside_h', 'inside_s', 'outside_h', 'outside_s', 'inside_c', 'vg_atten_score', 'atten_score'):
        assert _err(getattr(diora, k), ref[k]) <= 1e-4 * _scale(ref[k]), k
    smax = diora.all_atten_score.max(-1).values                                 # the region maxima, without the dense tensor
    assert _err(smax, ref['all_atten_score'].max(-1).values) <= 1e-4 * _scale(ref['all_atten_score'])
    for name, want in (('reconstruct_softmax_loss', l_rec), ('vg_loss', l_vg), ('contrastive_loss', l_con)):
        got = float(out[name].sum())
        assert abs(got - float(want)) <= 1e-4 * max(1.0, abs(float(want))), (name, got, float(want))
    named = dict(net.named_parameters())
    for k, p in P.items():
        if p.grad is not None:
            _grad_full(named['diora.' + k].grad, p.grad, mfma_mode, k)
    for k, p in iw.items():
        _grad_full(named['img_encoder.' + k].grad, p.grad, mfma_mode, k)
    _grad_full(named['embed.mat'].grad, mat.grad, mfma_mode, 'embed.mat')
    _grad_full(named['embed.mat1'].grad, mat1.grad, mfma_mode, 'embed.mat1')


def test_c5_treelstm_mixed_length_stream():
    """BASELINE configs[4] names mixed-length batches: TreeLSTM batches of L = 10, 40, 17, 40, 10 back to back through the plan cache
    (a different plan, workspace and launch geometry per length; the second visit of a length re-uses its cached plan), each checked
    against the CPU oracle on its own batch.  [TreeLSTM parity is unpinned: treelstm.py]"""
    from cliora_amd import _lib
    from cliora_amd.treelstm import DioraTreeLSTM
    from oracle import diora_ref as R
    D, B = 400, 2
    keys = ('inside_h', 'inside_c', 'inside_s', 'outside_h', 'outside_c', 'outside_s')
    P = R.init_params_treelstm(D, seed=17)
    m = DioraTreeLSTM(D)
    sd = m.state_dict()
    for k in sd:
        sd[k] = P[k if k in P else 'inside_' + k[len('outside_'):]].detach().clone()
    m.load_state_dict(sd)
    m = m.cuda()
    gen = torch.Generator().manual_seed(18)
    plans_before = len(_lib._plans)
    seen = {}
    for step, L in enumerate((10, 40, 17, 40, 10)):
        x = torch.randn(B, L, D, generator=gen)
        C = L * (L + 1) // 2
        cot = {k: torch.randn(B, C, 1 if k.endswith('_s') else D, generator=gen) for k in keys}
        for p_ in m.parameters():
            p_.grad = None
        xg = x.clone().cuda().requires_grad_(True)
        m(xg, xg)
        torch.autograd.backward([getattr(m, k) for k in keys], [cot[k].cuda() for k in keys])
        torch.cuda.synchronize()
        Pc = {k: v.detach().clone().requires_grad_(True) for k, v in P.items()}
        xc = x.clone().requires_grad_(True)
        ref = R.diora_forward(Pc, xc, xc, arch='treelstm')
        sum((ref[k] * cot[k]).sum() for k in keys).backward()
        for k in keys:
            assert _err(getattr(m, k), ref[k]) <= 1e-4 * _scale(ref[k]), (step, L, k)
        named = dict(m.named_parameters())
        for k, p in Pc.items():
            d = (named[k].grad.detach().cpu().double() - p.grad.double()).abs().flatten()
            sc = _scale(p.grad)
            assert float(torch.quantile(d[:: max(1, d.numel() // 200000)], 0.99)) <= 2e-4 * sc, (step, L, k)
            assert float(d.max()) <= 2e-2 * sc, (step, L, k, float(d.max()), sc)
        d = (xg.grad.cpu().double() - xc.grad.double()).abs()
        assert float(d.max()) <= 2e-2 * _scale(xc.grad), (step, L)
        seen[L] = seen.get(L, 0) + 1
    assert len(_lib._plans) - plans_before <= 3            # one plan per distinct length: the repeats hit the cache


def _dist(a, ref64):
    d = (a.detach().cpu().double() - ref64).abs().flatten()
    sc = max(1e-30, float(ref64.abs().max()))
    sub = d[:: max(1, d.numel() // 400000)]
    return dict(q50=float(sub.median()) / sc, q99=float(torch.quantile(sub, 0.99)) / sc, max=float(d.max()) / sc)


_FP64_CASE = {}


def test_c2_full_size_gradients_vs_fp64(mfma_mode):
    """B = 64, L = 20, D = 400: gradients of the HIP path and of the fp32 oracle, both measured against the fp64 oracle."""
    from cliora_amd.diora import DioraMLP
    from oracle import diora_ref as R
    from oracle import synth
    D, B, L = 400, 64, 20
    keys = ('inside_h', 'inside_s', 'outside_h', 'outside_s')
    P, x, cot = synth.diora_case(D, B, L, 1234)
    m = DioraMLP(D)
    sd = m.state_dict()
    for k in sd:
        sd[k] = P[k if k in P else 'inside_' + k[len('outside_'):]].detach().clone()
    m.load_state_dict(sd)
    m = m.cuda().train()
    xg = x.clone().cuda().requires_grad_(True)
    m(xg, xg)
    torch.autograd.backward([getattr(m, k) for k in keys], [cot[k].cuda() for k in keys])
    torch.cuda.synchronize()

    def oracle(dt):
        if dt not in _FP64_CASE:                 # the same two CPU runs serve both arithmetic modes of the GPU path (25 s each)
            Pd = {k: v.detach().to(dt).requires_grad_(True) for k, v in P.items()}
            xd = x.detach().to(dt).requires_grad_(True)
            o = R.diora_forward(Pd, xd, xd, training=True, keep_pairs=(dt == torch.float32))
            sum((o[k] * cot[k].to(dt)).sum() for k in keys).backward()
            _FP64_CASE[dt] = (Pd, xd, {k: o[k].detach() for k in keys})
            if dt == torch.float32:              # the reference's trees of all 64 sentences (analysis/cky.py:31-99 on the oracle's split scores)
                _FP64_CASE['trees'] = [str(t) for t in R.cky_trees(o['pair_s_in'], B, L)]
        return _FP64_CASE[dt]
    P64, x64, o64 = oracle(torch.float64)
    P32, x32, o32 = oracle(torch.float32)
    named = dict(m.named_parameters())
    report = {}
    for k in list(P) + ['x_span']:
        g64 = (x64.grad if k == 'x_span' else P64[k].grad).double()
        hip = _dist(xg.grad if k == 'x_span' else named[k].grad, g64)
        ref = _dist(x32.grad if k == 'x_span' else P32[k].grad, g64)
        report[k] = dict(hip=hip, fp32_oracle=ref, ratio={q: hip[q] / max(ref[q], 1e-12) for q in hip})
    for k in keys:
        report['out.' + k] = dict(hip=_dist(getattr(m, k), o64[k].detach()), fp32_oracle=_dist(o32[k], o64[k].detach()))
        # every chart of all 64 sentences against the fp32 oracle (the reference's arithmetic): the north-star 1e-4
        assert _err(getattr(m, k), o32[k]) <= 1e-4 * _scale(o32[k]), (k, _err(getattr(m, k), o32[k]))
    # "trees identical to reference" on the headline configuration itself: all 64 sentences, in this arithmetic mode (eval forward + GPU CKY)
    m.eval()
    with torch.no_grad():
        xe = x.clone().cuda()
        m(xe, xe)
        got = [str(t) for t in m.cky()]
    m.train()
    assert got == _FP64_CASE['trees'], [b for b in range(B) if got[b] != _FP64_CASE['trees'][b]]
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(ROOT, 'gpurun_out', 'accuracy_fp64_%s.json' % mfma_mode), 'w') as f:
        json.dump(dict(mode=mfma_mode, D=D, B=B, L=L, unit='error / max|fp64 value| of the tensor', tensors=report), f, indent=1)
    # exact-fp32 mode: as close to fp64 as the reference arithmetic is, within 3x at every quantile (+ a floor of 2e-7 of scale
    # for tensors where both are at rounding level).  Split-bf16 mode: operands carry 16 significant bits instead of 24, so
    # the median distance is allowed 16x the fp32 one (measured 4-8x), the 99th percentile 10x (measured <= 7x), the maximum -- set by
    # ReLU kinks in both -- the same factor 3.  Measured values: profiles/r02_accuracy_fp64_*.json.
    for k in list(P) + ['x_span']:
        r = report[k]
        if mfma_mode == 'f32':
            for q, floor in (('q50', 2e-7), ('q99', 2e-6), ('max', 2e-5)):
                assert r['hip'][q] <= 3.0 * r['fp32_oracle'][q] + floor, (k, q, r['hip'][q], r['fp32_oracle'][q])
        else:
            assert r['hip']['q50'] <= 16.0 * r['fp32_oracle']['q50'] + 1e-6, (k, r['hip'], r['fp32_oracle'])
            assert r['hip']['q99'] <= 10.0 * r['fp32_oracle']['q99'] + 1e-4, (k, r['hip'], r['fp32_oracle'])
            assert r['hip']['max'] <= 3.0 * r['fp32_oracle']['max'] + 1e-3, (k, r['hip'], r['fp32_oracle'])
