"""GPU parity of the CLIORA (vision-language) path against the golden vectors captured from
cliora/net/cliora.py + trainer.py losses (tests/golden/cliora_small.npz) and the CPU oracle."""
import numpy as np
import pytest
import torch

from conftest import grad_check, load_golden, params_from_golden

pytestmark = pytest.mark.gpu

OUT_TOL = 1e-4
GRAD_TOL = 2e-4
OUTS = ('inside_h', 'inside_c', 'inside_s', 'outside_h', 'outside_c', 'outside_s',
        'all_atten_score', 'vg_atten_score', 'atten_score')


def _module(g, share=True):
    from cliora_amd.cliora import DioraMLP
    P = params_from_golden(g)
    m = DioraMLP(g['meta']['D'], outside=True, normalize='unit', compress=False, share=share)
    sd = m.state_dict()
    for k in sd:
        sd[k] = P[k].clone()
    m.load_state_dict(sd)
    return m.cuda()


def _inputs(g, rg=False):
    t = {k: torch.from_numpy(g[k].copy()).cuda() for k in ('x_span', 'x_word', 'obj_span', 'obj_word')}
    if rg:
        for v in t.values():
            v.requires_grad_(True)
    return t


def _err(a, b):
    a = a.detach().float().cpu().numpy()
    return float(np.abs(a - np.asarray(b)).max())


def _scale(b):
    return max(1.0, float(np.abs(np.asarray(b)).max()))


def test_cliora_eval_outputs_and_trees(mfma_mode):
    g = load_golden('cliora_small.npz')
    m = _module(g).eval()
    t = _inputs(g)
    with torch.no_grad():
        m(t['x_span'], t['x_word'], t['obj_span'], t['obj_word'])
    for k in OUTS:
        assert _err(getattr(m, k), g['eval__' + k]) <= OUT_TOL * _scale(g['eval__' + k]), k
    assert [str(x) for x in m.cky()] == g['meta']['trees']


def _chart_mask(g):
    n = g['meta']['n_masks']
    return torch.cat([torch.from_numpy(g['mask_%d' % i]) for i in range(n)], 1).cuda()   # (B, C, R): leaves, level 1, ...


def test_cliora_train_with_recorded_dropout_and_grads(mfma_mode):
    from oracle import diora_ref as R
    g = load_golden('cliora_small.npz')
    m = _module(g).train()
    m.dropout_mask = _chart_mask(g)
    t = _inputs(g, rg=True)
    m(t['x_span'], t['x_word'], t['obj_span'], t['obj_word'])
    for k in ('inside_h', 'inside_s', 'outside_h', 'outside_s', 'all_atten_score', 'vg_atten_score', 'atten_score'):
        assert _err(getattr(m, k), g['train__' + k]) <= OUT_TOL * _scale(g['train__' + k]), k
    # the two VL losses of the reference (trainer.py:91-171), as torch ops on the native outputs
    lc = R.contrastive_loss(m.inside_s, m.outside_s, m.all_atten_score, 0.2, 1.0)
    lv = R.vg_loss(m.vg_atten_score, 1.0)
    assert abs(float(lc) - float(g['train__contrastive_loss'])) <= 1e-4 * max(1.0, abs(float(g['train__contrastive_loss'])))
    assert abs(float(lv) - float(g['train__vg_loss'])) <= 1e-4 * max(1.0, abs(float(g['train__vg_loss'])))
    cot = {k[5:]: torch.from_numpy(v).cuda() for k, v in g.items() if k.startswith('cot__')}
    (lc + lv + sum((getattr(m, k) * v).sum() for k, v in cot.items())).backward()
    torch.cuda.synchronize()
    named = dict(m.named_parameters())
    for k, v in g.items():
        if not k.startswith('grad__'):
            continue
        name = k[6:].replace('__', '.')
        tt = t[name].grad if name in t else named[name].grad
        assert tt is not None, k
        grad_check(tt, v, mfma_mode, GRAD_TOL, k)


@pytest.mark.parametrize('D,B,L,R,share,compress', [(64, 3, 5, 36, True, False), (48, 2, 6, 10, False, False), (400, 2, 7, 36, True, False),
                                                     (32, 2, 5, 64, True, False), (32, 2, 4, 1, True, False), (64, 3, 6, 36, True, True)])
def test_cliora_against_oracle(D, B, L, R, share, compress, mfma_mode):
    """Other shapes (Dp == D and Dp != D, R not a multiple of 4... of 16, unshared weights) vs the CPU oracle."""
    from cliora_amd.cliora import DioraMLP
    from oracle import diora_ref as Rf
    torch.manual_seed(5)
    P = Rf.init_params(D, share=share, seed=3, compress=compress)        # compress: cliora.py:355-356, as diora.py:342-343
    gen = torch.Generator().manual_seed(4)
    x_span, x_word = torch.randn(B, L, D, generator=gen), torch.randn(B, L, D, generator=gen)
    obj_span, obj_word = 0.3 * torch.randn(B, R, D, generator=gen), 0.3 * torch.randn(B, R, D, generator=gen)
    C = L * (L + 1) // 2
    mask = torch.nn.functional.dropout(torch.ones(B, C, R), 0.1, True)
    m = DioraMLP(D, share=share, compress=compress)
    sd = m.state_dict()
    for k in sd:
        sd[k] = P[k if k in P else 'inside_' + k[len('outside_'):]].clone()
    m.load_state_dict(sd)
    m = m.cuda().train()
    m.lazy_region_scores = False      # this test differentiates all_atten_score itself (a cotangent on the dense tensor)
    m.dropout_mask = mask.cuda()
    tg = {k: v.clone().cuda().requires_grad_(True) for k, v in dict(x_span=x_span, x_word=x_word, obj_span=obj_span, obj_word=obj_word).items()}
    m(tg['x_span'], tg['x_word'], tg['obj_span'], tg['obj_word'])

    # oracle with the same masks: replay them through F.dropout by patching it
    for v in P.values():
        v.requires_grad_(True)
    tc = {k: v.clone().requires_grad_(True) for k, v in dict(x_span=x_span, x_word=x_word, obj_span=obj_span, obj_word=obj_word).items()}
    off = [C - (L - lv) * (L - lv + 1) // 2 for lv in range(L)] + [C]
    calls = {'i': 0}
    orig = torch.nn.functional.dropout

    def replay(x, p, training):
        i = calls['i']
        calls['i'] += 1
        return x * mask[:, off[i]:off[i + 1]]
    Rf.F.dropout = replay
    try:
        ref = Rf.diora_forward(P, tc['x_span'], tc['x_word'], tc['obj_span'], tc['obj_word'], share=share, training=True)
    finally:
        Rf.F.dropout = orig
    keys = ('inside_h', 'inside_s', 'outside_h', 'outside_s', 'all_atten_score', 'vg_atten_score')
    gen2 = torch.Generator().manual_seed(9)
    cot = {k: torch.randn(ref[k].shape, generator=gen2) for k in keys}
    sum((ref[k] * cot[k]).sum() for k in keys).backward()
    torch.autograd.backward([getattr(m, k) for k in keys], [cot[k].cuda() for k in keys])
    torch.cuda.synchronize()
    for k in keys + ('inside_c', 'atten_score'):
        assert _err(getattr(m, k), ref[k].detach().numpy()) <= OUT_TOL * _scale(ref[k].detach().numpy()), k
    named = dict(m.named_parameters())
    for k, p in P.items():
        grad_check(named[k].grad, p.grad, mfma_mode, GRAD_TOL, k)
    for k in tg:
        grad_check(tg[k].grad, tc[k].grad, mfma_mode, GRAD_TOL, k)


@pytest.mark.parametrize('B,L,D,R', [(5, 7, 48, 6), (8, 9, 400, 36), (3, 5, 33, 5), (64, 20, 400, 36)])
def test_region_max_scorer_is_the_dense_scorer_then_max(B, L, D, R, mfma_mode):
    """all_atten_score.max(-1) in training mode (ContrastiveLoss, trainer.py:101) runs cliora_vl_scores_max_forward / _backward:
    values, region indices and every gradient must be exactly those of the dense (B, B, C, R) tensor followed by torch.max."""
    from cliora_amd.cliora import DioraMLP, LazyRegionScores
    torch.manual_seed(7)
    m = DioraMLP(D, outside=True, normalize='unit', compress=False, share=True).cuda().train()
    for p in m.parameters():
        torch.nn.init.normal_(p, std=0.3)
    C = L * (L + 1) // 2
    mask = torch.ones(B, C, R, device='cuda')
    w = torch.randn(B, B, C, device='cuda')
    res = {}
    for lazy in (True, False):
        m.lazy_region_scores, m.dropout_mask = lazy, mask
        g = torch.Generator().manual_seed(11)
        t = [torch.randn(sh, generator=g).cuda().requires_grad_(True) for sh in ((B, L, D), (B, L, D), (B, R, D), (B, R, D))]
        for p in m.parameters():
            p.grad = None
        m(*t)
        a = m.all_atten_score
        assert isinstance(a, LazyRegionScores) == lazy
        assert tuple(a.shape) == (B, B, C, R)
        r = a.max(-1)
        ((r.values * w).sum() + m.vg_atten_score.sum() * 0.1).backward()
        res[lazy] = (r.values.detach(), r.indices, [x.grad.clone() for x in t], {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None})
    assert torch.equal(res[True][0], res[False][0])          # same accumulators, max taken in the epilogue: bit for bit
    assert torch.equal(res[True][1], res[False][1])
    # the backward sums the arg-max rows directly (sparse) instead of a dense GEMM over 35/36 zeros: same terms, another order
    close = lambda x, y: float((x - y).abs().max()) <= 2e-5 * max(1.0, float(y.abs().max()))
    for ga, gb in zip(res[True][2], res[False][2]):
        assert close(ga, gb)
    for n in res[True][3]:
        assert close(res[True][3][n], res[False][3][n]), n
    # the lazy object still serves everything else the dense tensor does
    m.lazy_region_scores = True
    m(*[x.detach() for x in t])
    a = m.all_atten_score
    dense = a[:, :, :L]
    assert tuple(dense.shape) == (B, B, L, R) and torch.equal((a * 2.0).max(-1).values, res[False][0] * 2.0)


@pytest.mark.parametrize('B,L,D,R,share', [(16, 12, 400, 36, True), (6, 3, 48, 5, False), (64, 20, 400, 36, True)])
def test_cliora_wavefront_is_bitwise_the_sequential_order(B, L, D, R, share, mfma_mode):
    """The CLIORA levels (attention between the aggregate and the projection) on two streams, and as merged launches on one queue, against
    the sequential order on one stream: bit for bit."""
    from cliora_amd import _lib
    from cliora_amd.cliora import DioraMLP
    torch.manual_seed(3)
    m = DioraMLP(D, outside=True, normalize='unit', compress=False, share=share).cuda().train()
    for p in m.parameters():
        torch.nn.init.normal_(p, std=0.3)
    C = L * (L + 1) // 2
    m.dropout_mask = (torch.rand(B, C, R, device='cuda') > 0.1).float() / 0.9
    g = torch.Generator().manual_seed(5)
    t = [torch.randn(sh, generator=g).cuda().requires_grad_(True) for sh in ((B, L, D), (B, L, D), (B, R, D), (B, R, D))]
    keys = ('inside_h', 'inside_s', 'outside_h', 'outside_s')
    cot = [torch.randn(B, C, 1 if k.endswith('_s') else D, generator=g).cuda() for k in keys]
    res = {}
    prev = _lib.set_wavefront('off')
    try:
        for mode in ('off', 'on', 'on', 'merged', 'merged'):       # merged (round 5): the forward's two chains as one grid per phase on one queue
            _lib.set_wavefront(mode)
            for p in m.parameters():
                p.grad = None
            for x in t:
                x.grad = None
            m(*t)
            outs = [getattr(m, k) for k in keys]
            loss = m.all_atten_score.max(-1).values.sum() * 1e-2 + m.vg_atten_score.sum() * 1e-2
            torch.autograd.backward(outs + [loss], cot + [None])
            cur = ([o.detach().clone() for o in outs], [x.grad.clone() for x in t], {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None})
            if mode == 'off':
                res = cur
                continue
            for a, b in zip(cur[0] + cur[1], res[0] + res[1]):
                assert torch.equal(a, b)
            for n in res[2]:
                assert torch.equal(cur[2][n], res[2][n]), n
    finally:
        _lib.set_wavefront(prev)
