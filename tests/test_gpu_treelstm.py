"""DioraTreeLSTM on the GPU vs the reconstruction fixture and the CPU oracle.  PARITY UNPINNED: the
reference holds this composition only as commented-out text (cliora/net/vg.py:28-76); see
cliora_amd/treelstm.py and tests/golden/make_golden.py::treelstm_case."""
import numpy as np
import pytest
import torch

from conftest import load_golden, params_from_golden

pytestmark = pytest.mark.gpu
KEYS = ('inside_h', 'inside_c', 'inside_s', 'outside_h', 'outside_c', 'outside_s')


def _module(P, D, share=True):
    from cliora_amd.treelstm import DioraTreeLSTM
    m = DioraTreeLSTM(D, share=share)
    sd = m.state_dict()
    for k in sd:
        sd[k] = P[k if k in P else 'inside_' + k[len('outside_'):]].detach().clone()
    m.load_state_dict(sd)
    return m.cuda()


def _err(a, b):
    return float(np.abs(a.detach().float().cpu().numpy() - np.asarray(b)).max())


def _scale(b):
    return max(1.0, float(np.abs(np.asarray(b)).max()))


@pytest.mark.parametrize('name', ['treelstm_recon.npz', 'treelstm_recon_noshare.npz'])
def test_treelstm_reconstruction_fixture(name):
    g = load_golden(name)
    meta = g['meta']
    m = _module(params_from_golden(g), meta['D'], meta.get('share', True))
    x = torch.from_numpy(g['x_span']).cuda().requires_grad_(True)
    m(x, x)
    for k in KEYS:
        assert _err(getattr(m, k), g[k]) <= 1e-4 * _scale(g[k]), k
    torch.autograd.backward([getattr(m, k) for k in KEYS], [torch.from_numpy(g['cot__' + k]).cuda() for k in KEYS])
    named = dict(m.named_parameters())
    for k, v in g.items():
        if k.startswith('grad__'):
            name = k[6:].replace('__', '.')
            t = x.grad if name == 'x_span' else named[name].grad
            assert _err(t, v) <= 2e-4 * _scale(v), '%s %.3e' % (k, _err(t, v))
    m.eval()
    with torch.no_grad():
        m(x.detach(), x.detach())
    assert [str(t) for t in m.cky()] == meta['trees']


@pytest.mark.parametrize('D,B,L,share', [(400, 2, 12, True), (64, 4, 9, True), (400, 3, 11, False), (48, 5, 8, False)])
def test_treelstm_against_oracle(D, B, L, share):
    from oracle import diora_ref as R
    P = R.init_params_treelstm(D, seed=6, share=share)
    gen = torch.Generator().manual_seed(7)
    x = torch.randn(B, L, D, generator=gen)
    m = _module(P, D, share)
    xg = x.clone().cuda().requires_grad_(True)
    m(xg, xg)
    for v in P.values():
        v.requires_grad_(True)
    xc = x.clone().requires_grad_(True)
    ref = R.diora_forward(P, xc, xc, arch='treelstm', share=share)
    C = L * (L + 1) // 2
    cot = {k: torch.randn(B, C, 1 if k.endswith('_s') else D, generator=gen) for k in KEYS}
    sum((ref[k] * cot[k]).sum() for k in KEYS).backward()
    torch.autograd.backward([getattr(m, k) for k in KEYS], [cot[k].cuda() for k in KEYS])
    for k in KEYS:
        assert _err(getattr(m, k), ref[k].detach().numpy()) <= 1e-4 * _scale(ref[k].detach().numpy()), k
    named = dict(m.named_parameters())
    for k, p in P.items():
        assert _err(named[k].grad, p.grad.numpy()) <= 2e-4 * _scale(p.grad.numpy()), k
    assert _err(xg.grad, xc.grad.numpy()) <= 2e-4 * _scale(xc.grad.numpy())


def test_treelstm_hooks_receive_the_reference_states():
    """inside_hook / outside_hook on the TreeLSTM module: (level, h, c, s) in the reference's layout and order (diora.py:331, 398),
    checked against the oracle's per-split scores and outside compose outputs."""
    import types
    from oracle import diora_ref as R
    D, B, L = 48, 3, 7
    P = R.init_params_treelstm(D, seed=4)
    x = torch.randn(B, L, D, generator=torch.Generator().manual_seed(8))
    m = _module(P, D).eval()
    seen_in, seen_out = [], []
    m.inside_hook = types.MethodType(lambda self, level, h, c, s: seen_in.append((level, h.clone(), s.clone())), m)
    m.outside_hook = types.MethodType(lambda self, level, h, c, s: seen_out.append((level, h.clone(), s.clone())), m)
    with torch.no_grad():
        m(x.cuda(), x.cuda())
        ref = R.diora_forward(P, x, x, arch='treelstm', keep_pairs=True)
    assert [lv for lv, _, _ in seen_in] == list(range(1, L)) and [lv for lv, _, _ in seen_out] == list(range(L - 2, -1, -1))
    for level, h, s in seen_in:
        want = ref['pair_s_in'][level]
        assert tuple(s.shape) == tuple(want.shape) == (B, L - level, level, 1)
        assert _err(s, want) <= 2e-4 * _scale(want), level
        assert tuple(h.shape) == (B * (L - level) * level, D)
    for level, h, s in seen_out:
        Lc, N = L - level, L - level - 1
        want_s, want_h = ref['pair_s_out'][level], ref['pair_h_out'][level].reshape(B * N * Lc, D)
        assert tuple(s.shape) == (B, N, Lc, 1) and _err(s, want_s) <= 2e-4 * _scale(want_s), level
        assert tuple(h.shape) == (B * N * Lc, D) and _err(h, want_h) <= 1e-4 * _scale(want_h), level


@pytest.mark.parametrize('D,B,L,share', [(400, 8, 14, True), (48, 3, 6, True), (96, 4, 9, False)])
def test_treelstm_wavefront_is_bitwise_the_sequential_order(D, B, L, share):
    """TreeLSTM levels on two streams (cliora_set_wavefront) against the one-stream order: every output and gradient bit."""
    from cliora_amd import _lib
    from oracle import diora_ref as R
    P = R.init_params_treelstm(D, seed=2, share=share)
    m = _module(P, D, share)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(B, L, D, generator=g).cuda().requires_grad_(True)
    C = L * (L + 1) // 2
    cot = [torch.randn(B, C, 1 if k.endswith('_s') else D, generator=g).cuda() for k in KEYS]
    prev = _lib.set_wavefront('off')
    ref = None
    try:
        for mode in ('off', 'on', 'on'):
            _lib.set_wavefront(mode)
            for p in m.parameters():
                p.grad = None
            x.grad = None
            m(x, x)
            outs = [getattr(m, k) for k in KEYS]
            torch.autograd.backward(outs, cot)
            cur = [o.detach().clone() for o in outs] + [x.grad.clone()] + [p.grad.clone() for p in m.parameters() if p.grad is not None]
            if ref is None:
                ref = cur
                continue
            assert len(cur) == len(ref)
            for a, b in zip(cur, ref):
                assert torch.equal(a, b)
    finally:
        _lib.set_wavefront(prev)
