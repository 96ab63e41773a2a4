"""Rows-stationary forward compose (csrc/compose_rs_kernels.hpp; include/cliora_chart.h: cliora_set_rows_stationary).

Three comparisons:
  * 'on' vs 'geometry': the rows-stationary kernel against the weight-stationary kernel dealt the SAME tasks -- the k-steps, the
    epilogue and the reduction over the waves of a cell tile are the same arithmetic in the same order, so every chart, every
    per-split score, the ReLU bits (through the gradients) must agree to the BIT; a stage buffer read before its third landed, or
    overwritten while a wave still reads it, shows up here;
  * 'on' vs the CPU oracle (the reference restated), outputs and gradients, at the tolerances of tests/test_gpu_parity.py;
  * 'on' vs 'off': the plan's own geometry cuts a cell's split range into parts differently, so the aggregates agree to fp32
    rounding only; trees identical.
"""
import pytest
import torch

from test_gpu_parity import CHARTS, OUT_TOL, _err, _grad_ok, _module_from_params, _run_gpu, _scale

from cliora_amd import _lib as _lib_for_skip

# the kernel is an optional part of the build since round 4 (AUTO selects it for no level): these tests run against a library built
# with CLIORA_BUILD_EXTRA=-DCLIORA_WITH_ROWS_STATIONARY and are skipped otherwise
pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not _lib_for_skip.has_rows_stationary(), reason='library built without -DCLIORA_WITH_ROWS_STATIONARY')]

SHAPES = [
    # B, L, share, normalize          (d = 400: the only width the kernel is instantiated for)
    (64, 20, True, 'unit'),           # BASELINE configs[1]
    (16, 34, True, 'unit'),           # splits up to 33: the levels with more than 32 stay on the weight-stationary kernel
    (5, 9, False, 'unit'),            # unshared weights, ragged cell tiles (5 * Lc is never a multiple of 16)
    (3, 13, True, 'none'),
    (1, 2, True, 'unit'),             # one level, one split, one cell
]


def _grads(m):
    return {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}


@pytest.mark.parametrize('B,L,share,normalize', SHAPES)
def test_rows_stationary_kernel_is_bitwise_the_weight_stationary_kernel_on_the_same_tasks(B, L, share, normalize, mfma_mode):
    from cliora_amd import _lib
    from oracle import synth
    D = 400
    P, x, cot = synth.diora_case(D, B, L, 909, share=share)
    m = _module_from_params(P, D, share, normalize)
    prev = _lib.set_rows_stationary('geometry')
    try:
        outs0, xg0 = _run_gpu(m, x, cot)
        outs0 = {k: v.detach().clone() for k, v in outs0.items()}
        g0 = _grads(m)
        _lib.set_rows_stationary('on')
        for rep in range(2):
            for p_ in m.parameters():
                p_.grad = None
            outs1, xg1 = _run_gpu(m, x, cot)
            for k in CHARTS:
                assert torch.equal(outs0[k], outs1[k]), (k, rep, float((outs0[k] - outs1[k]).abs().max()))
            g1 = _grads(m)
            for n in g0:
                assert torch.equal(g0[n], g1[n]), (n, rep, float((g0[n] - g1[n]).abs().max()))
            assert torch.equal(xg0.grad, xg1.grad), rep
    finally:
        _lib.set_rows_stationary(prev)


def test_rows_stationary_hook_states_are_bitwise(mfma_mode):
    """An overridden inside_hook makes the compose kernels write the per-pair states y_n (the rows-stationary kernel stores them
    block by block); no_grad: no ReLU bits."""
    import types
    from cliora_amd import _lib
    from oracle import synth
    D, B, L = 400, 4, 8
    P, x, _ = synth.diora_case(D, B, L, 12)
    m = _module_from_params(P, D, True, 'unit')
    seen = {}

    def hook(self, level, h, c, s):
        seen[level] = (h.detach().clone(), s.detach().clone())
    m.inside_hook = types.MethodType(hook, m)
    res = {}
    prev = _lib.set_rows_stationary('geometry')
    try:
        for mode in ('geometry', 'on'):
            _lib.set_rows_stationary(mode)
            seen.clear()
            with torch.no_grad():
                xg = x.clone().cuda()
                m.eval()
                m(xg, xg)
                torch.cuda.synchronize()
            res[mode] = ({k: getattr(m, k).detach().clone() for k in CHARTS}, dict(seen))
        for k in CHARTS:
            assert torch.equal(res['geometry'][0][k], res['on'][0][k]), k
        assert set(res['on'][1]) == set(range(1, L))
        for lv in res['on'][1]:
            assert torch.equal(res['geometry'][1][lv][0], res['on'][1][lv][0]), ('pair states', lv)
            assert torch.equal(res['geometry'][1][lv][1], res['on'][1][lv][1]), ('pair scores', lv)
    finally:
        _lib.set_rows_stationary(prev)


def test_rows_stationary_against_the_oracle(mfma_mode):
    """d = 400, L = 20 at B = 2 (the shape of diora_c2_small.npz): every level on the rows-stationary kernel, full tensors and all
    gradients against the CPU oracle, trees against the reference's."""
    from conftest import load_golden
    from cliora_amd import _lib
    from oracle import diora_ref as R
    from oracle import synth
    g = load_golden('diora_c2_small.npz')
    meta = g['meta']
    P, x, cot = synth.diora_case(meta['D'], meta['B'], meta['L'], meta['seed'])
    m = _module_from_params(P, meta['D'], True, 'unit')
    prev = _lib.set_rows_stationary('on')
    try:
        outs, xg = _run_gpu(m, x, cot)
        for v in P.values():
            v.requires_grad_(True)
        xc = x.clone().requires_grad_(True)
        ref = R.diora_forward(P, xc, xc, training=True, keep_pairs=True)
        sum((ref[k] * cot[k]).sum() for k in CHARTS).backward()
        for k in CHARTS:
            assert _err(outs[k], ref[k]) <= OUT_TOL, k
        cells = g['cells']
        for k in ('inside_h', 'outside_h'):
            assert _err(outs[k][:, cells], g[k + '__cells']) <= OUT_TOL
        named = dict(m.named_parameters())
        for k, p in P.items():
            _grad_ok(named[k].grad, p.grad, k, mfma_mode)
        _grad_ok(xg.grad, xc.grad, 'x_span', mfma_mode)
        m.eval()
        with torch.no_grad():
            m(x.cuda(), x.cuda())
        assert [str(t) for t in m.cky()] == meta['trees']
    finally:
        _lib.set_rows_stationary(prev)


def test_rows_stationary_at_length_40_matches_the_plan_geometry():
    """B = 64, L = 40: every level of at most 32 splits on the rows-stationary kernel.  Against OFF: the aggregates are
    the same sums cut into parts differently -- charts within a few ulp of the unit rows, scores within 1e-5 of their scale."""
    from cliora_amd import _lib
    from oracle import synth
    D, B, L = 400, 64, 40
    P, x, cot = synth.diora_case(D, B, L, 4040)
    m = _module_from_params(P, D, True, 'unit')
    prev = _lib.set_rows_stationary('off')
    try:
        outs0, xg0 = _run_gpu(m, x, cot)
        outs0 = {k: v.detach().clone() for k, v in outs0.items()}
        g0 = _grads(m)
        _lib.set_rows_stationary('on')
        for p_ in m.parameters():
            p_.grad = None
        outs1, xg1 = _run_gpu(m, x, cot)
        for k in ('inside_h', 'outside_h'):
            assert float((outs0[k] - outs1[k]).detach().abs().max()) <= 2e-5, k
        for k in ('inside_s', 'outside_s'):
            assert float((outs0[k] - outs1[k]).detach().abs().max()) <= 1e-5 * _scale(outs0[k]), k
        g1 = _grads(m)
        for n in g0:
            sc = float(g0[n].abs().max()) + 1e-30
            assert float((g0[n] - g1[n]).abs().max()) <= 2e-3 * sc, n
    finally:
        _lib.set_rows_stationary(prev)


@pytest.mark.parametrize('B,L,R,share,compress', [(8, 9, 36, True, False), (3, 6, 20, False, False), (4, 7, 36, True, True)])
def test_rows_stationary_cliora_levels_are_bitwise(B, L, R, share, compress, mfma_mode):
    """The CLIORA levels (attention residual between the aggregate and the projection; cliora.py:140-157) and the compress root take
    the same compose kernels: rows-stationary against the weight-stationary kernel on the same tasks, every output and gradient bit."""
    from cliora_amd import _lib
    from cliora_amd.cliora import DioraMLP
    D = 400
    torch.manual_seed(13)
    m = DioraMLP(D, outside=True, normalize='unit', compress=compress, share=share).cuda().train()
    for p in m.parameters():
        torch.nn.init.normal_(p, std=0.3)
    C = L * (L + 1) // 2
    m.dropout_mask = (torch.rand(B, C, R, device='cuda') > 0.1).float() / 0.9
    g = torch.Generator().manual_seed(15)
    t = [torch.randn(sh, generator=g).cuda().requires_grad_(True) for sh in ((B, L, D), (B, L, D), (B, R, D), (B, R, D))]
    keys = ('inside_h', 'inside_s', 'outside_h', 'outside_s')
    cot = [torch.randn(B, C, 1 if k.endswith('_s') else D, generator=g).cuda() for k in keys]
    res = None
    prev = _lib.set_rows_stationary('geometry')
    try:
        for mode in ('geometry', 'on'):
            _lib.set_rows_stationary(mode)
            for p in m.parameters():
                p.grad = None
            for x in t:
                x.grad = None
            m(*t)
            outs = [getattr(m, k) for k in keys]
            loss = m.all_atten_score.max(-1).values.sum() * 1e-2 + m.vg_atten_score.sum() * 1e-2
            torch.autograd.backward(outs + [loss], cot + [None])
            cur = ([o.detach().clone() for o in outs], [x.grad.clone() for x in t], {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None})
            if res is None:
                res = cur
                continue
            for a, b in zip(cur[0] + cur[1], res[0] + res[1]):
                assert torch.equal(a, b)
            for n in res[2]:
                assert torch.equal(cur[2][n], res[2][n]), n
    finally:
        _lib.set_rows_stationary(prev)
