"""GPU parity: the HIP chart path (through the C ABI) against the golden vectors
captured from the reference and against the CPU oracle on the same seeded inputs.

Tolerance: outputs within 1e-4 absolute in fp32 (BASELINE.json north_star) in BOTH arithmetic modes of the
compose GEMMs; gradients within 2e-4 of the tensor's largest reference magnitude with exact fp32 products
(mode 'f32'), and the statistical check of conftest.grad_check in the default split-bf16 mode (median 5e-4, 99 % of
the elements 1e-2, every element 5e-2 of scale; at the d = 400 sizes median 2e-4, 99 % 4e-3, every element 2e-2: what
profiles/r02_accuracy_fp64_bf16x3.json measures there, with a margin of two).  The parity tests run under both.
"""
import numpy as np
import pytest
import torch

from conftest import grad_check, load_golden, params_from_golden

pytestmark = pytest.mark.gpu

OUT_TOL = 1e-4
GRAD_TOL = 2e-4

CHARTS = ('inside_h', 'inside_s', 'outside_h', 'outside_s')


def _module_from_params(P, D, share, normalize, outside=True):
    from cliora_amd.diora import DioraMLP
    m = DioraMLP(D, outside=outside, normalize=normalize, compress='root_mat_out' in P, share=share)
    sd = m.state_dict()
    for k in sd:
        src = k
        if share and k.startswith('outside_'):
            src = 'inside_' + k[len('outside_'):]
        sd[k] = P[src].detach().clone()
    m.load_state_dict(sd)
    return m.cuda()


def _err(a, b):
    a = a.detach().float().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().float().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    return float(np.abs(a - b).max()) if a.size else 0.0


def _scale(b):
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    return max(1.0, float(np.abs(b).max())) if b.size else 1.0


def _grad_ok(t, ref, what='', mode='f32'):
    """Kink-tolerant gradient check for the d=400 / L=20 shapes.

    With ~6 M ReLU pre-activations per step, about one of them sits within fp32 rounding of
    zero, and whichever side a given summation order lands on changes ONE row of a weight
    gradient by a finite amount.  The reference's own CPU fp32 path shows exactly this against
    an fp64 run of itself (seeds 1236/1238/1239: single rows off by 1e-4..2e-3 of the tensor
    scale, median element error ~1e-6).  So: 99% of the elements within GRAD_TOL, every
    element within 100x that.
    """
    if mode != 'f32':
        return grad_check(t, ref, mode, GRAD_TOL, what, full_size=ref.shape[-1] >= 256)     # split-bf16: the statistical check of conftest.grad_check (d = 400 bounds for wide tensors)
    a = t.detach().double().cpu().flatten()
    b = ref.detach().double().cpu().flatten()
    d = (a - b).abs()
    scale = max(1.0, float(b.abs().max()))
    q = float(torch.quantile(d[:: max(1, d.numel() // 200000)], 0.99))
    assert q <= GRAD_TOL * scale, '%s: q99 err %.3e scale %.3e' % (what, q, scale)
    assert float(d.max()) <= 100 * GRAD_TOL * scale, '%s: max err %.3e scale %.3e' % (what, float(d.max()), scale)


def _run_gpu(m, x, cot):
    xg = x.clone().cuda().requires_grad_(True)
    m.train()
    m(xg, xg)
    outs = {k: getattr(m, k) for k in CHARTS}
    torch.autograd.backward([outs[k] for k in CHARTS if k in cot], [cot[k].cuda() for k in CHARTS if k in cot])
    torch.cuda.synchronize()
    return outs, xg


@pytest.mark.parametrize('name', ['diora_c1.npz', 'diora_noshare.npz', 'diora_nonorm.npz', 'diora_len2.npz', 'diora_compress.npz'])
def test_golden_forward_backward(name, mfma_mode):
    g = load_golden(name)
    meta = g['meta']
    P = params_from_golden(g)
    m = _module_from_params(P, meta['D'], meta['share'], meta['normalize'])
    cot = {k[5:]: torch.from_numpy(v) for k, v in g.items() if k.startswith('cot__')}
    outs, xg = _run_gpu(m, torch.from_numpy(g['x_span']), cot)
    # without unit normalisation the vectors grow to ~1e3 over ten levels and so does every rounding error: the
    # split-bf16 mode is held to 3e-4 of the tensor's scale there (1.1e-4 measured), 1e-4 everywhere else
    out_tol = OUT_TOL * (3.0 if mfma_mode == 'bf16x3' and meta['normalize'] == 'none' else 1.0)
    for k in CHARTS:
        assert _err(outs[k], g[k]) <= out_tol * _scale(g[k]), k
    assert float(m.inside_c.abs().max()) == 0.0 and float(m.outside_c.abs().max()) == 0.0
    named = dict(m.named_parameters())
    for k, v in g.items():
        if not k.startswith('grad__'):
            continue
        name_ = k[6:].replace('__', '.')
        t = xg.grad if name_ == 'x_span' else named[name_].grad
        assert t is not None, k
        grad_check(t, v, mfma_mode, GRAD_TOL, k)


@pytest.mark.parametrize('name', ['diora_c1.npz', 'diora_noshare.npz', 'diora_len2.npz', 'diora_compress.npz'])
def test_hook_scores_and_trees(name, mfma_mode):
    from oracle import diora_ref as R
    g = load_golden(name)
    meta = g['meta']
    P = params_from_golden(g)
    m = _module_from_params(P, meta['D'], meta['share'], meta['normalize'])
    m.eval()
    saved = {}

    def hook(level, h, c, s):            # what analysis/utils.py:78-95 stores
        saved[level] = (s - s.max(2, keepdim=True)[0]).cpu()
        assert h.shape == (meta['B'] * (meta['L'] - level) * level, meta['D'])
    m.inside_hook = hook
    with torch.no_grad():
        x = torch.from_numpy(g['x_span']).cuda()
        m(x, x)
    for level in range(1, meta['L']):
        assert _err(saved[level], g['hook_s_%d' % level]) <= 2e-4
    trees = m.cky()
    assert [str(t) for t in trees] == meta['trees']
    assert [[list(s) for s in R.tree_spans(t)] for t in trees] == meta['spans']
    # the span lists as the device emits them (cliora_cky_spans: children before parents, the root last) are the reference's own
    # get_spans(get_actions(tree)) lists (analysis/utils.py:3-49) of the fixture
    assert m.cky_spans().tolist() == meta['spans']


def test_c2_shape_against_oracle_and_golden(mfma_mode):
    """d=400, L=20 (BASELINE config 2 shape) at B=2: full tensors vs the CPU oracle, checksums vs the reference."""
    from oracle import diora_ref as R
    from oracle import synth
    g = load_golden('diora_c2_small.npz')
    meta = g['meta']
    P, x, cot = synth.diora_case(meta['D'], meta['B'], meta['L'], meta['seed'])
    m = _module_from_params(P, meta['D'], True, 'unit')
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
    assert _err(outs['inside_s'], g['inside_s']) <= OUT_TOL * _scale(g['inside_s'])
    named = dict(m.named_parameters())
    for k, p in P.items():
        _grad_ok(named[k].grad, p.grad, k, mfma_mode)
    _grad_ok(xg.grad, xc.grad, 'x_span', mfma_mode)
    m.eval()
    with torch.no_grad():
        m(x.cuda(), x.cuda())
    assert [str(t) for t in m.cky()] == meta['trees']


def test_full_size_c2_properties(mfma_mode):
    """B=64, L=20, D=400 (BASELINE config 2): size-independent properties + oracle on a sentence subset."""
    from oracle import diora_ref as R
    from oracle import synth
    D, B, L = 400, 64, 20
    P, x, cot = synth.diora_case(D, B, L, 1234)
    m = _module_from_params(P, D, True, 'unit')
    outs, xg = _run_gpu(m, x, cot)
    ih, oh = outs['inside_h'], outs['outside_h']
    # every chart vector is unit length; leaf/root scores are the zero-initialised ones
    assert float((ih.norm(dim=-1) - 1).abs().max()) < 1e-5
    assert float((oh.norm(dim=-1) - 1).abs().max()) < 1e-5
    assert float(outs['inside_s'][:, :L].abs().max()) == 0.0
    assert float(outs['outside_s'][:, -1].abs().max()) == 0.0
    # sentences are independent: the first two must match the oracle run on them alone
    with torch.no_grad():
        ref = R.diora_forward(P, x[:2], x[:2])
    for k in CHARTS:
        assert _err(outs[k][:2], ref[k]) <= OUT_TOL, k
    # determinism (fixed-order reductions, no atomics): a second run is bitwise identical
    g1 = {n: p.grad.clone() for n, p in m.named_parameters()}
    for p in m.parameters():
        p.grad = None
    outs2, xg2 = _run_gpu(m, x, cot)
    for k in CHARTS:
        assert torch.equal(outs[k], outs2[k]), k
    for n, p in m.named_parameters():
        assert torch.equal(g1[n], p.grad), n
    assert torch.equal(xg.grad, xg2.grad)
    # backward is linear in the cotangent
    for p in m.parameters():
        p.grad = None
    cot2 = {k: 2.0 * v for k, v in cot.items()}
    _run_gpu(m, x, cot2)
    for n, p in m.named_parameters():
        assert _err(p.grad, 2.0 * g1[n]) <= 1e-4 * _scale(g1[n]), n


@pytest.mark.parametrize('D,B,L,share', [(400, 64, 20, True), (96, 16, 12, False), (48, 5, 9, True), (400, 3, 33, True), (64, 7, 3, True)])
def test_wavefront_is_bitwise_the_sequential_order(D, B, L, share, mfma_mode):
    """The two passes on two streams (include/cliora_chart.h: cliora_set_wavefront) against the reference's order on one stream:
    the same kernels on the same data, only scheduled differently -- every output and gradient bit must agree, run after run
    (a missing cross-stream dependency shows up here as a difference)."""
    from cliora_amd import _lib
    from oracle import synth
    P, x, cot = synth.diora_case(D, B, L, 99, share=share)
    m = _module_from_params(P, D, share, 'unit')
    prev = _lib.set_wavefront('off')
    try:
        outs0, xg0 = _run_gpu(m, x, cot)
        g0 = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
        # 'on': two chains on two streams; 'merged' (round 5): the two passes' launches of a step as ONE grid on the caller's stream
        for mode in ('on', 'merged'):
            _lib.set_wavefront(mode)
            for rep in range(3):
                for p_ in m.parameters():
                    p_.grad = None
                outs1, xg1 = _run_gpu(m, x, cot)
                for k in CHARTS:
                    assert torch.equal(outs0[k], outs1[k]), (mode, k, rep)
                for n, p_ in m.named_parameters():
                    if p_.grad is not None:
                        assert torch.equal(g0[n], p_.grad), (mode, n, rep)
                assert torch.equal(xg0.grad, xg1.grad), (mode, rep)
    finally:
        _lib.set_wavefront(prev)


def test_inside_only_eval_mode(mfma_mode):
    """run_eval turns the outside pass off for DIORA (scripts/train.py:130): charts stay zero, grads still flow."""
    from oracle import diora_ref as R
    from oracle import synth
    D, B, L = 48, 3, 7
    P, x, cot = synth.diora_case(D, B, L, 5)
    m = _module_from_params(P, D, True, 'unit', outside=False)
    cot = {k: cot[k] for k in ('inside_h', 'inside_s')}
    outs, xg = _run_gpu(m, x, cot)
    for v in P.values():
        v.requires_grad_(True)
    xc = x.clone().requires_grad_(True)
    ref = R.diora_forward(P, xc, xc, outside=False)
    sum((ref[k] * cot[k]).sum() for k in cot).backward()
    for k in CHARTS:
        assert _err(outs[k], ref[k]) <= OUT_TOL, k
    named = dict(m.named_parameters())
    for k, p in P.items():
        if p.grad is None:
            assert float(named[k].grad.abs().max()) == 0.0, k
        else:
            grad_check(named[k].grad, p.grad, mfma_mode, GRAD_TOL, k)
    grad_check(xg.grad, xc.grad, mfma_mode, GRAD_TOL, 'x_span')


@pytest.mark.parametrize('D,B,L,share', [(400, 3, 9, True), (48, 5, 6, False), (64, 2, 2, True), (33, 4, 1, True)])
def test_compress_root_against_oracle(D, B, L, share, mfma_mode):
    """compress = True (diora.py:342-343): the outside root of a sentence is unit(inside_h[root] @ root_mat_out), so the outside pass
    follows the inside pass and its gradient flows back into the inside root.  The reference's own output for this mode is
    tests/golden/diora_compress.npz (test_golden_forward_backward); here other widths, unshared weights, L = 2 and L = 1."""
    from oracle import diora_ref as R
    from oracle import synth
    P, x, cot = synth.diora_case(D, B, L, 31, share=share, compress=True)
    m = _module_from_params(P, D, share, 'unit')
    assert m.compress and hasattr(m, 'root_mat_out') and not hasattr(m, 'root_vector_out_h')
    outs, xg = _run_gpu(m, x, cot)
    for v in P.values():
        v.requires_grad_(True)
    xc = x.clone().requires_grad_(True)
    ref = R.diora_forward(P, xc, xc, share=share, training=True)
    sum((ref[k] * cot[k]).sum() for k in CHARTS).backward()
    for k in CHARTS:
        assert _err(outs[k], ref[k]) <= OUT_TOL * _scale(ref[k].detach().numpy()), k
    named = dict(m.named_parameters())
    for k, p in P.items():
        if p.grad is None:           # L = 1: the leaves and the root only, no compose / score parameter is used
            assert named[k].grad is None or float(named[k].grad.abs().max()) == 0.0, k
        else:
            _grad_ok(named[k].grad, p.grad, k, mfma_mode)        # kink-tolerant: with compress one ReLU on the fence moves EVERY gradient
    _grad_ok(xg.grad, xc.grad, 'x_span', mfma_mode)
    # the inside pass alone (scripts/train.py:130 at eval): no root is formed, root_mat_out gets a zero gradient
    m2 = _module_from_params(P, D, share, 'unit', outside=False)
    cot_in = {k: cot[k] for k in ('inside_h', 'inside_s')}
    outs2, _ = _run_gpu(m2, x, cot_in)
    assert torch.equal(outs2['inside_h'].detach(), outs['inside_h'].detach()) and float(outs2['outside_h'].detach().abs().max()) == 0.0
    assert float(m2.root_mat_out.grad.abs().max()) == 0.0


@pytest.mark.parametrize('B,L,share,outside', [(3, 7, False, True), (5, 6, True, False), (2, 9, False, False), (17, 5, True, True)])
def test_d400_tiled_pair_rows_variants_against_oracle(B, L, share, outside, mfma_mode):
    """d = 400 in the default arithmetic keeps X / DZ of the backward as 16-row tiles of bf16 hi / lo planes and runs the pair rows'
    weight gradient on them (csrc/wgrad_tiles.hpp).  The headline shape covers shared weights, both passes and whole tiles (B * Lc a
    multiple of 16); here the other branches of that path: unshared weights (the outside rows are a tile range of their own), the inside
    pass alone, and batches whose levels end in ragged tiles (rows past the last cell are stored as zeros)."""
    from oracle import diora_ref as R
    from oracle import synth
    D = 400
    P, x, cot = synth.diora_case(D, B, L, 17, share=share)
    m = _module_from_params(P, D, share, 'unit', outside=outside)
    if not outside:
        cot = {k: cot[k] for k in ('inside_h', 'inside_s')}
    outs, xg = _run_gpu(m, x, cot)
    for v in P.values():
        v.requires_grad_(True)
    xc = x.clone().requires_grad_(True)
    ref = R.diora_forward(P, xc, xc, share=share, outside=outside)
    sum((ref[k] * cot[k]).sum() for k in cot).backward()
    for k in cot:
        assert _err(outs[k], ref[k]) <= OUT_TOL * _scale(ref[k].detach().numpy()), k
    named = dict(m.named_parameters())
    for k, p_ in P.items():
        if p_.grad is None:
            assert named[k].grad is None or float(named[k].grad.abs().max()) == 0.0, k
        else:
            _grad_ok(named[k].grad, p_.grad, k, mfma_mode)
    _grad_ok(xg.grad, xc.grad, 'x_span', mfma_mode)


def test_cpu_tensor_fails_loudly():
    from cliora_amd.diora import DioraMLP
    from cliora_amd._lib import ChartLibError
    m = DioraMLP(16)
    with pytest.raises(ChartLibError):
        m(torch.randn(2, 4, 16), None)


def test_length_40_chart():
    """L=40 (the reference's --train_filter_length cap, BASELINE config 5 shape with the MLP composition):
    820 cells, 31 980 pairs per sentence; first sentence against the oracle, plus properties on all."""
    from oracle import diora_ref as R
    from oracle import synth
    D, B, L = 400, 4, 40
    P, x, cot = synth.diora_case(D, B, L, 77)
    m = _module_from_params(P, D, True, 'unit')
    outs, xg = _run_gpu(m, x, cot)
    with torch.no_grad():
        ref = R.diora_forward(P, x[:1], x[:1], keep_pairs=True)
    for k in CHARTS:
        assert _err(outs[k][:1], ref[k]) <= OUT_TOL * _scale(ref[k]), k
    assert float((outs['inside_h'].norm(dim=-1) - 1).abs().max()) < 1e-5
    assert float((outs['outside_h'].norm(dim=-1) - 1).abs().max()) < 1e-5
    assert all(torch.isfinite(p.grad).all() for p in m.parameters())
    m.eval()
    with torch.no_grad():
        m(x.cuda(), x.cuda())
    assert str(m.cky()[0]) == str(R.cky_trees(ref['pair_s_in'], 1, L)[0])


@pytest.mark.parametrize('D', [48, 64, 96, 400])
def test_weight_stationary_kernels_other_widths(D, mfma_mode):
    """Levels above 1 500 pair rows run the weight-stationary compose kernels; d = 400 has its own unrolled instance,
    every other width the run-time-K one with 1, 2 or 4 column tiles per block (Dp = 48 / 96 / 64).  B = 32, L = 14
    (5 824 rows at the widest level) against the CPU oracle, forward and backward, both arithmetic modes."""
    from oracle import diora_ref as R
    from oracle import synth
    B, L = (32, 14) if D < 400 else (8, 20)
    P, x, cot = synth.diora_case(D, B, L, 17)
    m = _module_from_params(P, D, True, 'unit')
    outs, xg = _run_gpu(m, x, cot)
    for v in P.values():
        v.requires_grad_(True)
    xc = x.clone().requires_grad_(True)
    ref = R.diora_forward(P, xc, xc, training=True)
    sum((ref[k] * cot[k]).sum() for k in CHARTS).backward()
    for k in CHARTS:
        assert _err(outs[k], ref[k]) <= OUT_TOL * _scale(ref[k]), k
    named = dict(m.named_parameters())
    for k, p in P.items():
        _grad_ok(named[k].grad, p.grad, k, mfma_mode)
    _grad_ok(xg.grad, xc.grad, 'x_span', mfma_mode)


@pytest.mark.parametrize('D,B,L', [(16, 2, 64), (512, 2, 6), (400, 1, 3), (33, 3, 9)])
def test_limits_of_the_plan(D, B, L, mfma_mode):
    """The largest chart length and hidden size a plan accepts (L = 64: 63 splits per cell, every lane of the split
    softmax in use; D = 512: two full 16-byte vectors per lane), a single sentence, and an odd width (D = 33 -> Dp = 48
    with 15 pad columns) -- forward and backward against the CPU oracle."""
    from oracle import diora_ref as R
    from oracle import synth
    P, x, cot = synth.diora_case(D, B, L, 23)
    m = _module_from_params(P, D, True, 'unit')
    outs, xg = _run_gpu(m, x, cot)
    for v in P.values():
        v.requires_grad_(True)
    xc = x.clone().requires_grad_(True)
    ref = R.diora_forward(P, xc, xc, training=True)
    sum((ref[k] * cot[k]).sum() for k in CHARTS).backward()
    for k in CHARTS:
        assert _err(outs[k], ref[k]) <= OUT_TOL * _scale(ref[k]), k
    named = dict(m.named_parameters())
    for k, p in P.items():
        _grad_ok(named[k].grad, p.grad, k, mfma_mode)
    _grad_ok(xg.grad, xc.grad, 'x_span', mfma_mode)
    m.eval()
    with torch.no_grad():
        m(x.cuda(), x.cuda())
    with torch.no_grad():       # trees identical to the reference's in BOTH arithmetic modes
        pair_s = R.diora_forward(P, x, x, keep_pairs=True)['pair_s_in']
        want = R.cky_trees(pair_s, B, L)
    got = m.cky()
    for b in range(B):
        if str(got[b]) == str(want[b]):
            continue
        # The one case seen to differ (split-bf16 mode, D = 16, L = 64, sentence 1): a 63-level chart where two splits of one
        # cell score within rounding of each other.  Then the tree found must be a tie of the reference's under the REFERENCE's
        # own scores (CKY objective within 1e-4), and only in the split mode.
        assert mfma_mode == 'bf16x3', 'exact-fp32 mode must reproduce the trees'
        gap = R.tree_score(pair_s, b, want[b]) - R.tree_score(pair_s, b, got[b])
        assert 0.0 <= gap <= 1e-4, (b, gap)


def test_outside_hook_receives_the_reference_states():
    """An overridden outside_hook gets (level, h, c, s) in the reference's layout and order (diora.py:364-398):
    s (B, N, Lc, 1), h (B*N*Lc, D), from level L-2 down to 0 -- checked against the oracle's per-split outside states."""
    import types
    from oracle import diora_ref as R
    from oracle import synth
    D, B, L = 24, 3, 7
    P, x, _ = synth.diora_case(D, B, L, 19)
    m = _module_from_params(P, D, True, 'unit').eval()
    seen = []
    m.outside_hook = types.MethodType(lambda self, level, h, c, s: seen.append((level, h.clone(), c.clone(), s.clone())), m)
    with torch.no_grad():
        m(x.cuda(), x.cuda())
        ref = R.diora_forward(P, x, x, keep_pairs=True)
    assert [lv for lv, _, _, _ in seen] == list(range(L - 2, -1, -1))
    for level, h, c, s in seen:
        Lc, N = L - level, L - level - 1
        want_s = ref['pair_s_out'][level]
        assert tuple(s.shape) == (B, N, Lc, 1) == tuple(want_s.shape)
        assert _err(s, want_s) <= 2e-4 * _scale(want_s), level
        assert tuple(h.shape) == (B * N * Lc, D) and float(c.abs().max()) == 0.0
        if 'pair_h_out' in ref:
            want_h = ref['pair_h_out'][level].reshape(B * N * Lc, D)      # un-normalised compose outputs (magnitude ~10)
            assert _err(h, want_h) <= OUT_TOL * _scale(want_h), level


def test_two_host_threads_share_the_device_lanes():
    """Two host threads, each with its own torch stream and model, stepping at the same time: the library's side streams and
    events are per device and shared (api_core.hip: device_lanes, guarded by a mutex while a call enqueues), so every result
    must still be bitwise what the same thread computes alone."""
    import threading
    from oracle import synth
    D, B, L = 400, 16, 12
    cases = [synth.diora_case(D, B, L, 40 + i) for i in range(2)]
    mods = [_module_from_params(P, D, True, 'unit') for P, _, _ in cases]

    def run(i, out, reps):
        P, x, cot = cases[i]
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(reps):
                for p in mods[i].parameters():
                    p.grad = None
                outs, xg = _run_gpu(mods[i], x, cot)
            s.synchronize()
            out[i] = ({k: v.detach().clone() for k, v in outs.items()}, xg.grad.clone(),
                      {n: p.grad.clone() for n, p in mods[i].named_parameters() if p.grad is not None})
    alone, together = {}, {}
    for i in range(2):
        run(i, alone, 1)
    th = [threading.Thread(target=run, args=(i, together, 6)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for i in range(2):
        for k in CHARTS:
            assert torch.equal(alone[i][0][k], together[i][0][k]), (i, k)
        assert torch.equal(alone[i][1], together[i][1]), i
        for n in alone[i][2]:
            assert torch.equal(alone[i][2][n], together[i][2][n]), (i, n)


def test_two_host_threads_make_their_first_call_on_one_cold_plan():
    """Plans are shared process-wide (cliora_amd/_lib.py) and ctypes drops the GIL: two threads whose FIRST forward hits the same
    (B, L, D) plan both find it not yet uploaded.  The upload is guarded per plan (api_core.hip: cliora_plan_ready, `uploaded`
    published last): both calls must succeed and give the result a warm plan gives.  A TreeLSTM plan is the hard case -- its upload
    builds the batch-expanded row maps (a std::vector resize) -- so both architectures are exercised, on shapes no other test uses."""
    import threading
    from cliora_amd import _lib
    from cliora_amd.treelstm import DioraTreeLSTM
    from oracle import diora_ref as R
    from oracle import synth
    for arch in ('mlp', 'treelstm'):
        D, B, L = (80, 7, 11) if arch == 'mlp' else (48, 5, 13)
        assert not any(k[0] == B and k[1] == L and k[2] == D for k in _lib._plans), 'shape already warm: pick another'
        if arch == 'mlp':
            P, x, _ = synth.diora_case(D, B, L, 77)
            mods = [_module_from_params(P, D, True, 'unit').eval() for _ in range(2)]
        else:
            P = R.init_params_treelstm(D, seed=5)
            x = torch.randn(B, L, D, generator=torch.Generator().manual_seed(6))
            mods = []
            for _ in range(2):
                m = DioraTreeLSTM(D)
                sd = m.state_dict()
                for k in sd:
                    sd[k] = P[k if k in P else 'inside_' + k[len('outside_'):]].detach().clone()
                m.load_state_dict(sd)
                mods.append(m.cuda().eval())
        go = threading.Barrier(2)
        res, errs = {}, []

        def run(i):
            try:
                s = torch.cuda.Stream()
                with torch.cuda.stream(s), torch.no_grad():
                    xg = x.clone().cuda()
                    go.wait()
                    mods[i](xg, xg)
                    s.synchronize()
                    res[i] = {k: getattr(mods[i], k).detach().clone() for k in CHARTS}
            except BaseException as e:           # surfaces below
                errs.append(e)
        th = [threading.Thread(target=run, args=(i,)) for i in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errs, errs
        with torch.no_grad():
            xg = x.clone().cuda()
            mods[0](xg, xg)
            torch.cuda.synchronize()
            warm = {k: getattr(mods[0], k).detach().clone() for k in CHARTS}
        for i in range(2):
            for k in CHARTS:
                assert torch.equal(res[i][k], warm[k]), (arch, i, k)


@pytest.mark.parametrize('D,B,L', [(64, 3, 17), (96, 2, 20), (400, 1, 16), (33, 5, 18), (256, 2, 16), (400, 7, 23)])
def test_early_weight_gradient_ranges(D, B, L, mfma_mode):
    """L >= 16 with shared weights: the pair rows' weight gradient is taken in three pieces -- the finished middle of the rows on the GEMM
    stream after backward step (L-1)/2, the two end ranges in the tail (api_mlp.hip).  Every width class of the weight-gradient kernels,
    odd batch sizes, and CLIORA_WGRAD_EARLY_STEP's default against the oracle; the pieces must add up to the reference's dW2 / db2."""
    from oracle import diora_ref as R
    from oracle import synth
    P, x, cot = synth.diora_case(D, B, L, 2024)
    m = _module_from_params(P, D, True, 'unit')
    outs, xg = _run_gpu(m, x, cot)
    for v in P.values():
        v.requires_grad_(True)
    xc = x.clone().requires_grad_(True)
    ref = R.diora_forward(P, xc, xc, training=True)
    sum((ref[k] * cot[k]).sum() for k in CHARTS).backward()
    for k in CHARTS:
        assert _err(outs[k], ref[k]) <= OUT_TOL * _scale(ref[k].detach().numpy()), k
    named = dict(m.named_parameters())
    for k, p in P.items():
        _grad_ok(named[k].grad, p.grad, k, mfma_mode)
    _grad_ok(xg.grad, xc.grad, 'x_span', mfma_mode)


@pytest.mark.parametrize('D,B,L', [(64, 2, 8), (24, 3, 7), (96, 1, 12)])
def test_no_normalization_scores_beyond_2_to_24(D, B, L):
    """normalize='none': the vectors grow by an order of magnitude per level and the scores pass 1e8.  The softmax backward's factor
    1 + (s_n - S) must be formed in that order -- as (1 + s_n) - S it loses the 1 beyond 2^24 and every gradient through the score of
    such a cell vanishes (found by tools/fuzz_parity.py in round 3; the reference's own fixture diora_nonorm.npz stays below 2^24).
    Exact-fp32 mode, every output and gradient against the oracle, relative to each tensor's scale."""
    from cliora_amd import _lib
    from oracle import diora_ref as R
    from oracle import synth
    prev = _lib.set_mfma_mode('f32')
    try:
        P, x, cot = synth.diora_case(D, B, L, 5)
        m = _module_from_params(P, D, True, 'none')
        outs, xg = _run_gpu(m, x, cot)
        for v in P.values():
            v.requires_grad_(True)
        xc = x.clone().requires_grad_(True)
        ref = R.diora_forward(P, xc, xc, normalize='none', training=True)
        assert float(ref['outside_s'].abs().max()) > 2.0 ** 24 or L < 8       # the regime the test is about (the small shape is the fixture's)
        sum((ref[k] * cot[k]).sum() for k in CHARTS).backward()
        rel = lambda a, b: float((a.detach().double().cpu() - b.detach().double()).abs().max() / (b.detach().double().abs().max() + 1e-30))
        for k in CHARTS:
            assert rel(outs[k], ref[k]) <= 2e-5, k
        named = dict(m.named_parameters())
        for k, p in P.items():
            assert rel(named[k].grad, p.grad) <= 2e-5, (k, rel(named[k].grad, p.grad))
        assert rel(xg.grad, xc.grad) <= 2e-5
    finally:
        _lib.set_mfma_mode(prev)


# ---------------------------------------------------------------------------------------------------------------------------
# The two regimes tools/fuzz_parity.py kept flagging in round 3, pinned with explicit expected tolerances.
# Errors are measured against the tensor's OWN scale (max |reference|, no floor at 1): in both regimes that scale is far from 1.
# Bounds = what tools/regimes.py measured on MI355X (profiles/r04_regimes.txt) with a margin of ~4; the fp32 oracle's own distance
# to an fp64 run of itself is of the same size in every case (same file), so these are the reference's conditioning, not a defect.
# ---------------------------------------------------------------------------------------------------------------------------
def _regime_case(D, B, L, normalize, compress, seed):
    from oracle import diora_ref as R
    P = R.init_params(D, share=True, seed=seed, compress=compress)
    m = _module_from_params(P, D, True, normalize).train()
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, L, D, generator=g)
    xg = x.clone().cuda().requires_grad_(True)
    m(xg, xg)
    Pd = {k: v.detach().clone().requires_grad_(True) for k, v in P.items()}
    xc = x.clone().requires_grad_(True)
    ref = R.diora_forward(Pd, xc, xc, share=True, normalize=normalize, training=True)
    cot = {k: torch.randn(ref[k].shape, generator=g) for k in CHARTS}
    if normalize == 'none':          # cotangents scaled to the outputs, so that every level's values carry weight in the gradients
        cot = {k: v / max(1e-30, float(ref[k].detach().abs().max())) for k, v in cot.items()}
    sum((ref[k] * cot[k]).sum() for k in CHARTS).backward()
    torch.autograd.backward([getattr(m, k) for k in CHARTS], [cot[k].cuda() for k in CHARTS])
    torch.cuda.synchronize()

    def rel(a, b):
        a = a.detach().double().cpu().flatten(); b = b.detach().double().cpu().flatten()
        sc = max(1e-30, float(b.abs().max()))
        d = (a - b).abs()
        return float(d.max()) / sc, float(d.median()) / sc
    outs = {k: rel(getattr(m, k), ref[k]) for k in CHARTS}
    named = dict(m.named_parameters())
    grads = {k: rel(named[k].grad, p.grad) for k, p in Pd.items() if p.grad is not None}
    grads['x_span'] = rel(xg.grad, xc.grad)
    return outs, grads


@pytest.mark.parametrize('L,seed', [(15, 1), (15, 2), (18, 1), (18, 2)])
def test_pinned_regime_no_normalization_on_long_charts(L, seed, mfma_mode):
    """normalize='none' (cliora/net/utils.py:17-29 'none') at L >= 15: chart values grow ~10x per level (1e17 at L 15, 5e19 at L 18,
    D 48) and every rounding of a low level is carried up.  Measured: exact-fp32 mode <= 8.3e-6 of scale (outputs), 4.6e-5 / 4.5e-6
    (gradients, max / median); split-bf16 mode 8.8e-5, 4.7e-4 / 4.6e-5 -- the fp32 oracle itself is 6.8e-6 and 1.9e-5 / 1.8e-6 from fp64."""
    outs, grads = _regime_case(48, 3, L, 'none', False, seed)
    out_tol, g_max, g_med = (4e-5, 2e-4, 2e-5) if mfma_mode == 'f32' else (4e-4, 2e-3, 2e-4)
    for k, (mx, _) in outs.items():
        assert mx <= out_tol, (k, mx)
    for k, (mx, med) in grads.items():
        assert mx <= g_max and med <= g_med, (k, mx, med)


@pytest.mark.parametrize('seed', [1, 2, 3, 4, 5, 6])
def test_pinned_regime_compress_at_small_width(seed, mfma_mode):
    """compress=True (diora.py:342-343) at d = 16, L = 9: the outside root is a projection of the inside root, so every outside value and
    every gradient hangs on one 16-wide ReLU layer.  Measured: exact-fp32 mode <= 1.0e-6 (outputs), 2.4e-6 / 5.0e-7 (gradients, max /
    median); split-bf16 mode 2.4e-5, 8.1e-5 / 2.9e-5."""
    outs, grads = _regime_case(16, 3, 9, 'unit', True, seed)
    out_tol, g_max, g_med = (5e-6, 1e-5, 2e-6) if mfma_mode == 'f32' else (1e-4, 4e-4, 1.2e-4)
    for k, (mx, _) in outs.items():
        assert mx <= out_tol, (k, mx)
    for k, (mx, med) in grads.items():
        assert mx <= g_max and med <= g_med, (k, mx, med)
