"""The level loop as ONE persistent launch (csrc/persist_kernels.hpp; include/cliora_chart.h: cliora_set_persistent) against the
launch-per-level path on the same inputs: the persistent kernel performs the arithmetic of level_compose_fwd / level_project /
score_cell / level_finish in the same order, so every chart, every per-split score and every gradient must agree to the BIT --
and a hand-off that reads a stale line (a missing barrier, a load that hits L1) shows up here as a difference.

The parity of the launch-per-level path itself (golden vectors of the reference, the CPU oracle) is tests/test_gpu_parity.py;
those tests run with the persistent kernel on (the default), so they pin it against the reference too.
"""
import pytest
import torch

from test_gpu_parity import CHARTS, _module_from_params, _run_gpu

from cliora_amd import _lib as _lib_for_skip

# round 5: the persistent forward is an optional part of the build (AUTO selects it for no BASELINE configuration): these tests run
# when the library was built with -DCLIORA_WITH_PERSISTENT
pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not _lib_for_skip.has_persistent(), reason='library built without -DCLIORA_WITH_PERSISTENT')]

SHAPES = [
    # D, B, L, share, normalize
    (400, 64, 20, True, 'unit'),      # BASELINE configs[1]
    (50, 8, 10, True, 'unit'),        # configs[0]: padded rows (Dp = 64), one resident weight block
    (96, 16, 12, False, 'unit'),      # unshared weights: inside and outside compose on different workgroups
    (48, 5, 9, True, 'none'),
    (64, 3, 2, True, 'unit'),         # shortest chart with a level
    (64, 3, 3, False, 'unit'),
    (400, 2, 40, True, 'unit'),       # long chart, few sentences: every level is split in parts
    (400, 96, 13, True, 'unit'),
]


def _grads(m):
    return {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}


@pytest.mark.parametrize('D,B,L,share,normalize', SHAPES)
def test_persistent_forward_is_bitwise_the_launch_per_level_path(D, B, L, share, normalize, mfma_mode):
    from cliora_amd import _lib
    from oracle import synth
    P, x, cot = synth.diora_case(D, B, L, 4242, share=share)
    m = _module_from_params(P, D, share, normalize)
    prev = _lib.set_persistent('off')
    try:
        outs0, xg0 = _run_gpu(m, x, cot)
        outs0 = {k: v.detach().clone() for k, v in outs0.items()}
        g0 = _grads(m)
        _lib.set_persistent('on')
        for rep in range(3):
            for p_ in m.parameters():
                p_.grad = None
            outs1, xg1 = _run_gpu(m, x, cot)
            for k in CHARTS:
                assert torch.equal(outs0[k], outs1[k]), (k, rep, float((outs0[k] - outs1[k]).abs().max()))
            g1 = _grads(m)
            for n in g0:
                assert torch.equal(g0[n], g1[n]), (n, rep, float((g0[n] - g1[n]).abs().max()))
            assert torch.equal(xg0.grad, xg1.grad), rep
        plan = _lib.get_plan(B, L, D, share, normalize, 0, torch.cuda.current_device())
        assert _lib.persistent_timeouts(plan) == 0
    finally:
        _lib.set_persistent(prev)


@pytest.mark.parametrize('seed', range(6))
def test_persistent_random_shapes(seed, mfma_mode):
    """Random chart shapes (widths with and without padding, one to eight resident weight blocks, odd batch sizes, L up to the
    64-split limit of the score tasks): forward outputs and the per-split state the backward reads, bit for bit."""
    import random
    from cliora_amd import _lib
    from oracle import synth
    rnd = random.Random(1000 + seed)
    D = rnd.choice([16, 24, 33, 64, 80, 128, 200, 256, 400, 512])
    L = rnd.choice([2, 3, 5, 9, 14, 23, 31, 64]) if D <= 128 else rnd.choice([2, 4, 7, 12, 18])
    B = rnd.choice([1, 2, 3, 7, 16]) if L > 20 else rnd.choice([1, 5, 16, 37])
    share = rnd.random() < 0.7
    normalize = 'unit' if rnd.random() < 0.8 else 'none'
    P, x, cot = synth.diora_case(D, B, L, 500 + seed, share=share)
    m = _module_from_params(P, D, share, normalize)
    prev = _lib.set_persistent('off')
    try:
        outs0, xg0 = _run_gpu(m, x, cot)
        outs0 = {k: v.detach().clone() for k, v in outs0.items()}
        g0 = _grads(m)
        _lib.set_persistent('on')
        for p_ in m.parameters():
            p_.grad = None
        outs1, xg1 = _run_gpu(m, x, cot)
        for k in CHARTS:
            assert torch.equal(outs0[k], outs1[k]), (D, B, L, share, normalize, k)
        g1 = _grads(m)
        for n in g0:
            assert torch.equal(g0[n], g1[n]), (D, B, L, share, normalize, n)
        assert torch.equal(xg0.grad, xg1.grad)
        plan = _lib.get_plan(B, L, D, share, normalize, 0, torch.cuda.current_device())
        assert _lib.persistent_timeouts(plan) == 0
    finally:
        _lib.set_persistent(prev)


def test_persistent_inside_only(mfma_mode):
    """outside = False (scripts/train.py:130 at eval): only the inside chain runs in the kernel; the outside charts stay zero."""
    from cliora_amd import _lib
    from oracle import synth
    D, B, L = 48, 3, 7
    P, x, cot = synth.diora_case(D, B, L, 5)
    cot = {k: cot[k] for k in ('inside_h', 'inside_s')}
    m = _module_from_params(P, D, True, 'unit', outside=False)
    prev = _lib.set_persistent('off')
    try:
        outs0, xg0 = _run_gpu(m, x, cot)
        outs0 = {k: v.detach().clone() for k, v in outs0.items()}
        _lib.set_persistent('on')
        for p_ in m.parameters():
            p_.grad = None
        outs1, xg1 = _run_gpu(m, x, cot)
        for k in CHARTS:
            assert torch.equal(outs0[k], outs1[k]), k
        assert torch.equal(xg0.grad, xg1.grad)
        assert float(outs1['outside_h'].abs().max()) == 0.0
    finally:
        _lib.set_persistent(prev)


def test_persistent_no_grad_and_hooks(mfma_mode):
    """torch.no_grad() (CLIORA_FWD_NO_BACKWARD: no ReLU bits written) and an overridden hook (per-pair states written) take the same
    kernel with different outputs enabled."""
    import types
    from cliora_amd import _lib
    from oracle import synth
    D, B, L = 64, 4, 8
    P, x, _ = synth.diora_case(D, B, L, 11)
    m = _module_from_params(P, D, True, 'unit')
    seen = {}

    def hook(self, level, h, c, s):
        seen[level] = (h.detach().clone(), s.detach().clone())
    m.inside_hook = types.MethodType(hook, m)
    res = {}
    prev = _lib.set_persistent('off')
    try:
        for mode in ('off', 'on'):
            _lib.set_persistent(mode)
            seen.clear()
            with torch.no_grad():
                xg = x.clone().cuda()
                m.eval()
                m(xg, xg)
                torch.cuda.synchronize()
            res[mode] = ({k: getattr(m, k).detach().clone() for k in CHARTS}, dict(seen))
        for k in CHARTS:
            assert torch.equal(res['off'][0][k], res['on'][0][k]), k
        assert set(res['off'][1]) == set(res['on'][1]) == set(range(1, L))
        for lv in res['off'][1]:
            assert torch.equal(res['off'][1][lv][0], res['on'][1][lv][0]), ('pair states', lv)
            assert torch.equal(res['off'][1][lv][1], res['on'][1][lv][1]), ('pair scores', lv)
    finally:
        _lib.set_persistent(prev)


def test_a_persistent_timeout_fails_the_next_call_loudly():
    """a persistent launch that gives up on a grid barrier returns with its chart partly written and only counts that
    in a device word.  The word now follows every persistent launch to the host, and the next library call on the device raises
    instead of training on garbage.  The give-up is injected (cliora_persistent_inject_timeout): nothing really times out here."""
    import ctypes as C
    from cliora_amd import _lib
    from oracle import synth
    D, B, L = 64, 4, 16
    P, x, cot = synth.diora_case(D, B, L, 5)
    m = _module_from_params(P, D, True, 'unit')
    prev, prev_r = _lib.set_persistent('on'), _lib.set_resident('off')
    try:
        _run_gpu(m, x, cot)                                   # a clean persistent step
        plan = _lib.get_plan(B, L, D, True, 'unit', 0, torch.cuda.current_device())
        _lib.check(_lib.lib().cliora_persistent_inject_timeout(plan.handle, C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'inject')
        xg = x.clone().cuda().requires_grad_(True)
        m(xg, xg)                                             # this launch carries the moved word to the host
        torch.cuda.synchronize()
        with pytest.raises(_lib.ChartLibError, match='gave up'):
            torch.autograd.backward([m.inside_h], [cot['inside_h'].cuda()])
        for p_ in m.parameters():
            p_.grad = None
        _run_gpu(m, x, cot)                                   # reported once: the next step is clean again
    finally:
        _lib.set_persistent(prev)
        _lib.set_resident(prev_r)
