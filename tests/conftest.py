import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    d = {k: z[k] for k in z.files}
    if 'meta' in d:
        d['meta'] = json.loads(str(d['meta']))
    return d


def params_from_golden(g, prefix='param__', as_torch=True, requires_grad=False):
    import torch
    out = {}
    for k, v in g.items():
        if k.startswith(prefix):
            name = k[len(prefix):].replace('__', '.')
            t = torch.from_numpy(v.copy())
            if requires_grad and t.is_floating_point():
                t.requires_grad_(True)
            out[name] = t if as_torch else v
    return out


@pytest.fixture(scope='session')
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]
    return get


MFMA_MODES = ('f32', 'bf16x3')


@pytest.fixture(params=MFMA_MODES)
def mfma_mode(request):
    """Run a GPU test under both arithmetic modes of the compose GEMMs (include/cliora_chart.h: cliora_set_mfma_mode).
    'f32' is the reference's arithmetic (strict gradient tolerance); 'bf16x3' is the default fast mode."""
    from cliora_amd import _lib
    prev = _lib.set_mfma_mode(request.param)
    yield request.param
    _lib.set_mfma_mode(prev)


def grad_tol(mode, strict):
    """Gradient tolerance relative to the tensor's largest reference magnitude: `strict` for exact fp32 products; the
    split-bf16 mode rounds every GEMM operand to 16 significant bits (2^-18), which the chart recursion amplifies to
    ~1e-3 of scale in the worst element (median ~1e-5; tools/accuracy.py)."""
    return strict if mode == 'f32' else max(strict, 2e-3)


def grad_check(t, ref, mode, strict, what=''):
    """Gradient parity.  Exact-product mode: every element within `strict` of the tensor's scale.  Split-bf16 mode:
    99 % of the elements within grad_tol(), every element within 100 x strict -- the second ReLU's pre-activation is
    perturbed by ~1e-5, and an element that changes sign there switches one row / column of a weight gradient by a
    finite amount (the network is not differentiable at the kink; see test_gpu_parity._grad_ok for the same effect
    between the reference's own fp32 and fp64 runs)."""
    import torch
    a = (t.detach().double().cpu() if isinstance(t, torch.Tensor) else torch.as_tensor(np.asarray(t)).double()).flatten()
    b = (ref.detach().double().cpu() if isinstance(ref, torch.Tensor) else torch.as_tensor(np.asarray(ref)).double()).flatten()
    d = (a - b).abs()
    scale = max(1.0, float(b.abs().max())) if b.numel() else 1.0
    if mode == 'f32':
        assert float(d.max()) <= strict * scale, '%s: max err %.3e scale %.3e' % (what, float(d.max()), scale)
        return
    q = float(torch.quantile(d[:: max(1, d.numel() // 200000)], 0.99))
    assert q <= grad_tol(mode, strict) * scale, '%s: q99 err %.3e scale %.3e' % (what, q, scale)
    assert float(d.max()) <= 100 * strict * scale, '%s: max err %.3e scale %.3e' % (what, float(d.max()), scale)
