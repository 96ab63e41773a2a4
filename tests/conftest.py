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
    """Tolerance on the MEDIAN element error of a gradient, relative to the tensor's largest reference magnitude:
    `strict` for exact fp32 products, 5e-4 in split-bf16 mode (measured medians against fp64 at B 64 / L 20 / D 400: 1e-6 ... 1.3e-4,
    profiles/r02_accuracy_fp64_bf16x3.json; three times the worst)."""
    return strict if mode == 'f32' else max(strict, 5e-4)


def grad_check(t, ref, mode, strict, what='', full_size=False):
    """Gradient parity.

    Exact-product mode ('f32'): every element within `strict` of the tensor's scale.

    Split-bf16 mode: the compose GEMMs round their operands to 16 significant bits, which perturbs the second ReLU's
    pre-activation by ~1e-5 of its scale.  Elements within that distance of zero change sign (a few dozen of the
    ~2 M per step at B 32, L 14), and each such flip switches that unit's gradient path on or off -- the network is not
    differentiable there, the forward value barely moves, but single rows / columns of the weight gradients and single
    positions of dx move by up to a few percent of the tensor's scale (the reference's own fp32 run shows the same
    against fp64, at a lower rate: tests/test_gpu_configs.py::test_c2_full_size_gradients_vs_fp64 measures both).  So the check is
    statistical: median error within grad_tol(), 99 % of the elements within 1e-2 of scale, every element within 5e-2
    (measured at full size: 2e-3 and 7e-3; the small widths of the other tests sit higher, 3.7e-2 at D = 48).  A wrong kernel (a dropped
    k-step, a transposed tile) moves the median by orders of magnitude and fails the first bound.

    full_size=True (the d = 400 configurations at their own size): the bounds are what profiles/r02_accuracy_fp64_bf16x3.json shows
    there with a margin of two -- median 2e-4 (measured <= 1.3e-4), 99th percentile 4e-3 (<= 2e-3), maximum 2e-2 (<= 7.3e-3, set by
    ReLU kinks that the fp32 reference shows against fp64 as well)."""
    import torch
    a = (t.detach().double().cpu() if isinstance(t, torch.Tensor) else torch.as_tensor(np.asarray(t)).double()).flatten()
    b = (ref.detach().double().cpu() if isinstance(ref, torch.Tensor) else torch.as_tensor(np.asarray(ref)).double()).flatten()
    d = (a - b).abs()
    scale = max(1.0, float(b.abs().max())) if b.numel() else 1.0
    if mode == 'f32':
        assert float(d.max()) <= strict * scale, '%s: max err %.3e scale %.3e' % (what, float(d.max()), scale)
        return
    sub = d[:: max(1, d.numel() // 200000)]
    med, q99 = float(sub.median()), float(torch.quantile(sub, 0.99))
    tol_med, tol_q99, tol_max = (2e-4, 4e-3, 2e-2) if full_size else (grad_tol(mode, strict), 1e-2, 5e-2)
    assert med <= tol_med * scale, '%s: median err %.3e scale %.3e' % (what, med, scale)
    if d.numel() >= 1000:          # on a 48-element bias the 99th percentile IS the maximum: the last bound covers it
        assert q99 <= tol_q99 * scale, '%s: q99 err %.3e scale %.3e' % (what, q99, scale)
    assert float(d.max()) <= tol_max * scale, '%s: max err %.3e scale %.3e' % (what, float(d.max()), scale)
