#!/usr/bin/env python
"""Prints per-tensor errors of the HIP path vs golden vectors / the CPU oracle (no early exit).
Debug aid for GPU runs: `python tools/gpu_diag.py > gpurun_out/diag.txt`."""
import os
import sys
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from conftest import load_golden, params_from_golden   # noqa: E402
from test_gpu_parity import _module_from_params, _run_gpu, _err, _scale, CHARTS   # noqa: E402


def golden_case(name):
    g = load_golden(name)
    meta = g['meta']
    print('==== %s  D=%d B=%d L=%d share=%s norm=%s' % (name, meta['D'], meta['B'], meta['L'], meta['share'], meta['normalize']))
    P = params_from_golden(g)
    m = _module_from_params(P, meta['D'], meta['share'], meta['normalize'])
    cot = {k[5:]: torch.from_numpy(v) for k, v in g.items() if k.startswith('cot__')}
    outs, xg = _run_gpu(m, torch.from_numpy(g['x_span']), cot)
    L = meta['L']
    off = [L * (L + 1) // 2 - (L - lv) * (L - lv + 1) // 2 for lv in range(L)] + [L * (L + 1) // 2]
    for k in CHARTS:
        a = outs[k].detach().cpu().numpy()
        print('  %-10s err %.3e  scale %.3e  nan=%d' % (k, _err(a, g[k]), _scale(g[k]), int(np.isnan(a).sum())))
        per = ['%.1e' % np.abs(a[:, off[lv]:off[lv + 1]] - g[k][:, off[lv]:off[lv + 1]]).max() for lv in range(L)]
        print('     per level:', ' '.join(per))
    named = dict(m.named_parameters())
    for k, v in sorted(g.items()):
        if k.startswith('grad__'):
            name_ = k[6:].replace('__', '.')
            t = xg.grad if name_ == 'x_span' else named[name_].grad
            print('  %-45s err %.3e  scale %.3e' % (k, _err(t, v) if t is not None else float('nan'), _scale(v)))
    try:
        m.eval()
        with torch.no_grad():
            x = torch.from_numpy(g['x_span']).cuda()
            m(x, x)
        trees = [str(t) for t in m.cky()]
        print('  trees identical:', trees == meta['trees'])
    except Exception:
        traceback.print_exc()


if __name__ == '__main__':
    print(torch.__version__, torch.cuda.get_device_name(0))
    for n in ():
        try:
            golden_case(n)
        except Exception:
            traceback.print_exc()


def oracle_case(D, B, L, seed, share=True):
    from oracle import diora_ref as R
    from oracle import synth
    print('==== oracle case D=%d B=%d L=%d seed=%d share=%s' % (D, B, L, seed, share))
    P, x, cot = synth.diora_case(D, B, L, seed, share=share)
    m = _module_from_params(P, D, share, 'unit')
    outs, xg = _run_gpu(m, x, cot)
    P64 = {k: v.double().requires_grad_(True) for k, v in P.items()}
    x64 = x.double().requires_grad_(True)
    orig_full = torch.full
    torch.full = lambda shape, val, dtype=None, **kw: orig_full(shape, val, dtype=torch.float64, **kw)
    try:
        ref = R.diora_forward(P64, x64, x64, share=share, training=True)
    finally:
        torch.full = orig_full
    sum((ref[k] * cot[k].double()).sum() for k in CHARTS).backward()
    off = [L * (L + 1) // 2 - (L - lv) * (L - lv + 1) // 2 for lv in range(L)] + [L * (L + 1) // 2]
    for k in CHARTS:
        a = outs[k].detach().cpu().double().numpy()
        r = ref[k].detach().numpy()
        per = ['%.1e' % np.abs(a[:, off[lv]:off[lv + 1]] - r[:, off[lv]:off[lv + 1]]).max() for lv in range(L)]
        print('  %-10s err %.3e scale %.3e | per level: %s' % (k, np.abs(a - r).max(), np.abs(r).max(), ' '.join(per)))
    named = dict(m.named_parameters())
    for k, p in P64.items():
        gq = named[k].grad.cpu().double()
        e, s = float((gq - p.grad).abs().max()), float(p.grad.abs().max())
        print('  grad %-42s err %.3e scale %.3e rel %.2e' % (k, e, s, e / s))
        d = (gq - p.grad).abs()
        if d.dim() == 2:
            rowmax = d.max(1).values
            top = torch.topk(rowmax, min(3, rowmax.numel()))
            print('       median elem err %.2e; worst rows %s -> %s' % (float(d.median()), top.indices.tolist(), ['%.1e' % v for v in top.values.tolist()]))
        elif d.dim() == 1:
            top = torch.topk(d, min(3, d.numel()))
            print('       median elem err %.2e; worst idx %s -> %s' % (float(d.median()), top.indices.tolist(), ['%.1e' % v for v in top.values.tolist()]))
    e, s = float((xg.grad.cpu().double() - x64.grad).abs().max()), float(x64.grad.abs().max())
    print('  grad %-42s err %.3e scale %.3e rel %.2e' % ('x_span', e, s, e / s))


if __name__ == '__main__':
    for args in [(400, 2, 20, 1234), (400, 2, 20, 1235), (400, 2, 20, 1236), (400, 2, 20, 1237)]:
        try:
            oracle_case(*args)
        except Exception:
            traceback.print_exc()
