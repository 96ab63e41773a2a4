#!/usr/bin/env python
"""The two regimes tools/fuzz_parity.py keeps flagging, measured so that tests can pin them with explicit bounds:
  (i)  normalize='none' at L >= 15 in split-bf16 mode: chart values grow by ~10x per level (1e20 at the root)
  (ii) compress=True at d = 16: one ReLU on the fence moves every gradient
Prints, per case, the output error relative to the tensor's scale and the gradients' median / max error relative to scale."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cliora_amd import _lib                                  # noqa: E402
from cliora_amd.diora import DioraMLP                        # noqa: E402
from oracle import diora_ref as R                            # noqa: E402

KEYS = ('inside_h', 'inside_s', 'outside_h', 'outside_s')


def rel(a, b):
    a = a.detach().double().cpu().flatten(); b = b.detach().double().cpu().flatten()
    sc = max(1e-30, float(b.abs().max()))
    d = (a - b).abs()
    return float(d.max()) / sc, float(d.median()) / sc, sc


def case(D, B, L, normalize, compress, mode, seed, dtype=torch.float32):
    _lib.set_mfma_mode(mode)
    P = R.init_params(D, share=True, seed=seed, compress=compress)
    m = DioraMLP(D, share=True, normalize=normalize, compress=compress)
    sd = m.state_dict()
    for k in sd:
        sd[k] = P[k if k in P else 'inside_' + k[len('outside_'):]].detach().clone()
    m.load_state_dict(sd)
    m = m.cuda().train()
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, L, D, generator=g)
    xg = x.clone().cuda().requires_grad_(True)
    m(xg, xg)
    Pd = {k: v.detach().to(dtype).requires_grad_(True) for k, v in P.items()}
    xc = x.clone().to(dtype).requires_grad_(True)
    ref = R.diora_forward(Pd, xc, xc, share=True, normalize=normalize, training=True)
    cot = {k: torch.randn(ref[k].shape, generator=g) for k in KEYS}
    if normalize == 'none':       # cotangents scaled to the outputs so that every level contributes
        cot = {k: v / max(1e-30, float(ref[k].detach().abs().max())) for k, v in cot.items()}
    sum((ref[k] * cot[k].to(dtype)).sum() for k in KEYS).backward()
    torch.autograd.backward([getattr(m, k) for k in KEYS], [cot[k].cuda() for k in KEYS])
    torch.cuda.synchronize()
    out = {k: rel(getattr(m, k), ref[k]) for k in KEYS}
    named = dict(m.named_parameters())
    gr = {k: rel(named[k].grad, p.grad) for k, p in Pd.items() if p.grad is not None}
    gr['x'] = rel(xg.grad, xc.grad)
    return out, gr


if __name__ == '__main__':
    for mode in ('f32', 'bf16x3'):
        for (D, L) in ((48, 12), (48, 15), (48, 18), (64, 20), (400, 15), (400, 20)):
            for seed in (1, 2):
                o, g = case(D, 3, L, 'none', False, mode, seed)
                o64, g64 = case(D, 3, L, 'none', False, mode, seed, torch.float64)
                print('none %-6s D %3d L %2d seed %d | out max %.2e (scale %.1e) vs fp64 %.2e | grad max %.2e med %.2e vs fp64 max %.2e med %.2e' % (
                    mode, D, L, seed, max(v[0] for v in o.values()), max(v[2] for v in o.values()), max(v[0] for v in o64.values()),
                    max(v[0] for v in g.values()), max(v[1] for v in g.values()), max(v[0] for v in g64.values()), max(v[1] for v in g64.values())), flush=True)
    for mode in ('f32', 'bf16x3'):
        for D in (16, 32):
            for seed in range(1, 7):
                o, g = case(D, 3, 9, 'unit', True, mode, seed)
                o64, g64 = case(D, 3, 9, 'unit', True, mode, seed, torch.float64)
                print('compress %-6s D %2d seed %d | out max %.2e | grad max %.2e med %.2e vs fp64 max %.2e med %.2e' % (
                    mode, D, seed, max(v[0] for v in o.values()), max(v[0] for v in g.values()), max(v[1] for v in g.values()),
                    max(v[0] for v in g64.values()), max(v[1] for v in g64.values())), flush=True)
