#!/usr/bin/env python
"""Error of the HIP chart against the CPU oracle on the d=400 / L=20 shape (B=2), for the arithmetic mode
selected by CLIORA_MFMA (f32 | bf16x3).  Prints one JSON line; used for the accuracy table in DESIGN.md."""
import json
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from oracle import diora_ref as R, synth          # noqa: E402  (checker only)
from test_gpu_parity import _module_from_params, _run_gpu, CHARTS   # noqa: E402

D, B, L, seed = int(os.environ.get('D', '400')), int(os.environ.get('B', '2')), int(os.environ.get('L', '20')), int(os.environ.get('SEED', '1234'))
P, x, cot = synth.diora_case(D, B, L, seed)
m = _module_from_params(P, D, True, 'unit')
outs, xg = _run_gpu(m, x, cot)
for v in P.values():
    v.requires_grad_(True)
xc = x.clone().requires_grad_(True)
ref = R.diora_forward(P, xc, xc, training=True, keep_pairs=True)
sum((ref[k] * cot[k]).sum() for k in CHARTS).backward()
res = {'mode': os.environ.get('CLIORA_MFMA', 'bf16x3'), 'seed': seed, 'B': B}
for k in CHARTS:
    d = (outs[k].detach().cpu().double() - ref[k].detach().double()).abs()
    res[k] = {'max_abs': float(d.max()), 'scale': float(ref[k].abs().max())}
named = dict(m.named_parameters())
grads = {}
for k, p in list(P.items()) + [('x_span', xc)]:
    a = (named[k].grad if k in named else xg.grad).detach().cpu().double().flatten()
    b = p.grad.detach().double().flatten()
    d = (a - b).abs()
    sc = max(1.0, float(b.abs().max()))
    grads[k] = {'q50': float(d.median()) / sc, 'q99': float(torch.quantile(d[:: max(1, d.numel() // 200000)], 0.99)) / sc, 'max': float(d.max()) / sc}
res['grads_rel'] = grads
print(json.dumps(res))
