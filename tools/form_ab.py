#!/usr/bin/env python
"""One process = one setting of CLIORA_WGRAD_FORM (read once by the library): chart forward + backward at a shape, the SHA-1 of every parameter
gradient (the formed-tile weight gradient must be bitwise the materialised one) and ms per step.
  CLIORA_WGRAD_FORM=1 python tools/form_ab.py [--length 20 --batch 64 --share 1 --vl 0]"""
import argparse
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

ap = argparse.ArgumentParser()
ap.add_argument('--length', type=int, default=20)
ap.add_argument('--dim', type=int, default=400)
ap.add_argument('--batch', type=int, default=64)
ap.add_argument('--share', type=int, default=1)
ap.add_argument('--vl', type=int, default=0)
ap.add_argument('--steps', type=int, default=30)
a = ap.parse_args()
torch.manual_seed(1234)
if a.vl:
    from cliora_amd.cliora import DioraMLP
else:
    from cliora_amd.diora import DioraMLP
m = DioraMLP(a.dim, share=bool(a.share)).cuda()
if a.vl:
    m.eval()
x = torch.randn(a.batch, a.length, a.dim, device='cuda')
obj = 0.3 * torch.randn(a.batch, 36, a.dim, device='cuda') if a.vl else None
C = a.length * (a.length + 1) // 2
cots = [torch.randn(a.batch, C, w, device='cuda') for w in (a.dim, 1, a.dim, 1)]
keys = ('inside_h', 'inside_s', 'outside_h', 'outside_s')


def step():
    for p in m.parameters():
        p.grad = None
    m(x, x, obj, obj) if a.vl else m(x, x)
    torch.autograd.backward([getattr(m, k) for k in keys], cots)


step()
torch.cuda.synchronize()
for n, p in m.named_parameters():
    print('%-45s %s %.6e' % (n, hashlib.sha1(p.grad.detach().cpu().numpy().tobytes()).hexdigest()[:16], float(p.grad.abs().sum())))
for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    step()
torch.cuda.synchronize()
print('FORM=%s L=%d B=%d share=%d vl=%d: %.3f ms/step' % (os.environ.get('CLIORA_WGRAD_FORM', 'default'), a.length, a.batch, a.share, a.vl, (time.perf_counter() - t0) / a.steps * 1e3))
