#!/usr/bin/env python
"""Chart forward + backward ms/step over shapes, to place the wavefront switch (CLIORA_WAVEFRONT=0/1 from the caller's env).
  for w in 0 1; do CLIORA_WAVEFRONT=$w python tools/wavefront_sweep.py; done"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cliora_amd.diora import DioraMLP                       # noqa: E402

dev = torch.device('cuda:0')
SHAPES = ((16, 20, 400), (32, 20, 400), (64, 20, 400), (128, 20, 400), (256, 20, 400), (64, 40, 400), (32, 40, 400), (128, 30, 400), (64, 20, 48), (8, 10, 50))
series = []
for B, L, D in SHAPES:
    torch.manual_seed(0)
    m = DioraMLP(D).to(dev).train()
    for p in m.parameters():
        torch.nn.init.normal_(p)
    x = torch.randn(B, L, D, device=dev, requires_grad=True)
    C = L * (L + 1) // 2
    keys = ('inside_h', 'inside_s', 'outside_h', 'outside_s')
    cot = [torch.randn(B, C, 1 if k.endswith('_s') else D, device=dev) for k in keys]

    def step():
        for p in m.parameters():
            p.grad = None
        x.grad = None
        m(x, x)
        torch.autograd.backward([getattr(m, k) for k in keys], cot)
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    n = 15
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    R = B * (L - 1) * L * (L + 1) // 2
    if L == 20 and D == 400:
        series.append((B, (time.perf_counter() - t0) / n * 1e3))
    print('wavefront=%s B %3d L %2d D %3d: %8.3f ms/step   (%d pair rows, %.0f per level)' % (os.environ.get('CLIORA_WAVEFRONT', '1'), B, L, D, (time.perf_counter() - t0) / n * 1e3, R, R / (2 * (L - 1))), flush=True)
    del m, x, cot
    torch.cuda.empty_cache()

# the one figure that says which half of the step an idea can touch: a least-squares line ms(B) = latency + per_sentence * B over the L 20 / d 400 runs
if len(series) >= 2:
    nB = len(series)
    sx, sy = sum(b for b, _ in series), sum(t for _, t in series)
    sxx, sxy = sum(b * b for b, _ in series), sum(b * t for b, t in series)
    slope = (nB * sxy - sx * sy) / (nB * sxx - sx * sx)
    icpt = (sy - slope * sx) / nB
    print('L 20 / d 400: step ms = %.3f (dependent latency) + %.4f x B (throughput-bound work per sentence); at B 64: %.2f + %.2f ms'
          % (icpt, slope, icpt, slope * 64))
