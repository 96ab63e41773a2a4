#!/usr/bin/env python
"""Is the chart step host-bound?  Per step, on an idle queue: host time to ENQUEUE forward + backward vs device time (events).
  python tools/host_bench.py [B L D]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cliora_amd.diora import DioraMLP                       # noqa: E402

B, L, D = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (64, 20, 400)
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = DioraMLP(D, share=True).to(dev).train() if 'share' in DioraMLP.__init__.__code__.co_varnames else DioraMLP(D).to(dev).train()
for p in m.parameters():
    torch.nn.init.normal_(p)
x = torch.randn(B, L, D, device=dev, requires_grad=True)
C = L * (L + 1) // 2
keys = ('inside_h', 'inside_s', 'outside_h', 'outside_s')
cot = [torch.randn(B, C, 1 if k.endswith('_s') else D, device=dev) for k in keys]


def fwd():
    for p in m.parameters():
        p.grad = None
    x.grad = None
    m(x, x)
    return [getattr(m, k) for k in keys]


for _ in range(10):
    torch.autograd.backward(fwd(), cot)
torch.cuda.synchronize()
hf, hb, dv = [], [], []
for _ in range(30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    outs = fwd()
    t1 = time.perf_counter()
    torch.autograd.backward(outs, cot)
    e1.record()
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    hf.append((t1 - t0) * 1e3); hb.append((t2 - t1) * 1e3); dv.append(e0.elapsed_time(e1))
med = lambda v: sorted(v)[len(v) // 2]
print('B %d L %d D %d: host enqueue forward %.3f ms, backward %.3f ms (sum %.3f); device %.3f ms' % (B, L, D, med(hf), med(hb), med(hf) + med(hb), med(dv)))
