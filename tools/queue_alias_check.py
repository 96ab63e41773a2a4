#!/usr/bin/env python
"""Does the wavefront survive other streams in the process?  Creates K torch streams BEFORE the library picks its side streams
(HIP deals streams round-robin onto hardware queues), optionally runs the chart on one of them, and times the c2 step.
  python tools/queue_alias_check.py K [use_stream]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cliora_amd.diora import DioraMLP                       # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 0
use = len(sys.argv) > 2
dev = torch.device('cuda:0')
streams = [torch.cuda.Stream() for _ in range(K)]
for s_ in streams:                                           # make sure the runtime really creates them
    with torch.cuda.stream(s_):
        torch.zeros(1, device=dev)
B, L, D = 64, 20, 400
torch.manual_seed(0)
m = DioraMLP(D).to(dev).train()
for p in m.parameters():
    torch.nn.init.normal_(p)
x = torch.randn(B, L, D, device=dev, requires_grad=True)
C = L * (L + 1) // 2
keys = ('inside_h', 'inside_s', 'outside_h', 'outside_s')
cot = [torch.randn(B, C, 1 if k.endswith('_s') else D, device=dev) for k in keys]
torch.cuda.synchronize()
ctx = torch.cuda.stream(streams[-1]) if (use and streams) else torch.cuda.stream(torch.cuda.current_stream())
with ctx:
    def step():
        for p in m.parameters():
            p.grad = None
        x.grad = None
        m(x, x)
        torch.autograd.backward([getattr(m, k) for k in keys], cot)
    for _ in range(6):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(40):
        step()
    torch.cuda.synchronize()
print('%d other streams%s: %.3f ms/step' % (K, ' (chart on the last one)' if use and streams else '', (time.perf_counter() - t0) / 40 * 1e3))
