#!/usr/bin/env python
"""ms/step (chart forward + backward) of the other BASELINE.json configurations on one GPU; prints one JSON line each.
  python tools/shapes.py            (on the MI355X box)"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cliora_amd import _lib                                  # noqa: E402


def run(name, make, B, L, D, R=0, steps=20, warmup=5):
    torch.manual_seed(0)
    dev = torch.device('cuda:0')
    m = make().to(dev).train()
    for p in m.parameters():
        torch.nn.init.normal_(p)
    x = torch.randn(B, L, D, device=dev, requires_grad=True)
    obj = 0.3 * torch.randn(B, R, D, device=dev) if R else None
    C = L * (L + 1) // 2
    keys = ('inside_h', 'inside_s', 'outside_h', 'outside_s')
    cot = {k: torch.randn(B, C, 1 if k.endswith('_s') else D, device=dev) for k in keys}

    def step():
        for p in m.parameters():
            p.grad = None
        x.grad = None
        if R:
            m(x, x, obj, obj)
        else:
            m(x, x)
        outs = [getattr(m, k) for k in keys]
        extra = []
        if R:
            extra = [m.all_atten_score.max(-1).values.sum() * 1e-3]
        torch.autograd.backward(outs + extra, [cot[k] for k in keys] + [None] * len(extra))

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(json.dumps(dict(shape=name, B=B, L=L, D=D, R=R, ms_per_step=round(dt * 1e3, 3), sentences_per_s=round(B / dt, 1))), flush=True)


if __name__ == '__main__':
    from cliora_amd.diora import DioraMLP
    from cliora_amd.cliora import DioraMLP as CDioraMLP
    from cliora_amd.treelstm import DioraTreeLSTM
    mode = _lib.set_mfma_mode('bf16x3')
    _lib.set_mfma_mode(mode)
    print(json.dumps(dict(mfma=mode)))
    only = sys.argv[1] if len(sys.argv) > 1 else ''
    cases = [('c1 DioraMLP', lambda: DioraMLP(50), 8, 10, 50, 0, 20, 5),
             ('c2 DioraMLP', lambda: DioraMLP(400), 64, 20, 400, 0, 20, 5),
             ('c3 CLIORA', lambda: CDioraMLP(400), 64, 20, 400, 36, 20, 5),
             ('DioraMLP len 40', lambda: DioraMLP(400), 64, 40, 400, 0, 8, 2),
             ('c5 DioraTreeLSTM len 20', lambda: DioraTreeLSTM(400), 64, 20, 400, 0, 20, 5),
             ('c5 DioraTreeLSTM len 40', lambda: DioraTreeLSTM(400), 64, 40, 400, 0, 8, 2)]
    # SHAPES_STEPS / SHAPES_WARMUP: short runs for the --pmc passes (tools/pmc_shape.sh counts steps + warmup launches of every kernel)
    ovs, ovw = os.environ.get('SHAPES_STEPS'), os.environ.get('SHAPES_WARMUP')
    for name, make, B, L, D, R, steps, warmup in cases:
        if only in name:
            run(name, make, B, L, D, R=R, steps=int(ovs) if ovs else steps, warmup=int(ovw) if ovw else warmup)
