"""Forward-only timing of the chart (eval / no_grad path) and of one training step, for kernel work:
python tools/fwd_bench.py [--length 20 --dim 400 --batch 64]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cliora_amd.diora import DioraMLP

ap = argparse.ArgumentParser()
ap.add_argument('--length', type=int, default=20)
ap.add_argument('--dim', type=int, default=400)
ap.add_argument('--batch', type=int, default=64)
ap.add_argument('--steps', type=int, default=30)
ap.add_argument('--only', default='', help="'fwd': the training-mode forward alone (for kernel traces)")
a = ap.parse_args()
torch.manual_seed(1234)
m = DioraMLP(a.dim).cuda()
x = torch.randn(a.batch, a.length, a.dim, device='cuda')
C = a.length * (a.length + 1) // 2
cots = [torch.randn(a.batch, C, w, device='cuda') for w in (a.dim, 1, a.dim, 1)]
keys = ('inside_h', 'inside_s', 'outside_h', 'outside_s')


def timeit(fn, n):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def fwd_nograd():
    with torch.no_grad():
        m(x, x)


def fwd_grad():
    m(x, x)


def step():
    for p in m.parameters():
        p.grad = None
    m(x, x)
    torch.autograd.backward([getattr(m, k) for k in keys], cots)


if a.only == 'fwd':
    print('forward (training) %.3f ms' % timeit(fwd_grad, a.steps))
    sys.exit(0)
print('forward (no_grad) %.3f ms | forward (training) %.3f ms | forward+backward %.3f ms' % (timeit(fwd_nograd, a.steps), timeit(fwd_grad, a.steps), timeit(step, a.steps)))
