"""A/B of the two forward schedules (launch per level vs one persistent launch) over a few chart shapes: forward-only and
forward + backward ms.  python tools/persist_ab.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                  # noqa: E402
from cliora_amd import _lib                   # noqa: E402
from cliora_amd.diora import DioraMLP         # noqa: E402

SHAPES = [(50, 8, 10), (64, 16, 16), (64, 64, 20), (400, 8, 20), (400, 16, 20), (400, 64, 8), (400, 64, 12), (400, 64, 20), (400, 128, 20), (400, 16, 40), (400, 64, 40)]
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


print('D B L | forward ms: per-level, persistent | forward+backward ms: per-level, persistent')
for D, B, L in SHAPES:
    torch.manual_seed(1234)
    m = DioraMLP(D).cuda()
    x = torch.randn(B, L, D, device='cuda')
    C = L * (L + 1) // 2
    cots = [torch.randn(B, C, w, device='cuda') for w in (D, 1, D, 1)]

    def fwd():
        with torch.no_grad():
            m(x, x)

    def step():
        for p in m.parameters():
            p.grad = None
        m(x, x)
        torch.autograd.backward([getattr(m, k) for k in keys], cots)
    r = []
    for mode in ('off', 'on'):
        _lib.set_persistent(mode)
        r.append((timeit(fwd, 30), timeit(step, 20)))
    print('%3d %3d %2d | %7.3f %7.3f | %7.3f %7.3f' % (D, B, L, r[0][0], r[1][0], r[0][1], r[1][1]), flush=True)
