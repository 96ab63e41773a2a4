"""A/B of the level-loop schedules for small hidden sizes: launches per level vs the
sentence-resident kernels (csrc/resident_kernels.hpp), forward-only and forward + backward ms over a few chart shapes.
python tools/resident_ab.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                  # noqa: E402
from cliora_amd import _lib                   # noqa: E402
from cliora_amd.diora import DioraMLP         # noqa: E402

import ast
SHAPES_ENV = os.environ.get('RESIDENT_AB_SHAPES')
SHAPES = ast.literal_eval(SHAPES_ENV) if SHAPES_ENV else [(50, 8, 10), (50, 1, 10), (50, 64, 10), (50, 256, 10), (64, 16, 16), (64, 64, 16), (64, 64, 20), (64, 256, 20), (64, 8, 30), (64, 64, 30), (64, 8, 40), (32, 64, 12), (16, 128, 8)]
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


print('D B L pairs/sentence | forward ms: launches, resident | forward+backward ms: launches, resident')
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
        _lib.set_resident(mode)
        r.append((timeit(fwd, 30), timeit(step, 20)))
    _lib.set_resident('auto')
    print('%3d %3d %2d %5d | %7.3f %7.3f | %7.3f %7.3f' % (D, B, L, (L - 1) * L * (L + 1) // 2, r[0][0], r[1][0], r[0][1], r[1][1]), flush=True)
