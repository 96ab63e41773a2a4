"""Per-phase timeline of the persistent forward kernel (CLIORA_PERSIST_TRACE=1): for every phase, when the first / last workgroup
got through the wait, how long the work took (median / max over workgroups), and the phase's span on the wall clock.
CLIORA_PERSIST_TRACE=1 python tools/persist_trace.py [--dim 400 --batch 64 --length 20]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

os.environ.setdefault('CLIORA_PERSIST_TRACE', '1')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cliora_amd import _lib                      # noqa: E402
from cliora_amd.diora import DioraMLP            # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--dim', type=int, default=400)
ap.add_argument('--batch', type=int, default=64)
ap.add_argument('--length', type=int, default=20)
ap.add_argument('--grad', action='store_true')
a = ap.parse_args()
D, B, L = a.dim, a.batch, a.length
torch.manual_seed(1234)
m = DioraMLP(D).cuda()
x = torch.randn(B, L, D, device='cuda')
_lib.set_persistent('on')
for _ in range(5):
    if a.grad:
        m(x, x)
    else:
        with torch.no_grad():
            m(x, x)
torch.cuda.synchronize()
plan = _lib.get_plan(B, L, D, True, 'unit', 0, torch.cuda.current_device())
NW, NPH = 256, 2 * (L + 1)
buf = np.zeros(NW * NPH * 10, dtype=np.uint64)
_lib.check(_lib.lib().cliora_persistent_trace(plan.handle, buf.ctypes.data_as(C.c_void_p), buf.size, None), 'trace')
fine = buf[NW * NPH * 2:].reshape(NW, NPH, 8)
t = buf[:NW * NPH * 2].reshape(NW, NPH, 2).astype(np.float64) / 100.0          # us
t0 = t[:, 0, 0].min()
names = ['C', 'P']
print('slot  k   start(first)  start(last)   work med   work max    end(last)   busy WGs  [us since the first scores ended]')
tot = {n: 0.0 for n in names}
for k in range(1, L + 1):
    for sub in range(2):
        ph = 2 * k + sub
        s, e = t[:, ph, 0] - t0, t[:, ph, 1] - t0
        w = e - s
        print('%-2s  k=%2d   %10.2f   %10.2f   %8.2f   %8.2f   %10.2f   %4d' % (names[sub], k, s.min(), s.max(), np.median(w), w.max(), e.max(), (w > 0.5).sum()))
        tot[names[sub]] += w.max()
print('sum of max work per slot kind:', {k: round(v, 1) for k, v in tot.items()}, 'span %.1f us' % (t[:, :, 1].max() - t0))

if fine.max() > 0:      # diagnostic build (-DCLIORA_PERSIST_STAMPS): wave 0 of every workgroup times its tasks in the P slots, by kind
    print('P slots, wave 0 of each workgroup: mean us per task (tasks) for projection tiles | scores | chart rows | final rows; busiest wave total us')
    for k in range(1, L + 1):
        ph = 2 * k + 1
        f = fine[:, ph, :].astype(np.float64)
        tot, cnt = f[:, :4] / 100.0, f[:, 4:]
        cells = []
        for q in range(4):
            n = cnt[:, q].sum()
            cells.append('%6.2f (%5d)' % (tot[:, q].sum() / n, n) if n else '     - (    0)')
        print('k=%2d  ' % k + ' | '.join(cells) + '   busiest %6.2f' % tot.sum(1).max())
