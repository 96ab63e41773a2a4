"""Phase times inside one rows-stationary compose launch (diagnostic build: CLIORA_BUILD_EXTRA=-DCLIORA_RS_STAMPS python -m
cliora_amd.build): every workgroup stamps its first task -- start, operands gathered, after each third of each column block, end.
  CLIORA_RS_TRACE_ROWS=25600 python tools/rs_trace.py [--batch 64 --length 40] [--no-outside]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cliora_amd import _lib                      # noqa: E402
from cliora_amd.diora import DioraMLP            # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=64)
ap.add_argument('--length', type=int, default=40)
ap.add_argument('--no-outside', action='store_true')
a = ap.parse_args()
B, L, D = a.batch, a.length, 400
torch.manual_seed(1234)
m = DioraMLP(D, outside=not a.no_outside).cuda()
x = torch.randn(B, L, D, device='cuda')
_lib.set_rows_stationary('on')
for _ in range(4):
    with torch.no_grad():
        m(x, x)
torch.cuda.synchronize()
plan = _lib.get_plan(B, L, D, True, 'unit', 0, torch.cuda.current_device())
NW = 256
buf = np.zeros(NW * 32, dtype=np.uint64)
_lib.check(_lib.lib().cliora_persistent_trace(plan.handle, buf.ctypes.data_as(C.c_void_p), buf.size, None), 'trace')
t = buf.reshape(NW, 32).astype(np.float64) / 100.0          # us
ok = t[:, 0] > 0
t = t[ok]
print('workgroups with a stamp: %d' % ok.sum())
t0 = t[:, 0].min()
names = ['start', 'gathered'] + ['cb%d third %d' % (c, k) for c in range(5) for k in range(3)] + ['end']
prev = None
for k, nm in enumerate(names):
    col = t[:, k]
    d = (col - t[:, k - 1]) if k else col - t0
    print('%-14s at %7.2f .. %7.2f us   step median %6.2f  max %6.2f' % (nm, col.min() - t0, col.max() - t0, np.median(d), d.max()))
