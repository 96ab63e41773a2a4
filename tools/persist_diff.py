"""Where do the persistent and the launch-per-level forward differ?  Compares the forward workspace region by region, level by level.
python tools/persist_diff.py [--dim 400 --batch 64 --length 20 --mfma f32]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cliora_amd import _lib                      # noqa: E402
from cliora_amd.diora import DioraMLP            # noqa: E402
from oracle import synth                         # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--dim', type=int, default=400)
ap.add_argument('--batch', type=int, default=64)
ap.add_argument('--length', type=int, default=20)
ap.add_argument('--mfma', default='f32')
ap.add_argument('--noshare', action='store_true')
a = ap.parse_args()
D, B, L = a.dim, a.batch, a.length
share = not a.noshare
_lib.set_mfma_mode(a.mfma)
P, x, cot = synth.diora_case(D, B, L, 4242, share=share)
m = DioraMLP(D, share=share)
sd = m.state_dict()
for k in sd:
    src = 'inside_' + k[len('outside_'):] if share and k.startswith('outside_') else k
    sd[k] = P[src].detach().clone()
m.load_state_dict(sd)
m = m.cuda()
plan = _lib.get_plan(B, L, D, share, 'unit', 0, torch.cuda.current_device())
res = {}
for mode in ('off', 'on'):
    _lib.set_persistent(mode)
    xg = x.clone().cuda().requires_grad_(True)
    m.train()
    m(xg, xg)
    torch.cuda.synchronize()
    ws = m._wss[0] if m._wss else None
    res[mode] = dict(ws=ws.clone().view(torch.float32).cpu().numpy() if ws is not None else None,
                     **{k: getattr(m, k).detach().cpu().numpy() for k in ('inside_h', 'inside_s', 'outside_h', 'outside_s')})
print('timeouts', _lib.persistent_timeouts(plan))
C = L * (L + 1) // 2
Dp = (D + 15) // 16 * 16
off = lambda n: _lib.lib().cliora_plan_fwd_offset(plan.handle, n.encode())
lev_off = [C - (L - l) * (L - l + 1) // 2 for l in range(L)]
nblk = 3 if share else 5


def rows(arr, level, width):
    v = arr.reshape(B, C, width)
    return v[:, lev_off[level]:lev_off[level] + (L - level)]


for name in ('inside_h', 'inside_s', 'outside_h', 'outside_s'):
    w = D if name.endswith('_h') else 1
    for lv in range(L):
        d = np.abs(rows(res['off'][name], lv, w) - rows(res['on'][name], lv, w))
        if d.max() > 0:
            print('%-10s level %2d: max diff %.3e, %d of %d elements differ' % (name, lv, d.max(), (d > 0).sum(), d.size))
w0, w1 = res['off']['ws'], res['on']['ws']
if w0 is not None:
    for name, width in (('pi', nblk * Dp), ('po', Dp), ('nrmi', 1), ('nrmo', 1)):
        o = off(name)
        a0 = w0[o:o + B * C * width]; a1 = w1[o:o + B * C * width]
        for lv in range(L):
            d = np.abs(rows(a0, lv, width) - rows(a1, lv, width))
            d = np.nan_to_num(d, nan=0.0)
            if d.max() > 0:
                print('%-10s level %2d: max diff %.3e, %d of %d' % (name, lv, d.max(), (d > 0).sum(), d.size))
    for name in ('hp', 'hp_o'):
        o = off(name)
        for s in range(4):
            a0 = w0[o + s * B * C * Dp:o + (s + 1) * B * C * Dp]; a1 = w1[o + s * B * C * Dp:o + (s + 1) * B * C * Dp]
            for lv in range(L):
                d = np.nan_to_num(np.abs(rows(a0, lv, Dp) - rows(a1, lv, Dp)), nan=0.0, posinf=0.0)
                if d.max() > 0:
                    print('%-10s part %d level %2d: max diff %.3e, %d of %d' % (name, s, lv, d.max(), (d > 0).sum(), d.size))
    R = B * (L - 1) * L * (L + 1) // 2
    for name in ('sp', 'pp'):
        o = off(name)
        d = np.abs(w0[o:o + R] - w1[o:o + R])
        nz = np.nonzero(d)[0]
        print('%-10s max diff %.3e, %d of %d differ%s' % (name, d.max(), nz.size, R, (', first row %d' % nz[0]) if nz.size else ''))
