#!/usr/bin/env python
"""Chart forward + backward of one seeded DioraMLP case, every gradient saved to an .npz -- run once per setting of a library switch
that is read once per process (tests/test_gpu_gemm3.py compares CLIORA_BWD_GEMM3=0 with the default and other tile shapes).
   python tools/gemm3_probe.py D B L share out.npz"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import synth                                     # noqa: E402  (inputs only: the seeded case generator)
from cliora_amd.diora import DioraMLP                        # noqa: E402

D, B, L, share = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), bool(int(sys.argv[4]))
P, x, cot = synth.diora_case(D, B, L, 23, share=share)
m = DioraMLP(D, outside=True, normalize='unit', compress=False, share=share)
sd = m.state_dict()
for k in sd:
    sd[k] = P[('inside_' + k[len('outside_'):]) if (share and k.startswith('outside_')) else k].detach().clone()
m.load_state_dict(sd)
m = m.cuda().train()
xg = x.clone().cuda().requires_grad_(True)
m(xg, xg)
keys = ('inside_h', 'inside_s', 'outside_h', 'outside_s')
torch.autograd.backward([getattr(m, k) for k in keys], [cot[k].cuda() for k in keys])
torch.cuda.synchronize()
out = {'x': xg.grad.cpu().numpy()}
for k, p_ in m.named_parameters():
    if p_.grad is not None:
        out[k] = p_.grad.cpu().numpy()
for k in keys:
    out['out_' + k] = getattr(m, k).detach().cpu().numpy()
np.savez(sys.argv[5], **out)
