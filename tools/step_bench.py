#!/usr/bin/env python
"""Whole training step (SURVEY section 8(d) item ii): Embed -> chart -> losses -> backward -> clip 5.0 -> Adam, on the
harness (cliora_amd/harness.py: torch ops around the native chart), synthetic inputs of SURVEY's shapes (V = 10 000,
1024-d embeddings, k_neg = 100, 36 x 2048 region features).  Prints one JSON line per configuration.
  python tools/step_bench.py            (on the MI355X box)"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cliora_amd import harness as H                        # noqa: E402


def run(name, vl, B=64, L=20, D=400, V=10000, E=1024, K=100, steps=20, warmup=5):
    torch.manual_seed(1234)
    emb = torch.nn.Embedding(V, E)
    net = H.build_net(D, emb, obj_feats=vl, img_dim=2048, k_neg=K, vg_loss=vl, use_contr=vl).cuda()
    if vl:
        for p in net.img_encoder.parameters():
            torch.nn.init.normal_(p, std=0.02)             # the reference's zero init makes every VL score 0
    tr = H.Trainer(net, lr=2e-3)
    g = torch.Generator().manual_seed(1234)
    bm = dict(sentences=torch.randint(0, V, (B, L), generator=g).cuda(), neg_samples=torch.randperm(V, generator=g)[:K].cuda())
    if vl:
        bm['obj_feats'] = torch.randn(B, 36, 2048, generator=g).cuda()
    for sync in (True, False):           # True: loss.item() every step like the reference (trainer.py:463); False: the host runs ahead
        for _ in range(warmup):
            tr.step(bm, train=True, sync=sync)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            tr.step(bm, train=True, sync=sync)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        print(json.dumps(dict(config=name + ('' if sync else ' (no per-step .item())'), B=B, L=L, D=D, ms_per_step=round(dt * 1e3, 3),
                              sentences_per_s=round(B / dt, 1))), flush=True)


if __name__ == '__main__':
    which = sys.argv[1] if len(sys.argv) > 1 else 'both'
    if which in ('both', 'diora'):
        run('DIORA whole step (c2 shape)', False)
    if which in ('both', 'cliora'):
        run('CLIORA whole step (c3 shape)', True)
