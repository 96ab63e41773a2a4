#!/usr/bin/env python
"""Loss curves of the whole training step (Embed -> chart -> reconstruction loss -> clip -> Adam, cliora_amd.harness)
on a fixed synthetic corpus under both arithmetic modes of the compose GEMMs; prints the per-step losses and their gap.
  python tools/train_curve.py [steps]          (on the MI355X box)"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cliora_amd import _lib, harness as H                 # noqa: E402
from cliora_amd.data import synthetic_batches              # noqa: E402


def run(mode, steps, D=400, V=2000, B=64, L=20, K=100):
    _lib.set_mfma_mode(mode)
    torch.manual_seed(0)
    emb = torch.nn.Embedding(V, 64)
    net = H.build_net(D, emb, obj_feats=False, k_neg=K).cuda()
    tr = H.Trainer(net, lr=2e-3)
    batches = list(synthetic_batches(V, [L] * (B * 8), B, seed=5, k_neg=K, device='cuda'))
    losses = []
    for s in range(steps):
        bm = batches[s % len(batches)]
        losses.append(tr.step(dict(sentences=bm['sentences'], neg_samples=bm['neg_samples']), train=True)['total_loss'])
    return losses


if __name__ == '__main__':
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    a = run('f32', steps)
    b = run('bf16x3', steps)
    gap = [abs(x - y) / max(1.0, abs(x)) for x, y in zip(a, b)]
    print(json.dumps(dict(steps=steps, f32_first=a[:3], f32_last=a[-3:], split_last=b[-3:],
                          max_rel_gap=max(gap), rel_gap_at=[gap[i] for i in (0, steps // 4, steps // 2, steps - 1)])))
