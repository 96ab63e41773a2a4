#!/usr/bin/env python
"""Random-shape parity sweep of the HIP path against the CPU oracle (outputs and gradients), over the module variants and the schedule
switches: DioraMLP / CLIORA / DioraTreeLSTM, share, normalize, compress, arithmetic mode, wavefront, sentence-resident.
Not a test (the committed tests pin chosen cases): a tool for hunting shape-dependent bugs.   python tools/fuzz_parity.py [n] [seed] [small]
('small': text-only DioraMLP at D <= 64 only -- the shapes of the sentence-resident kernels)"""
import os
import random
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cliora_amd import _lib                                  # noqa: E402
from oracle import diora_ref as R                            # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
KEYS = ('inside_h', 'inside_s', 'outside_h', 'outside_s')
SMALL = len(sys.argv) > 3 and sys.argv[3] == 'small'


def load(m, P, share):
    sd = m.state_dict()
    for k in sd:
        sd[k] = P[k if k in P else 'inside_' + k[len('outside_'):]].detach().clone()
    m.load_state_dict(sd)
    return m.cuda().train()


def rel(a, b):
    a = a.detach().double().cpu().flatten(); b = b.detach().double().cpu().flatten()
    sc = max(1.0, float(b.abs().max())) if b.numel() else 1.0
    d = (a - b).abs()
    return float(d.max()) / sc, float(d.median()) / sc if d.numel() else 0.0


bad = 0
t0 = time.time()
for case in range(n_cases):
    arch = rnd.choice(['mlp', 'mlp', 'cliora', 'treelstm'])
    D = rnd.choice([16, 33, 48, 64, 96, 128, 200, 256, 400, 400])
    if SMALL:
        arch, D = 'mlp', rnd.choice([5, 16, 20, 33, 48, 50, 64])
    L = rnd.randint(1, 22 if D <= 128 else 18)
    B = rnd.randint(1, 5)
    if rnd.random() < 0.35 and D <= 200 and L <= 12:
        B = rnd.choice([8, 9, 12, 16, 19, 24])      # round 5: batches of eight and more take the sentence-affine block order (chart_kernels.hpp: cell_of_block), with and without a remainder
    share = rnd.random() < 0.6
    normalize = 'unit' if rnd.random() < 0.8 else 'none'
    if arch == 'treelstm':
        normalize = 'unit'      # without normalisation the TreeLSTM chart is chaotic in the reference's own arithmetic (fp32 vs fp64 oracle: 5e-3 at L 9, 0.3 at L 16)
    compress = arch != 'treelstm' and rnd.random() < 0.25
    Rr = rnd.randint(1, 40)
    mode = rnd.choice(['f32', 'bf16x3'])
    wf = rnd.choice(['auto', 'off', 'on', 'merged'])
    rd = rnd.choice(['auto', 'off', 'on']) if D <= 64 else 'auto'
    seed = rnd.randint(0, 10 ** 6)
    desc = dict(arch=arch, D=D, L=L, B=B, share=share, normalize=normalize, compress=compress, R=Rr if arch == 'cliora' else 0, mode=mode,
                wavefront=wf, resident=rd, seed=seed)
    _lib.set_mfma_mode(mode); _lib.set_wavefront(wf); _lib.set_resident(rd)
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, L, D, generator=g)
    C = L * (L + 1) // 2
    try:
        if arch == 'treelstm':
            from cliora_amd.treelstm import DioraTreeLSTM
            P = R.init_params_treelstm(D, seed=seed, share=share)
            m = load(DioraTreeLSTM(D, share=share, normalize=normalize), P, share)
            keys = KEYS + ('inside_c', 'outside_c')
            xg = x.clone().cuda().requires_grad_(True)
            m(xg, xg)
            for v in P.values():
                v.requires_grad_(True)
            xc = x.clone().requires_grad_(True)
            ref = R.diora_forward(P, xc, xc, arch='treelstm', share=share, normalize=normalize)
            extra_g, extra_r = [], []
        elif arch == 'cliora':
            from cliora_amd.cliora import DioraMLP
            P = R.init_params(D, share=share, seed=seed, compress=compress)
            m = load(DioraMLP(D, share=share, normalize=normalize, compress=compress), P, share)
            m.lazy_region_scores = False
            mask = torch.bernoulli(torch.full((B, C, Rr), 0.9), generator=g) / 0.9        # dropout(p = 0.1) mask from the case's own generator: a re-run draws the same one
            m.dropout_mask = mask.cuda()
            xw = torch.randn(B, L, D, generator=g); ob = 0.3 * torch.randn(B, Rr, D, generator=g); ow = 0.3 * torch.randn(B, Rr, D, generator=g)
            tg = [t.clone().cuda().requires_grad_(True) for t in (x, xw, ob, ow)]
            m(*tg)
            xg = tg[0]
            for v in P.values():
                v.requires_grad_(True)
            tc = [t.clone().requires_grad_(True) for t in (x, xw, ob, ow)]
            xc = tc[0]
            off = [C - (L - lv) * (L - lv + 1) // 2 for lv in range(L)] + [C]
            calls = {'i': 0}
            orig = R.F.dropout

            def replay(t, p, training):
                i = calls['i']; calls['i'] += 1
                return t * mask[:, off[i]:off[i + 1]]
            R.F.dropout = replay
            try:
                ref = R.diora_forward(P, *tc, share=share, normalize=normalize, training=True)
            finally:
                R.F.dropout = orig
            keys = KEYS + ('all_atten_score', 'vg_atten_score')
            extra_g, extra_r = tg[1:], tc[1:]
        else:
            from cliora_amd.diora import DioraMLP
            P = R.init_params(D, share=share, seed=seed, compress=compress)
            m = load(DioraMLP(D, share=share, normalize=normalize, compress=compress), P, share)
            keys = KEYS
            xg = x.clone().cuda().requires_grad_(True)
            m(xg, xg)
            for v in P.values():
                v.requires_grad_(True)
            xc = x.clone().requires_grad_(True)
            ref = R.diora_forward(P, xc, xc, share=share, normalize=normalize, training=True)
            extra_g, extra_r = [], []
        cot = {k: torch.randn(ref[k].shape, generator=g) for k in keys}
        sum((ref[k] * cot[k]).sum() for k in keys).backward()
        torch.autograd.backward([getattr(m, k) for k in keys], [cot[k].cuda() for k in keys])
        torch.cuda.synchronize()
        worst = []
        tol_out = 1e-4 * (3.0 if normalize == 'none' else 1.0)
        if arch == 'treelstm' and normalize == 'none':
            tol_out = 5e-3                     # ill-conditioned in the reference's own arithmetic: the fp32 oracle is 5e-4 .. 2e-3 from an fp64 run there
        for k in keys:
            mx, _ = rel(getattr(m, k), ref[k])
            if mx > tol_out:
                worst.append(('out ' + k, mx))
        named = dict(m.named_parameters())
        gtol_med = 2e-4 if mode == 'f32' else 1e-3
        for k, p_ in list(P.items()) + [('x_span', xc)] + [('extra%d' % i, t) for i, t in enumerate(extra_r)]:
            gr = p_.grad
            gg = xg.grad if k == 'x_span' else extra_g[int(k[5:])].grad if k.startswith('extra') else named[k].grad
            if gr is None:
                continue
            mx, med = rel(gg, gr)
            if med > gtol_med or mx > 0.1:
                worst.append(('grad ' + k, mx, med))
        # trees of the evaluation forward (scripts/parse.py: eval mode, no grad) against the oracle's CKY over its own scores; a tree that
        # differs must be a tie under the REFERENCE's scores (within 1e-4 of the score scale)
        if L >= 2 and arch != 'cliora':
            m.eval()
            with torch.no_grad():
                xe = x.clone().cuda()
                m(xe, xe)
            trees = m.cky()
            with torch.no_grad():
                if arch == 'treelstm':
                    pe = R.diora_forward({k: v.detach() for k, v in P.items()}, x, x, arch='treelstm', share=share, normalize=normalize, keep_pairs=True)
                else:
                    pe = R.diora_forward({k: v.detach() for k, v in P.items()}, x, x, share=share, normalize=normalize, keep_pairs=True)
            want = R.cky_trees(pe['pair_s_in'], B, L)
            for b in range(B):
                if str(trees[b]) != str(want[b]):
                    sc = max(1.0, max(float(v.abs().max()) for v in pe['pair_s_in'].values()))
                    gap = abs(R.tree_score(pe['pair_s_in'], b, trees[b]) - R.tree_score(pe['pair_s_in'], b, want[b]))
                    if gap > 1e-4 * sc:
                        worst.append(('tree of sentence %d' % b, gap / sc))
        if worst:
            bad += 1
            print('MISMATCH', desc, worst[:4], flush=True)
    except Exception as e:                                   # noqa: BLE001
        bad += 1
        print('ERROR', desc, repr(e)[:300], flush=True)
print('%d cases, %d bad, %.0f s' % (n_cases, bad, time.time() - t0))
