#!/usr/bin/env python
"""Random-shape sweep of the head kernels (Embed / ImageEncoder projections, reconstruction loss, VG loss) against the oracle's torch
formulas: the bodies of tests/test_gpu_heads.py called with random sizes.   python tools/fuzz_heads.py [n] [seed]"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_gpu_heads as T                                   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(n):
    B, L = rnd.randint(1, 9), rnd.randint(1, 12)
    V, E = rnd.randint(L + 2, 300), rnd.choice([8, 16, 24, 32, 100, 256, 1024])
    D = rnd.choice([16, 33, 48, 50, 64, 100, 200, 400])
    K = rnd.randint(1, min(V - 1, 120))
    Rr, Kf = rnd.randint(1, 64), rnd.choice([16, 48, 100, 512, 2048])
    calls = [('embed', T.test_embed_projection, (B, L, V, E, D)), ('image', T.test_image_encoder_projection, (B, Rr, Kf, D)),
             ('recon', T.test_reconstruction_loss, (B, L, V, E, D, K)), ('vg', T.test_vg_loss, (B, L, Rr))]
    for name, fn, args in calls:
        try:
            fn(*args)
        except Exception as e:                               # noqa: BLE001
            bad += 1
            print('FAIL', name, args, repr(e)[:300], flush=True)
print('%d cases x 4 heads, %d failures' % (n, bad))
