"""The projection-backward GEMMs of DioraMLP in their two arithmetic forms (csrc/gemm_kernels.hpp): exact fp32 products on 16 x 16 tiles
(rows_gemm_ksplit, CLIORA_BWD_GEMM3=0) and split-bf16 products on RT x CT tiles (rows_gemm_ksplit3x: the default 32 x 48 and other shapes,
ragged last column blocks, K not a multiple of 32).  The switch is read once per process: tools/gemm3_probe.py runs each setting as a
child and saves every gradient; the forward outputs must be equal to the bit (the switch touches the backward only), the gradients
within the bound the split-bf16 compose GEMMs are held to against the oracle (conftest.grad_check)."""
import os
import subprocess
import sys

import numpy as np
import pytest
pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 2e-5        # of the tensor's largest magnitude; measured worst over the four cases and five shapes: 5.9e-6 (run with -s for the table)


def _probe(tmp_path, tag, env, D, B, L, share):
    out = str(tmp_path / ('g_%s.npz' % tag))
    e = dict(os.environ, **env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'gemm3_probe.py'), str(D), str(B), str(L), str(int(share)), out], cwd=ROOT, env=e,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return dict(np.load(out))


@pytest.mark.parametrize('D,B,L,share', [(400, 64, 20, True), (400, 5, 9, False), (208, 7, 6, True), (144, 16, 12, True)])
def test_split_bf16_tiles_against_the_fp32_kernel(tmp_path, D, B, L, share):
    base = _probe(tmp_path, 'f32', {'CLIORA_BWD_GEMM3': '0'}, D, B, L, share)
    worst = {}
    for shape in ('23', '22', '32', '25', '11'):
        got = _probe(tmp_path, 's' + shape, {'CLIORA_BWD_GEMM3': shape}, D, B, L, share)
        assert set(got) == set(base)
        for k in base:
            if k.startswith('out_'):
                assert np.array_equal(got[k], base[k]), (shape, k)          # the forward does not see the switch
            else:
                d = np.abs(got[k].astype(np.float64) - base[k].astype(np.float64))
                scale = max(1.0, float(np.abs(base[k]).max()))
                worst[k] = max(worst.get(k, 0.0), float(d.max()) / scale)
                # the two forms share every ReLU decision (same forward, same compose backward): no kink moves, only product rounding
                assert float(d.max()) <= TOL * scale, '%s %s: max err %.3e scale %.3e' % (shape, k, float(d.max()), scale)
    print('worst relative differences', {k: '%.2e' % v for k, v in worst.items()})
