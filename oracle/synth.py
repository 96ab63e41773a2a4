"""Seeded synthetic inputs shared by the golden generator, the tests and bench.py.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  The draw order below is the
one tests/golden/make_golden.py used when it ran the reference, so a case can be
re-created from its seed on any box with the same torch build.
"""
import torch

from . import chart_layout as CL
from .diora_ref import init_params


def diora_case(D, B, L, seed, share=True, compress=False):
    """-> (params, x_span, cotangents) exactly as make_golden.diora_case drew them."""
    P = init_params(D, share=share, seed=seed, compress=compress)
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(B, L, D, generator=g)
    C = CL.n_cells(L)
    cot = {}
    for k, w in (('inside_h', D), ('inside_s', 1), ('outside_h', D), ('outside_s', 1)):
        cot[k] = torch.randn((B, C, w), generator=g)
    return P, x, cot
