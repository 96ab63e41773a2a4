import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    d = {k: z[k] for k in z.files}
    if 'meta' in d:
        d['meta'] = json.loads(str(d['meta']))
    return d


def params_from_golden(g, prefix='param__', as_torch=True, requires_grad=False):
    import torch
    out = {}
    for k, v in g.items():
        if k.startswith(prefix):
            name = k[len(prefix):].replace('__', '.')
            t = torch.from_numpy(v.copy())
            if requires_grad and t.is_floating_point():
                t.requires_grad_(True)
            out[name] = t if as_torch else v
    return out


@pytest.fixture(scope='session')
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]
    return get
