"""SURVEY section 8 row f4 on the GPU: a checkpoint WRITTEN BY THE REFERENCE (Trainer.save_model, trainer.py:383-397; a
DistributedDataParallel-style file with `module.` prefixes and a foreign key as well) is loaded into the native Net by
cliora_amd.interchange.load_model (the reference's lenient loader, trainer.py:399-435), moved to the GPU and run on a fixture batch:
charts, loss and trees must be the ones the REFERENCE's own Net produced after ITS load_model of the same file
(tests/golden/interchange_run.npz, written by tests/golden/make_golden.py::interchange_run_case)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN as GOLDEN_DIR, load_golden

pytestmark = pytest.mark.gpu
CHARTS = ('inside_h', 'inside_s', 'outside_h', 'outside_s')


def _receiving_net(g):
    """A native Net with the receiving net's own initialisation (what survives a load for the keys the file does not carry)."""
    from cliora_amd import harness as H
    m = g['meta']
    net = H.build_net(m['D'], torch.nn.Embedding(m['V'], 16), obj_feats=False, img_dim=20, k_neg=m['K'])
    first = {k[len('dst0__'):].replace('__', '.'): torch.from_numpy(v.copy()) for k, v in g.items() if k.startswith('dst0__')}
    assert set(first) == set(net.state_dict()), 'parameter names differ from the reference Net'
    net.load_state_dict(first)
    return net


@pytest.mark.parametrize('resident', ['auto', 'off'])          # D = 24: the sentence-resident kernels by default; 'off': the level kernels
@pytest.mark.parametrize('tag,fname,origin_emb', [('noemb', 'ref_model_noemb.pt', False), ('ddp', 'ref_model_ddp.pt', True)])
def test_reference_checkpoint_runs_on_the_native_chart(tag, fname, origin_emb, resident, mfma_mode):
    from cliora_amd import _lib
    from cliora_amd import interchange as X
    g = load_golden('interchange_run.npz')
    net = _receiving_net(g)
    taken, kept = X.load_model(origin_emb, net, os.path.join(GOLDEN_DIR, fname))
    assert any('diora.' in k for k in taken) and all(('embeddings' in k) for k in kept)
    net = net.cuda().eval()
    sent, neg = torch.from_numpy(g['sentences']).cuda(), torch.from_numpy(g['neg_samples']).cuda()
    prev = _lib.set_resident(resident)
    try:
        with torch.no_grad():
            out = net(sent, None, neg)
        d = net.diora
        for k in CHARTS:
            want = g['%s__%s' % (tag, k)]
            err = np.abs(getattr(d, k).cpu().numpy() - want).max()
            assert err <= 1e-4 * max(1.0, np.abs(want).max()), (tag, k, err)
        want = g['%s__total_loss' % tag]
        assert np.abs(out['total_loss'].cpu().numpy() - want).max() <= 1e-4 * max(1.0, np.abs(want).max())
        assert [str(t) for t in d.cky()] == json.loads(str(g['%s__trees' % tag]))
    finally:
        _lib.set_resident(prev)


def test_checkpoint_round_trip_through_the_gpu(tmp_path):
    """save_model of the CUDA net writes the reference's layout (keys, order, CPU-loadable values); a fresh net that loads it computes the
    same charts to the bit."""
    from cliora_amd import interchange as X
    g = load_golden('interchange_run.npz')
    net = _receiving_net(g)
    X.load_model(True, net, os.path.join(GOLDEN_DIR, 'ref_model_emb.pt'))
    net = net.cuda().eval()
    path = str(tmp_path / 'native.pt')
    X.save_model(net, True, path)
    ref = torch.load(os.path.join(GOLDEN_DIR, 'ref_model_emb.pt'), map_location='cpu')['state_dict']
    mine = torch.load(path, map_location='cpu')['state_dict']
    assert list(mine) == list(ref)
    for k in ref:
        assert torch.equal(mine[k].cpu(), ref[k]), k
    other = _receiving_net(g)
    X.load_model(True, other, path)
    other = other.cuda().eval()
    sent, neg = torch.from_numpy(g['sentences']).cuda(), torch.from_numpy(g['neg_samples']).cuda()
    with torch.no_grad():
        net(sent, None, neg)
        other(sent, None, neg)
    for k in CHARTS:
        assert torch.equal(getattr(net.diora, k), getattr(other.diora, k)), k
