"""Whole training step on the GPU: Embed -> native chart -> losses -> backward -> clip -> Adam, against
the values captured from the reference's Net / Trainer (tests/golden/net_*.npz, trainer.py:243-304, 450-501)."""
import numpy as np
import pytest
import torch

from conftest import grad_check, load_golden

pytestmark = pytest.mark.gpu


def _build(g, vl):
    from cliora_amd import harness as H
    m = g['meta']
    emb = torch.nn.Embedding(m['V'], 32)
    net = H.build_net(m['D'], emb, obj_feats=vl, img_dim=48, k_neg=m['K'], vg_loss=vl, use_contr=vl,
                      vl_margin=0.2, alpha_contr=1.0, alpha_vg=1.0)
    sd = net.state_dict()
    for k in sd:
        sd[k] = torch.from_numpy(g['param__' + k.replace('.', '__')].copy())
    net.load_state_dict(sd)
    return net.cuda()


def _batch(g):
    return dict(sentences=torch.from_numpy(g['sentences']).cuda(), neg_samples=torch.from_numpy(g['neg_samples']).cuda(),
                obj_feats=torch.from_numpy(g['obj_feats']).cuda())


@pytest.mark.parametrize('name,vl', [('net_diora.npz', False), ('net_cliora.npz', True)])
def test_net_losses_grads_and_three_adam_steps(name, vl, mfma_mode):
    from cliora_amd import harness as H
    g = load_golden(name)
    net = _build(g, vl).eval()             # fixtures were captured with dropout off
    bm = _batch(g)
    out = net(bm['sentences'], bm['obj_feats'], bm['neg_samples'])
    ref = g['total_loss']
    assert np.abs(out['total_loss'].detach().cpu().numpy() - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max())
    out['total_loss'].mean(0).sum().backward()
    for k, p in net.named_parameters():
        gk = 'grad__' + k.replace('.', '__')
        if gk in g and p.grad is not None:
            grad_check(p.grad, g[gk], mfma_mode, 2e-4, k)
    # Trainer._step x3 (dropout kept off as in the fixture)
    net.zero_grad()
    tr = H.Trainer(net, lr=g['meta']['lr'])
    net.train = lambda mode=True: torch.nn.Module.train(net, False)
    for s in range(3):
        r = tr.step(bm, train=True)
        assert abs(r['total_loss'] - g['step_losses'][s]) <= 2e-4 * max(1.0, abs(g['step_losses'][s]))
    for k, p in net.named_parameters():
        if not p.requires_grad:
            continue
        want = torch.from_numpy(g['after3__' + k.replace('.', '__')])
        # Adam divides every gradient element by its own running magnitude: an element whose gradient is at the rounding-noise
        # level moves by up to lr per step whatever its size, so three steps of the split-bf16 mode (operands rounded to 16 bits)
        # are held to half an lr (lr = 2e-3 in the fixtures) and the exact-fp32 mode to a quarter of it
        tol = 5e-4 if mfma_mode == 'f32' else 1e-3
        have = p.detach().cpu()
        # ... and an element whose TRUE gradient is zero (img_encoder.fc_vis.bias: the same vector added to every region cancels in
        # the attention softmax; the reference's own value there is 1e-9 of rounding residue) takes a random walk of +-lr per step in
        # the reference as much as here: such elements are only held to the walk's bound from the starting value
        g0 = torch.from_numpy(g['grad__' + k.replace('.', '__')]) if 'grad__' + k.replace('.', '__') in g else None
        noise = (g0.abs() < 1e-7) if g0 is not None else torch.zeros_like(have, dtype=torch.bool)
        start = torch.from_numpy(g['param__' + k.replace('.', '__')])
        assert float((have - start)[noise].abs().max() if noise.any() else 0.0) <= 3.2 * g['meta']['lr'], k
        err = (have - want).abs()[~noise]
        assert float(err.max() if err.numel() else 0.0) <= tol * max(1.0, float(want.abs().max())), k


@pytest.mark.parametrize('B,L,margin', [(64, 20, 1.0), (7, 5, 0.2), (128, 6, 0.5), (2, 3, 1.0)])
def test_fused_contrastive_loss_matches_the_torch_formula(B, L, margin):
    """cliora_contrastive_loss (hinge + span-marginal weighting, trainer.py:103-128, one launch) against the same formula written
    with torch ops (ContrastiveLoss.fused = False), value and the gradient of every input, under a non-unit upstream cotangent."""
    from cliora_amd import harness as H

    class Chart:
        pass
    C = L * (L + 1) // 2
    g = torch.Generator().manual_seed(B * 100 + L)
    smax0 = torch.randn(B, B, C, generator=g)
    ins0, outs0 = 0.3 * torch.randn(B, C, 1, generator=g), 0.3 * torch.randn(B, C, 1, generator=g)
    res = {}
    for fused in (True, False):
        t = [x.clone().cuda().requires_grad_(True) for x in (smax0, ins0, outs0)]
        d = Chart()
        d.inside_s, d.outside_s = t[1], t[2]
        # what the module serves: an object whose .max(-1).values is the (B, B, C) region maximum
        d.all_atten_score = torch.stack([t[0], t[0] - 1.0], -1)
        loss_mod = H.ContrastiveLoss(margin=margin, alpha_contr=0.7)
        loss_mod.fused = fused
        loss = loss_mod(d)
        (loss * 1.7).backward()
        res[fused] = (float(loss), [x.grad.clone() for x in t])
    assert abs(res[True][0] - res[False][0]) <= 1e-5 * max(1.0, abs(res[False][0]))
    for a, b in zip(res[True][1], res[False][1]):
        assert float((a - b).abs().max()) <= 1e-6 * max(1.0, float(b.abs().max())) + 1e-9


def test_device_feed_on_the_gpu():
    """DeviceFeed with pinned buffers, a copy stream and per-batch events: the staged batches are the source's, in order."""
    from cliora_amd.data import DeviceFeed, synthetic_batches
    lengths = [6] * 12 + [9] * 8
    want = list(synthetic_batches(80, lengths, 4, seed=3, k_neg=5))
    feed = DeviceFeed(({**b, 'obj_feats': torch.full((b['batch_size'], 36, 64), float(i))} for i, b in enumerate(synthetic_batches(80, lengths, 4, seed=3, k_neg=5))),
                      'cuda:0', depth=2)
    n = 0
    for i, (a, b) in enumerate(zip(feed, want)):
        assert a['sentences'].is_cuda and torch.equal(a['sentences'].cpu(), b['sentences'])
        assert float(a['obj_feats'].sum()) == float(i) * b['batch_size'] * 36 * 64
        n += 1
    assert n == len(want)


def test_trainer_with_a_reducer_writes_chart_gradients_in_place_and_double_forward_is_summed():
    """(1) harness.Trainer(net, reducer=...) re-points every parameter into the optimizer's flat tensor AFTER the reducer
    mapped their addresses -- the chart backward must still find its slices (reducer.copied counts only the gradients torch autograd
    produced); (2) two chart calls under ONE backward with a live arena must give the sum of the two gradients, not twice the last."""
    from cliora_amd import harness as H
    from cliora_amd import parallel
    g = load_golden('net_diora.npz')
    net = _build(g, False).eval()
    bm = _batch(g)

    class LocalReducer(parallel.FlatGradAllReduce):          # the collective left out: one rank, no process group in the test session
        def _reduce(self):
            pass
    params = [p for p in net.parameters() if p.requires_grad]
    red = LocalReducer(params)
    tr = H.Trainer(net, lr=g['meta']['lr'], reducer=red)
    net.train = lambda mode=True: torch.nn.Module.train(net, False)
    assert tr.optimizer.grads is red
    tr.step(bm, train=True)
    chart = [n for n, p in net.named_parameters() if n.startswith('diora.') and p.requires_grad and p.grad is not None]
    assert chart
    for n, p in net.named_parameters():
        if n in chart:
            assert red.lookup(p.detach())[1].data_ptr() == p.grad.data_ptr(), n       # written where RCCL reduces
    assert red.copied <= len(params) - len(chart)
    # (2) the same batch twice under one backward: gradients = 2 x the single-call gradients
    tr.optimizer.zero_grad()
    out = net(bm['sentences'], bm['obj_feats'], bm['neg_samples'])
    out['total_loss'].mean(0).sum().backward()
    single = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
    tr.optimizer.zero_grad()
    o1 = net(bm['sentences'], bm['obj_feats'], bm['neg_samples'])['total_loss'].mean(0).sum()
    o2 = net(bm['sentences'], bm['obj_feats'], bm['neg_samples'])['total_loss'].mean(0).sum()
    (o1 + o2).backward()
    for n, p in net.named_parameters():
        if n in single:
            sc = max(1.0, float(single[n].abs().max()))
            assert float((p.grad - 2.0 * single[n]).abs().max()) <= 1e-5 * sc, n
    red.close()


def test_word_branch_on_the_caller_lane_is_the_single_stream_step():
    """A vision-language training step with the word branch (Embed's word projection, ImageEncoder.fc_vis, the word-region scorer, their
    backward, the region matrix's half of the region-max backward) on the library's caller lane (harness.Net.overlap_word_branch,
    cliora_device_side_stream) against the same step on one stream: the same kernels on the same data, only the streams differ -- losses,
    every gradient of the first step and the parameters after three steps agree to the bit (sums of two terms commute)."""
    from cliora_amd import harness as H
    res = {}
    for overlap in (True, False):
        torch.manual_seed(21)
        V, E, D, B, L, K, R = 200, 64, 48, 5, 6, 12, 36
        net = H.build_net(D, torch.nn.Embedding(V, E), obj_feats=True, img_dim=32, k_neg=K, vg_loss=True, use_contr=True).cuda()
        for p in net.img_encoder.parameters():
            torch.nn.init.normal_(p, std=0.05)
        net.overlap_word_branch = overlap
        g = torch.Generator().manual_seed(22)
        bm = dict(sentences=torch.randint(0, V, (B, L), generator=g).cuda(), neg_samples=torch.randperm(V, generator=g)[:K].cuda(),
                  obj_feats=torch.randn(B, R, 32, generator=g).cuda())
        C = L * (L + 1) // 2
        net.diora.dropout_mask = (torch.rand(B, C, R, generator=g) > 0.1).float().cuda() / 0.9        # the same recorded mask in both runs
        net.train()
        out = net(bm['sentences'], bm['obj_feats'], bm['neg_samples'])
        assert (net.diora.word_lane is not None) == overlap
        out.total().backward()
        torch.cuda.synchronize()
        grads = {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}
        net.zero_grad()
        tr = H.Trainer(net, lr=2e-3)
        losses = [tr.step(bm, train=True)['total_loss'] for _ in range(3)]
        res[overlap] = (float(out.total().detach()), grads, losses, {k: p.detach().clone() for k, p in net.named_parameters()})
    assert res[True][0] == res[False][0]
    assert set(res[True][1]) == set(res[False][1]) and len(res[True][1]) >= 10
    for k in res[False][1]:
        assert torch.equal(res[True][1][k], res[False][1][k]), k
    assert res[True][2] == res[False][2]
    for k in res[False][3]:
        assert torch.equal(res[True][3][k], res[False][3][k]), k
