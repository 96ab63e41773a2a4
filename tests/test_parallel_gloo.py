"""Data-parallel path on 2 CPU processes over gloo: the flat-gradient all-reduce and the
reference's rank-chunking rule (cliora/data/batch_iterator.py:53-66, trainer.py:572-574)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from cliora_amd.parallel import FlatGradAllReduce, rank_chunk
    from oracle import diora_ref as R
    from oracle import synth
    torch.manual_seed(0)
    D, B, L = 12, 4, 5
    P, x, cot = synth.diora_case(D, B, L, 17)
    params = [torch.nn.Parameter(v.clone()) for v in P.values()]
    extra = torch.nn.Parameter(torch.zeros(3))            # never used: must come back as zeros, not None
    params.append(extra)
    named = dict(zip(P.keys(), params))
    xs = rank_chunk(x, world, rank)                        # this rank's sentences
    cs = {k: rank_chunk(v, world, rank) for k, v in cot.items()}
    out = R.diora_forward(named, xs, xs)
    # loss averaged over the GLOBAL batch so that mean-of-rank-gradients == full-batch gradient * (1/world) * world
    loss = sum((out[k] * cs[k]).sum() for k in cs)
    loss.backward()
    red = FlatGradAllReduce(params)
    red.all_reduce_mean()
    flat = torch.cat([p.grad.reshape(-1) for p in params])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    if rank == 0:
        # every rank holds the same averaged gradient
        assert all(torch.equal(gathered[0], g) for g in gathered)
        # and it equals (full-batch gradient) / world
        full = [torch.nn.Parameter(v.clone()) for v in P.values()]
        fn = dict(zip(P.keys(), full))
        o2 = R.diora_forward(fn, x, x)
        sum((o2[k] * cot[k]).sum() for k in cot).backward()
        ref = torch.cat([p.grad.reshape(-1) for p in full] + [torch.zeros(3)]) / world
        ret['err'] = float((gathered[0] - ref).abs().max() / ref.abs().max())
        ret['extra_zero'] = bool((extra.grad == 0).all())
    # ---- producers that write their gradient straight into the flat buffer (what ChartFunction.backward does through
    # cliora_amd.diora._grad_out): autograd installs the view as .grad and the all-reduce has nothing to copy
    from cliora_amd.diora import _grad_out

    class WritesInPlace(torch.autograd.Function):
        @staticmethod
        def forward(ctx, w):
            ctx.w = w.detach()
            return w.sum().reshape(())

        @staticmethod
        def backward(ctx, g):
            out = _grad_out(ctx.w)                      # the parameter's slice of the live flat buffer
            out.copy_(torch.full_like(ctx.w, float(rank + 1)) * g)
            return out
    w = torch.nn.Parameter(torch.zeros(7, 3))
    red2 = FlatGradAllReduce([w])
    WritesInPlace.apply(w).backward()
    in_place = w.grad.data_ptr() == red2.views[0].data_ptr()
    red2.all_reduce_mean()
    if rank == 0:
        ret['in_place'] = bool(in_place) and red2.copied == 0
        ret['in_place_mean'] = float(w.grad.mean())         # (1 + 2) / 2
    # with a gradient already in place the producer must get a fresh tensor (autograd accumulates into .grad)
    WritesInPlace.apply(w).backward()
    if rank == 0:
        ret['accumulated'] = float(w.grad.mean())           # 1.5 + 1
    red2.close()
    red.close()
    dist.barrier()
    dist.destroy_process_group()


def _cliora_step_grads(R, P, heads, sentences, neg, obj):
    """Gradients of one rank's CLIORA training loss (Embed + ImageEncoder + chart + the three losses: the parameter set of
    `bench.py --workload c3`, BASELINE configs[3]) from the CPU oracle.  Returns {name: grad} over chart and head parameters."""
    Pq = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    Hq = {k: v.clone().requires_grad_(True) for k, v in heads.items()}
    xs, xw = R.embed_forward(Hq['emb'], Hq['mat'], Hq['mat1'], sentences)
    os_, ow = R.image_encoder_forward(Hq['fc.weight'], Hq['fc.bias'], Hq['fc_vis.weight'], Hq['fc_vis.bias'], obj)
    ref = R.diora_forward(Pq, xs, xw, os_, ow, training=False)
    loss = (R.reconstruction_loss(Hq['emb'], Hq['rmat'], sentences, neg, ref['outside_h']) + R.vg_loss(ref['vg_atten_score'], 1.0)
            + R.contrastive_loss(ref['inside_s'], ref['outside_s'], ref['all_atten_score'], 0.2, 1.0))
    loss.backward()
    out = {k: v.grad for k, v in Pq.items() if v.grad is not None}
    out.update({'head.' + k: v.grad for k, v in Hq.items() if v.grad is not None})
    return out


def _worker_cliora(rank, world, port, ret):
    """One flat all-reduce over chart + head + ImageEncoder gradients (the c3 / c4 parameter set): every rank trains on its own chunk
    of the global batch -- the in-batch negatives of the VG and contrastive losses stay local to a rank (trainer.py:101-121, 142-169),
    so the reduced gradient is the MEAN OF THE RANKS' gradients, which rank 0 recomputes serially."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from cliora_amd.parallel import FlatGradAllReduce, rank_chunk
    from oracle import diora_ref as R
    D, B, L, Rg, V, E, K, F = 12, 4, 5, 3, 30, 16, 5, 8
    g = torch.Generator().manual_seed(5)
    P = {k: v.detach() for k, v in R.init_params(D, share=True, seed=3).items()}
    heads = {'emb': torch.randn(V, E, generator=g), 'mat': torch.randn(D, E, generator=g), 'mat1': torch.randn(D, E, generator=g),
             'fc.weight': 0.1 * torch.randn(D, F, generator=g), 'fc.bias': 0.1 * torch.randn(D, generator=g),
             'fc_vis.weight': 0.1 * torch.randn(D, F, generator=g), 'fc_vis.bias': 0.1 * torch.randn(D, generator=g),
             'rmat': torch.randn(D, E, generator=g)}
    sentences = torch.randint(0, V, (B, L), generator=g)
    neg = torch.randperm(V, generator=g)[:K]
    obj = torch.relu(torch.randn(B, Rg, F, generator=g))
    mine = _cliora_step_grads(R, P, heads, rank_chunk(sentences, world, rank), neg, rank_chunk(obj, world, rank))
    names = sorted(mine)
    params = []
    for n in names:
        p = torch.nn.Parameter(torch.zeros_like(mine[n]))
        p.grad = mine[n].clone()
        params.append(p)
    red = FlatGradAllReduce(params)
    red.all_reduce_mean()
    if rank == 0:
        want = None
        for r in range(world):
            gr = _cliora_step_grads(R, P, heads, rank_chunk(sentences, world, r), neg, rank_chunk(obj, world, r))
            want = gr if want is None else {n: want[n] + gr[n] for n in names}
        err = max(float((p.grad - want[n] / world).abs().max() / max(1e-12, float(want[n].abs().max()))) for n, p in zip(names, params))
        ret['cliora_err'] = err
        ret['cliora_names'] = len(names)
        ret['cliora_floats'] = red.flat.numel()
    red.close()
    dist.barrier()
    dist.destroy_process_group()


def test_flat_grad_allreduce_world2_cliora_parameter_set():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_cliora, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert ret['cliora_err'] < 1e-5, ret['cliora_err']
    assert ret['cliora_names'] >= 14            # 6 chart tensors (shared weights) + Embed (3) + ImageEncoder (4) + the reconstruction head


def test_flat_grad_allreduce_world2():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert ret['err'] < 1e-5, ret['err']
    assert ret['extra_zero']
    assert ret['in_place'] and abs(ret['in_place_mean'] - 1.5) < 1e-6
    assert abs(ret['accumulated'] - 2.5) < 1e-6


def test_rank_chunk_matches_torch_chunk():
    from cliora_amd.parallel import rank_chunk
    t = torch.arange(10).view(5, 2)
    assert torch.equal(rank_chunk(t, 2, 0), t[:3]) and torch.equal(rank_chunk(t, 2, 1), t[3:])
