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
