"""Data parallelism for the chart path: one flat gradient buffer, one all-reduce per step.

The reference wraps the whole Net in DistributedDataParallel over NCCL with
find_unused_parameters=True (cliora/net/trainer.py:528-532, 572-574) and shards every batch
by ``torch.chunk(x, ngpus, 0)[rank]`` (cliora/data/batch_iterator.py:53-66, 134-136).
Sentences are independent through the recursion, so the only exchange is the gradient
average.  On MI355X that is a single RCCL all-reduce of one contiguous fp32 buffer over
xGMI (the d=400 DIORA parameters are 3.2 MB: latency-bound, so one call, not buckets).
"""
import torch
import torch.distributed as dist


def rank_chunk(t, world, rank):
    """The reference's sharding rule for one batch field (batch_iterator.py:53-66)."""
    return torch.chunk(t, world, 0)[rank]


class FlatGradAllReduce(object):
    def __init__(self, params, group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        n = sum(p.numel() for p in self.params)
        p0 = self.params[0]
        self.flat = torch.zeros(n, device=p0.device, dtype=torch.float32)
        self.views, o = [], 0
        for p in self.params:
            self.views.append(self.flat[o:o + p.numel()].view_as(p))
            o += p.numel()

    def all_reduce_mean(self):
        """Average .grad over the ranks; parameters without a grad contribute zeros (the
        find_unused_parameters=True behaviour of the reference's DDP wrapper)."""
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                v.zero_()
            else:
                v.copy_(p.grad)
        world = dist.get_world_size(self.group)
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        self.flat.div_(world)
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                p.grad = v.clone()
            else:
                p.grad.copy_(v)
