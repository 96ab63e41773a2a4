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


# The flat buffer doubles as the place the chart backward WRITES its parameter gradients: cliora_amd.diora.ChartFunction.backward
# asks grad_buffer_for() for the output tensor of every parameter and gets the view of the live FlatGradAllReduce (the C ABI
# writes gradients, it does not accumulate), autograd installs that view as .grad, and all_reduce_mean then has nothing to copy.
_ARENAS = []


def grad_buffer_for(param_tensor):
    """View of a live flat gradient buffer for the parameter whose storage `param_tensor` shares, or None."""
    for a in _ARENAS:
        pv = a.by_ptr.get(param_tensor.data_ptr())
        # only while the parameter holds no gradient: autograd then installs the view as .grad; with a gradient in place it would
        # ACCUMULATE the view into itself (zero_grad(set_to_none=False), gradient accumulation): those cases take a fresh tensor
        if pv is not None and pv[1].shape == param_tensor.shape and pv[0].grad is None:
            return pv[1].detach()        # a fresh alias: autograd only adopts a gradient tensor nobody else references
    return None


class FlatGradAllReduce(object):
    def __init__(self, params, group=None, adopt_chart_grads=True):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        n = sum(p.numel() for p in self.params)
        p0 = self.params[0]
        self.flat = torch.zeros(n, device=p0.device, dtype=torch.float32)
        self.views, o = [], 0
        for p in self.params:
            self.views.append(self.flat[o:o + p.numel()].view_as(p))
            o += p.numel()
        self.by_ptr = {p.data_ptr(): (p, v) for p, v in zip(self.params, self.views)}
        self.copied = 0                      # gradients copied in by the last all_reduce_mean (0 when every producer wrote in place)
        if adopt_chart_grads:
            _ARENAS.append(self)

    def close(self):
        if self in _ARENAS:
            _ARENAS.remove(self)

    def all_reduce_mean(self):
        """Average .grad over the ranks; parameters without a grad contribute zeros (the
        find_unused_parameters=True behaviour of the reference's DDP wrapper)."""
        self.copied = 0
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():          # produced elsewhere (torch autograd): bring it in
                v.copy_(p.grad)
                self.copied += 1
        world = dist.get_world_size(self.group)
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        self.flat.div_(world)
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                p.grad = v.clone()
            elif p.grad.data_ptr() != v.data_ptr():
                p.grad.copy_(v)
