"""Data parallelism for the chart path: one flat gradient buffer, one all-reduce per step.

The reference wraps the whole Net in DistributedDataParallel over NCCL with
find_unused_parameters=True (cliora/net/trainer.py:528-532, 572-574) and shards every batch
by ``torch.chunk(x, ngpus, 0)[rank]`` (cliora/data/batch_iterator.py:53-66, 134-136).
Sentences are independent through the recursion, so the only exchange is the gradient
average.  On MI355X that is a single RCCL all-reduce of one contiguous fp32 buffer over
xGMI (the d=400 DIORA parameters are 3.2 MB: latency-bound, so one call, not buckets).
"""
import weakref

import torch
import torch.distributed as dist


def rank_chunk(t, world, rank):
    """The reference's sharding rule for one batch field (batch_iterator.py:53-66)."""
    return torch.chunk(t, world, 0)[rank]


# The flat buffer doubles as the place the chart backward WRITES its parameter gradients: cliora_amd.diora.ChartFunction.backward
# asks grad_buffer_for() for the output tensor of every parameter and gets the view of the live FlatGradAllReduce (the C ABI
# writes gradients, it does not accumulate), autograd installs that view as .grad, and all_reduce_mean then has nothing to copy.
_ARENAS = []          # weak references: an arena lives as long as its owner (reducer / optimizer) does


def _live_arenas():
    out = []
    for r in list(_ARENAS):
        a = r()
        if a is None:
            _ARENAS.remove(r)
        else:
            out.append(a)
    return out


def grad_buffer_for(param_tensor):
    """View of a live flat gradient buffer for the parameter whose storage `param_tensor` shares, or None.

    A slice is handed out at most ONCE per backward pass (autograd graph task): two chart calls under one backward (two forwards
    with summed losses) would otherwise both write -- not accumulate -- into the same memory and autograd would then add two
    aliases of one buffer: twice the last gradient, silently.  The second request of a pass gets None (a fresh tensor), which
    autograd accumulates as usual."""
    task = torch._C._current_graph_task_id()
    for a in _live_arenas():
        pv = a.lookup(param_tensor)
        # only while the parameter holds no gradient: autograd then installs the view as .grad; with a gradient in place it would
        # ACCUMULATE the view into itself (zero_grad(set_to_none=False), gradient accumulation): those cases take a fresh tensor
        if pv is not None and pv[1].shape == param_tensor.shape and pv[0].grad is None:
            key = id(pv[0])
            if task >= 0 and a.handed.get(key) == task:
                return None
            a.handed[key] = task
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
        self.by_ptr, self._ptr_sig = {}, None
        self.handed = {}                     # id(param) -> autograd graph task its slice was last handed out in (grad_buffer_for)
        self.copied = 0                      # gradients copied in by the last all_reduce_mean (0 when every producer wrote in place)
        if adopt_chart_grads:
            _ARENAS.append(weakref.ref(self))

    def lookup(self, param_tensor):
        """(parameter, its slice) for the parameter whose storage `param_tensor` shares.  The address map is rebuilt when the
        parameters have moved (heads.FusedClipAdam re-points every p.data into one flat tensor after a reducer was built)."""
        sig = (self.params[0].data_ptr(), self.params[-1].data_ptr())
        if sig != self._ptr_sig:
            self.by_ptr = {p.data_ptr(): (p, v) for p, v in zip(self.params, self.views)}
            self._ptr_sig = sig
        return self.by_ptr.get(param_tensor.data_ptr())

    def close(self):
        for r in list(_ARENAS):
            if r() is self or r() is None:
                _ARENAS.remove(r)

    def _reduce(self):
        """The one collective of a step: sum over the ranks (RCCL all-reduce over xGMI on ROCm), then the mean."""
        world = dist.get_world_size(self.group)
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        self.flat.div_(world)

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
        self._reduce()
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                p.grad = v.clone()
            elif p.grad.data_ptr() != v.data_ptr():
                p.grad.copy_(v)
