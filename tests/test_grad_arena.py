"""The flat gradient arena (cliora_amd/parallel.py): producers that WRITE their parameter gradients into slices of one flat buffer
(the chart backward through grad_buffer_for) must never be handed the same slice twice inside one backward pass."""
import gc

import torch

from cliora_amd import parallel


class WritesItsGrad(torch.autograd.Function):
    """Stands in for ChartFunction.backward on the CPU: asks the arena for the output tensor of the weight gradient and writes --
    not accumulates -- into it, as the C ABI does."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.x, ctx.w = x, w.detach()
        return (x * w).sum()

    @staticmethod
    def backward(ctx, g):
        out = parallel.grad_buffer_for(ctx.w)
        if out is None:
            out = torch.empty_like(ctx.w)
        out.copy_(g * ctx.x)
        return None, out


def test_two_producers_under_one_backward_do_not_share_a_slice():
    w = torch.nn.Parameter(torch.tensor([1.0, 2.0, 3.0]))
    arena = parallel.FlatGradAllReduce([w])
    x1, x2 = torch.tensor([1.0, 10.0, 100.0]), torch.tensor([5.0, 6.0, 7.0])
    (WritesItsGrad.apply(x1, w) + WritesItsGrad.apply(x2, w)).backward()
    assert torch.equal(w.grad, x1 + x2)                      # before the fix: 2 * x1 (two aliases of one slice, the last write twice)
    # the next pass (gradient cleared) gets the slice again, in place
    w.grad = None
    WritesItsGrad.apply(x1, w).backward()
    assert torch.equal(w.grad, x1) and w.grad.data_ptr() == arena.views[0].data_ptr()
    # accumulation across passes (gradient kept): fresh tensor, autograd adds it in
    WritesItsGrad.apply(x2, w).backward()
    assert torch.equal(w.grad, x1 + x2)
    arena.close()


def test_lookup_follows_repointed_parameters_and_arenas_do_not_leak():
    w = torch.nn.Parameter(torch.tensor([1.0, 2.0]))
    v = torch.nn.Parameter(torch.tensor([3.0]))
    arena = parallel.FlatGradAllReduce([w, v])
    flat = torch.empty(3)
    flat[:2].copy_(w.data); flat[2:].copy_(v.data)
    w.data, v.data = flat[:2], flat[2:]                      # what heads.FusedClipAdam does after a reducer was built
    WritesItsGrad.apply(torch.tensor([4.0, 5.0]), w).backward()
    assert w.grad.data_ptr() == arena.views[0].data_ptr()    # still written in place
    n = len(parallel._ARENAS)
    del arena
    gc.collect()
    assert len(parallel._live_arenas()) == n - 1             # held weakly: gone with its owner
