"""Host logic of the training harness on CPU tensors (the chart itself is GPU-only: a stub stands in for it): the loss dictionary of
`Net.forward` -- named parts, the reference's `total_loss` row built on demand (trainer.py:300-303), `total()` = the scalar the reference
roots its backward at (trainer.py:487) -- and `Trainer.step` on the torch formulas (clip 5.0 + Adam, trainer.py:450-455)."""
import torch

from cliora_amd import harness as H


class StubChart(torch.nn.Module):
    def __init__(self, D):
        super().__init__()
        self.w = torch.nn.Parameter(torch.randn(D))

    def forward(self, x_span, x_word, obj_span=None, obj_word=None):
        B, L, D = x_span.shape
        self.outside_h = (x_span.mean(1, keepdim=True) * self.w).expand(B, L * (L + 1) // 2, D)


def test_loss_dict_and_cpu_trainer_step():
    torch.manual_seed(0)
    net = H.build_net(8, torch.nn.Embedding(30, 16), k_neg=4)
    net.diora = StubChart(8)
    bm = dict(sentences=torch.randint(0, 30, (3, 5)), neg_samples=torch.arange(4))
    out = net(bm['sentences'], None, bm['neg_samples'])
    assert isinstance(out, H.LossDict) and list(out) == ['reconstruct_softmax_loss']
    total = out.total()
    row = out['total_loss']                                   # built now, (1, n) like the reference's
    assert row.shape == (1, 1) and 'total_loss' in out
    assert torch.equal(row.mean(dim=0).sum(), total)
    tr = H.Trainer(net, lr=1e-2)
    assert not tr.fused                                       # CPU parameters: torch.optim.Adam + clip_grad_norm_
    before = {k: p.detach().clone() for k, p in net.named_parameters()}
    losses = [tr.step(bm)['total_loss'] for _ in range(4)]
    assert losses[-1] < losses[0]
    moved = [k for k, p in net.named_parameters() if not torch.equal(p.detach(), before[k])]
    assert 'diora.w' in moved and 'embed.mat' in moved and 'embed.mat1' not in moved      # the word projection has no gradient in a text-only net
