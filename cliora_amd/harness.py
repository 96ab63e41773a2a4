"""Minimal training harness around the native chart modules: the callers either side of the path.

The reference's CLI cannot run on the GPU box (torchvision / h5py / datasets absent), so this
module restates what sits around ``self.diora(...)`` in cliora/net/trainer.py: ``Embed`` (:204-224),
``ImageEncoder`` (net/utils.py:37-55), the three losses (:25-171), ``Net.forward`` (:272-304) and
the update rule of ``Trainer._step`` / ``gradient_update`` (:450-455, 483-501: backward,
clip_grad_norm_ 5.0, Adam).  On the GPU every one of them runs on this library's kernels
(cliora_amd/heads.py -> csrc/api_heads.hip: gather + fp32-MFMA projections, the fused
reconstruction / VG / contrastive heads, one clip + Adam launch sequence over a flat buffer);
the torch formulas below them are what runs on CPU tensors (the CPU tests) and what the kernels
are checked against.  Module and parameter names follow the reference so that its checkpoints
(``trainer.py:383-398``) load by key.  Pinned by tests/golden/net_*.npz.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import heads
from .diora import DioraMLP as TextDiora
from .cliora import DioraMLP as VLDiora

NATIVE_HEADS = True        # False: the torch formulas everywhere (A/B and tests)


def _native(*tensors):
    return NATIVE_HEADS and all(t is None or t.is_cuda for t in tensors)


def _normal_(module):
    for p in module.parameters():
        if p.requires_grad:
            p.data.normal_()


class Embed(nn.Module):
    """Two bias-free projections of the word embedding: span input and word input."""

    def __init__(self, embeddings, input_size, size):
        super().__init__()
        self.input_size, self.size, self.embeddings = input_size, size, embeddings
        self.mat = nn.Parameter(torch.empty(size, input_size))
        self.mat1 = nn.Parameter(torch.empty(size, input_size))
        _normal_(self)

    def forward(self, tokens, word_lane=None):
        """word_lane: a stream for the WORD projection (the second output), whose only reader is then the word-region scorer on the same
        stream (cliora_amd.cliora.DioraMLP.word_lane); None: both on the current stream."""
        B, L = tokens.shape
        w = self.embeddings.weight
        if _native(w, tokens) and w.shape[1] % 16 == 0:
            idx = tokens.reshape(-1)
            span = heads.proj(w, idx, self.mat).view(B, L, -1)
            if word_lane is None:
                return span, heads.proj(w, idx, self.mat1).view(B, L, -1)
            with torch.cuda.stream(word_lane):
                return span, heads.proj(w, idx, self.mat1).view(B, L, -1)
        e = self.embeddings(tokens.reshape(-1))
        return (e @ self.mat.t()).view(B, L, -1), (e @ self.mat1.t()).view(B, L, -1)


class ImageEncoder(nn.Module):
    """Two linear maps of the region features; zero-initialised like the reference (utils.py:45-50)."""

    def __init__(self, input_size, size):
        super().__init__()
        self.fc, self.fc_vis = nn.Linear(input_size, size), nn.Linear(input_size, size)
        for p in self.parameters():
            p.data.zero_()

    def forward(self, obj_feats, word_lane=None):
        x = obj_feats.float()
        if _native(x) and x.shape[-1] % 16 == 0:
            lead = x.shape[:-1]
            span = heads.proj(x, None, self.fc.weight, self.fc.bias).view(*lead, -1)
            if word_lane is None:
                return span, heads.proj(x, None, self.fc_vis.weight, self.fc_vis.bias).view(*lead, -1)
            with torch.cuda.stream(word_lane):          # see Embed.forward
                return span, heads.proj(x, None, self.fc_vis.weight, self.fc_vis.bias).view(*lead, -1)
        return self.fc(x), self.fc_vis(x)


class ReconstructionSoftmaxLoss(nn.Module):
    name = 'reconstruct_softmax_loss'

    def __init__(self, embeddings, input_size, size, k_neg=3):
        super().__init__()
        self.k_neg, self.embeddings = k_neg, embeddings
        self.mat = nn.Parameter(torch.empty(size, input_size))
        _normal_(self)

    def forward(self, tokens, neg_samples, diora):
        B, L = tokens.shape
        w = self.embeddings.weight
        if _native(w, tokens, neg_samples, diora.outside_h) and w.shape[1] % 16 == 0:
            return heads.recon_loss(w, self.mat, diora.outside_h, tokens, neg_samples)
        pos = self.embeddings(tokens) @ self.mat.t()                     # B,L,D
        neg = self.embeddings(neg_samples) @ self.mat.t()                # K,D
        cell = diora.outside_h[:, :L]                                    # B,L,D  leaf outside vectors
        logits = torch.cat([(pos * cell).sum(-1, keepdim=True), cell @ neg.t()], -1).view(B * L, -1)
        return F.cross_entropy(logits, torch.zeros(B * L, dtype=torch.long, device=logits.device))


class FusedContrastive(torch.autograd.Function):
    """cliora_contrastive_loss: the hinge + span-marginal weighting of trainer.py:103-128 on the region maxima (B, B, C), value and
    gradient in one launch (the backward scales the stored gradients by the incoming cotangent)."""

    @staticmethod
    def forward(ctx, smax, ins, outs, margin, alpha):
        from . import _lib
        import ctypes as C
        B, _, Cc = smax.shape
        smax, ins, outs = smax.contiguous().float(), ins.contiguous().float(), outs.contiguous().float()
        dev = smax.device
        loss = torch.empty(1, device=dev)
        d_smax, d_ins, d_outs = torch.empty_like(smax), torch.empty_like(ins), torch.empty_like(outs)
        nbytes = _lib.lib().cliora_contrastive_workspace_bytes(B, Cc)
        ws = torch.empty(nbytes, device=dev, dtype=torch.uint8)
        p = lambda t: C.c_void_p(t.data_ptr())
        with torch.cuda.device(dev):
            rc = _lib.lib().cliora_contrastive_loss(B, Cc, p(smax), p(ins), p(outs), float(margin), float(alpha), p(loss), p(d_smax), p(d_ins),
                                                   p(d_outs), p(ws), nbytes, C.c_void_p(torch.cuda.current_stream().cuda_stream))
        _lib.check(rc, 'cliora_contrastive_loss')
        ctx.save_for_backward(d_smax, d_ins, d_outs)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        d_smax, d_ins, d_outs = ctx.saved_tensors
        return d_smax * g, d_ins * g, d_outs * g, None, None


class ContrastiveLoss(nn.Module):
    name = 'contrastive_loss'
    fused = True            # native kernel when the scores are on the GPU and B <= 128 (False: the torch ops below)

    def __init__(self, margin=1.0, alpha_contr=0.01):
        super().__init__()
        self.margin, self.alpha, self.floor = margin, alpha_contr, 1e-8

    def forward(self, diora):
        ins, outs = diora.inside_s.squeeze(-1), diora.outside_s.squeeze(-1)
        B, C = ins.shape
        smax = diora.all_atten_score.max(-1).values                      # B(text),B(image),C
        if self.fused and smax.is_cuda and B <= 128:
            return FusedContrastive.apply(smax, ins, outs, self.margin, self.alpha)
        sc = smax.permute(2, 0, 1)                                       # C,B(text),B(image)
        pos = torch.diagonal(sc, 0, 1, 2).unsqueeze(-1)                  # C,B,1
        off_diag = ~torch.eye(B, dtype=torch.bool, device=sc.device)
        txt = (self.margin + sc - pos).clamp(min=self.floor) * off_diag
        img = (self.margin + sc - pos.transpose(1, 2)).clamp(min=self.floor) * off_diag
        per_span = (txt.mean(2) + img.mean(1)).t()                       # B,C
        marginal = torch.exp(ins + outs - ins[:, -1:])
        return (marginal * per_span)[:, :C // 2].sum(-1).mean() * self.alpha


class VGLoss(nn.Module):
    name = 'vg_loss'

    def __init__(self, alpha_vg=0.1):
        super().__init__()
        self.alpha = alpha_vg

    def forward(self, vg_atten_score):
        B, _, L, _ = vg_atten_score.shape
        if _native(vg_atten_score):
            return heads.vg_loss(vg_atten_score, self.alpha)
        logits = vg_atten_score.max(-1).values.sum(-1) / L
        return self.alpha * F.cross_entropy(logits, torch.arange(B, device=logits.device))


class LossDict(dict):
    """What Net.forward returns: the named losses, plus -- built only when somebody asks for it -- the reference's `total_loss`, the (1, n)
    row of the losses (trainer.py:300-303).  `total()` is the scalar the reference roots its backward at,
    `total_loss.mean(dim=0).sum()` (trainer.py:487), as the plain sum of the parts: the same value and the same gradients without the
    cat / mean / sum launches and their four backward launches per step."""
    parts = ()

    def __missing__(self, key):
        if key != 'total_loss':
            raise KeyError(key)
        v = torch.cat([p.view(1, 1) for p in self.parts], 1)
        self[key] = v
        return v

    def total(self):
        t = self.parts[0]
        for p in self.parts[1:]:
            t = t + p
        return t


class Net(nn.Module):
    def __init__(self, embed, image_encoder, diora, obj_feats, loss_funcs=()):
        super().__init__()
        self.obj_feats = obj_feats
        self.overlap_word_branch = True      # vision-language training steps: the word branch on the library's caller lane (forward())
        if obj_feats:
            self.img_encoder = image_encoder
        self.embed, self.diora = embed, diora
        self.loss_func_names = [m.name for m in loss_funcs]
        for m in loss_funcs:
            setattr(self, m.name, m)

    def forward(self, tokens, obj_feats=None, neg_samples=None, compute_loss=True):
        # The word branch of a vision-language training step -- Embed's word projection, ImageEncoder's fc_vis, the word-region scorer and,
        # in the backward, their gradients and the region matrix's half of the region-max backward -- shares nothing with the chart
        # (cliora.py:459-461, trainer.py:139-171): it runs on the library's caller lane beside the chart's forward and backward
        # (autograd runs a node's backward on its forward's stream and orders the streams itself; the lane joins the current stream
        # before the VG loss reads the scores, and again when backward() returns).
        lane = None
        if (self.overlap_word_branch and self.obj_feats and self.training and torch.is_grad_enabled() and tokens.is_cuda and heads.NATIVE_LANES
                and getattr(self.diora, 'lazy_region_scores', False)):
            from . import _lib
            lane = _lib.side_stream(tokens.device)
            lane.wait_stream(torch.cuda.current_stream(tokens.device))     # the step's inputs and the parameters of the last update
        if hasattr(self.diora, 'word_lane'):
            self.diora.word_lane = lane
            self.diora.word_inputs_on_lane = lane is not None
        if lane is not None:
            x_span, x_word = self.embed(tokens, word_lane=lane)
        else:
            x_span, x_word = self.embed(tokens)
        o_span = o_word = None
        if self.obj_feats:
            o_span, o_word = self.img_encoder(obj_feats, word_lane=lane) if lane is not None else self.img_encoder(obj_feats)
        self.diora(x_span, x_word, o_span, o_word)
        if not compute_loss:
            return {'total_loss': torch.ones(1, 1, device=x_span.device)}
        ret, parts = LossDict(), []
        for name in self.loss_func_names:
            fn = getattr(self, name)
            if 'reconstruct' in name:
                v = fn(tokens, neg_samples, self.diora)
            elif 'contrastive' in name:
                v = fn(self.diora)
            else:
                v = fn(self.diora.vg_atten_score)
            ret[name] = v
            parts.append(v)
        ret.parts = parts
        return ret


def select_diora(arch, obj_feats):
    """The module class `build_net` takes for --arch / --obj_feats (trainer.py:518-526).  The reference knows 'mlp' (cliora.DioraMLP with
    obj_feats, diora.DioraMLP without) and raises NotImplementedError for everything else; BASELINE config 5 names the TreeLSTM, which the
    reference ships only as commented text (vg.py:28-76): 'treelstm' selects this library's reconstruction of it (text-only; parity
    unpinned, cliora_amd/treelstm.py), any other name raises like the reference."""
    if arch == 'mlp':
        return VLDiora if obj_feats else TextDiora
    if arch == 'treelstm':
        if obj_feats:
            raise NotImplementedError('arch=treelstm is text-only (the reference has no vision-language TreeLSTM)')
        from .treelstm import DioraTreeLSTM
        return DioraTreeLSTM
    raise NotImplementedError('arch=%r (trainer.py:518-526 knows mlp; this library adds treelstm)' % (arch,))


def build_net(size, embeddings, obj_feats=False, img_dim=2048, k_neg=100, share=True, normalize='unit',
              vg_loss=False, use_contr=False, vl_margin=0.2, alpha_contr=1.0, alpha_vg=1.0, arch='mlp'):
    """What trainer.py:504-582 assembles, on the native chart modules."""
    embed = Embed(embeddings, embeddings.weight.shape[1], size)
    enc = ImageEncoder(img_dim, size)
    diora = select_diora(arch, obj_feats)(size, outside=True, normalize=normalize, compress=False, share=share)
    losses = [ReconstructionSoftmaxLoss(embeddings, embeddings.weight.shape[1], size, k_neg=k_neg)]
    if vg_loss:
        losses.append(VGLoss(alpha_vg))
    if obj_feats and use_contr:
        losses.append(ContrastiveLoss(vl_margin, alpha_contr))
    if obj_feats:
        embeddings.weight.requires_grad = False              # trainer.py:541
    return Net(embed, enc, diora, obj_feats, losses)


class Trainer(object):
    def __init__(self, net, lr=2e-3, reducer=None):
        self.net = net
        self.params = [p for p in net.parameters() if p.requires_grad]
        self.reducer = reducer                               # cliora_amd.parallel.FlatGradAllReduce or None
        self.defer_table_grads = True
        self._loss_host, self._loss_ev, self._loss_pending = None, None, False      # pinned scalar + event of the per-step loss read (step(sync=True))
        self.fused = _native(*self.params)                   # clip + Adam as three launches over one flat buffer (heads.FusedClipAdam)
        if self.fused:
            self.optimizer = heads.FusedClipAdam(self.params, lr=lr, betas=(0.9, 0.999), eps=1e-8, max_norm=5.0, reducer=reducer)
        else:
            self.optimizer = torch.optim.Adam(self.params, lr=lr, betas=(0.9, 0.999), eps=1e-8)

    def step(self, batch_map, train=True, compute_loss=True, sync=True):
        """One training / evaluation step (trainer.py:437-501).  sync=True returns the loss as a Python float like the reference's
        `.item()` (trainer.py:463) -- a device synchronisation per step; sync=False returns the 0-d tensor and lets the host run
        ahead of the device (read it when logging)."""
        self.net.train(train)
        with torch.set_grad_enabled(train):
            out = self.net(batch_map['sentences'], batch_map.get('obj_feats'), batch_map.get('neg_samples'), compute_loss)
        total = out.total() if isinstance(out, LossDict) else out['total_loss'].mean(dim=0).sum()
        # sync=True returns the loss as a Python float like the reference's `.item()` (trainer.py:463).  Its value is final once the FORWARD is:
        # it is copied to pinned host memory here, behind the forward, and read at the end of the step by waiting for THAT copy only -- the
        # host then goes on to enqueue the next step while this one's backward and update still run, instead of waiting for them too.
        loss_ev = None
        if sync and total.is_cuda:
            if self._loss_host is None:
                self._loss_host = torch.empty((), dtype=torch.float32, pin_memory=True)
                self._loss_ev = torch.cuda.Event()
            if self._loss_pending:                        # the previous step's copy into the same pinned scalar (already waited for by its step)
                self._loss_ev.synchronize()
            self._loss_host.copy_(total.detach(), non_blocking=True)
            self._loss_ev.record(torch.cuda.current_stream(total.device))
            self._loss_pending = True
            loss_ev = self._loss_ev
        if train:
            self.optimizer.zero_grad()
            # this step owns its gradients from backward to the update, so the embedding table's gradient is assembled once, in the flat
            # buffer, from every lookup's rows (heads.DeferredTableGrads) instead of one dense scatter per lookup + dense adds
            defer = heads.DeferredTableGrads(self.optimizer.grads) if (self.fused and self.defer_table_grads) else None
            heads._deferred = defer
            heads._step_lanes = lanes = set() if self.fused else None
            try:
                total.backward()
            finally:
                heads._deferred = None
                heads._step_lanes = None
            for lane in (lanes or ()):               # leaf gradients that backward nodes left to the caller lane (heads._step_lane)
                torch.cuda.current_stream(lane.device).wait_stream(lane)
            if defer is not None:
                defer.flush()
            if self.reducer is not None:
                self.reducer.all_reduce_mean()
            if self.fused:
                self.optimizer.step(gathered=self.reducer is not None and self.optimizer.grads is self.reducer)
            else:
                torch.nn.utils.clip_grad_norm_(self.params, 5.0)
                self.optimizer.step()
        if loss_ev is not None:
            loss_ev.synchronize()
            self._loss_pending = False
            return {'total_loss': float(self._loss_host)}
        return {'total_loss': float(total.detach()) if sync else total.detach()}
