"""The callers either side of the chart path on this library's kernels (include/cliora_chart.h, csrc/api_heads.hip):

  proj          Embed.forward (cliora/net/trainer.py:219-224) and ImageEncoder.forward (cliora/net/utils.py:52-55)
  recon_loss    ReconstructionSoftmaxLoss.forward (trainer.py:46-78)
  vg_loss       VGLoss.forward (trainer.py:139-171)
  FusedClipAdam Trainer.gradient_update (trainer.py:450-455): clip_grad_norm_ + Adam over one flat buffer

Each is a torch.autograd.Function around C-ABI calls; torch owns the memory and the autograd edges only.  The one torch op left on
these paths is the scatter of the looked-up rows' gradient into the embedding table (index_add_), when the table is trainable.
"""
import ctypes as C

import torch

from . import _lib


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _grad_out(t):
    from .diora import _grad_out as g
    return g(t)


def _st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def supported(*tensors):
    return all(t is None or (t.is_cuda and t.dtype in (torch.float32, torch.int64)) for t in tensors)


class DeferredTableGrads(object):
    """The embedding table's gradient of ONE training step, collected instead of scattered (harness.Trainer owns the step, so it may):
    every producer of a backward pass -- the reconstruction loss's positives and negatives (trainer.py:54-58), Embed's lookups (:219) --
    leaves its (rows, index) here and returns no gradient; flush() then writes the table's slice of the flat gradient buffer with ONE
    zero-fill + ONE cliora_rows_scatter_add_segments launch.  Autograd alone scatters each producer into its own dense (V, K) tensor and
    adds them (two 41 MB fills, two scatters and a 41 MB add per step at V 10 000 x 1024).  Producers arrive in autograd's (fixed)
    order, so the sum order -- and every bit of the result -- is the same run after run."""

    def __init__(self, arena):
        self.arena = arena                  # cliora_amd.parallel.FlatGradAllReduce: who owns which parameter's gradient slice
        self.pending = {}                   # id(param) -> (param, view, [(rows, index), ...])

    def offer(self, d_rows, index, table):
        """True when the contribution was taken (the caller then returns None as the table's gradient)."""
        pv = self.arena.lookup(table)
        if pv is None or pv[0].grad is not None or pv[1].shape != table.shape:
            return False
        ent = self.pending.setdefault(id(pv[0]), (pv[0], pv[1], []))
        if len(ent[2]) >= 4:                # cliora_rows_scatter_add_segments takes four segments
            return False
        ent[2].append((d_rows.contiguous(), index.contiguous().reshape(-1)))
        return True

    def flush(self):
        for param, view, segs in self.pending.values():
            n = len(segs)
            rows = (C.c_void_p * n)(*[t.data_ptr() for t, _ in segs])
            idx = (C.c_void_p * n)(*[i.data_ptr() for _, i in segs])
            cnt = (C.c_int32 * n)(*[int(i.numel()) for _, i in segs])
            with torch.cuda.device(view.device):
                _lib.check(_lib.lib().cliora_rows_scatter_add_segments(rows, idx, cnt, n, int(view.shape[1]), _p(view), int(view.shape[0]), _st()),
                           'cliora_rows_scatter_add_segments')
            param.grad = view.detach()      # an alias of the arena slice: the reducer / FusedClipAdam see a gradient already in place
        self.pending = {}


_step_lanes = None      # inside harness.Trainer.step: the set of caller lanes that backward nodes put work on (joined before the update)


def _step_lane(device):
    """The library's caller lane for work that only the parameter update waits for -- available while harness.Trainer.step runs the
    backward pass (it joins the lane before the all-reduce / clip + Adam), else None."""
    if _step_lanes is None or not NATIVE_LANES:
        return None
    lane = _lib.side_stream(device)
    _step_lanes.add(lane)
    return lane


NATIVE_LANES = True     # False: never use the library's caller lane (harness.Net.forward), everything on the current stream (A/B, tests)
_deferred = None        # a DeferredTableGrads while harness.Trainer.step runs a backward pass, else None


def scatter_rows(d_rows, index, table, producer_lane=None):
    """Gradient of the embedding `table` (V, K) from the gradients d_rows (n, K) of its looked-up rows index (n,): cliora_rows_scatter_add,
    written into the table's slice of a live flat gradient buffer when this is the first producer of the pass (else a fresh tensor that
    autograd adds).  Replaces zeros_like + index_add_ (round 3's last ATen op on the step), deterministic for repeated ids.
    Inside harness.Trainer.step the contribution is deferred instead (DeferredTableGrads) and None is returned.
    producer_lane: the stream d_rows is produced on when that is not the current one."""
    if _deferred is not None and _deferred.offer(d_rows, index, table):
        return None
    if producer_lane is not None:           # d_rows is still being written on another stream and the scatter runs here, now
        torch.cuda.current_stream(table.device).wait_stream(producer_lane)
    out = _grad_out(table)
    with torch.cuda.device(table.device):
        _lib.check(_lib.lib().cliora_rows_scatter_add(_p(d_rows.contiguous()), _p(index.contiguous()), int(index.numel()), int(table.shape[1]), _p(out),
                                                      int(table.shape[0]), _st()), 'cliora_rows_scatter_add')
    return out


class Proj(torch.autograd.Function):
    """y = gather(x, index) w^T + bias; index None = the rows of x themselves."""

    @staticmethod
    def forward(ctx, x, index, w, bias):
        x, w = x.contiguous().float(), w.contiguous().float()
        bias = bias.contiguous().float() if bias is not None else None
        index = index.contiguous() if index is not None else None
        K, D = x.shape[-1], w.shape[0]
        x2 = x.reshape(-1, K)
        nrows = index.numel() if index is not None else x2.shape[0]
        lib = _lib.lib()
        with torch.cuda.device(x.device):
            y = torch.empty((nrows, D), device=x.device, dtype=torch.float32)
            nb = lib.cliora_proj_workspace_bytes(nrows, K, D)
            ws = torch.empty(nb, device=x.device, dtype=torch.uint8)
            _lib.check(lib.cliora_proj_forward(_p(x2), _p(index), nrows, K, _p(w), _p(bias), D, _p(y), _p(ws), nb, _st()), 'cliora_proj_forward')
        ctx.save_for_backward(x2, index, w)
        ctx.has_bias, ctx.x_shape, ctx.ws, ctx.bias = bias is not None, x.shape, ws, (bias.detach() if bias is not None else None)
        return y

    @staticmethod
    def backward(ctx, d_y):
        x2, index, w = ctx.saved_tensors
        d_y = d_y.contiguous().float()
        nrows, D = d_y.shape
        K = x2.shape[1]
        lib = _lib.lib()
        need_x = ctx.needs_input_grad[0]
        with torch.cuda.device(d_y.device):
            # the parameter's slice of a live flat gradient buffer when there is one (cliora_amd.parallel: the all-reduce and the fused
            # clip + Adam then have nothing to copy), else a fresh tensor
            d_w = _grad_out(w) if ctx.needs_input_grad[2] else None
            d_b = _grad_out(ctx.bias) if (ctx.has_bias and ctx.needs_input_grad[3]) else None
            d_rows = torch.empty((nrows, K), device=d_y.device) if need_x else None
            nb = ctx.ws.numel()
            _lib.check(lib.cliora_proj_backward(_p(x2), _p(index), nrows, K, _p(w), _p(d_y), D, _p(d_w), _p(d_b), _p(d_rows), _p(ctx.ws), nb, _st()),
                       'cliora_proj_backward')
            d_x = None
            if need_x:
                if index is None:
                    d_x = d_rows.view(ctx.x_shape)
                else:       # the embedding table's gradient: rows of repeated tokens add up (in place in the flat gradient buffer when there is one)
                    d_x = scatter_rows(d_rows, index.reshape(-1), x2)
                    d_x = d_x.view(ctx.x_shape) if d_x is not None else None
        return d_x, None, d_w, d_b


def _pad16(t):
    """Zero-pad the last dimension to a multiple of 16 (differentiable).  The kernels reduce over 16-wide chunks; the reference's own
    embedding widths include 300 (GloVe, scripts/train.py:320) and 1324 (w2v + ELMo, data/embeddings.py:141), which are not."""
    k = t.shape[-1]
    return t if k % 16 == 0 else torch.nn.functional.pad(t, (0, 16 - k % 16))


def proj(x, index, w, bias=None):
    return Proj.apply(_pad16(x), index, _pad16(w), bias)


class ReconLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, emb, mat, outside_h, tokens, neg):
        emb, mat, outside_h = emb.contiguous().float(), mat.contiguous().float(), outside_h.contiguous().float()
        tokens, neg = tokens.contiguous(), neg.contiguous().reshape(-1)
        B, L = tokens.shape
        Cc, D = outside_h.shape[1], outside_h.shape[2]
        E, Kn = emb.shape[1], neg.numel()
        lib = _lib.lib()
        with torch.cuda.device(emb.device):
            loss = torch.empty(1, device=emb.device)
            nb = lib.cliora_recon_workspace_bytes(B * L, Kn, E, D)
            ws = torch.empty(nb, device=emb.device, dtype=torch.uint8)
            _lib.check(lib.cliora_recon_forward(_p(tokens), _p(neg), B, L, Cc, Kn, _p(emb), E, _p(mat), D, _p(outside_h), _p(loss), _p(ws), nb, _st()),
                       'cliora_recon_forward')
        ctx.save_for_backward(emb, mat, outside_h, tokens, neg)
        ctx.ws = ws
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        emb, mat, outside_h, tokens, neg = ctx.saved_tensors
        B, L = tokens.shape
        Cc, D = outside_h.shape[1], outside_h.shape[2]
        E, Kn = emb.shape[1], neg.numel()
        lib = _lib.lib()
        with torch.cuda.device(emb.device):
            g = g.contiguous().float().reshape(1)
            d_cell = torch.empty((B * L, D), device=emb.device) if ctx.needs_input_grad[2] else None
            d_mat = _grad_out(mat) if ctx.needs_input_grad[1] else None
            d_rows = torch.empty((B * L + Kn, E), device=emb.device) if ctx.needs_input_grad[0] else None
            nb = ctx.ws.numel()
            args = lambda dc, dm, dr: (_p(tokens), _p(neg), B, L, Cc, Kn, _p(emb), E, _p(mat), D, _p(outside_h), _p(g), _p(dc), _p(dm), _p(dr), _p(ctx.ws), nb, _st())
            lane = _step_lane(emb.device) if (d_cell is not None and (d_mat is not None or d_rows is not None)) else None
            if lane is None:
                _lib.check(lib.cliora_recon_backward(*args(d_cell, d_mat, d_rows)), 'cliora_recon_backward')
            else:
                # inside harness.Trainer.step: the chart backward waits for d cell only; the projection's and the looked-up rows' gradients
                # (two GEMMs, 0.07 ms at c2) follow on the caller lane, behind the first call (they share its workspace), beside the chart
                _lib.check(lib.cliora_recon_backward(*args(d_cell, None, None)), 'cliora_recon_backward')
                cur = torch.cuda.current_stream(emb.device)
                lane.wait_stream(cur)
                with torch.cuda.stream(lane):
                    _lib.check(lib.cliora_recon_backward(*args(None, d_mat, d_rows)), 'cliora_recon_backward')
            d_oh = None
            if d_cell is not None:
                d_oh = torch.zeros_like(outside_h)
                d_oh[:, :L] = d_cell.view(B, L, D)
            d_emb = None
            if d_rows is not None:
                d_emb = scatter_rows(d_rows, torch.cat([tokens.reshape(-1), neg]), emb, producer_lane=lane)
        return d_emb, d_mat, d_oh, None, None


def recon_loss(emb, mat, outside_h, tokens, neg):
    # emb (V, E), mat (D, E): the embedding width E is the reduction of the lookup projection
    return ReconLoss.apply(_pad16(emb), _pad16(mat), outside_h, tokens, neg)


class VGLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vg, alpha):
        vg = vg.contiguous().float()
        B, _, L, R = vg.shape
        lib = _lib.lib()
        with torch.cuda.device(vg.device):
            loss = torch.empty(1, device=vg.device)
            d_vg = torch.empty_like(vg) if ctx.needs_input_grad[0] else None
            nb = lib.cliora_vg_workspace_bytes(B, L)
            ws = torch.empty(nb, device=vg.device, dtype=torch.uint8)
            _lib.check(lib.cliora_vg_loss(B, L, R, _p(vg), float(alpha), _p(loss), _p(d_vg), _p(ws), nb, _st()), 'cliora_vg_loss')
        ctx.d_vg = d_vg
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        return (ctx.d_vg * g if ctx.d_vg is not None else None), None


def vg_loss(vg_atten, alpha):
    return VGLossFn.apply(vg_atten, alpha)


class FusedClipAdam(object):
    """clip_grad_norm_(params, max_norm) + Adam.step() as three launches over one flat buffer (cliora_clip_adam).

    The parameters are re-pointed at slices of one flat tensor (their values, names and shapes unchanged); the gradients are
    gathered into a second one -- the data-parallel reducer's buffer when there is one (cliora_amd.parallel.FlatGradAllReduce:
    the chart backward already writes there), else a buffer of the same kind owned here."""

    def __init__(self, params, lr=2e-3, betas=(0.9, 0.999), eps=1e-8, max_norm=5.0, reducer=None):
        from .parallel import FlatGradAllReduce
        self.params = [p for p in params if p.requires_grad]
        self.lr, self.betas, self.eps, self.max_norm = lr, betas, eps, max_norm
        dev = self.params[0].device
        n = sum(p.numel() for p in self.params)
        self.flat_p = torch.empty(n, device=dev, dtype=torch.float32)
        o = 0
        for p in self.params:
            v = self.flat_p[o:o + p.numel()].view_as(p)
            v.copy_(p.data)
            p.data = v
            o += p.numel()
        if reducer is not None and [id(p) for p in reducer.params] == [id(p) for p in self.params]:
            self.grads = reducer
        else:
            self.grads = FlatGradAllReduce(self.params)        # used as a gradient arena only: never all-reduced here
        self.m = torch.zeros_like(self.flat_p)
        self.v = torch.zeros_like(self.flat_p)
        self.t = 0
        # parameters that have never received a gradient (embed.mat1 of a text-only net: the word projection feeds only the vision-language
        # scorers) have zero moments and a zero arena slice: the dense launch leaves them exactly where they are (0 / (0 + eps)), so they
        # need neither the save / restore of the skipped segments nor a fill per step
        self.ever = [False] * len(self.params)
        self.grads.flat.zero_()
        self.ws = torch.empty(_lib.lib().cliora_clip_adam_workspace_bytes(), device=dev, dtype=torch.uint8)

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    def step(self, gathered=False):
        """gathered=True: the reducer has just brought every gradient into the flat buffer (all_reduce_mean)."""
        ar = self.grads
        skipped = []              # parameters without a gradient this step: torch.optim.Adam leaves them (and their moments) untouched
        if not gathered:
            o = 0
            for k, (p, v) in enumerate(zip(ar.params, ar.views)):
                if p.grad is None:
                    if self.ever[k]:
                        v.zero_()
                        skipped.append((o, p.numel()))
                else:
                    self.ever[k] = True
                    if p.grad.data_ptr() != v.data_ptr():
                        v.copy_(p.grad)
                o += p.numel()
        else:
            self.ever = [True] * len(self.params)          # the reducer zero-filled and averaged every slice: plain dense update
        # a zero gradient is not "no gradient" to Adam (the moments decay and the parameter keeps moving on its momentum): the
        # skipped segments are put back after the launch.  (Their bias-correction step count still advances with the rest: a
        # parameter that receives gradients only on SOME steps is corrected with the global count here, with its own in torch.)
        saved = [(o, n, self.flat_p[o:o + n].clone(), self.m[o:o + n].clone(), self.v[o:o + n].clone()) for o, n in skipped]
        self.t += 1
        with torch.cuda.device(self.flat_p.device):
            _lib.check(_lib.lib().cliora_clip_adam(_p(self.flat_p), _p(ar.flat), _p(self.m), _p(self.v), self.flat_p.numel(), float(self.max_norm),
                                                  float(self.lr), float(self.betas[0]), float(self.betas[1]), float(self.eps), self.t, _p(self.ws),
                                                  self.ws.numel(), _st()), 'cliora_clip_adam')
        for o, n, p0, m0, v0 in saved:
            self.flat_p[o:o + n].copy_(p0); self.m[o:o + n].copy_(m0); self.v[o:o + n].copy_(v0)
        # the clipped values are what the kernel applied; gradients that live outside the arena (torch-produced, copied in above)
        # keep their unclipped values in p.grad -- callers that read p.grad after step() see the arena's clipped copy only for the
        # in-place ones

