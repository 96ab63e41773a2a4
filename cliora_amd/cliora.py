"""MI355X-native drop-in for ``cliora.net.cliora.DioraMLP`` (cliora/net/cliora.py:205-488):
the vision-language chart model.  Same constructor and
``forward(x_span, x_word, obj_embed_span, obj_embed_word) -> None``; besides the six charts it
leaves ``all_atten_score (B,B,C,R)``, ``vg_atten_score (B,B,L,R)`` and ``atten_score (B,L,R)``
on the module (cliora.py:453-468).

Two autograd nodes, both HIP behind the C ABI: the chart (with the AttentionHead residual at
the leaves and in every inside aggregate, cliora.py:28-42, 71-80, 140-157) and the
span-region / word-region scorers.
"""
import ctypes as C

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from .diora import DioraBase, ComposeMLP, Bilinear, Chart, _ptr, _stream, _param_struct, _grad_out
from .index import Index

DROPOUT_P = 0.1   # AttentionHead: nn.Dropout(0.1), cliora.py:32


class VLChartFunction(torch.autograd.Function):
    """cliora_chart_forward / _backward with image regions (R > 0)."""

    @staticmethod
    @_lib.on_device(lambda ctx, plan, holder, run_outside, x_span, *a: x_span)
    def forward(ctx, plan, holder, run_outside, x_span, obj_span, drop_mask, *params):
        if not x_span.is_cuda:
            raise _lib.ChartLibError('the chart path runs on the GPU only (got a CPU tensor)')
        B, L, D, Cc = plan.B, plan.L, plan.D, plan.C
        x_span = x_span.contiguous().float()
        obj_span = obj_span.contiguous().float()
        drop_mask = drop_mask.contiguous().float() if drop_mask is not None else None
        ptens = {n: (p.detach().contiguous() if p is not None else None) for n, p in zip(_lib.PARAM_FIELDS, params)}
        dev = x_span.device
        inside_h = torch.empty((B, Cc, D), device=dev, dtype=torch.float32)
        inside_s = torch.empty((B, Cc, 1), device=dev, dtype=torch.float32)
        outside_h = torch.empty((B, Cc, D), device=dev, dtype=torch.float32)
        outside_s = torch.empty((B, Cc, 1), device=dev, dtype=torch.float32)
        inside_c = torch.zeros((B, Cc, D), device=dev, dtype=torch.float32)
        nbytes = plan.fwd_bytes + (plan.pair_bytes if int(run_outside) & _lib.FWD_PAIR_STATES else 0)
        ws = torch.empty(nbytes, device=dev, dtype=torch.uint8)
        pst = _param_struct(ptens)
        rc = _lib.lib().cliora_chart_forward(plan.handle, C.byref(pst), _ptr(x_span), _ptr(obj_span), _ptr(drop_mask),
                                            _ptr(inside_h), _ptr(inside_s), _ptr(outside_h), _ptr(outside_s), _ptr(inside_c),
                                            _ptr(ws), nbytes, int(run_outside), _stream())
        _lib.check(rc, 'cliora_chart_forward')
        ctx.plan, ctx.run_outside, ctx.ws, ctx.ptens, ctx.drop_mask = plan, int(run_outside) & 1, ws, ptens, drop_mask
        ctx.save_for_backward(x_span, obj_span, inside_h, inside_s, outside_h, outside_s)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(inside_c)
        holder.clear()
        holder.append(ws)
        return inside_h, inside_s, outside_h, outside_s, inside_c

    @staticmethod
    @_lib.on_device(lambda ctx, *a: ctx.saved_tensors[0])
    def backward(ctx, d_ih, d_is, d_oh, d_os, _d_ic):
        plan = ctx.plan
        x_span, obj_span, inside_h, inside_s, outside_h, outside_s = ctx.saved_tensors
        dev = x_span.device
        cont = lambda g: g.contiguous().float() if g is not None else None
        d_ih, d_is, d_oh, d_os = cont(d_ih), cont(d_is), cont(d_oh), cont(d_os)
        grads = {n: (_grad_out(t) if t is not None else None) for n, t in ctx.ptens.items()}
        d_x = torch.empty_like(x_span)
        d_obj = torch.empty_like(obj_span)
        wsb = torch.empty(plan.bwd_bytes, device=dev, dtype=torch.uint8)
        pst, gst = _param_struct(ctx.ptens), _param_struct(grads)
        rc = _lib.lib().cliora_chart_backward(plan.handle, C.byref(pst), _ptr(x_span), _ptr(obj_span), _ptr(ctx.drop_mask),
                                             _ptr(inside_h), _ptr(inside_s), _ptr(outside_h), _ptr(outside_s),
                                             _ptr(d_ih), _ptr(d_is), _ptr(d_oh), _ptr(d_os),
                                             _ptr(ctx.ws), plan.fwd_bytes, _ptr(wsb), plan.bwd_bytes,
                                             _ptr(d_x), _ptr(d_obj), C.byref(gst), ctx.run_outside, _stream())
        _lib.check(rc, 'cliora_chart_backward')
        return (None, None, None, d_x, d_obj, None) + tuple(grads[n] for n in _lib.PARAM_FIELDS)


class VLScoreFunction(torch.autograd.Function):
    """cliora_vl_scores_forward / _backward: the einsum('abx,cdx->acbd') scorers of cliora.py:453-466."""

    @staticmethod
    @_lib.on_device(lambda ctx, plan, training, inside_h, *a: inside_h)
    def forward(ctx, plan, training, inside_h, outside_h, obj_span, x_word, obj_word, want_all):
        """x_word / obj_word None: all_atten only; want_all False (training mode only): vg_atten only."""
        B, L, Cc, R = plan.B, plan.L, plan.C, plan.R
        tens = [t.contiguous().float() if t is not None else None for t in (inside_h, outside_h, obj_span, x_word, obj_word)]
        dev = tens[0].device
        all_att = torch.empty((B, B, Cc, R), device=dev, dtype=torch.float32) if want_all else None
        vg = torch.empty((B, B, L, R), device=dev, dtype=torch.float32) if x_word is not None else None
        nbytes = _lib.lib().cliora_plan_vl_workspace_bytes(plan.handle)
        ws = torch.empty(nbytes, device=dev, dtype=torch.uint8)
        rc = _lib.lib().cliora_vl_scores_forward(plan.handle, *[_ptr(t) for t in tens], int(training), _ptr(all_att), _ptr(vg),
                                                _ptr(ws), nbytes, _stream())
        _lib.check(rc, 'cliora_vl_scores_forward')
        ctx.plan, ctx.training, ctx.nbytes = plan, int(training), nbytes
        ctx.have_words = x_word is not None
        ctx.save_for_backward(*[t for t in tens if t is not None])
        ctx.set_materialize_grads(False)
        return all_att, vg

    @staticmethod
    @_lib.on_device(lambda ctx, *a: ctx.saved_tensors[0])
    def backward(ctx, d_all, d_vg):
        plan = ctx.plan
        if ctx.have_words:
            inside_h, outside_h, obj_span, x_word, obj_word = ctx.saved_tensors
        else:
            (inside_h, outside_h, obj_span), x_word, obj_word = ctx.saved_tensors, None, None
        dev = inside_h.device
        cont = lambda g: g.contiguous().float() if g is not None else None
        d_all, d_vg = cont(d_all), cont(d_vg)
        need = ctx.needs_input_grad          # (plan, training, inside_h, outside_h, obj_span, x_word, obj_word, want_all)
        d_sum = torch.empty_like(inside_h) if (need[2] or need[3]) else None
        d_obj_span = torch.empty_like(obj_span) if need[4] else None
        d_x_word = torch.empty_like(x_word) if (x_word is not None and need[5]) else None
        d_obj_word = torch.empty_like(obj_word) if (obj_word is not None and need[6]) else None
        ws = torch.empty(ctx.nbytes, device=dev, dtype=torch.uint8)
        rc = _lib.lib().cliora_vl_scores_backward(plan.handle, _ptr(inside_h), _ptr(outside_h), _ptr(obj_span), _ptr(x_word),
                                                 _ptr(obj_word), ctx.training, _ptr(d_all), _ptr(d_vg), _ptr(d_sum),
                                                 _ptr(d_obj_span), _ptr(d_x_word), _ptr(d_obj_word), _ptr(ws), ctx.nbytes, _stream())
        _lib.check(rc, 'cliora_vl_scores_backward')
        return None, None, d_sum, d_sum, d_obj_span, d_x_word, d_obj_word, None


class VLMaxFunction(torch.autograd.Function):
    """cliora_vl_scores_max_forward / _backward: max over the R regions of einsum('abx,cdx->acbd', inside_h + outside_h, obj)
    (cliora.py:457 followed by trainer.py:101) without the (B, B, C, R) tensor."""

    @staticmethod
    @_lib.on_device(lambda ctx, plan, inside_h, *a: inside_h)
    def forward(ctx, plan, inside_h, outside_h, obj_span):
        B, Cc = plan.B, plan.C
        tens = [t.contiguous().float() for t in (inside_h, outside_h, obj_span)]
        dev = tens[0].device
        vmax = torch.empty((B, B, Cc), device=dev, dtype=torch.float32)
        arg = torch.empty((B, B, Cc), device=dev, dtype=torch.int32)
        nbytes = _lib.lib().cliora_plan_vl_workspace_bytes(plan.handle)
        ws = torch.empty(nbytes, device=dev, dtype=torch.uint8)
        rc = _lib.lib().cliora_vl_scores_max_forward(plan.handle, *[_ptr(t) for t in tens], _ptr(vmax), _ptr(arg), _ptr(ws), nbytes, _stream())
        _lib.check(rc, 'cliora_vl_scores_max_forward')
        ctx.plan, ctx.nbytes = plan, nbytes
        ctx.save_for_backward(*tens, arg)
        ctx.mark_non_differentiable(arg)
        return vmax, arg

    @staticmethod
    @_lib.on_device(lambda ctx, *a: ctx.saved_tensors[0])
    def backward(ctx, d_max, _d_arg):
        plan = ctx.plan
        inside_h, outside_h, obj_span, arg = ctx.saved_tensors
        if d_max is None:
            return None, None, None, None
        d_max = d_max.contiguous().float()
        d_sum = torch.empty_like(inside_h)
        d_obj = torch.empty_like(obj_span)
        ws = torch.empty(ctx.nbytes, device=inside_h.device, dtype=torch.uint8)
        rc = _lib.lib().cliora_vl_scores_max_backward(plan.handle, _ptr(inside_h), _ptr(outside_h), _ptr(obj_span), _ptr(d_max), _ptr(arg),
                                                     _ptr(d_sum), _ptr(d_obj), _ptr(ws), ctx.nbytes, _stream())
        _lib.check(rc, 'cliora_vl_scores_max_backward')
        return None, d_sum, d_sum, d_obj


class VLMaxRows(torch.autograd.Function):
    """The region-max scorer as the node that owns the gradient of the CHART rows only (d inside_h = d outside_h = d_sum_h): the region
    matrix comes in detached.  Its twin VLMaxObj owns the region matrix's gradient and runs on the library's caller lane, so that the
    chart backward -- which needs d_sum_h -- does not wait for the 0.1-0.2 ms of the region-matrix half (cliora_vl_scores_max_backward
    takes either output as NULL).  JoinGrad hands the one cotangent to both."""

    @staticmethod
    @_lib.on_device(lambda ctx, plan, inside_h, *a: inside_h)
    def forward(ctx, plan, inside_h, outside_h, obj_span):
        B, Cc = plan.B, plan.C
        tens = [t.contiguous().float() for t in (inside_h, outside_h, obj_span)]
        dev = tens[0].device
        vmax = torch.empty((B, B, Cc), device=dev, dtype=torch.float32)
        arg = torch.empty((B, B, Cc), device=dev, dtype=torch.int32)
        nbytes = _lib.lib().cliora_plan_vl_workspace_bytes(plan.handle)
        ws = torch.empty(nbytes, device=dev, dtype=torch.uint8)
        rc = _lib.lib().cliora_vl_scores_max_forward(plan.handle, *[_ptr(t) for t in tens], _ptr(vmax), _ptr(arg), _ptr(ws), nbytes, _stream())
        _lib.check(rc, 'cliora_vl_scores_max_forward')
        ctx.plan, ctx.nbytes = plan, nbytes
        ctx.save_for_backward(*tens, arg)
        ctx.mark_non_differentiable(arg)
        return vmax, arg

    @staticmethod
    @_lib.on_device(lambda ctx, *a: ctx.saved_tensors[0])
    def backward(ctx, d_max, _d_arg):
        plan = ctx.plan
        inside_h, outside_h, obj_span, arg = ctx.saved_tensors
        if d_max is None:
            return None, None, None, None
        d_max = d_max.contiguous().float()
        d_sum = torch.empty_like(inside_h)
        ws = torch.empty(ctx.nbytes, device=inside_h.device, dtype=torch.uint8)
        rc = _lib.lib().cliora_vl_scores_max_backward(plan.handle, _ptr(inside_h), _ptr(outside_h), _ptr(obj_span), _ptr(d_max), _ptr(arg),
                                                     _ptr(d_sum), None, _ptr(ws), ctx.nbytes, _stream())
        _lib.check(rc, 'cliora_vl_scores_max_backward')
        return None, d_sum, d_sum, None


class VLMaxObj(torch.autograd.Function):
    """The region matrix's half of the region-max scorer's backward.  forward: no work (hands `vmax` through); it is called under the
    caller lane's stream so that autograd runs the backward there (a node's backward runs on its forward's stream)."""

    @staticmethod
    def forward(ctx, plan, obj_span, inside_h, outside_h, vmax, arg):
        ctx.plan = plan
        ctx.nbytes = _lib.lib().cliora_plan_vl_workspace_bytes(plan.handle)
        ctx.save_for_backward(inside_h, outside_h, obj_span, arg)
        return vmax.view_as(vmax)

    @staticmethod
    @_lib.on_device(lambda ctx, *a: ctx.saved_tensors[0])
    def backward(ctx, d_max):
        plan = ctx.plan
        inside_h, outside_h, obj_span, arg = ctx.saved_tensors
        if d_max is None:
            return None, None, None, None, None, None
        d_max = d_max.contiguous().float()
        d_obj = torch.empty_like(obj_span)
        ws = torch.empty(ctx.nbytes, device=inside_h.device, dtype=torch.uint8)
        rc = _lib.lib().cliora_vl_scores_max_backward(plan.handle, _ptr(inside_h), _ptr(outside_h), _ptr(obj_span), _ptr(d_max), _ptr(arg),
                                                     None, _ptr(d_obj), _ptr(ws), ctx.nbytes, _stream())
        _lib.check(rc, 'cliora_vl_scores_max_backward')
        return None, d_obj, None, None, None, None


class JoinGrad(torch.autograd.Function):
    """a (values) with b as a second path for the cotangent: forward returns a, backward hands the same cotangent to a's and b's producers."""

    @staticmethod
    def forward(ctx, a, b):
        return a.view_as(a)

    @staticmethod
    def backward(ctx, g):
        return g, g


def region_max(plan, inside_h, outside_h, obj_span, lane):
    """(max over regions, its region) of the span-region scores; with a lane: the region matrix's gradient as a node of its own there."""
    if lane is None or not obj_span.requires_grad or not torch.is_grad_enabled():
        return VLMaxFunction.apply(plan, inside_h, outside_h, obj_span)
    vmax_a, arg = VLMaxRows.apply(plan, inside_h, outside_h, obj_span.detach())
    with torch.cuda.stream(lane):
        vmax_b = VLMaxObj.apply(plan, obj_span, inside_h.detach(), outside_h.detach(), vmax_a.detach(), arg)
    return JoinGrad.apply(vmax_a, vmax_b), arg


class RegionMax(tuple):
    """What `all_atten_score.max(-1)` returns: (values, indices) with the field names of torch.return_types.max."""
    values = property(lambda self: self[0])
    indices = property(lambda self: self[1])


class LazyRegionScores:
    """`all_atten_score` of cliora.py:457 -- einsum('abx,cdx->acbd', inside_h + outside_h, obj_span), (B, B, C, R) -- in training
    mode, before anyone has asked for it.  Its one consumer there, ContrastiveLoss (trainer.py:101), takes `.max(-1).values`:
    that runs the fused scorer (the GEMM's epilogue keeps the maximum and its region per (sentence, image, span); the 124 MB
    tensor at B 64 / L 20 / R 36 is never written or differentiated through).  Anything else -- indexing, torch functions,
    tensor attributes -- materialises the dense tensor once through the ordinary scorer and behaves like it."""

    def __init__(self, plan, inside_h, outside_h, obj_span, lane=None):
        self._plan, self._args, self._lane = plan, (inside_h, outside_h, obj_span), lane
        self._dense, self._max = None, None
        self.shape = torch.Size((plan.B, plan.B, plan.C, plan.R))
        self.device, self.dtype = inside_h.device, torch.float32

    def size(self, dim=None):
        return self.shape if dim is None else self.shape[dim]

    def dim(self):
        return 4

    def max(self, dim=None, keepdim=False):
        if dim in (-1, 3) and not keepdim and self._dense is None:
            if self._max is None:
                vmax, arg = region_max(self._plan, *self._args, self._lane)
                self._max = RegionMax((vmax, arg.long()))
            return self._max
        return self.materialize().max() if dim is None else self.materialize().max(dim, keepdim)

    def materialize(self):
        if self._dense is None:
            ih, oh, obj = self._args
            self._dense = VLScoreFunction.apply(self._plan, True, ih, oh, obj, None, None, True)[0]
        return self._dense

    def __getattr__(self, name):                    # only reached for names not defined above
        if name.startswith('_'):                    # private / dunder probes (copy, pickle, ...) must not materialise or recurse
            raise AttributeError(name)
        return getattr(self.materialize(), name)

    def __getitem__(self, idx):
        return self.materialize()[idx]

    def __len__(self):
        return self.shape[0]

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        un = lambda a: a.materialize() if isinstance(a, LazyRegionScores) else a
        return func(*[un(a) for a in args], **{k: un(v) for k, v in (kwargs or {}).items()})


def _delegate(name):
    def op(self, *a, **k):
        return getattr(self.materialize(), name)(*a, **k)
    op.__name__ = name
    return op


for _n in ('__add__', '__radd__', '__sub__', '__rsub__', '__mul__', '__rmul__', '__truediv__', '__rtruediv__', '__neg__', '__pow__',
           '__matmul__', '__lt__', '__le__', '__gt__', '__ge__', '__eq__', '__ne__', '__iter__', '__repr__', '__array__'):
    setattr(LazyRegionScores, _n, _delegate(_n))
LazyRegionScores.__hash__ = object.__hash__


class AttentionHead(nn.Module):
    """Parameter-free; kept for the module tree and the dropout rate (cliora.py:28-42)."""

    def __init__(self, q_dim, k_dim, v_dim, h_dim):
        super().__init__()
        self.h_dim = h_dim
        self.dropout = nn.Dropout(DROPOUT_P)


class VLComposeMLP(ComposeMLP):
    pass


class DioraMLP(DioraBase):
    vision_language = True

    def init_parameters(self):
        self.atten_head = AttentionHead(self.size, self.size, self.size, self.size)
        self.inside_score_func = Bilinear(self.size)
        self.inside_compose_func = VLComposeMLP(self.size, leaf=True)
        if self.share:
            self.outside_score_func = self.inside_score_func
            self.outside_compose_func = self.inside_compose_func
        else:
            self.outside_score_func = Bilinear(self.size)
            self.outside_compose_func = VLComposeMLP(self.size)
        if self.compress:
            self.root_mat_out = nn.Parameter(torch.empty(self.size, self.size))
        else:
            self.root_vector_out_h = nn.Parameter(torch.empty(self.size))
        self.root_vector_out_c = None
        self.dropout_mask = None      # tests inject a (B, C, R) pre-scaled mask here; None = draw one per forward
        self.lazy_region_scores = True   # training mode: all_atten_score is a LazyRegionScores (False: always the dense tensor)
        # The library's caller lane (include/cliora_chart.h: cliora_device_side_stream) for the work of a training step that the chart does not
        # wait for: the word-region scorer (its inputs may already live there: harness.Net puts the word projections on it) and the region
        # matrix's half of the region-max backward.  None (default) = everything on the current stream; harness.Net sets it per forward.
        self.word_lane = None
        # True: x_word / obj_embed_word were produced ON the lane (harness.Net.forward), so the scorer needs no wait for the current stream --
        # it then runs beside the chart's forward instead of behind it
        self.word_inputs_on_lane = False

    def get_chart_wrapper(self):
        return self

    def forward(self, x_span, x_word, obj_embed_span=None, obj_embed_word=None):
        if self.index is None:
            self.index = Index(cuda=self.is_cuda)
        self.reset()
        if obj_embed_span is None:
            raise ValueError('the CLIORA module needs obj_embed_span (use cliora_amd.diora.DioraMLP for text only)')
        B, L, D = x_span.shape
        R = obj_embed_span.shape[1]
        assert D == self.size
        dev_index = x_span.device.index if x_span.is_cuda else -1
        plan = _lib.get_plan(B, L, D, self.share, self.normalize, R, dev_index)
        mask = None
        if self.training:
            mask = self.dropout_mask
            if mask is None:   # same distribution as nn.Dropout(0.1) on the (.., R) probabilities
                mask = F.dropout(torch.ones((B, plan.C, R), device=x_span.device), DROPOUT_P, True)
        holder = []
        params = self._param_tensors()
        needs_grad = torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in [x_span, obj_embed_span] + list(params))
        hooks = self._hook_overridden('inside_hook') or self._hook_overridden('outside_hook')
        flags = int(bool(self.outside)) | (0 if needs_grad else _lib.FWD_NO_BACKWARD) | (_lib.FWD_PAIR_STATES if hooks else 0)
        ih, is_, oh, os_, ic = VLChartFunction.apply(plan, holder, flags, x_span, obj_embed_span, mask, *params)
        ch = Chart()
        ch.inside_h, ch.inside_s, ch.outside_h, ch.outside_s, ch.inside_c = ih, is_, oh, os_, ic
        ch.outside_c = torch.zeros_like(oh)
        self.chart = ch
        self._wss, self._plan = holder, plan
        self.init_with_batch(ih[:, :L], ic[:, :L])
        self._serve_hooks(L)
        # cliora.py:453-468
        if self.training and self.lazy_region_scores:
            # training: vg_atten does not read all_atten (cliora.py:459-461) and the contrastive loss only wants its region max
            lane = self.word_lane if x_span.is_cuda else None
            if lane is not None:
                # the word-region scorer on the caller lane: neither its forward nor its backward (0.2 ms at c3) is on the chart's path; the
                # chart rows and the span region matrix are not read by it (detached: no gradient edge into the chart either)
                cur = torch.cuda.current_stream(x_span.device)
                if not self.word_inputs_on_lane:      # inputs produced on this stream: the lane waits for them (and for the whole chart before them)
                    lane.wait_stream(cur)
                with torch.cuda.stream(lane):
                    _, vg = VLScoreFunction.apply(plan, True, ih.detach(), oh.detach(), obj_embed_span.detach(), x_word, obj_embed_word, False)
                cur.wait_stream(lane)                 # whoever reads vg next on this stream (the VG loss) finds it complete
                vg.record_stream(cur)
            else:
                _, vg = VLScoreFunction.apply(plan, True, ih, oh, obj_embed_span, x_word, obj_embed_word, False)
            all_att = LazyRegionScores(plan, ih, oh, obj_embed_span, lane)
        else:
            all_att, vg = VLScoreFunction.apply(plan, self.training, ih, oh, obj_embed_span, x_word, obj_embed_word, True)
        self.all_atten_score = all_att
        self.vg_atten_score_word = vg
        self.vg_atten_score = vg
        self.atten_score = torch.diagonal(vg, 0, 0, 1).permute(2, 0, 1)
        return None
