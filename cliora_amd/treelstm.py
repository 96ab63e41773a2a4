"""``DioraTreeLSTM`` on the MI355X chart engine (BASELINE config 5).

PARITY UNPINNED.  This tree of the reference cannot build a TreeLSTM model: ``build_net`` raises
for any ``--arch`` but ``mlp`` (cliora/net/trainer.py:518-526) and the composition exists only as
commented-out text (cliora/net/vg.py:28-76).  What is implemented is that text on the
``DioraBase`` skeleton (cliora/net/diora.py:205-450) the way the original DIORA wires it: inside and
outside functions shared or (``share=False``, diora.py:459-464) a second compose / score module for the
outside pass, ``constant`` = 1 inside and 0 outside (diora.py:174), and ``root_vector_out_c`` a
parameter (the hint at diora.py:470-471).  The oracle's restatement is
checked against the commented text executed in memory (tests/golden/treelstm_recon.npz).

Parameters: ``inside_compose_func.W (3D,D)``, ``.U (5D,2D)``, ``.B (5D)``, ``inside_score_func.mat``,
``root_vector_out_h``, ``root_vector_out_c``; ``outside_*`` alias the inside modules, or with ``share=False`` are
``outside_compose_func.{U, B}`` and ``outside_score_func.mat`` of their own.
"""
import ctypes as C

import torch
import torch.nn as nn

from . import _lib
from .diora import DioraBase, Bilinear, Chart, _ptr, _stream, _param_struct, _grad_out
from .index import Index


class TreeLSTM(nn.Module):
    def __init__(self, size, ninput=2, leaf=False):
        super().__init__()
        self.size, self.ninput = size, ninput
        if leaf:
            self.W = nn.Parameter(torch.empty(3 * size, size))
        self.U = nn.Parameter(torch.empty(5 * size, ninput * size))
        self.B = nn.Parameter(torch.empty(5 * size))


class LSTMChartFunction(torch.autograd.Function):
    """cliora_lstm_forward / cliora_lstm_backward; six chart outputs (h, c, s for inside and outside)."""

    NAMES = ('lstm_w', 'lstm_u', 'lstm_b', 'in_mat', 'root_h', 'root_c', 'lstm_u_out', 'lstm_b_out', 'out_mat')

    @staticmethod
    @_lib.on_device(lambda ctx, plan, holder, run_outside, x_span, *a: x_span)
    def forward(ctx, plan, holder, run_outside, x_span, *params):
        if not x_span.is_cuda:
            raise _lib.ChartLibError('the chart path runs on the GPU only (got a CPU tensor)')
        B, D, Cc = plan.B, plan.D, plan.C
        x_span = x_span.contiguous().float()
        ptens = {n: p.detach().contiguous() for n, p in zip(LSTMChartFunction.NAMES, params) if p is not None}
        dev = x_span.device
        new = lambda w: torch.empty((B, Cc, w), device=dev, dtype=torch.float32)
        ih, ic, is_, oh, oc, os_ = new(D), new(D), new(1), new(D), new(D), new(1)
        ws = torch.empty(plan.fwd_bytes, device=dev, dtype=torch.uint8)
        pst = _param_struct(ptens)
        rc = _lib.lib().cliora_lstm_forward(plan.handle, C.byref(pst), _ptr(x_span), _ptr(ih), _ptr(ic), _ptr(is_), _ptr(oh), _ptr(oc),
                                           _ptr(os_), _ptr(ws), plan.fwd_bytes, int(run_outside), _stream())
        _lib.check(rc, 'cliora_lstm_forward')
        ctx.plan, ctx.run_outside, ctx.ws, ctx.ptens = plan, int(run_outside) & 1, ws, ptens
        ctx.save_for_backward(x_span, ih, ic, is_, oh, oc, os_)
        ctx.set_materialize_grads(False)
        holder.clear()
        holder.append(ws)
        return ih, ic, is_, oh, oc, os_

    @staticmethod
    @_lib.on_device(lambda ctx, *a: ctx.saved_tensors[0])
    def backward(ctx, *cots):
        plan = ctx.plan
        x_span, ih, ic, is_, oh, oc, os_ = ctx.saved_tensors
        dev = x_span.device
        cots = [g.contiguous().float() if g is not None else None for g in cots]
        grads = {n: _grad_out(t) for n, t in ctx.ptens.items()}
        d_x = torch.empty_like(x_span)
        wsb = torch.empty(plan.bwd_bytes, device=dev, dtype=torch.uint8)
        pst, gst = _param_struct(ctx.ptens), _param_struct(grads)
        rc = _lib.lib().cliora_lstm_backward(plan.handle, C.byref(pst), _ptr(x_span), _ptr(ih), _ptr(ic), _ptr(is_), _ptr(oh), _ptr(oc),
                                            _ptr(os_), *[_ptr(g) for g in cots], _ptr(ctx.ws), plan.fwd_bytes, _ptr(wsb),
                                            plan.bwd_bytes, _ptr(d_x), C.byref(gst), ctx.run_outside, _stream())
        _lib.check(rc, 'cliora_lstm_backward')
        return (None, None, None, d_x) + tuple(grads.get(n) for n in LSTMChartFunction.NAMES)


class DioraTreeLSTM(DioraBase):
    def init_parameters(self):
        if self.compress:
            raise NotImplementedError('DioraTreeLSTM with compress=True is not built (the reference never enables compress: trainer.py:552)')
        self.inside_score_func = Bilinear(self.size)
        self.inside_compose_func = TreeLSTM(self.size, leaf=True)
        if self.share:
            self.outside_score_func = self.inside_score_func
            self.outside_compose_func = self.inside_compose_func
        else:                       # diora.py:462-464
            self.outside_score_func = Bilinear(self.size)
            self.outside_compose_func = TreeLSTM(self.size)
        self.root_vector_out_h = nn.Parameter(torch.empty(self.size))
        self.root_vector_out_c = nn.Parameter(torch.empty(self.size))

    def forward(self, x_span, x_word=None, obj_embed_span=None, obj_embed_word=None):
        if self.index is None:
            self.index = Index(cuda=self.is_cuda)
        self.reset()
        if obj_embed_span is not None:
            raise NotImplementedError('DioraTreeLSTM is text-only')
        B, L, D = x_span.shape
        assert D == self.size
        dev_index = x_span.device.index if x_span.is_cuda else -1
        plan = _lib.get_plan(B, L, D, self.share, self.normalize, 0, dev_index, arch=1)
        cf, of = self.inside_compose_func, self.outside_compose_func
        outer = (None, None, None) if self.share else (of.U, of.B, self.outside_score_func.mat)
        holder = []
        # eval / torch.no_grad without a hook override: the per-split rows h_n, c_n are not written (flag bits of cliora_chart_forward)
        params = (cf.W, cf.U, cf.B, self.inside_score_func.mat, self.root_vector_out_h, self.root_vector_out_c) + outer
        needs_grad = torch.is_grad_enabled() and (x_span.requires_grad or any(t is not None and t.requires_grad for t in params))
        hooks = self._hook_overridden('inside_hook') or self._hook_overridden('outside_hook')
        flags = int(bool(self.outside)) | (0 if needs_grad else _lib.FWD_NO_BACKWARD) | (_lib.FWD_PAIR_STATES if hooks else 0)
        ih, ic, is_, oh, oc, os_ = LSTMChartFunction.apply(plan, holder, flags, x_span, cf.W, cf.U, cf.B,
                                                           self.inside_score_func.mat, self.root_vector_out_h, self.root_vector_out_c, *outer)
        ch = Chart()
        ch.inside_h, ch.inside_c, ch.inside_s, ch.outside_h, ch.outside_c, ch.outside_s = ih, ic, is_, oh, oc, os_
        self.chart = ch
        self._wss, self._plan = holder, plan
        self.init_with_batch(ih[:, :L], ic[:, :L])
        # hooks: per-split h (the pair rows the backward keeps anyway) and scores, in the reference's layout and order
        # (diora.py:331, 398); the per-split cell states are not kept: c is passed as zeros, as for DioraMLP
        self._serve_hooks(L)
        return None
