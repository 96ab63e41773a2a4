"""MI355X-native drop-in for ``cliora.net.diora.DioraMLP`` (cliora/net/diora.py:205-471).

Same constructor, same ``forward(x_span, x_word, obj_embed_span=None,
obj_embed_word=None) -> None`` with results left on attributes, same parameter
names (state_dict keys), same hooks -- but the whole chart recursion (leaf
transform, inside pass, outside pass) and its backward run as hand-written HIP
kernels behind the C ABI of include/cliora_chart.h.  torch only owns the device
memory, the stream and the autograd edge.  There is no CPU fallback: without
the HIP library (or on CPU tensors) ``forward`` raises.
"""
import ctypes as C

import torch
import torch.nn as nn

from . import _lib
from .index import Index


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _param_struct(tensors):
    return _lib.Params(*[(t.data_ptr() if t is not None else None) for t in map(tensors.get, _lib.PARAM_FIELDS)])


class _OnDevice(object):
    """`with torch.cuda.device(dev)` only when `dev` is not the current device already (the context manager costs ~10 us of host
    time per use, and a configs[0]-sized step is bound by the host: tools/host_bench.py)."""
    __slots__ = ('ctx',)

    def __init__(self, dev):
        self.ctx = None if dev.index is None or dev.index == torch.cuda.current_device() else torch.cuda.device(dev)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *a):
        if self.ctx is not None:
            return self.ctx.__exit__(*a)


def _grad_out(t):
    """Output tensor for the gradient of the parameter that `t` (its detached view) belongs to: the parameter's slice of a live
    flat all-reduce buffer (cliora_amd.parallel: the backward then writes straight into what RCCL reduces), else a fresh tensor."""
    global _grad_buffer_for
    if _grad_buffer_for is None:
        from .parallel import grad_buffer_for as _g
        _grad_buffer_for = _g
    v = _grad_buffer_for(t)
    return v if v is not None else torch.empty_like(t)


_grad_buffer_for = None


class ChartFunction(torch.autograd.Function):
    """One autograd node for the whole inside-outside chart.

    forward  -> cliora_chart_forward   (replaces diora.py:283-398 as executed by DioraBase.forward)
    backward -> cliora_chart_backward  (replaces what autograd replays through those lines)
    Inputs: plan, holder (list that receives the forward workspace), run_outside, x_span, then the parameter
    tensors in _lib.PARAM_FIELDS order (None for the out_* set when the weights are shared).
    Every C-ABI call runs under the tensors' device (the plan's index tables and the kernels' LDS attribute are per device).
    """

    @staticmethod
    def forward(ctx, plan, holder, run_outside, x_span, *params):
        if not x_span.is_cuda:
            raise _lib.ChartLibError('the chart path runs on the GPU only (got a CPU tensor)')
        B, L, D, Cc = plan.B, plan.L, plan.D, plan.C
        x_span = x_span.contiguous().float()
        ptens = {n: (p.detach().contiguous() if p is not None else None) for n, p in zip(_lib.PARAM_FIELDS, params)}
        dev = x_span.device
        with _OnDevice(dev):
            inside_h = torch.empty((B, Cc, D), device=dev, dtype=torch.float32)
            inside_s = torch.empty((B, Cc, 1), device=dev, dtype=torch.float32)
            outside_h = torch.empty((B, Cc, D), device=dev, dtype=torch.float32)
            outside_s = torch.empty((B, Cc, 1), device=dev, dtype=torch.float32)
            nbytes = plan.fwd_bytes + (plan.pair_bytes if int(run_outside) & _lib.FWD_PAIR_STATES else 0)
            ws = torch.empty(nbytes, device=dev, dtype=torch.uint8)
            pst = _param_struct(ptens)
            rc = _lib.lib().cliora_chart_forward(
                plan.handle, C.byref(pst), _ptr(x_span), None, None, _ptr(inside_h), _ptr(inside_s),
                _ptr(outside_h), _ptr(outside_s), None, _ptr(ws), nbytes, int(run_outside), _stream())
            _lib.check(rc, 'cliora_chart_forward')
        ctx.plan, ctx.run_outside, ctx.ws, ctx.ptens = plan, int(run_outside) & 1, ws, ptens
        ctx.save_for_backward(x_span, inside_h, inside_s, outside_h, outside_s)
        ctx.set_materialize_grads(False)
        holder.clear()
        holder.append(ws)
        return inside_h, inside_s, outside_h, outside_s

    @staticmethod
    def backward(ctx, d_ih, d_is, d_oh, d_os):
        plan = ctx.plan
        x_span, inside_h, inside_s, outside_h, outside_s = ctx.saved_tensors
        dev = x_span.device
        cont = lambda g: g.contiguous().float() if g is not None else None
        d_ih, d_is, d_oh, d_os = cont(d_ih), cont(d_is), cont(d_oh), cont(d_os)
        with _OnDevice(dev):
            d_x = torch.empty_like(x_span)
            pst = _param_struct(ctx.ptens)
            g = {n: (_grad_out(t) if t is not None else None) for n, t in ctx.ptens.items()}
            wsb = torch.empty(plan.bwd_bytes, device=dev, dtype=torch.uint8)
            gst = _param_struct(g)
            rc = _lib.lib().cliora_chart_backward(
                plan.handle, C.byref(pst), _ptr(x_span), None, None, _ptr(inside_h), _ptr(inside_s),
                _ptr(outside_h), _ptr(outside_s), _ptr(d_ih), _ptr(d_is), _ptr(d_oh), _ptr(d_os),
                _ptr(ctx.ws), plan.fwd_bytes, _ptr(wsb), plan.bwd_bytes, _ptr(d_x), None, C.byref(gst),
                ctx.run_outside, _stream())
            _lib.check(rc, 'cliora_chart_backward')
        return (None, None, None, d_x) + tuple(g[n] for n in _lib.PARAM_FIELDS)


# --------------------------------------------------------------------------------------
# Parameter containers with the reference's module tree, so state_dict keys match
# (diora.py:26-97, 453-471).  Their forward() is never used on the product path.
# --------------------------------------------------------------------------------------
class ComposeMLP(nn.Module):
    def __init__(self, size, ninput=2, leaf=False):
        super().__init__()
        self.size, self.ninput = size, ninput
        if leaf:
            self.leaf_fc = nn.Linear(size, size)
        self.h_fcs = nn.Sequential(nn.Linear(2 * size, size), nn.ReLU(), nn.Linear(size, size), nn.ReLU())


class Bilinear(nn.Module):
    def __init__(self, size):
        super().__init__()
        self.size = size
        self.mat = nn.Parameter(torch.empty(size, size))


class Chart(object):
    """Attribute bag with the six chart tensors (diora.py:7-23).  DioraMLP's cell states are identically zero (diora.py:60-61, 70):
    they are created when first read (two fill launches and ~20 us of host time per step otherwise, for tensors nothing on the
    training path reads)."""
    __slots__ = ('inside_h', 'inside_s', 'outside_h', 'outside_s', '_ic', '_oc')

    def __init__(self):
        self._ic = self._oc = None

    @property
    def inside_c(self):
        if self._ic is None:
            self._ic = torch.zeros_like(self.inside_h)
        return self._ic

    @inside_c.setter
    def inside_c(self, v):
        self._ic = v

    @property
    def outside_c(self):
        if self._oc is None:
            self._oc = torch.zeros_like(self.outside_h)
        return self._oc

    @outside_c.setter
    def outside_c(self, v):
        self._oc = v


class DioraBase(nn.Module):
    vision_language = False

    def __init__(self, size, word_mat=None, cate_mat=None, outside=True, normalize='unit', compress=False, share=True):
        super().__init__()
        assert normalize in ('none', 'unit'), 'Does not support "{}".'.format(normalize)
        self.size = size
        self.share = share
        self.outside = outside
        self.normalize = normalize
        self.compress = compress
        self.ninput = 2
        self.index = None
        self.charts = None
        self.init_parameters()
        self.reset_parameters()
        self.reset()

    def init_parameters(self):
        raise NotImplementedError

    def reset_parameters(self):
        # every parameter ~ N(0, 1): diora.py:234-237
        for p in self.parameters():
            if p.requires_grad:
                p.data.normal_()

    # ---- attribute surface read by the losses / CKY / eval scripts
    @property
    def device(self):
        return next(self.parameters()).device

    @property
    def is_cuda(self):
        d = self.device
        return d.index is not None and d.index >= 0

    inside_h = property(lambda self: self.chart.inside_h)
    inside_c = property(lambda self: self.chart.inside_c)
    inside_s = property(lambda self: self.chart.inside_s)
    outside_h = property(lambda self: self.chart.outside_h)
    outside_c = property(lambda self: self.chart.outside_c)
    outside_s = property(lambda self: self.chart.outside_s)

    def cuda(self, device=None):
        super().cuda(device)
        if self.index is not None:
            self.index.cuda = True
        return self

    def get(self, chart, level):
        L = self.length - level
        offset = self.index.get_offset(self.length)[level]
        return chart[:, offset:offset + L]

    def inside_hook(self, level, h, c, s):
        pass

    def outside_hook(self, level, h, c, s):
        pass

    def init_with_batch(self, h, c):
        # the native forward has already filled the chart; kept as an overridable method because
        # analysis/utils.py:67-76 wraps it with types.MethodType to attach its score store
        d = self.__dict__
        d['batch_size'], d['length'] = h.shape[0], h.shape[1]

    def reset(self):
        # plain attributes, written straight into the instance dict: nn.Module.__setattr__ walks its parameter / buffer / module
        # registries for every assignment (~3.5 us each; a configs[0]-sized step is bound by the host)
        self.__dict__.update(batch_size=None, length=None, chart=None, atten_score=None, all_atten_score=None, vg_atten_score=None,
                             _wss=None, _plan=None)

    def _hook_overridden(self, name):
        return name in self.__dict__ or getattr(type(self), name) is not getattr(DioraBase, name)

    def _param_tensors(self):
        # through the module / parameter registries directly: attribute access on an nn.Module falls back to __getattr__ for every
        # submodule and parameter (always the live objects: nothing is cached)
        mods, pars = self._modules, self._parameters
        ic = mods['inside_compose_func']._modules
        fc, h0, h2 = ic['leaf_fc']._parameters, ic['h_fcs']._modules['0']._parameters, ic['h_fcs']._modules['2']._parameters
        t = dict(leaf_w=fc['weight'], leaf_b=fc['bias'], in_w1=h0['weight'], in_b1=h0['bias'], in_w2=h2['weight'], in_b2=h2['bias'],
                 in_mat=mods['inside_score_func']._parameters['mat'])
        if self.compress:           # diora.py:342-343, 466-467: the outside root is inside_h[root] @ root_mat_out
            t['root_mat'] = pars['root_mat_out']
        else:
            t['root_h'] = pars['root_vector_out_h']
        if not self.share:
            oc = mods['outside_compose_func']._modules['h_fcs']._modules
            o0, o2 = oc['0']._parameters, oc['2']._parameters
            t.update(out_w1=o0['weight'], out_b1=o0['bias'], out_w2=o2['weight'], out_b2=o2['bias'],
                     out_mat=mods['outside_score_func']._parameters['mat'])
        return [t.get(n) for n in _lib.PARAM_FIELDS]

    def forward(self, x_span, x_word=None, obj_embed_span=None, obj_embed_word=None):
        if self.index is None:
            self.index = Index(cuda=self.is_cuda)
        self.reset()
        if obj_embed_span is not None:
            raise NotImplementedError('text-only DIORA module; use cliora_amd.cliora.DioraMLP with image regions')
        B, L, D = x_span.shape
        assert D == self.size
        dev_index = x_span.device.index if x_span.is_cuda else -1
        plan = _lib.get_plan(B, L, D, self.share, self.normalize, 0, dev_index)
        holder = []
        # eval / torch.no_grad: nothing will ask for the backward, so the per-pair state is not written -- unless a hook
        # override wants the per-split tensors (analysis/utils.py:67-95)
        params = self._param_tensors()
        needs_grad = torch.is_grad_enabled() and (x_span.requires_grad or any(t is not None and t.requires_grad for t in params))
        hooks = self._hook_overridden('inside_hook') or self._hook_overridden('outside_hook')
        flags = int(bool(self.outside)) | (0 if needs_grad else _lib.FWD_NO_BACKWARD) | (_lib.FWD_PAIR_STATES if hooks else 0)
        ih, is_, oh, os_ = ChartFunction.apply(plan, holder, flags, x_span, *params)
        ch = Chart()
        ch.inside_h, ch.inside_s, ch.outside_h, ch.outside_s = ih, is_, oh, os_       # the zero cell states: on first read (Chart)
        d = self.__dict__
        d['chart'], d['_wss'], d['_plan'] = ch, holder, plan
        if self._hook_overridden('init_with_batch'):      # analysis/utils.py:67-76 wraps it: then it gets the reference's arguments
            self.init_with_batch(ih[:, :L], ch.inside_c[:, :L])
        else:
            d['batch_size'], d['length'] = B, L
        if hooks:
            self._serve_hooks(L)
        return None

    # ---- un-aggregated per-split tensors the hooks receive (diora.py:295-334)
    def pair_states(self, level):
        """(h, s) of one inside level: h (B*Lc*N, D) compose outputs, s (B, Lc, N, 1) split scores."""
        plan = self._plan
        hs, ss = [], []
        for ws in self._wss:
            ps, ph, rows, ldh = C.c_void_p(), C.c_void_p(), C.c_size_t(), C.c_size_t()
            _lib.check(_lib.lib().cliora_inside_pair_states(plan.handle, _ptr(ws), level, C.byref(ps), C.byref(ph),
                                                            C.byref(rows), C.byref(ldh)), 'cliora_inside_pair_states')
            wsf = ws.view(torch.float32)
            so, ho = (ps.value - ws.data_ptr()) // 4, (ph.value - ws.data_ptr()) // 4
            n, ld = rows.value, ldh.value
            ss.append(wsf[so:so + n].view(plan.B, plan.L - level, level, 1))
            hs.append(wsf[ho:ho + n * ld].view(n, ld)[:, :plan.D])
        if len(hs) == 1:
            return hs[0], ss[0]
        return torch.cat(hs, 0), torch.cat(ss, 0)

    def pair_states_out(self, level):
        """(h, s) of one outside level in the REFERENCE's order (diora.py:364-398): h (B*N*Lc, D) compose outputs,
        s (B, N, Lc, 1) split scores, N = L-level-1 (parent, sibling) splits of the Lc = L-level target cells."""
        plan = self._plan
        L = plan.L
        Lc, N = L - level, L - level - 1
        ii = torch.arange(N).view(N, 1)
        jj = torch.arange(Lc).view(1, Lc).expand(N, Lc)
        nn_ = torch.where(jj < N - ii, L - 2 - ii - level, N - ii - 1)          # our split of the reference's (i, j)
        hs, ss = [], []
        for ws in self._wss:
            ps, ph, rows, ldh = C.c_void_p(), C.c_void_p(), C.c_size_t(), C.c_size_t()
            _lib.check(_lib.lib().cliora_outside_pair_states(plan.handle, _ptr(ws), level, C.byref(ps), C.byref(ph),
                                                             C.byref(rows), C.byref(ldh)), 'cliora_outside_pair_states')
            wsf = ws.view(torch.float32)
            so, ho = (ps.value - ws.data_ptr()) // 4, (ph.value - ws.data_ptr()) // 4
            n, ld = rows.value, ldh.value
            jd, nd = jj.to(ws.device), nn_.to(ws.device)
            s_ours = wsf[so:so + n].view(plan.B, Lc, N)
            h_ours = wsf[ho:ho + n * ld].view(plan.B, Lc, N, ld)
            ss.append(s_ours[:, jd, nd].unsqueeze(-1))                           # (B, N, Lc, 1)
            hs.append(h_ours[:, jd, nd][..., :plan.D].reshape(plan.B * N * Lc, plan.D))
        if len(hs) == 1:
            return hs[0], ss[0]
        return torch.cat(hs, 0), torch.cat(ss, 0)

    def _serve_hooks(self, L):
        """inside_hook / outside_hook overrides get the per-split states the reference passes them (diora.py:331, 398),
        after the native passes have run."""
        if self._hook_overridden('inside_hook'):
            for level in range(1, L):
                h, s = self.pair_states(level)
                self.inside_hook(level, h, torch.zeros_like(h), s)
        if self.outside and self._hook_overridden('outside_hook'):
            for level in range(L - 2, -1, -1):
                h, s = self.pair_states_out(level)
                self.outside_hook(level, h, torch.zeros_like(h), s)

    def cky_spans(self):
        """Constituent spans of the best binary tree per sentence (analysis/cky.py:31-99 decoded on the GPU, cliora_cky_spans): an int32
        array (B, L-1, 2) of (start, end) word positions, children before parents, the root last -- what the reference derives on the
        host with ``get_spans(get_actions(tree))`` (analysis/utils.py:3-49) and scores F1 on (scripts/train.py:184-204).  One
        device-to-host copy, no recursion."""
        plan = self._plan
        parts = []
        if plan.L < 2:                                   # one word: no constituent
            import numpy as np
            return np.zeros((plan.B * len(self._wss), 0, 2), dtype=np.int32)
        for ws in self._wss:
            with torch.cuda.device(ws.device):
                spans = torch.empty((plan.B, plan.L - 1, 2), device=ws.device, dtype=torch.int32)
                _lib.check(_lib.lib().cliora_cky_spans(plan.handle, _ptr(ws), None, _ptr(spans), _stream()), 'cliora_cky_spans')
            parts.append(spans)
        return (parts[0] if len(parts) == 1 else torch.cat(parts, 0)).cpu().numpy()

    @staticmethod
    def trees_from_spans(spans):
        """Nested tuples of word positions (what ``ParsePredictor.parse_batch`` returns, e.g. ``(0, ((1, 2), 3))``) from the
        children-before-parents span lists of cky_spans: a stack per sentence, no recursion."""
        trees = []
        for sent in spans.tolist():
            if not sent:
                trees.append(0)
                continue
            stack = []                                   # (start, end, tree) of the finished constituents
            for s, e in sent:
                if stack and stack[-1][1] == e and stack[-1][0] > s:
                    rs, _, right = stack.pop()
                else:
                    rs, right = e, e                     # the right child is the word e itself
                if stack and stack[-1][0] == s and stack[-1][1] == rs - 1:
                    left = stack.pop()[2]
                else:
                    left = s
                stack.append((s, e, (left, right)))
            trees.append(stack[-1][2])
        return trees

    def cky(self):
        """Best binary tree per sentence (analysis/cky.py:31-99) decoded on the GPU.

        Returns nested tuples of word positions, e.g. ``(0, ((1, 2), 3))`` -- a view of the device-built span lists (cky_spans)."""
        return self.trees_from_spans(self.cky_spans())


class DioraMLP(DioraBase):
    def init_parameters(self):
        self.inside_score_func = Bilinear(self.size)
        self.inside_compose_func = ComposeMLP(self.size, leaf=True)
        if self.share:
            self.outside_score_func = self.inside_score_func
            self.outside_compose_func = self.inside_compose_func
        else:
            self.outside_score_func = Bilinear(self.size)
            self.outside_compose_func = ComposeMLP(self.size)
        if self.compress:
            self.root_mat_out = nn.Parameter(torch.empty(self.size, self.size))
        else:
            self.root_vector_out_h = nn.Parameter(torch.empty(self.size))
        self.root_vector_out_c = None
