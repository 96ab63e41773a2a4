"""ctypes binding of include/cliora_chart.h (the C-ABI shared library).

The product path has NO CPU fallback: if the HIP library is missing or a call
fails, this raises.  Build it with ``python -m cliora_amd.build`` (or
``__graft_entry__.build()``).
"""
import collections
import ctypes as C
import functools
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('CLIORA_CHART_LIB') or os.path.join(HERE, 'libcliora_chart.so')   # override: kernel experiments only

NORM = {'none': 0, 'unit': 1}
FWD_NO_BACKWARD = 2      # include/cliora_chart.h: flag bits of cliora_chart_forward's run_outside word
FWD_PAIR_STATES = 4
KCLASS = {'compose_fwd': 0, 'compose_bwd': 1, 'wgrad': 2}

PARAM_FIELDS = ('leaf_w', 'leaf_b', 'in_w1', 'in_b1', 'in_w2', 'in_b2', 'in_mat',
                'out_w1', 'out_b1', 'out_w2', 'out_b2', 'out_mat', 'root_h',
                'lstm_w', 'lstm_u', 'lstm_b', 'root_c', 'lstm_u_out', 'lstm_b_out', 'root_mat')


class Params(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in PARAM_FIELDS]


class ChartLibError(RuntimeError):
    pass


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ChartLibError('HIP extension %s is missing: run `python -m cliora_amd.build` '
                            '(there is no CPU fallback for the chart path)' % LIB_PATH)
    # torch first: its bundled HIP runtime must be the one already loaded when our library is dlopen'ed, otherwise the
    # process ends up with two runtimes and ours sees no device ("no ROCm-capable device is detected" at the first hipMalloc)
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    vp, i32, sz = C.c_void_p, C.c_int, C.c_size_t
    L.cliora_plan_create.argtypes = [i32] * 6 + [C.POINTER(vp)]
    L.cliora_plan_create.restype = i32
    L.cliora_plan_create_ex.argtypes = [i32] * 7 + [C.POINTER(vp)]
    L.cliora_plan_create_ex.restype = i32
    L.cliora_lstm_forward.argtypes = [vp, C.POINTER(Params)] + [vp] * 7 + [vp, sz, i32, vp]
    L.cliora_lstm_forward.restype = i32
    L.cliora_lstm_backward.argtypes = [vp, C.POINTER(Params)] + [vp] * 13 + [vp, sz, vp, sz, vp, C.POINTER(Params), i32, vp]
    L.cliora_lstm_backward.restype = i32
    L.cliora_plan_destroy.argtypes = [vp]
    L.cliora_plan_destroy.restype = None
    L.cliora_plan_fwd_workspace_bytes.argtypes = [vp]
    L.cliora_plan_fwd_workspace_bytes.restype = sz
    L.cliora_plan_bwd_workspace_bytes.argtypes = [vp]
    L.cliora_plan_bwd_workspace_bytes.restype = sz
    L.cliora_plan_pair_states_bytes.argtypes = [vp]
    L.cliora_plan_pair_states_bytes.restype = sz
    L.cliora_plan_device_bytes.argtypes = [vp]
    L.cliora_plan_device_bytes.restype = sz
    L.cliora_plan_fwd_offset.argtypes = [vp, C.c_char_p]
    L.cliora_plan_fwd_offset.restype = sz
    L.cliora_plan_table.argtypes = [vp, C.c_char_p, C.POINTER(C.POINTER(C.c_int32)), C.POINTER(sz)]
    L.cliora_plan_table.restype = i32
    L.cliora_chart_forward.argtypes = [vp, C.POINTER(Params)] + [vp] * 8 + [vp, sz, i32, vp]
    L.cliora_chart_forward.restype = i32
    L.cliora_chart_backward.argtypes = [vp, C.POINTER(Params)] + [vp] * 11 + [vp, sz, vp, sz, vp, vp, C.POINTER(Params), i32, vp]
    L.cliora_chart_backward.restype = i32
    L.cliora_inside_pair_states.argtypes = [vp, vp, i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(sz), C.POINTER(sz)]
    L.cliora_inside_pair_states.restype = i32
    L.cliora_outside_pair_states.argtypes = [vp, vp, i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(sz), C.POINTER(sz)]
    L.cliora_outside_pair_states.restype = i32
    L.cliora_plan_vl_workspace_bytes.argtypes = [vp]
    L.cliora_plan_vl_workspace_bytes.restype = sz
    L.cliora_vl_scores_forward.argtypes = [vp] + [vp] * 5 + [i32, vp, vp, vp, sz, vp]
    L.cliora_vl_scores_forward.restype = i32
    L.cliora_vl_scores_backward.argtypes = [vp] + [vp] * 5 + [i32] + [vp] * 6 + [vp, sz, vp]
    L.cliora_vl_scores_backward.restype = i32
    L.cliora_vl_scores_max_forward.argtypes = [vp] + [vp] * 3 + [vp, vp, vp, sz, vp]
    L.cliora_vl_scores_max_forward.restype = i32
    L.cliora_vl_scores_max_backward.argtypes = [vp] + [vp] * 3 + [vp, vp, vp, vp, vp, sz, vp]
    L.cliora_vl_scores_max_backward.restype = i32
    L.cliora_contrastive_workspace_bytes.argtypes = [i32, i32]
    L.cliora_contrastive_workspace_bytes.restype = sz
    L.cliora_contrastive_loss.argtypes = [i32, i32, vp, vp, vp, C.c_float, C.c_float, vp, vp, vp, vp, vp, sz, vp]
    L.cliora_contrastive_loss.restype = i32
    L.cliora_proj_workspace_bytes.argtypes = [i32, i32, i32]
    L.cliora_proj_workspace_bytes.restype = sz
    L.cliora_proj_forward.argtypes = [vp, vp, i32, i32, vp, vp, i32, vp, vp, sz, vp]
    L.cliora_proj_forward.restype = i32
    L.cliora_proj_backward.argtypes = [vp, vp, i32, i32, vp, vp, i32, vp, vp, vp, vp, sz, vp]
    L.cliora_proj_backward.restype = i32
    L.cliora_recon_workspace_bytes.argtypes = [i32, i32, i32, i32]
    L.cliora_recon_workspace_bytes.restype = sz
    L.cliora_recon_forward.argtypes = [vp, vp, i32, i32, i32, i32, vp, i32, vp, i32, vp, vp, vp, sz, vp]
    L.cliora_recon_forward.restype = i32
    L.cliora_recon_backward.argtypes = [vp, vp, i32, i32, i32, i32, vp, i32, vp, i32, vp, vp, vp, vp, vp, vp, sz, vp]
    L.cliora_recon_backward.restype = i32
    L.cliora_rows_scatter_add.argtypes = [vp, vp, i32, i32, vp, C.c_int64, vp]     # mandatory: heads.scatter_rows has no other path
    L.cliora_rows_scatter_add.restype = i32
    if hasattr(L, 'cliora_rows_scatter_add_segments'):          # (absent from older builds loaded through CLIORA_CHART_LIB for A/B runs)
        L.cliora_rows_scatter_add_segments.argtypes = [C.POINTER(vp), C.POINTER(vp), C.POINTER(i32), i32, i32, vp, C.c_int64, vp]
        L.cliora_rows_scatter_add_segments.restype = i32
    L.cliora_vg_workspace_bytes.argtypes = [i32, i32]
    L.cliora_vg_workspace_bytes.restype = sz
    L.cliora_vg_loss.argtypes = [i32, i32, i32, vp, C.c_float, vp, vp, vp, sz, vp]
    L.cliora_vg_loss.restype = i32
    L.cliora_clip_adam_workspace_bytes.argtypes = []
    L.cliora_clip_adam_workspace_bytes.restype = sz
    L.cliora_clip_adam.argtypes = [vp, vp, vp, vp, sz, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, i32, vp, sz, vp]
    L.cliora_clip_adam.restype = i32
    L.cliora_cky_decode.argtypes = [vp, vp, vp, vp]
    L.cliora_cky_decode.restype = i32
    L.cliora_cky_spans.argtypes = [vp, vp, vp, vp, vp]
    L.cliora_cky_spans.restype = i32
    L.cliora_prof_enable.argtypes = [i32, i32]
    L.cliora_prof_enable.restype = i32
    L.cliora_prof_read.argtypes = [i32, C.POINTER(C.c_double), C.POINTER(C.c_longlong), vp]
    L.cliora_prof_read.restype = i32
    L.cliora_last_error.restype = C.c_char_p
    L.cliora_version.restype = C.c_char_p
    L.cliora_set_mfma_mode.argtypes = [i32]
    L.cliora_set_mfma_mode.restype = i32
    L.cliora_set_wavefront.argtypes = [i32]
    L.cliora_set_wavefront.restype = i32
    L.cliora_set_resident.argtypes = [i32]
    L.cliora_set_resident.restype = i32
    if hasattr(L, 'cliora_device_side_stream'):
        L.cliora_device_side_stream.argtypes = [vp, C.POINTER(vp)]
        L.cliora_device_side_stream.restype = i32
    if hasattr(L, 'cliora_resident_trace'):
        L.cliora_resident_trace.argtypes = [vp, vp, sz, vp]
        L.cliora_resident_trace.restype = i32
    _lib = L
    return L


_side_streams = {}


def side_stream(device):
    """torch handle of the library's caller lane on `device` (include/cliora_chart.h: cliora_device_side_stream): a stream that runs beside
    the current stream and the chart's own side streams.  One per device, created at the first request."""
    import torch
    idx = device.index if device.index is not None else torch.cuda.current_device()
    s = _side_streams.get(idx)
    if s is None:
        out = C.c_void_p()
        with torch.cuda.device(idx):
            check(lib().cliora_device_side_stream(C.c_void_p(torch.cuda.current_stream().cuda_stream), C.byref(out)), 'cliora_device_side_stream')
        s = torch.cuda.ExternalStream(out.value, device=torch.device('cuda', idx))
        _side_streams[idx] = s
    return s


def check(rc, what):
    if rc != 0:
        raise ChartLibError('%s failed (%d): %s' % (what, rc, lib().cliora_last_error().decode()))


class Plan:
    """Owns one cliora_plan (chart shape + device index tables)."""

    def __init__(self, B, L, D, share=True, normalize='unit', R=0, arch=0):
        self.key = (B, L, D, bool(share), normalize, R, arch)
        self.arch = arch
        self.B, self.L, self.D, self.share, self.normalize, self.R = B, L, D, bool(share), normalize, R
        self.C = L * (L + 1) // 2
        self.Dp = (D + 15) // 16 * 16
        h = C.c_void_p()
        check(lib().cliora_plan_create_ex(B, L, D, int(bool(share)), NORM[normalize], R, arch, C.byref(h)), 'cliora_plan_create_ex')
        self.handle = h
        self.fwd_bytes = lib().cliora_plan_fwd_workspace_bytes(h)
        self.bwd_bytes = lib().cliora_plan_bwd_workspace_bytes(h)
        self.pair_bytes = lib().cliora_plan_pair_states_bytes(h)    # optional workspace tail: per-pair compose outputs for the hooks
        self.table_bytes = lib().cliora_plan_device_bytes(h)     # index tables: this much host memory, and as much HBM after the first forward

    def table(self, name):
        import numpy as np
        ptr = C.POINTER(C.c_int32)()
        n = C.c_size_t()
        check(lib().cliora_plan_table(self.handle, name.encode(), C.byref(ptr), C.byref(n)), 'cliora_plan_table')
        if n.value == 0:
            return np.zeros(0, dtype=np.int32)
        return np.ctypeslib.as_array(ptr, shape=(n.value,)).copy()

    def __del__(self):
        try:
            if getattr(self, 'handle', None):
                lib().cliora_plan_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


# Plan cache: least-recently-used, bounded by the bytes of index tables the cached plans hold on the device AND by their number
# (the host-side use lists of a plan are several times its device tables).  Length-bucketed batches give one (B, L) key per bucket:
# a DioraMLP plan holds 0.1 MB (L 20) to 1 MB (L 40) of device tables, a TreeLSTM plan at B 64 / L 40 about 25 MB (its
# batch-expanded row maps).
PLAN_CACHE_BYTES = int(os.environ.get('CLIORA_PLAN_CACHE_MB', '512')) << 20
PLAN_CACHE_MAX = int(os.environ.get('CLIORA_PLAN_CACHE_MAX', '128'))
_plans = collections.OrderedDict()


def get_plan(B, L, D, share, normalize, R, device_index, arch=0):
    key = (B, L, D, bool(share), normalize, R, device_index, arch)
    pl = _plans.get(key)
    if pl is not None:
        _plans.move_to_end(key)
        return pl
    pl = _plans[key] = Plan(B, L, D, share, normalize, R, arch)
    total = sum(q.table_bytes for q in _plans.values())
    while (total > PLAN_CACHE_BYTES or len(_plans) > PLAN_CACHE_MAX) and len(_plans) > 1:
        _, old = _plans.popitem(last=False)          # evicted plans free their device tables when the last user drops them
        total -= old.table_bytes
    return pl


def on_device(pick):
    """Decorator for the autograd Function bodies: run under the device of the tensor `pick(*args)` returns, so that the
    plan's index tables, the kernels' per-device attributes and every allocation land on the tensors' GPU even when it is
    not the current one."""
    def deco(fn):
        @functools.wraps(fn)
        def run(*args):
            import torch
            t = pick(*args)
            if t is None or not t.is_cuda:
                return fn(*args)
            with torch.cuda.device(t.device):
                return fn(*args)
        return run
    return deco


def prof_enable(kclass, on=True):
    check(lib().cliora_prof_enable(KCLASS[kclass], int(on)), 'cliora_prof_enable')


def prof_read(kclass, stream=0):
    ms, n = C.c_double(), C.c_longlong()
    check(lib().cliora_prof_read(KCLASS[kclass], C.byref(ms), C.byref(n), C.c_void_p(stream)), 'cliora_prof_read')
    return ms.value, n.value


MFMA_MODES = {'f32': 0, 'bf16x3': 1}


WAVEFRONT_MODES = {'auto': -1, 'off': 0, 'on': 1}


def set_wavefront(mode):
    """Scheduling of the inside / outside passes (include/cliora_chart.h: cliora_set_wavefront): 'auto' (default), 'off' (the
    reference's order on the caller's stream alone), 'on' (two streams) or 'merged' (the two passes' launches of a step as one grid on one queue).  Results are bitwise identical.  Returns the previous mode."""
    prev = lib().cliora_set_wavefront({'merged': 2}.get(mode, WAVEFRONT_MODES.get(mode)))
    return {-1: 'auto', 0: 'off', 1: 'on', 2: 'merged'}[prev]


def set_resident(mode):
    """One workgroup per sentence for the level loops of a small-D DioraMLP plan (include/cliora_chart.h: cliora_set_resident):
    'auto' (default), 'off', 'on'.  Returns the previous mode's name."""
    prev = lib().cliora_set_resident(WAVEFRONT_MODES[mode])
    return {v: k for k, v in WAVEFRONT_MODES.items()}[prev]


def set_mfma_mode(mode):
    """Arithmetic of the compose-layer GEMMs and their weight gradient: 'bf16x3' (default; three bf16 MFMAs per
    product, fp32 accumulate) or 'f32' (fp32-input MFMA, the reference's arithmetic).  Returns the previous mode."""
    prev = lib().cliora_set_mfma_mode(MFMA_MODES[mode])
    return 'f32' if prev == 0 else 'bf16x3'
