"""CPU tests of the native host logic: the C-ABI library loads, exports every symbol the
header declares, and its chart tables agree with the tables captured from the reference."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from cliora_amd import _lib
from cliora_amd.index import Index
from oracle import chart_layout as CL


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, 'include', 'cliora_chart.h')).read()
    names = set(re.findall(r'\b(cliora_[a-z_]+)\s*\(', hdr))
    assert len(names) >= 12
    L = ctypes.CDLL(_lib.LIB_PATH)
    for n in sorted(names):
        assert hasattr(L, n), n
    assert b'gfx950' in _lib.lib().cliora_version()


def test_parameter_struct_is_the_headers_in_every_binding():
    """struct cliora_params: the ctypes mirror (cliora_amd/_lib.py) and the stub a maintainer would copy list the
    header's fields in the header's order -- a shorter struct would let the library read past it (root_mat selects compress)."""
    hdr = open(os.path.join(ROOT, 'include', 'cliora_chart.h')).read()
    body = re.search(r'typedef struct cliora_params \{(.*?)\} cliora_params;', hdr, re.S).group(1)
    body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
    fields = re.findall(r'\*\s*([a-z_0-9]+)\s*[,;]', body)
    assert tuple(fields) == tuple(_lib.PARAM_FIELDS)
    assert ctypes.sizeof(_lib.Params) == len(fields) * ctypes.sizeof(ctypes.c_void_p)
    doc = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    stub = re.search(r'class Params\(C\.Structure\):.*?_fields_ = \[\(n, C\.c_void_p\) for n in \((.*?)\)\]', doc, re.S).group(1)
    assert tuple(re.findall(r"'([a-z_0-9]+)'", stub)) == tuple(fields)


@pytest.mark.parametrize('L', [2, 3, 4, 7, 10, 20, 33, 40])
def test_tables_match_reference(golden, L):
    g = golden('index_tables.npz')
    ix = Index()
    off = ix.get_offset(L)
    assert [off[i] for i in range(L)] == g['off_%d' % L].tolist()
    li = np.concatenate([ix.get_inside_index(L, lv)[0].numpy() for lv in range(1, L)])
    ri = np.concatenate([ix.get_inside_index(L, lv)[1].numpy() for lv in range(1, L)])
    assert np.array_equal(li, g['lidx_%d' % L]) and np.array_equal(ri, g['ridx_%d' % L])
    pi = np.concatenate([ix.get_outside_index(L, lv)[0].numpy() for lv in range(L - 2, -1, -1)])
    si = np.concatenate([ix.get_outside_index(L, lv)[1].numpy() for lv in range(L - 2, -1, -1)])
    assert np.array_equal(pi, g['pidx_%d' % L]) and np.array_equal(si, g['sidx_%d' % L])


@pytest.mark.parametrize('B,L', [(1, 1), (2, 2), (3, 6), (2, 11)])
def test_use_lists_and_row_maps(B, L):
    pl = _lib.Plan(B, L, 20)
    C = L * (L + 1) // 2
    P_in, P_out = CL.n_inside_pairs(L), CL.n_outside_pairs(L)
    arow, brow, trow = pl.table('arow'), pl.table('brow'), pl.table('trow')
    assert len(arow) == B * (P_in + P_out)
    # every pair row appears exactly once in the a-role and once in the b-role use lists of its pass
    seen = {r: np.zeros(len(arow), dtype=np.int32) for r in ('ina', 'inb', 'outa', 'outb')}
    for role in seen:
        uo, ur, us, up = (pl.table('use_%s_%s' % (k, role)) for k in ('off', 'row', 'stride', 'partner'))
        assert len(uo) == C + 1
        for c in range(C):
            for u in range(uo[c], uo[c + 1]):
                for b in range(B):
                    row = ur[u] + b * us[u]
                    seen[role][row] += 1
                    mine, other = (arow, brow) if role in ('ina', 'outa') else (brow, arow)
                    assert mine[row] == b * C + c and other[row] == b * C + up[u]
    nin = B * P_in
    assert (seen['ina'][:nin] == 1).all() and (seen['inb'][:nin] == 1).all()
    assert (seen['outa'][nin:] == 1).all() and (seen['outb'][nin:] == 1).all()
    assert seen['ina'][nin:].sum() == 0 and seen['outa'][:nin].sum() == 0
    # a target's children tile its span: inside  a=[p..], b=[..end];  outside  sibling + target = parent
    off = pl.table('level_offset')

    def span(cell):
        lv = int(np.searchsorted(off, cell, side='right') - 1)
        pos = cell - off[lv]
        return pos, pos + lv
    for r in range(len(arow)):
        ta, tb, tt = span(arow[r] % C), span(brow[r] % C), span(trow[r] % C)
        if r < nin:
            assert ta[0] == tt[0] and tb[1] == tt[1] and ta[1] + 1 == tb[0]
        else:   # a = sibling (inside chart), b = parent (outside chart)
            assert tb[0] == min(ta[0], tt[0]) and tb[1] == max(ta[1], tt[1])
            assert (ta[1] + 1 == tt[0]) or (tt[1] + 1 == ta[0])


def test_plan_rejects_bad_shapes():
    for args in [(0, 5, 16), (2, 5, 600), (2, 70, 16)]:
        with pytest.raises(_lib.ChartLibError):
            _lib.Plan(*args)


def test_workspace_sizes_scale():
    a, b = _lib.Plan(8, 10, 50), _lib.Plan(64, 20, 400)
    assert 0 < a.fwd_bytes < b.fwd_bytes and 0 < a.bwd_bytes < b.bwd_bytes
    assert b.fwd_bytes < 2 << 30 and b.bwd_bytes < 2 << 30


def test_entry_points_reject_null_and_mismatched_arguments():
    """Error behaviour of the C ABI without a GPU: argument checks come before any HIP call, return a negative
    CLIORA_E* code (never throw / crash) and leave a message in cliora_last_error()."""
    import ctypes as C
    L = _lib.lib()
    p = _lib.Plan(2, 5, 16)
    one = C.c_void_p(16)                     # a non-null dummy pointer: never dereferenced before the checks fail
    prm = _lib.Params()
    # forward: missing plan / params / inputs / outputs / workspace
    L.cliora_chart_forward.restype = C.c_int
    rc = L.cliora_chart_forward(None, C.byref(prm), one, None, None, one, one, one, one, None, one, C.c_size_t(1 << 20), 1, None)
    assert rc < 0 and L.cliora_last_error()
    rc = L.cliora_chart_forward(p.handle, C.byref(prm), None, None, None, one, one, one, one, None, one, C.c_size_t(1 << 20), 1, None)
    assert rc < 0
    # a text-only plan (R = 0) must not be given region features
    rc = L.cliora_chart_forward(p.handle, C.byref(prm), one, one, None, one, one, one, one, None, one, C.c_size_t(p.fwd_bytes), 1, None)
    assert rc < 0 and b'obj_span' in L.cliora_last_error()
    # a workspace smaller than cliora_plan_fwd_workspace_bytes()
    rc = L.cliora_chart_forward(p.handle, C.byref(prm), one, None, None, one, one, one, one, None, one, C.c_size_t(16), 1, None)
    assert rc < 0 and b'workspace' in L.cliora_last_error().lower()
    # a TreeLSTM plan on the MLP entry point and the other way round
    q = _lib.Plan(2, 5, 16, arch=1)
    rc = L.cliora_chart_forward(q.handle, C.byref(prm), one, None, None, one, one, one, one, None, one, C.c_size_t(q.fwd_bytes), 1, None)
    assert rc < 0 and b'TreeLSTM' in L.cliora_last_error()
    # unknown table name
    n = C.c_size_t()
    ptr = C.POINTER(C.c_int32)()
    assert L.cliora_plan_table(p.handle, b'no_such_table', C.byref(ptr), C.byref(n)) < 0


def test_mfma_mode_switch_round_trips():
    prev = _lib.set_mfma_mode('f32')
    assert _lib.set_mfma_mode('bf16x3') == 'f32'
    assert _lib.set_mfma_mode(prev) == 'bf16x3'


def test_pair_row_tiles_follow_the_pair_rows_level_by_level():
    """The tiled split-bf16 operands of the pair rows' weight gradient (csrc/wgrad_tiles.hpp) are stored per 16-row tile = 16 target cells of
    one level x one split, levels in the pair rows' order.  tile_base_{in,out}[lv] is the first tile of level lv of a pass, entry L its end:
    N(lv) * ceil(B * Lc / 16) tiles per level, and exactly row_base / 16 where every level's cell count is a multiple of 16."""
    from cliora_amd import _lib
    for B, L in ((64, 20), (3, 9), (17, 5), (1, 1), (16, 2)):
        pl = _lib.Plan(B, L, 400, True, 'unit', 0, 0)
        tin, tout = pl.table('tile_base_in'), pl.table('tile_base_out')
        assert len(tin) == L + 1 and len(tout) == L + 1 and tin[0] == 0 and tout[0] == 0
        for lv in range(L):
            g16 = (B * (L - lv) + 15) // 16
            assert tin[lv + 1] - tin[lv] == g16 * lv, (B, L, lv)
            assert tout[lv + 1] - tout[lv] == g16 * (L - lv - 1), (B, L, lv)
        if B % 16 == 0:
            bin_, bout = pl.table('pair_lvl_base_in'), pl.table('pair_lvl_base_out')
            for lv in range(L):
                if lv >= 1:                 # levels with pairs: inside 1 .. L-1, outside 0 .. L-2
                    assert tin[lv] * 16 == B * int(bin_[lv]), (B, L, lv)
                if lv <= L - 2:
                    assert tout[lv] * 16 == B * int(bout[lv]), (B, L, lv)
