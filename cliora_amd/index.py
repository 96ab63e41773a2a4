"""Host-side chart index helper with the surface of the reference's ``Index``
(cliora/net/utils.py:67-134): ``get_offset(length)[level]``,
``get_inside_index(length, level)``, ``get_outside_index(length, level)``.

The tables come from the native plan (cliora_amd/csrc/plan.cpp), i.e. the same
ones the HIP kernels use; the outside table is re-ordered to the reference's
n-major enumeration (cliora/net/outside_index.py:39-62) for callers that index
with it.
"""
import numpy as np
import torch

from . import _lib


class Index(object):
    def __init__(self, cuda=False, enable_caching=True):
        self.cuda = cuda
        self._plans = {}
        self._cache = {}

    def _plan(self, length):
        if length not in self._plans:
            self._plans[length] = _lib.Plan(1, length, 16)
        return self._plans[length]

    def _dev(self, a):
        t = torch.from_numpy(np.ascontiguousarray(a).astype(np.int64))
        return t.cuda() if self.cuda else t

    def get_offset(self, length):
        key = ('off', length)
        if key not in self._cache:
            off = self._plan(length).table('level_offset')
            self._cache[key] = {lv: int(off[lv]) for lv in range(length)}
        return self._cache[key]

    def get_inside_index(self, length, level):
        key = ('in', length, level)
        if key not in self._cache:
            pl = self._plan(length)
            base = int(pl.table('pair_lvl_base_in')[level])
            n = (length - level) * level
            self._cache[key] = (self._dev(pl.table('pair_a_in')[base:base + n]),
                                self._dev(pl.table('pair_b_in')[base:base + n]))
        return self._cache[key]

    def get_outside_index(self, length, level):
        """(parent, sibling) in the reference's order: flat = i*Lc + j, see outside_index.py:39-62."""
        key = ('out', length, level)
        if key not in self._cache:
            pl = self._plan(length)
            Lc, N = length - level, length - level - 1
            base = int(pl.table('pair_lvl_base_out')[level])
            sib = pl.table('pair_a_out')[base:base + Lc * N].reshape(Lc, N)
            par = pl.table('pair_b_out')[base:base + Lc * N].reshape(Lc, N)
            # ours: target pos j, split n -- n < j: parent starts at q = n (target is the right child);
            # n >= j: parent ends at r = level + 1 + n.  Reference split i of target j: if j < N - i the
            # parent ends at length-1-i (-> n = length - 2 - i - level), else it starts at N-i-1 (-> n = N-i-1).
            pi = np.empty((N, Lc), dtype=np.int64)
            si = np.empty((N, Lc), dtype=np.int64)
            for i in range(N):
                for j in range(Lc):
                    n = (length - 2 - i - level) if j < N - i else (N - i - 1)
                    pi[i, j] = par[j, n]
                    si[i, j] = sib[j, n]
            self._cache[key] = (self._dev(pi.reshape(-1)), self._dev(si.reshape(-1)))
        return self._cache[key]
