"""Chart cell layout and span-pair tables (integer work, numpy).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates, in closed form, what the reference builds with Python loops:
  * level offsets            cliora/net/offset_cache.py:1-7
  * inside (left, right)     cliora/net/inside_index.py:131-197
  * outside (parent, sibling) cliora/net/outside_index.py:39-62, 93-127

A cell is (level, pos): the span of words [pos, pos+level].  Cells of one level
are contiguous: cell id = offset(level) + pos, leaves are ids [0, L), the root
is id C-1 with C = L(L+1)/2.
"""
import numpy as np


def n_cells(L):
    return L * (L + 1) // 2


def level_offset(L, level):
    """offset_cache.py:1-7  ->  C - (L-level)(L-level+1)/2."""
    rem = L - level
    return n_cells(L) - rem * (rem + 1) // 2


def level_offsets(L):
    return np.array([level_offset(L, lv) for lv in range(L)], dtype=np.int64)


def cell_id(L, level, pos):
    return level_offset(L, level) + pos


def inside_pairs(L, level):
    """(lidx, ridx), each (Lc*N,), flat index p*N + n  (pos-major).

    inside_index.py:131-197: split n of target (level, p) has
    left = (n, p), right = (level-n-1, p+n+1); the reference emits
    [n][p] lists and transposes them (:192-195).
    """
    Lc, N = L - level, level
    p = np.repeat(np.arange(Lc), N)
    n = np.tile(np.arange(N), Lc)
    off = level_offsets(L)
    lidx = off[n] + p
    ridx = off[level - n - 1] + p + n + 1
    return lidx.astype(np.int64), ridx.astype(np.int64)


def outside_pairs(L, level):
    """(pidx, sidx), each (N*Lc,), flat index i*Lc + j  (n-major).

    outside_index.py:39-62 enumerates, for i in [0,N) and target pos j in
    [0,Lc): if j < N-i the target is the LEFT child of the parent span
    [j, L-1-i] (sibling = [j+level+1, L-1-i]); otherwise it is the RIGHT child
    of the parent span [N-i-1, j+level] (sibling = [N-i-1, j-1]).  The tuple
    fields there are named inconsistently (:59, :104-107); the emitted integers
    are what is reproduced here.  Parent ids index the OUTSIDE chart, sibling
    ids the INSIDE chart.
    """
    Lc = L - level
    N = Lc - 1
    off = level_offsets(L)
    i = np.repeat(np.arange(N), Lc)
    j = np.tile(np.arange(Lc), N)
    left_child = j < (N - i)
    # target is left child
    par_start_a, par_end_a = j, L - 1 - i
    sib_start_a, sib_end_a = j + level + 1, L - 1 - i
    # target is right child
    par_start_b, par_end_b = N - i - 1, j + level
    sib_start_b, sib_end_b = N - i - 1, j - 1
    ps = np.where(left_child, par_start_a, par_start_b)
    pe = np.where(left_child, par_end_a, par_end_b)
    ss = np.where(left_child, sib_start_a, sib_start_b)
    se = np.where(left_child, sib_end_a, sib_end_b)
    pidx = off[pe - ps] + ps
    sidx = off[se - ss] + ss
    return pidx.astype(np.int64), sidx.astype(np.int64)


def n_inside_pairs(L):
    return (L - 1) * L * (L + 1) // 6


def n_outside_pairs(L):
    return (L - 1) * L * (L + 1) // 3
