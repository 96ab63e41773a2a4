"""Builds the C-ABI shared library (HIP, gfx950) in-tree: cliora_amd/libcliora_chart.so.

hipcc cross-compiles without a GPU.  Called by __graft_entry__.build(); also usable
as ``python -m cliora_amd.build``.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libcliora_chart.so')
SOURCES = ['chart_api.hip', 'plan.cpp']
HEADERS = ['chart_kernels.hpp', 'gemm_kernels.hpp', 'plan.hpp', os.path.join('..', '..', 'include', 'cliora_chart.h')]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=True):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-x', 'hip',
           '-Wall', '-Wno-unused-function', '-Wno-pass-failed', '-Wno-unused-value', '-o', LIB] + [os.path.join(CSRC, f) for f in SOURCES]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
