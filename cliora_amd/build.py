"""Builds the C-ABI shared library (HIP, gfx950) in-tree: cliora_amd/libcliora_chart.so.

hipcc cross-compiles without a GPU.  Every translation unit under csrc/ (*.hip, *.cpp) is compiled to an object in
csrc/build/ -- in parallel, each only when it or any header is newer than its object -- and the objects are linked into
the library.  Called by __graft_entry__.build(); also usable as ``python -m cliora_amd.build [--force]``.
"""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(CSRC, 'build')
LIB = os.path.join(HERE, 'libcliora_chart.so')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-Wall', '-Wno-unused-function', '-Wno-pass-failed', '-Wno-unused-value']


FLAGS += os.environ.get('CLIORA_BUILD_EXTRA', '').split()      # diagnostic builds (e.g. -DCLIORA_PERSIST_STAMPS); use with --force


def sources():
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')) + glob.glob(os.path.join(CSRC, '*.cpp')))


def headers():
    return sorted(glob.glob(os.path.join(CSRC, '*.hpp')) + glob.glob(os.path.join(HERE, '..', 'include', '*.h')))


def _obj(src):
    return os.path.join(OBJ, os.path.basename(src) + '.o')


def _newest(paths):
    return max(os.path.getmtime(p) for p in paths)


def stale_objects(force=False):
    hdr_t = _newest(headers())
    out = []
    for s in sources():
        o = _obj(s)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hdr_t):
            out.append(s)
    return out


def build(force=False, verbose=True):
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    os.makedirs(OBJ, exist_ok=True)
    todo = stale_objects(force)
    objs = [_obj(s) for s in sources()]
    for o in glob.glob(os.path.join(OBJ, '*.o')):          # objects of sources that no longer exist
        if o not in objs:
            os.remove(o)

    def compile_one(src):
        lang = ['-x', 'hip'] if src.endswith('.hip') else []
        cmd = [hipcc] + FLAGS + lang + ['-c', src, '-o', _obj(src)]
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)

    if todo:
        with ThreadPoolExecutor(max_workers=min(len(todo), int(os.environ.get('CLIORA_BUILD_JOBS', '6')))) as ex:
            list(ex.map(compile_one, todo))
    if todo or not os.path.exists(LIB) or os.path.getmtime(LIB) < _newest(objs):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
