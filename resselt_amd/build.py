"""Build ``libresselt_amd.so`` in-tree with hipcc for gfx950 (``python -m resselt_amd.build``).

The shared object lands next to this file so that it travels with the repository snapshot to the GPU
box (it is git-ignored, not gpurun-ignored).  No JIT, no torch extension machinery: one hipcc per source file + one link.
"""

from __future__ import annotations

import hashlib
import os
import re
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, 'csrc')
OUT = os.path.join(HERE, 'libresselt_amd.so')
STAMP = OUT + '.srchash'


def sources() -> list[str]:
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))


def _hash() -> str:
    h = hashlib.sha256()
    files = sources() + sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h'))
    files.append(os.path.join(ROOT, 'include', 'resselt_amd.h'))
    files.append(os.path.abspath(__file__))  # compile flags live here
    for f in files:
        h.update(os.path.relpath(f, ROOT).encode())  # relative: the same tree hashes the same wherever it is checked out
        with open(f, 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()


_INC = re.compile(r'^\s*#\s*include\s+"([^"]+)"', re.M)


def _deps(path: str, seen: set | None = None) -> set:
    """``path`` and every local header it includes, transitively (``#include "..."`` resolved in csrc/ and include/)."""
    seen = set() if seen is None else seen
    if path in seen:
        return seen
    seen.add(path)
    with open(path, encoding='utf-8', errors='replace') as fh:
        text = fh.read()
    for name in _INC.findall(text):
        for base in (os.path.dirname(path), CSRC, os.path.join(ROOT, 'include')):
            cand = os.path.normpath(os.path.join(base, name))
            if os.path.exists(cand):
                _deps(cand, seen)
                break
    return seen


def _unit_hash(src: str) -> str:
    """Hash of one translation unit: its source, the local headers it reaches, and this file (the compile flags)."""
    h = hashlib.sha256()
    for f in sorted(_deps(src)) + [os.path.abspath(__file__)]:
        h.update(os.path.relpath(f, ROOT).encode())
        with open(f, 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()


def _compile(args) -> str:
    hipcc, src, obj, verbose = args
    want = _unit_hash(src)
    stamp = obj + '.srchash'
    if os.path.exists(obj) and os.path.exists(stamp):
        with open(stamp) as fh:
            if fh.read().strip() == want:
                return obj  # this unit and everything it includes are unchanged
    cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-c', '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC, src, '-o', obj]
    if os.path.basename(src).startswith('conv_inst_ring'):
        # SLP vectorisation packs the epilogue's f32 adds / multiplies into v_pk_* instructions, which are slower beside MFMAs
        # (MI355X guide, 'price of one filler beside MFMAs'; measured +0.4..2 % per layer: profiles/r02_i_*)
        cmd.insert(3, '-fno-slp-vectorize')
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    with open(stamp, 'w') as fh:
        fh.write(want)
    return obj


def build(force: bool = False, verbose: bool = True) -> str:
    want = _hash()
    if not force and os.path.exists(OUT) and os.path.exists(STAMP) and open(STAMP).read().strip() == want:
        return OUT
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        raise RuntimeError('hipcc not found: cannot build libresselt_amd.so')
    # one hipcc per translation unit, in parallel (the convolution kernels dominate the compile time), then one link
    objdir = os.path.join(ROOT, 'build', 'obj')
    os.makedirs(objdir, exist_ok=True)
    jobs = [(hipcc, src, os.path.join(objdir, os.path.basename(src)[:-4] + '.o'), verbose) for src in sources()]
    from concurrent.futures import ThreadPoolExecutor

    with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as pool:
        objs = list(pool.map(_compile, jobs))
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', *objs, '-o', OUT]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    with open(STAMP, 'w') as fh:
        fh.write(want)
    return OUT


if __name__ == '__main__':
    build(force='--force' in sys.argv)
