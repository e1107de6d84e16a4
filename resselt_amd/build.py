"""Build ``libresselt_amd.so`` in-tree with hipcc for gfx950 (``python -m resselt_amd.build``).

The shared object lands next to this file so that it travels with the repository snapshot to the GPU
box (it is git-ignored, not gpurun-ignored).  No JIT, no torch extension machinery: one hipcc invocation.
"""

from __future__ import annotations

import hashlib
import os
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
    for f in files:
        h.update(f.encode())
        with open(f, 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> str:
    want = _hash()
    if not force and os.path.exists(OUT) and os.path.exists(STAMP) and open(STAMP).read().strip() == want:
        return OUT
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        raise RuntimeError('hipcc not found: cannot build libresselt_amd.so')
    cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC,
           *sources(), '-o', OUT]  # fmt: skip
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    with open(STAMP, 'w') as fh:
        fh.write(want)
    return OUT


if __name__ == '__main__':
    build(force='--force' in sys.argv)
