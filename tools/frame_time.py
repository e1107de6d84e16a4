#!/usr/bin/env python3
"""Forward time of RRDBNet-23 at 1080p with the library named by RSA_LIB (variant builds: tools/variant.sh); one line per run.
usage: [RSA_LIB=variants/lib_x.so] frame_time.py [precision] [rounds]"""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from resselt_amd.engine import lib as L  # noqa: E402

if os.environ.get('RSA_LIB'):
    _p = os.path.abspath(os.environ['RSA_LIB'])
    L.lib_path = lambda: _p  # type: ignore
import resselt_amd  # noqa: E402
from resselt_amd.utils import synth  # noqa: E402

dev = torch.device('cuda:0')
prec = sys.argv[1] if len(sys.argv) > 1 else 'auto'
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
m = resselt_amd.load_from_state_dict(dict(synth.rrdbnet_state_dict(nb=23, seed=0))).to(dev)
m.precision = prec
x = synth.synth_input((1, 3, 1080, 1920), seed=0).to(dev)
for _ in range(2):
    m(x)
torch.cuda.synchronize()
t = []
for _ in range(rounds):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        m(x)
    e1.record()
    torch.cuda.synchronize()
    t.append(e0.elapsed_time(e1) / 3)
L.check_status('frame_time')
print(f'{os.environ.get("RSA_LIB", "product")} {prec}: med {statistics.median(t):.2f} min {min(t):.2f} ms  aborts={L.ring_aborts()}', flush=True)
