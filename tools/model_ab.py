#!/usr/bin/env python3
"""In-process A/B of whole-model variants of RRDBNet-23 at 1080p (interleaved rounds, median / min of the forward time).

variants: name=attr:value[,attr:value...]   attrs: ring (0/1, rsa_debug_set_ring while the plan is built), plane_residuals (0/1)
usage: model_ab.py base=ring:1,plane_residuals:1 f32res=ring:1,plane_residuals:0 old=ring:0,plane_residuals:0
"""

import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import resselt_amd  # noqa: E402
from resselt_amd.engine import lib as L  # noqa: E402
from resselt_amd.utils import synth  # noqa: E402

dev = torch.device('cuda:0')
lib = L.load()
H, W = int(os.environ.get('AB_H', 1080)), int(os.environ.get('AB_W', 1920))
sd = synth.rrdbnet_state_dict(nb=23, seed=0)
x = synth.synth_input((1, 3, H, W), seed=0).to(dev)
models = {}
for a in sys.argv[1:]:
    name, spec = a.split('=')
    kv = dict(t.split(':') for t in spec.split(','))
    m = resselt_amd.load_from_state_dict(dict(sd)).to(dev)
    m.plane_residuals = bool(int(kv.get('plane_residuals', 1)))
    if 'tail_band_rows' in kv:
        m.tail_band_rows = int(kv['tail_band_rows'])
    lib.rsa_debug_set_ring(int(kv.get('ring', 1)))
    y = m(x)  # builds the plan (descriptors carry the schedule)
    torch.cuda.synchronize()
    lib.rsa_debug_set_ring(-1)
    models[name] = m
    del y
rounds = int(os.environ.get('AB_ROUNDS', 5))
times = {n: [] for n in models}
for r in range(rounds + 1):
    for n, m in models.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(2):
            m(x)
        e1.record()
        torch.cuda.synchronize()
        if r:
            times[n].append(e0.elapsed_time(e1) / 2)
print('  '.join(f'{n}: med {statistics.median(t):.2f} min {min(t):.2f} ms' for n, t in times.items()), f'aborts={L.ring_aborts()}', flush=True)
