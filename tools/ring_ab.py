#!/usr/bin/env python3
"""In-process A/B of the ring schedule (conv_ring.h) against the chunk-barrier kernels on single RRDB layers at 1080p.

Interleaved rounds, median and min per variant (rsa_debug_set_ring switches the schedule per descriptor).
usage: ring_ab.py [cin,cout[,H,W[,up]] ...]
"""

import ctypes as C
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from resselt_amd.engine import lib as L  # noqa: E402
from resselt_amd.engine import ops, tensors  # noqa: E402

dev = torch.device('cuda:0')
if os.environ.get('RSA_LIB'):
    _p = os.path.abspath(os.environ['RSA_LIB'])
    L.lib_path = lambda: _p  # an experiment build (tools/variant.sh)
lib = L.load()
configs = [(64, 32), (96, 32), (128, 32), (160, 32), (192, 64), (64, 64)]
if len(sys.argv) > 1:
    configs = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]]
rounds = int(os.environ.get('AB_ROUNDS', 7))
reps = int(os.environ.get('AB_REPS', 5))
for cfg in configs:
    cin, cout = cfg[0], cfg[1]
    H, W = (cfg[2], cfg[3]) if len(cfg) > 3 else (1080, 1920)
    up = bool(cfg[4]) if len(cfg) > 4 else False
    w = (torch.rand((cout, cin, 3, 3)) - 0.5) * 0.1
    wts = ops.ConvWeights.from_oihw(w, torch.zeros(cout), 3, device=dev)
    x = tensors.Planes.empty(1, cin // 8, H // 2 if up else H, W // 2 if up else W, dev)
    x.hi.copy_(torch.randn(x.hi.shape, device=dev).to(torch.bfloat16))
    x.lo.copy_((torch.randn(x.lo.shape, device=dev) * 0.004).to(torch.bfloat16))
    mode = os.environ.get('AB_DATA', 'random')  # operand-data experiments (power): zero, lozero, lo4 (lo keeps 4 significant bits)
    if mode == 'zero':
        x.hi.zero_(), x.lo.zero_()
    elif mode == 'lozero':
        x.lo.zero_()
    elif mode == 'lo4':
        x.lo.copy_((x.lo.view(torch.int16) & -16).view(torch.bfloat16))
    out = tensors.Planes.empty(1, (cout + 7) // 8, H, W, dev)
    descs = {}
    for name, mode in (('ring', 1), ('old', 0)):
        lib.rsa_debug_set_ring(mode)
        p = ops.conv_params(wts, x, H, W, upsample2x=up, out=out, act=L.ACT_LRELU, act_param=0.2)
        descs[name] = (L.ConvParams * 1)(p)
    lib.rsa_debug_set_ring(-1)
    stream = ops.current_stream_ptr(dev)
    times = {n: [] for n in descs}
    for r in range(rounds + 1):
        for n, arr in descs.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                L.conv2d_list(arr, stream)
            e1.record()
            torch.cuda.synchronize()
            if r:
                times[n].append(e0.elapsed_time(e1) / reps)
    flop = 2.0 * cin * 9 * cout * H * W * 3
    print(f'{cin}->{cout} {H}x{W}{" up" if up else ""}: ' + '  '.join(f'{n}: med {statistics.median(t):.3f} min {min(t):.3f} ms ({flop / statistics.median(t) / 1e9:.0f} TF issued)' for n, t in times.items())
          + f'  aborts={L.ring_aborts()}', flush=True)
