#!/usr/bin/env python3
"""A/B several builds of libresselt_amd.so in ONE process on single RRDB layers at 1080p (interleaved rounds, median / min).

usage: lib_ab.py name=path.so [name=path.so ...] -- cin,cout [cin,cout ...]
Each library packs its own weights and builds its own descriptor (schedules / layouts may differ between builds).
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

args = sys.argv[1:]
split = args.index('--')
libs = dict(a.split('=') for a in args[:split])
configs = [tuple(int(v) for v in a.split(',')) for a in args[split + 1 :]]
dev = torch.device('cuda:0')
H, W = int(os.environ.get('AB_H', 1080)), int(os.environ.get('AB_W', 1920))
rounds = int(os.environ.get('AB_ROUNDS', 7))
reps = int(os.environ.get('AB_REPS', 5))
for cin, cout in configs:
    ks = int(os.environ.get('AB_KS', 3))
    w = (torch.rand((cout, cin, ks, ks)) - 0.5) * 0.1
    x = tensors.Planes.empty(1, cin // 8, H, W, dev)
    x.hi.copy_(torch.randn(x.hi.shape, device=dev).to(torch.bfloat16))
    x.lo.copy_((torch.randn(x.lo.shape, device=dev) * 0.004).to(torch.bfloat16))
    out = tensors.Planes.empty(1, (cout + 7) // 8, H, W, dev)
    descs = {}
    for name, path in libs.items():
        L._lib = None
        _p = os.path.abspath(path)
        L.lib_path = lambda _p=_p: _p  # type: ignore
        lib = L.load()
        wts = ops.ConvWeights.from_oihw(w, torch.zeros(cout), 3, device=dev)
        p = ops.conv_params(wts, x, H, W, out=out, act=L.ACT_LRELU, act_param=0.2)
        descs[name] = (lib, (L.ConvParams * 1)(p), wts)
    stream = C.c_void_p(ops.current_stream_ptr(dev))
    times = {n: [] for n in descs}
    for r in range(rounds + 1):
        for n, (lib, arr, _) in descs.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                assert lib.rsa_conv2d_list(arr, 1, stream) == 0
            e1.record()
            torch.cuda.synchronize()
            if r:
                times[n].append(e0.elapsed_time(e1) / reps)
    flop = 2.0 * cin * ks * ks * cout * H * W * 3
    print(f'{cin}->{cout}: ' + '  '.join(f'{n}: med {statistics.median(t):.3f} min {min(t):.3f} ({flop / statistics.median(t) / 1e9:.0f} TF)' for n, t in times.items()), flush=True)
