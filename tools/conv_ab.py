#!/usr/bin/env python3
"""A/B several builds of libresselt_amd.so in ONE process (interleaved rounds, median + min per variant) on single conv layers.

usage: conv_ab.py name=path.so [name=path.so ...] -- cin,cout[,H,W[,ksize]] ...
"""

import ctypes as C
import statistics
import sys

import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from resselt_amd.engine import lib as L  # noqa: E402
from resselt_amd.engine import ops, tensors  # noqa: E402

args = sys.argv[1:]
split = args.index('--')
libs = dict(a.split('=') for a in args[:split])
configs = []
for a in args[split + 1 :]:
    v = [int(x) for x in a.split(',')]
    configs.append((v[0], v[1], v[2] if len(v) > 2 else 1080, v[3] if len(v) > 3 else 1920, v[4] if len(v) > 4 else 3))
dev = torch.device('cuda:0')
handles = {}
for name, path in libs.items():
    h = C.CDLL(path)
    h.rsa_conv2d_list.argtypes = [C.POINTER(L.ConvParams), C.c_int32, C.c_void_p]
    h.rsa_conv2d_list.restype = C.c_int
    handles[name] = h
rounds = int(os.environ.get('AB_ROUNDS', 7))
for products in (3, 1):
    for cin, cout, H, W, ks in configs:
        w = (torch.rand((cout, cin, ks, ks)) - 0.5) * 0.1
        wts = ops.ConvWeights.from_oihw(w, torch.zeros(cout), products, device=dev)
        x = tensors.Planes.empty(1, cin // 8, H, W, dev)
        x.hi.copy_(torch.randn(x.hi.shape, device=dev).to(torch.bfloat16))
        x.lo.copy_((torch.randn(x.lo.shape, device=dev) * 0.004).to(torch.bfloat16))
        out = tensors.Planes.empty(1, (cout + 7) // 8, H, W, dev)
        p = ops.conv_params(wts, x, H, W, out=out, act=L.ACT_LRELU, act_param=0.2)
        arr = (L.ConvParams * 1)(p)
        stream = C.c_void_p(ops.current_stream_ptr(dev))
        times = {n: [] for n in handles}
        for r in range(rounds + 1):
            for n, h in handles.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    assert h.rsa_conv2d_list(arr, 1, stream) == 0
                e1.record()
                torch.cuda.synchronize()
                if r:
                    times[n].append(e0.elapsed_time(e1) / 3)
        flop = 2.0 * cin * ks * ks * cout * H * W * products
        print(f'products={products} {cin}->{cout} k{ks} {H}x{W}: ' + '  '.join(f'{n}: med {statistics.median(t):.3f} min {min(t):.3f} ms ({flop / statistics.median(t) / 1e9:.0f} TF issued)' for n, t in times.items()), flush=True)
