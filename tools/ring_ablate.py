#!/usr/bin/env python3
"""Runtime ablations of the ring schedule on an experiment build (tools/variant.sh dbg "-DRSA_RING_DEBUG"; RSA_LIB=variants/lib_dbg.so).

mask bits: 1 no DMA, 2 no MFMA, 4 no weight loads, 8 no epilogue, 16 no LDS fragment reads.
usage: RSA_LIB=variants/lib_dbg.so [MODE=fp16] ring_ablate.py cin,cout [...] -- mask [mask ...]
MODE=fp16: the one-product fp16 layers of the 'mixed' policy (Cout 32: hi-only output; Cout 64: conv5 with both plane residuals, hi + lo output).
"""

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
from resselt_amd.engine import ops, tensors  # noqa: E402

args = sys.argv[1:]
split = args.index('--') if '--' in args else len(args)
configs = [tuple(int(v) for v in a.split(',')) for a in args[:split]] or [(160, 32), (192, 64)]
masks = [int(m) for m in args[split + 1 :]] or [0, 1, 2, 3, 4, 8, 16, 12, 28]
dev = torch.device('cuda:0')
lib = L.load()
H, W = 1080, 1920
reps = 5
fp16 = os.environ.get('MODE') == 'fp16'
for cin, cout in configs:
    w = (torch.rand((cout, cin, 3, 3)) - 0.5) * 0.1
    if fp16:
        wts = ops.ConvWeights.from_oihw(w, torch.zeros(cout), 1, device=dev, fmt=tensors.PF_F16)
        x = tensors.Planes.empty(1, cin // 8, H, W, dev, True, tensors.PF_F16, lo_planes=8)
        x.hi.copy_(torch.randn(x.hi.shape, device=dev).to(torch.float16))
        x.lo.copy_((torch.randn(x.lo.shape, device=dev) * 0.0004).to(torch.float16))
        if cout == 64:
            r2 = tensors.Planes.empty(1, cin // 8, H, W, dev, True, tensors.PF_F16, lo_planes=8)
            r2.hi.copy_(x.hi)
            r2.lo.copy_(x.lo)
            out = tensors.Planes.empty(1, cin // 8, H, W, dev, True, tensors.PF_F16, lo_planes=8)
            p = ops.conv_params(wts, x, H, W, out=out, res1=(x, 0), alpha=0.2, res2=(r2, 0), beta=0.2)
        else:
            out = tensors.Planes.empty(1, (cout + 7) // 8, H, W, dev, False, tensors.PF_F16)
            p = ops.conv_params(wts, x, H, W, out=out, act=L.ACT_LRELU, act_param=0.2)
    else:
        wts = ops.ConvWeights.from_oihw(w, torch.zeros(cout), 3, device=dev)
        x = tensors.Planes.empty(1, cin // 8, H, W, dev)
        x.hi.copy_(torch.randn(x.hi.shape, device=dev).to(torch.bfloat16))
        x.lo.copy_((torch.randn(x.lo.shape, device=dev) * 0.004).to(torch.bfloat16))
        out = tensors.Planes.empty(1, (cout + 7) // 8, H, W, dev)
        p = ops.conv_params(wts, x, H, W, out=out, act=L.ACT_LRELU, act_param=0.2)
    arr = (L.ConvParams * 1)(p)
    stream = ops.current_stream_ptr(dev)
    res = {}
    for rnd in range(4):
        for m in masks:
            assert lib.rsa_debug_ring_flags(m) == 0
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                L.conv2d_list(arr, stream)
            e1.record()
            torch.cuda.synchronize()
            if rnd:
                res.setdefault(m, []).append(e0.elapsed_time(e1) / reps)
    lib.rsa_debug_ring_flags(0)
    print(f'{L.conv_kernel_name(p)} {cin}->{cout}: ' + '  '.join(f'[{m}] {statistics.median(t):.3f}' for m, t in res.items()) + f'  aborts={L.ring_aborts()}', flush=True)
