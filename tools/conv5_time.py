#!/usr/bin/env python3
"""conv5 of a residual dense block (192 -> 64, one fp16 product, plane residuals) alone at 1080p, with one and with two residuals.
usage: [RSA_LIB=variants/lib_x.so] conv5_time.py"""
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

dev = torch.device('cuda:0')
H, W, cin, cout = 1080, 1920, 192, 64
w = (torch.rand((cout, cin, 3, 3)) - 0.5) * 0.1
wts = ops.ConvWeights.from_oihw(w, torch.zeros(cout), 1, device=dev, fmt=tensors.PF_F16)
mk = lambda: tensors.Planes.empty(1, cin // 8, H, W, dev, True, tensors.PF_F16, lo_planes=8)  # noqa: E731
x, r2, out = mk(), mk(), mk()
for t in (x, r2):
    t.hi.copy_(torch.randn(t.hi.shape, device=dev).to(torch.float16))
    t.lo.copy_((torch.randn(t.lo.shape, device=dev) * 0.0004).to(torch.float16))
res = []
for two in (False, True):
    kw = dict(res2=(r2, 0), beta=0.2) if two else {}
    p = ops.conv_params(wts, x, H, W, out=out, res1=(x, 0), alpha=0.2, **kw)
    arr = (L.ConvParams * 1)(p)
    st = ops.current_stream_ptr(dev)
    ts = []
    for rnd in range(5):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            L.conv2d_list(arr, st)
        e1.record()
        torch.cuda.synchronize()
        if rnd:
            ts.append(e0.elapsed_time(e1) / 10)
    res.append(f'{"two residuals" if two else "one residual"}: {statistics.median(ts) * 1e3:.1f} us')
print(os.environ.get('RSA_LIB', 'product'), os.environ.get('RSA_RING_XRES', ''), L.conv_kernel_name(p)[:40], ' | '.join(res), f'aborts={L.ring_aborts()}', flush=True)
