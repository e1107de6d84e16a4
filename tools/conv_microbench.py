#!/usr/bin/env python3
"""Time single fused-conv launches at BASELINE size (1080x1920) through the C-ABI; prints one line per config.

Used to A/B kernel variants (RSA_LIB=path/to/variant.so selects the library) and as the rocprofv3 --pmc target.
"""

import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from resselt_amd.engine import lib as L  # noqa: E402

if os.environ.get('RSA_LIB'):
    L.lib_path = lambda: os.environ['RSA_LIB']  # type: ignore
from resselt_amd.engine import ops, tensors  # noqa: E402

dev = torch.device('cuda:0')
H, W = int(os.environ.get('MB_H', 1080)), int(os.environ.get('MB_W', 1920))
iters = int(os.environ.get('MB_ITERS', 5))
configs = [(192, 64), (64, 32), (160, 32), (64, 64)]
if len(sys.argv) > 1:
    configs = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]]
for products in (3, 1):
    for cin, cout in configs:
        g = torch.Generator().manual_seed(0)
        w = (torch.rand((cout, cin, 3, 3), generator=g) - 0.5) * 0.1
        wts = ops.ConvWeights.from_oihw(w, torch.zeros(cout), products, device=dev)
        x = tensors.Planes.empty(1, cin // 8, H, W, dev)
        x.hi.copy_(torch.randn(x.hi.shape, device=dev).to(torch.bfloat16))
        x.lo.copy_((torch.randn(x.lo.shape, device=dev) * 0.004).to(torch.bfloat16))
        out = tensors.Planes.empty(1, cout // 8, H, W, dev)
        p = ops.conv_params(wts, x, H, W, out=out, act=L.ACT_LRELU, act_param=0.2)
        for _ in range(2):
            ops.run_convs([p], dev)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            ops.run_convs([p], dev)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        flop = 2.0 * cin * 9 * cout * H * W
        print(f'products={products} cin={cin:3d} cout={cout:3d}: {ms:8.3f} ms  {flop / ms / 1e9:8.1f} TF/s algorithmic  {flop * products / ms / 1e9:8.1f} TF/s issued ({flop * products / ms / 1e9 / 25:.1f}% of 2.5 PF)', flush=True)
