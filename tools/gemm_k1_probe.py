#!/usr/bin/env python3
"""One Linear layer of the transformer bodies through gemm_k1 (one fp16 product, hi planes in and out) at 512 x 512 tokens: time per launch with
the library named by RSA_LIB (ablation builds: tools/variant.sh gkablN "-DRSA_GK_ABL=N" gemm_k1).   usage: gemm_k1_probe.py [cin cout ...]"""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from resselt_amd.engine import lib as L  # noqa: E402
from resselt_amd.engine import ops, tensors  # noqa: E402
from resselt_amd.engine.tensors import PF_F16  # noqa: E402

dev = torch.device('cuda:0')
L.load()
H = W = 512
args = [int(a) for a in sys.argv[1:]] or [180, 540, 180, 180, 180, 360, 360, 180]
stream = ops.current_stream_ptr(dev)
for cin, cout in zip(args[::2], args[1::2]):
    x = tensors.Planes.empty(1, (cin + 7) // 8, H, W, dev, False, PF_F16)
    x.hi.copy_((torch.randn(x.hi.shape, device=dev) * 0.5).half())
    out = tensors.Planes.empty(1, (cout + 7) // 8, H, W, dev, False, PF_F16)
    w = ops.ConvWeights.from_oihw((torch.rand((cout, cin, 1, 1)) - 0.5) * 0.1, torch.zeros(cout), 1, device=dev, fmt=PF_F16)
    p = ops.conv_params(w, x, H, W, cin_planes=(cin + 7) // 8, out=out)
    arr = (L.ConvParams * 1)(p)
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            L.conv2d_list(arr, stream)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20 * 1e3)
    t = statistics.median(ts)
    gb = (x.hi.numel() + out.hi.numel()) * 2 / 1e9
    print(f'{os.environ.get("RSA_LIB", "product")}: {cin} -> {cout}: {L.conv_kernel_name(p)[:40]} {t:.1f} us  {2 * cin * cout * H * W / t / 1e6:.0f} TF/s  {gb / t * 1e3:.2f} TB/s', flush=True)
