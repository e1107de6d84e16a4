#!/usr/bin/env python3
"""LayerNorm stream kernel (csrc/swin.hip::layernorm_kernel) against plain copies of the same bytes: f32 map [n][C/4][H][W][4] in, fp16 hi planes out.
usage: ln_probe.py [C H W]"""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from resselt_amd.engine import lib as L  # noqa: E402
from resselt_amd.engine import ops, tensors  # noqa: E402

dev = torch.device('cuda:0')
C_, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (180, 512, 512)
lib = L.load()
stream = ops.current_stream_ptr(dev)
x = torch.randn((1, (C_ + 3) // 4, H, W, 4), device=dev)
out = tensors.Planes.empty(1, (C_ + 7) // 8, H, W, dev, False, tensors.PF_F16)
g, b = torch.ones(max(C_, 256), device=dev), torch.zeros(max(C_, 256), device=dev)
lp = L.LayerNormParams()
lp.batch, lp.H, lp.W, lp.C, lp.eps = 1, H, W, C_, 1e-5
lp.x_f32, lp.gamma, lp.beta = x.data_ptr(), g.data_ptr(), b.data_ptr()
lp.out_hi, lp.out_lo = out.hi_ptr(), None
lp.out_plane_stride, lp.out_batch_stride, lp.out_fmt = out.plane_stride, out.batch_stride, out.fmt


def timed(fn, reps=20):
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return statistics.median(ts)


def ln():
    rc = lib.rsa_layernorm(lp, stream)
    assert rc == 0


nbytes = x.numel() * 4 + out.hi.numel() * 2
t = timed(ln)
print(f'rsa_layernorm C={C_} {H}x{W}: {t:.1f} us  {nbytes / t / 1e6:.2f} TB/s ({nbytes / 1e6:.0f} MB)', flush=True)
y = torch.empty_like(x, dtype=torch.float16)
t = timed(lambda: torch.ops.aten.copy_(y, x))
print(f'torch f32 -> f16 copy of the same bytes: {t:.1f} us  {nbytes / t / 1e6:.2f} TB/s', flush=True)
z = torch.empty_like(x)
t = timed(lambda: z.copy_(x))
print(f'torch f32 copy: {t:.1f} us  {2 * x.numel() * 4 / t / 1e6:.2f} TB/s', flush=True)
ref = torch.nn.functional.layer_norm(tensors.f32map_to_nchw(x, C_).permute(0, 2, 3, 1), (C_,)).permute(0, 3, 1, 2)
got = tensors.planes_to_nchw(out, C_)
print('max err vs torch', (got - ref).abs().max().item(), flush=True)
