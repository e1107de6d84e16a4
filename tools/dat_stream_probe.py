#!/usr/bin/env python3
"""DAT's per-pixel statistics kernel (rsa_plane_stats_fmt) on data that stays in the Infinity Cache vs on a working set that does not
(the in-model case: the planes were written by another kernel ~100 MB of traffic earlier)."""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from resselt_amd.engine import lib as L  # noqa: E402
from resselt_amd.engine import ops, tensors  # noqa: E402

dev = torch.device('cuda:0')
lib = L.load()
stream = ops.current_stream_ptr(dev)
H = W = 512
C_ = 180
bufs = [tensors.Planes.empty(1, 23, H, W, dev, False, tensors.PF_F16) for _ in range(8)]
for b in bufs:
    b.hi.copy_(torch.randn(b.hi.shape, device=dev).half())
stats = torch.empty((H * W, 2), device=dev)


def run(b):
    L.check(lib.rsa_plane_stats_fmt(b.hi_ptr(0), None, b.plane_stride, b.batch_stride, 1, H, W, C_, 1e-5, b.fmt, stats.data_ptr(), stream), 'stats')


def timed(fn, reps=24):
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(reps):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return statistics.median(ts)


print(f'plane_stats, one 96 MB buffer again and again: {timed(lambda i: run(bufs[0])):.1f} us', flush=True)
print(f'plane_stats, eight buffers in turn (770 MB):   {timed(lambda i: run(bufs[i % 8])):.1f} us', flush=True)
