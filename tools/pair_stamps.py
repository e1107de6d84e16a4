#!/usr/bin/env python3
"""Phase accounting of the fused pair kernel on a diagnostic build (tools/variant.sh pair_stamps "-DRSA_PAIR_STAMPS" conv_inst_ringpair;
RSA_LIB=variants/lib_pair_stamps.so): per-wave s_memtime totals of one launch at 1080p, medians over the 256 workgroups.

columns (share of the wave's lifetime): ring = waiting for a ring slot (compute waves: FULL; loader: FREE), hand = waiting for the x_A image
hand-offs (loader: waiting for an older fill to land), epi = epilogue, xunit = layer B's last nine K steps (out of the image).
usage: RSA_LIB=variants/lib_pair_stamps.so pair_stamps.py [cinA ...]
"""

import ctypes as C
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from resselt_amd.engine import lib as L  # noqa: E402

_p = os.path.abspath(os.environ['RSA_LIB'])
L.lib_path = lambda: _p
from resselt_amd.engine import ops, tensors  # noqa: E402
from resselt_amd.engine.tensors import PF_F16  # noqa: E402

dev = torch.device('cuda:0')
lib = L.load()
raw = C.CDLL(_p)
raw.rsa_debug_pair_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
H, W = 1080, 1920
stream = ops.current_stream_ptr(dev)
for cin in [int(a) for a in sys.argv[1:]] or [64, 128]:
    ws = tensors.Planes.empty(1, 24, H, W, dev, True, PF_F16, lo_planes=8)
    ws.hi.copy_((torch.randn(ws.hi.shape, device=dev) * 0.5).to(torch.float16))
    wa = ops.ConvWeights.from_oihw((torch.rand((32, cin, 3, 3)) - 0.5) * 0.1, torch.zeros(32), 1, device=dev, fmt=PF_F16)
    wb = ops.ConvWeights.from_oihw((torch.rand((32, cin + 32, 3, 3)) - 0.5) * 0.1, torch.zeros(32), 1, device=dev, fmt=PF_F16)
    pa = cin // 8
    a = ops.conv_params(wa, ws, H, W, cin_planes=pa, out=ws, out_plane_off=pa, act=L.ACT_LRELU, act_param=0.2)
    b = ops.conv_params(wb, ws, H, W, cin_planes=pa + 4, out=ws, out_plane_off=pa + 4, act=L.ACT_LRELU, act_param=0.2)
    L.set_pair_fusion(1)
    for _ in range(5):
        L.conv2d_pair(a, b, stream)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * (256 * 9 * 8))()
    raw.rsa_debug_pair_stamps(buf, len(buf))  # clear
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    L.conv2d_pair(a, b, stream)
    e1.record()
    torch.cuda.synchronize()
    raw.rsa_debug_pair_stamps(buf, len(buf))
    v = list(buf)
    print(f'pair {cin}->32, {cin + 32}->32 at {H}x{W}: launch {e0.elapsed_time(e1) * 1e3:.1f} us (stamped build)')
    for role, waves in (('A waves (0-3)', range(0, 4)), ('B waves (4-7)', range(4, 8)), ('loader (8)', range(8, 9))):
        rows = [[v[((wg * 9 + w) * 8) + i] for i in range(6)] for wg in range(256) for w in waves if v[(wg * 9 + w) * 8] > 0]
        if not rows:
            continue
        tot = statistics.median(r[0] for r in rows)
        med = [statistics.median(r[i] for r in rows) for i in range(6)]
        print(f'  {role:14s} lifetime {tot:9.0f} ticks (100 MHz: {tot / 100:.1f} us)  ring {med[1] / tot:6.1%}  hand {med[2] / tot:6.1%}  epi {med[3] / tot:6.1%}  xunit {med[4] / tot:6.1%}'
              f'  tiles {med[5]:.0f}  other (multiply) {1 - (med[1] + med[2] + med[3] + med[4]) / tot:6.1%}')
L.set_pair_fusion(-1)
