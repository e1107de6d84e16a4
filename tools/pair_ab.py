#!/usr/bin/env python3
"""In-process A/B of the fused pair kernel (csrc/conv_ring_pair.h) against the two separate growth-convolution launches it replaces, on
the (conv1, conv2) and (conv3, conv4) shapes of a residual dense block at 1080p; then the whole RRDBNet-23 frame with the fusion on / off.

Interleaved rounds, median and min per variant (rsa_debug_set_pair switches what rsa_conv2d_list does with the same descriptors).
usage: pair_ab.py [--frame] [cinA[,H,W] ...]
"""

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
if os.environ.get('RSA_LIB'):
    _p = os.path.abspath(os.environ['RSA_LIB'])
    L.lib_path = lambda: _p  # an experiment build (tools/variant.sh)
lib = L.load()
# MASKS="0 1 2 ..." on a -DRSA_RING_DEBUG build: runtime ablations of the fused kernel
#   1 no LDS-DMA fills, 2 no MFMA, 4 no weight loads, 8 no epilogues, 16 no LDS fragment reads, 32 no x_A unit of layer B
masks = [int(m) for m in os.environ.get('MASKS', '').split()]
args = [a for a in sys.argv[1:] if not a.startswith('--')]
configs = [tuple(int(v) for v in a.split(',')) for a in args] or [(64,), (128,)]
rounds = int(os.environ.get('AB_ROUNDS', 7))
reps = int(os.environ.get('AB_REPS', 10))
stream = ops.current_stream_ptr(dev)


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for cfg in configs:
    cin = cfg[0]
    H, W = (cfg[1], cfg[2]) if len(cfg) > 2 else (1080, 1920)
    ws = tensors.Planes.empty(1, 24, H, W, dev, True, PF_F16, lo_planes=8)
    ws.hi.copy_((torch.randn(ws.hi.shape, device=dev) * 0.5).to(torch.float16))
    wa = ops.ConvWeights.from_oihw((torch.rand((32, cin, 3, 3)) - 0.5) * 0.1, torch.zeros(32), 1, device=dev, fmt=PF_F16)
    wb = ops.ConvWeights.from_oihw((torch.rand((32, cin + 32, 3, 3)) - 0.5) * 0.1, torch.zeros(32), 1, device=dev, fmt=PF_F16)
    pa = cin // 8
    a = ops.conv_params(wa, ws, H, W, cin_planes=pa, out=ws, out_plane_off=pa, act=L.ACT_LRELU, act_param=0.2)
    b = ops.conv_params(wb, ws, H, W, cin_planes=pa + 4, out=ws, out_plane_off=pa + 4, act=L.ACT_LRELU, act_param=0.2)
    b.tile_order = 1
    arr = (L.ConvParams * 2)(a, b)
    times = {'separate': [], 'fused': []}
    for r in range(rounds + 1):
        for name, mode in (('separate', 0), ('fused', 1)):
            L.set_pair_fusion(mode)
            t = timed(lambda: L.conv2d_list(arr, stream))
            if r:
                times[name].append(t)
    for m in masks:
        lib.rsa_debug_ring_flags(m)
        L.set_pair_fusion(1)
        ts = [timed(lambda: L.conv2d_list(arr, stream)) for _ in range(4)][1:]
        torch.cuda.synchronize()
        print(f'   mask {m:2d}: fused med {statistics.median(ts) * 1e3:.1f} min {min(ts) * 1e3:.1f} us  aborts={L.ring_aborts()}', flush=True)
    if masks:
        lib.rsa_debug_ring_flags(0)
    L.set_pair_fusion(-1)
    flop = 2.0 * 9 * 32 * (cin + cin + 32) * H * W
    print(f'pair {cin}->32, {cin + 32}->32  {H}x{W}: ' + '  '.join(f'{n}: med {statistics.median(t) * 1e3:.1f} min {min(t) * 1e3:.1f} us ({flop / statistics.median(t) / 1e9:.0f} TF algorithmic)'
                                                                      for n, t in times.items()) + f'  aborts={L.ring_aborts()}', flush=True)

if '--frame' in sys.argv:
    import resselt_amd
    from resselt_amd.utils import synth

    sd = synth.rrdbnet_state_dict(nb=23, seed=0)
    model = resselt_amd.load_from_state_dict(dict(sd)).to(dev)
    x = synth.synth_input((1, 3, 1080, 1920), seed=0).to(dev)
    reps = 5
    ys = {}
    times = {'separate': [], 'fused': []}
    for r in range(4):
        for name, mode in (('separate', 0), ('fused', 1)):
            L.set_pair_fusion(mode)
            t = timed(lambda: ys.__setitem__(name, model(x)))
            if r:
                times[name].append(t)
    L.set_pair_fusion(-1)
    L.check_status('pair_ab')
    same = torch.equal(ys['separate'], ys['fused'])
    print('RRDBNet-23 1080p frame: ' + '  '.join(f'{n}: med {statistics.median(t):.2f} min {min(t):.2f} ms' for n, t in times.items()) + f'  bit-identical={same}  aborts={L.ring_aborts()}', flush=True)
