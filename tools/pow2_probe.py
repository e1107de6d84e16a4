#!/usr/bin/env python3
"""Does a power-of-two plane stride (512 x 512, 1024 x 1024 maps: every channel group / plane of a token sits a multiple of 4 MiB apart) cost
the transformer models HBM channel conflicts?  Times each model at its published power-of-two size and at a slightly wider, non-power-of-two
size, and prints ns per input pixel.  usage: pow2_probe.py [swinir dat hat drct]"""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import resselt_amd  # noqa: E402
from resselt_amd.utils import synth  # noqa: E402

dev = torch.device('cuda:0')
CASES = {
    'swinir': (lambda: synth.swinir_state_dict(embed_dim=240, depths=[6] * 9, num_heads=[8] * 9, upscale=4, upsampler='nearest+conv', resi='3conv'), torch.bfloat16,
               [(1024, 1024), (1024, 1032), (1024, 1056), (1000, 1000)]),
    'dat': (lambda: synth.dat_state_dict(embed_dim=180, depth=(6,) * 6, num_heads=(6,) * 6, split_size=(8, 32), expansion_factor=4.0, upscale=4, img_size=64), torch.bfloat16,
            [(512, 512), (512, 544), (480, 544)]),
    'hat': (lambda: synth.hat_state_dict(embed_dim=180, depths=(6,) * 6, num_heads=(6,) * 6, window=16, upscale=4, mlp_ratio=2.0), torch.bfloat16, [(512, 512), (512, 528)]),
    'drct': (lambda: synth.drct_state_dict(num_layers=6, upscale=4), torch.bfloat16, [(512, 512), (512, 528)]),
}
for name in sys.argv[1:] or list(CASES):
    make, dt, shapes = CASES[name]
    model = resselt_amd.load_from_state_dict(dict(make())).to(dev)
    for h, w in shapes:
        x = synth.synth_input((1, 3, h, w), seed=0).to(dev).to(dt)
        for _ in range(2):
            model(x)
        torch.cuda.synchronize()
        ts = []
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            model(x)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        t = statistics.median(ts)
        print(f'{name} {h}x{w}: {t:.2f} ms = {t * 1e6 / (h * w):.1f} ns per input pixel', flush=True)
    del model
    torch.cuda.empty_cache()
