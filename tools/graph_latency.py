#!/usr/bin/env python3
"""Eager launch list vs hipGraph replay (model.use_graph) on small and large inputs: median latency per forward."""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import resselt_amd  # noqa: E402
from resselt_amd.utils import synth  # noqa: E402


def timed(model, x, reps=15):
    for _ in range(3):
        model(x)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        model(x)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts) * 1e3


def main():
    dev = torch.device('cuda:0')
    cases = [
        ('rrdbnet23_x4', synth.rrdbnet_state_dict(nb=23), [(1, 3, 128, 128), (1, 3, 256, 256), (1, 3, 1080, 1920)]),
        ('spanplus_x4', synth.spanplus_state_dict(upscale=4, upsampler='ps'), [(1, 3, 256, 256), (1, 3, 720, 1280)]),
        ('swinir_L_x4', synth.swinir_state_dict(embed_dim=240, depths=[6] * 9, num_heads=[8] * 9, upscale=4, upsampler='nearest+conv', resi='3conv'),
         [(1, 3, 128, 128), (1, 3, 256, 256)]),
        ('hat_x4', synth.hat_state_dict(embed_dim=180, depths=(6,) * 6, num_heads=(6,) * 6, window=16, upscale=4), [(1, 3, 128, 128)]),
    ]  # fmt: skip
    for name, sd, shapes in cases:
        model = resselt_amd.load_from_state_dict(dict(sd)).to(dev)
        for shape in shapes:
            x = synth.synth_input(shape).to(dev)
            model.use_graph = False
            te = timed(model, x)
            model.use_graph = True
            tg = timed(model, x)
            print(f'{name} {shape[2]}x{shape[3]}: eager {te:.3f} ms, hipGraph replay {tg:.3f} ms ({te / tg:.2f}x), launches {model.launches_per_forward()}', flush=True)
        del model
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
