#!/usr/bin/env python3
"""C5-scale input on ONE GPU through the tiling driver: RealESRGAN-x4plus on a 4320x7680 frame (16 tiles of 1080p + 32 px halo).
Prints time, throughput, peak memory and (RSA_BIG_VERBOSE=1) the memory after every tile with the cached plans' sizes."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import resselt_amd  # noqa: E402
from resselt_amd import tiling  # noqa: E402
from resselt_amd.utils import synth  # noqa: E402

dev = torch.device('cuda:0')
h, w = (int(v) for v in (sys.argv[1:3] if len(sys.argv) > 2 else (4320, 7680)))
model = resselt_amd.load_from_state_dict(dict(synth.rrdbnet_state_dict(nb=23))).to(dev)
model.max_plan_bytes = int(os.environ.get('RSA_PLAN_BYTES', 24 << 30))
x = synth.synth_input((1, 3, h, w)).to(dev)
if os.environ.get('RSA_BIG_VERBOSE'):
    _run = tiling.run_tile

    def run_tile(m, xx, t, s):
        y = _run(m, xx, t, s)
        torch.cuda.synchronize()
        print(f'tile {t.index}: read {t.ry1 - t.ry0}x{t.rx1 - t.rx0}  allocated {torch.cuda.memory_allocated() / 2**30:.1f} GiB  peak {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB  '
              f'plans {[round(e[0].buffer_bytes() / 2**30, 1) for e in model._plans.values()]}', flush=True)
        return y

    tiling.run_tile = run_tile
tiling.upscale_tiled(model, x[:, :, :1080, :1920], 4, (1080, 1920), halo=32)  # warm-up: packing + plan
torch.cuda.synchronize()
torch.cuda.reset_peak_memory_stats()
t0 = time.perf_counter()
y = tiling.upscale_tiled(model, x, 4, (1080, 1920), halo=32)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f'RRDBNet-23 x4 {h}x{w} -> {tuple(y.shape)} in {dt:.2f} s = {y.shape[2] * y.shape[3] / 1e6 / dt:.1f} output MP/s on one MI355X; '
      f'peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB; finite {bool(torch.isfinite(y[:, :, ::64, ::64]).all())}')
