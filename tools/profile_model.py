#!/usr/bin/env python3
"""Run a few forwards of one secondary config under rocprofv3 (`rocprofv3 --kernel-trace --stats -d DIR -o run -- python3 tools/profile_model.py dat`)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import resselt_amd  # noqa: E402
from resselt_amd.utils import synth  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else 'dat'
prec = sys.argv[2] if len(sys.argv) > 2 else 'auto'
dev = torch.device('cuda:0')
if which == 'dat':
    sd = synth.dat_state_dict(embed_dim=180, depth=(6,) * 6, num_heads=(6,) * 6, split_size=(8, 32), expansion_factor=4.0, upscale=4, img_size=64)
    shape, dt = (1, 3, 512, 512), torch.bfloat16
elif which == 'swinir':
    sd = synth.swinir_state_dict(embed_dim=240, depths=[6] * 9, num_heads=[8] * 9, upscale=4, upsampler='nearest+conv', resi='3conv')
    shape, dt = (1, 3, 1024, 1024), torch.bfloat16
elif which == 'hat':
    sd = synth.hat_state_dict(embed_dim=180, depths=(6,) * 6, num_heads=(6,) * 6, window=16, upscale=4, mlp_ratio=2.0)
    shape, dt = (1, 3, 512, 512), torch.bfloat16
elif which == 'drct':
    sd = synth.drct_state_dict(num_layers=6, upscale=4)
    shape, dt = (1, 3, 512, 512), torch.bfloat16
elif which == 'spanplus':
    sd = synth.spanplus_state_dict(upscale=4, upsampler='ps')
    shape, dt = (8, 3, 512, 512), torch.float16
else:
    raise SystemExit(f'unknown config {which}')
model = resselt_amd.load_from_state_dict(dict(sd)).to(dev)
model.precision = prec
x = synth.synth_input(shape, seed=0).to(dev).to(dt)
for _ in range(4):
    y = model(x)
torch.cuda.synchronize()
print(which, prec, tuple(y.shape), 'ok')
