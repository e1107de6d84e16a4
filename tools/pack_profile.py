#!/usr/bin/env python3
"""cProfile of weight packing (`_weights(device)`) for one synthetic model on the GPU: where the cold-start pack time goes.
usage: pack_profile.py [swinir_L|spanplus|hat|rrdbnet23]"""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import resselt_amd  # noqa: E402
from resselt_amd.utils import synth  # noqa: E402

dev = torch.device('cuda:0')
torch.zeros(1, device=dev)
which = sys.argv[1] if len(sys.argv) > 1 else 'swinir_L'
make = {
    'rrdbnet23': lambda: synth.rrdbnet_state_dict(nb=23),
    'swinir_L': lambda: synth.swinir_state_dict(embed_dim=240, depths=[6] * 9, num_heads=[8] * 9, upscale=4, upsampler='nearest+conv', resi='3conv'),
    'hat': lambda: synth.hat_state_dict(embed_dim=180, depths=(6,) * 6, num_heads=(6,) * 6, window=16, upscale=4),
    'spanplus': lambda: synth.spanplus_state_dict(upscale=4, upsampler='ps'),
}[which]
m = resselt_amd.load_from_state_dict(dict(make())).to(dev)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
m._weights(dev)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('cumulative').print_stats(28)
