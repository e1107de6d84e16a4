#!/usr/bin/env python3
"""Cold-start cost on the GPU: load + .to(device), weight packing, first forward (plan build), steady-state forward."""
import sys, time, torch
sys.path.insert(0, '.')
import resselt_amd
from resselt_amd.utils import synth
dev = torch.device('cuda:0')
torch.zeros(1, device=dev); torch.cuda.synchronize()
for name, sd, shape in (('rrdbnet23', synth.rrdbnet_state_dict(nb=23), (1, 3, 256, 256)), ('swinir_L', synth.swinir_state_dict(embed_dim=240, depths=[6]*9, num_heads=[8]*9, upscale=4, upsampler='nearest+conv', resi='3conv'), (1, 3, 256, 256)),
                        ('hat', synth.hat_state_dict(embed_dim=180, depths=(6,)*6, num_heads=(6,)*6, window=16, upscale=4), (1, 3, 128, 128))):
    t0 = time.perf_counter(); m = resselt_amd.load_from_state_dict(dict(sd)).to(dev); torch.cuda.synchronize(); t1 = time.perf_counter()
    W = m._weights(dev); torch.cuda.synchronize(); t2 = time.perf_counter()
    x = synth.synth_input(shape).to(dev); y = m(x); torch.cuda.synchronize(); t3 = time.perf_counter()
    y = m(x); torch.cuda.synchronize(); t4 = time.perf_counter()
    print(f'{name}: load+to(dev) {t1-t0:.2f} s, pack {t2-t1:.2f} s, first forward (plan build) {t3-t2:.2f} s, second forward {t4-t3:.3f} s', flush=True)
