#!/usr/bin/env python3
"""(Lives under tests/: like the test-suite it uses the CPU oracle as the checker.)

Randomised parity sweep on the GPU: every architecture x odd sizes x batch x tensor dtype against the CPU oracle.

usage: python tests/fuzz_parity.py [seed] [cases_per_arch]   -> one line per case, non-zero exit status on the first failure
"""

import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import torch  # noqa: E402

import resselt_amd  # noqa: E402
from helpers import oracle_forward  # noqa: E402
from resselt_amd.utils import synth  # noqa: E402


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    per_arch = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    rng = random.Random(seed)
    dev = torch.device('cuda:0')
    makers = {
        'esrgan': lambda s: (synth.rrdbnet_state_dict(nb=rng.choice([1, 2, 3]), scale=rng.choice([1, 2, 4]), plus=rng.random() < 0.3, seed=s), 1),
        'spanplus': lambda s: (synth.spanplus_state_dict(upscale=rng.choice([2, 3, 4]), upsampler=rng.choice(['ps', 'dys']), seed=s), 1),
        'span': lambda s: (synth.span_state_dict(upscale=rng.choice([2, 4]), seed=s), 1),
        'spanpp': lambda s: (synth.spanpp_state_dict(feature_channels=32, implicit_dim=32, latent_layers=2, seed=s), 1),
        'compact': lambda s: (synth.compact_state_dict(num_feat=rng.choice([32, 64]), num_conv=rng.choice([2, 5]), upscale=rng.choice([2, 4]), seed=s), 1),
        'swinir': lambda s: (synth.swinir_state_dict(embed_dim=60, depths=(2, 2), num_heads=(6, 6), upscale=rng.choice([2, 4]),
                                                     upsampler=rng.choice(['nearest+conv', 'pixelshuffle', 'pixelshuffledirect']), seed=s), 9),
        'swinir_restore': lambda s: ((lambda w7: synth.swinir_state_dict(in_ch=rng.choice([1, 3]), embed_dim=60, depths=(2, 2), num_heads=(6, 6), upscale=1,
                                                                         upsampler='', window=7 if w7 else 8, img_size=126 if w7 else 64, seed=s))(rng.random() < 0.5), 9),
        'dat': lambda s: (synth.dat_state_dict(embed_dim=64, depth=(3,), num_heads=(4,), split_size=rng.choice([(2, 4), (4, 8), (8, 8)]),
                                               upscale=rng.choice([2, 3]), img_size=16, seed=s), 2),
        'rtmosr': lambda s: (synth.rtmosr_state_dict(scale=rng.choice([2, 4]), dim=rng.choice([32, 48]), n_blocks=2, se=rng.random() < 0.6,
                                                     dccm=rng.random() < 0.7, seed=s), 3),
        'drct': lambda s: (synth.drct_state_dict(embed_dim=rng.choice([60, 96]), gc=16, window=rng.choice([4, 8]), num_layers=rng.choice([1, 2]), upscale=rng.choice([2, 3, 4]),
                                                 mlp_ratio=rng.choice([2.0, 4.0]), seed=s), 9),
        'esrganbig': lambda s: (synth.rrdbnet_state_dict(nb=2, scale=4, seed=s), 1),
        'hat': lambda s: (synth.hat_state_dict(embed_dim=60, depths=(2,), num_heads=(6,), window=rng.choice([4, 8]), upscale=rng.choice([2, 4]), seed=s), 9),
    }  # fmt: skip
    worst = 0.0
    for arch, make in makers.items():
        for k in range(per_arch):
            s = rng.randrange(1 << 20)
            sd, min_hw = make(s)
            n = 1 if arch == 'drct' else rng.choice([1, 1, 2, 3])
            h, w = rng.randint(min_hw, 45), rng.randint(min_hw, 45)
            if arch == 'esrganbig':  # many tiles per workgroup, ragged edges, a row-banded tail
                h, w = rng.randint(300, 420), rng.randint(500, 700)
            dt = rng.choice([torch.float32, torch.float32, torch.float16, torch.bfloat16])
            cin = sd['conv_first.weight'].shape[1] if arch == 'swinir_restore' else 3
            x = synth.synth_input((n, cin, h, w), seed=s).to(dt)
            with torch.no_grad():
                ref = oracle_forward(dict(arch='esrgan' if arch == 'esrganbig' else arch.split('_')[0]), sd, x.float())
            model = resselt_amd.load_from_state_dict(dict(sd)).to(dev)
            if arch == 'esrganbig':
                model.tail_band_rows = rng.choice([64, 100, 4096])
            xd = x.to(dev)
            if rng.random() < 0.3:  # a non-contiguous view of the same values
                xd = xd.permute(0, 1, 3, 2).contiguous().permute(0, 1, 3, 2)
            y = model(xd)
            torch.cuda.synchronize()
            scale = max(1.0, ref.abs().max().item())
            tol = {torch.float32: 3e-4, torch.float16: 2e-3, torch.bfloat16: 1.2e-2}[dt] * scale
            err = (y.float().cpu() - ref).abs().max().item()
            ok = y.shape == ref.shape and y.dtype == dt and err <= tol
            worst = max(worst, err / tol)
            print(f'{arch:14s} n={n} {h:2d}x{w:2d} {str(dt).split(".")[-1]:8s} -> {tuple(y.shape)} max-abs {err:.2e} (tol {tol:.1e}) {"ok" if ok else "FAIL"}', flush=True)
            if not ok:
                raise SystemExit(1)
            del model
    print(f'all cases passed; worst error / tolerance = {worst:.3f}')


if __name__ == '__main__':
    main()
