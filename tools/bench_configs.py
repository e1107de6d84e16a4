#!/usr/bin/env python3
"""Secondary BASELINE.json configs (C3 SPANPlus, C4 SwinIR-L, plus SPAN) on one MI355X: one JSON line per config.

The headline config (C2, RRDBNet-23 1080p) is bench.py; this script reports the other rows of SURVEY.md §8d with the same
conventions (input resident in HBM, synchronised, median of `--reps` after warm-up).
"""

import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import resselt_amd  # noqa: E402
from resselt_amd.utils import synth  # noqa: E402


def timed(model, x, reps, warm=2):
    for _ in range(warm):
        model(x)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        y = model(x)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return y, statistics.median(ts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--only', default='')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    cases = {
        'C3_spanplus_x4_ps_fp16_b8_512': (synth.spanplus_state_dict(upscale=4, upsampler='ps'), (8, 3, 512, 512), torch.float16, 53_154, 276),
        'C3_spanplus_x4_dys_fp16_b8_512': (synth.spanplus_state_dict(upscale=4, upsampler='dys'), (8, 3, 512, 512), torch.float16, None, None),
        'span_x4_fp16_b8_512': (synth.span_state_dict(upscale=4), (8, 3, 512, 512), torch.float16, 53_154, 276),
        'C4_swinir_L_x4_bf16_1024': (
            synth.swinir_state_dict(embed_dim=240, depths=[6] * 9, num_heads=[8] * 9, upscale=4, upsampler='nearest+conv', resi='3conv'),
            (1, 3, 1024, 1024), torch.bfloat16, 3_833_694, 27_600),
        # SURVEY.md §8 a17 (no BASELINE config): the published DAT x4 (embed 180, 6 groups x 6 blocks, 6 heads, split 8x32, expansion 4)
        'dat_x4_bf16_512': (
            synth.dat_state_dict(embed_dim=180, depth=(6,) * 6, num_heads=(6,) * 6, split_size=(8, 32), expansion_factor=4.0, upscale=4, img_size=64),
            (1, 3, 512, 512), torch.bfloat16, None, None),
        # HAT x4 at its published size (embed 180, 6 groups x 6 blocks, 6 heads, window 16, overlap 0.5, mlp 2)
        'hat_x4_bf16_512': (
            synth.hat_state_dict(embed_dim=180, depths=(6,) * 6, num_heads=(6,) * 6, window=16, upscale=4, mlp_ratio=2.0),
            (1, 3, 512, 512), torch.bfloat16, None, None),
        # DRCT x4 at its published size (embed 180, 6 dense groups, 6 heads, window 16, gc 32, mlp 2)
        'drct_x4_bf16_512': (synth.drct_state_dict(num_layers=6, upscale=4), (1, 3, 512, 512), torch.bfloat16, None, None),
        'compact_x4_fp16_b8_512': (synth.compact_state_dict(num_feat=64, num_conv=16, upscale=4), (8, 3, 512, 512), torch.float16, None, None),
    }  # fmt: skip
    for name, (sd, shape, dt, flop_px, bytes_px) in cases.items():
        if args.only and args.only not in name:
            continue
        model = resselt_amd.load_from_state_dict(dict(sd)).to(dev)
        resolved = None
        for prec in ('auto', 'bf16x3'):
            model.precision = prec
            if prec != 'auto' and model.resolved_precision() == resolved:
                continue  # 'auto' already is this mode
            resolved = model.resolved_precision()
            x = synth.synth_input(shape, seed=0).to(dev).to(dt)
            y, t = timed(model, x, args.reps)
            out_px = y.shape[0] * y.shape[2] * y.shape[3]
            macs = (model.macs_per_input_pixel() if hasattr(model, 'macs_per_input_pixel') else 0) * shape[0] * shape[2] * shape[3]
            rec = dict(config=name, precision=prec if prec == resolved else f'{prec} -> {resolved}', in_shape=list(shape), io_dtype=str(dt).split('.')[-1], ms=round(t * 1e3, 3),
                       out_mp_s=round(out_px / 1e6 / t, 2), algorithmic_tflops=round(2 * macs / t / 1e12, 2),
                       launches=model.launches_per_forward(), finite=bool(torch.isfinite(y.float()).all()))  # fmt: skip
            if bytes_px:
                rec['layerwise_hbm_gbs'] = round(bytes_px * out_px / t / 1e9, 1)
            print(json.dumps(rec), flush=True)
        del model
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
