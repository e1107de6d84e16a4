#!/usr/bin/env python3
"""(Lives under tests/: like the test-suite it uses the CPU oracle as the checker.)

Parity at (or near) the BASELINE config sizes (SURVEY.md §8d "Parity"): max-abs and mean-abs error of the GPU engine against the
fp32 CPU oracle on identical synthetic weights and inputs, per config and per precision mode.  One JSON line per (config, mode).

For 16-bit tensor I/O the figures include the rounding of the OUTPUT tensor to that dtype (half an ulp at |y|max: 2.4e-4*|y| for fp16,
2e-3*|y| for bf16), which dominates the bf16x3 rows of the fp16/bf16 configs; `out_round` prints that bound next to the error.

Sizes are bounded so that the CPU oracle finishes in about a minute per config on the GPU box's host cores:
  C2 RRDBNet-23 x4          3x256x256 crop of the synthetic 1080p frame (full frame: ~12 min of CPU, tens of GB)
  C3 SPANPlus x4 ps / dys   the full 8x3x512x512 fp16 batch
  SPAN x4                   2x3x512x512 fp16
  C4 SwinIR-L x4            3x256x256 (bf16 I/O)
  DAT x4 (published size)   3x128x128 (bf16 I/O)
"""

import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import resselt_amd  # noqa: E402
from resselt_amd.utils import synth  # noqa: E402


def main():
    from oracle.compact import compact_forward
    from oracle.dat import dat_forward
    from oracle.hat import hat_forward
    from oracle.rrdbnet import rrdbnet_forward
    from oracle.span import span_forward, spanplus_forward
    from oracle.swinir import swinir_forward

    dev = torch.device('cuda:0')
    cases = [
        ('C2_rrdbnet23_x4_fp32_crop256', synth.rrdbnet_state_dict(nb=23, seed=0), (1, 3, 256, 256), torch.float32, rrdbnet_forward),
        ('C3_spanplus_x4_ps_fp16_b8_512', synth.spanplus_state_dict(upscale=4, upsampler='ps'), (8, 3, 512, 512), torch.float16, spanplus_forward),
        ('C3_spanplus_x4_dys_fp16_b2_512', synth.spanplus_state_dict(upscale=4, upsampler='dys'), (2, 3, 512, 512), torch.float16, spanplus_forward),
        ('span_x4_fp16_b2_512', synth.span_state_dict(upscale=4), (2, 3, 512, 512), torch.float16, span_forward),
        ('C4_swinir_L_x4_bf16_256', synth.swinir_state_dict(embed_dim=240, depths=[6] * 9, num_heads=[8] * 9, upscale=4, upsampler='nearest+conv',
                                                          resi='3conv'), (1, 3, 256, 256), torch.bfloat16, swinir_forward),
        ('C4_swinir_L_x4_fp32_256', synth.swinir_state_dict(embed_dim=240, depths=[6] * 9, num_heads=[8] * 9, upscale=4, upsampler='nearest+conv',
                                                          resi='3conv'), (1, 3, 256, 256), torch.float32, swinir_forward),
        ('dat_x4_bf16_128', synth.dat_state_dict(embed_dim=180, depth=(6,) * 6, num_heads=(6,) * 6, split_size=(8, 32), expansion_factor=4.0,
                                                 upscale=4, img_size=64), (1, 3, 128, 128), torch.bfloat16, dat_forward),
        ('dat_x4_fp32_128', synth.dat_state_dict(embed_dim=180, depth=(6,) * 6, num_heads=(6,) * 6, split_size=(8, 32), expansion_factor=4.0,
                                                 upscale=4, img_size=64), (1, 3, 128, 128), torch.float32, dat_forward),
        ('hat_x4_fp32_128', synth.hat_state_dict(embed_dim=180, depths=(6,) * 6, num_heads=(6,) * 6, window=16, upscale=4, mlp_ratio=2.0),
         (1, 3, 128, 128), torch.float32, hat_forward),
        ('compact_x4_fp16_b2_512', synth.compact_state_dict(num_feat=64, num_conv=16, upscale=4), (2, 3, 512, 512), torch.float16, compact_forward),
    ]  # fmt: skip
    only = sys.argv[1] if len(sys.argv) > 1 else ''
    for name, sd, shape, dt, oracle in cases:
        if only and only not in name:
            continue
        x = synth.synth_input(shape, seed=0).to(dt)
        t0 = time.perf_counter()
        with torch.no_grad():
            ref = oracle(sd, x.float())  # the oracle sees the same (already rounded) input values
        t_cpu = time.perf_counter() - t0
        model = resselt_amd.load_from_state_dict(dict(sd)).to(dev)
        for prec in ('bf16x3', 'bf16'):
            model.precision = prec
            y = model(x.to(dev))
            torch.cuda.synchronize()
            d = (y.float().cpu() - ref).abs()
            print(json.dumps(dict(config=name, precision=prec, in_shape=list(shape), io_dtype=str(dt).split('.')[-1], max_abs=float(d.max()),
                                  mean_abs=float(d.mean()), ref_absmax=float(ref.abs().max()), out_round=float(torch.finfo(dt).eps / 2 * ref.abs().max()) if dt != torch.float32 else 0.0,
                                  cpu_oracle_s=round(t_cpu, 1))), flush=True)  # fmt: skip
        del model
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
