"""SwinIR cases of tools/gen_golden.py (imported by it; same shims, same rules)."""

import torch

import resselt  # the reference, already importable (gen_golden.py set up sys.path and the shims)

from resselt_amd.utils import synth


def swinir_cases(save, meta_of):
    cases = [
        # SwinIR-L real-SR wiring (nearest+conv, 3conv, 8 heads x 30), shortened to 2 RSTB x 2 blocks
        ('swinir_L_like_x4_24x40', dict(embed_dim=240, depths=[2, 2], num_heads=[8, 8], upscale=4, upsampler='nearest+conv', resi='3conv'), (1, 3, 24, 40), 51),
        # non-multiple-of-window input: reflect padding + crop; classical head (pixelshuffle), 1conv, embed 180 (not a multiple of 8 x heads)
        ('swinir_M_like_ps_x2_19x30', dict(embed_dim=180, depths=[2], num_heads=[6], upscale=2, upsampler='pixelshuffle', resi='1conv'), (1, 3, 19, 30), 52),
        # lightweight head (pixelshuffledirect), embed 60, batch 2, 3 blocks (shifted block last)
        ('swinir_S_like_psd_x3_b2_16x16', dict(embed_dim=60, depths=[3], num_heads=[6], upscale=3, upsampler='pixelshuffledirect', resi='1conv'), (2, 3, 16, 16), 53),
        # restoration heads (upsampler ''): grey denoising (1 channel, window 8) and the JPEG wiring (window 7 -> img_range 255, img_size 126)
        ('swinir_dn_gray_19x21', dict(in_ch=1, embed_dim=60, depths=[2], num_heads=[6], upscale=1, upsampler='', resi='1conv', img_size=128), (1, 1, 19, 21), 55),
        ('swinir_jpeg_w7_b2_20x23', dict(embed_dim=96, depths=[2, 2], num_heads=[6, 6], window=7, upscale=1, upsampler='', resi='1conv', img_size=126), (2, 3, 20, 23), 56),
        ('swinir_x8_nearest_8x16', dict(embed_dim=64, depths=[2], num_heads=[2], upscale=8, upsampler='nearest+conv', resi='1conv'), (1, 3, 8, 16), 54),
    ]
    for name, kw, shape, seed in cases:
        sd = synth.swinir_state_dict(seed=seed, **kw)
        model = resselt.load_from_state_dict(dict(sd))
        assert set(model.state_dict().keys()) == set(sd.keys()), set(model.state_dict().keys()) ^ set(sd.keys())
        x = synth.synth_input(shape, seed)
        y = model(x)
        save(name, dict(arch='swinir', synth=kw, seed=seed, metadata=meta_of(model)), x=x, y=y)
    # one block pair in isolation (unshifted + shifted) on a 16x24 token map
    from resselt.archs.swinir.arch import SwinTransformerBlock

    seed = 61
    sd = synth.swinir_state_dict(embed_dim=240, depths=[2], num_heads=[8], seed=seed)
    C = 240
    x = synth.synth_input((1, 16 * 24, C), seed) * 2 - 1
    outs = {}
    t = x
    for j, shift in ((0, 0), (1, 4)):
        blk = SwinTransformerBlock(C, (64, 64), 8, window_size=8, shift_size=shift, mlp_ratio=2.0)
        pre = f'layers.0.residual_group.blocks.{j}.'
        blk.load_state_dict({k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)})
        t = blk(t, (16, 24))
        outs[f'block{j}'] = t
    save('blocks_swin', dict(arch='swinir', seed=seed, synth=dict(embed_dim=240, depths=[2], num_heads=[8])), x=x, **outs)
