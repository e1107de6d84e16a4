"""Oracle for DRCT (TEST INFRASTRUCTURE, see oracle/__init__.py).

Functional fp32 restatement of ``resselt/archs/drct/arch.py`` over the checkpoint's own key names.  The Swin block and its window
attention are the ones of SwinIR (same code in the reference: drct/arch.py:102-198, 332-474), so ``oracle.swinir.swin_block`` is
reused; what is DRCT's own is the dense group (RDG, :204-329) and the top level (:617-792).
Pinned by tests/golden/drct_*.npz (outputs of the reference itself, tools/gen_golden.py).
"""

from __future__ import annotations

import math
from typing import Mapping

import torch
import torch.nn.functional as F

from .swinir import RGB_MEAN, _conv, _ln, swin_block


def drct_hparams(sd: Mapping[str, torch.Tensor]) -> dict:
    """What the reference loader infers (resselt/archs/drct/__init__.py:43-100)."""
    embed_dim = sd['conv_first.weight'].shape[0]
    n_layers = 1 + max(int(k.split('.')[1]) for k in sd if k.startswith('layers.'))
    heads = [sd[f'layers.{i}.swin1.attn.relative_position_bias_table'].shape[1] for i in range(n_layers)]
    window = (math.isqrt(sd['layers.0.swin1.attn.relative_position_bias_table'].shape[0]) + 1) // 2
    upscale = 1
    i = 0
    while f'upsample.{i}.weight' in sd:
        w = sd[f'upsample.{i}.weight']
        upscale *= math.isqrt(w.shape[0] // w.shape[1])
        i += 2
    return dict(in_ch=sd['conv_first.weight'].shape[1], embed_dim=embed_dim, n_layers=n_layers, heads=heads, window=window,
                gc=sd['layers.0.adjust1.weight'].shape[0], upscale=upscale, upsampler='pixelshuffle' if 'conv_last.weight' in sd else '')  # fmt: skip


def rdg_forward(sd, pre: str, x: torch.Tensor, H: int, W: int, window: int, num_heads: int) -> torch.Tensor:
    """RDG.forward (drct/arch.py:322-329): five Swin blocks over a growing concatenation, 1x1 'adjust' convs, x5 * 0.2 + x."""
    B, L, C = x.shape
    # a checkpoint saved without the attn_mask buffers loads with img_size = window (drct/__init__.py:78-82), and a block whose
    # input_resolution <= window drops its shift (drct/arch.py:373-376): swin2 / swin4 then run unshifted
    shifted = f'{pre}.swin2.attn_mask' in sd

    def adjust(j, t):  # pe(lrelu(adjust_j(pue(t)))): tokens -> image -> 1x1 conv -> tokens
        img = t.transpose(1, 2).reshape(B, t.shape[2], H, W)
        return _conv(sd, f'{pre}.adjust{j}', img).flatten(2).transpose(1, 2)

    feats = [x]
    for j in range(1, 6):
        cat = torch.cat(feats, -1)
        dim = cat.shape[-1]
        heads = num_heads - (dim % num_heads) if j > 1 else num_heads  # drct/arch.py:241, 256, 271, 286
        shift = window // 2 if j in (2, 4) and shifted else 0
        out = adjust(j, swin_block(sd, f'{pre}.swin{j}', cat, H, W, window, shift, heads))
        if j < 5:
            out = F.leaky_relu(out, 0.2)
        feats.append(out)
    return feats[5] * 0.2 + x


def drct_forward(sd: Mapping[str, torch.Tensor], x: torch.Tensor) -> torch.Tensor:
    """DRCT.forward (drct/arch.py:772-792), 'pixelshuffle' reconstruction."""
    hp = drct_hparams(sd)
    if hp['upsampler'] != 'pixelshuffle':
        raise NotImplementedError('the reference DRCT only reconstructs with the pixelshuffle upsampler')
    w, s = hp['window'], hp['upscale']
    H0, W0 = x.shape[-2:]
    mean = torch.tensor(RGB_MEAN, dtype=x.dtype).view(1, 3, 1, 1) if hp['in_ch'] == 3 else torch.zeros(1, 1, 1, 1)
    x = x - mean  # img_range = 1.0 (drct/__init__.py:43)
    if H0 % w or W0 % w:
        x = F.pad(x, (0, (w - W0 % w) % w, 0, (w - H0 % w) % w), 'reflect')
    first = _conv(sd, 'conv_first', x)
    B, C, H, W = first.shape
    t = first.flatten(2).transpose(1, 2)
    if 'patch_embed.norm.weight' in sd:
        t = _ln(sd, 'patch_embed.norm', t)
    for i in range(hp['n_layers']):
        t = rdg_forward(sd, f'layers.{i}', t, H, W, w, hp['heads'][i])
    t = _ln(sd, 'norm', t).transpose(1, 2).reshape(B, C, H, W)
    body = (_conv(sd, 'conv_after_body', t) if 'conv_after_body.weight' in sd else t) + first
    y = F.leaky_relu(_conv(sd, 'conv_before_upsample.0', body), 0.01)
    i = 0
    while f'upsample.{i}.weight' in sd:
        y = _conv(sd, f'upsample.{i}', y)
        y = F.pixel_shuffle(y, math.isqrt(y.shape[1] // sd[f'upsample.{i}.weight'].shape[1]))
        i += 2
    y = _conv(sd, 'conv_last', y) + mean
    return y[:, :, : H0 * s, : W0 * s]
