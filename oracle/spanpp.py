"""Oracle for SpanPP (TEST INFRASTRUCTURE, see oracle/__init__.py).

Functional fp32 restatement of ``resselt/archs/spanpp/arch.py`` in eval mode -- the only mode in which the reference module can run:
``IGConv.forward`` reads ``eval_convs``, which only ``.train(mode)`` / ``.eval()`` populates (arch.py:277-291), and ``.eval()`` also
replaces every RepConv by its fused 3x3 (arch.py:166-193).  Pinned by tests/golden/spanpp_*.npz (outputs of the reference itself).
"""

from __future__ import annotations

import math
from typing import Mapping

import torch
import torch.nn.functional as F


def fold_seqconv(sd, key):
    """SeqConv3x3.rep_params (arch.py:138-150): 1x1 then 3x3 (bias-padded) == one zero-padded 3x3."""
    k0, b0, k1, b1 = (sd[f'{key}.{n}'] for n in ('k0', 'b0', 'k1', 'b1'))
    w = F.conv2d(k1, k0.permute(1, 0, 2, 3))
    b = F.conv2d(torch.ones(1, k0.shape[0], 3, 3) * b0.view(1, -1, 1, 1), k1).view(-1) + b1
    return w, b


def fold_conv3xc(sd, key):
    """Conv3XC.update_params (arch.py:63-90)."""
    w1, w2, w3 = (sd[f'{key}.conv.{i}.weight'] for i in range(3))
    b1, b2, b3 = (sd[f'{key}.conv.{i}.bias'] for i in range(3))
    w = F.conv2d(w1.flip(2, 3).permute(1, 0, 2, 3), w2, padding=2).flip(2, 3).permute(1, 0, 2, 3)
    w = F.conv2d(w.flip(2, 3).permute(1, 0, 2, 3), w3).flip(2, 3).permute(1, 0, 2, 3)
    b = (w2 * b1.reshape(1, -1, 1, 1)).sum((1, 2, 3)) + b2
    b = (w3 * b.reshape(1, -1, 1, 1)).sum((1, 2, 3)) + b3
    return w + F.pad(sd[f'{key}.sk.weight'], [1, 1, 1, 1]), b + sd[f'{key}.sk.bias']


def fold_repconv(sd, key):
    """RepConv.fuse (arch.py:166-176): alpha-weighted sum of the three branches."""
    a = sd[f'{key}.alpha']
    w1, b1 = fold_seqconv(sd, f'{key}.conv1')
    w3, b3 = fold_conv3xc(sd, f'{key}.conv3')
    w = a[0] * w1 + a[1] * sd[f'{key}.conv2.weight'] + a[2] * w3
    b = a[0] * b1 + a[1] * sd[f'{key}.conv2.bias'] + a[2] * b3
    return w, b


def _repconv(sd, key, x):
    w, b = fold_repconv(sd, key)
    return F.conv2d(x, w, b, padding=1)


def igconv_kernel(sd, scale: int, max_scale: int) -> torch.Tensor:
    """IGConv._implicit_representation_latent (arch.py:293-312): the [3*s*s, C, k, k] kernel of the scale-s head."""
    freq, amp = sd['upsampler.freq'], sd['upsampler.amplitude']
    n = freq.shape[0]
    r = torch.ones(1, 1, scale, scale) / min(scale, max_scale) * 2
    seq = [-1 + (1 / scale) + (2 / scale) * torch.arange(scale).float()] * 2
    coords = torch.stack(torch.meshgrid(*seq, indexing='ij'), dim=-1).flip(-1)  # make_coord (arch.py:220-231)
    coords = coords.unsqueeze(0).permute(0, 3, 1, 2).repeat(n, 1, 1, 1)
    f = freq.repeat(1, 1, scale, scale)
    a = amp.repeat(1, 1, scale, scale)
    f1, f2 = f.chunk(2, dim=1)
    f = f1 * coords[:, :1] + f2 * coords[:, 1:] + F.conv2d(r, sd['upsampler.phase.weight'], sd['upsampler.phase.bias'])
    x = torch.cat([torch.cos(math.pi * f), torch.sin(math.pi * f)], dim=1) * a
    i = 0
    while f'upsampler.query_kernel.{i}.weight' in sd:
        x = F.conv2d(x, sd[f'upsampler.query_kernel.{i}.weight'], sd[f'upsampler.query_kernel.{i}.bias'])
        if f'upsampler.query_kernel.{i + 2}.weight' in sd:
            x = F.relu(x)
        i += 2
    c = n // 9
    # '(Cin Kh Kw) RGB rh rw -> (RGB rh rw) Cin Kh Kw'
    return x.reshape(c, 3, 3, 3, scale, scale).permute(3, 4, 5, 0, 1, 2).reshape(3 * scale * scale, c, 3, 3)


def _spab(sd, key, x):
    """SPAB.forward (arch.py:204-216); the in-place SiLU makes the returned ``out1`` the ACTIVATED tensor."""
    out1 = F.silu(_repconv(sd, f'{key}.c1_r', x))
    out2 = F.silu(_repconv(sd, f'{key}.c2_r', out1))
    out3 = _repconv(sd, f'{key}.c3_r', out2)
    return (out3 + x) * (torch.sigmoid(out3) - 0.5), out1


def spanpp_forward(sd: Mapping[str, torch.Tensor], x: torch.Tensor, scale: int | None = None) -> torch.Tensor:
    """SpanPP.forward (arch.py:358-373), eval mode; ``scale=None`` selects eval_base_scale = 2 (arch.py:323, 284)."""
    scales = [int(v) for v in sd['MetaIGConv']] if 'MetaIGConv' in sd else [1, 2, 3, 4]
    s = 2 if scale is None else scale
    feat = _repconv(sd, 'conv0', x)
    b1, _ = _spab(sd, 'block_1', feat)
    cur = b1
    for i in range(2, 6):
        cur, _ = _spab(sd, f'block_{i}', cur)
    b6, b5_2 = _spab(sd, 'block_6', cur)
    b6 = _repconv(sd, 'conv_2', b6)
    out = F.conv2d(torch.cat([feat, b6, b1, b5_2], 1), sd['conv_cat.weight'], sd['conv_cat.bias'])
    return F.pixel_shuffle(F.conv2d(out, igconv_kernel(sd, s, max(scales)), None, padding=1), s)
