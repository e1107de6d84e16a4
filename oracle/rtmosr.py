"""Oracle for RTMoSR (TEST INFRASTRUCTURE, see oracle/__init__.py).

Functional fp32 restatement of ``resselt/archs/rtmosr/arch.py`` in eval mode (every RepConv and OmniShift re-parameterised to one kernel,
arch.py:179-188, 253-277) over the checkpoint's own key names.  Pinned by tests/golden/rtmosr_*.npz (outputs of the reference itself).
"""

from __future__ import annotations

import math
from typing import Mapping

import torch
import torch.nn.functional as F

from .spanpp import fold_repconv


def _seq_len(sd, prefix: str) -> int:
    idx = {int(k[len(prefix) + 1 :].split('.')[0]) for k in sd if k.startswith(prefix + '.')}
    return max(idx) + 1 if idx else 0


def rtmosr_hparams(sd: Mapping[str, torch.Tensor]) -> dict:
    """What RTMoSRArch.load infers (resselt/archs/rtmosr/__init__.py:176-190)."""
    unshuffle = 'to_feat.1.alpha' in sd
    if unshuffle:
        scale = math.isqrt(sd['to_feat.1.conv_3x3_rep.weight'].shape[1] // 3)  # the loader's value: this is really the unshuffle factor
        dim = sd['to_feat.1.conv_3x3_rep.weight'].shape[0]
    else:
        scale = math.isqrt(sd['to_img.0.conv_3x3_rep.weight'].shape[0] // 3)
        dim = sd['to_feat.conv_3x3_rep.weight'].shape[0]
    return dict(unshuffle=unshuffle, scale=scale, dim=dim, dccm='body.0.fc2.alpha' in sd, se='body.0.conv.2.squeezing.0.weight' in sd,
                ffn=sd['body.0.fc1.conv_3x3_rep.weight'].shape[0] / dim / 2, n_blocks=_seq_len(sd, 'body'))  # fmt: skip


def _repconv(sd, key, x):
    w, b = fold_repconv(sd, key)
    return F.conv2d(x, w, b, padding=1)


def omnishift_kernel(sd, key):
    """OmniShift.reparam_5x5 (arch.py:253-277): identity + 1x1 + 3x3 + 5x5 depthwise branches as one 5x5 depthwise kernel."""
    a1, a2, a3, a4 = (sd[f'{key}.alpha{k}'].transpose(0, 1) for k in (1, 2, 3, 4))
    w1 = F.pad(sd[f'{key}.conv1x1.weight'], (2, 2, 2, 2))
    w3 = F.pad(sd[f'{key}.conv3x3.weight'], (1, 1, 1, 1))
    ident = F.pad(torch.ones_like(sd[f'{key}.conv1x1.weight']), (2, 2, 2, 2))
    w = a1 * ident + a2 * w1 + a3 * w3 + a4 * sd[f'{key}.conv5x5.weight']
    b = (sd[f'{key}.alpha2'].squeeze() * sd[f'{key}.conv1x1.bias'] + sd[f'{key}.alpha3'].squeeze() * sd[f'{key}.conv3x3.bias']
         + sd[f'{key}.alpha4'].squeeze() * sd[f'{key}.conv5x5.bias'])  # fmt: skip
    return w, b


def rmsnorm(sd, key, x, eps=1e-6):
    """RMSNorm, channels first (arch.py:25-37): x / (||x||_2 / sqrt(C) + eps) * scale + offset."""
    rms = x.norm(2, dim=1, keepdim=True) * x.shape[1] ** -0.5
    return sd[f'{key}.scale'][..., None, None] * (x / (rms + eps)) + sd[f'{key}.offset'][..., None, None]


def gated_block(sd, key, x, dim, hidden, dccm, se):
    """GatedCNNBlock.forward (arch.py:331-337)."""
    shortcut = x
    f = _repconv(sd, f'{key}.fc1', rmsnorm(sd, f'{key}.norm', x))
    g, i, c = torch.split(f, [hidden, hidden - dim, dim], dim=1)
    c = F.pixel_unshuffle(c, 2) + _repconv(sd, f'{key}.conv.0.poll.1', F.max_pool2d(c, 2, 2))
    w, b = omnishift_kernel(sd, f'{key}.conv.1')
    c = F.conv2d(c, w, b, padding=2, groups=c.shape[1])
    if se:
        s = c.mean(dim=(2, 3), keepdim=True)
        s = F.conv2d(F.relu(F.conv2d(s, sd[f'{key}.conv.2.squeezing.0.weight'], sd[f'{key}.conv.2.squeezing.0.bias'])),
                     sd[f'{key}.conv.2.squeezing.2.weight'], sd[f'{key}.conv.2.squeezing.2.bias'])  # fmt: skip
        c = c * F.hardsigmoid(s)
    c = F.pixel_shuffle(c, 2)
    y = F.mish(g) * torch.cat((i, c), dim=1)
    y = _repconv(sd, f'{key}.fc2', y) if dccm else F.conv2d(y, sd[f'{key}.fc2.weight'], sd[f'{key}.fc2.bias'])
    return F.mish(y) + shortcut


def rtmosr_forward(sd: Mapping[str, torch.Tensor], x: torch.Tensor) -> torch.Tensor:
    """RTMoSR.forward (arch.py:382-387), eval mode."""
    hp = rtmosr_hparams(sd)
    dim = hp['dim']
    hidden = int(hp['ffn'] * dim)
    s_int = math.isqrt(sd['to_img.0.conv_3x3_rep.weight'].shape[0] // 3)
    u = hp['scale'] if hp['unshuffle'] else 0  # unshuffle factor
    out_scale = s_int // u if u else s_int
    pad = (u if u else 1) * 2
    _, _, h, w = x.shape
    y = F.pad(x, (0, (pad - w % pad) % pad, 0, (pad - h % pad) % pad), 'reflect')
    if u:
        y = _repconv(sd, 'to_feat.1', F.pixel_unshuffle(y, u))
    else:
        y = _repconv(sd, 'to_feat', y)
    for i in range(hp['n_blocks']):
        y = gated_block(sd, f'body.{i}', y, dim, hidden, hp['dccm'], hp['se'])
    y = F.pixel_shuffle(_repconv(sd, 'to_img.0', y), s_int)
    return y[:, :, : h * out_scale, : w * out_scale] + F.interpolate(x, scale_factor=out_scale)
