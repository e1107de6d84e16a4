"""Oracle for RRDBNet / ESRGAN / Real-ESRGAN (TEST INFRASTRUCTURE, see oracle/__init__.py).

Functional restatement over the *old-arch* key names (``model.0``, ``model.1.sub.N.RDBk.convj.0`` ...)
that the reference module owns (resselt/archs/esrgan/arch.py:72-127).
"""

from __future__ import annotations

import math
from typing import Mapping

import torch
import torch.nn.functional as F


def _conv(sd: Mapping[str, torch.Tensor], key: str, x: torch.Tensor) -> torch.Tensor:
    # conv_block: Conv2d k3 s1 zero-pad 1 with bias (resselt/utilities/block.py:148-200)
    w = sd[f'{key}.weight']
    return F.conv2d(x, w, sd.get(f'{key}.bias'), padding=w.shape[-1] // 2)


def _lrelu(x: torch.Tensor) -> torch.Tensor:
    # act('leakyrelu') -> LeakyReLU(0.2) (resselt/utilities/block.py:17-30)
    return F.leaky_relu(x, 0.2)


def rdb_forward(sd: Mapping[str, torch.Tensor], prefix: str, x: torch.Tensor, plus: bool = False) -> torch.Tensor:
    """ResidualDenseBlock_5C.forward (resselt/utilities/block.py:454-465)."""
    x1 = _lrelu(_conv(sd, f'{prefix}.conv1.0', x))
    x2 = _lrelu(_conv(sd, f'{prefix}.conv2.0', torch.cat((x, x1), 1)))
    if plus:  # ESRGAN+ branch, block.py:457-463
        x2 = x2 + F.conv2d(x, sd[f'{prefix}.conv1x1.weight'])
    x3 = _lrelu(_conv(sd, f'{prefix}.conv3.0', torch.cat((x, x1, x2), 1)))
    x4 = _lrelu(_conv(sd, f'{prefix}.conv4.0', torch.cat((x, x1, x2, x3), 1)))
    if plus:
        x4 = x4 + x2
    x5 = _conv(sd, f'{prefix}.conv5.0', torch.cat((x, x1, x2, x3, x4), 1))  # CNA mode: no act on conv5 (block.py:437-452)
    return x5 * 0.2 + x


def rrdb_forward(sd: Mapping[str, torch.Tensor], prefix: str, x: torch.Tensor, plus: bool = False) -> torch.Tensor:
    """RRDB.forward (resselt/utilities/block.py:340-344)."""
    out = rdb_forward(sd, f'{prefix}.RDB1', x, plus)
    out = rdb_forward(sd, f'{prefix}.RDB2', out, plus)
    out = rdb_forward(sd, f'{prefix}.RDB3', out, plus)
    return out * 0.2 + x


def rrdbnet_hparams(sd: Mapping[str, torch.Tensor]) -> dict:
    """Hyper-parameters as ESRGANArch.load infers them (resselt/archs/esrgan/__init__.py:155-180)."""
    idx = sorted({int(k.split('.')[1]) for k in sd if k.startswith('model.')})
    seq_len = idx[-1] + 1
    nb = max(int(k.split('.')[3]) for k in sd if k.startswith('model.1.sub.'))
    in_nc = sd['model.0.weight'].shape[1]
    out_nc = sd[f'model.{seq_len - 1}.weight'].shape[0]
    scale = 2 ** ((seq_len - 5) // 3)
    shuffle = None
    if in_nc in (out_nc * 4, out_nc * 16):
        shuffle = int(math.sqrt(in_nc / out_nc))
    return dict(
        in_nc=in_nc,
        out_nc=out_nc,
        num_filters=sd['model.0.weight'].shape[0],
        num_blocks=nb,
        scale=scale,
        plus=any('.conv1x1.' in k for k in sd),
        shuffle_factor=shuffle,
    )


def rrdbnet_forward(sd: Mapping[str, torch.Tensor], x: torch.Tensor) -> torch.Tensor:
    """RRDBNet.forward (resselt/archs/esrgan/arch.py:129-138) on an old-arch state dict."""
    hp = rrdbnet_hparams(sd)
    nb, n_up = hp['num_blocks'], int(math.log2(hp['scale']))
    sf = hp['shuffle_factor']
    h0, w0 = x.shape[-2:]
    if sf:  # arch.py:130-137: reflect pad to a multiple, pixel-unshuffle, crop at the end
        x = F.pad(x, (0, (sf - w0 % sf) % sf, 0, (sf - h0 % sf) % sf), 'reflect')
        x = F.pixel_unshuffle(x, sf)
    fea = _conv(sd, 'model.0', x)
    t = fea
    for i in range(nb):
        t = rrdb_forward(sd, f'model.1.sub.{i}', t, hp['plus'])
    t = fea + _conv(sd, f'model.1.sub.{nb}', t)  # ShortcutBlock (block.py:83-91)
    k = 2
    for _ in range(n_up):  # upconv_block: nearest x2 -> conv -> lrelu (block.py:510-537)
        t = _lrelu(_conv(sd, f'model.{k + 1}', F.interpolate(t, scale_factor=2, mode='nearest')))
        k += 3
    t = _lrelu(_conv(sd, f'model.{k}', t))  # HR conv (arch.py:112-118)
    t = _conv(sd, f'model.{k + 2}', t)  # last conv (arch.py:120-126)
    if sf:
        s = hp['scale'] // sf
        t = t[:, :, : h0 * s, : w0 * s]
    return t
