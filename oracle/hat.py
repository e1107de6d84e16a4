"""Oracle for HAT (TEST INFRASTRUCTURE, see oracle/__init__.py).

Functional fp32 restatement of ``resselt/archs/hat/arch.py`` in eval mode (DropPath / Dropout are the identity) over the checkpoint's
own key names.  Pinned by tests/golden/hat_*.npz, which hold outputs of the reference itself.
"""

from __future__ import annotations

import math
from typing import Mapping

import torch
import torch.nn.functional as F

RGB_MEAN = (0.4488, 0.4371, 0.4040)  # resselt/archs/hat/arch.py:842


def _seq_len(sd, prefix: str) -> int:
    idx = {int(k[len(prefix) + 1 :].split('.')[0]) for k in sd if k.startswith(prefix + '.')}
    return max(idx) + 1 if idx else 0


def hat_hparams(sd: Mapping[str, torch.Tensor]) -> dict:
    """What HATArch.load infers (resselt/archs/hat/__init__.py)."""
    window = int(math.sqrt(sd['relative_position_index_SA'].shape[0]))
    ext = int(math.sqrt(sd['relative_position_index_OCA'].shape[1]))
    n_layers = _seq_len(sd, 'layers')
    upscale = 1
    for i in range(0, _seq_len(sd, 'upsample'), 2):
        w = sd[f'upsample.{i}.weight']
        upscale *= int(math.sqrt(w.shape[0] // w.shape[1]))
    return dict(
        window=window,
        ext=ext,
        depths=[_seq_len(sd, f'layers.{i}.residual_group.blocks') for i in range(n_layers)],
        heads=[sd[f'layers.{i}.residual_group.overlap_attn.relative_position_bias_table'].shape[1] for i in range(n_layers)],
        upscale=upscale,
        in_ch=sd['conv_first.weight'].shape[1],
    )


def _conv(sd, key, x, padding=1):
    return F.conv2d(x, sd[f'{key}.weight'], sd.get(f'{key}.bias'), padding=padding)


def _lin(sd, key, x):
    return F.linear(x, sd[f'{key}.weight'], sd.get(f'{key}.bias'))


def _ln(sd, key, x):
    w = sd[f'{key}.weight']
    return F.layer_norm(x, (w.shape[0],), w, sd[f'{key}.bias'], 1e-5)


def window_partition(x, ws):
    b, h, w, c = x.shape
    return x.view(b, h // ws, ws, w // ws, ws, c).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws, ws, c)


def window_reverse(win, ws, h, w):
    b = win.shape[0] // ((h // ws) * (w // ws))
    return win.view(b, h // ws, w // ws, ws, ws, -1).permute(0, 1, 3, 2, 4, 5).reshape(b, h, w, -1)


def shift_mask(h, w, ws, shift):
    """HAT.calculate_mask (arch.py:1036-1066)."""
    img = torch.zeros(1, h, w, 1)
    cnt = 0
    for a in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for b in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img[:, a, b, :] = cnt
            cnt += 1
    mw = window_partition(img, ws).view(-1, ws * ws)
    d = mw.unsqueeze(1) - mw.unsqueeze(2)
    return torch.where(d != 0, torch.full_like(d, -100.0), torch.zeros_like(d))


def cab(sd, key, x):
    """CAB + ChannelAttention (arch.py:20-59) on an NCHW map."""
    y = _conv(sd, f'{key}.cab.2', F.gelu(_conv(sd, f'{key}.cab.0', x)))
    a = F.adaptive_avg_pool2d(y, 1)
    a = torch.sigmoid(_conv(sd, f'{key}.cab.3.attention.3', F.relu(_conv(sd, f'{key}.cab.3.attention.1', a, padding=0)), padding=0))
    return y * a


def hab(sd, key, x, h, w, ws, heads, shift, rpi, mask, conv_scale=0.01):
    """HAB.forward (arch.py:297-348)."""
    b, _, c = x.shape
    shortcut = x
    y = _ln(sd, f'{key}.norm1', x).view(b, h, w, c)
    conv_x = cab(sd, f'{key}.conv_block', y.permute(0, 3, 1, 2)).permute(0, 2, 3, 1).reshape(b, h * w, c)
    if shift:
        y = torch.roll(y, shifts=(-shift, -shift), dims=(1, 2))
    n = ws * ws
    xw = window_partition(y, ws).view(-1, n, c)
    qkv = _lin(sd, f'{key}.attn.qkv', xw).reshape(-1, n, 3, heads, c // heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * (c // heads) ** -0.5, qkv[1], qkv[2]
    attn = q @ k.transpose(-2, -1)
    bias = sd[f'{key}.attn.relative_position_bias_table'][rpi.view(-1)].view(n, n, -1).permute(2, 0, 1)
    attn = attn + bias.unsqueeze(0)
    if shift:
        nw = mask.shape[0]
        attn = (attn.view(-1, nw, heads, n, n) + mask.view(1, nw, 1, n, n)).view(-1, heads, n, n)
    o = (attn.softmax(-1) @ v).transpose(1, 2).reshape(-1, n, c)
    o = _lin(sd, f'{key}.attn.proj', o)
    y = window_reverse(o.view(-1, ws, ws, c), ws, h, w)
    if shift:
        y = torch.roll(y, shifts=(shift, shift), dims=(1, 2))
    x = shortcut + y.reshape(b, h * w, c) + conv_x * conv_scale
    return x + _lin(sd, f'{key}.mlp.fc2', F.gelu(_lin(sd, f'{key}.mlp.fc1', _ln(sd, f'{key}.norm2', x))))


def ocab(sd, key, x, h, w, ws, ext, heads, rpi):
    """OCAB.forward (arch.py:403-482): queries from ws x ws windows, keys / values from the ext x ext windows around them (zero padded)."""
    b, _, c = x.shape
    shortcut = x
    qkv = _lin(sd, f'{key}.qkv', _ln(sd, f'{key}.norm1', x).view(b, h, w, c)).reshape(b, h, w, 3, c).permute(3, 0, 4, 1, 2)
    q = window_partition(qkv[0].permute(0, 2, 3, 1), ws).view(-1, ws * ws, c)
    kv = F.unfold(torch.cat((qkv[1], qkv[2]), dim=1), kernel_size=(ext, ext), stride=ws, padding=(ext - ws) // 2)  # b, 2c*ext*ext, nw
    nw = kv.shape[-1]
    kv = kv.view(b, 2, c, ext * ext, nw).permute(1, 0, 4, 3, 2).reshape(2, b * nw, ext * ext, c)
    d = c // heads
    qh = q.reshape(-1, ws * ws, heads, d).permute(0, 2, 1, 3) * d**-0.5
    kh = kv[0].reshape(-1, ext * ext, heads, d).permute(0, 2, 1, 3)
    vh = kv[1].reshape(-1, ext * ext, heads, d).permute(0, 2, 1, 3)
    attn = qh @ kh.transpose(-2, -1)
    bias = sd[f'{key}.relative_position_bias_table'][rpi.view(-1)].view(ws * ws, ext * ext, -1).permute(2, 0, 1)
    o = ((attn + bias.unsqueeze(0)).softmax(-1) @ vh).transpose(1, 2).reshape(-1, ws * ws, c)
    y = window_reverse(o.view(-1, ws, ws, c), ws, h, w).view(b, h * w, c)
    x = _lin(sd, f'{key}.proj', y) + shortcut
    return x + _lin(sd, f'{key}.mlp.fc2', F.gelu(_lin(sd, f'{key}.mlp.fc1', _ln(sd, f'{key}.norm2', x))))


def hat_forward(sd: Mapping[str, torch.Tensor], x: torch.Tensor) -> torch.Tensor:
    """HAT.forward (arch.py:1097-1110), eval mode, img_range 1."""
    hp = hat_hparams(sd)
    ws, ext, s = hp['window'], hp['ext'], hp['upscale']
    mean = torch.tensor(RGB_MEAN if hp['in_ch'] == 3 else [0.0], dtype=x.dtype).view(1, -1, 1, 1)
    h0, w0 = x.shape[2:]
    x = F.pad(x - mean, (0, (ws - w0 % ws) % ws, 0, (ws - h0 % ws) % ws), 'reflect')
    b, _, h, w = x.shape
    first = _conv(sd, 'conv_first', x)
    t = first.flatten(2).transpose(1, 2)
    if 'patch_embed.norm.weight' in sd:
        t = _ln(sd, 'patch_embed.norm', t)
    mask = shift_mask(h, w, ws, ws // 2)
    rpi_sa, rpi_oca = sd['relative_position_index_SA'], sd['relative_position_index_OCA']
    for i, depth in enumerate(hp['depths']):
        g = f'layers.{i}.residual_group'
        res = t
        for j in range(depth):
            t = hab(sd, f'{g}.blocks.{j}', t, h, w, ws, hp['heads'][i], 0 if j % 2 == 0 else ws // 2, rpi_sa, mask)
        t = ocab(sd, f'{g}.overlap_attn', t, h, w, ws, ext, hp['heads'][i], rpi_oca)
        img = t.transpose(1, 2).reshape(b, -1, h, w)
        if f'layers.{i}.conv.weight' in sd:
            img = _conv(sd, f'layers.{i}.conv', img)
        t = img.flatten(2).transpose(1, 2) + res
    t = _ln(sd, 'norm', t)
    y = t.transpose(1, 2).reshape(b, -1, h, w)
    if 'conv_after_body.weight' in sd:
        y = _conv(sd, 'conv_after_body', y)
    y = F.leaky_relu(_conv(sd, 'conv_before_upsample.0', y + first), 0.01)
    for i in range(0, _seq_len(sd, 'upsample'), 2):
        wt = sd[f'upsample.{i}.weight']
        y = F.pixel_shuffle(_conv(sd, f'upsample.{i}', y), int(math.sqrt(wt.shape[0] // wt.shape[1])))
    y = _conv(sd, 'conv_last', y) + mean
    return y[:, :, : h0 * s, : w0 * s]
