"""Oracle for SwinIR (TEST INFRASTRUCTURE, see oracle/__init__.py).

Functional fp32 restatement of ``resselt/archs/swinir/arch.py`` over the checkpoint's own key names.
"""

from __future__ import annotations

import math
from typing import Mapping

import torch
import torch.nn.functional as F

RGB_MEAN = (0.4488, 0.4371, 0.4040)  # resselt/archs/swinir/arch.py:788-790


def swinir_hparams(sd: Mapping[str, torch.Tensor]) -> dict:
    """Hyper-parameters as SwinIRArch.load infers them (resselt/archs/swinir/__init__.py:22-119)."""
    if 'conv_before_upsample.0.weight' in sd:
        upsampler = 'nearest+conv' if 'conv_up1.weight' in sd else 'pixelshuffle'
    elif 'upsample.0.weight' in sd:
        upsampler = 'pixelshuffledirect'
    else:
        upsampler = ''
    num_in_ch = sd['conv_first.weight'].shape[1]
    num_out_ch = sd['conv_last.weight'].shape[0] if 'conv_last.weight' in sd else num_in_ch
    upscale = 1
    if upsampler == 'nearest+conv':
        upscale = 2 ** len([k for k in sd if 'conv_up' in k and 'bias' not in k])
    elif upsampler == 'pixelshuffle':
        for i in range(0, 10, 2):
            w = sd.get(f'upsample.{i}.weight')
            if w is None:
                break
            upscale *= math.isqrt(w.shape[0] // w.shape[1])
    elif upsampler == 'pixelshuffledirect':
        upscale = int(math.sqrt(sd['upsample.0.bias'].shape[0] // num_out_ch))
    embed_dim = sd['conv_first.weight'].shape[0]
    window = int(math.sqrt(sd['layers.0.residual_group.blocks.0.attn.relative_position_index'].shape[0]))
    n_layers = 1 + max(int(k.split('.')[1]) for k in sd if k.startswith('layers.'))
    depths, heads = [], []
    for i in range(n_layers):
        depths.append(1 + max(int(k.split('.')[4]) for k in sd if k.startswith(f'layers.{i}.residual_group.blocks.')))
        heads.append(sd[f'layers.{i}.residual_group.blocks.0.attn.relative_position_bias_table'].shape[1])
    return dict(
        upsampler=upsampler,
        in_ch=num_in_ch,
        out_ch=num_out_ch,
        upscale=upscale,
        embed_dim=embed_dim,
        window=window,
        depths=depths,
        heads=heads,
        resi='1conv' if 'conv_after_body.weight' in sd else '3conv',
        img_range=255.0 if window == 7 else 1.0,  # swinir/__init__.py:90
    )


def _conv(sd, key, x):
    w = sd[f'{key}.weight']
    return F.conv2d(x, w, sd.get(f'{key}.bias'), padding=w.shape[-1] // 2)


def _resi_conv(sd, key, x, kind):
    """'1conv' or '3conv' residual tail (arch.py:562-574)."""
    if kind == '1conv':
        return _conv(sd, key, x)
    x = F.leaky_relu(_conv(sd, f'{key}.0', x), 0.2)
    x = F.leaky_relu(_conv(sd, f'{key}.2', x), 0.2)
    return _conv(sd, f'{key}.4', x)


def _ln(sd, key, x):
    return F.layer_norm(x, (x.shape[-1],), sd[f'{key}.weight'], sd[f'{key}.bias'], 1e-5)


def window_partition(x, w):
    """arch.py:43-55"""
    B, H, W, C = x.shape
    return x.view(B, H // w, w, W // w, w, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, w * w, C)


def window_reverse(win, w, H, W):
    """arch.py:58-72"""
    B = win.shape[0] // ((H // w) * (W // w))
    return win.view(B, H // w, W // w, w, w, -1).permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, -1)


def shift_mask(H, W, w, s):
    """SwinTransformerBlock.calculate_mask (arch.py:268-293): -100 between different roll regions."""
    img = torch.zeros(1, H, W, 1)
    cnt = 0
    for hs in (slice(0, -w), slice(-w, -s), slice(-s, None)):
        for ws in (slice(0, -w), slice(-w, -s), slice(-s, None)):
            img[:, hs, ws, :] = cnt
            cnt += 1
    mw = window_partition(img, w).squeeze(-1)
    diff = mw.unsqueeze(1) - mw.unsqueeze(2)
    return torch.where(diff != 0, torch.full_like(diff, -100.0), torch.zeros_like(diff))


def swin_block(sd, pre, x, H, W, w, shift, heads):
    """SwinTransformerBlock.forward (arch.py:295-335) with WindowAttention.forward (:133-173) and Mlp (:34-40)."""
    B, L, C = x.shape
    shortcut = x
    t = _ln(sd, f'{pre}.norm1', x).view(B, H, W, C)
    if shift > 0:
        t = torch.roll(t, shifts=(-shift, -shift), dims=(1, 2))
    win = window_partition(t, w)
    B_, N, _ = win.shape
    hd = C // heads
    qkv = F.linear(win, sd[f'{pre}.attn.qkv.weight'], sd[f'{pre}.attn.qkv.bias']).reshape(B_, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * hd**-0.5, qkv[1], qkv[2]
    attn = q @ k.transpose(-2, -1)
    idx = sd[f'{pre}.attn.relative_position_index'].view(-1).long()
    bias = sd[f'{pre}.attn.relative_position_bias_table'][idx].view(N, N, -1).permute(2, 0, 1)
    attn = attn + bias.unsqueeze(0)
    if shift > 0:
        mask = shift_mask(H, W, w, shift)
        nW = mask.shape[0]
        attn = (attn.view(B_ // nW, nW, heads, N, N) + mask.unsqueeze(1).unsqueeze(0)).view(-1, heads, N, N)
    attn = attn.softmax(-1)
    out = (attn @ v).transpose(1, 2).reshape(B_, N, C)
    out = F.linear(out, sd[f'{pre}.attn.proj.weight'], sd[f'{pre}.attn.proj.bias'])
    t = window_reverse(out, w, H, W)
    if shift > 0:
        t = torch.roll(t, shifts=(shift, shift), dims=(1, 2))
    x = shortcut + t.view(B, L, C)
    y = _ln(sd, f'{pre}.norm2', x)
    y = F.linear(F.gelu(F.linear(y, sd[f'{pre}.mlp.fc1.weight'], sd[f'{pre}.mlp.fc1.bias'])), sd[f'{pre}.mlp.fc2.weight'], sd[f'{pre}.mlp.fc2.bias'])
    return x + y


def swinir_forward(sd: Mapping[str, torch.Tensor], x: torch.Tensor) -> torch.Tensor:
    """SwinIR.forward (arch.py:962-1015) for the 'nearest+conv', 'pixelshuffle' and 'pixelshuffledirect' heads."""
    hp = swinir_hparams(sd)
    w, s = hp['window'], hp['upscale']
    H0, W0 = x.shape[-2:]
    x = F.pad(x, (0, (w - W0 % w) % w, 0, (w - H0 % w) % w), 'reflect') if (H0 % w or W0 % w) else x  # utilities/padding.py:24-29
    mean = torch.tensor(RGB_MEAN, dtype=x.dtype).view(1, 3, 1, 1) if hp['in_ch'] == 3 else torch.zeros(1, 1, 1, 1)
    x = (x - mean) * hp['img_range']
    first = _conv(sd, 'conv_first', x)
    B, C, H, W = first.shape
    t = first.flatten(2).transpose(1, 2)  # PatchEmbed (arch.py:638-642)
    if 'patch_embed.norm.weight' in sd:
        t = _ln(sd, 'patch_embed.norm', t)
    for i, depth in enumerate(hp['depths']):
        r = t
        for j in range(depth):
            r = swin_block(sd, f'layers.{i}.residual_group.blocks.{j}', r, H, W, w, 0 if j % 2 == 0 else w // 2, hp['heads'][i])
        img = r.transpose(1, 2).reshape(B, C, H, W)  # PatchUnEmbed
        t = _resi_conv(sd, f'layers.{i}.conv', img, hp['resi']).flatten(2).transpose(1, 2) + t  # RSTB.forward (arch.py:592-593)
    t = _ln(sd, 'norm', t).transpose(1, 2).reshape(B, C, H, W)
    body = _resi_conv(sd, 'conv_after_body', t, hp['resi']) + first
    if hp['upsampler'] == 'nearest+conv':
        y = F.leaky_relu(_conv(sd, 'conv_before_upsample.0', body), 0.01)  # nn.LeakyReLU default slope (arch.py:911)
        n_up = int(math.log2(s))
        for u in range(1, n_up + 1):
            y = F.leaky_relu(_conv(sd, f'conv_up{u}', F.interpolate(y, scale_factor=2, mode='nearest')), 0.2)
        y = _conv(sd, 'conv_last', F.leaky_relu(_conv(sd, 'conv_hr', y), 0.2))
    elif hp['upsampler'] == 'pixelshuffle':
        y = F.leaky_relu(_conv(sd, 'conv_before_upsample.0', body), 0.01)
        i = 0
        while f'upsample.{i}.weight' in sd:
            y = _conv(sd, f'upsample.{i}', y)
            y = F.pixel_shuffle(y, math.isqrt(y.shape[1] // sd[f'upsample.{i}.weight'].shape[1]))
            i += 2
        y = _conv(sd, 'conv_last', y)
    elif hp['upsampler'] == 'pixelshuffledirect':
        y = F.pixel_shuffle(_conv(sd, 'upsample.0', body), s)
    else:
        y = x + _conv(sd, 'conv_last', body)
    y = y / hp['img_range'] + mean
    return y[:, :, : H0 * s, : W0 * s]
