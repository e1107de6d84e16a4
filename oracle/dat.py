"""Oracle for DAT (TEST INFRASTRUCTURE, see oracle/__init__.py).

Functional fp32 restatement of ``resselt/archs/dat/arch.py`` in eval mode (BatchNorm uses its running statistics, DropPath and Dropout
are the identity) over the checkpoint's own key names.  Pinned by tests/golden/dat_*.npz, which hold outputs of the reference itself.
"""

from __future__ import annotations

import math
from typing import Mapping

import torch
import torch.nn.functional as F

RGB_MEAN = (0.4488, 0.4371, 0.4040)  # resselt/archs/dat/arch.py:879


def _seq_len(sd, prefix: str) -> int:
    idx = {int(k[len(prefix) + 1 :].split('.')[0]) for k in sd if k.startswith(prefix + '.')}
    return max(idx) + 1 if idx else 0


def dat_hparams(sd: Mapping[str, torch.Tensor]) -> dict:
    """Hyper-parameters as DatArch.load infers them (resselt/archs/dat/__init__.py:45-103)."""
    C = sd['conv_first.weight'].shape[0]
    in_ch = sd['conv_first.weight'].shape[1]
    n_layers = _seq_len(sd, 'layers')
    depth = [_seq_len(sd, f'layers.{i}.blocks') for i in range(n_layers)]
    heads = []
    for i in range(n_layers):
        if depth[i] >= 2:
            heads.append(sd[f'layers.{i}.blocks.1.attn.temperature'].shape[0])
        else:
            heads.append(sd[f'layers.{i}.blocks.0.attn.attns.0.pos.pos3.2.weight'].shape[0] * 2)
    upsampler = 'pixelshuffle' if 'conv_last.weight' in sd else 'pixelshuffledirect'
    if upsampler == 'pixelshuffle':
        upscale = 1
        for i in range(0, _seq_len(sd, 'upsample'), 2):
            w = sd[f'upsample.{i}.weight']
            upscale *= int(math.sqrt(w.shape[0] // w.shape[1]))
    else:
        upscale = int(math.sqrt(sd['upsample.0.weight'].shape[0] // in_ch))
    img_size = 64
    if 'layers.0.blocks.2.attn.attn_mask_0' in sd:
        nw, n, _ = sd['layers.0.blocks.2.attn.attn_mask_0'].shape
        img_size = int(math.sqrt(nw * n))
    split = [int(v) for v in (sd['layers.0.blocks.0.attn.attns.0.rpe_biases'][-1] + 1)]
    return dict(
        in_ch=in_ch,
        embed_dim=C,
        depth=depth,
        heads=heads,
        upsampler=upsampler,
        upscale=upscale,
        resi='1conv' if 'conv_after_body.weight' in sd else '3conv',
        qkv_bias='layers.0.blocks.0.attn.qkv.bias' in sd,
        expansion=float(sd['layers.0.blocks.0.ffn.fc1.weight'].shape[0] / C),
        img_size=img_size,
        split=split,
    )


def _shifted(rg: int, b: int) -> bool:  # arch.py:312, 453
    return (rg % 2 == 0 and b > 0 and (b - 2) % 4 == 0) or (rg % 2 != 0 and b % 4 == 0)


def _conv(sd, key, x, padding=1, groups=1):
    return F.conv2d(x, sd[f'{key}.weight'], sd.get(f'{key}.bias'), padding=padding, groups=groups)


def _lin(sd, key, x):
    return F.linear(x, sd[f'{key}.weight'], sd.get(f'{key}.bias'))


def _ln(sd, key, x):
    w = sd[f'{key}.weight']
    return F.layer_norm(x, (w.shape[0],), w, sd[f'{key}.bias'], 1e-5)


def _bn(sd, key, x):
    return F.batch_norm(x, sd[f'{key}.running_mean'], sd[f'{key}.running_var'], sd[f'{key}.weight'], sd[f'{key}.bias'], False, 0.0, 1e-5)


def _resi_conv(sd, key, x):
    if f'{key}.weight' in sd:
        return _conv(sd, key, x)
    x = F.leaky_relu(_conv(sd, f'{key}.0', x), 0.2)
    x = F.leaky_relu(_conv(sd, f'{key}.2', x, padding=0), 0.2)
    return _conv(sd, f'{key}.4', x)


def _to_windows(t, hs, ws, heads):
    """[B, H, W, C] -> [B*nW, heads, hs*ws, C/heads] (img2windows + the head split of im2win, arch.py:17-26, 216-222)."""
    B, H, W, C = t.shape
    t = t.view(B, H // hs, hs, W // ws, ws, heads, C // heads)
    return t.permute(0, 1, 3, 5, 2, 4, 6).reshape(-1, heads, hs * ws, C // heads)


def _from_windows(t, hs, ws, H, W):
    """[B*nW, heads, N, d] -> [B, H, W, heads*d] (arch.py:262-265, 29-39)."""
    nw = (H // hs) * (W // ws)
    B = t.shape[0] // nw
    heads, d = t.shape[1], t.shape[3]
    t = t.view(B, H // hs, W // ws, heads, hs, ws, d).permute(0, 1, 4, 2, 5, 3, 6)
    return t.reshape(B, H, W, heads * d)


def dynamic_pos_bias(sd, key: str) -> torch.Tensor:
    """DynamicPosBias (residual=False, arch.py:104-143) on the branch's rpe_biases, gathered to [heads, N, N] (arch.py:247-252)."""
    pos = _lin(sd, f'{key}.pos.pos_proj', sd[f'{key}.rpe_biases'])
    for k in ('pos1', 'pos2', 'pos3'):
        w = sd[f'{key}.pos.{k}.0.weight']
        pos = F.layer_norm(pos, (w.shape[0],), w, sd[f'{key}.pos.{k}.0.bias'], 1e-5)
        pos = _lin(sd, f'{key}.pos.{k}.2', F.relu(pos))
    idx = sd[f'{key}.relative_position_index']
    n = idx.shape[0]
    return pos[idx.reshape(-1)].view(n, n, -1).permute(2, 0, 1).contiguous()


def shift_masks(H, W, split, shift):
    """calculate_mask (arch.py:336-411): additive -100 masks [nW, N, N] for the two branches."""
    out = []
    for idx in (0, 1):
        hs, ws = (split[0], split[1]) if idx == 0 else (split[1], split[0])
        sh, sw = (shift[0], shift[1]) if idx == 0 else (shift[1], shift[0])
        img = torch.zeros(H, W)
        cnt = 0
        for a in (slice(0, -hs), slice(-hs, -sh), slice(-sh, None)):
            for b in (slice(0, -ws), slice(-ws, -sw), slice(-sw, None)):
                img[a, b] = cnt
                cnt += 1
        mw = img.view(H // hs, hs, W // ws, ws).permute(0, 2, 1, 3).reshape(-1, hs * ws)
        d = mw.unsqueeze(1) - mw.unsqueeze(2)
        out.append(torch.where(d != 0, torch.full_like(d, -100.0), torch.zeros_like(d)))
    return out


def spatial_attention(sd, key, q, k, v, hs, ws, heads, mask=None):
    """Spatial_Attention.forward (arch.py:224-267) on [B, H, W, C/2] maps."""
    B, H, W, C = q.shape
    scale = (C // heads) ** -0.5
    qw, kw, vw = (_to_windows(t, hs, ws, heads) for t in (q, k, v))
    attn = (qw * scale) @ kw.transpose(-2, -1)
    attn = attn + dynamic_pos_bias(sd, key).unsqueeze(0)
    if mask is not None:
        nw = mask.shape[0]
        n = attn.shape[-1]
        attn = (attn.view(B, nw, heads, n, n) + mask.view(1, nw, 1, n, n)).view(-1, heads, n, n)
    attn = attn.softmax(dim=-1)
    return _from_windows(attn @ vw, hs, ws, H, W)


def _aim_convs(sd, key, v_img):
    """dwconv -> BN -> GELU (arch.py:321-325)."""
    C = v_img.shape[1]
    return F.gelu(_bn(sd, f'{key}.dwconv.1', _conv(sd, f'{key}.dwconv.0', v_img, groups=C)))


def _channel_map(sd, key, img):
    x = F.adaptive_avg_pool2d(img, 1)
    x = F.gelu(_bn(sd, f'{key}.channel_interaction.2', _conv(sd, f'{key}.channel_interaction.1', x, padding=0)))
    return _conv(sd, f'{key}.channel_interaction.4', x, padding=0)  # [B, C, 1, 1]


def _spatial_map(sd, key, img):
    x = F.gelu(_bn(sd, f'{key}.spatial_interaction.1', _conv(sd, f'{key}.spatial_interaction.0', img, padding=0)))
    return _conv(sd, f'{key}.spatial_interaction.3', x, padding=0)  # [B, 1, H, W]


def adaptive_spatial_attention(sd, key, x, H, W, heads, split, shifted):
    """Adaptive_Spatial_Attention.forward (arch.py:430-513). ``x``: [B, H*W, C]."""
    B, L, C = x.shape
    qkv = _lin(sd, f'{key}.qkv', x).view(B, H, W, 3, C)
    v_img = qkv[:, :, :, 2].permute(0, 3, 1, 2)
    m = max(split)
    pad_r, pad_b = (m - W % m) % m, (m - H % m) % m
    qkv = F.pad(qkv, (0, 0, 0, 0, 0, pad_r, 0, pad_b))  # zeros AFTER the projection: padded tokens have q = k = v = 0
    Hp, Wp = H + pad_b, W + pad_r
    shift = [split[0] // 2, split[1] // 2]
    outs = []
    masks = shift_masks(Hp, Wp, split, shift) if shifted else (None, None)
    for idx in (0, 1):
        hs, ws = (split[0], split[1]) if idx == 0 else (split[1], split[0])
        sh, sw = (shift[0], shift[1]) if idx == 0 else (shift[1], shift[0])
        part = qkv[..., idx * (C // 2) : (idx + 1) * (C // 2)]
        if shifted:
            part = torch.roll(part, shifts=(-sh, -sw), dims=(1, 2))
        o = spatial_attention(sd, f'{key}.attns.{idx}', part[:, :, :, 0], part[:, :, :, 1], part[:, :, :, 2], hs, ws, heads // 2, masks[idx])
        if shifted:
            o = torch.roll(o, shifts=(sh, sw), dims=(1, 2))
        outs.append(o[:, :H, :W].reshape(B, L, C // 2))
    att = torch.cat(outs, dim=2)
    conv_x = _aim_convs(sd, key, v_img)
    cmap = _channel_map(sd, key, conv_x).view(B, 1, C)
    smap = _spatial_map(sd, key, att.transpose(1, 2).reshape(B, C, H, W))
    att = att * torch.sigmoid(cmap)
    conv_x = (torch.sigmoid(smap) * conv_x).permute(0, 2, 3, 1).reshape(B, L, C)
    return _lin(sd, f'{key}.proj', att + conv_x)


def adaptive_channel_attention(sd, key, x, H, W, heads):
    """Adaptive_Channel_Attention.forward (arch.py:565-612)."""
    B, N, C = x.shape
    d = C // heads
    qkv = _lin(sd, f'{key}.qkv', x).view(B, N, 3, heads, d).permute(2, 0, 3, 4, 1)  # [3, B, heads, d, N]
    q, k, v = qkv[0], qkv[1], qkv[2]
    v_img = v.reshape(B, C, H, W)
    q = F.normalize(q, dim=-1)
    k = F.normalize(k, dim=-1)
    attn = ((q @ k.transpose(-2, -1)) * sd[f'{key}.temperature']).softmax(dim=-1)
    att = (attn @ v).permute(0, 3, 1, 2).reshape(B, N, C)
    conv_x = _aim_convs(sd, key, v_img)
    cmap = _channel_map(sd, key, att.transpose(1, 2).reshape(B, C, H, W))
    smap = _spatial_map(sd, key, conv_x).permute(0, 2, 3, 1).reshape(B, N, 1)
    att = att * torch.sigmoid(smap)
    conv_x = (conv_x * torch.sigmoid(cmap)).permute(0, 2, 3, 1).reshape(B, N, C)
    return _lin(sd, f'{key}.proj', att + conv_x)


def sgfn(sd, key, x, H, W):
    """SGFN + SpatialGate (arch.py:42-101)."""
    B, N, _ = x.shape
    x = F.gelu(_lin(sd, f'{key}.fc1', x))
    x1, x2 = x.chunk(2, dim=-1)
    c2 = x2.shape[-1]
    x2 = _conv(sd, f'{key}.sg.conv', _ln(sd, f'{key}.sg.norm', x2).transpose(1, 2).reshape(B, c2, H, W), groups=c2)
    return _lin(sd, f'{key}.fc2', x1 * x2.flatten(2).transpose(1, 2))


def dat_forward(sd: Mapping[str, torch.Tensor], x: torch.Tensor) -> torch.Tensor:
    """DAT.forward (arch.py:970-990), eval mode."""
    hp = dat_hparams(sd)
    mean = torch.tensor(RGB_MEAN if hp['in_ch'] == 3 else [0.0], dtype=x.dtype).view(1, -1, 1, 1)
    x = x - mean  # img_range is 1.0 in every loader-built model
    B, _, H, W = x.shape
    first = _conv(sd, 'conv_first', x)
    t = _ln(sd, 'before_RG.1', first.flatten(2).transpose(1, 2))
    for i, d in enumerate(hp['depth']):
        res = t
        for j in range(d):
            b = f'layers.{i}.blocks.{j}'
            y = _ln(sd, f'{b}.norm1', t)
            if j % 2 == 0:
                t = t + adaptive_spatial_attention(sd, f'{b}.attn', y, H, W, hp['heads'][i], hp['split'], _shifted(i, j))
            else:
                t = t + adaptive_channel_attention(sd, f'{b}.attn', y, H, W, hp['heads'][i])
            t = t + sgfn(sd, f'{b}.ffn', _ln(sd, f'{b}.norm2', t), H, W)
        img = t.transpose(1, 2).reshape(B, -1, H, W)
        t = res + _resi_conv(sd, f'layers.{i}.conv', img).flatten(2).transpose(1, 2)
    t = _ln(sd, 'norm', t)
    x = _resi_conv(sd, 'conv_after_body', t.transpose(1, 2).reshape(B, -1, H, W)) + first
    if hp['upsampler'] == 'pixelshuffle':
        x = F.leaky_relu(_conv(sd, 'conv_before_upsample.0', x), 0.01)
        for i in range(0, _seq_len(sd, 'upsample'), 2):
            w = sd[f'upsample.{i}.weight']
            x = F.pixel_shuffle(_conv(sd, f'upsample.{i}', x), int(math.sqrt(w.shape[0] // w.shape[1])))
        x = _conv(sd, 'conv_last', x)
    else:
        x = F.pixel_shuffle(_conv(sd, 'upsample.0', x), hp['upscale'])
    return x + mean
