"""Oracle for SPAN / SPANPlus and DySample (TEST INFRASTRUCTURE, see oracle/__init__.py)."""

from __future__ import annotations

from typing import Mapping

import torch
import torch.nn.functional as F


def conv3xc_fold(sd: Mapping[str, torch.Tensor], prefix: str) -> tuple[torch.Tensor, torch.Tensor]:
    """Collapse Conv3XC (1x1 -> 3x3 valid -> 1x1, plus 1x1 skip) into one 3x3 kernel and bias.

    Restates ``update_params`` (resselt/archs/spanplus/arch.py:66-92, resselt/archs/span/arch.py:120-150)
    as explicit tensor contractions:  W[o,i,:,:] = sum_{m,n} w3[o,n] * w2[n,m,:,:] * w1[m,i]  (+ skip at the centre),
    b = w3 (sum_{m,k} w2[:,m,k] b1[m] + b2) + b3 + b_sk.
    """
    w1 = sd[f'{prefix}.conv.0.weight'].double()[:, :, 0, 0]  # [m, i]
    b1 = sd[f'{prefix}.conv.0.bias'].double()
    w2 = sd[f'{prefix}.conv.1.weight'].double()  # [n, m, 3, 3]
    b2 = sd[f'{prefix}.conv.1.bias'].double()
    w3 = sd[f'{prefix}.conv.2.weight'].double()[:, :, 0, 0]  # [o, n]
    b3 = sd[f'{prefix}.conv.2.bias'].double()
    w = torch.einsum('on,nmyx,mi->oiyx', w3, w2, w1)
    b = w3 @ (torch.einsum('nmyx,m->n', w2, b1) + b2) + b3
    w[:, :, 1, 1] += sd[f'{prefix}.sk.weight'].double()[:, :, 0, 0]
    b = b + sd[f'{prefix}.sk.bias'].double()
    return w.float(), b.float()


def conv3xc(sd: Mapping[str, torch.Tensor], prefix: str, x: torch.Tensor) -> torch.Tensor:
    """Conv3XC.forward in its training-mode form (spanplus/arch.py:95-97): conv(pad0(x)) + sk(x)."""
    xp = F.pad(x, (1, 1, 1, 1))
    y = F.conv2d(xp, sd[f'{prefix}.conv.0.weight'], sd[f'{prefix}.conv.0.bias'])
    y = F.conv2d(y, sd[f'{prefix}.conv.1.weight'], sd[f'{prefix}.conv.1.bias'])
    y = F.conv2d(y, sd[f'{prefix}.conv.2.weight'], sd[f'{prefix}.conv.2.bias'])
    return y + F.conv2d(x, sd[f'{prefix}.sk.weight'], sd[f'{prefix}.sk.bias'])


def spab(sd, prefix: str, x: torch.Tensor, act) -> tuple[torch.Tensor, torch.Tensor]:
    """SPAB.forward (spanplus/arch.py:117-130, span/arch.py:167-180); returns (out, out1).

    The reference's activation is IN-PLACE (``nn.Mish(inplace=True)`` spanplus/arch.py:114, ``nn.SiLU(inplace=True)``
    span/arch.py:164), so by the time ``out1`` is returned (``end=True``) it aliases the *activated* tensor:
    the value concatenated into ``conv_cat`` is act(c1_r(x)), not the pre-activation (pinned by
    tests/golden/blocks_span.npz ``spab_out1``).
    """
    out1 = act(conv3xc(sd, f'{prefix}.c1_r', x))
    out2 = conv3xc(sd, f'{prefix}.c2_r', out1)
    out3 = conv3xc(sd, f'{prefix}.c3_r', act(out2))
    return (out3 + x) * (torch.sigmoid(out3) - 0.5), out1


def dysample(sd, prefix: str, x: torch.Tensor, scale: int, groups: int = 4) -> torch.Tensor:
    """DySample.forward (resselt/utilities/dysample.py:47-83), written without the pin_memory tensor."""
    offset = F.conv2d(x, sd[f'{prefix}.offset.weight'], sd[f'{prefix}.offset.bias'])
    scope = F.conv2d(x, sd[f'{prefix}.scope.weight'])
    offset = offset * scope.sigmoid() * 0.5 + sd[f'{prefix}.init_pos']
    B, _, H, W = offset.shape
    offset = offset.view(B, 2, -1, H, W)
    cw = torch.arange(W, dtype=x.dtype) + 0.5
    ch = torch.arange(H, dtype=x.dtype) + 0.5
    # coords[0, c, 0, h, w]: c = 0 -> x coordinate (w + .5), c = 1 -> y coordinate (h + .5)  (dysample.py:54-61)
    coords = torch.stack(torch.meshgrid([cw, ch], indexing='ij')).transpose(1, 2).unsqueeze(1).unsqueeze(0)
    normalizer = torch.tensor([W, H], dtype=x.dtype).view(1, 2, 1, 1, 1)
    coords = 2 * (coords + offset) / normalizer - 1
    coords = (
        F.pixel_shuffle(coords.reshape(B, -1, H, W), scale)
        .view(B, 2, -1, scale * H, scale * W)
        .permute(0, 2, 3, 4, 1)
        .contiguous()
        .flatten(0, 1)
    )
    out = F.grid_sample(x.reshape(B * groups, -1, H, W), coords, mode='bilinear', align_corners=False, padding_mode='border')
    out = out.view(B, -1, scale * H, scale * W)
    return F.conv2d(out, sd[f'{prefix}.end_conv.weight'], sd[f'{prefix}.end_conv.bias'])


def spanplus_hparams(sd) -> dict:
    """SpanPlusArch.load inference (resselt/archs/spanplus/__init__.py:16-27)."""
    n_feats = max(int(k.split('.')[1]) for k in sd if k.startswith('feats.'))
    blocks = []
    for i in range(n_feats):
        blocks.append(1 + max(int(k.split('.')[3]) for k in sd if k.startswith(f'feats.{i + 1}.block_n.')))
    w0 = sd['feats.0.eval_conv.weight']
    if 'upsampler.0.weight' in sd:
        ups, out_ch = 'ps', w0.shape[1]
        upscale = int(round((sd['upsampler.0.weight'].shape[0] // out_ch) ** 0.5))
    else:
        ups, out_ch = 'dys', sd['upsampler.end_conv.weight'].shape[0]
        upscale = int(round((sd['upsampler.offset.weight'].shape[0] // 8) ** 0.5))
    return dict(num_in_ch=w0.shape[1], num_out_ch=out_ch, feature_channels=w0.shape[0], blocks=blocks, upscale=upscale, upsampler=ups)


def spanplus_forward(sd, x: torch.Tensor) -> torch.Tensor:
    """SpanPlus.forward (resselt/archs/spanplus/arch.py:199-201) incl. SPABS.forward (:146-151)."""
    hp = spanplus_hparams(sd)
    t = conv3xc(sd, 'feats.0', x)
    for bi, nblk in enumerate(hp['blocks']):
        pre = f'feats.{bi + 1}'
        out_b1, _ = spab(sd, f'{pre}.block_1', t, F.mish)
        out_x = out_b1
        for j in range(nblk):
            out_x, _ = spab(sd, f'{pre}.block_n.{j}', out_x, F.mish)
        out_end, out_x_2 = spab(sd, f'{pre}.block_end', out_x, F.mish)
        out_end = conv3xc(sd, f'{pre}.conv_2', out_end)  # Dropout2d(0) == identity
        cat = torch.cat([t, out_end, out_b1, out_x_2], 1)
        t = F.conv2d(cat, sd[f'{pre}.conv_cat.weight'], sd[f'{pre}.conv_cat.bias'])
    if hp['upsampler'] == 'ps':
        t = F.conv2d(t, sd['upsampler.0.weight'], sd['upsampler.0.bias'], padding=1)
        return F.pixel_shuffle(t, hp['upscale'])
    return dysample(sd, 'upsampler', t, hp['upscale'])


SPAN_MEAN = (0.4488, 0.4371, 0.4040)  # resselt/archs/span/__init__.py:29 (not deducible from the checkpoint)
SPAN_RANGE = 255.0  # resselt/archs/span/__init__.py:28


def span_forward(sd, x: torch.Tensor) -> torch.Tensor:
    """SPAN.forward (resselt/archs/span/arch.py:231-250): normalise, 6 SPAB (SiLU), conv_cat, conv+PixelShuffle."""
    if 'no_norm' not in sd:
        x = (x - torch.tensor(SPAN_MEAN, dtype=x.dtype).view(1, 3, 1, 1)) * SPAN_RANGE
    feat = conv3xc(sd, 'conv_1', x)
    out_b1, _ = spab(sd, 'block_1', feat, F.silu)
    t = out_b1
    for i in range(2, 6):
        t, _ = spab(sd, f'block_{i}', t, F.silu)
    out_b6, out_b5_2 = spab(sd, 'block_6', t, F.silu)
    out_b6 = conv3xc(sd, 'conv_2', out_b6)
    out = F.conv2d(torch.cat([feat, out_b6, out_b1, out_b5_2], 1), sd['conv_cat.weight'], sd['conv_cat.bias'])
    up = F.conv2d(out, sd['upsampler.0.weight'], sd['upsampler.0.bias'], padding=1)
    upscale = int(round((up.shape[1] // x.shape[1]) ** 0.5))
    return F.pixel_shuffle(up, upscale)
