"""SwinIR loader (drop-in for ``resselt/archs/swinir/__init__.py:10-119``)."""

from __future__ import annotations

import math
from typing import Mapping

from ...factory import Architecture, KeyCondition
from ...utilities.state_dict import get_pixelshuffle_params, get_seq_len
from .arch import SwinIR


class SwinIRArch(Architecture[SwinIR]):
    def __init__(self):
        super().__init__(
            uid='SwinIR',
            detect=KeyCondition.has_all(
                'layers.0.residual_group.blocks.0.norm1.weight',
                'conv_first.weight',
                'layers.0.residual_group.blocks.0.mlp.fc1.bias',
                'layers.0.residual_group.blocks.0.attn.relative_position_index',
            ),
        )

    def load(self, state_dict: Mapping[str, object]) -> SwinIR:
        sd = state_dict
        if 'conv_before_upsample.0.weight' in sd:
            upsampler = 'nearest+conv' if 'conv_up1.weight' in sd else 'pixelshuffle'
        elif 'upsample.0.weight' in sd:
            upsampler = 'pixelshuffledirect'
        else:
            upsampler = ''
        start_unshuffle = 1
        if 'conv_first.1.weight' in sd:  # the reference renames these keys in the caller's dict (swinir/__init__.py:45-47)
            sd['conv_first.weight'] = sd.pop('conv_first.1.weight')
            sd['conv_first.bias'] = sd.pop('conv_first.1.bias')
            start_unshuffle = round(math.sqrt(sd['conv_first.weight'].shape[1] // 3))
        num_in_ch = sd['conv_first.weight'].shape[1]
        num_out_ch = sd['conv_last.weight'].shape[0] if 'conv_last.weight' in sd else num_in_ch
        upscale = 1
        if upsampler == 'nearest+conv':
            upscale = 2 ** len([k for k in sd if 'conv_up' in k and 'bias' not in k])
        elif upsampler == 'pixelshuffle':
            upscale, _ = get_pixelshuffle_params(sd, 'upsample')
        elif upsampler == 'pixelshuffledirect':
            upscale = int(math.sqrt(sd['upsample.0.bias'].shape[0] // num_out_ch))
        embed_dim = sd['conv_first.weight'].shape[0]
        mlp_ratio = float(sd['layers.0.residual_group.blocks.0.mlp.fc1.bias'].shape[0] / embed_dim)
        window_size = int(math.sqrt(sd['layers.0.residual_group.blocks.0.attn.relative_position_index'].shape[0]))
        img_size = 64
        if 'layers.0.residual_group.blocks.1.attn_mask' in sd:
            img_size = int(math.sqrt(sd['layers.0.residual_group.blocks.1.attn_mask'].shape[0]) * window_size)
        depths, num_heads = [], []
        for i in range(get_seq_len(sd, 'layers')):
            depths.append(get_seq_len(sd, f'layers.{i}.residual_group.blocks'))
            num_heads.append(sd[f'layers.{i}.residual_group.blocks.0.attn.relative_position_bias_table'].shape[1])
        resi_connection = '1conv' if 'conv_after_body.weight' in sd else '3conv'
        img_range = 255.0 if window_size == 7 else 1.0  # only the JPEG models use window 7 and this range (:90)
        in_nc = num_in_ch // start_unshuffle**2
        model = SwinIR(img_size=img_size, patch_size=1, in_chans=in_nc, embed_dim=embed_dim, depths=depths, num_heads=num_heads,
                       window_size=window_size, mlp_ratio=mlp_ratio, upscale=upscale, img_range=img_range, upsampler=upsampler,
                       resi_connection=resi_connection, start_unshuffle=start_unshuffle)  # fmt: skip
        return self._enhance_model(model=model, in_channels=in_nc, out_channels=num_out_ch, upscale=upscale, name='SwinIR')
