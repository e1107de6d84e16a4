"""DRCT loader (drop-in for ``resselt/archs/drct/__init__.py:9-118``; the reference's class there is mis-named ``MoSRArch``)."""

from __future__ import annotations

import math
from typing import Mapping

from ...factory import Architecture, KeyCondition
from ...utilities.state_dict import get_pixelshuffle_params, get_seq_len
from .arch import DRCT


class DRCTArch(Architecture[DRCT]):
    def __init__(self):
        super().__init__(
            uid='DRCT',
            detect=KeyCondition.has_all(
                'conv_first.weight',
                'conv_first.bias',
                'layers.0.swin1.norm1.weight',
                'layers.0.swin1.norm1.bias',
                'layers.0.swin1.attn.relative_position_bias_table',
                'layers.0.swin1.attn.relative_position_index',
                'layers.0.swin1.attn.qkv.weight',
                'layers.0.swin1.attn.proj.weight',
                'layers.0.swin1.attn.proj.bias',
                'layers.0.swin1.norm2.weight',
                'layers.0.swin1.mlp.fc1.weight',
                'layers.0.swin1.mlp.fc1.bias',
                'layers.0.swin1.mlp.fc2.weight',
                'layers.0.adjust1.weight',
                'layers.0.swin2.norm1.weight',
                'layers.0.adjust2.weight',
                'layers.0.swin3.norm1.weight',
                'layers.0.adjust3.weight',
                'layers.0.swin4.norm1.weight',
                'layers.0.adjust4.weight',
                'layers.0.swin5.norm1.weight',
                'layers.0.adjust5.weight',
                'norm.weight',
                'norm.bias',
            ),
        )

    def load(self, state_dict: Mapping[str, object]) -> DRCT:
        sd = state_dict
        in_chans = sd['conv_first.weight'].shape[1]
        embed_dim = sd['conv_first.weight'].shape[0]
        num_layers = get_seq_len(sd, 'layers')
        num_heads = [sd[f'layers.{i}.swin1.attn.relative_position_bias_table'].shape[1] for i in range(num_layers)]
        mlp_ratio = sd['layers.0.swin1.mlp.fc1.weight'].shape[0] / embed_dim
        window_size = (math.isqrt(sd['layers.0.swin1.attn.relative_position_bias_table'].shape[0]) + 1) // 2
        if 'conv_last.weight' in sd:
            upsampler = 'pixelshuffle'
            upscale, _ = get_pixelshuffle_params(sd, 'upsample')
        else:
            upsampler, upscale = '', 1
        resi_connection = '1conv' if 'conv_after_body.weight' in sd else 'identity'
        gc = sd['layers.0.adjust1.weight'].shape[0]
        if 'layers.0.swin2.attn_mask' in sd:
            img_size = math.isqrt(sd['layers.0.swin2.attn_mask'].shape[0]) * window_size
        else:
            img_size = window_size
        model = DRCT(img_size=img_size, patch_size=1, in_chans=in_chans, embed_dim=embed_dim, depths=(6,) * num_layers, num_heads=num_heads,
                     window_size=window_size, mlp_ratio=mlp_ratio, qkv_bias='layers.0.swin1.attn.qkv.bias' in sd, ape='absolute_pos_embed' in sd,
                     patch_norm='patch_embed.norm.weight' in sd, upscale=upscale, img_range=1.0, upsampler=upsampler,
                     resi_connection=resi_connection, gc=gc)  # fmt: skip
        return self._enhance_model(model=model, in_channels=in_chans, out_channels=in_chans, upscale=upscale, name='DRCT')
