"""HAT loader (drop-in for ``resselt/archs/hat/__init__.py:51-215``)."""

from __future__ import annotations

import math
from typing import Mapping

from ...factory import Architecture, KeyCondition
from ...utilities.state_dict import get_pixelshuffle_params, get_seq_len
from .arch import HAT


def _get_overlap_ratio(window_size: int, with_overlap: int) -> float:
    """``with_overlap = int(window_size + window_size * ratio)`` does not define the ratio uniquely: prefer 'nice' values (hat/__init__.py:8-24)."""
    for ratio in [0, 1, 0.5, 0.25, 0.75, 0.1, 0.2, 0.3, 0.4, 0.6, 0.7, 0.8, 0.9]:
        if int(window_size + window_size * ratio) == with_overlap:
            return ratio
    return (with_overlap - window_size) / window_size + 0.01


def _inv_int_div(a: int, c: int) -> float:
    """A number ``b`` with ``a // b == c`` (hat/__init__.py:27-48)."""
    b = a / c
    if b.is_integer():
        return int(b)
    for cand in (math.ceil(b), math.floor(b), b, b - 0.01, b + 0.01):
        if c == a // cand:
            return cand
    raise ValueError(f'Could not find a number b such that a // b == c. a={a}, c={c}')


class HATArch(Architecture[HAT]):
    def __init__(self):
        super().__init__(
            uid='HAT',
            detect=KeyCondition.has_all(
                'relative_position_index_SA',
                'conv_first.weight',
                'layers.0.residual_group.blocks.0.norm1.weight',
                'layers.0.residual_group.blocks.0.conv_block.cab.0.weight',
                'layers.0.residual_group.blocks.0.conv_block.cab.2.weight',
                'layers.0.residual_group.blocks.0.conv_block.cab.3.attention.1.weight',
                'layers.0.residual_group.blocks.0.conv_block.cab.3.attention.3.weight',
                'layers.0.residual_group.blocks.0.mlp.fc1.bias',
                'layers.0.residual_group.blocks.0.mlp.fc2.weight',
                'layers.0.residual_group.overlap_attn.relative_position_bias_table',
                'layers.0.residual_group.overlap_attn.qkv.weight',
                'layers.0.residual_group.overlap_attn.proj.weight',
                'layers.0.residual_group.overlap_attn.mlp.fc1.weight',
                'layers.0.residual_group.overlap_attn.mlp.fc2.weight',
                'conv_last.weight',
            ),
        )

    def load(self, state_dict: Mapping[str, object]) -> HAT:
        sd = state_dict
        in_chans = sd['conv_first.weight'].shape[1]
        embed_dim = sd['conv_first.weight'].shape[0]
        num_feat = sd['conv_last.weight'].shape[1]
        upscale, _ = get_pixelshuffle_params(sd, 'upsample', num_feat)
        window_size = int(math.sqrt(sd['relative_position_index_SA'].shape[0]))
        overlap_ratio = _get_overlap_ratio(window_size, with_overlap=int(math.sqrt(sd['relative_position_index_OCA'].shape[1])))
        num_layers = get_seq_len(sd, 'layers')
        depths = [get_seq_len(sd, f'layers.{i}.residual_group.blocks') for i in range(num_layers)]
        num_heads = [sd[f'layers.{i}.residual_group.overlap_attn.relative_position_bias_table'].shape[1] for i in range(num_layers)]
        resi_connection = '1conv' if 'conv_after_body.weight' in sd else 'identity'
        compress_ratio = _inv_int_div(embed_dim, sd['layers.0.residual_group.blocks.0.conv_block.cab.0.weight'].shape[0])
        squeeze_factor = _inv_int_div(embed_dim, sd['layers.0.residual_group.blocks.0.conv_block.cab.3.attention.1.weight'].shape[0])
        qkv_bias = 'layers.0.residual_group.blocks.0.attn.qkv.bias' in sd
        patch_norm = 'patch_embed.norm.weight' in sd
        ape = 'absolute_pos_embed' in sd
        mlp_ratio = int(sd['layers.0.residual_group.blocks.0.mlp.fc1.weight'].shape[0]) / embed_dim
        img_size = 64
        if ape:
            img_size = int(math.sqrt(sd['absolute_pos_embed'].shape[1]))
        model = HAT(img_size=img_size, patch_size=1, in_chans=in_chans, embed_dim=embed_dim, depths=depths, num_heads=num_heads,
                    window_size=window_size, compress_ratio=compress_ratio, squeeze_factor=squeeze_factor, conv_scale=0.01,
                    overlap_ratio=overlap_ratio, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, ape=ape, patch_norm=patch_norm, upscale=upscale,
                    img_range=1.0, upsampler='pixelshuffle', resi_connection=resi_connection, num_feat=num_feat)  # fmt: skip
        return self._enhance_model(model=model, in_channels=in_chans, out_channels=in_chans, upscale=upscale, name='HAT')
