"""DAT loader (drop-in for ``resselt/archs/dat/__init__.py:10-105``)."""

from __future__ import annotations

import math
from typing import Mapping

from ...factory import Architecture, KeyCondition
from ...utilities.state_dict import get_seq_len, pixelshuffle_scale
from .arch import DAT


class DatArch(Architecture[DAT]):
    def __init__(self):
        super().__init__(
            uid='dat',
            detect=KeyCondition.has_all(
                'conv_first.weight',
                'before_RG.1.weight',
                'before_RG.1.bias',
                'layers.0.blocks.0.norm1.weight',
                'layers.0.blocks.0.norm2.weight',
                'layers.0.blocks.0.ffn.fc1.weight',
                'layers.0.blocks.0.ffn.sg.norm.weight',
                'layers.0.blocks.0.ffn.sg.conv.weight',
                'layers.0.blocks.0.ffn.fc2.weight',
                'layers.0.blocks.0.attn.qkv.weight',
                'layers.0.blocks.0.attn.proj.weight',
                'layers.0.blocks.0.attn.dwconv.0.weight',
                'layers.0.blocks.0.attn.dwconv.1.running_mean',
                'layers.0.blocks.0.attn.channel_interaction.1.weight',
                'layers.0.blocks.0.attn.channel_interaction.2.running_mean',
                'layers.0.blocks.0.attn.channel_interaction.4.weight',
                'layers.0.blocks.0.attn.spatial_interaction.0.weight',
                'layers.0.blocks.0.attn.spatial_interaction.1.running_mean',
                'layers.0.blocks.0.attn.spatial_interaction.3.weight',
                'layers.0.blocks.0.attn.attns.0.rpe_biases',
                'layers.0.blocks.0.attn.attns.0.relative_position_index',
                'layers.0.blocks.0.attn.attns.0.pos.pos_proj.weight',
                'layers.0.blocks.0.attn.attns.0.pos.pos1.0.weight',
                'layers.0.blocks.0.attn.attns.0.pos.pos3.0.weight',
                'norm.weight',
            ),
        )

    def load(self, state_dict: Mapping[str, object]) -> DAT:
        sd = state_dict
        in_chans = sd['conv_first.weight'].shape[1]
        embed_dim = sd['conv_first.weight'].shape[0]
        num_layers = get_seq_len(sd, 'layers')
        depth = [get_seq_len(sd, f'layers.{i}.blocks') for i in range(num_layers)]
        num_heads = []
        for i in range(num_layers):
            if depth[i] >= 2:
                num_heads.append(sd[f'layers.{i}.blocks.1.attn.temperature'].shape[0])
            else:  # only even head counts can be reconstructed from a lone spatial block (dat/__init__.py:63-66)
                num_heads.append(sd[f'layers.{i}.blocks.0.attn.attns.0.pos.pos3.2.weight'].shape[0] * 2)
        upsampler = 'pixelshuffle' if 'conv_last.weight' in sd else 'pixelshuffledirect'
        resi_connection = '1conv' if 'conv_after_body.weight' in sd else '3conv'
        upscale = 2
        if upsampler == 'pixelshuffle':
            upscale = 1
            for i in range(0, get_seq_len(sd, 'upsample'), 2):
                w = sd[f'upsample.{i}.weight']
                upscale *= int(math.sqrt(w.shape[0] // w.shape[1]))
        else:
            upscale = pixelshuffle_scale(sd['upsample.0.weight'].shape[0], in_chans)
        qkv_bias = 'layers.0.blocks.0.attn.qkv.bias' in sd
        expansion_factor = float(sd['layers.0.blocks.0.ffn.fc1.weight'].shape[0] / embed_dim)
        img_size = 64  # cannot be deduced from the state dict in general
        if 'layers.0.blocks.2.attn.attn_mask_0' in sd:
            nw, n, _ = sd['layers.0.blocks.2.attn.attn_mask_0'].shape
            img_size = int(math.sqrt(nw * n))
        split_size = [2, 4]
        if 'layers.0.blocks.0.attn.attns.0.rpe_biases' in sd:
            split_size = [int(v) for v in (sd['layers.0.blocks.0.attn.attns.0.rpe_biases'][-1] + 1)]
        model = DAT(img_size=img_size, in_chans=in_chans, embed_dim=embed_dim, split_size=split_size, depth=depth, num_heads=num_heads,
                    expansion_factor=expansion_factor, qkv_bias=qkv_bias, upscale=upscale, resi_connection=resi_connection,
                    upsampler=upsampler)  # fmt: skip
        return self._enhance_model(model=model, in_channels=in_chans, out_channels=in_chans, upscale=upscale, name='DAT')
