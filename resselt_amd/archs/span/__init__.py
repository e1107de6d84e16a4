"""SPAN loader (drop-in for ``resselt/archs/span/__init__.py:10-55``)."""

from __future__ import annotations

from typing import Mapping

import torch

from ...factory import Architecture, KeyCondition
from ...utilities.state_dict import pixelshuffle_scale
from .arch import SPAN


class SPANArch(Architecture[SPAN]):
    def __init__(self):
        super().__init__(
            uid='SPAN',
            detect=KeyCondition.has_all(
                'conv_1.sk.weight',
                'block_1.c1_r.sk.weight',
                'block_1.c1_r.eval_conv.weight',
                'block_1.c3_r.eval_conv.weight',
                'conv_cat.weight',
                'conv_2.sk.weight',
                'conv_2.eval_conv.weight',
                'upsampler.0.weight',
            ),
        )

    def load(self, state_dict: Mapping[str, object]) -> SPAN:
        # img_range and rgb_mean cannot be deduced from the checkpoint (span/__init__.py:28-29)
        num_in_ch = state_dict['conv_1.sk.weight'].shape[1]
        feature_channels = state_dict['conv_1.sk.weight'].shape[0]
        upscale = pixelshuffle_scale(state_dict['upsampler.0.weight'].shape[0], num_in_ch)
        norm = True
        if 'no_norm' in state_dict:
            norm = False
            state_dict['no_norm'] = torch.zeros(1)  # the reference normalises this marker in the caller's dict too (:41-43)
        model = SPAN(num_in_ch=num_in_ch, num_out_ch=num_in_ch, feature_channels=feature_channels, upscale=upscale, norm=norm,
                     img_range=255.0, rgb_mean=(0.4488, 0.4371, 0.4040))  # fmt: skip
        return self._enhance_model(model=model, in_channels=num_in_ch, out_channels=num_in_ch, upscale=upscale, name='SPAN')
