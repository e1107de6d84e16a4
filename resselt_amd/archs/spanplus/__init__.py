"""SPANPlus loader (drop-in for ``resselt/archs/spanplus/__init__.py:8-38``)."""

from __future__ import annotations

from typing import Mapping

from ...factory import Architecture, KeyCondition
from ...utilities.state_dict import dysample_scale, get_seq_len, pixelshuffle_scale
from .arch import SpanPlus


class SpanPlusArch(Architecture[SpanPlus]):
    def __init__(self):
        super().__init__(uid='spanplus', detect=KeyCondition.has_all('feats.0.eval_conv.weight'))

    def load(self, state_dict: Mapping[str, object]) -> SpanPlus:
        n_feats = get_seq_len(state_dict, 'feats') - 1
        blocks = [get_seq_len(state_dict, f'feats.{i + 1}.block_n') for i in range(n_feats)]
        first = state_dict['feats.0.eval_conv.weight']
        num_in_ch, feature_channels = first.shape[1], first.shape[0]
        if 'upsampler.0.weight' in state_dict:
            upsampler, num_out_ch = 'ps', num_in_ch
            upscale = pixelshuffle_scale(state_dict['upsampler.0.weight'].shape[0], num_out_ch)
        else:
            upsampler = 'dys'
            num_out_ch = state_dict['upsampler.end_conv.weight'].shape[0]
            upscale = dysample_scale(state_dict['upsampler.offset.weight'].shape[0])
        model = SpanPlus(num_in_ch=num_in_ch, num_out_ch=num_out_ch, blocks=blocks, feature_channels=feature_channels, upscale=upscale,
                         upsampler=upsampler)  # fmt: skip
        return self._enhance_model(model=model, in_channels=num_in_ch, out_channels=num_out_ch, upscale=upscale, name='SPANPlus')
