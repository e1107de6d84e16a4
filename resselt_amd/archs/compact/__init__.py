"""Compact (SRVGGNetCompact) loader (drop-in for ``resselt/archs/compact/__init__.py:8-38``)."""

from __future__ import annotations

from typing import Mapping

from ...factory import Architecture, KeyCondition
from ...utilities.state_dict import get_seq_len, pixelshuffle_scale
from .arch import SRVGGNetCompact


class CompactArch(Architecture[SRVGGNetCompact]):
    def __init__(self):
        super().__init__(uid='Compact', detect=KeyCondition.has_all('body.0.weight', 'body.1.weight'))

    def load(self, state_dict: Mapping[str, object]) -> SRVGGNetCompact:
        highest = get_seq_len(state_dict, 'body') - 1
        in_nc = state_dict['body.0.weight'].shape[1]
        num_feat = state_dict['body.0.weight'].shape[0]
        num_conv = (highest - 2) // 2
        scale = pixelshuffle_scale(state_dict[f'body.{highest}.bias'].shape[0], in_nc)
        model = SRVGGNetCompact(num_in_ch=in_nc, num_out_ch=in_nc, num_feat=num_feat, num_conv=num_conv, upscale=scale)
        return self._enhance_model(model=model, in_channels=in_nc, out_channels=in_nc, upscale=scale, name='Compact')
