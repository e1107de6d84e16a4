"""RTMoSR loader (drop-in for ``resselt/archs/rtmosr/__init__.py:8-192``)."""

from __future__ import annotations

import math
from typing import Mapping

from ...factory import Architecture, KeyCondition
from ...utilities.state_dict import get_seq_len
from ..spanpp import _REPCONV_KEYS
from .arch import RTMoSR


class RTMoSRArch(Architecture[RTMoSR]):
    def __init__(self):
        # the reference lists: the first block's norm, the RepConv keys of its fc1 / conv.0.poll.1 and of to_img.0, and its OmniShift
        keys = ['body.0.norm.scale', 'body.0.norm.offset']
        keys += [f'{p}.{k}' for p in ('body.0.fc1', 'body.0.conv.0.poll.1') for k in _REPCONV_KEYS]
        keys += [f'body.0.conv.1.alpha{i}' for i in (1, 2, 3, 4)]
        keys += [f'body.0.conv.1.{c}.{wb}' for c in ('conv1x1', 'conv3x3', 'conv5x5', 'conv5x5_reparam') for wb in ('weight', 'bias')]
        keys += [f'to_img.0.{k}' for k in _REPCONV_KEYS]
        super().__init__(uid='RTMoSR', detect=KeyCondition.has_all(*keys))

    def load(self, state: Mapping[str, object]) -> RTMoSR:
        unshuffle = False
        if 'to_feat.1.alpha' in state:
            unshuffle = True
            # (the reference reads the UNSHUFFLE factor here and passes it as the scale; only 2 -- scale 2, unshuffle 2 -- round-trips)
            scale = math.isqrt(state['to_feat.1.conv_3x3_rep.weight'].shape[1] // 3)
            dim = state['to_feat.1.conv_3x3_rep.weight'].shape[0]
        else:
            scale = math.isqrt(state['to_img.0.conv_3x3_rep.weight'].shape[0] // 3)
            dim = state['to_feat.conv_3x3_rep.weight'].shape[0]
        dccm = 'body.0.fc2.alpha' in state
        se = 'body.0.conv.2.squeezing.0.weight' in state
        ffn = state['body.0.fc1.conv_3x3_rep.weight'].shape[0] / dim / 2
        n_blocks = get_seq_len(state, 'body')
        model = RTMoSR(scale=scale, dim=dim, ffn_expansion=ffn, n_blocks=n_blocks, unshuffle_mod=unshuffle, dccm=dccm, se=se)
        # the reference reports upscale=2 for every RTMoSR checkpoint (rtmosr/__init__.py:192); kept for metadata parity
        return self._enhance_model(model=model, in_channels=3, out_channels=3, upscale=int(2), name='RTMoSR')
