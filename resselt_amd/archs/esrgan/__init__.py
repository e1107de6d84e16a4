"""ESRGAN-family loader (drop-in for ``resselt/archs/esrgan/__init__.py:124-194``)."""

from __future__ import annotations

import math
from typing import Mapping

from ...factory import Architecture, KeyCondition
from ...utilities.state_dict import get_seq_len
from .arch import RRDBNet, new_arch_to_old


class ESRGANArch(Architecture[RRDBNet]):
    def __init__(self) -> None:
        super().__init__(
            uid='ESRGAN',
            detect=KeyCondition.has_any(
                # old arch (ESRGAN)
                KeyCondition.has_all('model.0.weight', 'model.1.sub.0.RDB1.conv1.0.weight'),
                # new arch (Real-ESRGAN)
                KeyCondition.has_all('conv_first.weight', 'body.0.rdb1.conv1.weight', 'conv_body.weight', 'conv_last.weight'),
                # BSRGAN / RealSR
                KeyCondition.has_all('conv_first.weight', 'RRDB_trunk.0.RDB1.conv1.weight', 'trunk_conv.weight', 'conv_last.weight'),
                # ESRGAN+
                KeyCondition.has_all('model.0.weight', 'model.1.sub.0.RDB1.conv1x1.weight'),
            ),
        )

    def load(self, state_dict: Mapping[str, object]) -> RRDBNet:
        # hyper-parameter inference follows resselt/archs/esrgan/__init__.py:155-194
        sd = new_arch_to_old(state_dict)
        seq_len = get_seq_len(sd, 'model')
        in_nc = sd['model.0.weight'].shape[1]
        out_nc = sd[f'model.{seq_len - 1}.weight'].shape[0]
        scale = 2 ** ((seq_len - 5) // 3)
        num_blocks = get_seq_len(sd, 'model.1.sub') - 1
        num_filters = sd['model.0.weight'].shape[0]
        plus = any('.conv1x1.' in k for k in sd)
        shuffle_factor = int(math.sqrt(in_nc / out_nc)) if in_nc in (out_nc * 4, out_nc * 16) else None
        model = RRDBNet(in_nc=in_nc, out_nc=out_nc, num_filters=num_filters, num_blocks=num_blocks, scale=scale, plus=plus,
                        shuffle_factor=shuffle_factor)  # fmt: skip
        if shuffle_factor:
            in_nc //= shuffle_factor**2
            scale //= shuffle_factor
        return self._enhance_model(model=model, in_channels=in_nc, out_channels=out_nc, upscale=scale, name='ESRGAN')
