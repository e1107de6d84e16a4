"""SpanPP loader (drop-in for ``resselt/archs/spanpp/__init__.py:8-129``)."""

from __future__ import annotations

from typing import Mapping

from ...factory import Architecture, KeyCondition
from ...utilities.state_dict import get_seq_len
from .arch import SpanPP

_REPCONV_KEYS = ('alpha', 'conv1.k0', 'conv1.b0', 'conv1.k1', 'conv1.b1', 'conv2.weight', 'conv2.bias', 'conv3.sk.weight', 'conv3.sk.bias',
                 'conv3.conv.0.weight', 'conv3.conv.0.bias', 'conv3.conv.1.weight', 'conv3.conv.1.bias', 'conv3.conv.2.weight',
                 'conv3.conv.2.bias', 'conv3.eval_conv.weight', 'conv3.eval_conv.bias', 'conv_3x3_rep.weight', 'conv_3x3_rep.bias')  # fmt: skip


class SpanPPArch(Architecture[SpanPP]):
    def __init__(self):
        # the reference lists the RepConv keys of conv0, block_1.c1_r/c2_r/c3_r and the head of block_2.c1_r one by one
        keys = [f'{p}.{k}' for p in ('conv0', 'block_1.c1_r', 'block_1.c2_r', 'block_1.c3_r') for k in _REPCONV_KEYS]
        keys += [f'block_2.c1_r.{k}' for k in _REPCONV_KEYS[:11]]
        super().__init__(uid='SpanPP', detect=KeyCondition.has_all(*keys))

    def load(self, state_dict: Mapping[str, object]) -> SpanPP:
        state = state_dict
        dim, in_ch = state['conv0.conv_3x3_rep.weight'].shape[:2]
        scales = state['MetaIGConv'].tolist() if 'MetaIGConv' in state else [1, 2, 3, 4]
        _ig_kernel, implicit_dim = state['upsampler.freq'].shape[:2]
        latent_layers = get_seq_len(state, 'upsampler.query_kernel') // 2
        # (the reference passes the kernel size under a name its constructor ignores, so the generated kernels are always 3x3)
        model = SpanPP(num_in_ch=in_ch, feature_channels=dim, scale_list=scales, eval_base_scale=2, implicit_dim=implicit_dim,
                       latent_layers=latent_layers)  # fmt: skip
        return self._enhance_model(model=model, in_channels=in_ch, out_channels=in_ch, upscale=scales, name='SpanPP')
