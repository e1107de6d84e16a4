"""SRVGGNetCompact (Real-ESRGAN "animevideo"/"general" models) on the MI355X engine.

Reference module: ``resselt/archs/compact/arch.py:5-65``: conv3x3 + per-channel PReLU chain at input resolution, a last
conv to ``out_ch * s^2`` channels, PixelShuffle and the nearest-upsampled input added back.  Here: ``num_conv + 2`` fused
convolutions; PReLU is an epilogue with a slope vector, PixelShuffle and the base-image add happen in the final store.
"""

from __future__ import annotations

import torch

from ...engine import lib as L
from ...engine import ops
from ...engine.base import EngineModule, Plan, check_fp16_range
from ...engine.paramtree import build_param_tree


class SRVGGNetCompact(EngineModule):
    # 'fp16' (what 'auto' selects): every convolution in ONE fp16 product on hi planes -- the reference's own fp16 inference arithmetic with an
    # f32 accumulator.  Pinned at <= half an fp16 ulp of the output + 1e-4 by tests/test_baseline_configs_gpu.py (C3) and test_span_gpu.py.
    auto_precision = 'fp16'
    precisions = ('bf16x3', 'bf16', 'fp16')

    def __init__(self, num_in_ch=3, num_out_ch=3, num_feat=64, num_conv=16, upscale=4, act_type='prelu'):
        super().__init__()
        if act_type != 'prelu':
            raise NotImplementedError('the Compact loader only builds PReLU models (compact/arch.py:42)')
        if num_feat % 8:
            raise NotImplementedError('num_feat must be a multiple of 8')
        self.num_in_ch, self.num_out_ch, self.num_feat, self.num_conv, self.upscale = num_in_ch, num_out_ch, num_feat, num_conv, upscale
        shapes: dict = {}
        cin = num_in_ch
        for i in range(num_conv + 1):
            shapes[f'body.{2 * i}.weight'] = (num_feat, cin, 3, 3)
            shapes[f'body.{2 * i}.bias'] = (num_feat,)
            shapes[f'body.{2 * i + 1}.weight'] = (num_feat,)  # nn.PReLU(num_parameters=num_feat)
            cin = num_feat
        last = 2 * (num_conv + 1)
        shapes[f'body.{last}.weight'] = (num_out_ch * upscale * upscale, num_feat, 3, 3)
        shapes[f'body.{last}.bias'] = (num_out_ch * upscale * upscale,)
        self._last = last
        build_param_tree(self, shapes)

    def _pack(self, device, products):
        sd = {k: v.detach().to(device=device, dtype=torch.float32) for k, v in self.state_dict().items()}
        W = {}
        for i in range(self.num_conv + 1):
            W[f'body.{2 * i}'] = ops.ConvWeights.from_oihw(sd[f'body.{2 * i}.weight'], sd[f'body.{2 * i}.bias'], products, device=device)
            W[f'slope.{2 * i + 1}'] = ops.pad_bias(sd[f'body.{2 * i + 1}.weight'], self.num_feat, device)
        W[f'body.{self._last}'] = ops.ConvWeights.from_oihw(sd[f'body.{self._last}.weight'], sd[f'body.{self._last}.bias'], products, device=device)
        check_fp16_range(W.values())
        return W

    def macs_per_input_pixel(self) -> int:
        f = self.num_feat
        return 9 * (self.num_in_ch * f + self.num_conv * f * f + f * self.num_out_ch * self.upscale**2)

    def _build_plan(self, plan: Plan, W, x_shape, dtype, products):
        n, c, h, w = x_shape
        if c != self.num_in_ch:
            raise RuntimeError(f'model expects {self.num_in_ch} input channels, got {c}')
        if self.num_out_ch != self.num_in_ch:
            raise RuntimeError('the residual base image needs num_out_ch == num_in_ch (compact/arch.py:61-64)')
        with_lo = products == 3
        pf, s = self.num_feat // 8, self.upscale
        x_pl = plan.planes(n, (c + 7) // 8, h, w, with_lo)
        bufs = [plan.planes(n, pf, h, w, with_lo) for _ in range(2)]
        holder = {}

        def set_input(x):
            holder['x'] = x  # the final store adds the (nearest-upsampled) input back
            ops.nchw_to_planes(x, x_pl)

        cur = x_pl
        for i in range(self.num_conv + 1):
            dst = bufs[i & 1]
            plan.conv(ops.conv_params(W[f'body.{2 * i}'], cur, h, w, act=L.ACT_PRELU, act_vec=W[f'slope.{2 * i + 1}'], out=dst))
            cur = dst
        out_shape = (n, self.num_out_ch, h * s, w * s)
        out_buf = {'y': torch.empty(out_shape, dtype=dtype, device=plan.device)}
        base0 = torch.empty((n, c, h, w), dtype=dtype, device=plan.device)  # placeholder pointer, patched per call
        plan.conv(ops.conv_params(W[f'body.{self._last}'], cur, h, w, out_nchw=out_buf['y'], pixel_shuffle=s, out_base=base0))
        arr = plan.flush()
        last = arr[len(arr) - 1]

        def prepare():
            if 'y' not in out_buf:
                out_buf['y'] = torch.empty(out_shape, dtype=dtype, device=plan.device)
            last.out_nchw = out_buf['y'].data_ptr()
            last.out_base = holder['x'].data_ptr()

        plan.steps.insert(len(plan.steps) - 1, prepare)

        def get_output():
            holder.clear()
            return out_buf.pop('y')

        return set_input, get_output
