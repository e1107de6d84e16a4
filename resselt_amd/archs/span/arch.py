"""SPAN on the MI355X engine (reference module: ``resselt/archs/span/arch.py:183-250``)."""

from __future__ import annotations

import torch

from ...engine import lib as L
from ...engine import ops
from ...engine.base import EngineModule, Plan
from ...engine.paramtree import build_param_tree
from ...engine.spanblocks import SPAN_MIXED, SpabChain, conv3xc_shapes, pack_span_family, spab_shapes


class SPAN(EngineModule):
    # 'mixed' (what 'auto' selects; engine/spanblocks.py::SPAN_MIXED): the re-parameterised 3x3 convolutions in ONE fp16 product on hi planes,
    # conv_cat and the upsampler head in three fp16 products on hi + lo planes.  'fp16' = one product everywhere (2e-4: a benchmark mode).
    # SPAN multiplies its input by 255 (img_range): with activations that large the SPAB gates saturate at +-0.5 and no longer attenuate
    # what the one-product layers round off (measured 3.3e-4 * max|y| under 'mixed' on the x4 fixture, 1e-5 under 'bf16x3'), so 'auto' keeps
    # the three-product mode here; 'mixed' / 'fp16' remain available for callers that accept that error.
    auto_precision = 'bf16x3'
    precisions = ('bf16x3', 'bf16', 'fp16', 'mixed')
    precision_table = SPAN_MIXED
    hyperparameters = {}

    def __init__(self, *, num_in_ch: int, num_out_ch: int, feature_channels: int = 48, upscale: int = 4, norm: bool = True,
                 img_range: float = 255.0, rgb_mean=(0.4488, 0.4371, 0.4040)) -> None:  # fmt: skip
        super().__init__()
        if feature_channels % 8:
            raise NotImplementedError('feature_channels must be a multiple of 8')
        self.in_channels, self.out_channels = num_in_ch, num_out_ch
        self.fc, self.upscale = feature_channels, upscale
        self.img_range = img_range
        self.mean = torch.tensor(rgb_mean, dtype=torch.float32)
        self.is_norm = norm
        shapes: dict = {}
        fc = feature_channels
        conv3xc_shapes(shapes, 'conv_1', fc, num_in_ch)
        for i in range(1, 7):
            spab_shapes(shapes, f'block_{i}', fc)
        shapes['conv_cat.weight'] = (fc, fc * 4, 1, 1)
        shapes['conv_cat.bias'] = (fc,)
        conv3xc_shapes(shapes, 'conv_2', fc, fc)
        shapes['upsampler.0.weight'] = (num_out_ch * upscale * upscale, fc, 3, 3)
        shapes['upsampler.0.bias'] = (num_out_ch * upscale * upscale,)
        build_param_tree(self, shapes, {} if norm else {'no_norm': torch.zeros(1)})

    def _pack(self, device, products):
        names = ['conv_1', 'conv_2'] + [f'block_{i}.{r}' for i in range(1, 7) for r in ('c1_r', 'c2_r', 'c3_r')]
        W = pack_span_family(self, device, products, names, ['conv_cat', 'upsampler.0'])
        W['mean'] = self.mean.to(device)
        return W

    def macs_per_input_pixel(self) -> int:
        fc, s = self.fc, self.upscale
        return 9 * fc * self.in_channels + 19 * 9 * fc * fc + 4 * fc * fc + 9 * fc * self.out_channels * s * s

    def _build_plan(self, plan: Plan, W, x_shape, dtype, products):
        n, c, h, w = x_shape
        if c != self.in_channels:
            raise RuntimeError(f'model expects {self.in_channels} input channels, got {c}')
        if self.is_norm and c != self.mean.numel():
            raise RuntimeError('SPAN input normalisation needs a 3-channel input')
        fc, pf, s = self.fc, self.fc // 8, self.upscale
        with_lo = products == 3
        wide = with_lo or products.name == 'mixed'  # buffers read by a three-product layer (conv_cat, the head) keep hi + lo
        ring_first = W['conv_1'].cin_planes == 2  # a second, all-zero input plane: the first convolution takes the ring schedule (see SpanPlus)
        x_pl = plan.planes(n, 2 if ring_first else (c + 7) // 8, h, w, wide)
        if ring_first:
            x_pl.hi.zero_()
            if x_pl.lo is not None:
                x_pl.lo.zero_()
        chain = SpabChain(plan, W, n, h, w, fc, L.ACT_SILU, with_lo, cat_lo=wide)
        mean = W['mean'] if self.is_norm else None
        scale = self.img_range if self.is_norm else 1.0

        def set_input(x):
            # (x - mean) * img_range (span/arch.py:232-234) fused into the layout conversion
            ops.nchw_to_planes(x, x_pl, mean, scale)

        cat = chain.new_cat()
        xf = None if chain.plane_shortcut else plan.f32map(n, fc, h, w)  # (the gate's shortcut as an f32 map: plain-bf16 mode only)
        feat = plan.planes(n, pf, h, w, wide)
        plan.conv(ops.conv_params(W['conv_1'], x_pl, h, w, out=cat, out_plane_off=0, out_f32=xf))
        names = dict(first='block_1', middle=[f'block_{i}' for i in range(2, 6)], end='block_6', conv_2='conv_2', conv_cat='conv_cat')
        chain.run(names, cat, xf, feat, 0, None)
        out_shape = (n, self.out_channels, h * s, w * s)
        out_buf = {'y': torch.empty(out_shape, dtype=dtype, device=plan.device)}
        plan.conv(ops.conv_params(W['upsampler.0'], feat, h, w, out_nchw=out_buf['y'], pixel_shuffle=s))
        arr = plan.flush()
        last = arr[len(arr) - 1]

        def prepare_output():
            if 'y' not in out_buf:
                out_buf['y'] = torch.empty(out_shape, dtype=dtype, device=plan.device)
            last.out_nchw = out_buf['y'].data_ptr()

        plan.steps.insert(len(plan.steps) - 1, prepare_output)

        def get_output():
            return out_buf.pop('y')

        return set_input, get_output
