"""SpanPP on the MI355X engine (reference module: ``resselt/archs/spanpp/arch.py:315-373``), eval-mode semantics.

Everything that depends on the weights only happens once at pack time: every RepConv (SeqConv3x3 + 3x3 + Conv3XC, alpha-weighted,
arch.py:152-193) becomes one 3x3 kernel, and the implicit-grid upsampler (IGConv, arch.py:244-312) evaluates its Fourier-feature MLP into
one [3*s*s, C, 3, 3] kernel per scale of ``scale_list``.  The forward is then the SPAN launch list: 22 fused convolutions, the last one
storing through depth-to-space.  ``model(x, scale)`` picks the head; ``scale=None`` means ``eval_base_scale`` (2), as in the reference.

The reference module only runs after ``.eval()`` (``IGConv.forward`` reads a table that ``.train(mode)`` fills); the engine needs no mode switch.
"""

from __future__ import annotations

import math

import torch

from ...engine import lib as L
from ...engine import ops
from ...engine.base import EngineModule, Plan
from ...engine.paramtree import build_param_tree
from ...engine.base import check_fp16_range
from ...engine.spanblocks import SPAN_FIRST, SPAN_MIXED, SpabChain, conv3xc_shapes, fold_conv3xc, span_layer_policy


def repconv_shapes(shapes: dict, name: str, cout: int, cin: int) -> None:
    mid = 2 * cout
    shapes[f'{name}.alpha'] = (3,)
    shapes[f'{name}.conv1.k0'] = (mid, cin, 1, 1)
    shapes[f'{name}.conv1.b0'] = (mid,)
    shapes[f'{name}.conv1.k1'] = (cout, mid, 3, 3)
    shapes[f'{name}.conv1.b1'] = (cout,)
    shapes[f'{name}.conv2.weight'] = (cout, cin, 3, 3)
    shapes[f'{name}.conv2.bias'] = (cout,)
    conv3xc_shapes(shapes, f'{name}.conv3', cout, cin)
    shapes[f'{name}.conv_3x3_rep.weight'] = (cout, cin, 3, 3)
    shapes[f'{name}.conv_3x3_rep.bias'] = (cout,)


def fold_repconv(sd: dict, name: str) -> tuple[torch.Tensor, torch.Tensor]:
    """RepConv.fuse (arch.py:166-176) in f64: alpha0 * (k1 o k0) + alpha1 * conv2 + alpha2 * fold(Conv3XC)."""
    d = torch.float64
    a = sd[f'{name}.alpha'].to(d)
    k0 = sd[f'{name}.conv1.k0'].to(d)[:, :, 0, 0]
    b0 = sd[f'{name}.conv1.b0'].to(d)
    k1 = sd[f'{name}.conv1.k1'].to(d)
    w1 = torch.einsum('omyx,mi->oiyx', k1, k0)  # SeqConv3x3.rep_params (arch.py:138-150)
    b1 = torch.einsum('omyx,m->o', k1, b0) + sd[f'{name}.conv1.b1'].to(d)
    w3, b3 = fold_conv3xc(sd, f'{name}.conv3')
    w = a[0] * w1 + a[1] * sd[f'{name}.conv2.weight'].to(d) + a[2] * w3.to(d)
    b = a[0] * b1 + a[1] * sd[f'{name}.conv2.bias'].to(d) + a[2] * b3.to(d)
    return w.to(torch.float32), b.to(torch.float32)


def igconv_kernel(sd: dict, scale: int, max_scale: int) -> torch.Tensor:
    """IGConv._implicit_representation_latent (arch.py:293-312): the scale-s head as a [3*s*s, C, 3, 3] kernel."""
    F = torch.nn.functional
    freq, amp = sd['upsampler.freq'].float(), sd['upsampler.amplitude'].float()
    dev = freq.device
    n = freq.shape[0]
    r = torch.ones(1, 1, scale, scale, device=dev) / min(scale, max_scale) * 2
    seq = -1 + (1 / scale) + (2 / scale) * torch.arange(scale, device=dev).float()
    coords = torch.stack(torch.meshgrid(seq, seq, indexing='ij'), dim=-1).flip(-1).unsqueeze(0).permute(0, 3, 1, 2)
    f1, f2 = freq.repeat(1, 1, scale, scale).chunk(2, dim=1)
    f = f1 * coords[:, :1] + f2 * coords[:, 1:] + F.conv2d(r, sd['upsampler.phase.weight'].float(), sd['upsampler.phase.bias'].float())
    x = torch.cat([torch.cos(math.pi * f), torch.sin(math.pi * f)], dim=1) * amp.repeat(1, 1, scale, scale)
    i = 0
    while f'upsampler.query_kernel.{i}.weight' in sd:
        x = F.conv2d(x, sd[f'upsampler.query_kernel.{i}.weight'].float(), sd[f'upsampler.query_kernel.{i}.bias'].float())
        if f'upsampler.query_kernel.{i + 2}.weight' in sd:
            x = F.relu(x)
        i += 2
    c = n // 9
    return x.reshape(c, 3, 3, 3, scale, scale).permute(3, 4, 5, 0, 1, 2).reshape(3 * scale * scale, c, 3, 3).contiguous()


class SpanPP(EngineModule):
    hyperparameters = {}

    def __init__(self, *, num_in_ch=3, feature_channels=48, scale_list=(1, 2, 3, 4), eval_base_scale=2, ig_kernel_size=3, implicit_dim=256,
                 latent_layers=4, **kwargs) -> None:  # fmt: skip
        super().__init__()
        if feature_channels % 8:
            raise NotImplementedError('feature_channels must be a multiple of 8')
        if ig_kernel_size != 3 or implicit_dim % 2:
            raise NotImplementedError('the implicit upsampler is built for 3x3 kernels and an even implicit_dim')
        self.in_channels, self.fc = num_in_ch, feature_channels
        self.scale_list = sorted(set(int(s) for s in scale_list))
        self.base_scale = eval_base_scale
        self._scale = eval_base_scale
        shapes: dict = {}
        fc = feature_channels
        self._rep_names = ['conv0', 'conv_2'] + [f'block_{i}.{r}' for i in range(1, 7) for r in ('c1_r', 'c2_r', 'c3_r')]
        repconv_shapes(shapes, 'conv0', fc, num_in_ch)
        for i in range(1, 7):
            for r in ('c1_r', 'c2_r', 'c3_r'):
                repconv_shapes(shapes, f'block_{i}.{r}', fc, fc)
        shapes['conv_cat.weight'] = (fc, fc * 4, 1, 1)
        shapes['conv_cat.bias'] = (fc,)
        repconv_shapes(shapes, 'conv_2', fc, fc)
        shapes['upsampler.freq'] = (fc * 9, implicit_dim, 1, 1)
        shapes['upsampler.amplitude'] = (fc * 9, implicit_dim, 1, 1)
        shapes['upsampler.phase.weight'] = (implicit_dim // 2, 1, 1, 1)
        shapes['upsampler.phase.bias'] = (implicit_dim // 2,)
        for l in range(latent_layers):
            shapes[f'upsampler.query_kernel.{2 * l}.weight'] = (implicit_dim, implicit_dim, 1, 1)
            shapes[f'upsampler.query_kernel.{2 * l}.bias'] = (implicit_dim,)
        shapes[f'upsampler.query_kernel.{2 * latent_layers}.weight'] = (3, implicit_dim, 1, 1)
        shapes[f'upsampler.query_kernel.{2 * latent_layers}.bias'] = (3,)
        build_param_tree(self, shapes, {'MetaIGConv': torch.tensor(self.scale_list, dtype=torch.uint8)})

    def _convert_state_dict(self, state_dict):
        state_dict['MetaIGConv'] = self.MetaIGConv  # the reference injects its own buffer the same way (arch.py:354-356)
        return state_dict

    # 'mixed' (what 'auto' selects; engine/spanblocks.py::SPAN_MIXED, as SPANPlus): the re-parameterised 3x3 convolutions behind a SPAB gate in ONE
    # fp16 product on hi planes; the first convolution, conv_cat and the implicit upsampler's 3x3 kernel in three fp16 products.
    auto_precision = 'mixed'
    precisions = ('bf16x3', 'bf16', 'fp16', 'mixed')
    precision_table = SPAN_MIXED

    def _pack(self, device, products):
        sd = {k: v.detach().cpu() for k, v in self.state_dict().items()}  # folds and the IGConv kernel are host-side weight preprocessing (f64 / f32)
        fsd = {k: (v.to(torch.float32) if v.is_floating_point() else v) for k, v in sd.items()}
        mixed = products.name == 'mixed'

        def policy(name, conv3xc):
            return span_layer_policy(name, conv3xc) if mixed else (int(products), products.fmt)

        W = {}
        for name in self._rep_names:
            w, b = fold_repconv(fsd, name)
            prod, fmt = policy(name, True)
            # (the first convolution is packed over a whole 16-channel half chunk: its input carries a second, all-zero plane -> ring schedule)
            W[name] = ops.ConvWeights.from_oihw(w, b, prod, device=device, fmt=fmt, cin_planes=2 if name in SPAN_FIRST and w.shape[1] <= 8 else None)
        prod, fmt = policy('conv_cat', False)
        W['conv_cat'] = ops.ConvWeights.from_oihw(fsd['conv_cat.weight'], fsd['conv_cat.bias'], prod, device=device, fmt=fmt)
        for s in self.scale_list:
            W[f'up{s}'] = ops.ConvWeights.from_oihw(igconv_kernel(fsd, s, max(self.scale_list)), None, prod, device=device, fmt=fmt)
        check_fp16_range(W.values())
        return W

    def macs_per_input_pixel(self) -> int:
        fc, s = self.fc, self._scale
        return 9 * fc * self.in_channels + 19 * 9 * fc * fc + 4 * fc * fc + 9 * fc * 3 * s * s

    def forward(self, x: torch.Tensor, scale: int | None = None) -> torch.Tensor:
        s = self.base_scale if scale is None else int(scale)
        if s not in self.scale_list:
            raise KeyError(str(s))  # the reference's eval_convs lookup raises the same for a scale outside scale_list
        if s != self._scale:
            self._scale = s
            for key in list(self._plans):
                self._drop_plan(key)
        return super().forward(x)

    def _build_plan(self, plan: Plan, W, x_shape, dtype, products):
        n, c, h, w = x_shape
        if c != self.in_channels:
            raise RuntimeError(f'model expects {self.in_channels} input channels, got {c}')
        fc, pf, s = self.fc, self.fc // 8, self._scale
        with_lo = products == 3
        wide = with_lo or products.name == 'mixed'  # buffers read by a three-product layer (conv_cat, the head) keep hi + lo
        ring_first = W['conv0'].cin_planes == 2  # a second, all-zero input plane: the first convolution takes the ring schedule (see SpanPlus)
        x_pl = plan.planes(n, 2 if ring_first else (c + 7) // 8, h, w, wide)
        if ring_first:
            x_pl.hi.zero_()
            if x_pl.lo is not None:
                x_pl.lo.zero_()
        chain = SpabChain(plan, W, n, h, w, fc, L.ACT_SILU, with_lo, cat_lo=wide)

        def set_input(x):
            ops.nchw_to_planes(x, x_pl)

        cat = chain.new_cat()
        xf = None if chain.plane_shortcut else plan.f32map(n, fc, h, w)  # (the gate's shortcut as an f32 map: plain-bf16 mode only)
        feat = plan.planes(n, pf, h, w, wide)
        plan.conv(ops.conv_params(W['conv0'], x_pl, h, w, out=cat, out_plane_off=0, out_f32=xf))
        names = dict(first='block_1', middle=[f'block_{i}' for i in range(2, 6)], end='block_6', conv_2='conv_2', conv_cat='conv_cat')
        chain.run(names, cat, xf, feat, 0, None)
        out_shape = (n, 3, h * s, w * s)
        out_buf = {'y': torch.empty(out_shape, dtype=dtype, device=plan.device)}
        plan.conv(ops.conv_params(W[f'up{s}'], feat, h, w, out_nchw=out_buf['y'], pixel_shuffle=s))
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
