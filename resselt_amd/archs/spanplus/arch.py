"""SPANPlus on the MI355X engine (reference module: ``resselt/archs/spanplus/arch.py:154-201``)."""

from __future__ import annotations

import ctypes as C

import torch

from ...engine import lib as L
from ...engine import ops
from ...engine.base import EngineModule, Plan
from ...engine.paramtree import build_param_tree
from ...engine.spanblocks import SPAN_MIXED, SpabChain, conv3xc_shapes, pack_span_family, spab_shapes

_GROUPS = 4  # DySample groups, fixed by the reference (utilities/dysample.py:17)


def dysample_init_pos(scale: int, groups: int = _GROUPS) -> torch.Tensor:
    """The reference's registered buffer (utilities/dysample.py:43-45): sub-pixel centre offsets per group."""
    h = torch.arange((-scale + 1) / 2, (scale - 1) / 2 + 1) / scale
    return torch.stack(torch.meshgrid([h, h], indexing='ij')).transpose(1, 2).repeat(1, groups, 1).reshape(1, -1, 1, 1)


class SpanPlus(EngineModule):
    # 'mixed' (what 'auto' selects; engine/spanblocks.py::SPAN_MIXED): the re-parameterised 3x3 convolutions in ONE fp16 product on hi planes,
    # conv_cat and the upsampler head in three fp16 products on hi + lo planes.  'fp16' = one product everywhere (2e-4: a benchmark mode).
    auto_precision = 'mixed'
    precisions = ('bf16x3', 'bf16', 'fp16', 'mixed')
    precision_table = SPAN_MIXED
    def __init__(self, num_in_ch: int = 3, num_out_ch: int = 3, blocks=(4,), feature_channels: int = 48, upscale: int = 4,
                 drop_rate: float = 0.0, upsampler: str = 'dys') -> None:  # fmt: skip
        super().__init__()
        if not isinstance(blocks, (list, tuple)):
            blocks = [int(blocks)]
        if feature_channels % 8 or (upsampler == 'dys' and feature_channels % (4 * _GROUPS)):
            raise NotImplementedError('feature_channels must be a multiple of 8 (16 with the dys upsampler)')
        if upsampler not in ('ps', 'dys', 'conv'):
            raise NotImplementedError(f'upsampler: {upsampler} not supported, choose one of ["ps", "dys", "conv"]')
        if upsampler == 'conv' and upscale != 1:
            raise ValueError('conv supports only 1x')
        self.in_ch = num_in_ch
        self.out_ch = num_out_ch if upsampler == 'dys' else num_in_ch
        self.blocks = list(blocks)
        self.fc = feature_channels
        self.upscale = upscale
        self.upsampler_kind = upsampler
        shapes: dict = {}
        fc = feature_channels
        conv3xc_shapes(shapes, 'feats.0', fc, num_in_ch)
        for bi, nblk in enumerate(self.blocks):
            p = f'feats.{bi + 1}'
            spab_shapes(shapes, f'{p}.block_1', fc)
            for j in range(nblk):
                spab_shapes(shapes, f'{p}.block_n.{j}', fc)
            spab_shapes(shapes, f'{p}.block_end', fc)
            conv3xc_shapes(shapes, f'{p}.conv_2', fc, fc)
            shapes[f'{p}.conv_cat.weight'] = (fc, fc * 4, 1, 1)
            shapes[f'{p}.conv_cat.bias'] = (fc,)
        buffers = {}
        if upsampler == 'ps':
            shapes['upsampler.0.weight'] = (self.out_ch * upscale * upscale, fc, 3, 3)
            shapes['upsampler.0.bias'] = (self.out_ch * upscale * upscale,)
        elif upsampler == 'conv':
            shapes['upsampler.weight'] = (self.out_ch, fc, 3, 3)
            shapes['upsampler.bias'] = (self.out_ch,)
        else:
            oc = 2 * _GROUPS * upscale * upscale
            shapes['upsampler.end_conv.weight'] = (self.out_ch, fc, 1, 1)
            shapes['upsampler.end_conv.bias'] = (self.out_ch,)
            shapes['upsampler.offset.weight'] = (oc, fc, 1, 1)
            shapes['upsampler.offset.bias'] = (oc,)
            shapes['upsampler.scope.weight'] = (oc, fc, 1, 1)
            buffers['upsampler.init_pos'] = dysample_init_pos(upscale)
        build_param_tree(self, shapes, buffers)

    # ---------------------------------------------------------------- weights
    def _conv3xc_names(self) -> list[str]:
        names = ['feats.0']
        for bi, nblk in enumerate(self.blocks):
            p = f'feats.{bi + 1}'
            for blk in [f'{p}.block_1', *[f'{p}.block_n.{j}' for j in range(nblk)], f'{p}.block_end']:
                names += [f'{blk}.{r}' for r in ('c1_r', 'c2_r', 'c3_r')]
            names.append(f'{p}.conv_2')
        return names

    def _pack(self, device, products):
        plain = [f'feats.{bi + 1}.conv_cat' for bi in range(len(self.blocks))]
        if self.upsampler_kind == 'ps':
            plain.append('upsampler.0')
        elif self.upsampler_kind == 'conv':
            plain.append('upsampler')
        W = pack_span_family(self, device, products, self._conv3xc_names(), plain)
        if self.upsampler_kind == 'dys':
            sd = {k: v.detach().to(device=device, dtype=torch.float32) for k, v in self.state_dict().items()}
            # offset (with bias) and scope (no bias) 1x1 convs as ONE k1 convolution: channels [0,oc) | [oc,2oc)
            w = torch.cat([sd['upsampler.offset.weight'], sd['upsampler.scope.weight']], 0)
            b = torch.cat([sd['upsampler.offset.bias'], torch.zeros_like(sd['upsampler.offset.bias'])], 0)
            head = dict(products=3, fmt=1) if products.name == 'mixed' else dict(products=products)  # the head reads hi + lo planes (spanblocks.SPAN_MIXED)
            W['upsampler.offscope'] = ops.ConvWeights.from_oihw(w, b, device=device, **head)
            end_w = sd['upsampler.end_conv.weight'].reshape(self.out_ch, self.fc)
            if self.out_ch <= 4:
                # bilinear sampling is linear: the 1x1 end conv is applied per channel group BEFORE the sampling, at low resolution
                # (rsa_dysample's pre-projected mode): z[4g + o] = sum over the channels c of group g of W_end[o][c] * x[c]
                cpg = self.fc // _GROUPS
                wz = torch.zeros((4 * _GROUPS, self.fc), dtype=torch.float32, device=device)
                for g in range(_GROUPS):
                    wz[4 * g : 4 * g + self.out_ch, g * cpg : (g + 1) * cpg] = end_w[:, g * cpg : (g + 1) * cpg]
                W['upsampler.zproj'] = ops.ConvWeights.from_oihw(wz[:, :, None, None], None, device=device, **head)
            W['dys'] = dict(
                init_pos=sd['upsampler.init_pos'].reshape(-1).contiguous(),
                end_w=sd['upsampler.end_conv.weight'].reshape(self.out_ch, self.fc).contiguous(),
                end_b=sd['upsampler.end_conv.bias'].contiguous(),
            )
        return W

    def macs_per_input_pixel(self) -> int:
        fc, s = self.fc, self.upscale
        n_c3 = 1 + sum(3 * (nb + 2) + 1 for nb in self.blocks)  # Conv3XC count; first one has in_ch inputs
        macs = 9 * fc * self.in_ch + (n_c3 - 1) * 9 * fc * fc + len(self.blocks) * 4 * fc * fc
        if self.upsampler_kind == 'ps':
            macs += 9 * fc * self.out_ch * s * s
        elif self.upsampler_kind == 'dys':
            macs += 2 * (2 * _GROUPS * s * s) * fc + s * s * fc * (4 + self.out_ch)
        return macs

    # ---------------------------------------------------------------- plan
    def _build_plan(self, plan: Plan, W, x_shape, dtype, products):
        n, c, h, w = x_shape
        if c != self.in_ch:
            raise RuntimeError(f'model expects {self.in_ch} input channels, got {c}')
        fc, pf, s = self.fc, self.fc // 8, self.upscale
        with_lo = products == 3
        wide = with_lo or products.name == 'mixed'  # buffers read by a three-product layer (conv_cat, the head) keep hi + lo
        # The first convolution (3 -> fc) takes the ring schedule when its input has a whole half chunk of 16 channels: a second, all-zero plane
        # (zeroed once, here) beside the image's plane, weights padded to match (`_pack`).  The chunk-barrier kernel ran it at 0.36 ms per
        # 2 Mpx (one fill in flight per CU); fp16 three-product form only (the ring's half mode in three products exists for fp16 planes).
        ring_first = W['feats.0'].cin_planes == 2
        x_pl = plan.planes(n, 2 if ring_first else (c + 7) // 8, h, w, wide)
        if ring_first:
            x_pl.hi.zero_()
            if x_pl.lo is not None:
                x_pl.lo.zero_()
        chain = SpabChain(plan, W, n, h, w, fc, L.ACT_MISH, with_lo, cat_lo=wide)
        preproj = self.upsampler_kind == 'dys' and 'upsampler.zproj' in W
        need_f32_feat = self.upsampler_kind == 'dys' and not preproj

        def set_input(x):
            ops.nchw_to_planes(x, x_pl)

        # feats.0 -> slot 0 of the first cat buffer (or straight to the upsampler input when there is no SPABS)
        nb = len(self.blocks)
        feat = plan.planes(n, pf, h, w, wide)
        feat_f32 = plan.f32map(n, fc, h, w) if need_f32_feat else None
        # (the f32 copies of a SPABS input exist only where the gate's shortcut is read from an f32 map: plain-bf16 mode)
        xf = [None if chain.plane_shortcut else plan.f32map(n, fc, h, w) for _ in range(2)]
        if nb == 0:
            plan.conv(ops.conv_params(W['feats.0'], x_pl, h, w, out=feat, out_f32=feat_f32))
        else:
            cat = chain.new_cat()
            plan.conv(ops.conv_params(W['feats.0'], x_pl, h, w, out=cat, out_plane_off=0, out_f32=xf[0]))
            for bi, nblk in enumerate(self.blocks):
                p = f'feats.{bi + 1}'
                names = dict(first=f'{p}.block_1', middle=[f'{p}.block_n.{j}' for j in range(nblk)], end=f'{p}.block_end',
                             conv_2=f'{p}.conv_2', conv_cat=f'{p}.conv_cat')  # fmt: skip
                if bi + 1 < nb:
                    nxt = chain.new_cat()
                    chain.run(names, cat, xf[bi & 1], nxt, 0, xf[(bi + 1) & 1])
                    cat = nxt
                else:
                    chain.run(names, cat, xf[bi & 1], feat, 0, feat_f32)

        out_shape = (n, self.out_ch, h * s, w * s)
        out_buf = {'y': torch.empty(out_shape, dtype=dtype, device=plan.device)}
        if self.upsampler_kind in ('ps', 'conv'):
            key = 'upsampler.0' if self.upsampler_kind == 'ps' else 'upsampler'
            plan.conv(ops.conv_params(W[key], feat, h, w, out_nchw=out_buf['y'], pixel_shuffle=s))
            arr = plan.flush()
            last = arr[len(arr) - 1]

            def prepare_output():
                if 'y' not in out_buf:
                    out_buf['y'] = torch.empty(out_shape, dtype=dtype, device=plan.device)
                last.out_nchw = out_buf['y'].data_ptr()

            plan.steps.insert(len(plan.steps) - 1, prepare_output)
        else:
            oc = 2 * _GROUPS * s * s
            offscope = plan.f32map(n, 2 * oc, h, w)
            plan.conv(ops.conv_params(W['upsampler.offscope'], feat, h, w, out_f32=offscope))
            plan.flush()
            d = W['dys']
            dp = L.DySampleParams()
            if preproj:
                z = plan.f32map(n, 4 * _GROUPS, h, w)
                plan.conv(ops.conv_params(W['upsampler.zproj'], feat, h, w, out_f32=z))
                plan.flush()
            dp.batch, dp.H, dp.W, dp.C, dp.groups, dp.scale, dp.out_ch = n, h, w, (4 * _GROUPS if preproj else fc), _GROUPS, s, self.out_ch
            dp.x_f32, dp.offscope = (z if preproj else feat_f32).data_ptr(), offscope.data_ptr()
            dp.init_pos, dp.end_w, dp.end_b = d['init_pos'].data_ptr(), (None if preproj else d['end_w'].data_ptr()), d['end_b'].data_ptr()
            dp.out_dtype = ops.rsa_dtype(dtype)
            lib = L.load()
            dev = plan.device

            def run_dysample():
                if 'y' not in out_buf:
                    out_buf['y'] = torch.empty(out_shape, dtype=dtype, device=dev)
                dp.out_nchw = out_buf['y'].data_ptr()
                L.check(lib.rsa_dysample(C.byref(dp), C.c_void_p(ops.current_stream_ptr(dev))), 'rsa_dysample')

            plan.call(run_dysample)
            plan.count_launches(1)

        def get_output():
            return out_buf.pop('y')

        return set_input, get_output
