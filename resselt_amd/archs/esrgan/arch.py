"""RRDBNet (ESRGAN / BSRGAN / Real-ESRGAN) on the MI355X engine.

Reference module: ``resselt/archs/esrgan/arch.py:12-138`` with blocks from ``resselt/utilities/block.py``
(RRDB :277-344, ResidualDenseBlock_5C :347-465, upconv_block :510-537).  The parameter names are the
reference's old-arch names; the forward pass is the launch list below instead of 351 ``nn.Conv2d`` calls,
276 ``torch.cat`` copies and two ``nn.Upsample`` materialisations:

  * dense concatenation = plane offsets into one 24-plane (nf + 4*gc channels) workspace per RDB;
  * LeakyReLU, ``x5*0.2 + x``, ``out*0.2 + x`` (RRDB) and the trunk shortcut are conv epilogues
    (the residual stream stays f32, the conv operands are split bf16);
  * nearest x2 is folded into the read of the following convolution.
"""

from __future__ import annotations

import math
import os

import torch

from ...engine import lib as L
from ...engine import ops
from ...engine.base import EngineModule, Plan, check_fp16_range
from ...engine.tensors import PF_BF16, PF_F16
from ...engine.paramtree import build_param_tree

_GC = 32  # growth channels are fixed by the reference ctor (arch.py:88)


def rrdbnet_param_shapes(in_nc, out_nc, nf, nb, scale, plus) -> dict:
    shapes: dict = {}

    def conv(name, cout, cin, k=3, bias=True):
        shapes[f'{name}.weight'] = (cout, cin, k, k)
        if bias:
            shapes[f'{name}.bias'] = (cout,)

    conv('model.0', nf, in_nc)
    for i in range(nb):
        for r in (1, 2, 3):
            p = f'model.1.sub.{i}.RDB{r}'
            if plus:
                conv(f'{p}.conv1x1', _GC, nf, 1, bias=False)
            for j in range(1, 6):
                conv(f'{p}.conv{j}.0', _GC if j < 5 else nf, nf + (j - 1) * _GC)
    conv(f'model.1.sub.{nb}', nf, nf)
    k = 3
    for _ in range(int(math.log2(scale))):
        conv(f'model.{k}', nf, nf)
        k += 3
    k -= 1
    conv(f'model.{k}', nf, nf)
    conv(f'model.{k + 2}', out_nc, nf)
    return shapes


def new_arch_to_old(state_dict) -> dict:
    """Official Real-ESRGAN / BSRGAN key spellings -> the old-arch names this module owns.

    Same mapping as the reference's ``_to_old_arch`` (resselt/archs/esrgan/__init__.py:14-121); unlike the
    reference's registry, the converted dict is what actually gets loaded (SURVEY.md §3.1 deviation).
    """
    if 'conv_first.weight' not in state_dict:
        return state_dict
    body_keys = [k for k in state_dict if k.startswith(('body.', 'RRDB_trunk.'))]
    nb = 1 + max(int(k.split('.')[1]) for k in body_keys)
    ups = sorted({int(k.split('.')[0][-1]) for k in state_dict if k.startswith(('upconv', 'conv_up'))})
    hr_index = (max(ups) * 3 if ups else 0) + 2
    out = {}
    for key, value in state_dict.items():
        head, _, kind = key.rpartition('.')
        if head == 'conv_first':
            new = f'model.0.{kind}'
        elif head in ('conv_body', 'trunk_conv'):
            new = f'model.1.sub.{nb}.{kind}'
        elif head.startswith(('body.', 'RRDB_trunk.')):
            _, blk, rdb, conv = head.split('.')
            new = f'model.1.sub.{blk}.RDB{rdb[-1]}.{conv}.0.{kind}'
        elif head.startswith(('upconv', 'conv_up')):
            new = f'model.{int(head[-1]) * 3}.{kind}'
        elif head in ('HRconv', 'conv_hr'):
            new = f'model.{hr_index}.{kind}'
        elif head == 'conv_last':
            new = f'model.{hr_index + 2}.{kind}'
        else:
            new = key
        out[new] = value
    return out


class RRDBNet(EngineModule):
    hyperparameters = {}
    supports_u8 = True  # uint8 [N, H, W, C] images: /255 in the layout kernel, clamp*255+round in the last convolution's store
    # 'mixed' (what 'auto' selects): the convolutions inside residual dense blocks -- 92 % of the multiply-accumulates, attenuated by the
    # 0.2 * 0.2 residual scalings -- run ONE fp16 product on hi planes; conv_first, the upsampling / HR / last convolutions run three bf16
    # products and the trunk convolution three fp16 products on the fp16 residual stream.  Measured against the fp32 oracle on RRDBNet-23:
    # 1.2e-4 max-abs (uniform synthetic weights), 9e-5 (heavy-tailed), where 'bf16x3' gives 2.6e-5 and one product everywhere 1.9e-3
    # (tests/test_precision_policy.py emulates the table on the CPU; tests/test_baseline_configs_gpu.py pins the kernels).
    auto_precision = 'mixed'
    precisions = ('bf16x3', 'bf16', 'mixed')

    @staticmethod
    def layer_policy(name: str) -> tuple[int, int]:
        """(products, plane format of the inputs and weights) of convolution ``name`` under 'mixed'."""
        if '.RDB' in name:
            return 1, PF_F16
        if name.startswith('model.1.sub.'):  # the trunk convolution: reads the fp16 residual stream (hi + lo)
            return 3, PF_F16
        return 3, PF_BF16

    def __init__(self, in_nc: int = 3, out_nc: int = 3, num_filters: int = 64, num_blocks: int = 23, scale: int = 4,
                 plus: bool = False, shuffle_factor: int | None = None) -> None:  # fmt: skip
        super().__init__()
        if scale not in (1, 2, 4, 8):
            raise NotImplementedError(f'RRDBNet engine supports power-of-two scales, got {scale}')
        if num_filters % 8:
            raise NotImplementedError('num_filters must be a multiple of 8')
        self.in_nc, self.out_nc, self.nf, self.nb = in_nc, out_nc, num_filters, num_blocks
        self.net_scale = scale  # upsampling done by the conv stack
        self.plus = plus
        self.shuffle_factor = shuffle_factor
        self.scale = scale // shuffle_factor if shuffle_factor else scale
        self.tail_band_rows = 272  # low-resolution rows per band of the 2x / 4x tail (bounds the plan's HR buffers; >= image height: one band)
        self.plane_residuals = True  # residual stream kept as split planes only (False: the f32-map plan; A/B and plain-bf16 mode)
        self.stream_lo8 = os.environ.get('RSA_STREAM_LO8', '1') != '0'  # 'mixed': lo halves of the residual stream as 8-bit codes (A/B: RSA_STREAM_LO8=0)
        if plus or num_filters != 64:
            # the 'mixed' table is built for the x4plus / x2plus / ESRGAN trunk (64 channels: the fp16x3 trunk convolution is the four-tile
            # ring kernel); ESRGAN+ adds f32 side maps it has no plan for.  Such checkpoints run the conservative mode under 'auto'.
            self.auto_precision = 'bf16x3'
        build_param_tree(self, rrdbnet_param_shapes(in_nc, out_nc, num_filters, num_blocks, scale, plus))

    def _convert_state_dict(self, state_dict):
        return new_arch_to_old(state_dict)

    def macs_per_input_pixel(self) -> int:
        """Algorithmic multiply-accumulates per pixel of the network input grid (SURVEY.md §8d counts these)."""
        total, res = 0, 1
        n_up = int(math.log2(self.net_scale))
        first_up = 3
        for name, w in self.state_dict().items():
            if not name.endswith('.weight'):
                continue
            parts = name.split('.')
            if parts[1] not in ('0', '1'):
                idx = int(parts[1])
                ups_done = min(n_up, (idx - first_up) // 3 + 1) if idx >= first_up else 0
                res = 4**ups_done
            total += w.shape[0] * w.shape[1] * w.shape[2] * w.shape[3] * res
        return total

    # ---------------------------------------------------------------- weights
    def _pack(self, device, products):
        sd = {k: v.detach() for k, v in self.state_dict().items()}
        mixed = products.name == 'mixed'
        if mixed and self.plus:
            raise NotImplementedError("ESRGAN+ (conv1x1 branch) has no 'mixed' plan; use precision = 'bf16x3'")
        out = {}
        for name in sd:
            if name.endswith('.weight'):
                base = name[: -len('.weight')]
                prod, fmt = self.layer_policy(base) if mixed else (int(products), products.fmt)
                out[base] = ops.ConvWeights.from_oihw(sd[name], sd.get(f'{base}.bias'), prod, device=device, fmt=fmt)
        check_fp16_range(out.values())
        return out

    # ---------------------------------------------------------------- plan
    def _build_plan(self, plan: Plan, W, x_shape, dtype, products):
        n, c, h_in, w_in = x_shape
        sf = self.shuffle_factor
        if sf:
            # pixel-unshuffle front (arch.py:130-137): reflect-pad to a multiple, fold s x s pixels into channels
            pad_h, pad_w = (sf - h_in % sf) % sf, (sf - w_in % sf) % sf
            h, w = (h_in + pad_h) // sf, (w_in + pad_w) // sf
            c_net = c * sf * sf
        else:
            pad_h = pad_w = 0
            h, w, c_net = h_in, w_in, c
        if c_net != self.in_nc:
            raise RuntimeError(f'model expects {self.in_nc // (sf * sf if sf else 1)} input channels, got {c}')
        nf, nb, gc = self.nf, self.nb, _GC
        with_lo = products == 3
        mixed = products.name == 'mixed'
        pf, pg = nf // 8, gc // 8
        x_pl = plan.planes(n, (c_net + 7) // 8, h, w, with_lo)
        # The residual stream x of the RDBs lives ONLY as the split planes the next convolution reads (hi + lo, ~16 bits; measured
        # cost of rounding it there instead of keeping an f32 copy: 3e-5 max-abs on RRDBNet-23): conv5 takes `x5*0.2 + x` and the
        # RRDB's `out*0.2 + x` from planes.  Three workspaces rotate inside an RRDB: RDB1 0 -> 1, RDB2 1 -> 2, RDB3 2 -> 0, so the RRDB
        # input (workspace 0, planes 0..pf) is still intact when RDB3's conv5 adds it, and is then overwritten in place by that
        # same launch (each lane reads its residual elements before it stores them).  Plain-bf16 mode has no lo planes and keeps
        # the f32 residual maps (a bf16 residual stream would lose what little accuracy that mode has).
        # 'mixed': the workspaces are fp16; the trunk channels x (planes 0..pf) keep hi + lo -- the residual stream, 22 bits -- and the growth
        # channels x1..x4 hi only (their one-product consumers never read lo): 2 bytes per channel read and written instead of 4.
        plane_res = mixed or (with_lo and self.plane_residuals)
        # Round 4: the stream's lo halves travel as 8-bit codes (offsets from hi in 1/254 ulp: 3 bytes per channel instead of 4; rsa_conv_params.lo8_flags) from conv_first to
        # the last dense block, whose output the trunk convolution reads as fp16 hi + lo operands.  Same accuracy (1.2e-4 against fp32:
        # tests/test_precision_policy.py), 128-192 B per pixel less traffic in every conv5.
        lo8 = mixed and self.stream_lo8
        if mixed:
            ws = [plan.planes(n, pf + 4 * pg, h, w, True, PF_F16, lo_planes=pf) for _ in range(3)]
            if lo8:
                for b_ in ws:
                    b_.with_lo8(pf)
        else:
            ws = [plan.planes(n, pf + 4 * pg, h, w, with_lo) for _ in range(3 if plane_res else 2)]
        fea = plan.f32map(n, nf, h, w)
        pool = [] if plane_res else [plan.f32map(n, nf, h, w) for _ in range(4)]
        lrelu = dict(act=L.ACT_LRELU, act_param=0.2)

        def set_input(x):
            # reflect padding + pixel_unshuffle (when the checkpoint has the unshuffle front end) happen inside the layout kernel
            ops.nchw_to_planes(x, x_pl, unshuffle=sf or 1)

        # fea conv (arch.py:74-80): split planes into workspace 0 and the f32 copy the trunk shortcut adds at the end
        plan.conv(ops.conv_params(W['model.0'], x_pl, h, w, out=ws[0], out_plane_off=0, out_f32=fea, out_lo8=lo8))
        cur_f32, cur_ws = fea, 0
        free = list(pool)
        c11 = plan.f32map(n, gc, h, w) if self.plus else None
        x2_f32 = plan.f32map(n, gc, h, w) if self.plus else None
        for i in range(nb):
            rrdb_in = cur_f32
            taken = []
            for r in (1, 2, 3):
                p = f'model.1.sub.{i}.RDB{r}'
                nxt_ws = (r % 3) if plane_res else cur_ws ^ 1
                a, b = ws[cur_ws], ws[nxt_ws]
                for j in range(1, 5):
                    kw = dict(cin_planes=pf + (j - 1) * pg, out=a, out_plane_off=pf + (j - 1) * pg, **lrelu)
                    if self.plus and j == 2:
                        # ESRGAN+ (block.py:457-463): x2 = lrelu(conv2) + conv1x1(x); x4 = lrelu(conv4) + x2
                        plan.conv(ops.conv_params(W[f'{p}.conv1x1'], a, h, w, cin_planes=pf, out_f32=c11))
                        kw.update(res1=c11, alpha=1.0, out_f32=x2_f32)
                    if self.plus and j == 4:
                        kw.update(res1=x2_f32, alpha=1.0)
                    plan.conv(ops.conv_params(W[f'{p}.conv{j}.0'], a, h, w, **kw))
                if plane_res:
                    l8 = ('lo8',) if lo8 else ()
                    kw = dict(cin_planes=pf + 4 * pg, res1=(a, 0, *l8), alpha=0.2, out=b, out_plane_off=0,
                              out_lo8=lo8 and not (i == nb - 1 and r == 3))  # the last block hands fp16 lo halves to the trunk convolution
                    if r == 3:
                        kw.update(res2=(ws[0], 0, *l8), beta=0.2)  # RRDB.forward: out*0.2 + x (block.py:340-344); ws[0] is also `b`
                else:
                    nxt = free.pop()
                    taken.append(nxt)
                    kw = dict(cin_planes=pf + 4 * pg, res1=cur_f32, alpha=0.2, out=b, out_plane_off=0, out_f32=nxt)
                    if r == 3:
                        kw.update(res2=rrdb_in, beta=0.2)
                    cur_f32 = nxt
                plan.conv(ops.conv_params(W[f'{p}.conv5.0'], a, h, w, **kw))
                cur_ws = nxt_ws
            if not plane_res:
                # recycle f32 maps: everything except the RRDB output
                if rrdb_in is not fea:
                    free.append(rrdb_in)
                free.extend(taken[:2])
        # trunk conv + ShortcutBlock (block.py:83-91)
        u = plan.planes(n, pf, h, w, with_lo)
        if mixed:
            # run-time guard of the fp16 layers: an activation beyond the fp16 range becomes an infinity that `x5 * 0.2 + x` carries through every
            # later block into this map (block.py:463-465, :340-344), whatever the output dtype; EngineModule scans it behind every forward
            plan.range_probe = [u.hi]
        plan.conv(ops.conv_params(W[f'model.1.sub.{nb}'], ws[cur_ws], h, w, cin_planes=pf, res1=fea, alpha=1.0, out=u))
        # ---- tail at 2x / 4x resolution (upconv blocks, HR conv, last conv: arch.py:110-126), run in BANDS of low-resolution rows.
        # A 64-channel map at 4x resolution is 8.5 GB per 1080p frame in split planes; three of them made the plan 25 GB.  The tail is
        # a chain of 3x3 convolutions, so a band of rows needs a halo of 1 row per convolution at that convolution's resolution
        # (5 rows at 4x = 2 low-resolution rows); the band is computed with that halo, zero padding only at true image borders, and the
        # contaminated halo rows of the band's output are dropped when it is copied into the frame.
        n_up = int(math.log2(self.net_scale))
        s_net = 2**n_up
        halo_lr = 2
        band_lr = max(16, int(self.tail_band_rows))
        n_bands = max(1, -(-h // band_lr))
        band_lr = -(-h // n_bands)
        sub_h = min(h, band_lr + 2 * halo_lr)  # rows of the largest band with its halo
        bufs = []  # per resolution level: planes for `sub_h << level` rows
        for lv in range(1, n_up + 1):
            bufs.append(plan.planes(n, pf, sub_h << lv, w << lv, with_lo))
        hr_buf = plan.planes(n, pf, sub_h * s_net, w * s_net, with_lo)
        hh, wwid = h * s_net, w * s_net
        out_buf: dict = {}
        u8 = dtype == torch.uint8
        out_shape = (n, hh, wwid, self.out_nc) if u8 else (n, self.out_nc, hh, wwid)
        band_out = torch.empty(n * self.out_nc * sub_h * s_net * wwid, dtype=dtype, device=plan.device) if n_bands > 1 else None  # flat: each band views it densely
        plan.keep.append(band_out)
        from ...engine.tensors import PlaneRows

        last_entries = []
        copies = []
        for bi in range(n_bands):
            a0, a1 = bi * band_lr, min(h, (bi + 1) * band_lr)
            r0, r1 = max(0, a0 - halo_lr), min(h, a1 + halo_lr)
            rows = r1 - r0
            src = PlaneRows(u, r0, r1) if n_bands > 1 else u
            k = 3
            for lv in range(1, n_up + 1):
                dst = PlaneRows(bufs[lv - 1], 0, rows << lv) if n_bands > 1 else bufs[lv - 1]
                plan.conv(ops.conv_params(W[f'model.{k}'], src, rows << lv, w << lv, upsample2x=True, out=dst, **lrelu))
                src = dst
                k += 3
            k -= 1
            hr = PlaneRows(hr_buf, 0, rows * s_net) if n_bands > 1 else hr_buf
            plan.conv(ops.conv_params(W[f'model.{k}'], src, rows * s_net, wwid, out=hr, **lrelu))
            if n_bands > 1:
                tmp = band_out[: n * self.out_nc * rows * s_net * wwid]
                tmp = tmp.view(n, rows * s_net, wwid, self.out_nc) if u8 else tmp.view(n, self.out_nc, rows * s_net, wwid)
                plan.conv(ops.conv_params(W[f'model.{k + 2}'], hr, rows * s_net, wwid, out_nchw=tmp))
                arr = plan.flush()
                lo, hi_ = (a0 - r0) * s_net, (a1 - r0) * s_net

                def copy_band(tmp=tmp, lo=lo, hi_=hi_, y0=a0 * s_net, y1=a1 * s_net):
                    if u8:
                        out_buf['y'][:, y0:y1] = tmp[:, lo:hi_]
                    else:
                        out_buf['y'][:, :, y0:y1] = tmp[:, :, lo:hi_]

                plan.call(copy_band)
            else:
                placeholder = torch.empty(out_shape, dtype=dtype, device=plan.device)
                out_buf['y'] = placeholder
                plan.conv(ops.conv_params(W[f'model.{k + 2}'], hr, hh, wwid, out_nchw=placeholder))
                arr = plan.flush()
                last_entries.append(arr[len(arr) - 1])

        # a fresh output tensor per call (single band: patch the last descriptor's pointer; bands: the copies fill it)
        def prepare_output():
            if 'y' not in out_buf:
                out_buf['y'] = torch.empty(out_shape, dtype=dtype, device=plan.device)
            for e in last_entries:
                e.out_nchw = out_buf['y'].data_ptr()

        plan.steps.insert(0, prepare_output)
        if n_bands > 1:
            out_buf.pop('y', None)

        def get_output():
            y = out_buf.pop('y')
            if sf:
                y = y[:, : h_in * self.scale, : w_in * self.scale] if u8 else y[:, :, : h_in * self.scale, : w_in * self.scale]
            return y

        return set_input, get_output
