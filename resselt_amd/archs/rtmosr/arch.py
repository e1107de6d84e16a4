"""RTMoSR on the MI355X engine (reference module: ``resselt/archs/rtmosr/arch.py:340-387``), eval-mode semantics.

Every RepConv (SeqConv3x3 + 3x3 + Conv3XC, alpha-weighted) and every OmniShift (identity + 1x1 + 3x3 + 5x5 depthwise) is folded to one
kernel at pack time.  A GatedCNNBlock (arch.py:302-337) is then

  RMSNorm (rsa_rmsnorm) -> fc1 3x3 conv -> [ g | i | c ] plane ranges
  c: PixelUnshuffle(2) + RepConv(MaxPool2d(2))  (rsa_unshuffle_pool, then one conv whose residual operand is the unshuffled map)
     -> depthwise 5x5 (rsa_dwconv5x5) -> SE gate (rsa_channel_gate, ReLU / Hardsigmoid) -> PixelShuffle(2)
  mish(g) * cat(i, c)  (rsa_gated_shuffle_mul: shuffle and SE scaling happen in the read)  -> fc2 conv + Mish + shortcut (conv epilogue)

and the model is to_feat conv (pixel-unshuffle front end folded into the layout kernel) -> blocks -> to_img conv stored through
depth-to-space with the nearest-upsampled input added in the same store.
"""

from __future__ import annotations

import ctypes as C
import math

import torch

from ...engine import lib as L
from ...engine import ops
from ...engine.base import EngineModule, Plan
from ...engine.paramtree import build_param_tree
from ..spanpp.arch import fold_repconv, repconv_shapes


def fold_omnishift(sd: dict, key: str) -> tuple[torch.Tensor, torch.Tensor]:
    """OmniShift.reparam_5x5 (arch.py:253-277) in f64: one depthwise 5x5 kernel [C, 25] and bias [C]."""
    d = torch.float64
    F = torch.nn.functional
    a1, a2, a3, a4 = (sd[f'{key}.alpha{k}'].to(d).reshape(-1, 1, 1, 1) for k in (1, 2, 3, 4))
    w1 = F.pad(sd[f'{key}.conv1x1.weight'].to(d), (2, 2, 2, 2))
    w3 = F.pad(sd[f'{key}.conv3x3.weight'].to(d), (1, 1, 1, 1))
    ident = F.pad(torch.ones_like(sd[f'{key}.conv1x1.weight'].to(d)), (2, 2, 2, 2))
    w = a1 * ident + a2 * w1 + a3 * w3 + a4 * sd[f'{key}.conv5x5.weight'].to(d)
    b = (a2.reshape(-1) * sd[f'{key}.conv1x1.bias'].to(d) + a3.reshape(-1) * sd[f'{key}.conv3x3.bias'].to(d)
         + a4.reshape(-1) * sd[f'{key}.conv5x5.bias'].to(d))  # fmt: skip
    return w.reshape(w.shape[0], 25).to(torch.float32).contiguous(), b.to(torch.float32).contiguous()


class RTMoSR(EngineModule):
    hyperparameters = {}

    def __init__(self, *, scale: int = 2, dim: int = 32, ffn_expansion: float = 2, n_blocks: int = 2, unshuffle_mod: bool = False, dccm: bool = True,
                 se: bool = True) -> None:  # fmt: skip
        super().__init__()
        self.scale = scale
        unshuffle = 0
        s_int = scale
        if scale < 4 and unshuffle_mod:
            if scale == 3:
                raise ValueError('Unshuffle_mod does not support 3x')
            unshuffle = 4 // scale
            s_int = 4
        hidden = int(ffn_expansion * dim)
        if dim % 8 or hidden % 8 or (hidden - dim) % 8 or hidden < dim:
            raise NotImplementedError('dim, the hidden width and their difference must be multiples of 8 (plane-aligned g / i / c split)')
        self.dim, self.hidden, self.n_blocks, self.unshuffle, self.s_int, self.dccm, self.se = dim, hidden, n_blocks, unshuffle, s_int, dccm, se
        self.pad = (unshuffle if unshuffle > 0 else 1) * 2
        shapes: dict = {}
        repconv_shapes(shapes, 'to_feat.1' if unshuffle else 'to_feat', dim, 3 * unshuffle * unshuffle if unshuffle else 3)
        for i in range(n_blocks):
            b = f'body.{i}'
            shapes[f'{b}.norm.scale'] = (dim,)
            shapes[f'{b}.norm.offset'] = (dim,)
            repconv_shapes(shapes, f'{b}.fc1', 2 * hidden, dim)
            repconv_shapes(shapes, f'{b}.conv.0.poll.1', 4 * dim, dim)
            for k in (1, 2, 3, 4):
                shapes[f'{b}.conv.1.alpha{k}'] = (1, 4 * dim, 1, 1)
            for name, ks in (('conv1x1', 1), ('conv3x3', 3), ('conv5x5', 5), ('conv5x5_reparam', 5)):
                shapes[f'{b}.conv.1.{name}.weight'] = (4 * dim, 1, ks, ks)
                shapes[f'{b}.conv.1.{name}.bias'] = (4 * dim,)
            if se:
                shapes[f'{b}.conv.2.squeezing.0.weight'] = (2 * dim, 4 * dim, 1, 1)
                shapes[f'{b}.conv.2.squeezing.0.bias'] = (2 * dim,)
                shapes[f'{b}.conv.2.squeezing.2.weight'] = (4 * dim, 2 * dim, 1, 1)
                shapes[f'{b}.conv.2.squeezing.2.bias'] = (4 * dim,)
            if dccm:
                repconv_shapes(shapes, f'{b}.fc2', dim, hidden)
            else:
                shapes[f'{b}.fc2.weight'] = (dim, hidden, 1, 1)
                shapes[f'{b}.fc2.bias'] = (dim,)
        repconv_shapes(shapes, 'to_img.0', 3 * s_int * s_int, dim)
        build_param_tree(self, shapes, {})

    def _pack(self, device, products):
        sd = {k: v.detach().to(device=device, dtype=torch.float32) for k, v in self.state_dict().items()}
        W: dict = {}

        def rep(name):
            w, b = fold_repconv(sd, name)
            W[name] = ops.ConvWeights.from_oihw(w, b, products, device=device)

        rep('to_feat.1' if self.unshuffle else 'to_feat')
        for i in range(self.n_blocks):
            b = f'body.{i}'
            W[f'{b}.norm'] = (sd[f'{b}.norm.scale'].contiguous(), sd[f'{b}.norm.offset'].contiguous())
            rep(f'{b}.fc1')
            rep(f'{b}.conv.0.poll.1')
            W[f'{b}.omni'] = fold_omnishift(sd, f'{b}.conv.1')
            if self.se:
                W[f'{b}.se'] = (sd[f'{b}.conv.2.squeezing.0.weight'].reshape(2 * self.dim, 4 * self.dim).contiguous(), sd[f'{b}.conv.2.squeezing.0.bias'].contiguous(),
                                sd[f'{b}.conv.2.squeezing.2.weight'].reshape(4 * self.dim, 2 * self.dim).contiguous(), sd[f'{b}.conv.2.squeezing.2.bias'].contiguous())  # fmt: skip
            if self.dccm:
                rep(f'{b}.fc2')
            else:
                W[f'{b}.fc2'] = ops.ConvWeights.from_oihw(sd[f'{b}.fc2.weight'], sd[f'{b}.fc2.bias'], products, device=device)
        rep('to_img.0')
        return W

    def macs_per_input_pixel(self) -> int:
        """Algorithmic MACs per padded input pixel (convolutions and depthwise convolutions)."""
        d, h = self.dim, self.hidden
        u = self.unshuffle or 1
        cin0 = 3 * u * u
        per_feat_px = 9 * d * 2 * h + (9 * d * 4 * d + 25 * 4 * d) // 4 + (9 if self.dccm else 1) * h * d
        total = 9 * cin0 * d + self.n_blocks * per_feat_px + 9 * d * 3 * self.s_int**2
        return total // (u * u)

    def _build_plan(self, plan: Plan, W, x_shape, dtype, products):
        n, c, h0, w0 = x_shape
        if c != 3:
            raise RuntimeError(f'model expects 3 input channels, got {c}')
        pad, u = self.pad, (self.unshuffle or 1)
        Hp, Wp = h0 + (pad - h0 % pad) % pad, w0 + (pad - w0 % pad) % pad
        if Hp - h0 >= h0 or Wp - w0 >= w0:
            raise RuntimeError('input is too small for reflect padding to the block size')
        H, Wd = Hp // u, Wp // u  # feature resolution
        dim, hidden, s_int = self.dim, self.hidden, self.s_int
        out_scale = s_int // u
        with_lo = products == 3
        dev = plan.device
        lib = L.load()
        pd, ph = dim // 8, hidden // 8
        gi = (hidden - dim) // 8  # planes of i

        def stream():
            return C.c_void_p(ops.current_stream_ptr(dev))

        def call(fn):
            plan.call(fn)
            plan.count_launches(1)

        x_pl = plan.planes(n, (3 * u * u + 7) // 8, H, Wd, with_lo)
        holder = {}

        def set_input(x):
            holder['x'] = x  # the final store adds the nearest-upsampled input back (arch.py:387)
            ops.nchw_to_planes(x, x_pl, unshuffle=u)  # check_img_size's reflect padding and the PixelUnshuffle front end, fused

        cur = plan.f32map(n, dim, H, Wd)
        nxt = plan.f32map(n, dim, H, Wd)
        a_pl = plan.planes(n, pd, H, Wd, with_lo)
        f_pl = plan.planes(n, 2 * ph, H, Wd, with_lo)
        pool_pl = plan.planes(n, pd, H // 2, Wd // 2, with_lo)
        pu = plan.f32map(n, 4 * dim, H // 2, Wd // 2)
        c_pl = plan.planes(n, 4 * pd, H // 2, Wd // 2, with_lo)
        o_pl = plan.planes(n, 4 * pd, H // 2, Wd // 2, with_lo)
        m_pl = plan.planes(n, ph, H, Wd, with_lo)
        feat_pl = plan.planes(n, pd, H, Wd, with_lo)
        gate = torch.empty((n, 4 * dim), dtype=torch.float32, device=dev)
        ws_gate = torch.empty((max(int(lib.rsa_channel_gate_workspace_bytes(n, H // 2, Wd // 2, 4 * pd)), 16) // 4,), dtype=torch.float32, device=dev)
        plan.keep += [gate, ws_gate]

        plan.conv(ops.conv_params(W['to_feat.1' if self.unshuffle else 'to_feat'], x_pl, H, Wd, out_f32=cur))
        for i in range(self.n_blocks):
            b = f'body.{i}'
            sc, off = W[f'{b}.norm']

            def rms(src=cur, sc=sc, off=off):
                L.check(lib.rsa_rmsnorm(src.data_ptr(), n, H, Wd, dim, 1e-6, sc.data_ptr(), off.data_ptr(), a_pl.hi_ptr(), a_pl.lo_ptr(), a_pl.plane_stride,
                                        a_pl.batch_stride, stream()), 'rsa_rmsnorm')  # fmt: skip

            call(rms)
            plan.conv(ops.conv_params(W[f'{b}.fc1'], a_pl, H, Wd, out=f_pl))

            def unshuffle_pool():
                L.check(lib.rsa_unshuffle_pool(f_pl.hi_ptr(ph + gi), f_pl.lo_ptr(ph + gi), f_pl.plane_stride, f_pl.batch_stride, n, H, Wd, pd, pu.data_ptr(),
                                               pool_pl.hi_ptr(), pool_pl.lo_ptr(), pool_pl.plane_stride, pool_pl.batch_stride, stream()),
                        'rsa_unshuffle_pool')  # fmt: skip

            call(unshuffle_pool)
            plan.conv(ops.conv_params(W[f'{b}.conv.0.poll.1'], pool_pl, H // 2, Wd // 2, res1=pu, alpha=1.0, out=c_pl))
            ow, ob = W[f'{b}.omni']
            dp = L.DwConvParams()
            dp.batch, dp.H, dp.W, dp.planes, dp.act = n, H // 2, Wd // 2, 4 * pd, L.ACT_NONE
            dp.in_hi, dp.in_lo, dp.in_plane_stride, dp.in_batch_stride = c_pl.hi_ptr(), c_pl.lo_ptr(), c_pl.plane_stride, c_pl.batch_stride
            dp.weight, dp.bias = ow.data_ptr(), ob.data_ptr()
            dp.out_hi, dp.out_lo, dp.out_plane_stride, dp.out_batch_stride = o_pl.hi_ptr(), o_pl.lo_ptr(), o_pl.plane_stride, o_pl.batch_stride
            call(lambda dp=dp: L.check(lib.rsa_dwconv5x5(C.byref(dp), stream()), 'rsa_dwconv5x5'))
            if self.se:
                w1, b1, w2, b2 = W[f'{b}.se']
                gp = L.ChannelGateParams()
                gp.batch, gp.H, gp.W, gp.planes, gp.hidden, gp.relu = n, H // 2, Wd // 2, 4 * pd, w1.shape[0], 2
                gp.in_hi, gp.in_lo, gp.in_plane_stride, gp.in_batch_stride = o_pl.hi_ptr(), o_pl.lo_ptr(), o_pl.plane_stride, o_pl.batch_stride
                gp.w1, gp.b1, gp.w2, gp.b2 = w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr()
                gp.workspace, gp.gate = ws_gate.data_ptr(), gate.data_ptr()
                plan.call(lambda gp=gp: L.check(lib.rsa_channel_gate(C.byref(gp), stream()), 'rsa_channel_gate'))
                plan.count_launches(2)
            sp = L.GatedShuffleParams()
            sp.batch, sp.H, sp.W, sp.g_planes, sp.i_planes = n, H, Wd, ph, gi
            sp.f_hi, sp.f_lo, sp.f_plane_stride, sp.f_batch_stride = f_pl.hi_ptr(), f_pl.lo_ptr(), f_pl.plane_stride, f_pl.batch_stride
            sp.c_hi, sp.c_lo, sp.c_plane_stride, sp.c_batch_stride = o_pl.hi_ptr(), o_pl.lo_ptr(), o_pl.plane_stride, o_pl.batch_stride
            sp.gate, sp.gate_stride = (gate.data_ptr() if self.se else None), 4 * dim
            sp.out_hi, sp.out_lo, sp.out_plane_stride, sp.out_batch_stride = m_pl.hi_ptr(), m_pl.lo_ptr(), m_pl.plane_stride, m_pl.batch_stride
            call(lambda sp=sp: L.check(lib.rsa_gated_shuffle_mul(C.byref(sp), stream()), 'rsa_gated_shuffle_mul'))
            last = i == self.n_blocks - 1
            # mish(fc2(.)) + shortcut: activation, then the residual, both in the conv epilogue
            plan.conv(ops.conv_params(W[f'{b}.fc2'], m_pl, H, Wd, act=L.ACT_MISH, res1=cur, alpha=1.0, out_f32=nxt, out=feat_pl if last else None))
            cur, nxt = nxt, cur
        if self.n_blocks == 0:
            raise NotImplementedError('RTMoSR without blocks')

        out_shape = (n, 3, H * s_int, Wd * s_int)
        out_buf = {'y': torch.empty(out_shape, dtype=dtype, device=dev)}
        base0 = torch.empty((n, 3, h0, w0), dtype=dtype, device=dev)  # placeholder pointer, patched per call
        plan.conv(ops.conv_params(W['to_img.0'], feat_pl, H, Wd, out_nchw=out_buf['y'], pixel_shuffle=s_int, out_base=base0, out_base_div=out_scale))
        arr = plan.flush()
        last_entry = arr[len(arr) - 1]

        def prepare():
            if 'y' not in out_buf:
                out_buf['y'] = torch.empty(out_shape, dtype=dtype, device=dev)
            last_entry.out_nchw = out_buf['y'].data_ptr()
            last_entry.out_base = holder['x'].data_ptr()

        plan.steps.insert(len(plan.steps) - 1, prepare)

        def get_output():
            holder.clear()
            return out_buf.pop('y')[:, :, : h0 * out_scale, : w0 * out_scale]

        return set_input, get_output
