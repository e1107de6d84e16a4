"""HAT (Hybrid Attention Transformer) on the MI355X engine -- drop-in for ``resselt/archs/hat/arch.py:798-1110`` in eval mode.

Tokens are pixels, the residual stream is an f32 map, Linear layers are k1 launches of the convolution kernels:

  HAB  (arch.py:218-348)  LN -> { CAB: conv3x3 + GELU, conv3x3, channel attention (global mean -> 1x1 -> ReLU -> 1x1 -> sigmoid) }
                          and { qkv -> (shifted) 16x16 window attention -> proj };  x = shortcut + attn + CAB * 0.01;  LN -> MLP
  OCAB (arch.py:351-482)  LN -> qkv -> attention of every 16x16 window's queries over the 24x24 window around it (nn.Unfold with zero
                          padding) -> proj (+ shortcut);  LN -> MLP
  RHAG (arch.py:590-692)  blocks, one OCAB, conv3x3 (+ residual)

Both attention kinds run on ``rsa_rect_attention`` (csrc/dat.hip): the self-attention with its 256-token window resident in LDS, the
overlapping one in its cross-window mode (576 keys streamed through LDS in chunks of 256, flash-style state in registers).  The
relative-position tables are gathered once, at pack time, into the kernel's accumulator-fragment order.
"""

from __future__ import annotations

import ctypes as C
import os
import math

import torch

from ...engine import lib as L
from ...engine import ops, swinblocks
from ...engine.base import EngineModule, Plan, check_fp16_range
from ...engine.tensors import PF_BF16, PF_F16
from ...engine.paramtree import build_param_tree
from ..dat.arch import attn_tiles
from ..swinir.arch import HEAD_PAD, regroup_proj, regroup_qkv

RGB_MEAN = (0.4488, 0.4371, 0.4040)  # arch.py:842


def rpi_buffers(window: int, overlap_ratio: float):
    """``relative_position_index_SA`` / ``_OCA`` (arch.py:987-1034)."""
    co = torch.stack(torch.meshgrid([torch.arange(window), torch.arange(window)], indexing='ij')).flatten(1)
    rel = (co[:, :, None] - co[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += window - 1
    rel[:, :, 1] += window - 1
    rel[:, :, 0] *= 2 * window - 1
    sa = rel.sum(-1)
    ext = window + int(overlap_ratio * window)
    ce = torch.stack(torch.meshgrid([torch.arange(ext), torch.arange(ext)], indexing='ij')).flatten(1)
    rel = (ce[:, None, :] - co[:, :, None]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += window - ext + 1
    rel[:, :, 1] += window - ext + 1
    rel[:, :, 0] *= window + ext - 1
    return sa, rel.sum(-1)


def bias_fragments_qk(dense: torch.Tensor, qt: int, kt: int) -> torch.Tensor:
    """[heads, Nq, Nk] (query, key) position bias -> [heads][qt][kt][lane 64][16] f32 in the S^T accumulator order of rsa_rect_attention:
    lane l, element r  <->  query 32*q + (l & 31),  key 32*k + (r & 3) + 8*(r >> 2) + 4*(l >> 5).  Padded keys get -1e30."""
    heads, nq, nk = dense.shape
    full = torch.zeros((heads, 32 * qt, 32 * kt), dtype=torch.float32, device=dense.device)
    full[:, :, nk:] = -1e30
    full[:, :nq, :nk] = dense.to(torch.float32)
    lane = torch.arange(64, device=dense.device)
    r = torch.arange(16, device=dense.device)
    q_in = (lane & 31)[:, None].expand(64, 16)
    k_in = ((r & 3) + 8 * (r >> 2))[None, :] + 4 * (lane >> 5)[:, None]
    out = torch.empty((heads, qt, kt, 64, 16), dtype=torch.float32, device=dense.device)
    for a in range(qt):
        for b in range(kt):
            out[:, a, b] = full[:, 32 * a + q_in, 32 * b + k_in]
    return out.contiguous()


def hat_param_shapes(in_chans, embed_dim, depths, num_heads, window, compress_ratio, squeeze_factor, overlap_ratio, mlp_ratio, upscale, num_feat,
                     resi, patch_norm, qkv_bias):  # fmt: skip
    shapes: dict = {}
    C_ = embed_dim
    hidden = int(C_ * mlp_ratio)
    ext = window + int(overlap_ratio * window)

    def conv(name, co, ci, k):
        shapes[f'{name}.weight'] = (co, ci, k, k)
        shapes[f'{name}.bias'] = (co,)

    def lin(name, co, ci, bias=True):
        shapes[f'{name}.weight'] = (co, ci)
        if bias:
            shapes[f'{name}.bias'] = (co,)

    def ln(name):
        shapes[f'{name}.weight'] = (C_,)
        shapes[f'{name}.bias'] = (C_,)

    conv('conv_first', C_, in_chans, 3)
    if patch_norm:
        ln('patch_embed.norm')
    for i, depth in enumerate(depths):
        g = f'layers.{i}.residual_group'
        for j in range(depth):
            b = f'{g}.blocks.{j}'
            ln(f'{b}.norm1')
            shapes[f'{b}.attn.relative_position_bias_table'] = ((2 * window - 1) ** 2, num_heads[i])
            lin(f'{b}.attn.qkv', 3 * C_, C_, qkv_bias)
            lin(f'{b}.attn.proj', C_, C_)
            conv(f'{b}.conv_block.cab.0', int(C_ // compress_ratio), C_, 3)
            conv(f'{b}.conv_block.cab.2', C_, int(C_ // compress_ratio), 3)
            conv(f'{b}.conv_block.cab.3.attention.1', int(C_ // squeeze_factor), C_, 1)
            conv(f'{b}.conv_block.cab.3.attention.3', C_, int(C_ // squeeze_factor), 1)
            ln(f'{b}.norm2')
            lin(f'{b}.mlp.fc1', hidden, C_)
            lin(f'{b}.mlp.fc2', C_, hidden)
        o = f'{g}.overlap_attn'
        ln(f'{o}.norm1')
        lin(f'{o}.qkv', 3 * C_, C_, qkv_bias)
        shapes[f'{o}.relative_position_bias_table'] = ((window + ext - 1) ** 2, num_heads[i])
        lin(f'{o}.proj', C_, C_)
        ln(f'{o}.norm2')
        lin(f'{o}.mlp.fc1', hidden, C_)
        lin(f'{o}.mlp.fc2', C_, hidden)
        if resi == '1conv':
            conv(f'layers.{i}.conv', C_, C_, 3)
    ln('norm')
    if resi == '1conv':
        conv('conv_after_body', C_, C_, 3)
    conv('conv_before_upsample.0', num_feat, C_, 3)
    if upscale == 3:
        conv('upsample.0', 9 * num_feat, num_feat, 3)
    elif upscale & (upscale - 1) == 0:
        for u in range(int(math.log2(upscale))):
            conv(f'upsample.{2 * u}', 4 * num_feat, num_feat, 3)
    else:
        raise ValueError(f'scale {upscale} is not supported. Supported scales: 2^n and 3.')
    conv('conv_last', in_chans, num_feat, 3)
    sa, oca = rpi_buffers(window, overlap_ratio)
    return shapes, {'relative_position_index_SA': sa, 'relative_position_index_OCA': oca}


class HAT(EngineModule):
    hyperparameters = {}
    # 'mixed' (what 'auto' selects when the MLP halves run the fused kernel): everything between two reads of the f32 token stream -- qkv, the
    # window / overlapping cross-window attention, proj, the CAB's two 3x3 convolutions and the fused norm2 + fc1 + GELU + fc2 half -- runs ONE
    # fp16 product on hi planes; the group / head convolutions keep three bf16 products.
    precisions = ('bf16x3', 'bf16', 'mixed')
    precision_table = {'mixed': (3, PF_BF16)}

    @property
    def auto_precision(self) -> str:
        return 'mixed'

    @staticmethod
    def layer_policy(name: str) -> tuple[int, int]:
        """(products, plane format of inputs and weights) of layer ``name`` under 'mixed'."""
        if name.endswith(('.qkv', '.proj', '.mlp.fc1', '.mlp.fc2')) or '.conv_block.cab.' in name:
            return 1, PF_F16
        return 3, PF_BF16

    fused_mlp = os.environ.get('RSA_HAT_FUSED_MLP', '1') != '0'  # LayerNorm + fc1 + GELU + fc2 + shortcut as one launch where the widths allow it (engine/swinblocks.py)

    def __init__(self, *, img_size=64, patch_size=1, in_chans=3, embed_dim=96, depths=(6, 6, 6, 6), num_heads=(6, 6, 6, 6), window_size=7,
                 compress_ratio=3, squeeze_factor=30, conv_scale=0.01, overlap_ratio=0.5, mlp_ratio=4.0, qkv_bias=True, qk_scale=None,
                 drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.1, ape=False, patch_norm=True, use_checkpoint=False, upscale=1,
                 img_range=1.0, upsampler='pixelshuffle', resi_connection='1conv', num_feat=64) -> None:  # fmt: skip
        super().__init__()
        if patch_size != 1 or ape or qk_scale is not None or upsampler != 'pixelshuffle':
            raise NotImplementedError('HAT engine supports patch_size=1, ape=False, default qk scale, pixelshuffle (what the loader builds)')
        depths, num_heads = list(depths), list(num_heads)
        ext = window_size + int(overlap_ratio * window_size)
        if window_size * window_size > 256 or (ext - window_size) % 2:
            raise NotImplementedError('window_size must be <= 16 and the overlap symmetric')
        if any(embed_dim % h or embed_dim // h > HEAD_PAD for h in num_heads) or embed_dim % 4 or num_feat % 8:
            raise NotImplementedError('head_dim must divide embed_dim and be <= 32; embed_dim % 4 == 0; num_feat % 8 == 0')
        if int(embed_dim // squeeze_factor) < 1 or int(embed_dim // squeeze_factor) > 64:
            raise NotImplementedError('channel-attention width must be in 1..64')
        self.in_chans, self.embed_dim, self.depths, self.num_heads = in_chans, embed_dim, depths, num_heads
        self.window_size, self.ext, self.mlp_ratio, self.conv_scale = window_size, ext, mlp_ratio, conv_scale
        self.compress, self.upscale, self.img_range, self.resi = int(embed_dim // compress_ratio), upscale, img_range, resi_connection
        self.patch_norm, self.num_feat, self.img_size = patch_norm, num_feat, img_size
        shapes, buffers = hat_param_shapes(in_chans, embed_dim, depths, num_heads, window_size, compress_ratio, squeeze_factor, overlap_ratio,
                                           mlp_ratio, upscale, num_feat, resi_connection, patch_norm, qkv_bias)  # fmt: skip
        build_param_tree(self, shapes, buffers)

    # ---------------------------------------------------------------- weights
    def _pack(self, device, products):
        sd = {k: v.detach().to(device) for k, v in self.state_dict().items()}
        W: dict = {}
        C_, ws, ext = self.embed_dim, self.window_size, self.ext
        cp = (C_ + 7) // 8

        def f32(t):
            return t.to(torch.float32).contiguous()

        mixed = products.name == 'mixed'

        def policy(name):
            return self.layer_policy(name) if mixed else (int(products), products.fmt)

        def conv(name):
            prod, fmt = policy(name)
            W[name] = ops.ConvWeights.from_oihw(sd[f'{name}.weight'], sd.get(f'{name}.bias'), prod, device=device, fmt=fmt)

        def lin(name, w=None, b=None, cin_planes=None):
            w = sd[f'{name}.weight'] if w is None else w
            b = sd.get(f'{name}.bias') if b is None else b
            prod, fmt = policy(name)
            W[name] = ops.ConvWeights.from_oihw(w[:, :, None, None], b, prod, cin_planes=cin_planes, device=device, fmt=fmt)

        def ln(name):
            W[name] = (f32(sd[f'{name}.weight']), f32(sd[f'{name}.bias']))

        def attention(name, heads, table_key, rpi, nq, nk, cross):
            wq, bq = regroup_qkv(sd[f'{name}.qkv.weight'], sd.get(f'{name}.qkv.bias'), heads)
            lin(f'{name}.qkv', wq, bq)
            lin(f'{name}.proj', regroup_proj(sd[f'{name}.proj.weight'], heads), sd[f'{name}.proj.bias'], cin_planes=heads * HEAD_PAD // 8)
            dense = f32(sd[table_key])[rpi.reshape(-1).long()].view(nq, nk, heads).permute(2, 0, 1)
            qt = (nq + 31) // 32 if cross else attn_tiles(nq)
            kt = (nk + 31) // 32 if cross else attn_tiles(nk)
            W[f'{name}.bias_frag'] = bias_fragments_qk(dense, qt, kt)

        def pad_cols(t, n):
            out = torch.zeros((t.shape[0], n), dtype=torch.float32, device=device)
            out[:, : t.shape[1]] = t
            return out.contiguous()

        def pad_rows(t, n):
            out = torch.zeros((n,) + tuple(t.shape[1:]), dtype=torch.float32, device=device)
            out[: t.shape[0]] = t
            return out.contiguous()

        conv('conv_first')
        if self.patch_norm:
            ln('patch_embed.norm')
        rpi_sa, rpi_oca = sd['relative_position_index_SA'], sd['relative_position_index_OCA']
        for i, depth in enumerate(self.depths):
            heads = self.num_heads[i]
            g = f'layers.{i}.residual_group'
            for j in range(depth):
                b = f'{g}.blocks.{j}'
                ln(f'{b}.norm1')
                ln(f'{b}.norm2')
                attention(f'{b}.attn', heads, f'{b}.attn.relative_position_bias_table', rpi_sa, ws * ws, ws * ws, False)
                conv(f'{b}.conv_block.cab.0')
                conv(f'{b}.conv_block.cab.2')
                w1 = f32(sd[f'{b}.conv_block.cab.3.attention.1.weight']).reshape(-1, C_)
                w2 = f32(sd[f'{b}.conv_block.cab.3.attention.3.weight']).reshape(C_, -1)
                W[f'{b}.ca'] = (pad_cols(w1, cp * 8), f32(sd[f'{b}.conv_block.cab.3.attention.1.bias']), pad_rows(w2, cp * 8),
                                pad_rows(f32(sd[f'{b}.conv_block.cab.3.attention.3.bias']), cp * 8))  # fmt: skip
                lin(f'{b}.mlp.fc1')
                lin(f'{b}.mlp.fc2')
            o = f'{g}.overlap_attn'
            ln(f'{o}.norm1')
            ln(f'{o}.norm2')
            attention(o, heads, f'{o}.relative_position_bias_table', rpi_oca, ws * ws, ext * ext, True)
            lin(f'{o}.mlp.fc1')
            lin(f'{o}.mlp.fc2')
            if self.resi == '1conv':
                conv(f'layers.{i}.conv')
        ln('norm')
        if self.resi == '1conv':
            conv('conv_after_body')
        else:  # 'identity': the residual adds still run as (exact) identity k1 launches
            W['identity'] = ops.ConvWeights.from_oihw(torch.eye(C_, device=device)[:, :, None, None], None, int(products), device=device, fmt=products.fmt)
        for name in ('conv_before_upsample.0', 'conv_last', 'upsample.0', 'upsample.2', 'upsample.4'):
            if f'{name}.weight' in sd:
                conv(name)
        check_fp16_range(W.values())
        W['mean'] = torch.tensor(RGB_MEAN if self.in_chans == 3 else [0.0] * self.in_chans, dtype=torch.float32, device=device)
        return W

    def macs_per_input_pixel(self) -> int:
        """Algorithmic MACs per (window-padded) input pixel."""
        C_, ws, ext = self.embed_dim, self.window_size, self.ext
        hidden = int(C_ * self.mlp_ratio)
        macs = 9 * self.in_chans * C_
        for depth in self.depths:
            macs += depth * (4 * C_ * C_ + 2 * ws * ws * C_ + 2 * 9 * C_ * self.compress + 2 * C_ * hidden)
            macs += 4 * C_ * C_ + 2 * ext * ext * C_ + 2 * C_ * hidden
            macs += 9 * C_ * C_ if self.resi == '1conv' else 0
        macs += 9 * C_ * C_ if self.resi == '1conv' else 0
        nf, s = self.num_feat, self.upscale
        macs += 9 * C_ * nf
        res = 1
        if s == 3:
            macs += 9 * nf * 9 * nf
            res = 9
        else:
            for _ in range(int(math.log2(s))):
                macs += 9 * nf * 4 * nf * res
                res *= 4
        return macs + 9 * nf * self.in_chans * res

    # ---------------------------------------------------------------- plan
    def _build_plan(self, plan: Plan, W, x_shape, dtype, products):
        n, c, h0, w0 = x_shape
        if c != self.in_chans:
            raise RuntimeError(f'model expects {self.in_chans} input channels, got {c}')
        ws, ext = self.window_size, self.ext
        H, Wd = h0 + (ws - h0 % ws) % ws, w0 + (ws - w0 % ws) % ws
        if H - h0 >= h0 or Wd - w0 >= w0:
            raise RuntimeError('input is too small for reflect padding to a multiple of the window size')
        C_, s, nf = self.embed_dim, self.upscale, self.num_feat
        hidden = int(C_ * self.mlp_ratio)
        with_lo = products == 3
        cp = (C_ + 7) // 8
        dev = plan.device
        lib = L.load()
        max_heads = max(self.num_heads)

        def stream():
            return C.c_void_p(ops.current_stream_ptr(dev))

        def launch(fn_name, params, kernels=1):
            fn = getattr(lib, fn_name)
            plan.call(lambda: L.check(fn(C.byref(params), stream()), fn_name))
            plan.count_launches(kernels)

        x_pl = plan.planes(n, (c + 7) // 8, H, Wd, with_lo)
        mean = W['mean']

        def set_input(x):
            ops.nchw_to_planes(x, x_pl, mean, self.img_range)  # (x - mean) * img_range and check_image_size's reflect padding (arch.py:1091-1101)

        first = plan.f32map(n, C_, H, Wd)
        pool = [plan.f32map(n, C_, H, Wd) for _ in range(5)]
        mixed = products.name == 'mixed'
        one = dict(with_lo=False, fmt=PF_F16) if mixed else dict(with_lo=with_lo)  # what a one-product layer reads: an fp16 hi plane
        a_pl = plan.planes(n, cp, H, Wd, **one)  # norm1 -> qkv and the CAB
        n_pl = plan.planes(n, cp, H, Wd, with_lo) if mixed else a_pl  # the last LayerNorm -> conv_after_body (three products)
        qkv_pl = plan.planes(n, 3 * max_heads * HEAD_PAD // 8, H, Wd, **one)
        o_pl = plan.planes(n, max_heads * HEAD_PAD // 8, H, Wd, **one)
        fuse_mlp = self.fused_mlp and swinblocks.mlp_block_fits(C_, hidden)
        hid_pl = None if fuse_mlp else plan.planes(n, (hidden + 7) // 8, H, Wd, **one)
        body_pl = plan.planes(n, cp, H, Wd, with_lo)
        cab_a = plan.planes(n, (self.compress + 7) // 8, H, Wd, **one)
        cab_b = plan.planes(n, cp, H, Wd, with_lo)
        gate = torch.empty((n, cp * 8), dtype=torch.float32, device=dev)
        ws_gate = torch.empty((max(int(lib.rsa_channel_gate_workspace_bytes(n, H, Wd, cp)), 16) // 4,), dtype=torch.float32, device=dev)
        plan.keep += [gate, ws_gate]

        def layernorm(name, x_f32, out_planes=None, out_f32=None):
            g, b = W[name]
            lp = L.LayerNormParams()
            lp.batch, lp.H, lp.W, lp.C, lp.eps = n, H, Wd, C_, 1e-5
            lp.x_f32, lp.gamma, lp.beta = x_f32.data_ptr(), g.data_ptr(), b.data_ptr()
            if out_planes is not None:
                lp.out_hi, lp.out_lo = out_planes.hi_ptr(), out_planes.lo_ptr()
                lp.out_plane_stride, lp.out_batch_stride = out_planes.plane_stride, out_planes.batch_stride
                lp.out_fmt = out_planes.fmt
            lp.out_f32 = None if out_f32 is None else out_f32.data_ptr()
            launch('rsa_layernorm', lp)

        def attention(name, heads, shift, cross):
            ap = L.RectAttnParams()
            ap.batch, ap.H, ap.W, ap.Hp, ap.Wp = n, H, Wd, H, Wd
            ap.win_h, ap.win_w, ap.shift_h, ap.shift_w = ws, ws, shift, shift
            ap.heads, ap.head0, ap.heads_total, ap.products = heads, 0, heads, (1 if mixed else int(products))
            ap.fmt = qkv_pl.fmt
            ap.qkv_hi, ap.qkv_lo = qkv_pl.hi_ptr(), qkv_pl.lo_ptr()
            ap.qkv_plane_stride, ap.qkv_batch_stride = qkv_pl.plane_stride, qkv_pl.batch_stride
            ap.bias_frag = W[f'{name}.bias_frag'].data_ptr()
            ap.out_hi, ap.out_lo = o_pl.hi_ptr(), o_pl.lo_ptr()
            ap.out_plane_stride, ap.out_batch_stride = o_pl.plane_stride, o_pl.batch_stride
            if cross:
                ap.kwin_h = ap.kwin_w = ext
                ap.kpad_h = ap.kpad_w = (ext - ws) // 2
            launch('rsa_rect_attention', ap)

        def cab_scaled_shortcut(b, shortcut, out_f32):
            """out = shortcut + CAB(LN(x)) * conv_scale, CAB = conv-GELU-conv followed by its channel attention (arch.py:37-59, 345)."""
            plan.conv(ops.conv_params(W[f'{b}.conv_block.cab.0'], a_pl, H, Wd, cin_planes=cp, act=L.ACT_GELU, out=cab_a))
            plan.conv(ops.conv_params(W[f'{b}.conv_block.cab.2'], cab_a, H, Wd, out=cab_b))
            w1, b1, w2, b2 = W[f'{b}.ca']
            gp = L.ChannelGateParams()
            gp.batch, gp.H, gp.W, gp.planes, gp.hidden, gp.relu = n, H, Wd, cp, w1.shape[0], 1
            gp.in_hi, gp.in_lo = cab_b.hi_ptr(), cab_b.lo_ptr()
            gp.in_plane_stride, gp.in_batch_stride = cab_b.plane_stride, cab_b.batch_stride
            gp.w1, gp.b1, gp.w2, gp.b2 = w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr()
            gp.workspace, gp.gate = ws_gate.data_ptr(), gate.data_ptr()
            launch('rsa_channel_gate', gp, kernels=2)

            def run():
                L.check(lib.rsa_gated_add(cab_b.hi_ptr(), cab_b.lo_ptr(), cab_b.plane_stride, cab_b.batch_stride, n, H, Wd, C_, gate.data_ptr(),
                                          self.conv_scale, shortcut.data_ptr(), out_f32.data_ptr(), stream()), 'rsa_gated_add')  # fmt: skip

            plan.call(run)
            plan.count_launches(1)

        def mlp(b, x1, x2, out_planes=None):
            if fuse_mlp:  # one launch (csrc/swin_block.hip), nothing between leaves the chip
                swinblocks.mlp_block(plan, W[f'{b}.norm2'], W[f'{b}.mlp.fc1'], W[f'{b}.mlp.fc2'], n, H, Wd, C_, hidden, products, x1, x2, out_planes)
                return
            layernorm(f'{b}.norm2', x1, out_planes=a_pl)
            plan.conv(ops.conv_params(W[f'{b}.mlp.fc1'], a_pl, H, Wd, cin_planes=cp, act=L.ACT_GELU, out=hid_pl))
            plan.conv(ops.conv_params(W[f'{b}.mlp.fc2'], hid_pl, H, Wd, cin_planes=(hidden + 7) // 8, res1=x1, alpha=1.0, out_f32=x2, out=out_planes))

        plan.conv(ops.conv_params(W['conv_first'], x_pl, H, Wd, out_f32=first))
        free = list(pool)
        if self.patch_norm:
            cur = free.pop()
            layernorm('patch_embed.norm', first, out_f32=cur)
        else:
            cur = first
        for i, depth in enumerate(self.depths):
            heads = self.num_heads[i]
            hp = heads * HEAD_PAD // 8
            g = f'layers.{i}.residual_group'
            rg_in = cur

            def release(t):
                if t is not rg_in and t is not first:
                    free.append(t)

            for j in range(depth):
                b = f'{g}.blocks.{j}'
                layernorm(f'{b}.norm1', cur, out_planes=a_pl)
                sc = free.pop()
                cab_scaled_shortcut(b, cur, sc)
                plan.conv(ops.conv_params(W[f'{b}.attn.qkv'], a_pl, H, Wd, cin_planes=cp, out=qkv_pl))
                attention(f'{b}.attn', heads, 0 if j % 2 == 0 else ws // 2, False)
                x1 = free.pop()
                plan.conv(ops.conv_params(W[f'{b}.attn.proj'], o_pl, H, Wd, cin_planes=hp, res1=sc, alpha=1.0, out_f32=x1))
                free.append(sc)
                x2 = free.pop()
                mlp(b, x1, x2)
                release(cur)
                free.append(x1)
                cur = x2
            o = f'{g}.overlap_attn'
            layernorm(f'{o}.norm1', cur, out_planes=a_pl)
            plan.conv(ops.conv_params(W[f'{o}.qkv'], a_pl, H, Wd, cin_planes=cp, out=qkv_pl))
            attention(o, heads, 0, True)
            x1 = free.pop()
            plan.conv(ops.conv_params(W[f'{o}.proj'], o_pl, H, Wd, cin_planes=hp, res1=cur, alpha=1.0, out_f32=x1))
            x2 = free.pop()
            mlp(o, x1, x2, out_planes=body_pl)
            release(cur)
            free.append(x1)
            cur = x2
            out = free.pop()
            tail = W[f'layers.{i}.conv'] if self.resi == '1conv' else W['identity']
            plan.conv(ops.conv_params(tail, body_pl, H, Wd, cin_planes=cp, res1=rg_in, alpha=1.0, out_f32=out))
            if rg_in is not first:
                free.append(rg_in)
            free.append(cur)
            cur = out
        layernorm('norm', cur, out_planes=n_pl)
        tail = W['conv_after_body'] if self.resi == '1conv' else W['identity']
        plan.conv(ops.conv_params(tail, n_pl, H, Wd, cin_planes=cp, res1=first, alpha=1.0, out=body_pl))  # + conv_first output (arch.py:1104)

        out_shape = (n, self.in_chans, H * s, Wd * s)
        out_buf = {'y': torch.empty(out_shape, dtype=dtype, device=dev)}
        y = plan.planes(n, nf // 8, H, Wd, with_lo)
        plan.conv(ops.conv_params(W['conv_before_upsample.0'], body_pl, H, Wd, cin_planes=cp, act=L.ACT_LRELU, act_param=0.01, out=y))
        hh, ww = H, Wd
        i = 0
        while f'upsample.{i}' in W:
            r = math.isqrt(W[f'upsample.{i}'].cout // nf)
            shuffled = torch.empty((n, nf, hh * r, ww * r), dtype=torch.float32, device=dev)
            plan.keep.append(shuffled)
            plan.conv(ops.conv_params(W[f'upsample.{i}'], y, hh, ww, out_nchw=shuffled, pixel_shuffle=r))
            hh, ww = hh * r, ww * r
            ny = plan.planes(n, nf // 8, hh, ww, with_lo)
            plan.call(lambda src=shuffled, dst=ny: ops.nchw_to_planes(src, dst))
            y = ny
            i += 2
        plan.conv(ops.conv_params(W['conv_last'], y, hh, ww, out_nchw=out_buf['y'], out_scale=1.0 / self.img_range, out_shift=mean))
        arr = plan.flush()
        last_entry = arr[len(arr) - 1]

        def prepare_output():
            if 'y' not in out_buf:
                out_buf['y'] = torch.empty(out_shape, dtype=dtype, device=dev)
            last_entry.out_nchw = out_buf['y'].data_ptr()

        plan.steps.insert(len(plan.steps) - 1, prepare_output)

        def get_output():
            return out_buf.pop('y')[:, :, : h0 * s, : w0 * s]

        return set_input, get_output
