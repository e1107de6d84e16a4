"""DAT (Dual Aggregation Transformer) on the MI355X engine -- drop-in for ``resselt/archs/dat/arch.py:828-990`` in eval mode.

Tokens are pixels.  The residual stream is an f32 map, every Linear layer is a k1 launch of the convolution kernels, and the
rest of the block runs on the kernels of ``csrc/dat.hip``:

  DSTB (even blocks, arch.py:270-513)   LN -> qkv -> 2 x rect-window attention (8x32 and 32x8 on the two channel halves, shifted on
                                        every second one) -> depthwise conv(v) -> channel gate -> AIM combine -> proj (+ residual)
  DCTB (odd blocks,  arch.py:516-612)   LN -> qkv -> channel attention: Gram matrix over all tokens -> softmax -> packed 1x1
                                        weights -> attn @ v as one more conv launch -> depthwise conv(v) -> gate -> AIM -> proj
  SGFN (arch.py:42-101)                 LN -> fc1 + GELU -> per-pixel LN statistics of the second half -> depthwise conv of the
                                        normalised half, multiplied by the first half -> fc2 (+ residual)

Attention-side maps use the head-padded channel layout (head h = channels [32h, 32h+32)); every weight that touches them is
permuted once at pack time.  BatchNorm is folded with its running statistics (the reference module straight from the loader is
in training mode, where DropPath is random and BatchNorm uses batch statistics; inference callers use ``.eval()``).
"""

from __future__ import annotations

import ctypes as C
import math

import torch

from ...engine import lib as L
from ...engine import ops
from ...engine.base import EngineModule, Plan, check_fp16_range
from ...engine.paramtree import build_param_tree
from ...engine.tensors import PF_BF16, PF_F16, Planes
from ..swinir.arch import HEAD_PAD, regroup_proj, regroup_qkv

RGB_MEAN = (0.4488, 0.4371, 0.4040)  # arch.py:879
BN_EPS = 1e-5


def branch_geometry(pair, idx: int):
    """(h, w) of a per-branch quantity: branch 1 swaps the rectangle (arch.py:186-191)."""
    return (pair[0], pair[1]) if idx == 0 else (pair[1], pair[0])


def is_shifted(rg_idx: int, b_idx: int) -> bool:  # arch.py:312, 453
    return (rg_idx % 2 == 0 and b_idx > 0 and (b_idx - 2) % 4 == 0) or (rg_idx % 2 != 0 and b_idx % 4 == 0)


def rpe_buffers(hs: int, ws: int):
    """``rpe_biases`` and ``relative_position_index`` buffers of one Spatial_Attention (arch.py:193-211)."""
    bh, bw = torch.arange(1 - hs, hs), torch.arange(1 - ws, ws)
    biases = torch.stack(torch.meshgrid([bh, bw], indexing='ij')).flatten(1).transpose(0, 1).contiguous().float()
    coords = torch.stack(torch.meshgrid([torch.arange(hs), torch.arange(ws)], indexing='ij')).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += hs - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return biases, rel.sum(-1)


def shift_masks(H: int, W: int, split, shift):
    """The registered ``attn_mask_0/1`` buffers (arch.py:336-411); kept for state_dict parity, the kernel derives the mask itself."""
    out = []
    for idx in (0, 1):
        hs, ws = branch_geometry(split, idx)
        sh, sw = branch_geometry(shift, idx)
        img = torch.zeros(H, W)
        cnt = 0
        for a in (slice(0, -hs), slice(-hs, -sh), slice(-sh, None)):
            for b in (slice(0, -ws), slice(-ws, -sw), slice(-sw, None)):
                img[a, b] = cnt
                cnt += 1
        mw = img.view(H // hs, hs, W // ws, ws).permute(0, 2, 1, 3).reshape(-1, hs * ws)
        d = mw.unsqueeze(1) - mw.unsqueeze(2)
        out.append(torch.where(d != 0, torch.full_like(d, -100.0), torch.zeros_like(d)))
    return out


def attn_tiles(ntok: int) -> int:
    """Tiles of 32 tokens the rect-attention kernel is instantiated for."""
    t = (ntok + 31) // 32
    return 1 if t <= 1 else 2 if t <= 2 else 4 if t <= 4 else 8


def bias_fragments(dense: torch.Tensor) -> torch.Tensor:
    """[heads, N, N] (query, key) position bias -> [heads][T][T][lane 64][16] f32 in the S^T accumulator order of the kernel:
    lane l, element r  <->  query 32*qt + (l & 31),  key 32*kt + (r & 3) + 8*(r >> 2) + 4*(l >> 5).  Padded keys get -1e30."""
    heads, n, _ = dense.shape
    T = attn_tiles(n)
    full = torch.zeros((heads, 32 * T, 32 * T), dtype=torch.float32, device=dense.device)
    full[:, :, n:] = -1e30
    full[:, :n, :n] = dense.to(torch.float32)
    lane = torch.arange(64, device=dense.device)
    r = torch.arange(16, device=dense.device)
    q_in = (lane & 31)[:, None].expand(64, 16)
    k_in = ((r & 3) + 8 * (r >> 2))[None, :] + 4 * (lane >> 5)[:, None]
    out = torch.empty((heads, T, T, 64, 16), dtype=torch.float32, device=dense.device)
    for qt in range(T):
        for kt in range(T):
            out[:, qt, kt] = full[:, 32 * qt + q_in, 32 * kt + k_in]
    return out.contiguous()


def pad_heads(t: torch.Tensor, heads: int, dim: int = 0) -> torch.Tensor:
    """Scatter a length-C axis (head-major, C = heads*hd) into the head-padded layout of length heads*32 (zeros in the pads)."""
    c = t.shape[dim]
    hd = c // heads
    shape = list(t.shape)
    t = t.to(torch.float32).reshape(shape[:dim] + [heads, hd] + shape[dim + 1 :])
    out_shape = shape[:dim] + [heads, HEAD_PAD] + shape[dim + 1 :]
    out = torch.zeros(out_shape, dtype=torch.float32, device=t.device)
    out.narrow(dim + 1, 0, hd).copy_(t)
    return out.reshape(shape[:dim] + [heads * HEAD_PAD] + shape[dim + 1 :]).contiguous()


def pad_rows(t: torch.Tensor, rows: int) -> torch.Tensor:
    out = torch.zeros((rows,) + tuple(t.shape[1:]), dtype=torch.float32, device=t.device)
    out[: t.shape[0]] = t.to(torch.float32)
    return out.contiguous()


def dat_param_shapes(in_chans, embed_dim, split_size, depth, num_heads, expansion_factor, qkv_bias, upscale, resi, upsampler, img_size):
    shapes: dict = {}
    buffers: dict = {}
    C_ = embed_dim
    hidden = int(C_ * expansion_factor)
    shift_size = [split_size[0] // 2, split_size[1] // 2]

    def conv(name, co, ci, k):
        shapes[f'{name}.weight'] = (co, ci, k, k)
        shapes[f'{name}.bias'] = (co,)

    def lin(name, co, ci, bias=True):
        shapes[f'{name}.weight'] = (co, ci)
        if bias:
            shapes[f'{name}.bias'] = (co,)

    def ln(name, c):
        shapes[f'{name}.weight'] = (c,)
        shapes[f'{name}.bias'] = (c,)

    def bn(name, c):
        ln(name, c)
        buffers[f'{name}.running_mean'] = torch.zeros(c)
        buffers[f'{name}.running_var'] = torch.ones(c)
        buffers[f'{name}.num_batches_tracked'] = torch.tensor(0, dtype=torch.int64)

    def dw(name, c):
        shapes[f'{name}.weight'] = (c, 1, 3, 3)
        shapes[f'{name}.bias'] = (c,)

    def resi_conv(name):
        if resi == '1conv':
            conv(name, C_, C_, 3)
        else:
            conv(f'{name}.0', C_ // 4, C_, 3)
            conv(f'{name}.2', C_ // 4, C_ // 4, 1)
            conv(f'{name}.4', C_, C_ // 4, 3)

    def aim(name):
        dw(f'{name}.dwconv.0', C_)
        bn(f'{name}.dwconv.1', C_)
        conv(f'{name}.channel_interaction.1', C_ // 8, C_, 1)
        bn(f'{name}.channel_interaction.2', C_ // 8)
        conv(f'{name}.channel_interaction.4', C_, C_ // 8, 1)
        conv(f'{name}.spatial_interaction.0', C_ // 16, C_, 1)
        bn(f'{name}.spatial_interaction.1', C_ // 16)
        conv(f'{name}.spatial_interaction.3', 1, C_ // 16, 1)

    pos_dim = ((C_ // 2) // 4) // 4
    conv('conv_first', C_, in_chans, 3)
    ln('before_RG.1', C_)
    masks = None
    for i, d in enumerate(depth):
        heads = num_heads[i]
        for j in range(d):
            b = f'layers.{i}.blocks.{j}'
            ln(f'{b}.norm1', C_)
            if j % 2 == 0:
                lin(f'{b}.attn.qkv', 3 * C_, C_, qkv_bias)
                lin(f'{b}.attn.proj', C_, C_)
                for idx in (0, 1):
                    a = f'{b}.attn.attns.{idx}'
                    hs, ws = branch_geometry(split_size, idx)
                    buffers[f'{a}.rpe_biases'], buffers[f'{a}.relative_position_index'] = rpe_buffers(hs, ws)
                    lin(f'{a}.pos.pos_proj', pos_dim, 2)
                    for k, co in (('pos1', pos_dim), ('pos2', pos_dim), ('pos3', heads // 2)):
                        ln(f'{a}.pos.{k}.0', pos_dim)
                        lin(f'{a}.pos.{k}.2', co, pos_dim)
                if is_shifted(i, j):
                    if masks is None:
                        masks = shift_masks(img_size, img_size, split_size, shift_size)
                    buffers[f'{b}.attn.attn_mask_0'], buffers[f'{b}.attn.attn_mask_1'] = masks
            else:
                shapes[f'{b}.attn.temperature'] = (heads, 1, 1)
                lin(f'{b}.attn.qkv', 3 * C_, C_, qkv_bias)
                lin(f'{b}.attn.proj', C_, C_)
            aim(f'{b}.attn')
            lin(f'{b}.ffn.fc1', hidden, C_)
            ln(f'{b}.ffn.sg.norm', hidden // 2)
            dw(f'{b}.ffn.sg.conv', hidden // 2)
            lin(f'{b}.ffn.fc2', C_, hidden // 2)
            ln(f'{b}.norm2', C_)
        resi_conv(f'layers.{i}.conv')
    ln('norm', C_)
    resi_conv('conv_after_body')
    if upsampler == 'pixelshuffle':
        conv('conv_before_upsample.0', 64, C_, 3)
        if upscale == 3:
            conv('upsample.0', 9 * 64, 64, 3)
        elif upscale & (upscale - 1) == 0:
            for u in range(int(math.log2(upscale))):
                conv(f'upsample.{2 * u}', 4 * 64, 64, 3)
        else:
            raise ValueError(f'scale {upscale} is not supported. Supported scales: 2^n and 3.')
        conv('conv_last', in_chans, 64, 3)
    else:
        conv('upsample.0', upscale * upscale * in_chans, C_, 3)
    return shapes, buffers


class DAT(EngineModule):
    hyperparameters = {}
    # 'mixed' (what 'auto' selects).  Round 4: the WHOLE transformer body runs on fp16 hi planes in one product -- qkv, proj, fc1, fc2, the
    # rectangular-window attention (`rect_attention_kernel<1, T, f16>`), the channel attention's Gram matrix and its `attn @ v` (the weight
    # blob is written in fp16), the depthwise convolutions, the AIM, the spatial gate (their descriptors carry the plane format) -- as HAT and
    # DRCT have done since round 3.  The residual stream stays an f32 map; the 3x3 convolutions of the residual groups, conv_first,
    # conv_after_body and the reconstruction head keep three bf16 products.  (Round 3: only qkv and fc1, the layers a LayerNorm feeds.)
    auto_precision = 'mixed'
    precisions = ('bf16x3', 'bf16', 'mixed')
    precision_table = {'mixed': (3, PF_BF16)}

    @staticmethod
    def layer_policy(name: str) -> tuple[int, int]:
        """(products, plane format of inputs and weights) of layer ``name`` under 'mixed'."""
        return (1, PF_F16) if name.endswith(('.attn.qkv', '.attn.proj', '.ffn.fc1', '.ffn.fc2')) else (3, PF_BF16)

    def __init__(self, *, img_size=64, in_chans=3, embed_dim=180, split_size=(8, 32), depth=(6, 6, 6, 6, 6, 6), num_heads=(6, 6, 6, 6, 6, 6),
                 expansion_factor=2.0, qkv_bias=True, qk_scale=None, drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.1, use_chk=False,
                 upscale=4, img_range=1.0, resi_connection='1conv', upsampler='pixelshuffle') -> None:  # fmt: skip
        super().__init__()
        split_size, depth, num_heads = list(split_size), list(depth), list(num_heads)
        if qk_scale is not None:
            raise NotImplementedError('DAT engine supports the default qk scale (what the loader builds)')
        if upsampler not in ('pixelshuffle', 'pixelshuffledirect'):
            raise NotImplementedError(f'upsampler {upsampler!r} is not a DAT upsampler')
        if split_size[0] * split_size[1] > 256 or min(split_size) < 2:
            raise NotImplementedError('split_size must hold 4..256 tokens with both sides >= 2')
        if embed_dim % 4 or any(h % 2 or embed_dim % h or embed_dim // h > HEAD_PAD for h in num_heads):
            raise NotImplementedError('embed_dim must be a multiple of 4, heads even, head_dim <= 32')
        if not 1 <= embed_dim // 16 <= 16:
            raise NotImplementedError('embed_dim // 16 (spatial-interaction width) must be in 1..16')
        hidden = int(embed_dim * expansion_factor)
        if hidden % 2:
            raise NotImplementedError('the SGFN hidden width must be even')
        self.in_chans, self.embed_dim, self.split_size, self.depth, self.num_heads = in_chans, embed_dim, split_size, depth, num_heads
        self.hidden, self.qkv_bias, self.upscale, self.img_range = hidden, qkv_bias, upscale, img_range
        self.resi, self.upsampler, self.img_size = resi_connection, upsampler, img_size
        shapes, buffers = dat_param_shapes(in_chans, embed_dim, split_size, depth, num_heads, expansion_factor, qkv_bias, upscale, resi_connection,
                                           upsampler, img_size)  # fmt: skip
        build_param_tree(self, shapes, buffers)

    # ---------------------------------------------------------------- weights
    def _pack(self, device, products):
        sd = {k: v.detach().to(device) for k, v in self.state_dict().items()}
        W: dict = {}
        C_ = self.embed_dim

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

        def resi_conv(name):
            for sub in ([''] if self.resi == '1conv' else ['.0', '.2', '.4']):
                conv(name + sub)

        def bn_fold(name):
            """(scale, shift) of an eval-mode BatchNorm: y = x * scale + shift."""
            s = f32(sd[f'{name}.weight']) / torch.sqrt(f32(sd[f'{name}.running_var']) + BN_EPS)
            return s, f32(sd[f'{name}.bias']) - f32(sd[f'{name}.running_mean']) * s

        def pos_bias(a):
            """DynamicPosBias (residual=False, arch.py:104-143) on rpe_biases, gathered to [heads, N, N] (arch.py:247-252)."""
            F = torch.nn.functional
            pos = F.linear(f32(sd[f'{a}.rpe_biases']), f32(sd[f'{a}.pos.pos_proj.weight']), f32(sd[f'{a}.pos.pos_proj.bias']))
            for k in ('pos1', 'pos2', 'pos3'):
                g = f32(sd[f'{a}.pos.{k}.0.weight'])
                pos = F.layer_norm(pos, (g.shape[0],), g, f32(sd[f'{a}.pos.{k}.0.bias']), 1e-5)
                pos = F.linear(F.relu(pos), f32(sd[f'{a}.pos.{k}.2.weight']), f32(sd[f'{a}.pos.{k}.2.bias']))
            idx = sd[f'{a}.relative_position_index'].long()
            n = idx.shape[0]
            return pos[idx.reshape(-1)].view(n, n, -1).permute(2, 0, 1).contiguous()

        def aim(a, heads):
            s, t = bn_fold(f'{a}.dwconv.1')
            w = f32(sd[f'{a}.dwconv.0.weight']).reshape(C_, 9) * s[:, None]
            W[f'{a}.dw'] = (pad_heads(w, heads), pad_heads(f32(sd[f'{a}.dwconv.0.bias']) * s + t, heads))
            s, t = bn_fold(f'{a}.channel_interaction.2')
            w1 = f32(sd[f'{a}.channel_interaction.1.weight']).reshape(-1, C_) * s[:, None]
            b1 = f32(sd[f'{a}.channel_interaction.1.bias']) * s + t
            w2 = f32(sd[f'{a}.channel_interaction.4.weight']).reshape(C_, -1)
            W[f'{a}.ci'] = (pad_heads(w1, heads, dim=1), b1.contiguous(), pad_heads(w2, heads), pad_heads(f32(sd[f'{a}.channel_interaction.4.bias']), heads))
            s, t = bn_fold(f'{a}.spatial_interaction.1')
            w1 = f32(sd[f'{a}.spatial_interaction.0.weight']).reshape(-1, C_) * s[:, None]
            b1 = f32(sd[f'{a}.spatial_interaction.0.bias']) * s + t
            W[f'{a}.si'] = (pad_heads(w1, heads, dim=1), b1.contiguous(), f32(sd[f'{a}.spatial_interaction.3.weight']).reshape(-1),
                            float(sd[f'{a}.spatial_interaction.3.bias'].float().item()))  # fmt: skip

        half = self.hidden // 2
        P1 = (half + 7) // 8
        conv('conv_first')
        ln('before_RG.1')
        for i, d in enumerate(self.depth):
            heads = self.num_heads[i]
            for j in range(d):
                b = f'layers.{i}.blocks.{j}'
                ln(f'{b}.norm1')
                ln(f'{b}.norm2')
                spatial = j % 2 == 0
                wq, bq = regroup_qkv(sd[f'{b}.attn.qkv.weight'], sd.get(f'{b}.attn.qkv.bias'), heads, scale_q=spatial)
                lin(f'{b}.attn.qkv', wq, bq)
                lin(f'{b}.attn.proj', regroup_proj(sd[f'{b}.attn.proj.weight'], heads), sd[f'{b}.attn.proj.bias'], cin_planes=heads * HEAD_PAD // 8)
                if spatial:
                    for idx in (0, 1):
                        W[f'{b}.attn.bias{idx}'] = bias_fragments(pos_bias(f'{b}.attn.attns.{idx}'))
                else:
                    W[f'{b}.attn.temperature'] = f32(sd[f'{b}.attn.temperature']).reshape(-1)
                aim(f'{b}.attn', heads)
                # fc1 rows: x1 = rows [0, half) on planes [0, P1), x2 = rows [half, 2*half) on planes [P1, 2*P1)
                w1 = torch.zeros((2 * P1 * 8, C_), dtype=torch.float32, device=device)
                b1 = torch.zeros((2 * P1 * 8,), dtype=torch.float32, device=device)
                fw, fb = f32(sd[f'{b}.ffn.fc1.weight']), f32(sd[f'{b}.ffn.fc1.bias'])
                w1[:half], w1[P1 * 8 : P1 * 8 + half] = fw[:half], fw[half:]
                b1[:half], b1[P1 * 8 : P1 * 8 + half] = fb[:half], fb[half:]
                lin(f'{b}.ffn.fc1', w1, b1)
                lin(f'{b}.ffn.fc2')
                W[f'{b}.ffn.sg'] = (pad_rows(f32(sd[f'{b}.ffn.sg.conv.weight']).reshape(half, 9), P1 * 8), pad_rows(f32(sd[f'{b}.ffn.sg.conv.bias']), P1 * 8),
                                    pad_rows(f32(sd[f'{b}.ffn.sg.norm.weight']), P1 * 8), pad_rows(f32(sd[f'{b}.ffn.sg.norm.bias']), P1 * 8))  # fmt: skip
            resi_conv(f'layers.{i}.conv')
        ln('norm')
        resi_conv('conv_after_body')
        for name in ('conv_before_upsample.0', 'conv_last', 'upsample.0', 'upsample.2', 'upsample.4'):
            if f'{name}.weight' in sd:
                conv(name)
        check_fp16_range(W.values())
        W['mean'] = torch.tensor(RGB_MEAN if self.in_chans == 3 else [0.0] * self.in_chans, dtype=torch.float32, device=device)
        return W

    def macs_per_input_pixel(self) -> int:
        """Algorithmic MACs per input pixel (convs, Linear layers, both attention kinds, depthwise convs; AIM gates neglected)."""
        C_, hid = self.embed_dim, self.hidden
        ntok = self.split_size[0] * self.split_size[1]
        macs = 9 * self.in_chans * C_
        resi = 9 * C_ * C_ if self.resi == '1conv' else (9 * C_ * (C_ // 4) * 2 + (C_ // 4) ** 2)
        for i, d in enumerate(self.depth):
            hd = C_ // self.num_heads[i]
            for j in range(d):
                macs += 3 * C_ * C_ + C_ * C_ + 9 * C_  # qkv, proj, depthwise conv on v
                macs += 2 * ntok * C_ if j % 2 == 0 else 2 * hd * C_  # QK^T + PV, or Gram + attn @ v
                macs += C_ * hid + 9 * (hid // 2) + (hid // 2) * C_  # SGFN
            macs += resi
        macs += resi
        s = self.upscale
        if self.upsampler == 'pixelshuffle':
            macs += 9 * C_ * 64
            res = 1
            if s == 3:
                macs += 9 * 64 * 576
                res = 9
            else:
                for _ in range(int(math.log2(s))):
                    macs += 9 * 64 * 256 * res
                    res *= 4
            macs += 9 * 64 * self.in_chans * res
        else:
            macs += 9 * C_ * s * s * self.in_chans
        return macs

    # ---------------------------------------------------------------- plan
    def _build_plan(self, plan: Plan, W, x_shape, dtype, products):
        n, c, H, Wd = x_shape
        if c != self.in_chans:
            raise RuntimeError(f'model expects {self.in_chans} input channels, got {c}')
        C_, s = self.embed_dim, self.upscale
        with_lo = products == 3
        cp = (C_ + 7) // 8
        half = self.hidden // 2
        P1 = (half + 7) // 8
        dev = plan.device
        lib = L.load()
        max_heads = max(self.num_heads)
        hp_max = max_heads * HEAD_PAD // 8
        m = max(self.split_size)
        Hp, Wp = H + (m - H % m) % m, Wd + (m - Wd % m) % m
        shift = [self.split_size[0] // 2, self.split_size[1] // 2]

        def stream():
            return C.c_void_p(ops.current_stream_ptr(dev))

        def launch(fn_name, params):
            fn = getattr(lib, fn_name)
            plan.call(lambda: L.check(fn(C.byref(params), stream()), fn_name))
            plan.count_launches(1)

        x_pl = plan.planes(n, (c + 7) // 8, H, Wd, with_lo)
        mean = W['mean']

        def set_input(x):
            ops.nchw_to_planes(x, x_pl, mean, self.img_range)  # (x - mean) * img_range (arch.py:975-976)

        first = plan.f32map(n, C_, H, Wd)
        pool = [plan.f32map(n, C_, H, Wd) for _ in range(4)]
        mixed = products.name == 'mixed'
        body_kw = dict(with_lo=False, fmt=PF_F16) if mixed else dict(with_lo=with_lo)  # the transformer body: fp16 hi planes under 'mixed'
        bprod = 1 if mixed else int(products)  # matrix products of the body's attention kernels
        bfmt = PF_F16 if mixed else products.fmt
        a_pl = plan.planes(n, cp, H, Wd, **body_kw)  # norm1 / norm2 -> qkv / fc1
        n_pl = plan.planes(n, cp, H, Wd, with_lo) if mixed else a_pl  # the last LayerNorm -> conv_after_body (three products)
        qkv_pl = plan.planes(n, 3 * hp_max, H, Wd, **body_kw)
        att_pl = plan.planes(n, hp_max, H, Wd, **body_kw)
        conv_pl = plan.planes(n, hp_max, H, Wd, **body_kw)
        comb_pl = plan.planes(n, hp_max, H, Wd, **body_kw)
        hid_pl = plan.planes(n, 2 * P1, H, Wd, **body_kw)
        gate_pl = plan.planes(n, P1, H, Wd, **body_kw)
        body_pl = plan.planes(n, cp, H, Wd, with_lo)
        q4_a = plan.planes(n, (C_ // 4 + 7) // 8, H, Wd, with_lo) if self.resi == '3conv' else None
        q4_b = plan.planes(n, (C_ // 4 + 7) // 8, H, Wd, with_lo) if self.resi == '3conv' else None
        stats = torch.empty((n, H * Wd, 2), dtype=torch.float32, device=dev)
        gate = torch.empty((n, max_heads * HEAD_PAD), dtype=torch.float32, device=dev)
        ws_gate = torch.empty((max(int(lib.rsa_channel_gate_workspace_bytes(n, H, Wd, hp_max)), 16) // 4,), dtype=torch.float32, device=dev)
        has_dctb = any(d >= 2 for d in self.depth)
        ws_attn = torch.empty((max(int(lib.rsa_channel_attn_workspace_bytes(n, H, Wd, max_heads)), 16) // 4,), dtype=torch.float32, device=dev)
        zero_bias = torch.zeros((max_heads * HEAD_PAD,), dtype=torch.float32, device=dev)
        plan.keep += [stats, gate, ws_gate, ws_attn, zero_bias]
        wdyn = {}
        if has_dctb:
            for heads in sorted({h for h, d in zip(self.num_heads, self.depth) if d >= 2}):
                blob = int(lib.rsa_packed_weight_bytes(heads * HEAD_PAD, heads * 4, 1, bprod)) // 2
                wdyn[heads] = torch.zeros((n, blob), dtype=torch.bfloat16, device=dev)  # off-diagonal blocks stay zero forever
                plan.keep.append(wdyn[heads])

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

        def rect_attention(b, heads, shifted):
            for idx in (0, 1):
                ap = L.RectAttnParams()
                ap.batch, ap.H, ap.W, ap.Hp, ap.Wp = n, H, Wd, Hp, Wp
                ap.win_h, ap.win_w = branch_geometry(self.split_size, idx)
                ap.shift_h, ap.shift_w = branch_geometry(shift, idx) if shifted else (0, 0)
                ap.heads, ap.head0, ap.heads_total, ap.products = heads // 2, idx * (heads // 2), heads, bprod
                ap.fmt = bfmt
                ap.qkv_hi, ap.qkv_lo = qkv_pl.hi_ptr(), qkv_pl.lo_ptr()
                ap.qkv_plane_stride, ap.qkv_batch_stride = qkv_pl.plane_stride, qkv_pl.batch_stride
                ap.bias_frag = W[f'{b}.attn.bias{idx}'].data_ptr()
                ap.out_hi, ap.out_lo = att_pl.hi_ptr(), att_pl.lo_ptr()
                ap.out_plane_stride, ap.out_batch_stride = att_pl.plane_stride, att_pl.batch_stride
                launch('rsa_rect_attention', ap)

        def channel_attention(b, heads):
            hp = heads * 4
            cpar = L.ChannelAttnParams()
            cpar.batch, cpar.H, cpar.W, cpar.heads, cpar.head_dim, cpar.products = n, H, Wd, heads, C_ // heads, bprod
            cpar.fmt = bfmt
            cpar.q_hi, cpar.q_lo = qkv_pl.hi_ptr(0), qkv_pl.lo_ptr(0)
            cpar.k_hi, cpar.k_lo = qkv_pl.hi_ptr(hp), qkv_pl.lo_ptr(hp)
            cpar.plane_stride, cpar.batch_stride = qkv_pl.plane_stride, qkv_pl.batch_stride
            cpar.temperature = W[f'{b}.attn.temperature'].data_ptr()
            cpar.workspace, cpar.w_packed = ws_attn.data_ptr(), wdyn[heads].data_ptr()
            launch('rsa_channel_attention_weights', cpar)
            plan.count_launches(1)  # two kernels
            for bi in range(n):  # attn @ v: the weights differ per image
                wts = ops.ConvWeights(wdyn[heads][bi], zero_bias, heads * HEAD_PAD, heads * HEAD_PAD, hp, 1, bprod, fmt=bfmt)
                src = Planes(qkv_pl.hi[bi : bi + 1], None if qkv_pl.lo is None else qkv_pl.lo[bi : bi + 1])
                dst = Planes(att_pl.hi[bi : bi + 1], None if att_pl.lo is None else att_pl.lo[bi : bi + 1])
                plan.conv(ops.conv_params(wts, src, H, Wd, in_plane0=2 * hp, cin_planes=hp, out=dst))

        def dwconv(weights, src, src_plane0, planes, out, act=L.ACT_NONE, stats_t=None, gamma=None, beta=None, mul=None, mul_plane0=0):
            dp = L.DwConvParams()
            dp.batch, dp.H, dp.W, dp.planes, dp.act = n, H, Wd, planes, act
            dp.fmt = src.fmt  # (source, multiplier and output planes of a call share their format)
            dp.in_hi, dp.in_lo = src.hi_ptr(src_plane0), src.lo_ptr(src_plane0)
            dp.in_plane_stride, dp.in_batch_stride = src.plane_stride, src.batch_stride
            dp.weight, dp.bias = weights[0].data_ptr(), weights[1].data_ptr()
            if stats_t is not None:
                dp.stats, dp.gamma, dp.beta = stats_t.data_ptr(), gamma.data_ptr(), beta.data_ptr()
            if mul is not None:
                dp.mul_hi, dp.mul_lo = mul.hi_ptr(mul_plane0), mul.lo_ptr(mul_plane0)
                dp.mul_plane_stride, dp.mul_batch_stride = mul.plane_stride, mul.batch_stride
            dp.out_hi, dp.out_lo = out.hi_ptr(), out.lo_ptr()
            dp.out_plane_stride, dp.out_batch_stride = out.plane_stride, out.batch_stride
            launch('rsa_dwconv3x3', dp)

        def channel_gate(a, src, heads):
            w1, b1, w2, b2 = W[f'{a}.ci']
            gp = L.ChannelGateParams()
            gp.batch, gp.H, gp.W, gp.planes, gp.hidden = n, H, Wd, heads * 4, w1.shape[0]
            gp.fmt = src.fmt
            gp.in_hi, gp.in_lo = src.hi_ptr(), src.lo_ptr()
            gp.in_plane_stride, gp.in_batch_stride = src.plane_stride, src.batch_stride
            gp.w1, gp.b1, gp.w2, gp.b2 = w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr()
            gp.workspace, gp.gate = ws_gate.data_ptr(), gate.data_ptr()
            launch('rsa_channel_gate', gp)
            plan.count_launches(1)  # two kernels

        def aim_combine(a, heads, mode):
            w1, b1, w2, b2 = W[f'{a}.si']
            ap = L.AimParams()
            ap.batch, ap.H, ap.W, ap.planes, ap.hidden, ap.mode = n, H, Wd, heads * 4, w1.shape[0], mode
            ap.fmt = att_pl.fmt
            ap.att_hi, ap.att_lo = att_pl.hi_ptr(), att_pl.lo_ptr()
            ap.att_plane_stride, ap.att_batch_stride = att_pl.plane_stride, att_pl.batch_stride
            ap.conv_hi, ap.conv_lo = conv_pl.hi_ptr(), conv_pl.lo_ptr()
            ap.conv_plane_stride, ap.conv_batch_stride = conv_pl.plane_stride, conv_pl.batch_stride
            ap.gate, ap.w1, ap.b1, ap.w2, ap.b2 = gate.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2
            ap.out_hi, ap.out_lo = comb_pl.hi_ptr(), comb_pl.lo_ptr()
            ap.out_plane_stride, ap.out_batch_stride = comb_pl.plane_stride, comb_pl.batch_stride
            launch('rsa_aim_combine', ap)

        def plane_stats(src, plane0, channels):
            def run():
                L.check(lib.rsa_plane_stats_fmt(src.hi_ptr(plane0), src.lo_ptr(plane0), src.plane_stride, src.batch_stride, n, H, Wd, channels, 1e-5,
                                                src.fmt, stats.data_ptr(), stream()), 'rsa_plane_stats')  # fmt: skip

            plan.call(run)
            plan.count_launches(1)

        def resi_conv(name, src_planes, res, out_f32=None, out_planes=None):
            if self.resi == '1conv':
                plan.conv(ops.conv_params(W[name], src_planes, H, Wd, cin_planes=cp, res1=res, alpha=1.0, out_f32=out_f32, out=out_planes))
            else:
                lre = dict(act=L.ACT_LRELU, act_param=0.2)
                plan.conv(ops.conv_params(W[f'{name}.0'], src_planes, H, Wd, cin_planes=cp, out=q4_a, **lre))
                plan.conv(ops.conv_params(W[f'{name}.2'], q4_a, H, Wd, out=q4_b, **lre))
                plan.conv(ops.conv_params(W[f'{name}.4'], q4_b, H, Wd, res1=res, alpha=1.0, out_f32=out_f32, out=out_planes))

        plan.conv(ops.conv_params(W['conv_first'], x_pl, H, Wd, out_f32=first))
        free = list(pool)
        cur = free.pop()
        layernorm('before_RG.1', first, out_f32=cur)
        for i, d in enumerate(self.depth):
            heads = self.num_heads[i]
            hp = heads * 4
            rg_in = cur
            for j in range(d):
                b = f'layers.{i}.blocks.{j}'
                a = f'{b}.attn'
                layernorm(f'{b}.norm1', cur, out_planes=a_pl)
                plan.conv(ops.conv_params(W[f'{a}.qkv'], a_pl, H, Wd, cin_planes=cp, out=qkv_pl))
                if j % 2 == 0:
                    rect_attention(b, heads, is_shifted(i, j))
                    dwconv(W[f'{a}.dw'], qkv_pl, 2 * hp, hp, conv_pl, act=L.ACT_GELU)
                    channel_gate(a, conv_pl, heads)
                    aim_combine(a, heads, 0)
                else:
                    channel_attention(b, heads)
                    dwconv(W[f'{a}.dw'], qkv_pl, 2 * hp, hp, conv_pl, act=L.ACT_GELU)
                    channel_gate(a, att_pl, heads)
                    aim_combine(a, heads, 1)
                x1 = free.pop()
                plan.conv(ops.conv_params(W[f'{a}.proj'], comb_pl, H, Wd, cin_planes=hp, res1=cur, alpha=1.0, out_f32=x1))
                layernorm(f'{b}.norm2', x1, out_planes=a_pl)
                plan.conv(ops.conv_params(W[f'{b}.ffn.fc1'], a_pl, H, Wd, cin_planes=cp, act=L.ACT_GELU, out=hid_pl))
                sgw, sgb, sgg, sgbeta = W[f'{b}.ffn.sg']
                plane_stats(hid_pl, P1, half)
                dwconv((sgw, sgb), hid_pl, P1, P1, gate_pl, stats_t=stats, gamma=sgg, beta=sgbeta, mul=hid_pl, mul_plane0=0)
                x2 = free.pop()
                last = j == d - 1
                plan.conv(ops.conv_params(W[f'{b}.ffn.fc2'], gate_pl, H, Wd, cin_planes=P1, res1=x1, alpha=1.0, out_f32=x2,
                                          out=body_pl if last else None))  # fmt: skip
                if cur is not rg_in:
                    free.append(cur)
                free.append(x1)
                cur = x2
            out = free.pop()
            resi_conv(f'layers.{i}.conv', body_pl, rg_in, out_f32=out)
            free.append(rg_in)
            if cur is not rg_in:
                free.append(cur)
            cur = out
        layernorm('norm', cur, out_planes=n_pl)
        resi_conv('conv_after_body', n_pl, first, out_planes=body_pl)  # + conv_first output (arch.py:981, 986)

        out_shape = (n, self.in_chans, H * s, Wd * s)
        out_buf = {'y': torch.empty(out_shape, dtype=dtype, device=dev)}
        final = dict(out_scale=1.0 / self.img_range, out_shift=mean)  # x / img_range + mean (arch.py:989)
        if self.upsampler == 'pixelshuffle':
            y = plan.planes(n, 8, H, Wd, with_lo)
            plan.conv(ops.conv_params(W['conv_before_upsample.0'], body_pl, H, Wd, cin_planes=cp, act=L.ACT_LRELU, act_param=0.01, out=y))
            hh, ww = H, Wd
            i = 0
            while f'upsample.{i}' in W:
                r = math.isqrt(W[f'upsample.{i}'].cout // 64)
                shuffled = torch.empty((n, 64, hh * r, ww * r), dtype=torch.float32, device=dev)
                plan.keep.append(shuffled)
                plan.conv(ops.conv_params(W[f'upsample.{i}'], y, hh, ww, out_nchw=shuffled, pixel_shuffle=r))
                hh, ww = hh * r, ww * r
                ny = plan.planes(n, 8, hh, ww, with_lo)
                plan.call(lambda src=shuffled, dst=ny: ops.nchw_to_planes(src, dst))
                y = ny
                i += 2
            plan.conv(ops.conv_params(W['conv_last'], y, hh, ww, out_nchw=out_buf['y'], **final))
        else:
            plan.conv(ops.conv_params(W['upsample.0'], body_pl, H, Wd, cin_planes=cp, out_nchw=out_buf['y'], pixel_shuffle=s, **final))
        arr = plan.flush()
        last_entry = arr[len(arr) - 1]

        def prepare_output():
            if 'y' not in out_buf:
                out_buf['y'] = torch.empty(out_shape, dtype=dtype, device=dev)
            last_entry.out_nchw = out_buf['y'].data_ptr()

        plan.steps.insert(len(plan.steps) - 1, prepare_output)

        def get_output():
            return out_buf.pop('y')

        return set_input, get_output
