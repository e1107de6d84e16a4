"""Deterministic synthetic checkpoints (there is no network: no real weights can be downloaded).

Every tensor is drawn from uniform(-1/sqrt(fan_in), +1/sqrt(fan_in)) -- PyTorch's default conv/linear
init scale -- by a numpy PCG64 stream keyed by the tensor's *name*, so the same state dict can be rebuilt
bit-for-bit on the GPU box without the reference, and the golden fixtures in ``tests/golden`` only need to
store seeds, inputs and expected outputs.
"""

from __future__ import annotations

import zlib
from collections import OrderedDict

import numpy as np
import torch


def synth_tensor(name: str, shape, fan_in: int, seed: int = 0, scale: float = 1.0) -> torch.Tensor:
    rng = np.random.Generator(np.random.PCG64([zlib.crc32(name.encode()), seed]))
    bound = scale / float(np.sqrt(max(fan_in, 1)))
    a = rng.uniform(-bound, bound, size=tuple(shape)).astype(np.float32)
    return torch.from_numpy(a)


def synth_input(shape, seed: int = 0) -> torch.Tensor:
    """Image-like input in [0, 1), fp32."""
    rng = np.random.Generator(np.random.PCG64([0x1A6E, seed]))
    return torch.from_numpy(rng.random(size=tuple(shape), dtype=np.float32))


def _conv(sd, name, cout, cin, k, seed, bias=True, scale=1.0):
    fan_in = cin * k * k
    sd[f'{name}.weight'] = synth_tensor(f'{name}.weight', (cout, cin, k, k), fan_in, seed, scale)
    if bias:
        sd[f'{name}.bias'] = synth_tensor(f'{name}.bias', (cout,), fan_in, seed, scale)


def rrdbnet_state_dict(in_nc=3, out_nc=3, nf=64, nb=23, gc=32, scale=4, plus=False, seed=0, new_arch=False) -> 'OrderedDict[str, torch.Tensor]':
    """Old-arch (ESRGAN) keys; ``new_arch=True`` renames them to the official Real-ESRGAN spelling."""
    sd: OrderedDict = OrderedDict()
    _conv(sd, 'model.0', nf, in_nc, 3, seed)
    for i in range(nb):
        for r in (1, 2, 3):
            p = f'model.1.sub.{i}.RDB{r}'
            if plus:
                _conv(sd, f'{p}.conv1x1', gc, nf, 1, seed, bias=False)
            for j in range(1, 6):
                _conv(sd, f'{p}.conv{j}.0', gc if j < 5 else nf, nf + (j - 1) * gc, 3, seed)
    _conv(sd, f'model.1.sub.{nb}', nf, nf, 3, seed)
    k = 3
    n_up = {1: 0, 2: 1, 4: 2, 8: 3}[scale]
    for _ in range(n_up):
        _conv(sd, f'model.{k}', nf, nf, 3, seed)
        k += 3
    k -= 1
    _conv(sd, f'model.{k}', nf, nf, 3, seed)
    _conv(sd, f'model.{k + 2}', out_nc, nf, 3, seed)
    if not new_arch:
        return sd
    out: OrderedDict = OrderedDict()
    for key, v in sd.items():
        parts = key.split('.')
        kind = parts[-1]
        if key.startswith('model.0.'):
            out[f'conv_first.{kind}'] = v
        elif key.startswith('model.1.sub.') and len(parts) == 5:
            out[f'conv_body.{kind}'] = v
        elif key.startswith('model.1.sub.'):
            out[f'body.{parts[3]}.rdb{parts[4][3:]}.{parts[5]}.{kind}'] = v
        else:
            idx = int(parts[1])
            if idx == k:
                out[f'conv_hr.{kind}'] = v
            elif idx == k + 2:
                out[f'conv_last.{kind}'] = v
            else:
                out[f'conv_up{idx // 3}.{kind}'] = v
    return out


def rrdbnet_heavy_tailed_state_dict(nb=23, nf=64, gc=32, seed=0, df=3.0, outlier_gain=32.0, outlier_every=5) -> 'OrderedDict[str, torch.Tensor]':
    """RRDBNet checkpoint with the statistics uniform synthetic weights lack: heavy-tailed weights and outlier activation channels.

    * every weight tensor is re-drawn from a Student-t (``df`` degrees of freedom) scaled to the variance of the default uniform
      init, so single weights reach 10-30x the typical magnitude;
    * inside every RDB, every ``outlier_every``-th growth channel of x1..x4 is scaled by ``outlier_gain`` at its producer (weight row
      and bias) and by ``1 / outlier_gain`` at every consumer (weight column).  LeakyReLU is positively homogeneous, so the network
      function is unchanged, but those channels carry activations ``outlier_gain`` times larger than their neighbours -- the
      situation in which a hi/lo operand split and an f32 accumulator are stressed.
    """
    sd = rrdbnet_state_dict(nf=nf, nb=nb, gc=gc, seed=seed)
    for name in list(sd):
        t = sd[name]
        fan_in = t[0].numel() if name.endswith('.weight') else sd[name[: -len('bias')] + 'weight'][0].numel()
        rng = np.random.Generator(np.random.PCG64([zlib.crc32(name.encode()), seed, 0x7A11]))
        a = rng.standard_t(df, size=tuple(t.shape)).astype(np.float32)
        a *= float(np.sqrt(1.0 / (3.0 * fan_in)) / np.sqrt(df / (df - 2.0)))  # variance of uniform(+-1/sqrt(fan_in)) = 1 / (3 fan_in)
        sd[name] = torch.from_numpy(a)
    g = float(outlier_gain)
    for i in range(nb):
        for r in (1, 2, 3):
            p = f'model.1.sub.{i}.RDB{r}'
            for j in range(1, 5):
                chans = list(range((i + r + j) % outlier_every, gc, outlier_every))
                sd[f'{p}.conv{j}.0.weight'][chans] *= g
                sd[f'{p}.conv{j}.0.bias'][chans] *= g
                cols = [nf + (j - 1) * gc + c for c in chans]
                for m in range(j + 1, 6):
                    sd[f'{p}.conv{m}.0.weight'][:, cols] /= g
    return sd


def _conv3xc(sd, name, cout, cin, gain, seed):
    _conv(sd, f'{name}.sk', cout, cin, 1, seed)
    _conv(sd, f'{name}.conv.0', cin * gain, cin, 1, seed)
    _conv(sd, f'{name}.conv.1', cout * gain, cin * gain, 3, seed)
    _conv(sd, f'{name}.conv.2', cout, cout * gain, 1, seed)
    # stored eval_conv is overwritten by the fold on every reference forward; keep a placeholder of the right shape
    _conv(sd, f'{name}.eval_conv', cout, cin, 3, seed)


def _spab(sd, name, c, seed):
    for r in ('c1_r', 'c2_r', 'c3_r'):
        _conv3xc(sd, f'{name}.{r}', c, c, 2, seed)


def spanplus_state_dict(num_in_ch=3, num_out_ch=3, blocks=(4,), feature_channels=48, upscale=4, upsampler='ps', seed=0):
    sd: OrderedDict = OrderedDict()
    fc = feature_channels
    _conv3xc(sd, 'feats.0', fc, num_in_ch, 2, seed)
    for bi, nblk in enumerate(blocks):
        p = f'feats.{bi + 1}'
        _spab(sd, f'{p}.block_1', fc, seed)
        for j in range(nblk):
            _spab(sd, f'{p}.block_n.{j}', fc, seed)
        _spab(sd, f'{p}.block_end', fc, seed)
        _conv3xc(sd, f'{p}.conv_2', fc, fc, 2, seed)
        _conv(sd, f'{p}.conv_cat', fc, fc * 4, 1, seed)
    if upsampler == 'ps':
        _conv(sd, 'upsampler.0', num_in_ch * upscale * upscale, fc, 3, seed)
    else:
        groups = 4
        oc = 2 * groups * upscale * upscale
        _conv(sd, 'upsampler.end_conv', num_out_ch, fc, 1, seed)
        _conv(sd, 'upsampler.offset', oc, fc, 1, seed)
        _conv(sd, 'upsampler.scope', oc, fc, 1, seed, bias=False)
        h = torch.arange((-upscale + 1) / 2, (upscale - 1) / 2 + 1) / upscale
        sd['upsampler.init_pos'] = torch.stack(torch.meshgrid([h, h], indexing='ij')).transpose(1, 2).repeat(1, groups, 1).reshape(1, -1, 1, 1)
    return sd


def span_state_dict(num_in_ch=3, feature_channels=48, upscale=4, seed=0, norm=True):
    sd: OrderedDict = OrderedDict()
    fc = feature_channels
    _conv3xc(sd, 'conv_1', fc, num_in_ch, 2, seed)
    for i in range(1, 7):
        _spab(sd, f'block_{i}', fc, seed)
    _conv(sd, 'conv_cat', fc, fc * 4, 1, seed)
    _conv3xc(sd, 'conv_2', fc, fc, 2, seed)
    _conv(sd, 'upsampler.0', num_in_ch * upscale * upscale, fc, 3, seed)
    if not norm:
        sd['no_norm'] = torch.zeros(1)
    return sd


def swinir_state_dict(in_ch=3, embed_dim=60, depths=(2, 2), num_heads=(6, 6), window=8, mlp_ratio=2.0, upscale=2, upsampler='nearest+conv',
                      resi='1conv', img_size=64, seed=0, scale=1.0):
    """Keys of the reference SwinIR module (archs/swinir/arch.py:735-960) incl. its registered buffers.

    LayerNorm weights are 1 + u, biases and the relative-position table are drawn at the same +-1/sqrt(fan_in) scale.
    """
    sd: OrderedDict = OrderedDict()
    C = embed_dim
    hidden = int(C * mlp_ratio)

    def lin(name, cout, cin, bias=True):
        sd[f'{name}.weight'] = synth_tensor(f'{name}.weight', (cout, cin), cin, seed, scale)
        if bias:
            sd[f'{name}.bias'] = synth_tensor(f'{name}.bias', (cout,), cin, seed, scale)

    def ln(name):
        sd[f'{name}.weight'] = 1.0 + synth_tensor(f'{name}.weight', (C,), 16, seed)
        sd[f'{name}.bias'] = synth_tensor(f'{name}.bias', (C,), 16, seed)

    def resi_conv(name):
        if resi == '1conv':
            _conv(sd, name, C, C, 3, seed)
        else:
            _conv(sd, f'{name}.0', C // 4, C, 3, seed)
            _conv(sd, f'{name}.2', C // 4, C // 4, 1, seed)
            _conv(sd, f'{name}.4', C, C // 4, 3, seed)

    # relative_position_index buffer (arch.py:111-122)
    ch, cw = torch.arange(window), torch.arange(window)
    coords = torch.stack(torch.meshgrid([ch, cw], indexing='ij')).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += window - 1
    rel[:, :, 1] += window - 1
    rel[:, :, 0] *= 2 * window - 1
    rp_index = rel.sum(-1)

    def shift_mask():
        H = W = img_size
        s_ = window // 2
        img = torch.zeros(1, H, W, 1)
        cnt = 0
        for hs in (slice(0, -window), slice(-window, -s_), slice(-s_, None)):
            for ws in (slice(0, -window), slice(-window, -s_), slice(-s_, None)):
                img[:, hs, ws, :] = cnt
                cnt += 1
        mw = img.view(1, H // window, window, W // window, window, 1).permute(0, 1, 3, 2, 4, 5).reshape(-1, window * window)
        d = mw.unsqueeze(1) - mw.unsqueeze(2)
        return torch.where(d != 0, torch.full_like(d, -100.0), torch.zeros_like(d))

    _conv(sd, 'conv_first', C, in_ch, 3, seed)
    ln('patch_embed.norm')
    for i, depth in enumerate(depths):
        for j in range(depth):
            b = f'layers.{i}.residual_group.blocks.{j}'
            if j % 2 == 1:
                sd[f'{b}.attn_mask'] = shift_mask()
            ln(f'{b}.norm1')
            sd[f'{b}.attn.relative_position_bias_table'] = synth_tensor(f'{b}.rpb', ((2 * window - 1) ** 2, num_heads[i]), 4, seed)
            sd[f'{b}.attn.relative_position_index'] = rp_index.clone()
            lin(f'{b}.attn.qkv', 3 * C, C)
            lin(f'{b}.attn.proj', C, C)
            ln(f'{b}.norm2')
            lin(f'{b}.mlp.fc1', hidden, C)
            lin(f'{b}.mlp.fc2', C, hidden)
        resi_conv(f'layers.{i}.conv')
    ln('norm')
    resi_conv('conv_after_body')
    nf = 64
    if upsampler == 'nearest+conv':
        _conv(sd, 'conv_before_upsample.0', nf, C, 3, seed)
        for u in range(1, {2: 1, 4: 2, 8: 3}[upscale] + 1):
            _conv(sd, f'conv_up{u}', nf, nf, 3, seed)
        _conv(sd, 'conv_hr', nf, nf, 3, seed)
        _conv(sd, 'conv_last', in_ch, nf, 3, seed)
    elif upsampler == 'pixelshuffle':
        _conv(sd, 'conv_before_upsample.0', nf, C, 3, seed)
        if upscale == 3:
            _conv(sd, 'upsample.0', 9 * nf, nf, 3, seed)
        else:
            for u in range({2: 1, 4: 2, 8: 3}[upscale]):
                _conv(sd, f'upsample.{2 * u}', 4 * nf, nf, 3, seed)
        _conv(sd, 'conv_last', in_ch, nf, 3, seed)
    elif upsampler == 'pixelshuffledirect':
        _conv(sd, 'upsample.0', upscale * upscale * in_ch, C, 3, seed)
    else:
        _conv(sd, 'conv_last', in_ch, C, 3, seed)
    return sd


def compact_state_dict(num_in_ch=3, num_feat=64, num_conv=16, upscale=4, seed=0):
    """Keys of SRVGGNetCompact (archs/compact/arch.py:36-57): body.{2i} convs, body.{2i+1} PReLU slopes, last conv."""
    sd: OrderedDict = OrderedDict()
    cin = num_in_ch
    for i in range(num_conv + 1):
        _conv(sd, f'body.{2 * i}', num_feat, cin, 3, seed)
        sd[f'body.{2 * i + 1}.weight'] = 0.25 + synth_tensor(f'body.{2 * i + 1}.weight', (num_feat,), 16, seed)  # slopes in (0, 0.5)
        cin = num_feat
    _conv(sd, f'body.{2 * (num_conv + 1)}', num_in_ch * upscale * upscale, num_feat, 3, seed)
    return sd


def dat_geometry(split_size, idx: int):
    """(H_sp, W_sp) of attention branch ``idx`` (archs/dat/arch.py:186-191): branch 1 swaps the rectangle."""
    return (split_size[0], split_size[1]) if idx == 0 else (split_size[1], split_size[0])


def dat_shifted(rg_idx: int, b_idx: int) -> bool:
    """Which DATB blocks shift their windows (archs/dat/arch.py:312, 453)."""
    return (rg_idx % 2 == 0 and b_idx > 0 and (b_idx - 2) % 4 == 0) or (rg_idx % 2 != 0 and b_idx % 4 == 0)


def dat_shift_masks(H: int, W: int, split_size, shift_size):
    """The two additive shift masks [nW, N, N] (archs/dat/arch.py:336-411), one per branch."""
    out = []
    for idx in (0, 1):
        hs, ws = dat_geometry(split_size, idx)
        sh, sw = dat_geometry(shift_size, idx)
        img = torch.zeros(H, W)
        cnt = 0
        for hsl in (slice(0, -hs), slice(-hs, -sh), slice(-sh, None)):
            for wsl in (slice(0, -ws), slice(-ws, -sw), slice(-sw, None)):
                img[hsl, wsl] = cnt
                cnt += 1
        mw = img.view(H // hs, hs, W // ws, ws).permute(0, 2, 1, 3).reshape(-1, hs * ws)
        d = mw.unsqueeze(1) - mw.unsqueeze(2)
        out.append(torch.where(d != 0, torch.full_like(d, -100.0), torch.zeros_like(d)))
    return out


def dat_state_dict(in_chans=3, embed_dim=64, split_size=(2, 4), depth=(2,), num_heads=(4,), expansion_factor=2.0, qkv_bias=True, upscale=2,
                   resi='1conv', upsampler='pixelshuffle', img_size=16, seed=0):  # fmt: skip
    """Keys (parameters AND buffers) of the reference DAT module (archs/dat/arch.py:828-990).

    BatchNorm running statistics are synthetic too (mean ~ u, var in (0.5, 1.5)): the engine implements the eval-mode network.
    """
    sd: OrderedDict = OrderedDict()
    C = embed_dim
    hidden = int(C * expansion_factor)
    split_size = list(split_size)
    shift_size = [split_size[0] // 2, split_size[1] // 2]

    def lin(name, cout, cin, bias=True):
        sd[f'{name}.weight'] = synth_tensor(f'{name}.weight', (cout, cin), cin, seed)
        if bias:
            sd[f'{name}.bias'] = synth_tensor(f'{name}.bias', (cout,), cin, seed)

    def ln(name, c):
        sd[f'{name}.weight'] = 1.0 + synth_tensor(f'{name}.weight', (c,), 16, seed)
        sd[f'{name}.bias'] = synth_tensor(f'{name}.bias', (c,), 16, seed)

    def bn(name, c):
        ln(name, c)
        sd[f'{name}.running_mean'] = synth_tensor(f'{name}.running_mean', (c,), 16, seed)
        sd[f'{name}.running_var'] = 1.0 + 2.0 * synth_tensor(f'{name}.running_var', (c,), 16, seed)
        sd[f'{name}.num_batches_tracked'] = torch.tensor(100, dtype=torch.int64)

    def dw(name, c):
        sd[f'{name}.weight'] = synth_tensor(f'{name}.weight', (c, 1, 3, 3), 9, seed)
        sd[f'{name}.bias'] = synth_tensor(f'{name}.bias', (c,), 9, seed)

    def resi_conv(name):
        if resi == '1conv':
            _conv(sd, name, C, C, 3, seed)
        else:
            _conv(sd, f'{name}.0', C // 4, C, 3, seed)
            _conv(sd, f'{name}.2', C // 4, C // 4, 1, seed)
            _conv(sd, f'{name}.4', C, C // 4, 3, seed)

    def aim(name):
        dw(f'{name}.dwconv.0', C)
        bn(f'{name}.dwconv.1', C)
        _conv(sd, f'{name}.channel_interaction.1', C // 8, C, 1, seed)
        bn(f'{name}.channel_interaction.2', C // 8)
        _conv(sd, f'{name}.channel_interaction.4', C, C // 8, 1, seed)
        _conv(sd, f'{name}.spatial_interaction.0', C // 16, C, 1, seed)
        bn(f'{name}.spatial_interaction.1', C // 16)
        _conv(sd, f'{name}.spatial_interaction.3', 1, C // 16, 1, seed)

    def spatial_branch(name, idx, heads):
        hs, ws = dat_geometry(split_size, idx)
        bh, bw = torch.arange(1 - hs, hs), torch.arange(1 - ws, ws)
        sd[f'{name}.rpe_biases'] = torch.stack(torch.meshgrid([bh, bw], indexing='ij')).flatten(1).transpose(0, 1).contiguous().float()
        coords = torch.stack(torch.meshgrid([torch.arange(hs), torch.arange(ws)], indexing='ij')).flatten(1)
        rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
        rel[:, :, 0] += hs - 1
        rel[:, :, 1] += ws - 1
        rel[:, :, 0] *= 2 * ws - 1
        sd[f'{name}.relative_position_index'] = rel.sum(-1)
        pos_dim = ((C // 2) // 4) // 4  # DynamicPosBias(dim // 4): pos_dim = dim // 4 again (arch.py:116, 194)
        lin(f'{name}.pos.pos_proj', pos_dim, 2)
        for k, cout in (('pos1', pos_dim), ('pos2', pos_dim), ('pos3', heads)):
            sd[f'{name}.pos.{k}.0.weight'] = 1.0 + synth_tensor(f'{name}.pos.{k}.0.weight', (pos_dim,), 16, seed)
            sd[f'{name}.pos.{k}.0.bias'] = synth_tensor(f'{name}.pos.{k}.0.bias', (pos_dim,), 16, seed)
            lin(f'{name}.pos.{k}.2', cout, pos_dim)

    _conv(sd, 'conv_first', C, in_chans, 3, seed)
    ln('before_RG.1', C)
    for i, d in enumerate(depth):
        heads = num_heads[i]
        for j in range(d):
            b = f'layers.{i}.blocks.{j}'
            ln(f'{b}.norm1', C)
            if j % 2 == 0:  # DSTB: adaptive spatial attention
                lin(f'{b}.attn.qkv', 3 * C, C, qkv_bias)
                lin(f'{b}.attn.proj', C, C)
                for idx in (0, 1):
                    spatial_branch(f'{b}.attn.attns.{idx}', idx, heads // 2)
                if dat_shifted(i, j):
                    m0, m1 = dat_shift_masks(img_size, img_size, split_size, shift_size)
                    sd[f'{b}.attn.attn_mask_0'] = m0
                    sd[f'{b}.attn.attn_mask_1'] = m1
            else:  # DCTB: adaptive channel attention
                sd[f'{b}.attn.temperature'] = 1.0 + synth_tensor(f'{b}.attn.temperature', (heads, 1, 1), 4, seed)
                lin(f'{b}.attn.qkv', 3 * C, C, qkv_bias)
                lin(f'{b}.attn.proj', C, C)
            aim(f'{b}.attn')
            lin(f'{b}.ffn.fc1', hidden, C)
            ln(f'{b}.ffn.sg.norm', hidden // 2)
            dw(f'{b}.ffn.sg.conv', hidden // 2)
            lin(f'{b}.ffn.fc2', C, hidden // 2)
            ln(f'{b}.norm2', C)
        resi_conv(f'layers.{i}.conv')
    ln('norm', C)
    resi_conv('conv_after_body')
    if upsampler == 'pixelshuffle':
        _conv(sd, 'conv_before_upsample.0', 64, C, 3, seed)
        if upscale == 3:
            _conv(sd, 'upsample.0', 9 * 64, 64, 3, seed)
        else:
            for u in range({1: 0, 2: 1, 4: 2, 8: 3}[upscale]):
                _conv(sd, f'upsample.{2 * u}', 4 * 64, 64, 3, seed)
        _conv(sd, 'conv_last', in_chans, 64, 3, seed)
    else:
        _conv(sd, 'upsample.0', upscale * upscale * in_chans, C, 3, seed)
    return sd


def _repconv(sd, name, cout, cin, seed):
    """Keys of SpanPP's RepConv (archs/spanpp/arch.py:152-192): SeqConv3x3 (k0, b0, k1, b1), a plain 3x3, a Conv3XC, the fused conv, alpha."""
    mid = 2 * cout
    sd[f'{name}.alpha'] = 1.0 + synth_tensor(f'{name}.alpha', (3,), 4, seed)
    sd[f'{name}.conv1.k0'] = synth_tensor(f'{name}.conv1.k0', (mid, cin, 1, 1), cin, seed)
    sd[f'{name}.conv1.b0'] = synth_tensor(f'{name}.conv1.b0', (mid,), cin, seed)
    sd[f'{name}.conv1.k1'] = synth_tensor(f'{name}.conv1.k1', (cout, mid, 3, 3), mid * 9, seed)
    sd[f'{name}.conv1.b1'] = synth_tensor(f'{name}.conv1.b1', (cout,), mid * 9, seed)
    _conv(sd, f'{name}.conv2', cout, cin, 3, seed)
    _conv3xc(sd, f'{name}.conv3', cout, cin, 2, seed)
    _conv(sd, f'{name}.conv_3x3_rep', cout, cin, 3, seed)


def spanpp_state_dict(num_in_ch=3, feature_channels=48, scale_list=(1, 2, 3, 4), implicit_dim=256, latent_layers=4, seed=0):
    """Keys of the reference SpanPP module (archs/spanpp/arch.py:315-373), ``MetaIGConv`` buffer included."""
    sd: OrderedDict = OrderedDict()
    fc = feature_channels
    _repconv(sd, 'conv0', fc, num_in_ch, seed)
    for i in range(1, 7):
        for r in ('c1_r', 'c2_r', 'c3_r'):
            _repconv(sd, f'block_{i}.{r}', fc, fc, seed)
    _conv(sd, 'conv_cat', fc, 4 * fc, 1, seed)
    _repconv(sd, 'conv_2', fc, fc, seed)
    # freq / amplitude are randn * 0.02 in the reference; a larger spread makes the generated kernels non-trivial
    sd['upsampler.freq'] = synth_tensor('upsampler.freq', (fc * 9, implicit_dim, 1, 1), 1, seed, 0.5)
    sd['upsampler.amplitude'] = synth_tensor('upsampler.amplitude', (fc * 9, implicit_dim, 1, 1), 1, seed, 0.5)
    _conv(sd, 'upsampler.phase', implicit_dim // 2, 1, 1, seed)
    for l in range(latent_layers):
        _conv(sd, f'upsampler.query_kernel.{2 * l}', implicit_dim, implicit_dim, 1, seed)
    _conv(sd, f'upsampler.query_kernel.{2 * latent_layers}', 3, implicit_dim, 1, seed)
    sd['MetaIGConv'] = torch.tensor(sorted(set(scale_list)), dtype=torch.uint8)
    return sd


def hat_rpi(window: int, overlap_ratio: float):
    """The two registered index buffers of HAT (archs/hat/arch.py:987-1034): self-attention and overlapping cross-attention."""
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


def hat_state_dict(in_chans=3, embed_dim=60, depths=(2, 2), num_heads=(6, 6), window=8, compress_ratio=3, squeeze_factor=30, overlap_ratio=0.5,
                   mlp_ratio=2.0, upscale=2, num_feat=64, resi='1conv', seed=0):  # fmt: skip
    """Keys of the reference HAT module (archs/hat/arch.py:798-1110) incl. its two index buffers."""
    sd: OrderedDict = OrderedDict()
    C = embed_dim
    hidden = int(C * mlp_ratio)
    ext = window + int(overlap_ratio * window)

    def lin(name, cout, cin):
        sd[f'{name}.weight'] = synth_tensor(f'{name}.weight', (cout, cin), cin, seed)
        sd[f'{name}.bias'] = synth_tensor(f'{name}.bias', (cout,), cin, seed)

    def ln(name):
        sd[f'{name}.weight'] = 1.0 + synth_tensor(f'{name}.weight', (C,), 16, seed)
        sd[f'{name}.bias'] = synth_tensor(f'{name}.bias', (C,), 16, seed)

    sd['relative_position_index_SA'], sd['relative_position_index_OCA'] = hat_rpi(window, overlap_ratio)
    _conv(sd, 'conv_first', C, in_chans, 3, seed)
    ln('patch_embed.norm')
    for i, depth in enumerate(depths):
        g = f'layers.{i}.residual_group'
        for j in range(depth):
            b = f'{g}.blocks.{j}'
            ln(f'{b}.norm1')
            sd[f'{b}.attn.relative_position_bias_table'] = synth_tensor(f'{b}.rpb', ((2 * window - 1) ** 2, num_heads[i]), 4, seed)
            lin(f'{b}.attn.qkv', 3 * C, C)
            lin(f'{b}.attn.proj', C, C)
            _conv(sd, f'{b}.conv_block.cab.0', C // compress_ratio, C, 3, seed)
            _conv(sd, f'{b}.conv_block.cab.2', C, C // compress_ratio, 3, seed)
            _conv(sd, f'{b}.conv_block.cab.3.attention.1', C // squeeze_factor, C, 1, seed)
            _conv(sd, f'{b}.conv_block.cab.3.attention.3', C, C // squeeze_factor, 1, seed)
            ln(f'{b}.norm2')
            lin(f'{b}.mlp.fc1', hidden, C)
            lin(f'{b}.mlp.fc2', C, hidden)
        o = f'{g}.overlap_attn'
        ln(f'{o}.norm1')
        lin(f'{o}.qkv', 3 * C, C)
        sd[f'{o}.relative_position_bias_table'] = synth_tensor(f'{o}.rpb', ((window + ext - 1) ** 2, num_heads[i]), 4, seed)
        lin(f'{o}.proj', C, C)
        ln(f'{o}.norm2')
        lin(f'{o}.mlp.fc1', hidden, C)
        lin(f'{o}.mlp.fc2', C, hidden)
        if resi == '1conv':
            _conv(sd, f'layers.{i}.conv', C, C, 3, seed)
    ln('norm')
    if resi == '1conv':
        _conv(sd, 'conv_after_body', C, C, 3, seed)
    _conv(sd, 'conv_before_upsample.0', num_feat, C, 3, seed)
    if upscale == 3:
        _conv(sd, 'upsample.0', 9 * num_feat, num_feat, 3, seed)
    else:
        for u in range({1: 0, 2: 1, 4: 2, 8: 3}[upscale]):
            _conv(sd, f'upsample.{2 * u}', 4 * num_feat, num_feat, 3, seed)
    _conv(sd, 'conv_last', in_chans, num_feat, 3, seed)
    return sd


def rtmosr_state_dict(scale=2, dim=32, ffn_expansion=2.0, n_blocks=2, unshuffle_mod=False, dccm=True, se=True, seed=0):
    """Keys of the reference RTMoSR module (archs/rtmosr/arch.py:340-387)."""
    sd: OrderedDict = OrderedDict()
    unshuffle = 0
    s_int = scale
    if scale < 4 and unshuffle_mod:
        unshuffle = 4 // scale
        s_int = 4
    hidden = int(ffn_expansion * dim)
    if unshuffle:
        _repconv(sd, 'to_feat.1', dim, 3 * unshuffle * unshuffle, seed)
    else:
        _repconv(sd, 'to_feat', dim, 3, seed)
    for i in range(n_blocks):
        b = f'body.{i}'
        sd[f'{b}.norm.scale'] = 1.0 + synth_tensor(f'{b}.norm.scale', (dim,), 16, seed)
        sd[f'{b}.norm.offset'] = synth_tensor(f'{b}.norm.offset', (dim,), 16, seed)
        _repconv(sd, f'{b}.fc1', 2 * hidden, dim, seed)
        _repconv(sd, f'{b}.conv.0.poll.1', 4 * dim, dim, seed)
        o = f'{b}.conv.1'
        for k in (1, 2, 3, 4):
            sd[f'{o}.alpha{k}'] = 1.0 + synth_tensor(f'{o}.alpha{k}', (1, 4 * dim, 1, 1), 16, seed)
        for name, ks in (('conv1x1', 1), ('conv3x3', 3), ('conv5x5', 5), ('conv5x5_reparam', 5)):
            sd[f'{o}.{name}.weight'] = synth_tensor(f'{o}.{name}.weight', (4 * dim, 1, ks, ks), ks * ks, seed)
            sd[f'{o}.{name}.bias'] = synth_tensor(f'{o}.{name}.bias', (4 * dim,), ks * ks, seed)
        if se:
            _conv(sd, f'{b}.conv.2.squeezing.0', 2 * dim, 4 * dim, 1, seed)
            _conv(sd, f'{b}.conv.2.squeezing.2', 4 * dim, 2 * dim, 1, seed)
        if dccm:
            _repconv(sd, f'{b}.fc2', dim, hidden, seed)
        else:
            _conv(sd, f'{b}.fc2', dim, hidden, 1, seed)
    _repconv(sd, 'to_img.0', 3 * s_int * s_int, dim, seed)
    return sd


def drct_block_dims(embed_dim: int, gc: int, num_heads: int):
    """(dim, heads, mlp_ratio scale, shifted) of the five Swin blocks of a DRCT dense group (reference archs/drct/arch.py:225-298)."""
    out = []
    for j in range(5):
        dim = embed_dim + j * gc
        heads = num_heads if j == 0 else num_heads - (dim % num_heads)
        out.append((dim, heads, j in (1, 3)))
    return out


def drct_state_dict(in_chans=3, embed_dim=180, num_layers=2, num_heads=6, window=16, mlp_ratio=2.0, gc=32, upscale=2, resi='1conv', img_size=64,
                    seed=0, attn_mask=True):  # fmt: skip
    """Keys of the reference DRCT module (archs/drct/arch.py:617-792) incl. its registered buffers (relative_position_index, attn_mask of
    the shifted blocks swin2 / swin4).  The loader fixes depths = (6,) * num_layers and reads one head count per layer."""
    sd: OrderedDict = OrderedDict()
    C = embed_dim

    def lin(name, cout, cin):
        sd[f'{name}.weight'] = synth_tensor(f'{name}.weight', (cout, cin), cin, seed)
        sd[f'{name}.bias'] = synth_tensor(f'{name}.bias', (cout,), cin, seed)

    def ln(name, c):
        sd[f'{name}.weight'] = 1.0 + synth_tensor(f'{name}.weight', (c,), 16, seed)
        sd[f'{name}.bias'] = synth_tensor(f'{name}.bias', (c,), 16, seed)

    coords = torch.stack(torch.meshgrid([torch.arange(window), torch.arange(window)], indexing='ij')).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += window - 1
    rel[:, :, 1] += window - 1
    rel[:, :, 0] *= 2 * window - 1
    rp_index = rel.sum(-1)

    def shift_mask():
        s_ = window // 2
        img = torch.zeros(1, img_size, img_size, 1)
        cnt = 0
        for hs in (slice(0, -window), slice(-window, -s_), slice(-s_, None)):
            for ws in (slice(0, -window), slice(-window, -s_), slice(-s_, None)):
                img[:, hs, ws, :] = cnt
                cnt += 1
        mw = img.view(1, img_size // window, window, img_size // window, window, 1).permute(0, 1, 3, 2, 4, 5).reshape(-1, window * window)
        d = mw.unsqueeze(1) - mw.unsqueeze(2)
        return torch.where(d != 0, torch.full_like(d, -100.0), torch.zeros_like(d))

    _conv(sd, 'conv_first', C, in_chans, 3, seed)
    ln('patch_embed.norm', C)
    for i in range(num_layers):
        for j, (dim, heads, shifted) in enumerate(drct_block_dims(C, gc, num_heads), start=1):
            b = f'layers.{i}.swin{j}'
            hidden = int(dim * (mlp_ratio if j <= 3 else 1))
            if shifted and attn_mask:  # (a checkpoint saved without the mask buffers loads as img_size = window: no block is shifted)
                sd[f'{b}.attn_mask'] = shift_mask()
            ln(f'{b}.norm1', dim)
            sd[f'{b}.attn.relative_position_bias_table'] = synth_tensor(f'{b}.attn.relative_position_bias_table', ((2 * window - 1) ** 2, heads), 16, seed)
            sd[f'{b}.attn.relative_position_index'] = rp_index.clone()
            lin(f'{b}.attn.qkv', 3 * dim, dim)
            lin(f'{b}.attn.proj', dim, dim)
            ln(f'{b}.norm2', dim)
            lin(f'{b}.mlp.fc1', hidden, dim)
            lin(f'{b}.mlp.fc2', dim, hidden)
            _conv(sd, f'layers.{i}.adjust{j}', gc if j < 5 else C, dim, 1, seed)
    ln('norm', C)
    if resi == '1conv':
        _conv(sd, 'conv_after_body', C, C, 3, seed)
    _conv(sd, 'conv_before_upsample.0', 64, C, 3, seed)
    if upscale == 3:
        _conv(sd, 'upsample.0', 9 * 64, 64, 3, seed)
    else:
        for u in range(int(np.log2(upscale))):
            _conv(sd, f'upsample.{2 * u}', 4 * 64, 64, 3, seed)
    _conv(sd, 'conv_last', in_chans, 64, 3, seed)
    return sd
