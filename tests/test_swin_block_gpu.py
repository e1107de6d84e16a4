"""GPU parity of the fused Swin block halves (csrc/swin_block.hip) against the CPU oracle's block (oracle/swinir.py::swin_block,
which follows resselt/archs/swinir/arch.py:295-335) and plain torch ops.  Tolerances: bf16x3 2e-5 * scale, plain bf16 3e-2 * scale."""

import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from resselt_amd.archs.swinir.arch import bias_fragments16, regroup_proj, regroup_qkv, relative_position_index
from resselt_amd.engine import lib as L
from resselt_amd.engine import ops, tensors

pytestmark = pytest.mark.gpu


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * 2 - 1) * scale


def _block_sd(C_, heads, hidden, window, seed):
    pre = 'b'
    return {
        f'{pre}.norm1.weight': 1 + _rand((C_,), seed + 1, 0.3), f'{pre}.norm1.bias': _rand((C_,), seed + 2, 0.3),
        f'{pre}.norm2.weight': 1 + _rand((C_,), seed + 3, 0.3), f'{pre}.norm2.bias': _rand((C_,), seed + 4, 0.3),
        f'{pre}.attn.qkv.weight': _rand((3 * C_, C_), seed + 5, 2.0 / C_**0.5), f'{pre}.attn.qkv.bias': _rand((3 * C_,), seed + 6, 0.2),
        f'{pre}.attn.proj.weight': _rand((C_, C_), seed + 7, 1.0 / C_**0.5), f'{pre}.attn.proj.bias': _rand((C_,), seed + 8, 0.2),
        f'{pre}.attn.relative_position_bias_table': _rand(((2 * window - 1) ** 2, heads), seed + 9, 1.0),
        f'{pre}.attn.relative_position_index': relative_position_index(window),
        f'{pre}.mlp.fc1.weight': _rand((hidden, C_), seed + 10, 1.5 / C_**0.5), f'{pre}.mlp.fc1.bias': _rand((hidden,), seed + 11, 0.2),
        f'{pre}.mlp.fc2.weight': _rand((C_, hidden), seed + 12, 1.0 / hidden**0.5), f'{pre}.mlp.fc2.bias': _rand((C_,), seed + 13, 0.2),
    }  # fmt: skip


def _lin(w, b, products, device, cin_planes=None):
    return ops.ConvWeights.from_oihw(w[:, :, None, None], b, products, cin_planes=cin_planes, device=device)


def _run_attn(sd, x, heads, window, shift, products, device, inplace=False):
    n, C_, H, W = x.shape
    xm = tensors.nchw_to_f32map(x.to(device))
    out = xm if inplace else torch.full_like(xm, float('nan'))
    wq, bq = regroup_qkv(sd['b.attn.qkv.weight'], sd['b.attn.qkv.bias'], heads)
    qkv = _lin(wq, bq, products, device)
    proj = _lin(regroup_proj(sd['b.attn.proj.weight'], heads), sd['b.attn.proj.bias'], products, device, cin_planes=heads * 4)
    frag = bias_fragments16(sd['b.attn.relative_position_bias_table'], sd['b.attn.relative_position_index'], window).to(device)
    g, be = sd['b.norm1.weight'].to(device), sd['b.norm1.bias'].to(device)
    ap = L.SwinAttnBlockParams()
    ap.batch, ap.H, ap.W, ap.C, ap.heads, ap.window, ap.shift, ap.products, ap.eps = n, H, W, C_, heads, window, shift, products, 1e-5
    ap.x, ap.gamma, ap.beta = xm.data_ptr(), g.data_ptr(), be.data_ptr()
    ap.wqkv, ap.bqkv, ap.bias_frag16 = qkv.packed_for(0).data_ptr(), qkv.bias.data_ptr(), frag.data_ptr()
    ap.wproj, ap.bproj, ap.out = proj.packed_for(0).data_ptr(), proj.bias.data_ptr(), out.data_ptr()
    L.check(L.load().rsa_swin_attn_block(C.byref(ap), C.c_void_p(ops.current_stream_ptr(device))), 'rsa_swin_attn_block')
    torch.cuda.synchronize()
    return tensors.f32map_to_nchw(out, C_).cpu()


def _run_mlp(sd, x, hidden, products, device, planes=False, inplace=False):
    n, C_, H, W = x.shape
    xm = tensors.nchw_to_f32map(x.to(device))
    out = xm if inplace else torch.full_like(xm, float('nan'))
    fc1 = _lin(sd['b.mlp.fc1.weight'], sd['b.mlp.fc1.bias'], products, device)
    fc2 = _lin(sd['b.mlp.fc2.weight'], sd['b.mlp.fc2.bias'], products, device)
    g, be = sd['b.norm2.weight'].to(device), sd['b.norm2.bias'].to(device)
    mp = L.SwinMlpBlockParams()
    mp.batch, mp.H, mp.W, mp.C, mp.hidden, mp.products, mp.eps = n, H, W, C_, hidden, products, 1e-5
    mp.x, mp.gamma, mp.beta = xm.data_ptr(), g.data_ptr(), be.data_ptr()
    mp.w1, mp.b1, mp.w2, mp.b2 = fc1.packed_for(0).data_ptr(), fc1.bias.data_ptr(), fc2.packed_for(0).data_ptr(), fc2.bias.data_ptr()
    mp.out = out.data_ptr()
    pl = None
    if planes:
        pl = tensors.Planes.empty(n, (C_ + 7) // 8, H, W, device, with_lo=products == 3)
        pl.hi.fill_(float('nan'))
        mp.out_hi, mp.out_lo, mp.out_plane_stride, mp.out_batch_stride = pl.hi_ptr(), pl.lo_ptr(), pl.plane_stride, pl.batch_stride
    L.check(L.load().rsa_swin_mlp_block(C.byref(mp), C.c_void_p(ops.current_stream_ptr(device))), 'rsa_swin_mlp_block')
    torch.cuda.synchronize()
    return tensors.f32map_to_nchw(out, C_).cpu(), pl


def _ref_mlp(sd, x):
    t = x.permute(0, 2, 3, 1)
    y = F.layer_norm(t, (x.shape[1],), sd['b.norm2.weight'], sd['b.norm2.bias'], 1e-5)
    y = F.linear(F.gelu(F.linear(y, sd['b.mlp.fc1.weight'], sd['b.mlp.fc1.bias'])), sd['b.mlp.fc2.weight'], sd['b.mlp.fc2.bias'])
    return (t + y).permute(0, 3, 1, 2)


def _ref_attn(sd, x, heads, window, shift):
    """oracle block with a zero MLP: x + attention half."""
    from oracle.swinir import swin_block

    n, C_, H, W = x.shape
    sd0 = dict(sd)
    sd0['b.mlp.fc2.weight'] = torch.zeros_like(sd['b.mlp.fc2.weight'])
    sd0['b.mlp.fc2.bias'] = torch.zeros_like(sd['b.mlp.fc2.bias'])
    t = swin_block(sd0, 'b', x.permute(0, 2, 3, 1).reshape(n, H * W, C_), H, W, window, shift, heads)
    return t.reshape(n, H, W, C_).permute(0, 3, 1, 2)


TOL = {3: 2e-5, 1: 3e-2}


@pytest.mark.parametrize('products', [3, 1])
@pytest.mark.parametrize(
    'n,C_,hidden,h,w',
    [
        (1, 240, 480, 24, 40),   # SwinIR-L: 15 whole tiles
        (2, 180, 360, 10, 13),   # SwinIR-M: ragged last tile per image, 23 hidden tiles (one padded), 12 output tiles
        (1, 60, 120, 5, 70),     # lightweight: 8 planes of which the last is half empty
        (1, 96, 384, 7, 9),      # mlp_ratio 4; a single partial tile
        (1, 256, 512, 8, 8),     # the widest shape the kernel takes
    ],
)
def test_mlp_block_kernel(device, n, C_, hidden, h, w, products):
    sd = _block_sd(C_, 1, hidden, 8, 100 + C_)
    x = _rand((n, C_, h, w), 7, 2.0) + 0.3
    ref = _ref_mlp(sd, x)
    got, pl = _run_mlp(sd, x, hidden, products, device, planes=True)
    scale = ref.abs().max().item()
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max().item() <= TOL[products] * scale
    full = tensors.planes_to_nchw(pl, pl.planes * 8).cpu()
    assert (full[:, :C_] - ref).abs().max().item() <= (TOL[products] if products == 3 else 4e-2) * scale
    if pl.planes * 8 > C_:
        assert full[:, C_:].abs().max().item() == 0.0
    got2, _ = _run_mlp(sd, x, hidden, products, device, inplace=True)
    assert torch.equal(got2, got)


@pytest.mark.parametrize('products', [3, 1])
@pytest.mark.parametrize(
    'n,C_,heads,window,shift,h,w',
    [
        (1, 240, 8, 8, 0, 16, 24),
        (1, 240, 8, 8, 4, 24, 16),   # shifted: mask on the last window row / column, roll
        (2, 180, 6, 8, 4, 16, 16),   # six waves per workgroup, 12 output tiles
        (1, 60, 6, 8, 4, 16, 8),     # head_dim 10 in 32-channel slots
        (1, 96, 6, 7, 3, 14, 21),    # window 7: 49 tokens, padded keys / queries
        (1, 64, 2, 4, 2, 8, 12),     # window 4, head_dim 32, two waves
        (1, 256, 8, 8, 4, 8, 8),     # one window; every mask region at once
    ],
)
def test_attn_block_kernel(device, n, C_, heads, window, shift, h, w, products):
    sd = _block_sd(C_, heads, 2 * C_, window, 200 + C_ + window)
    x = _rand((n, C_, h, w), 9, 2.0) + 0.3
    ref = _ref_attn(sd, x, heads, window, shift)
    got = _run_attn(sd, x, heads, window, shift, products, device)
    scale = ref.abs().max().item()
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max().item() <= TOL[products] * scale
    got2 = _run_attn(sd, x, heads, window, shift, products, device, inplace=True)
    assert torch.equal(got2, got)


def test_block_halves_compose_to_the_oracle_block(device):
    from oracle.swinir import swin_block

    n, C_, heads, window, h, w = 1, 240, 8, 8, 24, 24
    sd = _block_sd(C_, heads, 480, window, 321)
    x = _rand((n, C_, h, w), 11, 2.0)
    for shift in (0, 4):
        ref = swin_block(sd, 'b', x.permute(0, 2, 3, 1).reshape(n, h * w, C_), h, w, window, shift, heads).reshape(n, h, w, C_).permute(0, 3, 1, 2)
        mid = _run_attn(sd, x, heads, window, shift, 3, device)
        got, _ = _run_mlp(sd, mid, 480, 3, device)
        assert (got - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()


def _run_block(sd, x, heads, window, shift, hidden, products, device, planes=False, inplace=False):
    n, C_, H, W = x.shape
    xm = tensors.nchw_to_f32map(x.to(device))
    out = xm if inplace else torch.full_like(xm, float('nan'))
    wq, bq = regroup_qkv(sd['b.attn.qkv.weight'], sd['b.attn.qkv.bias'], heads)
    qkv = _lin(wq, bq, products, device)
    proj = _lin(regroup_proj(sd['b.attn.proj.weight'], heads), sd['b.attn.proj.bias'], products, device, cin_planes=heads * 4)
    fc1 = _lin(sd['b.mlp.fc1.weight'], sd['b.mlp.fc1.bias'], products, device)
    fc2 = _lin(sd['b.mlp.fc2.weight'], sd['b.mlp.fc2.bias'], products, device)
    frag = bias_fragments16(sd['b.attn.relative_position_bias_table'], sd['b.attn.relative_position_index'], window).to(device)
    g1, be1, g2, be2 = (sd[k].to(device) for k in ('b.norm1.weight', 'b.norm1.bias', 'b.norm2.weight', 'b.norm2.bias'))
    bp = L.SwinBlockParams()
    bp.batch, bp.H, bp.W, bp.C, bp.heads, bp.window, bp.shift, bp.hidden, bp.products, bp.eps = n, H, W, C_, heads, window, shift, hidden, products, 1e-5
    bp.x, bp.gamma1, bp.beta1, bp.gamma2, bp.beta2 = xm.data_ptr(), g1.data_ptr(), be1.data_ptr(), g2.data_ptr(), be2.data_ptr()
    bp.wqkv, bp.bqkv, bp.bias_frag16 = qkv.packed_for(0).data_ptr(), qkv.bias.data_ptr(), frag.data_ptr()
    bp.wproj, bp.bproj = proj.packed_for(0).data_ptr(), proj.bias.data_ptr()
    bp.w1, bp.b1, bp.w2, bp.b2 = fc1.packed_for(0).data_ptr(), fc1.bias.data_ptr(), fc2.packed_for(0).data_ptr(), fc2.bias.data_ptr()
    bp.out = out.data_ptr()
    pl = None
    if planes:
        pl = tensors.Planes.empty(n, (C_ + 7) // 8, H, W, device, with_lo=products == 3)
        pl.hi.fill_(float('nan'))
        bp.out_hi, bp.out_lo, bp.out_plane_stride, bp.out_batch_stride = pl.hi_ptr(), pl.lo_ptr(), pl.plane_stride, pl.batch_stride
    L.check(L.load().rsa_swin_block(C.byref(bp), C.c_void_p(ops.current_stream_ptr(device))), 'rsa_swin_block')
    torch.cuda.synchronize()
    return tensors.f32map_to_nchw(out, C_).cpu(), pl


@pytest.mark.parametrize('products', [3, 1])
@pytest.mark.parametrize(
    'n,C_,heads,hidden,window,shift,h,w',
    [
        (1, 240, 8, 480, 8, 0, 16, 24),   # SwinIR-L
        (1, 240, 8, 480, 8, 4, 24, 16),   # ... shifted
        (2, 180, 6, 360, 8, 4, 16, 16),   # SwinIR-M: two waves without a head, 23 hidden tiles, 12 output tiles
        (1, 60, 6, 120, 8, 4, 16, 8),     # lightweight: head_dim 10, 15 of 16 channel groups per wave pair live
        (1, 96, 6, 384, 7, 3, 14, 21),    # window 7: 49 tokens (padded keys / queries / LayerNorm rows), mlp_ratio 4
        (1, 64, 2, 128, 4, 2, 8, 12),     # window 4, two heads
        (1, 256, 8, 512, 8, 4, 8, 8),     # the widest shape; one window with every mask region
    ],
)
def test_whole_block_kernel(device, n, C_, heads, hidden, window, shift, h, w, products):
    from oracle.swinir import swin_block

    sd = _block_sd(C_, heads, hidden, window, 400 + C_ + window)
    x = _rand((n, C_, h, w), 13, 2.0) + 0.3
    ref = swin_block(sd, 'b', x.permute(0, 2, 3, 1).reshape(n, h * w, C_), h, w, window, shift, heads).reshape(n, h, w, C_).permute(0, 3, 1, 2)
    got, pl = _run_block(sd, x, heads, window, shift, hidden, products, device, planes=True)
    scale = ref.abs().max().item()
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max().item() <= (3e-5 if products == 3 else 4e-2) * scale
    full = tensors.planes_to_nchw(pl, pl.planes * 8).cpu()
    assert (full[:, :C_] - ref).abs().max().item() <= (3e-5 if products == 3 else 5e-2) * scale
    if pl.planes * 8 > C_:
        assert full[:, C_:].abs().max().item() == 0.0
    got2, _ = _run_block(sd, x, heads, window, shift, hidden, products, device, inplace=True)
    assert torch.equal(got2, got)


def test_block_kernels_reject_unsupported_shapes(device):
    lib = L.load()
    ap = L.SwinAttnBlockParams()
    ap.batch, ap.H, ap.W, ap.C, ap.heads, ap.window, ap.shift, ap.products = 1, 8, 8, 288, 9, 8, 0, 3
    assert lib.rsa_swin_attn_block(C.byref(ap), None) == -2  # RSA_E_UNSUPPORTED
    mp = L.SwinMlpBlockParams()
    mp.batch, mp.H, mp.W, mp.C, mp.hidden, mp.products = 1, 8, 8, 240, 960, 3
    assert lib.rsa_swin_mlp_block(C.byref(mp), None) == -2  # RSA_E_UNSUPPORTED
    bp = L.SwinBlockParams()
    bp.batch, bp.H, bp.W, bp.C, bp.heads, bp.window, bp.shift, bp.hidden, bp.products = 1, 16, 16, 240, 8, 16, 0, 480, 3
    assert lib.rsa_swin_block(C.byref(bp), None) == -2  # window 16 (256 tokens) is the rect-attention path's


def test_whole_block_with_a_large_common_offset(device):
    """Tokens whose channels share an offset 200x their spread (mean >> std): norm1 / norm2 are two-pass (centred variance), so the
    result stays at the oracle's; a one-pass E[x^2] - E[x]^2 in f32 would lose the variance here."""
    from oracle.swinir import swin_block

    n, C_, heads, hidden, window, h, w = 1, 240, 8, 480, 8, 16, 16
    sd = _block_sd(C_, heads, hidden, window, 777)
    x = _rand((n, C_, h, w), 21, 0.5) + 100.0
    ref = swin_block(sd, 'b', x.permute(0, 2, 3, 1).reshape(n, h * w, C_), h, w, window, 4, heads).reshape(n, h, w, C_).permute(0, 3, 1, 2)
    got, _ = _run_block(sd, x, heads, window, 4, hidden, 3, device)
    # the block's update (ref - x) is O(1) on top of a residual stream at 100: compare the update, in its own scale
    upd_ref, upd_got = ref - x, got - x
    assert (upd_got - upd_ref).abs().max().item() <= 2e-3 * upd_ref.abs().max().item()
    assert (got - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()
