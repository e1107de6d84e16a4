"""GPU parity of the SwinIR path: LayerNorm kernel, window-attention kernel, and whole models against vectors
produced by the real reference (tests/golden/swinir_*.npz) and the CPU oracle.

Tolerances (bf16x3 mode): LayerNorm 3e-5 * scale, attention 1e-4 * scale, whole models 3e-4 * max(1, max|y|).
"""

import ctypes as C

import pytest
import torch
import torch.nn.functional as F

import resselt_amd
from helpers import golden_names, load_golden, oracle_forward, synth_state_dict
from resselt_amd.archs.swinir.arch import bias_fragments, relative_position_index
from resselt_amd.engine import lib as L
from resselt_amd.engine import ops, tensors
from resselt_amd.utils import synth

pytestmark = pytest.mark.gpu


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * 2 - 1) * scale


@pytest.mark.parametrize('C_,h,w', [(240, 9, 13), (180, 8, 8), (60, 5, 70), (12, 3, 3), (256, 4, 5), (320, 6, 7)])
def test_layernorm_kernel(device, C_, h, w):
    x = _rand((2, C_, h, w), 1, 3.0) + 0.5
    g, b = 1 + _rand((C_,), 2, 0.5), _rand((C_,), 3, 0.5)
    ref = F.layer_norm(x.permute(0, 2, 3, 1), (C_,), g, b, 1e-5).permute(0, 3, 1, 2)
    xm = tensors.nchw_to_f32map(x.to(device))
    out = tensors.Planes.empty(2, (C_ + 7) // 8, h, w, device)
    of32 = tensors.empty_f32map(2, C_, h, w, device)
    lp = L.LayerNormParams()
    lp.batch, lp.H, lp.W, lp.C, lp.eps = 2, h, w, C_, 1e-5
    gd, bd = g.to(device), b.to(device)
    lp.x_f32, lp.gamma, lp.beta = xm.data_ptr(), gd.data_ptr(), bd.data_ptr()
    lp.out_hi, lp.out_lo, lp.out_plane_stride, lp.out_batch_stride = out.hi_ptr(), out.lo_ptr(), out.plane_stride, out.batch_stride
    lp.out_f32 = of32.data_ptr()
    L.check(L.load().rsa_layernorm(C.byref(lp), C.c_void_p(ops.current_stream_ptr(device))), 'rsa_layernorm')
    torch.cuda.synchronize()
    assert (tensors.f32map_to_nchw(of32, C_).cpu() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()
    full = tensors.planes_to_nchw(out, out.planes * 8).cpu()
    assert (full[:, :C_] - ref).abs().max().item() <= 3e-5 * ref.abs().max().item()
    if out.planes * 8 > C_:
        assert full[:, C_:].abs().max().item() == 0.0


def _ref_window_attention(q, k, v, table, window, shift, heads):
    """q, k, v: [B, heads, hd, H, W] (q already scaled). Same data movement as the reference block (arch.py:295-335)."""
    from oracle.swinir import shift_mask, window_partition, window_reverse

    B, _, hd, H, W = q.shape
    N = window * window

    def to_windows(t):
        t = t.permute(0, 3, 4, 1, 2).reshape(B, H, W, heads * hd)
        if shift:
            t = torch.roll(t, shifts=(-shift, -shift), dims=(1, 2))
        return window_partition(t, window).view(-1, N, heads, hd).permute(0, 2, 1, 3)

    qw, kw, vw = to_windows(q), to_windows(k), to_windows(v)
    attn = qw @ kw.transpose(-2, -1)
    idx = relative_position_index(window).view(-1)
    attn = attn + table[idx].view(N, N, heads).permute(2, 0, 1).unsqueeze(0)
    if shift:
        mask = shift_mask(H, W, window, shift)
        nW = mask.shape[0]
        attn = (attn.view(-1, nW, heads, N, N) + mask.unsqueeze(1).unsqueeze(0)).view(-1, heads, N, N)
    out = (attn.softmax(-1) @ vw).transpose(1, 2).reshape(-1, N, heads * hd)
    t = window_reverse(out, window, H, W)
    if shift:
        t = torch.roll(t, shifts=(shift, shift), dims=(1, 2))
    return t.permute(0, 3, 1, 2).reshape(B, heads, hd, H, W)


@pytest.mark.parametrize('products,tol', [(3, 1e-4), (1, 3e-2)])  # q,k,v enter as 16-bit (hi+lo) values and logits reach +-15 here
@pytest.mark.parametrize('window,shift,heads,hd,H,W', [(8, 0, 3, 30, 16, 24), (8, 4, 3, 30, 16, 24), (8, 4, 2, 32, 8, 8), (7, 3, 2, 10, 14, 21), (4, 2, 1, 8, 8, 12)])
def test_window_attention_kernel(device, products, tol, window, shift, heads, hd, H, W):
    B = 2
    q = _rand((B, heads, hd, H, W), 11, 1.5)
    k = _rand((B, heads, hd, H, W), 12, 1.5)
    v = _rand((B, heads, hd, H, W), 13, 2.0)
    table = _rand(((2 * window - 1) ** 2, heads), 14, 1.0)
    ref = _ref_window_attention(q, k, v, table, window, shift, heads)

    def pad32(t):
        out = torch.zeros((B, heads, 32, H, W))
        out[:, :, :hd] = t
        return out

    qkv = torch.stack([pad32(q), pad32(k), pad32(v)], 1).reshape(B, 3 * heads * 32, H, W)
    qkv_pl = tensors.nchw_to_planes(qkv.to(device))
    o_pl = tensors.Planes.empty(B, heads * 4, H, W, device)
    frag = bias_fragments(table.to(device), relative_position_index(window).to(device), window)
    ap = L.WindowAttnParams()
    ap.batch, ap.H, ap.W, ap.heads, ap.window, ap.shift, ap.products = B, H, W, heads, window, shift, products
    ap.qkv_hi, ap.qkv_lo, ap.qkv_plane_stride, ap.qkv_batch_stride = qkv_pl.hi_ptr(), qkv_pl.lo_ptr(), qkv_pl.plane_stride, qkv_pl.batch_stride
    ap.bias_frag = frag.data_ptr()
    ap.out_hi, ap.out_lo, ap.out_plane_stride, ap.out_batch_stride = o_pl.hi_ptr(), o_pl.lo_ptr(), o_pl.plane_stride, o_pl.batch_stride
    L.check(L.load().rsa_window_attention(C.byref(ap), C.c_void_p(ops.current_stream_ptr(device))), 'rsa_window_attention')
    torch.cuda.synchronize()
    got = tensors.planes_to_nchw(o_pl, heads * 32).cpu().reshape(B, heads, 32, H, W)
    err = (got[:, :, :hd] - ref).abs().max().item()
    assert err <= tol * ref.abs().max().item(), f'max-abs {err:.3e}'
    if hd < 32:
        assert got[:, :, hd:].abs().max().item() == 0.0  # padded channels stay exact zeros


@pytest.mark.parametrize('precision', ['auto', 'bf16x3'])
@pytest.mark.parametrize('name', golden_names('swinir_'))
def test_swinir_matches_reference_vectors(device, name, precision):
    meta, arr = load_golden(name)
    sd = synth_state_dict(meta)
    m = resselt_amd.load_from_state_dict(dict(sd)).to(device)
    assert m.resolved_precision() == 'mixed'  # every fixture runs on the whole-block kernel: 'auto' is the per-layer table
    m.precision = precision
    assert vars(m.parameters_info) == {k: meta['metadata'][k] for k in ('in_channels', 'out_channels', 'upscale', 'name')}
    y = m(arr['x'].to(device))
    torch.cuda.synchronize()
    assert y.shape == arr['y'].shape
    err = (y.cpu() - arr['y']).abs().max().item()
    print(f'{name} {precision}: max-abs {err:.3e} (|y|max {arr["y"].abs().max():.3f})')
    assert err <= 3e-4 * max(1.0, arr['y'].abs().max().item()), f'{name} {precision}: max-abs {err:.3e}'



def test_swinir_L_vs_oracle_bf16_input(device):
    """SwinIR-L wiring (BASELINE config 4: embed 240, 8 heads, window 8, nearest+conv, 3conv), 3 RSTB x 6 blocks, bf16 tensor I/O."""
    sd = synth.swinir_state_dict(embed_dim=240, depths=[6, 6, 6], num_heads=[8, 8, 8], upscale=4, upsampler='nearest+conv', resi='3conv', seed=3)
    x = synth.synth_input((1, 3, 50, 70), seed=3)  # not a multiple of the window: reflect padding + crop
    with torch.no_grad():
        ref = oracle_forward(dict(arch='swinir'), sd, x)
    m = resselt_amd.load_from_state_dict(dict(sd)).to(device)
    for precision in ('bf16x3', 'auto'):
        m.precision = precision
        y = m(x.to(device))
        assert y.shape == ref.shape == (1, 3, 200, 280)
        err = (y.cpu() - ref).abs().max().item()
        print(f'SwinIR-L(3x6) {precision} ({m.resolved_precision()}) max-abs {err:.3e} (|y|max {ref.abs().max():.2f})')
        assert err <= 3e-4 * max(1.0, ref.abs().max().item())
    yb = m(x.to(device).bfloat16())
    assert yb.dtype == torch.bfloat16
    with torch.no_grad():
        refb = oracle_forward(dict(arch='swinir'), sd, x.bfloat16().float())
    assert (yb.float().cpu() - refb).abs().max().item() <= 1.6e-2 * max(1.0, refb.abs().max().item())
    m.precision = 'bf16'
    e1 = (m(x.to(device)).cpu() - ref).abs().max().item()
    print(f'SwinIR-L(3x6) plain bf16 max-abs {e1:.3e}')
    assert e1 <= 5e-2 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize('window,embed,heads', [(8, 240, 8), (7, 96, 6)])
def test_swinir_block_paths_agree(device, window, embed, heads):
    """The three ways a block can run -- one launch ('whole', csrc/swin_block_full.hip), one per half ('halves', csrc/swin_block.hip)
    and layer by layer (False) -- against the oracle and against each other on the same model."""
    sd = synth.swinir_state_dict(embed_dim=embed, depths=[2, 2], num_heads=[heads, heads], window=window, upscale=2,
                                 upsampler='pixelshuffle', resi='1conv', img_size=8 * window, seed=11)  # fmt: skip
    x = synth.synth_input((1, 3, 5 * window + 3, 4 * window), seed=12)
    with torch.no_grad():
        ref = oracle_forward(dict(arch='swinir'), sd, x)
    m = resselt_amd.load_from_state_dict(dict(sd)).to(device)
    m.precision = 'bf16x3'  # the three paths exist in the three-product mode ('auto' is one fp16 product, on the whole-block kernel only)
    outs = {}
    for mode in ('whole', 'halves', False):
        m.fused_blocks = mode
        m.invalidate()
        outs[mode] = m(x.to(device)).cpu()
        launches = m.launches_per_forward()
        assert (outs[mode] - ref).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item()), mode
        outs[mode, 'launches'] = launches
    assert outs['whole', 'launches'] < outs['halves', 'launches'] < outs[False, 'launches']
    assert (outs['whole'] - outs[False]).abs().max().item() <= 5e-5 * max(1.0, ref.abs().max().item())
