"""DRCT on the GPU engine: the wide-head window attention kernel against torch, the model against vectors of the
real reference (tests/golden/drct_*.npz) and against the oracle at another size."""

import ctypes as C

import pytest
import torch

import resselt_amd
from resselt_amd.archs.dat.arch import bias_fragments
from resselt_amd.archs.drct.arch import regroup_qkv_wide
from resselt_amd.engine import lib as L
from resselt_amd.engine import ops, tensors
from resselt_amd.utils import synth

from helpers import golden_names, load_golden, oracle_forward, synth_state_dict

pytestmark = pytest.mark.gpu


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * 2 - 1) * scale


@pytest.mark.parametrize('products,tol', [(3, 3e-5), (1, 2e-2)])
@pytest.mark.parametrize('win,heads,hd,H,W,shift', [(16, 4, 53, 32, 48, 0), (16, 2, 122, 32, 32, 8), (16, 6, 46, 48, 32, 8), (8, 2, 62, 24, 40, 4), (16, 4, 77, 16, 32, 0)])
def test_wide_head_window_attention_kernel(device, products, tol, win, heads, hd, H, W, shift):
    """softmax(q k^T * hd^-0.5 + bias (+ shift mask)) v per (shifted) window and head, head_dim 46..122 (head_chunks 2..4), against the
    reference formulation (archs/drct/arch.py:158-198, 407-474: roll, window_partition, calculate_mask) in torch fp32."""
    from oracle.swinir import shift_mask, window_partition, window_reverse

    B, Cdim, n = 1, heads * hd, win * win
    chunks = -(-hd // 32)
    pad = 32 * chunks
    qkv = _rand((B, 3 * Cdim, H, W), 1, 1.5)
    table = _rand(((2 * win - 1) ** 2, heads), 2, 1.0)
    from resselt_amd.archs.swinir.arch import relative_position_index

    idx = relative_position_index(win)
    dense = table[idx.reshape(-1)].reshape(n, n, heads).permute(2, 0, 1).contiguous()
    # reference: tokens -> (rolled) windows -> attention -> reverse -> roll back
    t = qkv.permute(0, 2, 3, 1)  # B H W 3C
    if shift:
        t = torch.roll(t, shifts=(-shift, -shift), dims=(1, 2))
    wv = window_partition(t, win)  # [nW, n, 3C]
    q, k, v = (wv[..., i * Cdim : (i + 1) * Cdim].reshape(-1, n, heads, hd).permute(0, 2, 1, 3) for i in range(3))
    attn = (q * hd**-0.5) @ k.transpose(-2, -1) + dense.unsqueeze(0)
    if shift:
        mask = shift_mask(H, W, win, shift)
        attn = (attn.view(B, mask.shape[0], heads, n, n) + mask.unsqueeze(1).unsqueeze(0)).view(-1, heads, n, n)
    out = (attn.softmax(-1) @ v).transpose(1, 2).reshape(-1, n, Cdim)
    r = window_reverse(out, win, H, W)
    if shift:
        r = torch.roll(r, shifts=(shift, shift), dims=(1, 2))
    ref = r.permute(0, 3, 1, 2)  # B C H W, channel = head*hd + d

    # engine layout: [which][head][pad] channels, q pre-scaled (what regroup_qkv_wide does to the Linear weights; here on the activations)
    eye = torch.eye(Cdim)
    wq, _ = regroup_qkv_wide(torch.cat([eye, eye, eye], 0), None, heads, pad)  # [3*heads*pad, C] selector (q rows scaled)
    sel = wq.reshape(3, heads * pad, Cdim)
    packed = torch.cat([torch.einsum('oc,bchw->bohw', sel[i], qkv[:, i * Cdim : (i + 1) * Cdim]) for i in range(3)], 1)
    qkv_pl = tensors.nchw_to_planes(packed.to(device), with_lo=products == 3)
    o_pl = tensors.Planes.empty(B, heads * pad // 8, H, W, device, products == 3)
    frag = bias_fragments(dense).to(device)
    ap = L.RectAttnParams()
    ap.batch, ap.H, ap.W, ap.Hp, ap.Wp = B, H, W, H, W
    ap.win_h = ap.win_w = win
    ap.shift_h = ap.shift_w = shift
    ap.heads, ap.head0, ap.heads_total, ap.products, ap.head_chunks = heads, 0, heads, products, chunks
    ap.qkv_hi, ap.qkv_lo, ap.qkv_plane_stride, ap.qkv_batch_stride = qkv_pl.hi_ptr(), qkv_pl.lo_ptr(), qkv_pl.plane_stride, qkv_pl.batch_stride
    ap.bias_frag = frag.data_ptr()
    ap.out_hi, ap.out_lo, ap.out_plane_stride, ap.out_batch_stride = o_pl.hi_ptr(), o_pl.lo_ptr(), o_pl.plane_stride, o_pl.batch_stride
    L.check(L.load().rsa_rect_attention(C.byref(ap), C.c_void_p(ops.current_stream_ptr(device))), 'rsa_rect_attention')
    torch.cuda.synchronize()
    got = tensors.planes_to_nchw(o_pl, heads * pad).cpu().reshape(B, heads, pad, H, W)
    err = (got[:, :, :hd].reshape(B, Cdim, H, W) - ref).abs().max().item()
    assert err <= tol * max(1.0, ref.abs().max().item()), f'max-abs {err:.3e}'
    assert got[:, :, hd:].abs().max().item() == 0.0  # padded channels of every head stay exact zeros


@pytest.mark.parametrize('name', golden_names('drct_'))
def test_drct_matches_reference_vectors(device, name):
    meta, arr = load_golden(name)
    sd = synth_state_dict(meta)
    m = resselt_amd.load_from_state_dict(dict(sd)).to(device)
    assert type(m).__name__ == 'DRCT'
    assert vars(m.parameters_info) == {k: meta['metadata'][k] for k in ('in_channels', 'out_channels', 'upscale', 'name')}
    y = m(arr['x'].to(device))
    torch.cuda.synchronize()
    assert y.shape == arr['y'].shape
    err = (y.cpu() - arr['y']).abs().max().item()
    print(f'{name}: max-abs {err:.3e}')
    assert err <= 1e-4 * max(1.0, arr['y'].abs().max().item()), f'{name}: max-abs {err:.3e}'


def test_drct_vs_oracle_other_size_and_dtypes(device):
    sd = synth.drct_state_dict(num_layers=2, upscale=2, seed=9)
    x = synth.synth_input((1, 3, 70, 45), seed=9)
    with torch.no_grad():
        ref = oracle_forward(dict(arch='drct'), sd, x)
    m = resselt_amd.load_from_state_dict(dict(sd)).to(device)
    y = m(x.to(device))
    assert (y.cpu() - ref).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item())
    yh = m(x.half().to(device))
    assert yh.dtype == torch.float16 and (yh.float().cpu() - ref).abs().max().item() <= 4e-3 * max(1.0, ref.abs().max().item())
    m.precision = 'bf16'
    assert (m(x.to(device)).cpu() - ref).abs().max().item() <= 3e-2 * max(1.0, ref.abs().max().item())
    x2 = synth.synth_input((1, 3, 70, 45), seed=10)
    for precision, bar in (('bf16x3', 1e-4), ('auto', 1e-4)):
        m.precision = precision
        y1 = m(x.to(device))
        assert (y1.cpu() - ref).abs().max().item() <= bar * max(1.0, ref.abs().max().item()), precision
        yb = m(torch.cat([x, x2]).to(device))  # a batch: the per-image launch list repeated inside one plan
        assert yb.shape[0] == 2 and torch.equal(yb[0:1], y1) and torch.equal(yb[1:2], m(x2.to(device))), precision
    assert m.resolved_precision() == 'mixed'  # the default: Linear layers in one fp16 product
