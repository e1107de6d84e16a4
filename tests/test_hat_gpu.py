"""GPU parity of the HAT path: the cross-window mode of rsa_rect_attention against a plain torch statement of OCAB's attention
(nn.Unfold with zero padding), the gated-add kernel, and whole models against vectors produced by the real reference
(tests/golden/hat_*.npz).  Tolerances as in test_dat_gpu.py.
"""

import ctypes as C

import pytest
import torch
import torch.nn.functional as F

import resselt_amd
from helpers import golden_names, load_golden, oracle_forward, synth_state_dict
from resselt_amd.archs.hat.arch import bias_fragments_qk
from resselt_amd.engine import lib as L
from resselt_amd.engine import ops, tensors
from resselt_amd.utils import synth

pytestmark = pytest.mark.gpu


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * 2 - 1) * scale


def _q16(x):
    hi = x.bfloat16().float()
    return hi + (x - hi).bfloat16().float()


@pytest.mark.parametrize('products,tol', [(3, 1e-4), (1, 3e-2)])
@pytest.mark.parametrize('ws,ext,heads,hd,H,W', [(16, 24, 6, 30, 32, 48), (8, 12, 2, 16, 16, 24), (4, 6, 1, 32, 8, 12)])
def test_cross_window_attention_kernel(device, products, tol, ws, ext, heads, hd, H, W):
    """OCAB's attention (archs/hat/arch.py:414-470): queries of a ws x ws window, keys / values of the ext x ext window around it."""
    from oracle.hat import window_partition, window_reverse

    B, C_ = 2, heads * hd
    q, k, v = (_q16(_rand((B, C_, H, W), s, 1.2)) for s in (1, 2, 3))
    nq, nk = ws * ws, ext * ext
    bias = _rand((heads, nq, nk), 4, 2.0)
    qw = window_partition(q.permute(0, 2, 3, 1), ws).view(-1, nq, heads, hd).permute(0, 2, 1, 3)
    kv = F.unfold(torch.cat((k, v), dim=1), kernel_size=(ext, ext), stride=ws, padding=(ext - ws) // 2)
    nw = kv.shape[-1]
    kv = kv.view(B, 2, C_, nk, nw).permute(1, 0, 4, 3, 2).reshape(2, B * nw, nk, heads, hd).permute(0, 1, 3, 2, 4)
    attn = (qw @ kv[0].transpose(-2, -1) + bias.unsqueeze(0)).softmax(-1)
    ref = window_reverse((attn @ kv[1]).transpose(1, 2).reshape(-1, ws, ws, C_), ws, H, W).permute(0, 3, 1, 2)

    def padded(t):  # [B, C, H, W] -> [B, heads*32, H, W]
        out = torch.zeros((B, heads, 32, H, W))
        out[:, :, :hd] = t.reshape(B, heads, hd, H, W)
        return out.reshape(B, heads * 32, H, W)

    pl = tensors.nchw_to_planes(torch.cat([padded(q), padded(k), padded(v)], dim=1).to(device), with_lo=products == 3)
    out = tensors.Planes.empty(B, heads * 4, H, W, device, with_lo=products == 3)
    frag = bias_fragments_qk(bias, (nq + 31) // 32, (nk + 31) // 32).to(device)
    ap = L.RectAttnParams()
    ap.batch, ap.H, ap.W, ap.Hp, ap.Wp = B, H, W, H, W
    ap.win_h = ap.win_w = ws
    ap.kwin_h = ap.kwin_w = ext
    ap.kpad_h = ap.kpad_w = (ext - ws) // 2
    ap.heads, ap.head0, ap.heads_total, ap.products = heads, 0, heads, products
    ap.qkv_hi, ap.qkv_lo, ap.qkv_plane_stride, ap.qkv_batch_stride = pl.hi_ptr(), pl.lo_ptr(), pl.plane_stride, pl.batch_stride
    ap.bias_frag = frag.data_ptr()
    ap.out_hi, ap.out_lo, ap.out_plane_stride, ap.out_batch_stride = out.hi_ptr(), out.lo_ptr(), out.plane_stride, out.batch_stride
    L.check(L.load().rsa_rect_attention(C.byref(ap), C.c_void_p(ops.current_stream_ptr(device))), 'rsa_rect_attention')
    torch.cuda.synchronize()
    got = tensors.planes_to_nchw(out, heads * 32).cpu().reshape(B, heads, 32, H, W)
    err = (got[:, :, :hd].reshape(B, C_, H, W) - ref).abs().max().item()
    assert err <= tol * max(1.0, ref.abs().max().item()), f'max-abs {err:.3e}'


def test_gated_add_kernel(device):
    n, c, h, w = 2, 60, 9, 13
    x, base = _q16(_rand((n, c, h, w), 1, 2.0)), _rand((n, c, h, w), 2, 3.0)
    gate = torch.zeros((n, 64))
    gate[:, :c] = torch.sigmoid(_rand((n, c), 3, 2.0))
    ref = base + x * gate[:, :c, None, None] * 0.01
    xp = tensors.nchw_to_planes(x.to(device))
    bm, om = tensors.nchw_to_f32map(base.to(device)), tensors.empty_f32map(n, c, h, w, device)
    gd = gate.to(device)
    L.check(L.load().rsa_gated_add(xp.hi_ptr(), xp.lo_ptr(), xp.plane_stride, xp.batch_stride, n, h, w, c, gd.data_ptr(), 0.01, bm.data_ptr(),
                                   om.data_ptr(), C.c_void_p(ops.current_stream_ptr(device))), 'rsa_gated_add')  # fmt: skip
    torch.cuda.synchronize()
    assert (tensors.f32map_to_nchw(om, c).cpu() - ref).abs().max().item() <= 1e-6 * ref.abs().max().item()


@pytest.mark.parametrize('name', golden_names('hat_'))
def test_hat_matches_reference_vectors(device, name):
    meta, arr = load_golden(name)
    sd = synth_state_dict(meta)
    m = resselt_amd.load_from_state_dict(dict(sd)).to(device)
    assert vars(m.parameters_info) == {k: meta['metadata'][k] for k in ('in_channels', 'out_channels', 'upscale', 'name')}
    y = m(arr['x'].to(device))
    torch.cuda.synchronize()
    assert y.shape == arr['y'].shape
    err = (y.cpu() - arr['y']).abs().max().item()
    print(f'{name}: max-abs {err:.3e}')
    assert err <= 3e-4 * max(1.0, arr['y'].abs().max().item()), f'{name}: max-abs {err:.3e}'


def test_hat_vs_oracle_dtypes_and_precision(device):
    """HAT-L-like wiring (embed 180, 6 heads, window 16, 2 groups x 3 blocks), fp16 tensor I/O, plain bf16 mode."""
    sd = synth.hat_state_dict(embed_dim=180, depths=(3, 3), num_heads=(6, 6), window=16, upscale=4, seed=8)
    x = synth.synth_input((1, 3, 45, 60), seed=8)
    with torch.no_grad():
        ref = oracle_forward(dict(arch='hat'), sd, x)
    m = resselt_amd.load_from_state_dict(dict(sd)).to(device)
    y = m(x.to(device))
    assert y.shape == ref.shape == (1, 3, 180, 240)
    err = (y.cpu() - ref).abs().max().item()
    print(f'HAT(2x3) bf16x3 max-abs {err:.3e} (|y|max {ref.abs().max():.2f})')
    assert err <= 3e-4 * max(1.0, ref.abs().max().item())
    yh = m(x.to(device).half())
    assert yh.dtype == torch.float16
    m.precision = 'bf16'
    e1 = (m(x.to(device)).cpu() - ref).abs().max().item()
    print(f'HAT(2x3) plain bf16 max-abs {e1:.3e}')
    assert e1 <= 5e-2 * max(1.0, ref.abs().max().item())
