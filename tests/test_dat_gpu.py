"""GPU parity of the DAT path: each kernel of csrc/dat.hip against a plain fp32 torch statement of the same op, the
rectangular window attention against the oracle's restatement of Adaptive_Spatial_Attention, and whole models against vectors
produced by the real reference (tests/golden/dat_*.npz).

Tolerances (bf16x3 mode, inputs quantised to hi+lo bf16 = 16 mantissa bits): elementwise kernels 3e-5 * scale,
attention 1e-4 * scale, whole models 3e-4 * max(1, max|y|).
"""

import ctypes as C

import pytest
import torch
import torch.nn.functional as F

import resselt_amd
from helpers import golden_names, load_golden, oracle_forward, synth_state_dict
from resselt_amd.archs.dat.arch import bias_fragments, branch_geometry
from resselt_amd.engine import lib as L
from resselt_amd.engine import ops, tensors
from resselt_amd.engine.pack import pad_bias
from resselt_amd.utils import synth

pytestmark = pytest.mark.gpu


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * 2 - 1) * scale


def _q16(x):
    """What the kernels see: the value rounded to hi+lo bf16."""
    hi = x.bfloat16().float()
    return hi + (x - hi).bfloat16().float()


def _stream(device):
    return C.c_void_p(ops.current_stream_ptr(device))


@pytest.mark.parametrize('norm,mul,gelu', [(False, False, True), (True, True, False), (False, False, False)])
@pytest.mark.parametrize('n,c,h,w', [(2, 24, 9, 13), (1, 96, 17, 40)])
def test_dwconv_kernel(device, norm, mul, gelu, n, c, h, w):
    x = _q16(_rand((n, c, h, w), 1, 2.0))
    wt, b = _rand((c, 1, 3, 3), 2, 0.4), _rand((c,), 3, 0.2)
    g, beta = 1 + _rand((c,), 4, 0.3), _rand((c,), 5, 0.2)
    m = _q16(_rand((n, c, h, w), 6, 1.5))
    src = x
    if norm:
        src = F.layer_norm(x.permute(0, 2, 3, 1), (c,), g, beta, 1e-5).permute(0, 3, 1, 2)
    ref = F.conv2d(src, wt, b, padding=1, groups=c)
    if gelu:
        ref = F.gelu(ref)
    if mul:
        ref = ref * m
    xp, mp = tensors.nchw_to_planes(x.to(device)), tensors.nchw_to_planes(m.to(device))
    out = tensors.Planes.empty(n, c // 8, h, w, device)
    keep = [t.to(device).contiguous() for t in (wt.reshape(c, 9), b, g, beta)]
    stats = torch.empty((n, h * w, 2), dtype=torch.float32, device=device)
    lib = L.load()
    if norm:
        L.check(lib.rsa_plane_stats(xp.hi_ptr(), xp.lo_ptr(), xp.plane_stride, xp.batch_stride, n, h, w, c, 1e-5, stats.data_ptr(), _stream(device)),
                'rsa_plane_stats')  # fmt: skip
    dp = L.DwConvParams()
    dp.batch, dp.H, dp.W, dp.planes, dp.act = n, h, w, c // 8, L.ACT_GELU if gelu else L.ACT_NONE
    dp.in_hi, dp.in_lo, dp.in_plane_stride, dp.in_batch_stride = xp.hi_ptr(), xp.lo_ptr(), xp.plane_stride, xp.batch_stride
    dp.weight, dp.bias = keep[0].data_ptr(), keep[1].data_ptr()
    if norm:
        dp.stats, dp.gamma, dp.beta = stats.data_ptr(), keep[2].data_ptr(), keep[3].data_ptr()
    if mul:
        dp.mul_hi, dp.mul_lo, dp.mul_plane_stride, dp.mul_batch_stride = mp.hi_ptr(), mp.lo_ptr(), mp.plane_stride, mp.batch_stride
    dp.out_hi, dp.out_lo, dp.out_plane_stride, dp.out_batch_stride = out.hi_ptr(), out.lo_ptr(), out.plane_stride, out.batch_stride
    L.check(lib.rsa_dwconv3x3(C.byref(dp), _stream(device)), 'rsa_dwconv3x3')
    torch.cuda.synchronize()
    got = tensors.planes_to_nchw(out, c).cpu()
    err = (got - ref).abs().max().item()
    assert err <= 3e-5 * max(1.0, ref.abs().max().item()), f'max-abs {err:.3e}'


@pytest.mark.parametrize('n,planes,hidden,h,w', [(2, 8, 8, 9, 13), (1, 24, 22, 70, 90)])
def test_channel_gate_kernel(device, n, planes, hidden, h, w):
    c = planes * 8
    x = _q16(_rand((n, c, h, w), 1, 2.0) + 0.3)
    w1, b1, w2, b2 = _rand((hidden, c), 2, 0.2), _rand((hidden,), 3, 0.2), _rand((c, hidden), 4, 0.4), _rand((c,), 5, 0.2)
    ref = torch.sigmoid(F.linear(F.gelu(F.linear(x.mean(dim=(2, 3)), w1, b1)), w2, b2))
    xp = tensors.nchw_to_planes(x.to(device))
    lib = L.load()
    ws = torch.empty((int(lib.rsa_channel_gate_workspace_bytes(n, h, w, planes)) // 4,), dtype=torch.float32, device=device)
    gate = torch.empty((n, c), dtype=torch.float32, device=device)
    keep = [t.to(device).contiguous() for t in (w1, b1, w2, b2)]
    gp = L.ChannelGateParams()
    gp.batch, gp.H, gp.W, gp.planes, gp.hidden = n, h, w, planes, hidden
    gp.in_hi, gp.in_lo, gp.in_plane_stride, gp.in_batch_stride = xp.hi_ptr(), xp.lo_ptr(), xp.plane_stride, xp.batch_stride
    gp.w1, gp.b1, gp.w2, gp.b2 = (t.data_ptr() for t in keep)
    gp.workspace, gp.gate = ws.data_ptr(), gate.data_ptr()
    L.check(lib.rsa_channel_gate(C.byref(gp), _stream(device)), 'rsa_channel_gate')
    torch.cuda.synchronize()
    assert (gate.cpu() - ref).abs().max().item() <= 2e-6


@pytest.mark.parametrize('mode', [0, 1])
@pytest.mark.parametrize('n,planes,hidden,h,w', [(2, 8, 4, 9, 13), (1, 24, 11, 33, 20)])
def test_aim_combine_kernel(device, mode, n, planes, hidden, h, w):
    c = planes * 8
    att, conv = _q16(_rand((n, c, h, w), 1, 2.0)), _q16(_rand((n, c, h, w), 2, 2.0))
    gate = torch.sigmoid(_rand((n, c), 3, 2.0))
    w1, b1, w2, b2 = _rand((hidden, c), 4, 0.2), _rand((hidden,), 5, 0.2), _rand((hidden,), 6, 0.5), 0.1
    src = att if mode == 0 else conv
    s = F.conv2d(F.gelu(F.conv2d(src, w1[:, :, None, None], b1)), w2[None, :, None, None], torch.tensor([b2]))
    g = gate[:, :, None, None]
    ref = att * g + torch.sigmoid(s) * conv if mode == 0 else att * torch.sigmoid(s) + conv * g
    ap_, cp_ = tensors.nchw_to_planes(att.to(device)), tensors.nchw_to_planes(conv.to(device))
    out = tensors.Planes.empty(n, planes, h, w, device)
    keep = [t.to(device).contiguous() for t in (gate, w1, b1, w2)]
    ap = L.AimParams()
    ap.batch, ap.H, ap.W, ap.planes, ap.hidden, ap.mode = n, h, w, planes, hidden, mode
    ap.att_hi, ap.att_lo, ap.att_plane_stride, ap.att_batch_stride = ap_.hi_ptr(), ap_.lo_ptr(), ap_.plane_stride, ap_.batch_stride
    ap.conv_hi, ap.conv_lo, ap.conv_plane_stride, ap.conv_batch_stride = cp_.hi_ptr(), cp_.lo_ptr(), cp_.plane_stride, cp_.batch_stride
    ap.gate, ap.w1, ap.b1, ap.w2, ap.b2 = keep[0].data_ptr(), keep[1].data_ptr(), keep[2].data_ptr(), keep[3].data_ptr(), b2
    ap.out_hi, ap.out_lo, ap.out_plane_stride, ap.out_batch_stride = out.hi_ptr(), out.lo_ptr(), out.plane_stride, out.batch_stride
    L.check(L.load().rsa_aim_combine(C.byref(ap), _stream(device)), 'rsa_aim_combine')
    torch.cuda.synchronize()
    err = (tensors.planes_to_nchw(out, c).cpu() - ref).abs().max().item()
    assert err <= 3e-5 * max(1.0, ref.abs().max().item()), f'max-abs {err:.3e}'


@pytest.mark.parametrize('n,heads,hd,h,w', [(2, 2, 16, 9, 13), (1, 6, 30, 50, 70), (1, 6, 10, 3, 5)])
def test_channel_attention_weights_and_apply(device, n, heads, hd, h, w):
    """softmax(normalize(q) normalize(k)^T * temperature) @ v over all tokens (arch.py:577-588): packed weights + one k1 conv."""
    N = h * w
    q, k, v = (_q16(_rand((n, heads, hd, N), s, 1.5)) for s in (1, 2, 3))
    temp = 1 + _rand((heads,), 4, 0.5)
    attn = ((F.normalize(q, dim=-1) @ F.normalize(k, dim=-1).transpose(-2, -1)) * temp.view(1, heads, 1, 1)).softmax(-1)
    ref = attn @ v  # [n, heads, hd, N]

    def padded(t):  # [n, heads, hd, N] -> NCHW with 32 channels per head
        out = torch.zeros((n, heads, 32, N))
        out[:, :, :hd] = t
        return out.reshape(n, heads * 32, h, w)

    qkv = tensors.nchw_to_planes(torch.cat([padded(q), padded(k), padded(v)], dim=1).to(device))
    hp = heads * 4
    lib = L.load()
    ws = torch.empty((int(lib.rsa_channel_attn_workspace_bytes(n, h, w, heads)) // 4,), dtype=torch.float32, device=device)
    blob = int(lib.rsa_packed_weight_bytes(heads * 32, hp, 1, 3)) // 2
    wdyn = torch.zeros((n, blob), dtype=torch.bfloat16, device=device)
    td = temp.to(device)
    cp = L.ChannelAttnParams()
    cp.batch, cp.H, cp.W, cp.heads, cp.head_dim, cp.products = n, h, w, heads, hd, 3
    cp.q_hi, cp.q_lo, cp.k_hi, cp.k_lo = qkv.hi_ptr(0), qkv.lo_ptr(0), qkv.hi_ptr(hp), qkv.lo_ptr(hp)
    cp.plane_stride, cp.batch_stride = qkv.plane_stride, qkv.batch_stride
    cp.temperature, cp.workspace, cp.w_packed = td.data_ptr(), ws.data_ptr(), wdyn.data_ptr()
    L.check(lib.rsa_channel_attention_weights(C.byref(cp), _stream(device)), 'rsa_channel_attention_weights')
    out = tensors.Planes.empty(n, hp, h, w, device)
    bias = pad_bias(None, heads * 32, device)
    for b in range(n):
        wts = ops.ConvWeights(wdyn[b], bias, heads * 32, heads * 32, hp, 1, 3)
        src = tensors.Planes(qkv.hi[b : b + 1], qkv.lo[b : b + 1])
        dst = tensors.Planes(out.hi[b : b + 1], out.lo[b : b + 1])
        ops.run_convs([ops.conv_params(wts, src, h, w, in_plane0=2 * hp, cin_planes=hp, out=dst)], device)
    torch.cuda.synchronize()
    got = tensors.planes_to_nchw(out, heads * 32).cpu().reshape(n, heads, 32, N)
    err = (got[:, :, :hd] - ref).abs().max().item()
    assert err <= 1e-4 * max(1.0, ref.abs().max().item()), f'max-abs {err:.3e}'
    if hd < 32:
        assert got[:, :, hd:].abs().max().item() == 0.0


@pytest.mark.parametrize('products,tol', [(3, 1e-4), (1, 3e-2)])
@pytest.mark.parametrize('split,shifted,heads,hd,H,W', [((8, 32), False, 6, 30, 40, 72), ((8, 32), True, 6, 30, 40, 72), ((8, 16), True, 4, 16, 16, 32),
                                                        ((2, 4), True, 2, 8, 13, 18), ((4, 8), False, 2, 32, 20, 28), ((8, 12), True, 2, 10, 24, 24)])  # fmt: skip
def test_rect_attention_kernel(device, products, tol, split, shifted, heads, hd, H, W):
    """Both branches of Adaptive_Spatial_Attention (arch.py:446-492) on random q, k, v and a random position bias."""
    from oracle.dat import _from_windows, _to_windows, shift_masks

    B = 2
    C_ = heads * hd
    q, k, v = (_q16(_rand((B, H, W, C_), s, 1.2)) for s in (1, 2, 3))
    m = max(split)
    Hp, Wp = H + (m - H % m) % m, W + (m - W % m) % m
    shift = [split[0] // 2, split[1] // 2]
    qkv = F.pad(torch.stack([q, k, v], dim=3), (0, 0, 0, 0, 0, Wp - W, 0, Hp - H))  # [B, Hp, Wp, 3, C]
    masks = shift_masks(Hp, Wp, split, shift) if shifted else (None, None)
    refs, frags = [], []
    for idx in (0, 1):
        hs, ws = branch_geometry(split, idx)
        sh, sw = branch_geometry(shift, idx)
        n_tok = hs * ws
        bias = _rand((heads // 2, n_tok, n_tok), 10 + idx, 2.0)
        frags.append(bias_fragments(bias).to(device))
        part = qkv[..., idx * (C_ // 2) : (idx + 1) * (C_ // 2)]
        if shifted:
            part = torch.roll(part, shifts=(-sh, -sw), dims=(1, 2))
        qw, kw, vw = (_to_windows(part[:, :, :, i].contiguous(), hs, ws, heads // 2) for i in range(3))
        attn = qw @ kw.transpose(-2, -1) + bias.unsqueeze(0)
        if shifted:
            nw = masks[idx].shape[0]
            attn = (attn.view(B, nw, heads // 2, n_tok, n_tok) + masks[idx].view(1, nw, 1, n_tok, n_tok)).view(-1, heads // 2, n_tok, n_tok)
        o = _from_windows(attn.softmax(-1) @ vw, hs, ws, Hp, Wp)
        if shifted:
            o = torch.roll(o, shifts=(sh, sw), dims=(1, 2))
        refs.append(o[:, :H, :W])
    ref = torch.cat(refs, dim=3).permute(0, 3, 1, 2)  # [B, C, H, W]

    def padded(t):  # [B, H, W, C] -> [B, heads*32, H, W]
        out = torch.zeros((B, heads, 32, H, W))
        out[:, :, :hd] = t.permute(0, 3, 1, 2).reshape(B, heads, hd, H, W)
        return out.reshape(B, heads * 32, H, W)

    pl = tensors.nchw_to_planes(torch.cat([padded(q), padded(k), padded(v)], dim=1).to(device), with_lo=products == 3)
    out = tensors.Planes.empty(B, heads * 4, H, W, device, with_lo=products == 3)
    out.hi.zero_()
    if out.lo is not None:
        out.lo.zero_()
    lib = L.load()
    for idx in (0, 1):
        ap = L.RectAttnParams()
        ap.batch, ap.H, ap.W, ap.Hp, ap.Wp = B, H, W, Hp, Wp
        ap.win_h, ap.win_w = branch_geometry(split, idx)
        ap.shift_h, ap.shift_w = branch_geometry(shift, idx) if shifted else (0, 0)
        ap.heads, ap.head0, ap.heads_total, ap.products = heads // 2, idx * (heads // 2), heads, products
        ap.qkv_hi, ap.qkv_lo, ap.qkv_plane_stride, ap.qkv_batch_stride = pl.hi_ptr(), pl.lo_ptr(), pl.plane_stride, pl.batch_stride
        ap.bias_frag = frags[idx].data_ptr()
        ap.out_hi, ap.out_lo, ap.out_plane_stride, ap.out_batch_stride = out.hi_ptr(), out.lo_ptr(), out.plane_stride, out.batch_stride
        L.check(lib.rsa_rect_attention(C.byref(ap), _stream(device)), 'rsa_rect_attention')
    torch.cuda.synchronize()
    got = tensors.planes_to_nchw(out, heads * 32).cpu().reshape(B, heads, 32, H, W)
    err = (got[:, :, :hd].reshape(B, C_, H, W) - ref).abs().max().item()
    assert err <= tol * max(1.0, ref.abs().max().item()), f'max-abs {err:.3e}'
    if hd < 32:
        assert got[:, :, hd:].abs().max().item() == 0.0


@pytest.mark.parametrize('name', golden_names('dat_'))
def test_dat_matches_reference_vectors(device, name):
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


def test_dat_vs_oracle_dtypes_and_precision(device):
    """DAT-S-like wiring (embed 180, 6 heads, split 8x16, 2 groups x 4 blocks incl. shifted ones), fp16 tensor I/O, plain bf16 mode."""
    sd = synth.dat_state_dict(embed_dim=180, depth=(4, 4), num_heads=(6, 6), split_size=(8, 16), upscale=4, img_size=32, seed=7)
    x = synth.synth_input((1, 3, 45, 60), seed=7)
    with torch.no_grad():
        ref = oracle_forward(dict(arch='dat'), sd, x)
    m = resselt_amd.load_from_state_dict(dict(sd)).to(device)
    y = m(x.to(device))
    assert y.shape == ref.shape == (1, 3, 180, 240)
    err = (y.cpu() - ref).abs().max().item()
    print(f'DAT(2x4) bf16x3 max-abs {err:.3e} (|y|max {ref.abs().max():.2f})')
    assert err <= 3e-4 * max(1.0, ref.abs().max().item())
    yh = m(x.to(device).half())
    assert yh.dtype == torch.float16
    with torch.no_grad():
        refh = oracle_forward(dict(arch='dat'), sd, x.half().float())
    assert (yh.float().cpu() - refh).abs().max().item() <= 2e-3 * max(1.0, refh.abs().max().item())
    m.precision = 'bf16'
    e1 = (m(x.to(device)).cpu() - ref).abs().max().item()
    print(f'DAT(2x4) plain bf16 max-abs {e1:.3e}')
    assert e1 <= 5e-2 * max(1.0, ref.abs().max().item())


def _h16(x):
    return x.half().float()


def test_stream_kernels_on_fp16_planes(device):
    """Round 4: the depthwise convolution (with the spatial gate's LayerNorm statistics and multiplier), the AIM, the channel gate and the
    channel attention weights on fp16 hi-only planes (`fmt = RSA_PF_F16` in their descriptors): what DAT's 'mixed' policy runs.
    Inputs are fp16 values, so the kernels see them exactly; outputs are rounded to fp16 (2^-11 relative)."""
    from resselt_amd.engine.tensors import PF_F16

    lib = L.load()
    n, c, h, w = 2, 48, 19, 27
    planes = c // 8
    tol = lambda ref: (2.0**-11 * 1.01 + 3e-5) * max(1.0, ref.abs().max().item())  # noqa: E731
    # ---- depthwise 3x3 over LayerNorm(x) * multiplier ----
    x, m = _h16(_rand((n, c, h, w), 1, 2.0)), _h16(_rand((n, c, h, w), 6, 1.5))
    wt, b = _rand((c, 1, 3, 3), 2, 0.4), _rand((c,), 3, 0.2)
    g, beta = 1 + _rand((c,), 4, 0.3), _rand((c,), 5, 0.2)
    ref = F.conv2d(F.layer_norm(x.permute(0, 2, 3, 1), (c,), g, beta, 1e-5).permute(0, 3, 1, 2), wt, b, padding=1, groups=c) * m
    xp, mp = tensors.nchw_to_planes(x.to(device), False, PF_F16), tensors.nchw_to_planes(m.to(device), False, PF_F16)
    out = tensors.Planes.empty(n, planes, h, w, device, False, PF_F16)
    keep = [t.to(device).contiguous() for t in (wt.reshape(c, 9), b, g, beta)]
    stats = torch.empty((n, h * w, 2), dtype=torch.float32, device=device)
    L.check(lib.rsa_plane_stats_fmt(xp.hi_ptr(), None, xp.plane_stride, xp.batch_stride, n, h, w, c, 1e-5, PF_F16, stats.data_ptr(), _stream(device)), 'rsa_plane_stats_fmt')
    dp = L.DwConvParams()
    dp.batch, dp.H, dp.W, dp.planes, dp.act, dp.fmt = n, h, w, planes, L.ACT_NONE, PF_F16
    dp.in_hi, dp.in_plane_stride, dp.in_batch_stride = xp.hi_ptr(), xp.plane_stride, xp.batch_stride
    dp.weight, dp.bias, dp.stats, dp.gamma, dp.beta = keep[0].data_ptr(), keep[1].data_ptr(), stats.data_ptr(), keep[2].data_ptr(), keep[3].data_ptr()
    dp.mul_hi, dp.mul_plane_stride, dp.mul_batch_stride = mp.hi_ptr(), mp.plane_stride, mp.batch_stride
    dp.out_hi, dp.out_plane_stride, dp.out_batch_stride = out.hi_ptr(), out.plane_stride, out.batch_stride
    L.check(lib.rsa_dwconv3x3(C.byref(dp), _stream(device)), 'rsa_dwconv3x3')
    torch.cuda.synchronize()
    assert (tensors.planes_to_nchw(out, c).cpu() - ref).abs().max().item() <= tol(ref)
    # ---- AIM combine (mode 0) with the channel gate of the same maps ----
    att, conv = _h16(_rand((n, c, h, w), 11, 2.0)), _h16(_rand((n, c, h, w), 12, 2.0))
    hidden = 6
    w1, b1, w2, b2 = _rand((hidden, c), 14, 0.2), _rand((hidden,), 15, 0.2), _rand((hidden,), 16, 0.5), 0.1
    gw1, gb1, gw2, gb2 = _rand((8, c), 21, 0.2), _rand((8,), 22, 0.2), _rand((c, 8), 23, 0.4), _rand((c,), 24, 0.2)
    gate_ref = torch.sigmoid(F.linear(F.gelu(F.linear(conv.mean(dim=(2, 3)), gw1, gb1)), gw2, gb2))
    ap_, cp_ = tensors.nchw_to_planes(att.to(device), False, PF_F16), tensors.nchw_to_planes(conv.to(device), False, PF_F16)
    ws = torch.empty((int(lib.rsa_channel_gate_workspace_bytes(n, h, w, planes)) // 4,), dtype=torch.float32, device=device)
    gate = torch.empty((n, c), dtype=torch.float32, device=device)
    gkeep = [t.to(device).contiguous() for t in (gw1, gb1, gw2, gb2)]
    gp = L.ChannelGateParams()
    gp.batch, gp.H, gp.W, gp.planes, gp.hidden, gp.fmt = n, h, w, planes, 8, PF_F16
    gp.in_hi, gp.in_plane_stride, gp.in_batch_stride = cp_.hi_ptr(), cp_.plane_stride, cp_.batch_stride
    gp.w1, gp.b1, gp.w2, gp.b2 = (t.data_ptr() for t in gkeep)
    gp.workspace, gp.gate = ws.data_ptr(), gate.data_ptr()
    L.check(lib.rsa_channel_gate(C.byref(gp), _stream(device)), 'rsa_channel_gate')
    torch.cuda.synchronize()
    assert (gate.cpu() - gate_ref).abs().max().item() <= 2e-6
    s = F.conv2d(F.gelu(F.conv2d(att, w1[:, :, None, None], b1)), w2[None, :, None, None], torch.tensor([b2]))
    ref = att * gate_ref[:, :, None, None] + torch.sigmoid(s) * conv
    out2 = tensors.Planes.empty(n, planes, h, w, device, False, PF_F16)
    akeep = [t.to(device).contiguous() for t in (w1, b1, w2)]
    ap = L.AimParams()
    ap.batch, ap.H, ap.W, ap.planes, ap.hidden, ap.mode, ap.fmt = n, h, w, planes, hidden, 0, PF_F16
    ap.att_hi, ap.att_plane_stride, ap.att_batch_stride = ap_.hi_ptr(), ap_.plane_stride, ap_.batch_stride
    ap.conv_hi, ap.conv_plane_stride, ap.conv_batch_stride = cp_.hi_ptr(), cp_.plane_stride, cp_.batch_stride
    ap.gate, ap.w1, ap.b1, ap.w2, ap.b2 = gate.data_ptr(), akeep[0].data_ptr(), akeep[1].data_ptr(), akeep[2].data_ptr(), b2
    ap.out_hi, ap.out_plane_stride, ap.out_batch_stride = out2.hi_ptr(), out2.plane_stride, out2.batch_stride
    L.check(lib.rsa_aim_combine(C.byref(ap), _stream(device)), 'rsa_aim_combine')
    torch.cuda.synchronize()
    assert (tensors.planes_to_nchw(out2, c).cpu() - ref).abs().max().item() <= tol(ref)
    # ---- channel attention: Gram matrix of fp16 q / k, fp16 weight blob, attn @ v as a one-product fp16 convolution ----
    heads, hd = 3, 14
    N = h * w
    q, k, v = (_h16(_rand((n, heads, hd, N), sd, 1.5)) for sd in (31, 32, 33))
    temp = 1 + _rand((heads,), 34, 0.5)
    attn = ((F.normalize(q, dim=-1) @ F.normalize(k, dim=-1).transpose(-2, -1)) * temp.view(1, heads, 1, 1)).softmax(-1)
    ref = _h16(attn) @ v  # the packed weights are fp16 values

    def padded(t):
        o = torch.zeros((n, heads, 32, N))
        o[:, :, :hd] = t
        return o.reshape(n, heads * 32, h, w)

    qkv = tensors.nchw_to_planes(torch.cat([padded(q), padded(k), padded(v)], dim=1).to(device), False, PF_F16)
    hp = heads * 4
    ws2 = torch.empty((int(lib.rsa_channel_attn_workspace_bytes(n, h, w, heads)) // 4,), dtype=torch.float32, device=device)
    blob = int(lib.rsa_packed_weight_bytes(heads * 32, hp, 1, 1)) // 2
    wdyn = torch.zeros((n, blob), dtype=torch.bfloat16, device=device)
    td = temp.to(device)
    cpar = L.ChannelAttnParams()
    cpar.batch, cpar.H, cpar.W, cpar.heads, cpar.head_dim, cpar.products, cpar.fmt = n, h, w, heads, hd, 1, PF_F16
    cpar.q_hi, cpar.k_hi = qkv.hi_ptr(0), qkv.hi_ptr(hp)
    cpar.plane_stride, cpar.batch_stride = qkv.plane_stride, qkv.batch_stride
    cpar.temperature, cpar.workspace, cpar.w_packed = td.data_ptr(), ws2.data_ptr(), wdyn.data_ptr()
    L.check(lib.rsa_channel_attention_weights(C.byref(cpar), _stream(device)), 'rsa_channel_attention_weights')
    out3 = tensors.Planes.empty(n, hp, h, w, device, False, PF_F16)
    bias = pad_bias(None, heads * 32, device)
    for bi in range(n):
        wts = ops.ConvWeights(wdyn[bi], bias, heads * 32, heads * 32, hp, 1, 1, fmt=PF_F16)
        src = tensors.Planes(qkv.hi[bi : bi + 1], None)
        dst = tensors.Planes(out3.hi[bi : bi + 1], None)
        ops.run_convs([ops.conv_params(wts, src, h, w, in_plane0=2 * hp, cin_planes=hp, out=dst)], device)
    torch.cuda.synchronize()
    got = tensors.planes_to_nchw(out3, heads * 32).cpu().reshape(n, heads, 32, N)
    assert (got[:, :, :hd] - ref).abs().max().item() <= tol(ref)


@pytest.mark.parametrize('c,offset,spread', [(180, 0.0, 1.0), (180, 300.0, 0.05), (52, -40.0, 2.0), (8, 5.0, 0.01)])
def test_plane_stats_single_pass_is_robust(device, c, offset, spread):
    """Round 4: rsa_plane_stats makes ONE pass over the planes (per-plane (mean, M2), combined pairwise).  Per-pixel mean and 1/sqrt(var + eps)
    against f64 on data far from zero (a shifted sum of squares would cancel there) and on a channel count that ends inside a plane."""
    n, h, w = 2, 11, 23
    x = _q16(_rand((n, c, h, w), 7, spread) + offset)  # hi + lo bf16 planes hold these values exactly
    cpad = (c + 7) // 8 * 8
    xpad = torch.zeros((n, cpad, h, w))
    xpad[:, :c] = x
    xpad[:, c:] = 777.0  # padding channels must not enter the statistics
    xp = tensors.nchw_to_planes(xpad.to(device))
    stats = torch.empty((n, h * w, 2), dtype=torch.float32, device=device)
    lib = L.load()
    L.check(lib.rsa_plane_stats(xp.hi_ptr(), xp.lo_ptr(), xp.plane_stride, xp.batch_stride, n, h, w, c, 1e-5, stats.data_ptr(), _stream(device)), 'rsa_plane_stats')
    torch.cuda.synchronize()
    xd = x.double().permute(0, 2, 3, 1).reshape(n, h * w, c)
    mean = xd.mean(-1)
    rstd = 1.0 / torch.sqrt(xd.var(-1, unbiased=False) + 1e-5)
    got = stats.cpu().double()
    assert (got[..., 0] - mean).abs().max().item() <= 2e-6 * max(1.0, abs(offset))
    assert ((got[..., 1] - rstd).abs() / rstd).max().item() <= 2e-5
