"""Kernel-level GPU parity of the streaming (non-convolution) kernels that the end-to-end vectors only cover indirectly:
rsa_dysample (both modes) and the four RTMoSR kernels, each through the C-ABI against the torch ops of the reference lines they replace.
"""

import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from resselt_amd.engine import lib as L
from resselt_amd.engine import ops, tensors

pytestmark = pytest.mark.gpu


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * 2 - 1) * scale


def _stream(device):
    return C.c_void_p(ops.current_stream_ptr(device))


def _q(x):
    """The value a split-plane map holds for x (hi + lo, ~16 bits)."""
    return tensors.planes_to_nchw(tensors.nchw_to_planes(x), x.shape[1])


# ------------------------------------------------------------------------------------------------------------- DySample
def _dysample_ref(x, offset_raw, scope_raw, init_pos, end_w, end_b, scale, groups=4):
    """resselt/utilities/dysample.py:47-83 from the outputs of its two 1x1 convs on (torch ops, fp32)."""
    offset = offset_raw * scope_raw.sigmoid() * 0.5 + init_pos.view(1, -1, 1, 1)
    B, _, H, W = offset.shape
    offset = offset.view(B, 2, -1, H, W)
    cw = torch.arange(W, dtype=x.dtype) + 0.5
    ch = torch.arange(H, dtype=x.dtype) + 0.5
    coords = torch.stack(torch.meshgrid([cw, ch], indexing='ij')).transpose(1, 2).unsqueeze(1).unsqueeze(0)
    normalizer = torch.tensor([W, H], dtype=x.dtype).view(1, 2, 1, 1, 1)
    coords = 2 * (coords + offset) / normalizer - 1
    coords = F.pixel_shuffle(coords.reshape(B, -1, H, W), scale).view(B, 2, -1, scale * H, scale * W).permute(0, 2, 3, 4, 1).contiguous().flatten(0, 1)
    out = F.grid_sample(x.reshape(B * groups, -1, H, W), coords, mode='bilinear', align_corners=False, padding_mode='border')
    out = out.view(B, -1, scale * H, scale * W)
    return F.conv2d(out, end_w[:, :, None, None], end_b)


@pytest.mark.parametrize('scale,n,h,w,out_ch,preproj', [(2, 2, 13, 21, 3, False), (4, 1, 9, 17, 3, True), (3, 1, 8, 8, 4, True), (4, 1, 6, 11, 8, False)])
def test_dysample_kernel(device, scale, n, h, w, out_ch, preproj):
    from resselt_amd.archs.spanplus.arch import dysample_init_pos

    groups, fc = 4, 48
    oc = 2 * groups * scale * scale
    x = _rand((n, fc, h, w), 1)
    off_raw = _rand((n, oc, h, w), 2, 1.5)  # large offsets: samples cross pixel boundaries and hit the border clamp
    scope_raw = _rand((n, oc, h, w), 3, 3.0)
    init_pos = dysample_init_pos(scale).reshape(-1)
    end_w = _rand((out_ch, fc), 4, 0.3)
    end_b = _rand((out_ch,), 5, 0.1)
    ref = _dysample_ref(x, off_raw, scope_raw, init_pos, end_w, end_b, scale, groups)

    p = L.DySampleParams()
    p.batch, p.H, p.W, p.groups, p.scale, p.out_ch = n, h, w, groups, scale, out_ch
    keep = []
    if preproj:
        if out_ch > 4:
            pytest.skip('pre-projected mode holds at most 4 output channels')
        cpg = fc // groups
        z = torch.zeros((n, 4 * groups, h, w))
        for g in range(groups):  # z[4g + o] = sum over the channels of group g of W_end[o][c] x[c]  (sampling is linear)
            z[:, 4 * g : 4 * g + out_ch] = torch.einsum('oc,nchw->nohw', end_w[:, g * cpg : (g + 1) * cpg], x[:, g * cpg : (g + 1) * cpg])
        xm = tensors.nchw_to_f32map(z.to(device))
        p.C, p.end_w = 4 * groups, None
    else:
        xm = tensors.nchw_to_f32map(x.to(device))
        ew = end_w.contiguous().to(device)
        keep.append(ew)
        p.C, p.end_w = fc, ew.data_ptr()
    osc = tensors.nchw_to_f32map(torch.cat([off_raw, scope_raw], 1).to(device))
    ip, eb = init_pos.contiguous().to(device), end_b.contiguous().to(device)
    out = torch.empty((n, out_ch, h * scale, w * scale), dtype=torch.float32, device=device)
    p.x_f32, p.offscope, p.init_pos, p.end_b = xm.data_ptr(), osc.data_ptr(), ip.data_ptr(), eb.data_ptr()
    p.out_nchw, p.out_dtype = out.data_ptr(), L.F32
    L.check(L.load().rsa_dysample(C.byref(p), _stream(device)), 'rsa_dysample')
    torch.cuda.synchronize()
    err = (out.cpu() - ref).abs().max().item()
    assert err <= 2e-5 * max(1.0, ref.abs().max().item()), f'max-abs {err:.3e}'


# ------------------------------------------------------------------------------------------------------------- RTMoSR kernels
def test_rmsnorm_kernel(device):
    """resselt/archs/rtmosr/arch.py:32-37."""
    n, c, h, w = 2, 44, 11, 19
    x = _rand((n, c, h, w), 11, 2.0)
    scale, offset = _rand((c,), 12) + 1.0, _rand((c,), 13, 0.2)
    rms = x.norm(2, dim=1, keepdim=True) * c**-0.5
    ref = scale[None, :, None, None] * (x / (rms + 1e-6)) + offset[None, :, None, None]
    xm = tensors.nchw_to_f32map(x.to(device))
    out = tensors.Planes.empty(n, (c + 7) // 8, h, w, device)
    sc, of = scale.to(device), offset.to(device)
    L.check(L.load().rsa_rmsnorm(xm.data_ptr(), n, h, w, c, 1e-6, sc.data_ptr(), of.data_ptr(), out.hi_ptr(), out.lo_ptr(), out.plane_stride,
                                 out.batch_stride, _stream(device)), 'rsa_rmsnorm')  # fmt: skip
    torch.cuda.synchronize()
    got = tensors.planes_to_nchw(out, out.planes * 8).cpu()
    assert (got[:, :c] - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()
    assert got[:, c:].abs().max().item() == 0.0  # pad channels of the last plane


def test_unshuffle_pool_kernel(device):
    """ParPixelUnshuffle's two reads (arch.py:292-299): PixelUnshuffle(2) as an f32 map, MaxPool2d(2) as planes."""
    n, planes, h, w = 2, 3, 10, 14
    c = planes * 8
    x = _q(_rand((n, c, h, w), 21))
    src = tensors.nchw_to_planes(x.to(device))
    pu = torch.empty((n, c, h // 2, w // 2, 4), dtype=torch.float32, device=device)
    pool = tensors.Planes.empty(n, planes, h // 2, w // 2, device)
    L.check(L.load().rsa_unshuffle_pool(src.hi_ptr(), src.lo_ptr(), src.plane_stride, src.batch_stride, n, h, w, planes, pu.data_ptr(), pool.hi_ptr(),
                                        pool.lo_ptr(), pool.plane_stride, pool.batch_stride, _stream(device)), 'rsa_unshuffle_pool')  # fmt: skip
    torch.cuda.synchronize()
    # f32 group c of the map = channels 4c..4c+3 of pixel_unshuffle(x, 2)
    want_pu = F.pixel_unshuffle(x, 2).reshape(n, c, 4, h // 2, w // 2).permute(0, 1, 3, 4, 2)
    assert torch.equal(pu.cpu(), want_pu)
    assert torch.equal(tensors.planes_to_nchw(pool, c).cpu(), _q(F.max_pool2d(x, 2)))


@pytest.mark.parametrize('with_gate', [False, True])
def test_gated_shuffle_mul_kernel(device, with_gate):
    """GatedCNNBlock.forward (arch.py:334-336): mish(g) * cat(i, PixelShuffle(2)(c * gate))."""
    n, h, w = 2, 8, 12
    g_planes, i_planes = 4, 1
    cg, ci = g_planes * 8, i_planes * 8
    cc = (g_planes - i_planes) * 8 * 4  # channels of c at half resolution
    g = _q(_rand((n, cg, h, w), 31, 2.0))
    i = _q(_rand((n, ci, h, w), 32))
    cmap = _q(_rand((n, cc, h // 2, w // 2), 33))
    gate = _rand((n, cc), 34) * 0.5 + 0.5
    f = tensors.nchw_to_planes(torch.cat([g, i], 1).to(device))
    cp = tensors.nchw_to_planes(cmap.to(device))
    out = tensors.Planes.empty(n, g_planes, h, w, device)
    gd = gate.contiguous().to(device)
    sp = L.GatedShuffleParams()
    sp.batch, sp.H, sp.W, sp.g_planes, sp.i_planes = n, h, w, g_planes, i_planes
    sp.f_hi, sp.f_lo, sp.f_plane_stride, sp.f_batch_stride = f.hi_ptr(), f.lo_ptr(), f.plane_stride, f.batch_stride
    sp.c_hi, sp.c_lo, sp.c_plane_stride, sp.c_batch_stride = cp.hi_ptr(), cp.lo_ptr(), cp.plane_stride, cp.batch_stride
    sp.gate, sp.gate_stride = (gd.data_ptr() if with_gate else None), cc
    sp.out_hi, sp.out_lo, sp.out_plane_stride, sp.out_batch_stride = out.hi_ptr(), out.lo_ptr(), out.plane_stride, out.batch_stride
    L.check(L.load().rsa_gated_shuffle_mul(C.byref(sp), _stream(device)), 'rsa_gated_shuffle_mul')
    torch.cuda.synchronize()
    cs = cmap * gate[:, :, None, None] if with_gate else cmap
    ref = F.mish(g) * torch.cat([i, F.pixel_shuffle(cs, 2)], 1)
    got = tensors.planes_to_nchw(out, cg).cpu()
    assert (got - ref).abs().max().item() <= 3e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize('with_mul', [False, True])
def test_dwconv5x5_kernel(device, with_mul):
    """OmniShift re-parameterised to one depthwise 5x5 (arch.py:253-289), zero padding 2."""
    n, planes, h, w = 2, 2, 9, 13
    c = planes * 8
    x = _q(_rand((n, c, h, w), 41))
    wt, b = _rand((c, 25), 42, 0.3), _rand((c,), 43, 0.1)
    m = _q(_rand((n, c, h, w), 44))
    ref = F.conv2d(x, wt.reshape(c, 1, 5, 5), b, padding=2, groups=c)
    if with_mul:
        ref = ref * m
    src, mp = tensors.nchw_to_planes(x.to(device)), tensors.nchw_to_planes(m.to(device))
    out = tensors.Planes.empty(n, planes, h, w, device)
    wd, bd = wt.contiguous().to(device), b.to(device)
    dp = L.DwConvParams()
    dp.batch, dp.H, dp.W, dp.planes, dp.act = n, h, w, planes, L.ACT_NONE
    dp.in_hi, dp.in_lo, dp.in_plane_stride, dp.in_batch_stride = src.hi_ptr(), src.lo_ptr(), src.plane_stride, src.batch_stride
    dp.weight, dp.bias = wd.data_ptr(), bd.data_ptr()
    if with_mul:
        dp.mul_hi, dp.mul_lo, dp.mul_plane_stride, dp.mul_batch_stride = mp.hi_ptr(), mp.lo_ptr(), mp.plane_stride, mp.batch_stride
    dp.out_hi, dp.out_lo, dp.out_plane_stride, dp.out_batch_stride = out.hi_ptr(), out.lo_ptr(), out.plane_stride, out.batch_stride
    L.check(L.load().rsa_dwconv5x5(C.byref(dp), _stream(device)), 'rsa_dwconv5x5')
    torch.cuda.synchronize()
    got = tensors.planes_to_nchw(out, c).cpu()
    assert (got - ref).abs().max().item() <= 3e-5 * max(1.0, ref.abs().max().item())
