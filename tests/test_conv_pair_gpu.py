"""GPU parity of the cross-layer fusion of a residual dense block (round 4): `rsa_conv2d_pair` / the pair fusion inside `rsa_conv2d_list`
(csrc/conv_ring_pair.h) against the two separate launches it replaces and against the f32 convolution of the fp16-rounded operands.

Reference: ResidualDenseBlock_5C.forward, resselt/utilities/block.py:454-465 -- conv1 -> conv2 and conv3 -> conv4 over the growing
concatenation.  Every accumulator of the fused kernel sees its layer's K steps in the order the layer-wise kernel runs them, so the bar
is BIT equality with the unfused launches (and, through them, the tolerances of tests/test_conv_fp16_gpu.py against the CPU convolution).
"""

import pytest
import torch
import torch.nn.functional as F

from resselt_amd.engine import lib as L
from resselt_amd.engine import ops, tensors
from resselt_amd.engine.tensors import PF_F16

pytestmark = pytest.mark.gpu


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * 2 - 1) * scale


def _h(t):
    return t.half().float()


def _conv(x, w, b):
    return F.conv2d(x.double(), w.double(), b.double(), padding=1).float()


def _workspace(x, planes, device, fill):
    """24-plane workspace of an RDB under the 'mixed' policy: x in its first planes, `fill` (a sentinel) everywhere else."""
    n, c, h, w = x.shape
    ws = tensors.Planes.empty(n, planes, h, w, device, True, PF_F16, lo_planes=8)
    ws.hi.fill_(fill)
    ws.lo.fill_(fill)
    src = tensors.nchw_to_planes(x.to(device), False, PF_F16)
    ws.hi[:, : c // 8] = src.hi
    return ws


def _pair(ws, wa, wb, h, w, cin, slope_a=0.2, slope_b=0.2, order=0):
    pa, pb = cin // 8, cin // 8 + 4
    act = lambda s: dict(act=L.ACT_LRELU, act_param=s) if s is not None else dict(act=L.ACT_NONE)  # noqa: E731
    a = ops.conv_params(wa, ws, h, w, cin_planes=pa, out=ws, out_plane_off=pa, **act(slope_a))
    b = ops.conv_params(wb, ws, h, w, cin_planes=pb, out=ws, out_plane_off=pb, **act(slope_b))
    a.tile_order = b.tile_order = order
    return a, b


@pytest.mark.parametrize(
    'n,cin,h,w,order',
    [
        (1, 64, 16, 30, 0),  # exactly one tile
        (1, 64, 37, 70, 0),  # ragged edges on both axes: 3 x 3 tiles
        (2, 64, 20, 45, 1),  # batch, reversed tile order
        (1, 128, 33, 61, 0),  # conv3 -> conv4: four chunks, the weight ring wraps inside a tile
        (1, 128, 5, 3, 1),  # smaller than a tile in both directions
        (1, 64, 1, 1, 0),
        (1, 64, 96, 200, 1),  # 6 x 7 tiles: interior tiles on the scalar-base loader path
        (1, 128, 130, 260, 0),  # more tiles than one round of the grid would need on a small part; several tiles per workgroup on 256 CUs: 9 x 9 = 81
    ],
)
def test_pair_is_bit_identical_to_the_two_launches(device, n, cin, h, w, order):
    x = _rand((n, cin, h, w), 11)
    wa_t = _rand((32, cin, 3, 3), 12, 1.0 / (cin * 9) ** 0.5)
    wb_t = _rand((32, cin + 32, 3, 3), 13, 1.0 / ((cin + 32) * 9) ** 0.5)
    ba, bb = _rand((32,), 14, 0.1), _rand((32,), 15, 0.1)
    wa = ops.ConvWeights.from_oihw(wa_t, ba, 1, device=device, fmt=PF_F16)
    wb = ops.ConvWeights.from_oihw(wb_t, bb, 1, device=device, fmt=PF_F16)
    planes = 24
    ws1 = _workspace(x, planes, device, 7.0)
    ws2 = _workspace(x, planes, device, 7.0)
    a1, b1 = _pair(ws1, wa, wb, h, w, cin, order=order)
    a2, b2 = _pair(ws2, wa, wb, h, w, cin, order=order)
    assert 'one fp16 product' in L.conv_kernel_name(a1)
    stream = ops.current_stream_ptr(device)
    try:
        L.set_pair_fusion(0)
        assert not L.conv_pair_fusable(a1, b1)
        L.conv2d_list([a1, b1], stream)
        L.set_pair_fusion(1)
        assert L.conv_pair_fusable(a2, b2)
        L.conv2d_pair(a2, b2, stream)
    finally:
        L.set_pair_fusion(-1)
    torch.cuda.synchronize()
    assert L.ring_aborts() == 0
    L.check_status('test')
    pa = cin // 8
    assert torch.equal(ws1.hi[:, pa : pa + 4], ws2.hi[:, pa : pa + 4]), 'layer A differs from the separate launch'
    assert torch.equal(ws1.hi[:, pa + 4 : pa + 8], ws2.hi[:, pa + 4 : pa + 8]), 'layer B differs from the separate launch'
    # nothing else was touched: the input planes, the planes behind B's output, the lo halves
    assert torch.equal(ws1.hi[:, :pa], ws2.hi[:, :pa])
    assert (ws2.hi[:, pa + 8 :] == 7.0).all() and (ws2.lo == 7.0).all()
    # and against the CPU: f32 convolutions of the fp16-rounded operands (accumulation order only)
    xa = _h(x)
    ya = F.leaky_relu(_conv(xa, _h(wa_t), ba), 0.2)
    got_a = tensors.planes_to_nchw(tensors.Planes(ws2.hi[:, pa : pa + 4].contiguous(), None), 32).cpu()
    sa = ya.abs().max().item()
    assert (got_a - ya).abs().max().item() <= 2.0**-11 * sa * 1.01 + 1e-5 * sa
    yb = F.leaky_relu(_conv(torch.cat((xa, got_a), 1), _h(wb_t), bb), 0.2)  # B reads A's ROUNDED output
    got_b = tensors.planes_to_nchw(tensors.Planes(ws2.hi[:, pa + 4 : pa + 8].contiguous(), None), 32).cpu()
    sb = yb.abs().max().item()
    assert (got_b - yb).abs().max().item() <= 2.0**-11 * sb * 1.01 + 1e-5 * sb


def test_pair_without_activation_and_mixed_slopes(device):
    n, cin, h, w = 1, 64, 21, 40
    x = _rand((n, cin, h, w), 21)
    wa_t = _rand((32, cin, 3, 3), 22, 1.0 / (cin * 9) ** 0.5)
    wb_t = _rand((32, cin + 32, 3, 3), 23, 1.0 / ((cin + 32) * 9) ** 0.5)
    ba, bb = _rand((32,), 24, 0.5), _rand((32,), 25, 0.5)
    wa = ops.ConvWeights.from_oihw(wa_t, ba, 1, device=device, fmt=PF_F16)
    wb = ops.ConvWeights.from_oihw(wb_t, bb, 1, device=device, fmt=PF_F16)
    ws1, ws2 = _workspace(x, 24, device, 0.0), _workspace(x, 24, device, 0.0)
    a1, b1 = _pair(ws1, wa, wb, h, w, cin, slope_a=None, slope_b=0.05)
    a2, b2 = _pair(ws2, wa, wb, h, w, cin, slope_a=None, slope_b=0.05)
    stream = ops.current_stream_ptr(device)
    try:
        L.set_pair_fusion(0)
        L.conv2d_list([a1, b1], stream)
        L.set_pair_fusion(1)
        L.conv2d_list([a2, b2], stream)  # fused by the list itself
    finally:
        L.set_pair_fusion(-1)
    torch.cuda.synchronize()
    assert L.ring_aborts() == 0
    assert torch.equal(ws1.hi, ws2.hi)
    ya = _conv(_h(x), _h(wa_t), ba)  # no activation on A: negative values reach B
    got_a = tensors.planes_to_nchw(tensors.Planes(ws2.hi[:, 8:12].contiguous(), None), 32).cpu()
    assert (got_a - ya).abs().max().item() <= 2.0**-11 * ya.abs().max().item() * 1.01 + 1e-5
    assert got_a.min().item() < -0.1


def test_pair_eligibility_is_strict(device):
    """What is NOT a pair runs layer by layer (and rsa_conv2d_pair refuses it): a different input buffer, a gap between A's output and B's
    input planes, 64 output channels, hi + lo outputs."""
    n, cin, h, w = 1, 64, 18, 34
    x = _rand((n, cin, h, w), 31)
    mk = lambda co, ci, s: ops.ConvWeights.from_oihw(_rand((co, ci, 3, 3), s, 0.05), _rand((co,), s + 1, 0.1), 1, device=device, fmt=PF_F16)  # noqa: E731
    wa, wb, w64 = mk(32, 64, 32), mk(32, 96, 34), mk(64, 96, 36)
    ws, other = _workspace(x, 24, device, 0.0), _workspace(x, 24, device, 0.0)
    lrelu = dict(act=L.ACT_LRELU, act_param=0.2)
    a = ops.conv_params(wa, ws, h, w, cin_planes=8, out=ws, out_plane_off=8, **lrelu)
    b = ops.conv_params(wb, ws, h, w, cin_planes=12, out=ws, out_plane_off=12, **lrelu)
    stream = ops.current_stream_ptr(device)
    try:
        L.set_pair_fusion(1)
        assert L.conv_pair_fusable(a, b)
        a_gap = ops.conv_params(wa, ws, h, w, cin_planes=8, out=ws, out_plane_off=16, **lrelu)
        b_other = ops.conv_params(wb, other, h, w, cin_planes=12, out=other, out_plane_off=12, **lrelu)
        b64 = ops.conv_params(w64, ws, h, w, cin_planes=12, out=other, out_plane_off=0)  # hi + lo output planes (the first eight of a workspace)
        for pa, pb in ((a_gap, b), (a, b_other), (a, b64), (b, a)):
            assert not L.conv_pair_fusable(pa, pb)
            assert L.load().rsa_conv2d_pair(pa, pb, stream) == -2  # RSA_E_UNSUPPORTED
        assert b'fusable' in L.load().rsa_last_error_string()
    finally:
        L.set_pair_fusion(-1)
    torch.cuda.synchronize()


def test_pair_failed_handoff_is_reported(device):
    """A hand-off that runs into its spin bound makes the fused kernel drain and report through the same failure word as the ring kernels."""
    n, cin, h, w = 1, 64, 64, 120
    x = _rand((n, cin, h, w), 41)
    mk = lambda co, ci, s: ops.ConvWeights.from_oihw(_rand((co, ci, 3, 3), s, 0.05), _rand((co,), s + 1, 0.1), 1, device=device, fmt=PF_F16)  # noqa: E731
    wa, wb = mk(32, 64, 42), mk(32, 96, 44)
    ws = _workspace(x, 24, device, 0.0)
    a, b = _pair(ws, wa, wb, h, w, cin)
    stream = ops.current_stream_ptr(device)
    try:
        L.set_pair_fusion(1)
        L.set_ring_spin_limit(1)
        L.conv2d_pair(a, b, stream)
        torch.cuda.synchronize()
        with pytest.raises(RuntimeError, match='hand-off'):
            L.check_status('test')
    finally:
        L.set_ring_spin_limit(1 << 18)
        L.set_pair_fusion(-1)
    assert L.ring_aborts() > 0  # reads and clears the device-side counters
    L.check_status('test')  # the word was consumed: clean again
    L.conv2d_pair(a, b, stream)
    torch.cuda.synchronize()
    assert L.ring_aborts() == 0
    L.check_status('test')


def test_rrdbnet_fused_equals_unfused(device):
    """The whole model: pair fusion on (default) and off give the same pixels, bit for bit."""
    import resselt_amd
    from resselt_amd.utils import synth

    sd = synth.rrdbnet_state_dict(nb=2, seed=5)
    model = resselt_amd.load_from_state_dict(dict(sd)).to(device)
    x = synth.synth_input((1, 3, 50, 77), seed=5).to(device)
    try:
        L.set_pair_fusion(1)
        y1 = model(x).clone()
        torch.cuda.synchronize()
        L.set_pair_fusion(0)
        y0 = model(x).clone()
        torch.cuda.synchronize()
    finally:
        L.set_pair_fusion(-1)
    assert L.ring_aborts() == 0
    assert torch.equal(y0, y1)
