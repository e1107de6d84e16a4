"""GPU parity of the fp16 plane format (round 3): one fp16 product on hi planes (`products = 1, fmt = RSA_PF_F16`: the ring schedule with
eight LDS slots, and the chunk-barrier kernels for the shapes the ring does not take), three fp16 products for the four-tile shape, fp16
plane outputs / residuals of the epilogue, and `rsa_pack_weights(fmt)`.

The reference of a one-product layer is the f32 convolution of the fp16-ROUNDED operands: every product of two 11-bit values is exact in
f32, so the kernel may differ from it by accumulation order only (tolerance 1e-5 * scale); against the unrounded fp32 convolution the
error is the operand rounding itself (2^-11 per operand), checked at 2e-3 * scale.
"""

import pytest
import torch
import torch.nn.functional as F

from resselt_amd.engine import lib as L
from resselt_amd.engine import ops, pack, tensors
from resselt_amd.engine.tensors import PF_BF16, PF_F16

pytestmark = pytest.mark.gpu


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * 2 - 1) * scale


def _h(t):
    return t.half().float()


def _conv(x, w, b=None):
    return F.conv2d(x.double(), w.double(), None if b is None else b.double(), padding=w.shape[-1] // 2).float()


@pytest.mark.parametrize(
    'n,cin,cout,h,w,k',
    [
        (2, 64, 32, 20, 45, 3),  # ring SHAPE 2 (two streams, eight slots), ragged tile edges
        (1, 160, 32, 33, 70, 3),  # five chunks: the weight ring wraps inside a tile
        (1, 192, 64, 19, 37, 3),  # ring SHAPE 1 (conv5)
        (1, 64, 64, 40, 40, 3),
        (1, 48, 48, 24, 24, 3),  # ring SHAPE 3, half mode
        (1, 96, 48, 17, 50, 3),  # ring SHAPE 3, whole chunks
        (1, 3, 48, 16, 32, 3),  # chunk-barrier kernel, channel padding 3 -> 8
        (1, 64, 3, 17, 31, 3),  # chunk-barrier kernel, one cout tile
        (1, 192, 48, 8, 40, 1),  # k1 (SPAN conv_cat)
        (1, 240, 720, 8, 32, 1),  # k1, several cout slabs
    ],
)
def test_conv_one_fp16_product(device, n, cin, cout, h, w, k):
    x = _rand((n, cin, h, w), 1)
    wt = _rand((cout, cin, k, k), 2, 1.0 / (cin * k * k) ** 0.5)
    b = _rand((cout,), 3, 0.1)
    ref = _conv(_h(x), _h(wt), b)
    exact = _conv(x, wt, b)
    wts = ops.ConvWeights.from_oihw(wt, b, 1, device=device, fmt=PF_F16)
    xin = tensors.nchw_to_planes(x.to(device), with_lo=False, fmt=PF_F16)
    out = tensors.Planes.empty(n, (cout + 7) // 8, h, w, device, with_lo=False, fmt=PF_F16)
    out2 = tensors.Planes.empty(n, (cout + 7) // 8, h, w, device, with_lo=True, fmt=PF_F16)
    of32 = tensors.empty_f32map(n, cout, h, w, device)
    onchw = torch.empty((n, cout, h, w), dtype=torch.float32, device=device)
    ps = [ops.conv_params(wts, xin, h, w, out=out), ops.conv_params(wts, xin, h, w, out=out2, out_f32=of32), ops.conv_params(wts, xin, h, w, out_nchw=onchw)]
    assert all(p.in_fmt == PF_F16 and p.products == 1 for p in ps)
    ops.run_convs(ps, device)
    torch.cuda.synchronize()
    assert L.ring_aborts() == 0
    L.check_status('test')
    scale = ref.abs().max().item()
    for name, got in (('out_nchw', onchw.cpu()), ('out_f32', tensors.f32map_to_nchw(of32, cout).cpu())):
        assert (got - ref).abs().max().item() <= 1e-5 * scale, name
        assert (got - exact).abs().max().item() <= 2e-3 * scale, name
    # hi-only fp16 planes: the value rounded to fp16; hi + lo planes: 22 bits
    hi_only = tensors.planes_to_nchw(out, cout).cpu()
    assert (hi_only - ref).abs().max().item() <= 2.0**-11 * scale * 1.01 + 1e-5 * scale
    both = tensors.planes_to_nchw(out2, cout).cpu()
    assert (both - ref).abs().max().item() <= 1.2e-5 * scale
    full = tensors.planes_to_nchw(out, out.planes * 8)
    if out.planes * 8 > cout:
        assert full[:, cout:].abs().max().item() == 0.0


def test_rdb_under_the_mixed_policy(device):
    """One residual dense block as the 'mixed' plan runs it: growth convolutions in one fp16 product writing hi-only planes at plane
    offsets of a workspace whose first 8 planes keep hi + lo, conv5 with `x5 * 0.2 + x` and the RRDB residual from fp16 planes."""
    n, h, w, pf, pg = 1, 37, 70, 8, 4
    g = torch.Generator().manual_seed(5)
    x = _rand((n, 64, h, w), 5)
    x0 = _rand((n, 64, h, w), 6)
    ws = tensors.Planes.empty(n, pf + 4 * pg, h, w, device, True, PF_F16, lo_planes=pf)
    ws.hi.zero_()
    src = tensors.nchw_to_planes(x.to(device), True, PF_F16)
    ws.hi[:, :pf] = src.hi
    ws.lo[:, :pf] = src.lo
    r0src = tensors.nchw_to_planes(x0.to(device), True, PF_F16)
    r0 = tensors.Planes.empty(n, pf + 4 * pg, h, w, device, True, PF_F16, lo_planes=pf)  # another workspace: both residuals share their strides
    r0.hi[:, :pf] = r0src.hi
    r0.lo[:, :pf] = r0src.lo
    out = tensors.Planes.empty(n, pf, h, w, device, True, PF_F16)
    ps, keep = [], []  # (a descriptor holds raw pointers: the weight blobs must outlive the launch)
    cat = _h(tensors.planes_to_nchw(src, 64).cpu())
    for j in range(1, 6):
        cin, cout = 64 + 32 * (j - 1), 32 if j < 5 else 64
        wt = (torch.rand((cout, cin, 3, 3), generator=g) * 2 - 1) / (cin * 9) ** 0.5
        b = (torch.rand((cout,), generator=g) * 2 - 1) * 0.1
        wts = ops.ConvWeights.from_oihw(wt, b, 1, device=device, fmt=PF_F16)
        keep.append(wts)
        y = _conv(cat, _h(wt), b)
        if j < 5:
            y = F.leaky_relu(y, 0.2)
            ps.append(ops.conv_params(wts, ws, h, w, cin_planes=cin // 8, out=ws, out_plane_off=pf + (j - 1) * pg, act=L.ACT_LRELU, act_param=0.2))
            assert ps[-1].out_lo is None and ps[-1].out_fmt == PF_F16
            cat = torch.cat((cat, _h(y)), 1)
        else:
            xs = tensors.planes_to_nchw(src, 64).cpu()
            r0v = tensors.planes_to_nchw(r0src, 64).cpu()
            ref = (y * 0.2 + xs) * 0.2 + r0v
            ps.append(ops.conv_params(wts, ws, h, w, cin_planes=cin // 8, res1=(ws, 0), alpha=0.2, res2=(r0, 0), beta=0.2, out=out))
            assert ps[-1].res1_lo is not None and ps[-1].out_lo is not None and ps[-1].res_fmt == PF_F16
    assert 'one fp16 product' in L.conv_kernel_name(ps[0]) and 'one fp16 product' in L.conv_kernel_name(ps[-1])
    ops.run_convs(ps, device)
    torch.cuda.synchronize()
    assert L.ring_aborts() == 0
    got = tensors.planes_to_nchw(out, 64).cpu()
    # (a growth value that lands on the other side of an fp16 rounding boundary than the reference's moves conv5 by 2^-11 * |w| * 0.04)
    assert (got - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()
    # the growth planes hold fp16(x1..x4)
    growth = tensors.planes_to_nchw(tensors.Planes(ws.hi[:, pf:].contiguous(), None), 128).cpu()
    assert (growth - cat[:, 64:]).abs().max().item() <= 2.0**-10 * cat.abs().max().item()


def test_three_fp16_products_four_tile_ring(device):
    """The trunk convolution of the 'mixed' plan: fp16 hi + lo input (the residual stream), fp16 hi + lo weights, f32-map shortcut,
    bf16 hi + lo output for the three-product upsampling layers behind it."""
    n, h, w = 1, 35, 66
    x = _rand((n, 64, h, w), 11, 2.0)
    wt = _rand((64, 64, 3, 3), 12, 1.0 / (64 * 9) ** 0.5)
    b = _rand((64,), 13, 0.1)
    r = _rand((n, 64, h, w), 14)
    xin = tensors.nchw_to_planes(x.to(device), True, PF_F16)
    xv = tensors.planes_to_nchw(xin, 64).cpu()  # hi + lo: 22 bits
    ref = _conv(xv, wt, b) + r
    wts = ops.ConvWeights.from_oihw(wt, b, 3, device=device, fmt=PF_F16)
    out = tensors.Planes.empty(n, 8, h, w, device, True, PF_BF16)
    of32 = tensors.empty_f32map(n, 64, h, w, device)
    p = ops.conv_params(wts, xin, h, w, res1=tensors.nchw_to_f32map(r.to(device)), alpha=1.0, out=out, out_f32=of32)
    assert 'three fp16 products' in L.conv_kernel_name(p) and p.out_fmt == PF_BF16
    ops.run_convs([p], device)
    torch.cuda.synchronize()
    scale = ref.abs().max().item()
    assert (tensors.f32map_to_nchw(of32, 64).cpu() - ref).abs().max().item() <= 3e-6 * scale  # lo*lo dropped: 2^-22
    assert (tensors.planes_to_nchw(out, 64).cpu() - ref).abs().max().item() <= 2e-5 * scale  # bf16 hi + lo: 2^-16


def test_whole_map_many_tiles_per_workgroup(device):
    """More tiles than workgroups (each workgroup's streams run several tiles through the eight-slot ring), batch 2, against F.conv2d."""
    n, cin, cout, h, w = 2, 96, 32, 300, 610
    x = _rand((n, cin, h, w), 21)
    wt = _rand((cout, cin, 3, 3), 22, 1.0 / (cin * 9) ** 0.5)
    b = _rand((cout,), 23, 0.1)
    ref = F.leaky_relu(F.conv2d(_h(x), _h(wt), b, padding=1), 0.2)
    wts = ops.ConvWeights.from_oihw(wt, b, 1, device=device, fmt=PF_F16)
    xin = tensors.nchw_to_planes(x.to(device), False, PF_F16)
    of32 = tensors.empty_f32map(n, cout, h, w, device)
    out = tensors.Planes.empty(n, 4, h, w, device, False, PF_F16)
    before = L.ring_aborts()
    for _ in range(3):
        ops.run_convs([ops.conv_params(wts, xin, h, w, act=L.ACT_LRELU, act_param=0.2, out=out, out_f32=of32)], device)
    torch.cuda.synchronize()
    assert L.ring_aborts() == before
    L.check_status('test')
    assert (tensors.f32map_to_nchw(of32, cout).cpu() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()


@pytest.mark.parametrize('products', [1, 3])
@pytest.mark.parametrize('layout,cin,cout,k', [(0, 40, 24, 3), (0, 240, 100, 1), (1, 64, 32, 3), (1, 192, 64, 3), (2, 48, 48, 3)])
def test_pack_weights_fp16(device, products, layout, cin, cout, k):
    w = _rand((cout, cin, k, k), 31, 0.3)
    cin_planes = (cin + 7) // 8
    got = ops.pack_weights_device(w.to(device), cin_planes, products, layout, PF_F16).view(torch.int16).cpu()
    if layout == 0:
        exp = pack.pack_conv_weights(w, cin_planes, products, torch.float16)
    elif layout == 1:
        exp = pack.pack_conv_weights_pairs(w, cin_planes, products, torch.float16)
    else:
        exp = pack.pack_conv_weights_halfpairs(w, cin_planes, products, torch.float16)
    assert torch.equal(got, exp.reshape(-1).view(torch.int16))


def test_failed_hand_off_is_an_error(device):
    """With the spin bound forced to one poll the ring kernels drain with wrong pixels; the host-visible failure word turns that into an
    exception at rsa_check_status and makes the next rsa_conv2d_list refuse to launch."""
    n, cin, cout, h, w = 1, 64, 32, 128, 256
    x = _rand((n, cin, h, w), 41)
    wt = _rand((cout, cin, 3, 3), 42, 0.05)
    wts = ops.ConvWeights.from_oihw(wt, None, 1, device=device, fmt=PF_F16)
    xin = tensors.nchw_to_planes(x.to(device), False, PF_F16)
    out = tensors.Planes.empty(n, 4, h, w, device, False, PF_F16)
    p = ops.conv_params(wts, xin, h, w, out=out)
    torch.cuda.synchronize()
    L.check_status('before')
    assert L.ring_aborts() == 0
    L.set_ring_spin_limit(1)
    try:
        ops.run_convs([p], device)
        torch.cuda.synchronize()
    finally:
        L.set_ring_spin_limit(1 << 18)
    if L.ring_aborts() == 0:  # (reads and clears the debug counter: later tests start from zero again)
        L.check_status('nothing timed out')
        pytest.skip('every hand-off was ready at its first poll: nothing timed out on this run')
    with pytest.raises(RuntimeError, match='hand-off'):
        ops.run_convs([p], device)  # refused: a failure is pending
    with pytest.raises(RuntimeError, match='hand-off'):
        L.check_status('forced failure')
    L.check_status('cleared')  # reported once, then clear
    ops.run_convs([p], device)
    torch.cuda.synchronize()
    L.check_status('after')


@pytest.mark.parametrize(
    'n,cin,cout,h,w,prod,fmt',
    [
        (2, 64, 32, 52, 75, 1, PF_F16),  # SHAPE 2, two images, ragged edges, several tiles per stream
        (1, 192, 64, 70, 130, 1, PF_F16),  # SHAPE 1
        (1, 48, 48, 40, 66, 1, PF_F16),  # SHAPE 3, half mode
        (1, 96, 32, 35, 64, 3, PF_BF16),  # three products
    ],
)
def test_conv_ring_reversed_tile_order(device, n, cin, cout, h, w, prod, fmt):
    """``rsa_conv_params.tile_order = 1`` (every other layer of a serpentine plan walks the map bottom-up) changes which workgroup computes a
    tile and when, never the tile's arithmetic: outputs are bit-identical to order 0; any other value is an argument error."""
    x = _rand((n, cin, h, w), 11)
    wt = _rand((cout, cin, 3, 3), 12, 1.0 / (cin * 9) ** 0.5)
    b = _rand((cout,), 13, 0.1)
    wts = ops.ConvWeights.from_oihw(wt, b, prod, device=device, fmt=fmt)
    xin = tensors.nchw_to_planes(x.to(device), with_lo=prod == 3, fmt=fmt)
    outs = []
    for order in (0, 1):
        o = tensors.Planes.empty(n, (cout + 7) // 8, h, w, device, with_lo=True, fmt=fmt)
        m = tensors.empty_f32map(n, cout, h, w, device)
        p = ops.conv_params(wts, xin, h, w, out=o, out_f32=m, act=L.ACT_LRELU, act_param=0.2)
        p.tile_order = order
        assert 'conv_ring' in L.conv_kernel_name(p)
        ops.run_convs([p], device)
        torch.cuda.synchronize()
        assert L.ring_aborts() == 0
        outs.append((tensors.planes_to_nchw(o, cout).cpu(), tensors.f32map_to_nchw(m, cout).cpu()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    ref = F.leaky_relu(_conv(x, wt, b), 0.2)
    assert (outs[1][1] - ref).abs().max().item() <= (2e-3 if prod == 1 else 1e-4) * ref.abs().max().item()
    p.tile_order = 2
    with pytest.raises(RuntimeError):
        ops.run_convs([p], device)


@pytest.mark.parametrize('two', [False, True])
@pytest.mark.parametrize('n,h,w', [(1, 40, 70), (2, 300, 610)])
def test_conv5_residual_from_the_ring(device, two, n, h, w):
    """conv5 of a residual dense block under the 'mixed' policy (`x5 * 0.2 + x`, utilities/block.py:454-465 of the reference; every third one also
    the RRDB's `out * 0.2 + x0`, :340-344): the kernel walks its K loop from the last chunk to the first and reads the hi halves of the
    residual x from the ring slots that still hold x's planes (conv_ring.h XRES).  Small map, and a map with several tiles per workgroup (the
    held-back slots rotate through the ring), two images, ragged edges."""
    cin, cout = 192, 64
    x = _rand((n, cin, h, w), 31, 1.5)
    wt = _rand((cout, cin, 3, 3), 32, 1.0 / (cin * 9) ** 0.5)
    b = _rand((cout,), 33, 0.1)
    ws = tensors.Planes.empty(n, cin // 8, h, w, device, True, PF_F16, lo_planes=8)
    src = tensors.nchw_to_planes(x.to(device), True, PF_F16)
    ws.hi.copy_(src.hi)
    ws.lo.copy_(src.lo[:, :8])
    xin = _h(x)  # what the one-product multiply sees
    xres = tensors.planes_to_nchw(tensors.Planes(src.hi[:, :8].contiguous(), src.lo[:, :8].contiguous()), 64).cpu()  # hi + lo: 22 bits
    ref = _conv(xin, _h(wt), b) * 0.2 + xres
    kw = {}
    if two:
        r0 = _rand((n, 64, h, w), 34)
        r0s = tensors.nchw_to_planes(r0.to(device), True, PF_F16)
        r0p = tensors.Planes.empty(n, cin // 8, h, w, device, True, PF_F16, lo_planes=8)  # another workspace: both residuals share their strides
        r0p.hi[:, :8].copy_(r0s.hi)
        r0p.lo.copy_(r0s.lo)
        ref = ref * 0.2 + tensors.planes_to_nchw(r0s, 64).cpu()
        kw = dict(res2=(r0p, 0), beta=0.2)
    wts = ops.ConvWeights.from_oihw(wt, b, 1, device=device, fmt=PF_F16)
    out = tensors.Planes.empty(n, 8, h, w, device, True, PF_F16)
    p = ops.conv_params(wts, ws, h, w, cin_planes=cin // 8, res1=(ws, 0), alpha=0.2, out=out, **kw)
    assert 'XRES' in L.conv_kernel_name(p)
    outs = []
    for order in (0, 1):
        p.tile_order = order
        ops.run_convs([p], device)
        torch.cuda.synchronize()
        assert L.ring_aborts() == 0
        L.check_status('test')
        outs.append(tensors.planes_to_nchw(out, 64).cpu())
    assert torch.equal(outs[0], outs[1])
    assert (outs[0] - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()


@pytest.mark.parametrize('act', ['mish', 'silu', 'gate_hi', 'gate_hilo', 'none', 'lrelu'])
@pytest.mark.parametrize('out_lo', [False, True])
def test_span_family_layer_direct_epilogue(device, act, out_lo):
    """The re-parameterised 48 -> 48 layers of the SPAN family (reference archs/spanplus/arch.py:94-130) on the ring form with the weight
    blob resident in LDS and one epilogue instantiation per activation class (conv_ring.h XRES 3): Mish, SiLU, the SPAB gate
    `(out3 + x) * (sigmoid(out3) - 0.5)` with its shortcut from hi-only and from hi + lo planes, no activation, LeakyReLU."""
    n, c, h, w = 2, 48, 37, 70
    x = _rand((n, c, h, w), 51)
    sc = _rand((n, c, h, w), 52)
    wt = _rand((c, c, 3, 3), 53, 1.0 / (c * 9) ** 0.5)
    b = _rand((c,), 54, 0.1)
    y = _conv(_h(x), _h(wt), b)
    wts = ops.ConvWeights.from_oihw(wt, b, 1, device=device, fmt=PF_F16)
    xin = tensors.nchw_to_planes(x.to(device), with_lo=False, fmt=PF_F16)
    out = tensors.Planes.empty(n, c // 8, h, w, device, with_lo=out_lo, fmt=PF_F16)
    kw = {}
    if act == 'mish':
        ref, kw = F.mish(y), dict(act=L.ACT_MISH)
    elif act == 'silu':
        ref, kw = F.silu(y), dict(act=L.ACT_SILU)
    elif act == 'lrelu':
        ref, kw = F.leaky_relu(y, 0.1), dict(act=L.ACT_LRELU, act_param=0.1)
    elif act == 'none':
        ref = y
    else:
        res = tensors.nchw_to_planes(sc.to(device), with_lo=act == 'gate_hilo', fmt=PF_F16)
        scv = tensors.planes_to_nchw(res, c).cpu()  # the shortcut as the planes hold it (11 or 22 bits)
        ref, kw = (y + scv) * (torch.sigmoid(y) - 0.5), dict(act=L.ACT_SPAB_GATE, res1=(res, 0))
    p = ops.conv_params(wts, xin, h, w, out=out, **kw)
    assert 'XRES 3' in L.conv_kernel_name(p), L.conv_kernel_name(p)
    ops.run_convs([p], device)
    torch.cuda.synchronize()
    assert L.ring_aborts() == 0
    L.check_status('test')
    got = tensors.planes_to_nchw(out, c).cpu()
    scale = max(1.0, ref.abs().max().item())
    # hi only: the value rounded to fp16; hi + lo: 22 bits; the activation functions run on hardware exp / rcp (1 ulp each)
    tol = (2.0**-11 * 1.01 if not out_lo else 2e-6) * scale + 3e-6 * scale
    assert (got - ref).abs().max().item() <= tol, (got - ref).abs().max().item()






@pytest.mark.parametrize('case', ['all_lo8', 'last_block', 'one_residual', 'first_layer'])
def test_residual_stream_with_8bit_lo_halves(device, case):
    """Round 4: the lo halves of the fp16 residual stream as 8-bit codes (offsets from hi in 1/254 ulp; rsa_conv_params.lo8_flags).  conv5 of a dense block with both
    residuals and its output in that form (the direct instantiation `XRES 4`), the last block of a trunk (8-bit lo in, fp16 lo out: `XRES 5`),
    one residual only, and a first layer (3 -> 64 on the chunk-barrier kernel) that writes hi + 8-bit lo planes beside its f32 map."""
    n, h, w, pf, pg = 1, 37, 70, 8, 4
    g = torch.Generator().manual_seed(15)

    def stream(seed):  # a 64-channel map as the engine stores it (fp16 hi + 8-bit code), with zeros, fp16 subnormals and powers of two in it
        v = _rand((n, 64, h, w), seed)
        v[:, :, 0, :8] = torch.tensor([0.0, 1e-7, -3e-6, 6.1e-5, 0.25, -0.5, 0.24999, 1.0001])
        v[:, :, 1, :6] = torch.tensor([1e-9, -1e-9, 2.9e-8, -2.5e-8, 1.0 + 2.0**-11, -3.0 - 2.0**-9])  # hi = +-0 with v != 0; ties
        hi, code = tensors.lo8_encode(v)
        return hi, code, tensors.lo8_decode(hi, code)

    def fill(pl, hc):  # hi planes + 8-bit lo planes of the first 8 planes
        hi, code = hc[:2]
        pl.hi[:, :pf] = hi.reshape(n, pf, 8, h, w).permute(0, 1, 3, 4, 2).to(device)
        pl.lo8.copy_(code.reshape(n, pf, 8, h, w).permute(0, 1, 3, 4, 2).to(device))

    if case == 'first_layer':
        x = _rand((n, 3, h, w), 1)
        wt, b = _rand((64, 3, 3, 3), 2, 0.2), _rand((64,), 3, 0.1)
        wts = ops.ConvWeights.from_oihw(wt, b, 3, device=device)
        xin = tensors.nchw_to_planes(x.to(device), True)
        out = tensors.Planes.empty(n, pf + 4 * pg, h, w, device, True, PF_F16, lo_planes=pf).with_lo8(pf)
        of32 = tensors.empty_f32map(n, 64, h, w, device)
        p = ops.conv_params(wts, xin, h, w, out=out, out_f32=of32, out_lo8=True)
        assert p.lo8_flags == L.LO8_OUT and 'XRES' not in L.conv_kernel_name(p)
        ops.run_convs([p], device)
        torch.cuda.synchronize()
        ref = tensors.f32map_to_nchw(of32, 64).cpu()  # the kernel's own f32 result: the planes must be its (hi, code) split, bit for bit
        hi, code = tensors.lo8_encode(ref)
        assert torch.equal(tensors.planes_to_nchw(tensors.Planes(out.hi[:, :pf].contiguous(), None), 64).cpu(), hi.float())
        assert torch.equal(out.lo8.cpu().permute(0, 1, 4, 2, 3).reshape(n, 64, h, w), code)
        got = tensors.planes_to_nchw(tensors.Planes(out.hi[:, :pf].contiguous(), None, out.lo8), 64, lo8=True).cpu()
        assert (got - ref).abs().max().item() <= 2.0**-19 * ref.abs().max().item()
        return
    xs, r0s = stream(5), stream(6)
    x, r0v = xs[2], r0s[2]
    ws = tensors.Planes.empty(n, pf + 4 * pg, h, w, device, True, PF_F16, lo_planes=pf).with_lo8(pf)
    ws.hi.zero_()
    fill(ws, xs)
    ws.hi[:, pf:] = tensors.nchw_to_planes(_rand((n, 128, h, w), 7).to(device), False, PF_F16).hi  # the growth channels x1 .. x4
    r0 = tensors.Planes.empty(n, pf + 4 * pg, h, w, device, True, PF_F16, lo_planes=pf).with_lo8(pf)
    fill(r0, r0s)
    out = tensors.Planes.empty(n, pf + 4 * pg, h, w, device, True, PF_F16, lo_planes=pf).with_lo8(pf)
    wt = (torch.rand((64, 192, 3, 3), generator=g) * 2 - 1) / (192 * 9) ** 0.5
    b = (torch.rand((64,), generator=g) * 2 - 1) * 0.1
    wts = ops.ConvWeights.from_oihw(wt, b, 1, device=device, fmt=PF_F16)
    cat = tensors.planes_to_nchw(tensors.Planes(ws.hi, None), 192).cpu()  # what the multiply reads: hi planes only
    y = _conv(cat, _h(wt), b)
    two = case != 'one_residual'
    ref = y * 0.2 + x
    if two:
        ref = ref * 0.2 + r0v
    kw = dict(res2=(r0, 0, 'lo8'), beta=0.2) if two else {}
    p = ops.conv_params(wts, ws, h, w, cin_planes=24, res1=(ws, 0, 'lo8'), alpha=0.2, out=out, out_lo8=case != 'last_block', **kw)
    name = L.conv_kernel_name(p)
    assert 'XRES' in name, name  # all three combinations have a direct instantiation (XRES 4: everything 8-bit; 5: the residuals only)
    ops.run_convs([p], device)
    torch.cuda.synchronize()
    assert L.ring_aborts() == 0
    L.check_status('test')
    if case == 'last_block':
        got = tensors.planes_to_nchw(tensors.Planes(out.hi[:, :pf].contiguous(), out.lo[:, :pf].contiguous()), 64).cpu()
        tol = 2e-5
    else:
        got = tensors.planes_to_nchw(tensors.Planes(out.hi[:, :pf].contiguous(), None, out.lo8), 64, lo8=True).cpu()
        tol = 2.0**-19 * ref.abs().max().item() + 2e-5  # the coding's step is 2^-19 of the value; 2e-5: summation order of a 1728-term sum of fp16 products
    assert (got - ref).abs().max().item() <= tol, (got - ref).abs().max().item()
