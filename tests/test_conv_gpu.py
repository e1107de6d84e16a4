"""GPU parity of the fused convolution kernel (C-ABI rsa_conv2d_list) against torch CPU fp32.

Tolerances: products=3 (split bf16, ~16-bit operands) is checked at 2e-5 * scale, products=1
(plain bf16 operands) at 1.5e-2 * scale, where scale = max|reference|.
"""

import pytest
import torch
import torch.nn.functional as F

from resselt_amd.engine import lib as L
from resselt_amd.engine import ops, tensors

pytestmark = pytest.mark.gpu

TOL = {3: 2e-5, 1: 1.5e-2}


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * 2 - 1) * scale


def _ref_conv(x, w, b, up=False):
    if up:
        x = F.interpolate(x, scale_factor=2, mode='nearest')
    return F.conv2d(x, w, b, padding=w.shape[-1] // 2)


def _check(got, ref, products, what):
    scale = ref.abs().max().item()
    err = (got.cpu() - ref).abs().max().item()
    assert err <= TOL[products] * max(scale, 1e-6), f'{what}: max-abs {err:.3e} vs scale {scale:.3e}'


@pytest.mark.parametrize('products', [3, 1])
@pytest.mark.parametrize(
    'n,cin,cout,h,w,k',
    [
        (2, 64, 32, 20, 45, 3),  # ragged tile edges, two tiles in x
        (1, 192, 64, 9, 33, 3),  # deepest RDB conv
        (1, 3, 64, 16, 32, 3),  # first conv, channel padding 3 -> 8
        (1, 64, 3, 17, 31, 3),  # last conv, cout padding
        (1, 192, 48, 8, 40, 1),  # SPAN conv_cat (k1)
        (1, 240, 720, 8, 32, 1),  # SwinIR qkv linear: several cout slabs
        (1, 48, 48, 24, 24, 3),  # SPAN width: half-filled last K chunk
    ],
)
def test_conv_plain(device, products, n, cin, cout, h, w, k):
    x = _rand((n, cin, h, w), 1)
    wt = _rand((cout, cin, k, k), 2, 1.0 / (cin * k * k) ** 0.5)
    b = _rand((cout,), 3, 0.1)
    ref = _ref_conv(x, wt, b)
    wts = ops.ConvWeights.from_oihw(wt, b, products, device=device)
    xin = tensors.nchw_to_planes(x.to(device), with_lo=True)
    out = tensors.Planes.empty(n, (cout + 7) // 8, h, w, device)
    of32 = tensors.empty_f32map(n, cout, h, w, device)
    onchw = torch.empty((n, cout, h, w), dtype=torch.float32, device=device)
    p = ops.conv_params(wts, xin, h, w, out=out, out_f32=of32)
    p2 = ops.conv_params(wts, xin, h, w, out_nchw=onchw)  # final-store instantiation
    ops.run_convs([p, p2], device)
    torch.cuda.synchronize()
    _check(onchw, ref, products, 'out_nchw')
    _check(tensors.f32map_to_nchw(of32, cout), ref, products, 'out_f32')
    # split planes hold ~16 bits: compare at the looser of the two tolerances
    got = tensors.planes_to_nchw(out, cout)
    scale = ref.abs().max().item()
    assert (got.cpu() - ref).abs().max().item() <= max(TOL[products], 1e-5) * scale * 1.5
    # channels padded up to the plane boundary must be exact zeros
    full = tensors.planes_to_nchw(out, out.planes * 8)
    if out.planes * 8 > cout:
        assert full[:, cout:].abs().max().item() == 0.0


@pytest.mark.parametrize('products', [3, 1])
def test_conv_rdb_epilogue(device, products):
    """conv5 of the last RDB of an RRDB: (acc*0.2 + x)*0.2 + x0, virtual concat input, plane-offset output."""
    n, h, w = 1, 19, 37
    x = _rand((n, 192, h, w), 5)
    wt = _rand((64, 192, 3, 3), 6, 1.0 / (192 * 9) ** 0.5)
    b = _rand((64,), 7, 0.1)
    r1 = _rand((n, 64, h, w), 8)
    r2 = _rand((n, 64, h, w), 9)
    ref = (_ref_conv(x, wt, b) * 0.2 + r1) * 0.2 + r2
    wts = ops.ConvWeights.from_oihw(wt, b, products, device=device)
    # input lives at plane offset 3 of a wider buffer (virtual concat / workspace reuse)
    big = tensors.Planes.empty(n, 30, h, w, device)
    big.hi.zero_()
    big.lo.zero_()
    src = tensors.nchw_to_planes(x.to(device))
    big.hi[:, 3:27] = src.hi
    big.lo[:, 3:27] = src.lo
    out = tensors.Planes.empty(n, 12, h, w, device)
    out.hi.fill_(7.0)
    out.lo.fill_(7.0)
    of32 = tensors.empty_f32map(n, 64, h, w, device)
    p = ops.conv_params(
        wts, big, h, w, in_plane0=3, res1=tensors.nchw_to_f32map(r1.to(device)), alpha=0.2,
        res2=tensors.nchw_to_f32map(r2.to(device)), beta=0.2, out=out, out_plane_off=4, out_f32=of32,
    )  # fmt: skip
    ops.run_convs([p], device)
    torch.cuda.synchronize()
    _check(tensors.f32map_to_nchw(of32, 64), ref, products, 'rdb epilogue f32')
    got = tensors.planes_to_nchw(tensors.Planes(out.hi[:, 4:12].contiguous(), out.lo[:, 4:12].contiguous()), 64)
    assert (got.cpu() - ref).abs().max().item() <= max(TOL[products], 1e-5) * ref.abs().max().item() * 1.5
    # planes outside [4,12) untouched
    assert torch.all(out.hi[:, :4].float() == 7.0) and torch.all(out.lo[:, :4].float() == 7.0)


@pytest.mark.parametrize('products', [3, 1])
def test_conv_upsample_lrelu(device, products):
    """upconv_block: nearest x2 fused into the read, LeakyReLU(0.2) epilogue."""
    n, h, w = 1, 11, 21
    x = _rand((n, 64, h, w), 11)
    wt = _rand((64, 64, 3, 3), 12, 1.0 / (64 * 9) ** 0.5)
    b = _rand((64,), 13, 0.1)
    ref = F.leaky_relu(_ref_conv(x, wt, b, up=True), 0.2)
    wts = ops.ConvWeights.from_oihw(wt, b, products, device=device)
    xin = tensors.nchw_to_planes(x.to(device))
    onchw = torch.empty((n, 64, 2 * h, 2 * w), dtype=torch.float32, device=device)
    p = ops.conv_params(wts, xin, 2 * h, 2 * w, upsample2x=True, act=L.ACT_LRELU, act_param=0.2, out_nchw=onchw)
    ops.run_convs([p], device)
    torch.cuda.synchronize()
    _check(onchw, ref, products, 'upconv')


@pytest.mark.parametrize('act,fn', [(L.ACT_MISH, F.mish), (L.ACT_SILU, F.silu), (L.ACT_GELU, F.gelu)])
def test_conv_activations(device, act, fn):
    n, h, w = 1, 13, 18
    x = _rand((n, 48, h, w), 21, 2.0)
    wt = _rand((48, 48, 3, 3), 22, 3.0 / (48 * 9) ** 0.5)
    b = _rand((48,), 23, 0.5)
    pre_ref = _ref_conv(x, wt, b)
    ref = fn(pre_ref)
    wts = ops.ConvWeights.from_oihw(wt, b, 3, device=device)
    xin = tensors.nchw_to_planes(x.to(device))
    onchw = torch.empty((n, 48, h, w), dtype=torch.float32, device=device)
    p = ops.conv_params(wts, xin, h, w, act=act, out_nchw=onchw)
    ops.run_convs([p], device)
    torch.cuda.synchronize()
    _check(onchw, ref, 3, 'activation')


def test_conv_spab_gate_and_pixelshuffle(device):
    n, h, w = 2, 10, 35
    x = _rand((n, 48, h, w), 31)
    wt = _rand((48, 48, 3, 3), 32, 2.0 / (48 * 9) ** 0.5)
    b = _rand((48,), 33, 0.2)
    skip = _rand((n, 48, h, w), 34)
    o3 = _ref_conv(x, wt, b)
    ref = (o3 + skip) * (torch.sigmoid(o3) - 0.5)
    wts = ops.ConvWeights.from_oihw(wt, b, 3, device=device)
    xin = tensors.nchw_to_planes(x.to(device))
    of32 = tensors.empty_f32map(n, 48, h, w, device)
    p = ops.conv_params(wts, xin, h, w, act=L.ACT_SPAB_GATE, res1=tensors.nchw_to_f32map(skip.to(device)), out_f32=of32)
    ops.run_convs([p], device)
    torch.cuda.synchronize()
    _check(tensors.f32map_to_nchw(of32, 48), ref, 3, 'spab gate')
    # conv -> PixelShuffle(4) with an output affine, stored as fp16
    wt2 = _rand((48, 48, 3, 3), 35, 1.0 / (48 * 9) ** 0.5)
    ref2 = F.pixel_shuffle(_ref_conv(x, wt2, None), 4) * 0.5 + torch.tensor([0.1, 0.2, 0.3]).view(1, 3, 1, 1)
    wts2 = ops.ConvWeights.from_oihw(wt2, None, 3, device=device)
    o16 = torch.empty((n, 3, 4 * h, 4 * w), dtype=torch.float16, device=device)
    shift = torch.tensor([0.1, 0.2, 0.3], device=device)
    p2 = ops.conv_params(wts2, xin, h, w, out_nchw=o16, pixel_shuffle=4, out_scale=0.5, out_shift=shift)
    ops.run_convs([p2], device)
    torch.cuda.synchronize()
    assert (o16.float().cpu() - ref2).abs().max().item() <= 2e-3


def test_layout_kernels(device):
    x = _rand((2, 5, 7, 9), 41)
    mean = torch.tensor([0.1, 0.2, 0.3, 0.4, 0.5])
    out = tensors.Planes.empty(2, 1, 7, 9, device)
    ops.nchw_to_planes(x.to(device), out, mean.to(device), 255.0)
    back = ops.planes_to_nchw(out, 8)
    torch.cuda.synchronize()
    ref = (x - mean.view(1, 5, 1, 1)) * 255.0
    assert (back[:, :5].cpu() - ref).abs().max().item() <= 2e-5 * 255
    assert back[:, 5:].abs().max().item() == 0.0
    xh = x.half().to(device)
    ops.nchw_to_planes(xh, out)
    assert (ops.planes_to_nchw(out, 5).cpu() - xh.float().cpu()).abs().max().item() <= 1e-5


def test_layout_kernel_reflect_pad_and_unshuffle(device):
    import torch.nn.functional as F2

    x = _rand((2, 3, 21, 30), 51)
    # reflect padding to a multiple of 8 (SwinIR.check_image_size)
    out = tensors.Planes.empty(2, 1, 24, 32, device)
    ops.nchw_to_planes(x.to(device), out)
    ref = F2.pad(x, (0, 2, 0, 3), 'reflect')
    assert (ops.planes_to_nchw(out, 3).cpu() - ref).abs().max().item() <= 1e-5
    # reflect pad to even size + pixel_unshuffle(2) (RRDBNet x2plus front end)
    out2 = tensors.Planes.empty(2, 2, 11, 15, device)
    ops.nchw_to_planes(x.to(device), out2, unshuffle=2)
    ref2 = F2.pixel_unshuffle(F2.pad(x, (0, 0, 0, 1), 'reflect'), 2)
    got2 = ops.planes_to_nchw(out2, 16).cpu()
    assert (got2[:, :12] - ref2).abs().max().item() <= 1e-5 and got2[:, 12:].abs().max().item() == 0.0


def test_argument_errors(device):
    wts = ops.ConvWeights.from_oihw(torch.zeros(8, 8, 3, 3), None, 3, device=device)
    xin = tensors.Planes.empty(1, 1, 8, 8, device)
    p = ops.conv_params(wts, xin, 8, 8)
    p.ksize = 5
    with pytest.raises(RuntimeError, match='ksize'):
        ops.run_convs([p], device)


@pytest.mark.parametrize('layout', [0, 1, 2, 3])
def test_pack_weights_kernel_matches_torch_packers(device, layout):
    """rsa_pack_weights (csrc/pack.hip) against the torch restatement of the three blob layouts (engine/pack.py)."""
    from resselt_amd.engine import pack

    cases = {
        0: [(27, 61, 8, 3, 3), (64, 192, 24, 3, 3), (3, 64, 8, 3, 1), (720, 240, 30, 1, 3), (20, 11, 2, 3, 3)],
        1: [(27, 61, 8, 3, 3), (64, 192, 24, 3, 3)],
        2: [(48, 48, 6, 3, 3), (43, 77, 10, 3, 3), (20, 11, 2, 3, 3)],
        3: [(64, 64, 8, 3, 3)],
    }[layout]
    ref = {0: lambda w, pl, pr: pack.pack_conv_weights(w, pl, pr), 1: lambda w, pl, pr: pack.pack_conv_weights_pairs(w, pl), 2: lambda w, pl, pr: pack.pack_conv_weights_halfpairs(w, pl), 3: lambda w, pl, pr: pack.pack_conv_weights_upphase(w)}[layout]
    for cout, cin, planes, k, products in cases:
        w = _rand((cout, cin, k, k), cout + cin)
        got = ops.pack_weights_device(w.to(device), planes, products, layout).cpu()
        want = ref(w, planes, products)
        assert got.numel() == want.numel()
        assert torch.equal(got.view(torch.int16), want.reshape(-1).view(torch.int16)), (layout, cout, cin, k, products)


@pytest.mark.parametrize(
    'n,cin,cout,h,w,up',
    [
        (1, 160, 32, 530, 1000, False),  # 1088 tiles: every workgroup runs 4-5 tiles (odd counts), both streams, ragged right/bottom edges
        (2, 64, 32, 200, 300, False),    # batch 2, 260 tiles: 252 workgroups with one tile, 4 with two
        (1, 192, 64, 270, 480, False),   # one-stream shape, 6 chunks, 510 tiles
        (1, 64, 64, 180, 260, True),     # nearest x2 folded into the loader (source 90 x 130)
        (1, 96, 24, 40, 70, False),      # cout not a multiple of 16: padded cout rows are zero weights
        (2, 48, 48, 150, 260, False),    # three cout tiles, three half chunks: half mode (layout 2), the SPAN family's layers
        (1, 80, 40, 70, 500, False),     # half mode with five half chunks, ragged cout
        (1, 64, 48, 200, 333, False),    # three cout tiles over whole chunks
    ],
)
def test_conv_ring_schedule_whole_map(device, n, cin, cout, h, w, up):
    """The ring schedule (csrc/conv_ring.h) over maps with many tiles per workgroup, compared over the WHOLE map."""
    x = _rand((n, cin, h // 2, w // 2) if up else (n, cin, h, w), 21)
    wt = _rand((cout, cin, 3, 3), 22, 1.0 / (cin * 9) ** 0.5)
    b = _rand((cout,), 23, 0.1)
    ref = F.leaky_relu(_ref_conv(x, wt, b, up=up), 0.2)
    wts = ops.ConvWeights.from_oihw(wt, b, 3, device=device)
    xin = tensors.nchw_to_planes(x.to(device))
    out = tensors.Planes.empty(n, (cout + 7) // 8, h, w, device)
    of32 = tensors.empty_f32map(n, cout, h, w, device)
    p = ops.conv_params(wts, xin, h, w, upsample2x=up, out=out, out_f32=of32, act=L.ACT_LRELU, act_param=0.2)
    assert p.w_layout == (2 if cin % 32 else 1) and 'conv_ring' in L.conv_kernel_name(p)
    before = L.ring_aborts()
    ops.run_convs([p], device)
    torch.cuda.synchronize()
    assert L.ring_aborts() == before == 0
    _check(tensors.f32map_to_nchw(of32, cout), ref, 3, 'ring f32')
    got = tensors.planes_to_nchw(out, cout)
    assert (got.cpu() - ref).abs().max().item() <= 1.5e-5 * ref.abs().max().item() * 1.5


def test_conv_ring_agrees_with_lockstep_schedule(device):
    """Same layer through the ring schedule and (rsa_debug_set_ring(0)) through the chunk-barrier kernels: both within tolerance of
    torch and of each other (the K order differs, so not bit-equal)."""
    n, cin, cout, h, w = 1, 128, 32, 90, 170
    x = _rand((n, cin, h, w), 31)
    wt = _rand((cout, cin, 3, 3), 32, 1.0 / (cin * 9) ** 0.5)
    ref = _ref_conv(x, wt, None)
    wts = ops.ConvWeights.from_oihw(wt, None, 3, device=device)
    xin = tensors.nchw_to_planes(x.to(device))
    outs = []
    lib = L.load()
    try:
        for mode in (1, 0):
            lib.rsa_debug_set_ring(mode)
            of32 = tensors.empty_f32map(n, cout, h, w, device)
            p = ops.conv_params(wts, xin, h, w, out_f32=of32)
            assert p.w_layout == mode
            ops.run_convs([p], device)
            torch.cuda.synchronize()
            outs.append(tensors.f32map_to_nchw(of32, cout).cpu())
    finally:
        lib.rsa_debug_set_ring(-1)
    for o in outs:
        _check(o, ref, 3, 'schedule')
    assert (outs[0] - outs[1]).abs().max().item() <= 2e-5 * ref.abs().max().item()


@pytest.mark.parametrize('ring', [1, 0])
def test_conv_rdb_epilogue_plane_residuals(device, ring):
    """conv5 of the last RDB of an RRDB with both residuals given as split planes (value = hi + lo), the second one being the planes
    the launch overwrites in place -- the RRDBNet plan's layout (archs/esrgan/arch.py)."""
    n, h, w = 1, 37, 70
    x = _rand((n, 192, h, w), 41)
    wt = _rand((64, 192, 3, 3), 42, 1.0 / (192 * 9) ** 0.5)
    b = _rand((64,), 43, 0.1)
    r2 = _rand((n, 64, h, w), 44)
    src = tensors.nchw_to_planes(x.to(device))  # residual 1 = the first 64 channels of the conv input, as in an RDB
    dst = tensors.Planes.empty(n, 24, h, w, device)
    r2p = tensors.nchw_to_planes(r2.to(device))
    dst.hi[:, :8] = r2p.hi
    dst.lo[:, :8] = r2p.lo
    x64 = tensors.planes_to_nchw(tensors.Planes(src.hi[:, :8].contiguous(), src.lo[:, :8].contiguous()), 64).cpu()
    r2q = tensors.planes_to_nchw(r2p, 64).cpu()
    ref = (_ref_conv(x, wt, b) * 0.2 + x64) * 0.2 + r2q
    wts = ops.ConvWeights.from_oihw(wt, b, 3, device=device)
    lib = L.load()
    try:
        lib.rsa_debug_set_ring(ring)
        p = ops.conv_params(wts, src, h, w, res1=(src, 0), alpha=0.2, res2=(dst, 0), beta=0.2, out=dst, out_plane_off=0)
        assert p.w_layout == ring
        ops.run_convs([p], device)
        torch.cuda.synchronize()
    finally:
        lib.rsa_debug_set_ring(-1)
    got = tensors.planes_to_nchw(tensors.Planes(dst.hi[:, :8].contiguous(), dst.lo[:, :8].contiguous()), 64)
    assert (got.cpu() - ref).abs().max().item() <= 3e-5 * ref.abs().max().item()
    assert L.ring_aborts() == 0


@pytest.mark.parametrize('n,h,w', [(1, 360, 520), (2, 66, 130), (1, 34, 70), (1, 2, 2)])
def test_conv_upsample_as_four_phase_kernels(device, n, h, w):
    """nearest x2 + 3x3 + LeakyReLU, 64 -> 64, split-plane output: csrc/conv_ring_up.h (four 2x2 convolutions on the source map, weight
    layout 3) against torch on the upsampled image, over the whole map (ragged source tiles, every border), and against the schedule that
    multiplies all nine taps on the upsampled halo tile (RSA_CONV_UP2 is read once per process, so that one runs through an f32 output)."""
    x = _rand((n, 64, h // 2, w // 2), 51)
    wt = _rand((64, 64, 3, 3), 52, 1.0 / (64 * 9) ** 0.5)
    b = _rand((64,), 53, 0.1)
    ref = F.leaky_relu(_ref_conv(x, wt, b, up=True), 0.2)
    wts = ops.ConvWeights.from_oihw(wt, b, 3, device=device)
    xin = tensors.nchw_to_planes(x.to(device))
    out = tensors.Planes.empty(n, 8, h, w, device)
    out.hi.fill_(float('nan'))
    p = ops.conv_params(wts, xin, h, w, upsample2x=True, out=out, act=L.ACT_LRELU, act_param=0.2)
    assert p.w_layout == 3 and 'conv_ring_up2' in L.conv_kernel_name(p)
    before = L.ring_aborts()
    ops.run_convs([p], device)
    torch.cuda.synchronize()
    assert L.ring_aborts() == before == 0
    got = tensors.planes_to_nchw(out, 64).cpu()
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
    of32 = tensors.empty_f32map(n, 64, h, w, device)
    out2 = tensors.Planes.empty(n, 8, h, w, device)
    p2 = ops.conv_params(wts, xin, h, w, upsample2x=True, out=out2, out_f32=of32, act=L.ACT_LRELU, act_param=0.2)  # f32 output: the nine-tap schedule
    assert p2.w_layout == 1
    ops.run_convs([p2], device)
    torch.cuda.synchronize()
    assert (tensors.planes_to_nchw(out2, 64).cpu() - got).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize('cout,dtype,ps', [(3, torch.float32, 1), (3, torch.uint8, 1), (12, torch.float16, 2), (32, torch.bfloat16, 1)])
def test_conv_ring_final_store_small_cout(device, cout, dtype, ps):
    """The last convolution of RRDBNet / SwinIR's nearest+conv head (64 -> 3, archs/esrgan/arch.py:120-126 of the reference) on the two-stream
    ring shape with a final store (round 3): plain tensor of any dtype, 8-bit image, depth-to-space; Cout <= 16 leaves the second cout tile
    multiplying zero weights.  Several tiles per stream, ragged edges, two images."""
    n, cin, h, w = 2, 64, 70, 150
    x = _rand((n, cin, h, w), 21)
    wt = _rand((cout, cin, 3, 3), 22, 1.0 / (cin * 9) ** 0.5)
    b = _rand((cout,), 23, 0.1) + (0.5 if dtype == torch.uint8 else 0.0)
    ref = _ref_conv(x, wt, b)
    wts = ops.ConvWeights.from_oihw(wt, b, 3, device=device)
    xin = tensors.nchw_to_planes(x.to(device), with_lo=True)
    oc = cout // (ps * ps)
    shape = (n, h * ps, w * ps, oc) if dtype == torch.uint8 else (n, oc, h * ps, w * ps)
    out = torch.empty(shape, dtype=dtype, device=device)
    p = ops.conv_params(wts, xin, h, w, out_nchw=out, pixel_shuffle=ps)
    assert 'conv_ring<2,0,1>' in L.conv_kernel_name(p)
    ops.run_convs([p], device)
    torch.cuda.synchronize()
    assert L.ring_aborts() == 0
    if ps > 1:
        ref = F.pixel_shuffle(ref, ps)
    if dtype == torch.uint8:
        want = (ref.clamp(0, 1) * 255).round().permute(0, 2, 3, 1)
        diff = (out.cpu().float() - want).abs()
        assert diff.max().item() <= 1 and (diff > 0).float().mean().item() < 1e-3  # a rounding tie broken by the 1e-6 arithmetic difference
    else:
        tol = {torch.float32: 2e-5, torch.float16: 1e-3, torch.bfloat16: 8e-3}[dtype]
        assert (out.cpu().float() - ref).abs().max().item() <= tol * ref.abs().max().item()
