"""GPU parity of gemm_k1's direct epilogues (round 4, csrc/gemm_k1.hip template parameter EPI): the Linear layers of the transformer bodies
(reference: nn.Linear over tokens, resselt/archs/swinir/arch.py:34-40,141,168; dat/arch.py:224-267; hat/arch.py:218-350; drct/arch.py:204-329)
in one fp16 product on hi planes.

  EPI 1  qkv:  fp16 hi planes out, no activation                EPI 2  fc1: the same with GELU
  EPI 3  proj / fc2 / adjust: no activation or LeakyReLU, an optional f32 residual map (* alpha), f32 map and / or fp16 hi planes out

Shapes on purpose off the happy path: token counts that are not multiples of the 64-token tile, batches whose images end inside a tile,
output widths that end inside a cout tile and take one, two and three passes, K in every instantiated width (6 / 8 / 16 chunks).
Reference = f32 arithmetic on the fp16-ROUNDED operands (exact products; accumulation order is the only difference: 2e-5 * scale), GELU in
f32 (erf), then the fp16 rounding of plane outputs (half an ulp)."""

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


CASES = [
    # n, cin, cout, h, w
    (1, 180, 540, 24, 40),  # qkv of the 180-channel bodies: three passes, the last cout tile partial (540 = 33.75 tiles); 6 K-chunks
    (2, 180, 360, 7, 19),  # 133 tokens per image: every image ends inside a tile
    (1, 64, 96, 5, 13),  # the smallest layer the schedule takes (cout >= 96), 65 tokens
    (1, 240, 720, 16, 33),  # 8 K-chunks
    (1, 308, 180, 9, 31),  # 16 K-chunks (DRCT's widest dense block), one pass
    (1, 360, 180, 16, 16),  # fc2 of the bodies
]


@pytest.mark.parametrize('n,cin,cout,h,w', CASES)
@pytest.mark.parametrize('form', ['planes', 'planes_gelu', 'f32_residual', 'planes_residual_lrelu', 'f32_and_planes'])
def test_direct_epilogues(device, n, cin, cout, h, w, form):
    x = _rand((n, cin, h, w), 1)
    wt = _rand((cout, cin, 1, 1), 2, 1.0 / cin**0.5)
    b = _rand((cout,), 3, 0.1)
    res = _rand((n, cout, h, w), 4)
    y = F.conv2d(_h(x).double(), _h(wt).double(), b.double()).float()
    wts = ops.ConvWeights.from_oihw(wt, b, 1, device=device, fmt=PF_F16)
    xin = tensors.nchw_to_planes(x.to(device), with_lo=False, fmt=PF_F16)
    cp = (cin + 7) // 8
    out_pl = tensors.Planes.empty(n, (cout + 7) // 8, h, w, device, with_lo=False, fmt=PF_F16)
    out_pl.hi.fill_(float('nan'))  # every unit of the output planes must be written (channels beyond Cout: zeros)
    of32 = tensors.empty_f32map(n, cout, h, w, device)
    rmap = tensors.nchw_to_f32map(res.to(device))
    kw, ref, want_pl, want_f32 = {}, y, False, False
    if form == 'planes':
        kw, want_pl = dict(out=out_pl), True
    elif form == 'planes_gelu':
        kw, want_pl, ref = dict(out=out_pl, act=L.ACT_GELU), True, F.gelu(y)
    elif form == 'f32_residual':
        kw, want_f32, ref = dict(out_f32=of32, res1=rmap, alpha=0.2), True, y * 0.2 + res
    elif form == 'planes_residual_lrelu':
        kw, want_pl = dict(out=out_pl, res1=rmap, alpha=1.0, act=L.ACT_LRELU, act_param=0.2), True
        ref = F.leaky_relu(y, 0.2) + res
    else:
        kw, want_pl, want_f32 = dict(out=out_pl, out_f32=of32), True, True
    p = ops.conv_params(wts, xin, h, w, cin_planes=cp, **kw)
    assert 'gemm_k1' in L.conv_kernel_name(p)
    ops.run_convs([p], device)
    torch.cuda.synchronize()
    L.check_status('gemm_k1 direct')
    scale = max(1.0, ref.abs().max().item())
    if want_f32:
        got = tensors.f32map_to_nchw(of32, cout).cpu()
        assert (got - ref).abs().max().item() <= 2e-5 * scale
    if want_pl:
        hi = out_pl.hi.cpu().float()
        assert torch.isfinite(hi).all()
        got = tensors.planes_to_nchw(tensors.Planes(out_pl.hi, None), ((cout + 7) // 8) * 8).cpu()
        assert (got[:, :cout] - ref).abs().max().item() <= 2.0**-11 * 1.01 * scale + 2e-5 * scale
        assert (got[:, cout:] == 0).all()  # padded channels are zeros, not stale data
