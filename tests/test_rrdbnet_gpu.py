"""End-to-end GPU parity of the RRDBNet engine against (a) vectors produced by the real reference
(tests/golden, tools/gen_golden.py) and (b) the CPU oracle on seeded inputs.

Tolerances (max-abs, outputs are O(1)): 'auto' (the default: residual dense blocks in one fp16 product, head / tail in three bf16
products) and bf16x3 2e-4 (north_star bar: 1e-3), plain bf16 mode 6e-2 (a smoke bound: that mode is not a default anywhere).
"""

import pytest
import torch

import resselt_amd
from helpers import golden_names, load_golden, synth_state_dict
from resselt_amd.utils import synth

pytestmark = pytest.mark.gpu

TOL = {'auto': 2e-4, 'bf16x3': 2e-4, 'bf16': 6e-2}


def _model(sd, device, precision='bf16x3'):
    m = resselt_amd.load_from_state_dict(dict(sd)).to(device)
    m.precision = precision
    return m


@pytest.fixture(autouse=True)
def _no_failed_hand_offs():
    """Every model-level test ends with the ring schedule's failure word clear (rsa_check_status after a synchronise)."""
    yield
    from resselt_amd.engine import lib as L

    if torch.cuda.is_available():
        torch.cuda.synchronize()
        L.check_status('end of test')


@pytest.mark.parametrize('precision', ['auto', 'bf16x3'])
@pytest.mark.parametrize('name', golden_names('rrdbnet_'))
def test_rrdbnet_matches_reference_vectors(device, name, precision):
    meta, arr = load_golden(name)
    sd = synth_state_dict(meta)
    m = _model(sd, device, precision)
    assert m.precision == precision and (precision != 'auto' or m.resolved_precision() in ('mixed', 'bf16x3'))
    assert vars(m.parameters_info) == {k: meta['metadata'][k] for k in ('in_channels', 'out_channels', 'upscale', 'name')}
    y = m(arr['x'].to(device))
    torch.cuda.synchronize()
    assert y.shape == arr['y'].shape and y.dtype == torch.float32
    err = (y.cpu() - arr['y']).abs().max().item()
    assert err <= TOL[precision], f'{name} {precision}: max-abs {err:.3e}'


@pytest.mark.parametrize('precision', ['auto', 'bf16x3', 'bf16'])
def test_rrdbnet23_vs_oracle_multi_tile(device, precision):
    from oracle.rrdbnet import rrdbnet_forward

    sd = synth.rrdbnet_state_dict(nb=23, seed=5)
    x = synth.synth_input((1, 3, 70, 100), seed=5)
    with torch.no_grad():
        ref = rrdbnet_forward(sd, x)
    m = _model(sd, device, precision)
    y = m(x.to(device))
    err = (y.cpu() - ref).abs().max().item()
    print(f'RRDBNet-23 {precision}: max-abs {err:.3e} (|y|max {ref.abs().max():.2f})')
    assert err <= TOL[precision]
    # same plan, second call, different data: buffers are reused correctly
    x2 = synth.synth_input((1, 3, 70, 100), seed=6)
    with torch.no_grad():
        ref2 = rrdbnet_forward(sd, x2)
    assert (m(x2.to(device)).cpu() - ref2).abs().max().item() <= TOL[precision]


def test_rrdbnet_new_arch_keys_and_dtypes(device):
    sd_old = synth.rrdbnet_state_dict(nb=2, seed=9)
    sd_new = synth.rrdbnet_state_dict(nb=2, seed=9, new_arch=True)
    x = synth.synth_input((2, 3, 33, 47), seed=9).to(device)
    m_old = _model(sd_old, device)
    m_new = _model({'params_ema': sd_new}, device)  # official Real-ESRGAN checkpoints wrap the dict
    y_old, y_new = m_old(x), m_new(x)
    assert torch.equal(y_old, y_new)
    # half / bfloat16 inputs: output dtype follows the input, values follow the f32 result
    for dt, tol in ((torch.float16, 2e-3), (torch.bfloat16, 1.6e-2)):
        y = m_old(x.to(dt))
        assert y.dtype == dt
        ref = m_old(x.to(dt).float())
        assert (y.float() - ref).abs().max().item() <= tol


def test_rrdbnet_rejects_cpu_and_bad_shapes(device):
    m = _model(synth.rrdbnet_state_dict(nb=1), device)
    with pytest.raises(RuntimeError, match='HIP kernels only'):
        m(torch.zeros(1, 3, 8, 8))
    with pytest.raises(RuntimeError, match='input channels'):
        m(torch.zeros(1, 4, 8, 8, device=device))
    with pytest.raises(ValueError):
        m(torch.zeros(3, 8, 8, device=device))


@pytest.mark.parametrize('precision', ['auto', 'bf16x3'])
def test_rrdbnet_full_frame_tile_consistency(device, precision):
    """BASELINE-size property check: a 1080p frame and a halo-padded crop of it agree in the crop's interior.

    (Translation equivariance of the conv stack; survey-measured effective receptive field: halo 32 -> 1.5e-6.)
    """
    sd = synth.rrdbnet_state_dict(nb=23, seed=0)
    m = _model(sd, device, precision)
    x = synth.synth_input((1, 3, 1080, 1920), seed=0).to(device)
    y = m(x)
    assert y.shape == (1, 3, 4320, 7680)
    assert torch.isfinite(y).all()
    y0, x0, s, halo = 400, 900, 128, 40
    crop = x[:, :, y0 - halo : y0 + s + halo, x0 - halo : x0 + s + halo].contiguous()
    yc = m(crop)
    a = y[:, :, 4 * y0 : 4 * (y0 + s), 4 * x0 : 4 * (x0 + s)]
    b = yc[:, :, 4 * halo : 4 * (halo + s), 4 * halo : 4 * (halo + s)]
    assert (a - b).abs().max().item() <= 1e-4
    # image borders use zero padding exactly like the reference: top-left corner equals a corner crop
    corner = m(x[:, :, : s + halo, : s + halo].contiguous())
    assert (y[:, :, : 4 * s, : 4 * s] - corner[:, :, : 4 * s, : 4 * s]).abs().max().item() <= 1e-4


def test_rrdbnet_tiled_driver_on_gpu(device):
    """resselt_amd.tiling over the real engine: sequential tiles with halo reproduce the full frame."""
    from resselt_amd.tiling import TileParallel, upscale_tiled

    sd = synth.rrdbnet_state_dict(nb=23, seed=1)
    m = _model(sd, device, 'auto')
    x = synth.synth_input((1, 3, 150, 210), seed=1).to(device)
    full = m(x)
    tiled = upscale_tiled(m, x, scale=4, tile=(80, 112), halo=40)
    assert tiled.shape == full.shape and (tiled - full).abs().max().item() <= 1e-4
    one = TileParallel(m, scale=4, halo=40, grid=(2, 2))(x)  # world size 1: all four tiles on this GPU
    assert (one - full).abs().max().item() <= 1e-4


def test_uint8_image_round_trip_and_upscale_helper(device):
    """uint8 HWC images either side of the path: conversion kernels bit-exact vs torch, helper == manual pipeline."""
    from resselt_amd.engine import ops
    from resselt_amd.tiling import upscale

    g = torch.Generator().manual_seed(5)
    img = torch.randint(0, 256, (2, 37, 53, 3), generator=g, dtype=torch.uint8)
    for dt in (torch.float32, torch.float16, torch.bfloat16):
        x = ops.image_u8_to_nchw(img.to(device), dt)
        ref = (img.permute(0, 3, 1, 2).float() / 255).to(dt)
        assert torch.equal(x.cpu(), ref)
    y = torch.randn((2, 3, 20, 31), generator=g) * 0.7 + 0.5
    y[0, 0, 0, :4] = torch.tensor([0.5 / 255, 1.5 / 255, 2.5 / 255, float('nan')])  # ties round to even; NaN -> 0
    out = ops.nchw_to_image_u8(y.to(device))
    ref = (torch.nan_to_num(y, nan=0.0).clamp(0, 1) * 255).round().to(torch.uint8).permute(0, 2, 3, 1)
    assert torch.equal(out.cpu(), ref)
    sd = synth.rrdbnet_state_dict(nb=2, scale=4, seed=9)
    m = resselt_amd.load_from_state_dict(dict(sd)).to(device)
    one = upscale(m, img[0].to(device), dtype=torch.float32)
    assert one.shape == (37 * 4, 53 * 4, 3) and one.dtype == torch.uint8
    manual = ops.nchw_to_image_u8(m(ops.image_u8_to_nchw(img[:1].to(device))))[0]
    assert torch.equal(one, manual)
    tiled = upscale(m, img[0].to(device), tile=(24, 32), halo=16, dtype=torch.float32)
    assert (tiled.int() - one.int()).abs().max().item() <= 1  # nb=2: receptive field < halo, so at most a rounding tie
    # the fused path itself: model(uint8 [N, H, W, C]) reads the bytes in the layout kernel and writes bytes from the last convolution's
    # store (rsa_dtype RSA_U8); bit-identical to the two conversion kernels around an fp32 forward, also with a row-banded tail
    both = m(img.to(device))
    manual2 = ops.nchw_to_image_u8(m(ops.image_u8_to_nchw(img.to(device))))
    assert both.dtype == torch.uint8 and tuple(both.shape) == (2, 148, 212, 3) and torch.equal(both, manual2)
    m.tail_band_rows = 16
    assert torch.equal(m(img.to(device)), manual2) and torch.equal(m(ops.image_u8_to_nchw(img.to(device))), m.__class__.forward(m, ops.image_u8_to_nchw(img.to(device))))
    banded = m(ops.image_u8_to_nchw(img.to(device)))
    m.tail_band_rows = 4096
    assert torch.equal(banded, m(ops.image_u8_to_nchw(img.to(device))))  # bands change nothing, bit for bit
    # a model whose kernels do not take 8-bit images (base image in the final store) goes through the conversion kernels
    c = resselt_amd.load_from_state_dict(dict(synth.compact_state_dict(num_feat=32, num_conv=3, upscale=2, seed=4))).to(device)
    cu = c(img[0].to(device))
    assert cu.dtype == torch.uint8 and torch.equal(cu, ops.nchw_to_image_u8(c(ops.image_u8_to_nchw(img[:1].to(device))))[0])


def test_graph_replay_matches_eager(device):
    """use_graph: the forward captured once as a hipGraph and replayed; same values as the eager launch list, new inputs take effect."""
    sd = synth.rrdbnet_state_dict(nb=3, scale=4, seed=12)
    m = resselt_amd.load_from_state_dict(dict(sd)).to(device)
    x1, x2 = synth.synth_input((1, 3, 40, 56), seed=1).to(device), synth.synth_input((1, 3, 40, 56), seed=2).to(device)
    e1, e2 = m(x1).clone(), m(x2).clone()
    m.use_graph = True
    g1 = m(x1)
    g2 = m(x2)
    g1b = m(x1)
    torch.cuda.synchronize()
    assert torch.equal(g1, e1) and torch.equal(g2, e2) and torch.equal(g1b, e1)
    sd2 = synth.compact_state_dict(num_feat=32, num_conv=3, upscale=2, seed=4)  # the final store of this model reads the INPUT as a base image
    c = resselt_amd.load_from_state_dict(dict(sd2)).to(device)
    ec = c(x2).clone()
    c.use_graph = True
    assert torch.equal(c(x1), c(x1)) and torch.equal(c(x2), ec)
