"""The BASELINE.json configurations at (or as near as the CPU oracle allows to) their own size and depth, run by the `-m gpu` suite:

  C2  RealESRGAN-x4plus (RRDBNet-23) bf16x3: a 256 x 256 crop of the bench frame against the oracle (the 1080p frame itself is covered by
      the tile-consistency property in test_rrdbnet_gpu.py); plus a checkpoint with heavy-tailed weights and outlier channels
  C3  SPANPlus 4x fp16, the full batch of 8 x 3 x 512 x 512
  C4  SwinIR-L 4x (embed 240, 9 x 6 blocks, 8 heads, window 8, nearest+conv, 3conv) at full depth on 256 x 256 against the oracle (bf16 and
      fp32 tensors), and at 1024 x 1024 through a size-independent property: the central region of the frame is reproduced by a sub-frame
  C5  the 4320 x 7680 input through the tiling driver on ONE GPU: bounded memory, and tiles agree with direct runs of their halo-padded crops

Tolerances are written where they are used; bf16x3 is the engine's default precision mode.
"""

import pytest
import torch

import resselt_amd
from resselt_amd.utils import synth

from helpers import oracle_forward

pytestmark = pytest.mark.gpu


# north_star: <= 1e-3 max-abs vs CPU fp32.  The engine's own bars: 'auto' (RRDBNet: residual dense blocks in ONE fp16 product, head / tail in
# three bf16 products -- the default) 2e-4; 'bf16x3' (three products everywhere, the conservative mode) 1e-4.
C2_BAR = {'auto': 2e-4, 'bf16x3': 1e-4}


@pytest.mark.parametrize('precision', ['auto', 'bf16x3'])
def test_c2_rrdbnet23_crop_of_the_bench_frame(device, precision):
    from resselt_amd.engine import lib as L

    sd = synth.rrdbnet_state_dict(nb=23, seed=0)
    x = synth.synth_input((1, 3, 1080, 1920), seed=0)[:, :, 400:656, 800:1056].contiguous()
    with torch.no_grad():
        ref = oracle_forward(dict(arch='esrgan'), sd, x)
    m = resselt_amd.load_from_state_dict(dict(sd)).to(device)
    assert m.precision == 'auto' and m.resolved_precision() == 'mixed'  # the default IS the policy
    m.precision = precision
    y = m(x.to(device))
    torch.cuda.synchronize()
    L.check_status('C2 crop')
    err = (y.cpu() - ref).abs().max().item()
    print(f'C2 crop 256^2 {precision}: max-abs {err:.3e} (|y|max {ref.abs().max():.2f})')
    assert err <= C2_BAR[precision]


@pytest.mark.parametrize('precision', ['auto', 'bf16x3'])
def test_c2_rrdbnet23_heavy_tailed_weights_and_outlier_channels(device, precision):
    from resselt_amd.engine import lib as L

    sd = synth.rrdbnet_heavy_tailed_state_dict(nb=23, seed=4)
    x = synth.synth_input((1, 3, 96, 112), seed=4)
    with torch.no_grad():
        ref = oracle_forward(dict(arch='esrgan'), sd, x)
    m = resselt_amd.load_from_state_dict(dict(sd)).to(device)
    m.precision = precision
    y = m(x.to(device))
    torch.cuda.synchronize()
    L.check_status('C2 heavy-tailed')
    scale = max(1.0, ref.abs().max().item())
    err = (y.cpu() - ref).abs().max().item()
    print(f'C2 heavy-tailed {precision}: max-abs {err:.3e} (|y|max {ref.abs().max():.2f})')
    assert torch.isfinite(y).all() and err <= 2e-4 * scale


def test_c2_fp16_range_guard_falls_back_to_three_products(device):
    """A weight beyond the fp16 range: 'auto' packs the conservative mode instead (with a warning); asking for 'mixed' raises."""
    sd = synth.rrdbnet_state_dict(nb=1, seed=2)
    sd['model.1.sub.0.RDB2.conv3.0.weight'][3, 5, 1, 1] = 7.0e4
    x = synth.synth_input((1, 3, 24, 40), seed=2)
    with torch.no_grad():
        ref = oracle_forward(dict(arch='esrgan'), sd, x)
    m = resselt_amd.load_from_state_dict(dict(sd)).to(device)
    with pytest.warns(RuntimeWarning, match='fp16 range'):
        y = m(x.to(device))
    assert m.resolved_precision() == 'bf16x3'
    assert (y.cpu() - ref).abs().max().item() <= 2e-4 * max(1.0, ref.abs().max().item())
    m2 = resselt_amd.load_from_state_dict(dict(sd)).to(device)
    m2.precision = 'mixed'
    with pytest.raises(ValueError, match='fp16 range'):
        m2(x.to(device))


def test_c3_spanplus_x4_fp16_full_batch(device):
    sd = synth.spanplus_state_dict(upscale=4, upsampler='ps', seed=0)
    x = synth.synth_input((8, 3, 512, 512), seed=0).half()
    with torch.no_grad():
        ref = oracle_forward(dict(arch='spanplus'), sd, x.float())  # the oracle sees the same (fp16-rounded) input values
    m = resselt_amd.load_from_state_dict(dict(sd)).to(device)
    y = m(x.to(device))
    assert y.dtype == torch.float16 and tuple(y.shape) == (8, 3, 2048, 2048)
    err = (y.float().cpu() - ref).abs().max().item()
    amax = ref.abs().max().item()
    import math

    half_ulp = 2.0 ** (math.floor(math.log2(amax)) - 11)  # of the largest output values (fp16: 10 fraction bits)
    print(f'C3 8x3x512x512 fp16 ({m.resolved_precision()}): max-abs {err:.3e} (|y|max {amax:.2f}, half an ulp there {half_ulp:.2e})')
    # the output rounding itself (half an fp16 ulp) + the 1e-4 the arithmetic is allowed
    assert m.resolved_precision() == 'mixed' and err <= half_ulp + 1e-4


@pytest.fixture(scope='module')
def swinir_l():
    return synth.swinir_state_dict(embed_dim=240, depths=[6] * 9, num_heads=[8] * 9, upscale=4, upsampler='nearest+conv', resi='3conv', seed=0)


def test_c4_swinir_l_full_depth_256(device, swinir_l):
    x = synth.synth_input((1, 3, 256, 256), seed=0).bfloat16()
    with torch.no_grad():
        ref = oracle_forward(dict(arch='swinir'), swinir_l, x.float())
    m = resselt_amd.load_from_state_dict(dict(swinir_l)).to(device)
    amax = ref.abs().max().item()
    assert m.precision == 'auto' and m.resolved_precision() == 'mixed'  # the default: blocks in one fp16 product (two windows per CU), head in three
    for precision, bar in (('bf16x3', 1e-4), ('auto', 2e-4)):
        m.precision = precision
        y32 = m(x.float().to(device))
        e32 = (y32.cpu() - ref).abs().max().item()
        print(f'C4 SwinIR-L 9x6 256^2 {precision}: fp32 tensors max-abs {e32:.3e} (|y|max {amax:.2f})')
        assert e32 <= bar
    y = m(x.to(device))
    assert y.dtype == torch.bfloat16 and tuple(y.shape) == (1, 3, 1024, 1024)
    err = (y.float().cpu() - ref).abs().max().item()
    print(f'C4 SwinIR-L 9x6 256^2: bf16 tensors max-abs {err:.3e}')
    assert err <= 2.0**-9 * max(amax, 0.25) + 1e-4  # half a bf16 ulp of the output + the arithmetic


def test_c4_swinir_l_1024_center_is_reproduced_by_a_subframe(device, swinir_l):
    """Size-independent property at the BASELINE size: window attention is shift-equivariant for window-aligned shifts, so a region far
    enough from every border (54 blocks x half a window + the convolutions << 256 px) does not depend on where the frame ends."""
    m = resselt_amd.load_from_state_dict(dict(swinir_l)).to(device)
    x = synth.synth_input((1, 3, 1024, 1024), seed=1).to(device)
    full = m(x)
    sub = m(x[:, :, 128:896, 128:896].contiguous())
    a = full[:, :, 4 * 384 : 4 * 640, 4 * 384 : 4 * 640]
    b = sub[:, :, 4 * 256 : 4 * 512, 4 * 256 : 4 * 512]
    err = (a - b).abs().max().item()
    print(f'C4 1024^2 centre vs 768^2 sub-frame: max-abs {err:.3e}')
    assert tuple(full.shape) == (1, 3, 4096, 4096) and err <= 2e-4


def test_c5_8k_input_tiled_on_one_gpu_bounded_memory(device):
    from resselt_amd.tiling import plan_tiles, run_tile, upscale_tiled

    sd = synth.rrdbnet_state_dict(nb=23, seed=0)
    m = resselt_amd.load_from_state_dict(dict(sd)).to(device)
    m.max_plan_bytes = 24 << 30
    import gc

    gc.collect()  # models of earlier tests in this process: their plans are reference cycles until collected
    torch.cuda.empty_cache()
    x = synth.synth_input((1, 3, 4320, 7680), seed=2).to(device)
    torch.cuda.reset_peak_memory_stats(device)
    base = torch.cuda.memory_allocated(device)
    y = upscale_tiled(m, x, 4, tile=(1080, 1920), halo=32)
    torch.cuda.synchronize()
    peak = torch.cuda.max_memory_allocated(device) - base
    print(f'C5 4320x7680 -> {tuple(y.shape)}: peak {peak / 2**30:.1f} GiB above the {base / 2**30:.1f} GiB held before the call')
    assert tuple(y.shape) == (1, 3, 17280, 30720)
    assert peak <= 64 << 30, f'peak {peak / 2**30:.1f} GiB'
    # two tiles (a corner and an interior one) equal direct runs of their halo-padded crops bit for bit, and the interior seam is
    # continuous: neighbours computed from different halos agree where their receptive fields overlap (survey: <= 2e-6 at halo 32)
    tiles = plan_tiles(4320, 7680, 4, 4, halo=32)
    for t in (tiles[0], tiles[5]):
        direct = run_tile(m, x, t, 4)
        assert torch.equal(y[:, :, 4 * t.y0 : 4 * t.y1, 4 * t.x0 : 4 * t.x1], direct)
    wide = m(x[:, :, 1080 - 96 : 1080 + 96, 1920 - 96 : 1920 + 96].contiguous())  # a window centred on the corner where four tiles meet
    seam = y[:, :, 4 * (1080 - 32) : 4 * (1080 + 32), 4 * (1920 - 32) : 4 * (1920 + 32)]
    assert (seam - wide[:, :, 4 * 64 : 4 * 128, 4 * 64 : 4 * 128]).abs().max().item() <= 1e-4
