"""Host-side weight preparation of the transformer / re-parameterised architectures (CPU only, no GPU compute):
fragment-order position-bias tables, head-padded permutations, RepConv / OmniShift / DySample folds against plain torch statements."""

import torch
import torch.nn.functional as F

from resselt_amd.archs.dat.arch import attn_tiles, bias_fragments, pad_heads
from resselt_amd.archs.hat.arch import bias_fragments_qk, rpi_buffers
from resselt_amd.archs.rtmosr.arch import fold_omnishift
from resselt_amd.archs.spanpp.arch import fold_repconv, igconv_kernel
from resselt_amd.utils import synth


def test_bias_fragment_order_matches_accumulator_layout():
    """lane l, element r of tile (qt, kt)  <->  query 32 qt + (l & 31), key 32 kt + (r & 3) + 8 (r >> 2) + 4 (l >> 5)."""
    g = torch.Generator().manual_seed(0)
    for nq, nk in ((64, 64), (96, 96), (256, 576), (16, 36)):
        dense = torch.randn((2, nq, nk), generator=g)
        qt, kt = ((nq + 31) // 32, (nk + 31) // 32) if nq != nk else (attn_tiles(nq), attn_tiles(nk))
        frag = bias_fragments_qk(dense, qt, kt) if nq != nk else bias_fragments(dense)
        assert tuple(frag.shape) == (2, qt, kt, 64, 16)
        for _ in range(300):
            h, a, b, lane, r = (int(torch.randint(0, n, (1,), generator=g)) for n in (2, qt, kt, 64, 16))
            q, k = 32 * a + (lane & 31), 32 * b + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
            got = frag[h, a, b, lane, r].item()
            if k >= nk:
                assert got < -1e29  # padded keys are masked out
            else:
                assert got == (dense[h, q, k].item() if q < nq else 0.0)


def test_hat_rpi_buffers_match_synth_and_shapes():
    sa, oca = rpi_buffers(16, 0.5)
    sa2, oca2 = synth.hat_rpi(16, 0.5)
    assert torch.equal(sa, sa2) and torch.equal(oca, oca2)
    assert sa.shape == (256, 256) and oca.shape == (256, 576) and int(sa.max()) == 31 * 31 - 1 and int(sa.min()) == 0
    # the reference's OCA index is shifted by (ws - ext + 1) < 0 and therefore contains NEGATIVE entries that wrap around when they index
    # the bias table (archs/hat/arch.py:1024-1031); the buffers, the oracle and the engine reproduce that verbatim
    assert int(oca.min()) < 0


def test_pad_heads_scatter():
    t = torch.arange(12.0).reshape(12, 1) * torch.ones(1, 3)
    out = pad_heads(t, heads=3)  # head_dim 4 -> 32 slots per head
    assert out.shape == (96, 3)
    for h in range(3):
        assert torch.equal(out[32 * h : 32 * h + 4], t[4 * h : 4 * h + 4]) and out[32 * h + 4 : 32 * h + 32].abs().max() == 0
    cols = pad_heads(torch.arange(12.0).reshape(1, 12), heads=3, dim=1)
    assert cols.shape == (1, 96) and cols[0, 32].item() == 4.0


def test_repconv_fold_equals_branch_sum():
    """RepConv (SpanPP / RTMoSR): alpha-weighted SeqConv3x3 + 3x3 + Conv3XC as ONE zero-padded 3x3 conv."""
    sd = {k: v.double() for k, v in synth.spanpp_state_dict(feature_channels=16, implicit_dim=32, latent_layers=1, seed=3).items() if v.is_floating_point()}
    key = 'block_2.c2_r'
    x = torch.randn((1, 16, 9, 11), dtype=torch.float64, generator=torch.Generator().manual_seed(1))
    a = sd[f'{key}.alpha']
    # SeqConv3x3 (1x1, pad with its bias, 3x3 valid)
    y0 = F.pad(F.conv2d(x, sd[f'{key}.conv1.k0'], sd[f'{key}.conv1.b0']), (1, 1, 1, 1))
    b0 = sd[f'{key}.conv1.b0'].view(1, -1, 1, 1)
    y0[:, :, :1], y0[:, :, -1:], y0[:, :, :, :1], y0[:, :, :, -1:] = b0, b0, b0, b0
    b1 = F.conv2d(y0, sd[f'{key}.conv1.k1'], sd[f'{key}.conv1.b1'])
    b2 = F.conv2d(x, sd[f'{key}.conv2.weight'], sd[f'{key}.conv2.bias'], padding=1)
    c = f'{key}.conv3'
    t = F.conv2d(F.pad(x, (1, 1, 1, 1)), sd[f'{c}.conv.0.weight'], sd[f'{c}.conv.0.bias'])
    t = F.conv2d(F.conv2d(t, sd[f'{c}.conv.1.weight'], sd[f'{c}.conv.1.bias']), sd[f'{c}.conv.2.weight'], sd[f'{c}.conv.2.bias'])
    b3 = t + F.conv2d(x, sd[f'{c}.sk.weight'], sd[f'{c}.sk.bias'])
    ref = a[0] * b1 + a[1] * b2 + a[2] * b3
    w, b = fold_repconv({k: v.float() for k, v in sd.items()}, key)
    got = F.conv2d(x, w.double(), b.double(), padding=1)
    assert (got - ref).abs().max().item() <= 2e-6 * ref.abs().max().item()


def test_omnishift_fold_equals_branch_sum():
    sd = synth.rtmosr_state_dict(dim=32, n_blocks=1, seed=5)
    key = 'body.0.conv.1'
    x = torch.randn((1, 128, 7, 9), generator=torch.Generator().manual_seed(2))
    ref = (sd[f'{key}.alpha1'] * x + sd[f'{key}.alpha2'] * F.conv2d(x, sd[f'{key}.conv1x1.weight'], sd[f'{key}.conv1x1.bias'], groups=128)
           + sd[f'{key}.alpha3'] * F.conv2d(x, sd[f'{key}.conv3x3.weight'], sd[f'{key}.conv3x3.bias'], padding=1, groups=128)
           + sd[f'{key}.alpha4'] * F.conv2d(x, sd[f'{key}.conv5x5.weight'], sd[f'{key}.conv5x5.bias'], padding=2, groups=128))  # fmt: skip
    w, b = fold_omnishift(sd, key)
    got = F.conv2d(x, w.reshape(128, 1, 5, 5), b, padding=2, groups=128)
    assert (got - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()


def test_igconv_kernel_shapes_per_scale():
    sd = synth.spanpp_state_dict(feature_channels=16, implicit_dim=32, latent_layers=2, seed=1)
    for s in (1, 2, 3, 4):
        k = igconv_kernel(sd, s, 4)
        assert k.shape == (3 * s * s, 16, 3, 3) and torch.isfinite(k).all()
