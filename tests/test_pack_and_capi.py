"""Host logic around the C-ABI: weight packing layout, library symbols, descriptor validation (no GPU compute)."""

import ctypes
import os
import re

import pytest
import torch

from resselt_amd.engine import lib as L
from resselt_amd.engine import pack, tensors

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bf16(x):
    return x.to(torch.bfloat16)


@pytest.mark.parametrize('cout,cin,planes,k,products', [(20, 11, 2, 3, 3), (64, 192, 24, 3, 3), (3, 64, 8, 3, 1), (48, 192, 24, 1, 3), (16, 8, 5, 1, 1)])
def test_pack_layout_matches_index_formula(cout, cin, planes, k, products):
    g = torch.Generator().manual_seed(cout * 1000 + cin)
    w = torch.randn((cout, cin, k, k), generator=g)
    blob = pack.pack_conv_weights(w, planes, products)
    assert tuple(blob.shape) == pack.packed_weight_shape(cout, planes, k, products) and blob.dtype == torch.bfloat16
    hi = _bf16(w)
    lo = _bf16(w - hi.float())
    q_n, t_n, ct_n, nhl, _, _ = blob.shape
    rng = torch.Generator().manual_seed(1)
    for _ in range(400):  # random probes of blob[q][t][ct][hl][lane][j]
        q, t, ct, hl, lane, j = (int(torch.randint(0, n, (1,), generator=rng)) for n in (q_n, t_n, ct_n, nhl, 64, 8))
        co, ci = 16 * ct + (lane & 15), 32 * q + 8 * (lane >> 4) + j
        src = (hi, lo)[hl]
        want = src[co, ci, t // k, t % k].item() if co < cout and ci < cin else 0.0
        assert blob[q, t, ct, hl, lane, j].item() == want
    # hi + lo reproduces the weight to ~2^-17 relative
    assert ((hi.float() + lo.float()) - w).abs().max() <= 2.0 ** -16 * w.abs().max()


def test_packed_bytes_agree_with_library():
    lib = L.load()
    for cout, planes, k, prod in [(64, 24, 3, 3), (3, 8, 3, 1), (720, 30, 1, 3), (48, 6, 3, 3)]:
        shape = pack.packed_weight_shape(cout, planes, k, prod)
        n = 1
        for s in shape:
            n *= s
        assert lib.rsa_packed_weight_bytes(cout, planes, k, prod) == 2 * n
    assert lib.rsa_packed_weight_bytes(0, 1, 3, 3) < 0 and lib.rsa_packed_weight_bytes(8, 1, 2, 3) < 0


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, 'include', 'resselt_amd.h')).read()
    declared = set(re.findall(r'^\s*(?:const\s+)?(?:int|int64_t|char\s*\*|void)\s*\*?\s*(rsa_\w+)\s*\(', header, flags=re.M))
    assert declared, 'no prototypes parsed from the header'
    assert declared == set(L.EXPORTS), (declared ^ set(L.EXPORTS))
    lib = ctypes.CDLL(L.lib_path())
    for name in declared:
        assert hasattr(lib, name), name
    assert L.load().rsa_version() == 400


def test_conv_params_struct_matches_header_field_order():
    header = open(os.path.join(ROOT, 'include', 'resselt_amd.h')).read()
    body = header[header.index('typedef struct rsa_conv_params {') : header.index('} rsa_conv_params;')]
    body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
    fields = re.findall(r'(?:const\s+)?(?:int32_t|int64_t|float|void\s*\*|float\s*\*)\s*\*?\s*([\w, ]+);', body)
    names = [n.strip() for grp in fields for n in grp.split(',')]
    assert names == [f[0] for f in L.ConvParams._fields_]


def test_cout_tile_choice():
    assert [L.cout_tiles(c) for c in (3, 16, 17, 32, 48, 64)] == [1, 1, 2, 2, 3, 4]
    assert L.cout_tiles(240) == 3 and L.cout_tiles(720) == 3 and L.cout_tiles(480) == 3 and L.cout_tiles(128) == 4


def test_argument_validation_without_gpu():
    lib = L.load()
    p = L.ConvParams()
    assert lib.rsa_conv2d(ctypes.byref(p), None) == -1  # RSA_E_ARG: bad geometry
    assert b'geometry' in lib.rsa_last_error_string()
    p.batch, p.H, p.W, p.cin_planes, p.cout, p.ksize, p.products = 1, 8, 8, 1, 8, 5, 3
    assert lib.rsa_conv2d(ctypes.byref(p), None) == -2 and b'ksize' in lib.rsa_last_error_string()
    p.ksize = 3
    assert lib.rsa_conv2d(ctypes.byref(p), None) == -1 and b'null' in lib.rsa_last_error_string()
    p.in_hi, p.in_lo, p.w_packed = 8, 16, 16
    assert lib.rsa_conv2d(ctypes.byref(p), None) == -3  # misaligned pointer
    assert lib.rsa_conv2d_list(None, 1, None) == -1


def test_layout_reference_conversions_roundtrip():
    x = torch.randn(2, 13, 5, 7)
    pl = tensors.nchw_to_planes(x)
    assert pl.hi.shape == (2, 2, 5, 7, 8) and pl.plane_stride == 35 and pl.batch_stride == 70
    back = tensors.planes_to_nchw(pl, 13)
    assert (back - x).abs().max() <= 2.0 ** -16 * x.abs().max()
    assert tensors.planes_to_nchw(pl, 16)[:, 13:].abs().max() == 0
    m = tensors.nchw_to_f32map(x)
    assert m.shape == (2, 4, 5, 7, 4) and torch.equal(tensors.f32map_to_nchw(m, 13), x)


def test_pair_layout_matches_index_formula():
    g = torch.Generator().manual_seed(5)
    cout, cin, planes = 27, 61, 8
    w = torch.randn((cout, cin, 3, 3), generator=g)
    blob = pack.pack_conv_weights_pairs(w, planes)
    assert tuple(blob.shape) == (2, 9, 2, 2, 64, 8)
    hi = _bf16(w)
    lo = _bf16(w - hi.float())
    seen = set()
    rng = torch.Generator().manual_seed(2)
    for _ in range(600):
        q, s, ct, hl, lane, j = (int(torch.randint(0, n, (1,), generator=rng)) for n in (2, 9, 2, 2, 64, 8))
        pl, ky, kx = pack.pair_layout_index(s, lane >> 4)
        co, ci = 16 * ct + (lane & 15), 32 * q + 8 * pl + j
        want = (hi, lo)[hl][co, ci, ky, kx].item() if co < cout and ci < cin else 0.0
        assert blob[q, s, ct, hl, lane, j].item() == want
    # the nine steps cover every (plane, tap) pair of a chunk exactly once
    for s in range(9):
        for lg in range(4):
            seen.add(pack.pair_layout_index(s, lg))
    assert len(seen) == 36 and seen == {(pl, ky, kx) for pl in range(4) for ky in range(3) for kx in range(3)}


def test_halfpair_layout_matches_index_formula():
    """Layout 2 (half mode of the ring schedule): five K steps per 16-channel half chunk, the fifth carrying tap (2,2) in lane groups
    0-1 and zero weights in lane groups 2-3; every (plane, tap) of a half chunk exactly once."""
    g = torch.Generator().manual_seed(6)
    cout, cin, planes = 43, 45, 6
    w = torch.randn((cout, cin, 3, 3), generator=g)
    blob = pack.pack_conv_weights_halfpairs(w, planes)
    assert blob.numel() == int(torch.tensor(pack.packed_weight_shape(cout, planes, 3, 3)).prod())
    body = blob[: 3 * 5 * 3 * 2 * 64 * 8].reshape(3, 5, 3, 2, 64, 8)
    assert blob[body.numel() :].abs().max() == 0
    hi = _bf16(w)
    lo = _bf16(w - hi.float())
    seen = []
    for s in range(5):
        for lg in range(4):
            h = lg >> 1
            if s == 4 and h:
                assert body[:, 4, :, :, lg * 16 : lg * 16 + 16].abs().max() == 0
                continue
            ky, kx = (s, h) if s < 3 else ((h, 2) if s == 3 else (2, 2))
            seen.append((lg & 1, ky, kx))
            for half in range(3):
                for ct in range(3):
                    for li in (0, 7, 15):
                        co = 16 * ct + li
                        for j in (0, 5):
                            ci = 16 * half + 8 * (lg & 1) + j
                            for hl, src in enumerate((hi, lo)):
                                want = src[co, ci, ky, kx].item() if co < cout and ci < cin else 0.0
                                assert body[half, s, ct, hl, lg * 16 + li, j].item() == want
    assert len(seen) == 18 and set(seen) == {(pl, ky, kx) for pl in range(2) for ky in range(3) for kx in range(3)}
    with pytest.raises(ValueError):
        pack.pack_conv_weights_pairs(w, planes)  # whole chunks only


def test_upphase_layout_is_the_upsampled_convolution():
    """Layout 3: the four 2x2 phase kernels (taps pre-summed) reproduce nearest x2 upsampling + 3x3 convolution exactly in f32, and the
    packed fragments hold those sums in the documented order."""
    import torch.nn.functional as F

    g = torch.Generator().manual_seed(9)
    w = torch.randn((64, 64, 3, 3), generator=g) * 0.05
    x = torch.randn((1, 64, 5, 7), generator=g)
    ref = F.conv2d(F.interpolate(x.double(), scale_factor=2, mode='nearest'), w.double(), padding=1)
    taps = {0: ((0,), (1, 2)), 1: ((0, 1), (2,))}
    out = torch.zeros_like(ref)
    xp = F.pad(x.double(), (1, 1, 1, 1))
    for py in range(2):
        for px in range(2):
            k2 = torch.zeros((64, 64, 2, 2), dtype=torch.float64)
            for s in range(2):
                for h in range(2):
                    for ky in taps[py][s]:
                        for kx in taps[px][h]:
                            k2[:, :, s, h] += w[:, :, ky, kx].double()
            # source rows y - 1 + py + s  ->  on the padded map rows y + py + s
            out[:, :, py::2, px::2] = F.conv2d(xp[:, :, py : py + 6, px : px + 8], k2)
    assert (out - ref).abs().max().item() <= 1e-12
    blob = pack.pack_conv_weights_upphase(w).reshape(4, 4, 2, 4, 2, 64, 8)
    for phase, half, s, ct, lane, j in [(0, 0, 0, 0, 0, 0), (1, 2, 1, 3, 37, 5), (2, 3, 0, 1, 63, 7), (3, 1, 1, 2, 20, 3)]:
        lg, li = lane >> 4, lane & 15
        py, px, h = phase >> 1, phase & 1, lg >> 1
        co, ci = 16 * ct + li, 16 * half + 8 * (lg & 1) + j
        want = sum(w[co, ci, ky, kx] for ky in taps[py][s] for kx in taps[px][h])
        hi = _bf16(torch.tensor(float(want)))
        assert blob[phase, half, s, ct, 0, lane, j].item() == hi.item()
        assert blob[phase, half, s, ct, 1, lane, j].item() == _bf16(torch.tensor(float(want)) - hi.float()).item()


def test_plan_alternates_the_tile_order_of_consecutive_layers():
    """Serpentine layer order: every other convolution of a plan carries ``tile_order = 1``; a builder can switch it off."""
    from resselt_amd.engine import base

    plan = base.Plan('cpu')
    orders = []
    for _ in range(5):
        p = L.ConvParams()
        p.batch, p.H, p.W, p.cin_planes, p.cout, p.products = 1, 16, 32, 8, 32, 1
        orders.append(plan.conv(p).tile_order)
    assert orders == ([0, 1, 0, 1, 0] if base._SERPENTINE else [0] * 5)
    plan2 = base.Plan('cpu')
    plan2.serpentine = False
    assert [plan2.conv(L.ConvParams()).tile_order for _ in range(3)] == [0, 0, 0]
