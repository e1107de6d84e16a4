"""Weight re-layout done once at ``load_state_dict`` time (reference: registry.py:113 hands the
checkpoint tensors to ``nn.Module.load_state_dict``; the engine models intercept that call and pack).

Blob layout consumed by ``conv_kernel`` (resselt_amd/csrc/conv_mfma.hip)::

    packed[q][t][ct][hl][lane][j]   bf16
      q   : K chunk of 4 planes = 32 input channels
      t   : tap ky*k + kx
      ct  : tile of 16 output channels
      hl  : 0 = hi (bf16 RNE of w), 1 = lo (bf16 of w - hi)   (only hl=0 when products == 1)
      lane: MFMA A-fragment lane l -> cout = 16*ct + (l & 15), cin = 32*q + 8*(l >> 4) + j

The product path packs on the GPU through the C-ABI (``rsa_pack_weights``, csrc/pack.hip; ``ops.pack_weights_device``).
The functions here are the torch restatement of the two blob layouts: host-side documentation of the format, checked on CPU
against the index formulas (tests/test_pack_and_capi.py) and used by the GPU tests as the expected output of ``rsa_pack_weights``.

Layout 1 (tap-pair K order of the ring schedule, csrc/conv_ring.h), 3x3 / three products / whole chunks only::

    packed[q][s][ct][hl][lane][j],  s = 0..8;  lane group lg = lane >> 4, h = lg >> 1, plane-in-half = lg & 1
      half A = planes 4q, 4q+1; half B = planes 4q+2, 4q+3
      s = 0,1,2: half A, tap (ky = s,   kx = h)     s = 3: half A, tap (ky = h, kx = 2)     s = 4: tap (2,2) of half A (h=0) / B (h=1)
      s = 5,6,7: half B, tap (ky = s-5, kx = h)     s = 8: half B, tap (ky = h, kx = 2)
"""

from __future__ import annotations

import torch


def split_halves(x: torch.Tensor, dtype: torch.dtype = torch.bfloat16) -> tuple[torch.Tensor, torch.Tensor]:
    """x (f32) -> (hi, lo) in ``dtype`` (bf16 or fp16) with hi = RNE(x), lo = RNE(x - hi)."""
    x = x.to(torch.float32)
    hi = x.to(dtype)
    lo = (x - hi.to(torch.float32)).to(dtype)
    return hi, lo


def split_bf16(x: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    return split_halves(x, torch.bfloat16)


def packed_weight_shape(cout: int, cin_planes: int, ksize: int, products: int) -> tuple[int, ...]:
    ct = (cout + 15) // 16
    q = (cin_planes + 3) // 4
    return (q, ksize * ksize, ct, 2 if products == 3 else 1, 64, 8)


def pack_conv_weights(w: torch.Tensor, cin_planes: int, products: int = 3, dtype: torch.dtype = torch.bfloat16) -> torch.Tensor:
    """OIHW f32 weights -> MFMA A-fragment blob (``dtype`` = bf16 or fp16, contiguous)."""
    if w.dim() != 4 or w.shape[2] != w.shape[3] or w.shape[2] not in (1, 3):
        raise ValueError(f'expected [cout, cin, k, k] with k in (1, 3), got {tuple(w.shape)}')
    cout, cin, k, _ = w.shape
    if cin > 8 * cin_planes:
        raise ValueError(f'cin={cin} does not fit in {cin_planes} planes')
    if products not in (1, 3):
        raise ValueError('products must be 1 or 3')
    ct = (cout + 15) // 16
    q = (cin_planes + 3) // 4
    wp = torch.zeros((ct * 16, q * 32, k * k), dtype=torch.float32, device=w.device)
    wp[:cout, :cin] = w.to(torch.float32).reshape(cout, cin, k * k)
    hi, lo = split_halves(wp, dtype)

    def frag(x: torch.Tensor) -> torch.Tensor:
        # [ct, i, q, g, j, t] -> [q, t, ct, g, i, j] -> lanes g*16+i
        x = x.reshape(ct, 16, q, 4, 8, k * k).permute(2, 5, 0, 3, 1, 4)
        return x.reshape(q, k * k, ct, 64, 8)

    parts = [frag(hi)] + ([frag(lo)] if products == 3 else [])
    return torch.stack(parts, dim=3).contiguous()


def pair_layout_index(s: int, lg: int) -> tuple[int, int, int]:
    """(plane within the 4-plane chunk, ky, kx) read by lane group ``lg`` in K step ``s`` of the tap-pair layout."""
    h = lg >> 1
    half = 0 if s < 4 else (h if s == 4 else 1)
    ss = s if s < 4 else (4 if s == 4 else s - 5)
    ky, kx = (ss, h) if ss < 3 else ((h, 2) if ss == 3 else (2, 2))
    return 2 * half + (lg & 1), ky, kx


def pack_conv_weights_pairs(w: torch.Tensor, cin_planes: int, products: int = 3, dtype: torch.dtype = torch.bfloat16) -> torch.Tensor:
    """OIHW f32 3x3 weights -> blob in the tap-pair layout (hi and lo for three products, hi only for one)."""
    cout, cin, k, _ = w.shape
    if k != 3 or cin_planes % 4 or cin > 8 * cin_planes:
        raise ValueError('the tap-pair layout needs a 3x3 layer with whole 32-channel chunks')
    ct = (cout + 15) // 16
    q = cin_planes // 4
    wp = torch.zeros((ct * 16, q * 32, 3, 3), dtype=torch.float32, device=w.device)
    wp[:cout, :cin] = w.to(torch.float32)
    hi, lo = split_halves(wp, dtype)
    nhl = 2 if products == 3 else 1
    out = torch.zeros((q, 9, ct, nhl, 64, 8), dtype=dtype, device=w.device)
    for s in range(9):
        for lg in range(4):
            pl, ky, kx = pair_layout_index(s, lg)
            for hl, src in enumerate((hi, lo)[:nhl]):
                # [ct*16, q*32] -> [q, ct, 16 lanes, 8]
                v = src[:, :, ky, kx].reshape(ct, 16, q, 4, 8)[:, :, :, pl, :].permute(2, 0, 1, 3)
                out[:, s, :, hl, lg * 16 : lg * 16 + 16, :] = v
    return out.contiguous()


def pack_conv_weights_halfpairs(w: torch.Tensor, cin_planes: int, products: int = 3, dtype: torch.dtype = torch.bfloat16) -> torch.Tensor:
    """Layout 2 (half mode of the ring schedule: an odd number of 16-channel half chunks): packed[half][s 0..4][ct][hl][lane][j];
    lane group lg reads plane 2*half + (lg & 1); s = 0,1,2: tap (ky = s, kx = h); s = 3: tap (ky = h, kx = 2); s = 4: tap (2,2) for
    h = lg >> 1 == 0 and ZERO weights for h == 1.  The blob is allocated at the size of the chunked layouts; the tail stays zero."""
    cout, cin, k, _ = w.shape
    if k != 3 or cin_planes % 2 or cin > 8 * cin_planes:
        raise ValueError('the half-chunk layout needs a 3x3 layer with an even number of input planes')
    ct = (cout + 15) // 16
    nh = cin_planes // 2
    wp = torch.zeros((ct * 16, nh * 16, 3, 3), dtype=torch.float32, device=w.device)
    wp[:cout, :cin] = w.to(torch.float32)
    hi, lo = split_halves(wp, dtype)
    nhl = 2 if products == 3 else 1
    out = torch.zeros((nh, 5, ct, nhl, 64, 8), dtype=dtype, device=w.device)
    for s in range(5):
        for lg in range(4):
            h = lg >> 1
            if s == 4 and h:
                continue
            ky, kx = (s, h) if s < 3 else ((h, 2) if s == 3 else (2, 2))
            for hl, src in enumerate((hi, lo)[:nhl]):
                v = src[:, :, ky, kx].reshape(ct, 16, nh, 2, 8)[:, :, :, lg & 1, :].permute(2, 0, 1, 3)  # [nh, ct, 16 lanes, 8]
                out[:, s, :, hl, lg * 16 : lg * 16 + 16, :] = v
    full = torch.zeros(packed_weight_shape(cout, cin_planes, 3, products), dtype=dtype, device=w.device).reshape(-1)
    full[: out.numel()] = out.reshape(-1)
    return full


def pack_conv_weights_upphase(w: torch.Tensor) -> torch.Tensor:
    """Layout 3 (csrc/conv_ring_up.h: nearest x2 upsampling + 3x3 as four 2x2 convolutions on the source map, 64 -> 64 channels):
    packed[phase][half][s][ct][hl][lane][j]; phase = 2*py + px (output pixel parity), lane group lg reads plane 2*half + (lg & 1) at
    source row s, source column h = lg >> 1; its weight is the f32 SUM of the 3x3 taps that read that source pixel on the upsampled
    image: rows {0} / {1, 2} for py = 0, {0, 1} / {2} for py = 1; columns likewise."""
    cout, cin, k, _ = w.shape
    if (cout, cin, k) != (64, 64, 3):
        raise ValueError('the upsampling-phase layout is for 64 -> 64 channel 3x3 layers')
    taps = {0: ((0,), (1, 2)), 1: ((0, 1), (2,))}  # parity -> source index -> taps
    w = w.to(torch.float32)
    out = torch.zeros((4, 4, 2, 4, 2, 64, 8), dtype=torch.bfloat16, device=w.device)
    for py in range(2):
        for px in range(2):
            for s in range(2):
                for lg in range(4):
                    h = lg >> 1
                    summed = torch.zeros((cout, cin), dtype=torch.float32, device=w.device)
                    for ky in taps[py][s]:  # (the kernel adds in the same order: ky outer, kx inner)
                        for kx in taps[px][h]:
                            summed = summed + w[:, :, ky, kx]
                    hi, lo = split_bf16(summed)
                    for hl, src in enumerate((hi, lo)):
                        v = src.reshape(4, 16, 4, 2, 8)[:, :, :, lg & 1, :].permute(2, 0, 1, 3)  # [half, ct, 16 lanes, 8]
                        out[2 * py + px, :, s, :, hl, lg * 16 : lg * 16 + 16, :] = v
    return out.reshape(-1)


def pad_bias(b: torch.Tensor | None, cout: int, device) -> torch.Tensor:
    """f32 bias padded to a multiple of 16 (zeros) so the kernel may read whole tiles."""
    out = torch.zeros(((cout + 15) // 16) * 16, dtype=torch.float32, device=device)
    if b is not None:
        out[:cout] = b.to(torch.float32)
    return out
