"""Weight re-layout done once at ``load_state_dict`` time (reference: registry.py:113 hands the
checkpoint tensors to ``nn.Module.load_state_dict``; the engine models intercept that call and pack).

Blob layout consumed by ``conv_kernel`` (resselt_amd/csrc/conv_mfma.hip)::

    packed[q][t][ct][hl][lane][j]   bf16
      q   : K chunk of 4 planes = 32 input channels
      t   : tap ky*k + kx
      ct  : tile of 16 output channels
      hl  : 0 = hi (bf16 RNE of w), 1 = lo (bf16 of w - hi)   (only hl=0 when products == 1)
      lane: MFMA A-fragment lane l -> cout = 16*ct + (l & 15), cin = 32*q + 8*(l >> 4) + j

Pure tensor ops: runs on whatever device ``w`` lives on and is unit-tested on CPU against the
index formula above (tests/test_pack.py).
"""

from __future__ import annotations

import torch


def split_bf16(x: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    """x (f32) -> (hi, lo) bf16 with hi = RNE(x), lo = RNE(x - hi)."""
    x = x.to(torch.float32)
    hi = x.to(torch.bfloat16)
    lo = (x - hi.to(torch.float32)).to(torch.bfloat16)
    return hi, lo


def packed_weight_shape(cout: int, cin_planes: int, ksize: int, products: int) -> tuple[int, ...]:
    ct = (cout + 15) // 16
    q = (cin_planes + 3) // 4
    return (q, ksize * ksize, ct, 2 if products == 3 else 1, 64, 8)


def pack_conv_weights(w: torch.Tensor, cin_planes: int, products: int = 3) -> torch.Tensor:
    """OIHW f32 weights -> MFMA A-fragment blob (bf16, contiguous)."""
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
    hi, lo = split_bf16(wp)

    def frag(x: torch.Tensor) -> torch.Tensor:
        # [ct, i, q, g, j, t] -> [q, t, ct, g, i, j] -> lanes g*16+i
        x = x.reshape(ct, 16, q, 4, 8, k * k).permute(2, 5, 0, 3, 1, 4)
        return x.reshape(q, k * k, ct, 64, 8)

    parts = [frag(hi)] + ([frag(lo)] if products == 3 else [])
    return torch.stack(parts, dim=3).contiguous()


def pad_bias(b: torch.Tensor | None, cout: int, device) -> torch.Tensor:
    """f32 bias padded to a multiple of 16 (zeros) so the kernel may read whole tiles."""
    out = torch.zeros(((cout + 15) // 16) * 16, dtype=torch.float32, device=device)
    if b is not None:
        out[:cout] = b.to(torch.float32)
    return out
