"""Device buffers in the engine's HBM layouts (include/resselt_amd.h, "Activation storage").

``Planes``   split-bf16 feature map  hi/lo[N][P][H][W][8]   (P planes of 8 channels, 16-byte units)
``f32 map``  residual stream         f32[N][P4][H][W][4]

Both are ordinary torch tensors so the PyTorch caching allocator owns the memory; the kernels
only ever see raw pointers and strides (in 16-byte units).
"""

from __future__ import annotations

from dataclasses import dataclass

import torch


@dataclass
class Planes:
    hi: torch.Tensor  # [N, P, H, W, 8] bf16
    lo: torch.Tensor | None  # same shape, or None in single-product (plain bf16) mode

    @staticmethod
    def empty(n: int, planes: int, h: int, w: int, device, with_lo: bool = True) -> 'Planes':
        hi = torch.empty((n, planes, h, w, 8), dtype=torch.bfloat16, device=device)
        lo = torch.empty_like(hi) if with_lo else None
        return Planes(hi, lo)

    @property
    def n(self) -> int:
        return self.hi.shape[0]

    @property
    def planes(self) -> int:
        return self.hi.shape[1]

    @property
    def h(self) -> int:
        return self.hi.shape[2]

    @property
    def w(self) -> int:
        return self.hi.shape[3]

    @property
    def plane_stride(self) -> int:  # units of 16 bytes
        return self.h * self.w

    @property
    def batch_stride(self) -> int:
        return self.planes * self.h * self.w

    def hi_ptr(self, plane: int = 0) -> int:
        return self.hi.data_ptr() + plane * self.plane_stride * 16

    def lo_ptr(self, plane: int = 0) -> int | None:
        if self.lo is None:
            return None
        return self.lo.data_ptr() + plane * self.plane_stride * 16


class PlaneRows:
    """Rows [r0, r1) of a ``Planes`` buffer as a convolution operand: same storage and strides, first-row pointer offset.
    Lets a plan run part of a network band by band (bounded buffers at 2x / 4x resolution) without copying."""

    def __init__(self, base: Planes, r0: int, r1: int):
        if not 0 <= r0 < r1 <= base.h:
            raise ValueError(f'rows [{r0}, {r1}) outside a map of {base.h} rows')
        self.base, self.r0, self.r1 = base, r0, r1
        self.hi, self.lo = base.hi, base.lo  # (the whole tensors: kept alive by whoever holds the view)

    n = property(lambda self: self.base.n)
    planes = property(lambda self: self.base.planes)
    w = property(lambda self: self.base.w)
    h = property(lambda self: self.r1 - self.r0)
    plane_stride = property(lambda self: self.base.plane_stride)
    batch_stride = property(lambda self: self.base.batch_stride)

    def hi_ptr(self, plane: int = 0) -> int:
        return self.base.hi_ptr(plane) + self.r0 * self.base.w * 16

    def lo_ptr(self, plane: int = 0) -> int | None:
        p = self.base.lo_ptr(plane)
        return None if p is None else p + self.r0 * self.base.w * 16


def empty_f32map(n: int, channels: int, h: int, w: int, device) -> torch.Tensor:
    return torch.empty((n, (channels + 3) // 4, h, w, 4), dtype=torch.float32, device=device)


# ---- reference conversions (torch ops; used by tests and debugging, not by the forward path) ----


def planes_to_nchw(p: Planes, channels: int) -> torch.Tensor:
    v = p.hi.to(torch.float32)
    if p.lo is not None:
        v = v + p.lo.to(torch.float32)
    n, pl, h, w, _ = v.shape
    return v.permute(0, 1, 4, 2, 3).reshape(n, pl * 8, h, w)[:, :channels].contiguous()


def nchw_to_planes(x: torch.Tensor, with_lo: bool = True) -> Planes:
    from .pack import split_bf16

    n, c, h, w = x.shape
    pl = (c + 7) // 8
    xp = torch.zeros((n, pl * 8, h, w), dtype=torch.float32, device=x.device)
    xp[:, :c] = x.to(torch.float32)
    xp = xp.reshape(n, pl, 8, h, w).permute(0, 1, 3, 4, 2).contiguous()
    hi, lo = split_bf16(xp)
    return Planes(hi.contiguous(), lo.contiguous() if with_lo else None)


def f32map_to_nchw(m: torch.Tensor, channels: int) -> torch.Tensor:
    n, p4, h, w, _ = m.shape
    return m.permute(0, 1, 4, 2, 3).reshape(n, p4 * 4, h, w)[:, :channels].contiguous()


def nchw_to_f32map(x: torch.Tensor) -> torch.Tensor:
    n, c, h, w = x.shape
    p4 = (c + 3) // 4
    xp = torch.zeros((n, p4 * 4, h, w), dtype=torch.float32, device=x.device)
    xp[:, :c] = x.to(torch.float32)
    return xp.reshape(n, p4, 4, h, w).permute(0, 1, 3, 4, 2).contiguous()
