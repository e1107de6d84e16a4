"""Device buffers in the engine's HBM layouts (include/resselt_amd.h, "Activation storage").

``Planes``   split feature map       hi/lo[N][P][H][W][8]   (P planes of 8 channels, 16-byte units; bf16 or fp16: ``fmt``)
``f32 map``  residual stream         f32[N][P4][H][W][4]

A ``Planes`` buffer may carry ``lo`` for its first ``lo_planes`` planes only (the residual stream of a residual dense block under the
one-product fp16 policy: the 64 trunk channels keep 22 bits, the growth channels 11): hi and lo then live in ONE allocation
``[N][P + lo_planes][H][W][8]`` so that both share the batch stride the kernels apply to either pointer.

Both are ordinary torch tensors so the PyTorch caching allocator owns the memory; the kernels
only ever see raw pointers and strides (in 16-byte units).
"""

from __future__ import annotations

from dataclasses import dataclass

import torch

PF_BF16, PF_F16 = 0, 1  # enum rsa_plane_fmt
_PF_DTYPE = {PF_BF16: torch.bfloat16, PF_F16: torch.float16}


@dataclass
class Planes:
    hi: torch.Tensor  # [N, P, H, W, 8] bf16 or fp16 (dense, or the first P planes of a [N, P + lo_planes, ...] allocation)
    lo: torch.Tensor | None  # [N, lo_planes <= P, H, W, 8] of the same dtype, or None (one-product consumers only)

    @staticmethod
    def empty(n: int, planes: int, h: int, w: int, device, with_lo: bool = True, fmt: int = PF_BF16, lo_planes: int | None = None) -> 'Planes':
        dt = _PF_DTYPE[fmt]
        if not with_lo or lo_planes == 0:
            return Planes(torch.empty((n, planes, h, w, 8), dtype=dt, device=device), None)
        if lo_planes is None or lo_planes >= planes:
            hi = torch.empty((n, planes, h, w, 8), dtype=dt, device=device)
            return Planes(hi, torch.empty_like(hi))
        store = torch.empty((n, planes + lo_planes, h, w, 8), dtype=dt, device=device)
        return Planes(store[:, :planes], store[:, planes:])

    @property
    def fmt(self) -> int:
        return PF_F16 if self.hi.dtype == torch.float16 else PF_BF16

    @property
    def lo_planes(self) -> int:
        return 0 if self.lo is None else self.lo.shape[1]

    def nbytes(self) -> int:
        return (self.hi.shape[1] + self.lo_planes) * self.hi.shape[0] * self.h * self.w * 16

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
    def batch_stride(self) -> int:  # units of 16 bytes (hi and lo share it: see ``empty``)
        return self.hi.stride(0) // 8

    def hi_ptr(self, plane: int = 0) -> int:
        return self.hi.data_ptr() + plane * self.plane_stride * 16

    def lo_ptr(self, plane: int = 0) -> int | None:
        if self.lo is None:
            return None
        return self.lo.data_ptr() + plane * self.plane_stride * 16

    def has_lo(self, plane0: int, nplanes: int) -> bool:
        """Whether planes [plane0, plane0 + nplanes) all carry lo halves."""
        return self.lo is not None and plane0 + nplanes <= self.lo.shape[1]


class PlaneRows:
    """Rows [r0, r1) of a ``Planes`` buffer as a convolution operand: same storage and strides, first-row pointer offset.
    Lets a plan run part of a network band by band (bounded buffers at 2x / 4x resolution) without copying."""

    def __init__(self, base: Planes, r0: int, r1: int):
        if not 0 <= r0 < r1 <= base.h:
            raise ValueError(f'rows [{r0}, {r1}) outside a map of {base.h} rows')
        self.base, self.r0, self.r1 = base, r0, r1
        self.hi, self.lo = base.hi, base.lo  # (the whole tensors: kept alive by whoever holds the view)

    n = property(lambda self: self.base.n)
    fmt = property(lambda self: self.base.fmt)
    lo_planes = property(lambda self: self.base.lo_planes)
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

    def has_lo(self, plane0: int, nplanes: int) -> bool:
        return self.base.has_lo(plane0, nplanes)


def empty_f32map(n: int, channels: int, h: int, w: int, device) -> torch.Tensor:
    return torch.empty((n, (channels + 3) // 4, h, w, 4), dtype=torch.float32, device=device)


# ---- reference conversions (torch ops; used by tests and debugging, not by the forward path) ----


def planes_to_nchw(p: Planes, channels: int) -> torch.Tensor:
    v = p.hi.to(torch.float32)
    if p.lo is not None:
        v = v.clone()
        v[:, : p.lo.shape[1]] += p.lo.to(torch.float32)
    n, pl, h, w, _ = v.shape
    return v.permute(0, 1, 4, 2, 3).reshape(n, pl * 8, h, w)[:, :channels].contiguous()


def nchw_to_planes(x: torch.Tensor, with_lo: bool = True, fmt: int = PF_BF16) -> Planes:
    from .pack import split_halves

    n, c, h, w = x.shape
    pl = (c + 7) // 8
    xp = torch.zeros((n, pl * 8, h, w), dtype=torch.float32, device=x.device)
    xp[:, :c] = x.to(torch.float32)
    xp = xp.reshape(n, pl, 8, h, w).permute(0, 1, 3, 4, 2).contiguous()
    hi, lo = split_halves(xp, _PF_DTYPE[fmt])
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
