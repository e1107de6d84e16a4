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
    # Round 4: the lo halves of the first planes as 8-bit codes (`lo8_encode`), [N, lo8_planes, H, W, 8] uint8 = 8-byte units: the residual stream of
    # RRDBNet's dense blocks under 'mixed' (3 bytes per channel; rsa_conv_params.lo8_flags).  Independent of ``lo``: a buffer may carry both (the
    # first and the last block of a trunk exchange fp16 lo halves with the three-product layers around it).
    lo8: torch.Tensor | None = None

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
        return (self.hi.shape[1] + self.lo_planes) * self.hi.shape[0] * self.h * self.w * 16 + (0 if self.lo8 is None else self.lo8.numel())

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

    def with_lo8(self, planes: int) -> 'Planes':
        """Attach an 8-bit lo buffer for the first ``planes`` planes (see ``lo8``)."""
        self.lo8 = torch.empty((self.n, planes, self.h, self.w, 8), dtype=torch.uint8, device=self.hi.device)
        return self

    def has_lo8(self, plane0: int, nplanes: int) -> bool:
        return self.lo8 is not None and plane0 + nplanes <= self.lo8.shape[1]

    def lo8_ptr(self, plane: int = 0) -> int:
        return self.lo8.data_ptr() + plane * self.plane_stride * 8

    @property
    def lo8_batch_stride(self) -> int:  # units of 8 bytes
        return self.lo8.stride(0) // 8


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


def lo8_encode(v: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    """f32 -> (fp16 hi, 8-bit code of v - hi): the distance from f32(hi) to v in steps of 32 f32 ulps along the bit patterns, a signed byte
    stored in a uint8 tensor -- `lo8_encode2` of csrc/conv_common.h, operation for operation (the distance saturates to 16 bits, so the code
    is exact whenever hi is a normal fp16 number; with hi = 0 or subnormal it decodes to within 2^-24 of the value)."""
    hi = v.to(torch.float16)
    d = v.contiguous().view(torch.int32) - hi.to(torch.float32).view(torch.int32)  # same sign bit in both: the difference of the magnitudes
    d = (d.clamp(-32768, 32767) + 16).clamp(max=32767)
    return hi, ((d >> 5).clamp(max=127) & 0xFF).to(torch.uint8)


def lo8_decode(hi: torch.Tensor, code: torch.Tensor) -> torch.Tensor:
    """(fp16 hi, code) -> f32, as `lo8_decode` in csrc/conv_common.h."""
    return (hi.to(torch.float32).contiguous().view(torch.int32) + (code.contiguous().view(torch.int8).to(torch.int32) << 5)).view(torch.float32)


def planes_to_nchw(p: Planes, channels: int, lo8: bool = False) -> torch.Tensor:
    """``lo8``: decode the 8-bit lo buffer instead of adding the 16-bit one."""
    v = p.hi.to(torch.float32)
    if lo8:
        v = v.clone()
        k = p.lo8.shape[1]
        v[:, :k] = lo8_decode(p.hi[:, :k], p.lo8)
    elif p.lo is not None:
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
