"""Python-side builders for the C-ABI launch descriptors (no arithmetic happens here)."""

from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import torch

from . import lib as L
from .pack import pack_conv_weights, pad_bias  # noqa: F401  (pad_bias re-exported for the model files)
from .tensors import PF_BF16, PF_F16, Planes  # noqa: F401

_TORCH_TO_RSA = {torch.float32: L.F32, torch.float16: L.F16, torch.bfloat16: L.BF16, torch.uint8: L.U8}


def rsa_dtype(dt: torch.dtype) -> int:
    try:
        return _TORCH_TO_RSA[dt]
    except KeyError:
        raise TypeError(f'resselt_amd supports float32/float16/bfloat16 tensors (and uint8 [N, H, W, C] images), got {dt}') from None


def current_stream_ptr(device) -> int:
    return int(torch.cuda.current_stream(device).cuda_stream)


def require_cuda(x: torch.Tensor, what: str) -> None:
    """The engine has no CPU path: fail loudly instead of silently computing elsewhere."""
    if not x.is_cuda:
        raise RuntimeError(
            f'{what}: resselt_amd runs on MI355X HIP kernels only; got a tensor on {x.device}. '
            'Move the model and input to a cuda (ROCm) device.'
        )


def pack_weights_device(w: torch.Tensor, cin_planes: int, products: int, layout: int, fmt: int = PF_BF16) -> torch.Tensor:
    """OIHW f32 weights on the GPU -> packed A-fragment blob (``rsa_pack_weights``, one kernel; csrc/pack.hip).  The blob is returned as
    a bf16-typed tensor whatever ``fmt`` says: it is an opaque 16-bit container."""
    require_cuda(w, 'pack_weights')
    w = w.to(torch.float32).contiguous()
    cout, cin, k, _ = w.shape
    lib = L.load()
    nbytes = int(lib.rsa_packed_weight_bytes_layout(cout, cin_planes, k, products, layout))
    if nbytes <= 0:
        raise ValueError(f'unsupported convolution shape cout={cout} cin_planes={cin_planes} k={k} products={products}')
    out = torch.empty(nbytes // 2, dtype=torch.bfloat16, device=w.device)
    with torch.cuda.device(w.device):
        L.check(lib.rsa_pack_weights(w.data_ptr(), cout, cin, cin_planes, k, products, layout, fmt, out.data_ptr(),
                                     C.c_void_p(current_stream_ptr(w.device))), 'rsa_pack_weights')  # fmt: skip
    return out


@dataclass
class ConvWeights:
    """One convolution's device-resident parameters.  ``packed`` is the blob in layout 0 (tap-major K order), or None when the
    weights are kept as f32 OIHW (``w``) and packed on demand in the layout the descriptor's schedule reads (``packed_for``)."""

    packed: torch.Tensor | None  # bf16 blob in layout 0, see pack.py / csrc/pack.hip
    bias: torch.Tensor  # f32, padded to 16
    cout: int
    cin: int
    cin_planes: int
    ksize: int
    products: int
    w: torch.Tensor | None = None  # f32 OIHW source on the device
    _by_layout: dict | None = None
    fmt: int = PF_BF16  # enum rsa_plane_fmt of the blob = of the input planes this layer multiplies

    @staticmethod
    def from_oihw(w: torch.Tensor, b: torch.Tensor | None, products: int, cin_planes: int | None = None, device=None, fmt: int | None = None) -> 'ConvWeights':
        """``products`` may be the module's ``Prec`` (an int that also names the plane format): ``fmt`` then defaults to it."""
        device = device if device is not None else w.device
        if fmt is None:
            fmt = getattr(products, 'fmt', PF_BF16)
        products = int(products)
        cout, cin, k, _ = w.shape
        if cin_planes is None:
            cin_planes = (cin + 7) // 8
        if cin > 8 * cin_planes:
            raise ValueError(f'cin={cin} does not fit in {cin_planes} planes')
        bias = pad_bias(None if b is None else b.to(device), cout, device)
        return ConvWeights(None, bias, cout, cin, cin_planes, k, products, w=w.detach().to(device=device, dtype=torch.float32).contiguous(), _by_layout={},
                           fmt=fmt)  # fmt: skip

    def packed_for(self, layout: int) -> torch.Tensor:
        if self.w is None:  # a blob written directly by a kernel (e.g. rsa_channel_attention_weights): layout 0 only
            if layout != 0:
                raise ValueError('pre-packed weights exist in layout 0 only')
            return self.packed
        blob = self._by_layout.get(layout)
        if blob is None:
            blob = self._by_layout[layout] = pack_weights_device(self.w, self.cin_planes, self.products, layout, self.fmt)
            if layout == 0:
                self.packed = blob
        return blob


def conv_params(
    wts: ConvWeights,
    x: Planes,
    H: int,
    W: int,
    *,
    in_plane0: int = 0,
    cin_planes: int | None = None,
    upsample2x: bool = False,
    act: int = L.ACT_NONE,
    act_param: float = 0.0,
    res1: 'torch.Tensor | tuple[Planes, int] | None' = None,
    alpha: float = 1.0,
    res2: 'torch.Tensor | tuple[Planes, int] | None' = None,
    beta: float = 1.0,
    out: Planes | None = None,
    out_plane_off: int = 0,
    out_f32: torch.Tensor | None = None,
    out_nchw: torch.Tensor | None = None,
    pixel_shuffle: int = 1,
    out_scale: float = 1.0,
    out_shift: torch.Tensor | None = None,
    act_vec: torch.Tensor | None = None,
    out_base: torch.Tensor | None = None,
    out_base_div: int = 0,
    out_lo8: bool = False,
) -> L.ConvParams:
    """Fill one ``rsa_conv_params``. ``H``, ``W`` are the OUTPUT size of the convolution."""
    p = L.ConvParams()
    p.batch = x.n
    p.H, p.W = H, W
    p.ksize = wts.ksize
    p.upsample2x = 1 if upsample2x else 0
    p.cin_planes = wts.cin_planes if cin_planes is None else cin_planes
    if p.cin_planes != wts.cin_planes:
        raise ValueError('cin_planes does not match the packed weights')
    p.cout = wts.cout
    p.products = wts.products
    exp_h, exp_w = (H // 2, W // 2) if upsample2x else (H, W)
    if (x.h, x.w) != (exp_h, exp_w):
        raise ValueError(f'input planes are {x.h}x{x.w}, expected {exp_h}x{exp_w}')
    if in_plane0 + p.cin_planes > x.planes:
        raise ValueError('input plane range exceeds the buffer')
    p.in_hi = x.hi_ptr(in_plane0)
    if x.fmt != wts.fmt:
        raise ValueError(f'input planes are in plane format {x.fmt} but the weights were packed for {wts.fmt}')
    p.in_fmt = wts.fmt
    if wts.products == 3:
        if not x.has_lo(in_plane0, p.cin_planes):
            raise ValueError('products=3 needs lo planes for every input plane')
        p.in_lo = x.lo_ptr(in_plane0)
    else:
        p.in_lo = None
    p.in_plane_stride = x.plane_stride
    p.in_batch_stride = x.batch_stride
    p.bias = wts.bias.data_ptr()
    p.act = act
    p.act_param = act_param
    p.alpha = alpha
    p.beta = beta
    p4 = (wts.cout + 3) // 4
    # a residual is an f32 map, or split planes given as (Planes, first plane): value = hi + lo
    # (a plane residual may carry a third element 'lo8': its lo halves are read from the buffer's 8-bit lo planes, rsa_conv_params.lo8_flags)
    plane_res = {}
    lo8_res = {}
    for name, r in (('res1', res1), ('res2', res2)):
        if isinstance(r, tuple):
            if len(r) == 3:
                if r[2] != 'lo8':
                    raise ValueError(f'{name}: the third element of a plane residual must be "lo8"')
                lo8_res[name] = True
                r = r[:2]
            pl, plane0 = r
            if (pl.n, pl.h, pl.w) != (x.n, H, W) or plane0 + (wts.cout + 7) // 8 > pl.planes or wts.cout % 8:
                raise ValueError(f'{name}: plane residual does not match the convolution output')
            plane_res[name] = (pl, plane0)
    if len({(pl.plane_stride, pl.batch_stride) for pl, _ in plane_res.values()}) > 1:
        raise ValueError('res1 and res2 plane residuals must share their strides')
    for name, r in (('res1', res1), ('res2', res2), ('out_f32', out_f32)):
        if r is not None and name not in plane_res:
            if tuple(r.shape) != (x.n, p4, H, W, 4) or r.dtype != torch.float32 or not r.is_contiguous():
                raise ValueError(f'{name} must be a contiguous f32 [N,{p4},{H},{W},4] map, got {tuple(r.shape)} {r.dtype}')
    p.res1 = None if res1 is None or 'res1' in plane_res else res1.data_ptr()
    p.res2 = None if res2 is None or 'res2' in plane_res else res2.data_ptr()
    if len({pl.fmt for pl, _ in plane_res.values()}) > 1:
        raise ValueError('res1 and res2 plane residuals must share their plane format')
    lo8_strides = set()
    for name, (pl, plane0) in plane_res.items():
        setattr(p, f'{name}_hi', pl.hi_ptr(plane0))
        if lo8_res.get(name):
            if pl.fmt != PF_F16 or not pl.has_lo8(plane0, (wts.cout + 7) // 8):
                raise ValueError(f'{name}: lo8 needs fp16 planes with an 8-bit lo buffer over the residual planes')
            setattr(p, f'{name}_lo', pl.lo8_ptr(plane0))
            p.lo8_flags |= L.LO8_RES1 if name == 'res1' else L.LO8_RES2
            lo8_strides.add(pl.lo8_batch_stride)
        else:
            setattr(p, f'{name}_lo', pl.lo_ptr(plane0) if pl.has_lo(plane0, (wts.cout + 7) // 8) else None)
        p.res_plane_stride, p.res_batch_stride = pl.plane_stride, pl.batch_stride
        p.res_fmt = pl.fmt
    p.out_f32 = None if out_f32 is None else out_f32.data_ptr()
    nplanes_out = (wts.cout + 7) // 8
    if out is not None:
        if (out.h, out.w, out.n) != (H, W, x.n) or out_plane_off + nplanes_out > out.planes:
            raise ValueError('output planes do not match the convolution output')
        p.out_hi = out.hi_ptr()
        # lo halves are written only where the buffer keeps them for EVERY plane this layer writes (see tensors.Planes.empty: lo_planes)
        if out_lo8:
            if out.fmt != PF_F16 or not out.has_lo8(out_plane_off, nplanes_out):
                raise ValueError('out_lo8 needs fp16 output planes with an 8-bit lo buffer over the written planes')
            p.out_lo = out.lo8_ptr()
            p.lo8_flags |= L.LO8_OUT
            lo8_strides.add(out.lo8_batch_stride)
        else:
            p.out_lo = out.lo_ptr() if out.has_lo(out_plane_off, nplanes_out) else None
        p.out_fmt = out.fmt
        p.out_plane_off = out_plane_off
        p.out_plane_stride = out.plane_stride
        p.out_batch_stride = out.batch_stride
    if out_lo8 and out is None:
        raise ValueError('out_lo8 without output planes')
    if p.lo8_flags:
        if len(lo8_strides) != 1:
            raise ValueError('the lo8 operands of a launch must share their batch stride')
        p.lo8_batch_stride = lo8_strides.pop()
    p.pixel_shuffle = pixel_shuffle
    p.out_scale = out_scale
    if out_nchw is not None:
        r = max(pixel_shuffle, 1)
        exp = (x.n, wts.cout // (r * r), H * r, W * r)
        if out_nchw.dtype == torch.uint8:  # 8-bit image, channel-interleaved
            exp = (x.n, H * r, W * r, wts.cout // (r * r))
        if tuple(out_nchw.shape) != exp or not out_nchw.is_contiguous():
            raise ValueError(f'out_nchw must be contiguous {exp}, got {tuple(out_nchw.shape)}')
        p.out_nchw = out_nchw.data_ptr()
        p.out_dtype = rsa_dtype(out_nchw.dtype)
        p.out_shift = None if out_shift is None else out_shift.data_ptr()
        if out_base is not None:
            if out_base.dim() != 4 or tuple(out_base.shape[:2]) != (x.n, exp[1]) or out_base.dtype != out_nchw.dtype or not out_base.is_contiguous():
                raise ValueError('out_base must be a contiguous [N, C_out, h, w] tensor of the output dtype')
            if out_base_div == 0 and tuple(out_base.shape[2:]) != (H, W):
                raise ValueError('out_base must be H x W unless out_base_div gives the nearest-upsampling factor')
            p.out_base = out_base.data_ptr()
            p.out_base_div, p.out_base_h, p.out_base_w = out_base_div, out_base.shape[2], out_base.shape[3]
    if act == L.ACT_PRELU:
        if act_vec is None or act_vec.numel() < ((wts.cout + 15) // 16) * 16 or act_vec.dtype != torch.float32:
            raise ValueError('PReLU needs f32 slopes padded to a multiple of 16')
        p.act_vec = act_vec.data_ptr()
    # the schedule this descriptor dispatches to decides the K order of the weight blob
    p.w_layout = int(L.load().rsa_conv_weight_layout(C.byref(p)))
    p.w_packed = wts.packed_for(p.w_layout).data_ptr()
    p.true_cin = min(wts.cin, 8 * p.cin_planes)  # python-side only (not part of the C struct): FLOP accounting in bench.py
    return p


def run_convs(params: list[L.ConvParams], device) -> None:
    L.conv2d_list(params, current_stream_ptr(device))


def nchw_to_planes(x: torch.Tensor, out: Planes, mean: torch.Tensor | None = None, scale: float = 1.0, unshuffle: int = 1) -> None:  # noqa: C901
    """Plain [N,C,h,w] tensor -> split planes on the GPU (rsa_nchw_to_planes).

    ``out`` may be larger than ``x`` (up to 2x-1): the extra rows/columns are reflect-padded (SwinIR window padding).
    ``unshuffle=r`` fuses ``pixel_unshuffle(x, r)``: ``out`` is then the (H/r x W/r) grid with C*r*r channels.
    """
    require_cuda(x, 'nchw_to_planes')
    if not x.is_contiguous():
        x = x.contiguous()
    if x.dtype == torch.uint8:  # 8-bit image [N, H, W, C]: v = byte / 255 happens in the same kernel
        n, h, w, c = x.shape
    else:
        n, c, h, w = x.shape
    r = unshuffle
    if out.n != n or out.h * r < h or out.w * r < w or out.planes < (c * r * r + 7) // 8:
        raise ValueError('output planes do not match the input tensor')
    lib = L.load()
    L.check(
        lib.rsa_nchw_to_planes(
            x.data_ptr(), rsa_dtype(x.dtype), n, c, out.h, out.w, h, w, r, None if mean is None else mean.data_ptr(), scale,
            out.hi_ptr(), out.lo_ptr() if out.has_lo(0, (c * r * r + 7) // 8) else None, out.plane_stride, out.batch_stride, out.fmt,
            C.c_void_p(current_stream_ptr(x.device)),
        ),
        'rsa_nchw_to_planes',
    )  # fmt: skip


def planes_to_nchw(p: Planes, channels: int) -> torch.Tensor:
    """Split planes -> f32 [N,C,H,W] on the GPU (rsa_planes_to_nchw); for parity checks of intermediates."""
    out = torch.empty((p.n, channels, p.h, p.w), dtype=torch.float32, device=p.hi.device)
    lib = L.load()
    L.check(
        lib.rsa_planes_to_nchw(
            p.hi_ptr(), p.lo_ptr() if p.has_lo(0, (channels + 7) // 8) else None, p.plane_stride, p.batch_stride, p.n, channels, p.h, p.w, p.fmt,
            out.data_ptr(), C.c_void_p(current_stream_ptr(out.device)),
        ),
        'rsa_planes_to_nchw',
    )  # fmt: skip
    return out


def image_u8_to_nchw(img: torch.Tensor, dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """uint8 [N, H, W, C] image batch (or [H, W, C]) on the GPU -> float [N, C, H, W] in [0, 1] (rsa_image_u8_to_nchw)."""
    require_cuda(img, 'image_u8_to_nchw')
    if img.dtype != torch.uint8 or img.dim() not in (3, 4):
        raise TypeError(f'expected a uint8 [N, H, W, C] or [H, W, C] image, got {img.dtype} {tuple(img.shape)}')
    if img.dim() == 3:
        img = img.unsqueeze(0)
    img = img.contiguous()
    n, h, w, c = img.shape
    out = torch.empty((n, c, h, w), dtype=dtype, device=img.device)
    L.check(L.load().rsa_image_u8_to_nchw(img.data_ptr(), n, h, w, c, out.data_ptr(), rsa_dtype(dtype), C.c_void_p(current_stream_ptr(img.device))),
            'rsa_image_u8_to_nchw')  # fmt: skip
    return out


def nchw_to_image_u8(x: torch.Tensor) -> torch.Tensor:
    """float [N, C, H, W] -> uint8 [N, H, W, C], ``(x.clamp(0, 1) * 255).round()`` (rsa_nchw_to_image_u8)."""
    require_cuda(x, 'nchw_to_image_u8')
    if x.dim() != 4:
        raise ValueError(f'expected a [N, C, H, W] tensor, got shape {tuple(x.shape)}')
    x = x.contiguous()
    n, c, h, w = x.shape
    out = torch.empty((n, h, w, c), dtype=torch.uint8, device=x.device)
    L.check(L.load().rsa_nchw_to_image_u8(x.data_ptr(), rsa_dtype(x.dtype), n, c, h, w, out.data_ptr(), C.c_void_p(current_stream_ptr(x.device))),
            'rsa_nchw_to_image_u8')  # fmt: skip
    return out
