"""Common machinery of the engine-backed models returned by the loaders.

An ``EngineModule`` is an ``nn.Module`` that *owns parameters under the reference's key names* (so the
registry's strict ``load_state_dict`` and the usual ``.to()/.half()/.state_dict()`` work) but whose
``forward`` is a cached *plan*: device buffers in the engine's layouts plus an array of C-ABI launch
descriptors that ``rsa_conv2d_list`` executes with one host call.
"""

from __future__ import annotations

import ctypes as C
import threading
from typing import Callable

import torch
from torch import nn

from . import lib as L
from . import ops
from .tensors import PF_BF16, PF_F16, Planes, empty_f32map

# precision modes: name -> (products, plane format) of a layer the architecture has no per-layer rule for
#   bf16x3  three bf16 products on split planes (hi + lo): ~16-bit operands, f32 range.  The conservative mode.
#   bf16    one bf16 product (8-bit operands)
#   fp16    one fp16 product on hi planes (11-bit operands, |v| < 65504): a third of the matrix instructions, half the activation bytes
#   mixed   a per-layer table of the architecture (RRDBNet: residual dense blocks in one fp16 product, head / tail in three bf16 products)
# 'auto' (the default) selects ``EngineModule.auto_precision``: the cheapest mode the architecture has pinned at <= 2e-4 max-abs
# against the fp32 oracle in the -m gpu suite.
PRECISIONS = {'bf16x3': (3, PF_BF16), 'bf16': (1, PF_BF16), 'fp16': (1, PF_F16), 'mixed': (3, PF_BF16)}


import os as _os

_SERPENTINE = _os.environ.get('RSA_SERPENTINE', '1') != '0'


class Prec(int):
    """``products`` (1 or 3; this IS the int, so architecture code written against an int keeps working) + plane format + mode name.
    For 'mixed' the pair is the DEFAULT of layers / buffers the architecture's table does not name."""

    def __new__(cls, name: str, table: dict | None = None):
        products, fmt = (table or {}).get(name, PRECISIONS[name])
        obj = super().__new__(cls, products)
        obj.name, obj.fmt = name, fmt
        return obj

    @property
    def with_lo(self) -> bool:
        return int(self) == 3


class Fp16Range(ValueError):
    """A weight does not fit fp16: the module falls back to 'bf16x3' (with a warning)."""


def check_fp16_range(weights) -> None:
    """Pack-time guard of the fp16 modes: raise ``Fp16Range`` when a convolution weight packed in fp16 exceeds the format's range
    (one device synchronisation per pack).  ``weights``: an iterable of ``ops.ConvWeights`` (others are skipped)."""
    ws = [cw.w for cw in weights if isinstance(cw, ops.ConvWeights) and cw.fmt == PF_F16 and cw.w is not None]
    # (one multi-tensor launch instead of two small kernels per weight: 45 ms of a SwinIR-L cold start)
    if ws and float(torch.stack(torch._foreach_norm(ws, float('inf'))).amax()) > 6.0e4:
        raise Fp16Range('a convolution weight exceeds the fp16 range (|w| > 6e4)')


def conv_algorithmic_bytes(p: L.ConvParams) -> int:
    """Bytes one launch must move if every operand is touched exactly once (no halo re-reads, weights excluded):
    input planes (16-byte units of 8 bf16 channels, hi and -- for 3 products -- lo), f32 residual maps, and every output it writes."""
    px_out = p.batch * p.H * p.W
    px_in = px_out // 4 if p.upsample2x else px_out
    total = p.cin_planes * 16 * (2 if p.products == 3 else 1) * px_in
    maps = (p.cout + 3) // 4 * 16 * px_out
    total += maps * sum(1 for r in (p.res1, p.res2, p.out_f32) if r)
    for hi, lo, l8 in ((p.res1_hi, p.res1_lo, p.lo8_flags & L.LO8_RES1), (p.res2_hi, p.res2_lo, p.lo8_flags & L.LO8_RES2)):
        if hi:  # plane residuals: 2 B / channel per 16-bit half, 1 B / channel for an 8-bit lo half
            total += (p.cout + 7) // 8 * (16 + (8 if l8 else 16 if lo else 0)) * px_out
    if p.out_hi:
        total += (p.cout + 7) // 8 * (16 + (8 if p.lo8_flags & L.LO8_OUT else 16 if p.out_lo else 0)) * px_out
    if p.out_nchw:
        esize = {L.F32: 4, L.F16: 2, L.BF16: 2, L.U8: 1}[p.out_dtype]
        total += p.cout * esize * px_out * (2 if p.out_base else 1)
    return total


class Plan:
    """Buffers + launch list for one (batch, H, W, dtype, device, precision) signature."""

    def __init__(self, device, fmt: int = PF_BF16):
        self.device = device
        self.fmt = fmt  # plane format of buffers a builder does not give one for (the module's resolved precision)
        self.keep: list[object] = []  # tensors referenced by raw pointer in the descriptors
        self.steps: list[Callable[[], None]] = []  # executed in order on the current stream
        self._pending: list[L.ConvParams] = []
        self.conv_arrays: list = []  # every flushed descriptor array, in launch order (bench.py replays single entries)
        self.kernel_calls: list = []  # (meta, closure) of the non-convolution launches that carry a price tag (see ``call``)
        self.conv_cin: list = []  # per array: the layers' true input channel counts (a descriptor only knows planes of 8)
        self._pending_cin: list = []
        # Serpentine layer order: every other convolution walks its output tiles bottom-up (``rsa_conv_params.tile_order``), so a layer
        # starts on the rows its producer wrote last, which are still in the 256 MB Infinity Cache (ring schedule only; RSA_SERPENTINE=0
        # switches it off for A/B runs).  Measured on RRDBNet-23 at 1080p: 101.0 -> 97.8 ms per frame (profiles/r03_c_*).
        self.serpentine = True
        self._n_conv = 0

    # ---- buffers ----
    def planes(self, n, planes, h, w, with_lo=True, fmt: int | None = None, lo_planes: int | None = None) -> Planes:
        p = Planes.empty(n, planes, h, w, self.device, with_lo, self.fmt if fmt is None else fmt, lo_planes)
        self.keep.append(p)
        return p

    def f32map(self, n, channels, h, w) -> torch.Tensor:
        t = empty_f32map(n, channels, h, w, self.device)
        self.keep.append(t)
        return t

    # ---- launch list ----
    def conv(self, params: L.ConvParams) -> L.ConvParams:
        if self.serpentine and _SERPENTINE:
            params.tile_order = self._n_conv & 1
        self._n_conv += 1
        self._pending.append(params)
        self._pending_cin.append(int(getattr(params, 'true_cin', params.cin_planes * 8)))
        self._conv_bytes = getattr(self, '_conv_bytes', 0) + conv_algorithmic_bytes(params)
        return params

    def flush(self) -> 'C.Array | None':
        """Close the current run of convolutions into one ``rsa_conv2d_list`` step."""
        if not self._pending:
            return None
        arr = (L.ConvParams * len(self._pending))(*self._pending)
        self._n_launches = getattr(self, '_n_launches', 0) + len(self._pending)
        self._pending = []
        self.conv_arrays.append(arr)
        self.conv_cin.append(self._pending_cin)
        self._pending_cin = []
        dev = self.device

        def step():
            L.conv2d_list(arr, ops.current_stream_ptr(dev))

        step._rsa_conv = (arr, self.conv_cin[-1])  # bench.py replays the list launch by launch (in-frame per-kernel timing)
        self.steps.append(step)
        return arr

    def call(self, fn: Callable[[], None], meta: dict | None = None) -> None:
        """A non-convolution launch.  ``meta`` (kernel name, algorithmic flop and bytes of the launch) lets bench.py replay and price it."""
        self.flush()
        self.steps.append(fn)
        if meta is not None:
            self.kernel_calls.append((meta, fn))
            try:
                fn._rsa_meta = meta
            except AttributeError:  # a callable that takes no attributes (functools.partial does; builtins do not): not priced
                pass

    def run(self) -> None:
        self.flush()
        for step in self.steps:
            step()

    def n_launches(self) -> int:
        return getattr(self, '_n_launches', 0)

    def count_launches(self, k: int = 1) -> None:
        """Book-keeping for kernels launched from ``call`` steps (the convolution list counts its own)."""
        self._n_launches = self.n_launches() + k

    def conv_bytes(self) -> int:
        """Algorithmic HBM bytes of the convolution launches in THIS engine's layouts (see ``conv_algorithmic_bytes``)."""
        return getattr(self, '_conv_bytes', 0)

    def release(self) -> None:
        """Drop every buffer and closure now.  A plan's steps are closures over its own buffers (reference cycles), so a plan that is
        merely dereferenced keeps its gigabytes until the cycle collector runs; the cache calls this when it evicts a plan."""
        self._released_bytes = self.buffer_bytes()
        self.keep.clear()
        self.steps.clear()
        self.conv_arrays.clear()
        self.kernel_calls.clear()
        self.conv_cin.clear()
        self._pending = []
        self._pending_cin = []

    def buffer_bytes(self) -> int:
        total = 0
        for k in self.keep:
            if k is None:
                continue
            if isinstance(k, Planes):
                total += k.nbytes()
            elif isinstance(k, torch.Tensor):
                total += k.numel() * k.element_size()
        return total


class EngineModule(nn.Module):
    """Base of the MI355X models.  Subclasses implement ``_pack`` and ``_build_plan``."""

    def __init__(self):
        super().__init__()
        self.precision: str = 'auto'  # 'auto' | 'bf16x3' | 'bf16' | whatever else ``precisions`` lists (see PRECISIONS)
        # Replay the whole forward as ONE hipGraph (torch.cuda.CUDAGraph) per input signature: removes the per-launch host cost and most of
        # the inter-kernel gaps, which dominate for small images (351 launches of ~10 us each for RRDBNet-23 at 256x256).  Costs one copy
        # of the input into, and of the output out of, graph-owned buffers.  Off by default.
        self.use_graph: bool = False
        self._packed: dict = {}
        self._packed_versions = None
        self._lock = threading.RLock()  # a plan's buffers are module state: one forward at a time per module (the reference nn.Module is re-entrant)
        self._plans: dict = {}
        self._max_plans = 8
        self.max_plan_bytes = 40 << 30  # byte budget of the cached plans' buffers (least recently used plans are dropped first)

    # -- copy / pickle: caches and the lock are per-instance runtime state, never part of the module's value --
    def __getstate__(self):
        state = self.__dict__.copy()
        state['_packed'], state['_plans'], state['_packed_versions'] = {}, {}, None
        state.pop('_lock', None)
        return state

    def __setstate__(self, state):
        super().__setstate__(state)
        self._lock = threading.RLock()

    # -- cache invalidation: anything that can change parameter values, dtype or device --
    def _drop_plan(self, key) -> None:
        entry = self._plans.pop(key)
        entry[0].release()
        entry[1:] = [None, None, None]  # input setter / output getter / graph: closures over the plan's buffers

    def invalidate(self) -> None:
        """Drop the packed weights and every plan.  Called automatically by load_state_dict / .to() / .half() and when a parameter's
        version counter changes; call it yourself after editing parameters through ``.data`` (PyTorch does not version those edits)."""
        self._invalidate()

    def _invalidate(self) -> None:
        self._packed = {}
        self._fp16_refused = False
        for key in list(getattr(self, '_plans', {})):
            self._drop_plan(key)
        self._plans = {}

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        result = super().load_state_dict(self._convert_state_dict(state_dict), strict=strict, assign=assign)
        self._invalidate()
        return result

    def _convert_state_dict(self, state_dict):
        return state_dict

    def _apply(self, fn, recurse=True):
        out = super()._apply(fn, recurse)
        self._invalidate()
        return out

    def launches_per_forward(self) -> int | None:
        """Kernel launches of the most recently built plan (None before the first forward)."""
        if not self._plans:
            return None
        return list(self._plans.values())[-1][0].n_launches()

    def last_plan(self) -> 'Plan | None':
        """The most recently built plan (None before the first forward)."""
        if not self._plans:
            return None
        return list(self._plans.values())[-1][0]

    def conv_bytes_per_forward(self) -> int | None:
        if not self._plans:
            return None
        return list(self._plans.values())[-1][0].conv_bytes()

    #: what ``precision = 'auto'`` resolves to, and the modes this architecture implements (subclasses widen both)
    auto_precision = 'bf16x3'
    precisions = ('bf16x3', 'bf16')
    precision_table: dict = {}  # mode name -> (default products, default plane format) where an architecture departs from PRECISIONS

    def resolved_precision(self) -> str:
        name = self.precision
        if name == 'auto':
            name = 'bf16x3' if getattr(self, '_fp16_refused', False) else self.auto_precision
        if name not in self.precisions:
            raise ValueError(f"precision must be 'auto' or one of {list(self.precisions)} for {type(self).__name__}, got {self.precision!r}")
        return name

    @property
    def products(self) -> Prec:
        """The resolved precision as an int-compatible ``Prec`` (``int(prec)`` = matrix products per layer without a per-layer rule)."""
        return Prec(self.resolved_precision(), self.precision_table)

    # -- hooks --
    def _pack(self, device, products: int):
        raise NotImplementedError

    def _build_plan(self, plan: Plan, packed, x_shape, dtype, products: int):
        """Return (input_setter, output_getter)."""
        raise NotImplementedError

    def _weights(self, device):
        key = (str(device), self.resolved_precision())
        # in-place edits of a parameter (p.data.copy_, optimizer steps, load into .data) bump its version counter: repack when any changed
        versions = tuple(p._version for p in self.parameters()) + tuple(b._version for b in self.buffers())
        if versions != self._packed_versions:
            self._packed = {}
            self._packed_versions = versions
        w = self._packed.get(key)
        if w is None:
            with torch.no_grad():
                try:
                    w = self._pack(device, self.products)
                except Fp16Range as e:
                    if self.precision != 'auto':
                        raise
                    # pack-time guard of the fp16 modes: a weight beyond the fp16 range -> the conservative mode, and say so
                    import warnings

                    warnings.warn(f'{type(self).__name__}: {e}; precision "auto" falls back to "bf16x3"', RuntimeWarning, stacklevel=3)
                    self._fp16_refused = True
                    key = (str(device), self.resolved_precision())
                    w = self._pack(device, self.products)
            self._packed = {key: w}
            for k in list(self._plans):
                self._drop_plan(k)
        return w

    #: models whose first layout kernel and last store take 8-bit images directly set this (``model(uint8 [N, H, W, C]) -> uint8 [N, H*s, W*s, C]``)
    supports_u8 = False

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        ops.require_cuda(x, type(self).__name__)
        if x.dtype == torch.uint8:
            return self._forward_u8(x)
        if x.dim() != 4:
            raise ValueError(f'expected a [N, C, H, W] tensor, got shape {tuple(x.shape)}')
        ops.rsa_dtype(x.dtype)  # raises for unsupported dtypes
        return self._forward(x, tuple(x.shape))

    def _forward_u8(self, img: torch.Tensor) -> torch.Tensor:
        """8-bit image in, 8-bit image out ([H, W, C] or [N, H, W, C], channel-interleaved): ``/255`` happens in the first layout kernel and
        ``clamp(0, 1) * 255`` + round-half-even in the last convolution's store when the model supports it; otherwise two extra kernels
        do the same around an fp32 forward (bit-identical results either way)."""
        if img.dim() not in (3, 4):
            raise ValueError(f'expected a uint8 [N, H, W, C] or [H, W, C] image, got shape {tuple(img.shape)}')
        squeeze = img.dim() == 3
        if squeeze:
            img = img.unsqueeze(0)
        if self.supports_u8:
            n, h, w, c = img.shape
            y = self._forward(img.contiguous(), (n, c, h, w))
        else:
            y = ops.nchw_to_image_u8(self._forward(ops.image_u8_to_nchw(img, torch.float32), None))
        return y[0] if squeeze else y

    def _forward(self, x: torch.Tensor, shape) -> torch.Tensor:
        with self._lock:
            return self._forward_locked(x, shape)

    def _forward_locked(self, x: torch.Tensor, shape) -> torch.Tensor:
        shape = tuple(x.shape) if shape is None else shape
        first = next(self.parameters(), None)
        if first is not None and first.device != x.device:
            raise RuntimeError(f'model parameters are on {first.device} but the input is on {x.device}')
        packed = self._weights(x.device)
        key = (shape, x.dtype, str(x.device), self.resolved_precision())
        entry = self._plans.pop(key, None)
        new_plan = entry is None
        if entry is None:
            plan = Plan(x.device, self.products.fmt)
            set_input, get_output = self._build_plan(plan, packed, shape, x.dtype, self.products)
            plan.flush()
            entry = [plan, set_input, get_output, None]
        self._plans[key] = entry  # most recently used last
        # least-recently-used plans go first when the cache holds too many plans or too many bytes (never the one about to run)
        while len(self._plans) > 1 and (len(self._plans) > self._max_plans or sum(e[0].buffer_bytes() for e in self._plans.values()) > self.max_plan_bytes):
            self._drop_plan(next(iter(self._plans)))
        plan, set_input, get_output, graph = entry
        name = type(self).__name__
        guard = self._fp16_guard()  # the resolved policy has fp16 layers: scan for non-finite values behind every forward
        auto = self.precision == 'auto'
        with torch.cuda.device(x.device):
            if not self.use_graph:
                set_input(x.contiguous())
                stale = None
                try:
                    plan.run()
                finally:
                    # The status words of the library are read (and cleared) exactly once per forward, whether or not the launch list
                    # raised: a failure reported by an EARLIER forward's kernels makes rsa_conv2d_list refuse to launch, and that refusal
                    # must not leave the word set for every later call.  Host-visible words, no synchronisation: they report every launch
                    # that has completed by now (``sync_check`` / ``tiling.upscale*`` synchronise first to judge this forward).
                    try:
                        L.check_status(name)
                    except L.Fp16RangeError as e:
                        stale = e  # an EARLIER forward's range check (this one's kernels are still in flight)
                y = get_output()
                if stale is not None:
                    if not auto:
                        raise stale
                    return self._fp16_fallback(x, shape, 'an earlier forward produced non-finite values in its fp16 layers (its result was invalid)')
                if guard:
                    self._range_probe(plan, y)
                    if auto and new_plan:
                        # the first forward of a plan under 'auto' is judged before it is returned: one synchronisation per input signature
                        torch.cuda.synchronize(x.device)
                        try:
                            L.check_status(name)
                        except L.Fp16RangeError:
                            return self._fp16_fallback(x, shape, 'activations left the fp16 range of the one-product layers')
                return y
            if graph is None:
                static_x = x.contiguous().clone()
                set_input(static_x)  # one eager pass first: lazy initialisation and allocator warm-up must not happen under capture
                try:
                    plan.run()
                finally:
                    L.check_status(name)
                y0 = get_output()
                if guard:
                    self._range_probe(plan, y0)
                torch.cuda.synchronize(x.device)
                try:
                    L.check_status(name)  # nothing pending when capture starts (a replay never goes through rsa_conv2d_list's own test)
                except L.Fp16RangeError:
                    if not auto:
                        raise
                    return self._fp16_fallback(x, shape, 'activations left the fp16 range of the one-product layers')
                del y0
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):  # every C-ABI launch of the plan asks for the CURRENT stream, i.e. the capture stream
                    set_input(static_x)
                    plan.run()
                    static_y = get_output()
                    if guard:
                        self._range_probe(plan, static_y)
                graph = entry[3] = (g, static_x, static_y)
            g, static_x, static_y = graph
            static_x.copy_(x)
            g.replay()
            y = static_y.clone()
            try:
                L.check_status(name)  # as in the eager path: whatever has completed by now
            except L.Fp16RangeError as e:
                if not auto:
                    raise
                return self._fp16_fallback(x, shape, str(e))
            return y

    # -- run-time guard of the fp16 precision policies --
    def _fp16_guard(self) -> bool:
        """Whether the resolved precision runs layers on fp16 planes (then every forward is followed by ``rsa_check_finite``)."""
        p = self.products
        return p.name in ('mixed', 'fp16') or p.fmt == PF_F16

    def _range_probe(self, plan: 'Plan', y: torch.Tensor) -> None:
        """Scan for non-finite values where an fp16 overflow of this forward must show: the tensors a plan builder named
        (``plan.range_probe``: e.g. RRDBNet's trunk output, which every residual dense block feeds), else the output itself."""
        stream = ops.current_stream_ptr(y.device)
        probes = getattr(plan, 'range_probe', None)
        if probes:
            for t in probes:
                L.check_finite(t, stream)
        elif y.is_floating_point():
            L.check_finite(y if y.is_contiguous() else y.contiguous(), stream)

    def _fp16_fallback(self, x: torch.Tensor, shape, why: str) -> torch.Tensor:
        """'auto' only: give up the fp16 policy for this module (until its weights change), say so, and run this input in three bf16 products."""
        import warnings

        warnings.warn(f'{type(self).__name__}: {why}; precision "auto" falls back to "bf16x3" (set model.precision = "bf16x3" to skip the attempt)',
                      RuntimeWarning, stacklevel=4)  # fmt: skip
        self._fp16_refused = True
        self._packed = {}
        for key in list(self._plans):
            self._drop_plan(key)
        return self._forward_locked(x, shape)

    def sync_check(self, device=None) -> None:
        """Synchronise ``device`` (default: the parameters' device) and raise if a kernel of any forward issued so far reported a failed
        hand-off -- the call that makes the result of the LAST forward trustworthy (``forward`` itself never synchronises)."""
        if device is None:
            first = next(self.parameters(), None)
            device = first.device if first is not None else None
        torch.cuda.synchronize(device)
        L.check_status(type(self).__name__)
