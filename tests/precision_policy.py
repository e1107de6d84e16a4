"""CPU emulation of the engine's per-layer arithmetic policies on top of the oracle (TEST INFRASTRUCTURE).

``oracle.rrdbnet._conv`` is replaced by a convolution whose operands are rounded the way the engine's kernels round them:

    'f32'     no rounding (the oracle itself)
    'bf16x3'  x = hi + lo, w = hi + lo in bf16;  acc = hi*hi + lo*hi + hi*lo   (the engine's conservative mode)
    'fp16x3'  the same with fp16 halves (the trunk convolution of the 'auto' policy: its input planes are fp16)
    'bf16'    acc = bf16(x) * bf16(w)
    'fp16'    acc = fp16(x) * fp16(w)                                          (one product on v_mfma_f32_16x16x32_f16)

Accumulation is f32 in every case, as in the MFMA.  ``policy`` maps an oracle convolution key (``model.0``,
``model.1.sub.3.RDB2.conv4.0`` ...) to one of those names.  The RRDBNet table of the engine's ``precision = 'auto'``
(resselt_amd/archs/esrgan/arch.py::RRDBNet.layer_policy) is what the tests feed in.

The residual stream between residual dense blocks is rounded to ``hi + lo`` of the storage format (about 22 bits for fp16 halves),
as the engine stores it (planes only, no f32 copy).
"""

from __future__ import annotations

from contextlib import contextmanager

import torch
import torch.nn.functional as F


def _halves(t: torch.Tensor, dt: torch.dtype):
    hi = t.to(dt).float()
    lo = (t - hi).to(dt).float()
    return hi, lo


def round_conv(x: torch.Tensor, w: torch.Tensor, b, mode: str) -> torch.Tensor:
    pad = w.shape[-1] // 2
    if mode == 'f32':
        return F.conv2d(x, w, b, padding=pad)
    if mode in ('bf16', 'fp16'):
        dt = torch.bfloat16 if mode == 'bf16' else torch.float16
        return F.conv2d(x.to(dt).float(), w.to(dt).float(), b, padding=pad)
    if mode in ('bf16x3', 'fp16x3'):
        dt = torch.bfloat16 if mode == 'bf16x3' else torch.float16
        xh, xl = _halves(x, dt)
        wh, wl = _halves(w, dt)
        return F.conv2d(xh, wl, None, padding=pad) + F.conv2d(xl, wh, None, padding=pad) + F.conv2d(xh, wh, b, padding=pad)
    raise ValueError(mode)


@contextmanager
def emulate(policy, stream_dtype: torch.dtype | None = None):
    """Patch the RRDBNet oracle: ``policy(key) -> mode``; ``stream_dtype`` rounds every RDB output to hi + lo of that dtype."""
    import oracle.rrdbnet as O

    orig_conv, orig_rdb = O._conv, O.rdb_forward

    def conv(sd, key, x):
        return round_conv(x, sd[f'{key}.weight'], sd.get(f'{key}.bias'), policy(key))

    def rdb(sd, prefix, x, plus=False):
        y = orig_rdb(sd, prefix, x, plus)
        if stream_dtype == 'lo8':  # round 4: fp16 hi + 8-bit code of the lo half (rsa_conv_params.lo8_flags)
            from resselt_amd.engine.tensors import lo8_decode, lo8_encode

            y = lo8_decode(*lo8_encode(y))
        elif stream_dtype is not None:
            hi, lo = _halves(y, stream_dtype)
            y = hi + lo
        return y

    O._conv, O.rdb_forward = conv, rdb
    try:
        yield
    finally:
        O._conv, O.rdb_forward = orig_conv, orig_rdb


def uniform(mode: str):
    return lambda key: mode


def rrdbnet_auto(key: str) -> str:
    """The engine's 'auto' table for RRDBNet: convolutions inside residual dense blocks in ONE fp16 product; conv_first, the upsampling
    convolutions, the HR convolution and the last convolution in three bf16 products; the trunk convolution in three fp16 products
    (its input is the fp16 residual stream)."""
    if '.RDB' in key:
        return 'fp16'
    if key.startswith('model.1.sub.'):
        return 'fp16x3'
    return 'bf16x3'


class _FProxy:
    """``torch.nn.functional`` with ``conv2d`` rounded per policy (the plain convolutions of the SPAN oracles: conv_cat, the upsampler)."""

    def __init__(self, policy):
        self.policy, self.idx = policy, 0

    def __getattr__(self, k):
        return getattr(F, k)

    def conv2d(self, x, w, b=None, padding=0, **kw):
        mode = self.policy(f'plain{self.idx}')
        self.idx += 1
        if w.shape[-1] == 1 or padding == 1:
            return round_conv(x, w, b, mode)
        return F.conv2d(x, w, b, padding=padding, **kw)


@contextmanager
def emulate_span(policy, shortcut_dtype: torch.dtype | None = None):
    """Patch the SPAN / SPANPlus oracle: every Conv3XC becomes its folded 3x3 convolution rounded as ``policy(prefix)`` says, the plain
    convolutions (numbered in call order: conv_cat ..., upsampler) as ``policy('plainN')``; the SPAB gate's shortcut is rounded to
    ``shortcut_dtype`` (the engine reads it from the hi plane in its one-product modes)."""
    import oracle.span as O

    orig = O.conv3xc, O.F, O.spab

    def c3(sd, prefix, x):
        w, b = O.conv3xc_fold(sd, prefix)
        return round_conv(x, w, b, policy(prefix))

    def spab(sd, prefix, x, act):
        out1 = act(c3(sd, f'{prefix}.c1_r', x))
        out2 = c3(sd, f'{prefix}.c2_r', out1)
        out3 = c3(sd, f'{prefix}.c3_r', act(out2))
        xs = x if shortcut_dtype is None else x.to(shortcut_dtype).float()
        return (out3 + xs) * (torch.sigmoid(out3) - 0.5), out1

    O.conv3xc, O.spab, O.F = c3, spab, _FProxy(policy)
    try:
        yield
    finally:
        O.conv3xc, O.F, O.spab = orig


def span_mixed(key: str) -> str:
    """The SPAN family's 'mixed' table (resselt_amd/engine/spanblocks.py::span_layer_policy): Conv3XC layers in one fp16 product; the first
    convolution and the plain convolutions (conv_cat, the upsampler head) in three fp16 products."""
    return 'fp16x3' if key.startswith('plain') or key in ('feats.0', 'conv_1') else 'fp16'

