"""Plan steps shared by the Swin-style transformer models (SwinIR, HAT): the fused block halves of csrc/swin_block.hip."""

from __future__ import annotations

import ctypes as C

from . import lib as L
from . import ops

MLP_MAX_C, MLP_MAX_HIDDEN = 256, 512  # limits of rsa_swin_mlp_block (include/resselt_amd.h)


def mlp_block_fits(channels: int, hidden: int) -> bool:
    return channels <= MLP_MAX_C and channels % 4 == 0 and hidden <= MLP_MAX_HIDDEN


def mlp_block(plan, norm, fc1, fc2, n, h, w, channels, hidden, products, x_f32, out_f32, out_planes=None, eps=1e-5):
    """``out = x + fc2(GELU(fc1(LayerNorm(x))))`` in one launch (reference archs/swinir/arch.py:331-335 with Mlp.forward :34-40; the
    same lines close a HAT block, archs/hat/arch.py).  ``norm`` = (gamma, beta) f32 tensors, ``fc1`` / ``fc2`` = ops.ConvWeights of the
    Linear layers, ``x_f32`` / ``out_f32`` = f32 NCHW4c maps (may be the same), ``out_planes`` = optional split-plane copy."""
    lib = L.load()
    dev = plan.device
    g, be = norm
    mp = L.SwinMlpBlockParams()
    if fc1.products != fc2.products or fc1.fmt != fc2.fmt:
        raise ValueError('fc1 and fc2 must be packed for the same arithmetic')
    # (the arithmetic is what the weights were packed for: an architecture's per-layer policy may run this half in one fp16 product)
    mp.batch, mp.H, mp.W, mp.C, mp.hidden, mp.products, mp.eps = n, h, w, channels, hidden, fc1.products, eps
    mp.fmt = fc1.fmt
    mp.x, mp.gamma, mp.beta = x_f32.data_ptr(), g.data_ptr(), be.data_ptr()
    mp.w1, mp.b1 = fc1.packed_for(0).data_ptr(), fc1.bias.data_ptr()
    mp.w2, mp.b2 = fc2.packed_for(0).data_ptr(), fc2.bias.data_ptr()
    mp.out = out_f32.data_ptr()
    if out_planes is not None:
        mp.out_hi, mp.out_lo = out_planes.hi_ptr(), out_planes.lo_ptr()
        mp.out_plane_stride, mp.out_batch_stride = out_planes.plane_stride, out_planes.batch_stride
    plan.call(lambda: L.check(lib.rsa_swin_mlp_block(C.byref(mp), C.c_void_p(ops.current_stream_ptr(dev))), 'rsa_swin_mlp_block'))
    plan.count_launches(1)
