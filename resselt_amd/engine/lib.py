"""ctypes binding of ``libresselt_amd.so`` (C-ABI declared in ``include/resselt_amd.h``).

The library is the product: there is no CPU or PyTorch fallback behind these wrappers.
``load()`` raises ``RuntimeError`` when the shared object is missing or a symbol is absent,
and every call raises ``RuntimeError`` with ``rsa_last_error_string()`` on a non-zero status.
"""

from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_LIB_NAME = 'libresselt_amd.so'
_lib: Optional[C.CDLL] = None

# enum rsa_act
ACT_NONE, ACT_LRELU, ACT_MISH, ACT_SILU, ACT_GELU, ACT_SPAB_GATE, ACT_PRELU = range(7)
# enum rsa_dtype
F32, F16, BF16, U8 = range(4)
LO8_RES1, LO8_RES2, LO8_OUT = 1, 2, 4  # rsa_conv_params.lo8_flags
# enum rsa_plane_fmt
PF_BF16, PF_F16 = range(2)
E_INTERNAL = -4


class ConvParams(C.Structure):
    """Mirror of ``struct rsa_conv_params`` (include/resselt_amd.h); field order is the ABI."""

    _fields_ = [
        ('batch', C.c_int32),
        ('H', C.c_int32),
        ('W', C.c_int32),
        ('ksize', C.c_int32),
        ('upsample2x', C.c_int32),
        ('cin_planes', C.c_int32),
        ('cout', C.c_int32),
        ('products', C.c_int32),
        ('in_hi', C.c_void_p),
        ('in_lo', C.c_void_p),
        ('in_plane_stride', C.c_int64),
        ('in_batch_stride', C.c_int64),
        ('w_packed', C.c_void_p),
        ('bias', C.c_void_p),
        ('act', C.c_int32),
        ('act_param', C.c_float),
        ('alpha', C.c_float),
        ('res1', C.c_void_p),
        ('beta', C.c_float),
        ('res2', C.c_void_p),
        ('out_hi', C.c_void_p),
        ('out_lo', C.c_void_p),
        ('out_plane_off', C.c_int32),
        ('out_plane_stride', C.c_int64),
        ('out_batch_stride', C.c_int64),
        ('out_f32', C.c_void_p),
        ('out_nchw', C.c_void_p),
        ('out_dtype', C.c_int32),
        ('pixel_shuffle', C.c_int32),
        ('out_scale', C.c_float),
        ('out_shift', C.c_void_p),
        ('act_vec', C.c_void_p),
        ('out_base', C.c_void_p),
        ('out_base_div', C.c_int32),
        ('out_base_h', C.c_int32),
        ('out_base_w', C.c_int32),
        ('w_layout', C.c_int32),
        ('res1_hi', C.c_void_p),
        ('res1_lo', C.c_void_p),
        ('res2_hi', C.c_void_p),
        ('res2_lo', C.c_void_p),
        ('res_plane_stride', C.c_int64),
        ('res_batch_stride', C.c_int64),
        ('in_fmt', C.c_int32),
        ('out_fmt', C.c_int32),
        ('res_fmt', C.c_int32),
        ('tile_order', C.c_int32),
        ('lo8_flags', C.c_int32),
        ('reserved_lo8', C.c_int32),
        ('lo8_batch_stride', C.c_int64),
    ]


class DySampleParams(C.Structure):
    """Mirror of ``struct rsa_dysample_params``."""

    _fields_ = [
        ('batch', C.c_int32),
        ('H', C.c_int32),
        ('W', C.c_int32),
        ('C', C.c_int32),
        ('groups', C.c_int32),
        ('scale', C.c_int32),
        ('out_ch', C.c_int32),
        ('x_f32', C.c_void_p),
        ('offscope', C.c_void_p),
        ('init_pos', C.c_void_p),
        ('end_w', C.c_void_p),
        ('end_b', C.c_void_p),
        ('out_nchw', C.c_void_p),
        ('out_dtype', C.c_int32),
    ]


class LayerNormParams(C.Structure):
    """Mirror of ``struct rsa_layernorm_params``."""

    _fields_ = [
        ('batch', C.c_int32),
        ('H', C.c_int32),
        ('W', C.c_int32),
        ('C', C.c_int32),
        ('eps', C.c_float),
        ('x_f32', C.c_void_p),
        ('gamma', C.c_void_p),
        ('beta', C.c_void_p),
        ('out_hi', C.c_void_p),
        ('out_lo', C.c_void_p),
        ('out_plane_stride', C.c_int64),
        ('out_batch_stride', C.c_int64),
        ('out_f32', C.c_void_p),
        ('out_fmt', C.c_int32),
        ('reserved0', C.c_int32),
    ]


class SwinAttnBlockParams(C.Structure):
    """Mirror of ``struct rsa_swin_attn_block_params``."""

    _fields_ = [
        ('batch', C.c_int32),
        ('H', C.c_int32),
        ('W', C.c_int32),
        ('C', C.c_int32),
        ('heads', C.c_int32),
        ('window', C.c_int32),
        ('shift', C.c_int32),
        ('products', C.c_int32),
        ('eps', C.c_float),
        ('x', C.c_void_p),
        ('gamma', C.c_void_p),
        ('beta', C.c_void_p),
        ('wqkv', C.c_void_p),
        ('bqkv', C.c_void_p),
        ('bias_frag16', C.c_void_p),
        ('wproj', C.c_void_p),
        ('bproj', C.c_void_p),
        ('out', C.c_void_p),
    ]


class SwinMlpBlockParams(C.Structure):
    """Mirror of ``struct rsa_swin_mlp_block_params``."""

    _fields_ = [
        ('batch', C.c_int32),
        ('H', C.c_int32),
        ('W', C.c_int32),
        ('C', C.c_int32),
        ('hidden', C.c_int32),
        ('products', C.c_int32),
        ('eps', C.c_float),
        ('x', C.c_void_p),
        ('gamma', C.c_void_p),
        ('beta', C.c_void_p),
        ('w1', C.c_void_p),
        ('b1', C.c_void_p),
        ('w2', C.c_void_p),
        ('b2', C.c_void_p),
        ('out', C.c_void_p),
        ('out_hi', C.c_void_p),
        ('out_lo', C.c_void_p),
        ('out_plane_stride', C.c_int64),
        ('out_batch_stride', C.c_int64),
        ('fmt', C.c_int32),
        ('reserved0', C.c_int32),
    ]


class SwinBlockParams(C.Structure):
    """Mirror of ``struct rsa_swin_block_params``."""

    _fields_ = [
        ('batch', C.c_int32),
        ('H', C.c_int32),
        ('W', C.c_int32),
        ('C', C.c_int32),
        ('heads', C.c_int32),
        ('window', C.c_int32),
        ('shift', C.c_int32),
        ('hidden', C.c_int32),
        ('products', C.c_int32),
        ('eps', C.c_float),
        ('x', C.c_void_p),
        ('gamma1', C.c_void_p),
        ('beta1', C.c_void_p),
        ('wqkv', C.c_void_p),
        ('bqkv', C.c_void_p),
        ('bias_frag16', C.c_void_p),
        ('wproj', C.c_void_p),
        ('bproj', C.c_void_p),
        ('gamma2', C.c_void_p),
        ('beta2', C.c_void_p),
        ('w1', C.c_void_p),
        ('b1', C.c_void_p),
        ('w2', C.c_void_p),
        ('b2', C.c_void_p),
        ('out', C.c_void_p),
        ('out_hi', C.c_void_p),
        ('out_lo', C.c_void_p),
        ('out_plane_stride', C.c_int64),
        ('out_batch_stride', C.c_int64),
        ('fmt', C.c_int32),
        ('reserved0', C.c_int32),
    ]


class WindowAttnParams(C.Structure):
    """Mirror of ``struct rsa_window_attn_params``."""

    _fields_ = [
        ('batch', C.c_int32),
        ('H', C.c_int32),
        ('W', C.c_int32),
        ('heads', C.c_int32),
        ('window', C.c_int32),
        ('shift', C.c_int32),
        ('products', C.c_int32),
        ('qkv_hi', C.c_void_p),
        ('qkv_lo', C.c_void_p),
        ('qkv_plane_stride', C.c_int64),
        ('qkv_batch_stride', C.c_int64),
        ('bias_frag', C.c_void_p),
        ('out_hi', C.c_void_p),
        ('out_lo', C.c_void_p),
        ('out_plane_stride', C.c_int64),
        ('out_batch_stride', C.c_int64),
    ]


class RectAttnParams(C.Structure):
    """Mirror of ``struct rsa_rect_attn_params``."""

    _fields_ = [
        ('batch', C.c_int32),
        ('H', C.c_int32),
        ('W', C.c_int32),
        ('Hp', C.c_int32),
        ('Wp', C.c_int32),
        ('win_h', C.c_int32),
        ('win_w', C.c_int32),
        ('shift_h', C.c_int32),
        ('shift_w', C.c_int32),
        ('heads', C.c_int32),
        ('head0', C.c_int32),
        ('heads_total', C.c_int32),
        ('products', C.c_int32),
        ('qkv_hi', C.c_void_p),
        ('qkv_lo', C.c_void_p),
        ('qkv_plane_stride', C.c_int64),
        ('qkv_batch_stride', C.c_int64),
        ('bias_frag', C.c_void_p),
        ('out_hi', C.c_void_p),
        ('out_lo', C.c_void_p),
        ('out_plane_stride', C.c_int64),
        ('out_batch_stride', C.c_int64),
        ('kwin_h', C.c_int32),
        ('kwin_w', C.c_int32),
        ('kpad_h', C.c_int32),
        ('kpad_w', C.c_int32),
        ('head_chunks', C.c_int32),
        ('fmt', C.c_int32),
        ('reserved0', C.c_int32),
    ]


class ChannelAttnParams(C.Structure):
    """Mirror of ``struct rsa_channel_attn_params``."""

    _fields_ = [
        ('batch', C.c_int32),
        ('H', C.c_int32),
        ('W', C.c_int32),
        ('heads', C.c_int32),
        ('head_dim', C.c_int32),
        ('products', C.c_int32),
        ('q_hi', C.c_void_p),
        ('q_lo', C.c_void_p),
        ('k_hi', C.c_void_p),
        ('k_lo', C.c_void_p),
        ('plane_stride', C.c_int64),
        ('batch_stride', C.c_int64),
        ('temperature', C.c_void_p),
        ('workspace', C.c_void_p),
        ('w_packed', C.c_void_p),
        ('fmt', C.c_int32),
        ('reserved1', C.c_int32),
    ]


class DwConvParams(C.Structure):
    """Mirror of ``struct rsa_dwconv_params``."""

    _fields_ = [
        ('batch', C.c_int32),
        ('H', C.c_int32),
        ('W', C.c_int32),
        ('planes', C.c_int32),
        ('act', C.c_int32),
        ('in_hi', C.c_void_p),
        ('in_lo', C.c_void_p),
        ('in_plane_stride', C.c_int64),
        ('in_batch_stride', C.c_int64),
        ('weight', C.c_void_p),
        ('bias', C.c_void_p),
        ('stats', C.c_void_p),
        ('gamma', C.c_void_p),
        ('beta', C.c_void_p),
        ('mul_hi', C.c_void_p),
        ('mul_lo', C.c_void_p),
        ('mul_plane_stride', C.c_int64),
        ('mul_batch_stride', C.c_int64),
        ('out_hi', C.c_void_p),
        ('out_lo', C.c_void_p),
        ('out_plane_stride', C.c_int64),
        ('out_batch_stride', C.c_int64),
        ('fmt', C.c_int32),
        ('reserved1', C.c_int32),
    ]


class ChannelGateParams(C.Structure):
    """Mirror of ``struct rsa_channel_gate_params``."""

    _fields_ = [
        ('batch', C.c_int32),
        ('H', C.c_int32),
        ('W', C.c_int32),
        ('planes', C.c_int32),
        ('hidden', C.c_int32),
        ('in_hi', C.c_void_p),
        ('in_lo', C.c_void_p),
        ('in_plane_stride', C.c_int64),
        ('in_batch_stride', C.c_int64),
        ('w1', C.c_void_p),
        ('b1', C.c_void_p),
        ('w2', C.c_void_p),
        ('b2', C.c_void_p),
        ('workspace', C.c_void_p),
        ('gate', C.c_void_p),
        ('relu', C.c_int32),
        ('fmt', C.c_int32),
    ]


class GatedShuffleParams(C.Structure):
    """Mirror of ``struct rsa_gated_shuffle_params``."""

    _fields_ = [
        ('batch', C.c_int32),
        ('H', C.c_int32),
        ('W', C.c_int32),
        ('g_planes', C.c_int32),
        ('i_planes', C.c_int32),
        ('f_hi', C.c_void_p),
        ('f_lo', C.c_void_p),
        ('f_plane_stride', C.c_int64),
        ('f_batch_stride', C.c_int64),
        ('c_hi', C.c_void_p),
        ('c_lo', C.c_void_p),
        ('c_plane_stride', C.c_int64),
        ('c_batch_stride', C.c_int64),
        ('gate', C.c_void_p),
        ('gate_stride', C.c_int64),
        ('out_hi', C.c_void_p),
        ('out_lo', C.c_void_p),
        ('out_plane_stride', C.c_int64),
        ('out_batch_stride', C.c_int64),
    ]


class AimParams(C.Structure):
    """Mirror of ``struct rsa_aim_params``."""

    _fields_ = [
        ('batch', C.c_int32),
        ('H', C.c_int32),
        ('W', C.c_int32),
        ('planes', C.c_int32),
        ('hidden', C.c_int32),
        ('mode', C.c_int32),
        ('att_hi', C.c_void_p),
        ('att_lo', C.c_void_p),
        ('att_plane_stride', C.c_int64),
        ('att_batch_stride', C.c_int64),
        ('conv_hi', C.c_void_p),
        ('conv_lo', C.c_void_p),
        ('conv_plane_stride', C.c_int64),
        ('conv_batch_stride', C.c_int64),
        ('gate', C.c_void_p),
        ('w1', C.c_void_p),
        ('b1', C.c_void_p),
        ('w2', C.c_void_p),
        ('b2', C.c_float),
        ('out_hi', C.c_void_p),
        ('out_lo', C.c_void_p),
        ('out_plane_stride', C.c_int64),
        ('out_batch_stride', C.c_int64),
        ('fmt', C.c_int32),
        ('reserved1', C.c_int32),
    ]


# every symbol include/resselt_amd.h declares (checked by tests/test_capi_symbols.py)
EXPORTS = (
    'rsa_version',
    'rsa_last_error_string',
    'rsa_conv2d',
    'rsa_conv2d_list',
    'rsa_conv_cout_tiles',
    'rsa_packed_weight_bytes',
    'rsa_packed_weight_bytes_layout',
    'rsa_conv_weight_layout',
    'rsa_pack_weights',
    'rsa_conv_kernel_name',
    'rsa_check_status',
    'rsa_check_finite',
    'rsa_debug_ring_aborts',
    'rsa_debug_set_ring_spin_limit',
    'rsa_debug_set_ring',
    'rsa_debug_set_pair',
    'rsa_conv2d_pair',
    'rsa_conv_pair_fusable',
    'rsa_nchw_to_planes',
    'rsa_planes_to_nchw',
    'rsa_dysample',
    'rsa_layernorm',
    'rsa_window_attention',
    'rsa_swin_attn_block',
    'rsa_swin_mlp_block',
    'rsa_swin_block',
    'rsa_rect_attention',
    'rsa_channel_attn_workspace_bytes',
    'rsa_channel_attention_weights',
    'rsa_dwconv3x3',
    'rsa_plane_stats',
    'rsa_plane_stats_fmt',
    'rsa_channel_gate_workspace_bytes',
    'rsa_channel_gate',
    'rsa_aim_combine',
    'rsa_gated_add',
    'rsa_dwconv5x5',
    'rsa_rmsnorm',
    'rsa_unshuffle_pool',
    'rsa_gated_shuffle_mul',
    'rsa_image_u8_to_nchw',
    'rsa_nchw_to_image_u8',
)


def lib_path() -> str:
    """The in-tree library; RSA_LIB=path selects an experiment build instead (tools/variant.sh: A/B timing, ablations)."""
    override = os.environ.get('RSA_LIB')
    if override:
        return os.path.abspath(override)
    return os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), _LIB_NAME)


def _warn_if_stale(path: str) -> None:
    """The build stamps the library with the hash of the sources it was built from (build.py); a library that no longer matches the
    sources next to it still loads (the GPU box only has the prebuilt file), but says so."""
    try:
        from .. import build as B

        if os.path.abspath(path) != os.path.abspath(B.OUT) or not os.path.isdir(B.CSRC):
            return
        have = None
        if os.path.exists(B.STAMP):
            with open(B.STAMP) as f:
                have = f.read().strip()
        if have != B._hash():
            import warnings

            warnings.warn(f'{_LIB_NAME} was built from other sources than the ones in {B.CSRC} (stamp {str(have)[:12]}); run `python -m resselt_amd.build`', RuntimeWarning, stacklevel=3)
    except OSError:
        pass


def load() -> C.CDLL:
    """Load the HIP library once; fail loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise RuntimeError(
            f'{_LIB_NAME} not found at {path}: build it with `python -c "import __graft_entry__ as g; g.build()"` '
            '(hipcc --offload-arch=gfx950). resselt_amd has no CPU fallback.'
        )
    _warn_if_stale(path)
    lib = C.CDLL(path)
    for name in EXPORTS:
        if not hasattr(lib, name):
            raise RuntimeError(f'{path} does not export {name}; rebuild the library')
    lib.rsa_version.restype = C.c_int
    lib.rsa_last_error_string.restype = C.c_char_p
    lib.rsa_conv2d.argtypes = [C.POINTER(ConvParams), C.c_void_p]
    lib.rsa_conv2d.restype = C.c_int
    lib.rsa_conv2d_list.argtypes = [C.POINTER(ConvParams), C.c_int32, C.c_void_p]
    lib.rsa_conv2d_list.restype = C.c_int
    lib.rsa_conv_cout_tiles.argtypes = [C.c_int32]
    lib.rsa_conv_cout_tiles.restype = C.c_int
    lib.rsa_packed_weight_bytes.argtypes = [C.c_int32] * 4
    lib.rsa_packed_weight_bytes.restype = C.c_int64
    lib.rsa_packed_weight_bytes_layout.argtypes = [C.c_int32] * 5
    lib.rsa_packed_weight_bytes_layout.restype = C.c_int64
    lib.rsa_conv_weight_layout.argtypes = [C.POINTER(ConvParams)]
    lib.rsa_conv_weight_layout.restype = C.c_int
    lib.rsa_pack_weights.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    lib.rsa_pack_weights.restype = C.c_int
    lib.rsa_conv_kernel_name.argtypes = [C.POINTER(ConvParams)]
    lib.rsa_conv_kernel_name.restype = C.c_char_p
    lib.rsa_debug_ring_aborts.argtypes = []
    lib.rsa_debug_ring_aborts.restype = C.c_int
    lib.rsa_debug_set_ring.argtypes = [C.c_int32]
    lib.rsa_debug_set_ring.restype = C.c_int
    lib.rsa_debug_set_pair.argtypes = [C.c_int32]
    lib.rsa_debug_set_pair.restype = C.c_int
    lib.rsa_conv2d_pair.argtypes = [C.POINTER(ConvParams), C.POINTER(ConvParams), C.c_void_p]
    lib.rsa_conv2d_pair.restype = C.c_int
    lib.rsa_conv_pair_fusable.argtypes = [C.POINTER(ConvParams), C.POINTER(ConvParams)]
    lib.rsa_conv_pair_fusable.restype = C.c_int
    lib.rsa_debug_set_ring_spin_limit.argtypes = [C.c_int32]
    lib.rsa_debug_set_ring_spin_limit.restype = C.c_int
    lib.rsa_check_finite.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_void_p]
    lib.rsa_check_finite.restype = C.c_int
    lib.rsa_check_status.argtypes = []
    lib.rsa_check_status.restype = C.c_int
    lib.rsa_nchw_to_planes.argtypes = [
        C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_float,
        C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p,
    ]  # fmt: skip
    lib.rsa_nchw_to_planes.restype = C.c_int
    lib.rsa_planes_to_nchw.argtypes = [
        C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
    ]  # fmt: skip
    lib.rsa_planes_to_nchw.restype = C.c_int
    lib.rsa_dysample.argtypes = [C.POINTER(DySampleParams), C.c_void_p]
    lib.rsa_dysample.restype = C.c_int
    lib.rsa_layernorm.argtypes = [C.POINTER(LayerNormParams), C.c_void_p]
    lib.rsa_layernorm.restype = C.c_int
    lib.rsa_window_attention.argtypes = [C.POINTER(WindowAttnParams), C.c_void_p]
    lib.rsa_window_attention.restype = C.c_int
    lib.rsa_swin_attn_block.argtypes = [C.POINTER(SwinAttnBlockParams), C.c_void_p]
    lib.rsa_swin_attn_block.restype = C.c_int
    lib.rsa_swin_mlp_block.argtypes = [C.POINTER(SwinMlpBlockParams), C.c_void_p]
    lib.rsa_swin_mlp_block.restype = C.c_int
    lib.rsa_swin_block.argtypes = [C.POINTER(SwinBlockParams), C.c_void_p]
    lib.rsa_swin_block.restype = C.c_int
    for name, struct in (('rsa_rect_attention', RectAttnParams), ('rsa_channel_attention_weights', ChannelAttnParams), ('rsa_dwconv3x3', DwConvParams),
                         ('rsa_channel_gate', ChannelGateParams), ('rsa_aim_combine', AimParams)):  # fmt: skip
        getattr(lib, name).argtypes = [C.POINTER(struct), C.c_void_p]
        getattr(lib, name).restype = C.c_int
    lib.rsa_channel_attn_workspace_bytes.argtypes = [C.c_int32] * 4
    lib.rsa_channel_attn_workspace_bytes.restype = C.c_int64
    lib.rsa_channel_gate_workspace_bytes.argtypes = [C.c_int32] * 4
    lib.rsa_channel_gate_workspace_bytes.restype = C.c_int64
    lib.rsa_plane_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p,
                                    C.c_void_p]  # fmt: skip
    lib.rsa_plane_stats.restype = C.c_int
    lib.rsa_plane_stats_fmt.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_int32,
                                        C.c_void_p, C.c_void_p]  # fmt: skip
    lib.rsa_plane_stats_fmt.restype = C.c_int
    lib.rsa_dwconv5x5.argtypes = [C.POINTER(DwConvParams), C.c_void_p]
    lib.rsa_dwconv5x5.restype = C.c_int
    lib.rsa_gated_shuffle_mul.argtypes = [C.POINTER(GatedShuffleParams), C.c_void_p]
    lib.rsa_gated_shuffle_mul.restype = C.c_int
    lib.rsa_rmsnorm.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_int64, C.c_int64, C.c_void_p]  # fmt: skip
    lib.rsa_rmsnorm.restype = C.c_int
    lib.rsa_unshuffle_pool.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]  # fmt: skip
    lib.rsa_unshuffle_pool.restype = C.c_int
    lib.rsa_gated_add.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_float,
                                  C.c_void_p, C.c_void_p, C.c_void_p]  # fmt: skip
    lib.rsa_gated_add.restype = C.c_int
    lib.rsa_image_u8_to_nchw.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]
    lib.rsa_image_u8_to_nchw.restype = C.c_int
    lib.rsa_nchw_to_image_u8.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    lib.rsa_nchw_to_image_u8.restype = C.c_int
    _lib = lib
    return lib


E_FP16_RANGE = -5  # RSA_E_FP16_RANGE


class Fp16RangeError(RuntimeError):
    """``rsa_check_status`` after ``rsa_check_finite`` met an infinity or a NaN: an activation of a one-product fp16 layer left the
    format's range (or the input was not finite).  ``precision = 'auto'`` answers it by re-running in three bf16 products."""


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().rsa_last_error_string()
        cls = Fp16RangeError if rc == E_FP16_RANGE else RuntimeError
        raise cls(f'{what} failed (status {rc}): {msg.decode() if msg else "?"}')


def check_finite(t, stream: int) -> None:
    """Scan a plain float tensor (or the hi / lo storage of a split-plane buffer) for non-finite values (``rsa_check_finite``; asynchronous:
    ``check_status`` raises ``Fp16RangeError`` once the launch has completed)."""
    import torch

    dt = {torch.float32: F32, torch.float16: F16, torch.bfloat16: BF16}[t.dtype]
    if not t.is_contiguous():
        raise ValueError('check_finite: the tensor must be contiguous')
    check(load().rsa_check_finite(t.data_ptr(), dt, t.numel(), C.c_void_p(stream)), 'rsa_check_finite')


def conv2d_list(params: 'C.Array[ConvParams] | list[ConvParams]', stream: int) -> None:
    """Launch a list of fused convolutions in order on ``stream`` with one host call."""
    lib = load()
    if isinstance(params, list):
        arr = (ConvParams * len(params))(*params)
    else:
        arr = params
    check(lib.rsa_conv2d_list(arr, len(arr), C.c_void_p(stream)), 'rsa_conv2d_list')


def conv_kernel_name(p: ConvParams) -> str:
    return load().rsa_conv_kernel_name(C.byref(p)).decode()


def ring_aborts() -> int:
    return int(load().rsa_debug_ring_aborts())


def check_status(what: str = 'forward') -> None:
    """Raise when a ring-schedule kernel of a COMPLETED launch reported a failed hand-off (``rsa_check_status``: reads a host-visible word,
    never synchronises).  Synchronise the stream first to judge the launches still in flight."""
    check(load().rsa_check_status(), f'{what}: rsa_check_status')


def conv2d_pair(a: ConvParams, b: ConvParams, stream: int) -> None:
    """Two consecutive growth convolutions of a residual dense block as ONE launch (``rsa_conv2d_pair``; csrc/conv_ring_pair.h)."""
    check(load().rsa_conv2d_pair(C.byref(a), C.byref(b), C.c_void_p(stream)), 'rsa_conv2d_pair')


def conv_pair_fusable(a: ConvParams, b: ConvParams) -> bool:
    return bool(load().rsa_conv_pair_fusable(C.byref(a), C.byref(b)))


def set_pair_fusion(mode: int) -> None:
    """Debug / A-B: 1 = ``rsa_conv2d_list`` fuses eligible neighbours, 0 = it launches them one by one, -1 = follow RSA_CONV_PAIR."""
    load().rsa_debug_set_pair(int(mode))


def set_ring_spin_limit(polls: int) -> None:
    load().rsa_debug_set_ring_spin_limit(int(polls))


def cout_tiles(cout: int) -> int:
    return int(load().rsa_conv_cout_tiles(cout))
