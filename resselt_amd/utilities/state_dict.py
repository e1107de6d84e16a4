"""Checkpoint-dictionary helpers (behavioural mirror of ``resselt/utilities/state_dict.py:5-96``)."""

from __future__ import annotations

import math
from typing import Mapping

# wrappers training frameworks put around the real tensor dict, probed in this order
_WRAPPER_KEYS = ('state_dict', 'params_ema', 'params-ema', 'params', 'model', 'net')
# prefixes DataParallel / GAN trainers add to every key
_COMMON_PREFIXES = ('module.', 'netG.')


def remove_common_prefix(state_dict: Mapping[str, object], prefixes) -> Mapping[str, object]:
    """Strip each prefix that is shared by *every* key (in the given order)."""
    if not state_dict:
        return state_dict
    for prefix in prefixes:
        if all(key.startswith(prefix) for key in state_dict):
            cut = len(prefix)
            state_dict = {key[cut:]: value for key, value in state_dict.items()}
    return state_dict


def canonicalize_state_dict(state_dict: Mapping[str, object]) -> Mapping[str, object]:
    """Unwrap one known container level, then drop ``module.`` / ``netG.`` prefixes."""
    for key in _WRAPPER_KEYS:
        inner = state_dict.get(key) if hasattr(state_dict, 'get') else None
        if isinstance(inner, dict):
            state_dict = inner
            break
    return remove_common_prefix(state_dict, _COMMON_PREFIXES)


def pixelshuffle_scale(ps_size: int, channels: int) -> int:
    """Upscale of a conv -> PixelShuffle head from the conv's output channel count."""
    return math.isqrt(ps_size // channels)


def dysample_scale(ds_size: int) -> int:
    """Upscale of a DySample head (offset conv has 2 * groups(4) * s^2 channels)."""
    return math.isqrt(ds_size // 8)


def get_pixelshuffle_params(state_dict: Mapping[str, object], upsample_key: str = 'upsample', default_nf: int = 64) -> tuple[int, int]:
    """(upscale, num_feat) of an ``Upsample`` stack: conv at even indices, PixelShuffle between."""
    upscale, num_feat = 1, default_nf
    for index in range(0, 10, 2):
        weight = state_dict.get(f'{upsample_key}.{index}.weight')
        if weight is None:
            break
        out_ch, num_feat = weight.shape[0], weight.shape[1]
        upscale *= math.isqrt(out_ch // num_feat)
    return upscale, num_feat


def get_seq_len(state_dict: Mapping[str, object], seq_key: str) -> int:
    """1 + the largest integer i for which a key ``{seq_key}.{i}.…`` (or ``{seq_key}.{i}``) exists."""
    prefix = seq_key + '.'
    best = -1
    for key in state_dict:
        if key.startswith(prefix):
            best = max(best, int(key[len(prefix) :].split('.', 1)[0]))
    return best + 1
