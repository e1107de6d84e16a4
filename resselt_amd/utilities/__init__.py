from .state_dict import (
    canonicalize_state_dict,
    dysample_scale,
    get_pixelshuffle_params,
    get_seq_len,
    pixelshuffle_scale,
    remove_common_prefix,
)

__all__ = [
    'canonicalize_state_dict',
    'dysample_scale',
    'get_pixelshuffle_params',
    'get_seq_len',
    'pixelshuffle_scale',
    'remove_common_prefix',
]
