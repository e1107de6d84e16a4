"""resselt_amd -- MI355X-native drop-in for the forward-pass hot path of rewaifu/resselt.

Public surface (same four functions as ``resselt/__init__.py:6-26``)::

    model = resselt_amd.load_from_file('RealESRGAN_x4plus.pth').cuda()
    y = model(x)                      # HIP kernels via the C-ABI library, no PyTorch compute fallback
    model.parameters_info             # ModelMetadata(in_channels, out_channels, upscale, name)
"""

from typing import Mapping

from .archs import internal_registry
from .factory import Architecture, KeyCondition, ModelMetadata
from .registry import ArchitectureNotFound, Registry

__version__ = '0.1.0'


def add(arch):
    """Register a new architecture (an ``Architecture`` instance)."""
    return internal_registry.add(arch)


def get(id: str):
    """Get architecture by ID (``KeyError`` when unknown)."""
    return internal_registry.get(id)


def load_from_file(path: str):
    """Read a .pth/.ckpt/.pt/.safetensors checkpoint, detect its architecture and load it."""
    return internal_registry.load_from_file(path)


def load_from_state_dict(state_dict: Mapping[str, object]):
    """Detect the architecture of a state dict and load it."""
    return internal_registry.load_from_state_dict(state_dict)


def upscale(model, image, **kwargs):
    """uint8 [H, W, C] / [N, H, W, C] GPU image in, upscaled uint8 image out (see ``resselt_amd.tiling.upscale``).  Not part of the reference API."""
    from .tiling import upscale as _upscale

    return _upscale(model, image, **kwargs)


__all__ = [
    'add',
    'get',
    'load_from_file',
    'load_from_state_dict',
    'Architecture',
    'ArchitectureNotFound',
    'KeyCondition',
    'ModelMetadata',
    'Registry',
    'upscale',
]
