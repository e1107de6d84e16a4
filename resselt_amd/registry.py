"""Architecture registry and checkpoint-file loading (drop-in for ``resselt/registry.py:14-116``).

Behaviour kept from the reference:
  * ``get`` of an unknown id raises ``KeyError`` (registry.py:73-77);
  * ``load_from_file`` dispatches on the extension: ``.pt`` tries TorchScript first and falls back to a
    restricted pickle, ``.pth``/``.ckpt`` use the restricted pickle, ``.safetensors`` uses safetensors,
    anything else raises ``ValueError`` (registry.py:79-104);
  * the restricted unpickler only admits OrderedDict, ``_rebuild_tensor_v2`` and six storage classes and
    raises ``pickle.UnpicklingError`` for every other global (registry.py:20-46);
  * ``load_from_state_dict`` canonicalises, walks the architectures in insertion order, builds the first
    match and calls ``model.load_state_dict(canonicalised_dict)`` (registry.py:106-116);
  * no match raises ``ArchitectureNotFound``.

Documented deviation: the reference hands the *unconverted* dict to ``load_state_dict``, so official
Real-ESRGAN "new-arch" checkpoints are detected but then fail with missing keys.  The
engine modules returned here accept both key spellings in ``load_state_dict`` so those checkpoints load.
"""

from __future__ import annotations

import os
import pickle
from types import SimpleNamespace
from typing import Dict, Iterator, Mapping

import torch

from .factory import Architecture
from .utilities.state_dict import canonicalize_state_dict


class ArchitectureNotFound(Exception):
    pass


_ALLOWED_GLOBALS = frozenset(
    {
        ('collections', 'OrderedDict'),
        ('typing', 'OrderedDict'),
        ('torch._utils', '_rebuild_tensor_v2'),
        ('torch', 'BFloat16Storage'),
        ('torch', 'FloatStorage'),
        ('torch', 'HalfStorage'),
        ('torch', 'IntStorage'),
        ('torch', 'LongStorage'),
        ('torch', 'DoubleStorage'),
    }
)


class RestrictedUnpickler(pickle.Unpickler):
    """Unpickler that can only rebuild plain tensor dictionaries."""

    def find_class(self, module: str, name: str):
        if (module, name) not in _ALLOWED_GLOBALS:
            raise pickle.UnpicklingError(f"Global '{module}.{name}' is forbidden")
        return super().find_class(module, name)


def _restricted_load(*args, **kwargs):
    return RestrictedUnpickler(*args, **kwargs).load()


# what torch.load(pickle_module=...) needs: a module-like object with Unpickler/load and a __name__
RestrictedUnpickle = SimpleNamespace(Unpickler=RestrictedUnpickler, __name__='pickle', load=_restricted_load)


def _read_checkpoint(path: str) -> Mapping[str, object]:
    ext = os.path.splitext(path)[1].lower()
    if ext == '.pt':
        try:
            return torch.jit.load(path).state_dict()
        except RuntimeError:
            try:
                return torch.load(path, pickle_module=RestrictedUnpickle)
            except Exception:
                pass
            raise
    if ext in ('.pth', '.ckpt'):
        return torch.load(path, pickle_module=RestrictedUnpickle)
    if ext == '.safetensors':
        import safetensors.torch

        return safetensors.torch.load_file(path)
    raise ValueError(f'Unsupported model file extension {ext}. Please try a supported model type.')


class Registry:
    def __init__(self):
        self.store: Dict[str, Architecture] = {}

    def __contains__(self, uid: str) -> bool:
        return uid in self.store

    def __iter__(self) -> Iterator[Architecture]:
        return iter(list(self.store.values()))

    def __len__(self) -> int:
        return len(self.store)

    def add(self, arch: Architecture) -> None:
        """Register an architecture *instance*; re-using an id replaces it in place."""
        self.store[arch.id] = arch

    def get(self, uid: str) -> Architecture:
        arch = self.store[uid]  # KeyError for unknown ids, as in the reference
        if not arch:
            raise ArchitectureNotFound
        return arch

    def load_from_file(self, path: str):
        return self.load_from_state_dict(_read_checkpoint(path))

    def load_from_state_dict(self, state_dict: Mapping[str, object]):
        state_dict = canonicalize_state_dict(state_dict)
        for arch in self.store.values():
            if arch.detect(state_dict):
                model = arch.load(state_dict)
                model.load_state_dict(state_dict)
                return model
        raise ArchitectureNotFound
