"""Plugin contract of the loader (drop-in for ``resselt/factory/arch.py:12-36``).

``Architecture`` is what users register with :func:`resselt_amd.add`; ``ModelMetadata`` is attached to
every loaded model as ``model.parameters_info``.
"""

from __future__ import annotations

from abc import ABC, abstractmethod
from dataclasses import dataclass
from typing import Generic, Mapping, Sequence, TypeVar, Union

import torch

from .key_condition import KeyCondition

T = TypeVar('T', bound=torch.nn.Module, covariant=True)


@dataclass
class ModelMetadata:
    """Facts about a loaded SR model that cannot be read off its tensors by a caller."""

    in_channels: int
    out_channels: int
    upscale: Union[int, Sequence[int]]
    name: str


class Architecture(ABC, Generic[T]):
    """One model family: a detector over checkpoint keys plus a builder."""

    def __init__(self, uid: str, detect: KeyCondition):
        self.id = uid
        self._detect = detect

    def detect(self, state_dict: Mapping[str, object]) -> bool:
        return self._detect(state_dict)

    @abstractmethod
    def load(self, state_dict: Mapping[str, object]) -> T:
        """Build the (still weight-less) module whose hyper-parameters match ``state_dict``."""
        raise NotImplementedError

    def _enhance_model(self, model: T, in_channels: int, out_channels: int, upscale: Union[int, Sequence[int]], name: str) -> T:
        model.parameters_info = ModelMetadata(in_channels=in_channels, out_channels=out_channels, upscale=upscale, name=name)
        return model
