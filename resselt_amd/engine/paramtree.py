"""Build an ``nn.Module`` whose parameter *names and shapes* equal the reference module's.

The engine models do not execute ``nn.Conv2d``; they only need to own tensors under the checkpoint's key
names so that ``state_dict()``, strict ``load_state_dict()``, ``.to()``, ``.half()`` behave exactly as for
the reference modules (SURVEY.md §8b "What the returned object must support").  ``ParamTree`` is a bare
container node; ``build_param_tree`` nests nodes along the dotted key path.
"""

from __future__ import annotations

from typing import Mapping, Sequence

import torch
from torch import nn


class ParamTree(nn.Module):
    """A container node: children are ParamTrees or Parameters/buffers named by one key component."""

    def forward(self, *args, **kwargs):  # pragma: no cover - never executed
        raise RuntimeError('ParamTree only stores parameters; the engine runs the forward pass')


def build_param_tree(root: nn.Module, shapes: Mapping[str, Sequence[int]], buffers: Mapping[str, torch.Tensor] | None = None) -> None:
    """Register an (uninitialised, zero) fp32 Parameter for every ``dotted.name -> shape``."""

    def node_for(path: list[str]) -> nn.Module:
        node = root
        for comp in path:
            child = node._modules.get(comp)
            if child is None:
                child = ParamTree()
                node.add_module(comp, child)
            node = child
        return node

    for name, shape in shapes.items():
        *path, leaf = name.split('.')
        node_for(path).register_parameter(leaf, nn.Parameter(torch.zeros(tuple(shape), dtype=torch.float32), requires_grad=False))
    for name, value in (buffers or {}).items():
        *path, leaf = name.split('.')
        node_for(path).register_buffer(leaf, value.clone())
