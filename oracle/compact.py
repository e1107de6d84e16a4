"""Oracle for SRVGGNetCompact (TEST INFRASTRUCTURE, see oracle/__init__.py)."""

from __future__ import annotations

from typing import Mapping

import torch
import torch.nn.functional as F


def compact_forward(sd: Mapping[str, torch.Tensor], x: torch.Tensor) -> torch.Tensor:
    """SRVGGNetCompact.forward (resselt/archs/compact/arch.py:55-65)."""
    last = max(int(k.split('.')[1]) for k in sd if k.startswith('body.'))
    out = x
    for i in range(0, last, 2):
        out = F.conv2d(out, sd[f'body.{i}.weight'], sd[f'body.{i}.bias'], padding=1)
        out = F.prelu(out, sd[f'body.{i + 1}.weight'])
    out = F.conv2d(out, sd[f'body.{last}.weight'], sd[f'body.{last}.bias'], padding=1)
    scale = int(round((out.shape[1] // x.shape[1]) ** 0.5))
    out = F.pixel_shuffle(out, scale)
    return out + F.interpolate(x, scale_factor=scale, mode='nearest')  # the network learns the residual (:61-64)
