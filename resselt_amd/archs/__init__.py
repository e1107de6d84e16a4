"""Architectures served by the MI355X engine, registered explicitly in detection order.

The reference discovers 31 architectures by walking the filesystem (``resselt/archs/__init__.py:11-28``);
this build registers the families of the hot path: ESRGAN/RRDBNet, SPANPlus, SPAN, SwinIR, DAT, HAT, and the "next" rows of §8f built so far (Compact, SpanPP, RTMoSR); the first
"next" row of §8f (Compact / SRVGGNetCompact, pure reuse of the conv kernel).
"""

from ..registry import Registry
from .compact import CompactArch
from .dat import DatArch
from .drct import DRCTArch
from .esrgan import ESRGANArch
from .hat import HATArch
from .rtmosr import RTMoSRArch
from .span import SPANArch
from .spanplus import SpanPlusArch
from .spanpp import SpanPPArch
from .swinir import SwinIRArch

internal_registry = Registry()
# relative order follows the reference's registry walk (tests/golden/registry_claims.npz): ESRGAN, HAT, dat, Compact, RTMoSR, spanplus, SwinIR, SpanPP, ..., SPAN
for _arch in (ESRGANArch, HATArch, DatArch, CompactArch, RTMoSRArch, SpanPlusArch, SwinIRArch, SpanPPArch, DRCTArch, SPANArch):
    internal_registry.add(_arch())
