"""Shared launch-list builders for the SPAN family (SPAN, SPANPlus).

Reference blocks: ``Conv3XC`` (resselt/archs/spanplus/arch.py:9-102, resselt/archs/span/arch.py:59-154), ``SPAB``
(spanplus/arch.py:105-130), ``SPABS`` (spanplus/arch.py:133-151).  What the engine does differently:

  * Conv3XC is folded ONCE, at pack time, into a single 3x3 kernel + bias (the reference re-folds the three
    convolutions and the skip on every forward call, spanplus/arch.py:99-100).  The parameters under ``conv.*`` and
    ``sk.*`` win over a stored ``eval_conv.*``, exactly as in the reference.
  * Mish / SiLU and the parameter-free attention ``(out3 + x) * (sigmoid(out3) - 0.5)`` are conv epilogues.
  * ``torch.cat([x, out_end, out_b1, out_x_2])`` is one 4-slot plane buffer that the producers write into directly;
    ``conv_cat`` is a k1 convolution over it.
  * The reference's activation is in-place, so the ``out1`` it concatenates is the *activated* tensor
    (oracle/span.py, pinned by tests/golden/blocks_span.npz); the engine therefore needs no extra pre-activation store.
"""

from __future__ import annotations

import torch

from . import lib as L
from . import ops
from .base import Plan, check_fp16_range
from .tensors import PF_F16, Planes


def fold_conv3xc(sd: dict, prefix: str) -> tuple[torch.Tensor, torch.Tensor]:
    """(1x1 -> 3x3 valid -> 1x1) + 1x1 skip  ==  one 3x3 conv with zero padding 1.  Done in f64, returned as f32.

    W[o,i,y,x] = sum_{n,m} w3[o,n] w2[n,m,y,x] w1[m,i]  (+ sk[o,i] at the centre tap)
    b[o]       = sum_n w3[o,n] (sum_{m,y,x} w2[n,m,y,x] b1[m] + b2[n]) + b3[o] + b_sk[o]
    The first 1x1 sees a zero-padded input, so its bias b1 reaches every tap, border pixels included --
    which is what the reference's ``conv(pad0(x))`` does as well (spanplus/arch.py:95-97).
    """
    dev = sd[f'{prefix}.conv.1.weight'].device

    def get(k):  # on the host: the tensors are a few hundred KB, and the first f64 contraction on the GPU costs 0.2 s of library start-up
        return sd[f'{prefix}.{k}'].detach().to('cpu', torch.float64)

    w1, b1 = get('conv.0.weight')[:, :, 0, 0], get('conv.0.bias')
    w2, b2 = get('conv.1.weight'), get('conv.1.bias')
    w3, b3 = get('conv.2.weight')[:, :, 0, 0], get('conv.2.bias')
    w = torch.einsum('on,nmyx,mi->oiyx', w3, w2, w1)
    b = w3 @ (torch.einsum('nmyx,m->n', w2, b1) + b2) + b3
    w[:, :, 1, 1] += get('sk.weight')[:, :, 0, 0]
    b = b + get('sk.bias')
    return w.to(torch.float32).to(dev), b.to(torch.float32).to(dev)


def conv3xc_shapes(shapes: dict, name: str, cout: int, cin: int, gain: int = 2) -> None:
    for sub, (co, ci, k) in {
        'sk': (cout, cin, 1),
        'conv.0': (cin * gain, cin, 1),
        'conv.1': (cout * gain, cin * gain, 3),
        'conv.2': (cout, cout * gain, 1),
        'eval_conv': (cout, cin, 3),
    }.items():
        shapes[f'{name}.{sub}.weight'] = (co, ci, k, k)
        shapes[f'{name}.{sub}.bias'] = (co,)


def spab_shapes(shapes: dict, name: str, c: int) -> None:
    for r in ('c1_r', 'c2_r', 'c3_r'):
        conv3xc_shapes(shapes, f'{name}.{r}', c, c)


class SpabChain:
    """Emits the launches of [block_1, block_n..., block_end, conv_2, conv_cat] into a plan."""

    def __init__(self, plan: Plan, W: dict, n: int, h: int, w: int, fc: int, act: int, with_lo: bool, cat_lo: bool | None = None):
        self.plan, self.W, self.n, self.h, self.w, self.fc, self.act = plan, W, n, h, w, fc, act
        self.pf = fc // 8
        self.with_lo = with_lo
        # 'mixed': the scratch planes between the convolutions of a SPAB are hi only, the cat buffer (input of the three-product conv_cat,
        # and the gate's shortcut of the first SPAB) keeps hi + lo
        self.cat_lo = with_lo if cat_lo is None else cat_lo
        # scratch shared by every SPAB of the model
        self.t1 = plan.planes(n, self.pf, h, w, with_lo)
        self.t2 = plan.planes(n, self.pf, h, w, with_lo)
        self.ping = [plan.planes(n, self.pf, h, w, with_lo) for _ in range(2)]
        # The gate's shortcut is read from the block input's own split planes (value = hi + lo, 16 bits) in bf16x3 mode: no f32 copy of
        # every block output is written and read back (-12 % of a SPAB's bytes).  Plain-bf16 mode has no lo planes and keeps the f32 maps;
        # fp16 mode reads the shortcut from the hi plane (11 bits: what an fp16 run of the reference modules carries between layers).
        self.plane_shortcut = with_lo or plan.fmt == PF_F16
        self.f32 = [None] * 3 if self.plane_shortcut else [plan.f32map(n, fc, h, w) for _ in range(3)]

    def new_cat(self) -> Planes:
        return self.plan.planes(self.n, 4 * self.pf, self.h, self.w, self.cat_lo)

    def _conv(self, name, x, **kw):
        self.plan.conv(ops.conv_params(self.W[name], x, self.h, self.w, cin_planes=self.pf, **kw))

    def spab(self, name: str, x: Planes, x_plane0: int, xf: torch.Tensor, out: Planes, out_plane0: int, out_f32: torch.Tensor | None,
             out1: tuple[Planes, int] | None = None) -> None:  # fmt: skip
        """One SPAB.  ``out1`` (block_end only): where act(c1_r(x)) must ALSO live (slot 3 of the cat buffer)."""
        a1, a1_off = out1 if out1 is not None else (self.t1, 0)
        self._conv(f'{name}.c1_r', x, in_plane0=x_plane0, act=self.act, out=a1, out_plane_off=a1_off)
        self._conv(f'{name}.c2_r', a1, in_plane0=a1_off, act=self.act, out=self.t2)
        if self.plane_shortcut:
            self._conv(f'{name}.c3_r', self.t2, act=L.ACT_SPAB_GATE, res1=(x, x_plane0), out=out, out_plane_off=out_plane0)
        else:
            self._conv(f'{name}.c3_r', self.t2, act=L.ACT_SPAB_GATE, res1=xf, out=out, out_plane_off=out_plane0, out_f32=out_f32)

    def run(self, names: dict, cat: Planes, xf: torch.Tensor, out: Planes, out_plane0: int, out_f32: torch.Tensor | None) -> None:
        """``cat`` slot 0 already holds x (split planes) and ``xf`` its f32 map; result of conv_cat goes to ``out``."""
        pf = self.pf
        f_b1, f_a, f_b = self.f32
        # block_1 -> slot 2
        self.spab(names['first'], cat, 0, xf, cat, 2 * pf, f_b1)
        cur, cur_off, cur_f = cat, 2 * pf, f_b1
        for i, name in enumerate(names['middle']):
            nxt, nf = self.ping[i & 1], (f_a, f_b)[i & 1]
            self.spab(name, cur, cur_off, cur_f, nxt, 0, nf)
            cur, cur_off, cur_f = nxt, 0, nf
        # block_end: act(out1) -> slot 3, gated output -> scratch
        end_out = self.ping[len(names['middle']) & 1]
        self.spab(names['end'], cur, cur_off, cur_f, end_out, 0, None, out1=(cat, 3 * pf))
        # conv_2 -> slot 1 ; conv_cat (k1 over all four slots)
        self._conv(names['conv_2'], end_out, out=cat, out_plane_off=pf)
        self.plan.conv(ops.conv_params(self.W[names['conv_cat']], cat, self.h, self.w, cin_planes=4 * pf, out=out, out_plane_off=out_plane0, out_f32=out_f32))


#: the SPAN family's 'mixed' table (what their 'auto' selects).  The 19-20 re-parameterised 3x3 convolutions -- 95 % of the multiply-accumulates,
#: every one followed by the SPAB gate's (sigmoid - 0.5) attenuation or another such block -- run ONE fp16 product on hi planes; the layers
#: whose output reaches the image at full amplitude (conv_cat over the four concatenated maps, the upsampler head) run three fp16 products on
#: hi + lo planes, and so does the first convolution (3 input channels: its output is slot 0 of the cat buffer).  CPU emulation on SPANPlus x4
#: (tests/test_precision_policy.py): 4e-6 max-abs against fp32 where one product everywhere gives 2.1e-4 (conv_cat, the head and the first
#: convolution contribute 0.7-1.6e-4 each) -- with fp16 output tensors the former is half an ulp of the output, the latter is not.
SPAN_MIXED = {'mixed': (1, PF_F16)}
SPAN_FIRST = ('feats.0', 'conv_1', 'conv0')  # the first convolution of SPANPlus / SPAN / SpanPP


def span_layer_policy(name: str, conv3xc: bool) -> tuple[int, int]:
    """(products, plane format) of a SPAN-family layer under 'mixed'."""
    return (1, PF_F16) if conv3xc and name not in SPAN_FIRST else (3, PF_F16)


def pack_span_family(module, device, products: int, conv3xc_names: list[str], plain_names: list[str]) -> dict:
    # The re-parameterisation is weight preprocessing on ~100 KB of tensors: it runs on the HOST in f64 (no vendor GEMM / reduction kernel is
    # launched from inside the package); ConvWeights.from_oihw uploads the folded f32 kernel and packs it with rsa_pack_weights.
    sd = {k: v.detach().to(device='cpu', dtype=torch.float32) for k, v in module.state_dict().items()}
    mixed = getattr(products, 'name', '') == 'mixed'
    W = {}
    for name in conv3xc_names:
        w, b = fold_conv3xc(sd, name)
        prod, fmt = span_layer_policy(name, True) if mixed else (int(products), getattr(products, 'fmt', None))
        # the first convolution (3 input channels) is packed over a whole 16-channel half chunk: its input planes carry a second, all-zero plane
        # so that the layer takes the ring schedule (archs/spanplus/arch.py::_build_plan) -- only where the ring takes the padded layer at all:
        # three products and three cout tiles (33..48 features); elsewhere the extra plane would only be extra work for the chunk-barrier kernel
        ring_first = name in SPAN_FIRST and w.shape[1] <= 8 and int(prod) == 3 and (w.shape[0] + 15) // 16 == 3
        W[name] = ops.ConvWeights.from_oihw(w, b, prod, device=device, fmt=fmt, cin_planes=2 if ring_first else None)
    for name in plain_names:
        prod, fmt = span_layer_policy(name, False) if mixed else (int(products), getattr(products, 'fmt', None))
        W[name] = ops.ConvWeights.from_oihw(sd[f'{name}.weight'], sd.get(f'{name}.bias'), prod, device=device, fmt=fmt)
    check_fp16_range(W.values())
    return W
