#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by running the REAL reference (rewaifu/resselt).

Runs only in the build container, where /root/reference is mounted read-only.  Nothing of the reference is
copied: it is imported, fed deterministic synthetic checkpoints (resselt_amd/utils/synth.py) and seeded
inputs, and only the resulting input/output *vectors* (+ the metadata the reference's loaders inferred)
are written, as compressed .npz files.

Shims applied before import (both are harness-side, the reference tree is untouched):
  1. ``typing.Self`` (Python 3.10 lacks it; archs/smosr/arch.py:1 and archs/spanpp/arch.py:2 import it);
  2. DySample asks for ``pin_memory=True`` (utilities/dysample.py:62), which raises on a GPU-less host: the
     ``torch`` name inside that module is replaced by a proxy whose ``tensor()`` drops the flag.

Usage:  python tools/gen_golden.py            # rewrites tests/golden/*.npz
"""

from __future__ import annotations

import json
import os
import sys
import typing

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

import typing_extensions  # noqa: E402

if not hasattr(typing, 'Self'):
    typing.Self = typing_extensions.Self  # shim 1
sys.path.insert(0, '/root/reference')

import torch  # noqa: E402

import resselt  # noqa: E402  (the reference)
import resselt.utilities.dysample as _ref_dys  # noqa: E402
from resselt.utilities import block as RB  # noqa: E402

from resselt_amd.utils import synth  # noqa: E402


class _TorchNoPin:  # shim 2
    def __getattr__(self, name):
        return getattr(torch, name)

    @staticmethod
    def tensor(*args, **kwargs):
        kwargs.pop('pin_memory', None)
        return torch.tensor(*args, **kwargs)


_ref_dys.torch = _TorchNoPin()

OUT = os.path.join(ROOT, 'tests', 'golden')
torch.manual_seed(0)
torch.set_grad_enabled(False)


def save(name: str, meta: dict, **arrays):
    os.makedirs(OUT, exist_ok=True)
    meta = dict(meta, torch=torch.__version__, generator='tools/gen_golden.py')
    np.savez_compressed(os.path.join(OUT, name + '.npz'), meta=np.array(json.dumps(meta)), **{k: np.asarray(v) for k, v in arrays.items()})
    sizes = {k: tuple(np.asarray(v).shape) for k, v in arrays.items()}
    print(f'{name}: {sizes}')


def meta_of(model) -> dict:
    pi = model.parameters_info
    return dict(in_channels=pi.in_channels, out_channels=pi.out_channels, upscale=pi.upscale, name=pi.name, cls=type(model).__name__)


# ------------------------------------------------------------------ RRDBNet / ESRGAN
def esrgan_cases():
    cases = [
        ('rrdbnet_x4_nb23_32x32', dict(nb=23, scale=4), (1, 3, 32, 32), 1),
        ('rrdbnet_x4_nb23_40x56', dict(nb=23, scale=4), (1, 3, 40, 56), 2),
        ('rrdbnet_x2_nb3_b2_19x27', dict(nb=3, scale=2), (2, 3, 19, 27), 3),
        ('rrdbnet_x1_nb2_17x33', dict(nb=2, scale=1), (1, 3, 17, 33), 4),
        ('rrdbnet_plus_x4_nb2_24x24', dict(nb=2, scale=4, plus=True), (1, 3, 24, 24), 5),
        ('rrdbnet_x2plus_unshuffle_nb2_21x30', dict(nb=2, scale=4, in_nc=12), (1, 3, 21, 30), 6),
        ('rrdbnet_x4_nf32_nb1_1x1', dict(nb=1, scale=4, nf=32), (1, 3, 1, 1), 7),
    ]
    for name, kw, shape, seed in cases:
        sd = synth.rrdbnet_state_dict(seed=seed, **kw)
        model = resselt.load_from_state_dict(dict(sd))
        x = synth.synth_input(shape, seed)
        y = model(x)
        save(name, dict(arch='esrgan', synth=kw, seed=seed, metadata=meta_of(model)), x=x, y=y)

    # loader-only fixtures: what the reference's ESRGANArch.load infers for the other key spellings.
    # (registry.load_from_state_dict fails for new-arch keys in the reference -- SURVEY.md §3.1 -- so arch.load is called directly.)
    arch = resselt.get('ESRGAN')
    infos = {}
    for tag, kw in (
        ('old', dict(nb=3, scale=4)),
        ('new', dict(nb=3, scale=4, new_arch=True)),
        ('new_x2', dict(nb=2, scale=2, new_arch=True)),
        ('plus', dict(nb=2, scale=4, plus=True)),
        ('x2plus', dict(nb=2, scale=4, in_nc=12)),
        ('x1_unshuffle4', dict(nb=2, scale=4, in_nc=48)),
    ):
        sd = synth.rrdbnet_state_dict(seed=0, **kw)
        assert arch.detect(sd)
        m = arch.load(dict(sd))
        infos[tag] = dict(synth=kw, metadata=meta_of(m), n_params=len(m.state_dict()), shuffle_factor=m.shuffle_factor, model_scale=m.scale)
    save('esrgan_loader', dict(arch='esrgan', cases=infos))


def block_cases():
    """Per-block vectors from the reference's shared blocks (utilities/block.py)."""
    seed = 11
    sd_full = synth.rrdbnet_state_dict(nb=1, seed=seed)
    x = synth.synth_input((1, 64, 12, 20), seed) * 2 - 1
    rdb = RB.ResidualDenseBlock_5C(64, 3, 32, 1, True, 'zero', None, 'leakyrelu', 'CNA')
    rdb.load_state_dict({k[len('model.1.sub.0.RDB1.') :]: v for k, v in sd_full.items() if k.startswith('model.1.sub.0.RDB1.')})
    rrdb = RB.RRDB(64, 3, 32, 1, True, 'zero', None, 'leakyrelu', 'CNA')
    rrdb.load_state_dict({k[len('model.1.sub.0.') :]: v for k, v in sd_full.items() if k.startswith('model.1.sub.0.')})
    up = RB.upconv_block(64, 64, act_type='leakyrelu')
    up.load_state_dict({'1.weight': sd_full['model.3.weight'], '1.bias': sd_full['model.3.bias']})
    save('blocks_rrdb', dict(arch='esrgan', seed=seed, synth=dict(nb=1)), x=x, rdb=rdb(x), rrdb=rrdb(x), upconv=up(x))


# ------------------------------------------------------------------ SPAN / SPANPlus
def span_cases():
    from resselt.archs.spanplus.arch import SPAB as RefSPAB
    from resselt.archs.spanplus.arch import Conv3XC as RefConv3XC

    cases = [
        ('spanplus_ps_x2_24x40', dict(upscale=2, upsampler='ps'), (1, 3, 24, 40), 21),
        ('spanplus_ps_x4_b2_17x23', dict(upscale=4, upsampler='ps'), (2, 3, 17, 23), 22),
        ('spanplus_dys_x2_24x40', dict(upscale=2, upsampler='dys'), (1, 3, 24, 40), 23),
        ('spanplus_dys_x4_20x28', dict(upscale=4, upsampler='dys'), (1, 3, 20, 28), 24),
        ('spanplus_ps_x4_blocks2_2_fc32', dict(upscale=4, upsampler='ps', blocks=(2, 2), feature_channels=32), (1, 3, 16, 16), 25),
    ]
    for name, kw, shape, seed in cases:
        sd = synth.spanplus_state_dict(seed=seed, **kw)
        model = resselt.load_from_state_dict(dict(sd))
        x = synth.synth_input(shape, seed)
        y_train = model(x)  # the loader returns the module in training mode (unfused Conv3XC path)
        y_eval = model.eval()(x)  # folded path
        kw = dict(kw, blocks=list(kw.get('blocks', (4,))))
        save(name, dict(arch='spanplus', synth=kw, seed=seed, metadata=meta_of(model), train_eval_maxdiff=float((y_train - y_eval).abs().max())), x=x, y=y_train)

    for name, kw, shape, seed in [
        ('span_x4_24x40', dict(upscale=4), (1, 3, 24, 40), 31),
        ('span_x2_nonorm_19x21', dict(upscale=2, norm=False), (1, 3, 19, 21), 32),
    ]:
        sd = synth.span_state_dict(seed=seed, **kw)
        model = resselt.load_from_state_dict(dict(sd))
        x = synth.synth_input(shape, seed)
        save(name, dict(arch='span', synth=kw, seed=seed, metadata=meta_of(model)), x=x, y=model(x))

    # Conv3XC fold and one SPAB, straight from the reference classes
    seed = 41
    sd = synth.spanplus_state_dict(seed=seed)
    c = RefConv3XC(48, 48, gain=2, s=1)
    c.load_state_dict({k[len('feats.1.block_1.c1_r.') :]: v for k, v in sd.items() if k.startswith('feats.1.block_1.c1_r.')})
    c.update_params()
    blk = RefSPAB(48, end=True)
    blk.load_state_dict({k[len('feats.1.block_1.') :]: v for k, v in sd.items() if k.startswith('feats.1.block_1.')})
    x = synth.synth_input((1, 48, 10, 14), seed) * 2 - 1
    out, out1 = blk(x)
    save('blocks_span', dict(arch='spanplus', seed=seed, synth={}), x=x, fold_w=c.eval_conv.weight.data, fold_b=c.eval_conv.bias.data, spab_out=out, spab_out1=out1)


def compact_cases():
    for name, kw, shape, seed in [
        ('compact_x4_nf64_nc16_20x28', dict(num_feat=64, num_conv=16, upscale=4), (1, 3, 20, 28), 71),
        ('compact_x2_nf32_nc4_b2_17x19', dict(num_feat=32, num_conv=4, upscale=2), (2, 3, 17, 19), 72),
    ]:
        sd = synth.compact_state_dict(seed=seed, **kw)
        model = resselt.load_from_state_dict(dict(sd))
        x = synth.synth_input(shape, seed)
        save(name, dict(arch='compact', synth=kw, seed=seed, metadata=meta_of(model)), x=x, y=model(x))


def dat_cases():
    """DAT end to end, eval mode (BatchNorm running statistics; the loader returns a train-mode module whose DropPath is random)."""
    cases = [
        ('dat_x2_e64_s2x4_d3_1_3conv_direct_16x16', dict(embed_dim=64, depth=(3, 1), num_heads=(4, 4), resi='3conv', upsampler='pixelshuffledirect'), (1, 3, 16, 16), 81),
        ('dat_x2_e64_s2x4_d3_1_3conv_direct_13x18', dict(embed_dim=64, depth=(3, 1), num_heads=(4, 4), resi='3conv', upsampler='pixelshuffledirect'), (1, 3, 13, 18), 82),
        ('dat_x4_e64_s4x8_d3_ps_b2_20x28', dict(embed_dim=64, depth=(3,), num_heads=(4,), split_size=(4, 8), upscale=4), (2, 3, 20, 28), 83),
        ('dat_x3_e180_s8x16_d2_ps_24x40', dict(embed_dim=180, depth=(2,), num_heads=(6,), split_size=(8, 16), upscale=3), (1, 3, 24, 40), 84),
        ('dat_x2_e180_s8x32_d3_ps_40x72', dict(embed_dim=180, depth=(3,), num_heads=(6,), split_size=(8, 32), upscale=2, expansion_factor=4.0, img_size=64), (1, 3, 40, 72), 85),
        ('dat_light_x2_e60_s8x32_d4_direct_33x47', dict(embed_dim=60, depth=(4,), num_heads=(6,), split_size=(8, 32), upscale=2, resi='3conv', upsampler='pixelshuffledirect', img_size=64), (1, 3, 33, 47), 86),
    ]
    for name, kw, shape, seed in cases:
        sd = synth.dat_state_dict(seed=seed, **kw)
        model = resselt.load_from_state_dict(dict(sd)).eval()
        x = synth.synth_input(shape, seed)
        y = model(x)
        kw = {k: (list(v) if isinstance(v, tuple) else v) for k, v in kw.items()}
        save(name, dict(arch='dat', synth=kw, seed=seed, metadata=meta_of(model), mode='eval'), x=x, y=y)


def spanpp_cases():
    """SpanPP end to end in eval mode (the only mode the reference module can run in: IGConv reads a table that .eval() fills)."""
    cases = [
        ('spanpp_fc48_x2_default_24x40', dict(), (1, 3, 24, 40), None, 91),
        ('spanpp_fc48_x4_b2_17x23', dict(), (2, 3, 17, 23), 4, 92),
        ('spanpp_fc32_x3_id64_l2_19x21', dict(feature_channels=32, implicit_dim=64, latent_layers=2), (1, 3, 19, 21), 3, 93),
        ('spanpp_fc32_x1_scales124_16x16', dict(feature_channels=32, implicit_dim=64, latent_layers=2, scale_list=(1, 2, 4)), (1, 3, 16, 16), 1, 94),
    ]
    for name, kw, shape, scale, seed in cases:
        sd = synth.spanpp_state_dict(seed=seed, **kw)
        model = resselt.load_from_state_dict(dict(sd)).eval()
        x = synth.synth_input(shape, seed)
        y = model(x) if scale is None else model(x, scale)
        kw = {k: (list(v) if isinstance(v, tuple) else v) for k, v in kw.items()}
        meta = meta_of(model)
        save(name, dict(arch='spanpp', synth=kw, seed=seed, scale=scale, metadata=meta, mode='eval'), x=x, y=y)


def hat_cases():
    """HAT end to end, eval mode (DropPath 0.1 is random in the train-mode module the loader returns)."""
    cases = [
        ('hat_x2_e60_w8_d2_2_20x27', dict(), (1, 3, 20, 27), 101),
        ('hat_x4_e180_w16_d2_b2_32x48', dict(embed_dim=180, depths=(2,), num_heads=(6,), window=16, upscale=4), (2, 3, 32, 48), 102),
        ('hat_x3_e96_w8_d3_identity_25x40', dict(embed_dim=96, depths=(3,), num_heads=(6,), upscale=3, mlp_ratio=4.0, resi='identity'), (1, 3, 25, 40), 103),
        ('hat_x2_e180_w16_d2_2_50x70', dict(embed_dim=180, depths=(2, 2), num_heads=(6, 6), window=16, upscale=2), (1, 3, 50, 70), 104),
    ]
    for name, kw, shape, seed in cases:
        sd = synth.hat_state_dict(seed=seed, **kw)
        model = resselt.load_from_state_dict(dict(sd)).eval()
        x = synth.synth_input(shape, seed)
        y = model(x)
        kw = {k: (list(v) if isinstance(v, tuple) else v) for k, v in kw.items()}
        save(name, dict(arch='hat', synth=kw, seed=seed, metadata=meta_of(model), mode='eval'), x=x, y=y)


def drct_cases():
    """DRCT end to end (dense groups of five Swin blocks with 30..122-channel heads; no stochastic layers, train mode == eval mode)."""
    cases = [
        ('drct_x2_e180_g32_l1_40x50', dict(num_layers=1, upscale=2), (1, 3, 40, 50), 201),
        ('drct_x4_e180_g32_l2_48x64', dict(num_layers=2, upscale=4), (1, 3, 48, 64), 202),
        ('drct_x3_e60_g16_w8_l2_33x33', dict(embed_dim=60, gc=16, window=8, num_layers=2, upscale=3, mlp_ratio=4.0), (1, 3, 33, 33), 203),
        # resi_connection='identity' (no conv_after_body: drct/arch.py:731-732), a batch of two
        ('drct_x2_e60_g16_w8_identity_b2_24x40', dict(embed_dim=60, gc=16, window=8, num_layers=1, upscale=2, resi='identity'), (2, 3, 24, 40), 204),
        # a checkpoint without attn_mask buffers: the loader passes img_size = window and no block is shifted (drct/arch.py:373-376)
        ('drct_x2_e60_g16_w8_nomask_32x24', dict(embed_dim=60, gc=16, window=8, num_layers=1, upscale=2, attn_mask=False), (1, 3, 32, 24), 205),
    ]
    for name, kw, shape, seed in cases:
        sd = synth.drct_state_dict(seed=seed, **kw)
        model = resselt.load_from_state_dict(dict(sd))
        x = synth.synth_input(shape, seed)
        y = model(x)
        save(name, dict(arch='drct', synth=kw, seed=seed, metadata=meta_of(model)), x=x, y=y)


def rtmosr_cases():
    """RTMoSR end to end, eval mode (every RepConv / OmniShift re-parameterised; the train-mode branches are the same function)."""
    cases = [
        ('rtmosr_x2_d32_b2_13x17', dict(), (1, 3, 13, 17), 111),
        ('rtmosr_x4_d48_b2_nose_b2_20x28', dict(scale=4, dim=48, se=False), (2, 3, 20, 28), 112),
        ('rtmosr_x2_unshuffle_d32_ffn15_nodccm_21x30', dict(scale=2, dim=32, ffn_expansion=1.5, n_blocks=1, unshuffle_mod=True, dccm=False), (1, 3, 21, 30), 113),
        # (scale 1 with unshuffle 4 cannot load in the reference: its loader reads the unshuffle factor as the scale, rtmosr/__init__.py:178-180)
    ]
    for name, kw, shape, seed in cases:
        sd = synth.rtmosr_state_dict(seed=seed, **kw)
        model = resselt.load_from_state_dict(dict(sd)).eval()
        x = synth.synth_input(shape, seed)
        y = model(x)
        save(name, dict(arch='rtmosr', synth=kw, seed=seed, metadata=meta_of(model), mode='eval'), x=x, y=y)


def registry_cases():
    """Detection order facts: which reference architecture claims each synthetic checkpoint."""
    claims = {}
    for tag, sd in (
        ('rrdbnet_old', synth.rrdbnet_state_dict(nb=1)),
        ('spanplus_ps', synth.spanplus_state_dict(upsampler='ps')),
        ('spanplus_dys', synth.spanplus_state_dict(upsampler='dys')),
        ('span', synth.span_state_dict()),
        ('compact', synth.compact_state_dict(num_conv=2)),
        ('swinir', synth.swinir_state_dict()),
        ('dat', synth.dat_state_dict()),
        ('hat', synth.hat_state_dict()),
        ('rtmosr', synth.rtmosr_state_dict()),
        ('drct', synth.drct_state_dict(num_layers=1)),
        ('spanpp', synth.spanpp_state_dict(feature_channels=16, implicit_dim=32, latent_layers=1)),
    ):
        for arch in resselt.archs.internal_registry.store.values():
            if arch.detect(sd):
                claims[tag] = arch.id
                break
    save('registry_claims', dict(claims=claims, order=list(resselt.archs.internal_registry.store.keys())))


if __name__ == '__main__':
    which = sys.argv[1:] or ['esrgan', 'blocks', 'span', 'compact', 'registry', 'swinir', 'dat', 'spanpp', 'hat', 'rtmosr', 'drct']
    if 'esrgan' in which:
        esrgan_cases()
    if 'blocks' in which:
        block_cases()
    if 'span' in which:
        span_cases()
    if 'compact' in which:
        compact_cases()
    if 'dat' in which:
        dat_cases()
    if 'spanpp' in which:
        spanpp_cases()
    if 'hat' in which:
        hat_cases()
    if 'drct' in which:
        drct_cases()
    if 'rtmosr' in which:
        rtmosr_cases()
    if 'registry' in which:
        registry_cases()
    if 'swinir' in which:
        try:
            from gen_golden_swinir import swinir_cases  # type: ignore
        except ImportError:
            swinir_cases = None
        if swinir_cases:
            swinir_cases(save, meta_of)
