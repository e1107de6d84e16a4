"""Host-side boundary: registry / factory / key conditions / state-dict utilities / file loading.

Mirrors the behaviour catalogued in SURVEY.md §2.1, §3.1, §8b (reference: resselt/registry.py,
resselt/factory/*, resselt/utilities/state_dict.py).  The reference ships no tests; expectations that come
from the reference itself are read from tests/golden/esrgan_loader.npz and registry_claims.npz.
"""

import os
import pickle

import pytest
import torch

import resselt_amd
from helpers import load_golden
from resselt_amd import Architecture, ArchitectureNotFound, KeyCondition, ModelMetadata, Registry
from resselt_amd.registry import RestrictedUnpickle
from resselt_amd.utilities import state_dict as U
from resselt_amd.utils import synth


def test_key_condition_trees():
    sd = {'a': 1, 'b': 2}
    assert KeyCondition.has_all('a', 'b')(sd)
    assert not KeyCondition.has_all('a', 'c')(sd)
    assert KeyCondition.has_any('c', 'b')(sd)
    assert not KeyCondition.has_any('c', 'd')(sd)
    assert KeyCondition.has_any(KeyCondition.has_all('a', 'c'), KeyCondition.has_all('a', 'b'))(sd)
    assert KeyCondition.has_all()(sd) and not KeyCondition.has_any()(sd)


def test_canonicalize_unwraps_and_strips_prefixes():
    inner = {'module.x.weight': 1, 'module.y.bias': 2}
    assert U.canonicalize_state_dict({'params_ema': inner}) == {'x.weight': 1, 'y.bias': 2}
    assert U.canonicalize_state_dict({'state_dict': {'netG.a': 1}}) == {'a': 1}
    # first wrapper key in the fixed probe order wins; non-dict values are not unwrapped
    assert U.canonicalize_state_dict({'params': {'p': 1}, 'state_dict': {'s': 1}}) == {'s': 1}
    assert U.canonicalize_state_dict({'model': 5, 'k': 1}) == {'model': 5, 'k': 1}
    # a prefix is stripped only when EVERY key has it
    assert U.canonicalize_state_dict({'module.a': 1, 'b': 2}) == {'module.a': 1, 'b': 2}
    assert U.canonicalize_state_dict({}) == {}


def test_seq_len_and_scale_helpers():
    sd = {'body.0.w': 0, 'body.7.x.y': 0, 'bodyguard.9': 0, 'up.0.weight': torch.zeros(64 * 4, 64, 3, 3), 'up.2.weight': torch.zeros(64 * 4, 64, 3, 3)}
    assert U.get_seq_len(sd, 'body') == 8
    assert U.get_seq_len(sd, 'nothing') == 0
    assert U.pixelshuffle_scale(48, 3) == 4 and U.pixelshuffle_scale(12, 3) == 2
    assert U.dysample_scale(2 * 4 * 16) == 4
    assert U.get_pixelshuffle_params(sd, 'up') == (4, 64)
    assert U.get_pixelshuffle_params({}, 'up', default_nf=32) == (1, 32)


class _DummyArch(Architecture):
    def __init__(self, uid='dummy', key='dummy.weight'):
        super().__init__(uid, KeyCondition.has_all(key))

    def load(self, state_dict):
        m = torch.nn.Linear(2, 2)
        return self._enhance_model(m, 1, 2, 3, 'Dummy')


def test_registry_contract():
    reg = Registry()
    a = _DummyArch()
    reg.add(a)
    assert 'dummy' in reg and reg.get('dummy') is a and list(reg) == [a]
    with pytest.raises(KeyError):
        reg.get('unknown')
    with pytest.raises(ArchitectureNotFound):
        reg.load_from_state_dict({'other': 1})
    # strict load_state_dict of the built model is part of the contract (registry.py:113)
    with pytest.raises(RuntimeError):
        reg.load_from_state_dict({'dummy.weight': torch.zeros(1)})
    # re-adding an id replaces in place and keeps detection order
    reg.add(_DummyArch('zzz', 'dummy.weight'))
    b = _DummyArch('dummy', 'other.key')
    reg.add(b)
    assert [x.id for x in reg] == ['dummy', 'zzz'] and reg.get('dummy') is b


def test_public_add_get_and_metadata():
    arch = _DummyArch('test-plugin', 'plugin.only.key')
    resselt_amd.add(arch)
    try:
        assert resselt_amd.get('test-plugin') is arch
        m = arch.load({})
        assert m.parameters_info == ModelMetadata(in_channels=1, out_channels=2, upscale=3, name='Dummy')
    finally:
        del resselt_amd.archs.internal_registry.store['test-plugin']
    assert resselt_amd.get('ESRGAN').id == 'ESRGAN'
    with pytest.raises(KeyError):
        resselt_amd.get('nope')


def test_esrgan_loader_matches_reference_inference():
    meta, _ = load_golden('esrgan_loader')
    arch = resselt_amd.get('ESRGAN')
    for tag, case in meta['cases'].items():
        sd = synth.rrdbnet_state_dict(seed=0, **case['synth'])
        assert arch.detect(sd), tag
        m = arch.load(sd)
        got = vars(m.parameters_info)
        assert got == {k: case['metadata'][k] for k in got}, tag
        assert type(m).__name__ == case['metadata']['cls'] == 'RRDBNet'
        assert len(m.state_dict()) == case['n_params'], tag
        assert m.shuffle_factor == case['shuffle_factor'] and m.scale == case['model_scale'], tag


def test_esrgan_roundtrip_and_new_arch_fix():
    sd = synth.rrdbnet_state_dict(nb=2, seed=1)
    m = resselt_amd.load_from_state_dict({'state_dict': {'module.' + k: v for k, v in sd.items()}})
    out = m.state_dict()
    assert list(out.keys()) == list(sd.keys()) and all(torch.equal(out[k], sd[k]) for k in sd)
    assert m.training  # the reference never calls .eval()
    # documented deviation: new-arch (official Real-ESRGAN) keys load instead of raising 'Missing key(s)'
    m2 = resselt_amd.load_from_state_dict(synth.rrdbnet_state_dict(nb=2, seed=1, new_arch=True))
    assert all(torch.equal(m2.state_dict()[k], sd[k]) for k in sd)
    # fp16 checkpoints are copied into fp32 parameters
    m3 = resselt_amd.load_from_state_dict({k: v.half() for k, v in sd.items()})
    assert all(p.dtype == torch.float32 for p in m3.parameters())
    # strictness
    bad = dict(sd)
    bad.pop('model.0.bias')
    with pytest.raises(RuntimeError, match='Missing key'):
        resselt_amd.load_from_state_dict(bad)


def test_detection_claims_match_reference():
    meta, _ = load_golden('registry_claims')
    ours = {'rrdbnet_old': 'ESRGAN', 'spanplus_ps': 'spanplus', 'spanplus_dys': 'spanplus', 'span': 'SPAN', 'compact': 'Compact', 'swinir': 'SwinIR', 'dat': 'dat', 'spanpp': 'SpanPP', 'hat': 'HAT', 'rtmosr': 'RTMoSR', 'drct': 'DRCT'}
    for tag, uid in ours.items():
        assert meta['claims'][tag] == uid
    built = {
        'rrdbnet_old': synth.rrdbnet_state_dict(nb=1),
        'spanplus_ps': synth.spanplus_state_dict(upsampler='ps'),
        'spanplus_dys': synth.spanplus_state_dict(upsampler='dys'),
        'span': synth.span_state_dict(),
        'compact': synth.compact_state_dict(num_conv=2),
        'swinir': synth.swinir_state_dict(),
        'dat': synth.dat_state_dict(),
        'spanpp': synth.spanpp_state_dict(feature_channels=16, implicit_dim=32, latent_layers=1),
        'hat': synth.hat_state_dict(),
        'rtmosr': synth.rtmosr_state_dict(),
        'drct': synth.drct_state_dict(num_layers=1),
    }
    for tag, sd in built.items():
        hit = [a.id for a in resselt_amd.archs.internal_registry if a.detect(sd)]
        assert hit and hit[0] == ours[tag], tag


def test_file_formats_and_restricted_pickle(tmp_path):
    sd = synth.rrdbnet_state_dict(nb=1, nf=16, seed=2)
    torch.save({'params_ema': sd}, tmp_path / 'a.pth')
    torch.save(sd, tmp_path / 'b.ckpt')
    torch.save(sd, tmp_path / 'c.pt')  # not TorchScript: falls back to the restricted pickle
    import safetensors.torch

    safetensors.torch.save_file({k: v.contiguous() for k, v in sd.items()}, str(tmp_path / 'd.safetensors'))
    for f in ('a.pth', 'b.ckpt', 'c.pt', 'd.safetensors'):
        m = resselt_amd.load_from_file(str(tmp_path / f))
        assert m.parameters_info.name == 'ESRGAN' and torch.equal(m.state_dict()['model.0.weight'], sd['model.0.weight'])
    with pytest.raises(ValueError, match='Unsupported model file extension'):
        resselt_amd.load_from_file(str(tmp_path / 'model.onnx'))

    class Evil:
        def __reduce__(self):
            return (os.system, ('true',))

    torch.save({'x': Evil()}, tmp_path / 'evil.pth')
    with pytest.raises(pickle.UnpicklingError, match='forbidden'):
        resselt_amd.load_from_file(str(tmp_path / 'evil.pth'))
    assert RestrictedUnpickle.__name__ == 'pickle'


def test_torchscript_pt(tmp_path):
    class Tiny(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.dummy = torch.nn.Conv2d(1, 1, 1)

        def forward(self, x):
            return self.dummy(x)

    torch.jit.script(Tiny()).save(str(tmp_path / 'ts.pt'))
    with pytest.raises(ArchitectureNotFound):  # loads as TorchScript, but no architecture claims its keys
        resselt_amd.load_from_file(str(tmp_path / 'ts.pt'))


def test_engine_module_copies_and_tracks_in_place_parameter_edits():
    """nn.Module conveniences of the returned object: deepcopy / pickle work (the plan cache and the forward lock are
    runtime state, not part of the module's value), and an in-place parameter edit invalidates the packed weights."""
    import copy

    m = resselt_amd.load_from_state_dict(dict(synth.rrdbnet_state_dict(nb=1, nf=16, seed=2)))
    m2 = copy.deepcopy(m)
    assert m2 is not m and m2.parameters_info == m.parameters_info
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))
    m3 = pickle.loads(pickle.dumps(m))
    assert list(m3.state_dict()) == list(m.state_dict())
    v0 = tuple(p._version for p in m.parameters())
    with torch.no_grad():
        next(m.parameters()).mul_(2.0)
    assert tuple(p._version for p in m.parameters()) != v0  # what EngineModule._weights keys the packed blobs on
    m.invalidate()  # the explicit form, for edits through `.data` (which PyTorch does not version)
