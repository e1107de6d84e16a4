"""GPU parity of the SPAN / SPANPlus engines (incl. the DySample kernel) against reference vectors and the oracle.

Tolerance: max-abs <= 2e-4 * max(1, max|y|) (SPAN multiplies its input by 255, so outputs are large), in the default precision ('auto' =
one fp16 product per layer for SPAN / SPANPlus / Compact) and in the conservative 'bf16x3' mode alike.
"""

import pytest
import torch

import resselt_amd
from helpers import golden_names, load_golden, oracle_forward, synth_state_dict
from resselt_amd.utils import synth

pytestmark = pytest.mark.gpu


def _tol(ref, rel=2e-4):
    return rel * max(1.0, ref.abs().max().item())


@pytest.fixture(autouse=True)
def _no_failed_hand_offs():
    yield
    from resselt_amd.engine import lib as L

    if torch.cuda.is_available():
        torch.cuda.synchronize()
        L.check_status('end of test')


@pytest.mark.parametrize('precision', ['auto', 'bf16x3'])
@pytest.mark.parametrize('name', golden_names('spanplus_') + golden_names('span_'))
def test_span_family_matches_reference_vectors(device, name, precision):
    meta, arr = load_golden(name)
    sd = synth_state_dict(meta)
    m = resselt_amd.load_from_state_dict(dict(sd)).to(device)
    assert m.precision == 'auto' and m.resolved_precision() == ('mixed' if name.startswith('spanplus_') else 'bf16x3')  # (SPAN: see archs/span/arch.py)
    m.precision = precision
    assert vars(m.parameters_info) == {k: meta['metadata'][k] for k in ('in_channels', 'out_channels', 'upscale', 'name')}
    y = m(arr['x'].to(device))
    torch.cuda.synchronize()
    assert y.shape == arr['y'].shape
    err = (y.cpu() - arr['y']).abs().max().item()
    print(f'{name} {precision}: max-abs {err:.3e} (|y|max {arr["y"].abs().max():.3f})')
    assert err <= _tol(arr['y']), f'{name} {precision}: max-abs {err:.3e} (|y|max {arr["y"].abs().max():.3f})'


@pytest.mark.parametrize('name', golden_names('spanpp_'))
def test_spanpp_matches_reference_vectors(device, name):
    """SpanPP (RepConv folds + implicit-grid upsampler kernels generated at pack time) against the reference in eval mode, every head scale."""
    meta, arr = load_golden(name)
    sd = synth_state_dict(meta)
    m = resselt_amd.load_from_state_dict(dict(sd)).to(device)
    assert vars(m.parameters_info) == {k: meta['metadata'][k] for k in ('in_channels', 'out_channels', 'upscale', 'name')}
    x = arr['x'].to(device)
    y = m(x) if meta['scale'] is None else m(x, meta['scale'])
    torch.cuda.synchronize()
    assert y.shape == arr['y'].shape
    err = (y.cpu() - arr['y']).abs().max().item()
    assert err <= _tol(arr['y']), f'{name}: max-abs {err:.3e} (|y|max {arr["y"].abs().max():.3f})'
    with pytest.raises(KeyError):
        m(x, 7)  # a scale outside scale_list, as in the reference
    y2 = m(x, 2)  # switching heads rebuilds the plan
    assert y2.shape[-1] == 2 * x.shape[-1]


@pytest.mark.parametrize('ups', ['ps', 'dys'])
def test_spanplus_x4_fp16_batch_vs_oracle(device, ups):
    """Shape of BASELINE config 3 (SPANPlus 4x, fp16 tensors, batched tiles), reduced to a size the oracle finishes quickly."""
    sd = synth.spanplus_state_dict(upscale=4, upsampler=ups, seed=77)
    x = synth.synth_input((3, 3, 96, 80), seed=77)
    meta = dict(arch='spanplus')
    with torch.no_grad():
        ref = oracle_forward(meta, sd, x)
    m = resselt_amd.load_from_state_dict(dict(sd)).to(device)
    y32 = m(x.to(device))
    print(f'spanplus {ups} x4 auto: {(y32.cpu() - ref).abs().max().item():.3e}')
    assert (y32.cpu() - ref).abs().max().item() <= _tol(ref)
    m.precision = 'bf16x3'
    assert (m(x.to(device)).cpu() - ref).abs().max().item() <= _tol(ref)
    m.precision = 'auto'
    y16 = m(x.half().to(device))
    assert y16.dtype == torch.float16 and y16.shape == ref.shape
    with torch.no_grad():
        ref16 = oracle_forward(meta, sd, x.half().float())
    assert (y16.float().cpu() - ref16).abs().max().item() <= 2e-3 * max(1.0, ref16.abs().max().item())
    # plain bf16 operands: survey-measured 4e-4 on default-init weights; allow 5e-3
    m.precision = 'bf16'
    assert (m(x.to(device)).cpu() - ref).abs().max().item() <= 5e-3 * max(1.0, ref.abs().max().item())


def test_span_eval_and_train_mode_agree(device):
    """The reference folds Conv3XC only in eval mode (SPANPlus) / always (SPAN); both paths give the same function."""
    sd = synth.spanplus_state_dict(upscale=2, upsampler='ps', seed=5)
    m = resselt_amd.load_from_state_dict(dict(sd)).to(device)
    x = synth.synth_input((1, 3, 40, 33), seed=5).to(device)
    assert m.training
    y_train = m(x)
    y_eval = m.eval()(x)
    assert torch.equal(y_train, y_eval)


@pytest.mark.parametrize('precision', ['auto', 'bf16x3'])
@pytest.mark.parametrize('name', golden_names('compact_'))
def test_compact_matches_reference_vectors(device, name, precision):
    """SRVGGNetCompact: PReLU epilogue, PixelShuffle + nearest base image in the final store (first "next" row of SURVEY §8f)."""
    meta, arr = load_golden(name)
    sd = synth_state_dict(meta)
    m = resselt_amd.load_from_state_dict(dict(sd)).to(device)
    m.precision = precision
    assert vars(m.parameters_info) == {k: meta['metadata'][k] for k in ('in_channels', 'out_channels', 'upscale', 'name')}
    y = m(arr['x'].to(device))
    assert y.shape == arr['y'].shape
    print(f'{name} {precision}: max-abs {(y.cpu() - arr["y"]).abs().max().item():.3e} (|y|max {arr["y"].abs().max():.3f})')
    assert (y.cpu() - arr['y']).abs().max().item() <= _tol(arr['y'])
    yh = m(arr['x'].half().to(device))
    assert yh.dtype == torch.float16 and (yh.float().cpu() - arr['y']).abs().max().item() <= 4e-3 * max(1.0, arr['y'].abs().max().item())


@pytest.mark.parametrize('name', golden_names('rtmosr_'))
def test_rtmosr_matches_reference_vectors(device, name):
    """RTMoSR (RMSNorm, gated block with unshuffle / max-pool / depthwise 5x5 / SE branch, nearest-upsampled base image) vs the reference in eval mode."""
    meta, arr = load_golden(name)
    sd = synth_state_dict(meta)
    m = resselt_amd.load_from_state_dict(dict(sd)).to(device)
    assert vars(m.parameters_info) == {k: meta['metadata'][k] for k in ('in_channels', 'out_channels', 'upscale', 'name')}
    y = m(arr['x'].to(device))
    torch.cuda.synchronize()
    assert y.shape == arr['y'].shape
    err = (y.cpu() - arr['y']).abs().max().item()
    print(f'{name}: max-abs {err:.3e}')
    assert err <= _tol(arr['y']), f'{name}: max-abs {err:.3e} (|y|max {arr["y"].abs().max():.3f})'
    yh = m(arr['x'].to(device).half())
    assert yh.dtype == torch.float16 and (yh.float().cpu() - arr['y']).abs().max().item() <= 10 * _tol(arr['y'])
