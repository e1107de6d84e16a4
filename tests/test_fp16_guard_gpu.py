"""Run-time guard of the fp16 precision policies (round 4) and the status words behind `rsa_check_status`.

The reference returns an fp32 module that works for any checkpoint (resselt/registry.py:106-116).  The engine's default policy runs most
layers on fp16 planes; an activation beyond +-65504 becomes an infinity there, travels with the residual stream to the end of the network
and is found by `rsa_check_finite` behind every forward: `precision = 'auto'` then falls back to three bf16 products with a RuntimeWarning,
an explicitly requested fp16 policy raises `Fp16RangeError`.
"""

import warnings

import pytest
import torch

import resselt_amd
from resselt_amd.engine import lib as L
from resselt_amd.engine import ops
from resselt_amd.utils import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16, torch.bfloat16])
def test_check_finite_kernel(device, dtype):
    stream = ops.current_stream_ptr(device)
    g = torch.Generator().manual_seed(1)
    n = 1_000_003  # not a multiple of the vector width: the tail elements are checked too
    x = (torch.rand(n, generator=g) * 2e4 - 1e4).to(dtype).to(device)
    L.check_finite(x, stream)
    torch.cuda.synchronize()
    L.check_status('finite data')
    for pos, val in ((0, float('inf')), (n // 2 + 1, float('-inf')), (n - 1, float('nan')), (n - 3, float('inf'))):
        y = x.clone()
        y[pos] = val
        L.check_finite(y, stream)
        torch.cuda.synchronize()
        with pytest.raises(L.Fp16RangeError, match='fp16 range'):
            L.check_status('non-finite data')
        L.check_status('cleared')  # reported once
    # the largest finite values of each format are fine
    big = torch.tensor([torch.finfo(dtype).max, -torch.finfo(dtype).max] * 8, dtype=dtype, device=device)
    L.check_finite(big, stream)
    torch.cuda.synchronize()
    L.check_status('largest finite values')


def _overflowing_checkpoint():
    """Every weight inside the fp16 range, but the first convolution amplifies the image by 3e5: the feature map (|v| up to ~4e5) does not
    fit the fp16 planes the residual dense blocks read."""
    sd = synth.rrdbnet_state_dict(nb=2, seed=9)
    sd['model.0.weight'] = sd['model.0.weight'] * 3e5
    sd['model.0.bias'] = sd['model.0.bias'] * 3e5
    return sd


def test_auto_falls_back_when_activations_leave_the_fp16_range(device):
    from oracle.rrdbnet import rrdbnet_forward

    sd = _overflowing_checkpoint()
    assert max(float(v.abs().max()) for k, v in sd.items() if '.RDB' in k) < 6e4
    x = synth.synth_input((1, 3, 40, 56), seed=9)
    with torch.no_grad():
        ref = rrdbnet_forward(sd, x)
    model = resselt_amd.load_from_state_dict(dict(sd)).to(device)
    assert model.resolved_precision() == 'mixed'
    with pytest.warns(RuntimeWarning, match='falls back to "bf16x3"'):
        y = model(x.to(device))
    model.sync_check()
    assert model.resolved_precision() == 'bf16x3'
    scale = ref.abs().max().item()
    assert torch.isfinite(y).all()
    assert (y.cpu() - ref).abs().max().item() <= 2e-4 * scale
    # later forwards (other shapes too) stay in the conservative mode without another attempt
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        y2 = model(x.to(device)[:, :, :32, :40])
    model.sync_check()
    assert torch.isfinite(y2).all()
    # the 8-bit path has no float output to scan: the probe inside the network finds the overflow all the same
    model2 = resselt_amd.load_from_state_dict(dict(sd)).to(device)
    img = (x * 255).round().to(torch.uint8).permute(0, 2, 3, 1).contiguous().to(device)
    with pytest.warns(RuntimeWarning, match='falls back'):
        model2(img)
    assert model2.resolved_precision() == 'bf16x3'
    # new weights give the fp16 policy another chance
    model.load_state_dict(synth.rrdbnet_state_dict(nb=2, seed=9))
    assert model.resolved_precision() == 'mixed'
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        model(x.to(device))
    model.sync_check()


def test_explicit_fp16_policy_raises(device):
    sd = _overflowing_checkpoint()
    model = resselt_amd.load_from_state_dict(dict(sd)).to(device)
    model.precision = 'mixed'
    x = synth.synth_input((1, 3, 40, 56), seed=9).to(device)
    model(x)  # never synchronises; the result is not to be trusted until ...
    with pytest.raises(L.Fp16RangeError):
        model.sync_check()
    model.sync_check()  # reported once
    # ... and a forward that finds an earlier forward's report raises it too
    model(x)
    torch.cuda.synchronize()
    with pytest.raises(L.Fp16RangeError):
        model(x)
    torch.cuda.synchronize()
    L.check_status('drained')  # the raising forward consumed the report and launched no new probe


def test_module_reports_a_failed_handoff_exactly_once(device):
    """Advisor finding of round 3: a timed-out ring hand-off must surface as ONE exception, and the module must work again afterwards
    (the refusal of rsa_conv2d_list while a failure is pending must not leave the word set)."""
    sd = synth.rrdbnet_state_dict(nb=1, seed=4)
    model = resselt_amd.load_from_state_dict(dict(sd)).to(device)
    x = synth.synth_input((1, 3, 64, 96), seed=4).to(device)
    good = model(x).clone()
    model.sync_check()
    raised = 0
    try:
        L.set_ring_spin_limit(1)
        try:
            model(x)  # its own status read may already see the first failures ...
        except RuntimeError as e:
            assert 'hand-off' in str(e)
            raised += 1
        torch.cuda.synchronize()
    finally:
        L.set_ring_spin_limit(1 << 18)
    # ... the rest is pending now: the next forward either finds the launch list refusing to run or reads the word after its launches
    try:
        model(x)
    except RuntimeError:
        raised += 1
    torch.cuda.synchronize()
    try:
        L.check_status('drain')
    except RuntimeError:
        raised += 1
    assert raised >= 1
    L.ring_aborts()  # clear the debug counters
    for _ in range(2):  # and the module is usable again: nothing stays set
        again = model(x)
        model.sync_check()
        assert torch.equal(again, good)


def test_graph_replay_checks_the_status_words(device):
    sd = synth.rrdbnet_state_dict(nb=1, seed=4)
    model = resselt_amd.load_from_state_dict(dict(sd)).to(device)
    model.use_graph = True
    x = synth.synth_input((1, 3, 48, 64), seed=4).to(device)
    y0 = model(x).clone()
    y1 = model(x).clone()
    model.sync_check()
    assert torch.equal(y0, y1)
    try:
        L.set_ring_spin_limit(1)
        model(x)  # replay: the kernels read the spin limit from the launch arguments captured earlier -> still fine
        model.sync_check()
    finally:
        L.set_ring_spin_limit(1 << 18)
