"""Pin the CPU oracle (oracle/) against vectors produced by the real reference (tools/gen_golden.py).

These are the only 'reference ran here' anchors: everything the GPU tests compare against is this oracle.
Tolerance 1e-5 relative to max|y| (fp32 CPU; op order differs slightly from the reference's modules).
"""

import pytest
import torch

from helpers import golden_names, load_golden, oracle_forward, synth_state_dict

E2E = golden_names('rrdbnet_') + golden_names('spanplus_') + golden_names('span_') + golden_names('swinir_') + golden_names('compact_') + golden_names('dat_') + golden_names('spanpp_') + golden_names('hat_') + golden_names('rtmosr_') + golden_names('drct_')


@pytest.mark.parametrize('name', E2E)
def test_oracle_matches_reference_end_to_end(name):
    meta, arr = load_golden(name)
    sd = synth_state_dict(meta)
    with torch.no_grad():
        y = oracle_forward(meta, sd, arr['x'])
    ref = arr['y']
    assert y.shape == ref.shape
    tol = 1e-5 * max(ref.abs().max().item(), 1.0)
    assert (y - ref).abs().max().item() <= tol


def test_oracle_rrdb_blocks():
    from oracle.rrdbnet import _conv, _lrelu, rdb_forward, rrdb_forward
    import torch.nn.functional as F

    meta, arr = load_golden('blocks_rrdb')
    sd = synth_state_dict(meta)
    x = arr['x']
    assert (rdb_forward(sd, 'model.1.sub.0.RDB1', x) - arr['rdb']).abs().max() <= 1e-5
    assert (rrdb_forward(sd, 'model.1.sub.0', x) - arr['rrdb']).abs().max() <= 1e-5
    up = _lrelu(_conv(sd, 'model.3', F.interpolate(x, scale_factor=2, mode='nearest')))
    assert (up - arr['upconv']).abs().max() <= 1e-5


def test_oracle_span_blocks():
    import torch.nn.functional as F

    from oracle.span import conv3xc_fold, spab

    meta, arr = load_golden('blocks_span')
    sd = synth_state_dict(meta)
    w, b = conv3xc_fold(sd, 'feats.1.block_1.c1_r')
    assert (w - arr['fold_w']).abs().max() <= 1e-6
    assert (b - arr['fold_b']).abs().max() <= 1e-6
    # NB the reference block under test was built with end=True
    out, out1 = spab(sd, 'feats.1.block_1', arr['x'], F.mish)
    assert (out - arr['spab_out']).abs().max() <= 1e-5
    assert (out1 - arr['spab_out1']).abs().max() <= 1e-5


def test_oracle_swin_blocks():
    from oracle.swinir import swin_block

    meta, arr = load_golden('blocks_swin')
    sd = synth_state_dict(meta)
    t = arr['x']
    for j, shift in ((0, 0), (1, 4)):
        t = swin_block(sd, f'layers.0.residual_group.blocks.{j}', t, 16, 24, 8, shift, 8)
        assert (t - arr[f'block{j}']).abs().max() <= 2e-5, j
