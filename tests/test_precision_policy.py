"""The 'auto' precision policy of RRDBNet, emulated on the CPU (tests/precision_policy.py patches the oracle's convolution to round
its operands the way the engine's kernels do): which layers may run ONE fp16 product and which need three products, pinned BEFORE any
kernel runs.  north_star's bar is 1e-3 max-abs against CPU fp32; the engine's bar for its default mode is 2e-4.

Measured here (RRDBNet-23, 128 x 128 crop of the bench frame / heavy-tailed checkpoint):
    mixed table (RDB convolutions fp16 x 1, trunk conv fp16 x 3, conv_first / upconvs / HR / last bf16 x 3)   1.2e-4 / 8.9e-5
    one fp16 product everywhere                                                                                1.9e-3 / 1.8e-3   (over)
    bf16 x 3 everywhere (the conservative mode)                                                                2.6e-5 / 3.1e-5
    RDB convolutions in one bf16 product, head / tail bf16 x 3                                                 6.8e-4 / 6.0e-4
"""

import torch

import precision_policy as P
from oracle.rrdbnet import rrdbnet_forward
from resselt_amd.utils import synth


def _errors(sd, x, cases):
    with torch.no_grad():
        ref = rrdbnet_forward(sd, x)
        out = {}
        for name, (policy, stream) in cases.items():
            with P.emulate(policy, stream):
                out[name] = (rrdbnet_forward(sd, x) - ref).abs().max().item()
    return out


def test_the_engine_table_is_the_emulated_table():
    """RRDBNet.layer_policy (what _pack uses) and the emulator's table name the same arithmetic for every convolution of the network."""
    from resselt_amd.archs.esrgan.arch import RRDBNet
    from resselt_amd.engine.tensors import PF_BF16, PF_F16

    names = {(1, PF_F16): 'fp16', (3, PF_F16): 'fp16x3', (3, PF_BF16): 'bf16x3'}
    sd = synth.rrdbnet_state_dict(nb=2, seed=0)
    for key in sd:
        if key.endswith('.weight'):
            base = key[: -len('.weight')]
            assert names[RRDBNet.layer_policy(base)] == P.rrdbnet_auto(base), base
    m = RRDBNet()
    assert m.precision == 'auto' and m.resolved_precision() == 'mixed'
    assert RRDBNet(plus=True).resolved_precision() == 'bf16x3' and RRDBNet(num_filters=32).resolved_precision() == 'bf16x3'


def test_mixed_policy_meets_the_bar_on_rrdbnet23():
    torch.set_num_threads(8)
    sd = synth.rrdbnet_state_dict(nb=23, seed=0)
    x = synth.synth_input((1, 3, 1080, 1920), seed=0)[:, :, 400:528, 800:928].contiguous()
    e = _errors(sd, x, {'mixed': (P.rrdbnet_auto, 'lo8'), 'mixed_fp16_lo': (P.rrdbnet_auto, torch.float16), 'fp16': (P.uniform('fp16'), None),
                        'bf16x3': (P.uniform('bf16x3'), torch.bfloat16)})  # fmt: skip
    print(e)
    # round 4: the stream's lo halves are 8-bit codes (hi fp16 + 8 more mantissa bits: 19); round 3 kept fp16 lo halves (22 bits): same error
    assert e['mixed'] <= 2e-4 and e['mixed_fp16_lo'] <= 2e-4 and e['bf16x3'] <= 1e-4
    assert e['fp16'] > 1e-3  # one product everywhere does NOT meet north_star's bar: the reason the policy is per layer


def test_mixed_policy_on_heavy_tailed_weights():
    torch.set_num_threads(8)
    sd = synth.rrdbnet_heavy_tailed_state_dict(nb=23, seed=4)
    x = synth.synth_input((1, 3, 96, 112), seed=4)
    e = _errors(sd, x, {'mixed': (P.rrdbnet_auto, 'lo8'), 'mixed_fp16_lo': (P.rrdbnet_auto, torch.float16)})
    print(e)
    assert e['mixed'] <= 2e-4 and e['mixed_fp16_lo'] <= 2e-4


def test_span_mixed_policy_on_spanplus_x4():
    """SPANPlus x4 (BASELINE config 3's network): which layers spend the error budget.  One fp16 product everywhere: 2.1e-4 (f32 output
    tensors), 0.7-1.6e-4 of it from each of conv_cat, the upsampler head and the first convolution; with those three in three fp16 products: 4e-6."""
    import oracle.span as O
    from resselt_amd.engine.spanblocks import span_layer_policy
    from resselt_amd.engine.tensors import PF_F16

    assert span_layer_policy('feats.1.block_1.c1_r', True) == (1, PF_F16) and span_layer_policy('feats.1.conv_cat', False) == (3, PF_F16)
    assert span_layer_policy('feats.0', True) == (3, PF_F16) and span_layer_policy('conv_1', True) == (3, PF_F16)
    torch.set_num_threads(8)
    sd = synth.spanplus_state_dict(upscale=4, upsampler='ps', seed=0)
    x = synth.synth_input((2, 3, 96, 96), seed=0).half().float()
    with torch.no_grad():
        ref = O.spanplus_forward(sd, x)
        with P.emulate_span(P.span_mixed, torch.float16):
            mixed = (O.spanplus_forward(sd, x) - ref).abs().max().item()
        with P.emulate_span(P.uniform('fp16'), torch.float16):
            one = (O.spanplus_forward(sd, x) - ref).abs().max().item()
    print(mixed, one)
    assert mixed <= 2e-5 and one > 1e-4


def test_transformer_policies_name_the_one_product_layers():
    """DRCT / HAT / DAT under 'mixed' (the default): which layers run ONE fp16 product.  The GPU suite pins the resulting error on the
    reference vectors (<= 1e-4 * max(1, |y|)); this test pins the tables themselves."""
    from resselt_amd.archs.dat.arch import DAT
    from resselt_amd.archs.drct.arch import DRCT
    from resselt_amd.archs.hat.arch import HAT
    from resselt_amd.engine.tensors import PF_BF16, PF_F16

    one, three = (1, PF_F16), (3, PF_BF16)
    b = 'layers.0.swin3'
    assert [DRCT.layer_policy(f'{b}.{n}') for n in ('attn.qkv', 'attn.proj', 'mlp.fc1', 'mlp.fc2')] == [one] * 4
    assert DRCT.layer_policy('layers.2.adjust4') == one and DRCT.layer_policy('conv_after_body') == three and DRCT.layer_policy('conv_last') == three
    h = 'layers.1.residual_group.blocks.2'
    assert [HAT.layer_policy(f'{h}.{n}') for n in ('attn.qkv', 'attn.proj', 'conv_block.cab.0', 'conv_block.cab.2', 'mlp.fc1', 'mlp.fc2')] == [one] * 6
    assert HAT.layer_policy('layers.1.residual_group.overlap_attn.qkv') == one and HAT.layer_policy('layers.1.conv') == three
    d = 'layers.0.blocks.1'
    assert DAT.layer_policy(f'{d}.attn.qkv') == one and DAT.layer_policy(f'{d}.ffn.fc1') == one
    # round 4: DAT's whole transformer body is on fp16 planes (proj and fc2 too); the 3x3 convolutions keep three products
    assert DAT.layer_policy(f'{d}.attn.proj') == one and DAT.layer_policy(f'{d}.ffn.fc2') == one
    assert DAT.layer_policy('conv_first') == three and DAT.layer_policy('layers.0.conv') == three and DAT.layer_policy('conv_after_body') == three
    sd = synth.drct_state_dict(num_layers=1, embed_dim=60, gc=16, window=8)
    import resselt_amd

    m = resselt_amd.load_from_state_dict(dict(sd))
    assert m.precision == 'auto' and m.resolved_precision() == 'mixed'
    m.precision = 'bf16x3'
    assert m.resolved_precision() == 'bf16x3'
