#!/usr/bin/env python3
"""Times the two fused Swin block kernels (csrc/swin_block.hip) alone at a model's shape (default SwinIR-L at 1024x1024: 1 Mi tokens,
C 240, 8 heads, window 8, hidden 480) and prices them against the MFMA work they issue.
usage: swin_block_bench.py [H W [C heads hidden]]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import torch  # noqa: E402

from resselt_amd.archs.swinir.arch import bias_fragments16, regroup_proj, regroup_qkv, relative_position_index  # noqa: E402
from resselt_amd.engine import lib as L  # noqa: E402
from resselt_amd.engine import ops, tensors  # noqa: E402

a = [int(v) for v in sys.argv[1:]]
H, W = (a[0], a[1]) if len(a) >= 2 else (1024, 1024)
C_, heads, hidden = (a[2], a[3], a[4]) if len(a) >= 5 else (240, 8, 480)
win = 8
dev = torch.device('cuda:0')
if os.environ.get('RSA_LIB'):
    _p = os.path.abspath(os.environ['RSA_LIB'])
    L.lib_path = lambda: _p  # an experiment build (tools/variant.sh)
g = torch.Generator().manual_seed(0)
rnd = lambda *s, k=1.0: (torch.rand(s, generator=g) * 2 - 1) * k  # noqa: E731
lin = lambda w, b, cp=None: ops.ConvWeights.from_oihw(w[:, :, None, None], b, 3, cin_planes=cp, device=dev)  # noqa: E731
x = tensors.nchw_to_f32map(rnd(1, C_, H, W, k=2.0).to(dev))
out = torch.empty_like(x)
wq, bq = regroup_qkv(rnd(3 * C_, C_, k=2 / C_**0.5), rnd(3 * C_, k=0.2), heads)
qkv, proj = lin(wq, bq), lin(regroup_proj(rnd(C_, C_, k=1 / C_**0.5), heads), rnd(C_, k=0.2), heads * 4)
fc1, fc2 = lin(rnd(hidden, C_, k=1.5 / C_**0.5), rnd(hidden, k=0.2)), lin(rnd(C_, hidden, k=1 / hidden**0.5), rnd(C_, k=0.2))
frag = bias_fragments16(rnd((2 * win - 1) ** 2, heads), relative_position_index(win), win).to(dev)
ga, be = (1 + rnd(C_, k=0.3)).to(dev), rnd(C_, k=0.3).to(dev)
lib = L.load()
st = C.c_void_p(ops.current_stream_ptr(dev))


def attn(shift):
    ap = L.SwinAttnBlockParams()
    ap.batch, ap.H, ap.W, ap.C, ap.heads, ap.window, ap.shift, ap.products, ap.eps = 1, H, W, C_, heads, win, shift, 3, 1e-5
    ap.x, ap.gamma, ap.beta = x.data_ptr(), ga.data_ptr(), be.data_ptr()
    ap.wqkv, ap.bqkv, ap.bias_frag16 = qkv.packed_for(0).data_ptr(), qkv.bias.data_ptr(), frag.data_ptr()
    ap.wproj, ap.bproj, ap.out = proj.packed_for(0).data_ptr(), proj.bias.data_ptr(), out.data_ptr()
    return lambda: L.check(lib.rsa_swin_attn_block(C.byref(ap), st), 'attn')


def mlp():
    mp = L.SwinMlpBlockParams()
    mp.batch, mp.H, mp.W, mp.C, mp.hidden, mp.products, mp.eps = 1, H, W, C_, hidden, 3, 1e-5
    mp.x, mp.gamma, mp.beta = x.data_ptr(), ga.data_ptr(), be.data_ptr()
    mp.w1, mp.b1, mp.w2, mp.b2 = fc1.packed_for(0).data_ptr(), fc1.bias.data_ptr(), fc2.packed_for(0).data_ptr(), fc2.bias.data_ptr()
    mp.out = out.data_ptr()
    return lambda: L.check(lib.rsa_swin_mlp_block(C.byref(mp), st), 'mlp')


def whole(shift):
    bp = L.SwinBlockParams()
    bp.batch, bp.H, bp.W, bp.C, bp.heads, bp.window, bp.shift, bp.hidden, bp.products, bp.eps = 1, H, W, C_, heads, win, shift, hidden, 3, 1e-5
    bp.x, bp.gamma1, bp.beta1, bp.gamma2, bp.beta2 = x.data_ptr(), ga.data_ptr(), be.data_ptr(), ga.data_ptr(), be.data_ptr()
    bp.wqkv, bp.bqkv, bp.bias_frag16 = qkv.packed_for(0).data_ptr(), qkv.bias.data_ptr(), frag.data_ptr()
    bp.wproj, bp.bproj = proj.packed_for(0).data_ptr(), proj.bias.data_ptr()
    bp.w1, bp.b1, bp.w2, bp.b2 = fc1.packed_for(0).data_ptr(), fc1.bias.data_ptr(), fc2.packed_for(0).data_ptr(), fc2.bias.data_ptr()
    bp.out = out.data_ptr()
    return lambda: L.check(lib.rsa_swin_block(C.byref(bp), st), 'block')


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


tok = H * W
hd = C_ // heads
flop_attn = 2 * tok * (3 * C_ * C_ + C_ * C_ + 2 * 64 * C_)
flop_mlp = 2 * tok * 2 * C_ * hidden
for name, fn, fl in (('attn shift 0', attn(0), flop_attn), ('attn shift 4', attn(4), flop_attn), ('mlp', mlp(), flop_mlp),
                     ('whole block shift 0', whole(0), flop_attn + flop_mlp), ('whole block shift 4', whole(4), flop_attn + flop_mlp)):
    ms = timed(fn)
    print(f'{name}: {ms:.3f} ms  {fl / ms / 1e9:.0f} TF algorithmic, {3 * fl / ms / 1e9:.0f} TF issued (bf16x3)', flush=True)
