#!/usr/bin/env python3
"""Cold-start cost on the GPU (the reference's cold path is registry.py:79-116): checkpoint file -> first pixels.

Per model, one JSON line: seconds for
  load      resselt_amd.load_from_file(<.safetensors written from the synthetic checkpoint>) + .to(device)
  pack      weight packing (`_pack`: OIHW f32 -> MFMA fragment blobs through rsa_pack_weights, the fp16 range check, pack-time folds)
  plan      the first forward minus a steady-state forward (plan build: buffers + launch descriptors, lazy blobs in the schedules' layouts)
  forward   a steady-state forward
and `pack_share` = pack / (load + pack + plan + forward): what a packed-weight cache keyed by the checkpoint hash could save at most;
`sha256_of_checkpoint_s` is what computing that key costs (hashlib, one core).
usage: cold_start.py [rrdbnet23 swinir_L hat spanplus]
"""

import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import resselt_amd  # noqa: E402
from resselt_amd.utils import synth  # noqa: E402

dev = torch.device('cuda:0')
torch.zeros(1, device=dev)
torch.cuda.synchronize()
from safetensors.torch import save_file  # noqa: E402

MODELS = {
    'rrdbnet23': (lambda: synth.rrdbnet_state_dict(nb=23), (1, 3, 256, 256)),
    'swinir_L': (lambda: synth.swinir_state_dict(embed_dim=240, depths=[6] * 9, num_heads=[8] * 9, upscale=4, upsampler='nearest+conv', resi='3conv'), (1, 3, 256, 256)),
    'hat': (lambda: synth.hat_state_dict(embed_dim=180, depths=(6,) * 6, num_heads=(6,) * 6, window=16, upscale=4), (1, 3, 128, 128)),
    'spanplus': (lambda: synth.spanplus_state_dict(upscale=4, upsampler='ps'), (1, 3, 256, 256)),
}
for name in sys.argv[1:] or list(MODELS):
    make, shape = MODELS[name]
    sd = {k: v.contiguous() for k, v in make().items()}
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, f'{name}.safetensors')
        save_file(sd, path)
        size_mb = os.path.getsize(path) / 1e6
        import hashlib

        th = time.perf_counter()
        with open(path, 'rb') as fh:
            hashlib.sha256(fh.read()).hexdigest()  # what a packed-weight cache keyed by the checkpoint's hash would have to pay first
        hash_s = time.perf_counter() - th
        t0 = time.perf_counter()
        m = resselt_amd.load_from_file(path).to(dev)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
    m._weights(dev)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    x = synth.synth_input(shape).to(dev)
    m(x)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    m(x)
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    load, pack, first, fwd = t1 - t0, t2 - t1, t3 - t2, t4 - t3
    total = load + pack + first
    print(json.dumps({'model': name, 'checkpoint_MB': round(size_mb, 1), 'input': list(shape), 'precision': m.resolved_precision(), 'load_s': round(load, 3), 'pack_s': round(pack, 3), 'sha256_of_checkpoint_s': round(hash_s, 3),
                      'plan_s': round(first - fwd, 3), 'forward_s': round(fwd, 4), 'cold_total_s': round(total, 3), 'pack_share': round(pack / total, 3),
                      'launches': m.launches_per_forward()}), flush=True)  # fmt: skip
