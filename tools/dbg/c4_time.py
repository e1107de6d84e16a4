import os, sys, statistics, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from resselt_amd.engine import lib as L
if os.environ.get('RSA_LIB'):
    _p = os.path.abspath(os.environ['RSA_LIB'])
    L.lib_path = lambda: _p  # an experiment build (tools/variant.sh)
import resselt_amd
from resselt_amd.utils import synth
dev = torch.device('cuda:0')
sd = synth.swinir_state_dict(embed_dim=240, depths=[6] * 9, num_heads=[8] * 9, upscale=4, upsampler='nearest+conv', resi='3conv')
m = resselt_amd.load_from_state_dict(dict(sd)).to(dev)
x = synth.synth_input((1, 3, 1024, 1024), seed=0).to(dev).bfloat16()
for prec in sys.argv[1:] or ['bf16']:
    m.precision = prec
    for _ in range(2): m(x)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); m(x); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(os.environ.get('RSA_LIB', 'product'), prec, f'{statistics.median(ts)*1e3:.2f} ms', flush=True)
