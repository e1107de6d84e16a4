"""Does a row band whose working set fits the 256 MB Infinity Cache run faster per pixel than the whole frame?

RRDBNet-23 x4 on Hx1920 inputs for several H; prints ms and ns per input pixel.  (Probe behind the band-skew idea in DESIGN.md.)
"""

import sys
import time

import torch

import resselt_amd
from resselt_amd.utils import synth


def main():
    dev = torch.device('cuda:0')
    sd = synth.rrdbnet_state_dict(seed=7)
    m = resselt_amd.load_from_state_dict(dict(sd)).to(dev)
    prec = sys.argv[1] if len(sys.argv) > 1 else 'bf16x3'
    m.precision = prec
    for H in (32, 64, 128, 256, 512, 1080):
        x = torch.rand(1, 3, H, 1920, device=dev)
        for _ in range(2):
            m(x)
        torch.cuda.synchronize()
        n = 3 if H >= 512 else 6
        t0 = time.perf_counter()
        for _ in range(n):
            m(x)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        print(f'{prec} H={H:5d} tiles={((H + 15) // 16) * 60:5d} {ms:8.2f} ms  {ms * 1e6 / (H * 1920):7.2f} ns/pixel', flush=True)


if __name__ == '__main__':
    main()
