#!/usr/bin/env python3
"""Rehearse the multi-GPU code path on ONE GPU: 2 ranks (gloo rendezvous, both on cuda:0) run TileParallel on the engine and every
rank must end with the same image as the single-process full-frame forward (interior of the halo-padded tiles is exact to ~1e-6).

usage: python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 tools/tile_parallel_gpu_check.py
"""

import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import resselt_amd  # noqa: E402
from resselt_amd.tiling import TileParallel  # noqa: E402
from resselt_amd.utils import synth  # noqa: E402


def main():
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device('cuda', 0)
    sd = synth.rrdbnet_state_dict(nb=2, seed=5)
    model = resselt_amd.load_from_state_dict(dict(sd)).to(dev)
    x = synth.synth_input((1, 3, 96, 160), seed=5).to(dev)
    full = model(x)
    for halo in (32, 0):
        y = TileParallel(model, scale=4, halo=halo, grid=(1, world))(x)
        assert y.shape == full.shape and y.device == full.device
        err = (y - full).abs().max().item()
        print(f'rank {rank}/{world} halo {halo}: max-abs vs full frame {err:.3e}', flush=True)
        if halo == 32:
            assert err <= 1e-4, err
    ys = [torch.empty_like(y.cpu()) for _ in range(world)]
    dist.all_gather(ys, y.cpu())
    assert all(torch.equal(ys[0], t) for t in ys), 'ranks disagree'
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print('tile-parallel GPU rehearsal OK')


if __name__ == '__main__':
    main()
