#!/usr/bin/env python3
"""Headline benchmark: output megapixels/s of RealESRGAN-x4plus (RRDBNet, 23 blocks) on 1080p input.

Contract (see the task statement): ``python bench.py --gpus N --steps K --warmup W`` prints ONE JSON line.
A *step* is one full forward pass of the hot path over one synthetic 3x1080x1920 frame per GPU
(1920x1080 -> 7680x4320 = 33.18 output MP), input resident in HBM before the timed region, output left in HBM.

N > 1 (launched by torch.distributed.run, one rank per GPU): tile-parallel inference (resselt_amd/tiling.py) of ONE
image made of N 1080p tiles (grid chosen by choose_grid: 1x2, 2x2, 2x4): every rank upsamples its own tile plus a
32-pixel input halo (weak scaling; no activation exchange -- the path shards by independent tiles) and
the upscaled tiles are reassembled on every rank by an RCCL all-gather over xGMI, which is inside the timed region.
``value`` = output MP of the whole image / max-over-ranks time.

Extra objects on the JSON line:
  roofline      dominant kernel (the fused conv kernel family = every launch of the forward), MFMA-bound:
                algorithmic FLOP (2,240,856 per output pixel) / HIP-event time of the launches
  roofline_hbm  the same time priced against the layer-wise HBM model of SURVEY.md §8d (7,790 B / output px)
  cpu_baseline  the CPU oracle (oracle/rrdbnet.py, a restatement of the reference forward; the reference
                cannot travel to the GPU box) timed on the host cores on a bounded 256x256 crop
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

FLOP_PER_OUT_PX = 2_240_856  # RRDBNet-23 x4, 2*MAC per output pixel
HBM_B_PER_OUT_PX = 7_790  # layer-wise bf16 byte model
MFMA_PEAK_TFLOPS = 2500.0  # dense bf16, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--height', type=int, default=1080)
    ap.add_argument('--width', type=int, default=1920)
    ap.add_argument('--blocks', type=int, default=23)
    ap.add_argument('--precision', default='bf16x3', choices=['bf16x3', 'bf16'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-crop', type=int, default=256)
    ap.add_argument('--halo', type=int, default=32)
    return ap.parse_args()


def cpu_baseline(sd, crop: int) -> dict:
    """Time the CPU oracle on a bounded crop of the same workload (reported baseline, not a target)."""
    from oracle.rrdbnet import rrdbnet_forward
    from resselt_amd.utils import synth

    x = synth.synth_input((1, 3, crop, crop), seed=0)
    threads = torch.get_num_threads()
    best = None
    with torch.no_grad():
        for _ in range(2):
            t0 = time.perf_counter()
            y = rrdbnet_forward(sd, x)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
    out_mp = y.shape[-1] * y.shape[-2] / 1e6
    return {
        'value': round(out_mp / best, 4),
        'unit': 'output megapixels/s',
        'cores': threads,
        'kind': 'port',
        'sample': f'oracle/rrdbnet.py fp32 on one 3x{crop}x{crop} crop of the synthetic frame (best of 2, {best:.2f} s), torch {torch.__version__} CPU',
    }


def measured_traffic():
    """HBM bytes per conv launch from the committed rocprofv3 PMC summary (tools/hbm_traffic.py; separate FETCH_SIZE / WRITE_SIZE
    passes of this same command, gfx950 x2 FETCH correction).  Counters cannot be read from inside the timed process."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_hbm_traffic.json')))
    if not files:
        return None, None
    rec = json.load(open(files[-1]))
    return rec.get('traffic_GB_per_launch', 0) * 1e9, os.path.basename(files[-1])


def main():
    args = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch multi-GPU runs with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the engine has no CPU path')
    # RSA_DIST_BACKEND=gloo rehearses the N > 1 code path with several ranks on ONE GPU (RCCL refuses two ranks per device)
    backend = os.environ.get('RSA_DIST_BACKEND', 'nccl')
    dev = torch.device('cuda', local_rank if backend == 'nccl' else local_rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)

    import resselt_amd
    from resselt_amd.utils import synth

    sd = synth.rrdbnet_state_dict(nb=args.blocks, seed=0)
    model = resselt_amd.load_from_state_dict(dict(sd)).to(dev)
    model.precision = args.precision
    H, W = args.height, args.width
    from resselt_amd.tiling import TileParallel, choose_grid

    rows, cols = choose_grid(world, world * H, W) if world > 1 else (1, 1)
    # the same full input image on every rank (3 channels: cheap to replicate); each rank owns one HxW tile of it
    x = synth.synth_input((1, 3, rows * H, cols * W), seed=0).to(dev)
    runner = TileParallel(model, scale=4, halo=args.halo, grid=(rows, cols)) if world > 1 else model

    def step():
        return runner(x)  # N > 1: tile forward + RCCL all-gather; every rank ends with the whole upscaled image

    for _ in range(args.warmup):
        step()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    barrier()
    t0 = time.perf_counter()
    ev0.record()
    y = None
    for _ in range(args.steps):
        y = step()
    ev1.record()
    barrier()
    dt = time.perf_counter() - t0
    ev_ms = ev0.elapsed_time(ev1)

    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    total_out_px = y.shape[-1] * y.shape[-2] * y.shape[0]  # whole image (on every rank after the all-gather)
    out_px = total_out_px // world  # one tile
    ms_per_step = dt / args.steps * 1e3
    value = total_out_px / 1e6 / (dt / args.steps)

    if rank == 0:
        # kernel-only time of the conv launches of one forward, from HIP events on the launch stream
        k0 = torch.cuda.Event(enable_timing=True)
        k1 = torch.cuda.Event(enable_timing=True)
        xt = x[:, :, :H, :W].contiguous()
        model(xt)
        torch.cuda.synchronize()
        k0.record()
        for _ in range(args.steps):
            model(xt)
        k1.record()
        torch.cuda.synchronize()
        kern_s = k0.elapsed_time(k1) / 1e3 / args.steps
        n_launch = model.launches_per_forward()
        macs = model.macs_per_input_pixel()
        if args.blocks == 23:
            assert 2 * macs == FLOP_PER_OUT_PX * 16, macs  # cross-check with SURVEY.md 8d
        flop = 2 * macs * H * W
        achieved_tf = flop / kern_s / 1e12
        achieved_gbs = HBM_B_PER_OUT_PX * out_px / kern_s / 1e9
        layout_bytes = model.conv_bytes_per_forward()  # same layer-wise model, priced in this engine's split-plane/f32-map layouts
        traffic, traffic_src = measured_traffic() if (args.precision == 'bf16x3' and (H, W, args.blocks) == (1080, 1920, 23)) else (None, None)
        res = {
            'metric': 'output megapixels/sec, RealESRGAN-x4plus 1080p\u21924K, 1/2/4/8 MI355X',
            'value': round(value, 3),
            'unit': 'output megapixels/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': round(ms_per_step, 3),
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'bf16',
            'data': 'synthetic',
            'config': {
                'workload': f'RealESRGAN-x4plus (RRDBNet nf64 nb{args.blocks} gc32 x4), 1x3x{H}x{W} fp32 frame per GPU -> 1x3x{4 * H}x{4 * W}, '
                + ('bf16 MFMA operands split hi+lo (3 products), f32 accumulate/residual' if args.precision == 'bf16x3' else 'plain bf16 MFMA operands, f32 accumulate/residual')
                + ', synthetic uniform(+-1/sqrt(fan_in)) weights',
                'precision': args.precision,
                'tile_parallel': f'{rows}x{cols} tiles of {H}x{W} (+{args.halo} px input halo), one per GPU, RCCL all-gather of fp32 output tiles' if world > 1 else 'single tile',
                'launches_per_step': n_launch,
            },
            'roofline': {
                'bound': 'mfma',
                'achieved': round(achieved_tf, 2),
                'peak': MFMA_PEAK_TFLOPS,
                'unit': 'TFLOP/s',
                'frac': round(achieved_tf / MFMA_PEAK_TFLOPS, 4),
                'traffic': traffic,
                'traffic_unit': 'HBM bytes per launch (avg over the 351 conv launches), from ' + traffic_src if traffic else None,
                'kernel': 'rsa::conv_kernel<KS,NCT,PROD,UP,OUTK> (all 351 conv launches of one forward)',
                'avg_launch_us': None if not n_launch else round(kern_s / n_launch * 1e6, 2),
                'mfma_issued_frac': round(achieved_tf * (3 if args.precision == 'bf16x3' else 1) / MFMA_PEAK_TFLOPS, 4),
            },
            'roofline_hbm': {
                'bound': 'hbm',
                'achieved': round(achieved_gbs, 1),
                'peak': HBM_PEAK_GBS,
                'unit': 'GB/s',
                'frac': round(achieved_gbs / HBM_PEAK_GBS, 4),
                'model': 'layer-wise bf16 bytes, 7,790 B per output pixel (SURVEY.md 8d)',
                'layout_bytes_per_launch': None if not layout_bytes else round(layout_bytes / n_launch),
                'layout_achieved': None if not layout_bytes else round(layout_bytes / kern_s / 1e9, 1),
                'layout_note': 'the same every-operand-once model priced in the engine layouts (hi+lo bf16 planes = 4 B/channel for 3 products, f32 residual maps); '
                'compare roofline.traffic against layout_bytes_per_launch',
            },
            'event_ms_per_step': round(ev_ms / args.steps, 3),
        }
        if world == 1 and not args.no_cpu_baseline:
            res['cpu_baseline'] = cpu_baseline(sd, args.cpu_crop)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
